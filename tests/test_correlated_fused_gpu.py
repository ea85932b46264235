"""The correlated update without a host round trip (cpm_selection_*, cpm_trace_selected, cpm_splat_delta) against the oracle
and against the entry points it fuses: same importances, same ascending index list, same photons bit for bit; the light
volume within the atomic splat's tolerance (rtol 1e-3, atol 2e-5 max: what the reference's own - old / + new splats are held to)."""
import numpy as np
import pytest

from test_parity_gpu import _t, _n, bits, _light_setup

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.402823466e+38)
UNCHANGED = 2147483647


def _traced_setup(cpm, oracle, n_side, vdim, max_inter=1, tf_pts=None):
    from oracle_binding import OTraceParams
    S = cpm.synthetic
    n = n_side * n_side
    vol_np = S.heterogeneous_volume(vdim)
    tf = S.tf_from_points(tf_pts or [(0.0, 1, 1, 1, 0.0), (0.6, 1, 1, 1, 0.0), (1.0, 1, 1, 1, 0.4)])
    d, o, u, v, area = _light_setup(cpm, (0.3, 0.5, -1.0))
    s = oracle.uniform_samples_2d(n_side, n_side)
    ls = oracle.directional_light_samples(s, (1, 1, 1), d, o, u, v, area)
    isect = oracle.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = oracle.glibc_rand_sequence(0, n)
    oracle.seed_streams(st, 1 << 40)
    p = OTraceParams()
    p.material[0] = 0.3
    p.step_size = 1 / vdim
    p.n_light_samples = n
    p.max_interactions = max_inter
    p.total_photons = n
    ph = np.zeros((n * max_inter, 8), np.float32)
    oracle.trace(oracle.volume(vol_np), tf, S.UNIT_CUBE_AABB, p, ls, isect, st, ph)
    return vol_np, tf, ls, isect, st, ph


@pytest.mark.parametrize("n_side,max_inter,fix,sparsity", [(96, 1, False, 0.5), (96, 3, False, 0.9), (61, 1, True, 0.97), (200, 1, False, 1.0),
                                                           (128, 1, False, 0.0)])
def test_importance_select_equals_importance_plus_select_changed(ctx, oracle, cpm, n_side, max_inter, fix, sparsity):
    """One light whose range starts at photon 5: importances as cpm_photon_importance / the oracle leave them, the list the
    changed photons ascending, the count on the device and in the mailbox."""
    from oracle_binding import default_matrices
    vdim, region = 64, 8
    n = n_side * n_side
    _, _, ls, isect, _, ph = _traced_setup(cpm, oracle, n_side, vdim, max_inter)
    gd = (vdim // region,) * 3
    rng = np.random.default_rng(9)
    grid = rng.random(gd[0] * gd[1] * gd[2], dtype=np.float32)
    grid[rng.random(grid.size) < sparsity] = 0          # sparsity 1.0: an all-zero grid (nothing selected); 0.0: dense
    t2i, _ = default_matrices((vdim,) * 3)
    imp0 = np.full(n + 10, UNCHANGED, np.uint32)
    imp0[rng.integers(0, n, 7)] -= 3                       # keys left over from an earlier, budget-limited evaluation
    imp_o = imp0.copy()
    oracle.photon_importance(grid, gd, (region,) * 3, t2i, ph, 5, ls, isect, n - 5, max_inter, n, imp_o, fix_exit_point=fix)
    want_idx, want_cnt = oracle.select_changed(imp_o[: n])
    sel = ctx.selection_create(n + 10)
    imp_d = _t(ctx, imp0)
    idx_d = ctx.torch.full((n + 10,), -1, dtype=ctx.torch.int32, device=ctx.device)
    for rep in range(2):                                   # a second selection on the same object: epochs advance
        imp_d.copy_(_t(ctx, imp0))
        sel.begin()
        sel.photon_importance(_t(ctx, grid), gd, (float(region),) * 3, t2i.tolist(), _t(ctx, ph), 5, _t(ctx, ls), _t(ctx, isect), n - 5,
                              max_inter, n, imp_d, fix_exit_point=fix)
        sel.finish(idx_d)
        got_cnt = sel.count()
        assert np.array_equal(_n(imp_d, np.uint32), imp_o)
        # the selection covers the light's range [5, n): keys outside it are not looked at
        in_range = want_idx[:want_cnt]
        in_range = in_range[(in_range >= 5) & (in_range < n)]
        assert got_cnt == in_range.size
        assert np.array_equal(_n(idx_d, np.uint32)[:got_cnt], in_range)
    if sparsity == 1.0:
        assert got_cnt <= 7
    else:
        assert got_cnt > 0
    sel.close()


def test_importance_select_two_lights_and_equal_importance(ctx, oracle, cpm):
    """Two lights sharing the photon buffer (ranges that do not end on tile boundaries) and the equal-importance detector."""
    from oracle_binding import default_matrices
    vdim, region, n_side = 64, 8, 90
    n1 = n_side * n_side
    _, _, ls, isect, _, ph1 = _traced_setup(cpm, oracle, n_side, vdim)
    n2 = 5000                                             # the second light: the first 5000 samples again, photons behind the first's
    N = n1 + n2
    ph = np.concatenate([ph1, ph1[:n2]])
    gd = (vdim // region,) * 3
    rng = np.random.default_rng(3)
    grid = rng.random(gd[0] * gd[1] * gd[2], dtype=np.float32)
    grid[rng.random(grid.size) < 0.8] = 0
    t2i, _ = default_matrices((vdim,) * 3)
    imp_o = np.full(N, UNCHANGED, np.uint32)
    oracle.photon_importance(grid, gd, (region,) * 3, t2i, ph, 0, ls, isect, n1, 1, N, imp_o)
    oracle.photon_importance(grid, gd, (region,) * 3, t2i, ph, n1, ls[:n2].copy(), isect[:n2].copy(), n2, 1, N, imp_o)
    want_idx, want_cnt = oracle.select_changed(imp_o)
    sel = ctx.selection_create(N)
    imp_d = _t(ctx, np.full(N, UNCHANGED, np.uint32))
    idx_d = ctx.torch.zeros(N, dtype=ctx.torch.int32, device=ctx.device)
    sel.begin()
    dgrid, dph = _t(ctx, grid), _t(ctx, ph)
    sel.photon_importance(dgrid, gd, (float(region),) * 3, t2i.tolist(), dph, 0, _t(ctx, ls), _t(ctx, isect), n1, 1, N, imp_d)
    sel.photon_importance(dgrid, gd, (float(region),) * 3, t2i.tolist(), dph, n1, _t(ctx, ls[:n2].copy()), _t(ctx, isect[:n2].copy()), n2, 1, N, imp_d)
    sel.finish(idx_d)
    cnt = sel.count()
    assert cnt == want_cnt > 0
    assert np.array_equal(_n(imp_d, np.uint32), imp_o)
    assert np.array_equal(_n(idx_d, np.uint32)[:cnt], want_idx[:cnt])
    # equal importance: every (100 / p)-th photon
    for pct, it in ((25, 3), (1, 0), (100, 7), (7, 2)):
        a = np.full(N, UNCHANGED, np.uint32)
        oracle.photon_importance_equal(1000, N - 1000, pct, it, a)
        widx, wcnt = oracle.select_changed(a)
        d = _t(ctx, np.full(N, UNCHANGED, np.uint32))
        sel.begin()
        sel.photon_importance_equal(1000, N - 1000, pct, it, d)
        sel.finish(idx_d)
        assert sel.count() == wcnt
        assert np.array_equal(_n(d, np.uint32), a)
        assert np.array_equal(_n(idx_d, np.uint32)[:wcnt], widx[:wcnt])
    # an empty selection publishes a count of zero
    sel.begin()
    sel.finish(idx_d)
    assert sel.count() == 0
    sel.close()


@pytest.mark.parametrize("max_inter", [1, 3])
def test_trace_selected_and_splat_delta(ctx, oracle, cpm, max_inter):
    """cpm_trace_selected == cpm_trace over the same indices (photons, kept old records, importance reset);
    cpm_splat_delta == the two cpm_splat_selected launches within the atomic tolerance; the brick mask == cpm_mark_touched_bricks x 2;
    apply_below leaves the volume alone."""
    S, P, B = cpm.synthetic, cpm.pipeline, cpm.binding
    torch = ctx.torch
    vdim, n_side, gdim = 64, 128, 32
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    vol_np = S.heterogeneous_volume(vdim)
    fr = P.PhotonFrame(ctx, vol_np, S.tf_from_points(base), n_side, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0),
                       max_interactions=max_inter, material=(0.3, 0, 0, 0))
    fr.trace()
    n = fr.n
    before = fr.photons.clone()
    fr.splat(all_interactions=True)
    lv0 = fr.light_volume.clone()
    fr.tf.update(S.tf_from_points(edit))
    # select every 4th photon through the fused equal-importance selection
    sel = ctx.selection_create(n)
    imp = _t(ctx, np.full(n, UNCHANGED, np.uint32))
    idx = torch.zeros(n, dtype=torch.int32, device=ctx.device)
    sel.begin()
    sel.photon_importance_equal(0, n, 25, 1, imp)
    sel.finish(idx)
    cnt = sel.count()
    assert cnt == n // 4
    # reference: cpm_trace over the same list
    ref_photons = before.clone()
    ctx.trace(fr.vol, fr.tf, fr.aabb, fr.params, fr.light_samples, fr.isect, fr.rng, ref_photons, recompute_indices=idx, n_recompute=cnt)
    old = torch.full((max_inter * n, 8), 7.0, dtype=torch.float32, device=ctx.device)
    ctx.trace_selected(fr.vol, fr.tf, fr.aabb, fr.params, fr.light_samples, fr.isect, idx, sel, n, fr.rng, fr.photons,
                       old_photons=old, reset_importances=imp)
    assert np.array_equal(bits(_n(fr.photons)), bits(_n(ref_photons)))
    assert (bits(_n(fr.photons)) != bits(_n(before))).any()
    assert (_n(imp, np.uint32) == UNCHANGED).all()
    sel_idx = _n(idx, np.uint32)[:cnt].astype(np.int64)
    b = _n(before).reshape(max_inter, n, 8)
    o = _n(old).reshape(max_inter, n, 8)
    assert np.array_equal(bits(o[:, :cnt]), bits(b[:, sel_idx]))
    assert (o[:, cnt:] == 7.0).all()                       # nothing beyond the count is written
    # light volume: delta == -old +new selected splats
    lv_ref = lv0.clone()
    ctx.splat_selected(before, idx, cnt, fr.grid, fr.radius, fr.scale, -1.0, n, max_inter, lv_ref)
    ctx.splat_selected(fr.photons, idx, cnt, fr.grid, fr.radius, fr.scale, 1.0, n, max_inter, lv_ref)
    nb = ((gdim + 3) // 4) ** 3
    mask_ref = torch.zeros(nb, dtype=torch.uint8, device=ctx.device)
    ctx.mark_touched_bricks(before, idx, cnt, n, max_inter, fr.grid, fr.radius, mask_ref)
    ctx.mark_touched_bricks(fr.photons, idx, cnt, n, max_inter, fr.grid, fr.radius, mask_ref)
    lv = lv0.clone()
    mask = torch.zeros(nb, dtype=torch.uint8, device=ctx.device)
    ctx.splat_delta(old, n, fr.photons, idx, sel, n, fr.grid, fr.radius, fr.scale, n, max_inter, lv, brick_mask=mask)
    full = torch.zeros_like(lv0)
    ctx.splat(fr.photons, n * max_inter, fr.grid, fr.radius, fr.scale, full)
    np.testing.assert_allclose(_n(lv), _n(lv_ref), rtol=1e-3, atol=2e-5 * float(full.max()))
    np.testing.assert_allclose(_n(lv), _n(full), rtol=1e-3, atol=2e-5 * float(full.max()))
    assert float((lv - lv0).abs().max()) > 0
    # a photon whose record did not change adds nothing and marks nothing: the fused mask is a subset of the reference's
    # that covers every brick whose voxels changed
    m, mr = _n(mask, np.uint8), _n(mask_ref, np.uint8)
    assert ((m == 1) <= (mr == 1)).all() and m.sum() > 0
    changed_vox = (_n(lv) != _n(lv0)).reshape(gdim, gdim, gdim)
    bz, by, bx = np.nonzero(changed_vox)
    assert (m.reshape((gdim + 3) // 4, (gdim + 3) // 4, (gdim + 3) // 4)[bz // 4, by // 4, bx // 4] == 1).all()
    # apply_below: the launch stands aside when the count reaches the threshold
    lv2 = lv0.clone()
    ctx.splat_delta(old, n, fr.photons, idx, sel, n, fr.grid, fr.radius, fr.scale, n, max_inter, lv2, apply_below=cnt)
    assert torch.equal(lv2, lv0)
    ctx.splat_delta(old, n, fr.photons, idx, sel, n, fr.grid, fr.radius, fr.scale, n, max_inter, lv2, apply_below=cnt + 1)
    np.testing.assert_allclose(_n(lv2), _n(lv_ref), rtol=1e-3, atol=2e-5 * float(full.max()))
    sel.close()


@pytest.mark.parametrize("max_inter", [1, 2])
def test_fused_update_equals_legacy_update(ctx, oracle, cpm, max_inter):
    """CorrelatedPhotonMapper: the fused evaluation and the launch-by-launch one (one host read in the middle) leave the same
    photons, importances and selection, and light volumes within the splat tolerance; a second edit (the revert) as well."""
    S, P = cpm.synthetic, cpm.pipeline
    vol_np = S.heterogeneous_volume(64)
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0), tf_points=base, max_interactions=max_inter, material=(0.3, 0, 0, 0))
    a = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), 160, (32, 32, 32), incremental_threshold_percent=100.0, **kw)
    b = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), 160, (32, 32, 32), incremental_threshold_percent=100.0, **kw)
    b.fused = False
    c2 = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), 160, (32, 32, 32), incremental_threshold_percent=100.0, **kw)
    c2.retrace_in_importance_pass = False            # select + compact + cpm_trace_selected instead of the one-launch form
    a.full_frame(); b.full_frame(); c2.full_frame()
    assert a.prev_photons is None and b.prev_photons is not None     # the fused configuration keeps no 32 MiB snapshot
    for pts in (edit, base, edit):
        a.set_transfer_function(pts); b.set_transfer_function(pts)
        c2.set_transfer_function(pts)
        na, nb = a.correlated_update(), b.correlated_update()
        assert c2.correlated_update() == na
        assert np.array_equal(bits(_n(c2.photons)), bits(_n(a.photons)))
        assert np.array_equal(_n(c2.indices, np.uint32)[:na], _n(a.indices, np.uint32)[:na])
        assert np.array_equal(_n(c2.importance, np.uint32), _n(a.importance, np.uint32))
        assert na == nb > 0
        assert a.last_path == b.last_path == "incremental"
        assert np.array_equal(bits(_n(a.photons)), bits(_n(b.photons)))
        assert np.array_equal(_n(a.indices, np.uint32)[:na], _n(b.indices, np.uint32)[:nb])
        assert np.array_equal(_n(a.importance, np.uint32), _n(b.importance, np.uint32))
        full = P.PhotonFrame(ctx, vol_np, S.tf_from_points(pts), 160, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0),
                             max_interactions=max_inter, material=(0.3, 0, 0, 0))
        lv_full = _n(full.frame())
        assert np.array_equal(bits(_n(a.photons)), bits(_n(full.photons)))
        np.testing.assert_allclose(_n(a.light_volume), lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
        np.testing.assert_allclose(_n(b.light_volume), lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
    # above the incremental threshold the delta launch stands aside and the volume is rebuilt
    c = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), 160, (32, 32, 32), incremental_threshold_percent=0.01, **kw)
    c.full_frame()
    c.set_transfer_function(edit)
    assert c.correlated_update() == na and c.last_path == "full"
    np.testing.assert_allclose(_n(c.light_volume), lv_full, rtol=2e-5, atol=1e-5 * float(lv_full.max()))
    # switching an instance to the legacy path after fused updates re-creates the snapshot it needs
    a.fused = False
    a.set_transfer_function(base); b.set_transfer_function(base)
    assert a.correlated_update() == b.correlated_update()
    assert np.array_equal(bits(_n(a.photons)), bits(_n(b.photons)))
    np.testing.assert_allclose(_n(a.light_volume), _n(b.light_volume), rtol=1e-3, atol=2e-5 * float(lv_full.max()))


@pytest.mark.parametrize("max_inter,dtype", [(1, np.uint8), (3, np.uint8), (1, np.uint16), (2, np.float32)])
def test_importance_retrace_equals_select_then_trace_selected(ctx, oracle, cpm, max_inter, dtype):
    """cpm_photon_importance_retrace against cpm_photon_importance_select + cpm_selection_finish + cpm_trace_selected on the same
    inputs: photons, index list, count and importance keys bit for bit; the replaced records at the photons' own indices."""
    S, P = cpm.synthetic, cpm.pipeline
    torch = ctx.torch
    vdim, n_side, region = 64, 128, 8
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    vol8 = S.heterogeneous_volume(vdim)
    vol_np = vol8 if dtype == np.uint8 else (vol8.astype(np.uint16) * 257 if dtype == np.uint16 else (vol8.astype(np.float32) / np.float32(255)))
    fr = P.PhotonFrame(ctx, vol_np, S.tf_from_points(base), n_side, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0),
                       max_interactions=max_inter, material=(0.3, 0, 0, 0))
    fr.trace()
    n = fr.n
    before = fr.photons.clone()
    fr.tf.update(S.tf_from_points(edit))
    gd = (vdim // region,) * 3
    rng = np.random.default_rng(21)
    grid = rng.random(gd[0] * gd[1] * gd[2], dtype=np.float32)
    grid[rng.random(grid.size) < 0.93] = 0
    dgrid = _t(ctx, grid)
    t2i = list(fr.vol.desc.texture_to_index)
    imp0 = np.full(n, UNCHANGED, np.uint32)
    imp0[rng.integers(0, n, 5)] -= 7                          # keys left from an earlier evaluation: re-traced and reset as well
    # reference: the three launches
    sel = ctx.selection_create(n)
    imp_a = _t(ctx, imp0)
    idx_a = torch.zeros(n, dtype=torch.int32, device=ctx.device)
    ph_a = before.clone()
    old_a = torch.zeros((max_inter * n, 8), dtype=torch.float32, device=ctx.device)
    sel.begin()
    sel.photon_importance(dgrid, gd, (float(region),) * 3, t2i, ph_a, 0, fr.light_samples, fr.isect, n, max_inter, n, imp_a)
    sel.finish(idx_a)
    ctx.trace_selected(fr.vol, fr.tf, fr.aabb, fr.params, fr.light_samples, fr.isect, idx_a, sel, n, fr.rng, ph_a, old_photons=old_a,
                       reset_importances=imp_a)
    cnt_a = sel.count()
    assert 0 < cnt_a < n
    # one launch
    imp_b = _t(ctx, imp0)
    idx_b = torch.zeros(n, dtype=torch.int32, device=ctx.device)
    ph_b = before.clone()
    old_b = torch.full((max_inter * n, 8), 7.0, dtype=torch.float32, device=ctx.device)
    sel.begin()
    sel.photon_importance_retrace(dgrid, gd, (float(region),) * 3, t2i, fr.vol, fr.tf, fr.aabb, fr.params, fr.light_samples, fr.isect, imp_b,
                                  fr.rng, ph_b, old_b)
    sel.finish(idx_b)
    cnt_b = sel.count()
    assert cnt_b == cnt_a
    assert np.array_equal(_n(idx_b, np.uint32)[:cnt_b], _n(idx_a, np.uint32)[:cnt_a])
    assert np.array_equal(bits(_n(ph_b)), bits(_n(ph_a)))
    assert (bits(_n(ph_b)) != bits(_n(before))).any()
    assert np.array_equal(_n(imp_b, np.uint32), _n(imp_a, np.uint32)) and (_n(imp_b, np.uint32) == UNCHANGED).all()
    # the replaced records: at the photons' own indices, nothing else written
    sel_idx = _n(idx_b, np.uint32)[:cnt_b].astype(np.int64)
    ob = _n(old_b).reshape(max_inter, n, 8)
    bf = _n(before).reshape(max_inter, n, 8)
    assert np.array_equal(bits(ob[:, sel_idx]), bits(bf[:, sel_idx]))
    untouched = np.ones(n, bool)
    untouched[sel_idx] = False
    assert (ob[:, untouched] == 7.0).all()
    # light volume: the delta from the indexed old records == the delta from the compact ones
    fr.photons.copy_(before)
    fr.splat(all_interactions=True)
    lv0 = fr.light_volume.clone()
    lva, lvb = lv0.clone(), lv0.clone()
    ctx.splat_delta(old_a, n, ph_a, idx_a, sel, n, fr.grid, fr.radius, fr.scale, n, max_inter, lva)
    ctx.splat_delta(old_b, 0, ph_b, idx_b, sel, n, fr.grid, fr.radius, fr.scale, n, max_inter, lvb)
    np.testing.assert_allclose(_n(lvb), _n(lva), rtol=1e-3, atol=2e-5 * float(lv0.max()))
    assert float((lvb - lv0).abs().max()) > 0
    sel.close()


def test_many_updates_cross_the_measured_selections(ctx, cpm):
    """The one-launch importance + re-trace takes its tiles in the order of their measured costs, re-measured every 32nd
    selection (cpm_selection_begin): 70 alternating TF edits -- index order, first measured order, two re-measurements -- leave
    the photons, importances and index lists of a mapper that runs the launch-by-launch branch, bit for bit."""
    S, P = cpm.synthetic, cpm.pipeline
    torch = ctx.torch
    vol = S.heterogeneous_volume(64)
    base = list(S.WORKSPACE_TF_POINTS)
    edit = list(base); edit[3] = (0.26,) + base[3][1:]
    mk = lambda: P.CorrelatedPhotonMapper(ctx, vol, S.workspace_tf(), 300, (32,) * 3, light_travel_direction=(0.3, 0.5, -1.0), tf_points=base)
    a, b = mk(), mk()
    b.fused = False
    for cm in (a, b):
        cm.incremental_threshold_percent = 100
        cm.full_frame()
    for rep in range(70):
        pts = edit if rep % 2 == 0 else base
        for cm in (a, b):
            cm.set_transfer_function(pts)
        ka, kb = a.correlated_update(), b.correlated_update()
        assert ka == kb and ka > 0
        torch.cuda.synchronize()
        assert torch.equal(a.photons.view(torch.int32), b.photons.view(torch.int32)), f"update {rep}"
        assert torch.equal(a.importance, b.importance)
        assert torch.equal(a.indices[:ka], b.indices[:kb])
    np.testing.assert_allclose(a.light_volume.cpu().numpy(), b.light_volume.cpu().numpy(), rtol=1e-3, atol=2e-5 * float(b.light_volume.max()))


@pytest.mark.parametrize("n_cells", [1, 63, 64, 65, 4096, 32768 + 17])
def test_importance_tf_occupancy_bits(ctx, cpm, n_cells):
    """cpm_importance_tf_occupancy: the same importances as cpm_importance_tf, and bit c of the occupancy words set exactly
    where importance[c] is not +0.0f (what the selection's own mask launch computes); whole 64-cell groups are written."""
    torch = ctx.torch
    S = cpm.synthetic
    rng = np.random.default_rng(n_cells)
    lo = rng.integers(0, 65535, n_cells).astype(np.uint16)
    hi = np.maximum(lo, rng.integers(0, 65535, n_cells).astype(np.uint16))
    mm = torch.from_numpy(np.stack([lo, hi], 1).copy().view(np.int16)).to(ctx.device)
    pos = np.array([0.0, 0.2, 0.3, 0.6, 1.0], np.float32)
    col = np.zeros((5, 4), np.float32)
    col[2] = (0.5, 0.1, 0.0, 0.25)   # the difference TF is non-zero only around 0.3: many cells get exactly +0
    want = torch.zeros(n_cells, dtype=torch.float32, device=ctx.device)
    ctx.importance_tf(mm, n_cells, pos, col, want)
    got = torch.full((n_cells,), -1.0, dtype=torch.float32, device=ctx.device)
    words = 2 * ((n_cells + 63) // 64)
    occ = torch.full((words,), -1, dtype=torch.int32, device=ctx.device)
    ctx.importance_tf(mm, n_cells, pos, col, got, occupancy=occ)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    w = want.cpu().numpy()
    assert (w.view(np.uint32) == 0).any() or n_cells < 64
    bits_ = np.unpackbits(occ.cpu().numpy().view(np.uint8), bitorder="little")
    expect = np.zeros(words * 32, np.uint8)
    expect[:n_cells] = (w.view(np.uint32) != 0).astype(np.uint8)
    assert np.array_equal(bits_, expect)


def test_failed_select_call_publishes_an_empty_selection(ctx, oracle, cpm):
    """tiles appended by a call that then fails must be taken back -- the compaction would otherwise sum counts no
    kernel wrote.  With the failure injected after the append (what a refused launch does), the selection's finish reports it
    and publishes a count of 0, a later light of the same selection included; the next selection is unaffected."""
    from oracle_binding import default_matrices
    B = cpm.binding
    vdim, region, n_side = 64, 8, 90
    n = n_side * n_side
    _, _, ls, isect, _, ph = _traced_setup(cpm, oracle, n_side, vdim)
    gd = (vdim // region,) * 3
    rng = np.random.default_rng(5)
    grid = rng.random(gd[0] * gd[1] * gd[2], dtype=np.float32)
    grid[rng.random(grid.size) < 0.7] = 0
    t2i, _ = default_matrices((vdim,) * 3)
    sel = ctx.selection_create(2 * n)
    idx_d = ctx.torch.full((2 * n,), -1, dtype=ctx.torch.int32, device=ctx.device)
    dgrid, dph2 = _t(ctx, grid), _t(ctx, np.concatenate([ph, ph]))
    dls, dis = _t(ctx, ls), _t(ctx, isect)
    args = (dgrid, gd, (float(region),) * 3, t2i.tolist(), dph2)

    def fresh_keys():
        return _t(ctx, np.full(2 * n, UNCHANGED, np.uint32))
    # reference: both lights selected
    imp = fresh_keys()
    sel.begin()
    sel.photon_importance(*args, 0, dls, dis, n, 1, 2 * n, imp)
    sel.photon_importance(*args, n, dls, dis, n, 1, 2 * n, imp)
    sel.finish(idx_d)
    want = sel.count()
    want_idx = _n(idx_d, np.uint32)[:want].copy()
    assert want > 0
    for failing in (0, 1):          # the first or the second light's call fails
        imp = fresh_keys()
        sel.begin()
        for k in range(2):
            if k == failing:
                ctx.lib.cpm_debug_fail_next_select(ctx.h, 1)
                with pytest.raises(B.CpmError):
                    sel.photon_importance(*args, k * n, dls, dis, n, 1, 2 * n, imp)
            else:
                sel.photon_importance(*args, k * n, dls, dis, n, 1, 2 * n, imp)
        with pytest.raises(B.CpmError):
            sel.finish(idx_d)
        assert sel.count() == 0
    # the selection object is as good as new
    imp = fresh_keys()
    sel.begin()
    sel.photon_importance(*args, 0, dls, dis, n, 1, 2 * n, imp)
    sel.photon_importance(*args, n, dls, dis, n, 1, 2 * n, imp)
    sel.finish(idx_d)
    assert sel.count() == want and np.array_equal(_n(idx_d, np.uint32)[:want], want_idx)
    sel.close()


def test_importance_grid_beyond_the_lds_budget(ctx, oracle, cpm):
    """An importance grid whose occupancy bits (1 per cell) do not fit beside the tracer's LUTs in the workgroup's LDS (80^3 cells =
    64 000 bytes of bits): select and the one-launch retrace walk the grid without the bits -- same importances, same list."""
    from oracle_binding import default_matrices
    vdim, region, n_side = 80, 1, 64
    n = n_side * n_side
    vol_np, tf, ls, isect, st, ph = _traced_setup(cpm, oracle, n_side, vdim)
    gd = (vdim // region,) * 3
    rng = np.random.default_rng(11)
    grid = (rng.random(gd[0] * gd[1] * gd[2], dtype=np.float32) * np.float32(0.02)).astype(np.float32)
    grid[rng.random(grid.size) < 0.9] = 0
    t2i, _ = default_matrices((vdim,) * 3)
    imp_o = np.full(n, UNCHANGED, np.uint32)
    oracle.photon_importance(grid, gd, (region,) * 3, t2i, ph, 0, ls, isect, n, 1, n, imp_o)
    want_idx, want_cnt = oracle.select_changed(imp_o)
    assert 0 < want_cnt < n
    sel = ctx.selection_create(n)
    imp_d = _t(ctx, np.full(n, UNCHANGED, np.uint32))
    idx_d = ctx.torch.zeros(n, dtype=ctx.torch.int32, device=ctx.device)
    sel.begin()
    sel.photon_importance(_t(ctx, grid), gd, (float(region),) * 3, t2i.tolist(), _t(ctx, ph), 0, _t(ctx, ls), _t(ctx, isect), n, 1, n, imp_d)
    sel.finish(idx_d)
    assert sel.count() == want_cnt
    assert np.array_equal(_n(imp_d, np.uint32), imp_o) and np.array_equal(_n(idx_d, np.uint32)[:want_cnt], want_idx[:want_cnt])
    # the one-launch form over the same grid: same list; the selected photons are re-traced (same TF: the same photons again)
    B = cpm.binding
    p = B.TraceParams()
    p.material[0] = 0.3
    p.step_size = 1 / vdim
    p.n_light_samples = n
    p.max_interactions = 1
    p.total_photons = n
    dvol, dtf = ctx.volume_create(vol_np), ctx.tf_create(tf)
    imp_d = _t(ctx, np.full(n, UNCHANGED, np.uint32))
    dph, old = _t(ctx, ph), ctx.torch.zeros((n, 8), dtype=ctx.torch.float32, device=ctx.device)
    sel.begin()
    sel.photon_importance_retrace(_t(ctx, grid), gd, (float(region),) * 3, t2i.tolist(), dvol, dtf, cpm.synthetic.UNIT_CUBE_AABB, p, _t(ctx, ls),
                                  _t(ctx, isect), imp_d, _t(ctx, st), dph, old)
    sel.finish(idx_d)
    assert sel.count() == want_cnt and np.array_equal(_n(idx_d, np.uint32)[:want_cnt], want_idx[:want_cnt])
    assert np.array_equal(bits(_n(dph)), bits(ph))
    sel.close()
