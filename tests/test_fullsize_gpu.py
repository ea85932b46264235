"""BASELINE.json configs 2 - 5 at their FULL sizes, HIP path (through the C-ABI) against the oracle.

Round-1's full-size test compared the gather with the HIP atomic splat and sampled every 97th photon
(VERDICT r01, weak 2-3).  Here every photon, the bin's `order` / `cell_start` / compact records and the
whole light volume are compared with the oracle bit for bit (the oracle runs a full config-2 frame in
well under a second on the GPU box's host cores, a config-4 frame in a few seconds):

  config 2  256^3 u8, 1 048 576 photons, 128^3 grid
  config 4  512^3 u8, 4 194 304 photons, 256^3 grid (one GPU's view of it: > 64 Ki bricks, 4096-key sort
            tiles, the widest cell key)
  config 3  config 2 + the workspace TF's point 4 moved 0.2218 -> 0.26: importance grid, per-photon
            importances and the selection equal the oracle's; correlated re-trace == full re-trace
  config 5  256^3 time-varying sequence: GPU difference / min-max / time-varying importance equal the
            oracle's; the re-traced photons equal an oracle trace of the same indices in the new volume
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(3.402823466e+38)
LIGHT_DIR = (0.3, 0.5, -1.0)


def _n(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a.view(dtype) if dtype is not None else a


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _oracle_trace(oracle, vol_np, tf, fr, step, indices=None):
    from oracle_binding import OTraceParams
    import cpm_amd
    S = cpm_amd.synthetic
    ls, isect = _n(fr.light_samples), _n(fr.isect)
    rng = _n(fr.rng_initial, np.uint32).copy()
    if indices is not None:
        ls, isect, rng = ls[indices].copy(), isect[indices].copy(), rng[indices].copy()
    n = ls.shape[0]
    po = OTraceParams()
    po.step_size = step
    po.n_light_samples = n
    po.max_interactions = 1
    po.total_photons = n
    ph = np.zeros((n, 8), np.float32)
    oracle.trace(oracle.volume(vol_np), tf, S.UNIT_CUBE_AABB, po, ls, isect, rng, ph)
    return ph


def _full_frame_vs_oracle(ctx, oracle, cpm, vdim, n_side, gdim):
    S, P = cpm.synthetic, cpm.pipeline
    vol_np, tf = S.heterogeneous_volume(vdim), S.workspace_tf()
    fr = P.PhotonFrame(ctx, vol_np, tf, n_side, (gdim,) * 3, light_travel_direction=LIGHT_DIR)
    lv = _n(fr.frame()).copy()
    n = fr.n
    # trace: every photon
    ph_o = _oracle_trace(oracle, vol_np, tf, fr, 1.0 / vdim)
    assert np.array_equal(bits(_n(fr.photons)), bits(ph_o))
    stored = int((ph_o[:, 0] != FLT_MAX).sum())
    assert 0 < stored < n
    # bin: order, cell starts, compact records
    og = oracle.grid((gdim,) * 3, 1)
    o_order, o_cs, o_srt = oracle.bin(ph_o, n, og)
    assert np.array_equal(_n(fr.order, np.uint32), o_order)
    assert np.array_equal(_n(fr.cell_start, np.uint32), o_cs)
    assert int(o_cs[-1]) == stored
    assert np.array_equal(bits(_n(fr.sorted)), bits(o_srt))
    # gather: the whole light volume
    want = np.zeros(gdim ** 3, np.float32)
    oracle.gather(o_srt, o_cs, n, og, fr.radius, fr.scale, want)
    assert np.array_equal(bits(lv), bits(want))
    assert lv.sum() > 0
    # run-to-run reproducible
    assert np.array_equal(bits(_n(fr.frame())), bits(lv))
    # the tolerance-mode path (brick bin + tile gather, include/cpm/cpm.h "fast formulation") against the same oracle
    # volume, at the tolerance the splat tests state; and bitwise reproducible itself
    lvf = _n(fr.frame_fast()).copy()
    np.testing.assert_allclose(lvf, want, rtol=2e-5, atol=1e-5 * float(want.max()))
    assert np.array_equal(bits(_n(fr.frame_fast())), bits(lvf))
    # and the reference formulation (atomic splat) within its tolerance
    sp = _n(fr.splat(ctx.torch.zeros_like(fr.light_volume)))
    np.testing.assert_allclose(sp, want, rtol=1e-4, atol=1e-5 * float(want.max()))
    return fr


def test_config2_full_frame_equals_oracle(ctx, oracle, cpm):
    _full_frame_vs_oracle(ctx, oracle, cpm, 256, 1024, 128)


def test_config4_full_frame_equals_oracle(ctx, oracle, cpm):
    """512^3 volume, 2048^2 lattice, 256^3 grid: 262 144 bricks (gather_records2_kernel), 4096-key sort tiles."""
    _full_frame_vs_oracle(ctx, oracle, cpm, 512, 2048, 256)


def test_config3_tf_edit_full_size(ctx, oracle, cpm):
    from oracle_binding import default_matrices
    S, P = cpm.synthetic, cpm.pipeline
    vdim, gdim, n_side, region = 256, 128, 1024, 8
    vol_np = S.heterogeneous_volume(vdim)
    base = list(S.WORKSPACE_TF_POINTS)
    edit = list(base)
    edit[3] = (0.26,) + base[3][1:]
    for budget in (100.0, 25.0):
        cm = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), n_side, (gdim,) * 3, light_travel_direction=LIGHT_DIR,
                                      tf_points=base, max_incremental_percent=budget, region=region)
        cm.full_frame()
        before = _n(cm.photons).copy()
        pos, col = cm.set_transfer_function(edit)
        n = cm.correlated_update()
        assert 0 < n < cm.n
        # importance grid, per-photon importances and the selection against the oracle
        ovol = oracle.volume(vol_np)
        mm = oracle.volume_minmax(ovol, region)
        assert np.array_equal(_n(cm.minmax, np.uint16).reshape(-1), np.asarray(mm, np.uint16).reshape(-1))
        grid = oracle.importance_tf(mm, pos, col)
        assert np.array_equal(bits(_n(cm.importance_grid)), bits(grid))
        t2i, _ = default_matrices((vdim,) * 3)
        imp = np.full(cm.n, 2147483647, np.uint32)
        gd = (vdim // region,) * 3
        oracle.photon_importance(grid, gd, (float(region),) * 3, t2i, before, 0, _n(cm.light_samples), _n(cm.isect), cm.n, 1, cm.n, imp)
        oidx, ocnt = oracle.select_changed(imp)
        assert ocnt == n                               # everything changed fits both budgets here
        idx = np.sort(_n(cm.indices, np.uint32)[:n])
        assert np.array_equal(idx, np.sort(oidx[:n]))
        # correlated re-trace == full re-trace with the edited TF, photon for photon -- and == the oracle's trace
        after = _n(cm.photons)
        want = _oracle_trace(oracle, vol_np, S.tf_from_points(edit), cm, 1.0 / vdim)
        assert np.array_equal(bits(after), bits(want))
        changed = (bits(after) != bits(before)).any(axis=1)
        assert 0 < changed.sum() <= n and np.isin(np.nonzero(changed)[0], idx).all()
        # light volume: the incremental +- update equals a full gather of the new photons within fp32 tolerance
        assert cm.last_path == "incremental"
        og = oracle.grid((gdim,) * 3, 1)
        _, o_cs, o_srt = oracle.bin(want, cm.n, og)
        lv_full = np.zeros(gdim ** 3, np.float32)
        oracle.gather(o_srt, o_cs, cm.n, og, cm.radius, cm.scale, lv_full)
        np.testing.assert_allclose(_n(cm.light_volume), lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
        assert (_n(cm.importance, np.uint32) == 2147483647).all()


def test_config3_exact_incremental_full_size(ctx, oracle, cpm):
    """The exact form of the update (touched bricks re-gathered) lands on the oracle's full frame bit for bit."""
    S, P = cpm.synthetic, cpm.pipeline
    vdim, gdim, n_side = 256, 128, 1024
    vol_np = S.heterogeneous_volume(vdim)
    base = list(S.WORKSPACE_TF_POINTS)
    edit = list(base)
    edit[3] = (0.26,) + base[3][1:]
    cm = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), n_side, (gdim,) * 3, light_travel_direction=LIGHT_DIR,
                                  tf_points=base, exact_update=True)
    cm.full_frame()
    cm.set_transfer_function(edit)
    n = cm.correlated_update()
    assert n > 0 and cm.last_path == "exact incremental"
    want = _oracle_trace(oracle, vol_np, S.tf_from_points(edit), cm, 1.0 / vdim)
    og = oracle.grid((gdim,) * 3, 1)
    _, o_cs, o_srt = oracle.bin(want, cm.n, og)
    lv_full = np.zeros(gdim ** 3, np.float32)
    oracle.gather(o_srt, o_cs, cm.n, og, cm.radius, cm.scale, lv_full)
    assert np.array_equal(bits(_n(cm.light_volume)), bits(lv_full))


@pytest.mark.parametrize("steps", [list(range(0, 9)), list(range(8, 17)), list(range(16, 25)), list(range(24, 32))])   # all 31 transitions of the 32-step sequence
def test_config5_time_varying_full_size(ctx, oracle, cpm, steps):
    from oracle_binding import default_matrices
    S, P = cpm.synthetic, cpm.pipeline
    vdim, gdim, n_side, region = 256, 128, 1024, 8
    tfp = list(S.WORKSPACE_TF_POINTS)
    vols = [S.heterogeneous_volume(vdim, S.sequence_blob_center(t, 32)) for t in steps]
    cm = P.CorrelatedPhotonMapper(ctx, vols[0], S.tf_from_points(tfp), n_side, (gdim,) * 3, light_travel_direction=LIGHT_DIR,
                                  tf_points=tfp, incremental_threshold_percent=100.0, region=region)
    cm.full_frame()
    t2i, _ = default_matrices((vdim,) * 3)
    gd = (vdim // region,) * 3
    pts = sorted(tfp)
    pos = [p[0] for p in pts]
    col = [list(p[1:]) for p in pts]
    if pos[0] > 0.0:
        pos.insert(0, 0.0); col.insert(0, col[0])
    if pos[-1] < 1.0:
        pos.append(1.0); col.append(col[-1])
    pos, col = np.asarray(pos, np.float32), np.asarray(col, np.float32)
    for t in range(1, len(steps)):
        before = _n(cm.photons).copy()
        cm.set_volume(vols[t])
        oa, ob = oracle.volume(vols[t - 1]), oracle.volume(vols[t])
        diff = oracle.volume_difference(oa, ob, region)
        mm_a, mm_b = oracle.volume_minmax(oa, region), oracle.volume_minmax(ob, region)
        want_grid = oracle.importance_tf(mm_b, pos, col, prev=mm_a, diff=diff)
        assert np.array_equal(bits(_n(cm.importance_grid)), bits(want_grid))
        n = cm.correlated_update()
        assert 0 < n < cm.n
        imp = np.full(cm.n, 2147483647, np.uint32)
        oracle.photon_importance(want_grid, gd, (float(region),) * 3, t2i, before, 0, _n(cm.light_samples), _n(cm.isect), cm.n, 1,
                                 cm.n, imp)
        oidx, ocnt = oracle.select_changed(imp)
        assert ocnt == n
        idx = np.sort(_n(cm.indices, np.uint32)[:n])
        assert np.array_equal(idx, np.sort(oidx[:n]))
        # the re-traced photons are exactly what the oracle traces for those indices in the new volume; the others are untouched
        after = _n(cm.photons)
        want_sel = _oracle_trace(oracle, vols[t], S.tf_from_points(tfp), cm, 1.0 / vdim, indices=idx.astype(np.int64))
        assert np.array_equal(bits(after[idx.astype(np.int64)]), bits(want_sel))
        keep = np.ones(cm.n, bool)
        keep[idx.astype(np.int64)] = False
        assert np.array_equal(bits(after[keep]), bits(before[keep]))
        # against a from-scratch trace of the new volume: near-total agreement (bricks have no apron, as in the
        # reference: uniformgridcl/cl/uniformgrid/volumeminmax.cl:43-45); stated bound 0.5 % of the photons
        fresh = _oracle_trace(oracle, vols[t], S.tf_from_points(tfp), cm, 1.0 / vdim)
        stale = (bits(after) != bits(fresh)).any(axis=1).mean()
        assert stale < 0.005, stale
        cm.full_frame()
