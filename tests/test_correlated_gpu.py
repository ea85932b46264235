"""Parity of the correlated re-trace helpers (C1-C6, S2-S4) against the oracle, on the GPU."""
import numpy as np
import pytest

from test_parity_gpu import _t, _n, bits, _light_setup

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.402823466e+38)


@pytest.mark.parametrize("streaming", [1, 0])
@pytest.mark.parametrize("shape,dtype,region", [((64, 64, 64), np.uint8, 8), ((19, 33, 50), np.uint8, 8),
                                                 ((32, 32, 32), np.uint16, 4), ((24, 24, 24), np.float32, 8),
                                                 ((21, 30, 48), np.uint8, 8),      # aligned rows, bricks clipped in y and z
                                                 ((17, 9, 32), np.uint8, 5),       # bricks that straddle 16-byte chunks
                                                 ((10, 12, 40), np.uint16, 16), ((12, 20, 36), np.float32, 3),
                                                 ((8, 8, 16), np.uint8, 1)])
def test_volume_minmax_and_difference(ctx, oracle, shape, dtype, region, streaming):
    ctx.lib.cpm_debug_set_brick_streaming(ctx.h, streaming)
    try:
        _minmax_difference_case(ctx, oracle, shape, dtype, region)
    finally:
        ctx.lib.cpm_debug_set_brick_streaming(ctx.h, 1)


def _minmax_difference_case(ctx, oracle, shape, dtype, region):
    rng = np.random.default_rng(sum(shape))
    if dtype == np.float32:
        a, b = rng.random(shape, dtype=np.float32), rng.random(shape, dtype=np.float32)
    else:
        hi = np.iinfo(dtype).max + 1
        a, b = rng.integers(0, hi, shape).astype(dtype), rng.integers(0, hi, shape).astype(dtype)
    # smooth some structure in so that bricks differ
    a[: shape[0] // 2] = a[: shape[0] // 2] // 4 if dtype != np.float32 else a[: shape[0] // 2] * 0.25
    va, vb = ctx.volume_create(a), ctx.volume_create(b)
    oa, ob = oracle.volume(a), oracle.volume(b)
    dims = shape[::-1]
    nb = int(np.prod([(d + region - 1) // region for d in dims]))
    mm = ctx.torch.zeros((nb, 2), dtype=ctx.torch.int16, device=ctx.device)
    ctx.volume_minmax(va, region, mm)
    assert np.array_equal(_n(mm, np.uint16), oracle.volume_minmax(oa, region))
    diff = ctx.torch.zeros(nb, dtype=ctx.torch.float32, device=ctx.device)
    ctx.volume_difference(va, vb, region, diff)
    assert np.array_equal(bits(_n(diff)), bits(oracle.volume_difference(oa, ob, region)))
    # the time step in one pass (cpm_volume_step): the difference and the NEXT volume's min / max, the same values
    diff2 = ctx.torch.full_like(diff, -1.0)
    mm2 = ctx.torch.zeros_like(mm)
    ctx.volume_step(va, vb, region, diff2, mm2)
    assert np.array_equal(bits(_n(diff2)), bits(oracle.volume_difference(oa, ob, region)))
    assert np.array_equal(_n(mm2, np.uint16), oracle.volume_minmax(ob, region))


def _tf_diff_points():
    # |TF_new - TF_old| break points as the processor would build them: zero-padded at 0 and 1
    pos = np.array([0.0, 0.20, 0.2218, 0.26, 0.2851, 0.40, 1.0], np.float32)
    col = np.array([[0, 0, 0, 0], [0, 0, 0, 0], [0.1, 0.05, 0.02, 0.19], [0.3, 0.1, 0.15, 0.2],
                    [0.05, 0.0, 0.01, 0.02], [0, 0, 0, 0], [0, 0, 0, 0]], np.float32)
    return pos, col


def test_importance_tf(ctx, oracle):
    rng = np.random.default_rng(3)
    n = 40_000
    lo = rng.integers(0, 65536, n)
    hi = np.minimum(lo + rng.integers(0, 20000, n), 65535)
    mm = np.stack([lo, hi], 1).astype(np.uint16)
    mm[:50, 1] = mm[:50, 0]  # degenerate ranges
    mm[50:60] = (0, 65535)
    pos, col = _tf_diff_points()
    out = ctx.torch.zeros(n, dtype=ctx.torch.float32, device=ctx.device)
    ctx.importance_tf(_t(ctx, mm), n, pos, col, out)
    want = oracle.importance_tf(mm, pos, col)
    assert np.array_equal(bits(_n(out)), bits(want))
    assert want.max() > 0 and (want == 0).any()
    # time-varying variant
    prev = mm.copy()
    prev[:, 0] = np.maximum(prev[:, 0].astype(np.int64) - 3000, 0)
    diff = rng.random(n, dtype=np.float32)
    ctx.importance_tf(_t(ctx, mm), n, pos, col, out, prev_minmax=_t(ctx, prev), volume_diff=_t(ctx, diff))
    want = oracle.importance_tf(mm, pos, col, prev=prev, diff=diff)
    assert np.array_equal(bits(_n(out)), bits(want))


@pytest.mark.parametrize("max_inter,fix", [(1, False), (1, True), (3, False)])
def test_photon_importance(ctx, oracle, cpm, max_inter, fix):
    from oracle_binding import OTraceParams, default_matrices
    S = cpm.synthetic
    n_side, vdim, region = 96, 64, 8
    n = n_side * n_side
    vol_np, tf = S.heterogeneous_volume(vdim), S.tf_from_points([(0.0, 1, 1, 1, 0.0), (0.6, 1, 1, 1, 0.0), (1.0, 1, 1, 1, 0.4)])
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
    assert (ph[:n, 0] == FLT_MAX).any() and (ph[:n, 0] != FLT_MAX).any()
    gd = (vdim // region,) * 3
    rng = np.random.default_rng(9)
    grid = rng.random(gd[0] * gd[1] * gd[2], dtype=np.float32)
    grid[rng.random(grid.size) < 0.5] = 0
    t2i, _ = default_matrices((vdim,) * 3)
    imp0 = np.full(n + 10, 2147483647, np.uint32)
    imp_o = imp0.copy()
    oracle.photon_importance(grid, gd, (region,) * 3, t2i, ph, 5, ls, isect, n - 5, max_inter, n, imp_o, fix_exit_point=fix)
    imp_d = _t(ctx, imp0)
    ctx.photon_importance(_t(ctx, grid), gd, (float(region),) * 3, t2i.tolist(), _t(ctx, ph), 5, _t(ctx, ls), _t(ctx, isect),
                          n - 5, max_inter, n, imp_d, fix_exit_point=fix)
    assert np.array_equal(_n(imp_d, np.uint32), imp_o)
    assert (imp_o[5:n] < 2147483647).any()
    assert (imp_o[:5] == 2147483647).all() and (imp_o[n:] == 2147483647).all()


def test_equal_importance_reset_and_select(ctx, oracle):
    n = 100_000
    imp0 = np.full(n, 2147483647, np.uint32)
    a = imp0.copy()
    oracle.photon_importance_equal(1000, n - 1000, 25, 3, a)
    d = _t(ctx, imp0)
    ctx.photon_importance_equal(1000, n - 1000, 25, 3, d)
    assert np.array_equal(_n(d, np.uint32), a)
    assert (a < 2147483647).sum() == ((np.arange(1000, n) + 3) % 4 == 0).sum()
    # select: threshold + count + iota + stable sort by importance
    rng = np.random.default_rng(4)
    imp = np.full(n, 2147483647, np.uint32)
    changed = rng.random(n) < 0.3
    imp[changed] -= rng.integers(1, 5000, changed.sum()).astype(np.uint32)
    want_imp = imp.copy()
    want_idx, want_cnt = oracle.select_recompute(want_imp)
    dimp = _t(ctx, imp)
    didx = ctx.torch.zeros(n, dtype=ctx.torch.int32, device=ctx.device)
    dcnt = ctx.torch.full((1,), -7, dtype=ctx.torch.int32, device=ctx.device)
    ctx.select_recompute(dimp, didx, dcnt)
    assert int(dcnt.item()) == want_cnt == int(changed.sum())
    assert np.array_equal(_n(didx, np.uint32), want_idx)
    assert np.array_equal(_n(dimp, np.uint32), want_imp)
    # reset
    ctx.reset_importance(dimp, 10, 500)
    got = _n(dimp, np.uint32)
    assert (got[10:510] == 2147483647).all() and np.array_equal(got[:10], want_imp[:10]) and np.array_equal(got[510:], want_imp[510:])
    ctx.select_recompute(ctx.torch.zeros(0, dtype=ctx.torch.int32, device=ctx.device), didx[:0], dcnt)
    assert int(dcnt.item()) == 0


def test_correlated_update_end_to_end(ctx, oracle, cpm):
    """BASELINE config 3 at reduced size: after a TF edit the importance-driven re-trace (same
    RNG streams) reproduces a full re-trace photon for photon, the +-delta light-volume update
    matches a full gather within fp32 tolerance, and the selection equals the oracle's."""
    from oracle_binding import default_matrices
    S, P = cpm.synthetic, cpm.pipeline
    vol_np = S.heterogeneous_volume(64)
    # a TF whose low end is transparent, so that photons travel and bricks differ in importance
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0), tf_points=base, incremental_threshold_percent=100.0)
    cm = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), 160, (32, 32, 32), **kw)
    cm.full_frame()
    before = _n(cm.photons).copy()
    lv_before = _n(cm.light_volume).copy()
    pos, col = cm.set_transfer_function(edit)
    n = cm.correlated_update()
    assert 0 < n < cm.n
    after = _n(cm.photons)
    # reference run: everything from scratch with the edited TF
    full = P.PhotonFrame(ctx, vol_np, S.tf_from_points(edit), 160, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0))
    lv_full = _n(full.frame())
    assert np.array_equal(bits(after), bits(_n(full.photons)))          # correlated == full re-trace
    changed = (bits(after) != bits(before)).any(axis=1)
    assert 0 < changed.sum() <= n                                        # only selected photons moved
    idx = np.sort(_n(cm.indices, np.uint32)[:n])
    assert np.isin(np.nonzero(changed)[0], idx).all()
    # light volume: incremental path taken, equals the full gather within fp32 tolerance
    assert cm.last_path == "incremental"
    lv = _n(cm.light_volume)
    np.testing.assert_allclose(lv, lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
    assert np.abs(lv - lv_before).max() > 0
    # the selection against the oracle
    t2i, _ = default_matrices((64, 64, 64))
    ovol = oracle.volume(vol_np)
    mm = oracle.volume_minmax(ovol, 8)
    grid = oracle.importance_tf(mm, pos, col)
    assert np.array_equal(bits(grid), bits(_n(cm.importance_grid)))
    imp = np.full(cm.n, 2147483647, np.uint32)
    oracle.photon_importance(grid, (8, 8, 8), (8.0,) * 3, t2i, before, 0, _n(cm.light_samples), _n(cm.isect), cm.n, 1, cm.n, imp)
    oidx, ocnt = oracle.select_recompute(imp)
    assert ocnt == n
    assert np.array_equal(np.sort(oidx[:n]), idx)
    # all importances are back to "unchanged" after a 100 % update
    assert (_n(cm.importance, np.uint32) == 2147483647).all()
    # progressive batches (25 % per evaluation) converge to the same photons
    cm2 = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), 160, (32, 32, 32), max_incremental_percent=5.0, **kw)
    cm2.full_frame()
    cm2.set_transfer_function(edit)
    done = cm2.correlated_update()
    rounds = 1
    while cm2.remaining > 0:
        done += cm2.continue_update()
        rounds += 1
    assert done == n and rounds > 1
    assert np.array_equal(bits(_n(cm2.photons)), bits(after))


def test_time_varying_sequence(ctx, oracle, cpm):
    """BASELINE config 5 at reduced size: a moving blob; per time step the GPU difference / min-max /
    time-varying importance drive a correlated re-trace that lands exactly on the photons of a
    from-scratch trace of the new volume, while re-tracing only part of the photons."""
    S, P = cpm.synthetic, cpm.pipeline
    tfp = [(0.0, 1, 1, 1, 0.0), (0.55, 1, 0.5, 0.2, 0.0), (0.7, 0.6, 0.3, 0.1, 0.3), (1.0, 0.1, 0.6, 0.7, 0.6)]
    n_steps = 4
    vols = [S.heterogeneous_volume(64, S.sequence_blob_center(t * 10, 32)) for t in range(n_steps)]
    cm = P.CorrelatedPhotonMapper(ctx, vols[0], S.tf_from_points(tfp), 128, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0),
                                  tf_points=tfp, incremental_threshold_percent=100.0)
    cm.full_frame()
    fractions = []
    for t in range(1, n_steps):
        cm.set_volume(vols[t])
        # importance grid against the oracle (difference, min/max and classification all on the GPU)
        oa, ob = oracle.volume(vols[t - 1]), oracle.volume(vols[t])
        diff = oracle.volume_difference(oa, ob, 8)
        mm_a, mm_b = oracle.volume_minmax(oa, 8), oracle.volume_minmax(ob, 8)
        pts = sorted(tfp)
        pos = np.array([p[0] for p in pts], np.float32)
        col = np.array([p[1:] for p in pts], np.float32)
        want = oracle.importance_tf(mm_b, pos, col, prev=mm_a, diff=diff)
        assert np.array_equal(bits(_n(cm.importance_grid)), bits(want))
        n = cm.correlated_update()
        fractions.append(n / cm.n)
        fresh = P.PhotonFrame(ctx, vols[t], S.tf_from_points(tfp), 128, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0))
        lv_full = _n(fresh.frame())
        # Unlike a TF edit, a data change can reach a photon through the trilinear footprint of a
        # neighbouring brick its path never enters (the reference's bricks have no apron either:
        # uniformgridcl/cl/uniformgrid/volumeminmax.cl:43-45), so agreement with a fresh trace is
        # near-total rather than exact.
        stale = (bits(_n(cm.photons)) != bits(_n(fresh.photons))).any(axis=1).mean()
        assert stale < 0.02, stale
        lv = _n(cm.light_volume)
        assert np.abs(lv - lv_full).sum() < 0.05 * np.abs(lv_full).sum()
        cm.full_frame()  # resynchronise before the next step so that the errors do not accumulate in the test
    assert all(0 < f < 1 for f in fractions), fractions


def test_time_steps_as_resident_volumes(ctx, cpm):
    """set_volume adopts a volume created beforehand (a sequence kept on the device) without copying it: same importance
    grid, same selection, same photons and light volume as stepping with raw voxels; and raw-voxel steps after adopted
    ones never write into a volume the caller owns."""
    S, P = cpm.synthetic, cpm.pipeline
    tfp = [(0.0, 1, 1, 1, 0.0), (0.55, 1, 0.5, 0.2, 0.0), (0.7, 0.6, 0.3, 0.1, 0.3), (1.0, 0.1, 0.6, 0.7, 0.6)]
    vols = [S.heterogeneous_volume(64, S.sequence_blob_center(t * 10, 32)) for t in range(4)]
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0), tf_points=tfp, incremental_threshold_percent=100.0)
    raw = P.CorrelatedPhotonMapper(ctx, vols[0], S.tf_from_points(tfp), 128, (32, 32, 32), **kw)
    resident = [ctx.volume_create(v) for v in vols]
    ado = P.CorrelatedPhotonMapper(ctx, resident[0], S.tf_from_points(tfp), 128, (32, 32, 32), **kw)
    raw.full_frame(); ado.full_frame()
    for t in (1, 2, 3, 1):
        raw.set_volume(vols[t])
        ado.set_volume(resident[t] if t != 3 else vols[3])   # a raw-voxel step in between: copied into a volume of the mapper's own
        assert np.array_equal(bits(_n(raw.importance_grid)), bits(_n(ado.importance_grid)))
        assert raw.correlated_update() == ado.correlated_update()
        assert np.array_equal(bits(_n(raw.photons)), bits(_n(ado.photons)))
        a, b = _n(raw.light_volume), _n(ado.light_volume)   # +- atomic splat of the update: order-dependent sums
        assert np.allclose(a, b, rtol=1e-4, atol=1e-5 * float(np.abs(a).max()))
    for v, h in zip(vols, resident):
        assert np.array_equal(h.download(), v)
    with pytest.raises(ValueError):
        ado.set_volume(ctx.volume_create(S.heterogeneous_volume(32)))


def test_sharded_correlated_update(ctx, cpm):
    """Multi-GPU semantics of the correlated path (per-shard selection), two shards emulated on one
    GPU: every rank scores, selects and re-traces its own photon range; the union of the selections and the
    re-traced photons equal the unsharded run's (photon i does not depend on its shard), and the sum of the
    per-rank light volumes -- what the one all-reduce per frame produces -- equals the unsharded light volume
    within summation-order tolerance, before and after the TF edit."""
    import importlib
    S, P = cpm.synthetic, cpm.pipeline
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    vol_np = S.heterogeneous_volume(64)
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0), tf_points=base, incremental_threshold_percent=100.0)
    n_side, world = 128, 2
    whole = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), n_side, (32, 32, 32), **kw)
    shards = [P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), n_side, (32, 32, 32),
                                       photon_range=sh.shard_range(n_side * n_side, r, world), **kw) for r in range(world)]
    whole.full_frame()
    for s in shards:
        s.full_frame()
    lv = _n(whole.light_volume)
    tol = dict(rtol=1e-4, atol=2e-6 * float(lv.max()))
    np.testing.assert_allclose(sum(_n(s.light_volume) for s in shards), lv, **tol)
    assert np.array_equal(bits(np.concatenate([_n(s.photons) for s in shards])), bits(_n(whole.photons)))

    whole.set_transfer_function(edit)
    n_whole = whole.correlated_update()
    picked = []
    for r, s in enumerate(shards):
        s.set_transfer_function(edit)
        n = s.correlated_update()
        lo, _ = sh.shard_range(n_side * n_side, r, world)
        picked.append(_n(s.indices, np.uint32)[:n].astype(np.int64) + lo)   # shard-local -> global photon index
    picked = np.sort(np.concatenate(picked))
    assert 0 < n_whole < whole.n and picked.size == n_whole
    assert np.array_equal(picked, np.sort(_n(whole.indices, np.uint32)[:n_whole].astype(np.int64)))
    assert np.array_equal(bits(np.concatenate([_n(s.photons) for s in shards])), bits(_n(whole.photons)))
    lv2 = _n(whole.light_volume)
    np.testing.assert_allclose(sum(_n(s.light_volume) for s in shards), lv2, rtol=1e-3, atol=2e-5 * float(lv2.max()))
    assert np.abs(lv2 - lv).max() > 0


@pytest.mark.parametrize("partition", [1, 0])
@pytest.mark.parametrize("n,frac", [(1, 1.0), (1, 0.0), (2, 0.5), (255, 0.3), (256, 0.5), (257, 0.9), (1000, 0.0), (1000, 1.0), (2048, 0.5),
                                    (2049, 0.01), (70_001, 0.013), (300_000, 0.4), (2_500_003, 0.07)])
def test_select_changed(ctx, oracle, n, frac, partition):
    """cpm_select_changed: changed photons first, both parts in ascending index order, the importances untouched --
    through the two-launch partition (default; tiles grow past 2048 photons beyond 1024 tiles) and the radix-pass form."""
    ctx.lib.cpm_debug_set_select_partition(ctx.h, partition)
    try:
        _select_changed_case(ctx, oracle, n, frac)
    finally:
        ctx.lib.cpm_debug_set_select_partition(ctx.h, 1)


def _select_changed_case(ctx, oracle, n, frac):
    rng = np.random.default_rng(n)
    imp = np.full(n, 2147483647, np.uint32)
    pick = rng.random(n) < frac
    imp[pick] = rng.integers(0, 2147483647, int(pick.sum()), dtype=np.uint32)
    imp_d = _t(ctx, imp.view(np.int32))
    idx = ctx.torch.full((n,), -1, dtype=ctx.torch.int32, device=ctx.device)
    cnt = ctx.torch.full((1,), -7, dtype=ctx.torch.int32, device=ctx.device)
    for _ in range(2):
        ctx.select_changed(imp_d, idx, cnt)
        want_idx, want_cnt = oracle.select_changed(imp)
        assert int(cnt.item()) == want_cnt == int(pick.sum())
        assert np.array_equal(_n(idx, np.uint32), want_idx)
        assert np.array_equal(_n(imp_d, np.uint32), imp)


@pytest.mark.parametrize("interactions", [1, 3])
def test_exact_incremental_update_is_bit_identical_to_a_full_frame(ctx, cpm, interactions):
    """exact_update: after the correlated re-trace, the bricks touched by an old or new position of a changed photon
    are re-gathered from the re-binned photons and nothing else is written -- the light volume equals, bit for bit,
    the one a full trace + bin + gather with the edited TF produces (the +-splat update only matches it within
    tolerance).  Also through progressive batches."""
    S, P = cpm.synthetic, cpm.pipeline
    vol_np = S.heterogeneous_volume(64)
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0), tf_points=base, incremental_threshold_percent=100.0, exact_update=True,
              max_interactions=interactions, material=(0.3, 0, 0, 0))
    full = P.PhotonFrame(ctx, vol_np, S.tf_from_points(edit), 160, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0),
                         max_interactions=interactions, material=(0.3, 0, 0, 0))
    lv_full = _n(full.frame()).copy()
    for pct in (100.0, 7.0):
        cm = P.CorrelatedPhotonMapper(ctx, vol_np, S.tf_from_points(base), 160, (32, 32, 32), max_incremental_percent=pct, **kw)
        cm.full_frame()
        lv_before = _n(cm.light_volume).copy()
        cm.set_transfer_function(edit)
        n = cm.correlated_update()
        assert n > 0 and cm.last_path == "exact incremental"
        while cm.remaining > 0:
            assert cm.continue_update() > 0 and cm.last_path == "exact incremental"
        assert np.array_equal(bits(_n(cm.photons)), bits(_n(full.photons)))
        lv = _n(cm.light_volume)
        assert np.array_equal(bits(lv), bits(lv_full))
        changed = bits(lv) != bits(lv_before)
        assert 0 < changed.sum() < lv.size          # a local update, not a rewrite of everything


@pytest.mark.parametrize("radius_vox,force_voxel", [(0.8, 0), (0.8, 1), (0.8, 2), (1.3, 0), (1.3, 2), (2.7, 0)])
def test_mark_touched_bricks_and_gather_bricks(ctx, oracle, cpm, radius_vox, force_voxel):
    """cpm_mark_touched_bricks marks exactly the bricks the selected photons' splat boxes overlap; cpm_gather_bricks
    rewrites exactly the marked bricks -- whichever gather kernel the radius (or the test hook) selects."""
    ctx.lib.cpm_debug_force_voxel_gather(ctx.h, int(force_voxel))
    try:
        _mark_and_gather_bricks(ctx, oracle, cpm, radius_vox)
    finally:
        ctx.lib.cpm_debug_force_voxel_gather(ctx.h, 0)


def _mark_and_gather_bricks(ctx, oracle, cpm, radius_vox):
    rng = np.random.default_rng(9)
    dims, n = (24, 20, 28), 5000
    ph = np.zeros((n, 8), np.float32)
    ph[:, :3] = rng.random((n, 3), dtype=np.float32)
    ph[:, 3:6] = rng.random((n, 3), dtype=np.float32)
    ph[::50, :3] = np.float32(3.402823466e+38)
    radius = float(np.float32(radius_vox / max(dims)))
    g, og = cpm.binding.default_grid_desc(dims, 1), oracle.grid(dims, 1)
    bdim = [(d + 3) // 4 for d in dims]
    nb = bdim[0] * bdim[1] * bdim[2]
    sel = np.sort(rng.choice(n, 60, replace=False)).astype(np.uint32)
    torch = ctx.torch
    mask = torch.zeros(nb, dtype=torch.uint8, device=ctx.device)
    ctx.mark_touched_bricks(_t(ctx, ph), _t(ctx, sel.view(np.int32)), sel.size, n, 1, g, radius, mask)
    # expected: bricks overlapped by the oracle's splat of each selected photon alone (any voxel inside its box)
    want = np.zeros(nb, np.uint8)
    t2i = np.array(og.texture_to_index, np.float32)
    for i in sel:
        p = ph[i, :3]
        if (p == np.float32(3.402823466e+38)).any():
            continue
        lo = [int(np.float32(np.float32(p[a] - np.float32(radius)) * t2i[5 * a] + t2i[12 + a])) for a in range(3)]
        hi = [int(np.float32(np.float32(np.float32(p[a] + np.float32(radius)) * t2i[5 * a] + t2i[12 + a]) + np.float32(1))) for a in range(3)]
        lo = [max(v, 0) for v in lo]
        hi = [min(v, dims[a]) for a, v in enumerate(hi)]
        for bz in range(lo[2] >> 2, ((hi[2] - 1) >> 2) + 1):
            for by in range(lo[1] >> 2, ((hi[1] - 1) >> 2) + 1):
                for bx in range(lo[0] >> 2, ((hi[0] - 1) >> 2) + 1):
                    want[bx + bdim[0] * (by + bdim[1] * bz)] = 1
    got = mask.cpu().numpy()
    assert np.array_equal(got, want) and 0 < want.sum() < nb
    cells = dims[0] * dims[1] * dims[2]
    order = torch.empty(n, dtype=torch.int32, device=ctx.device)
    cs = torch.empty(cells + 1, dtype=torch.int32, device=ctx.device)
    srt = torch.empty((n, 4), dtype=torch.float32, device=ctx.device)
    ctx.bin(_t(ctx, ph), n, g, order, cs, srt)
    out = torch.full((cells,), -5.0, dtype=torch.float32, device=ctx.device)
    ctx.gather_bricks(srt, cs, n, g, radius, 1.0, mask, out)
    fullv = torch.empty(cells, dtype=torch.float32, device=ctx.device)
    ctx.gather(srt, cs, n, g, radius, 1.0, fullv)
    o, f = _n(out).reshape(dims[::-1]), _n(fullv).reshape(dims[::-1])
    for bz in range(bdim[2]):
        for by in range(bdim[1]):
            for bx in range(bdim[0]):
                blk = (slice(4 * bz, 4 * bz + 4), slice(4 * by, 4 * by + 4), slice(4 * bx, 4 * bx + 4))
                if want[bx + bdim[0] * (by + bdim[1] * bz)]:
                    assert np.array_equal(bits(o[blk]), bits(f[blk]))
                else:
                    assert (o[blk] == np.float32(-5.0)).all()
