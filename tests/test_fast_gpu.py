"""The tolerance-mode formulation (cpm_bin_fast + cpm_gather_fast, include/cpm/cpm.h) on a real MI355X.

Two bars, both through the C-ABI:
  * bit for bit against the oracle's restatement of the SAME formulation (cpmo_gather_fast: fixed-point
    contributions, exact integer sums, one rounding) -- integer sums do not depend on order, so the
    unordered brick bin and the LDS atomics leave nothing to tolerate;
  * within the stated fp32 tolerance (rtol 2e-5, atol 1e-5 * max -- what the reference formulation's
    atomic splat is held to, tests/test_parity_gpu.py) against the oracle's reference-semantics gather
    (sequential fp32 sum, weight through sqrt and division: ref cl/photonstolightvolume.cl:42-75).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(3.402823466e+38)
RTOL, ATOL_OF_MAX = 2e-5, 1e-5


def _t(ctx, a):
    torch = ctx.torch
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).to(ctx.device)
    return torch.from_numpy(a).to(ctx.device)


def _n(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a.view(dtype) if dtype is not None else a


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def make_photons(rng, n, dims, *, sentinels=0.1, outside=0.02, cluster=None, negative=False):
    ph = np.zeros((n, 8), np.float32)
    ph[:, :3] = rng.random((n, 3), dtype=np.float32)
    if cluster is not None:  # pile a share of the photons onto one face / one cell, as light entering a volume does
        m = rng.random(n) < 0.5
        ph[m, 2] = np.float32(cluster) + rng.random(m.sum(), dtype=np.float32) * np.float32(0.5 / dims[2])
    out = rng.random(n) < outside
    ph[out, :3] = (rng.random((out.sum(), 3), dtype=np.float32) - np.float32(0.5)) * np.float32(3.0)
    ph[:, 3:6] = rng.random((n, 3), dtype=np.float32) * np.float32(10.0) + np.float32(0.01)
    if negative:
        ph[rng.random(n) < 0.3, 3:6] *= np.float32(-1.0)
    ph[:, 6:] = rng.random((n, 2), dtype=np.float32)
    s = rng.random(n) < sentinels
    ph[s, :3] = FLT_MAX
    ph[s, 4:6] = FLT_MAX
    return ph


def candidates_per_axis(radius, grid):
    """floor(2 r') + 1 with r' = r * textureToIndex + 1e-3, per axis."""
    t2i = np.array(grid.texture_to_index, np.float32)
    return [int(np.floor(np.float32(2.0) * np.float32(np.float32(radius) * t2i[5 * a] + np.float32(1e-3)))) + 1 for a in range(3)]


def brick_layout(dims, mc=None):
    """Bricks of 8 x 8 x 16 voxels -- for a candidate box wider than 4 along some axis (mc: candidates per axis) 16 voxels along the
    axes with 5 or more candidates and 8 along the others, at most 2048 voxels; beyond 8 Ki of them doubled along the axis with the
    most bricks (cpm.h): (log2 sizes, counts)."""
    lg = [3, 3, 4]
    if mc is not None and max(mc) > 4:
        lg = [4 if c >= 5 else 3 for c in mc]
        if sum(lg) > 11:
            lg[2] = 3
        # ... and a brick's voxels x the box's candidates within reach ((mc - 1) per axis) stay under 56 Ki: halved along the longest axis,
        # not below 8 voxels
        box = int(np.prod([7 if c > 8 else max(c - 1, 1) for c in mc]))
        while (1 << sum(lg)) * box > 56 * 1024:
            a = int(np.argmax(lg))
            if lg[a] <= 3:
                break
            lg[a] -= 1
    cnt = lambda: [(d + (1 << l) - 1) >> l for d, l in zip(dims, lg)]
    while np.prod(cnt()) > 8192:
        c = cnt()
        lg[int(np.argmax(c))] += 1          # (argmax takes the first of equals: x, then y, then z)
    return lg, cnt()


def brick_count(dims, mc=None):
    return int(np.prod(brick_layout(dims, mc)[1]))


def expected_filing(ph, dims, radius, grid):
    """(photon index, brick) pairs cpm_bin_fast must file: every stored photon under every brick its candidate voxels
    (integers within r * textureToIndex + 1e-3 of the index-space coordinate, clipped to the grid) lie in -- narrow boxes; a photon with a
    box of 4 or more candidates along some axis once, under the brick of the box's low corner (that brick's tile has a halo)."""
    mc = candidates_per_axis(radius, grid)
    lg, nbk = brick_layout(dims, mc)
    once = max(mc) > 3
    t2i = np.array(grid.texture_to_index, np.float32)
    pairs = []
    stored = np.where(ph[:, 0] != FLT_MAX)[0]
    rng_axes = []
    # at most floor(2 r') + 1 candidates along an axis, r' = r * textureToIndex + 1e-3 of that axis
    for a in range(3):
        s, t = t2i[5 * a], t2i[12 + a]
        maxc = int(np.floor(np.float32(2.0) * np.float32(np.float32(radius) * s + np.float32(1e-3)))) + 1
        u = (np.float64(s) * ph[stored, a].astype(np.float64) + np.float64(t)).astype(np.float32)   # = fma(s, p, t): the product is exact in double
        rg = np.float32(radius) * s + np.float32(1e-3)
        lo = np.clip(np.ceil((u - rg).astype(np.float32)), 0, dims[a]).astype(np.int64)
        hi = np.clip(np.floor((u + rg).astype(np.float32)), -1, dims[a] - 1).astype(np.int64)
        hi = np.minimum(hi, lo + maxc - 1)
        rng_axes.append((lo, hi))
    ok = np.ones(stored.size, bool)
    for lo, hi in rng_axes:
        ok &= lo <= hi
    for j in np.where(ok)[0]:
        (lx, hx), (ly, hy), (lz, hz) = [(int(lo[j]) >> lg[a], int(lo[j] if once else hi[j]) >> lg[a]) for a, (lo, hi) in enumerate(rng_axes)]
        for bz in range(lz, hz + 1):
            for by in range(ly, hy + 1):
                for bx in range(lx, hx + 1):
                    pairs.append((int(stored[j]), bx + nbk[0] * (by + nbk[1] * bz)))
    return pairs


def run_fast(ctx, cpm, ph, dims, channels, radius, scale, accumulate_into=None):
    B = cpm.binding
    n = ph.shape[0]
    grid = B.default_grid_desc(dims, channels)
    cells = dims[0] * dims[1] * dims[2]
    table = ctx.torch.zeros(max(ctx.fast_table_entries(grid, n), 1), dtype=ctx.torch.int32, device=ctx.device)
    srt = ctx.torch.zeros((max(ctx.fast_record_capacity(grid, n, radius), 1), 4 if channels == 1 else 8), dtype=ctx.torch.float32, device=ctx.device)
    d_ph = _t(ctx, ph) if n else ctx.torch.zeros((1, 8), dtype=ctx.torch.float32, device=ctx.device)
    ctx.bin_fast(d_ph, n, grid, radius, table, srt)
    if accumulate_into is None:
        out = ctx.torch.full((cells,) if channels == 1 else (cells, 4), 7.0, dtype=ctx.torch.float32, device=ctx.device)
        ctx.gather_fast(srt, table, n, grid, radius, scale, out)
    else:
        out = _t(ctx, accumulate_into)
        ctx.gather_fast(srt, table, n, grid, radius, scale, out, accumulate=True)
    return _n(out), _n(table, np.uint32), _n(srt)


def oracle_both(oracle, ph, dims, channels, radius, scale, accumulate_into=None):
    n = ph.shape[0]
    og = oracle.grid(dims, channels)
    cells = dims[0] * dims[1] * dims[2]
    shape = (cells,) if channels == 1 else (cells, 4)
    fast = np.zeros(shape, np.float32) if accumulate_into is None else accumulate_into.copy()
    oracle.gather_fast(ph, n, og, radius, scale, fast, accumulate=accumulate_into is not None)
    _, cs, srt = oracle.bin(ph, n, og)
    exact = np.zeros(shape, np.float32) if accumulate_into is None else accumulate_into.copy()
    oracle.gather(srt, cs, n, og, radius, scale, exact, accumulate=accumulate_into is not None)
    return fast, exact


CASES = [
    # dims, channels, n, radius in voxels of the longest axis, kwargs
    ((32, 32, 32), 1, 20_000, 0.866, {}),
    ((32, 32, 32), 4, 20_000, 0.866, {}),
    ((64, 64, 64), 1, 200_000, 0.866, dict(cluster=0.0)),
    ((64, 64, 64), 1, 50_000, 1.3, dict(cluster=0.97)),
    ((64, 64, 64), 1, 30_000, 1.9, {}),
    ((40, 24, 56), 1, 30_000, 0.7, {}),          # ragged: partial bricks on every axis, anisotropic radius in voxels
    ((40, 24, 56), 4, 30_000, 1.2, dict(negative=True)),
    ((1, 5, 17), 1, 3_000, 0.4, {}),             # degenerate
    ((7, 1, 3), 4, 1_000, 0.9, {}),
    ((128, 128, 128), 1, 300_000, 0.866, dict(cluster=0.0)),
    ((256, 128, 64), 1, 100_000, 1.0, {}),
    ((256, 256, 192), 1, 60_000, 1.0, {}),       # > 8 Ki bricks of 8x8x16: bigger bricks
    ((16, 16, 16), 1, 1, 0.866, dict(sentinels=0.0, outside=0.0)),
    ((16, 16, 16), 1, 63, 0.866, {}),
    ((16, 16, 16), 1, 4097, 0.866, {}),
    # boxes wider than 4 candidates: one filing per photon, tiles with a halo, staged and merged (fast_halo_kernel + fast_halo_merge_kernel)
    ((64, 64, 64), 1, 40_000, 2.3, {}),
    ((64, 64, 64), 4, 20_000, 3.4, dict(negative=True)),
    ((40, 24, 56), 1, 30_000, 3.1, dict(cluster=0.0)),
    ((256, 256, 48), 1, 200_000, 2.76, dict(cluster=0.9)),   # the workspace's light volume: 6 x 6 x 2 candidates
    ((96, 96, 18), 4, 30_000, 2.76, {}),
    ((256, 256, 192), 1, 60_000, 3.3, {}),                   # bigger bricks and wide boxes
    ((24, 64, 64), 1, 20_000, 3.0, {}),                      # narrow along x (3 candidates), wide along y and z (7)
    ((24, 64, 64), 4, 10_000, 3.0, dict(negative=True)),
]


@pytest.mark.parametrize("dims,channels,n,rvox,kw", CASES)
def test_fast_equals_restatement_and_reference_semantics(ctx, oracle, cpm, dims, channels, n, rvox, kw):
    rng = np.random.default_rng(n + dims[0] * 7 + channels)
    ph = make_photons(rng, n, dims, **kw)
    radius = float(np.float32(rvox) / np.float32(max(dims)))    # rvox voxels along the longest axis, fewer along the others
    scale = float(cpm.binding.relative_irradiance_scale(radius, float(n)))
    got, table, srt = run_fast(ctx, cpm, ph, dims, channels, radius, scale)
    want_fast, want_exact = oracle_both(oracle, ph, dims, channels, radius, scale)
    assert np.array_equal(bits(got), bits(want_fast))
    np.testing.assert_allclose(got, want_exact, rtol=RTOL, atol=ATOL_OF_MAX * float(np.abs(want_exact).max()))
    # the table: a photon is filed under every brick its candidate voxels lie in; brick starts are those counts, the records
    # of a brick are exactly the photons filed under it (in no particular order)
    grid = cpm.binding.default_grid_desc(dims, channels)
    pairs = expected_filing(ph, dims, radius, grid)
    nb = brick_count(dims, candidates_per_axis(radius, grid))
    total = len(pairs)
    assert table[0] == 0 and table[nb] == total and (np.diff(table[: nb + 1].astype(np.int64)) >= 0).all()
    assert total <= ctx.fast_record_capacity(grid, n, radius)
    counts = np.bincount(np.array([b for _, b in pairs], np.int64), minlength=nb) if pairs else np.zeros(nb, np.int64)
    assert np.array_equal(np.diff(table[: nb + 1].astype(np.int64)), counts)
    rec_of = (lambda i: ph[i, :4]) if channels == 1 else (lambda i: np.concatenate([ph[i, :6], np.zeros(2, np.float32)]))
    width = 4 if channels == 1 else 8
    by_brick = {}
    for i, b in pairs:
        by_brick.setdefault(b, []).append(rec_of(i))
    for b, recs in list(by_brick.items())[:400]:  # (every brick for the small cases, a sample for the big ones)
        got_b = np.ascontiguousarray(srt[table[b]:table[b + 1]]).view([("", np.uint32)] * width).reshape(-1)
        want_b = np.ascontiguousarray(np.stack(recs).astype(np.float32)).view([("", np.uint32)] * width).reshape(-1)
        assert np.array_equal(np.sort(got_b), np.sort(want_b)), b
    # bitwise reproducible although nothing orders the records inside a brick
    again, _, _ = run_fast(ctx, cpm, ph, dims, channels, radius, scale)
    assert np.array_equal(bits(again), bits(got))


def test_fast_accumulate_empty_and_all_sentinels(ctx, oracle, cpm):
    dims, radius = (32, 32, 32), 0.866 / 32
    rng = np.random.default_rng(3)
    ph = make_photons(rng, 10_000, dims)
    scale = 0.01
    base = rng.random(32 ** 3, dtype=np.float32)
    got, _, _ = run_fast(ctx, cpm, ph, dims, 1, radius, scale, accumulate_into=base)
    want_fast, want_exact = oracle_both(oracle, ph, dims, 1, radius, scale, accumulate_into=base)
    assert np.array_equal(bits(got), bits(want_fast))
    np.testing.assert_allclose(got, want_exact, rtol=RTOL, atol=ATOL_OF_MAX * float(want_exact.max()))
    # no photons, and only sentinels: a cleared light volume (accumulate: untouched)
    empty = np.zeros((0, 8), np.float32)
    got, _, _ = run_fast(ctx, cpm, empty, dims, 1, radius, scale)
    assert not got.any()
    sent = make_photons(rng, 500, dims, sentinels=1.1)
    got, table, _ = run_fast(ctx, cpm, sent, dims, 4, radius, scale)
    assert not got.any() and table[brick_count(dims)] == 0
    got, _, _ = run_fast(ctx, cpm, sent, dims, 1, radius, scale, accumulate_into=base)
    assert np.array_equal(bits(got), bits(base))


@pytest.mark.parametrize("dims,channels,rvox", [((64, 64, 64), 1, 2.6), ((40, 24, 56), 4, 3.1), ((96, 96, 18), 1, 2.76), ((16, 16, 8), 1, 2.76)])
def test_fast_accumulate_wide_boxes(ctx, oracle, cpm, dims, channels, rvox):
    """Wide boxes (tiles with a halo, staged and merged): accumulate mode adds to what is there -- also in the bricks no tile covers --
    ragged grids, 4 channels, a halo along z, a grid of one brick."""
    rng = np.random.default_rng(dims[0] * 7 + channels)
    n = 15_000
    ph = make_photons(rng, n, dims, negative=channels == 4)
    radius = float(np.float32(rvox) / np.float32(max(dims)))
    scale = float(cpm.binding.relative_irradiance_scale(radius, float(n)))
    cells = dims[0] * dims[1] * dims[2]
    base = rng.random(cells if channels == 1 else (cells, 4), dtype=np.float32)
    got, _, _ = run_fast(ctx, cpm, ph, dims, channels, radius, scale, accumulate_into=base)
    want_fast, _ = oracle_both(oracle, ph, dims, channels, radius, scale, accumulate_into=base)
    assert np.array_equal(bits(got), bits(want_fast))
    # twice in a row from the same staging slots: the same bits
    again, _, _ = run_fast(ctx, cpm, ph, dims, channels, radius, scale, accumulate_into=base)
    assert np.array_equal(bits(again), bits(got))


def test_fast_after_bigger_call_and_other_grid(ctx, oracle, cpm):
    """Scratch reuse: the histogram must be back to zero whatever ran before (bigger n, other grid, other channels)."""
    rng = np.random.default_rng(11)
    for dims, ch, n in [((64, 64, 64), 1, 100_000), ((32, 32, 32), 4, 5_000), ((64, 64, 64), 1, 1_000), ((24, 40, 8), 1, 7_000)]:
        ph = make_photons(rng, n, dims)
        radius = 0.9 / dims[0]
        got, _, _ = run_fast(ctx, cpm, ph, dims, ch, radius, 0.5)
        want_fast, _ = oracle_both(oracle, ph, dims, ch, radius, 0.5)
        assert np.array_equal(bits(got), bits(want_fast))


def test_fast_unsupported_radius_is_refused(ctx, cpm):
    B = cpm.binding
    grid = B.default_grid_desc((32, 32, 32), 1)
    assert ctx.gather_fast_supported(grid, 0.866 / 32)
    # up to 8 candidate voxels per axis (r < 3.5 voxels: unrolled loops up to 4, run-time loops beyond); 4.2 voxels is refused
    assert ctx.gather_fast_supported(grid, 2.2 / 32) and ctx.gather_fast_supported(grid, 3.4 / 32)
    assert not ctx.gather_fast_supported(grid, 4.2 / 32)
    # an anisotropic grid: the box is as wide as EACH axis' radius asks (the workspace's 256 x 256 x 48 light volume: 6 x 6 x 2)
    assert ctx.gather_fast_supported(B.default_grid_desc((256, 256, 48), 1), 0.010779)
    table = ctx.torch.zeros(ctx.fast_table_entries(grid, 16), dtype=ctx.torch.int32, device=ctx.device)
    assert ctx.fast_record_capacity(grid, 16, 0.866 / 32) == 8 * 16 and ctx.fast_record_capacity(grid, 16, 4.2 / 32) == 0
    assert ctx.fast_record_capacity(grid, 16, 0.2 / 32) == 16      # a candidate box one voxel wide: one brick per photon
    srt = ctx.torch.zeros((8 * 16, 4), dtype=ctx.torch.float32, device=ctx.device)
    out = ctx.torch.zeros(32 ** 3, dtype=ctx.torch.float32, device=ctx.device)
    photons = ctx.torch.zeros((16, 8), dtype=ctx.torch.float32, device=ctx.device)
    with pytest.raises(B.CpmError) as e:
        ctx.bin_fast(photons, 16, grid, 4.2 / 32, table, srt)
    assert e.value.status == -4  # CPM_ERR_UNSUPPORTED: callers fall back to cpm_bin + cpm_gather
    ctx.bin_fast(photons, 16, grid, 0.866 / 32, table, srt)
    with pytest.raises(B.CpmError) as e:
        ctx.gather_fast(srt, table, 16, grid, 4.2 / 32, 1.0, out)
    assert e.value.status == -4
    with pytest.raises(B.CpmError):                                  # the records were filed for another radius
        ctx.gather_fast(srt, table, 16, grid, 1.2 / 32, 1.0, out)
    ctx.gather_fast(srt, table, 16, grid, 0.866 / 32, 1.0, out)


def test_fast_non_default_grid_matrices(ctx, oracle, cpm):
    """A scaled and offset light volume (texture -> index is not dims * p - 0.5): the fast path reads the matrices."""
    B = cpm.binding
    dims = (32, 24, 16)
    rng = np.random.default_rng(5)
    ph = make_photons(rng, 20_000, dims, outside=0.0)
    grid = B.default_grid_desc(dims, 1)
    og = oracle.grid(dims, 1)
    # the light volume covers [0.1, 0.9]^3 of texture space: index = (p - 0.1) / 0.8 * dim - 0.5
    for a in range(3):
        s = np.float32(dims[a]) / np.float32(0.8)
        t = np.float32(-0.1) * s - np.float32(0.5)
        for m in (grid, og):
            m.texture_to_index[5 * a] = s
            m.texture_to_index[12 + a] = t
            m.index_to_texture[5 * a] = np.float32(1.0) / s
            m.index_to_texture[12 + a] = -t / s
    n = ph.shape[0]
    radius, scale = 0.02, 0.3
    table = ctx.torch.zeros(ctx.fast_table_entries(grid, n), dtype=ctx.torch.int32, device=ctx.device)
    srt = ctx.torch.zeros((ctx.fast_record_capacity(grid, n, radius), 4), dtype=ctx.torch.float32, device=ctx.device)
    out = ctx.torch.zeros(dims[0] * dims[1] * dims[2], dtype=ctx.torch.float32, device=ctx.device)
    ctx.bin_fast(_t(ctx, ph), n, grid, radius, table, srt)
    ctx.gather_fast(srt, table, n, grid, radius, scale, out)
    want = np.zeros(out.numel(), np.float32)
    oracle.gather_fast(ph, n, og, radius, scale, want)
    assert np.array_equal(bits(_n(out)), bits(want))
    sp = np.zeros(out.numel(), np.float32)
    oracle.splat(ph, n, og, radius, scale, sp)       # the reference formulation with the same matrices
    np.testing.assert_allclose(_n(out), sp, rtol=1e-4, atol=ATOL_OF_MAX * float(sp.max()))


def test_candidate_box_is_cut_to_the_host_reach(ctx, oracle, cpm):
    """with 2 r' an ulp(u) below an integer the rounding of u - r' / u + r' admits one candidate more than the
    host's floor(2 r') + 1 -- at a brick face that would be a second record for a photon whose capacity is one.  The box is
    cut to the host's reach (the extra candidate lies at r' > r, weight zero): HIP == oracle, one record per photon."""
    dims = (400, 8, 16)
    f32 = np.float32
    rg_target = np.nextafter(f32(0.5), f32(0))                       # 2 r' just below 1 -> reach 1
    radius = float((rg_target - f32(1e-3)) / f32(400))
    grid = cpm.binding.default_grid_desc(dims, 1)
    rg = f32(radius) * f32(grid.texture_to_index[0]) + f32(1e-3)
    assert int(np.floor(f32(2) * (rg))) + 1 == 1
    n = 5000
    rng = np.random.default_rng(5)
    ph = np.zeros((n, 8), f32)
    ph[:, 0] = f32(304.0) / f32(400)                                  # index coordinate 303.5: between voxels 303 | 304 = bricks 37 | 38
    ph[:, 1] = (rng.integers(0, 8, n).astype(f32) + f32(0.5)) / f32(8)     # voxel centres along y and z: r is a hundredth of a voxel there
    ph[:, 2] = (rng.integers(0, 16, n).astype(f32) + f32(0.5)) / f32(16)
    ph[:, 3:6] = 1.0
    u = f32(f32(grid.texture_to_index[0]) * ph[0, 0] + f32(grid.texture_to_index[12]))
    assert np.ceil(f32(u - rg)) < np.floor(f32(u + rg))               # the float box really holds two integers
    assert ctx.fast_record_capacity(grid, n, radius) == n
    scale = float(cpm.binding.relative_irradiance_scale(radius, float(n)))
    got, table, srt = run_fast(ctx, cpm, ph, dims, 1, radius, scale)
    want_fast, _ = oracle_both(oracle, ph, dims, 1, radius, scale)
    assert np.array_equal(bits(got), bits(want_fast))
    assert table[brick_count(dims)] == n                              # one record per photon


def test_non_finite_powers_are_ignored_photon_by_photon(ctx, oracle, cpm):
    """a NaN / inf power neither hides its neighbours' maximum (the fixed-point scale) nor reaches the sums: the
    volume is the one the finite photons alone give -- bit for bit, with HIP == oracle."""
    dims = (32, 32, 32)
    rng = np.random.default_rng(11)
    n = 20_000
    ph = make_photons(rng, n, dims, sentinels=0.05, outside=0.0)
    clean = ph.copy()
    bad = rng.random(n) < 0.02
    bad &= ph[:, 0] != FLT_MAX
    ph[bad, 3] = np.where(rng.random(bad.sum()) < 0.5, np.nan, np.inf).astype(np.float32)
    clean[bad, :3] = FLT_MAX                                          # the same frame without those photons
    clean[bad, 4:6] = FLT_MAX
    assert bad.sum() > 100
    radius = float(np.float32(0.866) / np.float32(32))
    scale = float(cpm.binding.relative_irradiance_scale(radius, float(n)))
    got, _, _ = run_fast(ctx, cpm, ph, dims, 1, radius, scale)
    want, _ = oracle_both(oracle, ph, dims, 1, radius, scale)
    ref, _, _ = run_fast(ctx, cpm, clean, dims, 1, radius, scale)
    assert np.isfinite(got).all() and got.max() > 0
    assert np.array_equal(bits(got), bits(want))
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("dims,channels,rvox", [((128, 128, 128), 1, 0.866), ((40, 24, 56), 4, 1.2), ((21, 7, 5), 1, 0.9), ((256, 256, 48), 1, 2.76),
                                                ((256, 256, 192), 1, 1.0), ((40, 24, 56), 4, 3.1), ((64, 64, 64), 1, 2.6), ((21, 37, 5), 1, 2.4)])
def test_gather_fast_marks_the_nonzero_bricks(ctx, oracle, cpm, dims, channels, rvox):
    """cpm_gather_fast_marked: the same light volume, plus one byte per 4x4x4-voxel brick saying whether it holds a non-zero value --
    every byte written (the buffer starts as garbage), ragged grids, 4 channels, bigger bricks and wide boxes included."""
    rng = np.random.default_rng(dims[0] + channels)
    n = 30_000
    ph = make_photons(rng, n, dims, cluster=0.9)
    radius = float(np.float32(rvox) / np.float32(max(dims)))
    scale = float(cpm.binding.relative_irradiance_scale(radius, float(n)))
    B = cpm.binding
    grid = B.default_grid_desc(dims, channels)
    cells = dims[0] * dims[1] * dims[2]
    table = ctx.torch.zeros(max(ctx.fast_table_entries(grid, n), 1), dtype=ctx.torch.int32, device=ctx.device)
    srt = ctx.torch.zeros((max(ctx.fast_record_capacity(grid, n, radius), 1), 4 if channels == 1 else 8), dtype=ctx.torch.float32, device=ctx.device)
    ctx.bin_fast(_t(ctx, ph), n, grid, radius, table, srt)
    shape = (cells,) if channels == 1 else (cells, 4)
    plain = ctx.torch.full(shape, 7.0, dtype=ctx.torch.float32, device=ctx.device)
    ctx.gather_fast(srt, table, n, grid, radius, scale, plain)
    bxn, byn, bzn = [(d + 3) // 4 for d in dims]
    nb = bxn * byn * bzn
    marks = ctx.torch.full((nb + 16,), 77, dtype=ctx.torch.uint8, device=ctx.device)
    out = ctx.torch.full(shape, 7.0, dtype=ctx.torch.float32, device=ctx.device)
    ctx.gather_fast(srt, table, n, grid, radius, scale, out, nonzero_bricks=marks)
    assert np.array_equal(bits(_n(out)), bits(_n(plain)))
    z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
    b = ((x // 4) + bxn * ((y // 4) + byn * (z // 4))).reshape(-1)
    nz = (_n(out).reshape(cells, -1) != 0).any(axis=1)
    want = np.zeros(nb, np.uint8)
    want[np.unique(b[nz])] = 1
    got = _n(marks)
    assert np.array_equal(got[:nb], want) and (got[nb:] == 77).all() and want.sum() > 0
    with pytest.raises(B.CpmError):     # with accumulate the marks would describe this launch's share only
        ctx.gather_fast(srt, table, n, grid, radius, scale, out, accumulate=True, nonzero_bricks=marks)
