"""Seeded random sweeps of the bin + gather path (through the C-ABI) against the oracle.

The parametrised parity tests pin the geometries the kernels specialise on; this file throws
shapes at the dispatch that nobody chose by hand: degenerate grids (an axis of 1 cell), grids
smaller than a brick, photon counts around wave / tile sizes, radii across every kernel's range,
photons piled into one cell or spread past the volume's faces.  Bit-exact, as everywhere.
"""
import os

import numpy as np
import pytest

from test_parity_gpu import _n, _t, bits

pytestmark = pytest.mark.gpu

F32_MAX = np.float32(3.402823466e+38)
MORE = int(os.environ.get("CPM_FUZZ_SCALE", "1"))      # CPM_FUZZ_SCALE=20 pytest ...: a longer hunt, same seeds first


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]
    shape_kind = pick("cube", "slab", "line", "tiny", "ragged", "ragged")
    if shape_kind == "cube":
        d = int(pick(8, 16, 24, 32)); dims = (d, d, d)
    elif shape_kind == "slab":
        dims = tuple(int(x) for x in rng.permutation([1, int(rng.integers(5, 40)), int(rng.integers(5, 40))]))
    elif shape_kind == "line":
        dims = tuple(int(x) for x in rng.permutation([1, 1, int(rng.integers(1, 70))]))
    elif shape_kind == "tiny":
        dims = tuple(int(x) for x in rng.integers(1, 4, 3))
    else:
        dims = tuple(int(x) for x in rng.integers(2, 41, 3))
    channels = int(pick(1, 1, 4))
    radius_vox = float(pick(0.05, 0.3, 0.7, 0.866, 0.999, 1.0, 1.2, 1.4999, 1.5, 1.7320508, 1.99, 2.0, 2.4, 3.1))
    n = int(pick(1, 2, 63, 64, 65, 255, 2047, 2048, 2049, 4100, 9000, 30000))
    if pick(0, 0, 0, 0, 0, 1):                                    # now and then a radius of many cells (voxel-major kernel)
        radius_vox, n = float(pick(6.0, 17.0, 40.0)), min(n, 2049)
    layout = pick("uniform", "uniform", "overspill", "one_cell", "faces", "all_missed")
    ph = np.zeros((n, 8), np.float32)
    if layout == "uniform":
        ph[:, :3] = rng.random((n, 3), dtype=np.float32)
    elif layout == "overspill":
        ph[:, :3] = (0.5 + (rng.random((n, 3)) - 0.5) * 1.3).astype(np.float32)
    elif layout == "one_cell":
        c = rng.random(3, dtype=np.float32)
        ph[:, :3] = c + (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(0.4 / max(dims))
    elif layout == "faces":
        ph[:, :3] = rng.integers(0, 2, (n, 3)).astype(np.float32)
        ph[:, int(rng.integers(3))] = rng.random(n, dtype=np.float32)
    else:
        ph[:, :3] = F32_MAX
    if layout != "all_missed" and n > 4:
        ph[:: int(pick(3, 7, 50)), :3] = F32_MAX                    # photons that left the volume
    ph[:, 3:6] = rng.random((n, 3), dtype=np.float32) * 3
    ph[:, 6:] = rng.random((n, 2), dtype=np.float32)
    return dims, channels, radius_vox, n, ph, f"{shape_kind}/{layout}"


@pytest.mark.parametrize("seed", range(48 * MORE))
def test_bin_and_gather_random_shapes(ctx, oracle, cpm, seed):
    dims, channels, radius_vox, n, ph, what = _case(seed)
    note = f"seed {seed}: {what} dims={dims} C={channels} r={radius_vox} vox n={n}"
    radius = float(np.float32(radius_vox / max(dims)))
    scale = oracle.relative_irradiance_scale(radius, n)
    g = cpm.binding.default_grid_desc(dims, channels)
    og = oracle.grid(dims, channels)
    cells = dims[0] * dims[1] * dims[2]
    torch = ctx.torch
    order = torch.empty(n, dtype=torch.int32, device=ctx.device)
    cs = torch.empty(cells + 1, dtype=torch.int32, device=ctx.device)
    srt = torch.empty((n, 4 if channels == 1 else 8), dtype=torch.float32, device=ctx.device)
    ctx.bin(_t(ctx, ph), n, g, order, cs, srt)
    o_order, o_cs, o_srt = oracle.bin(ph, n, og)
    assert np.array_equal(_n(order, np.uint32), o_order), note
    assert np.array_equal(_n(cs, np.uint32), o_cs), note
    valid = int(o_cs[cells])
    assert np.array_equal(bits(_n(srt))[:valid], bits(o_srt)[:valid]), note

    shape = (cells,) if channels == 1 else (cells, 4)
    out = torch.full(shape, -3.0, dtype=torch.float32, device=ctx.device)
    ctx.gather(srt, cs, n, g, radius, scale, out)
    want = np.full(shape, -3.0, np.float32)
    oracle.gather(o_srt, o_cs, n, og, radius, scale, want)
    assert np.array_equal(bits(_n(out)), bits(want)), note
    ctx.gather(srt, cs, n, g, radius, scale, out, accumulate=True)
    oracle.gather(o_srt, o_cs, n, og, radius, scale, want, accumulate=True)
    assert np.array_equal(bits(_n(out)), bits(want)), note


@pytest.mark.parametrize("seed", range(48 * MORE))
def test_fast_formulation_random_shapes(ctx, oracle, cpm, seed):
    """The tolerance-mode formulation over the same random cases: bit for bit against its restatement (integer sums), within
    rtol 2e-5 / atol 1e-5 max of the reference-semantics gather, the same bits on a second run and in accumulate mode; refused
    (not silently wrong) where cpm_gather_fast_supported says so."""
    dims, channels, radius_vox, n, ph, what = _case(seed)
    note = f"seed {seed}: {what} dims={dims} C={channels} r={radius_vox} vox n={n}"
    radius = float(np.float32(radius_vox / max(dims)))
    scale = oracle.relative_irradiance_scale(radius, n)
    g = cpm.binding.default_grid_desc(dims, channels)
    og = oracle.grid(dims, channels)
    cells = dims[0] * dims[1] * dims[2]
    torch = ctx.torch
    shape = (cells,) if channels == 1 else (cells, 4)
    table = torch.zeros(ctx.fast_table_entries(g, n), dtype=torch.int32, device=ctx.device)
    out = torch.full(shape, -3.0, dtype=torch.float32, device=ctx.device)
    d_ph = _t(ctx, ph)
    if not ctx.gather_fast_supported(g, radius):
        srt = torch.zeros((8 * n, 4 if channels == 1 else 8), dtype=torch.float32, device=ctx.device)
        with pytest.raises(cpm.binding.CpmError):
            ctx.bin_fast(d_ph, n, g, radius, table, srt)
        with pytest.raises(cpm.binding.CpmError):
            ctx.gather_fast(srt, table, n, g, radius, scale, out)
        return
    srt = torch.zeros((ctx.fast_record_capacity(g, n, radius), 4 if channels == 1 else 8), dtype=torch.float32, device=ctx.device)
    ctx.bin_fast(d_ph, n, g, radius, table, srt)
    ctx.gather_fast(srt, table, n, g, radius, scale, out)
    want = np.zeros(shape, np.float32)
    oracle.gather_fast(ph, n, og, radius, scale, want)
    got = _n(out).copy()
    assert np.array_equal(bits(got), bits(want)), note
    _, o_cs, o_srt = oracle.bin(ph, n, og)
    exact = np.zeros(shape, np.float32)
    oracle.gather(o_srt, o_cs, n, og, radius, scale, exact)
    # tolerance: rtol 2e-5 plus 1e-5 of the larger of the volume's maximum and ONE full-weight contribution
    # (max |power| * k * 0.75): a voxel reached only from the rim of the kernel (weight 0.75 (1 - d^2/r^2) -> 0) is a
    # difference of nearly equal numbers in either formulation, its error is absolute, not relative (seed 1234)
    stored = ph[ph[:, 0] != F32_MAX]
    pmax = float(np.abs(stored[:, 3:6] if channels == 4 else stored[:, 3:4]).max()) if stored.size else 0.0
    one = pmax * abs(scale) * 0.0795774715459476679 * 0.75
    np.testing.assert_allclose(got, exact, rtol=2e-5, atol=1e-5 * max(float(np.abs(exact).max()), one), err_msg=note)
    # again (the bin's two histograms alternate), accumulating on top
    ctx.bin_fast(d_ph, n, g, radius, table, srt)
    ctx.gather_fast(srt, table, n, g, radius, scale, out, accumulate=True)
    oracle.gather_fast(ph, n, og, radius, scale, want, accumulate=True)
    assert np.array_equal(bits(_n(out)), bits(want)), note


@pytest.mark.parametrize("seed", range(12 * MORE))
def test_sort_pairs_random_lengths_and_bits(ctx, seed):
    rng = np.random.default_rng(77 + seed)
    n = int(rng.choice([1, 2, 63, 65, 2047, 2049, 4096, 6143, 50_001, 300_000]))
    key_bits = int(rng.choice([1, 3, 8, 9, 16, 17, 22, 24, 25, 32]))
    hi = (1 << key_bits) - 1
    kind = int(rng.integers(3))
    if kind == 0:
        keys = rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32)
    elif kind == 1:                                                # a handful of distinct keys
        keys = rng.choice(rng.integers(0, hi + 1, 5, dtype=np.uint64), n).astype(np.uint32)
    else:                                                          # already sorted, descending
        keys = np.sort(rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    vals = rng.permutation(n).astype(np.uint32)
    kd, vd = _t(ctx, keys), _t(ctx, vals)
    ctx.sort_pairs(kd, vd, key_bits)
    o = np.argsort(keys, kind="stable")
    assert np.array_equal(_n(kd, np.uint32), keys[o]), (n, key_bits, kind)
    assert np.array_equal(_n(vd, np.uint32), vals[o]), (n, key_bits, kind)


@pytest.mark.parametrize("seed", range(24 * MORE))
def test_trace_random_setups(ctx, oracle, cpm, seed):
    """Random volumes (shape, voxel type, content), transfer functions (width, opacity range), light
    directions / point lights, interaction counts, phase functions and kernel flags: photons and
    RNG streams bit-identical to the oracle's."""
    from test_parity_gpu import _trace_case
    rng = np.random.default_rng(4000 + seed)
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]
    S, B = cpm.synthetic, cpm.binding
    shape = tuple(int(x) for x in rng.integers(2, 49, 3))            # [z, y, x]
    if pick(0, 0, 1):
        shape = tuple(int(x) for x in rng.permutation([1, int(rng.integers(2, 40)), int(rng.integers(2, 40))]))
        if shape[2] < 2:                                              # cpm_volume_create: at least 2 voxels along x
            shape = (shape[2], shape[1], shape[0])
    content = pick("noise", "smooth", "constant", "empty")
    if content == "noise":
        v01 = rng.random(shape)
    elif content == "smooth":
        z, y, x = np.meshgrid(*[np.linspace(0, 1, s) for s in shape], indexing="ij")
        v01 = 0.5 + 0.5 * np.sin(7 * x + 3 * y * y + 5 * z)
    elif content == "constant":
        v01 = np.full(shape, rng.random())
    else:
        v01 = np.zeros(shape)
    dtype = pick(np.uint8, np.uint8, np.uint16, np.float32)
    fmt = (0.0, 0.0)
    if dtype == np.uint8:
        vol = np.rint(v01 * 255).astype(np.uint8)
    elif dtype == np.uint16:
        if pick(0, 1):
            vol, fmt = np.rint(v01 * 4095).astype(np.uint16), (1.0 - 65535.0 / 4095.0, 0.0)
        else:
            vol = np.rint(v01 * 65535).astype(np.uint16)
    else:
        vol = v01.astype(np.float32)
    width = int(pick(2, 17, 256, 1024))
    a_hi = float(pick(0.0, 0.05, 0.6, 1.0))
    pts = [(0, 1, 1, 1, float(pick(0.0, 0.02))), (float(rng.uniform(0.2, 0.8)), 0.3, 0.9, 0.2, a_hi * float(rng.random())),
           (1, 1, 0.5, 0.1, a_hi)]
    tf = S.tf_from_points(pts, width=width)
    max_inter = int(pick(1, 1, 2, 5))
    flags = 0
    if max_inter > 1 and pick(0, 1):
        flags |= B.CPM_TRACE_NO_SINGLE_SCATTERING
    if pick(0, 1):
        flags |= B.CPM_TRACE_PROGRESSIVE
    shading, g = pick((0, 0.0), (1, 0.0), (0, 0.7), (0, -0.5), (0, 0.95))
    tfs = S.tf_from_points([(0, 1, 1, 1, 0.1), (1, 1, 1, 1, 0.9)], width=width) if (max_inter > 1 and pick(0, 1)) else None   # Inviwo TFs share one texture width
    direction = tuple(float(x) for x in rng.normal(size=3))
    if pick(0, 0, 0, 1):
        direction = pick((1.0, 0.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0))     # axis-aligned rays along cell boundaries
    point = tuple(float(x) for x in rng.uniform(-1.5, 2.5, 3)) if pick(0, 0, 1) else None
    n_side = int(pick(1, 7, 33, 64, 100))
    note = (f"seed {seed}: vol {shape} {np.dtype(dtype).name} {content} tf{width} a<={a_hi} I={max_inter} flags={flags} "
            f"shading={shading} g={g} dir={direction} point={point} n={n_side}^2")
    got, want, rng_g, rng_w, *_ = _trace_case(ctx, oracle, cpm, vol, tf, n_side, direction, max_inter=max_inter, flags=flags,
                                              point=point, shading=shading, g=g, tfs=tfs, fmt=fmt)
    assert np.array_equal(bits(got), bits(want)), note
    assert np.array_equal(rng_g, rng_w), note


def test_pointer_alignment_rules(ctx, oracle, cpm):
    """include/cpm/cpm.h, conventions: float8 / record buffers must be 16-byte aligned (refused otherwise);
    u32 tables need only 4 bytes -- the cell-start kernel's 16-byte fast path must not assume more."""
    rng = np.random.default_rng(3)
    dims, n = (20, 17, 9), 3000
    ph = np.zeros((n, 8), np.float32)
    ph[:, :3] = rng.random((n, 3), dtype=np.float32)
    ph[:, 3:6] = rng.random((n, 3), dtype=np.float32)
    g, og = cpm.binding.default_grid_desc(dims, 1), oracle.grid(dims, 1)
    cells = dims[0] * dims[1] * dims[2]
    torch = ctx.torch
    o_order, o_cs, o_srt = oracle.bin(ph, n, og)
    srt = torch.empty((n, 4), dtype=torch.float32, device=ctx.device)
    for shift in (1, 2, 3):                                    # tables at 4, 8, 12 bytes past a 16-byte boundary
        order = torch.empty(n + shift, dtype=torch.int32, device=ctx.device)[shift:]
        cs = torch.empty(cells + 1 + shift, dtype=torch.int32, device=ctx.device)[shift:]
        assert cs.data_ptr() % 16 == 4 * shift
        ctx.bin(_t(ctx, ph), n, g, order, cs, srt)
        assert np.array_equal(_n(order, np.uint32), o_order)
        assert np.array_equal(_n(cs, np.uint32), o_cs)
        out = torch.empty(cells + shift, dtype=torch.float32, device=ctx.device)[shift:]
        radius = float(np.float32(0.9 / max(dims)))
        ctx.gather(srt, cs, n, g, radius, 1.0, out)
        want = np.zeros(cells, np.float32)
        oracle.gather(o_srt, o_cs, n, og, radius, 1.0, want)
        assert np.array_equal(bits(_n(out)), bits(want))
    order = torch.empty(n, dtype=torch.int32, device=ctx.device)
    cs = torch.empty(cells + 1, dtype=torch.int32, device=ctx.device)
    flat = torch.zeros(n * 8 + 1, dtype=torch.float32, device=ctx.device)
    with pytest.raises(cpm.binding.CpmError, match="16-byte aligned"):
        ctx.bin(flat[1:].view(n, 8), n, g, order, cs, srt)     # photons 4 bytes off
    rec = torch.zeros(n * 4 + 2, dtype=torch.float32, device=ctx.device)
    with pytest.raises(cpm.binding.CpmError, match="16-byte aligned"):
        ctx.bin(_t(ctx, ph), n, g, order, cs, rec[2:].view(n, 4))
    with pytest.raises(cpm.binding.CpmError, match="16-byte aligned"):
        ctx.gather(rec[2:].view(n, 4), cs, n, g, 0.05, 1.0, torch.empty(cells, dtype=torch.float32, device=ctx.device))


@pytest.mark.parametrize("seed", range(16 * MORE))
def test_brick_analysis_random_shapes(ctx, oracle, seed):
    """min/max bricks, mean-abs-difference bricks and the TF importance over random volume shapes (rows that are and
    are not 16-byte multiples, bricks cut by every face), voxel types and brick sizes."""
    rng = np.random.default_rng(8000 + seed)
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]
    dtype = pick(np.uint8, np.uint8, np.uint16, np.float32)
    shape = tuple(int(x) for x in rng.integers(2, 70, 3))             # [z, y, x]
    if pick(0, 1):
        shape = shape[:2] + (int(pick(16, 32, 48, 64, 80)),)            # rows the streaming kernels take
    region = int(pick(1, 2, 3, 4, 5, 8, 8, 16, 64))
    if dtype == np.float32:
        a, b = rng.random(shape, dtype=np.float32), rng.random(shape, dtype=np.float32)
    else:
        hi = np.iinfo(dtype).max + 1
        a, b = rng.integers(0, hi, shape).astype(dtype), rng.integers(0, hi, shape).astype(dtype)
    if pick(0, 1):
        b = a.copy()                                                  # nothing changed between the time steps
    note = f"seed {seed}: {shape} {np.dtype(dtype).name} region {region}"
    va, vb = ctx.volume_create(a), ctx.volume_create(b)
    oa, ob = oracle.volume(a), oracle.volume(b)
    nb = int(np.prod([(d + region - 1) // region for d in shape]))
    mm = ctx.torch.zeros((nb, 2), dtype=ctx.torch.int16, device=ctx.device)
    ctx.volume_minmax(va, region, mm)
    want_mm = oracle.volume_minmax(oa, region)
    assert np.array_equal(_n(mm, np.uint16), want_mm), note
    diff = ctx.torch.zeros(nb, dtype=ctx.torch.float32, device=ctx.device)
    ctx.volume_difference(va, vb, region, diff)
    want_diff = oracle.volume_difference(oa, ob, region)
    assert np.array_equal(bits(_n(diff)), bits(want_diff)), note
    # TF importance of those bricks: random |TF_new - TF_old| break points, zero-padded at 0 and 1
    k = int(rng.integers(1, 9))
    pos = np.concatenate([[0.0], np.sort(rng.random(k)), [1.0]]).astype(np.float32)
    col = np.concatenate([np.zeros((1, 4)), rng.random((k, 4)) * (rng.random((k, 1)) < 0.7), np.zeros((1, 4))]).astype(np.float32)
    out = ctx.torch.zeros(nb, dtype=ctx.torch.float32, device=ctx.device)
    ctx.importance_tf(mm, nb, pos, col, out)
    assert np.array_equal(bits(_n(out)), bits(oracle.importance_tf(want_mm, pos, col))), note
    mm2 = ctx.torch.zeros((nb, 2), dtype=ctx.torch.int16, device=ctx.device)
    ctx.volume_minmax(vb, region, mm2)
    ctx.importance_tf(mm2, nb, pos, col, out, prev_minmax=mm, volume_diff=diff)
    want = oracle.importance_tf(_n(mm2, np.uint16), pos, col, prev=want_mm, diff=want_diff)
    assert np.array_equal(bits(_n(out)), bits(want)), note


@pytest.mark.parametrize("seed", range(16 * MORE))
def test_photon_importance_random_segments(ctx, oracle, cpm, seed):
    """The importance DDA over random grids (shape, cell size), random light samples (hits, misses, grazing and
    axis-parallel rays) and random stored photons (interior points, sentinels, multiple interactions)."""
    from oracle_binding import default_matrices
    rng = np.random.default_rng(9000 + seed)
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]
    S = cpm.synthetic
    region = int(pick(1, 2, 4, 8, 8, 16))
    gd = tuple(int(x) for x in rng.integers(1, 20, 3))                # importance cells per axis
    vdims = tuple(g * region - int(pick(0, 0, region - 1)) for g in gd)
    vdims = tuple(max(v, 2) for v in vdims)
    gd = tuple((v + region - 1) // region for v in vdims)
    n, I = int(pick(1, 63, 64, 1000, 5000)), int(pick(1, 1, 2, 4))
    ls = np.zeros((n, 8), np.float32)
    ls[:, :3] = rng.uniform(-1.0, 2.0, (n, 3)).astype(np.float32)
    ls[:, 3:6] = 1
    ls[:, 6] = rng.uniform(0, np.pi, n).astype(np.float32)
    ls[:, 7] = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    ax = rng.random(n) < 0.15                                         # axis-parallel directions
    ls[ax, 6] = rng.choice(np.array([0.0, np.pi / 2, np.pi], np.float32), int(ax.sum()))
    ls[ax, 7] = rng.choice(np.array([0.0, np.pi / 2, -np.pi / 2, np.pi], np.float32), int(ax.sum()))
    isect = oracle.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
    ph = np.zeros((n * I, 8), np.float32)
    ph[:, :3] = rng.random((n * I, 3), dtype=np.float32)
    ph[:, 3:6] = rng.random((n * I, 3), dtype=np.float32)
    ph[:, 6] = rng.uniform(0, np.pi, n * I).astype(np.float32)
    ph[:, 7] = rng.uniform(-np.pi, np.pi, n * I).astype(np.float32)
    gone = rng.random(n * I) < 0.3
    ph[gone, :3] = F32_MAX
    dead = gone & (rng.random(n * I) < 0.5)
    ph[dead, 3] = F32_MAX                                             # absorbed: power.x == FLT_MAX
    grid = rng.random(gd[0] * gd[1] * gd[2], dtype=np.float32) * np.float32(pick(1.0, 1e-3, 50.0))
    grid[rng.random(grid.size) < 0.4] = 0
    t2i, _ = default_matrices(vdims)
    fix = bool(pick(0, 1))
    off = int(pick(0, 0, 3)) if n > 8 else 0
    imp0 = np.full(n + 4, 2147483647, np.uint32)
    imp0[::7] = 2147483000                                            # already carrying importance from an earlier edit
    imp_o = imp0.copy()
    oracle.photon_importance(grid, gd, (region,) * 3, t2i, ph, off, ls, isect, n - off, I, n, imp_o, fix_exit_point=fix)
    imp_d = _t(ctx, imp0)
    ctx.photon_importance(_t(ctx, grid), gd, (float(region),) * 3, t2i.tolist(), _t(ctx, ph), off, _t(ctx, ls), _t(ctx, isect),
                          n - off, I, n, imp_d, fix_exit_point=fix)
    assert np.array_equal(_n(imp_d, np.uint32), imp_o), f"seed {seed}: grid {gd} region {region} vol {vdims} n={n} I={I} fix={fix} off={off}"
