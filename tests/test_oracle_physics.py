"""Self-consistency of the oracle where the reference offers nothing to pin against:
analytic Beer-Lambert behaviour of the Woodcock tracer, equivalence of the sort+gather
formulation with the reference's splat, sort stability, emission lattice, direction codec."""
import numpy as np
import pytest

FLT_MAX = np.float32(3.402823466e+38)


def _trace(oracle, cpm, vol, tf, n_side, direction=(0, 0, 1), max_inter=1, flags=0, shading=1, g=0.0):
    from oracle_binding import OTraceParams
    S, P = cpm.synthetic, cpm.pipeline
    n = n_side * n_side
    d = P._normalize(direction)
    origin = np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * d
    o, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
    area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
    s = oracle.uniform_samples_2d(n_side, n_side)
    ls = oracle.directional_light_samples(s, (1, 1, 1), d, o, u, v, area)
    isect = oracle.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = oracle.glibc_rand_sequence(0, n)
    oracle.seed_streams(st, 1 << 40)
    p = OTraceParams()
    p.material[0] = g
    p.step_size = 1.0 / vol.shape[0]
    p.n_light_samples = n
    p.max_interactions = max_inter
    p.total_photons = n
    p.shading_type = shading
    p.flags = flags
    ph = np.zeros((n * max_inter, 8), np.float32)
    steps = oracle.trace(oracle.volume(vol), tf, S.UNIT_CUBE_AABB, p, ls, isect, st, ph)
    return ph, ls, isect, steps, area


def test_math_contract_accuracy(oracle):
    x = (np.random.default_rng(0).integers(1, 2 ** 32, 50000).astype(np.float32) * np.float32(2.0 ** -32))
    got = oracle.log(x)
    ref = np.log(x.astype(np.float64))
    assert np.max(np.abs(got - ref) / np.spacing(np.abs(ref).astype(np.float32))) < 1.0
    assert oracle.lib.cpmo_log(1.0) == 0.0 and oracle.lib.cpmo_log(0.0) == -np.inf
    a = np.linspace(-2 * np.pi, 2 * np.pi, 20001).astype(np.float32)
    sc = oracle.sincos(a)
    assert np.max(np.abs(sc[:, 0] - np.sin(a.astype(np.float64)))) < 2e-7
    assert np.max(np.abs(sc[:, 1] - np.cos(a.astype(np.float64)))) < 2e-7


def test_direction_codec_round_trip(oracle):
    rng = np.random.default_rng(1)
    d = rng.normal(size=(2000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for v in list(d) + [np.array([0, 0, 1], np.float32), np.array([0, 0, -1], np.float32), np.array([1, 0, 0], np.float32)]:
        back = oracle.decode_direction(oracle.encode_direction(v))
        assert np.max(np.abs(back - v)) < 1e-6


def test_emission_lattice_quirk(oracle):
    # uniformsamplegenerator2d.cl:46: the row coordinate is i / nx, not floor(i / nx)
    s = oracle.uniform_samples_2d(8, 4)
    i = np.arange(32, dtype=np.float32)
    assert np.array_equal(s[:, 0], (np.float32(0.5) + np.fmod(i, np.float32(8))) / np.float32(8))
    assert np.array_equal(s[:, 1], (np.float32(0.5) + i / np.float32(8)) / np.float32(4))
    assert (s[:, 2] == 0).all() and (s[:, 3] == 1).all()


def test_beer_lambert_homogeneous(oracle, cpm):
    """Config 1: constant extinction sigma = alpha * 150 per unit length: collision depths are
    Exp(sigma), the surviving fraction is exp(-sigma L) and the stored power is P / alpha."""
    S = cpm.synthetic
    alpha = 0.02
    ph, ls, isect, steps, area = _trace(oracle, cpm, S.homogeneous_volume(32), S.homogeneous_tf(alpha), 256)
    n = ph.shape[0]
    hit = ph[:, 0] != FLT_MAX
    entered = isect[:, 0] < isect[:, 1]
    # light along +z through the unit cube: all samples enter except the last lattice row, which the
    # unfloored row coordinate (Q14) pushes just past the cube's edge
    assert entered.mean() > 0.99 and not hit[~entered].any()
    sigma = alpha * 150.0
    frac = hit[entered].mean()
    assert abs(frac - (1 - np.exp(-sigma * 1.0))) < 4 * np.sqrt(frac * (1 - frac) / n) + 1e-3
    depth = ph[hit, 2]  # entry plane z = 0
    assert abs(depth.mean() - (1 / sigma - np.exp(-sigma) / (1 - np.exp(-sigma)))) < 0.01
    # power: sample power (area / pdf = area) / max(alpha, 0.01)
    assert np.allclose(ph[hit, 3], np.float32(area) / np.float32(alpha), rtol=1e-6)
    # non-interacting photons: sentinel position, tracked power kept in .s3 (photontracer.cl:199-209)
    assert (ph[~hit, :3] == FLT_MAX).all() and np.allclose(ph[~hit, 3], area, rtol=1e-6)
    frac = hit.mean()
    # Woodcock iteration count: proposals until acceptance or exit ~ 150 per unit length in vacuum
    assert 0.8 * n / alpha * frac < steps < 1.2 * (n / alpha)


def test_gather_equals_splat_and_conserves_power(oracle, cpm):
    S = cpm.synthetic
    ph, *_ = _trace(oracle, cpm, S.heterogeneous_volume(32), S.workspace_tf(), 128, direction=(0.3, 0.5, -1.0))
    n = ph.shape[0]
    for dims, ch in (((16, 16, 16), 1), ((20, 12, 28), 4)):
        og = oracle.grid(dims, ch)
        radius = float(np.float32(0.866 / max(dims)))
        scale = oracle.relative_irradiance_scale(radius, n)
        cells = dims[0] * dims[1] * dims[2]
        shape = (cells,) if ch == 1 else (cells, 4)
        sp = np.zeros(shape, np.float32)
        oracle.splat(ph, n, og, radius, scale, sp)
        order, cs, srt = oracle.bin(ph, n, og)
        ga = np.zeros(shape, np.float32)
        oracle.gather(srt, cs, n, og, radius, scale, ga)
        np.testing.assert_allclose(ga, sp, rtol=2e-5, atol=1e-6 * float(sp.max()))
        assert sp.sum() > 0
        # bin invariants
        valid = ph[:, 0] != FLT_MAX
        assert cs[0] == 0 and cs[-1] == valid.sum() and (np.diff(cs.astype(np.int64)) >= 0).all()
        assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
        assert (ph[order[cs[-1]:], 0] == FLT_MAX).all()
        # threads do not change the result (the parallel loops write disjoint outputs)
        oracle.set_threads(4)
        gb = np.zeros(shape, np.float32)
        oracle.gather(srt, cs, n, og, radius, scale, gb)
        oracle.set_threads(1)
        assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32))


def test_splat_linearity_plus_minus(oracle, cpm):
    """splatSelected(+1) followed by splatSelected(-1) over the same photons cancels exactly
    (the correlated update relies on it: photontolightvolumeprocessorcl.cpp:268-274)."""
    rng = np.random.default_rng(5)
    n = 3000
    ph = np.zeros((n, 8), np.float32)
    ph[:, :3] = rng.random((n, 3), dtype=np.float32)
    ph[:, 3:6] = rng.random((n, 3), dtype=np.float32)
    idx = np.arange(0, n, 3, dtype=np.uint32)
    og = oracle.grid((16, 16, 16), 1)
    out = np.zeros(16 ** 3, np.float32)
    oracle.splat_selected(ph, idx, og, 0.05, 1.5, 1.0, n, 1, out)
    assert out.sum() > 0
    oracle.splat_selected(ph, idx, og, 0.05, 1.5, -1.0, n, 1, out)
    assert np.abs(out).max() <= 1e-6 * 1.5


def test_sort_is_stable_and_respects_key_bits(oracle):
    rng = np.random.default_rng(2)
    for n, bits_ in ((0, 32), (1, 32), (1000, 5), (70000, 21), (70000, 32)):
        keys = rng.integers(0, 1 << bits_, n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        k, v = keys.copy(), vals.copy()
        oracle.sort_pairs(k, v, bits_)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])


def test_multiple_scattering_bookkeeping(oracle, cpm):
    S = cpm.synthetic
    I = 4
    ph, ls, isect, steps, area = _trace(oracle, cpm, S.heterogeneous_volume(32), S.workspace_tf(), 64,
                                        direction=(0.3, 0.5, -1.0), max_inter=I, shading=0, g=0.5)
    n = 64 * 64
    ph = ph.reshape(I, n, 8)  # SoA by interaction: id = k * N + thread (photontracer.cl:166,202)
    stored = ph[:, :, 0] != FLT_MAX
    # interactions are stored contiguously from k = 0; once a sentinel, always a sentinel
    assert (np.diff(stored.astype(np.int8), axis=0) <= 0).all()
    assert stored[0].sum() > 0 and stored[1].sum() > 0
    inside = ph[:, :, :3][stored]
    assert (inside >= 0).all() and (inside <= 1).all()
    # sentinel records carry (P.x, FLT_MAX, FLT_MAX); an absorbed photon's tracked power is FLT_MAX
    sent = ph[~stored]
    assert (sent[:, 4] == FLT_MAX).all() and (sent[:, 5] == FLT_MAX).all()


def test_fast_formulation_restatement_matches_reference_semantics(oracle):
    """cpmo_gather_fast (fixed-point sums, weight from d^2) against cpmo_gather (sequential fp32 sum, weight through
    sqrt and division) and the sequential splat: the tolerance the GPU tests hold cpm_gather_fast to."""
    rng = np.random.default_rng(17)
    for dims, ch, rvox in [((24, 24, 24), 1, 0.866), ((20, 12, 28), 4, 1.3), ((16, 16, 16), 1, 1.9)]:
        n = 20_000
        ph = np.zeros((n, 8), np.float32)
        ph[:, :3] = rng.random((n, 3), dtype=np.float32) * np.float32(1.2) - np.float32(0.1)
        ph[:, 3:6] = rng.random((n, 3), dtype=np.float32) * np.float32(5.0)
        s = rng.random(n) < 0.1
        ph[s, :3] = np.float32(3.402823466e+38)
        radius, scale = float(np.float32(rvox) / np.float32(dims[0])), 0.37
        og = oracle.grid(dims, ch)
        cells = dims[0] * dims[1] * dims[2]
        shape = (cells,) if ch == 1 else (cells, 4)
        fast = np.zeros(shape, np.float32)
        oracle.gather_fast(ph, n, og, radius, scale, fast)
        _, cs, srt = oracle.bin(ph, n, og)
        exact = np.zeros(shape, np.float32)
        oracle.gather(srt, cs, n, og, radius, scale, exact)
        np.testing.assert_allclose(fast, exact, rtol=2e-5, atol=1e-5 * float(exact.max()))
        sp = np.zeros(shape, np.float32)
        oracle.splat(ph, n, og, radius, scale, sp)
        np.testing.assert_allclose(fast, sp, rtol=2e-5, atol=1e-5 * float(sp.max()))
        # order-free: any permutation of the photons gives the same bits
        perm = rng.permutation(n)
        again = np.zeros(shape, np.float32)
        oracle.gather_fast(np.ascontiguousarray(ph[perm]), n, og, radius, scale, again)
        assert np.array_equal(fast.view(np.uint32), again.view(np.uint32))
