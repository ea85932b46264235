"""The oracle against the reference: golden vectors produced by the reference's own OpenCL C
kernels (tests/golden/ref_kernels.npz, generator tests/golden/make_golden.py) and, where
oracle/_ref exists, those kernels themselves."""
import ctypes

import numpy as np
import pytest


def test_glibc_rand_bases(oracle, golden):
    # rndgenmwc64x/mwc64xseedgenerator.cpp:56-64: srand(0); rand() per stream
    b = oracle.glibc_rand_sequence(0, golden["bases"].size)
    assert np.array_equal(b, golden["bases"])
    assert b[0] == 1804289383 and b[1] == 846930886


def test_glibc_rand_other_seeds(oracle):
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 2, 12345, 0xFFFFFFFF):
        libc.srand(seed)
        want = np.array([libc.rand() for _ in range(500)], np.uint32)
        assert np.array_equal(oracle.glibc_rand_sequence(seed, 500), want)


def test_seed_streams_golden(oracle, golden):
    st = np.zeros_like(golden["seeded"])
    st[:, 0] = golden["bases"]
    st[:, 1] = 0xDEADBEEF  # ignored on input
    oracle.seed_streams(st, 1 << 40)
    assert np.array_equal(st, golden["seeded"])
    # SURVEY 8c known answers
    assert tuple(st[0]) == (1691772326, 3037691698) and tuple(st[1]) == (2872186077, 58936352)


def test_per_stream_seed_golden(oracle, golden):
    st = np.zeros_like(golden["per_stream_seeded"])
    st[:, 0] = golden["bases"][: st.shape[0]]
    oracle.seed_streams(st, int(golden["per_stream_gap"]))
    assert np.array_equal(st, golden["per_stream_seeded"])


def test_random_01_golden(oracle, golden):
    st = golden["seeded"].copy()
    out = oracle.random_fill(st, golden["random01"].shape[0])
    assert np.array_equal(out.view(np.uint32), golden["random01"].view(np.uint32))
    assert np.array_equal(st, golden["state_after"])
    np.testing.assert_allclose(out[:3, 0], [0.819719017, 0.756750345, 0.331362218], rtol=0, atol=1e-9)
    # random_01 is the 32-bit output converted to float and scaled by 2^-32
    want = golden["random_uint"].astype(np.float32) * np.float32(2.0 ** -32)
    assert np.array_equal(out, want)


def test_density_kernel_golden(oracle, golden):
    y = np.array([oracle.lib.cpmo_density_kernel(float(x)) for x in golden["kernel_x"]], np.float32)
    assert np.array_equal(y.view(np.uint32), golden["kernel_y"].view(np.uint32))


def test_threshold_count_iota_golden(oracle, golden):
    imp = golden["threshold_in"].copy()
    idx, cnt = oracle.select_recompute(imp)
    assert cnt == int(golden["threshold_out"].sum())
    assert np.array_equal(np.sort(idx), np.arange(imp.size, dtype=np.uint32))
    assert np.array_equal(np.arange(golden["iota"].size, dtype=np.uint32), golden["iota"])
    # stable ascending by importance
    order = np.argsort(golden["threshold_in"], kind="stable").astype(np.uint32)
    assert np.array_equal(idx, order)
    assert np.array_equal(imp, golden["threshold_in"][order])


def test_live_reference_rng(oracle, ref):
    rng = np.random.default_rng(7)
    n = 3000
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = rng.integers(0, 2**31, n)
    a, b = st.copy(), st.copy()
    oracle.seed_streams(a, 1 << 40)
    ref.generate_random_state(b)
    assert np.array_equal(a, b)
    oa = oracle.random_fill(a, 16)
    ob, _ = ref.random_fill(b, 16)
    assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32)) and np.array_equal(a, b)


def test_live_reference_density_kernel(oracle, ref):
    x = np.random.default_rng(3).random(20000, dtype=np.float32) * np.float32(1.2)
    want = ref.density_kernel(x)
    got = np.array([oracle.lib.cpmo_density_kernel(float(v)) for v in x], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_mix_kernel_golden(oracle, golden):
    """mixKernel (uniformgridcl/cl/buffermixer.cl) as the reference builds it for float grids and for the min/max
    grid (ushort2 through convert_float2 / convert_ushort2): the oracle reproduces the captured outputs bit for bit,
    including the index guard (the launch is rounded up to the work-group size)."""
    for k, a in enumerate(golden["mix_a"]):
        got = oracle.mix_f32(golden["mix_x"], golden["mix_y"], float(a))
        assert np.array_equal(got.view(np.uint32), golden["mix_f32"][k].view(np.uint32)), float(a)
        gotu = oracle.mix_u16x2(golden["mix_ux"], golden["mix_uy"], float(a))
        assert np.array_equal(gotu, golden["mix_u16x2"][k]), float(a)
    assert np.array_equal(golden["mix_f32"][0], golden["mix_x"])          # a = 0
    assert np.array_equal(golden["mix_u16x2"][0], golden["mix_ux"])


def test_live_reference_mix_kernel(oracle, ref):
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(5000) * 100).astype(np.float32)
    y = (rng.standard_normal(5000) * 100).astype(np.float32)
    ux = rng.integers(0, 65536, (3001, 2)).astype(np.uint16)
    uy = rng.integers(0, 65536, (3001, 2)).astype(np.uint16)
    for a in (0.0, 0.1, 0.5, 0.73, 1.0):
        assert np.array_equal(oracle.mix_f32(x, y, a).view(np.uint32), ref.mix_f32(x, y, a).view(np.uint32))
        assert np.array_equal(oracle.mix_u16x2(ux, uy, a), ref.mix_u16x2(ux, uy, a))


def test_random_number_kernel_golden(oracle, golden):
    """randomNumberGeneratorKernel (rndgenmwc64x/cl/randomnumbergenerator.cl): state loaded from the uint2 buffer, one
    random_01, state saved back -- three launches in a row.  The oracle's draw-and-write-back gives the same numbers and
    leaves the same states (the write-back photontracer.cl:211-215 performs through the same saveRandState)."""
    st = golden["seeded"][: golden["rng_kernel_state"].shape[0]].copy()
    for k in range(golden["rng_kernel_draws"].shape[0]):
        out = oracle.random_fill(st, 1)
        assert np.array_equal(out[0].view(np.uint32), golden["rng_kernel_draws"][k].view(np.uint32))
    assert np.array_equal(st, golden["rng_kernel_state"])
    # the same streams through random_01 alone (tests above): launch k of the kernel is draw k
    assert np.array_equal(golden["rng_kernel_draws"], golden["random01"][:3, : st.shape[0]])


def test_photon_record_layout_golden(golden):
    """readPhoton / writePhoton (progressivephotonmapping/cl/photon.cl): record id is the 8 consecutive floats at 8 * id
    -- the [n, 8] float32 arrays every test, the oracle and the C-ABI (cpm.h: photons8) address photons by."""
    buf, ids, ph = golden["photon_buffer"], golden["photon_ids"], golden["photon_in"]
    assert np.array_equal(buf[ids].view(np.uint32), ph.view(np.uint32))
    untouched = np.setdiff1d(np.arange(buf.shape[0]), ids)
    assert (buf[untouched] == np.float32(-7)).all()
    assert np.array_equal(golden["photon_read_back"].view(np.uint32), ph.view(np.uint32))


def test_copy_indexed_photons_reads_the_reference_layout(oracle, golden):
    # the oracle's indexed photon copy (copyIndexPhotonsKernel restated) fetches what the reference's readPhoton returns
    buf, ids = golden["photon_buffer"], golden["photon_ids"].astype(np.uint32)
    out = np.zeros((ids.size, 8), np.float32)
    oracle.copy_indexed_photons(np.ascontiguousarray(buf), ids, 1.0, buf.shape[0], 1, out)
    want = golden["photon_read_back"].copy()
    sent = want[:, 0] == np.float32(3.402823466e+38)
    assert np.array_equal(out[~sent].view(np.uint32), want[~sent].view(np.uint32))


def test_live_reference_random_number_kernel_and_photon_records(oracle, ref):
    if ref.lib2 is None:
        pytest.skip("oracle/_ref/libcpm_ref2.so not built")
    rng = np.random.default_rng(21)
    st = np.zeros((777, 2), np.uint32)
    st[:, 0] = rng.integers(0, 2**31, 777)
    oracle.seed_streams(st, 1 << 40)
    a, b = st.copy(), st.copy()
    for _ in range(5):
        assert np.array_equal(oracle.random_fill(a, 1)[0].view(np.uint32), ref.random_number_kernel(b).view(np.uint32))
    assert np.array_equal(a, b)
    ph = rng.standard_normal((100, 8)).astype(np.float32)
    ids = rng.permutation(128)[:100].astype(np.int32)
    buf, back = ref.photon_write_read(ph, ids, 128)
    assert np.array_equal(buf[ids], ph) and np.array_equal(back, ph)
