"""Parity of the HIP path (through the C-ABI) against the oracle, on a real MI355X.

Bit-exact for integer work (RNG, seeds, sort, bin order, cell starts) and -- because both
sides implement the same arithmetic contract -- also for the floating-point
kernels; the atomic splat is order-dependent and is compared with a stated tolerance.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(ctx, a):
    """numpy -> device tensor (uint32 travels as int32 bits)."""
    torch = ctx.torch
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).to(ctx.device)
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).to(ctx.device)
    return torch.from_numpy(a).to(ctx.device)


def _n(t, dtype=None):
    a = t.detach().cpu().numpy()
    if dtype is not None:
        a = a.view(dtype)
    return a


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ----------------------------------------------------------------------------- RNG

def test_seed_streams_and_random_fill(ctx, oracle, golden):
    n = 100_000
    bases = oracle.glibc_rand_sequence(0, n)
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = bases
    d = _t(ctx, st)
    ctx.seed_streams(d, 1 << 40)
    want = st.copy()
    oracle.seed_streams(want, 1 << 40)
    got = _n(d, np.uint32)
    assert np.array_equal(got, want)
    assert np.array_equal(got[: golden["seeded"].shape[0]], golden["seeded"])  # the reference's own output
    out = ctx.random_fill(d, 8)
    w = oracle.random_fill(want, 8)
    assert np.array_equal(bits(_n(out)), bits(w))
    assert np.array_equal(_n(d, np.uint32), want)
    assert np.array_equal(bits(_n(out))[:, : golden["random01"].shape[1]], bits(golden["random01"]))


def test_reference_photon_records_and_rng_kernel(ctx, golden):
    """Golden outputs of the reference's photon.cl (writePhoton / readPhoton) and randomnumbergenerator.cl
    (load state -> random_01 -> save state), tests/golden/ref_kernels.npz: the HIP side reads the same records and
    draws the same numbers with the same state written back."""
    buf, ids = golden["photon_buffer"], golden["photon_ids"].astype(np.uint32)
    out = ctx.torch.zeros((ids.size, 8), dtype=ctx.torch.float32, device=ctx.device)
    ctx.copy_indexed_photons(_t(ctx, buf), _t(ctx, ids), ids.size, 1.0, buf.shape[0], 1, out)
    want = golden["photon_read_back"]
    keep = want[:, 0] != np.float32(3.402823466e+38)  # (sentinel records are not copied by the indexed splat's copy)
    assert np.array_equal(bits(_n(out))[keep], bits(want)[keep])
    st = _t(ctx, golden["seeded"][: golden["rng_kernel_state"].shape[0]].copy())
    for k in range(golden["rng_kernel_draws"].shape[0]):
        draw = ctx.random_fill(st, 1)
        assert np.array_equal(bits(_n(draw))[0], bits(golden["rng_kernel_draws"][k]))
    assert np.array_equal(_n(st, np.uint32), golden["rng_kernel_state"])


def test_seed_streams_other_gap(ctx, oracle):
    rng = np.random.default_rng(5)
    st = np.zeros((5000, 2), np.uint32)
    st[:, 0] = rng.integers(0, 2**32, 5000, dtype=np.uint64).astype(np.uint32)
    for gap in (1, 1000, (1 << 44) + 12345):
        d = _t(ctx, st)
        ctx.seed_streams(d, gap)
        want = st.copy()
        oracle.seed_streams(want, gap)
        assert np.array_equal(_n(d, np.uint32), want)


# ----------------------------------------------------------------------------- emission

def test_uniform_samples_2d(ctx, oracle):
    for nx, ny in ((256, 256), (1024, 1024), (7, 13), (1, 1)):
        got = _n(ctx.uniform_samples_2d(nx, ny))
        assert np.array_equal(bits(got), bits(oracle.uniform_samples_2d(nx, ny)))
    assert ctx.uniform_samples_2d(0, 5).shape[0] == 0


def _light_setup(cpm, direction, dist=2.0):
    P = cpm.pipeline
    d = P._normalize(direction)
    origin = np.array([0.5, 0.5, 0.5], np.float32) - np.float32(dist) * d
    o, u, v = P.fit_plane_aligned_obb(cpm.synthetic.UNIT_CUBE_VERTICES, origin, d)
    area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
    return d, o, u, v, area


@pytest.mark.parametrize("direction", [(0, 0, 1), (0.3, 0.5, -1.0), (-1, 0.2, 0.1), (0, -1, 0)])
def test_directional_light_and_intersection(ctx, oracle, cpm, direction):
    d, o, u, v, area = _light_setup(cpm, direction)
    s = oracle.uniform_samples_2d(128, 128)
    ls = ctx.directional_light_samples(_t(ctx, s), (1, 0.5, 0.25), d, o, u, v, area)
    want = oracle.directional_light_samples(s, (1, 0.5, 0.25), d, o, u, v, area)
    assert np.array_equal(bits(_n(ls)), bits(want))
    aabb = cpm.synthetic.UNIT_CUBE_AABB
    isect = ctx.light_sample_box_intersection(ls, aabb)
    wi = oracle.light_sample_box_intersection(want, aabb)
    assert np.array_equal(bits(_n(isect)), bits(wi))
    # clipped box, including misses -> (0, -1)
    clip = (0.25, 0.25, 0.25, 1, 0.5, 0.75, 0.75, 1)
    isect = ctx.light_sample_box_intersection(ls, clip)
    wi = oracle.light_sample_box_intersection(want, clip)
    assert np.array_equal(bits(_n(isect)), bits(wi))
    assert (wi[:, 1] == -1).any()


def test_point_light_and_mesh_intersection(ctx, oracle, cpm):
    s = oracle.uniform_samples_2d(64, 64)
    ls = ctx.point_light_samples(_t(ctx, s), (1, 1, 1), (0.5, 0.5, -1.0))
    want = oracle.point_light_samples(s, (1, 1, 1), (0.5, 0.5, -1.0))
    assert np.array_equal(bits(_n(ls)), bits(want))
    vtx = cpm.synthetic.UNIT_CUBE_VERTICES
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    idx = np.array([i for q in quads for i in (q[0], q[1], q[2], q[0], q[2], q[3])], np.int32)
    got = ctx.light_sample_mesh_intersection(_t(ctx, vtx), _t(ctx, idx), ls)
    wm = oracle.light_sample_mesh_intersection(vtx, idx, want)
    assert np.array_equal(bits(_n(got)), bits(wm))


# ----------------------------------------------------------------------------- trace

def _trace_case(ctx, oracle, cpm, volume, tf, n_side, direction, max_inter=1, flags=0, point=None,
                shading=0, g=0.0, tfs=None, fmt=(0.0, 0.0), make_volume=None):
    from oracle_binding import OTraceParams
    B = cpm.binding
    n = n_side * n_side
    d, o, u, v, area = _light_setup(cpm, direction)
    s = oracle.uniform_samples_2d(n_side, n_side)
    if point is None:
        ls = oracle.directional_light_samples(s, (1, 0.8, 0.6), d, o, u, v, area)
    else:
        ls = oracle.point_light_samples(s, (1, 0.8, 0.6), point)
    aabb = cpm.synthetic.UNIT_CUBE_AABB
    isect = oracle.light_sample_box_intersection(ls, aabb)
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = oracle.glibc_rand_sequence(0, n)
    oracle.seed_streams(st, 1 << 40)

    code = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.float32): 2}[volume.dtype]
    desc = B.default_volume_desc(volume.shape[::-1], code)
    desc.format_scaling, desc.format_offset = fmt
    vol = ctx.volume_create(volume, desc) if make_volume is None else make_volume(volume, desc)
    tfh = ctx.tf_create(tf)
    tfsh = ctx.tf_create(tfs) if tfs is not None else None
    p = B.TraceParams()
    po = OTraceParams()
    for q in (p, po):
        q.material[0] = g
        q.step_size = 1.0 / max(volume.shape)
        q.photon_offset = 0
        q.n_light_samples = n
        q.max_interactions = max_inter
        q.total_photons = n
        q.shading_type = shading
        q.flags = flags
    rng_d = _t(ctx, st)
    photons_d = ctx.torch.zeros((n * max_inter, 8), dtype=ctx.torch.float32, device=ctx.device)
    ctx.trace(vol, tfh, aabb, p, _t(ctx, ls), _t(ctx, isect), rng_d, photons_d, tf_scattering=tfsh)
    # the same trace with the emission chain evaluated in the tracer (cpm_trace_emitted): identical photons and RNG states
    rad = (1, 0.8, 0.6)
    em = (B.directional_emitter(n_side, n_side, rad, d, o, u, v, area) if point is None
          else B.point_emitter(n_side, n_side, rad, point))
    rng_e = _t(ctx, st)
    photons_e = ctx.torch.zeros((n * max_inter, 8), dtype=ctx.torch.float32, device=ctx.device)
    ctx.trace_emitted(vol, tfh, aabb, p, em, rng_e, photons_e, tf_scattering=tfsh)
    assert np.array_equal(bits(_n(photons_e)), bits(_n(photons_d)))
    assert np.array_equal(_n(rng_e, np.uint32), _n(rng_d, np.uint32))

    ovol = oracle.volume(volume, *fmt)
    rng_o = st.copy()
    photons_o = np.zeros((n * max_inter, 8), np.float32)
    steps = oracle.trace(ovol, tf, aabb, po, ls, isect, rng_o, photons_o, tf_scattering=tfs)
    return _n(photons_d), photons_o, _n(rng_d, np.uint32), rng_o, steps, (ls, isect, st)


def test_trace_config1_homogeneous(ctx, oracle, cpm):
    S = cpm.synthetic
    got, want, rng_g, rng_w, steps, _ = _trace_case(ctx, oracle, cpm, S.homogeneous_volume(64), S.homogeneous_tf(0.25), 128, (0, 0, 1))
    assert np.array_equal(bits(got), bits(want))
    assert np.array_equal(rng_g, rng_w)  # not progressive: state untouched
    assert steps > 0


def test_trace_config1_point_light(ctx, oracle, cpm):
    S = cpm.synthetic
    got, want, *_ = _trace_case(ctx, oracle, cpm, S.homogeneous_volume(64), S.homogeneous_tf(0.25), 128, (0, 0, 1),
                                point=(0.5, 0.5, -1.0))
    assert np.array_equal(bits(got), bits(want))
    assert (want[:, 0] == np.float32(3.402823466e+38)).any()  # rays that miss the volume leave sentinels


@pytest.mark.parametrize("direction", [(0.3, 0.5, -1.0), (1, 0, 0), (-0.2, -1, 0.3)])
def test_trace_heterogeneous_workspace_tf(ctx, oracle, cpm, direction):
    S = cpm.synthetic
    got, want, *_ = _trace_case(ctx, oracle, cpm, S.heterogeneous_volume(64), S.workspace_tf(), 160, direction)
    assert np.array_equal(bits(got), bits(want))


def test_trace_sparse_tf_long_paths(ctx, oracle, cpm):
    # mostly transparent medium: hundreds of Woodcock iterations per photon, many exits
    S = cpm.synthetic
    tf = S.tf_from_points([(0.0, 1, 1, 1, 0.0), (0.7, 1, 1, 1, 0.0), (0.9, 1, 1, 1, 0.05), (1.0, 1, 1, 1, 0.3)])
    got, want, _, _, steps, _ = _trace_case(ctx, oracle, cpm, S.heterogeneous_volume(96), tf, 128, (0.3, 0.5, -1.0))
    assert np.array_equal(bits(got), bits(want))
    assert steps / (128 * 128) > 50


def test_trace_progressive_rng_write_back_single_interaction(ctx, oracle, cpm):
    # photontracer.cl:211-215 at max_interactions == 1: the stream state after the photon's last draw
    S = cpm.synthetic
    got, want, rng_g, rng_w, steps, (_, _, st) = _trace_case(ctx, oracle, cpm, S.heterogeneous_volume(64), S.workspace_tf(), 160,
                                                             (0.3, 0.5, -1.0), flags=cpm.binding.CPM_TRACE_PROGRESSIVE)
    assert np.array_equal(bits(got), bits(want))
    assert np.array_equal(rng_g, rng_w)
    assert (rng_g != st).any()


def test_trace_sample_counts_off_the_wave_size(ctx, oracle, cpm):
    # sample counts that are not multiples of a wave / workgroup; a point light (per-lane directions, many misses)
    S = cpm.synthetic
    for n_side in (1, 7, 9, 33):
        got, want, *_ = _trace_case(ctx, oracle, cpm, S.heterogeneous_volume(32), S.workspace_tf(), n_side, (0.3, 0.5, -1.0))
        assert np.array_equal(bits(got), bits(want))
    got, want, *_ = _trace_case(ctx, oracle, cpm, S.heterogeneous_volume(32), S.workspace_tf(), 77, (0, 0, 1), point=(0.4, 0.6, -0.7))
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_trace_other_voxel_types(ctx, oracle, cpm, dtype):
    S = cpm.synthetic
    v8 = S.heterogeneous_volume(48)
    if dtype == np.uint16:
        vol, fmt = (v8.astype(np.uint16) * 16), (1.0 - 65535.0 / 4095.0, 0.0)  # 12-bit data in 16 bits
    else:
        vol, fmt = (v8.astype(np.float32) / np.float32(255)), (0.0, 0.0)
    got, want, *_ = _trace_case(ctx, oracle, cpm, vol, S.workspace_tf(), 96, (0.3, 0.5, -1.0), fmt=fmt)
    assert np.array_equal(bits(got), bits(want))


def test_trace_nonpow2_volume_and_small_tf(ctx, oracle, cpm):
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 256, (19, 33, 50), dtype=np.uint8)  # [z, y, x]
    tf = cpm.synthetic.tf_from_points([(0, 1, 1, 1, 0.02), (1, 1, 1, 1, 0.6)], width=17)
    got, want, *_ = _trace_case(ctx, oracle, cpm, vol, tf, 100, (0.2, -0.4, 1.0))
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("case", ["directional", "point", "shard", "four interactions", "progressive", "re-trace by index"])
def test_emitted_trace_equals_buffer_trace(ctx, cpm, case):
    """PhotonFrame with the emission chain in the tracer (cpm_trace_emitted) against the same frame tracing
    from the emitters' buffers: bit-identical photons and RNG states -- whole lattices and shards (first_sample), both
    light kinds, several interactions, RNG write-back, and the PHOTON_RECOMPUTATION variant (thread j -> sample indices[j])."""
    S, P, B = cpm.synthetic, cpm.pipeline, cpm.binding
    torch = ctx.torch
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0))
    if case == "point":
        kw = dict(point_light_position=(0.5, 0.45, 2.5))
    if case == "shard":
        kw["photon_range"] = (1000, 9000)
    if case == "four interactions":
        kw["max_interactions"] = 4
    vol, tf = S.heterogeneous_volume(48), S.workspace_tf()
    a = P.PhotonFrame(ctx, vol, tf, (112, 96), (32, 32, 32), emit_in_tracer=True, **kw)
    b = P.PhotonFrame(ctx, vol, tf, (112, 96), (32, 32, 32), **kw)
    assert a.emitter is not None and b.emitter is None
    if case == "progressive":
        a.params.flags = b.params.flags = B.CPM_TRACE_PROGRESSIVE
    idx = None
    if case == "re-trace by index":
        a.trace(); b.trace()
        a.photons.zero_(); b.photons.zero_()
        pick = np.sort(np.random.default_rng(3).choice(a.n, 777, replace=False)).astype(np.int32)
        idx = torch.from_numpy(pick).to(ctx.device)
    for rep in range(2):  # the second pass continues from the written-back RNG states in the progressive case
        if idx is None:
            a.trace(); b.trace()
        else:
            a.trace(recompute_indices=idx, n_recompute=idx.numel()); b.trace(recompute_indices=idx, n_recompute=idx.numel())
        assert np.array_equal(bits(_n(a.photons)), bits(_n(b.photons)))
        assert np.array_equal(_n(a.rng, np.uint32), _n(b.rng, np.uint32))
    assert np.abs(_n(a.photons)[:, 3]).sum() > 0
    if case == "directional":  # two lights taking turns on one context: the per-light direction hint follows
        other = dict(light_travel_direction=(-0.2, 0.1, 1.0))
        c = P.PhotonFrame(ctx, vol, tf, (112, 96), (32, 32, 32), emit_in_tracer=True, **other)
        c2 = P.PhotonFrame(ctx, vol, tf, (112, 96), (32, 32, 32), **other)
        c2.trace()
        for _ in range(2):
            c.trace(); a.trace()
            assert np.array_equal(bits(_n(c.photons)), bits(_n(c2.photons)))
            assert np.array_equal(bits(_n(a.photons)), bits(_n(b.photons)))
    if case == "progressive":
        assert not np.array_equal(_n(a.rng, np.uint32), _n(a.rng_initial, np.uint32))


def test_emitted_trace_refusals(ctx, cpm):
    S, P, B = cpm.synthetic, cpm.pipeline, cpm.binding
    f = P.PhotonFrame(ctx, S.heterogeneous_volume(32), S.workspace_tf(), 64, (16, 16, 16), emit_in_tracer=True)
    e = f.emitter
    bad = B.directional_emitter(8, 8, (1, 1, 1), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0), 1.0)   # lattice smaller than the trace
    with pytest.raises(B.CpmError, match="exceeds the lattice"):
        ctx.trace_emitted(f.vol, f.tf, f.aabb, f.params, bad, f.rng, f.photons)
    late = B.directional_emitter(64, 64, (1, 1, 1), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0), 1.0, first_sample=1)
    with pytest.raises(B.CpmError, match="exceeds the lattice"):
        ctx.trace_emitted(f.vol, f.tf, f.aabb, f.params, late, f.rng, f.photons)
    e2 = B.EmitterDesc.from_buffer_copy(e)
    e2.kind = 7
    with pytest.raises(B.CpmError, match="emitter kind"):
        ctx.trace_emitted(f.vol, f.tf, f.aabb, f.params, e2, f.rng, f.photons)
    f.trace()  # the frame's own emitter still works after the refusals


@pytest.mark.parametrize("shape", [(19, 33, 50), (24, 40, 64), (1, 21, 48), (17, 1, 32), (2, 2, 2)])  # [z, y, x]
@pytest.mark.parametrize("how", ["host update", "device update", "device update, unaligned source", "mix", "written in place"])
def test_trace_reads_what_every_volume_writer_left(ctx, oracle, cpm, shape, how):
    """The tracer samples the volume's footprint copy (one fetch = the 2 x 2 x 2 neighbourhood); every writer of a volume --
    create, update from the host or from the device (one fused launch, or copy + re-layout for a source that is not
    4-byte aligned) -- rebuilds it, volume_mix leaves that to the trace that follows, for row lengths on and off the 16-byte path and for one-row / one-slice
    volumes (the y + 1 / z + 1 clamps)."""
    torch = ctx.torch
    rng = np.random.default_rng(sum(shape) + len(how))
    first = rng.integers(0, 256, shape, dtype=np.uint8)
    vol = rng.integers(0, 256, shape, dtype=np.uint8)
    tf = cpm.synthetic.tf_from_points([(0, 1, 1, 1, 0.05), (1, 1, 1, 1, 0.7)], width=64)

    def make(volume, desc):
        h = ctx.volume_create(first, desc)
        if how == "host update":
            h.update(volume)
        elif how == "device update":
            h.update(torch.from_numpy(volume).to(ctx.device))
        elif how == "device update, unaligned source":
            flat = torch.zeros(volume.size + 1, dtype=torch.uint8, device=ctx.device)
            flat[1:] = torch.from_numpy(volume.reshape(-1)).to(ctx.device)
            h.update(flat[1:])
        elif how == "written in place":  # a device producer writes the block itself, then hands the same pointer to update
            size = C.c_size_t()
            ctx.lib.cpm_volume_device_data.restype = C.c_void_p
            ptr = ctx.lib.cpm_volume_device_data(h.h, C.byref(size))
            assert size.value == volume.size
            src = torch.from_numpy(volume.reshape(-1)).to(ctx.device)
            torch.cuda.synchronize()
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip.hipMemcpy(ptr, src.data_ptr(), volume.size, 3) == 0   # device -> device, behind the library's back
            ctx._check(ctx.lib.cpm_volume_update(ctx.h, h.h, C.c_void_p(ptr), 1, ctx._stream()))   # re-derives the tracer's copy, no copy
        else:  # weight 1 -> the mixed volume is `volume` exactly (test_volume_mix); written by the mix kernel
            other = ctx.volume_create(volume, desc)
            ctx.volume_mix(ctx.volume_create(first, desc), other, 1.0, h)
        assert np.array_equal(h.download(), volume)
        return h

    got, want, *_ = _trace_case(ctx, oracle, cpm, vol, tf, 64, (0.2, -0.4, 1.0), make_volume=make)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("shading,g", [(1, 0.0), (0, 0.6), (0, -0.4)])
def test_trace_multiple_scattering(ctx, oracle, cpm, shading, g):
    S = cpm.synthetic
    flagsets = [0, cpm.binding.CPM_TRACE_PROGRESSIVE]
    for flags in flagsets:
        got, want, rng_g, rng_w, *_ = _trace_case(ctx, oracle, cpm, S.heterogeneous_volume(64), S.workspace_tf(), 128,
                                                  (0.3, 0.5, -1.0), max_inter=4, flags=flags, shading=shading, g=g,
                                                  tfs=S.workspace_tf(moved_point4=0.26))
        assert np.array_equal(bits(got), bits(want))
        assert np.array_equal(rng_g, rng_w)


def test_trace_no_single_scattering(ctx, oracle, cpm):
    S = cpm.synthetic
    got, want, *_ = _trace_case(ctx, oracle, cpm, S.heterogeneous_volume(64), S.workspace_tf(), 96, (0.3, 0.5, -1.0),
                                max_inter=3, flags=cpm.binding.CPM_TRACE_NO_SINGLE_SCATTERING, shading=0, g=0.3)
    assert np.array_equal(bits(got), bits(want))


def test_trace_recompute_indices_and_offsets(ctx, oracle, cpm):
    """-D PHOTON_RECOMPUTATION variant with two 'lights' sharing one photon buffer."""
    from oracle_binding import OTraceParams
    S, B = cpm.synthetic, cpm.binding
    n_side = 64
    n = n_side * n_side
    total = 2 * n
    vol_np, tf = S.heterogeneous_volume(64), S.workspace_tf()
    aabb = S.UNIT_CUBE_AABB
    st = np.zeros((total, 2), np.uint32)
    st[:, 0] = oracle.glibc_rand_sequence(0, total)
    oracle.seed_streams(st, 1 << 40)
    s = oracle.uniform_samples_2d(n_side, n_side)
    lights = []
    for direction in ((0.3, 0.5, -1.0), (0, 0, 1)):
        d, o, u, v, area = _light_setup(cpm, direction)
        ls = oracle.directional_light_samples(s, (1, 1, 1), d, o, u, v, area)
        lights.append((ls, oracle.light_sample_box_intersection(ls, aabb)))
    rng = np.random.default_rng(1)
    idx = np.sort(rng.choice(total, 3000, replace=False)).astype(np.uint32)
    vol, tfh = ctx.volume_create(vol_np), ctx.tf_create(tf)
    ovol = oracle.volume(vol_np)
    photons_d = ctx.torch.full((total, 8), -1.0, dtype=ctx.torch.float32, device=ctx.device)
    photons_o = np.full((total, 8), -1.0, np.float32)
    rng_d, rng_o = _t(ctx, st), st.copy()
    idx_d = _t(ctx, idx)
    for k, (ls, isect) in enumerate(lights):
        p, po = B.TraceParams(), OTraceParams()
        for q in (p, po):
            q.step_size = 1 / 64
            q.photon_offset = k * n
            q.n_light_samples = n
            q.max_interactions = 1
            q.total_photons = total
        ctx.trace(vol, tfh, aabb, p, _t(ctx, ls), _t(ctx, isect), rng_d, photons_d, recompute_indices=idx_d, n_recompute=idx.size)
        oracle.trace(ovol, tf, aabb, po, ls, isect, rng_o, photons_o, recompute_indices=idx, n_recompute=idx.size)
    got = _n(photons_d)
    assert np.array_equal(bits(got), bits(photons_o))
    untouched = np.setdiff1d(np.arange(total), idx)
    assert (got[untouched] == -1).all()


def test_trace_empty_and_errors(ctx, cpm):
    S, B = cpm.synthetic, cpm.binding
    vol, tfh = ctx.volume_create(S.homogeneous_volume(8)), ctx.tf_create(S.homogeneous_tf())
    p = B.TraceParams()
    p.max_interactions = 1
    ctx.trace(vol, tfh, S.UNIT_CUBE_AABB, p, None, None, None, None)  # zero photons: no-op
    p.max_interactions = 0
    with pytest.raises(B.CpmError):
        ctx.trace(vol, tfh, S.UNIT_CUBE_AABB, p, None, None, None, None)
    p.max_interactions = 1
    p.n_light_samples = 4
    p.total_photons = 2
    with pytest.raises(B.CpmError):
        ctx.trace(vol, tfh, S.UNIT_CUBE_AABB, p, None, None, None, None)


# ----------------------------------------------------------------------------- sort

@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 4097, 100_003, 1 << 20, (1 << 21) + 17])
def test_sort_pairs_stable(ctx, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 21, n, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    kd, vd = _t(ctx, keys), _t(ctx, vals)
    ctx.sort_pairs(kd, vd, 21)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(_n(kd, np.uint32), keys[order])
    assert np.array_equal(_n(vd, np.uint32), vals[order])


@pytest.mark.parametrize("bits_", [0, 1, 8, 9, 16, 31, 32])
def test_sort_key_bits(ctx, bits_):
    rng = np.random.default_rng(bits_)
    n = 50_000
    hi = 1 << (bits_ if bits_ else 32)
    keys = rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)
    if n:
        keys[::97] = hi - 1
        keys[::101] = 0
    vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    kd, vd = _t(ctx, keys), _t(ctx, vals)
    ctx.sort_pairs(kd, vd, bits_)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(_n(kd, np.uint32), keys[order])
    assert np.array_equal(_n(vd, np.uint32), vals[order])
    k2 = _t(ctx, keys)
    ctx.sort_keys(k2, bits_)
    assert np.array_equal(_n(k2, np.uint32), np.sort(keys))


def test_sort_skewed_keys(ctx):
    n = 300_000
    keys = np.zeros(n, np.uint32)
    keys[n // 2:] = 0x7FFFFFFF  # the importance buffer right after a reset
    keys[::5] = 5
    vals = np.arange(n, dtype=np.uint32)
    kd, vd = _t(ctx, keys), _t(ctx, vals)
    ctx.sort_pairs(kd, vd, 31)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(_n(vd, np.uint32), vals[order])


# ----------------------------------------------------------------------------- splat / bin / gather

def _random_photons(rng, n, sentinel_every=0, spread=1.0, rgb=False):
    ph = np.zeros((n, 8), np.float32)
    ph[:, :3] = (0.5 + (rng.random((n, 3)) - 0.5) * spread).astype(np.float32)
    ph[:, 3:6] = rng.random((n, 3)).astype(np.float32) * 3
    ph[:, 6:] = rng.random((n, 2)).astype(np.float32)
    if sentinel_every:
        ph[::sentinel_every, :3] = np.float32(3.402823466e+38)
    return ph


@pytest.mark.parametrize("dims,channels,radius_vox,force_voxel", [
    ((32, 32, 32), 1, 0.866, 0),       # record-major kernel specialised for r < 1 voxel (BASELINE geometry)
    ((32, 32, 32), 1, 0.866, 2),       # generic record-major kernel on the same input
    ((32, 32, 32), 1, 0.866, 1),       # voxel-major kernel on the same input
    ((30, 21, 9), 1, 0.7, 0),          # bricks cut by the grid edge
    ((24, 40, 16), 4, 1.7, 0),     # 4 x float32, 4 candidates per axis: voxel-major
    ((24, 40, 16), 4, 0.8, 0),     # 4 x float32 through the tuned record-major kernels (2 and 3 candidates per axis)
    ((22, 19, 33), 4, 1.3, 0),
    ((16, 16, 16), 1, 0.3, 0),
    ((20, 28, 36), 1, 1.2, 0),     # record-major kernel, 3 candidates per axis, bricks cut by the grid edge
    ((32, 32, 32), 1, 2.2, 0),     # radius too large for the record-major kernel: falls back
    ((32, 32, 32), 1, 1.7320508, 0),   # 1 voxel of a same-size volume (|(r, r, r)| = sqrt 3 cells): 4 candidates per axis, halo of 2
    ((26, 30, 21), 1, 1.9, 0),
])
def test_bin_and_gather_bit_exact(ctx, oracle, cpm, dims, channels, radius_vox, force_voxel):
    ctx.lib.cpm_debug_force_voxel_gather(ctx.h, int(force_voxel))
    try:
        _bin_and_gather_case(ctx, oracle, cpm, dims, channels, radius_vox)
    finally:
        ctx.lib.cpm_debug_force_voxel_gather(ctx.h, 0)


@pytest.mark.parametrize("dims,channels", [((32, 32, 32), 1), ((24, 40, 16), 4)])
def test_bin_with_separate_finalize_launch(ctx, oracle, cpm, dims, channels):
    """cpm_bin's default lets the last radix pass write order / records / run starts; the separate
    bin_finalize_kernel (used for n == 1 and the onesweep test mode) produces the same bin."""
    ctx.lib.cpm_debug_set_bin_fused(ctx.h, 0)
    try:
        _bin_and_gather_case(ctx, oracle, cpm, dims, channels, 0.866)
    finally:
        ctx.lib.cpm_debug_set_bin_fused(ctx.h, 1)


def _bin_and_gather_case(ctx, oracle, cpm, dims, channels, radius_vox):
    rng = np.random.default_rng(sum(dims))
    n = 20_000
    ph = _random_photons(rng, n, sentinel_every=13, spread=1.1)  # some photons outside [0,1]^3 and on the faces
    ph[5, :3] = (1.0, 1.0, 1.0)
    ph[6, :3] = (0.0, 0.0, 0.0)
    radius = float(np.float32(radius_vox / max(dims)))
    scale = oracle.relative_irradiance_scale(radius, n)
    assert scale == cpm.binding.relative_irradiance_scale(radius, n)
    ph[7:400, :3] = ph[7, :3] + (rng.random((393, 3), dtype=np.float32) - 0.5) * np.float32(0.5 / max(dims))  # a dense cluster
    g = cpm.binding.default_grid_desc(dims, channels)
    og = oracle.grid(dims, channels)
    assert list(g.index_to_texture) == list(og.index_to_texture)
    cells = dims[0] * dims[1] * dims[2]
    torch = ctx.torch
    order = torch.empty(n, dtype=torch.int32, device=ctx.device)
    cs = torch.empty(cells + 1, dtype=torch.int32, device=ctx.device)
    srt = torch.empty((n, 4 if channels == 1 else 8), dtype=torch.float32, device=ctx.device)
    ctx.bin(_t(ctx, ph), n, g, order, cs, srt)
    o_order, o_cs, o_srt = oracle.bin(ph, n, og)
    assert np.array_equal(_n(order, np.uint32), o_order)
    assert np.array_equal(_n(cs, np.uint32), o_cs)
    assert np.array_equal(bits(_n(srt)), bits(o_srt))

    shape = (cells,) if channels == 1 else (cells, 4)
    out = torch.full(shape, 7.0, dtype=torch.float32, device=ctx.device)
    ctx.gather(srt, cs, n, g, radius, scale, out)
    want = np.full(shape, 7.0, np.float32)
    oracle.gather(o_srt, o_cs, n, og, radius, scale, want)
    assert np.array_equal(bits(_n(out)), bits(want))
    # accumulate variant
    ctx.gather(srt, cs, n, g, radius, scale, out, accumulate=True)
    oracle.gather(o_srt, o_cs, n, og, radius, scale, want, accumulate=True)
    assert np.array_equal(bits(_n(out)), bits(want))

    # the gather reproduces the reference splat's sums (different summation order only)
    sp = np.zeros(shape, np.float32)
    oracle.splat(ph, n, og, radius, scale, sp)
    first = np.zeros(shape, np.float32)
    oracle.gather(o_srt, o_cs, n, og, radius, scale, first)
    tol = 1e-5 * max(1.0, float(np.abs(sp).max()))
    np.testing.assert_allclose(first, sp, rtol=2e-5, atol=tol)

    # HIP atomic splat vs oracle splat: fp32 sums in a different order
    spd = torch.zeros(shape, dtype=torch.float32, device=ctx.device)
    ctx.splat(_t(ctx, ph), n, g, radius, scale, spd)
    np.testing.assert_allclose(_n(spd), sp, rtol=2e-5, atol=tol)


@pytest.mark.parametrize("dims,radius_vox", [((32, 32, 32), 0.866), ((30, 21, 9), 0.7), ((16, 16, 16), 0.3), ((40, 12, 52), 0.8),
                                             ((20, 28, 36), 1.2), ((33, 17, 26), 1.0), ((24, 24, 24), 1.45),   # 3 candidates per axis
                                             ((24, 24, 24), 1.8), ((19, 22, 35), 1.7320508)])                     # 4, halo of 2 cells
def test_gather_one_wave_per_brick_kernel(ctx, oracle, cpm, dims, radius_vox):
    """The r < 1 voxel gather has two kernels with the same summation order: the cooperative one (default: a
    workgroup's four waves share four bricks, drains take turns) and one wave per brick."""
    for mode in (0, 2, 8, 4):
        ctx.lib.cpm_debug_set_gather_coop(ctx.h, mode)
        try:
            _bin_and_gather_case(ctx, oracle, cpm, dims, 1, radius_vox)
            if mode in (0, 4) and radius_vox < 1.5:
                _bin_and_gather_case(ctx, oracle, cpm, dims, 4, radius_vox)   # 4 x float32 light volume
        finally:
            ctx.lib.cpm_debug_set_gather_coop(ctx.h, 1)


@pytest.mark.parametrize("mode", [1, 0])
def test_gather_dense_clusters(ctx, oracle, cpm, mode):
    """6000 photons in one cell (more records around one brick than the 4096-record row-start bitmask holds: the
    binary-search row lookup), 600 at one identical position in a corner row, repeated launches."""
    dims, n = (24, 20, 28), 30_000
    rng = np.random.default_rng(5)
    ph = _random_photons(rng, n, sentinel_every=17, spread=1.05)
    ph[100:6100, :3] = np.float32(0.52) + (rng.random((6000, 3), dtype=np.float32) - 0.5) * np.float32(0.9 / max(dims))
    ph[7000:7600, :3] = (np.float32(0.999), np.float32(0.001), np.float32(0.5))
    radius = float(np.float32(0.8 / max(dims)))
    scale = oracle.relative_irradiance_scale(radius, n)
    g = cpm.binding.default_grid_desc(dims, 1)
    og = oracle.grid(dims, 1)
    cells = dims[0] * dims[1] * dims[2]
    torch = ctx.torch
    order = torch.empty(n, dtype=torch.int32, device=ctx.device)
    cs = torch.empty(cells + 1, dtype=torch.int32, device=ctx.device)
    srt = torch.empty((n, 4), dtype=torch.float32, device=ctx.device)
    ctx.bin(_t(ctx, ph), n, g, order, cs, srt)
    o_order, o_cs, o_srt = oracle.bin(ph, n, og)
    assert np.array_equal(_n(order, np.uint32), o_order) and np.array_equal(_n(cs, np.uint32), o_cs)
    want = np.zeros(cells, np.float32)
    oracle.gather(o_srt, o_cs, n, og, radius, scale, want)
    ctx.lib.cpm_debug_set_gather_coop(ctx.h, mode)
    try:
        for rep in range(3):
            out = torch.full((cells,), -3.0, dtype=torch.float32, device=ctx.device)
            ctx.gather(srt, cs, n, g, radius, scale, out)
            assert np.array_equal(bits(_n(out)), bits(want)), rep
    finally:
        ctx.lib.cpm_debug_set_gather_coop(ctx.h, 1)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 257, 2047, 2048, 2049, 32768, 32769, 70_001])
@pytest.mark.parametrize("channels", [1, 4])
def test_bin_sizes_around_tile_boundaries(ctx, oracle, cpm, n, channels):
    """cpm_bin at sizes around the wave / workgroup / sort-tile boundaries (the key kernel counts the first digit per
    sort tile, the last radix pass writes order / records / run starts): order, cell starts and records equal the
    oracle's, and a gather over them equals the oracle's gather."""
    dims = (12, 10, 14)
    rng = np.random.default_rng(n * 7 + channels)
    ph = _random_photons(rng, n, sentinel_every=5 if n > 4 else 0, spread=1.1)
    g, og = cpm.binding.default_grid_desc(dims, channels), oracle.grid(dims, channels)
    cells = dims[0] * dims[1] * dims[2]
    torch = ctx.torch
    order = torch.full((n,), -1, dtype=torch.int32, device=ctx.device)
    cs = torch.full((cells + 1,), -1, dtype=torch.int32, device=ctx.device)
    srt = torch.full((n, 4 if channels == 1 else 8), -1.0, dtype=torch.float32, device=ctx.device)
    ctx.bin(_t(ctx, ph), n, g, order, cs, srt)
    o_order, o_cs, o_srt = oracle.bin(ph, n, og)
    assert np.array_equal(_n(order, np.uint32), o_order)
    assert np.array_equal(_n(cs, np.uint32), o_cs)
    assert np.array_equal(bits(_n(srt)), bits(o_srt))
    radius = float(np.float32(0.8 / max(dims)))
    shape = (cells,) if channels == 1 else (cells, 4)
    out = torch.full(shape, 5.0, dtype=torch.float32, device=ctx.device)
    want = np.full(shape, 5.0, np.float32)
    ctx.gather(srt, cs, n, g, radius, 1.0, out)
    oracle.gather(o_srt, o_cs, n, og, radius, 1.0, want)
    assert np.array_equal(bits(_n(out)), bits(want))


def test_bin_empty_and_all_sentinel(ctx, oracle, cpm):
    dims = (8, 8, 8)
    g, og = cpm.binding.default_grid_desc(dims, 1), oracle.grid(dims, 1)
    torch = ctx.torch
    cs = torch.full((513,), -1, dtype=torch.int32, device=ctx.device)
    ctx.bin(None, 0, g, None, cs, None)
    assert (_n(cs) == 0).all()
    out = torch.full((512,), 3.0, dtype=torch.float32, device=ctx.device)
    ctx.gather(None, cs, 0, g, 0.1, 1.0, out)
    assert (_n(out) == 0).all()
    n = 1000
    ph = np.full((n, 8), 3.402823466e+38, np.float32)
    order = torch.empty(n, dtype=torch.int32, device=ctx.device)
    srt = torch.empty((n, 4), dtype=torch.float32, device=ctx.device)
    ctx.bin(_t(ctx, ph), n, g, order, cs, srt)
    assert (_n(cs) == 0).all()
    assert np.array_equal(_n(order, np.uint32), np.arange(n, dtype=np.uint32))


def test_splat_selected_and_copy_indexed(ctx, oracle, cpm):
    rng = np.random.default_rng(11)
    n, I = 5000, 3
    dims = (24, 24, 24)
    ph = _random_photons(rng, n * I, sentinel_every=9)
    idx = np.sort(rng.choice(n, 700, replace=False)).astype(np.uint32)
    radius, scale = 0.05, 2.5
    g, og = cpm.binding.default_grid_desc(dims, 4), oracle.grid(dims, 4)
    torch = ctx.torch
    for mult in (1.0, -1.0):
        out = torch.zeros((24 ** 3, 4), dtype=torch.float32, device=ctx.device)
        ctx.splat_selected(_t(ctx, ph), _t(ctx, idx), idx.size, g, radius, scale, mult, n, I, out)
        want = np.zeros((24 ** 3, 4), np.float32)
        oracle.splat_selected(ph, idx, og, radius, scale, mult, n, I, want)
        np.testing.assert_allclose(_n(out), want, rtol=2e-5, atol=1e-5 * float(np.abs(want).max()))
    al = torch.zeros((idx.size * I + 5, 8), dtype=torch.float32, device=ctx.device)
    ctx.copy_indexed_photons(_t(ctx, ph), _t(ctx, idx), idx.size, -1.0, n, I, al, 5)
    wa = np.zeros((idx.size * I + 5, 8), np.float32)
    oracle.copy_indexed_photons(ph, idx, -1.0, n, I, wa, 5)
    assert np.array_equal(bits(_n(al)), bits(wa))
    # snapshot refresh: only the selected photons (every interaction) move; out-of-range indices are ignored
    snap = rng.random((n * I, 8), dtype=np.float32)
    snap_d = _t(ctx, snap)
    idx_bad = np.concatenate([idx, np.array([n, n + 7, 0xffffffff], np.uint32)])
    ctx.snapshot_selected_photons(_t(ctx, ph), _t(ctx, idx_bad), idx_bad.size, n, I, snap_d)
    want = snap.copy()
    for k in range(I):
        want[idx.astype(np.int64) + k * n] = ph[idx.astype(np.int64) + k * n]
    assert np.array_equal(bits(_n(snap_d)), bits(want))
    ctx.snapshot_selected_photons(_t(ctx, ph), _t(ctx, idx), 0, n, I, snap_d)      # nothing selected: no-op
    assert np.array_equal(bits(_n(snap_d)), bits(want))


# ----------------------------------------------------------------------------- end to end

def test_frame_config1_end_to_end(ctx, oracle, cpm):
    """BASELINE config 1 (64^3 homogeneous, 65 536 photons, 32^3 grid): trace -> bin -> gather
    on the GPU equals the oracle bit for bit, and the oracle's reference-formulation splat
    within tolerance."""
    from oracle_binding import OTraceParams
    S, P = cpm.synthetic, cpm.pipeline
    vol_np, tf = S.homogeneous_volume(64), S.homogeneous_tf(0.25)
    fr = P.PhotonFrame(ctx, vol_np, tf, 256, (32, 32, 32))
    lv = _n(fr.frame())
    n = fr.n
    ls, isect = _n(fr.light_samples), _n(fr.isect)
    rng_o = _n(fr.rng_initial, np.uint32).copy()
    po = OTraceParams()
    po.step_size = 1 / 64
    po.n_light_samples = n
    po.max_interactions = 1
    po.total_photons = n
    photons_o = np.zeros((n, 8), np.float32)
    oracle.trace(oracle.volume(vol_np), tf, S.UNIT_CUBE_AABB, po, ls, isect, rng_o, photons_o)
    assert np.array_equal(bits(_n(fr.photons)), bits(photons_o))
    og = oracle.grid((32, 32, 32), 1)
    o_order, o_cs, o_srt = oracle.bin(photons_o, n, og)
    want = np.zeros(32 ** 3, np.float32)
    oracle.gather(o_srt, o_cs, n, og, fr.radius, fr.scale, want)
    assert np.array_equal(bits(lv), bits(want))
    sp = np.zeros(32 ** 3, np.float32)
    oracle.splat(photons_o, n, og, fr.radius, fr.scale, sp)
    np.testing.assert_allclose(lv, sp, rtol=2e-5, atol=1e-5 * float(sp.max()))
    assert lv.sum() > 0


def test_frame_config2_properties(ctx, oracle, cpm):
    """BASELINE config 2 at full size (256^3, 1 048 576 photons, 128^3 grid): size-independent
    properties + a sampled photon-for-photon check against the oracle."""
    from oracle_binding import OTraceParams
    S, P = cpm.synthetic, cpm.pipeline
    vol_np, tf = S.heterogeneous_volume(256), S.workspace_tf()
    fr = P.PhotonFrame(ctx, vol_np, tf, 1024, (128, 128, 128), light_travel_direction=(0.3, 0.5, -1.0))
    lv1 = fr.frame().clone()
    lv2 = fr.frame().clone()
    assert ctx.torch.equal(lv1, lv2)  # deterministic: no atomics on the path, RNG not advanced
    photons = _n(fr.photons)
    order = _n(fr.order, np.uint32)
    cs = _n(fr.cell_start, np.uint32)
    n = fr.n
    valid = photons[:, 0] != np.float32(3.402823466e+38)
    n_valid = int(valid.sum())
    # bin: a permutation, sentinels last, keys non-decreasing, cell_start = counts
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
    assert cs[0] == 0 and cs[-1] == n_valid and (np.diff(cs.astype(np.int64)) >= 0).all()
    cell = np.clip(np.floor(photons[order[:n_valid], :3] * np.float32(128)), 0, 127).astype(np.int64)
    key = cell[:, 0] + 128 * (cell[:, 1] + 128 * cell[:, 2])
    assert (np.diff(key) >= 0).all()
    assert np.array_equal(np.bincount(key, minlength=128 ** 3), np.diff(cs.astype(np.int64)))
    same = np.diff(key) == 0
    assert (np.diff(order[:n_valid].astype(np.int64))[same] > 0).all()  # stable within a cell
    # gather == reference splat (atomic, order-dependent) within fp32 tolerance, at full size
    sp = _n(fr.splat(ctx.torch.zeros_like(lv1)))
    lv = _n(lv1)
    np.testing.assert_allclose(lv, sp, rtol=1e-4, atol=1e-5 * float(sp.max()))
    # sampled photon-level parity with the oracle: every 97th photon
    sel = np.arange(0, n, 97)
    ls, isect = _n(fr.light_samples)[sel].copy(), _n(fr.isect)[sel].copy()
    rng_o = _n(fr.rng_initial, np.uint32)[sel].copy()
    po = OTraceParams()
    po.step_size = 1 / 256
    po.n_light_samples = sel.size
    po.max_interactions = 1
    po.total_photons = sel.size
    ph_o = np.zeros((sel.size, 8), np.float32)
    oracle.trace(oracle.volume(vol_np), tf, S.UNIT_CUBE_AABB, po, ls, isect, rng_o, ph_o)
    assert np.array_equal(bits(photons[sel]), bits(ph_o))


def test_frame_graph_replay_equals_eager(ctx, cpm):
    """The captured HIP graph of the frame produces exactly what the eager launches produce."""
    S, P = cpm.synthetic, cpm.pipeline
    fr = P.PhotonFrame(ctx, S.heterogeneous_volume(64), S.workspace_tf(), 256, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0))
    eager = fr.frame().clone()
    photons = fr.photons.clone()
    fr.capture()
    fr.light_volume.zero_()
    fr.photons.zero_()
    for _ in range(3):
        fr.replay()
    ctx.torch.cuda.synchronize()
    assert ctx.torch.equal(fr.light_volume, eager) and ctx.torch.equal(fr.photons, photons)


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("n,bits_", [(5, 32), (1023, 8), (1024, 9), (70_001, 22), (1 << 20, 22), (3_000_017, 31), (1 << 20, 32)])
def test_sort_pass_structures(ctx, mode, n, bits_):
    """Both pass structures -- onesweep (ticketed tiles + decoupled look-back, one launch per pass) and
    hist + rowscan + scatter -- give the stable ascending order, including skewed digit distributions."""
    rng = np.random.default_rng(n + bits_)
    keys = rng.integers(0, 1 << bits_, n, dtype=np.uint64).astype(np.uint32)
    keys[: n // 3] &= np.uint32(0xFF00FF)            # few distinct digits in some passes
    keys[n // 2:] = np.sort(keys[n // 2:])           # long presorted run: whole tiles with one digit
    vals = np.arange(n, dtype=np.uint32)
    ctx.lib.cpm_debug_set_sort_mode(ctx.h, mode)
    try:
        for rep in range(3):                         # repeated: scratch reuse, look-back state reset
            kd, vd = _t(ctx, keys), _t(ctx, vals)
            ctx.sort_pairs(kd, vd, bits_)
            order = np.argsort(keys & np.uint32((1 << bits_) - 1 if bits_ < 32 else 0xFFFFFFFF), kind="stable")
            assert np.array_equal(_n(vd, np.uint32), vals[order])
            assert np.array_equal(_n(kd, np.uint32), keys[order])
    finally:
        ctx.lib.cpm_debug_set_sort_mode(ctx.h, 0)


def test_non_default_grid_matrices_are_refused_by_the_cell_sorted_path(ctx, cpm):
    """cpm_bin / cpm_gather are written for Inviwo's own light-volume matrices; a scaled or offset light volume is refused
    (CPM_ERR_UNSUPPORTED) instead of being binned against the wrong cells.  cpm_splat and the tolerance-mode
    pair honour the matrices (tests/test_fast_gpu.py::test_fast_non_default_grid_matrices)."""
    B = cpm.binding
    grid = B.default_grid_desc((16, 16, 16), 1)
    grid.texture_to_index[0] = 20.0
    t = ctx.torch
    ph = t.zeros((8, 8), dtype=t.float32, device=ctx.device)
    order = t.zeros(8, dtype=t.int32, device=ctx.device)
    cs = t.zeros(16 ** 3 + 1, dtype=t.int32, device=ctx.device)
    srt = t.zeros((8, 4), dtype=t.float32, device=ctx.device)
    out = t.zeros(16 ** 3, dtype=t.float32, device=ctx.device)
    with pytest.raises(B.CpmError) as e:
        ctx.bin(ph, 8, grid, order, cs, srt)
    assert e.value.status == -4
    with pytest.raises(B.CpmError) as e:
        ctx.gather(srt, cs, 8, grid, 0.05, 1.0, out)
    assert e.value.status == -4
    ctx.splat(ph, 8, grid, 0.05, 1.0, out)            # the reference formulation reads the matrices


def test_gather_and_splat_at_several_interactions(ctx, oracle, cpm):
    """I = 3: bin + gather cover all N x I records (documented), the reference's full splat interaction 0 only;
    splat(all_interactions=True) is the gather's counterpart."""
    S, P = cpm.synthetic, cpm.pipeline
    fr = P.PhotonFrame(ctx, S.heterogeneous_volume(64), S.workspace_tf(), 160, (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0),
                       max_interactions=3, material=(0.4, 0.0, 0.0, 0.0))
    lv = _n(fr.frame()).copy()
    lvf = _n(fr.frame_fast()).copy()
    sp_all = _n(fr.splat(ctx.torch.zeros_like(fr.light_volume), all_interactions=True))
    np.testing.assert_allclose(lv, sp_all, rtol=1e-4, atol=1e-5 * float(sp_all.max()))
    np.testing.assert_allclose(lvf, sp_all, rtol=1e-4, atol=1e-5 * float(sp_all.max()))
    sp0 = _n(fr.splat(ctx.torch.zeros_like(fr.light_volume)))                    # interaction 0 only
    og = oracle.grid((32, 32, 32), 1)
    ph = _n(fr.photons)
    want0 = np.zeros(32 ** 3, np.float32)
    oracle.splat(ph, fr.n, og, fr.radius, fr.scale, want0)
    np.testing.assert_allclose(sp0, want0, rtol=1e-4, atol=1e-5 * float(want0.max()))
    assert sp_all.sum() > sp0.sum() > 0                                          # later interactions were stored and add light
