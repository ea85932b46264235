"""The C++ host layer (host/: Processor/Port mirror over the C-ABI), driven through its C facade."""
import ctypes as C

import numpy as np
import pytest

from test_parity_gpu import _n, bits

pytestmark = pytest.mark.gpu

CUBE_QUADS = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]

# what the CorrelatedPhotonMappingSingleVolume workspace binds (ref tracercl.cpp:61-99,
# ...processorcl.cpp:45-108, directionallightsamplerclprocessor.cpp:38-64, ...)
SURFACE = {
    "org.inviwo.ProgressivePhotonTracerCL": (
        {"volume", "recomputationImportance", "LightSamples"}, {"photons", "recomputedIndices"},
        {"samplingRate", "radius", "radianceScale", "maxIncrementalPhotonsToUpdate", "equalImportance", "spatialSorting",
         "maxScatteringEvents", "noSingleScattering", "alpha", "wgsize", "glsharing", "enableRefinement",
         "enableProgressiveRecomputation", "clipX", "clipY", "clipZ"}),
    "org.inviwo.PhotonToLightVolumeProcessorCL": (
        {"volume", "photons", "recomputedPhotonIndices"}, {"lightvolume"},
        {"incrementalRecomputationThreshold", "volumeSizeOption", "volumeDataType", "alignChangedPhotons", "wgsize", "glsharing"}),
    "org.inviwo.DirectionalLightSamplerCL": ({"SceneGeometry", "samples", "light"}, {"LightSamples"}, set()),
    "org.inviwo.UniformSampleGenerator2DCL": (set(), {"samples"}, {"nSamples"}),
    "org.inviwo.MinMaxUniformGrid3DImportanceCLProcessor": ({"minMaxUniformGrid3D", "volumeDifferenceInfo"}, {"importanceUniformGrid3D"},
                                                            {"incrementalImportance", "useAssociatedColor", "TFPointEpsilon"}),
    "org.inviwo.VolumeMinMaxCLProcessor": ({"volume", "VolumeSequenceInput"}, {"output", "UniformGrid3DVectorOut"}, {"region"}),
}


@pytest.fixture(scope="module")
def host(cpm, ctx):
    # ctx first: torch must bring up ITS HIP runtime before another library initialises one
    # (two libamdhip64 copies in one process do not both see the GPU)
    cpm.build.build_host_library()
    lib = C.CDLL(str(cpm.binding.LIB_PATH.parent / "libcpm_host.so"))
    lib.cpmh_create.restype = C.c_void_p
    lib.cpmh_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float * 3),
                                C.POINTER(C.c_float * 3), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    for name, res, args in [("cpmh_destroy", None, [C.c_void_p]), ("cpmh_evaluate", C.c_int, [C.c_void_p, C.c_int]),
                            ("cpmh_set_transfer_function", None, [C.c_void_p, C.c_void_p, C.c_int]),
                            ("cpmh_set_property_float", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_float]),
                            ("cpmh_set_property_string", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]),
                            ("cpmh_light_volume_dims", None, [C.c_void_p, C.POINTER(C.c_int * 3), C.POINTER(C.c_int)]),
                            ("cpmh_download_light_volume", C.c_int, [C.c_void_p, C.c_void_p]),
                            ("cpmh_n_photons", C.c_int, [C.c_void_p]), ("cpmh_download_photons", C.c_int, [C.c_void_p, C.c_void_p]),
                            ("cpmh_n_recomputed", C.c_int, [C.c_void_p]), ("cpmh_remaining", C.c_int, [C.c_void_p]),
                            ("cpmh_last_light_volume_path", C.c_char_p, [C.c_void_p]), ("cpmh_radius", C.c_double, [C.c_void_p]),
                            ("cpmh_light_plane", None, [C.c_void_p, C.POINTER(C.c_float * 10)]),
                            ("cpmh_light_direction", None, [C.c_void_p, C.POINTER(C.c_float * 3)]),
                            ("cpmh_tf_lut", None, [C.c_void_p, C.c_void_p]),
                            ("cpmh_enable_refinement", C.c_int, [C.c_void_p, C.c_int]), ("cpmh_refine", C.c_int, [C.c_void_p]),
                            ("cpmh_enable_shard_reduce", C.c_int, [C.c_void_p]), ("cpmh_last_reduce", C.c_char_p, [C.c_void_p]),
                            ("cpmh_share_light_volume_gl", None, [C.c_void_p, C.c_uint, C.c_int]), ("cpmh_last_gl_copy", C.c_char_p, [C.c_void_p]),
                            ("cpmh_describe_surface", C.c_char_p, [C.c_void_p])]:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


@pytest.fixture(scope="module")
def host_extras(cpm, ctx):
    """libcpm_host_extras.so: the same layer built with -DCPM_HOST_EXTRAS -- the processors outside the workspace's path
    (RadixSortCL node, UniformGrid3D export / selector / vector source); the product library does not carry them."""
    cpm.build.build_host_library(extras=True)
    return C.CDLL(str(cpm.binding.LIB_PATH.parent / "libcpm_host_extras.so"))


class Net:
    def __init__(self, lib, vol, n_side, light_pos, light_dir, tf_points, size_option=2, max_scattering=1, correlated=False):
        self.lib = lib
        self.max_scattering = max_scattering
        vol = np.ascontiguousarray(vol)
        pts = np.ascontiguousarray(np.asarray(tf_points, np.float32))
        self.h = lib.cpmh_create(vol.ctypes.data, 0, vol.shape[2], vol.shape[1], vol.shape[0], n_side, n_side,
                                 C.byref((C.c_float * 3)(*light_pos)), C.byref((C.c_float * 3)(*light_dir)), pts.ctypes.data,
                                 pts.shape[0], size_option, max_scattering, int(correlated))
        assert self.h

    def evaluate(self, first=False):
        assert self.lib.cpmh_evaluate(self.h, int(first)) == 0

    def set_tf(self, tf_points):
        pts = np.ascontiguousarray(np.asarray(tf_points, np.float32))
        self.lib.cpmh_set_transfer_function(self.h, pts.ctypes.data, pts.shape[0])

    def light_volume(self):
        dims, ch = (C.c_int * 3)(), C.c_int()
        self.lib.cpmh_light_volume_dims(self.h, C.byref(dims), C.byref(ch))
        out = np.zeros(dims[0] * dims[1] * dims[2] * ch.value, np.float32)
        assert self.lib.cpmh_download_light_volume(self.h, out.ctypes.data) == 0
        return out, tuple(dims), ch.value

    def photons(self):
        out = np.zeros((self.lib.cpmh_n_photons(self.h) * self.max_scattering, 8), np.float32)   # SoA by interaction: I x N records
        assert self.lib.cpmh_download_photons(self.h, out.ctypes.data) == 0
        return out

    def plane(self):
        p = (C.c_float * 10)()
        self.lib.cpmh_light_plane(self.h, C.byref(p))
        a = np.array(list(p), np.float32)
        return a[0:3], a[3:6], a[6:9], float(a[9])

    def direction(self):
        d = (C.c_float * 3)()
        self.lib.cpmh_light_direction(self.h, C.byref(d))
        return np.array(list(d), np.float32)

    def tf_lut(self):
        out = np.zeros((1024, 4), np.float32)
        self.lib.cpmh_tf_lut(self.h, out.ctypes.data)
        return out

    def close(self):
        self.lib.cpmh_destroy(self.h)


def _light(cpm, direction, dist=2.0):
    d = cpm.pipeline._normalize(direction)
    return (np.array([0.5, 0.5, 0.5], np.float32) - np.float32(dist) * d), d


def test_drop_in_surface(host, cpm):
    S = cpm.synthetic
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    net = Net(host, S.homogeneous_volume(16), 16, pos, d, S.WORKSPACE_TF_POINTS, correlated=True)
    text = host.cpmh_describe_surface(net.h).decode()
    seen = {}
    for line in text.strip().splitlines():
        cid, i, o, p = line.split("|")
        seen[cid] = (set(filter(None, i[3:].split(","))), set(filter(None, o[4:].split(","))), set(filter(None, p[5:].split(","))))
    for cid, (ins, outs, props) in SURFACE.items():
        assert cid in seen, cid
        assert ins <= seen[cid][0], (cid, ins - seen[cid][0])
        assert outs <= seen[cid][1], (cid, outs - seen[cid][1])
        assert props <= seen[cid][2], (cid, props - seen[cid][2])
    net.close()


def test_network_matches_abi_pipeline(host, ctx, cpm):
    """The C++ network (workspace wiring, light volume = 1/2 of the input volume) produces the photons and
    the light volume the thin Python driver produces from the same light plane: bit for bit."""
    S, P = cpm.synthetic, cpm.pipeline
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    net = Net(host, vol, 160, pos, d, S.WORKSPACE_TF_POINTS, size_option=2)
    net.evaluate(first=True)
    o, u, v, area = net.plane()
    # the host fit agrees with the Python twin of the reference's CPU code
    po, pu, pv = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, pos, d)
    assert abs(area - float(np.linalg.norm(pu) * np.linalg.norm(pv))) < 1e-5
    idx = np.array([i for q in CUBE_QUADS for i in (q[0], q[1], q[2], q[0], q[2], q[3])], np.int32)
    # the host's LUT (Inviwo: 1024 texels, piecewise linear) equals the Python twin's
    lut = net.tf_lut()
    np.testing.assert_allclose(lut, S.workspace_tf(), rtol=0, atol=1e-7)
    fr = P.PhotonFrame(ctx, vol, lut, 160, (32, 32, 32), light_travel_direction=net.direction(), light_plane=(o, u, v, area),
                       mesh_intersection=(S.UNIT_CUBE_VERTICES, idx))
    # the processor's default formulation is the tolerance-mode one (brick bin + fixed-point tile gather): bit for bit what
    # cpm_bin_fast + cpm_gather_fast give the Python driver
    lvf = _n(fr.frame_fast()).copy()
    hv, dims, ch = net.light_volume()
    assert dims == (32, 32, 32) and ch == 1
    assert abs(host.cpmh_radius(net.h) - fr.radius) < 1e-12
    assert np.array_equal(bits(net.photons()), bits(_n(fr.photons)))
    assert np.array_equal(bits(hv), bits(lvf))
    # "gather": the bit-exact sequential contract
    lv = _n(fr.frame())
    assert host.cpmh_set_property_string(net.h, b"lightvolume", b"formulation", b"gather") == 0
    net.evaluate()
    hv, _, _ = net.light_volume()
    assert np.array_equal(bits(hv), bits(lv))
    np.testing.assert_allclose(lvf, lv, rtol=2e-5, atol=1e-5 * float(lv.max()))
    # the reference formulation through the same processor: atomic splat, tolerance
    assert host.cpmh_set_property_string(net.h, b"lightvolume", b"formulation", b"splat") == 0
    net.evaluate()
    sv, _, _ = net.light_volume()
    np.testing.assert_allclose(sv, lv, rtol=1e-4, atol=1e-5 * float(lv.max()))
    # 4 x float32 output and radius-sized grid (volumeSizeOption "radius": ceil(1/r)^3)
    assert host.cpmh_set_property_string(net.h, b"lightvolume", b"volumeDataType", b"4xfloat32") == 0
    assert host.cpmh_set_property_float(net.h, b"lightvolume", b"volumeSizeOption", 0.0) == 0
    net.evaluate()
    rv, dims, ch = net.light_volume()
    n = int(np.ceil(1.0 / fr.radius))
    assert dims == (n, n, n) and ch == 4 and rv.reshape(-1, 4)[:, :3].sum() > 0
    net.close()


def test_network_progressive_refinement(host, ctx, cpm):
    """enableRefinement through the processor network: every timer tick is one more iteration (continued RNG streams, the
    radius schedule of PhotonData::advanceToNextIteration) and the light-volume processor keeps the running average --
    bit for bit what the Python driver's ProgressivePhotonMapper computes from the same light plane."""
    S, P = cpm.synthetic, cpm.pipeline
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    net = Net(host, vol, 160, pos, d, S.WORKSPACE_TF_POINTS, size_option=2)
    assert host.cpmh_enable_refinement(net.h, 1) == 0
    net.evaluate(first=True)
    o, u, v, area = net.plane()
    idx = np.array([i for q in CUBE_QUADS for i in (q[0], q[1], q[2], q[0], q[2], q[3])], np.int32)
    pm = P.ProgressivePhotonMapper(ctx, vol, net.tf_lut(), 160, (32, 32, 32), light_travel_direction=net.direction(),
                                   light_plane=(o, u, v, area), mesh_intersection=(S.UNIT_CUBE_VERTICES, idx), alpha=0.5)
    want = _n(pm.iterate()).copy()
    hv, _, _ = net.light_volume()
    assert np.array_equal(bits(hv), bits(want))
    for it in range(2, 5):
        assert host.cpmh_refine(net.h) == it
        assert host.cpmh_last_light_volume_path(net.h) == b"progressive"
        want = _n(pm.iterate()).copy()
        assert abs(host.cpmh_radius(net.h) - pm.radius) < 1e-12
        assert np.array_equal(bits(net.photons()), bits(_n(pm.photons)))
        hv, _, _ = net.light_volume()
        assert np.array_equal(bits(hv), bits(want)), it
    # a TF edit resets the iteration: full evaluation again
    net.set_tf(S.WORKSPACE_TF_POINTS)
    net.evaluate()
    assert host.cpmh_last_light_volume_path(net.h) == b"full"
    net.close()


def test_network_shard_reduce_call_site(host, cpm):
    """The multi-GPU call site of the light-volume processor with a real RCCL communicator of size 1: a full evaluation
    goes through cpm_allreduce_grid, an add-remove update through cpm_allreduce_grid_bricks (touched bricks only), and the
    outport carries what a network without communicator produces."""
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    nets = [Net(host, vol, 128, pos, d, base, correlated=True) for _ in range(2)]
    assert host.cpmh_enable_shard_reduce(nets[0].h) == 0
    for net in nets:
        assert host.cpmh_set_property_string(net.h, b"tracer", b"importanceBranchPolicy", b"always") == 0
    for net in nets:
        net.evaluate(first=True)
        assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    # (the first frames' payload is sized for a quarter of the bricks: a denser union is summed densely -- by the overflow
    # fall-back, or from the start once the policy knows the union)
    assert host.cpmh_last_reduce(nets[0].h) in (b"non-zero bricks", b"dense (overflow)", b"dense") and host.cpmh_last_reduce(nets[1].h) == b"none"
    a, _, _ = nets[0].light_volume()
    b, _, _ = nets[1].light_volume()
    assert np.array_equal(bits(a), bits(b))
    for net in nets:
        net.set_tf(edit)
        net.evaluate()
        assert host.cpmh_last_light_volume_path(net.h) == b"incremental"
    assert host.cpmh_last_reduce(nets[0].h) in (b"touched bricks", b"dense (overflow)", b"dense")
    a, _, _ = nets[0].light_volume()
    b, _, _ = nets[1].light_volume()
    # the reduced volume was refreshed in the touched bricks only; elsewhere it still holds the previous sum -- which is what
    # the partial volume holds there too (atomic +- splats leave untouched voxels alone)
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5 * float(b.max()))
    for net in nets:
        net.close()


def test_network_gl_sharing_call_site_without_context(host, cpm):
    """`glsharing` call site of the light-volume processor on a box without a display: a pixel-unpack buffer named by the host is
    not registered (no current OpenGL context), the evaluation completes and the outport's device buffer is unaffected."""
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    nets = [Net(host, vol, 128, pos, d, S.WORKSPACE_TF_POINTS) for _ in range(2)]
    assert host.cpmh_last_gl_copy(nets[0].h) == b"none"
    host.cpmh_share_light_volume_gl(nets[0].h, 5, 0)
    for net in nets:
        net.evaluate(first=True)
    assert host.cpmh_last_gl_copy(nets[0].h) == b"no context" and host.cpmh_last_gl_copy(nets[1].h) == b"none"
    a, _, _ = nets[0].light_volume()
    b, _, _ = nets[1].light_volume()
    assert np.array_equal(bits(a), bits(b)) and a.max() > 0
    for net in nets:
        net.close()


def test_network_correlated_tf_edit(host, cpm):
    """TF edit through the processor network: the importance branch re-traces only selected photons with
    their original RNG streams and lands on the photons of a from-scratch evaluation."""
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    net = Net(host, vol, 128, pos, d, base, correlated=True)
    net.evaluate(first=True)
    assert host.cpmh_n_recomputed(net.h) == -1 and host.cpmh_last_light_volume_path(net.h) == b"full"
    before = net.photons()
    assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    net.set_tf(edit)
    net.evaluate()
    n = host.cpmh_n_recomputed(net.h)
    assert 0 < n < before.shape[0]
    assert host.cpmh_last_light_volume_path(net.h) == b"incremental"
    after = net.photons()
    lv_inc, _, _ = net.light_volume()
    fresh = Net(host, vol, 128, pos, d, edit, correlated=False)
    fresh.evaluate(first=True)
    assert np.array_equal(bits(after), bits(fresh.photons()))
    assert 0 < (bits(after) != bits(before)).any(axis=1).sum() <= n
    lv_full, _, _ = fresh.light_volume()
    np.testing.assert_allclose(lv_inc, lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
    # progressive: 10 % per evaluation, continued until nothing remains
    net2 = Net(host, vol, 128, pos, d, base, correlated=True)
    net2.evaluate(first=True)
    assert host.cpmh_set_property_float(net2.h, b"tracer", b"maxIncrementalPhotonsToUpdate", 10.0) == 0
    net2.set_tf(edit)
    net2.evaluate()
    rounds = 1
    while host.cpmh_remaining(net2.h) > 0:
        net2.evaluate()
        rounds += 1
    assert rounds > 1
    assert np.array_equal(bits(net2.photons()), bits(after))
    # exact incremental update (not a reference property): bit-identical to the from-scratch light volume
    # (needs the bit-exact "gather" formulation on both sides)
    net3 = Net(host, vol, 128, pos, d, base, correlated=True)
    assert host.cpmh_set_property_string(net3.h, b"lightvolume", b"formulation", b"gather") == 0
    assert host.cpmh_set_property_string(fresh.h, b"lightvolume", b"formulation", b"gather") == 0
    fresh.evaluate()
    lv_full, _, _ = fresh.light_volume()
    # (set before the first evaluation, as a deserialised workspace does: the processor then keeps the photon snapshot the
    # exact add-remove reads -- switched on later, the first edit is served by a full gather, same bits)
    assert host.cpmh_set_property_float(net3.h, b"lightvolume", b"exactIncrementalUpdate", 1.0) == 0
    net3.evaluate(first=True)
    assert host.cpmh_set_property_float(net3.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    net3.set_tf(edit)
    net3.evaluate()
    assert host.cpmh_last_light_volume_path(net3.h) == b"exact incremental"
    lv_exact, _, _ = net3.light_volume()
    assert np.array_equal(bits(lv_exact), bits(lv_full))
    for x in (net, fresh, net2, net3):
        x.close()


def test_stage_timing_log(cpm, ctx):
    """CPM_PROFILING=1: the processors log per-stage kernel times like the reference's IVW_PROFILING lines
    (tracercl.cpp:562-598, ...processorcl.cpp:247-261).  Run in a child process: the flag is read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np, torch
torch.zeros(1, device="cuda")
import cpm_amd
S = cpm_amd.synthetic
lib = C.CDLL(str(cpm_amd.binding.LIB_PATH.parent / "libcpm_host.so"))
lib.cpmh_create.restype = C.c_void_p
lib.cpmh_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float * 3),
                            C.POINTER(C.c_float * 3), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
lib.cpmh_evaluate.argtypes = [C.c_void_p, C.c_int]
vol = S.homogeneous_volume(16)
tf = np.array(S.WORKSPACE_TF_POINTS, np.float32)
pos, d = (C.c_float * 3)(0.5, 0.5, 2.5), (C.c_float * 3)(0.0, 0.0, -1.0)
h = lib.cpmh_create(vol.ctypes.data, 0, 16, 16, 16, 32, 32, C.byref(pos), C.byref(d), tf.ctypes.data, tf.shape[0], 2, 1, 0)
assert h and lib.cpmh_evaluate(h, 1) == 0
''' % (str(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),)
    env = dict(os.environ, CPM_PROFILING="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    assert "Photon tracing: " in log and "trace_kernel" in log and " ms" in log
    assert "Computed photons: 1024 = 100.00 %" in log
    assert "Photons to light volume: " in log and "fast_scatter_kernel" in log and "fast_brick_kernel" in log


def test_radixsort_processor(host, host_extras, ctx):
    """(extras build) org.inviwo.RadixSortCL created through the module factory: keys sorted ascending, data permuted with them
    (stable), pass-through of the data buffer (radixsortcl.cpp:208-259)."""
    assert not hasattr(host, "cpmh_radixsort_processor")      # not in the product library
    host = host_extras
    host.cpmh_radixsort_processor.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(4)
    n = 100_003
    keys = rng.integers(0, 1 << 20, n).astype(np.uint32)
    data = np.arange(n, dtype=np.uint32)
    k, d = keys.copy(), data.copy()
    assert host.cpmh_radixsort_processor(k.ctypes.data, d.ctypes.data, n) == 0
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(d, data[order])
