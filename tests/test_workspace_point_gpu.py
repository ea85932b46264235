"""The workspace's own operating point through the C++ processors: TWO directional lights on the tracer's multi-inport (one
launch per light, photonOffset = the photons of the lights before: ref progressivephotontracercl.cpp:481-527,543-549; wiring
workspaces/CorrelatedPhotonMappingSingleVolume.inv:1195-1198,1267-1270), a 4 : 4 : 0.75 volume (512 x 512 x 96, :740-757), the
clip ranges of the proxy geometry (:393-420) and a light volume of half the size (:555-557) -- against the ORACLE: the photons
of both lights' ranges bit for bit, the light volume of the default (tolerance-mode) formulation bit for bit against its
fixed-point restatement and within the stated tolerance of the exact gather; then a correlated TF edit on the same network."""
import ctypes as C

import numpy as np
import pytest

from test_parity_gpu import bits
from test_host_layer_gpu import host, Net, CUBE_QUADS  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

# the workspace's two directional lights, world space (CorrelatedPhotonMappingSingleVolume.inv:214,1089); the volume's model
# spans [-1, 1]^3 there -- here data space [0, 1]^3: same directions, positions outside the volume along them
LIGHTS_WORLD = [(-90.045471, 104.828, 312.07489), (94.269867, 148.44716, 302.45557)]


def _lights(cpm):
    out = []
    for w in LIGHTS_WORLD:
        d = cpm.pipeline._normalize(tuple(-x for x in w))   # a directional light shines from its position towards the origin
        out.append((np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * d, d))
    return out


def _bind(host):
    for name, res, args in [("cpmh_add_light", C.c_int, [C.c_void_p, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]),
                            ("cpmh_n_lights", C.c_int, [C.c_void_p]),
                            ("cpmh_set_clip", None, [C.c_void_p] + [C.c_int] * 6),
                            ("cpmh_light_plane_of", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float * 10), C.POINTER(C.c_float * 3)]),
                            ("cpmh_last_tracer_decision", C.c_char_p, [C.c_void_p])]:
        f = getattr(host, name)
        f.restype, f.argtypes = res, args


def _network(host, cpm, vol, n_side, tf_pts, clip, correlated):
    lights = _lights(cpm)
    net = Net(host, vol, n_side, lights[0][0], lights[0][1], tf_pts, size_option=2, correlated=correlated)
    assert host.cpmh_add_light(net.h, C.byref((C.c_float * 3)(*lights[1][0])), C.byref((C.c_float * 3)(*lights[1][1]))) == 1
    assert host.cpmh_n_lights(net.h) == 2
    host.cpmh_set_clip(net.h, *clip)
    return net


class _OracleScene:
    """The two-light scene restated with the oracle, from the planes the host fitted: light k's samples / entry-exit pairs, the
    RNG streams of all N = 2 n photons (light k's photons at [k n, (k + 1) n)), the clipped box."""

    def __init__(self, oracle, cpm, host, net, vol, n_side, clip):
        P = cpm.pipeline
        self.oracle, self.vol = oracle, vol
        dz, dy, dx = vol.shape
        self.dims = (dx, dy, dz)
        self.n = n_side * n_side
        self.N = 2 * self.n
        lo = np.array([clip[0] / dx, clip[2] / dy, clip[4] / dz], np.float32)
        hi = np.array([clip[1] / dx, clip[3] / dy, clip[5] / dz], np.float32)
        vtx = np.array([[hi[0] if x else lo[0], hi[1] if y else lo[1], hi[2] if z else lo[2]] for z in (0, 1) for y in (0, 1) for x in (0, 1)], np.float32)
        idx = np.array([i for q in CUBE_QUADS for i in (q[0], q[1], q[2], q[0], q[2], q[3])], np.int32)
        self.aabb = (lo[0], lo[1], lo[2], 1.0, hi[0], hi[1], hi[2], 1.0)
        smp = oracle.uniform_samples_2d(n_side, n_side)
        self.st = np.zeros((self.N, 2), np.uint32)
        self.st[:, 0] = oracle.glibc_rand_sequence(0, self.N)
        oracle.seed_streams(self.st, 1 << 40)
        self.ovol = oracle.volume(vol)
        self.lights = []
        lights = _lights(cpm)
        for k in range(2):
            plane, d = (C.c_float * 10)(), (C.c_float * 3)()
            assert host.cpmh_light_plane_of(net.h, k, C.byref(plane), C.byref(d)) == 0
            a = np.array(list(plane), np.float32)
            o, u, v, area = a[0:3], a[3:6], a[6:9], float(a[9])
            d = np.array(list(d), np.float32)
            # the host's plane fit over the CLIPPED proxy mesh agrees with the Python twin of the reference's CPU code
            po, pu, pv = P.fit_plane_aligned_obb(vtx, lights[k][0], lights[k][1])
            assert abs(area - float(np.linalg.norm(pu) * np.linalg.norm(pv))) < 1e-5 * max(area, 1.0)
            ls = oracle.directional_light_samples(smp, (1, 1, 1), d, o, u, v, area)
            self.lights.append((ls, oracle.light_sample_mesh_intersection(vtx, idx, ls)))
        self.gdims = (dx // 2, dy // 2, dz // 2)
        self.radius = float(host.cpmh_radius(net.h))
        assert abs(self.radius - cpm.synthetic.photon_radius_texture((dx, dy, dz), 1.0)) < 1e-9
        self.scale = oracle.relative_irradiance_scale(self.radius, self.N)

    def trace(self, tf_lut, photons=None, indices=None):
        """All photons (indices None) or the listed ones, one launch per light with photonOffset = k n."""
        from oracle_binding import OTraceParams
        ph = np.zeros((self.N, 8), np.float32) if photons is None else photons
        for k, (ls, isect) in enumerate(self.lights):
            p = OTraceParams()
            p.step_size = 1.0 / max(self.dims)     # samplingRate 1 x the smallest voxel spacing (progressivephotontracercl.cpp:236-240)
            p.photon_offset = k * self.n
            p.n_light_samples = self.n
            p.max_interactions = 1
            p.total_photons = self.N
            if indices is None:
                self.oracle.trace(self.ovol, tf_lut, self.aabb, p, ls, isect, self.st, ph)
            else:
                self.oracle.trace(self.ovol, tf_lut, self.aabb, p, ls, isect, self.st, ph, recompute_indices=indices, n_recompute=indices.size)
        return ph

    def light_volumes(self, ph):
        og = self.oracle.grid(self.gdims, 1)
        fast = np.zeros(self.gdims[0] * self.gdims[1] * self.gdims[2], np.float32)
        self.oracle.gather_fast(ph, self.N, og, self.radius, self.scale, fast)
        _, cs, srt = self.oracle.bin(ph, self.N, og)
        exact = np.zeros_like(fast)
        self.oracle.gather(srt, cs, self.N, og, self.radius, self.scale, exact)
        return fast, exact

    def correlated_selection(self, photons, new_pts, prev_pts, pipeline, region=8):
        """The reference's importance branch: TF-difference importance of the min/max bricks, every stored path walked through
        it (per light), the photons whose importance moved -- ascending."""
        from oracle_binding import default_matrices
        pos, col = pipeline.tf_difference_points(new_pts, prev_pts)
        mm = self.oracle.volume_minmax(self.ovol, region)
        grid = self.oracle.importance_tf(mm, pos, col)
        gd = tuple(-(-d // region) for d in self.dims)
        t2i, _ = default_matrices(self.dims)
        imp = np.full(self.N, 2147483647, np.uint32)
        for k, (ls, isect) in enumerate(self.lights):
            self.oracle.photon_importance(grid, gd, (float(region),) * 3, t2i, photons, k * self.n, ls, isect, self.n, 1, self.N, imp)
        idx, cnt = self.oracle.select_changed(imp)
        return np.sort(idx[:cnt]).astype(np.uint32)


@pytest.mark.parametrize("dims,n_side,clip", [
    ((128, 128, 24), 256, (18, 128, 2, 128, 0, 24)),      # the workspace's proportions at 1/4 of its size
    ((512, 512, 96), 1024, (73, 512, 7, 512, 0, 96)),     # the workspace itself: 2 x 1024^2 photons, 512 x 512 x 96, light volume 256 x 256 x 48
])
def test_two_lights_noncubic_volume_against_the_oracle(host, ctx, oracle, cpm, dims, n_side, clip):
    _bind(host)
    S = cpm.synthetic
    vol = S.heterogeneous_volume(dims)
    net = _network(host, cpm, vol, n_side, S.WORKSPACE_TF_POINTS, clip, correlated=False)
    net.evaluate(first=True)
    n = n_side * n_side
    assert host.cpmh_n_photons(net.h) == 2 * n
    scene = _OracleScene(oracle, cpm, host, net, vol, n_side, clip)
    ph_o = scene.trace(net.tf_lut())
    fast_o, exact_o = scene.light_volumes(ph_o)
    gdims = scene.gdims
    got = net.photons()
    stored = got[:, 0] < 1e30
    assert stored[:n].sum() > 0 and stored[n:].sum() > 0            # both lights deposit photons
    assert np.array_equal(bits(got[:n]), bits(ph_o[:n])), "light 1's photon range differs from the oracle"
    assert np.array_equal(bits(got[n:]), bits(ph_o[n:])), "light 2's photon range (photonOffset = n) differs from the oracle"
    lv, d, ch = net.light_volume()
    assert d == gdims and ch == 1
    # the processor's default formulation is the tolerance-mode one where it covers the radius (cpm_gather_fast_supported);
    # at this point the radius -- |indexToTexture * (1, 1, 1)| of a 4 : 4 : 0.75 volume, dominated by the coarse z spacing -- is
    # 2.8 light-volume voxels along x and y
    radius = float(host.cpmh_radius(net.h))
    if ctx.gather_fast_supported(cpm.binding.default_grid_desc(gdims, 1), radius):
        assert np.array_equal(bits(lv), bits(fast_o)), "light volume differs from the fixed-point restatement"
    else:
        assert np.array_equal(bits(lv), bits(exact_o)), "light volume differs from the oracle's exact gather"
    np.testing.assert_allclose(lv, exact_o, rtol=2e-5, atol=1e-5 * float(exact_o.max()))
    np.testing.assert_allclose(fast_o, exact_o, rtol=2e-5, atol=1e-5 * float(exact_o.max()))
    # the bit-exact formulation through the same processor
    assert host.cpmh_set_property_string(net.h, b"lightvolume", b"formulation", b"gather") == 0
    net.evaluate()
    lv2, _, _ = net.light_volume()
    assert np.array_equal(bits(lv2), bits(exact_o))
    net.close()


@pytest.mark.parametrize("dims,n_side,clip", [((128, 128, 24), 256, (18, 128, 2, 128, 0, 24)), ((512, 512, 96), 1024, (73, 512, 7, 512, 0, 96))])
def test_correlated_tf_edit_at_the_workspace_point(host, ctx, oracle, cpm, dims, n_side, clip):
    """A TF edit on the two-light network: the importance branch re-traces the selected photons of BOTH lights (their ranges of
    the shared buffer) and lands on the photons of the oracle's from-scratch trace with the edited TF; the add-remove light volume
    stays within the atomic splat's tolerance of the oracle's exact gather."""
    _bind(host)
    S = cpm.synthetic
    vol = S.heterogeneous_volume(dims)
    base = list(S.WORKSPACE_TF_POINTS)
    net = _network(host, cpm, vol, n_side, base, clip, correlated=True)
    assert host.cpmh_set_property_string(net.h, b"tracer", b"importanceBranchPolicy", b"always") == 0
    net.evaluate(first=True)
    assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    n = n_side * n_side
    # BASELINE config 3's edit (TF point 4: 0.2218 -> 0.26: thousands of photons selected, few of them land elsewhere), then a
    # stronger one on top of it (point 5: 0.285 -> 0.40) that selects most photons of both lights.  Expected = the ORACLE's
    # importance branch: the branch is the reference's algorithm, conservative but not exact -- bricks carry no border voxel
    # (ref uniformgridcl/cl/uniformgrid/volumeminmax.cl:46-57), so a path that samples across a brick face it never crosses can
    # be missed: 6 of 2 M photons on the strong edit at full size differ from a from-scratch trace, here as in the oracle.
    scene = _OracleScene(oracle, cpm, host, net, vol, n_side, clip)
    pts = list(base)
    for point, where, both in ((3, 0.26, False), (4, 0.40, True)):
        before = net.photons()
        prev = list(pts)
        pts[point] = (where,) + pts[point][1:]
        net.set_tf(pts)
        net.evaluate()
        assert host.cpmh_last_tracer_decision(net.h) == b"importance branch"
        nre = host.cpmh_n_recomputed(net.h)
        assert 0 < nre < 2 * n
        after = net.photons()
        changed = (bits(after) != bits(before)).any(axis=1)
        assert changed.sum() <= nre
        if both:
            assert changed[:n].any() and changed[n:].any()      # photons of both lights' ranges were re-traced
        sel = scene.correlated_selection(before, pts, prev, cpm.pipeline)
        assert sel.size == nre and (sel < n).any() and (sel >= n).any()
        want = scene.trace(net.tf_lut(), photons=before.copy(), indices=sel)
        assert np.array_equal(bits(after), bits(want))
        scratch = scene.trace(net.tf_lut())
        assert (bits(want) != bits(scratch)).any(axis=1).sum() <= 1e-5 * 2 * n    # ... and (all but) every photon of a from-scratch trace
        _, exact_o = scene.light_volumes(want)
        lv, _, _ = net.light_volume()
        np.testing.assert_allclose(lv, exact_o, rtol=1e-3, atol=2e-5 * float(exact_o.max()))
    net.close()
