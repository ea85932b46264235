"""Progressive photon mapping end to end: K iterations of trace (RNG written back) -> bin -> gather with
the shrinking radius, light volume = running average of the iterations' estimates -- against the oracle, iteration by
iteration: photons and RNG streams bit for bit, the light volume bit for bit in both formulations (each against its own
restatement), and the radius schedule against the closed form (ref photondata.cpp:67-79,
processor/progressivephotontracercl.cpp:252-260, cl/photontracer.cl:211-215)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _n(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a.view(dtype) if dtype is not None else a


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("formulation", ["gather", "fast"])
@pytest.mark.parametrize("max_inter", [1, 3])
def test_progressive_iterations_equal_oracle(ctx, oracle, cpm, formulation, max_inter):
    from oracle_binding import OTraceParams
    S, P = cpm.synthetic, cpm.pipeline
    vdim, gdim, n_side, K, alpha = 64, 32, 192, 4, 0.5
    vol_np, tf = S.heterogeneous_volume(vdim), S.workspace_tf()
    pm = P.ProgressivePhotonMapper(ctx, vol_np, tf, n_side, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), alpha=alpha,
                                   formulation=formulation, max_interactions=max_inter, shading_type=cpm.binding.CPM_PHASE_HENYEY_GREENSTEIN,
                                   material=(0.3, 0.0, 0.0, 0.0))
    n = pm.n
    ls, isect = _n(pm.light_samples), _n(pm.isect)
    rng_o = _n(pm.rng_initial, np.uint32).copy()
    ovol = oracle.volume(vol_np)
    og = oracle.grid((gdim,) * 3, 1)
    radius = pm.radius0
    want = None
    radii = []
    for it in range(1, K + 1):
        pm.iterate()
        if it > 1:
            radius = radius * (((it - 1) + alpha) / (1.0 + (it - 1))) ** (1.0 / 3.0)     # photondata.cpp:72-77
        radii.append(radius)
        assert pm.iteration == it and abs(pm.radius - radius) < 1e-15
        po = OTraceParams()
        po.material[0] = 0.3
        po.step_size = 1.0 / vdim
        po.n_light_samples = n
        po.max_interactions = max_inter
        po.total_photons = n
        po.flags = 1                                            # PROGRESSIVE: the oracle writes its streams back too
        ph_o = np.zeros((n * max_inter, 8), np.float32)
        oracle.trace(ovol, tf, S.UNIT_CUBE_AABB, po, ls, isect, rng_o, ph_o)
        assert np.array_equal(bits(_n(pm.photons)), bits(ph_o))
        assert np.array_equal(_n(pm.rng, np.uint32), rng_o)      # the streams continue where they stopped
        scale = oracle.relative_irradiance_scale(radius, n)
        est = np.zeros(gdim ** 3, np.float32)
        if formulation == "fast":
            oracle.gather_fast(ph_o, n * max_inter, og, radius, scale, est)
        else:
            _, cs, srt = oracle.bin(ph_o, n * max_inter, og)
            oracle.gather(srt, cs, n * max_inter, og, radius, scale, est)
        want = est if it == 1 else oracle.mix_f32(want, est, 1.0 / it)   # running average
        assert np.array_equal(bits(_n(pm.light_volume)), bits(want)), it
    assert radii[-1] < radii[0] and want.sum() > 0
    # every iteration traced different photons (the streams moved on) ...
    first = _n(pm.photons).copy()
    pm.iterate()
    assert (bits(_n(pm.photons)) != bits(first)).any()
    # ... and a reset starts over: iteration 1 again, original streams and radius
    pm.reset()
    lv = _n(pm.iterate()).copy()
    pm2 = P.ProgressivePhotonMapper(ctx, vol_np, tf, n_side, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), alpha=alpha,
                                    formulation=formulation, max_interactions=max_inter, shading_type=cpm.binding.CPM_PHASE_HENYEY_GREENSTEIN,
                                    material=(0.3, 0.0, 0.0, 0.0))
    assert np.array_equal(bits(lv), bits(_n(pm2.iterate())))


def test_progressive_average_converges(ctx, cpm):
    """The running average over the first iterations moves towards the many-photon estimate (a sanity check of the
    accumulation, not a parity statement; later iterations shrink the radius below the light volume's voxel size, where
    the estimate on a fixed grid stops improving -- the reference's schedule, photondata.cpp:72-77, has no floor either)."""
    S, P = cpm.synthetic, cpm.pipeline
    vol_np, tf = S.heterogeneous_volume(64), S.workspace_tf()
    dense = P.PhotonFrame(ctx, vol_np, tf, 1024, (32,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
    ref = _n(dense.frame_fast()).astype(np.float64)
    pm = P.ProgressivePhotonMapper(ctx, vol_np, tf, 256, (32,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
    errs = []
    for it in range(6):
        lv = _n(pm.iterate()).astype(np.float64)
        errs.append(np.abs(lv - ref).sum() / np.abs(ref).sum())
    assert min(errs[1:]) < 0.85 * errs[0], errs
