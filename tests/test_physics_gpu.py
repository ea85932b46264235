"""Statistical checks of the tracer against closed forms -- independent of the oracle, which restates the reference's Woodcock
loop line by line but cannot be pinned to the reference's OpenCL output: in a homogeneous medium the
free path of delta (Woodcock) tracking is exponential with rate sigma = tauMax * sampleBaseInterval * opacity = 150 * alpha
(ref cl/transmittance.cl:126-144, cl/photontracer.cl:160), so the fraction of photons absorbed inside a slab of length L and
their mean depth are known numbers.  1 M photons: tolerances of four standard errors."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(3.402823466e+38)


@pytest.mark.parametrize("alpha", [0.002, 0.01, 0.05])
@pytest.mark.parametrize("direction,axis", [((0.0, 0.0, -1.0), 2), ((1.0, 0.0, 0.0), 0)])
def test_free_path_is_exponential(ctx, cpm, alpha, direction, axis):
    S, P = cpm.synthetic, cpm.pipeline
    torch = ctx.torch
    n_side = 1024
    fr = P.PhotonFrame(ctx, S.homogeneous_volume(64, 128), S.homogeneous_tf(alpha), n_side, (32,) * 3, light_travel_direction=direction)
    fr.trace()
    torch.cuda.synchronize()
    ph = fr.photons.cpu().numpy()
    isect = fr.isect.cpu().numpy()
    entered = isect[:, 0] < isect[:, 1]
    n = int(entered.sum())
    assert n > 0.9 * n_side * n_side          # the light plane covers the cube's face
    L = isect[entered, 1] - isect[entered, 0]
    assert np.allclose(L, 1.0, atol=1e-4)     # axis-aligned rays through the unit cube
    absorbed = entered & (ph[:, 0] != FLT_MAX)
    sigma = 150.0 * alpha
    p = 1.0 - math.exp(-sigma)
    frac = absorbed.sum() / n
    assert abs(frac - p) < 4.0 * math.sqrt(p * (1.0 - p) / n), (frac, p)
    # depth of the absorbed photons below the entry face: a truncated exponential
    coord = ph[absorbed, axis].astype(np.float64)
    depth = (1.0 - coord) if direction[axis] < 0 else coord
    assert depth.min() >= -1e-6 and depth.max() <= 1.0 + 1e-6
    mean = 1.0 / sigma - math.exp(-sigma) / p
    second = (2.0 / sigma ** 2 - math.exp(-sigma) * (1.0 + 2.0 / sigma + 2.0 / sigma ** 2)) / p
    se = math.sqrt(max(second - mean * mean, 1e-12) / absorbed.sum())
    assert abs(depth.mean() - mean) < 4.0 * se + 1e-4, (depth.mean(), mean)
    # ... and its power: the sample's power divided by the opacity at the collision (ref cl/photontracer.cl:174-176)
    ls = fr.light_samples.cpu().numpy()
    want = ls[absorbed, 3] / np.float32(max(alpha, 0.01))
    assert np.allclose(ph[absorbed, 3], want, rtol=1e-6)


@pytest.mark.parametrize("g,isotropic", [(0.6, False), (-0.4, False), (0.0, False), (0.7, True)])
def test_phase_function_mean_cosine(ctx, cpm, g, isotropic):
    """Multiple scattering: the direction after the first scatter event, read back from the second interaction's
    record (its encoded direction; a sentinel carries it too), against the phase function's first moment -- E[cos] = g for
    Henyey-Greenstein, 0 for isotropic scattering -- and the scattered fraction against the albedo (scattering TF = TF: 1/2)."""
    S, P, B = cpm.synthetic, cpm.pipeline, cpm.binding
    torch = ctx.torch
    direction = (0.3, 0.5, -1.0)
    fr = P.PhotonFrame(ctx, S.homogeneous_volume(64, 128), S.homogeneous_tf(0.05), 768, (32,) * 3, light_travel_direction=direction,
                       max_interactions=2, material=(g, 0.0, 0.0, 0.0),
                       shading_type=B.CPM_PHASE_ISOTROPIC if isotropic else B.CPM_PHASE_HENYEY_GREENSTEIN)
    fr.trace()
    torch.cuda.synchronize()
    ph = fr.photons.cpu().numpy().reshape(2, fr.n, 8)
    first_real = ph[0, :, 0] != FLT_MAX
    d0 = np.asarray(P._normalize(direction), np.float64)
    th, phi = ph[1, :, 6].astype(np.float64), ph[1, :, 7].astype(np.float64)
    d1 = np.stack([np.sin(th) * np.cos(phi), np.sin(th) * np.sin(phi), np.cos(th)], 1)
    cos = d1 @ d0
    scattered = first_real & (cos < 1.0 - 1e-6)          # the record after an absorption keeps the light's direction
    n_coll = int(first_real.sum())
    frac = scattered.sum() / n_coll
    assert n_coll > 100_000 and abs(frac - 0.5) < 4.0 * math.sqrt(0.25 / n_coll) + 1e-3, frac
    want = 0.0 if isotropic else g
    assert abs(cos[scattered].mean() - want) < 4.0 / math.sqrt(scattered.sum()) + 2e-3, (cos[scattered].mean(), want)
