"""The CPU oracle's tracer against closed forms (no GPU): in a homogeneous medium the free path of delta tracking is
exponential with rate 150 * alpha (ref cl/transmittance.cl:126-144, cl/photontracer.cl:160), so the absorbed fraction in a
slab and the mean depth are known numbers.  The oracle restates the reference's loop but cannot be pinned to the reference's
OpenCL output for this kernel; this pins its behaviour to the physics instead.  tests/test_physics_gpu.py
does the same for the HIP tracer at 1 M photons."""
import math
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
import cpm_amd  # noqa: E402
from oracle_binding import Oracle, OTraceParams  # noqa: E402

FLT_MAX = np.float32(3.402823466e+38)


@pytest.mark.parametrize("alpha", [0.005, 0.03])
def test_oracle_free_path_is_exponential(alpha):
    S, P = cpm_amd.synthetic, cpm_amd.pipeline
    o = Oracle()
    nx = ny = 384
    n = nx * ny
    d = P._normalize((0.0, 0.0, -1.0))
    origin = np.array([0.5] * 3, np.float32) - np.float32(2.0) * d
    po, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
    area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
    ls = o.directional_light_samples(o.uniform_samples_2d(nx, ny), (1, 1, 1), d, po, u, v, area)
    isect = o.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = o.glibc_rand_sequence(0, n)
    o.seed_streams(st, 1 << 40)
    p = OTraceParams()
    p.step_size = 1.0 / 32
    p.n_light_samples = n
    p.max_interactions = 1
    p.total_photons = n
    photons = np.zeros((n, 8), np.float32)
    o.trace(o.volume(S.homogeneous_volume(32, 128)), S.homogeneous_tf(alpha), S.UNIT_CUBE_AABB, p, ls, isect, st, photons)
    entered = isect[:, 0] < isect[:, 1]
    m = int(entered.sum())
    assert m > 0.9 * n and np.allclose(isect[entered, 1] - isect[entered, 0], 1.0, atol=1e-4)
    absorbed = entered & (photons[:, 0] != FLT_MAX)
    sigma = 150.0 * alpha
    pr = 1.0 - math.exp(-sigma)
    assert abs(absorbed.sum() / m - pr) < 4.0 * math.sqrt(pr * (1.0 - pr) / m)
    depth = 1.0 - photons[absorbed, 2].astype(np.float64)
    mean = 1.0 / sigma - math.exp(-sigma) / pr
    second = (2.0 / sigma ** 2 - math.exp(-sigma) * (1.0 + 2.0 / sigma + 2.0 / sigma ** 2)) / pr
    assert abs(depth.mean() - mean) < 4.0 * math.sqrt((second - mean * mean) / absorbed.sum()) + 1e-4
    assert np.allclose(photons[absorbed, 3], ls[absorbed, 3] / np.float32(max(alpha, 0.01)), rtol=1e-6)
