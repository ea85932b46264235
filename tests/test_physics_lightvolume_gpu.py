"""The density estimate against its closed form -- independent of the oracle (which restates splatPhoton, ref
cl/photonstolightvolume.cl:31-79, and the Epanechnikov kernel, cl/densityestimationkernel.cl:43-60, but can be pinned to the
reference's OpenCL for the kernel function only): for photons uniformly distributed well inside the grid the expected sum
of the light volume is   k * sum(power) * 0.75 * cells_per_unit_volume * integral over the ball of (1 - d^2 / r^2) dV
= k * sum(power) * 0.75 * G^3 * 8 pi r^3 / 15,   k = relativeIrradianceScale / (4 pi)   (uniform positions average the
discrete sum over the voxel centres into the integral).  All three formulations -- the reference's atomic splat, the bit-exact
gather, the tolerance-mode brick gather -- must land on it."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("radius_voxels", [0.7, 1.0])
def test_total_of_the_light_volume(ctx, cpm, radius_voxels):
    S, P = cpm.synthetic, cpm.pipeline
    torch = ctx.torch
    G = 64
    fr = P.PhotonFrame(ctx, S.homogeneous_volume(64, 128), S.homogeneous_tf(0.5), 1024, (G,) * 3, light_travel_direction=(0.0, 0.0, -1.0),
                       radius_voxels=radius_voxels)
    n = fr.n
    g = torch.Generator(device="cpu").manual_seed(7)
    rec = torch.zeros((n, 8), dtype=torch.float32)
    rec[:, 0:3] = 0.25 + 0.5 * torch.rand((n, 3), generator=g)      # well inside: no splat box leaves the grid
    rec[:, 3] = 0.5 + torch.rand(n, generator=g)                      # power
    rec[::97, 0] = 3.402823466e+38                                    # some sentinels: no contribution
    fr.photons.copy_(rec.to(ctx.device))
    live = rec[:, 0] != 3.402823466e+38
    r = fr.radius                                                     # texture units; sqrt(3) * radius_voxels / 64 -> 1.2 / 1.7 grid cells
    k = fr.scale / (4.0 * math.pi)
    want = k * float(rec[live, 3].double().sum()) * 0.75 * G ** 3 * 8.0 * math.pi * r ** 3 / 15.0
    totals = {}
    fr.bin_fast(); fr.gather_fast(); torch.cuda.synchronize()
    totals["brick gather"] = float(fr.light_volume.double().sum())
    fr.bin(); fr.gather(); torch.cuda.synchronize()
    totals["exact gather"] = float(fr.light_volume.double().sum())
    out = torch.zeros_like(fr.light_volume)
    fr.splat(out); torch.cuda.synchronize()
    totals["atomic splat"] = float(out.double().sum())
    for name, got in totals.items():
        assert abs(got / want - 1.0) < 2e-3, (name, got, want)        # 1 M positions: the lattice average is within 0.1 %
    assert abs(totals["brick gather"] / totals["exact gather"] - 1.0) < 1e-5
