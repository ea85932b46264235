"""The photon importance (DDA of a photon's stored path through the importance grid, ref
progressivephotonmapping/cl/photonrecomputationdetector.cl:92-157, uniformgridcl/cl/uniformgrid/uniformgrid.cl traversal) against
its closed form -- independent of the oracle's restatement: on a grid of constant value c the sum over the traversed cells of
c * (fraction of the segment inside the cell) * |segment| is c * |segment| whatever the cells are, so the importance of a photon
is ceil(100 * c * |x2 - x1|) with x1 the entry point and x2 the photon's position (or the exit point), both in voxel index
coordinates.  A cell missed or counted twice by the walk shows as a wrong length."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(3.402823466e+38)


@pytest.mark.parametrize("direction", [(0.3, 0.5, -1.0), (0.0, 0.0, -1.0), (-1.0, 0.2, 0.1)])
@pytest.mark.parametrize("region", [8, 4])
def test_importance_on_a_constant_grid_is_the_path_length(ctx, cpm, direction, region):
    S, P = cpm.synthetic, cpm.pipeline
    torch = ctx.torch
    vdim = 64
    fr = P.PhotonFrame(ctx, S.heterogeneous_volume(vdim), S.workspace_tf(), 256, (32,) * 3, light_travel_direction=direction)
    fr.trace()
    gd = (vdim // region,) * 3
    c = 0.37
    grid = torch.full((gd[0] * gd[1] * gd[2],), c, dtype=torch.float32, device=ctx.device)
    imp = torch.full((fr.n,), 2147483647, dtype=torch.int32, device=ctx.device)
    ctx.photon_importance(grid, gd, (float(region),) * 3, list(fr.vol.desc.texture_to_index), fr.photons, 0, fr.light_samples, fr.isect,
                          fr.n, 1, fr.n, imp, fix_exit_point=True)
    torch.cuda.synchronize()
    got = 2147483647 - imp.cpu().numpy().astype(np.int64)
    ls, isect, ph = fr.light_samples.cpu().numpy().astype(np.float64), fr.isect.cpu().numpy().astype(np.float64), fr.photons.cpu().numpy()
    d = np.asarray(P._normalize(direction), np.float64)
    inside = isect[:, 0] < isect[:, 1]
    assert inside.sum() > 0.5 * fr.n and (got[~inside] == 0).all()
    entry = ls[:, 0:3] + isect[:, 0:1] * d
    end = np.where((ph[:, 0:1] == FLT_MAX), ls[:, 0:3] + isect[:, 1:2] * d, ph[:, 0:3].astype(np.float64))
    length = np.linalg.norm((end - entry) * vdim, axis=1)             # textureToIndex of the default matrices: * dims (- 0.5 cancels)
    want = np.ceil(100.0 * c * length)
    err = np.abs(got[inside] - want[inside])
    # fp32 sums of the walk against a float64 length: the ceiling may fall on either side of an integer, and a stored direction
    # (two fp32 angles) is not the float64 one
    assert err.max() <= 2 and (err == 0).mean() > 0.97, (err.max(), (err == 0).mean())
    assert (ph[inside, 0] != FLT_MAX).any()   # (rays that leave unabsorbed -- the exit-point branch -- occur for the oblique lights)
