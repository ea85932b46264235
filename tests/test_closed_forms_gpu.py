"""Closed forms and independent numerical evaluations of kernels the oracle can only restate:
 * the entry / exit of a light sample's ray through the volume's box, and through its 12-triangle mesh, against a float64 slab test;
 * the TF importance of a brick's [min, max] range against a dense float64 evaluation of the piecewise-linear difference TF;
 * the volume's min / max bricks against numpy reductions over the same blocks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_box_and_mesh_intersection_against_a_float64_slab_test(ctx, cpm):
    S, P = cpm.synthetic, cpm.pipeline
    for direction in ((0.3, 0.5, -1.0), (-1.0, 0.0, 0.0), (0.6, -0.7, 0.2)):
        fr = P.PhotonFrame(ctx, S.homogeneous_volume(16, 128), S.homogeneous_tf(0.5), 192, (16,) * 3, light_travel_direction=direction)
        ls = fr.light_samples.cpu().numpy().astype(np.float64)
        d = np.asarray(P._normalize(direction), np.float64)
        o = ls[:, 0:3]
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (0.0 - o) / d
            t1 = (1.0 - o) / d
        lo, hi = np.minimum(t0, t1), np.maximum(t0, t1)
        for a in range(3):                      # an axis the ray runs along: inside the slab or not at all
            if d[a] == 0.0:
                ok = (o[:, a] >= 0.0) & (o[:, a] <= 1.0)
                lo[:, a] = np.where(ok, -np.inf, np.inf); hi[:, a] = np.where(ok, np.inf, -np.inf)
        tn, tf = lo.max(axis=1), hi.min(axis=1)
        hit = tn < tf
        got = fr.isect.cpu().numpy().astype(np.float64)
        got_hit = got[:, 0] < got[:, 1]
        clear = np.abs(tn - tf) > 1e-4          # (a ray grazing an edge may go either way ...
        for a in range(3):                      #  ... and so may one that runs along an axis exactly in a face)
            if d[a] == 0.0:
                clear &= (np.abs(o[:, a]) > 1e-6) & (np.abs(o[:, a] - 1.0) > 1e-6)
        assert np.array_equal(hit[clear], got_hit[clear]) and hit.sum() > 0.3 * len(hit)
        both = hit & got_hit
        assert np.abs(got[both, 0] - np.maximum(tn[both], 0.0)).max() < 2e-5 and np.abs(got[both, 1] - tf[both]).max() < 2e-5
        quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        idx = np.array([i for q in quads for i in (q[0], q[1], q[2], q[0], q[2], q[3])], np.int32)
        mesh = ctx.light_sample_mesh_intersection(ctx.torch.from_numpy(S.UNIT_CUBE_VERTICES).to(ctx.device),
                                                  ctx.torch.from_numpy(idx).to(ctx.device), fr.light_samples).cpu().numpy()
        mhit = mesh[:, 0] < mesh[:, 1]
        assert np.array_equal(hit[clear], mhit[clear])
        mb = hit & mhit
        assert np.abs(mesh[mb, 0] - np.maximum(tn[mb], 0.0)).max() < 2e-5 and np.abs(mesh[mb, 1] - tf[mb]).max() < 2e-5


def test_tf_importance_against_a_dense_evaluation(ctx):
    torch = ctx.torch
    rng = np.random.default_rng(3)
    n = 4096
    lo = rng.integers(0, 65535, n).astype(np.uint16)
    hi = np.minimum(65535, lo.astype(np.int64) + rng.integers(0, 20000, n)).astype(np.uint16)
    lo[:64] = hi[:64]                              # degenerate ranges
    pos = np.array([0.0, 0.1, 0.25, 0.26, 0.6, 0.85, 1.0], np.float32)
    col = rng.random((7, 4)).astype(np.float32)
    col[0] = col[-1] = 0.0
    mm = torch.from_numpy(np.stack([lo, hi], 1).copy().view(np.int16)).to(ctx.device)
    out = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    ctx.importance_tf(mm, n, pos, col, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    # per channel: the maximum of the piecewise-linear function over [rx, ry] is attained at rx, ry or a break point inside
    rx, ry = lo.astype(np.float64) / 65535.0, hi.astype(np.float64) / 65535.0
    p64, c64 = pos.astype(np.float64), col.astype(np.float64)
    want = np.zeros(n)
    for ch in range(4):
        f = lambda x: np.interp(x, p64, c64[:, ch])
        m = np.maximum(f(rx), f(ry))
        for k in range(len(p64)):
            inside = (p64[k] >= rx) & (p64[k] <= ry)
            m = np.where(inside, np.maximum(m, c64[k, ch]), m)
        want += m
    assert np.abs(got - want).max() < 1e-5 * max(1.0, want.max()), np.abs(got - want).max()


@pytest.mark.parametrize("dtype,region", [(np.uint8, 8), (np.uint16, 4), (np.float32, 8)])
def test_minmax_bricks_against_numpy(ctx, cpm, dtype, region):
    torch = ctx.torch
    rng = np.random.default_rng(5)
    dim = 64
    if dtype == np.float32:
        vol = rng.random((dim, dim, dim), dtype=np.float32)
        norm = lambda v: v
    else:
        top = np.iinfo(dtype).max
        vol = rng.integers(0, top + 1, (dim, dim, dim)).astype(dtype)
        norm = lambda v: v.astype(np.float64) / top
    v = ctx.volume_create(vol)
    nb = dim // region
    out = torch.zeros((nb ** 3, 2), dtype=torch.int16, device=ctx.device)
    ctx.volume_minmax(v, region, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16).astype(np.int64)
    blocks = norm(vol).reshape(nb, region, nb, region, nb, region)
    mn, mx = blocks.min(axis=(1, 3, 5)).reshape(-1), blocks.max(axis=(1, 3, 5)).reshape(-1)
    # normalised value -> uint16 (the reference stores convert_ushort_sat_rte(65535 * v)): within one unit of the float64 value
    assert np.abs(got[:, 0] - 65535.0 * mn).max() <= 1.0 and np.abs(got[:, 1] - 65535.0 * mx).max() <= 1.0
    assert (got[:, 0] <= got[:, 1]).all()
