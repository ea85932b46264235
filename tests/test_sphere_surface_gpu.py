"""Where the tolerance-mode gather and the reference differ BY CONSTRUCTION: the surface of a photon's sphere.

The reference splat tests  x = sqrt(d^2) / r <= 1  and weights 0.75 (1 - x^2) (ref progressivephotonmapping/cl/photonstolightvolume.cl:57-60,
cl/densityestimationkernel.cl:43-60); cpm_gather_fast tests  d^2 <= r^2  and weights 0.75 (1 - d^2 r^-2) (csrc/cpm_fastvolume.hip) -- no
sqrt, no division.  The two can disagree on whether a voxel centre that sits ON the sphere contributes; the oracle is unpinned for this
kernel (PARITY UNPINNED), so the question is settled on closed forms instead of on the restatement alone:

  photons placed so that a voxel centre lies at |c - p| = r (1 + k ulp(r)), k = -4 ... 4, along each axis in turn (and along diagonals), on
  an isotropic 64^3 grid and on the workspace's anisotropic 256 x 256 x 48 one, one photon per neighbourhood so that every voxel has at
  most one contributor.  (A float32 position cannot resolve ulps of r -- ulp(p) is 20 - 40 ulp(r) here -- so the nine k of a direction land
  on one or two representable positions within +- 2e-6 r of the sphere: as close as the data type lets a photon come.  The closed form is
  evaluated in float64 from the float32 positions and voxel centres the kernels see.)  For every such voxel: inside the sphere (k < 0) the float64 weight 0.75 (1 - |c - p|^2 / r^2) is O(k ulp) -- both
  formulations must give the photon's power * k_scale * that weight within the stated tolerance (atol 1e-5 of one full-weight contribution);
  outside (k > 0) both must give exactly 0 or a value below that atol.  The number of contributing voxels of each formulation is
  reported, and a voxel that contributes in one and not in the other is allowed only with |value| <= atol.
  Worst case found (documented in DESIGN section 2): printed by the test."""
import numpy as np
import pytest

from test_fast_gpu import run_fast, oracle_both, bits

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.402823466e+38)


def _placements(dims, radius, rng):
    """[(photon float8 row, target voxel (x, y, z), k, axis label)]: targets 12 voxels apart (no two photons reach the same voxel)."""
    dx, dy, dz = dims
    r = np.float64(np.float32(radius))
    rows = []
    step = (8, 8, 4 if dz < 64 else 8)     # (a photon reaches at most 3 voxels to either side along x / y here, fewer along a coarse z)
    targets = [(x, y, z) for z in range(3, dz - 3, step[2]) for y in range(8, dy - 8, step[1]) for x in range(8, dx - 8, step[0])]
    rng.shuffle(targets)
    dirs = {"x": (1.0, 0.0, 0.0), "y": (0.0, 1.0, 0.0), "z": (0.0, 0.0, 1.0), "-x": (-1.0, 0.0, 0.0), "xy": (2 ** -0.5, 2 ** -0.5, 0.0),
            "xyz": (3 ** -0.5, 3 ** -0.5, 3 ** -0.5)}
    ti = 0
    for label, d in dirs.items():
        for k in range(-4, 5):
            for rep in range(3):
                vx, vy, vz = targets[ti]; ti += 1
                c = np.array([(vx + 0.5) / dx, (vy + 0.5) / dy, (vz + 0.5) / dz], np.float64)
                dist = r * (1.0 + k * 2.0 ** -23)                     # r (1 + k ulp(1)): k units in the last place of r
                p = (c - dist * np.asarray(d, np.float64)).astype(np.float32)
                if not ((p > 0).all() and (p < 1).all()):
                    continue
                power = np.float32(rng.uniform(0.5, 4.0))
                rows.append((np.array([p[0], p[1], p[2], power, power, power, 0.0, 0.0], np.float32), (vx, vy, vz), k, label))
    return rows


@pytest.mark.parametrize("dims,rvox", [((64, 64, 64), 0.9), ((64, 64, 64), 1.45), ((256, 256, 48), 2.8), ((256, 256, 48), 0.9)])
def test_voxel_centres_on_the_spheres_surface(ctx, oracle, cpm, dims, rvox):
    rng = np.random.default_rng(dims[2] * 31 + int(rvox * 100))
    radius = float(np.float32(rvox) / np.float32(max(dims)))
    rows = _placements(dims, radius, rng)
    assert len(rows) > 120
    ph = np.stack([r[0] for r in rows])
    n = ph.shape[0]
    scale = 3.7
    k_scale = np.float64(np.float32(scale)) / (4.0 * np.pi)
    got, _, _ = run_fast(ctx, cpm, ph, dims, 1, radius, scale)
    want_fast, want_exact = oracle_both(oracle, ph, dims, 1, radius, scale)
    assert np.array_equal(bits(got), bits(want_fast))                 # the HIP path IS the fixed-point restatement, as everywhere
    splat = np.zeros(dims[0] * dims[1] * dims[2], np.float32)
    oracle.splat(ph, n, oracle.grid(dims, 1), radius, scale, splat)   # the reference's own formulation (sequential: a defined order)
    one = float(np.max(np.abs(ph[:, 3]))) * float(k_scale) * 0.75     # one full-weight contribution
    atol = 1e-5 * max(one, float(np.abs(want_exact).max()))
    dx, dy, dz = dims
    r64 = np.float64(np.float32(radius))
    worst = {"fast_vs_closed": 0.0, "ref_vs_closed": 0.0, "fast_vs_ref": 0.0, "disagree_on_contributing": 0, "targets": 0,
             "fast_contributes": 0, "ref_contributes": 0}
    for row, (vx, vy, vz), k, label in rows:
        v = vx + dx * (vy + dy * vz)
        # the voxel centre and the distance as the kernels form them (fp32 operands), evaluated in float64
        # (c = fma(indexToTexture scale, v, translate): one rounding -- the product and sum of two floats are exact in float64)
        c32 = np.array([np.float32(np.float64(np.float32(1.0 / dd)) * vv + np.float64(np.float32(0.5 / dd))) for dd, vv in ((dx, vx), (dy, vy), (dz, vz))],
                       np.float64)
        d2 = float(((c32 - row[:3].astype(np.float64)) ** 2).sum())
        closed = float(row[3]) * float(k_scale) * 0.75 * (1.0 - d2 / float(r64 * r64)) if d2 <= float(r64 * r64) else 0.0
        f, e, s = float(got[v]), float(want_exact[v]), float(splat[v])
        worst["targets"] += 1
        worst["fast_contributes"] += f != 0.0
        worst["ref_contributes"] += s != 0.0
        worst["disagree_on_contributing"] += (f != 0.0) != (s != 0.0)
        for name, a, b in (("fast_vs_closed", f, closed), ("ref_vs_closed", s, closed), ("fast_vs_ref", f, s)):
            worst[name] = max(worst[name], abs(a - b))
        assert abs(f - closed) <= atol and abs(s - closed) <= atol and abs(e - closed) <= atol, (label, k, f, s, e, closed, atol)
        if (f != 0.0) != (s != 0.0):      # one formulation counts the voxel in, the other does not: only with a weight that is nothing
            assert max(abs(f), abs(s)) <= atol, (label, k, f, s)
    # whole volumes: the stated tolerance of the tolerance mode (tests/test_fast_gpu.py) holds here too
    np.testing.assert_allclose(got, want_exact, rtol=2e-5, atol=atol)
    np.testing.assert_allclose(got, splat, rtol=2e-5, atol=atol)
    print(f"\nsphere surface {dims} r = {rvox} voxels: {worst['targets']} voxel centres within 4 ulp of a sphere; contributing: fast {worst['fast_contributes']}, "
          f"reference {worst['ref_contributes']}, counted differently {worst['disagree_on_contributing']}; worst |fast - closed form| {worst['fast_vs_closed']:.3e}, "
          f"|reference - closed form| {worst['ref_vs_closed']:.3e}, |fast - reference| {worst['fast_vs_ref']:.3e} against atol {atol:.3e} "
          f"(one full-weight contribution {one:.3e})")
    assert worst["targets"] > 120
