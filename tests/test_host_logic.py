"""Host-side logic that runs on the CPU in the reference too: the light-plane fit (E2), the
TF-difference break-point list (C2), the synthetic workloads and the TF LUT."""
import numpy as np
import pytest


def test_convex_hull_and_obb_axis_aligned(cpm):
    P, S = cpm.pipeline, cpm.synthetic
    # light along +z: the cube projects to the unit square; minimum rectangle = that square
    o, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, np.array([0.5, 0.5, -1.5], np.float32), (0, 0, 1))
    assert abs(np.linalg.norm(u) * np.linalg.norm(v) - 1.0) < 1e-5
    assert abs(np.dot(u, v)) < 1e-6 and abs(u[2]) < 1e-6 and abs(v[2]) < 1e-6
    corners = [o, o + u, o + v, o + u + v]
    xs = sorted(round(float(c[0]), 4) for c in corners)
    assert xs[0] == 0.0 and xs[-1] == 1.0 and abs(o[2] + 1.5) < 1e-6


@pytest.mark.parametrize("direction", [(0.3, 0.5, -1.0), (1, 1, 1), (-1, 0.2, 0.1), (0, 1, 0)])
def test_obb_covers_projected_cube(cpm, direction):
    P, S = cpm.pipeline, cpm.synthetic
    d = P._normalize(direction)
    origin = np.array([0.5, 0.5, 0.5], np.float32) - 2 * d
    o, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
    assert abs(np.dot(u, d)) < 1e-5 and abs(np.dot(v, d)) < 1e-5 and abs(np.dot(u, v)) < 1e-4
    assert abs(np.dot(o - origin, d)) < 1e-5  # the rectangle lies in the light plane
    lu, lv = np.linalg.norm(u), np.linalg.norm(v)
    for p in S.UNIT_CUBE_VERTICES:
        q = p - np.dot(p - origin, d) * d - o
        a, b = np.dot(q, u) / lu ** 2, np.dot(q, v) / lv ** 2
        assert -1e-4 <= a <= 1 + 1e-4 and -1e-4 <= b <= 1 + 1e-4
    # minimum area: not larger than the axis-aligned bounding rectangle in the (u0, v0) basis
    assert lu * lv <= 3.01  # projected unit cube: area <= sqrt(3) * ... (loose sanity bound)


def test_convex_hull_known_cases(cpm):
    P = cpm.pipeline
    # points in general position (no two share an x): the monotone chain returns the hull, open
    pts = [(0.0, 0.1), (0.3, -0.4), (0.9, 0.05), (1.0, 0.6), (0.55, 1.0), (0.1, 0.8), (0.5, 0.5), (0.4, 0.3)]
    hull = P.convex_hull_2d(pts)
    assert set(hull) == {(0.0, 0.1), (0.3, -0.4), (0.9, 0.05), (1.0, 0.6), (0.55, 1.0), (0.1, 0.8)}
    # counter-clockwise
    area2 = sum(hull[i][0] * hull[(i + 1) % len(hull)][1] - hull[(i + 1) % len(hull)][0] * hull[i][1] for i in range(len(hull)))
    assert area2 > 0
    assert P.convex_hull_2d([(0, 0), (1, 1), (2, 0)]) == [(0, 0), (1, 1), (2, 0)]
    line = P.convex_hull_2d([(0, 0), (0, 1), (0, 2), (0, 3)])
    assert line[0] == (0, 0) and (0, 3) in line
    # Reference behaviour kept as is (ref lightcl/convexhull2d.cpp:84-127): with several points on the
    # extreme x columns the index bookkeeping drops the (min x, max y) corner; the minimum rectangle
    # of what remains still covers an axis-aligned square (test_convex_hull_and_obb_axis_aligned).
    sq = P.convex_hull_2d([(0, 0), (1, 0), (1, 1), (0, 1), (0.5, 0.5), (0.2, 0.7)])
    assert set(sq) == {(0, 0), (1, 0), (1, 1)}


def test_tf_lut_and_workloads(cpm):
    S = cpm.synthetic
    tf = S.workspace_tf()
    assert tf.shape == (1024, 4) and tf.dtype == np.float32
    assert tf[0, 3] == 0 and abs(tf[-1, 3] - 0.53218883) < 1e-6 and (np.diff(tf[:, 3]) >= -1e-7).all()
    i = int(0.2851 * 1024)
    assert abs(tf[i, 3] - np.interp((i + 0.5) / 1024, [p[0] for p in S.WORKSPACE_TF_POINTS], [p[4] for p in S.WORKSPACE_TF_POINTS])) < 1e-6
    v = S.heterogeneous_volume(32)
    assert v.shape == (32, 32, 32) and v.dtype == np.uint8 and v.min() >= 255 // 4 - 1 and v.max() <= 255
    assert np.array_equal(v, S.heterogeneous_volume(32))  # deterministic
    assert (S.homogeneous_volume(8) == 128).all()
    assert abs(S.photon_radius_texture((256, 256, 256)) - 3 ** 0.5 / 256) < 1e-9
    assert S.photon_radius_texture((256, 256, 256)) == float(np.float32(S.photon_radius_texture((256, 256, 256))))
    c0, c31 = S.sequence_blob_center(0), S.sequence_blob_center(31)
    assert abs(c0[0] - 0.3) < 1e-12 and abs(c31[0] - 0.7) < 1e-12


def test_tf_difference_points(cpm):
    P, S = cpm.pipeline, cpm.synthetic
    old = list(S.WORKSPACE_TF_POINTS)
    # unchanged TF: no difference anywhere
    pos, col = P.tf_difference_points(old, old)
    assert pos[0] == 0 and pos[-1] == 1 and not col.any()
    # config 3 edit: point 4 moves 0.2218 -> 0.26: the difference lives between its neighbours
    new = list(old)
    new[3] = (0.26,) + old[3][1:]
    pos, col = P.tf_difference_points(new, old)
    assert pos[0] == 0 and pos[-1] == 1 and (np.diff(pos) >= 0).all()
    assert not col[0].any() and not col[-1].any()
    nz = pos[(col != 0).any(axis=1)]
    assert nz.min() >= old[2][0] - 1e-6 and nz.max() <= old[4][0] + 1e-6
    # the list is |TF_new - TF_old| at its break points
    lut_new, lut_old = S.tf_from_points(new, 4096), S.tf_from_points(old, 4096)
    for p_, c_ in zip(pos, col):
        if 0 < p_ < 1:
            k = min(int(p_ * 4096), 4095)
            assert np.allclose(c_, np.abs(lut_new[k] - lut_old[k]), atol=2e-3)
