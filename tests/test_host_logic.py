"""Host-side logic that runs on the CPU in the reference too: the light-plane fit (E2), the
TF-difference break-point list (C2), the synthetic workloads and the TF LUT.

The product's implementation (host/cpm_hostmath.cpp: what the C++ processors and pipeline.py both call) is written from the
rules; the oracle's (oracle/cpm_oracle_host.c) follows the reference statement by statement.  They must agree bit for bit,
tied hull columns and degenerate transfer functions included."""
import numpy as np
import pytest

from oracle_binding import Oracle


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _random_point_sets(rng):
    """Point sets that reach every branch of the hull rule: general position, lattice points (tied extreme columns, repeated
    points, collinear runs), one column, fewer than four points."""
    for case in range(240):
        kind = case % 6
        n = int(rng.integers(1, 14))
        if kind == 0:
            pts = rng.normal(size=(n, 2))
        elif kind == 1:
            pts = rng.integers(0, 3, size=(n, 2)).astype(np.float64)           # heavy ties
        elif kind == 2:
            pts = rng.integers(0, 5, size=(n, 2)) * 0.25
        elif kind == 3:
            pts = np.stack([np.full(n, 0.75), rng.integers(0, 4, size=n) * 0.5], axis=1)  # a single column
        elif kind == 4:
            corners = np.array([(x, y) for x in (0, 1) for y in (0, 1)], np.float64)      # the cube's corners in pairs
            pts = np.concatenate([corners, corners, rng.random((n % 3, 2))])
        else:
            pts = np.concatenate([rng.random((n, 2)), rng.integers(0, 2, size=(4, 2)).astype(np.float64)])
        yield np.ascontiguousarray(pts, np.float32)


def test_hull_cycle_and_rectangle_equal_the_oracle(cpm, oracle):
    P = cpm.pipeline
    rng = np.random.default_rng(20260501)
    tied = 0
    for pts in _random_point_sets(rng):
        want = oracle.convex_hull_2d(pts)
        got = np.asarray(P.convex_hull_2d(pts), np.float32).reshape(-1, 2)
        assert got.shape == want.shape and np.array_equal(_bits(got), _bits(want)), pts
        xs = np.sort(pts[:, 0])
        tied += len(pts) >= 4 and (xs[0] == xs[1] or xs[-1] == xs[-2])
        r_want = oracle.minimum_bounding_rectangle(want)
        r_got = np.concatenate(P.minimum_bounding_rectangle(got)) if len(got) else np.zeros(6, np.float32)
        assert np.array_equal(_bits(r_got), _bits(r_want)), pts
    assert tied >= 50  # the tied-column rule really was exercised


def test_light_rectangle_equals_the_oracle(cpm, oracle):
    """>= 50 random light directions and positions, axis-parallel and diagonal ones (projected corners coincide: tied columns),
    unit cube and clipped boxes."""
    P, S = cpm.pipeline, cpm.synthetic
    rng = np.random.default_rng(7)
    special = [(0, 0, 1), (0, 0, -1), (1, 0, 0), (0, -1, 0), (1, 1, 0), (0, 1, 1), (1, 0, -1), (1, 1, 1), (-1, 1, -1), (1e-7, 0, 1)]
    cases = [np.asarray(d, np.float64) for d in special] + [rng.normal(size=3) for _ in range(70)]
    for k, d in enumerate(cases):
        n = P._normalize(d)
        lo = rng.random(3) * 0.3 if k % 3 == 2 else np.zeros(3)
        hi = 1 - rng.random(3) * 0.3 if k % 3 == 2 else np.ones(3)
        box = np.array([(x, y, z) for z in (lo[2], hi[2]) for y in (lo[1], hi[1]) for x in (lo[0], hi[0])], np.float32)
        for through in (np.float32([0.5, 0.5, 0.5]) - 2 * n, np.float32(rng.normal(size=3)), np.float32([0.25, -1.0, 0.5])):
            want = oracle.fit_obb(box, through, P._normalize(n))  # fit_plane_aligned_obb normalises what it is given
            got = P.fit_plane_aligned_obb(box, through, n)
            for g, w in zip(got, want):
                assert np.array_equal(_bits(g), _bits(w)), (d, through)


def _random_tf(rng, kind):
    n = int(rng.integers(1, 8))
    if kind % 4 == 0:
        pos = np.sort(rng.integers(0, 9, size=n) / 8.0)       # coinciding positions within and across functions, 0 and 1 included
    else:
        pos = np.sort(rng.random(n))
    col = rng.random((n, 4)).astype(np.float32)
    if kind % 3 == 0:
        col[0, 3] = 0.0                                         # transparent first node (the moved-first-point rule)
    if kind % 5 == 0:
        col[rng.integers(0, n), 3] = 0.0
    return [(float(p), *map(float, c)) for p, c in zip(pos, col)]


def test_tf_difference_equals_the_oracle(cpm, oracle):
    """Random pairs of transfer functions (>= 50 of each flavour): unrelated functions, one node moved / recoloured / added /
    removed, identical functions, single-node functions; plain and associated colours; several epsilons."""
    P, S = cpm.pipeline, cpm.synthetic
    rng = np.random.default_rng(11)
    pairs = []
    for k in range(120):
        pairs.append((_random_tf(rng, k), _random_tf(rng, k + 1)))
    for k in range(120):
        old = _random_tf(rng, k) if k % 2 else list(S.WORKSPACE_TF_POINTS)
        new = [tuple(q) for q in old]
        j = int(rng.integers(0, len(new)))
        edit = k % 5
        if edit == 0:
            lo = new[j - 1][0] if j > 0 else 0.0
            hi = new[j + 1][0] if j + 1 < len(new) else 1.0
            new[j] = (float(lo + (hi - lo) * rng.random()),) + new[j][1:]
        elif edit == 1:
            new[j] = (new[j][0],) + tuple(float(c) for c in rng.random(4))
        elif edit == 2:
            new.append((float(rng.random()),) + tuple(float(c) for c in rng.random(4)))
        elif edit == 3 and len(new) > 1:
            del new[j]
        pairs.append((new, old))
    n_lists = 0
    for k, (new, old) in enumerate(pairs):
        for associated in (False, True):
            eps = (1e-4, 1e-2, 0.3)[k % 3]
            want_pos, want_col = oracle.tf_difference_points(new, old, eps, associated)
            got_pos, got_col = P.tf_difference_points(new, old, eps, associated)
            assert got_pos.shape == want_pos.shape, (new, old)
            assert np.array_equal(_bits(got_pos), _bits(want_pos)) and np.array_equal(_bits(got_col), _bits(want_col)), (new, old)
            n_lists += len(got_pos) > 2
    assert n_lists >= 100
    # both empty: the reference's two-point list; exactly one empty: no difference function
    pos, col = P.tf_difference_points([], [])
    assert pos.tolist() == [0, 0] and not col.any() and oracle.tf_difference_points([], [])[0].tolist() == [0, 0]
    with pytest.raises(ValueError):
        P.tf_difference_points([], list(S.WORKSPACE_TF_POINTS))
    assert oracle.tf_difference_points(list(S.WORKSPACE_TF_POINTS), [])[0] is None


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
    f32 = lambda q: (float(np.float32(q[0])), float(np.float32(q[1])))
    hull = [f32(q) for q in P.convex_hull_2d(pts)]
    assert set(hull) == {f32(q) for q in [(0.0, 0.1), (0.3, -0.4), (0.9, 0.05), (1.0, 0.6), (0.55, 1.0), (0.1, 0.8)]}
    # counter-clockwise
    area2 = sum(hull[i][0] * hull[(i + 1) % len(hull)][1] - hull[(i + 1) % len(hull)][0] * hull[i][1] for i in range(len(hull)))
    assert area2 > 0
    assert [f32(q) for q in P.convex_hull_2d([(0, 0), (1, 1), (2, 0)])] == [(0, 0), (1, 1), (2, 0)]
    line = [f32(q) for q in P.convex_hull_2d([(0, 0), (0, 1), (0, 2), (0, 3)])]
    assert line[0] == (0, 0) and (0, 3) in line
    # Rule H's tied columns (the reference's behaviour, ref lightcl/convexhull2d.cpp:84-127): with several points on the
    # extreme x columns the cycle misses the (min x, max y) corner; the minimum rectangle
    # of what remains still covers an axis-aligned square (test_convex_hull_and_obb_axis_aligned).
    sq = [f32(q) for q in P.convex_hull_2d([(0, 0), (1, 0), (1, 1), (0, 1), (0.5, 0.5), (0.2, 0.7)])]
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
