"""GPU parity tests of the cutting tool's intersection passes (fembrain_amd/csrc/cut.hip through the C ABI) against the
oracle (oracle/cut_oracle.c, itself pinned to the reference's Intersections.cpp by tests/test_oracle_cut.py).  fp32 work,
compared bit for bit."""
import os

import numpy as np
import pytest

from fembrain_amd import lib as _l
from fembrain_amd.cutting import FB_CUT_EDGES, FB_CUT_FACES, Cutting, segment_triangles
from fembrain_amd.meshgen import truth_cube
from oracle import pycut

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _deformed_cube(n, seed):
    v, t = truth_cube(n, n, n, 1.0 / (n - 1))
    rng = np.random.default_rng(seed)
    return v + rng.normal(scale=0.15 / (n - 1), size=v.shape), t


def test_reference_known_answer(gpu):
    """Cutting::computeFaceSegmentIntersectionTest (Cutting.cpp:351-369)"""
    tri = np.float32([[[-1, 0, -1, 1], [1, 0, -1, 1], [0, 0, 1, 1]]])
    assert np.array_equal(segment_triangles(tri, [0, -1, 0], [0, 1, 0]), np.float32([[0, 0, 0, 1]]))
    assert np.array_equal(segment_triangles(tri, [3, -1, 0], [3, 1, 0]), np.float32([[-1, -1, -1, 1]]))
    assert segment_triangles(np.zeros((0, 3, 4), np.float32), [0, 0, 0], [1, 1, 1]).shape == (0, 4)


def test_segment_triangles_match_reference_vectors(gpu):
    """the golden pairs the reference's IntersectSegmentTriangleF answered (tests/golden/make_cut_golden.py), pair by pair"""
    g = np.load(os.path.join(GOLD, "cut_pairs.npz"))
    tri4 = np.ones((len(g["tri"]), 3, 4), np.float32)
    tri4[:, :, :3] = g["tri"].reshape(-1, 3, 3)
    for i in range(0, len(tri4), 37):
        out = segment_triangles(tri4[i:i + 1], g["seg"][i, :3], g["seg"][i, 3:])
        want = np.float32([*g["xyz"][i], 1]) if g["hit"][i] else np.float32([-1, -1, -1, 1])
        assert np.array_equal(out[0], want), i
    # and every triangle against a few of the segments, versus the oracle
    for i in (0, 2, 100, 1001):
        got = segment_triangles(tri4, g["seg"][i, :3], g["seg"][i, 3:])
        assert np.array_equal(got, pycut.segment_tris(tri4, g["seg"][i, :3], g["seg"][i, 3:]))


@pytest.mark.parametrize("n,seed", [(2, 0), (5, 1), (12, 2), (23, 3)])
def test_face_and_edge_passes_match_oracle(gpu, n, seed):
    v, t = _deformed_cube(n, seed)
    c = Cutting(v, t)
    rng = np.random.default_rng(seed)
    c.compute_face_centroids()
    flags, pts = c.read(FB_CUT_FACES)
    _, oflags, opts = pycut.cut_faces(0, v, t)
    assert np.array_equal(flags, oflags) and np.array_equal(pts, opts)
    total = 0
    for _ in range(6):
        s0 = rng.uniform(-0.7, 0.7, 3) + [0, 0.5, 0]
        s1 = s0 + rng.normal(size=3) * rng.uniform(0.05, 1.5)
        cnt = c.compute_face_intersections(s0, s1)
        flags, pts = c.read(FB_CUT_FACES)
        on, oflags, opts = pycut.cut_faces(1, v, t, s0, s1)
        assert cnt == on == int(flags.sum())
        assert np.array_equal(flags, oflags) and np.array_equal(pts, opts)
        ids, hp = c.read_hits(FB_CUT_FACES)
        assert np.array_equal(ids, np.nonzero(flags)[0]) and np.array_equal(hp, pts[flags == 1])
        total += cnt
        quad = np.stack([s0, s1, s0 + [0.3, 0.1, 0.2], s1 + [0.25, 0.1, 0.3]])
        cnt = c.compute_edge_intersections(quad)
        flags, pts = c.read(FB_CUT_EDGES)
        on, oflags, opts = pycut.cut_edges(v, t, quad)
        assert cnt == on == int(flags.sum())
        assert np.array_equal(flags, oflags) and np.array_equal(pts, opts)
        ids, hp = c.read_hits(FB_CUT_EDGES)
        assert np.array_equal(ids, np.nonzero(flags)[0]) and np.array_equal(hp, pts[flags == 1])
        total += cnt
    assert n == 2 or total > 0


def test_moved_vertices_are_used(gpu):
    v, t = _deformed_cube(6, 9)
    c = Cutting(v, t)
    s0, s1 = (0.01, -1.0, 0.02), (0.01, 2.0, 0.02)
    a = c.compute_face_intersections(s0, s1)
    assert a > 0
    c.set_vertices(v + [5.0, 0, 0])
    assert c.compute_face_intersections(s0, s1) == 0
    with pytest.raises(_l.FbError):
        c.set_vertices(v[:-1])


def test_million_tet_mesh_properties(gpu):
    """BASELINE config 4's mesh (998,250 tets): size-independent checks -- count = number of flags = length of the ordered
    hit list; every face hit lies on the needle, inside the mesh; an edge is cut by a plane-sized quad exactly when its
    ends are on different sides."""
    v, t = truth_cube(56, 56, 56, 0.1)
    c = Cutting(v, t)
    s0, s1 = np.array([0.013, -1.0, 0.021]), np.array([0.013, 7.0, 0.021])
    cnt = c.compute_face_intersections(s0, s1)
    flags, pts = c.read(FB_CUT_FACES)
    ids, hp = c.read_hits(FB_CUT_FACES)
    assert cnt == int(flags.sum()) == len(ids) and cnt >= 2 * 55
    assert np.all(np.diff(ids.astype(np.int64)) > 0) and np.array_equal(ids, np.nonzero(flags)[0])
    assert np.allclose(hp[:, 0], 0.013, atol=2e-5) and np.allclose(hp[:, 2], 0.021, atol=2e-5)
    assert hp[:, 1].min() >= -1e-5 and hp[:, 1].max() <= 5.5 + 1e-5
    y = 2.7301
    quad = [(-9, y, -8.37), (9.2, y, -9.1), (-8.9, y, 9.3), (9.05, y, 8.77)]   # its diagonal misses the mesh's x = z edges
    cnt = c.compute_edge_intersections(quad)
    flags, pts = c.read(FB_CUT_EDGES)
    assert cnt == int(flags.sum())
    ends = v[t[:, [[0, 1], [1, 2], [2, 0], [0, 3], [1, 3], [2, 3]]]][..., 1].astype(np.float32)
    crossing = ((ends[..., 0] - np.float32(y)) * (ends[..., 1] - np.float32(y)) < 0).reshape(-1)
    assert np.array_equal(flags.astype(bool), crossing)
    assert np.allclose(pts[flags == 1][:, 1], y, atol=2e-5)
    ms_f = c.time_pass(FB_CUT_FACES, s0, s1)
    ms_e = c.time_pass(FB_CUT_EDGES, np.asarray(quad, float))
    print("cut passes at 998,250 tets: faces %.1f us, edges %.1f us" % (ms_f * 1e3, ms_e * 1e3))
    assert c.ct_edge_points == cnt   # the timing leaves the state of one clean pass


def test_perform_cut_path_logic(gpu):
    """Cutting::performCut (Cutting.cpp:499-535): no quad until the blade has moved 0.01 from a recorded position (and two
    positions are on record); then the most recent far-enough position closes the quad"""
    v, t = truth_cube(8, 8, 8, 0.1)
    c = Cutting(v, t)
    e1 = np.array([0.0, 0.35, 1.0])
    c.perform_cut([-0.2, 0.35, -1.0], [-0.2, 0.35, 1.0])
    assert not c.swept_quad_valid
    c.perform_cut([-0.195, 0.35, -1.0], [-0.195, 0.35, 1.0])   # one position on record is not enough (size() > 1)
    assert not c.swept_quad_valid
    c.perform_cut([-0.19, 0.35, -1.0], [-0.19, 0.35, 1.0])     # 0.005 from the last, 0.01 from the first
    assert c.swept_quad_valid and np.allclose(c.swept_quad[2], [-0.2, 0.35, -1.0])
    c.perform_cut([0.1, 0.35, -1.0], [0.1, 0.35, 1.0])
    assert c.swept_quad_valid and np.allclose(c.swept_quad[2], [-0.19, 0.35, -1.0]) and np.allclose(c.swept_quad[3], [-0.19, 0.35, 1.0])
    assert c.ct_edge_points > 0    # the quad from x = -0.19 to 0.1 at y = 0.35 crosses the vertical edges of the cube
    n, oflags, _ = pycut.cut_edges(v, t, c.swept_quad)
    assert n == c.ct_edge_points
    for _ in range(600):
        c.perform_cut(e1 * 0, e1)
    assert len(c._path0) == Cutting.MAX_PATH_NODES == len(c._path1)


def test_unstructured_mesh_and_many_blades(gpu):
    """Delaunay tetrahedra of random points (tets of every shape and orientation, shuffled), 40 random scalpel edges and
    swept quads: flags, points, counts and hit lists against the oracle, bit for bit"""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(123)
    v = rng.uniform(-1, 1, size=(4000, 3))
    t = Delaunay(v).simplices.astype(np.uint32)
    rng.shuffle(t)
    c = Cutting(v, t)
    seen_faces = seen_edges = 0
    for k in range(40):
        s0 = rng.uniform(-1.2, 1.2, 3)
        s1 = s0 + rng.normal(size=3) * rng.choice([0.02, 0.3, 2.5])
        n = c.compute_face_intersections(s0, s1)
        flags, pts = c.read(FB_CUT_FACES)
        on, oflags, opts = pycut.cut_faces(1, v, t, s0, s1)
        assert n == on and np.array_equal(flags, oflags) and np.array_equal(pts, opts), k
        quad = np.stack([s0, s1, s0 + rng.normal(size=3) * 0.4, s1 + rng.normal(size=3) * 0.4])   # not planar in general
        n = c.compute_edge_intersections(quad)
        flags, pts = c.read(FB_CUT_EDGES)
        on, oflags, opts = pycut.cut_edges(v, t, quad)
        assert n == on and np.array_equal(flags, oflags) and np.array_equal(pts, opts), k
        ids, hp = c.read_hits(FB_CUT_EDGES)
        assert np.array_equal(ids, np.nonzero(flags)[0]) and np.array_equal(hp, pts[flags == 1])
        seen_faces += int(oflags.sum() > 0)
        seen_edges += n
    assert seen_edges > 1000
