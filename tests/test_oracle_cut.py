"""CPU tests of the cutting oracle: the reference's known answer, agreement with the reference's own Intersections.cpp
(oracle/_ref/libcut_ref.so when present -- built from /root/reference, so in this container only) and golden vectors that
library produced (tests/golden/cut_pairs.npz, made by tests/golden/make_cut_golden.py) for the GPU box."""
import os

import numpy as np
import pytest

from oracle import pycut

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pairs(seed, n):
    """random segment / triangle pairs, about a third of them crossing, with degenerate ones mixed in"""
    rng = np.random.default_rng(seed)
    tri = rng.uniform(-1, 1, size=(n, 9)).astype(np.float32)
    seg = rng.uniform(-1, 1, size=(n, 6)).astype(np.float32)
    # aim half of the segments through their triangle
    w = rng.dirichlet([1, 1, 1], size=n).astype(np.float32)
    inside = (w[:, :, None] * tri.reshape(n, 3, 3)).sum(1)
    half = np.arange(n) % 2 == 0
    seg[half, 3:] = seg[half, :3] + (inside[half] - seg[half, :3]) * rng.uniform(0.5, 2.5, size=(half.sum(), 1)).astype(np.float32)
    seg[::97, 3:] = seg[::97, :3]            # zero-length segments
    tri[::89, 3:6] = tri[::89, 0:3]          # degenerate triangles
    return seg, tri


def test_reference_known_answer():
    """Cutting::computeFaceSegmentIntersectionTest (Cutting.cpp:351-369)"""
    tri = np.float32([[-1, 0, -1, 1], [1, 0, -1, 1], [0, 0, 1, 1]])
    out = pycut.segment_tris(tri[None], [0, -1, 0], [0, 1, 0])
    assert np.array_equal(out, np.float32([[0, 0, 0, 1]]))
    miss = pycut.segment_tris(tri[None], [3, -1, 0], [3, 1, 0])
    assert np.array_equal(miss, np.float32([[-1, -1, -1, 1]]))
    short = pycut.segment_tris(tri[None], [0, -1, 0], [0, -0.5, 0])   # ray hits, segment too short
    assert np.array_equal(short, np.float32([[-1, -1, -1, 1]]))


@pytest.mark.skipif(not pycut.have_ref(), reason="oracle/_ref/libcut_ref.so not built (needs /root/reference)")
def test_oracle_equals_reference_float_code():
    seg, tri = _pairs(11, 20000)
    h, x, t = pycut.segment_triangle_pairs(seg, tri)
    rh, rx, rt = pycut.ref_segment_triangle_pairs(seg, tri)
    assert np.array_equal(h, rh) and 0.2 < h.mean() < 0.7
    assert np.array_equal(x[h == 1], rx[h == 1]) and np.array_equal(t[h == 1], rt[h == 1])
    # and the double version agrees wherever the decision is not marginal
    dh, dx, _ = pycut.ref_segment_triangle_pairs(seg, tri, double=True)
    assert (dh != h).mean() < 2e-3
    both = (dh == 1) & (h == 1)
    assert np.abs(dx[both] - x[both]).max() < 1e-3


def test_oracle_matches_golden_vectors():
    g = np.load(os.path.join(GOLD, "cut_pairs.npz"))
    h, x, t = pycut.segment_triangle_pairs(g["seg"], g["tri"])
    assert np.array_equal(h, g["hit"])
    assert np.array_equal(x[h == 1], g["xyz"][h == 1]) and np.array_equal(t[h == 1], g["t"][h == 1])


def test_face_and_edge_passes_on_a_small_mesh():
    from fembrain_amd.meshgen import truth_cube
    v, tets = truth_cube(4, 4, 4, 0.25)   # x, z in [-0.5, 0.25], y in [0, 0.75]
    n, flags, pts = pycut.cut_faces(0, v, tets)
    assert n == 4 * len(tets) and flags.all() and np.all(pts[:, 3] == 1)
    face0 = v[tets[:, [0, 1, 2]]].astype(np.float32)
    assert np.allclose(pts[0::4, :3], face0.mean(1), atol=1e-6)
    s0, s1 = (0.07, -1.0, -0.09), (0.07, 2.0, -0.09)   # a vertical needle through the cube
    n, flags, pts = pycut.cut_faces(1, v, tets, s0, s1)
    assert n == flags.sum() and n >= 6
    hit = pts[flags == 1]
    assert np.allclose(hit[:, 0], 0.07, atol=1e-5) and np.allclose(hit[:, 2], -0.09, atol=1e-5)
    assert hit[:, 1].min() >= -1e-5 and hit[:, 1].max() <= 0.75 + 1e-5
    quad = [(-1, 0.3, -1), (2, 0.3, -1), (-1, 0.3, 2), (2, 0.3, 2)]   # the plane y = 0.3 across the whole cube
    n, flags, pts = pycut.cut_edges(v, tets, quad)
    assert n == flags.sum() and n > 0
    assert np.allclose(pts[flags == 1][:, 1], 0.3, atol=1e-5)
    assert np.all(pts[flags == 0][:, :3] == 0) and np.all(pts[:, 3] == 1)
    # an edge is cut exactly when its ends lie on different sides of the plane
    ends = v[tets[:, [[0, 1], [1, 2], [2, 0], [0, 3], [1, 3], [2, 3]]]][..., 1]
    crossing = ((ends[..., 0] - 0.3) * (ends[..., 1] - 0.3) < 0).reshape(-1)
    assert np.array_equal(flags.astype(bool), crossing)
