"""Writes tests/golden/cut_pairs.npz: segment / triangle pairs and what the reference's own IntersectSegmentTriangleF
(src/graphics/Intersections.cpp, compiled into oracle/_ref/libcut_ref.so by oracle/Makefile) answers for them.
Run in the build container (needs /root/reference):  python tests/golden/make_cut_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import pycut  # noqa: E402
from test_oracle_cut import _pairs  # noqa: E402

seg, tri = _pairs(2024, 4096)
hit, xyz, t = pycut.ref_segment_triangle_pairs(seg, tri)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "cut_pairs.npz"), seg=seg, tri=tri, hit=hit, xyz=xyz, t=t)
print("pairs", len(seg), "hits", int(hit.sum()))
