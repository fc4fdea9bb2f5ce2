"""TEST INFRASTRUCTURE ONLY.  ctypes access to the cutting oracle (oracle/cut_oracle.c in libfem_oracle.so) and to the
reference's own Intersections.cpp (oracle/_ref/libcut_ref.so, built by oracle/Makefile when /root/reference exists)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f, _d, _u, _i = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_int)
_orc = None
_ref = None


def _p(a, t):
    return a.ctypes.data_as(t)


def orc():
    global _orc
    if _orc is None:
        L = C.CDLL(os.path.join(_HERE, "libfem_oracle.so"))
        L.orc_cut_faces.restype = C.c_longlong
        L.orc_cut_faces.argtypes = [C.c_int, _d, C.c_longlong, _u, _d, _d, _u, _f]
        L.orc_cut_edges.restype = C.c_longlong
        L.orc_cut_edges.argtypes = [_d, C.c_longlong, _u, _d, _u, _f]
        L.orc_cut_segment_tris.restype = None
        L.orc_cut_segment_tris.argtypes = [C.c_int, _f, _f, _f, _f]
        L.orc_segment_triangle.restype = C.c_int
        L.orc_segment_triangle.argtypes = [_f, _f, _f, _f, _f]
        _orc = L
    return _orc


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libcut_ref.so"))


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libcut_ref.so"))
        L.ref_segment_triangle_f.restype = None
        L.ref_segment_triangle_f.argtypes = [C.c_int, _f, _f, _i, _f, _f]
        L.ref_segment_triangle_d.restype = None
        L.ref_segment_triangle_d.argtypes = [C.c_int, _d, _d, _i, _d, _d]
        _ref = L
    return _ref


def cut_faces(mode, vertices, tets, s0=None, s1=None):
    v = np.ascontiguousarray(vertices, np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(tets, np.uint32).reshape(-1, 4)
    a = np.zeros(3) if s0 is None else np.asarray(s0, np.float64).reshape(3).copy()
    b = np.zeros(3) if s1 is None else np.asarray(s1, np.float64).reshape(3).copy()
    flags, pts = np.empty(4 * len(t), np.uint32), np.empty((4 * len(t), 4), np.float32)
    n = orc().orc_cut_faces(mode, _p(v, _d), len(t), _p(t, _u), _p(a, _d), _p(b, _d), _p(flags, _u), _p(pts, _f))
    return int(n), flags, pts


def cut_edges(vertices, tets, quad):
    v = np.ascontiguousarray(vertices, np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(tets, np.uint32).reshape(-1, 4)
    q = np.ascontiguousarray(quad, np.float64).reshape(12)
    flags, pts = np.empty(6 * len(t), np.uint32), np.empty((6 * len(t), 4), np.float32)
    n = orc().orc_cut_edges(_p(v, _d), len(t), _p(t, _u), _p(q, _d), _p(flags, _u), _p(pts, _f))
    return int(n), flags, pts


def segment_tris(tri_xyzw, s0, s1):
    t = np.ascontiguousarray(tri_xyzw, np.float32).reshape(-1, 12)
    a, b = np.asarray(s0, np.float32).reshape(3).copy(), np.asarray(s1, np.float32).reshape(3).copy()
    out = np.empty((len(t), 4), np.float32)
    orc().orc_cut_segment_tris(len(t), _p(t, _f), _p(a, _f), _p(b, _f), _p(out, _f))
    return out


def segment_triangle_pairs(seg, tri):
    """oracle, pairwise: seg (n, 6), tri (n, 9) float32 -> hit (n,), xyz (n, 3), t (n,)"""
    seg = np.ascontiguousarray(seg, np.float32).reshape(-1, 6)
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    hit, xyz, tt = np.zeros(len(seg), np.int32), np.zeros((len(seg), 3), np.float32), np.zeros(len(seg), np.float32)
    L = orc()
    for i in range(len(seg)):
        x, t1 = (C.c_float * 3)(), C.c_float()
        hit[i] = L.orc_segment_triangle(_p(seg[i, :3], _f), _p(seg[i, 3:], _f), _p(tri[i], _f), x, C.byref(t1))
        if hit[i]:
            xyz[i], tt[i] = list(x), t1.value
    return hit, xyz, tt


def ref_segment_triangle_pairs(seg, tri, double=False):
    dt, fn, pt = (np.float64, ref().ref_segment_triangle_d, _d) if double else (np.float32, ref().ref_segment_triangle_f, _f)
    seg = np.ascontiguousarray(seg, dt).reshape(-1, 6)
    tri = np.ascontiguousarray(tri, dt).reshape(-1, 9)
    hit, xyz, tt = np.zeros(len(seg), np.int32), np.zeros((len(seg), 3), dt), np.zeros(len(seg), dt)
    fn(len(seg), _p(seg, pt), _p(tri, pt), _p(hit, _i), _p(xyz, pt), _p(tt, pt))
    return hit, xyz, tt
