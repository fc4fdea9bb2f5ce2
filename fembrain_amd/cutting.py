"""PS::FEM::Cutting (src/deformable/Cutting.h:34-111, Cutting.cpp) over the HIP library: the scalpel / tet-mesh intersection
passes of the cutting tool and the swept-quad bookkeeping of performCut.  No CPU fallback."""
import ctypes as C

import numpy as np

from . import lib as _l

FB_CUT_FACES, FB_CUT_EDGES = 0, 1


class Cutting:
    MIN_SWEPT_LENGTH = 0.01   # Cutting.cpp:501
    MAX_PATH_NODES = 512      # Cutting.cpp:522

    def __init__(self, vertices, tets, device=0):
        """vertices: (n, 3) current node positions; tets: (m, 4) node ids (Cutting::createMemBuffers, Cutting.cpp:87-167)"""
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(tets, dtype=np.uint32).reshape(-1, 4)
        self._L = _l.lib()
        self.h = C.c_void_p()
        self.n_vertices, self.n_tets = len(v), len(t)
        _l.check(self._L.fb_cut_create(C.byref(self.h), device, len(v), _l.dptr(v), len(t), _l.uptr(t)))
        self.ct_face_points = 0
        self.ct_edge_points = 0
        self.swept_quad = np.zeros((4, 3))
        self.swept_quad_valid = False
        self._path0, self._path1 = [], []

    def close(self):
        if getattr(self, "h", None):
            self._L.fb_cut_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_vertices(self, vertices):
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        _l.check(self._L.fb_cut_set_vertices(self.h, len(v), _l.dptr(v)))

    # Cutting::computeFaceCentroids
    def compute_face_centroids(self):
        _l.check(self._L.fb_cut_face_centroids(self.h))
        return 1

    # Cutting::computeFaceIntersections -> number of face points
    def compute_face_intersections(self, s0, s1):
        a, b = np.asarray(s0, np.float64).reshape(3).copy(), np.asarray(s1, np.float64).reshape(3).copy()
        n = C.c_int()
        _l.check(self._L.fb_cut_face_intersections(self.h, _l.dptr(a), _l.dptr(b), C.byref(n)))
        self.ct_face_points = n.value
        return n.value

    # Cutting::computeEdgeIntersections; returns the number of edge points (the reference returns 1 and leaves the
    # counting to its caller, Cutting.cpp:563-579)
    def compute_edge_intersections(self, swept_quad):
        q = np.ascontiguousarray(swept_quad, dtype=np.float64).reshape(12)
        n = C.c_int()
        _l.check(self._L.fb_cut_edge_intersections(self.h, _l.dptr(q), C.byref(n)))
        self.ct_edge_points = n.value
        return n.value

    def read(self, what):
        k = 4 if what == FB_CUT_FACES else 6
        flags = np.empty(k * self.n_tets, np.uint32)
        pts = np.empty((k * self.n_tets, 4), np.float32)
        _l.check(self._L.fb_cut_read(self.h, what, _l.uptr(flags), _l.fptr(pts)))
        return flags, pts

    def read_hits(self, what):
        n = C.c_int()
        _l.check(self._L.fb_cut_read_hits(self.h, what, 0, None, None, C.byref(n)))
        ids, pts = np.empty(n.value, np.uint32), np.empty((n.value, 4), np.float32)
        if n.value:
            _l.check(self._L.fb_cut_read_hits(self.h, what, n.value, _l.uptr(ids), _l.fptr(pts), C.byref(n)))
        return ids, pts

    # Cutting::performCut(edge0, edge1) (Cutting.cpp:499-535): pick the most recent blade position at least
    # MIN_SWEPT_LENGTH away as the far side of the swept quad, record the new position, cut if a quad was found
    def perform_cut(self, edge0, edge1):
        e0, e1 = np.asarray(edge0, np.float64).reshape(3).copy(), np.asarray(edge1, np.float64).reshape(3).copy()
        self.swept_quad_valid = False
        self.swept_quad[0], self.swept_quad[1] = e0, e1
        if len(self._path0) > 1:
            for i in range(len(self._path0) - 1, -1, -1):
                if np.sqrt(((e0 - self._path0[i]) ** 2).sum()) >= self.MIN_SWEPT_LENGTH:
                    self.swept_quad[2], self.swept_quad[3] = self._path0[i], self._path1[i]
                    self.swept_quad_valid = True
                    break
        self._path0.append(e0)
        self._path1.append(e1)
        if len(self._path0) > self.MAX_PATH_NODES:
            self._path0.pop(0)
        if len(self._path1) > self.MAX_PATH_NODES:
            self._path1.pop(0)
        if self.swept_quad_valid:
            self.perform_cut_quad(e0, e1, self.swept_quad)
        return 1

    # Cutting::performCut(edge0, edge1, sweptQuad) (Cutting.cpp:537-566) with the edge pass the reference has commented out
    def perform_cut_quad(self, edge0, edge1, swept_quad):
        self.ct_face_points = self.ct_edge_points = 0
        self.compute_face_intersections(edge0, edge1)
        self.compute_edge_intersections(swept_quad)
        return self.ct_face_points + self.ct_edge_points

    def time_pass(self, what, a, b=None, reps=20):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
        ms = C.c_double()
        _l.check(self._L.fb_cut_time(self.h, what, _l.dptr(a), _l.dptr(b), reps, C.byref(ms)))
        return ms.value


def segment_triangles(tri_xyzw, s0, s1, device=0):
    """kernel ComputeSegmentTriIntersections: (n, 3, 4) triangles against one segment -> (n, 4) points, (-1,-1,-1,1) = miss"""
    t = np.ascontiguousarray(tri_xyzw, dtype=np.float32).reshape(-1, 12)
    a, b = np.asarray(s0, np.float32).reshape(3).copy(), np.asarray(s1, np.float32).reshape(3).copy()
    out = np.empty((len(t), 4), np.float32)
    _l.check(_l.lib().fb_cut_segment_triangles(device, len(t), _l.fptr(t), _l.fptr(a), _l.fptr(b), _l.fptr(out)))
    return out
