"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/field_oracle.c (libfem_oracle.so).

``OrcPoly`` runs the reference pipeline pass by pass on the host (sweep, edge table, cell configs, included cells
with scatter-marked vertices, exclusive scans, vertex compaction, 6 tets per cell).  Never import from fembrain_amd/.
"""
import ctypes as C

import numpy as np

from .pyoracle import _load

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint)
_bp = C.POINTER(C.c_ubyte)


def _f(a):
    return a.ctypes.data_as(_fp)


def _lib():
    lib = _load("orc")
    if getattr(lib, "_field_bound", False):
        return lib
    tree = [C.c_int, _fp, C.c_int, _fp, C.c_int, _fp]
    lib.orc_field_array.argtypes = tree + [C.c_int, _fp]
    lib.orc_sweep.argtypes = tree + [_fp, C.c_float, _ip, _fp]
    lib.orc_sweep_f.argtypes = tree + [_fp, C.c_float, _ip, C.c_int, C.c_int, _fp]
    lib.orc_edge_table.argtypes = [_fp, _ip, _up, _bp]
    lib.orc_cell_configs.argtypes = [_fp, _ip, _bp]
    lib.orc_tet_cells.argtypes = [_bp, _ip, _up, _up]
    lib.orc_scan.restype = C.c_uint
    lib.orc_scan.argtypes = [_up, _up, C.c_size_t]
    lib.orc_tet_vertices.argtypes = [_fp, _up, _up, C.c_size_t, _fp]
    lib.orc_tet_elements.argtypes = [_up, _up, _up, _ip, _up]
    lib.orc_field_color_array.argtypes = tree + [C.c_int, _fp, _fp]
    lib.orc_cube_table.argtypes = [_bp, _bp]
    lib.orc_vertex_attribs.argtypes = tree + [_fp, _ip, _up, _bp, _up, _fp, _fp]
    lib.orc_cell_elements.argtypes = [_bp, _ip, _bp, _bp, _up, _up, _bp, _up]
    lib.orc_set_field_mode.argtypes = [C.c_int, C.c_int, _fp]
    lib.orc_cl_has_route.argtypes = [C.c_int, _fp]
    lib._field_bound = True
    return lib


FIELD_CPU, FIELD_CPU_BOX, FIELD_OPENCL = 0, 1, 2


class field_mode:
    """``with field_mode(FIELD_OPENCL): ...`` -- every oracle field evaluation inside follows that reference path (see the
    header of field_oracle.c); FIELD_CPU_BOX needs the blob's primitive boxes (``blob.prim_boxes``, n x 6)."""

    def __init__(self, mode, blob=None):
        self.mode, self.blob = mode, blob

    def __enter__(self):
        box = None
        if self.mode == FIELD_CPU_BOX:
            box = np.ascontiguousarray(self.blob.prim_boxes, np.float32).reshape(-1, 6)
        _lib().orc_set_field_mode(self.mode, 0 if box is None else len(box), None if box is None else _f(box))
        return self

    def __exit__(self, *a):
        _lib().orc_set_field_mode(FIELD_CPU, 0, None)


def cl_has_route(blob):
    ops = np.ascontiguousarray(blob.ops, np.float32)
    return bool(_lib().orc_cl_has_route(blob.n_ops, _f(ops))) if blob.n_ops else True


def cube_table():
    tri, nvert = np.empty((256, 16), np.uint8), np.empty(256, np.uint8)
    _lib().orc_cube_table(tri.ctypes.data_as(_bp), nvert.ctypes.data_as(_bp))
    return tri, nvert


class OrcPoly:
    def __init__(self, blob):
        self.blob = blob
        self.lib = _lib()
        ops = blob.ops if blob.n_ops else np.zeros((1, 16), np.float32)
        self._ops = np.ascontiguousarray(ops, np.float32)
        self._tree = (blob.n_ops, _f(self._ops), blob.n_prims, _f(blob.prims), len(blob.mtx), _f(blob.mtx))

    def field_array(self, xyzf):
        a = np.ascontiguousarray(xyzf, dtype=np.float32).reshape(-1, 4).copy()
        self.lib.orc_field_array(*self._tree, len(a), _f(a))
        return a

    def field_color_array(self, xyzf):
        """FieldComputer::fieldValueAndColor at n points: (xyzf with the field filled in, rgb)."""
        a = np.ascontiguousarray(xyzf, dtype=np.float32).reshape(-1, 4).copy()
        rgb = np.empty((len(a), 3), np.float32)
        self.lib.orc_field_color_array(*self._tree, len(a), _f(a), _f(rgb))
        return a, rgb

    def grid_dims(self, cellsize):
        lo, hi = self.blob.bbox
        ext = (hi - lo).astype(np.float32)
        return tuple(int(np.ceil(np.float32(e) / np.float32(cellsize))) + 2 for e in ext)

    def sweep_grid(self, lower, cellsize, dims):
        self.lo = np.asarray(lower, np.float32)
        self.g = np.asarray(dims, np.int32)
        self.cellsize = cellsize
        n = int(np.prod(self.g.astype(np.int64)))
        self.xyzf = np.empty((n, 4), np.float32)
        self.lib.orc_sweep(*self._tree, _f(self.lo), cellsize, self.g.ctypes.data_as(_ip), _f(self.xyzf))
        return self.xyzf

    def sweep(self, cellsize):
        return self.sweep_grid(self.blob.bbox[0], cellsize, self.grid_dims(cellsize))

    def sweep_f_slab(self, lower, cellsize, dims, z0, z1):
        lo, g = np.asarray(lower, np.float32), np.asarray(dims, np.int32)
        f = np.empty(int(g[0]) * int(g[1]) * (z1 - z0), np.float32)
        self.lib.orc_sweep_f(*self._tree, _f(lo), cellsize, g.ctypes.data_as(_ip), z0, z1, _f(f))
        return f

    def classify(self):
        g = self.g
        npts = len(self.xyzf)
        ncells = int((g[0] - 1) * (g[1] - 1) * (g[2] - 1))
        gp = g.ctypes.data_as(_ip)
        self.edge_count, self.edge_flags = np.empty(npts, np.uint32), np.empty(npts, np.uint8)
        self.lib.orc_edge_table(_f(self.xyzf), gp, self.edge_count.ctypes.data_as(_up), self.edge_flags.ctypes.data_as(_bp))
        self.config = np.empty(ncells, np.uint8)
        self.lib.orc_cell_configs(_f(self.xyzf), gp, self.config.ctypes.data_as(_bp))
        self.inc_cells, self.inc_verts = np.empty(ncells, np.uint32), np.empty(npts, np.uint32)
        self.lib.orc_tet_cells(self.config.ctypes.data_as(_bp), gp, self.inc_cells.ctypes.data_as(_up), self.inc_verts.ctypes.data_as(_up))
        return {"n_crossed_edges": int(self.edge_count.sum()), "n_surface_cells": int(((self.config != 0) & (self.config != 255)).sum()),
                "n_included_cells": int(self.inc_cells.sum()), "n_tet_vertices": int(self.inc_verts.sum())}

    def tetrahedralize(self):
        npts, ncells = len(self.xyzf), len(self.config)
        voff, coff = np.empty(npts, np.uint32), np.empty(ncells, np.uint32)
        nv = self.lib.orc_scan(self.inc_verts.ctypes.data_as(_up), voff.ctypes.data_as(_up), npts)
        nc = self.lib.orc_scan(self.inc_cells.ctypes.data_as(_up), coff.ctypes.data_as(_up), ncells)
        xyz = np.empty((nv, 3), np.float32)
        tets = np.empty((6 * nc, 4), np.uint32)
        self.lib.orc_tet_vertices(_f(self.xyzf), self.inc_verts.ctypes.data_as(_up), voff.ctypes.data_as(_up), npts, _f(xyz))
        self.lib.orc_tet_elements(voff.ctypes.data_as(_up), coff.ctypes.data_as(_up), self.inc_cells.ctypes.data_as(_up),
                                  self.g.ctypes.data_as(_ip), tets.ctypes.data_as(_up))
        return xyz, tets

    def surface(self):
        """GPUPoly::run steps 3,4,6,7 after classify(): (xyz, normals, triangles)."""
        npts, ncells = len(self.xyzf), len(self.config)
        gp = self.g.ctypes.data_as(_ip)
        tri, nvert = cube_table()
        eoff, coff = np.empty(npts, np.uint32), np.empty(ncells, np.uint32)
        nv = self.lib.orc_scan(self.edge_count.ctypes.data_as(_up), eoff.ctypes.data_as(_up), npts)
        per_cell = nvert[self.config].astype(np.uint32)
        ni = self.lib.orc_scan(per_cell.ctypes.data_as(_up), coff.ctypes.data_as(_up), ncells)
        pos, nrm, idx = np.empty((nv, 3), np.float32), np.empty((nv, 3), np.float32), np.empty(ni, np.uint32)
        self.lib.orc_vertex_attribs(*self._tree, _f(self.xyzf), gp, self.edge_count.ctypes.data_as(_up), self.edge_flags.ctypes.data_as(_bp),
                                    eoff.ctypes.data_as(_up), _f(pos), _f(nrm))
        self.lib.orc_cell_elements(self.config.ctypes.data_as(_bp), gp, tri.ctypes.data_as(_bp), nvert.ctypes.data_as(_bp), coff.ctypes.data_as(_up),
                                   eoff.ctypes.data_as(_up), self.edge_flags.ctypes.data_as(_bp), idx.ctypes.data_as(_up))
        return pos, nrm, idx.reshape(-1, 3)

    def surface_binding(self):
        """Per surface vertex (output order: grid point, then X, Y, Z edge): tet-mesh ids of the edge's two grid points
        (exclusive scan of the included-vertex marks, Tetrahedralizer.cl:39-64) and the root weight
        t = (0.5 - fa) / (fb - fa) of ComputeVertexAttribs (Polygonizer.cl:1540-1543), fp32."""
        gx, gy = int(self.g[0]), int(self.g[1])
        voff = np.cumsum(self.inc_verts, dtype=np.int64) - self.inc_verts
        pts, axes = [], []
        has = np.nonzero(self.edge_count)[0]
        for axis, bit in ((0, 4), (1, 2), (2, 1)):
            sel = has[(self.edge_flags[has] & bit) != 0]
            pts.append(sel)
            axes.append(np.full(len(sel), axis))
        pts, axes = np.concatenate(pts), np.concatenate(axes)
        order = np.lexsort((axes, pts))
        pts, axes = pts[order], axes[order]
        nb = pts + np.array([1, gx, gx * gy], np.int64)[axes]
        fa, fb = self.xyzf[pts, 3], self.xyzf[nb, 3]
        t = (np.float32(0.5) - fa) / (fb - fa)
        return np.stack([voff[pts], voff[nb]], 1).astype(np.uint32), t.astype(np.float32)

    def interpolate_displacements(self, rest, tet_displacements):
        pairs, t = self.surface_binding()
        u = np.asarray(tet_displacements, np.float64).reshape(-1, 3).astype(np.float32)
        da, db = u[pairs[:, 0]], u[pairs[:, 1]]
        return rest + (da + t[:, None] * (db - da))

    def run_tetrahedralizer(self, cellsize):
        self.sweep(cellsize)
        counts = self.classify()
        xyz, tets = self.tetrahedralize()
        return xyz, tets, counts
