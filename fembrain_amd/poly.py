"""Host-side mirror of the reference's polygonizer surface over the C ABI.

``GpuPoly`` follows ``PS::SKETCH::GPUPoly`` (reference src/implicit/OclPolygonizer.h:45-231) for the field /
classification / tetrahedralizer part of the class and ``PS::SKETCH::FieldComputer``
(src/implicit/FieldComputer.h:31-70) for the voxel-grid sweep; ``run_tetrahedralizer`` returns what
``GPUPoly::runTetrahedralizer`` + ``storeTetMeshInVegaFormat`` (OclPolygonizer.cpp:762-819,1651-1694) produce.
"""
import ctypes as C

import numpy as np

from . import lib as _l
from .blobtree import BlobTree, read_blob, sphere_blob  # noqa: F401


MESH_SURFACE, MESH_TET = 0, 1
# which of the reference's two field evaluations a handle follows (fembrain_hip.h: fb_poly_set_field_semantics)
FIELD_CPU, FIELD_CPU_BOX, FIELD_OPENCL = 0, 1, 2


def cube_table():
    """The marching-cubes triangle table (256 x 16 edge ids, 255 = end) and per-configuration index counts."""
    tri, nvert = np.empty((256, 16), np.uint8), np.empty(256, np.uint8)
    _l.check(_l.lib().fb_poly_cube_table(_l.bptr(tri), _l.bptr(nvert)))
    return tri, nvert


class GpuPoly:
    DEFAULT_CELL_SIZE = 0.14  # reference src/implicit/Polygonizer.h

    def __init__(self, blob, device=0):
        if not isinstance(blob, BlobTree):
            blob = read_blob(blob)
        self.blob = blob
        self._L = _l.lib()
        self.h = C.c_void_p()
        _l.check(self._L.fb_poly_create(C.byref(self.h), device, _l.fptr(blob.header), blob.n_ops, _l.fptr(blob.ops), blob.n_prims,
                                        _l.fptr(blob.prims), len(blob.mtx), _l.fptr(blob.mtx)))
        self.counts = None
        self.dims = None

    def close(self):
        if getattr(self, "h", None):
            self._L.fb_poly_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_field_semantics(self, semantics):
        """FIELD_CPU (default), FIELD_CPU_BOX (CPU path with its primitive box cull; boxes from the reader) or FIELD_OPENCL
        (the OpenCL kernels' evaluation -- what the reference's shipped .veg outputs were made with)."""
        box = None
        if semantics == FIELD_CPU_BOX:
            if self.blob.prim_boxes is None:
                raise ValueError("FIELD_CPU_BOX needs the primitive boxes of the reader (BlobTree.prim_boxes)")
            box = np.ascontiguousarray(self.blob.prim_boxes, dtype=np.float32).reshape(-1, 6)
        _l.check(self._L.fb_poly_set_field_semantics(self.h, semantics, None if box is None else _l.fptr(box)))
        self.counts = self.dims = None

    # GPUPoly::computeFieldArray / FieldComputer::field(n, 4, xyzf)
    def compute_field_array(self, xyzf):
        a = np.ascontiguousarray(xyzf, dtype=np.float32).reshape(-1, 4).copy()
        _l.check(self._L.fb_poly_field_array(self.h, len(a), _l.fptr(a)))
        return a

    def field(self, p):
        return float(self.compute_field_array([[p[0], p[1], p[2], 0.0]])[0, 3])

    # GPUPoly::computeAllFields / FieldComputer::fieldsForVoxelGrid
    def sweep(self, cellsize=DEFAULT_CELL_SIZE):
        d = (C.c_int * 3)()
        _l.check(self._L.fb_poly_sweep(self.h, cellsize, d))
        self.dims = tuple(d)
        return self.dims

    def sweep_grid(self, lower, cellsize, dims):
        lo = np.asarray(lower, dtype=np.float32)
        dm = np.asarray(dims, dtype=np.int32)
        _l.check(self._L.fb_poly_sweep_grid(self.h, _l.fptr(lo), cellsize, _l.iptr(dm)))
        self.dims = tuple(int(x) for x in dm)
        return self.dims

    # ---- multi-GPU field path: one z-slab of a larger grid (fembrain_hip.h, "z-slabs of one grid") ----
    def sweep_slab(self, lower, cellsize, dims, z_first, z_count):
        lo = np.asarray(lower, dtype=np.float32)
        dm = np.asarray(dims, dtype=np.int32)
        _l.check(self._L.fb_poly_sweep_slab(self.h, _l.fptr(lo), cellsize, _l.iptr(dm), z_first, z_count))
        self.dims = (int(dm[0]), int(dm[1]), int(z_count))
        return self.dims

    def slab_counts(self, own_first_plane, own_planes, own_layers):
        a, b = C.c_int(), C.c_int()
        _l.check(self._L.fb_poly_slab_counts(self.h, own_first_plane, own_planes, own_layers, C.byref(a), C.byref(b)))
        return a.value, b.value

    def read_tetmesh_slab(self, own_first_plane, own_planes, own_layers, vertex_base):
        nv, nt = self.slab_counts(own_first_plane, own_planes, own_layers)
        xyz, tets = np.empty((nv, 3), np.float32), np.empty((nt, 4), np.uint32)
        _l.check(self._L.fb_poly_read_tetmesh_slab(self.h, own_first_plane, own_planes, own_layers, vertex_base, _l.fptr(xyz), _l.uptr(tets)))
        return xyz, tets

    def run_tetrahedralizer_slab(self, lower, cellsize, dims, rank, world, allgather):
        """This rank's consecutive piece (vertices, tets with whole-grid vertex numbers) of the tet mesh of the grid `dims`.
        allgather(int) -> list of every rank's int (e.g. torch.distributed.all_gather_object)."""
        p0, p1, z_first, z_count, own_planes, own_layers = slab_plan(int(dims[2]), world, rank)
        self.sweep_slab(lower, cellsize, dims, z_first, z_count)
        self.classify()
        self.tetrahedralize()
        nv, _ = self.slab_counts(p0, own_planes, own_layers)
        base = sum(allgather(nv)[:rank])
        return self.read_tetmesh_slab(p0, own_planes, own_layers, base)

    # GPUPoly::readBackVoxelGridSamples
    def read_grid(self):
        n = self.dims[0] * self.dims[1] * self.dims[2]
        out = np.empty((n, 4), np.float32)
        _l.check(self._L.fb_poly_read_grid(self.h, _l.fptr(out)))
        return out

    def classify(self):
        c = _l.PolyCounts()
        _l.check(self._L.fb_poly_classify(self.h, C.byref(c)))
        self.counts = c
        return c

    def read_classification(self):
        npts, ncells = self.counts.n_points, self.counts.n_cells
        flags, cnt, cfg = np.empty(npts, np.uint8), np.empty(npts, np.uint32), np.empty(ncells, np.uint8)
        _l.check(self._L.fb_poly_read_classification(self.h, _l.bptr(flags), _l.uptr(cnt), _l.bptr(cfg)))
        return flags, cnt, cfg

    def tetrahedralize(self):
        c = _l.PolyCounts()
        _l.check(self._L.fb_poly_tetrahedralize(self.h, C.byref(c)))
        self.counts = c
        return c

    def read_tetmesh(self):
        xyz = np.empty((self.counts.n_tet_vertices, 3), np.float32)
        tets = np.empty((self.counts.n_tets, 4), np.uint32)
        _l.check(self._L.fb_poly_read_tetmesh(self.h, _l.fptr(xyz), _l.uptr(tets)))
        return xyz, tets

    # GPUPoly::runTetrahedralizer
    def run_tetrahedralizer(self, cellsize=DEFAULT_CELL_SIZE):
        self.sweep(cellsize)
        self.classify()
        self.tetrahedralize()
        return self.read_tetmesh()

    # GPUPoly::run (OclPolygonizer.cpp:616-760): marching-cubes surface of the current BlobTree
    def surface(self):
        c = _l.PolyCounts()
        _l.check(self._L.fb_poly_surface(self.h, C.byref(c)))
        self.counts = c
        return c

    # GPUPoly::readbackMeshV3T3
    def read_surface(self):
        nv, ni = self.counts.n_surface_vertices, self.counts.n_surface_indices
        xyz, nrm, idx = np.empty((nv, 3), np.float32), np.empty((nv, 3), np.float32), np.empty((ni // 3, 3), np.uint32)
        _l.check(self._L.fb_poly_read_surface(self.h, _l.fptr(xyz), _l.fptr(nrm), _l.uptr(idx)))
        return xyz, nrm, idx

    def run(self, cellsize=DEFAULT_CELL_SIZE):
        self.sweep(cellsize)
        self.classify()
        self.surface()
        return self.read_surface()

    # GPUPoly::applyFemDisplacements
    def apply_fem_displacements(self, displacements, mesh=MESH_SURFACE):
        u = np.ascontiguousarray(displacements, dtype=np.float64).reshape(-1)
        out = np.empty((len(u) // 3, 3), np.float32)
        _l.check(self._L.fb_poly_apply_displacements(self.h, mesh, len(u), _l.dptr(u), _l.fptr(out)))
        return out

    def interpolate_displacements(self, tet_displacements):
        """Deformed surface from the displacements of the tet mesh of the same grid (fb_poly_interpolate_displacements)."""
        u = np.ascontiguousarray(tet_displacements, dtype=np.float64).reshape(-1)
        out = np.empty((self.counts.n_surface_vertices, 3), np.float32)
        _l.check(self._L.fb_poly_interpolate_displacements(self.h, len(u), _l.dptr(u), _l.fptr(out)))
        return out

    def read_surface_binding(self):
        nv = self.counts.n_surface_vertices
        pairs, w = np.empty((nv, 2), np.uint32), np.empty(nv, np.float32)
        _l.check(self._L.fb_poly_read_surface_binding(self.h, _l.uptr(pairs), _l.fptr(w)))
        return pairs, w

    def read_surface_colors(self):
        """RGBA per surface vertex (ComputeVertexAttribs' colour output)."""
        out = np.empty((self.counts.n_surface_vertices, 4), np.float32)
        _l.check(self._L.fb_poly_read_surface_colors(self.h, _l.fptr(out)))
        return out

    def field_color_array(self, xyzf):
        """FieldComputer::fieldValueAndColor: (xyzf with the field filled in, rgb)."""
        a = np.ascontiguousarray(xyzf, dtype=np.float32).reshape(-1, 4).copy()
        rgb = np.empty((len(a), 3), np.float32)
        _l.check(self._L.fb_poly_field_color_array(self.h, len(a), _l.fptr(a), _l.fptr(rgb)))
        return a, rgb

    # GPUPoly::computeOffSurfacePointsAndFields
    def compute_off_surface_points_and_fields(self, length):
        out = np.empty((2 * self.counts.n_surface_vertices, 4), np.float32)
        _l.check(self._L.fb_poly_off_surface(self.h, length, _l.fptr(out)))
        return out

    def time_surface(self, reps=5):
        a = C.c_double(0)
        _l.check(self._L.fb_poly_time_surface(self.h, reps, C.byref(a)))
        return a.value

    def time_grid(self, reps=10):
        """seconds per sweep + materialisation of the float4 grid fb_poly_read_grid returns"""
        a = C.c_double(0)
        _l.check(self._L.fb_poly_time_grid(self.h, reps, C.byref(a)))
        return a.value

    def time_pipeline(self, reps=5):
        a, b = C.c_double(0), C.c_double(0)
        _l.check(self._L.fb_poly_time_pipeline(self.h, reps, C.byref(a), C.byref(b)))
        return a.value, b.value

    def time_stages(self, reps=5):
        """device seconds of (sweep, classification + scans, tet-mesh vertices, tet elements, all four) with HIP events between them"""
        out = np.zeros(5)
        _l.check(self._L.fb_poly_time_stages(self.h, reps, _l.dptr(out)))
        return tuple(float(x) for x in out)


def slab_plan(planes, world, rank):
    """z-slab of rank `rank`: owned point planes [p0, p1), the slab to sweep [z_first, z_first + z_count) (one plane below,
    two above, clipped to the grid), and how many planes / cell layers the rank owns."""
    if world < 1 or not 0 <= rank < world or planes < 2 * world:
        raise ValueError("cannot deal %d planes to %d ranks" % (planes, world))
    p0, p1 = planes * rank // world, planes * (rank + 1) // world
    z_first, z_last = max(p0 - 1, 0), min(p1 + 1, planes - 1)
    own_layers = min(p1, planes - 1) - p0
    return p0, p1, z_first, z_last - z_first + 1, p1 - p0, own_layers


def write_veg(path, xyz, tets, rho=1000.0, E=1e7, nu=0.45):
    """``GPUPoly::storeTetMeshInVegaFormat`` (reference src/implicit/OclPolygonizer.cpp:1651-1694): 1-indexed Vega text."""
    with open(path, "w") as f:
        f.write("# Vega Mesh File, Generated by FemBrain.\n# %d vertices, %d elements\n\n*VERTICES\n%d 3 0 0\n" % (len(xyz), len(tets), len(xyz)))
        for i, p in enumerate(xyz):
            f.write("%d %g %g %g\n" % (i + 1, p[0], p[1], p[2]))
        f.write("\n*ELEMENTS\nTET\n%d 4 0\n" % len(tets))
        for i, t in enumerate(tets):
            f.write("%d %d %d %d %d\n" % (i + 1, t[0] + 1, t[1] + 1, t[2] + 1, t[3] + 1))
        f.write("\n*MATERIAL BODY\nENU, %g, %g, %g\n\n*REGION\nallElements, BODY\n" % (rho, E, nu))
