"""ctypes binding of libfembrain_hip.so (the C ABI of include/fembrain_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfembrain_hip.so")

FB_OK, FB_EINVAL, FB_EDEVICE, FB_ENOMEM, FB_ESOLVER, FB_ECOMM = 0, -1, -2, -3, -4, -5
FB_MATRIX_F32, FB_MATRIX_F64, FB_MATRIX_AUTO = 0, 1, 2
FB_XCH_COLLECTIVE, FB_XCH_P2P, FB_XCH_P2P_SUMS, FB_XCH_P2P_FUSED = 1, 2, 3, 4
FB_PCG_MERGED, FB_PCG_REFERENCE, FB_PCG_PERSISTENT, FB_PCG_BLOCK_JACOBI = 0, 1, 3, 4  # (2 was an experiment, removed)
FB_PCG_PATH_TWO_LAUNCH, FB_PCG_PATH_PERSISTENT, FB_PCG_PATH_FALLBACK, FB_PCG_PATH_RESOLVED = 0, 1, 2, 3
FB_SPMV_AUTO, FB_SPMV_ROWS, FB_SPMV_SPLIT = 0, 1, 2
FB_INTEGRATOR_VOLUME_CONSERVING, FB_INTEGRATOR_NEWMARK = 0, 1
FB_RENUMBER_AUTO, FB_RENUMBER_ON, FB_RENUMBER_OFF = 0, 1, -1
FB_RESYNC_FULL, FB_RESYNC_DELTA_MERGED, FB_RESYNC_DELTA_REBUILT = 0, 1, 2

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint)
_bp = C.POINTER(C.c_ubyte)


class FbError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("fembrain_hip error %d: %s" % (code, text))
        self.code = code


class FemParams(C.Structure):
    _fields_ = [("E", C.c_double), ("nu", C.c_double), ("rho", C.c_double), ("timestep", C.c_double),
                ("damping_mass", C.c_double), ("damping_stiffness", C.c_double), ("cg_eps", C.c_double),
                ("cg_max_iter", C.c_int), ("matrix_precision", C.c_int), ("device", C.c_int),
                ("pcg_variant", C.c_int), ("spmv_kernel", C.c_int), ("linear", C.c_int), ("exact_tangent", C.c_int), ("integrator", C.c_int), ("renumber", C.c_int),
                ("expect_cuts", C.c_int), ("reserve_nodes", C.c_int), ("reserve_elements", C.c_int)]


class StepInfo(C.Structure):
    _fields_ = [("cg_iterations", C.c_int), ("converged", C.c_int), ("assembly_seconds", C.c_double),
                ("solve_seconds", C.c_double), ("rho0", C.c_double), ("rho", C.c_double), ("pcg_path", C.c_int),
                ("persist_fallbacks", C.c_int), ("newton_iterations", C.c_int)]


class PolyCounts(C.Structure):
    _fields_ = [("grid", C.c_int * 3), ("n_points", C.c_int), ("n_cells", C.c_int), ("n_crossed_edges", C.c_int),
                ("n_surface_cells", C.c_int), ("n_included_cells", C.c_int),
                ("n_tet_vertices", C.c_int), ("n_tets", C.c_int), ("n_surface_vertices", C.c_int),
                ("n_surface_indices", C.c_int)]


def source_sha256(part=None):
    """Hash of the kernel sources (csrc/*.hip, *.h, *.cpp).  A profile under profiles/ records it; bench.py reports PMC traffic
    from that profile only while the kernels are still the ones that were profiled.  part = "fem": everything but the field / cutting
    translation units (poly.hip, cut.hip); "poly": poly.hip and the shared headers it includes -- so that an edit of one half does not
    retire the other half's profile."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if not name.endswith((".hip", ".h", ".cpp")):
            continue
        if part == "fem" and name in ("poly.hip", "cut.hip"):
            continue
        if part == "poly" and name not in ("poly.hip", "common.h"):
            continue
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def build(force=False):
    """Compile libfembrain_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    mk = os.path.join(_HERE, "csrc", "Makefile")
    if force:
        subprocess.check_call(["make", "-s", "-f", mk, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-f", mk])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FbError(FB_EDEVICE, "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "(there is no CPU fallback for the HIP path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.fb_last_error.restype = C.c_char_p
    vp = C.c_void_p
    sig = {
        "fb_device_count": (C.c_int, []),
        "fb_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, _ip]),
        "fb_host_register": (C.c_int, [C.c_void_p, C.c_ulonglong]),
        "fb_host_unregister": (C.c_int, [C.c_void_p]),
        "fb_fem_default_params": (None, [C.POINTER(FemParams)]),
        "fb_fem_create": (C.c_int, [C.POINTER(vp), C.c_int, _dp, C.c_int, _ip, C.c_int, _ip, C.POINTER(FemParams)]),
        "fb_fem_create_sharded": (C.c_int, [C.POINTER(vp), C.c_int, _dp, C.c_int, _ip, C.c_int, _ip, C.POINTER(FemParams),
                                            C.c_int, C.c_int, _ip, vp]),
        "fb_fem_destroy": (C.c_int, [vp]),
        "fb_fem_resync": (C.c_int, [vp, C.c_int, _dp, C.c_int, _ip, C.c_int, _ip]),
        "fb_fem_resync_sharded": (C.c_int, [vp, C.c_int, _dp, C.c_int, _ip, C.c_int, _ip, _ip]),
        "fb_fem_resync_delta": (C.c_int, [vp, C.c_int, _ip, C.c_int, _ip, _ip, C.c_int, _ip, C.c_int, _dp, C.c_int, _ip]),
        "fb_fem_resync_path": (C.c_int, [vp]),
        "fb_fem_rebuild_elements": (C.c_int, [vp]),
        "fb_fem_set_external_forces": (C.c_int, [vp, _dp]),
        "fb_fem_add_external_forces": (C.c_int, [vp, _dp]),
        "fb_fem_set_external_forces_zero": (C.c_int, [vp]),
        "fb_fem_set_uniform_force": (C.c_int, [vp, C.c_int, C.c_double]),
        "fb_fem_step": (C.c_int, [vp, C.POINTER(StepInfo)]),
        "fb_fem_get_state": (C.c_int, [vp, _dp, _dp, _dp]),
        "fb_fem_set_state": (C.c_int, [vp, _dp, _dp, _dp]),
        "fb_fem_reset": (C.c_int, [vp]),
        "fb_fem_set_timestep": (C.c_int, [vp, C.c_double]),
        "fb_fem_set_damping": (C.c_int, [vp, C.c_double, C.c_double]),
        "fb_fem_set_internal_force_scaling": (C.c_int, [vp, C.c_double]),
        "fb_fem_set_cg": (C.c_int, [vp, C.c_double, C.c_int]),
        "fb_fem_set_constrained_dofs": (C.c_int, [vp, C.c_int, _ip]),
        "fb_fem_floor_collision": (C.c_int, [vp, C.c_double, C.c_double, _ip]),
        "fb_fem_num_nodes": (C.c_int, [vp]),
        "fb_fem_num_tets": (C.c_int, [vp]),
        "fb_fem_num_blocks": (C.c_int, [vp]),
        "fb_fem_matrix_precision": (C.c_int, [vp]),
        "fb_fem_owned_range": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "fb_fem_pattern": (C.c_int, [vp, _ip, _ip]),
        "fb_fem_element_stiffness": (C.c_int, [vp, C.c_int, C.c_int, _dp, _dp]),
        "fb_fem_assemble": (C.c_int, [vp, _dp, _dp, _dp]),
        "fb_fem_system": (C.c_int, [vp, _dp, _dp]),
        "fb_fem_mass": (C.c_int, [vp, _dp]),
        "fb_fem_spmv": (C.c_int, [vp, _dp, _dp]),
        "fb_fem_pcg": (C.c_int, [vp, _dp, _dp, C.c_double, C.c_int, _ip]),
        "fb_fem_time_spmv": (C.c_int, [vp, C.c_int, _dp]),
        "fb_fem_time_assembly": (C.c_int, [vp, C.c_int, _dp]),
        "fb_fem_spmv_bytes": (C.c_int, [vp, _dp]),
        "fb_fem_assembly_bytes": (C.c_int, [vp, _dp]),
        "fb_fem_set_newmark": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, C.c_double]),
        "fb_fem_persist_info": (C.c_int, [vp, _ip, _ip, _ip]),
        "fb_fem_pcg_path": (C.c_int, [vp, C.c_char_p, C.c_int, _ip, _ip, _ip]),
        "fb_fem_persist_stats": (C.c_int, [vp, _ip, _dp, C.POINTER(C.c_longlong)]),
        "fb_fem_time_persist": (C.c_int, [vp, C.c_int, C.c_int, _dp]),
        "fb_fem_iteration_bytes": (C.c_int, [vp, _dp]),
        "fb_comm_unique_id": (C.c_int, [_bp]),
        "fb_comm_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, _bp, C.c_int]),
        "fb_comm_destroy": (C.c_int, [vp]),
        "fb_comm_info": (C.c_int, [vp, _ip, _ip, _ip]),
        "fb_comm_create_local": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_int]),
        "fb_comm_test_allgather": (C.c_int, [vp, C.c_void_p, C.c_void_p, C.c_size_t]),
        "fb_fem_create_from_poly": (C.c_int, [C.POINTER(vp), vp, C.c_int, _ip, C.POINTER(FemParams)]),
        "fb_fem_device_plan_get": (C.c_longlong, [vp, C.c_char_p, _ip, C.c_longlong]),
        "fb_fem_plan_on_device": (C.c_int, [vp]),
        "fb_fem_assembly_kernel": (C.c_int, [vp]),
        "fb_fem_assembly_wide_slices": (C.c_int, [vp]),
        "fb_plan_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, _ip, C.c_int, _ip, C.c_int, C.c_int, _ip]),
        "fb_plan_destroy": (C.c_int, [vp]),
        "fb_plan_slab_order": (C.c_int, [C.c_int, _dp, C.c_int, _ip, _ip, _ip, _ip]),
        "fb_plan_shard_vote": (C.c_int, [C.c_int, C.c_int, _ip, C.c_int, C.c_int, _ip, _ip]),
        "fb_plan_info": (C.c_int, [vp, _ip]),
        "fb_plan_get": (C.c_int, [vp, C.c_char_p, _ip, C.c_size_t]),
    }
    poly_sig = {
        "fb_fem_transport": (C.c_int, [vp]),
        "fb_fem_set_exchange_mode": (C.c_int, [vp, C.c_int]),
        "fb_fem_persist_rearms": (C.c_int, [vp]),
        "fb_fem_persist_helpers": (C.c_int, [vp]),
        "fb_fem_persist_gather": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "fb_fem_renumbering": (C.c_int, [vp, _ip, _ip]),
        "fb_fem_owned_nodes": (C.c_int, [vp, _ip]),
        "fb_fem_halo_info": (C.c_int, [vp, _ip, _ip]),
        "fb_fem_sharded_persist": (C.c_int, [vp]),
        "fb_fem_set_sharded_persist": (C.c_int, [vp, C.c_int]),
        "fb_fem_time_exchange": (C.c_int, [vp, C.c_int, _dp, _dp]),
        "fb_fem_time_element_stiffness": (C.c_int, [vp, C.c_int, _dp]),
        "fb_poly_create": (C.c_int, [C.POINTER(vp), C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp]),
        "fb_poly_destroy": (C.c_int, [vp]),
        "fb_poly_field_array": (C.c_int, [vp, C.c_int, _fp]),
        "fb_poly_set_field_semantics": (C.c_int, [vp, C.c_int, _fp]),
        "fb_poly_field_semantics": (C.c_int, [vp]),
        "fb_poly_sweep": (C.c_int, [vp, C.c_float, _ip]),
        "fb_poly_sweep_grid": (C.c_int, [vp, _fp, C.c_float, _ip]),
        "fb_poly_read_grid": (C.c_int, [vp, _fp]),
        "fb_poly_classify": (C.c_int, [vp, C.POINTER(PolyCounts)]),
        "fb_poly_read_classification": (C.c_int, [vp, _bp, _up, _bp]),
        "fb_poly_tetrahedralize": (C.c_int, [vp, C.POINTER(PolyCounts)]),
        "fb_poly_read_tetmesh": (C.c_int, [vp, _fp, _up]),
        "fb_poly_time_pipeline": (C.c_int, [vp, C.c_int, _dp, _dp]),
        "fb_poly_time_grid": (C.c_int, [vp, C.c_int, _dp]),
        "fb_poly_time_stages": (C.c_int, [vp, C.c_int, _dp]),
        "fb_poly_surface": (C.c_int, [vp, C.POINTER(PolyCounts)]),
        "fb_poly_read_surface": (C.c_int, [vp, _fp, _fp, _up]),
        "fb_poly_cube_table": (C.c_int, [_bp, _bp]),
        "fb_poly_time_surface": (C.c_int, [vp, C.c_int, _dp]),
        "fb_poly_off_surface": (C.c_int, [vp, C.c_float, _fp]),
        "fb_poly_read_surface_colors": (C.c_int, [vp, _fp]),
        "fb_poly_field_color_array": (C.c_int, [vp, C.c_int, _fp, _fp]),
        "fb_poly_interpolate_displacements": (C.c_int, [vp, C.c_int, _dp, _fp]),
        "fb_poly_read_surface_binding": (C.c_int, [vp, _up, _fp]),
        "fb_poly_compile_info": (C.c_int, [C.c_int, _fp, C.c_int, _fp, _ip, _ip]),
        "fb_poly_apply_displacements": (C.c_int, [vp, C.c_int, C.c_int, _dp, _fp]),
        "fb_poly_sweep_slab": (C.c_int, [vp, _fp, C.c_float, _ip, C.c_int, C.c_int]),
        "fb_poly_slab_counts": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, _ip, _ip]),
        "fb_poly_read_tetmesh_slab": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_uint, _fp, _up]),
        "fb_cut_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, _dp, C.c_int, _up]),
        "fb_cut_destroy": (None, [vp]),
        "fb_cut_set_vertices": (C.c_int, [vp, C.c_int, _dp]),
        "fb_cut_face_centroids": (C.c_int, [vp]),
        "fb_cut_face_intersections": (C.c_int, [vp, _dp, _dp, _ip]),
        "fb_cut_edge_intersections": (C.c_int, [vp, _dp, _ip]),
        "fb_cut_read": (C.c_int, [vp, C.c_int, _up, _fp]),
        "fb_cut_read_hits": (C.c_int, [vp, C.c_int, C.c_int, _up, _fp, _ip]),
        "fb_cut_segment_triangles": (C.c_int, [C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
        "fb_cut_time": (C.c_int, [vp, C.c_int, _dp, _dp, C.c_int, _dp]),
    }
    sig.update(poly_sig)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not match include/fembrain_hip.h
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(code):
    if code != FB_OK:
        raise FbError(code, lib().fb_last_error().decode("utf-8", "replace"))


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def fptr(a):
    return None if a is None else a.ctypes.data_as(_fp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def uptr(a):
    return None if a is None else a.ctypes.data_as(_up)


def bptr(a):
    return None if a is None else a.ctypes.data_as(_bp)


def as_f64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    if n is not None and a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    return a


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
