"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for the two CPU oracles.

* ``OrcFem``  -> oracle/libfem_oracle.so  (our C restatement, fem_oracle.c; travels to the GPU box)
* ``RefFem``  -> oracle/_ref/libfem_ref.so (the reference's own VegaFEM TUs + ref_harness.cpp; prebuilt here)

Both expose the same methods so tests can run them side by side.  Never import this from fembrain_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def build(ref=True):
    """Compile the C restatement (and, where /root/reference exists, the reference build)."""
    targets = [os.path.join(_HERE, "libfem_oracle.so")]
    subprocess.check_call(["make", "-s", "-f", os.path.join(_HERE, "Makefile")] + targets)
    if ref:
        subprocess.check_call(["make", "-s", "-f", os.path.join(_HERE, "Makefile"), "ref"])


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libfem_ref.so"))


_libs = {}


def _load(kind):
    if kind in _libs:
        return _libs[kind]
    path = os.path.join(_HERE, "libfem_oracle.so") if kind == "orc" else os.path.join(_HERE, "_ref", "libfem_ref.so")
    if kind == "orc" and not os.path.exists(path):
        build(ref=False)
    lib = C.CDLL(path)
    p = "orc_" if kind == "orc" else "ref_"
    f = getattr(lib, p + "fem_create")
    f.restype = C.c_void_p
    f.argtypes = [C.c_int, _dp, C.c_int, _ip, C.c_double, C.c_double, C.c_double]
    getattr(lib, p + "fem_destroy").argtypes = [C.c_void_p]
    for name in ("fem_K0", "fem_Minv"):
        getattr(lib, p + name).argtypes = [C.c_void_p, C.c_int, _dp]
    getattr(lib, p + "fem_assemble").argtypes = [C.c_void_p, _dp, _dp, _dp]
    getattr(lib, p + "fem_set_linear").argtypes = [C.c_void_p, C.c_int]
    getattr(lib, p + "fem_set_warp").argtypes = [C.c_void_p, C.c_int]
    getattr(lib, p + "get_accel").argtypes = [C.c_void_p, _dp]
    getattr(lib, p + "set_accel").argtypes = [C.c_void_p, _dp]
    nm = getattr(lib, p + "newmark_step")
    nm.restype = C.c_int
    nm.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, _ip]
    g = getattr(lib, p + "integrator_create")
    g.argtypes = [C.c_void_p, C.c_int, _ip, C.c_double, C.c_double, C.c_double]
    getattr(lib, p + "set_state").argtypes = [C.c_void_p, _dp, _dp]
    getattr(lib, p + "get_state").argtypes = [C.c_void_p, _dp, _dp]
    getattr(lib, p + "set_external_forces").argtypes = [C.c_void_p, _dp]
    s = getattr(lib, p + "step")
    s.restype = C.c_int
    s.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, _dp, _dp]
    getattr(lib, p + "polar").restype = C.c_double
    getattr(lib, p + "polar").argtypes = [_dp, _dp, _dp, C.c_double]
    getattr(lib, p + "sys_rows").restype = C.c_int
    getattr(lib, p + "sys_rows").argtypes = [C.c_void_p]
    if kind == "orc":
        lib.orc_fem_nnz.argtypes = [C.c_void_p]
        lib.orc_fem_nblk.argtypes = [C.c_void_p]
        lib.orc_fem_csr.argtypes = [C.c_void_p, _ip, _ip]
        lib.orc_fem_blocks.argtypes = [C.c_void_p, _ip, _ip]
        lib.orc_fem_elblk.argtypes = [C.c_void_p, C.c_int, _ip]
        lib.orc_fem_mass_on_pattern.argtypes = [C.c_void_p, _dp]
        lib.orc_fem_element.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
        lib.orc_sys_nnz.argtypes = [C.c_void_p]
        lib.orc_sys_csr.argtypes = [C.c_void_p, _ip, _ip, _dp]
        lib.orc_pcg.restype = C.c_int
        lib.orc_pcg.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_double, C.c_int, _dp]
        lib.orc_spmv.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
        lib.orc_step_prepare.argtypes = [C.c_void_p]
        lib.orc_pcg_bounded.restype = C.c_int
        lib.orc_pcg_bounded.argtypes = [C.c_void_p, C.c_double, C.c_int]
    else:
        lib.ref_fem_topology.restype = C.c_int
        lib.ref_fem_topology.argtypes = [C.c_void_p, _ip, _ip]
        lib.ref_fem_mass.restype = C.c_int
        lib.ref_fem_mass.argtypes = [C.c_void_p, _ip, _ip, _dp]
        lib.ref_fem_elem_indices.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
        lib.ref_sys_csr.restype = C.c_int
        lib.ref_sys_csr.argtypes = [C.c_void_p, _ip, _ip, _dp]
        lib.ref_pcg.restype = C.c_int
        lib.ref_pcg.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_double, C.c_int]
    _libs[kind] = lib
    return lib


class _Fem:
    kind = None

    def __init__(self, verts, tets, E=1e7, nu=0.46, rho=1000.0):
        self.lib = _load(self.kind)
        self.p = "orc_" if self.kind == "orc" else "ref_"
        self.verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        self.tets = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
        self.nv, self.nt, self.r = len(self.verts), len(self.tets), 3 * len(self.verts)
        self.h = C.c_void_p(getattr(self.lib, self.p + "fem_create")(
            self.nv, _d(self.verts), self.nt, _i(self.tets), E, nu, rho))
        self._integ = False

    def close(self):
        if self.h:
            getattr(self.lib, self.p + "fem_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def K0(self, el):
        out = np.empty(144)
        getattr(self.lib, self.p + "fem_K0")(self.h, el, _d(out))
        return out.reshape(12, 12)

    def Minv(self, el):
        out = np.empty(16)
        getattr(self.lib, self.p + "fem_Minv")(self.h, el, _d(out))
        return out.reshape(4, 4)

    def set_linear(self, linear=True):
        """warp = 0 of ComputeForceAndStiffnessMatrix (corotationalLinearFEM.cpp:429-453) instead of FemBrain's warp = 1"""
        getattr(self.lib, self.p + "fem_set_linear")(self.h, 1 if linear else 0)

    def set_warp(self, warp):
        """the `warp` argument of ComputeForceAndStiffnessMatrix: 0 linear, 1 corotational (FemBrain's), 2 corotational with
        the exact tangent stiffness (corotationalLinearFEM.cpp:296-428)"""
        getattr(self.lib, self.p + "fem_set_warp")(self.h, int(warp))

    def newmark_step(self, beta=0.25, gamma=0.5, max_newton=1, epsilon=1e-6, cg_eps=1e-6, cg_maxiter=10000):
        """ImplicitNewmarkSparse::DoTimestep (implicitNewmarkSparse.cpp:183-379): (Newton iterations or -1, total PCG iterations)"""
        tot = C.c_int(0)
        n = getattr(self.lib, self.p + "newmark_step")(self.h, beta, gamma, max_newton, epsilon, cg_eps, cg_maxiter, C.byref(tot))
        return n, tot.value

    def get_accel(self):
        a = np.empty(self.r)
        getattr(self.lib, self.p + "get_accel")(self.h, _d(a))
        return a

    def assemble(self, u, want_K=True):
        u = np.ascontiguousarray(u, dtype=np.float64)
        f = np.empty(self.r)
        Kv = np.empty(self.nnz()) if want_K else None
        getattr(self.lib, self.p + "fem_assemble")(self.h, _d(u), _d(f), _d(Kv))
        return f, Kv

    def integrator(self, fixed_dofs, timestep=0.0333, cM=0.0, cK=0.01):
        fd = np.ascontiguousarray(fixed_dofs, dtype=np.int32)
        self.fixed = fd
        getattr(self.lib, self.p + "integrator_create")(self.h, len(fd), _i(fd), timestep, cM, cK)
        self._integ = True

    def set_state(self, q, qvel=None):
        q = np.ascontiguousarray(q, dtype=np.float64)
        qv = None if qvel is None else np.ascontiguousarray(qvel, dtype=np.float64)
        getattr(self.lib, self.p + "set_state")(self.h, _d(q), _d(qv))

    def get_state(self):
        q, qv = np.empty(self.r), np.empty(self.r)
        getattr(self.lib, self.p + "get_state")(self.h, _d(q), _d(qv))
        return q, qv

    def set_external_forces(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        getattr(self.lib, self.p + "set_external_forces")(self.h, _d(f))

    def step(self, cg_eps=1e-6, cg_maxiter=10000, want=False):
        keff = np.empty(self.nnz()) if want else None
        rhs = np.empty(self.r) if want else None
        dv = np.empty(self.r) if want else None
        info = getattr(self.lib, self.p + "step")(self.h, cg_eps, cg_maxiter, _d(keff), _d(rhs), _d(dv))
        return (info, keff, rhs, dv) if want else info

    def polar(self, F, tol=1e-6):
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
        R, S = np.empty(9), np.empty(9)
        det = getattr(self.lib, self.p + "polar")(_d(F), _d(R), _d(S), tol)
        return det, R.reshape(3, 3), S.reshape(3, 3)


class OrcFem(_Fem):
    kind = "orc"

    def nnz(self):
        return self.lib.orc_fem_nnz(self.h)

    def csr(self):
        ia, ja = np.empty(self.r + 1, np.int32), np.empty(self.nnz(), np.int32)
        self.lib.orc_fem_csr(self.h, _i(ia), _i(ja))
        return ia, ja

    def blocks(self):
        bptr, bcol = np.empty(self.nv + 1, np.int32), np.empty(self.lib.orc_fem_nblk(self.h), np.int32)
        self.lib.orc_fem_blocks(self.h, _i(bptr), _i(bcol))
        return bptr, bcol

    def mass_on_pattern(self):
        a = np.empty(self.nnz())
        self.lib.orc_fem_mass_on_pattern(self.h, _d(a))
        return a

    def element(self, el, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        R, Ke, fe = np.empty(9), np.empty(144), np.empty(12)
        self.lib.orc_fem_element(self.h, el, _d(u), _d(R), _d(Ke), _d(fe))
        return R.reshape(3, 3), Ke.reshape(12, 12), fe

    def sys_csr(self):
        n, nz = self.lib.orc_sys_rows(self.h), self.lib.orc_sys_nnz(self.h)
        ia, ja, a = np.empty(n + 1, np.int32), np.empty(nz, np.int32), np.empty(nz)
        self.lib.orc_sys_csr(self.h, _i(ia), _i(ja), _d(a))
        return ia, ja, a

    def step_prepare(self):
        self.lib.orc_step_prepare(self.h)

    def pcg_bounded(self, eps, maxit):
        return self.lib.orc_pcg_bounded(self.h, eps, maxit)


class RefFem(_Fem):
    kind = "ref"

    def nnz(self):
        return self.lib.ref_fem_topology(self.h, None, None)

    def csr(self):
        ia, ja = np.empty(self.r + 1, np.int32), np.empty(self.nnz(), np.int32)
        self.lib.ref_fem_topology(self.h, _i(ia), _i(ja))
        return ia, ja

    def mass_csr(self):
        nz = self.lib.ref_fem_mass(self.h, None, None, None)
        ia, ja, a = np.empty(self.r + 1, np.int32), np.empty(nz, np.int32), np.empty(nz)
        self.lib.ref_fem_mass(self.h, _i(ia), _i(ja), _d(a))
        return ia, ja, a

    def elem_indices(self, el):
        row, col = np.empty(4, np.int32), np.empty(16, np.int32)
        self.lib.ref_fem_elem_indices(self.h, el, _i(row), _i(col))
        return row, col

    def sys_csr(self):
        nz = self.lib.ref_sys_csr(self.h, None, None, None)
        n = self.lib.ref_sys_rows(self.h)
        ia, ja, a = np.empty(n + 1, np.int32), np.empty(nz, np.int32), np.empty(nz)
        self.lib.ref_sys_csr(self.h, _i(ia), _i(ja), _d(a))
        return ia, ja, a


def orc_pcg(ia, ja, a, b, eps=1e-6, maxit=10000, x0=None):
    lib = _load("orc")
    n = len(b)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    work = np.empty(4 * n)
    ia, ja = np.ascontiguousarray(ia, np.int32), np.ascontiguousarray(ja, np.int32)
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    info = lib.orc_pcg(n, _i(ia), _i(ja), _d(a), _d(b), _d(x), eps, maxit, _d(work))
    return info, x


def ref_pcg(ia, ja, a, b, eps=1e-6, maxit=10000, x0=None):
    lib = _load("ref")
    n = len(b)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    ia, ja = np.ascontiguousarray(ia, np.int32), np.ascontiguousarray(ja, np.int32)
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    info = lib.ref_pcg(n, _i(ia), _i(ja), _d(a), _d(b), _d(x), eps, maxit)
    return info, x
