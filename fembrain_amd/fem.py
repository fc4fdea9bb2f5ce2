"""Host-side mirror of the reference's FEM class surface over the C ABI.

``FemIntegrator`` plays the role of ``VolumeConservingIntegrator`` + ``CorotationalLinearFEMForceModel``
(reference src/deformable/PS_VolumeConservingIntegrator.{h,cpp}, vegafem/integrator/integratorBase*.h): same
method names in snake_case, same argument meaning, same error behaviour except that a failed solve raises
``FbError(FB_ESOLVER)`` instead of ``exit(-1)``.  ``Deformable`` mirrors the per-step driver of
``Deformable::timestep`` (reference src/deformable/Deformable.cpp:318-420).
"""
import ctypes as C

import numpy as np

from . import lib as _l
from .meshgen import fixed_vertices_to_dofs


class FemIntegrator:
    def __init__(self, verts, tets, fixed_dofs=(), E=1e7, nu=0.46, rho=1000.0, timestep=0.0333,
                 damping_mass=0.0, damping_stiffness=0.01, cg_eps=1e-6, cg_max_iter=10000,
                 matrix_precision=_l.FB_MATRIX_AUTO, device=0, shard=None, pcg_variant=_l.FB_PCG_MERGED, spmv_kernel=_l.FB_SPMV_AUTO,
                 linear=False, exact_tangent=False, integrator=_l.FB_INTEGRATOR_VOLUME_CONSERVING, renumber=_l.FB_RENUMBER_AUTO, expect_cuts=False,
                 reserve_nodes=0, reserve_elements=0):
        """shard = (n_ranks, rank, node_splits or None, comm_handle) for a domain-decomposed handle."""
        L = _l.lib()
        self._L = L
        self.verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        self.tets = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
        self.n_nodes, self.n_tets_global = len(self.verts), len(self.tets)
        self.r = 3 * self.n_nodes
        fd = _l.as_i32(fixed_dofs)
        p = _l.FemParams()
        L.fb_fem_default_params(C.byref(p))
        p.E, p.nu, p.rho, p.timestep = E, nu, rho, timestep
        p.damping_mass, p.damping_stiffness = damping_mass, damping_stiffness
        p.cg_eps, p.cg_max_iter, p.matrix_precision, p.device = cg_eps, cg_max_iter, matrix_precision, device
        p.pcg_variant = pcg_variant
        p.spmv_kernel = spmv_kernel
        p.linear = 1 if linear else 0
        p.exact_tangent = 1 if exact_tangent else 0
        p.integrator = integrator
        p.renumber = renumber
        p.expect_cuts, p.reserve_nodes, p.reserve_elements = int(bool(expect_cuts)), int(reserve_nodes), int(reserve_elements)
        self.params = p
        self.h = C.c_void_p()
        self.node_lo, self.node_hi = 0, self.n_nodes
        if shard is None:
            _l.check(L.fb_fem_create(C.byref(self.h), self.n_nodes, _l.dptr(self.verts), self.n_tets_global,
                                     _l.iptr(self.tets), len(fd), _l.iptr(fd), C.byref(p)))
        else:
            n_ranks, rank, splits, comm = shard
            sp = None if splits is None else _l.as_i32(splits)
            _l.check(L.fb_fem_create_sharded(C.byref(self.h), self.n_nodes, _l.dptr(self.verts), self.n_tets_global,
                                             _l.iptr(self.tets), len(fd), _l.iptr(fd), C.byref(p), n_ranks, rank,
                                             _l.iptr(sp), comm))
            if sp is None:
                sp = np.array([self.n_nodes * i // n_ranks for i in range(n_ranks + 1)], dtype=np.int32)
            self.node_lo, self.node_hi = int(sp[rank]), int(sp[rank + 1])
        self.last = _l.StepInfo()

    @classmethod
    def from_poly(cls, poly, fixed_dofs=(), E=1e7, nu=0.46, rho=1000.0, timestep=0.0333, damping_mass=0.0, damping_stiffness=0.01,
                  cg_eps=1e-6, cg_max_iter=10000, matrix_precision=_l.FB_MATRIX_AUTO, device=0, linear=False):
        """The tet mesh a GpuPoly holds on the device (after tetrahedralize) as the FEM mesh, without a host copy
        (fb_fem_create_from_poly).  verts / tets are read back only for the Python-side conveniences."""
        self = cls.__new__(cls)
        L = _l.lib()
        self._L = L
        xyz, tets = poly.read_tetmesh()
        self.verts = xyz.astype(np.float64)
        self.tets = tets.astype(np.int32)
        self.n_nodes, self.n_tets_global = len(self.verts), len(self.tets)
        self.r = 3 * self.n_nodes
        fd = _l.as_i32(fixed_dofs)
        p = _l.FemParams()
        L.fb_fem_default_params(C.byref(p))
        p.E, p.nu, p.rho, p.timestep = E, nu, rho, timestep
        p.damping_mass, p.damping_stiffness = damping_mass, damping_stiffness
        p.cg_eps, p.cg_max_iter, p.matrix_precision, p.device = cg_eps, cg_max_iter, matrix_precision, device
        p.linear = 1 if linear else 0
        self.params = p
        self.h = C.c_void_p()
        self.node_lo, self.node_hi = 0, self.n_nodes
        _l.check(L.fb_fem_create_from_poly(C.byref(self.h), poly.h, len(fd), _l.iptr(fd), C.byref(p)))
        self.last = _l.StepInfo()
        return self

    # -- life cycle --
    def time_element_stiffness(self, reps=5):
        """Seconds to form K0 = V B^T E B of every element with the fp64 MFMA kernel (inspection path)."""
        a = C.c_double(0)
        _l.check(self._L.fb_fem_time_element_stiffness(self.h, reps, C.byref(a)))
        return a.value

    def time_exchange(self, reps=100):
        """(seconds per halo refresh, seconds per 3-scalar global sum); collective on a sharded handle."""
        a, b = C.c_double(0), C.c_double(0)
        _l.check(self._L.fb_fem_time_exchange(self.h, reps, C.byref(a), C.byref(b)))
        return a.value, b.value

    def transport(self):
        """0 unsharded, else the exchange mode in use (lib.FB_XCH_*)."""
        return int(self._L.fb_fem_transport(self.h))

    def set_exchange_mode(self, mode):
        """Collective: every rank of a sharded handle switches between two solves."""
        _l.check(self._L.fb_fem_set_exchange_mode(self.h, mode))

    def renumbering(self):
        """(works in an internal node order?, widest element in the caller's order, in the internal order)"""
        a, b = C.c_int(0), C.c_int(0)
        on = self._L.fb_fem_renumbering(self.h, C.byref(a), C.byref(b))
        return bool(on), a.value, b.value

    def owned_nodes(self):
        """caller ids of the owned nodes in internal order"""
        ids = np.empty(self.node_hi - self.node_lo, np.int32)
        _l.check(self._L.fb_fem_owned_nodes(self.h, _l.iptr(ids)))
        return ids

    def halo_info(self):
        """(halo nodes, neighbour ranks) of a sharded handle"""
        a, b = C.c_int(0), C.c_int(0)
        _l.check(self._L.fb_fem_halo_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sharded_persist(self):
        """True when this sharded handle's solves run inside the sharded persistent launches (FEMBRAIN_SHARDED_PERSIST=1)."""
        return bool(self._L.fb_fem_sharded_persist(self.h))

    def set_sharded_persist(self, on):
        """Collective: every rank switches between two solves."""
        _l.check(self._L.fb_fem_set_sharded_persist(self.h, 1 if on else 0))

    def close(self):
        if getattr(self, "h", None):
            self._L.fb_fem_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resync(self, verts, tets, fixed_dofs=(), node_splits=None):
        """Deformable::syncForceModel after a cut.  On a sharded handle the call is collective; node_splits = the new node
        ranges (None: kept when the node count is unchanged, else the equal split)."""
        self.verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
        self.tets = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
        self.n_nodes, self.r = len(self.verts), 3 * len(self.verts)
        fd = _l.as_i32(fixed_dofs)
        sp = None if node_splits is None else _l.as_i32(node_splits)
        _l.check(self._L.fb_fem_resync_sharded(self.h, self.n_nodes, _l.dptr(self.verts), len(self.tets), _l.iptr(self.tets),
                                               len(fd), _l.iptr(fd), _l.iptr(sp)))
        rng = (C.c_int * 2)()
        _l.check(self._L.fb_fem_owned_range(self.h, rng))
        self.node_lo, self.node_hi = int(rng[0]), int(rng[1])

    def resync_delta(self, delta, fixed_dofs=(), track=True):
        """The same from a description of the change (fb_fem_resync_delta): delta = dict(removed, changed_ids, changed_nodes, added,
        new_xyz) in the terms of ``meshgen.apply_delta`` -- ids of the handle's current element list, new nodes appended.  The
        mesh stays on the device; ``self.verts`` / ``self.tets`` follow on the host for the caller's convenience (track=False: they
        are dropped instead -- timing runs)."""
        from .meshgen import apply_delta
        rem = _l.as_i32(delta.get("removed", ()))
        cid = _l.as_i32(delta.get("changed_ids", ()))
        cno = np.ascontiguousarray(delta.get("changed_nodes", ()), dtype=np.int32).reshape(-1)
        add = np.ascontiguousarray(delta.get("added", ()), dtype=np.int32).reshape(-1)
        nxy = np.ascontiguousarray(delta.get("new_xyz", ()), dtype=np.float64).reshape(-1)
        fd = _l.as_i32(fixed_dofs)
        _l.check(self._L.fb_fem_resync_delta(self.h, len(rem), _l.iptr(rem), len(cid), _l.iptr(cid), _l.iptr(cno), len(add) // 4, _l.iptr(add),
                                             len(nxy) // 3, _l.dptr(nxy), len(fd), _l.iptr(fd)))
        if track and self.verts is not None:
            self.verts, self.tets = apply_delta(self.verts, self.tets, dict(removed=rem, changed_ids=cid, changed_nodes=cno, added=add, new_xyz=nxy))
            self.n_nodes = len(self.verts)
        else:
            self.verts = self.tets = None
            self.n_nodes = int(self._L.fb_fem_num_nodes(self.h))
        self.r = 3 * self.n_nodes
        self.node_lo, self.node_hi = 0, self.n_nodes

    def resync_path(self):
        """FB_RESYNC_* of the handle's last (re-)build"""
        return int(self._L.fb_fem_resync_path(self.h))

    def rebuild_elements(self):
        _l.check(self._L.fb_fem_rebuild_elements(self.h))

    # -- IntegratorBase surface --
    def set_external_forces(self, f):
        _l.check(self._L.fb_fem_set_external_forces(self.h, _l.dptr(_l.as_f64(f, self.r))))

    def add_external_forces(self, f):
        _l.check(self._L.fb_fem_add_external_forces(self.h, _l.dptr(_l.as_f64(f, self.r))))

    def set_external_forces_to_zero(self):
        _l.check(self._L.fb_fem_set_external_forces_zero(self.h))

    def set_uniform_force(self, axis, value):
        _l.check(self._L.fb_fem_set_uniform_force(self.h, axis, value))

    def do_timestep(self):
        """Returns the PCG iteration count; raises FbError(FB_ESOLVER) when the solve fails."""
        _l.check(self._L.fb_fem_step(self.h, C.byref(self.last)))
        return self.last.cg_iterations

    def get_q_state(self):
        q, qv, qa = np.zeros(self.r), np.zeros(self.r), np.zeros(self.r)
        _l.check(self._L.fb_fem_get_state(self.h, _l.dptr(q), _l.dptr(qv), _l.dptr(qa)))
        return q, qv, qa

    def set_q_state(self, q, qvel=None, qaccel=None):
        qv = None if qvel is None else _l.as_f64(qvel, self.r)
        qa = None if qaccel is None else _l.as_f64(qaccel, self.r)
        _l.check(self._L.fb_fem_set_state(self.h, _l.dptr(_l.as_f64(q, self.r)), _l.dptr(qv), _l.dptr(qa)))

    def set_newmark(self, beta=0.25, gamma=0.5, max_newton_iterations=1, epsilon=1e-6):
        _l.check(self._L.fb_fem_set_newmark(self.h, beta, gamma, max_newton_iterations, epsilon))

    def reset_to_rest(self):
        _l.check(self._L.fb_fem_reset(self.h))

    def set_timestep(self, h):
        _l.check(self._L.fb_fem_set_timestep(self.h, h))

    def set_damping(self, mass_coef, stiffness_coef):
        _l.check(self._L.fb_fem_set_damping(self.h, mass_coef, stiffness_coef))

    def set_internal_force_scaling_factor(self, factor):
        _l.check(self._L.fb_fem_set_internal_force_scaling(self.h, factor))

    def set_cg(self, eps, max_iter):
        _l.check(self._L.fb_fem_set_cg(self.h, eps, max_iter))

    def set_constrained_dofs(self, fixed_dofs):
        fd = _l.as_i32(fixed_dofs)
        _l.check(self._L.fb_fem_set_constrained_dofs(self.h, len(fd), _l.iptr(fd)))

    def get_force_assembly_time(self):
        return self.last.assembly_seconds

    def get_system_solve_time(self):
        return self.last.solve_seconds

    def floor_collision(self, floor_y, restitution=0.4):
        n = C.c_int(0)
        _l.check(self._L.fb_fem_floor_collision(self.h, floor_y, restitution, C.byref(n)))
        return n.value

    # -- inspection (parity tests) --
    def num_tets(self):
        return self._L.fb_fem_num_tets(self.h)

    def num_blocks(self):
        return self._L.fb_fem_num_blocks(self.h)

    def matrix_precision(self):
        """lib.FB_MATRIX_F32 / FB_MATRIX_F64: the width the matrix values are stored in (FB_MATRIX_AUTO decides by size)"""
        return int(self._L.fb_fem_matrix_precision(self.h))

    def pattern(self):
        n_owned = self.node_hi - self.node_lo
        bptr, bcol = np.empty(n_owned + 1, np.int32), np.empty(self.num_blocks(), np.int32)
        _l.check(self._L.fb_fem_pattern(self.h, _l.iptr(bptr), _l.iptr(bcol)))
        return bptr, bcol

    def element_stiffness(self, first, count):
        K0, Mi = np.empty((count, 12, 12)), np.empty((count, 4, 4))
        _l.check(self._L.fb_fem_element_stiffness(self.h, first, count, _l.dptr(K0), _l.dptr(Mi)))
        return K0, Mi

    def assemble(self, u):
        f, K = np.zeros(self.r), np.empty((self.num_blocks(), 3, 3))
        _l.check(self._L.fb_fem_assemble(self.h, _l.dptr(_l.as_f64(u, self.r)), _l.dptr(f), _l.dptr(K)))
        return f, K

    def system(self):
        rhs, K = np.zeros(self.r), np.empty((self.num_blocks(), 3, 3))
        _l.check(self._L.fb_fem_system(self.h, _l.dptr(K), _l.dptr(rhs)))
        return K, rhs

    def mass(self):
        m = np.empty(self.num_blocks())
        _l.check(self._L.fb_fem_mass(self.h, _l.dptr(m)))
        return m

    def spmv(self, x):
        y = np.zeros(self.r)
        _l.check(self._L.fb_fem_spmv(self.h, _l.dptr(_l.as_f64(x, self.r)), _l.dptr(y)))
        return y

    def pcg(self, rhs, eps=1e-6, max_iter=10000):
        x, it = np.zeros(self.r), C.c_int(0)
        _l.check(self._L.fb_fem_pcg(self.h, _l.dptr(_l.as_f64(rhs, self.r)), _l.dptr(x), eps, max_iter, C.byref(it)))
        return it.value, x

    def time_spmv(self, reps=50):
        s = C.c_double(0)
        _l.check(self._L.fb_fem_time_spmv(self.h, reps, C.byref(s)))
        return s.value

    def persist_info(self):
        """(runs persistent PCG launches?, wavefronts per CU, workgroups, LDS-resident slots per slice)"""
        w, b, k = C.c_int(0), C.c_int(0), C.c_int(0)
        on = self._L.fb_fem_persist_info(self.h, C.byref(w), C.byref(b), C.byref(k))
        return bool(on == 1), w.value, b.value, k.value

    def persist_gather(self):
        """(published vector node by node?, cache lines a slot's gathers touch in planes, in 24-byte records) -- fb_fem_persist_gather"""
        a, b = C.c_double(0), C.c_double(0)
        on = self._L.fb_fem_persist_gather(self.h, C.byref(a), C.byref(b))
        return bool(on), a.value, b.value

    def pcg_path(self):
        """What ran: dict(path = FB_PCG_PATH_* of the last solve, kernel = the persistent instantiation this handle launches or '',
        launches, fallbacks, max_producers)"""
        name = C.create_string_buffer(96)
        nl, nf, mp = C.c_int(0), C.c_int(0), C.c_int(0)
        path = self._L.fb_fem_pcg_path(self.h, name, 96, C.byref(nl), C.byref(nf), C.byref(mp))
        return dict(path=path, kernel=name.value.decode(), launches=nl.value, fallbacks=nf.value, max_producers=mp.value)

    def persist_stats(self):
        """(persistent launches of this handle's solves so far, their device seconds, the PCG iterations they ran)"""
        n, sec, it = C.c_int(0), C.c_double(0), C.c_longlong(0)
        _l.check(self._L.fb_fem_persist_stats(self.h, C.byref(n), C.byref(sec), C.byref(it)))
        return n.value, sec.value, it.value

    def time_persist(self, reps=10, n_iters=29):
        s = C.c_double(0)
        _l.check(self._L.fb_fem_time_persist(self.h, reps, n_iters, C.byref(s)))
        return s.value

    def iteration_bytes(self):
        b = C.c_double(0)
        _l.check(self._L.fb_fem_iteration_bytes(self.h, C.byref(b)))
        return b.value

    def time_assembly(self, reps=10):
        s = C.c_double(0)
        _l.check(self._L.fb_fem_time_assembly(self.h, reps, C.byref(s)))
        return s.value

    def spmv_bytes(self):
        b = C.c_double(0)
        _l.check(self._L.fb_fem_spmv_bytes(self.h, C.byref(b)))
        return b.value

    def assembly_bytes(self):
        b = C.c_double(0)
        _l.check(self._L.fb_fem_assembly_bytes(self.h, C.byref(b)))
        return b.value


def bsr_to_scipy(bptr, bcol, blocks, n_cols_nodes=None):
    """3x3-block CSR (fb_fem_pattern order) -> scipy CSR, for comparisons in the tests."""
    import scipy.sparse as sp
    n_rows = len(bptr) - 1
    n_cols = n_cols_nodes if n_cols_nodes is not None else n_rows
    return sp.bsr_matrix((np.asarray(blocks).reshape(-1, 3, 3), bcol, bptr), shape=(3 * n_rows, 3 * n_cols)).tocsr()


def spread_haptic_forces(bptr, bcol, indices, forces, neighborhood_size, ext_forces):
    """``Deformable::applyHapticForces`` (reference src/deformable/Deformable.cpp:634-706): the force of each haptic
    vertex is added to it and, with the linear fall-off (size - j) / size, to the vertices first reached in ring
    j = 1 .. size-1 of a breadth-first walk over mesh edges.  Node neighbours are the off-diagonal columns of the
    stiffness pattern (= vertices sharing a tet edge; the reference's ``VolMesh::get_node_neighbors`` intends exactly
    that set -- its ``const_edgeAt(i)`` indexing slip, VolMesh.cpp:1346-1363, is not reproduced).  Adds in place."""
    for idx, frc in zip(indices, forces):
        ext_forces[3 * idx:3 * idx + 3] += frc
    for idx, frc in zip(indices, forces):
        affected, last = {int(idx)}, {int(idx)}
        for j in range(1, neighborhood_size):
            mag = 1.0 * (neighborhood_size - j) / float(neighborhood_size)
            new = set()
            for vtx in last:
                for nb in bcol[bptr[vtx]:bptr[vtx + 1]]:
                    nb = int(nb)
                    if nb != vtx and nb not in affected:
                        new.add(nb)
            last = set()
            for nb in sorted(new):
                ext_forces[3 * nb:3 * nb + 3] += mag * np.asarray(frc, dtype=np.float64)
                last.add(nb)
                affected.add(nb)
    return ext_forces


class Deformable:
    """Per-step driver of ``Deformable::timestep`` (reference src/deformable/Deformable.cpp:318-420) without the
    scene-graph / GL parts: external forces (gravity -10000 per y-DOF unless a collision happened in the previous
    step, haptic forces), DoTimestep, floor collision with velocity rewrite, deformation callback."""

    GRAVITY_FORCE = -10000.0  # Deformable.cpp:335

    def __init__(self, verts, tets, fixed_vertices=(), floor_y=None, gravity=True, **kw):
        self.fixed_vertices = sorted(int(v) for v in fixed_vertices)
        fd = fixed_vertices_to_dofs(self.fixed_vertices) if len(self.fixed_vertices) else np.zeros(0, np.int32)
        self.integrator = FemIntegrator(verts, tets, fd, **kw)
        self.dof = self.integrator.r
        self.gravity = bool(gravity)
        self.floor_y = floor_y
        self.ct_collided = 0
        self.ct_timestep = 0
        self.haptic_indices, self.haptic_forces = [], []
        self.haptic_in_progress = False
        self.haptic_force_neighborhood_size = 5  # DEFAULT_FORCE_NEIGHBORHOOD_SIZE, Deformable.h:41
        self._pattern = None
        self.on_deform = None  # FOnApplyDeformations(dof, q), Deformable.h:46

    def set_deform_callback(self, fn):
        self.on_deform = fn

    def haptic_set_current_forces(self, indices, forces):
        self.haptic_indices, self.haptic_forces = list(indices), [tuple(f) for f in forces]

    def set_haptic_force_radius(self, radius):
        self.haptic_force_neighborhood_size = int(radius)

    def get_haptic_force_radius(self):
        return self.haptic_force_neighborhood_size

    def haptic_start(self, index):
        self.haptic_in_progress = True

    def haptic_end(self):
        self.haptic_in_progress = False
        self.haptic_indices, self.haptic_forces = [], []

    def timestep(self):
        it = self.integrator
        apply_gravity = self.gravity and self.ct_collided == 0
        if self.haptic_in_progress and self.haptic_indices:
            f = np.zeros(self.dof)
            if apply_gravity:
                f[1::3] += self.GRAVITY_FORCE
            if self._pattern is None:
                self._pattern = it.pattern()
            spread_haptic_forces(self._pattern[0], self._pattern[1], self.haptic_indices, self.haptic_forces,
                                 self.haptic_force_neighborhood_size, f)
            it.set_external_forces(f)
        elif apply_gravity:
            it.set_uniform_force(1, self.GRAVITY_FORCE)
        else:
            it.set_external_forces_to_zero()
        iters = it.do_timestep()
        if self.floor_y is not None:
            self.ct_collided = it.floor_collision(self.floor_y, 0.4)
        self.ct_timestep += 1
        if self.on_deform is not None:
            q, _, _ = it.get_q_state()
            self.on_deform(self.dof, q)
        return iters

    def get_solver_time(self):
        return self.integrator.get_system_solve_time()

    def reset_deformations(self):
        self.integrator.reset_to_rest()
