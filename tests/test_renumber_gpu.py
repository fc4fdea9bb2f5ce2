"""The locality renumbering behind the ABI (fembrain_amd/csrc/renumber.h).

The reference numbers nodes badly for a banded solver: CuttableMesh::cut appends the nodes it creates at the end of the list
(src/deformable/VolMesh.cpp:1086-1091,1639-1642) and the shipped blobtree/*.veg meshes are TetGen outputs (surface vertices
first).  A handle may therefore work in an internal slab order of its own; everything that crosses the C ABI stays in the
CALLER's numbering.  These tests hold a renumbered handle against a handle that keeps the caller's order, against the CPU
oracle, and against the reference-built golden of the 1M-tet step, on the same meshes in scrambled node orders.
"""
import ctypes as C
import os

import numpy as np
import pytest

from fembrain_amd import lib as fl
from fembrain_amd.fem import FemIntegrator, bsr_to_scipy
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
from oracle.pyoracle import OrcFem

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def scramble(v, t, fixed_nodes, new_of_old):
    """node `old` of the mesh becomes node new_of_old[old]"""
    new_of_old = np.asarray(new_of_old, dtype=np.int64)
    v2 = np.empty_like(v)
    v2[new_of_old] = v
    return v2, np.ascontiguousarray(new_of_old[t].astype(np.int32)), np.sort(new_of_old[np.asarray(fixed_nodes, dtype=np.int64)]).astype(np.int32)


def orders(n_nodes, surface_mask, seed):
    """the three caller orders of VERDICT r3 item 1: random, 5 % of the nodes appended at the end (a cut), surface nodes first (TetGen)"""
    rng = np.random.default_rng(seed)
    yield "random", rng.permutation(n_nodes)
    moved = np.sort(rng.choice(n_nodes, max(1, n_nodes // 20), replace=False))
    keep = np.setdiff1d(np.arange(n_nodes), moved)
    m = np.empty(n_nodes, np.int64)
    m[keep] = np.arange(len(keep))
    m[rng.permutation(moved)] = len(keep) + np.arange(len(moved))
    yield "cut_appended", m
    order = np.concatenate([np.nonzero(surface_mask)[0], np.nonzero(~surface_mask)[0]])
    m = np.empty(n_nodes, np.int64)
    m[order] = np.arange(n_nodes)
    yield "surface_first", m


def cube_surface(n):
    ijk = np.stack(np.unravel_index(np.arange(n ** 3), (n, n, n)), axis=1)
    return ((ijk == 0) | (ijk == n - 1)).any(axis=1)


@pytest.mark.parametrize("prec", [fl.FB_MATRIX_F64, fl.FB_MATRIX_F32])
def test_renumbered_handle_answers_in_the_callers_numbering(gpu, prec):
    """Forced on a small scrambled cube: pattern (bit-exact, columns ascending in the caller's ids), mass, K and f at a seeded
    displacement, Keff / rhs, SpMV, PCG, steps, state round trips, constraints and forces at caller ids -- against a handle that
    keeps the caller's order and against the oracle on the same scrambled mesh.  Blocks and forces accumulate their element
    contributions in the caller's element order either way, so f, the mass and the element matrices agree BIT FOR BIT; whatever sums
    over the columns of a row in slot order -- the SpMV, the (hK + D) qvel term of the right-hand side, and the diagonal blocks, which
    are formed as minus the sum of the row's other blocks -- agrees to rounding."""
    n = 9
    v0, t0 = truth_cube(n, n, n, 0.1)
    for name, m in orders(len(v0), cube_surface(n), 11):
        v, t, fx = scramble(v0, t0, cube_fixed_plane_i0(n, n), m)
        fixed = fixed_vertices_to_dofs(fx)
        a = FemIntegrator(v, t, fixed, matrix_precision=prec, renumber=fl.FB_RENUMBER_ON)
        b = FemIntegrator(v, t, fixed, matrix_precision=prec, renumber=fl.FB_RENUMBER_OFF)
        on, span_c, span_i = a.renumbering()
        assert on and span_i < span_c and span_i <= n * n + n + 1, (name, span_c, span_i)   # the grid's own plane-by-plane order is found again
        assert b.renumbering() == (False, 0, 0)
        assert np.array_equal(v[a.owned_nodes()], v0)                                     # internal order = the grid's
        o = OrcFem(v, t)
        o.integrator(fixed)
        pa, pb = a.pattern(), b.pattern()
        assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
        ob = o.blocks()
        assert np.array_equal(pa[0], ob[0]) and np.array_equal(pa[1], ob[1])
        assert np.array_equal(a.mass(), b.mass())
        rng = np.random.default_rng(3)
        u = rng.normal(size=a.r) * 0.01
        (fa, Ka), (fb, Kb) = a.assemble(u), b.assemble(u)
        eps_k = 1e-13 if prec == fl.FB_MATRIX_F64 else 2e-7
        assert np.array_equal(fa, fb) and np.abs(Ka - Kb).max() <= eps_k * np.abs(Kb).max(), name
        off = np.repeat(np.arange(len(pa[0]) - 1), np.diff(pa[0])) != pa[1]
        assert np.array_equal(Ka[off], Kb[off])           # off-diagonal blocks: the same sums of the same element terms
        fo, _ = o.assemble(u)
        assert np.abs(fa - fo).max() <= 1e-9 * np.abs(fo).max()
        K0a, Mia = a.element_stiffness(0, len(t))
        K0b, Mib = b.element_stiffness(0, len(t))
        assert np.array_equal(K0a, K0b) and np.array_equal(Mia, Mib)                      # elements keep their order and their vertex order
        # state and forces at caller ids
        q0 = rng.normal(size=a.r) * 0.003
        w0 = rng.normal(size=a.r) * 0.1
        q0[fixed] = 0
        w0[fixed] = 0
        fext = rng.normal(size=a.r) * 5.0
        for g in (a, b):
            g.set_q_state(q0, w0)
            g.set_external_forces(fext)
            g.add_external_forces(0.5 * fext)
        qa, wa, _ = a.get_q_state()
        assert np.array_equal(qa, q0) and np.array_equal(wa, w0)
        (Ea, ra), (Eb, rb) = a.system(), b.system()
        assert np.abs(Ea - Eb).max() <= eps_k * np.abs(Eb).max() and np.abs(ra - rb).max() <= 1e-12 * np.abs(rb).max() and not ra[fixed].any()
        x = rng.normal(size=a.r)
        ya, yb = a.spmv(x), b.spmv(x)
        A = bsr_to_scipy(pa[0], pa[1], Ea)
        assert np.abs(ya - A @ x).max() <= 1e-12 * np.abs(A @ x).max() and np.abs(ya - yb).max() <= max(10 * eps_k, 1e-12) * np.abs(yb).max()
        ia, xa = a.pcg(ra, eps=1e-10, max_iter=20000)
        ib, xb = b.pcg(rb, eps=1e-10, max_iter=20000)
        assert ia > 0 and abs(ia - ib) <= max(3, 0.02 * ib) and np.abs(xa - xb).max() <= 1e-7 * np.abs(xb).max() and not xa[fixed].any()
        # three reference-load steps against the oracle on the scrambled mesh
        f = np.zeros(a.r)
        f[1::3] = -10000.0
        for g in (a, b):
            g.reset_to_rest()
        o.set_state(np.zeros(o.r), np.zeros(o.r))
        for _ in range(3):
            for g in (a, b, o):
                g.set_external_forces(f)
            it_a, it_b, it_o = a.do_timestep(), b.do_timestep(), abs(o.step())
            assert abs(it_a - it_o) <= max(3, 0.02 * it_o) and abs(it_a - it_b) <= max(3, 0.02 * it_b)
        qa, qb, qo = a.get_q_state()[0], b.get_q_state()[0], o.get_state()[0]
        tol = 2e-5 if prec == fl.FB_MATRIX_F64 else 2e-4
        assert np.abs(qa - qo).max() <= tol * np.abs(qo).max() and np.abs(qa - qb).max() <= tol * np.abs(qb).max() and not qa[fixed].any()
        # a uniform force is a per-node fill in any order; the floor counts the same nodes
        for g in (a, b):
            g.reset_to_rest()
            g.set_uniform_force(1, -10000.0)
            g.do_timestep()
        assert a.floor_collision(-0.05, 0.4) == b.floor_collision(-0.05, 0.4) > 0
        # new constraints, by caller DOF
        fixed2 = fixed_vertices_to_dofs(np.sort(np.asarray(m)[np.arange(n * n)]))[: 3 * n]
        for g in (a, b):
            g.set_constrained_dofs(fixed2)
            g.reset_to_rest()
            g.set_uniform_force(1, -100.0)
            g.do_timestep()
        qa, qb = a.get_q_state()[0], b.get_q_state()[0]
        assert not qa[fixed2].any() and np.abs(qa - qb).max() <= tol * np.abs(qb).max()
        a.close()
        b.close()


def test_device_order_is_the_host_restatement(gpu):
    """renumber.hip (bounding box, keys, radix sort on the device) against renumber.h's host restatement (fb_plan_slab_order), on an
    unstructured shipped mesh and a scrambled brick with three different extents"""
    L = fl.lib()
    gold = np.load(os.path.join(GOLD, "fem_peanut.npz"))
    v0, t0 = truth_cube(7, 12, 9, 0.1)
    rng = np.random.default_rng(2)
    m = rng.permutation(len(v0))
    vb, tb, _ = scramble(v0, t0, [0], m)
    for v, t in ((np.ascontiguousarray(gold["verts"], dtype=np.float64), np.ascontiguousarray(gold["tets"], dtype=np.int32)), (vb, tb)):
        g = FemIntegrator(v, t, (), renumber=fl.FB_RENUMBER_ON)
        on, sc, si = g.renumbering()
        o = np.empty(len(v), np.int32)
        a, b = C.c_int(0), C.c_int(0)
        fl.check(L.fb_plan_slab_order(len(v), fl.dptr(v), len(t), fl.iptr(t), fl.iptr(o), C.byref(a), C.byref(b)))
        assert on and np.array_equal(g.owned_nodes(), o) and (sc, si) == (a.value, b.value)
        g.close()
    # the brick: y is the longest axis, then z, then x
    assert np.array_equal(vb[o], v0[np.lexsort((v0[:, 0], v0[:, 2], v0[:, 1]))])


def test_shipped_unstructured_meshes_renumbered_against_the_reference_golden(gpu):
    """Vega's beam and FemBrain's peanut (TetGen meshes, surface vertices first): the renumbered handle against the vectors of the
    reference build (tests/golden/fem_beam3.npz, fem_peanut.npz), tolerances of the un-renumbered tests"""
    gold = np.load(os.path.join(GOLD, "fem_beam3.npz"))
    fixed = fixed_vertices_to_dofs(gold["fixed_vertices"])
    g = FemIntegrator(gold["verts"], gold["tets"], fixed, matrix_precision=fl.FB_MATRIX_F64, renumber=fl.FB_RENUMBER_ON)
    assert g.renumbering()[0]
    f = np.zeros(g.r)
    f[1::3] = -10.0
    for k in range(3):
        g.set_external_forces(f)
        it = g.do_timestep()
        q = g.get_q_state()[0]
        assert abs(it - int(gold["iters"][k])) <= max(5, 0.03 * int(gold["iters"][k]))
        assert np.abs(q - gold["q"][k]).max() <= 2e-5 * np.abs(gold["q"][k]).max()
    g.close()
    gold = np.load(os.path.join(GOLD, "fem_peanut.npz"))
    fixed = fixed_vertices_to_dofs(gold["fixed_vertices"])
    for prec, tol in ((fl.FB_MATRIX_F64, 2e-5), (fl.FB_MATRIX_F32, 3e-4)):
        g = FemIntegrator(gold["verts"].astype(np.float64), gold["tets"], fixed, matrix_precision=prec, renumber=fl.FB_RENUMBER_ON)
        on, sc, si = g.renumbering()
        assert on and si < sc // 2, (sc, si)      # TetGen order: the widest element spans most of the list
        f = np.zeros(g.r)
        f[1::3] = -10000.0
        for k in range(2):
            g.set_external_forces(f)
            it = g.do_timestep()
            q = g.get_q_state()[0]
            ref = gold["q_ref_load"][k]
            assert abs(it - int(gold["it_ref_load"][k])) <= max(5, 0.03 * int(gold["it_ref_load"][k]))
            assert np.abs(q - ref).max() <= tol * np.abs(ref).max()
        g.close()


def test_auto_keeps_small_and_banded_meshes_in_the_callers_order(gpu):
    """AUTO leaves alone: any mesh below 8,192 nodes, and a grid-ordered cube of any size (every existing parity test runs on the
    caller's numbering, bit for bit as before)"""
    v, t = truth_cube(12, 12, 12, 0.1)
    rng = np.random.default_rng(0)
    vs, ts, _ = scramble(v, t, [0], rng.permutation(len(v)))
    g = FemIntegrator(vs, ts, ())
    assert g.renumbering()[0] is False
    g.close()
    v, t = truth_cube(24, 24, 24, 0.1)
    g = FemIntegrator(v, t, ())
    on, sc, si = g.renumbering()
    assert not on and 24 * 24 <= sc == si <= 24 * 24 + 24 + 1
    g.close()


def test_scrambled_1M_tet_cube_gets_the_fast_path_and_the_reference_answer(gpu):
    """BASELINE config 4's mesh in a random node order (what a caller's numbering may be after cuts): AUTO renumbers it, the handle
    runs the same kernel as on the grid-ordered cube -- 16-bit column words, a neighbour-only producer list -- and reproduces the
    first reference-load step of the reference's own CorotationalLinearFEM + CGSolver (tests/golden/fem_cube56_step1.npz) in the
    caller's numbering.  The same mesh with the renumbering switched off polls every workgroup."""
    n = 56
    gold = np.load(os.path.join(GOLD, "fem_cube%d_step1.npz" % n))
    v0, t0 = truth_cube(n, n, n, 0.1)
    rng = np.random.default_rng(12345)
    m = rng.permutation(len(v0))
    v, t, fx = scramble(v0, t0, cube_fixed_plane_i0(n, n), m)
    fixed = fixed_vertices_to_dofs(fx)
    g = FemIntegrator(v, t, fixed)
    on, sc, si = g.renumbering()
    assert on and sc > len(v0) // 2 and n * n <= si <= n * n + n + 1
    p = g.pcg_path()
    assert p["kernel"] == "k_pcg_pipe<float,c16,12,6>" and 0 < p["max_producers"] <= 64, p
    g.set_uniform_force(1, -10000.0)
    it = g.do_timestep()
    q, qv, _ = g.get_q_state()
    want = int(abs(gold["iters"]))
    assert abs(it - want) <= max(3, 0.02 * want)
    idx = gold["idx"]                                     # DOFs of the grid-ordered cube
    mine = 3 * m[idx // 3] + idx % 3                      # the same DOFs in the caller's numbering
    assert np.abs(q[mine] - gold["q"]).max() <= 2e-4 * float(gold["q_maxabs"])
    assert np.abs(qv[mine] - gold["qvel"]).max() <= 2e-4 * float(gold["qvel_maxabs"])
    assert abs(np.linalg.norm(q) - float(gold["q_norm"])) <= 2e-4 * float(gold["q_norm"])
    assert g.pcg_path()["path"] == fl.FB_PCG_PATH_PERSISTENT and not q[fixed].any()
    us_on = g.last.solve_seconds / it * 1e6
    # a re-sync to the grid-ordered mesh drops the renumbering, one back to the scrambled mesh finds it again
    g.resync(v0, t0, fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n)))
    assert g.renumbering()[0] is False
    g.resync(v, t, fixed)
    assert g.renumbering()[0] is True
    g.set_uniform_force(1, -10000.0)
    assert g.do_timestep() == it and np.array_equal(g.get_q_state()[0], q)
    g.close()
    off = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_OFF)
    assert off.pcg_path()["max_producers"] == -1 and off.pcg_path()["kernel"] == "k_pcg_pipe<float,c32,12,6>"
    off.set_uniform_force(1, -10000.0)
    it_off = off.do_timestep()
    us_off = off.last.solve_seconds / it_off * 1e6
    assert abs(it_off - it) <= max(3, 0.02 * it) and np.abs(off.get_q_state()[0] - q).max() <= 2e-5 * np.abs(q).max()
    assert us_on < 0.6 * us_off, (us_on, us_off)          # measured: 15.8 vs 47 us per iteration
    off.close()


def test_polygonizer_mesh_handed_over_on_the_device_can_be_renumbered(gpu):
    """fb_fem_create_from_poly (float positions and ids on the device) with the renumbering forced: same steps as without"""
    from fembrain_amd.blobtree import sphere_blob
    from fembrain_amd.poly import GpuPoly
    p = GpuPoly(sphere_blob())
    xyz, tets = p.run_tetrahedralizer(0.1)
    low = np.nonzero(xyz[:, 1] < -0.35)[0]
    fixed = fixed_vertices_to_dofs(low)
    os.environ["FEMBRAIN_RENUMBER"] = "1"
    try:
        a = FemIntegrator.from_poly(p, fixed)
    finally:
        del os.environ["FEMBRAIN_RENUMBER"]
    b = FemIntegrator.from_poly(p, fixed)
    assert a.renumbering()[0] and not b.renumbering()[0]
    for g in (a, b):
        g.set_uniform_force(1, -10000.0)
        g.do_timestep()
    qa, qb = a.get_q_state()[0], b.get_q_state()[0]
    assert np.abs(qa - qb).max() <= 2e-5 * np.abs(qb).max() and not qa[fixed].any()
    a.close()
    b.close()


@pytest.mark.parametrize("kw", [dict(integrator=fl.FB_INTEGRATOR_NEWMARK), dict(exact_tangent=True, matrix_precision=fl.FB_MATRIX_F64), dict(linear=True),
                                dict(pcg_variant=fl.FB_PCG_REFERENCE), dict(pcg_variant=fl.FB_PCG_BLOCK_JACOBI), dict(spmv_kernel=fl.FB_SPMV_ROWS)])
def test_every_force_model_and_solver_option_works_on_a_renumbered_handle(gpu, kw):
    """SURVEY 8f-4's options (warp = 0 / 2, the Newmark step) and the solver variants on a scrambled cube with the renumbering forced: three
    steps equal to those of a handle that keeps the caller's order (both stop at the same tolerance; tolerances of the un-renumbered tests)"""
    n = 10
    v0, t0 = truth_cube(n, n, n, 0.1)
    m = np.random.default_rng(8).permutation(len(v0))
    v, t, fx = scramble(v0, t0, cube_fixed_plane_i0(n, n), m)
    fixed = fixed_vertices_to_dofs(fx)
    a = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_ON, **kw)
    b = FemIntegrator(v, t, fixed, renumber=fl.FB_RENUMBER_OFF, **kw)
    assert a.renumbering()[0] and not b.renumbering()[0]
    f = np.zeros(a.r)
    f[1::3] = -3000.0
    f[0::3] = 200.0 * np.sin(7.0 * v[:, 2])
    for _ in range(3):
        for g in (a, b):
            g.set_external_forces(f)
        ia, ib = a.do_timestep(), b.do_timestep()
        assert abs(ia - ib) <= max(3, 0.02 * ib), (kw, ia, ib)
    (qa, wa, aa), (qb, wb, ab) = a.get_q_state(), b.get_q_state()
    tol = 2e-5 if kw.get("matrix_precision") == fl.FB_MATRIX_F64 else 2e-4
    assert np.abs(qa - qb).max() <= tol * np.abs(qb).max() and np.abs(wa - wb).max() <= 5 * tol * np.abs(wb).max() and not qa[fixed].any()
    if "integrator" in kw:
        assert np.abs(aa - ab).max() <= 20 * tol * np.abs(ab).max()
    a.close()
    b.close()
