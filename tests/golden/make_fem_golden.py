"""Generates the FEM golden vectors under tests/golden/ by running the REFERENCE's own translation units
(oracle/_ref/libfem_ref.so = VegaFEM from /root/reference compiled where it lies, driven by oracle/ref_harness.cpp).
Run here once (`python tests/golden/make_fem_golden.py`); the .npz files are committed, the reference is not needed
at test time.

  fem_cube5.npz   truth cube 5^3 (125 nodes / 384 tets, plane i=0 clamped): per-element K0 / M^-1, stiffness pattern,
                  assembled f and K at a seeded displacement, Keff / rhs / PCG solution of one step from a seeded
                  state, q and qvel after 3 steps under the reference load (-10000/y-DOF) and a gentle one (-10)
  fem_cube5_linear.npz   the same cube with warp = 0 (linear elasticity): assembled f and K at a seeded displacement, q after 2 steps
  fem_cube5_warp2.npz    warp = 2 (exact tangent stiffness): assembled f and K at a seeded displacement
  fem_cube5_newmark.npz  ImplicitNewmarkSparse::DoTimestep (restated on the reference's objects): q, qvel, qaccel after 3 steps, 1 and 3 Newton iterations
  fem_disc.npz, fem_pyramid.npz   the other two tet meshes the reference ships (data/models/disc/disc.1.veg: 68 nodes / 121 tets,
                  data/models/pyramid/pyramid.1.veg: 34 / 32; written by FemBrain from TetGen output): mesh, f and K at a seeded
                  displacement, q / qvel after 3 steps with -10 per y-DOF, the nodes of the lowest quarter in x clamped
  fem_peanut.npz  data/models/blobtree/peanut.veg (3,224 nodes / 12,947 tets: a mesh FemBrain itself simulates -- its polygonizer's
                  surface vertices tetrahedralized by TetGen): mesh (float32 positions as the file prints them), q after 2 steps under
                  the reference load (-10000 per y-DOF) and under -10, the tenth of the nodes with the lowest y clamped
  fem_cube56_step1.npz, fem_cube58_step1.npz   (round 3, `python make_fem_golden.py cube 56`) the first reference-load step of the
                  56^3 / 58^3 truth cubes (998,250 / 1,111,158 tets): iteration count, norms, q and qvel at 2,000 seeded DOFs
  fem_{tumor,dumbel,dumbelclose,eggshell,implicit_sphere}.npz   (round 3, `python make_fem_golden.py round3`) the remaining tet meshes the
                  reference ships as polygonizer output: mesh, q after 2 steps under the reference load and under -10, iteration counts
  fem_beam3.npz   data/models/beam3/beam3_tet.veg (208 nodes / 450 tets, Vega's own sample) with beam3.bou clamps:
                  mesh, the reference's consistent mass matrix file beam3_tet.mass (a known answer shipped by the
                  reference), and q after 3 steps with -10 per y-DOF
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, read_veg, truth_cube
from oracle.pyoracle import RefFem

REF = "/root/reference/data/models"


def steps(v, t, fixed, load, n=3, eps=1e-6):
    r = RefFem(v, t)
    r.integrator(fixed)
    f = np.zeros(r.r)
    f[1::3] = load
    qs, vs, its = [], [], []
    for _ in range(n):
        r.set_external_forces(f)
        its.append(r.step(cg_eps=eps))
        q, qv = r.get_state()
        qs.append(q)
        vs.append(qv)
    return np.array(qs), np.array(vs), np.array(its)


def cube5():
    n = 5
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    r = RefFem(v, t)
    els = np.array([0, 1, 2, 3, 4, 5, 100, 383])
    K0 = np.array([r.K0(int(e)) for e in els])
    Mi = np.array([r.Minv(int(e)) for e in els])
    ia, ja = r.csr()
    mia, mja, ma = r.mass_csr()
    rng = np.random.default_rng(12345)
    u = rng.normal(size=r.r) * 0.01
    f, K = r.assemble(u)
    r.integrator(fixed)
    q0 = rng.normal(size=r.r) * 0.004
    v0 = rng.normal(size=r.r) * 0.1
    q0[fixed] = 0
    v0[fixed] = 0
    fext = np.zeros(r.r)
    fext[1::3] = -10.0
    r.set_state(q0, v0)
    r.set_external_forces(fext)
    info, keff, rhs, dv = r.step(cg_eps=1e-12, cg_maxiter=20000, want=True)
    q1, v1 = r.get_state()
    qa, va, ia_ = steps(v, t, fixed, -10000.0)
    qb, vb, ib_ = steps(v, t, fixed, -10.0)
    np.savez_compressed(os.path.join(HERE, "fem_cube5.npz"), n=n, fixed=fixed, els=els, K0=K0, Minv=Mi, ia=ia, ja=ja,
                        mass_ia=mia, mass_ja=mja, mass_a=ma, u=u, f=f, K=K, q0=q0, v0=v0, keff=keff, rhs=rhs, dv=dv, cg_info=info,
                        q1=q1, v1=v1, q_ref_load=qa, v_ref_load=va, it_ref_load=ia_, q_gentle=qb, v_gentle=vb, it_gentle=ib_)
    print("cube5: nnz", len(ja), "cg", info, "iters", ia_, ib_)


def cube5_linear():
    """warp = 0 (the linear branch of ComputeForceAndStiffnessMatrix): assembled f, K and 2 steps under the gentle load"""
    n = 5
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    r = RefFem(v, t)
    r.set_linear(True)
    rng = np.random.default_rng(54321)
    u = rng.normal(size=r.r) * 0.01
    f, K = r.assemble(u)
    r.integrator(fixed)
    fext = np.zeros(r.r)
    fext[1::3] = -10.0
    qs, its = [], []
    for _ in range(2):
        r.set_external_forces(fext)
        its.append(r.step(cg_eps=1e-6))
        qs.append(r.get_state()[0])
    np.savez_compressed(os.path.join(HERE, "fem_cube5_linear.npz"), n=n, fixed=fixed, u=u, f=f, K=K, q=np.array(qs), it=np.array(its))
    print("cube5 linear: |f|", np.linalg.norm(f), "iters", its)


def cube5_warp2():
    """warp = 2 (corotational with the exact tangent stiffness, corotationalLinearFEM.cpp:296-428): assembled f and K at a
    seeded displacement (K stays symmetric to rounding: |K - K^T| / |K| = 7e-15 here)"""
    n = 5
    v, t = truth_cube(n, n, n, 0.1)
    r = RefFem(v, t)
    r.set_warp(2)
    rng = np.random.default_rng(24680)
    u = rng.normal(size=r.r) * 0.01
    f, K = r.assemble(u)
    ia, ja = r.csr()
    np.savez_compressed(os.path.join(HERE, "fem_cube5_warp2.npz"), n=n, u=u, f=f, K=K, ia=ia, ja=ja)
    import scipy.sparse as sp
    A = sp.csr_matrix((K, ja, ia), shape=(r.r, r.r))
    print("cube5 warp2: |f|", np.linalg.norm(f), "|K - K^T| / |K|", abs(A - A.T).max() / abs(A).max())


def cube5_newmark():
    """ImplicitNewmarkSparse::DoTimestep restated on the reference's objects (oracle/ref_harness.cpp: ref_newmark_step), beta 1/4,
    gamma 1/2, reference load, plane i = 0 clamped: q, qvel, qaccel after each of 3 steps with 1 and with 3 Newton iterations"""
    n = 5
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    out = {}
    for mx in (1, 3):
        r = RefFem(v, t)
        r.integrator(fixed)
        f = np.zeros(r.r)
        f[1::3] = -10000.0
        qs, vs, acs, its = [], [], [], []
        for _ in range(3):
            r.set_external_forces(f)
            its.append(r.newmark_step(max_newton=mx))
            q, qv = r.get_state()
            qs.append(q); vs.append(qv); acs.append(r.get_accel())
        out["q_%d" % mx], out["qvel_%d" % mx], out["qaccel_%d" % mx], out["iters_%d" % mx] = np.array(qs), np.array(vs), np.array(acs), np.array(its)
        print("cube5 newmark, %d Newton iteration(s): (newton, pcg) per step" % mx, its)
    np.savez_compressed(os.path.join(HERE, "fem_cube5_newmark.npz"), n=n, fixed=fixed, **out)


def beam3():
    v, t = read_veg(os.path.join(REF, "beam3", "beam3_tet.veg"))
    bou = [int(x) for x in open(os.path.join(REF, "beam3", "beam3.bou")).read().replace("\n", "").split(",") if x.strip()]
    fixed_vertices = np.array(sorted(b - 1 for b in bou), np.int32)  # .bou is 1-indexed
    fixed = fixed_vertices_to_dofs(fixed_vertices)
    rows = []
    with open(os.path.join(REF, "beam3", "beam3_tet.mass")) as fh:
        toks = fh.read().split()
    nr, nc = int(toks[0]), int(toks[1])
    tri = np.array(toks[2:], dtype=np.float64).reshape(-1, 3)
    q, qv, its = steps(v, t, fixed, -10.0, n=3)
    np.savez_compressed(os.path.join(HERE, "fem_beam3.npz"), verts=v, tets=t, fixed_vertices=fixed_vertices, mass_n=nr,
                        mass_i=tri[:, 0].astype(np.int32), mass_j=tri[:, 1].astype(np.int32), mass_v=tri[:, 2], q=q, qvel=qv, iters=its)
    print("beam3:", v.shape, t.shape, "fixed", fixed_vertices, "iters", its, "mass entries", len(tri))


def shipped_meshes():
    """disc.1.veg and pyramid.1.veg: assembled f, K at a seeded displacement and 3 gentle steps of the reference build"""
    for name, rel in (("disc", "disc/disc.1.veg"), ("pyramid", "pyramid/pyramid.1.veg")):
        v, t = read_veg(os.path.join(REF, rel))
        fixed_vertices = np.nonzero(v[:, 0] <= v[:, 0].min() + 0.25 * (v[:, 0].max() - v[:, 0].min()))[0].astype(np.int32)  # the quarter of lowest x
        fixed = fixed_vertices_to_dofs(fixed_vertices)
        r = RefFem(v, t)
        ia, ja = r.csr()
        rng = np.random.default_rng(97)
        u = rng.normal(size=r.r) * 0.01 * (v.max() - v.min())
        f, K = r.assemble(u)
        # the disc is a thin plate of slivers (2,000+ PCG iterations on 204 DOFs): at the reference tolerance 1e-6 two correct solvers
        # agree only to ~1e-3, so its steps are solved to 1e-9
        eps = 1e-9 if name == "disc" else 1e-6
        q, qv, its = steps(v, t, fixed, -10.0, n=3, eps=eps)
        np.savez_compressed(os.path.join(HERE, "fem_%s.npz" % name), verts=v, tets=t, fixed_vertices=fixed_vertices, ia=ia, ja=ja, u=u, f=f, K=K,
                            q=q, qvel=qv, iters=its, cg_eps=eps)
        print(name, v.shape, t.shape, "fixed vertices", len(fixed_vertices), "iters", its, "|q|", np.abs(q[-1]).max())


def peanut():
    v, t = read_veg(os.path.join(REF, "blobtree", "peanut.veg"))
    v = v.astype(np.float32).astype(np.float64)  # (the file prints 6 digits)
    order = np.argsort(v[:, 1], kind="stable")
    fixed_vertices = np.sort(order[:len(v) // 10]).astype(np.int32)
    fixed = fixed_vertices_to_dofs(fixed_vertices)
    qa, _, ia_ = steps(v, t, fixed, -10000.0, n=2)
    qb, _, ib_ = steps(v, t, fixed, -10.0, n=2)
    np.savez_compressed(os.path.join(HERE, "fem_peanut.npz"), verts=v.astype(np.float32), tets=t.astype(np.int32), fixed_vertices=fixed_vertices,
                        q_ref_load=qa, it_ref_load=ia_, q_gentle=qb, it_gentle=ib_)
    print("peanut:", v.shape, t.shape, "fixed", len(fixed_vertices), "iters", ia_, ib_, "|q|", np.abs(qa[-1]).max(), np.abs(qb[-1]).max())


def shipped_tet_mesh(name, rel):
    """Round 3: the other tet meshes the reference ships as polygonizer output -- data/models/blobtree/{tumor,dumbel,dumbelclose,eggshell}.veg
    (GPUPoly surfaces tetrahedralized by TetGen, like peanut.veg) and data/models/sphere/implicit_sphere.veg (the tet polygonizer's own
    output: 624 cells of 6 tets, vertices not welded, so every cell is a body of its own) -- through the reference build: q after the
    second step under the reference load and under -10 per y-DOF, the tenth of the nodes with the lowest y clamped, iteration counts
    of both steps."""
    v, t = read_veg(os.path.join(REF, rel))
    v = v.astype(np.float32).astype(np.float64)  # (the files print 6 digits)
    order = np.argsort(v[:, 1], kind="stable")
    fixed_vertices = np.sort(order[:len(v) // 10]).astype(np.int32)
    fixed = fixed_vertices_to_dofs(fixed_vertices)
    r0 = RefFem(v, t)
    f0, K0 = r0.assemble(np.zeros(r0.r))
    if not (np.isfinite(K0).all() and np.isfinite(f0).all()):
        # tumor.veg: 524 of its 32,303 tets have zero volume at the 6 digits the file prints; inverse4x4 divides by zero and the
        # reference's K and f are NaN from the first assembly on (its PCG then "converges" in 0 iterations on NaN).  Recorded as such.
        p = v[t]
        vol = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0]) / 6
        np.savez_compressed(os.path.join(HERE, "fem_%s.npz" % name), verts=v.astype(np.float32), tets=t.astype(np.int32), fixed_vertices=fixed_vertices,
                            reference_K_finite=False, n_flat_tets=int((vol == 0).sum()), first_flat_tet=int(np.nonzero(vol == 0)[0][0]))
        print("%s: reference K is not finite (%d tets of exactly zero volume, first %d)" % (name, (vol == 0).sum(), np.nonzero(vol == 0)[0][0]))
        return
    qa, _, ia_ = steps(v, t, fixed, -10000.0, n=2)
    qb, _, ib_ = steps(v, t, fixed, -10.0, n=2)
    np.savez_compressed(os.path.join(HERE, "fem_%s.npz" % name), verts=v.astype(np.float32), tets=t.astype(np.int32), fixed_vertices=fixed_vertices,
                        q_ref_load=qa[-1].astype(np.float64), it_ref_load=ia_, q_gentle=qb[-1], it_gentle=ib_)
    print("%s:" % name, v.shape, t.shape, "fixed", len(fixed_vertices), "iters", ia_, ib_, "|q|", np.abs(qa[-1]).max(), np.abs(qb[-1]).max())


def cube_big(n):
    """Round 3: the FIRST step from rest of the n^3-node truth cube under the reference load (plane i = 0 clamped, -10000 per y-DOF,
    CG eps 1e-6) by the reference build -- 56^3 = BASELINE config 4 (998,250 tets), 58^3 = the largest cube of 12 slices per CU:
    PCG iteration count, |q|_2, |qvel|_2, max|q| and q, qvel at 2,000 seeded DOFs (one ~45 s reference run each)."""
    import time
    v, t = truth_cube(n, n, n, 0.1)
    fixed = fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n))
    t0 = time.time()
    r = RefFem(v, t)
    r.integrator(fixed)
    f = np.zeros(r.r)
    f[1::3] = -10000.0
    r.set_external_forces(f)
    t1 = time.time()
    it = r.step(cg_eps=1e-6)
    q, qv = r.get_state()
    idx = np.sort(np.random.default_rng(n).choice(r.r, size=2000, replace=False))
    np.savez_compressed(os.path.join(HERE, "fem_cube%d_step1.npz" % n), n=n, iters=it, q_norm=np.linalg.norm(q), qvel_norm=np.linalg.norm(qv),
                        q_maxabs=np.abs(q).max(), qvel_maxabs=np.abs(qv).max(), idx=idx, q=q[idx], qvel=qv[idx])
    print("cube%d: %d tets, set-up %.1f s, step %.1f s, %d PCG iterations, |q| %.6f, max|q| %.6f" % (n, len(t), t1 - t0, time.time() - t1, it,
          np.linalg.norm(q), np.abs(q).max()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "cube":
        cube_big(int(sys.argv[2]))
    elif len(sys.argv) > 1 and sys.argv[1] == "round3":
        for nm in ("tumor", "dumbel", "dumbelclose", "eggshell"):
            shipped_tet_mesh(nm, "blobtree/%s.veg" % nm)
        shipped_tet_mesh("implicit_sphere", "sphere/implicit_sphere.veg")
    elif len(sys.argv) > 1 and sys.argv[1] == "linear":
        cube5_linear()   # added later: leaves the other two files as they are
    elif len(sys.argv) > 1 and sys.argv[1] == "round2":
        cube5_warp2()    # round 2: the exact-tangent option and the Newmark step
        cube5_newmark()
    elif len(sys.argv) > 1 and sys.argv[1] == "shipped":
        shipped_meshes()
    elif len(sys.argv) > 1 and sys.argv[1] == "peanut":
        peanut()
    else:
        cube5()
        cube5_linear()
        beam3()
        shipped_meshes()
        peanut()
