"""The C++ class surfaces (include/fembrain/*.h) compile against the C ABI with a plain host compiler (CPU test) and a
host program using them produces the oracle's numbers on the GPU (gpu test)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_classes")


def _build():
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_classes.cpp"),
           "-o", EXE, "-L", os.path.join(ROOT, "fembrain_amd"), "-lfembrain_hip", "-Wl,-rpath," + os.path.join(ROOT, "fembrain_amd")]
    subprocess.check_call(cmd)


def test_cpp_adaptors_compile_and_link_with_gxx():
    _build()  # a C++11 host compiler and the C ABI are all the host application needs
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_program_matches_oracle(gpu):
    from fembrain_amd.blobtree import sphere_blob
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    from oracle.pyfield import OrcPoly
    from oracle.pyoracle import OrcFem
    _build()        # (always: a binary left from before a change of include/fembrain_hip.h -- fb_step_info grew in round 5 -- would run with the old layout)
    out = subprocess.check_output([EXE], text=True)
    kv = dict(line.split("=", 1) for line in out.strip().splitlines())
    n = 5
    v, t = truth_cube(n, n, n, 0.1)
    o = OrcFem(v, t)
    o.integrator(fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n)))
    f = np.zeros(o.r)
    f[1::3] = -10000.0
    for _ in range(2):
        o.set_external_forces(f)
        it = abs(o.step())
    q, _ = o.get_state()
    assert int(kv["CUBE_DOF"]) == o.r and int(kv["CUBE_CALLBACKS"]) == 2
    assert abs(float(kv["CUBE_MAXQ"]) - np.abs(q).max()) <= 2e-4 * np.abs(q).max()
    assert abs(int(kv["CUBE_ITERS"]) - it) <= 3
    assert abs(float(kv["CUBE_VOL0"]) - 0.4 ** 3) < 1e-12
    # ForceModel / CGSolver seams: force and stiffness norms at u = 1e-3 sin(0.37 i) against the oracle; the PCG solution
    # of Keff x = b (constrained rows are identity) satisfies the system through the black-box product
    uu = 1e-3 * np.sin(0.37 * np.arange(o.r))
    fo, Ko = o.assemble(uu)
    assert int(kv["SEAM_BLOCKS"]) == len(o.blocks()[1])
    assert abs(float(kv["SEAM_FNORM"]) - np.linalg.norm(fo)) <= 1e-9 * np.linalg.norm(fo)
    assert abs(float(kv["SEAM_KNORM"]) - np.linalg.norm(Ko)) <= 5e-7 * np.linalg.norm(Ko)
    assert int(kv["SEAM_CG_ITERS"]) > 0 and float(kv["SEAM_RESIDUAL"]) < 1e-6
    # IntegratorBaseSparse::GetTotalMass = 3 rho V (inflated consistent mass), GetKineticEnergy = 1/2 qvel^T M qvel
    assert abs(float(kv["TOTAL_MASS"]) - 3 * 1000.0 * 0.4 ** 3) < 1e-6 * 192
    import scipy.sparse as sp
    ia, ja = o.csr()
    M = sp.csr_matrix((o.mass_on_pattern(), ja, ia), shape=(o.r, o.r))
    _, qv = o.get_state()
    ke = 0.5 * qv @ (M @ qv)
    assert abs(float(kv["KINETIC"]) - ke) <= 1e-3 * ke
    cur = v + q.reshape(-1, 3)
    assert int(kv["PICKED"]) == int(np.argmin(((cur - np.array([10.0, 0.2, 10.0])) ** 2).sum(1)))
    inbox = ((cur >= np.array([-1.0, -1.0, -1.0])) & (cur <= np.array([-0.24, 1.0, 1.0]))).all(1)
    assert int(kv["PICK_BOX"]) == int(inbox.sum()) >= n * n   # the clamped plane and whatever sagged into the box
    assert int(kv["VOL_CHANGED"]) == 1 and int(kv["HAPTIC_FIXED"]) == 0   # a clamped vertex cannot be pulled
    op = OrcPoly(sphere_blob())
    ox, ot, oc = op.run_tetrahedralizer(0.1)
    sx, sn, st = op.surface()
    assert int(kv["SURF_VERTS"]) == len(sx) and int(kv["SURF_TRIS"]) == len(st) and int(kv["SURF_APPLIED"]) == 1
    assert abs(float(kv["SURF_SHIFT"]) - 0.25) < 1e-6
    assert int(kv["SURF_VOXELS"]) == int(kv["SURF_VOXELS_COUNT"]) == int(((op.config != 0) & (op.config != 255)).sum())
    assert int(kv["POLY_RUN"]) == 1 and int(kv["POLY_TETS"]) == len(ot) == 3744 and int(kv["POLY_VERTS"]) == len(ox)
    assert np.float32(kv["FIELD_025"]) == np.float32((1 - 0.0625) ** 3) and int(kv["GRID_POINTS"]) == 12 ** 3
    assert int(kv["BALL_FIXED"]) == int((ox[:, 1] < -0.35).sum()) and int(kv["BALL_ITERS"]) > 0
    assert kv["STAT"] == "%d,%d,TET,JACOBI PRECONDITIONED CG" % (len(t), len(v)) and abs(float(kv["STAT_VOL"]) - 0.4 ** 3) < 1e-12
    assert kv["COLLIDE_NOFLOOR"] == "0"
    assert int(kv["DIRECT_R"]) == 3 * len(ox) and kv["DIRECT_SAME"] == "1"
    # Deformable::syncForceModelDelta (fb_fem_resync_delta): list updated, and the step of a Deformable made from the resulting mesh
    assert kv["DELTA_PATH"] == "1" and int(kv["DELTA_NODES"]) == 126 and int(kv["DELTA_CELLS"]) == 384 + 3 and kv["DELTA_SAME"] == "1"
    assert int(kv["SLAB_VERTS"]) == len(ox) and int(kv["SLAB_TETS"]) == len(ot) and kv["SLAB_SAME"] == "1"
    # PS::FEM::Cutting on the ball: the reference's known answer, hit list = count, the swept quad closes at the third call
    assert kv["CUT_KNOWN"] == "0,0,0,1"
    assert int(kv["CUT_FACES"]) == int(kv["CUT_FACE_IDS"]) > 0 and int(kv["CUT_FACE_FLAGS"]) == 4 * len(ot)
    assert kv["CUT_QUAD"] == "01" and int(kv["CUT_EDGES"]) > 0 and float(kv["CUT_QUAD_X"]) == -0.1
    assert int(kv["CUT_NODES"]) == len(ox) and int(kv["CUT_CELLS"]) == len(ot)


def test_cpp_blob_and_veg_readers_match_the_python_readers(tmp_path):
    """include/fembrain/BlobReader.h (for C++ hosts that do not link the reference's ModelReader / VolMeshIO) against
    fembrain_amd/blobtree.py and meshgen.read_veg on every fixture model, instanced ones included.  Host only."""
    import glob
    from fembrain_amd.blobtree import read_blob
    from fembrain_amd.meshgen import read_veg, truth_cube
    from fembrain_amd.poly import write_veg
    exe = os.path.join(ROOT, "tests", "cpp", "read_models")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "read_models.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "fembrain_amd"), "-lfembrain_hip", "-Wl,-rpath," + os.path.join(ROOT, "fembrain_amd")])
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "blob", "*.blob")))
    assert len(files) >= 10
    for f in files:
        out = subprocess.check_output([exe, "blob", f], text=True)
        arr = {ln.split()[0]: np.array(ln.split()[2:], dtype=np.float32) for ln in out.strip().splitlines()}
        b = read_blob(f)
        for name, want in (("header", b.header), ("ops", b.ops), ("prims", b.prims), ("mtx", b.mtx), ("pbox", np.array(b.prim_boxes))):
            want = np.asarray(want, np.float32).reshape(-1)
            assert arr[name].shape == want.shape, (f, name)
            assert np.allclose(arr[name], want, rtol=2e-6, atol=2e-6), (f, name, np.abs(arr[name] - want).max())
    v, t = truth_cube(4, 4, 4, 0.1)
    p = str(tmp_path / "cube.veg")
    write_veg(p, v, t)
    out = subprocess.check_output([exe, "veg", p], text=True).strip().splitlines()
    cv = np.array(out[0].split()[2:], dtype=np.float64).reshape(-1, 3)
    ce = np.array(out[1].split()[2:], dtype=np.int32).reshape(-1, 4)
    pv, pt = read_veg(p)
    assert np.array_equal(ce, pt) and np.array_equal(ce, t) and np.allclose(cv, pv, rtol=0, atol=1e-12)
    bad = tmp_path / "bad.blob"
    bad.write_text("[Global]\nFileVersion=6\nRootIDs=(0)\n[BLOBNODE 0]\nIsOperator=0\nPrimitiveType=TORUS\n")
    assert subprocess.run([exe, "blob", str(bad)], capture_output=True, text=True).returncode == 1


SEAM = os.path.join(ROOT, "oracle", "_ref", "ref_seam")


def test_adaptors_derive_from_the_reference_abstract_classes_and_link_with_its_translation_units():
    """include/fembrain/VegaAdaptors.h (HipCorotationalForceModel : ForceModel, HipVolumeConservingIntegrator :
    IntegratorBaseSparse, the CGSolver black-box product) compiled against the reference's OWN headers and linked with its
    own sparseMatrix.cpp / CGSolver.cpp / forceModel.cpp / integratorBase*.cpp / corotationalLinearFEM.cpp / tetMesh.cpp
    (oracle/Makefile: _ref/ref_seam).  Where the reference tree is present it is rebuilt here; elsewhere the prebuilt binary
    (it travels like the other oracle/_ref artefacts) is inspected.  Running it needs a GPU (next test)."""
    if os.path.isdir("/root/reference/src/3rdparty/vegafem"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-B", os.path.join(ROOT, "oracle", "_ref", "ref_seam")])
    if not os.path.exists(SEAM):
        pytest.skip("oracle/_ref/ref_seam was not built (no reference tree here)")
    syms = subprocess.check_output(["nm", "-C", SEAM], text=True)
    # the reference's classes and ours in one image, ours overriding the reference's virtuals
    for needle in ("SparseMatrix::MultiplyVector", "CGSolver::SolveLinearSystemWithoutPreconditioner", "IntegratorBaseSparse::GetTotalMass",
                   "CorotationalLinearFEM::ComputeForceAndStiffnessMatrix", "vtable for PS::FEM::HipCorotationalForceModel",
                   "vtable for PS::FEM::HipVolumeConservingIntegrator", "PS::FEM::HipVolumeConservingIntegrator::DoTimestep",
                   "PS::FEM::hipBlackBoxProduct", "U fb_fem_step", "U fb_fem_spmv", "U fb_fem_assemble"):
        assert needle in syms, needle
    out = subprocess.run([SEAM], capture_output=True, text=True)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode == 1 and "ERROR=no device" in out.stdout   # no CPU fallback behind the seam either


@pytest.mark.gpu
def test_reference_classes_drive_the_hip_path_through_the_seam(gpu):
    """oracle/_ref/ref_seam on the GPU: the reference's CorotationalLinearFEM vs HipCorotationalForceModel through the
    ForceModel interface (same pattern, f and K, warp 1 and 2), the reference's CGSolver loop over fb_fem_spmv, and three
    DoTimestep() calls through an IntegratorBaseSparse pointer against the oracle."""
    if not os.path.exists(SEAM):
        pytest.skip("oracle/_ref/ref_seam did not travel")
    from fembrain_amd.meshgen import cube_fixed_plane_i0, fixed_vertices_to_dofs, truth_cube
    from oracle.pyoracle import OrcFem
    out = subprocess.check_output([SEAM], text=True)
    kv = dict(line.split("=", 1) for line in out.strip().splitlines() if "=" in line and not line.startswith("eps="))
    for w in ("1", "2"):
        assert kv["WARP%s_PATTERN" % w] == "1"
        assert float(kv["WARP%s_F_RELDIFF" % w]) < 1e-9 and float(kv["WARP%s_K_RELDIFF" % w]) < 1e-10, kv
        # the force alone is served by the handle the integrator steps on (fp32-STORED matrix; f itself is formed in fp64)
        assert float(kv["WARP%s_F_STEPPING_HANDLE_RELDIFF" % w]) < 1e-9, kv
    assert kv["MATRIX_HANDLE_CREATED"] == "0"   # (the model the integrator stepped on was never asked for a matrix: no fp64 handle was made)
    # the reference's own ForceModel::TestStiffnessMatrix (forceModel.cpp:47-109) on the warp = 2 device model: its lines
    # "eps=E: maxEntry=M ..." -- f(q + eps dq) - f(q) - K eps dq must shrink like eps^2 until rounding takes over
    fd = {}
    for line in out.splitlines():
        if line.startswith("eps="):
            e, m = line.split(":")[0][4:], line.split("maxEntry=")[1].split()[0]
            fd[float(e)] = float(m)
    assert len(fd) >= 10, out[-3000:]
    for e in (1e-1, 1e-2, 1e-3):
        lo = min(k for k in fd if abs(k / (e / 10) - 1) < 1e-6)
        hi = min(k for k in fd if abs(k / e - 1) < 1e-6)
        assert 50 < fd[hi] / fd[lo] < 200, (e, fd)          # second order: a factor ~100 per decade
    # the reference's SparseMatrix::CheckLinearSystemSolution (sparseMatrix.cpp:1560-1592) of the device's Jacobi-PCG solution (eps 1e-6)
    assert int(kv["DEVICE_PCG_ITERS"]) > 0 and float(kv["REF_CHECKLINEARSYSTEM_RELINF"]) < 1e-4, kv
    n = 5
    v, t = truth_cube(n, n, n, 0.1)
    o = OrcFem(v, t)
    o.integrator(fixed_vertices_to_dofs(cube_fixed_plane_i0(n, n)))
    f = np.zeros(o.r)
    f[1::3] = -10000.0
    for k in range(3):
        o.set_external_forces(f)
        it = abs(o.step())
        q, _ = o.get_state()
        assert kv["STEP%d_RC" % k] == "0"
        assert abs(float(kv["STEP%d_QNORM" % k]) - np.linalg.norm(q)) <= 2e-5 * np.linalg.norm(q)
        assert abs(int(kv["STEP%d_ITERS" % k]) - it) <= max(3, 0.02 * it)
    assert abs(float(kv["TOTAL_MASS"]) - 3 * 1000.0 * 0.4 ** 3) < 1e-6 * 192
    assert int(kv["REFCG_INFO"]) > 0 and float(kv["REFCG_RESIDUAL"]) < 1e-6


@pytest.mark.gpu
def test_link_unchanged_route_runs_the_measured_path(gpu):
    """`ref_seam --bench 40`: steps from the rest state through an IntegratorBaseSparse* (the reference's abstract class, what
    Deformable.cpp:205-214 holds) run the persistent solver of the fp32-stored handle -- the path bench.py measures -- without ever
    creating the fp64 matrix handle; iteration count of the 40^3 cube as the two-launch solver finds it (1413)."""
    if not os.path.exists(SEAM):
        pytest.skip("oracle/_ref/ref_seam did not travel")
    out = subprocess.check_output([SEAM, "--bench", "40", "3"], text=True)
    kv = dict(line.split("=", 1) for line in out.strip().splitlines() if "=" in line)
    assert kv["BENCH_PCG_PATH"] == "1" and kv["BENCH_MATRIX_HANDLE"] == "0", kv
    assert abs(float(kv["BENCH_ITERS_PER_STEP"]) - 1413) <= 28 and float(kv["BENCH_STEPS_PER_S"]) > 5, kv
