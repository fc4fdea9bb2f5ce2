// TEST INFRASTRUCTURE.  The drop-in seam, proven against the reference's OWN headers and translation units: this program is
// compiled with -I<reference>/src/3rdparty/vegafem/... and linked with the reference's sparseMatrix.cpp, CGSolver.cpp,
// forceModel.cpp, integratorBase.cpp, integratorBaseSparse.cpp, corotationalLinearFEM.cpp, tetMesh.cpp ... (compiled where
// they lie by oracle/Makefile into oracle/_ref/ref_seam) and with libfembrain_hip.so.  include/fembrain/VegaAdaptors.h derives
// from the reference's ForceModel and IntegratorBaseSparse; here the reference's classes drive it:
//   1. HipCorotationalForceModel fills a reference SparseMatrix; compared entry by entry with the reference's own
//      CorotationalLinearFEM::ComputeForceAndStiffnessMatrix on the same TetMesh and displacement (warp 1 and warp 2)
//   2. the reference's CGSolver runs its own CG loop on the device matrix through its black-box product hook
//      (CGSolver.h:65-66) -> fb_fem_spmv
//   3. HipVolumeConservingIntegrator (an IntegratorBaseSparse) steps; |q| printed for the Python test to compare with the oracle
//   4. the reference's own self-checks run on the device path: ForceModel::TestStiffnessMatrix (forceModel.cpp:47-109: finite
//      differences of f against K, on the warp = 2 model whose K is the exact tangent) and
//      SparseMatrix::CheckLinearSystemSolution (sparseMatrix.cpp:1560-1592) of the device's PCG solution in the reference's matrix
//   `ref_seam --bench N [steps]`: steps/s of N^3-node truth cube steps from the rest state THROUGH AN IntegratorBaseSparse* (what
//   FemBrain's Deformable holds), host arrays crossing the boundary every step -- to set beside bench.py's value_at_fixed_state.
// Prints KEY=value lines.  Needs a GPU to RUN; compiling and linking it is the CPU-side proof.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "CGSolver.h"
#include "corotationalLinearFEM.h"
#include "generateMassMatrix.h"
#include "tetMesh.h"

#include "fembrain/VegaAdaptors.h"

static void truth_cube(int n, double cell, std::vector<double>& v, std::vector<int>& t) {  // VolMeshSamples::CreateTruthCube layout
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      for (int k = 0; k < n; k++) { v.push_back(cell * i); v.push_back(cell * j); v.push_back(cell * k); }
  static const int pat[6][4] = {{0, 2, 4, 1}, {6, 2, 1, 4}, {6, 2, 3, 1}, {6, 4, 1, 5}, {6, 1, 3, 5}, {6, 3, 7, 5}};  // LBN.. corner = 4dx+2dy+dz
  for (int i = 0; i < n - 1; i++)
    for (int j = 0; j < n - 1; j++)
      for (int k = 0; k < n - 1; k++) {
        int c[8];
        for (int q = 0; q < 8; q++) c[q] = ((i + ((q >> 2) & 1)) * n + (j + ((q >> 1) & 1))) * n + (k + (q & 1));
        for (int e = 0; e < 6; e++)
          for (int a = 0; a < 4; a++) t.push_back(c[pat[e][a]]);
      }
}

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

// steps/s through the reference's abstract integrator interface; every step starts from the rest state (SetqState) under the
// reference load, as bench.py's value_at_fixed_state does
static int bench(int n, int steps) {
  std::vector<double> v;
  std::vector<int> t;
  truth_cube(n, 0.1, v, t);
  const int nv = (int)v.size() / 3, ne = (int)t.size() / 4, r = 3 * nv;
  TetMesh mesh(nv, &v[0], ne, &t[0], 1e7, 0.46, 1000.0);
  SparseMatrix* M = NULL;
  GenerateMassMatrix::computeMassMatrix(&mesh, &M, true);
  std::vector<int> fixed;
  for (int j = 0; j < n * n; j++)
    for (int k = 0; k < 3; k++) fixed.push_back(3 * j + k);
  PS::FEM::HipCorotationalForceModel hip(&mesh, 1e7, 0.46, 1000.0);
  if (!hip.ok()) { printf("ERROR=no device\n"); return 1; }
  PS::FEM::HipVolumeConservingIntegrator hipInt(r, 0.0333, M, &hip, 0, (int)fixed.size(), &fixed[0], 0.0, 0.01);
  IntegratorBaseSparse* integ = &hipInt;
  std::vector<double> fext(r, 0.0), zero(r, 0.0);
  for (int i = 0; i < nv; i++) fext[3 * i + 1] = -10000.0;
  double total = 0.0;
  int iters = 0;
  for (int s = -2; s < steps; s++) {  // two warm-up steps
    integ->SetqState(&zero[0], &zero[0], &zero[0]);
    const double t0 = now_s();
    integ->SetExternalForcesToZero();
    integ->SetExternalForces(&fext[0]);
    if (integ->DoTimestep() != 0) { printf("ERROR=step failed\n"); return 1; }
    const double dt = now_s() - t0;
    if (s >= 0) { total += dt; iters += hipInt.lastIterations(); }
  }
  double qn = 0;
  for (int i = 0; i < r; i++) qn += integ->Getq()[i] * integ->Getq()[i];
  printf("BENCH_TETS=%d\nBENCH_STEPS=%d\nBENCH_STEPS_PER_S=%.4f\nBENCH_MS_PER_STEP=%.4f\nBENCH_ITERS_PER_STEP=%.1f\nBENCH_PCG_PATH=%d\nBENCH_QNORM=%.10e\n"
         "BENCH_SOLVE_MS=%.4f\nBENCH_ASSEMBLY_MS=%.4f\nBENCH_MATRIX_HANDLE=%d\n",
         ne, steps, steps / total, total / steps * 1e3, (double)iters / steps, hipInt.lastPcgPath(), sqrt(qn), integ->GetSystemSolveTime() * 1e3,
         integ->GetForceAssemblyTime() * 1e3, hip.matrixHandle() != NULL);
  delete M;
  return 0;
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  if (argc >= 3 && !strcmp(argv[1], "--bench")) return bench(atoi(argv[2]), argc >= 4 ? atoi(argv[3]) : 5);
  const int n = 5;
  std::vector<double> v;
  std::vector<int> t;
  truth_cube(n, 0.1, v, t);
  const int nv = (int)v.size() / 3, ne = (int)t.size() / 4, r = 3 * nv;
  TetMesh mesh(nv, &v[0], ne, &t[0], 1e7, 0.46, 1000.0);
  SparseMatrix* M = NULL;
  GenerateMassMatrix::computeMassMatrix(&mesh, &M, true);
  std::vector<double> u(r), f(r), fr(r);
  for (int i = 0; i < r; i++) u[i] = 0.004 * sin(0.37 * i);

  // 1. force model through the reference's abstract interface vs the reference's own element code
  for (int warp = 1; warp <= 2; warp++) {
    PS::FEM::HipCorotationalForceModel hip(&mesh, 1e7, 0.46, 1000.0, warp);
    if (!hip.ok()) { printf("ERROR=no device\n"); return 1; }
    ForceModel* fm = &hip;  // what host code holds
    SparseMatrix *K = NULL, *Kr = NULL;
    fm->GetTangentStiffnessMatrixTopology(&K);
    fm->GetForceAndMatrix(&u[0], &f[0], K);
    CorotationalLinearFEM ref(&mesh);
    ref.GetStiffnessMatrixTopology(&Kr);
    ref.ComputeForceAndStiffnessMatrix(&u[0], &fr[0], Kr, warp);
    double df = 0, nf = 0, dk = 0, nk = 0;
    int same_pattern = K->GetNumRows() == Kr->GetNumRows();
    for (int i = 0; i < r; i++) { df = fmax(df, fabs(f[i] - fr[i])); nf = fmax(nf, fabs(fr[i])); }
    for (int i = 0; i < r && same_pattern; i++) {
      same_pattern = K->GetRowLength(i) == Kr->GetRowLength(i);
      for (int j = 0; j < Kr->GetRowLength(i) && same_pattern; j++) {
        same_pattern = K->GetColumnIndex(i, j) == Kr->GetColumnIndex(i, j);
        dk = fmax(dk, fabs(K->GetEntry(i, j) - Kr->GetEntry(i, j)));
        nk = fmax(nk, fabs(Kr->GetEntry(i, j)));
      }
    }
    // f alone comes from the stepping handle (fp32-stored matrix): it must be the same f
    std::vector<double> f2(r);
    fm->GetInternalForce(&u[0], &f2[0]);
    double df2 = 0;
    for (int i = 0; i < r; i++) df2 = fmax(df2, fabs(f2[i] - fr[i]));
    printf("WARP%d_PATTERN=%d\nWARP%d_F_RELDIFF=%.3e\nWARP%d_K_RELDIFF=%.3e\nWARP%d_F_STEPPING_HANDLE_RELDIFF=%.3e\n", warp, same_pattern, warp, df / nf, warp,
           dk / nk, warp, df2 / nf);
    delete K; delete Kr;
    if (warp == 2) {
      // 4a. the reference's finite-difference check of K against f (prints its own "eps=...: maxEntry=..." lines): with the exact
      // tangent, f(q + eps dq) - f(q) - K eps dq is O(eps^2)
      std::vector<double> dq(r);
      for (int i = 0; i < r; i++) dq[i] = 0.01 * cos(0.61 * i);
      printf("TESTSTIFFNESS_BEGIN=1\n");
      fm->TestStiffnessMatrix(&u[0], &dq[0]);
      printf("TESTSTIFFNESS_END=1\n");
    }
  }

  // 2 + 3. integrator derived from the reference's IntegratorBaseSparse; the reference's CGSolver on the device matrix
  std::vector<int> fixed;
  for (int j = 0; j < n * n; j++)
    for (int k = 0; k < 3; k++) fixed.push_back(3 * j + k);  // plane i = 0
  PS::FEM::HipCorotationalForceModel hip(&mesh, 1e7, 0.46, 1000.0);
  PS::FEM::HipVolumeConservingIntegrator hipInt(r, 0.0333, M, &hip, 0, (int)fixed.size(), &fixed[0], 0.0, 0.01);
  IntegratorBaseSparse* integ = &hipInt;
  std::vector<double> fext(r, 0.0);
  for (int i = 0; i < nv; i++) fext[3 * i + 1] = -10000.0;
  for (int step = 0; step < 3; step++) {
    integ->SetExternalForces(&fext[0]);
    const int rc = integ->DoTimestep();
    double s = 0;
    for (int i = 0; i < r; i++) s += integ->Getq()[i] * integ->Getq()[i];
    printf("STEP%d_RC=%d\nSTEP%d_QNORM=%.10e\nSTEP%d_ITERS=%d\n", step, rc, step, sqrt(s), step, hipInt.lastIterations());
  }
  printf("TOTAL_MASS=%.10e\nKINETIC=%.10e\n", integ->GetTotalMass(), integ->GetKineticEnergy());
  // the system the next step would solve, through the reference's CG loop (no preconditioner: the black-box constructor has no
  // matrix to take a diagonal from, CGSolver.cpp:44-47) -- the constrained rows are identity rows, so x stays 0 there
  std::vector<double> Kb(9 * (size_t)fb_fem_num_blocks(hip.handle())), rhs(r), x(r, 0.0), Ax(r);
  fb_fem_set_state(hip.handle(), integ->Getq(), integ->Getqvel(), NULL);
  fb_fem_system(hip.handle(), &Kb[0], &rhs[0]);
  CGSolver cg(r, PS::FEM::hipBlackBoxProduct, (void*)hip.handle());
  const int info = cg.SolveLinearSystemWithoutPreconditioner(&x[0], &rhs[0], 1e-8, 20000, 0);
  PS::FEM::hipBlackBoxProduct((void*)hip.handle(), &x[0], &Ax[0]);
  double res = 0, nb = 0;
  for (int i = 0; i < r; i++) { res += (Ax[i] - rhs[i]) * (Ax[i] - rhs[i]); nb += rhs[i] * rhs[i]; }
  printf("REFCG_INFO=%d\nREFCG_RESIDUAL=%.3e\n", info, sqrt(res / nb));
  // 4b. the device's own Jacobi-PCG solution of that system, checked by the reference's SparseMatrix::CheckLinearSystemSolution in
  // a reference SparseMatrix filled with the device's Keff
  {
    SparseMatrix* Keff = NULL;
    hip.GetTangentStiffnessMatrixTopology(&Keff);
    std::vector<int> bptr(nv + 1), bcol(fb_fem_num_blocks(hip.handle()));
    fb_fem_pattern(hip.handle(), &bptr[0], &bcol[0]);
    for (int a = 0; a < nv; a++)
      for (int p = bptr[a]; p < bptr[a + 1]; p++)
        for (int k = 0; k < 3; k++)
          for (int l = 0; l < 3; l++) Keff->SetEntry(3 * a + k, 3 * (p - bptr[a]) + l, Kb[9 * (size_t)p + 3 * k + l]);
    std::vector<double> xd(r, 0.0);
    int its = 0;
    fb_fem_pcg(hip.handle(), &rhs[0], &xd[0], 1e-6, 10000, &its);
    const double rel = Keff->CheckLinearSystemSolution(&xd[0], &rhs[0], 1);
    printf("DEVICE_PCG_ITERS=%d\nREF_CHECKLINEARSYSTEM_RELINF=%.3e\n", its, rel);
    delete Keff;
  }
  printf("MATRIX_HANDLE_CREATED=%d\n", hip.matrixHandle() != NULL);
  delete M;
  return 0;
}
