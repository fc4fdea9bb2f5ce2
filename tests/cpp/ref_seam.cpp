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
// Prints KEY=value lines.  Needs a GPU to RUN; compiling and linking it is the CPU-side proof.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

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

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
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
    printf("WARP%d_PATTERN=%d\nWARP%d_F_RELDIFF=%.3e\nWARP%d_K_RELDIFF=%.3e\n", warp, same_pattern, warp, df / nf, warp, dk / nk);
    delete K; delete Kr;
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
  delete M;
  return 0;
}
