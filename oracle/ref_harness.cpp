// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product path.
//
// C-ABI harness around the *reference's own* translation units (VegaFEM, vendored in
// /root/reference/src/3rdparty/vegafem), compiled where they lie by oracle/Makefile into
// oracle/_ref/libfem_ref.so.  This file is ours; it only *calls* reference classes:
//   TetMesh, CorotationalLinearFEM, GenerateMassMatrix, SparseMatrix, CGSolver, RemoveRows/InsertRows.
// The integrator TUs (vegafem/integrator/*.cpp, src/deformable/PS_VolumeConservingIntegrator.cpp)
// are NOT built: vegafem/integrator/integratorSolverSelection.h:38 hard-selects PARDISO (Intel MKL,
// absent here) and building them would need an edited header copy, which the rules forbid.  Their
// step sequence (PS_VolumeConservingIntegrator.cpp:46-260, implicitNewmarkSparse.cpp:39-83) is
// restated below *on the reference's own SparseMatrix/CGSolver objects*, operation by operation, and
// pinned by the norms SURVEY.md section 8c recorded from the full reference build (27^3 cube,
// |q|_2 = 730.25, 875.31, 531.02 after steps 1..3) -- see tests/test_oracle_ref.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tetMesh.h"
#include "corotationalLinearFEM.h"
#include "generateMassMatrix.h"
#include "sparseMatrix.h"
#include "CGSolver.h"
#include "insertRows.h"
#include "polarDecomposition.h"

namespace {

// exposes the protected per-element caches of the reference class (no behaviour added)
class CorotPeek : public CorotationalLinearFEM {
public:
  explicit CorotPeek(TetMesh* m) : CorotationalLinearFEM(m) {}
  const double* K0(int el) const { return KElementUndeformed[el]; }
  const double* Minv(int el) const { return MInverse[el]; }
  const int* rowIdx(int el) const { return rowIndices[el]; }
  const int* colIdx(int el) const { return columnIndices[el]; }
};

struct RefFem {
  TetMesh* mesh;
  CorotPeek* fem;
  SparseMatrix* mass;
  int r;
  // integrator state (restated ImplicitNewmarkSparse ctor)
  SparseMatrix* K;      // tangentStiffnessMatrix
  SparseMatrix* D;      // rayleighDampingMatrix
  SparseMatrix* damp;   // empty dampingMatrix
  SparseMatrix* sys;    // systemMatrix
  CGSolver* cg;
  std::vector<int> fixed;
  std::vector<double> q, qvel, qaccel, fext, fint, qres, qdelta, buf, bufc;
  double h, cM, cK, scale;
  int last_iters;
  int warp;  // the `warp` argument of ComputeForceAndStiffnessMatrix: 1 (FemBrain's default) or 0 (linear)
  RefFem() : mesh(NULL), fem(NULL), mass(NULL), r(0), K(NULL), D(NULL), damp(NULL), sys(NULL), cg(NULL),
             h(0.0333), cM(0.0), cK(0.01), scale(1.0), last_iters(0), warp(1) {}
};

}  // namespace

extern "C" {

void* ref_fem_create(int nv, const double* verts, int nt, const int* tets, double E, double nu, double rho) {
  RefFem* s = new RefFem;
  s->mesh = new TetMesh(nv, const_cast<double*>(verts), nt, const_cast<int*>(tets), E, nu, rho);
  s->fem = new CorotPeek(s->mesh);
  GenerateMassMatrix::computeMassMatrix(s->mesh, &s->mass, true);
  s->r = 3 * nv;
  return s;
}

void ref_fem_destroy(void* h) {
  RefFem* s = (RefFem*)h;
  delete s->cg; delete s->sys; delete s->damp; delete s->D; delete s->K;
  delete s->mass; delete s->fem; delete s->mesh;
  delete s;
}

void ref_fem_K0(void* h, int el, double* out144) { memcpy(out144, ((RefFem*)h)->fem->K0(el), 144 * sizeof(double)); }
void ref_fem_Minv(void* h, int el, double* out16) { memcpy(out16, ((RefFem*)h)->fem->Minv(el), 16 * sizeof(double)); }
void ref_fem_elem_indices(void* h, int el, int* row4, int* col16) {
  memcpy(row4, ((RefFem*)h)->fem->rowIdx(el), 4 * sizeof(int));
  memcpy(col16, ((RefFem*)h)->fem->colIdx(el), 16 * sizeof(int));
}

// CSR of the stiffness topology: returns nnz; ia[r+1], ja[nnz] filled when non-null
int ref_fem_topology(void* h, int* ia, int* ja) {
  RefFem* s = (RefFem*)h;
  SparseMatrix* T; s->fem->GetStiffnessMatrixTopology(&T);
  int nnz = T->GetNumEntries();
  if (ia && ja) T->GenerateCompressedRowMajorFormat(NULL, ia, ja, 0, 0);
  delete T;
  return nnz;
}

// mass matrix CSR
int ref_fem_mass(void* h, int* ia, int* ja, double* a) {
  RefFem* s = (RefFem*)h;
  int nnz = s->mass->GetNumEntries();
  if (ia && ja && a) s->mass->GenerateCompressedRowMajorFormat(a, ia, ja, 0, 0);
  return nnz;
}

void ref_fem_set_linear(void* h, int linear) { ((RefFem*)h)->warp = linear ? 0 : 1; }
// the `warp` argument of ComputeForceAndStiffnessMatrix: 0 linear, 1 corotational (FemBrain), 2 corotational with the exact
// tangent stiffness (corotationalLinearFEM.cpp:296-428)
void ref_fem_set_warp(void* h, int warp) { ((RefFem*)h)->warp = warp; }

// f_int and K values (CSR order of ref_fem_topology) for displacement u (warp = 1 unless ref_fem_set_linear)
void ref_fem_assemble(void* h, const double* u, double* f, double* Kvals) {
  RefFem* s = (RefFem*)h;
  SparseMatrix* T; s->fem->GetStiffnessMatrixTopology(&T);
  s->fem->ComputeForceAndStiffnessMatrix(const_cast<double*>(u), f, T, s->warp);
  if (Kvals) T->GenerateCompressedRowMajorFormat(Kvals, NULL, NULL, 0, 0);
  delete T;
}

double ref_polar(const double* F, double* R, double* S, double tol) {
  return PolarDecomposition::Compute(F, R, S, tol);
}

// --- restated ImplicitNewmarkSparse ctor (implicitNewmarkSparse.cpp:39-83) on reference objects ---
void ref_integrator_create(void* hh, int nfixed, const int* fixedDOFs, double timestep, double cM, double cK) {
  RefFem* s = (RefFem*)hh;
  int r = s->r;
  s->h = timestep; s->cM = cM; s->cK = cK;
  s->fixed.assign(fixedDOFs, fixedDOFs + nfixed);
  s->fem->GetStiffnessMatrixTopology(&s->K);
  SparseMatrixOutline emptyOutline(r);
  s->damp = new SparseMatrix(&emptyOutline);
  s->D = new SparseMatrix(*s->K);
  s->D->BuildSubMatrixIndices(*s->mass);
  s->K->BuildSubMatrixIndices(*s->mass);
  s->K->BuildSubMatrixIndices(*s->damp, 1);
  s->sys = new SparseMatrix(*s->K);
  s->sys->RemoveRowsColumns(nfixed, s->fixed.data());
  s->sys->BuildSuperMatrixIndices(nfixed, s->fixed.data(), s->K);
  s->cg = new CGSolver(s->sys);
  s->q.assign(r, 0); s->qvel.assign(r, 0); s->qaccel.assign(r, 0);
  s->fext.assign(r, 0); s->fint.assign(r, 0); s->qres.assign(r, 0);
  s->qdelta.assign(r, 0); s->buf.assign(r, 0); s->bufc.assign(r - nfixed, 0);
}

void ref_set_state(void* hh, const double* q, const double* qvel) {
  RefFem* s = (RefFem*)hh;
  memcpy(s->q.data(), q, sizeof(double) * s->r);
  if (qvel) memcpy(s->qvel.data(), qvel, sizeof(double) * s->r);
}
void ref_get_state(void* hh, double* q, double* qvel) {
  RefFem* s = (RefFem*)hh;
  if (q) memcpy(q, s->q.data(), sizeof(double) * s->r);
  if (qvel) memcpy(qvel, s->qvel.data(), sizeof(double) * s->r);
}
void ref_set_external_forces(void* hh, const double* f) {
  RefFem* s = (RefFem*)hh;
  memcpy(s->fext.data(), f, sizeof(double) * s->r);
}

// --- restated VolumeConservingIntegrator::DoTimestep (PS_VolumeConservingIntegrator.cpp:46-260),
//     dynamic branch, maxIterations = 1.  Returns CG info (>0 iterations converged, <0 not). ---
// Optional outputs (may be NULL): keff = Keff values in CSR order of the full pattern,
// rhs = right-hand side (full length r), dv = solution inserted back to full length.
int ref_step(void* hh, double cg_eps, int cg_maxiter, double* keff, double* rhs, double* dv) {
  RefFem* s = (RefFem*)hh;
  int r = s->r;
  for (int i = 0; i < r; i++) s->qaccel[i] = 0;
  s->fem->ComputeForceAndStiffnessMatrix(s->q.data(), s->fint.data(), s->K, s->warp);
  for (int i = 0; i < r; i++) s->fint[i] *= s->scale;
  *s->K *= s->scale;
  memset(s->qres.data(), 0, sizeof(double) * r);
  s->K->ScalarMultiply(s->cK, s->D);
  s->D->AddSubMatrix(s->cM, *s->mass);
  *s->K *= s->h;
  *s->K += *s->D;
  s->K->AddSubMatrix(1.0, *s->damp, 1);
  s->K->MultiplyVector(s->qvel.data(), s->qres.data());
  *s->K *= s->h;
  s->K->AddSubMatrix(1.0, *s->mass);
  for (int i = 0; i < r; i++) {
    s->qres[i] += s->fint[i] - s->fext[i];
    s->qres[i] *= -s->h;
    s->qdelta[i] = s->qres[i];
  }
  if (keff) s->K->GenerateCompressedRowMajorFormat(keff, NULL, NULL, 0, 0);
  if (rhs) memcpy(rhs, s->qres.data(), sizeof(double) * r);
  int nf = (int)s->fixed.size();
  RemoveRows(r, s->bufc.data(), s->qdelta.data(), nf, s->fixed.data());
  s->sys->AssignSuperMatrix(s->K);
  memset(s->buf.data(), 0, sizeof(double) * r);
  int info = s->cg->SolveLinearSystemWithJacobiPreconditioner(s->buf.data(), s->bufc.data(), cg_eps, cg_maxiter);
  s->last_iters = info;
  InsertRows(r, s->buf.data(), s->qdelta.data(), nf, s->fixed.data());
  if (dv) memcpy(dv, s->qdelta.data(), sizeof(double) * r);
  for (int i = 0; i < r; i++) {
    s->qvel[i] += s->qdelta[i];
    s->q[i] += s->h * s->qvel[i];
  }
  for (int i = 0; i < nf; i++) s->q[s->fixed[i]] = s->qvel[s->fixed[i]] = s->qaccel[s->fixed[i]] = 0.0;
  return info;
}

// --- restated ImplicitNewmarkSparse::DoTimestep (implicitNewmarkSparse.cpp:183-379; UpdateAlphas :96-104), dynamic branch, PCG
//     solver, on the reference's own SparseMatrix / CGSolver objects.  As there: `buf` (the solver's start vector) is NOT
//     cleared between solves, so every solve starts from the previous solution.  Returns the number of Newton iterations
//     performed, or -1 when a solve failed; cg_total receives the sum of the PCG iteration counts. ---
void ref_get_accel(void* hh, double* qaccel) { RefFem* s = (RefFem*)hh; memcpy(qaccel, s->qaccel.data(), sizeof(double) * s->r); }
void ref_set_accel(void* hh, const double* qaccel) { RefFem* s = (RefFem*)hh; memcpy(s->qaccel.data(), qaccel, sizeof(double) * s->r); }

int ref_newmark_step(void* hh, double beta, double gamma, int max_newton, double epsilon, double cg_eps, int cg_maxiter, int* cg_total) {
  RefFem* s = (RefFem*)hh;
  const int r = s->r, nf = (int)s->fixed.size();
  const double h = s->h;
  const double alpha1 = 1.0 / (beta * h * h), alpha2 = 1.0 / (beta * h), alpha3 = (1.0 - 2.0 * beta) / (2.0 * beta);
  const double alpha4 = gamma / (beta * h), alpha5 = 1 - gamma / beta, alpha6 = (1.0 - gamma / (2.0 * beta)) * h;
  std::vector<double> q_1(s->q), qvel_1(s->qvel), qaccel_1(s->qaccel);
  for (int i = 0; i < r; i++) {
    s->qaccel[i] = alpha1 * (s->q[i] - q_1[i]) - alpha2 * qvel_1[i] - alpha3 * qaccel_1[i];
    s->qvel[i] = alpha4 * (s->q[i] - q_1[i]) + alpha5 * qvel_1[i] + alpha6 * qaccel_1[i];
  }
  int numIter = 0, total = 0;
  double error0 = 0, errorQuotient;
  do {
    s->fem->ComputeForceAndStiffnessMatrix(s->q.data(), s->fint.data(), s->K, s->warp);
    for (int i = 0; i < r; i++) s->fint[i] *= s->scale;
    *s->K *= s->scale;
    memset(s->qres.data(), 0, sizeof(double) * r);
    s->K->ScalarMultiply(s->cK, s->D);
    s->D->AddSubMatrix(s->cM, *s->mass);
    s->D->ScalarMultiplyAdd(alpha4, s->K);
    s->K->AddSubMatrix(alpha4, *s->damp, 1);
    s->K->AddSubMatrix(alpha1, *s->mass);
    s->mass->MultiplyVector(s->qaccel.data(), s->qres.data());
    s->D->MultiplyVectorAdd(s->qvel.data(), s->qres.data());
    s->damp->MultiplyVectorAdd(s->qvel.data(), s->qres.data());
    for (int i = 0; i < r; i++) {
      s->qres[i] += s->fint[i] - s->fext[i];
      s->qres[i] *= -1;
      s->qdelta[i] = s->qres[i];
    }
    double error = 0;
    for (int i = 0; i < r; i++) error += s->qres[i] * s->qres[i];
    if (numIter == 0) { error0 = error; errorQuotient = 1.0; } else errorQuotient = error / error0;
    if (errorQuotient < epsilon * epsilon) break;
    RemoveRows(r, s->bufc.data(), s->qdelta.data(), nf, s->fixed.data());
    s->sys->AssignSuperMatrix(s->K);
    int info = s->cg->SolveLinearSystemWithJacobiPreconditioner(s->buf.data(), s->bufc.data(), cg_eps, cg_maxiter);
    s->last_iters = info;
    if (info < 0) { if (cg_total) *cg_total = total - info; return -1; }
    total += info;
    InsertRows(r, s->buf.data(), s->qdelta.data(), nf, s->fixed.data());
    for (int i = 0; i < r; i++) {
      s->q[i] += s->qdelta[i];
      s->qaccel[i] = alpha1 * (s->q[i] - q_1[i]) - alpha2 * qvel_1[i] - alpha3 * qaccel_1[i];
      s->qvel[i] = alpha4 * (s->q[i] - q_1[i]) + alpha5 * qvel_1[i] + alpha6 * qaccel_1[i];
    }
    for (int i = 0; i < nf; i++) s->q[s->fixed[i]] = s->qvel[s->fixed[i]] = s->qaccel[s->fixed[i]] = 0.0;
    numIter++;
  } while (numIter < max_newton);
  if (cg_total) *cg_total = total;
  return numIter;
}

// system matrix (constrained) CSR export after a step, for SpMV/CG goldens
int ref_sys_csr(void* hh, int* ia, int* ja, double* a) {
  RefFem* s = (RefFem*)hh;
  int nnz = s->sys->GetNumEntries();
  if (ia && ja && a) s->sys->GenerateCompressedRowMajorFormat(a, ia, ja, 0, 0);
  return nnz;
}
int ref_sys_rows(void* hh) { return ((RefFem*)hh)->sys->GetNumRows(); }

// stand-alone reference PCG on a caller CSR (row-array SparseMatrix built through the outline)
int ref_pcg(int n, const int* ia, const int* ja, const double* a, const double* b, double* x, double eps, int maxit) {
  SparseMatrixOutline o(n);
  for (int i = 0; i < n; i++)
    for (int k = ia[i]; k < ia[i + 1]; k++) o.AddEntry(i, ja[k], a[k]);
  SparseMatrix A(&o);
  CGSolver cg(&A);
  return cg.SolveLinearSystemWithJacobiPreconditioner(x, b, eps, maxit);
}

}  // extern "C"
