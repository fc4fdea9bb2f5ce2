/* TEST INFRASTRUCTURE ONLY -- a CPU restatement (plain C, fp64, flat CSR) of the reference's per-step
 * deformable FEM path.  It is the parity checker for the HIP path and the "port" CPU baseline of
 * bench.py; nothing under fembrain_amd/ may import, link or call it.
 *
 * Pinned against the reference's own translation units (oracle/_ref/libfem_ref.so, built from
 * /root/reference by oracle/Makefile) in tests/test_oracle_ref.py, and against the committed golden
 * vectors under tests/golden/ (generated from that build by tests/golden/make_fem_golden.py).
 *
 * What each function follows (reference file:line, paths under /root/reference/src):
 *   orc_fem_create      3rdparty/vegafem/corotationalLinearFEM/corotationalLinearFEM.cpp:40-146 (M^-1, K0),
 *                       :163-186 + sparseMatrix/sparseMatrix.cpp:238-262 (pattern, ascending columns),
 *                       :482-502 (element -> matrix position cache),
 *                       volumetricMesh/generateMassMatrix.cpp:33-76 + tetMesh.cpp:150-188 (mass)
 *   orc_polar           polarDecomposition/polarDecomposition.cpp:37-108
 *   orc_fem_assemble    corotationalLinearFEM.cpp:219-470 (warp = 1) and :191-211 (WarpMatrix)
 *   orc_integrator_*    integrator/implicitNewmarkSparse.cpp:39-83, sparseMatrix.cpp:1296-1358, :940-1002,
 *                       insertRows/insertRows.cpp:29-109
 *   orc_step            deformable/PS_VolumeConservingIntegrator.cpp:46-260 (dynamic branch, maxIterations 1)
 *   orc_pcg             sparseSolver/CGSolver.cpp:129-208 (Jacobi PCG, exact residual every 30th iteration)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int nv, nt, r;
  double *x0;   /* 3 nv rest positions */
  int *tets;    /* 4 nt */
  double *Minv; /* 16 nt, row-major, rows = vertices: [b_k | c_k] */
  double *K0;   /* 144 nt, row-major 12x12 */
  double lambda, mu, rho;
  int linear;   /* warp = 0: K = K0, f = K0 u (corotationalLinearFEM.cpp:429-453) */
  int exact;    /* warp = 2: the exact tangent terms are added to K (corotationalLinearFEM.cpp:296-428) */
  double *qaccel; /* Newmark state (orc_newmark_step) */
  /* node-level block pattern */
  int *bptr, *bcol, nblk;
  /* scalar CSR */
  int nnz, *ia, *ja;
  int *elblk;    /* 16 nt: block id of (vertex i, vertex j) of each tet */
  double *mblk;  /* nblk: consistent-mass scalar of each block (sits on the block's 3 diagonal entries) */
  /* integrator */
  int nfixed, *fixed, rc, nnzc;
  int *cia, *cja, *csrc; /* constrained system CSR + source position in the full CSR */
  int *c2f;              /* constrained row -> full row */
  double h, cM, cK, scale;
  double *K, *D, *sys;
  double *q, *qvel, *fext, *fint, *qres, *qdelta, *buf, *bufc;
  double *cg_r, *cg_d, *cg_q, *cg_inv;
  double t_assembly, t_solve;
} OrcFem;

static int cmp_pair(const void *a, const void *b) {
  const long long x = *(const long long *)a, y = *(const long long *)b;
  return (x > y) - (x < y);
}

/* 3x3 inverse by cofactors (row-major) */
static void inv3(const double *A, double *I) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02, id = 1.0 / det;
  I[0] = c00 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  I[3] = c01 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  I[6] = c02 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* M = [x0 x1 x2 x3; 1 1 1 1]; Minv rows k = [grad N_k | N_k(0)].  Same matrix the reference gets from its
 * general 4x4 cofactor inverse (corotationalLinearFEM.cpp:70-90,529-572), computed through the edge matrix. */
static void tet_minv(const double *p0, const double *p1, const double *p2, const double *p3, double *Mi) {
  double Dm[9], Di[9];
  for (int d = 0; d < 3; d++) { Dm[3 * d + 0] = p1[d] - p0[d]; Dm[3 * d + 1] = p2[d] - p0[d]; Dm[3 * d + 2] = p3[d] - p0[d]; }
  inv3(Dm, Di); /* rows of Di are grad N_1..N_3 */
  for (int d = 0; d < 3; d++) {
    Mi[4 * 1 + d] = Di[0 + d]; Mi[4 * 2 + d] = Di[3 + d]; Mi[4 * 3 + d] = Di[6 + d];
    Mi[d] = -(Di[0 + d] + Di[3 + d] + Di[6 + d]);
  }
  for (int k = 0; k < 4; k++)
    Mi[4 * k + 3] = (k == 0 ? 1.0 : 0.0) - (Mi[4 * k] * p0[0] + Mi[4 * k + 1] * p0[1] + Mi[4 * k + 2] * p0[2]);
}

static double tet_volume(const double *a, const double *b, const double *c, const double *d) {
  double u[3] = {a[0] - d[0], a[1] - d[1], a[2] - d[2]}, v[3] = {b[0] - d[0], b[1] - d[1], b[2] - d[2]},
         w[3] = {c[0] - d[0], c[1] - d[1], c[2] - d[2]};
  double cx = v[1] * w[2] - v[2] * w[1], cy = v[2] * w[0] - v[0] * w[2], cz = v[0] * w[1] - v[1] * w[0];
  return 1.0 / 6 * fabs(u[0] * cx + u[1] * cy + u[2] * cz);
}

/* K0 = V * B^T E B with B (6x12) built from the shape-function gradients (rows of Minv) */
static void tet_K0(const double *Mi, double lambda, double mu, double vol, double *K) {
  double B[72], E[36], EB[72];
  memset(B, 0, sizeof B); memset(E, 0, sizeof E); memset(EB, 0, sizeof EB);
  for (int v = 0; v < 4; v++) {
    double bx = Mi[4 * v], by = Mi[4 * v + 1], bz = Mi[4 * v + 2];
    B[0 * 12 + 3 * v + 0] = bx; B[1 * 12 + 3 * v + 1] = by; B[2 * 12 + 3 * v + 2] = bz;
    B[3 * 12 + 3 * v + 0] = by; B[3 * 12 + 3 * v + 1] = bx;
    B[4 * 12 + 3 * v + 1] = bz; B[4 * 12 + 3 * v + 2] = by;
    B[5 * 12 + 3 * v + 0] = bz; B[5 * 12 + 3 * v + 2] = bx;
  }
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) E[6 * i + j] = lambda; E[6 * i + i] = lambda + 2 * mu; E[6 * (i + 3) + i + 3] = mu; }
  for (int i = 0; i < 6; i++) for (int j = 0; j < 12; j++) for (int k = 0; k < 6; k++) EB[12 * i + j] += E[6 * i + k] * B[12 * k + j];
  for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) {
    double s = 0; for (int k = 0; k < 6; k++) s += B[12 * k + i] * EB[12 * k + j];
    K[12 * i + j] = s * vol;
  }
}

static int find_blk(const OrcFem *s, int a, int b) {
  int lo = s->bptr[a], hi = s->bptr[a + 1] - 1;
  while (lo <= hi) { int m = (lo + hi) >> 1; if (s->bcol[m] == b) return m; if (s->bcol[m] < b) lo = m + 1; else hi = m - 1; }
  return -1;
}

void *orc_fem_create(int nv, const double *verts, int nt, const int *tets, double E, double nu, double rho) {
  OrcFem *s = (OrcFem *)calloc(1, sizeof(OrcFem));
  s->nv = nv; s->nt = nt; s->r = 3 * nv; s->rho = rho;
  s->lambda = (nu * E) / ((1 + nu) * (1 - 2 * nu)); s->mu = E / (2 * (1 + nu));
  s->x0 = (double *)malloc(sizeof(double) * 3 * nv); memcpy(s->x0, verts, sizeof(double) * 3 * nv);
  s->tets = (int *)malloc(sizeof(int) * 4 * nt); memcpy(s->tets, tets, sizeof(int) * 4 * nt);
  s->Minv = (double *)malloc(sizeof(double) * 16 * (size_t)nt);
  s->K0 = (double *)malloc(sizeof(double) * 144 * (size_t)nt);
  double *vol = (double *)malloc(sizeof(double) * nt);
  for (int e = 0; e < nt; e++) {
    const int *t = tets + 4 * e;
    tet_minv(verts + 3 * t[0], verts + 3 * t[1], verts + 3 * t[2], verts + 3 * t[3], s->Minv + 16 * (size_t)e);
    vol[e] = tet_volume(verts + 3 * t[0], verts + 3 * t[1], verts + 3 * t[2], verts + 3 * t[3]);
    tet_K0(s->Minv + 16 * (size_t)e, s->lambda, s->mu, vol[e], s->K0 + 144 * (size_t)e);
  }
  /* block pattern: all vertex pairs of every tet, ascending columns per row */
  size_t np = 16 * (size_t)nt;
  long long *pairs = (long long *)malloc(sizeof(long long) * np);
  for (int e = 0; e < nt; e++) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
    pairs[16 * (size_t)e + 4 * i + j] = ((long long)tets[4 * e + i] << 32) | (unsigned)tets[4 * e + j];
  qsort(pairs, np, sizeof(long long), cmp_pair);
  size_t nu_ = 0; for (size_t k = 0; k < np; k++) if (k == 0 || pairs[k] != pairs[k - 1]) pairs[nu_++] = pairs[k];
  s->nblk = (int)nu_;
  s->bptr = (int *)calloc(nv + 1, sizeof(int)); s->bcol = (int *)malloc(sizeof(int) * nu_);
  for (size_t k = 0; k < nu_; k++) { s->bptr[(pairs[k] >> 32) + 1]++; s->bcol[k] = (int)(pairs[k] & 0xffffffff); }
  for (int a = 0; a < nv; a++) s->bptr[a + 1] += s->bptr[a];
  free(pairs);
  /* scalar CSR */
  s->nnz = 9 * s->nblk; s->ia = (int *)malloc(sizeof(int) * (s->r + 1)); s->ja = (int *)malloc(sizeof(int) * (size_t)s->nnz);
  int cnt = 0;
  for (int a = 0; a < nv; a++) for (int k = 0; k < 3; k++) {
    s->ia[3 * a + k] = cnt;
    for (int p = s->bptr[a]; p < s->bptr[a + 1]; p++) for (int l = 0; l < 3; l++) s->ja[cnt++] = 3 * s->bcol[p] + l;
  }
  s->ia[s->r] = cnt;
  /* element -> block cache and consistent mass (rho V / 20 (1 + delta_ij)) */
  s->elblk = (int *)malloc(sizeof(int) * 16 * (size_t)nt); s->mblk = (double *)calloc(s->nblk, sizeof(double));
  for (int e = 0; e < nt; e++) {
    double factor = rho * vol[e] / 20;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
      int b = find_blk(s, tets[4 * e + i], tets[4 * e + j]);
      s->elblk[16 * (size_t)e + 4 * i + j] = b;
      s->mblk[b] += factor * (i == j ? 2 : 1);
    }
  }
  free(vol);
  return s;
}

void orc_fem_destroy(void *h) {
  OrcFem *s = (OrcFem *)h;
  free(s->x0); free(s->tets); free(s->Minv); free(s->K0); free(s->bptr); free(s->bcol); free(s->ia); free(s->ja);
  free(s->elblk); free(s->mblk); free(s->fixed); free(s->cia); free(s->cja); free(s->csrc); free(s->c2f);
  free(s->K); free(s->D); free(s->sys); free(s->q); free(s->qvel); free(s->fext); free(s->fint); free(s->qres);
  free(s->qdelta); free(s->qaccel); free(s->buf); free(s->bufc); free(s->cg_r); free(s->cg_d); free(s->cg_q); free(s->cg_inv);
  free(s);
}

int orc_fem_nnz(void *h) { return ((OrcFem *)h)->nnz; }
int orc_fem_nblk(void *h) { return ((OrcFem *)h)->nblk; }
void orc_fem_csr(void *h, int *ia, int *ja) { OrcFem *s = (OrcFem *)h; memcpy(ia, s->ia, sizeof(int) * (s->r + 1)); memcpy(ja, s->ja, sizeof(int) * (size_t)s->nnz); }
void orc_fem_blocks(void *h, int *bptr, int *bcol) { OrcFem *s = (OrcFem *)h; memcpy(bptr, s->bptr, sizeof(int) * (s->nv + 1)); memcpy(bcol, s->bcol, sizeof(int) * (size_t)s->nblk); }
void orc_fem_K0(void *h, int el, double *out) { memcpy(out, ((OrcFem *)h)->K0 + 144 * (size_t)el, 144 * sizeof(double)); }
void orc_fem_Minv(void *h, int el, double *out) { memcpy(out, ((OrcFem *)h)->Minv + 16 * (size_t)el, 16 * sizeof(double)); }
void orc_fem_elblk(void *h, int el, int *out16) { memcpy(out16, ((OrcFem *)h)->elblk + 16 * (size_t)el, 16 * sizeof(int)); }
/* mass matrix as values on the full stiffness CSR pattern (zero off the block diagonals) */
void orc_fem_mass_on_pattern(void *h, double *a) {
  OrcFem *s = (OrcFem *)h; memset(a, 0, sizeof(double) * (size_t)s->nnz);
  for (int v = 0; v < s->nv; v++) for (int k = 0; k < 3; k++) for (int p = s->bptr[v]; p < s->bptr[v + 1]; p++)
    a[s->ia[3 * v + k] + 3 * (p - s->bptr[v]) + k] = s->mblk[p];
}

static double one_norm(const double *A) { double n = 0; for (int i = 0; i < 3; i++) { double c = fabs(A[i]) + fabs(A[i + 3]) + fabs(A[i + 6]); if (c > n) n = c; } return n; }
static double inf_norm(const double *A) { double n = 0; for (int i = 0; i < 3; i++) { double c = fabs(A[3 * i]) + fabs(A[3 * i + 1]) + fabs(A[3 * i + 2]); if (c > n) n = c; } return n; }
static void cross3(const double *a, const double *b, double *c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }

/* scaled Newton polar decomposition F = Q S; returns the last determinant */
double orc_polar(const double *F, double *Q, double *S, double tol) {
  double Mk[9], Ek[9], det, M1, Mi, E1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mk[3 * i + j] = F[3 * j + i];
  M1 = one_norm(Mk); Mi = inf_norm(Mk);
  do {
    double A[9];
    cross3(Mk + 3, Mk + 6, A); cross3(Mk + 6, Mk, A + 3); cross3(Mk, Mk + 3, A + 6);
    det = Mk[0] * A[0] + Mk[1] * A[1] + Mk[2] * A[2];
    if (det == 0.0) break;
    double A1 = one_norm(A), Ai = inf_norm(A);
    double gamma = sqrt(sqrt((A1 * Ai) / (M1 * Mi)) / fabs(det));
    double g1 = gamma * 0.5, g2 = 0.5 / (gamma * det);
    for (int i = 0; i < 9; i++) { Ek[i] = Mk[i]; Mk[i] = g1 * Mk[i] + g2 * A[i]; Ek[i] -= Mk[i]; }
    E1 = one_norm(Ek); M1 = one_norm(Mk); Mi = inf_norm(Mk);
  } while (E1 > M1 * tol);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[3 * i + j] = Mk[3 * j + i];
  if (S) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { S[3 * i + j] = 0; for (int k = 0; k < 3; k++) S[3 * i + j] += Mk[3 * i + k] * F[3 * k + j]; }
    for (int i = 0; i < 3; i++) for (int j = i; j < 3; j++) S[3 * i + j] = S[3 * j + i] = 0.5 * (S[3 * i + j] + S[3 * j + i]);
  }
  return det;
}

/* per-element rotation, warped stiffness and force; out pointers may be NULL */
void orc_fem_element(void *h, int e, const double *u, double *Rout, double *Ke, double *fe) {
  OrcFem *s = (OrcFem *)h;
  const int *t = s->tets + 4 * e; const double *Mi = s->Minv + 16 * (size_t)e, *K0 = s->K0 + 144 * (size_t)e;
  double P[12], F[9], R[9], RK[144], KE[144];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) P[4 * i + j] = s->x0[3 * t[j] + i] + u[3 * t[j] + i];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 4; k++) a += P[4 * i + k] * Mi[4 * k + j]; F[3 * i + j] = a; }
  if (s->linear) {  /* :429-446: KElement = KElementUndeformed, fElement = KElement u */
    if (fe) for (int i = 0; i < 12; i++) {
      double a = 0;
      for (int j = 0; j < 4; j++) a += K0[12 * i + 3 * j + 0] * u[3 * t[j] + 0] + K0[12 * i + 3 * j + 1] * u[3 * t[j] + 1] + K0[12 * i + 3 * j + 2] * u[3 * t[j] + 2];
      fe[i] = a;
    }
    if (Rout) { memset(Rout, 0, 9 * sizeof(double)); Rout[0] = Rout[4] = Rout[8] = 1.0; }
    if (Ke) memcpy(Ke, K0, sizeof KE);
    return;
  }
  double S[9];
  double det = orc_polar(F, R, S, 1e-6);
  if (det < 0) for (int i = 0; i < 9; i++) R[i] *= -1.0;
  memset(RK, 0, sizeof RK); memset(KE, 0, sizeof KE);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
    for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) for (int m = 0; m < 3; m++)
      RK[12 * (3 * i + k) + 3 * j + l] += R[3 * k + m] * K0[12 * (3 * i + m) + 3 * j + l];
    for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) for (int m = 0; m < 3; m++)
      KE[12 * (3 * i + k) + 3 * j + l] += RK[12 * (3 * i + k) + 3 * j + m] * R[3 * l + m];
  }
  if (fe) for (int i = 0; i < 12; i++) {
    double a = 0;
    for (int j = 0; j < 4; j++) for (int l = 0; l < 3; l++) a += KE[12 * i + 3 * j + l] * P[4 * l + j] - RK[12 * i + 3 * j + l] * s->x0[3 * t[j] + l];
    fe[i] = a;
  }
  if (s->exact && Ke) {
    /* corotationalLinearFEM.cpp:296-428: K += d(R)/dx terms.  G = (tr(S) I - S) R^T; for every entry (i,j) of F the
     * rotation's derivative is skew(omega_ij) R with G omega_ij = 2 skew_part(e_j r_i^T) (r_i = row i of R placed in column j);
     * chained with dF/dx = rows of Minv to dR/dx_l for the 12 element DOFs l = 3 k + j */
    double G[9], T[9], invG[9], rhs[27], omega[27], dRdF[81], dRdx[108];
    double tr = S[0] + S[4] + S[8];
    for (int i = 0; i < 9; i++) T[i] = -S[i];
    T[0] += tr; T[4] += tr; T[8] += tr;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) G[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
    inv3(G, invG);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < 3; k++) tmp[3 * k + j] = R[3 * i + k];
      double *w = rhs + 3 * (3 * i + j);  /* 2 * SKEW_PART */
      w[0] = tmp[7] - tmp[5]; w[1] = tmp[2] - tmp[6]; w[2] = tmp[3] - tmp[1];
    }
    for (int c = 0; c < 9; c++) for (int a = 0; a < 3; a++) omega[3 * c + a] = invG[3 * a] * rhs[3 * c] + invG[3 * a + 1] * rhs[3 * c + 1] + invG[3 * a + 2] * rhs[3 * c + 2];
    for (int c = 0; c < 9; c++) {
      const double *w = omega + 3 * c;
      double sk[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) dRdF[9 * c + 3 * a + b] = sk[3 * a] * R[b] + sk[3 * a + 1] * R[3 + b] + sk[3 * a + 2] * R[6 + b];
    }
    /* B[i][j][3 k + l] = dRdF[9 (3 j + l) + (3 i + k)];  dRdx column (3 v + j), rows 3 i .. 3 i + 2 = B[i][j] minv_v */
    for (int v = 0; v < 4; v++) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) {
      double a = 0;
      for (int l = 0; l < 3; l++) a += dRdF[9 * (3 * j + l) + (3 * i + k)] * Mi[4 * v + l];
      dRdx[9 * (3 * v + j) + 3 * i + k] = a;
    }
    double tv[12], av[12];
    for (int v = 0; v < 4; v++) for (int a = 0; a < 3; a++)
      tv[3 * v + a] = R[a] * P[v] + R[3 + a] * P[4 + v] + R[6 + a] * P[8 + v] - s->x0[3 * t[v] + a];
    for (int i = 0; i < 12; i++) { double a = 0; for (int j = 0; j < 12; j++) a += K0[12 * i + j] * tv[j]; av[i] = a; }
    for (int c = 0; c < 12; c++) {  /* term 1 */
      const double *D = dRdx + 9 * c;
      for (int j = 0; j < 4; j++) for (int a = 0; a < 3; a++)
        KE[12 * (3 * j + a) + c] += D[3 * a] * av[3 * j] + D[3 * a + 1] * av[3 * j + 1] + D[3 * a + 2] * av[3 * j + 2];
    }
    for (int v = 0; v < 4; v++) for (int a = 0; a < 3; a++) av[3 * v + a] = P[4 * a + v];
    for (int c = 0; c < 12; c++) {  /* term 2 */
      const double *D = dRdx + 9 * c;
      double b[12];
      for (int j = 0; j < 4; j++) for (int a = 0; a < 3; a++) b[3 * j + a] = D[a] * av[3 * j] + D[3 + a] * av[3 * j + 1] + D[6 + a] * av[3 * j + 2];
      for (int row = 0; row < 12; row++) { double a = 0; for (int j = 0; j < 12; j++) a += RK[12 * row + j] * b[j]; KE[12 * row + c] += a; }
    }
  }
  if (Rout) memcpy(Rout, R, sizeof R);
  if (Ke) memcpy(Ke, KE, sizeof KE);
}

void orc_fem_set_warp(void *h, int warp) { OrcFem *s = (OrcFem *)h; s->linear = warp == 0; s->exact = warp == 2; }
void orc_fem_set_linear(void *h, int linear) { ((OrcFem *)h)->linear = linear; ((OrcFem *)h)->exact = 0; }

/* f (may be NULL) and K values on the CSR pattern (may be NULL): zeroed, then accumulated in element order */
void orc_fem_assemble(void *h, const double *u, double *f, double *Kv) {
  OrcFem *s = (OrcFem *)h;
  if (f) memset(f, 0, sizeof(double) * s->r);
  if (Kv) memset(Kv, 0, sizeof(double) * (size_t)s->nnz);
  for (int e = 0; e < s->nt; e++) {
    double KE[144], fe[12]; const int *t = s->tets + 4 * e;
    orc_fem_element(h, e, u, NULL, KE, fe);
    if (f) for (int j = 0; j < 4; j++) for (int l = 0; l < 3; l++) f[3 * t[j] + l] += fe[3 * j + l];
    if (Kv) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
      int b = s->elblk[16 * (size_t)e + 4 * i + j], a = t[i], pos = b - s->bptr[a];
      for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) Kv[s->ia[3 * a + k] + 3 * pos + l] += KE[12 * (3 * i + k) + 3 * j + l];
    }
  }
}

void orc_spmv(int n, const int *ia, const int *ja, const double *a, const double *x, double *y) {
  for (int i = 0; i < n; i++) { double t = 0; for (int k = ia[i]; k < ia[i + 1]; k++) t += x[ja[k]] * a[k]; y[i] = t; }
}

/* Jacobi PCG; work = 4n doubles (r, d, q, invDiag).  Returns +iterations if converged, -iterations if not. */
int orc_pcg(int n, const int *ia, const int *ja, const double *a, const double *b, double *x, double eps, int maxit, double *work) {
  double *r = work, *d = work + n, *q = work + 2 * (size_t)n, *inv = work + 3 * (size_t)n;
  for (int i = 0; i < n; i++) { double dg = 0; for (int k = ia[i]; k < ia[i + 1]; k++) if (ja[k] == i) dg = a[k]; inv[i] = 1.0 / dg; }
  int it = 1;
  orc_spmv(n, ia, ja, a, x, r);
  for (int i = 0; i < n; i++) { r[i] = b[i] - r[i]; d[i] = inv[i] * r[i]; }
  double rn = 0; for (int i = 0; i < n; i++) rn += r[i] * r[i] * inv[i];
  double rn0 = rn;
  while (rn > eps * eps * rn0 && it <= maxit) {
    orc_spmv(n, ia, ja, a, d, q);
    double dq = 0; for (int i = 0; i < n; i++) dq += d[i] * q[i];
    double alpha = rn / dq;
    for (int i = 0; i < n; i++) x[i] += alpha * d[i];
    if (it % 30 == 0) { orc_spmv(n, ia, ja, a, x, r); for (int i = 0; i < n; i++) r[i] = b[i] - r[i]; }
    else for (int i = 0; i < n; i++) r[i] = r[i] - alpha * q[i];
    double old = rn; rn = 0; for (int i = 0; i < n; i++) rn += r[i] * r[i] * inv[i];
    double beta = rn / old;
    for (int i = 0; i < n; i++) d[i] = inv[i] * r[i] + beta * d[i];
    it++;
  }
  return (it - 1) * ((rn > eps * eps * rn0) ? -1 : 1);
}

/* fixed DOFs: ascending, 0-indexed (implicitNewmarkSparse.h:78-80) */
int orc_integrator_create(void *hh, int nfixed, const int *fixedDOFs, double timestep, double cM, double cK) {
  OrcFem *s = (OrcFem *)hh; int r = s->r;
  for (int i = 0; i < nfixed; i++) if (fixedDOFs[i] < 0 || fixedDOFs[i] >= r || (i && fixedDOFs[i] <= fixedDOFs[i - 1])) return -1;
  s->h = timestep; s->cM = cM; s->cK = cK; s->scale = 1.0; s->nfixed = nfixed;
  s->fixed = (int *)malloc(sizeof(int) * (nfixed > 0 ? nfixed : 1)); memcpy(s->fixed, fixedDOFs, sizeof(int) * nfixed);
  int *old2new = (int *)malloc(sizeof(int) * r); int c = 0, fi = 0;
  for (int i = 0; i < r; i++) { if (fi < nfixed && fixedDOFs[fi] == i) { old2new[i] = -1; fi++; } else old2new[i] = c++; }
  s->rc = c; s->c2f = (int *)malloc(sizeof(int) * c);
  for (int i = 0; i < r; i++) if (old2new[i] >= 0) s->c2f[old2new[i]] = i;
  s->cia = (int *)malloc(sizeof(int) * (c + 1)); int nz = 0;
  for (int i = 0; i < r; i++) if (old2new[i] >= 0) for (int k = s->ia[i]; k < s->ia[i + 1]; k++) if (old2new[s->ja[k]] >= 0) nz++;
  s->nnzc = nz; s->cja = (int *)malloc(sizeof(int) * (size_t)nz); s->csrc = (int *)malloc(sizeof(int) * (size_t)nz); nz = 0;
  for (int i = 0; i < r; i++) if (old2new[i] >= 0) {
    s->cia[old2new[i]] = nz;
    for (int k = s->ia[i]; k < s->ia[i + 1]; k++) if (old2new[s->ja[k]] >= 0) { s->cja[nz] = old2new[s->ja[k]]; s->csrc[nz] = k; nz++; }
  }
  s->cia[c] = nz; free(old2new);
  s->K = (double *)calloc(s->nnz, sizeof(double)); s->D = (double *)calloc(s->nnz, sizeof(double)); s->sys = (double *)calloc(nz, sizeof(double));
  s->q = (double *)calloc(r, sizeof(double)); s->qvel = (double *)calloc(r, sizeof(double)); s->fext = (double *)calloc(r, sizeof(double));
  s->fint = (double *)calloc(r, sizeof(double)); s->qres = (double *)calloc(r, sizeof(double)); s->qdelta = (double *)calloc(r, sizeof(double));
  s->buf = (double *)calloc(r, sizeof(double)); s->bufc = (double *)calloc(c > 0 ? c : 1, sizeof(double));
  s->cg_r = (double *)calloc(4 * (size_t)(c > 0 ? c : 1), sizeof(double));
  return 0;
}

void orc_set_state(void *hh, const double *q, const double *qvel) { OrcFem *s = (OrcFem *)hh; memcpy(s->q, q, sizeof(double) * s->r); if (qvel) memcpy(s->qvel, qvel, sizeof(double) * s->r); }
void orc_get_state(void *hh, double *q, double *qvel) { OrcFem *s = (OrcFem *)hh; if (q) memcpy(q, s->q, sizeof(double) * s->r); if (qvel) memcpy(qvel, s->qvel, sizeof(double) * s->r); }
void orc_set_external_forces(void *hh, const double *f) { OrcFem *s = (OrcFem *)hh; memcpy(s->fext, f, sizeof(double) * s->r); }
int orc_sys_rows(void *hh) { return ((OrcFem *)hh)->rc; }
int orc_sys_nnz(void *hh) { return ((OrcFem *)hh)->nnzc; }
void orc_sys_csr(void *hh, int *ia, int *ja, double *a) { OrcFem *s = (OrcFem *)hh; memcpy(ia, s->cia, sizeof(int) * (s->rc + 1)); memcpy(ja, s->cja, sizeof(int) * (size_t)s->nnzc); if (a) memcpy(a, s->sys, sizeof(double) * (size_t)s->nnzc); }

/* Builds Keff and the right-hand side exactly in the reference's operation order; no solve. */
static void orc_build_system(OrcFem *s) {
  int r = s->r; size_t nnz = (size_t)s->nnz;
  orc_fem_assemble(s, s->q, s->fint, s->K);
  for (int i = 0; i < r; i++) s->fint[i] *= s->scale;
  for (size_t k = 0; k < nnz; k++) s->K[k] *= s->scale;
  memset(s->qres, 0, sizeof(double) * r);
  for (size_t k = 0; k < nnz; k++) s->D[k] = s->K[k] * s->cK;
  for (int v = 0; v < s->nv; v++) for (int k = 0; k < 3; k++) for (int p = s->bptr[v]; p < s->bptr[v + 1]; p++)
    s->D[s->ia[3 * v + k] + 3 * (p - s->bptr[v]) + k] += s->cM * s->mblk[p];
  for (size_t k = 0; k < nnz; k++) s->K[k] *= s->h;
  for (size_t k = 0; k < nnz; k++) s->K[k] += s->D[k];
  orc_spmv(r, s->ia, s->ja, s->K, s->qvel, s->qres);
  for (size_t k = 0; k < nnz; k++) s->K[k] *= s->h;
  for (int v = 0; v < s->nv; v++) for (int k = 0; k < 3; k++) for (int p = s->bptr[v]; p < s->bptr[v + 1]; p++)
    s->K[s->ia[3 * v + k] + 3 * (p - s->bptr[v]) + k] += 1.0 * s->mblk[p];
  for (int i = 0; i < r; i++) { s->qres[i] += s->fint[i] - s->fext[i]; s->qres[i] *= -s->h; s->qdelta[i] = s->qres[i]; }
}

/* one semi-implicit step; optional outputs keff (nnz), rhs (r), dv (r).  Return value as orc_pcg. */
int orc_step(void *hh, double cg_eps, int cg_maxiter, double *keff, double *rhs, double *dv) {
  OrcFem *s = (OrcFem *)hh; int r = s->r;
  orc_build_system(s);
  if (keff) memcpy(keff, s->K, sizeof(double) * (size_t)s->nnz);
  if (rhs) memcpy(rhs, s->qres, sizeof(double) * r);
  for (int i = 0; i < s->rc; i++) s->bufc[i] = s->qdelta[s->c2f[i]];
  for (int k = 0; k < s->nnzc; k++) s->sys[k] = s->K[s->csrc[k]];
  memset(s->buf, 0, sizeof(double) * r);
  int info = orc_pcg(s->rc, s->cia, s->cja, s->sys, s->bufc, s->buf, cg_eps, cg_maxiter, s->cg_r);
  memset(s->qdelta, 0, sizeof(double) * r);
  for (int i = 0; i < s->rc; i++) s->qdelta[s->c2f[i]] = s->buf[i];
  if (dv) memcpy(dv, s->qdelta, sizeof(double) * r);
  for (int i = 0; i < r; i++) { s->qvel[i] += s->qdelta[i]; s->q[i] += s->h * s->qvel[i]; }
  for (int i = 0; i < s->nfixed; i++) s->q[s->fixed[i]] = s->qvel[s->fixed[i]] = 0.0;
  return info;
}

/* ImplicitNewmarkSparse::DoTimestep (implicitNewmarkSparse.cpp:183-379) with the PCG solver; alphas of UpdateAlphas (:96-104).
 * As there the solver's start vector `buf` is not cleared between solves.  Returns the Newton iterations performed or -1. */
void orc_get_accel(void *hh, double *qa) { OrcFem *s = (OrcFem *)hh; if (!s->qaccel) s->qaccel = (double *)calloc(s->r, sizeof(double)); memcpy(qa, s->qaccel, sizeof(double) * s->r); }
void orc_set_accel(void *hh, const double *qa) { OrcFem *s = (OrcFem *)hh; if (!s->qaccel) s->qaccel = (double *)calloc(s->r, sizeof(double)); memcpy(s->qaccel, qa, sizeof(double) * s->r); }

int orc_newmark_step(void *hh, double beta, double gamma, int max_newton, double epsilon, double cg_eps, int cg_maxiter, int *cg_total) {
  OrcFem *s = (OrcFem *)hh; int r = s->r; size_t nnz = (size_t)s->nnz; double h = s->h;
  if (!s->qaccel) s->qaccel = (double *)calloc(r, sizeof(double));
  double alpha1 = 1.0 / (beta * h * h), alpha2 = 1.0 / (beta * h), alpha3 = (1.0 - 2.0 * beta) / (2.0 * beta);
  double alpha4 = gamma / (beta * h), alpha5 = 1 - gamma / beta, alpha6 = (1.0 - gamma / (2.0 * beta)) * h;
  double *q1 = (double *)malloc(sizeof(double) * 3 * (size_t)r), *v1 = q1 + r, *a1 = v1 + r;
  memcpy(q1, s->q, sizeof(double) * r); memcpy(v1, s->qvel, sizeof(double) * r); memcpy(a1, s->qaccel, sizeof(double) * r);
  for (int i = 0; i < r; i++) {
    s->qaccel[i] = alpha1 * (s->q[i] - q1[i]) - alpha2 * v1[i] - alpha3 * a1[i];
    s->qvel[i] = alpha4 * (s->q[i] - q1[i]) + alpha5 * v1[i] + alpha6 * a1[i];
  }
  int numIter = 0, total = 0, rc = 0; double error0 = 0, quot;
  do {
    orc_fem_assemble(s, s->q, s->fint, s->K);
    for (int i = 0; i < r; i++) s->fint[i] *= s->scale;
    for (size_t k = 0; k < nnz; k++) s->K[k] *= s->scale;
    memset(s->qres, 0, sizeof(double) * r);
    for (size_t k = 0; k < nnz; k++) s->D[k] = s->K[k] * s->cK;
    for (int v = 0; v < s->nv; v++) for (int k = 0; k < 3; k++) for (int p = s->bptr[v]; p < s->bptr[v + 1]; p++)
      s->D[s->ia[3 * v + k] + 3 * (p - s->bptr[v]) + k] += s->cM * s->mblk[p];
    for (size_t k = 0; k < nnz; k++) s->K[k] += alpha4 * s->D[k];
    for (int v = 0; v < s->nv; v++) for (int k = 0; k < 3; k++) for (int p = s->bptr[v]; p < s->bptr[v + 1]; p++)
      s->K[s->ia[3 * v + k] + 3 * (p - s->bptr[v]) + k] += alpha1 * s->mblk[p];
    for (int v = 0; v < s->nv; v++) for (int k = 0; k < 3; k++) {  /* M qaccel */
      double a = 0;
      for (int p = s->bptr[v]; p < s->bptr[v + 1]; p++) a += s->qaccel[3 * s->bcol[p] + k] * s->mblk[p];
      s->qres[3 * v + k] = a;
    }
    for (int i = 0; i < r; i++) { double a = 0; for (int k = s->ia[i]; k < s->ia[i + 1]; k++) a += s->qvel[s->ja[k]] * s->D[k]; s->qres[i] += a; }
    for (int i = 0; i < r; i++) { s->qres[i] += s->fint[i] - s->fext[i]; s->qres[i] *= -1; s->qdelta[i] = s->qres[i]; }
    double error = 0;
    for (int i = 0; i < r; i++) error += s->qres[i] * s->qres[i];
    if (numIter == 0) { error0 = error; quot = 1.0; } else quot = error / error0;
    if (quot < epsilon * epsilon) break;
    for (int i = 0; i < s->rc; i++) s->bufc[i] = s->qdelta[s->c2f[i]];
    for (int k = 0; k < s->nnzc; k++) s->sys[k] = s->K[s->csrc[k]];
    int info = orc_pcg(s->rc, s->cia, s->cja, s->sys, s->bufc, s->buf, cg_eps, cg_maxiter, s->cg_r);
    if (info < 0) { total -= info; rc = -1; break; }
    total += info;
    memset(s->qdelta, 0, sizeof(double) * r);
    for (int i = 0; i < s->rc; i++) s->qdelta[s->c2f[i]] = s->buf[i];
    for (int i = 0; i < r; i++) {
      s->q[i] += s->qdelta[i];
      s->qaccel[i] = alpha1 * (s->q[i] - q1[i]) - alpha2 * v1[i] - alpha3 * a1[i];
      s->qvel[i] = alpha4 * (s->q[i] - q1[i]) + alpha5 * v1[i] + alpha6 * a1[i];
    }
    for (int i = 0; i < s->nfixed; i++) s->q[s->fixed[i]] = s->qvel[s->fixed[i]] = s->qaccel[s->fixed[i]] = 0.0;
    numIter++;
  } while (numIter < max_newton);
  free(q1);
  if (cg_total) *cg_total = total;
  return rc < 0 ? -1 : numIter;
}

/* bench helper: build the system once and time `iters` PCG iterations' worth of work (SpMV + vector ops)
 * without converging -- used only for the bounded cpu_baseline sample in bench.py */
int orc_step_prepare(void *hh) {
  OrcFem *s = (OrcFem *)hh;
  orc_build_system(s);
  for (int i = 0; i < s->rc; i++) s->bufc[i] = s->qdelta[s->c2f[i]];
  for (int k = 0; k < s->nnzc; k++) s->sys[k] = s->K[s->csrc[k]];
  memset(s->buf, 0, sizeof(double) * s->r);
  return 0;
}
int orc_pcg_bounded(void *hh, double eps, int maxit) {
  OrcFem *s = (OrcFem *)hh;
  memset(s->buf, 0, sizeof(double) * s->r);
  return orc_pcg(s->rc, s->cia, s->cja, s->sys, s->bufc, s->buf, eps, maxit, s->cg_r);
}
