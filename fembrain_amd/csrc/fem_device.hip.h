// FEM device kernels (included by fem.hip only).
#pragma once
#include <type_traits>
#include "fem_kernels.h"
#include "p2p_device.hip.h"

namespace fb {

// ------------------------------------------------------------------------------------------------------
// a1 (rest state): per tet shape-function gradients b_k (rows of M^-1, vegafem corotationalLinearFEM.cpp:66-90)
// and volume |det|/6 (tetMesh.cpp:184-188).  rest[16*e + 3*k + d] = b_k[d], rest[16*e + 12] = V,
// rest[16*e + 13 + ...] unused.  One thread per tet.
// ------------------------------------------------------------------------------------------------------
__device__ inline void inv3x3(const double* A, double* I) {
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double id = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
  I[0] = c00 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  I[3] = c01 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  I[6] = c02 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// first_flat (may be null): receives the lowest index of an element whose rest volume is zero or not finite -- the check
// build() otherwise makes on the host, same expression
// volf (may be null): the rest volume as the fp32 records hold it, one float per element (k_mass_blocks gathers it: 4 MB at 1M tets
// instead of one line of every 128-byte rest record)
__global__ __launch_bounds__(kBlock) void k_tet_rest(int nt, const int4* __restrict__ tets, const double* __restrict__ x0,
                                                     double* __restrict__ rest, int* __restrict__ first_flat, float* __restrict__ volf) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= nt) return;
  const int4 t = tets[e];
  const int id[4] = {t.x, t.y, t.z, t.w};
  double p[4][3];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int d = 0; d < 3; d++) p[k][d] = x0[3 * (size_t)id[k] + d];
  if (first_flat) {
    double a[3], b[3], c[3];
    for (int k = 0; k < 3; k++) { a[k] = p[1][k] - p[0][k]; b[k] = p[2][k] - p[0][k]; c[k] = p[3][k] - p[0][k]; }
    const double det = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
    if (!(det != 0.0) || !isfinite(det)) atomicMin(first_flat, e);
  }
  double Dm[9], Di[9];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    Dm[3 * d + 0] = p[1][d] - p[0][d];
    Dm[3 * d + 1] = p[2][d] - p[0][d];
    Dm[3 * d + 2] = p[3][d] - p[0][d];
  }
  inv3x3(Dm, Di);
  double* r = rest + 16 * (size_t)e;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    r[3 + d] = Di[d];
    r[6 + d] = Di[3 + d];
    r[9 + d] = Di[6 + d];
    r[d] = -(Di[d] + Di[3 + d] + Di[6 + d]);
  }
  // volume = 1/6 |(a-d).((b-d)x(c-d))|
  const double u[3] = {p[0][0] - p[3][0], p[0][1] - p[3][1], p[0][2] - p[3][2]};
  const double v[3] = {p[1][0] - p[3][0], p[1][1] - p[3][1], p[1][2] - p[3][2]};
  const double w[3] = {p[2][0] - p[3][0], p[2][1] - p[3][1], p[2][2] - p[3][2]};
  const double cx = v[1] * w[2] - v[2] * w[1], cy = v[2] * w[0] - v[0] * w[2], cz = v[0] * w[1] - v[1] * w[0];
  r[12] = (1.0 / 6) * fabs(u[0] * cx + u[1] * cy + u[2] * cz);
  if (volf) volf[e] = (float)r[12];
  r[13] = r[14] = r[15] = 0.0;
}

// scaled-Newton polar decomposition of F (row-major), R out; returns last determinant
// (vegafem polarDecomposition.cpp:37-108; the iteration is data dependent, capped for safety)
__device__ inline double one_norm3(const double* A) {
  return fmax(fmax(fabs(A[0]) + fabs(A[3]) + fabs(A[6]), fabs(A[1]) + fabs(A[4]) + fabs(A[7])), fabs(A[2]) + fabs(A[5]) + fabs(A[8]));
}
__device__ inline double inf_norm3(const double* A) {
  return fmax(fmax(fabs(A[0]) + fabs(A[1]) + fabs(A[2]), fabs(A[3]) + fabs(A[4]) + fabs(A[5])), fabs(A[6]) + fabs(A[7]) + fabs(A[8]));
}

__device__ inline double polar_rotation(const double* F, double* R, double tol) {
  double Mk[9], A[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Mk[3 * i + j] = F[3 * j + i];
  double M1 = one_norm3(Mk), Mi = inf_norm3(Mk), det = 0.0, E1;
  int guard = 0;
  do {
    A[0] = Mk[4] * Mk[8] - Mk[5] * Mk[7]; A[1] = Mk[5] * Mk[6] - Mk[3] * Mk[8]; A[2] = Mk[3] * Mk[7] - Mk[4] * Mk[6];
    A[3] = Mk[7] * Mk[2] - Mk[8] * Mk[1]; A[4] = Mk[8] * Mk[0] - Mk[6] * Mk[2]; A[5] = Mk[6] * Mk[1] - Mk[7] * Mk[0];
    A[6] = Mk[1] * Mk[5] - Mk[2] * Mk[4]; A[7] = Mk[2] * Mk[3] - Mk[0] * Mk[5]; A[8] = Mk[0] * Mk[4] - Mk[1] * Mk[3];
    det = Mk[0] * A[0] + Mk[1] * A[1] + Mk[2] * A[2];
    if (det == 0.0) break;
    const double A1 = one_norm3(A), Ai = inf_norm3(A);
    const double gamma = sqrt(sqrt((A1 * Ai) / (M1 * Mi)) / fabs(det));
    const double g1 = gamma * 0.5, g2 = 0.5 / (gamma * det);
    double Ek[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      Ek[i] = Mk[i];
      Mk[i] = g1 * Mk[i] + g2 * A[i];
      Ek[i] -= Mk[i];
    }
    E1 = one_norm3(Ek);
    M1 = one_norm3(Mk);
    Mi = inf_norm3(Mk);
  } while (E1 > M1 * tol && ++guard < 64);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) R[3 * i + j] = Mk[3 * j + i];
  return det;
}

// ------------------------------------------------------------------------------------------------------
// a3/a4 pass 1, one thread per tet: F = P M^-1, R = polar(F) (flipped if det < 0), rotated gradients
// c_k = R b_k and the element force f_e = R K0 (R^T x - x0) in closed form
//   f_e,i = V R [lambda tr(H) I + mu (H + H^T)] b_i,   H = sum_j (R^T x_j - x0_j) b_j^T
// (equal to corotationalLinearFEM.cpp:238-286 with K0 = V B^T E B, whose 3x3 blocks are
//   K0[ij] = V [lambda b_i b_j^T + mu b_j b_i^T + mu (b_i.b_j) I]).
// rec[16*e + 4*k + d] = c_k[d] (MT), rec[16*e + 4*k + 3] = V; fe[12*e + 3*k + d] fp64.
// ------------------------------------------------------------------------------------------------------
// y_i = sum_j K0[ij] v_j with the undeformed element stiffness in closed form, K0[ij] = V [lambda b_i b_j^T + mu b_j b_i^T + mu (b_i.b_j) I]
__device__ inline void k0_apply(const double b[4][3], double V, double lambda, double mu, const double* v, double* y) {
  double sl = 0.0, sm[3] = {0, 0, 0}, sg[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const double bv = b[j][0] * v[3 * j] + b[j][1] * v[3 * j + 1] + b[j][2] * v[3 * j + 2];
    sl += bv;                                   // sum_j b_j . v_j
#pragma unroll
    for (int a = 0; a < 3; a++) {
      sm[a] += b[j][a] * bv;                    // unused below (kept symmetric form): sum_j b_j (b_j . v_j)
#pragma unroll
      for (int c = 0; c < 3; c++) sg[a][c] += b[j][a] * v[3 * j + c];  // sum_j b_j v_j^T
    }
  }
  (void)sm;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      // lambda b_i (sum_j b_j.v_j) + mu sum_j b_j (b_i.v_j) + mu sum_j (b_i.b_j) v_j
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int c = 0; c < 3; c++) { t1 += sg[a][c] * b[i][c]; t2 += b[i][c] * sg[c][a]; }
      y[3 * i + a] = V * (lambda * b[i][a] * sl + mu * t1 + mu * t2);
    }
}

// kcorr (may be null): warp = 2 of CorotationalLinearFEMForceModel (corotationalLinearFEM.cpp:296-428) -- the two terms the
// derivative of the element rotation adds to the element stiffness, as a row-major 12 x 12 matrix per element (MT):
//   term 1: column l = blockdiag(dR/dx_l) K0 (R^T x - x0),   term 2: column l = R K0 blockdiag(dR/dx_l)^T x
// with dR/dF from G omega = 2 skew_part(.), G = (tr(S) I - S) R^T (S = the symmetric polar factor R^T F, R before the flip).
// TANGENT (warp = 2) is a template parameter: as a run-time branch its registers (512 + scratch) cost the default kernel a third of
// its speed
template <typename MT, bool TANGENT>
__global__ __launch_bounds__(kBlock) void k_tet_warp(int nt, const int4* __restrict__ tets, const double* __restrict__ x0,
                                                     const double* __restrict__ u, const double* __restrict__ rest,
                                                     MT* __restrict__ rec, double* __restrict__ fe, double* __restrict__ rot,
                                                     double lambda, double mu, int linear, MT* __restrict__ kcorr) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= nt) return;
  const int4 t = tets[e];
  const int id[4] = {t.x, t.y, t.z, t.w};
  double b[4][3], X0[4][3], P[4][3];
  const double* r = rest + 16 * (size_t)e;
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      b[k][d] = r[3 * k + d];
      X0[k][d] = x0[3 * (size_t)id[k] + d];
      P[k][d] = X0[k][d] + u[3 * (size_t)id[k] + d];
    }
  const double V = r[12];
  double F[9], R[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) F[3 * i + j] = P[0][i] * b[0][j] + P[1][i] * b[1][j] + P[2][i] * b[2][j] + P[3][i] * b[3][j];
  if (linear) {  // warp = 0 (corotationalLinearFEM.cpp:429-453): R = I, so c_k = b_k and f_e = K0 u
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    const double det = polar_rotation(F, R, 1e-6);
    if (TANGENT) {
      // S = sym(Q^T F) with Q the rotation BEFORE the flip (PolarDecomposition::Compute returns it that way and the
      // reference flips R only, corotationalLinearFEM.cpp:262-268)
      double S[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) S[3 * i + j] = R[i] * F[j] + R[3 + i] * F[3 + j] + R[6 + i] * F[6 + j];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i + 1; j < 3; j++) S[3 * i + j] = S[3 * j + i] = 0.5 * (S[3 * i + j] + S[3 * j + i]);
      double Rf[9];
#pragma unroll
      for (int i = 0; i < 9; i++) Rf[i] = det < 0 ? -R[i] : R[i];
      const double tr = S[0] + S[4] + S[8];
      double T[9], G[9], Gi[9];
#pragma unroll
      for (int i = 0; i < 9; i++) T[i] = ((i % 4 == 0) ? tr : 0.0) - S[i];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) G[3 * i + j] = T[3 * i] * Rf[3 * j] + T[3 * i + 1] * Rf[3 * j + 1] + T[3 * i + 2] * Rf[3 * j + 2];
      inv3x3(G, Gi);
      // a = K0 (R^T x - x0), xs = current positions
      double tv[12], av[12], xs[12];
#pragma unroll
      for (int v = 0; v < 4; v++)
#pragma unroll
        for (int a = 0; a < 3; a++) {
          tv[3 * v + a] = Rf[a] * P[v][0] + Rf[3 + a] * P[v][1] + Rf[6 + a] * P[v][2] - X0[v][a];
          xs[3 * v + a] = P[v][a];
        }
      k0_apply(b, V, lambda, mu, tv, av);
      MT* C = kcorr + 144 * (size_t)e;
      for (int v = 0; v < 4; v++)
        for (int j = 0; j < 3; j++) {
          // D = dR/dx_l, l = 3 v + j: rows 3i..3i+2 (i = 0..2) = B[i][j] b_v, B[i][j][3k+l'] = dRdF[column 3j+l'][row 3i+k],
          // dRdF column c = (i', j') : skew(omega_c) R, G omega_c = 2 skew_part(e_j' r_i'^T)
          double D[9];
#pragma unroll
          for (int q = 0; q < 9; q++) D[q] = 0.0;
          for (int lp = 0; lp < 3; lp++) {        // column c = 3 j + lp of dRdF, i.e. F entry (row j, col lp)
            // tmp = matrix with row j of R in column lp: tmp[3k + lp] = R[3j + k];  w = 2 skew_part(tmp)
            double w[3] = {0, 0, 0};
            {
              double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
              for (int k = 0; k < 3; k++) tmp[3 * k + lp] = Rf[3 * j + k];
              w[0] = tmp[7] - tmp[5]; w[1] = tmp[2] - tmp[6]; w[2] = tmp[3] - tmp[1];
            }
            const double om[3] = {Gi[0] * w[0] + Gi[1] * w[1] + Gi[2] * w[2], Gi[3] * w[0] + Gi[4] * w[1] + Gi[5] * w[2],
                                  Gi[6] * w[0] + Gi[7] * w[1] + Gi[8] * w[2]};
            const double sk[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
            const double bl = b[v][lp];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
              for (int k = 0; k < 3; k++)   // (skew R)[i][k] * b_v[lp]: entry (row 3i+k of the 9-vector) -> D[i][k]
                D[3 * i + k] += (sk[3 * i] * Rf[k] + sk[3 * i + 1] * Rf[3 + k] + sk[3 * i + 2] * Rf[6 + k]) * bl;
          }
          const int col = 3 * v + j;
          double bb[12], rk[12];
#pragma unroll
          for (int w4 = 0; w4 < 4; w4++)
#pragma unroll
            for (int a = 0; a < 3; a++) bb[3 * w4 + a] = D[a] * xs[3 * w4] + D[3 + a] * xs[3 * w4 + 1] + D[6 + a] * xs[3 * w4 + 2];  // D^T x_w
          k0_apply(b, V, lambda, mu, bb, rk);
#pragma unroll
          for (int w4 = 0; w4 < 4; w4++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
              const double t1 = D[3 * a] * av[3 * w4] + D[3 * a + 1] * av[3 * w4 + 1] + D[3 * a + 2] * av[3 * w4 + 2];
              const double t2 = Rf[3 * a] * rk[3 * w4] + Rf[3 * a + 1] * rk[3 * w4 + 1] + Rf[3 * a + 2] * rk[3 * w4 + 2];  // R (K0 bb)
              C[12 * (3 * w4 + a) + col] = (MT)(t1 + t2);
            }
        }
    }
    if (det < 0) {
#pragma unroll
      for (int i = 0; i < 9; i++) R[i] = -R[i];
    }
  }
  // H = sum_j y_j b_j^T with y_j = R^T P_j - X0_j
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double y[3];
#pragma unroll
    for (int a = 0; a < 3; a++) y[a] = R[a] * P[j][0] + R[3 + a] * P[j][1] + R[6 + a] * P[j][2] - X0[j][a];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) H[3 * a + c] += y[a] * b[j][c];
  }
  const double tr = H[0] + H[4] + H[8];
  double S[9];  // lambda tr I + mu (H + H^T)
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 3; c++) S[3 * a + c] = mu * (H[3 * a + c] + H[3 * c + a]) + (a == c ? lambda * tr : 0.0);
  MT* rc = rec + 16 * (size_t)e;
  double* f = fe + 12 * (size_t)e;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double sb[3], c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      sb[a] = S[3 * a] * b[k][0] + S[3 * a + 1] * b[k][1] + S[3 * a + 2] * b[k][2];
      c[a] = R[3 * a] * b[k][0] + R[3 * a + 1] * b[k][1] + R[3 * a + 2] * b[k][2];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
      f[3 * k + a] = V * (R[3 * a] * sb[0] + R[3 * a + 1] * sb[1] + R[3 * a + 2] * sb[2]);
      rc[4 * k + a] = (MT)c[a];
    }
    rc[4 * k + 3] = (MT)V;  // V rides in the 4th lane of every gradient: one 4-wide load gives (c_k, V)
  }
  if (rot) {
#pragma unroll
    for (int i = 0; i < 9; i++) rot[9 * (size_t)e + i] = R[i];
  }
}

// ------------------------------------------------------------------------------------------------------
// a3 scatter + a7 algebra, pass 2 ("row-gather"): lane = block row a.  For every slot (a,b) the lane sums its
// element contributions in ascending element order
//   K_ab = sum_e V_e [lambda c_i c_j^T + mu c_j c_i^T + mu (c_i.c_j) I],   m_ab = sum_e rho V_e/20 (1 + delta_ij)
// and writes, in one pass and with no atomics (deterministic):
//   vals   <- s_k K_ab + s_m m_ab I      (Keff = h(h+cK) K + (1+h cM) M; raw mode: s_k=1, s_m=0)
//             with constrained rows/columns replaced by identity rows (the reference removes them,
//             sparseMatrix.cpp:1296-1358 + :993-1002; identical solution, same arithmetic on the free DOFs)
//   t_a    += (g_k K_ab + g_m m_ab I) qvel_b      ((hK + D) qvel of PS_VolumeConservingIntegrator.cpp:114)
//   fint_a  = sum over the diagonal list of f_e,i  (the element-force scatter of corotationalLinearFEM.cpp:289-293)
//   rhs_a   = -h (t_a + fint_a - fext_a) on free DOFs, 0 on constrained ones;   invdiag_a = 1 / diag(vals_aa)
// ------------------------------------------------------------------------------------------------------
struct AsmParams {
  double lambda, mu, rho20;  // rho/20
  double s_k, s_m;           // stored matrix = s_k K + s_m M
  double g_k, g_m;           // rhs operator on qvel = g_k K + g_m M
  double g_a;                // rhs operator on qacc = g_a M (Newmark: M qaccel), 0 otherwise
  double rhs_scale;          // rhs = rhs_scale (t + fint - fext): -h (FemBrain's step), -1 (Newmark)
  int apply_mask;
};

// what both assembly kernels (k_assemble_rows: slot-major, k_assemble_tets: element-major) write
template <typename MT>
struct AsmOut {
  const uint8_t* dofmask;
  const uint8_t* nodemask;  // dofmask of a node packed into bits 0..2
  const double* qvel;
  const double* fext;
  const double* qacc;
  MT* vals;
  MT* dlo;
  double* mblk_out;
  const double* mblk_in;  // k_assemble_tets_st: the per-block mass entries (k_mass_blocks), [slot][64]
  double* fint_out;
  double* rhs;
  double* res_all;  // (Newmark with several Newton iterations) the residual of EVERY DOF, the reaction forces at the clamped ones included: what implicitNewmarkSparse.cpp:258-262 sums
  double* invdiag;
  double* invblk;
};

// what RowAlgebra::block reads at a block's column
struct RowGather {
  double qv[3], qa[3];
  uint8_t mb[3];
  // the same without run-time branches (k_assemble_tets: a branch per slot ends the basic block and with it the loads in flight)
  template <typename MT, bool NEWMARK>
  __device__ __forceinline__ void load_straight(const AsmOut<MT>& o, const AsmParams& ap, int col) {
    qv[0] = o.qvel[3 * (size_t)col]; qv[1] = o.qvel[3 * (size_t)col + 1]; qv[2] = o.qvel[3 * (size_t)col + 2];
    qa[0] = qa[1] = qa[2] = 0.0;
    if (NEWMARK) { qa[0] = o.qacc[3 * (size_t)col]; qa[1] = o.qacc[3 * (size_t)col + 1]; qa[2] = o.qacc[3 * (size_t)col + 2]; }
    const uint8_t nm = o.nodemask[col] | (ap.apply_mask ? (uint8_t)0 : (uint8_t)7);
    mb[0] = nm & 1; mb[1] = (nm >> 1) & 1; mb[2] = (nm >> 2) & 1;
  }
  template <typename MT>
  __device__ __forceinline__ void load(const AsmOut<MT>& o, const AsmParams& ap, int col) {
    qv[0] = o.qvel[3 * (size_t)col]; qv[1] = o.qvel[3 * (size_t)col + 1]; qv[2] = o.qvel[3 * (size_t)col + 2];
    qa[0] = qa[1] = qa[2] = 0.0;
    if (o.qacc) { qa[0] = o.qacc[3 * (size_t)col]; qa[1] = o.qacc[3 * (size_t)col + 1]; qa[2] = o.qacc[3 * (size_t)col + 2]; }
    mb[0] = mb[1] = mb[2] = 1;
    if (ap.apply_mask) {
      const uint8_t nm = o.nodemask[col];
      mb[0] = nm & 1; mb[1] = (nm >> 1) & 1; mb[2] = (nm >> 2) & 1;
    }
  }
};

// The a7 algebra of one block row, fed one finished block (K_ab, m_ab) at a time in slot order; the arithmetic and its order
// are the same whichever kernel summed the block.
template <typename MT>
struct RowAlgebra {
  double ta[3], fi[3], dg[3], off[9], msum;
  uint8_t ma[3];
  bool seen_diag;
  int kd, nd;

  __device__ __forceinline__ void begin(const AsmOut<MT>& o, const AsmParams& ap, int row, bool rvalid) {
#pragma unroll
    for (int a = 0; a < 3; a++) { ta[a] = 0.0; fi[a] = 0.0; dg[a] = 1.0; ma[a] = 1; }
#pragma unroll
    for (int v = 0; v < 9; v++) off[v] = 0.0;  // sum of the off-diagonal blocks exactly as stored (rounded)
    msum = 0.0;                                // sum_b m_ab: the row sum the stored row must reproduce
    seen_diag = false;
    kd = -1; nd = 0;                           // slot and contribution count of the diagonal block
    if (rvalid && ap.apply_mask) {
      ma[0] = o.dofmask[3 * (size_t)row];
      ma[1] = o.dofmask[3 * (size_t)row + 1];
      ma[2] = o.dofmask[3 * (size_t)row + 2];
    }
  }
  // the true diagonal block is the first slot whose column is the row itself (padding slots repeat the row id)
  __device__ __forceinline__ bool is_diag(int row, int col, bool rvalid) {
    const bool diag = rvalid && (col == row) && !seen_diag;
    seen_diag = seen_diag || diag;
    return diag;
  }
  __device__ __forceinline__ void block(const AsmOut<MT>& o, const AsmParams& ap, int slot, int lane, const RowGather& g, bool rvalid, bool diag, const double* K,
                                        double m, int nc) {
    // rhs operator on qvel (unmasked, as the reference multiplies the full matrix with the full qvel)
    const double* qv = g.qv;
    const uint8_t* mb = g.mb;
#pragma unroll
    for (int a = 0; a < 3; a++)
      ta[a] += ap.g_k * (K[3 * a] * qv[0] + K[3 * a + 1] * qv[1] + K[3 * a + 2] * qv[2]) + ap.g_m * m * qv[a];
    if (o.qacc) {
#pragma unroll
      for (int a = 0; a < 3; a++) ta[a] += ap.g_a * m * g.qa[a];
    }
    msum += m;
    if (diag) {
      kd = slot; nd = nc;  // written by finish()
    } else {
      MT* out = o.vals + (size_t)slot * 9 * 64 + lane;
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const MT u = (MT)(ap.s_k * K[3 * a + b] + (a == b ? ap.s_m * m : 0.0));
          off[3 * a + b] += (double)u;
          out[(3 * a + b) * 64] = (ma[a] && mb[b]) ? u : (MT)0;
        }
    }
    if (o.mblk_out && rvalid) o.mblk_out[(size_t)slot * 64 + lane] = m;
  }
  // Diagonal block.  Translation invariance of the element stiffness (sum_j K0[ij] = 0) makes every block row of
  // s_k K + s_m M sum to s_m (sum_b m_ab) I; the diagonal block is therefore formed as that row sum minus the
  // off-diagonal blocks AS STORED and kept as a hi + lo pair, so that rounding the stiffness to fp32 does not give a
  // free body a spurious translational stiffness of the order of its mass term (eps_f32 * |K| vs M ~ rho L^2/(20 h^2 E) |K|).
  __device__ __forceinline__ void finish(const AsmOut<MT>& o, const AsmParams& ap, int s, int lane, int row, bool rvalid) {
    double dfull[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // the diagonal block as stored (hi + lo), for the block-Jacobi option
    if (rvalid && kd >= 0) {
      MT* out = o.vals + (size_t)kd * 9 * 64 + lane;
      MT* lo = o.dlo + (size_t)s * 9 * 64 + lane;
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
          double v = (a == b ? ap.s_m * msum : 0.0) - 0.5 * (off[3 * a + b] + off[3 * b + a]);  // symmetric part: CG needs A = A^T
          if (!(ma[a] && ma[b])) v = (a == b) ? 1.0 : 0.0;
          if (nd == 0) v = (a == b) ? 1.0 : 0.0;  // a node no element references: identity row, stays at rest
          const MT hi = (MT)v;
          const MT l = (MT)(v - (double)hi);
          out[(3 * a + b) * 64] = hi;
          lo[(3 * a + b) * 64] = l;
          dfull[3 * a + b] = (double)hi + (double)l;
          if (a == b) dg[a] = (double)hi + (double)l;
        }
    } else if (kd < 0) {
      MT* lo = o.dlo + (size_t)s * 9 * 64 + lane;  // padding lanes of the last slice
#pragma unroll
      for (int v = 0; v < 9; v++) lo[v * 64] = (MT)0;
    }
    if (rvalid) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const size_t d = 3 * (size_t)row + a;
        if (o.fint_out) o.fint_out[d] = fi[a];
        if (o.rhs) o.rhs[d] = ma[a] ? ap.rhs_scale * (ta[a] + fi[a] - o.fext[d]) : 0.0;
        if (o.res_all) o.res_all[d] = ap.rhs_scale * (ta[a] + fi[a] - o.fext[d]);
        if (o.invdiag) o.invdiag[d] = 1.0 / dg[a];
      }
      if (o.invblk) {  // FB_PCG_BLOCK_JACOBI: the inverse of the row's 3x3 diagonal block (symmetric; identity on clamped DOFs)
        double inv[9];
        inv3x3(dfull, inv);
#pragma unroll
        for (int k = 0; k < 9; k++) o.invblk[9 * (size_t)row + k] = inv[k];
      }
    }
  }
};

// one element contribution (i, j) to the block sums: the same three stages in the same order in both kernels
template <typename MT>
__device__ __forceinline__ void add_contribution(double* K, double& m, const double* ci, const double* cj, double V, int i, int j, uint32_t e, const AsmParams& ap,
                                                 const MT* __restrict__ kcorr) {
  const double dij = ci[0] * cj[0] + ci[1] * cj[1] + ci[2] * cj[2];
  const double vl = V * ap.lambda, vm = V * ap.mu;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) K[3 * a + b] += vl * (ci[a] * cj[b]) + vm * (cj[a] * ci[b]);  // parenthesised so that K_ba == K_ab^T bitwise
  K[0] += vm * dij; K[4] += vm * dij; K[8] += vm * dij;
  if (kcorr) {  // warp = 2: the rotation-derivative terms of this element, symmetric part (the exact tangent is symmetric to rounding)
    const MT* C = kcorr + 144 * (size_t)e;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) K[3 * a + b] += 0.5 * ((double)C[12 * (3 * i + a) + 3 * j + b] + (double)C[12 * (3 * j + b) + 3 * i + a]);
  }
  m += ap.rho20 * V * (i == j ? 2.0 : 1.0);
}

template <typename MT>
__global__ __launch_bounds__(kBlock) void k_assemble_rows(SellView sv, const int* __restrict__ slot_coff,
                                                          const int* __restrict__ slot_ccnt, const uint32_t* __restrict__ contrib,
                                                          const MT* __restrict__ rec, const double* __restrict__ fe, AsmOut<MT> o, AsmParams ap,
                                                          const MT* __restrict__ kcorr, int min_width) {
  // min_width > 0: only the slices wider than that (the element-major kernel has done the others)
  const int lane = threadIdx.x & 63;
  for (SliceWalk w(sv.n_slices); w.valid(); w.next()) {
    const int s = w.s;
    const int row = s * 64 + lane;
    const bool rvalid = row < sv.n_owned;
    const int so = sv.slice_off[s], width = sv.slice_off[s + 1] - so;
    if (width <= min_width) continue;
    RowAlgebra<MT> ra;
    ra.begin(o, ap, row, rvalid);
    for (int k = 0; k < width; k++) {
      const int slot = so + k;
      const int col = sv.colidx[(size_t)slot * 64 + lane];
      const int coff = slot_coff[slot], ccnt = slot_ccnt[slot];
      const bool diag = ra.is_diag(row, col, rvalid);
      double K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      double m = 0.0;
      int nc = 0;
      // contributions in groups of 4: the four list words, then the eight record loads, are issued together before any
      // is used, so a lane has 4x the memory parallelism of a one-at-a-time walk; the arithmetic and its order are unchanged
      typedef MT mt4 __attribute__((ext_vector_type(4)));
      constexpr int kG = 4;
      for (int t0 = 0; t0 < ccnt; t0 += kG) {
        uint32_t cw[kG];
#pragma unroll
        for (int u = 0; u < kG; u++) cw[u] = (t0 + u < ccnt) ? contrib[((size_t)coff + t0 + u) * 64 + lane] : 0xFFFFFFFFu;
        mt4 rI[kG], rJ[kG];
#pragma unroll
        for (int u = 0; u < kG; u++) {
          const uint32_t c = cw[u] == 0xFFFFFFFFu ? 0u : cw[u];  // padding reads element 0's record and is discarded below
          const mt4* r = (const mt4*)(rec + 16 * (size_t)(c >> 4));
          rI[u] = r[(c >> 2) & 3];
          rJ[u] = r[c & 3];
        }
#pragma unroll
        for (int u = 0; u < kG; u++) {
          const uint32_t c = cw[u];
          if (c == 0xFFFFFFFFu) continue;
          nc++;
          const uint32_t e = c >> 4;
          const int i = (c >> 2) & 3, j = c & 3;
          const mt4 ri = rI[u], rj = rJ[u];
          const double ci[3] = {(double)ri.x, (double)ri.y, (double)ri.z};
          const double cj[3] = {(double)rj.x, (double)rj.y, (double)rj.z};
          add_contribution<MT>(K, m, ci, cj, (double)ri.w, i, j, e, ap, kcorr);
          if (diag) {
            const double* f = fe + 12 * (size_t)e + 3 * i;
            ra.fi[0] += f[0]; ra.fi[1] += f[1]; ra.fi[2] += f[2];
          }
        }
      }
      RowGather gq;
      gq.load(o, ap, col);
      ra.block(o, ap, slot, lane, gq, rvalid, diag, K, m, nc);
    }
    ra.finish(o, ap, s, lane, row, rvalid);
  }
}

// The slices too wide for the element-major kernels (hub nodes; hull nodes of a Delaunay mesh with 40-60 neighbours), one WORKGROUP per
// slice.  k_assemble_rows gives such a slice to one wavefront, which walks its slots one latency-bound contribution list after the other
// (0.8 ms for a 59-slot slice: the whole assembly of the 606k-tet probe waited for it).  Here the slots are dealt to the workgroup's
// wavefronts: each sums the blocks of its slots (the same operations in the same order as k_assemble_rows), forms what the a7 algebra
// needs of them -- the block's terms of t, the stored values u, the mass entry; element forces and contribution count where the column is
// the row -- and leaves it in a scratch area of HBM ([slot][kWideTerms][64] doubles per workgroup, read back from L2); then wavefront 0
// runs the order-dependent part (the running sums over the slots, the diagonal block) over the finished terms: no gather left in it.
// Sums and their order are k_assemble_rows': same bits (tests/test_fem_gpu.py).
constexpr int kWideTerms = 20;   // per slot and lane: t terms 0..2, qacc terms 3..5, u 6..14, m 15, contribution count 16, element forces 17..19
constexpr int kWideBlock = 512;
template <typename MT>
__global__ __launch_bounds__(kWideBlock) void k_assemble_wide(SellView sv, const int* __restrict__ wide_list, int n_wide, int max_slots, double* __restrict__ scratch,
                                                              const int* __restrict__ slot_coff, const int* __restrict__ slot_ccnt,
                                                              const uint32_t* __restrict__ contrib, const MT* __restrict__ rec, const double* __restrict__ fe,
                                                              AsmOut<MT> o, AsmParams ap, const MT* __restrict__ kcorr) {
  __shared__ int next_slot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sc = scratch + (size_t)blockIdx.x * max_slots * kWideTerms * 64 + lane;
  for (int wi = blockIdx.x; wi < n_wide; wi += gridDim.x) {
    const int s = wide_list[wi];
    const int row = s * 64 + lane;
    const bool rvalid = row < sv.n_owned;
    const int so = sv.slice_off[s], width = min(sv.slice_off[s + 1] - so, max_slots);
    uint8_t ma[3] = {1, 1, 1};
    if (rvalid && ap.apply_mask) { ma[0] = o.dofmask[3 * (size_t)row]; ma[1] = o.dofmask[3 * (size_t)row + 1]; ma[2] = o.dofmask[3 * (size_t)row + 2]; }
    if (threadIdx.x == 0) next_slot = 0;
    __syncthreads();
    // the slots go to whichever wavefront is free (which one sums a block does not touch its bits): the slot of the diagonal blocks has
    // ten times the contributions of the others
    for (;;) {
      int k = 0;
      if (lane == 0) k = atomicAdd(&next_slot, 1);
      k = __builtin_amdgcn_readfirstlane(k);
      if (k >= width) break;
      const int slot = so + k;
      const int col = sv.colidx[(size_t)slot * 64 + lane];
      const int colp = k > 0 ? sv.colidx[(size_t)(slot - 1) * 64 + lane] : -1;
      const int coff = slot_coff[slot], ccnt = slot_ccnt[slot];
      // a slot whose column is the row: the diagonal block if the slot before it has another column, else padding after the row's last
      // block (the rule of RowAlgebra::is_diag: padding repeats the row id behind the blocks, whose columns are distinct)
      const bool diag = rvalid && col == row && colp != row;
      double K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      double m = 0.0, fi[3] = {0.0, 0.0, 0.0};
      int nc = 0;
      typedef MT mt4 __attribute__((ext_vector_type(4)));
      constexpr int kG = 8;  // (k_assemble_rows: 4; one wavefront in eight is on the long list here, and it alone sets the time)
      for (int t0 = 0; t0 < ccnt; t0 += kG) {
        uint32_t cw[kG];
#pragma unroll
        for (int u = 0; u < kG; u++) cw[u] = (t0 + u < ccnt) ? contrib[((size_t)coff + t0 + u) * 64 + lane] : 0xFFFFFFFFu;
        mt4 rI[kG], rJ[kG];
#pragma unroll
        for (int u = 0; u < kG; u++) {
          const uint32_t c = cw[u] == 0xFFFFFFFFu ? 0u : cw[u];
          const mt4* r = (const mt4*)(rec + 16 * (size_t)(c >> 4));
          rI[u] = r[(c >> 2) & 3];
          rJ[u] = r[c & 3];
        }
#pragma unroll
        for (int u = 0; u < kG; u++) {
          const uint32_t c = cw[u];
          if (c == 0xFFFFFFFFu) continue;
          nc++;
          const uint32_t e = c >> 4;
          const int i = (c >> 2) & 3, j = c & 3;
          const mt4 ri = rI[u], rj = rJ[u];
          const double ci[3] = {(double)ri.x, (double)ri.y, (double)ri.z};
          const double cj[3] = {(double)rj.x, (double)rj.y, (double)rj.z};
          add_contribution<MT>(K, m, ci, cj, (double)ri.w, i, j, e, ap, kcorr);
          if (diag) {
            const double* f = fe + 12 * (size_t)e + 3 * i;
            fi[0] += f[0]; fi[1] += f[1]; fi[2] += f[2];
          }
        }
      }
      RowGather g;
      g.load(o, ap, col);
      double* t = sc + (size_t)k * kWideTerms * 64;
      MT* out = o.vals + (size_t)slot * 9 * 64 + lane;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        t[a * 64] = ap.g_k * (K[3 * a] * g.qv[0] + K[3 * a + 1] * g.qv[1] + K[3 * a + 2] * g.qv[2]) + ap.g_m * m * g.qv[a];
        t[(3 + a) * 64] = ap.g_a * m * g.qa[a];
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const MT u = (MT)(ap.s_k * K[3 * a + b] + (a == b ? ap.s_m * m : 0.0));
          t[(6 + 3 * a + b) * 64] = (double)u;
          if (!diag) out[(3 * a + b) * 64] = (ma[a] && g.mb[b]) ? u : (MT)0;
        }
        t[(17 + a) * 64] = fi[a];
      }
      t[15 * 64] = m;
      t[16 * 64] = (double)nc;
    }
    __syncthreads();  // (the terms of every slot are written: block-wide barrier = release + acquire at workgroup scope, and a workgroup is on one CU)
    if (wave == 0) {
      RowAlgebra<MT> ra;
      ra.begin(o, ap, row, rvalid);
      int kdiag = -1;
#pragma unroll 4
      for (int k = 0; k < width; k++) {  // no gather and no branch around a load: the terms of several slots are in flight together
        const int col = sv.colidx[(size_t)(so + k) * 64 + lane];
        const double* t = sc + (size_t)k * kWideTerms * 64;
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = t[q * 64];
        const bool diag = ra.is_diag(row, col, rvalid);
#pragma unroll
        for (int a = 0; a < 3; a++) ra.ta[a] += v[a];
        if (o.qacc) {
#pragma unroll
          for (int a = 0; a < 3; a++) ra.ta[a] += v[3 + a];
        }
        ra.msum += v[15];
        kdiag = diag ? k : kdiag;
#pragma unroll
        for (int q = 0; q < 9; q++) ra.off[q] = diag ? ra.off[q] : ra.off[q] + v[6 + q];
      }
      if (kdiag >= 0) {
        const double* t = sc + (size_t)kdiag * kWideTerms * 64;
        ra.kd = so + kdiag; ra.nd = (int)t[16 * 64];
#pragma unroll
        for (int a = 0; a < 3; a++) ra.fi[a] = t[(17 + a) * 64];
      }
      ra.finish(o, ap, s, lane, row, rvalid);
    }
    __syncthreads();  // (the scratch area and the slot counter are the next slice's)
  }
}

// ------------------------------------------------------------------------------------------------------
// Element-major assembly.  k_assemble_rows walks a row slot by slot, so the 16 contributions of an element reach for its 64-B
// record at 16 different times and every one of them misses L2 (an XCD's records are 8 MB at 1M tets).  Here a row's ELEMENTS
// are walked in ascending order instead -- the incidence list is the contribution list of the diagonal block -- the record is
// read once per element and its four blocks K_i0..K_i3 are added to the accumulators of the slots their columns have in the
// row.  The accumulators (9 stiffness values + the mass entry per slot, fp64) live in LDS as [slot][value][row of the slice]:
// every access is 64 consecutive doubles, conflict-free whatever the slots are.
//
// One workgroup of four wavefronts per slice, lane = row, and the VALUES are dealt to the wavefronts: wavefront a = 0..2 owns
// row a of every 3x3 block (values 3a..3a+2), wavefront 3 owns the mass entry and the element forces.  Every accumulator has
// one owner, so there is no ordering between lanes to keep, and the a7 algebra splits the same way (t_a, the stored row a,
// the row-a sums of the off-diagonal blocks), which puts the fp64 arithmetic on all SIMDs of the CU: with one lane per row and
// block the LDS allows two wavefronts per CU and the same arithmetic took 360 us; with (row, corner) lanes the four lanes of
// a row hit one LDS bank (slot stride = 0 mod 64 banks): 225 us.
// A block receives its contributions in ascending element order through the same operations as in k_assemble_rows, and the
// finished blocks go through the same algebra in slot order: the result is bit for bit k_assemble_rows'.
// LDS = slice width x 5 KB + 2 KB (78.8 KB at 15 slots: two workgroups per CU); the dispatcher falls back to k_assemble_rows when
// a slice is wider than 31 slots (or the widest has a single slot).
//
//   inc_off  [n_slices+1]      first list row of the slice (height = longest incidence list of its 64 rows)
//   inc      [rows][64] u32    element << 2 | corner of the lane's row in it, kNoInc past the end of the list
//   inc_slot [rows][64] u32    4 x u8: the slot (within the slice) of the block (row, node j of the element), j = 0..3
// ------------------------------------------------------------------------------------------------------
constexpr uint32_t kNoInc = 0xFFFFFFFFu;
constexpr int kAsmExtra = 3 + 1;  // LDS rows of 64 doubles after the accumulators: element forces, list lengths.  The row sums of the
// off-diagonal blocks and (block-Jacobi) the diagonal block pass between the wavefronts through the value rows of slots 0 and 1,
// each wavefront using the rows it owns; it zeroes them again before the next slice.

// elements of the rows, wavefront A (row A of the blocks).  Software pipeline over groups of G list rows: the words of group g+2
// and the records of group g+1 are in flight while group g is added up.
template <typename MT, int G, int A, bool TANGENT>
__device__ __forceinline__ void tets_accumulate(double* acc, int lane, int io, int height, const uint32_t* __restrict__ inc, const uint32_t* __restrict__ inc_slot,
                                                const MT* __restrict__ rec, const AsmParams& ap, const MT* __restrict__ kcorr) {
  typedef MT mt4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int b = 0; b < 3; b++) acc[(3 * A + b) * 64 + lane] = acc[640 + (3 * A + b) * 64 + lane] = 0.0;  // (what tets_algebra passed through them)
  uint32_t w0[G], s0[G], w1[G], s1[G];
  mt4 r0[G][4];
  if (height <= 0) return;
  // list rows past the end are read from the last row and turned into padding (no run-time branch around a load)
  auto load_words = [&](int t0, uint32_t* w, uint32_t* sl) {
#pragma unroll
    for (int u = 0; u < G; u++) {
      const size_t at = ((size_t)io + min(t0 + u, height - 1)) * 64 + lane;
      const uint32_t ww = inc[at], ss = inc_slot[at];
      const bool in = t0 + u < height;
      w[u] = in ? ww : kNoInc;
      sl[u] = in ? ss : 0u;
    }
  };
  auto load_records = [&](const uint32_t* w, mt4 (*r)[4]) {
#pragma unroll
    for (int u = 0; u < G; u++) {
      const uint32_t c = w[u] == kNoInc ? 0u : w[u];  // padding reads element 0's record and is discarded below
      const mt4* rp = (const mt4*)(rec + 16 * (size_t)(c >> 2));
#pragma unroll
      for (int k = 0; k < 4; k++) r[u][k] = rp[k];
    }
  };
  load_words(0, w0, s0);
  load_records(w0, r0);
  load_words(G, w1, s1);
  for (int t0 = 0; t0 < height; t0 += G) {
    mt4 r1[G][4];
    uint32_t w2[G], s2[G];
    load_records(w1, r1);
    load_words(t0 + 2 * G, w2, s2);
#pragma unroll
    for (int u = 0; u < G; u++) {
      if (w0[u] == kNoInc) continue;
      const uint32_t e = w0[u] >> 2;
      const int i = (int)(w0[u] & 3);
      const mt4 ri = i == 0 ? r0[u][0] : (i == 1 ? r0[u][1] : (i == 2 ? r0[u][2] : r0[u][3]));
      const double ci[3] = {(double)ri.x, (double)ri.y, (double)ri.z};
      const double V = (double)ri.w;
      const double vl = V * ap.lambda, vm = V * ap.mu;
      // the four blocks of an element sit in four different slots (its nodes are distinct): all twelve accumulators are read
      // before any is written back, one LDS round trip per element instead of four
      double* p[4];
      double K[4][3];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        p[j] = acc + ((s0[u] >> (8 * j)) & 255u) * 640 + (3 * A) * 64 + lane;
#pragma unroll
        for (int b = 0; b < 3; b++) K[j][b] = p[j][b * 64];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const mt4 rj = r0[u][j];
        const double cj[3] = {(double)rj.x, (double)rj.y, (double)rj.z};
        const double dij = ci[0] * cj[0] + ci[1] * cj[1] + ci[2] * cj[2];
#pragma unroll
        for (int b = 0; b < 3; b++) {
          K[j][b] += vl * (ci[A] * cj[b]) + vm * (cj[A] * ci[b]);  // the three stages of add_contribution on value (A, b)
          if (b == A) K[j][b] += vm * dij;
          if (TANGENT) {  // warp = 2 (a template parameter: a run-time test here splits the element's arithmetic into twelve basic blocks)
            const MT* C = kcorr + 144 * (size_t)e;
            K[j][b] += 0.5 * ((double)C[12 * (3 * i + A) + 3 * j + b] + (double)C[12 * (3 * j + b) + 3 * i + A]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int b = 0; b < 3; b++) p[j][b * 64] = K[j][b];
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      w0[u] = w1[u]; s0[u] = s1[u]; w1[u] = w2[u]; s1[u] = s2[u];
#pragma unroll
      for (int k = 0; k < 4; k++) r0[u][k] = r1[u][k];
    }
  }
}

// wavefront 3: the mass entries of the four blocks, the element forces and the length of the list
template <typename MT, int G>
__device__ __forceinline__ void tets_mass_and_forces(double* acc, double* facc, double* nacc, int lane, int io, int height, int zero_slots,
                                                     const uint32_t* __restrict__ inc, const uint32_t* __restrict__ inc_slot, const MT* __restrict__ rec,
                                                     const double* __restrict__ fe, const AsmParams& ap) {
  for (int k = 0; k < zero_slots; k++) acc[k * 640 + 9 * 64 + lane] = 0.0;  // (the algebra wavefronts only read the mass entries)
  if (height <= 0) {  // a slice of nodes no element references
    facc[lane] = facc[64 + lane] = facc[128 + lane] = 0.0;
    nacc[lane] = 0.0;
    return;
  }
  double fi[3] = {0, 0, 0};
  int nd = 0;
  uint32_t w[G], sl[G], w1[G], s1[G];
  double V[G], f[G][3];
  auto load_words = [&](int t0, uint32_t* ww, uint32_t* ss) {
#pragma unroll
    for (int u = 0; u < G; u++) {
      const size_t at = ((size_t)io + min(t0 + u, height - 1)) * 64 + lane;
      const uint32_t a = inc[at], b = inc_slot[at];
      const bool in = t0 + u < height;
      ww[u] = in ? a : kNoInc;
      ss[u] = in ? b : 0u;
    }
  };
  auto load_values = [&](const uint32_t* ww, double* VV, double (*ff)[3]) {
#pragma unroll
    for (int u = 0; u < G; u++) {
      const uint32_t c = ww[u] == kNoInc ? 0u : ww[u];
      VV[u] = (double)rec[16 * (size_t)(c >> 2) + 4 * (c & 3) + 3];
      const double* fp = fe + 12 * (size_t)(c >> 2) + 3 * (c & 3);
      ff[u][0] = fp[0]; ff[u][1] = fp[1]; ff[u][2] = fp[2];
    }
  };
  load_words(0, w, sl);
  load_values(w, V, f);
  load_words(G, w1, s1);
  for (int t0 = 0; t0 < height; t0 += G) {
    double V1[G], f1[G][3];
    uint32_t w2[G], s2[G];
    load_values(w1, V1, f1);
    load_words(t0 + 2 * G, w2, s2);
#pragma unroll
    for (int u = 0; u < G; u++) {
      if (w[u] == kNoInc) continue;
      nd++;
      const int i = (int)(w[u] & 3);
      double* p[4];
      double m[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        p[j] = acc + ((sl[u] >> (8 * j)) & 255u) * 640 + 9 * 64 + lane;
        m[j] = p[j][0];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) p[j][0] = m[j] + ap.rho20 * V[u] * (i == j ? 2.0 : 1.0);
      fi[0] += f[u][0]; fi[1] += f[u][1]; fi[2] += f[u][2];
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      w[u] = w1[u]; sl[u] = s1[u]; w1[u] = w2[u]; s1[u] = s2[u];
      V[u] = V1[u]; f[u][0] = f1[u][0]; f[u][1] = f1[u][1]; f[u][2] = f1[u][2];
    }
  }
  facc[lane] = fi[0]; facc[64 + lane] = fi[1]; facc[128 + lane] = fi[2];
  nacc[lane] = (double)nd;
}

// the a7 algebra of row A of the finished blocks, in slot order (RowAlgebra split by block row); leaves its accumulators zero
// (STRIDE: doubles per slot of the accumulators; MASS_G: the mass entries come from o.mblk_in instead of the tenth value row)
template <typename MT, int A, bool NEWMARK, int STRIDE = 640, bool MASS_G = false>
__device__ __forceinline__ void tets_algebra(double* acc, const double* facc, const double* nacc, const SellView& sv,
                                             const AsmOut<MT>& o, const AsmParams& ap, int s, int lane, int so, int width) {
  const int row = s * 64 + lane;
  const bool rvalid = row < sv.n_owned;
  uint8_t ma[3] = {1, 1, 1};
  if (rvalid && ap.apply_mask) {
    ma[0] = o.dofmask[3 * (size_t)row];
    ma[1] = o.dofmask[3 * (size_t)row + 1];
    ma[2] = o.dofmask[3 * (size_t)row + 2];
  }
  double ta = 0.0, msum = 0.0, off[3] = {0, 0, 0};
  bool seen_diag = false;
  int kd = -1;
  // software pipeline over chunks of kC slots: the columns of chunk c+2 and the gathers (which depend on the columns) of chunk
  // c+1 are in flight while chunk c goes through the algebra
  constexpr int kC = 4;
  int col[kC], col1[kC];
  RowGather gq[kC];
  double mq[kC];  // (MASS_G) the mass entries of the chunk, loaded with its gathers
  // (columns past the width are read from the last slot: no run-time branch around a load; the chunks that lie wholly inside the
  // width go through a copy of the body without any test)
  auto load_cols = [&](int k0, int* cc) {
#pragma unroll
    for (int c = 0; c < kC; c++) cc[c] = sv.colidx[((size_t)so + min(k0 + c, width - 1)) * 64 + lane];
  };
  auto chunk = [&](int k0, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    double Kc[kC][3], mc[kC];
#pragma unroll
    for (int c = 0; c < kC; c++) {
      double* p = acc + min(k0 + c, width - 1) * STRIDE + lane;
#pragma unroll
      for (int b = 0; b < 3; b++) Kc[c][b] = p[(3 * A + b) * 64];
      mc[c] = MASS_G ? mq[c] : p[(MASS_G ? 0 : 9) * 64];
    }
#pragma unroll
    for (int c = 0; c < kC; c++) {
      if (!FULL && k0 + c >= width) break;
      const int k = k0 + c, slot = so + k;
      const bool diag = rvalid && (col[c] == row) && !seen_diag;
      seen_diag = seen_diag || diag;
      double* p = acc + k * STRIDE + lane;
      const double* K = Kc[c];
#pragma unroll
      for (int b = 0; b < 3; b++) p[(3 * A + b) * 64] = 0.0;
      const double m = mc[c];
      const double* qv = gq[c].qv;
      ta += ap.g_k * (K[0] * qv[0] + K[1] * qv[1] + K[2] * qv[2]) + ap.g_m * m * qv[A];
      if (NEWMARK) ta += ap.g_a * m * gq[c].qa[A];
      msum += m;
      if (diag) {
        kd = slot;
      } else {
        MT* out = o.vals + (size_t)slot * 9 * 64 + lane;
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const MT u = (MT)(ap.s_k * K[b] + (A == b ? ap.s_m * m : 0.0));
          off[b] += (double)u;
          out[(3 * A + b) * 64] = (ma[A] && gq[c].mb[b]) ? u : (MT)0;
        }
      }
    }
  };
  if (width > 0) {
    load_cols(0, col);
#pragma unroll
    for (int c = 0; c < kC; c++) {
      gq[c].template load_straight<MT, NEWMARK>(o, ap, col[c]);
      mq[c] = MASS_G ? o.mblk_in[((size_t)so + min(c, width - 1)) * 64 + lane] : 0.0;
    }
    load_cols(kC, col1);
    for (int k0 = 0; k0 < width; k0 += kC) {
      RowGather gq1[kC];
      double mq1[kC];
      int col2[kC];
#pragma unroll
      for (int c = 0; c < kC; c++) {
        gq1[c].template load_straight<MT, NEWMARK>(o, ap, col1[c]);
        mq1[c] = MASS_G ? o.mblk_in[((size_t)so + min(k0 + kC + c, width - 1)) * 64 + lane] : 0.0;
      }
      load_cols(k0 + 2 * kC, col2);
      if (k0 + kC <= width) chunk(k0, std::true_type());
      else chunk(k0, std::false_type());
#pragma unroll
      for (int c = 0; c < kC; c++) { col[c] = col1[c]; col1[c] = col2[c]; gq[c] = gq1[c]; mq[c] = mq1[c]; }
    }
  }
  // the diagonal block needs the sums of the other rows: off[3a+b] and off[3b+a]
  double* xoff = acc;        // value rows of slot 0
  double* dblk = acc + STRIDE;  // value rows of slot 1
#pragma unroll
  for (int b = 0; b < 3; b++) xoff[(3 * A + b) * 64 + lane] = off[b];
  __syncthreads();
  const int nd = (int)nacc[lane];
  double dg = 1.0;
  if (rvalid && kd >= 0) {
    MT* out = o.vals + (size_t)kd * 9 * 64 + lane;
    MT* lo = o.dlo + (size_t)s * 9 * 64 + lane;
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double v = (A == b ? ap.s_m * msum : 0.0) - 0.5 * (xoff[(3 * A + b) * 64 + lane] + xoff[(3 * b + A) * 64 + lane]);  // symmetric part: CG needs A = A^T
      if (!(ma[A] && ma[b])) v = (A == b) ? 1.0 : 0.0;
      if (nd == 0) v = (A == b) ? 1.0 : 0.0;  // a node no element references: identity row, stays at rest
      const MT hi = (MT)v;
      const MT l = (MT)(v - (double)hi);
      out[(3 * A + b) * 64] = hi;
      lo[(3 * A + b) * 64] = l;
      if (o.invblk) dblk[(3 * A + b) * 64 + lane] = (double)hi + (double)l;
      if (A == b) dg = (double)hi + (double)l;
    }
  } else {
    if (kd < 0) {
      MT* lo = o.dlo + (size_t)s * 9 * 64 + lane;  // padding lanes of the last slice
#pragma unroll
      for (int b = 0; b < 3; b++) lo[(3 * A + b) * 64] = (MT)0;
    }
    if (o.invblk) {
#pragma unroll
      for (int b = 0; b < 3; b++) dblk[(3 * A + b) * 64 + lane] = A == b ? 1.0 : 0.0;
    }
  }
  if (rvalid) {
    const size_t d = 3 * (size_t)row + A;
    const double fi = facc[A * 64 + lane];
    if (o.fint_out) o.fint_out[d] = fi;
    if (o.rhs) o.rhs[d] = ma[A] ? ap.rhs_scale * (ta + fi - o.fext[d]) : 0.0;
    if (o.res_all) o.res_all[d] = ap.rhs_scale * (ta + fi - o.fext[d]);
    if (o.invdiag) o.invdiag[d] = 1.0 / dg;
  }
}

// The element-major kernels take slices of up to max_width (<= kIncMaxWidth) slots: the accumulators of a slice live in LDS.  A mesh may
// have a few wider ones (hull nodes of a Delaunay mesh with 40-60 neighbours: 111 of 1,424 slices on the 606k-tet probe) -- those are
// left to the slot-major kernel (k_assemble_rows with min_width = max_width, same bits), which used to take the WHOLE mesh as soon as
// one slice was too wide.  next_narrow: the first slice of the walk s, s + per, ... that is not too wide.
__device__ __forceinline__ int next_narrow(const SellView& sv, int s, int per, int hi, int max_width) {
  while (s < hi && sv.slice_off[s + 1] - sv.slice_off[s] > max_width) s += per;
  return s;
}

template <typename MT, int G, bool TANGENT, bool NEWMARK>
__global__ __launch_bounds__(kBlock) void k_assemble_tets(SellView sv, const int* __restrict__ inc_off, const uint32_t* __restrict__ inc,
                                                          const uint32_t* __restrict__ inc_slot, const MT* __restrict__ rec, const double* __restrict__ fe,
                                                          AsmOut<MT> o, AsmParams ap, const MT* __restrict__ kcorr, int max_width, unsigned long long* __restrict__ prof) {
  extern __shared__ double acc[];  // [slot][10][64], then kAsmExtra rows of 64
  unsigned long long tp[4] = {0, 0, 0, 0}, tc = prof ? wall_clock64() : 0;  // development aid: 100 MHz ticks per phase
  auto lap = [&](int k) { if (prof) { const unsigned long long n = wall_clock64(); tp[k] += n - tc; tc = n; } };
  double* facc = acc + (size_t)max_width * 640;
  double* nacc = facc + 3 * 64;
  const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int v = threadIdx.x; v < (max_width * 10 + kAsmExtra) * 64; v += kBlock) acc[v] = 0.0;
  __syncthreads();
  // XCD-aware walk, one workgroup per slice: workgroup b serves slab b % 8
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
  const int chunk = (sv.n_slices + 7) >> 3;
  const int hi = min((xcd + 1) * chunk, sv.n_slices);
  int prev_width = 0;
  for (int s = next_narrow(sv, xcd * chunk + (blockIdx.x >> 3), per, hi, max_width); s < hi; s = next_narrow(sv, s + per, per, hi, max_width)) {
    const int so = sv.slice_off[s], width = sv.slice_off[s + 1] - so;
    const int io = inc_off[s], height = inc_off[s + 1] - io;
    if (wq == 0) tets_accumulate<MT, G, 0, TANGENT>(acc, lane, io, height, inc, inc_slot, rec, ap, kcorr);
    else if (wq == 1) tets_accumulate<MT, G, 1, TANGENT>(acc, lane, io, height, inc, inc_slot, rec, ap, kcorr);
    else if (wq == 2) tets_accumulate<MT, G, 2, TANGENT>(acc, lane, io, height, inc, inc_slot, rec, ap, kcorr);
    else tets_mass_and_forces<MT, G>(acc, facc, nacc, lane, io, height, prev_width, inc, inc_slot, rec, fe, ap);
    prev_width = width;
    lap(0);
    __syncthreads();
    lap(1);
    if (wq == 0) tets_algebra<MT, 0, NEWMARK>(acc, facc, nacc, sv, o, ap, s, lane, so, width);
    else if (wq == 1) tets_algebra<MT, 1, NEWMARK>(acc, facc, nacc, sv, o, ap, s, lane, so, width);
    else if (wq == 2) tets_algebra<MT, 2, NEWMARK>(acc, facc, nacc, sv, o, ap, s, lane, so, width);
    else __syncthreads();  // (the one inside tets_algebra)
    if (o.invblk) {  // FB_PCG_BLOCK_JACOBI: the inverse of the row's 3x3 diagonal block (symmetric; identity on clamped DOFs)
      __syncthreads();
      const int row = s * 64 + lane;
      if (wq == 0 && row < sv.n_owned) {
        double dfull[9], inv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) dfull[k] = acc[640 + k * 64 + lane];
        inv3x3(dfull, inv);
#pragma unroll
        for (int k = 0; k < 9; k++) o.invblk[9 * (size_t)row + k] = inv[k];
      }
    }
    lap(2);
    __syncthreads();
    lap(3);
  }
  if (prof && lane == 0)
    for (int k = 0; k < 4; k++) atomicAdd(&prof[4 * wq + k], tp[k]);
}

// ------------------------------------------------------------------------------------------------------
// Element-major assembly with ONE fetch of every record per (row, element): k_assemble_tets_st (fp32 records, warp = 1).
// In k_assemble_tets the three value wavefronts each gather the lane's 64-byte record (four 16-byte requests to 64 different cache
// lines per wavefront and list row) and the fourth gathers V and the element forces: the texture-address unit of the CU works
// through ~1,000 line requests per list row and workgroup and is what the element phase waits for, not the arithmetic.  Here the
// FOURTH wavefront alone fetches the records (two list rows ahead, in registers) and stages them in LDS -- one list row per buffer,
// two buffers, 4 KB each, [quarter][lane] so that a 16-byte read per lane is conflict-free -- and the value wavefronts read them
// from there; one barrier per list row hands a buffer over.  The LDS for the staging comes from the MASS entries: they do not
// depend on the state, so k_mass_blocks forms them once per rebuild of the rest data (same sums, same order), the accumulators
// shrink from 10 to 9 value rows per slot (69 KB at 15 slots: still two workgroups per CU, with the 8 KB of staging), and the
// algebra reads them from memory with its gathers.  Same contributions, same operations, same order: bit for bit k_assemble_tets.
// ------------------------------------------------------------------------------------------------------
constexpr int kAsmStride = 576;                  // doubles per slot: 9 value rows of 64
constexpr int kAsmStageDoubles = 2 * 4 * 64 * 2;  // two buffers of 4 quarters x 64 lanes x 16 bytes

template <typename MT>
__global__ __launch_bounds__(kBlock) void k_mass_blocks(SellView sv, const int* __restrict__ inc_off, const uint32_t* __restrict__ inc,
                                                        const uint32_t* __restrict__ inc_slot, const float* __restrict__ volf, double rho20, int max_width,
                                                        double* __restrict__ mblk) {
  extern __shared__ double macc[];  // [wavefront][slot][64]
  const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = blockIdx.x * (kBlock / 64) + wq;
  if (s >= sv.n_slices) return;  // (no barrier in this kernel: a wavefront works on its own slice)
  double* acc = macc + (size_t)wq * max_width * 64;
  const int so = sv.slice_off[s], width = sv.slice_off[s + 1] - so;
  if (width > max_width) return;  // (a slice too wide for the element-major kernels: the slot-major kernel forms its mass entries itself)
  const int io = inc_off[s], height = inc_off[s + 1] - io;
  for (int k = 0; k < width; k++) acc[k * 64 + lane] = 0.0;
  // groups of four list rows: their words, then their volumes, are in flight together (one row at a time is two dependent memory
  // round trips per row: 75 us for the kernel at 1M tets); the sums themselves run in list order
  constexpr int G = 4;
  uint32_t wn[G], sn[G];
#pragma unroll
  for (int u = 0; u < G; u++) {
    const size_t at = ((size_t)io + min(u, max(height - 1, 0))) * 64 + lane;
    wn[u] = height > 0 ? inc[at] : kNoInc; sn[u] = height > 0 ? inc_slot[at] : 0u;
  }
  for (int t0 = 0; t0 < height; t0 += G) {
    uint32_t w[G], sl[G];
    double V[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
      w[u] = t0 + u < height ? wn[u] : kNoInc; sl[u] = sn[u];
      V[u] = (double)volf[(w[u] == kNoInc ? 0u : w[u]) >> 2];  // what k_tet_warp stores in the record
    }
#pragma unroll
    for (int u = 0; u < G; u++) {  // the next group's words
      const size_t at = ((size_t)io + min(t0 + G + u, height - 1)) * 64 + lane;
      wn[u] = inc[at]; sn[u] = inc_slot[at];
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      if (w[u] == kNoInc) continue;
      const int i = (int)(w[u] & 3);
      double* p[4];
      double m[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        p[j] = acc + ((sl[u] >> (8 * j)) & 255u) * 64 + lane;
        m[j] = p[j][0];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) p[j][0] = m[j] + rho20 * V[u] * (i == j ? 2.0 : 1.0);
    }
  }
  for (int k = 0; k < width; k++) mblk[((size_t)so + k) * 64 + lane] = acc[k * 64 + lane];
}

// The hand-over of a staging buffer: LDS operations done, then the workgroup barrier -- WITHOUT the wait for the vector memory loads
// in flight that __syncthreads() implies (its fence covers global memory: every list row would pay a full memory round trip for
// the records and words just requested; measured 390 vs 127 us for the element phase).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// value wavefront A of k_assemble_tets_st: tets_accumulate with the records read from the staging buffers
template <int A>
__device__ __forceinline__ void tets_accumulate_st(double* acc, const float4* stage, int lane, int io, int height, const uint32_t* __restrict__ inc,
                                                   const uint32_t* __restrict__ inc_slot, const AsmParams& ap) {
#pragma unroll
  for (int b = 0; b < 3; b++) acc[(3 * A + b) * 64 + lane] = acc[kAsmStride + (3 * A + b) * 64 + lane] = 0.0;  // (what tets_algebra passed through them)
  if (height <= 0) return;
  auto load_words = [&](int t, uint32_t& w, uint32_t& sl) {  // (rows past the end: read from the last row, turned into padding)
    const size_t at = ((size_t)io + min(t, height - 1)) * 64 + lane;
    const uint32_t ww = inc[at], ss = inc_slot[at];
    w = t < height ? ww : kNoInc;
    sl = t < height ? ss : 0u;
  };
  // words of rows t, t+1, t+2 and (wn, sn) of row t+3, requested one trip earlier: a copy never touches a load of the same trip
  uint32_t w0, s0, w1, s1, w2, s2, wn, sn;
  load_words(0, w0, s0);
  load_words(1, w1, s1);
  load_words(2, wn, sn);
  for (int t = 0; t < height; t++) {
    w2 = wn; s2 = sn;
    load_words(t + 3, wn, sn);
    lds_barrier();  // buffer t & 1 holds the records of list row t
    if (w0 != kNoInc) {
      const float4* st = stage + (t & 1) * 256 + lane;
      const float4 q0 = st[0], q1 = st[64], q2 = st[128], q3 = st[192];  // (named: an array indexed by the corner went to scratch memory)
      const int i = (int)(w0 & 3);
      const float4 ri = i == 0 ? q0 : (i == 1 ? q1 : (i == 2 ? q2 : q3));
      const double ci[3] = {(double)ri.x, (double)ri.y, (double)ri.z};
      const double V = (double)ri.w;
      const double vl = V * ap.lambda, vm = V * ap.mu;
      double* p[4];
      double K[4][3];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        p[j] = acc + ((s0 >> (8 * j)) & 255u) * kAsmStride + (3 * A) * 64 + lane;
#pragma unroll
        for (int b = 0; b < 3; b++) K[j][b] = p[j][b * 64];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 rj = j == 0 ? q0 : (j == 1 ? q1 : (j == 2 ? q2 : q3));  // (j is a constant after unrolling)
        const double cj[3] = {(double)rj.x, (double)rj.y, (double)rj.z};
        const double dij = ci[0] * cj[0] + ci[1] * cj[1] + ci[2] * cj[2];
#pragma unroll
        for (int b = 0; b < 3; b++) {
          K[j][b] += vl * (ci[A] * cj[b]) + vm * (cj[A] * ci[b]);  // the three stages of add_contribution on value (A, b)
          if (b == A) K[j][b] += vm * dij;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int b = 0; b < 3; b++) p[j][b * 64] = K[j][b];
    }
    w0 = w1; s0 = s1; w1 = w2; s1 = s2;
  }
}

// wavefront 3 of k_assemble_tets_st: fetches the records (two list rows ahead) and stages them, sums the element forces, counts the
// list.  Its pipeline registers (every one has a NAME: sets kept in arrays and handed to helpers ended up in scratch memory):
// three sets for the records (rows t+1, t+2, t+3) and the forces (rows t, t+1, t+2), rotated by unrolling the loop three times -- a
// register COPY of a value still in flight would wait for it, and the pipeline would be one list row deep whatever is requested
// ahead -- and the words of rows t .. t+4 in a chain whose newest link was requested one trip earlier.
// (plain local variables of the kernel and macros over them: a struct handed to helper functions ended up in scratch memory as well)
#define FB_ST_DECLARE                                                                  \
  float4 S0a, S0b, S0c, S0d, S1a, S1b, S1c, S1d, S2a, S2b, S2c, S2d;                   \
  double F0x, F0y, F0z, F1x, F1y, F1z, F2x = 0.0, F2y = 0.0, F2z = 0.0;                \
  uint32_t wa = kNoInc, wb = kNoInc, wc = kNoInc, wd = kNoInc, we = kNoInc;            \
  int st_io = 0, st_height = 0;
#define FB_ST_WORD(T) ((T) < st_height ? inc[((size_t)st_io + min((T), st_height - 1)) * 64 + lane] : (inc[((size_t)st_io + st_height - 1) * 64 + lane], kNoInc))
#define FB_ST_RECORD(W, S)                                                                               \
  {                                                                                                      \
    const float4* rp = (const float4*)(rec + 16 * (size_t)(((W) == kNoInc ? 0u : (W)) >> 2));            \
    S##a = rp[0]; S##b = rp[1]; S##c = rp[2]; S##d = rp[3]; /* (padding reads element 0's record) */     \
  }
#define FB_ST_FORCE(W, F)                                              \
  {                                                                    \
    const uint32_t c_ = (W) == kNoInc ? 0u : (W);                      \
    const double* fp = fe + 12 * (size_t)(c_ >> 2) + 3 * (c_ & 3);     \
    F##x = fp[0]; F##y = fp[1]; F##z = fp[2];                          \
  }
#define FB_ST_PUT(BUF, S)                                                \
  {                                                                      \
    float4* st = stage + (BUF) * 256 + lane;                             \
    st[0] = S##a; st[64] = S##b; st[128] = S##c; st[192] = S##d;         \
  }
// the first requests of slice SL: its words, the records of its list rows 0..2, the forces of rows 0..1; row 0 goes to buffer 0.
// Issued while the value wavefronts are in the algebra of the PREVIOUS slice (they do not touch the staging buffers there), so that
// the two dependent round trips (words, then records) are not paid at the start of every slice.
#define FB_ST_BEGIN(SL)                                                                                         \
  {                                                                                                             \
    st_io = inc_off[SL]; st_height = inc_off[(SL) + 1] - st_io;                                                 \
    if (st_height > 0) {                                                                                        \
      wa = FB_ST_WORD(0); wb = FB_ST_WORD(1); wc = FB_ST_WORD(2); wd = FB_ST_WORD(3); we = FB_ST_WORD(4);       \
      FB_ST_RECORD(wa, S0) FB_ST_RECORD(wb, S1) FB_ST_RECORD(wc, S2) FB_ST_FORCE(wa, F0) FB_ST_FORCE(wb, F1)    \
      FB_ST_PUT(0, S0)                                                                                          \
    }                                                                                                           \
  }
// trip t: Sw holds row t+1 (staged now), Sl is free and gets row t+3; Fc holds row t (summed now), Fl gets row t+2
#define FB_ST_TRIP(T, Sw, Sl, Fc, Fl)                                                                                      \
  {                                                                                                                         \
    const uint32_t wf = FB_ST_WORD((T) + 5);                                                                                \
    FB_ST_RECORD(wd, Sl)                                                                                                    \
    FB_ST_FORCE(wc, Fl)                                                                                                     \
    lds_barrier(); /* the value wavefronts have read buffer (T + 1) & 1 (list row T - 1); buffer T & 1 is theirs now */     \
    FB_ST_PUT(((T) + 1) & 1, Sw)                                                                                            \
    if (wa != kNoInc) {                                                                                                     \
      nd++;                                                                                                                 \
      fi[0] += Fc##x; fi[1] += Fc##y; fi[2] += Fc##z;                                                                       \
    }                                                                                                                       \
    wa = wb; wb = wc; wc = wd; wd = we; we = wf;                                                                            \
  }
// the element phase of the staging wavefront for the slice FB_ST_BEGIN prepared
#define FB_ST_RUN                                                                              \
  if (st_height <= 0) { /* a slice of nodes no element references */                           \
    facc[lane] = facc[64 + lane] = facc[128 + lane] = 0.0;                                     \
    nacc[lane] = 0.0;                                                                          \
  } else {                                                                                     \
    double fi[3] = {0, 0, 0};                                                                  \
    int nd = 0, t = 0;                                                                         \
    for (; t + 2 < st_height; t += 3) {                                                        \
      FB_ST_TRIP(t, S1, S0, F0, F2)                                                            \
      FB_ST_TRIP(t + 1, S2, S1, F1, F0)                                                        \
      FB_ST_TRIP(t + 2, S0, S2, F2, F1)                                                        \
    }                                                                                          \
    if (t < st_height) FB_ST_TRIP(t, S1, S0, F0, F2)                                           \
    if (t + 1 < st_height) FB_ST_TRIP(t + 1, S2, S1, F1, F0)                                   \
    facc[lane] = fi[0]; facc[64 + lane] = fi[1]; facc[128 + lane] = fi[2];                     \
    nacc[lane] = (double)nd;                                                                   \
  }

template <bool NEWMARK>
__global__ __launch_bounds__(kBlock) void k_assemble_tets_st(SellView sv, const int* __restrict__ inc_off, const uint32_t* __restrict__ inc,
                                                             const uint32_t* __restrict__ inc_slot, const float* __restrict__ rec, const double* __restrict__ fe,
                                                             AsmOut<float> o, AsmParams ap, int max_width, unsigned long long* __restrict__ prof) {
  extern __shared__ double acc[];  // [slot][9][64], then kAsmExtra rows of 64, then the staging buffers
  unsigned long long tp[4] = {0, 0, 0, 0}, tc = prof ? wall_clock64() : 0;  // development aid: 100 MHz ticks per phase
  auto lap = [&](int k) { if (prof) { const unsigned long long n = wall_clock64(); tp[k] += n - tc; tc = n; } };
  double* facc = acc + (size_t)max_width * kAsmStride;
  double* nacc = facc + 3 * 64;
  float4* stage = (float4*)(nacc + 64);
  const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int v = threadIdx.x; v < (max_width * 9 + kAsmExtra) * 64; v += kBlock) acc[v] = 0.0;
  __syncthreads();
  // XCD-aware walk, one workgroup per slice: workgroup b serves slab b % 8
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
  const int chunk = (sv.n_slices + 7) >> 3;
  const int hi = min((xcd + 1) * chunk, sv.n_slices);
  const int s_first = next_narrow(sv, xcd * chunk + (blockIdx.x >> 3), per, hi, max_width);
  if (wq == 3) {  // the staging wavefront: same barriers as the others, its own work in between
    FB_ST_DECLARE
    if (s_first < hi) FB_ST_BEGIN(s_first)
    for (int s = s_first, nxt; s < hi; s = nxt) {
      nxt = next_narrow(sv, s + per, per, hi, max_width);
      FB_ST_RUN
      lap(0);
      __syncthreads();
      lap(1);
      if (nxt < hi) FB_ST_BEGIN(nxt)  // (the others are in the algebra)
      __syncthreads();  // (the one inside tets_algebra)
      if (o.invblk) __syncthreads();
      lap(2);
      __syncthreads();
      lap(3);
    }
  } else {
    for (int s = s_first; s < hi; s = next_narrow(sv, s + per, per, hi, max_width)) {
      const int so = sv.slice_off[s], width = sv.slice_off[s + 1] - so;
      const int io = inc_off[s], height = inc_off[s + 1] - io;
      if (wq == 0) tets_accumulate_st<0>(acc, stage, lane, io, height, inc, inc_slot, ap);
      else if (wq == 1) tets_accumulate_st<1>(acc, stage, lane, io, height, inc, inc_slot, ap);
      else tets_accumulate_st<2>(acc, stage, lane, io, height, inc, inc_slot, ap);
      lap(0);
      __syncthreads();
      lap(1);
      if (wq == 0) tets_algebra<float, 0, NEWMARK, kAsmStride, true>(acc, facc, nacc, sv, o, ap, s, lane, so, width);
      else if (wq == 1) tets_algebra<float, 1, NEWMARK, kAsmStride, true>(acc, facc, nacc, sv, o, ap, s, lane, so, width);
      else tets_algebra<float, 2, NEWMARK, kAsmStride, true>(acc, facc, nacc, sv, o, ap, s, lane, so, width);
      if (o.invblk) {  // FB_PCG_BLOCK_JACOBI: the inverse of the row's 3x3 diagonal block (symmetric; identity on clamped DOFs)
        __syncthreads();
        const int row = s * 64 + lane;
        if (wq == 0 && row < sv.n_owned) {
          double dfull[9], inv[9];
#pragma unroll
          for (int k = 0; k < 9; k++) dfull[k] = acc[kAsmStride + k * 64 + lane];
          inv3x3(dfull, inv);
#pragma unroll
          for (int k = 0; k < 9; k++) o.invblk[9 * (size_t)row + k] = inv[k];
        }
      }
      lap(2);
      __syncthreads();
      lap(3);
    }
  }
  if (prof && lane == 0)
    for (int k = 0; k < 4; k++) atomicAdd(&prof[4 * wq + k], tp[k]);
}

// ------------------------------------------------------------------------------------------------------
// a9 SpMV on the SELL-64 3x3-block matrix (SparseMatrix::MultiplyVector, sparseMatrix.cpp:405-413), fused
// with the reduction its caller needs:
//   MODE 0: y = A x                                    (plain)
//   MODE 1: q = A d,  partial[b] = sum d.q             (PCG direction product, CGSolver.cpp:149-150)
//   MODE 2: r = b - A x, partial[b] = sum r^2 invdiag   (exact residual every 30th iteration, CGSolver.cpp:161-171)
//   MODE 3: q = A d with the three sums of the merged-reduction iteration: partial[b] = sum d.q,
//           partial[G+b] = sum invdiag r q, partial[2G+b] = sum invdiag q^2 (bvec carries r)
// ------------------------------------------------------------------------------------------------------
// XCH = 2 (sharded handles on the peer-to-peer transport, MODE 3 only): the halo refresh of x rides in the prologue.  The
// first blocks each store one chunk of this rank's boundary values into a neighbour's inbox and release that chunk's
// flag; every block then waits for the neighbours' chunks and gathers halo columns straight from the own inbox.  No launch of
// its own, no grid-wide ticket (same-address atomics cost ~12 ns each here) and no per-block fence (an agent-scope
// release writes back the whole per-XCD L2): x was completed by the previous kernel.
// NT: the value stream is loaded non-temporally.  On a system whose per-iteration working set (matrix + vector passes)
// exceeds the 256 MiB Infinity Cache the values can never be re-used from it, and loading them without allocating keeps
// the vectors resident instead: 168 vs 206 us at 8M tets (5.9 vs 4.8 TB/s).  On a system that fits (1M tets, 160 MB)
// the same hint evicts the matrix from the cache it would be served from: 24.7 vs 20.8 us -- so the host picks by size
// (crossover measured between 341 MB and 469 MB per iteration; FEMBRAIN_SPMV_NT=0/1 overrides).
// C16: column ids are read as 16-bit differences to the row (2 instead of 4 bytes per block; unsharded handles whose
// plan was built on the device, when every difference fits) -- same columns, same products, 4 % fewer bytes.
template <typename MT, int MODE, int XCH = 0, bool NT = false, int C16 = 0>
__global__ __launch_bounds__(kBlock) void k_spmv(SellView sv, const MT* __restrict__ vals, const MT* __restrict__ dlo, const double* __restrict__ x,
                                                 double* __restrict__ y, const double* __restrict__ bvec,
                                                 const double* __restrict__ invdiag, double* __restrict__ partial,
                                                 CGState* __restrict__ st, int parity, P2PArgs pa) {
  __shared__ double lds[4];
  if (MODE != 0 && st->done) return;
  if (MODE == 1 || MODE == 3) {
    // the while-condition of CGSolver.cpp:147, evaluated by every block from the same published scalars at the
    // head of each iteration; `done` is sticky so that every later launch of the batch is a no-op
    if (!(st->rho[parity] > st->eps2 * st->rho0) || st->iter >= st->max_iter) {
      if (blockIdx.x == 0 && threadIdx.x == 0) st->done = 1;
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  double acc = 0.0, acc1 = 0.0, acc2 = 0.0;
  const double* halo_in = nullptr;
  // one SELL slice (wave-uniform s): row products, diagonal low part, epilogue of the MODE
  auto do_slice = [&](const int s) {
    const int row = s * 64 + lane;
    const int so = sv.slice_off[s], width = sv.slice_off[s + 1] - so;
    const MT* v = vals + (size_t)so * 9 * 64 + lane;
    const int* ci = sv.colidx + (size_t)so * 64 + lane;
    const short* cd = C16 ? sv.coldelta + (size_t)so * 64 + lane : nullptr;
    const int hcol = C16 == 2 ? sv.n_owned + sv.halo_base[s] : 0;
    double y0 = 0, y1 = 0, y2 = 0;
#pragma unroll 2  // measured at 1M tets: 27.4 us per iteration with 2, 27.7 with 1, 28.1 with 3, 28.25 with 4, 41 with 8 (registers)
    for (int k = 0; k < width; k++) {
      int col;
      if (C16 == 2) {  // sharded handle: bit 0 of the word says whether it counts from the row or from the slice's first halo column
        const int wd = (int)cd[(size_t)k * 64];
        col = (wd & 1) ? hcol + (wd >> 1) : row + (wd >> 1);
      } else {
        col = C16 ? row + (int)cd[(size_t)k * 64] : ci[(size_t)k * 64];
      }
      const double* xp = ((XCH == 2 && col >= sv.n_owned) ? halo_in : x) + 3 * (size_t)col;
      const double x0 = xp[0], x1 = xp[1], x2 = xp[2];
      const MT* vk = v + (size_t)k * 9 * 64;
      auto lv = [&](int j) { return NT ? __builtin_nontemporal_load(vk + j * 64) : vk[j * 64]; };
      y0 += (double)lv(0) * x0 + (double)lv(1) * x1 + (double)lv(2) * x2;
      y1 += (double)lv(3) * x0 + (double)lv(4) * x1 + (double)lv(5) * x2;
      y2 += (double)lv(6) * x0 + (double)lv(7) * x1 + (double)lv(8) * x2;
    }
    if (row < sv.n_owned) {
      const size_t d = 3 * (size_t)row;
      {  // low part of the diagonal block (see k_assemble_rows)
        const MT* l = dlo + (size_t)s * 9 * 64 + lane;
        const double o0 = x[d], o1 = x[d + 1], o2 = x[d + 2];
        // the low part is symmetric (k_assemble_rows forms it from the symmetric part of the block): 6 of its 9 planes are read
        const double l01 = (double)l[1 * 64], l02 = (double)l[2 * 64], l12 = (double)l[5 * 64];
        y0 += (double)l[0 * 64] * o0 + l01 * o1 + l02 * o2;
        y1 += l01 * o0 + (double)l[4 * 64] * o1 + l12 * o2;
        y2 += l02 * o0 + l12 * o1 + (double)l[8 * 64] * o2;
      }
      if (MODE == 0) {
        y[d] = y0; y[d + 1] = y1; y[d + 2] = y2;
      } else if (MODE == 1) {
        y[d] = y0; y[d + 1] = y1; y[d + 2] = y2;
        acc += x[d] * y0 + x[d + 1] * y1 + x[d + 2] * y2;
      } else if (MODE == 3) {
        y[d] = y0; y[d + 1] = y1; y[d + 2] = y2;
        acc += x[d] * y0 + x[d + 1] * y1 + x[d + 2] * y2;
        const double i0 = invdiag[d], i1 = invdiag[d + 1], i2 = invdiag[d + 2];
        acc1 += i0 * bvec[d] * y0 + i1 * bvec[d + 1] * y1 + i2 * bvec[d + 2] * y2;
        acc2 += i0 * y0 * y0 + i1 * y1 * y1 + i2 * y2 * y2;
      } else {
        const double r0 = bvec[d] - y0, r1 = bvec[d + 1] - y1, r2 = bvec[d + 2] - y2;
        y[d] = r0; y[d + 1] = r1; y[d + 2] = r2;
        acc += r0 * r0 * invdiag[d] + r1 * r1 * invdiag[d + 1] + r2 * r2 * invdiag[d + 2];
      }
    }
  };
  if (XCH != 2) {
    for (SliceWalk w(sv.n_slices); w.valid(); w.next()) do_slice(w.s);
  } else {
    // The halo refresh rides here.  Sender jobs first; then two passes over this block's slices: the interior ones (no halo
    // column, the bulk) before the wait for the neighbours' values, the boundary ones after it; a block without boundary
    // slices never waits at all.
    p2p_send_halo_jobs(pa.dev, pa.halo_seq, 3, pa.send_ids, pa.send_off, x);
    for (SliceWalk w(sv.n_slices); w.valid(); w.next())
      if (!pa.slice_halo[w.s]) do_slice(w.s);
    bool mine = false;  // does any slice of this block touch the halo?  (block-uniform: every wave scans the block's slices)
    for (int j = blockIdx.x >> 3, chunk = (sv.n_slices + 7) >> 3, lo = (blockIdx.x & 7) * chunk, hi = min(lo + chunk, sv.n_slices),
             s0 = lo + j * kWavesPerBlock; s0 < hi; s0 += (gridDim.x >> 3) * kWavesPerBlock)
      for (int k = 0; k < kWavesPerBlock && s0 + k < hi; k++) mine = mine || pa.slice_halo[s0 + k];
    if (mine) {
      p2p_wait_halo(pa.dev, pa.halo_seq, pa.halo_off, 3);
      halo_in = p2p_halo_in(pa.dev, pa.halo_seq) - 3 * (size_t)sv.n_owned;
      for (SliceWalk w(sv.n_slices); w.valid(); w.next())
        if (pa.slice_halo[w.s]) do_slice(w.s);
    }
  }
  if (MODE != 0) {
    const double tot = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
  }
  if (MODE == 3) {
    const double t1 = block_sum(acc1, lds);
    const double t2 = block_sum(acc2, lds);
    if (threadIdx.x == 0) {
      partial[gridDim.x + blockIdx.x] = t1;
      partial[2 * gridDim.x + blockIdx.x] = t2;
    }
  }
  (void)acc1; (void)acc2;
}

// Small and mid-size meshes: SPLIT wavefronts share one slice (SPLIT = 4: one slice per block, up to 1024 slices ~ 350k
// tets; SPLIT = 2: two slices per block, up to 2048 slices ~ 750k tets).  With fewer slices than the chip has wave slots
// the row kernel above is bound by the latency of one wavefront walking all ~15 slots of its slice (6.8 us at 105k tets
// for 13 MB); dealing the slots to several wavefronts shortens that chain.  The partial row sums are added in wave order
// through LDS (deterministic; rounding differs from the row kernel in the last bit).
// XCH = 2 (sharded handles, FB_XCH_P2P_FUSED): as in k_spmv the halo refresh rides in the prologue -- the sender jobs are dealt
// to the first blocks, a block whose slice has a halo column waits for the neighbours' chunks and gathers those columns
// from the inbox; blocks of interior slices never wait.
template <typename MT, int MODE, int SPLIT, int XCH = 0>
__global__ __launch_bounds__(kBlock) void k_spmv_split(SellView sv, const MT* __restrict__ vals, const MT* __restrict__ dlo,
                                                       const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ bvec,
                                                       const double* __restrict__ invdiag, double* __restrict__ partial,
                                                       CGState* __restrict__ st, int parity, P2PArgs pa) {
  constexpr int kPerBlock = kWavesPerBlock / SPLIT;  // slices per block
  __shared__ double ylds[kWavesPerBlock][3][64];
  __shared__ double red[3][kWavesPerBlock];
  if (MODE != 0 && st->done) return;
  if (MODE == 1 || MODE == 3) {
    if (!(st->rho[parity] > st->eps2 * st->rho0) || st->iter >= st->max_iter) {
      if (blockIdx.x == 0 && threadIdx.x == 0) st->done = 1;
      return;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = wv % SPLIT, which = wv / SPLIT;
  const int xcd = blockIdx.x & 7, chunk = (sv.n_slices + 7) >> 3;
  const int j = (int)(blockIdx.x >> 3) * kPerBlock + which;
  const int s = xcd * chunk + j;
  const bool live = j < chunk && s < sv.n_slices;  // wave-uniform
  double y0 = 0, y1 = 0, y2 = 0;
  const double* halo_in = nullptr;
  if (XCH == 2) {
    p2p_send_halo_jobs(pa.dev, pa.halo_seq, 3, pa.send_ids, pa.send_off, x);
    bool mine = false;  // block-uniform: does any slice of this block touch the halo?
    for (int k = 0; k < kPerBlock; k++) {
      const int jj = (int)(blockIdx.x >> 3) * kPerBlock + k, ss = xcd * chunk + jj;
      if (jj < chunk && ss < sv.n_slices) mine = mine || pa.slice_halo[ss];
    }
    if (mine) {
      p2p_wait_halo(pa.dev, pa.halo_seq, pa.halo_off, 3);
      halo_in = p2p_halo_in(pa.dev, pa.halo_seq) - 3 * (size_t)sv.n_owned;
    }
  }
  if (live) {
    const int so = sv.slice_off[s], width = sv.slice_off[s + 1] - so;
    const MT* v = vals + (size_t)so * 9 * 64 + lane;
    const int* ci = sv.colidx + (size_t)so * 64 + lane;
#pragma unroll 2  // (4.40 vs 4.69 us per launch at 105k tets, 8.0 vs 8.5 at 329k)
    for (int k = sub; k < width; k += SPLIT) {
      const int col = ci[(size_t)k * 64];
      const double* xp = ((XCH == 2 && col >= sv.n_owned) ? halo_in : x) + 3 * (size_t)col;
      const double x0 = xp[0], x1 = xp[1], x2 = xp[2];
      const MT* vk = v + (size_t)k * 9 * 64;
      y0 += (double)vk[0 * 64] * x0 + (double)vk[1 * 64] * x1 + (double)vk[2 * 64] * x2;
      y1 += (double)vk[3 * 64] * x0 + (double)vk[4 * 64] * x1 + (double)vk[5 * 64] * x2;
      y2 += (double)vk[6 * 64] * x0 + (double)vk[7 * 64] * x1 + (double)vk[8 * 64] * x2;
    }
  }
  ylds[wv][0][lane] = y0; ylds[wv][1][lane] = y1; ylds[wv][2][lane] = y2;
  __syncthreads();
  double acc = 0.0, acc1 = 0.0, acc2 = 0.0;
  const int row = s * 64 + lane;
  if (sub == 0 && live && row < sv.n_owned) {  // the slice's first wavefront finishes the rows
    y0 = ylds[wv][0][lane]; y1 = ylds[wv][1][lane]; y2 = ylds[wv][2][lane];
#pragma unroll
    for (int k = 1; k < SPLIT; k++) { y0 += ylds[wv + k][0][lane]; y1 += ylds[wv + k][1][lane]; y2 += ylds[wv + k][2][lane]; }
    const size_t d = 3 * (size_t)row;
    {
      const MT* l = dlo + (size_t)s * 9 * 64 + lane;
      const double o0 = x[d], o1 = x[d + 1], o2 = x[d + 2];
      const double l01 = (double)l[1 * 64], l02 = (double)l[2 * 64], l12 = (double)l[5 * 64];  // symmetric, see k_spmv
      y0 += (double)l[0 * 64] * o0 + l01 * o1 + l02 * o2;
      y1 += l01 * o0 + (double)l[4 * 64] * o1 + l12 * o2;
      y2 += l02 * o0 + l12 * o1 + (double)l[8 * 64] * o2;
    }
    if (MODE == 0) {
      y[d] = y0; y[d + 1] = y1; y[d + 2] = y2;
    } else if (MODE == 1) {
      y[d] = y0; y[d + 1] = y1; y[d + 2] = y2;
      acc += x[d] * y0 + x[d + 1] * y1 + x[d + 2] * y2;
    } else if (MODE == 3) {
      y[d] = y0; y[d + 1] = y1; y[d + 2] = y2;
      acc += x[d] * y0 + x[d + 1] * y1 + x[d + 2] * y2;
      const double i0 = invdiag[d], i1 = invdiag[d + 1], i2 = invdiag[d + 2];
      acc1 += i0 * bvec[d] * y0 + i1 * bvec[d + 1] * y1 + i2 * bvec[d + 2] * y2;
      acc2 += i0 * y0 * y0 + i1 * y1 * y1 + i2 * y2 * y2;
    } else {
      const double r0 = bvec[d] - y0, r1 = bvec[d + 1] - y1, r2 = bvec[d + 2] - y2;
      y[d] = r0; y[d + 1] = r1; y[d + 2] = r2;
      acc += r0 * r0 * invdiag[d] + r1 * r1 * invdiag[d + 1] + r2 * r2 * invdiag[d + 2];
    }
  }
  if (MODE != 0) {  // one partial per block: the finishing wavefronts' sums in wave order
    const double t0 = wave_sum(acc), t1 = wave_sum(acc1), t2 = wave_sum(acc2);
    if (lane == 0) { red[0][wv] = t0; red[1][wv] = t1; red[2][wv] = t2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0, b = 0.0, c = 0.0;
      for (int k = 0; k < kWavesPerBlock; k += SPLIT) { a += red[0][k]; b += red[1][k]; c += red[2][k]; }
      partial[blockIdx.x] = a;
      if (MODE == 3) { partial[gridDim.x + blockIdx.x] = b; partial[2 * gridDim.x + blockIdx.x] = c; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// PCG vector kernels (CGSolver.cpp:129-190).  Same lane<->row map and XCD slabs as the SpMV so each XCD's L2
// keeps its share of x, r, d, q, invdiag between kernels.  `sc` (when non-null) holds an already
// all-reduced scalar {dq, rho_new} (multi-GPU); otherwise the per-block partials are summed in fixed order.
// ------------------------------------------------------------------------------------------------------
// x = 0, r = b, d = invdiag r, partial = sum r^2 invdiag
__global__ __launch_bounds__(kBlock) void k_cg_init(int n_slices, int n_owned, const double* __restrict__ b,
                                                    const double* __restrict__ invdiag, double* __restrict__ x,
                                                    double* __restrict__ r, double* __restrict__ d, double* __restrict__ partial) {
  __shared__ double lds[4];
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const size_t i = 3 * (size_t)row + a;
        const double ri = b[i], di = invdiag[i];
        x[i] = 0.0; r[i] = ri; d[i] = di * ri;
        acc += ri * ri * di;
      }
    }
  }
  const double tot = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// warm start (ImplicitNewmarkSparse leaves the previous solution in the solver's start vector): r = b - A x has been formed by
// k_spmv<MT,2> (which also left sum r^2 invdiag in the partials); here d = invdiag r
__global__ __launch_bounds__(kBlock) void k_cg_init_warm(int n_slices, int n_owned, const double* __restrict__ r,
                                                         const double* __restrict__ invdiag, double* __restrict__ d) {
  const int lane = threadIdx.x & 63;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const size_t i = 3 * (size_t)row + a;
        d[i] = invdiag[i] * r[i];
      }
    }
  }
}

// Newmark (implicitNewmarkSparse.cpp:190-201, :356-364): q += dq (dq may be null: the predictor), then
// qaccel = a1 (q - q_1) - a2 qvel_1 - a3 qaccel_1,  qvel = a4 (q - q_1) + a5 qvel_1 + a6 qaccel_1;  constrained DOFs -> 0
struct NewmarkAlphas { double a1, a2, a3, a4, a5, a6; };
__global__ __launch_bounds__(kBlock) void k_newmark_update(int n, const double* __restrict__ dq, const uint8_t* __restrict__ mask, NewmarkAlphas al,
                                                           const double* __restrict__ q1, const double* __restrict__ v1, const double* __restrict__ a1,
                                                           double* __restrict__ q, double* __restrict__ qvel, double* __restrict__ qacc) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (dq && !mask[i]) { q[i] = 0.0; qvel[i] = 0.0; qacc[i] = 0.0; return; }
  const double qn = dq ? q[i] + dq[i] : q[i];
  q[i] = qn;
  qacc[i] = al.a1 * (qn - q1[i]) - al.a2 * v1[i] - al.a3 * a1[i];
  qvel[i] = al.a4 * (qn - q1[i]) + al.a5 * v1[i] + al.a6 * a1[i];
}

// sum of squares of n doubles in one block, fixed order (Newton error test of the Newmark step)
__global__ __launch_bounds__(kBlock) void k_sumsq(size_t n, const double* __restrict__ v, double* __restrict__ out) {
  __shared__ double lds[4];
  double a = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kBlock) a += v[i] * v[i];
  const double t = block_sum(a, lds);
  if (threadIdx.x == 0) out[0] = t;
}

// ---- FB_PCG_BLOCK_JACOBI (opt-in, not part of the reference): the literal PCG of CGSolver.cpp:129-190 with z = B^-1 r, B the
// 3x3 diagonal blocks, in place of the Jacobi scaling r / diag ----
__device__ __forceinline__ void blk_apply(const double* __restrict__ B, const double* r, double* z) {
  z[0] = B[0] * r[0] + B[1] * r[1] + B[2] * r[2];
  z[1] = B[3] * r[0] + B[4] * r[1] + B[5] * r[2];
  z[2] = B[6] * r[0] + B[7] * r[1] + B[8] * r[2];
}
// x = 0, r = b, d = B^-1 r, partial = sum r . B^-1 r
__global__ __launch_bounds__(kBlock) void k_bj_init(int n_slices, int n_owned, const double* __restrict__ b, const double* __restrict__ invblk,
                                                    double* __restrict__ x, double* __restrict__ r, double* __restrict__ d, double* __restrict__ partial) {
  __shared__ double lds[4];
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
      const size_t i = 3 * (size_t)row;
      const double rr[3] = {b[i], b[i + 1], b[i + 2]};
      double z[3];
      blk_apply(invblk + 9 * (size_t)row, rr, z);
#pragma unroll
      for (int a = 0; a < 3; a++) { x[i + a] = 0.0; r[i + a] = rr[a]; d[i + a] = z[a]; acc += rr[a] * z[a]; }
    }
  }
  const double tot = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// warm start (x kept, r = b - A x left by the SpMV): d = B^-1 r, partial = sum r . B^-1 r
__global__ __launch_bounds__(kBlock) void k_bj_init_warm(int n_slices, int n_owned, const double* __restrict__ r, const double* __restrict__ invblk,
                                                         double* __restrict__ d, double* __restrict__ partial) {
  __shared__ double lds[4];
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
      const size_t i = 3 * (size_t)row;
      const double rr[3] = {r[i], r[i + 1], r[i + 2]};
      double z[3];
      blk_apply(invblk + 9 * (size_t)row, rr, z);
#pragma unroll
      for (int a = 0; a < 3; a++) { d[i + a] = z[a]; acc += rr[a] * z[a]; }
    }
  }
  const double tot = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// alpha = rho / (d.q); x += alpha d; REFRESH ? nothing more : (r -= alpha q, partial = sum r . B^-1 r)
template <bool REFRESH>
__global__ __launch_bounds__(kBlock) void k_bj_update(int n_slices, int n_owned, const CGState* __restrict__ st, int parity, const double* __restrict__ part_dq,
                                                      int n_partial, const double* __restrict__ d, const double* __restrict__ q,
                                                      const double* __restrict__ invblk, double* __restrict__ x, double* __restrict__ r,
                                                      double* __restrict__ part_rho) {
  __shared__ double lds[4];
  if (st->done) return;
  const double dq = sum_partials(part_dq, n_partial, lds);
  const double alpha = st->rho[parity] / dq;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
      const size_t i = 3 * (size_t)row;
      double rr[3], z[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        x[i + a] += alpha * d[i + a];
        if (!REFRESH) { rr[a] = r[i + a] - alpha * q[i + a]; r[i + a] = rr[a]; }
      }
      if (!REFRESH) {
        blk_apply(invblk + 9 * (size_t)row, rr, z);
        acc += rr[0] * z[0] + rr[1] * z[1] + rr[2] * z[2];
      }
    }
  }
  if (!REFRESH) {
    const double tot = block_sum(acc, lds);
    if (threadIdx.x == 0) part_rho[blockIdx.x] = tot;
  }
}
// partial = sum r . B^-1 r of the exact residual the SpMV left in r
__global__ __launch_bounds__(kBlock) void k_bj_rho(int n_slices, int n_owned, const CGState* __restrict__ st, const double* __restrict__ r,
                                                   const double* __restrict__ invblk, double* __restrict__ part_rho) {
  __shared__ double lds[4];
  if (st->done) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
      const size_t i = 3 * (size_t)row;
      const double rr[3] = {r[i], r[i + 1], r[i + 2]};
      double z[3];
      blk_apply(invblk + 9 * (size_t)row, rr, z);
      acc += rr[0] * z[0] + rr[1] * z[1] + rr[2] * z[2];
    }
  }
  const double tot = block_sum(acc, lds);
  if (threadIdx.x == 0) part_rho[blockIdx.x] = tot;
}
// beta = rho_new / rho; d = B^-1 r + beta d; block 0 publishes rho_new and the iteration count
__global__ __launch_bounds__(kBlock) void k_bj_direction(int n_slices, int n_owned, CGState* st, int parity, const double* __restrict__ part_rho, int n_partial,
                                                         const double* __restrict__ r, const double* __restrict__ invblk, double* __restrict__ d) {
  __shared__ double lds[4];
  if (st->done) return;
  const double rho_new = sum_partials(part_rho, n_partial, lds);
  const double beta = rho_new / st->rho[parity];
  const int lane = threadIdx.x & 63;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
      const size_t i = 3 * (size_t)row;
      const double rr[3] = {r[i], r[i + 1], r[i + 2]};
      double z[3];
      blk_apply(invblk + 9 * (size_t)row, rr, z);
#pragma unroll
      for (int a = 0; a < 3; a++) d[i + a] = z[a] + beta * d[i + a];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->rho[1 - parity] = rho_new;
    st->iter = st->iter + 1;
  }
}

// one block: rho0 from the partials (or the all-reduced scalar), initial state
__global__ __launch_bounds__(kBlock) void k_cg_begin(CGState* st, const double* partial, int n_partial, const double* sc,
                                                     double eps, int max_iter) {
  __shared__ double lds[4];
  const double rho0 = sc ? sc[0] : sum_partials(partial, n_partial, lds);
  if (threadIdx.x == 0) {
    st->rho[0] = rho0; st->rho[1] = rho0; st->rho0 = rho0; st->eps2 = eps * eps;
    st->iter = 0; st->max_iter = max_iter; st->done = 0; st->pad = 0;
  }
}

// alpha = rho / (d.q);  x += alpha d;  REFRESH ? nothing more : (r -= alpha q, partial = sum r^2 invdiag)
template <bool REFRESH>
__global__ __launch_bounds__(kBlock) void k_cg_update(int n_slices, int n_owned, const CGState* __restrict__ st, int parity,
                                                      const double* __restrict__ part_dq, int n_partial, const double* sc,
                                                      const double* __restrict__ d, const double* __restrict__ q,
                                                      const double* __restrict__ invdiag, double* __restrict__ x,
                                                      double* __restrict__ r, double* __restrict__ part_rho) {
  __shared__ double lds[4];
  if (st->done) return;
  const double dq = sc ? sc[0] : sum_partials(part_dq, n_partial, lds);
  const double alpha = st->rho[parity] / dq;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const size_t i = 3 * (size_t)row + a;
        x[i] += alpha * d[i];
        if (!REFRESH) {
          const double ri = r[i] - alpha * q[i];
          r[i] = ri;
          acc += ri * ri * invdiag[i];
        }
      }
    }
  }
  if (!REFRESH) {
    const double tot = block_sum(acc, lds);
    if (threadIdx.x == 0) part_rho[blockIdx.x] = tot;
  }
}

// beta = rho_new / rho;  d = invdiag r + beta d;  block 0 publishes rho_new, iteration count and the exit test
__global__ __launch_bounds__(kBlock) void k_cg_direction(int n_slices, int n_owned, CGState* st, int parity,
                                                         const double* __restrict__ part_rho, int n_partial, const double* sc,
                                                         const double* __restrict__ r, const double* __restrict__ invdiag,
                                                         double* __restrict__ d) {
  __shared__ double lds[4];
  if (st->done) return;
  const double rho_new = sc ? sc[0] : sum_partials(part_rho, n_partial, lds);
  const double beta = rho_new / st->rho[parity];
  const int lane = threadIdx.x & 63;
  for (SliceWalk w(n_slices); w.valid(); w.next()) {
    const int row = w.s * 64 + lane;
    if (row < n_owned) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const size_t i = 3 * (size_t)row + a;
        d[i] = invdiag[i] * r[i] + beta * d[i];
      }
    }
  }
  // single writer; rho[1-parity] and iter are read only by later launches (this launch reads done, rho[parity])
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->rho[1 - parity] = rho_new;
    st->iter = st->iter + 1;
  }
}

// Merged-reduction iteration, vector half: with S0 = d.q, S1 = sum invdiag r q, S2 = sum invdiag q^2 from the SpMV launch,
//   alpha = rho / S0,   rho_new = sum invdiag (r - alpha q)^2 = rho - 2 alpha S1 + alpha^2 S2,   beta = rho_new / rho
// so x, r and the next direction are updated in ONE pass with no second reduction (same iterates as CGSolver.cpp:149-182
// up to rounding; the exact residual and its exactly summed rho are restored every 30th iteration by the reference path).
// Flat element walk for the vector kernels: each XCD keeps the SAME contiguous slab of rows it owns in the SpMV
// (chunk slices * 192 doubles), walked in 16-byte double2 pieces (192 doubles per slice keep every slab 16-B aligned).
struct PairWalk {
  size_t i, hi, stride;
  __device__ PairWalk(int n_slices, int n_owned) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int chunk = (n_slices + 7) >> 3;
    const size_t lo = (size_t)xcd * chunk * 96;  // in double2 units: 64 rows * 3 / 2
    hi = min((size_t)(xcd + 1) * chunk * 96, ((size_t)3 * n_owned + 1) / 2);
    i = lo + (size_t)j * kBlock + threadIdx.x;
    stride = (size_t)per * kBlock;
  }
  __device__ bool valid() const { return i < hi; }
  __device__ void next() { i += stride; }
};

// WAIT (sharded handles on the peer-to-peer transport): the global sum rides in the prologue.  Block 0 folds this rank's
// per-block partials (complete: the SpMV kernel has finished) and posts the three sums into every rank's inbox; every
// block then waits for all ranks' posts and adds them in rank order.  The first operand loads are already in flight.
template <bool WAIT = false>
__global__ __launch_bounds__(kBlock) void k_cg_fused(int n_slices, int n_owned, CGState* st, int parity,
                                                     const double* __restrict__ part, int n_partial, const double* sc,
                                                     const double* __restrict__ q, const double* __restrict__ invdiag,
                                                     double* __restrict__ x, double* __restrict__ r, double* __restrict__ d, P2PArgs pa) {
  __shared__ double lds[12];
  if (st->done) return;
  const size_t n3 = 3 * (size_t)n_owned;  // an odd count ends in a half pair: loads stay in bounds (vectors carry 2 spare
                                          // doubles, see upload_plan), only the owned component is stored
  PairWalk w(n_slices, n_owned);
  // first piece of own operands is requested before the scalar prologue so its latency overlaps the partial sums
  double2 d0 = make_double2(0, 0), r0 = d0, q0 = d0, i0 = d0, x0 = d0;
  const bool first = w.valid();
  if (first) {
    d0 = ((const double2*)d)[w.i]; r0 = ((const double2*)r)[w.i]; q0 = ((const double2*)q)[w.i];
    i0 = ((const double2*)invdiag)[w.i]; x0 = ((const double2*)x)[w.i];
  }
  double s0, s1, s2;
  if (WAIT) {
    __shared__ double mine[8];
    if (blockIdx.x == 0) p2p_post_sums(pa.dev, pa.seq, part, n_partial, 3, lds, mine);
    p2p_wait_sums(pa.dev, pa.seq, 3, lds);
    s0 = lds[0]; s1 = lds[1]; s2 = lds[2];
  } else if (sc) {
    s0 = sc[0]; s1 = sc[1]; s2 = sc[2];
  } else {
    double a0 = 0, a1 = 0, a2 = 0;
    for (int k = threadIdx.x; k < n_partial; k += kBlock) { a0 += part[k]; a1 += part[n_partial + k]; a2 += part[2 * n_partial + k]; }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { lds[wv] = a0; lds[4 + wv] = a1; lds[8 + wv] = a2; }
    __syncthreads();
    s0 = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    s1 = (lds[4] + lds[5]) + (lds[6] + lds[7]);
    s2 = (lds[8] + lds[9]) + (lds[10] + lds[11]);
  }
  const double rho = st->rho[parity];
  const double alpha = rho / s0;
  const double rho_new = fmax(rho - 2.0 * alpha * s1 + alpha * alpha * s2, 0.0);
  const double beta = rho_new / rho;
  if (first) {
    for (;;) {
      double2 xn, rn, dn;
      xn.x = x0.x + alpha * d0.x; xn.y = x0.y + alpha * d0.y;
      rn.x = r0.x - alpha * q0.x; rn.y = r0.y - alpha * q0.y;
      dn.x = i0.x * rn.x + beta * d0.x; dn.y = i0.y * rn.y + beta * d0.y;
      if (2 * w.i + 1 < n3) {
        ((double2*)x)[w.i] = xn; ((double2*)r)[w.i] = rn; ((double2*)d)[w.i] = dn;
      } else {
        x[2 * w.i] = xn.x; r[2 * w.i] = rn.x; d[2 * w.i] = dn.x;
      }
      w.next();
      if (!w.valid()) break;
      d0 = ((const double2*)d)[w.i]; r0 = ((const double2*)r)[w.i]; q0 = ((const double2*)q)[w.i];
      i0 = ((const double2*)invdiag)[w.i]; x0 = ((const double2*)x)[w.i];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->rho[1 - parity] = rho_new;
    st->iter = st->iter + 1;
  }
}

// state update of PS_VolumeConservingIntegrator.cpp:229-237: qvel += dv, q += h qvel, constrained -> 0
__global__ __launch_bounds__(kBlock) void k_state_update(int n, const double* __restrict__ dv, const uint8_t* __restrict__ mask,
                                                         double h, double* __restrict__ q, double* __restrict__ qvel) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (mask[i]) {
    const double v = qvel[i] + dv[i];
    qvel[i] = v;
    q[i] += h * v;
  } else {
    q[i] = 0.0;
    qvel[i] = 0.0;
  }
}

__global__ __launch_bounds__(kBlock) void k_fill_axis(int n_nodes, int axis, double value, double* __restrict__ f) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_nodes) return;
  f[3 * (size_t)i] = axis == 0 ? value : 0.0;
  f[3 * (size_t)i + 1] = axis == 1 ? value : 0.0;
  f[3 * (size_t)i + 2] = axis == 2 ? value : 0.0;
}

__global__ __launch_bounds__(kBlock) void k_axpy(int n, double a, const double* __restrict__ x, double* __restrict__ y) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

// floor collision of Deformable::timestep (Deformable.cpp:350-400)
__global__ __launch_bounds__(kBlock) void k_floor(int n_nodes, const double* __restrict__ x0, double floor_y, double rest,
                                                  double* __restrict__ q, double* __restrict__ qvel, int* __restrict__ n_collided) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  int hit = 0;
  if (i < n_nodes) {
    const size_t d = 3 * (size_t)i;
    const double pry = x0[d + 1];
    const double pcy = pry + q[d + 1];
    // v = (0,1,0): vn = (0, vy, 0); vr = (v - vn) - rest * vn
    qvel[d + 1] = -rest * qvel[d + 1];
    if (pcy <= floor_y) {
      hit = 1;
      q[d + 1] = floor_y - pry;
    }
  }
  const unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_collided, __popcll(b));
}

// ------------------------------------------------------------------------------------------------------
// a1 batched element stiffness K0 = V B^T (E B) on the matrix cores: one wavefront per element, the 12x12
// product padded into one 16x16 tile, K = 6 strain rows padded to 8 = two v_mfma_f64_16x16x4_f64 steps
// (A = B^T: 12x6, B-operand = E B: 6x12).  fp64 MFMA keeps this inspection path within 1e-12 of the oracle.
// Operand lane maps (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D: col = lane&15, row = (lane>>4) + 4*reg.
// ------------------------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ inline double strainB(const double* b, int s, int col) {
  // B (6 x 12) entry: strain row s, dof col = 3*v + d, from the gradients b[v][*]  (corotationalLinearFEM.cpp:107-114)
  const int v = col / 3, d = col - 3 * v;
  const double bx = b[3 * v], by = b[3 * v + 1], bz = b[3 * v + 2];
  switch (s) {
    case 0: return d == 0 ? bx : 0.0;
    case 1: return d == 1 ? by : 0.0;
    case 2: return d == 2 ? bz : 0.0;
    case 3: return d == 0 ? by : (d == 1 ? bx : 0.0);
    case 4: return d == 1 ? bz : (d == 2 ? by : 0.0);
    default: return d == 0 ? bz : (d == 2 ? bx : 0.0);
  }
}

__global__ __launch_bounds__(kBlock) void k_element_K0_mfma(int first, int count, const double* __restrict__ rest,
                                                            double lambda, double mu, double* __restrict__ K0,
                                                            double* __restrict__ Minv, const double* __restrict__ x0,
                                                            const int4* __restrict__ tets) {
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (wave >= count) return;  // wave-uniform
  const int e = first + wave;
  const int lane = threadIdx.x & 63;
  const double* r = rest + 16 * (size_t)e;
  const double V = r[12];
  const int ij = lane & 15, kq = lane >> 4;  // kq = k index inside a 4-deep step
  v4d acc = {0, 0, 0, 0};
#pragma unroll
  for (int step = 0; step < 2; step++) {
    const int s = 4 * step + kq;  // strain row 0..7 (6,7 are padding)
    double a = 0.0, bb = 0.0;
    if (s < 6 && ij < 12) {
      a = strainB(r, s, ij);  // A[i][k] = B^T[i][s] = B[s][i]
      // (E B)[s][j] = sum_t E[s][t] B[t][j]
      if (s < 3) {
        bb = lambda * (strainB(r, 0, ij) + strainB(r, 1, ij) + strainB(r, 2, ij)) + 2.0 * mu * strainB(r, s, ij);
      } else {
        bb = mu * strainB(r, s, ij);
      }
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
  }
  if (ij < 12) {
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int row = kq + 4 * reg;
      if (row < 12) K0[144 * (size_t)wave + 12 * row + ij] = acc[reg] * V;
    }
  }
  if (Minv && lane < 16) {
    // row k = lane>>2: [b_k | N_k(0)], N_k(0) = delta_k0 - b_k . x_0
    const int k = lane >> 2, d = lane & 3;
    double val;
    if (d < 3) val = r[3 * k + d];
    else {
      const int n0 = tets[e].x;
      val = (k == 0 ? 1.0 : 0.0) - (r[3 * k] * x0[3 * (size_t)n0] + r[3 * k + 1] * x0[3 * (size_t)n0 + 1] + r[3 * k + 2] * x0[3 * (size_t)n0 + 2]);
    }
    Minv[16 * (size_t)wave + lane] = val;
  }
}

}  // namespace fb
