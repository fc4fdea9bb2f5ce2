// Persistent Jacobi-PCG iterations (included by fem.hip only): a run of merged-reduction iterations (CGSolver.cpp:149-182 in
// the one-reduction form of k_cg_fused) inside ONE launch, for systems whose vectors fit on the chip.
//
// Why: with a launch per phase an iteration moves, besides the matrix, 46 MB of vector traffic at 1M tets (the SpMV epilogue
// reads own d, r, 1/diag; the vector pass reads q, r, d, 1/diag, x and writes x, r, d) and pays two kernel boundaries.  Here one
// workgroup per CU owns a fixed run of SELL slices, one wavefront per slice, one lane per block row, and every lane keeps its
// row's x, r, d and 1/diag in REGISTERS for the whole run: the vector pass disappears, only the new search direction d is
// published (4.2 MB) because the neighbours' products gather it.  Per iteration:
//   A  q = A d for the own row (gathers of d: plain loads from global), the three merged sums S0 = d.q, S1 = sum r q / diag,
//      S2 = sum q^2 / diag: wave -> workgroup (LDS, fixed order) -> posted as 8-byte {half of the double, sequence number}
//      granules, each one sc1 store (an aligned 8-byte store is the unit that is never torn: MI355X_MICROARCH.md, persistent
//      kernels)
//   1  every workgroup sweeps all workgroups' granules (sc1 loads) until they carry this iteration's sequence number and adds
//      them in workgroup order -- the same total, bitwise, everywhere: alpha, rho', beta need no further exchange
//   B  x += alpha d, r -= alpha q, d = r / diag + beta d in registers; d stored sc1 (write-through), the stores drained
//      (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane posts the workgroup's flag = sequence number
//   2  one wave polls all flags (sc1), then an agent-scope acquire (invalidates this CU's L1: d lines the gathers cached),
//      s_waitcnt vmcnt(0), workgroup barrier -> next iteration's gathers are plain loads
// (producer: sc1 payload -> drained -> sc1 flag; consumer: poll -> agent acquire -> wait -> barrier -> plain loads: the valid
// hand-off form of the guide's table.)  Hazards: a workgroup overwrites d only after sweep 1 of the same iteration, i.e. after
// every workgroup has finished the gathers of that iteration; sums are double-buffered by sequence parity (a workgroup can be
// at most one sweep ahead of the slowest).  Every spin is bounded by the wall clock; on expiry an error word is set, every
// other spin sees it, the kernel drains and the host reports FB_EDEVICE.
//
// The arithmetic per row is that of k_spmv<MT,3> + k_cg_fused; the sums are grouped by workgroup instead of by 256-thread
// block, so iterates agree with FB_PCG_MERGED to rounding (not bitwise) and with themselves bitwise however the run is cut
// into launches (tests/test_fem_gpu.py::test_persistent_pcg_*).  The exact-residual iteration (every 30th) stays with the
// stand-alone kernels; the run ends there and at convergence, writing x and r back.
#pragma once
#include "fem_kernels.h"
#include "pcg_pipe.hip.h"

namespace fb {

constexpr int kPersistMaxWaves = 16;   // wavefronts (= slices) per workgroup
constexpr int kPersistMaxBlocks = 512;

struct PersistArgs {
  unsigned long long* post;   // [2][n_blocks][8] granules: (hi, lo) of S0, S1, S2, each | sequence << 32
  unsigned int* flags;        // [n_blocks (padded to 4)]: sequence number of the last published d
  unsigned int* error;        // set on a timed-out wait
  unsigned int seq_base;      // sequence number before the first iteration of this launch
  int first_iter;             // 1-based number of the first iteration of this launch
  int n_iters;
  long long timeout_ticks;    // wall_clock64 ticks (100 MHz)
  long long* timing;          // development aid (FEMBRAIN_PERSIST_TIMING=1), else null: per workgroup 5 accumulated phase times
  double* dsoa;               // the search direction as the gathers read it: three planes x | y | z of n_pad doubles each
  size_t n_pad;               // rows padded to whole slices
  int lds_slots;              // slots of every slice kept in LDS for the whole launch (resident part of the matrix)
};
constexpr int kPersistSyncDoubles = 3 * kPersistMaxWaves + 3 * kPersistMaxBlocks + 4;  // LDS in front of the resident values

// d (node-major xyz, what every other kernel reads) -> the three planes the persistent kernel gathers from.  A gather of one
// component for 64 consecutive columns is then one contiguous 512-byte request (4 cache lines) instead of 64 pieces 24 bytes
// apart (12 lines): the vector L1 serves lines, not bytes.
__global__ __launch_bounds__(kBlock) void k_persist_planes(int n_owned, size_t n_pad, const double* __restrict__ d, double* __restrict__ dsoa) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pad) return;
  const bool in = i < (size_t)n_owned;
  dsoa[i] = in ? d[3 * i] : 0.0;
  dsoa[n_pad + i] = in ? d[3 * i + 1] : 0.0;
  dsoa[2 * n_pad + i] = in ? d[3 * i + 2] : 0.0;
}



// slices of workgroup b: XCD b & 7 keeps the contiguous eighth of the rows it has in every other FEM kernel (SliceWalk); its
// gridDim/8 workgroups share that slab as evenly as whole slices allow
__device__ __forceinline__ void persist_slices(int n_slices, int* first, int* count) {
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int chunk = (n_slices + 7) >> 3;
  const int lo = xcd * chunk, len = max(0, min(chunk, n_slices - lo));
  const int base = len / per, rem = len - base * per;
  *first = lo + j * base + min(j, rem);
  *count = base + (j < rem ? 1 : 0);
}

// KR: the first KR slots of the slice (9 values + the column id per lane and slot) are loaded ONCE per launch and stay in
// registers for all its iterations -- the matrix does not change between PCG iterations, and the register files of the chip
// (512 KB per CU, 128 MB in all) are larger than the 102 MB matrix of the 1M-tet mesh: what is resident is never streamed
// again.  Slots beyond KR are streamed from memory every iteration as in k_spmv.  WMAX: wavefronts per workgroup the
// instantiation is bounded for (the register budget per lane is 512 / ceil(WMAX / 4)).
// amdgpu_waves_per_eu: one workgroup of WMAX wavefronts per CU is all that will ever be resident, i.e. ceil(WMAX / 4) per SIMD;
// without it the compiler aims at more wavefronts per SIMD than can exist here, caps itself at 128 registers and serialises
// the value loads of the streaming loop through one register (measured: phase A 38 instead of 13 us).
template <typename MT, bool C16, int KR, int WMAX, int KLT = 0>
__global__ __launch_bounds__(64 * WMAX) void k_pcg_persist(SellView sv, const MT* __restrict__ vals, const MT* __restrict__ dlo,
                                                                       const double* __restrict__ invdiag, double* __restrict__ x,
                                                                       double* __restrict__ r, double* d, CGState* __restrict__ st, PersistArgs pa) {
  extern __shared__ double lds[];  // [3 * waves] wave sums | [3 * gridDim] gathered sums | [4] broadcast; the request is padded so that one workgroup fills a CU
  const int n_waves = blockDim.x >> 6, nb = gridDim.x;
  double* wsum = lds;
  double* gath = lds + 3 * kPersistMaxWaves;
  double* bc = gath + 3 * kPersistMaxBlocks;
  if (st->done) return;  // grid-uniform: written by an earlier launch
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int first, count;
  persist_slices(sv.n_slices, &first, &count);
  const bool live = wv < count;  // wave-uniform
  const int s = first + wv;
  const int row = s * 64 + lane;
  const bool rvalid = live && row < sv.n_owned;
  const size_t dof = 3 * (size_t)(rvalid ? row : 0);
  // this row's vectors stay here for the whole run
  double xr[3] = {0, 0, 0}, rr[3] = {0, 0, 0}, dr[3] = {0, 0, 0}, iv[3] = {0, 0, 0};
  if (rvalid) {
#pragma unroll
    for (int a = 0; a < 3; a++) { xr[a] = x[dof + a]; rr[a] = r[dof + a]; dr[a] = d[dof + a]; iv[a] = invdiag[dof + a]; }
  }
  int so = 0, width = 0;
  if (live) { so = sv.slice_off[s]; width = sv.slice_off[s + 1] - so; }
  const MT* v = vals + (size_t)so * 9 * 64 + lane;
  const int* ci = sv.colidx + (size_t)so * 64 + lane;
  const short* cd = C16 ? sv.coldelta + (size_t)so * 64 + lane : nullptr;
  // low part of the diagonal block (symmetric: 6 planes), fixed for the run
  MT m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;
  if (rvalid) {
    const MT* l = dlo + (size_t)s * 9 * 64 + lane;
    m00 = l[0 * 64]; m01 = l[1 * 64]; m02 = l[2 * 64]; m11 = l[4 * 64]; m12 = l[5 * 64]; m22 = l[8 * 64];
  }
  // resident part of the matrix
  MT pv[KR > 0 ? KR : 1][9];
  int pc[KR > 0 ? KR : 1];
#pragma unroll
  for (int k = 0; k < KR; k++) {
    pc[k] = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) pv[k][j] = (MT)0;
    if (k < width) {
      pc[k] = C16 ? row + (int)cd[(size_t)k * 64] : ci[(size_t)k * 64];
      const MT* vk = v + (size_t)k * 9 * 64;
#pragma unroll
      for (int j = 0; j < 9; j++) pv[k][j] = vk[j * 64];
    }
  }
  // LDS-resident part: the first KL slots of this wave's slice, [KL][10][64] words (9 values + the column id), loaded once
  const int KL = min(KR + KLT, width);  // slots [KR, KR + KLT) live in LDS; KLT == pa.lds_slots (the host picks the instantiation)
  unsigned int* lres = (unsigned int*)(lds + kPersistSyncDoubles) + (size_t)wv * KLT * 10 * 64 + lane;
  for (int k = KR; k < KL; k++) {
    const MT* vk = v + (size_t)k * 9 * 64;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      if (sizeof(MT) == 4) lres[((k - KR) * 10 + j) * 64] = __float_as_uint((float)vk[j * 64]);
    }
    lres[((k - KR) * 10 + 9) * 64] = (unsigned int)(C16 ? row + (int)cd[(size_t)k * 64] : ci[(size_t)k * 64]);
  }
  double rho = st->rho[(pa.first_iter - 1) & 1];
  const double rho0 = st->rho0, eps2 = st->eps2;
  const int max_iter = st->max_iter;
  int iter = st->iter;
  bool done = false, failed = false;
  const long long t_limit = pa.timeout_ticks;
  int it = 0;
  long long tm[5] = {0, 0, 0, 0, 0}, tprev = pa.timing ? wall_clock64() : 0;
  auto lap = [&](int k) { if (pa.timing) { const long long t = wall_clock64(); tm[k] += t - tprev; tprev = t; } };
  for (; it < pa.n_iters; it++) {
    // the while-condition of CGSolver.cpp:147 at the head of the iteration; rho is bitwise the same in every workgroup
    if (!(rho > eps2 * rho0) || iter >= max_iter) { done = true; break; }
    const unsigned int seq = pa.seq_base + (unsigned int)it + 1u;
    // ---- A: q = A d ----
    double y0 = 0, y1 = 0, y2 = 0;
    if (live) {
      // resident slots, two at a time: the scheduling barrier keeps the compiler from hoisting ALL gathers to the front
      // (8 registers per slot in flight), which would not leave room for the resident values
#pragma unroll
      for (int k0 = 0; k0 < KR; k0 += 2) {
#pragma unroll
        for (int k = k0; k < k0 + 2 && k < KR; k++) {
          if (k < width) {
            const double* xp = pa.dsoa + (size_t)pc[k];
            const double x0 = xp[0], x1 = xp[pa.n_pad], x2 = xp[2 * pa.n_pad];
            // the resident values stay in their storage type: without the (empty) asm the compiler hoists the fp32 -> fp64
            // conversions out of the iteration loop and keeps 18 instead of 9 registers per slot
            MT t[9];
#pragma unroll
            for (int j = 0; j < 9; j++) { t[j] = pv[k][j]; asm volatile("" : "+v"(t[j])); }
            y0 += (double)t[0] * x0 + (double)t[1] * x1 + (double)t[2] * x2;
            y1 += (double)t[3] * x0 + (double)t[4] * x1 + (double)t[5] * x2;
            y2 += (double)t[6] * x0 + (double)t[7] * x1 + (double)t[8] * x2;
          }
        }
      }
      if (sizeof(MT) == 4) {
#pragma unroll
        for (int k = KR; k < KR + KLT; k++) if (k < KL) {  // LDS-resident slots
          const unsigned int* lk = lres + (size_t)(k - KR) * 10 * 64;
          const double* xp = pa.dsoa + (size_t)lk[9 * 64];
          const double x0 = xp[0], x1 = xp[pa.n_pad], x2 = xp[2 * pa.n_pad];
          y0 += (double)__uint_as_float(lk[0 * 64]) * x0 + (double)__uint_as_float(lk[1 * 64]) * x1 + (double)__uint_as_float(lk[2 * 64]) * x2;
          y1 += (double)__uint_as_float(lk[3 * 64]) * x0 + (double)__uint_as_float(lk[4 * 64]) * x1 + (double)__uint_as_float(lk[5 * 64]) * x2;
          y2 += (double)__uint_as_float(lk[6 * 64]) * x0 + (double)__uint_as_float(lk[7 * 64]) * x1 + (double)__uint_as_float(lk[8 * 64]) * x2;
        }
      }
#pragma unroll 2
      for (int k = (sizeof(MT) == 4 ? KR + KLT : KR); k < width; k++) {
        const int col = C16 ? row + (int)cd[(size_t)k * 64] : ci[(size_t)k * 64];
        const double* xp = pa.dsoa + (size_t)col;
        const double x0 = xp[0], x1 = xp[pa.n_pad], x2 = xp[2 * pa.n_pad];
        const MT* vk = v + (size_t)k * 9 * 64;
        y0 += (double)vk[0 * 64] * x0 + (double)vk[1 * 64] * x1 + (double)vk[2 * 64] * x2;
        y1 += (double)vk[3 * 64] * x0 + (double)vk[4 * 64] * x1 + (double)vk[5 * 64] * x2;
        y2 += (double)vk[6 * 64] * x0 + (double)vk[7 * 64] * x1 + (double)vk[8 * 64] * x2;
      }
    }
    double a0 = 0, a1 = 0, a2 = 0;
    if (rvalid) {
      MT u00 = m00, u01 = m01, u02 = m02, u11 = m11, u12 = m12, u22 = m22;  // (kept in storage type, see the resident slots)
      asm volatile("" : "+v"(u00), "+v"(u01), "+v"(u02), "+v"(u11), "+v"(u12), "+v"(u22));
      const double l00 = (double)u00, l01 = (double)u01, l02 = (double)u02, l11 = (double)u11, l12 = (double)u12, l22 = (double)u22;
      y0 += l00 * dr[0] + l01 * dr[1] + l02 * dr[2];
      y1 += l01 * dr[0] + l11 * dr[1] + l12 * dr[2];
      y2 += l02 * dr[0] + l12 * dr[1] + l22 * dr[2];
      a0 = dr[0] * y0 + dr[1] * y1 + dr[2] * y2;
      a1 = iv[0] * rr[0] * y0 + iv[1] * rr[1] * y1 + iv[2] * rr[2] * y2;
      a2 = iv[0] * y0 * y0 + iv[1] * y1 * y1 + iv[2] * y2 * y2;
    }
    lap(0);  // A: products
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { wsum[wv] = a0; wsum[kPersistMaxWaves + wv] = a1; wsum[2 * kPersistMaxWaves + wv] = a2; }
    __syncthreads();
    // Every wave of this workgroup has finished its gathers of d: drop the d lines this CU's L1 holds NOW (agent-scope
    // acquire = buffer_inv sc1, asynchronous) instead of after the wait for the neighbours' new d -- nothing loads d until
    // the next phase A, so the L1 cannot pick stale lines up again, and the invalidate runs behind the sweep.
    if (wv == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned long long* post = pa.post + ((size_t)(seq & 1u) * nb) * 8;
    if (threadIdx.x < 3) {  // workgroup sums in wave order, posted as two tagged halves each
      double t = 0.0;
      for (int w = 0; w < n_waves; w++) t += wsum[threadIdx.x * kPersistMaxWaves + w];
      const unsigned long long bits = (unsigned long long)__double_as_longlong(t), tag = (unsigned long long)seq << 32;
      st_sc1_u64(post + (size_t)blockIdx.x * 8 + 2 * threadIdx.x, (bits >> 32) | tag);
      st_sc1_u64(post + (size_t)blockIdx.x * 8 + 2 * threadIdx.x + 1, (bits & 0xffffffffULL) | tag);
    }
    // ---- 1: all workgroups' sums ----
    {
      const int pollers = min(n_waves, 4);
      if (wv < pollers) {
        const long long t0 = wall_clock64();
        for (int b = wv * 64 + lane; b - lane < nb; b += pollers * 64) {  // wave-uniform trip count
          const bool mine = b < nb;
          unsigned long long g[6] = {0, 0, 0, 0, 0, 0};
          for (;;) {
            bool ok = true;
            if (mine) {  // the record's six granules in three 16-byte requests (each granule is still one 8-byte store of its writer)
              uint4 q4[3];
              asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                           : "=&v"(q4[0]), "=&v"(q4[1]), "=&v"(q4[2]) : "v"(post + (size_t)b * 8) : "memory");
#pragma unroll
              for (int k = 0; k < 3; k++) {
                g[2 * k] = (unsigned long long)q4[k].x | ((unsigned long long)q4[k].y << 32);
                g[2 * k + 1] = (unsigned long long)q4[k].z | ((unsigned long long)q4[k].w << 32);
              }
#pragma unroll
              for (int k = 0; k < 6; k++) ok = ok && (unsigned int)(g[k] >> 32) == seq;
            }
            if (__ballot(!ok) == 0ULL) break;
            if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (failed) break;
          if (mine) {
#pragma unroll
            for (int k = 0; k < 3; k++)
              gath[k * kPersistMaxBlocks + b] = __longlong_as_double((long long)(((g[2 * k] & 0xffffffffULL) << 32) | (g[2 * k + 1] & 0xffffffffULL)));
          }
        }
        if (failed && lane == 0) st_sc1_u32(pa.error, 1u);
      }
      __syncthreads();
      if (wv == 0) {  // fixed order: lane l adds workgroups l, l + 64, ...; then the wave tree
        double t0s = 0, t1s = 0, t2s = 0;
        for (int b = lane; b < nb; b += 64) { t0s += gath[b]; t1s += gath[kPersistMaxBlocks + b]; t2s += gath[2 * kPersistMaxBlocks + b]; }
        t0s = wave_sum(t0s); t1s = wave_sum(t1s); t2s = wave_sum(t2s);
        // one lane reads the error word for the whole workgroup: the decision to leave must be workgroup-uniform
        if (lane == 0) { bc[0] = t0s; bc[1] = t1s; bc[2] = t2s; bc[3] = ld_sc1_u32(pa.error) != 0u ? 1.0 : 0.0; }
      }
      __syncthreads();
    }
    lap(1);  // post + sweep 1
    if (bc[3] != 0.0) { failed = true; break; }  // a wait timed out somewhere: every workgroup leaves within one phase
    const double s0 = bc[0], s1 = bc[1], s2 = bc[2];
    const double alpha = rho / s0;
    const double rho_new = fmax(rho - 2.0 * alpha * s1 + alpha * alpha * s2, 0.0);
    const double beta = rho_new / rho;
    // ---- B: vector update in registers, publish d ----
    if (rvalid) {
      const double q[3] = {y0, y1, y2};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        xr[a] = xr[a] + alpha * dr[a];
        rr[a] = rr[a] - alpha * q[a];
        dr[a] = iv[a] * rr[a] + beta * dr[a];
      }
#pragma unroll
      for (int a = 0; a < 3; a++) st_sc1_f64(pa.dsoa + a * pa.n_pad + (size_t)row, dr[a]);  // 512 contiguous bytes per wave and plane
    }
    rho = rho_new;
    iter++;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lap(2);  // B: update + publish d, drained
    if (threadIdx.x == 0) st_sc1_u32(pa.flags + blockIdx.x, seq);
    // ---- 2: every workgroup's d is out ----
    if (wv == 0) {
      const long long t0 = wall_clock64();
      for (int b = 4 * lane; b - 4 * lane < nb; b += 256) {  // four flags per lane in one 16-byte request
        for (;;) {
          bool ok = true;
          if (b < nb) {
            const uint4 f = ld_sc1_u128(pa.flags + b);
            ok = (int)(f.x - seq) >= 0 && (b + 1 >= nb || (int)(f.y - seq) >= 0) && (b + 2 >= nb || (int)(f.z - seq) >= 0) && (b + 3 >= nb || (int)(f.w - seq) >= 0);
          }
          if (__ballot(!ok) == 0ULL) break;
          if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (failed) break;
      }
      if (failed && lane == 0) st_sc1_u32(pa.error, 1u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) bc[3] = (failed || ld_sc1_u32(pa.error) != 0u) ? 1.0 : 0.0;
    }
    __syncthreads();
    lap(3);  // sweep 2 + acquire
    if (bc[3] != 0.0) { failed = true; break; }
  }
  (void)failed;
  if (pa.timing && lane == 0) {
    for (int k = 0; k < 4; k++) atomicAdd((unsigned long long*)pa.timing + ((size_t)blockIdx.x * kPersistMaxWaves + wv) * 5 + k, (unsigned long long)tm[k]);
    atomicAdd((unsigned long long*)pa.timing + ((size_t)blockIdx.x * kPersistMaxWaves + wv) * 5 + 4, (unsigned long long)it);
  }
  // write the run's result back for the stand-alone kernels (exact-residual iteration, state update)
  if (rvalid) {
#pragma unroll
    for (int a = 0; a < 3; a++) { x[dof + a] = xr[a]; r[dof + a] = rr[a]; d[dof + a] = dr[a]; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int next_it = pa.first_iter + it;  // the iteration that would run next
    st->rho[(next_it - 1) & 1] = rho;
    st->iter = iter;
    if (done) st->done = 1;
  }
}

}  // namespace fb
