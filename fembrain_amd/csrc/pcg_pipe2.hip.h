// k_pcg_pipe2: the persistent pipelined Jacobi-PCG of pcg_pipe.hip.h for 13..24 slices per CU (1.2M..2.2M tets on 256 CUs), where
// the registers of a CU no longer hold the state of one wavefront per slice.  Same recurrences, same hand-offs, same arithmetic
// per row; what differs:
//   * a wavefront owns up to TWO slices (lane = two rows, `h` = 0 / 1): slices first + wv and first + wv + (slice wavefronts) of its
//     workgroup -- a workgroup with 13 slices keeps twelve wavefronts busy, one of them with two slices (round 5; it used to give
//     wavefront wv the slices 2 wv and 2 wv + 1: six and a half busy wavefronts, and two wide slices of a cut on one of them);
//   * r, w, s and 1/diag of both rows stay in registers (48 per lane); x, p, z and the low part of the diagonal block live in
//     LDS (24 words per row: 147 KB for 12 wavefronts) -- they are touched once per iteration, in the recurrences;
//   * the vectors of a workgroup's slices take 6 KB each; what they leave of the CU's 160 KB holds the first slots of every slice, as in
//     k_pcg_pipe (round 5: 30 wavefront-slots at 13 slices per CU, 2 at 24; rounds 3-4 reserved 24 vector areas whatever the count and
//     streamed every slot).
// So an iteration moves nearly the whole matrix (111 MB per million tets) where k_pcg_pipe moves two thirds of it; it replaces the
// two-launch iteration (k_spmv + k_cg_fused: +46 MB of vector traffic per million tets and two kernel boundaries) in that range.
#pragma once
#include "pcg_pipe.hip.h"

namespace fb {

constexpr int kPipe2LdsWordsPerRow = 24;  // x[3], p[3], z[3] as doubles (18 words) + 6 floats of the diagonal block's low part
constexpr int kPipe2SliceBytes = kPipe2LdsWordsPerRow * 64 * 4;  // the vector area of one slice
constexpr int kPipe2Klt = 4;               // most LDS-resident slots per slice (the unroll bound of the LDS loop)
// LDS-resident slots a workgroup with `count` slices has room for: the sync buffers, count + 1 vector areas (the last one is the dump of
// the wavefronts' unused row sets), the rest in wavefront-slots
__host__ __device__ constexpr int pipe2_lds_slots(int count, bool c16) {
  return (160 * 1024 - (int)sizeof(double) * kPipeSyncDoubles - (count + 1) * kPipe2SliceBytes) / pipe_slot_bytes(c16) > 0
             ? (160 * 1024 - (int)sizeof(double) * kPipeSyncDoubles - (count + 1) * kPipe2SliceBytes) / pipe_slot_bytes(c16) : 0;
}

// SHARD: the kernel of a sharded handle ("k_pcg_pipe2_shard" in fb_fem_pcg_path; pcg_shard_box.hip.h) -- as for k_pcg_pipe
// XYZ: the published vector node by node instead of in planes (pcg_pipe.hip.h; unstructured meshes, unsharded handles)
template <bool C16, bool SHARD, bool XYZ = false>
__global__ __launch_bounds__(64 * kPipeMaxWaves) void k_pcg_pipe2(SellView sv, const float* __restrict__ vals, const float* __restrict__ dlo,
                                                                  const double* __restrict__ invdiag, const double* __restrict__ bvec,
                                                                  double* __restrict__ xg, double* __restrict__ rg, double* __restrict__ wg,
                                                                  double* __restrict__ zg, double* __restrict__ sg, double* __restrict__ pg,
                                                                  CGState* __restrict__ st, PipeArgs pa, ShardArgs sa) {
  static_assert(!SHARD || !C16, "a shard's columns are 32-bit local ids");
  static_assert(!SHARD || !XYZ, "a shard publishes planes");
  extern __shared__ double lds[];
  double* wsum = lds;                          // [2][16] wave sums
  double* gath = lds + 32;                     // [2][kPipeMaxBlocks] all workgroups' sums
  double* bc = gath + 2 * kPipeMaxBlocks;      // [0..1] totals, [2] a wait failed (sweep), [3] a wait failed (product), [4] a sum poller gave up
  const int n_waves = blockDim.x >> 6, nb = gridDim.x;
  if (pa.start == 0 && st->done) return;  // grid-uniform: written by an earlier launch
  if (threadIdx.x == 0) bc[4] = 0.0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int first, count;
  if constexpr (SHARD) { first = sa.wg_range[blockIdx.x].x; count = sa.wg_range[blockIdx.x].y; }
  else pipe_deal(pa.wg_first, sv.n_slices, nb, blockIdx.x, &first, &count);
  const bool spare = SHARD && wv == n_waves - 1;  // a shard's spare wavefront: sums, counters, proxy copies (it owns no slice)
  bool live[2], rvalid[2];
  int row[2], so[2], width[2], send_beg[2], send_end[2], jls[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int jl = wv + h * (n_waves - (SHARD ? 1 : 0));  // (the slice wavefronts: all, or all but a shard's spare one)
    live[h] = jl < count && !spare;  // wave-uniform
    jls[h] = live[h] ? jl : count;   // (the place of its vectors in LDS; `count` = the dump)
    const int sl = first + jl;
    row[h] = sl * 64 + lane;
    rvalid[h] = live[h] && row[h] < sv.n_owned;
    int o = 0, wd = 0;
    if (live[h]) { o = sv.slice_off[sl]; wd = sv.slice_off[sl + 1] - o; }
    so[h] = __builtin_amdgcn_readfirstlane(o); width[h] = __builtin_amdgcn_readfirstlane(wd);
    send_beg[h] = send_end[h] = 0;  // (SHARD) this row's entries of the send lists
    if constexpr (SHARD) if (rvalid[h]) { send_beg[h] = sa.row_send_off[row[h]]; send_end[h] = sa.row_send_off[row[h] + 1]; }
  }
  // LDS of this wavefront and row set h: nine doubles (x, p, z: k = 0..8) then six floats (low diagonal part) per lane, each as a
  // plane of 64 lanes -- 24 words per row, conflict-free
  // (the areas of the workgroup's slices lie one after the other, slice j at j; a row set without a slice -- the recurrences below run
  // over both row sets unconditionally -- reads and writes the dump area behind them, zeros from every wavefront that does)
  char* lbase = (char*)(lds + kPipeSyncDoubles);
  auto lds_d = [&](int h, int k) -> double* { return (double*)(lbase + (size_t)jls[h] * kPipe2SliceBytes + (size_t)k * 512) + lane; };
  auto lds_f = [&](int h, int k) -> float* { return (float*)(lbase + (size_t)jls[h] * kPipe2SliceBytes + 9 * 512 + (size_t)k * 256) + lane; };
  // LDS-resident part of the matrix: what the vector areas leave, dealt to the slices as in k_pcg_pipe (count-th part each, the remainder
  // one more for the first ones, at most kPipe2Klt); C16: [klt][9][64] value words then [klt][64] column differences, else [klt][10][64] words
  constexpr int kValWords = C16 ? 9 : 10;
  const int pool = pipe2_lds_slots(count, C16);
  const int pbase = min(kPipe2Klt, pool / max(count, 1)), prem = pbase < kPipe2Klt ? min(count, pool - pbase * count) : 0;  // workgroup-uniform
  int klt[2], KL[2];
  char* lres[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    klt[h] = __builtin_amdgcn_readfirstlane(live[h] ? pbase + (jls[h] < prem ? 1 : 0) : 0);
    KL[h] = min(klt[h], width[h]);
    lres[h] = lbase + (size_t)(count + 1) * kPipe2SliceBytes + (size_t)(jls[h] * pbase + min(jls[h], prem)) * pipe_slot_bytes(C16);
    unsigned int* lv = (unsigned int*)lres[h] + lane;
    short* lc = (short*)(lres[h] + (size_t)klt[h] * 9 * 256) + lane;
    for (int k = 0; k < KL[h]; k++) {
      const float* vk = vals + ((size_t)so[h] + k) * 9 * 64 + lane;
#pragma unroll
      for (int j = 0; j < 9; j++) lv[(k * kValWords + j) * 64] = __float_as_uint(vk[j * 64]);
      if constexpr (C16) lc[k * 64] = sv.coldelta[((size_t)so[h] + k) * 64 + lane];
      else lv[(k * 10 + 9) * 64] = (unsigned int)sv.colidx[((size_t)so[h] + k) * 64 + lane];
    }
  }
  double iv[2][3] = {{0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int h = 0; h < 2; h++) {
    float m6[6] = {0, 0, 0, 0, 0, 0};
    if (rvalid[h]) {
      const float* l = dlo + (size_t)(row[h] >> 6) * 9 * 64 + lane;
      m6[0] = l[0 * 64]; m6[1] = l[1 * 64]; m6[2] = l[2 * 64]; m6[3] = l[4 * 64]; m6[4] = l[5 * 64]; m6[5] = l[8 * 64];
#pragma unroll
      for (int a = 0; a < 3; a++) iv[h][a] = invdiag[3 * (size_t)row[h] + a];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) *lds_f(h, k) = m6[k];
#pragma unroll
    for (int k = 0; k < 9; k++) *lds_d(h, k) = 0.0;
  }
  const int n_prod = pa.prod_count[blockIdx.x];
  int my_prod = -1;
  if (wv == 0 && n_prod >= 0 && lane < n_prod) my_prod = pa.producers[(size_t)blockIdx.x * kPipeMaxProducers + lane];

  unsigned int send_mask = 0u;  // (SHARD) workgroup-uniform
  ShardBoxLayout BL = {};
  if constexpr (SHARD) { send_mask = sa.wg_send_mask[blockIdx.x]; BL = shard_box_layout(sa.halo_cap); }
  unsigned int pub = pa.seqs[0], sums = pa.seqs[1];
  const long long t_limit = pa.timeout_ticks;
  bool failed = false;

  // (publish / product: as in k_pcg_pipe, for the two rows of the lane; write-through stores always -- the plain-store form is
  // for the latency-bound small systems)
  const size_t xs = XYZ ? 1 : pa.n_pad, cs = XYZ ? 3 : 1;  // strides of a component and of a column in the published vector
  auto publish = [&](const double vin[2][3]) {
    pub++;
    double* pl = pa.planes + (size_t)(pub & 1u) * 3 * pa.n_pad;
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (rvalid[h]) {
        if constexpr (XYZ) {
          st_sc1_xyz(pl + 3 * (size_t)row[h], vin[h][0], vin[h][1], vin[h][2]);
        } else {
#pragma unroll
          for (int a = 0; a < 3; a++) st_sc1_f64(pl + a * xs + cs * (size_t)row[h], vin[h][a]);
        }
        if constexpr (SHARD) shard_send_row(sa, BL, pub, send_beg[h], send_end[h], vin[h]);
      }
  };
  auto product = [&](const double vin[2][3], double y[2][3], bool post_sums, bool late_acquire) {
    double* pl = pa.planes + (size_t)(pub & 1u) * 3 * pa.n_pad;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (post_sums) sums++;
    if (wv != 0 && live[0] && pa.prefetch_slots > 0 && width[0] > KL[0]) {
      int so_k = so[0] + KL[0];
      asm volatile("" : "+s"(so_k));
      pipe_prefetch_values(min(pa.prefetch_slots, width[0] - KL[0]), ((unsigned int)so_k * 9u * 64u + (unsigned int)lane) * (unsigned int)sizeof(float), vals);
    }
    if constexpr (SHARD) {
      if (spare) shard_service_product(sa, BL, pa, pub, pl, nb, lane, send_mask, t_limit, bc, failed);
    }
    if (wv == 0) {
      if (lane == 0) st_sc1_u32(pa.flags + blockIdx.x, pub);
      if (post_sums && lane < 2) {
        double t = 0.0;
        for (int w = 0; w < n_waves; w++) t += wsum[lane * 16 + w];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(t), tag = (unsigned long long)sums << 32;
        unsigned long long* post = pa.post + ((size_t)(sums & 1u) * nb + blockIdx.x) * 4 + 2 * lane;
        st_sc1_u64(post, (bits >> 32) | tag);
        st_sc1_u64(post + 1, (bits & 0xffffffffULL) | tag);
      }
      if (!late_acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (early where the product follows a sums sweep, else after the poll: pcg_pipe.hip.h)
      const long long t0 = wall_clock64();
      if (n_prod >= 0) {
        for (;;) {
          bool ok = true;
          if (my_prod >= 0) ok = (int)(ld_sc1_u32(pa.flags + my_prod) - pub) >= 0;
          if (__ballot(!ok) == 0ULL) break;
          if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      } else if constexpr (SHARD) {
        shard_poll_all(sa, pa, pub, nb, lane, t0, t_limit, failed);
      } else {
        for (int b = 4 * lane; b - 4 * lane < nb && !failed; b += 256) {
          for (;;) {
            bool ok = true;
            if (b < nb) {
              const uint4 f = ld_sc1_u128(pa.flags + b);
              ok = (int)(f.x - pub) >= 0 && (b + 1 >= nb || (int)(f.y - pub) >= 0) && (b + 2 >= nb || (int)(f.z - pub) >= 0) && (b + 3 >= nb || (int)(f.w - pub) >= 0);
            }
            if (__ballot(!ok) == 0ULL) break;
            if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (failed && lane == 0) st_sc1_u32(pa.error, 1u);
      if (late_acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) bc[3] = (failed || ld_sc1_u32(pa.error) != 0u) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (uniform_flag(bc[3] != 0.0 || (SHARD && bc[4] != 0.0))) { failed = true; return; }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      double y0 = 0, y1 = 0, y2 = 0;
      if (rvalid[h]) {  // the low part of the diagonal block times the own entry
        const double l00 = (double)*lds_f(h, 0), l01 = (double)*lds_f(h, 1), l02 = (double)*lds_f(h, 2), l11 = (double)*lds_f(h, 3), l12 = (double)*lds_f(h, 4),
                     l22 = (double)*lds_f(h, 5);
        y0 = l00 * vin[h][0] + l01 * vin[h][1] + l02 * vin[h][2];
        y1 = l01 * vin[h][0] + l11 * vin[h][1] + l12 * vin[h][2];
        y2 = l02 * vin[h][0] + l12 * vin[h][1] + l22 * vin[h][2];
      }
      if (live[h]) {  // LDS-resident slots
        const unsigned int* lv = (const unsigned int*)lres[h] + lane;
        const short* lc = (const short*)(lres[h] + (size_t)klt[h] * 9 * 256) + lane;
#pragma unroll
        for (int k = 0; k < kPipe2Klt; k++) if (k < KL[h]) {
          const unsigned int* lk = lv + (size_t)k * kValWords * 64;
          unsigned int col;
          if constexpr (C16) col = (unsigned int)(row[h] + (int)lc[k * 64]); else col = lk[9 * 64];
          const double* xp = pl + cs * (size_t)col;
          const double x0 = xp[0], x1 = xp[xs], x2 = xp[2 * xs];
          y0 += (double)__uint_as_float(lk[0 * 64]) * x0 + (double)__uint_as_float(lk[1 * 64]) * x1 + (double)__uint_as_float(lk[2 * 64]) * x2;
          y1 += (double)__uint_as_float(lk[3 * 64]) * x0 + (double)__uint_as_float(lk[4 * 64]) * x1 + (double)__uint_as_float(lk[5 * 64]) * x2;
          y2 += (double)__uint_as_float(lk[6 * 64]) * x0 + (double)__uint_as_float(lk[7 * 64]) * x1 + (double)__uint_as_float(lk[8 * 64]) * x2;
        }
      }
      if (live[h] && width[h] > KL[h]) {
        int so_k = so[h] + KL[h];
        asm volatile("" : "+s"(so_k));
        pipe_stream_slots<C16, XYZ>(width[h] - KL[h], ((unsigned int)so_k * 9u * 64u + (unsigned int)lane) * (unsigned int)sizeof(float),
                               ((unsigned int)so_k * 64u + (unsigned int)lane) * (unsigned int)(C16 ? sizeof(short) : sizeof(int)), vals,
                               C16 ? (const void*)sv.coldelta : (const void*)sv.colidx, pl, pl + xs, pl + 2 * xs, row[h], y0, y1, y2);
      }
      y[h][0] = y0; y[h][1] = y1; y[h][2] = y2;
    }
  };

  double rr[2][3] = {{0, 0, 0}, {0, 0, 0}}, wr[2][3] = {{0, 0, 0}, {0, 0, 0}}, sr[2][3] = {{0, 0, 0}, {0, 0, 0}};
  double rho0 = 0.0, eps2 = pa.eps2, gamma_old = 1.0, alpha_old = 1.0;
  int iter = 0, max_iter = pa.max_iter;
  enum { PH_WARM_X = 0, PH_INIT_W = 1, PH_ITER = 2, PH_REFRESH_X = 3, PH_REFRESH_W = 4 };
  int phase = PH_ITER;
  bool fresh = pa.start != 0;
  if (pa.start == 0) {
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (rvalid[h]) {
        const size_t dof = 3 * (size_t)row[h];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          rr[h][a] = rg[dof + a]; wr[h][a] = wg[dof + a]; sr[h][a] = sg[dof + a];
          *lds_d(h, a) = xg[dof + a]; *lds_d(h, 3 + a) = pg[dof + a]; *lds_d(h, 6 + a) = zg[dof + a];
        }
      }
    rho0 = uniform_f64(st->rho0); eps2 = uniform_f64(st->eps2); iter = st->iter; max_iter = st->max_iter;
    gamma_old = uniform_f64(pa.pstate[0]); alpha_old = uniform_f64(pa.pstate[1]);
  } else if (pa.start == 2) {
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (rvalid[h]) {
#pragma unroll
        for (int a = 0; a < 3; a++) *lds_d(h, a) = xg[3 * (size_t)row[h] + a];
      }
    phase = PH_WARM_X;
  } else {
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (rvalid[h]) {
#pragma unroll
        for (int a = 0; a < 3; a++) rr[h][a] = bvec[3 * (size_t)row[h] + a];
      }
    phase = PH_INIT_W;
  }

  bool unsettled = true;  // the next product does not follow a sums sweep of this launch: acquire after the poll
  bool done = false, published = false;
  double gamma = 0.0;
  int it_done = 0;
  while (!failed) {
    if (phase == PH_ITER && it_done >= pa.n_iters) break;
    double vin[2][3];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int a = 0; a < 3; a++)
        vin[h][a] = (phase == PH_WARM_X || phase == PH_REFRESH_X) ? *lds_d(h, a) : (phase == PH_ITER ? iv[h][a] * wr[h][a] : iv[h][a] * rr[h][a]);
    if (!published) publish(vin);
    published = false;
    if (phase == PH_ITER) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const double u[3] = {iv[h][0] * rr[h][0], iv[h][1] * rr[h][1], iv[h][2] * rr[h][2]};
        // (row set 1 is added to row set 0's sum: a fixed order)
        a0 += rr[h][0] * u[0] + rr[h][1] * u[1] + rr[h][2] * u[2];
        a1 += wr[h][0] * u[0] + wr[h][1] * u[1] + wr[h][2] * u[2];
      }
      a0 = wave_sum(a0); a1 = wave_sum(a1);
      int wvo = wv;
      asm volatile("" : "+v"(wvo));
      if (lane == 0) { wsum[wvo] = a0; wsum[16 + wvo] = a1; }
    }
    double y[2][3];
    product(vin, y, phase == PH_ITER, phase != PH_ITER || unsettled);
    unsettled = phase != PH_ITER;
    if (failed) break;
    if (phase == PH_WARM_X || phase == PH_REFRESH_X) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        unsigned int d3 = 3u * (unsigned int)(rvalid[h] ? row[h] : 0);
        asm volatile("" : "+v"(d3));
#pragma unroll
        for (int a = 0; a < 3; a++) rr[h][a] = rvalid[h] ? bvec[d3 + a] - y[h][a] : 0.0;
      }
      phase = phase == PH_WARM_X ? PH_INIT_W : PH_REFRESH_W;
      continue;
    }
    if (phase != PH_ITER) {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int a = 0; a < 3; a++) wr[h][a] = y[h][a];
      phase = PH_ITER;
      continue;
    }
    // ---- all workgroups' sums ----
    if constexpr (SHARD) {
      if (spare) shard_rank_sums(sa, BL, pa, sums, nb, lane, t_limit, bc, failed);
      __syncthreads();
    } else if (pa.service) {
      if (wv == n_waves - 1) {
        const long long t0 = wall_clock64();
        double t0s = 0, t1s = 0;
        pipe_collect_posts(pa, sums, nb, lane, 64, lane, t0, t_limit, failed, t0s, t1s);
        failed = uniform_flag(failed);
        if (failed && lane == 0) st_sc1_u32(pa.error, 1u);
        t0s = wave_sum(t0s); t1s = wave_sum(t1s);
        if (lane == 0) { bc[0] = t0s; bc[1] = t1s; bc[2] = (failed || ld_sc1_u32(pa.error) != 0u) ? 1.0 : 0.0; }
      }
      __syncthreads();
    } else {
      const int pollers = min(n_waves, 4);
      if (wv < pollers) {
        const long long t0 = wall_clock64();
        const unsigned long long* post = pa.post + (size_t)(sums & 1u) * nb * 4;
        for (int b = wv * 64 + lane; b - lane < nb && !failed; b += pollers * 64) {
          const bool mine = b < nb;
          uint4 q4[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
          for (;;) {
            bool ok = true;
            if (mine) {
              asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                           : "=&v"(q4[0]), "=&v"(q4[1]) : "v"(post + (size_t)b * 4) : "memory");
              ok = q4[0].y == sums && q4[0].w == sums && q4[1].y == sums && q4[1].w == sums;
            }
            if (__ballot(!ok) == 0ULL) break;
            if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (mine && !failed) {
            gath[b] = __longlong_as_double((long long)(((unsigned long long)q4[0].x << 32) | (unsigned long long)q4[0].z));
            gath[kPipeMaxBlocks + b] = __longlong_as_double((long long)(((unsigned long long)q4[1].x << 32) | (unsigned long long)q4[1].z));
          }
        }
        if (failed && lane == 0) { st_sc1_u32(pa.error, 1u); bc[4] = 1.0; }
      }
      __syncthreads();
      if (wv == 0) {
        double t0s = 0, t1s = 0;
        int l0 = lane;
        asm volatile("" : "+v"(l0));
        for (int b = l0; b < nb; b += 64) { t0s += gath[b]; t1s += gath[kPipeMaxBlocks + b]; }
        t0s = wave_sum(t0s); t1s = wave_sum(t1s);
        if (lane == 0) { bc[0] = t0s; bc[1] = t1s; bc[2] = (bc[4] != 0.0 || ld_sc1_u32(pa.error) != 0u) ? 1.0 : 0.0; }
      }
      __syncthreads();
    }
    if (uniform_flag(bc[2] != 0.0)) { failed = true; break; }
    gamma = uniform_f64(bc[0]);
    const double delta = uniform_f64(bc[1]);
    if (fresh) rho0 = gamma;
    if (!(gamma > eps2 * rho0) || iter >= max_iter) { done = true; break; }
    double alpha, beta;
    if (fresh) { beta = 0.0; alpha = gamma / delta; }
    else { beta = gamma / gamma_old; alpha = gamma / (delta - beta * gamma / alpha_old); }
    fresh = false;
    iter++;
    it_done++;
    gamma_old = gamma; alpha_old = alpha;
    const bool refresh = iter % 30 == 0;
    // ---- recurrences: z, p, x in LDS, the rest in registers ----
    double zn[2][3];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int a = 0; a < 3; a++) {
        zn[h][a] = y[h][a] + beta * *lds_d(h, 6 + a);
        *lds_d(h, 6 + a) = zn[h][a];
        sr[h][a] = wr[h][a] + beta * sr[h][a];
      }
    if (!refresh) {
      double m[2][3];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int a = 0; a < 3; a++) { wr[h][a] = wr[h][a] - alpha * zn[h][a]; m[h][a] = iv[h][a] * wr[h][a]; }
      publish(m);
      published = true;
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const double pn = iv[h][a] * rr[h][a] + beta * *lds_d(h, 3 + a);
        *lds_d(h, 3 + a) = pn;
        *lds_d(h, a) = *lds_d(h, a) + alpha * pn;
      }
    if (refresh) {
      phase = PH_REFRESH_X;
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int a = 0; a < 3; a++) rr[h][a] = rr[h][a] - alpha * sr[h][a];
    }
  }
  if (failed) return;  // nothing written back: the host re-solves from the vectors it handed over
  const bool capped = done && gamma > eps2 * rho0;  // (the iteration cap: x goes to pg for the host to check, pcg_pipe.hip.h)
#pragma unroll
  for (int h = 0; h < 2; h++)
    if (rvalid[h]) {
      unsigned int dof = 3u * (unsigned int)row[h];
      asm volatile("" : "+v"(dof));
      if (capped) {
#pragma unroll
        for (int a = 0; a < 3; a++) pg[dof + a] = *lds_d(h, a);
        continue;
      }
#pragma unroll
      for (int a = 0; a < 3; a++) xg[dof + a] = *lds_d(h, a);
      if (!done) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
          rg[dof + a] = rr[h][a]; wg[dof + a] = wr[h][a]; sg[dof + a] = sr[h][a];
          pg[dof + a] = *lds_d(h, 3 + a); zg[dof + a] = *lds_d(h, 6 + a);
        }
      }
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->rho0 = rho0; st->eps2 = eps2; st->max_iter = max_iter;
    st->iter = iter;
    st->rho[iter & 1] = done ? gamma : gamma_old;
    st->done = done ? 1 : 0;
    pa.pstate[0] = gamma_old; pa.pstate[1] = alpha_old;
    pa.seqs[0] = (SHARD && published) ? pub - 1u : pub; pa.seqs[1] = sums;  // (a shard's launch cut after a pre-publish takes the publish back: pcg_pipe.hip.h)
  }
}

}  // namespace fb
