// The SHARD = true instantiations of k_pcg_pipe / k_pcg_pipe2 (fb_fem_pcg_path: "k_pcg_pipe_shard<..>", "k_pcg_pipe2_shard"; rounds 3: two
// kernels of their own, near-copies of the unsharded ones -- round 4: one template each, and this file holds what only a shard does):
// the persistent pipelined Jacobi-PCG of pcg_pipe.hip.h on a SHARDED handle -- one launch per solve on every rank,
// the halo values and the global sums crossing the GPUs inside the launches (stores, atomics and polls on peer memory mapped through
// HIP IPC over xGMI; no collective library on the path).  UNMEASURED ON HARDWARE: the build box has one GPU; the kernel is exercised
// by two processes on one GPU, each confined to half the CUs (FEMBRAIN_CU_MASK), against the unsharded handle
// (tests/test_sharded_gpu.py).  Opt-in: FEMBRAIN_SHARDED_PERSIST=1.
//
// What the pipelined recurrence buys across GPUs is the same it buys across CUs: gamma and delta are sums of local quantities, known
// before the product starts, so the ALL-RANK reduction (rank sums posted into every peer's box, read after the product) hides behind
// the product, and the only wait is for the neighbours' part of the product's input vector -- now including the halo rows a
// neighbour RANK owns.  Per rank, in a fine-grained buffer the peers map ("box"):
//   counters[s]        u32, rank s's workgroups that own rows of MY halo add 1 each per publish (remote atomic) after their stores
//   rsum[p][s][4]      tagged 8-byte granules: rank s's gamma, delta of the sum sequence with parity p (posted by its workgroup 0)
//   halo[p][3][n_halo] the halo rows of the published vector, written by the owners' lanes straight from their registers
// and in ordinary device memory the planes [2][3][n_pad] of pcg_pipe.hip.h, n_pad covering owned AND halo columns: a PROXY -- the
// spare wavefront of one workgroup per neighbour rank -- waits for counters[s] to reach (workgroups of s that send to me) x (publish
// number), copies s's segment from the box into the halo part of the planes (so that the products' gathers stay cached, ordinary
// loads), drains and raises a flag of its own; slices with halo columns of rank s have that flag in their producer list.  A segment
// is dealt to up to 16 proxies (s, k) in 64-row groups -- a copy is a chain of uncached loads, so it is spread over as many
// wavefronts as the workgroups can spare (at most 4 duties per workgroup) -- with flags[n_blocks + s * n_proxy + k].  Hazards as in pcg_pipe.hip.h: the box's halo area is double-buffered by publish parity, and a sender can be two
// publishes ahead of a receiver only after it has received that receiver's publish in between (the neighbour relation is symmetric).
// Everything else -- state in registers, LDS-resident slots, the assembly loop over the streamed slots (32-bit local column ids),
// exact-residual iterations, launch cuts, bounded waits -- is k_pcg_pipe's.  A workgroup needs its spare wavefront (at most 11 / 7
// slices per CU).
#pragma once
// (included by pcg_pipe.hip.h after PipeArgs and the sc1 helpers)
#include "p2p_device.hip.h"
namespace fb {

struct ShardBoxLayout {  // byte offsets inside a rank's box (the same on every rank: n_halo_cap = the largest halo of all ranks)
  size_t counters, rsum, halo, bytes;
  long long halo_cap;
};
__host__ __device__ inline ShardBoxLayout shard_box_layout(long long halo_cap) {
  ShardBoxLayout L;
  L.halo_cap = halo_cap;
  L.counters = 0;
  L.rsum = 256;                                                     // kP2PMaxRanks * 4 B, padded
  L.halo = L.rsum + (size_t)2 * kP2PMaxRanks * 4 * 8;
  L.halo = (L.halo + 255) & ~(size_t)255;
  L.bytes = L.halo + (size_t)2 * 3 * (size_t)(halo_cap > 0 ? halo_cap : 1) * sizeof(double);
  return L;
}

struct ShardArgs {
  int rank, n_ranks;
  int n_owned, n_halo;
  char* box;                          // mine
  char* const* peer_box;              // [n_ranks] device array; peer_box[rank] == box
  const int* peer_seg;                // [n_ranks] device array: first position of MY rows inside peer q's halo
  long long halo_cap;
  const int* halo_off;                // [n_ranks + 1] my halo segments by owner rank (device copy)
  const int* row_send_off;            // [n_owned + 1] per owned row: its entries in the two lists below
  const int* row_send_rank;           // destination rank
  const int* row_send_pos;            // position inside my segment of that rank's halo
  const unsigned int* wg_send_mask;   // [n_blocks] ranks this workgroup has rows to send to
  const int* n_senders;               // [n_ranks] workgroups of rank s that send to me
  int n_proxy;                        // proxies per source rank: proxy (s, k) copies the k-th chunk of rank s's segment
  const int* proxy_wg;                // [n_ranks * n_proxy] the workgroup whose spare wavefront is proxy (s, k) (-1: nothing to copy)
  const int* wg_duty;                 // [n_blocks * kShardDuties] the proxies (s * n_proxy + k) of every workgroup, -1 padded
  const int2* wg_range;               // [n_blocks] first slice and slice count of every workgroup (those that gather halo rows get fewer: fem.hip)
  long long delay_ticks;              // development (FEMBRAIN_REMOTE_DELAY_US, p2p_device.hip.h remote_delay)
};
constexpr int kShardProxies = 16;     // at most, per source rank
constexpr int kShardDuties = 4;       // at most, per workgroup
// rows [lo, hi) of a halo segment [h0, h1) that proxy k of n copies (whole 64-row groups)
__host__ __device__ inline void shard_proxy_rows(int h0, int h1, int n, int k, int* lo, int* hi) {
  const int chunk = (((h1 - h0) + n - 1) / n + 63) & ~63;
  *lo = h0 + k * chunk < h1 ? h0 + k * chunk : h1;
  *hi = *lo + chunk < h1 ? *lo + chunk : h1;
}

__device__ __forceinline__ void st_sys_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned int ld_sys_u32(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long ld_sys_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double ld_sys_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// per slice: lowest and highest OWNED column, and the set of ranks whose halo columns it touches (bit s)
__global__ __launch_bounds__(kBlock) void k_slice_colrange_shard(int n_slices, int n_owned, int n_ranks, const int* __restrict__ slice_off,
                                                                 const int* __restrict__ colidx, const int* __restrict__ halo_off, int4* __restrict__ out) {
  const int s = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int row = s * 64 + lane;
  int lo = 0x7fffffff, hi = -1;
  unsigned int ranks = 0;
  if (row < n_owned)
    for (int k = slice_off[s]; k < slice_off[s + 1]; k++) {
      const int c = colidx[(size_t)k * 64 + lane];
      if (c < n_owned) { lo = min(lo, c); hi = max(hi, c); }
      else {
        const int hidx = c - n_owned;
        int q = 0;
        while (q + 1 < n_ranks && hidx >= halo_off[q + 1]) q++;
        ranks |= 1u << q;
      }
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); ranks |= __shfl_down(ranks, off, 64);
  }
  if (lane == 0) out[s] = make_int4(lo, hi, (int)ranks, 0);
}


// publish: the rows a neighbour rank gathers go straight from the owner's registers into that rank's box
__device__ __forceinline__ void shard_send_row(const ShardArgs& sa, const ShardBoxLayout& BL, unsigned int pub, int send_beg, int send_end, const double* vin) {
  for (int e = send_beg; e < send_end; e++) {
    const int q = sa.row_send_rank[e];
    double* hq = (double*)(sa.peer_box[q] + BL.halo) + (size_t)(pub & 1u) * 3 * (size_t)sa.halo_cap + (size_t)(sa.peer_seg[q] + sa.row_send_pos[e]);
#pragma unroll
    for (int a = 0; a < 3; a++) st_sys_f64(hq + (size_t)a * (size_t)sa.halo_cap, vin[a]);
  }
}

// product, the spare wavefront, after the workgroup's stores have drained: the counters of the ranks that gather this workgroup's rows,
// then this workgroup's proxy duties (halo segments copied from the box into the planes, a flag of their own each)
__device__ __forceinline__ void shard_service_product(const ShardArgs& sa, const ShardBoxLayout& BL, const PipeArgs& pa, unsigned int pub, double* pl, int nb, int lane,
                                                      unsigned int send_mask, long long t_limit, double* bc, bool& failed) {
  remote_delay(sa.delay_ticks);
  if (lane < sa.n_ranks && (send_mask >> lane & 1u))
    __hip_atomic_fetch_add((unsigned int*)(sa.peer_box[lane] + BL.counters) + sa.rank, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64();
  for (int d = 0; d < kShardDuties && !failed; d++) {
    const int duty = sa.wg_duty[blockIdx.x * kShardDuties + d];  // workgroup-uniform
    if (duty < 0) break;
    const int s = duty / sa.n_proxy, k = duty - s * sa.n_proxy;
    const unsigned int want = (unsigned int)sa.n_senders[s] * pub;
    const unsigned int* cnt = (const unsigned int*)(sa.box + BL.counters) + s;
    while ((int)(ld_sys_u32(cnt) - want) < 0) {
      if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (failed) break;
    remote_delay(sa.delay_ticks);
    int lo, hi;
    shard_proxy_rows(sa.halo_off[s], sa.halo_off[s + 1], sa.n_proxy, k, &lo, &hi);
    const double* in = (const double*)(sa.box + BL.halo) + (size_t)(pub & 1u) * 3 * (size_t)sa.halo_cap;
    double* out = pl + (size_t)sa.n_owned;
    for (int i0 = lo + lane; i0 < hi; i0 += 4 * 64) {  // twelve loads in flight per lane
      double t[3][4];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int a = 0; a < 3; a++) t[a][j] = i0 + 64 * j < hi ? ld_sys_f64(in + (size_t)a * (size_t)sa.halo_cap + i0 + 64 * j) : 0.0;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int a = 0; a < 3; a++) if (i0 + 64 * j < hi) st_sc1_f64(out + a * pa.n_pad + i0 + 64 * j, t[a][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st_sc1_u32(pa.flags + nb + duty, pub);
  }
  failed = uniform_flag(failed);
  if (failed && lane == 0) { st_sc1_u32(pa.error, 1u); bc[4] = 1.0; }
}

// the poll-all form of a shard: all workgroups and all proxies
__device__ __forceinline__ void shard_poll_all(const ShardArgs& sa, const PipeArgs& pa, unsigned int pub, int nb, int lane, long long t0, long long t_limit, bool& failed) {
  const int n_flags = nb + sa.n_ranks * sa.n_proxy;
  for (int b = lane; b - lane < n_flags && !failed; b += 64) {
    for (;;) {
      bool ok = true;
      if (b < nb || (b < n_flags && sa.proxy_wg[b - nb] >= 0)) ok = (int)(ld_sc1_u32(pa.flags + b) - pub) >= 0;
      if (__ballot(!ok) == 0ULL) break;
      if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// the sums of an iteration, the spare wavefront: workgroup 0 collects this rank's posts and stores the rank sums into every rank's box;
// every workgroup then adds all ranks' sums from its own rank's box in rank order -- the same bits on every rank and workgroup
__device__ __forceinline__ void shard_rank_sums(const ShardArgs& sa, const ShardBoxLayout& BL, const PipeArgs& pa, unsigned int sums, int nb, int lane, long long t_limit,
                                                double* bc, bool& failed) {
  const long long t0 = wall_clock64();
  if (blockIdx.x == 0) {
    double t0s = 0, t1s = 0;
    pipe_collect_posts(pa, sums, nb, lane, 64, lane, t0, t_limit, failed, t0s, t1s);
    t0s = wave_sum(t0s); t1s = wave_sum(t1s);
    t0s = __shfl(t0s, 0, 64); t1s = __shfl(t1s, 0, 64);
    remote_delay(sa.delay_ticks);
    if (lane < sa.n_ranks && !failed) {  // lane q posts this rank's two sums into rank q's box
      unsigned long long* dst = (unsigned long long*)(sa.peer_box[lane] + BL.rsum) + ((size_t)(sums & 1u) * kP2PMaxRanks + sa.rank) * 4;
      const unsigned long long b0 = (unsigned long long)__double_as_longlong(t0s), b1 = (unsigned long long)__double_as_longlong(t1s), tag = (unsigned long long)sums << 32;
      st_sys_u64(dst, (b0 >> 32) | tag); st_sys_u64(dst + 1, (b0 & 0xffffffffULL) | tag);
      st_sys_u64(dst + 2, (b1 >> 32) | tag); st_sys_u64(dst + 3, (b1 & 0xffffffffULL) | tag);
    }
  }
  double g0 = 0, g1 = 0;
  {
    const unsigned long long* rs = (const unsigned long long*)(sa.box + BL.rsum) + ((size_t)(sums & 1u) * kP2PMaxRanks + (lane < sa.n_ranks ? lane : 0)) * 4;
    unsigned long long g[4] = {0, 0, 0, 0};
    while (!failed) {
      bool ok = true;
      if (lane < sa.n_ranks) {
#pragma unroll
        for (int k = 0; k < 4; k++) { g[k] = ld_sys_u64(rs + k); ok = ok && (unsigned int)(g[k] >> 32) == sums; }
      }
      if (__ballot(!ok) == 0ULL) break;
      if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    remote_delay(sa.delay_ticks);
    const double v0 = __longlong_as_double((long long)(((g[0] & 0xffffffffULL) << 32) | (g[1] & 0xffffffffULL)));
    const double v1 = __longlong_as_double((long long)(((g[2] & 0xffffffffULL) << 32) | (g[3] & 0xffffffffULL)));
    for (int q = 0; q < sa.n_ranks; q++) { g0 += __shfl(v0, q, 64); g1 += __shfl(v1, q, 64); }
  }
  failed = uniform_flag(failed);
  if (failed && lane == 0) st_sc1_u32(pa.error, 1u);
  if (lane == 0) { bc[0] = g0; bc[1] = g1; bc[2] = (failed || ld_sc1_u32(pa.error) != 0u) ? 1.0 : 0.0; }
}

}  // namespace fb
