// Device side of the peer-to-peer transport (comm.h): inbox layout and the post / wait primitives shared by the
// stand-alone exchange kernels (p2p.hip) and the PCG kernels that carry an exchange in their prologue / epilogue.
#pragma once
#include <hip/hip_runtime.h>

#include "comm.h"
#include "fem_kernels.h"

namespace fb {

// inbox layout (bytes)
constexpr int kP2PChunks = 16;         // a neighbour's halo values arrive in up to 16 chunks, each from its own sender block
constexpr size_t kOffHaloFlag = 0;     // u64[kP2PMaxRanks][kP2PChunks]   written by peer q at [q][chunk]
constexpr size_t kOffRedFlag = 2048;   // u64[kP2PMaxRanks]
constexpr size_t kOffErr = 2304;       // u64
constexpr size_t kOffRed = 2560;       // double[2][kP2PMaxRanks][8]
constexpr size_t kOffHalo = 8192;      // double[2][cap * 12]
constexpr int kP2PMaxWidth = 12;

__device__ __forceinline__ unsigned long long ld_acquire_sys(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_release_sys(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Development knob for the one-GPU rehearsals of N > 1 (FEMBRAIN_REMOTE_DELAY_US = 1 / 2 / 5 ...): between processes that
// share one GPU a "remote" store lands in the same HBM within a microsecond, across xGMI it does not.  With the knob set, every signal a
// rank raises for a peer (chunk flag, sum flag, box counter, rank sums) is raised `ticks` later than its data was stored, and every such
// signal is acted on `ticks` after it was seen -- the protocol then runs as if each hop took that much longer.  0: nothing is executed.
__device__ __forceinline__ void remote_delay(long long ticks) {
  if (ticks <= 0) return;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

// spin until *flag >= seq; bounded by the wall clock.  A timeout anywhere poisons the inbox so that the queue drains.
__device__ inline bool p2p_wait_flag(const P2PDev& c, const unsigned long long* flag, unsigned long long seq) {
  unsigned long long* err = (unsigned long long*)(c.inbox + kOffErr);
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return false;
  const long long t0 = wall_clock64();
  while (ld_acquire_sys(flag) < seq) {
    if (wall_clock64() - t0 > c.timeout_ticks) {
      __hip_atomic_store(err, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
  remote_delay(c.delay_ticks);
  return true;
}

// ONE whole block, after every block's partial sums are visible: fold them in fixed order and post the `count` totals
// into every rank's inbox (own included).  lds: 4 doubles, mine: 8 doubles of shared memory.
__device__ inline void p2p_post_sums(const P2PDev& c, unsigned long long seq, const double* partial, int n, int count, double* lds, double* mine) {
  for (int k = 0; k < count; k++) {
    const double s = sum_partials(partial + (size_t)k * n, n, lds);
    if (threadIdx.x == 0) mine[k] = s;
  }
  __syncthreads();
  const int par = (int)(seq & 1ULL), t = threadIdx.x;
  if (t < c.n_ranks) {
    double* slot = (double*)(c.peer[t] + kOffRed) + ((size_t)par * kP2PMaxRanks + c.rank) * 8;
    for (int k = 0; k < count; k++) slot[k] = mine[k];
    __threadfence_system();
    remote_delay(c.delay_ticks);
    st_release_sys((unsigned long long*)(c.peer[t] + kOffRedFlag) + c.rank, seq);
  }
}

// any block: wait for every rank's post of `seq` and add the slots in rank order; all threads return after the totals
// are in tot[0..count) (shared memory, >= 8 doubles)
__device__ inline void p2p_wait_sums(const P2PDev& c, unsigned long long seq, int count, double* tot) {
  const int par = (int)(seq & 1ULL), t = threadIdx.x;
  // lane r polls rank r's flag: the flags share one 128-byte row, so a block's poll is one request and one latency
  if (t < c.n_ranks) p2p_wait_flag(c, (const unsigned long long*)(c.inbox + kOffRedFlag) + t, seq);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (t < count) {
    const double* slots = (const double*)(c.inbox + kOffRed) + (size_t)par * kP2PMaxRanks * 8;
    double s = 0.0;
    for (int r = 0; r < c.n_ranks; r++) s += __builtin_nontemporal_load(slots + (size_t)r * 8 + t);
    tot[t] = s;
  }
  __syncthreads();
}

// A neighbour's n nodes of `width` doubles travel in this many chunks (sender and receiver compute it alike): one sender
// block per chunk keeps many stores in flight (a single block sending a whole 9k-node plane was 2.5x slower) and each
// chunk has its own flag, so there is no grid-wide ticket.
__device__ __forceinline__ int p2p_chunks(int n, int width) { return max(1, min(kP2PChunks, (n * width + 1023) / 1024)); }

// every block, before it reads halo values from the inbox: wait for all chunks of the neighbours' post `seq`.  Thread
// 16 q + k polls chunk k of neighbour q, so all flags are polled side by side (one 128-byte flag row per neighbour) and
// the wait costs one poll latency, not one per flag.
__device__ inline void p2p_wait_halo(const P2PDev& c, unsigned long long seq, const int* __restrict__ halo_off, int width) {
  static_assert(kP2PMaxRanks * kP2PChunks <= kBlock, "one polling thread per (neighbour, chunk)");
  const int q = threadIdx.x / kP2PChunks, k = threadIdx.x % kP2PChunks;
  if (q < c.n_ranks && q != c.rank) {
    const int n = halo_off[q + 1] - halo_off[q];
    if (n > 0 && k < p2p_chunks(n, width)) p2p_wait_flag(c, (const unsigned long long*)(c.inbox + kOffHaloFlag) + q * kP2PChunks + k, seq);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// where the 3-vector halo values posted under `seq` sit in my inbox (indexed by halo node: local id - n_owned)
__device__ inline const double* p2p_halo_in(const P2PDev& c, unsigned long long seq) {
  return (const double*)(c.inbox + kOffHalo) + (size_t)(seq & 1ULL) * c.cap * kP2PMaxWidth;
}

// The sender jobs of a halo refresh -- one per (neighbour, chunk) -- are dealt round-robin to the blocks of the grid: a block
// stores its chunk of my boundary values into the neighbour's inbox and releases `seq` on that chunk's flag.  vec was
// completed by an earlier kernel, so the only fence is the sender's own.
__device__ inline void p2p_send_halo_jobs(const P2PDev& c, unsigned long long seq, int width, const int* __restrict__ send_ids,
                                          const int* __restrict__ send_off, const double* __restrict__ vec) {
  int job = 0;
  for (int q = 0; q < c.n_ranks; q++) {
    const int first = send_off[q], n = send_off[q + 1] - first;
    if (q == c.rank || n <= 0) continue;
    const int nc = p2p_chunks(n, width);
    for (int k = 0; k < nc; k++, job++) {
      if (job % (int)gridDim.x != (int)blockIdx.x) continue;  // block-uniform
      const int lo = (int)((long long)n * k / nc), hi = (int)((long long)n * (k + 1) / nc);
      double* dst = (double*)(c.peer[q] + kOffHalo) + (size_t)(seq & 1ULL) * c.peer_cap[q] * kP2PMaxWidth + (size_t)c.peer_seg[q] * width;
      for (int i = lo * width + threadIdx.x; i < hi * width; i += kBlock) {
        const int node = i / width, cc = i - node * width;
        dst[i] = vec[(size_t)width * send_ids[first + node] + cc];
      }
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) { remote_delay(c.delay_ticks); st_release_sys((unsigned long long*)(c.peer[q] + kOffHaloFlag) + c.rank * kP2PChunks + k, seq); }
    }
  }
}

}  // namespace fb
