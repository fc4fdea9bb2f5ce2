// Persistent PIPELINED Jacobi-PCG (included by fem.hip only): a whole solve of CGSolver.cpp:129-190 inside ONE launch, for
// systems whose vectors fit on the chip.  One workgroup per CU owns a fixed run of SELL slices, one wavefront per slice, one
// lane per block row; the lane keeps its row of x, r, w, z, s, p and 1/diag in REGISTERS for the whole solve.
//
// Why pipelined.  The merged-reduction iteration of round 2 (k_pcg_persist) crossed the chip twice per iteration: the three
// sums all-to-all (alpha, beta are needed before the new search direction exists), then the new direction to the neighbours
// -- 9 of its 22 us at 1M tets were those two hand-offs, with the memory system idle.  The recurrences of Ghysels & Vanroose
// (pipelined CG, Parallel Computing 40 (2014), Alg. 4; M = diag folded in: u = r/diag, m = w/diag) compute the SAME iterates in
// exact arithmetic from
//     gamma = r . u,  delta = w . u                (both sums of LOCAL quantities, known before the product starts)
//     n = A m,  m = w / diag                       (the product of the iteration)
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)
//     z = n + beta z,  s = w + beta s,  p = u + beta p,  x += alpha p,  r -= alpha s,  w -= alpha z
// so the sums are POSTED before the product and READ after it (their trip across the chip hides behind ~10 us of matrix
// stream), and the only wait of an iteration is for the m of the few workgroups whose rows this workgroup's columns touch
// (+-5 workgroups on the slab-ordered cube; per-workgroup producer lists built at plan time, "all" as the fall-back).
// Every 30th iteration takes the exact residual r = b - A x as the reference does (CGSolver.cpp:159-166), and with it
// w = A (r / diag), as two more in-launch products.  Measured against the literal solver on the oracle's systems
// (tools/pipelined_pcg_numerics.py): identical iteration counts at 4k..1M tets, solutions equal to 1e-7 (both stop at 1e-6).
//
// Hand-off form (MI355X_MICROARCH.md, "valid forms"): payload sc1 stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> ONE
// sc1 flag store per workgroup; consumer: agent-scope acquire issued early (buffer_inv sc1: nothing of this CU loads the
// planes between it and the barrier), sc1 poll of the producers' flags, s_waitcnt vmcnt(0), workgroup barrier, plain loads.
// The sums travel as tagged 8-byte granules (value half + sequence number in one store: no flag, no ordering).
// Hazards: the planes are double-buffered by publish parity -- a workgroup overwrites buffer P two publishes after it was
// filled, and it can be there only after every consumer of its rows has published in between, i.e. finished its reads
// (consumers of my rows = producers of my columns: A is symmetric, and the wait on them is part of every product); sums are
// double-buffered by sequence parity the same way.  Every spin is bounded by the wall clock; on expiry an error word is
// set, every other spin sees it, the launch drains WITHOUT writing any state back and the host re-solves with the
// two-launch form (fb_step_info.pcg_path says so).
#pragma once
#include "fem_kernels.h"
#include "pcg_pipe_stream.hip.h"

namespace fb {

constexpr int kPipeMaxWaves = 12;    // wavefronts (= slices) per workgroup
constexpr int kPipeMaxBlocks = 256;  // workgroups = CUs
constexpr int kPipeSyncDoubles = 2 * 16 + 2 * kPipeMaxBlocks + 8;  // LDS in front of the resident values: wave sums | gathered sums | broadcast
constexpr int kPipeMaxProducers = 64;
// LDS-resident slots of the matrix: 9 values and the column per lane.  With 32-bit columns a wavefront-slot is 10 x 64 words = 2,560 B and 62 fit
// the CU's 160 KB beside the sync buffers; with 16-bit column differences (C16) it is 9 x 64 words + 64 halfwords = 2,432 B and 65 fit (round 5;
// rounds 3-4: 60 slots of 2,560 B either way -- a streamed slot of a slice costs 1.0 us per iteration at 1M tets).
constexpr int pipe_slot_bytes(bool c16) { return c16 ? 9 * 256 + 128 : 10 * 256; }
constexpr int pipe_lds_slots(bool c16) { return c16 ? 65 : 62; }
static_assert(sizeof(double) * kPipeSyncDoubles + (size_t)pipe_lds_slots(false) * pipe_slot_bytes(false) <= 160 * 1024, "LDS budget of k_pcg_pipe");
static_assert(sizeof(double) * kPipeSyncDoubles + (size_t)pipe_lds_slots(true) * pipe_slot_bytes(true) <= 160 * 1024, "LDS budget of k_pcg_pipe<c16>");

struct PipeArgs {
  unsigned long long* post;   // [2][n_blocks][4] granules: (hi, lo) of gamma, delta, each | sequence << 32
  unsigned int* flags;        // [n_blocks (padded to 4)]: publish number of the workgroup's last complete plane store
  unsigned int* error;        // set on a timed-out wait
  unsigned int* seqs;         // [2] publish / sum sequence numbers reached by the previous launch (block 0 writes them at the end)
  const int* producers;       // [n_blocks][64] workgroups whose rows this workgroup's columns touch; prod_count < 0: poll all
  const int* prod_count;      // [n_blocks]
  const int* prod_xcd;        // [n_blocks] non-zero: a producer (= consumer) of this workgroup sits on another XCD, or it polls all
  int plain_local;            // non-zero: workgroups without such a neighbour publish with plain stores (see publish)
  unsigned int* xcc;          // [n_blocks] (first publish number of the launch) << 4 | HW_REG_XCC_ID of the workgroup: where it REALLY runs
  int prefetch_slots;         // streamed slots per slice whose values the idle wavefronts pull into L2 during the neighbour wait (0..4)
  int service;                // non-zero: the workgroups were launched with one wavefront more than slices; it collects the sums
  int start;                  // 0 continue a solve (state from memory), 1 new solve from x = 0, 2 new solve from the x in memory
  int n_iters;                // at most this many iterations in this launch
  double eps2;                // squared tolerance (start != 0; a continued solve reads the CGState)
  int max_iter;
  long long timeout_ticks;    // wall_clock64 ticks (100 MHz)
  long long* timing;          // development aid (FEMBRAIN_PERSIST_TIMING=1), else null: per wavefront 6 accumulated phase times
  double* planes;             // [2][3][n_pad]: the published vector as the gathers read it, x | y | z planes
  size_t n_pad;               // rows padded to whole slices
  double* pstate;             // [2] gamma_old, alpha_old between the launches of one solve
  // Helpers (round 5): on a mesh with a few very wide slices (hull nodes of a Delaunay mesh: 59 slots against 19 on average) the product
  // of an iteration is as slow as the widest slice of the slowest workgroup -- one wavefront streams a slice's slots one after the other
  // (18 us against 7.5 on average on the 606k-tet probe, profiles/r05_delaunay_phase_table.txt).  Wavefronts without a slice of their own
  // then take the upper part of a wide slice's slots.  tasks[b][wv] of a HELPER = (slice of the workgroup, first slot, end slot, its number
  // in the workgroup = its place in the LDS hand-over), slice -1 = no task; of the wavefront that OWNS a slice = (slots of the slice resident
  // in LDS, where in LDS (in wavefront-slots), end of what it streams itself, bit mask of the helpers whose partial sums it adds -- ascending:
  // a fixed order).  The resident slots of a workgroup are dealt by WIDTH (fem.hip, setup_persist): the slices stream equal numbers of slots
  // as far as the LDS goes, where the plain kernel gives every slice the same share.  nullptr: the plain kernel (every regular mesh).
  const int* wg_first;        // the deal of the slices to the workgroups balanced by slots (pipe_deal), or nullptr: equal numbers of slices
  const int4* tasks;          // [n_blocks][kPipeTaskStride]
  int n_help;                 // most helpers of any workgroup (0: none; LDS for their partial sums is set aside when > 0)
};
constexpr int kPipeTaskStride = 16;
constexpr int kPipeMaxHelpers = 8;   // per workgroup
constexpr int pipe_help_slots(bool c16) { return c16 ? 6 : 5; }  // LDS wavefront-slots set aside for their partial sums (8 x 3 x 64 doubles = 12,288 B, at the end of the slots)
static_assert(pipe_help_slots(true) * pipe_slot_bytes(true) >= kPipeMaxHelpers * 3 * 64 * 8 && pipe_help_slots(false) * pipe_slot_bytes(false) >= kPipeMaxHelpers * 3 * 64 * 8, "helpers' hand-over area");

__device__ __forceinline__ void st_sc1_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_u32(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int ld_sc1_u32(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a node's three doubles (24 consecutive bytes, 8-byte aligned) in two stores instead of three, write-through like st_sc1_f64
__device__ __forceinline__ void st_sc1_xyz(double* p, double x, double y, double z) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 xy = {x, y};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx2 %0, %2, off offset:16 sc1" : : "v"(p), "v"(xy), "v"(z) : "memory");
}
// one 16-byte sc1 load, waited for (a poll of four flags / two granules in one request)
__device__ __forceinline__ uint4 ld_sc1_u128(const void* p) {
  uint4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// a value every lane holds alike, moved to scalar registers (a uniform double in vector registers costs two per lane for nothing)
__device__ __forceinline__ double uniform_f64(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)(b & 0xffffffffLL)), hi = __builtin_amdgcn_readfirstlane((unsigned int)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ bool uniform_flag(bool f) { return __builtin_amdgcn_readfirstlane(f ? 1 : 0) != 0; }

// slices of workgroup b: XCD b & 7 keeps the contiguous eighth of the rows it has in every other FEM kernel (SliceWalk); its
// n_blocks/8 workgroups share that slab as evenly as whole slices allow.  Host and device use the same function.
__host__ __device__ inline void pipe_slices(int n_slices, int n_blocks, int b, int* first, int* count) {
  const int xcd = b & 7, j = b >> 3, per = n_blocks >> 3;
  const int chunk = (n_slices + 7) >> 3;
  const int lo = xcd * chunk;
  int len = n_slices - lo;
  len = len < 0 ? 0 : (len > chunk ? chunk : len);
  const int base = len / per, rem = len - base * per;
  *first = lo + j * base + (j < rem ? j : rem);
  *count = base + (j < rem ? 1 : 0);
}

// ... or, where the slices of a mesh differ much in width (hull nodes of a Delaunay mesh: 59 slots against 19 on average), from a table
// the host has balanced by SLOTS (fem.hip setup_persist): a CU's product takes as long as the slots it streams, whoever streams them --
// the workgroups of the hull took 18 us per product against 7.5 on average with an equal number of slices each (profiles/r05_delaunay_phase_table.txt).
// wg_first: n_blocks + 1 first slices, ascending inside every XCD's share; nullptr: the formula above.
__host__ __device__ inline void pipe_deal(const int* wg_first, int n_slices, int n_blocks, int b, int* first, int* count) {
  if (wg_first) { *first = wg_first[b]; *count = wg_first[n_blocks + 1 + b]; }
  else pipe_slices(n_slices, n_blocks, b, first, count);
}

// The workgroups whose rows the columns of a slice lie in: a bit mask over the (at most kPipeMaxBlocks = 256) workgroups, 8 words per
// slice.  EXACT, where a column range per slice (rounds 2-3) gave a superset: one element that joins two distant nodes (a sliver on the hull of a Delaunay
// mesh, a cut that was closed again) widens the range of its slice to "everyone" but adds ONE producer to the exact list.
__global__ __launch_bounds__(kBlock) void k_slice_producers(int n_slices, int n_owned, const int* __restrict__ slice_off, const int* __restrict__ colidx,
                                                            const int* __restrict__ slice_owner, unsigned int* __restrict__ out) {
  const int s = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int row = s * 64 + lane;
  unsigned int m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  if (row < n_owned) {
    // four slots at a time, their columns and then their owners in flight together (one slot after the other was a chain of two dependent
    // loads per slot: 38-44 us of every re-sync at 1.1M tets)
    const int k1 = slice_off[s + 1];
    for (int k0 = slice_off[s]; k0 < k1; k0 += 4) {
      int c[4], o[4];
#pragma unroll
      for (int j = 0; j < 4; j++) c[j] = k0 + j < k1 ? colidx[(size_t)(k0 + j) * 64 + lane] : n_owned;
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = c[j] < n_owned ? slice_owner[c[j] >> 6] : -1;  // (>= n_owned: a shard's halo column, the proxies' business)
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (o[j] >= 0) {
#pragma unroll
          for (int i = 0; i < 8; i++) m[i] |= (o[j] >> 5) == i ? 1u << (o[j] & 31) : 0u;
        }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[i] |= __shfl_xor(m[i], off, 64);
    if (lane == 0) out[(size_t)s * 8 + i] = m[i];
  }
}

// owner[slice] = the workgroup pipe_slices gives it to (a thread per workgroup)
__global__ __launch_bounds__(kBlock) void k_slice_owner(int n_slices, int nb, const int* __restrict__ wg_first, int* __restrict__ owner) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nb) return;
  int first, count;
  pipe_deal(wg_first, n_slices, nb, b, &first, &count);
  for (int k = 0; k < count; k++) owner[first + k] = b;
}

// The producer list of every workgroup from the masks of its slices (a thread per workgroup; round 4: on the device, the host loop
// with its download and three uploads was 0.2 ms of every re-sync): prod[b][0..cnt[b]) ascending, cnt[b] = -1 (poll everyone) beyond
// kPipeMaxProducers, far[b] = a producer sits on another XCD, stats[0] = longest list, stats[1] = some workgroup polls everyone.
__global__ __launch_bounds__(kBlock) void k_wg_producers(int n_slices, int nb, const int* __restrict__ wg_first, const unsigned int* __restrict__ mask, int poll_all,
                                                         int* __restrict__ prod, int* __restrict__ cnt, int* __restrict__ far, int* __restrict__ stats) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nb) return;
  int first, count;
  pipe_deal(wg_first, n_slices, nb, b, &first, &count);
  unsigned int m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  for (int k = 0; k < count; k++)
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] |= mask[(size_t)(first + k) * 8 + i];
#pragma unroll
  for (int i = 0; i < 8; i++)
    if ((b >> 5) == i) m[i] &= ~(1u << (b & 31));
  int n = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) n += __popc(m[i]);
  int* mine = prod + (size_t)b * kPipeMaxProducers;
  if (n > kPipeMaxProducers || poll_all) {
    for (int k = 0; k < kPipeMaxProducers; k++) mine[k] = -1;
    cnt[b] = -1; far[b] = 1;
    atomicOr(&stats[1], 1);
    return;
  }
  int k = 0, f = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    unsigned int w = m[i];
    while (w) {
      const int o = 32 * i + __ffs((int)w) - 1;
      w &= w - 1u;
      mine[k++] = o;
      if ((o & 7) != (b & 7)) f = 1;
    }
  }
  for (; k < kPipeMaxProducers; k++) mine[k] = -1;
  cnt[b] = n; far[b] = f;
  atomicMax(&stats[0], n);
}

// One wavefront's share of the collection of all workgroups' posted sums of sequence number `sums`: lane `l0` of `stride` takes workgroups
// l0, l0 + stride, ...; adds their two values to t0s / t1s in that order (the callers fix the order of the rest).  Bounded by the wall clock.
__device__ __forceinline__ void pipe_collect_posts(const PipeArgs& pa, unsigned int sums, int nb, int lane, int stride, int l0, long long t0, long long t_limit, bool& failed,
                                                   double& t0s, double& t1s) {
  const unsigned long long* post = pa.post + (size_t)(sums & 1u) * nb * 4;
  for (int b = l0; b - lane < nb && !failed; b += stride) {  // wave-uniform trip count
    const bool mine = b < nb;
    uint4 q4[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    for (;;) {
      bool ok = true;
      if (mine) {  // the record's four granules in two 16-byte requests (each granule is one 8-byte store of its writer)
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(q4[0]), "=&v"(q4[1]) : "v"(post + (size_t)b * 4) : "memory");
        ok = q4[0].y == sums && q4[0].w == sums && q4[1].y == sums && q4[1].w == sums;
      }
      if (__ballot(!ok) == 0ULL) break;
      if (ld_sc1_u32(pa.error) != 0u || wall_clock64() - t0 > t_limit) { failed = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (mine && !failed) {
      t0s += __longlong_as_double((long long)(((unsigned long long)q4[0].x << 32) | (unsigned long long)q4[0].z));
      t1s += __longlong_as_double((long long)(((unsigned long long)q4[1].x << 32) | (unsigned long long)q4[1].z));
    }
  }
}

// How many cache lines the gathers of a slot touch, planes against node-by-node storage of the gathered vector (sampled: every `step`-th
// slice, all its slots; a wavefront per slot).  out[0] += distinct 128-byte lines of one plane (x 3 planes = the lines a slot fetches),
// out[1] += distinct lines of the 24-byte records, out[2] += 1.
__global__ __launch_bounds__(256) void k_gather_lines(int n_slices, int step, const int* __restrict__ slice_off, const int* __restrict__ colidx,
                                                      unsigned long long* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  unsigned long long planes = 0, records = 0, slots = 0;
  for (int s = wave * step; s < n_slices; s += n_waves * step) {
    for (int slot = slice_off[s]; slot < slice_off[s + 1]; slot++) {
      const unsigned int col = (unsigned int)colidx[(size_t)slot * 64 + lane];
      const unsigned int kp = col >> 4, k0 = (col * 24u) >> 7, k1 = (col * 24u + 23u) >> 7;
      bool first_p = true, first_0 = true, first_1 = k1 != k0;
      for (int l = 0; l < 64; l++) {  // (is there an earlier lane with the same line?)
        const unsigned int op = __builtin_amdgcn_readlane(kp, l), o0 = __builtin_amdgcn_readlane(k0, l), o1 = __builtin_amdgcn_readlane(k1, l);
        if (l < lane) {
          first_p = first_p && op != kp;
          first_0 = first_0 && o0 != k0 && o1 != k0;
          first_1 = first_1 && o0 != k1 && o1 != k1;
        }
      }
      planes += __popcll(__ballot(first_p));
      records += __popcll(__ballot(first_0)) + __popcll(__ballot(first_1));
      slots++;
    }
  }
  if (lane == 0 && slots) { atomicAdd(out, planes); atomicAdd(out + 1, records); atomicAdd(out + 2, slots); }
}

}  // namespace fb
#include "pcg_shard_box.hip.h"
namespace fb {

// WMAX: wavefronts per workgroup the instantiation is bounded for (512 registers per lane and SIMD are shared by
// ceil(WMAX / 4) wavefronts).  LDS-resident part of the matrix: the first slots of every slice (9 values + the column id per lane)
// are loaded into LDS ONCE per launch -- the matrix does not change during a solve; slots beyond are streamed every product as in
// k_spmv.  The CU's LDS holds pipe_lds_slots() = 62 / 65 wavefront-slots beside the sync buffers; a workgroup deals them
// to its `count` live wavefronts, count-th part each and the remainder one more for the first ones (11 slices: 5 slots each and a
// sixth for seven of them; 10 slices: 6 each), at most KLT per wavefront (the unroll bound of the LDS loop).
// TIMING: the development build with per-phase clocks (FEMBRAIN_PERSIST_TIMING=1).
// SHARD: the kernel of a sharded handle (pcg_shard_box.hip.h; "k_pcg_pipe_shard<WMAX,KLT>" in fb_fem_pcg_path): the slices come from the
// plan's deal (sa.wg_range), the last wavefront is the spare one (sums, counters, proxy copies), rows a neighbour rank gathers are also
// stored into its box, the sums go through the rank level; 32-bit column words, write-through stores.  sa is not read otherwise.
// BJ (FB_PCG_BLOCK_JACOBI, opt-in and not part of the reference): the preconditioner is the inverse of the row's 3x3 diagonal block
// instead of 1/diag -- `invdiag` then points at 9 doubles per row (symmetric: 6 are loaded).  The Jacobi instantiations do not change.
// HELP: the instantiation with helper wavefronts (PipeArgs::tasks).  A template parameter, not a run-time test: with the helpers' second
// copy of the streamed product compiled in, the kernel of the headline mesh -- which has none -- ran 3 % slower (16.2 against 15.7 us per
// iteration on one box, tools/ab_r4 in round 5); with HELP = false nothing of it is there.
template <typename MT, bool C16, int WMAX, int KLT, bool TIMING, bool SHARD, bool BJ = false, bool HELP = false, bool XYZ = false>
__global__ __launch_bounds__(64 * WMAX) void k_pcg_pipe(SellView sv, const MT* __restrict__ vals, const MT* __restrict__ dlo,
                                                        const double* __restrict__ invdiag, const double* __restrict__ bvec, double* __restrict__ xg,
                                                        double* __restrict__ rg, double* __restrict__ wg, double* __restrict__ zg,
                                                        double* __restrict__ sg, double* __restrict__ pg, CGState* __restrict__ st, PipeArgs pa,
                                                        ShardArgs sa) {
  static_assert(!SHARD || (!C16 && !TIMING), "a shard's columns are 32-bit local ids; the phase clocks are built for the unsharded kernel");
  static_assert(!BJ || (!SHARD && !TIMING), "block-Jacobi is for unsharded handles");
  static_assert(!HELP || (!SHARD && !BJ), "helper wavefronts: unsharded handles, the Jacobi preconditioner");
  static_assert(sizeof(MT) == 4, "k_pcg_pipe keeps part of the matrix in LDS as fp32 words and streams the rest as fp32");
  extern __shared__ double lds[];  // the request is padded so that one workgroup fills a CU
  double* wsum = lds;                          // [2][16] wave sums
  double* gath = lds + 32;                     // [2][kPipeMaxBlocks] all workgroups' sums
  double* bc = gath + 2 * kPipeMaxBlocks;      // [0..1] totals, [2] a wait failed (sweep), [3] a wait failed (product), [4] a sum poller gave up
  const int n_waves = blockDim.x >> 6, nb = gridDim.x;
  if (pa.start == 0 && st->done) return;  // grid-uniform: written by an earlier launch
  if (threadIdx.x == 0) bc[4] = 0.0;      // set by a sum poller that gave up (read after the next barrier)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int first, count;
  if constexpr (SHARD) { first = sa.wg_range[blockIdx.x].x; count = sa.wg_range[blockIdx.x].y; }
  else pipe_deal(pa.wg_first, sv.n_slices, nb, blockIdx.x, &first, &count);
  const bool spare = SHARD && wv == n_waves - 1;  // a shard's spare wavefront: sums, counters, proxy copies (it owns no slice)
  const bool live = wv < count && !spare;         // wave-uniform
  const int sl = first + wv;
  const int row = sl * 64 + lane;
  const bool rvalid = live && row < sv.n_owned;
  const size_t dof = 3 * (size_t)(rvalid ? row : 0);
  int so = 0, width = 0;
  if (live) { so = sv.slice_off[sl]; width = sv.slice_off[sl + 1] - so; }
  so = __builtin_amdgcn_readfirstlane(so); width = __builtin_amdgcn_readfirstlane(width);  // wave-uniform: scalar registers
  const MT* v = vals + (size_t)so * 9 * 64 + lane;
  const int* ci = sv.colidx + (size_t)so * 64 + lane;
  const short* cd = C16 ? sv.coldelta + (size_t)so * 64 + lane : nullptr;
  // low part of the diagonal block (symmetric: 6 planes) and 1/diag, fixed for the solve
  MT m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;
  double iv[BJ ? 6 : 3] = {0};  // 1/diag, or (BJ) the inverse diagonal block: 00 01 02 11 12 22
  int send_beg = 0, send_end = 0;  // (SHARD) this row's entries of the send lists
  if (rvalid) {
    const MT* l = dlo + (size_t)sl * 9 * 64 + lane;
    m00 = l[0 * 64]; m01 = l[1 * 64]; m02 = l[2 * 64]; m11 = l[4 * 64]; m12 = l[5 * 64]; m22 = l[8 * 64];
    if constexpr (BJ) {
      const double* B = invdiag + 3 * dof;
      iv[0] = B[0]; iv[1] = B[1]; iv[2] = B[2]; iv[3] = B[4]; iv[4] = B[5]; iv[5] = B[8];
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++) iv[a] = invdiag[dof + a];
    }
    if constexpr (SHARD) { send_beg = sa.row_send_off[row]; send_end = sa.row_send_off[row + 1]; }
  }
  // LDS-resident part of the matrix: the first KL slots of this wave's slice, [klt_w][10][64] words (9 values + the column id)
  // (HELP: how many and where is the plan's decision, PipeArgs::tasks -- a slice of 27 slots among slices of 15 keeps 17 of them here, so
  // that every wavefront of the workgroup streams about the same number: an iteration ends with the slowest wavefront)
  int klt_w, lres_at;
  if constexpr (HELP) {
    const int4 tk = pa.tasks[(size_t)blockIdx.x * kPipeTaskStride + wv];
    klt_w = __builtin_amdgcn_readfirstlane(live ? tk.x : 0);
    lres_at = __builtin_amdgcn_readfirstlane(live ? tk.y : 0);
  } else {
    constexpr int kSlots = pipe_lds_slots(C16);
    const int lbase = min(KLT, kSlots / max(count, 1)), lrem = lbase < KLT ? min(count, kSlots - lbase * count) : 0;  // workgroup-uniform
    klt_w = __builtin_amdgcn_readfirstlane(live ? lbase + (wv < lrem ? 1 : 0) : 0);
    lres_at = wv * lbase + min(wv, lrem);
  }
  const int KL = min(klt_w, width);
  // this wavefront's part: C16 [klt_w][9][64] value words, then [klt_w][64] column differences (halfwords); else [klt_w][10][64] words, the tenth the column
  constexpr int kValWords = C16 ? 9 : 10;
  char* lwave = (char*)(lds + kPipeSyncDoubles) + (size_t)lres_at * pipe_slot_bytes(C16);
  unsigned int* lres = (unsigned int*)lwave + lane;
  short* lcd = (short*)(lwave + (size_t)klt_w * 9 * 256) + lane;  // (C16)
  auto lds_col = [&](int k) -> unsigned int {
    if constexpr (C16) return (unsigned int)(row + (int)lcd[k * 64]);
    else return lres[(k * 10 + 9) * 64];
  };
  if (sizeof(MT) == 4) {
    for (int k = 0; k < KL; k++) {
      const MT* vk = v + (size_t)k * 9 * 64;
#pragma unroll
      for (int j = 0; j < 9; j++) lres[(k * kValWords + j) * 64] = __float_as_uint((float)vk[j * 64]);
      if constexpr (C16) lcd[k * 64] = cd[(size_t)k * 64];
      else lres[(k * 10 + 9) * 64] = (unsigned int)ci[(size_t)k * 64];
    }
  }
  // this wavefront's task (helpers, see PipeArgs): the owner of a slice streams its slots up to own_k1 and adds the partial sums of the
  // helpers in help_mask; a helper streams [hk0, hk1) of slice help_sl for the rows of that slice
  // (every value below is wave-uniform and made so explicitly, outside any branch: scalar registers.  Without HELP they are constants and the
  // kernel is the round-4 one.)
  double* ypart = (double*)((char*)(lds + kPipeSyncDoubles) + (size_t)(pipe_lds_slots(C16) - pipe_help_slots(C16)) * pipe_slot_bytes(C16));  // [helper][3][64]
  int own_k1 = width, help_sl = -1, hk0 = 0, hk1 = 0, help_idx = 0, help_so = 0;
  unsigned int help_mask = 0u;
  if constexpr (HELP) {
    const int4 tk = pa.tasks[(size_t)blockIdx.x * kPipeTaskStride + wv];
    const bool helper = !live && !spare && tk.x >= 0;
    own_k1 = __builtin_amdgcn_readfirstlane(live ? tk.z : width);
    help_mask = (unsigned int)__builtin_amdgcn_readfirstlane(live ? tk.w : 0);
    help_sl = __builtin_amdgcn_readfirstlane(helper ? first + tk.x : -1);
    hk0 = __builtin_amdgcn_readfirstlane(helper ? tk.y : 0); hk1 = __builtin_amdgcn_readfirstlane(helper ? tk.z : 0);
    help_idx = __builtin_amdgcn_readfirstlane(helper ? tk.w : 0);
    help_so = __builtin_amdgcn_readfirstlane(sv.slice_off[help_sl < 0 ? 0 : help_sl]);
  }
  // the producers of this workgroup's columns: one per lane of wavefront 0
  const int n_prod = pa.prod_count[blockIdx.x];
  int my_prod = -1;
  if (wv == 0 && n_prod >= 0 && lane < n_prod) my_prod = pa.producers[(size_t)blockIdx.x * kPipeMaxProducers + lane];

  unsigned int send_mask = 0u;  // (SHARD) workgroup-uniform: the ranks this workgroup has rows to send to
  ShardBoxLayout BL = {};
  if constexpr (SHARD) { send_mask = sa.wg_send_mask[blockIdx.x]; BL = shard_box_layout(sa.halo_cap); }
  unsigned int pub = pa.seqs[0], sums = pa.seqs[1];  // grid-uniform: written by the previous launch
  const long long t_limit = pa.timeout_ticks;
  bool failed = false;
  long long tm[TIMING ? 6 : 1] = {0}, tprev = TIMING ? wall_clock64() : 0;  // (TIMING: the phase clocks cost 14 registers per lane)
  auto lap = [&](int k) { if (TIMING) { const long long t = wall_clock64(); tm[k] += t - tprev; tprev = t; } };

  // Stores of the next product's input vector into the plane buffer of the next publish number (512 contiguous bytes per wave
  // and plane).  Write-through (sc1) where a workgroup of ANOTHER XCD gathers these rows; a workgroup all of whose consumers
  // share its XCD (= its L2) may store plainly (pa.plain_local): the line then stays in that L2, the store is acknowledged
  // there, and the consumers -- their L1 invalidated -- read it from there.  Which workgroups share an XCD is NOT the launch's to
  // promise (blockIdx & 7 is the observed round-robin deal, good for speed only; a CU-masked stream deals otherwise): every
  // workgroup announces the XCC id the hardware reports, its first publish of a launch goes through, and it stores plainly from
  // the second on only if all its producers (= its consumers: A is symmetric) announced the same id (checked in product()).
  const bool plain_cand = !SHARD && pa.plain_local != 0 && pa.prod_xcd[blockIdx.x] == 0;  // workgroup-uniform
  bool through = true, xcc_known = false;
  unsigned int my_xcc = 0u;
  if constexpr (!SHARD) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
    if (threadIdx.x == 0) st_sc1_u32(pa.xcc + blockIdx.x, ((pub + 1u) << 4) | my_xcc);  // (visible before this workgroup's first flag: product() drains before it flags)
  }
  // XYZ: the published vector lies node by node (x, y, z side by side) instead of in three planes -- for irregular meshes, whose gathers
  // of a slot touch 64 different cache lines per plane: one line per lane instead of three (pcg_pipe_stream.hip.h).  606k-tet Delaunay
  // probe 18.6 -> 16.4 us per iteration; the cube after a cut, whose columns are mostly consecutive, 19.4 -> 20.6 (each of the three loads
  // of a slot then touches 12 lines instead of 4), so setup_persist decides from the columns (k_gather_lines).
  static_assert(!XYZ || HELP, "the node-by-node vector comes with the table-driven instantiation");
  const size_t xs = XYZ ? 1 : pa.n_pad, cs = XYZ ? 3 : 1;  // strides of a component and of a column in the published vector
  auto publish = [&](const double* vin) {
    pub++;
    double* pl = pa.planes + (size_t)(pub & 1u) * 3 * pa.n_pad;
    if (rvalid) {
      if (XYZ && through) {
        st_sc1_xyz(pl + 3 * (size_t)row, vin[0], vin[1], vin[2]);
      } else if (through) {
#pragma unroll
        for (int a = 0; a < 3; a++) st_sc1_f64(pl + a * xs + cs * (size_t)row, vin[a]);
      } else {
#pragma unroll
        for (int a = 0; a < 3; a++) pl[a * xs + cs * (size_t)row] = vin[a];
      }
      if constexpr (SHARD) shard_send_row(sa, BL, pub, send_beg, send_end, vin);
    }
  };
  // y = A vin for the vector published last: drain the stores, flag, wait for the producers of this workgroup's columns,
  // multiply.  post_sums: the two wave sums in wsum[.][wv] are this iteration's local parts of gamma and delta -- posted with
  // the flag, read after the product.
  auto product = [&](const double* vin, double* y, bool post_sums, bool late_acquire) {
    double* pl = pa.planes + (size_t)(pub & 1u) * 3 * pa.n_pad;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lap(0);  // publish, drained
    if (post_sums) sums++;
    if (wv != 0 && live && pa.prefetch_slots > 0 && own_k1 > klt_w) {
      // idle until wavefront 0 has seen the neighbours' flags: the first streamed slots' values go to L2 meanwhile (measured at 1M
      // tets, us per iteration with 0 / 2 / 3 / 4 slots: 17.15 / 16.25 / 16.1 / 16.05; the same during the drain of the publish
      // stores instead delays the flag and loses: 16.85)
      int so_k = so + klt_w;
      asm volatile("" : "+s"(so_k));  // (opaque, as for the streamed loop below)
      pipe_prefetch_values(min(pa.prefetch_slots, own_k1 - klt_w), ((unsigned int)so_k * 9u * 64u + (unsigned int)lane) * (unsigned int)sizeof(float), vals);
    }
    if constexpr (SHARD) {
      if (spare) shard_service_product(sa, BL, pa, pub, pl, nb, lane, send_mask, t_limit, bc, failed);
    }
    if (wv == 0) {
      if (lane == 0) st_sc1_u32(pa.flags + blockIdx.x, pub);
      if (post_sums && lane < 2) {  // workgroup sums in wave order, posted as two tagged halves each
        double t = 0.0;
        for (int w = 0; w < n_waves; w++) t += wsum[lane * 16 + w];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(t), tag = (unsigned long long)sums << 32;
        unsigned long long* post = pa.post + ((size_t)(sums & 1u) * nb + blockIdx.x) * 4 + 2 * lane;
        st_sc1_u64(post, (bits >> 32) | tag);
        st_sc1_u64(post + 1, (bits & 0xffffffffULL) | tag);
      }
      // Every wave of this workgroup is past its last gather from the buffer published now (two products ago) and none loads
      // from the planes before the barrier below: drop this CU's L1 lines NOW (asynchronous), the poll runs meanwhile.  That early
      // form rests on the sums: every workgroup has posted this iteration's sums, i.e. finished the product before the previous one, so
      // no CU of this XCD pulls a line of this buffer's previous contents into the shared L2 any more.  Products that do not follow a
      // sums sweep (the first of a launch, the exact-residual pair, the iteration after them) have no such guarantee about a sibling
      // CU two products behind -- they take the guide's order, acquire AFTER the poll.
      if (!late_acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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
        for (int b = 4 * lane; b - 4 * lane < nb && !failed; b += 256) {  // four flags per lane in one 16-byte request
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
      if (plain_cand && !xcc_known && !failed) {  // first product of the launch: the producers' flags are in, so are their XCC ids
        const bool same = my_prod < 0 || ld_sc1_u32(pa.xcc + my_prod) == ((pub << 4) | my_xcc);
        const bool all_same = __ballot(!same) == 0ULL;
        if (lane == 0) bc[5] = all_same ? 1.0 : 0.0;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) bc[3] = (failed || ld_sc1_u32(pa.error) != 0u) ? 1.0 : 0.0;  // one lane decides for the whole workgroup
    }
    __syncthreads();
    lap(1);  // flag + wait for the producers + acquire
    if (uniform_flag(bc[3] != 0.0 || (SHARD && bc[4] != 0.0))) { failed = true; return; }
    if (plain_cand && !xcc_known) { through = !uniform_flag(bc[5] != 0.0); xcc_known = true; }
    // the low part of the diagonal block times the own entry first: vin is not needed beyond this point
    double y0 = 0, y1 = 0, y2 = 0;
    if (rvalid) {
      MT u00 = m00, u01 = m01, u02 = m02, u11 = m11, u12 = m12, u22 = m22;  // kept in storage type: without the (empty) asm the
      asm volatile("" : "+v"(u00), "+v"(u01), "+v"(u02), "+v"(u11), "+v"(u12), "+v"(u22));  // compiler hoists the conversions and keeps 12 registers
      const double l00 = (double)u00, l01 = (double)u01, l02 = (double)u02, l11 = (double)u11, l12 = (double)u12, l22 = (double)u22;
      y0 = l00 * vin[0] + l01 * vin[1] + l02 * vin[2];
      y1 = l01 * vin[0] + l11 * vin[1] + l12 * vin[2];
      y2 = l02 * vin[0] + l12 * vin[1] + l22 * vin[2];
    }
    if (live) {
      if (sizeof(MT) == 4 && KLT >= 16) {
        // whole slices in LDS, five wavefronts per CU: registers to spare, so ALL gathers of the product are in flight before the first
        // multiplication (the loop below waits for each slot's three gathers in turn: 5.5 us for the slowest wavefronts at 105k tets, whose
        // columns come from another XCD's rows, against 2.2 on average -- profiles/r05_small_mesh_phase_table.txt)
        double gx[KLT][3];
#pragma unroll
        for (int k = 0; k < KLT; k++) {
          const unsigned int col = k < KL ? lds_col(k) : 0u;
          const double* xp = pl + cs * (size_t)col;
          gx[k][0] = k < KL ? xp[0] : 0.0; gx[k][1] = k < KL ? xp[xs] : 0.0; gx[k][2] = k < KL ? xp[2 * xs] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < KLT; k++) if (k < KL) {
          const unsigned int* lk = lres + (size_t)k * kValWords * 64;
          const double x0 = gx[k][0], x1 = gx[k][1], x2 = gx[k][2];
          y0 += (double)__uint_as_float(lk[0 * 64]) * x0 + (double)__uint_as_float(lk[1 * 64]) * x1 + (double)__uint_as_float(lk[2 * 64]) * x2;
          y1 += (double)__uint_as_float(lk[3 * 64]) * x0 + (double)__uint_as_float(lk[4 * 64]) * x1 + (double)__uint_as_float(lk[5 * 64]) * x2;
          y2 += (double)__uint_as_float(lk[6 * 64]) * x0 + (double)__uint_as_float(lk[7 * 64]) * x1 + (double)__uint_as_float(lk[8 * 64]) * x2;
        }
      } else if (sizeof(MT) == 4) {
#pragma unroll
        for (int k = 0; k < KLT; k++) if (k < KL) {  // LDS-resident slots
          const unsigned int* lk = lres + (size_t)k * kValWords * 64;
          const double* xp = pl + cs * (size_t)lds_col(k);
          const double x0 = xp[0], x1 = xp[xs], x2 = xp[2 * xs];
          y0 += (double)__uint_as_float(lk[0 * 64]) * x0 + (double)__uint_as_float(lk[1 * 64]) * x1 + (double)__uint_as_float(lk[2 * 64]) * x2;
          y1 += (double)__uint_as_float(lk[3 * 64]) * x0 + (double)__uint_as_float(lk[4 * 64]) * x1 + (double)__uint_as_float(lk[5 * 64]) * x2;
          y2 += (double)__uint_as_float(lk[6 * 64]) * x0 + (double)__uint_as_float(lk[7 * 64]) * x1 + (double)__uint_as_float(lk[8 * 64]) * x2;
        }
        if constexpr (HELP) {  // (a wide slice's share may be longer than the unroll bound: further groups of KLT, the gathers of a group in flight together)
          for (int kb = KLT; kb < KL; kb += KLT) {
            double gx[KLT][3];
#pragma unroll
            for (int j = 0; j < KLT; j++) {
              const int k = min(kb + j, KL - 1);  // (past the end: the last slot again, not used)
              const double* xp = pl + cs * (size_t)lds_col(k);
              gx[j][0] = xp[0]; gx[j][1] = xp[xs]; gx[j][2] = xp[2 * xs];
            }
#pragma unroll
            for (int j = 0; j < KLT; j++) if (kb + j < KL) {
              const unsigned int* lk = lres + (size_t)(kb + j) * kValWords * 64;
              const double x0 = gx[j][0], x1 = gx[j][1], x2 = gx[j][2];
              y0 += (double)__uint_as_float(lk[0 * 64]) * x0 + (double)__uint_as_float(lk[1 * 64]) * x1 + (double)__uint_as_float(lk[2 * 64]) * x2;
              y1 += (double)__uint_as_float(lk[3 * 64]) * x0 + (double)__uint_as_float(lk[4 * 64]) * x1 + (double)__uint_as_float(lk[5 * 64]) * x2;
              y2 += (double)__uint_as_float(lk[6 * 64]) * x0 + (double)__uint_as_float(lk[7 * 64]) * x1 + (double)__uint_as_float(lk[8 * 64]) * x2;
            }
          }
        }
      }
      // the streamed slots: hand-pipelined loads (pcg_pipe_stream.hip.h); those from own_k1 on are a helper's
      const int n_str = own_k1 - klt_w;
      if (n_str > 0) {
        int so_k = so + klt_w;
        asm volatile("" : "+s"(so_k));  // opaque: keeps the two offsets below from being hoisted out of the solver loop into live registers
        pipe_stream_slots<C16, XYZ>(n_str, ((unsigned int)so_k * 9u * 64u + (unsigned int)lane) * (unsigned int)sizeof(float),
                               ((unsigned int)so_k * 64u + (unsigned int)lane) * (unsigned int)(C16 ? sizeof(short) : sizeof(int)), vals,
                               C16 ? (const void*)sv.coldelta : (const void*)sv.colidx, pl, pl + xs, pl + 2 * xs, row, y0, y1, y2);
      }
    }
    if constexpr (HELP) if (help_sl >= 0) {  // a helper: its share of another wavefront's slice, handed over through LDS (the owner adds it after the next barrier)
      double h0 = 0, h1 = 0, h2 = 0;
      int so_k = help_so + hk0;
      asm volatile("" : "+s"(so_k));
      if (hk1 > hk0)  // (the stream loads its first slots unconditionally: never with none)
      pipe_stream_slots<C16, XYZ>(hk1 - hk0, ((unsigned int)so_k * 9u * 64u + (unsigned int)lane) * (unsigned int)sizeof(float),
                             ((unsigned int)so_k * 64u + (unsigned int)lane) * (unsigned int)(C16 ? sizeof(short) : sizeof(int)), vals,
                             C16 ? (const void*)sv.coldelta : (const void*)sv.colidx, pl, pl + xs, pl + 2 * xs, help_sl * 64 + lane, h0, h1, h2);
      double* hp = ypart + (size_t)help_idx * 3 * 64 + lane;
      hp[0] = h0; hp[64] = h1; hp[128] = h2;
    }
    y[0] = y0; y[1] = y1; y[2] = y2;
    lap(2);  // product
  };
  // the helpers' partial sums of this wavefront's slice, in the order of their numbers (after a workgroup barrier that follows the product)
  auto add_helpers = [&](double* y) {
    unsigned int m = help_mask;
    while (m) {
      const int hi = __ffs((int)m) - 1;
      m &= m - 1u;
      const double* hp = ypart + (size_t)hi * 3 * 64 + lane;
      y[0] += hp[0]; y[1] += hp[64]; y[2] += hp[128];
    }
  };

  // this row's vectors stay here for the whole solve
  double xr[3] = {0, 0, 0}, rr[3] = {0, 0, 0}, wr[3] = {0, 0, 0}, zr[3] = {0, 0, 0}, sr[3] = {0, 0, 0}, pr[3] = {0, 0, 0};
  double rho0 = 0.0, eps2 = pa.eps2, gamma_old = 1.0, alpha_old = 1.0;
  int iter = 0, max_iter = pa.max_iter;
  // One product per trip of the loop below (ONE copy of the product's code): what is multiplied and what becomes of the result
  // depends on the phase.  WARM_X / REFRESH_X: y = A x, r = b - y (CGSolver.cpp:131-136 / :159-166); INIT_W / REFRESH_W:
  // w = A (r / diag); ITER: an iteration.
  enum { PH_WARM_X = 0, PH_INIT_W = 1, PH_ITER = 2, PH_REFRESH_X = 3, PH_REFRESH_W = 4 };
  int phase = PH_ITER;
  bool fresh = pa.start != 0;  // the next iteration is the first of a solve (beta = 0)
  if (pa.start == 0) {
    if (rvalid) {
#pragma unroll
      for (int a = 0; a < 3; a++) { xr[a] = xg[dof + a]; rr[a] = rg[dof + a]; wr[a] = wg[dof + a]; zr[a] = zg[dof + a]; sr[a] = sg[dof + a]; pr[a] = pg[dof + a]; }
    }
    rho0 = uniform_f64(st->rho0); eps2 = uniform_f64(st->eps2); iter = st->iter; max_iter = st->max_iter;
    gamma_old = uniform_f64(pa.pstate[0]); alpha_old = uniform_f64(pa.pstate[1]);
  } else if (pa.start == 2) {
    if (rvalid) {
#pragma unroll
      for (int a = 0; a < 3; a++) xr[a] = xg[dof + a];
    }
    phase = PH_WARM_X;
  } else {
    if (rvalid) {  // x = 0: r = b
#pragma unroll
      for (int a = 0; a < 3; a++) rr[a] = bvec[dof + a];
    }
    phase = PH_INIT_W;
  }

  // out = M^-1 v: v / diag, or (BJ) the inverse diagonal block times v
  auto precond = [&](const double* vv, double* out) {
    if constexpr (BJ) {
      out[0] = iv[0] * vv[0] + iv[1] * vv[1] + iv[2] * vv[2];
      out[1] = iv[1] * vv[0] + iv[3] * vv[1] + iv[4] * vv[2];
      out[2] = iv[2] * vv[0] + iv[4] * vv[1] + iv[5] * vv[2];
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++) out[a] = iv[a] * vv[a];
    }
  };
  bool done = false, published = false;  // published: the stores of the next product's input are on their way already
  bool unsettled = true;                 // the next product does not follow a sums sweep of this launch: acquire after the poll
  double gamma = 0.0;
  int it_done = 0;
  while (!failed) {
    if (phase == PH_ITER && it_done >= pa.n_iters) break;  // a launch is cut between iterations only
    double vin[3];
    if (phase == PH_WARM_X || phase == PH_REFRESH_X) {
#pragma unroll
      for (int a = 0; a < 3; a++) vin[a] = xr[a];
    } else if (phase == PH_ITER) {
      precond(wr, vin);  // m = w / diag
    } else {
      precond(rr, vin);
    }
    if (!published) publish(vin);
    published = false;
    if (phase == PH_ITER) {
      // local parts of gamma = r . u and delta = w . u (u = r / diag), while the stores drain
      double u[3];
      precond(rr, u);
      double a0 = rr[0] * u[0] + rr[1] * u[1] + rr[2] * u[2];
      double a1 = wr[0] * u[0] + wr[1] * u[1] + wr[2] * u[2];
      a0 = wave_sum(a0); a1 = wave_sum(a1);
      int wvo = wv;
      asm volatile("" : "+v"(wvo));  // opaque: the LDS address is formed here instead of living in a register through the solve
      if (lane == 0) { wsum[wvo] = a0; wsum[16 + wvo] = a1; }
    }
    double y[3];
    product(vin, y, phase == PH_ITER, phase != PH_ITER || unsettled);
    unsettled = phase != PH_ITER;  // (an iteration's sums sweep settles the next product's early acquire)
    if (failed) break;
    if constexpr (HELP) if (phase != PH_ITER) {  // (an iteration's sweep of the sums has barriers of its own: the partial sums are added behind them)
      __syncthreads();
      add_helpers(y);
    }
    if (phase == PH_WARM_X || phase == PH_REFRESH_X) {
      unsigned int d3 = 3u * (unsigned int)(rvalid ? row : 0);
      asm volatile("" : "+v"(d3));  // opaque: the address of b is formed here, not kept in two registers through the whole solve
#pragma unroll
      for (int a = 0; a < 3; a++) rr[a] = rvalid ? bvec[d3 + a] - y[a] : 0.0;
      phase = phase == PH_WARM_X ? PH_INIT_W : PH_REFRESH_W;
      continue;
    }
    if (phase != PH_ITER) {
#pragma unroll
      for (int a = 0; a < 3; a++) wr[a] = y[a];
      phase = PH_ITER;
      continue;
    }
    // ---- all workgroups' sums (posted before their products: they are there) ----
    if constexpr (SHARD) {
      if (spare) shard_rank_sums(sa, BL, pa, sums, nb, lane, t_limit, bc, failed);
      __syncthreads();
    } else if (pa.service) {
      // A workgroup with a wavefront to spare (fewer slices than the instantiation's wavefronts) has its LAST wavefront -- no
      // slice, it idles through the product -- collect the sums meanwhile: same order of additions as below, so the same bits;
      // the others find the totals behind ONE barrier.
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
        for (int b = wv * 64 + lane; b - lane < nb && !failed; b += pollers * 64) {  // wave-uniform trip count
          const bool mine = b < nb;
          uint4 q4[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
          for (;;) {
            bool ok = true;
            if (mine) {  // the record's four granules in two 16-byte requests (each granule is one 8-byte store of its writer)
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
      if (wv == 0) {  // fixed order: lane l adds workgroups l, l + 64, ...; then the wave tree -- the same bits in every workgroup
        double t0s = 0, t1s = 0;
        int l0 = lane;
        asm volatile("" : "+v"(l0));  // (opaque, as above)
        for (int b = l0; b < nb; b += 64) { t0s += gath[b]; t1s += gath[kPipeMaxBlocks + b]; }
        t0s = wave_sum(t0s); t1s = wave_sum(t1s);
        if (lane == 0) { bc[0] = t0s; bc[1] = t1s; bc[2] = (bc[4] != 0.0 || ld_sc1_u32(pa.error) != 0u) ? 1.0 : 0.0; }
      }
      __syncthreads();
    }
    lap(3);  // sweep of the sums
    if (uniform_flag(bc[2] != 0.0)) { failed = true; break; }  // a wait timed out somewhere: every workgroup leaves within one phase
    if constexpr (HELP) add_helpers(y);  // (every path of the sweep ends in a workgroup barrier: the helpers' sums of this product are in LDS)
    gamma = uniform_f64(bc[0]);
    const double delta = uniform_f64(bc[1]);
    if (fresh) rho0 = gamma;
    // the while-condition of CGSolver.cpp:147 at the head of the iteration; gamma is bitwise the same in every workgroup
    if (!(gamma > eps2 * rho0) || iter >= max_iter) { done = true; break; }
    double alpha, beta;
    if (fresh) { beta = 0.0; alpha = gamma / delta; }
    else { beta = gamma / gamma_old; alpha = gamma / (delta - beta * gamma / alpha_old); }
    fresh = false;
    // ---- recurrences, in registers (y = n = A m) ----
    iter++;
    it_done++;
    gamma_old = gamma; alpha_old = alpha;
    const bool refresh = iter % 30 == 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      zr[a] = y[a] + beta * zr[a];
      sr[a] = wr[a] + beta * sr[a];
    }
    if (!refresh) {
      // w first: the next product's input m = w / diag leaves now, the rest of the recurrences runs while its stores drain.
      // (Every workgroup has posted this iteration's sums, i.e. finished the product before this one: the buffer is free.)
      double m[3];
#pragma unroll
      for (int a = 0; a < 3; a++) wr[a] = wr[a] - alpha * zr[a];
      precond(wr, m);
      publish(m);
      published = true;
    }
    if constexpr (BJ) {
      double ur[3];
      precond(rr, ur);
#pragma unroll
      for (int a = 0; a < 3; a++) pr[a] = ur[a] + beta * pr[a];
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++) pr[a] = iv[a] * rr[a] + beta * pr[a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) xr[a] = xr[a] + alpha * pr[a];
    if (refresh) {
      phase = PH_REFRESH_X;  // exact residual (CGSolver.cpp:159-166), and the w that goes with it: two more products
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++) rr[a] = rr[a] - alpha * sr[a];
    }
    lap(4);  // recurrences
  }
  if (TIMING && lane == 0) {
    for (int k = 0; k < 5; k++) atomicAdd((unsigned long long*)pa.timing + ((size_t)blockIdx.x * kPipeMaxWaves + wv) * 6 + k, (unsigned long long)tm[k]);
    atomicAdd((unsigned long long*)pa.timing + ((size_t)blockIdx.x * kPipeMaxWaves + wv) * 6 + 5, (unsigned long long)it_done);
  }
  if (failed) return;  // nothing written back: the host re-solves from the vectors it handed over
  // The iteration cap ended the solve: x is left in pg and the start vector stays -- the host checks the iterate's true residual and
  // either takes it or has the two-launch solver repeat the solve from the same start (pcg_solve_pipe)
  const bool capped = done && gamma > eps2 * rho0;
  if (rvalid) {
    unsigned int dof = 3u * (unsigned int)row;
    asm volatile("" : "+v"(dof));  // (formed here: see the exact-residual phase)
    if (capped) {
#pragma unroll
      for (int a = 0; a < 3; a++) pg[dof + a] = xr[a];
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++) xg[dof + a] = xr[a];
      if (!done) {  // the launch was cut (test knob): the next one continues from here
#pragma unroll
        for (int a = 0; a < 3; a++) { rg[dof + a] = rr[a]; wg[dof + a] = wr[a]; zg[dof + a] = zr[a]; sg[dof + a] = sr[a]; pg[dof + a] = pr[a]; }
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->rho0 = rho0; st->eps2 = eps2; st->max_iter = max_iter;
    st->iter = iter;
    st->rho[iter & 1] = done ? gamma : gamma_old;  // what the host's convergence test reads (CGSolver.cpp:189)
    st->done = done ? 1 : 0;
    pa.pstate[0] = gamma_old; pa.pstate[1] = alpha_old;
    // (a shard's launch cut after a pre-publish takes the publish back: its rows are in the planes and the boxes, but no counter was raised
    // for it, and the next launch publishes the same values under the same number)
    pa.seqs[0] = (SHARD && published) ? pub - 1u : pub; pa.seqs[1] = sums;
  }
}

}  // namespace fb
