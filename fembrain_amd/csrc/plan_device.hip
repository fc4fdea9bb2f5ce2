// Device-side builder of the FEM plan (the arrays of fem_plan.h that the per-step kernels read) -- the whole system of an
// unsharded handle, or one rank's rows of a sharded one (PlanShard: the host does the partition, numbering and halo lists,
// fem_plan.cpp build_fem_partition, everything per row happens here): block pattern,
// SELL-64 layout and the per-(row, slot) element contribution lists, straight from the tet list in device memory.
//
// Deformable::syncForceModel (src/deformable/Deformable.cpp:127-220) rebuilds all of this after every cut; on the host
// (fem_plan.cpp, 16 threads) that is 61-68 ms at 1M tets and was 90 % of a re-sync.  Here it is a handful of streaming
// passes around one radix sort:
//   1. every tet emits its 16 (row, col) vertex pairs, key = row << b | col (b = bits of a node id, so the sort runs over 2 b bits only), value = tet << 4 | i << 2 | j -- the
//      contribution word of fem_plan.h -- and every node one marker pair (a, a) so that a node no element references
//      still gets its diagonal block (fem_plan.cpp: "isolated node");
//   2. a stable radix sort by key puts the pairs in pattern order: rows ascending, columns ascending inside a row, and
//      inside one block the contributions in ascending (tet, i, j) order -- the order the reference's element loop
//      accumulates them in (corotationalLinearFEM.cpp:230-469) -- with the marker last;
//   3. run-length encoding of the keys gives the blocks (bptr / bcol) and their contribution counts; one wavefront per
//      64-row slice then lays out SELL-64 (width = longest row, padding columns = the row itself), the slot table of the
//      blocks and the per-slot list heights; two scans give the slot and list offsets; a last pass copies the sorted
//      contribution words into the [slot][t][lane] table.
// The result is bit for bit the host plan (tests/test_fem_gpu.py::test_device_plan_equals_host_plan).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "fem_plan.h"
#include "plan_device.h"

namespace fb {
namespace {

constexpr int kB = 256;

// rows [0, n_rows) of local ids are matrix rows; a pair whose row is a halo node gets the row value n_rows and so sorts behind
// every real pair.  The column part of the key is the GLOBAL id (node_lo + c for an owned column, halo[c - n_rows] otherwise;
// unsharded: the id itself).
struct PairGeom {
  int n_rows, node_lo, col_bits;
  const int* halo;
};
__device__ __forceinline__ unsigned int global_col(const PairGeom& g, int c) { return (unsigned int)(c < g.n_rows ? g.node_lo + c : g.halo[c - g.n_rows]); }

__global__ __launch_bounds__(kB) void k_plan_pairs(int n_tets, int n_nodes, PairGeom g, const int4* __restrict__ tets,
                                                   unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, int* __restrict__ first_bad) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  const long long n_tp = 16LL * n_tets;
  if (i < n_tp) {
    const int e = (int)(i >> 4), ij = (int)(i & 15);
    const int4 t = tets[e];
    const int id[4] = {t.x, t.y, t.z, t.w};
    const bool bad = (unsigned int)id[ij & 3] >= (unsigned int)n_nodes || (unsigned int)id[ij >> 2] >= (unsigned int)n_nodes;
    if (bad) atomicMin(first_bad, e);  // the host reports it; the keys of this run are never used
    const int row = id[ij >> 2];
    if (bad || row >= g.n_rows) {
      keys[i] = (unsigned long long)(unsigned int)g.n_rows << g.col_bits;
      vals[i] = kNoContrib;
    } else {
      keys[i] = ((unsigned long long)(unsigned int)row << g.col_bits) | global_col(g, id[ij & 3]);
      vals[i] = ((uint32_t)e << 4) | (uint32_t)ij;
    }
  } else if (i < n_tp + g.n_rows) {
    const unsigned int a = (unsigned int)(i - n_tp);
    keys[i] = ((unsigned long long)a << g.col_bits) | (unsigned int)(g.node_lo + (int)a);
    vals[i] = kNoContrib;
  }
}

// The same with 32-bit keys (round 4) for an unsharded mesh whose widest element `span` (largest id difference inside a tet, measured by
// renumber.hip) leaves room: key = row << cb | (col - row + span), cb = bits of 2 span + 1 -- the same order as (row, col), a third fewer
// bytes through every pass of the sort and one pass fewer (1M-tet cube: 18 + 13 = 31 bits against 36; the sort was a quarter of a re-sync).
__global__ __launch_bounds__(kB) void k_plan_pairs32(int n_tets, int n_nodes, int cb, int span, const int4* __restrict__ tets, unsigned int* __restrict__ keys,
                                                     uint32_t* __restrict__ vals, int* __restrict__ first_bad) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  const long long n_tp = 16LL * n_tets;
  if (i < n_tp) {
    const int e = (int)(i >> 4), ij = (int)(i & 15);
    const int4 t = tets[e];
    const int id[4] = {t.x, t.y, t.z, t.w};
    const bool bad = (unsigned int)id[ij & 3] >= (unsigned int)n_nodes || (unsigned int)id[ij >> 2] >= (unsigned int)n_nodes;
    if (bad) { atomicMin(first_bad, e); keys[i] = 0xFFFFFFFFu; vals[i] = kNoContrib; return; }  // (the host reports it; the keys of this run are never used)
    const int row = id[ij >> 2], col = id[ij & 3];
    keys[i] = ((unsigned int)row << cb) | (unsigned int)(col - row + span);
    vals[i] = ((uint32_t)e << 4) | (uint32_t)ij;
  } else if (i < n_tp + n_nodes) {
    const unsigned int a = (unsigned int)(i - n_tp);
    keys[i] = (a << cb) | (unsigned int)span;
    vals[i] = kNoContrib;
  }
}

__global__ __launch_bounds__(kB) void k_plan_rows32(int n_nodes, int n_blocks, int cb, int span, const unsigned int* __restrict__ ukeys, int* __restrict__ bptr,
                                                    int* __restrict__ bcol) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i < n_blocks) {
    const unsigned int k = ukeys[i];
    bcol[i] = (int)(k >> cb) + (int)(k & ((1u << cb) - 1u)) - span;
  }
  if (i <= n_nodes) {
    const unsigned long long want = (unsigned long long)(unsigned int)i << cb;  // (64-bit: row n_nodes may not fit the key)
    int lo = 0, hi = n_blocks;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((unsigned long long)ukeys[mid] < want) lo = mid + 1; else hi = mid;
    }
    bptr[i] = lo;
  }
}

// block p: row, column; bptr by binary search of the first block of every row
__global__ __launch_bounds__(kB) void k_plan_rows(int n_nodes, int n_blocks, PairGeom g, int n_halo, const unsigned long long* __restrict__ ukeys,
                                                  int* __restrict__ bptr, int* __restrict__ bcol) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i < n_blocks) {
    const int gc = (int)(unsigned int)(ukeys[i] & ((1ULL << g.col_bits) - 1ULL));
    int c = gc - g.node_lo;
    if (c < 0 || c >= g.n_rows) {  // a halo column: its local id is n_rows + its rank among the halo nodes
      int lo = 0, hi = n_halo;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (g.halo[mid] < gc) lo = mid + 1; else hi = mid;
      }
      c = g.n_rows + lo;
    }
    bcol[i] = c;
  }
  if (i <= n_nodes) {
    const unsigned long long want = (unsigned long long)(unsigned int)i << g.col_bits;
    int lo = 0, hi = n_blocks;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ukeys[mid] < want) lo = mid + 1; else hi = mid;
    }
    bptr[i] = lo;
  }
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// one wavefront per slice: width = longest row
__global__ __launch_bounds__(kB) void k_plan_widths(int n_nodes, int n_slices, const int* __restrict__ bptr, int* __restrict__ width) {
  const int s = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int a = s * 64 + lane;
  const int len = a < n_nodes ? bptr[a + 1] - bptr[a] : 0;
  const int w = wave_max(len);
  if (lane == 0) width[s] = w;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

// one wavefront per slice: column ids, the slot of every block, and the height of every slot's contribution list.
// 16-bit column words (coldelta), two forms:
//   halo_base == nullptr (unsharded): column - row;
//   else (a shard: columns >= n_nodes are halo nodes, numbered after the owned ones, far from any row): bit 0 = 0: (column - row) << 1;
//   bit 0 = 1: (column - n_nodes - halo_base[slice]) << 1 | 1, halo_base[slice] = the slice's lowest halo column -- the halo
//   columns of 64 consecutive rows lie close together.  `wide` is set when a value does not fit.
__global__ __launch_bounds__(kB) void k_plan_sell(int n_nodes, int n_slices, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                  const unsigned int* __restrict__ ucnt, const int* __restrict__ slice_off, int* __restrict__ colidx,
                                                  int* __restrict__ blk_slot, int* __restrict__ slot_ccnt, short* __restrict__ coldelta,
                                                  int* __restrict__ wide, int* __restrict__ halo_base) {
  const int s = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int a = s * 64 + lane;
  const int so = slice_off[s], w = slice_off[s + 1] - so;
  const int first = a < n_nodes ? bptr[a] : 0, len = a < n_nodes ? bptr[a + 1] - first : 0;
  int hb = 0;
  if (halo_base) {
    int lo = 0x7fffffff;
    for (int k = 0; k < len; k++) {
      const int col = bcol[first + k];
      if (col >= n_nodes) lo = min(lo, col - n_nodes);
    }
    lo = wave_min(lo);
    hb = lo == 0x7fffffff ? 0 : lo;
    if (lane == 0) halo_base[s] = hb;
  }
  // four slots at a time, their loads first: the loop was a chain of dependent global loads, one latency per slot (round 4)
  for (int k0 = 0; k0 < w; k0 += 4) {
    int colv[4], cntv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = k0 + j;
      colv[j] = k < len ? bcol[first + k] : 0;
      cntv[j] = k < len ? (int)ucnt[first + k] : 0;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = k0 + j;
      if (k >= w) break;  // wave-uniform
      int c = 0, col;
      if (k < len) {
        col = colv[j];
        colidx[((size_t)so + k) * 64 + lane] = col;
        blk_slot[first + k] = so + k;
        c = cntv[j] - (col == a ? 1 : 0);  // the diagonal block's run ends with the marker pair
      } else {
        col = a < n_nodes ? a : 0;
        colidx[((size_t)so + k) * 64 + lane] = col;  // padding: any valid column, its values stay zero
        if (a >= n_nodes) col = n_nodes - 1;       // (in the 16-bit form a lane past the last row points at the last row instead)
      }
      int word;
      if (!halo_base) {
        word = col - a;
        if (word < -32768 || word > 32767) atomicOr(wide, 1);
      } else if (col >= n_nodes) {
        const int off = col - n_nodes - hb;
        if (off > 16383) atomicOr(wide, 1);
        word = (off << 1) | 1;
      } else {
        const int d = col - a;
        if (d < -16384 || d > 16383) atomicOr(wide, 1);
        word = d * 2;
      }
      coldelta[((size_t)so + k) * 64 + lane] = (short)word;
      const int m = wave_max(c);
      if (lane == 0) slot_ccnt[so + k] = m;
    }
  }
}

// one wavefront per slice: the sorted contribution words of every block go to contrib[slot_coff[slot] + t][lane]
__global__ __launch_bounds__(kB) void k_plan_contrib(int n_nodes, int n_slices, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                     const unsigned int* __restrict__ ucnt,
                                                     const unsigned int* __restrict__ cstart, const uint32_t* __restrict__ vals,
                                                     const int* __restrict__ slice_off, const int* __restrict__ slot_coff,
                                                     const int* __restrict__ slot_ccnt, uint32_t* __restrict__ contrib) {
  const int s = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int a = s * 64 + lane;
  const int so = slice_off[s], w = slice_off[s + 1] - so;
  const int first = a < n_nodes ? bptr[a] : 0, len = a < n_nodes ? bptr[a + 1] - first : 0;
  for (int k = 0; k < w; k++) {
    const int height = slot_ccnt[so + k];  // wave-uniform
    const int cnt = k < len ? (int)ucnt[first + k] - (bcol[first + k] == a ? 1 : 0) : 0;
    const unsigned int from = k < len ? cstart[first + k] : 0u;
    uint32_t* out = contrib + (size_t)slot_coff[so + k] * 64 + lane;
    for (int t = 0; t < height; t++) out[(size_t)t * 64] = t < cnt ? vals[from + t] : kNoContrib;
  }
}

// The same through LDS (round 4): the contribution words of a slice's 64 rows are ONE contiguous run of the sorted list (rows ascending,
// a row's blocks ascending), so the wavefront fetches it with coalesced loads, and the lanes -- each walking its own row's part -- read
// LDS instead of 64 different cache lines per load (314 -> see DESIGN.md; 1.1M tets).  A slice whose run exceeds `cap` words (hub nodes)
// reads global memory as k_plan_contrib does.
__global__ __launch_bounds__(kB) void k_plan_contrib_lds(int n_nodes, int n_slices, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                          const unsigned int* __restrict__ ucnt, const unsigned int* __restrict__ cstart, const uint32_t* __restrict__ vals,
                                                          const int* __restrict__ slice_off, const int* __restrict__ slot_coff, const int* __restrict__ slot_ccnt,
                                                          uint32_t* __restrict__ contrib, int cap) {
  // one workgroup of four wavefronts per slice: all fetch the run, then wavefront w fills the slots w, w + 4, ...
  extern __shared__ uint32_t seg[];
  const int s = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (s >= n_slices) return;
  const int a = s * 64 + lane;
  const int so = slice_off[s], w = slice_off[s + 1] - so;
  const int first = a < n_nodes ? bptr[a] : 0, len = a < n_nodes ? bptr[a + 1] - first : 0;
  const int b0 = bptr[s * 64], b1 = bptr[min(s * 64 + 64, n_nodes)];
  const unsigned int seg_lo = b1 > b0 ? cstart[b0] : 0u, seg_hi = b1 > b0 ? cstart[b1 - 1] + ucnt[b1 - 1] : 0u;
  const bool staged = seg_hi - seg_lo <= (unsigned int)cap;  // workgroup-uniform
  if (staged) {
    for (unsigned int i = threadIdx.x; i < seg_hi - seg_lo; i += kB) seg[i] = vals[seg_lo + i];
    __syncthreads();
  }
  // (two of the wavefront's slots at a time, their table entries loaded first: one latency per pair instead of per slot)
  for (int k0 = wv; k0 < w; k0 += 2 * (kB / 64)) {
    int height[2], cnt[2], coff[2];
    unsigned int from[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int k = k0 + j * (kB / 64);
      const bool on = k < w;
      height[j] = on ? slot_ccnt[so + k] : 0;  // wave-uniform
      coff[j] = on ? slot_coff[so + k] : 0;
      const bool mine = on && k < len;
      cnt[j] = mine ? (int)ucnt[first + k] - (bcol[first + k] == a ? 1 : 0) : 0;
      from[j] = mine ? cstart[first + k] : seg_lo;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      uint32_t* out = contrib + (size_t)coff[j] * 64 + lane;
      if (staged) {
        const uint32_t* in = seg + (from[j] - seg_lo);
        for (int t = 0; t < height[j]; t++) out[(size_t)t * 64] = t < cnt[j] ? in[t] : kNoContrib;
      } else {
        for (int t = 0; t < height[j]; t++) out[(size_t)t * 64] = t < cnt[j] ? vals[from[j] + t] : kNoContrib;
      }
    }
  }
}

// ---- incidence lists of the element-major assembly (fem_device.hip.h k_assemble_tets) ---------------------------------------
// one wavefront per slice: length of every row's incidence list = contributions of its diagonal block
__global__ __launch_bounds__(kB) void k_inc_heights(int n_slices, int n_owned, const int* __restrict__ slice_off, const int* __restrict__ colidx, const int* __restrict__ slot_coff, const int* __restrict__ slot_ccnt,
                                                        const uint32_t* __restrict__ contrib, int* __restrict__ height) {
  const int s = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int row = s * 64 + lane;
  const int so = slice_off[s], width = slice_off[s + 1] - so;
  int cnt = 0;
  if (row < n_owned && width <= kIncMaxWidth)  // (a wider slice gets no lists: the slot-major assembly kernel takes it)
    for (int k = 0; k < width; k++)
      if (colidx[((size_t)so + k) * 64 + lane] == row) {  // the first such slot is the diagonal block (padding repeats the row id later)
        const int coff = slot_coff[so + k], ccnt = slot_ccnt[so + k];
        bool ended = false;
        for (int t0 = 0; t0 < ccnt && !ended; t0 += 8) {  // (eight entries of the list in flight: one at a time was a chain of ~25 dependent loads)
          uint32_t wd[8];
#pragma unroll
          for (int j = 0; j < 8; j++) wd[j] = t0 + j < ccnt ? contrib[((size_t)coff + t0 + j) * 64 + lane] : 0xFFFFFFFFu;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            if (!ended && wd[j] != 0xFFFFFFFFu) cnt++; else ended = true;
          }
        }
        break;
      }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt = max(cnt, __shfl_xor(cnt, o, 64));
  if (lane == 0) height[s] = cnt;
}

__global__ __launch_bounds__(kB) void k_inc_fill(int n_slices, int n_owned, const int* __restrict__ slice_off, const int* __restrict__ colidx, const int* __restrict__ slot_coff, const int* __restrict__ slot_ccnt,
                                                     const uint32_t* __restrict__ contrib, const int4* __restrict__ tets, const int* __restrict__ inc_off,
                                                     uint32_t* __restrict__ inc, uint32_t* __restrict__ inc_slot, int ascending) {
  // the columns of the slice are searched 4 x (list length) times per row: staged in LDS once (the builder is called for
  // slices of at most kIncMaxWidth slots, the limit of the kernel that reads the lists)
  __shared__ int cols[kB / 64][kIncMaxWidth][64];
  const int wv = threadIdx.x >> 6;
  const int s = blockIdx.x * (kB / 64) + wv, lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  const int row = s * 64 + lane;
  const int so = slice_off[s], width = min(slice_off[s + 1] - so, kIncMaxWidth);
  const int io = inc_off[s], height = inc_off[s + 1] - io;
  int kd = -1;
  for (int k = 0; k < width; k++) {
    const int c = colidx[((size_t)so + k) * 64 + lane];
    cols[wv][k][lane] = c;
    if (row < n_owned && kd < 0 && c == row) kd = k;
  }
  const int coff = kd >= 0 ? slot_coff[so + kd] : 0, ccnt = kd >= 0 ? slot_ccnt[so + kd] : 0;
  // the real blocks of the lane's row: ascending columns, then padding (the row id again).  Their number, so that the four searches per
  // list entry below are binary (round 5: the linear walk over up to 31 slots made this kernel 375 us of a re-sync at 1.1M tets)
  // (ascending: an unsharded plan.  A shard orders a row by GLOBAL column while the ids here are local -- halo nodes of a lower rank come first
  // and carry the larger local ids -- so it keeps the linear walk.)
  int len = width > 0 ? 1 : 0;
  while (ascending && len < width && cols[wv][len][lane] > cols[wv][len - 1][lane]) len++;
  // four list entries at a time: their words, then their elements' node ids, in flight together (one entry after the other was a chain
  // of two dependent loads per entry: 113 us of every re-sync at 1.1M tets)
  for (int t0 = 0; t0 < height; t0 += 4) {
    uint32_t c[4];
    int4 tt[4];
#pragma unroll
    for (int u = 0; u < 4; u++) c[u] = t0 + u < ccnt ? contrib[((size_t)coff + t0 + u) * 64 + lane] : 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < 4; u++) tt[u] = c[u] != 0xFFFFFFFFu ? tets[c[u] >> 4] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (t0 + u >= height) break;  // wave-uniform
      uint32_t w = kNoContrib, sl = 0;
      if (c[u] != 0xFFFFFFFFu) {
        const uint32_t e = c[u] >> 4, i = (c[u] >> 2) & 3;
        w = (e << 2) | i;
        const int id[4] = {tt[u].x, tt[u].y, tt[u].z, tt[u].w};
        for (int j = 0; j < 4; j++) {
          int lo = 0, hi = len;  // (found by construction: node j of an element on this row is a column of the row)
          if (ascending) {
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (cols[wv][mid][lane] < id[j]) lo = mid + 1; else hi = mid;
            }
          } else {
            while (lo < width && cols[wv][lo][lane] != id[j]) lo++;  // the first match is the real block
          }
          sl |= (uint32_t)(lo & 255) << (8 * j);
        }
      }
      inc[((size_t)io + t0 + u) * 64 + lane] = w;
      inc_slot[((size_t)io + t0 + u) * 64 + lane] = sl;
    }
  }
}

// ---- partition of one rank from its own elements ------------------------------------------------------------------------
struct PartCounters { int first_bad, bad_node, not_kept, pad; unsigned long long corners; };

__device__ __forceinline__ int owner_of_node(const int* __restrict__ splits, int n_ranks, int g) {
  int lo = 0, hi = n_ranks;  // largest q with splits[q] <= g
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (splits[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kB) void k_part_mark(int n_tets, const int4* __restrict__ tets, int n_global, int node_lo, int n_owned, const int* __restrict__ splits,
                                                  int n_ranks, unsigned char* __restrict__ nodeflag, unsigned long long* __restrict__ sendmask, unsigned char* __restrict__ keep,
                                                  PartCounters* __restrict__ cnt) {
  const int e = blockIdx.x * kB + threadIdx.x;
  int n_own = 0;
  if (e < n_tets) {
    const int4 t = tets[e];
    const int v[4] = {t.x, t.y, t.z, t.w};
    bool bad = false, own[4];
    for (int i = 0; i < 4; i++) {
      if ((unsigned int)v[i] >= (unsigned int)n_global) {
        if (!bad && atomicMin(&cnt->first_bad, e) > e) cnt->bad_node = v[i];  // (the node reported may belong to another bad tet under a race; the host re-reads its own list)
        bad = true;
        own[i] = false;
        continue;
      }
      own[i] = v[i] >= node_lo && v[i] < node_lo + n_owned;
      n_own += own[i];
    }
    if (!bad) {
      if (!n_own) atomicAdd(&cnt->not_kept, 1);
      else if (n_own < 4) {
        unsigned long long want = 0;
        for (int j = 0; j < 4; j++)
          if (!own[j]) {
            nodeflag[v[j]] = 1;
            want |= 1ull << owner_of_node(splits, n_ranks, v[j]);
          }
        for (int i = 0; i < 4; i++)
          if (own[i]) atomicOr(&sendmask[v[i] - node_lo], want);
      }
    } else {
      n_own = 0;
    }
    keep[e] = n_own > 0 ? 1 : 0;
  }
  // corners on owned nodes: one atomic per wavefront
  for (int o = 32; o; o >>= 1) n_own += __shfl_down(n_own, o);
  if ((threadIdx.x & 63) == 0 && n_own) atomicAdd(&cnt->corners, (unsigned long long)n_own);
}

struct WantedBy {
  const unsigned long long* mask;
  int q;
  __device__ bool operator()(int i) const { return (mask[i] >> q) & 1ull; }
};

__global__ __launch_bounds__(kB) void k_part_local(int n_tets, int4* __restrict__ tets, int node_lo, int n_owned, const int* __restrict__ halo, int n_halo) {
  const int e = blockIdx.x * kB + threadIdx.x;
  if (e >= n_tets) return;
  int4 t = tets[e];
  int* v = &t.x;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int g = v[i];
    if (g >= node_lo && g < node_lo + n_owned) { v[i] = g - node_lo; continue; }
    int lo = 0, hi = n_halo;  // lower bound; g is in the list by construction
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (halo[mid] < g) lo = mid + 1; else hi = mid;
    }
    v[i] = n_owned + lo;
  }
  tets[e] = t;
}

}  // namespace

int device_partition(hipStream_t s, int n_tets, DevBuf<int4>& tets, int n_global, int n_ranks, int rank, const std::vector<int>& splits,
                     DevicePartition& out, PlanWorkspace& W) {
  if (n_ranks > 64 || n_ranks < 2) return fail(FB_EINVAL, "device partition handles 2..64 ranks");
  const int node_lo = splits[rank], n_owned = splits[rank + 1] - splits[rank];
  out = DevicePartition();
  FB_TRY(W.nodeflag.reserve((size_t)n_global));
  FB_TRY(W.sendmask.reserve((size_t)n_owned));
  FB_TRY(W.picked.reserve((size_t)std::max(n_global, n_owned) + 4));
  FB_TRY(W.splits.reserve((size_t)n_ranks + 1));
  FB_TRY(W.keep.reserve((size_t)n_tets));
  FB_TRY(W.keys.reserve(4));  // the counters (8-byte aligned)
  PartCounters* d_cnt = reinterpret_cast<PartCounters*>(W.keys.p);
  PartCounters cnt = {0x7fffffff, 0, 0, 0, 0ull};
  FB_HIP(hipMemcpyAsync(d_cnt, &cnt, sizeof cnt, hipMemcpyHostToDevice, s));
  FB_HIP(hipMemcpyAsync(W.splits.p, splits.data(), sizeof(int) * (n_ranks + 1), hipMemcpyHostToDevice, s));
  FB_HIP(hipMemsetAsync(W.nodeflag.p, 0, (size_t)n_global, s));
  FB_HIP(hipMemsetAsync(W.sendmask.p, 0, sizeof(unsigned long long) * (size_t)n_owned, s));
  hipLaunchKernelGGL(k_part_mark, dim3((n_tets + kB - 1) / kB), dim3(kB), 0, s, n_tets, tets.p, n_global, node_lo, n_owned, W.splits.p, n_ranks, W.nodeflag.p,
                     W.sendmask.p, W.keep.p, d_cnt);
  FB_HIP(hipGetLastError());
  // halo = the marked nodes, ascending
  int* d_count = W.picked.p + std::max(n_global, n_owned);
  size_t bytes = 0;
  FB_HIP(rocprim::select(nullptr, bytes, rocprim::counting_iterator<int>(0), W.nodeflag.p, W.picked.p, d_count, (size_t)n_global, s));
  FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::select(W.temp.p, bytes, rocprim::counting_iterator<int>(0), W.nodeflag.p, W.picked.p, d_count, (size_t)n_global, s));
  int n_halo = 0;
  FB_HIP(hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, s));
  FB_HIP(hipMemcpyAsync(&n_halo, d_count, sizeof(int), hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  if (cnt.first_bad != 0x7fffffff) {
    out.first_bad_tet = cnt.first_bad;
    out.bad_node = cnt.bad_node;
    return fail(FB_EINVAL, "tet %d references a node outside [0,%d)", cnt.first_bad, n_global);
  }
  out.all_kept = cnt.not_kept == 0;
  out.n_kept = n_tets - cnt.not_kept;
  out.owned_corners = (long long)cnt.corners;
  if (out.n_kept <= 0) return fail(FB_EINVAL, "rank %d: no element touches its nodes [%d,%d)", rank, node_lo, node_lo + n_owned);
  if (!out.all_kept) {
    // the caller passed more than this rank's elements (the whole mesh): keep those with an owned node, in their order, and
    // remember which they were (error messages name global element ids)
    FB_TRY(W.tetsel.reserve((size_t)out.n_kept));
    FB_TRY(W.idsel.reserve((size_t)out.n_kept + 4));
    int* d_n = W.idsel.p + out.n_kept;
    bytes = 0;
    FB_HIP(rocprim::select(nullptr, bytes, tets.p, W.keep.p, W.tetsel.p, d_n, (size_t)n_tets, s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::select(W.temp.p, bytes, tets.p, W.keep.p, W.tetsel.p, d_n, (size_t)n_tets, s));
    bytes = 0;
    FB_HIP(rocprim::select(nullptr, bytes, rocprim::counting_iterator<int>(0), W.keep.p, W.idsel.p, d_n, (size_t)n_tets, s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::select(W.temp.p, bytes, rocprim::counting_iterator<int>(0), W.keep.p, W.idsel.p, d_n, (size_t)n_tets, s));
    out.tet_global.resize((size_t)out.n_kept);
    FB_HIP(hipMemcpyAsync(out.tet_global.data(), W.idsel.p, sizeof(int) * (size_t)out.n_kept, hipMemcpyDeviceToHost, s));
    FB_HIP(hipStreamSynchronize(s));
    tets.swap(W.tetsel);  // the handle's element buffer is now the compacted list (the old one serves the next re-sync)
  }
  n_tets = out.n_kept;
  int4* d_tets = tets.p;
  out.halo.resize((size_t)n_halo);
  if (n_halo) FB_HIP(hipMemcpyAsync(out.halo.data(), W.picked.p, sizeof(int) * (size_t)n_halo, hipMemcpyDeviceToHost, s));
  // local numbering of the elements while the halo list is still in W.picked
  hipLaunchKernelGGL(k_part_local, dim3((n_tets + kB - 1) / kB), dim3(kB), 0, s, n_tets, d_tets, node_lo, n_owned, W.picked.p, n_halo);
  FB_HIP(hipGetLastError());
  FB_HIP(hipStreamSynchronize(s));
  // send lists: the neighbours are the owners of the halo nodes (a rank that wants my nodes owns nodes of the same elements)
  out.send_off.assign((size_t)n_ranks + 1, 0);
  std::vector<char> neighbour((size_t)n_ranks, 0);
  {
    int q = 0;
    for (int g : out.halo) {
      while (g >= splits[q + 1]) q++;
      neighbour[q] = 1;
    }
  }
  for (int q = 0; q < n_ranks; q++) {
    if (neighbour[q]) {
      WantedBy pred = {W.sendmask.p, q};
      size_t b2 = 0;
      FB_HIP(rocprim::select(nullptr, b2, rocprim::counting_iterator<int>(0), W.picked.p, d_count, (size_t)n_owned, pred, s));
      FB_TRY(W.temp.reserve(std::max<size_t>(b2, 16)));
      FB_HIP(rocprim::select(W.temp.p, b2, rocprim::counting_iterator<int>(0), W.picked.p, d_count, (size_t)n_owned, pred, s));
      int n_send = 0;
      FB_HIP(hipMemcpyAsync(&n_send, d_count, sizeof(int), hipMemcpyDeviceToHost, s));
      FB_HIP(hipStreamSynchronize(s));
      const size_t at = out.send_local.size();
      out.send_local.resize(at + (size_t)n_send);
      if (n_send) FB_HIP(hipMemcpyAsync(out.send_local.data() + at, W.picked.p, sizeof(int) * (size_t)n_send, hipMemcpyDeviceToHost, s));
      FB_HIP(hipStreamSynchronize(s));
    }
    out.send_off[q + 1] = (int)out.send_local.size();
  }
  return FB_OK;
}

int build_incidence_device(hipStream_t s, int n_slices, int n_owned, const int* slice_off, const int* colidx, const int* slot_coff, const int* slot_ccnt,
                           const uint32_t* contrib, const int4* tets, DevBuf<int>& inc_off, DevBuf<uint32_t>& inc, DevBuf<uint32_t>& inc_slot, PlanWorkspace& W, bool ascending_columns) {
  FB_TRY(W.width.reserve((size_t)n_slices + 1));
  FB_HIP(hipMemsetAsync(W.width.p, 0, sizeof(int) * ((size_t)n_slices + 1), s));
  const dim3 sg((unsigned)((n_slices + kB / 64 - 1) / (kB / 64)));
  hipLaunchKernelGGL(k_inc_heights, sg, dim3(kB), 0, s, n_slices, n_owned, slice_off, colidx, slot_coff, slot_ccnt, contrib, W.width.p);
  FB_HIP(hipGetLastError());
  FB_TRY(inc_off.alloc((size_t)n_slices + 1));
  size_t bytes = 0;
  FB_HIP(rocprim::exclusive_scan(nullptr, bytes, W.width.p, inc_off.p, 0, (size_t)n_slices + 1, rocprim::plus<int>(), s));
  FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::exclusive_scan(W.temp.p, bytes, W.width.p, inc_off.p, 0, (size_t)n_slices + 1, rocprim::plus<int>(), s));
  int rows = 0;
  FB_TRY(inc_off.download(&rows, 1, s, (size_t)n_slices));
  FB_TRY(inc.alloc(std::max<size_t>(1, (size_t)rows * 64)));
  FB_TRY(inc_slot.alloc(std::max<size_t>(1, (size_t)rows * 64)));
  hipLaunchKernelGGL(k_inc_fill, sg, dim3(kB), 0, s, n_slices, n_owned, slice_off, colidx, slot_coff, slot_ccnt, contrib, tets, inc_off.p, inc.p, inc_slot.p, ascending_columns ? 1 : 0);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

static int plan_from_sorted(hipStream_t s, int n_nodes, long long n_pairs, long long n_valid, bool narrow, int cb32, int span, const PairGeom& geom, const PlanShard* shard,
                            DevicePlan& D, PlanWorkspace& W);

int build_plan_device(hipStream_t s, int n_nodes_local, int n_tets, const int4* d_tets, DevicePlan& D, PlanWorkspace& W, const PlanShard* shard, int span) {
  // n_nodes: matrix rows (owned nodes); n_nodes_local: range of the node ids in d_tets
  const int n_nodes = shard ? shard->n_rows : n_nodes_local;
  const long long n_pairs = 16LL * n_tets + n_nodes;
  if (n_pairs >= (1LL << 31)) return fail(FB_EINVAL, "mesh too large for the device plan builder (%lld pairs)", n_pairs);
  DevBuf<unsigned long long>&keys = W.keys, &keys_s = W.keys_s;
  DevBuf<uint32_t>&vals = W.vals, &vals_s = W.vals_s;
  DevBuf<char>& temp = W.temp;
  FB_TRY(W.flags.reserve(2));
  FB_TRY(keys.reserve((size_t)n_pairs));
  FB_TRY(keys_s.reserve((size_t)n_pairs));
  FB_TRY(vals.reserve((size_t)n_pairs));
  FB_TRY(vals_s.reserve((size_t)n_pairs));
  PairGeom geom;
  geom.n_rows = n_nodes; geom.node_lo = shard ? shard->node_lo : 0; geom.halo = shard ? shard->d_halo : nullptr;
  geom.col_bits = 1;
  while ((1LL << geom.col_bits) < (shard ? shard->n_global : n_nodes)) geom.col_bits++;
  int row_bits = 1;
  while ((1LL << row_bits) < (long long)n_nodes + (shard ? 1 : 0)) row_bits++;  // a shard needs the row value n_rows for the dropped pairs
  if (row_bits + geom.col_bits > 63) return fail(FB_EINVAL, "mesh too large for the device plan builder's sort key");
  const long long n_valid = shard ? shard->n_pairs : n_pairs;  // pairs that belong to a row
  // 32-bit keys where the mesh is banded enough (see k_plan_pairs32)
  int cb32 = 1;
  while (span >= 0 && cb32 < 31 && (1LL << cb32) < 2LL * span + 1) cb32++;
  int rb32 = 1;
  while ((1LL << rb32) < (long long)n_nodes) rb32++;
  const bool narrow = !shard && span >= 0 && span < n_nodes && rb32 + cb32 <= 32 && !(getenv("FEMBRAIN_PLAN_KEYS64") && atoi(getenv("FEMBRAIN_PLAN_KEYS64")) != 0);
  const int init[2] = {0x7fffffff, 0};  // [0] lowest tet with a bad node id, [1] "a column difference does not fit 16 bits"
  const int none = init[0];
  FB_HIP(hipMemcpyAsync(W.flags.p, init, sizeof init, hipMemcpyHostToDevice, s));
  struct { int* p; } bad = {W.flags.p};
  unsigned int* keys32 = reinterpret_cast<unsigned int*>(keys.p);
  unsigned int* keys32_s = reinterpret_cast<unsigned int*>(keys_s.p);
  if (narrow) hipLaunchKernelGGL(k_plan_pairs32, dim3((unsigned)((n_pairs + kB - 1) / kB)), dim3(kB), 0, s, n_tets, n_nodes_local, cb32, span, d_tets, keys32, vals.p, bad.p);
  else hipLaunchKernelGGL(k_plan_pairs, dim3((unsigned)((n_pairs + kB - 1) / kB)), dim3(kB), 0, s, n_tets, n_nodes_local, geom, d_tets, keys.p, vals.p, bad.p);
  FB_HIP(hipGetLastError());
  int first_bad = none;
  FB_HIP(hipMemcpyAsync(&first_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  D.first_bad_tet = first_bad == none ? -1 : first_bad;
  if (D.first_bad_tet >= 0) return fail(FB_EINVAL, "tet %d references a node outside [0,%d)", D.first_bad_tet, n_nodes_local);
  size_t bytes = 0;
  const unsigned key_bits = narrow ? (unsigned)(rb32 + cb32) : (unsigned)(row_bits + geom.col_bits);
  if (narrow) {
    FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys32, keys32_s, vals.p, vals_s.p, (size_t)n_pairs, 0u, key_bits, s));
    FB_TRY(temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::radix_sort_pairs(temp.p, bytes, keys32, keys32_s, vals.p, vals_s.p, (size_t)n_pairs, 0u, key_bits, s));
  } else {
    FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys.p, keys_s.p, vals.p, vals_s.p, (size_t)n_pairs, 0u, key_bits, s));
    FB_TRY(temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::radix_sort_pairs(temp.p, bytes, keys.p, keys_s.p, vals.p, vals_s.p, (size_t)n_pairs, 0u, key_bits, s));
  }
  return plan_from_sorted(s, n_nodes, n_pairs, n_valid, narrow, cb32, span, geom, shard, D, W);
}

// The plan from the sorted pair list in W.keys_s / W.vals_s (stage 3 of the header comment)
static int plan_from_sorted(hipStream_t s, int n_nodes, long long n_pairs, long long n_valid, bool narrow, int cb32, int span, const PairGeom& geom, const PlanShard* shard,
                            DevicePlan& D, PlanWorkspace& W) {
  DevBuf<unsigned long long>&keys_s = W.keys_s, &ukeys = W.ukeys;
  DevBuf<uint32_t>& vals_s = W.vals_s;
  DevBuf<unsigned int>&ucnt = W.ucnt, &cstart = W.cstart, &nruns = W.nruns;
  DevBuf<char>& temp = W.temp;
  unsigned int* keys32_s = reinterpret_cast<unsigned int*>(keys_s.p);
  size_t bytes = 0;
  static const bool timing = getenv("FEMBRAIN_TIMING") && atoi(getenv("FEMBRAIN_TIMING")) >= 2;  // development aid: the stages, each synchronised
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    fprintf(stderr, "[fembrain] plan from the sorted list: %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  FB_TRY(W.flags.reserve(2));
  FB_TRY(ukeys.reserve((size_t)n_pairs));
  FB_TRY(ucnt.reserve((size_t)n_pairs));
  FB_TRY(nruns.reserve(1));
  unsigned int* ukeys32 = reinterpret_cast<unsigned int*>(ukeys.p);
  if (narrow) {
    bytes = 0;  // blocks = runs of equal keys
    FB_HIP(rocprim::run_length_encode(nullptr, bytes, keys32_s, (unsigned int)n_valid, ukeys32, ucnt.p, nruns.p, s));
    FB_TRY(temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::run_length_encode(temp.p, bytes, keys32_s, (unsigned int)n_valid, ukeys32, ucnt.p, nruns.p, s));
  } else {
    bytes = 0;  // blocks = runs of equal keys
    FB_HIP(rocprim::run_length_encode(nullptr, bytes, keys_s.p, (unsigned int)n_valid, ukeys.p, ucnt.p, nruns.p, s));
    FB_TRY(temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::run_length_encode(temp.p, bytes, keys_s.p, (unsigned int)n_valid, ukeys.p, ucnt.p, nruns.p, s));
  }
  unsigned int nb = 0;
  FB_TRY(nruns.download(&nb, 1, s));
  lap("run-length encoding");
  D.n_blocks = (int)nb;
  FB_TRY(cstart.reserve((size_t)nb));
  bytes = 0;
  FB_HIP(rocprim::exclusive_scan(nullptr, bytes, ucnt.p, cstart.p, 0u, (size_t)nb, rocprim::plus<unsigned int>(), s));
  FB_TRY(temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::exclusive_scan(temp.p, bytes, ucnt.p, cstart.p, 0u, (size_t)nb, rocprim::plus<unsigned int>(), s));
  FB_TRY(D.bptr->alloc((size_t)n_nodes + 1));
  FB_TRY(D.bcol->alloc((size_t)nb));
  FB_TRY(D.blk_slot->alloc((size_t)nb));
  if (narrow)
    hipLaunchKernelGGL(k_plan_rows32, dim3((unsigned)((std::max<long long>(nb, n_nodes + 1) + kB - 1) / kB)), dim3(kB), 0, s, n_nodes, (int)nb, cb32, span, ukeys32, D.bptr->p,
                       D.bcol->p);
  else
    hipLaunchKernelGGL(k_plan_rows, dim3((unsigned)((std::max<long long>(nb, n_nodes + 1) + kB - 1) / kB)), dim3(kB), 0, s, n_nodes, (int)nb, geom, shard ? shard->n_halo : 0, ukeys.p,
                       D.bptr->p, D.bcol->p);
  FB_HIP(hipGetLastError());
  lap("block rows");
  if (D.ucnt_keep) {  // (the pairs of every block, kept with the pattern: fb_fem_resync_delta updates the plan from the plan, delta.hip)
    FB_TRY(D.ucnt_keep->alloc((size_t)nb));
    FB_HIP(hipMemcpyAsync(D.ucnt_keep->p, ucnt.p, sizeof(unsigned int) * (size_t)nb, hipMemcpyDeviceToDevice, s));
  }
  FB_TRY(plan_layout_from_csr(s, n_nodes, ucnt.p, shard != nullptr, D, W));
  lap("SELL layout and slot offsets");
  FB_TRY(D.contrib->alloc(std::max<size_t>(1, (size_t)D.n_crows * kSliceRows)));
  const int n_slices = D.n_slices;
  const dim3 sg((unsigned)((n_slices + kB / 64 - 1) / (kB / 64)));
  const bool direct = getenv("FEMBRAIN_PLAN_CONTRIB") && !strcmp(getenv("FEMBRAIN_PLAN_CONTRIB"), "direct");  // development aid: the round-1 kernel
  if (direct) {
    hipLaunchKernelGGL(k_plan_contrib, sg, dim3(kB), 0, s, n_nodes, n_slices, D.bptr->p, D.bcol->p, ucnt.p, cstart.p, vals_s.p, D.slice_off->p, D.slot_coff->p,
                       D.slot_ccnt->p, D.contrib->p);
  } else {
    constexpr int kSegCap = 10240;  // words: 40 KB of LDS per workgroup, four workgroups per CU
    FB_HIP(hipFuncSetAttribute((const void*)k_plan_contrib_lds, hipFuncAttributeMaxDynamicSharedMemorySize, kSegCap * (int)sizeof(uint32_t)));
    hipLaunchKernelGGL(k_plan_contrib_lds, dim3((unsigned)std::max(1, n_slices)), dim3(kB), kSegCap * sizeof(uint32_t), s, n_nodes, n_slices, D.bptr->p, D.bcol->p, ucnt.p, cstart.p,
                       vals_s.p, D.slice_off->p, D.slot_coff->p, D.slot_ccnt->p, D.contrib->p, kSegCap);
  }
  FB_HIP(hipGetLastError());
  // (no wait here: the workspace and the plan's buffers outlive the kernel; the caller's next stages queue behind it)
  lap("contribution table");
  return FB_OK;
}

// SELL-64 layout, slot table and list heights of the pattern in D.bptr / D.bcol with `ucnt` pairs per block (the diagonal block's count
// includes its marker): slice_off, colidx, blk_slot, coldelta, slot_ccnt, slot_coff; n_slices, n_slots, n_crows, max_width, deltas_fit16.
int plan_layout_from_csr(hipStream_t s, int n_nodes, const unsigned int* ucnt, bool shard, DevicePlan& D, PlanWorkspace& W) {
  DevBuf<int>& width = W.width;
  DevBuf<char>& temp = W.temp;
  size_t bytes = 0;
  FB_TRY(W.flags.reserve(2));
  // SELL-64
  const int n_slices = (n_nodes + kSliceRows - 1) / kSliceRows;
  D.n_slices = n_slices;
  const dim3 sg((unsigned)((n_slices + kB / 64 - 1) / (kB / 64)));
  FB_TRY(width.reserve((size_t)n_slices + 1));
  FB_HIP(hipMemsetAsync(width.p, 0, sizeof(int) * ((size_t)n_slices + 1), s));
  hipLaunchKernelGGL(k_plan_widths, sg, dim3(kB), 0, s, n_nodes, n_slices, D.bptr->p, width.p);
  FB_HIP(hipGetLastError());
  FB_TRY(D.slice_off->alloc((size_t)n_slices + 1));
  bytes = 0;
  FB_HIP(rocprim::exclusive_scan(nullptr, bytes, width.p, D.slice_off->p, 0, (size_t)n_slices + 1, rocprim::plus<int>(), s));
  FB_TRY(temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::exclusive_scan(temp.p, bytes, width.p, D.slice_off->p, 0, (size_t)n_slices + 1, rocprim::plus<int>(), s));
  D.slice_off_host.resize((size_t)n_slices + 1);
  FB_TRY(D.slice_off->download(D.slice_off_host.data(), (size_t)n_slices + 1, s));
  D.n_slots = D.slice_off_host[n_slices];
  FB_TRY(D.colidx->alloc((size_t)D.n_slots * kSliceRows));
  FB_TRY(D.slot_ccnt->alloc((size_t)D.n_slots + 1));
  FB_TRY(D.slot_ccnt->zero(s));
  FB_TRY(D.slot_coff->alloc((size_t)D.n_slots + 1));
  FB_TRY(D.coldelta->alloc(std::max<size_t>(1, (size_t)D.n_slots * kSliceRows)));
  if (shard && D.halo_base) FB_TRY(D.halo_base->alloc((size_t)std::max(1, n_slices)));
  struct { int* p; } wide = {W.flags.p + 1};
  FB_HIP(hipMemsetAsync(wide.p, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_plan_sell, sg, dim3(kB), 0, s, n_nodes, n_slices, D.bptr->p, D.bcol->p, ucnt, D.slice_off->p, D.colidx->p, D.blk_slot->p,
                     D.slot_ccnt->p, D.coldelta->p, wide.p, (shard && D.halo_base) ? D.halo_base->p : nullptr);
  FB_HIP(hipGetLastError());
  bytes = 0;
  FB_HIP(rocprim::exclusive_scan(nullptr, bytes, D.slot_ccnt->p, D.slot_coff->p, 0, (size_t)D.n_slots + 1, rocprim::plus<int>(), s));
  FB_TRY(temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::exclusive_scan(temp.p, bytes, D.slot_ccnt->p, D.slot_coff->p, 0, (size_t)D.n_slots + 1, rocprim::plus<int>(), s));
  int crows = 0, w = 0;  // (one wait for both)
  FB_HIP(hipMemcpyAsync(&w, wide.p, sizeof(int), hipMemcpyDeviceToHost, s));
  FB_HIP(hipMemcpyAsync(&crows, D.slot_coff->p + D.n_slots, sizeof(int), hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  D.deltas_fit16 = w == 0;
  if ((long long)crows * kSliceRows >= (1LL << 31)) return fail(FB_EINVAL, "contribution table too large (%d rows)", crows);
  D.n_crows = crows;
  return FB_OK;
}

}  // namespace fb
