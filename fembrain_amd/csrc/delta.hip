// fb_fem_resync_delta: the device side (see delta.h)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "delta.h"
#include "fem_plan.h"

namespace fb {
namespace {

constexpr int kB = 256;

inline int pad4(int x) { return (x + 3) & ~3; }
inline dim3 grid_for(long long n) { return dim3((unsigned)std::max<long long>(1, (n + kB - 1) / kB)); }

__global__ __launch_bounds__(kB) void k_delta_mark(int n_removed, const int* __restrict__ removed, int n_changed, const int* __restrict__ changed, unsigned char* __restrict__ estate) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i < n_removed) estate[removed[i]] = 1;
  else if (i < n_removed + n_changed) estate[changed[i - n_removed]] = 2;
}

struct Stays {
  __device__ int operator()(unsigned char st) const { return st != 1 ? 1 : 0; }
};

__global__ __launch_bounds__(kB) void k_delta_relabel(int n, int4* __restrict__ t, int n_nodes, const int* __restrict__ map) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  int4 v = t[i];
  v.x = map[v.x]; v.y = map[v.y]; v.z = map[v.z]; v.w = map[v.w];
  t[i] = v;
}

__global__ __launch_bounds__(kB) void k_delta_tets(int n_old, const int4* __restrict__ tets_old, const unsigned char* __restrict__ estate, const int* __restrict__ pos,
                                                   const int* __restrict__ imap, int n_changed, const int* __restrict__ changed_ids, const int4* __restrict__ changed_nodes,
                                                   int4* __restrict__ tets_new) {
  const int e = blockIdx.x * kB + threadIdx.x;
  if (e >= n_old) return;
  const unsigned char st = estate[e];
  if (st == 1) return;
  int4 t;
  if (st == 2) {
    int lo = 0, hi = n_changed;  // (e is in the list)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (changed_ids[mid] < e) lo = mid + 1; else hi = mid;
    }
    t = changed_nodes[lo];
  } else {
    t = tets_old[e];
    if (imap) { t.x = imap[t.x]; t.y = imap[t.y]; t.z = imap[t.z]; t.w = imap[t.w]; }
  }
  tets_new[pos[e]] = t;
}

__global__ __launch_bounds__(kB) void k_delta_keys(int n, const double* __restrict__ xyz, SlabKeyGeom g, unsigned long long* __restrict__ keys, uint32_t* __restrict__ ids) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  keys[i] = slab_key(g, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
  ids[i] = (uint32_t)i;
}

// elements of the change on every NEW node (caller ids n_old + k), and the new nodes' keys in a handle whose order has the second stage
// (renumber.h: window of the slab key << 10 | clipped count, descending)
__global__ __launch_bounds__(kB) void k_delta_new_counts(int n, const int4* __restrict__ t, int n_old, int* __restrict__ count) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const int4 v = t[i];
  const int id[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (id[k] >= n_old) atomicAdd(&count[id[k] - n_old], 1);
}
__global__ __launch_bounds__(kB) void k_delta_sigma_keys(int n, const int* __restrict__ count, int n_windows, const unsigned long long* __restrict__ win_keys,
                                                         unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  int lo = 0, hi = n_windows;  // windows whose first slab key is <= k
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (win_keys[mid] <= k) lo = mid + 1; else hi = mid;
  }
  const int w = max(lo - 1, 0), c = min(count[i], kSigmaMaxCount);
  keys[i] = ((unsigned long long)w << 10) | (unsigned long long)(kSigmaMaxCount - c);
}

// old internal node i moves up by the number of new nodes whose key is smaller (a new node with an equal key goes behind: its caller id is larger)
__global__ __launch_bounds__(kB) void k_delta_imap(int n_old, const unsigned long long* __restrict__ keys_old, int n_new, const unsigned long long* __restrict__ nks,
                                                   const int* __restrict__ old_of_new_old, int* __restrict__ imap, unsigned long long* __restrict__ keys_out,
                                                   int* __restrict__ old_of_new, int* __restrict__ new_of_old) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n_old) return;
  const unsigned long long k = keys_old[i];
  int lo = 0, hi = n_new;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (nks[mid] < k) lo = mid + 1; else hi = mid;
  }
  const int ni = i + lo, c = old_of_new_old[i];
  imap[i] = ni;
  keys_out[ni] = k;
  old_of_new[ni] = c;
  new_of_old[c] = ni;
}
__global__ __launch_bounds__(kB) void k_delta_newint(int n_new, const unsigned long long* __restrict__ nks, const uint32_t* __restrict__ nvs, int n_old,
                                                     const unsigned long long* __restrict__ keys_old, int* __restrict__ newint, unsigned long long* __restrict__ keys_out,
                                                     int* __restrict__ old_of_new, int* __restrict__ new_of_old) {
  const int j = blockIdx.x * kB + threadIdx.x;
  if (j >= n_new) return;
  const unsigned long long k = nks[j];
  int lo = 0, hi = n_old;  // old nodes with a key <= k
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys_old[mid] <= k) lo = mid + 1; else hi = mid;
  }
  const int ni = j + lo, which = (int)nvs[j], c = n_old + which;
  newint[which] = ni;
  keys_out[ni] = k;
  old_of_new[ni] = c;
  new_of_old[c] = ni;
}

__global__ __launch_bounds__(kB) void k_delta_positions(int n_old, int n_new, const double* __restrict__ x0_old, const double* __restrict__ new_xyz, const int* __restrict__ imap,
                                                        const int* __restrict__ newint, double* __restrict__ x0_new) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  const long long n3 = 3LL * (n_old + n_new);
  if (i >= n3) return;
  const int node = (int)(i / 3), k = (int)(i - 3LL * node);
  if (node < n_old) x0_new[3 * (size_t)(imap ? imap[node] : node) + k] = x0_old[i];
  else x0_new[3 * (size_t)(newint ? newint[node - n_old] : node) + k] = new_xyz[3 * (size_t)(node - n_old) + k];
}

// ---- the plan from the plan (round 5) --------------------------------------------------------------------------------------------
// Round 4 updated the SORTED PAIR LIST the plan was built from and ran the rest of the builder (run lengths, rows, SELL layout, contribution
// table) over all of it again: five passes over 142 MB at 1.1M tets, 0.4 of the 0.8 ms of kernels.  Now the plan itself is the state that
// is updated: the block pattern in CSR form with the length of every block's contribution list (bptr, bcol, ucnt: 20 MB), and the
// contribution table.  Rows no element of the change touches keep their blocks (columns through the monotone node map); the few rows that
// lose or gain contributions are merged one by one with the sorted pairs of the changed and added elements (thousands against millions); the
// table is written once from the old one with the element ids renumbered on the way.  The result is the full rebuild's plan bit for bit
// (tests/test_resync_delta_gpu.py), because it is the same function of the same mesh: blocks = vertex pairs of the elements + one
// marker per node, a block's list = its (element, i, j) words ascending.
struct OldPlan {
  const int* bptr; const int* bcol; const unsigned int* ucnt; const int* slice_off; const int* slot_coff; const uint32_t* contrib;
  int n_nodes;
};
struct FreshPairs {  // key = new row << 32 | new column, sorted by (key, word)
  const unsigned long long* k; const uint32_t* v; int n;
  const int2* row;  // per NEW row: its entries [x, y) (0, 0: none); written by k_touch, which sees every first entry of a row anyway
};
struct RowMaps {
  const int* oldrow;             // new row -> old row, -1 for a new node; nullptr: identity below n_old, new nodes behind
  const int* imap;               // old node -> new node (monotone); nullptr: identity
  const int* newid;              // old element -> new id, -1 if its contributions go (removed or changed)
  const int* touched;            // per NEW row: non-zero if it loses or gains contributions (a word each: rows of one cut plane are neighbours, and a bit map made their atomics collide)
  int n_old;
  const int* tlist;              // the touched rows, each once, in no particular order
  const int* n_touched;          // ... and their number
};
__device__ __forceinline__ int old_row_of(const RowMaps& M, int rn) { return M.oldrow ? M.oldrow[rn] : (rn < M.n_old ? rn : -1); }
__device__ __forceinline__ bool row_touched(const RowMaps& M, int rn) { return M.touched[rn] != 0; }
__device__ __forceinline__ int fresh_lower(const FreshPairs& F, unsigned long long key, int lo, int hi) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (F.k[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// contribution words of old block p (row r) whose element goes; eight words in flight at a time (the loop is a chain of dependent
// loads otherwise: word, then its element's entry of the id table)
__device__ __forceinline__ int dropped_words(const OldPlan& C, const RowMaps& M, int p, int r) {
  const int k = p - C.bptr[r], slot = C.slice_off[r >> 6] + k;
  const int n = (int)C.ucnt[p] - (C.bcol[p] == r ? 1 : 0);
  const uint32_t* w = C.contrib + (size_t)C.slot_coff[slot] * 64 + (r & 63);
  int gone = 0;
  for (int t0 = 0; t0 < n; t0 += 8) {
    uint32_t o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = t0 + j < n ? w[(size_t)(t0 + j) * 64] : 0u;
    int id[8];
#pragma unroll
    for (int j = 0; j < 8; j++) id[j] = t0 + j < n ? M.newid[o[j] >> 4] : 0;
#pragma unroll
    for (int j = 0; j < 8; j++) gone += id[j] < 0 ? 1 : 0;
  }
  return gone;
}
// The blocks of NEW row rn in ascending column order: visit(column, pairs incl. the marker, old block or -1, fresh entries [f0, f1)).
template <class V>
__device__ __forceinline__ void walk_row(const OldPlan& C, const FreshPairs& F, const RowMaps& M, int rn, V& visit) {
  const int r = old_row_of(M, rn);
  int p = r >= 0 ? C.bptr[r] : 0;
  const int pe = r >= 0 ? C.bptr[r + 1] : 0;
  const int2 fr = F.row[rn];
  int f = fr.x;
  const int fe = fr.y;
  while (p < pe || f < fe) {
    const int co = p < pe ? (M.imap ? M.imap[C.bcol[p]] : C.bcol[p]) : 0x7fffffff;
    const int cf = f < fe ? (int)(unsigned int)(F.k[f] & 0xFFFFFFFFULL) : 0x7fffffff;
    const int c = min(co, cf);
    int src = -1, cnt = 0;
    if (co == c) { src = p; cnt = (int)C.ucnt[p] - dropped_words(C, M, p, r); p++; }
    const int f0 = f;
    if (cf == c) while (f < fe && (int)(unsigned int)(F.k[f] & 0xFFFFFFFFULL) == c) f++;
    cnt += f - f0;
    if (cnt > 0) visit(c, cnt, src, f0, f);
  }
}

__global__ __launch_bounds__(kB) void k_newid(int n_old, const unsigned char* __restrict__ estate, const int* __restrict__ pos, int* __restrict__ newid) {
  const int e = blockIdx.x * kB + threadIdx.x;
  if (e < n_old) newid[e] = estate[e] ? -1 : pos[e];
}
__global__ __launch_bounds__(kB) void k_oldrow(int n_old, int n_new_nodes, const int* __restrict__ imap, const int* __restrict__ newint, int* __restrict__ oldrow) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i < n_old) oldrow[imap[i]] = i;
  else if (i < n_old + n_new_nodes) oldrow[newint[i - n_old]] = -1;
}
// rows of the fresh pairs, and the rows of the elements that go (their old nodes through the node map).  The fresh list is sorted by
// row: one atomic per run of a row, not per entry (every entry of a row hits the same word: 99 us of same-address atomics at 1.1M tets)
// appends the rows of the threads with `mine` to the list: ONE atomic on the counter per workgroup (an atomic per first entry: 35 us of
// same-address atomics at 1.1M tets; one per wavefront: still 3,400 of them on one address, 34 us).  Every thread of the workgroup calls it.
__device__ __forceinline__ void append_rows(bool mine, int r, int* __restrict__ tlist, int* __restrict__ count) {
  __shared__ int s_cnt[kB / 64], s_base;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long m = __ballot(mine);
  if (lane == 0) s_cnt[wv] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < kB / 64; w++) total += s_cnt[w];
    s_base = total ? atomicAdd(count, total) : 0;
  }
  __syncthreads();
  if (mine) {
    int base = s_base;
    for (int w = 0; w < wv; w++) base += s_cnt[w];
    tlist[base + __popcll(m & ((1ULL << lane) - 1ULL))] = r;
  }
}
__global__ __launch_bounds__(kB) void k_touch(int n_fresh, const unsigned long long* __restrict__ fk, int n_gone, const int* __restrict__ removed, int n_removed,
                                              const int* __restrict__ changed, const int4* __restrict__ tets_old, const int* __restrict__ imap, int* __restrict__ touched,
                                              int2* __restrict__ frow, int* __restrict__ tlist, int* __restrict__ count) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  bool first = false;
  int r = 0;
  if (i < n_fresh) {
    r = (int)(unsigned int)(fk[i] >> 32);
    const unsigned int before = i > 0 ? (unsigned int)(fk[i - 1] >> 32) : 0xFFFFFFFFu;
    if (before != (unsigned int)r) {
      first = atomicExch(&touched[r], 1) == 0;
      frow[r].x = (int)i;
      if (i > 0) frow[before].y = (int)i;
    }
    if (i == n_fresh - 1) frow[r].y = n_fresh;
  } else if (i < (long long)n_fresh + 4LL * n_gone) {
    const long long j = i - n_fresh;
    const int m = (int)(j >> 2), e = m < n_removed ? removed[m] : changed[m - n_removed];
    const int4 t = tets_old[e];
    const int id[4] = {t.x, t.y, t.z, t.w};
    r = imap ? imap[id[j & 3]] : id[j & 3];
    first = touched[r] == 0 && atomicExch(&touched[r], 1) == 0;
  }
  append_rows(first, r, tlist, count);
}

// ---- a touched row, one wavefront: lanes hold its old blocks and its fresh entries (up to 64 each; longer rows -- hub nodes -- are walked
// by one lane, walk_row).  lower bound over a sorted array held one element per lane, the same steps on every lane (shuffles).
__device__ __forceinline__ int wave_lower_bound(int mine, int n, int key) {
  int pos = 0;
#pragma unroll
  for (int step = 32; step > 0; step >>= 1) {
    const int idx = pos + step - 1;
    const int v = __shfl(mine, idx & 63, 64);
    if (idx < n && v < key) pos += step;
  }
  // (pos counts the elements below key among the first 63; the 64th is looked at on its own)
  const int last = __shfl(mine, 63, 64);
  if (pos == 63 && n == 64 && last < key) pos = 64;
  return pos;
}
struct RowMerge {
  // per lane: an old block (lane < n_old) and a fresh entry (lane < n_fresh)
  int n_old, n_fresh, f0, p0;
  int col_old, total_old, src_old;   // old block of this lane: new column, pairs after the change, its index
  int col_fresh, run_fresh;          // fresh entry of this lane: column, length of the run it heads
  unsigned long long mask_old, mask_new;  // old blocks that stay; fresh runs that make a new block
  int pos_old, pos_new;              // place of this lane's old block / new block in the merged row
  int len;
};
// returns false when the row does not fit the wavefront (the caller falls back to walk_row on one lane)
__device__ __forceinline__ bool merge_row_wave(const OldPlan& C, const FreshPairs& F, const RowMaps& M, int rn, int lane, RowMerge& R) {
  const int r = old_row_of(M, rn);
  R.p0 = r >= 0 ? C.bptr[r] : 0;
  R.n_old = r >= 0 ? C.bptr[r + 1] - R.p0 : 0;
  const int2 fr = F.row[rn];
  R.f0 = fr.x;
  R.n_fresh = fr.y - fr.x;
  if (R.n_old > 64 || R.n_fresh > 64) return false;  // (wave-uniform)
  // fresh entries: columns, heads of runs, run lengths
  R.col_fresh = lane < R.n_fresh ? (int)(unsigned int)(F.k[R.f0 + lane] & 0xFFFFFFFFULL) : 0x7fffffff;
  const int prev = __shfl_up(R.col_fresh, 1, 64);
  const bool head = lane < R.n_fresh && (lane == 0 || prev != R.col_fresh);
  const int after = wave_lower_bound(R.col_fresh, R.n_fresh, R.col_fresh == 0x7fffffff ? 0x7fffffff : R.col_fresh + 1);
  R.run_fresh = after - lane;
  // old blocks: column through the node map, pairs that stay
  R.src_old = R.p0 + lane;
  R.col_old = 0x7fffffff;
  int kept = 0;
  if (lane < R.n_old) {
    const int c = C.bcol[R.src_old];
    R.col_old = M.imap ? M.imap[c] : c;
    kept = (int)C.ucnt[R.src_old] - dropped_words(C, M, R.src_old, r);
  }
  const int below = wave_lower_bound(R.col_fresh, R.n_fresh, R.col_old);                                       // fresh entries of a smaller column
  const int upto = wave_lower_bound(R.col_fresh, R.n_fresh, R.col_old == 0x7fffffff ? 0x7fffffff : R.col_old + 1);
  R.total_old = kept + (lane < R.n_old ? upto - below : 0);
  R.mask_old = __ballot(lane < R.n_old && R.total_old > 0);
  // a fresh run makes a new block where no old block has its column
  const int at = wave_lower_bound(R.col_old, R.n_old, R.col_fresh);
  const int hit = __shfl(R.col_old, at & 63, 64);
  const bool matched = at < R.n_old && hit == R.col_fresh;
  R.mask_new = __ballot(head && !matched);
  const unsigned long long lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
  const unsigned long long lt_below = below == 0 ? 0ULL : (below >= 64 ? ~0ULL : (~0ULL >> (64 - below)));
  const unsigned long long lt_at = at == 0 ? 0ULL : (at >= 64 ? ~0ULL : (~0ULL >> (64 - at)));
  R.pos_old = __popcll(R.mask_old & lt) + __popcll(R.mask_new & lt_below);
  R.pos_new = __popcll(R.mask_new & lt) + __popcll(R.mask_old & lt_at);
  R.len = __popcll(R.mask_old) + __popcll(R.mask_new);
  return true;
}

struct CountBlocks {
  int n = 0;
  __device__ void operator()(int, int, int, int, int) { n++; }
};
// length of every new row (len[n_new] = 0 closes the scan).  An untouched row: a thread; the touched rows: the wavefronts of k_row_len_touched
// take them from the list (a wavefront per row of the whole matrix was 26 rounds of resident wavefronts, each a chain of four loads)
__global__ __launch_bounds__(kB) void k_row_len(int n_new, OldPlan C, RowMaps M, int* __restrict__ len) {
  const int rn = blockIdx.x * kB + threadIdx.x;
  if (rn > n_new) return;
  if (rn == n_new) { len[rn] = 0; return; }
  if (row_touched(M, rn)) return;
  const int r = old_row_of(M, rn);
  len[rn] = C.bptr[r + 1] - C.bptr[r];
}
__global__ __launch_bounds__(kB) void k_row_len_touched(OldPlan C, FreshPairs F, RowMaps M, int* __restrict__ len) {
  const int lane = threadIdx.x & 63, n_waves = gridDim.x * (kB / 64), n_t = M.n_touched[0];
  for (int i = blockIdx.x * (kB / 64) + (threadIdx.x >> 6); i < n_t; i += n_waves) {
    const int rn = M.tlist[i];
    RowMerge R;
    if (merge_row_wave(C, F, M, rn, lane, R)) {
      if (lane == 0) len[rn] = R.len;
    } else if (lane == 0) {
      CountBlocks v;
      walk_row(C, F, M, rn, v);
      len[rn] = v.n;
    }
  }
}
struct WriteBlocks {
  int* bcol; unsigned int* ucnt; int* src; int at;
  __device__ void operator()(int col, int cnt, int from, int, int) { bcol[at] = col; ucnt[at] = (unsigned int)cnt; src[at] = from; at++; }
};
// the blocks of every new row: column, pairs, and the old block its kept words come from (-1: none).  Untouched rows: a wavefront per 64 rows,
// the lanes walking the concatenated blocks of those rows (coalesced); touched rows: a wavefront each, lane = block
__global__ __launch_bounds__(kB) void k_row_blocks(int n_new, OldPlan C, RowMaps M, const int* __restrict__ bptr_new, int* __restrict__ bcol, unsigned int* __restrict__ ucnt,
                                                   int* __restrict__ src) {
  const int r0 = (blockIdx.x * (kB / 64) + (threadIdx.x >> 6)) * 64, lane = threadIdx.x & 63;
  if (r0 >= n_new) return;
  const int rn = min(r0 + lane, n_new);
  const int q_mine = bptr_new[rn];                         // (ascending over the lanes; lanes past the last row hold the end)
  const int q_end = bptr_new[min(r0 + 64, n_new)];
  const bool plain = r0 + lane < n_new && !row_touched(M, r0 + lane);
  const int p_mine = plain ? C.bptr[old_row_of(M, r0 + lane)] : 0;
  const int q0 = __shfl(q_mine, 0, 64);
  for (int qb = q0; qb < q_end; qb += 64) {  // (the same trips on every lane: the shuffles below read every lane's registers)
    const int q = qb + lane;
    // the row of block q: the last lane whose first block is <= q
    int j = wave_lower_bound(q_mine, 64, q + 1) - 1;
    if (j < 0) j = 0;
    const int qj = __shfl(q_mine, j, 64), pj = __shfl(p_mine, j, 64);
    const bool pl = __shfl((int)plain, j, 64) != 0;
    if (!pl || q >= q_end) continue;
    const int p = pj + (q - qj);
    const int c = C.bcol[p];
    bcol[q] = M.imap ? M.imap[c] : c;
    ucnt[q] = C.ucnt[p];
    src[q] = p;
  }
}
__global__ __launch_bounds__(kB) void k_row_blocks_touched(OldPlan C, FreshPairs F, RowMaps M, const int* __restrict__ bptr_new, int* __restrict__ bcol, unsigned int* __restrict__ ucnt,
                                                           int* __restrict__ src) {
  const int lane = threadIdx.x & 63, n_waves = gridDim.x * (kB / 64), n_t = M.n_touched[0];
  for (int i = blockIdx.x * (kB / 64) + (threadIdx.x >> 6); i < n_t; i += n_waves) {
    const int rn = M.tlist[i], q = bptr_new[rn];
    RowMerge R;
    if (merge_row_wave(C, F, M, rn, lane, R)) {
      if ((R.mask_old >> lane) & 1ULL) { bcol[q + R.pos_old] = R.col_old; ucnt[q + R.pos_old] = (unsigned int)R.total_old; src[q + R.pos_old] = R.src_old; }
      if ((R.mask_new >> lane) & 1ULL) { bcol[q + R.pos_new] = R.col_fresh; ucnt[q + R.pos_new] = (unsigned int)R.run_fresh; src[q + R.pos_new] = -1; }
    } else if (lane == 0) {
      WriteBlocks v = {bcol, ucnt, src, q};
      walk_row(C, F, M, rn, v);
    }
  }
}

// The contribution table of the new plan from the old one.  Rows the change does not touch: one wavefront per (new slice, slot), lane =
// row; block k of such a row is block k of its old row, in slot k of the old slice: its list is copied word by word with the element ids
// renumbered (no block table is looked at; the old list ends at its first empty word).  The kernel is a chain word -> id table -> store
// per list entry, so every lane keeps sixteen entries in flight.
__global__ __launch_bounds__(kB) void k_slot_slices(int n_slices, const int* __restrict__ slice_off, int* __restrict__ slot_slice) {
  const int s = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= n_slices) return;
  for (int g = slice_off[s] + lane; g < slice_off[s + 1]; g += 64) slot_slice[g] = s;
}
__global__ __launch_bounds__(kB) void k_table_plain(int n_new, int n_slots, OldPlan C, RowMaps M, const int* __restrict__ slice_off, const int* __restrict__ slot_slice,
                                                    const int* __restrict__ slot_coff, uint32_t* __restrict__ contrib) {
  const int g = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // slot of the new plan
  if (g >= n_slots) return;
  const int s = slot_slice[g], k = g - slice_off[s];
  const int rn = s * 64 + lane;
  const bool row = rn < n_new;
  const bool skip = row && row_touched(M, rn);  // (k_table_touched writes those columns)
  const int r = row && !skip ? old_row_of(M, rn) : -1;
  const int so_old = r >= 0 ? C.slice_off[r >> 6] : 0, w_old = r >= 0 ? C.slice_off[(r >> 6) + 1] - so_old : 0;
  const int coff = slot_coff[g], height = slot_coff[g + 1] - coff;  // wave-uniform
  uint32_t* out = contrib + (size_t)coff * 64 + lane;
  int base = 0, h_old = 0;
  if (k < w_old) { base = C.slot_coff[so_old + k]; h_old = C.slot_coff[so_old + k + 1] - base; }
  const uint32_t* in = C.contrib + (size_t)base * 64 + (r >= 0 ? (r & 63) : 0);
  for (int t0 = 0; t0 < height; t0 += 16) {
    uint32_t o[16];
#pragma unroll
    for (int j = 0; j < 16; j++) o[j] = t0 + j < h_old ? in[(size_t)(t0 + j) * 64] : kNoContrib;
    int id[16];
#pragma unroll
    for (int j = 0; j < 16; j++) id[j] = o[j] != kNoContrib ? M.newid[o[j] >> 4] : 0;
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (t0 + j < height && !skip) out[(size_t)(t0 + j) * 64] = o[j] != kNoContrib ? (((uint32_t)id[j] << 4) | (o[j] & 15u)) : kNoContrib;
  }
}
// ... and the touched rows: one wavefront per row, lane = slot of the row; the kept words of the old block and the fresh words of the block,
// both ascending, merged; the rest of the slot's column empty
__global__ __launch_bounds__(kB) void k_table_touched(int n_new, OldPlan C, FreshPairs F, RowMaps M, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                      const unsigned int* __restrict__ ucnt, const int* __restrict__ src, const int* __restrict__ slice_off,
                                                      const int* __restrict__ slot_coff, uint32_t* __restrict__ contrib) {
  const int lane = threadIdx.x & 63, n_waves = gridDim.x * (kB / 64), n_t = M.n_touched[0];
  for (int it = blockIdx.x * (kB / 64) + (threadIdx.x >> 6); it < n_t; it += n_waves) {
  const int rn = M.tlist[it];
  const int s = rn >> 6, so = slice_off[s], w = slice_off[s + 1] - so;
  const int first = bptr[rn], len = bptr[rn + 1] - first;
  const int r = old_row_of(M, rn);
  const int so_old = r >= 0 ? C.slice_off[r >> 6] : 0, first_old = r >= 0 ? C.bptr[r] : 0;
  const int2 fr = F.row[rn];
  for (int k = lane; k < w; k += 64) {
    const int coff = slot_coff[so + k], height = slot_coff[so + k + 1] - coff;
    uint32_t* out = contrib + (size_t)coff * 64 + (rn & 63);
    int t = 0;
    if (k < len) {
      const int col = bcol[first + k], from = src[first + k];
      const uint32_t* in = nullptr;
      int n_o = 0;
      if (from >= 0) {
        in = C.contrib + (size_t)C.slot_coff[so_old + (from - first_old)] * 64 + (r & 63);
        n_o = (int)C.ucnt[from] - (C.bcol[from] == r ? 1 : 0);
      }
      const unsigned long long key = ((unsigned long long)(unsigned int)rn << 32) | (unsigned int)col;
      int f = fresh_lower(F, key, fr.x, fr.y);
      int fe = fresh_lower(F, key + 1ULL, f, fr.y);
      if (fe > f && F.v[fe - 1] == kNoContrib) fe--;  // (the marker of a new node's diagonal block: last of its run, not a contribution)
      uint32_t fw = f < fe ? F.v[f] : 0xFFFFFFFFu;    // next fresh word
      // the old block eight words at a time (their loads, then their id-table entries, in flight together); a kept word goes out behind
      // the fresh words that are smaller
      for (int i0 = 0; i0 < n_o; i0 += 8) {
        uint32_t o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = i0 + j < n_o ? in[(size_t)(i0 + j) * 64] : 0u;
        int id[8];
#pragma unroll
        for (int j = 0; j < 8; j++) id[j] = i0 + j < n_o ? M.newid[o[j] >> 4] : -1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (id[j] < 0) continue;
          const uint32_t wo = ((uint32_t)id[j] << 4) | (o[j] & 15u);
          while (fw < wo) {
            out[(size_t)t * 64] = fw;
            t++;
            f++;
            fw = f < fe ? F.v[f] : 0xFFFFFFFFu;
          }
          out[(size_t)t * 64] = wo;
          t++;
        }
      }
      while (f < fe) { out[(size_t)t * 64] = F.v[f]; t++; f++; }
    }
    for (; t < height; t++) out[(size_t)t * 64] = kNoContrib;
  }
  }
}

// widest slice (the element-major assembly's limit is decided from it)
__global__ __launch_bounds__(kB) void k_max_width(int n_slices, const int* __restrict__ width, int* __restrict__ out) {
  const int i = blockIdx.x * kB + threadIdx.x;
  int w = i < n_slices ? width[i] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o, 64));
  if ((threadIdx.x & 63) == 0 && w > 0) atomicMax(out, w);
}

int bits_of(long long n) {
  int b = 1;
  while ((1LL << b) < n) b++;
  return b;
}

// the 16 pairs of every changed and added element, ascending in the new element id (changed ones keep an id below every added one), then
// the markers of the new nodes; key = new row << 32 | new column
__global__ __launch_bounds__(kB) void k_delta_new_pairs(int n_changed, int n_added, int n_new_nodes, int n_kept, int n_nodes_old, const int* __restrict__ changed_ids,
                                                        const int* __restrict__ pos, const int4* __restrict__ tets_new, const int* __restrict__ newint,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  const long long n_tp = 16LL * (n_changed + n_added);
  if (i < n_tp) {
    const int m = (int)(i >> 4), ij = (int)(i & 15);
    const int e = m < n_changed ? pos[changed_ids[m]] : n_kept + (m - n_changed);
    const int4 t = tets_new[e];
    const int id[4] = {t.x, t.y, t.z, t.w};
    keys[i] = ((unsigned long long)(unsigned int)id[ij >> 2] << 32) | (unsigned int)id[ij & 3];
    vals[i] = ((uint32_t)e << 4) | (uint32_t)ij;
  } else if (i < n_tp + n_new_nodes) {
    const int k = (int)(i - n_tp);
    const unsigned int r = (unsigned int)(newint ? newint[k] : n_nodes_old + k);
    keys[i] = ((unsigned long long)r << 32) | r;
    vals[i] = kNoContrib;
  }
}

}  // namespace

int delta_upload(hipStream_t s, int n_tets_old, int n_removed, const int* removed, int n_changed, const int* changed_ids, const int* changed_nodes, int n_added,
                 const int* added, int n_new_nodes, const double* new_xyz, MeshDelta& D, PlanWorkspace& W) {
  D.n_tets_old = n_tets_old; D.n_removed = n_removed; D.n_changed = n_changed; D.n_added = n_added; D.n_new_nodes = n_new_nodes;
  D.n_kept = n_tets_old - n_removed;
  D.mapped = false;
  const int o_chg = pad4(n_removed), o_cn = o_chg + pad4(n_changed), o_add = o_cn + 4 * n_changed, total = o_add + 4 * n_added + 4;
  std::vector<int> stage((size_t)total, 0);
  if (n_removed) memcpy(stage.data(), removed, sizeof(int) * (size_t)n_removed);
  if (n_changed) memcpy(stage.data() + o_chg, changed_ids, sizeof(int) * (size_t)n_changed);
  if (n_changed) memcpy(stage.data() + o_cn, changed_nodes, sizeof(int) * 4 * (size_t)n_changed);
  if (n_added) memcpy(stage.data() + o_add, added, sizeof(int) * 4 * (size_t)n_added);
  FB_TRY(D.ints.reserve((size_t)total));
  FB_HIP(hipMemcpyAsync(D.ints.p, stage.data(), sizeof(int) * (size_t)total, hipMemcpyHostToDevice, s));  // (pageable source: complete on return)
  D.removed = D.ints.p; D.changed_ids = D.ints.p + o_chg;
  D.changed_nodes = reinterpret_cast<int4*>(D.ints.p + o_cn); D.added = reinterpret_cast<int4*>(D.ints.p + o_add);
  FB_TRY(D.new_xyz.reserve((size_t)std::max(1, 3 * n_new_nodes)));
  if (n_new_nodes) FB_HIP(hipMemcpyAsync(D.new_xyz.p, new_xyz, sizeof(double) * 3 * (size_t)n_new_nodes, hipMemcpyHostToDevice, s));
  FB_TRY(D.estate.reserve((size_t)n_tets_old + 1));
  FB_HIP(hipMemsetAsync(D.estate.p, 0, (size_t)n_tets_old + 1, s));
  if (n_removed + n_changed) {
    hipLaunchKernelGGL(k_delta_mark, grid_for(n_removed + n_changed), dim3(kB), 0, s, n_removed, D.removed, n_changed, D.changed_ids, D.estate.p);
    FB_HIP(hipGetLastError());
  }
  FB_TRY(D.pos.reserve((size_t)n_tets_old + 1));
  const auto stays = rocprim::make_transform_iterator(static_cast<const unsigned char*>(D.estate.p), Stays());
  size_t bytes = 0;
  FB_HIP(rocprim::exclusive_scan(nullptr, bytes, stays, D.pos.p, 0, (size_t)n_tets_old, rocprim::plus<int>(), s));
  FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::exclusive_scan(W.temp.p, bytes, stays, D.pos.p, 0, (size_t)n_tets_old, rocprim::plus<int>(), s));
  return FB_OK;
}

int delta_relabel_nodes(hipStream_t s, MeshDelta& D, int n_nodes, const int* map) {
  // (changed_nodes and added lie back to back in the staging buffer)
  const int n = D.n_changed + D.n_added;
  if (n == 0) return FB_OK;
  hipLaunchKernelGGL(k_delta_relabel, grid_for(n), dim3(kB), 0, s, n, D.changed_nodes, n_nodes, map);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int delta_tets(hipStream_t s, const MeshDelta& D, const int4* tets_old, const int* imap, int4* tets_new) {
  hipLaunchKernelGGL(k_delta_tets, grid_for(D.n_tets_old), dim3(kB), 0, s, D.n_tets_old, tets_old, D.estate.p, D.pos.p, imap, D.n_changed, D.changed_ids, D.changed_nodes, tets_new);
  FB_HIP(hipGetLastError());
  if (D.n_added) FB_HIP(hipMemcpyAsync(tets_new + D.n_kept, D.added, sizeof(int4) * (size_t)D.n_added, hipMemcpyDeviceToDevice, s));
  return FB_OK;
}

int delta_node_order(hipStream_t s, MeshDelta& D, int n_old, const SlabKeyGeom& g, const unsigned long long* keys_old, const int* old_of_new_old, DevBuf<int>& old_of_new,
                     DevBuf<int>& new_of_old, PlanWorkspace& W, int n_windows, const unsigned long long* win_keys) {
  const int n_new = D.n_new_nodes, n = n_old + n_new;
  FB_TRY(D.imap.reserve((size_t)std::max(1, n_old)));
  FB_TRY(D.newint.reserve((size_t)std::max(1, n_new)));
  FB_TRY(D.node_keys.reserve((size_t)n));
  FB_TRY(old_of_new.alloc((size_t)n));
  FB_TRY(new_of_old.alloc((size_t)n));
  FB_TRY(D.nk.reserve((size_t)std::max(1, n_new)));
  FB_TRY(D.nks.reserve((size_t)std::max(1, n_new)));
  FB_TRY(D.nv.reserve((size_t)std::max(1, n_new)));
  FB_TRY(D.nvs.reserve((size_t)std::max(1, n_new)));
  if (n_new) {
    hipLaunchKernelGGL(k_delta_keys, grid_for(n_new), dim3(kB), 0, s, n_new, D.new_xyz.p, g, D.nk.p, D.nv.p);
    FB_HIP(hipGetLastError());
    unsigned key_bits = (unsigned)(g.bits[0] + g.bits[1] + g.bits[2]);
    if (n_windows > 0) {  // the order has the second stage: keys_old are its keys, and the new nodes get theirs
      FB_TRY(D.new_count.reserve((size_t)n_new));
      FB_HIP(hipMemsetAsync(D.new_count.p, 0, sizeof(int) * (size_t)n_new, s));
      const int n_el = D.n_changed + D.n_added;   // (changed_nodes and added lie back to back, still in the caller's ids)
      if (n_el) hipLaunchKernelGGL(k_delta_new_counts, grid_for(n_el), dim3(kB), 0, s, n_el, D.changed_nodes, n_old, D.new_count.p);
      hipLaunchKernelGGL(k_delta_sigma_keys, grid_for(n_new), dim3(kB), 0, s, n_new, D.new_count.p, n_windows, win_keys, D.nk.p);
      FB_HIP(hipGetLastError());
      key_bits = 10;
      while ((1LL << (key_bits - 10)) < n_windows) key_bits++;
    }
    size_t bytes = 0;
    FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, D.nk.p, D.nks.p, D.nv.p, D.nvs.p, (size_t)n_new, 0u, key_bits, s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::radix_sort_pairs(W.temp.p, bytes, D.nk.p, D.nks.p, D.nv.p, D.nvs.p, (size_t)n_new, 0u, key_bits, s));
  }
  hipLaunchKernelGGL(k_delta_imap, grid_for(n_old), dim3(kB), 0, s, n_old, keys_old, n_new, D.nks.p, old_of_new_old, D.imap.p, D.node_keys.p, old_of_new.p, new_of_old.p);
  FB_HIP(hipGetLastError());
  if (n_new) {
    hipLaunchKernelGGL(k_delta_newint, grid_for(n_new), dim3(kB), 0, s, n_new, D.nks.p, D.nvs.p, n_old, keys_old, D.newint.p, D.node_keys.p, old_of_new.p, new_of_old.p);
    FB_HIP(hipGetLastError());
  }
  D.mapped = true;
  return FB_OK;
}

int delta_positions(hipStream_t s, const MeshDelta& D, int n_old, const double* x0_old, double* x0_new) {
  hipLaunchKernelGGL(k_delta_positions, grid_for(3LL * (n_old + D.n_new_nodes)), dim3(kB), 0, s, n_old, D.n_new_nodes, x0_old, D.new_xyz.p, D.mapped ? D.imap.p : nullptr,
                     D.mapped ? D.newint.p : nullptr, x0_new);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int delta_reserve(hipStream_t s, MeshDelta& D, PlanWorkspace& W, long long n_fresh) {
  const size_t n = (size_t)std::max<long long>(1, n_fresh);
  FB_TRY(D.nk.reserve(n)); FB_TRY(D.nks.reserve(n)); FB_TRY(D.nv.reserve(n)); FB_TRY(D.nvs.reserve(n));
  size_t bytes = 0;
  FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, D.nk.p, D.nks.p, D.nv.p, D.nvs.p, n, 0u, 64u, s));
  return W.temp.reserve(std::max<size_t>(bytes, 16));
}

int delta_plan(hipStream_t s, MeshDelta& D, const OldPlanArrays& O, const int4* tets_old, const int4* tets_new, int n_new, DevicePlan& out, PlanWorkspace& W) {
  const int n_old = O.n_nodes;
  const long long n_fresh = 16LL * (D.n_changed + D.n_added) + D.n_new_nodes;
  if (n_fresh >= (1LL << 31) || 16LL * D.n_tets_new() + n_new >= (1LL << 31)) return fail(FB_EINVAL, "mesh too large for the device plan builder");
  if ((long long)D.n_tets_new() >= (1LL << 28)) return fail(FB_EINVAL, "too many tets for the packed contribution word");
  // 1. the pairs of the changed and added elements and the markers of the new nodes, sorted (stable: ascending words inside a block, a marker last)
  FB_TRY(D.nk.reserve((size_t)std::max<long long>(1, n_fresh)));
  FB_TRY(D.nks.reserve((size_t)std::max<long long>(1, n_fresh)));
  FB_TRY(D.nv.reserve((size_t)std::max<long long>(1, n_fresh)));
  FB_TRY(D.nvs.reserve((size_t)std::max<long long>(1, n_fresh)));
  if (n_fresh > 0) {
    hipLaunchKernelGGL(k_delta_new_pairs, grid_for(n_fresh), dim3(kB), 0, s, D.n_changed, D.n_added, D.n_new_nodes, D.n_kept, n_old, D.changed_ids, D.pos.p, tets_new,
                       D.mapped ? D.newint.p : nullptr, D.nk.p, D.nv.p);
    FB_HIP(hipGetLastError());
    const unsigned key_bits = 32u + (unsigned)bits_of(n_new);
    size_t bytes = 0;
    FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, D.nk.p, D.nks.p, D.nv.p, D.nvs.p, (size_t)n_fresh, 0u, key_bits, s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::radix_sort_pairs(W.temp.p, bytes, D.nk.p, D.nks.p, D.nv.p, D.nvs.p, (size_t)n_fresh, 0u, key_bits, s));
  }
  // 2. element ids, row map, touched rows
  FB_TRY(D.newid.reserve((size_t)D.n_tets_old + 1));
  hipLaunchKernelGGL(k_newid, grid_for(D.n_tets_old), dim3(kB), 0, s, D.n_tets_old, D.estate.p, D.pos.p, D.newid.p);
  const bool mapped = D.mapped && D.n_new_nodes > 0;
  if (mapped) {
    FB_TRY(D.oldrow.reserve((size_t)n_new));
    hipLaunchKernelGGL(k_oldrow, grid_for(n_new), dim3(kB), 0, s, n_old, D.n_new_nodes, D.imap.p, D.newint.p, D.oldrow.p);
  }
  // one buffer, one fill: the number of touched rows | a word per row | an int2 per row (its fresh entries) | the list of touched rows
  const size_t tw = ((size_t)n_new + 4) & ~(size_t)1;
  FB_TRY(D.touched.reserve(2 + tw + 2 * (size_t)n_new + 2 + (size_t)n_new));
  FB_HIP(hipMemsetAsync(D.touched.p, 0, sizeof(unsigned int) * (2 + tw + 2 * (size_t)n_new + 2), s));
  int* touched = reinterpret_cast<int*>(D.touched.p) + 2;
  int2* frow = reinterpret_cast<int2*>(D.touched.p + 2 + tw);
  int* tlist = reinterpret_cast<int*>(D.touched.p) + 2 + tw + 2 * (size_t)n_new + 2;
  const int n_gone = D.n_removed + D.n_changed;
  if (n_fresh + n_gone > 0) {
    hipLaunchKernelGGL(k_touch, grid_for(n_fresh + 4LL * n_gone), dim3(kB), 0, s, (int)n_fresh, D.nks.p, n_gone, D.removed, D.n_removed, D.changed_ids, tets_old,
                       mapped ? D.imap.p : nullptr, touched, frow, tlist, reinterpret_cast<int*>(D.touched.p));
  }
  FB_HIP(hipGetLastError());
  OldPlan C = {O.bptr, O.bcol, O.ucnt, O.slice_off, O.slot_coff, O.contrib, n_old};
  FreshPairs F = {D.nks.p, D.nvs.p, (int)n_fresh, frow};
  RowMaps M = {mapped ? D.oldrow.p : nullptr, mapped ? D.imap.p : nullptr, D.newid.p, touched, n_old, tlist, reinterpret_cast<const int*>(D.touched.p)};
  const dim3 touched_grid(1024);  // (4,096 wavefronts take the touched rows from the list in turn)
  // 3. the pattern: row lengths, their scan, the blocks
  FB_TRY(D.len.reserve((size_t)n_new + 1));
  hipLaunchKernelGGL(k_row_len, grid_for(n_new + 1), dim3(kB), 0, s, n_new, C, M, D.len.p);
  hipLaunchKernelGGL(k_row_len_touched, touched_grid, dim3(kB), 0, s, C, F, M, D.len.p);
  FB_HIP(hipGetLastError());
  FB_TRY(out.bptr->alloc((size_t)n_new + 1));
  size_t bytes = 0;
  FB_HIP(rocprim::exclusive_scan(nullptr, bytes, D.len.p, out.bptr->p, 0, (size_t)n_new + 1, rocprim::plus<int>(), s));
  FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::exclusive_scan(W.temp.p, bytes, D.len.p, out.bptr->p, 0, (size_t)n_new + 1, rocprim::plus<int>(), s));
  int nb = 0;
  FB_TRY(out.bptr->download(&nb, 1, s, (size_t)n_new));
  out.n_blocks = nb;
  FB_TRY(out.bcol->alloc((size_t)std::max(1, nb)));
  FB_TRY(out.blk_slot->alloc((size_t)std::max(1, nb)));
  FB_TRY(out.ucnt_keep->alloc((size_t)std::max(1, nb)));
  FB_TRY(D.src.reserve((size_t)std::max(1, nb)));
  hipLaunchKernelGGL(k_row_blocks, dim3((unsigned)(((n_new + 63) / 64 + kB / 64 - 1) / (kB / 64))), dim3(kB), 0, s, n_new, C, M, out.bptr->p, out.bcol->p, out.ucnt_keep->p, D.src.p);
  hipLaunchKernelGGL(k_row_blocks_touched, touched_grid, dim3(kB), 0, s, C, F, M, out.bptr->p, out.bcol->p, out.ucnt_keep->p, D.src.p);
  FB_HIP(hipGetLastError());
  // 4. SELL layout, slot table, list heights and offsets: the second half of the builder, on the new pattern
  FB_TRY(plan_layout_from_csr(s, n_new, out.ucnt_keep->p, false, out, W));
  // 5. the contribution table from the old one
  FB_TRY(out.contrib->alloc(std::max<size_t>(1, (size_t)out.n_crows * kSliceRows)));
  FB_TRY(D.slot_slice.reserve((size_t)std::max(1, out.n_slots)));
  hipLaunchKernelGGL(k_slot_slices, dim3((unsigned)std::max(1, (out.n_slices + kB / 64 - 1) / (kB / 64))), dim3(kB), 0, s, out.n_slices, out.slice_off->p, D.slot_slice.p);
  hipLaunchKernelGGL(k_table_plain, dim3((unsigned)std::max(1, (out.n_slots + kB / 64 - 1) / (kB / 64))), dim3(kB), 0, s, n_new, out.n_slots, C, M, out.slice_off->p, D.slot_slice.p,
                     out.slot_coff->p, out.contrib->p);
  hipLaunchKernelGGL(k_table_touched, touched_grid, dim3(kB), 0, s, n_new, C, F, M, out.bptr->p, out.bcol->p, out.ucnt_keep->p, D.src.p, out.slice_off->p, out.slot_coff->p,
                     out.contrib->p);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

}  // namespace fb
