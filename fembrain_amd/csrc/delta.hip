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
struct FreshPairs { const unsigned long long* k; const uint32_t* v; int n; };  // key = new row << 32 | new column, sorted by (key, word)
struct RowMaps {
  const int* oldrow;             // new row -> old row, -1 for a new node; nullptr: identity below n_old, new nodes behind
  const int* imap;               // old node -> new node (monotone); nullptr: identity
  const int* newid;              // old element -> new id, -1 if its contributions go (removed or changed)
  const unsigned int* touched;   // a bit per NEW row: it loses or gains contributions
  int n_old;
};
__device__ __forceinline__ int old_row_of(const RowMaps& M, int rn) { return M.oldrow ? M.oldrow[rn] : (rn < M.n_old ? rn : -1); }
__device__ __forceinline__ bool row_touched(const RowMaps& M, int rn) { return (M.touched[rn >> 5] >> (rn & 31)) & 1u; }
__device__ __forceinline__ int fresh_lower(const FreshPairs& F, unsigned long long key) {
  int lo = 0, hi = F.n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (F.k[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// contribution words of old block p (row r) whose element goes
__device__ __forceinline__ int dropped_words(const OldPlan& C, const RowMaps& M, int p, int r) {
  const int k = p - C.bptr[r], slot = C.slice_off[r >> 6] + k;
  const int n = (int)C.ucnt[p] - (C.bcol[p] == r ? 1 : 0);
  const uint32_t* w = C.contrib + (size_t)C.slot_coff[slot] * 64 + (r & 63);
  int gone = 0;
  for (int t = 0; t < n; t++) gone += M.newid[w[(size_t)t * 64] >> 4] < 0 ? 1 : 0;
  return gone;
}
// The blocks of NEW row rn in ascending column order: visit(column, pairs incl. the marker, old block or -1, fresh entries [f0, f1)).
template <class V>
__device__ __forceinline__ void walk_row(const OldPlan& C, const FreshPairs& F, const RowMaps& M, int rn, V& visit) {
  const int r = old_row_of(M, rn);
  int p = r >= 0 ? C.bptr[r] : 0;
  const int pe = r >= 0 ? C.bptr[r + 1] : 0;
  int f = fresh_lower(F, (unsigned long long)(unsigned int)rn << 32);
  const int fe = fresh_lower(F, (unsigned long long)((unsigned int)rn + 1u) << 32);
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
// rows of the fresh pairs, and the rows of the elements that go (their old nodes through the node map)
__global__ __launch_bounds__(kB) void k_touch(int n_fresh, const unsigned long long* __restrict__ fk, int n_gone, const int* __restrict__ removed, int n_removed,
                                              const int* __restrict__ changed, const int4* __restrict__ tets_old, const int* __restrict__ imap, unsigned int* __restrict__ touched) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  if (i < n_fresh) {
    const unsigned int r = (unsigned int)(fk[i] >> 32);
    atomicOr(&touched[r >> 5], 1u << (r & 31));
  } else if (i < (long long)n_fresh + 4LL * n_gone) {
    const long long j = i - n_fresh;
    const int m = (int)(j >> 2), e = m < n_removed ? removed[m] : changed[m - n_removed];
    const int4 t = tets_old[e];
    const int id[4] = {t.x, t.y, t.z, t.w};
    const int r = imap ? imap[id[j & 3]] : id[j & 3];
    atomicOr(&touched[r >> 5], 1u << (r & 31));
  }
}

struct CountBlocks {
  int n = 0;
  __device__ void operator()(int, int, int, int, int) { n++; }
};
// length of every new row (len[n_new] = 0 closes the scan)
__global__ __launch_bounds__(kB) void k_row_len(int n_new, OldPlan C, FreshPairs F, RowMaps M, int* __restrict__ len) {
  const int rn = blockIdx.x * kB + threadIdx.x;
  if (rn > n_new) return;
  if (rn == n_new) { len[rn] = 0; return; }
  if (!row_touched(M, rn)) {
    const int r = old_row_of(M, rn);
    len[rn] = C.bptr[r + 1] - C.bptr[r];
    return;
  }
  CountBlocks v;
  walk_row(C, F, M, rn, v);
  len[rn] = v.n;
}
struct WriteBlocks {
  int* bcol; unsigned int* ucnt; int* src; int at;
  __device__ void operator()(int col, int cnt, int from, int, int) { bcol[at] = col; ucnt[at] = (unsigned int)cnt; src[at] = from; at++; }
};
// the blocks of every new row: column, pairs, and the old block its kept words come from (-1: none)
__global__ __launch_bounds__(kB) void k_row_blocks(int n_new, OldPlan C, FreshPairs F, RowMaps M, const int* __restrict__ bptr_new, int* __restrict__ bcol, unsigned int* __restrict__ ucnt,
                                                   int* __restrict__ src) {
  const int rn = blockIdx.x * kB + threadIdx.x;
  if (rn >= n_new) return;
  const int q = bptr_new[rn];
  if (!row_touched(M, rn)) {
    const int r = old_row_of(M, rn), p = C.bptr[r], n = C.bptr[r + 1] - p;
    for (int k = 0; k < n; k++) {
      const int c = C.bcol[p + k];
      bcol[q + k] = M.imap ? M.imap[c] : c;
      ucnt[q + k] = C.ucnt[p + k];
      src[q + k] = p + k;
    }
    return;
  }
  WriteBlocks v = {bcol, ucnt, src, q};
  walk_row(C, F, M, rn, v);
}

// The contribution table of the new plan from the old one: a workgroup of four wavefronts per new slice, wavefront w the slots w, w + 4, ...;
// lane = row.  An untouched row copies its lists word by word (element ids renumbered); a touched one merges the kept words of the old
// block with the fresh words of the block, both ascending.
__global__ __launch_bounds__(kB) void k_table_from_table(int n_new, int n_slices, OldPlan C, FreshPairs F, RowMaps M, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                         const unsigned int* __restrict__ ucnt, const int* __restrict__ src, const int* __restrict__ slice_off,
                                                         const int* __restrict__ slot_coff, const int* __restrict__ slot_ccnt, uint32_t* __restrict__ contrib) {
  const int s = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (s >= n_slices) return;
  const int rn = s * 64 + lane;
  const int so = slice_off[s], w = slice_off[s + 1] - so;
  const int first = rn < n_new ? bptr[rn] : 0, len = rn < n_new ? bptr[rn + 1] - first : 0;
  const int r = rn < n_new ? old_row_of(M, rn) : -1;
  const bool slow = rn < n_new && row_touched(M, rn);
  const int so_old = r >= 0 ? C.slice_off[r >> 6] : 0, first_old = r >= 0 ? C.bptr[r] : 0;
  for (int k = wv; k < w; k += kB / 64) {
    const int height = slot_ccnt[so + k];  // wave-uniform
    uint32_t* out = contrib + (size_t)slot_coff[so + k] * 64 + lane;
    int cnt = 0, from = -1, col = -1;
    if (k < len) { col = bcol[first + k]; cnt = (int)ucnt[first + k] - (col == rn ? 1 : 0); from = src[first + k]; }
    if (!slow) {
      // (an untouched row: block k of the new row is block k of the old one, in slot k of the old slice)
      const uint32_t* in = from >= 0 ? C.contrib + (size_t)C.slot_coff[so_old + (from - first_old)] * 64 + (r & 63) : nullptr;
      for (int t = 0; t < height; t++) {
        uint32_t word = kNoContrib;
        if (t < cnt) {
          const uint32_t o = in[(size_t)t * 64];
          word = ((uint32_t)M.newid[o >> 4] << 4) | (o & 15u);
        }
        out[(size_t)t * 64] = word;
      }
    } else {
      int t = 0;
      if (k < len) {
        const uint32_t* in = nullptr;
        int n_o = 0;
        if (from >= 0) {
          in = C.contrib + (size_t)C.slot_coff[so_old + (from - first_old)] * 64 + (r & 63);
          n_o = (int)C.ucnt[from] - (C.bcol[from] == r ? 1 : 0);
        }
        const unsigned long long key = ((unsigned long long)(unsigned int)rn << 32) | (unsigned int)col;
        int f = fresh_lower(F, key);
        int fe = fresh_lower(F, key + 1ULL);
        if (fe > f && F.v[fe - 1] == kNoContrib) fe--;  // (the marker of a new node's diagonal block: last of its run, not a contribution)
        int i = 0;
        uint32_t wo = 0;
        bool have = false;
        for (;;) {
          while (!have && i < n_o) {  // next kept word of the old block
            const uint32_t o = in[(size_t)i * 64];
            i++;
            const int id = M.newid[o >> 4];
            if (id >= 0) { wo = ((uint32_t)id << 4) | (o & 15u); have = true; }
          }
          if (!have && f >= fe) break;
          uint32_t word;
          if (have && (f >= fe || wo < F.v[f])) { word = wo; have = false; }
          else word = F.v[f++];
          out[(size_t)t * 64] = word;
          t++;
        }
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
  const size_t tw = ((size_t)n_new + 31) / 32 + 1;
  FB_TRY(D.touched.reserve(tw));
  FB_HIP(hipMemsetAsync(D.touched.p, 0, sizeof(unsigned int) * tw, s));
  const int n_gone = D.n_removed + D.n_changed;
  if (n_fresh + n_gone > 0) {
    hipLaunchKernelGGL(k_touch, grid_for(n_fresh + 4LL * n_gone), dim3(kB), 0, s, (int)n_fresh, D.nks.p, n_gone, D.removed, D.n_removed, D.changed_ids, tets_old,
                       mapped ? D.imap.p : nullptr, D.touched.p);
  }
  FB_HIP(hipGetLastError());
  OldPlan C = {O.bptr, O.bcol, O.ucnt, O.slice_off, O.slot_coff, O.contrib, n_old};
  FreshPairs F = {D.nks.p, D.nvs.p, (int)n_fresh};
  RowMaps M = {mapped ? D.oldrow.p : nullptr, mapped ? D.imap.p : nullptr, D.newid.p, D.touched.p, n_old};
  // 3. the pattern: row lengths, their scan, the blocks
  FB_TRY(D.len.reserve((size_t)n_new + 1));
  hipLaunchKernelGGL(k_row_len, grid_for(n_new + 1), dim3(kB), 0, s, n_new, C, F, M, D.len.p);
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
  hipLaunchKernelGGL(k_row_blocks, grid_for(n_new), dim3(kB), 0, s, n_new, C, F, M, out.bptr->p, out.bcol->p, out.ucnt_keep->p, D.src.p);
  FB_HIP(hipGetLastError());
  // 4. SELL layout, slot table, list heights and offsets: the second half of the builder, on the new pattern
  FB_TRY(plan_layout_from_csr(s, n_new, out.ucnt_keep->p, false, out, W));
  // 5. the contribution table from the old one
  FB_TRY(out.contrib->alloc(std::max<size_t>(1, (size_t)out.n_crows * kSliceRows)));
  hipLaunchKernelGGL(k_table_from_table, dim3((unsigned)std::max(1, out.n_slices)), dim3(kB), 0, s, n_new, out.n_slices, C, F, M, out.bptr->p, out.bcol->p, out.ucnt_keep->p,
                     D.src.p, out.slice_off->p, out.slot_coff->p, out.slot_ccnt->p, out.contrib->p);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

}  // namespace fb
