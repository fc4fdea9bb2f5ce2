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
constexpr uint32_t kDropped = 0xFFFFFFFEu;  // (no contribution word: element ids stay below 2^28)

inline int pad4(int x) { return (x + 3) & ~3; }
inline dim3 grid_for(long long n) { return dim3((unsigned)std::max<long long>(1, (n + kB - 1) / kB)); }

__global__ __launch_bounds__(kB) void k_delta_mark(int n_removed, const int* __restrict__ removed, int n_changed, const int* __restrict__ changed, unsigned char* __restrict__ estate) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i < n_removed) estate[removed[i]] = 1;
  else if (i < n_removed + n_changed) estate[changed[i - n_removed]] = 2;
}

struct TileKept {
  __device__ int operator()(int gone) const { return 2048 - gone; }  // (kTile)
};
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

// ---- the pair list ----
struct PackDesc { int narrow, cb, span, col_bits; };
__device__ __forceinline__ void unpack_key(const PackDesc& p, unsigned long long k, int& row, int& col) {
  if (p.narrow) {
    row = (int)(k >> p.cb);
    col = row + (int)(k & ((1ULL << p.cb) - 1ULL)) - p.span;
  } else {
    row = (int)(k >> p.col_bits);
    col = (int)(k & ((1ULL << p.col_bits) - 1ULL));
  }
}
__device__ __forceinline__ unsigned long long pack_key(const PackDesc& p, int row, int col) {
  return p.narrow ? (((unsigned long long)(unsigned int)row << p.cb) | (unsigned long long)(unsigned int)(col - row + p.span))
                  : (((unsigned long long)(unsigned int)row << p.col_bits) | (unsigned long long)(unsigned int)col);
}

// an entry of the old list in the terms of the new one: nodes through imap, the element by its new id, the key in the new packing;
// entries of removed and changed elements are marked for the selection to drop
template <typename KIn, typename KOut>
struct PairXform {
  PackDesc in, out;
  const int* imap;
  const unsigned char* estate;
  const int* pos;
  __device__ rocprim::tuple<KOut, uint32_t> operator()(const rocprim::tuple<KIn, uint32_t>& t) const {
    const uint32_t v = rocprim::get<1>(t);
    int row, col;
    unpack_key(in, (unsigned long long)rocprim::get<0>(t), row, col);
    if (imap) { row = imap[row]; col = imap[col]; }
    uint32_t vo = v;
    if (v != kNoContrib) {
      const uint32_t e = v >> 4;
      vo = estate[e] ? kDropped : (((uint32_t)pos[e] << 4) | (v & 15u));
    }
    return rocprim::make_tuple((KOut)pack_key(out, row, col), vo);
  }
};
struct NotDropped {
  template <typename T>
  __device__ bool operator()(const T& t) const { return rocprim::get<1>(t) != kDropped; }
};
// (row, column), then the contribution word: ascending (element, i, j) inside a block, the marker of a diagonal block last
struct PairLess {
  template <typename T>
  __device__ bool operator()(const T& a, const T& b) const {
    return rocprim::get<0>(a) < rocprim::get<0>(b) || (rocprim::get<0>(a) == rocprim::get<0>(b) && rocprim::get<1>(a) < rocprim::get<1>(b));
  }
};

// the 16 pairs of every changed and added element, ascending in the new element id (changed ones keep an id below every added one), then
// the markers of the new nodes
template <typename KOut>
__global__ __launch_bounds__(kB) void k_delta_new_pairs(int n_changed, int n_added, int n_new_nodes, int n_kept, int n_nodes_old, PackDesc out, const int* __restrict__ changed_ids,
                                                        const int* __restrict__ pos, const int4* __restrict__ tets_new, const int* __restrict__ newint, KOut* __restrict__ keys,
                                                        uint32_t* __restrict__ vals) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  const long long n_tp = 16LL * (n_changed + n_added);
  if (i < n_tp) {
    const int m = (int)(i >> 4), ij = (int)(i & 15);
    const int e = m < n_changed ? pos[changed_ids[m]] : n_kept + (m - n_changed);
    const int4 t = tets_new[e];
    const int id[4] = {t.x, t.y, t.z, t.w};
    keys[i] = (KOut)pack_key(out, id[ij >> 2], id[ij & 3]);
    vals[i] = ((uint32_t)e << 4) | (uint32_t)ij;
  } else if (i < n_tp + n_new_nodes) {
    const int k = (int)(i - n_tp);
    const int r = newint ? newint[k] : n_nodes_old + k;
    keys[i] = (KOut)pack_key(out, r, r);
    vals[i] = kNoContrib;
  }
}

// ---- the update of the list in two passes of our own (the library's select over a transforming zip iterator took 490 us at 1M tets,
// its merge another 110): tiles of kTile old entries; pass 1 marks the entries that go and counts them per tile; pass 2 ranks the others
// (tile base + rank inside the tile) and puts the new entries that fall between two of them in their places.  New entries below the
// first or above the last old entry are copied by k_upd_ends.
constexpr int kTile = 2048, kTileItems = kTile / kB;
template <typename K>
__device__ __forceinline__ bool pair_less(K ka, uint32_t va, K kb, uint32_t vb) { return ka < kb || (ka == kb && va < vb); }

template <typename KIn, typename KOut>
__device__ __forceinline__ void xform_entry(const PackDesc& in, const PackDesc& out, const int* __restrict__ imap, const int* __restrict__ pos, KIn k, uint32_t v, KOut* ko,
                                            uint32_t* vo) {
  int row, col;
  unpack_key(in, (unsigned long long)k, row, col);
  if (imap) { row = imap[row]; col = imap[col]; }
  *ko = (KOut)pack_key(out, row, col);
  *vo = v == kNoContrib ? v : (((uint32_t)pos[v >> 4] << 4) | (v & 15u));
}

template <typename K>
__device__ __forceinline__ int lower_bound_pairs(const K* __restrict__ k, const uint32_t* __restrict__ v, int lo, int hi, K key, uint32_t val) {
  while (lo < hi) {  // first index whose entry is not less than (key, val)
    const int mid = (lo + hi) >> 1;
    if (pair_less<K>(k[mid], v[mid], key, val)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Pass 1: the entries that go are FOUND rather than looked for -- the 16 entries of every removed or changed element sit at (row, column,
// element << 4 | ij) of the old list: a binary search each, a bit in `drop`, a count per tile (a pass over the whole list that asked every
// entry whether its element stays took 87 us at 1.1M tets; this one 290,000 searches).  tets_old: the old element list in the old ids.
template <typename KIn>
__global__ __launch_bounds__(kB) void k_upd_dropped(int n_removed, const int* __restrict__ removed, int n_changed, const int* __restrict__ changed, const int4* __restrict__ tets_old,
                                                    PackDesc in, int n_a, const KIn* __restrict__ ka, const uint32_t* __restrict__ va, unsigned int* __restrict__ drop) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  if (i >= 16LL * (n_removed + n_changed)) return;
  const int m = (int)(i >> 4), ij = (int)(i & 15);
  const int e = m < n_removed ? removed[m] : changed[m - n_removed];
  const int4 t = tets_old[e];
  const int id[4] = {t.x, t.y, t.z, t.w};
  const KIn key = (KIn)pack_key(in, id[ij >> 2], id[ij & 3]);
  const uint32_t val = ((uint32_t)e << 4) | (uint32_t)ij;
  const int at = lower_bound_pairs<KIn>(ka, va, 0, n_a, key, val);  // (it is there)
  atomicOr(&drop[at >> 5], 1u << (at & 31));
}
// ... and the count per tile from the bits (a counter per tile bumped by every entry: 121 us of same-address atomics)
__global__ __launch_bounds__(kB) void k_upd_tile_counts(int n_tiles, const unsigned int* __restrict__ drop, int* __restrict__ tile_drop) {
  const int b = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= n_tiles) return;
  int c = __popc(drop[(size_t)b * (kTile / 32) + lane]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0 && c) atomicAdd(&tile_drop[b], c);  // (on top of the last tile's share beyond the list)
}

// per tile: the number of new entries below its first staying entry (-1: the tile keeps nothing); bounds[0] = that of the first such tile,
// bounds[1] = the number of new entries below the LAST staying entry of the whole list
template <typename KIn, typename KOut>
__global__ __launch_bounds__(kB) void k_upd_bounds(int n_tiles, long long n_stay, int n_a, const KIn* __restrict__ ka, const uint32_t* __restrict__ va, PackDesc in, PackDesc out,
                                                   const int* __restrict__ imap, const int* __restrict__ pos, const unsigned int* __restrict__ drop, const int* __restrict__ tile_drop,
                                                   const int* __restrict__ tile_base, int n_b, const KOut* __restrict__ kb, const uint32_t* __restrict__ vb, int* __restrict__ lo,
                                                   int* __restrict__ bounds) {
  const int b = blockIdx.x * kB + threadIdx.x;
  if (b >= n_tiles) return;
  const int kept = kTile - tile_drop[b];
  if (kept == 0) { lo[b] = -1; return; }
  int f = b * kTile;
  while (drop[f >> 5] >> (f & 31) & 1u) f++;  // (a staying entry exists)
  KOut k; uint32_t v;
  xform_entry<KIn, KOut>(in, out, imap, pos, ka[f], va[f], &k, &v);
  const int l = lower_bound_pairs<KOut>(kb, vb, 0, n_b, k, v);
  lo[b] = l;
  if (tile_base[b] == 0) bounds[0] = l;
  if ((long long)tile_base[b] + kept == n_stay) {
    int g = min(n_a, (b + 1) * kTile) - 1;
    while (drop[g >> 5] >> (g & 31) & 1u) g--;
    xform_entry<KIn, KOut>(in, out, imap, pos, ka[g], va[g], &k, &v);
    bounds[1] = lower_bound_pairs<KOut>(kb, vb, 0, n_b, k, v);
  }
}

// Pass 2: ranks the staying entries (tile base + rank inside the tile) and puts the new entries that fall between two of them in their places
template <typename KIn, typename KOut>
__global__ __launch_bounds__(kB) void k_upd_merge(long long n_a, const KIn* __restrict__ ka, const uint32_t* __restrict__ va, PackDesc in, PackDesc out,
                                                  const int* __restrict__ imap, const int* __restrict__ pos, const unsigned int* __restrict__ drop, int n_tiles,
                                                  const int* __restrict__ tile_drop, const int* __restrict__ tile_base, const int* __restrict__ lo_of, const int* __restrict__ bounds,
                                                  int n_b, const KOut* __restrict__ kb, const uint32_t* __restrict__ vb, KOut* __restrict__ kc, uint32_t* __restrict__ vc) {
  __shared__ KOut ck[kTile];
  __shared__ uint32_t cv[kTile];
  __shared__ int s_wave[kB / 64], s_run, s_lo, s_hi;
  const int kept = kTile - tile_drop[blockIdx.x];
  if (kept == 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    s_run = 0;
    s_lo = lo_of[blockIdx.x];
    int hi = bounds[1];
    for (int t = blockIdx.x + 1; t < n_tiles; t++)
      if (lo_of[t] >= 0) { hi = lo_of[t]; break; }
    s_hi = hi;
  }
  __syncthreads();
  const long long base = (long long)blockIdx.x * kTile;
#pragma unroll 1
  for (int i = 0; i < kTileItems; i++) {
    const long long idx = base + i * kB + threadIdx.x;
    KOut k = 0; uint32_t v = 0;
    const bool keep = idx < n_a && !(drop[idx >> 5] >> (idx & 31) & 1u);
    if (keep) xform_entry<KIn, KOut>(in, out, imap, pos, ka[idx], va[idx], &k, &v);
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ULL << lane) - 1ULL));
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    int off = s_run;
    for (int w = 0; w < wv; w++) off += s_wave[w];
    if (keep) { ck[off + before] = k; cv[off + before] = v; }
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < kB / 64; w++) t += s_wave[w]; s_run += t; }
    __syncthreads();
  }
  const int lo = s_lo, hi = s_hi;
  const long long gbase = tile_base[blockIdx.x];
  // the staying entries: behind the new ones that are smaller
  for (int q = threadIdx.x; q < kept; q += kB) {
    const KOut k = ck[q]; const uint32_t v = cv[q];
    const int nb = lower_bound_pairs<KOut>(kb, vb, lo, hi, k, v);
    kc[gbase + q + nb] = k; vc[gbase + q + nb] = v;
  }
  // the new entries between this tile's first staying entry and the next tile's: behind the staying ones that are smaller
  for (int j = lo + threadIdx.x; j < hi; j += kB) {
    const KOut k = kb[j]; const uint32_t v = vb[j];
    const int na = lower_bound_pairs<KOut>(ck, cv, 0, kept, k, v);
    kc[gbase + na + j] = k; vc[gbase + na + j] = v;
  }
}

// new entries below the first and above the last staying entry
template <typename KOut>
__global__ __launch_bounds__(kB) void k_upd_ends(long long n_stay, const int* __restrict__ bounds, int n_b, const KOut* __restrict__ kb, const uint32_t* __restrict__ vb,
                                                 KOut* __restrict__ kc, uint32_t* __restrict__ vc) {
  const int j = blockIdx.x * kB + threadIdx.x;
  if (j >= n_b) return;
  if (j < bounds[0]) { kc[j] = kb[j]; vc[j] = vb[j]; }
  else if (j >= bounds[1]) { kc[n_stay + j] = kb[j]; vc[n_stay + j] = vb[j]; }
}

int bits_of(long long n) {
  int b = 1;
  while ((1LL << b) < n) b++;
  return b;
}

template <typename KIn, typename KOut>
int update_pairs(hipStream_t s, MeshDelta& D, const int4* tets_old, const int4* tets_new, int n_nodes_old, const PackDesc& pin, const PackDesc& pout, unsigned key_bits, long long n_old_pairs,
                 long long n_new_pairs, PlanWorkspace& W) {
  const long long n_stay = n_old_pairs - 16LL * (D.n_removed + D.n_changed);
  const long long n_fresh = 16LL * (D.n_changed + D.n_added) + D.n_new_nodes;
  if (n_stay + n_fresh != n_new_pairs) return fail(FB_EINVAL, "internal: pair count of the change does not add up");
  // 1. the new entries, sorted among themselves (stable: ascending contribution words inside a block, its marker last)
  FB_TRY(D.nk.reserve((size_t)std::max<long long>(1, n_fresh)));
  FB_TRY(D.nks.reserve((size_t)std::max<long long>(1, n_fresh)));
  FB_TRY(D.nv.reserve((size_t)std::max<long long>(1, n_fresh)));
  FB_TRY(D.nvs.reserve((size_t)std::max<long long>(1, n_fresh)));
  KOut* nk = reinterpret_cast<KOut*>(D.nk.p);
  KOut* nks = reinterpret_cast<KOut*>(D.nks.p);
  if (n_fresh > 0) {
    hipLaunchKernelGGL(k_delta_new_pairs<KOut>, grid_for(n_fresh), dim3(kB), 0, s, D.n_changed, D.n_added, D.n_new_nodes, D.n_kept, n_nodes_old, pout, D.changed_ids, D.pos.p,
                       tets_new, D.mapped ? D.newint.p : nullptr, nk, D.nv.p);
    FB_HIP(hipGetLastError());
    size_t bytes = 0;
    FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, nk, nks, D.nv.p, D.nvs.p, (size_t)n_fresh, 0u, key_bits, s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::radix_sort_pairs(W.temp.p, bytes, nk, nks, D.nv.p, D.nvs.p, (size_t)n_fresh, 0u, key_bits, s));
  }
  const int* imap = D.mapped ? D.imap.p : nullptr;
  static const bool library = getenv("FEMBRAIN_DELTA_LIBRARY") && atoi(getenv("FEMBRAIN_DELTA_LIBRARY")) != 0;  // development aid: rocprim select + merge
  if (!library) {
    // 2. old list (keys_s / vals_s) + new entries -> keys / vals, then the buffers change places
    const int n_tiles = (int)((n_old_pairs + kTile - 1) / kTile);
    FB_TRY(W.keys.reserve((size_t)n_new_pairs));
    FB_TRY(W.vals.reserve((size_t)n_new_pairs));
    FB_TRY(D.tile_i.reserve((size_t)3 * (n_tiles + 1) + 2));
    FB_TRY(D.drop_bits.reserve((size_t)n_tiles * (kTile / 32)));
    int* tile_drop = D.tile_i.p;
    int* tile_base = tile_drop + (n_tiles + 1);
    int* tile_lo = tile_base + (n_tiles + 1);
    int* bounds = tile_lo + (n_tiles + 1);
    const KIn* ka = reinterpret_cast<const KIn*>(W.keys_s.p);
    KOut* kc = reinterpret_cast<KOut*>(W.keys.p);
    FB_HIP(hipMemsetAsync(D.drop_bits.p, 0, sizeof(unsigned int) * (size_t)n_tiles * (kTile / 32), s));
    FB_HIP(hipMemsetAsync(tile_drop, 0, sizeof(int) * (size_t)(n_tiles + 1), s));
    // (the part of the last tile beyond the list, and the scan's closing element, count as gone)
    const int tail[2] = {(int)((long long)n_tiles * kTile - n_old_pairs), kTile};
    FB_HIP(hipMemcpyAsync(tile_drop + (n_tiles - 1), tail, sizeof tail, hipMemcpyHostToDevice, s));
    const int4* tets_old_p = tets_old;
    if (D.n_removed + D.n_changed > 0) {
      hipLaunchKernelGGL(k_upd_dropped<KIn>, grid_for(16LL * (D.n_removed + D.n_changed)), dim3(kB), 0, s, D.n_removed, D.removed, D.n_changed, D.changed_ids, tets_old_p, pin,
                         (int)n_old_pairs, ka, W.vals_s.p, D.drop_bits.p);
      hipLaunchKernelGGL(k_upd_tile_counts, dim3((unsigned)((n_tiles + kB / 64 - 1) / (kB / 64))), dim3(kB), 0, s, n_tiles, D.drop_bits.p, tile_drop);
      FB_HIP(hipGetLastError());
    }
    const auto tile_kept = rocprim::make_transform_iterator(static_cast<const int*>(tile_drop), TileKept());
    size_t bytes = 0;
    FB_HIP(rocprim::exclusive_scan(nullptr, bytes, tile_kept, tile_base, 0, (size_t)n_tiles + 1, rocprim::plus<int>(), s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::exclusive_scan(W.temp.p, bytes, tile_kept, tile_base, 0, (size_t)n_tiles + 1, rocprim::plus<int>(), s));
    const int init_bounds[2] = {0, 0};
    FB_HIP(hipMemcpyAsync(bounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((k_upd_bounds<KIn, KOut>), grid_for(n_tiles), dim3(kB), 0, s, n_tiles, n_stay, (int)n_old_pairs, ka, W.vals_s.p, pin, pout, imap, D.pos.p, D.drop_bits.p,
                       tile_drop, tile_base, (int)n_fresh, nks, D.nvs.p, tile_lo, bounds);
    FB_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_upd_merge<KIn, KOut>), dim3(n_tiles), dim3(kB), 0, s, n_old_pairs, ka, W.vals_s.p, pin, pout, imap, D.pos.p, D.drop_bits.p, n_tiles, tile_drop,
                       tile_base, tile_lo, bounds, (int)n_fresh, nks, D.nvs.p, kc, W.vals.p);
    FB_HIP(hipGetLastError());
    if (n_fresh > 0) {
      hipLaunchKernelGGL(k_upd_ends<KOut>, grid_for(n_fresh), dim3(kB), 0, s, n_stay, bounds, (int)n_fresh, nks, D.nvs.p, kc, W.vals.p);
      FB_HIP(hipGetLastError());
    }
    W.keys.swap(W.keys_s);
    W.vals.swap(W.vals_s);
    return FB_OK;
  }
  // (library path) the old list, transformed, without the dropped entries: keys_s / vals_s -> keys / vals
  FB_TRY(W.nruns.reserve(1));
  FB_TRY(W.keys.reserve((size_t)std::max<long long>(1, n_stay)));
  FB_TRY(W.vals.reserve((size_t)std::max<long long>(1, n_stay)));
  {
    const auto in = rocprim::make_transform_iterator(
        rocprim::make_zip_iterator(rocprim::make_tuple(reinterpret_cast<const KIn*>(W.keys_s.p), static_cast<const uint32_t*>(W.vals_s.p))),
        PairXform<KIn, KOut>{pin, pout, imap, D.estate.p, D.pos.p});
    auto out = rocprim::make_zip_iterator(rocprim::make_tuple(reinterpret_cast<KOut*>(W.keys.p), W.vals.p));
    size_t bytes = 0;
    FB_HIP(rocprim::select(nullptr, bytes, in, out, W.nruns.p, (size_t)n_old_pairs, NotDropped(), s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::select(W.temp.p, bytes, in, out, W.nruns.p, (size_t)n_old_pairs, NotDropped(), s));
  }
  // merged into keys_s / vals_s (the old list is no longer needed: the buffers may grow)
  FB_TRY(W.keys_s.reserve((size_t)n_new_pairs));
  FB_TRY(W.vals_s.reserve((size_t)n_new_pairs));
  {
    const auto a = rocprim::make_zip_iterator(rocprim::make_tuple(reinterpret_cast<const KOut*>(W.keys.p), static_cast<const uint32_t*>(W.vals.p)));
    const auto b = rocprim::make_zip_iterator(rocprim::make_tuple(static_cast<const KOut*>(nks), static_cast<const uint32_t*>(D.nvs.p)));
    auto out = rocprim::make_zip_iterator(rocprim::make_tuple(reinterpret_cast<KOut*>(W.keys_s.p), W.vals_s.p));
    size_t bytes = 0;
    FB_HIP(rocprim::merge(nullptr, bytes, a, b, out, (size_t)n_stay, (size_t)n_fresh, PairLess(), s));
    FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
    FB_HIP(rocprim::merge(W.temp.p, bytes, a, b, out, (size_t)n_stay, (size_t)n_fresh, PairLess(), s));
  }
  // (the unsorted buffers must hold the new list too when the next full build sorts into them: they are reserved there)
  return FB_OK;
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

int delta_sorted_pairs(hipStream_t s, MeshDelta& D, const int4* tets_old, const int4* tets_new, int n_nodes_new, int span, PlanWorkspace& W) {
  SortedPairs& S = W.sorted;
  if (!S.valid) return fail(FB_EINVAL, "internal: no sorted pair list to update");
  const int n_nodes_old = S.n_nodes;
  const long long n_new_pairs = 16LL * D.n_tets_new() + n_nodes_new;
  if (n_new_pairs >= (1LL << 31)) return fail(FB_EINVAL, "mesh too large for the device plan builder (%lld pairs)", n_new_pairs);
  if ((long long)D.n_tets_new() >= (1LL << 28)) return fail(FB_EINVAL, "too many tets for the packed contribution word");
  const PackDesc pin = {S.narrow ? 1 : 0, S.cb, S.span, S.col_bits};
  // the key width of the new list: the rule of build_plan_device
  int cb32 = 1;
  while (span >= 0 && cb32 < 31 && (1LL << cb32) < 2LL * span + 1) cb32++;
  const int rb32 = bits_of(n_nodes_new), col_bits = bits_of(n_nodes_new), row_bits = bits_of(n_nodes_new);
  const bool narrow = span >= 0 && span < n_nodes_new && rb32 + cb32 <= 32 && !(getenv("FEMBRAIN_PLAN_KEYS64") && atoi(getenv("FEMBRAIN_PLAN_KEYS64")) != 0);
  const PackDesc pout = {narrow ? 1 : 0, cb32, span, col_bits};
  const unsigned key_bits = narrow ? (unsigned)(rb32 + cb32) : (unsigned)(row_bits + col_bits);
  S.valid = false;  // (until the new list is complete)
  int rc;
  if (S.narrow && narrow) rc = update_pairs<unsigned int, unsigned int>(s, D, tets_old, tets_new, n_nodes_old, pin, pout, key_bits, S.n_pairs, n_new_pairs, W);
  else if (S.narrow) rc = update_pairs<unsigned int, unsigned long long>(s, D, tets_old, tets_new, n_nodes_old, pin, pout, key_bits, S.n_pairs, n_new_pairs, W);
  else if (narrow) rc = update_pairs<unsigned long long, unsigned int>(s, D, tets_old, tets_new, n_nodes_old, pin, pout, key_bits, S.n_pairs, n_new_pairs, W);
  else rc = update_pairs<unsigned long long, unsigned long long>(s, D, tets_old, tets_new, n_nodes_old, pin, pout, key_bits, S.n_pairs, n_new_pairs, W);
  FB_TRY(rc);
  S.narrow = narrow; S.cb = cb32; S.span = span; S.col_bits = col_bits;
  S.n_pairs = n_new_pairs; S.n_nodes = n_nodes_new; S.n_tets = D.n_tets_new();
  S.valid = true;
  return FB_OK;
}

}  // namespace fb
