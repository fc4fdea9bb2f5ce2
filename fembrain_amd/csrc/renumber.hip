// Slab-order renumbering of the mesh nodes (renumber.h): a bounding-box reduction, one key per node, one radix sort of
// (key, node) pairs, the inverse map, and the widest element under both orders -- ~0.1 ms at 176k nodes / 1M tets.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <cmath>
#include <numeric>

#include <rocprim/rocprim.hpp>

#include "renumber.h"

namespace fb {
namespace {

constexpr int kB = 256;
constexpr int kBoxBlocks = 64;

__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// per block: min and max of each coordinate (6 doubles); the host folds the kBoxBlocks partials (min / max are exact in any order)
__global__ __launch_bounds__(kB) void k_bbox(int n, const double* __restrict__ xyz, double* __restrict__ part) {
  __shared__ double sh[kB / 64][6];
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
    for (int k = 0; k < 3; k++) {
      const double v = xyz[3 * (size_t)i + k];
      lo[k] = fmin(lo[k], v); hi[k] = fmax(hi[k], v);   // (fmin / fmax drop a NaN operand: a NaN coordinate is the flat-element check's to report)
    }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = 0; k < 3; k++) {
    const double a = wave_min_d(lo[k]), b = wave_max_d(hi[k]);
    if (lane == 0) { sh[w][k] = a; sh[w][3 + k] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = sh[0][threadIdx.x];
    for (int q = 1; q < kB / 64; q++) v = threadIdx.x < 3 ? fmin(v, sh[q][threadIdx.x]) : fmax(v, sh[q][threadIdx.x]);
    part[6 * blockIdx.x + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(kB) void k_slab_keys(int n, const double* __restrict__ xyz, SlabKeyGeom g, unsigned long long* __restrict__ keys, uint32_t* __restrict__ ids) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  keys[i] = slab_key(g, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
  ids[i] = (uint32_t)i;
}

// elements on every node (ids outside the range are the plan builder's to report)
__global__ __launch_bounds__(kB) void k_incident_count(int n_tets, const int4* __restrict__ tets, int n_nodes, int* __restrict__ count) {
  const int e = blockIdx.x * kB + threadIdx.x;
  if (e >= n_tets) return;
  const int4 t = tets[e];
  const int id[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; i++)
    if ((unsigned int)id[i] >= (unsigned int)n_nodes) return;
#pragma unroll
  for (int i = 0; i < 4; i++) atomicAdd(&count[id[i]], 1);
}
// per slice of 64 consecutive nodes of the order `ids`: the largest count and the sum (one wavefront per slice)
__global__ __launch_bounds__(kB) void k_slice_counts(int n, const uint32_t* __restrict__ ids, const int* __restrict__ count, int2* __restrict__ out) {
  const int s = blockIdx.x * (kB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s * 64 >= n) return;
  const int l = s * 64 + lane;
  int c = l < n ? count[ids[l]] : 0, m = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m = max(m, __shfl_xor(m, o, 64)); c += __shfl_xor(c, o, 64); }
  if (lane == 0) out[s] = make_int2(m, c);
}
__global__ __launch_bounds__(kB) void k_sigma_keys(int n, int window, const uint32_t* __restrict__ ids, const int* __restrict__ count, unsigned long long* __restrict__ keys,
                                                   uint32_t* __restrict__ vals) {
  const int l = blockIdx.x * kB + threadIdx.x;
  if (l >= n) return;
  const uint32_t id = ids[l];
  const int c = min(count[id], kSigmaMaxCount);
  keys[l] = ((unsigned long long)(l / window) << 10) | (unsigned long long)(kSigmaMaxCount - c);
  vals[l] = id;
}
__global__ __launch_bounds__(kB) void k_window_keys(int n_windows, int window, const unsigned long long* __restrict__ slab_keys, unsigned long long* __restrict__ out) {
  const int w = blockIdx.x * kB + threadIdx.x;
  if (w < n_windows) out[w] = slab_keys[(size_t)w * window];
}

__global__ __launch_bounds__(kB) void k_invert(int n, const uint32_t* __restrict__ old_of_new, int* __restrict__ o2n_out, int* __restrict__ new_of_old) {
  const int l = blockIdx.x * kB + threadIdx.x;
  if (l >= n) return;
  const int o = (int)old_of_new[l];
  o2n_out[l] = o;
  new_of_old[o] = l;
}

// out[0] = widest element, out[2..3] (one 64-bit word) = sum of the widths.  A fixed, small grid walks the list and ends with ONE pair of
// atomics per workgroup: same-address atomics serialise at 12-17 ns each on this part, and a pair per wavefront of a thread-per-element
// grid (15,600 wavefronts at 1M tets) made this trivial pass the second most expensive kernel of a re-sync (368 us).
constexpr int kSpanBlocks = 256;
__global__ __launch_bounds__(kB) void k_tet_span(int n_tets, const int4* __restrict__ tets, int n_nodes, const int* __restrict__ map, int* __restrict__ out) {
  __shared__ int s_max[kB / 64];
  __shared__ unsigned long long s_sum[kB / 64];
  int span = 0;
  unsigned long long sum = 0;
  for (int e = blockIdx.x * kB + threadIdx.x; e < n_tets; e += gridDim.x * kB) {
    const int4 t = tets[e];
    const int v[4] = {t.x, t.y, t.z, t.w};
    bool ok = true;
    for (int i = 0; i < 4; i++) ok = ok && (unsigned int)v[i] < (unsigned int)n_nodes;
    if (ok) {
      int lo = 0x7fffffff, hi = -1;
      for (int i = 0; i < 4; i++) {
        const int m = map ? map[v[i]] : v[i];
        lo = min(lo, m); hi = max(hi, m);
      }
      span = max(span, hi - lo);
      sum += (unsigned long long)(hi - lo);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { span = max(span, __shfl_xor(span, o, 64)); sum += __shfl_xor(sum, o, 64); }
  if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = span; s_sum[threadIdx.x >> 6] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kB / 64; w++) { span = max(span, s_max[w]); sum += s_sum[w]; }
    if (span > 0) {
      atomicMax(out, span);
      atomicAdd(reinterpret_cast<unsigned long long*>(out + 2), sum);
    }
  }
}

__global__ __launch_bounds__(kB) void k_relabel(int n_tets, int4* __restrict__ tets, int n_nodes, const int* __restrict__ map) {
  const int e = blockIdx.x * kB + threadIdx.x;
  if (e >= n_tets) return;
  int4 t = tets[e];
  int* v = &t.x;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if ((unsigned int)v[i] < (unsigned int)n_nodes) v[i] = map[v[i]];  // (an id out of range stays as it is: the plan builder reports it)
  tets[e] = t;
}

__global__ __launch_bounds__(kB) void k_gather_nodes(long long n, int width, const double* __restrict__ src, const int* __restrict__ map, double* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  if (i >= n * width) return;
  const long long node = i / width;
  const int c = (int)(i - node * width);
  dst[i] = src[(size_t)map[node] * width + c];
}

__global__ __launch_bounds__(kB) void k_scatter_nodes(long long n, int width, const double* __restrict__ src, const int* __restrict__ map, double* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * kB + threadIdx.x;
  if (i >= n * width) return;
  const long long node = i / width;
  const int c = (int)(i - node * width);
  dst[(size_t)map[node] * width + c] = src[i];
}

// the slab order alone pads by more than a tenth: 64 x the largest count of every slice against the sum of the counts
bool sigma_wanted(int n_nodes, long long padded, long long used) {
  const char* e = getenv("FEMBRAIN_SIGMA");  // 0 / 1: never / always (development)
  if (e) return atoi(e) != 0;
  return n_nodes >= kRenumberMinNodes && padded > 0 && 10 * (padded - used) > padded;  // (small meshes are all surface, and no one waits for them)
}

int bits_for(long long cells) {  // bits that hold the cell indices 0 .. cells - 1
  int b = 1;
  while ((1LL << b) < cells) b++;
  return b;
}

}  // namespace

bool slab_key_geometry(int n_nodes, const double lo[3], const double hi[3], SlabKeyGeom* out) {
  double ext[3], emax = 0.0;
  for (int k = 0; k < 3; k++) {
    ext[k] = hi[k] - lo[k];
    if (!(ext[k] >= 0.0) || !std::isfinite(ext[k])) return false;
    emax = std::max(emax, ext[k]);
  }
  if (!(emax > 0.0) || n_nodes < 2) return false;
  // a flat or thin body still has a volume to its cells: no extent counts as less than a thousandth of the longest
  double vol = 1.0;
  for (int k = 0; k < 3; k++) vol *= std::max(ext[k], 1e-3 * emax);
  double h = 0.8 * std::cbrt(vol / (double)n_nodes);
  if (!(h > 0.0) || !std::isfinite(h)) return false;
  // longest axis first; an axis moves ahead of an earlier one only if it is 5 % longer (a cube keeps x, y, z whatever the last bits
  // of its extents say)
  int ax[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++)
    for (int j = i; j > 0 && ext[ax[j]] > 1.05 * ext[ax[j - 1]]; j--) std::swap(ax[j], ax[j - 1]);
  for (;;) {
    int total = 0;
    for (int a = 0; a < 3; a++) {
      out->bits[a] = bits_for((long long)(ext[ax[a]] / h + 0.5) + 2);
      total += out->bits[a];
    }
    if (total <= 62) break;
    h *= 2.0;
  }
  for (int k = 0; k < 3; k++) { out->lo[k] = lo[k]; out->axis[k] = ax[k]; }
  out->inv_h = 1.0 / h;
  return true;
}

int tet_span_device(hipStream_t s, int n_tets, const int4* d_tets, int n_nodes, const int* d_new_of_old, PlanWorkspace& W, int* span, double* mean) {
  FB_TRY(W.flags.reserve(4));
  FB_HIP(hipMemsetAsync(W.flags.p, 0, 4 * sizeof(int), s));
  hipLaunchKernelGGL(k_tet_span, dim3((unsigned)std::max(1, std::min(kSpanBlocks, (n_tets + kB - 1) / kB))), dim3(kB), 0, s, n_tets, d_tets, n_nodes, d_new_of_old, W.flags.p);
  FB_HIP(hipGetLastError());
  int out[4];
  FB_HIP(hipMemcpyAsync(out, W.flags.p, sizeof out, hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  *span = out[0];
  unsigned long long sum;
  memcpy(&sum, out + 2, sizeof sum);
  if (mean) *mean = n_tets > 0 ? (double)sum / (double)n_tets : 0.0;
  return FB_OK;
}

int renumber_decide(hipStream_t s, int mode, int n_nodes, int n_tets, const int4* d_tets, PlanWorkspace& W, Renumbering& R, bool* want) {
  R.clear();
  *want = false;
  if (mode == FB_RENUMBER_OFF || n_nodes < 2 || n_tets < 1) return FB_OK;
  FB_TRY(tet_span_device(s, n_tets, d_tets, n_nodes, nullptr, W, &R.span_before, &R.mean_before));
  R.span_after = R.span_before; R.mean_after = R.mean_before;
  *want = mode == FB_RENUMBER_ON || (n_nodes >= kRenumberMinNodes && R.span_before > renumber_span_limit(n_nodes));
  return FB_OK;
}

int renumber_build(hipStream_t s, int mode, int n_nodes, int n_tets, const int4* d_tets, const double* d_xyz, PlanWorkspace& W, Renumbering& R, bool force_sigma) {
  R.active = false;
  // bounding box
  FB_TRY(W.temp.reserve(sizeof(double) * 6 * kBoxBlocks));
  double* d_part = reinterpret_cast<double*>(W.temp.p);
  hipLaunchKernelGGL(k_bbox, dim3(kBoxBlocks), dim3(kB), 0, s, n_nodes, d_xyz, d_part);
  FB_HIP(hipGetLastError());
  double part[6 * kBoxBlocks];
  FB_HIP(hipMemcpyAsync(part, d_part, sizeof part, hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int b = 0; b < kBoxBlocks; b++)
    for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], part[6 * b + k]); hi[k] = std::fmax(hi[k], part[6 * b + 3 + k]); }
  SlabKeyGeom g;
  if (!slab_key_geometry(n_nodes, lo, hi, &g)) return FB_OK;  // nothing to sort by: the caller's order stands
  // keys, stable sort, inverse
  FB_TRY(W.keys.reserve((size_t)n_nodes));
  FB_TRY(W.keys_s.reserve((size_t)n_nodes));
  FB_TRY(W.vals.reserve((size_t)n_nodes));
  FB_TRY(W.vals_s.reserve((size_t)n_nodes));
  const dim3 ng((unsigned)((n_nodes + kB - 1) / kB));
  hipLaunchKernelGGL(k_slab_keys, ng, dim3(kB), 0, s, n_nodes, d_xyz, g, W.keys.p, W.vals.p);
  FB_HIP(hipGetLastError());
  const unsigned key_bits = (unsigned)(g.bits[0] + g.bits[1] + g.bits[2]);
  size_t bytes = 0;
  FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, W.keys.p, W.keys_s.p, W.vals.p, W.vals_s.p, (size_t)n_nodes, 0u, key_bits, s));
  FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
  FB_HIP(rocprim::radix_sort_pairs(W.temp.p, bytes, W.keys.p, W.keys_s.p, W.vals.p, W.vals_s.p, (size_t)n_nodes, 0u, key_bits, s));
  // second stage (renumber.h: sigma_window): would the slab order pad the matrix by more than a tenth?
  R.sigma = false;
  R.n_windows = 0;
  {
    const int n_slices = (n_nodes + 63) / 64;
    FB_TRY(R.d_count.alloc((size_t)n_nodes));
    FB_TRY(R.d_count.zero(s));
    hipLaunchKernelGGL(k_incident_count, dim3((unsigned)((n_tets + kB - 1) / kB)), dim3(kB), 0, s, n_tets, d_tets, n_nodes, R.d_count.p);
    FB_TRY(W.temp.reserve(sizeof(int2) * (size_t)n_slices));
    int2* d_sc = reinterpret_cast<int2*>(W.temp.p);
    hipLaunchKernelGGL(k_slice_counts, dim3((unsigned)((n_slices + kB / 64 - 1) / (kB / 64))), dim3(kB), 0, s, n_nodes, W.vals_s.p, R.d_count.p, d_sc);
    FB_HIP(hipGetLastError());
    std::vector<int2> sc((size_t)n_slices);
    FB_HIP(hipMemcpyAsync(sc.data(), d_sc, sizeof(int2) * (size_t)n_slices, hipMemcpyDeviceToHost, s));
    FB_HIP(hipStreamSynchronize(s));
    long long padded = 0, used = 0;
    for (const int2& q : sc) { padded += 64LL * q.x; used += q.y; }
    if (force_sigma || sigma_wanted(n_nodes, padded, used)) {
      R.window = sigma_window(n_nodes);
      R.n_windows = (n_nodes + R.window - 1) / R.window;
      FB_TRY(R.d_win_keys.alloc((size_t)R.n_windows));
      hipLaunchKernelGGL(k_window_keys, dim3((unsigned)((R.n_windows + kB - 1) / kB)), dim3(kB), 0, s, R.n_windows, R.window, W.keys_s.p, R.d_win_keys.p);
      hipLaunchKernelGGL(k_sigma_keys, ng, dim3(kB), 0, s, n_nodes, R.window, W.vals_s.p, R.d_count.p, W.keys.p, W.vals.p);
      FB_HIP(hipGetLastError());
      unsigned bits2 = 10;
      while ((1LL << (bits2 - 10)) < R.n_windows) bits2++;
      bytes = 0;
      FB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, W.keys.p, W.keys_s.p, W.vals.p, W.vals_s.p, (size_t)n_nodes, 0u, bits2, s));
      FB_TRY(W.temp.reserve(std::max<size_t>(bytes, 16)));
      FB_HIP(rocprim::radix_sort_pairs(W.temp.p, bytes, W.keys.p, W.keys_s.p, W.vals.p, W.vals_s.p, (size_t)n_nodes, 0u, bits2, s));
      R.sigma = true;
    }
  }
  FB_TRY(R.d_old_of_new.alloc((size_t)n_nodes));
  FB_TRY(R.d_new_of_old.alloc((size_t)n_nodes));
  hipLaunchKernelGGL(k_invert, ng, dim3(kB), 0, s, n_nodes, W.vals_s.p, R.d_old_of_new.p, R.d_new_of_old.p);
  FB_HIP(hipGetLastError());
  FB_TRY(tet_span_device(s, n_tets, d_tets, n_nodes, R.d_new_of_old.p, W, &R.span_after, &R.mean_after));
  // AUTO keeps the new order only if the elements get a quarter narrower ON AVERAGE (the widest one may be an outlier -- a sliver on the
  // hull of a Delaunay mesh joins nodes a body apart in any order); ON keeps it
  if (mode != FB_RENUMBER_ON && R.mean_after * 4.0 > R.mean_before * 3.0) { R.span_after = R.span_before; R.mean_after = R.mean_before; return FB_OK; }
  R.geom = g;
  FB_TRY(R.d_keys.alloc((size_t)n_nodes));
  FB_HIP(hipMemcpyAsync(R.d_keys.p, W.keys_s.p, sizeof(unsigned long long) * (size_t)n_nodes, hipMemcpyDeviceToDevice, s));
  R.old_of_new.clear();  // (the host copies are fetched when an inspection entry point asks: Renumbering::host_maps)
  R.new_of_old.clear();
  R.n = n_nodes;
  R.active = true;
  return FB_OK;
}

int Renumbering::host_maps(hipStream_t s) {
  if (!active || (int)new_of_old.size() == n) return FB_OK;
  old_of_new.resize((size_t)n);
  new_of_old.resize((size_t)n);
  FB_TRY(d_old_of_new.download(old_of_new.data(), (size_t)n, s));
  return d_new_of_old.download(new_of_old.data(), (size_t)n, s);
}

namespace {
__global__ __launch_bounds__(kB) void k_fix_dofs(int n_fixed, const int* __restrict__ fixed, const int* __restrict__ new_of_old, unsigned char* __restrict__ dofmask) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n_fixed) return;
  const int d = fixed[i], node = d / 3;
  dofmask[3 * (size_t)(new_of_old ? new_of_old[node] : node) + (d - 3 * node)] = 0;
}
__global__ __launch_bounds__(kB) void k_node_masks(int n, const unsigned char* __restrict__ dofmask, unsigned char* __restrict__ nodemask) {
  const int l = blockIdx.x * kB + threadIdx.x;
  if (l >= n) return;
  nodemask[l] = (unsigned char)((dofmask[3 * (size_t)l] ? 1 : 0) | (dofmask[3 * (size_t)l + 1] ? 2 : 0) | (dofmask[3 * (size_t)l + 2] ? 4 : 0));
}
}  // namespace

int device_constraint_masks(hipStream_t s, int n_nodes, int n_fixed, const int* fixed_dofs, const int* d_new_of_old, DevBuf<int>& stage, DevBuf<unsigned char>& dofmask,
                            DevBuf<unsigned char>& nodemask) {
  const int r = 3 * n_nodes;
  for (int i = 0; i < n_fixed; i++) {
    if (fixed_dofs[i] < 0 || fixed_dofs[i] >= r) return fail(FB_EINVAL, "constrained DOF %d out of range [0,%d)", fixed_dofs[i], r);
    if (i && fixed_dofs[i] <= fixed_dofs[i - 1]) return fail(FB_EINVAL, "constrained DOFs must be strictly ascending (index %d)", i);
  }
  FB_TRY(dofmask.alloc((size_t)r));
  FB_TRY(nodemask.alloc((size_t)n_nodes));
  FB_HIP(hipMemsetAsync(dofmask.p, 1, (size_t)r, s));
  if (n_fixed > 0) {
    FB_TRY(stage.reserve((size_t)n_fixed));
    FB_HIP(hipMemcpyAsync(stage.p, fixed_dofs, sizeof(int) * (size_t)n_fixed, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_fix_dofs, dim3((unsigned)((n_fixed + kB - 1) / kB)), dim3(kB), 0, s, n_fixed, stage.p, d_new_of_old, dofmask.p);
    FB_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_node_masks, dim3((unsigned)((n_nodes + kB - 1) / kB)), dim3(kB), 0, s, n_nodes, dofmask.p, nodemask.p);
  FB_HIP(hipGetLastError());
  FB_HIP(hipStreamSynchronize(s));  // (the caller's list may go away)
  return FB_OK;
}

int relabel_tets(hipStream_t s, int n_tets, int4* d_tets, int n_nodes, const int* d_new_of_old) {
  hipLaunchKernelGGL(k_relabel, dim3((unsigned)((n_tets + kB - 1) / kB)), dim3(kB), 0, s, n_tets, d_tets, n_nodes, d_new_of_old);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int gather_nodes(hipStream_t s, int n, int width, const double* src, const int* map, double* dst) {
  const long long total = (long long)n * width;
  if (total <= 0) return FB_OK;
  hipLaunchKernelGGL(k_gather_nodes, dim3((unsigned)((total + kB - 1) / kB)), dim3(kB), 0, s, (long long)n, width, src, map, dst);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int scatter_nodes(hipStream_t s, int n, int width, const double* src, const int* map, double* dst) {
  const long long total = (long long)n * width;
  if (total <= 0) return FB_OK;
  hipLaunchKernelGGL(k_scatter_nodes, dim3((unsigned)((total + kB - 1) / kB)), dim3(kB), 0, s, (long long)n, width, src, map, dst);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int host_slab_order(int n_nodes, const double* xyz, int n_tets, const int* tets, std::vector<int>& old_of_new, int* span_before, int* span_after, double* mean_before,
                    double* mean_after) {
  old_of_new.resize((size_t)std::max(0, n_nodes));
  std::iota(old_of_new.begin(), old_of_new.end(), 0);
  double mean = 0.0;
  auto span_of = [&](const std::vector<int>* map) {
    int span = 0;
    unsigned long long sum = 0;
    for (int e = 0; e < n_tets; e++) {
      int lo = 0x7fffffff, hi = -1;
      bool ok = true;
      for (int i = 0; i < 4; i++) {
        const int v = tets[4 * (size_t)e + i];
        if (v < 0 || v >= n_nodes) { ok = false; break; }
        const int m = map ? (*map)[v] : v;
        lo = std::min(lo, m); hi = std::max(hi, m);
      }
      if (ok) { span = std::max(span, hi - lo); sum += (unsigned long long)(hi - lo); }
    }
    mean = n_tets > 0 ? (double)sum / (double)n_tets : 0.0;
    return span;
  };
  const int before = span_of(nullptr);
  if (span_before) *span_before = before;
  if (span_after) *span_after = before;
  if (mean_before) *mean_before = mean;
  if (mean_after) *mean_after = mean;
  if (n_nodes < 2) return FB_OK;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < n_nodes; i++)
    for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], xyz[3 * (size_t)i + k]); hi[k] = std::fmax(hi[k], xyz[3 * (size_t)i + k]); }
  SlabKeyGeom g;
  if (!slab_key_geometry(n_nodes, lo, hi, &g)) return FB_OK;
  std::vector<unsigned long long> key((size_t)n_nodes);
  for (int i = 0; i < n_nodes; i++) key[i] = slab_key(g, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
  std::stable_sort(old_of_new.begin(), old_of_new.end(), [&](int a, int b) { return key[a] < key[b]; });
  {  // second stage, as renumber_build decides and does it
    std::vector<int> count((size_t)n_nodes, 0);
    for (int e = 0; e < n_tets; e++) {
      bool ok = true;
      for (int i = 0; i < 4; i++) ok = ok && tets[4 * (size_t)e + i] >= 0 && tets[4 * (size_t)e + i] < n_nodes;
      if (ok) for (int i = 0; i < 4; i++) count[tets[4 * (size_t)e + i]]++;
    }
    long long padded = 0, used = 0;
    for (int s0 = 0; s0 < n_nodes; s0 += 64) {
      int m = 0;
      for (int l = s0; l < std::min(n_nodes, s0 + 64); l++) { m = std::max(m, count[old_of_new[l]]); used += count[old_of_new[l]]; }
      padded += 64LL * m;
    }
    if (sigma_wanted(n_nodes, padded, used)) {
      std::vector<unsigned long long> key2((size_t)n_nodes);
      const int window = sigma_window(n_nodes);
      for (int l = 0; l < n_nodes; l++) key2[old_of_new[l]] = sigma_key(l, window, count[old_of_new[l]]);
      std::stable_sort(old_of_new.begin(), old_of_new.end(), [&](int a, int b) { return key2[a] < key2[b]; });
    }
  }
  if (span_after || mean_after) {
    std::vector<int> new_of_old((size_t)n_nodes);
    for (int l = 0; l < n_nodes; l++) new_of_old[old_of_new[l]] = l;
    const int after = span_of(&new_of_old);
    if (span_after) *span_after = after;
    if (mean_after) *mean_after = mean;
  }
  return FB_OK;
}

}  // namespace fb
