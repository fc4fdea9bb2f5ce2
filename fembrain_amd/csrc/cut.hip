// Scalpel / tet-mesh intersection tests of the cutting tool: the four kernels of data/opencl/Cutting.cl (:158-341) and the
// device half of PS::FEM::Cutting (src/deformable/Cutting.cpp:87-497), hand-written for gfx950.
//
// All of it is streaming fp32 work over the tet list: a tet costs 16 B of indices, four gathered 16-B vertices (served by
// L2, every vertex is shared by ~24 tets) and 4 x 20 B (faces) or 6 x 20 B (edges) of output, so the kernels are bound by
// the output stream.  The reference runs one work-item per tet that walks its 4 faces / 6 edges and stores 16-B points at
// a stride of 64 / 96 B; here a lane owns ONE face or edge, so a wavefront's flag and point stores are contiguous
// (256 B and 1 KiB), the four / six lanes of a tet share the tet's index quad through one 16-B load, and the number of
// hits is counted in the same pass (wave ballot, one atomic per wave) instead of reading the flag array back to the host
// for a scan (Cutting.cpp:237-248).  Arithmetic follows the reference's fp32 ground truth IntersectSegmentTriangleF /
// IntersectRayTriangleF (src/graphics/Intersections.cpp:12-64) operation for operation, compiled without contraction,
// so results are bit-identical to it (the one difference: the determinant test uses the kernel's 1E-5f, Cutting.cl:13,
// where the C++ compares against the double 1E-5).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace fb {
namespace {

constexpr int kCB = 256;
constexpr float kEps = 1e-5f;  // Cutting.cl:13

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 xyz(float4 v) { return {v.x, v.y, v.z}; }

// IntersectRayTriangleF (Intersections.cpp:12-40; Cutting.cl:59-102)
__device__ __forceinline__ bool ray_triangle(V3 ro, V3 rd, V3 p0, V3 p1, V3 p2, float* t_out) {
  const V3 e1 = sub(p1, p0), e2 = sub(p2, p0);
  const V3 q = cross(rd, e2);
  const float a = dot(e1, q);
  if (fabsf(a) < kEps) return false;
  const float f = 1.0f / a;
  const V3 s = sub(ro, p0);
  const float u = f * dot(s, q);
  if (u < 0.0f) return false;
  const V3 r = cross(s, e1);
  const float v = f * dot(rd, r);
  if (v < 0.0f || u + v > 1.0f) return false;
  *t_out = f * dot(e2, r);
  return true;
}

// a segment prepared once: direction = delta * (1 / |delta|) (Vec3::normalize, base/Vec.h:371-379), length
struct Seg { V3 s0, rd; float len; };
__device__ __forceinline__ Seg make_seg(V3 s0, V3 s1) {
  Seg g;
  g.s0 = s0;
  V3 d = sub(s1, s0);
  g.len = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
  if (g.len != 0.0f) {
    const float inv = 1.0f / g.len;
    d.x *= inv; d.y *= inv; d.z *= inv;
  }
  g.rd = d;
  return g;
}

// IntersectSegmentTriangleF (Intersections.cpp:42-64; Cutting.cl:122-147)
__device__ __forceinline__ bool segment_triangle(const Seg& g, V3 p0, V3 p1, V3 p2, V3* xp) {
  float t;
  if (!ray_triangle(g.s0, g.rd, p0, p1, p2, &t)) return false;
  if (!(t >= 0.0f && t <= g.len)) return false;
  *xp = {g.s0.x + g.rd.x * t, g.s0.y + g.rd.y * t, g.s0.z + g.rd.z * t};
  return true;
}

__device__ __forceinline__ void count_hits(bool hit, unsigned int* counter) {
  const unsigned long long m = __ballot(hit);
  if (m != 0ULL && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(counter, (unsigned)__popcll(m));
}

__device__ __forceinline__ int tet_node(const uint4& t, int k) { return (int)(k == 0 ? t.x : k == 1 ? t.y : k == 2 ? t.z : t.w); }

// faceMask of Cutting.cl:174-176 / :223-225: {0,1,2} {1,2,3} {2,3,0} {0,1,3}, packed 2 bits per corner
__device__ __forceinline__ void face_nodes(const uint4& t, int f, int* a, int* b, int* c) {
  const unsigned packed = 0x24u | (0x39u << 6) | (0x0Eu << 12) | (0x34u << 18);  // (c<<4 | b<<2 | a) per face
  const unsigned m = packed >> (6 * f);
  *a = tet_node(t, m & 3); *b = tet_node(t, (m >> 2) & 3); *c = tet_node(t, (m >> 4) & 3);
}

// MODE 0: ComputePerTetCentroids (Cutting.cl:158-196): flag 1, centroid.
// MODE 1: ComputePerTetFaceIntersections (:207-256): flag = scalpel edge crosses the face, point = crossing, else centroid.
template <int MODE>
__global__ __launch_bounds__(kCB) void k_cut_faces(long long n_faces, const float4* __restrict__ verts, const uint4* __restrict__ tets, float4 s0,
                                                   float4 s1, unsigned int* __restrict__ flags, float4* __restrict__ points,
                                                   unsigned int* __restrict__ counter) {
  const long long i = (long long)blockIdx.x * kCB + threadIdx.x;
  bool hit = false;
  if (i < n_faces) {
    const uint4 t = tets[i >> 2];
    int a, b, c;
    face_nodes(t, (int)(i & 3), &a, &b, &c);
    const float4 p0 = verts[a], p1 = verts[b], p2 = verts[c];
    const float third = 1.0f / 3.0f;
    float4 out = make_float4(third * ((p0.x + p1.x) + p2.x), third * ((p0.y + p1.y) + p2.y), third * ((p0.z + p1.z) + p2.z), 1.0f);
    if (MODE == 0) {
      hit = true;
    } else {
      const Seg g = make_seg(xyz(s0), xyz(s1));
      V3 xp;
      hit = segment_triangle(g, xyz(p0), xyz(p1), xyz(p2), &xp);
      if (hit) { out.x = xp.x; out.y = xp.y; out.z = xp.z; }
    }
    flags[i] = hit ? 1u : 0u;
    __builtin_nontemporal_store(out.x, &points[i].x);
    __builtin_nontemporal_store(out.y, &points[i].y);
    __builtin_nontemporal_store(out.z, &points[i].z);
    __builtin_nontemporal_store(out.w, &points[i].w);
  }
  count_hits(hit, counter);
}

// ComputePerTetEdgeIntersections (Cutting.cl:262-316): edge {0,1} {1,2} {2,0} {0,3} {1,3} {2,3} against the two triangles
// (q0, q3, q1) and (q0, q2, q3) of the swept quad, first hit wins.  The reference leaves the point of a missed edge
// unwritten (stale memory); here it is (0, 0, 0, 1).
__global__ __launch_bounds__(kCB) void k_cut_edges(long long n_edges, const float4* __restrict__ verts, const uint4* __restrict__ tets, float4 q0, float4 q1,
                                                   float4 q2, float4 q3, unsigned int* __restrict__ flags, float4* __restrict__ points,
                                                   unsigned int* __restrict__ counter) {
  const long long i = (long long)blockIdx.x * kCB + threadIdx.x;
  bool hit = false;
  if (i < n_edges) {
    const long long tet = i / 6;
    const int e = (int)(i - tet * 6);
    const uint4 t = tets[tet];
    const unsigned lo = 0x24u | (0x0u << 6) | (0x1u << 8) | (0x2u << 10), hi = (1u) | (2u << 2) | (0u << 4) | (3u << 6) | (3u << 8) | (3u << 10);
    const int a = tet_node(t, (lo >> (2 * e)) & 3), b = tet_node(t, (hi >> (2 * e)) & 3);
    const Seg g = make_seg(xyz(verts[a]), xyz(verts[b]));
    V3 xp = {0.0f, 0.0f, 0.0f};
    hit = segment_triangle(g, xyz(q0), xyz(q3), xyz(q1), &xp);
    if (!hit) hit = segment_triangle(g, xyz(q0), xyz(q2), xyz(q3), &xp);
    if (!hit) xp = {0.0f, 0.0f, 0.0f};
    flags[i] = hit ? 1u : 0u;
    __builtin_nontemporal_store(xp.x, &points[i].x);
    __builtin_nontemporal_store(xp.y, &points[i].y);
    __builtin_nontemporal_store(xp.z, &points[i].z);
    __builtin_nontemporal_store(1.0f, &points[i].w);
  }
  count_hits(hit, counter);
}

// ComputeSegmentTriIntersections (Cutting.cl:321-341): loose triangles (3 float4 each) against one segment; (-1,-1,-1,1) = miss
__global__ __launch_bounds__(kCB) void k_cut_segment_tris(int n_tris, const float4* __restrict__ verts, float4 s0, float4 s1, float4* __restrict__ points) {
  const int i = blockIdx.x * kCB + threadIdx.x;
  if (i >= n_tris) return;
  const Seg g = make_seg(xyz(s0), xyz(s1));
  V3 xp;
  float4 out = make_float4(-1.0f, -1.0f, -1.0f, 1.0f);
  if (segment_triangle(g, xyz(verts[3 * i]), xyz(verts[3 * i + 1]), xyz(verts[3 * i + 2]), &xp)) { out.x = xp.x; out.y = xp.y; out.z = xp.z; }
  points[i] = out;
}

// ---- ordered compaction of the hit list: per-block counts, one-block scan of the counts, scatter ---------------------------
__global__ __launch_bounds__(kCB) void k_cut_block_counts(long long n, const unsigned int* __restrict__ flags, unsigned int* __restrict__ block_count) {
  __shared__ unsigned int waves[kCB / 64];
  const long long i = (long long)blockIdx.x * kCB + threadIdx.x;
  const unsigned long long m = __ballot(i < n && flags[i] != 0u);
  if ((threadIdx.x & 63) == 0) waves[threadIdx.x >> 6] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = waves[0] + waves[1] + waves[2] + waves[3];
}

__global__ __launch_bounds__(1024) void k_cut_scan_blocks(int n_blocks, unsigned int* __restrict__ block_count) {
  __shared__ unsigned int part[1024];
  const int per = (n_blocks + 1023) / 1024, lo = threadIdx.x * per, hi = min(n_blocks, lo + per);
  unsigned int s = 0;
  for (int i = lo; i < hi; i++) s += block_count[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned int v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned int run = part[threadIdx.x] - s;  // exclusive
  for (int i = lo; i < hi; i++) { const unsigned int c = block_count[i]; block_count[i] = run; run += c; }
}

__global__ __launch_bounds__(kCB) void k_cut_scatter(long long n, const unsigned int* __restrict__ flags, const float4* __restrict__ points,
                                                     const unsigned int* __restrict__ block_base, unsigned int* __restrict__ ids, float4* __restrict__ out) {
  __shared__ unsigned int waves[kCB / 64];
  const long long i = (long long)blockIdx.x * kCB + threadIdx.x;
  const bool hit = i < n && flags[i] != 0u;
  const unsigned long long m = __ballot(hit);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) waves[w] = (unsigned)__popcll(m);
  __syncthreads();
  if (!hit) return;
  unsigned int pos = block_base[blockIdx.x] + (unsigned)__popcll(m & ((1ULL << lane) - 1ULL));
  for (int k = 0; k < w; k++) pos += waves[k];
  ids[pos] = (unsigned int)i;
  out[pos] = points[i];
}

}  // namespace
}  // namespace fb

using namespace fb;

struct fb_cut_s {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int n_vertices = 0, n_tets = 0;
  DevBuf<float4> verts;
  DevBuf<uint4> tets;
  DevBuf<unsigned int> face_flags, edge_flags, counter, block_count, hit_ids;
  DevBuf<float4> face_points, edge_points, hit_points;
  unsigned int n_face_hits = 0, n_edge_hits = 0;
  bool faces_done = false, edges_done = false;
};

#define CHECK_CUT(h)                                    \
  if (!(h)) return fail(FB_EINVAL, "null cut handle");  \
  FB_HIP(hipSetDevice((h)->device))

static int upload_vertices(fb_cut_s* h, const double* xyz) {
  // Cutting::createMemBuffers (Cutting.cpp:137-147): float4 (x, y, z, 1) of the current node positions
  std::vector<float4> v((size_t)h->n_vertices);
  for (int i = 0; i < h->n_vertices; i++) v[i] = make_float4((float)xyz[3 * (size_t)i], (float)xyz[3 * (size_t)i + 1], (float)xyz[3 * (size_t)i + 2], 1.0f);
  for (int i = 0; i < h->n_vertices; i++)
    if (!std::isfinite(v[i].x) || !std::isfinite(v[i].y) || !std::isfinite(v[i].z)) return fail(FB_EINVAL, "vertex %d is not finite", i);
  return h->verts.upload(v, h->stream);
}

extern "C" {

int fb_cut_create(fb_cut_t* out, int device, int n_vertices, const double* xyz, int n_tets, const unsigned int* tets) {
  if (!out || n_vertices < 1 || !xyz || n_tets < 1 || !tets) return fail(FB_EINVAL, "bad tet mesh arrays");
  if ((long long)n_tets * 6 > 0xffffffffLL) return fail(FB_EINVAL, "too many tets");
  for (long long i = 0; i < 4LL * n_tets; i++)
    if (tets[i] >= (unsigned)n_vertices) return fail(FB_EINVAL, "tet %lld references vertex %u of %d", i / 4, tets[i], n_vertices);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FB_EDEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(FB_EINVAL, "device %d out of range", device);
  FB_HIP(hipSetDevice(device));
  fb_cut_s* h = new fb_cut_s;
  h->device = device; h->n_vertices = n_vertices; h->n_tets = n_tets;
  int rc = FB_OK;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(FB_EDEVICE, "hipStreamCreate failed");
  for (auto& e : h->ev)
    if (rc == FB_OK && hipEventCreate(&e) != hipSuccess) rc = fail(FB_EDEVICE, "hipEventCreate failed");
  if (rc == FB_OK) rc = upload_vertices(h, xyz);
  if (rc == FB_OK) rc = h->tets.upload((const uint4*)tets, (size_t)n_tets, h->stream);
  if (rc == FB_OK) rc = h->face_flags.alloc(4 * (size_t)n_tets);
  if (rc == FB_OK) rc = h->face_points.alloc(4 * (size_t)n_tets);
  if (rc == FB_OK) rc = h->edge_flags.alloc(6 * (size_t)n_tets);
  if (rc == FB_OK) rc = h->edge_points.alloc(6 * (size_t)n_tets);
  if (rc == FB_OK) rc = h->counter.alloc(2);
  if (rc != FB_OK) { fb_cut_destroy(h); return rc; }
  *out = h;
  return FB_OK;
}

void fb_cut_destroy(fb_cut_t h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) { (void)hipStreamSynchronize(h->stream); }
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  h->verts.release(); h->tets.release(); h->face_flags.release(); h->edge_flags.release(); h->counter.release();
  h->block_count.release(); h->hit_ids.release(); h->face_points.release(); h->edge_points.release(); h->hit_points.release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int fb_cut_set_vertices(fb_cut_t h, int n_vertices, const double* xyz) {
  CHECK_CUT(h);
  if (n_vertices != h->n_vertices || !xyz) return fail(FB_EINVAL, "expected %d vertices", h->n_vertices);
  h->faces_done = h->edges_done = false;
  return upload_vertices(h, xyz);
}

static int read_count(fb_cut_s* h, int which, unsigned int* out) {
  FB_HIP(hipMemcpyAsync(out, h->counter.p + which, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  return FB_OK;
}

static float4 f4(const double* p) { return make_float4((float)p[0], (float)p[1], (float)p[2], 1.0f); }

int fb_cut_face_centroids(fb_cut_t h) {
  CHECK_CUT(h);
  const long long n = 4LL * h->n_tets;
  FB_HIP(hipMemsetAsync(h->counter.p, 0, sizeof(unsigned int), h->stream));
  hipLaunchKernelGGL(k_cut_faces<0>, dim3((unsigned)((n + kCB - 1) / kCB)), dim3(kCB), 0, h->stream, n, h->verts.p, h->tets.p, make_float4(0, 0, 0, 1),
                     make_float4(0, 0, 0, 1), h->face_flags.p, h->face_points.p, h->counter.p);
  FB_HIP(hipGetLastError());
  FB_TRY(read_count(h, 0, &h->n_face_hits));
  h->faces_done = true;
  return FB_OK;
}

int fb_cut_face_intersections(fb_cut_t h, const double* s0, const double* s1, int* n_hits) {
  CHECK_CUT(h);
  if (!s0 || !s1) return fail(FB_EINVAL, "null scalpel edge");
  const long long n = 4LL * h->n_tets;
  FB_HIP(hipMemsetAsync(h->counter.p, 0, sizeof(unsigned int), h->stream));
  hipLaunchKernelGGL(k_cut_faces<1>, dim3((unsigned)((n + kCB - 1) / kCB)), dim3(kCB), 0, h->stream, n, h->verts.p, h->tets.p, f4(s0), f4(s1),
                     h->face_flags.p, h->face_points.p, h->counter.p);
  FB_HIP(hipGetLastError());
  FB_TRY(read_count(h, 0, &h->n_face_hits));
  h->faces_done = true;
  if (n_hits) *n_hits = (int)h->n_face_hits;
  return FB_OK;
}

int fb_cut_edge_intersections(fb_cut_t h, const double* quad12, int* n_hits) {
  CHECK_CUT(h);
  if (!quad12) return fail(FB_EINVAL, "null swept quad");
  const long long n = 6LL * h->n_tets;
  FB_HIP(hipMemsetAsync(h->counter.p + 1, 0, sizeof(unsigned int), h->stream));
  hipLaunchKernelGGL(k_cut_edges, dim3((unsigned)((n + kCB - 1) / kCB)), dim3(kCB), 0, h->stream, n, h->verts.p, h->tets.p, f4(quad12), f4(quad12 + 3),
                     f4(quad12 + 6), f4(quad12 + 9), h->edge_flags.p, h->edge_points.p, h->counter.p + 1);
  FB_HIP(hipGetLastError());
  FB_TRY(read_count(h, 1, &h->n_edge_hits));
  h->edges_done = true;
  if (n_hits) *n_hits = (int)h->n_edge_hits;
  return FB_OK;
}

int fb_cut_read(fb_cut_t h, int what, unsigned int* flags, float* points_xyzw) {
  CHECK_CUT(h);
  if (what != FB_CUT_FACES && what != FB_CUT_EDGES) return fail(FB_EINVAL, "what must be FB_CUT_FACES or FB_CUT_EDGES");
  const bool faces = what == FB_CUT_FACES;
  if (!(faces ? h->faces_done : h->edges_done)) return fail(FB_EINVAL, "run the intersection pass first");
  const size_t n = (size_t)(faces ? 4 : 6) * h->n_tets;
  if (flags) FB_TRY((faces ? h->face_flags : h->edge_flags).download(flags, n, h->stream));
  if (points_xyzw) FB_TRY((faces ? h->face_points : h->edge_points).download((float4*)points_xyzw, n, h->stream));
  return FB_OK;
}

int fb_cut_read_hits(fb_cut_t h, int what, int capacity, unsigned int* ids, float* points_xyzw, int* n_out) {
  CHECK_CUT(h);
  if (what != FB_CUT_FACES && what != FB_CUT_EDGES) return fail(FB_EINVAL, "what must be FB_CUT_FACES or FB_CUT_EDGES");
  const bool faces = what == FB_CUT_FACES;
  if (!(faces ? h->faces_done : h->edges_done)) return fail(FB_EINVAL, "run the intersection pass first");
  const unsigned int hits = faces ? h->n_face_hits : h->n_edge_hits;
  if (n_out) *n_out = (int)hits;
  if (hits == 0 || (!ids && !points_xyzw)) return FB_OK;
  if (capacity < (int)hits) return fail(FB_EINVAL, "%u hits do not fit a capacity of %d", hits, capacity);
  const long long n = (long long)(faces ? 4 : 6) * h->n_tets;
  const int blocks = (int)((n + kCB - 1) / kCB);
  const unsigned int* fl = (faces ? h->face_flags : h->edge_flags).p;
  FB_TRY(h->block_count.alloc((size_t)blocks));
  FB_TRY(h->hit_ids.alloc((size_t)hits));
  FB_TRY(h->hit_points.alloc((size_t)hits));
  hipLaunchKernelGGL(k_cut_block_counts, dim3(blocks), dim3(kCB), 0, h->stream, n, fl, h->block_count.p);
  hipLaunchKernelGGL(k_cut_scan_blocks, dim3(1), dim3(1024), 0, h->stream, blocks, h->block_count.p);
  hipLaunchKernelGGL(k_cut_scatter, dim3(blocks), dim3(kCB), 0, h->stream, n, fl, (faces ? h->face_points : h->edge_points).p, h->block_count.p, h->hit_ids.p,
                     h->hit_points.p);
  FB_HIP(hipGetLastError());
  if (ids) FB_TRY(h->hit_ids.download(ids, hits, h->stream));
  if (points_xyzw) FB_TRY(h->hit_points.download((float4*)points_xyzw, hits, h->stream));
  return FB_OK;
}

int fb_cut_segment_triangles(int device, int n_tris, const float* tri_xyzw, const float* s0, const float* s1, float* points_xyzw) {
  if (n_tris < 0 || (n_tris > 0 && (!tri_xyzw || !points_xyzw)) || !s0 || !s1) return fail(FB_EINVAL, "bad triangle array");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FB_EDEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(FB_EINVAL, "device %d out of range", device);
  FB_HIP(hipSetDevice(device));
  if (n_tris == 0) return FB_OK;
  DevBuf<float4> v, out;
  FB_TRY(v.upload((const float4*)tri_xyzw, 3 * (size_t)n_tris, nullptr));
  FB_TRY(out.alloc((size_t)n_tris));
  hipLaunchKernelGGL(k_cut_segment_tris, dim3((n_tris + kCB - 1) / kCB), dim3(kCB), 0, nullptr, n_tris, v.p, make_float4(s0[0], s0[1], s0[2], 1.0f),
                     make_float4(s1[0], s1[1], s1[2], 1.0f), out.p);
  FB_HIP(hipGetLastError());
  return out.download((float4*)points_xyzw, (size_t)n_tris, nullptr);
}

int fb_cut_time(fb_cut_t h, int what, const double* a, const double* b_or_quad_rest, int reps, double* ms_per_pass) {
  CHECK_CUT(h);
  if (reps < 1 || !ms_per_pass || !a) return fail(FB_EINVAL, "bad timing arguments");
  const bool faces = what == FB_CUT_FACES;
  if (faces && !b_or_quad_rest) return fail(FB_EINVAL, "null scalpel edge");
  const long long n = (long long)(faces ? 4 : 6) * h->n_tets;
  const dim3 grid((unsigned)((n + kCB - 1) / kCB));
  auto launch = [&]() {
    if (faces)
      hipLaunchKernelGGL(k_cut_faces<1>, grid, dim3(kCB), 0, h->stream, n, h->verts.p, h->tets.p, f4(a), f4(b_or_quad_rest), h->face_flags.p, h->face_points.p,
                         h->counter.p);
    else
      hipLaunchKernelGGL(k_cut_edges, grid, dim3(kCB), 0, h->stream, n, h->verts.p, h->tets.p, f4(a), f4(a + 3), f4(a + 6), f4(a + 9), h->edge_flags.p,
                         h->edge_points.p, h->counter.p + 1);
  };
  launch();
  FB_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; r++) launch();
  FB_HIP(hipEventRecord(h->ev[1], h->stream));
  FB_HIP(hipEventSynchronize(h->ev[1]));
  FB_HIP(hipGetLastError());
  float ms = 0.0f;
  FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *ms_per_pass = ms / reps;
  // leave the counters and flags as one clean pass would
  if (faces) return fb_cut_face_intersections(h, a, b_or_quad_rest, nullptr);
  return fb_cut_edge_intersections(h, a, nullptr);
}

}  // extern "C"
