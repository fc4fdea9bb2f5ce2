/* TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product path.
 *
 * CPU restatement of the cutting tool's intersection tests: the four kernels of data/opencl/Cutting.cl (:158-341) with the
 * fp32 arithmetic of the reference's own ground truth IntersectRayTriangleF / IntersectSegmentTriangleF
 * (src/graphics/Intersections.cpp:12-64, Vec3 helpers base/Vec.h:371-396, :432-485), written out in plain C floats and
 * compiled without contraction.  PINNED by (a) the reference's known answer in Cutting::computeFaceSegmentIntersectionTest
 * (Cutting.cpp:351-369: triangle (-1,0,-1) (1,0,-1) (0,0,1), segment (0,-1,0)-(0,1,0) -> (0,0,0)) and (b) the
 * reference's Intersections.cpp itself, compiled where it lies into oracle/_ref/libcut_ref.so and compared triangle by
 * triangle (tests/test_oracle_cut.py).  One deliberate difference from the C++: the determinant test compares against
 * the kernel's float 1E-5f (Cutting.cl:13), the C++ against the double 1E-5 -- they differ only for |a| == 1E-5f exactly. */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
static v3 v_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static float v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                       /* Vec.h:472-475 */
static v3 v_cross(v3 a, v3 b) { v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; } /* :478-485 */

/* Intersections.cpp:12-40 */
static int ray_triangle(v3 ro, v3 rd, const v3 p[3], float uvt[3]) {
  v3 e1 = v_sub(p[1], p[0]), e2 = v_sub(p[2], p[0]);
  v3 q = v_cross(rd, e2);
  float a = v_dot(e1, q);
  if (fabsf(a) < 1e-5f) return 0;
  float f = 1.0f / a;
  v3 s = v_sub(ro, p[0]);
  uvt[0] = f * v_dot(s, q);
  if (uvt[0] < 0.0f) return 0;
  v3 r = v_cross(s, e1);
  uvt[1] = f * v_dot(rd, r);
  if (uvt[1] < 0.0f || uvt[0] + uvt[1] > 1.0f) return 0;
  uvt[2] = f * v_dot(e2, r);
  return 1;
}

/* Intersections.cpp:42-64; normalize = multiply by 1/len (Vec.h:371-379) */
int orc_segment_triangle(const float s0[3], const float s1[3], const float tri[9], float xyz[3], float *t_out) {
  v3 a = {s0[0], s0[1], s0[2]}, b = {s1[0], s1[1], s1[2]};
  v3 p[3] = {{tri[0], tri[1], tri[2]}, {tri[3], tri[4], tri[5]}, {tri[6], tri[7], tri[8]}};
  v3 d = v_sub(b, a);
  float len = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
  if (len != 0.0f) {
    float inv = 1.0f / len;
    d.x *= inv; d.y *= inv; d.z *= inv;
  }
  float uvt[3] = {0.0f, 0.0f, 0.0f};
  if (!ray_triangle(a, d, p, uvt)) return 0;
  if (!(uvt[2] >= 0.0f && uvt[2] <= len)) return 0;
  xyz[0] = a.x + d.x * uvt[2]; xyz[1] = a.y + d.y * uvt[2]; xyz[2] = a.z + d.z * uvt[2];
  if (t_out) *t_out = uvt[2];
  return 1;
}

static const int kFace[4][3] = {{0, 1, 2}, {1, 2, 3}, {2, 3, 0}, {0, 1, 3}};                 /* Cutting.cl:174-176 */
static const int kEdge[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}};             /* Cutting.cl:290-292 */

static void load(const double *xyz, uint32_t node, float out[3]) {
  for (int k = 0; k < 3; k++) out[k] = (float)xyz[3 * (size_t)node + k];                    /* Cutting.cpp:139-142 */
}

/* mode 0: ComputePerTetCentroids (:158-196); mode 1: ComputePerTetFaceIntersections (:207-256).  Returns the hit count. */
long long orc_cut_faces(int mode, const double *xyz, long long n_tets, const uint32_t *tets, const double *s0d, const double *s1d, uint32_t *flags,
                        float *points) {
  float s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
  if (mode == 1) for (int k = 0; k < 3; k++) { s0[k] = (float)s0d[k]; s1[k] = (float)s1d[k]; }
  const float third = 1.0f / 3.0f;
  long long hits = 0;
  for (long long t = 0; t < n_tets; t++)
    for (int f = 0; f < 4; f++) {
      float tri[9], xp[3];
      for (int c = 0; c < 3; c++) load(xyz, tets[4 * t + kFace[f][c]], tri + 3 * c);
      float *o = points + 4 * (4 * t + f);
      for (int k = 0; k < 3; k++) o[k] = third * ((tri[k] + tri[3 + k]) + tri[6 + k]);
      o[3] = 1.0f;
      int hit = 1;
      if (mode == 1) {
        hit = orc_segment_triangle(s0, s1, tri, xp, 0);
        if (hit) memcpy(o, xp, sizeof xp);
      }
      flags[4 * t + f] = (uint32_t)hit;
      hits += hit;
    }
  return hits;
}

/* ComputePerTetEdgeIntersections (:262-316); the point of a missed edge is defined as (0, 0, 0, 1) */
long long orc_cut_edges(const double *xyz, long long n_tets, const uint32_t *tets, const double *quad12, uint32_t *flags, float *points) {
  float q[4][3], tri0[9], tri1[9];
  for (int i = 0; i < 4; i++) for (int k = 0; k < 3; k++) q[i][k] = (float)quad12[3 * i + k];
  memcpy(tri0, q[0], 12); memcpy(tri0 + 3, q[3], 12); memcpy(tri0 + 6, q[1], 12);
  memcpy(tri1, q[0], 12); memcpy(tri1 + 3, q[2], 12); memcpy(tri1 + 6, q[3], 12);
  long long hits = 0;
  for (long long t = 0; t < n_tets; t++)
    for (int e = 0; e < 6; e++) {
      float a[3], b[3], xp[3] = {0, 0, 0};
      load(xyz, tets[4 * t + kEdge[e][0]], a);
      load(xyz, tets[4 * t + kEdge[e][1]], b);
      int hit = orc_segment_triangle(a, b, tri0, xp, 0);
      if (!hit) hit = orc_segment_triangle(a, b, tri1, xp, 0);
      float *o = points + 4 * (6 * t + e);
      o[0] = hit ? xp[0] : 0.0f; o[1] = hit ? xp[1] : 0.0f; o[2] = hit ? xp[2] : 0.0f; o[3] = 1.0f;
      flags[6 * t + e] = (uint32_t)hit;
      hits += hit;
    }
  return hits;
}

/* ComputeSegmentTriIntersections (:321-341): triangles as 3 x float4 */
void orc_cut_segment_tris(int n_tris, const float *tri_xyzw, const float *s0, const float *s1, float *points) {
  for (int i = 0; i < n_tris; i++) {
    float tri[9], xp[3];
    for (int c = 0; c < 3; c++) memcpy(tri + 3 * c, tri_xyzw + 4 * (3 * (size_t)i + c), 12);
    float *o = points + 4 * (size_t)i;
    o[0] = o[1] = o[2] = -1.0f; o[3] = 1.0f;
    if (orc_segment_triangle(s0, s1, tri, xp, 0)) memcpy(o, xp, 12);
  }
}
