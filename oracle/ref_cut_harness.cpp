// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product path.
//
// C-ABI harness around the reference's own src/graphics/Intersections.cpp (which needs nothing but base/Vec.h and
// base/MathBase.h of the reference tree), compiled where it lies by oracle/Makefile into oracle/_ref/libcut_ref.so.
// This file is ours; it only CALLS PS::INTERSECTIONS::IntersectSegmentTriangleF / IntersectSegmentTriangle.
// src/deformable/Cutting_CPU.cpp (the TBB loop around them) is NOT built: TBB is absent from this image.
#include "graphics/Intersections.h"

extern "C" {

// n triangles (9 floats each) against n segments (6 floats each): hit flag, point (3 floats), t
void ref_segment_triangle_f(int n, const float* seg, const float* tri, int* hit, float* xyz, float* t) {
  for (int i = 0; i < n; i++) {
    PS::MATH::vec3f s0(seg[6 * i], seg[6 * i + 1], seg[6 * i + 2]), s1(seg[6 * i + 3], seg[6 * i + 4], seg[6 * i + 5]);
    PS::MATH::vec3f p[3];
    for (int c = 0; c < 3; c++) p[c] = PS::MATH::vec3f(tri[9 * i + 3 * c], tri[9 * i + 3 * c + 1], tri[9 * i + 3 * c + 2]);
    PS::MATH::vec3f uvw, x;
    float tt = 0.0f;
    hit[i] = PS::INTERSECTIONS::IntersectSegmentTriangleF(s0, s1, p, tt, uvw, x);
    if (hit[i]) { xyz[3 * i] = x.x; xyz[3 * i + 1] = x.y; xyz[3 * i + 2] = x.z; t[i] = tt; }
  }
}

void ref_segment_triangle_d(int n, const double* seg, const double* tri, int* hit, double* xyz, double* t) {
  for (int i = 0; i < n; i++) {
    PS::MATH::vec3d s0(seg[6 * i], seg[6 * i + 1], seg[6 * i + 2]), s1(seg[6 * i + 3], seg[6 * i + 4], seg[6 * i + 5]);
    PS::MATH::vec3d p[3];
    for (int c = 0; c < 3; c++) p[c] = PS::MATH::vec3d(tri[9 * i + 3 * c], tri[9 * i + 3 * c + 1], tri[9 * i + 3 * c + 2]);
    PS::MATH::vec3d uvw, x;
    double tt = 0.0;
    hit[i] = PS::INTERSECTIONS::IntersectSegmentTriangle(s0, s1, p, tt, uvw, x);
    if (hit[i]) { xyz[3 * i] = x.x; xyz[3 * i + 1] = x.y; xyz[3 * i + 2] = x.z; t[i] = tt; }
  }
}

}  // extern "C"
