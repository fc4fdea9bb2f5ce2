// BlobTree field sweep, grid classification and the tetrahedral polygonizer on gfx950 -- the GPU side of
// PS::SKETCH::GPUPoly (reference src/implicit/OclPolygonizer.{h,cpp}) and PS::SKETCH::FieldComputer
// (src/implicit/FieldComputer.{h,cpp}); kernels replace data/opencl/Polygonizer.cl and Tetrahedralizer.cl.
//
// Design (MI355X): every pass is a flat HBM-bound sweep with one thread per grid point / cell and no host
// round trip (the reference does D2H + host scan + H2D between passes, OclPolygonizer.cpp:663-730):
//   sweep      float4 (x,y,z,f) per point, 16 B/lane coalesced stores + one inside bit per point (wave ballot)
//   classify   cell configs (u8) and edge flags (u8) from the 2 MB inside bitmask only; included-cell and
//              included-vertex bitmasks by ballot; per-64 popcounts
//   scan       exclusive scan of the per-word popcounts (n/64 entries) -- any thread then gets the rank of any
//              point/cell as base[word] + popc(mask[word] & lower bits): no full-size offset arrays
//   emit       tet-mesh vertices (grid order) and 6 tets per included cell (Tetrahedralizer.cl:67-132 pattern)
//
// Field semantics follow the reference CPU path FieldComputer::fieldValue / computePrimitiveField
// (src/implicit/Polygonizer.cpp:1544-2108) evaluated in scalar fp32 with its exact operation order:
// range operators SUM their primitives, and the running `outField` is carried from one operator to the next
// exactly as that code does.  Not reproduced (documented in DESIGN.md): its all-SIMD-lanes-outside bounding-box
// cull (depends on the host SIMD width), the rsqrt+Newton approximation (1/sqrt here), the rational pow of the
// Ricci blend (powf here).
//
// Field semantics are a property of the handle (fb_poly_set_field_semantics), compiled into the kernels as a template
// parameter so that the default path is untouched:
//   SEM_CPU (default)  the CPU path as described above, without computePrimitiveField's primitive box cull
//   SEM_CPU_BOX        the same WITH that cull (isOutsidePrim, Polygonizer.cpp:1485-1505,1548-1552), tested per point
//   SEM_OPENCL         the OpenCL kernel's evaluation (data/opencl/Polygonizer.cl:483-886) -- the path that wrote the .veg
//                      files the reference ships: binary operators evaluated by ComputeOpField(operator INDEX, ..) (:825),
//                      range operators by their own type, instanced nodes 0, over the `next` links of
//                      LinearBlobTree::setTraversalRoute (LinearBlobTree.cpp:333-429)
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "common.h"

using namespace fb;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kPB = 256;  // threads per block
constexpr float kIso = 0.5f;
constexpr int kSweepPts = 4;  // 256-point runs per sweep block

enum PrimType { primPoint, primLine, primCylinder, primDisc, primRing, primCube, primTriangle, primQuadricPoint, primNULL, primInstance, primRBF };
enum OpType { opUnion, opIntersect, opDif, opSmoothDif, opBlend, opRicciBlend, opGradientBlend, opFastQuadricPointSet, opCache,
              opWarpTwist, opWarpTaper, opWarpBend, opWarpShear };
enum OpFlags { ofRightChildIsOp = 1, ofLeftChildIsOp = 2, ofChildIndexIsRange = 4, ofIsUnaryOp = 8, ofIsRightOp = 16, ofBreak = 32 };
enum { SEM_CPU = FB_FIELD_CPU, SEM_CPU_BOX = FB_FIELD_CPU_BOX, SEM_OPENCL = FB_FIELD_OPENCL };
constexpr int kNullBlob = 0xFFFF;         // NULL_BLOB, LinearBlobTree.h:18
constexpr int kZeroOperand = -0x7fffffff;  // OpenCL program: an operand register that still holds its initial 0.0f

// one step of the compiled evaluation order.  Values live in numbered slots of a per-thread LDS column; the running
// field `out` and the query point are registers.
struct Instr {
  int kind;    // 0 RANGE: out += prims a..b.  1 OP.  2 ENTER an instanced subtree.  3 LEAVE it.  4 ADDSLOT: out += slot a
               // OpenCL program (SEM_OPENCL) only: 5 CL_RANGE: out = fold of prims a..b by optype.  6 CL_OP: out = op(a, b)
  int a, b;    // OP: left / right operand, >= 0 primitive index, < 0 slot -1-k.  ENTER: a = matrix node, b = first of the
               // 5 frame slots (x, y, z, out, inside).  LEAVE: a = frame slots
  int optype;  // OP only
  int unary;
  int dst;     // slot that receives the result (RANGE, OP, LEAVE); -1 = only `out`
  int skip;    // ENTER: index of the matching LEAVE (taken when the mapped point is outside the box)
  int ca, cb;  // colour pass: OP: primitive whose colour a primitive operand carries (an instance of a primitive shows the
               // original's colour).  ADDSLOT: ca = primitive whose colour weighs the slot, cb = 1 if the slot REPLACES the colour
               // ENTER: ca = the instance primitive itself (its own box is tested under SEM_CPU_BOX)
  float p0, p1;
  float lo[3], hi[3];  // ENTER: box of the original operator (isOutsideOp, Polygonizer.cpp:1464-1483)
};

struct Grid {
  float lo[3];
  float cellsize;
  int g[3];  // points per axis
  int c[3];  // cells per axis
  long long n_points, n_cells;
  int z0;    // slab of a larger grid: global index of this grid's first point plane (positions are lo + cellsize * global index)
};

__device__ __forceinline__ float wyvill(float dd) {
  const float t = 1.0f - dd;
  return fmaxf(0.0f, (t * t) * t);  // Polygonizer.h:519-529
}

__device__ __forceinline__ bool in_box(const float* __restrict__ b, float x, float y, float z) {  // isOutsidePrim / isOutsideOp: inclusive bounds
  return x >= b[0] && b[3] >= x && y >= b[1] && b[4] >= y && z >= b[2] && b[5] >= z;
}

// computePrimitiveField (Polygonizer.cpp:1544-1908), scalar fp32.  SEM_CPU: no bounding-box cull.  SEM_CPU_BOX: 0 outside
// the primitive's box (cbox: lo, hi per primitive), tested with the incoming point at every level of an instance chain.
// SEM_OPENCL: ComputePrimitiveField of the kernel (Polygonizer.cl:483-695) -- instanced nodes are 0, a quadric point falls
// through to the Wyvill value outside its radius.
template <int SEM>
__device__ __forceinline__ float prim_field_t(const float* __restrict__ prims, const float* __restrict__ mtx, const float* __restrict__ cbox, int i,
                                              float pX, float pY, float pZ) {
  const float* P = prims + 20 * i;
  int type = (int)P[0];
  float x = pX, y = pY, z = pZ;
  if (SEM == SEM_OPENCL && type == primInstance) return 0.0f;  // Polygonizer.cl:505-531
  for (int hop = 0;; hop++) {
    if (SEM == SEM_CPU_BOX && !in_box(cbox + 6 * i, x, y, z)) return 0.0f;
    const int im = (int)P[1];
    if (im != 0) {
      const float* m = mtx + 12 * im;
      const float tx = m[0] * x + m[1] * y + m[2] * z + m[3];
      const float ty = m[4] * x + m[5] * y + m[6] * z + m[7];
      const float tz = m[8] * x + m[9] * y + m[10] * z + m[11];
      x = tx; y = ty; z = tz;
    }
    // an instance of a PRIMITIVE is that primitive at the mapped point (Polygonizer.cpp:1879-1901); instances of
    // operators never get here: compile_tree expands them into ENTER .. LEAVE blocks
    if (__builtin_expect(type != primInstance, 1) || hop == 8 || (int)P[14] != 0) break;
    i = (int)P[12];
    P = prims + 20 * i;
    type = (int)P[0];
  }
  const float posX = P[4], posY = P[5], posZ = P[6];
  const float dirX = P[8], dirY = P[9], dirZ = P[10];
  const float resX = P[12], resY = P[13], resZ = P[14];
  float dist2 = 0.0f;
  switch (type) {
    case primPoint: {
      const float dx = posX - x, dy = posY - y, dz = posZ - z;
      dist2 = (dx * dx) + (dy * dy) + (dz * dz);
    } break;
    case primLine: {
      const float lx = dirX - posX, ly = dirY - posY, lz = dirZ - posZ;
      const float ldot = lx * lx + ly * ly + lz * lz;
      float dx = x - posX, dy = y - posY, dz = z - posZ;
      float delta = dx * lx + dy * ly + dz * lz;
      delta = delta / ldot;
      dx = x - (posX + delta * lx);
      dy = y - (posY + delta * ly);
      dz = z - (posZ + delta * lz);
      dist2 = (dx * dx) + (dy * dy) + (dz * dz);
    } break;
    case primCylinder: {
      const float px = x - posX, py = y - posY, pz = z - posZ;
      float yy = px * dirX + py * dirY + pz * dirZ;
      const float xx = fmaxf(0.0f, sqrtf(px * px + py * py + pz * pz - yy * yy) - resX);
      const float mask = yy > 0.0f ? 1.0f : 0.0f;
      yy = mask * fmaxf(0.0f, yy - resY) + (1.0f - mask) * yy;
      dist2 = xx * xx + yy * yy;
    } break;
    case primTriangle:
      dist2 = FLT_MAX;
      break;
    case primCube: {
      const float side = resX, minusSide = -1.0f * resX;
      const float dx = x - posX, dy = y - posY, dz = z - posZ;
      float mm = minusSide > dx ? 1.0f : 0.0f, mp = dx > side ? 1.0f : 0.0f;
      float delta = (dx + side) * mm + (dx - side) * mp;
      dist2 = delta * delta;
      mm = minusSide > dy ? 1.0f : 0.0f; mp = dy > side ? 1.0f : 0.0f;
      delta = (dy + side) * mm + (dy - side) * mp;
      dist2 += delta * delta;
      mm = minusSide > dz ? 1.0f : 0.0f; mp = dz > side ? 1.0f : 0.0f;
      delta = (dz + side) * mm + (dz - side) * mp;
      dist2 += delta * delta;
    } break;
    case primDisc: {
      const float dX = x - posX, dY = y - posY, dZ = z - posZ;
      float dot = dirX * dX + dirY * dY + dirZ * dZ;
      float ex = dX - dirX * dot, ey = dY - dirY * dot, ez = dZ - dirZ * dot;
      dot = ex * ex + ey * ey + ez * ez;
      const float rs = 1.0f / sqrtf(dot);
      ex = ex * rs; ey = ey * rs; ez = ez * rs;
      const float nx = resX * ex - dX, ny = resX * ey - dY, nz = resX * ez - dZ;
      if (resX * resX >= dot) dist2 = fabsf(dX * dX + dY * dY + dZ * dZ - dot);
      else dist2 = nx * nx + ny * ny + nz * nz;
    } break;
    case primRing: {
      const float dX = x - posX, dY = y - posY, dZ = z - posZ;
      float dot = dirX * dX + dirY * dY + dirZ * dZ;
      float ex = dX - dirX * dot, ey = dY - dirY * dot, ez = dZ - dirZ * dot;
      dot = ex * ex + ey * ey + ez * ez;
      if (dot == 0.0f) {
        dist2 = resX * resX + dX * dX + dY * dY + dZ * dZ;
      } else {
        const float rs = 1.0f / sqrtf(dot);
        ex = ex * rs; ey = ey * rs; ez = ez * rs;
        const float nx = resX * ex - dX, ny = resX * ey - dY, nz = resX * ez - dZ;
        dist2 = nx * nx + ny * ny + nz * nz;
      }
    } break;
    case primQuadricPoint: {
      const float dX = x - posX, dY = y - posY, dZ = z - posZ;
      dist2 = dX * dX + dY * dY + dZ * dZ;
      if (SEM == SEM_OPENCL) {  // Polygonizer.cl:672-683
        if (dirZ > dist2) return dist2 * dist2 * resX + dist2 * resY + resZ;
        break;
      }
      return dirZ > dist2 ? (dist2 * dist2 * resX + dist2 * resY + resZ) : 0.0f;
    }
    case primNULL:
      dist2 = 10.0f;
      break;
    default:  // primInstance and unknown types contribute nothing
      return 0.0f;
  }
  return wyvill(dist2);
}

__device__ __forceinline__ float prim_field(const float* __restrict__ prims, const float* __restrict__ mtx, int i, float pX, float pY, float pZ) {
  return prim_field_t<SEM_CPU>(prims, mtx, nullptr, i, pX, pY, pZ);
}

__device__ __forceinline__ float apply_op(int optype, float lf, float rf, float p0, float p1, float keep) {
  switch (optype) {
    case opBlend: return lf + rf;
    case opRicciBlend: return powf(powf(lf, p0) + powf(rf, p0), p1);
    case opUnion: return fmaxf(lf, rf);
    case opIntersect: return fminf(lf, rf);
    case opDif: return fminf(lf, 1.0f - rf);
    case opSmoothDif: return lf * (1.0f - rf);
    case opWarpBend: case opWarpTwist: case opWarpTaper: case opWarpShear: return lf;
    default: return keep;  // the reference switch leaves outField untouched for the remaining types
  }
}

// Wave-level culling of primitives in the sweep.  The 64 points of a wavefront are a run of consecutive grid points; its
// bounding box is known from the run's first and last index alone (no cross-lane work).  A primitive whose SUPPORT box
// (where its field can be non-zero at all: skeleton grown by the Wyvill radius 1, mapped to world space, padded against
// rounding -- support_box() on the host) misses that box contributes exactly 0.0f to every lane, so skipping it leaves
// every field value bit-identical.  (This is not the reference's bounding-box cull, whose boxes are tighter than the
// support and whose outcome depends on the host SIMD width -- DESIGN.md section 2.)  Decisions are wave-uniform.
struct SegBox {
  float lo[3], hi[3];
  const float* pbox;  // 6 floats per primitive: support lo, hi
  const float* cbox;  // SEM_CPU_BOX: the reference's primitive boxes (PrepareAllBoxes), 6 floats per primitive
};

template <bool CULL, int SEM>
__device__ __forceinline__ float prim_val(const float* __restrict__ prims, const float* __restrict__ mtx, int i, float x, float y, float z,
                                          const SegBox& sb, int nest) {
  if (CULL && nest == 0) {  // inside an instanced subtree the point has been mapped: no culling there
    const float* b = sb.pbox + 6 * i;
    if (sb.lo[0] > b[3] || sb.hi[0] < b[0] || sb.lo[1] > b[4] || sb.hi[1] < b[1] || sb.lo[2] > b[5] || sb.hi[2] < b[2]) return 0.0f;
  }
  return prim_field_t<SEM>(prims, mtx, sb.cbox, i, x, y, z);
}

// ---- SEM_OPENCL: the kernel's operator functions ----
__device__ __forceinline__ float cl_op_field(int optype, float lf, float rf, float p0, float p1) {  // ComputeOpField, Polygonizer.cl:697-729
  switch (optype) {
    case opUnion: return fmaxf(lf, rf);
    case opIntersect: return fminf(lf, rf);
    case opBlend: return lf + rf;
    case opRicciBlend: return powf(powf(lf, p0) + powf(rf, p0), p1);
    case opDif: return fminf(lf, 1.0f - rf);
    case opSmoothDif: return lf * (1.0f - rf);
    default: return 0.0f;
  }
}

// ComputeField (Polygonizer.cl:848-886) over the program compile_tree_cl wrote: the control flow of the stackless walk does
// not depend on the field values, so it is run once on the host and leaves a straight list of CL_RANGE / CL_OP steps whose
// operands name a primitive, a value slot or the constant 0 a register still holds.
template <bool CULL>
__device__ __forceinline__ float eval_field_cl(const Instr* __restrict__ prog, int n_instr, const float* __restrict__ prims,
                                               const float* __restrict__ mtx, float x, float y, float z, float* stk, const SegBox& sb) {
  if (n_instr == 0) return prim_val<CULL, SEM_OPENCL>(prims, mtx, 0, x, y, z, sb, 0);  // no operators: primitive 0 alone (:884)
  float out = 0.0f;
  for (int k = 0; k < n_instr; k++) {
    const Instr& in = prog[k];
    float f = 0.0f;
    if (in.kind == 5) {  // ComputeRangeField :731-770
      switch (in.optype) {
        case opUnion: for (int i = in.a; i <= in.b; i++) f = fmaxf(f, prim_val<CULL, SEM_OPENCL>(prims, mtx, i, x, y, z, sb, 0)); break;
        case opIntersect: for (int i = in.a; i <= in.b; i++) f = fminf(f, prim_val<CULL, SEM_OPENCL>(prims, mtx, i, x, y, z, sb, 0)); break;
        case opBlend: for (int i = in.a; i <= in.b; i++) f += prim_val<CULL, SEM_OPENCL>(prims, mtx, i, x, y, z, sb, 0); break;
        case opRicciBlend:
          for (int i = in.a; i <= in.b; i++) f = powf(powf(f, in.p0) + powf(prim_val<CULL, SEM_OPENCL>(prims, mtx, i, x, y, z, sb, 0), in.p0), in.p1);
          break;
        default: break;
      }
    } else {
      const int a = in.a, b = in.b;
      const float lf = a >= 0 ? prim_val<CULL, SEM_OPENCL>(prims, mtx, a, x, y, z, sb, 0) : (a == kZeroOperand ? 0.0f : stk[(-1 - a) * kPB]);
      const float rf = b >= 0 ? prim_val<CULL, SEM_OPENCL>(prims, mtx, b, x, y, z, sb, 0) : (b == kZeroOperand ? 0.0f : stk[(-1 - b) * kPB]);
      f = cl_op_field(in.optype, lf, rf, in.p0, in.p1);
    }
    out = f;
    stk[in.dst * kPB] = f;
  }
  return out;
}

// FieldComputer::fieldValue (Polygonizer.cpp:1913-2108) through the compiled order.  `stk` is this thread's
// column of an LDS slot array [depth][kPB].
template <bool CULL, int SEM>
__device__ __forceinline__ float eval_field_t(const Instr* __restrict__ prog, int n_instr, int n_prims, const float* __restrict__ prims,
                                              const float* __restrict__ mtx, float x, float y, float z, float* stk, const SegBox& sb) {
  if (SEM == SEM_OPENCL) return eval_field_cl<CULL>(prog, n_instr, prims, mtx, x, y, z, stk, sb);
  float out = 0.0f;
  int nest = 0;
  if (n_instr == 0) {  // no operators: blend of all primitives (:2085-2096)
    for (int i = 0; i < n_prims; i++) out = out + prim_val<CULL, SEM>(prims, mtx, i, x, y, z, sb, nest);
    return out;
  }
  for (int k = 0; k < n_instr; k++) {
    const Instr& in = prog[k];
    switch (in.kind) {
      case 0:
        for (int i = in.a; i <= in.b; i++) out = out + prim_val<CULL, SEM>(prims, mtx, i, x, y, z, sb, nest);
        break;
      case 1: {
        const int a = in.a, b = in.b;
        const float lf = a >= 0 ? prim_val<CULL, SEM>(prims, mtx, a, x, y, z, sb, nest) : stk[(-1 - a) * kPB];
        float rf = 0.0f;
        if (!in.unary) rf = b >= 0 ? prim_val<CULL, SEM>(prims, mtx, b, x, y, z, sb, nest) : stk[(-1 - b) * kPB];
        out = apply_op(in.optype, lf, rf, in.p0, in.p1, out);
      } break;
      case 2: {  // computePrimitiveField of an operator instance: map the point, cull against the original's box
        float* f = stk + in.b * kPB;
        f[0] = x; f[kPB] = y; f[2 * kPB] = z; f[3 * kPB] = out;
        const bool own = SEM != SEM_CPU_BOX || in_box(sb.cbox + 6 * in.ca, x, y, z);  // the instance node's own box, incoming point
        if (in.a != 0) {
          const float* m = mtx + 12 * in.a;
          const float tx = m[0] * x + m[1] * y + m[2] * z + m[3];
          const float ty = m[4] * x + m[5] * y + m[6] * z + m[7];
          const float tz = m[8] * x + m[9] * y + m[10] * z + m[11];
          x = tx; y = ty; z = tz;
        }
        out = 0.0f;
        nest++;
        const bool inside = own && x >= in.lo[0] && in.hi[0] >= x && y >= in.lo[1] && in.hi[1] >= y && z >= in.lo[2] && in.hi[2] >= z;
        f[4 * kPB] = inside ? 1.0f : 0.0f;
        // the program counter stays wave-uniform (scalar instruction fetch): the block is skipped only when every lane
        // is outside; a lane that is outside while a neighbour is inside runs along and gets its 0 at the LEAVE
        if (__ballot(inside) == 0ULL) k = in.skip - 1;
      } continue;
      case 3: {
        const float* f = stk + in.a * kPB;
        const float v = f[4 * kPB] != 0.0f ? out : 0.0f;
        x = f[0]; y = f[kPB]; z = f[2 * kPB]; out = f[3 * kPB];
        stk[in.dst * kPB] = v;
        nest--;
      } continue;
      default:
        out = out + stk[in.a * kPB];
        continue;
    }
    if (in.dst >= 0) stk[in.dst * kPB] = out;
  }
  return out;
}

template <int SEM>
__device__ __forceinline__ float eval_field(const Instr* __restrict__ prog, int n_instr, int n_prims, const float* __restrict__ prims,
                                            const float* __restrict__ mtx, const float* __restrict__ cbox, float x, float y, float z, float* stk) {
  SegBox none;
  none.pbox = nullptr;
  none.cbox = cbox;
  return eval_field_t<false, SEM>(prog, n_instr, n_prims, prims, mtx, x, y, z, stk, none);
}

// Division of a grid index (< 2^31) by a grid extent without the ~25-instruction division sequence: n / d = mulhi(n, m) >> s with
// m = floor(2^(31 + c) / d) + 1, s = c - 1, c = ceil(log2 d) (exact for n < 2^31: the error term n e / 2^(31 + c) is below 1 / d).
// k_tet_vertices spent two thirds of its time in the two divisions per grid point (35 -> measured below at 256^3).
struct FastDiv {
  unsigned int mul, shift, one;  // one: d == 1
};
inline FastDiv fast_div(unsigned int d) {
  FastDiv f = {0u, 0u, d <= 1u ? 1u : 0u};
  if (d <= 1u) return f;
  unsigned int c = 0;
  while ((1ull << c) < d) c++;
  f.mul = (unsigned int)(((1ull << (31 + c)) / d) + 1ull);
  f.shift = c - 1;
  return f;
}
__device__ __forceinline__ unsigned int div_by(unsigned int n, const FastDiv f) { return f.one ? n : (__umulhi(n, f.mul) >> f.shift); }

// ---- ComputeAllFields (Polygonizer.cl:1215-1236): v = lo + cellsize*(ix,iy,iz), index iz*gx*gy + iy*gx + ix ----
template <bool CULL, int SEM>
__global__ __launch_bounds__(kPB) void k_sweep(Grid G, FastDiv dxy, FastDiv dx, const Instr* __restrict__ prog, int n_instr, int n_prims, int depth,
                                               const float* __restrict__ prims, const float* __restrict__ mtx, const float* __restrict__ pbox,
                                               const float* __restrict__ cbox, float* __restrict__ fout, unsigned long long* __restrict__ inside) {
  extern __shared__ float stack[];
  // kSweepPts consecutive 256-point runs per block: every store instruction of a wave is still one contiguous 1 KiB
  // float4 segment and every ballot one aligned 64-point word of the inside mask
  const unsigned int gx = (unsigned int)G.g[0], gxy = gx * (unsigned int)G.g[1];  // n_points < 2^31
#pragma unroll
  for (int k = 0; k < kSweepPts; k++) {
    const long long gid = ((long long)blockIdx.x * kSweepPts + k) * kPB + threadIdx.x;
    bool in = false;
    // every lane's grid index (lanes past the end of the grid compute one too: harmless, they evaluate nothing)
    const unsigned int g32 = (unsigned int)gid;
    const unsigned int iz = div_by(g32, dxy), rem = g32 - iz * gxy;
    const unsigned int iy = div_by(rem, dx), ix = rem - iy * gx;
    // bounding box of this wavefront's 64-point run from the indices of its first and last valid lane (scalar reads of
    // values the lanes hold anyway)
    SegBox sb;
    sb.pbox = pbox;
    sb.cbox = cbox;
    if (CULL) {
      const long long first = gid - (threadIdx.x & 63);
      const int last_lane = __builtin_amdgcn_readfirstlane((int)min(63LL, G.n_points - 1 - first));
      const int ll = last_lane < 0 ? 0 : last_lane;
      const unsigned int ax = __builtin_amdgcn_readlane(ix, 0), ay = __builtin_amdgcn_readlane(iy, 0), az = __builtin_amdgcn_readlane(iz, 0);
      const unsigned int bx = __builtin_amdgcn_readlane(ix, ll), by = __builtin_amdgcn_readlane(iy, ll), bz = __builtin_amdgcn_readlane(iz, ll);
      const bool same_plane = az == bz, same_row = same_plane && ay == by;
      const unsigned int x0 = same_row ? ax : 0u, x1 = same_row ? bx : gx - 1u;
      const unsigned int y0 = same_plane ? ay : 0u, y1 = same_plane ? by : (unsigned int)G.g[1] - 1u;
      sb.lo[0] = G.lo[0] + G.cellsize * (float)x0; sb.hi[0] = G.lo[0] + G.cellsize * (float)x1;
      sb.lo[1] = G.lo[1] + G.cellsize * (float)y0; sb.hi[1] = G.lo[1] + G.cellsize * (float)y1;
      sb.lo[2] = G.lo[2] + G.cellsize * (float)(az + (unsigned int)G.z0); sb.hi[2] = G.lo[2] + G.cellsize * (float)(bz + (unsigned int)G.z0);
    }
    if (gid < G.n_points) {
      const float x = G.lo[0] + G.cellsize * (float)ix;
      const float y = G.lo[1] + G.cellsize * (float)iy;
      const float z = G.lo[2] + G.cellsize * (float)(iz + (unsigned int)G.z0);
      const float f = eval_field_t<CULL, SEM>(prog, n_instr, n_prims, prims, mtx, x, y, z, stack + threadIdx.x, sb);
      // f alone (round 5): 4 bytes per point where rounds 1-4 wrote the float4 (x, y, z, f) the API returns -- 268 MB at 256^3 that
      // no stage of the tet path reads (classification reads the inside mask, emission computes positions from indices) and the
      // surface path needs only for f at the two ends of a crossed edge.  fb_poly_read_grid materialises the float4 grid on demand.
      if (fout) __builtin_nontemporal_store(f, &fout[gid]);
      in = f >= kIso;  // inside test of Polygonizer.cl:1367,1599 and Polygonizer.cpp:1052
    }
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && gid < ((G.n_points + 63) & ~63LL)) inside[gid >> 6] = b;
  }
  (void)depth;
}

// the float4 (x, y, z, f) grid of the API (readBackVoxelGridSamples, OclPolygonizer.cpp:1651-1694) from the stored field values:
// positions by the sweep's own expression, so the result is what the sweep used to store, bit for bit
__global__ __launch_bounds__(kPB) void k_grid_xyzf(Grid G, FastDiv dxy, FastDiv dx, const float* __restrict__ fval, float4* __restrict__ grid) {
  const long long gid = (long long)blockIdx.x * kPB + threadIdx.x;
  if (gid >= G.n_points) return;
  const unsigned int gx = (unsigned int)G.g[0], gxy = gx * (unsigned int)G.g[1];
  const unsigned int g32 = (unsigned int)gid;
  const unsigned int iz = div_by(g32, dxy), rem = g32 - iz * gxy;
  const unsigned int iy = div_by(rem, dx), ix = rem - iy * gx;
  const v4f o = {G.lo[0] + G.cellsize * (float)ix, G.lo[1] + G.cellsize * (float)iy, G.lo[2] + G.cellsize * (float)(iz + (unsigned int)G.z0), fval[gid]};
  __builtin_nontemporal_store(o, (v4f*)&grid[gid]);
}

// ComputeFieldArray (Polygonizer.cl:1262-1286)
template <int SEM>
__global__ __launch_bounds__(kPB) void k_field_array(int n, const Instr* __restrict__ prog, int n_instr, int n_prims,
                                                     const float* __restrict__ prims, const float* __restrict__ mtx, const float* __restrict__ cbox,
                                                     float4* __restrict__ pts) {
  extern __shared__ float stack[];
  const int i = blockIdx.x * kPB + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  p.w = eval_field<SEM>(prog, n_instr, n_prims, prims, mtx, cbox, p.x, p.y, p.z, stack + threadIdx.x);
  pts[i] = p;
}

__device__ inline int bit_at(const unsigned long long* __restrict__ m, long long i) { return (int)((m[i >> 6] >> (i & 63)) & 1ULL); }

// ---- classification on the inside bitmask, 64 grid points per thread -----------------------------------------
// Point p = iz*gx*gy + iy*gx + ix is bit p of `inside`.  A cell is addressed by its lower-corner point (cells keep the
// reference's linear order: the valid lower corners in increasing p), so every per-cell quantity is a bit array over
// points as well and the reference's per-item passes become shifts of the bit array by 1, gx and gx*gy:
//   cell config != 0   (Tetrahedralizer.cl:3-35, included)  = OR  of inside over the 8 corners   -> cinc
//   cell config == 255 (Polygonizer.cl:1564-1607)           = AND of inside over the 8 corners   -> surface = cinc & ~all
//   edge flags X/Y/Z   (Polygonizer.cl:1353-1415)           = inside ^ inside(+1 / +gx / +gx*gy), masked at the last x/y/z
//   included vertices  (TetMeshCells' scatter-marks)        = OR of cinc over the 8 cells around the point -> vinc
// word w of A shifted so that bit p holds A[p + k] (fwd) or A[p - k] (bwd); bits past either end read 0
__device__ inline unsigned long long fwd_word(const unsigned long long* __restrict__ a, long long nwords, long long w, long long k) {
  const long long q = w + (k >> 6);
  const int r = (int)(k & 63);
  unsigned long long lo = q < nwords ? a[q] : 0ULL;
  if (r == 0) return lo;
  const unsigned long long hi = (q + 1) < nwords ? a[q + 1] : 0ULL;
  return (lo >> r) | (hi << (64 - r));
}
__device__ inline unsigned long long bwd_word(const unsigned long long* __restrict__ a, long long nwords, long long w, long long k) {
  const long long q = w - (k >> 6);
  const int r = (int)(k & 63);
  unsigned long long hi = (q >= 0 && q < nwords) ? a[q] : 0ULL;
  if (r == 0) return hi;
  const unsigned long long lo = (q - 1 >= 0) ? a[q - 1] : 0ULL;
  return (hi << r) | (lo >> (64 - r));
}

// one-time per grid: bits of the points with ix == gx-1 / iy == gy-1 / iz == gz-1 and of the points inside the grid
__global__ __launch_bounds__(kPB) void k_grid_masks(Grid G, unsigned long long* __restrict__ lastx, unsigned long long* __restrict__ lasty,
                                                    unsigned long long* __restrict__ lastz, unsigned long long* __restrict__ valid,
                                                    unsigned long long* __restrict__ firstx, unsigned long long* __restrict__ firsty,
                                                    unsigned long long* __restrict__ firstz) {
  const long long gid = (long long)blockIdx.x * kPB + threadIdx.x;
  bool lx = false, ly = false, lz = false, in = false, fx = false, fy = false, fz = false;
  if (gid < G.n_points) {
    const int gxy = G.g[0] * G.g[1];
    const int z = (int)(gid / gxy);
    const int rem = (int)(gid - (long long)z * gxy);
    const int y = rem / G.g[0], x = rem - y * G.g[0];
    lx = x == G.g[0] - 1; ly = y == G.g[1] - 1; lz = z == G.g[2] - 1; in = true;
    fx = x == 0; fy = y == 0; fz = z == 0;
  }
  const unsigned long long bx = __ballot(lx), by = __ballot(ly), bz = __ballot(lz), bv = __ballot(in);
  const unsigned long long ax = __ballot(fx), ay = __ballot(fy), az = __ballot(fz);
  if ((threadIdx.x & 63) == 0 && gid < ((G.n_points + 63) & ~63LL)) {
    lastx[gid >> 6] = bx; lasty[gid >> 6] = by; lastz[gid >> 6] = bz; valid[gid >> 6] = bv;
    firstx[gid >> 6] = ax; firsty[gid >> 6] = ay; firstz[gid >> 6] = az;
  }
}

// on-demand materialisation of the reference's per-item outputs for read-back: edge flags (X=4,Y=2,Z=1) per point and
// the 8-bit configuration per cell (corner c = 4*dx + 2*dy + dz, cell index cz*cx*cy + cy*cx + cx)
__global__ __launch_bounds__(kPB) void k_materialize(Grid G, const unsigned long long* __restrict__ inside,
                                                     const unsigned long long* __restrict__ crossx, const unsigned long long* __restrict__ crossy,
                                                     const unsigned long long* __restrict__ crossz, unsigned char* __restrict__ flags,
                                                     unsigned char* __restrict__ config) {
  const long long gid = (long long)blockIdx.x * kPB + threadIdx.x;
  if (gid >= G.n_points) return;
  const int gxy = G.g[0] * G.g[1];
  const int z = (int)(gid / gxy);
  const int rem = (int)(gid - (long long)z * gxy);
  const int y = rem / G.g[0], x = rem - y * G.g[0];
  flags[gid] = (unsigned char)((bit_at(crossx, gid) << 2) | (bit_at(crossy, gid) << 1) | bit_at(crossz, gid));
  if (x < G.c[0] && y < G.c[1] && z < G.c[2]) {
    const long long gx = G.g[0], gxyl = (long long)gxy, p = gid;
    const int cfg = bit_at(inside, p) | (bit_at(inside, p + gxyl) << 1) | (bit_at(inside, p + gx) << 2) | (bit_at(inside, p + gx + gxyl) << 3) |
                    (bit_at(inside, p + 1) << 4) | (bit_at(inside, p + 1 + gxyl) << 5) | (bit_at(inside, p + 1 + gx) << 6) |
                    (bit_at(inside, p + 1 + gx + gxyl) << 7);
    config[(long long)z * G.c[0] * G.c[1] + (long long)y * G.c[0] + x] = (unsigned char)cfg;
  }
}

// Exclusive scan of the per-word popcounts in three small passes (n = points/64): per-chunk sums (kChunk words per
// block), one block scanning the chunk sums, per-chunk local scans.  `aux` (same length) is only summed.
constexpr int kChunk = 4096;  // words per block: 16 per thread

__device__ inline unsigned int block_excl_scan256(unsigned int v, unsigned int* total, unsigned int* sh) {
  // exclusive scan of one value per thread over a 256-thread block; *total = block sum
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  __syncthreads();
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  unsigned int woff = 0;
  for (int k = 0; k < w; k++) woff += sh[k];
  *total = sh[0] + sh[1] + sh[2] + sh[3];
  return woff + incl - v;
}

// ---- classification of a swept grid in TWO launches (round 4; rounds 1-3: k_classify_bits, k_vertex_bits and three scan launches.  A
// single launch with the scans inside -- every workgroup posting tagged sums and adding up its predecessors' -- measured 33 us at
// 256^3 against 8 + 4 here: a thousand workgroups spinning with uncached loads on the same 16 KB) ----
// 66 bits of the bit array `a` from bit position B on (B may be negative or past the end: zeros there): lo = bits [B, B + 64), hi = the two after
__device__ inline void window66(const unsigned long long* __restrict__ a, long long nwords, long long B, unsigned long long& lo, unsigned int& hi) {
  const long long q = B >> 6;  // (floor: arithmetic shift)
  const int r = (int)(B & 63);
  const unsigned long long w0 = (q >= 0 && q < nwords) ? a[q] : 0ULL, w1 = (q + 1 >= 0 && q + 1 < nwords) ? a[q + 1] : 0ULL;
  if (r == 0) { lo = w0; hi = (unsigned int)(w1 & 3ULL); return; }
  lo = (w0 >> r) | (w1 << (64 - r));
  unsigned long long h = w1 >> r;
  if (r == 63) h |= ((q + 2 >= 0 && q + 2 < nwords) ? a[q + 2] : 0ULL) << 1;
  hi = (unsigned int)(h & 3ULL);
}

// Per 64-point word w (bit i = grid point 64 w + i; a cell is addressed by its lower corner):
//   cinc   config != 0: OR of `inside` over the cell's 8 corners, at valid lower corners      (Tetrahedralizer.cl:3-35)
//   crossX/Y/Z  inside ^ inside(+1 / +gx / +gx gy), not on the last x / y / z                 (Polygonizer.cl:1353-1415)
//   vinc   the point is a corner of an included cell = OR of `inside` over the points p + {-1,0,1}^3 that share a valid cell with p:
//          per axis the offset +1 counts unless p is on the last plane of that axis, -1 unless on the first -- the dilation is separable,
//          so no pass over cinc is needed (rounds 1-3 made vinc from cinc in a second kernel)   (TetMeshCells' marks, Tetrahedralizer.cl:39-64)
// from nine 66-bit windows of `inside` (rows y-1, y, y+1 of planes z-1, z, z+1); per workgroup the four sums the ranks and totals are made of.
// ROWS64 (round 4): the grid's x extent is a multiple of 64, so a word lies inside one grid row and the seven position masks follow
// from its row and plane -- computed instead of loaded (14.7 MB of the kernel's 28 at 256^3); dxy / dx: divisions by the plane and row sizes in words.
template <bool ROWS64>
__global__ __launch_bounds__(kPB) void k_classify(Grid G, FastDiv dxy, FastDiv dx, long long nwords, const unsigned long long* __restrict__ inside,
                                                  const unsigned long long* __restrict__ lastx, const unsigned long long* __restrict__ lasty,
                                                  const unsigned long long* __restrict__ lastz, const unsigned long long* __restrict__ firstx,
                                                  const unsigned long long* __restrict__ firsty, const unsigned long long* __restrict__ firstz,
                                                  const unsigned long long* __restrict__ valid, unsigned long long* __restrict__ cinc,
                                                  unsigned long long* __restrict__ crossx, unsigned long long* __restrict__ crossy,
                                                  unsigned long long* __restrict__ crossz, unsigned long long* __restrict__ vinc,
                                                  unsigned int* __restrict__ aux_pop, uint4* __restrict__ block_sums) {
  __shared__ unsigned int sh[4];
  const long long gx = G.g[0], gxy = (long long)G.g[0] * G.g[1];
  const long long w = (long long)blockIdx.x * kPB + threadIdx.x;
  unsigned int pc = 0, pv = 0, ne = 0, ns = 0;
  if (w < nwords) {
    unsigned long long lx, ly, lz, fx, fy, fz, vd;
    if (ROWS64) {
      const unsigned int wpr = (unsigned int)(G.g[0] >> 6), wpp = wpr * (unsigned int)G.g[1];  // words per row / per plane
      const unsigned int w32 = (unsigned int)w, z = div_by(w32, dxy), rem = w32 - z * wpp, y = div_by(rem, dx), xw = rem - y * wpr;
      lx = xw == wpr - 1u ? 1ULL << 63 : 0ULL;
      fx = xw == 0u ? 1ULL : 0ULL;
      ly = y == (unsigned int)G.g[1] - 1u ? ~0ULL : 0ULL;
      fy = y == 0u ? ~0ULL : 0ULL;
      lz = z == (unsigned int)G.g[2] - 1u ? ~0ULL : 0ULL;
      fz = z == 0u ? ~0ULL : 0ULL;
      vd = ~0ULL;  // (every word is full: n_points is a multiple of 64)
    } else {
      lx = lastx[w]; ly = lasty[w]; lz = lastz[w]; fx = firstx[w]; fy = firsty[w]; fz = firstz[w]; vd = valid[w];
    }
    unsigned long long any = 0ULL, all = ~0ULL, dil = 0ULL, c0 = 0ULL, cx = 0ULL, cy = 0ULL, cz = 0ULL;
#pragma unroll
    for (int oz = -1; oz <= 1; oz++)
#pragma unroll
      for (int oy = -1; oy <= 1; oy++) {
        unsigned long long lo;
        unsigned int hi;
        window66(inside, nwords, 64 * w + oz * gxy + oy * gx - 1, lo, hi);
        const unsigned long long left = lo, mid = (lo >> 1) | ((unsigned long long)(hi & 1u) << 63), right = (lo >> 2) | ((unsigned long long)hi << 62);
        unsigned long long row = mid | (right & ~lx) | (left & ~fx);
        if (oy > 0) row &= ~ly;
        if (oy < 0) row &= ~fy;
        if (oz > 0) row &= ~lz;
        if (oz < 0) row &= ~fz;
        dil |= row;
        if (oy >= 0 && oz >= 0) { any |= mid | right; all &= mid & right; }
        if (oy == 0 && oz == 0) { c0 = mid; cx = right; }
        if (oy == 1 && oz == 0) cy = mid;
        if (oy == 0 && oz == 1) cz = mid;
      }
    const unsigned long long cellok = vd & ~(lx | ly | lz);
    any &= cellok; all &= cellok;
    const unsigned long long ex = (c0 ^ cx) & vd & ~lx, ey = (c0 ^ cy) & vd & ~ly, ez = (c0 ^ cz) & vd & ~lz;
    const unsigned long long v = dil & vd;
    cinc[w] = any; crossx[w] = ex; crossy[w] = ey; crossz[w] = ez; vinc[w] = v;
    ne = (unsigned int)(__popcll(ex) + __popcll(ey) + __popcll(ez)); ns = (unsigned int)__popcll(any & ~all);
    aux_pop[w] = ne | (ns << 16);  // (the marching-cubes pass reads the crossed-edge count of the word from here)
    pc = (unsigned int)__popcll(any); pv = (unsigned int)__popcll(v);
  }
  unsigned int tc, tv, te, ts;
  block_excl_scan256(pc, &tc, sh);
  block_excl_scan256(pv, &tv, sh);
  block_excl_scan256(ne, &te, sh);
  block_excl_scan256(ns, &ts, sh);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = make_uint4(tc, tv, te, ts);
}

// ranks: cbase[w] / vbase[w] = included cells / vertices in the words before w -- the sums of the workgroups before this one (plain
// cached loads of at most 16 bytes x workgroups: the kernel boundary makes them final) plus the scan inside the workgroup; the
// last workgroup leaves the totals ([0] crossed edges, [1] surface cells, [2] included cells, [3] tet vertices)
__global__ __launch_bounds__(kPB) void k_ranks(long long nwords, const unsigned long long* __restrict__ cinc, const unsigned long long* __restrict__ vinc,
                                               const uint4* __restrict__ block_sums, unsigned int* __restrict__ cbase, unsigned int* __restrict__ vbase,
                                               unsigned int* __restrict__ totals) {
  __shared__ unsigned int sh[4];
  __shared__ unsigned int s_pre[4];
  if (threadIdx.x < 4) s_pre[threadIdx.x] = 0u;
  __syncthreads();
  unsigned int acc[4] = {0u, 0u, 0u, 0u};
  for (unsigned int j = threadIdx.x; j < blockIdx.x; j += kPB) {
    const uint4 t = block_sums[j];
    acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off, 64);
    if ((threadIdx.x & 63) == 0 && acc[q]) atomicAdd(&s_pre[q], acc[q]);
  }
  __syncthreads();
  const long long w = (long long)blockIdx.x * kPB + threadIdx.x;
  const unsigned int pc = w < nwords ? (unsigned int)__popcll(cinc[w]) : 0u, pv = w < nwords ? (unsigned int)__popcll(vinc[w]) : 0u;
  unsigned int tc, tv;
  const unsigned int ec = block_excl_scan256(pc, &tc, sh), ev = block_excl_scan256(pv, &tv, sh);
  if (w < nwords) { cbase[w] = s_pre[0] + ec; vbase[w] = s_pre[1] + ev; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const uint4 mine = block_sums[blockIdx.x];
    totals[0] = s_pre[2] + mine.z; totals[1] = s_pre[3] + mine.w; totals[2] = s_pre[0] + tc; totals[3] = s_pre[1] + tv;
  }
}

// two independent scans share every launch: blockIdx.y picks the job
struct ScanJob {
  const unsigned int* in;    // per-word counts
  const unsigned int* aux;   // optional packed pair of counters (low 16 / high 16 bits), only summed
  unsigned int* sums;        // per-chunk sums, scanned in place
  unsigned int* aux_sums;    // 2 per chunk
  unsigned int* out;         // exclusive scan of `in`
  unsigned int* total;       // grand total of `in`
  unsigned int* aux_total;   // 2 grand totals of the packed counters
};
struct ScanJobs {
  ScanJob j[2];
};

__global__ __launch_bounds__(kPB) void k_chunk_sums(ScanJobs jobs, int n) {
  const unsigned int* __restrict__ in = jobs.j[blockIdx.y].in;
  const unsigned int* __restrict__ aux = jobs.j[blockIdx.y].aux;
  unsigned int* __restrict__ sums = jobs.j[blockIdx.y].sums;
  unsigned int* __restrict__ aux_sums = jobs.j[blockIdx.y].aux_sums;
  __shared__ unsigned int sh[4];
  const int base = blockIdx.x * kChunk;
  unsigned int a = 0, b = 0, c = 0;
  for (int k = 0; k < kChunk / kPB; k++) {
    const int i = base + k * kPB + threadIdx.x;
    if (i < n) {
      a += in[i];
      if (aux) { const unsigned int v = aux[i]; b += v & 0xFFFFu; c += v >> 16; }
    }
  }
  unsigned int ta, tb, tc;
  block_excl_scan256(a, &ta, sh);
  block_excl_scan256(b, &tb, sh);
  block_excl_scan256(c, &tc, sh);
  if (threadIdx.x == 0) {
    sums[blockIdx.x] = ta;
    if (aux) { aux_sums[2 * blockIdx.x] = tb; aux_sums[2 * blockIdx.x + 1] = tc; }
  }
}

// one block: exclusive scan of up to 16*kPB chunk sums in place; totals out
__global__ __launch_bounds__(kPB) void k_scan_chunks(ScanJobs jobs, int nchunks) {
  unsigned int* __restrict__ sums = jobs.j[blockIdx.x].sums;
  const unsigned int* __restrict__ aux_sums = jobs.j[blockIdx.x].aux ? jobs.j[blockIdx.x].aux_sums : nullptr;
  unsigned int* __restrict__ total_out = jobs.j[blockIdx.x].total;
  unsigned int* __restrict__ aux_total_out = jobs.j[blockIdx.x].aux_total;
  // aux_sums: two counters per chunk (may be null); aux_total_out[0], [1] receive their totals
  __shared__ unsigned int sh[4];
  unsigned int carry = 0, aux = 0, aux2 = 0;
  for (int base = 0; base < nchunks; base += kPB) {
    const int i = base + threadIdx.x;
    const unsigned int v = i < nchunks ? sums[i] : 0u;
    if (aux_sums && i < nchunks) { aux += aux_sums[2 * i]; aux2 += aux_sums[2 * i + 1]; }
    unsigned int tot;
    const unsigned int ex = block_excl_scan256(v, &tot, sh);
    if (i < nchunks) sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  unsigned int auxtot, auxtot2;
  block_excl_scan256(aux, &auxtot, sh);
  block_excl_scan256(aux2, &auxtot2, sh);
  if (threadIdx.x == 0) {
    *total_out = carry;
    if (aux_sums) { aux_total_out[0] = auxtot; aux_total_out[1] = auxtot2; }
  }
}

__global__ __launch_bounds__(kPB) void k_chunk_scan(ScanJobs jobs, int n) {
  const unsigned int* __restrict__ in = jobs.j[blockIdx.y].in;
  const unsigned int* __restrict__ chunk_base = jobs.j[blockIdx.y].sums;
  unsigned int* __restrict__ out = jobs.j[blockIdx.y].out;
  __shared__ unsigned int sh[4];
  const int base = blockIdx.x * kChunk + threadIdx.x * (kChunk / kPB);  // 16 consecutive words per thread
  unsigned int v[kChunk / kPB], s = 0;
#pragma unroll
  for (int k = 0; k < kChunk / kPB; k++) {
    v[k] = (base + k) < n ? in[base + k] : 0u;
    s += v[k];
  }
  unsigned int tot;
  unsigned int run = chunk_base[blockIdx.x] + block_excl_scan256(s, &tot, sh);
#pragma unroll
  for (int k = 0; k < kChunk / kPB; k++) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

__device__ inline unsigned int rank_of(const unsigned long long* __restrict__ mask, const unsigned int* __restrict__ base, long long i) {
  const unsigned long long w = mask[i >> 6];
  return base[i >> 6] + (unsigned int)__popcll(w & ((1ULL << (i & 63)) - 1ULL));
}

// wave-uniform lane pick of a 64-bit / 32-bit register (j is the same in every lane)
__device__ __forceinline__ unsigned long long pick64(unsigned long long v, int j) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, j);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), j);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned int pick32(unsigned int v, int j) { return (unsigned int)__builtin_amdgcn_readlane((int)v, j); }

constexpr int kRun = 62;  // mask words a wave prefetches at once: lane j holds word j of the run (and j + 1 for the row windows)

// TetMeshVertices (Tetrahedralizer.cl:39-64): included grid points compacted in grid order, xyz = the sweep's positions: the
// vertex of point p goes to rank vbase[p / 64] + popc(mask & lower bits).
constexpr int kVW = 16;  // mask words (1,024 grid points) per workgroup of k_tet_vertices
__global__ __launch_bounds__(kPB) void k_tet_vertices(Grid G, FastDiv dxy, FastDiv dx, const unsigned long long* __restrict__ vinc,
                                                      const unsigned int* __restrict__ vbase, float* __restrict__ xyz) {
  // Round 4: a WORKGROUP compacts kVW consecutive words of the mask: its vertices are one contiguous range of the output, staged in
  // LDS at their rank inside the workgroup (one pass, one barrier) and streamed out by all 256 threads, 1 KB per store instruction --
  // whole lines but for the two ends of the range.  Rounds 1-3 had every wavefront loop over its own words with an LDS round trip
  // and two wave barriers per word (32-35 us at 256^3 for 80 MB); a thread per point storing its three floats directly (12-byte
  // stride: three partial writes to every line) measured 38 us.
  __shared__ float stage[kVW * 64 * 3];
  __shared__ unsigned long long smask[kVW];
  __shared__ unsigned int sbase[kVW + 1];
  const long long nwords = (G.n_points + 63) >> 6;
  const long long w0 = (long long)blockIdx.x * kVW;
  if (threadIdx.x < kVW) {
    const long long w = w0 + threadIdx.x;
    smask[threadIdx.x] = w < nwords ? vinc[w] : 0ULL;
    sbase[threadIdx.x] = w < nwords ? vbase[w] : 0u;
  }
  __syncthreads();
  const unsigned int first = sbase[0];
  const unsigned int gx = (unsigned int)G.g[0], gxy = gx * (unsigned int)G.g[1];
  unsigned int total = 0;
#pragma unroll
  for (int k = 0; k < kVW * 64 / kPB; k++) {
    const int local = k * kPB + threadIdx.x, wl = local >> 6, lane = local & 63;  // (a wavefront = one word)
    const unsigned long long mask = smask[wl];
    if (mask == 0ULL) continue;  // (wave-uniform)
    // grid coordinates of the word's first point in scalar registers, the lane's from them (rows of at least 64 points: one carry per axis)
    const unsigned int g0 = __builtin_amdgcn_readfirstlane((unsigned int)(w0 * 64) + (unsigned int)(local & ~63));
    const unsigned int iz0 = div_by(g0, dxy), rem0 = g0 - iz0 * gxy;
    const unsigned int iy0 = div_by(rem0, dx), ix0 = rem0 - iy0 * gx;
    if ((mask >> lane) & 1ULL) {
      unsigned int ix, iy, iz;
      if (gx >= 64u) {
        ix = ix0 + (unsigned int)lane;
        const bool cx = ix >= gx;
        ix -= cx ? gx : 0u;
        iy = iy0 + (cx ? 1u : 0u);
        const bool cy = iy >= (unsigned int)G.g[1];
        iy -= cy ? (unsigned int)G.g[1] : 0u;
        iz = iz0 + (cy ? 1u : 0u);
      } else {
        const unsigned int g32 = g0 + (unsigned int)lane;
        iz = div_by(g32, dxy);
        const unsigned int rem = g32 - iz * gxy;
        iy = div_by(rem, dx);
        ix = rem - iy * gx;
      }
      float* o = &stage[3 * (sbase[wl] - first + (unsigned int)__popcll(mask & ((1ULL << lane) - 1ULL)))];
      o[0] = G.lo[0] + G.cellsize * (float)ix;
      o[1] = G.lo[1] + G.cellsize * (float)iy;
      o[2] = G.lo[2] + G.cellsize * (float)(iz + (unsigned int)G.z0);
    }
  }
  // vertices of this workgroup: up to and including its last non-empty word
  if (threadIdx.x == 0) {
    unsigned int n = 0;
    for (int k = kVW - 1; k >= 0; k--)
      if (smask[k]) { n = sbase[k] - first + (unsigned int)__popcll(smask[k]); break; }
    sbase[kVW] = n;
  }
  __syncthreads();
  total = 3u * sbase[kVW];
  float* out = xyz + 3 * (size_t)first;
  // (round 5: 16-byte stores for the body of the range -- up to three floats singly until the address is aligned, then a float4 per lane --
  // measured 29.2 us against 26.3 for this form at 256^3: the kernel waits on its two barriers per 1,024 points, not on the store width)
  for (unsigned int i = threadIdx.x; i < total; i += kPB) out[i] = stage[i];
}

// TetMeshElements (Tetrahedralizer.cl:67-132): 6 tets per included cell, corners LBN,LBF,LTN,LTF,RBN,RBF,RTN,RTF = 0..7.
// A wavefront covers one 64-bit word of the included-cell mask at a time (cells addressed by their lower-corner point), so its
// output is the contiguous range [6*cbase[word], 6*(cbase[word]+popc)) of uint4 records: lanes stage their 6 records in LDS
// at their rank inside the word and the wave then streams the range out with lane-contiguous 16-byte stores.  As in
// k_tet_vertices everything the loop needs from memory -- the cell masks and bases of the run and the vertex-mask windows
// of the four corner rows -- is fetched up front, one word per lane.
__global__ __launch_bounds__(kPB) void k_tet_elements(Grid G, const unsigned long long* __restrict__ cinc,
                                                      const unsigned int* __restrict__ cbase, const unsigned long long* __restrict__ vinc,
                                                      const unsigned int* __restrict__ vbase, uint4* __restrict__ tets) {
  __shared__ uint4 stage[kPB / 64][64 * 6];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long nwords = (G.n_points + 63) >> 6;
  const long long nwaves = (long long)gridDim.x * (kPB / 64);
  const long long gx = G.g[0], gxy = (long long)G.g[0] * G.g[1];
  const long long per = (nwords + nwaves - 1) / nwaves, wid = (long long)blockIdx.x * (kPB / 64) + wv;
  const long long end = min((wid + 1) * per, nwords);
  const long long rowoff[4] = {0, gxy, gx, gx + gxy};
  for (long long first = wid * per; first < end; first += kRun) {  // wave-uniform loops
    const int cnt = (int)min((long long)kRun, end - first);
    const unsigned long long my_mask = lane < cnt ? cinc[first + lane] : 0ULL;
    const unsigned int my_base = lane < cnt ? cbase[first + lane] : 0u;
    // vertex ranks of the four (y,z) rows of corners: the 64 lanes of a row cover 64 consecutive points, i.e. two words of
    // the vertex mask, word j + off and the next one for the run's word j (clamped: rows past the grid end are only touched
    // by lanes whose cell is not included, their ranks are unused)
    unsigned long long rm[4];
    unsigned int rb[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      const long long w = min(first + lane + (rowoff[r4] >> 6), nwords - 1);
      rm[r4] = lane <= cnt ? vinc[w] : 0ULL;
      rb[r4] = lane <= cnt ? vbase[w] : 0u;
    }
    for (int j = 0; j < cnt; j++) {
      const unsigned long long mask = pick64(my_mask, j);
      if (mask == 0ULL) continue;
      unsigned int c[8];
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const unsigned long long m0 = pick64(rm[r4], j), m1 = pick64(rm[r4], j + 1);
        const unsigned int b0 = pick32(rb[r4], j), b1 = pick32(rb[r4], j + 1);
        const int bit = (int)(rowoff[r4] & 63) + lane;                     // position inside the two-word window
        const unsigned long long mm = bit < 64 ? m0 : m1;
        const unsigned int bb = bit < 64 ? b0 : b1;
        c[r4] = bb + (unsigned int)__popcll(mm & ((1ULL << (bit & 63)) - 1ULL));
      }
      // all 8 corners of an included cell are included vertices and a rank is a prefix count in grid order, so the +x
      // corner of each row is simply the next rank
      c[4] = c[0] + 1; c[5] = c[1] + 1; c[6] = c[2] + 1; c[7] = c[3] + 1;
      if ((mask >> lane) & 1ULL) {
        enum { LBN, LBF, LTN, LTF, RBN, RBF, RTN, RTF };
        uint4* o = &stage[wv][6 * __popcll(mask & ((1ULL << lane) - 1ULL))];
        o[0] = make_uint4(c[LBN], c[LTN], c[RBN], c[LBF]);
        o[1] = make_uint4(c[RTN], c[LTN], c[LBF], c[RBN]);
        o[2] = make_uint4(c[RTN], c[LTN], c[LTF], c[LBF]);
        o[3] = make_uint4(c[RTN], c[RBN], c[LBF], c[RBF]);
        o[4] = make_uint4(c[RTN], c[LBF], c[LTF], c[RBF]);
        o[5] = make_uint4(c[RTN], c[LTF], c[RTF], c[RBF]);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int total = 6 * __popcll(mask);
      uint4* out = tets + 6 * (size_t)pick32(my_base, j);
      for (int i = lane; i < total; i += 64) out[i] = stage[wv][i];  // (non-temporal stores measured slower here: 161 vs 140 us)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

// ---- marching-cubes surface (GPUPoly::run steps 3,4,6,7: OclPolygonizer.cpp:663-757) ---------------------------------
// Triangle table: the cube table of Bloomenthal's polygonizer ("An implicit surface polygonizer", Graphics Gems IV),
// whose edge/corner/face naming the reference's _CellConfigTable.h carries: walk the crossed edges of each of the 256
// sign configurations clockwise around the faces into polygons, then fan every polygon around its last vertex.
// Generated at library load; tests pin it to the hash of the reference's own array (tests/golden/mc_table.json).
enum { eLB, eLT, eLN, eLF, eRB, eRT, eRN, eRF, eBN, eBF, eTN, eTF };
enum { fL, fR, fB, fT, fN, fF };
// corner = 4*dx + 2*dy + dz (LBN LBF LTN LTF RBN RBF RTN RTF)
const int kEdgeCorner1[12] = {0, 2, 0, 1, 4, 6, 4, 5, 0, 1, 2, 3};
const int kEdgeCorner2[12] = {1, 3, 2, 3, 5, 7, 6, 7, 4, 5, 6, 7};
const int kEdgeAxis[12] = {2, 2, 1, 1, 2, 2, 1, 1, 0, 0, 0, 0};

struct CubeTable {
  unsigned char tri[256][16];  // edge ids, 255 = end
  unsigned char nvert[256];
  // per edge: bits 0-2 first corner, bits 4-5 axis
  unsigned char edge_info[16];
};

int next_cw_edge(int edge, int face) {
  // the next edge clockwise around `face` (seen from outside the cube)
  static const int on_first[12] = {fL, fL, fL, fL, fR, fR, fR, fR, fB, fB, fT, fT};
  static const int if_first[12] = {eLF, eLN, eLB, eLT, eRN, eRF, eRT, eRB, eRB, eLB, eLT, eRT};
  static const int if_other[12] = {eBN, eTF, eTN, eBF, eBF, eTN, eBN, eTF, eLN, eRF, eRN, eLF};
  return face == on_first[edge] ? if_first[edge] : if_other[edge];
}

const CubeTable& cube_table() {
  static const CubeTable T = [] {
    static const int left_face[12] = {fB, fL, fL, fF, fR, fT, fN, fR, fN, fB, fT, fF};
    static const int right_face[12] = {fL, fT, fN, fL, fB, fR, fR, fF, fB, fF, fN, fT};
    CubeTable t;
    memset(t.tri, 255, sizeof t.tri);
    memset(t.edge_info, 0, sizeof t.edge_info);
    for (int e = 0; e < 12; e++) t.edge_info[e] = (unsigned char)(kEdgeCorner1[e] | (kEdgeAxis[e] << 4));
    for (int cfg = 0; cfg < 256; cfg++) {
      auto in = [cfg](int corner) { return (cfg >> corner) & 1; };
      bool done[12] = {false};
      int n = 0;
      for (int e0 = 0; e0 < 12; e0++) {
        if (done[e0] || in(kEdgeCorner1[e0]) == in(kEdgeCorner2[e0])) continue;
        int walk[12], nw = 0, e = e0, face = in(kEdgeCorner1[e0]) ? right_face[e0] : left_face[e0];
        do {
          e = next_cw_edge(e, face);
          done[e] = true;
          if (in(kEdgeCorner1[e]) != in(kEdgeCorner2[e])) {
            walk[nw++] = e;
            if (e != e0) face = face == left_face[e] ? right_face[e] : left_face[e];
          }
        } while (e != e0);
        // Bloomenthal prepends to his polygon list, so list order = reverse walk order; the fan apex is the list's last
        // entry (= first edge met) and the triangles come out from the far end of the list
        for (int k = nw - 3; k >= 0; k--) {
          t.tri[cfg][n++] = (unsigned char)walk[nw - 1 - k];
          t.tri[cfg][n++] = (unsigned char)walk[nw - 2 - k];
          t.tri[cfg][n++] = (unsigned char)walk[0];
        }
      }
      t.nvert[cfg] = (unsigned char)n;
    }
    return t;
  }();
  return T;
}

// 8-bit configuration of the cell whose lower corner is bit b of the word, from the eight shifted inside words
__device__ __forceinline__ int cfg_from_words(int b, unsigned long long c0, unsigned long long cz, unsigned long long cy, unsigned long long cyz,
                                              unsigned long long cx, unsigned long long cxz, unsigned long long cxy, unsigned long long cxyz) {
  return (int)(((c0 >> b) & 1ULL) | (((cz >> b) & 1ULL) << 1) | (((cy >> b) & 1ULL) << 2) | (((cyz >> b) & 1ULL) << 3) | (((cx >> b) & 1ULL) << 4) |
               (((cxz >> b) & 1ULL) << 5) | (((cxy >> b) & 1ULL) << 6) | (((cxyz >> b) & 1ULL) << 7));
}

// per 64-point word: surface-cell mask, number of surface vertices (crossed edges) and of triangle indices
__global__ __launch_bounds__(kPB) void k_surface_pops(Grid G, long long nwords, const unsigned long long* __restrict__ inside,
                                                      const unsigned long long* __restrict__ lastx, const unsigned long long* __restrict__ lasty,
                                                      const unsigned long long* __restrict__ lastz, const unsigned long long* __restrict__ valid,
                                                      const unsigned int* __restrict__ aux_pop, const unsigned char* __restrict__ nvert,
                                                      unsigned long long* __restrict__ surf, unsigned int* __restrict__ edge_pop,
                                                      unsigned int* __restrict__ idx_pop) {
  const long long w = (long long)blockIdx.x * kPB + threadIdx.x;
  if (w >= nwords) return;
  const long long gx = G.g[0], gxy = (long long)G.g[0] * G.g[1];
  const unsigned long long c0 = inside[w];
  const unsigned long long cx = fwd_word(inside, nwords, w, 1), cy = fwd_word(inside, nwords, w, gx), cz = fwd_word(inside, nwords, w, gxy);
  const unsigned long long cxy = fwd_word(inside, nwords, w, gx + 1), cxz = fwd_word(inside, nwords, w, gxy + 1),
                           cyz = fwd_word(inside, nwords, w, gxy + gx), cxyz = fwd_word(inside, nwords, w, gxy + gx + 1);
  const unsigned long long cellok = valid[w] & ~(lastx[w] | lasty[w] | lastz[w]);
  unsigned long long m = ((c0 | cx | cy | cz | cxy | cxz | cyz | cxyz) & ~(c0 & cx & cy & cz & cxy & cxz & cyz & cxyz)) & cellok;
  surf[w] = m;
  unsigned int n = 0;
  while (m) {
    const int b = __builtin_ctzll(m);
    n += nvert[cfg_from_words(b, c0, cz, cy, cyz, cx, cxz, cxy, cxyz)];
    m &= m - 1;
  }
  idx_pop[w] = n;
  edge_pop[w] = aux_pop[w] & 0xFFFFu;
}

// Work lists for the two emission kernels, one thread per 64-point word (after the scans): every crossed edge of the word
// in output order (point-major, X Y Z) -> elist[first vertex id of the word + r] = point << 2 | axis, and every triangle
// of every surface cell -> tlist[first triangle id of the word + r] = (lower-corner point, config | local triangle << 8).
// Crossings are sparse and clustered (a sphere crosses a 64-point word of a grid row once or twice), so emission is
// balanced by giving every vertex and every triangle its own lane instead of every word its own wavefront.
__global__ __launch_bounds__(kPB) void k_surface_lists(Grid G, long long nwords, const unsigned long long* __restrict__ inside,
                                                       const unsigned long long* __restrict__ surf, const unsigned long long* __restrict__ crossx,
                                                       const unsigned long long* __restrict__ crossy, const unsigned long long* __restrict__ crossz,
                                                       const unsigned int* __restrict__ ebase, const unsigned int* __restrict__ ibase,
                                                       const unsigned char* __restrict__ nvert, unsigned long long* __restrict__ elist,
                                                       uint2* __restrict__ tlist) {
  const long long w = (long long)blockIdx.x * kPB + threadIdx.x;
  if (w >= nwords) return;
  const unsigned long long mx = crossx[w], my = crossy[w], mz = crossz[w];
  unsigned long long m = mx | my | mz;
  if (m) {
    unsigned long long* e = elist + ebase[w];
    while (m) {
      const int b = __builtin_ctzll(m);
      m &= m - 1;
      const unsigned long long p4 = (unsigned long long)(w * 64 + b) << 2;
      if ((mx >> b) & 1ULL) *e++ = p4;
      if ((my >> b) & 1ULL) *e++ = p4 | 1ULL;
      if ((mz >> b) & 1ULL) *e++ = p4 | 2ULL;
    }
  }
  unsigned long long sm = surf[w];
  if (sm) {
    const long long gx = G.g[0], gxy = (long long)G.g[0] * G.g[1];
    const unsigned long long c0 = inside[w];
    const unsigned long long cx = fwd_word(inside, nwords, w, 1), cy = fwd_word(inside, nwords, w, gx), cz = fwd_word(inside, nwords, w, gxy);
    const unsigned long long cxy = fwd_word(inside, nwords, w, gx + 1), cxz = fwd_word(inside, nwords, w, gxy + 1),
                             cyz = fwd_word(inside, nwords, w, gxy + gx), cxyz = fwd_word(inside, nwords, w, gxy + gx + 1);
    uint2* t = tlist + ibase[w] / 3;
    while (sm) {
      const int b = __builtin_ctzll(sm);
      sm &= sm - 1;
      const int cfg = cfg_from_words(b, c0, cz, cy, cyz, cx, cxz, cxy, cxyz);
      const int ntri = nvert[cfg] / 3;
      for (int k = 0; k < ntri; k++) *t++ = make_uint2((unsigned int)(w * 64 + b), (unsigned int)(cfg | (k << 8)));
    }
  }
}

// ComputeVertexAttribs (Polygonizer.cl:1429-1561, linear root), one lane per surface vertex: the 4 field evaluations
// per vertex (root + forward-difference normal) are spread over full wavefronts and all stores are contiguous.
template <int SEM>
__global__ __launch_bounds__(kPB) void k_surface_vertices(Grid G, long long nv, const unsigned long long* __restrict__ elist,
                                                          const Instr* __restrict__ prog, int n_instr, int n_prims,
                                                          const float* __restrict__ prims, const float* __restrict__ mtx, const float* __restrict__ cbox,
                                                          const float* __restrict__ fval, const unsigned long long* __restrict__ vinc,
                                                          const unsigned int* __restrict__ vbase, float* __restrict__ pos, float* __restrict__ nrm,
                                                          uint2* __restrict__ ends, float* __restrict__ frac) {
  extern __shared__ float stack[];
  const long long j = (long long)blockIdx.x * kPB + threadIdx.x;
  if (j >= nv) return;
  const float delta = 0.0001f, dinv = 1.0f / 0.0001f;  // NORMAL_DELTA, Polygonizer.cl
  const unsigned long long ent = elist[j];
  const long long p = (long long)(ent >> 2);
  const int axis = (int)(ent & 3ULL);
  const long long nb = p + (axis == 0 ? 1LL : (axis == 1 ? (long long)G.g[0] : (long long)G.g[0] * G.g[1]));
  // the two ends of the edge: f as the sweep stored it, positions by the sweep's own expression (what the float4 grid held, bit for bit)
  const long long gxl = G.g[0], gxyl = gxl * G.g[1];
  const long long piz = p / gxyl, prem = p - piz * gxyl, piy = prem / gxl, pix = prem - piy * gxl;
  float4 va, vb;
  va.x = G.lo[0] + G.cellsize * (float)(unsigned int)pix; va.y = G.lo[1] + G.cellsize * (float)(unsigned int)piy;
  va.z = G.lo[2] + G.cellsize * (float)((unsigned int)piz + (unsigned int)G.z0); va.w = fval[p];
  vb.x = axis == 0 ? G.lo[0] + G.cellsize * (float)((unsigned int)pix + 1u) : va.x;
  vb.y = axis == 1 ? G.lo[1] + G.cellsize * (float)((unsigned int)piy + 1u) : va.y;
  vb.z = axis == 2 ? G.lo[2] + G.cellsize * (float)((unsigned int)piz + 1u + (unsigned int)G.z0) : va.z;
  vb.w = fval[nb];
  const float t = (kIso - va.w) / (vb.w - va.w);
  const float x = va.x + t * (vb.x - va.x), y = va.y + t * (vb.y - va.y), z = va.z + t * (vb.z - va.z);
  float* stk = stack + threadIdx.x;
  const float f = eval_field<SEM>(prog, n_instr, n_prims, prims, mtx, cbox, x, y, z, stk);
  float gx = eval_field<SEM>(prog, n_instr, n_prims, prims, mtx, cbox, x + delta, y, z, stk);
  float gy = eval_field<SEM>(prog, n_instr, n_prims, prims, mtx, cbox, x, y + delta, z, stk);
  float gz = eval_field<SEM>(prog, n_instr, n_prims, prims, mtx, cbox, x, y, z + delta, stk);
  gx = -1.0f * (dinv * (gx - f)); gy = -1.0f * (dinv * (gy - f)); gz = -1.0f * (dinv * (gz - f));
  const float len = sqrtf(gx * gx + gy * gy + gz * gz);
  float* o = pos + 3 * (size_t)j;
  o[0] = x; o[1] = y; o[2] = z;
  o = nrm + 3 * (size_t)j;
  o[0] = gx / len; o[1] = gy / len; o[2] = gz / len;
  // the two grid points of the edge are vertices of the tet mesh of the same grid (a crossed edge belongs to a cell
  // with config != 0): remember their tet-mesh ids and the interpolation weight for fb_poly_interpolate_displacements
  ends[j] = make_uint2(rank_of(vinc, vbase, p), rank_of(vinc, vbase, nb));
  frac[j] = t;
}

// ComputeElements (Polygonizer.cl:1610-1670), one lane per triangle: three table entries, each resolved to the vertex id
// of its grid edge = first id of the edge's lower grid point (exclusive scan of the edge counts, as rank of the point in
// the three crossing masks) + X before Y before Z on that point.
__global__ __launch_bounds__(kPB) void k_surface_elements(Grid G, long long ntri, const uint2* __restrict__ tlist,
                                                          const unsigned long long* __restrict__ crossx, const unsigned long long* __restrict__ crossy,
                                                          const unsigned long long* __restrict__ crossz, const unsigned int* __restrict__ ebase,
                                                          const unsigned char* __restrict__ tri, const unsigned char* __restrict__ edge_info,
                                                          unsigned int* __restrict__ indices) {
  const long long t = (long long)blockIdx.x * kPB + threadIdx.x;
  if (t >= ntri) return;
  const uint2 rec = tlist[t];
  const long long gx = G.g[0], gxy = (long long)G.g[0] * G.g[1];
  const unsigned char* row = tri + 16 * (int)(rec.y & 255u) + 3 * (int)(rec.y >> 8);
  unsigned int id[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int info = edge_info[row[k]];
    const int corner = info & 7, axis = info >> 4;
    const long long q = (long long)rec.x + ((corner >> 2) & 1) + ((corner >> 1) & 1) * gx + (corner & 1) * gxy;
    const long long qw = q >> 6;
    const unsigned long long bit = 1ULL << (q & 63), low = bit - 1ULL;
    const unsigned long long cx = crossx[qw], cy = crossy[qw], cz = crossz[qw];
    unsigned int v = ebase[qw] + (unsigned int)(__popcll(cx & low) + __popcll(cy & low) + __popcll(cz & low));
    if (axis >= 1) v += (cx & bit) ? 1u : 0u;
    if (axis == 2) v += (cy & bit) ? 1u : 0u;
    id[k] = v;
  }
  unsigned int* o = indices + 3 * (size_t)t;
  o[0] = id[0]; o[1] = id[1]; o[2] = id[2];
}

// FieldComputer::fieldValueAndColor (Polygonizer.cpp:2110-2353) through the compiled order: the reference evaluates the field
// once, keeping every primitive's and operator's value, and then walks the operators a second time IN THE SAME ORDER to
// combine the colours; here both happen in one pass over the program.  Every value slot has five channels (field as the
// operator sees it, field as the colour pass sees it, r, g, b) in a global scratch array [slot][channel][thread]; the
// evaluation runs once per surface vertex, so its speed does not matter.  Rules: range operators add 2 f * colour of their
// primitives onto the RUNNING colour; blend / Ricci 2 f_l c_l + 2 f_r c_r; union / intersection channel-wise max / min of
// the weighted colours; (smooth) difference the child whose value the operator took; warps the child's colour; an
// instanced subtree under a binary operator shows its own colour walk WITHOUT the box cull (computeInstancedNodeFieldAndColor,
// :2355-2410), inside a range it counts with the instance node's colour and the culled field.
__device__ float eval_field_color(const Instr* __restrict__ prog, int n_instr, int n_prims, const float* __restrict__ prims,
                                  const float* __restrict__ mtx, float x, float y, float z, float* __restrict__ S, size_t nth, float rgb[3]) {
  float out = 0.0f, col[3] = {0.0f, 0.0f, 0.0f};
  auto at = [&](int slot, int ch) -> float& { return S[((size_t)slot * 5 + ch) * nth]; };
  auto pcol = [&](int i, int k) { return prims[20 * i + 16 + k]; };
  if (n_instr == 0) {
    for (int i = 0; i < n_prims; i++) out = out + prim_field(prims, mtx, i, x, y, z);
    for (int k = 0; k < 3; k++) rgb[k] = pcol(0, k);
    return out;
  }
  for (int pc = 0; pc < n_instr; pc++) {
    const Instr& in = prog[pc];
    switch (in.kind) {
      case 0:
        for (int i = in.a; i <= in.b; i++) {
          const float f = prim_field(prims, mtx, i, x, y, z);
          out = out + f;
          const float cur = 2.0f * (0.5f + f) - 1.0f;
          for (int k = 0; k < 3; k++) col[k] = col[k] + cur * pcol(i, k);
        }
        break;
      case 1: {
        float lf, lfc, rf = 0.0f, rfc = 0.0f, lc[3], rc[3] = {0.0f, 0.0f, 0.0f};
        if (in.a >= 0) { lf = lfc = prim_field(prims, mtx, in.a, x, y, z); for (int k = 0; k < 3; k++) lc[k] = pcol(in.ca, k); }
        else { const int s = -1 - in.a; lf = at(s, 0); lfc = at(s, 1); for (int k = 0; k < 3; k++) lc[k] = at(s, 2 + k); }
        if (!in.unary) {
          if (in.b >= 0) { rf = rfc = prim_field(prims, mtx, in.b, x, y, z); for (int k = 0; k < 3; k++) rc[k] = pcol(in.cb, k); }
          else { const int s = -1 - in.b; rf = at(s, 0); rfc = at(s, 1); for (int k = 0; k < 3; k++) rc[k] = at(s, 2 + k); }
        }
        out = apply_op(in.optype, lf, rf, in.p0, in.p1, out);
        const float wa = 2.0f * (0.5f + lfc) - 1.0f, wb = 2.0f * (0.5f + rfc) - 1.0f;
        switch (in.optype) {
          case opBlend: case opRicciBlend: for (int k = 0; k < 3; k++) col[k] = wa * lc[k] + wb * rc[k]; break;
          case opUnion: for (int k = 0; k < 3; k++) col[k] = fmaxf(wa * lc[k], wb * rc[k]); break;
          case opIntersect: for (int k = 0; k < 3; k++) col[k] = fminf(wa * lc[k], wb * rc[k]); break;
          case opDif: case opSmoothDif: {
            const float a = (lfc == out) ? 1.0f : 0.0f, b = ((1.0f - rfc) == out) ? 1.0f : 0.0f;
            for (int k = 0; k < 3; k++) col[k] = a * lc[k] + b * rc[k];
          } break;
          case opWarpBend: case opWarpTwist: case opWarpTaper: case opWarpShear: for (int k = 0; k < 3; k++) col[k] = lc[k]; break;
          default: break;
        }
      } break;
      case 2: {  // frame: x, y, z, out, inside in channel 0 of the 5 frame slots, the running colour in channel 1 of the first 3
        at(in.b, 0) = x; at(in.b + 1, 0) = y; at(in.b + 2, 0) = z; at(in.b + 3, 0) = out;
        for (int k = 0; k < 3; k++) at(in.b + k, 1) = col[k];
        if (in.a != 0) {
          const float* m = mtx + 12 * in.a;
          const float tx = m[0] * x + m[1] * y + m[2] * z + m[3];
          const float ty = m[4] * x + m[5] * y + m[6] * z + m[7];
          const float tz = m[8] * x + m[9] * y + m[10] * z + m[11];
          x = tx; y = ty; z = tz;
        }
        const bool inside = x >= in.lo[0] && in.hi[0] >= x && y >= in.lo[1] && in.hi[1] >= y && z >= in.lo[2] && in.hi[2] >= z;
        at(in.b + 4, 0) = inside ? 1.0f : 0.0f;
        out = 0.0f; col[0] = col[1] = col[2] = 0.0f;
      } continue;
      case 3: {
        const float v = out, vc[3] = {col[0], col[1], col[2]};
        const bool inside = at(in.a + 4, 0) != 0.0f;
        x = at(in.a, 0); y = at(in.a + 1, 0); z = at(in.a + 2, 0); out = at(in.a + 3, 0);
        for (int k = 0; k < 3; k++) col[k] = at(in.a + k, 1);
        at(in.dst, 0) = inside ? v : 0.0f;   // what fieldValue saw (isOutsideOp cull)
        at(in.dst, 1) = v;                   // what the colour pass re-evaluates
        for (int k = 0; k < 3; k++) at(in.dst, 2 + k) = vc[k];
      } continue;
      default: {
        const float f = at(in.a, 0);
        out = out + f;
        if (in.cb) { for (int k = 0; k < 3; k++) col[k] = pcol(in.ca, k); }
        else { const float cur = 2.0f * (0.5f + f) - 1.0f; for (int k = 0; k < 3; k++) col[k] = col[k] + cur * pcol(in.ca, k); }
      } continue;
    }
    if (in.dst >= 0) { at(in.dst, 0) = out; at(in.dst, 1) = out; for (int k = 0; k < 3; k++) at(in.dst, 2 + k) = col[k]; }
  }
  for (int k = 0; k < 3; k++) rgb[k] = col[k];
  return out;
}

// vertex colours of the surface (ComputeVertexAttribs' ComputeFieldAndColor, Polygonizer.cl:1544-1553), RGBA with A = 1
__global__ __launch_bounds__(kPB) void k_surface_colors(long long first, long long count, const float* __restrict__ pos,
                                                        const Instr* __restrict__ prog, int n_instr, int n_prims, const float* __restrict__ prims,
                                                        const float* __restrict__ mtx, float* __restrict__ scratch, long long nth, float4* __restrict__ out) {
  const long long t = (long long)blockIdx.x * kPB + threadIdx.x;
  if (t >= count) return;
  const long long j = first + t;
  float rgb[3];
  (void)eval_field_color(prog, n_instr, n_prims, prims, mtx, pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], scratch + t, (size_t)nth, rgb);
  out[j] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
}

// also: the field and colour of arbitrary points (FieldComputer::fieldValueAndColor), for the parity tests
__global__ __launch_bounds__(kPB) void k_field_color_array(long long count, float4* __restrict__ pts, const Instr* __restrict__ prog, int n_instr,
                                                           int n_prims, const float* __restrict__ prims, const float* __restrict__ mtx,
                                                           float* __restrict__ scratch, long long nth, float* __restrict__ rgb_out) {
  const long long t = (long long)blockIdx.x * kPB + threadIdx.x;
  if (t >= count) return;
  float4 p = pts[t];
  float rgb[3];
  p.w = eval_field_color(prog, n_instr, n_prims, prims, mtx, p.x, p.y, p.z, scratch + t, (size_t)nth, rgb);
  pts[t] = p;
  rgb_out[3 * t] = rgb[0]; rgb_out[3 * t + 1] = rgb[1]; rgb_out[3 * t + 2] = rgb[2];
}

// ComputeOffSurfacePointsAndFields (Polygonizer.cl:1329-1350): v +- len * normal and the field there, (x, y, z, f) pairs
template <int SEM>
__global__ __launch_bounds__(kPB) void k_off_surface(long long nv, float len, const float* __restrict__ pos, const float* __restrict__ nrm,
                                                     const Instr* __restrict__ prog, int n_instr, int n_prims, const float* __restrict__ prims,
                                                     const float* __restrict__ mtx, const float* __restrict__ cbox, float4* __restrict__ out) {
  extern __shared__ float stack[];
  const long long j = (long long)blockIdx.x * kPB + threadIdx.x;
  if (j >= nv) return;
  const float vx = pos[3 * j], vy = pos[3 * j + 1], vz = pos[3 * j + 2];
  const float dx = len * nrm[3 * j], dy = len * nrm[3 * j + 1], dz = len * nrm[3 * j + 2];
  float* stk = stack + threadIdx.x;
  const float ox = vx + dx, oy = vy + dy, oz = vz + dz, ix = vx - dx, iy = vy - dy, iz = vz - dz;
  const float fo = eval_field<SEM>(prog, n_instr, n_prims, prims, mtx, cbox, ox, oy, oz, stk);
  const float fi = eval_field<SEM>(prog, n_instr, n_prims, prims, mtx, cbox, ix, iy, iz, stk);
  out[2 * j] = make_float4(ox, oy, oz, fo);
  out[2 * j + 1] = make_float4(ix, iy, iz, fi);
}

// surface vertex = rest + da + t (db - da): the FEM displacements of the two tet-mesh nodes of its grid edge, weighted as
// the vertex itself was placed on the edge
__global__ __launch_bounds__(kPB) void k_interpolate_displacements(long long nv, const float* __restrict__ rest, const uint2* __restrict__ ends,
                                                                   const float* __restrict__ frac, const double* __restrict__ u, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * kPB + threadIdx.x;
  if (i >= 3 * nv) return;
  const long long v = i / 3;
  const int c = (int)(i - 3 * v);
  const uint2 e = ends[v];
  const float da = (float)u[3 * (size_t)e.x + c], db = (float)u[3 * (size_t)e.y + c];
  out[i] = rest[i] + (da + frac[v] * (db - da));
}

// ApplyVertexDeformations (Polygonizer.cl:1417-1426) with the double -> float narrowing of GPUPoly::applyFemDisplacements
// (OclPolygonizer.cpp:1559-1564) folded in
__global__ __launch_bounds__(kPB) void k_apply_displacements(long long n, const float* __restrict__ rest, const double* __restrict__ u,
                                                             float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * kPB + threadIdx.x;
  if (i < n) out[i] = rest[i] + (float)u[i];
}

}  // namespace

struct fb_poly_s {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<float> header, ops, prims, mtx;
  int n_ops = 0, n_prims = 0, n_mtx = 0;
  std::vector<Instr> prog;
  int depth = 1;
  DevBuf<Instr> d_prog;
  DevBuf<float> d_prims, d_mtx, d_pbox, d_cbox;
  std::vector<float> pbox;  // support box of every primitive (support_boxes)
  std::vector<float> cbox;  // SEM_CPU_BOX: the reference's primitive boxes, given by the caller (fb_poly_set_field_semantics)
  int sem = SEM_CPU;
  Grid G;
  int gz_total = 0;  // point planes of the whole grid this one is a slab of (= G.g[2] for a grid of its own)
  bool have_grid = false, classified = false, tetra = false, materialized = false;
  DevBuf<float> fval;    // the field at every grid point, as k_sweep stores it
  DevBuf<float4> grid;   // (x, y, z, f) per point: materialised by fb_poly_read_grid only
  bool grid_current = false;
  DevBuf<unsigned long long> inside, cinc, vinc, lastx, lasty, lastz, valid, crossx, crossy, crossz, firstx, firsty, firstz;
  DevBuf<uint4> block_sums;                // k_classify -> k_ranks: included cells, tet vertices, crossed edges, surface cells per workgroup
  long long n_words = 0;
  DevBuf<unsigned char> config, flags;
  DevBuf<unsigned int> aux_pop, cbase, vbase, vsum;
  DevBuf<unsigned int> totals;  // [0] crossed edges [1] surface cells [2] included cells [3] tet vertices
  DevBuf<float> tv;
  DevBuf<uint4> tt;
  // marching-cubes surface
  bool surfaced = false;
  DevBuf<unsigned char> d_tri, d_nvert, d_edge_info;
  DevBuf<unsigned long long> surf;
  DevBuf<unsigned int> edge_pop, idx_pop, ebase, ibase, esum, isum;
  DevBuf<float> sv, sn, deformed, sfrac;
  DevBuf<uint2> sends, tlist;
  DevBuf<unsigned long long> elist;
  DevBuf<unsigned int> si;
  DevBuf<double> disp;
  fb_poly_counts counts;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

namespace {

#define CHECK_POLY(h)                                    \
  if (!(h)) return fail(FB_EINVAL, "null poly handle"); \
  FB_HIP(hipSetDevice((h)->device))

// Emulates the operator stack walk of FieldComputer::fieldValue (Polygonizer.cpp:1938-2080) once on the host and records
// the order in which operators get evaluated.  Operator results go to numbered slots (the reference keeps one float per
// operator; here a slot is recycled as soon as its consumer has been emitted).  An instance of an operator
// (computePrimitiveField :1879-1901 calls fieldValue on the original subtree at the mapped point) is expanded in place:
// ENTER (save point and running field, map the point, cull against the original's box) .. subtree .. LEAVE (result to a
// slot, restore).  Instances of primitives are followed by prim_field on the device.
struct TreeCompiler {
  fb_poly_s* h;
  std::vector<Instr> prog;
  std::vector<char> used;
  int depth = 1, nest = 0;

  int alloc(int n) {
    for (int base = 0;; base++) {
      bool ok = true;
      for (int k = 0; k < n && ok; k++) ok = base + k >= (int)used.size() || !used[base + k];
      if (!ok) continue;
      if ((int)used.size() < base + n) used.resize(base + n, 0);
      for (int k = 0; k < n; k++) used[base + k] = 1;
      depth = std::max(depth, base + n);
      return base;
    }
  }
  void release(int base, int n) { for (int k = 0; k < n; k++) used[base + k] = 0; }
  const float* prim(int i) const { return h->prims.data() + 20 * (size_t)i; }
  const float* op(int i) const { return h->ops.data() + 16 * (size_t)i; }

  // does primitive i, followed through instances of primitives, end in an instance of an operator?
  int resolves_to_op(int i, bool* yes) const {
    for (int hop = 0; hop < 8; hop++) {
      const float* P = prim(i);
      if ((int)P[0] != primInstance) { *yes = false; return FB_OK; }
      const int origin = (int)P[12], is_op = (int)P[14];
      if (is_op) {
        if (origin < 0 || origin >= h->n_ops) return fail(FB_EINVAL, "primitive %d instances operator %d of %d", i, origin, h->n_ops);
        *yes = true;
        return FB_OK;
      }
      if (origin < 0 || origin >= h->n_prims) return fail(FB_EINVAL, "primitive %d instances primitive %d of %d", i, origin, h->n_prims);
      i = origin;
    }
    return fail(FB_EINVAL, "instance chain of primitive %d does not end", i);
  }

  // value of primitive i (known to resolve to an operator instance) -> new slot
  int emit_instance(int i, int* slot) {
    if (++nest > 8) return fail(FB_EINVAL, "instanced subtrees nest deeper than 8 (primitive %d): an instance of its own ancestor?", i);
    const float* P = prim(i);
    const int origin = (int)P[12], is_op = (int)P[14];
    Instr in;
    memset(&in, 0, sizeof in);
    in.kind = 2; in.a = (int)P[1]; in.b = alloc(5); in.dst = -1; in.ca = i;
    for (int a = 0; a < 3; a++) { in.lo[a] = -FLT_MAX; in.hi[a] = FLT_MAX; }
    if (is_op)
      for (int a = 0; a < 3; a++) { in.lo[a] = op(origin)[8 + a]; in.hi[a] = op(origin)[12 + a]; }
    const size_t enter = prog.size();
    const int frame = in.b;
    prog.push_back(in);
    if (is_op) {
      FB_TRY(subtree(origin));
    } else {  // instance of a primitive that is itself an instance of an operator
      int inner = -1;
      FB_TRY(emit_instance(origin, &inner));
      Instr add;
      memset(&add, 0, sizeof add);
      add.kind = 4; add.a = inner; add.dst = -1;
      add.ca = origin; add.cb = 1;  // computeInstancedNodeFieldAndColor: the colour of the original primitive node
      prog.push_back(add);
      release(inner, 1);
    }
    Instr out;
    memset(&out, 0, sizeof out);
    out.kind = 3; out.a = frame;
    release(frame, 5);
    out.dst = *slot = alloc(1);
    prog[enter].skip = (int)prog.size();
    prog.push_back(out);
    if (prog.size() > (1u << 20)) return fail(FB_EINVAL, "instanced BlobTree expands to more than 2^20 evaluation steps");
    --nest;
    return FB_OK;
  }

  // primitive whose colour an operand primitive shows: itself, or -- for an instance of a primitive -- its immediate
  // original (computeInstancedNodeFieldAndColor, Polygonizer.cpp:2403-2409)
  int color_source(int i) const { return ((int)prim(i)[0] == primInstance && (int)prim(i)[14] == 0) ? (int)prim(i)[12] : i; }

  // operand reference of a primitive child: the primitive itself, or the slot of its expanded instance
  int operand(int i, int* ref, int* slot) {
    bool yes = false;
    FB_TRY(resolves_to_op(i, &yes));
    *slot = -1;
    if (!yes) { *ref = i; return FB_OK; }
    FB_TRY(emit_instance(i, slot));
    *ref = -1 - *slot;
    return FB_OK;
  }

  // the walk of fieldValue(.., idxRootNode = root); on return the root's value is the running field
  int subtree(int root) {
    const int nops = h->n_ops;
    std::vector<char> computed(nops, 0);
    std::vector<int> where(nops, -1);  // slot of a computed operator
    std::vector<int> stk{root};
    size_t guard = 0;
    while (!stk.empty()) {
      if (++guard > (size_t)4 * nops + 16) return fail(FB_EINVAL, "BlobTree operator graph is not a tree");
      const int n = stk.back();
      const float* o = op(n);
      const int type = (int)o[0], lc = (int)o[1], rc = (int)o[2], fl = (int)o[7];
      const bool unary = fl & ofIsUnaryOp, range = fl & ofChildIndexIsRange, lop = fl & ofLeftChildIsOp, rop = fl & ofRightChildIsOp;
      Instr in;
      memset(&in, 0, sizeof in);
      in.dst = -1;
      if (range) {
        if (lc < 0 || rc >= h->n_prims || lc > rc) return fail(FB_EINVAL, "operator %d: bad primitive range [%d,%d]", n, lc, rc);
        stk.pop_back();
        int first = lc;  // runs of plain primitives, split at operator instances (same summation order)
        for (int i = lc; i <= rc; i++) {
          bool yes = false;
          FB_TRY(resolves_to_op(i, &yes));
          if (!yes) continue;
          if (first < i) { in.kind = 0; in.a = first; in.b = i - 1; prog.push_back(in); }
          int s = -1;
          FB_TRY(emit_instance(i, &s));
          Instr add;
          memset(&add, 0, sizeof add);
          add.kind = 4; add.a = s; add.dst = -1;
          add.ca = i; add.cb = 0;  // inside a range the instance counts with its own node colour
          prog.push_back(add);
          release(s, 1);
          first = i + 1;
        }
        if (first <= rc) { in.kind = 0; in.a = first; in.b = rc; prog.push_back(in); }
      } else {
        if ((lop && (lc < 0 || lc >= nops)) || (!lop && (lc < 0 || lc >= h->n_prims))) return fail(FB_EINVAL, "operator %d: bad left child %d", n, lc);
        if (!unary && ((rop && (rc < 0 || rc >= nops)) || (!rop && (rc < 0 || rc >= h->n_prims)))) return fail(FB_EINVAL, "operator %d: bad right child %d", n, rc);
        const bool ready = unary ? !(lop && !computed[lc]) : !((lop && !computed[lc]) || (rop && !computed[rc]));
        if (!ready) {
          if (lop && !computed[lc]) stk.push_back(lc);
          if (!unary && rop && !computed[rc]) stk.push_back(rc);
          continue;
        }
        stk.pop_back();
        int sa = -1, sb = -1;
        if (lop) { in.a = -1 - where[lc]; sa = where[lc]; } else FB_TRY(operand(lc, &in.a, &sa));
        if (unary) in.b = 0;
        else if (rop) { in.b = -1 - where[rc]; sb = where[rc]; } else FB_TRY(operand(rc, &in.b, &sb));
        in.kind = 1; in.optype = type; in.unary = unary ? 1 : 0;
        in.p0 = o[4]; in.p1 = o[5];
        in.ca = in.a >= 0 ? color_source(in.a) : 0;
        in.cb = (!unary && in.b >= 0) ? color_source(in.b) : 0;
        if (sa >= 0) release(sa, 1);
        if (sb >= 0 && sb != sa) release(sb, 1);
        prog.push_back(in);
      }
      computed[n] = 1;
      if (n != root) {  // the result of an inner operator is kept for its parent
        where[n] = alloc(1);
        prog.back().dst = where[n];
        // a RANGE that ended in an instance has an ADDSLOT last: give the store its own step
        if (prog.back().kind == 4) {
          prog.back().dst = -1;
          Instr keep;
          memset(&keep, 0, sizeof keep);
          keep.kind = 0; keep.a = 0; keep.b = -1; keep.dst = where[n];  // empty range: only stores the running field
          prog.push_back(keep);
        }
      }
    }
    return FB_OK;
  }
};

// LinearBlobTree::setTraversalRoute (LinearBlobTree.cpp:333-429): start operator and `next` links of the stackless walk.
// The reference loop re-examines an operator after its operator children have been linked and pushes them again, so it
// never ends on an operator with two operator children; that (and a malformed graph) is reported instead of spinning.
int traversal_route(const fb_poly_s* h, std::vector<int>* next, int* start) {
  const int n = h->n_ops;
  next->assign(n, kNullBlob);
  *start = kNullBlob;
  std::vector<int> ops{0}, brk{kNullBlob};
  auto flags = [&](int o) { return (int)h->ops[16 * (size_t)o + 7]; };
  auto child = [&](int o, int k) { return (int)h->ops[16 * (size_t)o + 1 + k]; };
  auto is_op_id = [&](int c) { return c >= 0 && c < n; };
  size_t guard = 0;
  while (!ops.empty()) {
    if (++guard > (size_t)8 * n + 16) return fail(FB_EINVAL, "OpenCL field semantics: the reference's traversal-route builder does not terminate on this tree (an operator with two operator children)");
    int o = ops.back(), fl = flags(o);
    bool is_break = fl & ofBreak, lop = fl & ofLeftChildIsOp, rop = fl & ofRightChildIsOp;
    if (!lop && !rop) {
      if (brk.empty()) return fail(FB_EINVAL, "OpenCL field semantics: no traversal route (operator %d)", o);
      ops.pop_back();
      const int b = brk.back();
      brk.pop_back();
      if (b == kNullBlob) *start = o; else (*next)[b] = o;
      if (is_break) brk.push_back(o);
      while (!is_break && !ops.empty()) {
        o = ops.back();
        ops.pop_back();
        fl = flags(o);
        is_break = fl & ofBreak; lop = fl & ofLeftChildIsOp; rop = fl & ofRightChildIsOp;
        if (is_break) brk.push_back(o);
        if (lop) { if (!is_op_id(child(o, 0))) return fail(FB_EINVAL, "operator %d: bad left child", o); (*next)[child(o, 0)] = o; }
        if (rop) { if (!is_op_id(child(o, 1))) return fail(FB_EINVAL, "operator %d: bad right child", o); (*next)[child(o, 1)] = o; }
      }
    } else {
      if (lop) { if (!is_op_id(child(o, 0))) return fail(FB_EINVAL, "operator %d: bad left child", o); ops.push_back(child(o, 0)); }
      if (rop) { if (!is_op_id(child(o, 1))) return fail(FB_EINVAL, "operator %d: bad right child", o); ops.push_back(child(o, 1)); }
    }
  }
  if ((*next)[0] != kNullBlob) return fail(FB_EINVAL, "OpenCL field semantics: root operator has a next link (LinearBlobTree.cpp:408-412)");
  for (int i = 1; i < n; i++)
    if ((*next)[i] < 0 || (*next)[i] >= n) return fail(FB_EINVAL, "OpenCL field semantics: operator %d is not on the traversal route", i);
  return FB_OK;
}

// SEM_OPENCL: runs ComputeField / ComputeBranchField (Polygonizer.cl:781-886) symbolically.  The kernel keeps the values of
// finished subtrees in two registers per level (lf, rf: the outer pair of ComputeField, and the by-value copies of
// ComputeBranchField); which register an operator reads and writes depends on flags alone.  A register holds the constant 0,
// the value of a primitive at the query point (re-evaluated where it is read -- same value), or the result of an earlier step,
// which then lives in a slot for as long as a register names it.
int compile_tree_cl(fb_poly_s* h) {
  h->prog.clear();
  h->depth = 1;
  if (h->n_ops == 0) return FB_OK;
  std::vector<int> next;
  int start = kNullBlob;
  FB_TRY(traversal_route(h, &next, &start));
  const int n = h->n_ops;
  struct Val { int kind, id; };  // kind 0: constant 0, 1: primitive id, 2: slot id
  auto ref = [](const Val& v) { return v.kind == 0 ? kZeroOperand : (v.kind == 1 ? v.id : -1 - v.id); };
  Val OL{0, 0}, OR{0, 0}, IL{0, 0}, IR{0, 0};
  std::vector<char> used;
  auto new_slot = [&]() {
    const Val* regs[4] = {&OL, &OR, &IL, &IR};
    for (size_t k = 0; k < used.size(); k++) used[k] = 0;
    for (auto* r : regs)
      if (r->kind == 2) used[r->id] = 1;
    for (size_t k = 0; k < used.size(); k++)
      if (!used[k]) return (int)k;
    used.push_back(1);
    return (int)used.size() - 1;
  };
  size_t guard = 0;
  for (int op = start; op != kNullBlob;) {
    IL = OL; IR = OR;
    int next_op = kNullBlob;
    bool is_right = false;
    Val field{0, 0};
    for (int b = op; b != kNullBlob;) {
      if (++guard > (size_t)4 * n + 16) return fail(FB_EINVAL, "OpenCL field semantics: traversal route does not end");
      const float* o = h->ops.data() + 16 * (size_t)b;
      const int type = (int)o[0], lc = (int)o[1], rc = (int)o[2], fl = (int)o[7];
      next_op = next[b];
      const bool is_break = fl & ofBreak, unary = fl & ofIsUnaryOp, range = fl & ofChildIndexIsRange, lop = fl & ofLeftChildIsOp, rop = fl & ofRightChildIsOp;
      is_right = fl & ofIsRightOp;
      Instr in;
      memset(&in, 0, sizeof in);
      in.p0 = o[4]; in.p1 = o[5];
      if (range) {
        if (lc < 0 || rc >= h->n_prims || lc > rc) return fail(FB_EINVAL, "operator %d: bad primitive range [%d,%d]", b, lc, rc);
        in.kind = 5; in.optype = type; in.a = lc; in.b = rc;
      } else {
        if (!lop) { if (lc < 0 || lc >= h->n_prims) return fail(FB_EINVAL, "operator %d: bad left child %d", b, lc); IL = Val{1, lc}; }
        if (!unary && !rop) { if (rc < 0 || rc >= h->n_prims) return fail(FB_EINVAL, "operator %d: bad right child %d", b, rc); IR = Val{1, rc}; }
        in.kind = 6; in.optype = b;  // sic: ComputeOpField(idxBranchOp, ..), Polygonizer.cl:825
        in.a = ref(IL); in.b = ref(IR);
      }
      in.dst = new_slot();
      h->prog.push_back(in);
      field = Val{2, in.dst};
      if (is_right) IR = field; else IL = field;
      if (is_break) break;
      b = next_op;
    }
    if (is_right) OR = field; else OL = field;
    IL = OL; IR = OR;  // the by-value copies of the finished branch are gone
    op = next_op;
  }
  h->depth = std::max(1, (int)used.size());
  return FB_OK;
}

int compile_tree(fb_poly_s* h) {
  if (h->sem == SEM_OPENCL) return compile_tree_cl(h);
  h->prog.clear();
  h->depth = 1;
  for (int i = 0; i < h->n_prims; i++) {  // validates every instance chain, used or not
    bool yes = false;
    TreeCompiler probe{h};
    FB_TRY(probe.resolves_to_op(i, &yes));
    if (yes && h->n_ops == 0) return fail(FB_EINVAL, "primitive %d instances an operator but the tree has none", i);
  }
  if (h->n_ops == 0) return FB_OK;
  TreeCompiler c{h};
  FB_TRY(c.subtree(0));
  h->prog = std::move(c.prog);
  h->depth = c.depth;
  if (h->depth > 60) return fail(FB_EINVAL, "BlobTree needs %d value slots per point (limit 60)", h->depth);
  return FB_OK;
}

// World-space box outside which primitive i contributes exactly 0.0f (see SegBox).  Local support = skeleton grown by the
// Wyvill radius 1 (field = max(0, (1 - d^2)^3)); mapped with the forward matrix (inverse of the stored matrix node) and
// padded generously against fp32 rounding of the distance and of the affine maps.  Anything not provably bounded
// (infinite line, instances, non-unit directions, singular matrices) gets an infinite box = is never culled.
void support_boxes(fb_poly_s* h) {
  const float inf = FLT_MAX;
  h->pbox.assign(6 * (size_t)h->n_prims, 0.0f);
  for (int i = 0; i < h->n_prims; i++) {
    const float* P = h->prims.data() + 20 * (size_t)i;
    const int type = (int)P[0], im = (int)P[1];
    const double pos[3] = {P[4], P[5], P[6]}, dir[3] = {P[8], P[9], P[10]};
    const double r0 = P[12], r1 = P[13];
    double lo[3], hi[3];
    bool bounded = true, empty = false;
    const double dlen = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const bool unit = std::fabs(dlen - 1.0) < 1e-3;
    auto ball = [&](double rad) { for (int a = 0; a < 3; a++) { lo[a] = pos[a] - rad; hi[a] = pos[a] + rad; } };
    switch (type) {
      case primPoint: ball(1.0); break;
      case primCylinder:
        if (!unit || !(r0 >= 0) || !(r1 >= 0)) { bounded = false; break; }
        for (int a = 0; a < 3; a++) {
          const double e0 = pos[a] - dir[a], e1 = pos[a] + dir[a] * (r1 + 1.0);
          lo[a] = std::min(e0, e1) - (r0 + 1.0); hi[a] = std::max(e0, e1) + (r0 + 1.0);
        }
        break;
      case primDisc: case primRing:
        if (!unit || !(r0 >= 0)) { bounded = false; break; }
        ball(r0 + 1.0);
        break;
      case primCube:
        if (!(r0 >= 0)) { bounded = false; break; }
        ball(r0 + 1.0);
        break;
      case primQuadricPoint:
        if (h->sem == SEM_OPENCL) { ball(std::max(1.0, std::sqrt(std::max(0.0, (double)P[10])))); break; }  // Wyvill value outside the radius
        if (!(P[10] >= 0)) { empty = true; break; }
        ball(std::sqrt((double)P[10]));
        break;
      case primTriangle: case primNULL: empty = true; break;  // wyvill(FLT_MAX) = wyvill(10) = 0
      case primInstance:
        if (h->sem == SEM_OPENCL) empty = true; else bounded = false;  // the kernel returns 0 for instanced nodes
        break;
      case primLine: bounded = false; break;
      default: empty = true; break;                           // unknown types evaluate to 0
    }
    float* out = h->pbox.data() + 6 * (size_t)i;
    if (empty) { out[0] = out[1] = out[2] = inf; out[3] = out[4] = out[5] = -inf; continue; }
    if (bounded && im != 0) {  // forward map of the 8 corners
      const float* m = h->mtx.data() + 12 * (size_t)im;
      const double A[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]}, t[3] = {m[3], m[7], m[11]};
      const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
      if (!(std::fabs(det) > 1e-12) || !std::isfinite(det)) bounded = false;
      else {
        double I[9];
        I[0] = (A[4] * A[8] - A[5] * A[7]) / det; I[1] = (A[2] * A[7] - A[1] * A[8]) / det; I[2] = (A[1] * A[5] - A[2] * A[4]) / det;
        I[3] = (A[5] * A[6] - A[3] * A[8]) / det; I[4] = (A[0] * A[8] - A[2] * A[6]) / det; I[5] = (A[2] * A[3] - A[0] * A[5]) / det;
        I[6] = (A[3] * A[7] - A[4] * A[6]) / det; I[7] = (A[1] * A[6] - A[0] * A[7]) / det; I[8] = (A[0] * A[4] - A[1] * A[3]) / det;
        double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300};
        for (int c = 0; c < 8; c++) {
          const double q[3] = {((c & 1) ? hi[0] : lo[0]) - t[0], ((c & 2) ? hi[1] : lo[1]) - t[1], ((c & 4) ? hi[2] : lo[2]) - t[2]};
          for (int a = 0; a < 3; a++) {
            const double w = I[3 * a] * q[0] + I[3 * a + 1] * q[1] + I[3 * a + 2] * q[2];  // world = A^-1 (local - t)
            wlo[a] = std::min(wlo[a], w); whi[a] = std::max(whi[a], w);
          }
        }
        for (int a = 0; a < 3; a++) { lo[a] = wlo[a]; hi[a] = whi[a]; }
      }
    }
    for (int a = 0; a < 3; a++) {
      if (!bounded || !std::isfinite(lo[a]) || !std::isfinite(hi[a])) { out[a] = -inf; out[3 + a] = inf; continue; }
      const double pad = 1e-2 + 1e-4 * std::max(std::fabs(lo[a]), std::fabs(hi[a]));
      out[a] = (float)(lo[a] - pad); out[3 + a] = (float)(hi[a] + pad);
    }
  }
}

// launches KERNEL<SEM> for the handle's field semantics
#define FB_LAUNCH_SEM(h, KERNEL, grid, block, shmem, ...)                                                                        \
  do {                                                                                                                           \
    switch ((h)->sem) {                                                                                                          \
      case SEM_CPU_BOX: hipLaunchKernelGGL((KERNEL<SEM_CPU_BOX>), grid, block, shmem, (h)->stream, __VA_ARGS__); break;           \
      case SEM_OPENCL: hipLaunchKernelGGL((KERNEL<SEM_OPENCL>), grid, block, shmem, (h)->stream, __VA_ARGS__); break;             \
      default: hipLaunchKernelGGL((KERNEL<SEM_CPU>), grid, block, shmem, (h)->stream, __VA_ARGS__); break;                        \
    }                                                                                                                            \
  } while (0)

size_t stack_bytes(const fb_poly_s* h) { return (size_t)std::max(1, h->depth) * kPB * sizeof(float); }

// the float4 grid of the API from the stored field values, once per sweep
int materialize_grid(fb_poly_s* h) {
  if (h->grid_current) return FB_OK;
  const Grid& G = h->G;
  FB_TRY(h->grid.alloc((size_t)G.n_points));
  hipLaunchKernelGGL(k_grid_xyzf, dim3((unsigned)((G.n_points + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, G, fast_div((unsigned int)G.g[0] * (unsigned int)G.g[1]),
                     fast_div((unsigned int)G.g[0]), h->fval.p, h->grid.p);
  FB_HIP(hipGetLastError());
  h->grid_current = true;
  return FB_OK;
}

int do_sweep(fb_poly_s* h, bool store_grid) {
  const Grid& G = h->G;
  h->grid_current = false;
  const int blocks = (int)((G.n_points + (long long)kPB * kSweepPts - 1) / ((long long)kPB * kSweepPts));
  // with one or two primitives the box test costs more than it can save (sphere at 256^3: 66 vs 50 us)
#define FB_SWEEP(CULL, SEM)                                                                                                                             \
  hipLaunchKernelGGL((k_sweep<CULL, SEM>), dim3(blocks), dim3(kPB), stack_bytes(h), h->stream, G, fast_div((unsigned int)G.g[0] * (unsigned int)G.g[1]),                  \
                     fast_div((unsigned int)G.g[0]), h->d_prog.p, (int)h->prog.size(), h->n_prims, h->depth, \
                     h->d_prims.p, h->d_mtx.p, h->d_pbox.p, h->d_cbox.p, store_grid ? h->fval.p : nullptr, h->inside.p)
  const bool cull = h->n_prims > 2;
  switch (h->sem) {
    case SEM_CPU_BOX: if (cull) FB_SWEEP(true, SEM_CPU_BOX); else FB_SWEEP(false, SEM_CPU_BOX); break;
    case SEM_OPENCL: if (cull) FB_SWEEP(true, SEM_OPENCL); else FB_SWEEP(false, SEM_OPENCL); break;
    default: if (cull) FB_SWEEP(true, SEM_CPU); else FB_SWEEP(false, SEM_CPU); break;
  }
#undef FB_SWEEP
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int set_grid(fb_poly_s* h, const float lo[3], float cellsize, const int dims[3], int z0 = 0, int gz_total = 0) {
  if (!(cellsize > 0)) return fail(FB_EINVAL, "cellsize must be positive");
  Grid G;
  G.cellsize = cellsize;
  G.z0 = z0;
  h->gz_total = gz_total > 0 ? gz_total : dims[2];
  G.n_points = 1; G.n_cells = 1;
  for (int a = 0; a < 3; a++) {
    if (dims[a] < 2) return fail(FB_EINVAL, "grid needs at least 2 points per axis");
    G.lo[a] = lo[a]; G.g[a] = dims[a]; G.c[a] = dims[a] - 1;
    G.n_points *= dims[a]; G.n_cells *= (dims[a] - 1);
  }
  if (G.n_points >= (1LL << 31)) return fail(FB_EINVAL, "grid too large");
  h->G = G;
  const size_t pw = (size_t)((G.n_points + 63) / 64);
  FB_TRY(h->fval.alloc((size_t)G.n_points));
  h->grid_current = false;
  DevBuf<unsigned long long>* masks[] = {&h->inside, &h->cinc, &h->vinc, &h->lastx, &h->lasty, &h->lastz, &h->valid, &h->crossx, &h->crossy, &h->crossz,
                                         &h->surf, &h->firstx, &h->firsty, &h->firstz};
  for (auto* m : masks) FB_TRY(m->alloc(pw + 1));
  h->n_words = (long long)pw;
  FB_TRY(h->vbase.alloc(pw));
  FB_TRY(h->cbase.alloc(pw));
  FB_TRY(h->aux_pop.alloc(pw));
  const size_t pch = (pw + kChunk - 1) / kChunk;
  if (pch > 16 * kPB) return fail(FB_EINVAL, "grid too large for the chunked scan");
  FB_TRY(h->vsum.alloc(pch));  // (sized per chunk: the surface pass's scans and the emission grids read its length)
  FB_TRY(h->block_sums.alloc((pw + kPB - 1) / kPB));
  DevBuf<unsigned int>* words[] = {&h->edge_pop, &h->idx_pop, &h->ebase, &h->ibase};
  for (auto* m : words) FB_TRY(m->alloc(pw));
  FB_TRY(h->esum.alloc(pch));
  FB_TRY(h->isum.alloc(pch));
  FB_TRY(h->config.alloc((size_t)G.n_cells));
  FB_TRY(h->flags.alloc((size_t)G.n_points));
  FB_TRY(h->totals.alloc(6));
  hipLaunchKernelGGL(k_grid_masks, dim3((int)((G.n_points + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, G, h->lastx.p, h->lasty.p, h->lastz.p, h->valid.p,
                     h->firstx.p, h->firsty.p, h->firstz.p);
  FB_HIP(hipGetLastError());
  h->materialized = false;
  h->have_grid = h->classified = h->tetra = h->surfaced = false;
  return FB_OK;
}

int do_classify(fb_poly_s* h) {
  const Grid& G = h->G;
  const long long pw = h->n_words;
  const int nblk = (int)((pw + kPB - 1) / kPB);
  const bool no_rows64 = getenv("FEMBRAIN_CLASSIFY_MASKS") && atoi(getenv("FEMBRAIN_CLASSIFY_MASKS")) != 0;  // development aid: always load the masks
  if (G.g[0] % 64 == 0 && pw < (1LL << 31) && !no_rows64) {
    const unsigned int wpr = (unsigned int)(G.g[0] / 64);
    hipLaunchKernelGGL(k_classify<true>, dim3(nblk), dim3(kPB), 0, h->stream, G, fast_div(wpr * (unsigned int)G.g[1]), fast_div(wpr), pw, h->inside.p, h->lastx.p, h->lasty.p,
                       h->lastz.p, h->firstx.p, h->firsty.p, h->firstz.p, h->valid.p, h->cinc.p, h->crossx.p, h->crossy.p, h->crossz.p, h->vinc.p, h->aux_pop.p, h->block_sums.p);
  } else {
    hipLaunchKernelGGL(k_classify<false>, dim3(nblk), dim3(kPB), 0, h->stream, G, fast_div(1u), fast_div(1u), pw, h->inside.p, h->lastx.p, h->lasty.p, h->lastz.p, h->firstx.p,
                       h->firsty.p, h->firstz.p, h->valid.p, h->cinc.p, h->crossx.p, h->crossy.p, h->crossz.p, h->vinc.p, h->aux_pop.p, h->block_sums.p);
  }
  hipLaunchKernelGGL(k_ranks, dim3(nblk), dim3(kPB), 0, h->stream, pw, h->cinc.p, h->vinc.p, h->block_sums.p, h->cbase.p, h->vbase.p, h->totals.p);
  FB_HIP(hipGetLastError());
  h->materialized = false;
  return FB_OK;
}

int fetch_counts(fb_poly_s* h) {
  unsigned int t[4];
  FB_TRY(h->totals.download(t, 4, h->stream));
  fb_poly_counts& c = h->counts;
  memset(&c, 0, sizeof c);
  for (int a = 0; a < 3; a++) c.grid[a] = h->G.g[a];
  c.n_points = (int)h->G.n_points; c.n_cells = (int)h->G.n_cells;
  c.n_crossed_edges = (int)t[0]; c.n_surface_cells = (int)t[1];
  c.n_included_cells = (int)t[2]; c.n_tet_vertices = (int)t[3];
  c.n_tets = 6 * c.n_included_cells;
  return FB_OK;
}

int do_emit(fb_poly_s* h) {
  const Grid& G = h->G;
  // each wave owns a run of mask words: 4 words per wave measured best at 256^3 (16,384 blocks: 235 us pipeline; 16 words 249 us,
  // 1 word 323 us) -- the loaded waves sit in the part of the grid the surface encloses, shorter runs spread them over more CUs
  static const int emit_blocks = getenv("FB_EMIT_BLOCKS") ? atoi(getenv("FB_EMIT_BLOCKS")) : 0;  // tuning knob (development)
  const long long words = (G.n_points + 63) / 64;
  const int pb = (int)std::min<long long>((G.n_points + kPB - 1) / kPB, emit_blocks > 0 ? emit_blocks : std::max<long long>(1, (words + 15) / 16));
  hipLaunchKernelGGL(k_tet_vertices, dim3((unsigned)((((G.n_points + 63) >> 6) + kVW - 1) / kVW)), dim3(kPB), 0, h->stream, G, fast_div((unsigned int)G.g[0] * (unsigned int)G.g[1]), fast_div((unsigned int)G.g[0]), h->vinc.p, h->vbase.p, h->tv.p);
  FB_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_tet_elements, dim3(pb), dim3(kPB), 0, h->stream, G, h->cinc.p, h->cbase.p, h->vinc.p, h->vbase.p, h->tt.p);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

// scans + emission of the marching-cubes surface; totals[4] = surface vertices, totals[5] = triangle indices
int do_surface_counts(fb_poly_s* h) {
  const Grid& G = h->G;
  const long long pw = h->n_words;
  const int wb = (int)((pw + kPB - 1) / kPB), pch = (int)h->vsum.n;
  hipLaunchKernelGGL(k_surface_pops, dim3(wb), dim3(kPB), 0, h->stream, G, pw, h->inside.p, h->lastx.p, h->lasty.p, h->lastz.p, h->valid.p, h->aux_pop.p,
                     h->d_nvert.p, h->surf.p, h->edge_pop.p, h->idx_pop.p);
  ScanJobs jobs;
  jobs.j[0] = ScanJob{h->edge_pop.p, nullptr, h->esum.p, nullptr, h->ebase.p, h->totals.p + 4, nullptr};
  jobs.j[1] = ScanJob{h->idx_pop.p, nullptr, h->isum.p, nullptr, h->ibase.p, h->totals.p + 5, nullptr};
  hipLaunchKernelGGL(k_chunk_sums, dim3(pch, 2), dim3(kPB), 0, h->stream, jobs, (int)pw);
  hipLaunchKernelGGL(k_scan_chunks, dim3(2), dim3(kPB), 0, h->stream, jobs, pch);
  hipLaunchKernelGGL(k_chunk_scan, dim3(pch, 2), dim3(kPB), 0, h->stream, jobs, (int)pw);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int do_surface_emit(fb_poly_s* h) {
  const Grid& G = h->G;
  const long long pw = h->n_words;
  const long long nv = h->counts.n_surface_vertices, ntri = h->counts.n_surface_indices / 3;
  if (nv == 0 && ntri == 0) return FB_OK;
  hipLaunchKernelGGL(k_surface_lists, dim3((int)((pw + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, G, pw, h->inside.p, h->surf.p, h->crossx.p, h->crossy.p,
                     h->crossz.p, h->ebase.p, h->ibase.p, h->d_nvert.p, h->elist.p, h->tlist.p);
  FB_HIP(hipGetLastError());
  if (nv > 0) {
    FB_LAUNCH_SEM(h, k_surface_vertices, dim3((int)((nv + kPB - 1) / kPB)), dim3(kPB), stack_bytes(h), G, nv, h->elist.p, h->d_prog.p,
                  (int)h->prog.size(), h->n_prims, h->d_prims.p, h->d_mtx.p, h->d_cbox.p, h->fval.p, h->vinc.p, h->vbase.p, h->sv.p, h->sn.p, h->sends.p, h->sfrac.p);
    FB_HIP(hipGetLastError());
  }
  if (ntri > 0) {
    hipLaunchKernelGGL(k_surface_elements, dim3((int)((ntri + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, G, ntri, h->tlist.p, h->crossx.p, h->crossy.p,
                       h->crossz.p, h->ebase.p, h->d_tri.p, h->d_edge_info.p, h->si.p);
    FB_HIP(hipGetLastError());
  }
  return FB_OK;
}

}  // namespace

extern "C" {

int fb_poly_create(fb_poly_t* out, int device, const float* header12, int n_ops, const float* ops16, int n_prims, const float* prims20,
                   int n_mtx, const float* mtx12) {
  if (!out || !header12 || n_prims < 1 || !prims20 || n_ops < 0 || (n_ops > 0 && !ops16) || n_mtx < 1 || !mtx12)
    return fail(FB_EINVAL, "bad BlobTree arrays (need >= 1 primitive and the identity matrix node)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FB_EDEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(FB_EINVAL, "device %d out of range", device);
  FB_HIP(hipSetDevice(device));
  fb_poly_s* h = new fb_poly_s;
  h->device = device;
  h->header.assign(header12, header12 + 12);
  h->ops.assign(ops16, ops16 + 16 * (size_t)n_ops);
  h->prims.assign(prims20, prims20 + 20 * (size_t)n_prims);
  h->mtx.assign(mtx12, mtx12 + 12 * (size_t)n_mtx);
  h->n_ops = n_ops; h->n_prims = n_prims; h->n_mtx = n_mtx;
  int rc = FB_OK;
  for (int i = 0; i < n_prims && rc == FB_OK; i++) {
    const int im = (int)h->prims[20 * (size_t)i + 1];
    if (im < 0 || im >= n_mtx) rc = fail(FB_EINVAL, "primitive %d references matrix %d of %d", i, im, n_mtx);
  }
  if (rc == FB_OK) rc = compile_tree(h);
  if (rc == FB_OK && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(FB_EDEVICE, "hipStreamCreate failed");
  for (auto& e : h->ev)
    if (rc == FB_OK && hipEventCreate(&e) != hipSuccess) rc = fail(FB_EDEVICE, "hipEventCreate failed");
  if (rc == FB_OK) {
    std::vector<Instr> prog = h->prog;
    if (prog.empty()) prog.resize(1);  // keep a valid pointer
    rc = h->d_prog.upload(prog, h->stream);
  }
  if (rc == FB_OK) rc = h->d_prims.upload(h->prims, h->stream);
  if (rc == FB_OK) rc = h->d_mtx.upload(h->mtx, h->stream);
  if (rc == FB_OK) {
    support_boxes(h);
    rc = h->d_pbox.upload(h->pbox, h->stream);
  }
  const CubeTable& ct = cube_table();
  if (rc == FB_OK) rc = h->d_tri.upload(&ct.tri[0][0], sizeof ct.tri, h->stream);
  if (rc == FB_OK) rc = h->d_nvert.upload(ct.nvert, sizeof ct.nvert, h->stream);
  if (rc == FB_OK) rc = h->d_edge_info.upload(ct.edge_info, sizeof ct.edge_info, h->stream);
  if (rc != FB_OK) {
    std::string keep = last_error();
    fb_poly_destroy(h);
    last_error() = keep;
    return rc;
  }
  *out = h;
  return FB_OK;
}

int fb_poly_compile_info(int n_ops, const float* ops16, int n_prims, const float* prims20, int* n_steps, int* n_slots) {
  if (n_prims < 1 || !prims20 || n_ops < 0 || (n_ops > 0 && !ops16)) return fail(FB_EINVAL, "bad BlobTree arrays");
  fb_poly_s h;
  h.ops.assign(ops16, ops16 + 16 * (size_t)n_ops);
  h.prims.assign(prims20, prims20 + 20 * (size_t)n_prims);
  h.n_ops = n_ops; h.n_prims = n_prims;
  FB_TRY(compile_tree(&h));
  if (n_steps) *n_steps = (int)h.prog.size();
  if (n_slots) *n_slots = h.depth;
  return FB_OK;
}

int fb_poly_set_field_semantics(fb_poly_t h, int semantics, const float* prim_boxes6) {
  CHECK_POLY(h);
  if (semantics != SEM_CPU && semantics != SEM_CPU_BOX && semantics != SEM_OPENCL) return fail(FB_EINVAL, "unknown field semantics %d", semantics);
  if (semantics == SEM_CPU_BOX && !prim_boxes6) return fail(FB_EINVAL, "FB_FIELD_CPU_BOX needs the primitive boxes (6 floats per primitive)");
  FB_HIP(hipStreamSynchronize(h->stream));
  const int old = h->sem;
  const std::vector<Instr> old_prog = h->prog;
  const int old_depth = h->depth;
  h->sem = semantics;
  int rc = compile_tree(h);
  if (rc != FB_OK) {  // the handle keeps working with the semantics it had
    h->sem = old; h->prog = old_prog; h->depth = old_depth;
    return rc;
  }
  std::vector<Instr> prog = h->prog;
  if (prog.empty()) prog.resize(1);
  FB_TRY(h->d_prog.upload(prog, h->stream));
  h->cbox.clear();
  if (semantics == SEM_CPU_BOX) {
    h->cbox.assign(prim_boxes6, prim_boxes6 + 6 * (size_t)h->n_prims);
    FB_TRY(h->d_cbox.upload(h->cbox, h->stream));
  }
  support_boxes(h);
  FB_TRY(h->d_pbox.upload(h->pbox, h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  h->have_grid = h->classified = h->tetra = h->surfaced = false;  // results of the old semantics are void
  return FB_OK;
}

int fb_poly_field_semantics(fb_poly_t h) { return h ? h->sem : FB_EINVAL; }

int fb_poly_destroy(fb_poly_t h) {
  if (!h) return FB_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  hipStream_t s = h->stream;
  delete h;
  if (s) (void)hipStreamDestroy(s);
  return FB_OK;
}

int fb_poly_field_array(fb_poly_t h, int n, float* xyzf) {
  CHECK_POLY(h);
  if (n < 0 || (n > 0 && !xyzf)) return fail(FB_EINVAL, "bad point array");  // PS::SKETCH error codes, OclPolygonizer.h:26-33
  if (n == 0) return FB_OK;
  DevBuf<float4> pts;
  FB_TRY(pts.upload((const float4*)xyzf, (size_t)n, h->stream));
  FB_LAUNCH_SEM(h, k_field_array, dim3(ceil_div(n, kPB)), dim3(kPB), stack_bytes(h), n, h->d_prog.p, (int)h->prog.size(), h->n_prims, h->d_prims.p,
                h->d_mtx.p, h->d_cbox.p, pts.p);
  FB_HIP(hipGetLastError());
  return pts.download((float4*)xyzf, (size_t)n, h->stream);
}

int fb_poly_sweep_grid(fb_poly_t h, const float lower[3], float cellsize, const int dims[3]) {
  CHECK_POLY(h);
  if (!lower || !dims) return fail(FB_EINVAL, "null grid description");
  FB_TRY(set_grid(h, lower, cellsize, dims));
  FB_TRY(do_sweep(h, true));
  FB_HIP(hipStreamSynchronize(h->stream));
  h->have_grid = true;
  return FB_OK;
}

int fb_poly_sweep_slab(fb_poly_t h, const float lower[3], float cellsize, const int dims[3], int z_first, int z_count) {
  CHECK_POLY(h);
  if (!lower || !dims) return fail(FB_EINVAL, "null grid description");
  if (z_first < 0 || z_count < 2 || z_first + z_count > dims[2]) return fail(FB_EINVAL, "slab planes [%d, %d) do not fit a grid of %d planes", z_first, z_first + z_count, dims[2]);
  const int sub[3] = {dims[0], dims[1], z_count};
  FB_TRY(set_grid(h, lower, cellsize, sub, z_first, dims[2]));
  FB_TRY(do_sweep(h, true));
  FB_HIP(hipStreamSynchronize(h->stream));
  h->have_grid = true;
  return FB_OK;
}

namespace {
// number of set bits of a scanned mask below bit `bit` (base[w] = set bits before word w)
int count_before(fb_poly_s* h, const DevBuf<unsigned long long>& mask, const DevBuf<unsigned int>& base, long long bit, unsigned int total, unsigned int* out) {
  if (bit >= h->G.n_points) { *out = total; return FB_OK; }
  unsigned long long w = 0;
  unsigned int b = 0;
  FB_TRY(mask.download(&w, 1, h->stream, (size_t)(bit >> 6)));
  FB_TRY(base.download(&b, 1, h->stream, (size_t)(bit >> 6)));
  *out = b + (unsigned int)__builtin_popcountll(w & ((1ULL << (bit & 63)) - 1ULL));
  return FB_OK;
}

struct SlabRange { unsigned int v0, v1, c0, c1; };  // owned vertices [v0, v1) and included cells [c0, c1) in the slab's own numbering

int slab_range(fb_poly_s* h, int own_first_plane, int own_planes, int own_layers, SlabRange* r) {
  if (!h->tetra) return fail(FB_EINVAL, "tetrahedralize first");
  const Grid& G = h->G;
  const int p0 = own_first_plane - G.z0, p1 = p0 + own_planes, l1 = p0 + own_layers;
  if (own_planes < 0 || own_layers < 0 || p0 < 0 || p1 > G.g[2] || l1 > G.c[2])
    return fail(FB_EINVAL, "planes [%d, %d) / layers [%d, %d) are not inside the slab [%d, %d)", own_first_plane, own_first_plane + own_planes,
                own_first_plane, own_first_plane + own_layers, G.z0, G.z0 + G.g[2]);
  // the vertex marks of a plane are complete only if both cell layers next to it were classified here (or do not exist)
  if (p0 == 0 && G.z0 > 0 && own_planes > 0) return fail(FB_EINVAL, "the slab must start one plane below the first owned plane %d", own_first_plane);
  if (own_layers > 0 && l1 + 1 > G.c[2] && G.z0 + G.g[2] < h->gz_total)
    return fail(FB_EINVAL, "the slab must reach two planes above the last owned cell layer %d", own_first_plane + own_layers - 1);
  const long long gxy = (long long)G.g[0] * G.g[1];
  const unsigned int nv = (unsigned int)h->counts.n_tet_vertices, nc = (unsigned int)(h->counts.n_tets / 6);
  FB_TRY(count_before(h, h->vinc, h->vbase, p0 * gxy, nv, &r->v0));
  FB_TRY(count_before(h, h->vinc, h->vbase, p1 * gxy, nv, &r->v1));
  FB_TRY(count_before(h, h->cinc, h->cbase, p0 * gxy, nc, &r->c0));
  FB_TRY(count_before(h, h->cinc, h->cbase, l1 * gxy, nc, &r->c1));
  return FB_OK;
}

__global__ __launch_bounds__(kPB) void k_renumber_tets(long long n, const uint4* __restrict__ in, unsigned int delta, uint4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * kPB + threadIdx.x;
  if (i >= n) return;
  uint4 t = in[i];
  t.x += delta; t.y += delta; t.z += delta; t.w += delta;
  out[i] = t;
}
}  // namespace

int fb_poly_slab_counts(fb_poly_t h, int own_first_plane, int own_planes, int own_layers, int* n_vertices, int* n_tets) {
  CHECK_POLY(h);
  SlabRange r;
  FB_TRY(slab_range(h, own_first_plane, own_planes, own_layers, &r));
  if (n_vertices) *n_vertices = (int)(r.v1 - r.v0);
  if (n_tets) *n_tets = (int)(6 * (r.c1 - r.c0));
  return FB_OK;
}

int fb_poly_read_tetmesh_slab(fb_poly_t h, int own_first_plane, int own_planes, int own_layers, unsigned int vertex_base, float* xyz, unsigned int* tets) {
  CHECK_POLY(h);
  SlabRange r;
  FB_TRY(slab_range(h, own_first_plane, own_planes, own_layers, &r));
  if (xyz && r.v1 > r.v0) FB_TRY(h->tv.download(xyz, 3 * (size_t)(r.v1 - r.v0), h->stream, 3 * (size_t)r.v0));
  const long long nt = 6LL * (r.c1 - r.c0);
  if (tets && nt > 0) {
    DevBuf<uint4> out;
    FB_TRY(out.alloc((size_t)nt));
    // a vertex's number in the whole grid = its number here - (vertices of the planes below the owned ones) + vertex_base
    hipLaunchKernelGGL(k_renumber_tets, dim3((unsigned)((nt + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, nt, h->tt.p + 6 * (size_t)r.c0, vertex_base - r.v0, out.p);
    FB_HIP(hipGetLastError());
    FB_TRY(out.download((uint4*)tets, (size_t)nt, h->stream));
  }
  return FB_OK;
}

int fb_poly_sweep(fb_poly_t h, float cellsize, int dims_out[3]) {
  CHECK_POLY(h);
  if (cellsize < 0.01f) return fail(FB_EINVAL, "cellsize %g below the reference minimum 0.01 (GPUPoly::run)", cellsize);
  // OclPolygonizer.cpp:1363-1378: cells = ceil(extent / cellsize) + 1 per axis, points = cells + 1, origin = bbox lower
  float lo[3] = {h->header[0], h->header[1], h->header[2]};
  int dims[3];
  for (int a = 0; a < 3; a++) {
    const float extent = h->header[4 + a] - h->header[a];
    dims[a] = (int)ceilf(extent / cellsize) + 2;
  }
  if (dims_out) memcpy(dims_out, dims, sizeof dims);
  return fb_poly_sweep_grid(h, lo, cellsize, dims);
}

int fb_poly_read_grid(fb_poly_t h, float* xyzf) {
  CHECK_POLY(h);
  if (!h->have_grid || !xyzf) return fail(FB_EINVAL, "no swept grid / null buffer");
  FB_TRY(materialize_grid(h));
  return h->grid.download((float4*)xyzf, (size_t)h->G.n_points, h->stream);
}

int fb_poly_classify(fb_poly_t h, fb_poly_counts* counts) {
  CHECK_POLY(h);
  if (!h->have_grid) return fail(FB_EINVAL, "sweep the grid first");
  FB_TRY(do_classify(h));
  FB_TRY(fetch_counts(h));
  h->classified = true;
  h->surfaced = false;
  if (counts) *counts = h->counts;
  return FB_OK;
}

int fb_poly_read_classification(fb_poly_t h, unsigned char* edge_flags, unsigned int* edge_counts, unsigned char* cell_configs) {
  CHECK_POLY(h);
  if (!h->classified) return fail(FB_EINVAL, "classify first");
  const size_t np = (size_t)h->G.n_points;
  if (!h->materialized) {
    hipLaunchKernelGGL(k_materialize, dim3((int)((np + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, h->G, h->inside.p, h->crossx.p, h->crossy.p,
                       h->crossz.p, h->flags.p, h->config.p);
    FB_HIP(hipGetLastError());
    h->materialized = true;
  }
  if (edge_flags || edge_counts) {
    std::vector<unsigned char> f(np);
    FB_TRY(h->flags.download(f.data(), np, h->stream));
    if (edge_flags) memcpy(edge_flags, f.data(), np);
    if (edge_counts)
      for (size_t i = 0; i < np; i++) edge_counts[i] = (unsigned int)__builtin_popcount(f[i]);
  }
  if (cell_configs) FB_TRY(h->config.download(cell_configs, (size_t)h->G.n_cells, h->stream));
  return FB_OK;
}

int fb_poly_tetrahedralize(fb_poly_t h, fb_poly_counts* counts) {
  CHECK_POLY(h);
  if (!h->classified) return fail(FB_EINVAL, "classify first");
  FB_TRY(h->tv.alloc(std::max<size_t>(1, 3 * (size_t)h->counts.n_tet_vertices)));
  FB_TRY(h->tt.alloc(std::max<size_t>(1, (size_t)h->counts.n_tets)));
  FB_TRY(do_emit(h));
  FB_HIP(hipStreamSynchronize(h->stream));
  h->tetra = true;
  if (counts) *counts = h->counts;
  return FB_OK;
}

extern "C++" {
namespace fb {
int poly_device_tetmesh(fb_poly_t h, DeviceTetMesh* out) {
  CHECK_POLY(h);
  if (!h->tetra) return fail(FB_EINVAL, "tetrahedralize first");
  FB_HIP(hipStreamSynchronize(h->stream));
  out->device = h->device; out->n_vertices = h->counts.n_tet_vertices; out->n_tets = h->counts.n_tets;
  out->xyz = h->tv.p; out->tets = h->tt.p;
  return FB_OK;
}
}  // namespace fb
}  // extern "C++"

int fb_poly_read_tetmesh(fb_poly_t h, float* xyz, unsigned int* tets) {
  CHECK_POLY(h);
  if (!h->tetra) return fail(FB_EINVAL, "tetrahedralize first");
  if (xyz) FB_TRY(h->tv.download(xyz, 3 * (size_t)h->counts.n_tet_vertices, h->stream));
  if (tets) FB_TRY(h->tt.download((uint4*)tets, (size_t)h->counts.n_tets, h->stream));
  return FB_OK;
}

int fb_poly_cube_table(unsigned char tri[4096], unsigned char nvert[256]) {
  const CubeTable& ct = cube_table();
  if (tri) memcpy(tri, ct.tri, sizeof ct.tri);
  if (nvert) memcpy(nvert, ct.nvert, sizeof ct.nvert);
  return FB_OK;
}

int fb_poly_surface(fb_poly_t h, fb_poly_counts* counts) {
  CHECK_POLY(h);
  if (!h->classified) return fail(FB_EINVAL, "classify first");
  FB_TRY(do_surface_counts(h));
  unsigned int t[2];
  FB_TRY(h->totals.download(t, 2, h->stream, 4));
  if ((int)t[0] != h->counts.n_crossed_edges) return fail(FB_EDEVICE, "surface vertex scan (%u) disagrees with the edge table (%d)", t[0], h->counts.n_crossed_edges);
  h->counts.n_surface_vertices = (int)t[0];
  h->counts.n_surface_indices = (int)t[1];
  FB_TRY(h->sv.alloc(std::max<size_t>(1, 3 * (size_t)t[0])));
  FB_TRY(h->sn.alloc(std::max<size_t>(1, 3 * (size_t)t[0])));
  FB_TRY(h->si.alloc(std::max<size_t>(1, (size_t)t[1])));
  FB_TRY(h->sends.alloc(std::max<size_t>(1, (size_t)t[0])));
  FB_TRY(h->sfrac.alloc(std::max<size_t>(1, (size_t)t[0])));
  FB_TRY(h->elist.alloc(std::max<size_t>(1, (size_t)t[0])));
  FB_TRY(h->tlist.alloc(std::max<size_t>(1, (size_t)t[1] / 3)));
  FB_TRY(do_surface_emit(h));
  FB_HIP(hipStreamSynchronize(h->stream));
  h->surfaced = true;
  if (counts) *counts = h->counts;
  return FB_OK;
}

int fb_poly_read_surface(fb_poly_t h, float* xyz, float* normals, unsigned int* indices) {
  CHECK_POLY(h);
  if (!h->surfaced) return fail(FB_EINVAL, "run fb_poly_surface first");
  const size_t nv = 3 * (size_t)h->counts.n_surface_vertices;
  if (xyz) FB_TRY(h->sv.download(xyz, nv, h->stream));
  if (normals) FB_TRY(h->sn.download(normals, nv, h->stream));
  if (indices) FB_TRY(h->si.download(indices, (size_t)h->counts.n_surface_indices, h->stream));
  return FB_OK;
}

int fb_poly_apply_displacements(fb_poly_t h, int mesh, int n_dof, const double* displacements, float* xyz_out) {
  CHECK_POLY(h);
  const bool tet = mesh == FB_MESH_TET;
  if (mesh != FB_MESH_TET && mesh != FB_MESH_SURFACE) return fail(FB_EINVAL, "mesh must be FB_MESH_SURFACE or FB_MESH_TET");
  if (tet ? !h->tetra : !h->surfaced) return fail(FB_EINVAL, "no valid vertex buffer: polygonize first");  // m_isValidVertex
  const long long n = 3LL * (tet ? h->counts.n_tet_vertices : h->counts.n_surface_vertices);
  if (n_dof != n || !displacements) return fail(FB_EINVAL, "displacement vector has %d entries, the mesh has %lld", n_dof, n);
  if (n == 0) return FB_OK;
  FB_TRY(h->disp.upload(displacements, (size_t)n, h->stream));
  FB_TRY(h->deformed.alloc((size_t)n));
  hipLaunchKernelGGL(k_apply_displacements, dim3((int)((n + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, n, tet ? h->tv.p : h->sv.p, h->disp.p, h->deformed.p);
  FB_HIP(hipGetLastError());
  if (xyz_out) return h->deformed.download(xyz_out, (size_t)n, h->stream);
  FB_HIP(hipStreamSynchronize(h->stream));
  return FB_OK;
}

int fb_poly_read_surface_colors(fb_poly_t h, float* rgba) {
  CHECK_POLY(h);
  if (!h->surfaced) return fail(FB_EINVAL, "run fb_poly_surface first");
  if (!rgba) return fail(FB_EINVAL, "null output");
  if (h->sem != SEM_CPU) return fail(FB_EINVAL, "the colour pass is built for FB_FIELD_CPU semantics only");
  const long long nv = h->counts.n_surface_vertices;
  if (nv == 0) return FB_OK;
  const long long kChunk = 1 << 16;
  DevBuf<float> scratch;
  DevBuf<float4> out;
  FB_TRY(scratch.alloc((size_t)5 * std::max(1, h->depth) * (size_t)std::min(nv, kChunk)));
  FB_TRY(out.alloc((size_t)nv));
  for (long long first = 0; first < nv; first += kChunk) {
    const long long n = std::min(kChunk, nv - first), nth = std::min(nv, kChunk);
    hipLaunchKernelGGL(k_surface_colors, dim3((int)((n + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, first, n, h->sv.p, h->d_prog.p, (int)h->prog.size(),
                       h->n_prims, h->d_prims.p, h->d_mtx.p, scratch.p, nth, out.p);
    FB_HIP(hipGetLastError());
  }
  return out.download((float4*)rgba, (size_t)nv, h->stream);
}

int fb_poly_field_color_array(fb_poly_t h, int n, float* xyzf, float* rgb) {
  CHECK_POLY(h);
  if (n < 0 || (n > 0 && (!xyzf || !rgb))) return fail(FB_EINVAL, "bad point array");
  if (h->sem != SEM_CPU) return fail(FB_EINVAL, "the colour pass is built for FB_FIELD_CPU semantics only");
  if (n == 0) return FB_OK;
  DevBuf<float4> pts;
  DevBuf<float> scratch, col;
  FB_TRY(pts.upload((const float4*)xyzf, (size_t)n, h->stream));
  FB_TRY(scratch.alloc((size_t)5 * std::max(1, h->depth) * (size_t)n));
  FB_TRY(col.alloc((size_t)3 * n));
  hipLaunchKernelGGL(k_field_color_array, dim3(ceil_div(n, kPB)), dim3(kPB), 0, h->stream, (long long)n, pts.p, h->d_prog.p, (int)h->prog.size(), h->n_prims,
                     h->d_prims.p, h->d_mtx.p, scratch.p, (long long)n, col.p);
  FB_HIP(hipGetLastError());
  FB_TRY(pts.download((float4*)xyzf, (size_t)n, h->stream));
  return col.download(rgb, (size_t)3 * n, h->stream);
}

int fb_poly_off_surface(fb_poly_t h, float len, float* xyzf_pairs) {
  CHECK_POLY(h);
  if (!h->surfaced) return fail(FB_EINVAL, "run fb_poly_surface first");
  if (!xyzf_pairs) return fail(FB_EINVAL, "null output");
  const long long nv = h->counts.n_surface_vertices;
  if (nv == 0) return FB_OK;
  DevBuf<float4> out;
  FB_TRY(out.alloc((size_t)(2 * nv)));
  FB_LAUNCH_SEM(h, k_off_surface, dim3((int)((nv + kPB - 1) / kPB)), dim3(kPB), stack_bytes(h), nv, len, h->sv.p, h->sn.p, h->d_prog.p,
                (int)h->prog.size(), h->n_prims, h->d_prims.p, h->d_mtx.p, h->d_cbox.p, out.p);
  FB_HIP(hipGetLastError());
  return out.download((float4*)xyzf_pairs, (size_t)(2 * nv), h->stream);
}

int fb_poly_time_surface(fb_poly_t h, int reps, double* seconds) {
  CHECK_POLY(h);
  if (!h->surfaced || reps < 1 || !seconds) return fail(FB_EINVAL, "run fb_poly_surface once first");
  FB_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; r++) {
    FB_TRY(do_surface_counts(h));
    FB_TRY(do_surface_emit(h));  // same counts as the validated run: grid and tree are unchanged
  }
  FB_HIP(hipEventRecord(h->ev[1], h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  float ms = 0;
  FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *seconds = ms * 1e-3 / reps;
  return FB_OK;
}

int fb_poly_read_surface_binding(fb_poly_t h, unsigned int* tet_vertex_pairs, float* weights) {
  CHECK_POLY(h);
  if (!h->surfaced) return fail(FB_EINVAL, "run fb_poly_surface first");
  const size_t nv = (size_t)h->counts.n_surface_vertices;
  if (tet_vertex_pairs) FB_TRY(h->sends.download((uint2*)tet_vertex_pairs, nv, h->stream));
  if (weights) FB_TRY(h->sfrac.download(weights, nv, h->stream));
  return FB_OK;
}

int fb_poly_interpolate_displacements(fb_poly_t h, int n_tet_dof, const double* tet_displacements, float* xyz_out) {
  CHECK_POLY(h);
  if (!h->surfaced) return fail(FB_EINVAL, "run fb_poly_surface first");
  if (n_tet_dof != 3 * h->counts.n_tet_vertices || !tet_displacements)
    return fail(FB_EINVAL, "displacement vector has %d entries, the tet mesh of this grid has %d DOF", n_tet_dof, 3 * h->counts.n_tet_vertices);
  const long long nv = h->counts.n_surface_vertices;
  if (nv == 0) return FB_OK;
  FB_TRY(h->disp.upload(tet_displacements, (size_t)n_tet_dof, h->stream));
  FB_TRY(h->deformed.alloc((size_t)(3 * nv)));
  hipLaunchKernelGGL(k_interpolate_displacements, dim3((int)((3 * nv + kPB - 1) / kPB)), dim3(kPB), 0, h->stream, nv, h->sv.p, h->sends.p, h->sfrac.p, h->disp.p,
                     h->deformed.p);
  FB_HIP(hipGetLastError());
  if (xyz_out) return h->deformed.download(xyz_out, (size_t)(3 * nv), h->stream);
  FB_HIP(hipStreamSynchronize(h->stream));
  return FB_OK;
}

int fb_poly_time_pipeline(fb_poly_t h, int reps, double* sweep_seconds, double* pipeline_seconds) {
  CHECK_POLY(h);
  if (!h->tetra || reps < 1) return fail(FB_EINVAL, "run sweep, classify and tetrahedralize once first");
  float ms_s = 0, ms_p = 0;
  for (int r = 0; r < reps; r++) {
    FB_HIP(hipEventRecord(h->ev[0], h->stream));
    FB_TRY(do_sweep(h, true));
    FB_HIP(hipEventRecord(h->ev[1], h->stream));
    FB_TRY(do_classify(h));
    FB_TRY(do_emit(h));  // same counts as the validated run: the grid and tree are unchanged
    FB_HIP(hipEventRecord(h->ev[2], h->stream));
    FB_HIP(hipStreamSynchronize(h->stream));
    float a = 0, b = 0;
    FB_HIP(hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
    FB_HIP(hipEventElapsedTime(&b, h->ev[0], h->ev[2]));
    ms_s += a; ms_p += b;
  }
  if (sweep_seconds) *sweep_seconds = ms_s * 1e-3 / reps;
  if (pipeline_seconds) *pipeline_seconds = ms_p * 1e-3 / reps;
  return FB_OK;
}

int fb_poly_time_grid(fb_poly_t h, int reps, double* sweep_and_grid_seconds) {
  CHECK_POLY(h);
  if (!h->have_grid || reps < 1 || !sweep_and_grid_seconds) return fail(FB_EINVAL, "sweep a grid once first");
  FB_TRY(materialize_grid(h));  // (allocation outside the timed region)
  FB_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; r++) {
    FB_TRY(do_sweep(h, true));
    FB_TRY(materialize_grid(h));
  }
  FB_HIP(hipEventRecord(h->ev[1], h->stream));
  FB_HIP(hipStreamSynchronize(h->stream));
  float ms = 0;
  FB_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  *sweep_and_grid_seconds = ms * 1e-3 / reps;
  return FB_OK;
}

int fb_poly_time_stages(fb_poly_t h, int reps, double seconds[5]) {
  CHECK_POLY(h);
  if (!h->tetra || reps < 1 || !seconds) return fail(FB_EINVAL, "run sweep, classify and tetrahedralize once first");
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  for (auto& e : ev) FB_HIP(hipEventCreate(&e));
  double acc[5] = {0, 0, 0, 0, 0};
  int rc = FB_OK;
  const Grid& G = h->G;
  const long long words = (G.n_points + 63) / 64;
  const int pb = (int)std::min<long long>((G.n_points + kPB - 1) / kPB, std::max<long long>(1, (words + 15) / 16));  // (do_emit's grid)
  for (int r = 0; r < reps && rc == FB_OK; r++) {
    (void)hipEventRecord(ev[0], h->stream);
    rc = do_sweep(h, true);
    (void)hipEventRecord(ev[1], h->stream);
    if (rc == FB_OK) rc = do_classify(h);
    (void)hipEventRecord(ev[2], h->stream);
    hipLaunchKernelGGL(k_tet_vertices, dim3((unsigned)((((G.n_points + 63) >> 6) + kVW - 1) / kVW)), dim3(kPB), 0, h->stream, G, fast_div((unsigned int)G.g[0] * (unsigned int)G.g[1]), fast_div((unsigned int)G.g[0]), h->vinc.p, h->vbase.p, h->tv.p);
    (void)hipEventRecord(ev[3], h->stream);
    hipLaunchKernelGGL(k_tet_elements, dim3(pb), dim3(kPB), 0, h->stream, G, h->cinc.p, h->cbase.p, h->vinc.p, h->vbase.p, h->tt.p);
    (void)hipEventRecord(ev[4], h->stream);
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) { rc = fail(FB_EDEVICE, "stage timing failed"); break; }
    for (int k = 0; k < 4; k++) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
      acc[k] += ms * 1e-3;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev[0], ev[4]);
    acc[4] += ms * 1e-3;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  for (int k = 0; k < 5; k++) seconds[k] = acc[k] / reps;
  return rc;
}

}  // extern "C"
