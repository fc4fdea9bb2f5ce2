// Locality renumbering of the mesh nodes behind the ABI ("the same contiguous-range rule after an RCM /
// space-filling-curve renumbering done once").  The reference appends every node a cut creates at the END of the node list
// (src/deformable/VolMesh.cpp:1086-1091,1639-1642, called from CuttableMesh.cpp:283,407) and the tet meshes it ships
// (data/models/blobtree/*.veg) are TetGen outputs -- surface vertices first, Steiner points after.  On such numbering a row's
// columns lie anywhere in the vector: no 16-bit column words, gathers without an XCD-local slab, a producer list of "everyone"
// in the persistent solver, and every rank a neighbour of every other (measured at 1M tets: 46-47 us per PCG iteration against
// 15.8 in grid order, profiles/r04_numbering_probe_before.json).
//
// The internal order is a SLAB order: nodes sorted by the lexicographic key (q_major, q_mid, q_minor) of their rest position
// quantised to cells of edge h = 0.8 (bounding-box volume / nodes)^(1/3), major = the longest axis of the bounding box (an axis
// displaces an earlier one only when it is 5 % longer, so a cube keeps x, y, z), ties in caller order (stable sort).  A
// structured grid in any caller order gets its own plane-by-plane order back; an unstructured mesh gets slabs one cell thick, so
// a row's columns lie within ~2 slabs = O(n^(2/3)) ids, and a contiguous range of ids is a slab of the body with two
// neighbours.  Elements keep the caller's order (the order in which a block accumulates its contributions is the reference's,
// corotationalLinearFEM.cpp:230-469); only node ids are mapped, on the way in and on the way out.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "common.h"
#include "plan_device.h"

namespace fb {

struct SlabKeyGeom {
  double lo[3];
  double inv_h;
  int axis[3];  // major, mid, minor
  int bits[3];  // key bits of each, in that order
};

__host__ __device__ inline unsigned long long slab_key(const SlabKeyGeom& g, double x, double y, double z) {
  const double p[3] = {x, y, z};
  unsigned long long k = 0;
  for (int a = 0; a < 3; a++) {
    const int ax = g.axis[a];
    const double t = (p[ax] - g.lo[ax]) * g.inv_h + 0.5;
    const long long top = (1LL << g.bits[a]) - 1;
    // clamped in floating point BEFORE the conversion (a huge or infinite coordinate converts to anything on the host and saturates on
    // the device: the two orders would disagree); NaN compares false: cell 0
    const long long q = !(t > 0.0) ? 0 : (t >= (double)top ? top : (long long)t);
    k = (k << g.bits[a]) | (unsigned long long)q;
  }
  return k;
}

// key geometry from the bounding box; false when the nodes do not span a volume to cut into cells (all on one point)
bool slab_key_geometry(int n_nodes, const double lo[3], const double hi[3], SlabKeyGeom* out);

// AUTO renumbers a mesh of at least kRenumberMinNodes nodes whose widest element (largest difference of two node ids of one tet
// = the half bandwidth of the matrix) exceeds what the solver's fast paths take -- 32,767 (16-bit column words) or nodes / 8 (at
// most 64 of 256 workgroups produce a workgroup's columns) -- and keeps the result if it makes the elements at least a quarter narrower on average
constexpr int kRenumberMinNodes = 8192;
inline int renumber_span_limit(int n_nodes) { return n_nodes / 8 < 32767 ? n_nodes / 8 : 32767; }
// Second stage of the internal order (SELL-C-sigma, C = 64): inside windows of sigma_window() consecutive nodes of the slab order the
// nodes are sorted by the number of elements on them, descending (stable) -- rows of like length then share a 64-row slice, whose width
// is its longest row.  Applied when the slab order alone would pad the matrix by more than a tenth (estimated from the element counts:
// sum over slices of 64 x the largest count against the sum of the counts): a jittered-grid Delaunay mesh pads 33 % in slab order and
// 7.5 % after this stage (0.72 of the slots), the headline cube cut twice 48 % -> 10 % (0.58); a grid cube pads 3 % and is left alone.
constexpr int kSigmaMaxCount = 1023;  // element counts are clipped to 10 bits of the second sort key
// The window: the rows of about ONE workgroup of the persistent solver (n_slices / 256 slices, between 4 and 32) -- sorting inside a
// workgroup's rows leaves its producer list alone; windows of 1,024 rows pushed the 606k-tet Delaunay probe from 57 to more than 64
// producers (poll-all: 25.2 instead of 22.5 us per iteration) while a 1M-tet cut mesh gained either way.  A function of the node
// count only, so that the host restatement (host_slab_order) agrees without a device.  FEMBRAIN_SIGMA_WINDOW overrides (development).
inline int sigma_window(int n_nodes) {
  const char* e = getenv("FEMBRAIN_SIGMA_WINDOW");
  if (e && atoi(e) >= 64) return (atoi(e) / 64) * 64;
  int per = ((n_nodes + 63) / 64) / 256;
  per = per < 4 ? 4 : (per > 32 ? 32 : per);
  return 64 * per;
}
inline unsigned long long sigma_key(int position, int window, int count) {
  return ((unsigned long long)(position / window) << 10) | (unsigned long long)(kSigmaMaxCount - (count < kSigmaMaxCount ? count : kSigmaMaxCount));
}

struct Renumbering {
  bool active = false;
  int span_before = 0, span_after = 0;      // widest element before / after (span_after = span_before when inactive)
  double mean_before = 0, mean_after = 0;   // mean width of an element before / after
  DevBuf<int> d_old_of_new, d_new_of_old;   // internal id -> caller id and back
  SlabKeyGeom geom = {};                    // the key geometry the order was built with, and the nodes' keys in the internal order
  DevBuf<unsigned long long> d_keys;        // (fb_fem_resync_delta puts new nodes into the order under the same geometry)
  bool sigma = false;                       // the second stage ran: d_keys holds sigma_key values, d_win_keys the slab key of every window's first node
  int n_windows = 0, window = 0;
  DevBuf<unsigned long long> d_win_keys;
  DevBuf<int> d_count;                      // elements on every node (caller ids), scratch of the build
  std::vector<int> old_of_new, new_of_old;  // host copies, fetched on demand (inspection entry points)
  int n = 0;
  int host_maps(hipStream_t s);
  void clear() { active = false; sigma = false; n_windows = 0; span_before = span_after = 0; mean_before = mean_after = 0; old_of_new.clear(); new_of_old.clear(); }
};

// widest element of the list under the node map `new_of_old` (nullptr: identity); tets with an id outside [0, n_nodes) are left
// to the plan builder's range check.  Synchronises the stream.
int tet_span_device(hipStream_t s, int n_tets, const int4* d_tets, int n_nodes, const int* d_new_of_old, PlanWorkspace& W, int* span, double* mean);
// Phase 1 (mode: FB_RENUMBER_*): measures the widest element and says whether a renumbering is to be tried (*want); phase 2
// builds it from the caller-order rest positions on the device (3 doubles per node) and decides whether it stands (R.active).
// Both synchronise the stream.
int renumber_decide(hipStream_t s, int mode, int n_nodes, int n_tets, const int4* d_tets, PlanWorkspace& W, Renumbering& R, bool* want);
// force_sigma: the second stage whatever the padding (fb_fem_params.expect_cuts warms its kernels and temporaries at creation)
int renumber_build(hipStream_t s, int mode, int n_nodes, int n_tets, const int4* d_tets, const double* d_xyz, PlanWorkspace& W, Renumbering& R, bool force_sigma = false);
int relabel_tets(hipStream_t s, int n_tets, int4* d_tets, int n_nodes, const int* d_new_of_old);
// node-wise permutations of arrays of `width` doubles per node: dst[l] = src[map[l]] / dst[map[l]] = src[l]
int gather_nodes(hipStream_t s, int n, int width, const double* src, const int* map, double* dst);
int scatter_nodes(hipStream_t s, int n, int width, const double* src, const int* map, double* dst);

// Constraint masks of an unsharded handle built on the device: dofmask (1 free / 0 constrained per DOF) and nodemask (the three bytes of a
// node as bits 0..2) in the INTERNAL order from the caller's ascending list of constrained DOFs (IntegratorBaseSparse::setConstrainedDOF's
// argument; validated here as fem_plan.cpp's plan_set_constraints validates it).  d_new_of_old: nullptr = identity.
int device_constraint_masks(hipStream_t s, int n_nodes, int n_fixed, const int* fixed_dofs, const int* d_new_of_old, DevBuf<int>& stage, DevBuf<unsigned char>& dofmask,
                            DevBuf<unsigned char>& nodemask);

// the same on the host (fb_plan_* test entry points, CPU tests): old_of_new of the slab order, spans under both orders
int host_slab_order(int n_nodes, const double* xyz, int n_tets, const int* tets, std::vector<int>& old_of_new, int* span_before, int* span_after, double* mean_before = nullptr,
                    double* mean_after = nullptr);

}  // namespace fb
