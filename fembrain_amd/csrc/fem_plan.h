// Host-side plan of one rank's share of the FEM system: local numbering, halo lists, the 3x3-block
// sparsity pattern, its SELL-64 device layout and the per-(row,slot) element contribution lists that the
// row-gather assembly kernel walks.  No HIP calls in here -- the plan is exercised on CPU by the tests
// (including the world_size-2 gloo tests of the halo lists).
//
// Reference semantics restated: the pattern is every vertex pair of every tet with ascending columns
// (vegafem/corotationalLinearFEM/corotationalLinearFEM.cpp:163-186, sparseMatrix/sparseMatrix.cpp:238-262);
// the contribution lists replace the per-element rowIndices/columnIndices scatter caches
// (corotationalLinearFEM.cpp:482-502) by their transpose: for each matrix block, the (tet, i, j) that add to it,
// in ascending element order (the order the reference's element loop accumulates them, :230-469).
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <utility>
#include <vector>

namespace fb {

// vector whose resize() leaves new elements uninitialised: the 77 MB contribution table of a 1M-tet mesh is filled by the
// builder's threads, not zeroed first by one
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { typedef default_init_allocator<U> other; };
  default_init_allocator() = default;
  template <class U> default_init_allocator(const default_init_allocator<U>&) {}
  template <class U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};

constexpr int kSliceRows = 64;             // one wavefront lane per block row
constexpr uint32_t kNoContrib = 0xFFFFFFFFu;

struct FemPlan {
  // --- partition ---
  int n_global = 0, n_ranks = 1, rank = 0;
  std::vector<int> splits;                 // n_ranks+1 node range boundaries
  int node_lo = 0, node_hi = 0;
  int n_owned = 0, n_halo = 0, n_local = 0;
  std::vector<int> local2global;           // n_local: owned (ascending) then halo (ascending => grouped by owner)
  std::vector<int> halo_off;               // n_ranks+1: halo nodes owned by rank q are local ids n_owned+[halo_off[q],halo_off[q+1])
  std::vector<int> send_off;               // n_ranks+1
  std::vector<int> send_local;             // owned local ids to pack for rank q: [send_off[q], send_off[q+1]), ascending
  // --- local elements ---
  int n_tets = 0;                          // tets with at least one owned node
  std::vector<int> tet_global;             // n_tets global element ids (ascending); empty = 0..n_tets-1 of the caller's list
  std::vector<int> tets;                   // 4*n_tets local node ids
  long long n_owned_corners = 0;           // element corners on an owned node (the device builder's pair count)
  // --- block pattern of owned rows (CSR, ascending global column order) ---
  std::vector<int> bptr;                   // n_owned+1
  std::vector<int> bcol;                   // local column ids
  int n_blocks = 0;
  // --- SELL-64 layout ---
  int n_slices = 0;
  std::vector<int> slice_off;              // n_slices+1, in slots
  int n_slots = 0;                         // sum of slice widths
  std::vector<int> colidx;                 // n_slots*64 local column ids (padding: the row itself / 0)
  std::vector<int> blk_slot;               // n_blocks: global slot index (slice_off[s]+k) of CSR block p
  // --- contribution lists, [slot][t][lane] ---
  std::vector<int> slot_coff;              // n_slots: first contribution "row" of the slot
  std::vector<int> slot_ccnt;              // n_slots: max contributions over the 64 lanes
  int n_crows = 0;                         // sum of slot_ccnt
  std::vector<uint32_t, default_init_allocator<uint32_t>> contrib;  // n_crows*64, (tet<<4 | i<<2 | j) or kNoContrib
  // --- constraints ---
  std::vector<uint8_t> dofmask;            // 3*n_local: 1 free, 0 constrained
  int n_fixed_owned = 0;
};

// Builds the plan of `rank`.  tets: 4*n_tets global node ids.  fixed_dofs ascending global DOF ids.
// Returns 0 or a negative FB_E* code (text via fb::last_error()).
int build_fem_plan(FemPlan& plan, int n_nodes, int n_tets, const int* tets, int n_fixed, const int* fixed_dofs,
                   int n_ranks, int rank, const int* splits);
// The first half of build_fem_plan only: partition, local elements and numbering, halo and send lists -- what the device plan
// builder (plan_device.hip) needs from the host before it lays out pattern, SELL-64 and contribution lists itself.
// need_local_tets = false: when every element passed in is kept (per-rank ingest), plan.tets / plan.tet_global stay empty (the
// kept elements are the caller's, in the caller's order) and the caller numbers them locally itself.
int build_fem_partition(FemPlan& plan, int n_nodes, int n_tets, const int* tets, int n_ranks, int rank, const int* splits,
                        bool need_local_tets = true);
// the two ends of build_fem_partition, for a partition computed elsewhere (plan_device.hip device_partition): reset + node
// ranges (validated), and local numbering / per-owner offsets from the ascending halo list
int begin_fem_partition(FemPlan& plan, int n_nodes, int n_tets, int n_ranks, int rank, const int* splits);
void set_partition_halo(FemPlan& plan, const std::vector<int>& halo);
int plan_set_constraints(FemPlan& plan, int n_fixed, const int* fixed_dofs);
// What a rank's vote on the node order of a sharded handle is made of (fem.hip vote_shard_order): the number of OTHER ranks that the
// elements with a node in this rank's range couple it to under the caller's numbering (-1: bad ranges or a node id outside the
// mesh -- the builder reports those), a checksum of the whole element list as handed over, and one of the ranges; how many elements
// have a node of this rank, and how many of those also have a node of another rank.  splits may be null (equal ranges).  At most 64 ranks.
void shard_neighbour_count(int n_nodes, int n_tets, const int* tets, int n_ranks, int rank, const int* splits, int* neighbours, unsigned long long* list_sum,
                           int* splits_sum, int* own_elements, int* boundary_elements);

}  // namespace fb
