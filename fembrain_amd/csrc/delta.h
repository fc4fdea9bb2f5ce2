// Re-sync from a DESCRIPTION of a topology change (fb_fem_resync_delta) instead of the whole mesh.
//
// CuttableMesh::cut (src/deformable/CuttableMesh.cpp:283-470) subdivides the elements its blade crosses: the cut cells are erased from
// the cell list keeping the order of the rest (VolMesh::remove_cell_core, src/deformable/VolMesh.cpp:630: m_vCells.erase), the pieces
// and the new nodes are appended (insert_cell / insert_node: push_back, :1083-1088), and an edge split re-points the cells on that edge
// in place (:1630-1650).  Deformable::syncForceModel (src/deformable/Deformable.cpp:127-220) then rebuilds everything from the whole
// mesh.  Here the element list, the rest positions and the PLAN stay on the device between re-syncs and the change is applied to them
// there: the block pattern (CSR: bptr, bcol and the pairs of every block) keeps the rows no element of the change touches -- columns
// through the monotone node map -- and merges the few others with the sorted pairs of the changed and added elements; the SELL layout is
// laid out again from the pattern (plan_layout_from_csr, the second half of the builder); the contribution table is written from the old
// table with the element ids renumbered.  The plan is the same function of the same mesh as a full rebuild's and comes out bit for bit
// the same (round 4 updated the sorted pair list the builder starts from and ran five passes over it again; see delta.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "plan_device.h"
#include "renumber.h"

namespace fb {

// the change on the device; node ids of changed_nodes / added in the CALLER's numbering until delta_relabel_nodes maps them
struct MeshDelta {
  int n_tets_old = 0, n_removed = 0, n_changed = 0, n_added = 0, n_new_nodes = 0;
  int n_kept = 0;  // old elements that stay (changed ones included)
  DevBuf<int> ints;                 // staging: removed | changed_ids | changed_nodes | added (each padded to 4 ints)
  const int* removed = nullptr;     // ascending old element ids
  const int* changed_ids = nullptr; // ascending old element ids, none of them removed
  int4* changed_nodes = nullptr;
  int4* added = nullptr;
  DevBuf<double> new_xyz;           // rest positions of the appended nodes
  DevBuf<unsigned char> estate;     // per old element: 0 kept, 1 removed, 2 changed
  DevBuf<int> pos;                  // per old element: its new id (elements before it that stay)
  DevBuf<unsigned long long> nk, nks;  // keys of the new pairs (new row << 32 | new column), unsorted / sorted
  DevBuf<uint32_t> nv, nvs;
  DevBuf<int> newid;                // per old element: its new id, -1 if its contributions go (removed, or changed: those come back as new pairs)
  DevBuf<int> oldrow;               // per new row: the old row, -1 for a new node (renumbered handles with new nodes)
  DevBuf<unsigned int> touched;     // count of touched rows | a word per new row: it loses or gains contributions | its fresh entries | the list of touched rows
  DevBuf<int> len;                  // new row lengths
  DevBuf<int> slot_slice;           // per new slot: its slice
  DevBuf<int> src;                  // per new block: the old block its kept words come from, -1 none
  // the plan being built next to the handle's (swapped in when it is complete)
  DevBuf<int> bptr2, bcol2, blk_slot2, slice_off2, slot_coff2;
  DevBuf<unsigned int> ucnt2;
  DevBuf<uint32_t> contrib2;
  bool mapped = false;              // delta_node_order ran for this change: imap / newint / node_keys are its
  DevBuf<int> new_count;            // elements of the change on every new node
  DevBuf<int> imap, newint;         // renumbered handles: old internal id -> new internal id; new node k -> its internal id
  DevBuf<unsigned long long> node_keys;  // merged slab keys (internal order)
  int n_tets_new() const { return n_kept + n_added; }
};

// uploads the change (host arrays already validated), marks the elements, scans the new ids.  Does not synchronise.
int delta_upload(hipStream_t s, int n_tets_old, int n_removed, const int* removed, int n_changed, const int* changed_ids, const int* changed_nodes, int n_added,
                 const int* added, int n_new_nodes, const double* new_xyz, MeshDelta& D, PlanWorkspace& W);
// changed_nodes / added: node ids through `map` (caller id -> internal id)
int delta_relabel_nodes(hipStream_t s, MeshDelta& D, int n_nodes, const int* map);
// the new element list: kept elements in their order (node ids through imap, nullptr = as they are), changed ones with their new nodes, added ones behind
int delta_tets(hipStream_t s, const MeshDelta& D, const int4* tets_old, const int* imap, int4* tets_new);
// A renumbered handle: the new nodes take their place in the slab order under the key geometry the order was built with (frozen until the
// next full rebuild), ties behind the old nodes.  Outputs D.imap, D.newint, D.node_keys and the two maps of the new order.
// n_windows > 0: the order has the second stage (renumber.h, sigma_window): keys_old are its keys, win_keys the slab key of every window's first node.
int delta_node_order(hipStream_t s, MeshDelta& D, int n_old, const SlabKeyGeom& g, const unsigned long long* keys_old, const int* old_of_new_old,
                     DevBuf<int>& old_of_new, DevBuf<int>& new_of_old, PlanWorkspace& W, int n_windows, const unsigned long long* win_keys);
// rest positions in the new internal order
int delta_positions(hipStream_t s, const MeshDelta& D, int n_old, const double* x0_old, double* x0_new);
// The plan the handle holds (OldPlanArrays: device pointers of the CURRENT plan, n_nodes / n_blocks its sizes), updated to the new mesh.
// tets_old: the old element list in the old internal ids; tets_new: the new list in the new internal ids.  Builds into D's second set of
// buffers and into out.colidx / slot_ccnt / coldelta (which nothing reads any more); the caller swaps bptr2 ... contrib2 with its own.
// out.bptr etc. must point at D.bptr2 ...; fills out.n_blocks, n_slices, n_slots, n_crows, deltas_fit16, slice_off_host.
struct OldPlanArrays {
  const int* bptr; const int* bcol; const unsigned int* ucnt; const int* slice_off; const int* slot_coff; const uint32_t* contrib;
  int n_nodes, n_blocks;
};
// room for a change of n_fresh pairs (16 per changed or added element), so that the first change does not allocate it
int delta_reserve(hipStream_t s, MeshDelta& D, PlanWorkspace& W, long long n_fresh);
int delta_plan(hipStream_t s, MeshDelta& D, const OldPlanArrays& old_plan, const int4* tets_old, const int4* tets_new, int n_nodes_new, DevicePlan& out, PlanWorkspace& W);

}  // namespace fb
