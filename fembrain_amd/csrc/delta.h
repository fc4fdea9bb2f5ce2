// Re-sync from a DESCRIPTION of a topology change (fb_fem_resync_delta) instead of the whole mesh.
//
// CuttableMesh::cut (src/deformable/CuttableMesh.cpp:283-470) subdivides the elements its blade crosses: the cut cells are erased from
// the cell list keeping the order of the rest (VolMesh::remove_cell_core, src/deformable/VolMesh.cpp:630: m_vCells.erase), the pieces
// and the new nodes are appended (insert_cell / insert_node: push_back, :1083-1088), and an edge split re-points the cells on that edge
// in place (:1630-1650).  Deformable::syncForceModel (src/deformable/Deformable.cpp:127-220) then rebuilds everything from the whole
// mesh.  Here the element list and the rest positions stay on the device between re-syncs, the change is applied to them there, and
// the sorted (row, column) -> contribution list the plan was built from (plan_device.hip) is UPDATED -- pairs of removed and changed
// elements dropped, element ids renumbered (a monotone map: the order stands), the pairs of changed and added elements sorted among
// themselves and merged in -- instead of sorted again.  The rest of the plan builder runs on the same list a full rebuild would have
// sorted, so the plan comes out bit for bit the same.
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
  DevBuf<unsigned long long> nk, nks;  // keys of the new pairs, unsorted / sorted
  DevBuf<uint32_t> nv, nvs;
  DevBuf<int> tile_i;               // delta_sorted_pairs: per tile of the old list -- entries that go, prefix sums of those that stay, bounds in the new entries
  DevBuf<unsigned int> drop_bits;   // ... a bit per entry of the old list: it goes
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
// The sorted pair list of the workspace, updated (see the header comment).  tets_old: the old element list in the old internal ids (the
// entries of removed and changed elements are found by their keys); tets_new: the new element list in the new internal ids;
// span: its widest element (decides the key width).  On return W.sorted describes the new list in W.keys_s / W.vals_s.
int delta_sorted_pairs(hipStream_t s, MeshDelta& D, const int4* tets_old, const int4* tets_new, int n_nodes_new, int span, PlanWorkspace& W);

}  // namespace fb
