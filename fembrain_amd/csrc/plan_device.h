// Device-side plan builder (plan_device.hip): the per-step arrays of fem_plan.h for an unsharded handle, built on the GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "common.h"

namespace fb {

// temporaries of the builder, kept by the caller between builds (a re-sync after every cut would otherwise spend more time
// in hipMalloc / hipFree of ~700 MB than in the kernels)
struct PlanWorkspace {
  DevBuf<unsigned long long> keys, keys_s, ukeys;
  DevBuf<uint32_t> vals, vals_s;
  DevBuf<unsigned int> ucnt, cstart, nruns;
  DevBuf<int> width, flags;
  DevBuf<char> temp;
  DevBuf<unsigned char> nodeflag;          // device_partition: halo marks over the global nodes
  DevBuf<unsigned long long> sendmask;     // device_partition: per owned node, the ranks that want it
  DevBuf<int> picked, splits, idsel;       // device_partition: selected ids (+ count at the end), node ranges, kept element ids
  DevBuf<unsigned char> keep;              // device_partition: element has an owned node
  DevBuf<int4> tetsel;                     // device_partition: the kept elements (swapped with the handle's buffer)
  size_t bytes() const {  // what is really held: capacities, not the sizes in use
    return nodeflag.cap + sendmask.cap * 8 + picked.cap * 4 + idsel.cap * 4 + keep.cap + tetsel.cap * 16 + keys.cap * 8 + keys_s.cap * 8 + ukeys.cap * 8 + vals.cap * 4 + vals_s.cap * 4 +
           ucnt.cap * 4 + cstart.cap * 4 + nruns.cap * 4 + width.cap * 4 + flags.cap * 4 + temp.cap;
  }
  void release() {
    keys.release(); keys_s.release(); ukeys.release(); vals.release(); vals_s.release(); ucnt.release(); cstart.release(); nruns.release();
    width.release(); flags.release(); temp.release(); nodeflag.release(); sendmask.release(); picked.release(); splits.release(); idsel.release(); keep.release(); tetsel.release();
  }
};

struct DevicePlan {
  // outputs, allocated by the builder (the caller owns the buffers)
  DevBuf<int>*slice_off = nullptr, *colidx = nullptr, *slot_coff = nullptr, *slot_ccnt = nullptr;
  DevBuf<uint32_t>* contrib = nullptr;
  DevBuf<short>* coldelta = nullptr;  // 16-bit column word per slot and lane (plan_device.hip k_plan_sell); usable by the SpMV when deltas_fit16
  DevBuf<int>* halo_base = nullptr;   // shards: the slice's lowest halo column (the halo form of the 16-bit words), else unused
  bool deltas_fit16 = false;
  DevBuf<int>*bptr = nullptr, *bcol = nullptr, *blk_slot = nullptr;  // CSR pattern and slot of every block (inspection entry points)
  DevBuf<unsigned int>* ucnt_keep = nullptr;  // optional: the pairs of every block (contributions, + 1 for a diagonal block's marker), kept for fb_fem_resync_delta
  int n_blocks = 0, n_slices = 0, n_slots = 0, n_crows = 0;
  std::vector<int> slice_off_host;
  int first_bad_tet = -1;  // lowest tet with a node id outside [0, n_nodes), -1 if none (the build then fails with FB_EINVAL)
};

// One rank's share of a sharded system: rows are the owned nodes (local ids [0, n_rows)), columns local ids (owned, then the
// halo nodes in ascending global id); inside a row the blocks are ordered by GLOBAL column id -- the reference's order and
// what fem_plan.cpp builds -- so the builder sorts by (local row, global column) and maps the columns back.
struct PlanShard {
  int n_rows = 0;               // owned nodes
  int node_lo = 0;              // global id of local row 0
  int n_global = 0;             // nodes of the whole mesh (column key width)
  const int* d_halo = nullptr;  // device: ascending global ids of the n_halo halo nodes (local id n_rows + k)
  int n_halo = 0;
  long long n_pairs = 0;        // vertex pairs whose row is owned (host count) + n_rows markers
};

// The partition of one rank on the device: what fem_plan.cpp's build_fem_partition computes on the host.  tets: n_tets x int4
// GLOBAL node ids -- the rank's own elements (per-rank ingest) or any superset up to the whole mesh; on success it holds the
// n_kept elements with an owned node, in the caller's order, in local ids (owned first, then the halo ascending).  Outputs:
// halo (ascending global ids), per neighbour rank the owned local ids it wants (send_off / send_local, ascending), the count of
// element corners on owned nodes, the caller's indices of the kept elements.  n_ranks <= 64.  Synchronises the stream.
struct DevicePartition {
  std::vector<int> halo, send_off, send_local;
  std::vector<int> tet_global;  // ids of the kept elements in the caller's list; empty when all were kept
  long long owned_corners = 0;
  int first_bad_tet = -1, bad_node = 0, n_kept = 0;
  bool all_kept = false;
};
int device_partition(hipStream_t s, int n_tets, DevBuf<int4>& tets, int n_global, int n_ranks, int rank, const std::vector<int>& splits,
                     DevicePartition& out, PlanWorkspace& ws);

// d_tets: n_tets x int4 node ids (local ids for a shard); their range is checked here (first_bad_tet).  shard = nullptr: the
// unsharded system (every node a row).  The last kernel may still be running when it returns (stream order).
// span >= 0: the widest element of the list (largest id difference inside a tet, renumber.h) when the caller has measured it -- lets the sort use 32-bit keys
int build_plan_device(hipStream_t s, int n_nodes, int n_tets, const int4* d_tets, DevicePlan& out, PlanWorkspace& ws, const PlanShard* shard = nullptr, int span = -1);

// SELL-64 layout, slot table and list heights of the pattern in out.bptr / out.bcol with `ucnt` pairs per block (the diagonal block's count
// includes its marker): fills slice_off, colidx, blk_slot, coldelta, slot_ccnt, slot_coff and n_slices, n_slots, n_crows, deltas_fit16,
// slice_off_host.  The second half of the builder; delta.hip calls it on the pattern it has updated.  Synchronises the stream.
int plan_layout_from_csr(hipStream_t s, int n_nodes, const unsigned int* ucnt, bool shard, DevicePlan& out, PlanWorkspace& ws);

// Incidence lists of the element-major assembly kernel (fem_device.hip.h k_assemble_tets), derived from a plan that is already
// on the device (whichever builder made it): per slice the longest list of its 64 rows (inc_off = prefix sums), per (list row, lane)
// the word element << 2 | corner (kNoContrib past the end) and the slots of the element's four blocks in the lane's row (4 x u8).
// The incidence list of a row is the contribution list of its diagonal block, so the ascending element order carries over.
constexpr int kIncMaxWidth = 31;  // widest slice (in slots) the element-major assembly kernel takes
int build_incidence_device(hipStream_t s, int n_slices, int n_owned, const int* slice_off, const int* colidx, const int* slot_coff, const int* slot_ccnt,
                           const uint32_t* contrib, const int4* tets, DevBuf<int>& inc_off, DevBuf<uint32_t>& inc, DevBuf<uint32_t>& inc_slot, PlanWorkspace& ws,
                           bool ascending_columns);  // ascending_columns: the column ids of a row ascend (an unsharded plan): binary searches

}  // namespace fb
