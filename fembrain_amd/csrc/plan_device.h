// Device-side plan builder (plan_device.hip): the per-step arrays of fem_plan.h for an unsharded handle, built on the GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "common.h"

namespace fb {

struct DevicePlan {
  // outputs, allocated by the builder (the caller owns the buffers)
  DevBuf<int>*slice_off = nullptr, *colidx = nullptr, *slot_coff = nullptr, *slot_ccnt = nullptr;
  DevBuf<uint32_t>* contrib = nullptr;
  DevBuf<short>* coldelta = nullptr;  // column - row per slot and lane; usable by the SpMV when deltas_fit16
  bool deltas_fit16 = false;
  DevBuf<int>*bptr = nullptr, *bcol = nullptr, *blk_slot = nullptr;  // CSR pattern and slot of every block (inspection entry points)
  int n_blocks = 0, n_slices = 0, n_slots = 0, n_crows = 0;
  std::vector<int> slice_off_host;
  int first_bad_tet = -1;  // lowest tet with a node id outside [0, n_nodes), -1 if none (the build then fails with FB_EINVAL)
};

// d_tets: n_tets x int4 node ids; their range is checked here (first_bad_tet).  Synchronises the stream before it returns.
int build_plan_device(hipStream_t s, int n_nodes, int n_tets, const int4* d_tets, DevicePlan& out);

}  // namespace fb
