// RCCL-over-xGMI communicator used by sharded FEM handles.  librccl is opened lazily with dlopen so that a
// single-GPU process never needs it.  Two collectives only, both on the handle's stream:
//   * all-reduce (sum) of a few fp64 scalars      -- the PCG dots
//   * neighbour exchange of halo node triples     -- ncclSend/ncclRecv inside one group
#pragma once
#include <hip/hip_runtime.h>

struct fb_comm_s {
  int rank = 0, n_ranks = 1, device = 0;
  void* nccl = nullptr;   // ncclComm_t
  void* local = nullptr;  // host-staged shared-memory transport (test hook, fb_comm_create_local)
};

namespace fb {
int comm_allreduce_sum(fb_comm_s* c, double* dev_buf, int count, hipStream_t s);
// send_off/recv_off: n_ranks+1 offsets in NODES of `width` doubles; sendbuf packed by destination; recv lands at
// recv_base + width*recv_off[q]
int comm_exchange_nodes(fb_comm_s* c, const double* sendbuf, const int* send_off, double* recv_base, const int* recv_off, int width,
                        hipStream_t s);
}  // namespace fb
