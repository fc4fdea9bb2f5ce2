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
  bool want_p2p = false;  // try the direct xGMI mailbox transport for the per-iteration exchanges (p2p.hip)
};

namespace fb {
int comm_allreduce_sum(fb_comm_s* c, double* dev_buf, int count, hipStream_t s);
// send_off/recv_off: n_ranks+1 offsets in NODES of `width` doubles; sendbuf packed by destination; recv lands at
// recv_base + width*recv_off[q]
int comm_exchange_nodes(fb_comm_s* c, const double* sendbuf, const int* send_off, double* recv_base, const int* recv_off, int width,
                        hipStream_t s);
// every rank contributes `bytes` host bytes; `all` receives n_ranks * bytes in rank order (setup-time only)
int comm_allgather_bytes(fb_comm_s* c, const void* mine, void* all, size_t bytes, hipStream_t s);

// ---- direct peer-to-peer transport (p2p.hip) ------------------------------------------------------------------------
// The PCG iteration needs two tiny exchanges (a halo refresh of the search direction and a 3-scalar sum).  Through RCCL
// each costs a collective launch (tens of microseconds) -- more than the iteration's compute on a strongly-scaled mesh.
// Here every rank owns a fine-grained "inbox" in its HBM, maps the peers' inboxes through HIP IPC, and the exchange is
// done by ordinary kernels: stores straight into the peer's inbox over xGMI, a system-scope release of a sequence
// number, and a bounded spin on the own inbox flags.  RCCL stays in charge of bootstrap (handle all-gather) and is the
// fallback when peer mapping is unavailable.  Every wait has a wall-clock bound: on expiry the inbox error word is set,
// all later waits fall through, and the host reports FB_ECOMM at the next synchronisation point.
constexpr int kP2PMaxRanks = 16;
// FEMBRAIN_REMOTE_DELAY_US (development): wall_clock64 ticks by which remote signals are delayed in the one-GPU rehearsals of N > 1
inline long long remote_delay_ticks() {
  const char* e = getenv("FEMBRAIN_REMOTE_DELAY_US");
  return e ? (long long)(atof(e) * 100.0) : 0LL;
}

struct P2PDev {  // passed to kernels by value
  int rank, n_ranks;
  char* inbox;
  char* peer[kP2PMaxRanks];          // peer[rank] == inbox
  int peer_seg[kP2PMaxRanks];        // first node of MY segment inside peer q's halo
  long long peer_cap[kP2PMaxRanks];  // halo nodes per buffer of peer q's inbox
  long long cap;
  long long timeout_ticks;           // wall_clock64 ticks (100 MHz)
  long long delay_ticks;             // development (FEMBRAIN_REMOTE_DELAY_US): every remote signal is raised this much later and seen this much later
};
struct P2P;
struct P2PArgs {  // what a PCG kernel that carries an exchange in its prologue needs
  P2PDev dev;
  unsigned long long seq;        // of the 3-scalar sum the vector pass posts and awaits
  unsigned long long halo_seq;   // of the halo refresh the SpMV sends and awaits
  const int *send_ids, *send_off, *halo_off;  // device copies of the plan's exchange lists
  const unsigned char* slice_halo;            // per SELL slice: 1 if any of its columns is a halo node
};
// next sequence number of the 3-scalar sum / of the halo refresh, for kernels that carry the exchange themselves
P2PArgs p2p_next_sum(P2P* p);
unsigned long long p2p_next_halo(P2P* p);
// collective over the communicator; *out == nullptr (and FB_OK) when the ranks agree that the transport is unavailable
int p2p_attach(fb_comm_s* c, int n_halo_nodes, const int* halo_off, hipStream_t s, P2P** out);
void p2p_detach(P2P* p);
// halo refresh of a per-node array of `width` doubles (<= 12): my send list goes into the peers' inboxes, theirs lands
// in vec[width * n_owned ...]
int p2p_halo(P2P* p, int width, const int* send_ids, const int* send_off_dev, int n_halo, const int* halo_off_dev, int n_owned, double* vec,
             hipStream_t s);
// out[c] = sum over ranks (fixed rank order, bitwise identical everywhere) of sum_i partial[c * n + i], c < count <= 8
int p2p_reduce(P2P* p, const double* partial, int n, int count, double* out, hipStream_t s);
// FB_OK, or FB_ECOMM if any wait on this rank timed out since attach (synchronises the stream)
int p2p_check(P2P* p, hipStream_t s);
}  // namespace fb
