// Direct peer-to-peer exchange for the sharded PCG loop (see comm.h): inbox layout, the two exchange kernels and the
// collective set-up over HIP IPC.
#include "comm.h"
#include "common.h"
#include "fem_kernels.h"
#include "p2p_device.hip.h"

namespace fb {

namespace {

constexpr int kMaxWidth = kP2PMaxWidth;

struct Meta {  // what every rank publishes at attach time
  hipIpcMemHandle_t handle;
  long long cap;
  int halo_off[kP2PMaxRanks + 1];
  int ok;
};

// phase 1: the sender jobs (one per neighbour and chunk) are dealt to the blocks; phase 2: every block waits for the
// neighbours' chunks and copies its share of their values from my inbox into the halo part of the vector.  A block in
// phase 2 only depends on the peers' phase 1; no grid-wide ticket.
__global__ __launch_bounds__(kBlock) void k_p2p_halo(P2PDev c, int width, unsigned long long seq, const int* __restrict__ send_ids,
                                                     const int* __restrict__ send_off, int n_halo, const int* __restrict__ halo_off, int n_owned,
                                                     double* __restrict__ vec) {
  p2p_send_halo_jobs(c, seq, width, send_ids, send_off, vec);
  p2p_wait_halo(c, seq, halo_off, width);
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x, stride = (long long)gridDim.x * kBlock;
  const double* in = (const double*)(c.inbox + kOffHalo) + (size_t)(seq & 1ULL) * c.cap * kMaxWidth;
  double* out = vec + (size_t)width * n_owned;
  for (long long i = tid; i < (long long)n_halo * width; i += stride) out[i] = __builtin_nontemporal_load(in + i);
}

// one block: fold my per-block partial sums, post them into every rank's inbox (mine included), wait for everybody's,
// add in rank order.
__global__ __launch_bounds__(kBlock) void k_p2p_reduce(P2PDev c, unsigned long long seq, const double* __restrict__ partial, int n, int count,
                                                       double* __restrict__ out) {
  __shared__ double lds[4];
  __shared__ double mine[8];
  __shared__ double tot[8];
  p2p_post_sums(c, seq, partial, n, count, lds, mine);
  p2p_wait_sums(c, seq, count, tot);
  if ((int)threadIdx.x < count) out[threadIdx.x] = tot[threadIdx.x];
}

}  // namespace

struct P2P {
  P2PDev dev;
  fb_comm_s* comm = nullptr;
  void* opened[kP2PMaxRanks] = {nullptr};
  unsigned long long halo_seq = 0, red_seq = 0;
  double* probe = nullptr;  // 2 doubles of ordinary device memory
  size_t inbox_bytes = 0;
};

void p2p_detach(P2P* p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < p->dev.n_ranks; q++)
    if (p->opened[q]) (void)hipIpcCloseMemHandle(p->opened[q]);
  if (p->dev.inbox) (void)hipFree(p->dev.inbox);
  if (p->probe) (void)hipFree(p->probe);
  delete p;
}

int p2p_reduce(P2P* p, const double* partial, int n, int count, double* out, hipStream_t s) {
  if (count < 1 || count > 8 || n < 1 || n > kMaxPartials) return fail(FB_EINVAL, "p2p_reduce: bad sizes");
  hipLaunchKernelGGL(k_p2p_reduce, dim3(1), dim3(kBlock), 0, s, p->dev, ++p->red_seq, partial, n, count, out);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

P2PArgs p2p_next_sum(P2P* p) {
  P2PArgs a;
  memset(&a, 0, sizeof a);
  a.dev = p->dev;
  a.seq = ++p->red_seq;
  return a;
}

unsigned long long p2p_next_halo(P2P* p) { return ++p->halo_seq; }

int p2p_halo(P2P* p, int width, const int* send_ids, const int* send_off_dev, int n_halo, const int* halo_off_dev, int n_owned, double* vec,
             hipStream_t s) {
  if (width < 1 || width > kMaxWidth || n_halo > p->dev.cap) return fail(FB_EINVAL, "p2p_halo: bad sizes");
  const long long work = (long long)n_halo * width;
  const int blocks = (int)std::max<long long>(2 * kP2PChunks, std::min<long long>((work + kBlock - 1) / kBlock, 64));
  hipLaunchKernelGGL(k_p2p_halo, dim3(blocks), dim3(kBlock), 0, s, p->dev, width, ++p->halo_seq, send_ids, send_off_dev, n_halo, halo_off_dev, n_owned,
                     vec);
  FB_HIP(hipGetLastError());
  return FB_OK;
}

int p2p_check(P2P* p, hipStream_t s) {
  unsigned long long err = 0;
  FB_HIP(hipMemcpyAsync(&err, p->dev.inbox + kOffErr, sizeof err, hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  if (err) return fail(FB_ECOMM, "peer-to-peer exchange timed out on rank %d (a peer stopped or the mapping is broken)", p->dev.rank);
  return FB_OK;
}

int p2p_attach(fb_comm_s* c, int n_halo_nodes, const int* halo_off, hipStream_t s, P2P** out) {
  *out = nullptr;
  if (!c || c->n_ranks < 2 || !c->want_p2p) return FB_OK;
  if (c->n_ranks > kP2PMaxRanks) return FB_OK;
  const int n = c->n_ranks, me = c->rank;
  P2P* p = new P2P;
  p->comm = c;
  memset(&p->dev, 0, sizeof p->dev);
  p->dev.rank = me; p->dev.n_ranks = n;
  p->dev.cap = std::max(1, n_halo_nodes);
  const char* te = getenv("FEMBRAIN_P2P_TIMEOUT_MS");
  p->dev.timeout_ticks = (long long)(te ? atof(te) : 20000.0) * 100000LL;
  p->dev.delay_ticks = remote_delay_ticks();
  p->inbox_bytes = kOffHalo + 2 * (size_t)p->dev.cap * kMaxWidth * sizeof(double);
  Meta mine;
  memset(&mine, 0, sizeof mine);
  mine.cap = p->dev.cap;
  for (int q = 0; q <= n; q++) mine.halo_off[q] = halo_off[q];
  // local part; any failure is reported through mine.ok so that the ranks can agree to fall back together
  bool ok = hipExtMallocWithFlags((void**)&p->dev.inbox, p->inbox_bytes, hipDeviceMallocFinegrained) == hipSuccess;
  ok = ok && hipMemsetAsync(p->dev.inbox, 0, p->inbox_bytes, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  ok = ok && hipMalloc((void**)&p->probe, 2 * sizeof(double)) == hipSuccess;
  ok = ok && hipIpcGetMemHandle(&mine.handle, p->dev.inbox) == hipSuccess;
  (void)hipGetLastError();
  mine.ok = ok ? 1 : 0;
  std::vector<Meta> all(n);
  int rc = comm_allgather_bytes(c, &mine, all.data(), sizeof(Meta), s);
  if (rc != FB_OK) { p2p_detach(p); return rc; }
  bool all_ok = true;
  for (int q = 0; q < n; q++) all_ok = all_ok && all[q].ok;
  int opened_ok = 1;
  if (all_ok) {
    for (int q = 0; q < n; q++) {
      p->dev.peer_cap[q] = all[q].cap;
      p->dev.peer_seg[q] = all[q].halo_off[me];
      if (q == me) { p->dev.peer[q] = p->dev.inbox; continue; }
      void* ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, all[q].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened_ok = 0; break; }
      p->opened[q] = ptr;
      p->dev.peer[q] = (char*)ptr;
    }
  }
  // second agreement round: did everybody map everybody?
  std::vector<int> oks(n);
  rc = comm_allgather_bytes(c, &opened_ok, oks.data(), sizeof(int), s);
  if (rc != FB_OK) { p2p_detach(p); return rc; }
  for (int q = 0; q < n; q++) all_ok = all_ok && oks[q];
  if (!all_ok) { p2p_detach(p); return FB_OK; }  // fall back to the collective library, on every rank alike
  // handshake through the transport itself: the sum of ones must come back as n_ranks
  const double one = 1.0;
  bool good = hipMemcpyAsync(p->probe, &one, sizeof one, hipMemcpyHostToDevice, s) == hipSuccess;
  good = good && p2p_reduce(p, p->probe, 1, 1, p->probe + 1, s) == FB_OK;
  double got = 0.0;
  good = good && hipMemcpyAsync(&got, p->probe + 1, sizeof got, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  good = good && p2p_check(p, s) == FB_OK && got == (double)n;
  int g = good ? 1 : 0;
  rc = comm_allgather_bytes(c, &g, oks.data(), sizeof(int), s);
  if (rc != FB_OK) { p2p_detach(p); return rc; }
  for (int q = 0; q < n; q++) good = good && oks[q];
  if (!good) { p2p_detach(p); return FB_OK; }
  *out = p;
  return FB_OK;
}

}  // namespace fb
