// RCCL binding by dlopen (see comm.h).  If a librccl is already mapped into the process (e.g. the launcher's
// torch.distributed loaded torch/lib/librccl.so) that copy is reused, so one process never runs two RCCLs.
#include "comm.h"

#include <dlfcn.h>

#include "common.h"

namespace {

// the subset of rccl.h we call, with the ABI of /opt/rocm/include/rccl/rccl.h (ROCm 7.2)
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclSum = 0 };
enum { ncclFloat64 = 8 };

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return FB_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {  // prefer a copy that is already loaded
    lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (lib) break;
  }
  if (!lib) {
    if (const char* env = getenv("FEMBRAIN_RCCL")) lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (const char* n : names) {
      if (lib) break;
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
  }
  if (!lib) return fb::fail(FB_ECOMM, "librccl not found (%s)", dlerror());
#define SYM(field, name)                                                        \
  *(void**)(&g_rccl.field) = dlsym(lib, name);                                  \
  if (!g_rccl.field) return fb::fail(FB_ECOMM, "librccl lacks symbol %s", name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.lib = lib;
  return FB_OK;
}

#define FB_NCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) return fb::fail(FB_ECOMM, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); \
  } while (0)

}  // namespace

extern "C" {

int fb_comm_unique_id(unsigned char id[128]) {
  if (!id) return fb::fail(FB_EINVAL, "null id buffer");
  FB_TRY(load_rccl());
  ncclUniqueId u;
  FB_NCCL(g_rccl.GetUniqueId(&u));
  memcpy(id, u.internal, 128);
  return FB_OK;
}

int fb_comm_create(fb_comm_t* out, int rank, int n_ranks, const unsigned char id[128], int device) {
  if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fb::fail(FB_EINVAL, "bad communicator arguments");
  fb_comm_s* c = new fb_comm_s;
  c->rank = rank; c->n_ranks = n_ranks; c->device = device;
  if (n_ranks > 1 && !id) { delete c; return fb::fail(FB_EINVAL, "null unique id"); }
  if (id) {  // a unique id with n_ranks == 1 builds a real one-rank communicator (used to test the RCCL plumbing on one GPU)
    int r = load_rccl();
    if (r != FB_OK) { delete c; return r; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete c; return fb::fail(FB_EDEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e)); }
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    ncclComm_t comm = nullptr;
    ncclResult_t nr = g_rccl.CommInitRank(&comm, n_ranks, u, rank);
    if (nr != ncclSuccess) { delete c; return fb::fail(FB_ECOMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(nr)); }
    c->nccl = comm;
  }
  *out = c;
  return FB_OK;
}

int fb_comm_destroy(fb_comm_t c) {
  if (!c) return FB_OK;
  if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->nccl);
  delete c;
  return FB_OK;
}

}  // extern "C"

namespace fb {

int comm_allreduce_sum(fb_comm_s* c, double* dev_buf, int count, hipStream_t s) {
  if (!c || !c->nccl) return FB_OK;
  FB_NCCL(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)c->nccl, s));
  return FB_OK;
}

int comm_exchange_nodes(fb_comm_s* c, const double* sendbuf, const int* send_off, double* recv_base, const int* recv_off, int width,
                        hipStream_t s) {
  if (!c || !c->nccl || c->n_ranks == 1) return FB_OK;
  FB_NCCL(g_rccl.GroupStart());
  for (int q = 0; q < c->n_ranks; q++) {
    if (q == c->rank) continue;
    const int ns = send_off[q + 1] - send_off[q], nr = recv_off[q + 1] - recv_off[q];
    if (ns > 0) FB_NCCL(g_rccl.Send(sendbuf + (size_t)width * send_off[q], (size_t)width * ns, ncclFloat64, q, (ncclComm_t)c->nccl, s));
    if (nr > 0) FB_NCCL(g_rccl.Recv(recv_base + (size_t)width * recv_off[q], (size_t)width * nr, ncclFloat64, q, (ncclComm_t)c->nccl, s));
  }
  FB_NCCL(g_rccl.GroupEnd());
  return FB_OK;
}

}  // namespace fb
