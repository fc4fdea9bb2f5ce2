// RCCL binding by dlopen (see comm.h).  If a librccl is already mapped into the process (e.g. the launcher's
// torch.distributed loaded torch/lib/librccl.so) that copy is reused, so one process never runs two RCCLs.
#include "comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>

#include "common.h"
#include "../../include/fembrain_hip_testing.h"

namespace {

// the subset of rccl.h we call, with the ABI of /opt/rocm/include/rccl/rccl.h (ROCm 7.2)
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclSum = 0 };
enum { ncclFloat64 = 8 };
enum { ncclUint8 = 1 };

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
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
  SYM(CommCount, "ncclCommCount");
  SYM(AllReduce, "ncclAllReduce");
  SYM(AllGather, "ncclAllGather");
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

// ---- host-staged shared-memory transport (TEST HOOK) -----------------------------------------------------------
// Lets several processes that share ONE GPU run the sharded solver against each other: every collective synchronises
// the stream, stages through a POSIX shared-memory segment and meets the peers at a process barrier.  Same call
// sequence and data movement as the RCCL path (fixed rank-order sums, owner-grouped halo segments), none of its speed.
constexpr int kLocalMaxRanks = 8;
constexpr unsigned int kLocalMagic = 0xFB10CA1u;
struct LocalShm {
  std::atomic<unsigned int> magic;  // set last by rank 0: the segment is initialised
  std::atomic<int> poison;          // a rank failed or gave up waiting, or rank 0 of a LATER run found this segment: leave
  std::atomic<int> count;
  std::atomic<int> sense;
  double scal[kLocalMaxRanks][8];
  int send_off[kLocalMaxRanks][kLocalMaxRanks + 1];
  char meta[kLocalMaxRanks][512];
  size_t outbox_bytes;
  // followed by n_ranks outboxes of outbox_bytes each
};
struct LocalComm {
  LocalShm* shm = nullptr;
  size_t map_bytes = 0;
  int local_sense = 0;
  double timeout_s = 20.0;
  std::string name;
  char* outbox(int r) const { return (char*)(shm + 1) + (size_t)r * shm->outbox_bytes; }
};

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// sense-reversing barrier with a wall-clock bound: a peer that left (or a segment a later run has taken over) ends the wait
// with FB_ECOMM and poisons the segment, so every other rank leaves its own wait at once instead of spinning for ever
int local_barrier(LocalComm* L, int n_ranks) {
  if (L->shm->poison.load()) return fb::fail(FB_ECOMM, "local transport: a peer rank failed (segment %s is poisoned)", L->name.c_str());
  L->local_sense ^= 1;
  if (L->shm->count.fetch_add(1) == n_ranks - 1) {
    L->shm->count.store(0);
    L->shm->sense.store(L->local_sense);
    return FB_OK;
  }
  const double deadline = now_s() + L->timeout_s;
  while (L->shm->sense.load() != L->local_sense) {
    if (L->shm->poison.load()) return fb::fail(FB_ECOMM, "local transport: a peer rank failed (segment %s is poisoned)", L->name.c_str());
    if (now_s() > deadline) {
      L->shm->poison.store(1);
      return fb::fail(FB_ECOMM, "local transport: timed out after %.0f s waiting for the peer ranks", L->timeout_s);
    }
    usleep(20);
  }
  return FB_OK;
}

// an error between two barriers: tell the peers before leaving
int local_fail(LocalComm* L, int code) {
  L->shm->poison.store(1);
  return code;
}

int local_allreduce(fb_comm_s* c, double* dev_buf, int count, hipStream_t s) {
  LocalComm* L = (LocalComm*)c->local;
  if (count > 8) return fb::fail(FB_EINVAL, "local transport: at most 8 scalars");
  double v[8];
  FB_HIP(hipMemcpyAsync(v, dev_buf, sizeof(double) * count, hipMemcpyDeviceToHost, s));
  FB_HIP(hipStreamSynchronize(s));
  memcpy(L->shm->scal[c->rank], v, sizeof(double) * count);
  FB_TRY(local_barrier(L, c->n_ranks));
  for (int k = 0; k < count; k++) {
    double t = 0.0;
    for (int r = 0; r < c->n_ranks; r++) t += L->shm->scal[r][k];  // fixed rank order: identical on every rank
    v[k] = t;
  }
  FB_TRY(local_barrier(L, c->n_ranks));  // everyone has read before the slots are reused
  FB_HIP(hipMemcpyAsync(dev_buf, v, sizeof(double) * count, hipMemcpyHostToDevice, s));
  FB_HIP(hipStreamSynchronize(s));
  return FB_OK;
}

int local_exchange(fb_comm_s* c, const double* sendbuf, const int* send_off, double* recv_base, const int* recv_off, int width, hipStream_t s) {
  LocalComm* L = (LocalComm*)c->local;
  const int n = c->n_ranks, me = c->rank;
  const size_t bytes = sizeof(double) * (size_t)width * send_off[n];
  if (bytes > L->shm->outbox_bytes) return local_fail(L, fb::fail(FB_EINVAL, "local transport: outbox too small (%zu > %zu)", bytes, L->shm->outbox_bytes));
  if (bytes && hipMemcpyAsync(L->outbox(me), sendbuf, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return local_fail(L, fb::fail(FB_EDEVICE, "local transport: copy to the outbox failed"));
  if (hipStreamSynchronize(s) != hipSuccess) return local_fail(L, fb::fail(FB_EDEVICE, "local transport: stream error"));
  memcpy(L->shm->send_off[me], send_off, sizeof(int) * (n + 1));
  FB_TRY(local_barrier(L, n));
  for (int q = 0; q < n; q++) {
    if (q == me) continue;
    const int nr = recv_off[q + 1] - recv_off[q];
    if (nr <= 0) continue;
    const int* so = L->shm->send_off[q];
    if (so[me + 1] - so[me] != nr)
      return local_fail(L, fb::fail(FB_ECOMM, "local transport: rank %d sends %d nodes to rank %d, which expects %d", q, so[me + 1] - so[me], me, nr));
    if (hipMemcpyAsync(recv_base + (size_t)width * recv_off[q], L->outbox(q) + sizeof(double) * (size_t)width * so[me], sizeof(double) * (size_t)width * nr,
                       hipMemcpyHostToDevice, s) != hipSuccess)
      return local_fail(L, fb::fail(FB_EDEVICE, "local transport: copy from the outbox failed"));
  }
  if (hipStreamSynchronize(s) != hipSuccess) return local_fail(L, fb::fail(FB_EDEVICE, "local transport: stream error"));
  return local_barrier(L, n);  // outboxes may be overwritten again
}

int local_allgather(fb_comm_s* c, const void* mine, void* all, size_t bytes) {
  LocalComm* L = (LocalComm*)c->local;
  if (bytes > sizeof L->shm->meta[0]) return fb::fail(FB_EINVAL, "local transport: all-gather item too large");
  memcpy(L->shm->meta[c->rank], mine, bytes);
  FB_TRY(local_barrier(L, c->n_ranks));
  for (int r = 0; r < c->n_ranks; r++) memcpy((char*)all + (size_t)r * bytes, L->shm->meta[r], bytes);
  return local_barrier(L, c->n_ranks);
}

bool env_flag(const char* name, bool dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) != 0 : dflt;
}

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
    c->want_p2p = env_flag("FEMBRAIN_P2P", true);  // direct xGMI mailboxes for the per-iteration exchanges (comm.h)
  }
  *out = c;
  return FB_OK;
}

int fb_comm_create_local(fb_comm_t* out, int rank, int n_ranks, const char* shm_name, size_t outbox_bytes, int device) {
  if (!out || !shm_name || n_ranks < 1 || n_ranks > kLocalMaxRanks || rank < 0 || rank >= n_ranks) return fb::fail(FB_EINVAL, "bad local communicator arguments");
  const size_t total = sizeof(LocalShm) + (size_t)n_ranks * outbox_bytes;
  const char* te = getenv("FEMBRAIN_LOCAL_TIMEOUT_MS");
  const double timeout_s = te ? std::max(0.1, atof(te) * 1e-3) : 20.0;
  LocalShm* shm = nullptr;
  if (rank == 0) {
    // a segment of this name left by a killed run carries stale barrier state: poison it (a late rank of THIS run that has
    // already attached to it then comes back for the new one), unlink it, and create the segment afresh
    int old = shm_open(shm_name, O_RDWR, 0600);
    if (old >= 0) {
      struct stat st;
      if (fstat(old, &st) == 0 && (size_t)st.st_size >= sizeof(LocalShm)) {
        void* q = mmap(nullptr, sizeof(LocalShm), PROT_READ | PROT_WRITE, MAP_SHARED, old, 0);
        if (q != MAP_FAILED) { ((LocalShm*)q)->poison.store(1); munmap(q, sizeof(LocalShm)); }
      }
      close(old);
      shm_unlink(shm_name);
    }
    int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fb::fail(FB_ECOMM, "shm_open(%s, O_EXCL) failed", shm_name);
    if (ftruncate(fd, (off_t)total) != 0) { close(fd); shm_unlink(shm_name); return fb::fail(FB_ECOMM, "ftruncate(%s) failed", shm_name); }
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(shm_name); return fb::fail(FB_ECOMM, "mmap(%s) failed", shm_name); }
    shm = (LocalShm*)p;  // zero-filled: count = sense = poison = 0
    shm->outbox_bytes = outbox_bytes;
    shm->magic.store(kLocalMagic);
  }
  LocalComm* L = new LocalComm;
  L->map_bytes = total; L->name = shm_name; L->timeout_s = timeout_s;
  const double deadline = now_s() + 3.0 * timeout_s;
  for (;;) {  // the other ranks attach to the segment rank 0 made and meet it at a first barrier
    if (rank != 0) {
      shm = nullptr;
      int fd = shm_open(shm_name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= total) {
          void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
          if (p != MAP_FAILED) shm = (LocalShm*)p;
        }
        close(fd);
      }
      if (shm && (shm->magic.load() != kLocalMagic || shm->poison.load() || shm->outbox_bytes != outbox_bytes)) { munmap(shm, total); shm = nullptr; }
      if (!shm) {
        if (now_s() > deadline) { delete L; return fb::fail(FB_ECOMM, "local transport: rank 0 did not create %s in time", shm_name); }
        usleep(2000);
        continue;
      }
    }
    L->shm = shm;
    L->local_sense = 0;
    if (local_barrier(L, n_ranks) == FB_OK) break;
    // rank 0: the peers did not come.  Others: this was a stale segment that rank 0 has poisoned meanwhile -- look again.
    if (rank == 0 || now_s() > deadline) {
      munmap(shm, total);
      if (rank == 0) shm_unlink(shm_name);
      delete L;
      return FB_ECOMM;
    }
    munmap(shm, total);
  }
  fb_comm_s* c = new fb_comm_s;
  c->rank = rank; c->n_ranks = n_ranks; c->device = device; c->local = L;
  c->want_p2p = env_flag("FEMBRAIN_P2P", false);  // the test transport stays host-staged unless asked
  *out = c;
  return FB_OK;
}

int fb_comm_test_allgather(fb_comm_t c, const void* mine, void* all, size_t bytes) {
  if (!c || !c->local || !mine || !all) return fb::fail(FB_EINVAL, "fb_comm_test_allgather needs a host-staged communicator");
  return local_allgather(c, mine, all, bytes);
}

int fb_comm_info(fb_comm_t c, int* ranks, int* rccl_ranks, int* transport) {
  if (!c) return fb::fail(FB_EINVAL, "null communicator");
  int count = 0;
  if (c->nccl) FB_NCCL(g_rccl.CommCount((ncclComm_t)c->nccl, &count));
  if (ranks) *ranks = c->n_ranks;
  if (rccl_ranks) *rccl_ranks = count;
  if (transport) *transport = c->nccl ? 1 : (c->local ? 2 : 0);
  return FB_OK;
}

int fb_comm_destroy(fb_comm_t c) {
  if (!c) return FB_OK;
  if (c->local) {
    LocalComm* L = (LocalComm*)c->local;
    munmap(L->shm, L->map_bytes);
    if (c->rank == 0) shm_unlink(L->name.c_str());
    delete L;
  }
  if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->nccl);
  delete c;
  return FB_OK;
}

}  // extern "C"

namespace fb {

int comm_allreduce_sum(fb_comm_s* c, double* dev_buf, int count, hipStream_t s) {
  if (c && c->local) return local_allreduce(c, dev_buf, count, s);
  if (!c || !c->nccl) return FB_OK;
  FB_NCCL(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)c->nccl, s));
  return FB_OK;
}

int comm_allgather_bytes(fb_comm_s* c, const void* mine, void* all, size_t bytes, hipStream_t s) {
  if (!c || c->n_ranks == 1) { memcpy(all, mine, bytes); return FB_OK; }
  if (c->local) return local_allgather(c, mine, all, bytes);
  if (!c->nccl) return fail(FB_ECOMM, "communicator has no transport");
  DevBuf<unsigned char> in, out;
  FB_TRY(in.upload((const unsigned char*)mine, bytes, s));
  FB_TRY(out.alloc(bytes * (size_t)c->n_ranks));
  FB_NCCL(g_rccl.AllGather(in.p, out.p, bytes, ncclUint8, (ncclComm_t)c->nccl, s));
  return out.download((unsigned char*)all, bytes * (size_t)c->n_ranks, s);
}

int comm_exchange_nodes(fb_comm_s* c, const double* sendbuf, const int* send_off, double* recv_base, const int* recv_off, int width,
                        hipStream_t s) {
  if (c && c->local) return c->n_ranks == 1 ? FB_OK : local_exchange(c, sendbuf, send_off, recv_base, recv_off, width, s);
  if (!c || !c->nccl || c->n_ranks == 1) return FB_OK;
  FB_NCCL(g_rccl.GroupStart());
  ncclResult_t bad = ncclSuccess;  // an error inside the group must still close it
  const char* what = "";
  for (int q = 0; q < c->n_ranks && bad == ncclSuccess; q++) {
    if (q == c->rank) continue;
    const int ns = send_off[q + 1] - send_off[q], nr = recv_off[q + 1] - recv_off[q];
    if (ns > 0) { bad = g_rccl.Send(sendbuf + (size_t)width * send_off[q], (size_t)width * ns, ncclFloat64, q, (ncclComm_t)c->nccl, s); what = "ncclSend"; }
    if (nr > 0 && bad == ncclSuccess) { bad = g_rccl.Recv(recv_base + (size_t)width * recv_off[q], (size_t)width * nr, ncclFloat64, q, (ncclComm_t)c->nccl, s); what = "ncclRecv"; }
  }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (bad != ncclSuccess) return fail(FB_ECOMM, "%s failed: %s", what, g_rccl.GetErrorString(bad));
  if (end != ncclSuccess) return fail(FB_ECOMM, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(end));
  return FB_OK;
}

}  // namespace fb
