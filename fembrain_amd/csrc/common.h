// Shared host-side helpers of libfembrain_hip.so (error text, device buffers, timing).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fembrain_hip.h"

namespace fb {

std::string& last_error();
int fail(int code, const char* fmt, ...);

#define FB_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return fb::fail(FB_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define FB_TRY(expr)            \
  do {                          \
    int _r = (expr);            \
    if (_r != FB_OK) return _r; \
  } while (0)

// Slack of a fresh allocation in sixteenths of the size asked for (2 = an eighth, the default; 4 = a quarter for a handle whose caller
// has said it will cut: fb_fem_params.expect_cuts).  Set for the duration of a build by the entry point that runs it (SlackScope).
inline int& alloc_slack_sixteenths() {
  static thread_local int v = 2;
  return v;
}
struct SlackScope {
  int before;
  explicit SlackScope(int sixteenths) : before(alloc_slack_sixteenths()) { alloc_slack_sixteenths() = sixteenths; }
  ~SlackScope() { alloc_slack_sixteenths() = before; }
};

// device array owned by a handle
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;    // elements in use
  size_t cap = 0;  // elements allocated (>= n)
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    cap = 0;
  }
  // An allocation is kept while the new size fits it and fills at least half of it, and a new one gets an eighth of slack: a re-sync
  // after a cut changes every size a little (nodes and elements are added), and hipFree + hipMalloc of ~40 buffers of up to 100 MB
  // cost 7 of the 10 ms such a re-sync took at 1M tets (tools/probe_resync_cut.py) when every size was exact.
  int alloc(size_t count) {
    if (p && count <= cap && (count >= cap / 2 || cap * sizeof(T) <= (1u << 20)) && count > 0) { n = count; return FB_OK; }
    const size_t had = p ? cap : 0;
    release();
    if (count == 0) return FB_OK;
    // (a buffer that is GROWING gets half again what it had, so a mesh that grows cut by cut re-allocates every few cuts only)
    const size_t grown = had && count > had ? had + had / 2 : 0;
    const size_t want = std::max(count + (count * sizeof(T) >= (1u << 16) ? count / 16 * (size_t)alloc_slack_sixteenths() : 0), count * sizeof(T) >= (1u << 16) ? grown : 0);
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e != hipSuccess && want != count) {  // no room for the slack: the exact size
      (void)hipGetLastError();
      e = hipMalloc((void**)&p, count * sizeof(T));
      if (e == hipSuccess) { n = cap = count; return FB_OK; }
    }
    if (e != hipSuccess) return fail(FB_ENOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    n = count;
    cap = want;
    return FB_OK;
  }
  // grow-only: keeps a larger allocation (workspaces that are reused with varying sizes)
  int reserve(size_t count) {
    if (p && n >= count) return FB_OK;
    if (p && cap >= count) { n = count; return FB_OK; }
    return alloc(count);
  }
  void swap(DevBuf& o) {
    T* tp = p; p = o.p; o.p = tp;
    size_t t = n; n = o.n; o.n = t;
    t = cap; cap = o.cap; o.cap = t;
  }
  int zero(hipStream_t s) {
    if (n) FB_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
    return FB_OK;
  }
  int upload(const T* host, size_t count, hipStream_t s) {
    FB_TRY(alloc(count));
    if (count) {
      FB_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
      FB_HIP(hipStreamSynchronize(s));
    }
    return FB_OK;
  }
  int upload(const std::vector<T>& v, hipStream_t s) { return upload(v.data(), v.size(), s); }
  int download(T* host, size_t count, hipStream_t s, size_t offset = 0) const {
    if (count) {
      FB_HIP(hipMemcpyAsync(host, p + offset, count * sizeof(T), hipMemcpyDeviceToHost, s));
      FB_HIP(hipStreamSynchronize(s));
    }
    return FB_OK;
  }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// the tet mesh a polygonizer handle holds on its device after fb_poly_tetrahedralize (poly.hip -> fem.hip hand-off):
// positions 3 floats per vertex, elements 4 vertex ids per tet; the handle's stream has been synchronised
// (xyz64: the positions as doubles instead -- the handle's own device copy, fb_fem_resync_delta)
struct DeviceTetMesh { int device, n_vertices, n_tets; const float* xyz; const uint4* tets; const double* xyz64 = nullptr; };
int poly_device_tetmesh(fb_poly_t h, DeviceTetMesh* out);

}  // namespace fb
