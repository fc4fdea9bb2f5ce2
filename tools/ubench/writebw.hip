// Development micro-benchmark: streaming write rate (float4 per lane) for 268 MB and 630 MB. Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void wr(float4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ __launch_bounds__(256) void wr_nt(float4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 v = {(float)i, 1.f, 2.f, 3.f};
  if (i < n) __builtin_nontemporal_store(v, (f4*)p + i);
}
int main() {
  for (size_t mb : {268u, 630u}) {
    size_t bytes = mb << 20, n = bytes / 16;
    float4* p; hipMalloc(&p, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int v = 0; v < 2; v++) {
      for (int w = 0; w < 3; w++) { if (v) wr_nt<<<(n + 255) / 256, 256>>>(p, n); else wr<<<(n + 255) / 256, 256>>>(p, n); }
      hipEventRecord(a);
      for (int r = 0; r < 20; r++) { if (v) wr_nt<<<(n + 255) / 256, 256>>>(p, n); else wr<<<(n + 255) / 256, 256>>>(p, n); }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%s %zu MB: %.1f us -> %.0f GB/s\n", v ? "nontemporal" : "plain", mb, ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e9);
    }
    hipFree(p);
  }
  return 0;
}
