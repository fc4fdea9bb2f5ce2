// Development micro-benchmark: streaming read rate of a buffer that fits the 256 MiB Infinity Cache (102 MB) vs one
// that does not (1 GB), with 16-byte and 4-byte per-lane loads.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T>
__global__ __launch_bounds__(256) void rd(const T* __restrict__ p, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  float acc = 0;
  for (; i < n; i += stride) { T v = p[i]; acc += ((float*)&v)[0]; }
  if (acc == 12345.678f) out[0] = acc;
}
template <typename T>
void run(const char* name, size_t bytes, int grid) {
  T* p; float* o;
  hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 0, bytes);
  size_t n = bytes / sizeof(T);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; w++) rd<T><<<grid, 256>>>(p, n, o);
  hipEventRecord(a);
  const int reps = 50;
  for (int r = 0; r < reps; r++) rd<T><<<grid, 256>>>(p, n, o);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%s %zu MB grid %d: %.2f us -> %.0f GB/s\n", name, bytes >> 20, grid, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
  hipFree(p); hipFree(o);
}
int main() {
  for (int grid : {1024, 2048, 4096, 8192}) {
    run<float4>("float4", 102u << 20, grid);
    run<float>("float ", 102u << 20, grid);
  }
  run<float4>("float4", 1024u << 20, 4096);
  run<float>("float ", 1024u << 20, 4096);
  run<float4>("float4", 24u << 20, 2048);
  return 0;
}
