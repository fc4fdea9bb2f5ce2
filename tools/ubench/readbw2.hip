// Development micro-benchmark: how fast can a 104 MB buffer (Infinity-Cache resident) and a 1 GB buffer (HBM) be read
// with U independent 16-byte (or 4-byte) loads in flight per lane, and through LDS-DMA.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int U>
__global__ __launch_bounds__(256) void rd(const T* __restrict__ p, size_t n, float* out) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) acc += ((float*)&v[u])[0];
  }
  for (; i < n; i += stride) { T v = p[i]; acc += ((float*)&v)[0]; }
  if (acc == 12345.678f) out[0] = acc;
}
// LDS-DMA: each wave streams 1 KiB pieces into its own LDS ring of R slots, reads one dword back per piece
template <int R>
__global__ __launch_bounds__(256) void rd_lds(const float4* __restrict__ p, size_t n, float* out) {
  __shared__ float4 ring[4][R][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + wv, nw = (size_t)gridDim.x * 4;
  float acc = 0;
  const size_t pieces = n / 64;
  size_t k = wave;
  int slot = 0;
  for (; k < pieces; k += nw) {
    __builtin_amdgcn_global_load_lds((const void*)(p + k * 64 + lane), (__attribute__((address_space(3))) void*)&ring[wv][slot][0], 16, 0, 0);
    slot = (slot + 1) % R;
    if (slot == 0) {
      __builtin_amdgcn_s_waitcnt(0);
      acc += ring[wv][lane % R][lane].x;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  acc += ring[wv][0][lane].x;
  if (acc == 12345.678f) out[0] = acc;
}
template <typename K>
void timeit(const char* name, size_t bytes, K launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; w++) launch();
  hipEventRecord(a);
  const int reps = 40;
  for (int r = 0; r < reps; r++) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s %5zu MB: %8.2f us -> %5.0f GB/s\n", name, bytes >> 20, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
  fflush(stdout);
}
int main() {
  float* o; hipMalloc(&o, 4);
  for (size_t bytes : {(size_t)104 << 20, (size_t)1024 << 20}) {
    void* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes);
    for (int grid : {1024, 2048, 4096}) {
      char nm[64];
      snprintf(nm, 64, "float4 U=1 grid %d", grid); timeit(nm, bytes, [&] { rd<float4, 1><<<grid, 256>>>((const float4*)p, bytes / 16, o); });
      snprintf(nm, 64, "float4 U=4 grid %d", grid); timeit(nm, bytes, [&] { rd<float4, 4><<<grid, 256>>>((const float4*)p, bytes / 16, o); });
      snprintf(nm, 64, "float4 U=8 grid %d", grid); timeit(nm, bytes, [&] { rd<float4, 8><<<grid, 256>>>((const float4*)p, bytes / 16, o); });
      snprintf(nm, 64, "float  U=8 grid %d", grid); timeit(nm, bytes, [&] { rd<float, 8><<<grid, 256>>>((const float*)p, bytes / 4, o); });
      snprintf(nm, 64, "float  U=16 grid %d", grid); timeit(nm, bytes, [&] { rd<float, 16><<<grid, 256>>>((const float*)p, bytes / 4, o); });
      snprintf(nm, 64, "lds-dma R=8 grid %d", grid); timeit(nm, bytes, [&] { rd_lds<8><<<grid, 256>>>((const float4*)p, bytes / 16, o); });
      snprintf(nm, 64, "lds-dma R=16 grid %d", grid); timeit(nm, bytes, [&] { rd_lds<16><<<grid, 256>>>((const float4*)p, bytes / 16, o); });
    }
    hipFree(p);
  }
  return 0;
}
