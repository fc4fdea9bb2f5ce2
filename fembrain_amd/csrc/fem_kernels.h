// Device kernels of the FEM hot path (gfx950).  One wavefront lane owns one 3x3-block row (= mesh node);
// a wavefront owns a SELL-64 slice; workgroups are dealt to XCDs so that each XCD keeps one contiguous slab of
// rows (its x-vector gathers and per-tet records stay in that XCD's 4 MiB L2).
//
// Layout in HBM (MT = matrix storage type, float by default):
//   vals   [n_slots][9][64] MT   value v = 3*r + c of the 3x3 block of (lane's row, slot) -- every load is one
//                                 256-B (fp32) coalesced wavefront segment
//   colidx [n_slots][64]    int  local column node
//   vectors fp64, node-major xyz (3 doubles per node), owned nodes first then halo
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fb {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kMaxPartials = 1024;

// PCG scalars, device resident.  rho is double-buffered by iteration parity so that the single writer
// (block 0 of the direction-update kernel) never races the readers of the same launch.
struct CGState {
  double rho[2];
  double rho0;
  double eps2;
  int iter;
  int done;
  int max_iter;
  int pad;
};

struct SellView {
  const int* slice_off;  // n_slices+1
  const int* colidx;     // n_slots*64
  const short* coldelta; // n_slots*64: 16-bit column words where they all fit (k_spmv<..., C16>), else null: column - row (C16 = 1) or,
                         // on a sharded handle, the halo form (C16 = 2, plan_device.hip k_plan_sell)
  const int* halo_base;  // n_slices: the slice's lowest halo column (C16 = 2)
  int n_slices;
  int n_owned;
};

// ---- wave / block reductions (deterministic order) ----
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// all threads of a 256-thread block get the block sum (lds: 4 doubles)
__device__ inline double block_sum(double v, double* lds) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// fixed-order sum of n (<= kMaxPartials) per-block partials; every thread returns the same value
__device__ inline double sum_partials(const double* p, int n, double* lds) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) v += p[i];
  return block_sum(v, lds);
}

// XCD-aware slice walk: blocks b and b+8 share an XCD (round-robin dispatch), so block b serves slab b%8.
struct SliceWalk {
  int s, hi, stride;
  __device__ SliceWalk(int n_slices) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int chunk = (n_slices + 7) >> 3;
    const int lo = xcd * chunk;
    hi = min(lo + chunk, n_slices);
    s = lo + j * kWavesPerBlock + (threadIdx.x >> 6);
    stride = per * kWavesPerBlock;
  }
  __device__ bool valid() const { return s < hi; }
  __device__ void next() { s += stride; }
};

}  // namespace fb
