// Shared helpers for the flowsci HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flowsci_hip.h"

#define FS_REQUIRE_PTR(p) \
  do {                    \
    if ((p) == nullptr) return FS_ERR_NULLPTR; \
  } while (0)

// hipGetLastError() reports the last error of ANY earlier runtime call of this host thread --
// e.g. the hipErrorNotReady PyTorch's allocator gets from hipEventQuery -- so every entry point
// clears the slot first and FS_LAUNCH_CHECK then sees only its own launches.
#define FS_ENTER() (void)hipGetLastError()

#define FS_LAUNCH_CHECK()                                \
  do {                                                   \
    if (hipGetLastError() != hipSuccess) return FS_ERR_LAUNCH; \
  } while (0)

namespace fs {

constexpr int kWave = 64;

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// torch.linspace(-1, 1, n)[i] in fp32 (ATen RangeFactories: symmetric fill from both ends,
// step = (end - start) / (n - 1)).  Used by the RIFE warps, whose grids are linspaces
// (Flow-2D/model/warplayer.py:12-15, Flow-3D/model/warplayer.py:15-20).
__device__ __forceinline__ float linspace_pm1(int i, int n, float step) {
#pragma clang fp contract(off)
  return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// ATen clip_coordinates_set_grad (border padding): clamp to [0, size-1]; the gradient
// multiplier is 0 on and outside the border.
__device__ __forceinline__ float clip_border(float x, int size, float* gmul) {
  const float hi = (float)(size - 1);
  if (x <= 0.0f) { *gmul = 0.0f; return 0.0f; }
  if (x >= hi)   { *gmul = 0.0f; return hi; }
  *gmul = 1.0f;
  return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Second stage of the deterministic reductions: `nblocks` pairs of per-block partial sums in
// `ws` -> sums[0..1].  One block, fixed order, fp64 accumulation: bitwise reproducible.
static __global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ ws, int nblocks,
                                                           float* __restrict__ sums) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += (double)ws[2 * i]; b += (double)ws[2 * i + 1]; }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[0] = (float)red[0][0]; sums[1] = (float)red[1][0]; }
}


// block-level (256 threads) pair reduction into ws[2*blockIdx.x .. +1]
__device__ __forceinline__ void block_pair_to_ws(float s1, float s2, float* __restrict__ ws) {
  __shared__ float red_[2][4];
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red_[0][wv] = s1; red_[1][wv] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[2 * blockIdx.x] = (red_[0][0] + red_[0][1]) + (red_[0][2] + red_[0][3]);
    ws[2 * blockIdx.x + 1] = (red_[1][0] + red_[1][1]) + (red_[1][2] + red_[1][3]);
  }
}

}  // namespace fs
