// Shared helpers for the flowsci HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flowsci_hip.h"

#define FS_REQUIRE_PTR(p) \
  do {                    \
    if ((p) == nullptr) return FS_ERR_NULLPTR; \
  } while (0)

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

}  // namespace fs
