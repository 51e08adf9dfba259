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

// Measurement switches.  The shipped library reads NO environment variable: every dispatch decision is a function of
// the call's arguments.  A/B and ablation runs use a second build of the same sources (`make ablation`,
// -DFS_ABLATION -> ablation/libflowsci_hip_ab.so, loaded only through FLOWSCI_HIP_LIBRARY=<path>), in which the
// superseded kernels are compiled in and FLOWSCI_* variables (latched once per process) select them.
#ifdef FS_ABLATION
#include <stdlib.h>
#define FS_AB_ENV(name) (getenv(name) != nullptr)
#define FS_AB_ENV_LL(name, dflt) (getenv(name) ? atoll(getenv(name)) : (long long)(dflt))
#else
#define FS_AB_ENV(name) false
#define FS_AB_ENV_LL(name, dflt) ((long long)(dflt))
#endif

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

// One output row segment of ATen's upsample_trilinear3d (align_corners=False, scale factor given): the N
// consecutive outputs x_first .. x_first+N-1 of the row whose z / y source rows and lambdas the caller has
// resolved (s00 = row (z0, y0), s01 = (z0, y0+yp), s10 = (z0+zp, y0), s11 = (z0+zp, y0+yp) of the small
// volume, Wi its row length, rs = 1 / scale_factor), times `scale`.  ATen's index / lambda arithmetic and
// summation order, FMA contraction off: shared by fs_upsample3d_scale_add and the fused up-sample + warp
// kernel so that both write bit-identical flows.
template <int N>
__device__ __forceinline__ void trilinear_up_row(const float* __restrict__ s00, const float* __restrict__ s01,
                                                 const float* __restrict__ s10, const float* __restrict__ s11,
                                                 float lz0, float lz1, float ly0, float ly1, float rs, int x_first,
                                                 int Wi, float scale, float (&o)[N]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int x = x_first + i;
    float sx = rs * ((float)x + 0.5f) - 0.5f;
    sx = sx < 0.f ? 0.f : sx;
    const int x0 = (int)sx;
    const int xp = (x0 < Wi - 1) ? 1 : 0;
    const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
    const float v = lz0 * (ly0 * (lx0 * s00[x0] + lx1 * s00[x0 + xp]) + ly1 * (lx0 * s01[x0] + lx1 * s01[x0 + xp])) +
                    lz1 * (ly0 * (lx0 * s10[x0] + lx1 * s10[x0 + xp]) + ly1 * (lx0 * s11[x0] + lx1 * s11[x0 + xp]));
    o[i] = v * scale;
  }
}

// source index i0, "+1 exists" flag and lambdas of one output index along one axis (same arithmetic)
__device__ __forceinline__ void trilinear_axis(int o, int n_in, float rs, int& i0, int& ip, float& l0, float& l1) {
#pragma clang fp contract(off)
  float s = rs * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  ip = (i0 < n_in - 1) ? 1 : 0;
  l1 = s - (float)i0;
  l0 = 1.f - l1;
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
