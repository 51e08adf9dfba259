// prelu.hip -- backward of the per-channel PReLU that follows every IFNet convolution
// (Flow-*/model/IFNet.py `conv()` / `deconv()` helpers) for gfx950.
//
// ATen's prelu_backward writes TWO full-size tensors (grad_input and the un-reduced weight-gradient
// integrand) and then reduces the second in another pass: ~24 B/element, 1.25 ms per [2,64,64^3]
// activation, 90 ms per 256^3 train step.  Here: one pass, 12 B/element (read x, read g, write gx);
// the per-channel sum  ga[c] = sum g * x * [x <= 0]  is accumulated in registers, reduced per block,
// and finished by a second tiny kernel in a fixed order (fp64): deterministic, no atomics.
#include "common.hpp"

namespace {

constexpr int PCH = FS_PRELU_MAX_CHUNKS;  // spatial chunks per (b, c) row, upper bound

__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ a,
                                                        float* __restrict__ gx, float* __restrict__ ws,
                                                        float* __restrict__ wsb, int C, int S, int nchunk,
                                                        int chunk_len, int shared_a) {
  // blockIdx.x = (b*C + c) * nchunk + chunk
  const int chunk = blockIdx.x % nchunk;
  const long long row = blockIdx.x / nchunk;
  const int c = (int)(row % C);
  const float slope = a[shared_a ? 0 : c];
  const long long base = row * (long long)S;
  const int lo = chunk * chunk_len, hi = min(lo + chunk_len, S);
  float acc = 0.f, accb = 0.f;  // accb: sum of grad_x = the bias gradient of the producing convolution
  const bool vec = ((base + lo) % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) |
                                                reinterpret_cast<uintptr_t>(gx)) % 16 == 0);
  int i = lo + threadIdx.x * 4;
  if (vec) {
    for (; i + 3 < hi; i += 256 * 4) {
      const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
      float4 o;
      o.x = xv.x > 0.f ? gv.x : slope * gv.x; acc += xv.x > 0.f ? 0.f : xv.x * gv.x;
      o.y = xv.y > 0.f ? gv.y : slope * gv.y; acc += xv.y > 0.f ? 0.f : xv.y * gv.y;
      o.z = xv.z > 0.f ? gv.z : slope * gv.z; acc += xv.z > 0.f ? 0.f : xv.z * gv.z;
      o.w = xv.w > 0.f ? gv.w : slope * gv.w; acc += xv.w > 0.f ? 0.f : xv.w * gv.w;
      accb += (o.x + o.y) + (o.z + o.w);
      *reinterpret_cast<float4*>(gx + base + i) = o;
    }
    // tail (< 4 elements of this chunk): handled by the threads whose quad straddles `hi`
    for (int j = i; j < hi && j < i + 4; ++j) {
      const float xv = x[base + j], gv = g[base + j];
      const float o = xv > 0.f ? gv : slope * gv;
      gx[base + j] = o;
      accb += o;
      acc += xv > 0.f ? 0.f : xv * gv;
    }
  } else {
    for (int j = lo + threadIdx.x; j < hi; j += 256) {
      const float xv = x[base + j], gv = g[base + j];
      const float o = xv > 0.f ? gv : slope * gv;
      gx[base + j] = o;
      accb += o;
      acc += xv > 0.f ? 0.f : xv * gv;
    }
  }
  __shared__ float red[2][4];
  acc = fs::wave_sum(acc);
  accb = fs::wave_sum(accb);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = acc; red[1][threadIdx.x >> 6] = accb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    if (wsb != nullptr) wsb[blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// Finishing pass: blocks [0, nw) reduce the slope-gradient partials (per channel, or over everything when
// the slope is shared); blocks [nw, nw + C) -- present only with a bias gradient -- reduce the per-channel
// partials of sum(grad_x).  One block per output, fixed summation order, fp64: deterministic.
__global__ __launch_bounds__(256) void prelu_ga_kernel(const float* __restrict__ ws, const float* __restrict__ wsb,
                                                       float* __restrict__ ga, float* __restrict__ gb, int B,
                                                       int C, int nchunk, int nw) {
  const bool bias_blk = (int)blockIdx.x >= nw;
  const float* src = bias_blk ? wsb : ws;
  const int c = bias_blk ? blockIdx.x - nw : blockIdx.x;
  const bool all = !bias_blk && nw == 1 && C != 1;  // shared slope: sum over every (b, c)
  __shared__ double red[256];
  double s = 0.0;
  if (all) {
    const long long n = (long long)B * C * nchunk;
    for (long long i = threadIdx.x; i < n; i += 256) s += (double)src[i];
  } else {
    const int n = B * nchunk;
    for (int i = threadIdx.x; i < n; i += 256) {
      const int b = i / nchunk, k = i - b * nchunk;
      s += (double)src[((long long)b * C + c) * nchunk + k];
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) (bias_blk ? gb : ga)[c] = (float)red[0];
}

}  // namespace

extern "C" int fs_prelu_bwd(const float* x, const float* grad_out, const float* weight, float* grad_x,
                            float* grad_weight, float* grad_bias, float* ws, int B, int C, int S,
                            int num_weights, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(x); FS_REQUIRE_PTR(grad_out); FS_REQUIRE_PTR(weight);
  FS_REQUIRE_PTR(grad_x); FS_REQUIRE_PTR(grad_weight); FS_REQUIRE_PTR(ws);
  if (B < 1 || C < 1 || S < 1) return FS_ERR_SHAPE;
  if (num_weights != 1 && num_weights != C) return FS_ERR_ARG;
  // chunks of >= 8192 elements, a multiple of 4, at most PCH per row
  int nchunk = (S + 8191) / 8192;
  if (nchunk > PCH) nchunk = PCH;
  int chunk_len = (S + nchunk - 1) / nchunk;
  chunk_len = (chunk_len + 3) / 4 * 4;
  nchunk = (S + chunk_len - 1) / chunk_len;
  const long long blocks = (long long)B * C * nchunk;
  if (blocks >= (1ll << 31)) return FS_ERR_SHAPE;
  const int shared_a = (num_weights == 1) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  float* wsb = grad_bias ? ws + (size_t)B * C * PCH : nullptr;
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, grad_out, weight,
                     grad_x, ws, wsb, C, S, nchunk, chunk_len, shared_a);
  hipLaunchKernelGGL(prelu_ga_kernel, dim3(num_weights + (grad_bias ? C : 0)), dim3(256), 0, st, ws, wsb,
                     grad_weight, grad_bias, B, C, nchunk, num_weights);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
