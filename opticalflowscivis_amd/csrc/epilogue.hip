// epilogue.hip -- merge + distillation epilogues of IFNet for gfx950.
//
//   merged = w0 * sigmoid(m) + w1 * (1 - sigmoid(m))              Flow-*/model/IFNet.py "merged[i] = ..."
//   loss_mask = mean_c|merged - gt| > mean_c|merged_tea - gt| + 0.01   (detached)
//   distill  += mean( sqrt(mean_c (flow_tea - flow)^2) * loss_mask )
//                                     Flow-2D/model/IFNet.py:239-248, Flow-3D/model/IFNet.py:241-267
//
// In eager PyTorch each line is 3-10 elementwise passes over full-resolution tensors (at 256^3 a
// [2,6,256^3] flow is 805 MB).  Here: one pass each, lanes on the fastest axis, 16-B accesses
// where the extent allows, deterministic two-stage reduction for the scalar.
#include "common.hpp"

namespace {

__device__ __forceinline__ float sigmoidf_(float m) { return 1.0f / (1.0f + expf(-m)); }

__global__ __launch_bounds__(256) void merge_fwd_kernel(const float* __restrict__ w0,
                                                        const float* __restrict__ w1,
                                                        const float* __restrict__ m,
                                                        float* __restrict__ merged,
                                                        float* __restrict__ sig, long long CS, int S,
                                                        long long n) {
  // w0, w1, merged: [B,C,S]; m, sig: [B,1,S]
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const long long b = e / CS;
    const long long rem = e - b * CS;
    const int r = (int)(rem % S);
    const float s = sigmoidf_(m[b * S + r]);
    merged[e] = w0[e] * s + w1[e] * (1.0f - s);
    if (sig != nullptr && rem < S) sig[b * S + r] = s;
  }
}

__global__ __launch_bounds__(256) void merge_bwd_kernel(const float* __restrict__ w0,
                                                        const float* __restrict__ w1,
                                                        const float* __restrict__ m,
                                                        const float* __restrict__ gmerged,
                                                        const float* __restrict__ gsig,
                                                        float* __restrict__ gw0, float* __restrict__ gw1,
                                                        float* __restrict__ gm, int C, int S,
                                                        long long nBS) {
  // one thread per (b, voxel): loops the C channels so that gm needs no atomics
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nBS; i += (long long)gridDim.x * 256) {
    const long long b = i / S;
    const int r = (int)(i - b * S);
    const float s = sigmoidf_(m[i]);
    float acc = gsig ? gsig[i] : 0.f;
    for (int c = 0; c < C; ++c) {
      const long long e = (b * C + c) * (long long)S + r;
      const float g = gmerged[e];
      if (gw0) gw0[e] = g * s;
      if (gw1) gw1[e] = g * (1.0f - s);
      acc += g * (w0[e] - w1[e]);
    }
    if (gm) gm[i] = acc * s * (1.0f - s);
  }
}

struct DP {
  int C, F, S;      // image channels, flow channels, spatial size
  long long nBS;    // B*S
};

__device__ __forceinline__ float distill_mask(const float* __restrict__ mi, const float* __restrict__ mt,
                                              const float* __restrict__ gt, const DP& p, long long b,
                                              int r) {
  float a = 0.f, t = 0.f;
  for (int c = 0; c < p.C; ++c) {
    const long long e = (b * p.C + c) * (long long)p.S + r;
    const float g = gt[e];
    a += fabsf(mi[e] - g);
    t += fabsf(mt[e] - g);
  }
  return (a / (float)p.C > t / (float)p.C + 0.01f) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void distill_fwd_kernel(const float* __restrict__ mi,
                                                          const float* __restrict__ mt,
                                                          const float* __restrict__ gt,
                                                          const float* __restrict__ fi,
                                                          const float* __restrict__ ft,
                                                          float* __restrict__ ws, DP p) {
  float s1 = 0.f, s2 = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.nBS; i += (long long)gridDim.x * 256) {
    const long long b = i / p.S;
    const int r = (int)(i - b * p.S);
    const float lm = distill_mask(mi, mt, gt, p, b, r);
    float q = 0.f;
    for (int c = 0; c < p.F; ++c) {
      const long long e = (b * p.F + c) * (long long)p.S + r;
      const float d = ft[e] - fi[e];
      q += d * d;
    }
    s1 += sqrtf(q / (float)p.F) * lm;
    s2 += lm;
  }
  fs::block_pair_to_ws(s1, s2, ws);
}

__global__ __launch_bounds__(256) void distill_bwd_kernel(const float* __restrict__ mi,
                                                          const float* __restrict__ mt,
                                                          const float* __restrict__ gt,
                                                          const float* __restrict__ fi,
                                                          const float* __restrict__ ft,
                                                          const float* __restrict__ coef,
                                                          float* __restrict__ gfi, DP p) {
  const float k = coef[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.nBS; i += (long long)gridDim.x * 256) {
    const long long b = i / p.S;
    const int r = (int)(i - b * p.S);
    const float lm = distill_mask(mi, mt, gt, p, b, r);
    float q = 0.f;
    for (int c = 0; c < p.F; ++c) {
      const long long e = (b * p.F + c) * (long long)p.S + r;
      const float d = ft[e] - fi[e];
      q += d * d;
    }
    const float root = sqrtf(q / (float)p.F);
    // d/df sqrt(mean_c d_c^2) = -d_c / (F * root); the reference's pow(0.5) backward is inf at
    // root == 0 (0 * inf = NaN when the mask is 0): emit 0 there instead.
    const float sc = (root > 0.f) ? (k * lm / ((float)p.F * root)) : 0.f;
    for (int c = 0; c < p.F; ++c) {
      const long long e = (b * p.F + c) * (long long)p.S + r;
      gfi[e] = -(ft[e] - fi[e]) * sc;
    }
  }
}

unsigned blocks_for(long long n, int cap) {
  const long long want = (n + 255) / 256;
  return (unsigned)(want < cap ? want : cap);
}

}  // namespace

extern "C" int fs_merge_fwd(const float* w0, const float* w1, const float* mask_logit, float* merged,
                            float* sigmoid_out, int B, int C, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(w0); FS_REQUIRE_PTR(w1); FS_REQUIRE_PTR(mask_logit); FS_REQUIRE_PTR(merged);
  if (B < 1 || C < 1 || S < 1) return FS_ERR_SHAPE;
  const long long n = (long long)B * C * S;
  hipLaunchKernelGGL(merge_fwd_kernel, dim3(blocks_for(n, 16384)), dim3(256), 0, (hipStream_t)stream,
                     w0, w1, mask_logit, merged, sigmoid_out, (long long)C * S, S, n);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_merge_bwd(const float* w0, const float* w1, const float* mask_logit,
                            const float* grad_merged, const float* grad_sigmoid, float* grad_w0,
                            float* grad_w1, float* grad_mask_logit, int B, int C, int S,
                            fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(w0); FS_REQUIRE_PTR(w1); FS_REQUIRE_PTR(mask_logit); FS_REQUIRE_PTR(grad_merged);
  if (grad_w0 == nullptr && grad_w1 == nullptr && grad_mask_logit == nullptr) return FS_ERR_NULLPTR;
  if (B < 1 || C < 1 || S < 1) return FS_ERR_SHAPE;
  const long long nBS = (long long)B * S;
  hipLaunchKernelGGL(merge_bwd_kernel, dim3(blocks_for(nBS, 16384)), dim3(256), 0, (hipStream_t)stream,
                     w0, w1, mask_logit, grad_merged, grad_sigmoid, grad_w0, grad_w1, grad_mask_logit,
                     C, S, nBS);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_distill_fwd(const float* merged_i, const float* merged_tea, const float* gt,
                              const float* flow_i, const float* flow_tea, float* sums, float* ws, int B,
                              int C, int F, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(merged_i); FS_REQUIRE_PTR(merged_tea); FS_REQUIRE_PTR(gt);
  FS_REQUIRE_PTR(flow_i); FS_REQUIRE_PTR(flow_tea); FS_REQUIRE_PTR(sums); FS_REQUIRE_PTR(ws);
  if (B < 1 || C < 1 || F < 1 || S < 1) return FS_ERR_SHAPE;
  DP p = {C, F, S, (long long)B * S};
  const unsigned nb = blocks_for(p.nBS, FS_REDUCE_BLOCKS);
  hipLaunchKernelGGL(distill_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, merged_i,
                     merged_tea, gt, flow_i, flow_tea, ws, p);
  hipLaunchKernelGGL(fs::reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, (int)nb,
                     sums);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_distill_bwd(const float* merged_i, const float* merged_tea, const float* gt,
                              const float* flow_i, const float* flow_tea, const float* coef,
                              float* grad_flow_i, int B, int C, int F, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(merged_i); FS_REQUIRE_PTR(merged_tea); FS_REQUIRE_PTR(gt);
  FS_REQUIRE_PTR(flow_i); FS_REQUIRE_PTR(flow_tea); FS_REQUIRE_PTR(coef); FS_REQUIRE_PTR(grad_flow_i);
  if (B < 1 || C < 1 || F < 1 || S < 1) return FS_ERR_SHAPE;
  DP p = {C, F, S, (long long)B * S};
  hipLaunchKernelGGL(distill_bwd_kernel, dim3(blocks_for(p.nBS, 16384)), dim3(256), 0,
                     (hipStream_t)stream, merged_i, merged_tea, gt, flow_i, flow_tea, coef, grad_flow_i,
                     p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
