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
    constexpr int MAXF = 6;  // the differences stay in registers between the norm and the gradient (see distill3_bwd_kernel)
    float q = 0.f, dreg[MAXF];
    const bool keep = p.F <= MAXF;
    if (keep) {
#pragma unroll
      for (int c = 0; c < MAXF; ++c) {
        const long long e = (b * p.F + (c < p.F ? c : 0)) * (long long)p.S + r;
        const float d = ft[e] - fi[e];
        dreg[c] = d;
        if (c < p.F) q += d * d;
      }
    } else {
      for (int c = 0; c < p.F; ++c) {
        const long long e = (b * p.F + c) * (long long)p.S + r;
        const float d = ft[e] - fi[e];
        q += d * d;
      }
    }
    const float root = sqrtf(q / (float)p.F);
    // d/df sqrt(mean_c d_c^2) = -d_c / (F * root); the reference's pow(0.5) backward is inf at
    // root == 0 (0 * inf = NaN when the mask is 0): emit 0 there instead.
    // k == 0: the caller has discarded this loss term (Flow-2D's NaN / > 10 guard, RIFE.py:295-296, evaluated on
    // the device): the gradient is exactly 0 then, also where the flows themselves are not finite (0 * NaN).
    const float sc = (root > 0.f) ? (k * lm / ((float)p.F * root)) : 0.f;
    if (keep) {
#pragma unroll
      for (int c = 0; c < MAXF; ++c)
        if (c < p.F) gfi[(b * p.F + c) * (long long)p.S + r] = (k == 0.f) ? 0.f : -dreg[c] * sc;
    } else {
      for (int c = 0; c < p.F; ++c) {
        const long long e = (b * p.F + c) * (long long)p.S + r;
        gfi[e] = (k == 0.f) ? 0.f : -(ft[e] - fi[e]) * sc;
      }
    }
  }
}

// The three student terms of loss_distill (IFNet.py:259-262: one term per block, all against the same
// teacher) in one pass: the teacher's flow / merged frame and the ground truth are read once instead of three
// times (3.9 GB instead of 6.0 GB per pass at 2 x 256^3).
struct D3 { const float* mi[3]; const float* fi[3]; float* gfi[3]; };

__global__ __launch_bounds__(256) void distill3_fwd_kernel(D3 a, const float* __restrict__ mt,
                                                           const float* __restrict__ gt,
                                                           const float* __restrict__ ft, float* __restrict__ ws01,
                                                           float* __restrict__ ws2, DP p) {
  float s[3] = {0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.nBS; i += (long long)gridDim.x * 256) {
    const long long b = i / p.S;
    const int r = (int)(i - b * p.S);
    float t = 0.f, am[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < p.C; ++c) {
      const long long e = (b * p.C + c) * (long long)p.S + r;
      const float g = gt[e];
      t += fabsf(mt[e] - g);
#pragma unroll
      for (int k = 0; k < 3; ++k) am[k] += fabsf(a.mi[k][e] - g);
    }
    float q[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < p.F; ++c) {
      const long long e = (b * p.F + c) * (long long)p.S + r;
      const float f = ft[e];
#pragma unroll
      for (int k = 0; k < 3; ++k) { const float d = f - a.fi[k][e]; q[k] += d * d; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float lm = (am[k] / (float)p.C > t / (float)p.C + 0.01f) ? 1.0f : 0.0f;
      s[k] += sqrtf(q[k] / (float)p.F) * lm;
    }
  }
  fs::block_pair_to_ws(s[0], s[1], ws01);
  __syncthreads();
  fs::block_pair_to_ws(s[2], 0.f, ws2);
}

__global__ __launch_bounds__(256) void distill3_bwd_kernel(D3 a, const float* __restrict__ mt,
                                                           const float* __restrict__ gt,
                                                           const float* __restrict__ ft,
                                                           const float* __restrict__ coef, DP p) {
  const float kk = coef[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.nBS; i += (long long)gridDim.x * 256) {
    const long long b = i / p.S;
    const int r = (int)(i - b * p.S);
    float t = 0.f, am[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < p.C; ++c) {
      const long long e = (b * p.C + c) * (long long)p.S + r;
      const float g = gt[e];
      t += fabsf(mt[e] - g);
#pragma unroll
      for (int k = 0; k < 3; ++k) am[k] += fabsf(a.mi[k][e] - g);
    }
    // the flow differences stay in registers between the norm and the gradient (F <= 6: the 3-D / 2-D flow pairs): PMC
    // had this kernel fetching 1.76x its algorithmic reads when the second pass read the four flow tensors again
    constexpr int MAXF = 6;
    float q[3] = {0.f, 0.f, 0.f};
    float dreg[3][MAXF];
    const bool keep = p.F <= MAXF;
    if (keep) {
#pragma unroll
      for (int c = 0; c < MAXF; ++c) {
        const long long e = (b * p.F + (c < p.F ? c : 0)) * (long long)p.S + r;
        const float f = ft[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float d = f - a.fi[k][e];
          dreg[k][c] = d;
          if (c < p.F) q[k] += d * d;
        }
      }
    } else {
      for (int c = 0; c < p.F; ++c) {
        const long long e = (b * p.F + c) * (long long)p.S + r;
        const float f = ft[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float d = f - a.fi[k][e]; q[k] += d * d; }
      }
    }
    float sc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float lm = (am[k] / (float)p.C > t / (float)p.C + 0.01f) ? 1.0f : 0.0f;
      const float root = sqrtf(q[k] / (float)p.F);
      sc[k] = (root > 0.f) ? (kk * lm / ((float)p.F * root)) : 0.f;  // see distill_bwd_kernel
    }
    if (keep) {
#pragma unroll
      for (int c = 0; c < MAXF; ++c)
        if (c < p.F) {
          const long long e = (b * p.F + c) * (long long)p.S + r;
#pragma unroll
          for (int k = 0; k < 3; ++k) a.gfi[k][e] = (kk == 0.f) ? 0.f : -dreg[k][c] * sc[k];  // see distill_bwd_kernel
        }
    } else {
      for (int c = 0; c < p.F; ++c) {
        const long long e = (b * p.F + c) * (long long)p.S + r;
        const float f = ft[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) a.gfi[k][e] = (kk == 0.f) ? 0.f : -(f - a.fi[k][e]) * sc[k];  // see distill_bwd_kernel
      }
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

// Three student terms against one teacher in one launch each way (see distill3_fwd_kernel).  sums[0..2] = the
// three per-term sums (sums[3] scratch); ws: 4 * FS_REDUCE_BLOCKS floats.  bwd: coef = d(objective)/d(sum),
// shared by the three terms.
extern "C" int fs_distill3_fwd(const float* merged0, const float* merged1, const float* merged2,
                               const float* merged_tea, const float* gt, const float* flow0, const float* flow1,
                               const float* flow2, const float* flow_tea, float* sums, float* ws, int B, int C,
                               int F, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(merged0); FS_REQUIRE_PTR(merged1); FS_REQUIRE_PTR(merged2); FS_REQUIRE_PTR(merged_tea);
  FS_REQUIRE_PTR(gt); FS_REQUIRE_PTR(flow0); FS_REQUIRE_PTR(flow1); FS_REQUIRE_PTR(flow2);
  FS_REQUIRE_PTR(flow_tea); FS_REQUIRE_PTR(sums); FS_REQUIRE_PTR(ws);
  if (B < 1 || C < 1 || F < 1 || S < 1) return FS_ERR_SHAPE;
  DP p = {C, F, S, (long long)B * S};
  D3 a = {{merged0, merged1, merged2}, {flow0, flow1, flow2}, {nullptr, nullptr, nullptr}};
  const unsigned nb = blocks_for(p.nBS, FS_REDUCE_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(distill3_fwd_kernel, dim3(nb), dim3(256), 0, st, a, merged_tea, gt, flow_tea, ws,
                     ws + 2 * FS_REDUCE_BLOCKS, p);
  hipLaunchKernelGGL(fs::reduce_final_kernel, dim3(1), dim3(256), 0, st, ws, (int)nb, sums);
  hipLaunchKernelGGL(fs::reduce_final_kernel, dim3(1), dim3(256), 0, st, ws + 2 * FS_REDUCE_BLOCKS, (int)nb,
                     sums + 2);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_distill3_bwd(const float* merged0, const float* merged1, const float* merged2,
                               const float* merged_tea, const float* gt, const float* flow0, const float* flow1,
                               const float* flow2, const float* flow_tea, const float* coef, float* grad_flow0,
                               float* grad_flow1, float* grad_flow2, int B, int C, int F, int S,
                               fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(merged0); FS_REQUIRE_PTR(merged1); FS_REQUIRE_PTR(merged2); FS_REQUIRE_PTR(merged_tea);
  FS_REQUIRE_PTR(gt); FS_REQUIRE_PTR(flow0); FS_REQUIRE_PTR(flow1); FS_REQUIRE_PTR(flow2);
  FS_REQUIRE_PTR(flow_tea); FS_REQUIRE_PTR(coef);
  FS_REQUIRE_PTR(grad_flow0); FS_REQUIRE_PTR(grad_flow1); FS_REQUIRE_PTR(grad_flow2);
  if (B < 1 || C < 1 || F < 1 || S < 1) return FS_ERR_SHAPE;
  DP p = {C, F, S, (long long)B * S};
  D3 a = {{merged0, merged1, merged2}, {flow0, flow1, flow2}, {grad_flow0, grad_flow1, grad_flow2}};
  hipLaunchKernelGGL(distill3_bwd_kernel, dim3(blocks_for(p.nBS, 16384)), dim3(256), 0, (hipStream_t)stream, a,
                     merged_tea, gt, flow_tea, coef, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
