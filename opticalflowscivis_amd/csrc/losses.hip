// losses.hip -- photometric / census unsupervised losses of UPFlow for gfx950
//  as fused HIP kernels.
//
// (1) fs_census_dist_*: UPFlow/utils/loss.py:51-72.  The reference materialises two 49-channel
//     tensors with an identity convolution, then three more 49-channel temporaries; here one
//     thread owns one pixel, both grey tiles (+3 halo) sit in LDS and the 49 soft-ternary /
//     soft-Hamming terms never leave registers: 24 B/px read (2 x RGB), 4 B/px written.
//     Backward is gather-formulated (no atomics, reproducible): the gradient of pixel q collects
//     the 49 terms in which q is the neighbour plus the 49 in which it is the centre.
// (2) fs_robust_sum_*: the robust penalty + (masked) reduction shared by photo_loss_function
//     (loss.py:17-48, the tail of the census loss) and photo_loss_multi_type
//     (UPFlow/model/upflow.py:267-289): S1 = sum pen(x - y) * w, S2 = sum w in one pass over the
//     operands; two-stage deterministic reduction (per-block partials, then one block in fp64).
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// robust penalty + reduction
// ------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float pen_fwd(float v, float q, float eps) {
  if (MODE == FS_PEN_ABS_ROBUST) return powf(fabsf(v) + 0.01f, q);
  if (MODE == FS_PEN_CHARBONNIER) return powf(v * v + eps, q);
  if (MODE == FS_PEN_L1_EPS) return fabsf(v + 1e-6f);
  return fabsf(v);  // FS_PEN_L1
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

template <int MODE>
__device__ __forceinline__ float pen_bwd(float v, float q, float eps) {
  if (MODE == FS_PEN_ABS_ROBUST) return q * powf(fabsf(v) + 0.01f, q - 1.0f) * sgn(v);
  if (MODE == FS_PEN_CHARBONNIER) return q * powf(v * v + eps, q - 1.0f) * 2.0f * v;
  if (MODE == FS_PEN_L1_EPS) return sgn(v + 1e-6f);
  return sgn(v);
}

struct RSP {
  long long n;     // B*C*S elements
  long long CS;    // C*S
  int S;           // spatial size per channel
  int H, W;        // image extent (only used when border > 0: S == H*W)
  int border;      // weights of pixels closer than `border` to the image edge are 0
  int mode;
  float q, eps;
};

__device__ __forceinline__ float rs_weight(const RSP& p, const float* __restrict__ w, long long e,
                                           int& c_out) {
  const long long b = e / p.CS;
  const long long rem = e - b * p.CS;
  const int c = (int)(rem / p.S);
  const int r = (int)(rem - (long long)c * p.S);
  c_out = c;
  float wt = w[b * p.S + r];
  if (p.border > 0) {
    const int y = r / p.W, x = r - y * p.W;
    if (y < p.border || y >= p.H - p.border || x < p.border || x >= p.W - p.border) wt = 0.f;
  }
  return wt;
}

constexpr int RS_BLOCKS = FS_REDUCE_BLOCKS;

template <int MODE>
__global__ __launch_bounds__(256) void robust_sum_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ w,
                                                         float* __restrict__ ws, RSP p) {
  float s1 = 0.f, s2 = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < p.n; e += (long long)gridDim.x * 256) {
    const float v = y ? x[e] - y[e] : x[e];
    float l = pen_fwd<MODE>(v, p.q, p.eps);
    if (w) {
      int c;
      const float wt = rs_weight(p, w, e, c);
      l *= wt;
      if (c == 0) s2 += wt;
    }
    s1 += l;
  }
  fs::block_pair_to_ws(s1, s2, ws);
}

template <int MODE>
__global__ __launch_bounds__(256) void robust_sum_bwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ coef,
                                                             float* __restrict__ gx,
                                                             float* __restrict__ gy, RSP p) {
  const float k = coef[0];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < p.n; e += (long long)gridDim.x * 256) {
    const float v = y ? x[e] - y[e] : x[e];
    float g = k * pen_bwd<MODE>(v, p.q, p.eps);
    if (w) {
      int c;
      g *= rs_weight(p, w, e, c);
    }
    if (gx) gx[e] = g;
    if (gy) gy[e] = -g;
  }
}

int make_rsp(RSP& p, int B, int C, int S, int H, int W, int border, int mode, float q, float eps) {
  if (B < 1 || C < 1 || S < 1) return FS_ERR_SHAPE;
  if (mode < FS_PEN_ABS_ROBUST || mode > FS_PEN_L1) return FS_ERR_ARG;
  if (border < 0) return FS_ERR_ARG;
  if (border > 0 && (long long)H * W != S) return FS_ERR_SHAPE;
  p.n = (long long)B * C * S;
  p.CS = (long long)C * S;
  p.S = S; p.H = H; p.W = W; p.border = border; p.mode = mode; p.q = q; p.eps = eps;
  return FS_OK;
}

// ------------------------------------------------------------------------------------------
// census
// ------------------------------------------------------------------------------------------
constexpr int CT = 16;  // 16x16 pixel tile, 256 threads

__device__ __forceinline__ float gray_of(const float* __restrict__ img, size_t HW, size_t o) {
#pragma clang fp contract(off)
  return 0.2989f * img[o] + 0.5870f * img[HW + o] + 0.1140f * img[2 * HW + o];  // loss.py:56
}

template <int MD>
__global__ __launch_bounds__(256) void census_dist_kernel(const float* __restrict__ img1,
                                                          const float* __restrict__ img2,
                                                          float* __restrict__ dist, int H, int W) {
  constexpr int P = 2 * MD + 1, SW = CT + 2 * MD;
  __shared__ float g1[SW][SW + 1], g2[SW][SW + 1];
  const int b = blockIdx.z, y0 = blockIdx.y * CT, x0 = blockIdx.x * CT;
  const size_t HW = (size_t)H * W;
  const float* i1 = img1 + (size_t)b * 3 * HW;
  const float* i2 = img2 + (size_t)b * 3 * HW;
  for (int i = threadIdx.x; i < SW * SW; i += 256) {
    const int r = i / SW, c = i - r * SW;
    const int gy = y0 + r - MD, gx = x0 + c - MD;
    float a = 0.f, bb = 0.f;  // conv2d zero padding (loss.py:64)
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const size_t o = (size_t)gy * W + gx;
      a = gray_of(i1, HW, o);
      bb = gray_of(i2, HW, o);
    }
    g1[r][c] = a;
    g2[r][c] = bb;
  }
  __syncthreads();
  const int py = threadIdx.x / CT, px = threadIdx.x % CT;
  const int y = y0 + py, x = x0 + px;
  if (y >= H || x >= W) return;
  const float c1 = g1[py + MD][px + MD], c2 = g2[py + MD][px + MD];
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float u1 = g1[py + j][px + i] - c1, u2 = g2[py + j][px + i] - c2;  // :65
      // :66 t = u / sqrt(0.81 + u^2) and :71 d / (0.1 + d) with v_rsq / v_rcp (1 ulp) instead of the
      // IEEE divide + sqrt sequences: the kernel is VALU-bound (49 x 2 divides + sqrts per pixel)
      const float t1 = u1 * rsqrtf(0.81f + u1 * u1), t2 = u2 * rsqrtf(0.81f + u2 * u2);
      const float d = (t1 - t2) * (t1 - t2);  // :70
      acc += d * __builtin_amdgcn_rcpf(0.1f + d);
    }
  dist[(size_t)b * HW + (size_t)y * W + x] = acc;
}

// A pixel pair (q, n = q + delta) enters the distance twice: in the term with centre q and neighbour n (u = g[n] - g[q],
// weight k[q]) and in the term with centre n and neighbour q (u' = -u, weight k[n]).  T(u) = u / sqrt(0.81 + u^2) is odd,
// D(e) = e^2 / (0.1 + e^2) even, so both terms share every transcendental: with e = T(u1) - T(u2),
//   d dist / d g1[q] = -(k[q] + k[n]) D'(e) T'(u1),   d dist / d g2[q] = +(k[q] + k[n]) D'(e) T'(u2),
//   D'(e) = 0.2 e / (0.1 + e^2)^2,  T'(u) = 0.81 / (0.81 + u^2)^1.5
// -- 49 evaluations per pixel (2 v_rsq + 1 v_rcp each) instead of the 98 of rounds 1-3 (PMC then: 3 400 VALU
// instructions per pixel, a vector instruction issuing in 70 % of the CU-busy cycles).
__device__ __forceinline__ void census_pair_grad(float u1, float u2, float ksum, float& a1, float& a2) {
  const float r1 = rsqrtf(0.81f + u1 * u1), r2 = rsqrtf(0.81f + u2 * u2);
  const float e = u1 * r1 - u2 * r2;
  const float den = 0.1f + e * e;
  const float dD = ksum * (0.2f * 0.81f) * e * __builtin_amdgcn_rcpf(den * den);  // v_rcp: the kernel is VALU-bound
  a1 -= dD * (r1 * r1 * r1);
  a2 += dD * (r2 * r2 * r2);
}

template <int MD>
__global__ __launch_bounds__(256) void census_dist_bwd_kernel(const float* __restrict__ img1,
                                                              const float* __restrict__ img2,
                                                              const float* __restrict__ gdist,
                                                              float* __restrict__ gimg1,
                                                              float* __restrict__ gimg2, int H,
                                                              int W) {
  constexpr int P = 2 * MD + 1, SW = CT + 2 * MD;
  __shared__ float g1[SW][SW + 1], g2[SW][SW + 1], kk[SW][SW + 1];
  const int b = blockIdx.z, y0 = blockIdx.y * CT, x0 = blockIdx.x * CT;
  const size_t HW = (size_t)H * W;
  const float* i1 = img1 + (size_t)b * 3 * HW;
  const float* i2 = img2 + (size_t)b * 3 * HW;
  const float* gd = gdist + (size_t)b * HW;
  for (int i = threadIdx.x; i < SW * SW; i += 256) {
    const int r = i / SW, c = i - r * SW;
    const int gy = y0 + r - MD, gx = x0 + c - MD;
    float a = 0.f, bb = 0.f, k = 0.f;  // k = 0 outside: such centres do not exist
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const size_t o = (size_t)gy * W + gx;
      a = gray_of(i1, HW, o);
      bb = gray_of(i2, HW, o);
      k = gd[o];
    }
    g1[r][c] = a; g2[r][c] = bb; kk[r][c] = k;
  }
  __syncthreads();
  const int py = threadIdx.x / CT, px = threadIdx.x % CT;
  const int y = y0 + py, x = x0 + px;
  if (y >= H || x >= W) return;
  const float c1 = g1[py + MD][px + MD], c2 = g2[py + MD][px + MD], kc = kk[py + MD][px + MD];
  float a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < P; ++i) {
      // the pair (q, q + delta): q as the centre (zero-padded grey outside the image) and as the neighbour of the
      // centre q + delta (k = 0 where no such centre exists)
      census_pair_grad(g1[py + j][px + i] - c1, g2[py + j][px + i] - c2, kc + kk[py + j][px + i], a1, a2);
    }
  const size_t o = (size_t)y * W + x;
  if (gimg1) {
    float* o1 = gimg1 + (size_t)b * 3 * HW;
    o1[o] = 0.2989f * a1; o1[HW + o] = 0.5870f * a1; o1[2 * HW + o] = 0.1140f * a1;
  }
  if (gimg2) {
    float* o2 = gimg2 + (size_t)b * 3 * HW;
    o2[o] = 0.2989f * a2; o2[HW + o] = 0.5870f * a2; o2[2 * HW + o] = 0.1140f * a2;
  }
}

}  // namespace

extern "C" int fs_robust_sum(const float* x, const float* y, const float* w, float* sums, float* ws,
                             int B, int C, int S, int H, int W, int border, int mode, float q,
                             float eps, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(x); FS_REQUIRE_PTR(sums); FS_REQUIRE_PTR(ws);
  RSP p;
  const int rc = make_rsp(p, B, C, S, H, W, border, mode, q, eps);
  if (rc != FS_OK) return rc;
  const long long want = (p.n + 255) / 256;
  const int nb = (int)(want < RS_BLOCKS ? want : RS_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case FS_PEN_ABS_ROBUST:
      hipLaunchKernelGGL(robust_sum_kernel<FS_PEN_ABS_ROBUST>, dim3(nb), dim3(256), 0, st, x, y, w, ws, p);
      break;
    case FS_PEN_CHARBONNIER:
      hipLaunchKernelGGL(robust_sum_kernel<FS_PEN_CHARBONNIER>, dim3(nb), dim3(256), 0, st, x, y, w, ws, p);
      break;
    case FS_PEN_L1_EPS:
      hipLaunchKernelGGL(robust_sum_kernel<FS_PEN_L1_EPS>, dim3(nb), dim3(256), 0, st, x, y, w, ws, p);
      break;
    default:
      hipLaunchKernelGGL(robust_sum_kernel<FS_PEN_L1>, dim3(nb), dim3(256), 0, st, x, y, w, ws, p);
      break;
  }
  hipLaunchKernelGGL(fs::reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, nb, sums);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_robust_sum_bwd(const float* x, const float* y, const float* w, const float* coef,
                                 float* grad_x, float* grad_y, int B, int C, int S, int H, int W,
                                 int border, int mode, float q, float eps, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(x); FS_REQUIRE_PTR(coef);
  if (grad_x == nullptr && grad_y == nullptr) return FS_ERR_NULLPTR;
  if (grad_y != nullptr && y == nullptr) return FS_ERR_NULLPTR;
  RSP p;
  const int rc = make_rsp(p, B, C, S, H, W, border, mode, q, eps);
  if (rc != FS_OK) return rc;
  const long long want = (p.n + 255) / 256;
  const int nb = (int)(want < 16384 ? want : 16384);
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case FS_PEN_ABS_ROBUST:
      hipLaunchKernelGGL(robust_sum_bwd_kernel<FS_PEN_ABS_ROBUST>, dim3(nb), dim3(256), 0, st, x, y, w,
                         coef, grad_x, grad_y, p);
      break;
    case FS_PEN_CHARBONNIER:
      hipLaunchKernelGGL(robust_sum_bwd_kernel<FS_PEN_CHARBONNIER>, dim3(nb), dim3(256), 0, st, x, y, w,
                         coef, grad_x, grad_y, p);
      break;
    case FS_PEN_L1_EPS:
      hipLaunchKernelGGL(robust_sum_bwd_kernel<FS_PEN_L1_EPS>, dim3(nb), dim3(256), 0, st, x, y, w,
                         coef, grad_x, grad_y, p);
      break;
    default:
      hipLaunchKernelGGL(robust_sum_bwd_kernel<FS_PEN_L1>, dim3(nb), dim3(256), 0, st, x, y, w, coef,
                         grad_x, grad_y, p);
      break;
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_census_dist_fwd(const float* img1, const float* img2, float* dist, int B, int H,
                                  int W, int max_distance, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(img2); FS_REQUIRE_PTR(dist);
  if (B < 1 || H < 1 || W < 1 || B > 65535 || fs::cdiv(H, CT) > 65535) return FS_ERR_SHAPE;
  if (max_distance != 3) return FS_ERR_ARG;  // the reference's only value (loss.py:51)
  dim3 grid(fs::cdiv(W, CT), fs::cdiv(H, CT), B);
  hipLaunchKernelGGL(census_dist_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, img1, img2, dist,
                     H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_census_dist_bwd(const float* img1, const float* img2, const float* grad_dist,
                                  float* grad_img1, float* grad_img2, int B, int H, int W,
                                  int max_distance, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(img2); FS_REQUIRE_PTR(grad_dist);
  if (grad_img1 == nullptr && grad_img2 == nullptr) return FS_ERR_NULLPTR;
  if (B < 1 || H < 1 || W < 1 || B > 65535 || fs::cdiv(H, CT) > 65535) return FS_ERR_SHAPE;
  if (max_distance != 3) return FS_ERR_ARG;
  dim3 grid(fs::cdiv(W, CT), fs::cdiv(H, CT), B);
  hipLaunchKernelGGL(census_dist_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, img1, img2,
                     grad_dist, grad_img1, grad_img2, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
