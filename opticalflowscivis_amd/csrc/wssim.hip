// wssim.hip -- the 'SSIM' branch of photo_loss_multi_type for gfx950:
// network_tools.weighted_ssim (UPFlow/model/upflow.py:141-196) + the masked reduction (:285-289) fused.
//
//   apw  = avgpool3x3(w)                       (valid windows: output (H-2) x (W-2))
//   E[f] = avgpool3x3(f * (w + 0.01)) / (apw + 0.01)
//   r    = (2 (E[xy] - E[x]E[y]) + c2) / ((E[xx] - E[x]^2) + (E[yy] - E[y]^2) + c2),   c2 = 9e-6 (c1 = inf)
//   ld   = clamp((1 - r) / 2, 0, 1)
//   S1   = sum ld * (use_occ ? apw : 1),  S2 = sum apw          (loss = S1/(S2+1e-6) or S1/N)
//
// The reference runs 5 avg-pools on 3-channel temporaries and ~20 elementwise passes; here one thread
// owns one window and reads its 3x3 neighbourhoods once (L1-served overlap), with the same two-stage
// deterministic reduction as fs_robust_sum.  Backward is a gather: every window leaves 6 adjoint
// coefficients in LDS, every pixel sums the (up to) 9 windows it belongs to.  No atomics.
#include "common.hpp"

namespace {

constexpr float kC2 = 9e-6f, kEps = 0.01f;

struct SP { int B, C, H, W; int use_occ; };

struct Win { float mux, muy, sxx, syy, sxy, apw, inv; };

__device__ __forceinline__ Win window(const float* __restrict__ x, const float* __restrict__ y,
                                      const float* __restrict__ w, int W, int qy, int qx) {
  float sw = 0.f, sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int o = (qy + j) * W + qx + i;
      const float ww = w[o], we = ww + kEps, xv = x[o], yv = y[o];
      sw += ww;
      sx += xv * we; sy += yv * we;
      sxx += xv * xv * we; syy += yv * yv * we; sxy += xv * yv * we;
    }
  Win r;
  r.apw = sw * (1.0f / 9.0f);
  r.inv = 1.0f / (r.apw + kEps);
  const float k = r.inv * (1.0f / 9.0f);
  r.mux = sx * k; r.muy = sy * k;
  r.sxx = sxx * k - r.mux * r.mux;
  r.syy = syy * k - r.muy * r.muy;
  r.sxy = sxy * k - r.mux * r.muy;
  return r;
}

__global__ __launch_bounds__(256) void wssim_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ y,
                                                        const float* __restrict__ w,
                                                        float* __restrict__ ws, SP p) {
  const int Ho = p.H - 2, Wo = p.W - 2;
  const long long n = (long long)p.B * p.C * Ho * Wo;
  const size_t HW = (size_t)p.H * p.W;
  float s1 = 0.f, s2 = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int qx = (int)(e % Wo);
    long long t = e / Wo;
    const int qy = (int)(t % Ho); t /= Ho;
    const int c = (int)(t % p.C);
    const int b = (int)(t / p.C);
    const Win q = window(x + ((size_t)b * p.C + c) * HW, y + ((size_t)b * p.C + c) * HW, w + (size_t)b * HW,
                         p.W, qy, qx);
    const float r = (2.0f * q.sxy + kC2) / (q.sxx + q.syy + kC2);
    const float ld = fminf(fmaxf((1.0f - r) * 0.5f, 0.0f), 1.0f);
    s1 += p.use_occ ? ld * q.apw : ld;
    if (c == 0) s2 += q.apw;
  }
  fs::block_pair_to_ws(s1, s2, ws);
}

constexpr int ST = 16;  // 16x16 pixel tile; windows needed: 18x18

__global__ __launch_bounds__(256) void wssim_bwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ y,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ coef,
                                                        float* __restrict__ gx, float* __restrict__ gy, SP p) {
  // per window: A, B, A*mux, A*muy, B*mux, B*muy   (A = g dr/dsxx inv/9, B = g dr/dsxy inv/9)
  __shared__ float sw6[6][ST + 2][ST + 3];
  const int Ho = p.H - 2, Wo = p.W - 2;
  const int bc = blockIdx.z, b = bc / p.C;
  const int y0 = blockIdx.y * ST, x0 = blockIdx.x * ST;
  const size_t HW = (size_t)p.H * p.W;
  const float* xp = x + (size_t)bc * HW;
  const float* yp = y + (size_t)bc * HW;
  const float* wp = w + (size_t)b * HW;
  const float k = coef[0];
  for (int i = threadIdx.x; i < (ST + 2) * (ST + 2); i += 256) {
    const int r = i / (ST + 2), c = i - r * (ST + 2);
    const int qy = y0 + r - 2, qx = x0 + c - 2;  // windows whose 3x3 footprint touches the tile
    float A = 0.f, Bq = 0.f, mux = 0.f, muy = 0.f;
    if (qy >= 0 && qy < Ho && qx >= 0 && qx < Wo) {
      const Win q = window(xp, yp, wp, p.W, qy, qx);
      const float d = q.sxx + q.syy + kC2, nn = 2.0f * q.sxy + kC2;
      const float rr = nn / d;
      const float half = (1.0f - rr) * 0.5f;
      // d ld / d r = -1/2 inside the clamp, 0 outside (torch.clamp passes the gradient on the bounds)
      const float dld = (half >= 0.0f && half <= 1.0f) ? -0.5f : 0.0f;
      const float g = k * dld * (p.use_occ ? q.apw : 1.0f) * q.inv * (1.0f / 9.0f);
      A = g * (-nn / (d * d));
      Bq = g * (2.0f / d);
      mux = q.mux; muy = q.muy;
    }
    sw6[0][r][c] = A; sw6[1][r][c] = Bq;
    sw6[2][r][c] = A * mux; sw6[3][r][c] = A * muy;
    sw6[4][r][c] = Bq * mux; sw6[5][r][c] = Bq * muy;
  }
  __syncthreads();
  const int py = threadIdx.x / ST, px = threadIdx.x % ST;
  const int yy = y0 + py, xx = x0 + px;
  if (yy >= p.H || xx >= p.W) return;
  float a = 0.f, bb = 0.f, amx = 0.f, amy = 0.f, bmx = 0.f, bmy = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // window (yy - j, xx - i) -> tile-local (py + 2 - j, px + 2 - i)
      const int r = py + 2 - j, c = px + 2 - i;
      a += sw6[0][r][c]; bb += sw6[1][r][c];
      amx += sw6[2][r][c]; amy += sw6[3][r][c];
      bmx += sw6[4][r][c]; bmy += sw6[5][r][c];
    }
  const size_t o = (size_t)yy * p.W + xx;
  const float we = wp[o] + kEps, xv = xp[o], yv = yp[o];
  if (gx) gx[(size_t)bc * HW + o] = we * (2.0f * (xv * a - amx) + (yv * bb - bmy));
  if (gy) gy[(size_t)bc * HW + o] = we * (2.0f * (yv * a - amy) + (xv * bb - bmx));
}

int check(const SP& p) {
  if (p.B < 1 || p.C < 1 || p.H < 3 || p.W < 3) return FS_ERR_SHAPE;
  if ((long long)p.B * p.C > 65535 || fs::cdiv(p.H, ST) > 65535) return FS_ERR_SHAPE;
  return FS_OK;
}

}  // namespace

extern "C" int fs_wssim_fwd(const float* x, const float* y, const float* weight, float* sums, float* ws,
                            int B, int C, int H, int W, int use_occ, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(x); FS_REQUIRE_PTR(y); FS_REQUIRE_PTR(weight); FS_REQUIRE_PTR(sums); FS_REQUIRE_PTR(ws);
  SP p = {B, C, H, W, use_occ ? 1 : 0};
  const int rc = check(p);
  if (rc != FS_OK) return rc;
  const long long n = (long long)B * C * (H - 2) * (W - 2);
  const long long want = (n + 255) / 256;
  const int nb = (int)(want < FS_REDUCE_BLOCKS ? want : FS_REDUCE_BLOCKS);
  hipLaunchKernelGGL(wssim_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, weight, ws, p);
  hipLaunchKernelGGL(fs::reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, nb, sums);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_wssim_bwd(const float* x, const float* y, const float* weight, const float* coef,
                            float* grad_x, float* grad_y, int B, int C, int H, int W, int use_occ,
                            fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(x); FS_REQUIRE_PTR(y); FS_REQUIRE_PTR(weight); FS_REQUIRE_PTR(coef);
  if (grad_x == nullptr && grad_y == nullptr) return FS_ERR_NULLPTR;
  SP p = {B, C, H, W, use_occ ? 1 : 0};
  const int rc = check(p);
  if (rc != FS_OK) return rc;
  dim3 grid(fs::cdiv(W, ST), fs::cdiv(H, ST), B * C);
  hipLaunchKernelGGL(wssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, weight, coef, grad_x,
                     grad_y, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
