// corr2d.hip -- local-window correlation (cost volume) of UPFlow for gfx950.
//
// Replaces correlation_cuda.forward/backward as called by CorrelationFunction
// (UPFlow/model/correlation_package/correlation.py:26-27,42-43) with pad = max_displacement = md,
// kernel_size = 1, stride1 = stride2 = 1, corr_multiply = 1 (UPFlow/model/upflow.py:649,652):
//
//   out[b, (dy+md)(2md+1) + (dx+md), y, x] = (1/C) sum_c f1[b,c,y,x] * f2[b,c,y+dy,x+dx]
//
// f2 reads as 0 outside the image; dy-major channel order (pinned by Corr_pyTorch,
// UPFlow/utils/pytorch_correlation.py:27-50).
//
// Forward.  One workgroup = one 8x32-pixel tile of one sample, 2md+1 waves: wave `dy` owns one
// displacement row, lane = a quad of 4 consecutive pixels, so each lane keeps 4 x (2md+1)
// accumulators.  Channels are streamed through LDS in chunks of 8: the f2 search window
// (tile + md halo) and the f1 tile are loaded once per chunk with coalesced rows, then every
// lane reads its f2 row segment (4 + 2md floats) as ds_read_b128 and does 4*(2md+1) FMAs per
// 3 LDS instructions -- VALU-bound, not LDS-bound.  HBM traffic is the algorithmic
// 4*(2C + (2md+1)^2) B/pixel; the halo re-reads of f2 are served by L2.
// The channel reduction runs in registers (lanes = pixels); splitting C across lanes and
// reducing with wave shuffles would cost 6 DPP steps per (pixel, displacement) and only pays for
// C >> pixels, which no UPFlow level has.
//
// Backward.  grad_f1[c,p] = (1/C) sum_d g[d,p] f2[c,p+d] is a gather;  grad_f2 is the same gather
// with the roles swapped and the displacement negated:  grad_f2[c,q] = (1/C) sum_d gT[d,q] f1[c,q+d]
// with gT[d,q] = g[-d, q+d] (0 when q+d is outside).  One launch computes both (blockIdx.z picks),
// no atomics, bitwise reproducible: thread = pixel, its (2md+1)^2 upstream gradients live in
// registers, channel chunks of the other feature map are staged in LDS.
#include "common.hpp"

namespace {

constexpr int TY = 8, TX = 32, CC = 8;

template <int MD>
__global__ __launch_bounds__(64 * (2 * MD + 1)) void corr2d_fwd_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ st1,
    const float* __restrict__ st2, float* __restrict__ out, int C, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD;        // staged rows
  constexpr int SCOLS = TX + 2 * MD;     // staged cols (multiple of 4 for MD in {2,4}; padded below)
  constexpr int SW = (SCOLS + 3) / 4 * 4;  // row stride, 16-B aligned for ds_read_b128
  constexpr int NT = 64 * ND;
  __shared__ __attribute__((aligned(16))) float s2[CC][SR][SW];
  __shared__ __attribute__((aligned(16))) float s1[CC][TY][TX];

  const int b = blockIdx.z;
  const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
  const int t = threadIdx.x;
  const int lane = t & 63, dy = t >> 6;  // wave index = displacement row
  const int qy = lane >> 3, qx = (lane & 7) * 4;
  const size_t HW = (size_t)H * W;
  const float* f1b = f1 + (size_t)b * C * HW;
  const float* f2b = f2 + (size_t)b * C * HW;

  float acc[4][ND];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < ND; ++j) acc[i][j] = 0.f;

  for (int c0 = 0; c0 < C; c0 += CC) {
    // stage the f2 window and the f1 tile of CC channels (zeros outside image / channel range)
    for (int i = t; i < CC * SR * SCOLS; i += NT) {
      const int c = i / (SR * SCOLS), rem = i - c * (SR * SCOLS);
      const int r = rem / SCOLS, col = rem - r * SCOLS;
      const int gy = y0 + r - MD, gx = x0 + col - MD;
      float v = 0.f;
      if (c0 + c < C && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        v = f2b[(size_t)(c0 + c) * HW + (size_t)gy * W + gx];
        // normalize_features folded into the load (§8f.4): the zero padding applies AFTER it
        if (st2) { const float* q = st2 + 2 * ((size_t)b * C + c0 + c); v = (v - q[0]) * q[1]; }
      }
      s2[c][r][col] = v;
    }
    for (int i = t; i < CC * TY * TX; i += NT) {
      const int c = i / (TY * TX), rem = i - c * (TY * TX);
      const int r = rem / TX, col = rem - r * TX;
      const int gy = y0 + r, gx = x0 + col;
      float v = 0.f;
      if (c0 + c < C && gy < H && gx < W) {
        v = f1b[(size_t)(c0 + c) * HW + (size_t)gy * W + gx];
        if (st1) { const float* q = st1 + 2 * ((size_t)b * C + c0 + c); v = (v - q[0]) * q[1]; }
      }
      s1[c][r][col] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(&s1[c][qy][qx]);
      float row[4 + 2 * MD + 3];
      const float* rp = &s2[c][qy + dy][qx];
#pragma unroll
      for (int k = 0; k < (4 + 2 * MD + 3) / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(rp + 4 * k);
        row[4 * k] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
      }
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) acc[i][j] = fmaf(av[i], row[i + j], acc[i][j]);
    }
    __syncthreads();
  }

  const int y = y0 + qy;
  if (y >= H) return;
  const float fC = (float)C;
  float* ob = out + ((size_t)b * ND * ND + (size_t)dy * ND) * HW + (size_t)y * W;
#pragma unroll
  for (int j = 0; j < ND; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + qx + i;
      if (x < W) ob[(size_t)j * HW + x] = acc[i][j] / fC;  // torch.mean = sum / C
    }
  }
}

// grad[c,p] = (1/C) sum_d g(d,p) * other[c, p+d];  z < B: (g = gout, other = f2) -> grad_f1;
// z >= B: (g = gout transposed on the fly, other = f1) -> grad_f2.
template <int MD>
__global__ __launch_bounds__(256) void corr2d_bwd_kernel(const float* __restrict__ f1,
                                                         const float* __restrict__ f2,
                                                         const float* __restrict__ st1,
                                                         const float* __restrict__ st2,
                                                         const float* __restrict__ gout,
                                                         float* __restrict__ g1,
                                                         float* __restrict__ g2, int B, int C, int H,
                                                         int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD, SW = TX + 2 * MD;
  __shared__ float s[CC][SR][SW];

  const bool second = (int)blockIdx.z >= B;
  const int b = second ? blockIdx.z - B : blockIdx.z;
  float* grad = second ? g2 : g1;
  if (grad == nullptr) return;  // uniform per block
  const float* other = second ? f1 : f2;
  const float* ost = second ? st1 : st2;  // moments of `other` (NULL: plain correlation)
  const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
  const int t = threadIdx.x;
  const int py = t / TX, px = t % TX;
  const int y = y0 + py, x = x0 + px;
  const bool live = (y < H && x < W);
  const size_t HW = (size_t)H * W;
  const float* gb = gout + (size_t)b * ND * ND * HW;
  const float* ob = other + (size_t)b * C * HW;

  // this pixel's (2md+1)^2 upstream gradients
  float g[ND][ND];
#pragma unroll
  for (int j = 0; j < ND; ++j)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      float v = 0.f;
      if (live) {
        if (!second) {
          v = gb[(size_t)(j * ND + i) * HW + (size_t)y * W + x];
        } else {
          // gT[d, q] = g[-d, q + d]
          const int yy = y + (j - MD), xx = x + (i - MD);
          if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = gb[(size_t)((ND - 1 - j) * ND + (ND - 1 - i)) * HW + (size_t)yy * W + xx];
        }
      }
      g[j][i] = v;
    }

  const float fC = (float)C;
  for (int c0 = 0; c0 < C; c0 += CC) {
    for (int i = t; i < CC * SR * SW; i += 256) {
      const int c = i / (SR * SW), rem = i - c * (SR * SW);
      const int r = rem / SW, col = rem - r * SW;
      const int gy = y0 + r - MD, gx = x0 + col - MD;
      float v = 0.f;
      if (c0 + c < C && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        v = ob[(size_t)(c0 + c) * HW + (size_t)gy * W + gx];
        if (ost) { const float* q = ost + 2 * ((size_t)b * C + c0 + c); v = (v - q[0]) * q[1]; }
      }
      s[c][r][col] = v;
    }
    __syncthreads();
    for (int c = 0; c < CC && c0 + c < C; ++c) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int i = 0; i < ND; ++i) a = fmaf(g[j][i], s[c][py + j][px + i], a);
      if (live) grad[((size_t)b * C + c0 + c) * HW + (size_t)y * W + x] = a / fC;
    }
    __syncthreads();
  }
}

template <int MD>
int launch_fwd(const float* f1, const float* f2, const float* st1, const float* st2, float* out, int B,
               int C, int H, int W, hipStream_t st) {
  dim3 grid(fs::cdiv(W, TX), fs::cdiv(H, TY), B);
  hipLaunchKernelGGL(corr2d_fwd_kernel<MD>, grid, dim3(64 * (2 * MD + 1)), 0, st, f1, f2, st1, st2, out, C,
                     H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_bwd(const float* f1, const float* f2, const float* st1, const float* st2, const float* gout,
               float* g1, float* g2, int B, int C, int H, int W, hipStream_t st) {
  dim3 grid(fs::cdiv(W, TX), fs::cdiv(H, TY), 2 * B);
  hipLaunchKernelGGL(corr2d_bwd_kernel<MD>, grid, dim3(256), 0, st, f1, f2, st1, st2, gout, g1, g2, B, C, H,
                     W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int check_shape(int B, int C, int H, int W, int md) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return FS_ERR_SHAPE;
  if (2 * (long long)B > 65535 || fs::cdiv(H, TY) > 65535) return FS_ERR_SHAPE;
  if (md < 1 || md > 4) return FS_ERR_ARG;
  return FS_OK;
}

int run_fwd(const float* f1, const float* f2, const float* st1, const float* st2, float* out, int B, int C,
            int H, int W, int md, hipStream_t st) {
  switch (md) {
    case 1: return launch_fwd<1>(f1, f2, st1, st2, out, B, C, H, W, st);
    case 2: return launch_fwd<2>(f1, f2, st1, st2, out, B, C, H, W, st);
    case 3: return launch_fwd<3>(f1, f2, st1, st2, out, B, C, H, W, st);
    default: return launch_fwd<4>(f1, f2, st1, st2, out, B, C, H, W, st);
  }
}

int run_bwd(const float* f1, const float* f2, const float* st1, const float* st2, const float* gout,
            float* g1, float* g2, int B, int C, int H, int W, int md, hipStream_t st) {
  switch (md) {
    case 1: return launch_bwd<1>(f1, f2, st1, st2, gout, g1, g2, B, C, H, W, st);
    case 2: return launch_bwd<2>(f1, f2, st1, st2, gout, g1, g2, B, C, H, W, st);
    case 3: return launch_bwd<3>(f1, f2, st1, st2, gout, g1, g2, B, C, H, W, st);
    default: return launch_bwd<4>(f1, f2, st1, st2, gout, g1, g2, B, C, H, W, st);
  }
}

// ---- per-plane moments and the adjoint of (f - mean) * rstd (normalize_features, §8f.4) ----------
// One workgroup per (b, c) plane; UPFlow's feature planes have 24 .. 4294 elements.
__device__ __forceinline__ double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void plane_moments_kernel(const float* __restrict__ f, float* __restrict__ stats,
                                                            int S) {
  __shared__ double red[256];
  const float* p = f + (size_t)blockIdx.x * S;
  double s = 0.0;
  for (int i = threadIdx.x; i < S; i += 256) s += (double)p[i];
  const float mean = (float)(block_sum(s, red) / (double)S);
  double q = 0.0;
  for (int i = threadIdx.x; i < S; i += 256) { const float d = p[i] - mean; q += (double)(d * d); }
  const float var = (float)(block_sum(q, red) / (double)(S - 1));  // torch.var: unbiased
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = 1.0f / sqrtf(var + 1e-16f);  // upflow.py:127 std = sqrt(var + 1e-16)
  }
}

// n = (f - m) r,  m = mean f,  r = (var + eps)^-1/2,  var = sum (f - m)^2 / (S - 1):
//   df_i = r * ( dn_i - mean(dn) - n_i * sum_j(dn_j n_j) / (S - 1) )
__global__ __launch_bounds__(256) void plane_norm_bwd_kernel(const float* __restrict__ f,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gn, float* __restrict__ gf,
                                                             int S) {
  __shared__ double red[256];
  const size_t base = (size_t)blockIdx.x * S;
  const float m = stats[2 * blockIdx.x], r = stats[2 * blockIdx.x + 1];
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < S; i += 256) {
    const float g = gn[base + i], n = (f[base + i] - m) * r;
    a += (double)g;
    c += (double)(g * n);
  }
  const float s1 = (float)(block_sum(a, red) / (double)S);
  const float s2 = (float)(block_sum(c, red) / (double)(S - 1));
  for (int i = threadIdx.x; i < S; i += 256) {
    const float n = (f[base + i] - m) * r;
    gf[base + i] = r * (gn[base + i] - s1 - n * s2);
  }
}

}  // namespace

extern "C" int fs_corr2d_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W,
                             int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(out);
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_fwd(f1, f2, nullptr, nullptr, out, B, C, H, W, max_displacement, (hipStream_t)stream);
}

extern "C" int fs_corr2d_bwd(const float* f1, const float* f2, const float* grad_out, float* grad_f1,
                             float* grad_f2, int B, int C, int H, int W, int max_displacement,
                             fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(grad_out);
  if (grad_f1 == nullptr && grad_f2 == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_bwd(f1, f2, nullptr, nullptr, grad_out, grad_f1, grad_f2, B, C, H, W, max_displacement,
                 (hipStream_t)stream);
}

extern "C" int fs_plane_moments(const float* f, float* stats, int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f); FS_REQUIRE_PTR(stats);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;  // unbiased variance needs two samples
  hipLaunchKernelGGL(plane_moments_kernel, dim3(planes), dim3(256), 0, (hipStream_t)stream, f, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_plane_norm_bwd(const float* f, const float* stats, const float* grad_n, float* grad_f,
                                 int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f); FS_REQUIRE_PTR(stats); FS_REQUIRE_PTR(grad_n); FS_REQUIRE_PTR(grad_f);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;
  hipLaunchKernelGGL(plane_norm_bwd_kernel, dim3(planes), dim3(256), 0, (hipStream_t)stream, f, stats, grad_n,
                     grad_f, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_corr2d_norm_fwd(const float* f1, const float* f2, const float* stats1, const float* stats2,
                                  float* out, int B, int C, int H, int W, int max_displacement,
                                  fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(stats1); FS_REQUIRE_PTR(stats2); FS_REQUIRE_PTR(out);
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_fwd(f1, f2, stats1, stats2, out, B, C, H, W, max_displacement, (hipStream_t)stream);
}

extern "C" int fs_corr2d_norm_bwd(const float* f1, const float* f2, const float* stats1, const float* stats2,
                                  const float* grad_out, float* grad_n1, float* grad_n2, int B, int C, int H,
                                  int W, int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(stats1); FS_REQUIRE_PTR(stats2);
  FS_REQUIRE_PTR(grad_out);
  if (grad_n1 == nullptr && grad_n2 == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_bwd(f1, f2, stats1, stats2, grad_out, grad_n1, grad_n2, B, C, H, W, max_displacement,
                 (hipStream_t)stream);
}
