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
// The channel reduction of the tiled kernel runs in registers (lanes = pixels).  That form needs pixels:
// the three coarsest UPFlow levels at C3 are (C, h, w) = (196, 3, 8), (128, 5, 15), (96, 10, 29) -- 24 .. 290
// pixels per sample, C >> pixels -- where an 8x32 tile is mostly empty and the kernel is a serial chain of
// C/8 stage-barrier-compute rounds (86 us for 150 K output floats).  Those levels (h*w <= 512) run the
// direct kernels below instead: no LDS, no barrier, one thread per (sample, displacement, pixel) output and
// channel slice -- the C range is split over 1..8 lane groups of a wave and reduced with wave shuffles
// (2 * log2(slices) DPP steps per output) -- so that even a B = 2 launch fills the chip.
//
// Backward.  grad_f1[c,p] = (1/C) sum_d g[d,p] f2[c,p+d] is a gather;  grad_f2 is the same gather
// with the roles swapped and the displacement negated:  grad_f2[c,q] = (1/C) sum_d gT[d,q] f1[c,q+d]
// with gT[d,q] = g[-d, q+d] (0 when q+d is outside).  One launch computes both (blockIdx.z picks),
// no atomics, bitwise reproducible: thread = pixel, its (2md+1)^2 upstream gradients live in
// registers, channel chunks of the other feature map are staged in LDS.
#include "common.hpp"

namespace {

constexpr int TY = 8, TX = 32, CC = 8;

// Two problems of one shape per launch (UPFlow correlates both directions at every level,
// upflow.py:649,652); pointers of the second problem may equal the first's.
struct C2Set {
  const float* f1[2];
  const float* f2[2];
  const float* st1[2];  // per-(b,c) (mean, rstd) of f1 / f2, or nullptr: plain correlation
  const float* st2[2];
  float* out[2];        // forward: cost volume;  backward: unused
  const float* gout[2];
  float* g1[2];
  float* g2[2];
  int nsets;
};


template <int MD>
__global__ __launch_bounds__(64 * (2 * MD + 1)) void corr2d_fwd_kernel(C2Set a, int B, int C, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD;        // staged rows
  constexpr int SCOLS = TX + 2 * MD;     // staged cols (multiple of 4 for MD in {2,4}; padded below)
  constexpr int SW = (SCOLS + 3) / 4 * 4;  // row stride, 16-B aligned for ds_read_b128
  constexpr int NT = 64 * ND;
  __shared__ __attribute__((aligned(16))) float s2[CC][SR][SW];
  __shared__ __attribute__((aligned(16))) float s1[CC][TY][TX];

  const int set = blockIdx.z / B, b = blockIdx.z - set * B;
  const float* __restrict__ f1 = a.f1[set];
  const float* __restrict__ f2 = a.f2[set];
  const float* __restrict__ st1 = a.st1[set];
  const float* __restrict__ st2 = a.st2[set];
  float* __restrict__ out = a.out[set];
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
__global__ __launch_bounds__(256) void corr2d_bwd_kernel(C2Set a, int B, int C, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD, SW = TX + 2 * MD;
  __shared__ float s[CC][SR][SW];

  const int set = blockIdx.z / (2 * B), zr = blockIdx.z - set * 2 * B;
  const bool second = zr >= B;
  const int b = second ? zr - B : zr;
  float* __restrict__ grad = second ? a.g2[set] : a.g1[set];
  if (grad == nullptr) return;  // uniform per block
  const float* __restrict__ other = second ? a.f1[set] : a.f2[set];
  const float* __restrict__ ost = second ? a.st1[set] : a.st2[set];  // moments of `other` (NULL: plain)
  const float* __restrict__ gout = a.gout[set];
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

// ---- coarse pyramid levels: direct kernels (h*w <= 512) -----------------------------------------------
// out[b, d, p] = (1/C) sum_c f1n[b,c,p] * f2n[b,c,p+d]: item = (b, d, p), CS channel slices per item laid
// out CS-strided inside the wave (lane = slice * (64/CS) + item-in-wave), reduced with shuffles.
template <int MD, int CS>
__global__ __launch_bounds__(256) void corr2d_small_fwd_kernel(C2Set a, int B, int C, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int IPW = 64 / CS;  // items per wave
  const int set = blockIdx.y;
  const float* __restrict__ f1 = a.f1[set];
  const float* __restrict__ f2 = a.f2[set];
  const float* __restrict__ st1 = a.st1[set];
  const float* __restrict__ st2 = a.st2[set];
  float* __restrict__ out = a.out[set];
  const int HW = H * W;
  const long long items = (long long)B * ND * ND * HW;
  const int lane = threadIdx.x & 63;
  const int cs = lane / IPW, sub = lane - cs * IPW;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long n = wave * IPW + sub;
  const bool live = n < items;
  const long long nn = live ? n : 0;
  const int p = (int)(nn % HW);
  const long long r = nn / HW;
  const int d = (int)(r % (ND * ND));
  const int b = (int)(r / (ND * ND));
  const int y = p / W, x = p - y * W;
  const int y2 = y + d / ND - MD, x2 = x + d % ND - MD;
  const bool inb = live && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
  const int cper = (C + CS - 1) / CS;
  const int cbeg = cs * cper, cend = min(C, cbeg + cper);
  float acc = 0.f;
  if (inb) {
    const float* p1 = f1 + ((size_t)b * C + cbeg) * HW + p;
    const float* p2 = f2 + ((size_t)b * C + cbeg) * HW + y2 * W + x2;
    if (st1 != nullptr) {
      const float* q1 = st1 + 2 * ((size_t)b * C + cbeg);
      const float* q2 = st2 + 2 * ((size_t)b * C + cbeg);
#pragma unroll 4
      for (int c = cbeg; c < cend; ++c, p1 += HW, p2 += HW, q1 += 2, q2 += 2) {
        const float2 s1v = *reinterpret_cast<const float2*>(q1), s2v = *reinterpret_cast<const float2*>(q2);
        acc = fmaf((p1[0] - s1v.x) * s1v.y, (p2[0] - s2v.x) * s2v.y, acc);
      }
    } else {
#pragma unroll 4
      for (int c = cbeg; c < cend; ++c, p1 += HW, p2 += HW) acc = fmaf(p1[0], p2[0], acc);
    }
  }
#pragma unroll
  for (int o = IPW; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (live && cs == 0) out[n] = acc / (float)C;  // torch.mean = sum / C
}

// grad_f1[b,c,p] = (1/C) sum_d g[b,d,p] f2n[b,c,p+d];  grad_f2[b,c,q] = (1/C) sum_d g[b,d,q-d] f1n[b,c,q-d].
// One thread per (b, c, pixel); blockIdx.z picks the gradient; gathers only, reproducible.
template <int MD>
__global__ __launch_bounds__(256) void corr2d_small_bwd_kernel(C2Set a, int B, int C, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  const int set = blockIdx.y;
  const bool second = blockIdx.z == 1;
  float* __restrict__ grad = second ? a.g2[set] : a.g1[set];
  if (grad == nullptr) return;  // uniform per block
  const float* __restrict__ other = second ? a.f1[set] : a.f2[set];
  const float* __restrict__ ost = second ? a.st1[set] : a.st2[set];
  const float* __restrict__ gout = a.gout[set];
  const int HW = H * W;
  const long long total = (long long)B * C * HW;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int p = (int)(e % HW);
  const long long bc = e / HW;
  const int b = (int)(bc / C);
  const int y = p / W, x = p - y * W;
  const float* ob = other + (size_t)bc * HW;
  const float* gb = gout + (size_t)b * ND * ND * HW;
  float m = 0.f, rs = 1.f;
  if (ost != nullptr) { m = ost[2 * bc]; rs = ost[2 * bc + 1]; }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int yy = second ? y - (j - MD) : y + (j - MD);
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int xx = second ? x - (i - MD) : x + (i - MD);
      if (xx < 0 || xx >= W) continue;
      // first: g at the own pixel, other at p + d;  second: both at q - d (the pixel that saw q under d)
      const float g = gb[(size_t)(j * ND + i) * HW + (second ? yy * W + xx : p)];
      float v = ob[yy * W + xx];
      if (ost != nullptr) v = (v - m) * rs;
      acc = fmaf(g, v, acc);
    }
  }
  grad[e] = acc / (float)C;
}

constexpr int kSmallHW = 512;  // direct kernels up to this many pixels per sample

template <int MD>
int launch_small_fwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  constexpr int ND = 2 * MD + 1;
  const long long items = (long long)B * ND * ND * H * W;
  // channel slices: enough threads for ~2 waves of workgroups per CU, at least 12 channels per slice
  int cs = 1;
  while (cs < 8 && items * cs < 256ll * 256 * 2 && C / (2 * cs) >= 12) cs *= 2;
  const long long waves = (items + (64 / cs) - 1) / (64 / cs);
  const dim3 grid((unsigned)((waves + 3) / 4), a.nsets);
  switch (cs) {
    case 1: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 1>), grid, dim3(256), 0, st, a, B, C, H, W); break;
    case 2: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 2>), grid, dim3(256), 0, st, a, B, C, H, W); break;
    case 4: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 4>), grid, dim3(256), 0, st, a, B, C, H, W); break;
    default: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 8>), grid, dim3(256), 0, st, a, B, C, H, W); break;
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_small_bwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  const long long total = (long long)B * C * H * W;
  const dim3 grid((unsigned)((total + 255) / 256), a.nsets, 2);
  hipLaunchKernelGGL(corr2d_small_bwd_kernel<MD>, grid, dim3(256), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_fwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  if (H * W <= kSmallHW) return launch_small_fwd<MD>(a, B, C, H, W, st);
  dim3 grid(fs::cdiv(W, TX), fs::cdiv(H, TY), B * a.nsets);
  hipLaunchKernelGGL(corr2d_fwd_kernel<MD>, grid, dim3(64 * (2 * MD + 1)), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_bwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  if (H * W <= kSmallHW) return launch_small_bwd<MD>(a, B, C, H, W, st);
  dim3 grid(fs::cdiv(W, TX), fs::cdiv(H, TY), 2 * B * a.nsets);
  hipLaunchKernelGGL(corr2d_bwd_kernel<MD>, grid, dim3(256), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int check_shape(int B, int C, int H, int W, int md) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return FS_ERR_SHAPE;
  if (4 * (long long)B > 65535 || fs::cdiv(H, TY) > 65535) return FS_ERR_SHAPE;
  if (md < 1 || md > 4) return FS_ERR_ARG;
  return FS_OK;
}

int run_fwd(const C2Set& a, int B, int C, int H, int W, int md, hipStream_t st) {
  switch (md) {
    case 1: return launch_fwd<1>(a, B, C, H, W, st);
    case 2: return launch_fwd<2>(a, B, C, H, W, st);
    case 3: return launch_fwd<3>(a, B, C, H, W, st);
    default: return launch_fwd<4>(a, B, C, H, W, st);
  }
}

int run_bwd(const C2Set& a, int B, int C, int H, int W, int md, hipStream_t st) {
  switch (md) {
    case 1: return launch_bwd<1>(a, B, C, H, W, st);
    case 2: return launch_bwd<2>(a, B, C, H, W, st);
    case 3: return launch_bwd<3>(a, B, C, H, W, st);
    default: return launch_bwd<4>(a, B, C, H, W, st);
  }
}

C2Set one_set(const float* f1, const float* f2, const float* st1, const float* st2, float* out, const float* gout,
              float* g1, float* g2) {
  C2Set a = {};
  a.f1[0] = f1; a.f2[0] = f2; a.st1[0] = st1; a.st2[0] = st2; a.out[0] = out; a.gout[0] = gout;
  a.g1[0] = g1; a.g2[0] = g2; a.nsets = 1;
  return a;
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

struct P4 { const float* f[4]; const float* gn[4]; float* gf[4]; };

// blockIdx.y selects the tensor (up to 4 of one shape per launch: both operands of both directions);
// stats of tensor k live at stats + k * 2 * planes
__global__ __launch_bounds__(256) void plane_moments_kernel(P4 a, float* __restrict__ stats_all, int S) {
  __shared__ double red[256];
  const float* __restrict__ f = a.f[blockIdx.y];
  float* __restrict__ stats = stats_all + (size_t)blockIdx.y * 2 * gridDim.x;
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
__global__ __launch_bounds__(256) void plane_norm_bwd_kernel(P4 t4, const float* __restrict__ stats_all, int S) {
  __shared__ double red[256];
  const float* __restrict__ f = t4.f[blockIdx.y];
  const float* __restrict__ gn = t4.gn[blockIdx.y];
  float* __restrict__ gf = t4.gf[blockIdx.y];
  if (gf == nullptr) return;  // uniform per block
  const float* __restrict__ stats = stats_all + (size_t)blockIdx.y * 2 * gridDim.x;
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
  return run_fwd(one_set(f1, f2, nullptr, nullptr, out, nullptr, nullptr, nullptr), B, C, H, W, max_displacement,
                 (hipStream_t)stream);
}

extern "C" int fs_corr2d_bwd(const float* f1, const float* f2, const float* grad_out, float* grad_f1,
                             float* grad_f2, int B, int C, int H, int W, int max_displacement,
                             fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(grad_out);
  if (grad_f1 == nullptr && grad_f2 == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_bwd(one_set(f1, f2, nullptr, nullptr, nullptr, grad_out, grad_f1, grad_f2), B, C, H, W,
                 max_displacement, (hipStream_t)stream);
}

extern "C" int fs_plane_moments(const float* f, float* stats, int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f); FS_REQUIRE_PTR(stats);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;  // unbiased variance needs two samples
  P4 a = {};
  a.f[0] = f;
  hipLaunchKernelGGL(plane_moments_kernel, dim3(planes, 1), dim3(256), 0, (hipStream_t)stream, a, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_plane_norm_bwd(const float* f, const float* stats, const float* grad_n, float* grad_f,
                                 int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f); FS_REQUIRE_PTR(stats); FS_REQUIRE_PTR(grad_n); FS_REQUIRE_PTR(grad_f);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;
  P4 a = {};
  a.f[0] = f; a.gn[0] = grad_n; a.gf[0] = grad_f;
  hipLaunchKernelGGL(plane_norm_bwd_kernel, dim3(planes, 1), dim3(256), 0, (hipStream_t)stream, a, stats, S);
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
  return run_fwd(one_set(f1, f2, stats1, stats2, out, nullptr, nullptr, nullptr), B, C, H, W, max_displacement,
                 (hipStream_t)stream);
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
  return run_bwd(one_set(f1, f2, stats1, stats2, nullptr, grad_out, grad_n1, grad_n2), B, C, H, W,
                 max_displacement, (hipStream_t)stream);
}

// ---- both directions of one pyramid level per launch (UPFlow/model/upflow.py:649 and :652) -----------
// Set a = (f1a, f2a) -> outa, set b = (f1b, f2b) -> outb, identical shapes.  `stats` (nullable) = the four
// (mean, rstd) tables [4][B*C][2] of (f1a, f2a, f1b, f2b) as fs_plane_moments4 writes them: the
// normalize_features-folded variant.  The five levels of a step cannot share a launch: level l's features are
// warped with the flow estimated at level l-1 (upflow.py:621-633).
extern "C" int fs_corr2d_pair_fwd(const float* f1a, const float* f2a, const float* f1b, const float* f2b,
                                  const float* stats, float* outa, float* outb, int B, int C, int H, int W,
                                  int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1a); FS_REQUIRE_PTR(f2a); FS_REQUIRE_PTR(f1b); FS_REQUIRE_PTR(f2b);
  FS_REQUIRE_PTR(outa); FS_REQUIRE_PTR(outb);
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  const size_t n = (size_t)2 * B * C;
  C2Set a = {};
  a.nsets = 2;
  a.f1[0] = f1a; a.f2[0] = f2a; a.out[0] = outa;
  a.f1[1] = f1b; a.f2[1] = f2b; a.out[1] = outb;
  if (stats != nullptr) { a.st1[0] = stats; a.st2[0] = stats + n; a.st1[1] = stats + 2 * n; a.st2[1] = stats + 3 * n; }
  return run_fwd(a, B, C, H, W, max_displacement, (hipStream_t)stream);
}

// Gradients of both sets (each pointer nullable, at least one non-null); with `stats` they are the gradients
// w.r.t. the NORMALISED maps (chain them with fs_plane_norm_bwd4).
extern "C" int fs_corr2d_pair_bwd(const float* f1a, const float* f2a, const float* f1b, const float* f2b,
                                  const float* stats, const float* gouta, const float* goutb, float* g1a,
                                  float* g2a, float* g1b, float* g2b, int B, int C, int H, int W,
                                  int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1a); FS_REQUIRE_PTR(f2a); FS_REQUIRE_PTR(f1b); FS_REQUIRE_PTR(f2b);
  FS_REQUIRE_PTR(gouta); FS_REQUIRE_PTR(goutb);
  if (g1a == nullptr && g2a == nullptr && g1b == nullptr && g2b == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  const size_t n = (size_t)2 * B * C;
  C2Set a = {};
  a.nsets = 2;
  a.f1[0] = f1a; a.f2[0] = f2a; a.gout[0] = gouta; a.g1[0] = g1a; a.g2[0] = g2a;
  a.f1[1] = f1b; a.f2[1] = f2b; a.gout[1] = goutb; a.g1[1] = g1b; a.g2[1] = g2b;
  if (stats != nullptr) { a.st1[0] = stats; a.st2[0] = stats + n; a.st1[1] = stats + 2 * n; a.st2[1] = stats + 3 * n; }
  return run_bwd(a, B, C, H, W, max_displacement, (hipStream_t)stream);
}

// (mean, rstd) of every (b, c) plane of four tensors of one shape in one launch: stats [4][planes][2]
extern "C" int fs_plane_moments4(const float* fa, const float* fb, const float* fc, const float* fd, float* stats,
                                 int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(fa); FS_REQUIRE_PTR(fb); FS_REQUIRE_PTR(fc); FS_REQUIRE_PTR(fd); FS_REQUIRE_PTR(stats);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;
  P4 a = {};
  a.f[0] = fa; a.f[1] = fb; a.f[2] = fc; a.f[3] = fd;
  hipLaunchKernelGGL(plane_moments_kernel, dim3(planes, 4), dim3(256), 0, (hipStream_t)stream, a, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// grad_f[k] = adjoint of (f - mean) * rstd applied to grad_n[k], k = 0..3 (a null grad_f[k] skips tensor k)
extern "C" int fs_plane_norm_bwd4(const float* fa, const float* fb, const float* fc, const float* fd,
                                  const float* stats, const float* gna, const float* gnb, const float* gnc,
                                  const float* gnd, float* gfa, float* gfb, float* gfc, float* gfd, int planes,
                                  int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(stats);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;
  P4 a = {};
  const float* f[4] = {fa, fb, fc, fd};
  const float* gn[4] = {gna, gnb, gnc, gnd};
  float* gf[4] = {gfa, gfb, gfc, gfd};
  for (int k = 0; k < 4; ++k) {
    if (gf[k] != nullptr && (f[k] == nullptr || gn[k] == nullptr)) return FS_ERR_NULLPTR;
    a.f[k] = f[k]; a.gn[k] = gn[k]; a.gf[k] = gf[k];
  }
  hipLaunchKernelGGL(plane_norm_bwd_kernel, dim3(planes, 4), dim3(256), 0, (hipStream_t)stream, a, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
