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
// Tiled kernels (corr2d_fwd_q_kernel / corr2d_bwd_q_kernel, described at their definitions): one workgroup = one
// 8 x 32-pixel tile, lane = a quad of 4 consecutive pixels, the search window (tile + md halo) staged through LDS
// by unconditional 16-byte buffer loads at dword alignment, packed-FP32 FMAs on aligned register pairs, 16-byte
// stores, tiles dealt to the XCDs in contiguous ranges.  Both are bound by VALU instruction issue; HBM traffic is
// the algorithmic 4*(2C + (2md+1)^2) B/pixel forward, 4*(4C + (2md+1)^2) backward; halo re-reads hit L2.
// They need pixels: the coarsest UPFlow levels at C3 are (C, h, w) = (196, 3, 8) and (128, 5, 15), 24 / 75 pixels
// per sample, C >> pixels, where an 8 x 32 tile is mostly empty and the forward kernel is a serial chain of C/8
// stage-barrier-compute rounds.  Those run the direct kernels instead (thresholds: kSmall* below): no LDS, no
// barrier, one thread per (sample, displacement, pixel) output and channel slice -- the C range is split over
// 1..8 lane groups of a wave and reduced with wave shuffles -- so that even a B = 2 launch fills the chip.
//
// Backward.  grad_f1[c,p] = (1/C) sum_d g[d,p] f2[c,p+d] is a gather;  grad_f2 is the same gather
// with the roles swapped and the displacement negated:  grad_f2[c,q] = (1/C) sum_d gT[d,q] f1[c,q+d]
// with gT[d,q] = g[-d, q+d] (0 when q+d is outside).  One launch computes both, no atomics, bitwise reproducible.
#include "common.hpp"

namespace {

#include "corr_q.hpp"

// ---- coarse pyramid levels: direct kernels (thresholds: kSmall* below) ------------------------------------
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

// direct kernels up to this many pixels per sample (C3 levels (196, 3, 8) / (128, 5, 15) / (96, 10, 29), both
// directions per launch, direct vs tiled: forward 6.7 vs - / 16 vs 31 / 50 vs 26 us, with the normalisation folded
// in 15 vs - / 68 vs 34 / 152 vs 28 us; backward 17 vs - / 42 vs 26 / 186 vs 38 us)
constexpr int kSmallFwd = 128, kSmallFwdNorm = 32, kSmallBwd = 32;

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
  if (H * W <= (a.st1[0] != nullptr ? kSmallFwdNorm : kSmallFwd)) return launch_small_fwd<MD>(a, B, C, H, W, st);
  const dim3 grid((unsigned)((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * B * a.nsets));
  constexpr int DPW = FS_C2_FWD_DPW;
  hipLaunchKernelGGL((corr_fwd_q_kernel<MD, DPW>), grid, dim3(64 * ((2 * MD + DPW) / DPW)), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_bwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  if (H * W <= kSmallBwd) return launch_small_bwd<MD>(a, B, C, H, W, st);
  const dim3 grid((unsigned)((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * 2 * B * a.nsets * fs::cdiv(C, CBW)));
  hipLaunchKernelGGL(corr_bwd_q_kernel<MD>, grid, dim3(512), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int check_shape(int B, int C, int H, int W, int md) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return FS_ERR_SHAPE;
  if ((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * 4 * B * fs::cdiv(C, CBW) >= (1ll << 31)) return FS_ERR_SHAPE;  // 1-D grids
  if ((long long)B * C * H * W >= (1ll << 29) || 81ll * H * W >= (1ll << 29)) return FS_ERR_SHAPE;  // 32-bit byte offsets inside a tensor
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
  a.g1[0] = g1; a.g2[0] = g2; a.nsets = 1; a.D = 1; a.NZ = 1;
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
  a.nsets = 2; a.D = 1; a.NZ = 1;
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
  a.nsets = 2; a.D = 1; a.NZ = 1;
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
