// laplacian3d.hip -- 3-D Laplacian-pyramid L1 loss for gfx950 (second half: "a real 3-D
// version replacing the scipy CPU round-trip").
//
// PARITY UNPINNED.  The reference's Flow-3D/model/laplacian.py:37-91 is dead code (commented out of
// Model.update, Flow-3D/model/RIFE.py:126,132) and not a usable definition: its conv_gauss (:44-58) ignores
// the kernel argument, round-trips through scipy.ndimage.gaussian_filter(sigma=1) on the CPU over ALL five
// axes (batch and channel included) and detaches the result, so no gradient flows through the pyramid.  What
// is built here is the 3-D analogue of Flow-2D/model/laplacian.py:10-88 (the version that does run), which
// the 3-D file was evidently derived from:
//
//   filtered = G3(reflect-pad 2)(cur),  G3 = g (x) g (x) g,  g = [1,4,6,4,1]/16     (the 5x5 kernel's 1-D factor)
//   down     = filtered[::2, ::2, ::2]
//   up       = (8 * G3)(reflect-pad 2)(zero-interleave(down)) cropped to cur's extent  (laplacian.py:24-42:
//              the 3-D file multiplies the interleaved volume by 8 = 2^3, the gain that keeps the mean)
//   pyr[l]   = cur - up;   cur = down;     loss = sum_l mean | pyr_l(input) - pyr_l(target) |
//
// The oracle is this build's own restatement (oracle/ifnet_ref.py::lap_loss3d, stock F.pad + F.conv3d).
// Structure = laplacian.hip with a third axis: one pyramid of (input - target) (every step is linear), per
// level one gather kernel for gauss+decimate and one for interleave+gauss+subtract+|.| partial sums + sign,
// and two gather-form adjoint kernels backward whose tap weights come from evaluating the forward index map;
// deterministic reductions, no atomics.
#include "common.hpp"

namespace {

constexpr int kMaxLevels = 8;
constexpr int kMaxBlocks = 1024;

struct Lap3Dims {
  int levels;
  int D[kMaxLevels + 1], H[kMaxLevels + 1], W[kMaxLevels + 1];
  long long cur_off[kMaxLevels + 1];
  long long sgn_total, down_total;
  int nblk[kMaxLevels], blk_off[kMaxLevels], blk_total;
};

int make_dims(Lap3Dims& d, int N, int D, int H, int W, int levels) {
  if (N < 1 || levels < 1 || levels > kMaxLevels) return FS_ERR_SHAPE;
  d.levels = levels;
  d.sgn_total = d.down_total = 0;
  d.blk_total = 0;
  for (int l = 0; l <= levels; ++l) {
    d.D[l] = D; d.H[l] = H; d.W[l] = W;
    if (l < levels) {
      if (D < 3 || H < 3 || W < 3) return FS_ERR_SHAPE;  // reflect padding by 2
      const long long n = (long long)N * D * H * W;
      if (n >= (1ll << 31)) return FS_ERR_SHAPE;
      d.cur_off[l] = d.sgn_total;
      d.sgn_total += n;
      const long long want = (n + 255) / 256;
      d.nblk[l] = (int)(want < kMaxBlocks ? want : kMaxBlocks);
      d.blk_off[l] = d.blk_total;
      d.blk_total += d.nblk[l];
      D = (D + 1) / 2; H = (H + 1) / 2; W = (W + 1) / 2;
      d.down_total += (long long)N * D * H * W;
    }
  }
  return FS_OK;
}

struct Ext { int D, H, W, Dd, Hd, Wd; };

__device__ __forceinline__ int reflect2(int t, int n) {
  t = t < 0 ? -t : t;
  return t >= n ? 2 * n - 2 - t : t;
}

__device__ __forceinline__ float gw(int i) {
  return (i == 2) ? 0.375f : ((i == 1 || i == 3) ? 0.25f : 0.0625f);
}

// down[n,r,p,q] = sum_ijk g_k g_i g_j v[n, rf(2r+k-2), rf(2p+i-2), rf(2q+j-2)],  v = a - b (b may be null)
__global__ __launch_bounds__(256) void lap3_down_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ down, int N, Ext e) {
  const long long total = (long long)N * e.Dd * e.Hd * e.Wd;
  const long long vol = (long long)e.D * e.H * e.W;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int q = (int)(id % e.Wd);
    long long t = id / e.Wd;
    const int p = (int)(t % e.Hd); t /= e.Hd;
    const int r = (int)(t % e.Dd);
    const long long n = t / e.Dd;
    const float* pa = a + n * vol;
    const float* pb = b ? b + n * vol : nullptr;
    int xs[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) xs[j] = reflect2(2 * q + j - 2, e.W);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const long long pl = (long long)reflect2(2 * r + k - 2, e.D) * e.H * e.W;
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const long long row = pl + (long long)reflect2(2 * p + i - 2, e.H) * e.W;
        float rs = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const float v = pa[row + xs[j]] - (pb ? pb[row + xs[j]] : 0.f);
          rs = fmaf(gw(j), v, rs);
        }
        ps = fmaf(gw(i), rs, ps);
      }
      acc = fmaf(gw(k), ps, acc);
    }
    down[id] = acc;
  }
}

// up at (z, y, x) of the zero-interleaved 2Dd x 2Hd x 2Wd volume, reflect-padded, with the 8x kernel
__device__ __forceinline__ float lap3_up_at(const float* __restrict__ dn, int z, int y, int x, const Ext& e) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int s = reflect2(z + k - 2, 2 * e.Dd);
    if (s & 1) continue;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = reflect2(y + i - 2, 2 * e.Hd);
      if (t & 1) continue;
      float rs = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int u = reflect2(x + j - 2, 2 * e.Wd);
        if (u & 1) continue;
        rs = fmaf(gw(j), dn[((long long)(s >> 1) * e.Hd + (t >> 1)) * e.Wd + (u >> 1)], rs);
      }
      ps = fmaf(gw(i), rs, ps);
    }
    acc = fmaf(gw(k), ps, acc);
  }
  return 8.0f * acc;
}

__global__ __launch_bounds__(256) void lap3_level_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ down, float* __restrict__ sgn,
                                                         float* __restrict__ part, int N, Ext e, float inv_numel) {
  const long long vol = (long long)e.D * e.H * e.W, dvol = (long long)e.Dd * e.Hd * e.Wd;
  const long long total = (long long)N * vol;
  float s = 0.f;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int x = (int)(id % e.W);
    long long t = id / e.W;
    const int y = (int)(t % e.H); t /= e.H;
    const int z = (int)(t % e.D);
    const long long n = t / e.D;
    const float v = a[id] - (b ? b[id] : 0.f);
    const float lap = v - lap3_up_at(down + n * dvol, z, y, x, e);
    s += fabsf(lap);
    sgn[id] = (lap > 0.f ? inv_numel : (lap < 0.f ? -inv_numel : (lap == 0.f ? 0.f : lap)));
  }
  fs::block_pair_to_ws(s * inv_numel, 0.f, part);
}

// weight with which fine index f (of n_f) reads interleaved coarse index c (of n_c) in the up-sampling
__device__ __forceinline__ float up_w(int f, int n_f, int c, int n_c) {
  if (f < 0 || f >= n_f) return 0.f;
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (reflect2(f + i - 2, 2 * n_c) == 2 * c) a += gw(i);
  return a;
}

// weight with which coarse index c (of n_c) reads fine index f (of n_f) in the gauss + decimate step
__device__ __forceinline__ float down_w(int c, int n_c, int f, int n_f) {
  if (c < 0 || c >= n_c) return 0.f;
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (reflect2(2 * c + i - 2, n_f) == f) a += gw(i);
  return a;
}

// G[n,r,p,q] (+)= -8 gs * sum_{z,y,x} [weight of down[r,p,q] in up[z,y,x]] * sgn[n,z,y,x]
__global__ __launch_bounds__(256) void lap3_up_adj_kernel(const float* __restrict__ sgn, float* __restrict__ G,
                                                          const float* __restrict__ gscale, int accumulate, int N,
                                                          Ext e) {
  const long long total = (long long)N * e.Dd * e.Hd * e.Wd;
  const long long vol = (long long)e.D * e.H * e.W;
  const float gs = *gscale;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int q = (int)(id % e.Wd);
    long long t = id / e.Wd;
    const int p = (int)(t % e.Hd); t /= e.Hd;
    const int r = (int)(t % e.Dd);
    const long long n = t / e.Dd;
    float wz[5], wy[5], wx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      wz[k] = up_w(2 * r - 2 + k, e.D, r, e.Dd);
      wy[k] = up_w(2 * p - 2 + k, e.H, p, e.Hd);
      wx[k] = up_w(2 * q - 2 + k, e.W, q, e.Wd);
    }
    const float* sp = sgn + n * vol;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (wz[k] == 0.f) continue;
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (wy[i] == 0.f) continue;
        const long long row = ((long long)(2 * r - 2 + k) * e.H + (2 * p - 2 + i)) * e.W;
        float rs = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (wx[j] == 0.f) continue;
          rs = fmaf(wx[j], sp[row + 2 * q - 2 + j], rs);
        }
        ps = fmaf(wy[i], rs, ps);
      }
      acc = fmaf(wz[k], ps, acc);
    }
    const float base = accumulate ? G[id] : 0.f;
    G[id] = base - 8.0f * gs * acc;
  }
}

// gcur[n,z,y,x] = gs * sgn[n,z,y,x] + sum_{r,p,q} [weight of cur[z,y,x] in down[r,p,q]] * G[n,r,p,q]
__global__ __launch_bounds__(256) void lap3_down_adj_kernel(const float* __restrict__ sgn,
                                                            const float* __restrict__ G, float* __restrict__ gcur,
                                                            const float* __restrict__ gscale, int N, Ext e) {
  const long long vol = (long long)e.D * e.H * e.W, dvol = (long long)e.Dd * e.Hd * e.Wd;
  const long long total = (long long)N * vol;
  const float gs = *gscale;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int x = (int)(id % e.W);
    long long t = id / e.W;
    const int y = (int)(t % e.H); t /= e.H;
    const int z = (int)(t % e.D);
    const long long n = t / e.D;
    const int r0 = (z >> 1) - 2, p0 = (y >> 1) - 2, q0 = (x >> 1) - 2;
    float wz[5], wy[5], wx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      wz[k] = down_w(r0 + k, e.Dd, z, e.D);
      wy[k] = down_w(p0 + k, e.Hd, y, e.H);
      wx[k] = down_w(q0 + k, e.Wd, x, e.W);
    }
    const float* gp = G + n * dvol;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (wz[k] == 0.f) continue;
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (wy[i] == 0.f) continue;
        const long long row = ((long long)(r0 + k) * e.Hd + (p0 + i)) * e.Wd;
        float rs = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (wx[j] == 0.f) continue;
          rs = fmaf(wx[j], gp[row + q0 + j], rs);
        }
        ps = fmaf(wy[i], rs, ps);
      }
      acc = fmaf(wz[k], ps, acc);
    }
    gcur[id] = gs * sgn[id] + acc;
  }
}

unsigned grid_for(long long total) {
  const long long want = (total + 255) / 256;
  return (unsigned)(want < 65536 ? want : 65536);
}

Ext ext_of(const Lap3Dims& d, int l) { return {d.D[l], d.H[l], d.W[l], d.D[l + 1], d.H[l + 1], d.W[l + 1]}; }

}  // namespace

extern "C" int fs_laploss3d_sizes(int N, int D, int H, int W, int levels, long long* sgn_floats,
                                  long long* ws_fwd_floats, long long* ws_bwd_floats) {
  Lap3Dims d;
  const int rc = make_dims(d, N, D, H, W, levels);
  if (rc != FS_OK) return rc;
  if (sgn_floats) *sgn_floats = d.sgn_total;
  if (ws_fwd_floats) *ws_fwd_floats = d.down_total + 2ll * d.blk_total;
  if (ws_bwd_floats) *ws_bwd_floats = d.down_total;
  return FS_OK;
}

extern "C" int fs_laploss3d_fwd(const float* input, const float* target, float* sgn, float* ws, float* loss,
                                int N, int D, int H, int W, int levels, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(input); FS_REQUIRE_PTR(sgn); FS_REQUIRE_PTR(ws); FS_REQUIRE_PTR(loss);
  Lap3Dims d;
  const int rc = make_dims(d, N, D, H, W, levels);
  if (rc != FS_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* part = ws + d.down_total;
  const float* cur = input;
  const float* sub = target;
  float* down = ws;
  for (int l = 0; l < levels; ++l) {
    const Ext e = ext_of(d, l);
    const long long ncur = (long long)N * e.D * e.H * e.W, ndown = (long long)N * e.Dd * e.Hd * e.Wd;
    hipLaunchKernelGGL(lap3_down_kernel, dim3(grid_for(ndown)), dim3(256), 0, st, cur, sub, down, N, e);
    hipLaunchKernelGGL(lap3_level_kernel, dim3(d.nblk[l]), dim3(256), 0, st, cur, sub, down, sgn + d.cur_off[l],
                       part + 2 * d.blk_off[l], N, e, 1.0f / (float)ncur);
    cur = down;
    sub = nullptr;
    down += ndown;
  }
  hipLaunchKernelGGL(fs::reduce_final_kernel, dim3(1), dim3(256), 0, st, part, d.blk_total, loss);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_laploss3d_bwd(const float* sgn, const float* grad_loss, float* ws, float* grad_diff, int N,
                                int D, int H, int W, int levels, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(sgn); FS_REQUIRE_PTR(grad_loss); FS_REQUIRE_PTR(ws); FS_REQUIRE_PTR(grad_diff);
  Lap3Dims d;
  const int rc = make_dims(d, N, D, H, W, levels);
  if (rc != FS_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  long long goff[kMaxLevels + 1];
  goff[0] = 0;
  long long o = 0;
  for (int l = 0; l < levels; ++l) {
    goff[l + 1] = o;
    o += (long long)N * d.D[l + 1] * d.H[l + 1] * d.W[l + 1];
  }
  for (int l = levels - 1; l >= 0; --l) {
    const Ext e = ext_of(d, l);
    float* G = ws + goff[l + 1];
    hipLaunchKernelGGL(lap3_up_adj_kernel, dim3(grid_for((long long)N * e.Dd * e.Hd * e.Wd)), dim3(256), 0, st,
                       sgn + d.cur_off[l], G, grad_loss, (l == levels - 1) ? 0 : 1, N, e);
    float* gcur = (l == 0) ? grad_diff : ws + goff[l];
    hipLaunchKernelGGL(lap3_down_adj_kernel, dim3(grid_for((long long)N * e.D * e.H * e.W)), dim3(256), 0, st,
                       sgn + d.cur_off[l], G, gcur, grad_loss, N, e);
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}
