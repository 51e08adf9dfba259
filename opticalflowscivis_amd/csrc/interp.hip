// interp.hip -- adjoint (backward) of the trilinear resizes inside IFBlock for gfx950.
//
// IFBlock (Flow-3D/model/IFNet.py:80-120) resizes its input and flow down by `scale` and its
// outputs back up by `scale` with F.interpolate(mode="trilinear", align_corners=False,
// scale_factor given => source index = (dst + 0.5) / scale_factor - 0.5, clamped at 0).
// ATen's backward (upsample_trilinear3d_backward_out_frame) scatters with atomics and takes 14 ms per
// call on a [2,6,256^3] flow -- 27 % of the whole train step once the convolutions are fixed.
//
// Here the adjoint is a GATHER: one thread owns one INPUT voxel and sums the output gradients that
// reference it.  The interpolation weights are separable, so the thread computes NC candidate weights
// per axis from the forward formula itself (exactly the forward's index / lambda arithmetic: no
// separate derivation to get wrong at the borders) and runs an NC^3 loop with lanes on x.
//   integer up-sampling by s  : candidates o in [s*i - s/2, s*i + 3s/2)   -> NC = 2s
//   integer down-sampling by s: candidates o in [i/s - 1, i/s + 1]        -> NC = 3
// No atomics, bitwise reproducible.  HBM: reads grad_out once (overlapping windows are L1/L2 hits),
// writes grad_in once.
//
// Forward resizes: `fs_downsample3d_fwd` (IFBlock's input / flow down-scaling,
// Flow-3D/model/IFNet.py:85,88, with the `* 1/scale` of the flow folded in), `fs_upsample3d_scale_add`
// (:118-119 with the running-flow accumulation) and the 2-D bilinear pair `fs_resize2d_{fwd,bwd}`
// (Flow-2D/model/IFNet.py:89,92,115-116) all evaluate ATen's area_pixel_compute_source_index /
// lambda arithmetic in ATen's summation order with FMA contraction off (down-scaling: every lambda is 1/2 and
// the result is bit-identical to F.interpolate; up-scaling: within an ulp of ATen's builds, which may contract).
#include <cstdlib>

#include "common.hpp"

namespace {

struct IP {
  int Di, Hi, Wi;   // input (= grad_in) extent
  int Do, Ho, Wo;   // output (= grad_out) extent
  float rs;         // source-index scale = 1 / scale_factor
  int up;           // 1: up-sampling by s, 0: down-sampling by s
  int s;
  long long nBC;    // B*C
  // optional per-channel input planes (fs_downsample3d_fwd_ms: the C <= 12 channels of `small` live in different
  // tensors -- IFBlock's torch.cat in front of its down-sampling, Flow-3D/model/IFNet.py:183 + :85): src[c] =
  // channel c of sample 0, sbs[c] = that tensor's batch stride in floats.  nsrc = 0: one tensor.
  int C, nsrc;
  const float* src[12];
  long long sbs[12];
};

// start of input plane bc = b * C + c
__device__ __forceinline__ const float* in_plane(const float* __restrict__ small, const IP& p, long long bc, long long nin) {
  if (p.nsrc == 0) return small + bc * nin;
  const int b = (int)(bc / p.C), c = (int)(bc - (long long)b * p.C);
  const float* q = p.src[0];
  long long st = p.sbs[0];
#pragma unroll
  for (int i = 1; i < 12; ++i)
    if (i < p.nsrc && c == i) { q = p.src[i]; st = p.sbs[i]; }  // (selects, not an indexed read of the arguments)
  return q + (long long)b * st;
}

// weight with which output index o reads input index i along one axis (ATen
// area_pixel_compute_source_index + the i0/i1/lambda of upsample_trilinear3d)
__device__ __forceinline__ float axis_w(int o, int i, int n_in, int n_out, float rs) {
  if (o < 0 || o >= n_out) return 0.f;
  float src = rs * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  const int i0 = (int)src;
  const int i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
  const float l1 = src - (float)i0;
  return ((i0 == i) ? (1.0f - l1) : 0.f) + ((i1 == i) ? l1 : 0.f);
}

template <int NC>
__global__ __launch_bounds__(256) void interp3d_adjoint_kernel(const float* __restrict__ gout,
                                                               float* __restrict__ gin, IP p) {
  const long long nin = (long long)p.Di * p.Hi * p.Wi;
  const long long nout = (long long)p.Do * p.Ho * p.Wo;
  const long long total = p.nBC * nin;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bc = e / nin;
    const int r = (int)(e - bc * nin);
    const int x = r % p.Wi, y = (r / p.Wi) % p.Hi, z = r / (p.Wi * p.Hi);
    const int oz0 = p.up ? p.s * z - p.s / 2 : z / p.s - 1;
    const int oy0 = p.up ? p.s * y - p.s / 2 : y / p.s - 1;
    const int ox0 = p.up ? p.s * x - p.s / 2 : x / p.s - 1;
    float wz[NC], wy[NC], wx[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      wz[k] = axis_w(oz0 + k, z, p.Di, p.Do, p.rs);
      wy[k] = axis_w(oy0 + k, y, p.Hi, p.Ho, p.rs);
      wx[k] = axis_w(ox0 + k, x, p.Wi, p.Wo, p.rs);
    }
    const float* g = gout + bc * nout;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < NC; ++a) {
      if (wz[a] == 0.f) continue;
      const int oz = oz0 + a;
#pragma unroll
      for (int b = 0; b < NC; ++b) {
        const float wzy = wz[a] * wy[b];
        if (wzy == 0.f) continue;
        const float* row = g + ((long long)oz * p.Ho + (oy0 + b)) * p.Wo;
        float rowsum = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int ox = min(max(ox0 + c, 0), p.Wo - 1);  // weight is 0 where the index was clamped
          rowsum = fmaf(wx[c], row[ox], rowsum);
        }
        acc = fmaf(wzy, rowsum, acc);
      }
    }
    gin[e] = acc;
  }
}

// ---- fast paths ---------------------------------------------------------------------------
// Down-sampling by s with in == s * out: output o reads exactly the two inputs s*o + s/2 - 1 and
// s*o + s/2 with weight 1/2 per axis, so a fine voxel has at most one contributor: pure streaming.
__global__ __launch_bounds__(256) void interp3d_down_adjoint_exact(const float* __restrict__ gout,
                                                                  float* __restrict__ gin, IP p) {
  const long long nin = (long long)p.Di * p.Hi * p.Wi;
  const long long nout = (long long)p.Do * p.Ho * p.Wo;
  const long long total = p.nBC * nin;
  const int lo = p.s / 2 - 1, hi = p.s / 2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bc = e / nin;
    const int r = (int)(e - bc * nin);
    const int x = r % p.Wi, y = (r / p.Wi) % p.Hi, z = r / (p.Wi * p.Hi);
    const int rx = x % p.s, ry = y % p.s, rz = z % p.s;
    float v = 0.f;
    if ((rx == lo || rx == hi) && (ry == lo || ry == hi) && (rz == lo || rz == hi))
      v = 0.125f * gout[bc * nout + ((long long)(z / p.s) * p.Ho + (y / p.s)) * p.Wo + (x / p.s)];
    gin[e] = v;
  }
}

// Same, four consecutive x per thread (Wi % 4 == 0): the row test and the index arithmetic are paid
// once per float4 store -- the scalar version is bound by its integer divisions, not by HBM.
// IT: index type of the element decomposition -- unsigned when the float4 count fits 31 bits (three 64-bit divisions
// per store made this streaming kernel VALU-bound: a vector instruction issuing in 86 % of its cycles, round-3 PMC)
template <typename IT>
__global__ __launch_bounds__(256) void interp3d_down_adjoint_exact_v4(const float* __restrict__ gout,
                                                                     float4* __restrict__ gin, IP p, float scale) {
  const IT W4 = (IT)(p.Wi >> 2);
  const IT rows = (IT)(p.nBC * p.Di * p.Hi);
  const IT total = rows * W4;
  const int lo = p.s / 2 - 1, hi = p.s / 2;
  for (IT e = (IT)blockIdx.x * 256 + threadIdx.x; e < total; e += (IT)gridDim.x * 256) {
    const IT row = e / W4;
    const int x4 = (int)(e - row * W4);
    const int y = (int)(row % (IT)p.Hi);
    const IT t = row / (IT)p.Hi;
    const int z = (int)(t % (IT)p.Di);
    const long long bc = (long long)(t / (IT)p.Di);
    const int ry = y % p.s, rz = z % p.s;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((ry == lo || ry == hi) && (rz == lo || rz == hi)) {
      const float* g = gout + ((bc * p.Do + z / p.s) * p.Ho + y / p.s) * p.Wo;
      if (p.s == 2) {
        const float g0 = 0.125f * g[2 * x4] * scale, g1 = 0.125f * g[2 * x4 + 1] * scale;
        v = make_float4(g0, g0, g1, g1);
      } else {  // s == 4: residues 1 and 2 of each group of four
        const float g0 = 0.125f * g[x4] * scale;
        v = make_float4(0.f, g0, g0, 0.f);
      }
    }
    gin[e] = v;
  }
}

// Up-sampling adjoint as three separable 1-D passes (x, then y, then z): every fine gradient is
// read from HBM once and the per-voxel gather count drops from NC^3 to NC per pass (the direct
// kernel is bound by the 4-lanes-per-clock address path, not by HBM).
//   g [outer, n_out, inner] -> out [outer, n_in, inner],  out[., i, .] = sum_o w(o, i) g[., o, .]
template <int NC>
__global__ __launch_bounds__(256) void interp_axis_adjoint_kernel(const float* __restrict__ g,
                                                                  float* __restrict__ out,
                                                                  long long outer, int n_out, int n_in,
                                                                  int inner, int s, float rs, float oscale) {
  const long long total = outer * n_in * inner;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int in_i = (int)(e % inner);
    const long long t = e / inner;
    const int i = (int)(t % n_in);
    const long long ou = t / n_in;
    const int o0 = s * i - s / 2;
    const float* col = g + (ou * n_out) * inner + in_i;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float w = axis_w(o0 + k, i, n_in, n_out, rs);
      const int o = min(max(o0 + k, 0), n_out - 1);
      acc = fmaf(w, col[(long long)o * inner], acc);
    }
    out[e] = acc * oscale;
  }
}

// ---- forward: out = prev + scale * trilinear_upsample(small) -------------------------------------
// IFBlock's `flow = flow + interpolate(flow_d, scale_factor=s) * s` and `mask = mask + interpolate(mask_d, s)`
// (Flow-3D/model/IFNet.py:118-119 with :213-214 / :228-229) in one pass: write the sum once instead of
// materialising the upsampled tensor, scaling it and adding it (4.8 GB -> 1.6 GB for a 256^3 flow).
// Index / weight arithmetic and summation order are ATen's upsample_trilinear3d_out_frame.
__global__ __launch_bounds__(256) void upsample3d_scale_add_kernel(const float* __restrict__ small,
                                                                   const float* __restrict__ prev,
                                                                   float* __restrict__ out, IP p, float scale) {
#pragma clang fp contract(off)
  const long long nin = (long long)p.Di * p.Hi * p.Wi;
  const long long nout = (long long)p.Do * p.Ho * p.Wo;
  const long long total = p.nBC * nout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bc = e / nout;
    const int r = (int)(e - bc * nout);
    const int x = r % p.Wo, y = (r / p.Wo) % p.Ho, z = r / (p.Wo * p.Ho);
    float sz = p.rs * ((float)z + 0.5f) - 0.5f, sy = p.rs * ((float)y + 0.5f) - 0.5f,
          sx = p.rs * ((float)x + 0.5f) - 0.5f;
    sz = sz < 0.f ? 0.f : sz; sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    const int z0 = (int)sz, y0 = (int)sy, x0 = (int)sx;
    const int zp = (z0 < p.Di - 1) ? 1 : 0, yp = (y0 < p.Hi - 1) ? 1 : 0, xp = (x0 < p.Wi - 1) ? 1 : 0;
    const float lz1 = sz - (float)z0, ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float lz0 = 1.f - lz1, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* s = in_plane(small, p, bc, nin) + ((long long)z0 * p.Hi + y0) * p.Wi + x0;
    const int dy = yp * p.Wi, dz = zp * p.Hi * p.Wi;
    const float v =
        lz0 * (ly0 * (lx0 * s[0] + lx1 * s[xp]) + ly1 * (lx0 * s[dy] + lx1 * s[dy + xp])) +
        lz1 * (ly0 * (lx0 * s[dz] + lx1 * s[dz + xp]) + ly1 * (lx0 * s[dz + dy] + lx1 * s[dz + dy + xp]));
    const float up = v * scale;
    out[e] = prev ? prev[e] + up : up;
  }
}

// Four consecutive x per thread (Wo % 4 == 0, 16-byte aligned rows): the z / y weights and the row
// pointers are shared, prev / out move as float4 -- the scalar form is bound by its index arithmetic.
__global__ __launch_bounds__(256) void upsample3d_scale_add_v4_kernel(const float* __restrict__ small,
                                                                      const float4* __restrict__ prev,
                                                                      float4* __restrict__ out, IP p, float scale) {
#pragma clang fp contract(off)
  const int W4 = p.Wo >> 2;
  const long long rows = p.nBC * p.Do * p.Ho;
  const long long total = rows * W4;
  const long long nin = (long long)p.Di * p.Hi * p.Wi;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long row = e / W4;
    const int x4 = (int)(e - row * W4);
    const int y = (int)(row % p.Ho);
    const long long t = row / p.Ho;
    const int z = (int)(t % p.Do);
    const long long bc = t / p.Do;
    int z0, zp, y0, yp;
    float lz0, lz1, ly0, ly1;
    fs::trilinear_axis(z, p.Di, p.rs, z0, zp, lz0, lz1);
    fs::trilinear_axis(y, p.Hi, p.rs, y0, yp, ly0, ly1);
    const float* s00 = in_plane(small, p, bc, nin) + ((long long)z0 * p.Hi + y0) * p.Wi;
    const float* s01 = s00 + yp * p.Wi;
    const float* s10 = s00 + (long long)zp * p.Hi * p.Wi;
    const float* s11 = s10 + yp * p.Wi;
    float o[4];
    fs::trilinear_up_row<4>(s00, s01, s10, s11, lz0, lz1, ly0, ly1, p.rs, 4 * x4, p.Wi, scale, o);
    float4 r = make_float4(o[0], o[1], o[2], o[3]);
    if (prev) {
      const float4 q = prev[e];
      r.x = q.x + r.x; r.y = q.y + r.y; r.z = q.z + r.z; r.w = q.w + r.w;
    }
    out[e] = r;
  }
}

// x F up-sampling (F = 2, 4) through an LDS-staged source brick (round 4).  The kernel above reads 8 corners per output --
// 32 dword loads per thread through the texture addresser for 16 bytes written: 0.23 of the HBM roof.  Here a workgroup
// owns 8 z x 8 y x 64 x outputs; their (8 / F + 2)^2 x (64 / F + 2) source voxels are staged once (coalesced), every
// thread then forms four consecutive x of four (z, y) rows from LDS with the SAME function in the same order
// (fs::trilinear_up_row: bit-identical to the kernel above and to F.interpolate), adds `prev`, stores 16 bytes.
template <int F>
__global__ __launch_bounds__(256) void upsample3d_scale_add_tile_kernel(const float* __restrict__ small,
                                                                        const float4* __restrict__ prev,
                                                                        float4* __restrict__ out, IP p, float scale) {
#pragma clang fp contract(off)
  constexpr int OZ = 8, OY = 8, OX = 64;
  constexpr int SZ = OZ / F + 2, SY = OY / F + 2, SX = OX / F + 2;
  __shared__ float sm[SZ * SY * SX];
  long long tile = blockIdx.x;
  const int txn = p.Wo / OX, tyn = p.Ho / OY, tzn = p.Do / OZ;
  const int txi = (int)(tile % txn); tile /= txn;
  const int tyi = (int)(tile % tyn); tile /= tyn;
  const int tzi = (int)(tile % tzn);
  const long long bc = tile / tzn;
  const int z_a = tzi * OZ, y_a = tyi * OY, x_a = txi * OX;
  // first source index of the brick along each axis (the index ATen's arithmetic gives the tile's first output)
  int zs0, ys0, xs0, dummy;
  float l0, l1;
  fs::trilinear_axis(z_a, p.Di, p.rs, zs0, dummy, l0, l1);
  fs::trilinear_axis(y_a, p.Hi, p.rs, ys0, dummy, l0, l1);
  fs::trilinear_axis(x_a, p.Wi, p.rs, xs0, dummy, l0, l1);
  const long long nin = (long long)p.Di * p.Hi * p.Wi;
  const float* __restrict__ src = in_plane(small, p, bc, nin);
  for (int i = threadIdx.x; i < SZ * SY * SX; i += 256) {
    const int sz = i / (SY * SX), r = i - sz * (SY * SX);
    const int sy = r / SX, sx = r - sy * SX;
    const int gz = min(zs0 + sz, p.Di - 1), gy = min(ys0 + sy, p.Hi - 1), gx = min(xs0 + sx, p.Wi - 1);
    sm[i] = src[((long long)gz * p.Hi + gy) * p.Wi + gx];
  }
  __syncthreads();
  const int x4 = threadIdx.x & 15, yy = (threadIdx.x >> 4) & 7, zq = threadIdx.x >> 7;  // thread = (z quarter, y, 4 x)
  const int y = y_a + yy;
  int y0, yp;
  float ly0, ly1;
  fs::trilinear_axis(y, p.Hi, p.rs, y0, yp, ly0, ly1);
  const int W4 = p.Wo >> 2;
#pragma unroll
  for (int k = 0; k < OZ / 2; ++k) {
    const int z = z_a + 2 * k + zq;
    int z0, zp;
    float lz0, lz1;
    fs::trilinear_axis(z, p.Di, p.rs, z0, zp, lz0, lz1);
    // row pointers such that [x0] with the ABSOLUTE source column x0 addresses the staged brick
    const float* s00 = sm + ((z0 - zs0) * SY + (y0 - ys0)) * SX - xs0;
    const float* s01 = s00 + yp * SX;
    const float* s10 = s00 + zp * SY * SX;
    const float* s11 = s10 + yp * SX;
    float o[4];
    fs::trilinear_up_row<4>(s00, s01, s10, s11, lz0, lz1, ly0, ly1, p.rs, x_a + 4 * x4, p.Wi, scale, o);
    float4 r = make_float4(o[0], o[1], o[2], o[3]);
    const long long e = ((bc * p.Do + z) * p.Ho + y) * W4 + (x_a >> 2) + x4;
    if (prev) {
      const float4 q = prev[e];
      r.x = q.x + r.x; r.y = q.y + r.y; r.z = q.z + r.z; r.w = q.w + r.w;
    }
    out[e] = r;
  }
}

// Exact / 2 and / 4 down-sampling, four consecutive outputs per thread (round 4).  The generic kernel above read its
// 8 corners per output with 32 dword loads per thread (TA ~85 % busy, 0.36 of the HBM roof); here the 8 (factor 2) or 16
// (factor 4) source floats of a row move as two float4 / four 8-byte loads (the pairs 4x+1, 4x+2 sit at 4-byte alignment).
// Same arithmetic: for in = s * out every x lambda is exactly 1/2 and x0 = s * x + (s / 2 - 1) (ATen's
// area_pixel_compute_source_index in fp32 is exact here), the z / y terms and the summation order are the generic
// kernel's -- bit-identical to it and to F.interpolate (tests/test_gpu_resize.py).
struct __attribute__((packed, aligned(4))) F2 { float a, b; };

template <int F>
__global__ __launch_bounds__(256) void downsample3d_v4_kernel(const float* __restrict__ in, float4* __restrict__ out, IP p,
                                                              float scale) {
#pragma clang fp contract(off)
  const int W4 = p.Wo >> 2;
  const long long rows = p.nBC * p.Do * p.Ho;
  const long long total = rows * W4;
  const long long nin = (long long)p.Di * p.Hi * p.Wi;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long row = e / W4;
    const int x4 = (int)(e - row * W4);
    const int y = (int)(row % p.Ho);
    const long long t = row / p.Ho;
    const int z = (int)(t % p.Do);
    const long long bc = t / p.Do;
    int z0, zp, y0, yp;
    float lz0, lz1, ly0, ly1;
    fs::trilinear_axis(z, p.Di, p.rs, z0, zp, lz0, lz1);
    fs::trilinear_axis(y, p.Hi, p.rs, y0, yp, ly0, ly1);
    const float* s00 = in_plane(in, p, bc, nin) + ((long long)z0 * p.Hi + y0) * p.Wi + (long long)F * 4 * x4;
    const float* rp[4] = {s00, s00 + yp * p.Wi, s00 + (long long)zp * p.Hi * p.Wi,
                          s00 + (long long)zp * p.Hi * p.Wi + yp * p.Wi};
    float a[4][4], b[4][4];  // [row][output]: the two x taps
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (F == 2) {
        const float4 u = *reinterpret_cast<const float4*>(rp[r]), v = *reinterpret_cast<const float4*>(rp[r] + 4);
        a[r][0] = u.x; b[r][0] = u.y; a[r][1] = u.z; b[r][1] = u.w;
        a[r][2] = v.x; b[r][2] = v.y; a[r][3] = v.z; b[r][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const F2 q = *reinterpret_cast<const F2*>(rp[r] + 4 * i + 1);
          a[r][i] = q.a; b[r][i] = q.b;
        }
      }
    }
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = lz0 * (ly0 * (0.5f * a[0][i] + 0.5f * b[0][i]) + ly1 * (0.5f * a[1][i] + 0.5f * b[1][i])) +
                      lz1 * (ly0 * (0.5f * a[2][i] + 0.5f * b[2][i]) + ly1 * (0.5f * a[3][i] + 0.5f * b[3][i]));
      o[i] = v * scale;
    }
    out[e] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- 2-D bilinear resize (Flow-2D IFBlock) -------------------------------------------------------
// out[bc, y, x] = scale * bilinear(in) with ATen's upsample_bilinear2d index arithmetic (align_corners=False,
// scale factor given: source = rs * (dst + 0.5) - 0.5 clamped at 0).  One thread per output pixel.
__global__ __launch_bounds__(256) void resize2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       long long nBC, int Hi, int Wi, int Ho, int Wo, float rs,
                                                       float scale) {
#pragma clang fp contract(off)
  const long long nout = (long long)Ho * Wo, nin = (long long)Hi * Wi;
  const long long total = nBC * nout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bc = e / nout;
    const int r = (int)(e - bc * nout);
    const int x = r % Wo, y = r / Wo;
    float sy = rs * ((float)y + 0.5f) - 0.5f, sx = rs * ((float)x + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = (y0 < Hi - 1) ? 1 : 0, xp = (x0 < Wi - 1) ? 1 : 0;
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* s = in + bc * nin + (long long)y0 * Wi + x0;
    const int dy = yp * Wi;
    // upsample_bilinear2d_out_frame: h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (...)
    const float v = ly0 * (lx0 * s[0] + lx1 * s[xp]) + ly1 * (lx0 * s[dy] + lx1 * s[dy + xp]);
    out[e] = (scale == 1.0f) ? v : v * scale;
  }
}

// adjoint as a gather: one thread per INPUT pixel sums the output gradients that reference it (weights
// from the forward's own index arithmetic, axis_w); NC candidates per axis as in interp3d_adjoint_kernel.
template <int NC>
__global__ __launch_bounds__(256) void resize2d_adjoint_kernel(const float* __restrict__ gout,
                                                               float* __restrict__ gin, long long nBC, int Hi,
                                                               int Wi, int Ho, int Wo, float rs, int up, int s,
                                                               float scale) {
  const long long nin = (long long)Hi * Wi, nout = (long long)Ho * Wo;
  const long long total = nBC * nin;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bc = e / nin;
    const int r = (int)(e - bc * nin);
    const int x = r % Wi, y = r / Wi;
    const int oy0 = up ? s * y - s / 2 : y / s - 1;
    const int ox0 = up ? s * x - s / 2 : x / s - 1;
    float wy[NC], wx[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      wy[k] = axis_w(oy0 + k, y, Hi, Ho, rs);
      wx[k] = axis_w(ox0 + k, x, Wi, Wo, rs);
    }
    const float* g = gout + bc * nout;
    float acc = 0.f;
#pragma unroll
    for (int b = 0; b < NC; ++b) {
      if (wy[b] == 0.f) continue;
      const float* row = g + (long long)(oy0 + b) * Wo;
      float rowsum = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ox = min(max(ox0 + c, 0), Wo - 1);  // weight is 0 where the index was clamped
        rowsum = fmaf(wx[c], row[ox], rowsum);
      }
      acc = fmaf(wy[b], rowsum, acc);
    }
    gin[e] = acc * scale;
  }
}

unsigned grid_for(long long total) {
  const long long want = (total + 255) / 256;
  return (unsigned)(want < (1 << 20) ? want : (1 << 20));
}

template <int NC>
void up_adjoint_separable(const float* gout, float* gin, float* ws, const IP& p, hipStream_t st, float oscale) {
  const long long nbc = p.nBC;
  float* t1 = ws;                                              // [BC, Do, Ho, Wi]
  float* t2 = ws + nbc * p.Do * p.Ho * p.Wi;                   // [BC, Do, Hi, Wi]
  long long tot = nbc * p.Do * p.Ho * p.Wi;
  hipLaunchKernelGGL(interp_axis_adjoint_kernel<NC>, dim3(grid_for(tot)), dim3(256), 0, st, gout, t1,
                     nbc * p.Do * p.Ho, p.Wo, p.Wi, 1, p.s, p.rs, 1.0f);
  tot = nbc * p.Do * p.Hi * p.Wi;
  hipLaunchKernelGGL(interp_axis_adjoint_kernel<NC>, dim3(grid_for(tot)), dim3(256), 0, st, t1, t2,
                     nbc * p.Do, p.Ho, p.Hi, p.Wi, p.s, p.rs, 1.0f);
  tot = nbc * p.Di * p.Hi * p.Wi;
  hipLaunchKernelGGL(interp_axis_adjoint_kernel<NC>, dim3(grid_for(tot)), dim3(256), 0, st, t2, gin, nbc,
                     p.Do, p.Di, p.Hi * p.Wi, p.s, p.rs, oscale);
}

// The same adjoint in ONE pass: a workgroup owns a TZ x TY x 32 tile of the coarse gradient; its fine-gradient region
// (s (T + 1) per axis) is reduced along x straight from global memory into LDS, then along y and z inside LDS.  The fine
// gradient is read once (halo re-reads hit L2) and nothing intermediate goes to HBM: the three separable launches moved
// 805 -> 402 -> 201 -> 100 MB through HBM for a [2,6,256^3] flow gradient (1.4 ms per step for both scales), this
// one reads 805 MB and writes 100.  Tap weights from the forward's own index arithmetic (axis_w), as before.
template <int S, int TZ, int TY>
__global__ __launch_bounds__(256) void up_adjoint_fused_kernel(const float* __restrict__ g, float* __restrict__ out, IP p,
                                                               float oscale) {
  constexpr int NC = 2 * S, TX = 32;
  constexpr int HZ = S * (TZ + 1), HY = S * (TY + 1);
  __shared__ float w[(TZ + TY + TX) * NC];       // tap weights of the tile's coarse indices, per axis
  __shared__ float s1[HZ * HY * TX];             // reduced along x
  __shared__ float s2[HZ * TY * TX];             // ... and y
  const int t = threadIdx.x;
  const int ntx = (p.Di > 0) ? (p.Wi + TX - 1) / TX : 1, nty = (p.Hi + TY - 1) / TY, ntz = (p.Di + TZ - 1) / TZ;
  long long tile = blockIdx.x;
  const int x0 = (int)(tile % ntx) * TX; tile /= ntx;
  const int y0 = (int)(tile % nty) * TY; tile /= nty;
  const int z0 = (int)(tile % ntz) * TZ;
  const long long bc = tile / ntz;
  // coarse extent = (Di, Hi, Wi) (this is the gradient w.r.t. the up-sampling's INPUT), fine = (Do, Ho, Wo)
  for (int i = t; i < (TZ + TY + TX) * NC; i += 256) {
    const int a = i / NC, k = i - a * NC;
    float v;
    if (a < TZ) v = axis_w(S * (z0 + a) - S / 2 + k, z0 + a, p.Di, p.Do, p.rs);
    else if (a < TZ + TY) v = axis_w(S * (y0 + a - TZ) - S / 2 + k, y0 + a - TZ, p.Hi, p.Ho, p.rs);
    else v = axis_w(S * (x0 + a - TZ - TY) - S / 2 + k, x0 + a - TZ - TY, p.Wi, p.Wo, p.rs);
    w[i] = v;
  }
  __syncthreads();
  const float* gb = g + bc * ((long long)p.Do * p.Ho * p.Wo);
  const int hz0 = S * z0 - S / 2, hy0 = S * y0 - S / 2, hx0 = S * x0 - S / 2;
  {  // along x: thread = coarse column xl (weights in registers), rows strided over the workgroup
    const int xl = t & 31;
    float wx[NC];
    int ox[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      wx[k] = w[(TZ + TY + xl) * NC + k];
      ox[k] = min(max(hx0 + S * xl + k, 0), p.Wo - 1);  // (weight 0 where the index was clamped)
    }
    for (int r = t >> 5; r < HZ * HY; r += 8) {
      const int hz = r / HY, hy = r - hz * HY;
      const int oz = min(max(hz0 + hz, 0), p.Do - 1), oy = min(max(hy0 + hy, 0), p.Ho - 1);
      const float* row = gb + ((long long)oz * p.Ho + oy) * p.Wo;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) acc = fmaf(wx[k], row[ox[k]], acc);
      s1[r * TX + xl] = acc;
    }
  }
  __syncthreads();
  for (int i = t; i < HZ * TY * TX; i += 256) {  // along y
    const int xl = i & 31, q = i >> 5;
    const int yl = q % TY, hz = q / TY;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) acc = fmaf(w[(TZ + yl) * NC + k], s1[(hz * HY + S * yl + k) * TX + xl], acc);
    s2[i] = acc;
  }
  __syncthreads();
  for (int i = t; i < TZ * TY * TX; i += 256) {  // along z
    const int xl = i & 31, q = i >> 5;
    const int yl = q % TY, zl = q / TY;
    const int z = z0 + zl, y = y0 + yl, x = x0 + xl;
    if (z >= p.Di || y >= p.Hi || x >= p.Wi) continue;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) acc = fmaf(w[zl * NC + k], s2[((S * zl + k) * TY + yl) * TX + xl], acc);
    out[((bc * p.Di + z) * p.Hi + y) * (long long)p.Wi + x] = acc * oscale;
  }
}

template <int S, int TZ, int TY>
void up_adjoint_fused(const float* gout, float* gin, const IP& p, hipStream_t st, float oscale) {
  const long long tiles = (long long)((p.Wi + 31) / 32) * ((p.Hi + TY - 1) / TY) * ((p.Di + TZ - 1) / TZ) * p.nBC;
  hipLaunchKernelGGL((up_adjoint_fused_kernel<S, TZ, TY>), dim3((unsigned)tiles), dim3(256), 0, st, gout, gin, p, oscale);
}

}  // namespace

// grad_in = scale * adjoint(grad_out) for the up-sampling direction with a workspace (the separable form,
// `scale` applied by its last pass); every other form requires scale == 1.
extern "C" int fs_interp3d_bwd_scaled(const float* grad_out, float* grad_in, float* ws, int B, int C, int Din,
                                      int Hin, int Win, int Dout, int Hout, int Wout, int factor, int upsample,
                                      float scale, fs_stream_t stream) {
  FS_ENTER();
  // `scale`: the separable up-sampling adjoint, or the exact float4 down-sampling adjoint (the flow's `* 1/scale` of
  // IFBlock.forward: a power of two, so scaling the gradient inside the kernel is bitwise the separate pass)
  const bool down_v4 = !upsample && Din == Dout * factor && Hin == Hout * factor && Win == Wout * factor &&
                       (Win & 3) == 0 && ((uintptr_t)grad_in & 15) == 0;
  if (scale != 1.0f && !(upsample && ws != nullptr) && !down_v4) return FS_ERR_ARG;
  FS_REQUIRE_PTR(grad_out); FS_REQUIRE_PTR(grad_in);
  if (B < 1 || C < 1 || Din < 1 || Hin < 1 || Win < 1 || Dout < 1 || Hout < 1 || Wout < 1)
    return FS_ERR_SHAPE;
  if ((long long)Din * Hin * Win >= (1ll << 31) || (long long)Dout * Hout * Wout >= (1ll << 31))
    return FS_ERR_SHAPE;
  if (factor != 2 && factor != 4) return FS_ERR_ARG;
  IP p;
  p.C = 1; p.nsrc = 0;
  p.Di = Din; p.Hi = Hin; p.Wi = Win; p.Do = Dout; p.Ho = Hout; p.Wo = Wout;
  p.up = upsample ? 1 : 0;
  p.s = factor;
  p.rs = upsample ? 1.0f / (float)factor : (float)factor;  // 1 / scale_factor (exact for 2, 4)
  p.nBC = (long long)B * C;
  // F.interpolate's output size for these scale factors: floor(in * scale_factor)
  if (upsample) {
    if (Dout != Din * factor || Hout != Hin * factor || Wout != Win * factor) return FS_ERR_SHAPE;
  } else {
    if (Dout != Din / factor || Hout != Hin / factor || Wout != Win / factor) return FS_ERR_SHAPE;
  }
  const long long total = p.nBC * Din * Hin * Win;
  hipStream_t st = (hipStream_t)stream;
  if (!upsample) {
    const bool exact = Din == Dout * factor && Hin == Hout * factor && Win == Wout * factor;
    if (exact && (Win & 3) == 0 && ((uintptr_t)grad_in & 15) == 0 && total / 4 + (1ll << 28) < (1ll << 32))
      hipLaunchKernelGGL(interp3d_down_adjoint_exact_v4<unsigned>, dim3(grid_for(total / 4)), dim3(256), 0, st, grad_out,
                         (float4*)grad_in, p, scale);
    else if (exact && (Win & 3) == 0 && ((uintptr_t)grad_in & 15) == 0)
      hipLaunchKernelGGL(interp3d_down_adjoint_exact_v4<long long>, dim3(grid_for(total / 4)), dim3(256), 0, st, grad_out,
                         (float4*)grad_in, p, scale);
    else if (exact)
      hipLaunchKernelGGL(interp3d_down_adjoint_exact, dim3(grid_for(total)), dim3(256), 0, st, grad_out,
                         grad_in, p);
    else
      hipLaunchKernelGGL(interp3d_adjoint_kernel<3>, dim3(grid_for(total)), dim3(256), 0, st, grad_out,
                         grad_in, p);
  } else if (ws != nullptr) {
    // x2: one fused pass; FLOWSCI_INTERP_SEPARABLE=1: the three separable launches it replaced (A/B)
    static const bool separable = FS_AB_ENV("FLOWSCI_INTERP_SEPARABLE");
    const long long tiles2 = (long long)((p.Wi + 31) / 32) * ((p.Hi + 7) / 8) * ((p.Di + 3) / 4) * p.nBC;
    // (x2: 0.83 -> 0.38 ms for a [2,6,256^3] gradient; x4: the fused form's stride-4 row gathers make it slower than
    // the separable passes, 0.64 vs 0.45 ms, so that scale keeps them)
    if (!separable && factor == 2 && tiles2 < (1ll << 31)) up_adjoint_fused<2, 4, 8>(grad_out, grad_in, p, st, scale);
    else if (factor == 2) up_adjoint_separable<4>(grad_out, grad_in, ws, p, st, scale);
    else up_adjoint_separable<8>(grad_out, grad_in, ws, p, st, scale);
  } else if (factor == 2) {
    hipLaunchKernelGGL(interp3d_adjoint_kernel<4>, dim3(grid_for(total)), dim3(256), 0, st, grad_out,
                       grad_in, p);
  } else {
    hipLaunchKernelGGL(interp3d_adjoint_kernel<8>, dim3(grid_for(total)), dim3(256), 0, st, grad_out,
                       grad_in, p);
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_interp3d_bwd(const float* grad_out, float* grad_in, float* ws, int B, int C, int Din,
                               int Hin, int Win, int Dout, int Hout, int Wout, int factor, int upsample,
                               fs_stream_t stream) {
  return fs_interp3d_bwd_scaled(grad_out, grad_in, ws, B, C, Din, Hin, Win, Dout, Hout, Wout, factor, upsample,
                                1.0f, stream);
}

extern "C" int fs_upsample3d_scale_add(const float* small, const float* prev, float* out, int B, int C, int Din,
                                       int Hin, int Win, int factor, float scale, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(small); FS_REQUIRE_PTR(out);
  if (B < 1 || C < 1 || Din < 1 || Hin < 1 || Win < 1) return FS_ERR_SHAPE;
  if (factor != 2 && factor != 4) return FS_ERR_ARG;
  if ((long long)Din * Hin * Win * factor * factor * factor >= (1ll << 31)) return FS_ERR_SHAPE;
  IP p;
  p.C = 1; p.nsrc = 0;
  p.Di = Din; p.Hi = Hin; p.Wi = Win;
  p.Do = Din * factor; p.Ho = Hin * factor; p.Wo = Win * factor;
  p.up = 1; p.s = factor; p.rs = 1.0f / (float)factor;
  p.nBC = (long long)B * C;
  const long long total = p.nBC * p.Do * p.Ho * p.Wo;
  const bool al16 = (p.Wo & 3) == 0 && (((uintptr_t)out | (uintptr_t)prev) & 15) == 0;
  static const bool no_tile = FS_AB_ENV("FLOWSCI_UP_NO_TILE");
  const long long tiles = total / (8 * 8 * 64);
  if (al16 && !no_tile && p.Do % 8 == 0 && p.Ho % 8 == 0 && p.Wo % 64 == 0 && tiles < (1ll << 31) && (factor == 2 || factor == 4)) {
    if (factor == 2)
      hipLaunchKernelGGL(upsample3d_scale_add_tile_kernel<2>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, small,
                         (const float4*)prev, (float4*)out, p, scale);
    else
      hipLaunchKernelGGL(upsample3d_scale_add_tile_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, small,
                         (const float4*)prev, (float4*)out, p, scale);
  } else if (al16)
    hipLaunchKernelGGL(upsample3d_scale_add_v4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream,
                       small, (const float4*)prev, (float4*)out, p, scale);
  else
    hipLaunchKernelGGL(upsample3d_scale_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, small,
                       prev, out, p, scale);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// out = scale * F.interpolate(in, scale_factor=1/factor, mode="trilinear", align_corners=False): the generic
// ATen-order kernel with the source-index scale `factor`; output extent floor(in / factor).
static int downsample3d_impl(const float* in, const float* const* srcv, const long long* sbsv, float* out, int B, int C,
                             int Din, int Hin, int Win, int factor, float scale, fs_stream_t stream) {
  FS_REQUIRE_PTR(out);
  if (srcv == nullptr) FS_REQUIRE_PTR(in);
  if (B < 1 || C < 1 || Din < 1 || Hin < 1 || Win < 1) return FS_ERR_SHAPE;
  if (factor != 2 && factor != 4) return FS_ERR_ARG;
  if (Din / factor < 1 || Hin / factor < 1 || Win / factor < 1) return FS_ERR_SHAPE;
  if ((long long)Din * Hin * Win >= (1ll << 31)) return FS_ERR_SHAPE;
  IP p;
  p.C = 1; p.nsrc = 0;
  p.Di = Din; p.Hi = Hin; p.Wi = Win;
  p.Do = Din / factor; p.Ho = Hin / factor; p.Wo = Win / factor;
  p.up = 0; p.s = factor; p.rs = (float)factor;
  p.nBC = (long long)B * C;
  p.C = C; p.nsrc = 0;
  if (srcv != nullptr) {
    if (C > 12) return FS_ERR_ARG;
    p.nsrc = C;
    for (int c = 0; c < C; ++c) {
      if (srcv[c] == nullptr) return FS_ERR_NULLPTR;
      if (sbsv[c] < (long long)Din * Hin * Win) return FS_ERR_ARG;
      p.src[c] = srcv[c]; p.sbs[c] = sbsv[c];
    }
  }
  const long long total = p.nBC * p.Do * p.Ho * p.Wo;
  // 16-byte row pieces: rows start 16-byte aligned (Wi % 4 == 0, aligned planes); the / 2 and / 4 taps of four outputs
  // are then two float4 / four 4-byte-aligned pairs per source row
  bool al = (Win & 3) == 0 && ((long long)Hin * Win) % 4 == 0;
  if (srcv == nullptr) al = al && ((uintptr_t)in & 15) == 0;
  else for (int c = 0; c < C; ++c) al = al && ((uintptr_t)srcv[c] & 15) == 0 && sbsv[c] % 4 == 0;
  if ((p.Wo & 3) == 0 && ((uintptr_t)out & 15) == 0 && al) {
    if (factor == 2)
      hipLaunchKernelGGL(downsample3d_v4_kernel<2>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, in,
                         (float4*)out, p, scale);
    else
      hipLaunchKernelGGL(downsample3d_v4_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, in,
                         (float4*)out, p, scale);
  } else if ((p.Wo & 3) == 0 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL(upsample3d_scale_add_v4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream,
                       in, (const float4*)nullptr, (float4*)out, p, scale);
  else
    hipLaunchKernelGGL(upsample3d_scale_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                       (const float*)nullptr, out, p, scale);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_downsample3d_fwd(const float* in, float* out, int B, int C, int Din, int Hin, int Win,
                                   int factor, float scale, fs_stream_t stream) {
  FS_ENTER();
  return downsample3d_impl(in, nullptr, nullptr, out, B, C, Din, Hin, Win, factor, scale, stream);
}

// fs_downsample3d_fwd over an input that is never concatenated: channel c is the plane src[c] (sample 0) of a tensor
// with batch stride batch_strides[c] floats (host arrays of C <= 12 entries, read at launch).
extern "C" int fs_downsample3d_fwd_ms(const float* const* src, const long long* batch_strides, float* out, int B, int C,
                                      int Din, int Hin, int Win, int factor, float scale, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(src); FS_REQUIRE_PTR(batch_strides);
  return downsample3d_impl(nullptr, src, batch_strides, out, B, C, Din, Hin, Win, factor, scale, stream);
}

static int resize2d_dims(int Hin, int Win, int Hout, int Wout, int factor, int upsample) {
  if (factor != 2 && factor != 4) return FS_ERR_ARG;
  if (upsample) return (Hout == Hin * factor && Wout == Win * factor) ? FS_OK : FS_ERR_SHAPE;
  return (Hout == Hin / factor && Wout == Win / factor && Hout >= 1 && Wout >= 1) ? FS_OK : FS_ERR_SHAPE;
}

// out = scale * F.interpolate(in, scale_factor = factor or 1/factor, mode="bilinear", align_corners=False)
extern "C" int fs_resize2d_fwd(const float* in, float* out, int B, int C, int Hin, int Win, int Hout, int Wout,
                               int factor, int upsample, float scale, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(in); FS_REQUIRE_PTR(out);
  if (B < 1 || C < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return FS_ERR_SHAPE;
  const int rc = resize2d_dims(Hin, Win, Hout, Wout, factor, upsample);
  if (rc != FS_OK) return rc;
  if ((long long)Hin * Win >= (1ll << 31) || (long long)Hout * Wout >= (1ll << 31)) return FS_ERR_SHAPE;
  const float rs = upsample ? 1.0f / (float)factor : (float)factor;
  const long long total = (long long)B * C * Hout * Wout;
  hipLaunchKernelGGL(resize2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out,
                     (long long)B * C, Hin, Win, Hout, Wout, rs, scale);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// grad_in = scale * adjoint(grad_out) of the resize above (gather, no atomics)
extern "C" int fs_resize2d_bwd(const float* grad_out, float* grad_in, int B, int C, int Hin, int Win, int Hout,
                               int Wout, int factor, int upsample, float scale, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(grad_out); FS_REQUIRE_PTR(grad_in);
  if (B < 1 || C < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return FS_ERR_SHAPE;
  const int rc = resize2d_dims(Hin, Win, Hout, Wout, factor, upsample);
  if (rc != FS_OK) return rc;
  if ((long long)Hin * Win >= (1ll << 31) || (long long)Hout * Wout >= (1ll << 31)) return FS_ERR_SHAPE;
  const float rs = upsample ? 1.0f / (float)factor : (float)factor;
  const long long nbc = (long long)B * C, total = nbc * Hin * Win;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid_for(total)), b(256);
  if (!upsample)
    hipLaunchKernelGGL(resize2d_adjoint_kernel<3>, g, b, 0, st, grad_out, grad_in, nbc, Hin, Win, Hout, Wout, rs, 0,
                       factor, scale);
  else if (factor == 2)
    hipLaunchKernelGGL(resize2d_adjoint_kernel<4>, g, b, 0, st, grad_out, grad_in, nbc, Hin, Win, Hout, Wout, rs, 1,
                       factor, scale);
  else
    hipLaunchKernelGGL(resize2d_adjoint_kernel<8>, g, b, 0, st, grad_out, grad_in, nbc, Hin, Win, Hout, Wout, rs, 1,
                       factor, scale);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
