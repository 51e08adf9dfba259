// convtr.hip -- ConvTranspose3d(k=4, s=2, p=1) forward of the IFNet-3D heads, which is also the input
// gradient of the Conv3d(k=4, s=2, p=1) layers (IFBlock.conv0), for gfx950.
//
// Companion of convfwd.hip / convwrw.hip (not a §8(a) row).  MIOpen runs every one of these through
// GEMM + Col2Im3dU, one sample at a time: 64 ms of the 231 ms 256^3 step.
//
//   y[b, co, z,y,x] = bias[co] + sum_{ci} sum_{k : (z + 1 - kz) even, ...} x[b, ci, (z+1-kz)/2, ...] * W[ci, co, kz,ky,kx]
//
// Sub-pixel form: an output voxel of parity (pz,py,px) at z = 2q + pz sees exactly two taps per axis,
//   parity 0: (input q, k = 1), (input q-1, k = 3)        parity 1: (input q+1, k = 0), (input q, k = 2)
// so the layer is 8 independent 2x2x2 convolutions over the INPUT grid sharing one 3x3x3 input
// neighbourhood.  One thread / one MFMA column owns one input-grid position q and produces all 8
// output voxels 2q + {0,1}^3: stores are float2 (both x parities) and fully coalesced.
//
//   convtr_mfma_kernel (12 < Cout <= 32): implicit GEMM on v_mfma_f32_32x32x2_f32, M = 32 output
//       channels, N = 32 consecutive qx, 8 accumulator tiles (one per parity class) per wave; per chunk
//       of 4 input channels the 3x3x3-haloed input brick and the [4][64][32] weight slab sit in LDS.
//   convtr_mfma16_kernel (7..16 output channels: the gradient w.r.t. the 11/12-channel block input):
//       the same on v_mfma_f32_16x16x4_f32 (16 channels x 16 positions, the 4 chunk channels per MFMA).
//   convtr_valu_kernel<CO> (Cout <= 6: the flow / mask heads): padding 6 channels to a 16/32-row MFMA
//       tile wastes most of the matrix core, and the fp32 vector ALUs have the same peak as the fp32
//       matrix cores on this part.  One thread per two q (consecutive qy), 16*CO accumulators, weights
//       are wave-uniform -> scalar loads feeding v_fmac's SGPR operand (each feeds two FMAs).
#include <cstdint>
#include <cstdlib>

#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TP {
  int B, Cin, Cout;
  int CoutT;             // channels of the whole output tensor (> Cout when a launch covers a 32-channel slice of it)
  int Di, Hi, Wi;        // input extent
  int Dout, Hout, Wout;  // output extent (2*in or 2*in + 1 per axis)
  int Dq, Hq, Wq;        // input-grid positions that own outputs: ceil(out / 2)
  int tz, ty, tx;
  long long tiles;
  // optional fused PReLU: Z = prelu(Y) written next to Y
  const float* slope;
  float* Z;
  int nslope;
  // optional residual: Y = conv + bias + addend (same shape as Y) -- IFNet's `flow = flow + flow_d`
  const float* addend;
  const float* Ybase;  // = Y (to locate the addend plane of a channel)
  // 32-channel slices of a wider output in ONE launch of the 32-row kernels (blockIdx.y = slice): floats between the
  // slices' re-laid-out weights; the output-side pointers move by 32 channels per slice
  long long wslice;
#ifdef FS_ABLATION
  int ab = 0;  // measurement switches (FLOWSCI_TR_AB: 4 = the loader-wave kernels skip their epilogue, 8 / 16 = their loaders re-use
               // the input brick / weight slab of the first two chunks -- wrong results by design)
#endif
};

// blockIdx.y-th 32-channel slice of a launch that covers several (block0's 128 -> 64 deconvolution: two half-empty
// launches one after the other become one that fills the chip)
__device__ __forceinline__ void tr_slice(TP& p, const float*& Wt, const float*& bias, float*& Y) {
  const int sl = blockIdx.y;
  if (sl == 0) return;
  const size_t off = (size_t)sl * 32 * ((size_t)p.Dout * p.Hout * p.Wout);
  Wt += (size_t)sl * p.wslice;
  if (bias) bias += 32 * sl;
  Y += off;
  if (p.Z) p.Z += off;
  if (p.slope && p.nslope != 1) p.slope += 32 * sl;
  if (p.addend) p.addend += off;
  p.Ybase += off;
}

// tap a (0/1) of output parity p along one axis: input offset d and kernel index k
__device__ __forceinline__ constexpr int tap_d(int p, int a) { return p == 0 ? (a == 0 ? 0 : -1) : (a == 0 ? 1 : 0); }
__device__ __forceinline__ constexpr int tap_k(int p, int a) { return p == 0 ? (a == 0 ? 1 : 3) : (a == 0 ? 0 : 2); }

// the weight re-layouts (FS_WPREP_TR32 / TR16 / P8) live in wprep.hpp
#include "wprep.hpp"

// all 8 outputs of position q for channel co; float2 stores when the rows are 8-byte aligned.
// zc / sl: the same channel's plane of the fused PReLU output and its slope (zc may be null).
__device__ __forceinline__ void store8(float* __restrict__ yc, const float (&v)[8], int qz, int qy, int qx,
                                       const TP& p, float* __restrict__ zc = nullptr, float sl = 0.f) {
  const float* __restrict__ ac = p.addend ? p.addend + (yc - p.Ybase) : nullptr;
#pragma unroll
  for (int pz = 0; pz < 2; ++pz)
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      const int z = 2 * qz + pz, y = 2 * qy + py, x = 2 * qx;
      if (z >= p.Dout || y >= p.Hout || x >= p.Wout) continue;
      float* row = yc + ((size_t)z * p.Hout + y) * p.Wout + x;
      float v0 = v[(pz * 2 + py) * 2], v1 = v[(pz * 2 + py) * 2 + 1];
      if (ac != nullptr) {
        const float* arow = ac + (row - yc);
        v0 += arow[0];
        if (x + 1 < p.Wout) v1 += arow[1];
      }
      if ((p.Wout & 1) == 0) {
        *reinterpret_cast<float2*>(row) = make_float2(v0, v1);
      } else {
        row[0] = v0;
        if (x + 1 < p.Wout) row[1] = v1;
      }
      if (zc != nullptr) {
        float* zrow = zc + (row - yc);
        const float z0 = v0 > 0.f ? v0 : sl * v0, z1 = v1 > 0.f ? v1 : sl * v1;
        if ((p.Wout & 1) == 0) {
          *reinterpret_cast<float2*>(zrow) = make_float2(z0, z1);
        } else {
          zrow[0] = z0;
          if (x + 1 < p.Wout) zrow[1] = z1;
        }
      }
    }
}

// store8 for the loader-wave kernels, whose epilogue is not hidden behind a second workgroup and is store-ISSUE-bound
// (~75 cycles per store wave-instruction and CU whatever its width, csrc/convtr.hip::convtr_p8_kernel): neighbouring
// lanes (positions q, q + 1) swap one (x = 2q, 2q + 1) pair per z parity -- DPP quad_perm [1,0,3,2] -- so that an
// even lane holds the four consecutive outputs 2q .. 2q + 3 of the y-parity-0 row and its odd neighbour 2q - 2 ..
// 2q + 1 of the y-parity-1 row: 16-byte stores, half as many.  EVERY lane of the wave must call it (`ok`: this
// lane's channel and rows exist; columns are tested per element).
__device__ __forceinline__ void store8_quad(float* __restrict__ yc, const float (&v)[8], int qz, int qy, int qx, bool ok,
                                            const TP& p, float* __restrict__ zc = nullptr, float sl = 0.f) {
  const float* __restrict__ ac = p.addend ? p.addend + (yc - p.Ybase) : nullptr;
  const bool evn = (qx & 1) == 0;
#pragma unroll
  for (int pz = 0; pz < 2; ++pz) {
    const float a0 = v[(pz * 2 + 0) * 2], a1 = v[(pz * 2 + 0) * 2 + 1];  // row y = 2 qy
    const float b0 = v[(pz * 2 + 1) * 2], b1 = v[(pz * 2 + 1) * 2 + 1];  // row y = 2 qy + 1
    const float k0 = evn ? a0 : b0, k1 = evn ? a1 : b1;
    const float r0 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(evn ? b0 : a0), 0xB1, 0xF, 0xF, false));
    const float r1 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(evn ? b1 : a1), 0xB1, 0xF, 0xF, false));
    float4 o = evn ? make_float4(k0, k1, r0, r1) : make_float4(r0, r1, k0, k1);
    const int z = 2 * qz + pz, y = 2 * qy + (evn ? 0 : 1), x0 = 2 * (evn ? qx : qx - 1);
    if (!ok || z >= p.Dout || y >= p.Hout || x0 >= p.Wout) continue;
    float* row = yc + ((size_t)z * p.Hout + y) * p.Wout + x0;
    if (x0 + 3 < p.Wout) {
      if (ac != nullptr) {
        const float4 a4 = *reinterpret_cast<const float4*>(ac + (row - yc));
        o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
      }
      *reinterpret_cast<float4*>(row) = o;
      if (zc != nullptr)
        *reinterpret_cast<float4*>(zc + (row - yc)) = make_float4(o.x > 0.f ? o.x : sl * o.x, o.y > 0.f ? o.y : sl * o.y,
                                                                  o.z > 0.f ? o.z : sl * o.z, o.w > 0.f ? o.w : sl * o.w);
    } else {
      const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (x0 + e < p.Wout) {
          const float w = ov[e] + (ac != nullptr ? ac[(row - yc) + e] : 0.f);
          row[e] = w;
          if (zc != nullptr) zc[(row - yc) + e] = w > 0.f ? w : sl * w;
        }
    }
  }
}

template <int TZ, int TY>
__global__ __launch_bounds__(256, 2) void convtr_mfma_kernel(const float* __restrict__ X,
                                                          const float* __restrict__ Wt_,
                                                          const float* __restrict__ bias_,
                                                          float* __restrict__ Y_, TP p) {
  const float* Wt = Wt_;
  const float* bias = bias_;
  float* Y = Y_;
  tr_slice(p, Wt, bias, Y);
  static_assert(TZ * TY == 4, "one 32-position row per wave");
  constexpr int CI = 4;
  constexpr int ZT = TZ + 2, YT = TY + 2, XT = 34;
  constexpr int PS = YT * XT, CHS = ZT * PS;
  constexpr int NX = CI * CHS, NW = CI * 64 * 32;
  __shared__ float sX[NX];
  __shared__ __attribute__((aligned(16))) float sW[NW];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int col = lane & 31, kh = lane >> 5;
  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;  // contiguous brick range per XCD (see convfwd.hip)
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int qz0 = tzi * TZ, qy0 = tyi * TY, qx0 = txi * 32;
  const int wz = wv / TY, wy = wv % TY;

  const float* bB = sX + kh * 2 * CHS + (wz + 1) * PS + (wy + 1) * XT + (col + 1);
  const float* aB = sW + kh * 2 * 64 * 32 + col;

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  constexpr int ITX = (NX + 255) / 256, ITW = NW / 4 / 256;
  // chunk-invariant byte offsets of the brick elements this thread stages (~0u = outside -> zero);
  // the next chunk is fetched into registers before the MFMA phase and lands under it (convfwd.hip)
  unsigned xoff[ITX];
#pragma unroll
  for (int it = 0; it < ITX; ++it) {
    const int i = t + 256 * it;
    const int c = i / CHS, r1 = i - c * CHS;
    const int z = r1 / PS, r2 = r1 - z * PS;
    const int y = r2 / XT, x = r2 - y * XT;
    const int gz = qz0 - 1 + z, gy = qy0 - 1 + y, gx = qx0 - 1 + x;
    const bool ok = i < NX && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
    xoff[it] = ok ? ((unsigned)c * (unsigned)xvol + ((unsigned)gz * p.Hi + gy) * p.Wi + gx) * 4u : ~0u;
  }
  float rX[ITX];
  float4 rW[ITW];
  auto fetch = [&](int c0) {
    const char* xb = reinterpret_cast<const char*>(X + ((size_t)b * p.Cin + c0) * xvol);
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
      const bool chan_ok = (c0 + CI <= p.Cin) || (c0 + (t + 256 * it) / CHS < p.Cin);
      rX[it] = (xoff[it] != ~0u && chan_ok) ? *reinterpret_cast<const float*>(xb + xoff[it]) : 0.f;
    }
    const float4* wb = reinterpret_cast<const float4*>(Wt + (size_t)c0 * 64 * 32);
#pragma unroll
    for (int it = 0; it < ITW; ++it) rW[it] = wb[t + 256 * it];
  };
  fetch(0);
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
      const int i = t + 256 * it;
      if (i < NX) sX[i] = rX[it];
    }
#pragma unroll
    for (int it = 0; it < ITW; ++it) reinterpret_cast<float4*>(sW)[t + 256 * it] = rW[it];
    __syncthreads();
    if (c0 + CI < p.Cin) fetch(c0 + CI);

#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      float xn[27];  // the 3x3x3 input neighbourhood of this lane's position (channel cl / cl+2 by kh)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            xn[(dz * 3 + dy) * 3 + dx] = bB[cl * CHS + (dz - 1) * PS + (dy - 1) * XT + (dx - 1)];
#pragma unroll
      for (int cls = 0; cls < 8; ++cls) {
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) {
          const int a = tp >> 2, bb = (tp >> 1) & 1, c = tp & 1;
          const int kidx = (tap_k(pz, a) * 4 + tap_k(py, bb)) * 4 + tap_k(px, c);
          const int didx = ((tap_d(pz, a) + 1) * 3 + (tap_d(py, bb) + 1)) * 3 + (tap_d(px, c) + 1);
          const float av = aB[(cl * 64 + kidx) * 32];
          acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xn[didx], acc[cls], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  const int qz = qz0 + wz, qy = qy0 + wy, qx = qx0 + col;
  if (qz < p.Dq && qy < p.Hq) {  // wave-uniform: every lane takes part in the lane exchange of store8_quad
    const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const bool ok = co < p.Cout;
      const int cc = ok ? co : 0;
      const float bv = (bias && ok) ? bias[cc] : 0.f;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = acc[c][r] + bv;
      store8_quad(Y + ((size_t)b * p.CoutT + cc) * yvol, v, qz, qy, qx, ok, p,
                  p.Z ? p.Z + ((size_t)b * p.CoutT + cc) * yvol : nullptr, p.Z ? p.slope[p.nslope == 1 ? 0 : cc] : 0.f);
    }
  }
}

// ---- 7..16 output channels: v_mfma_f32_16x16x4_f32 ------------------------------------------------
// M = 16 output channels, N = 16 consecutive qx, and the 4 reduction elements of one MFMA are the
// four input channels of the chunk at one tap: a (class, tap) pair is ONE instruction per chunk and
// column tile, an accumulator tile is 4 VGPRs.  A wave owns one brick row = two 16-position tiles.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TZ, int TY>
__global__ __launch_bounds__(256, 2) void convtr_mfma16_kernel(const float* __restrict__ X,
                                                            const float* __restrict__ Wt,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ Y, TP p) {
  static_assert(TZ * TY == 4, "one 32-position row per wave");
  constexpr int CI = 4;
  constexpr int ZT = TZ + 2, YT = TY + 2, XT = 34;
  constexpr int PS = YT * XT, CHS = ZT * PS;
  constexpr int NX = CI * CHS, NW = CI * 64 * 16;
  __shared__ float sX[NX];
  __shared__ __attribute__((aligned(16))) float sW[NW];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int col = lane & 15, kq = lane >> 4;
  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int qz0 = tzi * TZ, qy0 = tyi * TY, qx0 = txi * 32;
  const int wz = wv / TY, wy = wv % TY;

  const float* bB = sX + kq * CHS + (wz + 1) * PS + (wy + 1) * XT + (col + 1);
  const float* aB = sW + kq * 64 * 16 + col;

  f32x4 acc[2][8];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[n][c][r] = 0.f;

  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  constexpr int ITX = (NX + 255) / 256, ITW = NW / 4 / 256;
  unsigned xoff[ITX];
#pragma unroll
  for (int it = 0; it < ITX; ++it) {
    const int i = t + 256 * it;
    const int c = i / CHS, r1 = i - c * CHS;
    const int z = r1 / PS, r2 = r1 - z * PS;
    const int y = r2 / XT, x = r2 - y * XT;
    const int gz = qz0 - 1 + z, gy = qy0 - 1 + y, gx = qx0 - 1 + x;
    const bool ok = i < NX && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
    xoff[it] = ok ? ((unsigned)c * (unsigned)xvol + ((unsigned)gz * p.Hi + gy) * p.Wi + gx) * 4u : ~0u;
  }
  float rX[ITX];
  float4 rW[ITW];
  auto fetch = [&](int c0) {
    const char* xb = reinterpret_cast<const char*>(X + ((size_t)b * p.Cin + c0) * xvol);
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
      const bool chan_ok = (c0 + CI <= p.Cin) || (c0 + (t + 256 * it) / CHS < p.Cin);
      rX[it] = (xoff[it] != ~0u && chan_ok) ? *reinterpret_cast<const float*>(xb + xoff[it]) : 0.f;
    }
    const float4* wb = reinterpret_cast<const float4*>(Wt + (size_t)c0 * 64 * 16);
#pragma unroll
    for (int it = 0; it < ITW; ++it) rW[it] = wb[t + 256 * it];
  };
  fetch(0);
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
      const int i = t + 256 * it;
      if (i < NX) sX[i] = rX[it];
    }
#pragma unroll
    for (int it = 0; it < ITW; ++it) reinterpret_cast<float4*>(sW)[t + 256 * it] = rW[it];
    __syncthreads();
    if (c0 + CI < p.Cin) fetch(c0 + CI);

    float xn[2][27];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            xn[n][(dz * 3 + dy) * 3 + dx] = bB[16 * n + (dz - 1) * PS + (dy - 1) * XT + (dx - 1)];
#pragma unroll
    for (int cls = 0; cls < 8; ++cls) {
      const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
        const int a = tp >> 2, bb = (tp >> 1) & 1, c = tp & 1;
        const int kidx = (tap_k(pz, a) * 4 + tap_k(py, bb)) * 4 + tap_k(px, c);
        const int didx = ((tap_d(pz, a) + 1) * 3 + (tap_d(py, bb) + 1)) * 3 + (tap_d(px, c) + 1);
        const float av = aB[kidx * 16];
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[n][cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xn[n][didx], acc[n][cls], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // D row (channel) = 4 * (lane >> 4) + r, column (position) = lane & 15
  const int qz = qz0 + wz, qy = qy0 + wy;
  if (qz < p.Dq && qy < p.Hq) {
    const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int qx = qx0 + 16 * n + col;
      if (qx >= p.Wq) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * kq + r;
        if (co >= p.Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = acc[n][c][r] + bv;
        store8(Y + ((size_t)b * p.Cout + co) * yvol, v, qz, qy, qx, p,
               p.Z ? p.Z + ((size_t)b * p.Cout + co) * yvol : nullptr, p.Z ? p.slope[p.nslope == 1 ? 0 : co] : 0.f);
      }
    }
  }
}

// ---- loader-wave forms (one 8-wave workgroup per CU, two LDS buffers) --------------------------------
// The two MFMA kernels above with the staging moved to partner waves and to `buffer_load_dwordx4 ... lds`
// (csrc/convwrw.hip has the measurements that led here): waves 4-7 issue the 16-byte pieces of chunk c+1 --
// the haloed input brick (rows start 4 floats left of the first position: 16-byte aligned when Wi % 4 == 0,
// pitch 40) and the weight slab -- into the second LDS buffer while waves 0-3 run the MFMA phase of chunk c.
// The MFMA phase of a chunk is short here (128 matrix instructions per wave) and the weight slab is 3/4 of
// the staged bytes: in the kernels above every thread spent ~33 load / ds_write instructions per chunk on it.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
[[maybe_unused]] constexpr unsigned DMA_OOB = 0x80000000u;

// the loader waves' whole life: CPW = channels per weight-slab row (32 / 16)
template <int TZ, int TY, int CPW>
__device__ __forceinline__ void tr_loader(const float* __restrict__ X, const float* __restrict__ Wt, const TP& p,
                                          float* lds, int wv, int lane, int b, int qz0, int qy0, int qx0) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass has neither the buffer-resource type nor the LDS-DMA builtin)
  __builtin_amdgcn_s_setprio(3);  // (a loader wave issues a handful of instructions per period: they should not queue behind the matrix wave's)
  constexpr int CI = 4;
  constexpr int ZT = TZ + 2, YT = TY + 2, XP = 40;
  constexpr int PS = YT * XP, CHS = ZT * PS;
  constexpr int NX = CI * CHS, NW = CI * 64 * CPW;
  constexpr int NXL = (NX + 255) / 256 * 256, NWL = (NW + 255) / 256 * 256;
  constexpr int NXW = (NXL / 256 + 3) / 4, NWW = (NWL / 256 + 3) / 4;
  constexpr int BUF = NXL + NWL;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  unsigned xoff[NXW], woff[NWW];
#pragma unroll
  for (int k = 0; k < NXW; ++k) {
    const int i = 256 * (wv + 4 * k) + 4 * lane;
    const int c = i / CHS, r1 = i - c * CHS;
    const int z = r1 / PS, r2 = r1 - z * PS;
    const int y = r2 / XP, x = r2 - y * XP;
    const int gz = qz0 - 1 + z, gy = qy0 - 1 + y, gx = qx0 - 4 + x;
    const bool ok = i < NX && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
    xoff[k] = ok ? ((unsigned)c * (unsigned)xvol + ((unsigned)gz * p.Hi + gy) * p.Wi + gx) * 4u : DMA_OOB;
  }
#pragma unroll
  for (int k = 0; k < NWW; ++k) {
    const int i = 256 * (wv + 4 * k) + 4 * lane;
    woff[k] = i < NW ? (unsigned)i * 4u : DMA_OOB;
  }
  auto stage = [&](int c0, int buf) {
    const int nch = (p.Cin - c0 < CI) ? (p.Cin - c0) : CI;  // channels past Cin read as zero
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(X + ((size_t)b * p.Cin + c0) * xvol), (short)0, (int)((unsigned)nch * (unsigned)xvol * 4u), 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Wt + (size_t)c0 * 64 * CPW), (short)0,
                                                                   0x7fffffff, 0x00020000);
    float* base = lds + buf * BUF;
#ifdef FS_ABLATION  // measurement: chunks past the second re-use the staged bytes (8: the input brick, 16: the weight slab)
    const bool skip_x = (p.ab & 8) && c0 >= 2 * CI, skip_w = (p.ab & 16) && c0 >= 2 * CI;
#else
    constexpr bool skip_x = false, skip_w = false;
#endif
#pragma unroll
    for (int k = 0; k < NXW; ++k)
      if (256 * (wv + 4 * k) < NXL && !skip_x)  // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + 256 * (wv + 4 * k)), 16, xoff[k], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NWW; ++k)
      if (256 * (wv + 4 * k) < NWL && !skip_w)  // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + NXL + 256 * (wv + 4 * k)), 16, woff[k], 0, 0, 0);
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
    if (c0 + CI < p.Cin) stage(c0 + CI, buf ^ 1);
    // the pieces of the next chunk have landed; past the barrier the matrix waves are done reading `buf`
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }
#endif
}

#include "convtr_s3.hpp"

template <int TZ, int TY>
__global__ __launch_bounds__(512, 2) void convtr_mfma_ws_kernel(const float* __restrict__ X,
                                                             const float* __restrict__ Wt_,
                                                             const float* __restrict__ bias_,
                                                             float* __restrict__ Y_, TP p) {
  const float* Wt = Wt_;
  const float* bias = bias_;
  float* Y = Y_;
  tr_slice(p, Wt, bias, Y);
  static_assert(TZ * TY == 4, "one 32-position row per matrix wave");
  constexpr int CI = 4;
  constexpr int ZT = TZ + 2, YT = TY + 2, XP = 40;
  constexpr int PS = YT * XP, CHS = ZT * PS;
  constexpr int NX = CI * CHS, NW = CI * 64 * 32;
  constexpr int NXL = (NX + 255) / 256 * 256, NWL = (NW + 255) / 256 * 256;
  constexpr int BUF = NXL + NWL;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;
  const int col = lane & 31, kh = lane >> 5;
  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int qz0 = tzi * TZ, qy0 = tyi * TY, qx0 = txi * 32;
  if (wave >= 4) {
    tr_loader<TZ, TY, 32>(X, Wt, p, lds, wv, lane, b, qz0, qy0, qx0);
    return;
  }
  const int wz = wv / TY, wy = wv % TY;
  const int bBo = kh * 2 * CHS + (wz + 1) * PS + (wy + 1) * XP + (col + 4);
  const int aBo = NXL + kh * 2 * 64 * 32 + col;

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  __builtin_amdgcn_s_barrier();  // chunk 0 has landed
  int buf = 0;
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
    const float* bB = lds + buf * BUF + bBo;
    const float* aB = lds + buf * BUF + aBo;
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      float xn[27];
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            xn[(dz * 3 + dy) * 3 + dx] = bB[cl * CHS + (dz - 1) * PS + (dy - 1) * XP + (dx - 1)];
#pragma unroll
      for (int cls = 0; cls < 8; ++cls) {
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) {
          const int a = tp >> 2, bb = (tp >> 1) & 1, c = tp & 1;
          const int kidx = (tap_k(pz, a) * 4 + tap_k(py, bb)) * 4 + tap_k(px, c);
          const int didx = ((tap_d(pz, a) + 1) * 3 + (tap_d(py, bb) + 1)) * 3 + (tap_d(px, c) + 1);
          const float av = aB[(cl * 64 + kidx) * 32];
          acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xn[didx], acc[cls], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_barrier();  // the next chunk has landed, everyone is done reading `buf`
    buf ^= 1;
  }

#ifdef FS_ABLATION
  if (p.ab & 4) return;
#endif
  const int qz = qz0 + wz, qy = qy0 + wy, qx = qx0 + col;
  if (qz < p.Dq && qy < p.Hq) {  // wave-uniform: every lane takes part in the lane exchange of store8_quad
    const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const bool ok = co < p.Cout;
      const int cc = ok ? co : 0;
      const float bv = (bias && ok) ? bias[cc] : 0.f;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = acc[c][r] + bv;
      // 16-byte stores (neighbouring positions swap a pair): the epilogue is store-issue-bound, 12-16 % of a launch
      store8_quad(Y + ((size_t)b * p.CoutT + cc) * yvol, v, qz, qy, qx, ok, p,
                  p.Z ? p.Z + ((size_t)b * p.CoutT + cc) * yvol : nullptr, p.Z ? p.slope[p.nslope == 1 ? 0 : cc] : 0.f);
    }
  }
}

template <int TZ, int TY>
__global__ __launch_bounds__(512, 2) void convtr_mfma16_ws_kernel(const float* __restrict__ X,
                                                               const float* __restrict__ Wt,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ Y, TP p) {
  static_assert(TZ * TY == 4, "one 32-position row per matrix wave");
  constexpr int CI = 4;
  constexpr int ZT = TZ + 2, YT = TY + 2, XP = 40;
  constexpr int PS = YT * XP, CHS = ZT * PS;
  constexpr int NX = CI * CHS, NW = CI * 64 * 16;
  constexpr int NXL = (NX + 255) / 256 * 256, NWL = (NW + 255) / 256 * 256;
  constexpr int BUF = NXL + NWL;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;
  const int col = lane & 15, kq = lane >> 4;
  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int qz0 = tzi * TZ, qy0 = tyi * TY, qx0 = txi * 32;
  if (wave >= 4) {
    tr_loader<TZ, TY, 16>(X, Wt, p, lds, wv, lane, b, qz0, qy0, qx0);
    return;
  }
  const int wz = wv / TY, wy = wv % TY;
  const int bBo = kq * CHS + (wz + 1) * PS + (wy + 1) * XP + (col + 4);
  const int aBo = NXL + kq * 64 * 16 + col;

  f32x4 acc[2][8];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[n][c][r] = 0.f;

  __builtin_amdgcn_s_barrier();  // chunk 0 has landed
  int buf = 0;
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
    const float* bB = lds + buf * BUF + bBo;
    const float* aB = lds + buf * BUF + aBo;
    float xn[2][27];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            xn[n][(dz * 3 + dy) * 3 + dx] = bB[16 * n + (dz - 1) * PS + (dy - 1) * XP + (dx - 1)];
#pragma unroll
    for (int cls = 0; cls < 8; ++cls) {
      const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
        const int a = tp >> 2, bb = (tp >> 1) & 1, c = tp & 1;
        const int kidx = (tap_k(pz, a) * 4 + tap_k(py, bb)) * 4 + tap_k(px, c);
        const int didx = ((tap_d(pz, a) + 1) * 3 + (tap_d(py, bb) + 1)) * 3 + (tap_d(px, c) + 1);
        const float av = aB[kidx * 16];
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[n][cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xn[n][didx], acc[n][cls], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();  // the next chunk has landed, everyone is done reading `buf`
    buf ^= 1;
  }

#ifdef FS_ABLATION
  if (p.ab & 4) return;
#endif
  const int qz = qz0 + wz, qy = qy0 + wy;
  if (qz < p.Dq && qy < p.Hq) {  // wave-uniform: every lane takes part in the lane exchange of store8_quad
    const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int qx = qx0 + 16 * n + col;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * kq + r;
        const bool ok = co < p.Cout;
        const int cc = ok ? co : 0;
        const float bv = (bias && ok) ? bias[cc] : 0.f;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = acc[n][c][r] + bv;
        store8_quad(Y + ((size_t)b * p.Cout + cc) * yvol, v, qz, qy, qx, ok, p,
                    p.Z ? p.Z + ((size_t)b * p.Cout + cc) * yvol : nullptr, p.Z ? p.slope[p.nslope == 1 ? 0 : cc] : 0.f);
      }
    }
  }
}

// ---- up to 12 output channels: v_mfma_f32_16x16x4_f32 with ALL THREE output parities in the rows ---------------
// Padding 6 (or 1, or 11) channels to a 16-row tile wastes most of it; but the parities of the output can share one
// B operand if they are paired across neighbouring positions: along every axis, output 2q (parity 0 of input
// position q: taps (q, k 1), (q-1, k 3)) and output 2q - 1 (parity 1 of position q-1: taps (q, k 0), (q-1, k 2))
// read inputs q and q-1 only.  So position (qz, qy, q) owns the 8 outputs (2qz - pz, 2qy - py, 2q - px), p in
// {0,1}^3, all of them functions of the 2x2x2 inputs {q, q-1}^3: the matrix rows are (parity, channel) -- 8 * Cout
// of them, 48 = three full 16-row tiles for the 6-channel flow head (a channels-only tile: 37.5 % useful; the
// vector-ALU kernel it replaces measured 28 % of peak), 8 of 16 for the 1-channel mask head, 88 / 96 of 96 for the
// 11 / 12-channel input gradient of conv0 -- and the reduction runs over 8 neighbours x Cin instead of 8 classes x 8
// taps x Cin.  Positions run over 0..Di, 0..Hi, 0..Wi (one more than the input per axis: the last odd outputs).
// Rows inside a 16-row tile: px * 8 + slot, slot = item % 8, item = (pz * 2 + py) * Cout + co, so that one
// v_permlane32_swap brings both x parities of an item into one lane.  The re-laid-out weights of ALL input channels
// stay in LDS (W'[ci][neighbour][row], <= 98 KB); persistent workgroups (one per CU, contiguous brick ranges) load
// them once; the loader waves stream only the input bricks (4 channels x 3 x 3 rows) and run one (brick, chunk)
// item ahead of the matrix waves across brick boundaries.  A matrix wave owns one (z, y) position row of the 2x2
// brick and NT position tiles of it.
constexpr int p8_ws_ci(int rt) { return 128 * rt + 16; }  // floats per input channel (+16: LDS bank spread)
__host__ __device__ constexpr int p8_k(int par, int d) { return par == 0 ? (d == 0 ? 1 : 3) : (d == 0 ? 0 : 2); }

#ifdef FS_TR_STAMPS
__device__ unsigned long long fs_tr_dbg[4 * 8];
#define TSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dt[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

// RT 16-row tiles, NT position tiles per wave (compile-time: a chunk is straight-line code); an x brick is
// 16 (NT - 1) positions plus one more tile that only the LAST brick of a row needs (q = Wi)
template <int RT, int NT, int CINP>
__global__ __launch_bounds__(512, 2) void convtr_p8_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                        const float* __restrict__ bias, float* __restrict__ Y,
                                                        TP p, int per) {
  constexpr int CI = 4, XB = 16 * (NT - 1);
  constexpr int ZT = 3, YT = 3, XP = XB + 16;         // rows q-1..q+1 of both axes; row = [q0 - 4, q0 + XB + 12)
  constexpr int PS = YT * XP;
  constexpr int CHS = ZT * PS + ((ZT * PS) % 32 == 16 ? 0 : 16);  // the two channel groups of a half-wave: different banks
  constexpr int NX = CI * CHS;
  constexpr int NXL = (NX + 255) / 256 * 256;
  constexpr int WSCI = p8_ws_ci(RT);
  constexpr int NW = CINP * WSCI;                      // whole weight table
  constexpr int NWL = (NW + 255) / 256 * 256;
  static_assert((NWL + 2 * NXL) * 4 <= 160 * 1024, "weights + two input buffers fit the CU's LDS");
  static_assert(NT % 2 == 1, "tiles are processed in pairs + one");
  __shared__ __attribute__((aligned(16))) float lds[NWL + 2 * NXL];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;
  const long long t0 = (long long)blockIdx.x * per, t1 = min(t0 + per, p.tiles);
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  auto decode = [&](long long tile, int& b, int& qz0, int& qy0, int& q0) {
    const int txi = (int)(tile % p.tx); tile /= p.tx;
    const int tyi = (int)(tile % p.ty); tile /= p.ty;
    const int tzi = (int)(tile % p.tz);
    b = (int)(tile / p.tz);
    qz0 = tzi * 2; qy0 = tyi * 2; q0 = txi * XB;
  };

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass has neither the buffer-resource type nor the LDS-DMA builtin)
    __builtin_amdgcn_s_setprio(3);  // (a loader wave issues a handful of instructions per period: they should not queue behind the matrix wave's)
    constexpr int NXW = (NXL / 256 + 3) / 4, NWW = (NWL / 256 + 3) / 4;
    {  // the whole weight table (the layer's (Cin + 3) / 4 * 4 channels: the workspace holds no more), once
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, (short)0, (p.Cin + 3) / 4 * 4 * WSCI * 4,
                                                                     0x00020000);
#pragma unroll
      for (int k = 0; k < NWW; ++k)
        if (256 * (wv + 4 * k) < NWL)  // wave-uniform
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(lds + 256 * (wv + 4 * k)), 16,
                                                   (unsigned)(256 * (wv + 4 * k) + 4 * lane) * 4u, 0, 0, 0);
    }
    unsigned xoff[NXW];
    int b = 0;
    auto offsets = [&](long long tile) {
      int qz0, qy0, q0;
      decode(tile, b, qz0, qy0, q0);
#pragma unroll
      for (int k = 0; k < NXW; ++k) {
        const int i = 256 * (wv + 4 * k) + 4 * lane;
        const int c = i / CHS, r1 = i - c * CHS;
        const int z = r1 / PS, r2 = r1 - z * PS;
        const int y = r2 / XP, x = r2 - y * XP;
        const int gz = qz0 - 1 + z, gy = qy0 - 1 + y, gx = q0 - 4 + x;
        const bool ok = i < NX && z < ZT && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
        xoff[k] = ok ? ((unsigned)c * (unsigned)xvol + ((unsigned)gz * p.Hi + gy) * p.Wi + gx) * 4u : DMA_OOB;
      }
    };
    auto stage = [&](int c0, int buf) {
      const int nch = (p.Cin - c0 < CI) ? (p.Cin - c0) : CI;  // channels past Cin read as zero
      __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(X + ((size_t)b * p.Cin + c0) * xvol), (short)0, (int)((unsigned)nch * (unsigned)xvol * 4u), 0x00020000);
      float* base = lds + NWL + buf * NXL;
#pragma unroll
      for (int k = 0; k < NXW; ++k)
        if (256 * (wv + 4 * k) < NXL)  // wave-uniform
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + 256 * (wv + 4 * k)), 16, xoff[k], 0, 0, 0);
    };
    int buf = 0;
    if (t0 < t1) {
      offsets(t0);
      stage(0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // weights and the first item have landed
    for (long long tile = t0; tile < t1; ++tile) {
      for (int c0 = 0; c0 < p.Cin; c0 += CI) {
        // stage the NEXT item of the (brick, chunk) stream
        if (c0 + CI < p.Cin) {
          stage(c0 + CI, buf ^ 1);
        } else if (tile + 1 < t1) {
          offsets(tile + 1);
          stage(0, buf ^ 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
      }
    }
#else
    (void)xvol;
#endif
    return;
  }

  // ---- matrix waves
  const int col = lane & 15, kq = lane >> 4;
  const int wz = wv >> 1, wy = wv & 1;
  const int bBo = NWL + kq * CHS + (wz + 1) * PS + (wy + 1) * XP + (col + 4);
  __builtin_amdgcn_s_barrier();  // weights and the first item have landed
  int buf = 0;
#ifdef FS_TR_STAMPS
  unsigned long long dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  for (long long tile = t0; tile < t1; ++tile) {
    int b, qz0, qy0, q0;
    decode(tile, b, qz0, qy0, q0);
    // positions of this x brick that are stored: [q0, q0 + XB), plus q = Wi in the last brick of the row; the tiles
    // beyond are computed on zeros / the neighbour's columns and dropped
    const int qend = (q0 + XB >= p.Wi) ? p.Wi + 1 : q0 + XB;
    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][n][r] = 0.f;

    for (int c0 = 0; c0 < p.Cin; c0 += CI) {
      // A operands of the chunk: W'[c0 + kq][neighbour d][row = 16 rt + (lane & 15)]
      const float* aB = lds + (size_t)(c0 + kq) * WSCI + col;
      float av[8][RT];
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) av[d][rt] = aB[(d * RT + rt) * 16];
      const float* bB = lds + bBo + buf * NXL;
      // B operands: the 2x2x2 neighbourhood {q, q-1}^3 of two position tiles, read one tile pair ahead of the MFMAs
      float xa[2][8], xb[2][8];
      auto load_x = [&](int n, float (&xn)[8]) {
#pragma unroll
        for (int d = 0; d < 8; ++d) xn[d] = bB[16 * n - ((d >> 2) & 1) * PS - ((d >> 1) & 1) * XP - (d & 1)];
      };
      // two tiles at a time: consecutive MFMAs go to different accumulators (40-cycle dependent latency)
      auto mma_pair = [&](int n, const float (&x0)[8], const float (&x1)[8], bool two) {
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d][rt], x0[d], acc[rt][n], 0, 0, 0);
            if (two) acc[rt][n + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d][rt], x1[d], acc[rt][n + 1], 0, 0, 0);
          }
      };
      load_x(0, xa[0]);
      if (NT > 1) load_x(1, xa[1]);
#pragma unroll
      for (int n = 0; n < NT; n += 4) {
        if (n + 2 < NT) { load_x(n + 2, xb[0]); if (n + 3 < NT) load_x(n + 3, xb[1]); }
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(n, xa[0], xa[1], n + 1 < NT);
        if (n + 2 < NT) {
          if (n + 4 < NT) { load_x(n + 4, xa[0]); if (n + 5 < NT) load_x(n + 5, xa[1]); }
          __builtin_amdgcn_sched_barrier(0);
          mma_pair(n + 2, xb[0], xb[1], n + 3 < NT);
        }
      }
      TSTAMP(0);
      __builtin_amdgcn_s_barrier();  // the next item has landed, everyone is done reading `buf`
      TSTAMP(1);
      buf ^= 1;
    }

    // D row = 4 kq + r of tile rt: x parity = kq >> 1 (0: x = 2q, 1: x = 2q - 1), slot = 4 (kq & 1) + r.
    // The epilogue is store-ISSUE-bound (s_memtime stamps: ~75 cycles per store wave-instruction and CU whatever its
    // width), so the stores are made as wide as the layout allows: (1) one v_permlane32_swap per register pair
    // (r, r + 2) brings both x parities of a slot into one lane -- lanes 0-31 hold (x = 2q - 1, x = 2q) of slot
    // 4 (kq & 1) + r, lanes 32-63 of slot + 2; (2) neighbouring lanes swap one (odd, even) pair (DPP quad_perm
    // [1,0,3,2]): an even lane ends up with the four consecutive outputs 2q - 1 .. 2q + 2 of slot r = 0, its odd
    // neighbour with 2q - 3 .. 2q of slot r = 1: one 16-byte store per lane, tile and position tile.
    const int qz = qz0 + wz, qy = qy0 + wy;
    if (qz <= p.Di && qy <= p.Hi) {
      const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
      const bool evn = (col & 1) == 0;
      const int slot = 4 * (kq & 1) + (evn ? 0 : 1) + 2 * (lane >> 5);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int item = 8 * rt + slot;
        const int pzy = item / p.Cout, co = item - pzy * p.Cout;
        const int zo = 2 * qz - (pzy >> 1), yo = 2 * qy - (pzy & 1);
        const bool iok = item < 4 * p.Cout && zo >= 0 && zo < p.Dout && yo >= 0 && yo < p.Hout;
        const float bv = (bias && iok) ? bias[co] : 0.f;
        float* __restrict__ row = Y + (((size_t)b * p.Cout + (iok ? co : 0)) * p.Dout + (iok ? zo : 0)) * ((size_t)p.Hout * p.Wout) +
                                  (size_t)(iok ? yo : 0) * p.Wout;
        const float* __restrict__ arow = p.addend ? p.addend + (row - Y) : nullptr;
        (void)yvol;
        // the running-flow / mask addend of this row tile: all its 16-byte loads first, so that they are in flight
        // together instead of one load -> wait -> add -> store chain per position tile
        float4 apre[NT];
        if (arow != nullptr) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int qa = evn ? q0 + 16 * n + col : q0 + 16 * n + col - 1;
            const int x0 = 2 * qa - 1;
            const bool full = iok && qa >= 1 && qa + 1 < qend && x0 + 3 < p.Wout;
            apre[n] = full ? *reinterpret_cast<const float4*>(arow + x0) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int q = q0 + 16 * n + col;
          const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[rt][n][0]), __float_as_uint(acc[rt][n][2]),
                                                           false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[rt][n][1]), __float_as_uint(acc[rt][n][3]),
                                                           false, false);
          // s*[0] = x 2q (even), s*[1] = x 2q - 1 (odd).  Keep the slot of this lane's parity, send the other
          const unsigned keep_e = evn ? s0[0] : s1[0], keep_o = evn ? s0[1] : s1[1];
          const unsigned send_e = evn ? s1[0] : s0[0], send_o = evn ? s1[1] : s0[1];
          const float re = __uint_as_float(__builtin_amdgcn_update_dpp(0u, send_e, 0xB1, 0xF, 0xF, false)) + bv;
          const float ro = __uint_as_float(__builtin_amdgcn_update_dpp(0u, send_o, 0xB1, 0xF, 0xF, false)) + bv;
          const float ke = __uint_as_float(keep_e) + bv, ko = __uint_as_float(keep_o) + bv;
          // even lane: x = 2q - 1 .. 2q + 2;  odd lane: x = 2q - 3 .. 2q
          const int qa = evn ? q : q - 1;  // first of the two positions in this lane's quad
          float4 v = evn ? make_float4(ko, ke, ro, re) : make_float4(ro, re, ko, ke);
          if (iok) {
            const int x0 = 2 * qa - 1;     // outputs x0 .. x0 + 3 <- positions qa, qa, qa + 1, qa + 1
            const bool full = qa >= 1 && qa + 1 < qend && x0 + 3 < p.Wout;
            if (full) {
              if (arow) { v.x += apre[n].x; v.y += apre[n].y; v.z += apre[n].z; v.w += apre[n].w; }
              *reinterpret_cast<float4*>(row + x0) = v;  // 4-byte aligned 16-byte store
            } else {
              const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int x = x0 + e, pos = qa + (e >> 1);
                if (pos >= 0 && pos < qend && x >= 0 && x < p.Wout) row[x] = vv[e] + (arow ? arow[x] : 0.f);
              }
            }
          }
        }
      }
    }
    TSTAMP(2);
  }
#ifdef FS_TR_STAMPS
  if (blockIdx.x == 3 && lane == 0) {
    for (int i = 0; i < 8; ++i) fs_tr_dbg[wv * 8 + i] = dt[i];
    fs_tr_dbg[wv * 8 + 7] = (unsigned long long)(t1 - t0);
  }
#endif
}

// CUs of the device the call runs on (per-device table: a process may drive several GPUs; an int store is the only shared write)
static int tr_ncu() {
  static int ncu_of[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int ncu = ncu_of[dev];
  if (ncu == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    ncu_of[dev] = ncu = n;
  }
  return ncu;
}

template <int RT, int NT, int CINP = 32>
void launch_p8(const float* x, const float* ws, const float* bias, float* y, const TP& p, hipStream_t st) {
  const int ncu = tr_ncu();  // one persistent workgroup per CU
  const long long nwg = p.tiles < ncu ? p.tiles : ncu;
  const int per = (int)((p.tiles + nwg - 1) / nwg);
  const unsigned grid = (unsigned)((p.tiles + per - 1) / per);
  hipLaunchKernelGGL((convtr_p8_kernel<RT, NT, CINP>), dim3(grid), dim3(512), 0, st, x, ws, bias, y, p, per);
}

// NP input-grid positions (consecutive qy) per thread: every scalar-loaded weight feeds NP FMAs.  The
// kernel is bound by the scalar weight stream (64*CO dwords per input channel and wave through ~100
// SGPRs), not by the vector ALUs or the address path -- packed FMAs / an LDS-staged input brick did not
// move it, a second position per thread did.
template <int CO, int NP>
__global__ __launch_bounds__(256) void convtr_valu_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          TP p) {
  const int Hq2 = (p.Hq + NP - 1) / NP;
  const long long nq = (long long)p.B * p.Dq * Hq2 * p.Wq;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= nq) return;
  const int qx = (int)(e % p.Wq);
  long long r = e / p.Wq;
  const int qy = (int)(r % Hq2) * NP; r /= Hq2;
  const int qz = (int)(r % p.Dq);
  const int b = (int)(r / p.Dq);

  // offsets of the 3 x (NP+2) x 3 neighbourhood inside one input channel; -1 = outside (reads as zero)
  constexpr int NY = NP + 2;
  int off[3 * NY * 3];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int dy = 0; dy < NY; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int z = qz + dz - 1, y = qy + dy - 1, x = qx + dx - 1;
        const bool ok = z >= 0 && z < p.Di && y >= 0 && y < p.Hi && x >= 0 && x < p.Wi;
        off[(dz * NY + dy) * 3 + dx] = ok ? (z * p.Hi + y) * p.Wi + x : -1;
      }

  float acc[NP][CO][8];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    const float bv = (bias && co < p.Cout) ? bias[co] : 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[q][co][c] = bv;
  }
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  const float* xc = X + (size_t)b * p.Cin * xvol;
  for (int ci = 0; ci < p.Cin; ++ci, xc += xvol) {
    float xn[3 * NY * 3];
#pragma unroll
    for (int i = 0; i < 3 * NY * 3; ++i) xn[i] = off[i] >= 0 ? xc[off[i]] : 0.f;
    const float* wc = W + (size_t)ci * p.Cout * 64;  // wave-uniform: scalar loads
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      if (co < p.Cout) {
#pragma unroll
        for (int cls = 0; cls < 8; ++cls) {
          const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
#pragma unroll
          for (int tp = 0; tp < 8; ++tp) {
            const int a = tp >> 2, bb = (tp >> 1) & 1, c = tp & 1;
            const int kidx = (tap_k(pz, a) * 4 + tap_k(py, bb)) * 4 + tap_k(px, c);
            const float wv = wc[co * 64 + kidx];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
              const int didx = ((tap_d(pz, a) + 1) * NY + (tap_d(py, bb) + 1 + q)) * 3 + (tap_d(px, c) + 1);
              acc[q][co][cls] = fmaf(xn[didx], wv, acc[q][co][cls]);
            }
          }
        }
      }
    }
  }
  const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int co = 0; co < CO; ++co)
      if (co < p.Cout && qy + q < p.Hq)
        store8(Y + ((size_t)b * p.Cout + co) * yvol, acc[q][co], qz, qy + q, qx, p,
               p.Z ? p.Z + ((size_t)b * p.Cout + co) * yvol : nullptr, p.Z ? p.slope[p.nslope == 1 ? 0 : co] : 0.f);
}

template <int CO>
void launch_valu(const float* x, const float* w, const float* bias, float* y, const TP& p, hipStream_t st) {
  constexpr int NP = 2;  // 4 was slower for the 1-channel mask head (0.64 vs 0.58 ms)
  const long long nq = (long long)p.B * p.Dq * ((p.Hq + NP - 1) / NP) * p.Wq;
  hipLaunchKernelGGL((convtr_valu_kernel<CO, NP>), dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, x, w, bias,
                     y, p);
}

}  // namespace

#ifdef FS_TR_STAMPS
extern "C" int fs_debug_tr_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs_tr_dbg), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

extern "C" long long fs_conv3d_tr_ws_floats(int Cin, int Cout) {
  if (Cin < 1 || Cout < 1 || (Cout > 32 && (Cout % 32 != 0 || Cout > 128))) return -1;
  const int slices = Cout > 32 ? Cout / 32 : 1;  // 32-channel slices, each with its own re-laid-out weights
  if (Cout > 32) Cout = 32;
  const long long cinp = (Cin + 3) / 4 * 4;
  // W'[ci][neighbour][row] (+ pad) of the all-parities kernel (<= 12 channels), or the slabs of the class kernels
  const long long p8 = Cout <= 12 ? cinp * p8_ws_ci(Cout <= 2 ? 1 : (Cout <= 6 ? 3 : 6)) : 0;
  long long cls = Cout <= 6 ? 0 : cinp * 64 * (Cout <= 16 ? 16 : 32);
  if (Cout > 16 && t3_slab_words(Cin) > cls) cls = t3_slab_words(Cin);  // the split-bf16 slabs (convtr_s3.hpp)
  if (Cout <= 16 && t3_slab_words16(Cin) > cls) cls = t3_slab_words16(Cin);
  return (p8 > cls ? p8 : cls) * slices;
}

static int conv3d_tr_slice(const float* x, const float* w, const float* bias, const float* slope, int nslope,
                          const float* addend, float* y, float* z, float* ws, int B, int Cin, int Cout, int Di,
                          int Hi, int Wi, int Dout, int Hout, int Wout, fs_stream_t stream, int CoutT = 0, int slices = 1,
                          WprepPlan* plan = nullptr) {
  // w == NULL: `ws` already holds the re-laid-out weights (fs_conv3d_wprep_batch); plan: record the re-layout this
  // shape needs and launch nothing
  FS_REQUIRE_PTR(x);
  if (plan == nullptr) FS_REQUIRE_PTR(y);
  if (z != nullptr && (slope == nullptr || (nslope != 1 && nslope != Cout))) return FS_ERR_ARG;
  if (B < 1 || Cin < 1 || Cout < 1 || Di < 1 || Hi < 1 || Wi < 1) return FS_ERR_SHAPE;
  if (Cout > 32) return FS_ERR_ARG;
  if (CoutT == 0) CoutT = Cout;
  // transposed convolution: out = 2 in; input gradient of Conv3d(4,2,1): in_x = 2 out or 2 out + 1
  if ((Dout != 2 * Di && Dout != 2 * Di + 1) || (Hout != 2 * Hi && Hout != 2 * Hi + 1) ||
      (Wout != 2 * Wi && Wout != 2 * Wi + 1))
    return FS_ERR_SHAPE;
  if ((long long)Di * Hi * Wi >= (1ll << 31) || (long long)Dout * Hout * Wout >= (1ll << 31))
    return FS_ERR_SHAPE;
  if ((long long)4 * Di * Hi * Wi * 4 >= (1ll << 32)) return FS_ERR_SHAPE;  // 32-bit chunk offsets
  TP p;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.CoutT = CoutT; p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  p.Dout = Dout; p.Hout = Hout; p.Wout = Wout;
  p.Dq = (Dout + 1) / 2; p.Hq = (Hout + 1) / 2; p.Wq = (Wout + 1) / 2;
  p.slope = slope; p.Z = z; p.nslope = nslope;
  p.addend = addend; p.Ybase = y;
  p.wslice = 0;
  hipStream_t st = (hipStream_t)stream;
  if ((long long)B * p.Dq * p.Hq * p.Wq >= (1ll << 31) * 256) return FS_ERR_SHAPE;
  static const bool reg_only = FS_AB_ENV("FLOWSCI_TR_REG");
  // (the 6-row-tile instantiation for 7..12 channels works -- tests/test_gpu_losses.py covers it through
  // FLOWSCI_TR_P8_ALL=1 -- but measured 2.50 vs 2.44 ms against the 16-row class kernel on the 32 -> 11 input
  // gradient at 128^3, so those layers stay there)
  static const bool p8_all = FS_AB_ENV("FLOWSCI_TR_P8_ALL");
  static const bool s3_small = FS_AB_ENV("FLOWSCI_TR_S3_SMALL");  // (measurement: <= 6 channels on the 16-row split-bf16 form)
  if (!s3_small && Cout <= (p8_all ? 12 : 6) && !reg_only && z == nullptr && ws != nullptr && Cin <= (Cout <= 6 ? 64 : 32) && Dout == 2 * Di && Hout == 2 * Hi &&
      Wout == 2 * Wi && Wi % 4 == 0 && (((uintptr_t)x | (uintptr_t)ws) & 15) == 0 &&
      (long long)4 * Di * Hi * Wi * 4 < (1ll << 31)) {
    // all-parities-in-rows MFMA kernel: 2x2 position rows x 128 (64) positions per brick; positions 0..Di, 0..Hi, 0..Wi
    const int rt = Cout <= 2 ? 1 : (Cout <= 6 ? 3 : 6);
    const bool wide = Wi > 64 && rt < 6;  // 128- or 64-position x bricks (6 row tiles: 64, for the accumulators)
    p.tz = fs::cdiv(Di + 1, 2); p.ty = fs::cdiv(Hi + 1, 2); p.tx = fs::cdiv(Wi, wide ? 128 : 64);
    p.tiles = (long long)B * p.tz * p.ty * p.tx;
    if (p.tiles >= 16 && p.tiles < (1ll << 31)) {
      const int cinp = (Cin + 3) / 4 * 4;
      wprep_do(wprep_job(FS_WPREP_P8, w, ws, (long long)cinp * p8_ws_ci(rt), Cin, Cout, cinp, rt), plan, st);
      if (plan != nullptr) return FS_OK;
      if (Cin > 32) {  // block0's heads (64 input channels): the weight table is twice as large
        if (rt == 1) { if (wide) launch_p8<1, 9, 64>(x, ws, bias, y, p, st); else launch_p8<1, 5, 64>(x, ws, bias, y, p, st); }
        else { if (wide) launch_p8<3, 9, 64>(x, ws, bias, y, p, st); else launch_p8<3, 5, 64>(x, ws, bias, y, p, st); }
      } else if (rt == 1) { if (wide) launch_p8<1, 9>(x, ws, bias, y, p, st); else launch_p8<1, 5>(x, ws, bias, y, p, st); }
      else if (rt == 3) { if (wide) launch_p8<3, 9>(x, ws, bias, y, p, st); else launch_p8<3, 5>(x, ws, bias, y, p, st); }
      else launch_p8<6, 5>(x, ws, bias, y, p, st);
      FS_LAUNCH_CHECK();
      return FS_OK;
    }
  }
  if (Cout <= 6 && !s3_small) {
    if (plan != nullptr) return FS_OK;        // the vector-ALU kernels read `w` as stored: nothing to prepare
    if (w == nullptr) return FS_ERR_NULLPTR;  // ... and therefore need it
    if (Cout == 1) launch_valu<1>(x, w, bias, y, p, st);
    else if (Cout <= 2) launch_valu<2>(x, w, bias, y, p, st);
    else if (Cout <= 4) launch_valu<4>(x, w, bias, y, p, st);
    else launch_valu<6>(x, w, bias, y, p, st);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
  FS_REQUIRE_PTR(ws);
  const int cinp = (Cin + 3) / 4 * 4;
  p.tz = fs::cdiv(p.Dq, 2); p.ty = fs::cdiv(p.Hq, 2); p.tx = fs::cdiv(p.Wq, 32);
  p.tiles = (long long)B * p.tz * p.ty * p.tx;
  if (p.tiles >= (1ll << 31)) return FS_ERR_SHAPE;
#ifdef FS_ABLATION
  static const int tr_ab = (int)FS_AB_ENV_LL("FLOWSCI_TR_AB", 0);
  p.ab = tr_ab;
#endif
  // loader-wave kernels: 16-byte pieces (Wi % 4 == 0, 16-byte aligned input and workspace), 31-bit byte offsets
  // inside a staged 4-channel chunk.  They also win when the launch cannot fill the chip (block0's 128 -> 64
  // deconvolution at 16^3: 128 bricks per 32-channel slice, 0.42 -> 0.29 ms): one 8-wave workgroup per CU overlaps
  // its staging with its MFMAs, two half-empty 4-wave workgroups do not.  `FLOWSCI_TR_REG=1`: the register-staged kernels.
  const bool ws_ok = !reg_only && Wi % 4 == 0 && (((uintptr_t)x | (uintptr_t)ws) & 15) == 0 && p.tiles * slices >= 128 &&
                     (long long)4 * Di * Hi * Wi * 4 < (1ll << 31);
  static const bool no_s3 = FS_AB_ENV("FLOWSCI_TR_NO_S3");
  const long long t3 = (long long)B * fs::cdiv(Di, T3_TZ) * fs::cdiv(Hi, T3_TY) * fs::cdiv(Wi, T3_TW);
  const bool t3_ok = !no_s3 && !reg_only && Dout == 2 * Di && Hout == 2 * Hi && Wout == 2 * Wi && Wi % 4 == 0 &&
                     (((uintptr_t)x | (uintptr_t)ws) & 15) == 0 && (long long)Di * Hi * Wi * 4 < (1ll << 31) &&
                     t3 * slices >= 256 && t3 < (1ll << 31) && (long long)32 * Dout * Hout * Wout * 4 < (1ll << 32);
  if (Cout <= 16) {
    if (t3_ok && slices == 1) {  // round 5: the 16-row split-bf16 form (convtr_s3.hpp)
      wprep_do(wprep_job(FS_WPREP_TRS3_16, w, ws, t3_slab_words16(Cin), Cin, Cout, (Cin + 3) / 4, p.CoutT), plan, st);
      if (plan != nullptr) return FS_OK;
      p.tz = fs::cdiv(Di, T3_TZ); p.ty = fs::cdiv(Hi, T3_TY); p.tx = fs::cdiv(Wi, T3_TW);
      p.tiles = t3;
      const int ncu = tr_ncu();
      const long long gx = ncu < p.tiles ? ncu : p.tiles;
      hipLaunchKernelGGL(convtr_s3_kernel<true>, dim3((unsigned)gx, 1), dim3(64 * (T3_NMW + T3_NLW)), 0, st, x,
                         reinterpret_cast<const unsigned*>(ws), bias, y, p);
      FS_LAUNCH_CHECK();
      return FS_OK;
    }
    wprep_do(wprep_job(FS_WPREP_TR16, w, ws, (long long)cinp * 64 * 16, Cin, Cout, cinp), plan, st);
    if (plan != nullptr) return FS_OK;
    if (ws_ok)
      hipLaunchKernelGGL((convtr_mfma16_ws_kernel<2, 2>), dim3((unsigned)p.tiles), dim3(512), 0, st, x, ws, bias, y, p);
    else
      hipLaunchKernelGGL((convtr_mfma16_kernel<2, 2>), dim3((unsigned)p.tiles), dim3(256), 0, st, x, ws, bias, y, p);
  } else {
    // round 5: fp32 accuracy on the bf16 matrix rate (three bf16 pieces per operand, six products: convtr_s3.hpp) where the
    // output is exactly twice the input, rows are 16-byte pieces and the 2 x 3 x 32-position bricks fill the chip
    if (t3_ok) {
      p.wslice = t3_slab_words(Cin);
      for (int sl = 0; sl < slices; ++sl)
        wprep_do(wprep_job(FS_WPREP_TRS3, w ? w + (size_t)sl * 32 * 64 : nullptr, ws + (size_t)sl * p.wslice, p.wslice, Cin,
                           Cout, (Cin + 3) / 4, p.CoutT), plan, st);
      if (plan != nullptr) return FS_OK;
      p.tz = fs::cdiv(Di, T3_TZ); p.ty = fs::cdiv(Hi, T3_TY); p.tx = fs::cdiv(Wi, T3_TW);
      p.tiles = t3;
      // persistent workgroups: one per CU over all slices
      const int ncu = tr_ncu();
      long long gx = ncu / slices < 1 ? 1 : ncu / slices;
      if (gx > p.tiles) gx = p.tiles;
      hipLaunchKernelGGL(convtr_s3_kernel<false>, dim3((unsigned)gx, slices), dim3(64 * (T3_NMW + T3_NLW)), 0, st, x,
                         reinterpret_cast<const unsigned*>(ws), bias, y, p);
      FS_LAUNCH_CHECK();
      return FS_OK;
    }
    p.wslice = (long long)cinp * 64 * 32;
    for (int sl = 0; sl < slices; ++sl)
      wprep_do(wprep_job(FS_WPREP_TR32, w ? w + (size_t)sl * 32 * 64 : nullptr, ws + (size_t)sl * p.wslice, p.wslice, Cin,
                         Cout, cinp, p.CoutT), plan, st);
    if (plan != nullptr) return FS_OK;
    if (ws_ok)
      hipLaunchKernelGGL((convtr_mfma_ws_kernel<2, 2>), dim3((unsigned)p.tiles, slices), dim3(512), 0, st, x, ws, bias, y, p);
    else
      hipLaunchKernelGGL((convtr_mfma_kernel<2, 2>), dim3((unsigned)p.tiles, slices), dim3(256), 0, st, x, ws, bias, y, p);
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// More than 32 output channels (block0's 128 -> 64 deconvolution): 32-channel slices of the output, one launch each
// (the same workspace: launches on one stream run in order); the slice's pointers are offset to its first channel
// and the kernels stride batches by the whole tensor's channel count.
static int conv3d_tr_impl(const float* x, const float* w, const float* bias, const float* slope, int nslope,
                          const float* addend, float* y, float* z, float* ws, int B, int Cin, int Cout, int Di,
                          int Hi, int Wi, int Dout, int Hout, int Wout, fs_stream_t stream, WprepPlan* plan = nullptr) {
  if (Cout <= 32) return conv3d_tr_slice(x, w, bias, slope, nslope, addend, y, z, ws, B, Cin, Cout, Di, Hi, Wi, Dout, Hout,
                                         Wout, stream, 0, 1, plan);
  if (Cout % 32 != 0 || Cout > 128 || Dout < 1 || Hout < 1 || Wout < 1) return FS_ERR_ARG;
  // all 32-channel slices in one launch (grid.y): pointers of slice 0, the kernels step to theirs
  return conv3d_tr_slice(x, w, bias, slope, nslope == 1 ? 1 : (nslope ? 32 : 0), addend, y, z, ws, B, Cin, 32, Di, Hi, Wi, Dout,
                         Hout, Wout, stream, Cout, Cout / 32, plan);
}

// The re-layout job(s) fs_conv3d_tr{,_add} (has_prelu_out = 0) / fs_conv3d_tr_prelu (1) would run for this shape: the
// dispatch above with `plan` set -- nothing is launched.  Returns the number of jobs (0: the kernel reads `w` as stored).
extern "C" int fs_conv3d_tr_wprep_jobs(FsWprepJob* jobs_host, int cap, const float* x, const float* w, float* ws, int B,
                                       int Cin, int Cout, int Di, int Hi, int Wi, int Dout, int Hout, int Wout,
                                       int has_prelu_out) {
  FS_ENTER();
  if (jobs_host == nullptr || w == nullptr) return -FS_ERR_NULLPTR;
  if (cap < 0) return -FS_ERR_ARG;
  WprepPlan plan = {jobs_host, cap, 0};
  static const float one = 1.f;
  float* zdummy = has_prelu_out ? ws : nullptr;  // only tested against NULL
  const int rc = conv3d_tr_impl(x, w, nullptr, has_prelu_out ? &one : nullptr, has_prelu_out ? 1 : 0, nullptr, nullptr,
                                zdummy, ws, B, Cin, Cout, Di, Hi, Wi, Dout, Hout, Wout, nullptr, &plan);
  return rc == FS_OK ? plan.n : -rc;
}

extern "C" int fs_conv3d_tr(const float* x, const float* w, const float* bias, float* y, float* ws, int B,
                            int Cin, int Cout, int Di, int Hi, int Wi, int Dout, int Hout, int Wout,
                            fs_stream_t stream) {
  FS_ENTER();
  return conv3d_tr_impl(x, w, bias, nullptr, 0, nullptr, y, nullptr, ws, B, Cin, Cout, Di, Hi, Wi, Dout, Hout, Wout,
                        stream);
}

extern "C" int fs_conv3d_tr_add(const float* x, const float* w, const float* bias, const float* addend, float* y,
                                float* ws, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Dout, int Hout,
                                int Wout, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(addend);
  return conv3d_tr_impl(x, w, bias, nullptr, 0, addend, y, nullptr, ws, B, Cin, Cout, Di, Hi, Wi, Dout, Hout, Wout,
                        stream);
}

extern "C" int fs_conv3d_tr_prelu(const float* x, const float* w, const float* bias, const float* prelu_weight,
                                  float* y, float* z, float* ws, int B, int Cin, int Cout, int Di, int Hi, int Wi,
                                  int Dout, int Hout, int Wout, int num_prelu_weights, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(prelu_weight); FS_REQUIRE_PTR(z);
  return conv3d_tr_impl(x, w, bias, prelu_weight, num_prelu_weights, nullptr, y, z, ws, B, Cin, Cout, Di, Hi, Wi,
                        Dout, Hout, Wout, stream);
}
