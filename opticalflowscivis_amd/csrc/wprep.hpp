// wprep.hpp -- re-layout of convolution weights into the slabs the implicit-GEMM kernels stage into LDS, as JOBS:
// one element function per layout, used by the per-launch preparation (one job per kernel launch, as before) and by
// fs_conv3d_wprep_batch (all layers of a model in ONE launch per optimiser step: the 256^3 step spent ~110 launches
// of ~5 us + their dispatch gaps on these).  Include inside the anonymous namespace of the .hip file that uses it.
#pragma once

// kinds / parameters (FsWprepJob::p), see the kernels that consume each slab:
//   FS_WPREP_FWD  p = {Cout, Cin, K3, CinP, CoutP, mode}   Wt[ci][tap][co]      (convfwd.hip)
//       mode 0: W[co][ci][tap] (Conv3d forward; ConvTranspose3d input gradient)
//       mode 1: W[ci][co][K3-1-tap] (stride-1 "same" Conv3d input gradient)
//   FS_WPREP_TR32 p = {Cin, Cout, CinP, CoutT}              Wt[ci][tap 0..63][co 0..31] <- W[ci][co][tap]
//       (`w` may point at a 32-channel slice of a [Cin][CoutT][64] tensor)            (convtr.hip)
//   FS_WPREP_TR16 p = {Cin, Cout, CinP}                     Wt[ci][tap][co 0..15]
//   FS_WPREP_P8   p = {Cin, Cout, CinP, RT}                 W'[ci][neighbour][row] (+16 pad) of the all-parities kernel
//   FS_WPREP_WINO p = {Cout, Cin, CinP, mode}              Ut[ci][kz*3+ky][t 0..3][co 0..63] (+16 pad per ci): the F(2,3)
//       filter transform along kx of the taps FS_WPREP_FWD would deliver (same two modes)          (convwino.hpp)
//   FS_WPREP_WINO4 p = {Cout, Cin, CinP, mode}             Ut[ci][kz*3+ky][t 0..5][co 0..63]: the F(4,3) filter transform
//       (points 0, +-1, +-2, inf) of the same taps                                                  (convwino4.hpp)
//   FS_WPREP_WINO2D p = {Cout, Cin, CinP, mode}            Ut[ci][kz][ty 0..3][tx 0..5][co 0..63]: F(2,3) along ky and F(4,3)
//       along kx of the same taps                                                                   (convwino2d.hpp)
//   FS_WPREP_S3K4 p = {Cout, Cin, ceil(Cin / 2), CP}      words [channel group][channel pair][kz][piece][kyp][kh][co][kx]:
//       the weights of a k = 4 layer as three bf16 pieces, two channels per 4-byte word                 (convfwd_s3.hpp)
//   FS_WPREP_TRS3 p = {Cin, Cout, ceil(Cin / 4), CoutT}   words [stage of 4 channels][parity class 8][z tap 2][piece 3][y tap 2]
//       [co 0..31][x tap 2][channel pair]: the transposed layers' weights as three bf16 pieces (`w` as for FS_WPREP_TR32) (convtr_s3.hpp)
//   FS_WPREP_TRS3_16 p = {Cin, Cout <= 16, ceil(Cin / 4), CoutT}  words [stage][parity class 8][piece 3][(y tap, x tap) 4][co 0..15]
//       [z tap 2][channel pair]: the same for the 16-row form (all eight taps of a class in one 32-element reduction)
enum { FS_WPREP_FWD = 0, FS_WPREP_TR32 = 1, FS_WPREP_TR16 = 2, FS_WPREP_P8 = 3, FS_WPREP_WINO = 4, FS_WPREP_WINO4 = 5,
       FS_WPREP_WINO2D = 6, FS_WPREP_S3K4 = 7, FS_WPREP_TRS3 = 8, FS_WPREP_TRS3_16 = 9 };

// two floats -> one word of two bf16 (round to nearest even; low half = the first)
__device__ __forceinline__ unsigned s3_pack(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t h = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, h);
}

// one word of the FS_WPREP_S3K4 slab: bf16 piece `piece` (a = a0 + a1 + a2) of the weights of channels (2 cp, 2 cp + 1)
__device__ __forceinline__ unsigned wprep_s3_word(const FsWprepJob& j, int e) {
  const int Cout = j.p[0], Cin = j.p[1], CinP2 = j.p[2], CP = j.p[3];
  const float* __restrict__ w = j.w;
  const int kx = e & 3;
  int r = e >> 2;
  const int col = r % CP; r /= CP;
  const int kh = r & 1, kyp = (r >> 1) & 1; r >>= 2;
  const int piece = r % 3; r /= 3;
  const int kz = r & 3; r >>= 2;
  const int cp = r % CinP2, mg = r / CinP2;
  const int co = mg * CP + col, tap = (kz * 4 + 2 * kyp + kh) * 4 + kx;
  float v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = 2 * cp + i;
    float a = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 64 + tap] : 0.f;
    for (int q = 0; q < piece; ++q) a = a - (float)(__bf16)a;  // exact
    v[i] = a;
  }
  return s3_pack(v[0], v[1]);
}
__host__ __device__ constexpr int wprep_p8_k(int par, int d) { return par == 0 ? (d == 0 ? 1 : 3) : (d == 0 ? 0 : 2); }

// one word of the FS_WPREP_TRS3 slab.  Word index: ((((stage 8 + class) 2 + z tap) 3 + piece) 2 + y tap) 128 + co 4 + x tap 2 +
// channel pair; tap a of output parity `par` has kernel index wprep_p8_k(par, a) and input offset par - a; the two x taps
// are stored in the order of their input positions (slot 0 = tap 1)
__device__ __forceinline__ unsigned wprep_t3_word(const FsWprepJob& j, int e) {
  const int Cin = j.p[0], Cout = j.p[1], CoutT = j.p[3];
  const float* __restrict__ w = j.w;
  const int c2 = e & 1, dxa = (e >> 1) & 1, co = (e >> 2) & 31, kh = (e >> 7) & 1;
  int r = e >> 8;
  const int piece = r % 3; r /= 3;
  const int az = r & 1, cls = (r >> 1) & 7, st = r >> 4;
  const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
  const int tap = (wprep_p8_k(pz, az) * 4 + wprep_p8_k(py, kh)) * 4 + wprep_p8_k(px, 1 - dxa);  // x slot 0 = tap 1 (the lower input position)
  float v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = 4 * st + 2 * c2 + i;
    float a = (co < Cout && ci < Cin) ? w[((size_t)ci * CoutT + co) * 64 + tap] : 0.f;
    for (int q = 0; q < piece; ++q) a = a - (float)(__bf16)a;  // exact
    v[i] = a;
  }
  return s3_pack(v[0], v[1]);
}
// one word of the FS_WPREP_TRS3_16 slab.  Word index: ((((stage 8 + class) 3 + piece) 4 + (y tap 2 + x tap)) 16 + co) 4 +
// z tap 2 + channel pair
__device__ __forceinline__ unsigned wprep_t3_word16(const FsWprepJob& j, int e) {
  const int Cin = j.p[0], Cout = j.p[1], CoutT = j.p[3];
  const float* __restrict__ w = j.w;
  const int c2 = e & 1, az = (e >> 1) & 1, co = (e >> 2) & 15, kq = (e >> 6) & 3;
  int r = e >> 8;
  const int piece = r % 3; r /= 3;
  const int cls = r & 7, st = r >> 3;
  const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
  const int tap = (wprep_p8_k(pz, az) * 4 + wprep_p8_k(py, kq >> 1)) * 4 + wprep_p8_k(px, kq & 1);
  float v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = 4 * st + 2 * c2 + i;
    float a = (co < Cout && ci < Cin) ? w[((size_t)ci * CoutT + co) * 64 + tap] : 0.f;
    for (int q = 0; q < piece; ++q) a = a - (float)(__bf16)a;  // exact
    v[i] = a;
  }
  return s3_pack(v[0], v[1]);
}
constexpr int FS_WINO_UCH = 9 * 4 * 64 + 16;  // floats per input channel of the Winograd slab (== WN_UCH)
constexpr int FS_WINO4_UCH = 9 * 6 * 64;      // ... of the F(4,3) slab (== W4_UCH)
constexpr int FS_WINO2D_UCH = 3 * 24 * 64;    // ... of the F(2,3) x F(4,3) slab (== W2_UCH)

// F(4,3) filter transform G g (points 0, +-1, +-2, inf), component tt
__device__ __forceinline__ float wprep_g43(const float (&g)[3], int tt) {
  switch (tt) {
    case 0: return 0.25f * g[0];
    case 1: return (-1.f / 6.f) * ((g[0] + g[2]) + g[1]);
    case 2: return (-1.f / 6.f) * ((g[0] + g[2]) - g[1]);
    case 3: return (1.f / 24.f) * g[0] + ((1.f / 12.f) * g[1] + (1.f / 6.f) * g[2]);
    case 4: return (1.f / 24.f) * g[0] + ((-1.f / 12.f) * g[1] + (1.f / 6.f) * g[2]);
    default: return g[2];
  }
}

__device__ __forceinline__ float wprep_elem(const FsWprepJob& j, int e) {
  const float* __restrict__ w = j.w;
  switch (j.kind) {
    case FS_WPREP_FWD: {
      const int Cout = j.p[0], Cin = j.p[1], K3 = j.p[2], CoutP = j.p[4], mode = j.p[5];
      const int co = e % CoutP;
      const int t = e / CoutP;
      const int tap = t % K3, ci = t / K3;
      if (co >= Cout || ci >= Cin) return 0.f;
      return mode ? w[((size_t)ci * Cout + co) * K3 + (K3 - 1 - tap)] : w[((size_t)co * Cin + ci) * K3 + tap];
    }
    case FS_WPREP_TR32: {
      const int Cin = j.p[0], Cout = j.p[1], CoutT = j.p[3];
      const int co = e & 31, tap = (e >> 5) & 63, ci = e >> 11;
      return (co < Cout && ci < Cin) ? w[((size_t)ci * CoutT + co) * 64 + tap] : 0.f;
    }
    case FS_WPREP_TR16: {
      const int Cin = j.p[0], Cout = j.p[1];
      const int co = e & 15, tap = (e >> 4) & 63, ci = e >> 10;
      return (co < Cout && ci < Cin) ? w[((size_t)ci * Cout + co) * 64 + tap] : 0.f;
    }
    case FS_WPREP_WINO: {
      const int Cout = j.p[0], Cin = j.p[1], mode = j.p[3];
      const int ci = e / FS_WINO_UCH, i = e - ci * FS_WINO_UCH;
      if (i >= 9 * 4 * 64 || ci >= Cin) return 0.f;
      const int kk = i >> 8, tt = (i >> 6) & 3, co = i & 63;
      if (co >= Cout) return 0.f;
      float g[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        g[kx] = mode ? w[((size_t)ci * Cout + co) * 27 + (26 - (kk * 3 + kx))] : w[((size_t)co * Cin + ci) * 27 + kk * 3 + kx];
      return tt == 0 ? g[0] : tt == 1 ? 0.5f * ((g[0] + g[1]) + g[2]) : tt == 2 ? 0.5f * ((g[0] - g[1]) + g[2]) : g[2];
    }
    case FS_WPREP_WINO4: {
      const int Cout = j.p[0], Cin = j.p[1], mode = j.p[3];
      const int ci = e / FS_WINO4_UCH, i = e - ci * FS_WINO4_UCH;
      if (ci >= Cin) return 0.f;
      const int kk = i / 384, r = i - kk * 384;
      const int tt = r >> 6, co = r & 63;
      if (co >= Cout) return 0.f;
      float g[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        g[kx] = mode ? w[((size_t)ci * Cout + co) * 27 + (26 - (kk * 3 + kx))] : w[((size_t)co * Cin + ci) * 27 + kk * 3 + kx];
      return wprep_g43(g, tt);
    }
    case FS_WPREP_WINO2D: {
      const int Cout = j.p[0], Cin = j.p[1], mode = j.p[3];
      const int ci = e / FS_WINO2D_UCH, i = e - ci * FS_WINO2D_UCH;
      if (ci >= Cin) return 0.f;
      const int kz = i / 1536, r = i - kz * 1536;
      const int ty = r / 384, r2 = r - ty * 384;
      const int tx = r2 >> 6, co = r2 & 63;
      if (co >= Cout) return 0.f;
      float u[3];  // the x-transformed taps of the three ky rows
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float g[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int tap = (kz * 3 + ky) * 3 + kx;
          g[kx] = mode ? w[((size_t)ci * Cout + co) * 27 + (26 - tap)] : w[((size_t)co * Cin + ci) * 27 + tap];
        }
        u[ky] = wprep_g43(g, tx);
      }
      return ty == 0 ? u[0] : ty == 1 ? 0.5f * ((u[0] + u[1]) + u[2]) : ty == 2 ? 0.5f * ((u[0] - u[1]) + u[2]) : u[2];
    }
    default: {  // FS_WPREP_P8
      const int Cin = j.p[0], Cout = j.p[1], RT = j.p[3];
      const int wsci = 128 * RT + 16;
      const int ci = e / wsci, i = e - ci * wsci;
      if (i >= 128 * RT || ci >= Cin) return 0.f;
      const int d = i / (16 * RT), r2 = i - d * 16 * RT;
      const int rt = r2 >> 4, row = r2 & 15;
      const int px = row >> 3, item = 8 * rt + (row & 7);
      if (item >= 4 * Cout) return 0.f;
      const int pzy = item / Cout, co = item - pzy * Cout;
      const int kz = wprep_p8_k(pzy >> 1, (d >> 2) & 1), ky = wprep_p8_k(pzy & 1, (d >> 1) & 1), kx = wprep_p8_k(px, d & 1);
      return w[((size_t)ci * Cout + co) * 64 + (kz * 4 + ky) * 4 + kx];
    }
  }
}

// FS_WPREP_WINO2D by blocks (round 4): wprep_elem gathers an element's nine taps with lanes = output channels, i.e. nine
// loads whose 64 lanes touch 64 different cache lines each (the step's 32 such slabs were most of a 0.28 ms launch).  Here
// a block takes one (input channel, kz): the 64 x 9 taps are read with lanes running along the taps of a channel (a few
// lines per load), parked in LDS, and every thread forms six of the block's 24 x 64 slab entries from there -- the
// same arithmetic as wprep_elem (bit-identical slabs), coalesced stores.
__device__ __forceinline__ void wprep_wino2d_blocks(const FsWprepJob& j, int first, int stride) {
  __shared__ float g9[64 * 9 + 64];  // [co][ky*3+kx], pitch 10 (bank spread)
  const int Cout = j.p[0], Cin = j.p[1], CinP = j.p[2], mode = j.p[3];
  const float* __restrict__ w = j.w;
  const int t = threadIdx.x;
  for (int blk = first; blk < CinP * 3; blk += stride) {
    const int ci = blk / 3, kz = blk - ci * 3;
    __syncthreads();
    for (int i = t; i < 64 * 9; i += 256) {
      const int co = i / 9, k9 = i - co * 9;
      const int tap = kz * 9 + k9;
      float v = 0.f;
      if (co < Cout && ci < Cin) v = mode ? w[((size_t)ci * Cout + co) * 27 + (26 - tap)] : w[((size_t)co * Cin + ci) * 27 + tap];
      g9[co * 10 + k9] = v;
    }
    __syncthreads();
    const int co = t & 63, q = t >> 6;
    float* __restrict__ dst = j.ws + (size_t)ci * FS_WINO2D_UCH + kz * 1536 + co;
#pragma unroll
    for (int k = 0; k < 6; ++k) {      // slab entries (ty, tx) = 6 q + k of the block's 24
      const int e = 6 * q + k, ty = e / 6, tx = e - ty * 6;
      float u[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float g[3] = {g9[co * 10 + ky * 3], g9[co * 10 + ky * 3 + 1], g9[co * 10 + ky * 3 + 2]};
        u[ky] = wprep_g43(g, tx);
      }
      const float v = ty == 0 ? u[0] : ty == 1 ? 0.5f * ((u[0] + u[1]) + u[2]) : ty == 2 ? 0.5f * ((u[0] - u[1]) + u[2]) : u[2];
      dst[ty * 384 + tx * 64] = (co < Cout && ci < Cin) ? v : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void wprep_one_kernel(FsWprepJob j) {
  if (j.kind == FS_WPREP_WINO2D && j.total == j.p[2] * FS_WINO2D_UCH) {
    wprep_wino2d_blocks(j, blockIdx.x, gridDim.x);
    return;
  }
  if (j.kind == FS_WPREP_S3K4 || j.kind == FS_WPREP_TRS3 || j.kind == FS_WPREP_TRS3_16) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < j.total; e += gridDim.x * 256)
      reinterpret_cast<unsigned*>(j.ws)[e] = j.kind == FS_WPREP_S3K4 ? wprep_s3_word(j, e)
                                             : (j.kind == FS_WPREP_TRS3 ? wprep_t3_word(j, e) : wprep_t3_word16(j, e));
    return;
  }
  for (int e = blockIdx.x * 256 + threadIdx.x; e < j.total; e += gridDim.x * 256) j.ws[e] = wprep_elem(j, e);
}

// blockIdx.y = job, blockIdx.x strides over its elements
__global__ __launch_bounds__(256) void wprep_batch_kernel(const FsWprepJob* __restrict__ jobs) {
  const FsWprepJob j = jobs[blockIdx.y];
  if (j.kind == FS_WPREP_WINO2D && j.total == j.p[2] * FS_WINO2D_UCH) {
    wprep_wino2d_blocks(j, blockIdx.x, gridDim.x);
    return;
  }
  if (j.kind == FS_WPREP_S3K4 || j.kind == FS_WPREP_TRS3 || j.kind == FS_WPREP_TRS3_16) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < j.total; e += gridDim.x * 256)
      reinterpret_cast<unsigned*>(j.ws)[e] = j.kind == FS_WPREP_S3K4 ? wprep_s3_word(j, e)
                                             : (j.kind == FS_WPREP_TRS3 ? wprep_t3_word(j, e) : wprep_t3_word16(j, e));
    return;
  }
  for (int e = blockIdx.x * 256 + threadIdx.x; e < j.total; e += gridDim.x * 256) j.ws[e] = wprep_elem(j, e);
}

inline FsWprepJob wprep_job(int kind, const float* w, float* ws, long long total, int p0, int p1, int p2 = 0, int p3 = 0,
                            int p4 = 0, int p5 = 0) {
  FsWprepJob j;
  j.w = w; j.ws = ws; j.kind = kind; j.total = (int)total;
  j.p[0] = p0; j.p[1] = p1; j.p[2] = p2; j.p[3] = p3; j.p[4] = p4; j.p[5] = p5;
  return j;
}

// What a convolution entry point does with the re-layout its kernel needs:
//   plan != nullptr  -> record the job and launch nothing (fs_conv3d_*_wprep_jobs: the caller batches it)
//   w == nullptr     -> `ws` already holds the slab (prepared by fs_conv3d_wprep_batch), nothing to do
//   otherwise        -> prepare it now, one small launch in front of the convolution (the classic path)
struct WprepPlan { FsWprepJob* jobs; int cap; int n; };
inline void wprep_do(const FsWprepJob& j, WprepPlan* plan, hipStream_t st) {
  if (plan != nullptr) {
    if (plan->n < plan->cap) plan->jobs[plan->n] = j;
    plan->n++;
    return;
  }
  if (j.w == nullptr) return;
  hipLaunchKernelGGL(wprep_one_kernel, dim3((j.total + 255) / 256), dim3(256), 0, st, j);
}
