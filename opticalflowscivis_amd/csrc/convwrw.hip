// convwrw.hip -- weight gradient of the IFNet-3D convolutions as an implicit GEMM on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Not one of the §8(a) rows: the convolutions themselves stay on MIOpen.  But MIOpen (ROCm 7.2, no
// gfx950 tuning db) spends 10-80 ms per layer on this one reduction, and the stock-PyTorch
// replacement (convgrad.py: materialised im2col + split-K GEMM) still moves ~125 GB of im2col per
// 256^3 step.  The weight gradient is a GEMM with a tiny output and a huge reduction,
//
//   dW[g, c, kz,ky,kx] = sum_{b,oz,oy,ox} G[b,g,oz,oy,ox] * S[b,c, oz*s+kz-p, oy*s+ky-p, ox*s+kx-p]
//
//   M = Cg (32..128),  N = Cs * k^3,  K = B*Do*Ho*Wo (10^5 .. 10^7),
//
// so it is done here without ever forming im2col: for Conv3d  G = grad_out, S = input;  for
// ConvTranspose3d  G = input, S = grad_out  (dW then already has the [Cin, Cout, k,k,k] layout).
//
// Decomposition.  A workgroup (4 waves) owns 32*MT rows of M, one chunk of NC source channels
// (N tile = NC*k^3 columns, 216 for k=3/NC=8, 256 for k=4/NC=4) and a run of K-steps; a K-step is 32
// consecutive ox of one output row (b, oz, oy).  Per step it stages in LDS the G tile (32*MT x 32)
// and the NC*k*k source rows those 32 outputs touch ((32-1)*s + k floats each, zero padded), then
// each wave feeds 32x32x2 MFMAs for its N tiles: the B operand of column (c,kz,ky,kx) at reduction
// index ox is simply  row[c][kz][ky][ox*s + kx]  -- a per-lane constant offset plus ox*s -- so im2col
// exists only as an LDS addressing pattern.  Row pitches are padded so that both operand reads are
// bank-conflict-free (pitch = k mod 32 for the source rows, 33 for G).  Accumulators stay in registers
// over the whole run; the epilogue adds the partial tile to dW with float atomics (128 contiguous
// bytes per half-wave = the full-rate shape; ~100 MB of atomic traffic per layer).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KW = 32;  // reduction elements (consecutive ox) per K-step

struct WP {
  int B, Cg, Cs;
  int Do, Ho, Wo;   // extent of G (the "output grid" of the reduction)
  int Di, Hi, Wi;   // extent of S
  int pad;
  int segs;         // ceil(Wo / KW)
  long long steps;  // B*Do*Ho*segs
  int spw;          // K-steps per workgroup
};

template <int K, int S, int NC, int MT>
__global__ __launch_bounds__(256, 2) void conv3d_wrw_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ Src,
                                                         float* __restrict__ dW, WP p) {
  constexpr int K3 = K * K * K;
  constexpr int NTOT = NC * K3;               // live columns of this N chunk
  constexpr int NT32 = (NTOT + 31) / 32;      // 32-column MFMA tiles
  constexpr int NPW = (NT32 + 3) / 4;         // N tiles per wave
  constexpr int RL = (KW - 1) * S + K;        // source row piece needed by 32 outputs
  constexpr int RLP = RL + ((K - RL % 32) % 32 + 32) % 32;  // padded so that RLP % 32 == K
  constexpr int ROWS = NC * K * K;
  constexpr int GLD = KW + 1;
  static_assert(RLP % 32 == K % 32, "source row pitch");
  // k = 3: double-buffered tiles, one barrier per K-step, prefetch interleaved with the MFMAs.
  // k = 4 (25 prefetch registers per thread): single buffer, two barriers, prefetch issued up
  // front -- the leaner structure keeps it under 256 VGPRs without spills.
  constexpr bool DB = (K == 3);
  constexpr bool INTER = (K == 3);
  __shared__ float sG[DB ? 2 : 1][32 * MT][GLD];
  __shared__ float sS[DB ? 2 : 1][ROWS][RLP];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int c0 = blockIdx.y * NC;            // first source channel of this chunk
  const int g0 = blockIdx.z * 32 * MT;       // first G channel of this M tile
  const size_t gvol = (size_t)p.Do * p.Ho * p.Wo, svol = (size_t)p.Di * p.Hi * p.Wi;

  // per-lane constants of the B operand: column j -> offset of (c, kz, ky, kx) in sS
  int boff[NPW];
#pragma unroll
  for (int n = 0; n < NPW; ++n) {
    const int j = (wv + 4 * n) * 32 + (lane & 31);
    int off = 0;
    if (j < NTOT) {
      const int c = j / K3, r = j - c * K3;
      const int kz = r / (K * K), ky = (r / K) % K, kx = r % K;
      off = ((c * K + kz) * K + ky) * RLP + kx;
    }
    boff[n] = off;
  }
  const int kh = lane >> 5;  // which of the 2 reduction elements of an MFMA this lane feeds

  f32x16 acc[MT][NPW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NPW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  constexpr int ITG = (32 * MT * KW + 255) / 256;  // G-tile elements per thread and step
  constexpr int ITS = (ROWS * RL + 255) / 256;     // source-row elements per thread and step
  float rG[ITG], rS[ITS];

  // position of the step being FETCHED, advanced incrementally (seg fastest, then oy, oz, b)
  const long long s0 = (long long)blockIdx.x * p.spw;
  const long long s1 = min(s0 + p.spw, p.steps);
  int f_seg, f_oy, f_oz, f_b;
  {
    long long q = s0;
    f_seg = (int)(q % p.segs); q /= p.segs;
    f_oy = (int)(q % p.Ho); q /= p.Ho;
    f_oz = (int)(q % p.Do);
    f_b = (int)(q / p.Do);
  }
  auto advance = [&]() {
    if (++f_seg == p.segs) { f_seg = 0; if (++f_oy == p.Ho) { f_oy = 0; if (++f_oz == p.Do) { f_oz = 0; ++f_b; } } }
  };

  // global -> registers for a quarter of one K-step (zero where the tile leaves G / the padded
  // source).  Called four times per step, spread over the MFMA loop, so that the address
  // arithmetic issues in the shadow of the 64-cycle matrix instructions.
  auto fetch = [&](int part) {
    const int ox0 = f_seg * KW;
    // wave-uniform bases + 32-bit byte offsets: one address VGPR per load (saddr form)
    const char* gb = reinterpret_cast<const char*>(G + ((size_t)f_b * p.Cg + g0) * gvol +
                                                   ((size_t)f_oz * p.Ho + f_oy) * p.Wo + ox0);
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      if (part >= 0 && (it & 3) != part) continue;
      const int i = t + 256 * it;
      const int r = i / KW, col = i - r * KW;
      float v = 0.f;
      if (i < 32 * MT * KW && g0 + r < p.Cg && ox0 + col < p.Wo)
        v = *reinterpret_cast<const float*>(gb + ((unsigned)r * (unsigned)gvol + (unsigned)col) * 4u);
      rG[it] = v;
    }
    const int ix0 = ox0 * S - p.pad;
    const char* sb = reinterpret_cast<const char*>(Src + ((size_t)f_b * p.Cs + c0) * svol);
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      if (part >= 0 && (it & 3) != part) continue;
      const int i = t + 256 * it;
      const int r = i / RL, col = i - r * RL;
      const int c = r / (K * K), kz = (r / K) % K, ky = r % K;
      const int iz = f_oz * S + kz - p.pad, iy = f_oy * S + ky - p.pad, ix = ix0 + col;
      float v = 0.f;
      if (i < ROWS * RL && c0 + c < p.Cs && iz >= 0 && iz < p.Di && iy >= 0 && iy < p.Hi && ix >= 0 &&
          ix < p.Wi)
        v = *reinterpret_cast<const float*>(
            sb + ((unsigned)c * (unsigned)svol + ((unsigned)iz * p.Hi + iy) * p.Wi + ix) * 4u);
      rS[it] = v;
    }
  };
  auto park = [&](int buf) {  // registers -> LDS buffer `buf`
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int i = t + 256 * it;
      if (i < 32 * MT * KW) sG[buf][i / KW][i % KW] = rG[it];
    }
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int i = t + 256 * it;
      if (i < ROWS * RL) sS[buf][i / RL][i % RL] = rS[it];
    }
  };

  if (s0 < s1) {
    fetch(-1);  // whole step
    advance();
  }
  for (long long st = s0; st < s1; ++st) {
    const int buf = DB ? (int)((st - s0) & 1) : 0;
    park(buf);
    // double-buffered: ONE barrier per step (the other buffer was last read in the previous step's
    // MFMA phase, which every wave has left before it can arrive here)
    __syncthreads();
    const bool more = (st + 1 < s1);
    const float* sSf = &sS[buf][0][0];
    if (!INTER && more) fetch(-1);
    // operands of reduction pair kk: one A float per M tile, one B float per N tile of this wave.
    // They are read from LDS ONE PAIR AHEAD of the MFMAs that use them (register rotation): issued
    // right before use, every pair would expose an LDS round trip (~100 cycles per 256 MFMA cycles).
    auto lds_ops = [&](int kk, float (&a)[MT], float (&bq)[NPW]) {
      const int ox = 2 * kk + kh;
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = sG[buf][m * 32 + (lane & 31)][ox];
#pragma unroll
      for (int n = 0; n < NPW; ++n) bq[n] = sSf[boff[n] + ox * S];
    };
    auto mma = [&](const float (&a)[MT], const float (&bq)[NPW]) {
#pragma unroll
      for (int n = 0; n < NPW; ++n) {
        if ((wv + 4 * n) < NT32) {  // wave-uniform
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bq[n], acc[m][n], 0, 0, 0);
        }
      }
    };
    float a0[MT], b0[NPW], a1[MT], b1[NPW];
    lds_ops(0, a0, b0);
#pragma unroll
    for (int part = 0; part < 4; ++part) {
      if (INTER && more) fetch(part);  // next step's quarter: in flight under the matrix instructions
#pragma unroll
      for (int k4 = 0; k4 < KW / 8; k4 += 2) {
        const int kk = part * (KW / 8) + k4;
        lds_ops(kk + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of the MFMAs they do not feed
        mma(a0, b0);
        if (kk + 2 < KW / 2) lds_ops(kk + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
      }
    }
    if (more) advance();
    if (!DB) __syncthreads();  // single buffer: MFMA reads done before the next park
  }

  // ---- epilogue: dW[g, c0*K3 + j] += acc  (row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31)
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
      const int nt = wv + 4 * n;
      if (nt >= NT32) continue;
      const int j = nt * 32 + (lane & 31);
      if (j >= NTOT || c0 * K3 + j >= p.Cs * K3) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int g = g0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (g < p.Cg) atomicAdd(dW + (size_t)g * p.Cs * K3 + (size_t)c0 * K3 + j, acc[m][n][r]);
      }
    }
}

template <int K, int S, int NC>
int launch(const float* G, const float* Src, float* dW, WP& p, hipStream_t st) {
  const int mt = (p.Cg > 32) ? 2 : 1;
  const int mtiles = fs::cdiv(p.Cg, 32 * mt);
  const int nchunks = fs::cdiv(p.Cs, NC);
  // enough workgroups to fill the chip a few times, runs long enough to amortise the epilogue
  long long want = 4096 / ((long long)mtiles * nchunks);
  if (want < 1) want = 1;
  long long spw = (p.steps + want - 1) / want;
  if (spw < 8) spw = 8;
  p.spw = (int)(spw > (1 << 20) ? (1 << 20) : spw);
  const long long gx = (p.steps + p.spw - 1) / p.spw;
  if (gx >= (1ll << 31) || nchunks > 65535 || mtiles > 65535) return FS_ERR_SHAPE;
  dim3 grid((unsigned)gx, nchunks, mtiles);
  if (mt == 2)
    hipLaunchKernelGGL((conv3d_wrw_kernel<K, S, NC, 2>), grid, dim3(256), 0, st, G, Src, dW, p);
  else
    hipLaunchKernelGGL((conv3d_wrw_kernel<K, S, NC, 1>), grid, dim3(256), 0, st, G, Src, dW, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // namespace

extern "C" int fs_conv3d_wrw(const float* g, const float* src, float* dw, int B, int Cg, int Cs, int Do,
                             int Ho, int Wo, int Di, int Hi, int Wi, int kernel, int stride, int pad,
                             fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(g); FS_REQUIRE_PTR(src); FS_REQUIRE_PTR(dw);
  if (B < 1 || Cg < 1 || Cs < 1 || Do < 1 || Ho < 1 || Wo < 1 || Di < 1 || Hi < 1 || Wi < 1)
    return FS_ERR_SHAPE;
  if (!((kernel == 3 && stride == 1) || (kernel == 4 && stride == 2)) || pad < 0 || pad >= kernel)
    return FS_ERR_ARG;
  // every output position must read inside the padded source: (Do-1)*s + k - 1 - pad <= Di - 1 + pad
  if ((Do - 1) * stride + kernel > Di + 2 * pad || (Ho - 1) * stride + kernel > Hi + 2 * pad ||
      (Wo - 1) * stride + kernel > Wi + 2 * pad)
    return FS_ERR_SHAPE;
  // 32-bit byte offsets inside one (b, chunk) slab of G / src
  if ((long long)64 * Do * Ho * Wo * 4 >= (1ll << 32) || (long long)8 * Di * Hi * Wi * 4 >= (1ll << 32))
    return FS_ERR_SHAPE;
  WP p;
  p.B = B; p.Cg = Cg; p.Cs = Cs; p.Do = Do; p.Ho = Ho; p.Wo = Wo; p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  p.pad = pad;
  p.segs = fs::cdiv(Wo, KW);
  p.steps = (long long)B * Do * Ho * p.segs;
  hipStream_t st = (hipStream_t)stream;
  if (kernel == 3) return launch<3, 1, 8>(g, src, dw, p, st);
  // k = 4: 64 columns per source channel.  NC = 4 gives every wave two 32-column tiles, NC = 2 one;
  // pick the chunking with less padded matrix work (Cs = 1, 2, 5, 6: the IFNet heads / block0 input)
  const int cost2 = (Cs + 1) / 2, cost4 = 2 * ((Cs + 3) / 4);
  if (cost2 < cost4) return launch<4, 2, 2>(g, src, dw, p, st);
  return launch<4, 2, 4>(g, src, dw, p, st);
}
