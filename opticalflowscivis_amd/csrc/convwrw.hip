// convwrw.hip -- weight gradient of the IFNet-3D convolutions as an implicit GEMM on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Not one of the §8(a) rows (companions: convfwd.hip, convtr.hip).  MIOpen (ROCm 7.2, no gfx950 tuning
// db) spends 10-80 ms per layer on this one reduction, and a stock-PyTorch replacement (materialised im2col
// + split-K GEMM, round 1's first attempt, since removed) still moved ~125 GB of im2col per 256^3 step.
// The weight gradient is a GEMM with a tiny output and a huge reduction,
//
//   dW[g, c, kz,ky,kx] = sum_{b,oz,oy,ox} G[b,g,oz,oy,ox] * S[b,c, oz*s+kz-p, oy*s+ky-p, ox*s+kx-p]
//
//   M = Cg (32..128),  N = Cs * k^3,  K = B*Do*Ho*Wo (10^5 .. 10^7),
//
// so it is done here without ever forming im2col: for Conv3d  G = grad_out, S = input;  for
// ConvTranspose3d  G = input, S = grad_out  (dW then already has the [Cin, Cout, k,k,k] layout).
//
// Decomposition.  A workgroup (4 waves) owns 32*MT rows of M, one chunk of NC source channels
// (N tile = NC*k^3 columns, 216 for k=3/NC=8, 256 for k=4/NC=4) and a run of K-steps; a K-step is a
// TZ x TY x 32 brick of output positions (4 rows of 32 ox for k=3, 2 rows for k=4).  Per step it stages
// in LDS the G brick (32*MT channels x rows x 32) and the source brick with its halo (zero padded),
// then each wave feeds 32x32x2 MFMAs for its N tiles: the B operand of column (c,kz,ky,kx) at
// reduction index (tz,ty,ox) is  sS[c][tz*s+kz][ty*s+ky][ox*s+kx]  -- a per-lane constant offset plus
// a compile-time one -- so im2col exists only as an LDS addressing pattern.  Pitches are padded so that
// both operand reads are bank-conflict-free (column offset == column index mod 32 for the source brick,
// odd row pitch for G).  The next brick is prefetched into registers under the MFMA phase; operands
// are read from LDS one reduction pair ahead of the MFMAs that use them.  Accumulators stay in
// registers over the whole run; the epilogue adds the partial tile to dW with float atomics (128
// contiguous bytes per half-wave = the full-rate shape).
#include <cstdint>
#include <cstdlib>
#include <type_traits>

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
  int nsrc = 0;                  // multi-source src (WB::src), loader-wave kernel only
  const float* const* srcv = nullptr;
  const long long* sbsv = nullptr;
};

// (round-1 history: a K-step used to be ONE row of 32 ox -- 9.6x source re-read for k=3, a barrier per
// 64 MFMAs, 79 TFLOP/s on the 64-channel layers; the brick form shares the halo between rows (4.8x),
// runs 256 MFMAs between barriers and reaches 98.)
struct WB {
  int B, Cg, Cs;
  int Do, Ho, Wo, Di, Hi, Wi;
  int pad;
  int bz, by, bx;    // bricks per axis
  long long bricks;  // B*bz*by*bx
  int spw;           // bricks per workgroup
  // optional multi-source `src` (loader-wave kernel): its Cs = nsrc <= 12 channels are planes of different
  // tensors, src[c] = channel c of sample 0, sbs[c] = that tensor's batch stride in floats (the never-materialised
  // torch.cat of IFBlock's input, see csrc/convfwd.hip FP::src).  nsrc = 0: one tensor.
  int nsrc;
  const float* src[12];
  long long sbs[12];
  // deterministic mode (fs_conv3d_wrw_det): slab != 0 = floats per private copy of dW; run blockIdx.x STORES its partial
  // tile into copy blockIdx.x of the workspace passed as `dW` (no atomics), wrw_reduce_kernel sums the copies in run order
  long long slab;
};

// Deterministic weight gradients: every kernel below splits the positions into runs (blockIdx.x) whose partial tiles
// overlap in dW.  Default: float atomics into a zero-filled dW (order of the adds = order of arrival: last-bit noise from
// run to run).  With a workspace of (runs x |dW|) floats each run writes its own copy -- every (blockIdx.y, blockIdx.z) of a
// run covers a disjoint part of dW, together all of it -- and one more launch adds the copies in run order.
struct WDet {
  float* ws;         // workspace, or nullptr with `need` set
  long long cap;     // floats available in ws
  long long* need;   // != nullptr: write the floats the dispatch would need and launch nothing
};

__global__ __launch_bounds__(256) void wrw_reduce_kernel(const float* __restrict__ ws, int runs, long long n,
                                                         float* __restrict__ dW) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float a = ws[i];
    for (int r = 1; r < runs; ++r) a += ws[(size_t)r * n + i];
    dW[i] = a;
  }
}

// host side of the deterministic mode, shared by the three launchers: returns < 0 to go on with the launch (the pointer
// the kernel writes through is *out, p_slab its slab stride), or a status to return
static inline int wrw_det_begin(const WDet* det, long long runs, long long dwf, float* dW, float** out, long long* p_slab) {
  *out = dW; *p_slab = 0;
  if (det == nullptr) return -1;
  if (det->need != nullptr) { *det->need = runs * dwf; return FS_OK; }
  if (det->ws == nullptr || runs * dwf > det->cap) return FS_ERR_ARG;
  *out = det->ws; *p_slab = dwf;
  return -1;
}

static inline void wrw_det_end(const WDet* det, long long runs, long long dwf, float* dW, hipStream_t st) {
  if (det == nullptr || det->need != nullptr) return;
  const long long blocks = (dwf + 255) / 256;
  hipLaunchKernelGGL(wrw_reduce_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, det->ws, (int)runs,
                     dwf, dW);
}

constexpr int pad_to(int n, int want) { return n + (((want - n) % 32) + 32) % 32; }

template <int K, int S, int NC, int MT, int TZ, int TY>
__global__ __launch_bounds__(256, 2) void conv3d_wrw_brick_kernel(const float* __restrict__ G,
                                                               const float* __restrict__ Src,
                                                               float* __restrict__ dW, WB p) {
  constexpr int K3 = K * K * K;
  constexpr int NTOT = NC * K3;
  constexpr int NT32 = (NTOT + 31) / 32;
  constexpr int NPW = (NT32 + 3) / 4;
  constexpr int ROWS = TZ * TY;
  constexpr int ZT = (TZ - 1) * S + K, YT = (TY - 1) * S + K, XT = (KW - 1) * S + K;
  constexpr int XP = pad_to(XT, K);                  // == K      (mod 32)
  constexpr int PSP = pad_to(YT * XP, K * K);        // == K^2    (mod 32)
  constexpr int CHSP = pad_to(ZT * PSP, K * K * K);  // == K^3    (mod 32)
  constexpr int GP = ROWS * KW + 1;
  constexpr int NG = 32 * MT * ROWS * KW;            // G brick elements
  constexpr int NS = NC * ZT * YT * XT;              // source brick elements (unpadded count)
  __shared__ float sG[32 * MT * GP];
  __shared__ float sS[NC * CHSP];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int c0 = blockIdx.y * NC;
  const int g0 = blockIdx.z * 32 * MT;
  const size_t gvol = (size_t)p.Do * p.Ho * p.Wo, svol = (size_t)p.Di * p.Hi * p.Wi;

  int boff[NPW];
#pragma unroll
  for (int n = 0; n < NPW; ++n) {
    const int j = (wv + 4 * n) * 32 + (lane & 31);
    int off = 0;
    if (j < NTOT) {
      const int c = j / K3, r = j - c * K3;
      const int kz = r / (K * K), ky = (r / K) % K, kx = r % K;
      off = c * CHSP + kz * PSP + ky * XP + kx;
    }
    boff[n] = off;
  }
  const int kh = lane >> 5;

  f32x16 acc[MT][NPW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NPW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  constexpr int ITG = NG / 256;
  constexpr int ITS = (NS + 255) / 256;
  // brick-invariant part of the source staging: the element's (z, y, x, c) inside the brick, packed
  unsigned szyx[ITS];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int i = t + 256 * it;
    const int c = i / (ZT * YT * XT), r1 = i - c * (ZT * YT * XT);
    const int z = r1 / (YT * XT), r2 = r1 - z * (YT * XT);
    const int y = r2 / XT, x = r2 - y * XT;
    const bool ok = i < NS && c0 + c < p.Cs;
    szyx[it] = ok ? ((unsigned)z | ((unsigned)y << 8) | ((unsigned)x << 16) | ((unsigned)c << 24)) : ~0u;
  }
  float rG[ITG], rS[ITS];
  static_assert(256 % (ROWS * KW) == 0, "G brick rows per thread");
  constexpr int RPI = 256 / (ROWS * KW);  // G channels covered by one pass of the 256 threads
  const int gr_t = t / (ROWS * KW), g_row = (t / KW) % ROWS, g_col = t % KW;

  const long long s0 = (long long)blockIdx.x * p.spw;
  const long long s1 = min(s0 + p.spw, p.bricks);
  auto fetch = [&](long long q) {
    const int bxi = (int)(q % p.bx); q /= p.bx;
    const int byi = (int)(q % p.by); q /= p.by;
    const int bzi = (int)(q % p.bz);
    const int b = (int)(q / p.bz);
    const int oz0 = bzi * TZ, oy0 = byi * TY, ox0 = bxi * KW;
    // G brick: element t + 256*it is (channel gr_t + RPI*it, row g_row, column g_col): the position part
    // is the same for all of a thread's elements.  (k = 3 keeps the per-element form: measured 12 %
    // faster there -- the compiler overlaps its longer address chains with the previous MFMA phase --
    // while the shared form wins 4 % for k = 4.)
    const char* gb = reinterpret_cast<const char*>(G + ((size_t)b * p.Cg + g0) * gvol);
    if (K == 3) {
#pragma unroll
      for (int it = 0; it < ITG; ++it) {
        const int i = t + 256 * it;
        const int r = i / (ROWS * KW), row = (i / KW) % ROWS, col = i % KW;
        const int oz = oz0 + row / TY, oy = oy0 + row % TY, ox = ox0 + col;
        float v = 0.f;
        if (g0 + r < p.Cg && oz < p.Do && oy < p.Ho && ox < p.Wo)
          v = *reinterpret_cast<const float*>(
              gb + ((unsigned)r * (unsigned)gvol + ((unsigned)oz * p.Ho + oy) * p.Wo + ox) * 4u);
        rG[it] = v;
      }
    } else {
      const int oz = oz0 + g_row / TY, oy = oy0 + g_row % TY, ox = ox0 + g_col;
      const bool pos_ok = oz < p.Do && oy < p.Ho && ox < p.Wo;
      const unsigned pos_off = ((unsigned)oz * p.Ho + oy) * p.Wo + ox;
#pragma unroll
      for (int it = 0; it < ITG; ++it) {
        const int r = gr_t + RPI * it;
        float v = 0.f;
        if (pos_ok && g0 + r < p.Cg)
          v = *reinterpret_cast<const float*>(gb + ((unsigned)r * (unsigned)gvol + pos_off) * 4u);
        rG[it] = v;
      }
    }
    const int gz0 = oz0 * S - p.pad, gy0 = oy0 * S - p.pad, gx0 = ox0 * S - p.pad;
    const char* sb = reinterpret_cast<const char*>(Src + ((size_t)b * p.Cs + c0) * svol);
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const unsigned zyx = szyx[it];
      const int gz = gz0 + (int)(zyx & 255u), gy = gy0 + (int)((zyx >> 8) & 255u), gx = gx0 + (int)((zyx >> 16) & 255u);
      float v = 0.f;
      if (zyx != ~0u && (unsigned)gz < (unsigned)p.Di && (unsigned)gy < (unsigned)p.Hi &&
          (unsigned)gx < (unsigned)p.Wi)
        v = *reinterpret_cast<const float*>(
            sb + ((zyx >> 24) * (unsigned)svol + ((unsigned)gz * p.Hi + gy) * p.Wi + gx) * 4u);
      rS[it] = v;
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      if (K == 3) {
        const int i = t + 256 * it;
        sG[(i / (ROWS * KW)) * GP + (i % (ROWS * KW))] = rG[it];
      } else {
        sG[(gr_t + RPI * it) * GP + g_row * KW + g_col] = rG[it];
      }
    }
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int i = t + 256 * it;
      if (i < NS) {
        const int c = i / (ZT * YT * XT), r1 = i - c * (ZT * YT * XT);
        const int z = r1 / (YT * XT), r2 = r1 - z * (YT * XT);
        const int y = r2 / XT, x = r2 - y * XT;
        sS[c * CHSP + z * PSP + y * XP + x] = rS[it];
      }
    }
  };

  if (s0 < s1) fetch(s0);
  for (long long st = s0; st < s1; ++st) {
    park();
    __syncthreads();
    if (st + 1 < s1) fetch(st + 1);
    // reduction pair kk of row `row`: positions ox = 2 kk + kh
    auto lds_ops = [&](int q, float (&a)[MT], float (&bq)[NPW]) {
      const int row = q / (KW / 2), kk = q % (KW / 2);
      const int ox = 2 * kk + kh;
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = sG[(m * 32 + (lane & 31)) * GP + row * KW + ox];
#pragma unroll
      for (int n = 0; n < NPW; ++n)
        bq[n] = sS[boff[n] + (row / TY) * S * PSP + (row % TY) * S * XP + ox * S];
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
    constexpr int NQ = ROWS * (KW / 2);
    float a0[MT], b0[NPW], a1[MT], b1[NPW];
    lds_ops(0, a0, b0);
#pragma unroll
    for (int q = 0; q < NQ; q += 2) {
      lds_ops(q + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      if (q + 2 < NQ) lds_ops(q + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
    }
    __syncthreads();
  }

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
        if (g < p.Cg) {
          float* q = dW + (size_t)blockIdx.x * p.slab + (size_t)g * p.Cs * K3 + (size_t)c0 * K3 + j;
          if (p.slab) *q = acc[m][n][r]; else atomicAdd(q, acc[m][n][r]);
        }
      }
    }
}

template <int K, int S, int NC, int TZ, int TY>
int launch_brick(const float* G, const float* Src, float* dW, const WP& w, hipStream_t st, const WDet* det = nullptr) {
  WB p;
  p.B = w.B; p.Cg = w.Cg; p.Cs = w.Cs; p.Do = w.Do; p.Ho = w.Ho; p.Wo = w.Wo;
  p.Di = w.Di; p.Hi = w.Hi; p.Wi = w.Wi; p.pad = w.pad;
  p.bz = fs::cdiv(p.Do, TZ); p.by = fs::cdiv(p.Ho, TY); p.bx = fs::cdiv(p.Wo, KW);
  p.bricks = (long long)p.B * p.bz * p.by * p.bx;
  const int mt = (p.Cg > 32) ? 2 : 1;
  const int mtiles = fs::cdiv(p.Cg, 32 * mt);
  const int nchunks = fs::cdiv(p.Cs, NC);
  // ~1024 workgroups (two per workgroup slot of the chip): fewer, longer runs keep the atomic epilogue
  // (one 32*MT x NC*k^3 tile per workgroup) small; measured best among 512..8192 on the 256^3 layers
  long long want = 1024 / ((long long)mtiles * nchunks);
  if (want < 1) want = 1;
  long long spw = (p.bricks + want - 1) / want;
  if (spw < 2) spw = 2;
  p.spw = (int)(spw > (1 << 20) ? (1 << 20) : spw);
  const long long gx = (p.bricks + p.spw - 1) / p.spw;
  if (gx >= (1ll << 31) || nchunks > 65535 || mtiles > 65535) return FS_ERR_SHAPE;
  dim3 grid((unsigned)gx, nchunks, mtiles);
  constexpr int K3 = K * K * K;
  float* out;
  const long long dwf = (long long)p.Cg * p.Cs * K3;
  const int drc = wrw_det_begin(det, gx, dwf, dW, &out, &p.slab);
  if (drc >= 0) return drc;
  if (mt == 2)
    hipLaunchKernelGGL((conv3d_wrw_brick_kernel<K, S, NC, 2, TZ, TY>), grid, dim3(256), 0, st, G, Src, out, p);
  else
    hipLaunchKernelGGL((conv3d_wrw_brick_kernel<K, S, NC, 1, TZ, TY>), grid, dim3(256), 0, st, G, Src, out, p);
  wrw_det_end(det, gx, dwf, dW, st);
  FS_LAUNCH_CHECK();
  return FS_OK;
}


// ---- DMA-staged brick kernel: one 8-wave workgroup per CU, loader waves + matrix waves, two LDS buffers -------
// Same implicit GEMM, restructured around `buffer_load_dwordx4 ... lds` (global -> LDS with no VGPR stop and no
// ds_write pass).  Waves 4-7 (one per SIMD) only stage: they decode the next brick, build two buffer descriptors
// and issue ~20 LDS-DMA pieces each into the second LDS buffer; waves 0-3 (their SIMD partners) only read operands
// and issue MFMAs on the first; one barrier per K-step.  What that buys over the register-staged kernel above:
//   * the matrix waves' instruction stream is operand reads + MFMAs only (28 350 cycles per 448-MFMA brick is
//     back-to-back issue; measured 29 600 with the loaders running beside them);
//   * no staging registers (52-70 per thread above), so a matrix wave holds 7-8 accumulator tiles: the
//     32*MT x NC*k^3 tile of a workgroup is dealt to its four matrix waves in equal shares -- FULL whole
//     column tiles (all MT row tiles) plus, with HALF, one row tile of a column tile shared by a wave pair; for
//     k = 3 / NC = 16 that is 13.5 column tiles -> 14 x 2 = 28 MFMA tiles, 7 per wave (27/28 useful; the
//     kernel above runs 4,4,4,2 tiles on 216 of 256 columns: 84 %);
//   * NC = 16 source channels per chunk for k = 3 (8 above): every G brick is read by half as many workgroups;
//   * ~256 workgroups instead of ~1024: a quarter of the atomic epilogue traffic.
// Measured on the way (s_memtime stamps, -DFS_WRW_STAMPS, scripts/wrw_stamps.py; 64-channel k3 layer at 64^3):
// an LDS-DMA wave-instruction costs the issuing wave ~90-130 cycles whatever its width -- 76 dword pieces per
// wave and brick took 6 800 cycles in front of the MFMA phase (1.02 ms per launch) and 10 000 spread between the
// MFMAs (1.10 ms); 21 sixteen-byte pieces 2 000 cycles (0.90 ms); moved to partner waves 0.85 ms = 137 TFLOP/s
// (the register-staged kernel: 1.17 ms).  Hence 16-byte pieces: every row of the source brick starts 4 floats
// left of the first output column (16-byte aligned in memory when W % 4 == 0) and all pitches are multiples of
// 4 floats.  An LDS-DMA writes 64 consecutive 16-byte slots; slots that are padding, halo outside the volume
// or channels past Cs carry an out-of-range offset and receive 0 (measured on gfx950).  A piece can only be
// outside the volume in the halo of a brick that touches a border (pad <= stride, W % 4 == 0): a 6-bit border
// class per piece, tested against the brick's border mask, is the whole per-brick address arithmetic.  The
// pitches are no longer conflict-free (multiples of 4): the 7 operand reads per 7 MFMAs of a wave leave the
// LDS far from saturated.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned DMA_OOB = 0x80000000u;
#ifdef FS_WRW_STAMPS
__device__ unsigned long long fs_wrw_dbg[4 * 8];
#define STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dt[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

constexpr int up4(int n) { return (n + 3) / 4 * 4; }

// KWX = x extent of a brick (32, or 16 for layers with 16 output columns: a 32-element reduction row is then two
// consecutive y rows of 16, which are contiguous in G when Wo == 16); TY counts 32-element rows per z slice
template <int K, int S, int NC, int MT, int TZ, int TY, int FULL, int HALF, int KWX = 32>
__global__ __launch_bounds__(512, 2) void conv3d_wrw_dma_kernel(const float* __restrict__ G,
                                                             const float* __restrict__ Src,
                                                             float* __restrict__ dW, WB p) {
  constexpr int K3 = K * K * K;
  constexpr int NTOT = NC * K3;
  constexpr int NT32 = 4 * FULL + 2 * HALF;
  static_assert(NT32 * 32 >= NTOT && (NT32 - 1) * 32 < NTOT + 32, "column tiles cover the chunk");
  static_assert(HALF == 0 || MT == 2, "a shared column tile is split by row tile");
  constexpr int NB = FULL + HALF;  // B operands per reduction pair
  constexpr int ROWS = TZ * TY;
  static_assert(KWX == 32 || KWX == 16, "brick x extent");
  constexpr int YR = KW / KWX;                           // y rows per 32-element reduction row
  constexpr int TYB = TY * YR;                           // y rows of the brick
  constexpr int ZT = (TZ - 1) * S + K, YT = (TYB - 1) * S + K;
  constexpr int XL = 4;                                  // floats between the row start and output column 0's tap 0 + pad
  constexpr int XP = up4(XL + (KWX - 1) * S + K);        // row pitch = staged row length (the pad shifts taps, not rows)
  constexpr int PSP = YT * XP, CHSP = ZT * PSP;
  constexpr int GP = ROWS * KW + 4;
  constexpr int NGF = 32 * MT * GP;                      // floats of the G image
  constexpr int NGL = (NGF + 255) / 256 * 256;
  constexpr int NSF = NC * CHSP;                         // floats of the source image
  constexpr int NSL = (NSF + 255) / 256 * 256;
  constexpr int NGW = (NGL / 256 + 3) / 4;               // G pieces (1-KiB wave-instructions) per loader wave
  constexpr int NSW = (NSL / 256 + 3) / 4;               // source pieces per loader wave
  constexpr int BUF = NGL + NSL;
  static_assert(2 * BUF * 4 <= 160 * 1024, "two buffers fit the CU's LDS");
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;           // waves 0-3: matrix waves, one per SIMD; waves 4-7: their loader partners
  const bool loader = wave >= 4;
  const int l31 = lane & 31, kh = lane >> 5;
  const int c0 = blockIdx.y * NC;
  const int g0 = blockIdx.z * 32 * MT;
  const size_t gvol = (size_t)p.Do * p.Ho * p.Wo, svol = (size_t)p.Di * p.Hi * p.Wi;
  const long long s0 = (long long)blockIdx.x * p.spw;
  const long long s1 = min(s0 + p.spw, p.bricks);

  if (loader) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass has neither the buffer-resource type nor the LDS-DMA builtin)
    __builtin_amdgcn_s_setprio(3);  // (a loader wave issues a handful of instructions per period: they should not queue behind the matrix wave's)
    // ---- brick-invariant part of the staging.  Piece k of loader wave wv fills the 16-byte slots
    // 256 (wv + 4 k) + 4 lane .. + 3 of an image.
    // source image: byte offset from the brick origin and border class (bit 0/1: low / high z halo, 2/3: y,
    // 4/5: x); pad slots and channels past Cs are permanently out of range
    unsigned soff[NSW], scls[NSW];
    int sch[NSW];  // (multi-source) the chunk channel a piece belongs to
    const int zhi = p.Di + p.pad - (p.bz - 1) * TZ * S, yhi = p.Hi + p.pad - (p.by - 1) * TYB * S,
              xhi = p.Wi + XL - (p.bx - 1) * KWX * S;  // first invalid brick coordinate in the LAST brick of an axis
#pragma unroll
    for (int k = 0; k < NSW; ++k) {
      const int q = 256 * (wv + 4 * k) + 4 * lane;
      const int c = q / CHSP, r1 = q - c * CHSP;
      const int z = r1 / PSP, r2 = r1 - z * PSP;
      const int y = r2 / XP, x = r2 - y * XP;
      const bool ok = q < NSF && c0 + c < p.Cs;
      // (multi-source: every channel has its own descriptor, the offset stays inside the plane)
      soff[k] = ok ? ((p.nsrc ? 0u : (unsigned)c * (unsigned)svol) + ((unsigned)z * p.Hi + (unsigned)y) * p.Wi + (unsigned)x) * 4u
                   : DMA_OOB;
      sch[k] = q < NSF ? c : NC - 1;
      scls[k] = (z < p.pad ? 1u : 0u) | (z >= zhi ? 2u : 0u) | (y < p.pad ? 4u : 0u) | (y >= yhi ? 8u : 0u) |
                (x < XL ? 16u : 0u) | (x >= xhi ? 32u : 0u);
    }
    // G image [channel][GP]: byte offset of the piece from the brick's first position, packed (rz, ry, col) for
    // the bounds test of bricks that stick out of the output grid
    unsigned goff[NGW], gpk[NGW];
#pragma unroll
    for (int k = 0; k < NGW; ++k) {
      const int f = 256 * (wv + 4 * k) + 4 * lane;
      const int r = f / GP, w = f - r * GP;
      const int row = w / KW, e = w % KW;
      const int rz = row / TY, ry = (row % TY) * YR + e / KWX, col = e % KWX;
      const bool ok = f < NGF && w < ROWS * KW && g0 + r < p.Cg;
      goff[k] = ok ? ((unsigned)r * (unsigned)gvol + (unsigned)((rz * p.Ho + ry) * p.Wo + col)) * 4u : DMA_OOB;
      gpk[k] = (unsigned)rz | ((unsigned)ry << 8) | ((unsigned)col << 16);
    }
    const bool g_exact = p.Do % TZ == 0 && p.Ho % TYB == 0 && p.Wo % KWX == 0;
    int bxi, byi, bzi, b;  // the brick to stage next (decoded once, then advanced like an odometer)
    {
      long long q = s0;
      bxi = (int)(q % p.bx); q /= p.bx;
      byi = (int)(q % p.by); q /= p.by;
      bzi = (int)(q % p.bz);
      b = (int)(q / p.bz);
    }
    auto stage = [&](int buf) {
      const int oz0 = bzi * TZ, oy0 = byi * TYB, ox0 = bxi * KWX;
      float* dbase = lds + buf * BUF;
      __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(G + ((size_t)b * p.Cg + g0) * gvol), (short)0,
                                                                     0x7fffffff, 0x00020000);
      const unsigned pos0 = (unsigned)(((oz0 * p.Ho + oy0) * p.Wo + ox0) * 4);
      const long long org = ((long long)(oz0 * S - p.pad) * p.Hi + (oy0 * S - p.pad)) * p.Wi + (ox0 * S - XL);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Src + ((size_t)b * p.Cs + c0) * svol + org),
                                                                     (short)0, 0x7fffffff, 0x00020000);
      const unsigned bmask = (bzi == 0 ? 1u : 0u) | (bzi == p.bz - 1 ? 2u : 0u) | (byi == 0 ? 4u : 0u) |
                             (byi == p.by - 1 ? 8u : 0u) | (bxi == 0 ? 16u : 0u) | (bxi == p.bx - 1 ? 32u : 0u);
#pragma unroll
      for (int i = 0; i < NGW; ++i) {
        if (256 * (wv + 4 * i) < NGL) {  // wave-uniform
          unsigned v = goff[i];
          if (!g_exact) {
            const bool ok = oz0 + (int)(gpk[i] & 255u) < p.Do && oy0 + (int)((gpk[i] >> 8) & 255u) < p.Ho &&
                            ox0 + (int)(gpk[i] >> 16) < p.Wo;
            v = ok ? v : DMA_OOB;
          }
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr_t)(dbase + 256 * (wv + 4 * i)), 16, v, pos0, 0, 0);
        }
      }
      if (p.nsrc == 0) {
#pragma unroll
        for (int k = 0; k < NSW; ++k) {
          if (256 * (wv + 4 * k) < NSL) {  // wave-uniform
            const unsigned vo = (scls[k] & bmask) == 0u ? soff[k] : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dbase + NGL + 256 * (wv + 4 * k)), 16, vo, 0, 0, 0);
          }
        }
      } else {
        // one descriptor per channel of the chunk; a piece that straddles two channels is issued once per channel
        // under the lanes' predicate
        __amdgpu_buffer_rsrc_t rc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const bool live = c0 + c < p.Cs;
          const int ch = live ? c0 + c : 0;
          rc[c] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src[ch] + (size_t)b * (size_t)p.sbs[ch] + org), (short)0,
                                                    live ? 0x7fffffff : 0, 0x00020000);
        }
#pragma unroll
        for (int k = 0; k < NSW; ++k) {
          if (256 * (wv + 4 * k) < NSL) {  // wave-uniform
            const unsigned vo = (scls[k] & bmask) == 0u ? soff[k] : DMA_OOB;
#pragma unroll
            for (int c = 0; c < NC; ++c)
              if (256 * (wv + 4 * k) < (c + 1) * CHSP && 256 * (wv + 4 * k) + 256 > c * CHSP)  // the piece touches channel c
                if (sch[k] == c)
                  __builtin_amdgcn_raw_ptr_buffer_load_lds(rc[c], (lds_ptr_t)(dbase + NGL + 256 * (wv + 4 * k)), 16, vo, 0, 0, 0);
          }
        }
      }
      if (++bxi == p.bx) { bxi = 0; if (++byi == p.by) { byi = 0; if (++bzi == p.bz) { bzi = 0; ++b; } } }
    };
    if (s0 < s1) stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (long long st = s0; st < s1; ++st) {
      if (st + 1 < s1) stage(buf ^ 1);
      // the pieces of the next brick have landed; past the barrier the matrix waves are done reading `buf`
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      buf ^= 1;
    }
#else
    (void)gvol; (void)svol; (void)NGW; (void)NSW; (void)DMA_OOB;
#endif
    return;
  }

  // ---- matrix waves
  // B operand column offsets of this wave's column tiles
  int boff[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int tile = (n < FULL) ? wv * FULL + n : 4 * FULL + (wv >> 1);
    const int j = tile * 32 + l31;
    int off = 0;
    if (j < NTOT) {
      const int c = j / K3, r = j - c * K3;
      const int kz = r / (K * K), ky = (r / K) % K, kx = r % K;
      off = c * CHSP + kz * PSP + ky * XP + kx;
    }
    boff[n] = off + XL - p.pad;
  }

  f32x16 acc[FULL > 0 ? FULL : 1][MT];
  f32x16 acch;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acch[r] = 0.f;
#pragma unroll
    for (int n = 0; n < FULL; ++n)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[n][m][r] = 0.f;
  }

  __builtin_amdgcn_s_barrier();  // brick s0 has landed
  int buf = 0;
#ifdef FS_WRW_STAMPS
  unsigned long long dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  for (long long st = s0; st < s1; ++st) {
    const float* sG = lds + buf * BUF;
    const float* sS = sG + NGL;
    // reduction pair kk of row `row`: positions ox = 2 kk + kh
    auto lds_ops = [&](int q, float (&a)[MT], float& ah, float (&bq)[NB]) {
      const int row = q / (KW / 2), kk = q % (KW / 2);
      const int ox = 2 * kk + kh;
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = sG[(m * 32 + l31) * GP + row * KW + ox];
      if (HALF) ah = sG[((wv & 1) * 32 + l31) * GP + row * KW + ox];
#pragma unroll
      for (int n = 0; n < NB; ++n)
        bq[n] = sS[boff[n] + (row / TY) * S * PSP + ((row % TY) * YR + (2 * kk) / KWX) * S * XP +
                   ((2 * kk) % KWX + kh) * S];
    };
    auto mma = [&](const float (&a)[MT], const float& ah, const float (&bq)[NB]) {
#pragma unroll
      for (int n = 0; n < FULL; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bq[n], acc[n][m], 0, 0, 0);
      if (HALF) acch = __builtin_amdgcn_mfma_f32_32x32x2f32(ah, bq[NB - 1], acch, 0, 0, 0);
    };
    constexpr int NQ = ROWS * (KW / 2);
    float a0[MT], b0[NB], a1[MT], b1[NB], h0 = 0.f, h1 = 0.f;
    lds_ops(0, a0, h0, b0);
#pragma unroll
    for (int q = 0; q < NQ; q += 2) {
      lds_ops(q + 1, a1, h1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, h0, b0);
      if (q + 2 < NQ) lds_ops(q + 2, a0, h0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, h1, b1);
    }
    STAMP(2);
    __builtin_amdgcn_s_barrier();  // the next brick has landed, everyone is done reading `buf`
    STAMP(4);
    buf ^= 1;
  }
#ifdef FS_WRW_STAMPS
  if (blockIdx.x == 3 && blockIdx.y == 1 && blockIdx.z == 0 && lane == 0) {
    for (int i = 0; i < 8; ++i) fs_wrw_dbg[wv * 8 + i] = dt[i];
    fs_wrw_dbg[wv * 8 + 7] = (unsigned long long)(s1 - s0);
  }
#endif

  auto flush = [&](const f32x16& a, int tile, int m) {
    const int j = tile * 32 + l31;
    if (j >= NTOT || c0 * K3 + j >= p.Cs * K3) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int g = g0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (g < p.Cg) {
        float* q = dW + (size_t)blockIdx.x * p.slab + (size_t)g * p.Cs * K3 + (size_t)c0 * K3 + j;
        if (p.slab) *q = a[r]; else atomicAdd(q, a[r]);
      }
    }
  };
#pragma unroll
  for (int n = 0; n < FULL; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) flush(acc[n][m], wv * FULL + n, m);
  if (HALF) flush(acch, 4 * FULL + (wv >> 1), wv & 1);
}

template <int K, int S, int NC, int MT, int TZ, int TY, int FULL, int HALF, int KWX = 32>
int launch_dma(const float* G, const float* Src, float* dW, const WP& w, hipStream_t st, const WDet* det = nullptr) {
  WB p;
  p.B = w.B; p.Cg = w.Cg; p.Cs = w.Cs; p.Do = w.Do; p.Ho = w.Ho; p.Wo = w.Wo;
  p.Di = w.Di; p.Hi = w.Hi; p.Wi = w.Wi; p.pad = w.pad;
  p.nsrc = w.nsrc;
  for (int c = 0; c < w.nsrc; ++c) { p.src[c] = w.srcv[c]; p.sbs[c] = w.sbsv[c]; }
  p.bz = fs::cdiv(p.Do, TZ); p.by = fs::cdiv(p.Ho, TY * (KW / KWX)); p.bx = fs::cdiv(p.Wo, KWX);
  p.bricks = (long long)p.B * p.bz * p.by * p.bx;
  const int mtiles = fs::cdiv(p.Cg, 32 * MT);
  const int nchunks = fs::cdiv(p.Cs, NC);
  // one workgroup per CU: 256 equal runs of bricks
  long long want = 256 / ((long long)mtiles * nchunks);
  if (want < 1) want = 1;
  long long spw = (p.bricks + want - 1) / want;
  if (spw < 1) spw = 1;
  p.spw = (int)(spw > (1 << 20) ? (1 << 20) : spw);
  const long long gx = (p.bricks + p.spw - 1) / p.spw;
  if (gx >= (1ll << 31) || nchunks > 65535 || mtiles > 65535) return FS_ERR_SHAPE;
  constexpr int K3 = K * K * K;
  float* out;
  const long long dwf = (long long)p.Cg * p.Cs * K3;
  const int drc = wrw_det_begin(det, gx, dwf, dW, &out, &p.slab);
  if (drc >= 0) return drc;
  hipLaunchKernelGGL((conv3d_wrw_dma_kernel<K, S, NC, MT, TZ, TY, FULL, HALF, KWX>), dim3((unsigned)gx, nchunks, mtiles),
                     dim3(512), 0, st, G, Src, out, p);
  wrw_det_end(det, gx, dwf, dW, st);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

#include "convwrwwino.hpp"
#include "convwrwwino4.hpp"

}  // namespace

#ifdef FS_WRW_STAMPS
extern "C" int fs_debug_wrw_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs_wrw_dbg), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

// `plan` != nullptr: write the FS_WRW_KERNEL_* id the call dispatches to and launch nothing (fs_conv3d_wrw_kernel_id)
#define FS_WRW_PICK(id, ...)        \
  do {                              \
    if (plan != nullptr) { *plan = (id); return FS_OK; } \
    return __VA_ARGS__;             \
  } while (0)

static int conv3d_wrw_impl(const float* g, const float* src, const float* const* srcv, const long long* sbsv, float* dw,
                           int B, int Cg, int Cs, int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kernel, int stride,
                           int pad, fs_stream_t stream, int* plan = nullptr, const WDet* det = nullptr) {
  FS_REQUIRE_PTR(g);
  if (plan == nullptr) FS_REQUIRE_PTR(dw);
  if (srcv == nullptr) FS_REQUIRE_PTR(src);
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
  bool ms_aligned = true;
  if (srcv != nullptr) {
    if (Cs > 12) return FS_ERR_UNSUPPORTED;
    p.nsrc = Cs; p.srcv = srcv; p.sbsv = sbsv;
    for (int c = 0; c < Cs; ++c) {
      if (srcv[c] == nullptr) return FS_ERR_NULLPTR;
      if (sbsv[c] < (long long)Di * Hi * Wi) return FS_ERR_ARG;
      ms_aligned = ms_aligned && ((uintptr_t)srcv[c] & 15) == 0 && sbsv[c] % 4 == 0;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  // DMA-staged kernel: 16-byte pieces (W % 4 == 0 on both grids, 16-byte aligned tensors), a full brick column,
  // pad <= stride, 31-bit byte offsets inside one (b, chunk) slab.  `FLOWSCI_WRW_REG=1`: the register-staged
  // kernel everywhere (scripts/wrwbench.py compares the two).
  static const bool reg_only = FS_AB_ENV("FLOWSCI_WRW_REG");
  const bool dma_ok = !reg_only && pad <= stride && (Wo >= KW || Wo == 16) && Wo % 4 == 0 && Wi % 4 == 0 &&
                      (((uintptr_t)g | (srcv ? (uintptr_t)0 : (uintptr_t)src)) & 15) == 0 && ms_aligned &&
                      (long long)64 * Do * Ho * Wo * 4 < (1ll << 31) && (long long)16 * Di * Hi * Wi * 4 < (1ll << 31);
  // per-channel planes: only the loader-wave kernel of IFBlock's conv0[0] (k = 4, <= 32 gradient channels)
  if (srcv != nullptr) {
    if (dma_ok && kernel == 4 && Cg <= 32 && Cs >= 3 && Wo >= KW) FS_WRW_PICK(FS_WRW_KERNEL_DMA, launch_dma<4, 2, 6, 1, 2, 2, 3, 0>(g, src, dw, p, st, det));
    return FS_ERR_UNSUPPORTED;
  }
  // the 64 -> 64 k3 layers of the 64^3 trunk: the Winograd F(4,3) form (convwrwwino4.hpp), or F(2,3) (convwrwwino.hpp)
  if (dma_ok && wrw_wino4_ok(p, g, src, kernel, stride)) FS_WRW_PICK(FS_WRW_KERNEL_WINO43, launch_wrw_wino4(g, src, dw, p, st, det));
#ifdef FS_ABLATION
  if (dma_ok && det == nullptr && wrw_wino_ok(p, g, src, kernel, stride)) FS_WRW_PICK(FS_WRW_KERNEL_WINO23, launch_wrw_wino(g, src, dw, p, st));
#endif
  if (dma_ok) {
    if (kernel == 3 && Cg > 32 && Cs >= 8 && Wo == 16) FS_WRW_PICK(FS_WRW_KERNEL_DMA, launch_dma<3, 1, 16, 2, 1, 4, 3, 1, 16>(g, src, dw, p, st, det));
    if (kernel == 3 && Cg > 32 && Cs >= 8) FS_WRW_PICK(FS_WRW_KERNEL_DMA, launch_dma<3, 1, 16, 2, 1, 4, 3, 1>(g, src, dw, p, st, det));
    if (kernel == 4 && Cg > 32 && Cs >= 4 && Wo == 16) FS_WRW_PICK(FS_WRW_KERNEL_DMA, launch_dma<4, 2, 8, 2, 1, 2, 4, 0, 16>(g, src, dw, p, st, det));
    if (kernel == 4 && Cg > 32 && Cs >= 4 && Wo >= KW) FS_WRW_PICK(FS_WRW_KERNEL_DMA, launch_dma<4, 2, 8, 2, 1, 2, 4, 0>(g, src, dw, p, st, det));
    if (kernel == 4 && Cg <= 32 && Cs >= 3 && Wo >= KW) FS_WRW_PICK(FS_WRW_KERNEL_DMA, launch_dma<4, 2, 6, 1, 2, 2, 3, 0>(g, src, dw, p, st, det));
    // 1-2 source channels (the mask head: 128 columns, one 32-column tile per matrix wave): bound by the G stream
    if (kernel == 4 && Cg <= 32 && Wo >= KW) FS_WRW_PICK(FS_WRW_KERNEL_DMA, launch_dma<4, 2, 2, 1, 2, 2, 1, 0>(g, src, dw, p, st, det));
  }
  if (kernel == 3) FS_WRW_PICK(FS_WRW_KERNEL_BRICK, launch_brick<3, 1, 8, 1, 4>(g, src, dw, p, st, det));
  // k = 4: 64 columns per source channel.  NC = 4 gives every wave two 32-column tiles, NC = 2 one;
  // pick the chunking with less padded matrix work (Cs = 1, 2, 5, 6: the IFNet heads / block0 input)
  const int cost2 = (Cs + 1) / 2, cost4 = 2 * ((Cs + 3) / 4);
  if (cost2 < cost4) FS_WRW_PICK(FS_WRW_KERNEL_BRICK, launch_brick<4, 2, 2, 1, 2>(g, src, dw, p, st, det));
  FS_WRW_PICK(FS_WRW_KERNEL_BRICK, launch_brick<4, 2, 4, 1, 2>(g, src, dw, p, st, det));
}

extern "C" int fs_conv3d_wrw_kernel_id(const float* g, const float* src, int B, int Cg, int Cs, int Do, int Ho, int Wo,
                                       int Di, int Hi, int Wi, int kernel, int stride, int pad) {
  int id = -1;
  const int rc = conv3d_wrw_impl(g, src, nullptr, nullptr, nullptr, B, Cg, Cs, Do, Ho, Wo, Di, Hi, Wi, kernel, stride, pad,
                                 nullptr, &id);
  return rc == FS_OK ? id : -rc;
}

extern "C" int fs_conv3d_wrw(const float* g, const float* src, float* dw, int B, int Cg, int Cs, int Do,
                             int Ho, int Wo, int Di, int Hi, int Wi, int kernel, int stride, int pad,
                             fs_stream_t stream) {
  FS_ENTER();
  return conv3d_wrw_impl(g, src, nullptr, nullptr, dw, B, Cg, Cs, Do, Ho, Wo, Di, Hi, Wi, kernel, stride, pad, stream);
}

// Deterministic form: the same kernels with a caller-owned workspace instead of float atomics (see WDet above); dw is
// overwritten (no zero fill needed).  fs_conv3d_wrw_det_ws_floats: the workspace the call needs (its own dispatch, nothing
// launched), or -(FS_ERR_*).  `src_planes` / `batch_strides` non-NULL: the multi-source form (fs_conv3d_wrw_ms's arguments).
extern "C" long long fs_conv3d_wrw_det_ws_floats(const float* g, const float* src, const float* const* src_planes,
                                                 const long long* batch_strides, int B, int Cg, int Cs, int Do, int Ho,
                                                 int Wo, int Di, int Hi, int Wi, int kernel, int stride, int pad) {
  long long need = 0;
  WDet det = {nullptr, 0, &need};
  float dummy;
  const int rc = conv3d_wrw_impl(g, src, src_planes, batch_strides, &dummy, B, Cg, Cs, Do, Ho, Wo, Di, Hi, Wi, kernel, stride,
                                 pad, nullptr, nullptr, &det);
  return rc == FS_OK ? need : -(long long)rc;
}

extern "C" int fs_conv3d_wrw_det(const float* g, const float* src, const float* const* src_planes,
                                 const long long* batch_strides, float* dw, float* ws, long long ws_floats, int B, int Cg,
                                 int Cs, int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kernel, int stride, int pad,
                                 fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(ws);
  if ((src_planes == nullptr) != (batch_strides == nullptr)) return FS_ERR_NULLPTR;
  WDet det = {ws, ws_floats, nullptr};
  return conv3d_wrw_impl(g, src, src_planes, batch_strides, dw, B, Cg, Cs, Do, Ho, Wo, Di, Hi, Wi, kernel, stride, pad, stream,
                         nullptr, &det);
}

// fs_conv3d_wrw over a source that is never concatenated: channel c of the [B, Cs, Di,Hi,Wi] source is the plane
// src[c] (sample 0) of a tensor with batch stride batch_strides[c] floats (host arrays, read at launch; Cs <= 12,
// 16-byte aligned planes, strides multiples of 4).  FS_ERR_UNSUPPORTED when the shape has no loader-wave kernel.
extern "C" int fs_conv3d_wrw_ms(const float* g, const float* const* src, const long long* batch_strides, float* dw,
                                int B, int Cg, int Cs, int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kernel,
                                int stride, int pad, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(src); FS_REQUIRE_PTR(batch_strides);
  return conv3d_wrw_impl(g, nullptr, src, batch_strides, dw, B, Cg, Cs, Do, Ho, Wo, Di, Hi, Wi, kernel, stride, pad, stream);
}
