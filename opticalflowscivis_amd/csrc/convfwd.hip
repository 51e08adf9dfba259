// convfwd.hip -- forward pass of the IFNet-3D convolutions (and, with flipped / transposed weights,
// the input gradient of its stride-1 layers and of its transposed convolutions) as an implicit GEMM
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) for gfx950, NCDHW in, NCDHW out.
//
// Not one of the §8(a) rows: a companion of convwrw.hip.  MIOpen (ROCm 7.2, no gfx950 tuning db)
// runs these layers through CK's NDHWC kernels behind layout transposes (~58 TFLOP/s + 14 ms of
// batched_transpose per 256^3 step), and the strided / transposed ones through GEMM + Col2Im
// (~27 TFLOP/s) -- profiles/r01_bench_256_kernel_stats.csv.
//
//   Y[b, co, oz,oy,ox] = bias[co] + sum_{ci,kz,ky,kx} W[co, ci, kz,ky,kx] * X[b, ci, oz*s+kz-p, ...]
//
//   M = Cout (32 / 64 per workgroup),  N = output voxels,  K = Cin * k^3.
//
// Decomposition.  A workgroup (4 waves) owns a TZ x (TY*32/TW) x TW brick of output voxels for 32*MT
// output channels.  The reduction runs over chunks of CI input channels: per chunk the input brick
// with its halo ((TZ-1)s+k x (TYR-1)s+k x (TW-1)s+k, zero padded) and the CI*k^3 x 32*MT weight slab
// are staged in LDS, then each wave issues MT*NT MFMAs per reduction pair for its NT 32-voxel
// column tiles.  The two reduction elements of a 32x32x2 MFMA are the same tap of input channels cl
// and cl + CI/2, so both operand addresses are "per-lane base + compile-time immediate": im2col is
// only an LDS addressing pattern.  LDS reads run one pair ahead of the MFMAs (register rotation).
// Two workgroups per CU (<= 80 KB LDS each): one stages while the other feeds the matrix cores.
//
// Weights arrive re-laid-out as Wt[ci][tap][co] (co contiguous, zero padded to the tile sizes) by
// wprep_kernel, which also applies the flip + transpose that turns the stride-1 input gradient
// into a forward convolution.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FP {
  int B, Cin, Cout, CoutP;
  int Di, Hi, Wi, Do, Ho, Wo;
  int pad;
  int tz, ty, tx;  // tiles per axis
  long long tiles;
  // optional fused PReLU: Z = prelu(Y) written next to Y (Y is kept: the PReLU backward needs it)
  const float* slope;
  float* Z;
  int nslope;  // 1 (shared) or Cout
  // optional tensor of the output's shape added in the epilogue: to Z after the PReLU when Z is
  // written (the residual of an IFBlock unit, `convblock(x) + x`), else to Y (the residual branch of
  // that unit's input gradient)
  const float* addend;
  // optional fused PReLU backward (loader-wave kernels, input-gradient use): the convolution's result g is the
  // gradient w.r.t. z = prelu(act_y); the epilogue stores g * prelu'(act_y) instead and leaves per-wave partial sums
  // of the slope gradient (sum g * act_y * [act_y <= 0]) and of the stored values (the producing layer's bias
  // gradient) in dpart[(workgroup * 4 + wave) * CP * 2 + channel * 2 + {0, 1}]
  const float* dy;
  const float* dslope;
  int dnslope;
  float* dpart;
  // optional multi-source input (loader-wave kernels): the Cin = nsrc <= 12 input channels are planes of
  // different tensors -- src[c] = channel c of sample 0, sbs[c] = its tensor's batch stride in floats -- so that
  // IFBlock's `torch.cat((img0, img1, warped0, warped1, mask, flow), 1)` (Flow-3D/model/IFNet.py:183) is never
  // materialised: the loader waves read each plane where it lies.  nsrc = 0: one tensor X.
  int nsrc;
  const float* src[12];
  long long sbs[12];
#ifdef FS_ABLATION
  int ab = 0;  // measurement switches of the kernel a launch takes (FLOWSCI_WINO2D_AB)
#endif
};

// the weight re-layout Wt[ci][tap][co] (FS_WPREP_FWD) lives in wprep.hpp
#include "wprep.hpp"

template <int K, int S, int CI, int MT, int NT, int TZ, int TY, int TW>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_kernel(const float* __restrict__ X,
                                                         const float* __restrict__ Wt,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ Y, FP p) {
  constexpr int K3 = K * K * K;
  constexpr int R = 32 / TW;    // output rows (y) inside one 32-column MFMA tile
  constexpr int TYR = TY * R;   // output rows (y) of the brick
  static_assert(TZ * TY == 4 * NT && TY % NT == 0 && (CI % 2) == 0 && (TW == 32 || TW == 16), "tile shape");
  constexpr int ZT = (TZ - 1) * S + K, YT = (TYR - 1) * S + K, XT = (TW - 1) * S + K;
  constexpr int PS = YT * XT, CHS = ZT * PS;  // plane / channel pitch of the staged brick
  constexpr int CP = 32 * MT;
  constexpr int NP = (CI / 2) * K3;           // reduction pairs per chunk
  constexpr int NX = CI * CHS, NW = CI * K3 * CP;
  __shared__ float sX[NX];
  __shared__ __attribute__((aligned(16))) float sW[NW];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int col = lane & 31, kh = lane >> 5;

  // brick of this workgroup.  Consecutive workgroup ids land on different XCDs (round robin over
  // 8); give each XCD a contiguous range of bricks so that halo re-reads hit its own L2.
  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int oz0 = tzi * TZ, oy0 = tyi * TYR, ox0 = txi * TW;
  const int co0 = blockIdx.y * CP;

  const int ly = col / TW, lx = col % TW;
  const int rr0 = wv * NT;                // first row slot of this wave (all NT share one z)
  const int wz = rr0 / TY, wy = rr0 % TY;
  const float* bB = sX + kh * (CI / 2) * CHS + (wz * S) * PS + ((wy * R + ly) * S) * XT + lx * S;
  const float* aB = sW + kh * (CI / 2) * K3 * CP + col;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  const int gz0 = oz0 * S - p.pad, gy0 = oy0 * S - p.pad, gx0 = ox0 * S - p.pad;
  constexpr int ITX = (NX + 255) / 256;
  constexpr int ITW = (NW / 4 + 255) / 256;

  // Byte offset of every brick element this thread stages, relative to the chunk's first channel
  // (~0u = outside the volume -> zero).  They do not depend on the chunk, so the per-chunk staging is
  // loads only: wave-uniform base + 32-bit lane offset, one address VGPR per load.
  unsigned xoff[ITX];
#pragma unroll
  for (int it = 0; it < ITX; ++it) {
    const int i = t + 256 * it;
    const int c = i / CHS, r1 = i - c * CHS;
    const int z = r1 / PS, r2 = r1 - z * PS;
    const int y = r2 / XT, x = r2 - y * XT;
    const int gz = gz0 + z, gy = gy0 + y, gx = gx0 + x;
    const bool ok = i < NX && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
    xoff[it] = ok ? ((unsigned)c * (unsigned)xvol + ((unsigned)gz * p.Hi + gy) * p.Wi + gx) * 4u : ~0u;
  }
  // PF (k = 4: short MFMA phases, large bricks): the next chunk is fetched into registers before the
  // MFMA phase and lands under it.  k = 3 stages straight into LDS (its MFMA phase is 3x longer and
  // its 128 accumulators leave no room for 50 prefetch registers).
  constexpr bool PF = true, PFW = (K == 4);
  float rX[PF ? ITX : 1];
  float4 rW[PFW ? ITW : 1];
  auto load_x = [&](int c0, int it) -> float {
    const char* xb = reinterpret_cast<const char*>(X + ((size_t)b * p.Cin + c0) * xvol);
    const bool chan_ok = (c0 + CI <= p.Cin) || (c0 + (t + 256 * it) / CHS < p.Cin);
    return (xoff[it] != ~0u && chan_ok) ? *reinterpret_cast<const float*>(xb + xoff[it]) : 0.f;
  };
  auto load_w = [&](int c0, int it) -> float4 {
    const int i4 = t + 256 * it;
    const int row = i4 / (CP / 4), j4 = i4 - row * (CP / 4);
    const float* wb = Wt + (size_t)c0 * K3 * p.CoutP + co0;
    return (i4 < NW / 4) ? *reinterpret_cast<const float4*>(wb + (size_t)row * p.CoutP + 4 * j4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto fetch = [&](int c0) {
#pragma unroll
    for (int it = 0; it < ITX; ++it) rX[PF ? it : 0] = load_x(c0, it);
#pragma unroll
    for (int it = 0; it < ITW; ++it)
      if (PFW) rW[PFW ? it : 0] = load_w(c0, it);
  };
  auto park = [&](int c0) {  // PF: registers -> LDS;  !PF: global -> LDS
#pragma unroll
    for (int it = 0; it < ITX; ++it) {
      const int i = t + 256 * it;
      const float v = PF ? rX[PF ? it : 0] : load_x(c0, it);
      if (i < NX) sX[i] = v;
    }
#pragma unroll
    for (int it = 0; it < ITW; ++it) {
      const int i4 = t + 256 * it;
      const float4 v = PFW ? rW[PFW ? it : 0] : load_w(c0, it);
      if (i4 < NW / 4) reinterpret_cast<float4*>(sW)[i4] = v;
    }
  };

  if (PF) fetch(0);
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
    park(c0);
    __syncthreads();
    if (PF && c0 + CI < p.Cin) fetch(c0 + CI);

    // ---- MFMA phase: pair j = (cl, tap); lanes 32..63 feed channel cl + CI/2
    auto lds_ops = [&](int j, float (&a)[MT], float (&bq)[NT]) {
      const int cl = j / K3, tap = j - cl * K3;
      const int kz = tap / (K * K), ky = (tap / K) % K, kx = tap % K;
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = aB[(cl * K3 + tap) * CP + m * 32];
#pragma unroll
      for (int n = 0; n < NT; ++n) bq[n] = bB[cl * CHS + kz * PS + (ky + n * R * S) * XT + kx];
    };
    auto mma = [&](const float (&a)[MT], const float (&bq)[NT]) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bq[n], acc[m][n], 0, 0, 0);
    };
    float a0[MT], b0[NT], a1[MT], b1[NT];
    lds_ops(0, a0, b0);
#pragma unroll
    for (int j = 0; j < NP; j += 2) {
      if (j + 1 < NP) lds_ops(j + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      if (j + 2 < NP) lds_ops(j + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < NP) mma(a1, b1);
    }
    __syncthreads();
  }

  // ---- epilogue: row (channel) = (r&3) + 8*(r>>2) + 4*kh, column (voxel) = lane & 31
  // (restrict-qualified local copies: without them every addend load has to wait for the previous
  // store -- the outputs might alias it -- and the epilogue becomes a chain of memory round trips)
  const float* __restrict__ ad = p.addend;
  const float* __restrict__ slope = p.slope;
  float* __restrict__ Zp = p.Z;
  float* __restrict__ Yp = Y;
  const int oz = oz0 + wz;
  const int ox = ox0 + lx;
  if (oz < p.Do && ox < p.Wo) {
    const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;
    float bv[MT][16], sv[MT][16];  // this lane's 16*MT channels: bias and PReLU slope, loaded once
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        bv[m][r] = (bias != nullptr && co < p.Cout) ? bias[co] : 0.f;
        sv[m][r] = (Zp != nullptr && co < p.Cout) ? slope[p.nslope == 1 ? 0 : co] : 0.f;
      }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int oy = oy0 + (wy + n) * R + ly;
      if (oy >= p.Ho) continue;
      const size_t o0 = (size_t)b * p.Cout * yvol + ((size_t)oz * p.Ho + oy) * p.Wo + ox;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // all loads of the tile first
          const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          av[r] = (ad != nullptr && co < p.Cout) ? ad[o0 + (size_t)co * yvol] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (co < p.Cout) {
            const float v = acc[m][n][r] + bv[m][r];
            const size_t o = o0 + (size_t)co * yvol;
            if (Zp != nullptr) {
              Yp[o] = v;
              Zp[o] = (v > 0.f ? v : sv[m][r] * v) + av[r];
            } else {
              Yp[o] = v + av[r];
            }
          }
        }
      }
    }
  }
}


// ---- loader-wave form (one 8-wave workgroup per CU, two LDS buffers) ---------------------------------
// Same decomposition; the staging moves to partner waves and to `buffer_load_dwordx4 ... lds` (global -> LDS,
// no VGPR stop, no ds_write pass; csrc/convwrw.hip has the measurements that led here).  Waves 4-7 issue the
// 16-byte pieces of chunk c+1 -- the haloed input brick and the weight slab -- into the second LDS buffer while
// waves 0-3 run the MFMA phase of chunk c on the first; one barrier per chunk.  The matrix waves' stream is
// operand reads + MFMAs only, and with no staging registers they can hold twice the column tiles (k = 4).
// Rows of the staged brick start 4 floats left of output column 0's first tap + pad (16-byte aligned in
// memory when Wi % 4 == 0), the row pitch is a multiple of 4 floats; a piece is inside or outside the volume
// as a whole, and outside pieces / channels past Cin carry an out-of-range offset: the DMA writes 0 there.
// A workgroup keeps its brick for all chunks, so the piece offsets are computed once.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned DMA_OOB = 0x80000000u;
constexpr int up4(int n) { return (n + 3) / 4 * 4; }

template <int K, int S, int CI, int MT, int NT, int TZ, int TY, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_fwd_ws_kernel(const float* __restrict__ X,
                                                            const float* __restrict__ Wt,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ Y, FP p) {
  constexpr int K3 = K * K * K;
  constexpr int R = 32 / TW;
  constexpr int TYR = TY * R;
  static_assert(TZ * TY == 4 * NT && TY % NT == 0 && (CI % 2) == 0 && (TW == 32 || TW == 16), "tile shape");
  constexpr int XL = 4;
  constexpr int ZT = (TZ - 1) * S + K, YT = (TYR - 1) * S + K;
  constexpr int XP = up4(XL + (TW - 1) * S + K);  // row pitch = staged row length
  constexpr int PS = YT * XP, CHS = ZT * PS;
  constexpr int CP = 32 * MT;
  constexpr int EPI_ROWS = 4;  // matrix waves per workgroup (convfwd_epilogue.hpp)
  constexpr int NP = (CI / 2) * K3;
  constexpr int NX = CI * CHS, NW = CI * K3 * CP;
  constexpr int NXL = (NX + 255) / 256 * 256, NWL = (NW + 255) / 256 * 256;
  constexpr int NXW = (NXL / 256 + 3) / 4, NWW = (NWL / 256 + 3) / 4;  // pieces per loader wave
  constexpr int BUF = NXL + NWL;
  static_assert(2 * BUF * 4 <= 160 * 1024, "two buffers fit the CU's LDS");
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
  const int oz0 = tzi * TZ, oy0 = tyi * TYR, ox0 = txi * TW;
  const int co0 = blockIdx.y * CP;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass has neither the buffer-resource type nor the LDS-DMA builtin)
    __builtin_amdgcn_s_setprio(3);  // (a loader wave issues a handful of instructions per period: they should not queue behind the matrix wave's)
    // piece k of loader wave wv fills the 16-byte slots 256 (wv + 4 k) + 4 lane .. + 3 of an image
    const int gz0 = oz0 * S - p.pad, gy0 = oy0 * S - p.pad, gx0 = ox0 * S - XL;
    // x pieces: offset inside their channel and the channel (0 .. CI-1) of the chunk they belong to: a chunk's
    // channels have one descriptor each (they may be planes of different tensors), a piece that straddles two
    // channels is issued once per channel under the lanes' predicate
    unsigned xoff[NXW], woff[NWW];
    int xch[NXW];
#pragma unroll
    for (int k = 0; k < NXW; ++k) {
      const int i = 256 * (wv + 4 * k) + 4 * lane;
      const int c = i / CHS, r1 = i - c * CHS;
      const int z = r1 / PS, r2 = r1 - z * PS;
      const int y = r2 / XP, x = r2 - y * XP;
      const int gz = gz0 + z, gy = gy0 + y, gx = gx0 + x;
      const bool ok = i < NX && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
      xoff[k] = ok ? (((unsigned)gz * p.Hi + gy) * p.Wi + gx) * 4u : DMA_OOB;
      xch[k] = i < NX ? c : CI - 1;
    }
#pragma unroll
    for (int k = 0; k < NWW; ++k) {
      const int i = 256 * (wv + 4 * k) + 4 * lane;
      const int row = i / CP, j = i - row * CP;
      woff[k] = i < NW ? ((unsigned)row * (unsigned)p.CoutP + (unsigned)j) * 4u : DMA_OOB;
    }
    auto stage = [&](int c0, int buf) {
      // channels past Cin read as zero: their descriptor is empty
      __amdgpu_buffer_rsrc_t rx[CI];
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const int ch = c0 + c;
        const bool live = ch < p.Cin;
        const int chc = live ? ch : 0;
        const float* base = p.nsrc ? p.src[chc] + (size_t)b * (size_t)p.sbs[chc]
                                   : X + ((size_t)b * p.Cin + chc) * xvol;
        rx[c] = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, live ? (int)((unsigned)xvol * 4u) : 0, 0x00020000);
      }
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(Wt + (size_t)c0 * K3 * p.CoutP + co0), (short)0, 0x7fffffff, 0x00020000);
      float* base = lds + buf * BUF;
#pragma unroll
      for (int k = 0; k < NXW; ++k)
        if (256 * (wv + 4 * k) < NXL) {  // wave-uniform
#pragma unroll
          for (int c = 0; c < CI; ++c)
            if (256 * (wv + 4 * k) < (c + 1) * CHS && 256 * (wv + 4 * k) + 256 > c * CHS)  // the piece touches channel c
              if (xch[k] == c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx[c], (lds_ptr_t)(base + 256 * (wv + 4 * k)), 16, xoff[k], 0, 0, 0);
        }
#pragma unroll
      for (int k = 0; k < NWW; ++k)
        if (256 * (wv + 4 * k) < NWL)  // wave-uniform
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
#else
    (void)xvol; (void)NXW; (void)NWW; (void)DMA_OOB;
#endif
    return;
  }

  // ---- matrix waves
  const int ly = col / TW, lx = col % TW;
  const int rr0 = wv * NT;
  const int wz = rr0 / TY, wy = rr0 % TY;
  const int bBo = kh * (CI / 2) * CHS + (wz * S) * PS + ((wy * R + ly) * S) * XP + lx * S + XL - p.pad;
  const int aBo = NXL + kh * (CI / 2) * K3 * CP + col;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  __builtin_amdgcn_s_barrier();  // chunk 0 has landed
  int buf = 0;
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
    const float* bB = lds + buf * BUF + bBo;
    const float* aB = lds + buf * BUF + aBo;
    auto lds_ops = [&](int j, float (&a)[MT], float (&bq)[NT]) {
      const int cl = j / K3, tap = j - cl * K3;
      const int kz = tap / (K * K), ky = (tap / K) % K, kx = tap % K;
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = aB[(cl * K3 + tap) * CP + m * 32];
#pragma unroll
      for (int n = 0; n < NT; ++n) bq[n] = bB[cl * CHS + kz * PS + (ky + n * R * S) * XP + kx];
    };
    auto mma = [&](const float (&a)[MT], const float (&bq)[NT]) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bq[n], acc[m][n], 0, 0, 0);
    };
    float a0[MT], b0[NT], a1[MT], b1[NT];
    lds_ops(0, a0, b0);
#pragma unroll
    for (int j = 0; j < NP; j += 2) {
      if (j + 1 < NP) lds_ops(j + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
#ifdef FS_ABLATION
      if (p.ab & 8) continue;  // (measurement: operand reads without the MFMAs -- FLOWSCI_FWD_AB=8)
#endif
      mma(a0, b0);
      if (j + 2 < NP) lds_ops(j + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < NP) mma(a1, b1);
    }
    __builtin_amdgcn_s_barrier();  // the next chunk has landed, everyone is done reading `buf`
    buf ^= 1;
  }

#include "convfwd_epilogue.hpp"
}

#ifdef FS_ABLATION  // the 1-D forms convwino2d.hpp superseded on every shape they cover: measurement builds only
#include "convwino.hpp"
#include "convwino4.hpp"
#endif
#include "convwino2d.hpp"
#include "convfwd_s3.hpp"

template <int K, int S, int CI, int MT, int NT, int TZ, int TY, int TW>
int launch_ws(const float* X, const float* Wt, const float* bias, float* Y, FP& p, hipStream_t st) {
  constexpr int TYR = TY * (32 / TW);
  p.tz = fs::cdiv(p.Do, TZ); p.ty = fs::cdiv(p.Ho, TYR); p.tx = fs::cdiv(p.Wo, TW);
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  const int mgroups = p.CoutP / (32 * MT);
  if (p.tiles >= (1ll << 31) || mgroups > 65535) return FS_ERR_SHAPE;
#ifdef FS_ABLATION
  static const int fwd_ab = (int)FS_AB_ENV_LL("FLOWSCI_FWD_AB", 0);  // 4: no epilogue, 8: no MFMAs (wrong results by design)
  p.ab = fwd_ab;
#endif
  hipLaunchKernelGGL((conv3d_fwd_ws_kernel<K, S, CI, MT, NT, TZ, TY, TW>), dim3((unsigned)p.tiles, mgroups),
                     dim3(512), 0, st, X, Wt, bias, Y, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int K, int S, int CI, int MT, int NT, int TZ, int TY, int TW>
int launch(const float* X, const float* Wt, const float* bias, float* Y, FP& p, hipStream_t st) {
  constexpr int TYR = TY * (32 / TW);
  p.tz = fs::cdiv(p.Do, TZ); p.ty = fs::cdiv(p.Ho, TYR); p.tx = fs::cdiv(p.Wo, TW);
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  const int mgroups = p.CoutP / (32 * MT);
  if (p.tiles >= (1ll << 31) || mgroups > 65535) return FS_ERR_SHAPE;
  hipLaunchKernelGGL((conv3d_fwd_kernel<K, S, CI, MT, NT, TZ, TY, TW>), dim3((unsigned)p.tiles, mgroups),
                     dim3(256), 0, st, X, Wt, bias, Y, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// The per-wave partial rows [channel group][rows][CP channels][2: slope-gradient term, bias-gradient term] are summed
// in two fixed stages, fp64 -- deterministic: kFinishBlocks workgroups per channel group reduce contiguous row ranges
// reading whole rows (a block per output walking its column with a 256-byte stride took 47 us per head), then one
// block per output adds the kFinishBlocks partials in order.  ga[c]: all channels when the slope is shared.
constexpr int kFinishBlocks = 128;
// part: [channel group][rows][CP channels][2]; blockIdx = (row range, channel group)
__global__ __launch_bounds__(256) void dprelu_finish1_kernel(const float* __restrict__ part, int rows, int CP,
                                                             double* __restrict__ partial) {
  const int CP2 = 2 * CP;  // 64 or 128 columns
  const int col = threadIdx.x % CP2, sub = threadIdx.x / CP2, nsub = 256 / CP2;
  const int chunk = (rows + kFinishBlocks - 1) / kFinishBlocks;
  const int r0 = blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
  const float* src = part + (size_t)blockIdx.y * rows * CP2;
  double s = 0.0;
  for (int r = r0 + sub; r < r1; r += nsub) s += (double)src[(size_t)r * CP2 + col];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0) {
    double t = red[col];
    for (int k = 1; k < nsub; ++k) t += red[k * CP2 + col];
    partial[((size_t)blockIdx.y * kFinishBlocks + blockIdx.x) * CP2 + col] = t;
  }
}
__global__ __launch_bounds__(64) void dprelu_finish2_kernel(const double* __restrict__ partial, int CP, int Cout,
                                                            float* __restrict__ ga, float* __restrict__ gb, int nslope) {
  const bool bias_blk = (int)blockIdx.x >= nslope;
  const int c = bias_blk ? blockIdx.x - nslope : blockIdx.x;
  const bool all = !bias_blk && nslope == 1 && Cout != 1;
  const int c_lo = all ? 0 : c, c_hi = all ? Cout : c + 1;
  double s = 0.0;
  for (int cc = c_lo; cc < c_hi; ++cc) {
    const double* src = partial + (size_t)(cc / CP) * kFinishBlocks * 2 * CP + (cc % CP) * 2 + (bias_blk ? 1 : 0);
    for (int g = threadIdx.x; g < kFinishBlocks; g += 64) s += src[(size_t)g * 2 * CP];
  }
  __shared__ double red[64];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 32; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) (bias_blk ? gb : ga)[c] = (float)red[0];
}

// channel padding of the re-laid-out weights for a layer
void wt_dims(int Cin, int Cout, int kernel, int* CinP, int* CoutP) {
  const int ci = (kernel == 3) ? 4 : 2;
  *CinP = (Cin + ci - 1) / ci * ci;
  *CoutP = (Cout <= 32 && kernel == 4) ? 32 : (Cout + 63) / 64 * 64;
}

}  // namespace

extern "C" long long fs_conv3d_fwd_ws_floats(int Cin, int Cout, int kernel) {
  if (Cin < 1 || Cout < 1 || (kernel != 3 && kernel != 4)) return -1;
  int cinp, coutp;
  wt_dims(Cin, Cout, kernel, &cinp, &coutp);
  const long long direct = (long long)cinp * kernel * kernel * kernel * coutp;
  // the Winograd slabs of the 64-channel k3 layers are larger: F(2,3) 4/3, F(4,3) twice the taps
  constexpr int wino_uch = FS_WINO2D_UCH > FS_WINO4_UCH ? FS_WINO2D_UCH : (FS_WINO4_UCH > FS_WINO_UCH ? FS_WINO4_UCH : FS_WINO_UCH);
  const long long wino = (kernel == 3 && coutp == 64) ? (long long)cinp * wino_uch : 0;
  // the pre-split bf16 slab of the k = 4 layers (convfwd_s3.hpp): three 2-byte pieces per weight
  const long long s3 = (kernel == 4 && coutp <= 64) ? s3_slab_words(Cin, coutp) : 0;
  const long long m = direct > wino ? direct : wino;
  return m > s3 ? m : s3;
}

struct DPrelu {  // fused PReLU backward of the layer that produced this convolution's (gradient) output
  const float* act_y = nullptr;
  const float* slope = nullptr;
  int nslope = 0;
  float* part = nullptr;
  float* ga = nullptr;
  float* gb = nullptr;
};

struct MultiSrc {  // the input as per-channel planes (FP::src / FP::sbs)
  const float* const* src;
  const long long* batch_strides;
};

static int conv3d_fwd_impl(const float* x, const float* w, const float* bias, const float* slope, int nslope,
                           const float* addend, float* y, float* z, float* ws, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do,
                           int Ho, int Wo, int kernel, int stride, int pad, int wmode, fs_stream_t stream,
                           const DPrelu* dp = nullptr, const MultiSrc* ms = nullptr, WprepPlan* plan = nullptr) {
  if (ms == nullptr && plan == nullptr) FS_REQUIRE_PTR(x);
  // w == NULL: `ws` already holds the re-laid-out weights (fs_conv3d_wprep_batch)
  if (plan == nullptr) FS_REQUIRE_PTR(y);
  FS_REQUIRE_PTR(ws);
  if (z != nullptr && (slope == nullptr || (nslope != 1 && nslope != Cout))) return FS_ERR_ARG;
  if (B < 1 || Cin < 1 || Cout < 1 || Di < 1 || Hi < 1 || Wi < 1 || Do < 1 || Ho < 1 || Wo < 1)
    return FS_ERR_SHAPE;
  if (!((kernel == 3 && stride == 1) || (kernel == 4 && stride == 2)) || pad < 0 || pad >= kernel ||
      (wmode != 0 && wmode != 1))
    return FS_ERR_ARG;
  // the output grid must be the convolution's: floor((in + 2p - k) / s) + 1
  if (Do != (Di + 2 * pad - kernel) / stride + 1 || Ho != (Hi + 2 * pad - kernel) / stride + 1 ||
      Wo != (Wi + 2 * pad - kernel) / stride + 1)
    return FS_ERR_SHAPE;
  if ((long long)Di * Hi * Wi >= (1ll << 31) || (long long)Do * Ho * Wo >= (1ll << 31)) return FS_ERR_SHAPE;
  // 32-bit byte offsets inside one staged channel chunk (4 channels for k = 3, 2 for k = 4)
  if ((long long)(kernel == 3 ? 4 : 2) * Di * Hi * Wi * 4 >= (1ll << 32)) return FS_ERR_SHAPE;
  FP p;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
  p.pad = pad;
  p.slope = slope; p.Z = z; p.nslope = nslope; p.addend = addend;
  p.dy = dp ? dp->act_y : nullptr; p.dslope = dp ? dp->slope : nullptr; p.dnslope = dp ? dp->nslope : 0;
  p.dpart = dp ? dp->part : nullptr;
  p.nsrc = 0;
  bool ms_aligned = true;
  if (ms != nullptr) {
    if (Cin > 12) return FS_ERR_UNSUPPORTED;
    p.nsrc = Cin;
    const long long vol = (long long)Di * Hi * Wi;
    for (int c = 0; c < Cin; ++c) {
      if (ms->src[c] == nullptr) return FS_ERR_NULLPTR;
      if (ms->batch_strides[c] < vol) return FS_ERR_ARG;
      p.src[c] = ms->src[c]; p.sbs[c] = ms->batch_strides[c];
      ms_aligned = ms_aligned && ((uintptr_t)ms->src[c] & 15) == 0 && ms->batch_strides[c] % 4 == 0;
    }
  }
  int cinp;
  wt_dims(Cin, Cout, kernel, &cinp, &p.CoutP);
  hipStream_t st = (hipStream_t)stream;
  // after a loader-wave launch with the fused PReLU-backward epilogue: finish its partial sums (CP = channels per
  // workgroup; the stage-1 partials live behind the rows in `part`, 8-byte aligned)
  auto dp_finish = [&](int CP, int rows_per_brick = 4) {
    const long long rows = p.tiles * rows_per_brick;
    const int mg = p.CoutP / CP;
    double* partial = reinterpret_cast<double*>(dp->part + (rows * mg * CP * 2 + 1) / 2 * 2);
    hipLaunchKernelGGL(dprelu_finish1_kernel, dim3(kFinishBlocks, mg), dim3(256), 0, st, dp->part, (int)rows, CP, partial);
    hipLaunchKernelGGL(dprelu_finish2_kernel, dim3(dp->nslope + Cout), dim3(64), 0, st, partial, CP, Cout, dp->ga, dp->gb,
                       dp->nslope);
    return hipGetLastError() == hipSuccess ? FS_OK : FS_ERR_LAUNCH;
  };
  const int K3 = kernel * kernel * kernel;
  // the 64-channel k3 layers of the 64^3 trunk: 1-D Winograd F(2,3) along x (convwino.hpp), its own filter slab
  p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  // ... F(2,3) along y x F(4,3) along x (convwino2d.hpp: a third of the direct form's multiply-adds) on the 64^3 trunk
  if (wino2d_ok(p, x, ws, Cin, Cout, kernel, stride, ms != nullptr) &&
      (dp == nullptr || (bias == nullptr && z == nullptr && addend == nullptr))) {
    wprep_do(wprep_job(FS_WPREP_WINO2D, w, ws, (long long)cinp * FS_WINO2D_UCH, Cout, Cin, cinp, wmode), plan, st);
    if (plan != nullptr) return FS_OK;
    const int rc = launch_wino2d(x, ws, bias, y, p, st);
    if (rc != FS_OK || dp == nullptr) return rc;
    return dp_finish(64, wino2d_part_rows());
  }
#ifdef FS_ABLATION
  // ... F(4,3) (convwino4.hpp: half the direct form's multiply-adds) where its 4 x 2 x 64 bricks fill the chip
  if (wino4_ok(p, x, ws, Cin, Cout, kernel, stride, ms != nullptr) &&
      (dp == nullptr || (bias == nullptr && z == nullptr && addend == nullptr))) {
    wprep_do(wprep_job(FS_WPREP_WINO4, w, ws, (long long)cinp * FS_WINO4_UCH, Cout, Cin, cinp, wmode), plan, st);
    if (plan != nullptr) return FS_OK;
    const int rc = launch_wino4(x, ws, bias, y, p, st);
    if (rc != FS_OK || dp == nullptr) return rc;
    return dp_finish(64);
  }
  if (wino_ok(p, x, ws, Cin, Cout, kernel, stride, ms != nullptr) &&
      (dp == nullptr || (bias == nullptr && z == nullptr && addend == nullptr))) {
    wprep_do(wprep_job(FS_WPREP_WINO, w, ws, (long long)cinp * FS_WINO_UCH, Cout, Cin, cinp, wmode), plan, st);
    if (plan != nullptr) return FS_OK;
    const int rc = launch_wino(x, ws, bias, y, p, st);
    if (rc != FS_OK || dp == nullptr) return rc;
    return dp_finish(64);
  }
#endif
  // round 5: the k = 4 stride-2 layers with enough bricks run with fp32 accuracy on the bf16 matrix rate (three bf16
  // pieces per operand, six products, fp32 accumulation: convfwd_s3.hpp): 16-byte rows (Wi % 4 == 0), pad 1, 31-bit byte
  // offsets inside one channel, its own pre-split weight slab
  static const bool no_s3 = FS_AB_ENV("FLOWSCI_FWD_NO_S3");
  if (!no_s3 && kernel == 4 && stride == 2 && pad == 1 && wmode == 0 && (p.CoutP == 32 || p.CoutP == 64) && Wi % 4 == 0 &&
      ms_aligned && (((ms ? (uintptr_t)0 : (uintptr_t)x) | (uintptr_t)ws) & 15) == 0 &&
      (long long)Di * Hi * Wi * 4 < (1ll << 31) &&
      (long long)B * Do * fs::cdiv(Ho, S3_TY) * fs::cdiv(Wo, S3_TW) >= 256 &&
      (dp == nullptr || (p.CoutP == 32 && bias == nullptr && z == nullptr && addend == nullptr))) {
    const int cp = p.CoutP;  // one channel group per workgroup
    wprep_do(wprep_job(FS_WPREP_S3K4, w, ws, s3_slab_words(Cin, p.CoutP), Cout, Cin, (Cin + 1) / 2, cp, 0, 0), plan, st);
    if (plan != nullptr) return FS_OK;
    // 8 matrix waves (two per SIMD) + 4 loader waves: 8 loaders measured 3-5 % slower on every layer (and spill at 64
    // output channels); the ablation build keeps them behind FLOWSCI_S3_NLW=8
    int rc;
#ifdef FS_ABLATION
    static const int s3_nlw = (int)FS_AB_ENV_LL("FLOWSCI_S3_NLW", 4);
    if (s3_nlw == 8) rc = p.CoutP == 32 ? launch_s3<1, S3_NMW, 8>(x, ws, bias, y, p, st) : launch_s3<2, S3_NMW, 8>(x, ws, bias, y, p, st);
    else
#endif
    rc = p.CoutP == 32 ? launch_s3<1, S3_NMW, 4>(x, ws, bias, y, p, st) : launch_s3<2, S3_NMW, 4>(x, ws, bias, y, p, st);
    if (rc != FS_OK || dp == nullptr) return rc;
    return dp_finish(32, S3_NMW);
  }
  wprep_do(wprep_job(FS_WPREP_FWD, w, ws, (long long)cinp * K3 * p.CoutP, Cout, Cin, K3, cinp, p.CoutP, wmode), plan, st);
  if (plan != nullptr) return FS_OK;  // every kernel below reads the same slab
  // loader-wave kernels (k = 4: 100 / 129 TFLOP/s vs 92 / 120 for conv0a / conv0b at 256^3; k = 3: every 32-column
  // layer, see below): 16-byte
  // pieces (Wi % 4 == 0, 16-byte aligned input and workspace), 31-bit byte offsets inside one staged channel
  // chunk, enough bricks to fill the chip with one workgroup per CU
  static const bool reg_only = FS_AB_ENV("FLOWSCI_FWD_REG");
  const bool ws_ok = !reg_only && Wi % 4 == 0 && Wo > 16 && (((ms ? (uintptr_t)0 : (uintptr_t)x) | (uintptr_t)ws) & 15) == 0 &&
                     ms_aligned && (long long)(kernel == 3 ? 4 : 2) * Di * Hi * Wi * 4 < (1ll << 31);
  // only the 32-channel loader-wave kernel reads per-channel planes (IFBlock's conv0[0] at scale 1)
  if (ms != nullptr && !(kernel == 4 && ws_ok && dp == nullptr)) return FS_ERR_UNSUPPORTED;
  if (kernel == 3) {
    // big bricks (2 x 8 x 32 / 2 x 16 x 16 voxels) when they fill the chip, else quarter-size bricks
    // (1 x 4 x 32 / 1 x 8 x 16): the 16^3 / 32^3 trunk layers of the coarse blocks have only 8K-64K
    // output voxels per launch
    const bool wide = Wo > 16;
    const long long big = (long long)B * fs::cdiv(Do, 2) * fs::cdiv(Ho, wide ? 8 : 16) * fs::cdiv(Wo, wide ? 32 : 16) *
                          (p.CoutP / 64);
    // every 32-column layer and the 16-column layers of block0 take the loader-wave form (16-byte pieces: Wi % 4 == 0,
    // aligned input and workspace, 31-bit byte offsets inside a staged chunk): 64 -> 64 at 64^3 0.886 -> 0.860 ms =
    // 135 TFLOP/s since its epilogue stores 16 bytes per lane (the two forms tied at 132 before); 64 -> 64 at 32^3
    // 0.145 -> 0.125 ms; 128 -> 128 at 16^3 0.107 -> 0.068 ms -- with about one workgroup per CU, eight waves that
    // overlap staging and MFMAs beat two independent four-wave workgroups
    const bool ws3 = !reg_only && Wi % 4 == 0 && (((ms ? (uintptr_t)0 : (uintptr_t)x) | (uintptr_t)ws) & 15) == 0 &&
                     (long long)4 * Di * Hi * Wi * 4 < (1ll << 31);
    // the fused PReLU-backward epilogue (dp) exists in the loader-wave kernels only: the caller falls back otherwise
    if (dp != nullptr && (bias != nullptr || z != nullptr || addend != nullptr || !ws3)) return FS_ERR_UNSUPPORTED;
    int rc = FS_ERR_UNSUPPORTED, cp = 0;
    const long long small = (long long)B * Do * fs::cdiv(Ho, wide ? 4 : 8) * fs::cdiv(Wo, wide ? 32 : 16) * (p.CoutP / 64);
    if (big >= 512) {
      if (ws3 && wide) { rc = launch_ws<3, 1, 4, 2, 4, 2, 8, 32>(x, ws, bias, y, p, st); cp = 64; }
      else if (dp != nullptr) return FS_ERR_UNSUPPORTED;
      else if (wide) return launch<3, 1, 4, 2, 4, 2, 8, 32>(x, ws, bias, y, p, st);
      else return launch<3, 1, 4, 2, 4, 2, 8, 16>(x, ws, bias, y, p, st);
    } else if (small < 256) {
      // fewer than one 64-channel workgroup per CU (block0's 128-channel layers at 16^3: 64 bricks x 2): 32 output
      // channels per workgroup instead -- twice the workgroups, half the serial MFMA chain of each
      if (ws3 && !wide && Cin % 8 == 0) { rc = launch_ws<3, 1, 8, 1, 1, 1, 4, 16>(x, ws, bias, y, p, st); cp = 32; }
      else if (dp != nullptr) return FS_ERR_UNSUPPORTED;
      else if (!wide && Cin % 8 == 0) return launch<3, 1, 8, 1, 1, 1, 4, 16>(x, ws, bias, y, p, st);  // 8-channel chunks
      else if (wide) return launch<3, 1, 4, 1, 1, 1, 4, 32>(x, ws, bias, y, p, st);
      else return launch<3, 1, 4, 1, 1, 1, 4, 16>(x, ws, bias, y, p, st);
    } else {
      if (ws3 && wide) { rc = launch_ws<3, 1, 4, 2, 1, 1, 4, 32>(x, ws, bias, y, p, st); cp = 64; }
      else if (dp != nullptr) return FS_ERR_UNSUPPORTED;
      else if (wide) return launch<3, 1, 4, 2, 1, 1, 4, 32>(x, ws, bias, y, p, st);
      else return launch<3, 1, 4, 2, 1, 1, 4, 16>(x, ws, bias, y, p, st);
    }
    if (rc != FS_OK || dp == nullptr) return rc;
    return dp_finish(cp);
  }
  const long long k4tiles = (long long)B * fs::cdiv(Do, 2) * fs::cdiv(Ho, 8) * fs::cdiv(Wo, 32);
  if (dp != nullptr) {
    // fused PReLU backward: only the 32-channel loader-wave kernel has that epilogue (the input gradient of the heads'
    // second deconvolution); everything else takes the unfused entry points
    if (!(kernel == 4 && p.CoutP == 32 && ws_ok && k4tiles >= 512 && bias == nullptr && z == nullptr && addend == nullptr))
      return FS_ERR_UNSUPPORTED;
    const int rc = launch_ws<4, 2, 2, 1, 4, 2, 8, 32>(x, ws, bias, y, p, st);
    if (rc != FS_OK) return rc;
    return dp_finish(32);
  }
  if (ms != nullptr && !(p.CoutP == 32 && k4tiles >= 512)) return FS_ERR_UNSUPPORTED;
  if (p.CoutP == 32) {
    if (ws_ok && k4tiles >= 512) return launch_ws<4, 2, 2, 1, 4, 2, 8, 32>(x, ws, bias, y, p, st);
    if (Wo > 16) return launch<4, 2, 2, 1, 2, 1, 8, 32>(x, ws, bias, y, p, st);
    return launch<4, 2, 2, 1, 2, 1, 8, 16>(x, ws, bias, y, p, st);
  }
  if (ws_ok && 2 * k4tiles * (p.CoutP / 64) >= 512) return launch_ws<4, 2, 2, 2, 2, 1, 8, 32>(x, ws, bias, y, p, st);
  // coarse blocks (block1's conv0[1] at 32^3, block0's at 16^3 and the input gradients of their heads): 2 x 8 x 32
  // bricks are 128 / 32 workgroups for 256 CUs -- quarter-size bricks, and 32-channel workgroups below one per CU
  // (64 -> 128 at 32^3 -> 16^3: 0.29 ms at 28 TFLOP/s with 32 workgroups)
  const bool wide = Wo > 16;
  const long long big = (long long)B * fs::cdiv(Do, 2) * fs::cdiv(Ho, wide ? 8 : 16) * fs::cdiv(Wo, wide ? 32 : 16) * (p.CoutP / 64);
  if (big >= 256) {
    if (wide) return launch<4, 2, 2, 2, 2, 1, 8, 32>(x, ws, bias, y, p, st);
    return launch<4, 2, 2, 2, 2, 1, 8, 16>(x, ws, bias, y, p, st);
  }
  const long long small = (long long)B * Do * fs::cdiv(Ho, wide ? 4 : 8) * fs::cdiv(Wo, wide ? 32 : 16) * (p.CoutP / 64);
  // (loader-wave form here too: 64 -> 128 at 32^3 0.111 -> 0.081 ms, 32 -> 64 at 64^3 0.164 -> 0.151 ms)
  const bool ws4 = !reg_only && Wi % 4 == 0 && (((uintptr_t)x | (uintptr_t)ws) & 15) == 0 && ms == nullptr &&
                   (long long)2 * Di * Hi * Wi * 4 < (1ll << 31);
  if (small >= 256) {
    if (ws4 && wide) return launch_ws<4, 2, 2, 2, 1, 1, 4, 32>(x, ws, bias, y, p, st);
    if (ws4) return launch_ws<4, 2, 2, 2, 1, 1, 4, 16>(x, ws, bias, y, p, st);
    if (wide) return launch<4, 2, 2, 2, 1, 1, 4, 32>(x, ws, bias, y, p, st);
    return launch<4, 2, 2, 2, 1, 1, 4, 16>(x, ws, bias, y, p, st);
  }
  if (ws4 && wide) return launch_ws<4, 2, 2, 1, 1, 1, 4, 32>(x, ws, bias, y, p, st);
  if (ws4) return launch_ws<4, 2, 2, 1, 1, 1, 4, 16>(x, ws, bias, y, p, st);
  if (wide) return launch<4, 2, 2, 1, 1, 1, 4, 32>(x, ws, bias, y, p, st);
  return launch<4, 2, 2, 1, 1, 1, 4, 16>(x, ws, bias, y, p, st);
}

// The re-layout job fs_conv3d_fwd* runs for this call (the arguments of fs_conv3d_fwd; `x` is only inspected for its
// alignment): conv3d_fwd_impl's own dispatch with `plan` set -- nothing is launched.
extern "C" int fs_conv3d_fwd_wprep_jobs(FsWprepJob* jobs_host, int cap, const float* x, const float* w, float* ws, int B,
                                        int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int kernel,
                                        int stride, int pad, int wmode) {
  FS_ENTER();
  if (jobs_host == nullptr || w == nullptr || ws == nullptr) return -FS_ERR_NULLPTR;
  if (cap < 0) return -FS_ERR_ARG;
  WprepPlan plan = {jobs_host, cap, 0};
  const int rc = conv3d_fwd_impl(x, w, nullptr, nullptr, 0, nullptr, nullptr, nullptr, ws, B, Cin, Cout, Di, Hi, Wi, Do, Ho,
                                 Wo, kernel, stride, pad, wmode, nullptr, nullptr, nullptr, &plan);
  return rc == FS_OK ? plan.n : -rc;
}

extern "C" int fs_conv3d_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int B,
                             int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int kernel,
                             int stride, int pad, int wmode, fs_stream_t stream) {
  FS_ENTER();
  return conv3d_fwd_impl(x, w, bias, nullptr, 0, nullptr, y, nullptr, ws, B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo,
                         kernel, stride, pad, wmode, stream);
}

extern "C" int fs_conv3d_fwd_add(const float* x, const float* w, const float* bias, const float* addend, float* y,
                                 float* ws, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho,
                                 int Wo, int kernel, int stride, int pad, int wmode, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(addend);
  return conv3d_fwd_impl(x, w, bias, nullptr, 0, addend, y, nullptr, ws, B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo,
                         kernel, stride, pad, wmode, stream);
}

extern "C" int fs_conv3d_fwd_prelu(const float* x, const float* w, const float* bias, const float* prelu_weight,
                                   const float* residual, float* y, float* z, float* ws, int B, int Cin, int Cout,
                                   int Di, int Hi, int Wi, int Do, int Ho, int Wo, int kernel, int stride, int pad,
                                   int num_prelu_weights, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(prelu_weight); FS_REQUIRE_PTR(z);
  return conv3d_fwd_impl(x, w, bias, prelu_weight, num_prelu_weights, residual, y, z, ws, B, Cin, Cout, Di, Hi, Wi,
                         Do, Ho, Wo, kernel, stride, pad, 0, stream);
}

// fs_conv3d_fwd_prelu over an input that is never concatenated: channel c of the [B, Cin, D,H,W] input is the plane
// src[c] (sample 0) of a tensor with batch stride batch_strides[c] floats (host arrays, read at launch; Cin <= 12,
// 16-byte aligned planes, strides multiples of 4).  FS_ERR_UNSUPPORTED when the shape has no loader-wave kernel
// (the caller then concatenates and takes fs_conv3d_fwd_prelu).
extern "C" int fs_conv3d_fwd_prelu_ms(const float* const* src, const long long* batch_strides, const float* w,
                                      const float* bias, const float* prelu_weight, float* y, float* z, float* ws,
                                      int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                      int kernel, int stride, int pad, int num_prelu_weights, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(src); FS_REQUIRE_PTR(batch_strides); FS_REQUIRE_PTR(prelu_weight); FS_REQUIRE_PTR(z);
  const MultiSrc ms = {src, batch_strides};
  return conv3d_fwd_impl(nullptr, w, bias, prelu_weight, num_prelu_weights, nullptr, y, z, ws, B, Cin, Cout, Di, Hi, Wi,
                         Do, Ho, Wo, kernel, stride, pad, 0, stream, nullptr, &ms);
}

// Input gradient of a convolution whose INPUT was z = prelu(act_y) (a ConvTranspose3d(4,2,1) read as the strided
// convolution of its grad_out, wmode 0): writes grad_act_y = conv(x) * prelu'(act_y) instead of the gradient w.r.t. z,
// and the PReLU weight gradient / the producing layer's bias gradient (per-wave partials in `part`, finished in a
// fixed order).  FS_ERR_UNSUPPORTED when the shape / alignment has no such kernel: use fs_conv3d_fwd + fs_prelu_bwd.
// rows x channels of per-wave partials + the stage-1 partials of the finish, for the brick choice conv3d_fwd_impl makes
// (kernel 4: the 32-channel loader-wave kernel; kernel 3: its three loader-wave variants)
extern "C" long long fs_conv3d_fwd_dprelu_part_floats(int B, int Cout, int Do, int Ho, int Wo) {
  if (B < 1 || Cout < 1 || Cout > 32 || Do < 1 || Ho < 1 || Wo < 1) return -1;
  // bricks of the fp32 loader-wave kernel (2 x 8 x 32) or of the split-bf16 kernel (1 x 16 x 32), whichever are more
  const long long t32 = (long long)B * fs::cdiv(Do, 2) * fs::cdiv(Ho, 8) * fs::cdiv(Wo, 32);
  const long long ts3 = (long long)B * Do * fs::cdiv(Ho, S3_TY) * fs::cdiv(Wo, S3_TW);
  const long long rows = t32 * 4 > ts3 * S3_NMW ? t32 * 4 : ts3 * S3_NMW;  // one row of partial sums per matrix wave
  return rows * 32 * 2 + 2 + kFinishBlocks * 64 * 2;
}
extern "C" long long fs_conv3d_fwd_dprelu_part_floats_k3(int B, int Cout, int Do, int Ho, int Wo) {
  if (B < 1 || Cout < 1 || Do < 1 || Ho < 1 || Wo < 1) return -1;
  const long long coutp = (Cout + 63) / 64 * 64;
  const bool wide = Wo > 16;
  const long long mg64 = coutp / 64;
  const long long big = (long long)B * fs::cdiv(Do, 2) * fs::cdiv(Ho, wide ? 8 : 16) * fs::cdiv(Wo, wide ? 32 : 16);
  const long long small = (long long)B * Do * fs::cdiv(Ho, wide ? 4 : 8) * fs::cdiv(Wo, wide ? 32 : 16);
  long long tiles = big * mg64 >= 512 ? big : small;  // bricks per channel group
  // the Winograd kernel's bricks (2 x 2 x 64) where it applies: twice as many as the direct kernel's big bricks
  const long long wino = (long long)B * fs::cdiv(Do, 2) * fs::cdiv(Ho, 2) * fs::cdiv(Wo, 64);
  if (coutp == 64 && wino > tiles) tiles = wino;
  // rows = 4 per brick and channel group; a row holds the group's channels x 2: coutp x 2 floats per brick row in total
  return tiles * 4 * coutp * 2 + 2 + (long long)kFinishBlocks * coutp * 2 * 2;
}

extern "C" int fs_conv3d_fwd_dprelu(const float* x, const float* w, const float* act_y, const float* prelu_weight,
                                    int num_prelu_weights, float* grad_act_y, float* grad_prelu_weight,
                                    float* grad_bias, float* part, float* ws, int B, int Cin, int Cout, int Di, int Hi,
                                    int Wi, int Do, int Ho, int Wo, int kernel, int stride, int pad,
                                    fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(act_y); FS_REQUIRE_PTR(prelu_weight); FS_REQUIRE_PTR(grad_act_y); FS_REQUIRE_PTR(grad_prelu_weight);
  FS_REQUIRE_PTR(grad_bias); FS_REQUIRE_PTR(part);
  if (num_prelu_weights != 1 && num_prelu_weights != Cout) return FS_ERR_ARG;
  DPrelu dp;
  dp.act_y = act_y; dp.slope = prelu_weight; dp.nslope = num_prelu_weights; dp.part = part;
  dp.ga = grad_prelu_weight; dp.gb = grad_bias;
  // kernel 4: the strided convolution of a ConvTranspose3d's grad_out (wmode 0); kernel 3: the input gradient of a
  // stride-1 'same' Conv3d, w as stored [Cout_conv][Cin_conv][27] read flipped and transposed (wmode 1)
  return conv3d_fwd_impl(x, w, nullptr, nullptr, 0, nullptr, grad_act_y, nullptr, ws, B, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo,
                         kernel, stride, pad, kernel == 3 ? 1 : 0, stream, &dp);
}

#ifdef FS_W2_STAMPS
extern "C" int fs_debug_w2_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs_w2_dbg), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#endif
