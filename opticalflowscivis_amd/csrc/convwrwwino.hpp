// convwrwwino.hpp -- weight gradient of the 64-channel k3 s1 p1 convolutions of the IFNet-3D trunk (24 launches and
// 15.5 of the ~101 ms of the 2 x 256^3 step as the direct implicit GEMM of convwrw.hip, at 0.87 of the fp32 MFMA peak)
// in the 1-D Winograd F(2,3) domain of convwino.hpp: 2/3 of the matrix-core work.  Included inside convwrw.hip's
// anonymous namespace.
//
//   forward (convwino.hpp):  M_t[co, p, j] = sum_{ci, kz, ky} U_t[co, ci, kz, ky] V_t[ci, p + (kz, ky) - 1, j],
//                            y[2j] = M_0 + M_1 + M_2,  y[2j+1] = M_1 - M_2 - M_3,  U = G g,  V = B^T d
//   so with dy = grad_out:   dM = (dy0, dy0 + dy1, dy0 - dy1, -dy1)                                   (A dy)
//                            dU_t[co, ci, kz, ky] = sum_{b, p, j} dM_t[co, p, j] V_t[ci, p + (kz, ky) - 1, j]
//                            dg = (dU_0 + (dU_1 + dU_2)/2, (dU_1 - dU_2)/2, (dU_1 + dU_2)/2 + dU_3)      (G^T dU)
//   Four GEMMs with M = co, N = (ci, kz, ky), K = x-TILES (half as many as outputs): 4 x 9 instead of 27 x 2
//   multiply-adds per (co, ci, output pair) -- 2/3.
//
// Both operands are transformed WHEN THEY ARE READ, not when they are staged: the loaders move the raw gradient and
// source rows into LDS with `buffer_load_dwordx4 ... lds` exactly as the direct kernel does (the transformed
// operands would be twice the size of the raw ones, and LDS capacity is what bounds the brick), and a matrix wave
// forms dM_t / V_t from one 8-byte and one 4-byte LDS read with one VALU operation each.
//
// Decomposition.  A workgroup owns all 64 gradient channels, a half of the source channels (32) and ONE kz (the
// three ky of it = three 32-column tiles: column = ky * 32 + ci), i.e. 6 column groups per K-slab, and walks a slab
// of position bricks (1 z x 2 y x 64 x = 32 x-tiles per row).  Its four matrix waves are (row tile m in {0, 1}) x
// (component pair {t = 0, 1} or {t = 2, 3}): 3 column tiles x 2 components = 6 accumulator tiles (96 VGPRs).  The
// loaders stage, per brick, the gradient rows [64][2 rows x 64] and the source rows z + kz - 1, y - 1 .. y + 2 of the
// 32 channels (with 4 floats of margin left and right: 16-byte pieces, rows outside the volume read 0).
// Epilogue: each wave adds its share of G^T dU into dW with float atomics (as the direct kernel does with its tiles).
constexpr int WW_TY = 2;                        // y rows of a position brick
constexpr int WW_GP = WW_TY * 64 + 4;           // gradient row pitch per channel (+4: spreads the channels over the banks)
constexpr int WW_XP = 72;                       // staged source row: 4 margin + 64 + 4 margin
constexpr int WW_CHS = (WW_TY + 2) * WW_XP + 4; // source channel pitch (+4, as above)
constexpr int WW_NG = 64 * WW_GP;               // 8448 floats
constexpr int WW_NS = 32 * WW_CHS;              // 9344 floats
// an LDS-DMA wave-instruction fills 256 floats: both images are rounded up to that, or the last instruction of the
// source image would write its out-of-range zeros over the first gradient row of the OTHER buffer
constexpr int WW_NSL = (WW_NS + 255) / 256 * 256; // 9472
constexpr int WW_BUF = WW_NG + WW_NSL;          // 17920 floats = 70 KB; two buffers
static_assert(WW_NG % 256 == 0, "the gradient image ends on a wave-instruction boundary");

struct WWP {
  int B, D, H, W;    // one extent: stride 1, pad 1
  int by;            // bricks along y (H / 2)
  long long bricks;  // B * D * by * (W / 64)
  int spw;           // bricks per workgroup
  long long slab;    // deterministic mode: floats per private copy of dW (0: float atomics), see convwrw.hip WDet
};

// The loader waves (4-7) of the Winograd-domain weight-gradient kernels: per position brick, the RAW gradient rows
// [64][2 rows x 64] and the source rows z + kz - 1, y - 1 .. y + 2 of 32 channels (4 floats of margin left and right)
// by `buffer_load_dwordx4 ... lds`, one brick ahead of the matrix waves, one barrier per brick.
template <int DBG>
__device__ __forceinline__ void ww_loader_waves(const float* __restrict__ G, const float* __restrict__ Src, const WWP& p,
                                                float* lds, int wv, int lane, int kz, int c0, long long s0, long long s1) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(3);  // (a loader wave issues a handful of instructions per period: they should not queue behind the matrix wave's)
  constexpr int NGP = (WW_NG / 4 + 63) / 64, NSP = (WW_NS / 4 + 63) / 64;  // 16-byte pieces in wave-instructions
  constexpr int NGW = (NGP + 3) / 4, NSW = (NSP + 3) / 4;
  const size_t vol = (size_t)p.D * p.H * p.W;
  const int bxn = p.W / 64;
  // piece k of loader wave wv fills the 16-byte slots 64 (wv + 4 k) + lane of an image
  unsigned goff[NGW], soff[NSW];
  int srow[NSW], sx[NSW];
#pragma unroll
  for (int k = 0; k < NGW; ++k) {
    const int f = (64 * (wv + 4 * k) + lane) * 4;
    const int co = f / WW_GP, w = f - co * WW_GP;
    const int row = w / 64, x = w % 64;
    goff[k] = (f < WW_NG && w < WW_TY * 64) ? ((unsigned)co * (unsigned)vol + (unsigned)(row * p.W + x)) * 4u : DMA_OOB;
  }
#pragma unroll
  for (int k = 0; k < NSW; ++k) {
    const int f = (64 * (wv + 4 * k) + lane) * 4;
    const int c = f / WW_CHS, r = f - c * WW_CHS;
    const int y = r / WW_XP, x = r - y * WW_XP;  // staged row y = source row oy0 - 1 + y, column ox0 - 4 + x
    const bool ok = f < WW_NS && r < (WW_TY + 2) * WW_XP;
    soff[k] = ok ? ((unsigned)c * (unsigned)vol + (unsigned)(y * p.W + x)) * 4u : DMA_OOB;
    srow[k] = y;
    sx[k] = x;
  }
  int bxi, byi, z, b;
  {
    long long q = s0;
    bxi = (int)(q % bxn); q /= bxn;
    byi = (int)(q % p.by); q /= p.by;
    z = (int)(q % p.D);
    b = (int)(q / p.D);
  }
  auto stage = [&](int buf) {
    if (DBG == 1) return;
    const int oy0 = byi * WW_TY, ox0 = bxi * 64;
    float* dbase = lds + buf * WW_BUF;
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(G + (size_t)b * 64 * vol), (short)0, 0x7fffffff, 0x00020000);
    const unsigned pos0 = (unsigned)(((z * p.H + oy0) * p.W + ox0) * 4);
#pragma unroll
    for (int k = 0; k < NGW; ++k)
      if (wv + 4 * k < NGP)  // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr_t)(dbase + 256 * (wv + 4 * k)), 16, goff[k], pos0, 0, 0);
    // source rows of plane z + kz - 1 (all zero when that plane is outside the volume)
    const int sz = z + kz - 1;
    const bool zok = sz >= 0 && sz < p.D;
    const long long org = ((long long)(zok ? sz : 0) * p.H + (oy0 - 1)) * p.W + (ox0 - 4);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Src + ((size_t)b * 64 + c0) * vol + org), (short)0,
                                                                   zok ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < NSW; ++k)
      if (wv + 4 * k < NSP) {  // wave-uniform
        const int gy = oy0 - 1 + srow[k], gx = ox0 - 4 + sx[k];
        const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;  // W % 64 == 0: a 16-byte piece is in or out whole
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dbase + WW_NG + 256 * (wv + 4 * k)), 16,
                                                 in ? soff[k] : DMA_OOB, 0, 0, 0);
      }
    if (++bxi == bxn) { bxi = 0; if (++byi == p.by) { byi = 0; if (++z == p.D) { z = 0; ++b; } } }
  };
  if (s0 < s1) stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (long long st = s0; st < s1; ++st) {
    if (st + 1 < s1) stage(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }
#endif
}

// DBG (measurement builds, FLOWSCI_WINO_DBG): 1 = the loaders only keep the barrier protocol, 2 = the matrix waves skip
// their operand reads and MFMAs.  Measured at 2 x 64^3 (tests/tools/wino_wrw_bench.py, ~0.04 ms of harness included):
// 0.78 ms as built; matrix waves + epilogue alone 0.79 -> the MFMA work at the kernel's clock is 0.57; loaders +
// epilogue alone 0.26; the direct kernel 0.96.  Letting the compiler interleave operand reads and transforms with the
// MFMAs (no sched_barrier, or sched_group_barrier MFMA / DS / VALU triples) changes nothing.
#ifdef FS_ABLATION  // the F(2,3) kernel: superseded by convwrwwino4.hpp on every shape (measurement builds only)
template <int DBG>
__global__ __launch_bounds__(512, 1) void conv3d_wrw_wino_kernel(const float* __restrict__ G,
                                                                const float* __restrict__ Src,
                                                                float* __restrict__ dW, WWP p) {
  static_assert(2 * WW_BUF * 4 <= 160 * 1024 && (WW_NG % 4) == 0 && (WW_NS % 4) == 0, "two aligned buffers in LDS");
  __shared__ __attribute__((aligned(16))) float lds[2 * WW_BUF];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;
  const int kz = blockIdx.y % 3, chalf = blockIdx.y / 3;  // column group: kz, source-channel half
  const int c0 = chalf * 32;
  const long long s0 = (long long)blockIdx.x * p.spw;
  const long long s1 = min(s0 + (long long)p.spw, p.bricks);

  if (wave >= 4) {
    ww_loader_waves<DBG>(G, Src, p, lds, wv, lane, kz, c0, s0, s1);
    return;
  }

  // ---- matrix waves: (brick row h, component pair tp), both 32-channel row tiles of the gradient.  (First form: waves
  // = (row tile, pair), every wave walking both brick rows: each read all three source tiles -- 7 LDS reads per 6
  // MFMAs, with the 4-way bank conflicts of the 16-byte-aligned pitches the LDS was the bound: 1.28 ms per launch,
  // slower than the direct kernel.  A wave now owns ONE row of the brick and both row tiles: 8 reads per 12 MFMAs.)
  const int l31 = lane & 31, kh = lane >> 5;
  const int h = wv >> 1, tp = wv & 1;
  const int aBo = l31 * WW_GP + h * 64 + 2 * kh;                                  // + m * 32 * GP + 4 kk: (dy[2j], dy[2j+1])
  const int bPo = WW_NG + l31 * WW_CHS + h * WW_XP + 4 + 2 * kh;                  // + n * XP + 4 kk: the aligned pair (d1, d2)
  const int bBo = bPo + ((wv & 1) ? 2 : -1);                                            // d0 (tp 0) or d3 (tp 1)

  f32x16 acc[3][2][2];
  // (the component pair is wave-uniform: two specialised instances of the loop, selected once, instead of selects in
  // front of every MFMA)
  auto kloop = [&](auto TPc) {
  constexpr int tp = decltype(TPc)::value;
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][m][c][r] = 0.f;

  __builtin_amdgcn_s_barrier();  // brick s0 has landed
  int buf = 0;
  for (long long st = s0; st < s1; ++st) {
    const float* base = lds + buf * WW_BUF;
    // reduction step kk: x-tiles 2 kk + kh of this wave's row
    auto lds_ops = [&](int kk, float2 (&a)[2], float2 (&bp)[3], float (&be)[3]) {
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const float2*>(base + aBo + m * 32 * WW_GP + 4 * kk);
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        bp[n] = *reinterpret_cast<const float2*>(base + bPo + n * WW_XP + 4 * kk);
        be[n] = base[bBo + n * WW_XP + 4 * kk];
      }
    };
    auto mma = [&](const float2 (&a)[2], const float2 (&bp)[3], const float (&be)[3]) {
      // dM: t = 0: dy0, t = 1: dy0 + dy1 | t = 2: dy0 - dy1, t = 3: dy1 (its sign is applied in the epilogue)
      float a0[2], a1[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a0[m] = tp ? a[m].x - a[m].y : a[m].x;
        a1[m] = tp ? a[m].y : a[m].x + a[m].y;
      }
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        // V: t = 0: d0 - d2, t = 1: d1 + d2 | t = 2: d2 - d1, t = 3: d1 - d3     (bp = (d1, d2), be = d0 or d3)
        const float v0 = tp ? bp[n].y - bp[n].x : be[n] - bp[n].y;
        const float v1 = tp ? bp[n].x - be[n] : bp[n].x + bp[n].y;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[n][m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], v0, acc[n][m][0], 0, 0, 0);
          acc[n][m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], v1, acc[n][m][1], 0, 0, 0);
        }
      }
    };
    float2 a0[2], a1[2], p0[3], p1[3];
    float e0[3], e1[3];
    if (DBG != 2) lds_ops(0, a0, p0, e0);
#pragma unroll
    for (int q = 0; q < (DBG == 2 ? 0 : 16); q += 2) {
      lds_ops(q + 1, a1, p1, e1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, p0, e0);
      if (q + 2 < 16) lds_ops(q + 2, a0, p0, e0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, p1, e1);
    }
    __builtin_amdgcn_s_barrier();  // the next brick has landed, everyone is done reading `buf`
    buf ^= 1;
  }
  };
  if (wv & 1) kloop(std::integral_constant<int, 1>{}); else kloop(std::integral_constant<int, 0>{});

  // ---- epilogue.  The four waves' shares of G^T dU are first combined in LDS (dg[co][ci][ky, kx], 72 KB of the now idle
  // staging buffers), then added to dW with float atomics whose lanes walk dW's own order (runs of 9 contiguous floats
  // per (co, ci)).  The first form added every wave's share straight to dW, lane = ci: 18.6 M atomics per launch, each
  // wave-instruction touching 64 cache lines (108-byte lane stride) -- device-scope atomics are served at the memory
  // side, and that epilogue alone took 1.0 ms of a 1.6 ms launch (ablation builds).  Now 4.6 M atomics in ~7-line
  // instructions.
  //   tp 0 (dU_0, dU_1):   kx 0 += dU_0 + dU_1 / 2,  kx 1 += dU_1 / 2,  kx 2 += dU_1 / 2
  //   tp 1 (dU_2, dU_3'):  kx 0 += dU_2 / 2,         kx 1 -= dU_2 / 2,  kx 2 += dU_2 / 2 - dU_3'    (dU_3 = -dU_3')
  float* dg = lds;
  constexpr int NDG = 64 * 32 * 9;
  static_assert(NDG <= 2 * WW_BUF, "the combine buffer fits the staging buffers");
  for (int i = t; i < NDG; i += 256) dg[i] = 0.f;   // (the loaders have left: 256 matrix threads, and the barriers
  __builtin_amdgcn_s_barrier();                     //  below count the live waves only)
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float* dst = dg + (co * 32 + l31) * 9 + n * 3;  // lane stride 9 floats: conflict-free
        const float u0 = acc[n][m][0][r], u1 = acc[n][m][1][r];
        if (tp == 0) {
          atomicAdd(dst + 0, u0 + 0.5f * u1);
          atomicAdd(dst + 1, 0.5f * u1);
          atomicAdd(dst + 2, 0.5f * u1);
        } else {
          atomicAdd(dst + 0, 0.5f * u0);
          atomicAdd(dst + 1, -0.5f * u0);
          atomicAdd(dst + 2, 0.5f * u0 - u1);
        }
      }
  __builtin_amdgcn_s_barrier();
  for (int i = t; i < NDG; i += 256) {
    const int co = i / 288, r2 = i - co * 288;
    const int ci = r2 / 9, k9 = r2 - ci * 9;
    atomicAdd(dW + ((size_t)co * 64 + c0 + ci) * 27 + kz * 9 + k9, dg[i]);
  }
}

#endif  // FS_ABLATION

inline bool wrw_wino_ok(const WP& w, const float* g, const float* src, int kernel, int stride) {
  static const bool off = FS_AB_ENV("FLOWSCI_WRW_NO_WINO");
  if (off || kernel != 3 || stride != 1 || w.pad != 1 || w.nsrc != 0) return false;
  if (w.Cg != 64 || w.Cs != 64) return false;
  if (w.Do != w.Di || w.Ho != w.Hi || w.Wo != w.Wi || w.Wo % 64 != 0 || w.Ho % WW_TY != 0) return false;
  if ((((uintptr_t)g | (uintptr_t)src) & 15) != 0) return false;
  if ((long long)64 * w.Do * w.Ho * w.Wo * 4 >= (1ll << 31)) return false;
  // enough bricks for 42 slabs per column group (6 groups: ~one workgroup per CU) of a few bricks each
  return (long long)w.B * w.Do * (w.Ho / WW_TY) * (w.Wo / 64) >= 1024;
}

#ifdef FS_ABLATION
inline int launch_wrw_wino(const float* G, const float* Src, float* dW, const WP& w, hipStream_t st) {
  WWP p;
  p.B = w.B; p.D = w.Do; p.H = w.Ho; p.W = w.Wo;
  p.by = w.Ho / WW_TY;
  p.bricks = (long long)w.B * w.Do * p.by * (w.Wo / 64);
  const long long slabs = 42;  // x 6 column groups = 252 workgroups: one per CU
  long long spw = (p.bricks + slabs - 1) / slabs;
  p.spw = (int)spw;
  const long long gx = (p.bricks + spw - 1) / spw;
  static const int dbg = (int)FS_AB_ENV_LL("FLOWSCI_WINO_DBG", 0);
  if (dbg == 1) hipLaunchKernelGGL(conv3d_wrw_wino_kernel<1>, dim3((unsigned)gx, 6, 1), dim3(512), 0, st, G, Src, dW, p);
  else if (dbg == 2) hipLaunchKernelGGL(conv3d_wrw_wino_kernel<2>, dim3((unsigned)gx, 6, 1), dim3(512), 0, st, G, Src, dW, p);
  else hipLaunchKernelGGL(conv3d_wrw_wino_kernel<0>, dim3((unsigned)gx, 6, 1), dim3(512), 0, st, G, Src, dW, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
#endif  // FS_ABLATION
