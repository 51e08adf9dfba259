// convwrwwino4.hpp -- weight gradient of the 64-channel k3 s1 p1 trunk convolutions in the 1-D Winograd F(4,3) domain
// of convwino4.hpp: HALF the matrix-core work of the direct form (convwrwwino.hpp's F(2,3): two thirds).  Included
// inside convwrw.hip's anonymous namespace, after convwrwwino.hpp (same staging, same loader waves, same grid).
//
//   forward (convwino4.hpp): M_t = sum U_t V_t (t = 0..5),  y = A^T M,  U = G g,  V = B^T d, x-tile = 4 outputs.  So
//     dM = A dy:   dM0 = dy0   dM1 = (dy0 + dy2) + (dy1 + dy3)   dM2 = (dy0 + dy2) - (dy1 + dy3)
//                  dM3 = (dy0 + 4 dy2) + (2 dy1 + 8 dy3)   dM4 = (dy0 + 4 dy2) - (2 dy1 + 8 dy3)   dM5 = dy3
//     dU_t[co, ci, kz, ky] = sum_{b, p, j} dM_t[co, p, j] V_t[ci, p + (kz, ky) - 1, j]
//     dg = G^T dU: dg0 = dU0/4 - (dU1 + dU2)/6 + (dU3 + dU4)/24      dg1 = (dU2 - dU1)/6 + (dU3 - dU4)/12
//                  dg2 = -(dU1 + dU2)/6 + (dU3 + dU4)/6 + dU5
//   Six GEMMs with K = x-TILES (a quarter as many as outputs): 6 x 9 instead of 27 x 4 multiply-adds per (co, ci, four
//   outputs).  fp32 rounding against fp64: ~3x the direct kernel's in the mean (tests/test_gpu_wino.py).
//
// Both operands are transformed when they are READ from LDS (16-byte reads at channel pitches of an odd number of
// 16-byte slots: conflict-free), as in convwrwwino.hpp.  The 36 accumulator tiles (2 row tiles x 3 ky x 6 components)
// of a workgroup's (64 co x 32 ci x one kz) share are dealt to its four matrix waves as (row tile m) x (component
// triple {0,1,2} / {3,4,5}): 9 tiles (144 VGPRs) per wave, every wave walks both rows of the 1 x 2 x 64 brick --
// 16 steps of 9 MFMAs, one 16-byte gradient read and three (16 + 4)-byte source reads per step.
template <int DBG>
__global__ __launch_bounds__(512, 1) void conv3d_wrw_wino4_kernel(const float* __restrict__ G,
                                                                 const float* __restrict__ Src,
                                                                 float* __restrict__ dW, WWP p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * WW_BUF];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;
  // (run of bricks, column group) of this workgroup.  The six column groups of a run read the SAME gradient / source bricks;
  // dispatched as (blockIdx.x, blockIdx.y) they landed on four XCDs (linear id % 8) and each XCD's L2 fetched the bricks
  // for itself: 3.4x the algorithmic bytes from HBM (profiles/r04_pmc_traffic.json).  So the linear id is re-read as
  // (XCD, slot) and an XCD takes a contiguous range of (run, group) tasks: the groups of a run share one L2.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int total = gridDim.x * 6, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int c = lin & 7, j = lin >> 3;
    const int q8 = total >> 3, r8 = total & 7;                 // XCD c holds q8 + (c < r8) workgroups
    const int task = c * q8 + (c < r8 ? c : r8) + j;           // its tasks: a contiguous range, group-major inside a run
    bx = task / 6; by = task - bx * 6;
  }
  const int kz = by % 3, chalf = by / 3;  // column group: kz, source-channel half
  const int c0 = chalf * 32;
  const long long s0 = (long long)bx * p.spw;
  const long long s1 = min(s0 + (long long)p.spw, p.bricks);

  if (wave >= 4) {
    ww_loader_waves<DBG>(G, Src, p, lds, wv, lane, kz, c0, s0, s1);
    return;
  }

  const int l31 = lane & 31, kh = lane >> 5;
  const int m = wv >> 1;
  const int aBo = (m * 32 + l31) * WW_GP + 4 * kh;             // + h * 64 + 8 kk: dy[4j .. 4j + 3], x-tile j = 2 kk + kh
  const int bPo = WW_NG + l31 * WW_CHS + 4 + 4 * kh;           // + (h + ky) * XP + 8 kk: (d1 .. d4)
  const int bEo = bPo + ((wv & 1) ? 4 : -1);                   // d5 (triple 1) or d0 (triple 0)

  f32x16 acc[3][3];  // [ky][component of the triple]
  auto kloop = [&](auto C3c) {
    constexpr int c3 = decltype(C3c)::value;
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][c][r] = 0.f;

    __builtin_amdgcn_s_barrier();  // brick s0 has landed
    int buf = 0;
    for (long long st = s0; st < s1; ++st) {
      const float* base = lds + buf * WW_BUF;
      // reduction step s = 8 h + kk: x-tiles 2 kk + kh of brick row h
      auto lds_ops = [&](int s, float4& a, float4 (&bp)[3], float (&be)[3]) {
        const int h = s >> 3, kk = s & 7;
        a = *reinterpret_cast<const float4*>(base + aBo + h * 64 + 8 * kk);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          bp[n] = *reinterpret_cast<const float4*>(base + bPo + (h + n) * WW_XP + 8 * kk);
          be[n] = base[bEo + (h + n) * WW_XP + 8 * kk];
        }
      };
      auto mma = [&](const float4& a, const float4 (&bp)[3], const float (&be)[3]) {
        float am[3];
        if (c3 == 0) {
          const float s02 = a.x + a.z, s13 = a.y + a.w;
          am[0] = a.x; am[1] = s02 + s13; am[2] = s02 - s13;
        } else {
          const float e = fmaf(4.f, a.z, a.x), o = fmaf(8.f, a.w, 2.f * a.y);
          am[0] = e + o; am[1] = e - o; am[2] = a.w;
        }
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          const float d1 = bp[n].x, d2 = bp[n].y, d3 = bp[n].z, d4 = bp[n].w;
          float v[3];
          if (c3 == 0) {
            v[0] = fmaf(4.f, be[n], fmaf(-5.f, d2, d4));
            v[1] = fmaf(-4.f, d1 + d2, d3 + d4);
            v[2] = fmaf(4.f, d1 - d2, d4 - d3);
          } else {
            const float p31 = d3 - d1, r42 = d4 - d2;
            v[0] = fmaf(2.f, p31, r42);
            v[1] = fmaf(-2.f, p31, r42);
            v[2] = fmaf(4.f, d1, fmaf(-5.f, d3, be[n]));
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[n][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[c], v[c], acc[n][c], 0, 0, 0);
        }
      };
      float4 a0, a1, p0[3], p1[3];
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

  // ---- epilogue.  G^T dU of the two component triples is combined in LDS (dg[co][ci][ky, kx], 72 KB of the now idle
  // staging buffers: the triple-0 waves store, the triple-1 waves add), then added to dW with float atomics whose lanes
  // walk dW's own order (see convwrwwino.hpp).
  float* dg = lds;
  constexpr int NDG = 64 * 32 * 9;
  static_assert(NDG <= 2 * WW_BUF, "the combine buffer fits the staging buffers");
  const bool second = (wv & 1) != 0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if ((pass == 1) == second) {  // wave-uniform
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          float* dst = dg + (co * 32 + l31) * 9 + n * 3;  // lane stride 9 floats: conflict-free
          const float u0 = acc[n][0][r], u1 = acc[n][1][r], u2 = acc[n][2][r];
          if (!second) {  // (dU0, dU1, dU2)
            const float s12 = u1 + u2;
            dst[0] = fmaf(0.25f, u0, (-1.f / 6.f) * s12);
            dst[1] = (1.f / 6.f) * (u2 - u1);
            dst[2] = (-1.f / 6.f) * s12;
          } else {        // (dU3, dU4, dU5)
            const float s34 = u0 + u1;
            dst[0] += (1.f / 24.f) * s34;
            dst[1] += (1.f / 12.f) * (u0 - u1);
            dst[2] += fmaf(1.f / 6.f, s34, u2);
          }
        }
    }
    __builtin_amdgcn_s_barrier();  // (the loaders have left: the barrier counts the live waves only)
  }
  for (int i = t; i < NDG; i += 256) {
    const int co = i / 288, r2 = i - co * 288;
    const int ci = r2 / 9, k9 = r2 - ci * 9;
    float* q = dW + (size_t)bx * p.slab + ((size_t)co * 64 + c0 + ci) * 27 + kz * 9 + k9;
    if (p.slab) *q = dg[i]; else atomicAdd(q, dg[i]);
  }
}

inline bool wrw_wino4_ok(const WP& w, const float* g, const float* src, int kernel, int stride) {
  static const bool off = FS_AB_ENV("FLOWSCI_WRW_NO_WINO4");
  return !off && wrw_wino_ok(w, g, src, kernel, stride);  // same shapes, same bricks
}

inline int launch_wrw_wino4(const float* G, const float* Src, float* dW, const WP& w, hipStream_t st, const WDet* det = nullptr) {
  WWP p;
  p.B = w.B; p.D = w.Do; p.H = w.Ho; p.W = w.Wo;
  p.by = w.Ho / WW_TY;
  p.bricks = (long long)w.B * w.Do * p.by * (w.Wo / 64);
  const long long slabs = 42;  // x 6 column groups = 252 workgroups: one per CU
  long long spw = (p.bricks + slabs - 1) / slabs;
  p.spw = (int)spw;
  const long long gx = (p.bricks + spw - 1) / spw;
  float* out;
  const long long dwf = 64ll * 64 * 27;
  const int drc = wrw_det_begin(det, gx, dwf, dW, &out, &p.slab);
  if (drc >= 0) return drc;
  float* const real = dW;
  dW = out;
#ifdef FS_ABLATION  // instantiations that SKIP work (wrong results by design): measurement builds only
  static const int dbg = (int)FS_AB_ENV_LL("FLOWSCI_WINO_DBG", 0);
  if (dbg == 1) hipLaunchKernelGGL(conv3d_wrw_wino4_kernel<1>, dim3((unsigned)gx, 6, 1), dim3(512), 0, st, G, Src, dW, p);
  else if (dbg == 2) hipLaunchKernelGGL(conv3d_wrw_wino4_kernel<2>, dim3((unsigned)gx, 6, 1), dim3(512), 0, st, G, Src, dW, p);
  else
#endif
    hipLaunchKernelGGL(conv3d_wrw_wino4_kernel<0>, dim3((unsigned)gx, 6, 1), dim3(512), 0, st, G, Src, dW, p);
  wrw_det_end(det, gx, dwf, real, st);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
