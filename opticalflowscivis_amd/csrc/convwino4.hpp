// convwino4.hpp -- the 64-channel k3 s1 "same" convolutions of the IFNet-3D trunk with a 1-D Winograd F(4,3) transform
// along x: 6 multiplications for every 4 outputs and 3 taps instead of 12, i.e. HALF the matrix-core work of the direct
// form (convwino.hpp's F(2,3): two thirds).  Included inside convfwd.hip's anonymous namespace, after convwino.hpp.
//
//   x-tile j = outputs x = 4j .. 4j+3; inputs d_i = in[4j - 1 + i], i = 0..5; taps g = w[.., kz, ky, 0..2]
//   (interpolation points 0, +-1, +-2, inf):
//     V = B^T d:  V0 = 4 d0 - 5 d2 + d4          V1 = -4 (d1 + d2) + (d3 + d4)    V2 = 4 (d1 - d2) + (d4 - d3)
//                 V3 = 2 (d3 - d1) + (d4 - d2)   V4 = -2 (d3 - d1) + (d4 - d2)     V5 = 4 d1 - 5 d3 + d5
//     U = G g:    U0 = g0 / 4   U1 = -(g0 + g1 + g2) / 6   U2 = -(g0 - g1 + g2) / 6
//                 U3 = g0 / 24 + g1 / 12 + g2 / 6   U4 = g0 / 24 - g1 / 12 + g2 / 6   U5 = g2      (weight re-layout)
//     M_t[co, z, y, j] = sum_{ci, kz, ky} U_t[co, ci, kz, ky] * V_t[ci, z + kz - 1, y + ky - 1, j]          t = 0..5
//     y = A^T M:  y0 = M0 + (M1 + M2) + (M3 + M4)          y1 = (M1 - M2) + 2 (M3 - M4)
//                 y2 = (M1 + M2) + 4 (M3 + M4)             y3 = (M1 - M2) + 8 (M3 - M4) + M5
//   13.5 Cin multiply-adds per output instead of 27 Cin.  fp32 rounding: the coefficients reach 8 and 1/24; measured
//   against fp64 (tests/test_gpu_wino.py, tests/tools/wino_bench.py) the error is ~2x the direct kernel's in the mean
//   and ~4x in the maximum -- 8e-6 of the output's mean magnitude.
//
// Kernel: loader-wave form.  One 8-wave workgroup per CU owns a brick of 4 z x 2 y rows x 64 x (16 x-tiles) for 64
// output channels; reduction in chunks of CI = 2 input channels (one 32x32x2 MFMA k-pair), two LDS buffers of 48 KB.
//   loader waves 4-7: the U slab of the chunk (2 x 9 x 6 x 64 floats = 27 LDS-DMA wave-instructions); the 6 x 4 staged
//     input rows of both channels as 12 (channel, z row) units of 4 y rows x 16 lanes, three per wave: one float4 per
//     lane = one x-tile, the two neighbouring columns through DPP row shifts (halo columns by a dword load in lanes
//     0 / 15), transformed in registers, written as V[ci][z row][y row][t][x-tile].
//   matrix waves 0-3: one z row each; an MFMA column = (y row, x-tile), 6 transformed accumulator sets x 2 channel
//     tiles = 12 tiles (192 VGPRs); both operands are "lane base + immediate" LDS reads.
//   epilogue: the output transform leaves 4 consecutive x of one channel in a lane -- one 16-byte store, no exchange;
//     same fused epilogues as the direct kernel.
constexpr int W4_RP = 6 * 16 + 16;            // floats per staged row ([t][x-tile]; +16: the two y rows of an MFMA
                                              // operand read land on different halves of the 32 banks)
constexpr int W4_VCH = 24 * W4_RP;            // channel pitch of V: 6 z x 4 y staged rows
constexpr int W4_UCH = FS_WINO4_UCH;          // channel pitch of U: [kz*3+ky][t 0..5][co 0..63]

__global__ __launch_bounds__(512, 1) void conv3d_wino4_ws_kernel(const float* __restrict__ X,
                                                                const float* __restrict__ Ut,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ Y, FP p) {
  constexpr int CI = 2;
  constexpr int NV = CI * W4_VCH, NU = CI * W4_UCH;
  constexpr int NUP = NU / 256;            // LDS-DMA wave-instructions (64 x 16 bytes) of the U slab
  constexpr int NUW = (NUP + 3) / 4;       // per loader wave
  constexpr int BUF = NV + NU;
  static_assert(NU % 256 == 0 && NV % 4 == 0 && 2 * BUF * 4 <= 160 * 1024, "two 16-byte aligned buffers in LDS");
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;

  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);  // contiguous brick ranges per XCD
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int oz0 = tzi * 4, oy0 = tyi * 2, ox0 = txi * 64;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int yr = lane >> 4, q = lane & 15;
    const int gy = oy0 - 1 + yr, gx = ox0 + 4 * q;
    const int hx = q == 0 ? ox0 - 1 : ox0 + 64;
    // units u = wv, wv + 4, wv + 8 of the 12 (channel c = u / 6, staged z row zr = u % 6)
    unsigned voff[3], hoff[3];
    int vdst[3], uc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int u = wv + 4 * k;
      uc[k] = u / 6;
      const int zr = u - 6 * uc[k];
      const int gz = oz0 - 1 + zr;
      const bool rowok = gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
      const unsigned rbase = ((unsigned)(rowok ? gz : 0) * p.Hi + (rowok ? gy : 0)) * p.Wi;
      voff[k] = (rowok && gx < p.Wi) ? (rbase + gx) * 4u : DMA_OOB;   // Wi % 64 == 0: a float4 is in or out whole
      hoff[k] = (rowok && (q == 0 || q == 15) && hx >= 0 && hx < p.Wi) ? (rbase + hx) * 4u : DMA_OOB;
      vdst[k] = uc[k] * W4_VCH + (zr * 4 + yr) * W4_RP + q;
    }
    unsigned uoff[NUW];
#pragma unroll
    for (int k = 0; k < NUW; ++k) uoff[k] = (unsigned)(64 * (wv + 4 * k) + lane) * 16u;
    float xa[3], xb[3], xc[3], xd[3], xh[3];
    auto fetch = [&](int c0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ch = c0 + uc[k];
        const bool live = ch < p.Cin;
        const float* base = X + ((size_t)b * p.Cin + (live ? ch : 0)) * xvol;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, live ? (int)((unsigned)xvol * 4u) : 0, 0x00020000);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff[k], 0, 0);
        xa[k] = __uint_as_float(v[0]); xb[k] = __uint_as_float(v[1]); xc[k] = __uint_as_float(v[2]); xd[k] = __uint_as_float(v[3]);
        xh[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, hoff[k], 0, 0));
      }
    };
    auto dma_u = [&](int c0, int buf) {
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Ut + (size_t)c0 * W4_UCH), (short)0, NU * 4, 0x00020000);
      float* base = lds + buf * BUF + NV;
#pragma unroll
      for (int k = 0; k < NUW; ++k)
        if (wv + 4 * k < NUP)  // wave-uniform
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + 256 * (wv + 4 * k)), 16, uoff[k], 0, 0, 0);
    };
    auto put = [&](int buf) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // d0 = left neighbour's last column (x = 4q - 1), d5 = right neighbour's first (x = 4q + 4); lanes 0 / 15 of a
        // row keep the halo column
        const float d0 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(xh[k]), __float_as_uint(xd[k]), 0x111, 0xF, 0xF, false));
        const float d5 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(xh[k]), __float_as_uint(xa[k]), 0x101, 0xF, 0xF, false));
        const float d1 = xa[k], d2 = xb[k], d3 = xc[k], d4 = xd[k];
        float* dst = lds + buf * BUF + vdst[k];
        const float p31 = d3 - d1, r42 = d4 - d2;
        dst[0 * 16] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        dst[1 * 16] = fmaf(-4.f, d1 + d2, d3 + d4);
        dst[2 * 16] = fmaf(4.f, d1 - d2, d4 - d3);
        dst[3 * 16] = fmaf(2.f, p31, r42);
        dst[4 * 16] = fmaf(-2.f, p31, r42);
        dst[5 * 16] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      }
    };
    fetch(0);
    dma_u(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    put(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int c0 = 0; c0 < p.Cin; c0 += CI) {
      if (c0 + CI < p.Cin) {
        fetch(c0 + CI);
        dma_u(c0 + CI, buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        put(buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // the next chunk is in LDS; the matrix waves are done reading `buf`
      buf ^= 1;
    }
#else
    (void)xvol; (void)NUW;
#endif
    return;
  }

  // ---- matrix waves: wave wv owns z row oz0 + wv; MFMA column = (y row col >> 4, x-tile col & 15)
  const int col = lane & 31, kh = lane >> 5;
  const int yy = col >> 4, tl = col & 15;
  const int bBo = kh * W4_VCH + (wv * 4 + yy) * W4_RP + tl;
  const int aBo = NV + kh * W4_UCH + col;
  constexpr int NP = 54;  // reduction steps per chunk: (kz, ky) x t, one channel pair

  f32x16 acc[6][2];
#pragma unroll
  for (int tt = 0; tt < 6; ++tt)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][m][r] = 0.f;

  __builtin_amdgcn_s_barrier();  // chunk 0 has landed
  int buf = 0;
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
    const float* bB = lds + buf * BUF + bBo;
    const float* aB = lds + buf * BUF + aBo;
    auto lds_ops = [&](int j, float (&a)[2], float& bq) {
      const int kk = j / 6, tt = j - kk * 6;
      a[0] = aB[kk * 384 + tt * 64];
      a[1] = aB[kk * 384 + tt * 64 + 32];
      bq = bB[((kk / 3) * 4 + (kk % 3)) * W4_RP + tt * 16];
    };
    float a0[2], a1[2], b0, b1;
    lds_ops(0, a0, b0);
#pragma unroll
    for (int j = 0; j < NP; j += 2) {
      lds_ops(j + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      acc[j % 6][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0, acc[j % 6][0], 0, 0, 0);
      acc[j % 6][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0, acc[j % 6][1], 0, 0, 0);
      if (j + 2 < NP) lds_ops(j + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      acc[(j + 1) % 6][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1, acc[(j + 1) % 6][0], 0, 0, 0);
      acc[(j + 1) % 6][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1, acc[(j + 1) % 6][1], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }

  // ---- epilogue: output transform (4 consecutive x of one channel per lane and accumulator register), fused epilogues
  const int oz = oz0 + wv, oy = oy0 + yy;
  const int xq = ox0 + 4 * tl;
  const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;
  const bool live = oz < p.Do && oy < p.Ho;
  const size_t orow = ((size_t)(live ? oz : 0) * p.Ho + (live ? oy : 0)) * p.Wo + xq;
  const float* __restrict__ ad = p.addend;
  const float* __restrict__ slope = p.slope;
  const float* __restrict__ dy = p.dy;
  float* __restrict__ Zp = p.Z;
  const float* __restrict__ src = dy != nullptr ? dy : ad;
  float* __restrict__ prow = dy != nullptr ? p.dpart + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 64 * 2 : nullptr;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      float4 v[4], pre[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // addend / act_y of the four channels first: four loads in flight
        const int co = m * 32 + 8 * jb + 4 * kh + i;
        const size_t o = ((size_t)b * p.Cout + (co < p.Cout ? co : 0)) * yvol + orow;
        pre[i] = (src != nullptr && live && co < p.Cout) ? *reinterpret_cast<const float4*>(src + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * jb + i;
        const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r], m5 = acc[5][m][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        v[i] = make_float4((m0 + s12) + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m5);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = m * 32 + 8 * jb + 4 * kh + i;
        const bool ok = live && co < p.Cout;
        const size_t o = ((size_t)b * p.Cout + (co < p.Cout ? co : 0)) * yvol + orow;
        if (dy != nullptr) {
          // fused PReLU backward: g * prelu'(act_y) stored; the wave's sums of the slope and bias gradient terms
          const float sl = p.dslope[p.dnslope == 1 ? 0 : (co < p.Cout ? co : 0)];
          const float g4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
          const float y4[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
          float o4[4], sa = 0.f, sb = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            o4[k] = y4[k] > 0.f ? g4[k] : sl * g4[k];
            sa += y4[k] > 0.f ? 0.f : y4[k] * g4[k];
            sb += o4[k];
          }
          if (ok) *reinterpret_cast<float4*>(Y + o) = make_float4(o4[0], o4[1], o4[2], o4[3]);
          if (!ok) { sa = 0.f; sb = 0.f; }
          // the 32 lanes of a half-wave hold the same channel
#pragma unroll
          for (int sft = 1; sft < 32; sft <<= 1) {
            sa += __shfl_xor(sa, sft);
            sb += __shfl_xor(sb, sft);
          }
          if (col == 0 && co < 64) {
            prow[co * 2] = sa;
            prow[co * 2 + 1] = sb;
          }
        } else if (ok) {
          const float bv = bias != nullptr ? bias[co] : 0.f;
          const float4 w4 = make_float4(v[i].x + bv, v[i].y + bv, v[i].z + bv, v[i].w + bv);
          const float4 av = pre[i];
          if (Zp != nullptr) {
            const float sv = slope[p.nslope == 1 ? 0 : co];
            *reinterpret_cast<float4*>(Y + o) = w4;
            *reinterpret_cast<float4*>(Zp + o) = make_float4((w4.x > 0.f ? w4.x : sv * w4.x) + av.x, (w4.y > 0.f ? w4.y : sv * w4.y) + av.y,
                                                             (w4.z > 0.f ? w4.z : sv * w4.z) + av.z, (w4.w > 0.f ? w4.w : sv * w4.w) + av.w);
          } else {
            *reinterpret_cast<float4*>(Y + o) = make_float4(w4.x + av.x, w4.y + av.y, w4.z + av.z, w4.w + av.w);
          }
        }
      }
    }
}

inline bool wino4_ok(const FP& p, const float* x, const float* ws, int Cin, int Cout, int kernel, int stride, bool has_ms) {
  static const bool off = FS_AB_ENV("FLOWSCI_FWD_NO_WINO4") || FS_AB_ENV("FLOWSCI_FWD_NO_WINO");
  if (off || kernel != 3 || stride != 1 || p.pad != 1 || has_ms) return false;
  if (Cin % 4 != 0 || Cout > 64 || p.CoutP != 64) return false;
  if (p.Wi != p.Wo || p.Wi % 64 != 0 || p.Di != p.Do || p.Hi != p.Ho) return false;
  if ((((uintptr_t)x | (uintptr_t)ws) & 15) != 0) return false;
  if ((long long)p.Di * p.Hi * p.Wi * 4 >= (1ll << 31)) return false;
  // enough bricks for two rounds of one workgroup per CU (the 64^3 trunk of the scale-1 blocks: 2 x 16 x 32 x 1 = 1024)
  static const long long min_bricks = FS_AB_ENV_LL("FLOWSCI_WINO4_MIN", 512);
  return (long long)p.B * fs::cdiv(p.Do, 4) * fs::cdiv(p.Ho, 2) * (p.Wo / 64) >= min_bricks;
}

inline int launch_wino4(const float* X, const float* Ut, const float* bias, float* Y, FP& p, hipStream_t st) {
  p.tz = fs::cdiv(p.Do, 4); p.ty = fs::cdiv(p.Ho, 2); p.tx = p.Wo / 64;
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  if (p.tiles >= (1ll << 31)) return FS_ERR_SHAPE;
  hipLaunchKernelGGL(conv3d_wino4_ws_kernel, dim3((unsigned)p.tiles, 1), dim3(512), 0, st, X, Ut, bias, Y, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
