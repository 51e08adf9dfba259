// convwino2d.hpp -- the 64-channel k3 s1 "same" convolutions of the 64^3 IFNet-3D trunk with a 2-D Winograd transform:
// F(4,3) along x (convwino4.hpp) and F(2,3) along y.  A tile of 2 y x 4 x outputs takes 4 x 6 = 24 products per (ci, kz)
// instead of 72: ONE THIRD of the direct form's matrix-core work (1-D F(4,3): half).  Included inside convfwd.hip's
// anonymous namespace, after convwino4.hpp.
//
//   U[ci][kz][ty][tx][co] = sum_{ky, kx} Gy[ty][ky] Gx[tx][kx] g[co][ci][kz][ky][kx]          (weight re-layout)
//   V[ci][z][ty][tx][j]   = By^T (4 input rows y0-1 .. y0+2)  then  Bx^T (6 input columns 4j-1 .. 4j+4)
//   M[ty][tx][co][z][j]   = sum_{ci, kz} U[ci][kz][ty][tx][co] V[ci][z + kz - 1][ty][tx][j]      24 GEMMs, K = 3 Cin
//   y[co][z][y0 + 0..1][4j + 0..3] = Ay^T Ax^T M
//   By^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3),  Gy = rows (1,0,0), (1,1,1)/2, (1,-1,1)/2, (0,0,1),
//   Ay^T m = (m0 + m1 + m2, m1 - m2 - m3);  the x matrices are convwino4.hpp's.
//
// Kernel: loader-wave form, one 8-wave workgroup per CU, brick = 2 z x 2 y x 64 x (MFMA column = (z row, x-tile)), chunks of
// 2 input channels (one 32x32x2 k-pair; the filter slab is 18 KB per channel), THREE LDS buffers of 49 KB: a chunk lasts
// only ~2 300 MFMA cycles, so the slab is requested two chunks ahead and the input rows one chunk ahead of their transform.
//   loader waves 4-7: the transformed input (a lane owns one x-tile of one staged z row with all four y rows -- 4 float4
//     loads; 16 lanes per z row, the four z rows of the brick in one wave: By^T in registers, the neighbouring columns of
//     the y-transformed rows by DPP row shifts (halo columns by dword loads in lanes 0 / 15), Bx^T, 24 dword writes) and
//     the U slab of the chunk by `buffer_load_dwordx4 ... lds` (36 wave-instructions), both dealt evenly to the four
//     waves and to the periods (see the loader code).
//   matrix waves 0-3: one y-component ty each: 6 x-components x 2 channel tiles = 12 accumulator tiles (192 VGPRs).
//   epilogue: Ax^T in registers (4 consecutive x per lane and channel), the four waves' results meet in LDS (128 KB of
//     the idle staging buffers), every wave finishes 16 channels: Ay^T, the direct kernel's fused epilogues, 16-byte stores.
constexpr int W2_ZP = 4 * 96 + 16;            // floats per staged z row: [ty][tx][x-tile 16] (+16: the two z rows of an MFMA
                                              // operand read land on different halves of the 32 banks)
constexpr int W2_VCH = 4 * W2_ZP;             // channel pitch of V
constexpr int W2_UCH = FS_WINO2D_UCH;         // channel pitch of U: [kz][ty][tx][co] = 3 * 24 * 64

// DBG (measurement builds, FLOWSCI_WINO_DBG): 1 = the U DMA is skipped, 2 = the input waves skip loads and transforms,
// 3 = both (matrix waves + epilogue alone), 4 = the matrix waves skip their operand reads and MFMAs (loaders alone).
// XT = x-tiles per row of a brick: 16 (rows of 64 x, brick 2 z x 2 y x 64 x) or 8 (rows of 32 x: the 16 x-tile slots of a z
// row are then 2 y-tiles x 8 x-tiles, brick 2 z x 4 y x 32 x -- the 32^3 trunk of the scale-2 block).
template <int DBG, int XT>
__global__ __launch_bounds__(512, 1) void conv3d_wino2d_ws_kernel(const float* __restrict__ X,
                                                                 const float* __restrict__ Ut,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ Y, FP p) {
  constexpr int CI = 2;
  constexpr int NV = CI * W2_VCH, NU = CI * W2_UCH;
  constexpr int NUP = NU / 256;            // LDS-DMA wave-instructions (64 x 16 bytes) of the U slab
  constexpr int NUW = NUP / 4;             // per loader wave
  constexpr int BUF = NV + NU;
  constexpr int NEX = 4 * 64 * 32 * 4;     // the epilogue's exchange image: [ty][co][column] float4
  constexpr int NB = 3;                    // staging buffers: the U slab is requested TWO chunks ahead (a chunk is only ~2 300
                                           // MFMA cycles long -- less than a trip to L2 and back)
  constexpr int NLDS = NB * BUF > NEX ? NB * BUF : NEX;
  static_assert(NU % 256 == 0 && NV % 4 == 0 && NLDS * 4 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[NLDS];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;

  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);  // contiguous brick ranges per XCD
  }
  const long long brick = tile;
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  constexpr int YT = 16 / XT;  // y-tiles of a brick
  const int oz0 = tzi * 2, oy0 = tyi * 2 * YT, ox0 = txi * 4 * XT;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    // ---- loader waves.  The vector-ALU work of a loader wave is NOT hidden behind the matrix wave it shares a SIMD with
    // (ablation builds: the input transform alone cost 0.11 of 0.48 ms when two waves did all of it), so it is spread
    // evenly over all four SIMDs and all periods: a wave handles input channel cc = lw & 1 of every second chunk
    // (parity lw >> 1) in two half-steps -- y components 0, 1 in one period, 2, 3 in the next -- and a quarter of
    // every chunk's U slab.  Chunk j is read by the matrix waves in period j and is complete at the end of period j - 1:
    //   period j - 3: its rows are requested (after the wave's previous chunk is finished)
    //   period j - 2: By^T, then Bx^T of ty 0, 1 -> buffer j % 3      period j - 1: Bx^T of ty 2, 3
    const int lw = wave - 4, par = lw >> 1, cc = lw & 1;
    const int zr = lane >> 4, q = lane & 15;  // lane = (staged z row, x-tile)
    const int qx = q & (XT - 1), qy = q / XT;   // x-tile, y-tile of the lane's slot
    const int gz = oz0 - 1 + zr, gx = ox0 + 4 * qx;
    const int hx = qx == 0 ? ox0 - 1 : ox0 + 4 * XT;
    unsigned voff[4], hoff[4];
#pragma unroll
    for (int yr = 0; yr < 4; ++yr) {
      const int gy = oy0 + 2 * qy - 1 + yr;
      const bool rowok = gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
      const unsigned rbase = ((unsigned)(rowok ? gz : 0) * p.Hi + (rowok ? gy : 0)) * p.Wi;
      voff[yr] = (rowok && gx < p.Wi) ? (rbase + gx) * 4u : DMA_OOB;   // Wi % (4 XT) == 0: a float4 is in or out whole
      hoff[yr] = (rowok && (qx == 0 || qx == XT - 1) && hx >= 0 && hx < p.Wi) ? (rbase + hx) * 4u : DMA_OOB;
    }
    const int vdst = cc * W2_VCH + zr * W2_ZP + q;
    unsigned uoff[NUW];
#pragma unroll
    for (int k = 0; k < NUW; ++k) uoff[k] = (unsigned)(64 * (lw + 4 * k) + lane) * 16u;
    auto dma_u = [&](int chunk, int buf) {
      if (DBG == 1 || DBG == 3) return;
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Ut + (size_t)chunk * CI * W2_UCH), (short)0, NU * 4, 0x00020000);
      float* base = lds + buf * BUF + NV;
#pragma unroll
      for (int k = 0; k < NUW; ++k)  // NUP == 4 NUW: every wave issues all of its NUW
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + 256 * (lw + 4 * k)), 16, uoff[k], 0, 0, 0);
    };
    float xr[4][4], xh[4];   // the raw rows in flight
    float yt[4][4], yh[4];   // By^T of the chunk in work (lives across the two half-steps)
    auto fetch = [&](int chunk) {
      if (DBG == 2 || DBG == 3) return;
      const int ch = chunk * CI + cc;
      const bool live = ch < p.Cin;
      const float* base = X + ((size_t)b * p.Cin + (live ? ch : 0)) * xvol;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, live ? (int)((unsigned)xvol * 4u) : 0, 0x00020000);
#pragma unroll
      for (int yr = 0; yr < 4; ++yr) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff[yr], 0, 0);
        xr[yr][0] = __uint_as_float(v[0]); xr[yr][1] = __uint_as_float(v[1]);
        xr[yr][2] = __uint_as_float(v[2]); xr[yr][3] = __uint_as_float(v[3]);
        xh[yr] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, hoff[yr], 0, 0));
      }
    };
    auto ytrans = [&]() {  // By^T over the four rows, column by column (the 4 own columns and the halo column)
      if (DBG == 2 || DBG == 3) return;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        yt[0][i] = xr[0][i] - xr[2][i];
        yt[1][i] = xr[1][i] + xr[2][i];
        yt[2][i] = xr[2][i] - xr[1][i];
        yt[3][i] = xr[1][i] - xr[3][i];
      }
      yh[0] = xh[0] - xh[2]; yh[1] = xh[1] + xh[2]; yh[2] = xh[2] - xh[1]; yh[3] = xh[1] - xh[3];
    };
    auto put_half = [&](int buf, auto HALF) {
      if (DBG == 2 || DBG == 3) return;
      constexpr int half = decltype(HALF)::value;
      float* dstb = lds + buf * BUF + vdst;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const float d1 = yt[2 * half + t2][0], d2 = yt[2 * half + t2][1], d3 = yt[2 * half + t2][2], d4 = yt[2 * half + t2][3];
        const float dh = yh[2 * half + t2];
        // d0 = left neighbour's last column, d5 = right neighbour's first; the first / last x-tile of a row keeps the halo
        // column (lanes 0 / 15 of the DPP row by the shift itself; XT = 8: also the seam between the two y-tiles)
        float d0 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(dh), __float_as_uint(d4), 0x111, 0xF, 0xF, false));
        float d5 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(dh), __float_as_uint(d1), 0x101, 0xF, 0xF, false));
        if (XT < 16) {
          d0 = qx == 0 ? dh : d0;
          d5 = qx == XT - 1 ? dh : d5;
        }
        float* dst = dstb + (2 * half + t2) * 96;
        const float p31 = d3 - d1, r42 = d4 - d2;
        dst[0 * 16] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        dst[1 * 16] = fmaf(-4.f, d1 + d2, d3 + d4);
        dst[2 * 16] = fmaf(4.f, d1 - d2, d4 - d3);
        dst[3 * 16] = fmaf(2.f, p31, r42);
        dst[4 * 16] = fmaf(-2.f, p31, r42);
        dst[5 * 16] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      }
    };
    static_assert(NUP == 4 * NUW, "every loader wave issues NUW slab instructions per chunk");
    const int nch = p.Cin / CI;
    const std::integral_constant<int, 0> H0{};
    const std::integral_constant<int, 1> H1{};
    // prologue = "period -1": chunk 0 whole (parity 0), the first half of chunk 1 (parity 1); slabs of chunks 0 and 1
    dma_u(0, 0);
    if (nch > 1) dma_u(1, 1);
    if (par == 0) {
      fetch(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ytrans();
      put_half(0, H0);
      put_half(0, H1);
      if (nch > 2) fetch(2);
    } else if (nch > 1) {
      fetch(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ytrans();
      put_half(1, H0);
    }
    if (par != 0 || nch <= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the slabs)
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                               // ... all but the 8 row loads of chunk 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t0 = 0; t0 < nch; ++t0) {
      // request this wave's quarter of chunk t0 + 2's slab, then wait for everything it requested in the previous period
      // (a quarter of chunk t0 + 1's slab, maybe rows: they have had a whole period to arrive)
      if (t0 + 2 < nch) {
        dma_u(t0 + 2, (t0 + 2) % NB);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DBG == 1 || DBG == 3 ? 0 : NUW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (((t0 + 1) & 1) == par) {
        if (t0 + 1 < nch) {          // second half of chunk t0 + 1, then its successor's rows
          put_half((t0 + 1) % NB, H1);
          if (t0 + 3 < nch) fetch(t0 + 3);
        }
      } else if (t0 + 2 < nch) {     // first half of chunk t0 + 2
        ytrans();
        put_half((t0 + 2) % NB, H0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // chunk t0 + 1 is in LDS; the matrix waves are done reading chunk t0
    }
#else
    (void)xvol; (void)NUW;
#endif
    return;
  }

  // ---- matrix waves: wave wv owns the y-component ty = wv; MFMA column = (z row col >> 4, x-tile col & 15)
  const int col = lane & 31, kh = lane >> 5;
  const int bBo = kh * W2_VCH + (col >> 4) * W2_ZP + wv * 96 + (col & 15);
  const int aBo = NV + kh * W2_UCH + wv * 384 + col;
  constexpr int NP = 18;  // reduction steps per chunk: kz x tx, one channel pair

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
      const int kz = j / 6, tt = j - kz * 6;
      a[0] = aB[kz * 1536 + tt * 64];
      a[1] = aB[kz * 1536 + tt * 64 + 32];
      bq = bB[kz * W2_ZP + tt * 16];
    };
    float a0[2], a1[2], b0, b1;
    if (DBG != 4) lds_ops(0, a0, b0);
#pragma unroll
    for (int j = 0; j < (DBG == 4 ? 0 : NP); j += 2) {
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
    buf = buf == NB - 1 ? 0 : buf + 1;
  }

  // ---- epilogue 1: Ax^T in registers, this wave's (ty) results to the exchange image ex[ty][co][column] (float4 = 4 x)
  float4* ex = reinterpret_cast<float4*>(lds);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
      const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r], m5 = acc[5][m][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      ex[(wv * 64 + co) * 32 + col] = make_float4((m0 + s12) + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m5);
    }
  __builtin_amdgcn_s_barrier();  // (the loaders have left: the barrier counts the live waves only)

  // ---- epilogue 2: wave wv finishes channels 16 wv .. 16 wv + 15; lane = (y row lane >> 5, z row (lane >> 4) & 1, x-tile)
  const int yy = lane >> 5, c5 = lane & 31;
  const int oz = oz0 + (c5 >> 4), oy = oy0 + 2 * ((c5 & 15) / XT) + yy;
  const int xq = ox0 + 4 * (c5 & (XT - 1));
  const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;
  const bool live = oz < p.Do && oy < p.Ho;
  const size_t orow = ((size_t)(live ? oz : 0) * p.Ho + (live ? oy : 0)) * p.Wo + xq;
  const float* __restrict__ ad = p.addend;
  const float* __restrict__ slope = p.slope;
  const float* __restrict__ dy = p.dy;
  float* __restrict__ Zp = p.Z;
  const float* __restrict__ src = dy != nullptr ? dy : ad;
  float* __restrict__ prow = dy != nullptr ? p.dpart + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 * 2 : nullptr;
  (void)brick;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    float4 v[4], pre[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // addend / act_y of four channels first: four loads in flight
      const int co = 16 * wv + 4 * g4 + i;
      const size_t o = ((size_t)b * p.Cout + (co < p.Cout ? co : 0)) * yvol + orow;
      pre[i] = (src != nullptr && live && co < p.Cout) ? *reinterpret_cast<const float4*>(src + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = 16 * wv + 4 * g4 + i;
      // Ay^T: y row 0 = (m0 + m1) + m2, y row 1 = (m1 - m2) - m3
      const float4 ma = ex[((yy + 0) * 64 + co) * 32 + c5];
      const float4 mb = ex[((yy + 1) * 64 + co) * 32 + c5];
      const float4 mc = ex[((yy + 2) * 64 + co) * 32 + c5];
      v[i] = yy == 0 ? make_float4((ma.x + mb.x) + mc.x, (ma.y + mb.y) + mc.y, (ma.z + mb.z) + mc.z, (ma.w + mb.w) + mc.w)
                     : make_float4((ma.x - mb.x) - mc.x, (ma.y - mb.y) - mc.y, (ma.z - mb.z) - mc.z, (ma.w - mb.w) - mc.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = 16 * wv + 4 * g4 + i;
      const bool ok = live && co < p.Cout;
      const size_t o = ((size_t)b * p.Cout + (co < p.Cout ? co : 0)) * yvol + orow;
      if (dy != nullptr) {
        // fused PReLU backward: g * prelu'(act_y) stored; the wave's sums of the slope and bias gradient terms
        const float sl = p.dslope[p.dnslope == 1 ? 0 : (co < p.Cout ? co : 0)];
        const float g4v[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        const float y4[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
        float o4[4], sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o4[k] = y4[k] > 0.f ? g4v[k] : sl * g4v[k];
          sa += y4[k] > 0.f ? 0.f : y4[k] * g4v[k];
          sb += o4[k];
        }
        if (ok) *reinterpret_cast<float4*>(Y + o) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        if (!ok) { sa = 0.f; sb = 0.f; }
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {  // all 64 lanes hold the same channel
          sa += __shfl_xor(sa, sft);
          sb += __shfl_xor(sb, sft);
        }
        if (lane == 0) {
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

// Measured and not kept (tests/tools/wino_bench.py, 64 -> 64 at 2 x 64^3, 0.466 ms as built incl. the re-layout launch;
// ablation builds: matrix waves + epilogue alone 0.354, loaders alone 0.276): two waves doing all of the input transform
// and two all of the slab requests 0.479; the last two reduction steps of a chunk held back across the barrier so that
// their MFMAs cover the first operand reads of the next chunk, operands read two steps ahead: 0.461 (within noise).
inline int wino2d_xt(const FP& p) { return p.Wo % 64 == 0 ? 16 : 8; }

inline bool wino2d_ok(const FP& p, const float* x, const float* ws, int Cin, int Cout, int kernel, int stride, bool has_ms) {
  static const bool off = FS_AB_ENV("FLOWSCI_FWD_NO_WINO2D") || FS_AB_ENV("FLOWSCI_FWD_NO_WINO4") || FS_AB_ENV("FLOWSCI_FWD_NO_WINO");
  if (off || kernel != 3 || stride != 1 || p.pad != 1 || has_ms) return false;
  if (Cin % 4 != 0 || Cout > 64 || p.CoutP != 64) return false;
  if (p.Wi != p.Wo || p.Wi % 32 != 0 || p.Di != p.Do || p.Hi != p.Ho) return false;
  if ((((uintptr_t)x | (uintptr_t)ws) & 15) != 0) return false;
  if ((long long)p.Di * p.Hi * p.Wi * 4 >= (1ll << 31)) return false;
  // one brick (2 z x 2 y x 64 x, or 2 z x 4 y x 32 x) per CU is enough: measured (tests/tools/wino_bench.py, 64 -> 64 layer)
  // 0.072 ms at 256 bricks against 0.129 for the direct small-brick kernel, 0.131 / 0.143 / 0.183 / 0.257 (2-D / F(4,3) /
  // F(2,3) / direct) at 512; the 64^3 trunk has 2048
  static const long long min_bricks = FS_AB_ENV_LL("FLOWSCI_WINO2D_MIN", 256);
  const int xt = wino2d_xt(p);
  return (long long)p.B * fs::cdiv(p.Do, 2) * fs::cdiv(p.Ho, 2 * (16 / xt)) * (p.Wo / (4 * xt)) >= min_bricks;
}

template <int XT>
void launch_wino2d_t(const float* X, const float* Ut, const float* bias, float* Y, const FP& p, hipStream_t st) {
  const dim3 g((unsigned)p.tiles, 1);
#ifdef FS_ABLATION  // instantiations that SKIP work (wrong results by design): measurement builds only
  static const int dbg = (int)FS_AB_ENV_LL("FLOWSCI_WINO_DBG", 0);
  if (dbg == 1) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<1, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 2) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<2, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 3) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<3, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 4) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<4, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else
#endif
    hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<0, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
}

inline int launch_wino2d(const float* X, const float* Ut, const float* bias, float* Y, FP& p, hipStream_t st) {
  const int xt = wino2d_xt(p);
  p.tz = fs::cdiv(p.Do, 2); p.ty = fs::cdiv(p.Ho, 2 * (16 / xt)); p.tx = p.Wo / (4 * xt);
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  if (p.tiles >= (1ll << 31)) return FS_ERR_SHAPE;
  if (xt == 16) launch_wino2d_t<16>(X, Ut, bias, Y, p, st);
  else launch_wino2d_t<8>(X, Ut, bias, Y, p, st);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
