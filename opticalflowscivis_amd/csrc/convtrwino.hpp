// convtrwino.hpp -- ConvTranspose3d(4, 2, 1) forward / Conv3d(4, 2, 1) input gradient for 17..32 output channels on
// rows of 64 input positions (the 64 -> 32 layers between the 64^3 trunk and the 128^3 grid: the heads' first
// deconvolution and conv0b's input gradient -- 6 launches, 7.3 of the 87 ms of the 2 x 256^3 step at 0.71-0.74 of the fp32
// MFMA peak as the class kernel of convtr.hip) in a 1-D Winograd F(4,2) domain along x.  Included inside convtr.hip's
// anonymous namespace.
//
//   Pairing of convtr_p8_kernel: position q owns the outputs 2q - p, p in {0,1} per axis, and along every axis
//     out[2q - p] = w[k(p,1)] in[q - 1] + w[k(p,0)] in[q],   k(0,.) = (1, 3), k(1,.) = (0, 2)
//   i.e. both parities are 2-tap filters over the SAME input window.  Along x, a tile of 4 positions q = 4j .. 4j + 3:
//     X_i = in[4j - 1 + i], i = 0..4;  g = (w[k(px,1)], w[k(px,0)])            (out_i = g0 X_i + g1 X_{i+1})
//     V = B^T X:  V0 = 2 X0 - X1 - 2 X2 + X3   V1 = -2 X1 - X2 + X3   V2 = 2 X1 - 3 X2 + X3   V3 = X3 - X1
//                 V4 = 2 X1 - X2 - 2 X3 + X4                                   (points 0, 1, -1, 2, inf)
//     U = G g:    U0 = g0 / 2   U1 = -(g0 + g1) / 2   U2 = (g1 - g0) / 6   U3 = g0 / 6 + g1 / 3   U4 = g1
//     M_t[(pz,py)][px, co][qz, qy, j] = sum_{ci, dz, dy} U_t[ci][(pz,py)][(dz,dy)][px, co] V_t[ci][qz - dz][qy - dy][j]
//     out = A^T M:  o0 = M0 + M1 + M2 + M3   o1 = M1 - M2 + 2 M3   o2 = M1 + M2 + 4 M3   o3 = M1 - M2 + 8 M3 + M4
//   5 products per 4 positions and parity instead of 8: 0.625 of the class form's multiply-adds; V is shared by both x
//   parities, whose filters sit side by side in the 64 matrix rows (px, co).  A lane ends with the 8 consecutive outputs
//   x = 8j - 1 .. 8j + 6 of a channel.  The LAST output column (x = 2 Wi - 1 = position q = Wi, which no tile of 4 covers)
//   is one tap of one input column: convtr_wino_edge_kernel.
//
// Kernel: loader-wave form, one 8-wave workgroup per CU.  The filter slab is what bounds this kernel (20 KB per input
// channel against 2.3 KB of transformed input; first form: all 32 channels per workgroup, 40 KB of slab per 40 MFMAs --
// ablation builds: loaders alone 0.5 of a 1.2 ms launch), so a workgroup takes HALF the output channels (blockIdx.y; 32
// matrix rows = (px, 16 channels): 10 KB of slab per input channel) and twice the positions: brick = position rows (qz; qy0
// .. qy0 + 3) x 64 positions = two MFMA column tiles of (2 position rows, 16 x-tiles); matrix wave = parity class (pz, py):
// 5 components x 2 column tiles = 10 accumulator tiles; reduction over (ci, dz, dy) in chunks of 4 input channels (80
// MFMAs per wave against 40 KB of slab); the slab is requested two chunks ahead (three LDS buffers), the input rows one
// chunk ahead.  Loader waves: a quarter of the slab each by LDS-DMA, and ten of
// the chunk's forty (channel, input row) units each (16 lanes per unit: float4 + the left neighbour's last column by DPP,
// 10 VALU operations, 5 dword writes).
constexpr int TW_RP = 5 * 16;                 // floats per staged input row: [t][x-tile]  (80 = 16 mod 32: the two position
                                              // rows of an MFMA operand read land on different halves of the 32 banks)
constexpr int TW_VCH = 10 * TW_RP;            // channel pitch of V: (qz - 1, qz) x (qy0 - 1 .. qy0 + 3)
constexpr int TW_UCH = FS_TRWINO_UCH;         // channel pitch of U: [half 2][class 4][neighbour 4][t 5][px 2][co 16]
constexpr int TW_UH = TW_UCH / 2;             // one channel half of it

// DBG (measurement builds, FLOWSCI_WINO_DBG): 1 = no slab requests, 2 = no input rows / transform, 3 = both (matrix waves +
// epilogue alone), 4 = no operand reads / MFMAs (loaders + epilogue alone), 5 = no epilogue stores
template <int DBG>
__global__ __launch_bounds__(512, 1) void convtr_wino_kernel(const float* __restrict__ X, const float* __restrict__ Ut,
                                                            const float* __restrict__ bias, float* __restrict__ Y, TP p) {
  constexpr int CI = 4;
  constexpr int NV = CI * TW_VCH, NU = CI * TW_UH;
  constexpr int NUP = NU / 256, NUW = NUP / 4;
  // three slab buffers (requested two chunks ahead: a trip to L2 and back does not fit into the chunk before) and two
  // buffers of transformed input rows (requested one chunk ahead into registers)
  constexpr int NBU = 3, NBV = 2;
  static_assert(NU % 1024 == 0 && TW_UH % 256 == 0 && (NBU * NU + NBV * NV) * 4 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) float ldsU[NBU * NU];
  __shared__ __attribute__((aligned(16))) float ldsV[NBV * NV];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;
  const int half = blockIdx.y;  // output channels 16 half .. 16 half + 15
  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);  // contiguous brick ranges per XCD
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int qz = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int qy0 = tyi * 4, q0 = txi * 64;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  const int nch = p.Cin / CI;

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    // units of this wave: (channel, input row) numbers 10 wv + 4 pass + (lane >> 4) of the chunk's 40, passes 0..2 (the
    // third pass has two units: lanes 32-63 idle)
    const int g = lane >> 4, jq = lane & 15;
    unsigned voff[3], hoff[3];
    int vdst[3], uc[3];
    bool act[3];
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
      act[ps] = 4 * ps + g < 10;
      const int u = 10 * wv + (act[ps] ? 4 * ps + g : 0);
      uc[ps] = u / 10;
      const int rr = u - 10 * uc[ps];  // staged row = zi * 5 + yi
      const int gz = qz - 1 + rr / 5, gy = qy0 - 1 + rr % 5, gx = q0 + 4 * jq;
      const bool rowok = act[ps] && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
      const unsigned rbase = ((unsigned)(rowok ? gz : 0) * p.Hi + (rowok ? gy : 0)) * p.Wi;
      voff[ps] = (rowok && gx < p.Wi) ? (rbase + gx) * 4u : DMA_OOB;  // Wi % 64 == 0: a float4 is in or out whole
      hoff[ps] = (rowok && jq == 0 && q0 > 0) ? (rbase + q0 - 1) * 4u : DMA_OOB;
      vdst[ps] = uc[ps] * TW_VCH + rr * TW_RP + jq;
    }
    // slab: the chunk's four channels x this half: 4 x TW_UH floats, contiguous in LDS, pitch TW_UCH in the workspace
    unsigned uoff[NUW];
#pragma unroll
    for (int k = 0; k < NUW; ++k) {
      const int f = (64 * (wv + 4 * k) + lane) * 4;  // float index inside the chunk's LDS image
      const int c = f / TW_UH, r = f - c * TW_UH;
      uoff[k] = (unsigned)(c * TW_UCH + half * TW_UH + r) * 4u;
    }
    auto dma_u = [&](int chunk, int buf) {
      if (DBG == 1 || DBG == 3) return;
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Ut + (size_t)chunk * CI * TW_UCH), (short)0, CI * TW_UCH * 4, 0x00020000);
      float* base = ldsU + buf * NU;
#pragma unroll
      for (int k = 0; k < NUW; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + 256 * (wv + 4 * k)), 16, uoff[k], 0, 0, 0);
    };
    float x1[3], x2[3], x3[3], x4[3], xh[3];
    auto fetch = [&](int chunk) {
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {
        if (DBG == 2 || DBG == 3) { x1[ps] = x2[ps] = x3[ps] = x4[ps] = xh[ps] = 0.f; continue; }
        const int ch = chunk * CI + uc[ps];
        const bool live = ch < p.Cin;
        const float* base = X + ((size_t)b * p.Cin + (live ? ch : 0)) * xvol;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, live ? (int)((unsigned)xvol * 4u) : 0, 0x00020000);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff[ps], 0, 0);
        x1[ps] = __uint_as_float(v[0]); x2[ps] = __uint_as_float(v[1]); x3[ps] = __uint_as_float(v[2]); x4[ps] = __uint_as_float(v[3]);
        xh[ps] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, hoff[ps], 0, 0));
      }
    };
    auto put = [&](int buf) {
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {
        // X0 = the left neighbour's last column (lane 0 of a unit: the column left of the brick, 0 outside the volume)
        const float x0 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(xh[ps]), __float_as_uint(x4[ps]), 0x111, 0xF, 0xF, false));
        if (act[ps]) {
          float* dst = ldsV + buf * NV + vdst[ps];
          const float d31 = x3[ps] - x1[ps];
          dst[0 * 16] = fmaf(2.f, x0 - x2[ps], d31);
          dst[1 * 16] = fmaf(-2.f, x1[ps], x3[ps] - x2[ps]);
          dst[2 * 16] = fmaf(2.f, x1[ps], fmaf(-3.f, x2[ps], x3[ps]));
          dst[3 * 16] = d31;
          dst[4 * 16] = fmaf(-2.f, d31, x4[ps] - x2[ps]);
        }
      }
    };
    // prologue: chunk 0 complete, chunk 1's slab and rows requested
    dma_u(0, 0);
    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    put(0);
    if (nch > 1) { dma_u(1, 1); fetch(1); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nch; ++k) {
      // period k: chunk k + 2's slab quarter is requested; what was requested a period ago -- chunk k + 1's slab quarter and
      // rows -- has landed: the rows are transformed into the other row buffer, then chunk k + 2's rows are requested
      if (k + 2 < nch) {
        dma_u(k + 2, (k + 2) % NBU);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DBG == 1 || DBG == 3 ? 0 : NUW) : "memory");  // all but the newest slab quarter
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (k + 1 < nch) {
        put((k + 1) & 1);
        if (k + 2 < nch) fetch(k + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // chunk k + 1 is in LDS; the matrix waves are done reading chunk k
    }
#else
    (void)xvol; (void)NUW;
#endif
    return;
  }

  // ---- matrix waves: wave wv = parity class (pz, py); MFMA column of tile ct = (position row 2 ct + (col >> 4), x-tile)
  const int col = lane & 31, kh = lane >> 5;
  const int pz = wv >> 1, py = wv & 1;
  const int bBo = kh * TW_VCH + (col >> 4) * TW_RP + (col & 15);
  const int aBo = kh * TW_UH + wv * 640 + col;
  constexpr int NP = (CI / 2) * 20;  // reduction steps per chunk: channel pair x (dz, dy) x t

  f32x16 acc[5][2];
#pragma unroll
  for (int tt = 0; tt < 5; ++tt)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][m][r] = 0.f;

  __builtin_amdgcn_s_barrier();  // chunk 0 has landed
  int bu = 0;
  for (int k = 0; k < nch; ++k) {
    const float* bB = ldsV + (k & 1) * NV + bBo;
    const float* aB = ldsU + bu * NU + aBo;
    auto lds_ops = [&](int j, float& a, float (&bq)[2]) {
      const int cl = j / 20, r = j - cl * 20;
      const int nb = r / 5, tt = r - nb * 5;
      const int dz = nb >> 1, dy = nb & 1;
      a = aB[cl * 2 * TW_UH + nb * 160 + tt * 32];
      const int row = (1 - dz) * 5 + (1 - dy);  // input row (qz - dz, qy0 + r - dy) of position row r = 0
      bq[0] = bB[cl * 2 * TW_VCH + row * TW_RP + tt * 16];
      bq[1] = bB[cl * 2 * TW_VCH + (row + 2) * TW_RP + tt * 16];  // column tile 1: position rows 2, 3
    };
    float a0, a1, b0[2], b1[2];
    if (DBG != 4) lds_ops(0, a0, b0);
#pragma unroll
    for (int j = 0; j < (DBG == 4 ? 0 : NP); j += 2) {
      lds_ops(j + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      acc[j % 5][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[0], acc[j % 5][0], 0, 0, 0);
      acc[j % 5][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[1], acc[j % 5][1], 0, 0, 0);
      if (j + 2 < NP) lds_ops(j + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      acc[(j + 1) % 5][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[0], acc[(j + 1) % 5][0], 0, 0, 0);
      acc[(j + 1) % 5][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[1], acc[(j + 1) % 5][1], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    bu = bu == NBU - 1 ? 0 : bu + 1;
  }

  // ---- epilogue: A^T per x parity gives the lane the 8 consecutive outputs 8j - 1 .. 8j + 6 (j = x-tile) of a channel.
  // Stores at that offset are 16-byte stores on 4-byte alignment -- measured at 1.6 TB/s (0.33 of a 1.22 ms launch,
  // ablation builds) -- so every lane passes its FIRST value to its left neighbour (one DPP row shift per channel) and
  // stores the aligned 8j .. 8j + 7.  Left over: the first value of a brick's first tile (x = 2 q0 - 1: the previous
  // brick's last tile has no right neighbour; scalar store, nothing at q0 = 0) and the last column of the row (edge kernel).
  const int tl = col & 15;
  const int xa = 2 * q0 + 8 * tl;  // first of the lane's aligned 8
  const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
  const int qy = qy0 + 2 * ct + (col >> 4);
  const int oz = 2 * qz - pz, oy = 2 * qy - py;
  const bool rowok = oz >= 0 && oz < p.Dout && oy >= 0 && oy < p.Hout;
  const size_t orow = rowok ? ((size_t)oz * p.Hout + oy) * p.Wout : 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    // matrix row 8 (r >> 2) + 4 kh + (r & 3) = channel of this half, x parity 0; the same row + 16 (register r + 8): parity 1
    const int co = 16 * half + 8 * (r >> 2) + 4 * kh + (r & 3);
    float v[9];
#pragma unroll
    for (int m = 0; m < 2; ++m) {  // m = px: 0 -> even outputs xa + 2 i, 1 -> odd outputs xa - 1 + 2 i
      const int rg = r + 8 * m;
      const float m0 = acc[0][ct][rg], m1 = acc[1][ct][rg], m2 = acc[2][ct][rg], m3 = acc[3][ct][rg], m4 = acc[4][ct][rg];
      const float s12 = m1 + m2, d12 = m1 - m2;
      v[0 + (1 - m)] = (m0 + s12) + m3;
      v[2 + (1 - m)] = fmaf(2.f, m3, d12);
      v[4 + (1 - m)] = fmaf(4.f, m3, s12);
      v[6 + (1 - m)] = fmaf(8.f, m3, d12) + m4;
    }
    // v[0..7] = outputs xa - 1 .. xa + 6; v[8] = output xa + 7 = the right neighbour's v[0] (none for the row's last tile)
    v[8] = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[0]), 0x101, 0xF, 0xF, false));
    if (!rowok || co >= p.Cout) continue;
    if (DBG == 5 && v[0] != 12345.f) continue;
    const float bv = bias != nullptr ? bias[co] : 0.f;
    const size_t o = ((size_t)b * p.CoutT + co) * yvol + orow;
    float* __restrict__ yrow = Y + o;
    const float* __restrict__ arow = p.addend ? p.addend + o : nullptr;
    float* __restrict__ zrow = p.Z ? p.Z + o : nullptr;
    const float sl = p.Z ? p.slope[p.nslope == 1 ? 0 : co] : 0.f;
    float4 lo = make_float4(v[1] + bv, v[2] + bv, v[3] + bv, v[4] + bv);
    float4 hi = make_float4(v[5] + bv, v[6] + bv, v[7] + bv, v[8] + bv);
    if (arow != nullptr) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + xa);
      lo.x += a4.x; lo.y += a4.y; lo.z += a4.z; lo.w += a4.w;
    }
    *reinterpret_cast<float4*>(yrow + xa) = lo;
    if (zrow != nullptr)
      *reinterpret_cast<float4*>(zrow + xa) = make_float4(lo.x > 0.f ? lo.x : sl * lo.x, lo.y > 0.f ? lo.y : sl * lo.y,
                                                          lo.z > 0.f ? lo.z : sl * lo.z, lo.w > 0.f ? lo.w : sl * lo.w);
    if (tl < 15) {
      if (arow != nullptr) {
        const float4 b4 = *reinterpret_cast<const float4*>(arow + xa + 4);
        hi.x += b4.x; hi.y += b4.y; hi.z += b4.z; hi.w += b4.w;
      }
      *reinterpret_cast<float4*>(yrow + xa + 4) = hi;
      if (zrow != nullptr)
        *reinterpret_cast<float4*>(zrow + xa + 4) = make_float4(hi.x > 0.f ? hi.x : sl * hi.x, hi.y > 0.f ? hi.y : sl * hi.y,
                                                                hi.z > 0.f ? hi.z : sl * hi.z, hi.w > 0.f ? hi.w : sl * hi.w);
    } else {  // the row's last tile: xa + 4 .. xa + 6 (xa + 7 belongs to the next brick's first tile or to the edge kernel)
      const float h3[3] = {hi.x, hi.y, hi.z};
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const float w = h3[e] + (arow != nullptr ? arow[xa + 4 + e] : 0.f);
        yrow[xa + 4 + e] = w;
        if (zrow != nullptr) zrow[xa + 4 + e] = w > 0.f ? w : sl * w;
      }
    }
    if (tl == 0 && q0 > 0) {  // x = 2 q0 - 1: the value no left neighbour took
      const float w = v[0] + bv + (arow != nullptr ? arow[xa - 1] : 0.f);
      yrow[xa - 1] = w;
      if (zrow != nullptr) zrow[xa - 1] = w > 0.f ? w : sl * w;
    }
  }
  }
}

// The last output column x = Wout - 1 = 2 Wi - 1 (parity px = 1 of position q = Wi): one tap along x, k = 2, of the input
// column Wi - 1 -- a 2-D transposed convolution over (z, y).  Its weights are read back from the slab (component 0 of
// the px = 1 rows holds w[.., k = 2] / 2).  One workgroup per (b, output plane z, 64 output rows y): the input column's
// rows qz - 1, qz and the plane's two classes' weights go through LDS in chunks of 8 input channels; thread = (y mod 8,
// co) with 8 outputs; a weight read from LDS feeds 4 of them.
__global__ __launch_bounds__(256) void convtr_wino_edge_kernel(const float* __restrict__ X, const float* __restrict__ Ut,
                                                               const float* __restrict__ bias, float* __restrict__ Y, TP p) {
  constexpr int CE = 8, NY = 8, YB = 8 * NY;   // 64 output rows per workgroup
  __shared__ float xs[CE][2][YB / 2 + 2];     // [ci][input row qz - 1 / qz][iy - iy0]
  __shared__ float wl[2][4][CE][32];          // [py][neighbour][ci][co]: w[.., k = 2] of the plane's two classes
  const int nyb = (p.Hout + YB - 1) / YB;
  const int yb = blockIdx.x % nyb, z = (blockIdx.x / nyb) % p.Dout, b = blockIdx.x / (nyb * p.Dout);
  const int pz = z & 1, qz = (z + pz) >> 1;
  const int y0 = yb * YB, iy0 = y0 / 2 - 1;    // input rows iy0 .. iy0 + YB / 2 + 1 serve outputs y0 .. y0 + YB - 1
  const int t = threadIdx.x, co = t & 31, ys = t >> 5;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi, yvol = (size_t)p.Dout * p.Hout * p.Wout;
  float s[NY];
#pragma unroll
  for (int k = 0; k < NY; ++k) s[k] = 0.f;
  for (int c0 = 0; c0 < p.Cin; c0 += CE) {
    __syncthreads();
    for (int i = t; i < CE * 2 * (YB / 2 + 2); i += 256) {
      const int j = i % (YB / 2 + 2), r = i / (YB / 2 + 2);
      const int zi = r & 1, c = r >> 1;
      const int iz = qz - 1 + zi, iy = iy0 + j;
      const bool ok = c0 + c < p.Cin && iz >= 0 && iz < p.Di && iy >= 0 && iy < p.Hi;
      xs[c][zi][j] = ok ? X[((size_t)b * p.Cin + c0 + c) * xvol + ((size_t)iz * p.Hi + iy) * p.Wi + (p.Wi - 1)] : 0.f;
    }
    for (int i = t; i < 2 * 4 * CE * 32; i += 256) {
      const int o = i & 31, c = (i >> 5) % CE, nb = (i / (32 * CE)) & 3, py = i / (32 * CE * 4);
      wl[py][nb][c][o] = (c0 + c < p.Cin) ? 2.f * Ut[(size_t)(c0 + c) * TW_UCH + (o >> 4) * TW_UH + (pz * 2 + py) * 640 + nb * 160 + 16 + (o & 15)] : 0.f;
    }
    __syncthreads();
    // y = y0 + ys + 8 k: parity py = ys & 1 for all k (y0 and 8 k are even); qy = (y + py) / 2; input row qy - dy
    const int py = ys & 1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int zi = 1 - (nb >> 1), dy = nb & 1;
#pragma unroll
      for (int c = 0; c < CE; ++c) {
        const float w = wl[py][nb][c][co];
#pragma unroll
        for (int k = 0; k < NY; ++k) {
          const int j = ((ys + 8 * k + py) >> 1) - dy + 1;   // = qy - dy - iy0 (out-of-volume rows hold 0)
          s[k] = fmaf(w, xs[c][zi][j], s[k]);
        }
      }
    }
  }
  if (co >= p.Cout) return;
  const float bv = bias != nullptr ? bias[co] : 0.f;
  const float sl = p.Z ? p.slope[p.nslope == 1 ? 0 : co] : 0.f;
#pragma unroll
  for (int k = 0; k < NY; ++k) {
    const int y = y0 + ys + 8 * k;
    if (y >= p.Hout) break;
    const size_t o = ((size_t)b * p.CoutT + co) * yvol + ((size_t)z * p.Hout + y) * p.Wout + (p.Wout - 1);
    const float w = s[k] + bv + (p.addend != nullptr ? p.addend[o] : 0.f);
    Y[o] = w;
    if (p.Z != nullptr) p.Z[o] = w > 0.f ? w : sl * w;
  }
}

inline bool trwino_ok(const TP& p, const float* x, const float* ws, int slices) {
  static const bool off = getenv("FLOWSCI_TR_NO_WINO") != nullptr;
  if (off || slices != 1 || p.Cout <= 16 || p.Cout > 32 || p.Cin % 4 != 0) return false;
  if (p.Dout != 2 * p.Di || p.Hout != 2 * p.Hi || p.Wout != 2 * p.Wi || p.Wi % 64 != 0) return false;
  if ((((uintptr_t)x | (uintptr_t)ws) & 15) != 0) return false;
  if ((long long)p.Di * p.Hi * p.Wi * 4 >= (1ll << 31)) return false;
  // one workgroup (1 x 4 position rows x 64 positions x half the channels) per CU at least
  return (long long)p.B * (p.Di + 1) * fs::cdiv(p.Hi + 1, 4) * (p.Wi / 64) * 2 >= 256;
}

inline int launch_trwino(const float* X, const float* Ut, const float* bias, float* Y, TP& p, hipStream_t st) {
  p.tz = p.Di + 1; p.ty = fs::cdiv(p.Hi + 1, 4); p.tx = p.Wi / 64;
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  if (p.tiles >= (1ll << 31)) return FS_ERR_SHAPE;
  static const int dbg = getenv("FLOWSCI_WINO_DBG") ? atoi(getenv("FLOWSCI_WINO_DBG")) : 0;
  const dim3 g((unsigned)p.tiles, 2);  // y: the two halves of the output channels
  if (dbg == 1) hipLaunchKernelGGL(convtr_wino_kernel<1>, g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 2) hipLaunchKernelGGL(convtr_wino_kernel<2>, g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 3) hipLaunchKernelGGL(convtr_wino_kernel<3>, g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 4) hipLaunchKernelGGL(convtr_wino_kernel<4>, g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 5) hipLaunchKernelGGL(convtr_wino_kernel<5>, g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else hipLaunchKernelGGL(convtr_wino_kernel<0>, g, dim3(512), 0, st, X, Ut, bias, Y, p);
  if (dbg == 0)
    hipLaunchKernelGGL(convtr_wino_edge_kernel, dim3((unsigned)((long long)p.B * p.Dout * ((p.Hout + 63) / 64))), dim3(256), 0, st, X, Ut, bias,
                       Y, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
