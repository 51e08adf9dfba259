// convwino.hpp -- the 64-channel k3 s1 "same" convolutions of the IFNet-3D trunk (forward and input gradient: 32
// launches and 27.7 of the 107.5 ms of the 2 x 256^3 step, at 0.86 of the fp32 MFMA peak as a direct implicit GEMM)
// with a 1-D Winograd F(2,3) transform along x: 4 multiplications for every 2 outputs and 3 taps instead of 6, i.e.
// 2/3 of the matrix-core work of the direct form.  Included inside convfwd.hip's anonymous namespace.
//
//   x-tile j = outputs x = 2j, 2j+1; inputs d_i = in[2j - 1 + i], i = 0..3; taps g = w[.., kz, ky, 0..2]:
//     V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)          input transform   (B^T d)
//     U = (g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2)  filter transform  (G g), done once by the weight re-layout
//     M_t[co, z, y, j] = sum_{ci, kz, ky} U_t[co, ci, kz, ky] * V_t[ci, z + kz - 1, y + ky - 1, j]     t = 0..3
//     y[2j] = M_0 + M_1 + M_2,   y[2j+1] = M_1 - M_2 - M_3                                  output transform (A^T M)
//   The sum over (ci, kz, ky) runs in the transformed domain: four independent implicit GEMMs with M = 64 output
//   channels, N = (z, y, x-tile), K = 9 Cin each -- 18 Cin multiply-adds per output instead of 27 Cin.  Only the x
//   axis is transformed: every coefficient is +-1 or 1/2 (fp32 rounding grows by ~2x, measured against fp64 in
//   tests/test_gpu_losses.py), the weights grow by 4/3 (a 3-D F(2,3)^3 would need 64/27 and a 1 MB filter per layer).
//
// Kernel: the loader-wave form of convfwd.hip.  One 8-wave workgroup per CU owns a brick of 2 z x 2 y rows x 64 x
// (32 x-tiles) for 64 output channels; reduction in chunks of CI = 4 input channels, two LDS buffers.
//   loader waves 4-7: the U slab of the chunk (CI x 9 x 4 x 64 floats) by `buffer_load_dwordx4 ... lds`; the input
//     rows global -> registers (one float4 per lane, 16 lanes per 64-float row, the two halo columns by one more
//     dword load in lanes 0 / 15 of the row), neighbours' columns through DPP row shifts, transformed, and written to
//     LDS as V[ci][staged row][t][x-tile]: wave w stages staged-z row w of every channel (4 y rows x 16 lanes).
//   matrix waves 0-3: one output row each (4 transformed accumulator sets x 2 channel tiles = 8 MFMA tiles, as
//     many as the direct kernel holds); both MFMA operands are "lane base + immediate" LDS reads.
//   epilogue: output transform in registers, one DPP swap between neighbouring x-tiles so that a lane holds 4
//     consecutive x of one channel, then the same fused epilogues as the direct kernel (bias, PReLU output, residual
//     addend; or the PReLU-backward form with its per-wave partial sums).
constexpr int WN_ROWF = 128;                    // floats per staged row: [t 0..3][x-tile 0..31]
constexpr int WN_VCH = 16 * WN_ROWF + 16;       // channel pitch of V (4 x 4 staged rows; +16: the two reduction halves
                                                // of an MFMA operand read land on different banks)
constexpr int WN_UCH = 9 * 4 * 64 + 16;         // channel pitch of U ([kz*3+ky][t][co], + the same pad)

template <int CI>
__global__ __launch_bounds__(512, 1) void conv3d_wino_ws_kernel(const float* __restrict__ X,
                                                               const float* __restrict__ Ut,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ Y, FP p) {
  constexpr int NV = CI * WN_VCH, NU = CI * WN_UCH;
  constexpr int NUP = (NU / 4 + 63) / 64;  // wave-instructions of 64 x 16 bytes for the U slab
  constexpr int NUW = (NUP + 3) / 4;       // per loader wave
  // (an LDS-DMA wave-instruction fills 256 floats: the slab's slot is rounded up to that, or its last instruction
  // would write out-of-range zeros over the first staged row of the OTHER buffer)
  constexpr int BUF = NV + NUP * 256;
  static_assert((NV % 4) == 0 && (NU % 4) == 0 && 2 * BUF * 4 <= 160 * 1024, "two 16-byte aligned buffers in LDS");
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
  const int oz0 = tzi * 2, oy0 = tyi * 2, ox0 = txi * 64;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int yr = lane >> 4, q = lane & 15;
    const int gz = oz0 - 1 + wv, gy = oy0 - 1 + yr, gx = ox0 + 4 * q;
    const bool rowok = gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
    const unsigned rbase = ((unsigned)(rowok ? gz : 0) * p.Hi + (rowok ? gy : 0)) * p.Wi;
    const unsigned voff = (rowok && gx < p.Wi) ? (rbase + gx) * 4u : DMA_OOB;   // Wi % 64 == 0: a float4 is in or out whole
    const int hx = q == 0 ? ox0 - 1 : ox0 + 64;
    const unsigned hoff = (rowok && (q == 0 || q == 15) && hx >= 0 && hx < p.Wi) ? (rbase + hx) * 4u : DMA_OOB;
    unsigned uoff[NUW];
#pragma unroll
    for (int k = 0; k < NUW; ++k) {
      const int piece = 64 * (wv + 4 * k) + lane;
      uoff[k] = piece < NU / 4 ? (unsigned)piece * 16u : DMA_OOB;
    }
    const int vdst = (wv * 4 + yr) * WN_ROWF + 2 * q;
    float xa[CI], xb[CI], xc[CI], xd[CI], xh[CI];
    auto fetch = [&](int c0) {
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const int ch = c0 + c;
        const bool live = ch < p.Cin;
        const float* base = X + ((size_t)b * p.Cin + (live ? ch : 0)) * xvol;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, live ? (int)((unsigned)xvol * 4u) : 0, 0x00020000);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
        xa[c] = __uint_as_float(v[0]); xb[c] = __uint_as_float(v[1]); xc[c] = __uint_as_float(v[2]); xd[c] = __uint_as_float(v[3]);
        xh[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, hoff, 0, 0));
      }
    };
    auto dma_u = [&](int c0, int buf) {
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Ut + (size_t)c0 * WN_UCH), (short)0, 0x7fffffff, 0x00020000);
      float* base = lds + buf * BUF + NV;
#pragma unroll
      for (int k = 0; k < NUW; ++k)
        if (wv + 4 * k < NUP)  // wave-uniform
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + 256 * (wv + 4 * k)), 16, uoff[k], 0, 0, 0);
    };
    auto put = [&](int buf) {
      float* base = lds + buf * BUF + vdst;
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        // left neighbour's d (x = 4q - 1) and right neighbour's a (x = 4q + 4); lanes 0 / 15 of a row keep the halo
        const float L = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(xh[c]), __float_as_uint(xd[c]), 0x111, 0xF, 0xF, false));
        const float Rr = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(xh[c]), __float_as_uint(xa[c]), 0x101, 0xF, 0xF, false));
        float* dst = base + c * WN_VCH;
        // tile 2q: d = (L, a, b, c); tile 2q + 1: d = (b, c, d, Rr)
        *reinterpret_cast<float2*>(dst + 0 * 32) = make_float2(L - xb[c], xb[c] - xd[c]);
        *reinterpret_cast<float2*>(dst + 1 * 32) = make_float2(xa[c] + xb[c], xc[c] + xd[c]);
        *reinterpret_cast<float2*>(dst + 2 * 32) = make_float2(xb[c] - xa[c], xd[c] - xc[c]);
        *reinterpret_cast<float2*>(dst + 3 * 32) = make_float2(xa[c] - xc[c], xc[c] - Rr);
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

  // ---- matrix waves: wave wv owns output row (z = oz0 + wv / 2, y = oy0 + wv % 2)
  const int col = lane & 31, kh = lane >> 5;
  const int wz = wv >> 1, wy = wv & 1;
  const int bBo = kh * (CI / 2) * WN_VCH + (wz * 4 + wy) * WN_ROWF + col;
  const int aBo = NV + kh * (CI / 2) * WN_UCH + col;
  constexpr int NP = (CI / 2) * 36;  // reduction steps per chunk: channel pair x (kz, ky) x t

  f32x16 acc[4][2];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
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
      const int cl = j / 36, r = j - cl * 36;
      const int kk = r >> 2, tt = r & 3;
      a[0] = aB[cl * WN_UCH + kk * 256 + tt * 64];
      a[1] = aB[cl * WN_UCH + kk * 256 + tt * 64 + 32];
      bq = bB[cl * WN_VCH + ((kk / 3) * 4 + (kk % 3)) * WN_ROWF + tt * 32];
    };
    float a0[2], a1[2], b0, b1;
    lds_ops(0, a0, b0);
#pragma unroll
    for (int j = 0; j < NP; j += 2) {
      if (j + 1 < NP) lds_ops(j + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      acc[j & 3][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0, acc[j & 3][0], 0, 0, 0);
      acc[j & 3][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0, acc[j & 3][1], 0, 0, 0);
      if (j + 2 < NP) lds_ops(j + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < NP) {
        acc[(j + 1) & 3][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1, acc[(j + 1) & 3][0], 0, 0, 0);
        acc[(j + 1) & 3][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1, acc[(j + 1) & 3][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }

  // ---- epilogue: output transform, pair exchange, fused epilogues
  const int oz = oz0 + wz, oy = oy0 + wy;
  const bool odd = (col & 1) != 0;
  const int xq = ox0 + 2 * (col & ~1);
  const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;
  const bool rowlive = oz < p.Do && oy < p.Ho;  // wave-uniform
  auto swap1 = [](float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, false)); };
  const float* __restrict__ ad = p.addend;
  const float* __restrict__ slope = p.slope;
  const float* __restrict__ dy = p.dy;
  float* __restrict__ Zp = p.Z;
  float pa[2][4][2], pb[2][4][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) pa[m][jb][0] = pa[m][jb][1] = pb[m][jb][0] = pb[m][jb][1] = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      float ye[4], yo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * jb + i;
        const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r];
        ye[i] = (m0 + m1) + m2;
        yo[i] = (m1 - m2) - m3;
      }
      // neighbouring x-tiles (lanes col, col ^ 1) trade halves: the even lane ends with channels i = 0, 1, the odd lane
      // with i = 2, 3, each with 4 consecutive x
      const float r0 = swap1(odd ? ye[0] : ye[2]), r1 = swap1(odd ? yo[0] : yo[2]);
      const float r2 = swap1(odd ? ye[1] : ye[3]), r3 = swap1(odd ? yo[1] : yo[3]);
      float4 v[2];
      v[0] = odd ? make_float4(r0, r1, ye[2], yo[2]) : make_float4(ye[0], yo[0], r0, r1);
      v[1] = odd ? make_float4(r2, r3, ye[3], yo[3]) : make_float4(ye[1], yo[1], r2, r3);
      if (!rowlive) continue;
      float4 pre[2];  // addend / act_y of both channels first: two loads in flight
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int co = m * 32 + 8 * jb + 4 * kh + (odd ? 2 : 0) + e;
        const size_t o = ((size_t)b * p.Cout + (co < p.Cout ? co : 0)) * yvol + ((size_t)oz * p.Ho + oy) * p.Wo + xq;
        const float* src = dy != nullptr ? dy : ad;
        pre[e] = (src != nullptr && co < p.Cout) ? *reinterpret_cast<const float4*>(src + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int co = m * 32 + 8 * jb + 4 * kh + (odd ? 2 : 0) + e;
        if (co >= p.Cout) continue;
        const size_t o = ((size_t)b * p.Cout + co) * yvol + ((size_t)oz * p.Ho + oy) * p.Wo + xq;
        if (dy != nullptr) {
          // fused PReLU backward: g * prelu'(act_y) stored; partial sums of the slope and bias gradients
          const float sl = p.dslope[p.dnslope == 1 ? 0 : co];
          const float g4[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
          const float y4[4] = {pre[e].x, pre[e].y, pre[e].z, pre[e].w};
          float o4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            o4[k] = y4[k] > 0.f ? g4[k] : sl * g4[k];
            pa[m][jb][e] += y4[k] > 0.f ? 0.f : y4[k] * g4[k];
            pb[m][jb][e] += o4[k];
          }
          *reinterpret_cast<float4*>(Y + o) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        } else {
          const float bv = bias != nullptr ? bias[co] : 0.f;
          const float4 w4 = make_float4(v[e].x + bv, v[e].y + bv, v[e].z + bv, v[e].w + bv);
          const float4 av = pre[e];
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
  if (dy != nullptr) {
    // the 16 lanes with equal (col & 1, kh) hold the same channels: butterfly over them, lanes col < 2 write the wave's
    // row [channel][slope-gradient term, bias-gradient term]
    float* __restrict__ prow = p.dpart + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 64 * 2;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float a = pa[m][jb][e], bsum = pb[m][jb][e];
#pragma unroll
          for (int sft = 2; sft < 32; sft <<= 1) {
            a += __shfl_xor(a, sft);
            bsum += __shfl_xor(bsum, sft);
          }
          if (col < 2) {
            const int cl = m * 32 + 8 * jb + 4 * kh + (odd ? 2 : 0) + e;
            prow[cl * 2] = a;
            prow[cl * 2 + 1] = bsum;
          }
        }
  }
}

inline bool wino_ok(const FP& p, const float* x, const float* ws, int Cin, int Cout, int kernel, int stride, bool has_ms) {
  static const bool off = FS_AB_ENV("FLOWSCI_FWD_NO_WINO");
  if (off || kernel != 3 || stride != 1 || p.pad != 1 || has_ms) return false;
  if (Cin % 4 != 0 || Cout > 64 || p.CoutP != 64) return false;
  if (p.Wi != p.Wo || p.Wi % 64 != 0 || p.Di != p.Do || p.Hi != p.Ho) return false;
  if ((((uintptr_t)x | (uintptr_t)ws) & 15) != 0) return false;
  if ((long long)p.Di * p.Hi * p.Wi * 4 >= (1ll << 31)) return false;
  // enough bricks for one workgroup per CU (the 64^3 trunk of the scale-1 blocks: 2 x 32 x 32 x 1 = 2048)
  return (long long)p.B * fs::cdiv(p.Do, 2) * fs::cdiv(p.Ho, 2) * (p.Wo / 64) >= 512;
}

inline int launch_wino(const float* X, const float* Ut, const float* bias, float* Y, FP& p, hipStream_t st) {
  p.tz = fs::cdiv(p.Do, 2); p.ty = fs::cdiv(p.Ho, 2); p.tx = p.Wo / 64;
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  if (p.tiles >= (1ll << 31)) return FS_ERR_SHAPE;
  // (measured and not kept, tests/tools/wino_bench.py on the 64 -> 64 layer at 2 x 64^3, 0.69 ms as built: 2-channel
  // chunks with two workgroups per CU 0.87 ms; persistent workgroups whose loaders run across brick boundaries 0.72 ms
  // at 256 VGPRs; operand reads a whole 8-MFMA group ahead and input rows requested two chunks ahead: no change;
  // s_setprio for the matrix waves: no change.  The matrix waves alone -- loaders reduced to their barriers -- take
  // 0.57 ms = the MFMA work at the ~2.06 GHz these kernels hold; the loaders alone 0.21 ms.)
  hipLaunchKernelGGL((conv3d_wino_ws_kernel<4>), dim3((unsigned)p.tiles, 1), dim3(512), 0, st, X, Ut, bias, Y, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
