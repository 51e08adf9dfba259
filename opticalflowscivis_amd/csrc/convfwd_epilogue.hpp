// convfwd_epilogue.hpp -- the epilogue of the loader-wave forward convolution kernels (conv3d_fwd_ws_kernel and, round 5,
// conv3d_fwd_s3_kernel), included textually inside the kernel body after the reduction: plain (bias, optional PReLU output
// Z, optional addend), or the fused PReLU-backward form (p.dy).  Names it uses from the including kernel: acc[MT][NT], p,
// bias, Y, lane, col, kh, wv, wz, wy, ly, lx, oz0, oy0, ox0, co0, b and the constants MT, NT, CP, R, EPI_ROWS (matrix waves
// per workgroup = rows of partial sums per brick).  A matrix lane holds
// output column `col` of NT rows for 16 channels per 32-channel block (the 32x32 MFMA accumulator layout).
#ifdef FS_ABLATION
  if (p.ab & 4) return;  // (measurement: no epilogue -- FLOWSCI_FWD_AB=4)
#endif
  if (p.dy != nullptr) {
    // ---- fused PReLU-backward epilogue: g * prelu'(act_y) stored instead of g, per-wave partial sums of the slope
    // gradient (g * act_y where act_y <= 0) and of the stored values (the producing layer's bias gradient).  Same 4 x 4
    // (lane x register) quad transpose as the plain epilogue below: a lane ends up with 4 consecutive x of ONE
    // channel -- one 16-byte load of act_y and one 16-byte store per (channel block, row) instead of four dword pairs.
    const float* __restrict__ dy = p.dy;
    float* __restrict__ Yg = Y;
    const int oz = oz0 + wz;
    const int qi = lane & 3;
    const int xq = ox0 + (lx & ~3);
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    auto swap1 = [](float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, false)); };
    auto swap2 = [](float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, false)); };
    float pa[MT][4], pb[MT][4];  // this lane's channel of block j: qi + 8 j + 4 kh
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) pa[m][j] = pb[m][j] = 0.f;
    const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;
    if (oz < p.Do) {  // wave-uniform
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co0 + m * 32 + qi + 8 * j + 4 * kh;
          const bool cok = co < p.Cout;
          const float sl = cok ? p.dslope[p.dnslope == 1 ? 0 : co] : 0.f;
          float4 yv[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) {  // all loads of the channel first
            const int oy = oy0 + (wy + n) * R + ly;
            const bool in = cok && oy < p.Ho && xq + 3 < p.Wo;
            const size_t o = ((size_t)b * p.Cout + (cok ? co : 0)) * yvol + ((size_t)oz * p.Ho + (oy < p.Ho ? oy : 0)) * p.Wo + xq;
            yv[n] = in ? *reinterpret_cast<const float4*>(dy + o) : make_float4(1.f, 1.f, 1.f, 1.f);
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            float a0 = acc[m][n][4 * j], a1 = acc[m][n][4 * j + 1], a2 = acc[m][n][4 * j + 2], a3 = acc[m][n][4 * j + 3];
            {
              const float rA = swap1(b0 ? a0 : a1), rB = swap1(b0 ? a2 : a3);
              if (b0) { a0 = rA; a2 = rB; } else { a1 = rA; a3 = rB; }
              const float rC = swap2(b1 ? a0 : a2), rD = swap2(b1 ? a1 : a3);
              if (b1) { a0 = rC; a1 = rD; } else { a2 = rC; a3 = rD; }
            }
            const int oy = oy0 + (wy + n) * R + ly;
            if (!cok || oy >= p.Ho || xq >= p.Wo) continue;
            const size_t o = ((size_t)b * p.Cout + co) * yvol + ((size_t)oz * p.Ho + oy) * p.Wo + xq;
            if (xq + 3 < p.Wo) {
              const float g[4] = {a0, a1, a2, a3};
              const float y4[4] = {yv[n].x, yv[n].y, yv[n].z, yv[n].w};
              float o4[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                o4[e] = y4[e] > 0.f ? g[e] : sl * g[e];
                pa[m][j] += y4[e] > 0.f ? 0.f : y4[e] * g[e];
                pb[m][j] += o4[e];
              }
              *reinterpret_cast<float4*>(Yg + o) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            } else {
              const float g[4] = {a0, a1, a2, a3};
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (xq + e < p.Wo) {
                  const float y1 = dy[o + e];
                  const float o1 = y1 > 0.f ? g[e] : sl * g[e];
                  pa[m][j] += y1 > 0.f ? 0.f : y1 * g[e];
                  pb[m][j] += o1;
                  Yg[o + e] = o1;
                }
            }
          }
        }
    }
    // lanes with equal (lane & 3) of a half-wave hold the same channels: butterfly over those 8, the first quad of
    // each half-wave writes the wave's row
    float* __restrict__ prow = p.dpart + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * EPI_ROWS + wv) * CP * 2;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = pa[m][j], bsum = pb[m][j];
#pragma unroll
        for (int sft = 4; sft < 32; sft <<= 1) {
          a += __shfl_xor(a, sft);
          bsum += __shfl_xor(bsum, sft);
        }
        if (col < 4) {
          const int cl = m * 32 + qi + 8 * j + 4 * kh;
          prow[cl * 2] = a;
          prow[cl * 2 + 1] = bsum;
        }
      }
    return;
  }

  // ---- epilogue.  With one workgroup per CU nothing hides it, and it is store-ISSUE-bound (~75 cycles per store
  // wave-instruction and CU whatever its width: 64 dword stores per lane were ~15 % of a conv0a brick).  A lane holds
  // ONE x for 16 channels; the four lanes of a quad transpose 4 x 4 blocks (4 consecutive x  x  channels r & 3 ..) with
  // two DPP swap stages, after which lane i of the quad holds 4 consecutive x of channel (r & 3) = i: one 16-byte store
  // instead of four dword stores.  Every lane takes part in the exchange; columns past Wo are dropped at the store.
  const float* __restrict__ ad = p.addend;
  const float* __restrict__ slope = p.slope;
  float* __restrict__ Zp = p.Z;
  float* __restrict__ Yp = Y;
  const int oz = oz0 + wz;
  if (oz < p.Do) {  // wave-uniform
    const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;
    const int qi = lane & 3;                       // this lane's channel inside a block after the transpose
    const int xq = ox0 + (lx & ~3);                // first of its four columns
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    auto swap1 = [](float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, false)); };
    auto swap2 = [](float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, false)); };
    // the per-channel values of all the lane's channels first: a load inside the store loop cannot be moved above the
    // previous channel's stores by the compiler
    float bvs[MT][4], svs[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + m * 32 + qi + 8 * j + 4 * kh;
        const bool cok = co < p.Cout;
        bvs[m][j] = (bias != nullptr && cok) ? bias[co] : 0.f;
        svs[m][j] = (Zp != nullptr && cok) ? slope[p.nslope == 1 ? 0 : co] : 0.f;
      }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + m * 32 + qi + 8 * j + 4 * kh;
        const bool cok = co < p.Cout;
        const float bv = bvs[m][j];
        const float sv = svs[m][j];
        // the addend (a residual unit's skip tensor / skip gradient) of this channel: its NT 16-byte loads first, in
        // flight together, instead of one load -> wait -> add -> store chain per row
        float4 apre[NT];
        if (ad != nullptr) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int oy = oy0 + (wy + n) * R + ly;
            const bool in = cok && oy < p.Ho && xq + 3 < p.Wo;
            const size_t o = ((size_t)b * p.Cout + (cok ? co : 0)) * yvol + ((size_t)oz * p.Ho + (oy < p.Ho ? oy : 0)) * p.Wo + xq;
            apre[n] = in ? *reinterpret_cast<const float4*>(ad + o) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float a0 = acc[m][n][4 * j], a1 = acc[m][n][4 * j + 1], a2 = acc[m][n][4 * j + 2], a3 = acc[m][n][4 * j + 3];
          {  // lanes x registers 4 x 4 transpose
            const float rA = swap1(b0 ? a0 : a1), rB = swap1(b0 ? a2 : a3);
            if (b0) { a0 = rA; a2 = rB; } else { a1 = rA; a3 = rB; }
            const float rC = swap2(b1 ? a0 : a2), rD = swap2(b1 ? a1 : a3);
            if (b1) { a0 = rC; a1 = rD; } else { a2 = rC; a3 = rD; }
          }
          const int oy = oy0 + (wy + n) * R + ly;
          if (!cok || oy >= p.Ho || xq >= p.Wo) continue;
          const size_t o = ((size_t)b * p.Cout + co) * yvol + ((size_t)oz * p.Ho + oy) * p.Wo + xq;
          float4 v = make_float4(a0 + bv, a1 + bv, a2 + bv, a3 + bv);
          if (xq + 3 < p.Wo) {
            const float4 av = ad != nullptr ? apre[n] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (Zp != nullptr) {
              *reinterpret_cast<float4*>(Yp + o) = v;
              *reinterpret_cast<float4*>(Zp + o) = make_float4((v.x > 0.f ? v.x : sv * v.x) + av.x, (v.y > 0.f ? v.y : sv * v.y) + av.y,
                                                               (v.z > 0.f ? v.z : sv * v.z) + av.z, (v.w > 0.f ? v.w : sv * v.w) + av.w);
            } else {
              *reinterpret_cast<float4*>(Yp + o) = make_float4(v.x + av.x, v.y + av.y, v.z + av.z, v.w + av.w);
            }
          } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (xq + e < p.Wo) {
                const float ae = ad != nullptr ? ad[o + e] : 0.f;
                if (Zp != nullptr) {
                  Yp[o + e] = vv[e];
                  Zp[o + e] = (vv[e] > 0.f ? vv[e] : sv * vv[e]) + ae;
                } else {
                  Yp[o + e] = vv[e] + ae;
                }
              }
          }
        }
      }
  }
