// corr3d.hip -- 3-D local-window correlation (cost volume) for gfx950.
//
// NEW CAPABILITY named by BASELINE.json (config 4: "3D correlation + trilinear warp HIP kernels");
// the reference has NO 3-D correlation (its Flow-3D IFNet has no cost volume at all), so
// the semantics are this build's generalisation of the 2-D layer (corr2d.hip / Corr_pyTorch):
//
//   out[b, ((dz+md)(2md+1) + (dy+md))(2md+1) + (dx+md), z, y, x]
//        = (1/C) sum_c f1[b,c,z,y,x] * f2[b,c,z+dz,y+dy,x+dx],     zero outside, dz-major.
//
// pinned to the reference only in the degenerate D = 1 case (== corr2d on the dz = 0 plane); oracle:
// oracle/corr.py::corr3d_closed.  The (2md+1)^3-channel output makes this a coarse-pyramid-level op
// (md = 4: 729 channels).
//
// Forward: one workgroup = one 8x32-pixel tile of slice z for ALL (2md+1)^3 displacements: the f1 tile of
// every channel is staged in LDS once (C <= 32 / C <= 64: 1 KB per channel), then the kernel loops over the 2md+1
// displacement planes dz; per plane, 2md+1 waves = dy rows, lane = 4 consecutive x (4 x (2md+1)
// accumulators), the f2 search window of plane z+dz streamed through LDS in chunks of 8 channels, and the
// plane's (2md+1)^2 output channels written before the next dz starts.  (Round 1 launched one workgroup per
// (z, dz) pair, which re-staged the f1 tile 2md+1 times.)  C > 64 re-stages f1 per chunk.
// Algorithmic HBM bytes: 4 (2C + (2md+1)^3) per voxel -- the (2md+1)^3-channel output dominates (729 of the
// 793 floats per voxel at C = 32, md = 4): the kernel is bound by its output stream.
// Backward, thread = voxel: per group of 16 channels (16 accumulators in registers) it loops dz with the
// (2md+1)^2 upstream values of that plane in registers -- every upstream value is read once per 16 channels
// (round 1: once per 8) -- and the other map's plane staged through LDS in chunks of 8 channels; grad_f2
// through the transposed-displacement identity; no atomics.  Algorithmic bytes 4 (4C + (2md+1)^3) per voxel.
#include "common.hpp"

namespace {

constexpr int TY = 8, TX = 32, CC = 8;

// C1: channels of the f1 tile kept resident in LDS (32 / 64; 0 = re-staged per chunk)
template <int MD, int C1>
__global__ __launch_bounds__(64 * (2 * MD + 1)) void corr3d_fwd_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out, int C, int D,
    int H, int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD, SCOLS = TX + 2 * MD, SW = (SCOLS + 3) / 4 * 4, NT = 64 * ND;
  __shared__ __attribute__((aligned(16))) float s2[CC][SR][SW];
  constexpr bool RESIDENT = C1 > 0;
  __shared__ __attribute__((aligned(16))) float s1[RESIDENT ? C1 : CC][TY][TX];

  const int z = blockIdx.z % D;
  const int b = blockIdx.z / D;
  const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
  const int t = threadIdx.x, lane = t & 63, dy = t >> 6;
  const int qy = lane >> 3, qx = (lane & 7) * 4;
  const size_t HW = (size_t)H * W, vol = (size_t)D * HW;
  const float* f1b = f1 + (size_t)b * C * vol + (size_t)z * HW;
  const int y = y0 + qy;
  const float fC = (float)C;

  auto stage_f1 = [&](int c0, int nch) {  // channels c0 .. c0+nch-1 -> s1[0 .. nch-1] (zeros past C / the image)
    for (int i = t; i < nch * TY * TX; i += NT) {
      const int c = i / (TY * TX), rem = i - c * (TY * TX);
      const int r = rem / TX, col = rem - r * TX;
      const int gy = y0 + r, gx = x0 + col;
      float v = 0.f;
      if (c0 + c < C && gy < H && gx < W) v = f1b[(size_t)(c0 + c) * vol + (size_t)gy * W + gx];
      s1[c][r][col] = v;
    }
  };
  if (RESIDENT) stage_f1(0, (C + CC - 1) / CC * CC);  // visible after the first barrier below

  // f2 window staging: chunk- and plane-invariant in-plane offsets resolved once; a chunk's loads are issued
  // back to back, branch-free, into registers, the next chunk's under the current chunk's FMA phase
  // (corr2d.hip's staging)
  constexpr int N2 = CC * SR * SCOLS, IT2 = (N2 + NT - 1) / NT;
  int off2[IT2];  // in-plane offset, -1: outside the image
#pragma unroll
  for (int it = 0; it < IT2; ++it) {
    const int i = t + NT * it;
    const int rem = i % (SR * SCOLS);
    const int r = rem / SCOLS, col = rem - r * SCOLS;
    const int gy = y0 + r - MD, gx = x0 + col - MD;
    off2[it] = (i < N2 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
  }
  float r2[IT2];
  auto fetch = [&](const float* f2b, int c0) {
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
      const int c = c0 + (t + NT * it) / (SR * SCOLS);
      const bool ok = off2[it] >= 0 && c < C;
      const float v = f2b[ok ? (size_t)c * vol + off2[it] : 0];
      r2[it] = ok ? v : 0.f;
    }
  };

  for (int dzi = 0; dzi < ND; ++dzi) {
    const int z2 = z + dzi - MD;
    float acc[4][ND];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < ND; ++j) acc[i][j] = 0.f;
    if (z2 >= 0 && z2 < D) {  // uniform per block; outside: the plane's channels are zero
      const float* f2b = f2 + (size_t)b * C * vol + (size_t)z2 * HW;
      fetch(f2b, 0);
      for (int c0 = 0; c0 < C; c0 += CC) {
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
          const int i = t + NT * it;
          const int c = i / (SR * SCOLS), rem = i - c * (SR * SCOLS);
          if (i < N2) s2[c][rem / SCOLS][rem % SCOLS] = r2[it];
        }
        if (!RESIDENT) stage_f1(c0, CC);
        __syncthreads();
        if (c0 + CC < C) fetch(f2b, c0 + CC);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
          const float4 a = *reinterpret_cast<const float4*>(&s1[RESIDENT ? c0 + c : c][qy][qx]);
          float row[4 + 2 * MD + 3];
          const float* rp = &s2[c][qy + dy][qx];
#pragma unroll
          for (int k = 0; k < (4 + 2 * MD + 3) / 4; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(rp + 4 * k);
            row[4 * k] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
          }
          const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < ND; ++j) acc[i][j] = fmaf(av[i], row[i + j], acc[i][j]);
        }
        __syncthreads();
      }
    }
    if (y < H) {
      // channel = (dzi*ND + dy)*ND + dx
      float* ob = out + (((size_t)b * ND * ND * ND + ((size_t)dzi * ND + dy) * ND) * D + z) * HW + (size_t)y * W;
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int x = x0 + qx + i;
          if (x < W) ob[(size_t)j * vol + x] = acc[i][j] / fC;
        }
    }
  }
}

// grad[c, p] = (1/C) sum_d g(d, p) * other[c, p + d]; blockIdx.z < B*D: (gout, f2) -> grad_f1,
// else (gout transposed on the fly, f1) -> grad_f2.  Thread = voxel.  Loop order: channel group (CG = 16
// accumulators in registers) > displacement plane dz (its (2md+1)^2 upstream values loaded ONCE per group,
// branch-free) > chunks of 8 channels of the other map's plane z+dz staged in LDS.
template <int MD, int CG>
__global__ __launch_bounds__(256) void corr3d_bwd_kernel(const float* __restrict__ f1,
                                                         const float* __restrict__ f2,
                                                         const float* __restrict__ gout,
                                                         float* __restrict__ g1, float* __restrict__ g2,
                                                         int B, int C, int D, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD, SW = TX + 2 * MD;
  constexpr int NS = CC * SR * SW, ITS = (NS + 255) / 256;
  __shared__ float s[CC][SR][SW];
  int bz = blockIdx.z;
  const bool second = bz >= B * D;
  if (second) bz -= B * D;
  const int z = bz % D, b = bz / D;
  float* __restrict__ grad = second ? g2 : g1;
  if (grad == nullptr) return;
  const float* __restrict__ other = second ? f1 : f2;
  const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
  const int t = threadIdx.x, py = t / TX, px = t % TX;
  const int y = y0 + py, x = x0 + px;
  const bool live = (y < H && x < W);
  const size_t HW = (size_t)H * W, vol = (size_t)D * HW;
  const float* gb = gout + (size_t)b * ND * ND * ND * vol;
  const float* ob = other + (size_t)b * C * vol;
  const float fC = (float)C;

  int offs[ITS];  // in-plane offset of the staged elements, -1: outside the image
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int i = t + 256 * it;
    const int rem = i % (SR * SW);
    const int r = rem / SW, col = rem - r * SW;
    const int gy = y0 + r - MD, gx = x0 + col - MD;
    offs[it] = (i < NS && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
  }

  for (int cg = 0; cg < C; cg += CG) {
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = 0.f;
    for (int k = 0; k < ND; ++k) {       // displacement plane dz = k - MD
      const int zz = z + k - MD;
      if (zz < 0 || zz >= D) continue;  // uniform per block
      float g[ND][ND];
#pragma unroll
      for (int j = 0; j < ND; ++j)
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          // 32-bit element offsets inside one sample's upstream tensor (check_shape bounds it): one address
          // register per load in flight instead of two
          unsigned idx;
          bool ok = live;
          if (!second) {
            idx = ((unsigned)((k * ND + j) * ND + i) * D + z) * (unsigned)HW + (unsigned)(y * W + x);
          } else {  // gT[d, q] = g[-d, q + d]
            const int yy = y + (j - MD), xx = x + (i - MD);
            ok = ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
            idx = ((unsigned)(((ND - 1 - k) * ND + (ND - 1 - j)) * ND + (ND - 1 - i)) * D + zz) * (unsigned)HW +
                  (unsigned)(yy * W + xx);
          }
          const float v = gb[ok ? idx : 0u];
          g[j][i] = ok ? v : 0.f;
        }
      const float* op = ob + (size_t)zz * HW;
#pragma unroll
      for (int cc = 0; cc < CG; cc += CC) {
        const int c0 = cg + cc;
        if (c0 >= C) break;  // uniform
        float rs[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int c = c0 + (t + 256 * it) / (SR * SW);
          const bool ok = offs[it] >= 0 && c < C;
          const float v = op[ok ? (size_t)c * vol + offs[it] : 0];
          rs[it] = ok ? v : 0.f;
        }
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int i = t + 256 * it;
          if (i < NS) (&s[0][0][0])[i] = rs[it];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CC; ++c) {
          float a = acc[cc + c];
#pragma unroll
          for (int j = 0; j < ND; ++j)
#pragma unroll
            for (int i = 0; i < ND; ++i) a = fmaf(g[j][i], s[c][py + j][px + i], a);
          acc[cc + c] = a;
        }
        __syncthreads();
      }
    }
    if (live)
#pragma unroll
      for (int c = 0; c < CG; ++c)
        if (cg + c < C) grad[((size_t)b * C + cg + c) * vol + (size_t)z * HW + (size_t)y * W + x] = acc[c] / fC;
  }
}

int check_shape(int B, int C, int D, int H, int W, int md) {
  if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return FS_ERR_SHAPE;
  if (md < 1 || md > 4) return FS_ERR_ARG;
  const long long nd = 2 * md + 1;
  if (2ll * B * D > 65535 || fs::cdiv(H, TY) > 65535) return FS_ERR_SHAPE;
  if (nd * nd * nd * D * H * W >= (1ll << 30)) return FS_ERR_SHAPE;  // 32-bit offsets inside one sample's cost volume
  return FS_OK;
}

template <int MD>
int launch(const float* f1, const float* f2, const float* gout, float* out, float* g1, float* g2, int B,
           int C, int D, int H, int W, bool bwd, hipStream_t st) {
  constexpr int ND = 2 * MD + 1;
  if (!bwd) {
    dim3 grid(fs::cdiv(W, TX), fs::cdiv(H, TY), B * D);
    if (C <= 32)
      hipLaunchKernelGGL((corr3d_fwd_kernel<MD, 32>), grid, dim3(64 * ND), 0, st, f1, f2, out, C, D, H, W);
    else if (C <= 64)
      hipLaunchKernelGGL((corr3d_fwd_kernel<MD, 64>), grid, dim3(64 * ND), 0, st, f1, f2, out, C, D, H, W);
    else
      hipLaunchKernelGGL((corr3d_fwd_kernel<MD, 0>), grid, dim3(64 * ND), 0, st, f1, f2, out, C, D, H, W);
  } else {
    dim3 grid(fs::cdiv(W, TX), fs::cdiv(H, TY), 2 * B * D);
    hipLaunchKernelGGL((corr3d_bwd_kernel<MD, 8>), grid, dim3(256), 0, st, f1, f2, gout, g1, g2, B, C, D, H, W);
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int dispatch(const float* f1, const float* f2, const float* gout, float* out, float* g1, float* g2, int B,
             int C, int D, int H, int W, int md, bool bwd, hipStream_t st) {
  switch (md) {
    case 1: return launch<1>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
    case 2: return launch<2>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
    case 3: return launch<3>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
    default: return launch<4>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
  }
}

}  // namespace

extern "C" int fs_corr3d_fwd(const float* f1, const float* f2, float* out, int B, int C, int D, int H,
                             int W, int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(out);
  const int rc = check_shape(B, C, D, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return dispatch(f1, f2, nullptr, out, nullptr, nullptr, B, C, D, H, W, max_displacement, false,
                  (hipStream_t)stream);
}

extern "C" int fs_corr3d_bwd(const float* f1, const float* f2, const float* grad_out, float* grad_f1,
                             float* grad_f2, int B, int C, int D, int H, int W, int max_displacement,
                             fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(grad_out);
  if (grad_f1 == nullptr && grad_f2 == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, D, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return dispatch(f1, f2, grad_out, nullptr, grad_f1, grad_f2, B, C, D, H, W, max_displacement, true,
                  (hipStream_t)stream);
}
