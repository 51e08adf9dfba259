// warp3d.hip -- Flow-3D trilinear backward warp for gfx950.
//
// Reference semantics: Flow-3D/model/warplayer.py:9-41.  The reference feeds a grid whose
// channels are (linspace over H, linspace over D, linspace over W) to a 5-D grid_sample
// whose (x, y, z) address input dims (W, H, D).  Net effect, reproduced here exactly:
//
//   out[b,c,d,h,w] = trilinear( in[b,c], iz = ((gz+1)/2)(D-1), iy = ((gy+1)/2)(H-1),
//                                         ix = ((gx+1)/2)(W-1) ),   border clamp,
//   gx = lin_H[h] + F0/((H-1)/2),  gy = lin_D[d] + F1/((D-1)/2),  gz = lin_W[w] + F2/((W-1)/2)
//
// i.e. for cubic volumes out[d,h,w] = in[w+F2, d+F1, h+F0]: the gather walks the input's
// fastest axis (W) along the OUTPUT's h axis.  A thread-per-voxel kernel with lanes on w
// (the output's fastest axis) would therefore gather with a D-plane stride between lanes.
//
// MI355X design: one workgroup (4 waves) owns an output tile of 1 d x 64 h x TW w.
//   phase 1  flow tile (3 x 64 x TW) is read with lanes on w (coalesced 128/256-B rows)
//            and parked in LDS with a +1 padded row so it can be re-read transposed;
//   phase 2  lanes switch to h: lane l handles h0+l, each wave a slice of the w range.
//            The 8 corner gathers of one wave-instruction now hit ~256 contiguous input
//            bytes (row iy, plane iz); results go back to LDS (own slot, conflict-free);
//   phase 3  lanes back on w: coalesced store of the output tile.
// HBM traffic is the algorithmic 20 B/voxel (12 flow + 4 gather + 4 store) as long as the
// iy/iy+1 row pairs shared by neighbouring d tiles hit in L2: tiles of consecutive d map to
// the same XCD (tiles-per-d-slab is a multiple of 8 for the BASELINE sizes).
//
// Backward (grad_flow, optional grad_in): same tiling; coordinates and the 8 corners are
// recomputed (nothing is saved by forward), grad_flow leaves through the LDS transpose,
// grad_in (only when the caller asks for it) by float atomics.
#include "common.hpp"

namespace {

constexpr int TH = 64;   // tile extent along h = one wave of lanes in phase 2
constexpr int NT = 256;  // 4 waves

struct W3P {
  int B, C, D, H, W;          // batch, image channels, extent of the flow == of the output
  int Di, Hi, Wi;             // extent of the sampled volume (the reference lets it differ:
                              // the grid comes from the flow's shape, warplayer.py:11-22)
  int tilesH, tilesW;
  float stepD, stepH, stepW;  // linspace steps 2/(n-1) over the FLOW dims
  float sD, sH, sW;           // (n-1)/2 of the INPUT dims (warplayer.py:24-26)
  int flowC;                  // channels of the flow tensor: 3 (single) or 6 (IFNet pair)
};

// blockIdx.y selects the member of a pair: IFNet warps img0 with flow[:, :3] and img1 with
// flow[:, 3:6] (Flow-3D/model/IFNet.py:190-191); one launch serves both and reads / writes the
// 6-channel flow tensors in place (no slicing copies).
struct W3Fwd { const float* in[2]; float* out[2]; };
struct W3Bwd { const float* in[2]; const float* gout[2]; float* gin[2]; };

struct Samp3 {
  float ix, iy, iz;     // clipped, un-normalised coordinates
  float mx, my, mz;     // border-clip gradient multipliers (0 or 1)
  int x0, y0, z0;       // floor
};

__device__ __forceinline__ Samp3 w3_coords(const W3P& p, int d, int h, int w, float f0, float f1,
                                           float f2) {
#pragma clang fp contract(off)
  Samp3 s;
  const float gx = fs::linspace_pm1(h, p.H, p.stepH) + f0 / p.sH;  // warplayer.py:15,24
  const float gy = fs::linspace_pm1(d, p.D, p.stepD) + f1 / p.sD;  // warplayer.py:17,25
  const float gz = fs::linspace_pm1(w, p.W, p.stepW) + f2 / p.sW;  // warplayer.py:19,26
  // grid_sampler_unnormalize, align_corners=True; x -> dim W, y -> dim H, z -> dim D
  float ix = ((gx + 1.0f) / 2.0f) * (float)(p.Wi - 1);
  float iy = ((gy + 1.0f) / 2.0f) * (float)(p.Hi - 1);
  float iz = ((gz + 1.0f) / 2.0f) * (float)(p.Di - 1);
  s.ix = fs::clip_border(ix, p.Wi, &s.mx);
  s.iy = fs::clip_border(iy, p.Hi, &s.my);
  s.iz = fs::clip_border(iz, p.Di, &s.mz);
  s.x0 = (int)floorf(s.ix);
  s.y0 = (int)floorf(s.iy);
  s.z0 = (int)floorf(s.iz);
  return s;
}

// 8 corner values, out-of-range corners (only the +1 ones can be) read as 0 like ATen's
// within_bounds checks.  v[z][y][x].
struct Corners {
  float v000, v001, v010, v011, v100, v101, v110, v111;
};

__device__ __forceinline__ Corners w3_gather(const float* __restrict__ vol, const W3P& p,
                                             const Samp3& s) {
  const int x1ok = (s.x0 + 1 < p.Wi), y1ok = (s.y0 + 1 < p.Hi), z1ok = (s.z0 + 1 < p.Di);
  const int x1 = x1ok ? s.x0 + 1 : s.x0;
  const int y1 = y1ok ? s.y0 + 1 : s.y0;
  const int z1 = z1ok ? s.z0 + 1 : s.z0;
  const int HW = p.Hi * p.Wi;
  const int r00 = s.z0 * HW + s.y0 * p.Wi, r01 = s.z0 * HW + y1 * p.Wi;
  const int r10 = z1 * HW + s.y0 * p.Wi, r11 = z1 * HW + y1 * p.Wi;
  Corners c;
  c.v000 = vol[r00 + s.x0];
  c.v001 = vol[r00 + x1];
  c.v010 = vol[r01 + s.x0];
  c.v011 = vol[r01 + x1];
  c.v100 = vol[r10 + s.x0];
  c.v101 = vol[r10 + x1];
  c.v110 = vol[r11 + s.x0];
  c.v111 = vol[r11 + x1];
  if (!x1ok) { c.v001 = 0.f; c.v011 = 0.f; c.v101 = 0.f; c.v111 = 0.f; }
  if (!y1ok) { c.v010 = 0.f; c.v011 = 0.f; c.v110 = 0.f; c.v111 = 0.f; }
  if (!z1ok) { c.v100 = 0.f; c.v101 = 0.f; c.v110 = 0.f; c.v111 = 0.f; }
  return c;
}

// blockIdx.x -> (b, d, h-tile, w-tile), w-tile fastest.  Returns the w-tile index.
__device__ __forceinline__ int decode_tile(const W3P& p, int& b, int& d, int& h0) {
  int bid = blockIdx.x;
  const int tw = bid % p.tilesW; bid /= p.tilesW;
  const int th = bid % p.tilesH; bid /= p.tilesH;
  d = bid % p.D;
  b = bid / p.D;
  h0 = th * TH;
  return tw;
}

template <int TW>
__global__ __launch_bounds__(NT) void warp3d_fwd_kernel(W3Fwd io, const float* __restrict__ flow,
                                                        W3P p) {
  const float* __restrict__ in = io.in[blockIdx.y];
  float* __restrict__ out = io.out[blockIdx.y];
  constexpr int LDW = TW + 1;
  constexpr int RP = NT / TW;  // tile rows covered per pass in the w-major phases
  constexpr int NW = TW / 4;   // voxels per thread in the h-major phase
  __shared__ float sF[3][TH][LDW];
  __shared__ float sO[TH][LDW];

  int b, d, h0;
  const int w0 = decode_tile(p, b, d, h0) * TW;
  const int t = threadIdx.x;
  const int HW = p.H * p.W;
  const size_t vol = (size_t)p.D * HW;
  const size_t ivol = (size_t)p.Di * p.Hi * p.Wi;

  // phase 1: flow tile, lanes on w
  {
    const float* fb = flow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol + (size_t)d * HW;
    const int lw = t % TW, r = t / TW;
    const int w = min(w0 + lw, p.W - 1);
#pragma unroll
    for (int row = r; row < 3 * TH; row += RP) {
      const int c = row / TH, hh = row % TH;
      const int h = min(h0 + hh, p.H - 1);
      sF[c][hh][lw] = fb[(size_t)c * vol + h * p.W + w];
    }
  }
  __syncthreads();

  const int lane = t & 63, wv = t >> 6;
  const int h = min(h0 + lane, p.H - 1);
  for (int c = 0; c < p.C; ++c) {
    const float* __restrict__ vin = in + ((size_t)b * p.C + c) * ivol;
    // phase 2: lanes on h; each wave takes NW consecutive w of the tile
#pragma unroll 4
    for (int k = 0; k < NW; ++k) {
      const int lw = wv * NW + k;
      const int w = min(w0 + lw, p.W - 1);
      const Samp3 s = w3_coords(p, d, h, w, sF[0][lane][lw], sF[1][lane][lw], sF[2][lane][lw]);
      const Corners q = w3_gather(vin, p, s);
      float r;
      {
#pragma clang fp contract(off)
        const float ax = s.ix - (float)s.x0, bx = (float)(s.x0 + 1) - s.ix;
        const float ay = s.iy - (float)s.y0, by = (float)(s.y0 + 1) - s.iy;
        const float az = s.iz - (float)s.z0, bz = (float)(s.z0 + 1) - s.iz;
        // ATen order: tnw, tne, tsw, tse, bnw, bne, bsw, bse
        r = q.v000 * (bx * by * bz);
        r += q.v001 * (ax * by * bz);
        r += q.v010 * (bx * ay * bz);
        r += q.v011 * (ax * ay * bz);
        r += q.v100 * (bx * by * az);
        r += q.v101 * (ax * by * az);
        r += q.v110 * (bx * ay * az);
        r += q.v111 * (ax * ay * az);
      }
      sO[lane][lw] = r;
    }
    __syncthreads();
    // phase 3: lanes on w, coalesced store
    {
      float* ob = out + ((size_t)b * p.C + c) * vol + (size_t)d * HW;
      const int lw = t % TW, r = t / TW;
      const int w = w0 + lw;
#pragma unroll
      for (int row = r; row < TH; row += RP) {
        const int hh = h0 + row;
        if (hh < p.H && w < p.W) ob[hh * p.W + w] = sO[row][lw];
      }
    }
    if (c + 1 < p.C) __syncthreads();
  }
}

template <int TW, bool WITH_GIN>
__global__ __launch_bounds__(NT) void warp3d_bwd_kernel(W3Bwd io, const float* __restrict__ flow,
                                                        float* __restrict__ gflow, W3P p) {
  const float* __restrict__ in = io.in[blockIdx.y];
  const float* __restrict__ gout = io.gout[blockIdx.y];
  float* __restrict__ gin = io.gin[blockIdx.y];
  constexpr int LDW = TW + 1;
  constexpr int RP = NT / TW;
  constexpr int NW = TW / 4;
  __shared__ float sF[3][TH][LDW];
  __shared__ float sG[TH][LDW];

  int b, d, h0;
  const int w0 = decode_tile(p, b, d, h0) * TW;
  const int t = threadIdx.x;
  const int HW = p.H * p.W;
  const size_t vol = (size_t)p.D * HW;
  const size_t ivol = (size_t)p.Di * p.Hi * p.Wi;
  const int iHW = p.Hi * p.Wi;
  const int lwW = t % TW, rW = t / TW;  // w-major phase coordinates

  {
    const float* fb = flow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol + (size_t)d * HW;
    const int w = min(w0 + lwW, p.W - 1);
#pragma unroll
    for (int row = rW; row < 3 * TH; row += RP) {
      const int c = row / TH, hh = row % TH;
      const int h = min(h0 + hh, p.H - 1);
      sF[c][hh][lwW] = fb[(size_t)c * vol + h * p.W + w];
    }
  }

  const int lane = t & 63, wv = t >> 6;
  const int hq = h0 + lane;
  const int h = min(hq, p.H - 1);
  float acc[NW][3];
#pragma unroll
  for (int k = 0; k < NW; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;

  for (int c = 0; c < p.C; ++c) {
    if (c > 0) __syncthreads();  // previous channel's sG readers are done
    {
      const float* gb = gout + ((size_t)b * p.C + c) * vol + (size_t)d * HW;
      const int w = w0 + lwW;
#pragma unroll
      for (int row = rW; row < TH; row += RP) {
        const int hh = h0 + row;
        // out-of-tile voxels get a zero upstream gradient: they then add nothing anywhere
        sG[row][lwW] = (hh < p.H && w < p.W) ? gb[hh * p.W + w] : 0.f;
      }
    }
    __syncthreads();
    const float* __restrict__ vin = in + ((size_t)b * p.C + c) * ivol;
    float* __restrict__ gvin = WITH_GIN ? gin + ((size_t)b * p.C + c) * ivol : nullptr;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int lw = wv * NW + k;
      const int w = min(w0 + lw, p.W - 1);
      const Samp3 s = w3_coords(p, d, h, w, sF[0][lane][lw], sF[1][lane][lw], sF[2][lane][lw]);
      const Corners q = w3_gather(vin, p, s);
      const float g = sG[lane][lw];
      const float ax = s.ix - (float)s.x0, bx = (float)(s.x0 + 1) - s.ix;
      const float ay = s.iy - (float)s.y0, by = (float)(s.y0 + 1) - s.iy;
      const float az = s.iz - (float)s.z0, bz = (float)(s.z0 + 1) - s.iz;
      // d out / d ix etc. (ATen grid_sampler_3d_backward, factored)
      const float dx = bz * (by * (q.v001 - q.v000) + ay * (q.v011 - q.v010)) +
                       az * (by * (q.v101 - q.v100) + ay * (q.v111 - q.v110));
      const float dy = bz * (bx * (q.v010 - q.v000) + ax * (q.v011 - q.v001)) +
                       az * (bx * (q.v110 - q.v100) + ax * (q.v111 - q.v101));
      const float dz = by * (bx * (q.v100 - q.v000) + ax * (q.v101 - q.v001)) +
                       ay * (bx * (q.v110 - q.v010) + ax * (q.v111 - q.v011));
      acc[k][0] += g * dx * s.mx;
      acc[k][1] += g * dy * s.my;
      acc[k][2] += g * dz * s.mz;
      if (WITH_GIN) {
        if (hq < p.H && w0 + lw < p.W) {
          const int x1ok = (s.x0 + 1 < p.Wi), y1ok = (s.y0 + 1 < p.Hi), z1ok = (s.z0 + 1 < p.Di);
          const int r00 = s.z0 * iHW + s.y0 * p.Wi, r01 = r00 + p.Wi;
          const int r10 = r00 + iHW, r11 = r10 + p.Wi;
          atomicAdd(gvin + r00 + s.x0, g * (bx * by * bz));
          if (x1ok) atomicAdd(gvin + r00 + s.x0 + 1, g * (ax * by * bz));
          if (y1ok) atomicAdd(gvin + r01 + s.x0, g * (bx * ay * bz));
          if (y1ok && x1ok) atomicAdd(gvin + r01 + s.x0 + 1, g * (ax * ay * bz));
          if (z1ok) {
            atomicAdd(gvin + r10 + s.x0, g * (bx * by * az));
            if (x1ok) atomicAdd(gvin + r10 + s.x0 + 1, g * (ax * by * az));
            if (y1ok) atomicAdd(gvin + r11 + s.x0, g * (bx * ay * az));
            if (y1ok && x1ok) atomicAdd(gvin + r11 + s.x0 + 1, g * (ax * ay * az));
          }
        }
      }
    }
  }

  if (gflow == nullptr) return;
  // chain rule through unnormalize ((size-1)/2) and through flow/((dim-1)/2):
  //   F0 -> gx (H-normalised) -> ix (W),  F1 -> gy (D) -> iy (H),  F2 -> gz (W) -> iz (D)
  const float k0 = ((float)(p.Wi - 1) * 0.5f) / p.sH;
  const float k1 = ((float)(p.Hi - 1) * 0.5f) / p.sD;
  const float k2 = ((float)(p.Di - 1) * 0.5f) / p.sW;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int lw = wv * NW + k;  // own slots: nobody else reads or writes them
    sF[0][lane][lw] = acc[k][0] * k0;
    sF[1][lane][lw] = acc[k][1] * k1;
    sF[2][lane][lw] = acc[k][2] * k2;
  }
  __syncthreads();
  {
    float* gb = gflow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol + (size_t)d * HW;
    const int w = w0 + lwW;
#pragma unroll
    for (int row = rW; row < 3 * TH; row += RP) {
      const int c = row / TH, hh = row % TH;
      const int hg = h0 + hh;
      if (hg < p.H && w < p.W) gb[(size_t)c * vol + hg * p.W + w] = sF[c][hh][lwW];
    }
  }
}

int make_params(W3P& p, int B, int C, const int* in_dhw, int D, int H, int W, int TW) {
  const int Di = in_dhw ? in_dhw[0] : D, Hi = in_dhw ? in_dhw[1] : H, Wi = in_dhw ? in_dhw[2] : W;
  if (B < 1 || C < 1 || D < 2 || H < 2 || W < 2 || Di < 2 || Hi < 2 || Wi < 2) return FS_ERR_SHAPE;
  if ((long long)D * H * W >= (1ll << 31) || (long long)Di * Hi * Wi >= (1ll << 31))
    return FS_ERR_SHAPE;
  p.B = B; p.C = C; p.D = D; p.H = H; p.W = W;
  p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  p.tilesH = fs::cdiv(H, TH);
  p.tilesW = fs::cdiv(W, TW);
  if ((long long)B * D * p.tilesH * p.tilesW >= (1ll << 31)) return FS_ERR_SHAPE;
  // fp32 like the reference: linspace step (end-start)/(steps-1); divisor (dim-1.0)/2.0
  p.stepD = 2.0f / (float)(D - 1);
  p.stepH = 2.0f / (float)(H - 1);
  p.stepW = 2.0f / (float)(W - 1);
  p.sD = ((float)Di - 1.0f) / 2.0f;
  p.sH = ((float)Hi - 1.0f) / 2.0f;
  p.sW = ((float)Wi - 1.0f) / 2.0f;
  return FS_OK;
}

constexpr int kTW = 32;


int launch_fwd(const W3Fwd& io, int npair, const float* flow, W3P& p, fs_stream_t stream) {
  const unsigned grid = (unsigned)((long long)p.B * p.D * p.tilesH * p.tilesW);
  p.flowC = 3 * npair;
  hipLaunchKernelGGL(warp3d_fwd_kernel<kTW>, dim3(grid, npair), dim3(NT), 0, (hipStream_t)stream, io,
                     flow, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int launch_bwd(const W3Bwd& io, int npair, bool with_gin, const float* flow, float* gflow, W3P& p,
               fs_stream_t stream) {
  const unsigned grid = (unsigned)((long long)p.B * p.D * p.tilesH * p.tilesW);
  p.flowC = 3 * npair;
  if (with_gin)
    hipLaunchKernelGGL((warp3d_bwd_kernel<kTW, true>), dim3(grid, npair), dim3(NT), 0,
                       (hipStream_t)stream, io, flow, gflow, p);
  else
    hipLaunchKernelGGL((warp3d_bwd_kernel<kTW, false>), dim3(grid, npair), dim3(NT), 0,
                       (hipStream_t)stream, io, flow, gflow, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // namespace

extern "C" int fs_warp3d_fwd(const float* in, const float* flow, float* out, int B, int C,
                             const int* in_dhw, int D, int H, int W, fs_stream_t stream) {
  FS_REQUIRE_PTR(in); FS_REQUIRE_PTR(flow); FS_REQUIRE_PTR(out);
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W, kTW);
  if (rc != FS_OK) return rc;
  W3Fwd io = {{in, nullptr}, {out, nullptr}};
  return launch_fwd(io, 1, flow, p, stream);
}

extern "C" int fs_warp3d_bwd(const float* in, const float* flow, const float* grad_out,
                             float* grad_in, float* grad_flow, int B, int C, const int* in_dhw,
                             int D, int H, int W, fs_stream_t stream) {
  FS_REQUIRE_PTR(in); FS_REQUIRE_PTR(flow); FS_REQUIRE_PTR(grad_out);
  if (grad_in == nullptr && grad_flow == nullptr) return FS_ERR_NULLPTR;
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W, kTW);
  if (rc != FS_OK) return rc;
  W3Bwd io = {{in, nullptr}, {grad_out, nullptr}, {grad_in, nullptr}};
  return launch_bwd(io, 1, grad_in != nullptr, flow, grad_flow, p, stream);
}

extern "C" int fs_warp3d_pair_fwd(const float* img0, const float* img1, const float* flow6,
                                  float* out0, float* out1, int B, int C, const int* in_dhw,
                                  int D, int H, int W, fs_stream_t stream) {
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow6);
  FS_REQUIRE_PTR(out0); FS_REQUIRE_PTR(out1);
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W, kTW);
  if (rc != FS_OK) return rc;
  W3Fwd io = {{img0, img1}, {out0, out1}};
  return launch_fwd(io, 2, flow6, p, stream);
}

extern "C" int fs_warp3d_pair_bwd(const float* img0, const float* img1, const float* flow6,
                                  const float* grad_out0, const float* grad_out1, float* grad_img0,
                                  float* grad_img1, float* grad_flow6, int B, int C,
                                  const int* in_dhw, int D, int H, int W, fs_stream_t stream) {
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow6);
  FS_REQUIRE_PTR(grad_out0); FS_REQUIRE_PTR(grad_out1);
  if ((grad_img0 == nullptr) != (grad_img1 == nullptr)) return FS_ERR_NULLPTR;  // both or neither
  if (grad_img0 == nullptr && grad_flow6 == nullptr) return FS_ERR_NULLPTR;
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W, kTW);
  if (rc != FS_OK) return rc;
  W3Bwd io = {{img0, img1}, {grad_out0, grad_out1}, {grad_img0, grad_img1}};
  return launch_bwd(io, 2, grad_img0 != nullptr, flow6, grad_flow6, p, stream);
}
