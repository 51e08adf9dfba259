// warp2d.hip -- the four 2-D bilinear backward warps of the reference (SURVEY §8 a1, a5, a6,
// a7, a11) as one templated gfx950 kernel pair.
//
// All four are "gather 4 corners at (x+u, y+v) under some coordinate convention"; they differ
// only in how (ix, iy) is derived from (x, y, u, v), in the padding rule and in an optional
// validity mask.  One thread owns one output pixel (lanes on x: flow reads, output stores and,
// for smooth flows, the corner gathers are all coalesced) and loops over the C channels that
// share its flow vector.  HBM-bound: 8 B flow + 4 B gather + 4 B store per pixel-channel.
//
//  RIFE    Flow-2D/model/warplayer.py:7-26        border, align_corners=True
//  PWC     UPFlow/model/pwc_modules.py:184-207    zeros,  align_corners=False, optional mask
//          UPFlow/utils/tools.py:1317-1361
//  PHOTO   Flow-2D/model/RIFE.py:244-262          zeros,  align_corners=False, grid=(x+u)*2/W-1
//  DILATED UPFlow/utils/tools.py:412-541          clamped indices, unclamped weights
#include "common.hpp"

namespace {

struct W2P {
  int B, C, H, W;              // extent of the flow == of the output
  int Hi, Wi;                  // extent of the sampled image (RIFE only may differ: the reference builds the
                               // grid from the flow's shape and normalises by the input's, warplayer.py:10-20)
  float stepH, stepW, sH, sW;  // RIFE: linspace steps over the FLOW dims and (n-1)/2 of the INPUT dims
  float dW, dH;                // PWC: max(W-1,1), max(H-1,1)
  float fW, fH;                // PHOTO: float(2/W), float(2/H)
  int flowC;                   // channels of the flow tensor: 2 (single) or 4 (IFNet pair)
};

// blockIdx.y selects the member of a pair (img0 with flow[:, :2], img1 with flow[:, 2:4]:
// Flow-2D/model/IFNet.py:191-192); the 4-channel flow tensors are used in place.
struct W2Fwd { const float* in[2]; float* out[2]; };
struct W2Bwd { const float* in[2]; const float* gout[2]; float* gin[2]; };

struct Samp2 {
  int x0, x1, y0, y1;      // addressing indices (always inside the image)
  int v00, v10, v01, v11;  // corner participates (ATen within_bounds)
  float ax, bx, ay, by;    // weights: corner (x0,y0) gets bx*by, (x1,y0) ax*by, ...
  float mx, my;            // d ix / d u, d iy / d v (chain to the flow)
};

template <int MODE>
__device__ __forceinline__ Samp2 w2_sample(const W2P& p, int b, int x, int y, float u, float v,
                                           const float* __restrict__ start) {
#pragma clang fp contract(off)
  Samp2 s;
  float ix, iy;
  if (MODE == FS_WARP2D_RIFE) {
    const float gx = fs::linspace_pm1(x, p.W, p.stepW) + u / p.sW;  // warplayer.py:12,18
    const float gy = fs::linspace_pm1(y, p.H, p.stepH) + v / p.sH;  // warplayer.py:14,19
    ix = ((gx + 1.0f) / 2.0f) * (float)(p.Wi - 1);
    iy = ((gy + 1.0f) / 2.0f) * (float)(p.Hi - 1);
    float cx, cy;
    ix = fs::clip_border(ix, p.Wi, &cx);
    iy = fs::clip_border(iy, p.Hi, &cy);
    s.mx = cx * ((float)(p.Wi - 1) / 2.0f) / p.sW;
    s.my = cy * ((float)(p.Hi - 1) / 2.0f) / p.sH;
  } else if (MODE == FS_WARP2D_PWC) {
    const float vx = 2.0f * ((float)x + u) / p.dW - 1.0f;  // pwc_modules.py:199-200
    const float vy = 2.0f * ((float)y + v) / p.dH - 1.0f;
    ix = ((vx + 1.0f) * (float)p.W - 1.0f) / 2.0f;  // unnormalize, align_corners=False
    iy = ((vy + 1.0f) * (float)p.H - 1.0f) / 2.0f;
    s.mx = ((float)p.W / 2.0f) * (2.0f / p.dW);
    s.my = ((float)p.H / 2.0f) * (2.0f / p.dH);
  } else if (MODE == FS_WARP2D_PHOTO) {
    const float gx = (u + (float)x) * p.fW - 1.0f;  // RIFE.py:256-259
    const float gy = (v + (float)y) * p.fH - 1.0f;
    ix = ((gx + 1.0f) * (float)p.W - 1.0f) / 2.0f;
    iy = ((gy + 1.0f) * (float)p.H - 1.0f) / 2.0f;
    s.mx = ((float)p.W / 2.0f) * p.fW;
    s.my = ((float)p.H / 2.0f) * p.fH;
  } else {  // DILATED
    const float sx = start ? start[2 * b + 0] : 0.0f;
    const float sy = start ? start[2 * b + 1] : 0.0f;
    ix = ((float)x + sx) + u;  // tools.py:409,538
    iy = ((float)y + sy) + v;
    s.mx = 1.0f;
    s.my = 1.0f;
  }
  // keep the float->int conversion defined for wild flows; outside [-2, size+1] every
  // corner is out of range (zeros modes) or clamps to the same edge (DILATED)
  const float fx = fminf(fmaxf(ix, -2.0f), (float)p.Wi + 1.0f);
  const float fy = fminf(fmaxf(iy, -2.0f), (float)p.Hi + 1.0f);
  const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
  const int x1 = x0 + 1, y1 = y0 + 1;
  if (MODE == FS_WARP2D_DILATED) {
    s.x0 = min(max(x0, 0), p.Wi - 1); s.x1 = min(max(x1, 0), p.Wi - 1);
    s.y0 = min(max(y0, 0), p.Hi - 1); s.y1 = min(max(y1, 0), p.Hi - 1);
    s.v00 = s.v10 = s.v01 = s.v11 = 1;
    s.ax = ix - (float)s.x0; s.bx = (float)s.x1 - ix;  // tools.py:504-507 (clamped corners)
    s.ay = iy - (float)s.y0; s.by = (float)s.y1 - iy;
  } else {
    const int x0ok = (x0 >= 0 && x0 < p.Wi), x1ok = (x1 >= 0 && x1 < p.Wi);
    const int y0ok = (y0 >= 0 && y0 < p.Hi), y1ok = (y1 >= 0 && y1 < p.Hi);
    s.v00 = x0ok & y0ok; s.v10 = x1ok & y0ok; s.v01 = x0ok & y1ok; s.v11 = x1ok & y1ok;
    s.x0 = min(max(x0, 0), p.Wi - 1); s.x1 = min(max(x1, 0), p.Wi - 1);
    s.y0 = min(max(y0, 0), p.Hi - 1); s.y1 = min(max(y1, 0), p.Hi - 1);
    s.ax = ix - (float)x0; s.bx = (float)x1 - ix;
    s.ay = iy - (float)y0; s.by = (float)y1 - iy;
  }
  return s;
}

// validity mask of WarpingLayer_no_div: grid_sample(ones) >= 1.0 (pwc_modules.py:202-206)
__device__ __forceinline__ float w2_mask(const Samp2& s) {
#pragma clang fp contract(off)
  float m = 0.0f;
  if (s.v00) m += s.bx * s.by;
  if (s.v10) m += s.ax * s.by;
  if (s.v01) m += s.bx * s.ay;
  if (s.v11) m += s.ax * s.ay;
  return (m >= 1.0f) ? 1.0f : 0.0f;
}

// A workgroup is PX = 256 / SL pixels x SL channel slices (thread t: pixel t % PX, slice t / PX; lanes run along
// x).  SL = 1 for images and fine pyramid levels; the coarse UPFlow levels have few pixels and many channels
// ([32,196,3,8]: 768 pixels, 196 channels each) -- with one thread per pixel the launch was 3 workgroups walking
// 196 channels (and 4 atomics per channel) one after the other, 157 us per backward launch; with SL = 16 it is 48
// workgroups of 12-channel walks.
template <int MODE, bool MASK, int SL>
__global__ __launch_bounds__(256) void warp2d_fwd_kernel(W2Fwd io, const float* __restrict__ flow,
                                                         const float* __restrict__ start, W2P p) {
  constexpr int PX = 256 / SL;
  const float* __restrict__ in = io.in[blockIdx.y];
  float* __restrict__ out = io.out[blockIdx.y];
  const int HW = p.H * p.W;
  const long long n = (long long)p.B * HW;
  const int px = threadIdx.x % PX, sl = threadIdx.x / PX;
  for (long long i = (long long)blockIdx.x * PX + px; i < n; i += (long long)gridDim.x * PX) {
    const int b = (int)(i / HW);
    const int r = (int)(i - (long long)b * HW);
    const int y = r / p.W, x = r - y * p.W;
    const float* fb = flow + ((size_t)b * p.flowC + 2 * blockIdx.y) * HW;
    const Samp2 s = w2_sample<MODE>(p, b, x, y, fb[r], fb[HW + r], start);
    const float mk = MASK ? w2_mask(s) : 1.0f;
    const int o00 = s.y0 * p.Wi + s.x0, o10 = s.y0 * p.Wi + s.x1;
    const int o01 = s.y1 * p.Wi + s.x0, o11 = s.y1 * p.Wi + s.x1;
    for (int c = sl; c < p.C; c += SL) {
#pragma clang fp contract(off)
      const float* ic = in + ((size_t)b * p.C + c) * ((size_t)p.Hi * p.Wi);
      const float q00 = s.v00 ? ic[o00] : 0.f, q10 = s.v10 ? ic[o10] : 0.f;
      const float q01 = s.v01 ? ic[o01] : 0.f, q11 = s.v11 ? ic[o11] : 0.f;
      float acc;
      if (MODE == FS_WARP2D_DILATED) {  // wa*Ia + wb*Ib + wc*Ic + wd*Id (tools.py:508)
        acc = (s.bx * s.by) * q00 + (s.bx * s.ay) * q01 + (s.ax * s.by) * q10 + (s.ax * s.ay) * q11;
      } else {  // ATen: nw, ne, sw, se
        acc = q00 * (s.bx * s.by);
        acc += q10 * (s.ax * s.by);
        acc += q01 * (s.bx * s.ay);
        acc += q11 * (s.ax * s.ay);
      }
      out[((size_t)b * p.C + c) * HW + r] = MASK ? acc * mk : acc;
    }
  }
}

// grad_flow sums over the channels: the SL slices of a pixel leave their partial sums in LDS and slice 0 adds them
// in slice order (deterministic; with SL = 1 it is the plain per-thread running sum).  grad_in stays a scatter with
// float atomics.
template <int MODE, bool MASK, bool WITH_GIN, int SL>
__global__ __launch_bounds__(256) void warp2d_bwd_kernel(W2Bwd io, const float* __restrict__ flow,
                                                         const float* __restrict__ start,
                                                         float* __restrict__ gflow, W2P p) {
  constexpr int PX = 256 / SL;
  __shared__ float red[SL > 1 ? 2 * 256 : 1];
  const float* __restrict__ in = io.in[blockIdx.y];
  const float* __restrict__ gout = io.gout[blockIdx.y];
  float* __restrict__ gin = io.gin[blockIdx.y];
  const int HW = p.H * p.W;
  const long long n = (long long)p.B * HW;
  const int px = threadIdx.x % PX, sl = threadIdx.x / PX;
  // every thread of the workgroup runs the same number of rounds (barriers inside)
  for (long long base = (long long)blockIdx.x * PX; base < n; base += (long long)gridDim.x * PX) {
    const long long i = base + px;
    const bool live = i < n;
    float gx = 0.f, gy = 0.f, mx = 0.f, my = 0.f;
    int b = 0, r = 0;
    if (live) {
      b = (int)(i / HW);
      r = (int)(i - (long long)b * HW);
      const int y = r / p.W, x = r - y * p.W;
      const float* fb = flow + ((size_t)b * p.flowC + 2 * blockIdx.y) * HW;
      const Samp2 s = w2_sample<MODE>(p, b, x, y, fb[r], fb[HW + r], start);
      const float mk = MASK ? w2_mask(s) : 1.0f;
      const int o00 = s.y0 * p.Wi + s.x0, o10 = s.y0 * p.Wi + s.x1;
      const int o01 = s.y1 * p.Wi + s.x0, o11 = s.y1 * p.Wi + s.x1;
      mx = s.mx; my = s.my;
      for (int c = sl; c < p.C; c += SL) {
        const size_t pl = ((size_t)b * p.C + c) * ((size_t)p.Hi * p.Wi);
        const float g = gout[((size_t)b * p.C + c) * HW + r] * mk;
        const float* ic = in + pl;
        const float q00 = s.v00 ? ic[o00] : 0.f, q10 = s.v10 ? ic[o10] : 0.f;
        const float q01 = s.v01 ? ic[o01] : 0.f, q11 = s.v11 ? ic[o11] : 0.f;
        gx += g * (s.by * (q10 - q00) + s.ay * (q11 - q01));
        gy += g * (s.bx * (q01 - q00) + s.ax * (q11 - q10));
        if (WITH_GIN) {
          float* gc = gin + pl;
          if (s.v00) atomicAdd(gc + o00, g * (s.bx * s.by));
          if (s.v10) atomicAdd(gc + o10, g * (s.ax * s.by));
          if (s.v01) atomicAdd(gc + o01, g * (s.bx * s.ay));
          if (s.v11) atomicAdd(gc + o11, g * (s.ax * s.ay));
        }
      }
    }
    if (gflow != nullptr) {
      if (SL > 1) {
        red[threadIdx.x] = gx;
        red[256 + threadIdx.x] = gy;
        __syncthreads();
        if (sl == 0) {
          gx = red[px]; gy = red[256 + px];
          for (int k = 1; k < SL; ++k) { gx += red[k * PX + px]; gy += red[256 + k * PX + px]; }
        }
        __syncthreads();
      }
      if (live && sl == 0) {
        float* gb = gflow + ((size_t)b * p.flowC + 2 * blockIdx.y) * HW;
        gb[r] = gx * mx;
        gb[HW + r] = gy * my;
      }
    }
  }
}

// grad_in of the small planes (UPFlow's feature pyramid: <= 38 x 113) without global atomics: a workgroup OWNS the
// planes (b, c0 .. c0+nc) -- whole planes fit in LDS -- scatters every pixel's four corner contributions into them
// with LDS float adds and then adds the finished planes to grad_in with coalesced, non-atomic read-modify-writes (it
// is the only writer of those planes in this launch).  Round 3's path issued four global float atomics per (pixel,
// channel) -- 0.06 of the HBM roof on the dominant hot-path kernel of the C3 step; atomics execute at the memory side
// at ~1.3 TB/s of added bytes chip-wide whatever their locality (MI355X_MICROARCH.md).  grad_flow comes from the
// gather kernel above (WITH_GIN = false: deterministic, no atomics); this kernel only re-computes the sample
// positions (flow reads hit L2) and streams grad_out.
template <int MODE, bool MASK>
__global__ __launch_bounds__(256) void warp2d_gin_plane_kernel(W2Bwd io, const float* __restrict__ flow,
                                                               const float* __restrict__ start, W2P p, int NC, int CG) {
  extern __shared__ float acc[];  // [nc][Hi * Wi]
  const float* __restrict__ gout = io.gout[blockIdx.y];
  float* __restrict__ gin = io.gin[blockIdx.y];
  const int HW = p.H * p.W, HWi = p.Hi * p.Wi;
  const int b = blockIdx.x / CG, c0 = (blockIdx.x - b * CG) * NC;
  const int nc = min(NC, p.C - c0);
  for (int i = threadIdx.x; i < nc * HWi; i += 256) acc[i] = 0.f;
  __syncthreads();
  const float* fb = flow + ((size_t)b * p.flowC + 2 * blockIdx.y) * HW;
  const float* gb = gout + ((size_t)b * p.C + c0) * HW;
  for (int r = threadIdx.x; r < HW; r += 256) {
    const int y = r / p.W, x = r - y * p.W;
    const Samp2 s = w2_sample<MODE>(p, b, x, y, fb[r], fb[HW + r], start);
    const float mk = MASK ? w2_mask(s) : 1.0f;
    const int o00 = s.y0 * p.Wi + s.x0, o10 = s.y0 * p.Wi + s.x1;
    const int o01 = s.y1 * p.Wi + s.x0, o11 = s.y1 * p.Wi + s.x1;
    const float w00 = s.bx * s.by, w10 = s.ax * s.by, w01 = s.bx * s.ay, w11 = s.ax * s.ay;
    for (int c = 0; c < nc; ++c) {
      const float g = gb[(size_t)c * HW + r] * mk;
      float* a = acc + c * HWi;
      if (s.v00) atomicAdd(a + o00, g * w00);
      if (s.v10) atomicAdd(a + o10, g * w10);
      if (s.v01) atomicAdd(a + o01, g * w01);
      if (s.v11) atomicAdd(a + o11, g * w11);
    }
  }
  __syncthreads();
  float* go = gin + ((size_t)b * p.C + c0) * HWi;
  for (int i = threadIdx.x; i < nc * HWi; i += 256) go[i] += acc[i];
}

// ---- forward-backward occlusion check + outgoing mask -------------------------
// UPFlow/utils/tools.py:592-630 (_forward_backward_occ_check), :683-709 (torch_outgoing_occ_check),
// :711-719 (torch_get_obj_occ_check), dispatch :560-590.  The reference runs two torch_warp calls
// (a6) and ~25 elementwise passes over [B,1,H,W] / [B,2,H,W] tensors; here one thread owns one
// pixel, reads both flows once, gathers the other flow at its PWC sample position (same
// w2_sample<PWC> arithmetic as fs_warp2d_fwd) and writes both masks: 16 B read + 8 B written
// per pixel.  Comparisons only -- the masks carry no gradient in the reference either (.float()
// of a bool).
__device__ __forceinline__ void occ_gather(const Samp2& s, const float* __restrict__ f, int W, int HW,
                                           float* ox, float* oy) {
#pragma clang fp contract(off)
  const int o00 = s.y0 * W + s.x0, o10 = s.y0 * W + s.x1;
  const int o01 = s.y1 * W + s.x0, o11 = s.y1 * W + s.x1;
  float r[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float* ic = f + (size_t)c * HW;
    const float q00 = s.v00 ? ic[o00] : 0.f, q10 = s.v10 ? ic[o10] : 0.f;
    const float q01 = s.v01 ? ic[o01] : 0.f, q11 = s.v11 ? ic[o11] : 0.f;
    float acc = q00 * (s.bx * s.by);
    acc += q10 * (s.ax * s.by);
    acc += q01 * (s.bx * s.ay);
    acc += q11 * (s.ax * s.ay);
    r[c] = acc;
  }
  *ox = r[0]; *oy = r[1];
}

__device__ __forceinline__ float occ_outgoing(float px, float py, int H, int W) {
  // tools.py:700-707: ones, zeroed where pos > size-1 or pos < 0 (NaN compares false -> stays 1)
  return (px > (float)(W - 1) || px < 0.f || py > (float)(H - 1) || py < 0.f) ? 0.f : 1.f;
}

__global__ __launch_bounds__(256) void occ_check2d_kernel(const float* __restrict__ flow_f,
                                                          const float* __restrict__ flow_b,
                                                          float* __restrict__ occ_f,
                                                          float* __restrict__ occ_b, W2P p, float alpha1,
                                                          float alpha2s, int mode) {
#pragma clang fp contract(off)
  const int HW = p.H * p.W;
  const long long n = (long long)p.B * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int r = (int)(i - (long long)b * HW);
    const int y = r / p.W, x = r - y * p.W;
    const float* ff = flow_f + (size_t)b * 2 * HW;
    const float* fb = flow_b + (size_t)b * 2 * HW;
    const float uf = ff[r], vf = ff[HW + r], ub = fb[r], vb = fb[HW + r];
    float of = 1.f, ob = 1.f;
    if (mode != FS_OCC_OUT) {
      // length_sq_v0: sum_c pow(x_c^2, 0.5) = |x_0| + |x_1|   (tools.py:596-601)
      const float mag = (fabsf(uf) + fabsf(vf)) + (fabsf(ub) + fabsf(vb));
      float wx, wy;
      occ_gather(w2_sample<FS_WARP2D_PWC>(p, b, x, y, uf, vf, nullptr), fb, p.W, HW, &wx, &wy);
      const float dfx = uf + wx, dfy = vf + wy;  // flow_fw + warp(flow_bw, flow_fw)
      occ_gather(w2_sample<FS_WARP2D_PWC>(p, b, x, y, ub, vb, nullptr), ff, p.W, HW, &wx, &wy);
      const float dbx = ub + wx, dby = vb + wy;
      const float thresh = alpha1 * mag + alpha2s;
      of = ((fabsf(dfx) + fabsf(dfy)) < thresh) ? 1.f : 0.f;  // 0 = occluded
      ob = ((fabsf(dbx) + fabsf(dby)) < thresh) ? 1.f : 0.f;
    }
    if (mode != FS_OCC_ALL) {
      const float outf = occ_outgoing((float)x + uf, (float)y + vf, p.H, p.W);
      const float outb = occ_outgoing((float)x + ub, (float)y + vb, p.H, p.W);
      if (mode == FS_OCC_OUT) { of = outf; ob = outb; }
      else {  // obj: visible, or hidden only because it leaves the frame (tools.py:711-719)
        of = (of == 1.f || outf == 0.f) ? 1.f : 0.f;
        ob = (ob == 1.f || outb == 0.f) ? 1.f : 0.f;
      }
    }
    occ_f[i] = of;
    occ_b[i] = ob;
  }
}

int make_params(W2P& p, int B, int C, int H, int W, int mode, const int* in_hw = nullptr) {
  const int Hi = in_hw ? in_hw[0] : H, Wi = in_hw ? in_hw[1] : W;
  if (B < 1 || C < 1 || H < 1 || W < 1 || Hi < 1 || Wi < 1) return FS_ERR_SHAPE;
  if ((Hi != H || Wi != W) && mode != FS_WARP2D_RIFE) return FS_ERR_ARG;  // only RIFE defines it
  if (mode == FS_WARP2D_RIFE && (H < 2 || W < 2 || Hi < 2 || Wi < 2)) return FS_ERR_SHAPE;  // (dim-1)/2 divisor
  if ((long long)H * W >= (1ll << 31) || (long long)Hi * Wi >= (1ll << 31)) return FS_ERR_SHAPE;
  p.B = B; p.C = C; p.H = H; p.W = W; p.Hi = Hi; p.Wi = Wi;
  p.stepH = 2.0f / (float)(H - 1);
  p.stepW = 2.0f / (float)(W - 1);
  p.sH = ((float)Hi - 1.0f) / 2.0f;
  p.sW = ((float)Wi - 1.0f) / 2.0f;
  p.dW = (float)(W - 1 > 1 ? W - 1 : 1);
  p.dH = (float)(H - 1 > 1 ? H - 1 : 1);
  p.fW = (float)(2.0 / (double)W);  // python float 2/w, then FloatTensor (RIFE.py:258)
  p.fH = (float)(2.0 / (double)H);
  return FS_OK;
}

unsigned grid_for(const W2P& p, int slices = 1) {
  const long long n = (long long)p.B * p.H * p.W;
  const int px = 256 / slices;
  const long long g = (n + px - 1) / px;
  return (unsigned)(g < 16384 ? g : 16384);  // grid-stride above 64 blocks per CU
}

// channel slices per pixel: enough threads to fill the chip (>= 2 waves per SIMD) when the pixels alone do not
int slices_for(const W2P& p) {
  const long long n = (long long)p.B * p.H * p.W;
  if (n >= 131072 || p.C < 4) return 1;
  if (n * 4 >= 131072 || p.C < 16) return 4;
  return 16;
}

template <int MODE, bool MASK>
void launch_fwd(const W2Fwd& io, int npair, const float* flow, const float* start, W2P& p,
                hipStream_t st) {
  p.flowC = 2 * npair;
  const int sl = slices_for(p);
  const dim3 g(grid_for(p, sl), npair);
  if (sl == 1) hipLaunchKernelGGL((warp2d_fwd_kernel<MODE, MASK, 1>), g, dim3(256), 0, st, io, flow, start, p);
  else if (sl == 4) hipLaunchKernelGGL((warp2d_fwd_kernel<MODE, MASK, 4>), g, dim3(256), 0, st, io, flow, start, p);
  else hipLaunchKernelGGL((warp2d_fwd_kernel<MODE, MASK, 16>), g, dim3(256), 0, st, io, flow, start, p);
}

template <int MODE, bool MASK>
void launch_bwd(const W2Bwd& io, int npair, const float* flow, const float* start, float* gflow,
                W2P& p, hipStream_t st) {
  p.flowC = 2 * npair;
  const int sl = slices_for(p);
  const dim3 g(grid_for(p, sl), npair);
#define FS_W2_BWD(GIN, SLN) \
  hipLaunchKernelGGL((warp2d_bwd_kernel<MODE, MASK, GIN, SLN>), g, dim3(256), 0, st, io, flow, start, gflow, p)
  // planes that fit LDS: grad_in by plane-owning workgroups (no global atomics), grad_flow by the gather kernel
  const long long plane = (long long)p.Hi * p.Wi * 4;
  // (plane-owning workgroups add their planes with plain read-modify-writes: every grad_in of the launch must be present
  // and they must be different tensors -- aliased or partly missing ones keep the atomic kernel; include/flowsci_hip.h)
  bool own = io.gin[0] != nullptr;
  for (int k = 1; k < npair; ++k) own = own && io.gin[k] != nullptr && io.gin[k] != io.gin[0];
  if (own && plane <= 48 * 1024 && (long long)p.B * p.C < (1ll << 31)) {  // (grid.x = B * channel groups <= B * C)
    if (gflow != nullptr) {
      if (sl == 1) FS_W2_BWD(false, 1); else if (sl == 4) FS_W2_BWD(false, 4); else FS_W2_BWD(false, 16);
    }
    long long nc = 48 * 1024 / plane;                      // planes per workgroup: as many as fit ...
    const long long fill = (long long)p.C * p.B / 512;     // ... but keep >= 512 workgroups when the layer allows
    if (nc > fill) nc = fill;
    if (nc > p.C) nc = p.C;
    if (nc < 1) nc = 1;
    const int CG = (int)((p.C + nc - 1) / nc);
    hipLaunchKernelGGL((warp2d_gin_plane_kernel<MODE, MASK>), dim3((unsigned)(p.B * CG), npair), dim3(256),
                       (size_t)(nc * plane), st, io, flow, start, p, (int)nc, CG);
    return;
  }
  if (io.gin[0] != nullptr) {
    if (sl == 1) FS_W2_BWD(true, 1); else if (sl == 4) FS_W2_BWD(true, 4); else FS_W2_BWD(true, 16);
  } else {
    if (sl == 1) FS_W2_BWD(false, 1); else if (sl == 4) FS_W2_BWD(false, 4); else FS_W2_BWD(false, 16);
  }
#undef FS_W2_BWD
}

int dispatch_fwd(const W2Fwd& io, int npair, const float* flow, const float* start, W2P& p, int mode,
                 int with_mask, hipStream_t st) {
  switch (mode) {
    case FS_WARP2D_RIFE: launch_fwd<FS_WARP2D_RIFE, false>(io, npair, flow, start, p, st); break;
    case FS_WARP2D_PWC:
      if (with_mask) launch_fwd<FS_WARP2D_PWC, true>(io, npair, flow, start, p, st);
      else launch_fwd<FS_WARP2D_PWC, false>(io, npair, flow, start, p, st);
      break;
    case FS_WARP2D_PHOTO: launch_fwd<FS_WARP2D_PHOTO, false>(io, npair, flow, start, p, st); break;
    default: launch_fwd<FS_WARP2D_DILATED, false>(io, npair, flow, start, p, st); break;
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int dispatch_bwd(const W2Bwd& io, int npair, const float* flow, const float* start, float* gflow,
                 W2P& p, int mode, int with_mask, hipStream_t st) {
  switch (mode) {
    case FS_WARP2D_RIFE: launch_bwd<FS_WARP2D_RIFE, false>(io, npair, flow, start, gflow, p, st); break;
    case FS_WARP2D_PWC:
      if (with_mask) launch_bwd<FS_WARP2D_PWC, true>(io, npair, flow, start, gflow, p, st);
      else launch_bwd<FS_WARP2D_PWC, false>(io, npair, flow, start, gflow, p, st);
      break;
    case FS_WARP2D_PHOTO: launch_bwd<FS_WARP2D_PHOTO, false>(io, npair, flow, start, gflow, p, st); break;
    default: launch_bwd<FS_WARP2D_DILATED, false>(io, npair, flow, start, gflow, p, st); break;
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int check_mode(int mode, int with_mask, const float* start) {
  if (mode < FS_WARP2D_RIFE || mode > FS_WARP2D_DILATED) return FS_ERR_ARG;
  if (with_mask && mode != FS_WARP2D_PWC) return FS_ERR_ARG;
  if (start != nullptr && mode != FS_WARP2D_DILATED) return FS_ERR_ARG;
  return FS_OK;
}

}  // namespace

extern "C" int fs_warp2d_fwd(const float* in, const float* flow, const float* start, float* out,
                             int B, int C, const int* in_hw, int H, int W, int mode, int with_mask,
                             fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(in); FS_REQUIRE_PTR(flow); FS_REQUIRE_PTR(out);
  int rc = check_mode(mode, with_mask, start);
  if (rc != FS_OK) return rc;
  W2P p;
  rc = make_params(p, B, C, H, W, mode, in_hw);
  if (rc != FS_OK) return rc;
  W2Fwd io = {{in, nullptr}, {out, nullptr}};
  return dispatch_fwd(io, 1, flow, start, p, mode, with_mask, (hipStream_t)stream);
}

extern "C" int fs_warp2d_bwd(const float* in, const float* flow, const float* start,
                             const float* grad_out, float* grad_in, float* grad_flow, int B, int C,
                             const int* in_hw, int H, int W, int mode, int with_mask, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(in); FS_REQUIRE_PTR(flow); FS_REQUIRE_PTR(grad_out);
  if (grad_in == nullptr && grad_flow == nullptr) return FS_ERR_NULLPTR;
  int rc = check_mode(mode, with_mask, start);
  if (rc != FS_OK) return rc;
  W2P p;
  rc = make_params(p, B, C, H, W, mode, in_hw);
  if (rc != FS_OK) return rc;
  W2Bwd io = {{in, nullptr}, {grad_out, nullptr}, {grad_in, nullptr}};
  return dispatch_bwd(io, 1, flow, start, grad_flow, p, mode, with_mask, (hipStream_t)stream);
}

extern "C" int fs_warp2d_pair_fwd(const float* img0, const float* img1, const float* flow4,
                                  float* out0, float* out1, int B, int C, const int* in_hw, int H, int W,
                                  int mode, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow4);
  FS_REQUIRE_PTR(out0); FS_REQUIRE_PTR(out1);
  int rc = check_mode(mode, 0, nullptr);
  if (rc != FS_OK) return rc;
  W2P p;
  rc = make_params(p, B, C, H, W, mode, in_hw);
  if (rc != FS_OK) return rc;
  W2Fwd io = {{img0, img1}, {out0, out1}};
  return dispatch_fwd(io, 2, flow4, nullptr, p, mode, 0, (hipStream_t)stream);
}

extern "C" int fs_warp2d_pair_bwd(const float* img0, const float* img1, const float* flow4,
                                  const float* grad_out0, const float* grad_out1, float* grad_img0,
                                  float* grad_img1, float* grad_flow4, int B, int C, const int* in_hw, int H,
                                  int W, int mode, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow4);
  FS_REQUIRE_PTR(grad_out0); FS_REQUIRE_PTR(grad_out1);
  if ((grad_img0 == nullptr) != (grad_img1 == nullptr)) return FS_ERR_NULLPTR;
  if (grad_img0 == nullptr && grad_flow4 == nullptr) return FS_ERR_NULLPTR;
  int rc = check_mode(mode, 0, nullptr);
  if (rc != FS_OK) return rc;
  W2P p;
  rc = make_params(p, B, C, H, W, mode, in_hw);
  if (rc != FS_OK) return rc;
  W2Bwd io = {{img0, img1}, {grad_out0, grad_out1}, {grad_img0, grad_img1}};
  return dispatch_bwd(io, 2, flow4, nullptr, grad_flow4, p, mode, 0, (hipStream_t)stream);
}

extern "C" int fs_occ_check2d(const float* flow_f, const float* flow_b, float* occ_f, float* occ_b,
                              int B, int H, int W, float alpha1, float alpha2_over_scale, int mode,
                              fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(flow_f); FS_REQUIRE_PTR(flow_b); FS_REQUIRE_PTR(occ_f); FS_REQUIRE_PTR(occ_b);
  if (mode < FS_OCC_ALL || mode > FS_OCC_OUT) return FS_ERR_ARG;
  W2P p;
  int rc = make_params(p, B, 2, H, W, FS_WARP2D_PWC);
  if (rc != FS_OK) return rc;
  p.flowC = 2;
  hipLaunchKernelGGL(occ_check2d_kernel, dim3(grid_for(p)), dim3(256), 0, (hipStream_t)stream, flow_f,
                     flow_b, occ_f, occ_b, p, alpha1, alpha2_over_scale, mode);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
