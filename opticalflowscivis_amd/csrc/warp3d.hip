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
// MI355X design: one workgroup owns an output tile of 64 h x 32 w and walks 4 consecutive d.
//   phase 1  the flow tile (3 x 64 x 32) is read with lanes on w (float4, 128-B rows) and parked
//            TRANSPOSED in LDS ([w][h], +1 pad: both access directions are conflict-free);
//   phase 2  lanes switch to h: lane l handles h0+l, each wave a slice of the w range.  The corner
//            gathers of one wave-instruction now hit ~256 contiguous input bytes (row iy, plane
//            iz); x0/x0+1 corner pairs travel as one 8-byte load; results go back to LDS;
//   phase 3  lanes back on w: coalesced float4 store of the output tile; the next slice's flow
//            tile is already in flight.
// HBM traffic is the algorithmic 20 B/voxel (12 flow + 4 gather + 4 store): measured with
// FETCH_SIZE / WRITE_SIZE (profiles/).  Tiles of consecutive d are walked by the same workgroup,
// so the iy / iy+1 row pairs they share hit in L1/L2.
//
// Backward (grad_flow, optional grad_in): same tiling; coordinates and the 8 corners are
// recomputed (nothing is saved by forward), grad_flow leaves through the LDS transpose,
// grad_in (only when the caller asks for it) by float atomics.
#include <stdint.h>

#include <type_traits>

#include "common.hpp"

namespace {

constexpr int TH = 64;   // tile extent along h = one wave of lanes in phase 2

struct W3P {
  int B, C, D, H, W;          // batch, image channels, extent of the flow == of the output
  int Di, Hi, Wi;             // extent of the sampled volume (the reference lets it differ:
                              // the grid comes from the flow's shape, warplayer.py:11-22)
  int tilesH, tilesW;
  int dc, nDC;                // d-slices per workgroup, number of d-chunks
  float stepD, stepH, stepW;  // linspace steps 2/(n-1) over the FLOW dims
  float rD, rH, rW;           // 1 / ((n-1)/2) of the INPUT dims (warplayer.py:24-26)
  float mD, mH, mW;           // (n-1) of the INPUT dims as float
  unsigned rowB, planeB;      // input row / plane pitch in bytes
  int flowC;                  // channels of the flow tensor: 3 (single) or 6 (IFNet pair)
};

// blockIdx.y selects the member of a pair: IFNet warps img0 with flow[:, :3] and img1 with
// flow[:, 3:6] (Flow-3D/model/IFNet.py:190-191); one launch serves both and reads / writes the
// 6-channel flow tensors in place (no slicing copies).
struct W3Fwd { const float* in[2]; float* out[2]; };
struct W3Bwd {
  const float* in[2]; const float* gout[2]; float* gin[2];
  long long gbs[2];  // batch stride of gout in floats (0 = dense, C*D*H*W): the gradient of a warped frame may arrive as
                     // a channel slice of a wider tensor (torch.cat's backward of the next block's input)
};

// Fused "flow = prev + scale * trilinear_upsample(delta)" producer of the forward kernel (SURVEY §8f.1,
// Flow-3D/model/IFNet.py:118 followed by :190-191): the flow tile of a slice is formed in registers from
// the low-resolution 6-channel delta, written ONCE to `fout` (the accumulated flow has three more
// consumers) and handed to the warp through the same LDS transpose -- the warp never re-reads it from HBM.
struct UpP {
  const float* small;  // [B, 6, Ds, Hs, Ws] head output at the block's working resolution
  const float* prev;   // [B, 6, D, H, W] running flow, or nullptr (block 0)
  float* fout;         // [B, 6, D, H, W] accumulated flow
  int Ds, Hs, Ws;
  float rs, scale;     // 1 / factor, flow scale (= factor)
};

// The kernels are VALU-issue-bound once HBM traffic is at the algorithmic minimum (measured:
// FETCH/WRITE_SIZE == algorithmic bytes, SQ_ACTIVE_INST_VALU ~ 90 % of the kernel's cycles at
// ~180 VALU instructions per voxel), so the per-voxel instruction count is the budget:
//  * flow / ((dim-1)/2) is a multiply by the reciprocal (<= 1 ulp from the reference's divide,
//    ~1e-7 px), the border clip is one v_med3, integer index math is 24-bit (full rate),
//    corner addresses are 32-bit byte offsets off a scalar base (saddr form);
//  * the +1 corners outside the volume need no masking: after the clip, a coordinate on the far
//    border has floor == coordinate, so the +1 weight is exactly 0 -- the index is only kept
//    in range (offset delta 0) so the load is legal;
//  * the blend is three nested lerps (14 instructions) instead of ATen's 8 weights x 8 products.
struct Samp3 {
  float ax, ay, az;        // fractional parts = weights of the +1 corners
  float mx, my, mz;        // border-clip gradient multipliers (0 or 1); backward only
  unsigned o000;           // byte offset of corner (z0, y0, x0)
  unsigned dx, dy, dz;     // byte deltas to the +1 corners (0 when that corner is outside)
};

__device__ __forceinline__ float w3_unnorm(float g, float m) {
#pragma clang fp contract(off)
  return ((g + 1.0f) * 0.5f) * m;  // grid_sampler_unnormalize, align_corners=True
}

template <bool BWD>
__device__ __forceinline__ Samp3 w3_sample(const W3P& p, float lin_h, float lin_d, float lin_w,
                                           float f0, float f1, float f2) {
  Samp3 s;
  float ix, iy, iz;
  {
#pragma clang fp contract(off)
    // warplayer.py:15-26: channel 0 pairs the H-linspace with F0 and addresses input dim W, ...
    ix = w3_unnorm(lin_h + f0 * p.rH, p.mW);
    iy = w3_unnorm(lin_d + f1 * p.rD, p.mH);
    iz = w3_unnorm(lin_w + f2 * p.rW, p.mD);
  }
  if (BWD) {  // ATen clip_coordinates_set_grad: zero gradient on and outside the border
    s.mx = (ix > 0.0f && ix < p.mW) ? 1.0f : 0.0f;
    s.my = (iy > 0.0f && iy < p.mH) ? 1.0f : 0.0f;
    s.mz = (iz > 0.0f && iz < p.mD) ? 1.0f : 0.0f;
  }
  ix = __builtin_amdgcn_fmed3f(ix, 0.0f, p.mW);
  iy = __builtin_amdgcn_fmed3f(iy, 0.0f, p.mH);
  iz = __builtin_amdgcn_fmed3f(iz, 0.0f, p.mD);
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  s.ax = ix - fx; s.ay = iy - fy; s.az = iz - fz;
  const unsigned x0 = (unsigned)fx, y0 = (unsigned)fy, z0 = (unsigned)fz;
  s.o000 = (__umul24(__umul24(z0, (unsigned)p.Hi) + y0, (unsigned)p.Wi) + x0) * 4u;
  s.dx = (fx < p.mW) ? 4u : 0u;
  s.dy = (fy < p.mH) ? p.rowB : 0u;
  s.dz = (fz < p.mD) ? p.planeB : 0u;
  return s;
}

__device__ __forceinline__ float ldb(const float* __restrict__ base, unsigned off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off);
}

// the 8 corners v[z][y][x]
struct Corners {
  float v000, v001, v010, v011, v100, v101, v110, v111;
};

// The texture-address path retires 4 lanes per clock whatever the access width, so the 8 dword
// gathers per voxel (not HBM, not L1 hit rate) bound the kernel.  The x0 / x0+1 corners are
// adjacent in memory: fetch each pair with ONE 8-byte load (4-byte aligned, which global loads
// allow).  On the far border (dx == 0, weight of the +1 corner exactly 0) the pair is shifted one
// element down so that it stays inside the row.
struct __attribute__((packed, aligned(4))) Pair { float a, b; };

__device__ __forceinline__ void ld_pair(const float* __restrict__ base, unsigned off, unsigned dx,
                                        float& v0, float& v1) {
  const unsigned o = dx ? off : off - 4u;
  const Pair q = *reinterpret_cast<const Pair*>(reinterpret_cast<const char*>(base) + o);
  v0 = dx ? q.a : q.b;
  v1 = q.b;
}

__device__ __forceinline__ Corners w3_gather(const float* __restrict__ vol, const Samp3& s) {
  Corners c;
  const unsigned o010 = s.o000 + s.dy, o100 = s.o000 + s.dz, o110 = o100 + s.dy;
  ld_pair(vol, s.o000, s.dx, c.v000, c.v001);
  ld_pair(vol, o010, s.dx, c.v010, c.v011);
  ld_pair(vol, o100, s.dx, c.v100, c.v101);
  ld_pair(vol, o110, s.dx, c.v110, c.v111);
  return c;
}

__device__ __forceinline__ float lerp(float a, float b, float t) { return fmaf(t, b - a, a); }

constexpr int TW = 32;       // tile extent along w
constexpr int LDH = TH + 1;  // LDS tiles are kept TRANSPOSED, [w][h], +1 pad: the w-major phases
                             // (lanes on w) and the h-major phase (lanes on h) are both conflict-free

// blockIdx.x -> (b, d-chunk, h-tile, w-tile), w-tile fastest
__device__ __forceinline__ void decode_tile(const W3P& p, int& b, int& d0, int& h0, int& w0) {
  int bid = blockIdx.x;
  const int tw = bid % p.tilesW; bid /= p.tilesW;
  const int th = bid % p.tilesH; bid /= p.tilesH;
  const int dk = bid % p.nDC;
  b = bid / p.nDC;
  d0 = dk * p.dc;
  h0 = th * TH;
  w0 = tw * TW;
}

// ---- w-major tile movers.  VEC: 8 lanes x float4 per 128-B tile row (needs W % 4 == 0);
//      otherwise 32 lanes x float.  `NR` = tile rows (h) x planes moved, rows are h-clamped.
template <int NT, bool VEC>
struct Mover {
  static constexpr int LPR = VEC ? 8 : 32;    // lanes per tile row
  static constexpr int RP = NT / LPR;         // rows per pass
  static constexpr int PASSES = TH / RP;      // passes per 64-row plane
  using elem = typename std::conditional<VEC, float4, float>::type;

  // global plane (one d-slice of one channel, row pitch W) -> registers
  __device__ static __forceinline__ void load(const float* __restrict__ plane, const W3P& p, int h0,
                                              int w0, elem (&r)[PASSES]) {
    const int t = threadIdx.x, col = t % LPR, row0 = t / LPR;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int h = min(h0 + row0 + it * RP, p.H - 1);
      if (VEC) {
        const int w = min(w0 + 4 * col, p.W - 4);
        *reinterpret_cast<float4*>(&r[it]) = *reinterpret_cast<const float4*>(plane + (size_t)h * p.W + w);
      } else {
        const int w = min(w0 + col, p.W - 1);
        *reinterpret_cast<float*>(&r[it]) = plane[(size_t)h * p.W + w];
      }
    }
  }
  // ---- fused up-sampling producer ----------------------------------------------------------------------
  // The flow plane of slice d is prev + scale * trilinear_up(delta) with the arithmetic of
  // fs_upsample3d_scale_add (bit-identical).  The low-resolution source brick of the workgroup's whole
  // 4-slice tile (<= 4 x 34 x 18 voxels per channel at factor 2, 2 x 18 x 10 at factor 4) is staged in LDS
  // ONCE (`up_stage`); per slice, `up_prefetch` issues the running flow's float4 loads where the plain warp
  // issues its flow loads (they land under the store phase and the barrier) and `up_finish` blends the 8
  // corners from LDS, adds, writes the accumulated flow once and leaves the tile in registers in the layout
  // `load` produces.
  // (PZ: source slices under dc = 4 x factor output slices -- 8 at factor 2, 16 at factor 4 -- plus the two halo slices)
  static constexpr int PZ = 6, PY = TH / 2 + 2, PX = TW / 2 + 2, PXP = PX + 1;
  typedef float Brick[PZ][PY][PXP];

  __device__ static __forceinline__ int up_i0(int o, int n_in, float rs) {
    int i0, ip;
    float l0, l1;
    fs::trilinear_axis(o, n_in, rs, i0, ip, l0, l1);
    return i0;
  }

  // stage channels c0..c0+2 of the delta (volumes sv + k * svol) around the tile (d0.., h0.., w0..)
  __device__ static __forceinline__ void up_stage(Brick* sS, const float* __restrict__ sv, size_t svol,
                                                  const W3P& p, const UpP& u, int d0, int h0, int w0) {
    const int bz = up_i0(d0, u.Ds, u.rs), by = up_i0(min(h0, p.H - 1), u.Hs, u.rs),
              bx = up_i0(min(w0, p.W - 1), u.Ws, u.rs);
    for (int i = threadIdx.x; i < 3 * PZ * PY * PX; i += NT) {
      const int c = i / (PZ * PY * PX), r1 = i - c * (PZ * PY * PX);
      const int z = r1 / (PY * PX), r2 = r1 - z * (PY * PX);
      const int y = r2 / PX, x = r2 - y * PX;
      const int gz = min(bz + z, u.Ds - 1), gy = min(by + y, u.Hs - 1), gx = min(bx + x, u.Ws - 1);
      sS[c][z][y][x] = sv[(size_t)c * svol + ((size_t)gz * u.Hs + gy) * u.Ws + gx];
    }
  }

  __device__ static __forceinline__ void up_prefetch(const float* __restrict__ pplane, const W3P& p, int h0,
                                                     int w0, elem (&q)[PASSES]) {
    const int t = threadIdx.x, col = t % LPR, row0 = t / LPR;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int h = min(h0 + row0 + it * RP, p.H - 1);
      if (VEC) {
        const int w = min(w0 + 4 * col, p.W - 4);
        *reinterpret_cast<float4*>(&q[it]) = *reinterpret_cast<const float4*>(pplane + (size_t)h * p.W + w);
      } else {
        const int w = min(w0 + col, p.W - 1);
        *reinterpret_cast<float*>(&q[it]) = pplane[(size_t)h * p.W + w];
      }
    }
  }

  __device__ static __forceinline__ void up_finish(const Brick& sb, bool has_prev, float* __restrict__ oplane,
                                                   const W3P& p, const UpP& u, int d0, int d, int h0, int w0,
                                                   const elem (&q)[PASSES], elem (&r)[PASSES]) {
#pragma clang fp contract(off)
    const int t = threadIdx.x, col = t % LPR, row0 = t / LPR;
    const int bz = up_i0(d0, u.Ds, u.rs), by = up_i0(min(h0, p.H - 1), u.Hs, u.rs),
              bx = up_i0(min(w0, p.W - 1), u.Ws, u.rs);
    int z0, zp;
    float lz0, lz1;
    fs::trilinear_axis(d, u.Ds, u.rs, z0, zp, lz0, lz1);
    const int za = z0 - bz, zb = za + zp;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int hq = h0 + row0 + it * RP;
      const int h = min(hq, p.H - 1);
      int y0, yp;
      float ly0, ly1;
      fs::trilinear_axis(h, u.Hs, u.rs, y0, yp, ly0, ly1);
      const int ya = y0 - by, yb = ya + yp;
      const float* s00 = &sb[za][ya][0];
      const float* s01 = &sb[za][yb][0];
      const float* s10 = &sb[zb][ya][0];
      const float* s11 = &sb[zb][yb][0];
      const int wq = VEC ? w0 + 4 * col : w0 + col;
      const int w = VEC ? min(wq, p.W - 4) : min(wq, p.W - 1);
      float o[4];
#pragma unroll
      for (int i = 0; i < (VEC ? 4 : 1); ++i) {
        int x0, xp;
        float lx0, lx1;
        fs::trilinear_axis(w + i, u.Ws, u.rs, x0, xp, lx0, lx1);
        const int a = x0 - bx, b = a + xp;  // brick column j holds source column min(bx + j, Ws - 1)
        // upsample_trilinear3d_out_frame's expression (trilinear_up_row in common.hpp)
        const float v = lz0 * (ly0 * (lx0 * s00[a] + lx1 * s00[b]) + ly1 * (lx0 * s01[a] + lx1 * s01[b])) +
                        lz1 * (ly0 * (lx0 * s10[a] + lx1 * s10[b]) + ly1 * (lx0 * s11[a] + lx1 * s11[b]));
        o[i] = v * u.scale;
      }
      if (VEC) {
        float4 v = make_float4(o[0], o[1], o[2], o[3]);
        if (has_prev) {
          const float4 pq = *reinterpret_cast<const float4*>(&q[it]);
          v.x = pq.x + v.x; v.y = pq.y + v.y; v.z = pq.z + v.z; v.w = pq.w + v.w;
        }
        if (hq < p.H && wq < p.W) *reinterpret_cast<float4*>(oplane + (size_t)h * p.W + w) = v;
        *reinterpret_cast<float4*>(&r[it]) = v;
      } else {
        float v = o[0];
        if (has_prev) v = *reinterpret_cast<const float*>(&q[it]) + v;
        if (hq < p.H && wq < p.W) oplane[(size_t)h * p.W + w] = v;
        *reinterpret_cast<float*>(&r[it]) = v;
      }
    }
  }
  // registers -> transposed LDS tile
  __device__ static __forceinline__ void to_lds(float (*tile)[LDH], const elem (&r)[PASSES]) {
    const int t = threadIdx.x, col = t % LPR, row0 = t / LPR;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int hh = row0 + it * RP;
      if (VEC) {
        const float4 v = *reinterpret_cast<const float4*>(&r[it]);
        tile[4 * col + 0][hh] = v.x; tile[4 * col + 1][hh] = v.y;
        tile[4 * col + 2][hh] = v.z; tile[4 * col + 3][hh] = v.w;
      } else {
        tile[col][hh] = *reinterpret_cast<const float*>(&r[it]);
      }
    }
  }
  // transposed LDS tile (+ the same positions of `add`, when given) -> global plane (guarded).  `add`
  // may be the plane itself: every element is read and then written by the same thread.
  __device__ static __forceinline__ void store(float* plane, const W3P& p, int h0, int w0,
                                               const float (*tile)[LDH], const float* add = nullptr,
                                               const float* add1 = nullptr, const float* add2 = nullptr) {
    const int t = threadIdx.x, col = t % LPR, row0 = t / LPR;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
      const int hh = row0 + it * RP;
      const int h = h0 + hh;
      if (VEC) {
        const int w = w0 + 4 * col;
        if (h < p.H && w < p.W) {
          float4 v;
          v.x = tile[4 * col + 0][hh]; v.y = tile[4 * col + 1][hh];
          v.z = tile[4 * col + 2][hh]; v.w = tile[4 * col + 3][hh];
          if (add != nullptr) {
            const float4 a = *reinterpret_cast<const float4*>(add + (size_t)h * p.W + w);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          if (add1 != nullptr) {
            const float4 a = *reinterpret_cast<const float4*>(add1 + (size_t)h * p.W + w);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          if (add2 != nullptr) {
            const float4 a = *reinterpret_cast<const float4*>(add2 + (size_t)h * p.W + w);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          *reinterpret_cast<float4*>(plane + (size_t)h * p.W + w) = v;
        }
      } else {
        const int w = w0 + col;
        if (h < p.H && w < p.W) {
          float v = tile[col][hh] + (add != nullptr ? add[(size_t)h * p.W + w] : 0.f);
          if (add1 != nullptr) v += add1[(size_t)h * p.W + w];
          if (add2 != nullptr) v += add2[(size_t)h * p.W + w];
          plane[(size_t)h * p.W + w] = v;
        }
      }
    }
  }
};

// (UPS: two 512-thread workgroups per CU = 4 waves per SIMD, i.e. <= 128 VGPRs, like the plain kernel)
template <int NT, bool VEC, bool UPS>
__global__ __launch_bounds__(NT, UPS ? 4 : 1) void warp3d_fwd_kernel(W3Fwd io, const float* __restrict__ flow, UpP u,
                                                        W3P p) {
  using M = Mover<NT, VEC>;
  constexpr int NW = TW / (NT / 64);  // voxels per thread and slice in the h-major phase
  const float* __restrict__ in = io.in[blockIdx.y];
  float* __restrict__ out = io.out[blockIdx.y];
  __shared__ float sF[3][TW][LDH];
  __shared__ float sO[TW][LDH];

  int b, d0, h0, w0;
  decode_tile(p, b, d0, h0, w0);
  const int HW = p.H * p.W;
  const size_t vol = (size_t)p.D * HW;
  const size_t ivol = (size_t)p.Di * p.Hi * p.Wi;
  const size_t fch = (size_t)b * p.flowC + 3 * blockIdx.y;  // first flow channel of this pair member
  const float* fb = UPS ? nullptr : flow + fch * vol;
  const int dEnd = min(d0 + p.dc, p.D);

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = min(h0 + lane, p.H - 1);
  const float lin_h = fs::linspace_pm1(h, p.H, p.stepH);

  typename M::elem r0[M::PASSES], r1[M::PASSES], r2[M::PASSES];
  typename M::elem q0[M::PASSES], q1[M::PASSES], q2[M::PASSES];  // UPS: the running flow's tile (dead otherwise)
  __shared__ typename M::Brick sS[UPS ? 3 : 1];
  if constexpr (UPS) {
    const size_t svol = (size_t)u.Ds * u.Hs * u.Ws;
    M::up_stage(sS, u.small + fch * svol, svol, p, u, d0, h0, w0);
    __syncthreads();
  }
  // the three flow planes of slice d: loads issued here -- of the flow itself (plain warp) or of the running
  // flow (UPS; blended with the up-sampled delta by up_tiles at the top of the slice's iteration)
  auto flow_tiles = [&](int d) {
    if constexpr (UPS) {
      if (u.prev != nullptr) {
        const float* pp = u.prev + fch * vol + (size_t)d * HW;
        M::up_prefetch(pp, p, h0, w0, q0); M::up_prefetch(pp + vol, p, h0, w0, q1);
        M::up_prefetch(pp + 2 * vol, p, h0, w0, q2);
      }
    } else {
      const float* f = fb + (size_t)d * HW;
      M::load(f, p, h0, w0, r0); M::load(f + vol, p, h0, w0, r1); M::load(f + 2 * vol, p, h0, w0, r2);
    }
  };
  auto up_tiles = [&](int d) {
    float* op = u.fout + fch * vol + (size_t)d * HW;
    const bool hp = u.prev != nullptr;
    M::up_finish(sS[0], hp, op, p, u, d0, d, h0, w0, q0, r0);
    M::up_finish(sS[UPS ? 1 : 0], hp, op + vol, p, u, d0, d, h0, w0, q1, r1);
    M::up_finish(sS[UPS ? 2 : 0], hp, op + 2 * vol, p, u, d0, d, h0, w0, q2, r2);
  };
  flow_tiles(d0);
  for (int d = d0; d < dEnd; ++d) {
    if constexpr (UPS) up_tiles(d);  // blend the loaded corners, write the accumulated flow, fill r0..r2
    // phase 1: flow tile of this slice (already in registers) -> LDS
    M::to_lds(sF[0], r0); M::to_lds(sF[1], r1); M::to_lds(sF[2], r2);
    __syncthreads();
    const float lin_d = fs::linspace_pm1(d, p.D, p.stepD);
    for (int c = 0; c < p.C; ++c) {
      const float* __restrict__ vin = in + ((size_t)b * p.C + c) * ivol;
      // phase 2: lanes on h; wave wv takes NW consecutive w of the tile
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int lw = wv * NW + k;
        const int w = min(w0 + lw, p.W - 1);
        const Samp3 s = w3_sample<false>(p, lin_h, lin_d, fs::linspace_pm1(w, p.W, p.stepW),
                                         sF[0][lw][lane], sF[1][lw][lane], sF[2][lw][lane]);
        const Corners q = w3_gather(vin, s);
        const float c00 = lerp(q.v000, q.v001, s.ax), c01 = lerp(q.v010, q.v011, s.ax);
        const float c10 = lerp(q.v100, q.v101, s.ax), c11 = lerp(q.v110, q.v111, s.ax);
        sO[lw][lane] = lerp(lerp(c00, c01, s.ay), lerp(c10, c11, s.ay), s.az);
      }
      __syncthreads();
      if (c + 1 == p.C && d + 1 < dEnd) {
        // next slice's flow tile: in flight during the store phase and the next LDS hand-over
        flow_tiles(d + 1);
      }
      // phase 3: lanes on w, coalesced store
      M::store(out + ((size_t)b * p.C + c) * vol + (size_t)d * HW, p, h0, w0, sO);
      if (c + 1 < p.C) __syncthreads();
    }
    // the next iteration's first barrier separates this phase 3 from the next phase 2 (sO),
    // and every wave has left phase 2 before sF is overwritten
  }
}

// ---- round 4: the plain forward warp as a software pipeline (warp3d_fwd_ring_kernel) ------------------------------
// One workgroup per CU: 8 compute waves + 2 mover waves, a ring of NF flow-tile stages in LDS.
//   * movers: bring the flow tiles of slices k+1 .. k+NF-1 global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR
//     stop, no ds_write pass, non-temporal: the flow is read once) and stream the output tile of slice k-1 from LDS to
//     global (16-byte stores, 128-byte rows); counted vmcnt waits: only slice k+2 must have landed at barrier k.
//   * compute waves: lane = h, wave = one float4 column (4 consecutive w).  The flow of a thread's 4 voxels is ONE
//     ds_read_b128 per plane; the gathers of slice k+1 are issued BEFORE slice k is blended (two register sets that
//     hold the RAW pairs: nothing touches a loaded value in the half that issues it), so they fly across the barrier
//     and a whole iteration; the output tile goes to LDS with one ds_write_b128.
// ONE barrier per slice.  LDS tile layout: a plane tile is 64 h x 8 float4 (4 consecutive w each); float4 slot (h, q)
// sits at index h * 8 + (q ^ swz(h)), swz(h) = (h ^ h >> 3) & 7 -- conflict-free for all four access patterns (b128
// column reads with lane = h, b128 row reads / writes with 8 lanes per row, the movers' linear slot order), found by
// enumeration.  The DMA writes LDS linearly (slot = 64 j + lane), so the swizzle is applied on the GLOBAL side: lane ->
// (row, logical column) of the tile; 8 lanes still cover one 128-byte row.  Values are bit-identical to the round-3
// kernel (same arithmetic on the same operands; scripts/w3bench.py prints CRCs).
// What it bought, and what bounds the family (profiles/r04_w3_pmc.txt, profiles/r04_w3_ablation.txt): 0.362 -> 0.348 ms
// per pair launch in the micro-benchmark -- NOT the 0.26 the latency picture predicted.  The round-3 reading ("no unit
// saturated, so barrier-separated phases") was incomplete: the CU's vector L1 sustains ~64 line requests in flight
// (x 128 B / ~700-900 cycles of L2 / HBM latency under load = ~10 B/clk/CU, which summed over 256 CUs IS the HBM rate),
// the backward kernel already averages 56 in flight (TCP_TCC_READ_REQ_LATENCY / cycles) and sits at that bound; the
// forward kernel averaged 38 because every gather wave-instruction costs the texture-address path ~26 cycles (64 lanes
// x 8 unaligned bytes) and TA work -- 128 gather + 24 tile + 8 store instructions per tile-slice, ~3 900 cycles -- adds to
// rather than hides behind the streaming: ablations of the v1 pipeline: flow DMA alone 0.154 ms (5.2 TB/s), + output
// stores 0.227, + compute without gathers 0.263, + gathers from an L2-resident 64 KB source 0.366, real gathers 0.404;
// insensitive to workgroups per CU (1 vs 2), d-chunk, and cache policy.  Fewer TA instructions would need the gather
// SOURCE staged in LDS (a flow-dependent window: DESIGN.md §5), not a different schedule.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int TQ = TW / 4;        // float4 slots per tile row
constexpr int TSLOTS = TH * TQ;   // 512 float4 = 8 KB per plane tile
constexpr int NCW = 8;            // compute waves of a workgroup
constexpr int NMW = 2;            // mover waves
constexpr int NT2 = 64 * (NCW + NMW);

__device__ __forceinline__ int t_swz(int h) { return (h ^ (h >> 3)) & 7; }
__device__ __forceinline__ int t_slot(int h, int q) { return h * TQ + (q ^ t_swz(h)); }

// A mover lane's share of a plane tile: slots 64 j + lane of instruction j (4 per mover wave).  VEC: one 16-byte
// piece per slot; otherwise (W % 4 != 0 or unaligned tensors) four dword pieces per slot -- the same LDS image.
template <bool VEC>
struct MoverLane {
  static constexpr int NJ = TSLOTS / 64 / NMW;        // slot groups (64 slots each) per mover wave: 4
  static constexpr int NV = VEC ? NJ : 4 * NJ;        // DMA instructions per plane and mover wave
  unsigned voff[NV];                                  // byte offset inside a [H][W] plane (slice-invariant)
  int hq[NJ], wq[NJ];                                 // unclamped (h, w) of the lane's slot in group j (stores)
  int m;

  __device__ __forceinline__ void init(const W3P& p, int m_, int lane, int h0, int w0) {
    m = m_;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int s = 64 * (NJ * m + i) + lane, hr = s / TQ, q = (s % TQ) ^ t_swz(hr);
      hq[i] = h0 + hr; wq[i] = w0 + 4 * q;
      if (VEC) voff[i] = ((unsigned)min(hq[i], p.H - 1) * (unsigned)p.W + (unsigned)min(wq[i], p.W - 4)) * 4u;
    }
    if (!VEC) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {  // dword pieces: instruction i covers slots 16 i .. 16 i + 15 of this wave's share
        const int s = 64 * NJ * m + 16 * i + (lane >> 2), hr = s / TQ, q = (s % TQ) ^ t_swz(hr);
        voff[i] = ((unsigned)min(h0 + hr, p.H - 1) * (unsigned)p.W + (unsigned)min(w0 + 4 * q + (lane & 3), p.W - 1)) * 4u;
      }
    }
  }
  // global plane (one d-slice of one channel) -> LDS plane tile; completion: the issuing wave's vmcnt
  template <int AUX>
  __device__ __forceinline__ void dma(const float* plane, float4* tile) const {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)plane, (short)0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (VEC) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(tile + 64 * (NJ * m + i)), 16, voff[i], 0, 0, AUX);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(tile + 64 * NJ * m + 16 * i), 4, voff[i], 0, 0, AUX);
    }
#endif
  }
  // LDS plane tile -> registers (the lane's NJ slots)
  __device__ __forceinline__ void get(const float4* tile, float4 (&v)[NJ]) const {
#pragma unroll
    for (int i = 0; i < NJ; ++i) v[i] = tile[64 * (NJ * m + i) + (threadIdx.x & 63)];
  }
  // registers -> global plane (guarded)
  __device__ __forceinline__ void put(float* plane, const W3P& p, const float4 (&v)[NJ]) const {
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (hq[i] >= p.H) continue;
      float* dst = plane + (size_t)hq[i] * p.W + wq[i];
      if (VEC) {
        if (wq[i] < p.W) *reinterpret_cast<float4*>(dst) = v[i];
      } else {
        if (wq[i] + 0 < p.W) dst[0] = v[i].x;
        if (wq[i] + 1 < p.W) dst[1] = v[i].y;
        if (wq[i] + 2 < p.W) dst[2] = v[i].z;
        if (wq[i] + 3 < p.W) dst[3] = v[i].w;
      }
    }
  }
};

// ---- v2: a ring of NF flow stages + software-pipelined compute waves -------------------------------------------
// PMC of v1 (profiles/r04_w3_pmc.txt): the CU's vector L1 holds ~38 line requests in flight on average at ~700
// cycles each -- the per-CU fill rate (outstanding lines x 128 B / latency) is what bounds the kernel, and it is bursty:
// all compute waves compute addresses together, then gather together.  v2 keeps the queue fed: the movers run NF - 1
// slices ahead (counted vmcnt waits), and a compute wave issues the gathers of slice k + 1 BEFORE it blends slice k
// (two register sets), so its gathers fly across the barrier and a whole iteration.
// the workgroup barrier of the ring kernel: LDS operations of this wave done, barrier, and -- the "memory" clobber -- no LDS
// or global access of the compiler's moved across it (the bare s_barrier builtin is not a memory barrier to the compiler)
__device__ __forceinline__ void w3_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct GState {  // gathers in flight for one slice: 4 voxels of a thread.  The RAW pairs: nothing may touch a loaded
  Pair r[4][4];  // value before `back` (a select on it would make the wave wait for the load where it was issued)
  unsigned dx[4];
  float ax[4], ay[4], az[4];
};

__device__ __forceinline__ Pair ld_pair_raw(const float* __restrict__ base, unsigned off, unsigned dx) {
  return *reinterpret_cast<const Pair*>(reinterpret_cast<const char*>(base) + (dx ? off : off - 4u));
}

// DBG (ablation build only; wrong results): 1 = every pair gather at an 8-byte ALIGNED address, 2 = dword gathers
template <bool VEC, int AUX, int NF, int DBG = 0>
__global__ __launch_bounds__(NT2, 2) void warp3d_fwd_ring_kernel(W3Fwd io, const float* __restrict__ flow, W3P p) {
  static_assert(NF >= 3 && NF <= 5, "ring depth");
  const float* __restrict__ in = io.in[blockIdx.y];
  float* __restrict__ out = io.out[blockIdx.y];
  __shared__ float4 sF[NF][3][TSLOTS];
  __shared__ float4 sO[2][TSLOTS];

  int b, d0, h0, w0;
  decode_tile(p, b, d0, h0, w0);
  const int HW = p.H * p.W;
  const size_t vol = (size_t)p.D * HW;
  const size_t ivol = (size_t)p.Di * p.Hi * p.Wi;
  const float* fb = flow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol;
  const int n = min(d0 + p.dc, p.D) - d0;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // (C > 1: the image channels of a slice are stages of the same pipeline; the flow ring advances per slice)
  const int C = p.C, nst = n * C;

  if (wv >= NCW) {
    // ---------------- movers ----------------
    MoverLane<VEC> mv;
    mv.init(p, wv - NCW, lane, h0, w0);
    constexpr int PER = 3 * MoverLane<VEC>::NV;  // DMA instructions per slice and mover wave
    auto load_slice = [&](int k) {
      const float* f = fb + (size_t)(d0 + k) * HW;
      mv.template dma<AUX>(f, sF[k % NF][0]);
      mv.template dma<AUX>(f + vol, sF[k % NF][1]);
      mv.template dma<AUX>(f + 2 * vol, sF[k % NF][2]);
    };
    auto store_stage = [&](int s) {
      const int k = s / C, c = s - k * C;
      float4 v[MoverLane<VEC>::NJ];
      mv.get(sO[s & 1], v);
      mv.put(out + ((size_t)b * C + c) * vol + (size_t)(d0 + k) * HW, p, v);
    };
    // wait until at most `younger` slices' DMA instructions (issued after the one that must have landed) are pending;
    // stores in between only make the wait stricter
    auto wait_younger = [&](int younger) {
      if (VEC) {
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER <= 63 ? 3 * PER : 63) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER <= 63 ? 2 * PER : 63) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER <= 63 ? PER : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the dword-piece form issues 48 instructions per slice)
      }
    };
    // prologue: slices 0 .. NF-2 requested, 0 and 1 landed
    const int pre = min(n, NF - 1);
    for (int k = 0; k < pre; ++k) load_slice(k);
    wait_younger(max(pre - 2, 0));
    w3_lds_barrier();
    for (int s = 0; s < nst; ++s) {
      const int k = s / C, c = s - k * C;
      if (s > 0) store_stage(s - 1);
      int issued_to = k + NF - 2;                       // last slice requested before this iteration
      if (c == 0 && k + NF - 1 < n) {     // slice k - 1's buffer: its last reader ran an iteration ago
        load_slice(k + NF - 1);
        issued_to = k + NF - 1;
      }
      // before anyone is released, slice k + 2 (read by the compute waves in the next iteration) has landed
      wait_younger(max(min(issued_to, n - 1) - (k + 2), 0));
      w3_lds_barrier();
    }
    store_stage(nst - 1);
    return;
  }
  // ---------------- compute waves ----------------
  const int h = min(h0 + lane, p.H - 1);
  const float lin_h = fs::linspace_pm1(h, p.H, p.stepH);
  float lin_w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lin_w[i] = fs::linspace_pm1(min(w0 + 4 * wv + i, p.W - 1), p.W, p.stepW);
  const int slot = t_slot(lane, wv);
  auto front = [&](int s, GState& g) {  // stage s: flow from LDS, sample positions, gathers issued
    const int k = s / C, c = s - k * C;
    const float lin_d = fs::linspace_pm1(d0 + k, p.D, p.stepD);
    const float4 f0 = sF[k % NF][0][slot], f1 = sF[k % NF][1][slot], f2 = sF[k % NF][2][slot];
    const float fa[4] = {f0.x, f0.y, f0.z, f0.w}, fbv[4] = {f1.x, f1.y, f1.z, f1.w}, fc[4] = {f2.x, f2.y, f2.z, f2.w};
    const float* __restrict__ vin = in + ((size_t)b * C + c) * ivol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Samp3 sm = w3_sample<false>(p, lin_h, lin_d, lin_w[i], fa[i], fbv[i], fc[i]);
      unsigned o000 = sm.o000;
      if (DBG == 1) o000 &= ~7u;
      const unsigned o010 = o000 + sm.dy, o100 = o000 + sm.dz, o110 = o100 + sm.dy;
      if (DBG == 2) {
        g.r[i][0].a = ldb(vin, o000); g.r[i][1].a = ldb(vin, o010); g.r[i][2].a = ldb(vin, o100); g.r[i][3].a = ldb(vin, o110);
        g.r[i][0].b = g.r[i][1].b = g.r[i][2].b = g.r[i][3].b = sm.ax;
        g.dx[i] = sm.dx; g.ax[i] = sm.ax; g.ay[i] = sm.ay; g.az[i] = sm.az;
        continue;
      }
      if (DBG == 3 || DBG == 4) {  // half / a quarter of the gather instructions
        g.r[i][0] = ld_pair_raw(vin, o000, sm.dx);
        g.r[i][1] = (DBG == 3) ? ld_pair_raw(vin, o110, sm.dx) : g.r[i][0];
        g.r[i][2] = g.r[i][0]; g.r[i][3] = g.r[i][1];
        g.dx[i] = sm.dx; g.ax[i] = sm.ax; g.ay[i] = sm.ay; g.az[i] = sm.az;
        continue;
      }
      if (DBG == 5) {  // no gathers at all
        g.r[i][0].a = sm.ax; g.r[i][0].b = __uint_as_float(o000 + o110); g.r[i][1] = g.r[i][2] = g.r[i][3] = g.r[i][0];
        g.dx[i] = sm.dx; g.ax[i] = sm.ax; g.ay[i] = sm.ay; g.az[i] = sm.az;
        continue;
      }
      g.r[i][0] = ld_pair_raw(vin, o000, sm.dx); g.r[i][1] = ld_pair_raw(vin, o010, sm.dx);
      g.r[i][2] = ld_pair_raw(vin, o100, sm.dx); g.r[i][3] = ld_pair_raw(vin, o110, sm.dx);
      g.dx[i] = sm.dx; g.ax[i] = sm.ax; g.ay[i] = sm.ay; g.az[i] = sm.az;
    }
  };
  auto back = [&](int s, const GState& g) {  // stage s: blend, output tile to LDS
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = g.dx[i] != 0u;  // (far border: the pair was fetched one element down, see ld_pair)
      const float c00 = lerp(in ? g.r[i][0].a : g.r[i][0].b, g.r[i][0].b, g.ax[i]);
      const float c01 = lerp(in ? g.r[i][1].a : g.r[i][1].b, g.r[i][1].b, g.ax[i]);
      const float c10 = lerp(in ? g.r[i][2].a : g.r[i][2].b, g.r[i][2].b, g.ax[i]);
      const float c11 = lerp(in ? g.r[i][3].a : g.r[i][3].b, g.r[i][3].b, g.ax[i]);
      o[i] = lerp(lerp(c00, c01, g.ay[i]), lerp(c10, c11, g.ay[i]), g.az[i]);
    }
    sO[s & 1][slot] = make_float4(o[0], o[1], o[2], o[3]);
  };
  GState ga, gb;
  w3_lds_barrier();  // slices 0 and 1 are in LDS
  front(0, ga);
  int s = 0;
  // steady state: both `front`s unconditional -- a conditional issue would make the compiler's vmcnt bookkeeping
  // assume the shorter path and wait for the gathers it has just issued
  for (; s + 2 < nst; s += 2) {
    front(s + 1, gb);
    back(s, ga);
    w3_lds_barrier();
    front(s + 2, ga);
    back(s + 1, gb);
    w3_lds_barrier();
  }
  if (s + 1 < nst) {
    front(s + 1, gb);
    back(s, ga);
    w3_lds_barrier();
    back(s + 1, gb);
    w3_lds_barrier();
  } else {
    back(s, ga);
    w3_lds_barrier();
  }
}

// up to three gradients reaching the flow from its other consumers: [B, >= flowC, D,H,W] tensors or channel
// slices of wider ones (`bs` = batch stride in floats; the channel stride is always D*H*W)
struct W3Add {
  const float* a[3];
  long long bs[3];
};

template <int NT, bool VEC, bool WITH_GIN>
__global__ __launch_bounds__(NT) void warp3d_bwd_kernel(W3Bwd io, const float* __restrict__ flow,
                                                        float* gflow, W3Add gadd, W3P p) {
  using M = Mover<NT, VEC>;
  constexpr int NW = TW / (NT / 64);
  const float* __restrict__ in = io.in[blockIdx.y];
  const float* __restrict__ gout = io.gout[blockIdx.y];
  float* __restrict__ gin = io.gin[blockIdx.y];
  __shared__ float sF[3][TW][LDH];
  __shared__ float sG[TW][LDH];

  int b, d0, h0, w0;
  decode_tile(p, b, d0, h0, w0);
  const int HW = p.H * p.W;
  const size_t vol = (size_t)p.D * HW;
  const size_t ivol = (size_t)p.Di * p.Hi * p.Wi;
  const float* fb = flow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol;
  float* gfb = gflow ? gflow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol : nullptr;
  const float* gob = gout + (size_t)b * (io.gbs[blockIdx.y] ? (size_t)io.gbs[blockIdx.y] : (size_t)p.C * vol);
  // gradients reaching the flow from its other consumers, summed into the stored tile (the first may alias gflow)
  const float* gab[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    gab[i] = (gflow && gadd.a[i]) ? gadd.a[i] + (size_t)b * gadd.bs[i] + (size_t)(3 * blockIdx.y) * vol : nullptr;
  const int dEnd = min(d0 + p.dc, p.D);

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hq = h0 + lane;
  const int h = min(hq, p.H - 1);
  const float lin_h = fs::linspace_pm1(h, p.H, p.stepH);
  // chain rule through unnormalize ((size-1)/2) and through flow * 1/((dim-1)/2):
  //   F0 -> gx (H-normalised) -> ix (W),  F1 -> gy (D) -> iy (H),  F2 -> gz (W) -> iz (D)
  const float k0 = (p.mW * 0.5f) * p.rH, k1 = (p.mH * 0.5f) * p.rD, k2 = (p.mD * 0.5f) * p.rW;

  typename M::elem r0[M::PASSES], r1[M::PASSES], r2[M::PASSES], rg[M::PASSES];
  {
    const float* f = fb + (size_t)d0 * HW;
    M::load(f, p, h0, w0, r0); M::load(f + vol, p, h0, w0, r1); M::load(f + 2 * vol, p, h0, w0, r2);
    M::load(gob + (size_t)d0 * HW, p, h0, w0, rg);
  }
  for (int d = d0; d < dEnd; ++d) {
    M::to_lds(sF[0], r0); M::to_lds(sF[1], r1); M::to_lds(sF[2], r2);
    const float lin_d = fs::linspace_pm1(d, p.D, p.stepD);
    float acc[NW][3];
#pragma unroll
    for (int k = 0; k < NW; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;

    for (int c = 0; c < p.C; ++c) {
      if (c > 0) {
        __syncthreads();  // previous channel's sG readers are done
        M::load(gob + (size_t)c * vol + (size_t)d * HW, p, h0, w0, rg);
      }
      M::to_lds(sG, rg);
      __syncthreads();
      const float* __restrict__ vin = in + ((size_t)b * p.C + c) * ivol;
      float* __restrict__ gvin = WITH_GIN ? gin + ((size_t)b * p.C + c) * ivol : nullptr;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int lw = wv * NW + k;
        const int w = min(w0 + lw, p.W - 1);
        const bool live = (hq < p.H) && (w0 + lw < p.W);
        const Samp3 s = w3_sample<true>(p, lin_h, lin_d, fs::linspace_pm1(w, p.W, p.stepW),
                                        sF[0][lw][lane], sF[1][lw][lane], sF[2][lw][lane]);
        const Corners q = w3_gather(vin, s);
        const float g = live ? sG[lw][lane] : 0.f;  // clamped duplicates add nothing anywhere
        // d out / d(ix, iy, iz) through the nested lerps (== ATen grid_sampler_3d_backward, factored)
        const float c00 = lerp(q.v000, q.v001, s.ax), c01 = lerp(q.v010, q.v011, s.ax);
        const float c10 = lerp(q.v100, q.v101, s.ax), c11 = lerp(q.v110, q.v111, s.ax);
        const float dx = lerp(lerp(q.v001 - q.v000, q.v011 - q.v010, s.ay),
                              lerp(q.v101 - q.v100, q.v111 - q.v110, s.ay), s.az);
        const float dy = lerp(c01 - c00, c11 - c10, s.az);
        const float dz = lerp(c10, c11, s.ay) - lerp(c00, c01, s.ay);
        acc[k][0] = fmaf(g * s.mx, dx, acc[k][0]);
        acc[k][1] = fmaf(g * s.my, dy, acc[k][1]);
        acc[k][2] = fmaf(g * s.mz, dz, acc[k][2]);
        if (WITH_GIN) {
          if (live) {
            // scatter; +1 corners outside the volume have weight exactly 0 (offset delta 0): skipped
            const float bx = 1.0f - s.ax, by = 1.0f - s.ay, bz = 1.0f - s.az;
            char* gb8 = reinterpret_cast<char*>(gvin);
            const unsigned o010 = s.o000 + s.dy, o100 = s.o000 + s.dz, o110 = o100 + s.dy;
            atomicAdd(reinterpret_cast<float*>(gb8 + s.o000), g * (bx * by * bz));
            if (s.dx) atomicAdd(reinterpret_cast<float*>(gb8 + s.o000 + s.dx), g * (s.ax * by * bz));
            if (s.dy) atomicAdd(reinterpret_cast<float*>(gb8 + o010), g * (bx * s.ay * bz));
            if (s.dy && s.dx)
              atomicAdd(reinterpret_cast<float*>(gb8 + o010 + s.dx), g * (s.ax * s.ay * bz));
            if (s.dz) {
              atomicAdd(reinterpret_cast<float*>(gb8 + o100), g * (bx * by * s.az));
              if (s.dx) atomicAdd(reinterpret_cast<float*>(gb8 + o100 + s.dx), g * (s.ax * by * s.az));
              if (s.dy) atomicAdd(reinterpret_cast<float*>(gb8 + o110), g * (bx * s.ay * s.az));
              if (s.dy && s.dx)
                atomicAdd(reinterpret_cast<float*>(gb8 + o110 + s.dx), g * (s.ax * s.ay * s.az));
            }
          }
        }
      }
    }
    if (gfb != nullptr) {
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int lw = wv * NW + k;  // own slots: nobody else reads or writes them in this phase
        sF[0][lw][lane] = acc[k][0] * k0;
        sF[1][lw][lane] = acc[k][1] * k1;
        sF[2][lw][lane] = acc[k][2] * k2;
      }
    }
    __syncthreads();
    if (d + 1 < dEnd) {  // next slice's tiles fly during the store phase
      const float* f = fb + (size_t)(d + 1) * HW;
      M::load(f, p, h0, w0, r0); M::load(f + vol, p, h0, w0, r1); M::load(f + 2 * vol, p, h0, w0, r2);
      M::load(gob + (size_t)(d + 1) * HW, p, h0, w0, rg);
    }
    if (gfb != nullptr) {
      float* g = gfb + (size_t)d * HW;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t o = (size_t)c * vol + (size_t)d * HW;
        M::store(g + (size_t)c * vol, p, h0, w0, sF[c], gab[0] ? gab[0] + o : nullptr, gab[1] ? gab[1] + o : nullptr,
                 gab[2] ? gab[2] + o : nullptr);
      }
    }
    __syncthreads();  // stores have read sF before the next slice's flow overwrites it
  }
}

#include "warp3d_rc.hpp"

int make_params(W3P& p, int B, int C, const int* in_dhw, int D, int H, int W) {
  const int Di = in_dhw ? in_dhw[0] : D, Hi = in_dhw ? in_dhw[1] : H, Wi = in_dhw ? in_dhw[2] : W;
  if (B < 1 || C < 1 || D < 2 || H < 2 || W < 2 || Di < 2 || Hi < 2 || Wi < 2) return FS_ERR_SHAPE;
  if ((long long)D * H * W >= (1ll << 31)) return FS_ERR_SHAPE;
  if ((long long)H * W * 4 >= (1ll << 31)) return FS_ERR_SHAPE;  // 31-bit byte offsets inside one [H][W] plane (tile DMA)
  // 32-bit byte offsets into one input volume, 24-bit index products
  if ((long long)Di * Hi * Wi * 4 >= (1ll << 32) || Wi >= (1 << 24) || (long long)Di * Hi >= (1 << 24))
    return FS_ERR_SHAPE;
  p.B = B; p.C = C; p.D = D; p.H = H; p.W = W;
  p.Di = Di; p.Hi = Hi; p.Wi = Wi;
  p.tilesH = fs::cdiv(H, TH);
  p.tilesW = fs::cdiv(W, TW);
  static const int dc_env = (int)FS_AB_ENV_LL("FLOWSCI_W3_DC", 4);
  p.dc = dc_env;  // d-slices per workgroup: amortises the tile set-up, keeps >= 8 K workgroups at 256^3
  p.nDC = fs::cdiv(D, p.dc);
  if ((long long)B * p.nDC * p.tilesH * p.tilesW >= (1ll << 31)) return FS_ERR_SHAPE;
  // fp32 like the reference: linspace step (end-start)/(steps-1); divisor (dim-1.0)/2.0
  p.stepD = 2.0f / (float)(D - 1);
  p.stepH = 2.0f / (float)(H - 1);
  p.stepW = 2.0f / (float)(W - 1);
  p.rD = 1.0f / (((float)Di - 1.0f) / 2.0f);
  p.rH = 1.0f / (((float)Hi - 1.0f) / 2.0f);
  p.rW = 1.0f / (((float)Wi - 1.0f) / 2.0f);
  p.mD = (float)(Di - 1); p.mH = (float)(Hi - 1); p.mW = (float)(Wi - 1);
  p.rowB = (unsigned)Wi * 4u;
  p.planeB = (unsigned)Hi * (unsigned)Wi * 4u;
  return FS_OK;
}

bool vec_ok(const W3P& p, const void* a, const void* b, const void* c, const void* d) {
  auto al = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return (p.W % 4 == 0) && al(a) && al(b) && al(c) && al(d);
}

int launch_fwd(const W3Fwd& io, int npair, const float* flow, const UpP* up, W3P& p, fs_stream_t stream) {
  p.flowC = 3 * npair;
  hipStream_t st = (hipStream_t)stream;
  if (up != nullptr) {
    const bool vec = vec_ok(p, up->prev, up->fout, io.out[0], io.out[1]);
    // 4 x factor slices per workgroup: the low-resolution brick (staged once per workgroup) then carries 4 source slices + 2
    // of halo instead of 2 + 2 -- its re-reads were the 1.26x HBM traffic of round 3 (profiles/r03_pmc_traffic.json)
    p.dc = (int)FS_AB_ENV_LL("FLOWSCI_W3_UPS_DC", 4 * (up->rs < 0.3f ? 4 : 2));
    p.nDC = fs::cdiv(p.D, p.dc);
    const dim3 g((unsigned)((long long)p.B * p.nDC * p.tilesH * p.tilesW), npair);
    if (vec) hipLaunchKernelGGL((warp3d_fwd_kernel<512, true, true>), g, dim3(512), 0, st, io, flow, *up, p);
    else hipLaunchKernelGGL((warp3d_fwd_kernel<512, false, true>), g, dim3(512), 0, st, io, flow, *up, p);
  } else {
    const bool vec = vec_ok(p, flow, io.out[0], io.out[1], nullptr);
    // round 5: ring pipeline with the gather source in an LDS row cache (warp3d_rc.hpp) wherever its window fits the volume
    static const int rc_mode = (int)FS_AB_ENV_LL("FLOWSCI_W3_RC", 1);  // ablation build: 0 = round-4 kernels, 11.. = DBG forms
    if (vec && rc_mode != 0 && rc::applicable(p, io.in[0], io.in[1])) {
      p.dc = (int)FS_AB_ENV_LL("FLOWSCI_W3_RC_DC", rc::pick_dc(p, npair));
      p.nDC = fs::cdiv(p.D, p.dc);
      const dim3 gc((unsigned)((long long)p.B * p.nDC * p.tilesH * p.tilesW), npair);
      const dim3 bc(64 * (NCW + 2));
#ifdef FS_ABLATION
      if (rc_mode == 11) { hipLaunchKernelGGL((rc::warp3d_rc_kernel<false, 2, 6, 1>), gc, bc, 0, st, io, W3Bwd{}, flow, nullptr, W3Add{}, p); FS_LAUNCH_CHECK(); return FS_OK; }
      if (rc_mode == 12) { hipLaunchKernelGGL((rc::warp3d_rc_kernel<false, 2, 6, 2>), gc, bc, 0, st, io, W3Bwd{}, flow, nullptr, W3Add{}, p); FS_LAUNCH_CHECK(); return FS_OK; }
      if (rc_mode == 13) { hipLaunchKernelGGL((rc::warp3d_rc_kernel<false, 2, 6, 3>), gc, bc, 0, st, io, W3Bwd{}, flow, nullptr, W3Add{}, p); FS_LAUNCH_CHECK(); return FS_OK; }
#endif
      hipLaunchKernelGGL((rc::warp3d_rc_kernel<false, 2, 6>), gc, bc, 0, st, io, W3Bwd{}, flow, nullptr, W3Add{}, p);
      FS_LAUNCH_CHECK();
      return FS_OK;
    }
    // the ring kernel walks 16 slices per workgroup (one workgroup per CU; 2 048 workgroups at 2 x 256^3)
    p.dc = (int)FS_AB_ENV_LL("FLOWSCI_W3_RING_DC", 16);
    p.nDC = fs::cdiv(p.D, p.dc);
    const dim3 gr((unsigned)((long long)p.B * p.nDC * p.tilesH * p.tilesW), npair);
#ifdef FS_ABLATION
    static const int ring = (int)FS_AB_ENV_LL("FLOWSCI_W3_RING", 0);  // 3..5: that many flow stages; +10: non-temporal DMA
#define W3_RING_CASE(R, A) \
    if (vec && ring == R + (A ? 10 : 0)) { hipLaunchKernelGGL((warp3d_fwd_ring_kernel<true, A, R>), gr, dim3(NT2), 0, st, io, flow, p); FS_LAUNCH_CHECK(); return FS_OK; }
    W3_RING_CASE(3, 0) W3_RING_CASE(4, 0) W3_RING_CASE(5, 0) W3_RING_CASE(3, 2) W3_RING_CASE(5, 2)
    if (vec && ring == 101) { hipLaunchKernelGGL((warp3d_fwd_ring_kernel<true, 2, 4, 1>), gr, dim3(NT2), 0, st, io, flow, p); FS_LAUNCH_CHECK(); return FS_OK; }
    if (vec && ring == 103) { hipLaunchKernelGGL((warp3d_fwd_ring_kernel<true, 2, 4, 3>), gr, dim3(NT2), 0, st, io, flow, p); FS_LAUNCH_CHECK(); return FS_OK; }
    if (vec && ring == 104) { hipLaunchKernelGGL((warp3d_fwd_ring_kernel<true, 2, 4, 4>), gr, dim3(NT2), 0, st, io, flow, p); FS_LAUNCH_CHECK(); return FS_OK; }
    if (vec && ring == 105) { hipLaunchKernelGGL((warp3d_fwd_ring_kernel<true, 2, 4, 5>), gr, dim3(NT2), 0, st, io, flow, p); FS_LAUNCH_CHECK(); return FS_OK; }
    if (vec && ring == 102) { hipLaunchKernelGGL((warp3d_fwd_ring_kernel<true, 2, 4, 2>), gr, dim3(NT2), 0, st, io, flow, p); FS_LAUNCH_CHECK(); return FS_OK; }
#endif
    if (vec) hipLaunchKernelGGL((warp3d_fwd_ring_kernel<true, 2, 4>), gr, dim3(NT2), 0, st, io, flow, p);
    else hipLaunchKernelGGL((warp3d_fwd_ring_kernel<false, 0, 3>), gr, dim3(NT2), 0, st, io, flow, p);
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int NT, bool VEC>
void launch_bwd_t(const W3Bwd& io, const dim3& g, bool with_gin, const float* flow, float* gflow,
                  const W3Add& gadd, const W3P& p, hipStream_t st) {
  if (with_gin)
    hipLaunchKernelGGL((warp3d_bwd_kernel<NT, VEC, true>), g, dim3(NT), 0, st, io, flow, gflow, gadd, p);
  else
    hipLaunchKernelGGL((warp3d_bwd_kernel<NT, VEC, false>), g, dim3(NT), 0, st, io, flow, gflow, gadd, p);
}

int launch_bwd(const W3Bwd& io, int npair, bool with_gin, const float* flow, float* gflow, const W3Add& gadd,
               W3P& p, fs_stream_t stream) {
  const unsigned grid = (unsigned)((long long)p.B * p.nDC * p.tilesH * p.tilesW);
  p.flowC = 3 * npair;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = vec_ok(p, flow, gflow, io.gout[0], io.gout[1]) && vec_ok(p, gadd.a[0], gadd.a[1], gadd.a[2], nullptr) &&
                   gadd.bs[0] % 4 == 0 && gadd.bs[1] % 4 == 0 && gadd.bs[2] % 4 == 0 && io.gbs[0] % 4 == 0 && io.gbs[1] % 4 == 0;
  static const int rc_mode = (int)FS_AB_ENV_LL("FLOWSCI_W3_RC", 1);
  if (vec && rc_mode != 0 && !with_gin && gflow != nullptr && rc::applicable(p, io.in[0], io.in[1])) {
    p.dc = (int)FS_AB_ENV_LL("FLOWSCI_W3_RC_DC", rc::pick_dc(p, npair));
    p.nDC = fs::cdiv(p.D, p.dc);
    const dim3 gc((unsigned)((long long)p.B * p.nDC * p.tilesH * p.tilesW), npair);
    const dim3 bc(64 * (NCW + 4));
#ifdef FS_ABLATION
    if (rc_mode == 11) { hipLaunchKernelGGL((rc::warp3d_rc_kernel<true, 4, 5, 1>), gc, bc, 0, st, W3Fwd{}, io, flow, gflow, gadd, p); FS_LAUNCH_CHECK(); return FS_OK; }
    if (rc_mode == 12) { hipLaunchKernelGGL((rc::warp3d_rc_kernel<true, 4, 5, 2>), gc, bc, 0, st, W3Fwd{}, io, flow, gflow, gadd, p); FS_LAUNCH_CHECK(); return FS_OK; }
    if (rc_mode == 13) { hipLaunchKernelGGL((rc::warp3d_rc_kernel<true, 4, 5, 3>), gc, bc, 0, st, W3Fwd{}, io, flow, gflow, gadd, p); FS_LAUNCH_CHECK(); return FS_OK; }
#endif
    hipLaunchKernelGGL((rc::warp3d_rc_kernel<true, 4, 5>), gc, bc, 0, st, W3Fwd{}, io, flow, gflow, gadd, p);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
  const dim3 g(grid, npair);
  if (vec) launch_bwd_t<256, true>(io, g, with_gin, flow, gflow, gadd, p, st);
  else launch_bwd_t<256, false>(io, g, with_gin, flow, gflow, gadd, p, st);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // namespace

extern "C" int fs_warp3d_fwd(const float* in, const float* flow, float* out, int B, int C,
                             const int* in_dhw, int D, int H, int W, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(in); FS_REQUIRE_PTR(flow); FS_REQUIRE_PTR(out);
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W);
  if (rc != FS_OK) return rc;
  W3Fwd io = {{in, nullptr}, {out, nullptr}};
  return launch_fwd(io, 1, flow, nullptr, p, stream);
}

extern "C" int fs_warp3d_bwd(const float* in, const float* flow, const float* grad_out,
                             float* grad_in, float* grad_flow, int B, int C, const int* in_dhw,
                             int D, int H, int W, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(in); FS_REQUIRE_PTR(flow); FS_REQUIRE_PTR(grad_out);
  if (grad_in == nullptr && grad_flow == nullptr) return FS_ERR_NULLPTR;
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W);
  if (rc != FS_OK) return rc;
  W3Bwd io = {{in, nullptr}, {grad_out, nullptr}, {grad_in, nullptr}};
  return launch_bwd(io, 1, grad_in != nullptr, flow, grad_flow, W3Add{}, p, stream);
}

extern "C" int fs_warp3d_pair_fwd(const float* img0, const float* img1, const float* flow6,
                                  float* out0, float* out1, int B, int C, const int* in_dhw,
                                  int D, int H, int W, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow6);
  FS_REQUIRE_PTR(out0); FS_REQUIRE_PTR(out1);
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W);
  if (rc != FS_OK) return rc;
  W3Fwd io = {{img0, img1}, {out0, out1}};
  return launch_fwd(io, 2, flow6, nullptr, p, stream);
}

extern "C" int fs_warp3d_pair_bwd(const float* img0, const float* img1, const float* flow6,
                                  const float* grad_out0, const float* grad_out1, float* grad_img0,
                                  float* grad_img1, float* grad_flow6, int B, int C,
                                  const int* in_dhw, int D, int H, int W, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow6);
  FS_REQUIRE_PTR(grad_out0); FS_REQUIRE_PTR(grad_out1);
  if ((grad_img0 == nullptr) != (grad_img1 == nullptr)) return FS_ERR_NULLPTR;  // both or neither
  if (grad_img0 == nullptr && grad_flow6 == nullptr) return FS_ERR_NULLPTR;
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W);
  if (rc != FS_OK) return rc;
  W3Bwd io = {{img0, img1}, {grad_out0, grad_out1}, {grad_img0, grad_img1}};
  return launch_bwd(io, 2, grad_img0 != nullptr, flow6, grad_flow6, W3Add{}, p, stream);
}

// fs_warp3d_pair_bwd with the gradient that reaches the flow from its OTHER consumers (`grad_flow_add`,
// nullable, [B,6,D,H,W]) summed into grad_flow6 by the same launch: the flow of an IFNet block feeds the
// warp, the next block's input, the next block's accumulation and the distillation term, and autograd would
// otherwise add the warp's gradient to the others in a separate 2.4 GB pass.  grad_flow_add may BE grad_flow6
// (in-place accumulation).
extern "C" int fs_warp3d_pair_bwd_acc(const float* img0, const float* img1, const float* flow6,
                                      const float* grad_out0, const float* grad_out1, float* grad_img0,
                                      float* grad_img1, const float* grad_flow_add, float* grad_flow6, int B, int C,
                                      const int* in_dhw, int D, int H, int W, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow6);
  FS_REQUIRE_PTR(grad_out0); FS_REQUIRE_PTR(grad_out1); FS_REQUIRE_PTR(grad_flow6);
  if ((grad_img0 == nullptr) != (grad_img1 == nullptr)) return FS_ERR_NULLPTR;
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W);
  if (rc != FS_OK) return rc;
  W3Bwd io = {{img0, img1}, {grad_out0, grad_out1}, {grad_img0, grad_img1}};
  const long long bs = (long long)6 * D * H * W;
  return launch_bwd(io, 2, grad_img0 != nullptr, flow6, grad_flow6, W3Add{{grad_flow_add, nullptr, nullptr}, {bs, 0, 0}}, p,
                    stream);
}

// The same with up to THREE such gradients, each a [B,6,D,H,W] tensor or a 6-channel slice of a wider one
// (`batch_stride*` in floats): the flow of an IFNet block is consumed by the next block's input concatenation
// (its gradient arrives as channels 5..10 of the 11-channel input gradient), the next block's accumulation and
// the distillation term; handing each consumer its own alias of the flow keeps autograd from summing them in
// two extra passes over 805 MB tensors (opticalflowscivis_amd/ops.py::_WarpPairAcc).
extern "C" int fs_warp3d_pair_bwd_acc3(const float* img0, const float* img1, const float* flow6,
                                       const float* grad_out0, long long gout_batch_stride0, const float* grad_out1,
                                       long long gout_batch_stride1, float* grad_img0,
                                       float* grad_img1, const float* add0, long long batch_stride0,
                                       const float* add1, long long batch_stride1, const float* add2,
                                       long long batch_stride2, float* grad_flow6, int B, int C, const int* in_dhw,
                                       int D, int H, int W, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(flow6);
  FS_REQUIRE_PTR(grad_out0); FS_REQUIRE_PTR(grad_out1); FS_REQUIRE_PTR(grad_flow6);
  if ((grad_img0 == nullptr) != (grad_img1 == nullptr)) return FS_ERR_NULLPTR;
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W);
  if (rc != FS_OK) return rc;
  const long long fl = (long long)6 * D * H * W;
  if ((add0 && batch_stride0 < fl) || (add1 && batch_stride1 < fl) || (add2 && batch_stride2 < fl)) return FS_ERR_ARG;
  const long long gl = (long long)C * D * H * W;  // 0 = dense
  if ((gout_batch_stride0 && gout_batch_stride0 < gl) || (gout_batch_stride1 && gout_batch_stride1 < gl)) return FS_ERR_ARG;
  W3Bwd io = {{img0, img1}, {grad_out0, grad_out1}, {grad_img0, grad_img1}, {gout_batch_stride0, gout_batch_stride1}};
  return launch_bwd(io, 2, grad_img0 != nullptr, flow6, grad_flow6,
                    W3Add{{add0, add1, add2}, {batch_stride0, batch_stride1, batch_stride2}}, p, stream);
}

extern "C" int fs_warp3d_kernel_id(const float* in0, const float* in1, const float* flow, int B, int C, const int* in_dhw,
                                   int D, int H, int W, int backward, int with_grad_in) {
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, D, H, W);
  if (rc != FS_OK) return -rc;
  // launch_fwd / launch_bwd's own conditions (the remaining operands of a call are 16-byte aligned whenever these are:
  // the binding allocates them)
  const bool vec = vec_ok(p, flow, nullptr, nullptr, nullptr);
  static const int rc_mode = (int)FS_AB_ENV_LL("FLOWSCI_W3_RC", 1);
  if (vec && rc_mode != 0 && !(backward && with_grad_in) && rc::applicable(p, in0, in1)) return FS_W3_KERNEL_RC;
  return FS_W3_KERNEL_GATHER;
}

// "upsample flow x scale -> warp" in one kernel.  flow_out = prev_flow + scale *
// trilinear_upsample(delta, factor) (prev_flow nullable), out0 = warp(img0, flow_out[:, :3]),
// out1 = warp(img1, flow_out[:, 3:6]); delta [B,6,Ds,Hs,Ws], everything else at factor x that extent.
extern "C" int fs_upsample_warp3d_pair_fwd(const float* img0, const float* img1, const float* delta,
                                           const float* prev_flow, float* flow_out, float* out0, float* out1,
                                           int B, int C, const int* in_dhw, int Ds, int Hs, int Ws, int factor,
                                           float scale, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(img0); FS_REQUIRE_PTR(img1); FS_REQUIRE_PTR(delta); FS_REQUIRE_PTR(flow_out);
  FS_REQUIRE_PTR(out0); FS_REQUIRE_PTR(out1);
  if (factor != 2 && factor != 4) return FS_ERR_ARG;
  if (Ds < 1 || Hs < 1 || Ws < 1) return FS_ERR_SHAPE;
  if ((long long)Ds * Hs * Ws * factor * factor * factor >= (1ll << 31)) return FS_ERR_SHAPE;
  W3P p;
  const int rc = make_params(p, B, C, in_dhw, Ds * factor, Hs * factor, Ws * factor);
  if (rc != FS_OK) return rc;
  UpP u = {delta, prev_flow, flow_out, Ds, Hs, Ws, 1.0f / (float)factor, scale};
  W3Fwd io = {{img0, img1}, {out0, out1}};
  return launch_fwd(io, 2, nullptr, &u, p, stream);
}

extern "C" int fs_interp3d_bwd_scaled(const float* grad_out, float* grad_in, float* ws, int B, int C, int Din,
                                      int Hin, int Win, int Dout, int Hout, int Wout, int factor, int upsample,
                                      float scale, fs_stream_t stream);

// Backward of the fused node: grad_flow_total = d(warps)/d(flow) + grad_flow_add (the gradient reaching
// flow_out from its other consumers, nullable; may alias grad_flow_total) -- which is also the gradient of
// prev_flow -- and grad_delta = scale * adjoint_upsample(grad_flow_total).  ws: B*6*(D*H*Ws + D*Hs*Ws) floats.
extern "C" int fs_upsample_warp3d_pair_bwd3(const float* img0, const float* img1, const float* flow6,
                                            const float* grad_out0, long long gout_batch_stride0,
                                            const float* grad_out1, long long gout_batch_stride1, const float* add0,
                                            long long batch_stride0, const float* add1, long long batch_stride1,
                                            const float* add2, long long batch_stride2, float* grad_flow_total,
                                            float* grad_delta, float* ws, int B, int C, const int* in_dhw, int Ds,
                                            int Hs, int Ws, int factor, float scale, fs_stream_t stream) {
  FS_REQUIRE_PTR(grad_delta); FS_REQUIRE_PTR(ws);
  if (factor != 2 && factor != 4) return FS_ERR_ARG;
  if (Ds < 1 || Hs < 1 || Ws < 1) return FS_ERR_SHAPE;
  if ((long long)Ds * Hs * Ws * factor * factor * factor >= (1ll << 31)) return FS_ERR_SHAPE;
  const int D = Ds * factor, H = Hs * factor, W = Ws * factor;
  int rc = fs_warp3d_pair_bwd_acc3(img0, img1, flow6, grad_out0, gout_batch_stride0, grad_out1, gout_batch_stride1,
                                   nullptr, nullptr, add0, batch_stride0, add1,
                                   batch_stride1, add2, batch_stride2, grad_flow_total, B, C, in_dhw, D, H, W, stream);
  if (rc != FS_OK) return rc;
  return fs_interp3d_bwd_scaled(grad_flow_total, grad_delta, ws, B, 6, Ds, Hs, Ws, D, H, W, factor, 1, scale,
                                stream);
}

extern "C" int fs_upsample_warp3d_pair_bwd(const float* img0, const float* img1, const float* flow6,
                                           const float* grad_out0, const float* grad_out1,
                                           const float* grad_flow_add, float* grad_flow_total, float* grad_delta,
                                           float* ws, int B, int C, const int* in_dhw, int Ds, int Hs, int Ws,
                                           int factor, float scale, fs_stream_t stream) {
  FS_REQUIRE_PTR(grad_delta); FS_REQUIRE_PTR(ws);
  if (factor != 2 && factor != 4) return FS_ERR_ARG;
  if (Ds < 1 || Hs < 1 || Ws < 1) return FS_ERR_SHAPE;
  if ((long long)Ds * Hs * Ws * factor * factor * factor >= (1ll << 31)) return FS_ERR_SHAPE;
  const int D = Ds * factor, H = Hs * factor, W = Ws * factor;
  int rc = fs_warp3d_pair_bwd_acc(img0, img1, flow6, grad_out0, grad_out1, nullptr, nullptr, grad_flow_add,
                                  grad_flow_total, B, C, in_dhw, D, H, W, stream);
  if (rc != FS_OK) return rc;
  return fs_interp3d_bwd_scaled(grad_flow_total, grad_delta, ws, B, 6, Ds, Hs, Ws, D, H, W, factor, 1, scale,
                                stream);
}
