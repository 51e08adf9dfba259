// corr2d.hip -- local-window correlation (cost volume) of UPFlow for gfx950.
//
// Replaces correlation_cuda.forward/backward as called by CorrelationFunction
// (UPFlow/model/correlation_package/correlation.py:26-27,42-43) with pad = max_displacement = md,
// kernel_size = 1, stride1 = stride2 = 1, corr_multiply = 1 (UPFlow/model/upflow.py:649,652):
//
//   out[b, (dy+md)(2md+1) + (dx+md), y, x] = (1/C) sum_c f1[b,c,y,x] * f2[b,c,y+dy,x+dx]
//
// f2 reads as 0 outside the image; dy-major channel order (pinned by Corr_pyTorch,
// UPFlow/utils/pytorch_correlation.py:27-50).
//
// Forward.  One workgroup = one 8x32-pixel tile of one sample, 2md+1 waves: wave `dy` owns one
// displacement row, lane = a quad of 4 consecutive pixels, so each lane keeps 4 x (2md+1)
// accumulators.  Channels are streamed through LDS in chunks of 8: the f2 search window
// (tile + md halo) and the f1 tile are loaded once per chunk with coalesced rows, then every
// lane reads its f2 row segment (4 + 2md floats) as ds_read_b128 and does 4*(2md+1) FMAs per
// 3 LDS instructions -- VALU-bound, not LDS-bound.  HBM traffic is the algorithmic
// 4*(2C + (2md+1)^2) B/pixel; the halo re-reads of f2 are served by L2.
// The channel reduction of the tiled kernel runs in registers (lanes = pixels).  That form needs pixels:
// the three coarsest UPFlow levels at C3 are (C, h, w) = (196, 3, 8), (128, 5, 15), (96, 10, 29) -- 24 .. 290
// pixels per sample, C >> pixels -- where an 8x32 tile is mostly empty and the kernel is a serial chain of
// C/8 stage-barrier-compute rounds (86 us for 150 K output floats).  Those levels (h*w <= 512) run the
// direct kernels below instead: no LDS, no barrier, one thread per (sample, displacement, pixel) output and
// channel slice -- the C range is split over 1..8 lane groups of a wave and reduced with wave shuffles
// (2 * log2(slices) DPP steps per output) -- so that even a B = 2 launch fills the chip.
//
// Backward.  grad_f1[c,p] = (1/C) sum_d g[d,p] f2[c,p+d] is a gather;  grad_f2 is the same gather
// with the roles swapped and the displacement negated:  grad_f2[c,q] = (1/C) sum_d gT[d,q] f1[c,q+d]
// with gT[d,q] = g[-d, q+d] (0 when q+d is outside).  One launch computes both (blockIdx.z picks),
// no atomics, bitwise reproducible: thread = pixel, its (2md+1)^2 upstream gradients live in
// registers, channel chunks of the other feature map are staged in LDS.
#include "common.hpp"

namespace {

constexpr int TY = 8, TX = 32, CC = 8;
// displacement rows per wave of the tiled kernels (forward: 1 -> 2md+1 waves, 36 accumulators per lane at md = 4:
// 1 / 2 / 3 rows per wave measured 24 / 27 / 37 us at the (64, 19, 57) level, equal at (32, 38, 113))
#define FS_C2_FWD_DPW 1
#define FS_C2_BWD_DPW 3

// Two problems of one shape per launch (UPFlow correlates both directions at every level,
// upflow.py:649,652); pointers of the second problem may equal the first's.
struct C2Set {
  const float* f1[2];
  const float* f2[2];
  const float* st1[2];  // per-(b,c) (mean, rstd) of f1 / f2, or nullptr: plain correlation
  const float* st2[2];
  float* out[2];        // forward: cost volume;  backward: unused
  const float* gout[2];
  float* g1[2];
  float* g2[2];
  int nsets;
};


// 4 consecutive floats at any dword alignment: gfx950 global memory takes multi-dword accesses at 4-byte
// alignment (feature rows are W = 57, 113, ... floats long, so row starts are not 16-byte aligned)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void store4_masked(float* __restrict__ p, float4 v, int mask) {
  if (mask == 0xF) {
    f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *reinterpret_cast<f4u*>(p) = t;
  } else {
    if (mask & 1) p[0] = v.x;
    if (mask & 2) p[1] = v.y;
    if (mask & 4) p[2] = v.z;
    if (mask & 8) p[3] = v.w;
  }
}

// bit e: column gx + e of a valid row is inside the image
__device__ __forceinline__ int row_mask(bool rowok, int gx, int W) {
  if (!rowok) return 0;
  return ((unsigned)gx < (unsigned)W ? 1 : 0) | ((unsigned)(gx + 1) < (unsigned)W ? 2 : 0) |
         ((unsigned)(gx + 2) < (unsigned)W ? 4 : 0) | ((unsigned)(gx + 3) < (unsigned)W ? 8 : 0);
}

__device__ __forceinline__ float4 keep4(float4 v, int mask) {
  return make_float4((mask & 1) ? v.x : 0.f, (mask & 2) ? v.y : 0.f, (mask & 4) ? v.z : 0.f, (mask & 8) ? v.w : 0.f);
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id; each XCD has its own L2.  A 1-D grid is re-mapped so
// that every XCD works on one contiguous range of logical tiles: x / y neighbours (which share halo rows of the
// inputs and, with W % 32 != 0, cache lines of the outputs) then meet in the same L2.
__device__ __forceinline__ long long xcd_tile(long long id, long long total) {
  const long long per = total / 8;
  return id < per * 8 ? (id & 7) * per + (id >> 3) : id;
}

// lane -> (row, quad) of the 8 x 8 quads of a tile, chosen for ds_read_b128: the LDS serves a wave's 16-byte reads in
// four fixed groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32), 64 banks of 4 bytes.  A
// group is mapped to the 8 quads of row r and of row r + 4: with the staged row pitch of 40 floats the two rows
// start 160 floats = 32 banks apart, so the 16 reads cover all 64 banks once -- conflict-free -- where the plain
// (lane / 8, lane % 8) order gives 2- to 3-way conflicts.
__device__ __forceinline__ void lane_quad(int lane, int& qy, int& qx) {
  const int code = (0x73261540u >> (4 * ((lane & 31) >> 2))) & 7;  // (group-in-half << 2) | rank of this lane quartet
  const int g = (code & 3) * 4 + (lane & 3);                        // 0 .. 15 inside the group
  qy = 2 * (lane >> 5) + (code >> 2) + 4 * (g >> 3);
  qx = (g & 7) * 4;
}

#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass has neither the buffer-resource type nor its builtins)
// Loads go through a buffer descriptor over the rest of the tensor: every staged vector is ONE unconditional
// buffer_load_dwordx4 -- no per-lane branch, so the loads of a chunk are all in flight together (with a masked
// global load the compiler waits after every vector) -- and the elements outside the image are cleared by mask
// afterwards.  A vector that leaves the tensor (before its first or past its last float) reads 0 there: the
// range check is per dword.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* base, long long floats) {
  const long long bytes = floats * 4;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes,
                                           0x00020000);
}
__device__ __forceinline__ float4 bload4(rsrc_t r, int off_floats) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, off_floats * 4, 0, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

// The staged window of one tile: CH channels x (TY + 2 md) rows x SV float4, as per-thread items whose geometry
// (offset inside a channel group, LDS offset, element mask) is fixed for the whole workgroup.
template <int MD, int CH, int NT>
struct Window {
  static constexpr int SR = TY + 2 * MD, SV = (TX + 2 * MD + 3) / 4;
  static constexpr int SW = 40;  // row pitch: see lane_quad
  static_assert(4 * SV <= SW, "staged row fits the pitch");
  static constexpr int N = CH * SR * SV, K = (N + NT - 1) / NT;
  static constexpr int FLOATS = CH * SR * SW;
  int off[K], lo[K], cm[K];  // offset (floats), LDS offset, (channel << 4) | mask

  __device__ __forceinline__ void init(int t, int y0, int x0, int H, int W) {
    const int HW = H * W;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = t + NT * k;
      const int c = i / (SR * SV), rem = i - c * (SR * SV);
      const int r = rem / SV, v = rem - r * SV;
      const int gy = y0 + r - MD, gx = x0 + 4 * v - MD;
      off[k] = c * HW + gy * W + gx;
      lo[k] = (c * SR + r) * SW + 4 * v;
      cm[k] = (c << 4) | (i < N ? row_mask(gy >= 0 && gy < H, gx, W) : 0);
    }
  }
  __device__ __forceinline__ void load(float4 (&pf)[K], rsrc_t r, int coff) const {
#pragma unroll
    for (int k = 0; k < K; ++k) pf[k] = bload4(r, coff + off[k]);
  }
  // st (nullable) = (mean, rstd) of the group's first channel: normalize_features folded in (§8f.4) -- the zero
  // padding applies AFTER it
  __device__ __forceinline__ void put(const float4 (&pf)[K], float* __restrict__ s, int t,
                                      const float* __restrict__ st, int nch) const {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (t + NT * k >= N) continue;
      const int c = cm[k] >> 4;
      float4 v = pf[k];
      if (st != nullptr) {
        const float2 q = *reinterpret_cast<const float2*>(st + 2 * (c < nch ? c : 0));
        v = make_float4((v.x - q.x) * q.y, (v.y - q.x) * q.y, (v.z - q.x) * q.y, (v.w - q.x) * q.y);
      }
      *reinterpret_cast<float4*>(s + lo[k]) = keep4(v, c < nch ? (cm[k] & 0xF) : 0);
    }
  }
};
#endif

// Forward.  One workgroup = one 8 x 32-pixel tile of one sample; lane = a quad of 4 consecutive pixels, wave w owns
// the displacement rows 3 w .. 3 w + 2, so a lane keeps 4 x (2md+1) x 3 accumulators and does 36 FMAs (md = 4) per
// 3 ds_read_b128 of an f2 row.  Channels stream through two LDS buffers in chunks of 8: the loads of chunk k + 1
// (16-byte, dword-aligned; geometry precomputed, no div/mod, no branches) are in flight while chunk k is
// computed; one barrier per chunk.  The epilogue writes 16-byte vectors.
template <int MD, int DPW>
__global__ __launch_bounds__(64 * ((2 * MD + DPW) / DPW)) void corr2d_fwd_q_kernel(C2Set a, int B, int C, int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ND = 2 * MD + 1;
  constexpr int NWV = (ND + DPW - 1) / DPW, NT = 64 * NWV;
  using W2 = Window<MD, CC, NT>;
  constexpr int SR = W2::SR, SW = W2::SW;
  constexpr int N1 = CC * TY * (TX / 4), K1 = (N1 + NT - 1) / NT;
  constexpr int P1 = 40;  // f1 row pitch, as the window's (lane_quad)
  constexpr int BUF = W2::FLOATS + CC * TY * P1;
  constexpr int RV = (4 + 2 * MD + 3) / 4;  // float4 reads per f2 row segment
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
  long long tile = xcd_tile(blockIdx.x, gridDim.x);
  const int x0 = (int)(tile % ntx) * TX; tile /= ntx;
  const int y0 = (int)(tile % nty) * TY; tile /= nty;
  const int b = (int)(tile % B), set = (int)(tile / B);
  const float* __restrict__ st1 = a.st1[set];
  const float* __restrict__ st2 = a.st2[set];
  float* __restrict__ out = a.out[set];
  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  int qy, qx;
  lane_quad(lane, qy, qx);
  const int HW = H * W;
  const long long rest = (long long)(B - b) * C * HW;
  const rsrc_t r1 = make_rsrc(a.f1[set] + (size_t)b * C * HW, rest);
  const rsrc_t r2 = make_rsrc(a.f2[set] + (size_t)b * C * HW, rest);
  if (st1 != nullptr) { st1 += 2 * (size_t)b * C; st2 += 2 * (size_t)b * C; }

  W2 w2;
  w2.init(t, y0, x0, H, W);
  int off1[K1], lo1[K1], cm1[K1];
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    const int i = t + NT * k;
    const int c = i / (TY * (TX / 4)), rem = i - c * (TY * (TX / 4));
    const int r = rem / (TX / 4), v = rem - r * (TX / 4);
    const int gy = y0 + r, gx = x0 + 4 * v;
    off1[k] = c * HW + gy * W + gx;
    lo1[k] = W2::FLOATS + (c * TY + r) * P1 + 4 * v;
    cm1[k] = (c << 4) | (i < N1 ? row_mask(gy < H, gx, W) : 0);
  }
  float4 pf2[W2::K], pf1[K1];
  auto fetch = [&](int c0) {
    w2.load(pf2, r2, c0 * HW);
#pragma unroll
    for (int k = 0; k < K1; ++k) pf1[k] = bload4(r1, c0 * HW + off1[k]);
  };
  auto put = [&](float* s, int c0) {
    const int nch = C - c0;
    w2.put(pf2, s, t, st2 != nullptr ? st2 + 2 * c0 : nullptr, nch);
#pragma unroll
    for (int k = 0; k < K1; ++k) {
      if (t + NT * k >= N1) continue;
      const int c = cm1[k] >> 4;
      float4 v = pf1[k];
      if (st1 != nullptr) {
        const float2 q = *reinterpret_cast<const float2*>(st1 + 2 * (c0 + (c < nch ? c : 0)));
        v = make_float4((v.x - q.x) * q.y, (v.y - q.x) * q.y, (v.z - q.x) * q.y, (v.w - q.x) * q.y);
      }
      *reinterpret_cast<float4*>(s + lo1[k]) = keep4(v, c < nch ? (cm1[k] & 0xF) : 0);
    }
  };

  float acc[DPW][4][ND];
#pragma unroll
  for (int d = 0; d < DPW; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < ND; ++j) acc[d][i][j] = 0.f;

  fetch(0);
  put(lds, 0);
  __syncthreads();
  int buf = 0;
  for (int c0 = 0; c0 < C; c0 += CC) {
    const bool more = c0 + CC < C;
    fetch(more ? c0 + CC : c0);  // unconditional (a conditional load becomes a copy behind a wait); unused at the end
    const float* s2 = lds + buf * BUF;
    const float* s1 = s2 + W2::FLOATS;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const float4 av4 = *reinterpret_cast<const float4*>(s1 + (c * TY + qy) * P1 + qx);
      const float av[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
      for (int d = 0; d < DPW; ++d) {
        const int dy = DPW * wv + d;
        if (dy >= ND) continue;  // wave-uniform
        float row[4 * RV];
        const float* rp = s2 + (c * SR + qy + dy) * SW + qx;
#pragma unroll
        for (int k = 0; k < RV; ++k) {
          const float4 v = *reinterpret_cast<const float4*>(rp + 4 * k);
          row[4 * k] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < ND; ++j) acc[d][i][j] = fmaf(av[i], row[i + j], acc[d][i][j]);
      }
    }
    if (more) put(lds + (buf ^ 1) * BUF, c0 + CC);
    __syncthreads();
    buf ^= 1;
  }

  const int y = y0 + qy, x = x0 + qx;
  const int smask = row_mask(y < H, x, W);
  if (smask == 0) return;
  const float fC = (float)C;
#pragma unroll
  for (int d = 0; d < DPW; ++d) {
    const int dy = DPW * wv + d;
    if (dy >= ND) continue;
    float* ob = out + ((size_t)b * ND * ND + (size_t)dy * ND) * HW + (size_t)y * W + x;
#pragma unroll
    for (int j = 0; j < ND; ++j)  // torch.mean = sum / C
      store4_masked(ob + (size_t)j * HW,
                    make_float4(acc[d][0][j] / fC, acc[d][1][j] / fC, acc[d][2][j] / fC, acc[d][3][j] / fC), smask);
  }
#endif
}

// Backward.  grad[c,p] = (1/C) sum_d g(d,p) * other[c, p+d];  first: (g = gout, other = f2) -> grad_f1;  second:
// (g = gout transposed on the fly, gT[d,q] = g[-d, q+d], other = f1) -> grad_f2.  One workgroup = one 8 x 32 tile,
// 16 channels of one sample; their `other` window is staged once (41 KB).  lane = a quad of 4 consecutive pixels,
// wave w owns the displacement rows 3 w .. 3 w + 2: it loads the 4 x (2md+1) upstream gradients of a row as
// 16-byte vectors (each gradient value is read once per workgroup) and runs them against all 16 channels --
// 36 FMAs (md = 4) per 3 ds_read_b128.  The three partial sums meet in LDS; wave 0 writes 16-byte vectors.
// No atomics, bitwise reproducible.
constexpr int CBW = 16;
template <int MD, int DPW>
__global__ __launch_bounds__(64 * ((2 * MD + DPW) / DPW)) void corr2d_bwd_q_kernel(C2Set a, int B, int C, int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ND = 2 * MD + 1;
  constexpr int NWV = (ND + DPW - 1) / DPW, NT = 64 * NWV;
  using WO = Window<MD, CBW, NT>;
  constexpr int SR = WO::SR, SW = WO::SW;
  constexpr int RV = (4 + 2 * MD + 3) / 4;
  constexpr int RED = (NWV - 1) * CBW * 4 * 64;  // the partial sums of waves 1.. (float4-interleaved by lane)
  constexpr int LDSF = WO::FLOATS > RED ? WO::FLOATS : RED;
  __shared__ __attribute__((aligned(16))) float s[LDSF];

  const int CG = (C + CBW - 1) / CBW;
  const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
  long long tile = xcd_tile(blockIdx.x, gridDim.x);
  const int x0 = (int)(tile % ntx) * TX; tile /= ntx;
  const int y0 = (int)(tile % nty) * TY; tile /= nty;
  const int cg = (int)(tile % CG); tile /= CG;
  const int b = (int)(tile % B); tile /= B;
  const bool second = (tile & 1) != 0;
  const int set = (int)(tile >> 1);
  float* __restrict__ grad = second ? a.g2[set] : a.g1[set];
  if (grad == nullptr) return;  // uniform per block
  const float* __restrict__ other = second ? a.f1[set] : a.f2[set];
  const float* __restrict__ ost = second ? a.st1[set] : a.st2[set];  // moments of `other` (NULL: plain)
  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  int qy, qx;
  lane_quad(lane, qy, qx);
  const int HW = H * W;
  const int c0 = cg * CBW;
  const int y = y0 + qy, x = x0 + qx;
  const int pmask = row_mask(y < H, x, W);  // this quad's pixels inside the image

  // this wave's upstream gradients: rows j = 3 wv .. 3 wv + 2 of the displacement window, all in flight at once
  const rsrc_t rg = make_rsrc(a.gout[set] + (size_t)b * ND * ND * HW, (long long)(B - b) * ND * ND * HW);
  float4 g[DPW][ND];
#pragma unroll
  for (int d = 0; d < DPW; ++d) {
    const int j = DPW * wv + d;
    if (j >= ND) continue;  // wave-uniform
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      // first: g[(j,i), p];  second: gT[(j,i), q] = g[(ND-1-j, ND-1-i), q + d]
      const int plane = second ? (ND - 1 - j) * ND + (ND - 1 - i) : j * ND + i;
      const int yy = second ? y + (j - MD) : y, xx = second ? x + (i - MD) : x;
      g[d][i] = bload4(rg, plane * HW + yy * W + xx);
    }
  }
  {
    WO wo;
    wo.init(t, y0, x0, H, W);
    float4 pf[WO::K];
    wo.load(pf, make_rsrc(other + ((size_t)b * C + c0) * HW, ((long long)(B - b) * C - c0) * HW), 0);
    wo.put(pf, s, t, ost != nullptr ? ost + 2 * ((size_t)b * C + c0) : nullptr, C - c0);
  }
#pragma unroll
  for (int d = 0; d < DPW; ++d) {
    const int j = DPW * wv + d;
    if (j >= ND) continue;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int yy = second ? y + (j - MD) : y, xx = second ? x + (i - MD) : x;
      g[d][i] = keep4(g[d][i], row_mask(yy >= 0 && yy < H, xx, W) & pmask);
    }
  }
  __syncthreads();

  float acc[CBW][4];
#pragma unroll
  for (int c = 0; c < CBW; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;

#pragma unroll
  for (int d = 0; d < DPW; ++d) {
    const int j = DPW * wv + d;
    if (j >= ND) continue;  // wave-uniform
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
      float row[4 * RV];
      const float* rp = s + (c * SR + qy + j) * SW + qx;
#pragma unroll
      for (int k = 0; k < RV; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(rp + 4 * k);
        row[4 * k] = v.x; row[4 * k + 1] = v.y; row[4 * k + 2] = v.z; row[4 * k + 3] = v.w;
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        acc[c][0] = fmaf(g[d][i].x, row[i], acc[c][0]);
        acc[c][1] = fmaf(g[d][i].y, row[i + 1], acc[c][1]);
        acc[c][2] = fmaf(g[d][i].z, row[i + 2], acc[c][2]);
        acc[c][3] = fmaf(g[d][i].w, row[i + 3], acc[c][3]);
      }
    }
  }

  if (NWV > 1) {
    __syncthreads();  // everyone is done with the window
    if (wv > 0) {
      float* rp = s + (size_t)(wv - 1) * CBW * 256 + lane * 4;
#pragma unroll
      for (int c = 0; c < CBW; ++c)
        *reinterpret_cast<float4*>(rp + c * 256) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int w = 1; w < NWV; ++w) {
      const float* rp = s + (size_t)(w - 1) * CBW * 256 + lane * 4;
#pragma unroll
      for (int c = 0; c < CBW; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(rp + c * 256);
        acc[c][0] += v.x; acc[c][1] += v.y; acc[c][2] += v.z; acc[c][3] += v.w;
      }
    }
  }
  if (pmask == 0) return;
  const float fC = (float)C;
  float* op = grad + ((size_t)b * C + c0) * HW + (size_t)y * W + x;
#pragma unroll
  for (int c = 0; c < CBW; ++c)
    if (c0 + c < C)
      store4_masked(op + (size_t)c * HW, make_float4(acc[c][0] / fC, acc[c][1] / fC, acc[c][2] / fC, acc[c][3] / fC), pmask);
#endif
}

// ---- coarse pyramid levels: direct kernels (h*w <= 512) -----------------------------------------------
// out[b, d, p] = (1/C) sum_c f1n[b,c,p] * f2n[b,c,p+d]: item = (b, d, p), CS channel slices per item laid
// out CS-strided inside the wave (lane = slice * (64/CS) + item-in-wave), reduced with shuffles.
template <int MD, int CS>
__global__ __launch_bounds__(256) void corr2d_small_fwd_kernel(C2Set a, int B, int C, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  constexpr int IPW = 64 / CS;  // items per wave
  const int set = blockIdx.y;
  const float* __restrict__ f1 = a.f1[set];
  const float* __restrict__ f2 = a.f2[set];
  const float* __restrict__ st1 = a.st1[set];
  const float* __restrict__ st2 = a.st2[set];
  float* __restrict__ out = a.out[set];
  const int HW = H * W;
  const long long items = (long long)B * ND * ND * HW;
  const int lane = threadIdx.x & 63;
  const int cs = lane / IPW, sub = lane - cs * IPW;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long n = wave * IPW + sub;
  const bool live = n < items;
  const long long nn = live ? n : 0;
  const int p = (int)(nn % HW);
  const long long r = nn / HW;
  const int d = (int)(r % (ND * ND));
  const int b = (int)(r / (ND * ND));
  const int y = p / W, x = p - y * W;
  const int y2 = y + d / ND - MD, x2 = x + d % ND - MD;
  const bool inb = live && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
  const int cper = (C + CS - 1) / CS;
  const int cbeg = cs * cper, cend = min(C, cbeg + cper);
  float acc = 0.f;
  if (inb) {
    const float* p1 = f1 + ((size_t)b * C + cbeg) * HW + p;
    const float* p2 = f2 + ((size_t)b * C + cbeg) * HW + y2 * W + x2;
    if (st1 != nullptr) {
      const float* q1 = st1 + 2 * ((size_t)b * C + cbeg);
      const float* q2 = st2 + 2 * ((size_t)b * C + cbeg);
#pragma unroll 4
      for (int c = cbeg; c < cend; ++c, p1 += HW, p2 += HW, q1 += 2, q2 += 2) {
        const float2 s1v = *reinterpret_cast<const float2*>(q1), s2v = *reinterpret_cast<const float2*>(q2);
        acc = fmaf((p1[0] - s1v.x) * s1v.y, (p2[0] - s2v.x) * s2v.y, acc);
      }
    } else {
#pragma unroll 4
      for (int c = cbeg; c < cend; ++c, p1 += HW, p2 += HW) acc = fmaf(p1[0], p2[0], acc);
    }
  }
#pragma unroll
  for (int o = IPW; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (live && cs == 0) out[n] = acc / (float)C;  // torch.mean = sum / C
}

// grad_f1[b,c,p] = (1/C) sum_d g[b,d,p] f2n[b,c,p+d];  grad_f2[b,c,q] = (1/C) sum_d g[b,d,q-d] f1n[b,c,q-d].
// One thread per (b, c, pixel); blockIdx.z picks the gradient; gathers only, reproducible.
template <int MD>
__global__ __launch_bounds__(256) void corr2d_small_bwd_kernel(C2Set a, int B, int C, int H, int W) {
  constexpr int ND = 2 * MD + 1;
  const int set = blockIdx.y;
  const bool second = blockIdx.z == 1;
  float* __restrict__ grad = second ? a.g2[set] : a.g1[set];
  if (grad == nullptr) return;  // uniform per block
  const float* __restrict__ other = second ? a.f1[set] : a.f2[set];
  const float* __restrict__ ost = second ? a.st1[set] : a.st2[set];
  const float* __restrict__ gout = a.gout[set];
  const int HW = H * W;
  const long long total = (long long)B * C * HW;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int p = (int)(e % HW);
  const long long bc = e / HW;
  const int b = (int)(bc / C);
  const int y = p / W, x = p - y * W;
  const float* ob = other + (size_t)bc * HW;
  const float* gb = gout + (size_t)b * ND * ND * HW;
  float m = 0.f, rs = 1.f;
  if (ost != nullptr) { m = ost[2 * bc]; rs = ost[2 * bc + 1]; }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int yy = second ? y - (j - MD) : y + (j - MD);
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int xx = second ? x - (i - MD) : x + (i - MD);
      if (xx < 0 || xx >= W) continue;
      // first: g at the own pixel, other at p + d;  second: both at q - d (the pixel that saw q under d)
      const float g = gb[(size_t)(j * ND + i) * HW + (second ? yy * W + xx : p)];
      float v = ob[yy * W + xx];
      if (ost != nullptr) v = (v - m) * rs;
      acc = fmaf(g, v, acc);
    }
  }
  grad[e] = acc / (float)C;
}

constexpr int kSmallHW = 512;  // direct kernels up to this many pixels per sample

template <int MD>
int launch_small_fwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  constexpr int ND = 2 * MD + 1;
  const long long items = (long long)B * ND * ND * H * W;
  // channel slices: enough threads for ~2 waves of workgroups per CU, at least 12 channels per slice
  int cs = 1;
  while (cs < 8 && items * cs < 256ll * 256 * 2 && C / (2 * cs) >= 12) cs *= 2;
  const long long waves = (items + (64 / cs) - 1) / (64 / cs);
  const dim3 grid((unsigned)((waves + 3) / 4), a.nsets);
  switch (cs) {
    case 1: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 1>), grid, dim3(256), 0, st, a, B, C, H, W); break;
    case 2: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 2>), grid, dim3(256), 0, st, a, B, C, H, W); break;
    case 4: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 4>), grid, dim3(256), 0, st, a, B, C, H, W); break;
    default: hipLaunchKernelGGL((corr2d_small_fwd_kernel<MD, 8>), grid, dim3(256), 0, st, a, B, C, H, W); break;
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_small_bwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  const long long total = (long long)B * C * H * W;
  const dim3 grid((unsigned)((total + 255) / 256), a.nsets, 2);
  hipLaunchKernelGGL(corr2d_small_bwd_kernel<MD>, grid, dim3(256), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_fwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  if (H * W <= kSmallHW) return launch_small_fwd<MD>(a, B, C, H, W, st);
  const dim3 grid((unsigned)((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * B * a.nsets));
  constexpr int DPW = FS_C2_FWD_DPW;
  hipLaunchKernelGGL((corr2d_fwd_q_kernel<MD, DPW>), grid, dim3(64 * ((2 * MD + DPW) / DPW)), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_bwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  if (H * W <= kSmallHW) return launch_small_bwd<MD>(a, B, C, H, W, st);
  const dim3 grid((unsigned)((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * 2 * B * a.nsets * fs::cdiv(C, CBW)));
  constexpr int DPW = FS_C2_BWD_DPW;
  hipLaunchKernelGGL((corr2d_bwd_q_kernel<MD, DPW>), grid, dim3(64 * ((2 * MD + DPW) / DPW)), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int check_shape(int B, int C, int H, int W, int md) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return FS_ERR_SHAPE;
  if ((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * 4 * B * fs::cdiv(C, CBW) >= (1ll << 31)) return FS_ERR_SHAPE;  // 1-D grids
  if ((long long)C * H * W >= (1ll << 29) || 81ll * H * W >= (1ll << 29)) return FS_ERR_SHAPE;  // 32-bit byte offsets inside a sample
  if (md < 1 || md > 4) return FS_ERR_ARG;
  return FS_OK;
}

int run_fwd(const C2Set& a, int B, int C, int H, int W, int md, hipStream_t st) {
  switch (md) {
    case 1: return launch_fwd<1>(a, B, C, H, W, st);
    case 2: return launch_fwd<2>(a, B, C, H, W, st);
    case 3: return launch_fwd<3>(a, B, C, H, W, st);
    default: return launch_fwd<4>(a, B, C, H, W, st);
  }
}

int run_bwd(const C2Set& a, int B, int C, int H, int W, int md, hipStream_t st) {
  switch (md) {
    case 1: return launch_bwd<1>(a, B, C, H, W, st);
    case 2: return launch_bwd<2>(a, B, C, H, W, st);
    case 3: return launch_bwd<3>(a, B, C, H, W, st);
    default: return launch_bwd<4>(a, B, C, H, W, st);
  }
}

C2Set one_set(const float* f1, const float* f2, const float* st1, const float* st2, float* out, const float* gout,
              float* g1, float* g2) {
  C2Set a = {};
  a.f1[0] = f1; a.f2[0] = f2; a.st1[0] = st1; a.st2[0] = st2; a.out[0] = out; a.gout[0] = gout;
  a.g1[0] = g1; a.g2[0] = g2; a.nsets = 1;
  return a;
}

// ---- per-plane moments and the adjoint of (f - mean) * rstd (normalize_features, §8f.4) ----------
// One workgroup per (b, c) plane; UPFlow's feature planes have 24 .. 4294 elements.
__device__ __forceinline__ double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

struct P4 { const float* f[4]; const float* gn[4]; float* gf[4]; };

// blockIdx.y selects the tensor (up to 4 of one shape per launch: both operands of both directions);
// stats of tensor k live at stats + k * 2 * planes
__global__ __launch_bounds__(256) void plane_moments_kernel(P4 a, float* __restrict__ stats_all, int S) {
  __shared__ double red[256];
  const float* __restrict__ f = a.f[blockIdx.y];
  float* __restrict__ stats = stats_all + (size_t)blockIdx.y * 2 * gridDim.x;
  const float* p = f + (size_t)blockIdx.x * S;
  double s = 0.0;
  for (int i = threadIdx.x; i < S; i += 256) s += (double)p[i];
  const float mean = (float)(block_sum(s, red) / (double)S);
  double q = 0.0;
  for (int i = threadIdx.x; i < S; i += 256) { const float d = p[i] - mean; q += (double)(d * d); }
  const float var = (float)(block_sum(q, red) / (double)(S - 1));  // torch.var: unbiased
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = 1.0f / sqrtf(var + 1e-16f);  // upflow.py:127 std = sqrt(var + 1e-16)
  }
}

// n = (f - m) r,  m = mean f,  r = (var + eps)^-1/2,  var = sum (f - m)^2 / (S - 1):
//   df_i = r * ( dn_i - mean(dn) - n_i * sum_j(dn_j n_j) / (S - 1) )
__global__ __launch_bounds__(256) void plane_norm_bwd_kernel(P4 t4, const float* __restrict__ stats_all, int S) {
  __shared__ double red[256];
  const float* __restrict__ f = t4.f[blockIdx.y];
  const float* __restrict__ gn = t4.gn[blockIdx.y];
  float* __restrict__ gf = t4.gf[blockIdx.y];
  if (gf == nullptr) return;  // uniform per block
  const float* __restrict__ stats = stats_all + (size_t)blockIdx.y * 2 * gridDim.x;
  const size_t base = (size_t)blockIdx.x * S;
  const float m = stats[2 * blockIdx.x], r = stats[2 * blockIdx.x + 1];
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < S; i += 256) {
    const float g = gn[base + i], n = (f[base + i] - m) * r;
    a += (double)g;
    c += (double)(g * n);
  }
  const float s1 = (float)(block_sum(a, red) / (double)S);
  const float s2 = (float)(block_sum(c, red) / (double)(S - 1));
  for (int i = threadIdx.x; i < S; i += 256) {
    const float n = (f[base + i] - m) * r;
    gf[base + i] = r * (gn[base + i] - s1 - n * s2);
  }
}

}  // namespace

extern "C" int fs_corr2d_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W,
                             int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(out);
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_fwd(one_set(f1, f2, nullptr, nullptr, out, nullptr, nullptr, nullptr), B, C, H, W, max_displacement,
                 (hipStream_t)stream);
}

extern "C" int fs_corr2d_bwd(const float* f1, const float* f2, const float* grad_out, float* grad_f1,
                             float* grad_f2, int B, int C, int H, int W, int max_displacement,
                             fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(grad_out);
  if (grad_f1 == nullptr && grad_f2 == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_bwd(one_set(f1, f2, nullptr, nullptr, nullptr, grad_out, grad_f1, grad_f2), B, C, H, W,
                 max_displacement, (hipStream_t)stream);
}

extern "C" int fs_plane_moments(const float* f, float* stats, int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f); FS_REQUIRE_PTR(stats);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;  // unbiased variance needs two samples
  P4 a = {};
  a.f[0] = f;
  hipLaunchKernelGGL(plane_moments_kernel, dim3(planes, 1), dim3(256), 0, (hipStream_t)stream, a, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_plane_norm_bwd(const float* f, const float* stats, const float* grad_n, float* grad_f,
                                 int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f); FS_REQUIRE_PTR(stats); FS_REQUIRE_PTR(grad_n); FS_REQUIRE_PTR(grad_f);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;
  P4 a = {};
  a.f[0] = f; a.gn[0] = grad_n; a.gf[0] = grad_f;
  hipLaunchKernelGGL(plane_norm_bwd_kernel, dim3(planes, 1), dim3(256), 0, (hipStream_t)stream, a, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" int fs_corr2d_norm_fwd(const float* f1, const float* f2, const float* stats1, const float* stats2,
                                  float* out, int B, int C, int H, int W, int max_displacement,
                                  fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(stats1); FS_REQUIRE_PTR(stats2); FS_REQUIRE_PTR(out);
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_fwd(one_set(f1, f2, stats1, stats2, out, nullptr, nullptr, nullptr), B, C, H, W, max_displacement,
                 (hipStream_t)stream);
}

extern "C" int fs_corr2d_norm_bwd(const float* f1, const float* f2, const float* stats1, const float* stats2,
                                  const float* grad_out, float* grad_n1, float* grad_n2, int B, int C, int H,
                                  int W, int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(stats1); FS_REQUIRE_PTR(stats2);
  FS_REQUIRE_PTR(grad_out);
  if (grad_n1 == nullptr && grad_n2 == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return run_bwd(one_set(f1, f2, stats1, stats2, nullptr, grad_out, grad_n1, grad_n2), B, C, H, W,
                 max_displacement, (hipStream_t)stream);
}

// ---- both directions of one pyramid level per launch (UPFlow/model/upflow.py:649 and :652) -----------
// Set a = (f1a, f2a) -> outa, set b = (f1b, f2b) -> outb, identical shapes.  `stats` (nullable) = the four
// (mean, rstd) tables [4][B*C][2] of (f1a, f2a, f1b, f2b) as fs_plane_moments4 writes them: the
// normalize_features-folded variant.  The five levels of a step cannot share a launch: level l's features are
// warped with the flow estimated at level l-1 (upflow.py:621-633).
extern "C" int fs_corr2d_pair_fwd(const float* f1a, const float* f2a, const float* f1b, const float* f2b,
                                  const float* stats, float* outa, float* outb, int B, int C, int H, int W,
                                  int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1a); FS_REQUIRE_PTR(f2a); FS_REQUIRE_PTR(f1b); FS_REQUIRE_PTR(f2b);
  FS_REQUIRE_PTR(outa); FS_REQUIRE_PTR(outb);
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  const size_t n = (size_t)2 * B * C;
  C2Set a = {};
  a.nsets = 2;
  a.f1[0] = f1a; a.f2[0] = f2a; a.out[0] = outa;
  a.f1[1] = f1b; a.f2[1] = f2b; a.out[1] = outb;
  if (stats != nullptr) { a.st1[0] = stats; a.st2[0] = stats + n; a.st1[1] = stats + 2 * n; a.st2[1] = stats + 3 * n; }
  return run_fwd(a, B, C, H, W, max_displacement, (hipStream_t)stream);
}

// Gradients of both sets (each pointer nullable, at least one non-null); with `stats` they are the gradients
// w.r.t. the NORMALISED maps (chain them with fs_plane_norm_bwd4).
extern "C" int fs_corr2d_pair_bwd(const float* f1a, const float* f2a, const float* f1b, const float* f2b,
                                  const float* stats, const float* gouta, const float* goutb, float* g1a,
                                  float* g2a, float* g1b, float* g2b, int B, int C, int H, int W,
                                  int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1a); FS_REQUIRE_PTR(f2a); FS_REQUIRE_PTR(f1b); FS_REQUIRE_PTR(f2b);
  FS_REQUIRE_PTR(gouta); FS_REQUIRE_PTR(goutb);
  if (g1a == nullptr && g2a == nullptr && g1b == nullptr && g2b == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  const size_t n = (size_t)2 * B * C;
  C2Set a = {};
  a.nsets = 2;
  a.f1[0] = f1a; a.f2[0] = f2a; a.gout[0] = gouta; a.g1[0] = g1a; a.g2[0] = g2a;
  a.f1[1] = f1b; a.f2[1] = f2b; a.gout[1] = goutb; a.g1[1] = g1b; a.g2[1] = g2b;
  if (stats != nullptr) { a.st1[0] = stats; a.st2[0] = stats + n; a.st1[1] = stats + 2 * n; a.st2[1] = stats + 3 * n; }
  return run_bwd(a, B, C, H, W, max_displacement, (hipStream_t)stream);
}

// (mean, rstd) of every (b, c) plane of four tensors of one shape in one launch: stats [4][planes][2]
extern "C" int fs_plane_moments4(const float* fa, const float* fb, const float* fc, const float* fd, float* stats,
                                 int planes, int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(fa); FS_REQUIRE_PTR(fb); FS_REQUIRE_PTR(fc); FS_REQUIRE_PTR(fd); FS_REQUIRE_PTR(stats);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;
  P4 a = {};
  a.f[0] = fa; a.f[1] = fb; a.f[2] = fc; a.f[3] = fd;
  hipLaunchKernelGGL(plane_moments_kernel, dim3(planes, 4), dim3(256), 0, (hipStream_t)stream, a, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// grad_f[k] = adjoint of (f - mean) * rstd applied to grad_n[k], k = 0..3 (a null grad_f[k] skips tensor k)
extern "C" int fs_plane_norm_bwd4(const float* fa, const float* fb, const float* fc, const float* fd,
                                  const float* stats, const float* gna, const float* gnb, const float* gnc,
                                  const float* gnd, float* gfa, float* gfb, float* gfc, float* gfd, int planes,
                                  int S, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(stats);
  if (planes < 1 || S < 2) return FS_ERR_SHAPE;
  P4 a = {};
  const float* f[4] = {fa, fb, fc, fd};
  const float* gn[4] = {gna, gnb, gnc, gnd};
  float* gf[4] = {gfa, gfb, gfc, gfd};
  for (int k = 0; k < 4; ++k) {
    if (gf[k] != nullptr && (f[k] == nullptr || gn[k] == nullptr)) return FS_ERR_NULLPTR;
    a.f[k] = f[k]; a.gn[k] = gn[k]; a.gf[k] = gf[k];
  }
  hipLaunchKernelGGL(plane_norm_bwd_kernel, dim3(planes, 4), dim3(256), 0, (hipStream_t)stream, a, stats, S);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
