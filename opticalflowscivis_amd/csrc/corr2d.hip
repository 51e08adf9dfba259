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
// Tiled kernels (corr2d_fwd_q_kernel / corr2d_bwd_q_kernel, described at their definitions): one workgroup = one
// 8 x 32-pixel tile, lane = a quad of 4 consecutive pixels, the search window (tile + md halo) staged through LDS
// by unconditional 16-byte buffer loads at dword alignment, packed-FP32 FMAs on aligned register pairs, 16-byte
// stores, tiles dealt to the XCDs in contiguous ranges.  Both are bound by VALU instruction issue; HBM traffic is
// the algorithmic 4*(2C + (2md+1)^2) B/pixel forward, 4*(4C + (2md+1)^2) backward; halo re-reads hit L2.
// They need pixels: the coarsest UPFlow levels at C3 are (C, h, w) = (196, 3, 8) and (128, 5, 15), 24 / 75 pixels
// per sample, C >> pixels, where an 8 x 32 tile is mostly empty and the forward kernel is a serial chain of C/8
// stage-barrier-compute rounds.  Those run the direct kernels instead (thresholds: kSmall* below): no LDS, no
// barrier, one thread per (sample, displacement, pixel) output and channel slice -- the C range is split over
// 1..8 lane groups of a wave and reduced with wave shuffles -- so that even a B = 2 launch fills the chip.
//
// Backward.  grad_f1[c,p] = (1/C) sum_d g[d,p] f2[c,p+d] is a gather;  grad_f2 is the same gather
// with the roles swapped and the displacement negated:  grad_f2[c,q] = (1/C) sum_d gT[d,q] f1[c,q+d]
// with gT[d,q] = g[-d, q+d] (0 when q+d is outside).  One launch computes both, no atomics, bitwise reproducible.
#include "common.hpp"

namespace {

constexpr int TY = 8, TX = 32;
[[maybe_unused]] constexpr int CC = 8;  // channels per staged chunk of the forward kernel
// displacement rows per wave of the tiled kernels (forward: 1 -> 2md+1 waves, 36 accumulators per lane at md = 4:
// 1 / 2 / 3 rows per wave measured 24 / 27 / 37 us at the (64, 19, 57) level, equal at (32, 38, 113))
#define FS_C2_FWD_DPW 1

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
typedef float v2f __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ unsigned xcd_tile(unsigned id, unsigned total) {
  const unsigned per = total >> 3;
  return id < per * 8 ? (id & 7) * per + (id >> 3) : id;
}

// sum / C as torch.mean computes it: a multiplication by 1/C is the same float whenever C is a power of two
// (every tiled UPFlow level), and ~10 instructions cheaper per value
__device__ __forceinline__ float4 mean4(float4 v, float fC, float rC, bool pow2) {
  return pow2 ? make_float4(v.x * rC, v.y * rC, v.z * rC, v.w * rC)
              : make_float4(v.x / fC, v.y / fC, v.z / fC, v.w / fC);
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
// afterwards.  A vector that runs past the tensor's last float reads 0 there (the range check is per dword).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* base, long long floats) {
  const long long bytes = floats * 4;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes,
                                           0x00020000);
}
// (the descriptor starts at the tensor's first float, so only the one vector that straddles it has a negative
// offset with live elements: it is loaded from offset 0 and shifted into place by fix_head -- a negative offset is
// out of range as a whole, not per dword)
__device__ __forceinline__ float4 bload4(rsrc_t r, int off_floats) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (off_floats < 0 ? 0 : off_floats) * 4, 0, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ float4 fix_head(float4 v, int off_floats) {
  if (off_floats >= 0 || off_floats <= -4) return v;
  if (off_floats == -1) return make_float4(0.f, v.x, v.y, v.z);
  if (off_floats == -2) return make_float4(0.f, 0.f, v.x, v.y);
  return make_float4(0.f, 0.f, 0.f, v.x);
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
                                      const float* __restrict__ st, int nch, int coff) const {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (t + NT * k >= N) continue;
      const int c = cm[k] >> 4;
      float4 v = fix_head(pf[k], coff + off[k]);
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
  unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int x0 = (int)(tile % (unsigned)ntx) * TX; tile /= (unsigned)ntx;
  const int y0 = (int)(tile % (unsigned)nty) * TY; tile /= (unsigned)nty;
  const int b = (int)(tile % (unsigned)B), set = (int)(tile / (unsigned)B);
  const float* __restrict__ st1 = a.st1[set];
  const float* __restrict__ st2 = a.st2[set];
  float* __restrict__ out = a.out[set];
  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  int qy, qx;
  lane_quad(lane, qy, qx);
  const int HW = H * W;
  const rsrc_t r1 = make_rsrc(a.f1[set], (long long)B * C * HW);
  const rsrc_t r2 = make_rsrc(a.f2[set], (long long)B * C * HW);
  const int boff = b * C * HW;
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
    w2.load(pf2, r2, boff + c0 * HW);
#pragma unroll
    for (int k = 0; k < K1; ++k) pf1[k] = bload4(r1, boff + c0 * HW + off1[k]);  // (never negative)
  };
  auto put = [&](float* s, int c0) {
    const int nch = C - c0;
    w2.put(pf2, s, t, st2 != nullptr ? st2 + 2 * c0 : nullptr, nch, boff + c0 * HW);
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

  // accumulators as aligned register pairs for v_pk_fma_f32: out(px i, dx j) += f1[i] * row[i + j], paired over j so
  // that the row operand (row[i + j], row[i + j + 1]) starts at an even register: i even -> pairs j = (0,1), (2,3),
  // .. and a single j = ND - 1;  i odd -> a single j = 0 and pairs j = (1,2), (3,4), ..
  constexpr int NP = (ND - 1) / 2;
  v2f accp[DPW][4][NP];
  float accs[DPW][4];
#pragma unroll
  for (int d = 0; d < DPW; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      accs[d][i] = 0.f;
#pragma unroll
      for (int q = 0; q < NP; ++q) accp[d][i][q] = v2f{0.f, 0.f};
    }

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
        v2f r2[2 * RV];
        const float* rp = s2 + (c * SR + qy + dy) * SW + qx;
#pragma unroll
        for (int k = 0; k < RV; ++k) {
          const float4 v = *reinterpret_cast<const float4*>(rp + 4 * k);
          r2[2 * k] = v2f{v.x, v.y}; r2[2 * k + 1] = v2f{v.z, v.w};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const v2f a2 = v2f{av[i], av[i]};
#pragma unroll
          for (int q = 0; q < NP; ++q)
            accp[d][i][q] = __builtin_elementwise_fma(a2, r2[(i + (i & 1)) / 2 + q], accp[d][i][q]);
          const float rs = (i & 1) ? r2[(i - 1) / 2].y : r2[(i + ND - 1) / 2].x;
          accs[d][i] = fmaf(av[i], rs, accs[d][i]);
        }
      }
    }
    if (more) put(lds + (buf ^ 1) * BUF, c0 + CC);
    __syncthreads();
    buf ^= 1;
  }

  const int y = y0 + qy, x = x0 + qx;
  const int smask = row_mask(y < H, x, W);
  if (smask == 0) return;
  const float fC = (float)C, rC = 1.0f / fC;
  const bool pow2 = (C & (C - 1)) == 0;
#pragma unroll
  for (int d = 0; d < DPW; ++d) {
    const int dy = DPW * wv + d;
    if (dy >= ND) continue;
    float* ob = out + ((size_t)b * ND * ND + (size_t)dy * ND) * HW + (size_t)y * W + x;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // (i, j) lives in a pair or in the single of its parity class
        const int jj = j - (i & 1);
        v[i] = (i & 1) ? (j == 0 ? accs[d][i] : ((jj & 1) ? accp[d][i][jj / 2].y : accp[d][i][jj / 2].x))
                       : (j == ND - 1 ? accs[d][i] : ((j & 1) ? accp[d][i][j / 2].y : accp[d][i][j / 2].x));
      }
      store4_masked(ob + (size_t)j * HW, mean4(make_float4(v[0], v[1], v[2], v[3]), fC, rC, pow2), smask);
    }
  }
#endif
}

// Backward.  grad[c,p] = (1/C) sum_d g(d,p) * other[c, p+d];  first: (g = gout, other = f2) -> grad_f1;  second:
// (g = gout transposed on the fly, gT[d,q] = g[-d, q+d], other = f1) -> grad_f2.  The forward kernel with the roles
// of channel and displacement exchanged: the reduction runs over d, the outputs are indexed by c.  One workgroup =
// one 8 x 32 tile x 32 channels of one sample (all channels of the finest UPFlow level: the gradient tile is read
// once per gradient), 8 waves x 2 channel PAIRS.  These kernels are bound by VALU instruction issue, not by LDS or
// HBM (measured: 3500 instructions per wave around 648 useful v_pk_fma_f32), so the layout serves the packed FMA:
// the `other` window sits in LDS with the two channels of a pair interleaved per pixel, [pair][row][x][2]; a
// ds_read_b128 then returns aligned register pairs (c0, c1) and one v_pk_fma_f32 updates both channels of a pixel
// with the gradient value broadcast by op_sel -- no register moves.  The upstream gradient passes through two
// small LDS buffers one displacement row (2md+1 planes of the tile) at a time; every global load of the workgroup
// is issued up front and consumed in issue order.  Thread -> staging item maps are shifts and masks only.
// lane = a quad of 4 consecutive pixels.  No cross-wave reduction, no atomics, bitwise reproducible; 16-byte stores.
constexpr int CBW = 32;
template <int MD>
__global__ __launch_bounds__(512) void corr2d_bwd_q_kernel(C2Set a, int B, int C, int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD, SV = (TX + 2 * MD + 3) / 4;
  constexpr int RV = (4 + 2 * MD + 3) / 4;    // float4 per channel of a row segment
  constexpr int WP = 88;                       // window row pitch (2 channels x 40 px, + 8: rows r and r + 4 half a bank cycle apart)
  constexpr int WPAIR = SR * WP;
  constexpr int WFLOATS = (CBW / 2) * WPAIR;
  constexpr int GP = 40;                       // row pitch of a staged gradient plane (lane_quad)
  constexpr int GBUF = ND * TY * GP;           // one displacement row: ND planes x 8 rows
  constexpr int KW = CBW / 4, KG = (ND + 7) / 8;
  static_assert(SR <= 16 && SV <= 16 && 8 * SV <= WP, "staging maps");
  __shared__ __attribute__((aligned(16))) float s[WFLOATS + 2 * GBUF];

  const int CG = (C + CBW - 1) / CBW;
  const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
  unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int x0 = (int)(tile % (unsigned)ntx) * TX; tile /= (unsigned)ntx;
  const int y0 = (int)(tile % (unsigned)nty) * TY; tile /= (unsigned)nty;
  const int cg = (int)(tile % (unsigned)CG); tile /= (unsigned)CG;
  const int b = (int)(tile % (unsigned)B); tile /= (unsigned)B;
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
  const int c0 = cg * CBW, nch = C - c0;

  // window items: thread = (pair parity t >> 8, row (t >> 4) & 15, float4 column t & 15), item k = pair (t >> 8) + 2 k
  const int wvx = t & 15, wr = (t >> 4) & 15, wp0 = t >> 8;
  const int wgy = y0 + wr - MD, wgx = x0 + 4 * wvx - MD;
  const bool wlive = wr < SR && wvx < SV;
  const int wmask = wlive ? row_mask(wgy >= 0 && wgy < H, wgx, W) : 0;
  const int woff = (b * C + c0) * HW + wgy * W + wgx;
  const rsrc_t rw = make_rsrc(other, (long long)B * C * HW);
  float4 pa[KW], pb[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const int cp = wp0 + 2 * k;
    pa[k] = bload4(rw, (2 * cp) * HW + woff);
    pb[k] = bload4(rw, (2 * cp + 1) * HW + woff);
  }

  // gradient items of displacement row j: thread = (dx index i = (t >> 6) + 8 k, tile row (t >> 3) & 7, float4 t & 7)
  //   first : g[(j, i)] at (y0 + r, x0 + 4 v)
  //   second: gT[(j, i)] = g[(ND-1-j, ND-1-i)] at (y0 + r + j - MD, x0 + 4 v + i - MD)
  const int gv = t & 7, gr = (t >> 3) & 7, gi0 = t >> 6;
  const int ggy = y0 + gr, ggx = x0 + 4 * gv;
  const int own = row_mask(ggy < H, ggx, W);
  // (a gradient vector with live elements never starts before its plane: no negative offsets here)
  const rsrc_t rg = make_rsrc(a.gout[set] + (size_t)b * ND * ND * HW, (long long)(B - b) * ND * ND * HW);
  float4 pg[ND][KG];
  auto gfetch = [&](int j) {
    const int step = second ? (ND - 1 - j) * ND * HW + (j - MD) * W : j * ND * HW;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int i = gi0 + 8 * k;
      const int off = second ? (ND - 1 - i) * HW + ggy * W + ggx + (i - MD) : i * HW + ggy * W + ggx;
      pg[j][k] = bload4(rg, off + step);
    }
  };
  gfetch(0);
  auto gput = [&](float* gs, int j) {
    const int yy = ggy + (j - MD);
    const bool rok = !second || (yy >= 0 && yy < H);
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int i = gi0 + 8 * k;
      if (i >= ND) continue;
      const int m = !rok ? 0 : (second ? own & row_mask(true, ggx + i - MD, W) : own);
      *reinterpret_cast<float4*>(gs + (i * TY + gr) * GP + 4 * gv) = keep4(pg[j][k], m);
    }
  };

  // window: normalise, clear what lies outside the image / past the last channel, interleave the pair, store
  if (wlive) {
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int cp = wp0 + 2 * k;
      float4 va = fix_head(pa[k], (2 * cp) * HW + woff), vb = fix_head(pb[k], (2 * cp + 1) * HW + woff);
      if (ost != nullptr) {
        const float* q = ost + 2 * ((size_t)b * C + c0);
        const float2 qa = *reinterpret_cast<const float2*>(q + 2 * (2 * cp < nch ? 2 * cp : 0));
        const float2 qb = *reinterpret_cast<const float2*>(q + 2 * (2 * cp + 1 < nch ? 2 * cp + 1 : 0));
        va = make_float4((va.x - qa.x) * qa.y, (va.y - qa.x) * qa.y, (va.z - qa.x) * qa.y, (va.w - qa.x) * qa.y);
        vb = make_float4((vb.x - qb.x) * qb.y, (vb.y - qb.x) * qb.y, (vb.z - qb.x) * qb.y, (vb.w - qb.x) * qb.y);
      }
      va = keep4(va, 2 * cp < nch ? wmask : 0);
      vb = keep4(vb, 2 * cp + 1 < nch ? wmask : 0);
      float* wp = s + cp * WPAIR + wr * WP + 8 * wvx;
      *reinterpret_cast<float4*>(wp) = make_float4(va.x, vb.x, va.y, vb.y);
      *reinterpret_cast<float4*>(wp + 4) = make_float4(va.z, vb.z, va.w, vb.w);
    }
  }
  gput(s + WFLOATS, 0);
  // the remaining displacement rows: all in flight from here on, consumed in issue order (the window's registers
  // are free now)
#pragma unroll
  for (int j = 1; j < ND; ++j) gfetch(j);
  __syncthreads();

  v2f acc[2][4];  // [pair of this wave][pixel] = (channel 4 wv + 2 pair, + 1)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = v2f{0.f, 0.f};

#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const float* gs = s + WFLOATS + (j & 1) * GBUF + qy * GP + qx;
    float4 g[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) g[i] = *reinterpret_cast<const float4*>(gs + i * TY * GP);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      v2f r2[4 * RV];  // r2[m] = (c0, c1) at pixel qx + m of window row qy + j
      const float* rp = s + (wv * 2 + c) * WPAIR + (qy + j) * WP + 2 * qx;
#pragma unroll
      for (int k = 0; k < 2 * RV; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(rp + 4 * k);
        r2[2 * k] = v2f{v.x, v.y}; r2[2 * k + 1] = v2f{v.z, v.w};
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        acc[c][0] = __builtin_elementwise_fma(v2f{g[i].x, g[i].x}, r2[i], acc[c][0]);
        acc[c][1] = __builtin_elementwise_fma(v2f{g[i].y, g[i].y}, r2[i + 1], acc[c][1]);
        acc[c][2] = __builtin_elementwise_fma(v2f{g[i].z, g[i].z}, r2[i + 2], acc[c][2]);
        acc[c][3] = __builtin_elementwise_fma(v2f{g[i].w, g[i].w}, r2[i + 3], acc[c][3]);
      }
    }
    // pin this row's FMAs here: left alone, the compiler sinks the FMAs of all rows behind the last barrier and
    // parks every LDS read of the kernel in scratch on the way (2 KB per lane)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[c][i]));
    if (j + 1 < ND) {
      gput(s + WFLOATS + ((j + 1) & 1) * GBUF, j + 1);
      __syncthreads();
    }
  }

  const int y = y0 + qy, x = x0 + qx;
  const int pmask = row_mask(y < H, x, W);
  if (pmask == 0) return;
  const float fC = (float)C, rC = 1.0f / fC;
  const bool pow2 = (C & (C - 1)) == 0;
  const int cw = c0 + wv * 4;
  float* op = grad + ((size_t)b * C + cw) * HW + (size_t)y * W + x;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (cw + 2 * c < C)
      store4_masked(op + (size_t)(2 * c) * HW,
                    mean4(make_float4(acc[c][0].x, acc[c][1].x, acc[c][2].x, acc[c][3].x), fC, rC, pow2), pmask);
    if (cw + 2 * c + 1 < C)
      store4_masked(op + (size_t)(2 * c + 1) * HW,
                    mean4(make_float4(acc[c][0].y, acc[c][1].y, acc[c][2].y, acc[c][3].y), fC, rC, pow2), pmask);
  }
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

// direct kernels up to this many pixels per sample (C3 levels (196, 3, 8) / (128, 5, 15) / (96, 10, 29), both
// directions per launch, direct vs tiled: forward 6.7 vs - / 16 vs 31 / 50 vs 26 us, with the normalisation folded
// in 15 vs - / 68 vs 34 / 152 vs 28 us; backward 17 vs - / 42 vs 26 / 186 vs 38 us)
constexpr int kSmallFwd = 128, kSmallFwdNorm = 32, kSmallBwd = 32;

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
  if (H * W <= (a.st1[0] != nullptr ? kSmallFwdNorm : kSmallFwd)) return launch_small_fwd<MD>(a, B, C, H, W, st);
  const dim3 grid((unsigned)((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * B * a.nsets));
  constexpr int DPW = FS_C2_FWD_DPW;
  hipLaunchKernelGGL((corr2d_fwd_q_kernel<MD, DPW>), grid, dim3(64 * ((2 * MD + DPW) / DPW)), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <int MD>
int launch_bwd(const C2Set& a, int B, int C, int H, int W, hipStream_t st) {
  if (H * W <= kSmallBwd) return launch_small_bwd<MD>(a, B, C, H, W, st);
  const dim3 grid((unsigned)((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * 2 * B * a.nsets * fs::cdiv(C, CBW)));
  hipLaunchKernelGGL(corr2d_bwd_q_kernel<MD>, grid, dim3(512), 0, st, a, B, C, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int check_shape(int B, int C, int H, int W, int md) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return FS_ERR_SHAPE;
  if ((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * 4 * B * fs::cdiv(C, CBW) >= (1ll << 31)) return FS_ERR_SHAPE;  // 1-D grids
  if ((long long)B * C * H * W >= (1ll << 29) || 81ll * H * W >= (1ll << 29)) return FS_ERR_SHAPE;  // 32-bit byte offsets inside a tensor
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
