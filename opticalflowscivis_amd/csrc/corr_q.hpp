// corr_q.hpp -- the tiled local-window correlation kernels of csrc/corr2d.hip (2-D, UPFlow cost volume) and
// csrc/corr3d.hip (3-D), one implementation: a 3-D problem is (2md+1) 2-D problems per slice (f1 slice z against f2
// slice z + dz), so the kernels take the planes-per-sample count D (1 = 2-D) and the channel stride D*H*W.
// Included inside each file's anonymous namespace.
#pragma once

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
  int D;   // planes per sample (1 for the 2-D layer)
  int NZ;  // displacement planes: 1 = the 2-D layer;  2md+1 = the 3-D layer (csrc/corr3d.hip)
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

  // cs = channel stride (floats); `pok` false: the whole plane lies outside the volume (reads as 0)
  __device__ __forceinline__ void init(int t, int y0, int x0, int H, int W, int cs, bool pok) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = t + NT * k;
      const int c = i / (SR * SV), rem = i - c * (SR * SV);
      const int r = rem / SV, v = rem - r * SV;
      const int gy = y0 + r - MD, gx = x0 + 4 * v - MD;
      off[k] = c * cs + gy * W + gx;
      lo[k] = (c * SR + r) * SW + 4 * v;
      cm[k] = (c << 4) | (i < N ? row_mask(pok && gy >= 0 && gy < H, gx, W) : 0);
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
__global__ __launch_bounds__(64 * ((2 * MD + DPW) / DPW)) void corr_fwd_q_kernel(C2Set a, int B, int C, int H, int W) {
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
  // 3-D: one workgroup per (slice z, displacement plane dz) = the 2-D problem (f1 slice z, f2 slice z + dz)
  const int D = a.D, NZ = a.NZ;
  const int dzi = (int)(tile % (unsigned)NZ); tile /= (unsigned)NZ;
  const int z = (int)(tile % (unsigned)D); tile /= (unsigned)D;
  const int z2 = NZ > 1 ? z + dzi - MD : 0;
  const bool zok = z2 >= 0 && z2 < D;
  const int b = (int)(tile % (unsigned)B), set = (int)(tile / (unsigned)B);
  const float* __restrict__ st1 = a.st1[set];
  const float* __restrict__ st2 = a.st2[set];
  float* __restrict__ out = a.out[set];
  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  int qy, qx;
  lane_quad(lane, qy, qx);
  const int HW = H * W, CS = D * HW;
  const rsrc_t r1 = make_rsrc(a.f1[set], (long long)B * C * CS);
  const rsrc_t r2 = make_rsrc(a.f2[set], (long long)B * C * CS);
  const int boff1 = b * C * CS + z * HW, boff2 = b * C * CS + (zok ? z2 : 0) * HW;
  if (st1 != nullptr) { st1 += 2 * (size_t)b * C; st2 += 2 * (size_t)b * C; }

  W2 w2;
  w2.init(t, y0, x0, H, W, CS, zok);
  int off1[K1], lo1[K1], cm1[K1];
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    const int i = t + NT * k;
    const int c = i / (TY * (TX / 4)), rem = i - c * (TY * (TX / 4));
    const int r = rem / (TX / 4), v = rem - r * (TX / 4);
    const int gy = y0 + r, gx = x0 + 4 * v;
    off1[k] = c * CS + gy * W + gx;
    lo1[k] = W2::FLOATS + (c * TY + r) * P1 + 4 * v;
    cm1[k] = (c << 4) | (i < N1 ? row_mask(gy < H, gx, W) : 0);
  }
  float4 pf2[W2::K], pf1[K1];
  auto fetch = [&](int c0) {
    w2.load(pf2, r2, boff2 + c0 * CS);
#pragma unroll
    for (int k = 0; k < K1; ++k) pf1[k] = bload4(r1, boff1 + c0 * CS + off1[k]);  // (never negative)
  };
  auto put = [&](float* s, int c0) {
    const int nch = C - c0;
    w2.put(pf2, s, t, st2 != nullptr ? st2 + 2 * c0 : nullptr, nch, boff2 + c0 * CS);
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
    float* ob = out + (((size_t)b * NZ * ND * ND + (size_t)(dzi * ND + dy) * ND) * D + z) * HW + (size_t)y * W + x;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // (i, j) lives in a pair or in the single of its parity class
        const int jj = j - (i & 1);
        v[i] = (i & 1) ? (j == 0 ? accs[d][i] : ((jj & 1) ? accp[d][i][jj / 2].y : accp[d][i][jj / 2].x))
                       : (j == ND - 1 ? accs[d][i] : ((j & 1) ? accp[d][i][j / 2].y : accp[d][i][j / 2].x));
      }
      store4_masked(ob + (size_t)j * CS, mean4(make_float4(v[0], v[1], v[2], v[3]), fC, rC, pow2), smask);
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
__global__ __launch_bounds__(512) void corr_bwd_q_kernel(C2Set a, int B, int C, int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ND = 2 * MD + 1;
  constexpr int SR = TY + 2 * MD, SV = (TX + 2 * MD + 3) / 4;
  constexpr int RV = (4 + 2 * MD + 3) / 4;    // float4 per channel of a row segment
  // window row pitch: 2 channels x 40 px, + 8 spare.  A lane reads 16 bytes at float offset 8 quad + 4 k of its row (the
  // pair interleave), so the 8 quads of a row cover every other group of 4 banks; the ds_read_b128 lane groups hold
  // rows r and r + 4 (lane_quad), which must fill the gaps: rows with bit 2 set are stored 4 floats to the right
  // (4 * 88 + 4 = 356 = 36 mod 64 banks).  Without the skew 34 % of the kernel's LDS cycles were bank conflicts
  // (profiles/r02_corr_pmc_counters.txt).
  constexpr int WP = 88;
  constexpr int WPAIR = SR * WP;
  constexpr int WFLOATS = (CBW / 2) * WPAIR;
  constexpr int GP = 40;                       // row pitch of a staged gradient plane (lane_quad)
  constexpr int GBUF = ND * TY * GP;           // one displacement row: ND planes x 8 rows
  constexpr int KW = CBW / 4, KG = (ND + 7) / 8;
  static_assert(SR <= 16 && SV <= 16 && 8 * SV + 4 <= WP, "staging maps");
  __shared__ __attribute__((aligned(16))) float s[WFLOATS + 2 * GBUF];

  const int CG = (C + CBW - 1) / CBW;
  const int ntx = (W + TX - 1) / TX, nty = (H + TY - 1) / TY;
  unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const int x0 = (int)(tile % (unsigned)ntx) * TX; tile /= (unsigned)ntx;
  const int y0 = (int)(tile % (unsigned)nty) * TY; tile /= (unsigned)nty;
  const int cg = (int)(tile % (unsigned)CG); tile /= (unsigned)CG;
  const int D = a.D, NZ = a.NZ;
  const int z = (int)(tile % (unsigned)D); tile /= (unsigned)D;
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
  const int HW = H * W, CS = D * HW;
  const int c0 = cg * CBW, nch = C - c0;

  // window items: thread = (pair parity t >> 8, row (t >> 4) & 15, float4 column t & 15), item k = pair (t >> 8) + 2 k
  const int wvx = t & 15, wr = (t >> 4) & 15, wp0 = t >> 8;
  const int wgy = y0 + wr - MD, wgx = x0 + 4 * wvx - MD;
  const bool wlive = wr < SR && wvx < SV;
  const int wmask = wlive ? row_mask(wgy >= 0 && wgy < H, wgx, W) : 0;
  const rsrc_t rw = make_rsrc(other, (long long)B * C * CS);

  // gradient items of displacement row j: thread = (dx index i = (t >> 6) + 8 k, tile row (t >> 3) & 7, float4 t & 7)
  //   first : g[(j, i)] at (y0 + r, x0 + 4 v)
  //   second: gT[(j, i)] = g[(ND-1-j, ND-1-i)] at (y0 + r + j - MD, x0 + 4 v + i - MD)
  const int gv = t & 7, gr = (t >> 3) & 7, gi0 = t >> 6;
  const int ggy = y0 + gr, ggx = x0 + 4 * gv;
  const int own = row_mask(ggy < H, ggx, W);
  // (a gradient vector with live elements never starts before its plane: no negative offsets here)
  const rsrc_t rg = make_rsrc(a.gout[set] + (size_t)b * NZ * ND * ND * CS, (long long)(B - b) * NZ * ND * ND * CS);

  v2f acc[2][4];  // [pair of this wave][pixel] = (channel 4 wv + 2 pair, + 1)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = v2f{0.f, 0.f};

  // 3-D: the displacement planes dz one after the other, each the 2-D problem against slice z + dz of `other`
  bool fresh = true;
  for (int dzi = 0; dzi < NZ; ++dzi) {
  const int z2 = NZ > 1 ? z + dzi - MD : 0;
  if (z2 < 0 || z2 >= D) continue;  // uniform: this plane reads zeros
  if (!fresh) __syncthreads();      // everyone is done with the previous plane's window and gradient rows
  fresh = false;
  const int woff = (b * C + c0) * CS + z2 * HW + wgy * W + wgx;
  float4 pa[KW], pb[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const int cp = wp0 + 2 * k;
    pa[k] = bload4(rw, (2 * cp) * CS + woff);
    pb[k] = bload4(rw, (2 * cp + 1) * CS + woff);
  }
  float4 pg[ND][KG];
  auto gfetch = [&](int j) {
    // plane (dz, j, i) at slice z;  transposed: plane (-dz, -j, -i) at slice z + dz, row + (j - md), column + (i - md)
    const int step = second ? (((NZ - 1 - dzi) * ND + (ND - 1 - j)) * ND) * CS + z2 * HW + (j - MD) * W
                            : ((dzi * ND + j) * ND) * CS + z * HW;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int i = gi0 + 8 * k;
      const int off = second ? (ND - 1 - i) * CS + ggy * W + ggx + (i - MD) : i * CS + ggy * W + ggx;
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
      float4 va = fix_head(pa[k], (2 * cp) * CS + woff), vb = fix_head(pb[k], (2 * cp + 1) * CS + woff);
      if (ost != nullptr) {
        const float* q = ost + 2 * ((size_t)b * C + c0);
        const float2 qa = *reinterpret_cast<const float2*>(q + 2 * (2 * cp < nch ? 2 * cp : 0));
        const float2 qb = *reinterpret_cast<const float2*>(q + 2 * (2 * cp + 1 < nch ? 2 * cp + 1 : 0));
        va = make_float4((va.x - qa.x) * qa.y, (va.y - qa.x) * qa.y, (va.z - qa.x) * qa.y, (va.w - qa.x) * qa.y);
        vb = make_float4((vb.x - qb.x) * qb.y, (vb.y - qb.x) * qb.y, (vb.z - qb.x) * qb.y, (vb.w - qb.x) * qb.y);
      }
      va = keep4(va, 2 * cp < nch ? wmask : 0);
      vb = keep4(vb, 2 * cp + 1 < nch ? wmask : 0);
      float* wp = s + cp * WPAIR + wr * WP + (wr & 4) + 8 * wvx;
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

#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const float* gs = s + WFLOATS + (j & 1) * GBUF + qy * GP + qx;
    float4 g[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) g[i] = *reinterpret_cast<const float4*>(gs + i * TY * GP);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      v2f r2[4 * RV];  // r2[m] = (c0, c1) at pixel qx + m of window row qy + j
      const float* rp = s + (wv * 2 + c) * WPAIR + (qy + j) * WP + ((qy + j) & 4) + 2 * qx;
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
  }  // dz

  const int y = y0 + qy, x = x0 + qx;
  const int pmask = row_mask(y < H, x, W);
  if (pmask == 0) return;
  const float fC = (float)C, rC = 1.0f / fC;
  const bool pow2 = (C & (C - 1)) == 0;
  const int cw = c0 + wv * 4;
  float* op = grad + (((size_t)b * C + cw) * D + z) * HW + (size_t)y * W + x;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (cw + 2 * c < C)
      store4_masked(op + (size_t)(2 * c) * CS,
                    mean4(make_float4(acc[c][0].x, acc[c][1].x, acc[c][2].x, acc[c][3].x), fC, rC, pow2), pmask);
    if (cw + 2 * c + 1 < C)
      store4_masked(op + (size_t)(2 * c + 1) * CS,
                    mean4(make_float4(acc[c][0].y, acc[c][1].y, acc[c][2].y, acc[c][3].y), fC, rC, pow2), pmask);
  }
#endif
}

