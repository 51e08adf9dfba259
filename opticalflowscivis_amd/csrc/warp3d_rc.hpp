// warp3d_rc.hpp -- round 5: the trilinear warp pair (forward AND backward) as one ring pipeline whose gather SOURCE lives in
// LDS.  Included by warp3d.hip inside its anonymous namespace (uses W3P, W3Fwd, W3Bwd, W3Add, Pair, t_slot, ...).
//
// Why (profiles/r04_w3_ablation.txt, profiles/r04_w3_pmc.txt): with HBM traffic already at the algorithmic minimum the family
// stopped at ~0.5 of the HBM roof because of the texture-address work of its gathers -- 128 unaligned 8-byte gather
// wave-instructions per 64 x 32 tile-slice (~26 TA cycles each, whether or not they hit), plus, in the backward kernels,
// barrier-separated load / compute / store phases.  Here a workgroup (one per CU) is 8 compute waves + NMW mover waves:
//   * movers bring every tile global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): the flow tile (3 planes; + the
//     grad_out tile in the backward kernel) two slices ahead into a ring of 3 stages, and the SOURCE ROWS of the next slice
//     into a rolling row cache: for cubic volumes out[d,h,w] = in[w+F2, d+F1, h+F0], so the 64 h x 32 w tile of slice d reads
//     input rows y = d + F1 (+1) of planes z = w0.. + F2 at columns x = h0.. + F0 -- a slab of NP planes x NX columns whose row
//     index advances by one per slice.  One cached row (all NP planes) is 11 DMA wave-instructions of 1 KiB; the cache keeps R
//     rows as a ring.  They also stream results out (16-byte stores) and, backward, add the up-to-three extra gradient
//     tensors of the flow on the way (their loads are issued one iteration before they are used).
//   * compute waves (lane = h, wave = 4 consecutive w) read flow / grad_out with one ds_read_b128 per plane, the 8 corners
//     with four ds_read2_b32 from the cache, and write their result tile back to LDS; ONE barrier per slice.
// The window is flow-dependent and a prediction: its (x, z) origin is fixed per workgroup from the first slice's flow
// range, its rows follow the range of y0 the compute waves measured one iteration earlier.  Correctness never depends on
// the prediction: every voxel tests its own corners against the published window and gathers from global memory
// (the round-1..4 path, same arithmetic) when they are outside -- white-noise flows run entirely on that path.  Values are
// bit-identical to the kernels this replaces (tests/test_gpu_warps.py, scripts/w3bench.py CRCs).
#pragma once

namespace rc {

constexpr int NP = 37;                  // cached planes (z window: 32 + 1 + a spread of ~3)
constexpr int NQ = 18;                  // float4 per cached plane row
constexpr int NX = 4 * NQ;              // cached columns (x window: 64 + 1 + spread / alignment slack)
constexpr int ROW_INSTR = 11;           // 64-slot DMA groups per cached row (NP * NQ = 666 of 704 slots used)
constexpr int RSLOTS = 64 * ROW_INSTR;  // float4 slots per cached row
constexpr unsigned PLB = NX * 4u;       // bytes per cached plane row
constexpr unsigned RSB = RSLOTS * 16u;  // bytes per cached row
constexpr int NFR = 3;                  // ring stages of flow (+ grad_out) tiles

struct Mail {
  int xb, zb;             // window origin (fixed per workgroup)
  int lo[2], nv[2], sl[2];  // per slice parity: first valid row, number of valid rows, ring slot of the first valid row
  int st[2][NCW][2];      // per slice parity and compute wave: min / max of y0 over the wave's voxels
};

template <bool BWD>
struct SampI {
  float ax, ay, az;  // fractional parts
  float mx, my, mz;  // border-clip gradient multipliers (backward)
  int x0, y0, z0;    // floor corner
  bool px, py, pz;   // the +1 corner exists (inside the volume)
};

// w3_sample's arithmetic (identical expressions), integer corner kept
template <bool BWD>
__device__ __forceinline__ SampI<BWD> sample_i(const W3P& p, float lin_h, float lin_d, float lin_w, float f0, float f1,
                                               float f2) {
  SampI<BWD> s;
  float ix, iy, iz;
  {
#pragma clang fp contract(off)
    ix = w3_unnorm(lin_h + f0 * p.rH, p.mW);
    iy = w3_unnorm(lin_d + f1 * p.rD, p.mH);
    iz = w3_unnorm(lin_w + f2 * p.rW, p.mD);
  }
  if (BWD) {
    s.mx = (ix > 0.0f && ix < p.mW) ? 1.0f : 0.0f;
    s.my = (iy > 0.0f && iy < p.mH) ? 1.0f : 0.0f;
    s.mz = (iz > 0.0f && iz < p.mD) ? 1.0f : 0.0f;
  }
  ix = __builtin_amdgcn_fmed3f(ix, 0.0f, p.mW);
  iy = __builtin_amdgcn_fmed3f(iy, 0.0f, p.mH);
  iz = __builtin_amdgcn_fmed3f(iz, 0.0f, p.mD);
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  s.ax = ix - fx; s.ay = iy - fy; s.az = iz - fz;
  s.x0 = (int)(unsigned)fx; s.y0 = (int)(unsigned)fy; s.z0 = (int)(unsigned)fz;
  s.px = fx < p.mW; s.py = fy < p.mH; s.pz = fz < p.mD;
  return s;
}

// floor corner of one coordinate alone (window prediction in the movers' prologue; monotone in f)
__device__ __forceinline__ int corner_of(float lin, float f, float r, float m) {
  float i;
  {
#pragma clang fp contract(off)
    i = w3_unnorm(lin + f * r, m);
  }
  i = __builtin_amdgcn_fmed3f(i, 0.0f, m);
  return (int)(unsigned)floorf(i);
}

// wave-wide min / max as a wave-uniform value: four rotations inside every row of 16 lanes (DPP row_ror 8, 4, 2, 1), then
// the four rows through v_readlane (scalar result; no LDS traffic, unlike ds_bpermute shuffles)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, dpp_i<0x128>(v)); v = min(v, dpp_i<0x124>(v)); v = min(v, dpp_i<0x122>(v)); v = min(v, dpp_i<0x121>(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
  v = max(v, dpp_i<0x128>(v)); v = max(v, dpp_i<0x124>(v)); v = max(v, dpp_i<0x122>(v)); v = max(v, dpp_i<0x121>(v));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the number of this wave's vector-memory instructions that may stay
// in flight): the immediate form through a uniform switch; n above the table waits for a smaller count (stricter)
__device__ __forceinline__ void wait_vm(int n) {
#define W3RC_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    W3RC_W(0) W3RC_W(1) W3RC_W(2) W3RC_W(3) W3RC_W(4) W3RC_W(5) W3RC_W(6) W3RC_W(7) W3RC_W(8) W3RC_W(9) W3RC_W(10)
    W3RC_W(11) W3RC_W(12) W3RC_W(13) W3RC_W(14) W3RC_W(15) W3RC_W(16) W3RC_W(17) W3RC_W(18) W3RC_W(19) W3RC_W(20)
    W3RC_W(21) W3RC_W(22) W3RC_W(23) W3RC_W(24) W3RC_W(25) W3RC_W(26) W3RC_W(27) W3RC_W(28) W3RC_W(29) W3RC_W(30)
    W3RC_W(31) W3RC_W(32) W3RC_W(33) W3RC_W(34) W3RC_W(35) W3RC_W(36) W3RC_W(37) W3RC_W(38) W3RC_W(39) W3RC_W(40)
    default:
      if (n > 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef W3RC_W
}

// A mover lane's share of a 64 h x 32 w plane tile (512 float4 slots, t_slot layout): slots 64 (NJ m + i) + lane.
template <int NMW_>
struct TileLane {
  static constexpr int NJ = TSLOTS / 64 / NMW_;
  unsigned voff[NJ];   // byte offset inside a [H][W] plane (clamped: loads are always legal)
  int hq[NJ], wq[NJ];  // unclamped (h, w) of the slot (store guards)
  int m;

  __device__ __forceinline__ void init(const W3P& p, int m_, int lane, int h0, int w0) {
    m = m_;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int s = 64 * (NJ * m + i) + lane, hr = s / TQ, q = (s % TQ) ^ t_swz(hr);
      hq[i] = h0 + hr; wq[i] = w0 + 4 * q;
      voff[i] = ((unsigned)min(hq[i], p.H - 1) * (unsigned)p.W + (unsigned)min(wq[i], p.W - 4)) * 4u;
    }
  }
  __device__ __forceinline__ void dma(const float* plane, float4* tile) const {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)plane, (short)0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(tile + 64 * (NJ * m + i)), 16, voff[i], 0, 0, 2);
#endif
  }
  __device__ __forceinline__ void get(const float4* tile, float4 (&v)[NJ]) const {
#pragma unroll
    for (int i = 0; i < NJ; ++i) v[i] = tile[64 * (NJ * m + i) + (threadIdx.x & 63)];
  }
  __device__ __forceinline__ void ldg(const float* plane, float4 (&v)[NJ]) const {  // the lane's slots of a global plane
#pragma unroll
    for (int i = 0; i < NJ; ++i) v[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(plane) + voff[i]);
  }
  __device__ __forceinline__ void put(float* plane, const W3P& p, const float4 (&v)[NJ]) const {
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      if (hq[i] < p.H && wq[i] < p.W) *reinterpret_cast<float4*>(plane + (size_t)hq[i] * p.W + wq[i]) = v[i];
  }
};

// min / max over a whole plane tile in LDS (every lane returns the tile's value)
__device__ __forceinline__ void tile_minmax(const float4* tile, float& mn, float& mx) {
  const int lane = threadIdx.x & 63;
  float a = INFINITY, b = -INFINITY;
#pragma unroll
  for (int i = 0; i < TSLOTS / 64; ++i) {
    const float4 v = tile[64 * i + lane];
    a = fminf(fminf(a, v.x), fminf(fminf(v.y, v.z), v.w));
    b = fmaxf(fmaxf(b, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a = fminf(a, __shfl_xor(a, o, 64));
    b = fmaxf(b, __shfl_xor(b, o, 64));
  }
  mn = a; mx = b;
}

// blockIdx.x -> tile with the workgroups of one XCD (blockIdx.x % 8) on a contiguous range of tiles: neighbouring tiles
// share halo planes / columns of the source, which then meet in one L2
__device__ __forceinline__ void decode_tile_xcd(const W3P& p, int& b, int& d0, int& h0, int& w0) {
  const int nb = gridDim.x;
  int bid = blockIdx.x;
  const int per = nb >> 3;
  if ((nb & 7) == 0) bid = (bid & 7) * per + (bid >> 3);
  const int tw = bid % p.tilesW; bid /= p.tilesW;
  const int th = bid % p.tilesH; bid /= p.tilesH;
  const int dk = bid % p.nDC;
  b = bid / p.nDC;
  d0 = dk * p.dc;
  h0 = th * TH;
  w0 = tw * TW;
}

// R: rows of the cache ring; NMW_: mover waves.  C == 1 only (the IFNet call sites); DBG (ablation build): 1 = every
// voxel takes the global-gather path, 2 = no window test (wrong values outside the window), 3 = no row DMA + no gathers
template <bool BWD, int NMW_, int R, int DBG = 0>
__global__ __launch_bounds__(64 * (NCW + NMW_)) void warp3d_rc_kernel(W3Fwd fio, W3Bwd bio, const float* __restrict__ flow,
                                                                      float* gflow, W3Add gadd, W3P p) {
  constexpr int NPL = BWD ? 4 : 3;  // planes per ring stage: flow (3) [+ grad_out]
  __shared__ float4 sF[NFR][NPL][TSLOTS];
  __shared__ float4 sO[BWD ? 1 : 2][BWD ? 1 : TSLOTS];  // forward: the output tile of slices k, k - 1
  __shared__ float4 sC[R][RSLOTS];
  __shared__ Mail mail;
  using TL = TileLane<NMW_>;

  const float* __restrict__ in = BWD ? bio.in[blockIdx.y] : fio.in[blockIdx.y];
  int b, d0, h0, w0;
  decode_tile_xcd(p, b, d0, h0, w0);
  const int HW = p.H * p.W;
  const size_t vol = (size_t)p.D * HW;
  const size_t ivol = (size_t)p.Di * p.Hi * p.Wi;
  const float* fb = flow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol;
  const float* __restrict__ vin = in + (size_t)b * ivol;  // C == 1
  const int n = min(d0 + p.dc, p.D) - d0;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

  if (wv >= NCW) {
    // ------------------------------------------------ movers ------------------------------------------------
    TL tl;
    tl.init(p, wv - NCW, lane, h0, w0);
    const int m = wv - NCW;
    const float* gob = nullptr;
    float* gfb = nullptr;
    float* outb = nullptr;
    const float* gab[3] = {nullptr, nullptr, nullptr};
    if constexpr (BWD) {
      gob = bio.gout[blockIdx.y] + (size_t)b * (bio.gbs[blockIdx.y] ? (size_t)bio.gbs[blockIdx.y] : vol);
      gfb = gflow + ((size_t)b * p.flowC + 3 * blockIdx.y) * vol;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (gadd.a[i] != nullptr) { gab[i] = gadd.a[i] + (size_t)b * gadd.bs[i] + (size_t)(3 * blockIdx.y) * vol; }
    } else {
      outb = fio.out[blockIdx.y] + (size_t)b * vol;
    }
    constexpr int DMA_PER = NPL * TL::NJ;  // tile DMA instructions per slice and mover wave
    auto load_slice = [&](int k) {
      const float* f = fb + (size_t)(d0 + k) * HW;
      float4(*st)[TSLOTS] = sF[k % NFR];
      tl.dma(f, st[0]); tl.dma(f + vol, st[1]); tl.dma(f + 2 * vol, st[2]);
      if constexpr (BWD) tl.dma(gob + (size_t)(d0 + k) * HW, st[3]);
    };
    // ---- prologue: slices 0 and 1 requested; slice 0's flow decides the window origin and its rows
    load_slice(0);
    if (n > 1) load_slice(1);
    wait_vm(n > 1 ? DMA_PER : 0);
    w3_lds_barrier();  // A: every mover's share of slice 0 is in LDS
    int xb, zb, a0, b0;
    {
      float f0n, f0x, f1n, f1x, f2n, f2x;
      tile_minmax(sF[0][0], f0n, f0x);
      tile_minmax(sF[0][1], f1n, f1x);
      tile_minmax(sF[0][2], f2n, f2x);
      const float lh0 = fs::linspace_pm1(min(h0, p.H - 1), p.H, p.stepH), lh1 = fs::linspace_pm1(min(h0 + TH - 1, p.H - 1), p.H, p.stepH);
      const float lw0 = fs::linspace_pm1(min(w0, p.W - 1), p.W, p.stepW), lw1 = fs::linspace_pm1(min(w0 + TW - 1, p.W - 1), p.W, p.stepW);
      const float ld = fs::linspace_pm1(d0, p.D, p.stepD);
      const int x0n = corner_of(lh0, f0n, p.rH, p.mW), x0x = corner_of(lh1, f0x, p.rH, p.mW);
      const int z0n = corner_of(lw0, f2n, p.rW, p.mD), z0x = corner_of(lw1, f2x, p.rW, p.mD);
      a0 = corner_of(ld, f1n, p.rD, p.mH); b0 = corner_of(ld, f1x, p.rD, p.mH);
      const int sx = max(NX - (x0x - x0n + 2), 0) >> 1, sz = max(NP - (z0x - z0n + 2), 0) >> 1;
      xb = min(max((x0n - sx) & ~3, 0), p.Wi - NX);
      zb = min(max(z0n - sz, 0), p.Di - NP);
      xb = __builtin_amdgcn_readfirstlane(xb); zb = __builtin_amdgcn_readfirstlane(zb);
      a0 = __builtin_amdgcn_readfirstlane(a0); b0 = __builtin_amdgcn_readfirstlane(b0);
    }
    // row DMA: this wave's instructions j = m, m + NMW_, ... < ROW_INSTR of a row; lane -> (plane, float4 column)
    constexpr int RJ = (ROW_INSTR + NMW_ - 1) / NMW_;
    unsigned roff[RJ];
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
      const int s = 64 * (m + NMW_ * jj) + lane;
      const int pl = s / NQ, q = s - pl * NQ;
      roff[jj] = (s < NP * NQ) ? (unsigned)pl * p.planeB + (unsigned)q * 16u : 0u;  // pad slots re-read slot 0 (never used)
    }
    const float* cbase = vin + ((size_t)zb * p.Hi * p.Wi + xb);
    (void)cbase;
    auto load_row = [&](int y, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (DBG == 3) return;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)cbase, (short)0, -1, 0x00020000);
      const unsigned so = (unsigned)y * p.rowB;
#pragma unroll
      for (int jj = 0; jj < RJ; ++jj)
        if (m + NMW_ * jj < ROW_INSTR)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(&sC[slot][64 * (m + NMW_ * jj)]), 16, roff[jj], so, 0, 0);
#endif
    };
    // cache state (identical scalars in every mover wave): rows [clo, chi) are resident, row clo in ring slot sclo
    int clo = 0, chi = 0, sclo = 0;
    // make rows [lo_req ..] up to chi_t (exclusive) resident for slice `kn` and publish its window; rows being read by
    // the slice in flight (>= clo) are never overwritten: chi <= clo + R
    auto advance = [&](int kn, int lo_req, int chi_t) {
      chi_t = min(chi_t, p.Hi);
      lo_req = min(max(lo_req, 0), p.Hi - 1);
      bool ok = true;
      if (chi == clo) { clo = chi = lo_req; sclo = 0; }  // empty: nobody reads the cache in this iteration
      else if (lo_req > chi || chi_t <= clo) { clo = chi; ok = false; }  // discontinuity: one slice without the cache
      if (ok) {
        const int nchi = max(min(chi_t, clo + R), chi);
        for (int y = chi; y < nchi; ++y) {
          int s = sclo + (y - clo);
          s -= (s >= R) ? R : 0;
          s -= (s >= R) ? R : 0;
          load_row(y, s);
        }
        chi = nchi;
        const int nclo = max(max(clo, chi - R), min(lo_req, chi));
        int s = sclo + (nclo - clo);
        s -= (s >= R) ? R : 0;
        s -= (s >= R) ? R : 0;
        sclo = s; clo = nclo;
      }
      if (m == 0 && lane == 0) {
        mail.lo[kn & 1] = clo; mail.nv[kn & 1] = ok ? chi - clo : 0; mail.sl[kn & 1] = sclo;
      }
    };
    if (m == 0 && lane == 0) { mail.xb = xb; mail.zb = zb; }
    advance(0, a0, b0 + 2);
    wait_vm(0);
    w3_lds_barrier();  // B: slice 0's rows, slice 1's tiles and the mailbox are visible
    float4 ad[BWD ? 3 : 1][BWD ? 3 : 1][TL::NJ];  // backward: the addend tiles of the slice being computed
    for (int k = 0; k < n; ++k) {
      // (1) results of slice k - 1 leave
      if (k > 0) {
        if constexpr (BWD) {
          wait_vm(k + 1 < n ? DMA_PER : 0);  // its addends were requested an iteration ago, ahead of slice k + 1's tiles
          float* g = gfb + (size_t)(d0 + k - 1) * HW;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float4 v[TL::NJ];
            tl.get(sF[(k - 1) % NFR][c], v);
#pragma unroll
            for (int a = 0; a < 3; ++a)
              if (gab[a] != nullptr) {
#pragma unroll
                for (int i = 0; i < TL::NJ; ++i) {
                  v[i].x += ad[a][c][i].x; v[i].y += ad[a][c][i].y; v[i].z += ad[a][c][i].z; v[i].w += ad[a][c][i].w;
                }
              }
            tl.put(g + (size_t)c * vol, p, v);
          }
        } else {
          float4 v[TL::NJ];
          tl.get(sO[(k - 1) & 1], v);
          tl.put(outb + (size_t)(d0 + k - 1) * HW, p, v);
        }
      }
      // (2) rows of slice k + 1, predicted from the y0 range measured on slice k - 1 (slice 0: from its own flow)
      int younger = 0;
      if (k + 1 < n) {
        int a, bb, sh;
        if (k == 0) { a = a0; bb = b0; sh = 1; }
        else {
          int mn = mail.st[(k - 1) & 1][lane & (NCW - 1)][0], mx = mail.st[(k - 1) & 1][lane & (NCW - 1)][1];
#pragma unroll
          for (int o = 1; o < NCW; o <<= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
          a = __builtin_amdgcn_readfirstlane(mn); bb = __builtin_amdgcn_readfirstlane(mx); sh = 2;
        }
        // the predicted rows [a + sh, bb + sh + 1] plus, while the R - 1 rows a steady window holds leave room, one row
        // below and one above (the range drifts by a fraction of a row per slice)
        const int core = bb - a + 2;
        const int below = (core + 1 <= R - 1) ? 1 : 0, above = (core + below + 1 <= R - 1) ? 1 : 0;
        advance(k + 1, a + sh - below, bb + sh + 2 + above);
      }
      // (3) backward: the addends of slice k (used in the next iteration)
      if constexpr (BWD) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
          if (gab[a] != nullptr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) tl.ldg(gab[a] + (size_t)c * vol + (size_t)(d0 + k) * HW, ad[a][c]);
            younger += 3 * TL::NJ;
          }
      }
      // (4) tiles of slice k + 2 into the stage slice k - 1 has just left (this wave reads / refills its own slots only;
      //     the other planes of that stage were last read by the compute waves before the previous barrier)
      if (k + 2 < n) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        load_slice(k + 2);
        younger += DMA_PER;
      }
      // (5) the rows of slice k + 1 (and everything older: slice k + 1's tiles) have landed
      wait_vm(younger);
      w3_lds_barrier();
    }
    // results of the last slice
    if constexpr (BWD) {
      wait_vm(0);
      float* g = gfb + (size_t)(d0 + n - 1) * HW;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float4 v[TL::NJ];
        tl.get(sF[(n - 1) % NFR][c], v);
#pragma unroll
        for (int a = 0; a < 3; ++a)
          if (gab[a] != nullptr) {
#pragma unroll
            for (int i = 0; i < TL::NJ; ++i) {
              v[i].x += ad[a][c][i].x; v[i].y += ad[a][c][i].y; v[i].z += ad[a][c][i].z; v[i].w += ad[a][c][i].w;
            }
          }
        tl.put(g + (size_t)c * vol, p, v);
      }
    } else {
      float4 v[TL::NJ];
      tl.get(sO[(n - 1) & 1], v);
      tl.put(outb + (size_t)(d0 + n - 1) * HW, p, v);
    }
    return;
  }
  // -------------------------------------------------- compute waves --------------------------------------------------
  const int hq = h0 + lane;
  const int h = min(hq, p.H - 1);
  const float lin_h = fs::linspace_pm1(h, p.H, p.stepH);
  float lin_w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lin_w[i] = fs::linspace_pm1(min(w0 + 4 * wv + i, p.W - 1), p.W, p.stepW);
  const int slot = t_slot(lane, wv);
  const float k0 = (p.mW * 0.5f) * p.rH, k1 = (p.mH * 0.5f) * p.rD, k2 = (p.mD * 0.5f) * p.rW;
  typedef const __attribute__((address_space(3))) char* lds_cchar_t;
  const lds_cchar_t cch = (lds_cchar_t)(&sC[0][0]);
  typedef const __attribute__((address_space(3))) float* lds_cfloat_t;
  auto lds_pair = [&](unsigned off) { Pair q; lds_cfloat_t f = (lds_cfloat_t)(cch + off); q.a = f[0]; q.b = f[1]; return q; };
  w3_lds_barrier();  // A
  w3_lds_barrier();  // B
  const int xb = mail.xb, zb = mail.zb;
  for (int k = 0; k < n; ++k) {
    float4(*st)[TSLOTS] = sF[k % NFR];
    const float lin_d = fs::linspace_pm1(d0 + k, p.D, p.stepD);
    const float4 f0 = st[0][slot], f1 = st[1][slot], f2 = st[2][slot];
    const float fa[4] = {f0.x, f0.y, f0.z, f0.w}, fbv[4] = {f1.x, f1.y, f1.z, f1.w}, fc[4] = {f2.x, f2.y, f2.z, f2.w};
    float gv[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BWD) {
      const float4 g4 = st[3][slot];
      gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
    }
    const int lo = mail.lo[k & 1], nv = mail.nv[k & 1], sl = mail.sl[k & 1];
    int ymin = 0x7fffffff, ymax = 0;
    float o0[4], o1[4], o2[4];
    // pass 1: sample positions, window test, all corner reads of the four voxels issued (cache reads unconditionally --
    // a voxel outside the window reads slot 0 and is overwritten by pass 1b); pass 1b: global gathers of the voxels
    // outside the window, all issued before any is used; pass 2: blend.  One wait per slice instead of one per voxel.
    SampI<BWD> sv[4];
    Pair r00[4], r01[4], r10[4], r11[4];  // (z, y) corner pairs along x
    bool hitv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const SampI<BWD> s = sample_i<BWD>(p, lin_h, lin_d, lin_w[i], fa[i], fbv[i], fc[i]);
      sv[i] = s;
      ymin = min(ymin, s.y0); ymax = max(ymax, s.y0);
      // the pair (x0, x0 + 1) starts one column lower on the far border (ld_pair's rule: the +1 weight is exactly 0)
      const int rxs = s.x0 - xb - (s.px ? 0 : 1), rz = s.z0 - zb, ry = s.y0 - lo;
      bool hit = (unsigned)rxs < (unsigned)(NX - 1) && (unsigned)rz < (unsigned)(NP - (s.pz ? 1 : 0)) &&
                 (unsigned)ry < (unsigned)max(nv - (s.py ? 1 : 0), 0);
      if (DBG == 1) hit = false;
      if (DBG == 2 || DBG == 3) hit = true;
      hitv[i] = hit;
      int sy0 = sl + ry;
      sy0 -= (sy0 >= R) ? R : 0;
      int sy1 = sy0 + (s.py ? 1 : 0);
      sy1 -= (sy1 >= R) ? R : 0;
      unsigned c0 = (unsigned)sy0 * RSB + (unsigned)rz * PLB + (unsigned)rxs * 4u;
      unsigned c1 = (unsigned)sy1 * RSB + (unsigned)rz * PLB + (unsigned)rxs * 4u;
      if (DBG == 2 || DBG == 3) { c0 %= (unsigned)(R * RSB - PLB - 8); c1 %= (unsigned)(R * RSB - PLB - 8); c0 &= ~3u; c1 &= ~3u; }
      unsigned dzb = s.pz ? PLB : 0u;
      if (!hit) { c0 = 0u; c1 = 0u; dzb = 0u; }
      r00[i] = lds_pair(c0);
      r01[i] = lds_pair(c1);
      r10[i] = lds_pair(c0 + dzb);
      r11[i] = lds_pair(c1 + dzb);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!hitv[i]) {
        const SampI<BWD>& s = sv[i];
        const unsigned o000 = (__umul24(__umul24((unsigned)s.z0, (unsigned)p.Hi) + (unsigned)s.y0, (unsigned)p.Wi) + (unsigned)s.x0) * 4u;
        const unsigned dx = s.px ? 4u : 0u, dy = s.py ? p.rowB : 0u, dz = s.pz ? p.planeB : 0u;
        const unsigned o010 = o000 + dy, o100 = o000 + dz, o110 = o100 + dy;
        r00[i] = ld_pair_raw(vin, o000, dx); r01[i] = ld_pair_raw(vin, o010, dx);
        r10[i] = ld_pair_raw(vin, o100, dx); r11[i] = ld_pair_raw(vin, o110, dx);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const SampI<BWD>& s = sv[i];
      const float v000 = s.px ? r00[i].a : r00[i].b, v001 = r00[i].b, v010 = s.px ? r01[i].a : r01[i].b, v011 = r01[i].b;
      const float v100 = s.px ? r10[i].a : r10[i].b, v101 = r10[i].b, v110 = s.px ? r11[i].a : r11[i].b, v111 = r11[i].b;
      const float c00 = lerp(v000, v001, s.ax), c01 = lerp(v010, v011, s.ax);
      const float c10 = lerp(v100, v101, s.ax), c11 = lerp(v110, v111, s.ax);
      if constexpr (BWD) {
        const bool live = (hq < p.H) && (w0 + 4 * wv + i < p.W);
        const float g = live ? gv[i] : 0.f;
        const float dx = lerp(lerp(v001 - v000, v011 - v010, s.ay), lerp(v101 - v100, v111 - v110, s.ay), s.az);
        const float dy = lerp(c01 - c00, c11 - c10, s.az);
        const float dz = lerp(c10, c11, s.ay) - lerp(c00, c01, s.ay);
        o0[i] = fmaf(g * s.mx, dx, 0.f) * k0;
        o1[i] = fmaf(g * s.my, dy, 0.f) * k1;
        o2[i] = fmaf(g * s.mz, dz, 0.f) * k2;
      } else {
        o0[i] = lerp(lerp(c00, c01, s.ay), lerp(c10, c11, s.ay), s.az);
      }
    }
    if constexpr (BWD) {
      st[0][slot] = make_float4(o0[0], o0[1], o0[2], o0[3]);
      st[1][slot] = make_float4(o1[0], o1[1], o1[2], o1[3]);
      st[2][slot] = make_float4(o2[0], o2[1], o2[2], o2[3]);
    } else {
      sO[k & 1][slot] = make_float4(o0[0], o0[1], o0[2], o0[3]);
    }
    ymin = wave_min_i(ymin); ymax = wave_max_i(ymax);
    if (lane == 0) { mail.st[k & 1][wv][0] = ymin; mail.st[k & 1][wv][1] = ymax; }
    w3_lds_barrier();
  }
}

// does the cached kernel apply?  (the window must fit inside the sampled volume; 16-byte row pieces)
inline bool applicable(const W3P& p, const void* in0, const void* in1) {
  auto al = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return p.C == 1 && p.Wi % 4 == 0 && p.Wi >= NX && p.Di >= NP && p.Hi >= 2 && al(in0) && al(in1) &&
         (long long)p.Di * p.Hi * p.Wi * 4 < (1ll << 31);
}

// slices per workgroup: long runs amortise the window set-up and the (dc + 1 + spread) / dc row over-fetch, but the
// launch should still deal four workgroups to every CU (same-box repeats at 2 x 256^3, scripts/gpu/r5_w3dc.sh: forward 0.315 /
// 0.292 / 0.290 ms at 16 / 32 / 64 slices, three-addend backward 0.933 / 0.947 / 0.971)
inline int pick_dc(const W3P& p, int npair) {
  int dc = 64;
  while (dc > 8 && (long long)p.B * fs::cdiv(p.D, dc) * p.tilesH * p.tilesW * npair < 4 * 256) dc >>= 1;
  return min(dc, max(p.D, 1));
}

}  // namespace rc
