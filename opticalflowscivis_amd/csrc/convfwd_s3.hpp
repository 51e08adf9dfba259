// convfwd_s3.hpp -- round 5: the k = 4, stride 2, pad 1 forward convolution (IFBlock's conv0 pair, and the input gradient
// of the heads' transposed convolutions) with fp32 ACCURACY on the bf16 matrix rate.  Included by convfwd.hip inside its
// anonymous namespace.
//
// An fp32 operand is exactly the sum of three bf16 pieces (a = a0 + a1 + a2, 8 + 8 + 8 significant bits); every pairwise
// product of pieces is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so
//     a * b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)      [the three dropped terms are <= 2^-26 |a b|]
// costs 6 bf16 MFMAs of K = 16 (32 cycles each) where the fp32 path issues 8 v_mfma_f32_32x32x2_f32 of K = 2 (64 cycles
// each): 192 vs 512 matrix-pipe cycles per 16 reduction elements.  scripts/micro/split_bf16_gemm.hip is the go / no-go
// measurement (profiles/r05_split_bf16.txt): on a K = 1728 reduction of random data the 6-product form's maximum error
// is 0.97 x the fp32 MFMA's (which is bitwise the fmaf chain), and its LDS-fed tile loop runs 1.75 x as fast.
//
// Decomposition.  A workgroup owns 1 z x 16 y x 32 x output voxels for CP = 32 MT output channels: eight matrix waves
// (two per SIMD: while one waits for its LDS operands the other issues MFMAs) with NT = 2 output rows each.  The reduction runs in STAGES of (input-channel pair, kz): per stage the loader waves
// bring ONE input z-slice of the two channels (34 rows x 72 columns) global -> registers, split every value into its three
// pieces and park them in LDS as [piece][row][column][2 channels] (a 4-byte word = the bf16 pair of the two channels), and
// LDS-DMA the stage's slab of pre-split weights.  The 16 reduction elements of one MFMA are 2 ky x 4 kx x 2 channels: the
// lane half `kh` takes ky = 2 kyp + kh, and a lane's 8 values are 4 consecutive kx of the channel pair = 16 contiguous
// LDS bytes at column 2 x + kx of its row (8-byte aligned: two ds_read_b64); the weight operand is one ds_read_b128 of
// [piece][kyp][kh][co][4 kx x 2 channels].  Two LDS input stages + a ring of three weight slabs (requested a stage
// ahead); the loaders run one stage ahead; ONE barrier per stage.  The accumulators have the layout of the fp32 kernels', so the epilogue (bias, PReLU output, addend, fused PReLU
// backward with its partial sums) is the same text (convfwd_epilogue.hpp).
//
// Alignment trick of the loader: global rows are read as 16-byte pieces at x = 2 ox0 - 4 + 4 j (Wi % 4 == 0: a piece is
// inside or outside the volume as a whole, outside ones come back as zeros from the buffer descriptor); the matrix
// lanes want column 2 x + 3 + kx of those, an ODD word -- so the LDS image is shifted by one word: slot j of a row holds
// {word 3 of piece j - 1, words 0..2 of piece j}, the neighbour's word arriving by one DPP wave_shr per piece.
#pragma once

typedef __bf16 s3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s3_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned s3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned s3_u32x2 __attribute__((ext_vector_type(2)));

constexpr int S3_TY = 16, S3_TW = 32;
constexpr int S3_NMW = 8;                    // matrix waves per workgroup: two per SIMD (one hides the other's LDS operand latency)
constexpr int S3_YT = (S3_TY - 1) * 2 + 4;   // 34 staged rows
constexpr int S3_XP = 72;                    // staged words per row (64 + 4 taps + the left piece + shift)
constexpr int S3_ROWB = S3_XP * 4, S3_PIECEB = (S3_YT + 1) * S3_ROWB, S3_INB = 3 * S3_PIECEB;  // (+1: a spare row the
                                             // idle loader lanes write to, so that the conversion has no divergent code)

// words of the pre-split weight slab: [channel group of CP][channel pair][kz][piece 3][kyp 2][kh 2][co CP][kx 4]
inline long long s3_slab_words(int Cin, int CoutP) { return (long long)CoutP * ((Cin + 1) / 2) * 4 * 48; }

// (MT = 1: 78 KB of LDS and <= 80 registers: TWO workgroups per CU -- a layer with 32 output channels has few stages per
// workgroup (24 at 12 input channels), so one workgroup's prologue / epilogue and stage-start bubbles hide behind the other's MFMAs)
template <int MT, int NMW, int NLW>
__global__ __launch_bounds__(64 * (NMW + NLW), MT == 1 ? 6 : 1) void conv3d_fwd_s3_kernel(const float* __restrict__ X,
                                                                         const unsigned* __restrict__ Ws,
                                                                         const float* __restrict__ bias,
                                                                         float* __restrict__ Y, FP p) {
  constexpr int NT = S3_TY / NMW, R = 1, TY = S3_TY, TW = S3_TW;  // output rows per matrix wave
  constexpr int EPI_ROWS = NMW;
  constexpr int CP = 32 * MT;
  constexpr int WB = 192 * CP;                 // bytes of a stage's weight slab
  constexpr int WOFF = 2 * S3_INB;             // LDS: two input stages, then a ring of three weight slabs
  constexpr int NWI = WB / 1024;               // LDS-DMA instructions per weight slab
  constexpr int NWW = (NWI + NLW - 1) / NLW;   // ... per loader wave (a wave whose piece would lie past the slab copies an earlier
                                               // piece again: same bytes to the same place, and every wave issues NWW pieces)
  constexpr int WBP = WB;                      // LDS stride of a slab
  constexpr int RPP = 3 * NLW;                 // rows per loader pass (3 rows of 18 pieces per wave)
  constexpr int PASSES = (S3_YT + RPP - 1) / RPP;
  static_assert(WOFF + 3 * WBP <= 160 * 1024, "the stages fit the CU's LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[WOFF + 3 * WBP];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave;  // (matrix waves: 0 .. NMW-1)
  const int col = lane & 31, kh = lane >> 5;

  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);
  }
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  const int oz0 = tzi, oy0 = tyi * TY, ox0 = txi * TW;
  const int co0 = blockIdx.y * CP;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  const int CinP2 = (p.Cin + 1) / 2;
  const int NS = CinP2 * 4;  // stages: (channel pair, kz)

  if (wave >= NMW) {
#if defined(__HIP_DEVICE_COMPILE__)
    // ---- loader waves.  They are on the critical path (a stage is ~200 loader instructions against 24-48 MFMAs per matrix
    // wave), so the loop is kept lean: stage-invariant lane offsets, scalar-only address arithmetic per stage, no divergent
    // code (idle lanes write a spare LDS row), hand-counted waits.
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - NMW;
    // this lane's piece of the rows it converts: slot = lane / 18 (3 rows per wave and pass), piece j = lane % 18
    const int slot = lane / 18, j = lane - 18 * slot;
    unsigned goff[PASSES];   // byte offset inside one (channel, z-slice) plane, or out of range (zeros come back)
    unsigned loff[PASSES];   // byte offset of the 16-byte slot inside a piece image (idle lanes: the spare row)
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int yy = ps * RPP + lw * 3 + slot;
      const int gy = oy0 * 2 - 1 + yy, gx = ox0 * 2 - 4 + 4 * j;
      const bool act = lane < 54 && yy < S3_YT;
      const bool ok = act && gy >= 0 && gy < p.Hi && gx >= 0 && gx + 3 < p.Wi;
      goff[ps] = ok ? ((unsigned)gy * (unsigned)p.Wi + (unsigned)gx) * 4u : DMA_OOB;
      loff[ps] = (unsigned)((act ? yy : S3_YT) * S3_ROWB + (act ? j : (lane & 15)) * 16);
    }
    const unsigned planeB = (unsigned)p.Hi * (unsigned)p.Wi * 4u;
    // Input pieces by inline assembly: the compiler's wait-count pass puts `s_waitcnt vmcnt(0..2)` in front of the first use
    // of a register loaded in an earlier loop trip (it loses the issue order across the back edge), which would wait for
    // the loads of the NEXT stages as well -- the whole prefetch.  The loader's vector-memory traffic is counted by hand
    // instead; `tie` makes every use of a set come after the wait.
    typedef int s3_i32x4 __attribute__((ext_vector_type(4)));
    s3_u32x4 ld2[MT == 1 ? 1 : 2][PASSES][2];  // register sets of input pieces in flight (MT = 2: stages s + 1 and s + 2)
    auto issue_loads = [&](int s, s3_u32x4 (&ld)[PASSES][2]) {
      const int cp = s >> 2, kz = s & 3;
      const int gz = oz0 * 2 - 1 + kz;
      const bool zok = gz >= 0 && gz < p.Di;
      const unsigned zoff = zok ? (unsigned)gz * planeB : 0u;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ch = 2 * cp + c;
        const bool live = zok && ch < p.Cin;
        const int chc = ch < p.Cin ? ch : 0;
        const float* base = p.nsrc ? p.src[chc] + (size_t)b * (size_t)p.sbs[chc] : X + ((size_t)b * p.Cin + chc) * xvol;
        const unsigned long long a = (unsigned long long)base + zoff;
        s3_i32x4 r;  // one z-slice of one channel; zero records: every lane out of range, zeros come back
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane(live ? (int)planeB : 0);
        r[3] = 0x00020000;
        // (s_nop 4: the resource words come from v_readfirstlane, and a vector-memory instruction must not read a scalar
        // register within 5 cycles of a vector-ALU write to it -- the compiler inserts such wait states for its own
        // instructions, not for text inside an asm statement)
        asm volatile("s_nop 4" ::"s"(r));
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[ps][c]) : "v"(goff[ps]), "s"(r));
      }
    };
    // wait for EVERYTHING in flight and tie the set's registers to the wait in ONE statement.  To the compiler the
    // destination of an inline-assembly load is an ordinary value from the asm statement on: it may COPY it (tied operands,
    // live-range splits, loop exits) -- the copy reads stale data -- and then hand the original register to another value,
    // which the load overwrites when it lands.  Silent on a warm cache, wrong on a cold one.  With wait and tie in one
    // statement every use comes behind the wait and the registers stay the set's; scripts/check_inflight_regs.py verifies on
    // the compiled code of every build that nothing reads OR writes a register between its load and the wait that covers it
    auto wait_all = [&](s3_u32x4 (&ld)[PASSES][2]) {
      static_assert(PASSES <= 3, "operand list below");
      if constexpr (PASSES == 3)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0][0]), "+v"(ld[0][1]), "+v"(ld[1][0]), "+v"(ld[1][1]), "+v"(ld[2][0]), "+v"(ld[2][1]) : : "memory");
      else if constexpr (PASSES == 2)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0][0]), "+v"(ld[0][1]), "+v"(ld[1][0]), "+v"(ld[1][1]) : : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0][0]), "+v"(ld[0][1]) : : "memory");
    };
    // The lane offsets of the slab copies live in registers of their own, written once and kept to the last wait (a
    // precaution from the bring-up of csrc/convtr_s3.hpp; scripts/micro/lds_dma_hazards.hip later showed that a copy does
    // tolerate a rewrite of its address register -- what had corrupted the offsets there was a register handed out while an
    // inline-assembly load into it was still in flight, see wait_all)
    unsigned woff[NWW];
#pragma unroll
    for (int k = 0; k < NWW; ++k) {
      int i = lw + NLW * k;
      if (i >= NWI) i -= NLW;  // (wave-uniform)
      woff[k] = (unsigned)(1024 * i + 16 * lane);
      asm volatile("" : "+v"(woff[k]));
    }
    auto issue_wdma = [&](int s) {
      // the stage's slab: words [mg][cp][kz] x 48 CP, contiguous
      const unsigned* src = Ws + ((size_t)blockIdx.y * NS + s) * (size_t)(48 * CP);
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, WB, 0x00020000);
      unsigned char* dst = lds + WOFF + (s % 3) * WBP;
#pragma unroll
      for (int k = 0; k < NWW; ++k) {
        int i = lw + NLW * k;
        if (i >= NWI) i -= NLW;  // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + 1024 * i), 16, woff[k], 0, 0, 0);
      }
    };
    auto convert = [&](int s, const s3_u32x4 (&ld)[PASSES][2]) {
      unsigned char* dst = lds + (s & 1) * S3_INB;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const float a[4] = {__uint_as_float(ld[ps][0].x), __uint_as_float(ld[ps][0].y), __uint_as_float(ld[ps][0].z), __uint_as_float(ld[ps][0].w)};
        const float c[4] = {__uint_as_float(ld[ps][1].x), __uint_as_float(ld[ps][1].y), __uint_as_float(ld[ps][1].z), __uint_as_float(ld[ps][1].w)};
        unsigned h[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float ra = a[i], rc = c[i];
          h[0][i] = s3_pack(ra, rc);
          ra -= __uint_as_float(h[0][i] << 16); rc -= __uint_as_float(h[0][i] & 0xffff0000u);
          asm volatile("" : "+v"(ra), "+v"(rc));  // (keeps the SLP vectoriser from pairing the subtractions: v_pk_add_f32 + moves)
          h[1][i] = s3_pack(ra, rc);
          ra -= __uint_as_float(h[1][i] << 16); rc -= __uint_as_float(h[1][i] & 0xffff0000u);
          asm volatile("" : "+v"(ra), "+v"(rc));
          h[2][i] = s3_pack(ra, rc);
        }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const unsigned prev3 = __builtin_amdgcn_update_dpp(0u, h[pc][3], 0x138, 0xf, 0xf, false);  // wave_shr:1
          const s3_u32x4 v = {prev3, h[pc][0], h[pc][1], h[pc][2]};
          *reinterpret_cast<s3_u32x4*>(dst + pc * S3_PIECEB + loff[ps]) = v;
        }
      }
    };
    // The weight slab of a stage is requested a whole stage ahead (a ring of three slabs), its input pieces TWO stages
    // ahead (two register sets).  At the wait of stage s the requests younger than what it needs are the weight pieces of
    // stage s + 1 (NWW) and the input pieces of stage s + 1 (2 PASSES).
    if constexpr (MT == 1) {
      // two workgroups per CU (80 registers): ONE register set -- the pieces of stage s + 1 are requested while stage s is
      // converted; the partner workgroup covers what that leaves exposed
      s3_u32x4 (&ld)[PASSES][2] = ld2[0];
      issue_wdma(0);
      issue_loads(0, ld);
      for (int s = 0; s < NS; ++s) {
        // everything requested so far has landed: the pieces of stage s and its weight slab (vmcnt(0), the next slab requested
        // BEHIND it: nothing is counted past).  One statement with the tie: see wait_all.
        wait_all(ld);
        if (s + 1 < NS) {
          issue_wdma(s + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef FS_ABLATION
        if (!(p.ab & 16))
#endif
        convert(s, ld);
        if (s + 1 < NS) issue_loads(s + 1, ld);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
    auto stage = [&](int s, s3_u32x4 (&ld)[PASSES][2]) {
      // the pieces of stage s (requested two stages ago) and its slab have landed; so have, as a rule, the pieces of stage
      // s + 1, a stage old: vmcnt(0) instead of a count that would rely on copies and loads completing in issue order
      wait_all(ld);
      issue_wdma(s + 1);                               // (its slab was last read three stages ago)
      __builtin_amdgcn_sched_barrier(0);
#ifdef FS_ABLATION
      if (!(p.ab & 16))  // (measurement: no conversion / LDS writes -- FLOWSCI_S3_AB=16)
#endif
      convert(s, ld);
      if (s + 2 < NS) issue_loads(s + 2, ld);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // stage s is ready; the matrix waves are done with stage s - 1
    };
    issue_wdma(0);
    issue_loads(0, ld2[0]);
    issue_loads(1, ld2[1]);                            // (NS >= 4)
    int s = 0;
    for (; s + 2 < NS; s += 2) {                       // NS is even: the last two stages are peeled
      stage(s, ld2[0]);
      stage(s + 1, ld2[1]);
    }
    stage(s, ld2[0]);                                  // s = NS - 2: no more input pieces to request
    wait_all(ld2[1]);                                  // last stage: everything has landed
    convert(s + 1, ld2[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    }
    // (a use of the copies' address registers behind the last wait: their registers are not handed to anything else while
    // a copy may still read them)
#pragma unroll
    for (int k = 0; k < NWW; ++k) asm volatile("" ::"v"(woff[k]));
#else
    (void)xvol; (void)NWW; (void)PASSES; (void)WBP;
#endif
    return;
  }

  // ---- matrix waves: wave wv owns output rows oy0 + NT wv .. + NT - 1, lane column `col`, ky parity kh
  constexpr int ly = 0;
  const int lx = col;
  const int wz = 0, wy = wv * NT;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  // byte offsets inside a stage: input word 2 lx + 4 of row (wy + n) 2 + 2 kyp + kh; weight slot (kyp, kh, col + 32 m)
  const unsigned bO = (unsigned)((wy * 2 + kh) * S3_ROWB + (2 * lx + 4) * 4);
  const unsigned aO = (unsigned)(WOFF + (kh * CP + col) * 16);

  __builtin_amdgcn_s_barrier();  // stage 0 is ready
  int wslot = 0;  // s % 3
  for (int s = 0; s < NS; ++s) {
    const unsigned char* sb = lds + (s & 1) * S3_INB;   // input pieces of this stage
    const unsigned char* sw = lds + wslot * WBP;         // its weight slab (aO carries the ring's base)
    wslot = wslot == 2 ? 0 : wslot + 1;
#ifdef FS_ABLATION
    if (!(p.ab & 64))
#endif
#pragma unroll
    for (int kyp = 0; kyp < 2; ++kyp) {
      s3_bf16x8 a[3][MT], bq[3][NT];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a[pc][m] = __builtin_bit_cast(s3_bf16x8, *reinterpret_cast<const s3_u32x4*>(sw + aO + ((pc * 2 + kyp) * 2 * CP + m * 32) * 16));
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const unsigned char* q = sb + bO + pc * S3_PIECEB + (n * 2 + 2 * kyp) * S3_ROWB;
          const s3_u32x2 lo = *reinterpret_cast<const s3_u32x2*>(q), hi = *reinterpret_cast<const s3_u32x2*>(q + 8);
          const s3_u32x4 v = {lo.x, lo.y, hi.x, hi.y};
          bq[pc][n] = __builtin_bit_cast(s3_bf16x8, v);
        }
      }
#ifdef FS_ABLATION
      if (p.ab & 32) continue;  // (measurement: operand reads without the MFMAs -- FLOWSCI_S3_AB=32)
#endif
      // the six products, small terms first: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][m], bq[PB[q]][n], acc[m][n], 0, 0, 0);
    }
    if (s + 1 < NS) __builtin_amdgcn_s_barrier();  // stage s + 1 is ready, everyone is done reading stage s
  }

#include "convfwd_epilogue.hpp"
}

template <int MT, int NMW, int NLW>
int launch_s3(const float* X, const float* Ws, const float* bias, float* Y, FP& p, hipStream_t st) {
  p.tz = p.Do; p.ty = fs::cdiv(p.Ho, S3_TY); p.tx = fs::cdiv(p.Wo, S3_TW);
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  const int mgroups = p.CoutP / (32 * MT);
  if (p.tiles >= (1ll << 31) || mgroups > 65535) return FS_ERR_SHAPE;
#ifdef FS_ABLATION
  static const int s3_ab = (int)FS_AB_ENV_LL("FLOWSCI_S3_AB", 0);  // 16: no conversion, 32: no MFMAs, 64: no operand reads either (wrong results by design)
  p.ab = s3_ab;
#endif
  hipLaunchKernelGGL((conv3d_fwd_s3_kernel<MT, NMW, NLW>), dim3((unsigned)p.tiles, mgroups), dim3(64 * (NMW + NLW)), 0, st, X,
                     reinterpret_cast<const unsigned*>(Ws), bias, Y, p);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
