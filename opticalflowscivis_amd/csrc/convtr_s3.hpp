// convtr_s3.hpp -- round 5: ConvTranspose3d(4, 2, 1) with 17..32 output channels per slice (IFBlock's deconv1 and the input
// gradient of conv0's second convolution, 64 -> 32) with fp32 ACCURACY on the bf16 matrix rate: the split-operand form of
// convfwd_s3.hpp (three bf16 pieces per operand, six products, fp32 accumulation) for the transposed layers.  Included by
// convtr.hip inside its anonymous namespace.
//
// Why a kernel of its own: profiles/r05_tr_bound_ab.txt -- the fp32 kernels of these layers are matrix-bound (no change
// with all their global -> LDS traffic switched off), at 0.75 of the fp32 matrix peak; only a faster MFMA moves them.
//
// Decomposition.  Sub-pixel form as in convtr.hip: output parity class (pz, py, px) of input-grid position q reads two taps
// per axis.  A workgroup owns 2 z x 3 y x 32 x positions and all 8 classes of 32 output channels: eight matrix waves,
// wave = (pz, py, z row): BOTH x parities of three y rows (6 accumulator tiles, 96 registers) -- a lane then holds the two
// x-neighbouring outputs 2 qx, 2 qx + 1 of a channel: 8-byte stores, no exchange.  The reduction runs in STAGES of four input
// channels.  The 16 reduction elements of one v_mfma_f32_32x32x16_bf16 are 2 y-taps x 2 x-taps x 4 channels: the lane half
// `kh` takes the y tap, a lane's 8 values are the four channels at the two x-tap positions -- two ds_read_b64 of an LDS image
// [piece][z 4][y 5][x 40][4 channels] (8 bytes per position, so the taps' +-1 offsets stay aligned; the three positions
// x - 1, x, x + 1 serve both x parities).  The z tap is the outer loop (2 per class).  Weight operand: one ds_read_b128 of the
// pre-split slab [stage][class][z tap][piece][kh][co][x tap, the lower position first][channel] (FS_WPREP_TRS3), 48 KB per stage, LDS-DMA'd one stage
// ahead into the other of two buffers; the loader waves also convert the stage's input brick (20 rows x 10 sixteen-byte
// pieces x 4 channels = one item per loader lane: four loads, 16 values -> three pieces -> six ds_write_b128).  Two input
// stages + two weight stages = 135 KB of LDS, ONE barrier per stage.  PERSISTENT workgroups (one per CU, bricks dealt round
// robin): the loaders' stage stream runs across brick boundaries (bricks requested two stages ahead into two register sets),
// so the first stages of the next brick are staged while the matrix waves store the previous one, and those stores drain
// under the next brick's MFMAs (the one-workgroup-per-brick form: 0.81 ms for 64 -> 32 at 64^3, 21 % of it a store burst of all
// CUs at once, `scripts/gpu/r5_t3.sh`).
#pragma once

typedef __bf16 t3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned t3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned t3_u32x2 __attribute__((ext_vector_type(2)));
typedef int t3_i32x4 __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) t3_u32x2* t3_lds_v64;  // (volatile LDS read of 8 bytes: never paired)

constexpr int T3_TZ = 2, T3_TY = 3, T3_TW = 32;
constexpr int T3_NMW = 8, T3_NLW = 4;
constexpr int T3_ZT = T3_TZ + 2, T3_YT = T3_TY + 2, T3_XP = 40;   // staged rows / positions per row (x: qx0 - 4 .. qx0 + 35)
constexpr int T3_NROW = T3_ZT * T3_YT;                            // 20
constexpr int T3_NQ = T3_XP / 4;                                  // 16-byte pieces per row and channel
constexpr int T3_ROWB = T3_XP * 8;                                // bytes per staged row of one piece image (4 bf16 per position)
constexpr int T3_PIECEB = (T3_NROW + 1) * T3_ROWB;                // (+1: a spare row for the idle loader lanes)
constexpr int T3_INB = 3 * T3_PIECEB;
constexpr int T3_WB32 = 8 * 2 * 3 * 1024;                         // bytes of a stage's weight slab: [class][z tap][piece] x 1 KB
constexpr int T3_WORDS32 = T3_WB32 / 4;
static_assert(T3_NROW * T3_NQ <= 64 * T3_NLW, "one (row, piece) item per loader lane");

// 4-byte words of the pre-split slab of ONE 32-channel slice
inline long long t3_slab_words(int Cin) { return (long long)((Cin + 3) / 4) * T3_WORDS32; }
// ... of the 16-row form (7..16 output channels): [stage][class 8][piece 3] x 1 KB = [x/y tap 4][co 16][z tap 2][channel 4]
constexpr int T3_WB16 = 8 * 3 * 1024;
inline long long t3_slab_words16(int Cin) { return (long long)((Cin + 3) / 4) * (T3_WB16 / 4); }

// M16: the 16-row form for 7..16 output channels (the input gradient of conv0[0], 32 -> 11 / 12): v_mfma_f32_16x16x32_bf16,
// whose 32 reduction elements are ALL eight taps of a class x 4 channels -- the lane quarter `kq` takes the (y tap, x tap)
// pair, a lane's 8 values are the 4 channels at the two z taps (two ds_read_b64 of the same LDS image); a wave's 6 row tiles
// become 12 tiles of 16 positions (48 accumulator registers).  Same loaders, same stages, half the slab.
template <bool M16>
__global__ __launch_bounds__(64 * (T3_NMW + T3_NLW), 1) void convtr_s3_kernel(const float* __restrict__ X,
                                                                             const unsigned* __restrict__ Ws_,
                                                                             const float* __restrict__ bias_,
                                                                             float* __restrict__ Y_, TP p) {
  constexpr int NMW = T3_NMW, NLW = T3_NLW;
  constexpr int WOFF = 2 * T3_INB;
  constexpr int T3_WB = M16 ? T3_WB16 : T3_WB32;  // (this instantiation's stage slab)
  [[maybe_unused]] constexpr int T3_WORDS = T3_WB / 4;
  constexpr int NWW = T3_WB / 1024 / NLW;  // LDS-DMA instructions per weight slab and loader wave
  constexpr int STG1 = 4 * 256 * 16;       // one staging area of the loader waves: [channel][item] x 16 bytes
  constexpr int NSTG = M16 ? 2 : 1;        // the 16-row form has the LDS for two: bricks requested TWO stages ahead (its
                                           // stages are half as long: with one, the loaders' request -> convert chain was the period)
  constexpr int STGB = NSTG * STG1;
  static_assert(WOFF + 2 * T3_WB + STGB + 256 <= 160 * 1024, "the stages fit the CU's LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[WOFF + 2 * T3_WB + STGB + 256];
  float* const sBias = reinterpret_cast<float*>(lds + WOFF + 2 * T3_WB + STGB);  // [32] bias, [32] PReLU slopes: a global
  float* const sSlope = sBias + 32;  // load per register inside the store loop is a trip to L2 on the brick's critical path

  const float* wd = reinterpret_cast<const float*>(Ws_);  // (p.wslice counts 4-byte words)
  const float* bias = bias_;
  float* Y = Y_;
  tr_slice(p, wd, bias, Y);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int col = lane & 31, kh = lane >> 5;
  const int NS = (p.Cin + 3) / 4;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;
  // this workgroup's bricks: step i -> brick i G + (the workgroup's place inside the band of G bricks all workgroups work on
  // together; the workgroups of one XCD, blockIdx % 8, on a contiguous part of the band: neighbours share halo rows in one L2)
  const int G = gridDim.x;
  int place = blockIdx.x;
  if ((G & 7) == 0) place = (place & 7) * (G >> 3) + (place >> 3);
  const int nbr = (int)((p.tiles - place + G - 1) / G);  // bricks of this workgroup (place < tiles)
  auto brick_origin = [&](int i, int& b, int& qz0, int& qy0, int& qx0) {
    unsigned tile = (unsigned)((long long)i * G + place);
    const int txi = (int)(tile % (unsigned)p.tx); tile /= (unsigned)p.tx;
    const int tyi = (int)(tile % (unsigned)p.ty); tile /= (unsigned)p.ty;
    const int tzi = (int)(tile % (unsigned)p.tz);
    b = __builtin_amdgcn_readfirstlane((int)(tile / (unsigned)p.tz));
    qz0 = __builtin_amdgcn_readfirstlane(tzi * T3_TZ);
    qy0 = __builtin_amdgcn_readfirstlane(tyi * T3_TY);
    qx0 = __builtin_amdgcn_readfirstlane(txi * T3_TW);
  };

  if (wave >= NMW) {
#if defined(__HIP_DEVICE_COMPILE__)
    // ---- loader waves: lane = one (staged row, 16-byte piece) item of the brick, all four channels of the stage
    __builtin_amdgcn_s_setprio(3);
    const unsigned* Ws = reinterpret_cast<const unsigned*>(wd);
    const int lw = wave - NMW;
    const int item = lw * 64 + lane;
    const bool act = item < T3_NROW * T3_NQ;
    const int row = act ? item / T3_NQ : T3_NROW, jq = item % T3_NQ;
    const int zr = row / T3_YT, yr = row - zr * T3_YT;
    const unsigned loff = (unsigned)(row * T3_ROWB + jq * 32);
    const unsigned volB = (unsigned)(xvol * 4u);
    // The brick of a stage reaches the conversion through LDS, not through registers: a 16 KB staging area
    // [channel 4][item 256] x 16 bytes that every loader wave fills (LDS-DMA) and reads back for its OWN 64 items.  A first
    // form kept the bricks in flight in inline-assembly register sets (as csrc/convfwd_s3.hpp does): the compiler copied such
    // registers in front of the hand-counted wait whatever the source looked like (tied operands, live-range splits, loop
    // exits), read the stale copy, and handed the original registers to other values -- the lane offsets of the slab copies,
    // conversion temporaries -- which the loads then overwrote when they landed: wrong first bricks on a cold cache, fine on a
    // warm one.  In LDS there is nothing to copy, and the reads are ordinary ds_read the compiler waits for by itself.
    unsigned char* const stg = lds + WOFF + 2 * T3_WB + (lw * 64 + lane) * 16;
    int l_i = 0, l_s = 0, l_b = 0;   // the request stream's (brick, stage) position: NSTG stages ahead of the conversion
    // Two address registers, one per brick parity (`goff` of brick i is written while only copies of brick i - 1 can be in
    // flight, which use the other one): kept from the bring-up, when a corrupted offset register was first read as "a copy
    // reads its address register again after issue" -- scripts/micro/lds_dma_hazards.hip shows it does not; see above
    unsigned goffA = DMA_OOB, goffB = DMA_OOB;
    auto l_brick = [&]() {
      int qz0, qy0, qx0;
      brick_origin(l_i, l_b, qz0, qy0, qx0);
      const int gz = qz0 - 1 + zr, gy = qy0 - 1 + yr, gx = qx0 - 4 + 4 * jq;
      const bool ok = act && gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi && gx >= 0 && gx + 3 < p.Wi;
      const unsigned v = ok ? (((unsigned)gz * (unsigned)p.Hi + (unsigned)gy) * (unsigned)p.Wi + (unsigned)gx) * 4u : DMA_OOB;
      if (l_i & 1) goffB = v; else goffA = v;
    };
    l_brick();
    auto issue_stage = [&](int sbuf) {
      unsigned char* const dst = lds + WOFF + 2 * T3_WB + sbuf * STG1 + lw * 64 * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ch = 4 * l_s + c;
        const int chc = ch < p.Cin ? ch : 0;
        // (inline assembly: through the builtin the compiler selects between the two registers into a TEMPORARY and hands
        // that to the copy -- and re-uses the temporary a few instructions later.  s_nop 4: the resource words may come from
        // v_readfirstlane, and a vector-memory instruction must not read a scalar register within 5 cycles of a vector-ALU
        // write to it -- a hazard the compiler handles for its own instructions, not for text inside an asm statement)
        const unsigned long long a = (unsigned long long)(X + ((size_t)l_b * p.Cin + chc) * xvol);
        t3_i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane(ch < p.Cin ? (int)volB : 0);
        r[3] = 0x00020000;
        const unsigned m0v = (unsigned)(unsigned long long)(lds_ptr_t)(dst + c * 4096);
        if (l_i & 1)
          asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(goffB), "s"(r) : "memory");
        else
          asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(goffA), "s"(r) : "memory");
      }
    };
    auto advance = [&]() {  // the next request position (a new brick's offsets go to the register of ITS parity)
      if (++l_s == NS) { l_s = 0; if (l_i + 1 < nbr) { ++l_i; l_brick(); } }
    };
    // The lane offsets of the slab copies live in NWW registers of their own, written once (kept from the bring-up: with one
    // temporary per copy the slab in LDS came out wrong, the more so the later the copy in the queue -- the temporary's
    // register was one that an in-flight brick load later landed in, see above; the reproducer in scripts/micro/ shows the
    // copies themselves are indifferent to the rewrite)
    unsigned woff[NWW];
#pragma unroll
    for (int k = 0; k < NWW; ++k) {
      woff[k] = (unsigned)(1024 * (lw + NLW * k) + 16 * lane);
      asm volatile("" : "+v"(woff[k]));
    }
    auto issue_wdma = [&](int s, int buf) {
      const unsigned* src = Ws + (size_t)s * T3_WORDS;
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, T3_WB, 0x00020000);
      unsigned char* dst = lds + WOFF + buf * T3_WB;
#pragma unroll
      for (int k = 0; k < NWW; ++k) {
        const int i = lw + NLW * k;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + 1024 * i), 16, woff[k], 0, 0, 0);
      }
    };
    auto convert = [&](int buf, const t3_u32x4 (&ld)[4]) {
      unsigned char* dst = lds + buf * T3_INB + loff;
      const unsigned u[4][4] = {{ld[0].x, ld[0].y, ld[0].z, ld[0].w}, {ld[1].x, ld[1].y, ld[1].z, ld[1].w},
                                {ld[2].x, ld[2].y, ld[2].z, ld[2].w}, {ld[3].x, ld[3].y, ld[3].z, ld[3].w}};
      unsigned h[3][4][2];  // [piece][position][channel pair]
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          float ra = __uint_as_float(u[2 * cp][i]), rc = __uint_as_float(u[2 * cp + 1][i]);
          h[0][i][cp] = s3_pack(ra, rc);
          ra -= __uint_as_float(h[0][i][cp] << 16); rc -= __uint_as_float(h[0][i][cp] & 0xffff0000u);
          asm volatile("" : "+v"(ra), "+v"(rc));  // (keeps the SLP vectoriser from pairing the subtractions)
          h[1][i][cp] = s3_pack(ra, rc);
          ra -= __uint_as_float(h[1][i][cp] << 16); rc -= __uint_as_float(h[1][i][cp] & 0xffff0000u);
          asm volatile("" : "+v"(ra), "+v"(rc));
          h[2][i][cp] = s3_pack(ra, rc);
        }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const t3_u32x4 v0 = {h[pc][0][0], h[pc][0][1], h[pc][1][0], h[pc][1][1]};
        const t3_u32x4 v1 = {h[pc][2][0], h[pc][2][1], h[pc][3][0], h[pc][3][1]};
        *reinterpret_cast<t3_u32x4*>(dst + pc * T3_PIECEB) = v0;
        *reinterpret_cast<t3_u32x4*>(dst + pc * T3_PIECEB + 16) = v1;
      }
    };
    // Iteration g of the stream (stage g % NS of brick g / NS) runs beside the matrix waves' stage g - 1 and ends in barrier
    // #g.  Only LDS-DMA copies are ever in flight here, and they complete in the order they were issued:
    //   vmcnt(0): the staged brick of stage g has landed (requested an iteration ago) -> the wave's items to registers;
    //   the slab of stage g is requested (into the weight buffer stage g - 2 has left), THEN the brick of stage g + 1 (the
    //   staging slots are free: this wave alone reads them, and has); the items are split and parked as stage g's image;
    //   vmcnt(4): the slab is in, the four brick copies behind it may still fly.
    // (NSTG = 2: vmcnt(4) at the top lets the brick of stage g + 1 fly,
    // the brick of stage g + 2 is requested into the staging area just read, and vmcnt(4) at the bottom has the slab and
    // stage g + 1's brick in)
    const int total = nbr * NS;
    issue_stage(0);
    if constexpr (NSTG == 2) {
      if (total > 1) { advance(); issue_stage(1); }
    }
    int s = 0;
    for (int g = 0; g < total; ++g) {
      if (NSTG == 2 && g + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t3_u32x4 ld[4];
      const unsigned char* sp = stg + (NSTG == 2 ? (g & 1) * STG1 : 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) ld[c] = *reinterpret_cast<const t3_u32x4*>(sp + c * 4096);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_wdma(s, g & 1);
      const bool more = g + NSTG < total;
      if (more) {
        advance();
        issue_stage(NSTG == 2 ? (g & 1) : 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef FS_ABLATION
      if (!(p.ab & 32))
#endif
      convert(g & 1, ld);
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (++s == NS) s = 0;
    }
#else
    (void)xvol; (void)NWW;
#endif
    return;
  }

  // ---- matrix waves, 16-row form: wave = (pz, py, z row g), both x parities of three y rows = 12 tiles of 16 positions
  if constexpr (M16) {
    typedef float t3_f32x4 __attribute__((ext_vector_type(4)));
    const int cg = wave & 3, g = wave >> 2;
    const int pz = cg >> 1, py = cg & 1;
    const int n16 = lane & 15, kq = lane >> 4, ay = kq >> 1, ax = kq & 1;
    if (t < 32) {
      sBias[t] = (bias != nullptr && t < p.Cout) ? bias[t] : 0.f;
      sSlope[t] = (p.Z != nullptr) ? p.slope[p.nslope == 1 ? 0 : (t < p.Cout ? t : 0)] : 0.f;
    }
    // input: position (z row g + dz, y row 0 + py - ay, x = n16 + px - ax) of a piece image; weights: slot (kq, co = n16)
    const unsigned bO = (unsigned)(((g + 1) * T3_YT + (1 + py - ay)) * T3_ROWB + (n16 + 4 - ax) * 8);
    const unsigned aO = (unsigned)(WOFF + (cg * 2) * 3 * 1024 + (kq * 16 + n16) * 16);
    const int dz0 = pz * T3_YT * T3_ROWB, dz1 = (pz - 1) * T3_YT * T3_ROWB;  // the two z taps
    const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
    int gs = 0;
    for (int bi = 0; bi < nbr; ++bi) {
      t3_f32x4 acc[2][T3_TY][2];
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int y = 0; y < T3_TY; ++y)
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[px][y][h] = t3_f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < NS; ++s, ++gs) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* sb = lds + (gs & 1) * T3_INB + bO;
        const unsigned char* sw = lds + (gs & 1) * T3_WB + aO;
#ifdef FS_ABLATION
        if (p.ab & 64) continue;
#endif
        // operands of tile i + 1 are read before the MFMAs of tile i (two register sets; the scheduling barriers keep the
        // compiler from hoisting ALL 36 operand reads of a stage to its top: 45 spilled registers)
        auto read_b = [&](int px, int i, t3_bf16x8 (&bq)[3]) {
          const int y = i >> 1, h = i & 1;
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            const unsigned char* q = sb + pc * T3_PIECEB + y * T3_ROWB + (16 * h + px) * 8;
            const t3_u32x2 lo = *(t3_lds_v64)(lds_ptr_t)(q + dz0), hi = *(t3_lds_v64)(lds_ptr_t)(q + dz1);  // (volatile: see the 32-row form)
            const t3_u32x4 v = {lo.x, lo.y, hi.x, hi.y};
            bq[pc] = __builtin_bit_cast(t3_bf16x8, v);
          }
        };
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          t3_bf16x8 a[3], b0[3], b1[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            a[pc] = __builtin_bit_cast(t3_bf16x8, *reinterpret_cast<const t3_u32x4*>(sw + (px * 3 + pc) * 1024));
          read_b(px, 0, b0);
          constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int i = 0; i < 2 * T3_TY; i += 2) {
            read_b(px, i + 1, b1);
            __builtin_amdgcn_sched_barrier(0);
#ifdef FS_ABLATION
            if (!(p.ab & 128))
#endif
#pragma unroll
            for (int q6 = 0; q6 < 6; ++q6)
              acc[px][i >> 1][i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q6]], b0[PB[q6]], acc[px][i >> 1][i & 1], 0, 0, 0);
            if (i + 2 < 2 * T3_TY) read_b(px, i + 2, b0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef FS_ABLATION
            if (!(p.ab & 128))
#endif
#pragma unroll
            for (int q6 = 0; q6 < 6; ++q6)
              acc[px][(i + 1) >> 1][(i + 1) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q6]], b1[PB[q6]], acc[px][(i + 1) >> 1][(i + 1) & 1], 0, 0, 0);
          }
        }
      }
#ifdef FS_ABLATION
      if (p.ab & 4) continue;
#endif
      // ---- epilogue: lane = position n16 of a half row, its four registers = channels 4 kq .. 4 kq + 3; the two x parities
      // of a channel are neighbours in memory (8 bytes), one quad exchange makes 16-byte stores as in the 32-row form
      int b, qz0, qy0, qx0;
      brick_origin(bi, b, qz0, qy0, qx0);
      const int qz = qz0 + g;
      if (qz >= p.Dq) continue;
      const unsigned yv = (unsigned)yvol;
      const size_t sb0 = (size_t)b * p.CoutT * yvol;
      float* const yb = Y + sb0;
      const float* const ab = p.addend != nullptr ? p.addend + (Y - p.Ybase) + sb0 : nullptr;
      float* const zb = p.Z != nullptr ? p.Z + sb0 : nullptr;
      int kqe = kq;
      asm volatile("" : "+v"(kqe));
      const bool evn = (n16 & 1) == 0;
#pragma unroll
      for (int y = 0; y < T3_TY; ++y) {
        const int qy = qy0 + y;
        if (qy >= p.Hq) break;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int qx = qx0 + 16 * h + n16, qxe = evn ? qx : qx - 1;
          const bool xin = qxe < p.Wq;
          const unsigned lo = 4u * (unsigned)kqe * yv + ((unsigned)(2 * qz + pz) * (unsigned)p.Hout + (unsigned)(2 * qy + py)) * (unsigned)p.Wout + 2u * (unsigned)qxe;
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const float a0 = acc[0][y][h][r], a1 = acc[1][y][h][r], b0 = acc[0][y][h][r + 1], b1 = acc[1][y][h][r + 1];
            const float s0 = evn ? b0 : a0, s1 = evn ? b1 : a1;
            const float r0 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s0), 0xB1, 0xF, 0xF, false));
            const float r1 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s1), 0xB1, 0xF, 0xF, false));
            float4 o = evn ? make_float4(a0, a1, r0, r1) : make_float4(r0, r1, b0, b1);
            const int rr = evn ? r : r + 1;
            const int co = 4 * kqe + rr;
            if (co >= p.Cout || !xin) continue;
            const unsigned lo2 = lo + (unsigned)rr * yv;
            const float bv = sBias[co];
            o.x += bv; o.y += bv; o.z += bv; o.w += bv;
            if (ab != nullptr) {
              const float4 a4 = *reinterpret_cast<const float4*>(ab + lo2);
              o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
            }
            *reinterpret_cast<float4*>(yb + lo2) = o;
            if (zb != nullptr) {
              const float sl = sSlope[co];
              *reinterpret_cast<float4*>(zb + lo2) = make_float4(o.x > 0.f ? o.x : sl * o.x, o.y > 0.f ? o.y : sl * o.y,
                                                                 o.z > 0.f ? o.z : sl * o.z, o.w > 0.f ? o.w : sl * o.w);
            }
          }
        }
      }
    }
    return;
  }
  // ---- matrix waves: wave = (pz, py, z row g); both x parities of the three y rows
  const int wv = wave;
  if (t < 32) {  // (visible to every wave behind the first stage's barrier)
    sBias[t] = (bias != nullptr && t < p.Cout) ? bias[t] : 0.f;
    sSlope[t] = (p.Z != nullptr) ? p.slope[p.nslope == 1 ? 0 : (t < p.Cout ? t : 0)] : 0.f;
  }
  const int cg = wv & 3, g = wv >> 2;
  const int pz = cg >> 1, py = cg & 1;
  // tap a of parity par: input offset d = par - a (tap_d), kernel index 1 - par + 2 a (tap_k)
  // input: byte offset of position (z row g + dz, y row 0 + dy(kh), x col) in a piece image; weights: slot (kh, col)
  const unsigned bO = (unsigned)(((g + 1) * T3_YT + (1 + py - kh)) * T3_ROWB + (col + 4) * 8);
  const unsigned aO = (unsigned)(WOFF + (cg * 2) * 2 * 3 * 1024 + (kh * 32 + col) * 16);
  const size_t yvol = (size_t)p.Dout * p.Hout * p.Wout;
  int gs = 0;  // stage stream position (buffers: gs & 1)
  for (int bi = 0; bi < nbr; ++bi) {
    f32x16 acc[2][T3_TY];
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int y = 0; y < T3_TY; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[px][y][r] = 0.f;
    for (int s = 0; s < NS; ++s, ++gs) {
      __builtin_amdgcn_s_barrier();  // stage gs is ready (and every wave has left stage gs - 1)
      const unsigned char* sb = lds + (gs & 1) * T3_INB + bO;
      const unsigned char* sw = lds + (gs & 1) * T3_WB + aO;
#ifdef FS_ABLATION
      if (p.ab & 64) continue;
#endif
#pragma unroll
      for (int az = 0; az < 2; ++az) {
        const int dzr = (pz - az) * T3_YT * T3_ROWB;  // (wave-uniform)
        t3_bf16x8 a[2][3];
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            a[px][pc] = __builtin_bit_cast(t3_bf16x8, *reinterpret_cast<const t3_u32x4*>(sw + ((px * 2 + az) * 3 + pc) * 1024));
#pragma unroll
        for (int y = 0; y < T3_TY; ++y) {
          t3_u32x2 P[3][3];  // [piece][position x - 1, x, x + 1]
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int d = 0; d < 3; ++d)
              // (volatile: the compiler would pair two of these into one ds_read2_b64, which is banked on 32 banks -- 2-way
              // conflicts on 256 contiguous bytes -- where ds_read_b64 uses all 64)
              P[pc][d] = *(t3_lds_v64)(lds_ptr_t)(sb + dzr + pc * T3_PIECEB + y * T3_ROWB + (d - 1) * 8);
          t3_bf16x8 bq[2][3];
#pragma unroll
          for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
              // x taps of parity px: offsets (px - 1, px) in INCREASING position order -> positions [px], [px + 1]: the operand
              // is four consecutive registers of P (the slab stores the x taps in that order: slot 0 = tap 1, slot 1 = tap 0)
              const t3_u32x4 v = {P[pc][px].x, P[pc][px].y, P[pc][px + 1].x, P[pc][px + 1].y};
              bq[px][pc] = __builtin_bit_cast(t3_bf16x8, v);
            }
#ifdef FS_ABLATION
          if (p.ab & 128) continue;
#endif
          // the six products, small terms first: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
          constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int px = 0; px < 2; ++px)
              acc[px][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[px][PA[q]], bq[px][PB[q]], acc[px][y], 0, 0, 0);
        }
      }
    }

#ifdef FS_ABLATION
    if (p.ab & 4) continue;
#endif
    // ---- epilogue: the lane's two x-neighbouring outputs of a channel leave as one 8-byte store.  Addresses = a wave-uniform
    // 64-bit base per (sample, channel of the register) + ONE 32-bit lane offset per row (per-lane 64-bit addresses for all 16
    // registers x 3 tensors were hoisted out of the brick loop and spilled)
    int b, qz0, qy0, qx0;
    brick_origin(bi, b, qz0, qy0, qx0);
    const int qz = qz0 + g, qx = qx0 + col;
    if (qz >= p.Dq) continue;
    const unsigned yv = (unsigned)yvol;
    const size_t sb0 = (size_t)b * p.CoutT * yvol;
    float* const yb = Y + sb0;
    const float* const ab = p.addend != nullptr ? p.addend + (Y - p.Ybase) + sb0 : nullptr;
    float* const zb = p.Z != nullptr ? p.Z + sb0 : nullptr;
    // (the lane half as an opaque value: everything derived from it -- sixteen per-lane 64-bit bias / slope addresses -- was
    // otherwise hoisted out of the brick loop and spilled)
    int khe = kh;
    asm volatile("" : "+v"(khe));
    const unsigned khv = 4u * (unsigned)khe * yv;
    // 16-byte stores (the epilogue is bound by store wave-instructions, ~75 cycles each whatever their width): neighbouring
    // lanes (positions qx, qx + 1) swap one register's pair -- DPP quad_perm [1,0,3,2] -- so that the even lane holds the
    // four consecutive outputs 2 qx .. 2 qx + 3 of channel row r, its odd neighbour those of row r + 1
    const bool evn = (col & 1) == 0;
    const int qxe = evn ? qx : qx - 1;   // position of the even lane of the pair
    const bool xin = qxe < p.Wq;         // (Wq is even: the pair is inside or outside as a whole)
#pragma unroll
    for (int y = 0; y < T3_TY; ++y) {
      const int qy = qy0 + y;
      if (qy >= p.Hq) break;
      const unsigned lo = khv + ((unsigned)(2 * qz + pz) * (unsigned)p.Hout + (unsigned)(2 * qy + py)) * (unsigned)p.Wout + 2u * (unsigned)qxe;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        // rows of the two registers: r -> cr, r + 1 -> cr + 1 (+ 4 kh, in `lo`)
        const int cr = (r & 3) + 8 * (r >> 2);
        const float a0 = acc[0][y][r], a1 = acc[1][y][r], b0 = acc[0][y][r + 1], b1 = acc[1][y][r + 1];
        const float s0 = evn ? b0 : a0, s1 = evn ? b1 : a1;
        const float r0 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s0), 0xB1, 0xF, 0xF, false));
        const float r1 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s1), 0xB1, 0xF, 0xF, false));
        float4 o = evn ? make_float4(a0, a1, r0, r1) : make_float4(r0, r1, b0, b1);
        const int crr = evn ? cr : cr + 1;
        const int co = crr + 4 * khe;
        if (co >= p.Cout || !xin) continue;
        const size_t ro = (size_t)cr * yvol;   // (wave-uniform; the odd lane's + yvol is a lane offset)
        const unsigned lo2 = lo + (evn ? 0u : yv);
        const float bv = sBias[co];
        o.x += bv; o.y += bv; o.z += bv; o.w += bv;
        if (ab != nullptr) {
          const float4 a4 = *reinterpret_cast<const float4*>(ab + ro + lo2);
          o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
        }
        *reinterpret_cast<float4*>(yb + ro + lo2) = o;
        if (zb != nullptr) {
          const float sl = sSlope[co];
          *reinterpret_cast<float4*>(zb + ro + lo2) = make_float4(o.x > 0.f ? o.x : sl * o.x, o.y > 0.f ? o.y : sl * o.y,
                                                                  o.z > 0.f ? o.z : sl * o.z, o.w > 0.f ? o.w : sl * o.w);
        }
      }
    }
  }
}
