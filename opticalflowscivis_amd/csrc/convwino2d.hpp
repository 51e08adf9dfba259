// convwino2d.hpp -- the 64-channel k3 s1 "same" convolutions of the 64^3 IFNet-3D trunk with a 2-D Winograd transform:
// F(4,3) along x (convwino4.hpp) and F(2,3) along y.  A tile of 2 y x 4 x outputs takes 4 x 6 = 24 products per (ci, kz)
// instead of 72: ONE THIRD of the direct form's matrix-core work (1-D F(4,3): half).  Included inside convfwd.hip's
// anonymous namespace, after convwino4.hpp.
//
//   U[ci][kz][ty][tx][co] = sum_{ky, kx} Gy[ty][ky] Gx[tx][kx] g[co][ci][kz][ky][kx]          (weight re-layout)
//   V[ci][z][ty][tx][j]   = By^T (4 input rows y0-1 .. y0+2)  then  Bx^T (6 input columns 4j-1 .. 4j+4)
//   M[ty][tx][co][z][j]   = sum_{ci, kz} U[ci][kz][ty][tx][co] V[ci][z + kz - 1][ty][tx][j]      24 GEMMs, K = 3 Cin
//   y[co][z][y0 + 0..1][4j + 0..3] = Ay^T Ax^T M
//   By^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3),  Gy = rows (1,0,0), (1,1,1)/2, (1,-1,1)/2, (0,0,1),
//   Ay^T m = (m0 + m1 + m2, m1 - m2 - m3);  the x matrices are convwino4.hpp's.
//
// Kernel: loader-wave form, one 8-wave workgroup per CU, brick = 2 z x 2 y x 64 x (MFMA column = (z row, x-tile)), chunks of
// 2 input channels (one 32x32x2 k-pair; the filter slab is 18 KB per channel), THREE LDS buffers of 49 KB: a chunk lasts
// only ~2 300 MFMA cycles, so the slab is requested two chunks ahead and the input rows one chunk ahead of their transform.
//   loader waves 4-7: the transformed input (a lane owns one x-tile of one staged z row with all four y rows -- 4 float4
//     loads; 16 lanes per z row, the four z rows of the brick in one wave: By^T in registers, the neighbouring columns of
//     the y-transformed rows by DPP row shifts (halo columns by dword loads in lanes 0 / 15), Bx^T, 24 dword writes) and
//     the U slab of the chunk by `buffer_load_dwordx4 ... lds` (36 wave-instructions), both dealt evenly to the four
//     waves and to the periods (see the loader code).
//   matrix waves 0-3: one y-component ty each: 6 x-components x 2 channel tiles = 12 accumulator tiles (192 VGPRs).
//   epilogue: Ax^T in registers (4 consecutive x per lane and channel), the four waves' results meet in LDS (128 KB of
//     the idle staging buffers), every wave finishes 16 channels: Ay^T, the direct kernel's fused epilogues, 16-byte stores.
constexpr int W2_ZP = 4 * 96 + 16;            // floats per staged z row: [ty][tx][x-tile 16] (+16: the two z rows of an MFMA
                                              // operand read land on different halves of the 32 banks)
constexpr int W2_VCH = 4 * W2_ZP;             // channel pitch of V
constexpr int W2_UCH = FS_WINO2D_UCH;         // channel pitch of U: [kz][ty][tx][co] = 3 * 24 * 64

// DBG (measurement builds, FLOWSCI_WINO_DBG): 1 = the U DMA is skipped, 2 = the input waves skip loads and transforms,
// 3 = both (matrix waves + epilogue alone), 4 = the matrix waves skip their operand reads and MFMAs (loaders alone).
// XT = x-tiles per row of a brick: 16 (rows of 64 x, brick 2 z x 2 y x 64 x) or 8 (rows of 32 x: the 16 x-tile slots of a z
// row are then 2 y-tiles x 8 x-tiles, brick 2 z x 4 y x 32 x -- the 32^3 trunk of the scale-2 block).
//
// Round 4: the kernel rebuilt around what cycle stamps inside it showed (scripts/w2_stamps.py, `make w2stamps`;
// profiles/r04_wino2d_stamps.txt).  The round-3 form (kept in the ablation build, FLOWSCI_WINO2D_R3=1) ran one workgroup per
// brick, 0.434 ms per launch, matrix pipe 0.64 busy; round 3 blamed the loaders' vector-ALU work, round 4 first the fill rate.
// Neither: (1) its loader waves waited, every second period, for the nine slab requests they had JUST issued -- the
// compiler's wait-count pass put `s_waitcnt vmcnt(0)` in front of the first use of the loop-carried row registers --, (2) a
// period ended when the slowest loader arrived, and its work came after its waits, (3) every brick paid ~8-10 us of fixed
// time around its 32 periods (workgroup launch, 192 accumulators zeroed, first slab and rows with nothing to overlap, a
// 128 KB exchange, a global load per channel inside the store loop, the last stores drained before the next workgroup).
//   * PERSISTENT workgroups (one per CU): a workgroup does a run of bricks and its loader waves never stop -- the chunks of
//     ALL its bricks are one stream through the three staging buffers, so the first chunks of brick i + 1 are fetched,
//     transformed and staged while the matrix waves finish brick i.
//   * loader waves: row loads and slab copies are INLINE ASSEMBLY with hand-counted waits (rows requested three periods
//     ahead into two register sets; a period's transform work first, the wait for the previous period's slab last); the
//     transforms in packed fp32 arithmetic on columns loaded -- not shifted -- from the neighbours (55 -> 28 vector
//     instructions per period: with the SIMD's matrix wave issuing MFMAs a loader gets an instruction through every ~16
//     cycles), two y components per period; a wave's slab quarter is contiguous (three M0 values per nine copies).
//   * matrix wave = (channel tile m, y-component pair typ): components ty = 2 typ, 2 typ + 1, all six tx -- 12 accumulator
//     tiles as before, operands of two steps per LDS read (`ds_read2_b32`: the LDS instruction stream of the matrix waves
//     is what the loaders' LDS traffic queues behind), and Ax^T AND most of Ay^T stay in registers: y row 0 =
//     (X0 + X1) + X2 is finished by the typ = 0 wave, y row 1 = (X1 - X2) - X3 by the typ = 1 wave (X_ty = Ax^T M_ty; same
//     operations in the same order as the round-3 kernel: BIT-IDENTICAL outputs, scripts/wino2d_ab.py), each needs ONE
//     x-transformed component of its partner: 64 KB through LDS instead of 128 KB written + 192 KB read, in two rounds of
//     32 KB that fit the staging buffer the brick's last chunk has just left (the other two hold the next brick's chunks).
//   * first chunk of a brick: MFMAs with C = 0 instead of zeroed accumulators; bias / slopes live in LDS; addend / act_y of
//     a round are requested before its exchange; every global access of the epilogue is a buffer operation with ONE 32-bit
//     lane offset (no 64-bit address per channel: they spilled), 16 bytes per lane and channel straight from the MFMA layout.
// Per launch (2 x 64 x 64^3, same box, scripts/wino2d_ab.py): 0.433 vs 0.464 ms plain, 0.466 vs 0.517 with PReLU +
// residual, 0.479 vs 0.529 with the PReLU-backward epilogue.  Stamps now: MFMA loop 2 750 cycles per period (36 MFMAs =
// 2 304) + 180 at the barrier, loaders 1 450 work + 500-800 waits; the epilogue 9 500 cycles per brick (32 periods).
#ifdef FS_W2_STAMPS  // s_memtime sums per wave of workgroup 3 (scripts/w2_stamps.py; a build of its own, never shipped)
__device__ unsigned long long fs_w2_dbg[8 * 8];
#define W2_T() __builtin_amdgcn_s_memtime()
#define W2_ACC(k, t0) do { const unsigned long long t_ = W2_T(); w2s[k] += t_ - (t0); (t0) = t_; } while (0)
#else
#define W2_ACC(k, t0) do { } while (0)
#endif
template <int DBG, int XT>
__global__ __launch_bounds__(512, 1) void conv3d_wino2d_ps_kernel(const float* __restrict__ X,
                                                                 const float* __restrict__ Ut,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ Y, FP p) {
  constexpr int CI = 2;
  constexpr int NV = CI * W2_VCH, NU = CI * W2_UCH;
  constexpr int NUP = NU / 256;            // LDS-DMA wave-instructions (64 x 16 bytes) of the U slab
  constexpr int NUW = NUP / 4;             // per loader wave
  constexpr int BUF = NV + NU;
  constexpr int NB = 3;                    // staging buffers: the U slab is requested TWO chunks ahead (a chunk is only ~2 300
                                           // MFMA cycles long -- less than a trip to L2 and back)
  constexpr int NEXR = 4 * 8 * 64 * 4;     // one exchange round: [wave][8 registers][lane] float4
  constexpr int NCV = 3 * 64;              // per-channel vectors behind the buffers: bias, PReLU slope, PReLU-backward slope
  static_assert(NU % 256 == 0 && NV % 4 == 0 && (NB * BUF + NCV) * 4 <= 160 * 1024 && NEXR <= BUF, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[NB * BUF + NCV];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;

  // this workgroup's run of bricks: contiguous, and the runs of the workgroups of one XCD (blockIdx % 8) adjacent
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const long long per = p.tiles / nwg, rem = p.tiles % nwg;
  const long long s0 = wg * per + (wg < rem ? wg : rem);
  const int nb = (int)per + (wg < rem ? 1 : 0);
  if (nb == 0) return;
  // Which brick a workgroup does in its step t.  The bricks are rows (y-tiles) x slabs (z-tiles, samples) of a grid; every
  // brick re-reads half of its y neighbour's and half of its z neighbour's input rows, and an XCD's 32 workgroups together
  // stream ~8 MB per step through a 4 MB L2 -- a halo shared with the workgroup's OWN next brick is gone from L2 by then
  // (PMC: 3.8x the input read from HBM).  So where the shape allows it the 32 workgroups of an XCD work on a compact patch
  // of 4 z-tiles x 8 y-tiles at the same time (halos inside the patch are read together, by two workgroups at once) and the
  // patch moves through the XCD's part of the grid; otherwise a workgroup walks its contiguous run.
  const int xw = nwg >> 3;                                  // workgroups per XCD
  const bool patch = (nwg & 7) == 0 && xw == 32 && rem == 0 && p.tx == 1 && (p.ty & 7) == 0 && (32 * nb) % p.ty == 0 &&
                     ((32 * nb) / p.ty) % 4 == 0;
  const int pj = wg % 32;                                    // index inside the XCD (wg is already XCD-major)
  const long long xz0 = (long long)(wg / 32) * ((32 * nb) / (patch ? p.ty : 1));   // first z-tile of the XCD's part
  const int ysteps = patch ? p.ty / 8 : 1;
  auto brick_of = [&](int t) -> long long {
    if (!patch) return s0 + t;
    const int sz = t / ysteps, sy = t - sz * ysteps;
    return (xz0 + sz * 4 + (pj >> 3)) * p.ty + sy * 8 + (pj & 7);
  };
  const int nch = p.Cin / CI;              // even (Cin % 4 == 0): a chunk's parity is the same in every brick
  const int G = nb * nch;                  // the chunk stream: chunk g = channel pair g % nch of brick s0 + g / nch, buffer g % 3
  constexpr int YT = 16 / XT;  // y-tiles of a brick
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    // ---- loader waves.  The vector-ALU work of a loader wave is NOT hidden behind the matrix wave it shares a SIMD with
    // (fp32 MFMAs run on the same lanes), so it is spread evenly over all four SIMDs and all periods: a wave handles input
    // channel cc = lw & 1 of every second chunk (parity lw >> 1) in two half-steps -- y components 0, 1 in one period, 2, 3
    // in the next -- and a quarter of every chunk's U slab.  Chunk g is read by the matrix waves in period g and is complete
    // at the end of period g - 1:
    //   period g - 3: its rows are requested (after the wave's previous chunk is finished)
    //   period g - 2: By^T, then Bx^T of ty 0, 1 -> buffer g % 3      period g - 1: Bx^T of ty 2, 3
    const int lw = wave - 4, par = lw >> 1, cc = lw & 1;
    const int zr = lane >> 4, q = lane & 15;  // lane = (staged z row, x-tile)
#ifdef FS_ABLATION
    if (p.ab & 1) __builtin_amdgcn_s_setprio(3);
#else
    __builtin_amdgcn_s_setprio(3);
#endif
    const int qx = q & (XT - 1), qy = q / XT;   // x-tile, y-tile of the lane's slot
    // the rows of the brick the wave's NEXT fetch belongs to
    unsigned voff[4], loff[4], roff[4];
    int fbat = 0;
    auto brick_rows = [&](long long brick) {
      // (32-bit: tiles < 2^31; readfirstlane: the division is expanded on the vector ALU, and a buffer resource built from
      // a vector register is applied lane by lane)
      unsigned tile = (unsigned)brick;
      const int txi = __builtin_amdgcn_readfirstlane((int)(tile % (unsigned)p.tx)); tile /= (unsigned)p.tx;
      const int tyi = __builtin_amdgcn_readfirstlane((int)(tile % (unsigned)p.ty)); tile /= (unsigned)p.ty;
      const int tzi = __builtin_amdgcn_readfirstlane((int)(tile % (unsigned)p.tz));
      fbat = __builtin_amdgcn_readfirstlane((int)(tile / (unsigned)p.tz));
      const int oz0 = tzi * 2, oy0 = tyi * 2 * YT, ox0 = txi * 4 * XT;
      const int gz = oz0 - 1 + zr, gx = ox0 + 4 * qx;
#pragma unroll
      for (int yr = 0; yr < 4; ++yr) {
        const int gy = oy0 + 2 * qy - 1 + yr;
        const bool rowok = gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
        const unsigned rbase = ((unsigned)(rowok ? gz : 0) * p.Hi + (rowok ? gy : 0)) * p.Wi;
        voff[yr] = (rowok && gx < p.Wi) ? (rbase + gx) * 4u : DMA_OOB;   // Wi % (4 XT) == 0: a float4 is in or out whole
        loff[yr] = (rowok && gx > 0) ? (rbase + gx - 1) * 4u : DMA_OOB;  // columns 4 j - 1 and 4 j + 4 (zero padding: out of range)
        roff[yr] = (rowok && gx + 4 < p.Wi) ? (rbase + gx + 4) * 4u : DMA_OOB;
      }
    };
    brick_rows(brick_of(0));
    int f_c = par;                 // channel pair of the wave's next fetch (its chunks: every second one)
    int f_t = 0;                   // ... and the step (brick of the workgroup) it belongs to
    const int vdst = cc * W2_VCH + zr * W2_ZP + q;
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    // The filter slab: loader wave lw copies the quarter [lw * 9 KB, + 9 KB) of chunk g + 2's 36 KB slab to LDS at the top of
    // period g -- nine 1 KB `buffer_load_dwordx4 ... lds` in three groups of four that share M0 and the lane offset (the
    // instruction's 12-bit offset applies to both sides of the copy): 15 instructions instead of 36.  Inline assembly like
    // the row loads: the loader waves' vector-memory traffic is counted by hand (see the waits below).
    i32x4_t rU;
    {
      const unsigned long long ua = (unsigned long long)Ut;
      rU[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ua);
      rU[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ua >> 32) & 0xffffu));
      rU[2] = 0x7fffffff;
      rU[3] = 0x00020000;
    }
    static_assert(NU * 4 == 4 * 9216, "a quarter of the slab is nine 1 KB pieces");
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_ptr_t)lds;
    const unsigned uvoff = (unsigned)lw * 9216u + (unsigned)lane * 16u;
    int u_c = 0, u_buf = 0;        // channel pair and buffer of the next slab request (every chunk, in order)
    auto dma_u = [&]() {
      if (DBG != 1 && DBG != 3) {
        const unsigned m0v = lds0 + (unsigned)(u_buf * BUF + NV) * 4u + (unsigned)lw * 9216u;
        const unsigned so = (unsigned)u_c * (unsigned)(NU * 4);
        asm volatile(
            "s_mov_b32 m0, %0\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
            "buffer_load_dwordx4 %3, %4, %5 offen offset:1024 lds\n\t"
            "buffer_load_dwordx4 %3, %4, %5 offen offset:2048 lds\n\t"
            "buffer_load_dwordx4 %3, %4, %5 offen offset:3072 lds\n\t"
            "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %3, %4, %6 offen lds\n\t"
            "buffer_load_dwordx4 %3, %4, %6 offen offset:1024 lds\n\t"
            "buffer_load_dwordx4 %3, %4, %6 offen offset:2048 lds\n\t"
            "buffer_load_dwordx4 %3, %4, %6 offen offset:3072 lds\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %3, %4, %7 offen lds"
            :
            : "s"(m0v), "s"(m0v + 4096u), "s"(m0v + 8192u), "v"(uvoff), "s"(rU), "s"(so), "s"(so + 4096u), "s"(so + 8192u)
            : "memory");
      }
      u_c = u_c + 1 == nch ? 0 : u_c + 1;
      u_buf = u_buf == NB - 1 ? 0 : u_buf + 1;
    };
    // The raw rows in flight: TWO register sets, a chunk's rows are requested three periods before By^T reads them (they come
    // from HBM: requested one period ahead, as in round 3, they were not there in time).  They are loaded and awaited by
    // INLINE ASSEMBLY: the compiler's own wait-count insertion treats a loop-carried load conservatively -- with plain loads
    // it put `s_waitcnt vmcnt(0)` in front of By^T, i.e. the wave waited for the nine slab requests it had issued a moment
    // before, a full trip to L2 in every second period (the round-3 kernel had the same stall).  The asm statements hide
    // the loads from that pass; the counted wait names the registers as in/out operands, so nothing that reads them can
    // move above it; every fetch is executed by both parities (masked when there is nothing to fetch) and the loop leaves
    // by `break`, so that no join -- and no register copy -- sits between a request and its wait.
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    // a lane's six columns of a row: its four (one 16-byte load) and the neighbours 4 j - 1, 4 j + 4 (two 4-byte loads that
    // hit the lines the 16-byte loads of the same instruction group bring in) -- no cross-lane shifts, no seam logic
    f32x4_t xr[2][4];
    float xl[2][4], xg[2][4];
    // (one fetch = 12 vector-memory instructions: the counted waits below depend on it)
    auto fetch = [&](auto SET, bool want) {   // want (wave-uniform): the wave has another chunk to fetch
      constexpr int set = decltype(SET)::value;
      if (DBG == 2 || DBG == 3) return;
      const unsigned long long a = (unsigned long long)(X + ((size_t)fbat * p.Cin + (f_c * CI + cc)) * xvol);  // (f_c * CI + cc < Cin)
      i32x4_t r;
      r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
      r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
      r[2] = want ? (int)((unsigned)xvol * 4u) : 0;   // zero records: every lane out of range, nothing is read
      r[3] = 0x00020000;
#pragma unroll
      for (int yr = 0; yr < 4; ++yr) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(xr[set][yr]) : "v"(voff[yr]), "s"(r));
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(xl[set][yr]) : "v"(loff[yr]), "s"(r));
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(xg[set][yr]) : "v"(roff[yr]), "s"(r));
      }
      if (want) {
        f_c += 2;
        if (f_c >= nch) {            // the wave's next chunk is in the next brick
          f_c -= nch;
          ++f_t;
          if (f_t < nb) brick_rows(brick_of(f_t));
        }
      }
    };
    // wait until at most N vector-memory operations of this wave are outstanding; the rows of set S are operands (see above)
#define W2_WAIT_ROWS(S, N)                                                                                                  \
  asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                    \
               : "+v"(xr[S][0]), "+v"(xr[S][1]), "+v"(xr[S][2]), "+v"(xr[S][3]), "+v"(xl[S][0]), "+v"(xl[S][1]), "+v"(xl[S][2]), \
                 "+v"(xl[S][3]), "+v"(xg[S][0]), "+v"(xg[S][1]), "+v"(xg[S][2]), "+v"(xg[S][3])                             \
               :                                                                                                             \
               : "memory")
    // The transforms in PACKED fp32 arithmetic (two values per instruction): while the SIMD's matrix wave issues MFMAs a
    // loader wave gets about one vector instruction in per MFMA (cycle stamps, scripts/w2_stamps.py: 55 instructions per
    // period against 36 MFMAs made the loaders the critical path, the matrix waves waited ~800 of 3 300 cycles per
    // period), so what counts is the NUMBER of vector instructions: 29 per period in this form.  Same operations on
    // the same operands as the scalar form: bit-identical results.
    f32x2_t ya[4], yb[4];    // By^T of the chunk in work: columns (d1, d2) and (d3, d4) of the four components ...
    float yl[4], yg[4];      // ... and d0, d5 (lives across the two half-steps)
// (inline assembly is safe HERE because these results go to LDS: a packed result read by an MFMA needs wait states that only
// the compiler's hazard pass inserts -- an asm version of the weight-gradient kernel's operand transforms computed wrong
// sums, and its compiler-generated packed form gained nothing: convwrwwino4.hpp stays scalar)
#define W2_PK_ADD(d, x, y) asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define W2_PK_SUB(d, x, y) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y))
    auto ytrans = [&](auto SET) {
      constexpr int set = decltype(SET)::value;
      if (DBG == 2 || DBG == 3) return;
      f32x2_t lo[4], hi[4];
#pragma unroll
      for (int yr = 0; yr < 4; ++yr) {
        lo[yr] = __builtin_shufflevector(xr[set][yr], xr[set][yr], 0, 1);
        hi[yr] = __builtin_shufflevector(xr[set][yr], xr[set][yr], 2, 3);
      }
      W2_PK_SUB(ya[0], lo[0], lo[2]); W2_PK_ADD(ya[1], lo[1], lo[2]); W2_PK_SUB(ya[2], lo[2], lo[1]); W2_PK_SUB(ya[3], lo[1], lo[3]);
      W2_PK_SUB(yb[0], hi[0], hi[2]); W2_PK_ADD(yb[1], hi[1], hi[2]); W2_PK_SUB(yb[2], hi[2], hi[1]); W2_PK_SUB(yb[3], hi[1], hi[3]);
      yl[0] = xl[set][0] - xl[set][2]; yl[1] = xl[set][1] + xl[set][2]; yl[2] = xl[set][2] - xl[set][1]; yl[3] = xl[set][1] - xl[set][3];
      yg[0] = xg[set][0] - xg[set][2]; yg[1] = xg[set][1] + xg[set][2]; yg[2] = xg[set][2] - xg[set][1]; yg[3] = xg[set][1] - xg[set][3];
    };
    const f32x2_t c44 = {-4.f, 4.f}, c22 = {2.f, -2.f};
    // Bx^T of the y components [R0, R0 + NR) of the chunk in work -> buffer `buf`.  The first component goes with By^T in the
    // chunk's first period, the other three in its second: ~27 vector instructions per period either way.
    auto put_rows = [&](int buf, auto R0c, auto NRc) {
      if (DBG == 2 || DBG == 3) return;
      constexpr int R0 = decltype(R0c)::value, NR = decltype(NRc)::value;
      float* dstb = lds + buf * BUF + vdst;
#pragma unroll
      for (int ty = R0; ty < R0 + NR; ++ty) {
        const f32x2_t A = ya[ty], B = yb[ty];   // (d1, d2), (d3, d4)
        const float d0 = yl[ty], d5 = yg[ty];
        f32x2_t PR, S, T, O12, O34;
        W2_PK_SUB(PR, B, A);                                                                             // (p31, r42) = (d3 - d1, d4 - d2)
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(S) : "v"(A));   // (d1 + d2, d1 - d2)
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(T) : "v"(B));   // (d4 + d3, d4 - d3)
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(O12) : "s"(c44), "v"(S), "v"(T));                     // -4 (d1 + d2) + (d3 + d4),  4 (d1 - d2) + (d4 - d3)
        asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(O34) : "s"(c22), "v"(PR));  // 2 p31 + r42,  -2 p31 + r42
        float* dst = dstb + ty * 96;
        dst[0 * 16] = fmaf(4.f, d0, fmaf(-5.f, A.y, B.y));
        dst[1 * 16] = O12.x;
        dst[2 * 16] = O12.y;
        dst[3 * 16] = O34.x;
        dst[4 * 16] = O34.y;
        dst[5 * 16] = fmaf(4.f, A.x, fmaf(-5.f, B.x, d5));
      }
    };
    const std::integral_constant<int, 0> I0{};
    const std::integral_constant<int, 2> I2{};
    static_assert(NUP == 4 * NUW, "every loader wave issues NUW slab instructions per chunk");
    // prologue = "period -1".  Parity 0 (even chunks): chunk 0 whole and the first half of chunk 2 (buffer 2 is idle until
    // period 2), chunk 4's rows requested.  Parity 1: the first half of chunk 1, chunk 3's rows requested.  Slabs 0 and 1.
    const std::integral_constant<int, 0> S0{};
    const std::integral_constant<int, 1> S1{};
    constexpr bool NODMA = DBG == 1 || DBG == 3;
    dma_u();
    if (G > 1) dma_u();
    fetch(S0, par == 0 || G > 1);            // chunk `par`
    W2_WAIT_ROWS(0, 0);
    ytrans(S0);
    put_rows(par, I0, I2);                   // (parity 1 with a single chunk: zeros into the idle buffer 1)
    if (par == 0) put_rows(0, I2, I2);
    fetch(S0, par == 0 && G > 2);            // parity 0: chunk 2
    W2_WAIT_ROWS(0, 0);
    if (par == 0) {
      ytrans(S0);
      put_rows(2, I0, I2);                   // (a single pair of chunks: zeros into the idle buffer 2)
    }
    fetch(S1, par + 4 - 2 * par < G);        // parity 0: chunk 4, parity 1: chunk 3 -- consumed by the first "b" period
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int pb = 0, pc = 0;            // g0 % 3, g0 % nch
#ifdef FS_W2_STAMPS
    unsigned long long w2s[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 0 waits (vmcnt), 1 work, 2 LDS drain, 3 period barrier, 4 exchange barriers
    unsigned long long w2t = W2_T();
#endif
    // the end of period g0: chunk g0 + 1 is in LDS, the matrix waves are done reading chunk g0; when a brick ends, the
    // matrix waves' exchange rounds follow (they use the buffer chunk g0 has left)
    auto period_end = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W2_ACC(2, w2t);
      __builtin_amdgcn_s_barrier();
      W2_ACC(3, w2t);
      pb = pb == NB - 1 ? 0 : pb + 1;
      if (++pc == nch) {
        pc = 0;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        W2_ACC(4, w2t);
      }
    };
    // What a period costs (cycle stamps, scripts/w2_stamps.py): while the matrix wave of its SIMD issues MFMAs, a loader
    // wave gets one instruction through every ~16 cycles (7 alone), and a wave's nine slab requests plus their wait take
    // ~700 cycles wherever they are issued (moved to the matrix waves they lengthen the MFMA loop by as much).  The
    // period is the loaders' once their instruction stream exceeds the 36 MFMAs' ~2 500 cycles: round 3's ~170
    // instructions with the waits in front took 3 500.  Hence: requests first, the transform work next (packed arithmetic,
    // neighbour columns loaded instead of shifted in, the four y components split 1 + 3 over a chunk's two periods: ~850
    // cycles), the wait for the PREVIOUS period's slab last.
    // a period in which the wave finishes a chunk (components 1-3 of chunk g0 + 1) and requests rows (chunk g0 + 5, set SET)
    auto period_a = [&](int g0, auto SET) {
      const bool req = g0 + 2 < G;
      if (req) dma_u();
      if (g0 + 1 < G) put_rows(pb == NB - 1 ? 0 : pb + 1, I2, I2);
      fetch(SET, g0 + 5 < G);
      W2_ACC(1, w2t);
      // the previous period's slab requests have landed: outstanding may be this period's (9) and the twelve row loads
      if (req && !NODMA) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      W2_ACC(0, w2t);
      period_end();
    };
    // a period in which the wave starts a chunk: By^T of rows requested three periods ago (set SET), component 0 of chunk
    // g0 + 2.  Issued after those rows: 9 + (9 + 12) + 9 requests (fewer in the first periods: everything is awaited there).
    auto period_b = [&](int g0, auto SET) {
      constexpr int set = decltype(SET)::value;
      const bool req = g0 + 2 < G;
      if (req) {
        dma_u();
        if (g0 < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!NODMA) { if (set == 0) W2_WAIT_ROWS(0, 39); else W2_WAIT_ROWS(1, 39); }
        else { if (set == 0) W2_WAIT_ROWS(0, 12); else W2_WAIT_ROWS(1, 12); }
        W2_ACC(0, w2t);
        ytrans(SET);
        put_rows(pb == 0 ? NB - 1 : pb - 1, I0, I2);
        W2_ACC(1, w2t);
        // the previous period's slab requests have landed: outstanding may be its row loads (12) and this period's requests (9)
        if (!NODMA) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail: nothing left to transform)
      }
      W2_ACC(0, w2t);
      period_end();
    };
    int g0 = 0;
    if (par == 0) {                // parity 0 did period 0's transform in the prologue: slab request only
      if (2 < G) dma_u();
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // (the prologue's slabs and rows; not this period's requests)
      period_end();
      g0 = 1;
    }
    for (;;) {                     // four periods per trip: a (rows -> set 0), b (set 1), a (rows -> set 1), b (set 0)
      if (g0 >= G) break;
      period_a(g0, S0);
      if (g0 + 1 >= G) break;
      period_b(g0 + 1, S1);
      if (g0 + 2 >= G) break;
      period_a(g0 + 2, S1);
      if (g0 + 3 >= G) break;
      period_b(g0 + 3, S0);
      g0 += 4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (masked requests of the last periods)
#ifdef FS_W2_STAMPS
    if (blockIdx.x == 3 && lane == 0)
      for (int k = 0; k < 8; ++k) fs_w2_dbg[wave * 8 + k] = w2s[k];
#endif
#undef W2_WAIT_ROWS
#undef W2_PK_ADD
#undef W2_PK_SUB
#else
    (void)xvol; (void)NUW; (void)G;
#endif
    return;
  }

  // ---- matrix waves: wave = (channel tile m, y-component pair typ); MFMA column = (z row col >> 4, x-tile col & 15)
#ifdef FS_ABLATION
  if (p.ab & 2) __builtin_amdgcn_s_setprio(3);
#endif
  const int m = wv >> 1, typ = wv & 1;
  const int col = lane & 31, kh = lane >> 5;
  const int bBo = kh * W2_VCH + (col >> 4) * W2_ZP + 2 * typ * 96 + (col & 15);
  const int aBo = NV + kh * W2_UCH + 2 * typ * 384 + m * 32 + col;
  constexpr int NP = 36;  // reduction steps per chunk: kz x (ty of the pair) x tx, one channel pair

  f32x16 acc[2][6];  // [ty of the pair][tx]
  const float* __restrict__ ad = p.addend;
  const float* __restrict__ slope = p.slope;
  const float* __restrict__ dy = p.dy;
  float* __restrict__ Zp = p.Z;
  const float* __restrict__ src = dy != nullptr ? dy : ad;
  const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  // the per-channel vectors go to LDS once (a load per channel in the store loop, or sixteen more registers in flight across
  // the exchange, cost the epilogue thousands of cycles per brick): cvec[0] bias, [1] PReLU slope, [2] PReLU-backward slope;
  // out-of-range channels and absent vectors read 0
  float* cvec = lds + NB * BUF;
  if (wv == 0) {
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)bias, (short)0, bias != nullptr ? p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc((void*)slope, (short)0, slope != nullptr ? p.nslope * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dslope, (short)0, dy != nullptr ? p.dnslope * 4 : 0, 0x00020000);
    cvec[lane] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rB, (unsigned)lane * 4u, 0, 0));
    cvec[64 + lane] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rL, p.nslope == 1 ? 0u : (unsigned)lane * 4u, 0, 0));
    cvec[128 + lane] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rD, p.dnslope == 1 ? 0u : (unsigned)lane * 4u, 0, 0));
  }

  __builtin_amdgcn_s_barrier();  // chunk 0 has landed
#ifdef FS_W2_STAMPS
  unsigned long long w2s[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 0 MFMA loop, 1 period barrier, 2 epilogue work, 3 epilogue barriers
  unsigned long long w2t = W2_T();
#endif
  int buf = 0;
  for (int ib = 0; ib < nb; ++ib) {
    auto chunk = [&](auto FIRSTc) {
      constexpr bool FIRST = decltype(FIRSTc)::value;
      const float* bB = lds + buf * BUF + bBo;
      const float* aB = lds + buf * BUF + aBo;
      // operands of two steps (x components 2 u, 2 u + 1 of one (kz, ty)) per pair of LDS reads -- `ds_read2_b32`: half the
      // LDS instructions --, PD pairs of steps ahead
      constexpr int PD = 2;   // pairs of steps the reads run ahead of the MFMAs
      float a[PD + 1][2], bq[PD + 1][2];
      auto lds_ops = [&](int u, float (&av)[2], float (&bv)[2]) {   // u = pair of steps: (kz, ty of the pair, x components 2 (u % 3), + 1)
        const int kz = u / 6, r6 = u - kz * 6, tl = r6 / 3, tt = 2 * (r6 - tl * 3);
        av[0] = aB[kz * 1536 + tl * 384 + tt * 64];
        av[1] = aB[kz * 1536 + tl * 384 + tt * 64 + 64];
        bv[0] = bB[kz * W2_ZP + tl * 96 + tt * 16];
        bv[1] = bB[kz * W2_ZP + tl * 96 + tt * 16 + 16];
      };
      constexpr int NPP = NP / 2;
      if (DBG != 4) {
#pragma unroll
        for (int u = 0; u < PD; ++u) lds_ops(u, a[u], bq[u]);
      }
#pragma unroll
      for (int u = 0; u < (DBG == 4 ? 0 : NPP); ++u) {
        if (u + PD < NPP) lds_ops(u + PD, a[(u + PD) % (PD + 1)], bq[(u + PD) % (PD + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        const int kz = u / 6, r6 = u - kz * 6, tl = r6 / 3, tt = 2 * (r6 - tl * 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (FIRST && kz == 0) {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            acc[tl][tt + h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % (PD + 1)][h], bq[u % (PD + 1)][h], zero, 0, 0, 0);
          } else {
            acc[tl][tt + h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % (PD + 1)][h], bq[u % (PD + 1)][h], acc[tl][tt + h], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (DBG == 4 && FIRST) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
          for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tl][tt][r] = 0.f;
      }
      W2_ACC(0, w2t);
      __builtin_amdgcn_s_barrier();
      W2_ACC(1, w2t);
      buf = buf == NB - 1 ? 0 : buf + 1;
    };
    chunk(std::true_type{});
    for (int c = 1; c < nch; ++c) chunk(std::false_type{});
    const int fbuf = buf == 0 ? NB - 1 : buf - 1;  // the buffer the last chunk has just left: free until the next brick's period 0

    // ---- epilogue, in two rounds of eight registers (= channels) so that accumulators, transformed values and the
    // addend in flight fit the register file
    // this brick's output rows: lane = (channel half kh, z row, x-tile), the wave's y row = typ.  All global accesses of the
    // epilogue are buffer operations: base (sample, channel tile) in scalar registers, the channel of register r as a scalar
    // offset, ONE 32-bit lane offset (row + 4 kh channels; out of range = masked) -- no 64-bit address per channel.
    const long long brick = brick_of(ib);
    unsigned tile = (unsigned)brick;
    const int txi = __builtin_amdgcn_readfirstlane((int)(tile % (unsigned)p.tx)); tile /= (unsigned)p.tx;
    const int tyi = __builtin_amdgcn_readfirstlane((int)(tile % (unsigned)p.ty)); tile /= (unsigned)p.ty;
    const int tzi = __builtin_amdgcn_readfirstlane((int)(tile % (unsigned)p.tz));
    const int b = __builtin_amdgcn_readfirstlane((int)(tile / (unsigned)p.tz));
    const int oz = tzi * 2 + (col >> 4), oy = tyi * 2 * YT + 2 * ((col & 15) / XT) + typ;
    const int xq = txi * 4 * XT + 4 * (col & (XT - 1));
    const bool live = oz < p.Do && oy < p.Ho;
    const unsigned yv4 = (unsigned)yvol * 4u;
    const unsigned loff = live ? (unsigned)(4 * kh) * yv4 + (((unsigned)oz * p.Ho + oy) * p.Wo + xq) * 4u : DMA_OOB;
    const size_t cbase = ((size_t)b * p.Cout + m * 32) * yvol;
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + cbase), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)(Zp != nullptr ? Zp + cbase : nullptr), (short)0, Zp != nullptr ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)(src != nullptr ? src + cbase : nullptr), (short)0, src != nullptr ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dy != nullptr ? p.dpart + (((size_t)brick * 2 + typ) * 64 + m * 32) * 2 : nullptr), (short)0, dy != nullptr ? 32 * 2 * 4 : 0, 0x00020000);
    const unsigned poff = col == 0 ? (unsigned)(4 * kh) * 8u : DMA_OOB;
    float4* ex = reinterpret_cast<float4*>(lds + fbuf * BUF);
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      // Ax^T in registers: X[ty of the pair][i] = 4 consecutive x of (channel register 8 rnd + i, this lane's column)
      float4 xa[8], xb[8];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = 8 * rnd + i;
          const float m0 = acc[tl][0][r], m1 = acc[tl][1][r], m2 = acc[tl][2][r], m3 = acc[tl][3][r], m4 = acc[tl][4][r], m5 = acc[tl][5][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          float4 xv = make_float4((m0 + s12) + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m5);
          asm volatile("" : "+v"(xv.x), "+v"(xv.y), "+v"(xv.z), "+v"(xv.w));  // (keeps the SLP vectoriser from pairing rows: moves + pressure)
          if (tl == 0) xa[i] = xv; else xb[i] = xv;
        }
      __builtin_amdgcn_sched_barrier(0);  // the accumulators of this round are dead BEFORE the addend's registers go live
      W2_ACC(5, w2t);
      // addend / act_y of the round's eight channels: requested here, in flight across the exchange
      u32x4 pre[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int cl = 8 * ((8 * rnd + i) >> 2) + (i & 3);  // channel of the register inside the tile, less 4 kh
        const unsigned off = (m * 32 + 4 * kh + cl) < p.Cout ? loff : DMA_OOB;
        pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rS, off, cl * yv4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // y row 0 = (X0 + X1) + X2: the typ = 0 wave needs its partner's X2; y row 1 = (X1 - X2) - X3: typ = 1 needs X1
#pragma unroll
      for (int i = 0; i < 8; ++i)   // (selected value by value: a select of the two arrays would put them in scratch memory)
        ex[(wv * 8 + i) * 64 + lane] = make_float4(typ == 0 ? xb[i].x : xa[i].x, typ == 0 ? xb[i].y : xa[i].y,
                                                   typ == 0 ? xb[i].z : xa[i].z, typ == 0 ? xb[i].w : xa[i].w);
      W2_ACC(2, w2t);
      __builtin_amdgcn_s_barrier();
      W2_ACC(3, w2t);
      float4 pvq[4];
      float c0q[4], c1q[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int cl = 8 * ((8 * rnd + i) >> 2) + (i & 3);
        const int co = m * 32 + 4 * kh + cl;
        const unsigned off = co < p.Cout ? loff : DMA_OOB;
        // the partner's values and the per-channel values of four channels are read together (one LDS latency per four
        // channels instead of one per channel)
        if ((i & 3) == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ck = m * 32 + 4 * kh + 8 * ((8 * rnd + i + k) >> 2) + ((i + k) & 3);
            pvq[k] = ex[((wv ^ 1) * 8 + i + k) * 64 + lane];
            c0q[k] = cvec[(dy != nullptr ? 128 : 0) + ck];
            c1q[k] = cvec[64 + ck];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const float4 pv = pvq[i & 3];
        const float4 u = xa[i], w = xb[i];
        const float4 v = typ == 0 ? make_float4((u.x + w.x) + pv.x, (u.y + w.y) + pv.y, (u.z + w.z) + pv.z, (u.w + w.w) + pv.w)
                                  : make_float4((pv.x - u.x) - w.x, (pv.y - u.y) - w.y, (pv.z - u.z) - w.z, (pv.w - u.w) - w.w);
        const float y4[4] = {__uint_as_float(pre[i][0]), __uint_as_float(pre[i][1]), __uint_as_float(pre[i][2]), __uint_as_float(pre[i][3])};
        if (dy != nullptr) {
          // fused PReLU backward: g * prelu'(act_y) stored; the sums of the slope and bias gradient terms over the wave's
          // 32 columns (one partial row per brick and y row; masked lanes read act_y = 0 and add 0 to both)
          const float sl = c0q[i & 3];
          const float g4v[4] = {v.x, v.y, v.z, v.w};
          float o4[4], sa = 0.f, sb = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            o4[k] = y4[k] > 0.f ? g4v[k] : sl * g4v[k];
            sa += y4[k] > 0.f ? 0.f : y4[k] * g4v[k];
            sb += o4[k];
          }
          const u32x4 ov = {__float_as_uint(o4[0]), __float_as_uint(o4[1]), __float_as_uint(o4[2]), __float_as_uint(o4[3])};
          __builtin_amdgcn_raw_buffer_store_b128(ov, rY, off, cl * yv4, 0);
          if (off == DMA_OOB) { sa = 0.f; sb = 0.f; }
#pragma unroll
          for (int sft = 1; sft < 32; sft <<= 1) {  // the 32 lanes of a half hold the same channel
            sa += __shfl_xor(sa, sft);
            sb += __shfl_xor(sb, sft);
          }
          const u32x2 sv = {__float_as_uint(sa), __float_as_uint(sb)};
          __builtin_amdgcn_raw_buffer_store_b64(sv, rP, poff + cl * 8, 0, 0);
        } else {
          const float bv = c0q[i & 3];
          const float4 w4 = make_float4(v.x + bv, v.y + bv, v.z + bv, v.w + bv);
          if (Zp != nullptr) {
            const float sv = c1q[i & 3];
            const u32x4 yo = {__float_as_uint(w4.x), __float_as_uint(w4.y), __float_as_uint(w4.z), __float_as_uint(w4.w)};
            const u32x4 zo = {__float_as_uint((w4.x > 0.f ? w4.x : sv * w4.x) + y4[0]), __float_as_uint((w4.y > 0.f ? w4.y : sv * w4.y) + y4[1]),
                              __float_as_uint((w4.z > 0.f ? w4.z : sv * w4.z) + y4[2]), __float_as_uint((w4.w > 0.f ? w4.w : sv * w4.w) + y4[3])};
            __builtin_amdgcn_raw_buffer_store_b128(yo, rY, off, cl * yv4, 0);
            __builtin_amdgcn_raw_buffer_store_b128(zo, rZ, off, cl * yv4, 0);
          } else {
            const u32x4 yo = {__float_as_uint(w4.x + y4[0]), __float_as_uint(w4.y + y4[1]), __float_as_uint(w4.z + y4[2]), __float_as_uint(w4.w + y4[3])};
            __builtin_amdgcn_raw_buffer_store_b128(yo, rY, off, cl * yv4, 0);
          }
        }
      }
      W2_ACC(4, w2t);
      __builtin_amdgcn_s_barrier();  // the round's image has been read: the next round / the next brick's slab may overwrite it
      W2_ACC(3, w2t);
    }
  }
#ifdef FS_W2_STAMPS
  if (blockIdx.x == 3 && lane == 0)
    for (int k = 0; k < 8; ++k) fs_w2_dbg[wave * 8 + k] = w2s[k];
#endif
}

#ifdef FS_ABLATION  // the round-3 form (one workgroup per brick): bitwise A/B of the persistent kernel, FLOWSCI_WINO2D_R3=1
template <int DBG, int XT>
__global__ __launch_bounds__(512, 1) void conv3d_wino2d_ws_kernel(const float* __restrict__ X,
                                                                 const float* __restrict__ Ut,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ Y, FP p) {
  constexpr int CI = 2;
  constexpr int NV = CI * W2_VCH, NU = CI * W2_UCH;
  constexpr int NUP = NU / 256;            // LDS-DMA wave-instructions (64 x 16 bytes) of the U slab
  constexpr int NUW = NUP / 4;             // per loader wave
  constexpr int BUF = NV + NU;
  constexpr int NEX = 4 * 64 * 32 * 4;     // the epilogue's exchange image: [ty][co][column] float4
  constexpr int NB = 3;                    // staging buffers: the U slab is requested TWO chunks ahead (a chunk is only ~2 300
                                           // MFMA cycles long -- less than a trip to L2 and back)
  constexpr int NLDS = NB * BUF > NEX ? NB * BUF : NEX;
  static_assert(NU % 256 == 0 && NV % 4 == 0 && NLDS * 4 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[NLDS];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wv = wave & 3;

  long long tile = blockIdx.x;
  {
    const long long per = p.tiles / 8;
    if (per > 0 && tile < per * 8) tile = (tile & 7) * per + (tile >> 3);  // contiguous brick ranges per XCD
  }
  const long long brick = tile;
  const int txi = (int)(tile % p.tx); tile /= p.tx;
  const int tyi = (int)(tile % p.ty); tile /= p.ty;
  const int tzi = (int)(tile % p.tz);
  const int b = (int)(tile / p.tz);
  constexpr int YT = 16 / XT;  // y-tiles of a brick
  const int oz0 = tzi * 2, oy0 = tyi * 2 * YT, ox0 = txi * 4 * XT;
  const size_t xvol = (size_t)p.Di * p.Hi * p.Wi;

  if (wave >= 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    // ---- loader waves.  The vector-ALU work of a loader wave is NOT hidden behind the matrix wave it shares a SIMD with
    // (ablation builds: the input transform alone cost 0.11 of 0.48 ms when two waves did all of it), so it is spread
    // evenly over all four SIMDs and all periods: a wave handles input channel cc = lw & 1 of every second chunk
    // (parity lw >> 1) in two half-steps -- y components 0, 1 in one period, 2, 3 in the next -- and a quarter of
    // every chunk's U slab.  Chunk j is read by the matrix waves in period j and is complete at the end of period j - 1:
    //   period j - 3: its rows are requested (after the wave's previous chunk is finished)
    //   period j - 2: By^T, then Bx^T of ty 0, 1 -> buffer j % 3      period j - 1: Bx^T of ty 2, 3
    const int lw = wave - 4, par = lw >> 1, cc = lw & 1;
    const int zr = lane >> 4, q = lane & 15;  // lane = (staged z row, x-tile)
    const int qx = q & (XT - 1), qy = q / XT;   // x-tile, y-tile of the lane's slot
    const int gz = oz0 - 1 + zr, gx = ox0 + 4 * qx;
    const int hx = qx == 0 ? ox0 - 1 : ox0 + 4 * XT;
    unsigned voff[4], hoff[4];
#pragma unroll
    for (int yr = 0; yr < 4; ++yr) {
      const int gy = oy0 + 2 * qy - 1 + yr;
      const bool rowok = gz >= 0 && gz < p.Di && gy >= 0 && gy < p.Hi;
      const unsigned rbase = ((unsigned)(rowok ? gz : 0) * p.Hi + (rowok ? gy : 0)) * p.Wi;
      voff[yr] = (rowok && gx < p.Wi) ? (rbase + gx) * 4u : DMA_OOB;   // Wi % (4 XT) == 0: a float4 is in or out whole
      hoff[yr] = (rowok && (qx == 0 || qx == XT - 1) && hx >= 0 && hx < p.Wi) ? (rbase + hx) * 4u : DMA_OOB;
    }
    const int vdst = cc * W2_VCH + zr * W2_ZP + q;
    unsigned uoff[NUW];
#pragma unroll
    for (int k = 0; k < NUW; ++k) uoff[k] = (unsigned)(64 * (lw + 4 * k) + lane) * 16u;
    auto dma_u = [&](int chunk, int buf) {
      if (DBG == 1 || DBG == 3) return;
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Ut + (size_t)chunk * CI * W2_UCH), (short)0, NU * 4, 0x00020000);
      float* base = lds + buf * BUF + NV;
#pragma unroll
      for (int k = 0; k < NUW; ++k)  // NUP == 4 NUW: every wave issues all of its NUW
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + 256 * (lw + 4 * k)), 16, uoff[k], 0, 0, 0);
    };
    float xr[4][4], xh[4];   // the raw rows in flight
    float yt[4][4], yh[4];   // By^T of the chunk in work (lives across the two half-steps)
    auto fetch = [&](int chunk) {
      if (DBG == 2 || DBG == 3) return;
      const int ch = chunk * CI + cc;
      const bool live = ch < p.Cin;
      const float* base = X + ((size_t)b * p.Cin + (live ? ch : 0)) * xvol;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, live ? (int)((unsigned)xvol * 4u) : 0, 0x00020000);
#pragma unroll
      for (int yr = 0; yr < 4; ++yr) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff[yr], 0, 0);
        xr[yr][0] = __uint_as_float(v[0]); xr[yr][1] = __uint_as_float(v[1]);
        xr[yr][2] = __uint_as_float(v[2]); xr[yr][3] = __uint_as_float(v[3]);
        xh[yr] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, hoff[yr], 0, 0));
      }
    };
    auto ytrans = [&]() {  // By^T over the four rows, column by column (the 4 own columns and the halo column)
      if (DBG == 2 || DBG == 3) return;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        yt[0][i] = xr[0][i] - xr[2][i];
        yt[1][i] = xr[1][i] + xr[2][i];
        yt[2][i] = xr[2][i] - xr[1][i];
        yt[3][i] = xr[1][i] - xr[3][i];
      }
      yh[0] = xh[0] - xh[2]; yh[1] = xh[1] + xh[2]; yh[2] = xh[2] - xh[1]; yh[3] = xh[1] - xh[3];
    };
    auto put_half = [&](int buf, auto HALF) {
      if (DBG == 2 || DBG == 3) return;
      constexpr int half = decltype(HALF)::value;
      float* dstb = lds + buf * BUF + vdst;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const float d1 = yt[2 * half + t2][0], d2 = yt[2 * half + t2][1], d3 = yt[2 * half + t2][2], d4 = yt[2 * half + t2][3];
        const float dh = yh[2 * half + t2];
        // d0 = left neighbour's last column, d5 = right neighbour's first; the first / last x-tile of a row keeps the halo
        // column (lanes 0 / 15 of the DPP row by the shift itself; XT = 8: also the seam between the two y-tiles)
        float d0 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(dh), __float_as_uint(d4), 0x111, 0xF, 0xF, false));
        float d5 = __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(dh), __float_as_uint(d1), 0x101, 0xF, 0xF, false));
        if (XT < 16) {
          d0 = qx == 0 ? dh : d0;
          d5 = qx == XT - 1 ? dh : d5;
        }
        float* dst = dstb + (2 * half + t2) * 96;
        const float p31 = d3 - d1, r42 = d4 - d2;
        dst[0 * 16] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        dst[1 * 16] = fmaf(-4.f, d1 + d2, d3 + d4);
        dst[2 * 16] = fmaf(4.f, d1 - d2, d4 - d3);
        dst[3 * 16] = fmaf(2.f, p31, r42);
        dst[4 * 16] = fmaf(-2.f, p31, r42);
        dst[5 * 16] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      }
    };
    static_assert(NUP == 4 * NUW, "every loader wave issues NUW slab instructions per chunk");
    const int nch = p.Cin / CI;
    const std::integral_constant<int, 0> H0{};
    const std::integral_constant<int, 1> H1{};
    // prologue = "period -1": chunk 0 whole (parity 0), the first half of chunk 1 (parity 1); slabs of chunks 0 and 1
    dma_u(0, 0);
    if (nch > 1) dma_u(1, 1);
    if (par == 0) {
      fetch(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ytrans();
      put_half(0, H0);
      put_half(0, H1);
      if (nch > 2) fetch(2);
    } else if (nch > 1) {
      fetch(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ytrans();
      put_half(1, H0);
    }
    if (par != 0 || nch <= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the slabs)
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                               // ... all but the 8 row loads of chunk 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t0 = 0; t0 < nch; ++t0) {
      // request this wave's quarter of chunk t0 + 2's slab, then wait for everything it requested in the previous period
      // (a quarter of chunk t0 + 1's slab, maybe rows: they have had a whole period to arrive)
      if (t0 + 2 < nch) {
        dma_u(t0 + 2, (t0 + 2) % NB);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DBG == 1 || DBG == 3 ? 0 : NUW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (((t0 + 1) & 1) == par) {
        if (t0 + 1 < nch) {          // second half of chunk t0 + 1, then its successor's rows
          put_half((t0 + 1) % NB, H1);
          if (t0 + 3 < nch) fetch(t0 + 3);
        }
      } else if (t0 + 2 < nch) {     // first half of chunk t0 + 2
        ytrans();
        put_half((t0 + 2) % NB, H0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // chunk t0 + 1 is in LDS; the matrix waves are done reading chunk t0
    }
#else
    (void)xvol; (void)NUW;
#endif
    return;
  }

  // ---- matrix waves: wave wv owns the y-component ty = wv; MFMA column = (z row col >> 4, x-tile col & 15)
  const int col = lane & 31, kh = lane >> 5;
  const int bBo = kh * W2_VCH + (col >> 4) * W2_ZP + wv * 96 + (col & 15);
  const int aBo = NV + kh * W2_UCH + wv * 384 + col;
  constexpr int NP = 18;  // reduction steps per chunk: kz x tx, one channel pair

  f32x16 acc[6][2];
#pragma unroll
  for (int tt = 0; tt < 6; ++tt)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][m][r] = 0.f;

  __builtin_amdgcn_s_barrier();  // chunk 0 has landed
  int buf = 0;
  for (int c0 = 0; c0 < p.Cin; c0 += CI) {
    const float* bB = lds + buf * BUF + bBo;
    const float* aB = lds + buf * BUF + aBo;
    auto lds_ops = [&](int j, float (&a)[2], float& bq) {
      const int kz = j / 6, tt = j - kz * 6;
      a[0] = aB[kz * 1536 + tt * 64];
      a[1] = aB[kz * 1536 + tt * 64 + 32];
      bq = bB[kz * W2_ZP + tt * 16];
    };
    float a0[2], a1[2], b0, b1;
    if (DBG != 4) lds_ops(0, a0, b0);
#pragma unroll
    for (int j = 0; j < (DBG == 4 ? 0 : NP); j += 2) {
      lds_ops(j + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      acc[j % 6][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0, acc[j % 6][0], 0, 0, 0);
      acc[j % 6][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0, acc[j % 6][1], 0, 0, 0);
      if (j + 2 < NP) lds_ops(j + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      acc[(j + 1) % 6][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1, acc[(j + 1) % 6][0], 0, 0, 0);
      acc[(j + 1) % 6][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1, acc[(j + 1) % 6][1], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    buf = buf == NB - 1 ? 0 : buf + 1;
  }

  // ---- epilogue 1: Ax^T in registers, this wave's (ty) results to the exchange image ex[ty][co][column] (float4 = 4 x)
  float4* ex = reinterpret_cast<float4*>(lds);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
      const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r], m5 = acc[5][m][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      ex[(wv * 64 + co) * 32 + col] = make_float4((m0 + s12) + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m5);
    }
  __builtin_amdgcn_s_barrier();  // (the loaders have left: the barrier counts the live waves only)

  // ---- epilogue 2: wave wv finishes channels 16 wv .. 16 wv + 15; lane = (y row lane >> 5, z row (lane >> 4) & 1, x-tile)
  const int yy = lane >> 5, c5 = lane & 31;
  const int oz = oz0 + (c5 >> 4), oy = oy0 + 2 * ((c5 & 15) / XT) + yy;
  const int xq = ox0 + 4 * (c5 & (XT - 1));
  const size_t yvol = (size_t)p.Do * p.Ho * p.Wo;
  const bool live = oz < p.Do && oy < p.Ho;
  const size_t orow = ((size_t)(live ? oz : 0) * p.Ho + (live ? oy : 0)) * p.Wo + xq;
  const float* __restrict__ ad = p.addend;
  const float* __restrict__ slope = p.slope;
  const float* __restrict__ dy = p.dy;
  float* __restrict__ Zp = p.Z;
  const float* __restrict__ src = dy != nullptr ? dy : ad;
  float* __restrict__ prow = dy != nullptr ? p.dpart + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 * 2 : nullptr;
  (void)brick;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    float4 v[4], pre[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // addend / act_y of four channels first: four loads in flight
      const int co = 16 * wv + 4 * g4 + i;
      const size_t o = ((size_t)b * p.Cout + (co < p.Cout ? co : 0)) * yvol + orow;
      pre[i] = (src != nullptr && live && co < p.Cout) ? *reinterpret_cast<const float4*>(src + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = 16 * wv + 4 * g4 + i;
      // Ay^T: y row 0 = (m0 + m1) + m2, y row 1 = (m1 - m2) - m3
      const float4 ma = ex[((yy + 0) * 64 + co) * 32 + c5];
      const float4 mb = ex[((yy + 1) * 64 + co) * 32 + c5];
      const float4 mc = ex[((yy + 2) * 64 + co) * 32 + c5];
      v[i] = yy == 0 ? make_float4((ma.x + mb.x) + mc.x, (ma.y + mb.y) + mc.y, (ma.z + mb.z) + mc.z, (ma.w + mb.w) + mc.w)
                     : make_float4((ma.x - mb.x) - mc.x, (ma.y - mb.y) - mc.y, (ma.z - mb.z) - mc.z, (ma.w - mb.w) - mc.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = 16 * wv + 4 * g4 + i;
      const bool ok = live && co < p.Cout;
      const size_t o = ((size_t)b * p.Cout + (co < p.Cout ? co : 0)) * yvol + orow;
      if (dy != nullptr) {
        // fused PReLU backward: g * prelu'(act_y) stored; the wave's sums of the slope and bias gradient terms
        const float sl = p.dslope[p.dnslope == 1 ? 0 : (co < p.Cout ? co : 0)];
        const float g4v[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        const float y4[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
        float o4[4], sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o4[k] = y4[k] > 0.f ? g4v[k] : sl * g4v[k];
          sa += y4[k] > 0.f ? 0.f : y4[k] * g4v[k];
          sb += o4[k];
        }
        if (ok) *reinterpret_cast<float4*>(Y + o) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        if (!ok) { sa = 0.f; sb = 0.f; }
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {  // all 64 lanes hold the same channel
          sa += __shfl_xor(sa, sft);
          sb += __shfl_xor(sb, sft);
        }
        if (lane == 0) {
          prow[co * 2] = sa;
          prow[co * 2 + 1] = sb;
        }
      } else if (ok) {
        const float bv = bias != nullptr ? bias[co] : 0.f;
        const float4 w4 = make_float4(v[i].x + bv, v[i].y + bv, v[i].z + bv, v[i].w + bv);
        const float4 av = pre[i];
        if (Zp != nullptr) {
          const float sv = slope[p.nslope == 1 ? 0 : co];
          *reinterpret_cast<float4*>(Y + o) = w4;
          *reinterpret_cast<float4*>(Zp + o) = make_float4((w4.x > 0.f ? w4.x : sv * w4.x) + av.x, (w4.y > 0.f ? w4.y : sv * w4.y) + av.y,
                                                           (w4.z > 0.f ? w4.z : sv * w4.z) + av.z, (w4.w > 0.f ? w4.w : sv * w4.w) + av.w);
        } else {
          *reinterpret_cast<float4*>(Y + o) = make_float4(w4.x + av.x, w4.y + av.y, w4.z + av.z, w4.w + av.w);
        }
      }
    }
  }
}

#endif

// Measured and not kept (tests/tools/wino_bench.py, 64 -> 64 at 2 x 64^3, 0.466 ms as built incl. the re-layout launch;
// ablation builds: matrix waves + epilogue alone 0.354, loaders alone 0.276): two waves doing all of the input transform
// and two all of the slab requests 0.479; the last two reduction steps of a chunk held back across the barrier so that
// their MFMAs cover the first operand reads of the next chunk, operands read two steps ahead: 0.461 (within noise).
inline int wino2d_xt(const FP& p) { return p.Wo % 64 == 0 ? 16 : 8; }

inline bool wino2d_ok(const FP& p, const float* x, const float* ws, int Cin, int Cout, int kernel, int stride, bool has_ms) {
  static const bool off = FS_AB_ENV("FLOWSCI_FWD_NO_WINO2D") || FS_AB_ENV("FLOWSCI_FWD_NO_WINO4") || FS_AB_ENV("FLOWSCI_FWD_NO_WINO");
  if (off || kernel != 3 || stride != 1 || p.pad != 1 || has_ms) return false;
  if (Cin % 4 != 0 || Cout > 64 || p.CoutP != 64) return false;
  if (p.Wi != p.Wo || p.Wi % 32 != 0 || p.Di != p.Do || p.Hi != p.Ho) return false;
  if ((((uintptr_t)x | (uintptr_t)ws) & 15) != 0) return false;
  if ((long long)p.Di * p.Hi * p.Wi * 4 >= (1ll << 31)) return false;
  if ((long long)p.Do * p.Ho * p.Wo * 4 * 32 >= (1ll << 32)) return false;  // the epilogue's 32-bit channel offsets inside a 32-channel tile
  // one brick (2 z x 2 y x 64 x, or 2 z x 4 y x 32 x) per CU is enough: measured (tests/tools/wino_bench.py, 64 -> 64 layer)
  // 0.072 ms at 256 bricks against 0.129 for the direct small-brick kernel, 0.131 / 0.143 / 0.183 / 0.257 (2-D / F(4,3) /
  // F(2,3) / direct) at 512; the 64^3 trunk has 2048
  static const long long min_bricks = FS_AB_ENV_LL("FLOWSCI_WINO2D_MIN", 256);
  const int xt = wino2d_xt(p);
  return (long long)p.B * fs::cdiv(p.Do, 2) * fs::cdiv(p.Ho, 2 * (16 / xt)) * (p.Wo / (4 * xt)) >= min_bricks;
}

// partial rows per brick of the fused PReLU-backward epilogue (fs_conv3d_fwd_dprelu: one per y row of the brick's pair)
inline bool wino2d_r3() {
  static const bool r3 = FS_AB_ENV("FLOWSCI_WINO2D_R3");
  return r3;
}
inline int wino2d_part_rows() { return wino2d_r3() ? 1 : 2; }

template <int XT>
void launch_wino2d_t(const float* X, const float* Ut, const float* bias, float* Y, const FP& p, hipStream_t st) {
  // one persistent workgroup per CU of the device the call runs on (per-device table: a process may drive several GPUs,
  // and two threads may make their first call together -- an int store is the only shared write)
  static int ncu_of[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int ncu = ncu_of[dev];
  if (ncu == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    ncu_of[dev] = ncu = n;
  }
  const dim3 g((unsigned)(p.tiles < ncu ? p.tiles : ncu), 1);
#ifdef FS_ABLATION  // instantiations that SKIP work (wrong results by design) and the round-3 form: measurement builds only
  static const int dbg = (int)FS_AB_ENV_LL("FLOWSCI_WINO_DBG", 0);
  if (wino2d_r3()) {
    const dim3 g3((unsigned)p.tiles, 1);
    if (dbg == 1) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<1, XT>), g3, dim3(512), 0, st, X, Ut, bias, Y, p);
    else if (dbg == 2) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<2, XT>), g3, dim3(512), 0, st, X, Ut, bias, Y, p);
    else if (dbg == 3) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<3, XT>), g3, dim3(512), 0, st, X, Ut, bias, Y, p);
    else if (dbg == 4) hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<4, XT>), g3, dim3(512), 0, st, X, Ut, bias, Y, p);
    else hipLaunchKernelGGL((conv3d_wino2d_ws_kernel<0, XT>), g3, dim3(512), 0, st, X, Ut, bias, Y, p);
    return;
  }
  if (dbg == 1) hipLaunchKernelGGL((conv3d_wino2d_ps_kernel<1, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 2) hipLaunchKernelGGL((conv3d_wino2d_ps_kernel<2, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 3) hipLaunchKernelGGL((conv3d_wino2d_ps_kernel<3, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else if (dbg == 4) hipLaunchKernelGGL((conv3d_wino2d_ps_kernel<4, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
  else
#endif
    hipLaunchKernelGGL((conv3d_wino2d_ps_kernel<0, XT>), g, dim3(512), 0, st, X, Ut, bias, Y, p);
}

inline int launch_wino2d(const float* X, const float* Ut, const float* bias, float* Y, FP& p, hipStream_t st) {
  const int xt = wino2d_xt(p);
#ifdef FS_ABLATION
  static const int ab = (int)FS_AB_ENV_LL("FLOWSCI_WINO2D_AB", 1);  // bit 0: loader waves at raised priority (the product's setting)
  p.ab = ab;
#endif
  p.tz = fs::cdiv(p.Do, 2); p.ty = fs::cdiv(p.Ho, 2 * (16 / xt)); p.tx = p.Wo / (4 * xt);
  p.tiles = (long long)p.B * p.tz * p.ty * p.tx;
  if (p.tiles >= (1ll << 31)) return FS_ERR_SHAPE;
  if (xt == 16) launch_wino2d_t<16>(X, Ut, bias, Y, p, st);
  else launch_wino2d_t<8>(X, Ut, bias, Y, p, st);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
