// lds_dma_hazards.hip -- a reproducer for what the bring-up of csrc/convtr_s3.hpp found: is the ADDRESS
// register of a 16-byte LDS-DMA copy (`buffer_load_dwordx4 vaddr, rsrc, 0 offen lds`) safe to overwrite right after the copy
// has been issued?  A workgroup of four waves copies a 48 KB slab from memory into LDS as 48 copies of 1 KB (12 per wave,
// 64 lanes x 16 bytes each), waits for vmcnt(0), meets at a barrier and compares LDS with memory.
//   MODE 0  the 12 lane offsets live in 12 registers written once before the first copy
//   MODE 1  ONE register, re-computed (v_or_b32) in front of every copy -- what a compiler makes of `1024 * i + 16 * lane`
//   MODE 2  as MODE 1, with QUEUE register loads of a cold buffer issued first (the memory pipeline is backed up when the
//           copies arrive)
//   MODE 3  as MODE 2 with the loads as inline-assembly buffer loads (the failing kernel's form)
//   MODE 4  as MODE 3 with the failing build's exact instruction sequence per copy (s_add_i32 m0, base, literal; v_or_b32;
//           s_add_i32 of the next offset; the copy)
// The workgroup has the failing kernel's shape: 12 waves (8 of them only wait at the barrier), 135 KB of LDS, the slab image
// at byte 40 320.
// Prints the mismatching 4-byte words per mode, summed over all workgroups and repetitions, and per copy index.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_dma_hazards.hip -o /tmp/lds_dma_hazards && /tmp/lds_dma_hazards
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int SLAB = 48 * 1024, NCOPY = 12, QUEUE = 8, LDSOFF = 40320, LDSALL = LDSOFF + 2 * SLAB;  // (the failing kernel's LDS map)

template <int MODE>
__global__ __launch_bounds__(768) void k(const unsigned* __restrict__ slab, const u32x4* __restrict__ cold, size_t cold_n,
                                         unsigned long long* __restrict__ bad, unsigned long long* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[LDSALL];
  unsigned char* const lds = lds_all + LDSOFF;
  if (threadIdx.x >= 256) {  // eight more waves that only wait (the failing kernel's matrix waves)
    __syncthreads();
    __syncthreads();
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds;
  // zero the LDS image (a stale image of a previous workgroup would match)
  for (int i = threadIdx.x; i < SLAB / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  i32x4 r;
  {
    const unsigned long long a = (unsigned long long)slab;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = SLAB;
    r[3] = 0x00020000;
  }
  u32x4 q[QUEUE];
  if (MODE >= 3) {
    // ... as inline-assembly buffer loads, like the failing kernel's
    const unsigned long long a = (unsigned long long)cold;
    i32x4 rc;
    rc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    rc[2] = 0x40000000;  // = the cold buffer's 1 GiB: the descriptor's range check keeps every lane inside the allocation
    rc[3] = 0x00020000;
    const unsigned voff = (unsigned)((((size_t)blockIdx.x * 256 + threadIdx.x) * 7919u) % (((size_t)1 << 26) - 4096)) * 16u;  // (+ 4096 i < 1 GiB)
#pragma unroll
    for (int i = 0; i < QUEUE; ++i) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q[i]) : "v"(voff + 4096u * i), "s"(rc));
  }
  if (MODE == 2) {
    // register loads of a cold buffer first: the copies queue behind them
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 97u;
#pragma unroll
    for (int i = 0; i < QUEUE; ++i) q[i] = cold[(base + (size_t)i * 1315423911u) % cold_n];
  }
  const unsigned lane16 = 16u * (unsigned)lane;
  if (MODE == 0) {
    unsigned off[NCOPY];
#pragma unroll
    for (int kk = 0; kk < NCOPY; ++kk) {
      off[kk] = 1024u * (unsigned)(wv + 4 * kk) + lane16;
      asm volatile("" : "+v"(off[kk]));
    }
#pragma unroll
    for (int kk = 0; kk < NCOPY; ++kk) {
      const unsigned m0v = lds0 + 1024u * (unsigned)(wv + 4 * kk);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(off[kk]), "s"(r) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < NCOPY; ++kk) asm volatile("" ::"v"(off[kk]));
  } else if (MODE == 4) {
    unsigned t = 0;
    int s4 = wv << 10, soff = wv << 10;
    asm volatile("" : "+s"(s4), "+s"(soff));
#pragma unroll
    for (int kk = 0; kk < NCOPY; ++kk)
      asm volatile("s_add_i32 m0, %2, %3\n\tv_or_b32 %0, %1, %4\n\ts_add_i32 %1, %2, %5\n\tbuffer_load_dwordx4 %0, %6, 0 offen lds"
                   : "+v"(t), "+s"(soff) : "s"(s4), "n"(LDSOFF + 4096 * kk), "v"(lane16), "n"(4096 * (kk + 1)), "s"(r) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    unsigned t = 0;
#pragma unroll
    for (int kk = 0; kk < NCOPY; ++kk) {
      const unsigned m0v = lds0 + 1024u * (unsigned)(wv + 4 * kk);
      const unsigned sk = 1024u * (unsigned)(wv + 4 * kk);
      // ONE register, rewritten in front of every copy (and therefore right behind the previous one)
      asm volatile("v_or_b32 %0, %1, %2\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %4, 0 offen lds"
                   : "+v"(t) : "s"(sk), "v"(lane16), "s"(m0v), "s"(r) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  unsigned long long acc = 0;
  if (MODE >= 2) {
    if (MODE >= 3) asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
#pragma unroll
    for (int i = 0; i < QUEUE; ++i) acc += q[i].x + q[i].w;
  }
  unsigned nb = 0;
  for (int i = threadIdx.x; i < SLAB / 4; i += 256) {
    const bool m = reinterpret_cast<const unsigned*>(lds)[i] != slab[i];
    nb += m ? 1u : 0u;
    if (m) atomicAdd(&bad[1 + (i >> 8) / 4], 1ull);   // per copy index kk = chunk / 4
  }
  if (nb) atomicAdd(&bad[0], (unsigned long long)nb);
  if (acc == 0x123456789ull) sink[0] = acc;
}

int main() {
  std::vector<unsigned> h(SLAB / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) ^ 0x5bd1e995u;
  unsigned* slab;
  CHECK(hipMalloc(&slab, SLAB));
  CHECK(hipMemcpy(slab, h.data(), SLAB, hipMemcpyHostToDevice));
  const size_t cold_n = (size_t)1 << 26;  // 1 GiB of 16-byte elements
  u32x4* cold;
  CHECK(hipMalloc(&cold, cold_n * 16));
  CHECK(hipMemset(cold, 1, cold_n * 16));
  unsigned long long *bad, *sink;
  CHECK(hipMalloc(&bad, 16 * 8));
  CHECK(hipMalloc(&sink, 8));
  const int WG = 4096, REPS = 20;
  for (int mode = 0; mode < 5; ++mode) {
    CHECK(hipMemset(bad, 0, 16 * 8));
    for (int rep = 0; rep < REPS; ++rep) {
      CHECK(hipMemset(cold, rep & 255, cold_n * 16));  // everything cold again (and the slab out of L2)
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(WG), dim3(768), 0, 0, slab, cold, cold_n, bad, sink);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(WG), dim3(768), 0, 0, slab, cold, cold_n, bad, sink);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(WG), dim3(768), 0, 0, slab, cold, cold_n, bad, sink);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(WG), dim3(768), 0, 0, slab, cold, cold_n, bad, sink);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(WG), dim3(768), 0, 0, slab, cold, cold_n, bad, sink);
      CHECK(hipDeviceSynchronize());
    }
    unsigned long long hb[16];
    CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
    const double total = (double)WG * REPS * (SLAB / 4);
    printf("MODE %d: %llu mismatching words of %.3g (%.4f %%); per copy index:", mode, hb[0], total, 100.0 * hb[0] / total);
    for (int kk = 0; kk < NCOPY; ++kk) printf(" %llu", hb[1 + kk]);
    printf("\n");
  }
  return 0;
}
