// split_bf16_gemm.hip -- go / no-go experiment: can an fp32-ACCURATE convolution inner loop run on
// the bf16 matrix rate?  An fp32 operand is exactly the sum of three bf16 pieces (a = a0 + a1 + a2, 8 + 8 + 8 significant
// bits), every pairwise product of pieces is exact in fp32, and v_mfma_f32_32x32x16_bf16 accumulates in fp32:
//     a * b  =  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)  [+ a1 b2 + a2 b1 + a2 b2  ~ 2^-27 |a b|]
// i.e. 6 (or 9) bf16 MFMAs of K = 16 (32 cycles each) replace 8 fp32 MFMAs of K = 2 (64 cycles each): 192 (288) vs 512
// matrix-pipe cycles per K = 16.
//
// The loop modelled is the 64 -> 64 channel k3 direct convolution tile loop (csrc/convfwd.hip): C[64 co][positions] +=
// A_tap[64 co][ci] . B[ci][positions]; a workgroup stages a B tile (256 positions x 32 input channels) in LDS once and
// uses it for REUSE = 27 "taps", each with its own weight slab A_r streamed from L2 through LDS; every MFMA operand is read
// from LDS (as in the real kernels); fp32 operands are split into bf16 pieces on the LOADER side (the activations: when
// they are written to LDS; the weights: once, on the host = the per-step weight re-layout).
//   MODE 0  v_mfma_f32_32x32x2_f32            (what csrc/ runs today)
//   MODE 1  3 bf16 pieces, 6 products
//   MODE 2  3 bf16 pieces, 9 products
// Prints: max / rms error against fp64 of a K = 1728 GEMM on random normal data for the three modes (REUSE = 1), then the
// sustained rate of the tile loop on random data for one 64 -> 64 layer at 2 x 64^3 positions (116 GFLOP), LDS bytes read
// per matrix-pipe cycle, and the effective clock.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/split_bf16_gemm.hip -o /tmp/split_bf16_gemm && /tmp/split_bf16_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int M = 64;     // output channels of the tile
constexpr int TN = 256;   // positions per workgroup tile (4 waves x 64)
constexpr int KC = 32;    // reduction elements staged per chunk
constexpr int PF = KC + 1;   // fp32 LDS pitch (floats): conflict-free ds_read_b32 columns
constexpr int PH = KC + 8;   // bf16 LDS pitch (elements; 80 bytes): conflict-free ds_read_b128 columns

__device__ __forceinline__ void split3(float a, __bf16& h0, __bf16& h1, __bf16& h2) {
  h0 = (__bf16)a;
  float r = a - (float)h0;  // exact
  h1 = (__bf16)r;
  r = r - (float)h1;        // exact
  h2 = (__bf16)r;
}

// A: MODE 0: [R][64][Kst] fp32;  MODE 1/2: [R][3][64][Kst] bf16 (pre-split).  B: [N][Kst] fp32.  C: [64][N] fp32.
template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const void* __restrict__ Av, const float* __restrict__ B,
                                                       float* __restrict__ C, int N, int Kst, int R) {
  constexpr bool SPLIT = MODE != 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SPLIT ? (3 * M * PH * 2 + 3 * TN * PH * 2) : (M * PF * 4 + TN * PF * 4)];
  float(*sAf)[PF] = reinterpret_cast<float(*)[PF]>(smem);
  float(*sBf)[PF] = reinterpret_cast<float(*)[PF]>(smem + M * PF * 4);
  __bf16(*sAh)[M][PH] = reinterpret_cast<__bf16(*)[M][PH]>(smem);
  __bf16(*sBh)[TN][PH] = reinterpret_cast<__bf16(*)[TN][PH]>(smem + 3 * M * PH * 2);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int l32 = lane & 31, lh = lane >> 5;
  const int ntiles = N / TN;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    f32x16 acc[2][2] = {};
    for (int kc = 0; kc < Kst; kc += KC) {
      // ---- the B tile of this chunk: 256 positions x 32 reduction elements, split on the way into LDS
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = t + 256 * j, pos = idx >> 3, k4 = idx & 7;
        const float4 v = *reinterpret_cast<const float4*>(B + (size_t)(tile * TN + pos) * Kst + kc + 4 * k4);
        if (!SPLIT) {
          sBf[pos][4 * k4 + 0] = v.x; sBf[pos][4 * k4 + 1] = v.y; sBf[pos][4 * k4 + 2] = v.z; sBf[pos][4 * k4 + 3] = v.w;
        } else {
          bf16x4 p0, p1, p2;
          const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) { __bf16 a, b, c; split3(f[i], a, b, c); p0[i] = a; p1[i] = b; p2[i] = c; }
          *reinterpret_cast<bf16x4*>(&sBh[0][pos][4 * k4]) = p0;
          *reinterpret_cast<bf16x4*>(&sBh[1][pos][4 * k4]) = p1;
          *reinterpret_cast<bf16x4*>(&sBh[2][pos][4 * k4]) = p2;
        }
      }
      for (int r = 0; r < R; ++r) {
        // ---- weight slab of "tap" r for this chunk (from L2), already in pieces for the split modes
        if (!SPLIT) {
          const float* A = reinterpret_cast<const float*>(Av) + (size_t)r * M * Kst;
          float4 v[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int idx = t + 256 * j, row = idx >> 3, k4 = idx & 7;
            v[j] = *reinterpret_cast<const float4*>(A + (size_t)row * Kst + kc + 4 * k4);
          }
          __syncthreads();  // the previous tap's MFMAs have read sA
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int idx = t + 256 * j, row = idx >> 3, k4 = idx & 7;
            sAf[row][4 * k4 + 0] = v[j].x; sAf[row][4 * k4 + 1] = v[j].y; sAf[row][4 * k4 + 2] = v[j].z; sAf[row][4 * k4 + 3] = v[j].w;
          }
        } else {
          const __bf16* A = reinterpret_cast<const __bf16*>(Av) + (size_t)r * 3 * M * Kst;
          const int row = t >> 2, part = t & 3;
          bf16x8 v[3];
#pragma unroll
          for (int p = 0; p < 3; ++p)
            v[p] = *reinterpret_cast<const bf16x8*>(A + ((size_t)p * M + row) * Kst + kc + 8 * part);
          __syncthreads();
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(&sAh[p][row][8 * part]) = v[p];
        }
        __syncthreads();
        // ---- matrix phase: every operand read from LDS
        if (!SPLIT) {
#pragma unroll
          for (int s = 0; s < KC / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              a[i] = sAf[i * 32 + l32][2 * s + lh];
              b[i] = sBf[wv * 64 + i * 32 + l32][2 * s + lh];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int s = 0; s < KC / 16; ++s) {
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                a[p][i] = *reinterpret_cast<const bf16x8*>(&sAh[p][i * 32 + l32][16 * s + 8 * lh]);
                b[p][i] = *reinterpret_cast<const bf16x8*>(&sBh[p][wv * 64 + i * 32 + l32][16 * s + 8 * lh]);
              }
            // small terms first
            constexpr int NPROD = MODE == 1 ? 6 : 9;
            constexpr int PA[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0}, PB[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 9 - NPROD; q < 9; ++q)
#pragma unroll
              for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    // ---- C[row][position]: acc[i][j][v]: row = 32 i + 8 (v / 4) + 4 lh + v % 4, position = 64 wv + 32 j + l32
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = 32 * i + 8 * (v / 4) + 4 * lh + (v % 4);
          C[(size_t)row * N + (size_t)tile * TN + wv * 64 + 32 * j + l32] = acc[i][j][v];
        }
  }
}

static uint16_t f2bf(float f) {  // round to nearest even
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static double gauss() {
  double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
static float launch(const void* A, const float* B, float* C, int N, int Kst, int R, int blocks, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(gemm_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, A, B, C, N, Kst, R);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, A, B, C, N, Kst, R);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.name, cus);
  srand(1234);
  // ---------------- accuracy: one K = 1728 reduction (64 input channels x 27 taps), N = 1024 positions
  {
    const int N = 1024, K = 1728;
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    for (auto& x : A) x = (float)(gauss() * 0.05);   // weights of a trained layer: small
    for (auto& x : B) x = (float)gauss();            // activations
    std::vector<uint16_t> As((size_t)3 * M * K);
    for (size_t i = 0; i < (size_t)M * K; ++i) {
      const float a = A[i];
      const uint16_t h0 = f2bf(a);
      const float r1 = a - bf2f(h0);
      const uint16_t h1 = f2bf(r1);
      const float r2 = r1 - bf2f(h1);
      As[i] = h0; As[(size_t)M * K + i] = h1; As[(size_t)2 * M * K + i] = f2bf(r2);
    }
    std::vector<double> ref((size_t)M * N);
    double cmax = 0;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
        ref[(size_t)m * N + n] = s;
        cmax = fmax(cmax, fabs(s));
      }
    float *dA, *dB, *dC;
    void* dAs;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dAs, As.size() * 2));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dAs, As.data(), As.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> out((size_t)M * N);
    const char* names[3] = {"fp32 MFMA 32x32x2      ", "3 bf16 pieces, 6 products", "3 bf16 pieces, 9 products"};
    double e0max = 0;
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0) launch<0>(dA, dB, dC, N, K, 1, N / TN, 1);
      if (mode == 1) launch<1>(dAs, dB, dC, N, K, 1, N / TN, 1);
      if (mode == 2) launch<2>(dAs, dB, dC, N, K, 1, N / TN, 1);
      CK(hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost));
      double emax = 0, e2 = 0;
      for (size_t i = 0; i < out.size(); ++i) {
        const double e = fabs((double)out[i] - ref[i]);
        emax = fmax(emax, e);
        e2 += e * e;
      }
      if (mode == 0) e0max = emax;
      printf("accuracy K=1728  %s  max err %.3e (%.2e of max|C| %.3f)  rms %.3e   max err / fp32-MFMA's %.2f\n", names[mode],
             emax, emax / cmax, cmax, sqrt(e2 / out.size()), emax / e0max);
    }
    // the same with a sequential fp32 fma chain on the host (what "fp32 accuracy" means for a K = 1728 dot product)
    {
      double emax = 0;
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
          float s = 0.f;
          for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)m * K + k], B[(size_t)n * K + k], s);
          emax = fmax(emax, fabs((double)s - ref[(size_t)m * N + n]));
        }
      printf("accuracy K=1728  host fp32 fmaf chain       max err %.3e   / fp32-MFMA's %.2f\n", emax, emax / e0max);
    }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dAs);
  }
  // ---------------- rate: one 64 -> 64 k3 layer at 2 x 64^3 positions: 64 staged channels x 27 taps
  {
    const int N = 2 * 64 * 64 * 64, Kst = 64, R = 27;
    std::vector<float> A((size_t)R * M * Kst), B((size_t)N * Kst);
    for (auto& x : A) x = (float)(gauss() * 0.05);
    for (size_t i = 0; i < B.size(); ++i) B[i] = (float)gauss();
    std::vector<uint16_t> As((size_t)R * 3 * M * Kst);
    for (int r = 0; r < R; ++r)
      for (size_t i = 0; i < (size_t)M * Kst; ++i) {
        const float a = A[(size_t)r * M * Kst + i];
        const uint16_t h0 = f2bf(a);
        const float r1 = a - bf2f(h0);
        const uint16_t h1 = f2bf(r1);
        const float r2 = r1 - bf2f(h1);
        uint16_t* d = &As[(size_t)r * 3 * M * Kst];
        d[i] = h0; d[(size_t)M * Kst + i] = h1; d[(size_t)2 * M * Kst + i] = f2bf(r2);
      }
    float *dA, *dB, *dC;
    void* dAs;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dAs, As.size() * 2));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dAs, As.data(), As.size() * 2, hipMemcpyHostToDevice));
    const double flop = 2.0 * M * (double)N * Kst * R;
    const int blocks = 2 * cus;
    // warm the clocks: ~2 s of back-to-back launches, then measure
    for (int mode = 0; mode < 3; ++mode) {
      float ms = 0;
      if (mode == 0) { launch<0>(dA, dB, dC, N, Kst, R, blocks, 300); ms = launch<0>(dA, dB, dC, N, Kst, R, blocks, 200); }
      if (mode == 1) { launch<1>(dAs, dB, dC, N, Kst, R, blocks, 300); ms = launch<1>(dAs, dB, dC, N, Kst, R, blocks, 200); }
      if (mode == 2) { launch<2>(dAs, dB, dC, N, Kst, R, blocks, 300); ms = launch<2>(dAs, dB, dC, N, Kst, R, blocks, 200); }
      // matrix-pipe cycles per wave: fp32: K/2 MFMAs x 4 tiles x 64 cycles; split: K/16 x products x 4 tiles x 32 cycles
      const double kt = (double)Kst * R;
      const double mf_cycles = (mode == 0 ? kt / 2 * 4 * 64 : kt / 16 * (mode == 1 ? 6 : 9) * 4 * 32) * (N / TN) * 4.0;  // all waves
      const double per_simd = mf_cycles / (cus * 4.0);
      // LDS bytes read by the matrix phase per wave and K = 16: fp32 8 steps x 4 reads x 256 B; split 12 reads x 1 KiB
      const double lds_bytes = (mode == 0 ? kt / 2 * 4 * 256.0 : kt / 16 * 12 * 1024.0) * (N / TN) * 4.0;
      printf("rate  %s  %.4f ms  %.1f TFLOP/s useful   matrix-pipe busy if 2.4 GHz: %.2f   LDS read %.1f B/clk/CU at that time   (x%.2f vs fp32 MFMA)\n",
             mode == 0 ? "fp32 MFMA 32x32x2       " : mode == 1 ? "3 bf16 pieces, 6 products" : "3 bf16 pieces, 9 products", ms,
             flop / (ms * 1e-3) / 1e12, per_simd / (ms * 1e-3 * 2.4e9), lds_bytes / cus / (ms * 1e-3 * 2.4e9), 0.0);
    }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dAs);
  }
  return 0;
}
