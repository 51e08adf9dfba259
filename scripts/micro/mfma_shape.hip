// mfma_shape.hip -- the fp32-input MFMA rate the chip sustains from registers, for both shapes the convolution
// kernels use (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32): the ceiling the implicit-GEMM kernels are priced
// against next to the guide's 157.3 TFLOP/s, and the clock the matrix pipes actually hold under that load.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  if (SHAPE == 32) {
    f32x16 acc[4] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) s += acc[k][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) for (int j = 0; j < 4; ++j) s += acc[k][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int SHAPE>
double run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);  // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_iter = SHAPE == 32 ? 4.0 * 2 * 32 * 32 * 2 : 8.0 * 2 * 16 * 16 * 4;  // flops per wave and iteration
  return per_iter * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8;  // 8 workgroups of 4 waves per CU
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * blocks);
  const int iters = 200000;
  const double t32 = run<32>(out, blocks, iters), t16 = run<16>(out, blocks, iters);
  // 256 flops per CU and clock for either shape (4 SIMDs x 64 flops): clock = rate / (CUs x 256)
  printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
  printf("v_mfma_f32_32x32x2_f32: %.1f TFLOP/s  (matrix-pipe clock %.2f GHz)\n", t32, t32 * 1e12 / (p.multiProcessorCount * 256.0) / 1e9);
  printf("v_mfma_f32_16x16x4_f32: %.1f TFLOP/s  (matrix-pipe clock %.2f GHz)\n", t16, t16 * 1e12 / (p.multiProcessorCount * 256.0) / 1e9);
  hipFree(out);
  return 0;
}
