// Sustained fp32-MFMA ceiling of this box: every CU issues independent v_mfma_f32_32x32x2_f32 back to back from
// registers (no memory traffic).  Prints TFLOP/s and the in-kernel clock (s_memtime vs s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f + 0.5f, y = blockIdx.x * 1e-4f + 0.25f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// same output patch per wave (64x64 = 16 tiles of 16x16, 64 accumulator registers) on v_mfma_f32_16x16x4_f32
__global__ __launch_bounds__(256) void k16(float* out, int iters, unsigned long long* clk) {
  f32x4 a[16];
  for (int t = 0; t < 16; ++t) a[t] = (f32x4){0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f + 0.5f, y = blockIdx.x * 1e-4f + 0.25f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {  // 16 MFMAs of 1024 MAC = the work of 4 32x32x2 MFMAs x 2 -> run iters/2... keep flops equal below
#pragma unroll
    for (int t = 0; t < 16; ++t) a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32((t & 1) ? x : y, (t & 2) ? x : y, a[t], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int t = 0; t < 16; ++t) s += a[t][0] + a[t][1] + a[t][2] + a[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main(int argc, char** argv) {
  int wg_per_cu = argc > 1 ? atoi(argv[1]) : 1;
  const int blocks = 256 * wg_per_cu, iters = 20000;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, clk); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * 4096.0;
    printf("32x32x2: wg/cu=%d  %.2f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", wg_per_cu, ms, flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters, clk); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * 2048.0;
    printf("16x16x4: wg/cu=%d  %.2f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", wg_per_cu, ms, flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
  }
  return 0;
}
