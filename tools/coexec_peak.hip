// Feasibility probe: do the fp32 MFMA pipe and the fp32 VALU pipe (v_pk_fma_f32) sustain their rates TOGETHER on one
// SIMD?  512-thread workgroups, one per CU: waves 0-3 issue v_mfma_f32_32x32x2_f32 back to back, waves 4-7 issue
// v_pk_fma_f32 back to back (registers only).  Prints each role's TFLOP/s alone and together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 1 mfma only, 2 valu only, 3 both
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (MODE & 1) {
      f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      float x = threadIdx.x * 1e-3f + 0.5f, y = blockIdx.x * 1e-4f + 0.25f;
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
      for (int q = 0; q < 16; ++q) r += a0[q] + a1[q] + a2[q] + a3[q];
    }
  } else {
    if (MODE & 2) {
      // 32 independent packed accumulators; 64 v_pk_fma per MFMA-iteration-equivalent (4 MFMAs = 256 cycles = 64 pk_fma)
      f32x2 c[32];
      f32x2 a = {threadIdx.x * 1e-3f + 0.5f, 0.25f}, b = {0.999f, 1.001f};
#pragma unroll
      for (int q = 0; q < 32; ++q) c[q] = f32x2{(float)q, (float)-q};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int q = 0; q < 32; ++q) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a), "v"(b));
      }
      for (int q = 0; q < 32; ++q) r += c[q][0] + c[q][1];
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const double mf = 256.0 * 4 * iters * 4 * 4096.0;        // flops of the MFMA role
  const double vf = 256.0 * 4 * iters * 64 * 256.0;        // 64 pk_fma x 64 lanes x 2 FMA x 2 flop
  float t1 = run<1>(out, iters), t2 = run<2>(out, iters), t3 = run<3>(out, iters);
  printf("MFMA alone  %.2f ms  %.1f TF\n", t1, mf / t1 / 1e9);
  printf("VALU alone  %.2f ms  %.1f TF\n", t2, vf / t2 / 1e9);
  printf("both        %.2f ms  %.1f TF total (MFMA-equivalent share %.1f, VALU share %.1f)\n", t3, (mf + vf) / t3 / 1e9, mf / t3 / 1e9, vf / t3 / 1e9);
  return 0;
}
