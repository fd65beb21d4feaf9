// Lab copy of the 64x64-tile decode linear (gdr_amd/csrc/gemm_small.hip) with parts switched off, to see what the ~9 us that
// a launch costs beyond its K loop at M = 640 are made of.  Not product code.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/small_lab.hip -o tools/lab/small_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16m __attribute__((ext_vector_type(16)));
constexpr int SB = 64, SBK = 32, SLD = SBK + 4;

struct Args {
  const float* A;
  const float* W;
  float* C;
  int64_t lda, ldw, ldc, M;
  int N, K, tiles_n;
};

// FLAGS: 1 = no output stores; 2 = every K-step re-reads the first one (operands stay in L1/L2); 4 = no MFMAs;
//        8 = three register stages (loads of K-steps 0,1,2 issued together at the start)
template <int FLAGS>
__global__ __launch_bounds__(256, 4) void lab_kernel(const Args g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * SB * SLD];
  float* const As = smem;
  float* const Bs = smem + 2 * SB * SLD;
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const unsigned tile = bid;
  const int64_t m0 = (int64_t)(tile / (unsigned)g.tiles_n) * SB;
  const int n0 = (int)(tile % (unsigned)g.tiles_n) * SB;
  const int tid = threadIdx.x;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const float* a_src[2];
  const float* w_src[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    a_src[p] = g.A + (m0 + lrow + 32 * p) * g.lda + lcol;
    w_src[p] = g.W + (int64_t)(n0 + lrow + 32 * p) * g.ldw + lcol;
  }
  const int st_off = lrow * SLD + lcol;
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int a_rd = (wm * 32 + l31) * SLD + 4 * h;
  const int b_rd = (wn * 32 + l31) * SLD + 4 * h;
  const int nk = g.K / SBK;
  f32x16m acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
#define S_LOAD(R, kt_)                                                                       \
  {                                                                                          \
    int t_ = (kt_) < nk ? (kt_) : nk - 1;                                                    \
    if (FLAGS & 2) t_ = 0;                                                                   \
    R##a0 = *reinterpret_cast<const float4*>(a_src[0] + t_ * SBK);                           \
    R##a1 = *reinterpret_cast<const float4*>(a_src[1] + t_ * SBK);                           \
    R##b0 = *reinterpret_cast<const float4*>(w_src[0] + t_ * SBK);                           \
    R##b1 = *reinterpret_cast<const float4*>(w_src[1] + t_ * SBK);                           \
  }
#define S_STORE(R, buf_)                                                                     \
  {                                                                                          \
    float* as_ = As + (buf_)*SB * SLD + st_off;                                              \
    float* bs_ = Bs + (buf_)*SB * SLD + st_off;                                              \
    *reinterpret_cast<float4*>(as_) = R##a0, *reinterpret_cast<float4*>(as_ + 32 * SLD) = R##a1; \
    *reinterpret_cast<float4*>(bs_) = R##b0, *reinterpret_cast<float4*>(bs_ + 32 * SLD) = R##b1; \
  }
#define S_COMPUTE(buf_)                                                                      \
  {                                                                                          \
    const float* a = As + (buf_)*SB * SLD + a_rd;                                            \
    const float* b = Bs + (buf_)*SB * SLD + b_rd;                                            \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                       \
      const float4 av = *reinterpret_cast<const float4*>(a + 8 * jj);                        \
      const float4 bv = *reinterpret_cast<const float4*>(b + 8 * jj);                        \
      if (FLAGS & 4) {                                                                       \
        acc[jj] += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;                    \
      } else {                                                                               \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);                \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);                \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);                \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);                \
      }                                                                                      \
    }                                                                                        \
  }
  S_LOAD(p, 0)
  if (FLAGS & 8) {
    S_LOAD(q, 1)
  }
  S_STORE(p, 0)
  if (FLAGS & 8) {
    S_LOAD(p, 2)
  } else {
    S_LOAD(p, 1)
    S_LOAD(q, 2)
  }
  __syncthreads();
  if (FLAGS & 8) {  // order of stages: q holds 1, p holds 2
    for (int kt = 0; kt < nk; kt += 2) {
      S_COMPUTE(0)
      S_STORE(q, 1)
      S_LOAD(q, kt + 3)
      __syncthreads();
      if (kt + 1 >= nk) break;
      S_COMPUTE(1)
      S_STORE(p, 0)
      S_LOAD(p, kt + 4)
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; kt += 2) {
      S_COMPUTE(0)
      S_STORE(p, 1)
      S_LOAD(p, kt + 3)
      __syncthreads();
      if (kt + 1 >= nk) break;
      S_COMPUTE(1)
      S_STORE(q, 0)
      S_LOAD(q, kt + 4)
      __syncthreads();
    }
  }
  const int n = n0 + wn * 32 + l31;
  if (FLAGS & 1) {
    if (acc[0] == 123.456f) g.C[0] = acc[1];
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    g.C[m * g.ldc + n] = acc[r];
  }
}

template <int FLAGS>
static void run(const char* name, const Args& base, const std::vector<int>& Ks, int M, int N) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int tiles = (M / SB) * (N / SB);
  printf("%-46s", name);
  double t512 = 0, t3072 = 0;
  for (int K : Ks) {
    Args g = base;
    g.K = K;
    for (int i = 0; i < 10; ++i) (void)0, hipLaunchKernelGGL(lab_kernel<FLAGS>, dim3(tiles), dim3(256), 0, 0, g);
    hipDeviceSynchronize();
    const int reps = 200;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) (void)0, hipLaunchKernelGGL(lab_kernel<FLAGS>, dim3(tiles), dim3(256), 0, 0, g);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf(" K=%d: %6.2f", K, us);
    if (K == 512) t512 = us;
    if (K == 3072) t3072 = us;
  }
  const double slope = (t3072 - t512) / (3072 - 512) * 32;
  printf("  | %.3f us/K-step, intercept %.2f us\n", slope, t512 - slope * 16);
}

int main(int argc, char** argv) {
  const int M = 640, LD = 3072;
  float *A, *W, *C;
  hipMalloc(&A, (size_t)M * LD * 4), hipMalloc(&W, (size_t)6144 * LD * 4), hipMalloc(&C, (size_t)M * 6144 * 4);
  hipMemset(A, 0, (size_t)M * LD * 4), hipMemset(W, 0, (size_t)6144 * LD * 4);
  const std::vector<int> Ks = {128, 256, 512, 768, 3072};
  for (int N : {768, 1536, 3072, 6144}) {
    Args g{A, W, C, LD, LD, N, M, N, 0, N / SB};
    printf("N = %d (%d workgroups)\n", N, (M / SB) * (N / SB));
    run<0>("  as shipped", g, Ks, M, N);
    run<1>("  no output stores", g, Ks, M, N);
    run<2>("  operands: K-step 0 re-read", g, Ks, M, N);
    run<3>("  K-step 0 re-read + no stores", g, Ks, M, N);
    run<4>("  no MFMAs (VALU stand-in)", g, Ks, M, N);
    run<8>("  loads of K-steps 0,1 issued together", g, Ks, M, N);
  }
  return 0;
}
