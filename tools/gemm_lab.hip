// Kernel lab for the fp32 MFMA GEMM core: interleaved A/B timing of loop-structure variants in ONE process
// (cdna_hip_programming.md §5.4 rule 24).  Not product code; results feed gdr_amd/csrc/gemm_f32.hip.
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_lab.hip -o tools/gemm_lab && ./tools/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32, LS = BK + 4;

struct Args {
  const float* A; const float* W; float* C;
  int M, N, K, tiles_n;
};

// VAR: 0 baseline (product loop) | 1 no global loads / LDS writes (ablation) | 2 also no barrier | 3 one WG per CU
//      4 pipelined frag reads across barrier (no sched_barrier) | 5 baseline + stagger of odd workgroups
//      6 baseline with m-fastest supertile order | 7 staging one tile ahead, LDS writes + global loads
//      interleaved between MFMAs by sched_group_barrier, branch-free loop | 8 = 7 + supertile order
template <int VAR>
__global__ __launch_bounds__(256, 2) void gemm(const Args g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LS + 2 * BN * LS + (VAR == 3 ? 6 * 1024 : 0)];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LS;
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  int mt, nt;
  if (VAR == 6 || VAR == 8) {
    // supertile: groups of 8 m-tiles, n-tiles inside a group fastest over m (keeps 8 A panels + streams W once per group)
    const int tiles_m = gridDim.x / g.tiles_n, G = 8;
    const int per_group = G * g.tiles_n;
    const int grp = bid / per_group, in = bid % per_group;
    const int gm = min(G, tiles_m - grp * G);
    mt = grp * G + in % gm;
    nt = in / gm;
  } else {
    mt = bid / g.tiles_n;
    nt = bid % g.tiles_n;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lrow = tid >> 3, lcol = (tid & 7) * 4;
  const float* a_src[4];
  const float* w_src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    a_src[p] = g.A + (size_t)min(m0 + lrow + 32 * p, g.M - 1) * g.K + lcol;
    w_src[p] = g.W + (size_t)min(n0 + lrow + 32 * p, g.N - 1) * g.K + lcol;
  }
  const int st_off = lrow * LS + lcol;
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int a_rd = (wm * 64 + l31) * LS + 4 * h, b_rd = (wn * 64 + l31) * LS + 4 * h;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = g.K / BK;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GLOAD(kt_) do { const int koff = (kt_)*BK; \
    ra0 = *(const float4*)(a_src[0] + koff); ra1 = *(const float4*)(a_src[1] + koff); \
    ra2 = *(const float4*)(a_src[2] + koff); ra3 = *(const float4*)(a_src[3] + koff); \
    rb0 = *(const float4*)(w_src[0] + koff); rb1 = *(const float4*)(w_src[1] + koff); \
    rb2 = *(const float4*)(w_src[2] + koff); rb3 = *(const float4*)(w_src[3] + koff); } while (0)
#define LSTORE(buf_) do { float* a_ = As + (buf_)*BM * LS + st_off; float* b_ = Bs + (buf_)*BN * LS + st_off; \
    *(float4*)(a_) = ra0; *(float4*)(a_ + 32 * LS) = ra1; *(float4*)(a_ + 64 * LS) = ra2; *(float4*)(a_ + 96 * LS) = ra3; \
    *(float4*)(b_) = rb0; *(float4*)(b_ + 32 * LS) = rb1; *(float4*)(b_ + 64 * LS) = rb2; *(float4*)(b_ + 96 * LS) = rb3; } while (0)
#define READ(A0, A1, B0, B1, ap, bp, jj) \
    A0 = *(const float4*)((ap) + 8 * (jj)); A1 = *(const float4*)((ap) + 32 * LS + 8 * (jj)); \
    B0 = *(const float4*)((bp) + 8 * (jj)); B1 = *(const float4*)((bp) + 32 * LS + 8 * (jj));
#define MFMA4(A0, A1, B0, B1, x_) \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B0.x_, acc[0][0], 0, 0, 0); \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B1.x_, acc[0][1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B0.x_, acc[1][0], 0, 0, 0); \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B1.x_, acc[1][1], 0, 0, 0);
#define MFMA16(A0, A1, B0, B1) MFMA4(A0, A1, B0, B1, x) MFMA4(A0, A1, B0, B1, y) MFMA4(A0, A1, B0, B1, z) MFMA4(A0, A1, B0, B1, w)

  GLOAD(0);
  LSTORE(0);
  __syncthreads();
  if (VAR == 5 && (blockIdx.x & 8)) __builtin_amdgcn_s_sleep(100);  // desynchronise co-resident workgroups

  if (VAR == 9) {
    // v7 + next tile's first two fragment chunks read right behind the barrier, under the trailing MFMAs
    GLOAD(nk > 1 ? 1 : 0);
    float4 c0a0, c0a1, c0b0, c0b1, c1a0, c1a1, c1b0, c1b1, c2a0, c2a1, c2b0, c2b1, c3a0, c3a1, c3b0, c3b1;
    { const float* a = As + a_rd; const float* b = Bs + b_rd;
      READ(c0a0, c0a1, c0b0, c0b1, a, b, 0) READ(c1a0, c1a1, c1b0, c1b1, a, b, 1) }
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const float* a = As + buf * BM * LS + a_rd;
      const float* b = Bs + buf * BN * LS + b_rd;
      const float* an = As + (buf ^ 1) * BM * LS + a_rd;
      const float* bn = Bs + (buf ^ 1) * BN * LS + b_rd;
      LSTORE(buf ^ 1);
      const int nxt = kt + 2 < nk ? kt + 2 : nk - 1;
      MFMA16(c0a0, c0a1, c0b0, c0b1)
      READ(c2a0, c2a1, c2b0, c2b1, a, b, 2) READ(c3a0, c3a1, c3b0, c3b1, a, b, 3)
      GLOAD(nxt);
      MFMA16(c1a0, c1a1, c1b0, c1b1)
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      READ(c0a0, c0a1, c0b0, c0b1, an, bn, 0) READ(c1a0, c1a1, c1b0, c1b1, an, bn, 1)   // next tile (or stale, unused)
      __builtin_amdgcn_sched_barrier(0);
      MFMA16(c2a0, c2a1, c2b0, c2b1) MFMA16(c3a0, c3a1, c3b0, c3b1)
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (VAR == 7 || VAR == 8) {
    GLOAD(nk > 1 ? 1 : 0);   // staging registers now hold tile 1
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const float* a = As + buf * BM * LS + a_rd;
      const float* b = Bs + buf * BN * LS + b_rd;
      float4 c0a0, c0a1, c0b0, c0b1, c1a0, c1a1, c1b0, c1b1, c2a0, c2a1, c2b0, c2b1, c3a0, c3a1, c3b0, c3b1;
      READ(c0a0, c0a1, c0b0, c0b1, a, b, 0) READ(c1a0, c1a1, c1b0, c1b1, a, b, 1)
      LSTORE(buf ^ 1);                                   // tile kt+1 (valid or a harmless repeat of the last tile)
      const int nxt = kt + 2 < nk ? kt + 2 : nk - 1;
      MFMA16(c0a0, c0a1, c0b0, c0b1)
      READ(c2a0, c2a1, c2b0, c2b1, a, b, 2) READ(c3a0, c3a1, c3b0, c3b1, a, b, 3)
      GLOAD(nxt);
      MFMA16(c1a0, c1a1, c1b0, c1b1) MFMA16(c2a0, c2a1, c2b0, c2b1) MFMA16(c3a0, c3a1, c3b0, c3b1)
      // desired issue order for this basic block
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                 // 8 ds_read (chunks 0,1)
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);               // ds_write
      }
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // ds_read (chunks 2,3)
      }
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // global load
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 40, 0);
      __syncthreads();
    }
  } else if (VAR == 4) {
    float4 p0a0, p0a1, p0b0, p0b1, p1a0, p1a1, p1b0, p1b1, r0a0, r0a1, r0b0, r0b1, r1a0, r1a1, r1b0, r1b1;
    { const float* a = As + a_rd; const float* b = Bs + b_rd;
      READ(p0a0, p0a1, p0b0, p0b1, a, b, 0) READ(p1a0, p1a1, p1b0, p1b1, a, b, 1) }
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1; const bool more = kt + 1 < nk;
      const float* a = As + buf * BM * LS + a_rd; const float* b = Bs + buf * BN * LS + b_rd;
      READ(r0a0, r0a1, r0b0, r0b1, a, b, 2) READ(r1a0, r1a1, r1b0, r1b1, a, b, 3)
      if (more) GLOAD(kt + 1);
      MFMA16(p0a0, p0a1, p0b0, p0b1) MFMA16(p1a0, p1a1, p1b0, p1b1) MFMA16(r0a0, r0a1, r0b0, r0b1)
      if (more) LSTORE(buf ^ 1);
      __syncthreads();
      if (more) { const float* an = As + (buf ^ 1) * BM * LS + a_rd; const float* bn = Bs + (buf ^ 1) * BN * LS + b_rd;
        READ(p0a0, p0a1, p0b0, p0b1, an, bn, 0) READ(p1a0, p1a1, p1b0, p1b1, an, bn, 1) }
      MFMA16(r1a0, r1a1, r1b0, r1b1)
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = (VAR == 1 || VAR == 2) ? 0 : (kt & 1);
      const bool more = kt + 1 < nk;
      if (more && VAR != 1 && VAR != 2) GLOAD(kt + 1);
      const float* a = As + buf * BM * LS + a_rd;
      const float* b = Bs + buf * BN * LS + b_rd;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float4 a0, a1, b0, b1;
        READ(a0, a1, b0, b1, a, b, jj)
        MFMA16(a0, a1, b0, b1)
      }
      if (more && VAR != 1 && VAR != 2) LSTORE(buf ^ 1);
      if (VAR != 2) __syncthreads();
    }
  }
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
    if (n >= g.N) continue;
    for (int mi = 0; mi < 2; ++mi)
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M) g.C[(size_t)m * g.N + n] = acc[mi][ni][r];
      }
  }
}

template <int VAR>
static float run(const Args& a, hipEvent_t e0, hipEvent_t e1) {
  const int blocks = ((a.M + BM - 1) / BM) * a.tiles_n;
  hipEventRecord(e0);
  hipLaunchKernelGGL(gemm<VAR>, dim3(blocks), dim3(256), 0, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  struct Shape { int M, N, K; const char* name; } shapes[] = {
      {20480, 2304, 768, "qkv"}, {20480, 3072, 768, "wi"}, {20480, 768, 3072, "wo_ff"}, {20480, 768, 768, "o"},
      {320000, 512, 768, "sim"}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (auto& s : shapes) {
    float *A, *W, *C;
    hipMalloc(&A, (size_t)s.M * s.K * 4);
    hipMalloc(&W, (size_t)s.N * s.K * 4);
    hipMalloc(&C, (size_t)s.M * s.N * 4);
    std::vector<float> h((size_t)s.M * s.K);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    h.resize((size_t)s.N * s.K);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    Args a{A, W, C, s.M, s.N, s.K, (s.N + BN - 1) / BN};
    const double gf = 2.0 * s.M * s.N * s.K / 1e9;
    std::vector<float> t[10];
    for (int rep = 0; rep < 7; ++rep) {
      t[0].push_back(run<0>(a, e0, e1));
      t[1].push_back(run<1>(a, e0, e1));
      t[2].push_back(run<2>(a, e0, e1));
      t[3].push_back(run<3>(a, e0, e1));
      t[4].push_back(run<4>(a, e0, e1));
      t[5].push_back(run<5>(a, e0, e1));
      t[6].push_back(run<6>(a, e0, e1));
      t[7].push_back(run<7>(a, e0, e1));
      t[8].push_back(run<8>(a, e0, e1));
      t[9].push_back(run<9>(a, e0, e1));
    }
    {  // v7/v8 must produce the same C as v0 (bitwise: same k order)
      std::vector<float> c0((size_t)1 << 16), c7((size_t)1 << 16);
      run<0>(a, e0, e1); hipMemcpy(c0.data(), C + ((size_t)s.M * s.N - c0.size()), c0.size() * 4, hipMemcpyDeviceToHost);
      hipMemset(C, 0, (size_t)s.M * s.N * 4);
      run<9>(a, e0, e1); hipMemcpy(c7.data(), C + ((size_t)s.M * s.N - c7.size()), c7.size() * 4, hipMemcpyDeviceToHost);
      size_t bad = 0; for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c7[i];
      if (bad) printf("!! v9 differs from v0 in %zu of %zu checked outputs\n", bad, c0.size());
    }
    printf("%-6s M=%d N=%d K=%d (%.1f GFLOP):", s.name, s.M, s.N, s.K, gf);
    for (int v = 0; v < 10; ++v) {
      std::sort(t[v].begin(), t[v].end());
      printf("  v%d %.0f", v, gf / t[v][t[v].size() / 2]);
    }
    printf("  TFLOP/s (median of 7)\n");
    hipFree(A); hipFree(W); hipFree(C);
  }
  return 0;
}
