// fp32 "NT" linear for grids that cannot fill the chip with 128x128 tiles (decode: M = batch*beams rows).
//
// 64x64 output tiles: four times the workgroups of the 128x128 core for the same problem, so far fewer (or no) K splits
// are needed to occupy 256 CUs and the partial slabs that a split costs are a quarter of the size.  A decode-step linear
// is ~1 GFLOP: the launch is over in microseconds and what matters is how evenly the work lands on the SIMDs, not the
// last percent of a steady-state loop — hence the plain structure: 4 waves 2x2, one 32x32 accumulator tile per wave
// (v_mfma_f32_32x32x2_f32, same k order per output element as the big core), register-staged double-buffered LDS
// (36.9 KB -> 4 workgroups per CU), one barrier per K-step, latency hidden by occupancy.
// Split-K: block (tile, s) accumulates K range [s*kchunk, (s+1)*kchunk) and stores a raw 64x64 partial slab; a second
// kernel sums the slabs in fixed order s = 0..S-1 (deterministic) and applies the epilogue.
#include <stdlib.h>

#include "common.h"

namespace gdr {

typedef float f32x16m __attribute__((ext_vector_type(16)));

constexpr int SB = 64, SBK = 32, SLD = SBK + 4;

struct SmallGemmArgs {
  const float* A;
  const float* W;
  float* C;        // output, or the slab scratch when ksplit > 1
  const float* bias;
  const float* residual;
  int64_t lda, ldw, ldc, ldr;
  int64_t M;
  int N, K, tiles_n, tiles_m;
  int has_bias, has_residual, act;  // 0 none, 1 relu, 2 gelu
  int ksplit, kchunk;               // kchunk in elements, multiple of 32
  const int64_t* m_dev;             // may be null: live row count on the device (<= M); tiles past it exit at once
  const int32_t* live;              // may be null: *live == 0 -> the whole launch exits (common.h StreamK::live)
};

__device__ __forceinline__ float gelu_erf_s(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256, 4) void gemm_nt_f32_small_kernel(const SmallGemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * SB * SLD];
  float* const As = smem;
  float* const Bs = smem + 2 * SB * SLD;
  unsigned bid = blockIdx.x;
  const int64_t Mrows = g.m_dev ? *g.m_dev : g.M;
  // live row tiles: with a device-side row count the grid is sized for M and the remap below (a contiguous logical range per
  // XCD) runs over the live workgroups only — otherwise the tiles that exit would all belong to the last XCDs
  const unsigned tiles_m = g.m_dev ? (unsigned)((Mrows + SB - 1) / SB) : (unsigned)g.tiles_m;
  {
    unsigned nblk = gridDim.x;
    const unsigned live = tiles_m * (unsigned)g.tiles_n * (unsigned)g.ksplit;
    if (live < nblk) {
      if (bid >= live) return;  // uniform
      nblk = live;
    }
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  // bid is now contiguous per XCD.  Row tile fastest, then column tile, K split slowest: the n/8 workgroups of an XCD
  // then touch every row panel of A but only n/(8*tiles_m) column panels of W, all over ONE K range — the footprint that
  // its L2 pulls through the Infinity Cache is (tiles_m + n/(8*tiles_m)) panels of kchunk instead of all of W over all of K.
  // (Measured on MI355X: generate() is unchanged to +-0.3 % against the old split-fastest, row-major order — these launches
  // are not bound by where their operands come from, profiles/r03_small_gemm_tile_order_ab.txt — so this is traffic, not time.)
  const unsigned tiles = tiles_m * (unsigned)g.tiles_n, t = bid % tiles;
  const int split = (int)(bid / tiles);
  const unsigned tile = (t % tiles_m) * (unsigned)g.tiles_n + t / tiles_m;  // slab index stays row-major
  const int64_t m0 = (int64_t)(tile / (unsigned)g.tiles_n) * SB;
  const int n0 = (int)(tile % (unsigned)g.tiles_n) * SB;
  if (m0 >= Mrows) return;  // uniform
  const int tid = threadIdx.x;
  // staging: 8 lanes cover one 128-B row segment, 32 rows per pass, 2 passes per operand
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const float* a_src[2];
  const float* w_src[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int64_t ra = m0 + lrow + 32 * p;
    ra = ra < Mrows ? ra : Mrows - 1;
    int rw = n0 + lrow + 32 * p;
    rw = rw < g.N ? rw : g.N - 1;
    a_src[p] = g.A + ra * g.lda + lcol + (int64_t)split * g.kchunk;
    w_src[p] = g.W + (int64_t)rw * g.ldw + lcol + (int64_t)split * g.kchunk;
  }
  const int st_off = lrow * SLD + lcol;
  const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int a_rd = (wm * 32 + l31) * SLD + 4 * h;
  const int b_rd = (wn * 32 + l31) * SLD + 4 * h;
  int klen = g.K - split * g.kchunk;
  klen = klen < g.kchunk ? klen : g.kchunk;
  const int nk = klen / SBK;  // the launcher guarantees K % 32 == 0

  f32x16m acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // Two register stages: the loads of K-step kt+2 are issued while kt is computed and kt+1 is written to LDS, so a
  // load has two K-steps (plus the other resident workgroups) to arrive.
  float4 pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
#define S_LOAD(R, kt_)                                                                       \
  {                                                                                          \
    const int t_ = (kt_) < nk ? (kt_) : nk - 1; /* past the end: re-fetch the last tile */   \
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
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) { /* k permuted inside chunks of 8 as in gemm_f32.hip */ \
      const float4 av = *reinterpret_cast<const float4*>(a + 8 * jj);                        \
      const float4 bv = *reinterpret_cast<const float4*>(b + 8 * jj);                        \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);                  \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);                  \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);                  \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);                  \
    }                                                                                        \
  }
  const int live_v = g.live ? *g.live : 1;  // scalar load: its latency hides behind the first operand loads
  S_LOAD(p, 0)
  if (live_v == 0) return;  // uniform; every query of the generate call is done, the step's output is never read
  S_STORE(p, 0)
  S_LOAD(p, 1)
  S_LOAD(q, 2)
  __syncthreads();
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
#undef S_LOAD
#undef S_STORE
#undef S_COMPUTE
  // accumulator map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (g.ksplit > 1) {
    float* slab = g.C + ((int64_t)tile * g.ksplit + split) * (SB * SB);
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * SB + wn * 32 + l31] = acc[r];
    return;
  }
  const int n = n0 + wn * 32 + l31;
  if (n >= g.N) return;
  const float bia = g.has_bias ? g.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (m >= Mrows) continue;
    float v = acc[r] + bia;
    if (g.has_residual) v += g.residual[m * g.ldr + n];
    if (g.act == 1) v = fmaxf(v, 0.f);
    if (g.act == 2) v = gelu_erf_s(v);
    g.C[m * g.ldc + n] = v;
  }
}

// C tile = epilogue(sum_s slab[tile][s]) in fixed order.  4 blocks per 64x64 tile, a thread owns 4 consecutive columns.
__global__ __launch_bounds__(256) void splitk_reduce_small_kernel(const float* __restrict__ partial, int S, int tiles_n,
                                                                  int64_t M, int N, float* __restrict__ C, int64_t ldc,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ residual, int64_t ldr,
                                                                  int act, const int64_t* __restrict__ m_dev,
                                                                  const int32_t* __restrict__ live) {
  if (live && *live == 0) return;  // every query of the generate call is done (StreamK::live): the slabs were never written
  if (m_dev) M = *m_dev;
  const int tile = blockIdx.x >> 2;
  const int e = ((blockIdx.x & 3) << 8) + threadIdx.x;  // float4 index inside the tile: 64 rows x 16 float4
  const int r = e >> 4, c = (e & 15) << 2;
  const int64_t m = (int64_t)(tile / tiles_n) * SB + r;
  const int n = (tile % tiles_n) * SB + c;
  if (m >= M || n >= N) return;
  const float* p = partial + (int64_t)tile * S * (SB * SB) + r * SB + c;
  float4 v = *reinterpret_cast<const float4*>(p);
  int s = 1;
  for (; s + 4 <= S; s += 4) {  // fixed summation order s = 0..S-1, four slab loads in flight per round trip
    const float4 t0 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 0) * (SB * SB));
    const float4 t1 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 1) * (SB * SB));
    const float4 t2 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 2) * (SB * SB));
    const float4 t3 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 3) * (SB * SB));
    v.x += t0.x, v.y += t0.y, v.z += t0.z, v.w += t0.w;
    v.x += t1.x, v.y += t1.y, v.z += t1.z, v.w += t1.w;
    v.x += t2.x, v.y += t2.y, v.z += t2.z, v.w += t2.w;
    v.x += t3.x, v.y += t3.y, v.z += t3.z, v.w += t3.w;
  }
  for (; s < S; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)s * (SB * SB));
    v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
  }
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (n + q >= N) break;
    float x = o[q];
    if (bias) x += bias[n + q];
    if (residual) x += residual[m * ldr + n + q];
    if (act == 1) x = fmaxf(x, 0.f);
    if (act == 2) x = gelu_erf_s(x);
    C[m * ldc + n + q] = x;
  }
}

// Split-K reduction of a full-row output (N = the model width, <= 1024) fused with what always follows it on the decode
// path: epilogue (bias, residual, activation) -> C, then a norm of the finished row -> Y.  ONE ROW PER WORKGROUP (thread <->
// float4 column): every slab load of a row (S <= 8) is in flight at once on M workgroups, the row stays in registers between
// the reduction and the norm, and the row statistics are combined across the four waves through LDS.  Same summation order
// as splitk_reduce_small_kernel (slabs s = 0..S-1, then bias, then residual) and the same arithmetic as rmsnorm_kernel /
// layernorm_kernel (layers.hip).  (A wave-per-row form served round 2's first version; measured slower at every row count
// of the decode path, 10 .. 640 rows: -2 .. -7 % per generate() for this one — removed.)
//   kind 1: Y = w1 * x / sqrt(mean(x^2) + eps)                               (T5LayerNorm, modeling_t5.py:164-171)
//   kind 2: Y = LN(x; w1, b1)                                                (torch.nn.LayerNorm)
//   kind 3: Y = LN(LN(x; w1, b1) + addv; w2, b2)                             (the adaptor's norm1 -> norm2 with its constant
//                                                                             single-key cross-attention output in between)
__global__ __launch_bounds__(256) void splitk_reduce_norm_row_kernel(const float* __restrict__ partial, int S, int tiles_n,
                                                                    int64_t M, int N, float* __restrict__ C, int64_t ldc,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ residual, int64_t ldr, int act,
                                                                    const int64_t* __restrict__ m_dev, const NormEpilogue ne,
                                                                    const int32_t* __restrict__ live) {
  __shared__ float red[4];
  if (live && *live == 0) return;  // uniform: every query of the generate call is done
  if (m_dev) M = *m_dev;
  const int64_t m = blockIdx.x;
  if (m >= M) return;  // uniform
  const int tid = threadIdx.x, n4 = N >> 2, wave = tid >> 6;
  const bool on = tid < n4;
  const int tm = (int)(m / SB), r = (int)(m - (int64_t)tm * SB);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on) {
    const int n = tid << 2, tn = n / SB, c = n - tn * SB;
    const float* p = partial + ((int64_t)(tm * tiles_n + tn) * S) * (SB * SB) + r * SB + c;
    float4 b = v, q = v, t[8];
    if (bias) b = *reinterpret_cast<const float4*>(bias + n);
    if (residual) q = *reinterpret_cast<const float4*>(residual + m * ldr + n);
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < S) t[s] = *reinterpret_cast<const float4*>(p + (int64_t)s * (SB * SB));
    v = t[0];
#pragma unroll
    for (int s = 1; s < 8; ++s)
      if (s < S) v.x += t[s].x, v.y += t[s].y, v.z += t[s].z, v.w += t[s].w;
    for (int s = 8; s < S; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(p + (int64_t)s * (SB * SB));
      v.x += u.x, v.y += u.y, v.z += u.z, v.w += u.w;
    }
    if (bias) v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
    if (residual) v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
    if (act == 1) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    *reinterpret_cast<float4*>(C + m * ldc + n) = v;
  }
  auto block_sum = [&](float x) {  // every thread gets the sum over the workgroup (fixed order: lanes by butterfly, waves 0..3)
    x = wave_sum(x);
    __syncthreads();  // the previous reduction's reads are done
    if ((tid & 63) == 0) red[wave] = x;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
  };
  float* yr = ne.Y + m * ne.ldy;
  if (ne.kind == 1) {
    const float ss = block_sum(on ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : 0.f);
    const RowDivisor over(sqrtf(ss / (float)N + ne.eps));  // x / denom, bit for bit (common.h)
    if (on) {
      const float4 g = reinterpret_cast<const float4*>(ne.w1)[tid];
      *reinterpret_cast<float4*>(yr + 4 * tid) = make_float4(g.x * over(v.x), g.y * over(v.y), g.z * over(v.z), g.w * over(v.w));
    }
    return;
  }
  const float inv_d = 1.0f / (float)N;
  auto layer_norm = [&](const float* w, const float* b) {
    const float mean = block_sum(on ? v.x + v.y + v.z + v.w : 0.f) * inv_d;
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    const float rstd = 1.0f / sqrtf(block_sum(on ? a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 : 0.f) * inv_d + ne.eps);
    if (on) {
      const float4 g = reinterpret_cast<const float4*>(w)[tid], bb = reinterpret_cast<const float4*>(b)[tid];
      v = make_float4(a0 * rstd * g.x + bb.x, a1 * rstd * g.y + bb.y, a2 * rstd * g.z + bb.z, a3 * rstd * g.w + bb.w);
    }
  };
  layer_norm(ne.w1, ne.b1);
  if (ne.kind == 3) {
    if (on) {
      const float4 t = reinterpret_cast<const float4*>(ne.addv)[tid];
      v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
    }
    layer_norm(ne.w2, ne.b2);
  }
  if (on) *reinterpret_cast<float4*>(yr + 4 * tid) = v;
}

// Returns 1 when the shape is left to the 128x128 core, 0 after launching, < 0 on error.
int launch_linear_f32_small(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                            int K, int has_bias, int has_residual, int act, const float* bias, const float* residual,
                            int64_t ldr, float* ws, size_t ws_bytes, hipStream_t stream, const int64_t* m_dev,
                            const NormEpilogue* ne, SlabRef* slabs, const int32_t* live) {
  constexpr int target = 512;  // workgroups wanted per launch (measured flat between 256 and 512, worse outside)
  if (K % SBK != 0) return 1;
  const int64_t tiles_m = (M + SB - 1) / SB;
  const int tiles_n = (N + SB - 1) / SB;
  const int64_t tiles = tiles_m * tiles_n;
  if (tiles > 4096) return 1;
  const int nk = K / SBK;
  int S = (int)((target + tiles / 2) / tiles);  // nearest
  if (S > nk / 4) S = nk / 4;
  if (S < 1) S = 1;
  const size_t slab = (size_t)SB * SB * sizeof(float);
  if (S > 1 && (!ws || (size_t)S * tiles * slab > ws_bytes)) S = ws ? (int)(ws_bytes / (tiles * slab)) : 1;
  if (S < 1) S = 1;
  const int chunk_steps = (nk + S - 1) / S;
  S = (nk + chunk_steps - 1) / chunk_steps;
  SmallGemmArgs g{};
  g.A = A, g.W = W, g.bias = bias, g.residual = residual;
  g.lda = lda, g.ldw = ldw, g.ldc = ldc, g.ldr = ldr, g.M = M, g.N = N, g.K = K, g.tiles_n = tiles_n;
  g.ksplit = S, g.kchunk = chunk_steps * SBK;
  g.m_dev = m_dev, g.live = live;
  g.tiles_m = (int)tiles_m;
  const double flops = 2.0 * (double)M * (double)N * (double)K;
  if (ne && (S == 1 || N % 4 != 0 || N > 1024 || act > 1)) return 2;  // the fused norm needs split slabs of a row it can hold
  if (slabs) {
    if (has_bias || has_residual || act || ne) return 1;  // slabs carry raw partial sums only
    slabs->part = nullptr, slabs->S = 1, slabs->tiles_n = tiles_n;
  }
  if (S == 1) {
    g.C = C, g.has_bias = has_bias, g.has_residual = has_residual, g.act = act;
    ProfScope prof(PROF_LINEAR, flops, stream);
    hipLaunchKernelGGL(gemm_nt_f32_small_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, g);
    GDR_CHECK_LAUNCH("gemm_nt_f32_small_kernel");
    return 0;
  }
  g.C = ws;
  {
    ProfScope prof(PROF_LINEAR, flops, stream);
    hipLaunchKernelGGL(gemm_nt_f32_small_kernel, dim3((unsigned)(tiles * S)), dim3(256), 0, stream, g);
  }
  GDR_CHECK_LAUNCH("gemm_nt_f32_small_kernel(split)");
  if (slabs) {  // the consumer reduces
    slabs->part = ws, slabs->S = S, slabs->tiles_n = tiles_n;
    return 0;
  }
  ProfScope prof_r(PROF_REDUCE, 0.0, stream);
  if (ne) {
    const float* bp = has_bias ? bias : nullptr;
    const float* rp = has_residual ? residual : nullptr;
    hipLaunchKernelGGL(splitk_reduce_norm_row_kernel, dim3((unsigned)M), dim3(256), 0, stream, ws, S, tiles_n, M, N, C, ldc, bp, rp,
                       ldr, act, m_dev, *ne, live);
    GDR_CHECK_LAUNCH("splitk_reduce_norm_row_kernel");
    return 0;
  }
  hipLaunchKernelGGL(splitk_reduce_small_kernel, dim3((unsigned)(tiles * 4)), dim3(256), 0, stream, ws, S, tiles_n, M, N, C, ldc,
                     has_bias ? bias : nullptr, has_residual ? residual : nullptr, ldr, act, m_dev, live);
  GDR_CHECK_LAUNCH("splitk_reduce_small_kernel");
  return 0;
}

}  // namespace gdr
