// fp32 "NT" linear for the decode chain's row counts (M = batch x beams, ~100 .. ~2000 rows) — wave-split-K form.
//
// What bounded the 64x64 split-K kernel (gemm_small.hip) at these sizes (profiles/r02_generate_kernel_stats.csv, r03 trace):
// a decode linear is 0.75-3 GFLOP, i.e. 5-20 us at the fp32 MFMA peak; to occupy 256 CUs the N = K = 768 projections were
// split 4 ways along K ACROSS workgroups, the partial slabs went to memory, and a second launch summed them and applied
// the norm that always follows — 8 us + 7.6 us for 4.8 us of MFMA work, MFMA busy 0.16-0.38.
//
// Here the K split happens INSIDE a workgroup: a workgroup owns a 32 x 64 output tile (640 x 768 -> 240 workgroups, one per
// CU, no slabs), its four waves take one quarter of K each and run as four independent streams —
//   * operands by LDS-DMA (global_load_lds_dwordx4) into a wave-PRIVATE 3-stage ring (12 KB per stage: 32 A rows + 64 W
//     rows x 128 B); the image is lane-linear, so the bank swizzle (16-byte chunk ^ ((row >> 1) & 7), conflict-free for
//     ds_read_b128 over 16 consecutive rows) is applied to the per-lane SOURCE address and again on the fragment reads;
//   * no workgroup barrier in the K loop: a wave waits for its own DMA with a counted s_waitcnt vmcnt (two stages stay in
//     flight), reads its fragments, issues 32 MFMAs (v_mfma_f32_32x32x2_f32, two 32x32 accumulators) per stage and refills
//     the stage it just consumed;
//   * at the end the four partial tiles meet in LDS, are summed in fixed order ((w0 + w1) + w2) + w3, and the epilogue
//     (residual add, ReLU) is applied with 32-byte row segments per thread.
// The RMS norm between two linears never runs as a kernel of its own: a producer's epilogue also writes, per row and
// 64-column tile, the sum of squares of what it stored (out_part[M][N/64]); the consumer adds a row's partials in fixed order,
// forms 1/sqrt(mean + eps) and scales its A fragments (x * rinv * w[k]) on the way from LDS to the MFMA
// (T5LayerNorm, modeling_t5.py:164-171: x / sqrt(mean(x^2) + eps) * w; the reciprocal-multiply differs from the division
// by at most one ulp per element, far inside the 1e-4 parity tolerance).
// Per decoder layer and step that leaves 8 launches (qkv, self-attention, o, q_c, cross-attention, o_c, wi, wo) where round 2
// ran 11, and no slab traffic.
#include <stdlib.h>

#include "common.h"

// M0 is written and read inside one asm statement (the LDS-DMA issue below); clang flags the clobber as "reserved"
#pragma clang diagnostic ignored "-Winline-asm"

namespace gdr {

typedef float f32x16w __attribute__((ext_vector_type(16)));

constexpr int WSK_BM = 32, WSK_BN = 64;
constexpr int WSK_PLD = WSK_BN + 4;  // padded row of a partial tile (floats)

struct WskArgs {
  const float* A;          // [M, lda]
  const float* W;          // [N, ldw]
  float* C;                // [M, ldc]
  const float* residual;   // [M, ldr] or null; may alias C
  const float* norm_part;  // [M, norm_nt] sums of squares of A's rows by 64-column tile, or null (A is used as it is)
  const float* norm_w;     // [K] T5LayerNorm weight (with norm_part)
  float* out_part;         // [M, tiles_n] sums of squares of the rows this launch stores, or null
  int64_t lda, ldw, ldc, ldr;
  int M, N, K, tiles_m, tiles_n, norm_nt;
  int relu;
  float eps;
};

// s_waitcnt vmcnt(stages * per_stage) for a small runtime `stages` (the immediate must be a constant)
template <int PER_STAGE>
__device__ __forceinline__ void wsk_wait_stages(int stages) {
  if (stages >= 3) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER_STAGE) : "memory");
  } else if (stages == 2) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_STAGE) : "memory");
  } else if (stages == 1) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// BK: k per stage and wave (32: 128-byte row segments, 12 KB stages; 16: 64-byte segments, 6 KB stages);
// NSTAGE: ring depth (<= 4); NW: waves per workgroup = K splits (4 or 8).  LDS per workgroup = NW * NSTAGE * 96 * BK * 4 bytes
// (+ the norm's K + NW * 32 * norm_nt floats): <32,3,4> 147 KB — one workgroup per CU, deepest prefetch; <16,2,4> 48 KB — three
// per CU, latency hidden by occupancy, and co-resident with the adaptor chain's kernels that run beside the decoder stack.
template <int BK, int NSTAGE, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_nt_f32_wsk_kernel(const WskArgs g) {
  constexpr int CH = BK / 4;                 // 16-byte chunks per row segment
  constexpr int RPI = 64 / CH;               // rows per DMA instruction (64 lanes x 16 B)
  constexpr int SW = CH == 8 ? 1 : 2;        // swizzle: chunk ^ ((row >> SW) & (CH - 1)) — conflict-free ds_read_b128 over 16 rows
  constexpr int STAGE_BYTES = (WSK_BM + WSK_BN) * BK * 4;
  constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
  constexpr int GPS = (WSK_BM + WSK_BN) / RPI;  // DMA instructions per stage
  static_assert(WSK_BM * WSK_PLD * 4 <= RING_BYTES, "the partial tile reuses the wave's ring");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4 && (NSTAGE - 1) * GPS <= 48, "vmcnt holds 6 bits; wsk_wait_stages serves <= 3 stages");
  extern __shared__ __attribute__((aligned(1024))) char wsk_smem[];  // NW rings, K norm weights, NW x 32 x norm_nt partial sums
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int mt = (int)(bid % (unsigned)g.tiles_m), nt = (int)(bid / (unsigned)g.tiles_m);  // neighbours share the W tile
  const int m0 = mt * WSK_BM, n0 = nt * WSK_BN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int Kw = g.K / NW, nst = Kw / BK;  // this wave's K slice, in stages
  const int k0 = wave * Kw;
  char* const ring = wsk_smem + wave * RING_BYTES;
  const int Kw64 = (Kw + 63) & ~63;  // a wave's norm-weight slice, padded to whole 64-lane DMA pieces
  float* const wn = reinterpret_cast<float*>(wsk_smem + NW * RING_BYTES) + wave * Kw64;
  float* const wp = reinterpret_cast<float*>(wsk_smem + NW * RING_BYTES) + NW * Kw64 + wave * (32 * g.norm_nt);

  // ---- DMA source addresses: instruction i covers tile rows RPI*i .. ; lane = (row_in_group, chunk') ----------------
  const int srow = lane / CH, schunk = lane % CH;
  const char* a_src[WSK_BM / RPI];
  const char* w_src[WSK_BN / RPI];
#pragma unroll
  for (int i = 0; i < WSK_BM / RPI; ++i) {
    const int row = RPI * i + srow;
    int ra = m0 + row;
    ra = ra < g.M ? ra : g.M - 1;  // rows past the edge are computed and discarded
    a_src[i] = reinterpret_cast<const char*>(g.A + (int64_t)ra * g.lda + k0) + ((schunk ^ ((row >> SW) & (CH - 1))) << 4);
  }
#pragma unroll
  for (int i = 0; i < WSK_BN / RPI; ++i) {
    const int row = RPI * i + srow;
    w_src[i] = reinterpret_cast<const char*>(g.W + (int64_t)(n0 + row) * g.ldw + k0) + ((schunk ^ ((row >> SW) & (CH - 1))) << 4);
  }
  // The DMA is issued from inline asm: hipcc (ROCm 7.2) otherwise treats every pending global_load_lds as a store that may
  // alias ANY later ds_read and puts s_waitcnt vmcnt(0) in front of the first fragment read of each stage — the ring would
  // drain every stage.  Invisible to the compiler's wait bookkeeping, the DMA is ordered by the counted waits written out
  // below (and by "memory" clobbers); M0 (the wave-uniform LDS base) is written in the statement that uses it.
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(ring));
#define WSK_GLDS(gptr_, lds_)                                                                                       \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_), "v"(gptr_) : "memory", "m0")
#define WSK_ISSUE(stage_, buf_)                                                                                      \
  {                                                                                                                  \
    const unsigned dst_ = ring_lds + (unsigned)(buf_)*STAGE_BYTES;                                                   \
    const int ko_ = (stage_)*BK * 4;                                                                                 \
    _Pragma("unroll") for (int i = 0; i < WSK_BM / RPI; ++i) WSK_GLDS(a_src[i] + ko_, dst_ + i * 1024);              \
    _Pragma("unroll") for (int i = 0; i < WSK_BN / RPI; ++i) WSK_GLDS(w_src[i] + ko_, dst_ + WSK_BM * BK * 4 + i * 1024); \
  }

  // ---- A-side RMS norm: the wave's slice of the norm weight and its 32 rows' partial sums of squares ride in by DMA too,
  //      ahead of the stages — every load of the main path is then ordered by the counted waits below, none by the compiler
  const int r31 = lane & 31, h = lane >> 5;
  if (g.norm_part) {
    const unsigned wn_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(wn));
    for (int e = 0; e < Kw; e += 64) {  // ceil(Kw / 64) instructions of 64 x 4 bytes, lane-linear; the tail lands in padding
      const int kk = k0 + e + lane;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(wn_lds + e * 4),
                   "v"(g.norm_w + (kk < g.K ? kk : g.K - 1))
                   : "memory", "m0");
    }
    int row = m0 + r31;
    row = row < g.M ? row : g.M - 1;
    const float* prow = g.norm_part + (int64_t)row * g.norm_nt + h;  // instruction i: partials 2i (lanes 0-31), 2i+1 (32-63)
    const unsigned wp_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(wp));
    for (int i = 0; 2 * i < g.norm_nt; ++i)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(wp_lds + i * 256), "v"(prow + 2 * i)
                   : "memory", "m0");
  }
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s)
    if (s < nst) WSK_ISSUE(s, s)
  float rinv = 1.0f;
  if (g.norm_part) {  // everything older than the stages has landed once only the stages' DMA is outstanding
    wsk_wait_stages<GPS>(nst < NSTAGE ? nst : NSTAGE);
    float ss = 0.f;
    for (int t = 0; t < g.norm_nt; ++t) ss += wp[(t >> 1) * 64 + (t & 1) * 32 + r31];  // fixed order t = 0 .. nt-1
    rinv = 1.0f / sqrtf(ss / (float)g.K + g.eps);
  }

  // ---- fragment offsets inside a stage: lane (r31, h) reads 16-byte chunk 2jj + h of row r31 (A) / r31, 32 + r31 (W) --
  int foff[CH / 2];
#pragma unroll
  for (int jj = 0; jj < CH / 2; ++jj) foff[jj] = r31 * (BK * 4) + (((2 * jj + h) ^ ((r31 >> SW) & (CH - 1))) << 4);

  f32x16w acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;

  int buf = 0;
  for (int s = 0; s < nst; ++s) {
    // stage s has landed when at most the DMA of the stages issued after it are outstanding
    const int later = nst - 1 - s;
    wsk_wait_stages<GPS>(later < NSTAGE - 1 ? later : NSTAGE - 1);
    const char* st = ring + buf * STAGE_BYTES;
    float4 av[CH / 2], b0[CH / 2], b1[CH / 2];
#pragma unroll
    for (int jj = 0; jj < CH / 2; ++jj) {
      av[jj] = *reinterpret_cast<const float4*>(st + foff[jj]);
      b0[jj] = *reinterpret_cast<const float4*>(st + WSK_BM * BK * 4 + foff[jj]);
      b1[jj] = *reinterpret_cast<const float4*>(st + WSK_BM * BK * 4 + 32 * BK * 4 + foff[jj]);
    }
    if (g.norm_part) {
#pragma unroll
      for (int jj = 0; jj < CH / 2; ++jj) {
        const float4 nw = *reinterpret_cast<const float4*>(wn + s * BK + 8 * jj + 4 * h);
        av[jj].x = nw.x * (av[jj].x * rinv), av[jj].y = nw.y * (av[jj].y * rinv);
        av[jj].z = nw.z * (av[jj].z * rinv), av[jj].w = nw.w * (av[jj].w * rinv);
      }
    }
#pragma unroll
    for (int jj = 0; jj < CH / 2; ++jj) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].x, b0[jj].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].x, b1[jj].x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].y, b0[jj].y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].y, b1[jj].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].z, b0[jj].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].z, b1[jj].z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].w, b0[jj].w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj].w, b1[jj].w, acc1, 0, 0, 0);
    }
    // every fragment of this stage is in registers (the MFMAs above consumed them): refill it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (s + NSTAGE < nst) WSK_ISSUE(s + NSTAGE, buf)
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
  }
#undef WSK_ISSUE
#undef WSK_GLDS

  // ---- the NW K-slice partials meet in LDS (each wave's own ring holds its tile: no barrier before the writes) --------
  float* P = reinterpret_cast<float*>(ring);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // accumulator map: col = lane & 31, row as here
    P[row * WSK_PLD + r31] = acc0[r];
    P[row * WSK_PLD + 32 + r31] = acc1[r];
  }
  __syncthreads();
  constexpr int TPR = 2 * NW, EPT = WSK_BN / TPR;  // threads per row, elements per thread (8 or 4)
  const int erow = tid / TPR, ecg = (tid % TPR) * EPT;
  float v[EPT];
  {
    const float* p0 = reinterpret_cast<const float*>(wsk_smem) + erow * WSK_PLD + ecg;
#pragma unroll
    for (int j = 0; j < EPT; j += 4) {
      const float4 x = *reinterpret_cast<const float4*>(p0 + j);
      v[j] = x.x, v[j + 1] = x.y, v[j + 2] = x.z, v[j + 3] = x.w;
    }
#pragma unroll
    for (int w = 1; w < NW; ++w) {  // ((w0 + w1) + w2) + ... in wave order
      const float* pw = reinterpret_cast<const float*>(wsk_smem + w * RING_BYTES) + erow * WSK_PLD + ecg;
#pragma unroll
      for (int j = 0; j < EPT; j += 4) {
        const float4 y = *reinterpret_cast<const float4*>(pw + j);
        v[j] += y.x, v[j + 1] += y.y, v[j + 2] += y.z, v[j + 3] += y.w;
      }
    }
  }
  const int m = m0 + erow, n = n0 + ecg;
  const bool live = m < g.M;
  if (g.residual && live) {
#pragma unroll
    for (int j = 0; j < EPT; j += 4) {
      const float4 r = *reinterpret_cast<const float4*>(g.residual + (int64_t)m * g.ldr + n + j);
      v[j] += r.x, v[j + 1] += r.y, v[j + 2] += r.z, v[j + 3] += r.w;
    }
  }
  if (g.relu) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < EPT; j += 4)
      *reinterpret_cast<float4*>(g.C + (int64_t)m * g.ldc + n + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
  }
  if (g.out_part) {  // sum of squares of the 64 stored values of this row: EPT per thread, then the TPR threads of the row
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) ss += v[j] * v[j];
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) ss += __shfl_xor(ss, o);
    if ((tid % TPR) == 0 && live) g.out_part[(int64_t)m * g.tiles_n + nt] = ss;
  }
}

struct WskCfg {
  int bk, stages, nw;
};
static size_t wsk_cfg_lds(const WskCfg& c, int K, int norm_nt) {
  const size_t kw64 = (size_t)((K / c.nw + 63) & ~63);
  return (size_t)c.nw * c.stages * (WSK_BM + WSK_BN) * c.bk * 4 + (norm_nt > 0 ? (c.nw * kw64 + (size_t)c.nw * 32 * norm_nt) * sizeof(float) : 0);
}
// Which form serves (M, N, K).  GDR_WSK_CFG="bk,stages,waves" overrides (lab use).
static WskCfg wsk_pick(int64_t M, int N, int K) {
  static const WskCfg forced = [] {
    WskCfg c{0, 0, 0};
    if (const char* e = getenv("GDR_WSK_CFG")) sscanf(e, "%d,%d,%d", &c.bk, &c.stages, &c.nw);
    return c;
  }();
  if (forced.bk) return forced;
  const int64_t tiles = ((M + WSK_BM - 1) / WSK_BM) * (N / WSK_BN);
  if (tiles <= 128 && K % (8 * 16) == 0) return WskCfg{16, 3, 8};  // few tiles (one query x 100 beams): split K eight ways
  return WskCfg{16, 3, 4};
}
size_t wsk_lds_bytes(int K, int norm_nt) { return wsk_cfg_lds(WskCfg{32, 3, 4}, K, norm_nt); }  // the largest form

// Returns 1 when the shape is not served here (the caller keeps its old path), 0 after launching, < 0 on error.
// norm_part / norm_w: fold T5LayerNorm(A) into the A operand (norm_nt partials per row); out_part: emit the partials of C.
int launch_linear_f32_wsk(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N, int K,
                          int relu, const float* residual, int64_t ldr, const float* norm_part, int norm_nt, const float* norm_w,
                          float eps, float* out_part, hipStream_t stream) {
  if (M <= 0) return 0;
  const WskCfg c = wsk_pick(M, N, K);
  if (!((c.bk == 16 || c.bk == 32) && c.stages >= 2 && c.stages <= 4 && (c.nw == 4 || c.nw == 8))) return 1;
  if (N % WSK_BN != 0 || K % (c.nw * c.bk) != 0 || M > (1 << 20)) return 1;
  if (lda % 4 != 0 || ldw % 4 != 0 || ldc % 4 != 0 || (residual && ldr % 4 != 0)) return 1;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 15) || (residual && ((uintptr_t)residual & 15))) return 1;
  if (norm_part && (!norm_w || norm_nt <= 0 || (norm_nt & 1) || ((uintptr_t)norm_w & 3))) return 1;
  const size_t lds = wsk_cfg_lds(c, K, norm_part ? norm_nt : 0);
  if (lds > 160 * 1024) return 1;
  WskArgs g{};
  g.A = A, g.W = W, g.C = C, g.residual = residual, g.norm_part = norm_part, g.norm_w = norm_w, g.out_part = out_part;
  g.lda = lda, g.ldw = ldw, g.ldc = ldc, g.ldr = ldr;
  g.M = (int)M, g.N = N, g.K = K, g.tiles_m = (int)((M + WSK_BM - 1) / WSK_BM), g.tiles_n = N / WSK_BN, g.norm_nt = norm_nt;
  g.relu = relu, g.eps = eps;
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
  ProfScope prof(PROF_LINEAR, 2.0 * (double)M * (double)N * (double)K, stream);
#define WSK_CASE(BK_, ST_, NW_)                                                                                              \
  if (c.bk == BK_ && c.stages == ST_ && c.nw == NW_) {                                                                       \
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(gemm_nt_f32_wsk_kernel<BK_, ST_, NW_>), 160 * 1024,            \
                                "linear(wave-split-K)"))                                                                     \
      return rc;                                                                                                             \
    hipLaunchKernelGGL((gemm_nt_f32_wsk_kernel<BK_, ST_, NW_>), grid, dim3(NW_ * 64), lds, stream, g);                       \
    GDR_CHECK_LAUNCH("gemm_nt_f32_wsk_kernel");                                                                              \
    return 0;                                                                                                                \
  }
  WSK_CASE(32, 3, 4)
  WSK_CASE(32, 2, 4)
  WSK_CASE(16, 2, 4)
  WSK_CASE(16, 3, 4)
  WSK_CASE(16, 4, 4)
  WSK_CASE(16, 2, 8)
  WSK_CASE(16, 3, 8)
  WSK_CASE(32, 2, 8)
#undef WSK_CASE
  return 1;
}

}  // namespace gdr

// Test / lab entry: C = epilogue(norm?(A) W^T) on the wave-split-K kernel alone (returns GDR_EINVAL when the shape is not
// served).  part_in [M, part_in_nt] and norm_w [K] select the fused T5LayerNorm of A; part_out [M, N/64] receives the sums
// of squares of the stored rows.
extern "C" int gdr_linear_f32_wsk(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                                  int K, int relu, const float* residual, int64_t ldr, const float* part_in, int part_in_nt,
                                  const float* norm_w, float eps, float* part_out, void* stream) {
  const int rc = gdr::launch_linear_f32_wsk(A, lda, W, ldw, C, ldc, M, N, K, relu, residual, ldr, part_in, part_in_nt, norm_w, eps,
                                            part_out, static_cast<hipStream_t>(stream));
  if (rc == 1) {
    gdr::set_error("linear_f32_wsk: shape M=%lld N=%d K=%d (N %% 64, K %% 128, 16-byte alignment) is not served", (long long)M, N, K);
    return GDR_EINVAL;
  }
  return rc;
}
