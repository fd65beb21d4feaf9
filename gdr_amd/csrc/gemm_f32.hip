// fp32 "NT" GEMM core for gfx950:  C[M,N] = epilogue(A[M,K] · W[N,K]^T), both operands K-contiguous.
//
// One kernel serves every dense contraction on GDR's inference path (reference call sites in
// include/gdr_hip.h): the T5 / adaptor linears and — with the docs in the row role and the queries in
// the column role — the corpus similarity Q·Dᵀ, whose epilogue filters scores against a per-query
// threshold instead of storing them (sim_topk.hip).
//
// CDNA4 mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD): 256-thread workgroup = 4 waves as 2x2, each
//     wave owns a 64x64 output patch = 2x2 MFMA tiles (64 accumulator VGPRs), block tile 128x128, BK 32.
//   * operands staged global -> VGPR (16 B/lane, full 128-B lines) -> LDS, double-buffered, one barrier
//     per K-step; LDS rows padded to 36 floats so the ds_read_b128 fragment reads are conflict-free
//     (row*36 mod 64 hits 16 distinct 4-bank slots for any 16-lane read group).
//   * one ds_read_b128 feeds 4 MFMAs: the k index inside a K-chunk of 8 is permuted (lane half h reads
//     k = 8j+4h..+3, MFMA s pairs element s of both halves) — the same permutation on both operands,
//     so the sum over k is unchanged.
//   * 1-D grid with the bijective XCD remap: the blocks that share a row panel (A tile) run on one XCD
//     back to back, so the panel is fetched from HBM once and re-read from that XCD's L2.
#include <stdlib.h>

#include "common.h"

namespace gdr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_STRIDE = BK + 4;  // floats; 144 B rows, 16-B aligned
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
  const float* A;
  const float* W;
  float* C;
  const float* bias;
  const float* residual;
  int64_t lda, ldw, ldc, ldr;
  int64_t M;
  int N, K;
  int tiles_n;
  int has_bias, has_residual, act;  // linear epilogue, runtime so that every linear shares one kernel
  int ksplit, kchunk;               // split-K: block handles K range [s*kchunk, (s+1)*kchunk) of its tile and stores a raw
                                    // 128x128 partial at C[(local_tile*ksplit + s)][128][128] (C = scratch)
  int tile_base;                    // first linear tile index of this launch (tail launches start past the full rounds)
  const int64_t* m_dev;             // linear only, may be null: the row count lives on the device (*m_dev <= M, ragged
                                    // batches, encoder.hip); the grid is sized for M and tiles past *m_dev exit at once
  SimEpilogue sim;
};

// Device-side row count of a packed batch, rounded UP to whole 128-row panels (never past the buffers' M rows): the rows
// between the live count and the panel's end are scratch — nothing reads them — so the last panel runs the interior tile
// path like every other instead of the clamped / masked edge path (20 live rows in the 97th panel of the C2 batch cost
// 1.5 % of the linears' time).  Live rows are unaffected: a GEMM row depends on nothing but itself.
__device__ __forceinline__ int64_t live_rows_padded(const GemmArgs& g) {
  if (!g.m_dev) return g.M;
  // the count is the same for every lane, but a plain global load leaves it (and every tile / segment index derived from it) in
  // VGPRs: read it through the scalar unit's view so that the whole tile bookkeeping lives in SGPRs
  const int64_t raw = *g.m_dev;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(raw & 0xffffffffll));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)raw >> 32));
  const int64_t live = (int64_t)(((unsigned long long)hi << 32) | lo);
  const int64_t up = (live + (BM - 1)) / BM * BM;
  return up < g.M ? up : g.M;
}


enum { EPI_LINEAR = 0, EPI_SIM_SAMPLE = 100, EPI_SIM_FILTER = 101 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// BF16 = false: fp32 operands on v_mfma_f32_32x32x2_f32.  BF16 = true: bf16 operands (fp32 accumulate) on
// v_mfma_f32_32x32x16_bf16.  A K-step is 128 BYTES of every row either way (32 floats / 64 bf16), so staging, LDS
// image and fragment addressing are byte-identical; only the MFMA issued per 16-byte fragment pair differs.
template <int EPI, bool BF16 = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_f32_kernel(const GemmArgs g) {
  constexpr int ES = BF16 ? 2 : 4;       // element size
  constexpr int BKE = 128 / ES;          // elements per K-step
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDS_STRIDE + 2 * BN * LDS_STRIDE];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LDS_STRIDE;

  // ---- block -> tile (XCD-aware, bijective for any grid size) ----
  unsigned bid = blockIdx.x;
  int64_t Mrows = g.M;
  {
    // With a device-side row count the grid is sized for the padded batch and the tiles past the live rows are the last
    // logical ones: the remap (a contiguous logical range per XCD) runs over the LIVE block count, or the last XCDs would get
    // nothing but tiles that exit (gemm_bf16.hip has the measurement).  Hardware blocks past the live count exit.
    unsigned nblk = gridDim.x;
    if (EPI == EPI_LINEAR && g.m_dev) {
      Mrows = live_rows_padded(g);
      const int64_t live = (((Mrows + BM - 1) / BM) * (int64_t)g.tiles_n - g.tile_base) * (g.ksplit > 1 ? g.ksplit : 1);
      if (live < (int64_t)nblk) {
        if ((int64_t)bid >= live) return;  // uniform
        nblk = (unsigned)live;
      }
    }
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  int split = 0;
  if (EPI == EPI_LINEAR && g.ksplit > 1) {  // splits of one tile are neighbours in the grid
    split = bid % (unsigned)g.ksplit;
    bid /= (unsigned)g.ksplit;
  }
  const unsigned local_tile = bid;
  if (EPI == EPI_LINEAR) bid += (unsigned)g.tile_base;
  int64_t mt = bid / (unsigned)g.tiles_n;
  const int nt = bid % (unsigned)g.tiles_n;
  int64_t slot_base = 0;
  if (EPI == EPI_SIM_SAMPLE) {
    slot_base = mt * BM;
    mt = mt * g.sim.tile_stride;
  } else if (EPI == EPI_SIM_FILTER) {
    const int s1 = g.sim.tile_stride - 1;
    mt = (mt / s1) * g.sim.tile_stride + 1 + (mt % s1);
  }
  const int64_t m0 = mt * BM;
  const int n0 = nt * BN;
  if (EPI == EPI_LINEAR && g.m_dev && m0 >= Mrows) return;  // uniform: a tile past the device-side row count

  const int tid = threadIdx.x;
  // ---- staging map: 8 lanes cover one 128-B row segment, 32 rows per pass, 4 passes ----
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;        // LDS column in floats (16-byte chunk index * 4)
  const int lcole = (tid & 7) * (16 / ES);  // the same chunk in elements
  const char* a_src[4];
  const char* w_src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int64_t ra = m0 + lrow + 32 * p;
    ra = ra < Mrows ? ra : Mrows - 1;  // clamp: rows past the edge are computed and discarded
    int rw = n0 + lrow + 32 * p;
    rw = rw < g.N ? rw : g.N - 1;
    a_src[p] = reinterpret_cast<const char*>(g.A) + (ra * g.lda + lcole + split * g.kchunk) * ES;
    w_src[p] = reinterpret_cast<const char*>(g.W) + ((int64_t)rw * g.ldw + lcole + split * g.kchunk) * ES;
  }
  const int st_off = lrow * LDS_STRIDE + lcol;

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int a_rd = (wm * 64 + l31) * LDS_STRIDE + 4 * h;
  const int b_rd = (wn * 64 + l31) * LDS_STRIDE + 4 * h;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int klen = g.K;
  if (EPI == EPI_LINEAR && g.ksplit > 1) klen = min(g.kchunk, g.K - split * g.kchunk);
  const int nk = (klen + BKE - 1) / BKE;
  // K tail (K % 32 != 0, K % 4 == 0): loads past K re-read the row's last float4 (a valid address) and the
  // VALUE is zeroed — never select between addresses, that demotes the loads to flat + scratch.
  const bool ktail = (klen % BKE) != 0;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;

#define GDR_GLOAD(kt_)                                                                    \
  do {                                                                                    \
    int koff = (kt_)*128; /* bytes */                                                     \
    bool ok = true;                                                                       \
    if (ktail) {                                                                          \
      ok = (kt_)*BKE + lcole < klen;                                                      \
      koff = ok ? koff : (klen - 16 / ES - lcole) * ES;                                   \
    }                                                                                     \
    ra0 = *reinterpret_cast<const float4*>(a_src[0] + koff);                              \
    ra1 = *reinterpret_cast<const float4*>(a_src[1] + koff);                              \
    ra2 = *reinterpret_cast<const float4*>(a_src[2] + koff);                              \
    ra3 = *reinterpret_cast<const float4*>(a_src[3] + koff);                              \
    rb0 = *reinterpret_cast<const float4*>(w_src[0] + koff);                              \
    rb1 = *reinterpret_cast<const float4*>(w_src[1] + koff);                              \
    rb2 = *reinterpret_cast<const float4*>(w_src[2] + koff);                              \
    rb3 = *reinterpret_cast<const float4*>(w_src[3] + koff);                              \
    if (ktail && !ok) {                                                                   \
      ra0 = ra1 = ra2 = ra3 = make_float4(0.f, 0.f, 0.f, 0.f);                            \
      rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);                            \
    }                                                                                     \
  } while (0)

#define GDR_LSTORE(buf_)                                                                  \
  do {                                                                                    \
    float* a_ = As + (buf_)*BM * LDS_STRIDE + st_off;                                     \
    float* b_ = Bs + (buf_)*BN * LDS_STRIDE + st_off;                                     \
    *reinterpret_cast<float4*>(a_) = ra0;                                                 \
    *reinterpret_cast<float4*>(a_ + 32 * LDS_STRIDE) = ra1;                               \
    *reinterpret_cast<float4*>(a_ + 64 * LDS_STRIDE) = ra2;                               \
    *reinterpret_cast<float4*>(a_ + 96 * LDS_STRIDE) = ra3;                               \
    *reinterpret_cast<float4*>(b_) = rb0;                                                 \
    *reinterpret_cast<float4*>(b_ + 32 * LDS_STRIDE) = rb1;                               \
    *reinterpret_cast<float4*>(b_ + 64 * LDS_STRIDE) = rb2;                               \
    *reinterpret_cast<float4*>(b_ + 96 * LDS_STRIDE) = rb3;                               \
  } while (0)

  GDR_GLOAD(0);
  GDR_LSTORE(0);
  __syncthreads();

#define GDR_READ(A0, A1, B0, B1, ap, bp, jj)                                         \
  A0 = *reinterpret_cast<const float4*>((ap) + 8 * (jj));                             \
  A1 = *reinterpret_cast<const float4*>((ap) + 32 * LDS_STRIDE + 8 * (jj));           \
  B0 = *reinterpret_cast<const float4*>((bp) + 8 * (jj));                             \
  B1 = *reinterpret_cast<const float4*>((bp) + 32 * LDS_STRIDE + 8 * (jj));
#define GDR_MFMA4(A0, A1, B0, B1, x_)                                                 \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B0.x_, acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B1.x_, acc[0][1], 0, 0, 0); \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B0.x_, acc[1][0], 0, 0, 0); \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B1.x_, acc[1][1], 0, 0, 0);
#define GDR_MFMA_BF(A, B, C) \
  C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, B), C, 0, 0, 0);
#define GDR_MFMA16(A0, A1, B0, B1)                                                                                   \
  if constexpr (BF16) {                                                                                              \
    GDR_MFMA_BF(A0, B0, acc[0][0]) GDR_MFMA_BF(A0, B1, acc[0][1]) GDR_MFMA_BF(A1, B0, acc[1][0])                     \
    GDR_MFMA_BF(A1, B1, acc[1][1])                                                                                   \
  } else {                                                                                                           \
    GDR_MFMA4(A0, A1, B0, B1, x) GDR_MFMA4(A0, A1, B0, B1, y) GDR_MFMA4(A0, A1, B0, B1, z) GDR_MFMA4(A0, A1, B0, B1, w) \
  }

  if (!ktail) {
    // Steady-state loop (K % 32 == 0), branch-free so that it is ONE scheduling region: the staging registers run
    // one tile ahead (tile kt+1 is written to LDS while tile kt+2 is being fetched), and the 8 ds_write, 8 late
    // ds_read and 8 global_load of a K-step are each slotted behind one MFMA (64 issue cycles) instead of sitting
    // in front of / behind the MFMA stream with the matrix pipe idle.  Lab numbers (tools/gemm_lab.hip, MI355X):
    // 118 -> 129..137 TFLOP/s on the encoder shapes, bit-identical output (same k order).
    // The last two iterations re-fetch / re-store the last tile: in-bounds, never read.
    GDR_GLOAD(nk > 1 ? 1 : 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const float* a = As + buf * BM * LDS_STRIDE + a_rd;
      const float* b = Bs + buf * BN * LDS_STRIDE + b_rd;
      float4 c0a0, c0a1, c0b0, c0b1, c1a0, c1a1, c1b0, c1b1, c2a0, c2a1, c2b0, c2b1, c3a0, c3a1, c3b0, c3b1;
      GDR_READ(c0a0, c0a1, c0b0, c0b1, a, b, 0)
      GDR_READ(c1a0, c1a1, c1b0, c1b1, a, b, 1)
      GDR_LSTORE(buf ^ 1);
      const int nxt = kt + 2 < nk ? kt + 2 : nk - 1;
      GDR_MFMA16(c0a0, c0a1, c0b0, c0b1)
      GDR_READ(c2a0, c2a1, c2b0, c2b1, a, b, 2)
      GDR_READ(c3a0, c3a1, c3b0, c3b1, a, b, 3)
      GDR_GLOAD(nxt);
      GDR_MFMA16(c1a0, c1a1, c1b0, c1b1)
      GDR_MFMA16(c2a0, c2a1, c2b0, c2b1)
      GDR_MFMA16(c3a0, c3a1, c3b0, c3b1)
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // ds_read: chunks 0,1
      if constexpr (BF16) {  // 16 MFMAs of 32 cycles per K-step: two memory instructions behind each of the first 12
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // ds_write
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ds_read: chunks 2,3
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // global_load
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 40, 0);
      }
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < nk;
      if (more) GDR_GLOAD(kt + 1);
      const float* a = As + buf * BM * LDS_STRIDE + a_rd;
      const float* b = Bs + buf * BN * LDS_STRIDE + b_rd;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float4 a0, a1, b0, b1;
        GDR_READ(a0, a1, b0, b1, a, b, jj)
        GDR_MFMA16(a0, a1, b0, b1)
      }
      if (more) GDR_LSTORE(buf ^ 1);
      __syncthreads();
    }
  }
#undef GDR_READ
#undef GDR_MFMA4
#undef GDR_MFMA16
#undef GDR_GLOAD
#undef GDR_LSTORE

  // ---- epilogue.  Accumulator map (32x32 MFMA): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  if (EPI == EPI_SIM_SAMPLE || EPI == EPI_SIM_FILTER) {
    // rows = docs, cols = queries: this lane owns query n for each of its two column tiles
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + l31;
      if (n >= g.N) continue;
      float* cv = g.sim.cand_val + (int64_t)n * g.sim.cap;
      int32_t* ci = g.sim.cand_idx + (int64_t)n * g.sim.cap;
      if (EPI == EPI_SIM_SAMPLE) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int roff = wm * 64 + mi * 32 + 8 * q4 + 4 * h;  // 4 consecutive docs
            const int64_t m = m0 + roff;
            float4 v;
            int4 id;
            v.x = m + 0 < g.M ? acc[mi][ni][4 * q4 + 0] : -INFINITY;
            v.y = m + 1 < g.M ? acc[mi][ni][4 * q4 + 1] : -INFINITY;
            v.z = m + 2 < g.M ? acc[mi][ni][4 * q4 + 2] : -INFINITY;
            v.w = m + 3 < g.M ? acc[mi][ni][4 * q4 + 3] : -INFINITY;
            id.x = m + 0 < g.M ? (int)m : -1, id.y = m + 1 < g.M ? (int)m + 1 : -1;
            id.z = m + 2 < g.M ? (int)m + 2 : -1, id.w = m + 3 < g.M ? (int)m + 3 : -1;  // -1 = padding slot
            *reinterpret_cast<float4*>(cv + slot_base + roff) = v;
            *reinterpret_cast<int4*>(ci + slot_base + roff) = id;
          }
        }
      } else {
        // one counter reservation per query and wave tile: survivors are flagged first, the two lane halves that
        // share a query pool their counts, and a single returning atomic replaces up to 32 serial round trips
        const float thr = g.sim.thr[n];
        unsigned keep[2] = {0u, 0u};
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            keep[mi] |= (acc[mi][ni][r] >= thr && m < g.M) ? (1u << r) : 0u;
          }
        const int mine = __popc(keep[0]) + __popc(keep[1]);
        const int other = __shfl_xor(mine, 32);
        int base = 0;
        if (h == 0 && mine + other > 0) base = atomicAdd(g.sim.cand_cnt + (int64_t)n * CNT_STRIDE, mine + other);
        base = __shfl(base, l31);
        int pos = base + (h ? other : 0);
        if (mine) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (keep[mi] >> r & 1u) {
                if (pos < g.sim.cap) {
                  cv[pos] = acc[mi][ni][r];
                  ci[pos] = (int)(m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
                } else if (g.sim.status) {
                  *g.sim.status = 1;
                }
                ++pos;
              }
        }
      }
    }
    return;
  }

  if (EPI == EPI_LINEAR && g.ksplit > 1) {  // raw partial tile, unconditional (the scratch holds whole tiles)
    float* slab = g.C + ((int64_t)local_tile * g.ksplit + split) * (BM * BN);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * BN + wn * 64 + ni * 32 + l31] = acc[mi][ni][r];
    return;
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
    if (n >= g.N) continue;
    const float bia = g.has_bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= Mrows) continue;
        float v = acc[mi][ni][r] + bia;
        if (g.has_residual) v += g.residual[m * g.ldr + n];
        if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
        if (g.act == ACT_GELU) v = gelu_erf(v);
        g.C[m * g.ldc + n] = v;
      }
    }
  }
}

// ---- persistent form of the linear kernel ---------------------------------------------------------------------
// The one-tile-per-workgroup launch above runs its 512 resident workgroups in lockstep: every tile of a round reaches
// its epilogue at the same moment, so each round ends with the whole chip storing (and re-loading residuals) while the
// matrix pipes idle, then starts with every workgroup waiting on its first operand loads.  Measured: time(K) at
// M=20480 has an intercept worth 13-15 us per round of tiles — 13 % of a K=768 tile, the reason the K=768 linears ran
// at 103-113 TFLOP/s against 134-139 asymptotically (tools/bench_linear_ksweep.py).
// Here a workgroup owns a strided list of tiles and runs ONE flattened K-step stream across them: the operand prefetch
// (two K-steps ahead) simply continues into the next tile while the current one finishes, and the epilogue is a burst
// of stores issued between two K-steps that drains in the background under the next tile's MFMAs.
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_f32_persistent_kernel(const GemmArgs g, const int total_tiles_host) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDS_STRIDE + 2 * BN * LDS_STRIDE];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LDS_STRIDE;
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int G = (int)gridDim.x;
  const int tid = threadIdx.x;
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;
  const int st_off = lrow * LDS_STRIDE + lcol;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int a_rd = (wm * 64 + l31) * LDS_STRIDE + 4 * h;
  const int b_rd = (wn * 64 + l31) * LDS_STRIDE + 4 * h;
  const int nk = g.K / BK;
  const int64_t Mrows = live_rows_padded(g);  // ragged batches: a device-side value <= g.M, in whole row panels
  const int tiles_m = (int)((Mrows + BM - 1) / BM);
  const int total_tiles = g.m_dev ? tiles_m * g.tiles_n : total_tiles_host;
  constexpr int S = 1;  // whole tiles only (the stream-K kernel's macros of the same name also serve (tile, chunk) units)
  // a device-side row count of 0 (prefix-table decode when every beam hits the table): no tile exists to park the load
  // stream on — the prologue below would divide by gm_ = 0 and clamp rows to -1.  Uniform exit before anything is loaded.
  if (total_tiles <= 0) return;

  int64_t a_ld[4];  // element offsets from g.A / g.W — kept as integers: pointers that pass through the tile-crossing
  int64_t w_ld[4];  // select lose their address space and the loads degrade to flat_load (vmcnt AND lgkmcnt)
// tile index -> (row panel, column tile): supertiles of GM row panels x all column tiles, row-panel-fastest inside, so
// that the 64 tiles an XCD works on at a time are ~8 row panels x 8 column tiles (16 operand panels through its L2)
// instead of a few row panels x every column tile (N=3072: 27 panels).  GM travels in g.ksplit (unused by this kernel).
#define P_TILE_MN(item_, mt_, nt_)                                           \
  {                                                                          \
    const int tile_ = S > 1 ? (item_) / S : (item_);                         \
    const int per_ = g.ksplit * g.tiles_n;                                   \
    const int grp_ = (tile_) / per_, loc_ = (tile_) - grp_ * per_;           \
    const int gm_ = min(g.ksplit, tiles_m - grp_ * g.ksplit);                \
    nt_ = loc_ / gm_;                                                        \
    mt_ = grp_ * g.ksplit + (loc_ - nt_ * gm_);                              \
  }
#define P_SETPTRS(tile_)                                                     \
  {                                                                          \
    int mt_, nt_;                                                            \
    P_TILE_MN(tile_, mt_, nt_)                                               \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                          \
      int64_t ra_ = (int64_t)mt_ * BM + lrow + 32 * p;                       \
      ra_ = ra_ < Mrows ? ra_ : Mrows - 1;                                   \
      int rw_ = nt_ * BN + lrow + 32 * p;                                    \
      rw_ = rw_ < g.N ? rw_ : g.N - 1;                                       \
      a_ld[p] = ra_ * g.lda + lcol;                                          \
      w_ld[p] = (int64_t)rw_ * g.ldw + lcol;                                 \
    }                                                                        \
  }
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define P_GLOAD                                          \
  ra0 = *reinterpret_cast<const float4*>(g.A + a_ld[0]); \
  ra1 = *reinterpret_cast<const float4*>(g.A + a_ld[1]); \
  ra2 = *reinterpret_cast<const float4*>(g.A + a_ld[2]); \
  ra3 = *reinterpret_cast<const float4*>(g.A + a_ld[3]); \
  rb0 = *reinterpret_cast<const float4*>(g.W + w_ld[0]); \
  rb1 = *reinterpret_cast<const float4*>(g.W + w_ld[1]); \
  rb2 = *reinterpret_cast<const float4*>(g.W + w_ld[2]); \
  rb3 = *reinterpret_cast<const float4*>(g.W + w_ld[3]);
#define P_ADVANCE                                                            \
  if (++ld_kt == nk) { /* the load stream crosses into the workgroup's next tile (or parks on its first) */ \
    ld_kt = 0;                                                               \
    ld_tile += G;                                                            \
    const int t_ = ld_tile < total_tiles ? ld_tile : park;                   \
    P_SETPTRS(t_)                                                            \
  } else {                                                                   \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) a_ld[p] += BK, w_ld[p] += BK; \
  }
#define P_LSTORE(buf_)                                                       \
  {                                                                          \
    float* a_ = As + (buf_)*BM * LDS_STRIDE + st_off;                        \
    float* b_ = Bs + (buf_)*BN * LDS_STRIDE + st_off;                        \
    *reinterpret_cast<float4*>(a_) = ra0;                                    \
    *reinterpret_cast<float4*>(a_ + 32 * LDS_STRIDE) = ra1;                  \
    *reinterpret_cast<float4*>(a_ + 64 * LDS_STRIDE) = ra2;                  \
    *reinterpret_cast<float4*>(a_ + 96 * LDS_STRIDE) = ra3;                  \
    *reinterpret_cast<float4*>(b_) = rb0;                                    \
    *reinterpret_cast<float4*>(b_ + 32 * LDS_STRIDE) = rb1;                  \
    *reinterpret_cast<float4*>(b_ + 64 * LDS_STRIDE) = rb2;                  \
    *reinterpret_cast<float4*>(b_ + 96 * LDS_STRIDE) = rb3;                  \
  }
#define P_READ(A0, A1, B0, B1, ap, bp, jj)                                   \
  A0 = *reinterpret_cast<const float4*>((ap) + 8 * (jj));                    \
  A1 = *reinterpret_cast<const float4*>((ap) + 32 * LDS_STRIDE + 8 * (jj));  \
  B0 = *reinterpret_cast<const float4*>((bp) + 8 * (jj));                    \
  B1 = *reinterpret_cast<const float4*>((bp) + 32 * LDS_STRIDE + 8 * (jj));
#define P_MFMA4(A0, A1, B0, B1, x_)                                                   \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B0.x_, acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B1.x_, acc[0][1], 0, 0, 0); \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B0.x_, acc[1][0], 0, 0, 0); \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B1.x_, acc[1][1], 0, 0, 0);
#define P_MFMA16(A0, A1, B0, B1) \
  P_MFMA4(A0, A1, B0, B1, x) P_MFMA4(A0, A1, B0, B1, y) P_MFMA4(A0, A1, B0, B1, z) P_MFMA4(A0, A1, B0, B1, w)

  // a workgroup whose list is exhausted (or empty: with a device-side row count the grid may exceed the tile count)
  // parks its load stream on a valid tile
  const int park = (int)bid < total_tiles ? (int)bid : 0;
  int ld_tile = (int)bid, ld_kt = 0;
  P_SETPTRS(park)
  P_GLOAD
  P_ADVANCE
  P_LSTORE(0)
  __syncthreads();
  P_GLOAD
  P_ADVANCE

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int tile = (int)bid, kt = 0, buf = 0;
  while (tile < total_tiles) {
    {
      const float* a = As + buf * BM * LDS_STRIDE + a_rd;
      const float* b = Bs + buf * BN * LDS_STRIDE + b_rd;
      float4 c0a0, c0a1, c0b0, c0b1, c1a0, c1a1, c1b0, c1b1, c2a0, c2a1, c2b0, c2b1, c3a0, c3a1, c3b0, c3b1;
      P_READ(c0a0, c0a1, c0b0, c0b1, a, b, 0)
      P_READ(c1a0, c1a1, c1b0, c1b1, a, b, 1)
      P_LSTORE(buf ^ 1)
      P_MFMA16(c0a0, c0a1, c0b0, c0b1)
      P_READ(c2a0, c2a1, c2b0, c2b1, a, b, 2)
      P_READ(c3a0, c3a1, c3b0, c3b1, a, b, 3)
      P_GLOAD
      P_MFMA16(c1a0, c1a1, c1b0, c1b1)
      P_MFMA16(c2a0, c2a1, c2b0, c2b1)
      P_MFMA16(c3a0, c3a1, c3b0, c3b1)
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // ds_read: chunks 0,1
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // ds_write
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ds_read: chunks 2,3
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // global_load
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 40, 0);
    }
    __syncthreads();  // stays in the MFMA block: the compiler hoists it above the trailing 40 MFMAs, which then cover the wait
    P_ADVANCE
    buf ^= 1;
    if (++kt == nk) {
      // epilogue of this tile: stores only (plus the residual / bias loads), no LDS — the other waves are already
      // in the next tile's first K-step.  Accumulator map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
      int mt, nt;
      P_TILE_MN(tile, mt, nt)
      const int64_t m0 = (int64_t)mt * BM;
      const int n0 = nt * BN;
      if (m0 + BM <= Mrows && n0 + BN <= g.N) {
        // interior tile: uniform branches only, so that the 64 residual loads go out as one batch and the 64 stores
        // as another (per-element flag tests serialise them behind a vmcnt(0) each)
        const int64_t mrow = m0 + wm * 64 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {  // one column tile (32 accumulators) at a time: bounds the registers in flight
          const int ncol = n0 + wn * 64 + ni * 32 + l31;
          if (g.has_bias) {
            const float bia = g.bias[ncol];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] += bia;
          }
          if (g.has_residual) {
            const float* rp = g.residual + mrow * g.ldr + ncol;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] += rp[(int64_t)(mi * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr];
          }
          if (g.act == ACT_RELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = fmaxf(acc[mi][ni][r], 0.f);
          } else if (g.act == ACT_GELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = gelu_erf(acc[mi][ni][r]);
          }
          float* cp = g.C + mrow * g.ldc + ncol;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              cp[(int64_t)(mi * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc] = acc[mi][ni][r];
              acc[mi][ni][r] = 0.f;
            }
        }
      } else {
        // edge tile (the last, partial row panel of a packed batch; a column tail): the same batched structure — rows and
        // columns past the matrix are CLAMPED for the loads (legal addresses, one batch) and masked for the stores.  With a
        // per-element `continue` every residual load sat behind its own vmcnt(0) and the 6..24 edge tiles of a launch finished
        // after everyone else: 12 350 live rows (62 in the last panel) 17.2 -> 16.05 ms of linears per encoder pass
        // (tools/exp_edge.py; 12 416 rows = 97 full panels: 15.8 ms).
        const int64_t mrow = m0 + wm * 64 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int n = n0 + wn * 64 + ni * 32 + l31;
          const bool n_ok = n < g.N;
          const int nc = n_ok ? n : g.N - 1;
          if (g.has_bias) {
            const float bia = g.bias[nc];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] += bia;
          }
          if (g.has_residual) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int64_t m = mrow + mi * 32 + (r & 3) + 8 * (r >> 2);
                acc[mi][ni][r] += g.residual[(m < Mrows ? m : Mrows - 1) * g.ldr + nc];
              }
          }
          if (g.act == ACT_RELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = fmaxf(acc[mi][ni][r], 0.f);
          } else if (g.act == ACT_GELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = gelu_erf(acc[mi][ni][r]);
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int64_t m = mrow + mi * 32 + (r & 3) + 8 * (r >> 2);
              if (n_ok && m < Mrows) g.C[m * g.ldc + n] = acc[mi][ni][r];
              acc[mi][ni][r] = 0.f;
            }
        }
      }
      kt = 0;
      tile += G;
    }
  }
#undef P_TILE_MN
#undef P_SETPTRS
#undef P_GLOAD
#undef P_ADVANCE
#undef P_LSTORE
#undef P_READ
#undef P_MFMA4
#undef P_MFMA16
}

// ---- stream-K tail for the persistent kernel ----------------------------------------------------------------------
// The persistent kernel above deals whole tiles: with T tiles on G = 512 resident workgroups it takes ceil(T/G) rounds
// where T/G would do — 582 tiles (ragged C2 batch, N = 768) cost two rounds for 1.14 rounds of work.  This kernel runs all
// rounds but the last full one exactly like the persistent kernel (workgroup b takes tiles b, b+G, ...: lockstep, operand
// slices shared in time through the L2) and deals the remaining G + (T mod G) tiles by the K-STEP: their K-steps are cut
// into G equal contiguous ranges (1..2 tiles' worth), so a range covers the end of one tile (begun by the previous
// workgroup), possibly a whole tile, and the beginning of another (finished by the next workgroup).  A tile that spans two
// workgroups is NOT reduced from two partial sums (that would change the summation order): the first workgroup stores its
// raw accumulators and raises a flag, the second one loads them and simply continues the k loop — the result is
// bit-identical to the one-workgroup tile.  In the tail a workgroup runs its BEGINNING fragment first (it depends on
// nobody), then its whole tile, and the CONTINUED fragment last, by which time the predecessor published long ago (a range
// is at least one tile long, so the predecessor's fragment is shorter than what precedes the take-over); the wait is a
// bounded spin.  Same flattened K-step stream, same K-step body and schedule as above.
// Memory model (MI355X_MICROARCH.md, visibility): the XCDs' L2s are not coherent with each other inside a kernel.  The
// producer stores its accumulators write-through (sc1 buffer stores: no L2 write-back fence needed), every wave drains its
// stores (s_waitcnt vmcnt(0)), the workgroup meets, one lane stores the flag (relaxed, agent scope).  The consumer's lane 0
// polls the flag relaxed, issues ONE agent-scope acquire (drops this CU's stale L1 lines of the buffer, left by an earlier
// launch), drains it, the workgroup meets, then plain loads.  (Round 2's first form — plain stores + __threadfence() by
// every thread + a release flag store — cost 30 us more per launch on the o projection: 176 -> 139 us.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct StreamKArgs {
  float* part;
  int32_t* flag;
  int32_t epoch;
  int32_t* err;  // host-mapped word (streamk_err_word): a take-over that timed out raises it; may be null
  // split-K UNITS (S > 1): the work items are (tile, K chunk s) pairs of g.kchunk contraction steps each, in the order
  // unit = tile * S + s, and a finished unit stores its raw 128x128 partial at slabs[unit] — the slab layout and the k order
  // inside a chunk are those of gemm_nt_f32_kernel's split-K form, so splitk_reduce_kernel gives the bit-identical result.
  // Dealing the units by K-step ranges balances launches whose (tile, chunk) count sits between one and two per CU.
  int S;
  float* slabs;
};
// A take-over spin that runs out (the predecessor workgroup never became resident for seconds — a co-tenant kernel that
// starves it; the kernel itself assumes its <= 512 workgroups are co-resident, 2 per CU) used to trap, which aborts the
// whole process.  Now the waiting lane raises this process-wide word (pinned host memory, mapped into every device) and
// the workgroup goes on with whatever the hand-off buffer holds: the launch completes, its output is WRONG.  The word is
// STICKY: it stays raised until the caller acknowledges it with gdr_device_fault_clear(), every stream-K launch enqueued
// while it is raised fails with GDR_EHIP, and gdr_device_fault_pending() lets a caller check it where it synchronises
// anyway (the Python side does after every result read-back: ops.finish_generate_output, sim_topk's status read,
// GDRRetriever's step output) — so an invalid result is never handed out as GDR_OK unnoticed, whichever launch was last.
static int32_t* streamk_err_word() {
  static int32_t* word = [] {
    int32_t* h = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&h), 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      return static_cast<int32_t*>(nullptr);
    }
    *h = 0;
    return h;
  }();
  return word;
}
static const char* const kStreamKFault =
    "linear(stream-K): a hand-off between workgroups timed out in an earlier launch (its 512 workgroups were not co-resident: "
    "another kernel held the CUs for seconds); results produced since the last gdr_device_fault_clear() are invalid";
static int streamk_poll_error() {
  int32_t* w = streamk_err_word();
  if (w && __atomic_load_n(w, __ATOMIC_RELAXED) != 0) {
    set_error("%s", kStreamKFault);
    return GDR_EHIP;
  }
  return GDR_OK;
}
}  // namespace gdr
extern "C" int gdr_device_fault_pending(void) {
  int32_t* w = gdr::streamk_err_word();
  if (w && __atomic_load_n(w, __ATOMIC_RELAXED) != 0) {
    gdr::set_error("%s", gdr::kStreamKFault);
    return 1;
  }
  return 0;
}
extern "C" void gdr_device_fault_clear(void) {
  int32_t* w = gdr::streamk_err_word();
  if (w) __atomic_store_n(w, 0, __ATOMIC_RELAXED);
}
extern "C" void gdr_device_fault_inject_for_tests(void) {   // raises the word exactly as a timed-out take-over would
  int32_t* w = gdr::streamk_err_word();
  if (w) __atomic_store_n(w, 1, __ATOMIC_RELAXED);
}
namespace gdr {
// When it runs: the whole-tile form takes ceil(T/G) rounds (measured: a last round with <= 256 tiles is no shorter), this
// form T/G rounds plus the hand-off (one 64 KB write-through publish and one 64 KB take-over per workgroup, ~5 K-steps of
// time at 2 workgroups per CU).  It is used when the whole-tile form's idle tail, (G - T mod G)/G of a round, exceeds
// GDR_GEMM_STREAMK K-steps (default 6; 0 = never, 1 = always) — tools/run_sk.sh sweeps 8 batch sizes x 4 shapes
// (profiles/r02_streamk_sweep.txt).  The tile count comes from the caller's row hint when the real count lives on the
// device — a tuning input only: both kernels are correct for any row count.
static bool streamk_fits(int64_t M, int64_t lda, int N, int64_t ldw) {  // 32-bit element offsets inside the kernel
  return M * lda < 0x7fffffffLL && (int64_t)N * ldw < 0x7fffffffLL;
}
// Between one and two tiles per CU (256 < T <= 512) one workgroup per tile leaves every CU waiting for the few that got
// two (T = 270: 4.3 ms for a 48-query encoder pass against 4.6 ms for 64 queries).  The same kernel on 256 workgroups, one
// per CU, deals T/256 tiles' worth of K-steps to each instead: 48 / 56 / 64 queries 4.31 / 4.57 / 4.60 -> 3.54 / 4.23 / 4.28 ms.
static bool streamk_mid_wanted(int64_t tiles) {
  static const int mid_max = [] {
    const char* e = getenv("GDR_GEMM_STREAMK_MID");  // largest T served this way; 0 = off
    return e ? atoi(e) : 512;
  }();
  return tiles > 256 && tiles <= mid_max;
}
static bool streamk_wanted(int64_t tiles, int nk) {
  static const int thr = [] {
    const char* e = getenv("GDR_GEMM_STREAMK");
    return e ? atoi(e) : 6;
  }();
  if (thr <= 0 || tiles < 512) return false;
  if (thr == 1) return true;
  const int64_t rem = tiles % 512;
  return rem != 0 && (512 - rem) * nk > (int64_t)thr * 512;
}
template <bool UNITS>  // UNITS: the work items are split-K (tile, chunk) units (StreamKArgs.S > 1), else whole tiles
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_f32_streamk_kernel(const GemmArgs g, const int total_tiles_host,
                                                                              const StreamKArgs sk) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDS_STRIDE + 2 * BN * LDS_STRIDE];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LDS_STRIDE;
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int G = (int)gridDim.x;
  const int tid = threadIdx.x;
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;
  const int st_off = lrow * LDS_STRIDE + lcol;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int a_rd = (wm * 64 + l31) * LDS_STRIDE + 4 * h;
  const int b_rd = (wn * 64 + l31) * LDS_STRIDE + 4 * h;
  const int S = UNITS ? sk.S : 1;  // a compile-time 1 in the whole-tile kernel: the unit bookkeeping costs it no register
  const int nk = (S > 1 ? g.kchunk : g.K) / BK;
  const int64_t Mrows = live_rows_padded(g);  // ragged batches: a device-side value <= g.M, in whole row panels
  const int tiles_m = (int)((Mrows + BM - 1) / BM);
  const int total_tiles = g.m_dev ? tiles_m * g.tiles_n * S : total_tiles_host;  // work items: tiles, or (tile, chunk) units

  int a_ld[4];  // element offsets from g.A / g.W, 32-bit (the launcher sends operands of >= 2^31 elements to the whole-tile
  int w_ld[4];  // kernel): with 64-bit offsets the segment bookkeeping pushed three spills into the K-step
// tile index -> (row panel, column tile): supertiles of GM row panels x all column tiles, row-panel-fastest inside, so
// that the 64 tiles an XCD works on at a time are ~8 row panels x 8 column tiles (16 operand panels through its L2)
// instead of a few row panels x every column tile (N=3072: 27 panels).  GM travels in g.ksplit (unused by this kernel).
#define P_TILE_MN(item_, mt_, nt_)                                           \
  {                                                                          \
    const int tile_ = S > 1 ? (item_) / S : (item_);                         \
    const int per_ = g.ksplit * g.tiles_n;                                   \
    const int grp_ = (tile_) / per_, loc_ = (tile_) - grp_ * per_;           \
    const int gm_ = min(g.ksplit, tiles_m - grp_ * g.ksplit);                \
    nt_ = loc_ / gm_;                                                        \
    mt_ = grp_ * g.ksplit + (loc_ - nt_ * gm_);                              \
  }
#define P_SETPTRS(tile_)                                                     \
  {                                                                          \
    int mt_, nt_;                                                            \
    P_TILE_MN(tile_, mt_, nt_)                                               \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                          \
      int64_t ra_ = (int64_t)mt_ * BM + lrow + 32 * p;                       \
      ra_ = ra_ < Mrows ? ra_ : Mrows - 1;                                   \
      int rw_ = nt_ * BN + lrow + 32 * p;                                    \
      rw_ = rw_ < g.N ? rw_ : g.N - 1;                                       \
      const int ks_ = S > 1 ? ((tile_) % S) * g.kchunk : 0;                  \
      a_ld[p] = (int)(ra_ * g.lda + lcol + ks_);                             \
      w_ld[p] = (int)((int64_t)rw_ * g.ldw + lcol + ks_);                    \
    }                                                                        \
  }
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define P_GLOAD                                          \
  ra0 = *reinterpret_cast<const float4*>(g.A + a_ld[0]); \
  ra1 = *reinterpret_cast<const float4*>(g.A + a_ld[1]); \
  ra2 = *reinterpret_cast<const float4*>(g.A + a_ld[2]); \
  ra3 = *reinterpret_cast<const float4*>(g.A + a_ld[3]); \
  rb0 = *reinterpret_cast<const float4*>(g.W + w_ld[0]); \
  rb1 = *reinterpret_cast<const float4*>(g.W + w_ld[1]); \
  rb2 = *reinterpret_cast<const float4*>(g.W + w_ld[2]); \
  rb3 = *reinterpret_cast<const float4*>(g.W + w_ld[3]);
#define P_ADVANCE                                                            \
  if (++ld_kt == ld_kend) { /* the load stream crosses into the next segment (or parks on the last one's start) */ \
    ld_seg = ld_seg + 1 < nseg ? ld_seg + 1 : nseg - 1;                      \
    SK_SEG(ld_seg, ld_tile, ld_kt, ld_kend)                                  \
    P_SETPTRS(ld_tile)                                                       \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) a_ld[p] += ld_kt * BK, w_ld[p] += ld_kt * BK; \
  } else {                                                                   \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) a_ld[p] += BK, w_ld[p] += BK; \
  }
#define P_LSTORE(buf_)                                                       \
  {                                                                          \
    float* a_ = As + (buf_)*BM * LDS_STRIDE + st_off;                        \
    float* b_ = Bs + (buf_)*BN * LDS_STRIDE + st_off;                        \
    *reinterpret_cast<float4*>(a_) = ra0;                                    \
    *reinterpret_cast<float4*>(a_ + 32 * LDS_STRIDE) = ra1;                  \
    *reinterpret_cast<float4*>(a_ + 64 * LDS_STRIDE) = ra2;                  \
    *reinterpret_cast<float4*>(a_ + 96 * LDS_STRIDE) = ra3;                  \
    *reinterpret_cast<float4*>(b_) = rb0;                                    \
    *reinterpret_cast<float4*>(b_ + 32 * LDS_STRIDE) = rb1;                  \
    *reinterpret_cast<float4*>(b_ + 64 * LDS_STRIDE) = rb2;                  \
    *reinterpret_cast<float4*>(b_ + 96 * LDS_STRIDE) = rb3;                  \
  }
#define P_READ(A0, A1, B0, B1, ap, bp, jj)                                   \
  A0 = *reinterpret_cast<const float4*>((ap) + 8 * (jj));                    \
  A1 = *reinterpret_cast<const float4*>((ap) + 32 * LDS_STRIDE + 8 * (jj));  \
  B0 = *reinterpret_cast<const float4*>((bp) + 8 * (jj));                    \
  B1 = *reinterpret_cast<const float4*>((bp) + 32 * LDS_STRIDE + 8 * (jj));
#define P_MFMA4(A0, A1, B0, B1, x_)                                                   \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B0.x_, acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x_, B1.x_, acc[0][1], 0, 0, 0); \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B0.x_, acc[1][0], 0, 0, 0); \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x_, B1.x_, acc[1][1], 0, 0, 0);
#define P_MFMA16(A0, A1, B0, B1) \
  P_MFMA4(A0, A1, B0, B1, x) P_MFMA4(A0, A1, B0, B1, y) P_MFMA4(A0, A1, B0, B1, z) P_MFMA4(A0, A1, B0, B1, w)

  // ---- this workgroup's range of K-steps and its segments, in processing order:
  //      [begun fragment: tile t_last, k 0..k_last) -> stored]  [whole tiles t_full0 .. t_full0+n_full)]
  //      [continued fragment: tile t_first, k k_first..nk) <- loaded]
  int t_first, k_first, t_last, k_last, dp_rounds = 0;
  if (total_tiles >= G) {
    {  // all rounds but the last full one as whole tiles in lockstep, the rest (G..2G-1 tiles) by K-step ranges
      const int rounds = total_tiles / G;
      dp_rounds = total_tiles - rounds * G ? rounds - 1 : rounds;
    }
    const int sk_base = dp_rounds * G;
    const int64_t iters = (int64_t)(total_tiles - sk_base) * nk;
    const int64_t lo = (int64_t)bid * iters / G, hi = (int64_t)(bid + 1) * iters / G;
    t_first = (int)(lo / nk), k_first = (int)(lo - (int64_t)t_first * nk);
    t_last = (int)(hi / nk), k_last = (int)(hi - (int64_t)t_last * nk);
    t_first += sk_base, t_last += sk_base;
  } else {  // fewer tiles than workgroups (device-side row count): whole tiles, one each
    t_first = min((int)bid, total_tiles), k_first = 0;
    t_last = min((int)bid + 1, total_tiles), k_last = 0;
  }
  const bool has_head = k_first != 0, has_tail = k_last != 0;
  const int t_full0 = has_head ? t_first + 1 : t_first;
  const int n_full = t_last - t_full0;
  const int nseg = dp_rounds + (has_tail ? 1 : 0) + n_full + (has_head ? 1 : 0);
  // segment s -> tile, first K-step, end K-step
#define SK_SEG(s_, tile_, k0_, k1_)                                   \
  {                                                                   \
    int q_ = (s_) - dp_rounds;                                        \
    if (q_ < 0) {                                                     \
      tile_ = (int)bid + (s_) * G, k0_ = 0, k1_ = nk;                 \
    } else if (has_tail && q_ == 0) {                                 \
      tile_ = t_last, k0_ = 0, k1_ = k_last;                          \
    } else {                                                          \
      q_ -= has_tail ? 1 : 0;                                         \
      if (q_ < n_full) {                                              \
        tile_ = t_full0 + q_, k0_ = 0, k1_ = nk;                      \
      } else {                                                        \
        tile_ = t_first, k0_ = k_first, k1_ = nk;                     \
      }                                                               \
    }                                                                 \
  }
  if (nseg == 0) return;
  // take over the predecessor's accumulators of the tile both share (the CONTINUED fragment)
#define SK_TAKEOVER                                                                                               \
  {                                                                                                               \
    if (tid == 0) {                                                                                               \
      int spins_ = 0;                                                                                             \
      while (__hip_atomic_load(sk.flag + (bid - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) {    \
        __builtin_amdgcn_s_sleep(8);                                                                              \
        if (++spins_ > (1 << 24)) { /* seconds: the predecessor never ran — flag the launch as failed, do not hang */ \
          if (sk.err) __hip_atomic_store(sk.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                 \
          break;                                                                                                  \
        }                                                                                                         \
      }                                                                                                           \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* one lane: drops this CU's stale L1 lines of the buffer */ \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
    }                                                                                                             \
    __syncthreads();                                                                                              \
    const float4* src_ = reinterpret_cast<const float4*>(sk.part + (size_t)(bid - 1) * (BM * BN)) + tid;          \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)             \
        _Pragma("unroll") for (int r4 = 0; r4 < 4; ++r4) {                                                        \
      const float4 v_ = src_[((mi * 2 + ni) * 4 + r4) * GEMM_THREADS];                                            \
      acc[mi][ni][4 * r4] = v_.x, acc[mi][ni][4 * r4 + 1] = v_.y, acc[mi][ni][4 * r4 + 2] = v_.z, acc[mi][ni][4 * r4 + 3] = v_.w; \
    }                                                                                                             \
  }
  int ld_seg = 0, ld_kt, ld_kend, ld_tile;
  SK_SEG(0, ld_tile, ld_kt, ld_kend)
  P_SETPTRS(ld_tile)
  _Pragma("unroll") for (int p = 0; p < 4; ++p) a_ld[p] += ld_kt * BK, w_ld[p] += ld_kt * BK;
  P_GLOAD
  P_ADVANCE
  P_LSTORE(0)
  __syncthreads();
  P_GLOAD
  P_ADVANCE

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int seg = 0, tile, kt, kend, buf = 0;
  SK_SEG(0, tile, kt, kend)  // never the continued fragment: with total_tiles >= G a range is at least nk steps long, so
                             // something (a begun fragment or a whole tile) always precedes it
  while (seg < nseg) {
    {
      const float* a = As + buf * BM * LDS_STRIDE + a_rd;
      const float* b = Bs + buf * BN * LDS_STRIDE + b_rd;
      float4 c0a0, c0a1, c0b0, c0b1, c1a0, c1a1, c1b0, c1b1, c2a0, c2a1, c2b0, c2b1, c3a0, c3a1, c3b0, c3b1;
      P_READ(c0a0, c0a1, c0b0, c0b1, a, b, 0)
      P_READ(c1a0, c1a1, c1b0, c1b1, a, b, 1)
      P_LSTORE(buf ^ 1)
      P_MFMA16(c0a0, c0a1, c0b0, c0b1)
      P_READ(c2a0, c2a1, c2b0, c2b1, a, b, 2)
      P_READ(c3a0, c3a1, c3b0, c3b1, a, b, 3)
      P_GLOAD
      P_MFMA16(c1a0, c1a1, c1b0, c1b1)
      P_MFMA16(c2a0, c2a1, c2b0, c2b1)
      P_MFMA16(c3a0, c3a1, c3b0, c3b1)
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // ds_read: chunks 0,1
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // ds_write
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ds_read: chunks 2,3
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // global_load
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 40, 0);
    }
    __syncthreads();  // stays in the MFMA block: the compiler hoists it above the trailing 40 MFMAs, which then cover the wait
    P_ADVANCE
    buf ^= 1;
    if (++kt == kend) {
     if (kend != nk) {
      // a BEGUN fragment (always this workgroup's first segment): hand the raw accumulators to the successor
      // write-through (sc1) stores need no L2 write-back fence: every wave drains its own stores, the workgroup meets,
      // one lane raises the flag (MI355X_MICROARCH.md, visibility: "publish-large", 3.0 us against 8.2 for plain + release)
      const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(sk.part + (size_t)bid * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            u32x4 v_;
            v_[0] = __float_as_uint(acc[mi][ni][4 * r4]), v_[1] = __float_as_uint(acc[mi][ni][4 * r4 + 1]);
            v_[2] = __float_as_uint(acc[mi][ni][4 * r4 + 2]), v_[3] = __float_as_uint(acc[mi][ni][4 * r4 + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(v_, dst, ((((mi * 2 + ni) * 4 + r4) * GEMM_THREADS) + tid) * 16, 0, 16);
            acc[mi][ni][4 * r4] = acc[mi][ni][4 * r4 + 1] = acc[mi][ni][4 * r4 + 2] = acc[mi][ni][4 * r4 + 3] = 0.f;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(sk.flag + bid, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
     } else {
      // epilogue of this tile: stores only (plus the residual / bias loads), no LDS — the other waves are already
      // in the next tile's first K-step.  Accumulator map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
      int mt, nt;
      P_TILE_MN(tile, mt, nt)
      const int64_t m0 = (int64_t)mt * BM;
      const int n0 = nt * BN;
      if (UNITS) {  // a finished (tile, chunk) unit: the raw partial, in gemm_nt_f32_kernel's slab layout
        float* slab = sk.slabs + (int64_t)tile * (BM * BN) + (wm * 64 + 4 * h) * BN + wn * 64 + l31;  // unit index = tile * S + s already
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              slab[(mi * 32 + (r & 3) + 8 * (r >> 2)) * BN + ni * 32] = acc[mi][ni][r];
              acc[mi][ni][r] = 0.f;
            }
      } else if (m0 + BM <= Mrows && n0 + BN <= g.N) {
        // interior tile: uniform branches only, so that the 64 residual loads go out as one batch and the 64 stores
        // as another (per-element flag tests serialise them behind a vmcnt(0) each)
        const int64_t mrow = m0 + wm * 64 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {  // one column tile (32 accumulators) at a time: bounds the registers in flight
          const int ncol = n0 + wn * 64 + ni * 32 + l31;
          if (g.has_bias) {
            const float bia = g.bias[ncol];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] += bia;
          }
          if (g.has_residual) {
            const float* rp = g.residual + mrow * g.ldr + ncol;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] += rp[(int64_t)(mi * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr];
          }
          if (g.act == ACT_RELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = fmaxf(acc[mi][ni][r], 0.f);
          } else if (g.act == ACT_GELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = gelu_erf(acc[mi][ni][r]);
          }
          float* cp = g.C + mrow * g.ldc + ncol;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              cp[(int64_t)(mi * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc] = acc[mi][ni][r];
              acc[mi][ni][r] = 0.f;
            }
        }
      } else {
        // edge tile (the last, partial row panel of a packed batch; a column tail): the same batched structure — rows and
        // columns past the matrix are CLAMPED for the loads (legal addresses, one batch) and masked for the stores.  With a
        // per-element `continue` every residual load sat behind its own vmcnt(0) and the 6..24 edge tiles of a launch finished
        // after everyone else: 12 350 live rows (62 in the last panel) 17.2 -> 16.05 ms of linears per encoder pass
        // (tools/exp_edge.py; 12 416 rows = 97 full panels: 15.8 ms).
        const int64_t mrow = m0 + wm * 64 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int n = n0 + wn * 64 + ni * 32 + l31;
          const bool n_ok = n < g.N;
          const int nc = n_ok ? n : g.N - 1;
          if (g.has_bias) {
            const float bia = g.bias[nc];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] += bia;
          }
          if (g.has_residual) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int64_t m = mrow + mi * 32 + (r & 3) + 8 * (r >> 2);
                acc[mi][ni][r] += g.residual[(m < Mrows ? m : Mrows - 1) * g.ldr + nc];
              }
          }
          if (g.act == ACT_RELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = fmaxf(acc[mi][ni][r], 0.f);
          } else if (g.act == ACT_GELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = gelu_erf(acc[mi][ni][r]);
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int64_t m = mrow + mi * 32 + (r & 3) + 8 * (r >> 2);
              if (n_ok && m < Mrows) g.C[m * g.ldc + n] = acc[mi][ni][r];
              acc[mi][ni][r] = 0.f;
            }
        }
      }
     }
      if (++seg < nseg) {
        SK_SEG(seg, tile, kt, kend)
        if (kt != 0) SK_TAKEOVER  // the CONTINUED fragment (always last)
      }
    }
  }
#undef SK_SEG
#undef SK_TAKEOVER
#undef P_TILE_MN
#undef P_SETPTRS
#undef P_GLOAD
#undef P_ADVANCE
#undef P_LSTORE
#undef P_READ
#undef P_MFMA4
#undef P_MFMA16
}

template <int EPI, bool BF16 = false>
static int launch(const GemmArgs& g, int64_t tiles_m, hipStream_t stream) {
  const int64_t blocks = tiles_m * g.tiles_n;
  if (blocks <= 0) return GDR_OK;
  if (blocks > 0x7fffffffLL) {
    set_error("gemm: grid too large (%lld blocks)", (long long)blocks);
    return GDR_EINVAL;
  }
  hipLaunchKernelGGL((gemm_nt_f32_kernel<EPI, BF16>), dim3((unsigned)blocks), dim3(GEMM_THREADS), 0, stream, g);
  GDR_CHECK_LAUNCH("gemm_nt_f32_kernel");
  return GDR_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// C tile = epilogue(sum_s partial[tile][s]) in fixed order s = 0..S-1 (deterministic).  16 blocks per 128x128 tile.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int S, int tile_base,
                                                            int tiles_n, int64_t M, int N, float* __restrict__ C,
                                                            int64_t ldc, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, int64_t ldr, int act,
                                                            const int64_t* __restrict__ m_dev) {
  if (m_dev) {  // device-side row count: the slabs of tiles past it (in whole row panels, as the GEMM rounds it) were never written
    const int64_t up = (*m_dev + (BM - 1)) / BM * BM;
    M = up < M ? up : M;
  }
  const int local_tile = blockIdx.x >> 4;
  const int e = ((blockIdx.x & 15) << 8) + threadIdx.x;  // float4 index inside the tile: 128 rows x 32 float4
  const int r = e >> 5, c = (e & 31) << 2;
  const int tile = local_tile + tile_base;
  const int64_t m = (int64_t)(tile / tiles_n) * BM + r;
  const int n = (tile % tiles_n) * BN + c;
  if (m >= M || n >= N) return;
  const float* p = partial + (int64_t)local_tile * S * (BM * BN) + r * BN + c;
  float4 v = *reinterpret_cast<const float4*>(p);
  int s = 1;
  for (; s + 4 <= S; s += 4) {  // fixed summation order s = 0..S-1, four slab loads in flight per round trip
    const float4 t0 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 0) * (BM * BN));
    const float4 t1 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 1) * (BM * BN));
    const float4 t2 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 2) * (BM * BN));
    const float4 t3 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 3) * (BM * BN));
    v.x += t0.x, v.y += t0.y, v.z += t0.z, v.w += t0.w;
    v.x += t1.x, v.y += t1.y, v.z += t1.z, v.w += t1.w;
    v.x += t2.x, v.y += t2.y, v.z += t2.z, v.w += t2.w;
    v.x += t3.x, v.y += t3.y, v.z += t3.z, v.w += t3.w;
  }
  for (; s < S; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)s * (BM * BN));
    v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
  }
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (n + q >= N) break;
    float x = o[q];
    if (bias) x += bias[n + q];
    if (residual) x += residual[m * ldr + n + q];
    if (act == ACT_RELU) x = fmaxf(x, 0.f);
    if (act == ACT_GELU) x = gelu_erf(x);
    C[m * ldc + n + q] = x;
  }
}

int launch_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M,
                      int N, int K, int epilogue, const float* bias, const float* residual, int64_t ldr,
                      hipStream_t stream) {
  return launch_linear_f32_ws(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, residual, ldr, nullptr, 0, stream, nullptr);
}

// With a scratch buffer, linears whose tile grid cannot fill the chip (decode: M = batch*beams rows) are split
// along K into partial slabs + a fixed-order reduction that applies the epilogue.
int launch_linear_f32_ws_dev(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M_max,
                             const int64_t* m_dev, int N, int K, int epilogue, const float* bias, const float* residual,
                             int64_t ldr, float* splitk_ws, size_t splitk_ws_bytes, hipStream_t stream, StreamK* sk) {
  if (M_max == 0) return GDR_OK;
  GDR_CHECK_ARG(m_dev, "linear(dev rows): null row count");
  const int64_t tiles = ((M_max + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles < 192 && M_max <= 1536 && K / BK >= 4 && K % BK == 0) {  // the small-tile form, as launch_linear_f32_ws picks it
    GDR_CHECK_ARG(A && W && C, "linear(dev rows): null pointer");
    GDR_CHECK_ARG(epilogue >= GDR_EPI_NONE && epilogue <= GDR_EPI_BIAS_GELU, "linear(dev rows): unknown epilogue %d", epilogue);
    GDR_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K && ldc >= N && aligned16(A) && aligned16(W),
                  "linear(dev rows): bad leading dimension / alignment");
    const bool nb = epilogue == GDR_EPI_BIAS || epilogue == GDR_EPI_BIAS_RELU || epilogue == GDR_EPI_BIAS_RESIDUAL ||
                    epilogue == GDR_EPI_BIAS_GELU;
    const bool nr = epilogue == GDR_EPI_RESIDUAL || epilogue == GDR_EPI_BIAS_RESIDUAL;
    GDR_CHECK_ARG((!nb || bias) && (!nr || (residual && ldr >= N)), "linear(dev rows): epilogue %d lacks an operand", epilogue);
    const int act = (epilogue == GDR_EPI_RELU || epilogue == GDR_EPI_BIAS_RELU) ? ACT_RELU
                    : epilogue == GDR_EPI_BIAS_GELU                             ? ACT_GELU
                                                                                : ACT_NONE;
    const int rc = launch_linear_f32_small(A, lda, W, ldw, C, ldc, M_max, N, K, nb, nr, act, bias, residual, ldr, splitk_ws,
                                           splitk_ws_bytes, stream, m_dev, nullptr, nullptr, sk ? sk->live : nullptr);
    if (rc <= 0) return rc;
  }
  return launch_linear_f32_dev(A, lda, W, ldw, C, ldc, M_max, m_dev, N, K, epilogue, bias, residual, ldr, -1, stream, sk);
}

// m_dev (may be null): the live row count on the device, *m_dev <= M.  Every kernel FORM is chosen from M alone, so a row's
// result does not depend on how many rows are live (the padded and the packed encoder forms agree bit for bit); tiles past
// the live rows exit at once and the stream-K forms deal only the live tiles' K-steps.  prof_rows: the live count if the
// host happens to know it — profiler flop accounting only.
int launch_linear_f32_ws(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M,
                         int N, int K, int epilogue, const float* bias, const float* residual, int64_t ldr,
                         float* splitk_ws, size_t splitk_ws_bytes, hipStream_t stream, StreamK* sk, const int64_t* m_dev,
                         int64_t prof_rows) {
  if (M == 0) return GDR_OK;  // empty batch: nothing to do (pointers of empty tensors may be null)
  GDR_CHECK_ARG(A && W && C, "linear: null pointer");
  GDR_CHECK_ARG(M >= 0 && N > 0 && K > 0, "linear: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  GDR_CHECK_ARG(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "linear: K, lda, ldw must be multiples of 4");
  GDR_CHECK_ARG(lda >= K && ldw >= K && ldc >= N, "linear: leading dimension smaller than the row");
  GDR_CHECK_ARG(aligned16(A) && aligned16(W), "linear: A and W must be 16-byte aligned");
  const bool needs_bias = epilogue == GDR_EPI_BIAS || epilogue == GDR_EPI_BIAS_RELU ||
                          epilogue == GDR_EPI_BIAS_RESIDUAL || epilogue == GDR_EPI_BIAS_GELU;
  const bool needs_res = epilogue == GDR_EPI_RESIDUAL || epilogue == GDR_EPI_BIAS_RESIDUAL;
  GDR_CHECK_ARG(!needs_bias || bias, "linear: epilogue %d needs bias", epilogue);
  GDR_CHECK_ARG(!needs_res || (residual && ldr >= N), "linear: epilogue %d needs residual", epilogue);
  if (M == 0) return GDR_OK;
  GemmArgs g{};
  g.A = A, g.W = W, g.C = C, g.bias = bias, g.residual = residual;
  g.lda = lda, g.ldw = ldw, g.ldc = ldc, g.ldr = ldr;
  g.M = M, g.N = N, g.K = K, g.m_dev = m_dev;
  g.tiles_n = (N + BN - 1) / BN;
  const int64_t tiles_m = (M + BM - 1) / BM;
  g.has_bias = needs_bias, g.has_residual = needs_res;
  g.act = (epilogue == GDR_EPI_RELU || epilogue == GDR_EPI_BIAS_RELU) ? ACT_RELU
          : epilogue == GDR_EPI_BIAS_GELU                             ? ACT_GELU
                                                                      : ACT_NONE;
  if (epilogue < GDR_EPI_NONE || epilogue > GDR_EPI_BIAS_GELU) {
    set_error("linear: unknown epilogue %d", epilogue);
    return GDR_EINVAL;
  }
  const double flops = 2.0 * (double)(m_dev && prof_rows >= 0 ? prof_rows : M) * (double)N * (double)K;  // profiler, by tile share
  const int64_t tiles = tiles_m * g.tiles_n;
  const int nk = K / BK;
  constexpr int64_t SLOTS = 512;                      // 256 CUs x 2 resident workgroups
  const size_t tile_bytes = (size_t)BM * BN * sizeof(float);
  auto split_launch = [&](int64_t first_tile, int64_t n_tiles, int S) -> int {
    const int chunk_steps = (nk + S - 1) / S;
    S = (nk + chunk_steps - 1) / chunk_steps;
    GemmArgs p = g;
    p.C = splitk_ws, p.has_bias = 0, p.has_residual = 0, p.act = ACT_NONE;
    p.ksplit = S, p.kchunk = chunk_steps * BK, p.tile_base = (int)first_tile;
    const int64_t blocks = n_tiles * S;
    {
      ProfScope prof(PROF_LINEAR, flops * (double)n_tiles / (double)tiles, stream);
      hipLaunchKernelGGL(gemm_nt_f32_kernel<EPI_LINEAR>, dim3((unsigned)blocks), dim3(GEMM_THREADS), 0, stream, p);
    }
    GDR_CHECK_LAUNCH("gemm_nt_f32_kernel(split)");
    ProfScope prof_r(PROF_REDUCE, 0.0, stream);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(n_tiles * 16)), dim3(256), 0, stream, splitk_ws, S,
                       (int)first_tile, g.tiles_n, M, N, C, ldc, needs_bias ? bias : nullptr,
                       needs_res ? residual : nullptr, ldr, g.act, m_dev);
    GDR_CHECK_LAUNCH("splitk_reduce_kernel");
    return GDR_OK;
  };
  if (tiles < 192 && M <= 1536 && nk >= 4 && K % BK == 0) {
    // few rows (decode: M = batch*beams): 64x64 tiles (gemm_small.hip), split along K only as far as still needed.
    // With more rows (a 64-query encoder batch) the 128x128 core with split-K measured 5 % faster.
    const int rc = launch_linear_f32_small(A, lda, W, ldw, C, ldc, M, N, K, needs_bias, needs_res, g.act, bias, residual, ldr,
                                           splitk_ws, splitk_ws_bytes, stream, m_dev, nullptr, nullptr, sk ? sk->live : nullptr);
    if (rc <= 0) return rc;
  }
  if (splitk_ws && K % BK == 0 && tiles < 0x7fffffff / 64) {
    if (tiles < 192 && nk >= 4) {
      // (a) the grid cannot fill the chip (decode: M = batch*beams rows): split every tile along K
      constexpr int target = 384;  // desired number of workgroups per split launch
      int S = (int)((target + tiles - 1) / tiles);
      if (S > nk / 2) S = nk / 2;
      if ((size_t)S * tiles * tile_bytes > splitk_ws_bytes) S = (int)(splitk_ws_bytes / (tiles * tile_bytes));
      if (S >= 2) {
        // A packed batch (device-side row count, live rows known to the host as a hint) whose LIVE (tile, chunk) units number
        // between one and two per CU: one workgroup per unit leaves every CU waiting for those that got two (a 64-query
        // encoder pass, N = 768: 78 live tiles x 4 chunks = 312 workgroups on 256 CUs, 94 us for K = 3072).  The stream-K
        // kernel on 256 workgroups deals the same units by K-step ranges instead — same chunks, same k order inside a chunk,
        // same slabs, same reduction: bit-identical — and the busiest CU runs ~1.2 units' worth instead of 2.
        const int chunk_steps = (nk + S - 1) / S;
        const int S2 = (nk + chunk_steps - 1) / chunk_steps;
        // Taken for every packed batch of this size class: the live row count is on the device (a host copy would cost a
        // synchronisation), and the kernel is right for any count — up to 256 live units each gets its own workgroup like
        // in the one-unit-per-workgroup form, above that they are dealt by ranges.  With the host's hint (the bench) a
        // launch whose units all fit one per CU keeps the plain form.
        const int64_t units_hint = m_dev && prof_rows >= 0 ? ((prof_rows + BM - 1) / BM) * g.tiles_n * S2 : -1;
        const size_t need = STREAMK_BYTES + (size_t)tiles * S2 * tile_bytes;
        if (sk && m_dev && sk->part == splitk_ws && nk % chunk_steps == 0 && tiles * S2 > 256 && (units_hint < 0 || units_hint > 256) &&
            need <= splitk_ws_bytes && streamk_fits(M, lda, N, ldw)) {
          if (int rc_ = streamk_poll_error()) return rc_;
          float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(splitk_ws) + STREAMK_BYTES);
          GemmArgs p = g;
          p.has_bias = 0, p.has_residual = 0, p.act = ACT_NONE;
          p.ksplit = 1 /* supertile height, unused at these widths */, p.kchunk = chunk_steps * BK, p.tile_base = 0;
          const StreamKArgs ska{sk->part, sk->flag, ++sk->epoch, streamk_err_word(), S2, slabs};
          {
            ProfScope prof(PROF_LINEAR, flops, stream);
            hipLaunchKernelGGL(gemm_nt_f32_streamk_kernel<true>, dim3(256), dim3(GEMM_THREADS), 0, stream, p, (int)(tiles * S2), ska);
          }
          GDR_CHECK_LAUNCH("gemm_nt_f32_streamk_kernel(units)");
          ProfScope prof_r(PROF_REDUCE, 0.0, stream);
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(tiles * 16)), dim3(256), 0, stream, slabs, S2, 0, g.tiles_n, M, N,
                             C, ldc, needs_bias ? bias : nullptr, needs_res ? residual : nullptr, ldr, g.act, m_dev);
          GDR_CHECK_LAUNCH("splitk_reduce_kernel");
          return GDR_OK;
        }
        return split_launch(0, tiles, S);
      }
    }
  }
  ProfScope prof(PROF_LINEAR, flops, stream);
  if (sk && streamk_mid_wanted(tiles) && K % BK == 0 && streamk_fits(M, lda, N, ldw)) {
    g.ksplit = g.tiles_n >= 12 ? 8 : 1;
    if (int rc_ = streamk_poll_error()) return rc_;
      const StreamKArgs ska{sk->part, sk->flag, ++sk->epoch, streamk_err_word()};
    hipLaunchKernelGGL(gemm_nt_f32_streamk_kernel<false>, dim3(256), dim3(GEMM_THREADS), 0, stream, g, (int)tiles, ska);
    GDR_CHECK_LAUNCH("gemm_nt_f32_streamk_kernel(256)");
    return GDR_OK;
  }
  if (tiles > SLOTS && tiles < 0x7fffffff && K % BK == 0) {  // more than one round of tiles: persistent form
    {
      // Measured HBM-side fetch per launch (PMC, M = 20480): N=3072 1112 -> 557 MB and N=2304 653 -> 443 MB with 8-panel
      // supertiles, but N=768 552 -> 684 MB (its 6 column tiles already fit one XCD's L2 next to 10 row panels): wide
      // outputs only.  Time is unchanged either way (the kernel is MFMA-bound); this is traffic and energy.
      g.ksplit = g.tiles_n >= 12 ? 8 : 1;
    }
    if (sk && streamk_wanted(tiles, K / BK) && streamk_fits(M, lda, N, ldw)) {
      if (int rc_ = streamk_poll_error()) return rc_;
      const StreamKArgs ska{sk->part, sk->flag, ++sk->epoch, streamk_err_word()};
      hipLaunchKernelGGL(gemm_nt_f32_streamk_kernel<false>, dim3((unsigned)SLOTS), dim3(GEMM_THREADS), 0, stream, g, (int)tiles, ska);
      GDR_CHECK_LAUNCH("gemm_nt_f32_streamk_kernel");
      return GDR_OK;
    }
    hipLaunchKernelGGL(gemm_nt_f32_persistent_kernel, dim3((unsigned)SLOTS), dim3(GEMM_THREADS), 0, stream, g, (int)tiles);
    GDR_CHECK_LAUNCH("gemm_nt_f32_persistent_kernel");
    return GDR_OK;
  }
  return launch<EPI_LINEAR>(g, tiles_m, stream);
}

// The linear over a device-side row count (ragged batches): rows [0, *m_dev) of A, *m_dev <= M_max.  The grid is sized for
// M_max and the kernels read the count themselves, so there is no host round trip.  No split-K / small-tile forms (their
// k order differs: a row's result must not depend on how many rows happen to be live) — callers use this at encoder
// batch sizes, where the 128x128 grid fills the chip anyway.  prof_rows: the live-row count if the host happens to
// know it (< 0: unknown, M_max is used) — only the opt-in profiler's flop accounting reads it.
int launch_linear_f32_dev(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M_max,
                          const int64_t* m_dev, int N, int K, int epilogue, const float* bias, const float* residual,
                          int64_t ldr, int64_t prof_rows, hipStream_t stream, StreamK* sk) {
  if (M_max == 0) return GDR_OK;
  GDR_CHECK_ARG(A && W && C && m_dev, "linear(dev rows): null pointer");
  GDR_CHECK_ARG(M_max > 0 && N > 0 && K > 0 && K % BK == 0, "linear(dev rows): bad shape M<=%lld N=%d K=%d (K %% 32 == 0)",
                (long long)M_max, N, K);
  GDR_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K && ldc >= N, "linear(dev rows): bad leading dimension");
  GDR_CHECK_ARG(aligned16(A) && aligned16(W), "linear(dev rows): A and W must be 16-byte aligned");
  GDR_CHECK_ARG(epilogue >= GDR_EPI_NONE && epilogue <= GDR_EPI_BIAS_GELU, "linear(dev rows): unknown epilogue %d", epilogue);
  const bool needs_bias = epilogue == GDR_EPI_BIAS || epilogue == GDR_EPI_BIAS_RELU ||
                          epilogue == GDR_EPI_BIAS_RESIDUAL || epilogue == GDR_EPI_BIAS_GELU;
  const bool needs_res = epilogue == GDR_EPI_RESIDUAL || epilogue == GDR_EPI_BIAS_RESIDUAL;
  GDR_CHECK_ARG(!needs_bias || bias, "linear(dev rows): epilogue %d needs bias", epilogue);
  GDR_CHECK_ARG(!needs_res || (residual && ldr >= N), "linear(dev rows): epilogue %d needs residual", epilogue);
  GemmArgs g{};
  g.A = A, g.W = W, g.C = C, g.bias = bias, g.residual = residual;
  g.lda = lda, g.ldw = ldw, g.ldc = ldc, g.ldr = ldr;
  g.M = M_max, g.N = N, g.K = K, g.m_dev = m_dev;
  g.tiles_n = (N + BN - 1) / BN;
  g.has_bias = needs_bias, g.has_residual = needs_res;
  g.act = (epilogue == GDR_EPI_RELU || epilogue == GDR_EPI_BIAS_RELU) ? ACT_RELU
          : epilogue == GDR_EPI_BIAS_GELU                             ? ACT_GELU
                                                                      : ACT_NONE;
  const int64_t tiles_m = (M_max + BM - 1) / BM, tiles = tiles_m * g.tiles_n;
  GDR_CHECK_ARG(tiles < 0x7fffffff, "linear(dev rows): grid too large");
  ProfScope prof(PROF_LINEAR, 2.0 * (double)(prof_rows >= 0 ? prof_rows : M_max) * (double)N * (double)K, stream);
  const int64_t tiles_live = prof_rows >= 0 ? ((prof_rows + BM - 1) / BM) * g.tiles_n : tiles;
  if (sk && tiles > 256 && streamk_mid_wanted(tiles_live) && streamk_fits(M_max, lda, N, ldw)) {
    g.ksplit = g.tiles_n >= 12 ? 8 : 1;
    if (int rc_ = streamk_poll_error()) return rc_;
      const StreamKArgs ska{sk->part, sk->flag, ++sk->epoch, streamk_err_word()};
    hipLaunchKernelGGL(gemm_nt_f32_streamk_kernel<false>, dim3(256), dim3(GEMM_THREADS), 0, stream, g, (int)tiles, ska);
    GDR_CHECK_LAUNCH("gemm_nt_f32_streamk_kernel(256, dev rows)");
    return GDR_OK;
  }
  if (tiles > 512) {
    g.ksplit = g.tiles_n >= 12 ? 8 : 1;  // supertile height, as in launch_linear_f32_ws
    if (sk && streamk_wanted(tiles_live, K / BK) && streamk_fits(M_max, lda, N, ldw)) {
      if (int rc_ = streamk_poll_error()) return rc_;
      const StreamKArgs ska{sk->part, sk->flag, ++sk->epoch, streamk_err_word()};
      hipLaunchKernelGGL(gemm_nt_f32_streamk_kernel<false>, dim3(512), dim3(GEMM_THREADS), 0, stream, g, (int)tiles, ska);
      GDR_CHECK_LAUNCH("gemm_nt_f32_streamk_kernel(dev rows)");
      return GDR_OK;
    }
    hipLaunchKernelGGL(gemm_nt_f32_persistent_kernel, dim3(512), dim3(GEMM_THREADS), 0, stream, g, (int)tiles);
    GDR_CHECK_LAUNCH("gemm_nt_f32_persistent_kernel(dev rows)");
    return GDR_OK;
  }
  return launch<EPI_LINEAR>(g, tiles_m, stream);
}

// bf16 operands (fp32 accumulate, fp32 output): the precision mode of config C5.  Same tiling as the fp32 linears on
// v_mfma_f32_32x32x16_bf16; no split-K form.
int launch_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                       int K, int epilogue, const float* bias, const float* residual, int64_t ldr, hipStream_t stream,
                       const int64_t* m_dev) {
  if (M == 0) return GDR_OK;
  GDR_CHECK_ARG(A && W && C, "linear_bf16: null pointer");
  GDR_CHECK_ARG(M >= 0 && N > 0 && K > 0, "linear_bf16: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  GDR_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "linear_bf16: K, lda, ldw must be multiples of 8");
  GDR_CHECK_ARG(lda >= K && ldw >= K && ldc >= N, "linear_bf16: leading dimension smaller than the row");
  GDR_CHECK_ARG(aligned16(A) && aligned16(W), "linear_bf16: A and W must be 16-byte aligned");
  if (epilogue < GDR_EPI_NONE || epilogue > GDR_EPI_BIAS_GELU) {
    set_error("linear_bf16: unknown epilogue %d", epilogue);
    return GDR_EINVAL;
  }
  const bool needs_bias = epilogue == GDR_EPI_BIAS || epilogue == GDR_EPI_BIAS_RELU ||
                          epilogue == GDR_EPI_BIAS_RESIDUAL || epilogue == GDR_EPI_BIAS_GELU;
  const bool needs_res = epilogue == GDR_EPI_RESIDUAL || epilogue == GDR_EPI_BIAS_RESIDUAL;
  GDR_CHECK_ARG(!needs_bias || bias, "linear_bf16: epilogue %d needs bias", epilogue);
  GDR_CHECK_ARG(!needs_res || (residual && ldr >= N), "linear_bf16: epilogue %d needs residual", epilogue);
  if (M == 0) return GDR_OK;
  GemmArgs g{};
  g.A = static_cast<const float*>(A), g.W = static_cast<const float*>(W);  // opaque: the kernel addresses operands in bytes
  g.C = C, g.bias = bias, g.residual = residual;
  g.lda = lda, g.ldw = ldw, g.ldc = ldc, g.ldr = ldr;
  g.M = M, g.N = N, g.K = K;
  g.tiles_n = (N + BN - 1) / BN;
  g.has_bias = needs_bias, g.has_residual = needs_res;
  g.act = (epilogue == GDR_EPI_RELU || epilogue == GDR_EPI_BIAS_RELU) ? ACT_RELU
          : epilogue == GDR_EPI_BIAS_GELU                             ? ACT_GELU
                                                                      : ACT_NONE;
  ProfScope prof(PROF_LINEAR, 2.0 * (double)M * (double)N * (double)K, stream);
  {  // the LDS-DMA kernel (gemm_bf16.hip) serves K % 64 == 0 with 16-byte aligned operands; other shapes the generic core
    const int rc = launch_linear_bf16_glds(A, lda, W, ldw, C, ldc, M, N, K, needs_bias, needs_res, g.act, bias, residual, ldr, 0,
                                           stream, m_dev);
    if (rc <= 0) return rc;
  }
  g.m_dev = m_dev;
  return launch<EPI_LINEAR, true>(g, (M + BM - 1) / BM, stream);
}

int launch_sim_gemm(const void* D_, int64_t N, const void* Q_, int B, int d, const SimEpilogue& ep, bool bf16,
                    hipStream_t stream) {
  const float* D = static_cast<const float*>(D_);   // opaque: the kernel addresses operands in bytes
  const float* Q = static_cast<const float*>(Q_);
  GemmArgs g{};
  g.A = D, g.W = Q, g.lda = d, g.ldw = d;
  g.M = N, g.N = B, g.K = d;
  g.tiles_n = (B + BN - 1) / BN;
  g.sim = ep;
  const int64_t tiles_m = (N + BM - 1) / BM;
  const int64_t n_sample_tiles = (tiles_m + ep.tile_stride - 1) / ep.tile_stride;
  if (ep.mode == 1) {
    int64_t rows = n_sample_tiles * BM;
    if (rows > N) rows = N;
    ProfScope prof(PROF_SIM_SAMPLE, 2.0 * (double)rows * (double)B * (double)d, stream);
    if (bf16) {
      const int rc = launch_sim_bf16_glds(D_, N, Q_, B, d, ep, n_sample_tiles, stream);
      if (rc <= 0) return rc;
    }
    return bf16 ? launch<EPI_SIM_SAMPLE, true>(g, n_sample_tiles, stream) : launch<EPI_SIM_SAMPLE>(g, n_sample_tiles, stream);
  }
  int64_t rows = (tiles_m - n_sample_tiles) * BM;
  if (rows > N) rows = N;
  ProfScope prof(PROF_SIM_FILTER, 2.0 * (double)rows * (double)B * (double)d, stream);
  if (bf16) {
    const int rc = launch_sim_bf16_glds(D_, N, Q_, B, d, ep, tiles_m - n_sample_tiles, stream);
    if (rc <= 0) return rc;
  }
  return bf16 ? launch<EPI_SIM_FILTER, true>(g, tiles_m - n_sample_tiles, stream)
              : launch<EPI_SIM_FILTER>(g, tiles_m - n_sample_tiles, stream);
}

}  // namespace gdr

extern "C" int gdr_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                              int64_t M, int N, int K, int epilogue, const float* bias, const float* residual,
                              int64_t ldr, void* stream) {
  return gdr::launch_linear_f32(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, residual, ldr,
                                static_cast<hipStream_t>(stream));
}

extern "C" int gdr_linear_f32_splitk(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                                     int64_t M, int N, int K, int epilogue, const float* bias, const float* residual,
                                     int64_t ldr, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace gdr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // a workspace that can hold the stream-K hand-off scratch also serves the balanced forms of the big grids (the same
  // region: a launch uses it for split-K slabs or for the hand-off, never both); its 2 KB of flags are zeroed per call
  const int64_t tiles = ((M + BM - 1) / BM) * (int64_t)((N + BN - 1) / BN);
  if (workspace && workspace_bytes >= STREAMK_BYTES && K > 0 && K % BK == 0 && M > 0 && N > 0 && lda >= K && ldw >= K &&
      (streamk_mid_wanted(tiles) || (tiles > 512 && streamk_wanted(tiles, K / BK))) && streamk_fits(M, lda, N, ldw)) {
    StreamK sk{static_cast<float*>(workspace),
               reinterpret_cast<int32_t*>(static_cast<char*>(workspace) + STREAMK_PART_BYTES), 0};
    if (hipMemsetAsync(sk.flag, 0, 512 * sizeof(int32_t), st) != hipSuccess) {
      set_error("linear: memset of the stream-K flags failed");
      return GDR_EHIP;
    }
    return launch_linear_f32_ws(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, residual, ldr, static_cast<float*>(workspace),
                                workspace_bytes, st, &sk);
  }
  return launch_linear_f32_ws(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, residual, ldr, static_cast<float*>(workspace),
                              workspace_bytes, st);
}

extern "C" int gdr_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int64_t M,
                               int N, int K, int epilogue, const float* bias, const float* residual, int64_t ldr,
                               void* stream) {
  return gdr::launch_linear_bf16(A, lda, W, ldw, C, ldc, M, N, K, epilogue, bias, residual, ldr,
                                 static_cast<hipStream_t>(stream));
}
