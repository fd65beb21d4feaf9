// Latency-mode similarity (B <= 32 queries, fp32): the HBM-bound regime of SURVEY §8d / the north-star's "coalesced
// HBM reads".  At B <= 40 the contraction Q·Dᵀ has intensity 2B/4 flop per corpus byte, below the fp32-MFMA ridge, so
// the corpus stream is the roof — and the general 128x128-tile GEMM core wastes 3/4 of its MFMA work on absent queries.
//
// Shape: the queries are STATIONARY — Q (<= 32 x d fp32, <= 128 KB) is loaded once per workgroup into LDS — and every
// wave streams its own 32-doc slices of the corpus straight from HBM into MFMA A-operand registers: no LDS round trip
// for the streamed operand, no workgroup barrier after the prologue (waves are independent), a 4-deep register
// prefetch ring (4 x 64 B per lane in flight) to cover HBM latency.  One persistent workgroup of 8 waves per CU.
//   per 32-k group and wave: 4 global_load_dwordx4 (each lane: 64 contiguous bytes of its doc row; the two lane
//   halves cover one 128-B line) + 4 ds_read_b128 of the Q fragments (row stride d+4 floats: conflict-free) +
//   16 v_mfma_f32_32x32x2_f32; the k index inside a group is permuted identically on both operands.
// Epilogues are those of the GEMM core: store the strided sample / compare with thr[q] and append survivors.
#include <stdlib.h>

#include "common.h"

namespace gdr {

typedef float f32x16s __attribute__((ext_vector_type(16)));

struct StreamArgs {
  const float* D;
  const float* Q;
  int64_t N;
  int B, d;
  int64_t n_tiles;   // 128-doc tiles of this launch
  SimEpilogue sim;
  int capl;          // filter pass: survivors a workgroup collects per query in LDS before its ONE append to the query's list
};

#ifndef STREAM_THREADS
#define STREAM_THREADS 512
#endif
constexpr int STREAM_WAVES = STREAM_THREADS / 64;

template <int MODE>  // 1 sample, 2 filter
__global__ __launch_bounds__(STREAM_THREADS) void sim_stream_f32_kernel(const StreamArgs g) {
  extern __shared__ __attribute__((aligned(16))) float qs[];  // [32][d + 4]
  const int d = g.d, QS = d + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  // sample pass: the tickets of the sliced threshold / select tails (sim_topk.hip sim_sliced_select_kernel) start this call at zero
  if (MODE == 1 && blockIdx.x == 0 && tid < g.B) g.sim.cand_cnt[tid * CNT_STRIDE + 1] = 0, g.sim.cand_cnt[tid * CNT_STRIDE + 2] = 0;
  for (int e = tid; e < 32 * (d >> 2); e += STREAM_THREADS) {
    const int r = e / (d >> 2), c = e - r * (d >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < g.B) v = *reinterpret_cast<const float4*>(g.Q + (int64_t)r * d + 4 * c);
    *reinterpret_cast<float4*>(qs + r * QS + 4 * c) = v;
  }
  __syncthreads();
  const float* qrow = qs + l31 * QS + 16 * h;
  const int G = d >> 5;  // 32-k groups, multiple of 4
  const float thr = (MODE == 2 && l31 < g.B) ? g.sim.thr[l31] : 0.f;
  float* cv = g.sim.cand_val + (int64_t)l31 * g.sim.cap;
  int32_t* ci = g.sim.cand_idx + (int64_t)l31 * g.sim.cap;
  // Filter pass: survivors go to per-workgroup lists in LDS and reach the query's global list with ONE returning atomic per
  // (workgroup, query) at the end.  One atomic per (32-doc slice, query) on 32 shared counters was ~4 000 same-address
  // atomics per counter (~12 ns each, serialised in L2): the pass took 222 us at 32 queries against 174 us at one.  A list that
  // fills up (capl entries; the expected load is ~1/9 of it) sends the rest of that query's survivors down the direct path;
  // lfirst[q] = the LDS count at the first reservation that did not fit = the number of entries that are really there.
  const int capl = MODE == 2 ? g.capl : 0;
  int* lcnt = reinterpret_cast<int*>(qs + 32 * QS);  // [32]
  int* lfirst = lcnt + 32;                           // [32]
  // row stride capl + 1: with a stride that is a multiple of 64 words the 32 queries' lists start in ONE bank and every survivor
  // store of a wave was a 32-way conflict
  const int lstr = capl + 1;
  float* lval = reinterpret_cast<float*>(lfirst + 32);  // [32][capl + 1]
  int* lidx = reinterpret_cast<int*>(lval + 32 * lstr);  // [32][capl + 1]
  if (capl && tid < 32) lcnt[tid] = 0, lfirst[tid] = capl;
  if (capl) __syncthreads();

  const int64_t n_units = g.n_tiles * 4;  // 32-doc slices
  for (int64_t u = (int64_t)blockIdx.x * STREAM_WAVES + wave; u < n_units; u += (int64_t)gridDim.x * STREAM_WAVES) {
    int64_t t = u >> 2;
    const int sub = (int)(u & 3);
    int64_t slot_base = 0;
    if (MODE == 1) {
      slot_base = t * 128;
      t = t * g.sim.tile_stride;
    } else {
      const int s1 = g.sim.tile_stride - 1;
      t = (t / s1) * g.sim.tile_stride + 1 + (t % s1);
    }
    const int64_t m0 = t * 128 + sub * 32;
    int64_t row = m0 + l31;
    row = row < g.N ? row : g.N - 1;
    const float* drow = g.D + row * d + 16 * h;

    f32x16s acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 a0[4], a1[4], a2[4], a3[4];
#define LOADG(A, grp)                                                         \
  A[0] = *reinterpret_cast<const float4*>(drow + 32 * (grp));                 \
  A[1] = *reinterpret_cast<const float4*>(drow + 32 * (grp) + 4);             \
  A[2] = *reinterpret_cast<const float4*>(drow + 32 * (grp) + 8);             \
  A[3] = *reinterpret_cast<const float4*>(drow + 32 * (grp) + 12); \
  asm volatile("" ::: "memory"); /* pin the issue point: the optimiser otherwise sinks the prefetch to its use */
#define MFMA4(av, bv)                                                  \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
#define COMPUTE(A, grp)                                                                  \
  {                                                                                      \
    const float4 b0 = *reinterpret_cast<const float4*>(qrow + 32 * (grp));               \
    const float4 b1 = *reinterpret_cast<const float4*>(qrow + 32 * (grp) + 4);           \
    const float4 b2 = *reinterpret_cast<const float4*>(qrow + 32 * (grp) + 8);           \
    const float4 b3 = *reinterpret_cast<const float4*>(qrow + 32 * (grp) + 12);          \
    MFMA4(A[0], b0) MFMA4(A[1], b1) MFMA4(A[2], b2) MFMA4(A[3], b3)                      \
  }
    LOADG(a0, 0) LOADG(a1, 1) LOADG(a2, 2)
    int gq = 0;
    for (; gq + 8 <= G; gq += 4) {  // ring of 4 register sets, 3 groups in flight ahead of the one consumed;
      LOADG(a3, gq + 3)             // branch-free body so the compiler keeps counted vmcnt waits
      COMPUTE(a0, gq)
      LOADG(a0, gq + 4)
      COMPUTE(a1, gq + 1)
      LOADG(a1, gq + 5)
      COMPUTE(a2, gq + 2)
      LOADG(a2, gq + 6)
      COMPUTE(a3, gq + 3)
    }
    LOADG(a3, gq + 3)
    COMPUTE(a0, gq)
    COMPUTE(a1, gq + 1)
    COMPUTE(a2, gq + 2)
    COMPUTE(a3, gq + 3)
#undef LOADG
#undef MFMA4
#undef COMPUTE
    // accumulator map: col (query) = l31, row (doc in slice) = (r&3) + 8*(r>>2) + 4*h
    if (l31 < g.B) {
      if (MODE == 1) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int roff = sub * 32 + 8 * q4 + 4 * h;
          const int64_t m = m0 + 8 * q4 + 4 * h;
          float4 v;
          int4 id;
          v.x = m + 0 < g.N ? acc[4 * q4 + 0] : -INFINITY, v.y = m + 1 < g.N ? acc[4 * q4 + 1] : -INFINITY;
          v.z = m + 2 < g.N ? acc[4 * q4 + 2] : -INFINITY, v.w = m + 3 < g.N ? acc[4 * q4 + 3] : -INFINITY;
          id.x = m + 0 < g.N ? (int)m : -1, id.y = m + 1 < g.N ? (int)m + 1 : -1;
          id.z = m + 2 < g.N ? (int)m + 2 : -1, id.w = m + 3 < g.N ? (int)m + 3 : -1;
          *reinterpret_cast<float4*>(cv + slot_base + roff) = v;
          *reinterpret_cast<int4*>(ci + slot_base + roff) = id;
        }
      } else {
        unsigned keep = 0u;  // flag survivors, pool the two lane halves of a query, reserve with ONE returning atomic
        if (m0 + 32 <= g.N) {  // uniform: only the corpus' last slice tests rows
#pragma unroll
          for (int r = 0; r < 16; ++r) keep |= acc[r] >= thr ? (1u << r) : 0u;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            keep |= (acc[r] >= thr && m < g.N) ? (1u << r) : 0u;
          }
        }
        const int mine = __popc(keep);
        const int other = __shfl_xor(mine, 32);
        const int total = mine + other;
        int base = 0;
        bool local = false;
        if (capl) {
          if (h == 0 && total > 0) {
            base = atomicAdd(lcnt + l31, total);  // LDS
            local = base + total <= capl;
            if (!local) atomicMin(lfirst + l31, base);
          }
          local = __shfl((int)local, l31) != 0;
        }
        if (local) {
          base = __shfl(base, l31);
          int pos = l31 * lstr + base + (h ? other : 0);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (keep >> r & 1u) {
              lval[pos] = acc[r];
              lidx[pos] = (int)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h);
              ++pos;
            }
        } else {
          if (h == 0 && total > 0) base = atomicAdd(g.sim.cand_cnt + l31 * CNT_STRIDE, total);
          base = __shfl(base, l31);
          int pos = base + (h ? other : 0);
          if (mine) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (keep >> r & 1u) {
                if (pos < g.sim.cap) {
                  cv[pos] = acc[r];
                  ci[pos] = (int)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h);
                }
                ++pos;
              }
          }
        }
      }
    }
  }
  if (capl) {  // flush: wave w appends the lists of queries w, w + 8, ...
    __syncthreads();
    for (int q = wave; q < g.B; q += STREAM_WAVES) {
      const int n = min(lcnt[q], lfirst[q]);
      if (n <= 0) continue;  // wave-uniform
      int gbase = 0;
      if (lane == 0) gbase = atomicAdd(g.sim.cand_cnt + q * CNT_STRIDE, n);
      gbase = __shfl(gbase, 0);
      float* qv = g.sim.cand_val + (int64_t)q * g.sim.cap;
      int32_t* qi = g.sim.cand_idx + (int64_t)q * g.sim.cap;
      for (int i = lane; i < n; i += 64)
        if (gbase + i < g.sim.cap) qv[gbase + i] = lval[q * lstr + i], qi[gbase + i] = lidx[q * lstr + i];
    }
  }
}

// The same stream for a bf16 corpus / bf16 queries (gdr_sim_topk_bf16, and the corpus-wide pass of gdr_sim_topk_prefilter) at
// B <= 32: half the bytes per doc, v_mfma_f32_32x32x16_bf16 (an eighth of the fp32 form's matrix-pipe time, so the pass is the
// HBM stream alone).  Per 64-k group a lane reads 64 contiguous bytes of its doc row (k = 64 g + 32 h .. + 31: the two lane halves
// cover one 128-byte line) as four 16-byte MFMA A operands; the matching Q fragment (row l31, the same 8 k values) comes from LDS.
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
template <int MODE>  // 1 sample, 2 filter
__global__ __launch_bounds__(STREAM_THREADS) void sim_stream_bf16_kernel(const StreamArgs g) {
  extern __shared__ __attribute__((aligned(16))) char qsb[];  // [32][d * 2 + 16] bytes
  const int d = g.d, QSB = d * 2 + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const char* Qb = reinterpret_cast<const char*>(g.Q);
  // sample pass: the tickets of the sliced threshold / select tails (sim_topk.hip sim_sliced_select_kernel) start this call at zero
  if (MODE == 1 && blockIdx.x == 0 && tid < g.B) g.sim.cand_cnt[tid * CNT_STRIDE + 1] = 0, g.sim.cand_cnt[tid * CNT_STRIDE + 2] = 0;
  for (int e = tid; e < 32 * (d >> 3); e += STREAM_THREADS) {
    const int r = e / (d >> 3), c = e - r * (d >> 3);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < g.B) v = *reinterpret_cast<const float4*>(Qb + ((int64_t)r * d + 8 * c) * 2);
    *reinterpret_cast<float4*>(qsb + r * QSB + 16 * c) = v;
  }
  __syncthreads();
  const char* qrow = qsb + l31 * QSB + 64 * h;
  const int G = d >> 6;  // 64-k groups, multiple of 4
  const float thr = (MODE == 2 && l31 < g.B) ? g.sim.thr[l31] : 0.f;
  float* cv = g.sim.cand_val + (int64_t)l31 * g.sim.cap;
  int32_t* ci = g.sim.cand_idx + (int64_t)l31 * g.sim.cap;
  const int capl = MODE == 2 ? g.capl : 0;
  int* lcnt = reinterpret_cast<int*>(qsb + 32 * QSB);  // [32]
  int* lfirst = lcnt + 32;                            // [32]
  // row stride capl + 1: with a stride that is a multiple of 64 words the 32 queries' lists start in ONE bank and every survivor
  // store of a wave was a 32-way conflict
  const int lstr = capl + 1;
  float* lval = reinterpret_cast<float*>(lfirst + 32);  // [32][capl + 1]
  int* lidx = reinterpret_cast<int*>(lval + 32 * lstr);  // [32][capl + 1]
  if (capl && tid < 32) lcnt[tid] = 0, lfirst[tid] = capl;
  if (capl) __syncthreads();

  const int64_t n_units = g.n_tiles * 4;  // 32-doc slices
  for (int64_t u = (int64_t)blockIdx.x * STREAM_WAVES + wave; u < n_units; u += (int64_t)gridDim.x * STREAM_WAVES) {
    int64_t t = u >> 2;
    const int sub = (int)(u & 3);
    int64_t slot_base = 0;
    if (MODE == 1) {
      slot_base = t * 128;
      t = t * g.sim.tile_stride;
    } else {
      const int s1 = g.sim.tile_stride - 1;
      t = (t / s1) * g.sim.tile_stride + 1 + (t % s1);
    }
    const int64_t m0 = t * 128 + sub * 32;
    int64_t row = m0 + l31;
    row = row < g.N ? row : g.N - 1;
    const char* drow = reinterpret_cast<const char*>(g.D) + row * d * 2 + 64 * h;

    f32x16s acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 a0[4], a1[4], a2[4], a3[4];
#define LOADG(A, grp)                                                          \
  A[0] = *reinterpret_cast<const float4*>(drow + 128 * (grp));                 \
  A[1] = *reinterpret_cast<const float4*>(drow + 128 * (grp) + 16);            \
  A[2] = *reinterpret_cast<const float4*>(drow + 128 * (grp) + 32);            \
  A[3] = *reinterpret_cast<const float4*>(drow + 128 * (grp) + 48);            \
  asm volatile("" ::: "memory"); /* pin the issue point: the optimiser otherwise sinks the prefetch to its use */
#define COMPUTE(A, grp)                                                                                      \
  {                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
      const float4 b_ = *reinterpret_cast<const float4*>(qrow + 128 * (grp) + 16 * j);                       \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, A[j]), __builtin_bit_cast(bf16x8s, b_), acc, 0, 0, 0); \
    }                                                                                                        \
  }
    LOADG(a0, 0) LOADG(a1, 1) LOADG(a2, 2)
    int gq = 0;
    for (; gq + 8 <= G; gq += 4) {
      LOADG(a3, gq + 3)
      COMPUTE(a0, gq)
      LOADG(a0, gq + 4)
      COMPUTE(a1, gq + 1)
      LOADG(a1, gq + 5)
      COMPUTE(a2, gq + 2)
      LOADG(a2, gq + 6)
      COMPUTE(a3, gq + 3)
    }
    LOADG(a3, gq + 3)
    COMPUTE(a0, gq)
    COMPUTE(a1, gq + 1)
    COMPUTE(a2, gq + 2)
    COMPUTE(a3, gq + 3)
#undef LOADG
#undef COMPUTE
    // accumulator map: col (query) = l31, row (doc in slice) = (r&3) + 8*(r>>2) + 4*h
    if (l31 < g.B) {
      if (MODE == 1) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int roff = sub * 32 + 8 * q4 + 4 * h;
          const int64_t m = m0 + 8 * q4 + 4 * h;
          float4 v;
          int4 id;
          v.x = m + 0 < g.N ? acc[4 * q4 + 0] : -INFINITY, v.y = m + 1 < g.N ? acc[4 * q4 + 1] : -INFINITY;
          v.z = m + 2 < g.N ? acc[4 * q4 + 2] : -INFINITY, v.w = m + 3 < g.N ? acc[4 * q4 + 3] : -INFINITY;
          id.x = m + 0 < g.N ? (int)m : -1, id.y = m + 1 < g.N ? (int)m + 1 : -1;
          id.z = m + 2 < g.N ? (int)m + 2 : -1, id.w = m + 3 < g.N ? (int)m + 3 : -1;
          *reinterpret_cast<float4*>(cv + slot_base + roff) = v;
          *reinterpret_cast<int4*>(ci + slot_base + roff) = id;
        }
      } else {
        unsigned keep = 0u;
        if (m0 + 32 <= g.N) {  // uniform: only the corpus' last slice tests rows
#pragma unroll
          for (int r = 0; r < 16; ++r) keep |= acc[r] >= thr ? (1u << r) : 0u;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            keep |= (acc[r] >= thr && m < g.N) ? (1u << r) : 0u;
          }
        }
        const int mine = __popc(keep);
        const int other = __shfl_xor(mine, 32);
        const int total = mine + other;
        int base = 0;
        bool local = false;
        if (capl) {
          if (h == 0 && total > 0) {
            base = atomicAdd(lcnt + l31, total);  // LDS
            local = base + total <= capl;
            if (!local) atomicMin(lfirst + l31, base);
          }
          local = __shfl((int)local, l31) != 0;
        }
        if (local) {
          base = __shfl(base, l31);
          int pos = l31 * lstr + base + (h ? other : 0);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (keep >> r & 1u) {
              lval[pos] = acc[r];
              lidx[pos] = (int)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h);
              ++pos;
            }
        } else {
          if (h == 0 && total > 0) base = atomicAdd(g.sim.cand_cnt + l31 * CNT_STRIDE, total);
          base = __shfl(base, l31);
          int pos = base + (h ? other : 0);
          if (mine) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (keep >> r & 1u) {
                if (pos < g.sim.cap) {
                  cv[pos] = acc[r];
                  ci[pos] = (int)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h);
                }
                ++pos;
              }
          }
        }
      }
    }
  }
  if (capl) {  // flush: wave w appends the lists of queries w, w + 8, ...
    __syncthreads();
    for (int q = wave; q < g.B; q += STREAM_WAVES) {
      const int n = min(lcnt[q], lfirst[q]);
      if (n <= 0) continue;  // wave-uniform
      int gbase = 0;
      if (lane == 0) gbase = atomicAdd(g.sim.cand_cnt + q * CNT_STRIDE, n);
      gbase = __shfl(gbase, 0);
      float* qv = g.sim.cand_val + (int64_t)q * g.sim.cap;
      int32_t* qi = g.sim.cand_idx + (int64_t)q * g.sim.cap;
      for (int i = lane; i < n; i += 64)
        if (gbase + i < g.sim.cap) qv[gbase + i] = lval[q * lstr + i], qi[gbase + i] = lidx[q * lstr + i];
    }
  }
}

// Sample pass when it holds only a few hundred 32-doc slices (sqrt(kN) docs): one slice per WORKGROUP, the K range
// dealt round-robin to the 8 waves (every load of a slice in flight at once instead of a 24-group dependent chain per
// wave), partial accumulators summed through LDS in fixed wave order.
__global__ __launch_bounds__(STREAM_THREADS) void sim_stream_sample_splitk_kernel(const StreamArgs g) {
  extern __shared__ __attribute__((aligned(16))) float qs[];  // [32][d + 4] then red[STREAM_WAVES][16][64]
  const int d = g.d, QS = d + 4;
  float* red = qs + 32 * QS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  // sample pass: the tickets of the sliced threshold / select tails (sim_topk.hip sim_sliced_select_kernel) start this call at zero
  if (blockIdx.x == 0 && tid < g.B) g.sim.cand_cnt[tid * CNT_STRIDE + 1] = 0, g.sim.cand_cnt[tid * CNT_STRIDE + 2] = 0;
  const int64_t u = blockIdx.x;
  const int64_t t = (u >> 2) * g.sim.tile_stride;
  const int sub = (int)(u & 3);
  const int64_t slot_base = (u >> 2) * 128;
  const int64_t m0 = t * 128 + sub * 32;
  int64_t row = m0 + l31;
  row = row < g.N ? row : g.N - 1;
  const float* drow = g.D + row * d + 16 * h;
  const int G = d >> 5;
  float4 a[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int grp = wave + STREAM_WAVES * i;
    const float* src = drow + 32 * (grp < G ? grp : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[i][j] = *reinterpret_cast<const float4*>(src + 4 * j);
  }
  for (int e = tid; e < 32 * (d >> 2); e += STREAM_THREADS) {
    const int r = e / (d >> 2), c = e - r * (d >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < g.B) v = *reinterpret_cast<const float4*>(g.Q + (int64_t)r * d + 4 * c);
    *reinterpret_cast<float4*>(qs + r * QS + 4 * c) = v;
  }
  __syncthreads();
  const float* qrow = qs + l31 * QS + 16 * h;
  f32x16s acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int grp = wave + STREAM_WAVES * i;
    if (grp < G) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(qrow + 32 * grp + 4 * j);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j].x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j].y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j].z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j].w, b.w, acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (l31 < g.B) {
    float* cv = g.sim.cand_val + (int64_t)l31 * g.sim.cap + slot_base + sub * 32;
    int32_t* ci = g.sim.cand_idx + (int64_t)l31 * g.sim.cap + slot_base + sub * 32;
    for (int r = wave; r < 16; r += STREAM_WAVES) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < STREAM_WAVES; ++w) v += red[(w * 16 + r) * 64 + lane];
      const int roff = (r & 3) + 8 * (r >> 2) + 4 * h;
      const bool ok = m0 + roff < g.N;
      cv[roff] = ok ? v : -INFINITY;
      ci[roff] = ok ? (int)(m0 + roff) : -1;
    }
  }
}

bool sim_stream_supported(int B, int d, bool bf16) {
  return bf16 ? (B <= 32 && d % 256 == 0 && d <= 2048) : (B <= 32 && d % 128 == 0 && d <= 1024);
}

int launch_sim_stream(const void* D, int64_t N, const void* Q, int B, int d, const SimEpilogue& ep, bool bf16, hipStream_t stream) {
  StreamArgs g{};
  g.D = static_cast<const float*>(D), g.Q = static_cast<const float*>(Q), g.N = N, g.B = B, g.d = d, g.sim = ep;  // opaque in the bf16 form
  const int64_t tiles_m = (N + 127) / 128;
  const int64_t n_sample_tiles = (tiles_m + ep.tile_stride - 1) / ep.tile_stride;
  g.n_tiles = ep.mode == 1 ? n_sample_tiles : tiles_m - n_sample_tiles;
  if (g.n_tiles <= 0) return GDR_OK;
  size_t lds = bf16 ? (size_t)32 * (d * 2 + 16) : (size_t)32 * (d + 4) * sizeof(float);
  const size_t lds_q = lds;
  if (ep.mode == 2) {  // per-workgroup survivor lists beside the queries: up to 192 entries per query, what LDS leaves otherwise
    static const int capl_max = [] { const char* e = getenv("GDR_SIM_LOCAL_LIST"); return e ? atoi(e) : 192; }();
    int64_t room = ((int64_t)160 * 1024 - (int64_t)lds_q - 256) / (32 * 8) - 1;
    room = room > capl_max ? capl_max : room;
    g.capl = room >= 16 ? (int)(room & ~7) : 0;
    if (g.capl) lds += 256 + (size_t)32 * (g.capl + 1) * 8;
  }
  for (const void* fn : {reinterpret_cast<const void*>(sim_stream_f32_kernel<1>),
                         reinterpret_cast<const void*>(sim_stream_f32_kernel<2>),
                         reinterpret_cast<const void*>(sim_stream_bf16_kernel<1>),
                         reinterpret_cast<const void*>(sim_stream_bf16_kernel<2>),
                         reinterpret_cast<const void*>(sim_stream_sample_splitk_kernel)})
    if (int rc = ensure_dyn_lds(fn, 160 * 1024, "sim_stream")) return rc;
  int64_t blocks = (g.n_tiles * 4 + STREAM_WAVES - 1) / STREAM_WAVES;
  if (blocks > 256) blocks = 256;  // persistent: one workgroup per CU
  const double rows = (double)(g.n_tiles * 128 < N ? g.n_tiles * 128 : N);
  ProfScope prof(ep.mode == 1 ? PROF_SIM_SAMPLE : PROF_SIM_FILTER, 2.0 * rows * B * d, stream);
  const size_t lds_split = lds_q + (size_t)STREAM_WAVES * 16 * 64 * sizeof(float);
  if (bf16) {  // no split-K form: a slice is 12 groups of the ring, every wave takes at most a slice or two of the sample pass
    if (ep.mode == 1)
      hipLaunchKernelGGL(sim_stream_bf16_kernel<1>, dim3((unsigned)blocks), dim3(STREAM_THREADS), lds, stream, g);
    else
      hipLaunchKernelGGL(sim_stream_bf16_kernel<2>, dim3((unsigned)blocks), dim3(STREAM_THREADS), lds, stream, g);
  } else if (ep.mode == 1 && g.n_tiles * 4 <= 128 * STREAM_WAVES && lds_split <= 160 * 1024)  // fewer slices than half the waves
    hipLaunchKernelGGL(sim_stream_sample_splitk_kernel, dim3((unsigned)(g.n_tiles * 4)), dim3(STREAM_THREADS), lds_split,
                       stream, g);
  else if (ep.mode == 1)
    hipLaunchKernelGGL(sim_stream_f32_kernel<1>, dim3((unsigned)blocks), dim3(STREAM_THREADS), lds, stream, g);
  else
    hipLaunchKernelGGL(sim_stream_f32_kernel<2>, dim3((unsigned)blocks), dim3(STREAM_THREADS), lds, stream, g);
  GDR_CHECK_LAUNCH("sim_stream_f32_kernel");
  return GDR_OK;
}

}  // namespace gdr
