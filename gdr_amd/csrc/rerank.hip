// In-cluster rerank (GDR stage 2) for gfx950 — replaces the host loops of the reference at
// GDR_model/main_models.py:1434-1462 (one .cuda()+cat per candidate doc), :1574-1582 (a [B,Ncand,768]
// temporary and the full B x sum(Ncand) cross product) and :1598-1637 (Python slicing per cluster and one
// topk per alpha).  Only the block diagonal is computed: each query against its own decoded clusters.
//
// One workgroup per query:
//   gather+dot  a wave per candidate row: 16-byte coalesced reads of D[id] (HBM/L2-bound gather), lane-split
//               dot product with q held in registers, wave-shuffle reduction, tanh/sigmoid  -> LDS
//   softmax     of the R length-penalised beam scores (main_models.py:1598-1601)            -> LDS
//   per alpha   key = orderable(sim + alpha*p[cluster]) << 32 | ~position ; bitonic sort in LDS ; first k
//               (ties: higher score, then earlier candidate — torch leaves tie order unspecified)
#include <math.h>

#include "common.h"

namespace gdr {

constexpr int RR_THREADS = 256;
constexpr int RR_MAX_CAND = 8192;
constexpr int RR_MAX_BEAMS = 1024;

__device__ __forceinline__ uint32_t rr_fkey(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rr_fkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(RR_THREADS) void rerank_kernel(const float* __restrict__ q, const float* __restrict__ D,
                                                            int d4, const int32_t* __restrict__ cand_offsets,
                                                            const int32_t* __restrict__ cand_ids,
                                                            const float* __restrict__ beam_scores, int R,
                                                            const float* __restrict__ alphas, int A, int k, int func,
                                                            int npad, float* __restrict__ out_val,
                                                            int32_t* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);  // [npad]
  float* sim = reinterpret_cast<float*>(keys + npad);                          // [npad]
  float* addp = sim + npad;                                                    // [npad]  p[cluster(c)]
  float* prob = addp + npad;                                                   // [R]
  __shared__ float red[8];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int base = cand_offsets[(int64_t)b * R];
  const int ncand_all = cand_offsets[(int64_t)(b + 1) * R] - base;
  const int ncand = ncand_all < npad ? ncand_all : npad;

  // ---- softmax over the R beam scores ----
  float mx = -INFINITY;
  for (int j = tid; j < R; j += RR_THREADS) mx = fmaxf(mx, beam_scores[(int64_t)b * R + j]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sm = 0.f;
  for (int j = tid; j < R; j += RR_THREADS) {
    const float e = expf(beam_scores[(int64_t)b * R + j] - mx);
    prob[j] = e;
    sm += e;
  }
  for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
  if (lane == 0) red[4 + wave] = sm;
  __syncthreads();
  sm = red[4] + red[5] + red[6] + red[7];
  for (int j = tid; j < R; j += RR_THREADS) prob[j] = prob[j] / sm;
  __syncthreads();
  // ---- which cluster segment each candidate sits in ----
  for (int j = wave; j < R; j += 4) {
    const int lo = cand_offsets[(int64_t)b * R + j] - base, hi = cand_offsets[(int64_t)b * R + j + 1] - base;
    const float p = prob[j];
    for (int c = lo + lane; c < hi && c < npad; c += 64) addp[c] = p;
  }
  // ---- gather + dot ----
  const float4* q4 = reinterpret_cast<const float4*>(q) + (int64_t)b * d4;
  for (int c = wave; c < ncand; c += 4) {
    const int32_t id = cand_ids[base + c];
    const float4* dr = reinterpret_cast<const float4*>(D) + (int64_t)id * d4;
    float acc = 0.f;
    for (int e = lane; e < d4; e += 64) {
      const float4 x = q4[e], y = dr[e];
      acc = fmaf(x.x, y.x, acc);
      acc = fmaf(x.y, y.y, acc);
      acc = fmaf(x.z, y.z, acc);
      acc = fmaf(x.w, y.w, acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) sim[c] = func == 0 ? tanhf(acc) : 1.0f / (1.0f + expf(-acc));
  }
  __syncthreads();
  // ---- one sorted top-k per alpha ----
  for (int ai = 0; ai < A; ++ai) {
    const float alpha = alphas[ai];
    for (int c = tid; c < npad; c += RR_THREADS) {
      unsigned long long key = 0ull;
      if (c < ncand) {
        const float s = __fadd_rn(sim[c], __fmul_rn(alpha, addp[c]));  // s + alpha*p, two roundings as in torch
        key = ((unsigned long long)rr_fkey(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
      }
      keys[c] = key;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (npad >> 1); t += RR_THREADS) {
          const int lo = (t / stride) * (stride << 1) + (t % stride);
          const int hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long x = keys[lo], y = keys[hi];
          if ((x < y) == desc) {
            keys[lo] = y;
            keys[hi] = x;
          }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < k; i += RR_THREADS) {
      float v = -INFINITY;
      int32_t id = -1;
      if (i < ncand) {
        const unsigned long long key = keys[i];
        v = rr_fkey_inv((uint32_t)(key >> 32));
        id = cand_ids[base + (int)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull))];
      }
      out_val[((int64_t)b * A + ai) * k + i] = v;
      out_idx[((int64_t)b * A + ai) * k + i] = id;
    }
    __syncthreads();
  }
}

}  // namespace gdr

extern "C" int gdr_rerank_topk(const float* q, const float* D, int d, const int32_t* cand_offsets,
                               const int32_t* cand_ids, const float* beam_scores, int B, int R, const float* alphas,
                               int A, int k, int func, float* out_val, int32_t* out_idx, int max_cand, void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GDR_CHECK_ARG(q && D && cand_offsets && cand_ids && beam_scores && alphas && out_val && out_idx, "rerank: null pointer");
  GDR_CHECK_ARG(B > 0 && R > 0 && R <= RR_MAX_BEAMS && A > 0 && k > 0 && d > 0 && d % 4 == 0,
                "rerank: bad shape B=%d R=%d A=%d k=%d d=%d", B, R, A, k, d);
  GDR_CHECK_ARG(func == 0 || func == 1, "rerank: func must be 0 (tanh) or 1 (sigmoid)");
  GDR_CHECK_ARG(max_cand >= 1 && max_cand <= RR_MAX_CAND, "rerank: max_cand=%d must be in [1,%d]", max_cand, RR_MAX_CAND);
  int npad = 64;
  while (npad < max_cand) npad <<= 1;
  const size_t lds = (size_t)npad * (8 + 4 + 4) + (size_t)R * 4;
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(rerank_kernel), 150 * 1024, "rerank")) return rc__;
  hipLaunchKernelGGL(rerank_kernel, dim3(B), dim3(RR_THREADS), lds, stream, q, D, d / 4, cand_offsets, cand_ids,
                     beam_scores, R, alphas, A, k, func, npad, out_val, out_idx);
  GDR_CHECK_LAUNCH("rerank_kernel");
  return GDR_OK;
}
