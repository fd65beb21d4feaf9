// In-cluster rerank (GDR stage 2) for gfx950 — replaces the host loops of the reference at
// GDR_model/main_models.py:1398-1462 (decode_token -> id_mapping lookup -> one .cuda()+cat per candidate doc),
// :1574-1582 (a [B,Ncand,768] temporary and the full B x sum(Ncand) cross product) and :1598-1637 (Python slicing per
// cluster and one topk per alpha).  Only the block diagonal is computed: each query against its own decoded clusters.
//
// Three kernels, nothing visits the host between the beam search and the ranked doc lists:
//   cluster_candidates  decoded token rows -> cluster index (exact-match hash table over the clusters' token bodies) ->
//                    per-query block of candidate doc ids, segments in beam order = the reference's order (:1441-1443)
//   rerank_dot       a wave per 4 candidate rows: 16-byte coalesced gather of D[id] (fp32 or bf16 rows; HBM/L2-bound),
//                    lane-split fmaf chain against q held in registers, butterfly sum, tanh / sigmoid -> sim[b][c].
//                    Grid = (candidate chunks, queries): one query x 1 200 candidates (infer.sh: 100 beams) spreads over
//                    75 workgroups instead of the single one the round-2 kernel gave it.
//   rerank_select    a workgroup per (alpha, query): softmax of the R length-penalised beam scores (:1598-1601),
//                    key = orderable(sim + alpha*p[cluster]) << 32 | ~position, one bitonic sort in LDS, first k
//                    (ties: higher score, then earlier candidate — torch leaves tie order unspecified).
// Sharded corpus (GDR mode): a rank passes its row block [doc_lo, doc_hi) — candidates outside are skipped —
// and asks for candidate POSITIONS; per-candidate scores do not depend on the shard, and a merge by "higher score, then
// lower position" (gdr_topk_merge_packed) of the per-shard lists reproduces the unsharded list bit for bit.
#include <math.h>

#include "common.h"

namespace gdr {

constexpr int RR_MAX_CAND = 8192;
constexpr int RR_MAX_BEAMS = 1024;
constexpr int RR_CH = 16;  // candidates per workgroup of the dot pass: 4 waves x 4 rows in flight each

__device__ __forceinline__ uint32_t rr_fkey(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rr_fkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- sim[b][c] = f(q[b] . D[cand c of query b]) ---------------------------------------------------------------------
// Candidate lists come in two layouts (include/gdr_hip.h): cand_stride == 0 — ONE CSR over all B*R segments, the
// reference's concatenation (main_models.py:1441-1443); cand_stride > 0 — a block per query: offsets [B][R+1] relative to
// the block, ids [B][cand_stride] (what gdr_cluster_candidates emits and what travels between ranks).
struct CandSeg {
  const int32_t* off;  // R + 1 segment bounds of the query; candidate c of the query is cand_ids[base + c], lo/hi - off[0]
  int64_t base;
};
__device__ __forceinline__ CandSeg cand_seg(const int32_t* cand_offsets, int b, int R, int cand_stride) {
  if (cand_stride > 0) return CandSeg{cand_offsets + (int64_t)b * (R + 1), (int64_t)b * cand_stride};
  const int32_t* off = cand_offsets + (int64_t)b * R;
  return CandSeg{off, (int64_t)off[0]};
}

template <bool BF16>
__global__ __launch_bounds__(256) void rerank_dot_kernel(const float* __restrict__ q, const void* __restrict__ D, int d4,
                                                         const int32_t* __restrict__ cand_offsets,
                                                         const int32_t* __restrict__ cand_ids, int R, int func,
                                                         int32_t doc_lo, int32_t doc_hi, int max_cand, int cand_stride,
                                                         float* __restrict__ sim) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const CandSeg cs = cand_seg(cand_offsets, b, R, cand_stride);
  const int64_t base = cs.base;
  int ncand = cs.off[R] - cs.off[0];
  ncand = ncand < max_cand ? ncand : max_cand;
  const int c0 = blockIdx.x * RR_CH + wave * 4;
  if (c0 >= ncand) return;  // wave-uniform
  int64_t row[4];
  bool on[4];
  int64_t any_row = -1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + i;
    const int32_t id = c < ncand ? cand_ids[base + c] : -1;
    on[i] = c < ncand && id >= doc_lo && id < doc_hi;
    row[i] = (int64_t)id - doc_lo;
    if (on[i]) any_row = row[i];
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (any_row >= 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (!on[i]) row[i] = any_row;  // rows outside the shard read a valid row (uniform control flow), masked below
    const float4* q4 = reinterpret_cast<const float4*>(q) + (int64_t)b * d4;
    for (int e = lane; e < d4; e += 64) {
      const float4 x = q4[e];
      float4 y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (BF16) {  // 4 bf16 = 8 bytes per lane; widening to fp32 is exact
          const uint2 u = reinterpret_cast<const uint2*>(D)[row[i] * d4 + e];
          y[i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                             __uint_as_float(u.y & 0xFFFF0000u));
        } else {
          y[i] = reinterpret_cast<const float4*>(D)[row[i] * d4 + e];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = fmaf(x.x, y[i].x, acc[i]);
        acc[i] = fmaf(x.y, y[i].y, acc[i]);
        acc[i] = fmaf(x.z, y[i].z, acc[i]);
        acc[i] = fmaf(x.w, y[i].w, acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = acc[i];
    a = wave_sum(a);
    if (lane == 0 && c0 + i < ncand)
      sim[(int64_t)b * max_cand + c0 + i] = !on[i] ? -INFINITY : (func == 0 ? tanhf(a) : 1.0f / (1.0f + expf(-a)));
  }
}

// Bitonic sort of npad (power of two) 64-bit keys in LDS, descending.  A wave owns a block of npad / nwaves consecutive
// keys: every stage whose compare-exchange pairs stay inside a block needs no workgroup barrier (LDS operations of one
// wave execute in order), so of the 66 stages of a 2048-key sort on 16 waves only the ones with stride >= 64 cost one.
__device__ __forceinline__ void rr_bitonic_desc(unsigned long long* keys, int npad) {
  const int tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
  const int epw = npad / nwaves;
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (2 * stride <= epw) {
        for (int p = lane; p < (epw >> 1); p += 64) {
          const int t = (epw >> 1) * wave + p;
          const int lo = (t / stride) * (stride << 1) + (t % stride), hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long x = keys[lo], y = keys[hi];
          if ((x < y) == desc) keys[lo] = y, keys[hi] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else {
        __syncthreads();
        for (int t = tid; t < (npad >> 1); t += nthr) {
          const int lo = (t / stride) * (stride << 1) + (t % stride), hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long x = keys[lo], y = keys[hi];
          if ((x < y) == desc) keys[lo] = y, keys[hi] = x;
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
}

// ---- per (alpha, query): sorted top-k of sim + alpha * softmax(beam scores)[cluster of the candidate] ---------------
__global__ __launch_bounds__(1024) void rerank_select_kernel(const float* __restrict__ sim,
                                                             const int32_t* __restrict__ cand_offsets,
                                                             const int32_t* __restrict__ cand_ids,
                                                             const float* __restrict__ beam_scores, int R,
                                                             const float* __restrict__ alphas, int A, int k, int npad,
                                                             int max_cand, int cand_stride, int positions,
                                                             float* __restrict__ out_val, int32_t* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);  // [npad]
  float* prob = reinterpret_cast<float*>(keys + npad);                         // [R]
  const int ai = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
  const CandSeg cs = cand_seg(cand_offsets, b, R, cand_stride);
  const int64_t base = cs.base;
  int ncand = cs.off[R] - cs.off[0];
  ncand = ncand < npad ? ncand : npad;
  ncand = ncand < max_cand ? ncand : max_cand;
  for (int c = tid; c < npad; c += nthr) keys[c] = 0ull;
  // softmax over the R beam scores (main_models.py:1598-1601), by the first wave: the summation order does not depend
  // on the workgroup size
  if (wave == 0) {
    const float* bs = beam_scores + (int64_t)b * R;
    float mx = -INFINITY;
    for (int j = lane; j < R; j += 64) mx = fmaxf(mx, bs[j]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int j = lane; j < R; j += 64) {
      const float e = expf(bs[j] - mx);
      prob[j] = e;
      sm += e;
    }
    sm = wave_sum(sm);
    for (int j = lane; j < R; j += 64) prob[j] = prob[j] / sm;
  }
  __syncthreads();
  const float alpha = alphas[ai];
  const float* sb = sim + (int64_t)b * max_cand;
  for (int j = wave; j < R; j += nwaves) {  // a wave per cluster segment
    const int lo = cs.off[j] - cs.off[0], hi = cs.off[j + 1] - cs.off[0];
    const float ap = __fmul_rn(alpha, prob[j]);
    for (int c = lo + lane; c < hi && c < ncand; c += 64) {
      const float s0 = sb[c];
      if (s0 > -INFINITY)  // -inf marks a candidate outside this rank's shard
        keys[c] = ((unsigned long long)rr_fkey(__fadd_rn(s0, ap)) << 32) |  // s + alpha*p: two roundings as in torch
                  (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
    }
  }
  __syncthreads();
  rr_bitonic_desc(keys, npad);
  for (int i = tid; i < k; i += nthr) {
    float v = -INFINITY;
    int32_t id = -1;
    const unsigned long long key = i < npad ? keys[i] : 0ull;
    if (key != 0ull) {
      v = rr_fkey_inv((uint32_t)(key >> 32));
      const int c = (int)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
      id = positions ? c : cand_ids[base + c];
    }
    out_val[((int64_t)b * A + ai) * k + i] = v;
    out_idx[((int64_t)b * A + ai) * k + i] = id;
  }
}

static int rerank_impl(const float* q, const void* D, bool bf16, int d, const int32_t* cand_offsets, const int32_t* cand_ids,
                       const float* beam_scores, int B, int R, const float* alphas, int A, int k, int func, float* out_val,
                       int32_t* out_idx, int max_cand, int cand_stride, int32_t doc_lo, int32_t doc_hi, int flags,
                       void* workspace, size_t workspace_bytes, hipStream_t stream) {
  GDR_CHECK_ARG(q && cand_offsets && cand_ids && beam_scores && alphas && out_val && out_idx && workspace, "rerank: null pointer");
  GDR_CHECK_ARG(D || doc_hi <= doc_lo, "rerank: null corpus pointer for a non-empty row range");
  GDR_CHECK_ARG(B > 0 && R > 0 && R <= RR_MAX_BEAMS && A > 0 && k > 0 && d > 0 && d % 4 == 0,
                "rerank: bad shape B=%d R=%d A=%d k=%d d=%d", B, R, A, k, d);
  GDR_CHECK_ARG(func == 0 || func == 1, "rerank: func must be 0 (tanh) or 1 (sigmoid)");
  GDR_CHECK_ARG(max_cand >= 1 && max_cand <= RR_MAX_CAND, "rerank: max_cand=%d must be in [1,%d]", max_cand, RR_MAX_CAND);
  GDR_CHECK_ARG((flags & ~GDR_RERANK_POSITIONS) == 0, "rerank: unknown flags %d", flags);
  GDR_CHECK_ARG(cand_stride == 0 || cand_stride >= max_cand, "rerank: cand_stride=%d must be 0 (one CSR) or >= max_cand=%d",
                cand_stride, max_cand);
  const size_t need = gdr_rerank_workspace_bytes(B, max_cand);
  if (workspace_bytes < need) {
    set_error("rerank: workspace %zu < required %zu", workspace_bytes, need);
    return GDR_ENOSPC;
  }
  float* sim = static_cast<float*>(workspace);
  int npad = 64;
  while (npad < max_cand) npad <<= 1;
  const double bytes = (double)B * max_cand * d * (bf16 ? 2 : 4);  // upper bound: the live candidate count lives on the device
  {
    ProfScope prof(PROF_RERANK, bytes, stream);
    const dim3 grid((unsigned)((max_cand + RR_CH - 1) / RR_CH), (unsigned)B);
    if (bf16)
      hipLaunchKernelGGL(rerank_dot_kernel<true>, grid, dim3(256), 0, stream, q, D, d / 4, cand_offsets, cand_ids, R, func, doc_lo,
                         doc_hi, max_cand, cand_stride, sim);
    else
      hipLaunchKernelGGL(rerank_dot_kernel<false>, grid, dim3(256), 0, stream, q, D, d / 4, cand_offsets, cand_ids, R, func, doc_lo,
                         doc_hi, max_cand, cand_stride, sim);
    GDR_CHECK_LAUNCH("rerank_dot_kernel");
  }
  const size_t lds = (size_t)npad * 8 + (size_t)R * 4;
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(rerank_select_kernel), 80 * 1024, "rerank")) return rc__;
  ProfScope prof(PROF_SELECT, 0.0, stream);
  hipLaunchKernelGGL(rerank_select_kernel, dim3((unsigned)A, (unsigned)B), dim3(npad >= 2048 ? 1024 : 256), lds, stream, sim,
                     cand_offsets, cand_ids, beam_scores, R, alphas, A, k, npad, max_cand, cand_stride,
                     (flags & GDR_RERANK_POSITIONS) ? 1 : 0, out_val, out_idx);
  GDR_CHECK_LAUNCH("rerank_select_kernel");
  return GDR_OK;
}

// ---- decoded token rows -> cluster -> candidate CSR (main_models.py:1398, 1441-1443) ---------------------------------
// A row of generate()'s output is START, tokens..., [EOS], PAD...; decode_token (main_models.py:322-346) drops START and
// cuts at the first EOS — and decodes a row WITHOUT any EOS whole, START included.  The digit string it prints is a
// one-to-one image of that token body (given V and --position), so "id_mapping[string]" is an exact-match lookup of the
// body among the clusters' bodies: open addressing over a 64-bit hash, verified against the stored body.
__host__ __device__ __forceinline__ uint64_t ci_hash_step(uint64_t h, int32_t t) {
  h = (h ^ (uint64_t)(uint32_t)t) * 0x100000001B3ull;
  return h ^ (h >> 29);
}

// One workgroup per query, a thread per beam row (R <= 1024): lookup, scan of the cluster sizes over the query's R rows,
// then all threads copy the members into the query's block.
__global__ __launch_bounds__(256) void cluster_candidates_kernel(GdrClusterIndex ci, const int64_t* __restrict__ ids, int R,
                                                                 int max_length, int cand_stride,
                                                                 int32_t* __restrict__ cluster_of,
                                                                 int32_t* __restrict__ cand_offsets,
                                                                 int32_t* __restrict__ cand_ids) {
  extern __shared__ int32_t csm[];  // [R] cluster of each row, [R + 1] offsets
  int32_t* cl_s = csm;
  int32_t* off_s = csm + R;
  __shared__ int wsum[4];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) carry = 0, off_s[0] = 0;
  __syncthreads();
  for (int j0 = 0; j0 < R; j0 += 256) {
    const int j = j0 + tid;
    int cl = -1, cnt = 0;
    if (j < R) {
      const int64_t* row = ids + ((int64_t)b * R + j) * max_length;
      int eos = -1;
      for (int t = 0; t < max_length; ++t)
        if (row[t] == 1) {
          eos = t;
          break;
        }
      const int t0 = eos >= 0 ? 1 : 0, t1 = eos >= 0 ? eos : max_length;  // `lst[1:lst.index(1)] if 1 in lst else lst`
      const int len = t1 - t0;
      if (len <= ci.key_len) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)len;
        for (int t = t0; t < t1; ++t) h = ci_hash_step(h, (int32_t)row[t]);
        uint32_t slot = (uint32_t)h & (uint32_t)(ci.table_size - 1);
        for (int probe = 0; probe < ci.table_size; ++probe) {
          const int32_t c = ci.slots[slot];
          if (c < 0) break;
          const int32_t* key = ci.keys + (int64_t)c * ci.key_len;
          bool same = ci.key_lens[c] == len;
          for (int t = 0; same && t < len; ++t) same = key[t] == (int32_t)row[t0 + t];
          if (same) {
            cl = c;
            break;
          }
          slot = (slot + 1) & (uint32_t)(ci.table_size - 1);
        }
      }
      if (cl >= 0) cnt = ci.offsets[cl + 1] - ci.offsets[cl];
      cl_s[j] = cl;
      cluster_of[(int64_t)b * R + j] = cl;
    }
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (j < R) off_s[j + 1] = before + inc;
    __syncthreads();
    if (tid == 255) carry = before + inc;
    __syncthreads();
  }
  for (int j = tid; j <= R; j += 256) cand_offsets[(int64_t)b * (R + 1) + j] = off_s[j];
  for (int j = wave; j < R; j += 4) {  // a wave per segment
    const int cl = cl_s[j];
    if (cl < 0) continue;
    const int src = ci.offsets[cl], n = ci.offsets[cl + 1] - src, dst = off_s[j];
    for (int i = lane; i < n && dst + i < cand_stride; i += 64) cand_ids[(int64_t)b * cand_stride + dst + i] = ci.members[src + i];
  }
}

// ---- the sharded GDR mode's exchange row (gdr_hip.h gdr_rerank_wire_pack): {q | beam scores | offsets | ids} as int32 ----
template <bool PACK>
__global__ __launch_bounds__(256) void rerank_wire_kernel(int32_t* __restrict__ wire, int32_t* __restrict__ q,
                                                          int32_t* __restrict__ beam, int32_t* __restrict__ offs,
                                                          int32_t* __restrict__ ids, int d, int R, int stride) {
  const int b = blockIdx.x, W = d + 2 * R + 1 + stride;
  int32_t* row = wire + (int64_t)b * W;
  for (int i = threadIdx.x; i < W; i += 256) {
    int32_t* p;
    if (i < d) p = q + (int64_t)b * d + i;
    else if (i < d + R) p = beam + (int64_t)b * R + (i - d);
    else if (i < d + 2 * R + 1) p = offs + (int64_t)b * (R + 1) + (i - d - R);
    else p = ids + (int64_t)b * stride + (i - d - 2 * R - 1);
    if (PACK) row[i] = *p;
    else *p = row[i];
  }
}

__global__ __launch_bounds__(256) void rerank_pos_to_id_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ ids,
                                                               int64_t n, int per_query, int stride, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t p = pos[i];
  out[i] = (p >= 0 && p < stride) ? ids[(i / per_query) * stride + p] : -1;
}

}  // namespace gdr

extern "C" int gdr_rerank_wire_pack(const float* q, const float* beam_scores, const int32_t* cand_offsets,
                                    const int32_t* cand_ids, int B, int d, int R, int cand_stride, int32_t* wire,
                                    void* stream_) {
  using namespace gdr;
  GDR_CHECK_ARG(q && beam_scores && cand_offsets && cand_ids && wire, "rerank_wire_pack: null pointer");
  GDR_CHECK_ARG(B > 0 && d > 0 && R > 0 && cand_stride > 0, "rerank_wire_pack: bad shape B=%d d=%d R=%d stride=%d", B, d, R, cand_stride);
  hipLaunchKernelGGL(rerank_wire_kernel<true>, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream_), wire,
                     reinterpret_cast<int32_t*>(const_cast<float*>(q)), reinterpret_cast<int32_t*>(const_cast<float*>(beam_scores)),
                     const_cast<int32_t*>(cand_offsets), const_cast<int32_t*>(cand_ids), d, R, cand_stride);
  GDR_CHECK_LAUNCH("rerank_wire_kernel<pack>");
  return GDR_OK;
}

extern "C" int gdr_rerank_wire_unpack(const int32_t* wire, int B, int d, int R, int cand_stride, float* q, float* beam_scores,
                                      int32_t* cand_offsets, int32_t* cand_ids, void* stream_) {
  using namespace gdr;
  GDR_CHECK_ARG(q && beam_scores && cand_offsets && cand_ids && wire, "rerank_wire_unpack: null pointer");
  GDR_CHECK_ARG(B > 0 && d > 0 && R > 0 && cand_stride > 0, "rerank_wire_unpack: bad shape B=%d d=%d R=%d stride=%d", B, d, R, cand_stride);
  hipLaunchKernelGGL(rerank_wire_kernel<false>, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     const_cast<int32_t*>(wire), reinterpret_cast<int32_t*>(q), reinterpret_cast<int32_t*>(beam_scores), cand_offsets,
                     cand_ids, d, R, cand_stride);
  GDR_CHECK_LAUNCH("rerank_wire_kernel<unpack>");
  return GDR_OK;
}

extern "C" int gdr_rerank_positions_to_ids(const int32_t* pos, const int32_t* cand_ids, int B, int per_query, int cand_stride,
                                           int32_t* out_ids, void* stream_) {
  using namespace gdr;
  GDR_CHECK_ARG(pos && cand_ids && out_ids, "rerank_positions_to_ids: null pointer");
  GDR_CHECK_ARG(B > 0 && per_query > 0 && cand_stride > 0, "rerank_positions_to_ids: bad shape B=%d per_query=%d stride=%d", B,
                per_query, cand_stride);
  const int64_t n = (int64_t)B * per_query;
  hipLaunchKernelGGL(rerank_pos_to_id_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), pos,
                     cand_ids, n, per_query, cand_stride, out_ids);
  GDR_CHECK_LAUNCH("rerank_pos_to_id_kernel");
  return GDR_OK;
}

extern "C" size_t gdr_rerank_workspace_bytes(int B, int max_cand) {
  if (B <= 0 || max_cand <= 0) return 0;
  return gdr::align_up((size_t)B * (size_t)max_cand * sizeof(float), 256);
}

extern "C" int gdr_rerank_topk(const float* q, const float* D, int d, const int32_t* cand_offsets,
                               const int32_t* cand_ids, const float* beam_scores, int B, int R, const float* alphas,
                               int A, int k, int func, float* out_val, int32_t* out_idx, int max_cand, int cand_stride,
                               int32_t doc_lo, int32_t doc_hi, int flags, void* workspace, size_t workspace_bytes,
                               void* stream_) {
  return gdr::rerank_impl(q, D, false, d, cand_offsets, cand_ids, beam_scores, B, R, alphas, A, k, func, out_val, out_idx,
                          max_cand, cand_stride, doc_lo, doc_hi, flags, workspace, workspace_bytes,
                          static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_rerank_topk_bf16(const float* q, const void* D_bf16, int d, const int32_t* cand_offsets,
                                    const int32_t* cand_ids, const float* beam_scores, int B, int R, const float* alphas,
                                    int A, int k, int func, float* out_val, int32_t* out_idx, int max_cand,
                                    int cand_stride, int32_t doc_lo, int32_t doc_hi, int flags, void* workspace,
                                    size_t workspace_bytes, void* stream_) {
  return gdr::rerank_impl(q, D_bf16, true, d, cand_offsets, cand_ids, beam_scores, B, R, alphas, A, k, func, out_val, out_idx,
                          max_cand, cand_stride, doc_lo, doc_hi, flags, workspace, workspace_bytes,
                          static_cast<hipStream_t>(stream_));
}

extern "C" uint64_t gdr_cluster_key_hash(const int32_t* tokens, int len) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)len;
  for (int t = 0; t < len; ++t) h = gdr::ci_hash_step(h, tokens[t]);
  return h;
}

extern "C" int gdr_cluster_candidates(const GdrClusterIndex* ci, const int64_t* out_ids, int B, int R, int max_length,
                                      int32_t* cluster_of, int32_t* cand_offsets, int32_t* cand_ids, int cand_stride,
                                      void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GDR_CHECK_ARG(ci && out_ids && cluster_of && cand_offsets && cand_ids, "cluster_candidates: null pointer");
  GDR_CHECK_ARG(ci->slots && ci->keys && ci->key_lens && ci->offsets && ci->members, "cluster_candidates: null index array");
  GDR_CHECK_ARG(ci->table_size > 0 && (ci->table_size & (ci->table_size - 1)) == 0 && ci->table_size > ci->n_clusters,
                "cluster_candidates: table_size=%d must be a power of two above n_clusters=%d", ci->table_size, ci->n_clusters);
  GDR_CHECK_ARG(B > 0 && R > 0 && R <= RR_MAX_BEAMS && max_length >= 2 && ci->key_len >= 1 && cand_stride >= 1,
                "cluster_candidates: bad shape B=%d R=%d max_length=%d cand_stride=%d", B, R, max_length, cand_stride);
  hipLaunchKernelGGL(cluster_candidates_kernel, dim3((unsigned)B), dim3(256), (size_t)(2 * R + 1) * sizeof(int32_t), stream, *ci,
                     out_ids, R, max_length, cand_stride, cluster_of, cand_offsets, cand_ids);
  GDR_CHECK_LAUNCH("cluster_candidates_kernel");
  return GDR_OK;
}
