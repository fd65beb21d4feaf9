// Docid beam decode on gfx950 — device-resident replacement of the reference's generate() hot loop
// (GDR_model/transformers/generation_utils.py:629-921 + BeamHypotheses :1052-1099, driving
//  T5ForConditionalGeneration.forward's decode branch, transformers/modeling_t5.py:1529-1646).
//
// What changes relative to the reference (arithmetic per produced number is the same):
//   * use_cache=False recomputes the whole prefix every step; here every layer's K/V rows are written once per
//     position into a [Tmax][rows][3*width] cache and beams reach their ancestors' rows through an int32 table
//     (`kv_rows`) that the beam-update kernel rewrites each step — no cache reordering copies, no recompute.
//   * cross-attention K/V are projected once per QUERY (beams share them: kv_group = num_beams) instead of once
//     per beam row per step (generation_utils.py:459-461 expands the encoder states to B*beams rows).
//   * the adaptor's cross-attention over a single learned key (modeling_t5.py:1628-1631) is the constant
//     out_proj(v_proj(adaptor_embeddings)) — softmax over one key is exactly 1 — precomputed at load time.
//   * the adaptor_linear head (modeling_t5.py:1634-1639: [R*t,768]x[768,768*302] + a [R,t,768,302] broadcast add)
//     is evaluated for the last position and its V+1 unmasked columns only: one GEMM against a re-laid weight
//     slice [V+1][d][d] and a fused dot kernel; masked columns are exactly -1e9 either way.
//   * beam bookkeeping (top-2R, EOS handling, BeamHypotheses.add/is_done, finalisation) runs in three small
//     kernels per step with no host round trip; the reference syncs per candidate through .item().
//   * the adaptor chain and the head matrix W = adaptor_linear(adaptor(prefix)) + lm_head depend on the TOKEN PREFIX
//     only, never on the query (modeling_t5.py:1618-1639): with a GdrPrefixTable — built once from the weights and the
//     corpus' docid trie (gdr_t5_prefix_table_build) — a row whose prefix is a trie node reads its head matrix (and,
//     for its descendants, its adaptor K/V) from the table; only rows whose prefix left the trie are compacted and run
//     through the adaptor + head GEMM (device-side row count, no host round trip).
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "layers.h"
#include "decode_fused.h"

namespace gdr {

constexpr int PAD_ID = 0, EOS_ID = 1, START_ID = 0;
constexpr int MAXLEN_CAP = 32;
constexpr size_t SPLITK_WS_BYTES = (size_t)48 << 20;  // <= 640 partial 128x128 tiles + slack
static_assert(STREAMK_BYTES <= SPLITK_WS_BYTES, "stream-K scratch must fit the split-K region");

struct BeamBufs {
  int32_t* seq[2];      // [rows][maxlen] token history, double buffered
  float* beam_scores;   // [rows]
  int64_t* cur_tok;     // [rows] last token of each row (embedding input of the step)
  int32_t* parent;      // [rows] row of the previous step that each row extends
  int32_t* anc[2];      // [rows][maxlen] row whose cache slot holds position p of this row's prefix
  int32_t* kv_rows;     // [rows][s+1] absolute cache row = p*rows + anc
  int32_t* node[2];     // [rows] trie node reached by the row's prefix (-1: left the tree), double buffered
  int32_t* miss_rows;   // [rows] prefix-table mode: rows whose prefix has no table entry, compacted (ascending)
  int32_t* miss_index;  // [rows] inverse: position of a row in miss_rows, or -1 when its prefix hit the table
  int32_t* kv_rows_c;   // [rows][maxlen] kv_rows of the compacted rows
  int64_t* n_miss;      // [1] number of compacted rows (read by the kernels of the adaptor chain)
  float* logits;        // [rows][V+1] unmasked-column logits of the step
  float* cand_score;    // [B][2R]
  int32_t* cand_idx;    // [B][2R]  beam*Vd + token
  double* hyp_score;    // [B][R+1]
  int32_t* hyp_len;     // [B][R+1]
  int32_t* hyp_tok;     // [B][R+1][maxlen]
  int32_t* hyp_seq;     // [B][R+1] insertion number of each live hypothesis (list order of the reference's self.beams)
  int32_t* hyp_next;    // [B] next insertion number
  int32_t* hyp_cnt;     // [B]
  double* hyp_worst;    // [B]
  int32_t* done;        // [B]
  int32_t* n_done;      // [1] queries whose beam search is done (generation_utils.py:827-829)
  int32_t* live;        // [1] 1 until n_done reaches B: the linears and the miss-row chain of later steps look at it
  const int32_t* live_gate;  // = live when the call may skip the steps behind the last query's end (no per-step trace wanted), else
                             // null: the attention / reduction / head / beam kernels of such a step exit at once through it
  int32_t* all_done_host;  // host-mapped word (may be null): receives `done_epoch` when n_done reaches B
  int32_t done_epoch;
};

struct BeamDims {
  int B, R, V, Vd, maxlen, nret;
  double lp;
  // docid trie: child[node*V + c] = next node or -1 for the digit c of the node's depth.  trie_child alone = rows only
  // TRACK the node their prefix reaches (prefix-table mode); with trie_eos (eos_ok[node] = 1 if EOS is a child) the
  // trie also CONSTRAINS the beams (generation_utils_previous.py:714-729).  Both null = the shipped behaviour.
  const int32_t* trie_child;
  const int32_t* trie_eos;
  int trie_nodes;
};

static size_t carve(size_t& o, size_t bytes) {
  const size_t at = o;
  o += align_up(bytes, 256);
  return at;
}

static size_t beam_layout(const BeamDims& bd, char* base, BeamBufs* bb) {
  const size_t rows = (size_t)bd.B * bd.R, ml = bd.maxlen, V1 = bd.V + 1;
  size_t o = 0;
#define CARVE(field, type, count)                                       \
  do {                                                                  \
    const size_t at__ = carve(o, sizeof(type) * (count));               \
    if (bb) bb->field = reinterpret_cast<type*>(base + at__);           \
  } while (0)
  CARVE(seq[0], int32_t, rows * ml);
  CARVE(seq[1], int32_t, rows * ml);
  CARVE(beam_scores, float, rows);
  CARVE(cur_tok, int64_t, rows);
  CARVE(parent, int32_t, rows);
  CARVE(anc[0], int32_t, rows * ml);
  CARVE(anc[1], int32_t, rows * ml);
  CARVE(kv_rows, int32_t, rows * ml);
  CARVE(node[0], int32_t, rows);
  CARVE(node[1], int32_t, rows);
  CARVE(miss_rows, int32_t, rows);
  CARVE(miss_index, int32_t, rows);
  CARVE(kv_rows_c, int32_t, rows * ml);
  CARVE(n_miss, int64_t, 1);
  CARVE(logits, float, rows * V1);
  CARVE(cand_score, float, (size_t)bd.B * 2 * bd.R);
  CARVE(cand_idx, int32_t, (size_t)bd.B * 2 * bd.R);
  CARVE(hyp_score, double, (size_t)bd.B * (bd.R + 1));
  CARVE(hyp_len, int32_t, (size_t)bd.B * (bd.R + 1));
  CARVE(hyp_tok, int32_t, (size_t)bd.B * (bd.R + 1) * ml);
  CARVE(hyp_seq, int32_t, (size_t)bd.B * (bd.R + 1));
  CARVE(hyp_next, int32_t, bd.B);
  CARVE(hyp_cnt, int32_t, bd.B);
  CARVE(hyp_worst, double, bd.B);
  CARVE(done, int32_t, bd.B);
  CARVE(n_done, int32_t, 1);
  CARVE(live, int32_t, 1);
#undef CARVE
  return o;
}

// ------------------------------------------------------------------------------------------ kernels
__global__ void beam_init_kernel(BeamBufs bb, BeamDims bd, int dedup0) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = bd.B * bd.R;
  if (r < rows) {
    bb.seq[0][(size_t)r * bd.maxlen] = START_ID;
    bb.beam_scores[r] = (r % bd.R == 0) ? 0.f : -1e9f;  // generation_utils.py:663-668
    bb.cur_tok[r] = START_ID;
    bb.parent[r] = r;
    // dedup0: step 0 computes ONE row per query (identical beam rows), filed as row b of the position-0 cache slots
    bb.anc[0][(size_t)r * bd.maxlen] = dedup0 ? r / bd.R : r;
    bb.kv_rows[r] = r;  // step 0: stride 1, position 0 (dedup0: the first B entries serve the B computed rows)
    bb.node[0][r] = 0;  // trie root
  }
  if (r < bd.B) {
    bb.hyp_cnt[r] = 0;
    bb.hyp_next[r] = 0;
    bb.hyp_worst[r] = 1e9;  // :1062
    bb.done[r] = 0;
  }
  if (r == 0) *bb.n_done = 0, *bb.live = 1;
}

__device__ __forceinline__ uint32_t dfkey(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dfkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// logits[r][c] = sum_i (h[r][i] * d^-0.5) * (A[r][c*d + i] + E[c][i])      (modeling_t5.py:1575-1576,1637-1639)
__global__ __launch_bounds__(256) void head_logits_kernel(const float* __restrict__ h, const float* __restrict__ A,
                                                          const float* __restrict__ E, int rows, int V1, int d,
                                                          float scale, float* __restrict__ logits,
                                                          const int32_t* __restrict__ live) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)rows * V1) return;
  if (live && *live == 0) return;  // every query is done: these logits are never ranked
  const int r = (int)(item / V1), c = (int)(item % V1), lane = threadIdx.x & 63;
  const float4* h4 = reinterpret_cast<const float4*>(h + (size_t)r * d);
  const float4* a4 = reinterpret_cast<const float4*>(A + ((size_t)r * V1 + c) * d);
  const float4* e4 = reinterpret_cast<const float4*>(E + (size_t)c * d);
  float acc = 0.f;
  for (int i = lane; i < d / 4; i += 64) {
    const float4 hv = h4[i], av = a4[i], ev = e4[i];
    acc = fmaf(hv.x * scale, av.x + ev.x, acc);
    acc = fmaf(hv.y * scale, av.y + ev.y, acc);
    acc = fmaf(hv.z * scale, av.z + ev.z, acc);
    acc = fmaf(hv.w * scale, av.w + ev.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) logits[item] = acc;
}

// Prefix-table form of the same dot: a row whose prefix is a table node reads its finished head matrix
// W[node][c][i] = A + E (built at load); any other row reads the A computed this step for its compacted slot.
__global__ __launch_bounds__(256) void head_logits_table_kernel(const float* __restrict__ h, const float* __restrict__ A_c,
                                                                const float* __restrict__ E, const float* __restrict__ tabW,
                                                                const int32_t* __restrict__ node,
                                                                const int32_t* __restrict__ miss_index, int n_table,
                                                                int rows, int V1, int d, float scale,
                                                                float* __restrict__ logits, const int32_t* __restrict__ live,
                                                                const float* __restrict__ partial = nullptr) {
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)rows * V1) return;
  if (live && *live == 0) return;  // every query is done: these logits are never ranked
  const int r = (int)(item / V1), c = (int)(item % V1), lane = threadIdx.x & 63;
  const float4* h4 = reinterpret_cast<const float4*>(h + (size_t)r * d);
  const int mi = miss_index[r];
  float acc = 0.f;
  if (mi < 0) {
    const float4* w4 = reinterpret_cast<const float4*>(tabW + ((size_t)node[r] * V1 + c) * d);
    for (int i = lane; i < d / 4; i += 64) {
      const float4 hv = h4[i], wv = w4[i];
      acc = fmaf(hv.x * scale, wv.x, acc);
      acc = fmaf(hv.y * scale, wv.y, acc);
      acc = fmaf(hv.z * scale, wv.z, acc);
      acc = fmaf(hv.w * scale, wv.w, acc);
    }
  } else if (partial) {
    // the head GEMM's epilogue already took the dot (launch_linear_bf16_headdot): d / 64 partial sums per (row, c), added in a fixed tree
    const int slots = d >> 6;
    if (lane < slots) acc = partial[((size_t)mi * V1 + c) * slots + lane];
  } else {
    const float4* a4 = reinterpret_cast<const float4*>(A_c + ((size_t)mi * V1 + c) * d);
    const float4* e4 = reinterpret_cast<const float4*>(E + (size_t)c * d);
    for (int i = lane; i < d / 4; i += 64) {
      const float4 hv = h4[i], av = a4[i], ev = e4[i];
      acc = fmaf(hv.x * scale, av.x + ev.x, acc);
      acc = fmaf(hv.y * scale, av.y + ev.y, acc);
      acc = fmaf(hv.z * scale, av.z + ev.z, acc);
      acc = fmaf(hv.w * scale, av.w + ev.w, acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) logits[item] = acc;
  (void)n_table;
}

// Prefix-table mode, once per step (one workgroup): which rows have a table entry for their prefix?  The others are
// compacted in ascending row order (deterministic) together with their ancestor rows; *n_miss is what every kernel of
// the adaptor chain reads as its row count.
__global__ __launch_bounds__(1024) void prefix_plan_kernel(BeamBufs bb, int rows, int stride, int cur, int n_table) {
  // A thread owns ceil(rows / 1024) CONSECUTIVE rows: it counts its misses, one scan over the 1024 counts gives every thread the
  // number of misses before its first row, and it numbers its own misses from there — ascending row order, one pass.  (Through
  // round 3 the workgroup walked the rows 1 024 at a time with three barriers per chunk: 470 us at 20 480 rows, at the head of
  // the adaptor chain of every step.)
  __shared__ int wsum[16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per = (rows + 1023) >> 10;
  const int r0 = tid * per, r1 = min(rows, r0 + per);
  int cnt = 0;
  for (int r = r0; r < r1; ++r) {
    const int nd = bb.node[cur][r];
    cnt += (nd >= 0 && nd < n_table) ? 0 : 1;
  }
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) before += wsum[w];
    total += wsum[w];
  }
  int pos = before + inc - cnt;
  for (int r = r0; r < r1; ++r) {
    const int nd = bb.node[cur][r];
    const bool miss = !(nd >= 0 && nd < n_table);
    bb.miss_index[r] = miss ? pos : -1;
    if (miss) bb.miss_rows[pos++] = r;
  }
  // every query done (beam_update_kernel): the miss rows' adaptor chain and head GEMM of this step run on zero rows; the
  // index arrays stay as computed, so whoever still reads them (the head lookup of a step nobody consumes) stays in bounds
  if (tid == 0) *bb.n_miss = *bb.live ? total : 0;
  (void)stride;
}

// kv_rows_c[i][p] = kv_rows[miss_rows[i]][p]; hit rows: the table's (q,k,v) of the node -> this step's slot of every
// adaptor layer's cache, so that a descendant that later leaves the trie finds its ancestors' K/V where it looks.
__global__ __launch_bounds__(256) void prefix_fill_kernel(BeamBufs bb, int rows, int stride, int cur,
                                                          const float* __restrict__ tab_kv, int64_t tab_layer_stride,
                                                          float* __restrict__ acache_slot, int64_t cache_layer_stride,
                                                          int n_layers, int d3) {
  const int r = blockIdx.x;
  if (bb.live_gate && *bb.live_gate == 0) return;  // every query is done: nobody will look for ancestors in this step's slots
  const int mi = bb.miss_index[r];
  if (mi >= 0) {
    for (int p = threadIdx.x; p < stride; p += 256) bb.kv_rows_c[(size_t)mi * stride + p] = bb.kv_rows[(size_t)r * stride + p];
    return;
  }
  const int nd = bb.node[cur][r];
  const int n4 = d3 >> 2;
  for (int l = 0; l < n_layers; ++l) {
    const float4* src = reinterpret_cast<const float4*>(tab_kv + l * tab_layer_stride + (size_t)nd * d3);
    float4* dst = reinterpret_cast<float4*>(acache_slot + l * cache_layer_stride + (size_t)r * d3);
    for (int c = threadIdx.x; c < n4; c += 256) dst[c] = src[c];
  }
}

// xd[r] = table[tok[r]] and nx[r] = T5LayerNorm(xd[r]; w) in one launch: the embedding of a decode step and the first norm
// of the decoder stack (modeling_t5.py:725, :164-171), arithmetic of rmsnorm_kernel (layers.hip).  One wave per row.
__global__ __launch_bounds__(256) void embed_rmsnorm_kernel(const float* __restrict__ table, const int64_t* __restrict__ tok,
                                                            int rows, int d4, int vocab, const float* __restrict__ w, float eps,
                                                            float* __restrict__ xd, float* __restrict__ nx,
                                                            const int32_t* __restrict__ live) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (live && *live == 0) return;  // every query is done (beam_update_kernel): this step's rows are never read
  const int lane = threadIdx.x & 63;
  int64_t id = tok[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* src = reinterpret_cast<const float4*>(table) + id * d4;
  float4* xr = reinterpret_cast<float4*>(xd) + (int64_t)row * d4;
  float ss = 0.f;
  for (int c = lane; c < d4; c += 64) {
    const float4 v = src[c];
    xr[c] = v;
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const RowDivisor over(sqrtf(ss / (float)(d4 * 4) + eps));  // x / denom, bit for bit (common.h)
  float4* yr = reinterpret_cast<float4*>(nx) + (int64_t)row * d4;
  const float4* wr = reinterpret_cast<const float4*>(w);
  for (int c = lane; c < d4; c += 64) {
    const float4 v = src[c], g = wr[c];
    yr[c] = make_float4(g.x * over(v.x), g.y * over(v.y), g.z * over(v.z), g.w * over(v.w));
  }
}

// out[i] = table[tok[rows_map[i]]] for i < *n_dev
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ tok,
                                                         const int32_t* __restrict__ rows_map,
                                                         const int64_t* __restrict__ n_dev, int d4, int vocab,
                                                         float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= *n_dev) return;
  int64_t id = tok[rows_map[i]];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* src = reinterpret_cast<const float4*>(table) + id * d4;
  float4* dst = reinterpret_cast<float4*>(out) + i * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) dst[c] = src[c];
}

// slot[rows_map[i]][c0 ..] = src[i][c0 ..] (rows of n4 float4, columns from c0 on), i < *n_dev: the compacted (k, v) of this
// step into the cache slot — the q third of a row stays in the compacted buffer, where this step's attention reads it; nobody
// reads q from the cache in the prefix-table mode (descendants need their ancestors' K / V only)
__global__ __launch_bounds__(256) void scatter_slot_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows_map,
                                                           const int64_t* __restrict__ n_dev, int n4, int c0,
                                                           float* __restrict__ slot) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= *n_dev) return;
  const float4* s4 = reinterpret_cast<const float4*>(src) + i * n4;
  float4* d4 = reinterpret_cast<float4*>(slot) + (int64_t)rows_map[i] * n4;
  for (int c = c0 + (threadIdx.x & 63); c < n4; c += 64) d4[c] = s4[c];
}

// Teacher forcing: logits[r][c] = table[b][pos][last_token][token(pos,c)]
__global__ void table_logits_kernel(const float* __restrict__ table, const int64_t* __restrict__ cur_tok, int rows,
                                    int R, int V, int Vd, int maxlen, int pos, float* __restrict__ logits) {
  const int item = blockIdx.x * blockDim.x + threadIdx.x;
  const int V1 = V + 1;
  if (item >= rows * V1) return;
  const int r = item / V1, c = item % V1, b = r / R;
  const int tok = c < V ? pos * V + 2 + c : EOS_ID;
  logits[item] = table[(((size_t)b * maxlen + pos) * Vd + (size_t)cur_tok[r]) * Vd + tok];
}

// LDS operations of one wave execute in order: between steps that only exchange data inside a wave this wave-level barrier
// (plus fences for the compiler) is all the synchronisation needed.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per query: log_softmax of every beam's row (masked columns contribute exp(-1e9 - max) = 0 exactly), add the
// beam score, take the 2R best of the R*(V+1) unmasked candidates, sorted (generation_utils.py:698,766-775).
__global__ __launch_bounds__(1024) void beam_topk_kernel(BeamBufs bb, BeamDims bd, int pos, int npad, int cur, int bcast,
                                                        float* __restrict__ step_scores,
                                                        int32_t* __restrict__ step_tokens) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // [npad]
  if (bb.live_gate && *bb.live_gate == 0) return;  // uniform: every query is done, nothing is ranked any more
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int R = bd.R, V1 = bd.V + 1, nthr = blockDim.x, nwaves = blockDim.x >> 6;
  float* lse_max = reinterpret_cast<float*>(keys + npad);  // [R]
  float* lse_log = lse_max + R;                            // [R]
  float* lg_s = lse_log + R;                               // [R*V1] this query's logits, staged once (coalesced)
  const int ncand = R * V1;
  // bcast (step 0 with one computed row per query): logits holds [B][V1], every beam of the query reads row b
  for (int e = tid; e < ncand; e += nthr) lg_s[e] = bcast ? bb.logits[(size_t)b * V1 + e % V1] : bb.logits[(size_t)b * ncand + e];
  __syncthreads();
  for (int j = wave; j < R; j += nwaves) {
    const float* lg = lg_s + j * V1;
    float mx = -INFINITY;
    for (int c = lane; c < V1; c += 64) mx = fmaxf(mx, lg[c]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int c = lane; c < V1; c += 64) sm += expf(lg[c] - mx);
    sm = wave_sum(sm);
    if (lane == 0) {
      lse_max[j] = mx;
      lse_log[j] = logf(sm);
    }
  }
  __syncthreads();
  for (int e = tid; e < npad; e += nthr) {
    unsigned long long key = 0ull;
    if (e < ncand) {
      const int j = e / V1, c = e - j * V1;
      const float logp = (lg_s[e] - lse_max[j]) - lse_log[j];
      float s = logp + bb.beam_scores[(size_t)b * R + j];
      if (bd.trie_eos) {  // scores += mask(-inf on tokens that are not children of the prefix' node)
        const int nd = bb.node[cur][(size_t)b * R + j];
        const bool ok = c < bd.V ? (nd >= 0 && bd.trie_child[(size_t)nd * bd.V + c] >= 0) : (nd < 0 || bd.trie_eos[nd] != 0);
        if (!ok) s = -INFINITY;
      }
      const int tok = c < bd.V ? pos * bd.V + 2 + c : EOS_ID;
      const uint32_t flat = (uint32_t)(j * bd.Vd + tok);
      key = ((unsigned long long)dfkey(s) << 32) | (unsigned long long)(0xFFFFFFFFu - flat);
    }
    keys[e] = key;
  }
  __syncthreads();
  // Bitonic sort, descending.  A wave owns a block of epw consecutive keys: every stage whose compare-exchange pairs stay
  // inside a block (2*stride <= epw) needs no workgroup barrier — LDS operations of one wave execute in order — so of the
  // 78 stages of a 4096-key sort (100 beams) only the 10 with stride >= 256 cost a barrier pair.  The kernel takes 49 us
  // per step at 100 beams (rocprofv3); keeping the keys in registers (4 per lane: in-thread swaps, 64-bit shuffles inside a
  // wave, LDS only across waves) was built and measured no faster (54 us) — 8 ds_bpermute per shuffle stage cost what the
  // LDS round trip of a stage costs.
  const int epw = npad / nwaves;
  if (epw >= 2 * R && nwaves > 1) {
    // Only the 2R best of the npad keys are wanted and a wave's block holds at least that many: every wave sorts its own
    // block (descending, no workgroup barrier), then blocks are merged pairwise keeping the better half — for A and B
    // sorted descending, max(A[i], B[epw-1-i]) is a bitonic sequence holding the epw largest of both, which log2(epw)
    // wave-local stages sort — until block 0 holds the top epw of all.  log2(nwaves) + 1 workgroup barriers where the full
    // network below takes a barrier pair on each of its 10 wide stages: 49 -> (measured) us per step at 100 beams.
    unsigned long long* blk = keys + wave * epw;
    for (int size = 2; size <= epw; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int q = lane; q < (epw >> 1); q += 64) {
          const int lo = (q / stride) * (stride << 1) + (q % stride), hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long x = blk[lo], y = blk[hi];
          if ((x < y) == desc) blk[lo] = y, blk[hi] = x;
        }
        wave_sync();
      }
    }
    for (int nb = nwaves; nb > 1; nb >>= 1) {
      __syncthreads();
      if (wave < (nb >> 1)) {
        const unsigned long long* other = keys + (wave + (nb >> 1)) * epw;
        for (int q = lane; q < epw; q += 64) {
          const unsigned long long x = blk[q], y = other[epw - 1 - q];
          blk[q] = x > y ? x : y;
        }
        wave_sync();
        for (int stride = epw >> 1; stride > 0; stride >>= 1) {
          for (int q = lane; q < (epw >> 1); q += 64) {
            const int lo = (q / stride) * (stride << 1) + (q % stride), hi = lo + stride;
            const unsigned long long x = blk[lo], y = blk[hi];
            if (x < y) blk[lo] = y, blk[hi] = x;
          }
          wave_sync();
        }
      }
    }
  } else {
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (2 * stride <= epw) {
        for (int q = lane; q < (epw >> 1); q += 64) {
          const int t = (epw >> 1) * wave + q;
          const int lo = (t / stride) * (stride << 1) + (t % stride), hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long x = keys[lo], y = keys[hi];
          if ((x < y) == desc) {
            keys[lo] = y;
            keys[hi] = x;
          }
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
          if ((x < y) == desc) {
            keys[lo] = y;
            keys[hi] = x;
          }
        }
        __syncthreads();
      }
    }
  }
  }
  __syncthreads();
  for (int i = tid; i < 2 * R; i += nthr) {
    const unsigned long long key = keys[i];
    const float s = dfkey_inv((uint32_t)(key >> 32));
    const int32_t flat = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
    bb.cand_score[(size_t)b * 2 * R + i] = s;
    bb.cand_idx[(size_t)b * 2 * R + i] = flat;
    if (step_scores) {
      step_scores[(size_t)b * 2 * R + i] = s;
      step_tokens[(size_t)b * 2 * R + i] = flat;
    }
  }
}

// ---- BeamHypotheses of one query, held in LDS by the single wave that serves the query -------------------------------
// The reference keeps `self.beams` as a Python list (generation_utils.py:1052-1099): add() appends, an over-full list
// drops its worst entry with `del` (order of the rest kept), the final pick is a stable sort by score popped from the
// end.  Only the RELATIVE list order of surviving entries is ever observed (tie-breaks), so each entry carries its
// insertion number and the storage order is free: a delete moves the last entry into the hole.
// The hypothesis-heap helpers below are executed by ONE wave (the first wave of the block): LDS operations of a wave
// execute in order, so a wave-level barrier (plus fences for the compiler) is all the synchronisation they need.
struct HypLds {
  double* sc;   // [R+1]
  int32_t* ln;  // [R+1]
  int32_t* sq;  // [R+1]
  int32_t* tk;  // [R+1][ml]
  int n, next;
  double worst;
};
__host__ __device__ static size_t hyp_lds_bytes(int R, int ml) { return (size_t)(R + 1) * (8 + 4 + 4 + 4 * (size_t)ml) + 8; }

__device__ __forceinline__ void hyp_load(HypLds& H, char* smem, const BeamBufs& bb, const BeamDims& bd, int b, int lane) {
  const int cap = bd.R + 1, ml = bd.maxlen;
  H.sc = reinterpret_cast<double*>(smem);
  H.ln = reinterpret_cast<int32_t*>(H.sc + cap);
  H.sq = H.ln + cap;
  H.tk = H.sq + cap;
  H.n = bb.hyp_cnt[b], H.next = bb.hyp_next[b], H.worst = bb.hyp_worst[b];
  for (int i = lane; i < H.n; i += 64) {
    H.sc[i] = bb.hyp_score[(size_t)b * cap + i];
    H.ln[i] = bb.hyp_len[(size_t)b * cap + i];
    H.sq[i] = bb.hyp_seq[(size_t)b * cap + i];
  }
  for (int e = lane; e < H.n * ml; e += 64) H.tk[e] = bb.hyp_tok[(size_t)b * cap * ml + e];
  wave_sync();
}

// the same view of the LDS arrays for waves that do not take part in the load (beam_finalize_kernel)
__device__ __forceinline__ void hyp_load_ptrs(HypLds& H, char* smem, const BeamBufs& bb, const BeamDims& bd, int b) {
  const int cap = bd.R + 1;
  H.sc = reinterpret_cast<double*>(smem);
  H.ln = reinterpret_cast<int32_t*>(H.sc + cap);
  H.sq = H.ln + cap;
  H.tk = H.sq + cap;
  H.n = bb.hyp_cnt[b], H.next = bb.hyp_next[b], H.worst = bb.hyp_worst[b];
}

__device__ __forceinline__ void hyp_store(const HypLds& H, const BeamBufs& bb, const BeamDims& bd, int b, int lane) {
  const int cap = bd.R + 1, ml = bd.maxlen;
  wave_sync();
  for (int i = lane; i < H.n; i += 64) {
    bb.hyp_score[(size_t)b * cap + i] = H.sc[i];
    bb.hyp_len[(size_t)b * cap + i] = H.ln[i];
    bb.hyp_seq[(size_t)b * cap + i] = H.sq[i];
  }
  for (int e = lane; e < H.n * ml; e += 64) bb.hyp_tok[(size_t)b * cap * ml + e] = H.tk[e];
  if (lane == 0) bb.hyp_cnt[b] = H.n, bb.hyp_next[b] = H.next, bb.hyp_worst[b] = H.worst;
}

// BeamHypotheses.add (generation_utils.py:1070-1084); Python-float arithmetic = double.  Called by all 64 lanes of the
// query's wave with uniform arguments.
// `len_pow` = pow(len, length_penalty), computed once by the caller (every add of one launch has the same length).
__device__ __forceinline__ void hyp_add(HypLds& H, const BeamDims& bd, const int32_t* toks, int len, double sum_logp, int lane, double len_pow) {
  const int ml = bd.maxlen;
  const double score = sum_logp / len_pow;
  if (!(H.n < bd.R || score > H.worst)) return;
  if (lane == 0) H.sc[H.n] = score, H.ln[H.n] = len, H.sq[H.n] = H.next;
  for (int t = lane; t < len; t += 64) H.tk[(size_t)H.n * ml + t] = toks[t];
  ++H.n, ++H.next;
  wave_sync();
  if (H.n > bd.R) {
    // sorted([(s, idx)])[0] is removed (lowest score, then lowest list index = lowest insertion number);
    // [1] gives the new worst score
    double bs = INFINITY;
    int bq = 0x7fffffff, bi = -1;
    for (int i = lane; i < H.n; i += 64) {
      const double v = H.sc[i];
      const int q = H.sq[i];
      if (v < bs || (v == bs && q < bq)) bs = v, bq = q, bi = i;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double os = __shfl_xor(bs, off);
      const int oq = __shfl_xor(bq, off), oi = __shfl_xor(bi, off);
      if (os < bs || (os == bs && oq < bq)) bs = os, bq = oq, bi = oi;
    }
    const int lo = bi;
    double second = INFINITY;
    for (int i = lane; i < H.n; i += 64)
      if (i != lo && H.sc[i] < second) second = H.sc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) second = fmin(second, __shfl_xor(second, off));
    const int last = H.n - 1;
    if (lo != last) {
      if (lane == 0) H.sc[lo] = H.sc[last], H.ln[lo] = H.ln[last], H.sq[lo] = H.sq[last];
      for (int t = lane; t < ml; t += 64) H.tk[(size_t)lo * ml + t] = H.tk[(size_t)last * ml + t];
    }
    --H.n;
    H.worst = second;
    wave_sync();
  } else {
    H.worst = score < H.worst ? score : H.worst;
  }
}

// The ancestor table of a query's R rows for the position about to be processed (pos = cur_len: the new rows have length
// cur_len + 1): position p < pos is inherited from the row's parent, position pos is the row itself.  Runs at the end of
// beam_update_kernel (the parents were written by this workgroup: visible after its barrier) — a launch of its own in round 2.
__device__ __forceinline__ void anc_update_query(const BeamBufs& bb, const BeamDims& bd, int b, int pos, int cur, int tid) {
  __syncthreads();
  const int R = bd.R, ml = bd.maxlen, stride = pos + 1, rows = bd.B * R;
  for (int e = tid; e < R * stride; e += 256) {
    const int j = e / stride, p = e - j * stride, r = b * R + j;
    const int a = p < pos ? bb.anc[cur][(size_t)bb.parent[r] * ml + p] : r;
    bb.anc[cur ^ 1][(size_t)r * ml + p] = a;
    bb.kv_rows[(size_t)r * stride + p] = p * rows + a;
  }
}

// The host loop of generation_utils.py:783-850, one workgroup of four waves per query.  What that loop decides is fixed
// by the ranked candidate list alone: the next beams are the first R non-EOS candidates, and the EOS candidates that are
// offered to the hypothesis heap are those of rank < R that come before the R-th non-EOS one — so the non-EOS ranks are
// found with ballots / popcounts, and only the (few) EOS candidates walk the heap one after another, in rank order, on the
// first wave.  All four waves then fill the rows.  cur = index of the current seq buffer.
__global__ __launch_bounds__(256) void beam_update_kernel(BeamBufs bb, BeamDims bd, int cur_len, int cur) {
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  __shared__ int n_sh;
  // every query was done BEFORE this step (the word is only cleared by a previous launch of this kernel): the step would pad
  // every entry (:786-794) and nothing reads the padding — beam_finalize takes done queries from their hypothesis heaps
  if (bb.live_gate && *bb.live_gate == 0) return;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = bd.R, ml = bd.maxlen;
  const int32_t* seq_c = bb.seq[cur];
  int32_t* seq_n = bb.seq[cur ^ 1];
  if (bb.done[b]) {  // :786-794 padded batch entry
    for (int j = tid; j < R; j += 256) {
      const int row = b * R + j;
      bb.beam_scores[row] = 0.f;
      bb.cur_tok[row] = PAD_ID;
      bb.parent[row] = b * R;
      bb.node[cur ^ 1][row] = -1;
    }
    for (int e = tid; e < R * (cur_len + 1); e += 256) {
      const int j = e / (cur_len + 1), t = e - j * (cur_len + 1);
      seq_n[(size_t)(b * R + j) * ml + t] = t < cur_len ? seq_c[(size_t)(b * R) * ml + t] : PAD_ID;
    }
    anc_update_query(bb, bd, b, cur_len, cur, tid);
    return;
  }
  int32_t* sel = reinterpret_cast<int32_t*>(bsm + ((hyp_lds_bytes(R, ml) + 15) & ~(size_t)15));  // [R] chosen candidate ranks
  int32_t* ci = sel + R;                               // [2R] ranked candidates, staged once
  float* cs = reinterpret_cast<float*>(ci + 2 * R);    // [2R]
  for (int e = tid; e < 2 * R; e += 256) {
    ci[e] = bb.cand_idx[(size_t)b * 2 * R + e];
    cs[e] = bb.cand_score[(size_t)b * 2 * R + e];
  }
  __syncthreads();
  if (wave == 0) {
    HypLds H;
    hyp_load(H, bsm, bb, bd, b, lane);
    // pass 1: rank of the j-th non-EOS candidate -> sel[j]; `end` = one past the rank at which the R-th one is taken
    int n = 0, end = 2 * R;
    for (int base = 0; base < 2 * R && n < R; base += 64) {
      const int rank = base + lane;
      const bool live = rank < 2 * R && (ci[rank] % bd.Vd) != EOS_ID;
      const unsigned long long m = __ballot(live);
      const int before = n + __popcll(m & ((1ull << lane) - 1ull));
      if (live && before < R) sel[before] = rank;
      const int tot = n + __popcll(m);
      if (tot >= R) {  // the R-th non-EOS candidate sits in this chunk: find its rank
        const unsigned long long hit = __ballot(live && before == R - 1);
        end = base + (int)__ffsll((long long)hit);
      }
      n = tot < R ? tot : R;
    }
    // pass 2: EOS candidates of rank < min(R, end) enter the heap, in rank order (:811-817)
    bool touched = false;
    const double len_pow = pow((double)cur_len, bd.lp);
    const int lim = end < R ? end : R;
    for (int base = 0; base < lim; base += 64) {
      const int rank = base + lane;
      unsigned long long m = __ballot(rank < lim && (ci[rank] % bd.Vd) == EOS_ID);
      while (m) {
        const int rk = base + (int)__ffsll((long long)m) - 1;
        m &= m - 1;
        hyp_add(H, bd, seq_c + (size_t)(b * R + ci[rk] / bd.Vd) * ml, cur_len, (double)cs[rk], lane, len_pow);
        touched = true;
      }
    }
    if (touched) hyp_store(H, bb, bd, b, lane);
    // :827-829  is_done(best_sum_logprobs = next_scores[b].max(), cur_len)
    if (lane == 0) {
      n_sh = n;
      if (H.n >= R) {
        const double cur_score = (double)cs[0] / len_pow;
        if (H.worst >= cur_score) {
          bb.done[b] = 1;  // set once: a done query takes the early return at the top from now on
          // generation_utils.py:836-838 `if all(done): break` — the host polls this word between steps (generate_impl)
          if (atomicAdd(bb.n_done, 1) + 1 == bd.B) {
            *bb.live = 0;  // the steps already enqueued skip their linears (StreamK::live) and their miss-row chain
            if (bb.all_done_host) {
              __hip_atomic_store(bb.all_done_host + 64, cur_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the step it happened at
              // release: a host that sees the epoch also sees the step word stored just above
              __hip_atomic_store(bb.all_done_host, bb.done_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const int n = n_sh;
  for (int j = tid; j < n; j += 256) {
    const int rank = sel[j];
    const int flat = ci[rank];
    const int beam = flat / bd.Vd, tok = flat % bd.Vd;
    const int eff = b * R + beam, row = b * R + j;
    bb.beam_scores[row] = cs[rank];
    bb.cur_tok[row] = tok;
    bb.parent[row] = eff;
    if (bd.trie_child) {
      const int nd = bb.node[cur][eff], c = tok - ((cur_len - 1) * bd.V + 2);
      int nx = (nd >= 0 && c >= 0 && c < bd.V) ? bd.trie_child[(size_t)nd * bd.V + c] : -1;
      if (nx >= bd.trie_nodes) nx = -1;  // a malformed table must not send later lookups out of bounds
      bb.node[cur ^ 1][row] = nx;
    }
  }
  for (int e = tid; e < n * (cur_len + 1); e += 256) {
    const int j = e / (cur_len + 1), t = e - j * (cur_len + 1);
    const int flat = ci[sel[j]];
    const int beam = flat / bd.Vd, tok = flat % bd.Vd;
    seq_n[(size_t)(b * R + j) * ml + t] = t < cur_len ? seq_c[(size_t)(b * R + beam) * ml + t] : tok;
  }
  anc_update_query(bb, bd, b, cur_len, cur, tid);
}

// :862-919 finalise open beams, pick the nret best per query, lay out tokens / EOS / PAD.  One workgroup of four waves per query.
// The R open beams of a query that is not done enter its hypothesis heap one after another in the reference (:863-883).  What that
// sequence leaves behind is decided by the scores alone except inside ONE tie group: with v = the R-th largest score over the old
// entries and the R new ones, every entry above v survives (when it is offered the full heap holds something smaller, and nothing
// can evict it), nothing below v does (a full heap's minimum never decreases), and when the entries equal to v all fit they all
// survive — the heap ends with R entries and they are the only candidates left.  So the survivors are found by counting (R^2
// compares spread over 256 lanes instead of R dependent heap updates of one wave: 165 us at 100 beams); only when the tie group AT
// v is cut by the capacity — equal double-precision scores of different hypotheses — the admission / eviction order matters
// (strict `>` against the worst score, the oldest of the worst evicted) and the original one-by-one walk runs.  The insertion numbers
// only ever decide the relative order of equal scores: new entry j takes next + j (offer order).
__global__ __launch_bounds__(256) void beam_finalize_kernel(BeamBufs bb, BeamDims bd, int final_len, int cur, int max_length,
                                                            int64_t* out_ids, int32_t* out_len, double* out_scores) {
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  __shared__ int n_fin, cut_tie, n_keep;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = bd.R, ml = bd.maxlen;
  HypLds H;
  if (wave == 0) hyp_load(H, bsm, bb, bd, b, lane);
  else hyp_load_ptrs(H, bsm, bb, bd, b);
  int32_t* order = reinterpret_cast<int32_t*>(bsm + ((hyp_lds_bytes(R, ml) + 15) & ~(size_t)15));  // [nret] entry of rank j
  int32_t* rows_s = order + bd.nret;                         // [R][ml] the open beams' rows
  float* sc_s = reinterpret_cast<float*>(rows_s + R * ml);   // [R]
  double* all_sc = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(sc_s + R) + 7) & ~(uintptr_t)7);  // [2R + 1] old entries' scores, then the new ones'
  int32_t* keep = reinterpret_cast<int32_t*>(all_sc + 2 * R + 1);  // [2R + 1] slot of a surviving entry, -1 otherwise
  if (tid == 0) n_fin = H.n, cut_tie = 0, n_keep = 0;
  __syncthreads();
  const bool open = !bb.done[b];  // uniform
  if (open) {
    // the rows are staged in LDS first: read one by one from device memory, every add paid a dependent round trip
    for (int e = tid; e < R * ml; e += 256) rows_s[e] = bb.seq[cur][(size_t)b * R * ml + e];
    for (int j = tid; j < R; j += 256) sc_s[j] = bb.beam_scores[b * R + j];
    __syncthreads();
    const double len_pow = pow((double)final_len, bd.lp);
    const int n0 = H.n, T = n0 + R;
    for (int i = tid; i < T; i += 256) all_sc[i] = i < n0 ? H.sc[i] : (double)sc_s[i - n0] / len_pow;
    __syncthreads();
    // survivors by counting; a cut tie group is flagged
    for (int i = tid; i < T; i += 256) {
      const double v = all_sc[i];
      int gt = 0, eq = 0;
      for (int j = 0; j < T; ++j) {
        const double w = all_sc[j];
        gt += w > v ? 1 : 0, eq += w == v ? 1 : 0;
      }
      const bool in = gt + eq <= R;               // the whole group of equal scores fits
      if (!in && gt < R && eq > 1) cut_tie = 1;   // the group at the boundary is cut and holds several entries (benign race: all store 1)
      // a cut group of ONE entry cannot exist (gt < R and eq == 1 means gt + eq <= R)
      keep[i] = in ? 1 : 0;
    }
    __syncthreads();
    if (cut_tie) {
      if (wave == 0) {  // the reference's walk, one add after another
        for (int j = 0; j < R; ++j) hyp_add(H, bd, rows_s + j * ml, final_len, (double)sc_s[j], lane, len_pow);
        if (lane == 0) n_fin = H.n;
      }
    } else {
      // compact the survivors into the heap arrays: old entries first (their relative storage order is free), then new ones;
      // old entries that do not survive leave holes that surviving NEW entries fill
      if (wave == 0) {
        // slots: survivors are numbered in index order by ballot prefix sums
        int base = 0;
        for (int i0 = 0; i0 < T; i0 += 64) {
          const int i = i0 + lane;
          const bool kp = i < T && keep[i] != 0;
          const unsigned long long m = __ballot(kp);
          if (i < T) keep[i] = kp ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
          base += __popcll(m);
        }
        if (lane == 0) n_fin = base;
      }
      __syncthreads();
      // Move in two steps through registers-free staging: scores / lengths / insertion numbers / tokens of survivors are first
      // gathered into the tail region (rows_s is reused as scratch for tokens is not possible — it is the source), so the old entries
      // are moved in ascending slot order by one wave (slot <= index: a survivor never moves up), then new entries are written.
      if (wave == 0) {
        for (int i = 0; i < n0; ++i) {  // uniform loop; slot <= i, so the source of a later move is never overwritten earlier
          const int sl = keep[i];
          if (sl >= 0 && sl != i) {
            if (lane == 0) H.sc[sl] = H.sc[i], H.ln[sl] = H.ln[i], H.sq[sl] = H.sq[i];
            for (int t = lane; t < ml; t += 64) H.tk[(size_t)sl * ml + t] = H.tk[(size_t)i * ml + t];
            wave_sync();
          }
        }
      }
      __syncthreads();
      for (int j = wave; j < R; j += 4) {  // a wave per new entry
        const int sl = keep[n0 + j];
        if (sl < 0) continue;
        if (lane == 0) H.sc[sl] = all_sc[n0 + j], H.ln[sl] = final_len, H.sq[sl] = H.next + j;
        for (int t = lane; t < final_len; t += 64) H.tk[(size_t)sl * ml + t] = rows_s[j * ml + t];
      }
    }
  }
  __syncthreads();
  // sorted(beams, key=score) is stable ascending and pop() takes the last: highest score first, among equal scores the
  // later list entry first.  rank(i) = number of entries that precede i in that order.
  const int n = n_fin;
  for (int i = tid; i < n; i += 256) {
    const double v = H.sc[i];
    const int q = H.sq[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double w = H.sc[j];
      rank += (w > v || (w == v && H.sq[j] > q)) ? 1 : 0;
    }
    if (rank < bd.nret) order[rank] = i;
  }
  __syncthreads();
  for (int j = tid; j < bd.nret; j += 256) {
    const size_t o = (size_t)b * bd.nret + j;
    // fewer hypotheses than requested cannot happen when V^depth >= R (the reference would raise)
    out_len[o] = j < n ? H.ln[order[j]] : 0;
    out_scores[o] = j < n ? H.sc[order[j]] : -INFINITY;
  }
  for (int e = tid; e < bd.nret * max_length; e += 256) {
    const int j = e / max_length, t = e - j * max_length;
    int64_t v = PAD_ID;
    if (j < n) {
      const int best = order[j], len = H.ln[best];
      if (t < len) v = H.tk[(size_t)best * ml + t];
      else if (t == len) v = EOS_ID;  // :915-916 (len < max_length)
    }
    out_ids[((size_t)b * bd.nret + j) * max_length + t] = v;
  }
}

// ------------------------------------------------------------------------------------------ driver pieces
static int next_pow2i(int x) {
  int p = 64;
  while (p < x) p <<= 1;
  return p;
}

static int check_beam_dims(const BeamDims& bd, int max_length) {
  GDR_CHECK_ARG(bd.B > 0 && bd.R >= 2 && bd.R <= 256, "beam: B=%d num_beams=%d (need 2..256)", bd.B, bd.R);
  GDR_CHECK_ARG(bd.nret >= 1 && bd.nret <= bd.R, "beam: num_return_sequences=%d must be in [1,num_beams]", bd.nret);
  GDR_CHECK_ARG(max_length >= 2 && max_length <= MAXLEN_CAP, "beam: max_length=%d must be in [2,%d]", max_length,
                MAXLEN_CAP);
  GDR_CHECK_ARG(bd.V >= 1 && bd.Vd >= bd.V * (max_length - 1) + 2, "beam: decode vocab %d too small for V=%d, max_length=%d",
                bd.Vd, bd.V, max_length);
  GDR_CHECK_ARG(bd.R * (bd.V + 1) <= 8192, "beam: num_beams*(V+1)=%d exceeds the 8192-entry sort buffer", bd.R * (bd.V + 1));
  GDR_CHECK_ARG(bd.lp > 0.0, "beam: length_penalty must be > 0");
  return GDR_OK;
}

// One decode step's beam machinery after bb.logits holds the step's unmasked-column logits.
static int beam_step(const BeamBufs& bb, const BeamDims& bd, int pos, int cur, float* step_scores,
                     int32_t* step_tokens, hipStream_t stream, bool bcast = false) {
  const int npad = next_pow2i(bd.R * (bd.V + 1));
  const size_t lds = (size_t)npad * 8 + (size_t)bd.R * 8 + (size_t)bd.R * (bd.V + 1) * 4;
  const size_t tr = (size_t)pos * bd.B * 2 * bd.R;
  hipLaunchKernelGGL(beam_topk_kernel, dim3(bd.B), dim3(npad >= 2048 ? 1024 : 256), lds, stream, bb, bd, pos, npad, cur,
                     bcast ? 1 : 0, step_scores ? step_scores + tr : nullptr, step_tokens ? step_tokens + tr : nullptr);
  GDR_CHECK_LAUNCH("beam_topk_kernel");
  const size_t hyp_lds = ((hyp_lds_bytes(bd.R, bd.maxlen) + 15) & ~(size_t)15) + (size_t)bd.R * 4 + (size_t)bd.R * 16;
  hipLaunchKernelGGL(beam_update_kernel, dim3(bd.B), dim3(256), hyp_lds, stream, bb, bd, pos + 1, cur);
  GDR_CHECK_LAUNCH("beam_update_kernel");
  return GDR_OK;  // the ancestor table of the next position is rebuilt at the end of beam_update_kernel
}

static int beam_begin(const BeamBufs& bb, const BeamDims& bd, hipStream_t stream, bool dedup0 = false) {
  const int rows = bd.B * bd.R;
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(beam_topk_kernel), 96 * 1024, "beam")) return rc__;
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(beam_finalize_kernel), 96 * 1024, "beam")) return rc__;
  hipLaunchKernelGGL(beam_init_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, bb, bd, dedup0 ? 1 : 0);
  GDR_CHECK_LAUNCH("beam_init_kernel");
  return GDR_OK;
}

static int beam_end(const BeamBufs& bb, const BeamDims& bd, int max_length, int cur, int64_t* out_ids,
                    int32_t* out_len, double* out_scores, hipStream_t stream) {
  const size_t hyp_lds = ((hyp_lds_bytes(bd.R, bd.maxlen) + 15) & ~(size_t)15) + (size_t)bd.nret * 4 +
                         (size_t)bd.R * bd.maxlen * 4 + (size_t)(bd.R + 2) * 4 +   // + the staged beam rows and scores
                         (size_t)(2 * bd.R + 2) * 12 + 16;                         // + all scores (double) and survivor slots
  hipLaunchKernelGGL(beam_finalize_kernel, dim3(bd.B), dim3(256), hyp_lds, stream, bb, bd, max_length, cur, max_length, out_ids,
                     out_len, out_scores);
  GDR_CHECK_LAUNCH("beam_finalize_kernel");
  return GDR_OK;
}

// The adaptor chain (decode_embeddings -> 4 post-LN layers) and the T5 decoder stack of one step are independent until
// the head consumes both, and each of their kernels fills only part of the chip at decode batch sizes: they run
// concurrently, the adaptor on a side stream forked / joined with events (graph-capturable pattern).  Side streams are
// leased per CALL from a per-device pool (created on first use on the device that is current in the calling thread), so
// concurrent gdr_t5_generate calls from several host threads or on several devices never share one.
// `if all(done): break` (generation_utils.py:836-838) without a host sync: beam_update_kernel stores the call's epoch into a
// host-mapped word when the last query becomes done, and the host looks at that word before it enqueues the next step.  The
// host runs ahead of the GPU by less than a step's worth of launches at decode batch sizes, so a call whose beams all finish
// at step s enqueues at most a step or two more (which change nothing: done queries only pad, :786-794) instead of all
// max_length - 1.  Random weights (bench.py) never finish early; a trained model does after the docid's length + 1 steps, and so
// does every trie-constrained call.  64 words used round-robin by epoch: a stale store from the call 64 epochs ago cannot match.
static int32_t* done_words() {
  static int32_t* words = [] {
    int32_t* h = nullptr;
    // words 0..63: the epoch words; words 64..127: the decode step (cur_len) at which word i's call saw its last query finish
    if (hipHostMalloc(reinterpret_cast<void**>(&h), 128 * sizeof(int32_t), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) !=
        hipSuccess) {
      (void)hipGetLastError();
      return static_cast<int32_t*>(nullptr);
    }
    for (int i = 0; i < 128; ++i) h[i] = 0;
    return h;
  }();
  return words;
}
static std::atomic<int32_t> g_done_epoch{0};
static std::atomic<int64_t> g_early_exits{0};
constexpr int SIDE_MAX_LAYERS = 48;
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  hipEvent_t ckv[SIDE_MAX_LAYERS] = {};  // "cross-attention K/V of layer l are projected" (gdr_t5_generate, before step 0)
  int dev = -1;
};
static std::mutex g_side_mu;
static std::vector<SideStream*> g_side_free;

static SideStream* side_stream_acquire() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  {
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (size_t i = 0; i < g_side_free.size(); ++i)
      if (g_side_free[i]->dev == dev) {
        SideStream* x = g_side_free[i];
        g_side_free.erase(g_side_free.begin() + i);
        return x;
      }
  }
  SideStream* x = new SideStream();
  x->dev = dev;
  if (hipStreamCreateWithFlags(&x->s, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&x->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&x->join, hipEventDisableTiming) != hipSuccess) {
    delete x;  // the caller falls back to running the adaptor chain on the main stream
    return nullptr;
  }
  for (int l = 0; l < SIDE_MAX_LAYERS; ++l)
    if (hipEventCreateWithFlags(&x->ckv[l], hipEventDisableTiming) != hipSuccess) {
      delete x;
      return nullptr;
    }
  return x;
}
// A lease ends when the call has finished ENQUEUEING: the stream is in-order, so the next lessee's work simply queues
// behind what is still in flight, and a wait already enqueued on `join` keeps the event state it captured.
struct SideLease {
  SideStream* ss;
  SideLease() : ss(side_stream_acquire()) {}
  ~SideLease() {
    if (!ss) return;
    std::lock_guard<std::mutex> lock(g_side_mu);
    g_side_free.push_back(ss);
  }
};

// One linear of the decode path.  fp32: the split-K / small-tile launcher (device-side row count when m_dev is given).
// bf16 precision mode (BASELINE config C5): W points to bf16 data, the activation operand is rounded to bf16 into `abf`
// (RNE) and the GEMM accumulates in fp32; bias / residual / output stay fp32.
// bf16 mode: the norm that produced a linear's fp32 operand also left it rounded to bf16 (`buf`), so the linear needs no cast
// launch of its own — `src` says which fp32 buffer the image belongs to; anything else is cast as before.  One per chain
// (stream); whoever rewrites a normed buffer by other means resets `src`.
struct Bf16Image {
  const float* src = nullptr;
  void* buf = nullptr;
};

static int dec_linear(bool bf16, void* abf, const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                      int64_t M, const int64_t* m_dev, int N, int K, int epi, const float* bias, const float* res, int64_t ldr,
                      float* skw, hipStream_t st, StreamK* sk = nullptr, const Bf16Image* img = nullptr) {
  if (!bf16)
    return m_dev ? launch_linear_f32_ws_dev(A, lda, W, ldw, C, ldc, M, m_dev, N, K, epi, bias, res, ldr, skw, SPLITK_WS_BYTES, st, sk)
                 : launch_linear_f32_ws(A, lda, W, ldw, C, ldc, M, N, K, epi, bias, res, ldr, skw, SPLITK_WS_BYTES, st, sk);
  if (M == 0) return GDR_OK;
  GDR_CHECK_ARG(lda == K && K % 8 == 0, "decode(bf16): the activation operand must be dense with K %% 8 == 0");
  if (img && img->buf && img->src == A)  // the producing norm already wrote this operand's bf16 image
    return launch_linear_bf16(img->buf, K, W, ldw, C, ldc, M, N, K, epi, bias, res, ldr, st, m_dev);
  if (int rc = launch_cast_f32_bf16(A, abf, M * (int64_t)K, st)) return rc;
  return launch_linear_bf16(abf, K, W, ldw, C, ldc, M, N, K, epi, bias, res, ldr, st, m_dev);
}
// bf16 mode: a linear whose output feeds nothing but the next linear (wi -> wo_ff, linear1 -> linear2) emits it as bf16 straight
// from the GEMM epilogue — no fp32 copy is written and the consumer needs no cast launch (the rounding is the cast kernel's RNE of
// the same fp32 value: results are bit-identical).  out->buf receives the rows, out->src is set to `C` (the fp32 buffer the
// consumer names as its operand).  Returns 1 when the LDS-DMA kernel does not serve the shape (the caller runs the fp32 form).
static int dec_linear_out16(void* abf, const float* A, int64_t lda, const float* W, int64_t ldw, const float* C, int64_t ldc,
                            int64_t M, const int64_t* m_dev, int N, int K, int epi, const float* bias, hipStream_t st,
                            const Bf16Image* img, Bf16Image* out) {
  if (!out || !out->buf || lda != K || K % 64 != 0 || ldc != N) return 1;
  if (M == 0) return GDR_OK;
  const bool nb = epi == GDR_EPI_BIAS || epi == GDR_EPI_BIAS_RELU;
  const int act = (epi == GDR_EPI_RELU || epi == GDR_EPI_BIAS_RELU) ? 1 : 0;
  if (epi == GDR_EPI_RESIDUAL || epi == GDR_EPI_BIAS_RESIDUAL || epi == GDR_EPI_BIAS_GELU || (nb && !bias)) return 1;
  const void* a16 = abf;
  if (img && img->buf && img->src == A) {
    a16 = img->buf;
  } else if (int rc = launch_cast_f32_bf16(A, abf, M * (int64_t)K, st)) {
    return rc;
  }
  ProfScope prof(PROF_LINEAR, 2.0 * (double)M * (double)N * (double)K, st);
  const int rc = launch_linear_bf16_glds(a16, K, W, ldw, static_cast<float*>(out->buf), ldc, M, N, K, nb, 0, act, bias, nullptr, 0, 1, st,
                                         m_dev);
  if (rc == 0) out->src = C;
  return rc;
}
// A linear whose output rows are whole model rows, followed by the norm that always comes next on the decode path:
// C = epilogue(A·W^T), ne.Y = norm(C).  fp32 with split-K: the reduction kernel applies the norm while it holds the
// finished row (one launch instead of two or three); otherwise the linear and the norm kernels run one after another.
static int dec_linear_norm(bool bf16, void* abf, const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                           int64_t M, const int64_t* m_dev, int N, int K, int epi, const float* bias, const float* res,
                           int64_t ldr, float* skw, hipStream_t st, const NormEpilogue& ne, StreamK* sk = nullptr,
                           Bf16Image* img = nullptr, const Bf16Image* a_img = nullptr) {
  // img: receives the bf16 image of the norm's output; a_img (optional): where THIS linear's operand already lies as bf16
  // (an attention context or a ReLU output emitted as bf16 by its producer) — otherwise img is looked at for the operand too
  if (M == 0) return GDR_OK;
  static const bool fuse_on = [] {
    const char* e = getenv("GDR_DECODE_FUSE_NORM");  // A/B knob: 0 = always separate launches
    return e ? atoi(e) != 0 : true;
  }();
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  // measured in round 2: 1 x 100 beams 8.62 -> 8.36 ms, 16 x 10 beams 9.88 -> 9.55, 64 x 10 beams 18.00 -> 17.60
  if (fuse_on && !bf16 && tiles < 192 && M <= 1536 && K % 32 == 0 && K / 32 >= 4 && ldc == N && ne.ldy == N) {
    const bool nb = epi == GDR_EPI_BIAS || epi == GDR_EPI_BIAS_RELU || epi == GDR_EPI_BIAS_RESIDUAL;
    const bool nr = epi == GDR_EPI_RESIDUAL || epi == GDR_EPI_BIAS_RESIDUAL;
    const int act = (epi == GDR_EPI_RELU || epi == GDR_EPI_BIAS_RELU) ? 1 : 0;
    GDR_CHECK_ARG(epi != GDR_EPI_BIAS_GELU && (!nb || bias) && (!nr || res), "decode: bad epilogue for a fused norm");
    const int rc = launch_linear_f32_small(A, lda, W, ldw, C, ldc, M, N, K, nb, nr, act, bias, res, ldr, skw, SPLITK_WS_BYTES, st,
                                           m_dev, &ne, nullptr, sk ? sk->live : nullptr);
    if (rc <= 0) return rc;
  }
  if (int rc = dec_linear(bf16, abf, A, lda, W, ldw, C, ldc, M, m_dev, N, K, epi, bias, res, ldr, skw, st, sk, a_img ? a_img : img))
    return rc;
  // bf16 mode: the last norm of this call also leaves its output rounded to bf16 for the linear that reads it next
  void* y16 = (bf16 && img && img->buf && ne.ldy == N) ? img->buf : nullptr;
  if (img) img->src = y16 ? ne.Y : nullptr;
  if (ne.kind == 1) {
    if (y16 && ne.y16_only)  // nobody reads these rows as fp32 (bf16 mode: every consumer is a linear): write the bf16 image alone
      return m_dev ? launch_rmsnorm_bf16_dev(C, ne.w1, y16, m_dev, M, N, ne.eps, st) : launch_rmsnorm_bf16(C, ne.w1, y16, M, N, ne.eps, st);
    return m_dev ? launch_rmsnorm_dev(C, ne.w1, ne.Y, m_dev, M, N, ne.eps, st, y16)
                 : launch_rmsnorm(C, ne.w1, ne.Y, M, N, ne.eps, nullptr, 1, st, y16);
  }
  if (ne.kind == 3) {  // norm1 -> + addv -> norm2 with the row held in registers (one launch, no intermediate pass)
    const int rc2 = launch_layernorm2(C, ne.w1, ne.b1, ne.w2, ne.b2, ne.addv, ne.Y, m_dev, M, N, ne.eps, st, y16);
    if (rc2 <= 0) return rc2;
  }
  // LayerNorm(s): the second one reads the first one's output through Y
  float* y1 = ne.kind == 3 ? C : ne.Y;  // kind 3: norm1 may overwrite C (the pre-norm rows are not needed again)
  if (int rc = m_dev ? launch_layernorm_dev(C, ne.w1, ne.b1, y1, m_dev, M, N, ne.eps, nullptr, st, ne.kind == 3 ? nullptr : y16)
                     : launch_layernorm(C, ne.w1, ne.b1, y1, M, N, ne.eps, nullptr, st, ne.kind == 3 ? nullptr : y16))
    return rc;
  if (ne.kind == 3)
    return m_dev ? launch_layernorm_dev(y1, ne.w2, ne.b2, ne.Y, m_dev, M, N, ne.eps, ne.addv, st, y16)
                 : launch_layernorm(y1, ne.w2, ne.b2, ne.Y, M, N, ne.eps, ne.addv, st, y16);
  return GDR_OK;
}
// element offset into a linear weight (fp32 or bf16 storage)
static const float* w_at(const float* W, size_t elems, bool bf16) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(W) + elems * (bf16 ? 2 : 4));
}

// GDR_DECODE_FUSED, read once: bit mask 1 self-attention, 2 cross-attention, 4 feed-forward sub-block fused (decode_fused.hip).
// OFF by default: measured slower than the per-phase launches at every decode shape (profiles/r06_fused_decode_ab.txt: 64 x 10 beams
// 11.97 -> 14.55 ms, 1 x 100 6.48 -> 7.54 with all three on; the self-attention form alone is neutral at 640 rows).  Kept exact and
// tested (test_gpu_decode_knobs.py).  The slab scratch is only part of the workspace when the knob is on.
static int decode_fused_mask() {
  static const int mask = [] {
    const char* e = getenv("GDR_DECODE_FUSED");
    return e ? atoi(e) : 0;
  }();
  return mask;
}

// ------------------------------------------------------------------------------------------ model workspace
struct GenWs {
  size_t beam, dcache, acache, crosskv, xd, xa, nx, ctx, qc, ff, tmp, A, hl, splitk, ctx2, ff2, splitk2, qkv_c, abf, abf2, img1, img2, c16a, c16b, f16a, f16b, fslab, total;
};

static GenWs gen_ws(const GdrT5DecoderWeights& w, const BeamDims& bd, int L) {
  GenWs g{};
  const GdrT5Dims& dm = w.dims;
  const size_t rows = (size_t)bd.B * bd.R, d = dm.d_model, inner = (size_t)dm.num_heads * dm.d_kv;
  const size_t tmax = bd.maxlen;
  size_t o = 0;
  g.beam = carve(o, beam_layout(bd, nullptr, nullptr));
  g.dcache = carve(o, 4 * (size_t)dm.num_layers * tmax * rows * 3 * inner);
  g.acache = carve(o, 4 * (size_t)w.adaptor_layers * tmax * rows * 3 * d);
  g.crosskv = carve(o, 4 * (size_t)dm.num_layers * bd.B * L * 2 * inner);
  g.xd = carve(o, 4 * rows * d);
  g.xa = carve(o, 4 * rows * d);
  g.nx = carve(o, 4 * rows * d);
  g.ctx = carve(o, 4 * rows * (inner > d ? inner : d));
  g.qc = carve(o, 4 * rows * inner);
  const size_t ffw = (size_t)(dm.d_ff > w.adaptor_ff ? dm.d_ff : w.adaptor_ff);
  g.ff = carve(o, 4 * rows * ffw);
  g.tmp = carve(o, 4 * rows * d);
  g.A = carve(o, 4 * rows * (size_t)(bd.V + 1) * d);
  g.hl = carve(o, 4 * rows * d);
  g.splitk = carve(o, SPLITK_WS_BYTES);
  g.ctx2 = carve(o, 4 * rows * d);                       // adaptor chain runs on a side stream: own scratch
  g.ff2 = carve(o, 4 * rows * (size_t)w.adaptor_ff);
  g.splitk2 = carve(o, SPLITK_WS_BYTES);
  g.qkv_c = carve(o, 4 * rows * 3 * d);                 // prefix-table mode: (q,k,v) of the compacted rows
  const size_t abf_main = rows * ffw > (size_t)bd.B * L * d ? rows * ffw : (size_t)bd.B * L * d;
  g.abf = carve(o, 2 * abf_main);                       // bf16 mode: the rounded activation operand of a linear (main stream)
  g.abf2 = carve(o, 2 * abf_main);                      //            ... of the side stream (adaptor chain, cross K/V projections)
  g.img1 = carve(o, 2 * rows * d);                      // bf16 mode: the bf16 image of the last normed rows of either chain (Bf16Image)
  g.img2 = carve(o, 2 * rows * d);
  g.c16a = carve(o, 2 * rows * (inner > d ? inner : d));   // bf16 mode: attention contexts / ReLU outputs emitted as bf16 by their
  g.c16b = carve(o, 2 * rows * d);                         // producers (a: decoder chain, b: adaptor chain) — the operand of
  g.f16a = carve(o, 2 * rows * ffw);                       // the linear behind them, which then needs no cast launch
  g.f16b = carve(o, 2 * rows * (size_t)w.adaptor_ff);
  // fused sub-blocks (decode_fused.hip, <= 1 024 rows): the partial slabs of one sub-block, per chain
  const bool fused = decode_fused_mask() != 0 && decode_fused_rt((int64_t)rows, (int)d, (int)inner, dm.d_kv) != 0;
  g.fslab = carve(o, fused ? decode_fused_slab_bytes((int64_t)rows, (int)d, dm.d_ff, dm.num_heads) : 0);
  g.total = o;
  return g;
}

}  // namespace gdr

extern "C" size_t gdr_t5_generate_workspace_bytes(const GdrT5DecoderWeights* w, int B, int L, int num_beams,
                                                  int max_length) {
  if (!w || B <= 0 || L <= 0 || num_beams <= 0 || max_length < 2) return 0;
  (void)gdr::done_words();  // pinned host words of the early exit: allocated here, never inside a caller's stream capture
  gdr::BeamDims bd{B, num_beams, w->out_vocab, w->dims.vocab_size, max_length, num_beams, 1.0, nullptr, nullptr, 0};
  return gdr::gen_ws(*w, bd, L).total;
}

namespace gdr {
static int generate_impl(const GdrT5DecoderWeights* w, const float* enc_hidden, const int64_t* enc_mask, int B, int L,
                         int num_beams, int max_length, double length_penalty, int num_return_sequences,
                         const GdrTrie* trie, const GdrPrefixTable* ptab, int64_t* out_ids, int32_t* out_len,
                         double* out_scores, float* step_scores, int32_t* step_tokens, void* workspace,
                         size_t workspace_bytes, bool bf16, hipStream_t stream) {
  GDR_CHECK_ARG(w && enc_hidden && enc_mask && out_ids && out_len && out_scores && workspace, "generate: null pointer");
  const GdrT5Dims& dm = w->dims;
  GDR_CHECK_ARG(!trie || (trie->child && trie->eos_ok && trie->n_nodes > 0), "generate: bad trie");
  GDR_CHECK_ARG(!trie || trie->V == w->out_vocab, "generate: trie built for V=%d but the head has output_vocab_size=%d",
                trie ? trie->V : 0, w->out_vocab);
  GDR_CHECK_ARG(!ptab || (ptab->child && ptab->kv && ptab->W && ptab->n_nodes > 0 && ptab->n_table > 0 &&
                          ptab->n_table <= ptab->n_nodes && ptab->V == w->out_vocab),
                "generate: bad prefix table (V=%d, head output_vocab_size=%d)", ptab ? ptab->V : 0, w->out_vocab);
  GDR_CHECK_ARG(!ptab || !trie || (trie->child == ptab->child && trie->n_nodes == ptab->n_nodes),
                "generate: the trie constraint and the prefix table must be built over the same trie arrays");
  GDR_CHECK_ARG(!ptab || (dm.d_model % 32 == 0 && w->adaptor_ff % 32 == 0), "generate: prefix-table mode needs d %% 32 == 0");
  if (ptab && ptab->complete_levels > 1) {
    // complete_levels = c claims that levels 0 .. c-1 of the table hold ALL V^s prefixes of their length: at steps s < c the
    // miss-row chain is not even enqueued, so an overstated c would read head rows nobody computed.  It can be checked
    // against the table's own size: sum_{s<c} V^s nodes must fit in n_table, and no level lies beyond the decode length.
    GDR_CHECK_ARG(ptab->complete_levels <= w->max_out_len - 1, "generate: prefix table complete_levels=%d exceeds the %d decode steps",
                  ptab->complete_levels, w->max_out_len - 1);
    int64_t need = 0, pw = 1;
    for (int s = 0; s < ptab->complete_levels; ++s) {
      need += pw;
      pw *= ptab->V;
      if (need > ptab->n_table) break;
    }
    GDR_CHECK_ARG(need <= ptab->n_table, "generate: prefix table claims %d complete levels (>= %lld nodes) but holds %d nodes",
                  ptab->complete_levels, (long long)need, ptab->n_table);
  }
  BeamDims bd{B, num_beams, w->out_vocab, dm.vocab_size, max_length, num_return_sequences, length_penalty,
              trie ? trie->child : (ptab ? ptab->child : nullptr), trie ? trie->eos_ok : nullptr,
              trie ? trie->n_nodes : (ptab ? ptab->n_nodes : 0)};
  int rc = check_beam_dims(bd, max_length);
  if (rc) return rc;
  GDR_CHECK_ARG(L >= 1 && L <= 128, "generate: L=%d must be in [1,128]", L);
  GDR_CHECK_ARG(max_length <= w->max_out_len, "generate: max_length=%d > max_output_length=%d of the head", max_length,
                w->max_out_len);
  GDR_CHECK_ARG(dm.d_model % 4 == 0 && dm.d_kv % 4 == 0 && dm.d_model % w->adaptor_nhead == 0 &&
                    (dm.d_model / w->adaptor_nhead) % 4 == 0,
                "generate: unsupported dims");
  GDR_CHECK_ARG(w->dec_embed && w->self_rel_bias && w->cross_rel_bias && w->final_ln && w->layers && w->alayers &&
                    w->head_w && w->head_e,
                "generate: null weight pointer");
  const GenWs g = gen_ws(*w, bd, L);
  if (workspace_bytes < g.total) {
    set_error("generate: workspace %zu < required %zu", workspace_bytes, g.total);
    return GDR_ENOSPC;
  }
  GDR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "generate: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  BeamBufs bb{};
  beam_layout(bd, base + g.beam, &bb);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  float *dcache = F(g.dcache), *acache = F(g.acache), *crosskv = F(g.crosskv), *xd = F(g.xd), *xa = F(g.xa),
        *nx = F(g.nx), *ctx = F(g.ctx), *qc = F(g.qc), *ff = F(g.ff), *tmp = F(g.tmp), *A = F(g.A), *hl = F(g.hl),
        *skw = F(g.splitk), *ctx2 = F(g.ctx2), *ff2 = F(g.ff2), *skw2 = F(g.splitk2), *qkv_c = F(g.qkv_c), *fslab = F(g.fslab);
  void *abf = base + g.abf, *abf2 = base + g.abf2;
  Bf16Image img1{nullptr, bf16 ? base + g.img1 : nullptr}, img2{nullptr, bf16 ? base + g.img2 : nullptr};
  Bf16Image c16a{nullptr, bf16 ? base + g.c16a : nullptr}, c16b{nullptr, bf16 ? base + g.c16b : nullptr};
  Bf16Image f16a{nullptr, bf16 ? base + g.f16a : nullptr}, f16b{nullptr, bf16 ? base + g.f16b : nullptr};
  // the bf16-mode attention kernels write the context's bf16 image only (d_kv % 4 == 0 rows of 8-byte pieces)
  auto ctx_to = [&](AttnArgs& at_, Bf16Image& im, float* fp32_ctx) {
    if (bf16 && im.buf) at_.out = nullptr, at_.out_bf16 = im.buf, im.src = fp32_ctx;
    else at_.out = fp32_ctx;
    at_.live = bb.live_gate;  // a step behind the last query's end: the attention launch exits at once
  };
  GDR_CHECK_ARG(!bf16 || (dm.d_model % 8 == 0 && dm.d_ff % 8 == 0 && (dm.num_heads * dm.d_kv) % 8 == 0 && w->adaptor_ff % 8 == 0),
                "generate(bf16): dims must be multiples of 8");
  const int d = dm.d_model, H = dm.num_heads, dk = dm.d_kv, inner = H * dk;
  const int rows = B * num_beams, V1 = bd.V + 1;
  const int aH = w->adaptor_nhead, ahd = d / aH, aff = w->adaptor_ff;
  const size_t dslab = (size_t)rows * 3 * inner;  // one position of one decoder layer's cache
  const size_t aslab = (size_t)rows * 3 * d;
  const size_t dlayer = (size_t)max_length * dslab, alayer = (size_t)max_length * aslab;
  const size_t ckv_layer = (size_t)B * L * 2 * inner;

#define LIN(A_, lda_, W_, ldw_, C_, ldc_, M_, N_, K_, epi_, bias_, res_, ldr_) \
  dec_linear(bf16, abf, A_, lda_, W_, ldw_, C_, ldc_, M_, nullptr, N_, K_, epi_, bias_, res_, ldr_, skw, stream, &sk1, &img1)
#define LIN2(A_, lda_, W_, ldw_, C_, ldc_, M_, N_, K_, epi_, bias_, res_, ldr_) \
  dec_linear(bf16, abf2, A_, lda_, W_, ldw_, C_, ldc_, M_, nullptr, N_, K_, epi_, bias_, res_, ldr_, skw2, as, &sk2, &img2)
#define LINN(A_, lda_, W_, ldw_, C_, ldc_, M_, N_, K_, epi_, bias_, res_, ldr_, ne_) \
  dec_linear_norm(bf16, abf, A_, lda_, W_, ldw_, C_, ldc_, M_, nullptr, N_, K_, epi_, bias_, res_, ldr_, skw, stream, ne_, &sk1, &img1)
#define LIN2N(A_, lda_, W_, ldw_, C_, ldc_, M_, md_, N_, K_, epi_, bias_, res_, ldr_, ne_) \
  dec_linear_norm(bf16, abf2, A_, lda_, W_, ldw_, C_, ldc_, M_, md_, N_, K_, epi_, bias_, res_, ldr_, skw2, as, ne_, &sk2, &img2)
  // the decoder's normed rows `nx` feed linears only; in the bf16 mode those read the bf16 image, so the fp32 copy is not written
  // (the final norm's rows `hl` go into the fp32 head dot: kept)
  auto rms = [&](const float* wgt, float* y) {
    return NormEpilogue{1, wgt, nullptr, nullptr, nullptr, nullptr, dm.eps, y, (int64_t)dm.d_model, (bf16 && y != hl) ? 1 : 0};
  };
  auto ln = [&](const float* w1, const float* b1, float* y) {
    return NormEpilogue{2, w1, b1, nullptr, nullptr, nullptr, w->adaptor_eps, y, (int64_t)dm.d_model};
  };
  auto ln2 = [&](const GdrAdaptorLayer& al, float* y) {  // norm1 -> + cross_const -> norm2
    return NormEpilogue{3, al.ln1_w, al.ln1_b, al.ln2_w, al.ln2_b, al.cross_const, w->adaptor_eps, y, (int64_t)dm.d_model};
  };
#define GDR_TRY(x)        \
  do {                    \
    if ((rc = (x))) return rc; \
  } while (0)

  static const bool dedup0 = [] {
    const char* e = getenv("GDR_DECODE_DEDUP0");  // exact A/B knob: 0 = run step 0 on all B*R (identical) beam rows
    return e ? atoi(e) != 0 : true;
  }();
  GDR_TRY(beam_begin(bb, bd, stream, dedup0));
  // stream-K hand-off scratch of the two chains' big linears (gemm_f32.hip) inside their split-K regions — a launch uses one
  // or the other; both flag blocks are zeroed here, on the caller's stream, before the adaptor stream is first forked
  StreamK sk1{skw, reinterpret_cast<int32_t*>(base + g.splitk + STREAMK_PART_BYTES), 0};
  StreamK sk2{skw2, reinterpret_cast<int32_t*>(base + g.splitk2 + STREAMK_PART_BYTES), 0};
  if (hipMemsetAsync(sk1.flag, 0, 512 * sizeof(int32_t), stream) != hipSuccess ||
      hipMemsetAsync(sk2.flag, 0, 512 * sizeof(int32_t), stream) != hipSuccess) {
    set_error("generate: memset of the stream-K flags failed");
    return GDR_EHIP;
  }
  // early exit (see done_words): not while a caller wants the per-step trace — its rows of the steps not run would be undefined
  static const bool early_on = [] {
    const char* e = getenv("GDR_DECODE_EARLY_EXIT");  // A/B knob: 0 = always run all max_length - 1 steps
    return e ? atoi(e) != 0 : true;
  }();
  int32_t* done_word = nullptr;
  int32_t my_epoch = 0;
  if (early_on && !step_scores && !step_tokens) {
    if (int32_t* words = done_words()) {
      my_epoch = g_done_epoch.fetch_add(1) + 1;  // never 0
      done_word = words + (my_epoch & 63);
      bb.all_done_host = done_word, bb.done_epoch = my_epoch;  // travels with every beam_update launch
    }
    sk1.live = sk2.live = bb.live;  // the device-side half: steps already enqueued when the last query finishes skip their work
    bb.live_gate = bb.live;         // ... and so do their attention, reduction, head and beam bookkeeping launches
  }
  SideLease lease;
  struct { bool ok; hipStream_t s; hipEvent_t fork, join; } ss{lease.ss != nullptr, lease.ss ? lease.ss->s : nullptr,
                                                            lease.ss ? lease.ss->fork : nullptr, lease.ss ? lease.ss->join : nullptr};
  // cross-attention K/V once per query and layer (modeling_t5.py:365-368 recomputes them per beam row per step).  They
  // depend on the encoder states only: with a side stream they are projected THERE (the adaptor chain's scratch), layer by
  // layer, while the main stream already runs step 0 — a chain of small latency-bound kernels over one row per query that
  // leaves most of the chip idle — and layer l's cross-attention of step 0 waits for layer l's event.
  const bool ckv_side = ss.ok && dm.num_layers <= SIDE_MAX_LAYERS;
  if (ckv_side) {
    if (hipEventRecord(ss.fork, stream) != hipSuccess || hipStreamWaitEvent(ss.s, ss.fork, 0) != hipSuccess) {
      set_error("generate: fork to the side stream failed");
      return GDR_EHIP;
    }
    for (int l = 0; l < dm.num_layers; ++l) {
      GDR_TRY(dec_linear(bf16, abf2, enc_hidden, d, w->layers[l].wkv_c, d, crosskv + l * ckv_layer, 2 * inner, (int64_t)B * L, nullptr,
                         2 * inner, d, GDR_EPI_NONE, nullptr, nullptr, 0, skw2, ss.s, &sk2));
      if (hipEventRecord(lease.ss->ckv[l], ss.s) != hipSuccess) {
        set_error("generate: event record on the side stream failed");
        return GDR_EHIP;
      }
    }
  } else {
    for (int l = 0; l < dm.num_layers; ++l)
      GDR_TRY(LIN(enc_hidden, d, w->layers[l].wkv_c, d, crosskv + l * ckv_layer, 2 * inner, (int64_t)B * L,
                                2 * inner, d, GDR_EPI_NONE, nullptr, nullptr, 0));
  }

  static const int fused_mask = decode_fused_mask();
  static const bool slab_q_on = [] {
    const char* e = getenv("GDR_DECODE_SLAB_Q");  // A/B knob: 0 = reduce the cross-attention q projection in its own launch
    return e ? atoi(e) != 0 : true;
  }();
  const BucketLut lut_uni = make_bucket_lut(dm.rel_buckets, dm.rel_max_distance);
  const BucketLut lut_bi = make_bucket_lut(dm.rel_buckets / 2, dm.rel_max_distance);
  int cur = 0;
  for (int s = 0; s + 1 < max_length; ++s) {  // position s, cur_len = s + 1 (generation_utils.py:676)
    if (done_word && s > 0 && __atomic_load_n(done_word, __ATOMIC_RELAXED) == my_epoch) {  // :836-838 all(done)
      g_early_exits.fetch_add(1);
      break;
    }
    hipStream_t as = ss.ok ? ss.s : stream;   // adaptor stream
    img1.src = img2.src = nullptr;            // the embedding kernels below rewrite nx / xa without a bf16 image
    // Step 0: the R beam rows of a query hold the same START token and the same encoder states, so their decoder /
    // adaptor / head outputs are identical rows (generation_utils.py:437-442 expands the encoder states, :663-668 starts
    // every beam but the first at -1e9): one row per query is computed (row b of the step-0 cache slots; every row's
    // ancestor at position 0 is b, beam_init_kernel) and beam_topk_kernel reads the query's logits for all of its beams.
    const int rows_s = (s == 0 && dedup0) ? B : rows;
    const int R_s = (s == 0 && dedup0) ? 1 : num_beams;
    hipLaunchKernelGGL(embed_rmsnorm_kernel, dim3((unsigned)((rows_s + 3) / 4)), dim3(256), 0, stream, w->dec_embed, bb.cur_tok, rows_s,
                       d / 4, dm.vocab_size, w->layers[0].ln_self, dm.eps, xd, nx, sk1.live);
    GDR_CHECK_LAUNCH("embed_rmsnorm_kernel");
    if (ss.ok) {
      if (hipEventRecord(ss.fork, stream) != hipSuccess || hipStreamWaitEvent(ss.s, ss.fork, 0) != hipSuccess) {
        set_error("generate: fork to the adaptor stream failed");
        return GDR_EHIP;
      }
    }
    if (!ptab) {
      GDR_TRY(launch_embed(w->dec_embed, bb.cur_tok, rows_s, d, dm.vocab_size, xa, as));
    }
    // ---------------- adaptor: post-LN nn.TransformerDecoder over decode_embeddings(ids) (modeling_t5.py:1615-1633)
    auto ad_layer_plain = [&](int l) -> int {
        const GdrAdaptorLayer& al = w->alayers[l];
        float* cache = acache + l * alayer;
        float* slot = cache + s * aslab;
        GDR_TRY(LIN2(xa, d, al.in_w, d, slot, 3 * d, rows_s, 3 * d, d, GDR_EPI_BIAS, al.in_b, nullptr, 0));
        AttnArgs at{};
        at.q = slot, at.k = cache + d, at.v = cache + 2 * d;
        ctx_to(at, c16b, ctx2);
        at.ldq = at.ldk = at.ldv = 3 * d, at.ldo = d;
        at.q_bstride = 1, at.k_bstride = 0, at.o_bstride = 1;
        at.B = rows_s, at.H = aH, at.dk = ahd, at.Lq = 1, at.Lk = s + 1, at.q_pos0 = s;
        at.scale = 1.0f / sqrtf((float)ahd);
        at.rel_bias = nullptr, at.bidirectional = 0, at.num_buckets = 0, at.lut = lut_uni;
        at.key_mask = nullptr, at.mask_bstride = 0, at.causal = 1, at.causal_neg_inf = 1;
        at.kv_rows = bb.kv_rows, at.kv_group = 1;
        GDR_TRY(launch_attention(at, as));
        // tmp = norm2(norm1(out_proj(ctx) + xa) + cross_const); xa = norm3(lin2(relu(lin1(tmp))) + tmp)
        GDR_TRY(dec_linear_norm(bf16, abf2, ctx2, d, al.out_w, d, tmp, d, rows_s, nullptr, d, d, GDR_EPI_BIAS_RESIDUAL, al.out_b, xa, d,
                                skw2, as, ln2(al, tmp), &sk2, &img2, &c16b));
        int r16 = bf16 ? dec_linear_out16(abf2, tmp, d, al.lin1_w, d, ff2, aff, rows_s, nullptr, aff, d, GDR_EPI_BIAS_RELU, al.lin1_b, as,
                                          &img2, &f16b) : 1;
        if (r16 < 0) return r16;
        if (r16 == 1) {
          f16b.src = nullptr;
          GDR_TRY(LIN2(tmp, d, al.lin1_w, d, ff2, aff, rows_s, aff, d, GDR_EPI_BIAS_RELU, al.lin1_b, nullptr, 0));
        }
        GDR_TRY(dec_linear_norm(bf16, abf2, ff2, aff, al.lin2_w, aff, xa, d, rows_s, nullptr, d, aff, GDR_EPI_BIAS_RESIDUAL, al.lin2_b, tmp,
                                d, skw2, as, ln(al.ln3_w, al.ln3_b, xa), &sk2, &img2, &f16b));
        return GDR_OK;
    };
    // prefix-table mode, steps at which every possible prefix is a table node (step 0: the root; deeper while the trie's
    // levels are complete, GdrPrefixTable.complete_levels): the adaptor chain and the head GEMM would run over zero rows
    // (34 launches per step that exit at once — 0.2 ms of the side queue and, at one query x 100 beams, of the host's time)
    const bool adaptor_idle = ptab != nullptr && s < (ptab->complete_levels > 1 ? ptab->complete_levels : 1);
    const int64_t* nm = bb.n_miss;
    if (ptab) {
      // ---------------- prefix-table mode: rows whose prefix is a table node take everything from the table; the rest
      // are compacted (row count *bb.n_miss lives on the device) and run the same chain + the head GEMM on `as`
      hipLaunchKernelGGL(prefix_plan_kernel, dim3(1), dim3(1024), 0, as, bb, rows_s, s + 1, cur, ptab->n_table);
      GDR_CHECK_LAUNCH("prefix_plan_kernel");
      hipLaunchKernelGGL(prefix_fill_kernel, dim3(rows_s), dim3(256), 0, as, bb, rows_s, s + 1, cur, ptab->kv,
                         (int64_t)ptab->n_table * 3 * d, acache + s * aslab, (int64_t)alayer, w->adaptor_layers, 3 * d);
      GDR_CHECK_LAUNCH("prefix_fill_kernel");
      if (!adaptor_idle) {  // idle steps: every row sits on a table node, nothing is compacted
        hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)((rows_s + 3) / 4)), dim3(256), 0, as, w->dec_embed, bb.cur_tok,
                           bb.miss_rows, nm, d / 4, dm.vocab_size, xa);
        GDR_CHECK_LAUNCH("embed_rows_kernel");
      }
    }
#define LIN2D(A_, lda_, W_, ldw_, C_, ldc_, N_, K_, epi_, bias_, res_, ldr_) \
  dec_linear(bf16, abf2, A_, lda_, W_, ldw_, C_, ldc_, rows_s, nm, N_, K_, epi_, bias_, res_, ldr_, skw2, as, &sk2, &img2)
    auto ad_layer_tab = [&](int l) -> int {
        const GdrAdaptorLayer& al = w->alayers[l];
        float* cache = acache + l * alayer;
        float* slot = cache + s * aslab;
        GDR_TRY(LIN2D(xa, d, al.in_w, d, qkv_c, 3 * d, 3 * d, d, GDR_EPI_BIAS, al.in_b, nullptr, 0));
        hipLaunchKernelGGL(scatter_slot_kernel, dim3((unsigned)((rows_s + 3) / 4)), dim3(256), 0, as, qkv_c, bb.miss_rows, nm,
                           3 * d / 4, d / 4, slot);
        GDR_CHECK_LAUNCH("scatter_slot_kernel");
        AttnArgs at{};
        at.q = qkv_c, at.k = cache + d, at.v = cache + 2 * d;
        ctx_to(at, c16b, ctx2);
        at.ldq = at.ldk = at.ldv = 3 * d, at.ldo = d;
        at.q_bstride = 1, at.k_bstride = 0, at.o_bstride = 1;
        at.B = rows_s, at.H = aH, at.dk = ahd, at.Lq = 1, at.Lk = s + 1, at.q_pos0 = s;
        at.scale = 1.0f / sqrtf((float)ahd);
        at.rel_bias = nullptr, at.bidirectional = 0, at.num_buckets = 0, at.lut = lut_uni;
        at.key_mask = nullptr, at.mask_bstride = 0, at.causal = 1, at.causal_neg_inf = 1;
        at.kv_rows = bb.kv_rows_c, at.kv_group = 1, at.b_count_dev = nm;
        GDR_TRY(launch_attention(at, as));
        GDR_TRY(dec_linear_norm(bf16, abf2, ctx2, d, al.out_w, d, tmp, d, rows_s, nm, d, d, GDR_EPI_BIAS_RESIDUAL, al.out_b, xa, d, skw2,
                                as, ln2(al, tmp), &sk2, &img2, &c16b));
        int r16 = bf16 ? dec_linear_out16(abf2, tmp, d, al.lin1_w, d, ff2, aff, rows_s, nm, aff, d, GDR_EPI_BIAS_RELU, al.lin1_b, as, &img2,
                                          &f16b) : 1;
        if (r16 < 0) return r16;
        if (r16 == 1) {
          f16b.src = nullptr;
          GDR_TRY(LIN2D(tmp, d, al.lin1_w, d, ff2, aff, aff, d, GDR_EPI_BIAS_RELU, al.lin1_b, nullptr, 0));
        }
        GDR_TRY(dec_linear_norm(bf16, abf2, ff2, aff, al.lin2_w, aff, xa, d, rows_s, nm, d, aff, GDR_EPI_BIAS_RESIDUAL, al.lin2_b, tmp, d,
                                skw2, as, ln(al.ln3_w, al.ln3_b, xa), &sk2, &img2, &f16b));
        return GDR_OK;
    };
    // ---------------- T5 decoder stack (modeling_t5.py:498-584, 685-821)
    auto dec_layer = [&](int l) -> int {
      const GdrT5DecLayer& ly = w->layers[l];
      float* cache = dcache + l * dlayer;
      float* slot = cache + s * dslab;
      // fused sub-blocks (decode_fused.hip): <= 1 024 rows, fp32, d_kv = 64 — (row panel, head / d_ff chunk) workgroups write
      // partial slabs, one reduction launch folds them with the residual and the next norm
      const int frt = (!bf16 && s + 1 <= 16 && dm.d_ff % 256 == 0) ? decode_fused_rt(rows_s, d, inner, dk) : 0;
      const bool f_sa = frt && (fused_mask & 1), f_ca = frt && (fused_mask & 2) && R_s > 1, f_ff = frt && (fused_mask & 4);
      FusedArgs fa{};
      fa.X = nx, fa.slabs = fslab, fa.M = rows_s, fa.d = d, fa.live = sk1.live, fa.H = H, fa.q_pos0 = s, fa.scale = 1.0f;
      fa.num_buckets = dm.rel_buckets;
      if (f_sa) {
        FusedArgs g1 = fa;
        g1.W1 = ly.wqkv, g1.w1_seg_stride = inner, g1.w1_slice_rows = 64, g1.W3 = ly.wo, g1.ld3 = inner, g1.n_slices = H;
        g1.slot = slot, g1.kbase = cache + inner, g1.vbase = cache + 2 * inner, g1.ld_kv = 3 * inner, g1.k_off = inner, g1.v_off = 2 * inner;
        g1.kv_rows = bb.kv_rows, g1.Lk = s + 1, g1.rel_bias = w->self_rel_bias, g1.lut = lut_uni;
        GDR_TRY(launch_decode_fused(FUSED_SA, g1, frt, 192, stream));
        GDR_TRY(launch_slab_reduce_norm(fslab, H, rows_s, d, xd, d, nullptr, xd, d, rms(ly.ln_cross, nx), sk1.live, stream));
      } else {
      // every later RMS norm rides on the reduction of the residual linear in front of it (dec_linear_norm)
      // (the first block's norm rides on the embedding launch: embed_rmsnorm_kernel)
      GDR_TRY(LIN(nx, d, ly.wqkv, d, slot, 3 * inner, rows_s, 3 * inner, d, GDR_EPI_NONE, nullptr, nullptr, 0));
      AttnArgs at{};
      at.q = slot, at.k = cache + inner, at.v = cache + 2 * inner;
      ctx_to(at, c16a, ctx);
      at.ldq = at.ldk = at.ldv = 3 * inner, at.ldo = inner;
      at.q_bstride = 1, at.k_bstride = 0, at.o_bstride = 1;
      at.B = rows_s, at.H = H, at.dk = dk, at.Lq = 1, at.Lk = s + 1, at.q_pos0 = s, at.scale = 1.0f;
      at.rel_bias = w->self_rel_bias, at.bidirectional = 0, at.num_buckets = dm.rel_buckets, at.lut = lut_uni;
      at.key_mask = nullptr, at.mask_bstride = 0, at.causal = 1, at.causal_neg_inf = 0;
      at.kv_rows = bb.kv_rows, at.kv_group = 1;
      GDR_TRY(launch_attention(at, stream));
      GDR_TRY(dec_linear_norm(bf16, abf, ctx, inner, ly.wo, inner, xd, d, rows_s, nullptr, d, inner, GDR_EPI_RESIDUAL, nullptr, xd, d, skw,
                              stream, rms(ly.ln_cross, nx), &sk1, &img1, &c16a));
      }
      // cross attention over the encoder states of the row's query
      if (f_ca) {
        FusedArgs g2 = fa;
        g2.W1 = ly.wq_c, g2.w1_seg_stride = 0, g2.w1_slice_rows = 64, g2.W3 = ly.wo_c, g2.ld3 = inner, g2.n_slices = H;
        g2.ck = crosskv + l * ckv_layer, g2.cv = crosskv + l * ckv_layer + inner, g2.ld_c = 2 * inner, g2.L = L, g2.R = R_s;
        g2.key_mask = enc_mask, g2.rel_bias = w->cross_rel_bias, g2.lut = lut_bi;
        if (s == 0 && ckv_side && hipStreamWaitEvent(stream, lease.ss->ckv[l], 0) != hipSuccess) {
          set_error("generate: wait for the cross K/V of layer %d failed", l);
          return GDR_EHIP;
        }
        GDR_TRY(launch_decode_fused(FUSED_CA, g2, frt, 64, stream));
        GDR_TRY(launch_slab_reduce_norm(fslab, H, rows_s, d, xd, d, nullptr, xd, d, rms(ly.ln_ff, nx), sk1.live, stream));
      } else {
      // the q projection's split-K slabs go to the attention kernel un-reduced (it sums them while it stages the beam rows_s'
      // queries): one dependent launch less per layer and step
      SlabRef qsl{nullptr, 1, 0};
      bool q_from_slabs = false;
      if (slab_q_on && !bf16 && R_s > 1 /* the Lq = 1 kernel of the de-duplicated step 0 takes finished q rows */ &&
          ((rows_s + 127) / 128) * ((inner + 127) / 128) < 192 && rows_s <= 1536 && d % 32 == 0 && d / 32 >= 4) {
        const int rc_ = launch_linear_f32_small(nx, d, ly.wq_c, d, qc, inner, rows_s, inner, d, 0, 0, 0, nullptr, nullptr, 0, skw,
                                                SPLITK_WS_BYTES, stream, nullptr, nullptr, &qsl, sk1.live);
        if (rc_ < 0) return rc_;
        q_from_slabs = rc_ == 0;
      }
      if (!q_from_slabs) GDR_TRY(LIN(nx, d, ly.wq_c, d, qc, inner, rows_s, inner, d, GDR_EPI_NONE, nullptr, nullptr, 0));
      AttnArgs ca{};
      const float* ckv = crosskv + l * ckv_layer;
      // the R beam rows_s of a query are consecutive and share its K/V: one workgroup per (query, head) stages K/V
      // once and serves all R rows_s (they all sit at decoder position s)
      ca.q = qc, ca.k = ckv, ca.v = ckv + inner;
      ctx_to(ca, c16a, ctx);
      ca.ldq = inner, ca.ldk = ca.ldv = 2 * inner, ca.ldo = inner;
      ca.q_bstride = R_s, ca.k_bstride = L, ca.o_bstride = R_s;
      ca.B = B, ca.H = H, ca.dk = dk, ca.Lq = R_s, ca.Lk = L, ca.q_pos0 = s, ca.scale = 1.0f, ca.q_same_pos = 1;
      ca.rel_bias = w->cross_rel_bias, ca.bidirectional = 1, ca.num_buckets = dm.rel_buckets, ca.lut = lut_bi;
      ca.key_mask = enc_mask, ca.mask_bstride = L, ca.causal = 0, ca.causal_neg_inf = 0;
      ca.kv_rows = nullptr, ca.kv_group = 1;
      if (q_from_slabs && qsl.S > 1) ca.q_part = qsl.part, ca.q_S = qsl.S, ca.q_tiles_n = qsl.tiles_n;
      if (s == 0 && ckv_side && hipStreamWaitEvent(stream, lease.ss->ckv[l], 0) != hipSuccess) {
        set_error("generate: wait for the cross K/V of layer %d failed", l);
        return GDR_EHIP;
      }
      GDR_TRY(launch_attention(ca, stream));
      GDR_TRY(dec_linear_norm(bf16, abf, ctx, inner, ly.wo_c, inner, xd, d, rows_s, nullptr, d, inner, GDR_EPI_RESIDUAL, nullptr, xd, d, skw,
                              stream, rms(ly.ln_ff, nx), &sk1, &img1, &c16a));
      }
      const bool last = l + 1 == dm.num_layers;  // the norm behind the block: the next block's first, or final_layer_norm
      if (f_ff) {
        FusedArgs g3 = fa;
        const int n1 = frt == 1 ? 128 : 256;
        g3.W1 = ly.wi, g3.w1_seg_stride = 0, g3.w1_slice_rows = n1, g3.W3 = ly.wo_ff, g3.ld3 = dm.d_ff, g3.n_slices = dm.d_ff / n1;
        GDR_TRY(launch_decode_fused(FUSED_FFN, g3, frt, n1, stream));
        GDR_TRY(launch_slab_reduce_norm(fslab, dm.d_ff / n1, rows_s, d, xd, d, nullptr, xd, d,
                                        rms(last ? w->final_ln : w->layers[l + 1].ln_self, last ? hl : nx), sk1.live, stream));
        return GDR_OK;
      }
      int r16 = bf16 ? dec_linear_out16(abf, nx, d, ly.wi, d, ff, dm.d_ff, rows_s, nullptr, dm.d_ff, d, GDR_EPI_RELU, nullptr, stream, &img1,
                                        &f16a) : 1;
      if (r16 < 0) return r16;
      if (r16 == 1) {
          f16a.src = nullptr;
          GDR_TRY(LIN(nx, d, ly.wi, d, ff, dm.d_ff, rows_s, dm.d_ff, d, GDR_EPI_RELU, nullptr, nullptr, 0));
        }
      GDR_TRY(dec_linear_norm(bf16, abf, ff, dm.d_ff, ly.wo_ff, dm.d_ff, xd, d, rows_s, nullptr, d, dm.d_ff, GDR_EPI_RESIDUAL, nullptr, xd, d,
                              skw, stream, rms(last ? w->final_ln : w->layers[l + 1].ln_self, last ? hl : nx), &sk1, &img1, &f16a));
      return GDR_OK;
    };
    // The two chains are enqueued layer by layer in turn: a host thread that first enqueued the whole adaptor chain left
    // the main stream idle for as long as those ~55 launches take to issue (measured: a fifth of a 100-beam step).
    for (int l = 0; l < dm.num_layers || l < w->adaptor_layers; ++l) {
      if (l < dm.num_layers) GDR_TRY(dec_layer(l));
      if (l < w->adaptor_layers && !adaptor_idle) GDR_TRY(ptab ? ad_layer_tab(l) : ad_layer_plain(l));
    }
    // bf16 mode from ~1 000 beam rows on: the head GEMM waits for the decoder stack and takes the dot with its hidden state in its own
    // epilogue (launch_linear_bf16_headdot) instead of writing rows x (V+1) x d floats for head_logits to read back (1.46 GB per step at
    // 15 360 rows).  At these sizes both chains fill the chip, so the adaptor chain gives up nothing by ending one GEMM earlier
    // (generate() 512 x 30 beams 44.5 -> 41.8 ms, 2 048 x 10 62.4 -> 58.9, 256 x 10 14.43 -> 14.15, 64 x 30 12.64 -> 12.5; below
    // HEAD_DOT_ROWS the GEMM stays on the adaptor chain, beside the decoder stack).  Same products, another fp32 summation tree.
    constexpr int HEAD_DOT_ROWS = 1024;
    static const bool head_dot_on = [] {
      const char* e = getenv("GDR_DECODE_FUSE_NORM");  // the A/B knob of the other fused consumer (0 = every linear writes its output and
      return e ? atoi(e) != 0 : true;                  // its consumer runs as a launch of its own); the fp32 norm fusion is off in bf16 mode
    }();
    const bool fuse_head = bf16 && ptab && !adaptor_idle && head_dot_on && rows_s >= HEAD_DOT_ROWS && d % 128 == 0;
    if (ptab && !adaptor_idle && !fuse_head)  // the head GEMM of the compacted rows belongs to the adaptor chain (it needs nothing from the decoder stack)
      GDR_TRY(LIN2D(xa, d, w_at(w->head_w, (size_t)s * V1 * d * d, bf16), d, A, (int64_t)V1 * d, V1 * d, d, GDR_EPI_NONE, nullptr,
                    nullptr, 0));
#undef LIN2D
    if (ss.ok) {
      if (hipEventRecord(ss.join, ss.s) != hipSuccess) {
        set_error("generate: adaptor stream join failed");
        return GDR_EHIP;
      }
    }
    if (ss.ok && hipStreamWaitEvent(stream, ss.join, 0) != hipSuccess) {
      set_error("generate: adaptor stream join failed");
      return GDR_EHIP;
    }
    // ---------------- head: last position, unmasked columns only (modeling_t5.py:1634-1646)
    const float* hw = w_at(w->head_w, (size_t)s * V1 * d * d, bf16);
    const float* he = w->head_e + (size_t)s * V1 * d;
    if (!ptab) {
      GDR_TRY(LIN(xa, d, hw, d, A, (int64_t)V1 * d, rows_s, V1 * d, d, GDR_EPI_NONE, nullptr, nullptr, 0));
      const int64_t items = (int64_t)rows_s * V1;
      hipLaunchKernelGGL(head_logits_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, hl, A, he, rows_s, V1, d,
                         1.0f / sqrtf((float)d), bb.logits, bb.live_gate);
      GDR_CHECK_LAUNCH("head_logits_kernel");
    } else {
      const float* partial = nullptr;
      if (fuse_head) {
        const void* a16 = (img2.buf && img2.src == xa) ? img2.buf : nullptr;  // the last adaptor norm's bf16 image of xa
        if (!a16) {
          GDR_TRY(launch_cast_f32_bf16(xa, abf2, (int64_t)rows_s * d, stream));
          a16 = abf2;
        }
        const int rc_ = launch_linear_bf16_headdot(a16, d, hw, d, rows_s, nm, V1 * d, d, hl, d, bb.miss_rows, he, d, 1.0f / sqrtf((float)d), A,
                                                   stream);
        if (rc_ < 0) return rc_;
        if (rc_ == 0)
          partial = A;
        else  // shape not served: the plain form, now behind the join
          GDR_TRY(dec_linear(bf16, abf2, xa, d, hw, d, A, (int64_t)V1 * d, rows_s, nm, V1 * d, d, GDR_EPI_NONE, nullptr, nullptr, 0, skw2,
                             stream, &sk2, &img2));
      }
      const int64_t items = (int64_t)rows_s * V1;
      hipLaunchKernelGGL(head_logits_table_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, hl, A, he, ptab->W,
                         bb.node[cur], bb.miss_index, ptab->n_table, rows_s, V1, d, 1.0f / sqrtf((float)d), bb.logits, bb.live_gate,
                         partial);
      GDR_CHECK_LAUNCH("head_logits_table_kernel");
    }
    GDR_TRY(beam_step(bb, bd, s, cur, step_scores, step_tokens, stream, s == 0 && dedup0));
    cur ^= 1;
  }
  return beam_end(bb, bd, max_length, cur, out_ids, out_len, out_scores, stream);
#undef GDR_TRY
#undef LIN
#undef LIN2
#undef LINN
#undef LIN2N
}
}  // namespace gdr

extern "C" int64_t gdr_t5_generate_early_exits(void) { return gdr::g_early_exits.load(); }

extern "C" int gdr_t5_generate_last_done_step(void) {
  int32_t* words = gdr::done_words();
  const int32_t e = gdr::g_done_epoch.load();
  if (!words || e == 0) return 0;
  return __atomic_load_n(words + (e & 63), __ATOMIC_ACQUIRE) == e ? __atomic_load_n(words + 64 + (e & 63), __ATOMIC_RELAXED) : 0;
}

extern "C" int gdr_t5_generate(const GdrT5DecoderWeights* w, const float* enc_hidden, const int64_t* enc_mask, int B,
                               int L, int num_beams, int max_length, double length_penalty,
                               int num_return_sequences, const GdrTrie* trie, const GdrPrefixTable* ptab,
                               int64_t* out_ids, int32_t* out_len, double* out_scores, float* step_scores,
                               int32_t* step_tokens, void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::generate_impl(w, enc_hidden, enc_mask, B, L, num_beams, max_length, length_penalty, num_return_sequences, trie,
                            ptab, out_ids, out_len, out_scores, step_scores, step_tokens, workspace, workspace_bytes, false,
                            static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_t5_generate_bf16(const GdrT5DecoderWeights* w, const float* enc_hidden, const int64_t* enc_mask, int B,
                                    int L, int num_beams, int max_length, double length_penalty,
                                    int num_return_sequences, const GdrTrie* trie, const GdrPrefixTable* ptab,
                                    int64_t* out_ids, int32_t* out_len, double* out_scores, float* step_scores,
                                    int32_t* step_tokens, void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::generate_impl(w, enc_hidden, enc_mask, B, L, num_beams, max_length, length_penalty, num_return_sequences, trie,
                            ptab, out_ids, out_len, out_scores, step_scores, step_tokens, workspace, workspace_bytes, true,
                            static_cast<hipStream_t>(stream_));
}

// ------------------------------------------------------------------------------------------------ prefix table build
// Level by level over the trie (nodes in breadth-first order, so a level is a contiguous row range and the GEMMs write
// straight into the table): the adaptor chain of every node of depth s attends over the node's ancestors, whose (q,k,v)
// rows of every layer are already in the table; the head GEMM with the level's slice of adaptor_linear and lm_head as
// bias gives W[node] = A + E.  Same arithmetic as the in-call chain of gdr_t5_generate, at corpus scale.
namespace gdr {
struct TabWs {
  size_t xa, tmp, ctx, ff, splitk, abf, total;
};
static TabWs tab_ws(const GdrT5DecoderWeights& w, int max_level_nodes) {
  TabWs t{};
  const size_t n = (size_t)max_level_nodes, d = w.dims.d_model;
  size_t o = 0;
  t.xa = carve(o, 4 * n * d);
  t.tmp = carve(o, 4 * n * d);
  t.ctx = carve(o, 4 * n * d);
  t.ff = carve(o, 4 * n * (size_t)w.adaptor_ff);
  t.splitk = carve(o, SPLITK_WS_BYTES);
  t.abf = carve(o, 2 * n * (size_t)(w.adaptor_ff > (int)d ? w.adaptor_ff : (int)d));
  t.total = o;
  return t;
}
}  // namespace gdr

extern "C" size_t gdr_t5_prefix_table_workspace_bytes(const GdrT5DecoderWeights* w, int max_level_nodes) {
  if (!w || max_level_nodes <= 0) return 0;
  return gdr::tab_ws(*w, max_level_nodes).total;
}

namespace gdr {
static int table_build_impl(const GdrT5DecoderWeights* w, int n_levels, const int32_t* level_off, const int64_t* node_tok,
                            const int32_t* node_anc, float* kv, float* W, void* workspace, size_t workspace_bytes, bool bf16,
                            hipStream_t stream) {
  GDR_CHECK_ARG(w && level_off && node_tok && node_anc && kv && W && workspace, "prefix_table_build: null pointer");
  const GdrT5Dims& dm = w->dims;
  GDR_CHECK_ARG(n_levels >= 1 && n_levels <= w->max_out_len - 1 && n_levels <= MAXLEN_CAP,
                "prefix_table_build: n_levels=%d must be in [1, max_output_length - 1 = %d]", n_levels, w->max_out_len - 1);
  GDR_CHECK_ARG(level_off[0] == 0 && level_off[1] == 1, "prefix_table_build: level 0 must be the root alone");
  const int d = dm.d_model, V1 = w->out_vocab + 1, aH = w->adaptor_nhead, ahd = d / aH, aff = w->adaptor_ff;
  GDR_CHECK_ARG(d % 4 == 0 && d % aH == 0 && ahd % 4 == 0, "prefix_table_build: unsupported dims");
  int max_n = 0;
  for (int s = 0; s < n_levels; ++s) {
    GDR_CHECK_ARG(level_off[s + 1] > level_off[s], "prefix_table_build: empty level %d", s);
    max_n = level_off[s + 1] - level_off[s] > max_n ? level_off[s + 1] - level_off[s] : max_n;
  }
  const TabWs t = tab_ws(*w, max_n);
  if (workspace_bytes < t.total) {
    set_error("prefix_table_build: workspace %zu < required %zu", workspace_bytes, t.total);
    return GDR_ENOSPC;
  }
  GDR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "prefix_table_build: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  float *xa = F(t.xa), *tmp = F(t.tmp), *ctx = F(t.ctx), *ff = F(t.ff), *skw = F(t.splitk);
  void* abf = base + t.abf;
  const int64_t n_table = level_off[n_levels];
  const size_t layer_stride = (size_t)n_table * 3 * d;
  const BucketLut lut = make_bucket_lut(dm.rel_buckets, dm.rel_max_distance);
  int rc;
#define TLIN(A_, lda_, W_, ldw_, C_, ldc_, M_, N_, K_, epi_, bias_, res_, ldr_) \
  dec_linear(bf16, abf, A_, lda_, W_, ldw_, C_, ldc_, M_, nullptr, N_, K_, epi_, bias_, res_, ldr_, skw, stream)
#define T_TRY(x)              \
  do {                        \
    if ((rc = (x))) return rc; \
  } while (0)
  size_t anc_off = 0;  // level s block of node_anc: [n_s][s + 1]
  for (int s = 0; s < n_levels; ++s) {
    const int lo = level_off[s], n = level_off[s + 1] - lo;
    T_TRY(launch_embed(w->dec_embed, node_tok + lo, n, d, dm.vocab_size, xa, stream));
    for (int l = 0; l < w->adaptor_layers; ++l) {
      const GdrAdaptorLayer& al = w->alayers[l];
      float* tab = kv + l * layer_stride;       // [n_table][3d] of this layer
      float* slot = tab + (size_t)lo * 3 * d;   // this level's rows
      T_TRY(TLIN(xa, d, al.in_w, d, slot, 3 * d, n, 3 * d, d, GDR_EPI_BIAS, al.in_b, nullptr, 0));
      AttnArgs at{};
      at.q = slot, at.k = tab + d, at.v = tab + 2 * d, at.out = ctx;
      at.ldq = at.ldk = at.ldv = 3 * d, at.ldo = d;
      at.q_bstride = 1, at.k_bstride = 0, at.o_bstride = 1;
      at.B = n, at.H = aH, at.dk = ahd, at.Lq = 1, at.Lk = s + 1, at.q_pos0 = s;
      at.scale = 1.0f / sqrtf((float)ahd);
      at.rel_bias = nullptr, at.bidirectional = 0, at.num_buckets = 0, at.lut = lut;
      at.key_mask = nullptr, at.mask_bstride = 0, at.causal = 1, at.causal_neg_inf = 1;
      at.kv_rows = node_anc + anc_off, at.kv_group = 1;
      T_TRY(launch_attention(at, stream));
      T_TRY(TLIN(ctx, d, al.out_w, d, tmp, d, n, d, d, GDR_EPI_BIAS_RESIDUAL, al.out_b, xa, d));
      T_TRY(launch_layernorm(tmp, al.ln1_w, al.ln1_b, xa, n, d, w->adaptor_eps, nullptr, stream));
      T_TRY(launch_layernorm(xa, al.ln2_w, al.ln2_b, tmp, n, d, w->adaptor_eps, al.cross_const, stream));
      T_TRY(TLIN(tmp, d, al.lin1_w, d, ff, aff, n, aff, d, GDR_EPI_BIAS_RELU, al.lin1_b, nullptr, 0));
      T_TRY(TLIN(ff, aff, al.lin2_w, aff, xa, d, n, d, aff, GDR_EPI_BIAS_RESIDUAL, al.lin2_b, tmp, d));
      T_TRY(launch_layernorm(xa, al.ln3_w, al.ln3_b, xa, n, d, w->adaptor_eps, nullptr, stream));
    }
    // W[node][c][i] = sum_k xa[node][k] * head_w[s][c][i][k] + head_e[s][c][i]      (modeling_t5.py:1634-1639)
    T_TRY(TLIN(xa, d, w_at(w->head_w, (size_t)s * V1 * d * d, bf16), d, W + (size_t)lo * V1 * d, (int64_t)V1 * d, n, V1 * d, d,
               GDR_EPI_BIAS, w->head_e + (size_t)s * V1 * d, nullptr, 0));
    anc_off += (size_t)n * (s + 1);
  }
#undef TLIN
#undef T_TRY
  return GDR_OK;
}
}  // namespace gdr

extern "C" int gdr_t5_prefix_table_build(const GdrT5DecoderWeights* w, int n_levels, const int32_t* level_off,
                                         const int64_t* node_tok, const int32_t* node_anc, float* kv, float* W,
                                         void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::table_build_impl(w, n_levels, level_off, node_tok, node_anc, kv, W, workspace, workspace_bytes, false,
                               static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_t5_prefix_table_build_bf16(const GdrT5DecoderWeights* w, int n_levels, const int32_t* level_off,
                                              const int64_t* node_tok, const int32_t* node_anc, float* kv, float* W,
                                              void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::table_build_impl(w, n_levels, level_off, node_tok, node_anc, kv, W, workspace, workspace_bytes, true,
                               static_cast<hipStream_t>(stream_));
}

extern "C" size_t gdr_beam_search_table_workspace_bytes(int B, int num_beams, int max_length, int out_vocab) {
  if (B <= 0 || num_beams <= 0 || max_length < 2) return 0;
  gdr::BeamDims bd{B, num_beams, out_vocab, out_vocab * max_length + 2, max_length, num_beams, 1.0, nullptr, nullptr, 0};
  return gdr::beam_layout(bd, nullptr, nullptr);
}

extern "C" int gdr_beam_search_table(const float* table, int B, int out_vocab, int num_beams, int max_length,
                                     double length_penalty, int num_return_sequences, const GdrTrie* trie,
                                     int64_t* out_ids, int32_t* out_len, double* out_scores, void* workspace,
                                     size_t workspace_bytes, void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GDR_CHECK_ARG(table && out_ids && out_len && out_scores && workspace, "beam_search_table: null pointer");
  BeamDims bd{B, num_beams, out_vocab, out_vocab * max_length + 2, max_length, num_return_sequences, length_penalty,
              trie ? trie->child : nullptr, trie ? trie->eos_ok : nullptr, trie ? trie->n_nodes : 0};
  GDR_CHECK_ARG(!trie || (trie->child && trie->eos_ok && trie->n_nodes > 0), "beam_search_table: bad trie");
  GDR_CHECK_ARG(!trie || trie->V == out_vocab, "beam_search_table: trie built for V=%d but out_vocab=%d",
                trie ? trie->V : 0, out_vocab);
  int rc = check_beam_dims(bd, max_length);
  if (rc) return rc;
  const size_t need = beam_layout(bd, nullptr, nullptr);
  if (workspace_bytes < need) {
    set_error("beam_search_table: workspace %zu < required %zu", workspace_bytes, need);
    return GDR_ENOSPC;
  }
  BeamBufs bb{};
  beam_layout(bd, static_cast<char*>(workspace), &bb);
  if ((rc = beam_begin(bb, bd, stream))) return rc;
  const int rows = B * num_beams, V1 = out_vocab + 1;
  int cur = 0;
  for (int s = 0; s + 1 < max_length; ++s) {
    const int n = rows * V1;
    hipLaunchKernelGGL(table_logits_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, table, bb.cur_tok, rows,
                       num_beams, out_vocab, bd.Vd, max_length, s, bb.logits);
    GDR_CHECK_LAUNCH("table_logits_kernel");
    if ((rc = beam_step(bb, bd, s, cur, nullptr, nullptr, stream))) return rc;
    cur ^= 1;
  }
  return beam_end(bb, bd, max_length, cur, out_ids, out_len, out_scores, stream);
}
