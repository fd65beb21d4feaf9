// BERT / DPR doc tower forward on gfx950 — replaces `EncoderModel.forward(passage=...)` of the reference
// (GDR_model/main_models.py:79-89 -> transformers/modeling_dpr.py:146-191 -> transformers/modeling_bert.py):
// embeddings (word + position + token_type, LayerNorm eps 1e-12), 12 post-LN blocks with scaled attention
// (scores / sqrt(dh) + (1-m)*-1e9, modeling_bert.py:260-265 with the mask of modeling_utils.py:271-272), erf-GeLU FFN,
// pooled = sequence_output[:, 0] (projection_dim = 0).  This is the producer of the corpus matrix D
// (Data_process/NQ_dataset/bert/bert.py:69-71) and of GDR's stage-2 re-encode path (main_models.py:1445-1455).
//
// Same kernels as the T5 path: the fp32 MFMA GEMM core with fused bias / residual / GeLU epilogues, the MFMA
// attention kernel (no bias table, scale dh^-0.5), LayerNorm with 16-byte loads and wave reductions.
#include "layers.h"

namespace gdr {

__global__ __launch_bounds__(256) void bert_embed_kernel(const float* __restrict__ word, const float* __restrict__ pos,
                                                         const float* __restrict__ type,
                                                         const int64_t* __restrict__ ids,
                                                         const int64_t* __restrict__ type_ids, int64_t rows, int L,
                                                         int d4, int vocab, int type_vocab, float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  int64_t tt = type_ids ? type_ids[row] : 0;
  tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
  const float4* w = reinterpret_cast<const float4*>(word) + id * d4;
  const float4* p = reinterpret_cast<const float4*>(pos) + (row % L) * d4;
  const float4* t = reinterpret_cast<const float4*>(type) + tt * d4;
  float4* o = reinterpret_cast<float4*>(out) + row * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) {
    const float4 a = w[c], b = p[c], e = t[c];
    float4 r;
    r.x = (a.x + b.x) + e.x, r.y = (a.y + b.y) + e.y, r.z = (a.z + b.z) + e.z, r.w = (a.w + b.w) + e.w;
    o[c] = r;
  }
}

__global__ __launch_bounds__(256) void take_rows_kernel(const float* __restrict__ x, int64_t n, int every, int d4,
                                                        float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float4* s = reinterpret_cast<const float4*>(x) + r * every * d4;
  float4* o = reinterpret_cast<float4*>(out) + r * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) o[c] = s[c];
}

constexpr size_t BERT_SCRATCH_BYTES = (size_t)112 << 20;  // split-K partial slabs of small batches / stream-K hand-off scratch
static_assert(STREAMK_BYTES <= BERT_SCRATCH_BYTES, "stream-K scratch must fit the GEMM scratch region");
struct BertWs {
  size_t x, t, qkv, ctx, ff, scratch, total;
};
static BertWs bert_ws(const GdrBertWeights& w, int64_t M) {
  BertWs b{};
  size_t o = 0;
  const size_t d = w.d_model;
  b.x = o, o += align_up(M * d * 4, 256);
  b.t = o, o += align_up(M * d * 4, 256);
  b.qkv = o, o += align_up(M * 3 * d * 4, 256);
  b.ctx = o, o += align_up(M * d * 4, 256);
  b.ff = o, o += align_up(M * (size_t)w.d_ff * 4, 256);
  b.scratch = o, o += BERT_SCRATCH_BYTES;
  b.total = o;
  return b;
}

}  // namespace gdr

extern "C" size_t gdr_bert_encoder_workspace_bytes(const GdrBertWeights* w, int B, int L) {
  if (!w || B <= 0 || L <= 0) return 0;
  return gdr::bert_ws(*w, (int64_t)B * L).total;
}

extern "C" int gdr_bert_encoder_forward(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                                        const int64_t* token_type_ids, int B, int L, float* out_hidden,
                                        float* out_pooled, void* workspace, size_t workspace_bytes, void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GDR_CHECK_ARG(w && ids && workspace && (out_hidden || out_pooled), "bert: null pointer");
  GDR_CHECK_ARG(B > 0 && L > 0 && L <= 128 && L <= w->max_pos, "bert: B=%d L=%d (L must be <= min(128, max_pos))", B, L);
  const int d = w->d_model, H = w->num_heads;
  GDR_CHECK_ARG(d % 4 == 0 && H > 0 && d % H == 0 && (d / H) % 4 == 0 && w->d_ff % 4 == 0, "bert: unsupported dims");
  GDR_CHECK_ARG(w->word_emb && w->pos_emb && w->type_emb && w->emb_ln_w && w->emb_ln_b && w->layers, "bert: null weight");
  const int64_t M = (int64_t)B * L;
  const BertWs ws = bert_ws(*w, M);
  if (workspace_bytes < ws.total) {
    set_error("bert: workspace %zu < required %zu", workspace_bytes, ws.total);
    return GDR_ENOSPC;
  }
  GDR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "bert: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  float* x = reinterpret_cast<float*>(base + ws.x);
  float* t = reinterpret_cast<float*>(base + ws.t);
  float* qkv = reinterpret_cast<float*>(base + ws.qkv);
  float* ctx = reinterpret_cast<float*>(base + ws.ctx);
  float* ff = reinterpret_cast<float*>(base + ws.ff);
  float* scr = reinterpret_cast<float*>(base + ws.scratch);
  int rc;
  StreamK sk{};  // the linears' stream-K tail (gemm_f32.hip): scratch inside the GEMM scratch region, flags zeroed per call
  sk.part = scr, sk.flag = reinterpret_cast<int32_t*>(base + ws.scratch + STREAMK_PART_BYTES), sk.epoch = 0;
  if (hipMemsetAsync(sk.flag, 0, 512 * sizeof(int32_t), stream) != hipSuccess) {
    set_error("bert: memset of the stream-K flags failed");
    return GDR_EHIP;
  }
  hipLaunchKernelGGL(bert_embed_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, w->word_emb, w->pos_emb,
                     w->type_emb, ids, token_type_ids, M, L, d / 4, w->vocab_size, w->type_vocab, t);
  GDR_CHECK_LAUNCH("bert_embed_kernel");
  if ((rc = launch_layernorm(t, w->emb_ln_w, w->emb_ln_b, x, M, d, w->eps, nullptr, stream))) return rc;

  AttnArgs at{};
  at.q = qkv, at.k = qkv + d, at.v = qkv + 2 * d, at.out = ctx;
  at.ldq = at.ldk = at.ldv = 3 * d, at.ldo = d;
  at.q_bstride = at.k_bstride = at.o_bstride = L;
  at.B = B, at.H = H, at.dk = d / H, at.Lq = L, at.Lk = L, at.q_pos0 = 0;
  at.scale = 1.0f / sqrtf((float)(d / H));
  at.rel_bias = nullptr, at.bidirectional = 1, at.num_buckets = 0;
  at.key_mask = mask, at.mask_bstride = L, at.causal = 0, at.causal_neg_inf = 0, at.kv_rows = nullptr, at.kv_group = 1;

  for (int i = 0; i < w->num_layers; ++i) {
    const GdrBertLayer& ly = w->layers[i];
    GDR_CHECK_ARG(ly.wqkv && ly.bqkv && ly.wo && ly.bo && ly.ln1_w && ly.ln1_b && ly.wi && ly.bi && ly.wo2 && ly.bo2 &&
                      ly.ln2_w && ly.ln2_b,
                  "bert: layer %d null weight", i);
    if ((rc = launch_linear_f32_ws(x, d, ly.wqkv, d, qkv, 3 * d, M, 3 * d, d, GDR_EPI_BIAS, ly.bqkv, nullptr, 0, scr, BERT_SCRATCH_BYTES,
                                   stream, &sk))) return rc;
    if ((rc = launch_attention(at, stream))) return rc;
    if ((rc = launch_linear_f32_ws(ctx, d, ly.wo, d, t, d, M, d, d, GDR_EPI_BIAS_RESIDUAL, ly.bo, x, d, scr, BERT_SCRATCH_BYTES, stream,
                                   &sk))) return rc;
    if ((rc = launch_layernorm(t, ly.ln1_w, ly.ln1_b, x, M, d, w->eps, nullptr, stream))) return rc;
    if ((rc = launch_linear_f32_ws(x, d, ly.wi, d, ff, w->d_ff, M, w->d_ff, d, GDR_EPI_BIAS_GELU, ly.bi, nullptr, 0, scr,
                                   BERT_SCRATCH_BYTES, stream, &sk)))
      return rc;
    if ((rc = launch_linear_f32_ws(ff, w->d_ff, ly.wo2, w->d_ff, t, d, M, d, w->d_ff, GDR_EPI_BIAS_RESIDUAL, ly.bo2, x, d, scr,
                                   BERT_SCRATCH_BYTES, stream, &sk)))
      return rc;
    float* dst = (i + 1 == w->num_layers && out_hidden) ? out_hidden : x;
    if ((rc = launch_layernorm(t, ly.ln2_w, ly.ln2_b, dst, M, d, w->eps, nullptr, stream))) return rc;
    if (i + 1 == w->num_layers && out_pooled) {  // pooled_output = sequence_output[:, 0, :]  (modeling_dpr.py:178-179)
      hipLaunchKernelGGL(take_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, dst, (int64_t)B, L, d / 4,
                         out_pooled);
      GDR_CHECK_LAUNCH("take_rows_kernel");
    }
  }
  return GDR_OK;
}

// ------------------------------------------------------------------------------------------------ ragged form (r06)
// The reference pads each batch of passages to its longest member (Data_process/NQ_dataset/bert/bert.py:69-71, padding=True) and
// BertModel computes every position; a PAD key contributes exp(-1e9 - max) = 0 to a live query's softmax and PAD rows never reach
// pooled = sequence_output[:, 0] (modeling_dpr.py:178-181).  As for the T5 encoder (encoder.hip, "ragged form"): the live token rows
// are packed front to back (launch_pack_plan), every linear / LayerNorm / residual runs over the packed rows (row count on the
// device), attention works per sequence on its own length, and a pooled-only call runs the last block's o / LN / FFN / LN on the B
// CLS rows alone.  fp32: kept rows are BIT-IDENTICAL to gdr_bert_encoder_forward (same k order per output element).
// bf16 precision mode (config C5's corpus is bf16): bf16 linear operands with fp32 accumulate, the producers (LayerNorm, attention,
// GeLU epilogue) emit the bf16 operand of the next linear; q, k, v are emitted as bf16 for the bf16-MFMA attention, whose 1/sqrt(dh)
// scale — a power of two at dh = 64 — is folded into the q rows of wqkv / bqkv by the caller (exact).
namespace gdr {

__global__ __launch_bounds__(256) void bert_embed_packed_kernel(const float* __restrict__ word, const float* __restrict__ pos,
                                                                const float* __restrict__ type, const int64_t* __restrict__ ids,
                                                                const int64_t* __restrict__ type_ids,
                                                                const int32_t* __restrict__ row_src,
                                                                const int64_t* __restrict__ rows_dev, int L, int d4, int vocab,
                                                                int type_vocab, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= *rows_dev) return;
  const int64_t row = row_src[r];  // b * L + position
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  int64_t tt = type_ids ? type_ids[row] : 0;
  tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
  const float4* w = reinterpret_cast<const float4*>(word) + id * d4;
  const float4* p = reinterpret_cast<const float4*>(pos) + (row % L) * d4;
  const float4* t = reinterpret_cast<const float4*>(type) + tt * d4;
  float4* o = reinterpret_cast<float4*>(out) + r * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) {
    const float4 a = w[c], b = p[c], e = t[c];
    float4 v;
    v.x = (a.x + b.x) + e.x, v.y = (a.y + b.y) + e.y, v.z = (a.z + b.z) + e.z, v.w = (a.w + b.w) + e.w;  // bert_embed_kernel's order
    o[c] = v;
  }
}

struct BertRagWs {
  size_t seq_len, seq_off, row_src, rows_total, ctx_cls, x_cls, t_cls, ff_cls, x16, pl_ff, total;
};
static BertRagWs bert_rag_ws(const GdrBertWeights& w, int B, int L) {
  BertRagWs r{};
  const size_t d = w.d_model, M = (size_t)B * L;
  size_t o = bert_ws(w, (int64_t)M).total;
  r.seq_len = o, o += align_up((size_t)B * 4, 256);
  r.seq_off = o, o += align_up((size_t)(B + 1) * 4, 256);
  r.row_src = o, o += align_up(M * 4, 256);
  r.rows_total = o, o += 256;
  r.ctx_cls = o, o += align_up((size_t)B * d * 4, 256);
  r.x_cls = o, o += align_up((size_t)B * d * 4, 256);
  r.t_cls = o, o += align_up((size_t)B * d * 4, 256);
  r.ff_cls = o, o += align_up((size_t)B * w.d_ff * 4, 256);
  r.x16 = o, o += align_up(M * 2 * d * 2, 256);  // bf16 mode: the bf16 image of the block input x; fp16 x 2 form: its plane rows [M, 2 d]
  // fp16 x 2 form: the plane rows of the GeLU output [M, 2 d_ff] — from 8 192 rows on the wi GEMM's epilogue writes them and the fp32 `ff`
  // buffer (the same size) is free to hold them; below that both exist
  r.pl_ff = o, o += align_up((M < 8192 ? M : 0) * 2 * (size_t)w.d_ff * 2, 256);
  r.total = o;
  return r;
}

// prec: 0 fp32, 1 bf16 precision mode, 2 the fp16 x 2 split form of the fp32 linears (r06, exploratory: fp32-level error on the fp16 MFMA
// path; weights as plane rows [N, 2 K] fp16; everything but the linears is the fp32 path's)
static int bert_ragged_impl(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask, const int64_t* token_type_ids, int B,
                            int L, float* out_hidden, float* out_pooled, int64_t live_rows_hint, void* workspace,
                            size_t workspace_bytes, int prec, hipStream_t stream) {
  const bool bf16 = prec == 1, f16s = prec == 2;
  if (B == 0) return GDR_OK;
  GDR_CHECK_ARG(w && ids && mask && workspace && (out_hidden || out_pooled), "bert_ragged: null pointer");
  GDR_CHECK_ARG(B > 0 && L > 0 && L <= 128 && L <= w->max_pos, "bert_ragged: B=%d L=%d (L must be <= min(128, max_pos))", B, L);
  const int d = w->d_model, H = w->num_heads;
  GDR_CHECK_ARG(d % 4 == 0 && H > 0 && d % H == 0 && (d / H) % 4 == 0 && w->d_ff % 4 == 0, "bert_ragged: unsupported dims");
  GDR_CHECK_ARG(w->word_emb && w->pos_emb && w->type_emb && w->emb_ln_w && w->emb_ln_b && w->layers, "bert_ragged: null weight");
  const int64_t M = (int64_t)B * L;
  const int dh = d / H, dff = w->d_ff;
  const BertWs ws = bert_ws(*w, M);
  const BertRagWs rw = bert_rag_ws(*w, B, L);
  if (workspace_bytes < rw.total) {
    set_error("bert_ragged: workspace %zu < required %zu", workspace_bytes, rw.total);
    return GDR_ENOSPC;
  }
  GDR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "bert_ragged: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  int rc;
  int32_t* seq_len = reinterpret_cast<int32_t*>(base + rw.seq_len);
  int32_t* seq_off = reinterpret_cast<int32_t*>(base + rw.seq_off);
  int32_t* row_src = reinterpret_cast<int32_t*>(base + rw.row_src);
  int64_t* rows_dev = reinterpret_cast<int64_t*>(base + rw.rows_total);
  if ((rc = launch_pack_plan(mask, B, L, seq_len, seq_off, row_src, rows_dev, stream))) return rc;
  const int64_t tiles = ((M + 127) / 128) * ((d + 127) / 128);
  const bool packs = dh == 64 && d % 32 == 0 && dff % 32 == 0;
  if (bf16 || f16s) {
    GDR_CHECK_ARG(packs && d % 128 == 0 && dff % 128 == 0,
                  "bert_ragged(bf16 / split): needs head width 64 and d, d_ff multiples of 128 (d=%d H=%d d_ff=%d)", d, H, dff);
  } else if (!packs || tiles < 192) {
    // small problem / other head size: the padded forward, then the rows the packed form would not have computed are zeroed
    float* full = out_hidden ? out_hidden : reinterpret_cast<float*>(base + ws.qkv);  // qkv is dead when the last LayerNorm runs
    if ((rc = gdr_bert_encoder_forward(w, ids, mask, token_type_ids, B, L, full, out_pooled, workspace, ws.total, stream))) return rc;
    return out_hidden ? launch_zero_dead_rows(out_hidden, seq_len, B, L, d, stream) : GDR_OK;
  }
  float* x = reinterpret_cast<float*>(base + ws.x);
  float* t = reinterpret_cast<float*>(base + ws.t);
  float* qkv = reinterpret_cast<float*>(base + ws.qkv);
  float* ctx = reinterpret_cast<float*>(base + ws.ctx);
  float* ff = reinterpret_cast<float*>(base + ws.ff);
  float* scr = reinterpret_cast<float*>(base + ws.scratch);
  float* ctx_cls = reinterpret_cast<float*>(base + rw.ctx_cls);
  float* x_cls = reinterpret_cast<float*>(base + rw.x_cls);
  float* t_cls = reinterpret_cast<float*>(base + rw.t_cls);
  float* ff_cls = reinterpret_cast<float*>(base + rw.ff_cls);
  void* x16 = base + rw.x16;
  StreamK sk{};
  sk.part = scr, sk.flag = reinterpret_cast<int32_t*>(base + ws.scratch + STREAMK_PART_BYTES), sk.epoch = 0;
  if (hipMemsetAsync(sk.flag, 0, 512 * sizeof(int32_t), stream) != hipSuccess) {
    set_error("bert_ragged: memset of the stream-K flags failed");
    return GDR_EHIP;
  }
  hipLaunchKernelGGL(bert_embed_packed_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, w->word_emb, w->pos_emb, w->type_emb,
                     ids, token_type_ids, row_src, rows_dev, L, d / 4, w->vocab_size, w->type_vocab, t);
  GDR_CHECK_LAUNCH("bert_embed_packed_kernel");
  void* pl_ff = M < 8192 ? static_cast<void*>(base + rw.pl_ff) : static_cast<void*>(ff);
  const int ld_d = 2 * d, ld_ff = 2 * dff;  // fp16 x 2 plane rows
  if ((rc = launch_layernorm_dev(t, w->emb_ln_w, w->emb_ln_b, x, rows_dev, M, d, w->eps, nullptr, stream, (bf16 || f16s) ? x16 : nullptr,
                                 f16s ? ld_d : 0)))
    return rc;

  AttnArgs at{};
  at.ldq = at.ldk = at.ldv = 3 * d, at.ldo = d;
  at.q_bstride = at.k_bstride = at.o_bstride = L;
  at.B = B, at.H = H, at.dk = dh, at.Lq = L, at.Lk = L, at.q_pos0 = 0;
  at.rel_bias = nullptr, at.bidirectional = 1, at.num_buckets = 0;
  at.key_mask = mask, at.mask_bstride = L, at.causal = 0, at.causal_neg_inf = 0, at.kv_rows = nullptr, at.kv_group = 1;
  at.seq_off = seq_off, at.seq_len = seq_len;
  if (bf16) {
    const __bf16* q16 = reinterpret_cast<const __bf16*>(qkv);
    at.q = reinterpret_cast<const float*>(q16), at.k = reinterpret_cast<const float*>(q16 + d), at.v = reinterpret_cast<const float*>(q16 + 2 * d);
    at.qkv_bf16 = 1, at.out = nullptr, at.out_bf16 = ctx;  // the context as bf16 [rows, d] in the fp32 ctx buffer
    at.scale = 1.0f;                                       // 1 / sqrt(dh) is folded into the q rows of wqkv / bqkv (gdr_hip.h)
  } else {
    at.q = qkv, at.k = qkv + d, at.v = qkv + 2 * d, at.out = ctx;
    at.scale = 1.0f / sqrtf((float)dh);
  }
  auto linear = [&](const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int N, int K, int epi, const float* bias,
                    const float* residual) -> int {
    return launch_linear_f32_dev(A, lda, W, K, C, ldc, M, rows_dev, N, K, epi, bias, residual, ldc, live_rows_hint, stream, &sk);
  };
  // bf16 linear: A bf16 [rows, K]; act 0 none / 2 gelu; out_bf16: the output is the next linear's bf16 operand
  auto lin16 = [&](const void* A, const float* W, float* C, int64_t ldc, int64_t rows, const int64_t* md, int N, int K, int act,
                   const float* bias, const float* residual, int out_bf16) -> int {
    ProfScope prof(PROF_LINEAR, 2.0 * (double)(md && live_rows_hint >= 0 ? live_rows_hint : rows) * (double)N * (double)K, stream);
    const int rc_ = launch_linear_bf16_glds(A, K, W, K, C, ldc, rows, N, K, bias != nullptr, residual != nullptr, act, bias, residual, ldc,
                                            out_bf16, stream, md);
    if (rc_ > 0) {
      set_error("bert_ragged(bf16): shape not served by the LDS-DMA linear");
      return GDR_EINVAL;
    }
    return rc_;
  };
  // fp16 x 2 split GEMM over plane rows P [rows, 2 K]; out_planes: the (activated) output leaves as plane rows [rows, 2 N]
  auto gsplit = [&](const void* P, const float* W, void* C, int64_t ldc, int64_t rows, const int64_t* md, int N, int K, int act,
                    const float* bias, const float* residual, int out_planes) -> int {
    ProfScope prof(PROF_LINEAR, 2.0 * (double)(md && live_rows_hint >= 0 ? live_rows_hint : rows) * (double)N * (double)K, stream);
    const int rc_ = launch_linear_bf16_glds(P, 2 * K, W, 2 * K, static_cast<float*>(C), ldc, rows, N, K, bias != nullptr, residual != nullptr, act,
                                            bias, residual, ldc, out_planes ? 3 : 0, stream, md, 2);
    if (rc_ > 0) {
      set_error("bert_ragged(split): shape not served by the LDS-DMA linear");
      return GDR_EINVAL;
    }
    return rc_;
  };
  const bool pooled_only = out_hidden == nullptr;
  for (int i = 0; i < w->num_layers; ++i) {
    const GdrBertLayer& ly = w->layers[i];
    GDR_CHECK_ARG(ly.wqkv && ly.bqkv && ly.wo && ly.bo && ly.ln1_w && ly.ln1_b && ly.wi && ly.bi && ly.wo2 && ly.bo2 && ly.ln2_w &&
                      ly.ln2_b,
                  "bert_ragged: layer %d null weight", i);
    const bool last = i + 1 == w->num_layers;
    if (f16s) {
      // x (fp32) and its plane rows x16 come from the LayerNorm in front; attention is the fp32 path's
      const bool big = M >= 8192;  // the plane epilogue lives in the 256-row tile kernel
      if ((rc = gsplit(x16, ly.wqkv, qkv, 3 * d, M, rows_dev, 3 * d, d, 0, ly.bqkv, nullptr, 0))) return rc;
      if ((rc = launch_attention(at, stream))) return rc;
      if (pooled_only && last) {
        if ((rc = launch_gather_rows(ctx, seq_off, B, d, ctx_cls, stream))) return rc;
        if ((rc = launch_gather_rows(x, seq_off, B, d, x_cls, stream))) return rc;
        if ((rc = launch_split_f32_bf16x3(ctx_cls, d, x16, ld_d, B, d, nullptr, stream, 1))) return rc;
        if ((rc = gsplit(x16, ly.wo, t_cls, d, B, nullptr, d, d, 0, ly.bo, x_cls, 0))) return rc;
        if ((rc = launch_layernorm(t_cls, ly.ln1_w, ly.ln1_b, x_cls, B, d, w->eps, nullptr, stream, x16, ld_d))) return rc;
        if ((rc = gsplit(x16, ly.wi, ff_cls, dff, B, nullptr, dff, d, 2, ly.bi, nullptr, 0))) return rc;
        if ((rc = launch_split_f32_bf16x3(ff_cls, dff, pl_ff, ld_ff, B, dff, nullptr, stream, 1))) return rc;
        if ((rc = gsplit(pl_ff, ly.wo2, t_cls, d, B, nullptr, d, dff, 0, ly.bo2, x_cls, 0))) return rc;
        return launch_layernorm(t_cls, ly.ln2_w, ly.ln2_b, out_pooled, B, d, w->eps, nullptr, stream);
      }
      if ((rc = launch_split_f32_bf16x3(ctx, d, x16, ld_d, M, d, rows_dev, stream, 1))) return rc;  // x16 is free: x's planes fed qkv already
      if ((rc = gsplit(x16, ly.wo, t, d, M, rows_dev, d, d, 0, ly.bo, x, 0))) return rc;
      if ((rc = launch_layernorm_dev(t, ly.ln1_w, ly.ln1_b, x, rows_dev, M, d, w->eps, nullptr, stream, x16, ld_d))) return rc;
      if (big) {
        if ((rc = gsplit(x16, ly.wi, pl_ff, ld_ff, M, rows_dev, dff, d, 2, ly.bi, nullptr, 1))) return rc;  // GeLU, plane rows out
      } else {
        if ((rc = gsplit(x16, ly.wi, ff, dff, M, rows_dev, dff, d, 2, ly.bi, nullptr, 0))) return rc;
        if ((rc = launch_split_f32_bf16x3(ff, dff, pl_ff, ld_ff, M, dff, rows_dev, stream, 1))) return rc;
      }
      if ((rc = gsplit(pl_ff, ly.wo2, t, d, M, rows_dev, d, dff, 0, ly.bo2, x, 0))) return rc;
      if ((rc = launch_layernorm_dev(t, ly.ln2_w, ly.ln2_b, x, rows_dev, M, d, w->eps, nullptr, stream, last ? nullptr : x16, last ? 0 : ld_d)))
        return rc;
      continue;
    }
    if (bf16) {
      if ((rc = lin16(x16, ly.wqkv, qkv, 3 * d, M, rows_dev, 3 * d, d, 0, ly.bqkv, nullptr, 1))) return rc;  // q, k, v as bf16
      if ((rc = launch_attention(at, stream))) return rc;
      if (pooled_only && last) {
        if ((rc = launch_gather_rows(ctx, seq_off, B, d / 2, ctx_cls, stream))) return rc;  // bf16 rows are d / 2 floats wide
        if ((rc = launch_gather_rows(x, seq_off, B, d, x_cls, stream))) return rc;
        if ((rc = lin16(ctx_cls, ly.wo, t_cls, d, B, nullptr, d, d, 0, ly.bo, x_cls, 0))) return rc;
        if ((rc = launch_layernorm(t_cls, ly.ln1_w, ly.ln1_b, x_cls, B, d, w->eps, nullptr, stream, ctx_cls))) return rc;  // + bf16 image
        if ((rc = lin16(ctx_cls, ly.wi, ff_cls, dff, B, nullptr, dff, d, 2, ly.bi, nullptr, 1))) return rc;
        if ((rc = lin16(ff_cls, ly.wo2, t_cls, d, B, nullptr, d, dff, 0, ly.bo2, x_cls, 0))) return rc;
        return launch_layernorm(t_cls, ly.ln2_w, ly.ln2_b, out_pooled, B, d, w->eps, nullptr, stream);
      }
      if ((rc = lin16(ctx, ly.wo, t, d, M, rows_dev, d, d, 0, ly.bo, x, 0))) return rc;
      if ((rc = launch_layernorm_dev(t, ly.ln1_w, ly.ln1_b, x, rows_dev, M, d, w->eps, nullptr, stream, x16))) return rc;
      if ((rc = lin16(x16, ly.wi, ff, dff, M, rows_dev, dff, d, 2, ly.bi, nullptr, 1))) return rc;  // gelu, bf16 out (in the ff buffer)
      if ((rc = lin16(ff, ly.wo2, t, d, M, rows_dev, d, dff, 0, ly.bo2, x, 0))) return rc;
      if ((rc = launch_layernorm_dev(t, ly.ln2_w, ly.ln2_b, x, rows_dev, M, d, w->eps, nullptr, stream, last ? nullptr : x16))) return rc;
      continue;
    }
    if ((rc = linear(x, d, ly.wqkv, qkv, 3 * d, 3 * d, d, GDR_EPI_BIAS, ly.bqkv, nullptr))) return rc;
    if ((rc = launch_attention(at, stream))) return rc;
    if (pooled_only && last) {
      // only sequence_output[:, 0] leaves the call: the rest of the block on the B CLS rows (packed row seq_off[b]); un-split
      // kernels, so the k order — and with it every bit of the result — is that of the full-batch GEMMs
      if ((rc = launch_gather_rows(ctx, seq_off, B, d, ctx_cls, stream))) return rc;
      if ((rc = launch_gather_rows(x, seq_off, B, d, x_cls, stream))) return rc;
      if ((rc = launch_linear_f32(ctx_cls, d, ly.wo, d, t_cls, d, B, d, d, GDR_EPI_BIAS_RESIDUAL, ly.bo, x_cls, d, stream))) return rc;
      if ((rc = launch_layernorm(t_cls, ly.ln1_w, ly.ln1_b, x_cls, B, d, w->eps, nullptr, stream))) return rc;
      if ((rc = launch_linear_f32(x_cls, d, ly.wi, d, ff_cls, dff, B, dff, d, GDR_EPI_BIAS_GELU, ly.bi, nullptr, 0, stream))) return rc;
      if ((rc = launch_linear_f32(ff_cls, dff, ly.wo2, dff, t_cls, d, B, d, dff, GDR_EPI_BIAS_RESIDUAL, ly.bo2, x_cls, d, stream))) return rc;
      return launch_layernorm(t_cls, ly.ln2_w, ly.ln2_b, out_pooled, B, d, w->eps, nullptr, stream);
    }
    if ((rc = linear(ctx, d, ly.wo, t, d, d, d, GDR_EPI_BIAS_RESIDUAL, ly.bo, x))) return rc;
    if ((rc = launch_layernorm_dev(t, ly.ln1_w, ly.ln1_b, x, rows_dev, M, d, w->eps, nullptr, stream))) return rc;
    if ((rc = linear(x, d, ly.wi, ff, dff, dff, d, GDR_EPI_BIAS_GELU, ly.bi, nullptr))) return rc;
    if ((rc = linear(ff, dff, ly.wo2, t, d, d, dff, GDR_EPI_BIAS_RESIDUAL, ly.bo2, x))) return rc;
    if ((rc = launch_layernorm_dev(t, ly.ln2_w, ly.ln2_b, x, rows_dev, M, d, w->eps, nullptr, stream))) return rc;
  }
  // packed rows back into the [B, L, d] layout (PAD rows zero) and / or the CLS pool
  if (out_pooled && (rc = launch_gather_rows(x, seq_off, B, d, out_pooled, stream))) return rc;
  if (out_hidden) {
    if (hipMemsetAsync(out_hidden, 0, (size_t)M * d * sizeof(float), stream) != hipSuccess) {
      set_error("bert_ragged: memset failed");
      return GDR_EHIP;
    }
    if ((rc = launch_scatter_rows(x, row_src, rows_dev, M, d, out_hidden, stream))) return rc;
  }
  return GDR_OK;
}
}  // namespace gdr

extern "C" size_t gdr_bert_encoder_ragged_workspace_bytes(const GdrBertWeights* w, int B, int L) {
  if (!w || B <= 0 || L <= 0) return 0;
  return gdr::bert_rag_ws(*w, B, L).total;
}

extern "C" int gdr_bert_encoder_forward_ragged(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                                               const int64_t* token_type_ids, int B, int L, float* out_hidden, float* out_pooled,
                                               int64_t live_rows_hint, void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::bert_ragged_impl(w, ids, mask, token_type_ids, B, L, out_hidden, out_pooled, live_rows_hint, workspace, workspace_bytes,
                               0, static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_bert_encoder_forward_ragged_bf16(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                                                    const int64_t* token_type_ids, int B, int L, float* out_hidden, float* out_pooled,
                                                    int64_t live_rows_hint, void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::bert_ragged_impl(w, ids, mask, token_type_ids, B, L, out_hidden, out_pooled, live_rows_hint, workspace, workspace_bytes,
                               1, static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_bert_encoder_forward_ragged_split(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                                                     const int64_t* token_type_ids, int B, int L, float* out_hidden, float* out_pooled,
                                                     int64_t live_rows_hint, void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::bert_ragged_impl(w, ids, mask, token_type_ids, B, L, out_hidden, out_pooled, live_rows_hint, workspace, workspace_bytes,
                               2, static_cast<hipStream_t>(stream_));
}
