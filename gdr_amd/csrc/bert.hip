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
