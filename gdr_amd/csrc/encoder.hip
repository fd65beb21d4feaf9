// T5 encoder forward on gfx950 — replaces `T5Stack.forward` of the reference
// (GDR_model/transformers/modeling_t5.py:685-821, blocks :498-584, eval mode: dropout = identity).
//
// Per layer (pre-norm, T5 v1.0):   h += o(attn(qkv(rms(h))))    ;    h += wo(relu(wi(rms(h))))
//   rms           layers.hip  rmsnorm_kernel                                  (modeling_t5.py:164-171)
//   qkv / o / wi / wo   gemm_f32.hip on v_mfma_f32_32x32x2_f32 (persistent form at encoder batch sizes); q,k,v fused into
//                 one [3*inner,d] GEMM,
//                 residual add and ReLU fused into the GEMM epilogues        (:360-364,:413,:182-185,:199)
//   attn          layers.hip  attention_mfma16_kernel (d_kv = 64: S^T = K·Q^T and O^T = V^T·P^T on v_mfma_f32_16x16x4_f32, one
//                 wave per 16-query tile) or attention_kernel (any d_kv): QKᵀ (no 1/sqrt(d)) + bucketed relative bias +
//                 pad mask (1-m)*-1e9 + fp32 softmax + PV                           (:384-413, :290-314)
// The position bias is never materialised as a [B,H,L,L] tensor: the kernel re-derives it from the
// [buckets,H] table (layer 0's, shared by all layers as in :790-795).
#include <stdlib.h>

#include "layers.h"

namespace gdr {

constexpr size_t ENC_SPLITK_BYTES = (size_t)112 << 20;  // <= 384 tail tiles x 4 splits x 64 KiB + slack; the same region
                                                        // serves as the stream-K hand-off scratch (a launch uses one or the other)
static_assert(STREAMK_BYTES <= ENC_SPLITK_BYTES, "stream-K scratch must fit the split-K region");

// the stream-K scratch inside the split-K region of an encoder workspace; flags zeroed once per call
static int enc_streamk(char* splitk_region, StreamK* sk, hipStream_t stream) {
  sk->part = reinterpret_cast<float*>(splitk_region);
  sk->flag = reinterpret_cast<int32_t*>(splitk_region + STREAMK_PART_BYTES);
  sk->epoch = 0;
  if (hipMemsetAsync(sk->flag, 0, 512 * sizeof(int32_t), stream) != hipSuccess) {
    set_error("t5_encoder: memset of the stream-K flags failed");
    return GDR_EHIP;
  }
  return GDR_OK;
}

struct EncWs {
  size_t off_h, off_nx, off_qkv, off_ctx, off_ff, off_splitk, off_bf, total;
};

static EncWs enc_ws(const GdrT5Dims& dm, int64_t M, bool bf16 = false) {
  EncWs w{};
  const size_t inner = (size_t)dm.num_heads * dm.d_kv;
  size_t o = 0;
  w.off_h = o, o += align_up((size_t)M * dm.d_model * 4, 256);
  w.off_nx = o, o += align_up((size_t)M * dm.d_model * 4, 256);
  w.off_qkv = o, o += align_up((size_t)M * 3 * inner * 4, 256);
  w.off_ctx = o, o += align_up((size_t)M * inner * 4, 256);
  w.off_ff = o, o += align_up((size_t)M * dm.d_ff * 4, 256);
  w.off_splitk = o, o += ENC_SPLITK_BYTES;  // small batches: split-K partial slabs (gemm_f32.hip)
  w.off_bf = o;
  if (bf16) o += align_up((size_t)M * (dm.d_ff > dm.d_model ? dm.d_ff : dm.d_model) * 2, 256);  // bf16 copy of a GEMM's A operand
  w.total = o;
  return w;
}

}  // namespace gdr

extern "C" size_t gdr_t5_encoder_workspace_bytes(const GdrT5Dims* dims, int B, int L) {
  if (!dims || B <= 0 || L <= 0) return 0;
  return gdr::enc_ws(*dims, (int64_t)B * L).total;
}

namespace gdr {
static int t5_encoder_impl(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L,
                           float* out_hidden, float* out_pooled, void* workspace, size_t workspace_bytes, bool bf16,
                           hipStream_t stream) {
  if (B == 0) return GDR_OK;  // empty batch
  GDR_CHECK_ARG(w && ids && out_hidden && workspace, "t5_encoder: null pointer");
  const GdrT5Dims& dm = w->dims;
  GDR_CHECK_ARG(B > 0 && L > 0 && L <= 128, "t5_encoder: B=%d L=%d (L must be in [1,128])", B, L);
  GDR_CHECK_ARG(dm.d_model % 4 == 0 && dm.d_kv % 4 == 0 && dm.d_ff % 4 == 0, "t5_encoder: dims must be multiples of 4");
  GDR_CHECK_ARG(w->embed && w->rel_bias && w->final_ln && w->layers, "t5_encoder: null weight pointer");
  const int64_t M = (int64_t)B * L;
  const EncWs ws = enc_ws(dm, M, bf16);
  GDR_CHECK_ARG(!bf16 || (dm.d_model % 8 == 0 && dm.d_ff % 8 == 0 && (dm.num_heads * dm.d_kv) % 8 == 0),
                "t5_encoder_bf16: dims must be multiples of 8");
  if (workspace_bytes < ws.total) {
    set_error("t5_encoder: workspace %zu < required %zu", workspace_bytes, ws.total);
    return GDR_ENOSPC;
  }
  GDR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "t5_encoder: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  float* h = reinterpret_cast<float*>(base + ws.off_h);
  float* nx = reinterpret_cast<float*>(base + ws.off_nx);
  float* qkv = reinterpret_cast<float*>(base + ws.off_qkv);
  float* ctx = reinterpret_cast<float*>(base + ws.off_ctx);
  float* ff = reinterpret_cast<float*>(base + ws.off_ff);
  float* skw = reinterpret_cast<float*>(base + ws.off_splitk);
  void* abf = base + ws.off_bf;
  StreamK sk{};
  if (!bf16)
    if (int rc_ = enc_streamk(base + ws.off_splitk, &sk, stream)) return rc_;
  // one linear: fp32 as is; in bf16 mode the activation operand is rounded to bf16 into `abf` first (the weights were
  // rounded once by the caller) and the GEMM runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulate / epilogue / output
  auto linear = [&](const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int N, int K, int epi,
                    const float* residual) -> int {
    if (!bf16)
      return launch_linear_f32_ws(A, lda, W, K, C, ldc, M, N, K, epi, nullptr, residual, ldc, skw, ENC_SPLITK_BYTES, stream, &sk);
    GDR_CHECK_ARG(lda == K, "t5_encoder_bf16: operand must be dense");
    int rc_ = launch_cast_f32_bf16(A, abf, M * (int64_t)K, stream);
    if (rc_) return rc_;
    return launch_linear_bf16(abf, K, W, K, C, ldc, M, N, K, epi, nullptr, residual, ldc, stream);
  };
  const int d = dm.d_model, H = dm.num_heads, dk = dm.d_kv, inner = H * dk;

  int rc = launch_embed(w->embed, ids, M, d, dm.vocab_size, h, stream);  // modeling_t5.py:725
  if (rc) return rc;

  AttnArgs at{};
  at.q = qkv, at.k = qkv + inner, at.v = qkv + 2 * inner, at.out = ctx;
  at.ldq = at.ldk = at.ldv = 3 * inner, at.ldo = inner;
  at.q_bstride = at.k_bstride = at.o_bstride = L;
  at.B = B, at.H = H, at.dk = dk, at.Lq = L, at.Lk = L;
  at.q_pos0 = 0, at.scale = 1.0f;
  at.rel_bias = w->rel_bias, at.bidirectional = 1, at.num_buckets = dm.rel_buckets;
  at.lut = make_bucket_lut(dm.rel_buckets / 2, dm.rel_max_distance);
  at.key_mask = mask, at.mask_bstride = L, at.causal = 0, at.causal_neg_inf = 0;
  at.kv_rows = nullptr, at.kv_group = 1;

  // bf16 mode, fused form (every contraction length a multiple of 64, so the LDS-DMA linear serves all four): the
  // producers write the bf16 operand of the next linear directly — RMSNorm, the attention context and the ReLU
  // epilogue of wi — instead of fp32 + a cast pass.  Rounding points are the same as in the cast form.
  const bool fused16 = bf16 && d % 64 == 0 && inner % 64 == 0 && dm.d_ff % 64 == 0 && (((uintptr_t)w->layers[0].wqkv) & 15) == 0;
  auto lin16 = [&](const void* A, const float* W, float* C, int64_t ldc, int N, int K, int act, const float* residual,
                   int out_bf16) -> int {
    ProfScope prof(PROF_LINEAR, 2.0 * (double)M * (double)N * (double)K, stream);
    const int rc_ = launch_linear_bf16_glds(A, K, W, K, C, ldc, M, N, K, 0, residual != nullptr, act, nullptr, residual, ldc,
                                            out_bf16, stream);
    if (rc_ > 0) {
      set_error("t5_encoder_bf16: shape not served by the LDS-DMA linear");
      return GDR_EINVAL;
    }
    return rc_;
  };
  for (int i = 0; i < dm.num_layers; ++i) {
    const GdrT5EncLayer& ly = w->layers[i];
    GDR_CHECK_ARG(ly.ln_attn && ly.wqkv && ly.wo && ly.ln_ff && ly.wi && ly.wo_ff, "t5_encoder: layer %d null weight", i);
    if (fused16) {
      if ((rc = launch_rmsnorm_bf16(h, ly.ln_attn, abf, M, d, dm.eps, stream))) return rc;
      // d_kv = 64 (the MFMA attention form): q, k, v leave the linear as bf16 and are widened inside the attention kernel
      const bool qkv16 = dk == 64 && L <= 128;
      if ((rc = lin16(abf, ly.wqkv, qkv, 3 * inner, 3 * inner, d, 0, nullptr, qkv16 ? 1 : 0))) return rc;
      if (qkv16) {
        const __bf16* q16 = reinterpret_cast<const __bf16*>(qkv);
        at.q = reinterpret_cast<const float*>(q16), at.k = reinterpret_cast<const float*>(q16 + inner);
        at.v = reinterpret_cast<const float*>(q16 + 2 * inner), at.qkv_bf16 = 1;
      }
      at.out_bf16 = abf;  // ctx as bf16 [M, inner] (nx is dead)
      if ((rc = launch_attention(at, stream))) return rc;
      if ((rc = lin16(abf, ly.wo, h, d, d, inner, 0, h, 0))) return rc;
      if ((rc = launch_rmsnorm_bf16(h, ly.ln_ff, abf, M, d, dm.eps, stream))) return rc;
      if ((rc = lin16(abf, ly.wi, ff, dm.d_ff, dm.d_ff, d, 1, nullptr, 1))) return rc;  // relu, bf16 out (in the fp32 ff buffer)
      if ((rc = lin16(ff, ly.wo_ff, h, d, d, dm.d_ff, 0, h, 0))) return rc;
      continue;
    }
    if ((rc = launch_rmsnorm(h, ly.ln_attn, nx, M, d, dm.eps, nullptr, 1, stream))) return rc;
    if ((rc = linear(nx, d, ly.wqkv, qkv, 3 * inner, 3 * inner, d, GDR_EPI_NONE, nullptr))) return rc;
    if ((rc = launch_attention(at, stream))) return rc;
    if ((rc = linear(ctx, inner, ly.wo, h, d, d, inner, GDR_EPI_RESIDUAL, h))) return rc;
    if ((rc = launch_rmsnorm(h, ly.ln_ff, nx, M, d, dm.eps, nullptr, 1, stream))) return rc;
    if ((rc = linear(nx, d, ly.wi, ff, dm.d_ff, dm.d_ff, d, GDR_EPI_RELU, nullptr))) return rc;
    if ((rc = linear(ff, dm.d_ff, ly.wo_ff, h, d, d, dm.d_ff, GDR_EPI_RESIDUAL, h))) return rc;
  }
  // final_layer_norm (:803) + CLS pool h[:,0] (main_models.py:102-109)
  return launch_rmsnorm(h, w->final_ln, out_hidden, M, d, dm.eps, out_pooled, L, stream);
}
}  // namespace gdr

extern "C" int gdr_t5_encoder_forward(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B,
                                      int L, float* out_hidden, float* out_pooled, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
  return gdr::t5_encoder_impl(w, ids, mask, B, L, out_hidden, out_pooled, workspace, workspace_bytes, false,
                              static_cast<hipStream_t>(stream_));
}

// ------------------------------------------------------------------------------------------------ ragged form
// The same forward over the token rows that can influence the requested outputs (include/gdr_hip.h):
//   * PAD rows are never computed: the live rows of the batch are packed front to back (launch_pack_plan), every linear /
//     norm / residual runs over the packed rows, attention works per sequence on its own length.  A row's arithmetic is
//     unchanged — GEMM rows are independent and keep their k order, a PAD key contributes exp(-1e9 - max) = 0 to a live
//     query's softmax in the padded form and is simply absent here — so live rows are BIT-IDENTICAL to the padded form.
//   * pooled-only calls (out_hidden == NULL: config C2, EncoderModel.encode_query) stop computing non-CLS rows after the
//     last layer's attention: its o / wi / wo and the final norm run on the B CLS rows alone.
// The row count is a device-side value (the mask lives on the device): grids are sized for B*L and kernels read the count,
// so the call stays free of host synchronisation.
namespace gdr {
struct RagWs {
  size_t seq_len, seq_off, row_src, rows_total, ctx_cls, h_cls, nx_cls, ff_cls, total;
};
static RagWs rag_ws(const GdrT5Dims& dm, int B, int L, size_t base) {
  RagWs r{};
  const size_t inner = (size_t)dm.num_heads * dm.d_kv;
  size_t o = base;
  r.seq_len = o, o += align_up((size_t)B * 4, 256);
  r.seq_off = o, o += align_up((size_t)(B + 1) * 4, 256);
  r.row_src = o, o += align_up((size_t)B * L * 4, 256);
  r.rows_total = o, o += 256;
  r.ctx_cls = o, o += align_up((size_t)B * inner * 4, 256);
  r.h_cls = o, o += align_up((size_t)B * dm.d_model * 4, 256);
  r.nx_cls = o, o += align_up((size_t)B * dm.d_model * 4, 256);
  r.ff_cls = o, o += align_up((size_t)B * dm.d_ff * 4, 256);
  r.total = o;
  return r;
}
// The packed form needs the d_kv = 64 attention kernel.  ragged_packs: a 128x128 tile grid that fills the chip without
// split-K (the narrowest linear has N = d_model) — the un-split forms, with the pooled-only tail on the CLS rows.
// ragged_packs_small (r03): smaller batches (C3's 64-query encoder pass).  The padded form runs those on the split-K /
// stream-K forms of launch_linear_f32_ws, chosen from B*L alone; the packed form takes the SAME forms with the live row
// count on the device, so kept rows are still bit-identical — only the pooled-only tail is not taken (its un-split
// kernels would sum in another order than the padded form's split ones).
static bool ragged_packs(const GdrT5Dims& dm, int B, int L) {
  const int64_t tiles = (((int64_t)B * L + 127) / 128) * ((dm.d_model + 127) / 128);
  return dm.d_kv == 64 && tiles >= 192 && dm.d_model % 32 == 0 && dm.d_ff % 32 == 0 && (dm.num_heads * dm.d_kv) % 32 == 0;
}
static bool ragged_packs_small(const GdrT5Dims& dm, int B, int L) {
  return dm.d_kv == 64 && (int64_t)B * L >= 256 && dm.d_model % 32 == 0 && dm.d_ff % 32 == 0 && (dm.num_heads * dm.d_kv) % 32 == 0;
}
}  // namespace gdr

extern "C" size_t gdr_t5_encoder_ragged_workspace_bytes(const GdrT5Dims* dims, int B, int L) {
  if (!dims || B <= 0 || L <= 0) return 0;
  return gdr::rag_ws(*dims, B, L, gdr::enc_ws(*dims, (int64_t)B * L, true).total).total;  // serves both precisions
}

namespace gdr {
static int ragged_impl(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L, float* out_hidden,
                       float* out_pooled, int64_t live_rows_hint, void* workspace, size_t workspace_bytes, bool bf16,
                       hipStream_t stream) {
  if (B == 0) return GDR_OK;
  GDR_CHECK_ARG(w && ids && mask && workspace && (out_hidden || out_pooled), "t5_encoder_ragged: null pointer");
  const GdrT5Dims& dm = w->dims;
  GDR_CHECK_ARG(B > 0 && L > 0 && L <= 128, "t5_encoder_ragged: B=%d L=%d (L must be in [1,128])", B, L);
  GDR_CHECK_ARG(dm.d_model % 4 == 0 && dm.d_kv % 4 == 0 && dm.d_ff % 4 == 0, "t5_encoder_ragged: dims must be multiples of 4");
  GDR_CHECK_ARG(w->embed && w->rel_bias && w->final_ln && w->layers, "t5_encoder_ragged: null weight pointer");
  const int64_t M = (int64_t)B * L;
  const EncWs ws = enc_ws(dm, M, true);
  const RagWs rw = rag_ws(dm, B, L, ws.total);
  if (workspace_bytes < rw.total) {
    set_error("t5_encoder_ragged: workspace %zu < required %zu", workspace_bytes, rw.total);
    return GDR_ENOSPC;
  }
  GDR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "t5_encoder_ragged: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  const int d = dm.d_model, H = dm.num_heads, dk = dm.d_kv, inner = H * dk;
  int rc;
  int32_t* seq_len = reinterpret_cast<int32_t*>(base + rw.seq_len);
  int32_t* seq_off = reinterpret_cast<int32_t*>(base + rw.seq_off);
  int32_t* row_src = reinterpret_cast<int32_t*>(base + rw.row_src);
  int64_t* rows_dev = reinterpret_cast<int64_t*>(base + rw.rows_total);
  if ((rc = launch_pack_plan(mask, B, L, seq_len, seq_off, row_src, rows_dev, stream))) return rc;
  // bf16 precision mode: the packed form exists for the fused producer chain only (every contraction a multiple of 64)
  const bool fused16 = bf16 && d % 64 == 0 && inner % 64 == 0 && dm.d_ff % 64 == 0 && (((uintptr_t)w->layers[0].wqkv) & 15) == 0;
  const bool big = ragged_packs(dm, B, L);
  const bool small = !big && !bf16 && ragged_packs_small(dm, B, L);
  if ((!big && !small) || (bf16 && !fused16)) {
    // small problem / other head size: the padded forward, then the rows that the packed form would not have computed
    // are zeroed so that the output contract does not depend on which form ran
    float* full = out_hidden ? out_hidden : reinterpret_cast<float*>(base + ws.off_qkv);  // qkv is dead when the final norm runs
    if ((rc = t5_encoder_impl(w, ids, mask, B, L, full, out_pooled, workspace, ws.total, bf16, stream))) return rc;
    return out_hidden ? launch_zero_dead_rows(out_hidden, seq_len, B, L, d, stream) : GDR_OK;
  }
  if (bf16) {
    // ---- packed form of the bf16 precision mode: the producers write the bf16 operand of the next linear (RMSNorm, the
    //      attention context, the ReLU epilogue of wi); same rounding points as gdr_t5_encoder_forward_bf16, rows independent
    float* h = reinterpret_cast<float*>(base + ws.off_h);
    float* nx = reinterpret_cast<float*>(base + ws.off_nx);
    float* qkv = reinterpret_cast<float*>(base + ws.off_qkv);
    float* ff = reinterpret_cast<float*>(base + ws.off_ff);
    void* abf = base + ws.off_bf;
    float* h_cls = reinterpret_cast<float*>(base + rw.h_cls);
    void* ctx_cls16 = base + rw.ctx_cls;
    void* nx_cls16 = base + rw.nx_cls;
    void* ff_cls16 = base + rw.ff_cls;
    auto lin16 = [&](const void* A, const float* W, float* C, int64_t ldc, int64_t rows, const int64_t* md, int N, int K, int act,
                     const float* residual, int out_bf16) -> int {
      ProfScope prof(PROF_LINEAR, 2.0 * (double)(md && live_rows_hint >= 0 ? live_rows_hint : rows) * (double)N * (double)K, stream);
      const int rc_ = launch_linear_bf16_glds(A, K, W, K, C, ldc, rows, N, K, 0, residual != nullptr, act, nullptr, residual, ldc,
                                              out_bf16, stream, md);
      if (rc_ > 0) {
        set_error("t5_encoder_ragged_bf16: shape not served by the LDS-DMA linear");
        return GDR_EINVAL;
      }
      return rc_;
    };
    if ((rc = launch_embed_packed(w->embed, ids, row_src, rows_dev, M, d, dm.vocab_size, h, stream))) return rc;
    AttnArgs at{};
    {
      const __bf16* q16 = reinterpret_cast<const __bf16*>(qkv);  // the packed form always runs the d_kv = 64 MFMA attention
      at.q = reinterpret_cast<const float*>(q16), at.k = reinterpret_cast<const float*>(q16 + inner);
      at.v = reinterpret_cast<const float*>(q16 + 2 * inner), at.qkv_bf16 = 1;
    }
    at.out = nullptr, at.out_bf16 = abf;
    at.ldq = at.ldk = at.ldv = 3 * inner, at.ldo = inner;
    at.q_bstride = at.k_bstride = at.o_bstride = L;
    at.B = B, at.H = H, at.dk = dk, at.Lq = L, at.Lk = L;
    at.q_pos0 = 0, at.scale = 1.0f;
    at.rel_bias = w->rel_bias, at.bidirectional = 1, at.num_buckets = dm.rel_buckets;
    at.lut = make_bucket_lut(dm.rel_buckets / 2, dm.rel_max_distance);
    at.key_mask = mask, at.mask_bstride = L, at.causal = 0, at.causal_neg_inf = 0;
    at.kv_rows = nullptr, at.kv_group = 1;
    at.seq_off = seq_off, at.seq_len = seq_len;
    const bool pooled_only = out_hidden == nullptr;
    for (int i = 0; i < dm.num_layers; ++i) {
      const GdrT5EncLayer& ly = w->layers[i];
      GDR_CHECK_ARG(ly.ln_attn && ly.wqkv && ly.wo && ly.ln_ff && ly.wi && ly.wo_ff, "t5_encoder_ragged: layer %d null weight", i);
      if ((rc = launch_rmsnorm_bf16_dev(h, ly.ln_attn, abf, rows_dev, M, d, dm.eps, stream))) return rc;
      if ((rc = lin16(abf, ly.wqkv, qkv, 3 * inner, M, rows_dev, 3 * inner, d, 0, nullptr, 1))) return rc;  // q,k,v as bf16
      if ((rc = launch_attention(at, stream))) return rc;  // ctx -> abf as bf16 [rows, inner]
      if (pooled_only && i == dm.num_layers - 1) {
        // bf16 rows are inner/2 (d/2, d_ff/2) floats wide for the row mover
        if ((rc = launch_gather_rows(static_cast<const float*>(abf), seq_off, B, inner / 2, static_cast<float*>(ctx_cls16), stream)))
          return rc;
        if ((rc = launch_gather_rows(h, seq_off, B, d, h_cls, stream))) return rc;
        if ((rc = lin16(ctx_cls16, ly.wo, h_cls, d, B, nullptr, d, inner, 0, h_cls, 0))) return rc;
        if ((rc = launch_rmsnorm_bf16(h_cls, ly.ln_ff, nx_cls16, B, d, dm.eps, stream))) return rc;
        if ((rc = lin16(nx_cls16, ly.wi, static_cast<float*>(ff_cls16), dm.d_ff, B, nullptr, dm.d_ff, d, 1, nullptr, 1))) return rc;
        if ((rc = lin16(ff_cls16, ly.wo_ff, h_cls, d, B, nullptr, d, dm.d_ff, 0, h_cls, 0))) return rc;
        return launch_rmsnorm(h_cls, w->final_ln, out_pooled, B, d, dm.eps, nullptr, 1, stream);
      }
      if ((rc = lin16(abf, ly.wo, h, d, M, rows_dev, d, inner, 0, h, 0))) return rc;
      if ((rc = launch_rmsnorm_bf16_dev(h, ly.ln_ff, abf, rows_dev, M, d, dm.eps, stream))) return rc;
      if ((rc = lin16(abf, ly.wi, ff, dm.d_ff, M, rows_dev, dm.d_ff, d, 1, nullptr, 1))) return rc;  // relu, bf16 out (in the ff buffer)
      if ((rc = lin16(ff, ly.wo_ff, h, d, M, rows_dev, d, dm.d_ff, 0, h, 0))) return rc;
    }
    if ((rc = launch_rmsnorm_dev(h, w->final_ln, nx, rows_dev, M, d, dm.eps, stream))) return rc;
    if (out_pooled && (rc = launch_gather_rows(nx, seq_off, B, d, out_pooled, stream))) return rc;
    if (out_hidden) {
      if (hipMemsetAsync(out_hidden, 0, (size_t)M * d * sizeof(float), stream) != hipSuccess) {
        set_error("t5_encoder_ragged: memset failed");
        return GDR_EHIP;
      }
      if ((rc = launch_scatter_rows(nx, row_src, rows_dev, M, d, out_hidden, stream))) return rc;
    }
    return GDR_OK;
  }
  float* h = reinterpret_cast<float*>(base + ws.off_h);
  float* nx = reinterpret_cast<float*>(base + ws.off_nx);
  float* qkv = reinterpret_cast<float*>(base + ws.off_qkv);
  float* ctx = reinterpret_cast<float*>(base + ws.off_ctx);
  float* ff = reinterpret_cast<float*>(base + ws.off_ff);
  float* ctx_cls = reinterpret_cast<float*>(base + rw.ctx_cls);
  float* h_cls = reinterpret_cast<float*>(base + rw.h_cls);
  float* nx_cls = reinterpret_cast<float*>(base + rw.nx_cls);
  float* ff_cls = reinterpret_cast<float*>(base + rw.ff_cls);
  StreamK sk{};
  if ((rc = enc_streamk(base + ws.off_splitk, &sk, stream))) return rc;
  float* skw = reinterpret_cast<float*>(base + ws.off_splitk);
  auto linear = [&](const float* A, int64_t lda, const float* W, float* C, int64_t ldc, int N, int K, int epi,
                    const float* residual) -> int {
    if (small)  // the forms the padded forward picks for B*L rows (split-K / stream-K), over the live rows only
      return launch_linear_f32_ws(A, lda, W, K, C, ldc, M, N, K, epi, nullptr, residual, ldc, skw, ENC_SPLITK_BYTES, stream, &sk,
                                  rows_dev, live_rows_hint);
    return launch_linear_f32_dev(A, lda, W, K, C, ldc, M, rows_dev, N, K, epi, nullptr, residual, ldc, live_rows_hint, stream,
                                 &sk);
  };
  if ((rc = launch_embed_packed(w->embed, ids, row_src, rows_dev, M, d, dm.vocab_size, h, stream))) return rc;

  AttnArgs at{};
  at.q = qkv, at.k = qkv + inner, at.v = qkv + 2 * inner, at.out = ctx;
  at.ldq = at.ldk = at.ldv = 3 * inner, at.ldo = inner;
  at.q_bstride = at.k_bstride = at.o_bstride = L;
  at.B = B, at.H = H, at.dk = dk, at.Lq = L, at.Lk = L;
  at.q_pos0 = 0, at.scale = 1.0f;
  at.rel_bias = w->rel_bias, at.bidirectional = 1, at.num_buckets = dm.rel_buckets;
  at.lut = make_bucket_lut(dm.rel_buckets / 2, dm.rel_max_distance);
  at.key_mask = mask, at.mask_bstride = L, at.causal = 0, at.causal_neg_inf = 0;
  at.kv_rows = nullptr, at.kv_group = 1;
  at.seq_off = seq_off, at.seq_len = seq_len;

  const bool pooled_only = out_hidden == nullptr && big;
  for (int i = 0; i < dm.num_layers; ++i) {
    const GdrT5EncLayer& ly = w->layers[i];
    GDR_CHECK_ARG(ly.ln_attn && ly.wqkv && ly.wo && ly.ln_ff && ly.wi && ly.wo_ff, "t5_encoder_ragged: layer %d null weight", i);
    if ((rc = launch_rmsnorm_dev(h, ly.ln_attn, nx, rows_dev, M, d, dm.eps, stream))) return rc;
    if ((rc = linear(nx, d, ly.wqkv, qkv, 3 * inner, 3 * inner, d, GDR_EPI_NONE, nullptr))) return rc;
    if ((rc = launch_attention(at, stream))) return rc;
    if (pooled_only && i == dm.num_layers - 1) {
      // only h[:,0] leaves this call: the rest of the block on the B CLS rows (packed row seq_off[b]); unsplit kernels,
      // so the k order — and with it every bit of the result — is that of the full-batch GEMMs
      if ((rc = launch_gather_rows(ctx, seq_off, B, inner, ctx_cls, stream))) return rc;
      if ((rc = launch_gather_rows(h, seq_off, B, d, h_cls, stream))) return rc;
      if ((rc = launch_linear_f32(ctx_cls, inner, ly.wo, inner, h_cls, d, B, d, inner, GDR_EPI_RESIDUAL, nullptr, h_cls, d, stream)))
        return rc;
      if ((rc = launch_rmsnorm(h_cls, ly.ln_ff, nx_cls, B, d, dm.eps, nullptr, 1, stream))) return rc;
      if ((rc = launch_linear_f32(nx_cls, d, ly.wi, d, ff_cls, dm.d_ff, B, dm.d_ff, d, GDR_EPI_RELU, nullptr, nullptr, 0, stream)))
        return rc;
      if ((rc = launch_linear_f32(ff_cls, dm.d_ff, ly.wo_ff, dm.d_ff, h_cls, d, B, d, dm.d_ff, GDR_EPI_RESIDUAL, nullptr, h_cls, d,
                                  stream)))
        return rc;
      return launch_rmsnorm(h_cls, w->final_ln, out_pooled, B, d, dm.eps, nullptr, 1, stream);
    }
    if ((rc = linear(ctx, inner, ly.wo, h, d, d, inner, GDR_EPI_RESIDUAL, h))) return rc;
    if ((rc = launch_rmsnorm_dev(h, ly.ln_ff, nx, rows_dev, M, d, dm.eps, stream))) return rc;
    if ((rc = linear(nx, d, ly.wi, ff, dm.d_ff, dm.d_ff, d, GDR_EPI_RELU, nullptr))) return rc;
    if ((rc = linear(ff, dm.d_ff, ly.wo_ff, h, d, d, dm.d_ff, GDR_EPI_RESIDUAL, h))) return rc;
  }
  // final_layer_norm over the live rows, then back into the [B,L,d] layout (PAD rows zero) and / or the CLS pool
  if ((rc = launch_rmsnorm_dev(h, w->final_ln, nx, rows_dev, M, d, dm.eps, stream))) return rc;
  if (out_pooled && (rc = launch_gather_rows(nx, seq_off, B, d, out_pooled, stream))) return rc;
  if (out_hidden) {
    if (hipMemsetAsync(out_hidden, 0, (size_t)M * d * sizeof(float), stream) != hipSuccess) {
      set_error("t5_encoder_ragged: memset failed");
      return GDR_EHIP;
    }
    if ((rc = launch_scatter_rows(nx, row_src, rows_dev, M, d, out_hidden, stream))) return rc;
  }
  return GDR_OK;
}
}  // namespace gdr

// ------------------------------------------------------------------------------------------------ split form (r06, exploratory)
// The ragged forward with every linear in the split-bf16 form (gemm_bf16.hip: fp32 operands carried as three bf16 planes, the six
// leading products on the bf16 MFMA path, fp32 accumulate — the fp32 linear's error against float64, not its bits).  Everything
// else — embedding, T5LayerNorm, attention (fp32 MFMA), residual stream, CLS tail — is the fp32 path's.  The linear weights of `w`
// point to plane-form bf16 rows [N, gdr_split_row_elems(K)]; activations are split by their producers' followers
// (split_f32_bf16x3_kernel) into one plane buffer.  Beside gdr_t5_encoder_forward_ragged, never instead of it.
namespace gdr {
// two plane buffers: one for the d_model / inner wide operands (normed rows, attention context), one for the d_ff wide ReLU output,
// which the wi GEMM's epilogue writes while it still reads the first
static size_t split_ws_a_bytes(const GdrT5Dims& dm, int64_t M) {  // sized for the bf16 x 3 form (the fp16 x 2 rows are shorter)
  const int inner = dm.num_heads * dm.d_kv;
  int ld = split_row_elems(dm.d_model);
  if (split_row_elems(inner) > ld) ld = split_row_elems(inner);
  return align_up((size_t)M * ld * 2, 256);
}
static size_t split_ws_bytes(const GdrT5Dims& dm, int64_t M) {
  return split_ws_a_bytes(dm, M) + align_up((size_t)M * split_row_elems(dm.d_ff) * 2, 256);
}

static int ragged_split_impl(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L, float* out_hidden,
                             float* out_pooled, int64_t live_rows_hint, int terms, void* workspace, size_t workspace_bytes,
                             hipStream_t stream) {
  if (B == 0) return GDR_OK;
  GDR_CHECK_ARG(terms == 6 || terms == 3 || terms == 2, "t5_encoder_split: terms must be 6 or 3 (bf16 planes) or 2 (fp16 x 2)");
  const int f16 = terms == 2 ? 1 : 0;
  GDR_CHECK_ARG(w && ids && mask && workspace && (out_hidden || out_pooled), "t5_encoder_split: null pointer");
  const GdrT5Dims& dm = w->dims;
  GDR_CHECK_ARG(B > 0 && L > 0 && L <= 128, "t5_encoder_split: B=%d L=%d (L must be in [1,128])", B, L);
  GDR_CHECK_ARG(w->embed && w->rel_bias && w->final_ln && w->layers, "t5_encoder_split: null weight pointer");
  const int d = dm.d_model, H = dm.num_heads, dk = dm.d_kv, inner = H * dk, dff = dm.d_ff;
  GDR_CHECK_ARG(dk == 64 && d % 64 == 0 && inner % 64 == 0 && dff % 64 == 0,
                "t5_encoder_split: needs d_kv = 64 and d_model, inner, d_ff multiples of 64 (the packed attention and the LDS-DMA linear)");
  const int64_t M = (int64_t)B * L;
  const EncWs ws = enc_ws(dm, M, true);
  const RagWs rw = rag_ws(dm, B, L, ws.total);
  const size_t planes_off = rw.total, total = planes_off + split_ws_bytes(dm, M);
  if (workspace_bytes < total) {
    set_error("t5_encoder_split: workspace %zu < required %zu", workspace_bytes, total);
    return GDR_ENOSPC;
  }
  GDR_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "t5_encoder_split: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  int rc;
  int32_t* seq_len = reinterpret_cast<int32_t*>(base + rw.seq_len);
  int32_t* seq_off = reinterpret_cast<int32_t*>(base + rw.seq_off);
  int32_t* row_src = reinterpret_cast<int32_t*>(base + rw.row_src);
  int64_t* rows_dev = reinterpret_cast<int64_t*>(base + rw.rows_total);
  if ((rc = launch_pack_plan(mask, B, L, seq_len, seq_off, row_src, rows_dev, stream))) return rc;
  float* h = reinterpret_cast<float*>(base + ws.off_h);
  float* nx = reinterpret_cast<float*>(base + ws.off_nx);
  float* qkv = reinterpret_cast<float*>(base + ws.off_qkv);
  float* ctx = reinterpret_cast<float*>(base + ws.off_ctx);
  float* ff = reinterpret_cast<float*>(base + ws.off_ff);
  float* ctx_cls = reinterpret_cast<float*>(base + rw.ctx_cls);
  float* h_cls = reinterpret_cast<float*>(base + rw.h_cls);
  float* nx_cls = reinterpret_cast<float*>(base + rw.nx_cls);
  float* ff_cls = reinterpret_cast<float*>(base + rw.ff_cls);
  void* planes = base + planes_off;                                  // d_model / inner wide operands
  void* planes_ff = base + planes_off + split_ws_a_bytes(dm, M);     // the d_ff wide ReLU output of wi
  static const bool fuse_on = [] {
    const char* e = getenv("GDR_SPLIT_FUSE");  // A/B knob: 0 = every operand split by a launch of its own (split_f32_bf16x3_kernel)
    return e ? atoi(e) != 0 : true;
  }();
  // the split GEMM over plane rows P [rows, split_row_elems(K)]; out_planes: the output leaves as plane rows (row stride ldc elements)
  auto gemm = [&](const void* P, const float* W, void* C, int64_t ldc, int64_t rows, const int64_t* md, int N, int K, int act,
                  const float* residual, int out_planes) -> int {
    const int ld = split_row_elems(K, f16);
    ProfScope prof(PROF_LINEAR, 2.0 * (double)(md && live_rows_hint >= 0 ? live_rows_hint : rows) * (double)N * (double)K, stream);
    const int rc_ = launch_linear_bf16_glds(P, ld, W, ld, static_cast<float*>(C), ldc, rows, N, K, 0, residual != nullptr, act, nullptr, residual,
                                            ldc, out_planes ? (f16 ? 3 : 2) : 0, stream, md, terms);
    if (rc_ > 0) {
      set_error("t5_encoder_split: shape not served by the LDS-DMA linear");
      return GDR_EINVAL;
    }
    return rc_;
  };
  // one linear over `rows` fp32 rows (md: their device-side count, or null): split the operand into planes, then the split GEMM
  auto linear = [&](const float* A, const float* W, float* C, int64_t ldc, int64_t rows, const int64_t* md, int N, int K, int act,
                    const float* residual) -> int {
    void* P = K == dff ? planes_ff : planes;
    if (int rc_ = launch_split_f32_bf16x3(A, K, P, split_row_elems(K, f16), rows, K, md, stream, f16)) return rc_;
    return gemm(P, W, C, ldc, rows, md, N, K, act, residual, 0);
  };
  // the wi GEMM's plane epilogue lives in the 256-row tile kernel, which serves >= 8 192 rows
  const bool fuse = fuse_on && M >= 8192;
  const int ld_d = split_row_elems(d, f16), ld_ff = split_row_elems(dff, f16);
  if ((rc = launch_embed_packed(w->embed, ids, row_src, rows_dev, M, d, dm.vocab_size, h, stream))) return rc;
  AttnArgs at{};
  at.q = qkv, at.k = qkv + inner, at.v = qkv + 2 * inner, at.out = ctx;
  at.ldq = at.ldk = at.ldv = 3 * inner, at.ldo = inner;
  at.q_bstride = at.k_bstride = at.o_bstride = L;
  at.B = B, at.H = H, at.dk = dk, at.Lq = L, at.Lk = L;
  at.q_pos0 = 0, at.scale = 1.0f;
  at.rel_bias = w->rel_bias, at.bidirectional = 1, at.num_buckets = dm.rel_buckets;
  at.lut = make_bucket_lut(dm.rel_buckets / 2, dm.rel_max_distance);
  at.key_mask = mask, at.mask_bstride = L, at.causal = 0, at.causal_neg_inf = 0;
  at.kv_rows = nullptr, at.kv_group = 1;
  at.seq_off = seq_off, at.seq_len = seq_len;
  const bool pooled_only = out_hidden == nullptr;
  for (int i = 0; i < dm.num_layers; ++i) {
    const GdrT5EncLayer& ly = w->layers[i];
    GDR_CHECK_ARG(ly.ln_attn && ly.wqkv && ly.wo && ly.ln_ff && ly.wi && ly.wo_ff, "t5_encoder_split: layer %d null weight", i);
    if (fuse) {  // the norm writes the operand's planes itself (no fp32 copy, no split launch)
      if ((rc = launch_rmsnorm_planes(h, ly.ln_attn, nullptr, planes, ld_d, rows_dev, M, d, dm.eps, stream, f16))) return rc;
      if ((rc = gemm(planes, ly.wqkv, qkv, 3 * inner, M, rows_dev, 3 * inner, d, 0, nullptr, 0))) return rc;
    } else {
      if ((rc = launch_rmsnorm_dev(h, ly.ln_attn, nx, rows_dev, M, d, dm.eps, stream))) return rc;
      if ((rc = linear(nx, ly.wqkv, qkv, 3 * inner, M, rows_dev, 3 * inner, d, 0, nullptr))) return rc;
    }
    if ((rc = launch_attention(at, stream))) return rc;
    if (pooled_only && i == dm.num_layers - 1) {
      if ((rc = launch_gather_rows(ctx, seq_off, B, inner, ctx_cls, stream))) return rc;
      if ((rc = launch_gather_rows(h, seq_off, B, d, h_cls, stream))) return rc;
      if ((rc = linear(ctx_cls, ly.wo, h_cls, d, B, nullptr, d, inner, 0, h_cls))) return rc;
      if ((rc = launch_rmsnorm(h_cls, ly.ln_ff, nx_cls, B, d, dm.eps, nullptr, 1, stream))) return rc;
      if ((rc = linear(nx_cls, ly.wi, ff_cls, dff, B, nullptr, dff, d, 1, nullptr))) return rc;
      if ((rc = linear(ff_cls, ly.wo_ff, h_cls, d, B, nullptr, d, dff, 0, h_cls))) return rc;
      return launch_rmsnorm(h_cls, w->final_ln, out_pooled, B, d, dm.eps, nullptr, 1, stream);
    }
    if ((rc = linear(ctx, ly.wo, h, d, M, rows_dev, d, inner, 0, h))) return rc;
    if (fuse) {  // norm -> planes; wi's ReLU epilogue writes the planes of wo_ff's operand
      if ((rc = launch_rmsnorm_planes(h, ly.ln_ff, nullptr, planes, ld_d, rows_dev, M, d, dm.eps, stream, f16))) return rc;
      if ((rc = gemm(planes, ly.wi, planes_ff, ld_ff, M, rows_dev, dff, d, 1, nullptr, 1))) return rc;
      if ((rc = gemm(planes_ff, ly.wo_ff, h, d, M, rows_dev, d, dff, 0, h, 0))) return rc;
    } else {
      if ((rc = launch_rmsnorm_dev(h, ly.ln_ff, nx, rows_dev, M, d, dm.eps, stream))) return rc;
      if ((rc = linear(nx, ly.wi, ff, dff, M, rows_dev, dff, d, 1, nullptr))) return rc;
      if ((rc = linear(ff, ly.wo_ff, h, d, M, rows_dev, d, dff, 0, h))) return rc;
    }
  }
  if ((rc = launch_rmsnorm_dev(h, w->final_ln, nx, rows_dev, M, d, dm.eps, stream))) return rc;
  if (out_pooled && (rc = launch_gather_rows(nx, seq_off, B, d, out_pooled, stream))) return rc;
  if (out_hidden) {
    if (hipMemsetAsync(out_hidden, 0, (size_t)M * d * sizeof(float), stream) != hipSuccess) {
      set_error("t5_encoder_split: memset failed");
      return GDR_EHIP;
    }
    if ((rc = launch_scatter_rows(nx, row_src, rows_dev, M, d, out_hidden, stream))) return rc;
  }
  return GDR_OK;
}
}  // namespace gdr

extern "C" size_t gdr_t5_encoder_split_workspace_bytes(const GdrT5Dims* dims, int B, int L) {
  if (!dims || B <= 0 || L <= 0) return 0;
  const int64_t M = (int64_t)B * L;
  return gdr::rag_ws(*dims, B, L, gdr::enc_ws(*dims, M, true).total).total + gdr::split_ws_bytes(*dims, M);
}

extern "C" int gdr_t5_encoder_forward_ragged_split(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L,
                                                   float* out_hidden, float* out_pooled, int64_t live_rows_hint, int terms,
                                                   void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::ragged_split_impl(w, ids, mask, B, L, out_hidden, out_pooled, live_rows_hint, terms, workspace, workspace_bytes,
                                static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_t5_encoder_forward_ragged(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B,
                                             int L, float* out_hidden, float* out_pooled, int64_t live_rows_hint,
                                             void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::ragged_impl(w, ids, mask, B, L, out_hidden, out_pooled, live_rows_hint, workspace, workspace_bytes, false,
                          static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_t5_encoder_forward_ragged_bf16(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B,
                                                  int L, float* out_hidden, float* out_pooled, int64_t live_rows_hint,
                                                  void* workspace, size_t workspace_bytes, void* stream_) {
  return gdr::ragged_impl(w, ids, mask, B, L, out_hidden, out_pooled, live_rows_hint, workspace, workspace_bytes, true,
                          static_cast<hipStream_t>(stream_));
}

extern "C" size_t gdr_t5_encoder_bf16_workspace_bytes(const GdrT5Dims* dims, int B, int L) {
  if (!dims || B <= 0 || L <= 0) return 0;
  return gdr::enc_ws(*dims, (int64_t)B * L, true).total;
}

extern "C" int gdr_t5_encoder_forward_bf16(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B,
                                           int L, float* out_hidden, float* out_pooled, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
  return gdr::t5_encoder_impl(w, ids, mask, B, L, out_hidden, out_pooled, workspace, workspace_bytes, true,
                              static_cast<hipStream_t>(stream_));
}
