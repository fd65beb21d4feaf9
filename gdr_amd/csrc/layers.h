// Internal launchers of the non-GEMM layer kernels (layers.hip), shared by encoder.hip / decode.hip.
#pragma once
#include "common.h"

namespace gdr {

// T5 relative-position bucket of distance n = query_pos - key_pos >= 0 for n in [0,128); distances
// >= 128 saturate to the last bucket (reference modeling_t5.py:242-288, max_distance hard-coded 128).
struct BucketLut {
  uint8_t v[128];
};
// nb = buckets per direction (num_buckets/2 when bidirectional, num_buckets otherwise).
BucketLut make_bucket_lut(int nb, int max_distance);

struct AttnArgs {
  const float* q;  // row (b*q_bstride + i), head h at column h*dk
  const float* k;  // row (b*k_bstride + j)
  const float* v;
  float* out;      // row (b*o_bstride + i), head h at column h*dk
  void* out_bf16;  // when non-null the context is written here as bf16 (same indexing, ldo in elements) instead of `out`:
                   // it only feeds the next linear of the bf16 precision mode (every kernel form).
  int64_t ldq, ldk, ldv, ldo;          // row strides in floats
  int64_t q_bstride, k_bstride, o_bstride;  // rows per batch entry
  int B, H, dk, Lq, Lk;
  int q_pos0;                  // absolute position of query row 0 (decode with cache)
  float scale;                 // 1.0 for T5 (modeling_t5.py:384-386), hd^-0.5 for nn.MultiheadAttention
  const float* rel_bias;       // [num_buckets, H] or null
  int bidirectional, num_buckets;
  BucketLut lut;
  const int64_t* key_mask;     // int64 [B, Lk] (1 = attend) or null  -> adds (1-m)*-1e9
  int64_t mask_bstride;        // elements between batch entries of key_mask
  int causal;                  // key j allowed iff j <= q_pos0 + i  -> adds -1e9 otherwise (neg_inf: -inf)
  int causal_neg_inf;          // nn.Transformer masks use -inf (modeling_t5.py:1622-1627)
  // decode-time K/V sources (defaults: kv_rows = null, kv_group = 1)
  const int32_t* kv_rows;      // int32 [B, Lk]: absolute row of key j of batch entry b in k/v (beam-ancestor cache)
  int kv_group;                // batch entries sharing one K/V block: K/V batch index = b / kv_group (beams of a query)
  int q_same_pos;              // all Lq query rows of a batch entry sit at position q_pos0 (beam rows of one decode step)
  // packed (ragged) self-attention: batch entry b owns rows seq_off[b] .. seq_off[b] + seq_len[b] - 1 of q/k/v/out
  // (instead of b*bstride + i) and attends over its seq_len[b] keys; Lq = Lk = the longest possible sequence.  Only the
  // dk = 64 full self-attention form (attention_mfma16_kernel) reads these.
  const int32_t* seq_off;
  const int32_t* seq_len;
  int qkv_bf16;                // q, k, v point to bf16 data (ldq / ldk / ldv in bf16 elements): the qkv linear of the bf16
                               // precision mode emits them so; attention_mfma16_kernel only (d_kv = 64 self-attention)
  const int64_t* b_count_dev;  // Lq = 1 decode form only, may be null: only batch entries b < *b_count_dev are computed
  // q left as split-K slabs by the projection in front (common.h SlabRef; generic kernel only): when q_part != null the
  // query element (row m, column n) is the sum over s < q_S of the slabs, in order, and `q` is not read
  const float* q_part;
  int q_S, q_tiles_n;
  // decode only, may be null: the generate call's "some query is still undecided" word (common.h StreamK::live); 0 = every
  // query is done (generation_utils.py:836-838), the launch exits at once — its output is never read
  const int32_t* live;
};
int launch_attention(const AttnArgs& a, hipStream_t stream);

int launch_embed(const float* table, const int64_t* ids, int64_t rows, int d, int vocab, float* out,
                 hipStream_t stream);
// Ragged batches (encoder.hip): which token rows of a [B,L] batch are computed at all.
//   seq_len[b] = number of leading ones of mask row b when the row is a non-empty prefix of ones (right padding, what the
//                tokenizer produces), else L: such a sequence keeps all its positions and its mask (a fully masked row
//                attends uniformly over PAD keys too, modeling_utils.py:271-272, so nothing of it may be dropped);
//   seq_off[b] = exclusive prefix sum (seq_off[B] = total), row_src[r] = b*L + pos of packed row r, *rows_total = total.
int launch_pack_plan(const int64_t* mask, int B, int L, int32_t* seq_len, int32_t* seq_off, int32_t* row_src,
                     int64_t* rows_total, hipStream_t stream);
// out[r] = table[ids[row_src[r]]] for r < *rows_dev (grid sized for max_rows)
int launch_embed_packed(const float* table, const int64_t* ids, const int32_t* row_src, const int64_t* rows_dev,
                        int64_t max_rows, int d, int vocab, float* out, hipStream_t stream);
// dst[i] = src[idx[i]] (rows of d floats), i < n
int launch_gather_rows(const float* src, const int32_t* idx, int n, int d, float* dst, hipStream_t stream);
// dst[row_src[r]] = src[r] for r < *rows_dev: packed rows back into the [B*L, d] layout (dst pre-zeroed by the caller)
int launch_scatter_rows(const float* src, const int32_t* row_src, const int64_t* rows_dev, int64_t max_rows, int d,
                        float* dst, hipStream_t stream);
// y = x / sqrt(mean(x^2) + eps) * w     (T5LayerNorm, modeling_t5.py:164-171); optional second output
// `pooled` receives rows r with r % pool_every == 0 (CLS pool h[:,0]) when non-null.
// y16 (optional, everywhere below): the same output once more, rounded to bf16 (RNE) — the operand of a bf16-mode linear
int launch_rmsnorm(const float* x, const float* w, float* y, int64_t rows, int d, float eps, float* pooled,
                   int pool_every, hipStream_t stream, void* y16 = nullptr);
// x[b, j, :] = 0 for j >= seq_len[b]  (x is [B, L, d])
int launch_zero_dead_rows(float* x, const int32_t* seq_len, int B, int L, int d, hipStream_t stream);
// same over the first *rows_dev rows (a device-side count; the grid is sized for max_rows)
int launch_rmsnorm_dev(const float* x, const float* w, float* y, const int64_t* rows_dev, int64_t max_rows, int d, float eps,
                       hipStream_t stream, void* y16 = nullptr);
// the split-bf16 form's operand: y (nullable) fp32 rows; planes bf16 [rows, ld >= 3 d], a row = [hi | mid | lo] of the normed row
int launch_rmsnorm_planes(const float* x, const float* w, float* y, void* planes, int64_t ld, const int64_t* rows_dev, int64_t max_rows,
                          int d, float eps, hipStream_t stream, int f16x2 = 0);  // f16x2: rows [fp16 hi | fp16 (x - hi) * 2^11], ld >= 2 d
// same, output rounded to bf16 (RNE): the activation operand of a bf16-mode linear
int launch_rmsnorm_bf16(const float* x, const float* w, void* y_bf16, int64_t rows, int d, float eps, hipStream_t stream);
int launch_rmsnorm_bf16_dev(const float* x, const float* w, void* y_bf16, const int64_t* rows_dev, int64_t max_rows, int d,
                            float eps, hipStream_t stream);
// y = (x - mean) / sqrt(var + eps) * w + b   (torch.nn.LayerNorm; BERT / nn.TransformerDecoderLayer)
// optional addv [d]: y = LN(x + addv)  (the adaptor's constant single-key cross-attention output)
// f16x2_ld != 0: y16 receives plane rows [fp16 hi | fp16 (y - hi) * 2^11] of f16x2_ld >= 2 d elements (the fp16 x 2 split form's operand)
int launch_layernorm(const float* x, const float* w, const float* b, float* y, int64_t rows, int d, float eps,
                     const float* addv, hipStream_t stream, void* y16 = nullptr, int f16x2_ld = 0);
int launch_layernorm_dev(const float* x, const float* w, const float* b, float* y, const int64_t* rows_dev, int64_t max_rows,
                         int d, float eps, const float* addv, hipStream_t stream, void* y16 = nullptr, int f16x2_ld = 0);
// y = LN(LN(x; w1, b1) + addv; w2, b2) in one pass (bit-identical to the two launches); rows_dev may be null (= max_rows rows);
// returns 1 when d > 2048 (not served: run the two launches)
int launch_layernorm2(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* addv, float* y,
                      const int64_t* rows_dev, int64_t max_rows, int d, float eps, hipStream_t stream, void* y16 = nullptr);

}  // namespace gdr
