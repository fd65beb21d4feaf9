/*
 * gdr_hip.h — C ABI of libgdr_hip.so, the MI355X (gfx950) implementation of GDR's inference hot path.
 *
 * The reference (ypw0102/GDR) has no FFI: its boundary is the Python call surface of
 * GDR_model/main_models.py / GDR_model/transformers.  gdr_amd/modeling.py keeps
 * that surface (generate(), get_encoder(), encode_query(), compute_similarity()) and binds the entry
 * points below with ctypes (gdr_amd/_ffi.py); INTEGRATION.md shows the stub a reference maintainer
 * would add.  Each entry point names the reference call site it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (tensor.data_ptr()); outputs are
 *     pre-allocated by the caller; the library allocates nothing — scratch comes from the caller's
 *     workspace, sized by the matching *_workspace_bytes();
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls are asynchronous on it and
 *     contain no host synchronisation (hipGraph-capturable);
 *   - return 0 on success, a negative GDR_E* code otherwise; gdr_last_error() gives the thread-local
 *     message; no exceptions cross the ABI.  Re-entrant from several host threads and for several devices in one
 *     process (the device current in the calling thread is the one used): the only process-wide state is the
 *     thread-local message, a mutex-guarded (kernel, device) table of raised dynamic-LDS limits, a mutex-guarded pool
 *     of per-device side streams that gdr_t5_generate leases per call, and the opt-in profiler below (single-threaded);
 *   - row-major fp32 unless stated; ids int64 where the reference uses LongTensor inputs, int32 for
 *     doc ids produced on device.
 */
#ifndef GDR_HIP_H
#define GDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDR_OK 0
#define GDR_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define GDR_ENOSPC (-2)   /* workspace too small */
#define GDR_EHIP (-3)     /* HIP runtime error at launch, or a device-side failure of an EARLIER launch (see stream-K below) */

const char* gdr_last_error(void);
int gdr_abi_version(void);
/* Kernel launches this library has enqueued in this process so far (every entry point, every stream): a monitoring counter —
 * bench.py reports launches per generate() call from it. */
int64_t gdr_launch_count(void);

/* Opt-in launch profiler used by bench.py for the roofline line: while enabled, every launch of the dense
 * kernels is bracketed by a hipEvent pair recorded on the stream it is launched on.  Process-global, not
 * thread-safe, off by default (then no event is created or recorded).  gdr_prof_collect synchronises,
 * fills three host arrays of length 8 indexed by kernel class {0 linear GEMM, 1 sim sample GEMM,
 * 2 sim filter GEMM} with {launch count, total ms, total flops}, and disables the profiler again. */
int gdr_prof_enable(int max_events);
int gdr_prof_collect(int64_t* launches, double* total_ms, double* total_work);
/* Sampling: while the gate is 0 an enabled profiler records nothing (and costs nothing).  Two hipEventRecord packets per launch are
 * not free — around every dense launch of the C2 step they took 2.4 % of its throughput (28.69 k against 28.02 k q/s, r05) —, so
 * bench.py opens the gate for every 4th step of its timed region only.  The gate is open after gdr_prof_enable. */
void gdr_prof_gate(int on);

/* ------------------------------------------------------------------------------------------------
 * Dense linear:  C[M,N] = epilogue(A[M,K] · W[N,K]^T)      (nn.Linear layout, both K-contiguous)
 * replaces every `nn.Linear` / `torch.matmul` call site on the path:
 *   transformers/modeling_t5.py:360-364,413 (q/k/v/o), :182-185 (wi/ReLU/wo), :1634 (adaptor_linear),
 *   dense.py:21-23 (pooler), torch nn.TransformerDecoderLayer linears (modeling_t5.py:1241-1244).
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (bit-identical to a k-ordered fmaf chain).
 * K % 4 == 0; lda/ldw/ldc in elements, multiples of 4.  `residual` may alias C.
 * ---------------------------------------------------------------------------------------------- */
enum {
  GDR_EPI_NONE = 0,
  GDR_EPI_RESIDUAL = 1,       /* C = acc + residual                     (h + dropout(y), modeling_t5.py:199,452) */
  GDR_EPI_RELU = 2,           /* C = max(acc, 0)                        (modeling_t5.py:183)                      */
  GDR_EPI_BIAS = 3,           /* C = acc + bias[n]                                                                */
  GDR_EPI_BIAS_RELU = 4,      /* C = max(acc + bias[n], 0)              (TransformerDecoderLayer linear1)         */
  GDR_EPI_BIAS_RESIDUAL = 5,  /* C = acc + bias[n] + residual                                                     */
  GDR_EPI_BIAS_GELU = 6       /* C = gelu_erf(acc + bias[n])            (BERT intermediate, modeling_bert.py)     */
};
int gdr_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                   int64_t M, int N, int K, int epilogue, const float* bias, const float* residual,
                   int64_t ldr, void* stream);
/* Same, with a scratch buffer: when the 128x128 tile grid cannot fill the 256 CUs (decode-time linears with
 * M = batch*beams rows) the K dimension is split into partial slabs [S][M][N] in `workspace` and reduced in fixed
 * order (deterministic) by a second kernel that applies the epilogue.  workspace may be NULL (= gdr_linear_f32).
 * A workspace of at least 33 558 528 bytes (512 x 64 KiB + 4 KiB) also serves grids of more than 256 tiles: the last
 * tiles of a launch are then dealt by K-step ranges with an exact accumulator hand-off between workgroups (bit-identical
 * to whole tiles, DESIGN.md §4 "stream-K tail") instead of leaving CUs idle in a partial last round of tiles.
 * Co-residency: that form launches at most 512 workgroups (2 per CU) and a workgroup waits for its predecessor's
 * accumulators, so all of them must become resident while the launch runs — true on an otherwise idle or normally shared
 * GPU.  If another kernel keeps a predecessor off the chip for seconds, the waiting workgroup stops waiting, the launch
 * finishes with INVALID output, and the next linear launched through this library returns GDR_EHIP once (no trap, no hang);
 * the same holds for the encoder / decode entry points, which use this form internally. */
int gdr_linear_f32_splitk(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc,
                          int64_t M, int N, int K, int epilogue, const float* bias, const float* residual,
                          int64_t ldr, void* workspace, size_t workspace_bytes, void* stream);

/* bf16 operands (A [M,K], W [N,K] bf16, round-to-nearest-even of the fp32 tensors), fp32 accumulate, epilogue and output:
 * the linear of the opt-in bf16 precision mode (BASELINE config C5; the reference itself runs precision=32).  K, lda,
 * ldw multiples of 8.  K % 64 == 0 takes the LDS-DMA kernel (gemm_bf16.hip), other K the generic core. */
int gdr_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                    int K, int epilogue, const float* bias, const float* residual, int64_t ldr, void* stream);
/* EXPLORATORY (r06), beside the fp32 linear, never instead of it: fp32 operands carried as three bf16 planes — x = hi + mid + lo, hi =
 * bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (gdr_split_f32_bf16x3: fp32 [rows, K] -> bf16 [rows, ld_out >= 3K], a row = [hi | mid | lo | pad]) —
 * and a linear that keeps the six leading products hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid on the bf16 MFMA path, fp32 accumulate
 * (the call sites of gdr_linear_f32: modeling_t5.py:360-364,413,182-185).  24 significand bits are carried: the error against float64 is
 * that of the strict-fp32 MFMA linear (tools/exp_split_bf16.py), the bits are not.  A3 [M, lda >= 3K], W3 [N, ldw >= 3K] bf16; K % 64 == 0. */
int gdr_split_row_elems(int K, int terms);   /* row length (elements) of a plane-form operand: 3 K (terms 6 / 3) or 2 K (terms 2), rounded up to 64 */
int gdr_split_f32_bf16x3(const float* in, void* out_planes, int64_t rows, int K, int64_t ld_out, void* stream);
/* fp16 x 2 planes (terms = 2 below): a row = [hi | lo'] with hi = fp16(x), lo' = fp16((x - hi) * 2^11) — 22 significand bits; the linear
 * computes hi.hi + 2^-11 (hi.lo' + lo'.hi) on v_mfma_f32_16x16x32_f16 (three K-blocks, the two cross blocks first, scaled, then hi.hi).
 * |x| must stay below fp16's 65 504 (normed activations, ReLU outputs and weights of the path do; a residual stream would not). */
int gdr_split_f32_f16x2(const float* in, void* out_planes, int64_t rows, int K, int64_t ld_out, void* stream);
/* terms = 6: the form above.  terms = 2: the fp16 x 2 form (A3 / W3 rows [hi | lo'], lda / ldw >= 2 K, K % 128 == 0).  terms = 3: hi.hi + hi.mid + mid.hi only — 16 significand bits, NARROWER than fp32 (error ~3e-5 of mean |c|
 * against fp32's ~8e-6 on the encoder's shapes), half the MFMA work: a measured knob, reported as such, never called fp32. */
int gdr_linear_split_bf16(const void* A3, int64_t lda, const void* W3, int64_t ldw, float* C, int64_t ldc, int64_t M, int N, int K,
                          int terms, int epilogue, const float* bias, const float* residual, int64_t ldr, void* stream);
/* Which tile form gdr_linear_bf16 gives a well-aligned [M,K] x [N,K] launch (host-only, no GPU work; for tests and profiles):
 * 64 / 128 = the 128-column kernel with 64- / 128-row tiles, 192 / 256 = the 256-row tile of that width, 0 = generic core. */
int gdr_linear_bf16_tile_form(int64_t M, int N, int K, int epilogue);

/* T5LayerNorm (modeling_t5.py:164-171): y = w * (x / sqrt(mean(x^2) + eps)) per row, fp32 — the norm every T5 block of the
 * path applies, as an operator of its own.  The quotient is the IEEE division's, bit for bit (the kernels divide a row by its
 * one denominator with a correctly rounded reciprocal and one exact-residual correction per element).  x, y fp32 [rows, d],
 * d % 4 == 0; y may alias x. */
int gdr_t5_layer_norm(const float* x, const float* w, float* y, int64_t rows, int d, float eps, void* stream);

/* y[r] = x[r] / max(||x[r]||_2, eps) — `torch.nn.functional.normalize(rep, dim=-1)` of DensePooler (dense.py:24-25;
 * torch's eps is 1e-12).  x, y fp32 [rows, d]; y may alias x. */
int gdr_l2_normalize(const float* x, float* y, int64_t rows, int d, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * T5 encoder forward — replaces `model.get_encoder()(input_ids, attention_mask=, return_dict=True)
 * .last_hidden_state` (transformers/modeling_t5.py:685-821; called at generation_utils.py:410-411).
 * Weights: pointer table built once from the reference state_dict; wqkv is the
 * row-concatenation [q;k;v] of the three [inner,d] projection matrices.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t vocab_size, d_model, d_kv, d_ff, num_heads, num_layers;
  int32_t rel_buckets, rel_max_distance;
  float eps;
} GdrT5Dims;

typedef struct {
  const float* ln_attn;   /* [d]            block.i.layer.0.layer_norm.weight              */
  const float* wqkv;      /* [3*inner, d]   layer.0.SelfAttention.{q,k,v}.weight row-concat */
  const float* wo;        /* [d, inner]     layer.0.SelfAttention.o.weight                 */
  const float* ln_ff;     /* [d]            layer.1.layer_norm.weight                      */
  const float* wi;        /* [d_ff, d]      layer.1.DenseReluDense.wi.weight               */
  const float* wo_ff;     /* [d, d_ff]      layer.1.DenseReluDense.wo.weight               */
} GdrT5EncLayer;

typedef struct {
  GdrT5Dims dims;
  const float* embed;       /* [vocab, d]     shared.weight                                          */
  const float* rel_bias;    /* [buckets, H]   encoder.block.0...relative_attention_bias.weight       */
  const float* final_ln;    /* [d]            encoder.final_layer_norm.weight                        */
  const GdrT5EncLayer* layers;  /* host array of num_layers entries (device pointers inside)        */
} GdrT5EncoderWeights;

size_t gdr_t5_encoder_workspace_bytes(const GdrT5Dims* dims, int B, int L);
/* ids/mask int64[B,L]; out_hidden fp32[B,L,d]; optional out_pooled fp32[B,d] = hidden[:,0]
 * (CLS pool, main_models.py:102-109 / dense.py:39,50) or NULL. */
int gdr_t5_encoder_forward(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L,
                           float* out_hidden, float* out_pooled, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Ragged form of the same forward: only the token rows that can reach the requested outputs are computed.
 *   - PAD rows are dropped: a sequence whose mask row is a non-empty prefix of ones (right padding, what the reference's
 *     tokenizer emits) keeps its first sum(mask) positions; any other mask row (empty, holes, left padding) keeps all L
 *     positions and its mask, so every case the reference handles is still handled.  Kept rows are bit-identical to
 *     gdr_t5_encoder_forward (GEMM rows are independent; a PAD key adds exp(-1e9 - max) = 0 to a live query's softmax).
 *   - out_hidden (may be NULL): kept rows exact, dropped rows ZERO — they are not the reference's values there, which is
 *     why model.get_encoder() / gdr_t5_encoder_forward stay the exact default; the decode path (cross-attention masks
 *     those keys) and the CLS pool never read them.
 *   - out_pooled (may be NULL; one of the two must be given) = hidden[:,0].  With out_hidden == NULL the last block's
 *     o / wi / wo and the final norm run on the B CLS rows only.
 *   - the live-row count is derived from the mask ON THE DEVICE; kernels read it, the call never synchronises.
 *     live_rows_hint: that count if the host happens to know it, else -1 — a TUNING input: the launcher picks between
 *     kernel forms that are bit-identical to one another (whole tiles, the stream-K tail, the 256-workgroup launch) by the
 *     tile count it implies, and the opt-in profiler (gdr_prof_*) prices flops with it; a wrong or absent hint can cost
 *     speed, never a bit of the result.
 * Batches of fewer than 256 token rows, or d_kv != 64, run the padded form internally (same outputs); below 4 096 token
 * rows the packed form runs on the split-K / stream-K kernel forms the padded forward picks for the same B*L (kept rows
 * still bit-identical) and the pooled-only shortcut of the last block is not taken. */
size_t gdr_t5_encoder_ragged_workspace_bytes(const GdrT5Dims* dims, int B, int L);
int gdr_t5_encoder_forward_ragged(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L,
                                  float* out_hidden, float* out_pooled, int64_t live_rows_hint, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* The ragged form in the bf16 precision mode below (weights as for gdr_t5_encoder_forward_bf16): same contract, kept rows
 * bit-identical to gdr_t5_encoder_forward_bf16.  The packed kernels serve the fused bf16 chain (d_model, inner and d_ff
 * multiples of 64); other shapes run the padded bf16 form internally.  gdr_t5_encoder_ragged_workspace_bytes serves both. */
int gdr_t5_encoder_forward_ragged_bf16(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L,
                                       float* out_hidden, float* out_pooled, int64_t live_rows_hint, void* workspace,
                                       size_t workspace_bytes, void* stream);
/* EXPLORATORY (r06), beside gdr_t5_encoder_forward_ragged, never instead of it: the ragged forward with every linear in the split-bf16
 * form of gdr_linear_split_bf16 (24 significand bits carried through bf16 MFMAs; the fp32 linear's error, not its bits); embedding,
 * T5LayerNorm, attention and the residual stream are the fp32 path's.  The linear weight pointers of `w` (wqkv, wo, wi, wo_ff) point
 * to plane-form bf16 rows [N, gdr_split_row_elems(K)] made by gdr_split_f32_bf16x3.  Needs d_kv = 64 and d_model, inner, d_ff
 * multiples of 64.  Hidden states within 2e-4 of the reference golden like the fp32 path (tests/test_gpu_parity.py). */
size_t gdr_t5_encoder_split_workspace_bytes(const GdrT5Dims* dims, int B, int L);
int gdr_t5_encoder_forward_ragged_split(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L,
                                        float* out_hidden, float* out_pooled, int64_t live_rows_hint, int terms, void* workspace,
                                        size_t workspace_bytes, void* stream);   /* terms: 6 / 3 (weights as bf16 planes) or 2 (weights as fp16 x 2 rows), as gdr_linear_split_bf16 */

/* bf16 precision mode (BASELINE config C5): the SAME structs, but the four linear weights of every layer (wqkv, wo, wi,
 * wo_ff) point to bf16 [N,K] matrices (round-to-nearest-even of the fp32 checkpoint, e.g. gdr_cast_f32_bf16); the
 * `const float*` field type is nominal for them.  Each linear rounds its activation operand to bf16 and accumulates in
 * fp32; with d_kv = 64 (the MFMA attention form) the qkv linear also EMITS q, k, v as bf16 — QK^T and PV take those
 * rounded operands, accumulate in fp32, softmax in fp32; embedding, norms, the residual stream and both outputs stay fp32.  The
 * reference has no such mode (it runs precision=32, main.py:61,91): parity is against the fp32 path within bf16
 * tolerance and against the oracle's bf16 emulation (oracle/t5_ref.py bf16_linears). */
size_t gdr_t5_encoder_bf16_workspace_bytes(const GdrT5Dims* dims, int B, int L);
int gdr_t5_encoder_forward_bf16(const GdrT5EncoderWeights* w, const int64_t* ids, const int64_t* mask, int B, int L,
                                float* out_hidden, float* out_pooled, void* workspace, size_t workspace_bytes,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * Corpus similarity + top-k, fused: never materialises the [B,N] score matrix.
 * replaces `compute_similarity` (dense.py:53-54, encoder.py:128-129: q @ p.T) followed by
 * `Tensor.topk(k, largest=True, sorted=True)` (as at main_models.py:1625).
 *   Q fp32[B,d], D fp32[N,d] (the resident corpus shard) -> out_val fp32[B,k] descending,
 *   out_idx int32[B,k] = row in D + idx_offset.  Ties: higher score first, then lower id.
 *   status (device int32[B], may be NULL): status[q] = 1 if query q's candidate list overflowed — only possible on
 *   degenerate data (tens of thousands of docs tied at / above the sampled threshold, e.g. duplicated embeddings);
 *   its result is then the top-k of a subset.  Re-running those queries with GDR_SIM_EXHAUSTIVE (every score kept,
 *   workspace B*N*8 bytes) is exact for any input.  The C entry point never synchronises, so the re-run is the
 *   caller's job: gdr_amd.ops.sim_topk does it by default (exact_on_overflow=True: one status read-back per call);
 *   latency-critical callers pass exact_on_overflow=False and receive the device status tensor instead.
 * d % 4 == 0, 1 <= k <= 1024, k <= N.
 * ---------------------------------------------------------------------------------------------- */
#define GDR_SIM_EXHAUSTIVE 1
#define GDR_SIM_NO_STREAM 2   /* force the tiled GEMM core even at B <= 32 (A/B testing of the latency-mode kernel) */
size_t gdr_sim_topk_workspace_bytes(int B, int64_t N, int d, int k, int flags);
int gdr_sim_topk(const float* Q, int B, const float* D, int64_t N, int d, int k, int32_t idx_offset,
                 float* out_val, int32_t* out_idx, int32_t* status, int flags, void* workspace,
                 size_t workspace_bytes, void* stream);

/* bf16 corpus / bf16 queries, fp32 accumulate on v_mfma_f32_32x32x16_bf16 (BASELINE config C5: 1M x 768 bf16).
 * Q, D are bf16 (uint16 bit patterns) [B,d] / [N,d]; d % 8 == 0 and d*2 % 128 == 0 is the fast path.  Scores and the
 * top-k semantics are those of gdr_sim_topk applied to the bf16-rounded inputs (products of bf16 values are exact in
 * fp32; only the summation order differs from a CPU fp32 matmul of the same rounded inputs). */
int gdr_sim_topk_bf16(const void* Q, int B, const void* D, int64_t N, int d, int k, int32_t idx_offset,
                      float* out_val, int32_t* out_idx, int32_t* status, int flags, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same fp32 result through a bf16 PRE-FILTER (r05; replaces the same call site, dense.py:53-54 + topk as at main_models.py:1625):
 * the corpus-wide pass runs on the bf16 MFMA path over D_bf16 (= gdr_cast_f32_bf16(D), resident beside D), and only the docs whose
 * bf16-operand score lies within 2*eps_q of the k-th largest one are scored in fp32 (from D) and ranked exactly — with
 *     eps_q = ||q|| * dnorm_max * (2^-7 + 2^-16 + d * 2^-22)   >=   |bf16-operand score - fp32 score|   for every doc,
 * that band provably contains the fp32 top-k including every doc tied at the cut (derivation: csrc/sim_topk.hip), so out_val / out_idx
 * are the top-k of the fp32 scores for every input, ties as in gdr_sim_topk (higher score, then lower id); the values are fp32 dot
 * products of the same operands in another summation order.  dnorm_max: the largest ||D[r]||_2 (sqrt of gdr_row_norm2_max's result).
 * status as in gdr_sim_topk (1 = an overflowed list: re-run that query with gdr_sim_topk).  d % 8 == 0, d <= 1024. */
size_t gdr_sim_topk_prefilter_workspace_bytes(int B, int64_t N, int d, int k);
int gdr_sim_topk_prefilter(const float* Q, int B, const float* D, const void* D_bf16, float dnorm_max, int64_t N, int d, int k,
                           int32_t idx_offset, float* out_val, int32_t* out_idx, int32_t* status, void* workspace,
                           size_t workspace_bytes, void* stream);
/* out_dev[0] = max over rows of ||D[r]||_2^2 (device float; the call zeroes it first).  d % 4 == 0. */
int gdr_row_norm2_max(const float* D, int64_t N, int d, float* out_dev, void* stream);
/* fp32 -> bf16, round-to-nearest-even (v_cvt_pk_bf16_f32), n % 4 == 0. */
int gdr_cast_f32_bf16(const float* in, void* out_bf16, int64_t n, void* stream);

/* Merge of per-shard top-k lists after the RCCL all-gather (no reference analogue):
 * vals/idx [G,B,k] (shard-major) -> [B,k]; same tie rule, so every rank computes identical output. */
int gdr_topk_merge(const float* vals, const int32_t* idx, int G, int B, int k, float* out_val, int32_t* out_idx,
                   void* stream);
/* The wire form of a per-shard result, so that the exchange is ONE collective: per query row k+1 entries of
 * 8 bytes, entry j < k = {fp32 score, int32 id}, entry k = {0, status[q]} (status may be NULL = 0).
 * gdr_topk_pack writes pairs[B, k+1]; gdr_topk_merge_packed merges pairs[G, B, k+1] (shard-major, as an all-gather /
 * all-to-all lays them out) into out_val/out_idx [B,k] with gdr_topk_merge's tie rule and, when out_status is given,
 * out_status[q] = 1 if any shard flagged query q (its list was the top-k of a subset, see gdr_sim_topk). */
int gdr_topk_pack(const float* vals, const int32_t* idx, const int32_t* status, int B, int k, void* pairs, void* stream);
int gdr_topk_merge_packed(const void* pairs, int G, int B, int k, float* out_val, int32_t* out_idx,
                          int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------------------------------
 * In-cluster rerank — replaces main_models.py:1574-1637, block-diagonal only.
 *   q fp32[B,d]; D fp32[N,d]; beam_scores fp32[B,R] (length-penalised); candidate lists in one of two layouts:
 *     cand_stride == 0: cand_offsets int32[B*R+1], ONE CSR over the decoded clusters (query-major, beam order) into
 *                       cand_ids int32[...] — the reference's concatenation (main_models.py:1441-1443);
 *     cand_stride  > 0: a block per query — cand_offsets int32[B][R+1] relative to the block ([b][0] = 0), cand_ids
 *                       int32[B][cand_stride]: what gdr_cluster_candidates emits and what ranks exchange (fixed size);
 *   alphas fp32[A]; out_val fp32[B,A,k], out_idx int32[B,A,k] (doc ids; -1 / -inf padding when a
 *   query has fewer than k candidates — the reference raises there).  func: 0 tanh, 1 sigmoid.
 *   max_cand: upper bound of any query's candidate count (num beams x largest cluster; <= 8192) — sizes the score
 *   scratch and the LDS sort buffer; candidates past it are ignored.  The CSR may live on the device only
 *   (gdr_cluster_candidates below): nothing here needs its contents on the host.
 *   Row-sharded corpus (GDR mode): D points to rows [doc_lo, doc_hi) of the corpus; candidates outside
 *   the range are skipped.  Unsharded: doc_lo = 0, doc_hi = N.  With GDR_RERANK_POSITIONS out_idx holds the candidate's
 *   POSITION in its query's list (0-based) instead of the doc id: merging the per-shard lists by "higher score, then
 *   lower position" (gdr_topk_merge_packed over B*A rows) reproduces the unsharded list bit for bit, because a
 *   candidate's score does not depend on which shard computed it.
 *   workspace: gdr_rerank_workspace_bytes(B, max_cand).
 * ---------------------------------------------------------------------------------------------- */
#define GDR_RERANK_POSITIONS 1
size_t gdr_rerank_workspace_bytes(int B, int max_cand);
int gdr_rerank_topk(const float* q, const float* D, int d, const int32_t* cand_offsets, const int32_t* cand_ids,
                    const float* beam_scores, int B, int R, const float* alphas, int A, int k, int func,
                    float* out_val, int32_t* out_idx, int max_cand, int cand_stride, int32_t doc_lo, int32_t doc_hi,
                    int flags, void* workspace, size_t workspace_bytes, void* stream);
/* The same over a bf16 corpus (BASELINE config C5: 1M x 768 bf16): rows are gathered as bf16 and widened (exact), the dot
 * product is the same fp32 fmaf chain against the fp32 query, same keys, same order — i.e. gdr_rerank_topk applied to the
 * bf16-rounded corpus.  The corpus is never up-cast as a whole. */
int gdr_rerank_topk_bf16(const float* q, const void* D_bf16, int d, const int32_t* cand_offsets, const int32_t* cand_ids,
                         const float* beam_scores, int B, int R, const float* alphas, int A, int k, int func,
                         float* out_val, int32_t* out_idx, int max_cand, int cand_stride, int32_t doc_lo, int32_t doc_hi,
                         int flags, void* workspace, size_t workspace_bytes, void* stream);

/* Decoded docid rows -> clusters -> candidate CSR on the device — replaces `decode_token` + the `id_mapping` dict lookup +
 * the candidate concatenation of main_models.py:1398,1441-1443 (the reference walks Python strings per beam).
 * decode_token (main_models.py:322-346) drops START, cuts at the first EOS and prints token - (i*V + 2) per position; a row
 * WITHOUT EOS is printed whole, START included.  That string is a one-to-one image of the token body, so
 * `id_mapping[string]` is an exact-match lookup of the body: `keys[c]` holds cluster c's body tokens (the tokens whose
 * decode is the cluster's name), `slots` an open-addressing table (linear probing) over gdr_cluster_key_hash(body).
 * A body that matches no cluster gives an empty segment, as the reference's KeyError path does.
 *   out_ids int64[B*R, max_length] = gdr_t5_generate's out_ids (untrimmed);
 *   cluster_of int32[B*R] (out: cluster index or -1); cand_offsets int32[B][R+1] and cand_ids int32[B][cand_stride] (out):
 *   the per-query block layout of gdr_rerank_topk — members in cluster order, beams in order = the reference's
 *   concatenation order within a query; entries past cand_stride are dropped (size it num_beams * largest cluster). */
typedef struct {
  int32_t n_clusters;
  int32_t key_len;          /* ints per stored body (>= the longest body) */
  int32_t table_size;       /* power of two > n_clusters */
  const int32_t* slots;     /* device int32 [table_size]: cluster index or -1 */
  const int32_t* keys;      /* device int32 [n_clusters, key_len] */
  const int32_t* key_lens;  /* device int32 [n_clusters] */
  const int32_t* offsets;   /* device int32 [n_clusters + 1]  CSR of the member doc ids */
  const int32_t* members;   /* device int32 [N] */
} GdrClusterIndex;
uint64_t gdr_cluster_key_hash(const int32_t* tokens_host, int len);   /* host routine, the hash the device lookup uses */
int gdr_cluster_candidates(const GdrClusterIndex* ci, const int64_t* out_ids, int B, int R, int max_length,
                           int32_t* cluster_of, int32_t* cand_offsets, int32_t* cand_ids, int cand_stride, void* stream);

/* The exchange row of the SHARDED in-cluster rerank (gdr_amd/dist.py ShardedIndex.rerank_own; the arithmetic being sharded
 * is main_models.py:1434-1462,1574-1637, the per-GPU layout follows Data_process/NQ_dataset/bert/bert_NQ.sh:5-12): per query
 * ONE int32 row  wire[b] = { q fp32[d] | beam_scores fp32[R] | cand_offsets int32[R+1] | cand_ids int32[cand_stride] },
 * so that every rank's queries + candidate blocks travel in ONE fixed-size all-gather.  gdr_rerank_wire_pack builds
 * wire[B][d + 2R + 1 + cand_stride]; gdr_rerank_wire_unpack splits the gathered rows back into the four contiguous arrays
 * gdr_rerank_topk reads; gdr_rerank_positions_to_ids maps the merged candidate POSITIONS (GDR_RERANK_POSITIONS lists after
 * gdr_topk_merge_packed) of query b = i / per_query back to doc ids through the query's own candidate block
 * (position < 0, the "fewer than k candidates" padding, stays -1). */
int gdr_rerank_wire_pack(const float* q, const float* beam_scores, const int32_t* cand_offsets, const int32_t* cand_ids,
                         int B, int d, int R, int cand_stride, int32_t* wire, void* stream);
int gdr_rerank_wire_unpack(const int32_t* wire, int B, int d, int R, int cand_stride, float* q, float* beam_scores,
                           int32_t* cand_offsets, int32_t* cand_ids, void* stream);
int gdr_rerank_positions_to_ids(const int32_t* pos, const int32_t* cand_ids, int B, int per_query, int cand_stride,
                                int32_t* out_ids, void* stream);

/* Device-side faults that do not abort the process.  The stream-K form of the linear GEMM hands raw accumulators from one
 * workgroup to the next behind a bounded spin; if that spin ever runs out (the launch's workgroups were not co-resident)
 * the launch completes with INVALID output and raises a sticky, process-wide fault word.  While it is raised every
 * stream-K launch fails with GDR_EHIP; gdr_device_fault_pending() returns 1 (message in gdr_last_error()) so that a
 * caller can check it wherever it synchronises with the device before trusting a result; gdr_device_fault_clear()
 * acknowledges it.  Nothing clears it implicitly.  gdr_device_fault_inject_for_tests() raises it by hand (host only). */
int gdr_device_fault_pending(void);
void gdr_device_fault_clear(void);
void gdr_device_fault_inject_for_tests(void);

/* T5 relative-position buckets (transformers/modeling_t5.py:242-288) for relative_position =
 * key_pos - query_pos, written to a HOST int32[qlen*klen] table; the attention kernels use the same
 * host routine, so tests pin it bit-exact against the reference. */
int gdr_t5_relative_bucket_table(int bidirectional, int num_buckets, int max_distance, int qlen, int klen,
                                 int32_t* out_host);

/* ------------------------------------------------------------------------------------------------
 * Doc tower — replaces `EncoderModel.forward(passage=...)` (main_models.py:79-89) = DPRContextEncoder
 * (transformers/modeling_dpr.py:146-191) over BertModel (transformers/modeling_bert.py); pooled = hidden[:,0].
 * Producer of the corpus matrix D (Data_process/NQ_dataset/bert/bert.py:69-71).  L <= 128 (encoder_max_len).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float *wqkv, *bqkv;      /* [3d,d],[3d]  attention.self.{query,key,value} row-concat   */
  const float *wo, *bo;          /* attention.output.dense                                       */
  const float *ln1_w, *ln1_b;    /* attention.output.LayerNorm                                   */
  const float *wi, *bi;          /* intermediate.dense (GeLU, erf form)                          */
  const float *wo2, *bo2;        /* output.dense                                                 */
  const float *ln2_w, *ln2_b;    /* output.LayerNorm                                             */
} GdrBertLayer;

typedef struct {
  int32_t vocab_size, d_model, num_heads, d_ff, num_layers, max_pos, type_vocab;
  float eps;
  const float *word_emb, *pos_emb, *type_emb, *emb_ln_w, *emb_ln_b;
  const GdrBertLayer* layers;    /* host array [num_layers] */
} GdrBertWeights;

size_t gdr_bert_encoder_workspace_bytes(const GdrBertWeights* w, int B, int L);
/* ids/mask int64[B,L], token_type_ids int64[B,L] or NULL (zeros); out_hidden fp32[B,L,d] and/or out_pooled fp32[B,d]. */
int gdr_bert_encoder_forward(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                             const int64_t* token_type_ids, int B, int L, float* out_hidden, float* out_pooled,
                             void* workspace, size_t workspace_bytes, void* stream);
/* The same forward over the token rows that can reach the requested outputs (r06; the T5 side's gdr_t5_encoder_forward_ragged for the
 * doc tower).  The reference pads a batch of passages to its longest member (Data_process/NQ_dataset/bert/bert.py:69-71) and computes
 * every position (modeling_bert.py:210-285); PAD keys carry weight exp(-1e9 - max) = 0 and PAD rows never reach
 * pooled = sequence_output[:, 0] (modeling_dpr.py:178-181).  Here the live rows are packed, attention runs per sequence on its own
 * length, and a pooled-only call (out_hidden == NULL) carries only the CLS rows through the last block.  Kept rows are BIT-IDENTICAL to
 * gdr_bert_encoder_forward; PAD rows of out_hidden are zero.  Masks that are not a prefix of ones keep all their positions.
 * live_rows_hint: the number of kept rows if the caller knows it (-1 otherwise) — prices the profiler's flops, never a result input.
 * Small batches / head widths other than 64 run the padded forward (same results). */
size_t gdr_bert_encoder_ragged_workspace_bytes(const GdrBertWeights* w, int B, int L);
int gdr_bert_encoder_forward_ragged(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                                    const int64_t* token_type_ids, int B, int L, float* out_hidden, float* out_pooled,
                                    int64_t live_rows_hint, void* workspace, size_t workspace_bytes, void* stream);
/* ... in the bf16 precision mode (BASELINE config C5 keeps its corpus in bf16; the reference itself runs precision = 32): every linear
 * weight pointer of `w` (wqkv, wo, wi, wo2) points to bf16 data (RNE of the fp32 tensor, gdr_cast_f32_bf16), biases / LayerNorm /
 * embeddings stay fp32; bf16 operands, fp32 accumulate, fp32 residual stream and norms; q, k, v are emitted as bf16 for the bf16-MFMA
 * attention.  The attention's 1/sqrt(dh) must be FOLDED INTO THE q ROWS of wqkv and bqkv by the caller (dh = 64: the factor 1/8 is a
 * power of two, the fold is exact).  Needs head width 64 and d_model, d_ff multiples of 64.  Same workspace size. */
int gdr_bert_encoder_forward_ragged_bf16(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                                         const int64_t* token_type_ids, int B, int L, float* out_hidden, float* out_pooled,
                                         int64_t live_rows_hint, void* workspace, size_t workspace_bytes, void* stream);
/* EXPLORATORY (r06), beside gdr_bert_encoder_forward_ragged: every linear in the fp16 x 2 split form of gdr_linear_split_bf16 (terms = 2:
 * fp32 operands carried as [fp16 hi | fp16 (x - hi) * 2^11], fp32-level error on the fp16 MFMA path); embeddings, LayerNorm, attention,
 * GeLU and the residual stream are the fp32 path's.  The linear weight pointers of `w` point to fp16 plane rows [N, 2 K]
 * (gdr_split_f32_f16x2); biases stay fp32.  Needs head width 64 and d_model, d_ff multiples of 128.  Same workspace size. */
int gdr_bert_encoder_forward_ragged_split(const GdrBertWeights* w, const int64_t* ids, const int64_t* mask,
                                          const int64_t* token_type_ids, int B, int L, float* out_hidden, float* out_pooled,
                                          int64_t live_rows_hint, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Docid beam decode — replaces `_generate_beam_search` (transformers/generation_utils.py:629-921, with
 * `BeamHypotheses` :1052-1099) driving `T5ForConditionalGeneration.forward`'s decode branch
 * (transformers/modeling_t5.py:1529-1646) as GDR calls it (main_models.py:1380-1397): greedy beams,
 * early_stopping=False, use_cache=False semantics, positional vocabulary mask.
 *
 * Device-resident: no host synchronisation between steps (the reference syncs per candidate via .item()).
 * Same arithmetic, restructured: K/V caches addressed through a beam-ancestor table instead of
 * recomputing the whole prefix every step; cross-attention K/V projected once per query instead of once per
 * beam row per step; the adaptor's single-key cross-attention folded into a per-layer constant; the
 * adaptor_linear head evaluated for the last position and its V+1 unmasked columns only.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* ln_self;   /* [d]           decoder.block.i.layer.0.layer_norm.weight              */
  const float* wqkv;      /* [3*inner, d]  layer.0.SelfAttention.{q,k,v}.weight row-concat        */
  const float* wo;        /* [d, inner]    layer.0.SelfAttention.o.weight                         */
  const float* ln_cross;  /* [d]           layer.1.layer_norm.weight                              */
  const float* wq_c;      /* [inner, d]    layer.1.EncDecAttention.q.weight                       */
  const float* wkv_c;     /* [2*inner, d]  layer.1.EncDecAttention.{k,v}.weight row-concat        */
  const float* wo_c;      /* [d, inner]    layer.1.EncDecAttention.o.weight                       */
  const float* ln_ff;     /* [d]           layer.2.layer_norm.weight                              */
  const float* wi;        /* [d_ff, d]     layer.2.DenseReluDense.wi.weight                       */
  const float* wo_ff;     /* [d, d_ff]     layer.2.DenseReluDense.wo.weight                       */
} GdrT5DecLayer;

typedef struct {          /* one torch.nn.TransformerDecoderLayer (post-LN, ReLU), modeling_t5.py:1241-1244 */
  const float *in_w, *in_b;        /* self_attn.in_proj_{weight[3d,d],bias[3d]}                              */
  const float *out_w, *out_b;      /* self_attn.out_proj                                                    */
  const float *ln1_w, *ln1_b;
  const float* cross_const;        /* [d] = multihead_attn.out_proj(v_proj(adaptor_embeddings)): a single-key
                                      softmax is exactly 1, so the cross-attention output is this constant  */
  const float *ln2_w, *ln2_b;
  const float *lin1_w, *lin1_b;    /* [aff,d],[aff] */
  const float *lin2_w, *lin2_b;    /* [d,aff],[d]   */
  const float *ln3_w, *ln3_b;
} GdrAdaptorLayer;

typedef struct {
  GdrT5Dims dims;                  /* num_layers = num_decoder_layers; vocab_size = decode_vocab_size (Vd) */
  int32_t out_vocab;               /* V  (--output_vocab_size / --kary)                                     */
  int32_t max_out_len;             /* --max_output_length; Vd = V*max_out_len + 2                           */
  int32_t adaptor_layers, adaptor_nhead, adaptor_ff;
  float adaptor_eps;
  const float* dec_embed;          /* [Vd, d]   decode_embeddings.weight (= decoder.embed_tokens = lm_head) */
  const float* self_rel_bias;      /* [buckets,H] decoder.block.0.layer.0.SelfAttention.relative_attention_bias */
  const float* cross_rel_bias;     /* [buckets,H] decoder.block.0.layer.1.EncDecAttention.relative_attention_bias */
  const float* final_ln;           /* [d]       decoder.final_layer_norm.weight                              */
  const GdrT5DecLayer* layers;     /* host array [num_layers]                                                */
  const GdrAdaptorLayer* alayers;  /* host array [adaptor_layers]                                            */
  /* head slices: position p in [0, max_out_len-1), local column c in [0, V] (c < V: token p*V+2+c; c == V: EOS=1)
   *   head_w[p][c][i][k] = adaptor_linear.weight[i*Vd + token(p,c), k]      (modeling_t5.py:1634-1636)
   *   head_e[p][c][i]    = lm_head.weight[token(p,c), i]                                                    */
  const float* head_w;             /* [(max_out_len-1), V+1, d, d] */
  const float* head_e;             /* [(max_out_len-1), V+1, d]    */
} GdrT5DecoderWeights;

/* Optional trie constraint = the NCI semantics of the reference's earlier, un-imported generation_utils_previous.py:
 * 714-729 (`--tree 1`; the shipped generation_utils.py has the block commented out): after log_softmax,
 * -inf on every token that is not a child of the trie node reached by the beam's prefix (trie = TreeBuilder of
 * main_models.py:112-151 over the docids); a prefix that left the tree may only emit EOS.  Flattened for the device:
 * child[node*V + c] = next node (or -1) for digit c at the node's depth, eos_ok[node] = 1 if EOS is a child;
 * node 0 is the root.  Pass NULL for the shipped behaviour (positional vocabulary mask only). */
typedef struct {
  const int32_t* child;   /* device int32 [n_nodes, V] */
  const int32_t* eos_ok;  /* device int32 [n_nodes]    */
  int32_t n_nodes;
  int32_t V;              /* digits per level the table was built for; must equal the head's out_vocab (checked) */
} GdrTrie;

/* Prefix table (optional, exact): the adaptor chain and the head matrix W = adaptor_linear(adaptor(prefix))[V+1 columns] +
 * lm_head depend only on the decoded TOKEN PREFIX, never on the query (modeling_t5.py:1618-1639:
 * decode_embeddings(decoder_input_ids) -> adaptor(memory = adaptor_embeddings) -> adaptor_linear).  For every node of the
 * corpus' docid trie (nodes in BREADTH-FIRST order, root = 0; the first n_table nodes = all nodes of depth < n_levels) the
 * table holds, per adaptor layer, the in_proj output (q,k,v) of the node's position and the finished head matrix.  With
 * it, gdr_t5_generate reads W[node] for a beam row whose prefix is a table node; rows whose prefix left the trie (or is
 * deeper than the table) are compacted on the device and run the adaptor + head GEMM for themselves, attending over
 * ancestors' K/V taken from the table or computed earlier in the call.  Built once per (weights, corpus): a pure function
 * of those two, nothing query-dependent is ever stored.  Size: n_table * (adaptor_layers*3*d + (V+1)*d) floats
 * (320k-doc corpus, t5-base: 27 598 nodes, 3.6 GB). */
typedef struct {
  const int32_t* child;   /* device int32 [n_nodes, V]: the trie, as GdrTrie.child                                   */
  int32_t n_nodes, V;
  int32_t n_table;        /* nodes with entries: the first n_table nodes (breadth-first order)                       */
  const float* kv;        /* device [adaptor_layers][n_table][3*d]                                                   */
  const float* W;         /* device [n_table][V+1][d]   (adaptor_linear slice + lm_head rows of the node's position) */
  int32_t complete_levels; /* c >= 1: every one of the V^s prefixes of length s is a table node for all s < c (level 0 = the
                            * root always is).  A beam row of decode step s sits on a prefix of length s, so at steps s < c
                            * no row can miss and gdr_t5_generate does not enqueue the miss-row chain at all (1 = step 0 only;
                            * a value larger than the truth gives WRONG logits for the rows that do miss)                    */
} GdrPrefixTable;

size_t gdr_t5_prefix_table_workspace_bytes(const GdrT5DecoderWeights* w, int max_level_nodes);
/* level_off: HOST int32 [n_levels+1], node range of each depth (level 0 = the root alone);
 * node_tok: device int64 [n_table], the token that leads to the node (root: START = 0);
 * node_anc: device int32, level after level, for each node of depth s its s+1 ancestors root..self (node ids);
 * kv, W: the table storage (device), filled by this call.  n_levels <= max_output_length - 1. */
int gdr_t5_prefix_table_build(const GdrT5DecoderWeights* w, int n_levels, const int32_t* level_off, const int64_t* node_tok,
                              const int32_t* node_anc, float* kv, float* W, void* workspace, size_t workspace_bytes,
                              void* stream);

size_t gdr_t5_generate_workspace_bytes(const GdrT5DecoderWeights* w, int B, int L, int num_beams, int max_length);
/* enc_hidden fp32[B,L,d] (NOT beam-expanded), enc_mask int64[B,L].
 * out_ids int64[B*nret, max_length] (hypothesis tokens incl. START, then EOS if it fits, then PAD),
 * out_len int32[B*nret] (= len(hyp), EOS excluded), out_scores fp64[B*nret] (sum_logprobs / len^length_penalty,
 * computed in double like the reference's Python floats).  Optional trace (NULL to skip):
 * step_scores fp32[max_length-1, B, 2R], step_tokens int32[...] = the per-step topk(2R) (generation_utils.py:775).
 * 2 <= num_beams <= 256, num_return_sequences <= num_beams, max_length <= max_out_len. */
int gdr_t5_generate(const GdrT5DecoderWeights* w, const float* enc_hidden, const int64_t* enc_mask, int B, int L,
                    int num_beams, int max_length, double length_penalty, int num_return_sequences,
                    const GdrTrie* trie, const GdrPrefixTable* prefix_table /* NULL: compute every row */, int64_t* out_ids,
                    int32_t* out_len, double* out_scores, float* step_scores, int32_t* step_tokens, void* workspace,
                    size_t workspace_bytes, void* stream);

/* bf16 precision mode of the decode path (BASELINE config C5; the reference itself runs precision=32, main.py:61,91): the
 * SAME structs, but every linear weight — GdrT5DecLayer.{wqkv, wo, wq_c, wkv_c, wo_c, wi, wo_ff},
 * GdrAdaptorLayer.{in_w, out_w, lin1_w, lin2_w} and head_w — points to bf16 data (round-to-nearest-even of the fp32
 * checkpoint); the `const float*` field type is nominal for them.  Each linear rounds its activation operand to bf16 and
 * accumulates in fp32; embeddings, norms, attention, biases, the residual stream, the head dot product (h · W), log-softmax
 * and all beam arithmetic stay fp32, hypothesis scores fp64.  A prefix table for this mode is built by the _bf16 builder
 * (same rounding points as the in-call computation); its storage stays fp32.  Parity: against the oracle's emulation of
 * exactly these rounding points (oracle/t5_ref.py bf16_linears) and within bf16 tolerance of the fp32 path.
 * Workspace: gdr_t5_generate_workspace_bytes / gdr_t5_prefix_table_workspace_bytes serve both modes. */
int gdr_t5_generate_bf16(const GdrT5DecoderWeights* w, const float* enc_hidden, const int64_t* enc_mask, int B, int L,
                         int num_beams, int max_length, double length_penalty, int num_return_sequences,
                         const GdrTrie* trie, const GdrPrefixTable* prefix_table, int64_t* out_ids, int32_t* out_len,
                         double* out_scores, float* step_scores, int32_t* step_tokens, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Early exit of the step loop — `if all(done): break`, generation_utils.py:836-838, without a host synchronisation.  When the
 * last query of a call becomes done (BeamHypotheses.is_done, :827-829) the beam bookkeeping kernel (a) clears a device word
 * that the 64x64-tile linears and the miss-row chain of every step already enqueued look at — they exit at once — and
 * (b) stores the call's epoch in host-mapped memory, which the host reads before it enqueues the next step and then stops.
 * The steps that still run change nothing (done queries only pad, :786-794): outputs are bit-identical to running all
 * max_length - 1 steps.  Not taken while the per-step trace is requested; under graph capture only (a) applies.  Random
 * weights never finish early; a trained model does after the docid's length + 1 steps, a trie-constrained call always does.
 * Returns how many generate calls of this process left their loop on the host side (monitoring / tests). */
int64_t gdr_t5_generate_early_exits(void);
/* The device half of the same exit, observable: the decode step (cur_len, 1-based) at which the LAST query of the most recent
 * gdr_t5_generate call of this process became done, written by the beam bookkeeping kernel into host-mapped memory; 0 when
 * that call ran to max_length without every query finishing (or ran with the per-step trace).  Read it after the call's
 * stream has been synchronised.  Deterministic (unlike the host-side counter above, which races with the GPU by design) —
 * for STRICTLY SERIAL calls only: "the most recent call" is the most recently STARTED one of the process, so with several calls
 * in flight (GDRRetriever.validation_steps at depth >= 2, several host threads) or after a call that took no epoch (per-step trace
 * requested) it reports another call's step, or 0.  A monitoring / test hook, not part of the result. */
int gdr_t5_generate_last_done_step(void);
int gdr_t5_prefix_table_build_bf16(const GdrT5DecoderWeights* w, int n_levels, const int32_t* level_off,
                                   const int64_t* node_tok, const int32_t* node_anc, float* kv, float* W, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* The same device beam search driven by a logit table instead of the model (teacher forcing):
 * logits(prefix) = table[b, pos, last_token, :] (fp32 [B, max_length, Vd, Vd]) with the positional mask.
 * Exercises EOS / early-done / eviction paths that random weights never reach. */
size_t gdr_beam_search_table_workspace_bytes(int B, int num_beams, int max_length, int out_vocab);
int gdr_beam_search_table(const float* table, int B, int out_vocab, int num_beams, int max_length,
                          double length_penalty, int num_return_sequences, const GdrTrie* trie, int64_t* out_ids,
                          int32_t* out_len, double* out_scores, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDR_HIP_H */
