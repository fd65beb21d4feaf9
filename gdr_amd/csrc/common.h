// Shared host-side helpers for libgdr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/gdr_hip.h"

namespace gdr {

void set_error(const char* fmt, ...);

#define GDR_CHECK_ARG(cond, ...)                \
  do {                                          \
    if (!(cond)) {                              \
      gdr::set_error(__VA_ARGS__);              \
      return GDR_EINVAL;                        \
    }                                           \
  } while (0)

void count_launch();
// every kernel launch of the library passes here: a relaxed process-wide count (gdr_launch_count: the bench reports launches per call)
#define GDR_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    gdr::count_launch();                                                         \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      gdr::set_error("%s: %s", what, hipGetErrorString(e__));                    \
      return GDR_EHIP;                                                           \
    }                                                                            \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (kernel, device): the
// attribute lives on the device's code object, so a process that drives several GPUs must set it on each of them.
// Thread-safe.  Returns GDR_OK or GDR_EHIP (message set).
int ensure_dyn_lds(const void* kernel, int bytes, const char* what);

// ---- opt-in launch profiler (gdr_prof_* in include/gdr_hip.h): hipEvent pairs around the launches of one
// kernel class, recorded on the stream the kernel runs on.  Off by default: no events, no state touched.
enum ProfClass { PROF_LINEAR = 0, PROF_SIM_SAMPLE = 1, PROF_SIM_FILTER = 2, PROF_ATTENTION = 3, PROF_NORM = 4,
                 PROF_SELECT = 5, PROF_RERANK = 6, PROF_REDUCE = 7, PROF_NCLASS = 8 };
struct ProfScope {
  int slot;
  hipStream_t stream;
  ProfScope(int cls, double work, hipStream_t s);
  ~ProfScope();
};

// ---- internal launchers shared between translation units -------------------------------------
// Similarity epilogue parameters of the GEMM core (see gemm_f32.hip / sim_topk.hip).
// The per-query list counters live 128 bytes apart: atomics on words of ONE cache line are serialised by that line's L2 channel —
// with the B counters packed (32 per line) the filter passes' appends queued behind one another (bf16 stream pass at 32 queries:
// 139 us with appends against 89 us without; 8 192 flush atomics on one line).
constexpr int CNT_STRIDE = 32;
struct SimEpilogue {
  float* cand_val;        // [B][cap]
  int32_t* cand_idx;      // [B][cap]
  int32_t* cand_cnt;      // [B][CNT_STRIDE]: counter of query q at cand_cnt[q * CNT_STRIDE]
  const float* thr;       // [B]
  int32_t* status;        // [1] or null
  int32_t cap;
  int32_t tile_stride;    // sample tiles are the m-tiles with index % tile_stride == 0
  int32_t mode;           // 1: sample tiles, store every score; 2: other tiles, append score >= thr
};

// Scratch of the stream-K form of the persistent linear kernel (gemm_f32.hip), owned by the caller: one per stream of
// launches.  `flag` must be zeroed once (hipMemsetAsync on the same stream) before the first launch that uses it; every
// launch then takes the next epoch, so flags never need clearing between launches.
struct StreamK {
  float* part;     // device [512][128*128]: raw accumulators handed from a workgroup to its successor
  int32_t* flag;   // device [512]
  int32_t epoch;   // host-side launch counter
  // decode only (may be null): a device word that gdr_t5_generate's beam bookkeeping clears when every query of the call is
  // done (generation_utils.py:836-838 `if all(done): break`); the 64x64-tile linears of the steps still enqueued exit at once
  const int32_t* live = nullptr;
};
constexpr size_t STREAMK_PART_BYTES = (size_t)512 * 128 * 128 * 4;
constexpr size_t STREAMK_BYTES = STREAMK_PART_BYTES + 4096;

int launch_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M,
                      int N, int K, int epilogue, const float* bias, const float* residual, int64_t ldr,
                      hipStream_t stream);
int launch_linear_f32_ws(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M,
                         int N, int K, int epilogue, const float* bias, const float* residual, int64_t ldr,
                         float* splitk_ws, size_t splitk_ws_bytes, hipStream_t stream, StreamK* sk = nullptr,
                         const int64_t* m_dev = nullptr, int64_t prof_rows = -1);
// Decode-time linear over a device-side row count (*m_dev <= M_max live rows; grids sized for M_max): the same kernel
// choice as launch_linear_f32_ws makes for M_max — small-tile / split-K forms included, so its results for a row equal
// what launch_linear_f32_ws(M = M_max) gives that row.
int launch_linear_f32_ws_dev(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M_max,
                             const int64_t* m_dev, int N, int K, int epilogue, const float* bias, const float* residual,
                             int64_t ldr, float* splitk_ws, size_t splitk_ws_bytes, hipStream_t stream,
                             StreamK* sk = nullptr);
int launch_linear_f32_dev(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M_max,
                          const int64_t* m_dev, int N, int K, int epilogue, const float* bias, const float* residual,
                          int64_t ldr, int64_t prof_rows, hipStream_t stream, StreamK* sk = nullptr);
// docs D[N,d] take the GEMM's row role, queries Q[B,d] the column role: a lane owns one query.
int launch_sim_gemm(const void* D, int64_t N, const void* Q, int B, int d, const SimEpilogue& ep, bool bf16,
                    hipStream_t stream);

int launch_linear_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                       int K, int epilogue, const float* bias, const float* residual, int64_t ldr, hipStream_t stream,
                       const int64_t* m_dev = nullptr);
// A norm fused behind a split-K linear whose output rows are whole model rows (gemm_small.hip splitk_reduce_norm_kernel).
struct NormEpilogue {
  int kind;                 // 1 RMS (w1), 2 LayerNorm (w1,b1), 3 LayerNorm(w1,b1) -> + addv -> LayerNorm(w2,b2)
  const float *w1, *b1, *w2, *b2, *addv;
  float eps;
  float* Y;                 // normed rows [M, ldy]
  int64_t ldy;
  int y16_only = 0;         // bf16 mode: the normed rows feed bf16 linears only — when their bf16 image is written, the fp32 copy is not
};
// Split-K slabs left UN-reduced for a consumer that sums them itself (the decode cross-attention reads its q rows so and the
// reduction launch disappears): element (m, n) = sum over s < S, in that order, of
//   part[((m/64 * tiles_n + n/64) * S + s) * 4096 + (m%64)*64 + n%64].   S == 1: nothing was split, C holds the result.
struct SlabRef {
  const float* part;
  int S, tiles_n;
};
int launch_linear_f32_small(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                            int K, int has_bias, int has_residual, int act, const float* bias, const float* residual,
                            int64_t ldr, float* ws, size_t ws_bytes, hipStream_t stream, const int64_t* m_dev = nullptr,
                            const NormEpilogue* ne = nullptr,  // with ne: returns 2 if the fused form does not apply
                            SlabRef* slabs = nullptr,          // with slabs (no epilogue allowed): the reduction is left to the caller
                            const int32_t* live = nullptr);    // StreamK::live
int launch_linear_bf16_headdot(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, const int64_t* m_dev, int N, int K,
                               const float* h, int64_t ldh, const int32_t* rows_map, const float* E, int dot_d, float scale,
                               float* partial, hipStream_t stream);
int linear_bf16_tile_form(int64_t M, int N, int K, int has_residual, int out_bf16);
int launch_linear_bf16_glds(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                            int K, int has_bias, int has_residual, int act, const float* bias, const float* residual,
                            int64_t ldr, int out_bf16, hipStream_t stream, const int64_t* m_dev = nullptr,
                            int split = 0);  // split = 6 / 3: K is K0, rows of A / W hold the three bf16 planes [hi | mid | lo]; 2: fp16 x 2 rows [hi | lo'] (gemm_bf16.hip)
// x -> [bf16(x) | bf16(x - hi) | bf16(x - hi - mid)] per row (ld_out >= 3 K elements); rows_dev may be null
int split_row_elems(int K, int f16x2 = 0);  // row length (elements) of a plane-form operand: 3 K (bf16 x 3) or 2 K (fp16 x 2), rounded up to 64
int launch_split_f32_bf16x3(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int K, const int64_t* rows_dev,
                            hipStream_t stream, int f16x2 = 0);  // f16x2: rows [fp16 hi | fp16 (x - hi) * 2^11]
int launch_sim_bf16_glds(const void* D, int64_t N, const void* Q, int B, int d, const SimEpilogue& ep, int64_t n_doc_tiles,
                         hipStream_t stream);
int launch_cast_f32_bf16(const float* in, void* out_bf16, int64_t n, hipStream_t stream);

// latency-mode similarity (B <= 32, fp32): stationary queries in LDS, corpus streamed from HBM (sim_stream.hip)
bool sim_stream_supported(int B, int d, bool bf16);
int launch_sim_stream(const void* D, int64_t N, const void* Q, int B, int d, const SimEpilogue& ep, bool bf16, hipStream_t stream);

// Sums / maxima over the 64 lanes as the xor butterfly 32, 16, 8, 4, 2, 1 (every lane ends with the total).  The four steps
// inside a row of 16 lanes are DPP row rotations: after the xor-16 step the values have period 16, so "rotate by 8" pairs lane i
// with lane i ^ 8 exactly; the values then have period 8 inside the row, so "rotate by 4" meets the partner "xor 4" would
// (a + b in the same order up to commutation), and so on down — the result is the xor butterfly's, bit for bit, at one
// VALU instruction per step instead of an LDS permute and its address arithmetic.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {  // == v += shfl_xor(v, 8), 4, 2, 1 — same pairs, same bits
  v += dpp_row<0x128>(v);
  v += dpp_row<0x124>(v);
  v += dpp_row<0x122>(v);
  v += dpp_row<0x121>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_row<0x128>(v));
  v = fmaxf(v, dpp_row<0x124>(v));
  v = fmaxf(v, dpp_row<0x122>(v));
  v = fmaxf(v, dpp_row<0x121>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  return row16_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 32));
  v = fmaxf(v, __shfl_xor(v, 16));
  return row16_max(v);
}
// x / b for many x and one b > 0 (a norm's denominator): r = RN(1 / b) once — the hardware reciprocal and one Newton step —,
// then per element q = x * r, the exact residual e = x - b * q (one fma) and q + e * r: the correctly rounded quotient
// (Markstein: a faithful q corrected with the correctly rounded reciprocal), i.e. the bits of the IEEE division the reference's
// `x / torch.sqrt(variance + eps)` (modeling_t5.py:167) produces, at 3 VALU instructions instead of the ~11 of a full
// division.  Finite operands in the normal range only (a row with an inf / nan is garbage in the reference too).
struct RowDivisor {
  float b, r;
  __device__ __forceinline__ explicit RowDivisor(float b_) : b(b_) {
    const float y = __builtin_amdgcn_rcpf(b_);
    r = fmaf(fmaf(-b_, y, 1.0f), y, y);
  }
  __device__ __forceinline__ float operator()(float x) const {
    const float q = x * r;
    return fmaf(fmaf(-b, q, x), r, q);
  }
};

}  // namespace gdr
