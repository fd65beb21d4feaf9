// Fused decode sub-blocks (decode_fused.hip): one workgroup owns (row panel, slice of the inner dimension) of a self-attention,
// cross-attention or feed-forward sub-block of a decoder block at decode time and writes a partial of its output rows as a slab.
#pragma once
#include "common.h"
#include "layers.h"

namespace gdr {

enum { FUSED_SA = 0, FUSED_CA = 1, FUSED_FFN = 2 };

struct FusedArgs {
  // phase 1: C1[m, c] = sum_k X[m, k] * W1[w1_row(c), k] (+ bias1[w1_row(c)])
  const float* X;       // [M, d] rows (lda = d)
  const float* W1;      // [*, d]
  const float* bias1;   // may be null
  int w1_seg_stride;    // output column c reads weight row (c / 64) * w1_seg_stride + slice * w1_slice_rows + c % 64   (SA)
  int w1_slice_rows;    //                            or  slice * w1_slice_rows + c                                      (CA, FFN: seg_stride = 0)
  // phase 3: P[m, n] = sum_k mid[m, k] * W3[n, slice * K3 + k]
  const float* W3;      // [d, ld3]
  int64_t ld3;
  float* slabs;         // [S][M][d]
  int64_t M;            // rows of this launch (<= 1 024)
  int d;                // model width: K of phase 1, N of phase 3
  int n_slices, n_panels;
  const int32_t* live;  // may be null: *live == 0 -> exit at once (StreamK::live)
  // ---- self-attention (FUSED_SA): the slot of this step in the K/V cache, the ancestors' rows
  float* slot;              // [M, ld_kv]: this step's cache rows; k at +k_off, v at +v_off (+ 64 * head)
  const float* kbase;       // cache + k_off: row r of the cache at kbase + r * ld_kv
  const float* vbase;
  int64_t ld_kv;
  int k_off, v_off;
  const int32_t* kv_rows;   // [M, Lk] absolute cache row of key j (the last entry is the row's own new slot)
  int Lk;                   // s + 1 <= 16
  float scale;
  const float* rel_bias;    // [buckets, H] or null
  int H, q_pos0, num_buckets;
  BucketLut lut;
  // ---- cross-attention (FUSED_CA): the query's encoder K / V
  const float* ck;          // row (b * L + j) * ld_c + 64 * head
  const float* cv;
  int64_t ld_c;
  int L, R;                 // encoder keys per query, beam rows per query (row m belongs to query m / R)
  const int64_t* key_mask;  // [B, L] (1 = attend)
};

// panel height in 16-row tiles (1 or 2) for M rows, or 0 when the fused sub-blocks do not serve the shape
int decode_fused_rt(int64_t M, int d, int inner, int dk);
// scratch for the slabs of one sub-block at M rows (the finest slicing used)
size_t decode_fused_slab_bytes(int64_t M, int d, int d_ff, int H);
// mode: FUSED_*; rt from decode_fused_rt; n1: phase-1 columns per slice (SA 192, CA 64, FFN 128 or 256).  g.n_slices set by the
// caller; n_panels is derived.  Returns 1 when the combination is not built, 0 after the launch, < 0 on error.
int launch_decode_fused(int mode, FusedArgs g, int rt, int n1, hipStream_t stream);
// C = sum_s slabs[s] (+ bias) (+ residual), ne.Y = norm(C): one launch.  Returns 1 when N is not served.
int launch_slab_reduce_norm(const float* slabs, int S, int64_t M, int N, float* C, int64_t ldc, const float* bias, const float* residual,
                            int64_t ldr, const NormEpilogue& ne, const int32_t* live, hipStream_t stream);

}  // namespace gdr
