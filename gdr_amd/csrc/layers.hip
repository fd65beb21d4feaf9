// Non-GEMM layer kernels for gfx950: embedding gather, T5 RMS norm, LayerNorm, and a fused attention
// (QKᵀ + relative-position bias + pad/causal mask + fp32 softmax + PV in one pass, K/V staged in LDS).
//
// All of them are HBM/LDS-bound at GDR's sequence lengths (L <= 128, 40 typical): 16-byte coalesced
// global accesses, wave64 shuffle reductions, no score matrix in HBM.
#include <math.h>

#include <stdlib.h>

#include "layers.h"

namespace gdr {

BucketLut make_bucket_lut(int nb, int max_distance) {
  // reference: transformers/modeling_t5.py:272-287 (float32 log, truncation, clamp); checked bit-exact
  // against the reference for all distances < 128 in tests/test_host_logic.py.
  BucketLut l;
  const int max_exact = nb / 2;
  for (int n = 0; n < 128; ++n) {
    int b;
    if (n < max_exact) {
      b = n;
    } else {
      const float v = logf((float)n / (float)max_exact) / (float)log((double)max_distance / max_exact) *
                      (float)(nb - max_exact);
      b = max_exact + (int)v;
      if (b > nb - 1) b = nb - 1;
    }
    l.v[n] = (uint8_t)b;
  }
  return l;
}

// ------------------------------------------------------------------------------------------ embed
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                    int64_t rows, int d4, int vocab, float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* src = reinterpret_cast<const float4*>(table) + id * d4;
  float4* dst = reinterpret_cast<float4*>(out) + row * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) dst[c] = src[c];
}

int launch_embed(const float* table, const int64_t* ids, int64_t rows, int d, int vocab, float* out,
                 hipStream_t stream) {
  GDR_CHECK_ARG(d % 4 == 0, "embed: d %% 4 != 0");
  if (rows == 0) return GDR_OK;
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, table, ids, rows, d / 4,
                     vocab, out);
  GDR_CHECK_LAUNCH("embed_kernel");
  return GDR_OK;
}

// ------------------------------------------------------------------------------------------ ragged batches
// Planning a ragged batch, two small kernels: (1) a wave per mask row: length + prefix test (B/4 workgroups — one workgroup
// walking 512 rows of dependent loads took 43 us); (2) one workgroup: exclusive scan of the lengths and the total.  The
// row map is written by (3), again a wave per sequence.
__global__ __launch_bounds__(256) void pack_len_kernel(const int64_t* __restrict__ mask, int B, int L,
                                                       int32_t* __restrict__ seq_len) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  int cnt = 0, last = -1;
  for (int j = lane; j < L; j += 64) {
    const bool on = mask[(int64_t)b * L + j] != 0;
    cnt += on ? 1 : 0;
    last = on ? j : last;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o);
    last = max(last, __shfl_xor(last, o));
  }
  if (lane == 0) seq_len[b] = (cnt > 0 && last + 1 == cnt) ? cnt : L;   // a non-empty prefix of ones, else keep all
}

__global__ __launch_bounds__(1024) void pack_scan_kernel(int B, const int32_t* __restrict__ seq_len,
                                                         int32_t* __restrict__ seq_off, int64_t* __restrict__ rows_total) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int b = base + tid;
    const int v = b < B ? seq_len[b] : 0;
    int inc = v;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (b < B) seq_off[b] = before + inc - v;
    __syncthreads();
    if (tid == 1023) carry = before + inc;
    __syncthreads();
  }
  if (tid == 0) {
    seq_off[B] = carry;
    *rows_total = carry;
  }
}

__global__ __launch_bounds__(256) void pack_rows_kernel(int B, int L, const int32_t* __restrict__ seq_len,
                                                        const int32_t* __restrict__ seq_off, int32_t* __restrict__ row_src) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const int o = seq_off[b], n = seq_len[b];
  for (int j = lane; j < n; j += 64) row_src[o + j] = b * L + j;
}

int launch_pack_plan(const int64_t* mask, int B, int L, int32_t* seq_len, int32_t* seq_off, int32_t* row_src,
                     int64_t* rows_total, hipStream_t stream) {
  GDR_CHECK_ARG(mask && seq_len && seq_off && row_src && rows_total, "pack_plan: null pointer");
  GDR_CHECK_ARG((int64_t)B * L < 0x7fffffffLL, "pack_plan: batch too large");
  const unsigned g = (unsigned)((B + 3) / 4);
  hipLaunchKernelGGL(pack_len_kernel, dim3(g), dim3(256), 0, stream, mask, B, L, seq_len);
  GDR_CHECK_LAUNCH("pack_len_kernel");
  hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, stream, B, seq_len, seq_off, rows_total);
  GDR_CHECK_LAUNCH("pack_scan_kernel");
  hipLaunchKernelGGL(pack_rows_kernel, dim3(g), dim3(256), 0, stream, B, L, seq_len, seq_off, row_src);
  GDR_CHECK_LAUNCH("pack_rows_kernel");
  return GDR_OK;
}

__global__ __launch_bounds__(256) void embed_packed_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                           const int32_t* __restrict__ row_src,
                                                           const int64_t* __restrict__ rows_dev, int d4, int vocab,
                                                           float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= *rows_dev) return;
  int64_t id = ids[row_src[row]];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float4* src = reinterpret_cast<const float4*>(table) + id * d4;
  float4* dst = reinterpret_cast<float4*>(out) + row * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) dst[c] = src[c];
}

int launch_embed_packed(const float* table, const int64_t* ids, const int32_t* row_src, const int64_t* rows_dev,
                        int64_t max_rows, int d, int vocab, float* out, hipStream_t stream) {
  GDR_CHECK_ARG(d % 4 == 0, "embed: d %% 4 != 0");
  if (max_rows == 0) return GDR_OK;
  hipLaunchKernelGGL(embed_packed_kernel, dim3((unsigned)((max_rows + 3) / 4)), dim3(256), 0, stream, table, ids, row_src,
                     rows_dev, d / 4, vocab, out);
  GDR_CHECK_LAUNCH("embed_packed_kernel");
  return GDR_OK;
}

// SCATTER = false: dst[i] = src[map[i]];  SCATTER = true: dst[map[i]] = src[i]   (rows of d4 float4), i < n or *n_dev
template <bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ map,
                                                        int64_t n, const int64_t* __restrict__ n_dev, int d4,
                                                        float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n_dev) n = *n_dev;
  if (i >= n) return;
  const int64_t m = map[i];
  const float4* s = reinterpret_cast<const float4*>(src) + (SCATTER ? i : m) * d4;
  float4* o = reinterpret_cast<float4*>(dst) + (SCATTER ? m : i) * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) o[c] = s[c];
}

int launch_gather_rows(const float* src, const int32_t* idx, int n, int d, float* dst, hipStream_t stream) {
  GDR_CHECK_ARG(d % 4 == 0, "gather_rows: d %% 4 != 0");
  if (n == 0) return GDR_OK;
  hipLaunchKernelGGL(move_rows_kernel<false>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, src, idx, (int64_t)n,
                     (const int64_t*)nullptr, d / 4, dst);
  GDR_CHECK_LAUNCH("gather_rows");
  return GDR_OK;
}

int launch_scatter_rows(const float* src, const int32_t* row_src, const int64_t* rows_dev, int64_t max_rows, int d,
                        float* dst, hipStream_t stream) {
  GDR_CHECK_ARG(d % 4 == 0, "scatter_rows: d %% 4 != 0");
  if (max_rows == 0) return GDR_OK;
  hipLaunchKernelGGL(move_rows_kernel<true>, dim3((unsigned)((max_rows + 3) / 4)), dim3(256), 0, stream, src, row_src,
                     max_rows, rows_dev, d / 4, dst);
  GDR_CHECK_LAUNCH("scatter_rows");
  return GDR_OK;
}

__global__ __launch_bounds__(256) void zero_dead_rows_kernel(float* __restrict__ x, const int32_t* __restrict__ seq_len,
                                                             int64_t rows, int L, int d4) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = (int)(row / L), j = (int)(row - (int64_t)b * L);
  if (j < seq_len[b]) return;
  float4* o = reinterpret_cast<float4*>(x) + row * d4;
  for (int c = threadIdx.x & 63; c < d4; c += 64) o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int launch_zero_dead_rows(float* x, const int32_t* seq_len, int B, int L, int d, hipStream_t stream) {
  GDR_CHECK_ARG(d % 4 == 0, "zero_dead_rows: d %% 4 != 0");
  const int64_t rows = (int64_t)B * L;
  if (rows == 0) return GDR_OK;
  hipLaunchKernelGGL(zero_dead_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, seq_len, rows, L, d / 4);
  GDR_CHECK_LAUNCH("zero_dead_rows_kernel");
  return GDR_OK;
}

// ------------------------------------------------------------------------------------------ norms
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
  union {
    __bf16 h[4];
    uint2 u;
  } o;
  o.h[0] = (__bf16)a, o.h[1] = (__bf16)b, o.h[2] = (__bf16)c, o.h[3] = (__bf16)d;  // v_cvt_pk_bf16_f32: RNE
  return o.u;
}

template <bool BF16OUT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ y, int64_t rows, int d4, float eps,
                                                      float* __restrict__ pooled, int pool_every,
                                                      const int64_t* __restrict__ rows_dev, uint2* __restrict__ y16 = nullptr,
                                                      int planes_ld = 0) {
  // planes_ld != 0 (the split-bf16 form, r06): y16 receives the row as three bf16 planes [hi | mid | lo] (row stride planes_ld
  // elements, plane stride d) instead of one rounded image, and y may be null (the fp32 copy is not written)
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_dev) rows = *rows_dev;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4* xr = reinterpret_cast<const float4*>(x) + row * d4;
  // the row stays in registers between the two passes (d <= 2048: 8 float4 per lane; wider rows are read twice as before): the
  // second read was an L1 / L2 round trip on a kernel that is a chain of round trips (23 us for 76 MB at 12 308 rows)
  constexpr int MAXC = 8;
  const bool in_regs = d4 <= 64 * MAXC;
  float4 xv[MAXC];
  float ss = 0.f;
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < d4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < d4) ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
  } else {
    for (int c = lane; c < d4; c += 64) {
      const float4 v = xr[c];
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  }
  ss = wave_sum(ss);
  const RowDivisor over(sqrtf(ss / (float)(d4 * 4) + eps));
  float4* yr = reinterpret_cast<float4*>(y) + row * d4;
  float4* pr = (pooled && row % pool_every == 0) ? reinterpret_cast<float4*>(pooled) + (row / pool_every) * d4
                                                  : nullptr;
  const float4* wr = reinterpret_cast<const float4*>(w);
  auto emit = [&](int c, const float4 v) {
    const float4 g = wr[c];
    float4 o;
    o.x = g.x * over(v.x), o.y = g.y * over(v.y), o.z = g.z * over(v.z), o.w = g.w * over(v.w);
    if (BF16OUT) {
      (reinterpret_cast<uint2*>(y) + row * d4)[c] = pack_bf16x4(o.x, o.y, o.z, o.w);
    } else {
      if (y) yr[c] = o;
      if (pr) pr[c] = o;
      if (y16 && planes_ld > 0) {  // bf16 x 3 planes [hi | mid | lo]
        __bf16* pl = reinterpret_cast<__bf16*>(y16) + row * planes_ld + 4 * c;
        const float v4[4] = {o.x, o.y, o.z, o.w};
        union {
          __bf16 h[4];
          uint2 u;
        } hi, mid, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hi.h[j] = (__bf16)v4[j];
          const float r1 = v4[j] - (float)hi.h[j];
          mid.h[j] = (__bf16)r1;
          lo.h[j] = (__bf16)(r1 - (float)mid.h[j]);
        }
        *reinterpret_cast<uint2*>(pl) = hi.u;
        *reinterpret_cast<uint2*>(pl + 4 * d4) = mid.u;
        *reinterpret_cast<uint2*>(pl + 8 * d4) = lo.u;
      } else if (y16 && planes_ld < 0) {  // fp16 x 2 planes [hi | (x - hi) * 2^11], row stride -planes_ld
        _Float16* pl = reinterpret_cast<_Float16*>(y16) + row * (int64_t)(-planes_ld) + 4 * c;
        const float v4[4] = {o.x, o.y, o.z, o.w};
        union {
          _Float16 h[4];
          uint2 u;
        } hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hi.h[j] = (_Float16)v4[j];
          lo.h[j] = (_Float16)((v4[j] - (float)hi.h[j]) * 2048.0f);
        }
        *reinterpret_cast<uint2*>(pl) = hi.u;
        *reinterpret_cast<uint2*>(pl + 4 * d4) = lo.u;
      } else if (y16) {
        (y16 + row * d4)[c] = pack_bf16x4(o.x, o.y, o.z, o.w);  // the same values, rounded: a bf16 linear's operand
      }
    }
  };
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < d4) emit(lane + 64 * i, xv[i]);
  } else {
    for (int c = lane; c < d4; c += 64) emit(c, xr[c]);
  }
}

int launch_rmsnorm(const float* x, const float* w, float* y, int64_t rows, int d, float eps, float* pooled,
                   int pool_every, hipStream_t stream, void* y16) {
  GDR_CHECK_ARG(d % 4 == 0, "rmsnorm: d %% 4 != 0");
  if (rows == 0) return GDR_OK;
  hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, w, y, rows, d / 4, eps,
                     pooled, pool_every > 0 ? pool_every : 1, (const int64_t*)nullptr, static_cast<uint2*>(y16));
  GDR_CHECK_LAUNCH("rmsnorm_kernel");
  return GDR_OK;
}

int launch_rmsnorm_dev(const float* x, const float* w, float* y, const int64_t* rows_dev, int64_t max_rows, int d, float eps,
                       hipStream_t stream, void* y16) {
  GDR_CHECK_ARG(d % 4 == 0, "rmsnorm: d %% 4 != 0");
  if (max_rows == 0) return GDR_OK;
  hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3((unsigned)((max_rows + 3) / 4)), dim3(256), 0, stream, x, w, y, max_rows,
                     d / 4, eps, (float*)nullptr, 1, rows_dev, static_cast<uint2*>(y16));
  GDR_CHECK_LAUNCH("rmsnorm_kernel(dev rows)");
  return GDR_OK;
}

// y (nullable) fp32 rows; planes: bf16 [rows, ld] rows = [hi | mid | lo] of the normed row (split-bf16 operand); rows_dev may be null
int launch_rmsnorm_planes(const float* x, const float* w, float* y, void* planes, int64_t ld, const int64_t* rows_dev, int64_t max_rows,
                          int d, float eps, hipStream_t stream, int f16x2) {
  GDR_CHECK_ARG(d % 4 == 0 && ld >= (f16x2 ? 2 : 3) * (int64_t)d && ld % 4 == 0, "rmsnorm(planes): d %% 4 != 0 or the plane row is too short");
  if (max_rows == 0) return GDR_OK;
  hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3((unsigned)((max_rows + 3) / 4)), dim3(256), 0, stream, x, w, y, max_rows, d / 4, eps,
                     (float*)nullptr, 1, rows_dev, static_cast<uint2*>(planes), f16x2 ? -(int)ld : (int)ld);
  GDR_CHECK_LAUNCH("rmsnorm_kernel(planes)");
  return GDR_OK;
}

int launch_rmsnorm_bf16(const float* x, const float* w, void* y_bf16, int64_t rows, int d, float eps, hipStream_t stream) {
  GDR_CHECK_ARG(d % 4 == 0, "rmsnorm: d %% 4 != 0");
  if (rows == 0) return GDR_OK;
  hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, w,
                     static_cast<float*>(y_bf16), rows, d / 4, eps, (float*)nullptr, 1, (const int64_t*)nullptr);
  GDR_CHECK_LAUNCH("rmsnorm_kernel<bf16>");
  return GDR_OK;
}

int launch_rmsnorm_bf16_dev(const float* x, const float* w, void* y_bf16, const int64_t* rows_dev, int64_t max_rows, int d,
                            float eps, hipStream_t stream) {
  GDR_CHECK_ARG(d % 4 == 0, "rmsnorm: d %% 4 != 0");
  if (max_rows == 0) return GDR_OK;
  hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3((unsigned)((max_rows + 3) / 4)), dim3(256), 0, stream, x, w,
                     static_cast<float*>(y_bf16), max_rows, d / 4, eps, (float*)nullptr, 1, rows_dev);
  GDR_CHECK_LAUNCH("rmsnorm_kernel<bf16>(dev rows)");
  return GDR_OK;
}

// y = x / max(||x||_2, eps) per row: torch.nn.functional.normalize(rep, dim=-1) of DensePooler (dense.py:24-25)
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows,
                                                           int d, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * d;
  float ss = 0.f;
  for (int c = lane; c < d; c += 64) ss += xr[c] * xr[c];
  const float nrm = fmaxf(sqrtf(wave_sum(ss)), eps);
  for (int c = lane; c < d; c += 64) y[row * d + c] = xr[c] / nrm;
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y,
                                                        int64_t rows, int d4, float eps,
                                                        const float* __restrict__ addv,
                                                        const int64_t* __restrict__ rows_dev, uint2* __restrict__ y16,
                                                        int f16x2_ld = 0) {
  // f16x2_ld != 0 (the fp16 x 2 split form, r06): y16 receives the row as plane rows [fp16 hi | fp16 (y - hi) * 2^11] of f16x2_ld elements
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_dev) rows = *rows_dev;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4* xr = reinterpret_cast<const float4*>(x) + row * d4;
  const float inv_d = 1.0f / (float)(d4 * 4);
  const float4* av = reinterpret_cast<const float4*>(addv);
  auto ld = [&](int c) {
    float4 v = xr[c];
    if (av) {
      const float4 t = av[c];
      v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
    }
    return v;
  };
  float s = 0.f;
  for (int c = lane; c < d4; c += 64) {
    const float4 v = ld(c);
    s += v.x + v.y + v.z + v.w;
  }
  const float mean = wave_sum(s) * inv_d;
  float ss = 0.f;
  for (int c = lane; c < d4; c += 64) {
    const float4 v = ld(c);
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) * inv_d + eps);
  float4* yr = reinterpret_cast<float4*>(y) + row * d4;
  const float4* wr = reinterpret_cast<const float4*>(w);
  const float4* br = reinterpret_cast<const float4*>(b);
  for (int c = lane; c < d4; c += 64) {
    const float4 v = ld(c), g = wr[c], bb = br[c];
    float4 o;
    o.x = (v.x - mean) * rstd * g.x + bb.x, o.y = (v.y - mean) * rstd * g.y + bb.y;
    o.z = (v.z - mean) * rstd * g.z + bb.z, o.w = (v.w - mean) * rstd * g.w + bb.w;
    yr[c] = o;
    if (y16 && f16x2_ld) {
      _Float16* pl = reinterpret_cast<_Float16*>(y16) + row * (int64_t)f16x2_ld + 4 * c;
      const float v4[4] = {o.x, o.y, o.z, o.w};
      union {
        _Float16 h[4];
        uint2 u;
      } hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hi.h[j] = (_Float16)v4[j];
        lo.h[j] = (_Float16)((v4[j] - (float)hi.h[j]) * 2048.0f);
      }
      *reinterpret_cast<uint2*>(pl) = hi.u;
      *reinterpret_cast<uint2*>(pl + 4 * d4) = lo.u;
    } else if (y16) {
      (y16 + row * d4)[c] = pack_bf16x4(o.x, o.y, o.z, o.w);
    }
  }
}

// y = LN(LN(x; w1, b1) + addv; w2, b2) — the adaptor's norm1 -> (+ the constant single-key cross-attention output) -> norm2
// (nn.TransformerDecoderLayer, post-LN; modeling_t5.py:1241-1244) in ONE pass: the row stays in registers between the two
// norms instead of going through memory and a second launch.  Same expressions and reduction order as two layernorm_kernel
// launches: bit-identical.  d <= 2048.
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, const float* __restrict__ addv,
                                                         float* __restrict__ y, int64_t rows, int d4, float eps,
                                                         const int64_t* __restrict__ rows_dev, uint2* __restrict__ y16) {
  constexpr int MAXC = 8;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_dev) rows = *rows_dev;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float inv_d = 1.0f / (float)(d4 * 4);
  const float4* xr = reinterpret_cast<const float4*>(x) + row * d4;
  float4 v[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < d4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float4* wr[2] = {reinterpret_cast<const float4*>(w1), reinterpret_cast<const float4*>(w2)};
  const float4* br[2] = {reinterpret_cast<const float4*>(b1), reinterpret_cast<const float4*>(b2)};
  const float4* av = reinterpret_cast<const float4*>(addv);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && av) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < d4) {
          const float4 t = av[c];
          v[i].x += t.x, v[i].y += t.y, v[i].z += t.z, v[i].w += t.w;
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < d4) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) * inv_d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < d4) {
        const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
      }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < d4) {
        const float4 g = wr[pass][c], bb = br[pass][c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + bb.x, o.y = (v[i].y - mean) * rstd * g.y + bb.y;
        o.z = (v[i].z - mean) * rstd * g.z + bb.z, o.w = (v[i].w - mean) * rstd * g.w + bb.w;
        v[i] = o;
      }
    }
  }
  float4* yr = reinterpret_cast<float4*>(y) + row * d4;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < d4) {
      yr[c] = v[i];
      if (y16) (y16 + row * d4)[c] = pack_bf16x4(v[i].x, v[i].y, v[i].z, v[i].w);
    }
  }
}

// returns 1 when d is too wide for the one-pass form (the caller runs two layernorm launches)
int launch_layernorm2(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* addv, float* y,
                      const int64_t* rows_dev, int64_t max_rows, int d, float eps, hipStream_t stream, void* y16) {
  GDR_CHECK_ARG(d % 4 == 0, "layernorm: d %% 4 != 0");
  if (d > 2048) return 1;
  if (max_rows == 0) return GDR_OK;
  hipLaunchKernelGGL(layernorm2_kernel, dim3((unsigned)((max_rows + 3) / 4)), dim3(256), 0, stream, x, w1, b1, w2, b2, addv, y,
                     max_rows, d / 4, eps, rows_dev, static_cast<uint2*>(y16));
  GDR_CHECK_LAUNCH("layernorm2_kernel");
  return GDR_OK;
}

int launch_layernorm(const float* x, const float* w, const float* b, float* y, int64_t rows, int d, float eps,
                     const float* addv, hipStream_t stream, void* y16, int f16x2_ld) {
  GDR_CHECK_ARG(d % 4 == 0 && (f16x2_ld == 0 || (y16 && f16x2_ld >= 2 * d && f16x2_ld % 4 == 0)), "layernorm: d %% 4 != 0 or a bad plane row");
  if (rows == 0) return GDR_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, w, b, y, rows, d / 4,
                     eps, addv, (const int64_t*)nullptr, static_cast<uint2*>(y16), f16x2_ld);
  GDR_CHECK_LAUNCH("layernorm_kernel");
  return GDR_OK;
}

int launch_layernorm_dev(const float* x, const float* w, const float* b, float* y, const int64_t* rows_dev, int64_t max_rows,
                         int d, float eps, const float* addv, hipStream_t stream, void* y16, int f16x2_ld) {
  GDR_CHECK_ARG(d % 4 == 0 && (f16x2_ld == 0 || (y16 && f16x2_ld >= 2 * d && f16x2_ld % 4 == 0)), "layernorm: d %% 4 != 0 or a bad plane row");
  if (max_rows == 0) return GDR_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((max_rows + 3) / 4)), dim3(256), 0, stream, x, w, b, y, max_rows, d / 4,
                     eps, addv, rows_dev, static_cast<uint2*>(y16), f16x2_ld);
  GDR_CHECK_LAUNCH("layernorm_kernel(dev rows)");
  return GDR_OK;
}

// ------------------------------------------------------------------------------------------ attention
// One workgroup per (batch, head): K and V rows of that head are staged once in LDS (row stride dk+4 floats:
// ≡ 4 mod 64 banks for dk = 64, so the per-lane ds_read_b128 of "my key's row" is conflict-free); each of
// the 4 waves then walks query rows: lane j scores keys j and j+64, softmax by wave shuffles, PV with lane = d.
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (a.live && *a.live == 0) return;  // uniform: every query of the generate call is done
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int dk = a.dk, dks = dk + 4, c4 = dk >> 2;
  const int Lk = a.Lk, Lkp = (Lk + 3) & ~3;
  float* Ks = smem;
  float* Vs = Ks + Lk * dks;
  // query rows are dealt to gridDim.y workgroups per (batch, head): with few (batch, head) pairs and many rows (one
  // query decoded with 100 beams) a single workgroup per pair left the chip idle
  const int rows_per = (a.Lq + (int)gridDim.y - 1) / (int)gridDim.y;
  const int i0 = (int)blockIdx.y * rows_per, i1 = min(a.Lq, i0 + rows_per);
  const int iters = i1 > i0 ? (i1 - i0 + 3) >> 2 : 0;
  // up to 32 query rows per workgroup are staged together with K / V (one global round trip for the whole workgroup);
  // beyond that a wave loads its row when it gets to it
  const bool pre = rows_per <= 32;
  const int qrows = pre ? ((rows_per + 3) & ~3) : 4;
  float* Qs = Vs + Lk * dks;      // [qrows][dk]
  float* Ps = Qs + qrows * dk;    // [4][Lkp]
  float* Bs = Ps + 4 * Lkp;       // [num_buckets]
  const int tid = threadIdx.x;
  const int kb = b / a.kv_group;
  for (int e = tid; e < Lk * c4; e += 256) {
    const int j = e / c4, c = e - j * c4;
    const int64_t rk = a.kv_rows ? (int64_t)a.kv_rows[(int64_t)b * Lk + j] : (int64_t)kb * a.k_bstride + j;
    *reinterpret_cast<float4*>(Ks + j * dks + 4 * c) =
        *reinterpret_cast<const float4*>(a.k + rk * a.ldk + h * dk + 4 * c);
    *reinterpret_cast<float4*>(Vs + j * dks + 4 * c) =
        *reinterpret_cast<const float4*>(a.v + rk * a.ldv + h * dk + 4 * c);
  }
  if (pre) {
    for (int e = tid; e < (i1 - i0) * c4; e += 256) {
      const int r = e / c4, c = e - r * c4;
      float4 q;
      if (a.q_part) {  // the projection's split-K slabs, summed here in the reduction kernel's order
        const int64_t m = (int64_t)b * a.q_bstride + i0 + r;
        const int n = h * dk + 4 * c;
        const float* p = a.q_part + ((m >> 6) * a.q_tiles_n + (n >> 6)) * (int64_t)a.q_S * 4096 + (m & 63) * 64 + (n & 63);
        float4 t[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s < a.q_S) t[s] = *reinterpret_cast<const float4*>(p + (int64_t)s * 4096);
        q = t[0];
#pragma unroll
        for (int s = 1; s < 8; ++s)
          if (s < a.q_S) q.x += t[s].x, q.y += t[s].y, q.z += t[s].z, q.w += t[s].w;
        for (int s = 8; s < a.q_S; ++s) {
          const float4 u = *reinterpret_cast<const float4*>(p + (int64_t)s * 4096);
          q.x += u.x, q.y += u.y, q.z += u.z, q.w += u.w;
        }
      } else {
        q = *reinterpret_cast<const float4*>(a.q + ((int64_t)b * a.q_bstride + i0 + r) * a.ldq + h * dk + 4 * c);
      }
      q.x *= a.scale, q.y *= a.scale, q.z *= a.scale, q.w *= a.scale;
      *reinterpret_cast<float4*>(Qs + r * dk + 4 * c) = q;
    }
  }
  if (a.rel_bias && tid < a.num_buckets) Bs[tid] = a.rel_bias[tid * a.H + h];
  const int wave = tid >> 6, lane = tid & 63;
  float* ps = Ps + wave * Lkp;
  __syncthreads();  // K / V, the bias table and the staged query rows are visible; from here on a wave only touches its own
                    // strips (q row, probabilities), whose LDS operations complete in order: no workgroup barrier below
  const int half = a.num_buckets >> 1;
  const float masked = a.causal_neg_inf ? -INFINITY : -1e9f;
  for (int it = 0; it < iters; ++it) {
    const int i = i0 + it * 4 + wave;
    const bool active = i < i1;
    float* qs = Qs + (pre ? it * 4 + wave : wave) * dk;
    if (active && !pre) {
      const float* qr = a.q + ((int64_t)b * a.q_bstride + i) * a.ldq + h * dk;
      for (int d = lane; d < dk; d += 64) qs[d] = qr[d] * a.scale;
    }
    __builtin_amdgcn_wave_barrier();
    float sc[2];
    float mx = -INFINITY;
    if (active) {
      const int i_abs = a.q_pos0 + (a.q_same_pos ? 0 : i);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        float s = -INFINITY;
        if (j < Lk) {
          float acc = 0.f;
          const float* kr = Ks + j * dks;
          for (int c = 0; c < c4; ++c) {
            const float4 qq = *reinterpret_cast<const float4*>(qs + 4 * c);
            const float4 kk = *reinterpret_cast<const float4*>(kr + 4 * c);
            acc = fmaf(qq.x, kk.x, acc);
            acc = fmaf(qq.y, kk.y, acc);
            acc = fmaf(qq.z, kk.z, acc);
            acc = fmaf(qq.w, kk.w, acc);
          }
          float add = 0.f;
          if (a.rel_bias) {
            int n = i_abs - j;  // = -(memory_position - context_position)
            int bucket = 0;
            if (a.bidirectional) {
              if (n < 0) {
                bucket = half;
                n = -n;
              }
            } else if (n < 0) {
              n = 0;
            }
            bucket += a.lut.v[n < 127 ? n : 127];
            add = Bs[bucket];
          }
          bool allowed = true;
          if (a.causal) allowed = j <= i_abs;
          if (a.key_mask) allowed = allowed && (a.key_mask[(int64_t)kb * a.mask_bstride + j] != 0);
          if (!allowed) add += masked;
          s = acc + add;
        }
        sc[t] = s;
        mx = fmaxf(mx, s);
      }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        if (j < Lk) {
          const float p = expf(sc[t] - mx);
          ps[j] = p;
          sum += p;
        }
      }
      sum = wave_sum(sum);
      sc[0] = sum;
    }
    __builtin_amdgcn_wave_barrier();  // probabilities written before any lane reads them back
    if (active) {
      const float inv = 1.0f / sc[0];
      float* orow = a.out + ((int64_t)b * a.o_bstride + i) * a.ldo + h * dk;
      for (int d = lane; d < dk; d += 64) {
        float o = 0.f;
        int j = 0;
        for (; j + 4 <= Lk; j += 4) {
          const float4 pp = *reinterpret_cast<const float4*>(ps + j);
          o = fmaf(pp.x * inv, Vs[(j + 0) * dks + d], o);
          o = fmaf(pp.y * inv, Vs[(j + 1) * dks + d], o);
          o = fmaf(pp.z * inv, Vs[(j + 2) * dks + d], o);
          o = fmaf(pp.w * inv, Vs[(j + 3) * dks + d], o);
        }
        for (; j < Lk; ++j) o = fmaf(ps[j] * inv, Vs[j * dks + d], o);
        if (a.out_bf16)
          static_cast<__bf16*>(a.out_bf16)[((int64_t)b * a.o_bstride + i) * a.ldo + h * dk + d] = (__bf16)o;
        else
          orow[d] = o;
      }
    }
    __builtin_amdgcn_wave_barrier();  // before the next row overwrites qs / ps
  }
}


// ------------------------------------------------------------------------------------------ attention on MFMA
// Full self-attention (Lq == Lk = L <= 128, dk = 64) on v_mfma_f32_16x16x4_f32: one workgroup per (batch, head), one wave
// per 16-query tile against 16-key tiles (a T5 query batch of L = 40 computes 48 x 48 scores; the 32x32x2 form this
// replaced in round 2 computed 64 x 64 — it is gone, rocprof history: 104 -> 66 us per layer at B = 512).  K, V of the
// head are staged once in LDS (row stride 68 floats: conflict-free ds_read_b128).  S^T = K·Q^T so that a lane
// holds scores of ONE query (4 lanes share a query: lane>>4 picks 4 of every 16 keys; max / sum need two shuffles), and
// the probability registers are the B operand of the PV MFMAs as they stand (MFMA step s contracts keys 16t + 4q + s,
// which is register s of lane-quarter q; the A operand reads V[that key][d] from LDS).  Arithmetic is the reference's
// (scores + bias + mask, fp32 softmax, PV; modeling_t5.py:384-413), fp32 throughout.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int NT>  // 16-key tiles = 16-query tiles = waves per workgroup: ceil(L / 16), 1..8
__global__ __launch_bounds__(64 * NT) void attention_mfma16_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DK = 64, DS = DK + 4, LP = 16 * NT, NTHR = 64 * NT;
  const int Lr = a.Lk;
  float* Ks = smem;  // q never passes through LDS: a wave reads the fragments of its own 16 query rows straight from global
  float* Vs = Ks + Lr * DS;
  float* RelB = Vs + Lr * DS;   // [2*LP]
  float* Mk = RelB + 2 * LP;    // [LP]
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int L = a.seq_len ? a.seq_len[b] : a.Lk, tid = threadIdx.x;  // ragged: this sequence's own length
  const int64_t qrow0 = a.seq_off ? a.seq_off[b] : (int64_t)b * a.q_bstride;
  const int64_t krow0 = a.seq_off ? a.seq_off[b] : (int64_t)b * a.k_bstride;
  const int64_t orow0 = a.seq_off ? a.seq_off[b] : (int64_t)b * a.o_bstride;
  // this wave's query fragments (B operand of S^T = K·Q^T): lane (c16, q4) holds Q[16w + c16][16jj + 4q4 .. +3], issued
  // first so that the loads are in flight while K / V are staged
  float4 qv[DK / 16];
  {
    const int wq = tid >> 6, lq = tid & 63;
    const int64_t qr = qrow0 + min(16 * wq + (lq & 15), L - 1);
    if (16 * wq < L) {
      if (a.qkv_bf16) {
        const __bf16* q16 = reinterpret_cast<const __bf16*>(a.q) + qr * a.ldq + h * DK + 4 * (lq >> 4);
#pragma unroll
        for (int jj = 0; jj < DK / 16; ++jj) {
          const uint2 u = *reinterpret_cast<const uint2*>(q16 + 16 * jj);
          qv[jj] = make_float4(__uint_as_float(u.x << 16) * a.scale, __uint_as_float(u.x & 0xffff0000u) * a.scale,
                               __uint_as_float(u.y << 16) * a.scale, __uint_as_float(u.y & 0xffff0000u) * a.scale);
        }
      } else {
        const float* q32 = a.q + qr * a.ldq + h * DK + 4 * (lq >> 4);
#pragma unroll
        for (int jj = 0; jj < DK / 16; ++jj) qv[jj] = *reinterpret_cast<const float4*>(q32 + 16 * jj);  // scaled behind the barrier
      }
    }
  }
  // Everything this workgroup reads is requested before anything is consumed.  Written as loops of "load, convert, store to LDS"
  // the prologue compiled into a chain of eight dependent round trips (q waited for at once for its scale, bucket table ->
  // bias, mask, then the K / V staging loop one iteration at a time) in front of ~1 us of arithmetic: 44 us per launch for
  // 6 144 (sequence, head) pairs at 12 308 rows.  Same values into the same LDS cells — results are bit-identical.
  int rb_bucket = -1;  // this thread's cell of the relative-position bias table (2*LP <= NTHR * 2: two cells at most)
  int rb_lutoff[2];
  unsigned char rb_lut[2] = {0, 0};
  int rb_base[2] = {0, 0};
  constexpr int RBI = (2 * LP + NTHR - 1) / NTHR;  // 1 (2*LP = 32*NT <= 64*NT)
  static_assert(RBI == 1, "one bias cell per thread");
  if (a.rel_bias && tid < 2 * LP) {
    int n = tid - (LP - 1), bucket = 0;
    if (a.bidirectional) {
      if (n < 0) {
        bucket = a.num_buckets >> 1;
        n = -n;
      }
    } else if (n < 0) {
      n = 0;
    }
    rb_base[0] = bucket;
    rb_lutoff[0] = n < 127 ? n : 127;
    rb_lut[0] = a.lut.v[rb_lutoff[0]];  // (a load from the kernel arguments: the first of the two dependent ones)
    rb_bucket = 0;
  }
  int64_t mk_raw = 1;
  const bool mk_mine = tid < LP && tid < L && a.key_mask;
  if (mk_mine) mk_raw = a.key_mask[(int64_t)b * a.mask_bstride + tid];
  if (a.qkv_bf16) {
    // bf16 q / k / v (the qkv linear of the bf16 precision mode emits them so): 16-byte loads of 8 elements, widened to
    // fp32 in the LDS image — the MFMAs below are unchanged (products of bf16 values are exact in fp32)
    const __bf16* k16 = reinterpret_cast<const __bf16*>(a.k);
    const __bf16* v16 = reinterpret_cast<const __bf16*>(a.v);
    auto widen = [](uint4 u, float4& lo, float4& hi) {
      lo = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
      hi = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                       __uint_as_float(u.w & 0xffff0000u));
    };
    for (int e = tid; e < L * (DK / 8); e += NTHR) {
      const int r = e >> 3, c = e & 7;
      float4 lo, hi;
      widen(*reinterpret_cast<const uint4*>(k16 + (krow0 + r) * a.ldk + h * DK + 8 * c), lo, hi);
      *reinterpret_cast<float4*>(Ks + r * DS + 8 * c) = lo;
      *reinterpret_cast<float4*>(Ks + r * DS + 8 * c + 4) = hi;
      widen(*reinterpret_cast<const uint4*>(v16 + (krow0 + r) * a.ldv + h * DK + 8 * c), lo, hi);
      *reinterpret_cast<float4*>(Vs + r * DS + 8 * c) = lo;
      *reinterpret_cast<float4*>(Vs + r * DS + 8 * c + 4) = hi;
    }
  } else {
    constexpr int KI = (LP * (DK / 4) + NTHR - 1) / NTHR;  // 4 for every NT
    float4 kst[KI], vst[KI];
#pragma unroll
    for (int it = 0; it < KI; ++it) {  // unconditional loads (the address is clamped to the sequence's last element): the staging
      const int e = min(tid + it * NTHR, L * (DK / 4) - 1);  // registers stay registers and all 2 * KI requests go out back to back
      const int r = e >> 4, c = e & 15;
      kst[it] = *reinterpret_cast<const float4*>(a.k + (krow0 + r) * a.ldk + h * DK + 4 * c);
      vst[it] = *reinterpret_cast<const float4*>(a.v + (krow0 + r) * a.ldv + h * DK + 4 * c);
    }
    // nothing above may sink below this line, nothing below may rise above it: the compiler otherwise moves every load next to
    // its LDS store again (one round trip per staging iteration)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // the bias cell: the second of its two dependent loads goes out while K / V are still arriving (the bucket byte was
    // requested before them, so waiting for it leaves them in flight)
    float rb_val = 0.f;
    if (rb_bucket >= 0) rb_val = a.rel_bias[(rb_base[0] + rb_lut[0]) * a.H + h];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      asm volatile("" : "+v"(kst[it].x), "+v"(kst[it].y), "+v"(kst[it].z), "+v"(kst[it].w));
      asm volatile("" : "+v"(vst[it].x), "+v"(vst[it].y), "+v"(vst[it].z), "+v"(vst[it].w));
    }
    asm volatile("" : "+v"(mk_raw));
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      const int e = tid + it * NTHR;
      if (e < L * (DK / 4)) {
        const int r = e >> 4, c = e & 15;
        *reinterpret_cast<float4*>(Ks + r * DS + 4 * c) = kst[it];
        *reinterpret_cast<float4*>(Vs + r * DS + 4 * c) = vst[it];
      }
    }
    if (tid < 2 * LP) RelB[tid] = rb_val;
  }
  if (a.qkv_bf16 && tid < 2 * LP) RelB[tid] = rb_bucket >= 0 ? a.rel_bias[(rb_base[0] + rb_lut[0]) * a.H + h] : 0.f;
  if (tid < LP) Mk[tid] = tid < L ? ((mk_mine && mk_raw == 0) ? (a.causal_neg_inf ? -INFINITY : -1e9f) : 0.f) : -INFINITY;
  __syncthreads();
  if (!a.qkv_bf16) {
#pragma unroll
    for (int jj = 0; jj < DK / 16; ++jj) qv[jj].x *= a.scale, qv[jj].y *= a.scale, qv[jj].z *= a.scale, qv[jj].w *= a.scale;
  }

  const int w = tid >> 6, lane = tid & 63, c16 = lane & 15, q4 = lane >> 4;
  if (16 * w >= L) return;  // ragged: a query tile past this sequence's end (no barrier follows)
  // key tiles past this sequence's end are skipped (wave-uniform): their keys carry -inf, i.e. probability exactly 0 and a
  // contribution of exactly +0 to every sum below — the result is bit-identical to computing them
  const int nt_live = (L + 15) >> 4;
  f32x4_t st[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) st[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // ---- S^T tiles: A = K rows (keys), B = Q rows (queries); d index permuted inside chunks of 16 (element s of the
  //      float4 read at d = 16*jj + 4*q4 is the operand of MFMA step s) identically on both operands
#pragma unroll
  for (int jj = 0; jj < DK / 16; ++jj) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= nt_live) continue;
      const float4 kv = *reinterpret_cast<const float4*>(Ks + min(16 * t + c16, L - 1) * DS + 4 * q4 + 16 * jj);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qv[jj].x, st[t], 0, 0, 0);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qv[jj].y, st[t], 0, 0, 0);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qv[jj].z, st[t], 0, 0, 0);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qv[jj].w, st[t], 0, 0, 0);
    }
  }
  // ---- bias + mask + softmax over keys: this lane holds keys 16t + 4*q4 + r of query i
  const int i = 16 * w + c16;
  const float masked = a.causal_neg_inf ? -INFINITY : -1e9f;
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= nt_live) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * t + 4 * q4 + r;
      float add = RelB[i - j + (LP - 1)] + Mk[j];          // (bias + mask) first, as the reference (modeling_t5.py:399-400)
      if (a.causal && j > i && Mk[j] == 0.f) add += masked;  // one -1e9 per masked key, never two
      const float s = st[t][r] + add;
      st[t][r] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= nt_live) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __expf(st[t][r] - mx);  // padded keys: exp(-inf) = 0
      st[t][r] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  // ---- O^T = V^T · P^T, four 16-wide d tiles: lane ends with d = 16*dt + 4*q4 + 0..3 of its query
#pragma unroll
  for (int dt = 0; dt < DK / 16; ++dt) {
    f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= nt_live) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * t + 4 * q4 + r;
        const float vv = Vs[min(j, L - 1) * DS + 16 * dt + c16];
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, st[t][r], o, 0, 0, 0);
      }
    }
    if (i < L) {
      const int64_t off = (orow0 + i) * a.ldo + h * DK + 16 * dt + 4 * q4;
      const float4 ov = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      if (a.out_bf16)
        *reinterpret_cast<uint2*>(static_cast<__bf16*>(a.out_bf16) + off) = pack_bf16x4(ov.x, ov.y, ov.z, ov.w);
      else
        *reinterpret_cast<float4*>(a.out + off) = ov;
    }
  }
}

// The same self-attention when q, k, v arrive as bf16 (the qkv linear of the bf16 precision mode emits them so, BASELINE config
// C5) on v_mfma_f32_16x16x32_bf16 — 16x the fp32 MFMA rate: the kernel above spends about half of its time in 96 fp32 MFMAs per
// wave (3 key tiles), this one needs 6 + 24.
//   S^T = K.Q^T : both operands are bf16 as they stand (products exact in fp32, fp32 accumulate): K rows from a padded bf16 LDS
//                 image (144-byte rows: conflict-free 16-byte fragment reads), Q fragments straight from global memory.
//   O^T = V^T.P^T: the probabilities are fp32 registers; each is split EXACTLY into three bf16 pieces p = hi + mid + lo
//                 (truncations of the successive remainders: 3 x 8 mantissa bits cover fp32's 24), three MFMAs per 32-key block
//                 accumulate hi.V + mid.V + lo.V — the fp32 product p.v summed in fp32, no new rounding point.  The k-slots of a
//                 block are permuted so that the S^T registers are the B operand as they stand: slot e of lane-quarter q4 is key
//                 16*(2u + e/4) + 4*q4 + e%4; the A operand reads V transposed (Vt[d][key], bf16, 272-byte rows) to match.
// Same bias / mask / softmax arithmetic as above; T5's scale of 1.0 only (launch_attention checks).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_trunc_bits(float x) { return __float_as_uint(x) & 0xffff0000u; }

template <int NT>
__global__ __launch_bounds__(64 * NT) void attention_mfma_bf16_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  constexpr int DK = 64, KS = 144, VS = 272, LP = 16 * NT, NTHR = 64 * NT, NB = (NT + 1) / 2, VKEYS = 32 * NB;
  const int Lr = a.Lk;
  unsigned char* Ks = smem_b;                                   // [Lr][144 B]  K rows, bf16
  unsigned char* Vt = Ks + (size_t)((Lr * KS + 15) & ~15);      // [64][272 B]  V transposed: Vt[d][key], bf16, keys < VKEYS
  float* RelB = reinterpret_cast<float*>(Vt + DK * VS);         // [2*LP]
  float* Mk = RelB + 2 * LP;                                    // [LP]
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int L = a.seq_len ? a.seq_len[b] : a.Lk, tid = threadIdx.x;
  const int64_t qrow0 = a.seq_off ? a.seq_off[b] : (int64_t)b * a.q_bstride;
  const int64_t krow0 = a.seq_off ? a.seq_off[b] : (int64_t)b * a.k_bstride;
  const int64_t orow0 = a.seq_off ? a.seq_off[b] : (int64_t)b * a.o_bstride;
  const __bf16* q16 = reinterpret_cast<const __bf16*>(a.q);
  const __bf16* k16 = reinterpret_cast<const __bf16*>(a.k);
  const __bf16* v16 = reinterpret_cast<const __bf16*>(a.v);
  // this wave's query fragments (B operand of S^T): lane (c16, q4) holds Q[16w + c16][32kk + 8q4 .. +7]
  uint4 qv[2];
  {
    const int wq = tid >> 6, lq = tid & 63;
    if (16 * wq < L) {
      const int64_t qr = qrow0 + min(16 * wq + (lq & 15), L - 1);
      const __bf16* qp = q16 + qr * a.ldq + h * DK + 8 * (lq >> 4);
      qv[0] = *reinterpret_cast<const uint4*>(qp);
      qv[1] = *reinterpret_cast<const uint4*>(qp + 32);
    }
  }
  for (int e = tid; e < 2 * LP; e += NTHR) {
    float v = 0.f;
    if (a.rel_bias) {
      int n = e - (LP - 1), bucket = 0;
      if (a.bidirectional) {
        if (n < 0) {
          bucket = a.num_buckets >> 1;
          n = -n;
        }
      } else if (n < 0) {
        n = 0;
      }
      bucket += a.lut.v[n < 127 ? n : 127];
      v = a.rel_bias[bucket * a.H + h];
    }
    RelB[e] = v;
  }
  for (int j = tid; j < LP; j += NTHR) {
    float v = -INFINITY;
    if (j < L) v = (a.key_mask && a.key_mask[(int64_t)b * a.mask_bstride + j] == 0) ? (a.causal_neg_inf ? -INFINITY : -1e9f) : 0.f;
    Mk[j] = v;
  }
  // V^T keys past this sequence's end must be finite: their probability is exactly 0, and 0 x garbage must stay 0
  for (int e = tid; e < DK * (VKEYS / 2); e += NTHR) {
    const int d = e / (VKEYS / 2), kp = e - d * (VKEYS / 2);
    if (2 * kp + 1 >= L) reinterpret_cast<uint32_t*>(Vt + d * VS)[kp] = 0u;   // pairs (2kp, 2kp+1); a live key of a mixed pair is rewritten below
  }
  __syncthreads();
  for (int e = tid; e < L * (DK / 8); e += NTHR) {
    const int r = e >> 3, c = e & 7;
    *reinterpret_cast<uint4*>(Ks + r * KS + 16 * c) = *reinterpret_cast<const uint4*>(k16 + (krow0 + r) * a.ldk + h * DK + 8 * c);
    const uint4 u = *reinterpret_cast<const uint4*>(v16 + (krow0 + r) * a.ldv + h * DK + 8 * c);
    const uint32_t wv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      reinterpret_cast<uint16_t*>(Vt + (8 * c + 2 * i) * VS)[r] = (uint16_t)(wv[i] & 0xffffu);
      reinterpret_cast<uint16_t*>(Vt + (8 * c + 2 * i + 1) * VS)[r] = (uint16_t)(wv[i] >> 16);
    }
  }
  __syncthreads();

  const int w = tid >> 6, lane = tid & 63, c16 = lane & 15, q4 = lane >> 4;
  if (16 * w >= L) return;
  const int nt_live = (L + 15) >> 4;
  f32x4_t st[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) st[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= nt_live) continue;
    const unsigned char* kr = Ks + min(16 * t + c16, L - 1) * KS + 16 * q4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint4 kv = *reinterpret_cast<const uint4*>(kr + 64 * kk);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kv), __builtin_bit_cast(bf16x8_t, qv[kk]), st[t], 0, 0, 0);
    }
  }
  const int i = 16 * w + c16;
  const float masked = a.causal_neg_inf ? -INFINITY : -1e9f;
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= nt_live) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * t + 4 * q4 + r;
      float add = RelB[i - j + (LP - 1)] + Mk[j];
      if (a.causal && j > i && Mk[j] == 0.f) add += masked;
      const float s = st[t][r] + add;
      st[t][r] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= nt_live) {
      st[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};   // a skipped tile's probabilities are exactly 0 (its keys carry -inf)
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __expf(st[t][r] - mx);
      st[t][r] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  // exact three-way split of every probability, packed as the B operand of its 32-key block
  uint4 ph[NB], pm[NB], pl[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    uint32_t hi[8], mid[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = 2 * u + (e >> 2);
      const float p = t < NT ? st[t < NT ? t : 0][e & 3] : 0.f;
      const uint32_t h1 = bf16_trunc_bits(p);
      const float r1 = p - __uint_as_float(h1);
      const uint32_t h2 = bf16_trunc_bits(r1);
      const float r2 = r1 - __uint_as_float(h2);
      hi[e] = h1 >> 16, mid[e] = h2 >> 16, lo[e] = bf16_trunc_bits(r2) >> 16;
    }
    ph[u] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    pm[u] = make_uint4(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16), mid[4] | (mid[5] << 16), mid[6] | (mid[7] << 16));
    pl[u] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
  }
#pragma unroll
  for (int dt = 0; dt < DK / 16; ++dt) {
    f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned char* vr = Vt + (16 * dt + c16) * VS + 8 * q4;   // Vt[d][32u + 16*(e/4) + 4*q4 + e%4]
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (2 * u >= nt_live) continue;
      const uint2 v0 = *reinterpret_cast<const uint2*>(vr + 64 * u);
      const uint2 v1 = *reinterpret_cast<const uint2*>(vr + 64 * u + 32);
      const bf16x8_t va = __builtin_bit_cast(bf16x8_t, make_uint4(v0.x, v0.y, v1.x, v1.y));
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, __builtin_bit_cast(bf16x8_t, ph[u]), o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, __builtin_bit_cast(bf16x8_t, pm[u]), o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, __builtin_bit_cast(bf16x8_t, pl[u]), o, 0, 0, 0);
    }
    if (i < L) {
      const int64_t off = (orow0 + i) * a.ldo + h * DK + 16 * dt + 4 * q4;
      const float4 ov = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      if (a.out_bf16)
        *reinterpret_cast<uint2*>(static_cast<__bf16*>(a.out_bf16) + off) = pack_bf16x4(ov.x, ov.y, ov.z, ov.w);
      else
        *reinterpret_cast<float4*>(a.out + off) = ov;
    }
  }
}

template <int NT>
static int launch_attention_mfma_bf16(const AttnArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)((a.Lk * 144 + 15) & ~15) + 64 * 272 + sizeof(float) * 3 * 16 * NT;
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(attention_mfma_bf16_kernel<NT>), 160 * 1024, "attention")) return rc__;
  hipLaunchKernelGGL(attention_mfma_bf16_kernel<NT>, dim3((unsigned)(a.B * a.H)), dim3(64 * NT), lds, stream, a);
  GDR_CHECK_LAUNCH("attention_mfma_bf16_kernel");
  return GDR_OK;
}

template <int NT>
static int launch_attention_mfma16(const AttnArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(float) * ((size_t)2 * a.Lk * 68 + 3 * 16 * NT);
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(attention_mfma16_kernel<NT>), 160 * 1024, "attention")) return rc__;
  hipLaunchKernelGGL(attention_mfma16_kernel<NT>, dim3((unsigned)(a.B * a.H)), dim3(64 * NT), lds, stream, a);
  GDR_CHECK_LAUNCH("attention_mfma16_kernel");
  return GDR_OK;
}


// ------------------------------------------------------------------------------------------ beam rows x encoder keys on MFMA
// Decode-time cross-attention at scale (T5Attention over the encoder states, modeling_t5.py:316-421, for the R beam rows of a
// query, which all sit at decoder position q_pos0 and share the query's K / V): the generic kernel above walks one query row
// per wave with lane <-> key and re-reads the whole K / V image of the head from LDS for every row — 90 LDS instructions per
// row, which is what bounds it once the grid fills the chip (15 360 beam rows x 12 heads: 229 us per call at 40 keys, against
// 25 us for the K / V bytes at the HBM rate).  Here a wave owns a 16-row tile: S^T = K.Q^T on v_mfma_f32_16x16x4_f32 (a lane
// holds 4 of every 16 keys' scores of ONE beam row, exactly the encoder kernel's scheme), softmax with two shuffles, and the
// probability registers are the B operand of the P.V MFMAs; K / V are read from LDS once per 16 rows.  Same arithmetic:
// (score + bias) + mask, fp32 softmax with expf, P.V — only the summation order over d and over keys is the MFMA's.
// One workgroup per (query, head); all its waves stage K / V, then ceil(Lq / 16) of them compute.
template <int NT>  // 16-key tiles: ceil(Lk / 16), 1..8
__global__ __launch_bounds__(512) void attention_cross_mfma16_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (a.live && *a.live == 0) return;  // uniform: every query of the generate call is done
  constexpr int DK = 64, DS = DK + 4, LP = 16 * NT;
  const int Lk = a.Lk, Lq = a.Lq;
  float* Ks = smem;
  float* Vs = Ks + Lk * DS;
  float* Add = Vs + Lk * DS;  // [LP] bias + mask of key j (every row of the tile sits at position q_pos0); -inf past Lk
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int w = tid >> 6, lane = tid & 63, c16 = lane & 15, q4 = lane >> 4;
  const int64_t krow0 = (int64_t)(b / a.kv_group) * a.k_bstride;
  // this wave's query fragments (B operand): lane (c16, q4) holds Q[16w + c16][16jj + 4q4 .. +3]; in flight while K / V stage
  float4 qv[DK / 16];
  const bool has_rows = 16 * w < Lq;
  if (has_rows) {
    const int64_t qr = (int64_t)b * a.q_bstride + min(16 * w + c16, Lq - 1);
    const float* q32 = a.q + qr * a.ldq + h * DK + 4 * q4;
#pragma unroll
    for (int jj = 0; jj < DK / 16; ++jj) {
      float4 t = *reinterpret_cast<const float4*>(q32 + 16 * jj);
      t.x *= a.scale, t.y *= a.scale, t.z *= a.scale, t.w *= a.scale;
      qv[jj] = t;
    }
  }
  for (int j = tid; j < LP; j += nthr) {
    float v = -INFINITY;
    if (j < Lk) {
      v = 0.f;
      if (a.rel_bias) {
        int n = a.q_pos0 - j, bucket = 0;
        if (a.bidirectional) {
          if (n < 0) {
            bucket = a.num_buckets >> 1;
            n = -n;
          }
        } else if (n < 0) {
          n = 0;
        }
        bucket += a.lut.v[n < 127 ? n : 127];
        v = a.rel_bias[bucket * a.H + h];
      }
      bool allowed = true;
      if (a.causal) allowed = j <= a.q_pos0;
      if (a.key_mask) allowed = allowed && (a.key_mask[(int64_t)(b / a.kv_group) * a.mask_bstride + j] != 0);
      if (!allowed) v += a.causal_neg_inf ? -INFINITY : -1e9f;
    }
    Add[j] = v;
  }
  for (int e = tid; e < Lk * (DK / 4); e += nthr) {
    const int r = e >> 4, c = e & 15;
    const float4 k = *reinterpret_cast<const float4*>(a.k + (krow0 + r) * a.ldk + h * DK + 4 * c);
    const float4 v = *reinterpret_cast<const float4*>(a.v + (krow0 + r) * a.ldv + h * DK + 4 * c);
    *reinterpret_cast<float4*>(Ks + r * DS + 4 * c) = k;
    *reinterpret_cast<float4*>(Vs + r * DS + 4 * c) = v;
  }
  __syncthreads();
  if (!has_rows) return;  // staging-only waves (no barrier follows)
  f32x4_t st[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) st[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < DK / 16; ++jj) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 kv = *reinterpret_cast<const float4*>(Ks + min(16 * t + c16, Lk - 1) * DS + 4 * q4 + 16 * jj);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qv[jj].x, st[t], 0, 0, 0);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qv[jj].y, st[t], 0, 0, 0);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qv[jj].z, st[t], 0, 0, 0);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qv[jj].w, st[t], 0, 0, 0);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = st[t][r] + Add[16 * t + 4 * q4 + r];  // keys past Lk: -inf
      st[t][r] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = expf(st[t][r] - mx);  // exp(-inf) = 0 for the padding keys
      st[t][r] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  const int i = 16 * w + c16;
#pragma unroll
  for (int dt = 0; dt < DK / 16; ++dt) {
    f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * t + 4 * q4 + r;
        const float vv = Vs[min(j, Lk - 1) * DS + 16 * dt + c16];
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, st[t][r], o, 0, 0, 0);
      }
    }
    if (i < Lq) {
      const int64_t off = ((int64_t)b * a.o_bstride + i) * a.ldo + h * DK + 16 * dt + 4 * q4;
      const float4 ov = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      if (a.out_bf16)
        *reinterpret_cast<uint2*>(static_cast<__bf16*>(a.out_bf16) + off) = pack_bf16x4(ov.x, ov.y, ov.z, ov.w);
      else
        *reinterpret_cast<float4*>(a.out + off) = ov;
    }
  }
}

template <int NT>
static int launch_attention_cross_mfma16(const AttnArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(float) * ((size_t)2 * a.Lk * 68 + 16 * NT);
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(attention_cross_mfma16_kernel<NT>), 160 * 1024, "attention")) return rc__;
  const int nw = (a.Lq + 15) / 16;
  hipLaunchKernelGGL(attention_cross_mfma16_kernel<NT>, dim3((unsigned)(a.B * a.H)), dim3(64 * (nw < 4 ? 4 : nw)), lds, stream, a);
  GDR_CHECK_LAUNCH("attention_cross_mfma16_kernel");
  return GDR_OK;
}
// When the beam-row form above serves a call: the shared-K/V decode shape (all Lq rows at one position, d_kv = 64, finished q
// rows — not split-K slabs —, 16-byte rows).  Measured against the generic kernel inside generate() (tools/exp_cross_attn.py,
// profiles/r04_cross_attention_ab.txt): never slower — 768 (query, head) pairs x 30 beam rows 29.7 -> 28.2 ms per call, 6 144
// pairs x 10 rows 70.2 -> 67.0 ms, bf16 mode 6 144 pairs x 30 rows 56.9 -> 51.2 ms — and equal within noise where few pairs
// make the launch latency-bound (12 pairs x 100 rows), so there is no size threshold.
static bool cross_mfma_wanted(const AttnArgs& a) {
  return a.q_same_pos && a.Lq > 1 && a.Lq <= 128 && a.dk == 64 && !a.q_part && !a.kv_rows && !a.seq_off && !a.qkv_bf16 &&
         a.ldo % 4 == 0 && a.ldq % 4 == 0;
}

// ------------------------------------------------------------------------------------------ attention, Lq == 1
// Decode-time attention (one query row per batch entry): one WAVE per (row, head), four per workgroup, nothing staged in
// LDS but a score strip; same semantics as attention_kernel (bias, masks, kv_rows, kv_group).  K / V reads are COALESCED:
// LPR consecutive lanes read one row together (a 256-byte row = 16 lanes x 16 B), so an instruction covers 64/LPR rows in
// full cache lines (the lane-per-key form of round 1 touched 64 cache lines per instruction and ran at the texture
// path's pace: 138 -> 93 us per call at 5 120 rows x 12 heads); the 4-element partial dot products are summed over the
// LPR lanes of a row group by a butterfly, scores / probabilities pass through a per-wave LDS strip, and P·V accumulates
// a float4 per lane that is finally summed over the row groups.
template <int LPR>  // lanes per K / V row: 16 (dk <= 64) or 32 (dk <= 128)
__global__ __launch_bounds__(256) void attention_decode_rows_kernel(const AttnArgs a) {
  constexpr int RPI = 64 / LPR;  // rows per wave instruction
  __shared__ float strip[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + wave;
  if (item >= a.B * a.H) return;
  if (a.live && *a.live == 0) return;  // every query of the generate call is done
  const int b = item / a.H, h = item % a.H;
  if (a.b_count_dev && b >= (int)*a.b_count_dev) return;  // only the first *b_count_dev batch entries are live
  const int dk = a.dk, c4 = dk >> 2, Lk = a.Lk, kb = b / a.kv_group;
  const int g = lane / LPR, c = lane % LPR;
  const bool col_ok = c < c4;
  const int i_abs = a.q_pos0;
  const int half = a.num_buckets >> 1;
  const float masked = a.causal_neg_inf ? -INFINITY : -1e9f;
  float* S = strip[wave];
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) {
    q = *reinterpret_cast<const float4*>(a.q + (int64_t)b * a.q_bstride * a.ldq + h * dk + 4 * c);
    q.x *= a.scale, q.y *= a.scale, q.z *= a.scale, q.w *= a.scale;
  }
  auto row_of = [&](int j) -> int64_t {
    return a.kv_rows ? (int64_t)a.kv_rows[(int64_t)b * Lk + j] : (int64_t)kb * a.k_bstride + j;
  };
  // ---- scores: key j = j0 + g, its row read by the LPR lanes of group g (short key lists — a decode step — take
  // attention_decode_short_kernel below)
  float mx = -INFINITY;
  for (int j0 = 0; j0 < Lk; j0 += RPI) {
    const int j = j0 + g;
    float part = 0.f;
    if (j < Lk && col_ok) {
      const float4 kk = *reinterpret_cast<const float4*>(a.k + row_of(j) * a.ldk + h * dk + 4 * c);
      part = fmaf(q.x, kk.x, fmaf(q.y, kk.y, fmaf(q.z, kk.z, q.w * kk.w)));
    }
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (j < Lk) {
      float add = 0.f;
      if (a.rel_bias) {
        int n = i_abs - j, bucket = 0;
        if (a.bidirectional) {
          if (n < 0) {
            bucket = half;
            n = -n;
          }
        } else if (n < 0) {
          n = 0;
        }
        bucket += a.lut.v[n < 127 ? n : 127];
        add = a.rel_bias[bucket * a.H + h];
      }
      bool allowed = true;
      if (a.causal) allowed = j <= i_abs;
      if (a.key_mask) allowed = allowed && (a.key_mask[(int64_t)kb * a.mask_bstride + j] != 0);
      if (!allowed) add += masked;
      const float sj = part + add;
      if (c == 0) S[j] = sj;
      mx = fmaxf(mx, sj);
    }
  }
  mx = wave_max(mx);
  __builtin_amdgcn_wave_barrier();  // the strip is private to this wave: LDS operations of one wave complete in order
  // ---- softmax over the strip (lane <-> key), probabilities back into the strip
  float p0 = 0.f, p1 = 0.f;
  if (lane < Lk) p0 = expf(S[lane] - mx);
  if (lane + 64 < Lk) p1 = expf(S[lane + 64] - mx);
  const float inv = 1.0f / wave_sum(p0 + p1);
  __builtin_amdgcn_wave_barrier();
  if (lane < Lk) S[lane] = p0 * inv;
  if (lane + 64 < Lk) S[lane + 64] = p1 * inv;
  __builtin_amdgcn_wave_barrier();
  // ---- O = P·V: group g takes keys j0 + g, lane (g, c) accumulates columns 4c..4c+3; then the groups are summed
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int j0 = 0; j0 < Lk; j0 += RPI) {
    const int j = j0 + g;
    if (j < Lk && col_ok) {
      const float4 vv = *reinterpret_cast<const float4*>(a.v + row_of(j) * a.ldv + h * dk + 4 * c);
      const float pj = S[j];
      o.x = fmaf(pj, vv.x, o.x), o.y = fmaf(pj, vv.y, o.y), o.z = fmaf(pj, vv.z, o.z), o.w = fmaf(pj, vv.w, o.w);
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    o.x += __shfl_xor(o.x, off), o.y += __shfl_xor(o.y, off), o.z += __shfl_xor(o.z, off), o.w += __shfl_xor(o.w, off);
  }
  if (g == 0 && col_ok) {
    const int64_t off = (int64_t)b * a.o_bstride * a.ldo + h * dk + 4 * c;
    if (a.out_bf16)  // the context only feeds the next bf16-mode linear: emit its operand directly (same RNE as the cast kernel)
      *reinterpret_cast<uint2*>(static_cast<__bf16*>(a.out_bf16) + off) = pack_bf16x4(o.x, o.y, o.z, o.w);
    else
      *reinterpret_cast<float4*>(a.out + off) = o;
  }
}

// Short key lists (a decode step: Lk <= 16 resp. 12 keys, the ancestors reached through kv_rows): the form above spends its
// time in the vector ALU, not in memory (rocprofv3 counters at 15 360 rows x 12 heads, profiles/r04_decode_attention_pmc.txt:
// 375 VALU instructions per wave = 3/4 of the SIMDs' issue cycles, 69 % of wave time parked, HBM at ~2 of 8 TB/s) — 64-bit index multiplies, LDS-permute butterflies and a
// relative-position bias looked up by every lane for every key.  Same arithmetic per (row, head), in the same order (bit-
// identical output), with the bookkeeping cut down: every memory operand of the wave is requested up front (lane j fetches
// the row index of key j, a shuffle hands it to the lanes that read that row; V does not depend on the scores); row offsets
// are one 32 x 32 -> 64-bit multiply-add; the sum over the LPR lanes of a key is four DPP row rotations (the xor butterfly's
// pairs exactly: after the xor-8 step the values have period 8 inside a row, so "rotate by 4" meets the same partner as
// "xor 4", and so on down); bias and mask are added where lane <-> key (once per key, not once per lane per key); max and
// sum of the <= 16 scores stay inside DPP row 0.
template <int LPR>
__global__ __launch_bounds__(256) void attention_decode_short_kernel(const AttnArgs a) {
  constexpr int RPI = 64 / LPR;
  constexpr int MAXIT = LPR == 16 ? 4 : 6;  // 16 / 12 keys
  __shared__ float strip[4][16];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave-uniform: b, h and every
                                                                                          // offset made of them live on the scalar unit
  // XCD-aware order: the hardware deals consecutive workgroups round-robin over the 8 XCDs, each with an L2 of
  // its own.  The R beam rows of a query reach mostly the SAME ancestor K / V rows (their prefixes are shared), and their H
  // (row, head) items are neighbours in the item order — dealt round-robin, every XCD fetched its own copy of those rows from
  // HBM / MALL.  With each XCD owning a contiguous range of items a query's rows meet in one L2 (-0.3 % on generate() at 512 x 30).
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q_ = nblk >> 3, r_ = nblk & 7u, xcd_ = bid & 7u, j_ = bid >> 3;
    bid = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + j_;
  }
  const int item = (int)bid * 4 + wave;
  if (item >= a.B * a.H) return;
  if (a.live && *a.live == 0) return;  // every query of the generate call is done
  const int b = item / a.H, h = item % a.H;
  if (a.b_count_dev && b >= (int)*a.b_count_dev) return;
  const int dk = a.dk, Lk = a.Lk, kb = b / a.kv_group;
  const int g = lane / LPR, c = lane % LPR;
  const bool col_ok = c < (dk >> 2);
  const int ldk = (int)a.ldk, ldv = (int)a.ldv;
  const int col = h * dk + 4 * c;
  float* S = strip[wave];
  int myrow = 0;
  if (lane < Lk) myrow = a.kv_rows ? a.kv_rows[(int64_t)b * Lk + lane] : kb * (int)a.k_bstride + lane;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) q = *reinterpret_cast<const float4*>(a.q + (int64_t)b * a.q_bstride * a.ldq + col);
  // bias + mask of key j, by lane j
  float add = 0.f;
  if (lane < Lk) {
    const int j = lane, i_abs = a.q_pos0;
    if (a.rel_bias) {
      int n = i_abs - j, bucket = 0;
      if (a.bidirectional) {
        if (n < 0) {
          bucket = a.num_buckets >> 1;
          n = -n;
        }
      } else if (n < 0) {
        n = 0;
      }
      bucket += a.lut.v[n < 127 ? n : 127];
      add = a.rel_bias[bucket * a.H + h];
    }
    bool allowed = true;
    if (a.causal) allowed = j <= i_abs;
    if (a.key_mask) allowed = allowed && (a.key_mask[(int64_t)kb * a.mask_bstride + j] != 0);
    if (!allowed) add += a.causal_neg_inf ? -INFINITY : -1e9f;
  }
  float4 kreg[MAXIT], vreg[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = it * RPI + g;
    const int rj = __shfl(myrow, j < Lk ? j : 0);
    if (j < Lk && col_ok) {  // (read below under the same condition only)
      kreg[it] = *reinterpret_cast<const float4*>(a.k + ((int64_t)rj * ldk + col));
      vreg[it] = *reinterpret_cast<const float4*>(a.v + ((int64_t)rj * ldv + col));
    }
  }
  q.x *= a.scale, q.y *= a.scale, q.z *= a.scale, q.w *= a.scale;
  // ---- q·k of key j = it * RPI + g, summed over the LPR lanes of group g
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = it * RPI + g;
    if (it * RPI >= Lk) break;
    float part = 0.f;
    if (j < Lk && col_ok) {
      const float4 kk = kreg[it];
      part = fmaf(q.x, kk.x, fmaf(q.y, kk.y, fmaf(q.z, kk.z, q.w * kk.w)));
    }
    if (LPR == 32) part += __shfl_xor(part, 16);
    part = row16_sum(part);
    if (c == 0 && j < Lk) S[j] = part;
  }
  __builtin_amdgcn_wave_barrier();  // the strip is private to this wave: LDS operations of one wave complete in order
  // ---- softmax, lane <-> key (all keys sit in DPP row 0)
  const float sj = lane < Lk ? S[lane] + add : -INFINITY;
  const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, row16_max(sj))));
  const float p0 = lane < Lk ? expf(sj - mx) : 0.f;
  const float inv = 1.0f / __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, row16_sum(p0))));
  __builtin_amdgcn_wave_barrier();
  if (lane < Lk) S[lane] = p0 * inv;
  __builtin_amdgcn_wave_barrier();
  // ---- O = P·V: group g takes keys it * RPI + g, lane (g, c) accumulates columns 4c..4c+3; then the groups are summed
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int j = it * RPI + g;
    if (j < Lk && col_ok) {
      const float4 vv = vreg[it];
      const float pj = S[j];
      o.x = fmaf(pj, vv.x, o.x), o.y = fmaf(pj, vv.y, o.y), o.z = fmaf(pj, vv.z, o.z), o.w = fmaf(pj, vv.w, o.w);
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    o.x += __shfl_xor(o.x, off), o.y += __shfl_xor(o.y, off), o.z += __shfl_xor(o.z, off), o.w += __shfl_xor(o.w, off);
  }
  if (g == 0 && col_ok) {
    const int64_t off = (int64_t)b * a.o_bstride * a.ldo + col;
    if (a.out_bf16)  // the context only feeds the next bf16-mode linear: emit its operand directly (same RNE as the cast kernel)
      *reinterpret_cast<uint2*>(static_cast<__bf16*>(a.out_bf16) + off) = pack_bf16x4(o.x, o.y, o.z, o.w);
    else
      *reinterpret_cast<float4*>(a.out + off) = o;
  }
}

// Four heads per wave (r05; d_kv = 64, H % 4 == 0, Lq = 1, <= 16 keys): at thousands of beam rows the form above is bound by its
// INSTRUCTIONS — a wave per (row, head) runs ~175 vector instructions for 18 multiply-adds per lane (184 000 waves at 15 360 rows: its
// launch time did not move when its bytes were halved, its loads de-duplicated or its heads batched per wave,
// profiles/r05_decode_attention_ab.txt).  Here a wave serves FOUR adjacent heads of a row, one per 16-lane DPP row: lane (hg, c) owns
// columns 4c .. 4c+3 of head 4*h0 + hg, so a key's K (or V) slice of the four heads is ONE contiguous 1 KB wave load (the form above
// gathers four different rows per load), the row table is read once per four heads, key indices are scalar (readlane), and every
// reduction (q.k over a head's 16 lanes, softmax max / sum over a head's keys) is the same DPP row operation on all four rows at
// once.  Per (row, head) the arithmetic and its order are the form above's — q.k: the lane's 4-term fmaf chain, then row16_sum;
// softmax: lane c <-> key c inside the row; P.V: four partial sums over the keys j = g (mod 4) in ascending order combined as
// (a0 + a1) + (a2 + a3), which is what the other form's two xor-shuffles compute — so the output is bit-identical
// (tools/exp_ab_bits.py, tests/test_gpu_decode.py).
template <int MAXK>
__global__ __launch_bounds__(256) void attention_decode_heads4_kernel(const AttnArgs a) {
  __shared__ float strip[4][4][16];  // [wave][head row][key]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q_ = nblk >> 3, r_ = nblk & 7u, xcd_ = bid & 7u, j_ = bid >> 3;
    bid = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + j_;
  }
  const int HG = a.H >> 2;  // head quads per row
  const int item = (int)bid * 4 + wave;
  if (item >= a.B * HG) return;
  if (a.live && *a.live == 0) return;  // every query of the generate call is done
  const int b = item / HG, h0 = item % HG;
  if (a.b_count_dev && b >= (int)*a.b_count_dev) return;
  const int Lk = a.Lk, kb = b / a.kv_group;
  const int hg = lane >> 4, c = lane & 15, h = h0 * 4 + hg;
  const int colw = h0 * 256 + lane * 4;  // this lane's first column: the wave covers 256 consecutive floats of a row
  float* S = strip[wave][hg];
  int myrow = 0;
  if (lane < Lk) myrow = a.kv_rows ? a.kv_rows[(int64_t)b * Lk + lane] : kb * (int)a.k_bstride + lane;
  float4 q = *reinterpret_cast<const float4*>(a.q + (int64_t)b * a.q_bstride * a.ldq + colw);
  // bias + mask of key c of this lane's head (lane c <-> key c inside every DPP row)
  float add = 0.f;
  if (c < Lk) {
    const int j = c, i_abs = a.q_pos0;
    if (a.rel_bias) {
      int n = i_abs - j, bucket = 0;
      if (a.bidirectional) {
        if (n < 0) {
          bucket = a.num_buckets >> 1;
          n = -n;
        }
      } else if (n < 0) {
        n = 0;
      }
      bucket += a.lut.v[n < 127 ? n : 127];
      add = a.rel_bias[bucket * a.H + h];
    }
    bool allowed = true;
    if (a.causal) allowed = j <= i_abs;
    if (a.key_mask) allowed = allowed && (a.key_mask[(int64_t)kb * a.mask_bstride + j] != 0);
    if (!allowed) add += a.causal_neg_inf ? -INFINITY : -1e9f;
  }
  float4 kreg[MAXK], vreg[MAXK];
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    if (j < Lk) {  // uniform
      const int64_t rj = __builtin_amdgcn_readlane(myrow, j);
      kreg[j] = *reinterpret_cast<const float4*>(a.k + (rj * a.ldk + colw));
      vreg[j] = *reinterpret_cast<const float4*>(a.v + (rj * a.ldv + colw));
    }
  }
  q.x *= a.scale, q.y *= a.scale, q.z *= a.scale, q.w *= a.scale;
  float sj = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    if (j < Lk) {
      const float4 kk = kreg[j];
      float part = fmaf(q.x, kk.x, fmaf(q.y, kk.y, fmaf(q.z, kk.z, q.w * kk.w)));
      part = row16_sum(part);  // every lane of the head's row holds the score of key j
      if (c == j) sj = part + add;
    }
  }
  const float mx = row16_max(sj);
  const float p0 = c < Lk ? expf(sj - mx) : 0.f;
  const float inv = 1.0f / row16_sum(p0);
  if (c < Lk) S[c] = p0 * inv;
  __builtin_amdgcn_wave_barrier();  // the strip is private to this wave: LDS operations of one wave complete in order
  float4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    if (j < Lk) {
      const float4 vv = vreg[j];
      const float pj = S[j];
      float4& o = acc[j & 3];
      o.x = fmaf(pj, vv.x, o.x), o.y = fmaf(pj, vv.y, o.y), o.z = fmaf(pj, vv.z, o.z), o.w = fmaf(pj, vv.w, o.w);
    }
  }
  float4 o;
  o.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), o.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
  o.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), o.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
  const int64_t off = (int64_t)b * a.o_bstride * a.ldo + colw;
  if (a.out_bf16)
    *reinterpret_cast<uint2*>(static_cast<__bf16*>(a.out_bf16) + off) = pack_bf16x4(o.x, o.y, o.z, o.w);
  else
    *reinterpret_cast<float4*>(a.out + off) = o;
}

// Measured and dropped (r05): a workgroup per (query, head) that numbers the DISTINCT (position, row) pairs of the query's R beam rows,
// stages each of them once in LDS and serves all R rows from there (97 distinct of 270 gathered rows per query at 30 beams,
// tools/exp_prefix_sharing.py) — bit-identical, but generate() at 512 x 30 beams took 47.9 ms (50.3 with the queries prefetched and a
// compare-based numbering) against 46.2 ms for the per-row form above: that form is bound by wave slots x the two-deep load chain
// (kv_rows -> K / V), the shared rows already meet in L2, and the staging adds barriers and a third dependent phase.
int launch_attention(const AttnArgs& a, hipStream_t stream) {
  GDR_CHECK_ARG(a.dk % 4 == 0 && a.dk >= 4 && a.dk <= 256, "attention: dk=%d unsupported", a.dk);
  GDR_CHECK_ARG(a.Lk >= 1 && a.Lk <= 128, "attention: Lk=%d must be in [1,128]", a.Lk);
  GDR_CHECK_ARG(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0, "attention: row strides must be multiples of 4");
  GDR_CHECK_ARG(!a.rel_bias || (a.num_buckets >= 2 && a.num_buckets <= 256), "attention: bad num_buckets");
  GDR_CHECK_ARG(a.kv_group >= 1, "attention: kv_group must be >= 1");
  if (a.B == 0 || a.Lq == 0) return GDR_OK;
  ProfScope prof(PROF_ATTENTION, 4.0 * a.B * a.H * (double)a.Lq * a.Lk * a.dk, stream);
  GDR_CHECK_ARG(!a.q_part || (a.Lq > 1 && a.q_same_pos && a.dk % 4 == 0 && a.q_S >= 1 && (a.H * a.dk) % 4 == 0),
                "attention: slab-sourced q serves the shared-K/V decode form (generic kernel) only");
  // Lq = 1 (a decode step): the row-group kernel serves dk <= 128 with 16-byte output rows; anything else (wide heads,
  // odd output strides) falls through to the generic kernel below, which handles every dk <= 256 — only the device-side
  // batch count of the prefix-table mode exists in the row-group form alone
  GDR_CHECK_ARG(!(a.Lq == 1 && a.b_count_dev && !(a.dk <= 128 && a.ldo % 4 == 0)),
                "attention: a device-side batch count needs the Lq = 1 row-group form (dk <= 128, ldo %% 4 == 0; dk=%d ldo=%lld)", a.dk,
                (long long)a.ldo);
  if (a.Lq == 1 && a.dk <= 128 && a.ldo % 4 == 0) {
    if (a.Lk <= (a.dk <= 64 ? 16 : 12)) {  // a decode step's key list
      // the beam rows of a query share their ancestors: one workgroup per (query, head) stages the distinct K / V rows once
#ifndef GDR_LAB_ATTN_NO_HEADS4
      // from ~1 400 beam rows on: four heads per wave (at 640 rows x 12 heads the two forms tie: 12.07 against 12.02 ms per generate())
      if (a.dk == 64 && a.H % 4 == 0 && !a.q_part && (int64_t)a.B * a.H >= 16384) {
        const dim3 g4((unsigned)((a.B * (a.H / 4) + 3) / 4));
        if (a.Lk <= 4)
          hipLaunchKernelGGL(attention_decode_heads4_kernel<4>, g4, dim3(256), 0, stream, a);
        else if (a.Lk <= 8)
          hipLaunchKernelGGL(attention_decode_heads4_kernel<8>, g4, dim3(256), 0, stream, a);
        else if (a.Lk <= 12)
          hipLaunchKernelGGL(attention_decode_heads4_kernel<12>, g4, dim3(256), 0, stream, a);
        else
          hipLaunchKernelGGL(attention_decode_heads4_kernel<16>, g4, dim3(256), 0, stream, a);
        GDR_CHECK_LAUNCH("attention_decode_heads4_kernel");
        return GDR_OK;
      }
#endif
      const dim3 grids((unsigned)((a.B * a.H + 3) / 4));
      if (a.dk <= 64)
        hipLaunchKernelGGL(attention_decode_short_kernel<16>, grids, dim3(256), 0, stream, a);
      else
        hipLaunchKernelGGL(attention_decode_short_kernel<32>, grids, dim3(256), 0, stream, a);
      GDR_CHECK_LAUNCH("attention_decode_short_kernel");
      return GDR_OK;
    }
    const dim3 grid((unsigned)((a.B * a.H + 3) / 4));
    if (a.dk <= 64)
      hipLaunchKernelGGL(attention_decode_rows_kernel<16>, grid, dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL(attention_decode_rows_kernel<32>, grid, dim3(256), 0, stream, a);
    GDR_CHECK_LAUNCH("attention_decode_rows_kernel");
    return GDR_OK;
  }
  const bool packed = a.seq_off != nullptr || a.qkv_bf16;  // both exist in the MFMA form only
  GDR_CHECK_ARG(!a.seq_off || a.seq_len, "attention: seq_off without seq_len");
  GDR_CHECK_ARG(!packed || (a.Lq == a.Lk && a.q_pos0 == 0 && !a.kv_rows && a.kv_group == 1 && !a.q_same_pos &&
                            a.dk == 64 && a.ldo % 4 == 0),
                "attention: the packed (ragged) form and bf16 q/k/v serve full self-attention with d_kv = 64 only");
  GDR_CHECK_ARG(!a.qkv_bf16 || (a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0), "attention: bf16 q/k/v need row strides %% 8 == 0");
  if (a.Lq == a.Lk && a.q_pos0 == 0 && !a.kv_rows && a.kv_group == 1 && !a.q_same_pos && a.dk == 64 && a.ldo % 4 == 0) {
    if (a.qkv_bf16 && a.scale == 1.0f) {  // bf16 operands as they stand: the bf16-MFMA form
      switch ((a.Lk + 15) / 16) {
        case 1: return launch_attention_mfma_bf16<1>(a, stream);
        case 2: return launch_attention_mfma_bf16<2>(a, stream);
        case 3: return launch_attention_mfma_bf16<3>(a, stream);
        case 4: return launch_attention_mfma_bf16<4>(a, stream);
        case 5: return launch_attention_mfma_bf16<5>(a, stream);
        case 6: return launch_attention_mfma_bf16<6>(a, stream);
        case 7: return launch_attention_mfma_bf16<7>(a, stream);
        default: return launch_attention_mfma_bf16<8>(a, stream);
      }
    }
    switch ((a.Lk + 15) / 16) {
      case 1: return launch_attention_mfma16<1>(a, stream);
      case 2: return launch_attention_mfma16<2>(a, stream);
      case 3: return launch_attention_mfma16<3>(a, stream);
      case 4: return launch_attention_mfma16<4>(a, stream);
      case 5: return launch_attention_mfma16<5>(a, stream);
      case 6: return launch_attention_mfma16<6>(a, stream);
      case 7: return launch_attention_mfma16<7>(a, stream);
      default: return launch_attention_mfma16<8>(a, stream);
    }
  }
  if (cross_mfma_wanted(a)) {
    switch ((a.Lk + 15) / 16) {
      case 1: return launch_attention_cross_mfma16<1>(a, stream);
      case 2: return launch_attention_cross_mfma16<2>(a, stream);
      case 3: return launch_attention_cross_mfma16<3>(a, stream);
      case 4: return launch_attention_cross_mfma16<4>(a, stream);
      case 5: return launch_attention_cross_mfma16<5>(a, stream);
      case 6: return launch_attention_cross_mfma16<6>(a, stream);
      case 7: return launch_attention_cross_mfma16<7>(a, stream);
      default: return launch_attention_cross_mfma16<8>(a, stream);
    }
  }
  const int dks = a.dk + 4, Lkp = (a.Lk + 3) & ~3;
  int chunks = (512 + a.B * a.H - 1) / (a.B * a.H);  // aim at ~2 workgroups per CU, at least 4 rows (one per wave) each
  const int max_chunks = (a.Lq + 3) / 4;
  chunks = chunks < 1 ? 1 : (chunks > max_chunks ? max_chunks : chunks);
  int rows_per = (a.Lq + chunks - 1) / chunks;
  if (a.q_part && rows_per > 32) {  // slab-sourced q rows are summed while they are staged: keep a workgroup's share stageable
    chunks = (a.Lq + 31) / 32;
    rows_per = (a.Lq + chunks - 1) / chunks;
  }
  const int qrows = rows_per <= 32 ? ((rows_per + 3) & ~3) : 4;  // as the kernel lays its query strip out
  const size_t lds = sizeof(float) * ((size_t)2 * a.Lk * dks + (size_t)qrows * a.dk + 4 * Lkp + 256);
  GDR_CHECK_ARG(lds <= 160 * 1024, "attention: Lk=%d dk=%d needs %zu B of LDS", a.Lk, a.dk, lds);
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(attention_kernel), 160 * 1024, "attention")) return rc__;
  hipLaunchKernelGGL(attention_kernel, dim3((unsigned)(a.B * a.H), (unsigned)chunks), dim3(256), lds, stream, a);
  GDR_CHECK_LAUNCH("attention_kernel");
  return GDR_OK;
}

}  // namespace gdr

extern "C" int gdr_t5_relative_bucket_table(int bidirectional, int num_buckets, int max_distance, int qlen, int klen,
                                            int32_t* out_host) {
  using namespace gdr;
  GDR_CHECK_ARG(out_host && num_buckets >= 2 && num_buckets <= 256 && qlen > 0 && klen > 0, "bucket_table: bad args");
  const int nb = bidirectional ? num_buckets / 2 : num_buckets;
  const BucketLut lut = make_bucket_lut(nb, max_distance);
  for (int i = 0; i < qlen; ++i)
    for (int j = 0; j < klen; ++j) {
      int n = i - j, bucket = 0;
      if (bidirectional) {
        if (n < 0) {
          bucket = nb;
          n = -n;
        }
      } else if (n < 0) {
        n = 0;
      }
      out_host[i * klen + j] = bucket + lut.v[n < 127 ? n : 127];
    }
  return GDR_OK;
}

extern "C" int gdr_t5_layer_norm(const float* x, const float* w, float* y, int64_t rows, int d, float eps, void* stream_) {
  using namespace gdr;
  GDR_CHECK_ARG(x && w && y && rows >= 0 && d > 0 && d % 4 == 0, "t5_layer_norm: bad arguments (d %% 4 == 0)");
  return launch_rmsnorm(x, w, y, rows, d, eps, nullptr, 1, static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_l2_normalize(const float* x, float* y, int64_t rows, int d, float eps, void* stream_) {
  using namespace gdr;
  GDR_CHECK_ARG(x && y && rows >= 0 && d > 0, "l2_normalize: bad arguments");
  if (rows == 0) return GDR_OK;
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream_), x, y,
                     rows, d, eps);
  GDR_CHECK_LAUNCH("l2_normalize_kernel");
  return GDR_OK;
}
