// Fused sub-blocks of one T5 decoder block at decode time, for M = batch * beams <= 1 024 rows (r06).
//
// A decode step of one block is a chain of dependent phases over the same M rows (modeling_t5.py:498-584 at Lq = 1):
//     norm -> q/k/v (:360-364) -> cache append -> attention over <= 16 ancestor keys (:384-403) -> o (:413) -> +residual (:452)
//     norm -> q (:360) -> attention over the query's L encoder keys -> o -> +residual
//     norm -> wi -> ReLU -> wo (:182-185) -> +residual (:199)
// As separate launches that is 12-13 kernels of 5-25 us per block and step, each one global-load -> compute -> store latency
// chain on a grid that cannot fill the chip (profiles/r05_generate_steps.txt: 1 054 launches per generate() at 64 x 10 beams,
// 0.40 of the executed floor; 1 208 at 1 x 100, 0.12).  The rows are independent through a whole block for one step, and the
// three sub-blocks factor over a SLICE of their inner dimension:
//     self-attention:   head h needs only the 3 x 64 rows of wqkv that produce its q/k/v, and contributes ctx_h . wo[:, 64h:64h+64]^T
//     cross-attention:  head h needs the 64 rows of wq_c, the query's cached K/V of head h, and contributes ctx_h . wo_c[:, 64h..]^T
//     feed-forward:     a chunk of d_ff columns needs those rows of wi and contributes relu(.) . wo_ff[:, chunk]^T
// so ONE workgroup owns (a panel of 16 / 32 rows, a slice): phase 1 = X[panel, d] . W1[slice, d]^T on v_mfma_f32_16x16x4_f32
// (operands staged global -> registers (three chunks of 32 k in flight) -> LDS), the middle (attention / ReLU) on the rows in
// LDS, phase 3 = mid[panel, K3] . W3[:, slice]^T -> a PARTIAL of the sub-block's output rows, written as slab `slice` of
// [S][M][d].  slab_reduce_norm_row_kernel folds the S slabs in fixed order s = 0 .. S-1, adds bias / residual and applies the
// norm that always follows (the arithmetic of splitk_reduce_norm_row_kernel).  A block of a step is 6 launches instead of 12-13,
// every sum has a fixed order (deterministic), and nothing but slabs, the K/V cache rows and the residual stream touches HBM.
//
// The adaptor's post-LN nn.TransformerDecoderLayer (modeling_t5.py:1241-1244,1615-1633) takes the same three forms with biases:
// self-attention heads of width d / nhead (96 at t5-base) run as slices of one head each when that width is 64, else through the
// unfused kernels; its feed-forward (linear1 -> ReLU -> linear2) is the FFN form with bias1 in phase 1 and bias2 in the reduction.
#include <stdlib.h>

#include <type_traits>

#include "decode_fused.h"

namespace gdr {

typedef float f32x4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int t5_bucket(int n, int bidirectional, int num_buckets, const BucketLut& lut) {
  int bucket = 0;
  if (bidirectional) {
    if (n < 0) {
      bucket = num_buckets >> 1;
      n = -n;
    }
  } else if (n < 0) {
    n = 0;
  }
  return bucket + lut.v[n < 127 ? n : 127];
}

// Dynamic LDS layout (floats):
//   [stage region]  phase 1: As[2][RM][36], Ws[2][N1][36]; later reused by the cross-attention K / V image and by phase 3's W3 chunk
//   [Cs]            [RM][N1 + 4]   phase 1's output rows (q | k | v, q, or relu(h))
//   [Ctx]           [RM][68]       attention context (SA / CA)
//   [Add]           [128]          bias + mask per key
//   [Strip]         [16][128]      cross-attention scores per row group
template <int MODE, int RT, int N1>
struct FusedCfg {
  static constexpr int RM = 16 * RT;
  static constexpr int K3 = MODE == FUSED_FFN ? N1 : 64;
  static constexpr int NC3 = 16384 / K3;          // output columns per phase-3 chunk (64 KB of W3): 256 / 128 / 64
  static constexpr int CT = N1 / 64;              // phase-1 column tiles (16 wide) per wave
  static constexpr int CT3 = NC3 / 64;            // phase-3 column tiles per wave
  static constexpr int SLD = 36;
  static constexpr int CLD = N1 + 4;
  static constexpr int XLD = 68;
  static constexpr int LD3 = K3 + 4;
  static constexpr int P1_FLOATS = 2 * (RM + N1) * SLD;
  static constexpr int P3_FLOATS = NC3 * LD3;
  __host__ __device__ static constexpr int stage_floats(int L) {
    int s = P1_FLOATS > P3_FLOATS ? P1_FLOATS : P3_FLOATS;
    const int kv = MODE == FUSED_CA ? 2 * L * XLD : 0;
    return s > kv ? s : kv;
  }
  __host__ __device__ static constexpr int lds_floats(int L) {
    return stage_floats(L) + RM * CLD + (MODE == FUSED_FFN ? 0 : RM * XLD) + 128 + (MODE == FUSED_CA ? 16 * 128 : 0);
  }
};

template <int MODE, int RT, int N1>
__global__ __launch_bounds__(256) void decode_fused_kernel(const FusedArgs g) {
  using Cfg = FusedCfg<MODE, RT, N1>;
  constexpr int RM = Cfg::RM, K3 = Cfg::K3, NC3 = Cfg::NC3, CT = Cfg::CT, CT3 = Cfg::CT3, SLD = Cfg::SLD, CLD = Cfg::CLD,
                XLD = Cfg::XLD, LD3 = Cfg::LD3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (g.live && *g.live == 0) return;  // uniform: every query of the generate call is done
  float* const stage = smem;
  float* const Cs = smem + Cfg::stage_floats(MODE == FUSED_CA ? g.L : 0);
  float* const Ctx = Cs + RM * CLD;                                   // SA / CA only
  float* const Add = Ctx + (MODE == FUSED_FFN ? 0 : RM * XLD);
  float* const Strip = Add + 128;                                     // CA only
  // XCD-aware order: consecutive workgroups are dealt round-robin over the 8 XCDs; with each XCD owning a contiguous range of
  // (slice, panel) items its workgroups share few weight slices, which its L2 then fetches once for all of their row panels
  unsigned bid = blockIdx.x;
  {
    const unsigned nblk = gridDim.x, q_ = nblk >> 3, r_ = nblk & 7u, xcd_ = bid & 7u, j_ = bid >> 3;
    bid = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + j_;
  }
  const int slice = (int)(bid / (unsigned)g.n_panels), panel = (int)(bid % (unsigned)g.n_panels);
  const int64_t m0 = (int64_t)panel * RM;
  const int rows_here = (int)((g.M - m0) < RM ? (g.M - m0) : RM);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c16 = lane & 15, q4 = lane >> 4;
  const int d = g.d, nk = d / 32;

  // ------------------------------------------------------------------------------------------ phase 1
  // staging map: float4 index e = tid + 256 i over [rows][8 float4]; 8 lanes cover one 128-B row segment
  constexpr int NA = (RM * 8 + 255) / 256;   // 1
  constexpr int NW = N1 * 8 / 256;           // 2 / 4 / 6 / 8
  const float* a_src[NA];
  const float* w_src[NW];
  int a_st[NA], w_st[NW];
  bool a_on[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = tid + 256 * i, r = e >> 3, c4 = e & 7;
    a_on[i] = r < RM;
    const int64_t m = m0 + (r < rows_here ? r : rows_here - 1);
    a_src[i] = g.X + m * d + 4 * c4;
    a_st[i] = (r < RM ? r : 0) * SLD + 4 * c4;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int e = tid + 256 * i, c = e >> 3, c4 = e & 7;
    const int wrow = MODE == FUSED_SA ? (c >> 6) * g.w1_seg_stride + slice * g.w1_slice_rows + (c & 63) : slice * g.w1_slice_rows + c;
    w_src[i] = g.W1 + (int64_t)wrow * d + 4 * c4;
    w_st[i] = c * SLD + 4 * c4;
  }
  float* const As = stage;
  float* const Ws = stage + 2 * RM * SLD;
  f32x4f acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4f){0.f, 0.f, 0.f, 0.f};
  f32x4f pa[NA], pw[NW], qa[NA], qw[NW], ra[NA], rw[NW];  // first-class vectors: arrays of HIP's float4 struct stayed in scratch (allocas not promoted)
#define F_LOAD(R, kt_)                                                                              \
  {                                                                                                 \
    const int t_ = (kt_) < nk ? (kt_) : nk - 1; /* past the end: re-fetch the last chunk */         \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) R##a[i] = *reinterpret_cast<const f32x4f*>(a_src[i] + t_ * 32); \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) R##w[i] = *reinterpret_cast<const f32x4f*>(w_src[i] + t_ * 32); \
  }
#define F_STORE(R, buf_)                                                                            \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) if (a_on[i]) *reinterpret_cast<f32x4f*>(As + (buf_)*RM * SLD + a_st[i]) = R##a[i]; \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) *reinterpret_cast<f32x4f*>(Ws + (buf_)*N1 * SLD + w_st[i]) = R##w[i];              \
  }
#define F_COMPUTE(buf_)                                                                             \
  {                                                                                                 \
    const float* a_ = As + (buf_)*RM * SLD + c16 * SLD + 4 * q4;                                    \
    const float* b_ = Ws + (buf_)*N1 * SLD + (wave * 16 * CT + c16) * SLD + 4 * q4;                 \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                              \
      float4 av[RT], bv[CT];                                                                        \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) av[rt] = *reinterpret_cast<const float4*>(a_ + rt * 16 * SLD + 16 * jj); \
      _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) bv[ct] = *reinterpret_cast<const float4*>(b_ + ct * 16 * SLD + 16 * jj); \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                     \
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].x, bv[ct].x, acc[rt][ct], 0, 0, 0);                          \
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].y, bv[ct].y, acc[rt][ct], 0, 0, 0);                          \
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].z, bv[ct].z, acc[rt][ct], 0, 0, 0);                          \
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].w, bv[ct].w, acc[rt][ct], 0, 0, 0);                          \
      }                                                                                             \
    }                                                                                               \
  }
  // LDS holds chunk kt (buffer kt & 1); registers hold chunks kt+1 (next to be stored), kt+2, kt+3 (in flight)
  F_LOAD(p, 0)
  F_LOAD(q, 1)
  F_LOAD(r, 2)
  F_STORE(p, 0)
  F_LOAD(p, 3)
  __syncthreads();
  // No conditional around a load or a store: LLVM sinks a load whose only use sits in a later conditional block down to that use
  // (measured on the first build: load -> s_waitcnt vmcnt(0) -> ds_write in every step, 78-110 us per launch instead of ~15).  Past the
  // end the clamped loads re-fetch the last chunk and the extra store lands in the buffer nobody reads again.
#define F_STEP(kt_, R)             \
  F_COMPUTE(buf)                   \
  F_STORE(R, buf ^ 1)              \
  F_LOAD(R, (kt_) + 4)             \
  __syncthreads();                 \
  if ((kt_) + 1 >= nk) break;      \
  buf ^= 1;
  // three steps per trip = the period of the register stages: a stage loaded in step X of one trip is stored in step X of the NEXT
  // trip, i.e. every staged value is loop-carried.  (With six steps per trip — the period of stages x LDS buffers — half of the
  // loads had their use later in the SAME trip, and LLVM's machine sink moved those loads down to their use, across the barriers.)
  // The LDS buffer therefore alternates through a run-time index.
  int buf = 0;
  for (int kt = 0;; kt += 3) {
    F_STEP(kt, q)
    F_STEP(kt + 1, r)
    F_STEP(kt + 2, p)
  }
#undef F_STEP
#undef F_LOAD
#undef F_STORE
#undef F_COMPUTE

  // ------------------------------------------------------------------------------------------ phase 3 prefetch (weights only)
  // W3 chunk j: rows n = j * NC3 .. + NC3 - 1, columns slice * K3 .. + K3 - 1, as [NC3][LD3] in the stage region.
  constexpr int N3 = NC3 * (K3 / 4) / 256;  // float4 per thread per chunk: 16
  constexpr int F4R = K3 / 4;               // float4 per row
  f32x4f w3[N3];
  const int nchunk3 = (d + NC3 - 1) / NC3;
  auto w3_load = [&](int j) {
#pragma unroll
    for (int i = 0; i < N3; ++i) {
      const int e = tid + 256 * i, n = e / F4R, c4 = e % F4R;
      int nn = j * NC3 + n;
      nn = nn < d ? nn : d - 1;
      w3[i] = *reinterpret_cast<const f32x4f*>(g.W3 + (int64_t)nn * g.ld3 + (int64_t)slice * K3 + 4 * c4);
    }
  };
  w3_load(0);  // in flight during the middle phase (weights do not depend on it)

  // ------------------------------------------------------------------------------------------ phase 1 -> Cs
  // accumulator map: acc[rt][ct][r] = C1[16 rt + 4 q4 + r][16 (CT wave + ct) + c16]
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * (CT * wave + ct) + c16;
    float b1 = 0.f;
    if (g.bias1) {
      const int wrow = MODE == FUSED_SA ? (c >> 6) * g.w1_seg_stride + slice * g.w1_slice_rows + (c & 63) : slice * g.w1_slice_rows + c;
      b1 = g.bias1[wrow];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[rt][ct][r] + b1;
        if (MODE == FUSED_FFN) v = fmaxf(v, 0.f);
        Cs[(16 * rt + 4 * q4 + r) * CLD + c] = v;
      }
  }
  const int grp = tid >> 4, c = tid & 15;  // 16 row groups of 16 lanes (one DPP row each): lane c owns columns 4c .. 4c+3 of the head
  if (MODE == FUSED_SA) {
    if (tid < g.Lk) {
      float v = 0.f;
      if (g.rel_bias) v = g.rel_bias[t5_bucket(g.q_pos0 - tid, 0, g.num_buckets, g.lut) * g.H + slice];
      Add[tid] = v;  // causal: every key j <= q_pos0 is allowed
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------------------------------ middle
  if (MODE == FUSED_SA) {
    constexpr int MAXK = 16;
    // cache append: this step's k / v rows of the head (rows of the panel, coalesced float4)
    for (int e = tid; e < rows_here * 32; e += 256) {
      const int r = e >> 5, part = (e >> 4) & 1, cc = e & 15;
      const float4 v = *reinterpret_cast<const float4*>(Cs + r * CLD + 64 * (1 + part) + 4 * cc);
      *reinterpret_cast<float4*>(g.slot + (m0 + r) * g.ld_kv + (part ? g.v_off : g.k_off) + 64 * slice + 4 * cc) = v;
    }
    const int Lk = g.Lk;
    // every load and every use unconditional (clamped key index, masked with selects): see F_STEP.  MK = key capacity of the form
    auto attend = [&](auto mk_tag) {
      constexpr int MK = decltype(mk_tag)::value;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = grp + 16 * rt;
        const bool on = r < rows_here;
        const int rr = on ? r : 0;
        f32x4f kreg[MK], vreg[MK];
        const int32_t* kvr = g.kv_rows + (m0 + rr) * Lk;
#pragma unroll
        for (int j = 0; j < MK; ++j) {
          const int jc = j < Lk - 1 ? j : (Lk > 1 ? Lk - 2 : 0);  // ancestors 0 .. Lk-2; Lk == 1: entry 0 (the row's own slot: valid memory, unused)
          const int64_t row = kvr[jc];
          kreg[j] = *reinterpret_cast<const f32x4f*>(g.kbase + row * g.ld_kv + 64 * slice + 4 * c);
          vreg[j] = *reinterpret_cast<const f32x4f*>(g.vbase + row * g.ld_kv + 64 * slice + 4 * c);
        }
        float4 q = *reinterpret_cast<const float4*>(Cs + rr * CLD + 4 * c);
        q.x *= g.scale, q.y *= g.scale, q.z *= g.scale, q.w *= g.scale;
        const float4 kown = *reinterpret_cast<const float4*>(Cs + rr * CLD + 64 + 4 * c);
        const float4 vown = *reinterpret_cast<const float4*>(Cs + rr * CLD + 128 + 4 * c);
        float sc[MK];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < MK; ++j) {
          const bool own = j == Lk - 1;
          f32x4f kk = kreg[j];
          kk.x = own ? kown.x : kk.x, kk.y = own ? kown.y : kk.y, kk.z = own ? kown.z : kk.z, kk.w = own ? kown.w : kk.w;
          float part = fmaf(q.x, kk.x, fmaf(q.y, kk.y, fmaf(q.z, kk.z, q.w * kk.w)));
          part = row16_sum(part);
          sc[j] = j < Lk ? part + Add[j < Lk ? j : 0] : -INFINITY;
          mx = fmaxf(mx, sc[j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < MK; ++j) {
          sc[j] = expf(sc[j] - mx);  // exp(-inf) = 0 for the slots past Lk
          sum += sc[j];
        }
        const float inv = 1.0f / sum;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < MK; ++j) {
          const bool own = j == Lk - 1;
          f32x4f vv = vreg[j];
          vv.x = own ? vown.x : vv.x, vv.y = own ? vown.y : vv.y, vv.z = own ? vown.z : vv.z, vv.w = own ? vown.w : vv.w;
          const float pj = sc[j] * inv;
          o.x = fmaf(pj, vv.x, o.x), o.y = fmaf(pj, vv.y, o.y), o.z = fmaf(pj, vv.z, o.z), o.w = fmaf(pj, vv.w, o.w);
        }
        *reinterpret_cast<float4*>(Ctx + r * XLD + 4 * c) = on ? o : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    if (Lk <= 4) attend(std::integral_constant<int, 4>{});
    else if (Lk <= 8) attend(std::integral_constant<int, 8>{});
    else attend(std::integral_constant<int, MAXK>{});
    __syncthreads();
  } else if (MODE == FUSED_CA) {
    // rows m0 .. m0 + rows_here - 1 belong to queries b_lo .. b_hi (R consecutive rows each); per query: stage its K / V of this
    // head, then one row group per row: scores over the L keys (lane c: 4 of the 64 columns, summed over the DPP row), softmax, P.V
    const int L = g.L, R = g.R;
    float* const Ks = stage;
    float* const Vs = stage + L * XLD;
    float* const S = Strip + grp * 128;
    const int b_lo = (int)(m0 / R), b_hi = (int)((m0 + rows_here - 1) / R);
    for (int b = b_lo; b <= b_hi; ++b) {
      __syncthreads();  // the previous query's K / V (and, first time, phase 1's operands) are no longer read
      for (int e = tid; e < L * 16; e += 256) {
        const int j = e >> 4, cc = e & 15;
        const int64_t off = ((int64_t)b * L + j) * g.ld_c + 64 * slice + 4 * cc;
        *reinterpret_cast<float4*>(Ks + j * XLD + 4 * cc) = *reinterpret_cast<const float4*>(g.ck + off);
        *reinterpret_cast<float4*>(Vs + j * XLD + 4 * cc) = *reinterpret_cast<const float4*>(g.cv + off);
      }
      for (int j = tid; j < L; j += 256) {
        float v = 0.f;
        if (g.rel_bias) v = g.rel_bias[t5_bucket(g.q_pos0 - j, 1, g.num_buckets, g.lut) * g.H + slice];
        if (g.key_mask && g.key_mask[(int64_t)b * L + j] == 0) v += -1e9f;
        Add[j] = v;
      }
      __syncthreads();
      const int r_lo = (int)((int64_t)b * R > m0 ? (int64_t)b * R - m0 : 0);
      const int r_hi = (int)(((int64_t)(b + 1) * R - m0) < rows_here ? ((int64_t)(b + 1) * R - m0) : rows_here);  // exclusive
      for (int r = r_lo + grp; r < r_hi; r += 16) {
        float4 q = *reinterpret_cast<const float4*>(Cs + r * CLD + 4 * c);
        q.x *= g.scale, q.y *= g.scale, q.z *= g.scale, q.w *= g.scale;
        float mx = -INFINITY;
        for (int j = 0; j < L; ++j) {
          const float4 kk = *reinterpret_cast<const float4*>(Ks + j * XLD + 4 * c);
          float part = fmaf(q.x, kk.x, fmaf(q.y, kk.y, fmaf(q.z, kk.z, q.w * kk.w)));
          part = row16_sum(part) + Add[j];
          mx = fmaxf(mx, part);
          if (c == (j & 15)) S[j] = part;
        }
        // the strip is private to the 16 lanes of this group, which are lanes of ONE wave: its LDS operations complete in order
        __builtin_amdgcn_wave_barrier();
        float sum = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < L; ++j) {
          const float pj = expf(S[j] - mx);
          sum += pj;
          const float4 vv = *reinterpret_cast<const float4*>(Vs + j * XLD + 4 * c);
          o.x = fmaf(pj, vv.x, o.x), o.y = fmaf(pj, vv.y, o.y), o.z = fmaf(pj, vv.z, o.z), o.w = fmaf(pj, vv.w, o.w);
        }
        const float inv = 1.0f / sum;
        *reinterpret_cast<float4*>(Ctx + r * XLD + 4 * c) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
        __builtin_amdgcn_wave_barrier();
      }
    }
    for (int e = tid; e < (RM - rows_here) * 16; e += 256)  // rows past the panel's end: zero operands (their output is not stored)
      *reinterpret_cast<float4*>(Ctx + (rows_here + (e >> 4)) * XLD + 4 * (e & 15)) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
  }

  // ------------------------------------------------------------------------------------------ phase 3
  const float* const A3 = MODE == FUSED_FFN ? Cs : Ctx;
  constexpr int ALD = MODE == FUSED_FFN ? CLD : XLD;
  float* const B3 = stage;
  float* const slab = g.slabs + (int64_t)slice * g.M * d;
  for (int j = 0; j < nchunk3; ++j) {
    if (j > 0) __syncthreads();  // the previous chunk's fragments have been read
#pragma unroll
    for (int i = 0; i < N3; ++i) {
      const int e = tid + 256 * i, n = e / F4R, c4 = e % F4R;
      *reinterpret_cast<f32x4f*>(B3 + n * LD3 + 4 * c4) = w3[i];
    }
    w3_load(j + 1 < nchunk3 ? j + 1 : j);  // unconditional (see F_STEP): the last iteration re-fetches its own chunk
    __syncthreads();
    f32x4f o3[RT][CT3];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT3; ++ct) o3[rt][ct] = (f32x4f){0.f, 0.f, 0.f, 0.f};
    const float* a_ = A3 + c16 * ALD + 4 * q4;
    const float* b_ = B3 + (wave * 16 * CT3 + c16) * LD3 + 4 * q4;
#pragma unroll 4
    for (int kk = 0; kk < K3 / 16; ++kk) {
      float4 av[RT], bv[CT3];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) av[rt] = *reinterpret_cast<const float4*>(a_ + rt * 16 * ALD + 16 * kk);
#pragma unroll
      for (int ct = 0; ct < CT3; ++ct) bv[ct] = *reinterpret_cast<const float4*>(b_ + ct * 16 * LD3 + 16 * kk);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT3; ++ct) {
          o3[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].x, bv[ct].x, o3[rt][ct], 0, 0, 0);
          o3[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].y, bv[ct].y, o3[rt][ct], 0, 0, 0);
          o3[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].z, bv[ct].z, o3[rt][ct], 0, 0, 0);
          o3[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].w, bv[ct].w, o3[rt][ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT3; ++ct) {
      const int n = j * NC3 + 16 * (CT3 * wave + ct) + c16;
      if (n >= d) continue;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rt + 4 * q4 + r;
          if (row < rows_here) slab[(m0 + row) * d + n] = o3[rt][ct][r];
        }
    }
  }
}

// C[m] = epilogue(sum_s slabs[s][m]) in fixed order s = 0 .. S-1, then bias, then residual; Y[m] = norm(C[m]).  One row per
// workgroup, thread <-> float4 column (N <= 1 024), eight slab loads of a row in flight at once; the arithmetic, its order and
// the norms are splitk_reduce_norm_row_kernel's (gemm_small.hip) — only the slab addressing differs ([S][M][N] row-major).
__global__ __launch_bounds__(256) void slab_reduce_norm_row_kernel(const float* __restrict__ slabs, int S, int64_t M, int N,
                                                                   float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                                                   const float* __restrict__ residual, int64_t ldr,
                                                                   const NormEpilogue ne, const int32_t* __restrict__ live) {
  __shared__ float red[4];
  if (live && *live == 0) return;
  const int64_t m = blockIdx.x;
  const int tid = threadIdx.x, n4 = N >> 2, wave = tid >> 6;
  const bool on = tid < n4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on) {
    const int n = tid << 2;
    const float* p = slabs + m * N + n;
    const int64_t ss = M * (int64_t)N;
    float4 b = v, q = v;
    if (bias) b = *reinterpret_cast<const float4*>(bias + n);
    if (residual) q = *reinterpret_cast<const float4*>(residual + m * ldr + n);
    v = *reinterpret_cast<const float4*>(p);
    int s = 1;
    for (; s + 8 <= S; s += 8) {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(p + (int64_t)(s + u) * ss);
#pragma unroll
      for (int u = 0; u < 8; ++u) v.x += t[u].x, v.y += t[u].y, v.z += t[u].z, v.w += t[u].w;
    }
    {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s + u < S) t[u] = *reinterpret_cast<const float4*>(p + (int64_t)(s + u) * ss);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s + u < S) v.x += t[u].x, v.y += t[u].y, v.z += t[u].z, v.w += t[u].w;
    }
    if (bias) v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
    if (residual) v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
    *reinterpret_cast<float4*>(C + m * ldc + n) = v;
  }
  auto block_sum = [&](float x) {
    x = wave_sum(x);
    __syncthreads();
    if ((tid & 63) == 0) red[wave] = x;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
  };
  float* yr = ne.Y + m * ne.ldy;
  if (ne.kind == 1) {
    const float ss = block_sum(on ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : 0.f);
    const RowDivisor over(sqrtf(ss / (float)N + ne.eps));
    if (on) {
      const float4 gw = reinterpret_cast<const float4*>(ne.w1)[tid];
      *reinterpret_cast<float4*>(yr + 4 * tid) = make_float4(gw.x * over(v.x), gw.y * over(v.y), gw.z * over(v.z), gw.w * over(v.w));
    }
    return;
  }
  const float inv_d = 1.0f / (float)N;
  auto layer_norm = [&](const float* w, const float* b) {
    const float mean = block_sum(on ? v.x + v.y + v.z + v.w : 0.f) * inv_d;
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    const float rstd = 1.0f / sqrtf(block_sum(on ? a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 : 0.f) * inv_d + ne.eps);
    if (on) {
      const float4 gw = reinterpret_cast<const float4*>(w)[tid], bb = reinterpret_cast<const float4*>(b)[tid];
      v = make_float4(a0 * rstd * gw.x + bb.x, a1 * rstd * gw.y + bb.y, a2 * rstd * gw.z + bb.z, a3 * rstd * gw.w + bb.w);
    }
  };
  layer_norm(ne.w1, ne.b1);
  if (ne.kind == 3) {
    if (on) {
      const float4 t = reinterpret_cast<const float4*>(ne.addv)[tid];
      v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
    }
    layer_norm(ne.w2, ne.b2);
  }
  if (on) *reinterpret_cast<float4*>(yr + 4 * tid) = v;
}

template <int MODE, int RT, int N1>
static int launch_one(const FusedArgs& g, hipStream_t stream) {
  using Cfg = FusedCfg<MODE, RT, N1>;
  const size_t lds = sizeof(float) * (size_t)Cfg::lds_floats(MODE == FUSED_CA ? g.L : 0);
  if (lds > 160 * 1024) return 1;
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(decode_fused_kernel<MODE, RT, N1>), 160 * 1024, "decode_fused")) return rc__;
  const int64_t blocks = (int64_t)g.n_panels * g.n_slices;
  hipLaunchKernelGGL((decode_fused_kernel<MODE, RT, N1>), dim3((unsigned)blocks), dim3(256), lds, stream, g);
  GDR_CHECK_LAUNCH("decode_fused_kernel");
  return 0;
}

// Which rows run the fused sub-blocks: dk = 64 heads, a width the slab reduction holds in one workgroup, few enough rows that the
// unfused grids cannot fill the chip.  Returns the panel height in 16-row tiles (1 or 2), 0 = not served.
int decode_fused_rt(int64_t M, int d, int inner, int dk) {
  if (M < 1 || M > 1024 || dk != 64 || d % 64 != 0 || d > 1024 || inner % 64 != 0) return 0;
  return M <= 256 ? 1 : 2;
}
size_t decode_fused_slab_bytes(int64_t M, int d, int d_ff, int H) {
  const int s_ffn = d_ff / 128;  // the finest chunking used
  const int S = s_ffn > H ? s_ffn : H;
  return (size_t)S * (size_t)M * (size_t)d * sizeof(float);
}

// The launchers: phase kernels + the slab reduction with its epilogue and norm.  Return 1 when the shape is not served.
int launch_decode_fused(int mode, FusedArgs g, int rt, int n1, hipStream_t stream) {
  g.n_panels = (int)((g.M + 16 * rt - 1) / (16 * rt));
#define GO(MODE_, RT_, N1_) \
  if (mode == MODE_ && rt == RT_ && n1 == N1_) return launch_one<MODE_, RT_, N1_>(g, stream);
  GO(FUSED_SA, 1, 192) GO(FUSED_SA, 2, 192) GO(FUSED_CA, 1, 64) GO(FUSED_CA, 2, 64)
  GO(FUSED_FFN, 1, 128) GO(FUSED_FFN, 2, 128) GO(FUSED_FFN, 1, 256) GO(FUSED_FFN, 2, 256)
#undef GO
  return 1;
}

int launch_slab_reduce_norm(const float* slabs, int S, int64_t M, int N, float* C, int64_t ldc, const float* bias, const float* residual,
                            int64_t ldr, const NormEpilogue& ne, const int32_t* live, hipStream_t stream) {
  if (N % 4 != 0 || N > 1024 || S < 1) return 1;
  ProfScope prof_r(PROF_REDUCE, 0.0, stream);
  hipLaunchKernelGGL(slab_reduce_norm_row_kernel, dim3((unsigned)M), dim3(256), 0, stream, slabs, S, M, N, C, ldc, bias, residual, ldr, ne,
                     live);
  GDR_CHECK_LAUNCH("slab_reduce_norm_row_kernel");
  return 0;
}

}  // namespace gdr
