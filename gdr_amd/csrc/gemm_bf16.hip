// bf16 "NT" linear for gfx950 (C5 precision mode):  C[M,N] fp32 = epilogue(A[M,K] bf16 · W[N,K]^T bf16), fp32 accumulate.
//
// At bf16 MFMA rates a 128x128 tile consumes 32 KB of operands per 512 MFMA cycles, so the operand path is the kernel:
//   * staging is LDS-DMA (`global_load_lds_dwordx4`): no VGPR round trip, no ds_write pass; one wave-instruction
//     deposits 8 rows x 128 B; the LDS image is lane-linear, so the bank swizzle (16-byte chunk ^ ((row>>1)&7),
//     conflict-free for the 16 consecutive rows of a ds_read_b128 lane group) is applied to the per-lane SOURCE address
//     and again on the fragment reads;
//   * one 32 KB LDS buffer, two barriers per K-step (BK = 64), ~110 VGPRs: three to four workgroups per CU overlap one
//     another's stage / compute phases instead of an in-kernel software pipeline;
//   * v_mfma_f32_16x16x32_bf16, 4 waves 2x2, 64x64 per wave = 4x4 accumulator tiles; the W block takes the MFMA "A" role
//     so that a lane holds 4 consecutive output columns of one row (16-byte epilogue accesses).
#include "common.h"

namespace gdr {

typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8b __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8b __attribute__((ext_vector_type(8)));

struct Bf16GemmArgs {
  const char* A;  // bf16 [M, lda]
  const char* W;  // bf16 [N, ldw]
  float* C;       // fp32 [M, ldc], or bf16 [M, ldc] when out_bf16 (the output only feeds another bf16 linear)
  const float* bias;
  const float* residual;
  int64_t lda, ldw, ldc, ldr;  // elements
  int64_t M;      // rows of A
  int64_t Nrows;  // rows of W (= N; 64-bit: the corpus in the similarity form)
  int N, K, tiles_n;
  int has_bias, has_residual, act;  // act: 0 none, 1 relu, 2 gelu
  int out_bf16;
  const int64_t* m_dev;  // linear only, may be null: live row count on the device (<= M); tiles past it exit at once
  // split form (r06, "fp32 carried as three bf16 planes"): split_k0 = K0 != 0 -> a row of A / W holds the planes [hi | mid | lo], K0
  // elements each (lda, ldw >= 3 K0), and the contraction runs over K = 6 K0 virtual elements: block p of K0 pairs plane colA(p) of A
  // with plane colW(p) of W — hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid, the six leading terms of (hi+mid+lo).(hi+mid+lo)
  int split_k0;
  // EPI 3 (the decode head, see launch_linear_bf16_headdot): nothing is stored; per output row m and vocabulary entry c = n / dot_d
  // the tile leaves  sum_i (h[dot_rows[m]][i] * dot_scale) * (acc[m][c*dot_d + i] + dot_e[c][i])  over its 64 columns per wave in
  // dot_out[(m * (N / dot_d) + c) * (dot_d / 64) + (n % dot_d) / 64]
  const float* dot_h;
  int64_t dot_ldh;
  const int32_t* dot_rows;
  const float* dot_e;
  float* dot_out;
  float dot_scale;
  int dot_d;
  // similarity epilogues (EPI 1 sample / 2 filter): A = queries [B,d] (lane role), W = docs [N,d] (register role);
  // tiles_n then counts QUERY tiles (fastest in the grid: the workgroups that share a doc tile are neighbours)
  SimEpilogue sim;
};

__device__ __forceinline__ float gelu_erf_b(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// SPLIT (template parameter of the linear kernels, so that the plain forms keep their K loops untouched):
//   0  plain bf16 rows.
//   1  bf16 planes: a row holds [hi | mid | lo] of k0 elements each; block p of k0 virtual elements pairs plane {0,0,1,0,2,1}[p] of A with
//      plane {0,1,0,2,0,1}[p] of W (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; the first 3 blocks alone = the 16-bit form).
//   2  fp16 x 2: a row holds [hi | lo'] with hi = fp16(x), lo' = fp16((x - hi) * 2^11) — 22 significand bits, lo' in hi's range (no
//      subnormal loss) — on v_mfma_f32_16x16x32_f16; blocks (hi, lo'), (lo', hi) first, the accumulators are then scaled by 2^-11
//      (kt == 2 k0 / 64), then (hi, hi): c = hi.hi + 2^-11 (hi.lo' + lo'.hi), the dropped lo'.lo' term is 2^-22 of the product.
template <int SPLIT>
__device__ __forceinline__ int plane_of(int p, bool is_w) {
  if (SPLIT == 1) return ((is_w ? 0x102010 : 0x120100) >> (4 * p)) & 3;
  return ((is_w ? 0x001 : 0x010) >> (4 * p)) & 3;
}
// Byte offsets of virtual K-tile kt (64 elements) inside a row of A resp. W.  kt is wave-uniform: scalar arithmetic; inv = ceil(2^16 / (k0 / 64)).
template <int SPLIT>
__device__ __forceinline__ void ktile_offsets(int kt, int split_k0, int inv, int& off_a, int& off_w) {
  if (SPLIT == 0) {
    off_a = off_w = kt * 128;
    return;
  }
  const int nk0 = split_k0 >> 6;
  const int p = (kt * inv) >> 16, k0 = kt - p * nk0;  // exact for kt < 6 * nk0 <= 6 * 1024
  off_a = (plane_of<SPLIT>(p, false) * split_k0 + k0 * 64) * 2;
  off_w = (plane_of<SPLIT>(p, true) * split_k0 + k0 * 64) * 2;
}
template <int SPLIT>
__device__ __forceinline__ f32x4b mfma16(const float4 fb, const float4 fa, const f32x4b acc) {
  if constexpr (SPLIT == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8b, fb), __builtin_bit_cast(f16x8b, fa), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, fb), __builtin_bit_cast(bf16x8b, fa), acc, 0, 0, 0);
}
// v -> the 4 + 4 (+ 4) plane values of one output quad, stored at pl (plane stride `ps` elements): form 2 = bf16 x 3, form 3 = fp16 x 2
__device__ __forceinline__ void store_planes(void* base, int64_t elem_off, int64_t ps, const float (&v)[4], int form) {
  if (form == 3) {
    union {
      _Float16 h[4];
      uint2 u;
    } hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi.h[j] = (_Float16)v[j];
      lo.h[j] = (_Float16)((v[j] - (float)hi.h[j]) * 2048.0f);
    }
    _Float16* pl = static_cast<_Float16*>(base) + elem_off;
    *reinterpret_cast<uint2*>(pl) = hi.u;
    *reinterpret_cast<uint2*>(pl + ps) = lo.u;
    return;
  }
  union {
    __bf16 h[4];
    uint2 u;
  } hi, mid, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi.h[j] = (__bf16)v[j];
    const float r1 = v[j] - (float)hi.h[j];
    mid.h[j] = (__bf16)r1;
    lo.h[j] = (__bf16)(r1 - (float)mid.h[j]);
  }
  __bf16* pl = static_cast<__bf16*>(base) + elem_off;
  *reinterpret_cast<uint2*>(pl) = hi.u;
  *reinterpret_cast<uint2*>(pl + ps) = mid.u;
  *reinterpret_cast<uint2*>(pl + 2 * ps) = lo.u;
}

// BM = rows of A per tile: 128, or 64 for the linears of a few thousand rows (the decode legs of config C5: 1 920 beam rows x
// N = 768 are 90 tiles of 128 x 128 on 256 CUs; 64-row tiles double the workgroups — the W operand is re-read twice as often,
// which L2 absorbs at these sizes — and halve the accumulators, 32 x 64 per wave).  The similarity forms use 128.
template <int EPI, int BM = 128, int SPLIT = 0>  // EPI: 0 linear, 1 similarity sample, 2 similarity filter, 3 linear whose output is only dotted with h
__global__ __launch_bounds__(256, 3) void gemm_nt_bf16_glds_kernel(const Bf16GemmArgs g) {
  constexpr bool LIN = EPI == 0 || EPI == 3;
  static_assert(BM == 128 || (BM == 64 && LIN), "64-row tiles serve the linear forms only");
  constexpr int MI = BM / 32;   // 16-row accumulator blocks per wave along M: 4 or 2
  constexpr int AI = BM / 32;   // A staging instructions per wave (8 rows each): 4 or 2
  __shared__ __attribute__((aligned(1024))) char smem[BM * 128 + 128 * 128];  // As [BM rows][128 B], Bs [128 rows][128 B]
  char* const As = smem;
  char* const Bs = smem + BM * 128;
  unsigned bid = blockIdx.x;
  const int64_t Mv = (LIN && g.m_dev) ? *g.m_dev : g.M;
  {
    // The XCD-aware remap hands every XCD a CONTIGUOUS range of logical tiles.  With a device-side row count the grid is sized
    // for the padded batch and the tiles past the live rows are the LAST logical ones: remapped over the whole grid they all
    // belong to the last XCDs, which then idle while the first ones do all the work (C2's packed batch, 60 % live: XCDs 5-7 got
    // nothing; the bf16 linears ran 25-30 % slower in the encoder than alone).  So the remap runs over the LIVE tile count and
    // hardware blocks past it exit — the hardware deals consecutive block ids round-robin over the XCDs, so the live ones
    // are spread evenly.
    unsigned nblk = gridDim.x;
    if (LIN && g.m_dev) {
      const int64_t live = ((Mv + BM - 1) / BM) * (int64_t)g.tiles_n;
      if (live < (int64_t)nblk) {
        if ((int64_t)bid >= live) return;  // uniform
        nblk = (unsigned)live;
      }
    }
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  int64_t m0, slot_base = 0;
  int64_t n0;
  if (LIN) {
    if (g.tiles_n >= 32) {
      // wide outputs (the decode head: N = 23 808 = 186 column tiles, 36.6 MB of W): in row-panel-major order every row panel streams
      // ALL of W past an XCD's 4 MiB L2 — 120 panels x 36.6 MB = 4.4 GB per launch at 15 360 rows, the launch ran at 7 TB/s of
      // HBM-side fetches (profiles/r05_generate_bf16_512x30_pmc_by_kernel.txt).  Supertiles of 8 row panels x all column tiles,
      // row-panel-fastest inside: a W tile is fetched once per 8 panels, the 8 A panels (1.5 MB) stay resident.
      constexpr unsigned GM = 8;
      const unsigned tiles_m = (unsigned)((Mv + BM - 1) / BM);
      const unsigned per = GM * (unsigned)g.tiles_n, grp = bid / per, loc = bid - grp * per;
      const unsigned gm = min(GM, tiles_m - grp * GM);
      const unsigned nt = loc / gm;
      m0 = (int64_t)(grp * GM + (loc - nt * gm)) * BM;
      n0 = (int64_t)nt * 128;
    } else {
      m0 = (int64_t)(bid / (unsigned)g.tiles_n) * BM;
      n0 = (int64_t)(bid % (unsigned)g.tiles_n) * 128;
    }
  } else {
    int64_t dt = bid / (unsigned)g.tiles_n;  // doc tile of this pass
    m0 = (int64_t)(bid % (unsigned)g.tiles_n) * 128;  // query tile
    if (EPI == 1) {
      slot_base = dt * 128;
      dt = dt * g.sim.tile_stride;
    } else {
      const int s1 = g.sim.tile_stride - 1;
      dt = (dt / s1) * g.sim.tile_stride + 1 + (dt % s1);
    }
    n0 = dt * 128;
  }
  if (LIN && m0 >= Mv) return;  // uniform (cannot happen after the live-count remap; kept as the bound it documents)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, q4 = lane >> 4;

  // ---- staging map: instruction i of this wave covers tile rows (wave*4 + i)*8 .. +7; lane = (row_in, chunk') ----
  const int srow = lane >> 3, schunk = lane & 7;
  const char* a_src[AI];
  const char* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + srow;
    const int chunk = schunk ^ ((row >> 1) & 7);
    int64_t rw = n0 + row;
    rw = rw < g.Nrows ? rw : g.Nrows - 1;
    w_src[i] = g.W + (rw * g.ldw) * 2 + chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (wave * AI + i) * 8 + srow;
    const int chunk = schunk ^ ((row >> 1) & 7);
    int64_t ra = m0 + row;
    ra = ra < Mv ? ra : Mv - 1;  // rows past the edge are computed and discarded
    a_src[i] = g.A + (ra * g.lda) * 2 + chunk * 16;
  }
  // ---- fragment reads: lane (r16, q4) reads 16 B = k 8*q4..+7 of a 32-k half; 4 row blocks of A, 4 of W ----
  int a_off[MI], b_off[4], a_sw[MI], b_sw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb = wn * 64 + i * 16 + r16;
    b_off[i] = rb * 128, b_sw[i] = (rb >> 1) & 7;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ra = wm * (BM / 2) + i * 16 + r16;
    a_off[i] = ra * 128, a_sw[i] = (ra >> 1) & 7;
  }

  f32x4b acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4b){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K >> 6;
  const int split_inv = SPLIT ? (65536 + (g.split_k0 >> 6) - 1) / (g.split_k0 >> 6) : 0;
  for (int kt = 0; kt < nk; ++kt) {
    int koff, koff_w;
    ktile_offsets<SPLIT>(kt, g.split_k0, split_inv, koff, koff_w);
    if (SPLIT == 2 && kt == 2 * (g.split_k0 >> 6)) {  // the two cross blocks are in: scale them by 2^-11 before the hi.hi block
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= 0.00048828125f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < AI)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                         (__attribute__((address_space(3))) void*)(As + (wave * AI + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + koff_w),
                                       (__attribute__((address_space(3))) void*)(Bs + (wave * 4 + i) * 1024), 16, 0, 0);
    }
    __syncthreads();  // emits vmcnt(0): the DMA writes are complete and visible
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float4 fa[MI], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < MI) fa[i] = *reinterpret_cast<const float4*>(As + a_off[i] + (((kk * 4 + q4) ^ a_sw[i]) << 4));
        fb[i] = *reinterpret_cast<const float4*>(Bs + b_off[i] + (((kk * 4 + q4) ^ b_sw[i]) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = mfma16<SPLIT>(fb[ni], fa[mi], acc[mi][ni]);
    }
    __syncthreads();  // every fragment read done before the next K-step's DMA overwrites the buffer
  }

  // ---- similarity epilogues: lane&15 <-> query, registers <-> 4 consecutive docs ----
  if (EPI == 1) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t q = m0 + wm * 64 + mi * 16 + r16;
      if (q >= Mv) continue;
      float* cv = g.sim.cand_val + q * g.sim.cap + slot_base;
      int32_t* ci = g.sim.cand_idx + q * g.sim.cap + slot_base;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int off = wn * 64 + ni * 16 + 4 * q4;
        const int64_t n = n0 + off;
        float4 v;
        int4 id;
        v.x = n + 0 < g.Nrows ? acc[mi][ni][0] : -INFINITY, v.y = n + 1 < g.Nrows ? acc[mi][ni][1] : -INFINITY;
        v.z = n + 2 < g.Nrows ? acc[mi][ni][2] : -INFINITY, v.w = n + 3 < g.Nrows ? acc[mi][ni][3] : -INFINITY;
        id.x = n + 0 < g.Nrows ? (int)n : -1, id.y = n + 1 < g.Nrows ? (int)n + 1 : -1;
        id.z = n + 2 < g.Nrows ? (int)n + 2 : -1, id.w = n + 3 < g.Nrows ? (int)n + 3 : -1;  // -1 = padding slot
        *reinterpret_cast<float4*>(cv + off) = v;
        *reinterpret_cast<int4*>(ci + off) = id;
      }
    }
    return;
  }
  if (EPI == 2) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t q = m0 + wm * 64 + mi * 16 + r16;
      const bool q_ok = q < Mv;
      const float thr = q_ok ? g.sim.thr[q] : INFINITY;
      unsigned keep = 0u;  // bit 4*ni + r
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t n = n0 + wn * 64 + ni * 16 + 4 * q4 + r;
          keep |= (q_ok && acc[mi][ni][r] >= thr && n < g.Nrows) ? (1u << (4 * ni + r)) : 0u;
        }
      // the 4 lanes that share this query (lane>>4 = 0..3) pool their counts: one returning atomic per query and tile
      const int mine = __popc(keep);
      const int c0 = __shfl(mine, r16), c1 = __shfl(mine, r16 + 16), c2 = __shfl(mine, r16 + 32), c3 = __shfl(mine, r16 + 48);
      const int total = c0 + c1 + c2 + c3;
      int base = 0;
      if (q4 == 0 && total > 0) base = atomicAdd(g.sim.cand_cnt + (int64_t)q * CNT_STRIDE, total);
      base = __shfl(base, r16);
      int pos = base + (q4 > 0 ? c0 : 0) + (q4 > 1 ? c1 : 0) + (q4 > 2 ? c2 : 0);
      if (mine) {
        float* cv = g.sim.cand_val + q * g.sim.cap;
        int32_t* ci = g.sim.cand_idx + q * g.sim.cap;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (keep >> (4 * ni + r) & 1u) {
              if (pos < g.sim.cap) {
                cv[pos] = acc[mi][ni][r];
                ci[pos] = (int)(n0 + wn * 64 + ni * 16 + 4 * q4 + r);
              }
              ++pos;
            }
      }
    }
    return;
  }
  if (EPI == 3) {
    // ---- head dot: the lane's 4 x 4 columns of each of its rows, then the four lanes (lane >> 4) that share the row ----
    const int c = (int)(n0 / g.dot_d), i0 = (int)(n0 % g.dot_d) + wn * 64;  // a tile lies inside one vocabulary entry (dot_d % 128 == 0)
    const float* er = g.dot_e + (int64_t)c * g.dot_d + i0 + 4 * q4;
    const int slots = g.dot_d >> 6;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int64_t m = m0 + wm * (BM / 2) + mi * 16 + r16;
      const int64_t mc = m < Mv ? m : Mv - 1;
      const float* hr = g.dot_h + (int64_t)g.dot_rows[mc] * g.dot_ldh + i0 + 4 * q4;
      float part = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const float4 hv = *reinterpret_cast<const float4*>(hr + ni * 16);
        const float4 ev = *reinterpret_cast<const float4*>(er + ni * 16);
        part = fmaf(hv.x * g.dot_scale, acc[mi][ni][0] + ev.x, part);
        part = fmaf(hv.y * g.dot_scale, acc[mi][ni][1] + ev.y, part);
        part = fmaf(hv.z * g.dot_scale, acc[mi][ni][2] + ev.z, part);
        part = fmaf(hv.w * g.dot_scale, acc[mi][ni][3] + ev.w, part);
      }
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      if (q4 == 0 && m < Mv) g.dot_out[(m * (int64_t)(g.N / g.dot_d) + c) * slots + (i0 >> 6)] = part;
    }
    return;
  }
  // ---- epilogue: row m = block*16 + lane&15, columns n = block*16 + 4*(lane>>4) + 0..3 ----
  // 16-byte accesses need whole column quads inside N and aligned rows; rows past the live count (the partial last row
  // panel of a packed batch) are simply skipped — a per-ROW test, so the valid rows of an edge panel keep the vector path
  const bool cols_vec = n0 + 128 <= g.N && (g.ldc & 3) == 0 && (!g.has_residual || (g.ldr & 3) == 0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int64_t m = m0 + wm * (BM / 2) + mi * 16 + r16;
    if (m >= Mv) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = (int)n0 + wn * 64 + ni * 16 + 4 * q4;
      float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (cols_vec) {
        if (g.has_bias) {
          const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
          v[0] += b.x, v[1] += b.y, v[2] += b.z, v[3] += b.w;
        }
        if (g.has_residual) {
          const float4 r = *reinterpret_cast<const float4*>(g.residual + m * g.ldr + n);
          v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (g.act == 1) v[j] = fmaxf(v[j], 0.f);
          if (g.act == 2) v[j] = gelu_erf_b(v[j]);
        }
        if (g.out_bf16) {
          union {
            __bf16 h[4];
            uint2 u;
          } o;
          o.h[0] = (__bf16)v[0], o.h[1] = (__bf16)v[1], o.h[2] = (__bf16)v[2], o.h[3] = (__bf16)v[3];
          *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(g.C) + m * g.ldc + n) = o.u;
        } else {
          *reinterpret_cast<float4*>(g.C + m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (m >= Mv || n + j >= g.N) continue;
          float x = v[j];
          if (g.has_bias) x += g.bias[n + j];
          if (g.has_residual) x += g.residual[m * g.ldr + n + j];
          if (g.act == 1) x = fmaxf(x, 0.f);
          if (g.act == 2) x = gelu_erf_b(x);
          if (g.out_bf16)
            reinterpret_cast<__bf16*>(g.C)[m * g.ldc + n + j] = (__bf16)x;
          else
            g.C[m * g.ldc + n + j] = x;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tiles, 8 waves (2 x 4), BK = 64, one workgroup per CU, 128 KiB of LDS = 2 buffers x {A_lo, A_hi, B_lo, B_hi} half-tiles of
// 128 rows x 128 B — the "256^2 8-phase" schedule of cdna_hip_programming.md: four phases per K-tile, each [fragment reads of the
// phase + one half-tile of LDS-DMA prefetch] barrier [16 MFMAs on one 64 x 32 quadrant of the wave's 128 x 64 block] barrier; the two
// wave rows run one barrier apart so that on every SIMD one wave is in its MFMA cluster while the other reads / stages; the DMA stays
// in flight across barriers behind ONE counted vmcnt(6) per K-tile.  Twice the flops per operand byte of the 128^2 kernel above: its K
// loop runs at ~1 PFLOP/s where that one's runs at ~0.7 — but with one workgroup per CU every CU reaches its epilogue alone, so a launch
// only wins where the K loop is long or the output small (round-3 lab, profiles/r03_bf16_lab_tile256.txt; adopted per shape in round 4:
// K >= 2048 — the FFN's second linear: 88.9 -> 78.3 us at 12 308 rows, 125.9 -> 93.1 us = 1.04 PFLOP/s at 20 480,
// profiles/r04_bf16_tile_height_sweep.txt).  Same k order per output element as the 128^2 kernel: bit-identical results (tests/test_gpu_parity.py).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // the LDS-DMA's destination travels in m0: named in the clobber list on purpose
#define GLDS16(gptr_, lds_) \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_), "v"(gptr_) : "memory", "m0")

constexpr int HALF_BYTES = 128 * 128;                 // one half-tile
constexpr int OFF_A_LO = 0, OFF_A_HI = HALF_BYTES, OFF_B_LO = 2 * HALF_BYTES, OFF_B_HI = 3 * HALF_BYTES, BUF_BYTES = 4 * HALF_BYTES;

// BN = 256, or 192 (each wave column 48 wide: a "lo" half of two 16-column blocks and a "hi" half of one): the tile WIDTH is chosen
// per launch so that the tile count quantises well on 256 CUs — 15 360 beam rows x N = 768 are 180 tiles of 256 x 256 (70 % of one
// round) but 240 of 256 x 192 (94 %); N = 2 304: 540 tiles = 3 rounds at 70 % against 720 = 3 rounds at 94 %.  Same k order.
template <int BN, int SPLIT = 0>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_tile256_kernel(const Bf16GemmArgs g) {
  static_assert(BN == 256 || BN == 192, "tile width");
  constexpr int WCOLS = BN / 4;           // columns per wave column: 64 or 48
  constexpr int NBH = BN == 256 ? 2 : 1;  // 16-column blocks in the hi half (the lo half always holds 2)
  constexpr int NB = 2 + NBH;
  constexpr int HI_ROWS = 16 * NBH;       // B rows per wave column in the hi half-tile
  extern __shared__ __attribute__((aligned(1024))) char smem256[];
  unsigned bid = blockIdx.x;
  const int64_t Mv = g.m_dev ? *g.m_dev : g.M;
  const int tiles_n = (g.N + BN - 1) / BN;
  {
    // XCD-aware remap over the LIVE tile count (see gemm_nt_bf16_glds_kernel): hardware blocks past it exit
    unsigned nblk = gridDim.x;
    if (g.m_dev) {
      const int64_t live = ((Mv + 255) / 256) * (int64_t)tiles_n;
      if (live < (int64_t)nblk) {
        if ((int64_t)bid >= live) return;  // uniform
        nblk = (unsigned)live;
      }
    }
    const unsigned q_ = nblk >> 3, r_ = nblk & 7u, xcd_ = bid & 7u, j_ = bid >> 3;
    bid = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + j_;
  }
  const int64_t m0 = (int64_t)(bid / (unsigned)tiles_n) * 256;
  if (m0 >= Mv) return;  // uniform
  const int n0 = (int)(bid % (unsigned)tiles_n) * BN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, q4 = lane >> 4;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem256));
  // ---- staging: instruction i of wave w covers local rows (w*2 + i)*8 + (lane>>3) of a half-tile (A, B lo: 128 rows = 2 instructions
  //      per wave; B hi at BN = 192: 64 rows = 1)
  const int srow = lane >> 3, schunk = lane & 7;
  const char* a_src[2][2];  // [half][instr]
  const char* b_lo_src[2];
  const char* b_hi_src[NBH];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave * 2 + i) * 8 + srow;
      const int chunk = schunk ^ ((lr >> 1) & 7);
      int64_t ra = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
      ra = ra < Mv ? ra : Mv - 1;
      a_src[h][i] = g.A + (ra * g.lda) * 2 + chunk * 16;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = (wave * 2 + i) * 8 + srow;
    const int chunk = schunk ^ ((lr >> 1) & 7);
    int rb = n0 + (lr >> 5) * WCOLS + (lr & 31);
    rb = rb < g.N ? rb : g.N - 1;
    b_lo_src[i] = g.W + ((int64_t)rb * g.ldw) * 2 + chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < NBH; ++i) {
    const int lr = NBH == 2 ? (wave * 2 + i) * 8 + srow : wave * 8 + srow;
    const int chunk = schunk ^ ((lr >> 1) & 7);
    int rb = n0 + (lr / HI_ROWS) * WCOLS + 32 + (lr % HI_ROWS);
    rb = rb < g.N ? rb : g.N - 1;
    b_hi_src[i] = g.W + ((int64_t)rb * g.ldw) * 2 + chunk * 16;
  }
  const unsigned st_dst = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 2048;  // + i*1024 + half offset + buffer offset
  const unsigned st_dst1 = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024;  // the one-instruction half-tile (B hi, BN = 192)
  // koff_: the K-tile's byte offset inside a row (plain form kt * 128; split form: KTile below — A's and W's plane schedules differ)
#define STAGE(src_, half_off_, buf_, koff_)                                                 \
  {                                                                                         \
    GLDS16(src_[0] + (koff_), st_dst + (buf_)*BUF_BYTES + (half_off_));                     \
    GLDS16(src_[1] + (koff_), st_dst + (buf_)*BUF_BYTES + (half_off_) + 1024);              \
  }
#define STAGE_BHI(buf_, koff_)                                                              \
  {                                                                                         \
    if (NBH == 2) STAGE(b_hi_src, OFF_B_HI, buf_, koff_)                                    \
    else GLDS16(b_hi_src[0] + (koff_), st_dst1 + (buf_)*BUF_BYTES + OFF_B_HI);              \
  }
  // ---- fragment reads
  const int sw = (r16 >> 1) & 7;
  const int c0 = (q4 ^ sw) << 4;
  const char* const fa_base = smem256 + (wr * 64 + r16) * 128 + c0;
  const char* const fbl_base = smem256 + (wc * 32 + r16) * 128 + c0;
  const char* const fbh_base = smem256 + (wc * HI_ROWS + r16) * 128 + c0;
  float4 fa[4][2], fbl[2][2], fbh[NBH][2];
#define READ_A(buf_, half_off_)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
    fa[i][0] = *reinterpret_cast<const float4*>(fa_base + (buf_)*BUF_BYTES + (half_off_) + i * 2048);             \
    fa[i][1] = *reinterpret_cast<const float4*>((fa_base + (buf_)*BUF_BYTES + (half_off_) + i * 2048) + 64 - 2 * (c0 & 64)); \
  }
#define READ_BL(buf_)                                                                                             \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
    fbl[j][0] = *reinterpret_cast<const float4*>(fbl_base + (buf_)*BUF_BYTES + OFF_B_LO + j * 2048);              \
    fbl[j][1] = *reinterpret_cast<const float4*>((fbl_base + (buf_)*BUF_BYTES + OFF_B_LO + j * 2048) + 64 - 2 * (c0 & 64)); \
  }
#define READ_BH(buf_)                                                                                             \
  _Pragma("unroll") for (int j = 0; j < NBH; ++j) {                                                               \
    fbh[j][0] = *reinterpret_cast<const float4*>(fbh_base + (buf_)*BUF_BYTES + OFF_B_HI + j * 2048);              \
    fbh[j][1] = *reinterpret_cast<const float4*>((fbh_base + (buf_)*BUF_BYTES + OFF_B_HI + j * 2048) + 64 - 2 * (c0 & 64)); \
  }
  f32x4b acc[8][NB];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4b){0.f, 0.f, 0.f, 0.f};
#define MFMA_Q(mh_, nb0_, nbn_, fb_)                                                                               \
  {                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < (nbn_); ++j)               \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                           \
            acc[(mh_)*4 + i][(nb0_) + j] = mfma16<SPLIT>(fb_[j][kk], fa[i][kk], acc[(mh_)*4 + i][(nb0_) + j]);    \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  }
#define BAR()                              \
  {                                        \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
  }
  // the counted wait of phases 4 / 8 leaves the three youngest half-tiles (B lo, A lo, B hi of the K-tile after next) in flight
#define WAIT_3HALVES() \
  { if (NBH == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
  const int nk = g.K >> 6;  // even (launcher)
  // The K-tile walker: byte offsets (A row, W row) of K-tile `t`, advanced one tile at a time and parked on the last tile (the loop's
  // clamped prefetches).  Plain form: t * 128 for both.  Split form: block p = t / nk0 pairs plane colA(p) of A with colW(p) of W; kept as
  // running scalars (p, k0) — a division per staged half-tile cost the 8-phase loop 15 % (profiles/r06_split_bf16.txt).
  struct KTile {
    int t, p, k0, a, w;
  } kw{0, 0, 0, 0, 0};
  const int nk0 = SPLIT ? g.split_k0 >> 6 : nk;
  auto k_next = [&]() {
    if (kw.t + 1 >= nk) return;
    ++kw.t;
    if (!SPLIT) {
      kw.a = kw.w = kw.t * 128;
      return;
    }
    if (++kw.k0 == nk0) kw.k0 = 0, ++kw.p;
    kw.a = (plane_of<SPLIT>(kw.p, false) * g.split_k0 + kw.k0 * 64) * 2;
    kw.w = (plane_of<SPLIT>(kw.p, true) * g.split_k0 + kw.k0 * 64) * 2;
  };
  if (SPLIT) kw.a = plane_of<SPLIT>(0, false) * g.split_k0 * 2, kw.w = plane_of<SPLIT>(0, true) * g.split_k0 * 2;  // K-tile 0 of block 0
  // prologue: K-tile 0 whole, K-tile 1 without A_hi
  STAGE(b_lo_src, OFF_B_LO, 0, kw.w)
  STAGE(a_src[0], OFF_A_LO, 0, kw.a)
  STAGE_BHI(0, kw.w)
  STAGE(a_src[1], OFF_A_HI, 0, kw.a)
  k_next();
  int a_o1 = kw.a;  // A offset of K-tile kt + 1 (its A_hi half is staged in phase 1)
  STAGE(b_lo_src, OFF_B_LO, 1, kw.w)
  STAGE(a_src[0], OFF_A_LO, 1, kw.a)
  STAGE_BHI(1, kw.w)
  WAIT_3HALVES()
  BAR()
  if (wr == 1) BAR()
  for (int kt = 0; kt < nk; kt += 2) {
    if (SPLIT == 2 && kt == 2 * nk0) {  // fp16 x 2: the two cross blocks are in — scale them by 2^-11 before the hi.hi block (nk0 even)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] *= 0.00048828125f;
    }
    k_next();
    const int a_e2 = kw.a, w_e2 = kw.w;  // K-tile min(kt + 2, nk - 1)
    k_next();
    const int a_o3 = kw.a, w_o3 = kw.w;  // K-tile min(kt + 3, nk - 1)
    // p1
    READ_BL(0)
    __builtin_amdgcn_sched_barrier(0);
    READ_A(0, OFF_A_LO)
    STAGE(a_src[1], OFF_A_HI, 1, a_o1)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 0, 2, fbl)
    BAR()
    // p2
    READ_BH(0)
    STAGE(b_lo_src, OFF_B_LO, 0, w_e2)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 2, NBH, fbh)
    BAR()
    // p3
    READ_A(0, OFF_A_HI)
    STAGE(a_src[0], OFF_A_LO, 0, a_e2)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(1, 2, NBH, fbh)
    BAR()
    // p4
    STAGE_BHI(0, w_e2)
    WAIT_3HALVES()
    BAR()
    MFMA_Q(1, 0, 2, fbl)
    BAR()
    // p5
    READ_BL(1)
    __builtin_amdgcn_sched_barrier(0);
    READ_A(1, OFF_A_LO)
    STAGE(a_src[1], OFF_A_HI, 0, a_e2)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 0, 2, fbl)
    BAR()
    // p6
    READ_BH(1)
    STAGE(b_lo_src, OFF_B_LO, 1, w_o3)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 2, NBH, fbh)
    BAR()
    // p7
    READ_A(1, OFF_A_HI)
    STAGE(a_src[0], OFF_A_LO, 1, a_o3)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(1, 2, NBH, fbh)
    BAR()
    // p8
    STAGE_BHI(1, w_o3)
    WAIT_3HALVES()
    BAR()
    MFMA_Q(1, 0, 2, fbl)
    BAR()
    a_o1 = a_o3;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wr == 0) BAR()
  // ---- epilogue: row m = m0 + wr*128 + mi*16 + r16, columns n0 + wc*WCOLS + ni*16 + 4*q4 + 0..3 (N % 4 == 0, 16-byte rows: launcher)
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int64_t m = m0 + wr * 128 + mi * 16 + r16;
    if (m >= Mv) continue;
#pragma unroll
    for (int ni = 0; ni < NB; ++ni) {
      const int n = n0 + wc * WCOLS + ni * 16 + 4 * q4;
      if (n >= g.N) continue;
      float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (g.has_bias) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += b.x, v[1] += b.y, v[2] += b.z, v[3] += b.w;
      }
      if (g.has_residual) {
        const float4 r = *reinterpret_cast<const float4*>(g.residual + m * g.ldr + n);
        v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (g.act == 1) v[j] = fmaxf(v[j], 0.f);
        if (g.act == 2) v[j] = gelu_erf_b(v[j]);
      }
      if (g.out_bf16 >= 2) {  // plane rows of a split-form operand (row stride ldc elements, plane stride N): 2 = bf16 x 3, 3 = fp16 x 2
        store_planes(g.C, m * g.ldc + n, g.N, v, g.out_bf16);
      } else if (g.out_bf16) {
        union {
          __bf16 h[4];
          uint2 u;
        } o;
        o.h[0] = (__bf16)v[0], o.h[1] = (__bf16)v[1], o.h[2] = (__bf16)v[2], o.h[3] = (__bf16)v[3];
        *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(g.C) + m * g.ldc + n) = o.u;
      } else {
        *reinterpret_cast<float4*>(g.C + m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}
#pragma clang diagnostic pop
#undef GLDS16
#undef STAGE
#undef STAGE_BHI
#undef READ_A
#undef READ_BL
#undef READ_BH
#undef MFMA_Q
#undef WAIT_3HALVES
#undef BAR

// 0: the 128-row forms; 1: 256 x 256 tiles; 2: 256 x 192 tiles.
// WHETHER a 256-row tile runs is a hard gate measured IN SITU: only the deep contractions (K >= 2048) at >= 8 192 rows keep it.  Alone and
// replayed back to back the wide K = 768 linears gain 5-15 % on it, inside the encoder / decode chains they LOSE (bf16 C2 step 117.6 k ->
// 112.3 k q/s, C5 10 410 -> 10 328 with every shape routed by a cost model against the 128-row form): a 512-thread workgroup that owns a
// CU's whole LDS shuts out the other chain's kernels, and between other launches its operands are no longer L2 / MALL-hot as in a replay.
// WHICH width is a cost model in microseconds fitted to one sweep of the forms forced over M = 4 096 .. 20 480 x the six call shapes
// (tools/exp_bf16_linear.py on builds with -DGDR_LAB_BF16_FORCE=1/2, profiles/r05_bf16_tile_forms.txt); nk = K / 64:
//   256-row tiles of width w (one per CU):   ceil(T / 256) x (fix_w + 1.49 nk w / 256),  fix = 12.1 (w = 256) / 11.5 (w = 192)
//   either width:                            at least bytes / 3.5 TB/s (operands + output + residual)
// (lab forcing: -DGDR_LAB_BF16_FORCE=0/1/2 builds a library that always answers 0 / 1 / 2 where the launcher's preconditions allow)
static int pick_tile256(int64_t M, int N, int K, int has_residual, int out_bf16) {
#ifdef GDR_LAB_BF16_FORCE
  return GDR_LAB_BF16_FORCE;
#else
  if (K < 2048 || M < 8192) return 0;
  const double nk = K / 64.0;
  const double bytes = 2.0 * M * K + 2.0 * N * K + (out_bf16 ? 2.0 : 4.0) * M * N + (has_residual ? 4.0 * M * N : 0.0);
  const double floor_us = bytes / 3.5e6;
  const int64_t pm = (M + 255) / 256;
  auto est = [&](int bn, double fix) {
    const int64_t tiles = pm * ((N + bn - 1) / bn);
    const double t = (double)((tiles + 255) / 256) * (fix + 1.49 * nk * bn / 256.0);
    return t > floor_us ? t : floor_us;
  };
  return est(256, 12.1) < est(192, 11.5) ? 1 : 2;
#endif
}

// The form launch_linear_bf16_glds gives an aligned launch (gdr_linear_bf16_tile_form): must mirror the branches below.
int linear_bf16_tile_form(int64_t M, int N, int K, int has_residual, int out_bf16) {
  if (K % 64 != 0) return 0;
  if (K % 128 == 0 && N % 4 == 0) {
    const int sel = pick_tile256(M, N, K, has_residual, out_bf16);
    if (sel) return sel == 1 ? 256 : 192;
  }
  return ((M + 127) / 128) * (int64_t)((N + 127) / 128) < 512 ? 64 : 128;
}

// Returns 1 if the shape is not served here (caller falls back to the generic core), 0 on launch, < 0 on error.
int launch_linear_bf16_glds(const void* A, int64_t lda, const void* W, int64_t ldw, float* C, int64_t ldc, int64_t M, int N,
                            int K, int has_bias, int has_residual, int act, const float* bias, const float* residual,
                            int64_t ldr, int out_bf16, hipStream_t stream, const int64_t* m_dev, int split) {
  if (K % 64 != 0 || lda % 8 != 0 || ldw % 8 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return 1;  // 16-byte DMA pieces
  // split form: K is the real contraction length K0, rows hold [hi | mid | lo]; `split` = number of plane-pair blocks of K0 virtual
  // elements each: 6 = the fp32-level form; 3 = hi.hi + hi.mid + mid.hi only (16 significand bits: a measured, narrower knob)
  // split = 2: the fp16 x 2 form — rows hold [hi | lo'] (2 K0 elements), three blocks (SPLIT = 2 of the kernels)
  const int split_k0 = split ? K : 0;
  const int kform = split == 2 ? 2 : split ? 1 : 0;
  if (split) {
    const int planes = split == 2 ? 2 : 3;
    if ((split != 2 && split != 3 && split != 6) || lda < planes * (int64_t)K || ldw < planes * (int64_t)K) return 1;
    if (split == 2 && K % 128 != 0) return 1;  // the scale point must fall between two K-tile pairs
    K *= split == 2 ? 3 : split;
  }
  if (((uintptr_t)C & 15) || (has_bias && ((uintptr_t)bias & 15)) || (has_residual && ((uintptr_t)residual & 15))) return 1;
  Bf16GemmArgs g{};
  g.A = static_cast<const char*>(A), g.W = static_cast<const char*>(W), g.C = C, g.bias = bias, g.residual = residual;
  g.lda = lda, g.ldw = ldw, g.ldc = ldc, g.ldr = ldr, g.M = M, g.N = N, g.Nrows = N, g.K = K;
  g.tiles_n = (N + 127) / 128;
  g.has_bias = has_bias, g.has_residual = has_residual, g.act = act, g.out_bf16 = out_bf16;
  g.m_dev = m_dev;
  g.split_k0 = split_k0;
  int64_t blocks = ((M + 127) / 128) * g.tiles_n;
  if (blocks <= 0) return 0;
  if (blocks > 0x7fffffffLL) {
    set_error("linear_bf16: grid too large");
    return GDR_EINVAL;
  }
  // The 256-row tile (gemm_nt_bf16_tile256_kernel, 8-phase schedule, one workgroup per CU), 256 or 192 columns wide: twice the flops
  // per operand byte of the 128^2 kernel, but every CU reaches its prologue / epilogue alone, so it pays where the K loop is long or
  // where its tile count fills whole rounds of 256 CUs.  pick_tile256: a hard in-situ gate (K >= 2048, M >= 8 192), then the width by
  // a fitted cost model.
  if (K % 128 == 0 && N % 4 == 0 && (ldc & 3) == 0 && (!has_residual || (ldr & 3) == 0)) {
    const int sel = pick_tile256(M, N, K, has_residual, out_bf16);
    if (sel) {
      const int bn = sel == 1 ? 256 : 192;
      const int64_t b256 = ((M + 255) / 256) * (int64_t)((N + bn - 1) / bn);
#define LAUNCH256(BN_, SP_)                                                                                                        \
  {                                                                                                                              \
    if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(gemm_nt_bf16_tile256_kernel<BN_, SP_>), 2 * BUF_BYTES, "linear_bf16")) return rc__; \
    hipLaunchKernelGGL((gemm_nt_bf16_tile256_kernel<BN_, SP_>), dim3((unsigned)b256), dim3(512), 2 * BUF_BYTES, stream, g);      \
  }
      if (sel == 1) {
        if (kform == 2) LAUNCH256(256, 2) else if (kform == 1) LAUNCH256(256, 1) else LAUNCH256(256, 0)
      } else {
        if (kform == 2) LAUNCH256(192, 2) else if (kform == 1) LAUNCH256(192, 1) else LAUNCH256(192, 0)
      }
#undef LAUNCH256
      GDR_CHECK_LAUNCH("gemm_nt_bf16_tile256_kernel");
      return 0;
    }
  }
  // fewer than 2 tiles of 128 rows per CU: 64-row tiles (same k order per output element: bit-identical results)
  // (tools/exp_bf16_linear.py, profiles/r04_bf16_tile_height_sweep.txt: 1 920 rows qkv 21.6 -> 18.2 us, o 18.9 -> 15.2, wi 22.8 ->
  // 19.9, wo 57.4 -> 43.6; 4 096 rows o 23.2 -> 17.5, wo 58.5 -> 46.2; from ~2 tiles per CU on the 128-row form is as fast or faster)
  if (blocks < 512) {
    blocks = ((M + 63) / 64) * g.tiles_n;
    if (kform == 2)
      hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<0, 64, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, g);
    else if (kform == 1)
      hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<0, 64, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, g);
    else
      hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<0, 64>), dim3((unsigned)blocks), dim3(256), 0, stream, g);
    GDR_CHECK_LAUNCH("gemm_nt_bf16_glds_kernel<64-row tiles>");
    return 0;
  }
  if (kform == 2)
    hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<0, 128, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, g);
  else if (kform == 1)
    hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<0, 128, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<0, 128>), dim3((unsigned)blocks), dim3(256), 0, stream, g);
  GDR_CHECK_LAUNCH("gemm_nt_bf16_glds_kernel");
  return 0;
}

// The decode head in the bf16 precision mode (modeling_t5.py:1634-1646): per row the product with head_w is a [V+1, d] matrix that is
// only ever dotted with the row's hidden state, logits[c] = sum_i (h[i] * d^-0.5) * (A[c*d + i] + E[c][i]).  With thousands of rows that
// matrix is 1.46 GB per step at 15 360 rows — written by the GEMM, read once by head_logits.  Here the GEMM's epilogue takes the dot
// itself: each wave leaves the sum over its 64 columns (d / 64 partials per (row, c), 23 MB instead of 1.46 GB), and the logits kernel adds
// them in a fixed order.  A, W bf16 as in the linear; h fp32 rows reached through rows_map (the compacted row's original row).
// Returns 1 if the shape is not served (the caller runs the linear + the plain dot).
int launch_linear_bf16_headdot(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, const int64_t* m_dev, int N, int K,
                               const float* h, int64_t ldh, const int32_t* rows_map, const float* E, int dot_d, float scale,
                               float* partial, hipStream_t stream) {
  if (K % 64 != 0 || lda % 8 != 0 || ldw % 8 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return 1;
  if (dot_d % 128 != 0 || N % dot_d != 0 || ldh % 4 != 0 || ((uintptr_t)h & 15) || ((uintptr_t)E & 15) || !rows_map || !partial) return 1;
  Bf16GemmArgs g{};
  g.A = static_cast<const char*>(A), g.W = static_cast<const char*>(W);
  g.lda = lda, g.ldw = ldw, g.M = M, g.N = N, g.Nrows = N, g.K = K;
  g.tiles_n = N / 128;
  g.m_dev = m_dev;
  g.dot_h = h, g.dot_ldh = ldh, g.dot_rows = rows_map, g.dot_e = E, g.dot_out = partial, g.dot_scale = scale, g.dot_d = dot_d;
  int64_t blocks = ((M + 127) / 128) * g.tiles_n;
  if (blocks <= 0) return 0;
  if (blocks > 0x7fffffffLL) {
    set_error("linear_bf16(head dot): grid too large");
    return GDR_EINVAL;
  }
  hipLaunchKernelGGL((gemm_nt_bf16_glds_kernel<3, 128>), dim3((unsigned)blocks), dim3(256), 0, stream, g);
  GDR_CHECK_LAUNCH("gemm_nt_bf16_glds_kernel(head dot)");
  return 0;
}

// Similarity passes on the LDS-DMA core (bf16 corpus, bf16 queries, d % 64 == 0).  Returns 1 if not served.
int launch_sim_bf16_glds(const void* D, int64_t N, const void* Q, int B, int d, const SimEpilogue& ep, int64_t n_doc_tiles,
                         hipStream_t stream) {
  if (d % 64 != 0 || ((uintptr_t)D & 15) || ((uintptr_t)Q & 15)) return 1;
  Bf16GemmArgs g{};
  g.A = static_cast<const char*>(Q), g.W = static_cast<const char*>(D);
  g.lda = d, g.ldw = d, g.M = B, g.Nrows = N, g.N = 0, g.K = d;
  g.tiles_n = (B + 127) / 128;  // query tiles
  g.sim = ep;
  const int64_t blocks = n_doc_tiles * g.tiles_n;
  if (blocks <= 0) return 0;
  if (blocks > 0x7fffffffLL) {
    set_error("sim_bf16: grid too large");
    return GDR_EINVAL;
  }
  if (ep.mode == 1)
    hipLaunchKernelGGL(gemm_nt_bf16_glds_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL(gemm_nt_bf16_glds_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, g);
  GDR_CHECK_LAUNCH("gemm_nt_bf16_glds_kernel(sim)");
  return 0;
}

}  // namespace gdr

extern "C" int gdr_linear_bf16_tile_form(int64_t M, int N, int K, int epilogue) {
  const int has_residual = epilogue == GDR_EPI_RESIDUAL || epilogue == GDR_EPI_BIAS_RESIDUAL;
  return gdr::linear_bf16_tile_form(M, N, K, has_residual, 0);
}

// ---- fp32 carried as three bf16 planes (r06, exploratory: VERDICT r05 #8) --------------------------------------------------------
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (RNE; the two differences are exact in fp32): 24
// significand bits.  A linear over such operands keeps the six leading products hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid on the
// bf16 MFMA path with fp32 accumulate (the dropped ones are below 2^-24 of the product): measured error against float64 on the
// encoder's shapes 7e-6 .. 1.3e-5 of mean |c| — the strict-fp32 MFMA linear's own 6e-6 .. 1.5e-5 (tools/exp_split_bf16.py) — at 1/16 x 6
// of the fp32 MFMA cost.  NOT bit-identical to the fp32 chain: an opt-in mode beside the fp32 path, never its replacement.
namespace gdr {
__global__ __launch_bounds__(256) void split_f32_bf16x3_kernel(const float* __restrict__ in, int64_t ld_in, __bf16* __restrict__ out,
                                                              int64_t ld_out, int64_t rows, int K, const int64_t* __restrict__ rows_dev,
                                                              int form) {
  const int k4 = K >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = i / k4;
  const int c = (int)(i - r * k4) * 4;
  if (r >= (rows_dev ? *rows_dev : rows)) return;
  const float4 v = *reinterpret_cast<const float4*>(in + r * ld_in + c);
  const float x[4] = {v.x, v.y, v.z, v.w};
  store_planes(out, r * ld_out + c, K, x, form);  // form 2: bf16 x 3 [hi | mid | lo]; 3: fp16 x 2 [hi | lo']
}

int launch_split_f32_bf16x3(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int K, const int64_t* rows_dev,
                            hipStream_t stream, int f16x2) {
  if (rows == 0) return GDR_OK;
  GDR_CHECK_ARG(in && out && K > 0 && K % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_out >= (f16x2 ? 2 : 3) * (int64_t)K &&
                    ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 7) == 0,
                "split: bad arguments (K %% 4 == 0, ld_out >= 3 K (bf16 x 3) or 2 K (fp16 x 2))");
  const int64_t n = rows * (K / 4);
  hipLaunchKernelGGL(split_f32_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, ld_in, static_cast<__bf16*>(out),
                     ld_out, rows, K, rows_dev, f16x2 ? 3 : 2);
  GDR_CHECK_LAUNCH("split_f32_bf16x3_kernel");
  return GDR_OK;
}
}  // namespace gdr

namespace gdr {
// Elements per row of a plane-form operand with contraction length K: 3 K rounded up to 64 elements (128 B, the DMA's row piece).
// (A first measurement suggested a sensitivity to the row stride — 254 us at 4 608 B against 203 us at 5 248 B; repeated in a fixed
// order it was the clock ramp of the first seconds of a process, tools/exp_split_pad.py: 255 / 254 / 235 / 225 / 216 / 211 us for pads
// 0 / 320 / 0 / 320 / 1024 / 320 — so no padding rule is kept.)
int split_row_elems(int K, int f16x2) { return ((f16x2 ? 2 : 3) * K + 63) / 64 * 64; }
}  // namespace gdr

extern "C" int gdr_split_row_elems(int K, int terms) { return K > 0 ? gdr::split_row_elems(K, terms == 2) : 0; }

extern "C" int gdr_split_f32_bf16x3(const float* in, void* out_planes, int64_t rows, int K, int64_t ld_out, void* stream) {
  return gdr::launch_split_f32_bf16x3(in, K, out_planes, ld_out, rows, K, nullptr, static_cast<hipStream_t>(stream), 0);
}

extern "C" int gdr_split_f32_f16x2(const float* in, void* out_planes, int64_t rows, int K, int64_t ld_out, void* stream) {
  return gdr::launch_split_f32_bf16x3(in, K, out_planes, ld_out, rows, K, nullptr, static_cast<hipStream_t>(stream), 1);
}

extern "C" int gdr_linear_split_bf16(const void* A3, int64_t lda, const void* W3, int64_t ldw, float* C, int64_t ldc, int64_t M, int N, int K,
                                     int terms, int epilogue, const float* bias, const float* residual, int64_t ldr, void* stream) {
  using namespace gdr;
  GDR_CHECK_ARG(A3 && W3 && C && M >= 0 && N > 0 && K > 0 && (terms == 6 || terms == 3 || terms == 2),
                "linear_split_bf16: bad arguments (terms: 6 or 3 = bf16 planes, 2 = fp16 x 2)");
  if (M == 0) return GDR_OK;
  const bool nb = epilogue == GDR_EPI_BIAS || epilogue == GDR_EPI_BIAS_RELU || epilogue == GDR_EPI_BIAS_RESIDUAL || epilogue == GDR_EPI_BIAS_GELU;
  const bool nr = epilogue == GDR_EPI_RESIDUAL || epilogue == GDR_EPI_BIAS_RESIDUAL;
  const int act = (epilogue == GDR_EPI_RELU || epilogue == GDR_EPI_BIAS_RELU) ? 1 : epilogue == GDR_EPI_BIAS_GELU ? 2 : 0;
  GDR_CHECK_ARG((!nb || bias) && (!nr || residual), "linear_split_bf16: the epilogue's bias / residual pointer is null");
  ProfScope prof(PROF_LINEAR, 2.0 * (double)M * (double)N * (double)K, static_cast<hipStream_t>(stream));
  const int rc = launch_linear_bf16_glds(A3, lda, W3, ldw, C, ldc, M, N, K, nb, nr, act, bias, residual, ldr, 0, static_cast<hipStream_t>(stream),
                                         nullptr, terms);
  if (rc > 0) {
    set_error("linear_split_bf16: shape not served (K %% 64 == 0, lda / ldw >= 3 K and multiples of 8, 16-byte aligned operands)");
    return GDR_EINVAL;
  }
  return rc;
}
