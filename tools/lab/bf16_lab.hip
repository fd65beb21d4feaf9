// Lab: the bf16 LDS-DMA linear (gdr_amd/csrc/gemm_bf16.hip) as one tile per workgroup against a persistent form with a
// stream-K tail (whole-tile rounds, then K-step ranges with an exact accumulator hand-off, as gemm_f32.hip does for fp32).
// Not product code.   hipcc -O3 --offload-arch=gfx950 tools/lab/bf16_lab.hip -o tools/lab/bf16_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8b __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const char* A;
  const char* W;
  float* C;
  const float* residual;
  int64_t lda, ldw, ldc, ldr, M;
  int N, K, tiles_n, has_residual, act;
};
struct SkArgs {
  float* part;
  int32_t* flag;
  int32_t epoch;
  int32_t* err;
};

#define XCD_REMAP(bid)                                                                                         \
  {                                                                                                            \
    const unsigned nblk = gridDim.x, q_ = nblk >> 3, r_ = nblk & 7u, xcd_ = bid & 7u, j_ = bid >> 3;           \
    bid = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + j_;                               \
  }

template <int OUT_BF16 = 0>
__device__ __forceinline__ void epilogue_tile(const Args& g, f32x4b (&acc)[4][4], int64_t m0, int64_t n0, int wm, int wn, int r16,
                                              int q4) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t m = m0 + wm * 64 + mi * 16 + r16;
    if (m >= g.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = (int)n0 + wn * 64 + ni * 16 + 4 * q4;
      float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (g.has_residual) {
        const float4 r = *reinterpret_cast<const float4*>(g.residual + m * g.ldr + n);
        v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
      }
      if (g.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (OUT_BF16) {
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

template <int OUT_BF16 = 0>
__global__ __launch_bounds__(256, 3) void one_tile_kernel(const Args g) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 128 * 128];
  char* const As = smem;
  char* const Bs = smem + 128 * 128;
  unsigned bid = blockIdx.x;
  XCD_REMAP(bid)
  const int64_t m0 = (int64_t)(bid / (unsigned)g.tiles_n) * 128, n0 = (int64_t)(bid % (unsigned)g.tiles_n) * 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;
  const int srow = lane >> 3, schunk = lane & 7;
  const char* a_src[4];
  const char* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + srow;
    const int chunk = schunk ^ ((row >> 1) & 7);
    int64_t ra = m0 + row;
    ra = ra < g.M ? ra : g.M - 1;
    int64_t rw = n0 + row;
    rw = rw < g.N ? rw : g.N - 1;
    a_src[i] = g.A + (ra * g.lda) * 2 + chunk * 16;
    w_src[i] = g.W + (rw * g.ldw) * 2 + chunk * 16;
  }
  int a_off[4], b_off[4], a_sw[4], b_sw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + r16, rb = wn * 64 + i * 16 + r16;
    a_off[i] = ra * 128, a_sw[i] = (ra >> 1) & 7;
    b_off[i] = rb * 128, b_sw[i] = (rb >> 1) & 7;
  }
  f32x4b acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  const int nk = g.K >> 6;
  for (int kt = 0; kt < nk; ++kt) {
    const int koff = kt * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(As + (wave * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + koff),
                                       (__attribute__((address_space(3))) void*)(Bs + (wave * 4 + i) * 1024), 16, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float4 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const float4*>(As + a_off[i] + (((kk * 4 + q4) ^ a_sw[i]) << 4));
        fb[i] = *reinterpret_cast<const float4*>(Bs + b_off[i] + (((kk * 4 + q4) ^ b_sw[i]) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, fb[ni]),
                                                                __builtin_bit_cast(bf16x8b, fa[mi]), acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }
  epilogue_tile<OUT_BF16>(g, acc, m0, n0, wm, wn, r16, q4);
}

// Persistent + stream-K tail.  WPC = workgroups per CU the launch bound allows.
template <int WPC>
__global__ __launch_bounds__(256, WPC) void streamk_kernel(const Args g, const int total_tiles, const SkArgs sk) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 128 * 128];
  char* const As = smem;
  char* const Bs = smem + 128 * 128;
  unsigned bid = blockIdx.x;
  XCD_REMAP(bid)
  const int G = (int)gridDim.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;
  const int srow = lane >> 3, schunk = lane & 7;
  int a_off[4], b_off[4], a_sw[4], b_sw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + r16, rb = wn * 64 + i * 16 + r16;
    a_off[i] = ra * 128, a_sw[i] = (ra >> 1) & 7;
    b_off[i] = rb * 128, b_sw[i] = (rb >> 1) & 7;
  }
  const int nk = g.K >> 6;
  // ---- this workgroup's segments, in processing order (gemm_f32.hip): whole tiles bid + s*G of the lockstep rounds,
  //      [begun fragment: tile t_last, k 0..k_last) -> published]  [whole tiles]  [continued fragment <- taken over]
  int t_first, k_first, t_last, k_last, dp_rounds = 0;
  if (total_tiles >= G) {
    const int rounds = total_tiles / G;
    dp_rounds = total_tiles - rounds * G ? rounds - 1 : rounds;
    const int sk_base = dp_rounds * G;
    const int64_t iters = (int64_t)(total_tiles - sk_base) * nk;
    const int64_t lo = (int64_t)bid * iters / G, hi = (int64_t)(bid + 1) * iters / G;
    t_first = (int)(lo / nk), k_first = (int)(lo - (int64_t)t_first * nk);
    t_last = (int)(hi / nk), k_last = (int)(hi - (int64_t)t_last * nk);
    t_first += sk_base, t_last += sk_base;
  } else {
    t_first = min((int)bid, total_tiles), k_first = 0;
    t_last = min((int)bid + 1, total_tiles), k_last = 0;
  }
  const bool has_head = k_first != 0, has_tail = k_last != 0;
  const int t_full0 = has_head ? t_first + 1 : t_first;
  const int n_full = t_last - t_full0;
  const int nseg = dp_rounds + (has_tail ? 1 : 0) + n_full + (has_head ? 1 : 0);
  f32x4b acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4b){0.f, 0.f, 0.f, 0.f};

  for (int seg = 0; seg < nseg; ++seg) {
    int tile, k0, k1;
    {
      int q_ = seg - dp_rounds;
      if (q_ < 0) {
        tile = (int)bid + seg * G, k0 = 0, k1 = nk;
      } else if (has_tail && q_ == 0) {
        tile = t_last, k0 = 0, k1 = k_last;
      } else {
        q_ -= has_tail ? 1 : 0;
        if (q_ < n_full)
          tile = t_full0 + q_, k0 = 0, k1 = nk;
        else
          tile = t_first, k0 = k_first, k1 = nk;
      }
    }
    const int64_t m0 = (int64_t)(tile / g.tiles_n) * 128, n0 = (int64_t)(tile % g.tiles_n) * 128;
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + srow;
      const int chunk = schunk ^ ((row >> 1) & 7);
      int64_t ra = m0 + row;
      ra = ra < g.M ? ra : g.M - 1;
      int64_t rw = n0 + row;
      rw = rw < g.N ? rw : g.N - 1;
      a_src[i] = g.A + (ra * g.lda) * 2 + chunk * 16;
      w_src[i] = g.W + (rw * g.ldw) * 2 + chunk * 16;
    }
    if (k0 != 0) {  // the CONTINUED fragment: take over the predecessor's accumulators
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(sk.flag + (bid - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 24)) {
            if (sk.err) __hip_atomic_store(sk.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      const float4* src = reinterpret_cast<const float4*>(sk.part + (size_t)(bid - 1) * (128 * 128)) + tid;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const float4 v = src[(mi * 4 + ni) * 256];
          acc[mi][ni][0] = v.x, acc[mi][ni][1] = v.y, acc[mi][ni][2] = v.z, acc[mi][ni][3] = v.w;
        }
    }
    for (int kt = k0; kt < k1; ++kt) {
      const int koff = kt * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                         (__attribute__((address_space(3))) void*)(As + (wave * 4 + i) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + koff),
                                         (__attribute__((address_space(3))) void*)(Bs + (wave * 4 + i) * 1024), 16, 0, 0);
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float4 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = *reinterpret_cast<const float4*>(As + a_off[i] + (((kk * 4 + q4) ^ a_sw[i]) << 4));
          fb[i] = *reinterpret_cast<const float4*>(Bs + b_off[i] + (((kk * 4 + q4) ^ b_sw[i]) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, fb[ni]),
                                                                  __builtin_bit_cast(bf16x8b, fa[mi]), acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
    }
    if (k1 != nk) {  // a BEGUN fragment: publish the raw accumulators write-through, then the flag
      const __amdgpu_buffer_rsrc_t dst =
          __builtin_amdgcn_make_buffer_rsrc(sk.part + (size_t)bid * (128 * 128), 0, 128 * 128 * 4, 0x00020000);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          u32x4 v;
          v[0] = __float_as_uint(acc[mi][ni][0]), v[1] = __float_as_uint(acc[mi][ni][1]);
          v[2] = __float_as_uint(acc[mi][ni][2]), v[3] = __float_as_uint(acc[mi][ni][3]);
          __builtin_amdgcn_raw_buffer_store_b128(v, dst, (((mi * 4 + ni) * 256) + tid) * 16, 0, 16);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(sk.flag + bid, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      epilogue_tile(g, acc, m0, n0, wm, wn, r16, q4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4), BK = 64, one workgroup per CU, 128 KiB LDS = 2 buffers x {A_lo, A_hi, B_lo, B_hi} half-tiles of
// 128 rows x 128 B.  Four phases per K-tile, each: [fragment reads of the phase + one half-tile of LDS-DMA prefetch] barrier
// [16 MFMAs on one 64x32 quadrant of the wave's 128x64 block] barrier.  The two wave rows run one barrier apart, so that on every
// SIMD one wave is in its MFMA cluster while the other reads / stages.  The DMA stays in flight across barriers: counted
// vmcnt(6) once per K-tile, never 0 in the loop.  Half h of A holds, for each wave row wr, tile rows wr*128 + h*64 .. +63; half
// h of B holds, for each wave column wc, tile columns wc*64 + h*32 .. +31 — so "quadrant (mh, nh)" reads only halves mh / nh.
// Stage schedule (cdna_hip_programming.md "The 256^2 8-phase template"): over two K-tiles E (buffer 0), O (buffer 1)
//   p1 reads E.B_lo E.A_lo   stages O.A_hi           p5 reads O.B_lo O.A_lo   stages E'.A_hi
//   p2 reads E.B_hi          stages E'.B_lo          p6 reads O.B_hi          stages O'.B_lo
//   p3 reads E.A_hi          stages E'.A_lo          p7 reads O.A_hi          stages O'.A_lo
//   p4 -                     stages E'.B_hi, vmcnt(6) p8 -                    stages O'.B_hi, vmcnt(6)
#define GLDS16(gptr_, lds_) \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_), "v"(gptr_) : "memory", "m0")

constexpr int HALF_BYTES = 128 * 128;                 // one half-tile
constexpr int OFF_A_LO = 0, OFF_A_HI = HALF_BYTES, OFF_B_LO = 2 * HALF_BYTES, OFF_B_HI = 3 * HALF_BYTES, BUF_BYTES = 4 * HALF_BYTES;

__global__ __launch_bounds__(512, 1) void tile256_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(1024))) char smem256[];
  unsigned bid = blockIdx.x;
  XCD_REMAP(bid)
  const int tiles_n = (g.N + 255) / 256;
  const int64_t m0 = (int64_t)(bid / (unsigned)tiles_n) * 256;
  const int n0 = (int)(bid % (unsigned)tiles_n) * 256;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, q4 = lane >> 4;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem256));
  // ---- staging: instruction i of wave w covers local rows (w*2 + i)*8 + (lane>>3) of a half-tile
  const int srow = lane >> 3, schunk = lane & 7;
  const char* a_src[2][2];  // [half][instr]
  const char* b_src[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave * 2 + i) * 8 + srow;
      const int chunk = schunk ^ ((lr >> 1) & 7);
      int64_t ra = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
      ra = ra < g.M ? ra : g.M - 1;
      int rb = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
      rb = rb < g.N ? rb : g.N - 1;
      a_src[h][i] = g.A + (ra * g.lda) * 2 + chunk * 16;
      b_src[h][i] = g.W + ((int64_t)rb * g.ldw) * 2 + chunk * 16;
    }
  const unsigned st_dst = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 2048;  // + i*1024 + half offset + buffer offset
#define STAGE(src_, half_off_, buf_, kt_)                                                   \
  {                                                                                         \
    const int koff_ = (kt_) * 128;                                                          \
    GLDS16(src_[0] + koff_, st_dst + (buf_)*BUF_BYTES + (half_off_));                       \
    GLDS16(src_[1] + koff_, st_dst + (buf_)*BUF_BYTES + (half_off_) + 1024);                \
  }
  // ---- fragment reads
  const int sw = (r16 >> 1) & 7;
  const int c0 = (q4 ^ sw) << 4;
  const char* const fa_base = smem256 + (wr * 64 + r16) * 128 + c0;
  const char* const fb_base = smem256 + (wc * 32 + r16) * 128 + c0;
  float4 fa[4][2], fbl[2][2], fbh[2][2];
#define READ_A(buf_, half_off_)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
    fa[i][0] = *reinterpret_cast<const float4*>(fa_base + (buf_)*BUF_BYTES + (half_off_) + i * 2048);             \
    fa[i][1] = *reinterpret_cast<const float4*>((fa_base + (buf_)*BUF_BYTES + (half_off_) + i * 2048) + 64 - 2 * (c0 & 64)); \
  }
#define READ_B(fb_, buf_, half_off_)                                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
    fb_[j][0] = *reinterpret_cast<const float4*>(fb_base + (buf_)*BUF_BYTES + (half_off_) + j * 2048);            \
    fb_[j][1] = *reinterpret_cast<const float4*>((fb_base + (buf_)*BUF_BYTES + (half_off_) + j * 2048) + 64 - 2 * (c0 & 64)); \
  }
  f32x4b acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4b){0.f, 0.f, 0.f, 0.f};
#define MFMA_Q(mh_, nh_, fb_)                                                                                      \
  {                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                           \
            acc[(mh_)*4 + i][(nh_)*2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                               \
                __builtin_bit_cast(bf16x8b, fb_[j][kk]), __builtin_bit_cast(bf16x8b, fa[i][kk]), acc[(mh_)*4 + i][(nh_)*2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  }
#define BAR()                              \
  {                                        \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
  }
  const int nk = g.K >> 6;  // even (launcher)
  // prologue: K-tile 0 whole, K-tile 1 without A_hi
  STAGE(b_src[0], OFF_B_LO, 0, 0)
  STAGE(a_src[0], OFF_A_LO, 0, 0)
  STAGE(b_src[1], OFF_B_HI, 0, 0)
  STAGE(a_src[1], OFF_A_HI, 0, 0)
  STAGE(b_src[0], OFF_B_LO, 1, 1)
  STAGE(a_src[0], OFF_A_LO, 1, 1)
  STAGE(b_src[1], OFF_B_HI, 1, 1)
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  BAR()
  if (wr == 1) BAR()
  for (int kt = 0; kt < nk; kt += 2) {
    const int e2 = kt + 2 < nk ? kt + 2 : nk - 1, o1 = kt + 1, o3 = kt + 3 < nk ? kt + 3 : nk - 1;
    // p1
    READ_B(fbl, 0, OFF_B_LO)
    __builtin_amdgcn_sched_barrier(0);
    READ_A(0, OFF_A_LO)
    STAGE(a_src[1], OFF_A_HI, 1, o1)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 0, fbl)
    BAR()
    // p2
    READ_B(fbh, 0, OFF_B_HI)
    STAGE(b_src[0], OFF_B_LO, 0, e2)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 1, fbh)
    BAR()
    // p3
    READ_A(0, OFF_A_HI)
    STAGE(a_src[0], OFF_A_LO, 0, e2)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(1, 1, fbh)
    BAR()
    // p4
    STAGE(b_src[1], OFF_B_HI, 0, e2)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BAR()
    MFMA_Q(1, 0, fbl)
    BAR()
    // p5
    READ_B(fbl, 1, OFF_B_LO)
    __builtin_amdgcn_sched_barrier(0);
    READ_A(1, OFF_A_LO)
    STAGE(a_src[1], OFF_A_HI, 0, e2)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 0, fbl)
    BAR()
    // p6
    READ_B(fbh, 1, OFF_B_HI)
    STAGE(b_src[0], OFF_B_LO, 1, o3)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(0, 1, fbh)
    BAR()
    // p7
    READ_A(1, OFF_A_HI)
    STAGE(a_src[0], OFF_A_LO, 1, o3)
    BAR()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MFMA_Q(1, 1, fbh)
    BAR()
    // p8
    STAGE(b_src[1], OFF_B_HI, 1, o3)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BAR()
    MFMA_Q(1, 0, fbl)
    BAR()
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wr == 0) BAR()
  // ---- epilogue: row m = m0 + wr*128 + mi*16 + r16, columns n0 + wc*64 + ni*16 + 4*q4 + 0..3
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int64_t m = m0 + wr * 128 + mi * 16 + r16;
    if (m >= g.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * q4;
      if (n >= g.N) continue;
      float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (g.has_residual) {
        const float4 r = *reinterpret_cast<const float4*>(g.residual + m * g.ldr + n);
        v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
      }
      if (g.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      *reinterpret_cast<float4*>(g.C + m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same schedule as a PERSISTENT kernel: workgroup b takes tiles b, b + G, ... and runs their K-tiles as one flattened
// stream — the prefetch of a tile's last two K-tiles already stages the next tile's first two, and the epilogue's stores drain
// under the next tile's MFMAs.  32-bit operand offsets + SGPR bases (global_load_lds ... saddr) keep two tiles' pointers affordable.
#define GLDS16S(off_, base_, lds_) \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_), "v"(off_), "s"(base_) : "memory", "m0")

template <int OUT_BF16, int DBG = 0>  // DBG 1: no output stores; 2: every tile reads the A rows of tile row 0 (hot in L2); 3: both
__global__ __launch_bounds__(512, 1) void tile256p_kernel(const Args g, const int total_tiles) {
  extern __shared__ __attribute__((aligned(1024))) char smem256[];
  unsigned bid = blockIdx.x;
  XCD_REMAP(bid)
  const int G = (int)gridDim.x;
  int tile = (int)bid;
  if (tile >= total_tiles) return;
  const int tiles_n = (g.N + 255) / 256;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, q4 = lane >> 4;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem256));
  const int srow = lane >> 3, schunk = lane & 7;
  unsigned a_lo[2], a_hi[2], b_lo[2], b_hi[2];  // byte offsets from g.A / g.W of this thread's two DMA pieces per half-tile
#define SET_A(dst_, t_, h_)                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                    \
    const int lr = (wave * 2 + i) * 8 + srow;                                        \
    int64_t ra = (int64_t)((DBG & 2) ? 0 : (t_) / tiles_n) * 256 + (lr >> 6) * 128 + (h_)*64 + (lr & 63); \
    ra = ra < g.M ? ra : g.M - 1;                                                    \
    dst_[i] = (unsigned)((ra * g.lda) * 2 + ((schunk ^ ((lr >> 1) & 7)) << 4));      \
  }
#define SET_B(dst_, t_, h_)                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                    \
    const int lr = (wave * 2 + i) * 8 + srow;                                        \
    int rb = ((t_) % tiles_n) * 256 + (lr >> 5) * 64 + (h_)*32 + (lr & 31);          \
    rb = rb < g.N ? rb : g.N - 1;                                                    \
    dst_[i] = (unsigned)(((int64_t)rb * g.ldw) * 2 + ((schunk ^ ((lr >> 1) & 7)) << 4)); \
  }
  SET_A(a_lo, tile, 0) SET_A(a_hi, tile, 1) SET_B(b_lo, tile, 0) SET_B(b_hi, tile, 1)
  const unsigned st_dst = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 2048;
#define STAGE_S(off_, base_, half_off_, buf_, kt_)                                                \
  {                                                                                               \
    const unsigned koff_ = (unsigned)(kt_) * 128u;                                                \
    GLDS16S(off_[0] + koff_, base_, st_dst + (buf_)*BUF_BYTES + (half_off_));                     \
    GLDS16S(off_[1] + koff_, base_, st_dst + (buf_)*BUF_BYTES + (half_off_) + 1024);              \
  }
  const int sw = (r16 >> 1) & 7;
  const int c0 = (q4 ^ sw) << 4;
  const char* const fa_base = smem256 + (wr * 64 + r16) * 128 + c0;
  const char* const fb_base = smem256 + (wc * 32 + r16) * 128 + c0;
  float4 fa[4][2], fbl[2][2], fbh[2][2];
  f32x4b acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  const int nk = g.K >> 6;  // even
  STAGE_S(b_lo, g.W, OFF_B_LO, 0, 0)
  STAGE_S(a_lo, g.A, OFF_A_LO, 0, 0)
  STAGE_S(b_hi, g.W, OFF_B_HI, 0, 0)
  STAGE_S(a_hi, g.A, OFF_A_HI, 0, 0)
  STAGE_S(b_lo, g.W, OFF_B_LO, 1, 1)
  STAGE_S(a_lo, g.A, OFF_A_LO, 1, 1)
  STAGE_S(b_hi, g.W, OFF_B_HI, 1, 1)
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  BAR()
  if (wr == 1) BAR()
  // one K-tile = four phases; BUF is its buffer, KA the K-tile whose A_hi phase 1 stages (into the other buffer), K2 the K-tile
  // whose B_lo / A_lo / B_hi phases 2-4 stage (into this buffer)
#define KTILE(BUF, KA, K2)                                    \
  READ_B(fbl, BUF, OFF_B_LO)                                  \
  __builtin_amdgcn_sched_barrier(0);                          \
  READ_A(BUF, OFF_A_LO)                                       \
  STAGE_S(a_hi, g.A, OFF_A_HI, (BUF) ^ 1, KA)                 \
  asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          \
  BAR()                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
  __builtin_amdgcn_sched_barrier(0);                          \
  MFMA_Q(0, 0, fbl)                                           \
  BAR()                                                       \
  READ_B(fbh, BUF, OFF_B_HI)                                  \
  STAGE_S(b_lo, g.W, OFF_B_LO, BUF, K2)                       \
  BAR()                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
  __builtin_amdgcn_sched_barrier(0);                          \
  MFMA_Q(0, 1, fbh)                                           \
  BAR()                                                       \
  READ_A(BUF, OFF_A_HI)                                       \
  STAGE_S(a_lo, g.A, OFF_A_LO, BUF, K2)                       \
  BAR()                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
  __builtin_amdgcn_sched_barrier(0);                          \
  MFMA_Q(1, 1, fbh)                                           \
  BAR()                                                       \
  STAGE_S(b_hi, g.W, OFF_B_HI, BUF, K2)                       \
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            \
  BAR()                                                       \
  MFMA_Q(1, 0, fbl)                                           \
  BAR()
  for (;;) {
    const int nxt = tile + G < total_tiles ? tile + G : tile;  // no successor: the stream re-stages this tile's first K-tiles (never read)
    for (int kt = 0; kt < nk; kt += 2) {
      const bool last = kt + 2 == nk;
      if (last) {  // phases 2-4 of this K-tile stage the successor's K-tile 0
        SET_A(a_lo, nxt, 0) SET_B(b_lo, nxt, 0) SET_B(b_hi, nxt, 1)
      }
      KTILE(0, kt + 1, last ? 0 : kt + 2)
      if (last) {  // phase 1 of the odd K-tile stages the successor's A_hi of K-tile 0
        SET_A(a_hi, nxt, 1)
      }
      KTILE(1, last ? 0 : kt + 2, last ? 1 : kt + 3)
    }
    // ---- epilogue: row m = m0 + wr*128 + mi*16 + r16, columns n0 + wc*64 + ni*16 + 4*q4 + 0..3
    {
      const int64_t m0 = (int64_t)(tile / tiles_n) * 256;
      const int n0 = (tile % tiles_n) * 256;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int64_t m = m0 + wr * 128 + mi * 16 + r16;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = n0 + wc * 64 + ni * 16 + 4 * q4;
          float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
          acc[mi][ni] = (f32x4b){0.f, 0.f, 0.f, 0.f};
          if (m >= g.M || n >= g.N) continue;
          if ((DBG & 1) && v[0] != 123.456f) continue;
          if (g.has_residual) {
            const float4 r = *reinterpret_cast<const float4*>(g.residual + m * g.ldr + n);
            v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
          }
          if (g.act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if (OUT_BF16) {
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
    if (nxt == tile) break;
    tile = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wr == 0) BAR()
}

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

template <typename F>
static double time_us(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int MAXM = 20480, MAXN = 3072, MAXK = 3072;
  std::vector<uint16_t> ha((size_t)MAXM * MAXK), hw((size_t)MAXN * MAXK);
  srand(1);
  for (auto& x : ha) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& x : hw) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.06f);
  char *A, *W;
  float *C0, *C1, *R, *part;
  int32_t* flag;
  CK(hipMalloc(&A, ha.size() * 2));
  CK(hipMalloc(&W, hw.size() * 2));
  CK(hipMalloc(&C0, (size_t)MAXM * MAXN * 4));
  CK(hipMalloc(&C1, (size_t)MAXM * MAXN * 4));
  CK(hipMalloc(&R, (size_t)MAXM * MAXN * 4));
  CK(hipMalloc(&part, (size_t)1024 * 128 * 128 * 4));
  CK(hipMalloc(&flag, 1024 * 4));
  CK(hipMemset(flag, 0, 1024 * 4));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(R, 0, (size_t)MAXM * MAXN * 4));
  CK(hipFuncSetAttribute((const void*)tile256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
  CK(hipFuncSetAttribute((const void*)tile256p_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
  CK(hipFuncSetAttribute((const void*)tile256p_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
  CK(hipFuncSetAttribute((const void*)tile256p_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
  CK(hipFuncSetAttribute((const void*)tile256p_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
  CK(hipFuncSetAttribute((const void*)tile256p_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
  int epoch = 0;
  std::vector<float> h0, h1;
  struct Shape {
    const char* name;
    int N, K, res, act;
  } shapes[] = {{"qkv", 2304, 768, 0, 0}, {"o", 768, 768, 1, 0}, {"wi", 3072, 768, 0, 1}, {"wo", 768, 3072, 1, 0}};
  for (int M : {12308, 20480, 4096}) {
    for (const Shape& s : shapes) {
      Args g{A, W, C0, R, s.K, s.K, s.N, s.N, M, s.N, s.K, s.N / 128, s.res, s.act};
      const int tiles = ((M + 127) / 128) * g.tiles_n;
      const double flop = 2.0 * M * s.N * s.K;
      const double t0 = time_us([&] { hipLaunchKernelGGL(one_tile_kernel<0>, dim3(tiles), dim3(256), 0, 0, g); }, 50);
      const double t0b = time_us([&] { hipLaunchKernelGGL(one_tile_kernel<1>, dim3(tiles), dim3(256), 0, 0, g); }, 50);
      printf("M=%5d %-3s tiles=%4d  one-tile %7.1f us %6.0f TF/s (bf16-out %6.1f us) |", M, s.name, tiles, t0, flop / t0 / 1e6, t0b);
      h0.resize((size_t)M * s.N);
      CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
      Args g1 = g;
      g1.C = C1;
      {
        const int t256 = ((M + 255) / 256) * ((s.N + 255) / 256);
        CK(hipMemset(C1, 0xff, (size_t)M * s.N * 4));
        const double t2 = time_us([&] { hipLaunchKernelGGL(tile256_kernel, dim3(t256), dim3(512), 2 * BUF_BYTES, 0, g1); }, 50);
        h1.resize(h0.size());
        CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
        printf("  256^2 (%d tiles): %7.1f us %6.0f TF/s %s(%zu)", t256, t2, flop / t2 / 1e6, bad ? "DIFF" : "==", bad);
      }
      {
        const int t256 = ((M + 255) / 256) * ((s.N + 255) / 256);
        CK(hipMemset(C1, 0xff, (size_t)M * s.N * 4));
        const int Gp = t256 < 256 ? t256 : 256;
        const double t2 = time_us([&] { hipLaunchKernelGGL(tile256p_kernel<0>, dim3(Gp), dim3(512), 2 * BUF_BYTES, 0, g1, t256); }, 50);
        h1.resize(h0.size());
        CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
        const double t3 = time_us([&] { hipLaunchKernelGGL(tile256p_kernel<1>, dim3(Gp), dim3(512), 2 * BUF_BYTES, 0, g1, t256); }, 50);
        printf("  persistent: %7.1f us %6.0f TF/s %s(%zu)  bf16-out %7.1f us", t2, flop / t2 / 1e6, bad ? "DIFF" : "==", bad, t3);
        const double d1 = time_us([&] { hipLaunchKernelGGL((tile256p_kernel<0, 1>), dim3(Gp), dim3(512), 2 * BUF_BYTES, 0, g1, t256); }, 50);
        const double d2 = time_us([&] { hipLaunchKernelGGL((tile256p_kernel<0, 2>), dim3(Gp), dim3(512), 2 * BUF_BYTES, 0, g1, t256); }, 50);
        const double d3 = time_us([&] { hipLaunchKernelGGL((tile256p_kernel<0, 3>), dim3(Gp), dim3(512), 2 * BUF_BYTES, 0, g1, t256); }, 50);
        printf("  | no-store %6.1f  A-hot %6.1f  both %6.1f", d1, d2, d3);
      }
      for (int G : std::vector<int>{}) {
        int Ge = G;
        if (tiles < Ge) Ge = tiles / 256 * 256;
        if (Ge == 0) Ge = tiles;
        auto launch = [&] {
          const SkArgs ska{part, flag, ++epoch, nullptr};
          if (G <= 768)
            hipLaunchKernelGGL(streamk_kernel<3>, dim3(Ge), dim3(256), 0, 0, g1, tiles, ska);
          else
            hipLaunchKernelGGL(streamk_kernel<4>, dim3(Ge), dim3(256), 0, 0, g1, tiles, ska);
        };
        CK(hipMemset(C1, 0xff, (size_t)M * s.N * 4));
        const double t1 = time_us(launch, 50);
        h1.resize(h0.size());
        CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
        const bool same = memcmp(h0.data(), h1.data(), h0.size() * 4) == 0;
        printf("  G=%4d: %7.1f us %6.0f TF/s %s", Ge, t1, flop / t1 / 1e6, same ? "==" : "DIFF");
      }
      printf("\n");
    }
  }
  return 0;
}
