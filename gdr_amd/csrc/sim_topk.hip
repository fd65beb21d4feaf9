// Fused corpus similarity + top-k for gfx950 (include/gdr_hip.h: gdr_sim_topk, gdr_topk_merge).
//
// Replaces `scores = q_reps @ p_reps.T` (reference GDR_model/dense.py:53-54, encoder.py:128-129)
// followed by `scores.topk(k, largest=True, sorted=True)` (as at main_models.py:1625) without ever
// writing the [B,N] score matrix:
//
//   1. SAMPLE   the GEMM core (gemm_f32.hip) runs over every `stride`-th 128-doc tile of the corpus and
//               stores those scores into per-query candidate lists  cand[q][0 .. n_slots).
//   2. THRESH   one workgroup per query radix-selects the k-th largest sample score thr[q].  The k-th
//               largest over ANY subset of the docs is a lower bound of the k-th largest over all of
//               them, so `score >= thr[q]` can never reject a true top-k doc (ties included).
//   3. FILTER   the GEMM core runs over the remaining tiles; its epilogue compares the accumulators
//               (a lane owns one query: docs sit in the row role) against thr and appends the few
//               survivors (expected k*N/n_sample per query) with one returning atomic each.
//   4. SELECT   one workgroup per query radix-selects the exact k largest 64-bit keys
//               (orderable(score) << 32 | ~doc) — distinct keys, so ties resolve as "higher score, then
//               lower doc id" — then bitonic-sorts them in LDS and writes values + ids.
//
// n_sample ≈ sqrt(k·N) balances list length against survivors.  Small corpora take steps 1, 2, 4 only.
#include <stdlib.h>

#include "common.h"

namespace gdr {

constexpr int TILE = 128;
constexpr int SEL_THREADS = 256;

struct SimPlan {
  int64_t tiles_m;
  int stride;          // sample tiles: index % stride == 0
  int64_t n_sample_tiles;
  int64_t n_slots;     // n_sample_tiles * TILE
  int64_t cap;         // candidate list capacity per query (multiple of TILE)
  size_t off_val, off_idx, off_cnt, off_thr, off_part, total;
};

static SimPlan make_plan(int B, int64_t N, int k, bool exhaustive) {
  SimPlan p{};
  p.tiles_m = (N + TILE - 1) / TILE;
  double target = sqrt((double)k * (double)N);
  if (target < k) target = k;
  int64_t ts = (int64_t)((target + TILE - 1) / TILE);
  if (ts < 1) ts = 1;
  if (exhaustive || N <= 16384 || p.tiles_m < 4 * ts) {
    p.stride = 1;
  } else {
    p.stride = (int)(p.tiles_m / ts);
  }
  p.n_sample_tiles = (p.tiles_m + p.stride - 1) / p.stride;
  p.n_slots = p.n_sample_tiles * TILE;
  if (p.stride == 1) {
    p.cap = p.n_slots;
  } else {
    const double n_sample = (double)p.n_slots;
    const double expect = (double)k * (double)N / n_sample;
    p.cap = p.n_slots + (int64_t)(4.0 * expect) + 4096;
    p.cap = (p.cap + TILE - 1) / TILE * TILE;
  }
  size_t o = 0;
  p.off_val = o, o += align_up((size_t)B * p.cap * sizeof(float), 256);
  p.off_idx = o, o += align_up((size_t)B * p.cap * sizeof(int32_t), 256);
  p.off_cnt = o, o += align_up((size_t)B * CNT_STRIDE * sizeof(int32_t), 256);
  p.off_thr = o, o += align_up((size_t)B * sizeof(float), 256);
  // latency mode (B <= 32): the slices' top-k keys of the sliced threshold / select tails, [B][16 slices][128] x 8 bytes
  p.off_part = o, o += B <= 32 ? align_up((size_t)B * 16 * 128 * sizeof(unsigned long long), 256) : 0;
  p.total = o;
  return p;
}

// ---- orderable keys --------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fkey(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Given hist[256] (LDS) find the highest bin b with  sum(hist[b..255]) >= need.
// Returns b and the count strictly above it through *above.  Every thread of the block calls it (blockDim >= 256).
// `need` is clamped to the number of keys counted (a merge input may hold fewer than k live entries).
// The suffix sums are formed inside the four waves that own the bins (shuffles, no barrier) and joined through four words
// of LDS: two workgroup barriers per call (the Hillis-Steele scan over LDS this replaces took sixteen; with up to eight
// digit passes per select that was most of the kernel's synchronisation).
__device__ __forceinline__ int find_bin(int* hist, int* scan, int& need, int* above, int* res) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int v = t < 256 ? hist[t] : 0;
  int sfx = v;  // suffix sum over this wave's 64 bins: sum of bins lane .. 63
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_down(sfx, off);
    if (lane + off < 64) sfx += u;
  }
  if (t < 256 && lane == 0) scan[wave] = sfx;
  if (t == 0) res[0] = 0, res[1] = 0;
  __syncthreads();
  const int w0 = scan[0], w1 = scan[1], w2 = scan[2], w3 = scan[3];
  const int total = w0 + w1 + w2 + w3;
  if (total < need) need = total;
  if (t < 256 && need > 0) {
    const int higher = wave == 0 ? w1 + w2 + w3 : wave == 1 ? w2 + w3 : wave == 2 ? w3 : 0;
    const int mine = sfx + higher, next = mine - v;  // bins t..255 / t+1..255
    if (mine >= need && next < need) {
      res[0] = t;
      res[1] = next;
    }
  }
  __syncthreads();
  *above = res[1];
  return res[0];
}

// Bitonic sort of kpad <= 1024 64-bit keys in LDS, descending, by the FIRST WAVE alone: LDS operations of one wave execute
// in order, so the 28 .. 55 stages need no workgroup barrier (one before, one after — the callers').
__device__ __forceinline__ void wave0_bitonic_desc(unsigned long long* buf, int kpad) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < (kpad >> 1); t += 64) {
        const int lo_i = (t / stride) * (stride << 1) + (t % stride), hi_i = lo_i + stride;
        const bool desc = ((lo_i & size) == 0);
        const unsigned long long a = buf[lo_i], b = buf[hi_i];
        if ((a < b) == desc) buf[lo_i] = b, buf[hi_i] = a;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// Block-wide min / max of 32-bit keys (red[] holds 2 * 16 words).  Result broadcast to every thread.
__device__ __forceinline__ void block_minmax(uint32_t& lo, uint32_t& hi, uint32_t* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
  }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = lo, red[16 + w] = hi;
  __syncthreads();
  lo = red[0], hi = red[16];
  for (int i = 1; i < nw; ++i) lo = min(lo, red[i]), hi = max(hi, red[16 + i]);
  __syncthreads();
}

// The radix passes below work on RANGE-NORMALISED keys: (key - lo) left-aligned so that the first 8-bit digit already
// spreads the live range over 256 bins.  Similarity scores share sign and exponent, so digits taken from the raw
// float bits spend two full sweeps (and ~10^4 same-address LDS atomics each) without separating anything.

// ---- step 2: per-query threshold = k-th largest of the sample scores ----------------------------
// sub != null (the bf16 pre-filter below): thr[q] = value - sub[q]
__global__ __launch_bounds__(1024) void sim_threshold_kernel(const float* __restrict__ cand_val, int64_t cap,
                                                             int n_slots, int k, float* thr, int32_t* cand_cnt,
                                                             const float* __restrict__ sub) {
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ int res[2];
  __shared__ uint32_t red[32];
  const int q = blockIdx.x;
  const float* v = cand_val + (int64_t)q * cap;
  // the sample scores are read ONCE: up to RC keys per thread stay in registers across the digit passes (every pass used to
  // re-read them from memory — a dependent L2 round trip per sweep); longer lists fall back to re-reading
  constexpr int RC = 8;
  const bool cached = n_slots <= RC * (int)blockDim.x;
  uint32_t rk[RC];
#pragma unroll
  for (int u = 0; u < RC; ++u) {
    const int i = threadIdx.x + u * (int)blockDim.x;
    rk[u] = (cached && i < n_slots) ? fkey(v[i]) : 0u;  // fkey(-inf) = 0x007fffff > 0: 0 marks "no entry"
  }
  const uint32_t kneg = fkey(-INFINITY);
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;  // range of the real scores; -inf padding slots of a ragged tile stay below it
  if (cached) {
#pragma unroll
    for (int u = 0; u < RC; ++u)
      if (rk[u] > kneg) lo = min(lo, rk[u]), hi = max(hi, rk[u]);
  } else {
#pragma unroll 4
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
      const float x = v[i];
      if (x > -INFINITY) {
        const uint32_t key = fkey(x);
        lo = min(lo, key), hi = max(hi, key);
      }
    }
  }
  block_minmax(lo, hi, red);
  if (hi < lo) lo = hi = fkey(-INFINITY);  // nothing but padding
  const int nbits = hi > lo ? 32 - __clz(hi - lo) : 0;
  const int lsh = 32 - nbits;  // nbits == 0: every key equal, no pass runs
  uint32_t prefix = 0;
  int need = k;
  for (int pass = 0; 8 * pass < nbits; ++pass) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    if (cached) {
#pragma unroll
      for (int u = 0; u < RC; ++u) {
        const uint32_t raw = rk[u];
        const uint32_t key = (raw - lo) << lsh;
        const bool match = pass == 0 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
        if (raw != 0u && match && raw >= lo) atomicAdd(&hist[(key >> shift) & 255u], 1);
      }
    } else {
#pragma unroll 4
      for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
        const uint32_t raw = fkey(v[i]);
        const uint32_t key = (raw - lo) << lsh;
        const bool match = pass == 0 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
        if (match && raw >= lo) atomicAdd(&hist[(key >> shift) & 255u], 1);
      }
    }
    __syncthreads();
    int above;
    const int b = find_bin(hist, scan, need, &above, res);
    need -= above;
    prefix |= (uint32_t)b << shift;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float t = fkey_inv(nbits ? (prefix >> lsh) + lo : lo);
    thr[q] = sub ? t - sub[q] : t;
    cand_cnt[(int64_t)q * CNT_STRIDE] = n_slots;  // survivors of the filter pass are appended behind the sample block
  }
}

// ---- step 4 / merge: exact top-k of a candidate list, sorted ------------------------------------
// MERGE = false: entries cand_val/cand_idx[q*cap + i], i < min(cnt[q], cap); thr[q] (the k-th largest sample score,
//                at least k entries reach it) bounds the live range from below
// MERGE = true : entries vals/idx[((g*B + q)*rs + j)*es], i = g*k + j < G*k.  rs = entries per (shard, query) row (k, or
//                k+1 when a trailing status entry rides along), es = element stride (2 when vals/idx interleave as
//                {score, id} pairs).  With rs > k and `status`, status[q] = OR over shards of the trailing entry's id.
template <bool MERGE>
__global__ __launch_bounds__(1024) void topk_select_kernel(const float* __restrict__ vals,
                                                           const int32_t* __restrict__ idxs,
                                                           const int32_t* __restrict__ cnt, int64_t cap, int G, int B,
                                                           int k, int kpad, int32_t idx_offset,
                                                           const float* __restrict__ thr, float* out_val,
                                                           int32_t* out_idx, int32_t* __restrict__ status, int rs,
                                                           int es) {
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ int res[2];
  __shared__ uint32_t red[32];
  __shared__ int n_out;
  extern __shared__ __attribute__((aligned(16))) unsigned long long sortbuf[];  // kpad entries
  const int q = blockIdx.x;
  int count;
  if (MERGE) {
    count = G * k;
    if (status && rs > k && threadIdx.x == 0) {
      int32_t any = 0;
      for (int g = 0; g < G; ++g) any |= idxs[(((int64_t)g * B + q) * rs + k) * es];
      status[q] = any != 0 ? 1 : 0;
    }
  } else {
    const int c = cnt[(int64_t)q * CNT_STRIDE];  // the similarity passes' counters (common.h CNT_STRIDE)
    count = c < (int)cap ? c : (int)cap;
    if (status && threadIdx.x == 0) status[q] = c > (int)cap ? 1 : 0;  // list overflowed: result is a subset's top-k
  }
  auto addr_of = [&](int i) -> int64_t {
    if (MERGE) {
      const int g = i / k, j = i - g * k;
      return (((int64_t)g * B + q) * rs + j) * es;
    }
    return (int64_t)q * cap + i;
  };
  // The entries are read ONCE: up to RC raw keys (score key : 32 | ~id : 32; 0 = padding) per thread stay in registers across
  // the sweeps below — range, up to eight digit passes, gather — each of which used to be a dependent trip to memory per entry
  // (36 us per call at 32 queries, most of it those trips).  Longer lists (count > RC * blockDim) re-read as before.
  constexpr int RC = 16;
  const bool cached = count <= RC * (int)blockDim.x;
  auto raw_at = [&](int i) -> unsigned long long {
    const int64_t a = addr_of(i);
    const int32_t id = idxs[a];
    if (id < 0) return 0ull;
    return ((unsigned long long)fkey(vals[a]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)id);
  };
  unsigned long long rk[RC];
#pragma unroll
  for (int u = 0; u < RC; ++u) {
    const int i = threadIdx.x + u * (int)blockDim.x;
    rk[u] = (cached && i < count) ? raw_at(i) : 0ull;
  }
  // sweep 0: live range of the score keys
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  if (cached) {
#pragma unroll
    for (int u = 0; u < RC; ++u)
      if (rk[u] != 0ull) {
        const uint32_t key = (uint32_t)(rk[u] >> 32);
        lo = min(lo, key), hi = max(hi, key);
      }
  } else {
#pragma unroll 4
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
      const unsigned long long raw = raw_at(i);
      if (raw != 0ull) {
        const uint32_t key = (uint32_t)(raw >> 32);
        lo = min(lo, key), hi = max(hi, key);
      }
    }
  }
  block_minmax(lo, hi, red);
  if (!MERGE && thr) lo = max(lo, fkey(thr[q]));
  if (hi < lo) hi = lo;  // no valid entry at all
  const int nbits = hi > lo ? 32 - __clz(hi - lo) : 0;
  const int lsh = 32 - nbits;
  // normalised 64-bit key: (score key - lo) : nbits | ~id : 32, left-aligned; 0 = padding / below the live range
  auto norm = [&](unsigned long long raw) -> unsigned long long {
    if (raw == 0ull) return 0ull;
    const uint32_t sk = (uint32_t)(raw >> 32);
    if (sk < lo) return 0ull;
    return ((((unsigned long long)(sk - lo)) << 32) | (raw & 0xFFFFFFFFull)) << lsh;
  };
  // number of entries in the live range bounds `need` (a merge input may hold fewer than k valid entries)
  unsigned long long prefix = 0ull;
  int need = k < count ? k : count;
  int want = need;
  bool exact = false;
  for (int pass = 0; pass < 8 && !exact; ++pass) {
    const int shift = 56 - 8 * pass;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    if (cached) {
#pragma unroll
      for (int u = 0; u < RC; ++u) {
        const unsigned long long key = norm(rk[u]);
        const bool match = pass == 0 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
        if (match && key != 0ull) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
      }
    } else {
#pragma unroll 4
      for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const unsigned long long key = norm(raw_at(i));
        const bool match = pass == 0 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
        if (match && key != 0ull) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
      }
    }
    __syncthreads();
    int above;
    const int b = find_bin(hist, scan, need, &above, res);
    if (pass == 0) want = need;  // clamped to the live entries
    need -= above;
    prefix |= (unsigned long long)b << shift;
    exact = hist[b] == need;  // the whole bin is taken: the low bits need no refinement
    __syncthreads();
  }
  // gather the `want` keys >= prefix (raw keys: the sort below needs no range)
  if (threadIdx.x == 0) n_out = 0;
  for (int i = threadIdx.x; i < kpad; i += blockDim.x) sortbuf[i] = 0ull;
  __syncthreads();
  if (cached) {
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      const unsigned long long key = norm(rk[u]);
      if (key >= prefix && key != 0ull) {
        const int p = atomicAdd(&n_out, 1);
        if (p < kpad) sortbuf[p] = rk[u];
      }
    }
  } else {
#pragma unroll 4
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
      const unsigned long long raw = raw_at(i);
      const unsigned long long key = norm(raw);
      if (key >= prefix && key != 0ull) {
        const int p = atomicAdd(&n_out, 1);
        if (p < kpad) sortbuf[p] = raw;
      }
    }
  }
  __syncthreads();
  wave0_bitonic_desc(sortbuf, kpad);  // descending; the first wave alone, no workgroup barriers inside
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const unsigned long long key = sortbuf[i];
    float v = -INFINITY;
    int32_t id = -1;
    if (i < want && key != 0ull) {
      v = fkey_inv((uint32_t)(key >> 32));
      id = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) + idx_offset;
    }
    out_val[(int64_t)q * k + i] = v;
    out_idx[(int64_t)q * k + i] = id;
  }
}

// ---- latency mode (B <= 32, the stream kernels): threshold and select SPREAD over slices (r06) -----------------------------------
// One workgroup per query left 1 .. 32 workgroups on 256 CUs for the two tails of the call (sim_threshold_kernel 22.6 us +
// topk_select_kernel 54.0 us = 29 % of a 32-query call, profiles/r05_bench_kernel_stats.csv), and most of their time was same-bin LDS
// atomics: 5 900 / 11 000 keys of one query counted into one 256-bin histogram, scores bunched just above the threshold.  Here a
// query's list is cut into NS slices, one 256-thread workgroup each: the slice radix-selects ITS top-k (the global top-k is a subset
// of the union of the slices' top-k lists), stores the keys write-through, and takes a ticket; the LAST arriver of a query loads the
// NS x k keys and selects again — then either the k-th largest score (threshold form) or sort + write (select form).  Exact, and
// deterministic: keys are distinct 64-bit values (score key : 32 | ~id : 32), a top-k set does not depend on arrival order.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility; the stream-K idiom of gemm_f32.hip): sc1 stores, every wave drains,
// the workgroup meets, one lane takes the ticket; the last arriver's lane 0 issues ONE agent-scope acquire, the workgroup meets, plain loads.
constexpr int SL_THREADS = 256, SL_RC = 8, SL_NS_MAX = 16, SL_KPAD_MAX = 128;
struct SelLds {
  int hist[256];
  int scan[256];
  int res[2];
  uint32_t red[32];
  int n_out, last;
};

// The `k` largest of the workgroup's register-held raw keys (0 = no entry; keys whose score key is below lo_floor do not count)
// -> outbuf[0 .. want), unsorted, zero-filled up to kpad.  Returns want = min(k, live keys).  topk_select_kernel's passes.
template <int RC>
__device__ __forceinline__ int block_select_keys(const unsigned long long (&rk)[RC], int k, int kpad, uint32_t lo_floor, SelLds& L,
                                                 unsigned long long* outbuf) {
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
  for (int u = 0; u < RC; ++u)
    if (rk[u] != 0ull) {
      const uint32_t key = (uint32_t)(rk[u] >> 32);
      lo = min(lo, key), hi = max(hi, key);
    }
  block_minmax(lo, hi, L.red);
  lo = max(lo, lo_floor);
  if (hi < lo) hi = lo;
  const int nbits = hi > lo ? 32 - __clz(hi - lo) : 0;
  const int lsh = 32 - nbits;
  auto norm = [&](unsigned long long raw) -> unsigned long long {
    if (raw == 0ull) return 0ull;
    const uint32_t sk = (uint32_t)(raw >> 32);
    if (sk < lo) return 0ull;
    return ((((unsigned long long)(sk - lo)) << 32) | (raw & 0xFFFFFFFFull)) << lsh;
  };
  unsigned long long prefix = 0ull;
  int need = k, want = k;
  bool exact = false;
  for (int pass = 0; pass < 8 && !exact; ++pass) {
    const int shift = 56 - 8 * pass;
    L.hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      const unsigned long long key = norm(rk[u]);
      const bool match = pass == 0 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
      if (match && key != 0ull) atomicAdd(&L.hist[(int)((key >> shift) & 255ull)], 1);
    }
    __syncthreads();
    int above;
    const int b = find_bin(L.hist, L.scan, need, &above, L.res);
    if (pass == 0) want = need;  // clamped to the live keys
    need -= above;
    prefix |= (unsigned long long)b << shift;
    exact = L.hist[b] == need;
    __syncthreads();
  }
  if (threadIdx.x == 0) L.n_out = 0;
  for (int i = threadIdx.x; i < kpad; i += SL_THREADS) outbuf[i] = 0ull;
  __syncthreads();
  if (want > 0) {
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      const unsigned long long key = norm(rk[u]);
      if (key >= prefix && key != 0ull) {
        const int p = atomicAdd(&L.n_out, 1);
        if (p < kpad) outbuf[p] = rk[u];
      }
    }
  }
  __syncthreads();
  return want;
}

struct SlicedArgs {
  const float* vals;        // [B][cap]
  const int32_t* idxs;      // [B][cap]
  int32_t* cnt;             // [B][CNT_STRIDE]: [0] list length, [1] ticket of the threshold form, [2] ticket of the select form
  int64_t cap;
  int n_slots;              // threshold form: entries [0, n_slots) are the sample scores
  int k, kpad, NS;
  int32_t idx_offset;
  float* thr;               // threshold form: out; select form: in (lower bound of the live range)
  const float* sub;         // threshold form, may be null: thr[q] = value - sub[q]
  unsigned long long* part; // [B][NS][kpad]
  float* out_val;
  int32_t* out_idx;
  int32_t* status;
};

template <bool THR>
__global__ __launch_bounds__(SL_THREADS) void sim_sliced_select_kernel(const SlicedArgs a) {
  __shared__ SelLds L;
  __shared__ __attribute__((aligned(16))) unsigned long long outbuf[SL_KPAD_MAX];
  const int q = blockIdx.x / a.NS, s = blockIdx.x % a.NS, tid = threadIdx.x;
  int count;
  if (THR) {
    count = a.n_slots;
  } else {
    const int c = a.cnt[(int64_t)q * CNT_STRIDE];
    count = c < (int)a.cap ? c : (int)a.cap;
  }
  const int per = (count + a.NS - 1) / a.NS;  // <= SL_RC * SL_THREADS by the launcher's choice of NS
  const int i0 = s * per, i1 = min(count, i0 + per);
  const float* v = a.vals + (int64_t)q * a.cap;
  const int32_t* ix = a.idxs + (int64_t)q * a.cap;
  unsigned long long rk[SL_RC];
#pragma unroll
  for (int u = 0; u < SL_RC; ++u) {
    const int i = i0 + tid + u * SL_THREADS;
    unsigned long long raw = 0ull;
    if (i < i1) {
      const float x = v[i];
      if (THR) {
        if (x > -INFINITY) raw = ((unsigned long long)fkey(x) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
      } else {
        const int32_t id = ix[i];
        if (id >= 0) raw = ((unsigned long long)fkey(x) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)id);
      }
    }
    rk[u] = raw;
  }
  const uint32_t floor_key = (!THR && a.thr) ? fkey(a.thr[q]) : 0u;
  block_select_keys<SL_RC>(rk, a.k, a.kpad, floor_key, L, outbuf);
  // publish this slice's keys write-through, then the ticket
  unsigned long long* mine = a.part + ((int64_t)q * a.NS + s) * a.kpad;
  if (tid < a.kpad) {
    const unsigned long long key = outbuf[tid];
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 w;
    w[0] = (unsigned int)(key & 0xFFFFFFFFull), w[1] = (unsigned int)(key >> 32);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(mine, 0, a.kpad * 8, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(w, dst, tid * 8, 0, 16);  // aux 16 = sc1: write-through
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int32_t* ticket = a.cnt + (int64_t)q * CNT_STRIDE + (THR ? 1 : 2);
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    L.last = t == a.NS - 1;
    if (L.last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one lane: drops this CU's stale L1 lines of `part`
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!L.last) return;  // uniform
  // ---- the last arriver of query q: NS x kpad keys -> top-k
  const unsigned long long* all = a.part + (int64_t)q * a.NS * a.kpad;
  const int n_all = a.NS * a.kpad;  // <= SL_RC * SL_THREADS
#pragma unroll
  for (int u = 0; u < SL_RC; ++u) {
    const int i = tid + u * SL_THREADS;
    rk[u] = i < n_all ? all[i] : 0ull;
  }
  const int want = block_select_keys<SL_RC>(rk, a.k, a.kpad, 0u, L, outbuf);
  if (THR) {
    // the k-th largest score = the smallest selected key's score (fewer than k live entries: the smallest of all; none: -inf)
    unsigned long long m = tid < want ? outbuf[tid] : ~0ull;  // want <= kpad <= 128 <= SL_THREADS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(m, off);
      m = o < m ? o : m;
    }
    if ((tid & 63) == 0) reinterpret_cast<unsigned long long*>(L.hist)[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
      const unsigned long long* w4 = reinterpret_cast<const unsigned long long*>(L.hist);
      unsigned long long mm = w4[0];
      for (int i = 1; i < SL_THREADS / 64; ++i) mm = w4[i] < mm ? w4[i] : mm;
      const float t = want > 0 ? fkey_inv((uint32_t)(mm >> 32)) : -INFINITY;
      a.thr[q] = a.sub ? t - a.sub[q] : t;
      a.cnt[(int64_t)q * CNT_STRIDE] = a.n_slots;  // survivors of the filter pass are appended behind the sample block
      a.cnt[(int64_t)q * CNT_STRIDE + 1] = 0;      // both tickets ready for their next use (the select form's: this call)
      a.cnt[(int64_t)q * CNT_STRIDE + 2] = 0;
    }
    return;
  }
  wave0_bitonic_desc(outbuf, a.kpad);
  __syncthreads();
  for (int i = tid; i < a.k; i += SL_THREADS) {
    const unsigned long long key = outbuf[i];
    float val = -INFINITY;
    int32_t id = -1;
    if (i < want && key != 0ull) {
      val = fkey_inv((uint32_t)(key >> 32));
      id = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) + a.idx_offset;
    }
    a.out_val[(int64_t)q * a.k + i] = val;
    a.out_idx[(int64_t)q * a.k + i] = id;
  }
  if (tid == 0) {
    const int c = a.cnt[(int64_t)q * CNT_STRIDE];
    if (a.status) a.status[q] = c > (int)a.cap ? 1 : 0;  // list overflowed: result is a subset's top-k
    a.cnt[(int64_t)q * CNT_STRIDE + 2] = 0;
  }
}

// slices for a list of up to `len` entries: ~1 024 entries (4 per thread) each, at most SL_NS_MAX; 0 = not served
static int sliced_ns(int64_t len, int kpad) {
  if (kpad > SL_KPAD_MAX || len > (int64_t)SL_NS_MAX * SL_RC * SL_THREADS) return 0;
  int ns = (int)((len + 1023) / 1024);
  const int need = (int)((len + SL_RC * SL_THREADS - 1) / (SL_RC * SL_THREADS));
  if (ns < need) ns = need;
  if (ns < 1) ns = 1;
  if (ns > SL_NS_MAX) ns = SL_NS_MAX;
  while ((int64_t)ns * kpad > SL_RC * SL_THREADS) --ns;  // the merge holds NS x kpad keys in registers
  return ns >= need ? ns : 0;
}

static int next_pow2(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

}  // namespace gdr

extern "C" size_t gdr_sim_topk_workspace_bytes(int B, int64_t N, int d, int k, int flags) {
  (void)d;
  if (B <= 0 || N <= 0 || k <= 0) return 0;
  return gdr::make_plan(B, N, k, (flags & GDR_SIM_EXHAUSTIVE) != 0).total;
}

namespace gdr {
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float4* __restrict__ in, uint2* __restrict__ out,
                                                            int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = in[i];
  union {
    __bf16 h[4];
    uint2 u;
  } o;
  o.h[0] = (__bf16)v.x, o.h[1] = (__bf16)v.y, o.h[2] = (__bf16)v.z, o.h[3] = (__bf16)v.w;  // v_cvt_pk_bf16_f32: RNE, NaN-safe
  out[i] = o.u;
}

static int sim_topk_impl(const void* Q, int B, const void* D, int64_t N, int d, int k, int32_t idx_offset, float* out_val,
                         int32_t* out_idx, int32_t* status, int flags, void* workspace, size_t workspace_bytes,
                         bool bf16, hipStream_t stream);
}  // namespace gdr

namespace gdr {
int launch_cast_f32_bf16(const float* in, void* out_bf16, int64_t n, hipStream_t stream) {
  if (n == 0) return GDR_OK;
  GDR_CHECK_ARG(in && out_bf16 && n >= 0 && n % 4 == 0, "cast: null pointer or n %% 4 != 0");
  GDR_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)out_bf16 & 7) == 0, "cast: misaligned pointer");
  if (n == 0) return GDR_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const float4*>(in), reinterpret_cast<uint2*>(out_bf16), n / 4);
  GDR_CHECK_LAUNCH("cast_f32_bf16_kernel");
  return GDR_OK;
}
}  // namespace gdr

extern "C" int gdr_cast_f32_bf16(const float* in, void* out_bf16, int64_t n, void* stream_) {
  return gdr::launch_cast_f32_bf16(in, out_bf16, n, static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_sim_topk(const float* Q, int B, const float* D, int64_t N, int d, int k, int32_t idx_offset,
                            float* out_val, int32_t* out_idx, int32_t* status, int flags, void* workspace,
                            size_t workspace_bytes, void* stream_) {
  return gdr::sim_topk_impl(Q, B, D, N, d, k, idx_offset, out_val, out_idx, status, flags, workspace, workspace_bytes,
                            false, static_cast<hipStream_t>(stream_));
}

extern "C" int gdr_sim_topk_bf16(const void* Q, int B, const void* D, int64_t N, int d, int k, int32_t idx_offset,
                                 float* out_val, int32_t* out_idx, int32_t* status, int flags, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
  return gdr::sim_topk_impl(Q, B, D, N, d, k, idx_offset, out_val, out_idx, status, flags, workspace, workspace_bytes,
                            true, static_cast<hipStream_t>(stream_));
}

int gdr::sim_topk_impl(const void* Q, int B, const void* D, int64_t N, int d, int k, int32_t idx_offset,
                       float* out_val, int32_t* out_idx, int32_t* status, int flags, void* workspace,
                       size_t workspace_bytes, bool bf16, hipStream_t stream) {
  if (B == 0) return GDR_OK;  // empty query batch
  GDR_CHECK_ARG(Q && D && out_val && out_idx && workspace, "sim_topk: null pointer");
  GDR_CHECK_ARG(B > 0 && N > 0 && d > 0 && d % (bf16 ? 8 : 4) == 0, "sim_topk: bad shape B=%d N=%lld d=%d", B,
                (long long)N, d);
  GDR_CHECK_ARG(k >= 1 && k <= 1024 && k <= N, "sim_topk: k=%d must be in [1, min(1024, N)]", k);
  GDR_CHECK_ARG(N < 0x7fffffffLL - 256, "sim_topk: shard too large for int32 doc ids");
  GDR_CHECK_ARG(((uintptr_t)Q & 15) == 0 && ((uintptr_t)D & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
                "sim_topk: Q, D must be 16-byte and workspace 256-byte aligned");
  const SimPlan p = make_plan(B, N, k, (flags & GDR_SIM_EXHAUSTIVE) != 0);
  if (workspace_bytes < p.total) {
    set_error("sim_topk: workspace %zu < required %zu", workspace_bytes, p.total);
    return GDR_ENOSPC;
  }
  char* ws = static_cast<char*>(workspace);
  SimEpilogue ep{};
  ep.cand_val = reinterpret_cast<float*>(ws + p.off_val);
  ep.cand_idx = reinterpret_cast<int32_t*>(ws + p.off_idx);
  ep.cand_cnt = reinterpret_cast<int32_t*>(ws + p.off_cnt);
  float* thr = reinterpret_cast<float*>(ws + p.off_thr);
  ep.thr = thr;
  ep.status = nullptr;  // overflow is reported per query by the select kernel (cnt > cap)
  ep.cap = (int32_t)p.cap;
  ep.tile_stride = p.stride;
  ep.mode = 1;
  const bool stream_mode = sim_stream_supported(B, d, bf16) && !(flags & GDR_SIM_NO_STREAM);
  int rc = stream_mode ? launch_sim_stream(D, N, Q, B, d, ep, bf16, stream) : launch_sim_gemm(D, N, Q, B, d, ep, bf16, stream);
  if (rc) return rc;
  const int sel_threads = 1024;  // 1024 lanes per query: measured faster than 512 with twice the entries per lane (26.7 vs 37.7 us at 32 queries)
  const int kpad = next_pow2(k);
  static const bool sliced_on = [] {
    const char* e = getenv("GDR_SIM_SLICED");  // A/B knob: 0 = one workgroup per query for the threshold / select tails (the r05 form)
    return e ? atoi(e) != 0 : true;
  }();
  // the stream kernels' sample pass zeroes the tickets: only their calls may take the sliced tails
  SlicedArgs sa{};
  sa.vals = ep.cand_val, sa.idxs = ep.cand_idx, sa.cnt = ep.cand_cnt, sa.cap = p.cap, sa.n_slots = (int)p.n_slots, sa.k = k, sa.kpad = kpad;
  sa.idx_offset = idx_offset, sa.thr = thr, sa.sub = nullptr, sa.part = reinterpret_cast<unsigned long long*>(ws + p.off_part);
  sa.out_val = out_val, sa.out_idx = out_idx, sa.status = status;
  const int ns_thr = (stream_mode && sliced_on && B <= 32) ? sliced_ns(p.n_slots, kpad) : 0;
  const int ns_sel = (stream_mode && sliced_on && B <= 32) ? sliced_ns(p.cap, kpad) : 0;
  if (ns_thr) {
    sa.NS = ns_thr;
    ProfScope prof(PROF_SELECT, 0.0, stream);
    hipLaunchKernelGGL(sim_sliced_select_kernel<true>, dim3(B * ns_thr), dim3(SL_THREADS), 0, stream, sa);
    GDR_CHECK_LAUNCH("sim_sliced_select_kernel(threshold)");
  } else {
    hipLaunchKernelGGL(sim_threshold_kernel, dim3(B), dim3(sel_threads), 0, stream, ep.cand_val, p.cap, (int)p.n_slots,
                       k, thr, ep.cand_cnt, (const float*)nullptr);
    GDR_CHECK_LAUNCH("sim_threshold_kernel");
  }
  if (p.stride > 1) {
    ep.mode = 2;
    rc = stream_mode ? launch_sim_stream(D, N, Q, B, d, ep, bf16, stream) : launch_sim_gemm(D, N, Q, B, d, ep, bf16, stream);
    if (rc) return rc;
  }
  if (ns_sel) {
    sa.NS = ns_sel;
    ProfScope prof(PROF_SELECT, 0.0, stream);
    hipLaunchKernelGGL(sim_sliced_select_kernel<false>, dim3(B * ns_sel), dim3(SL_THREADS), 0, stream, sa);
    GDR_CHECK_LAUNCH("sim_sliced_select_kernel(select)");
    return GDR_OK;
  }
  hipLaunchKernelGGL(topk_select_kernel<false>, dim3(B), dim3(sel_threads), kpad * sizeof(unsigned long long), stream,
                     ep.cand_val, ep.cand_idx, ep.cand_cnt, p.cap, 1, B, k, kpad, idx_offset, (const float*)thr, out_val,
                     out_idx, status, 0, 1);
  GDR_CHECK_LAUNCH("topk_select_kernel");
  return GDR_OK;
}

namespace gdr {
// ------------------------------------------------------------------------------------------------------------------------------
// fp32 similarity + top-k through a bf16 PRE-FILTER (r05; B > 32): the corpus-wide pass — 2·B·N·d flop, the second largest item of
// the C2 step — runs on the bf16 MFMA path over a bf16 image of the corpus, and only a few hundred docs per query are scored in
// fp32.  The result is the top-k of the FP32 scores, exactly, for every input; the bf16 pass only decides which docs get an fp32 score:
//   * s(doc) = the fp32 score the rescoring computes, s~(doc) = the bf16-operand score (products of bf16 values are exact in fp32,
//     fp32 accumulate).  bf16 keeps 8 significand bits (1 implicit + 7 stored), so round-to-nearest-even has unit roundoff u = 2^-8
//     (r05 coded 2^-9 — half the true bound; r06 fix, tests/test_gpu_prefilter.py holds a coherent-rounding input that needs it).
//     For any summation orders:
//         |s~ - s| <= ||q||·||d||·(2u + u^2 + 2·d·2^-24·1.01) =: eps(q, d) <= eps_q := ||q||·max_doc||d||·(2^-7 + 2^-16 + d·2^-22)
//   * t~_k = k-th largest s~ over all docs.  The k docs with the largest s~ have s >= t~_k - eps, so T_k (k-th largest s) >= t~_k - eps;
//     a doc of the true top-k has s >= T_k, hence s~ >= t~_k - 2·eps.  So {s~ >= t~_k - 2 eps_q} contains the true top-k (with every doc
//     tied at T_k), and the exact select over their fp32 scores (higher score, then lower id — as the fp32 path) is the brute-force result.
//   * the sample threshold L (k-th largest s~ over the sample) is <= t~_k, so the filter pass keeps s~ >= L - 2 eps_q, a superset.
// Lists that overflow (the candidate list of the bf16 pass, or more than cap2 docs inside the 2-eps band) are flagged in status[q] like
// gdr_sim_topk's; ops.sim_topk recomputes such a query on the fp32 path.
__global__ __launch_bounds__(256) void sim_qprep_kernel(const float* __restrict__ Q, int d, float dnorm_max, __bf16* __restrict__ Q16,
                                                        float* __restrict__ eps2) {
  __shared__ float red[4];
  const int q = blockIdx.x;
  const float* row = Q + (int64_t)q * d;
  float ss = 0.f;
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss))));
    union {
      __bf16 h[4];
      uint2 u;
    } o;
    o.h[0] = (__bf16)v.x, o.h[1] = (__bf16)v.y, o.h[2] = (__bf16)v.z, o.h[3] = (__bf16)v.w;
    *reinterpret_cast<uint2*>(Q16 + (int64_t)q * d + c) = o.u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nq = sqrtf(red[0] + red[1] + red[2] + red[3]) * 1.0001f;  // the norm itself is rounded: a hair of slack
    const float c = 0.0078125f + 1.52587890625e-5f + (float)d * 2.384185791015625e-7f;  // 2^-7 + 2^-16 + d * 2^-22  (u_bf16 = 2^-8)
    eps2[q] = 2.0f * nq * dnorm_max * c * 1.01f;
  }
}

// max over rows of ||D[r]||_2^2 (positive floats order like their bit patterns): one wave per row, atomicMax on the bits
__global__ __launch_bounds__(256) void row_norm2_max_kernel(const float* __restrict__ D, int64_t N, int d, unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  float best = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < N; r += (int64_t)gridDim.x * 4) {
    const float* row = D + r * d;
    float ss = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(row + c);
      ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    best = fmaxf(best, ss);
  }
  if (lane == 0) atomicMax(out, __float_as_uint(best));
}

// The tail of the pre-filter in ONE launch per call (was four: k-th largest of the whole list, gather of the band, rescoring, exact
// select — 75 us of kernels plus three launch gaps at 32 queries): a workgroup per query
//   a. reads its candidate list once (keys cached in registers), radix-selects the k-th largest bf16-operand score t~_k,
//   b. gathers the ids of the entries inside the band s~ >= t~_k - 2 eps_q into LDS,
//   c. scores them in fp32 (a wave per candidate pair: the doc row in coalesced 16-byte pieces, q in registers),
//   d. sorts (fp32 score, ~id) keys in LDS (bitonic, all waves) and writes the first k: higher score, then lower id.
template <int RC>
__global__ __launch_bounds__(1024) void prefilter_tail_kernel(const float* __restrict__ cand_val, const int32_t* __restrict__ cand_idx,
                                                              const int32_t* __restrict__ cand_cnt, int64_t cap, int k, int cap2,
                                                              int cap2p, const float* __restrict__ eps2, const float* __restrict__ Q,
                                                              const float* __restrict__ D, int d, int32_t idx_offset,
                                                              float* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                              int32_t* __restrict__ status) {
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ int res[2];
  __shared__ uint32_t red[32];
  __shared__ int n_sh;
  extern __shared__ __attribute__((aligned(16))) unsigned long long pkeys[];  // [cap2p]; the ids of the band first live in its upper half
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c_all = cand_cnt[(int64_t)q * CNT_STRIDE], count = c_all < (int)cap ? c_all : (int)cap;
  const float* v = cand_val + (int64_t)q * cap;
  const int32_t* ix = cand_idx + (int64_t)q * cap;
  const bool cached = count <= RC * 1024;
  uint32_t rk[RC];
#pragma unroll
  for (int u = 0; u < RC; ++u) {
    const int i = tid + u * 1024;
    rk[u] = (cached && i < count) ? fkey(v[i]) : 0u;  // fkey(-inf) = 0x007fffff > 0: 0 marks "no entry"
  }
  const uint32_t kneg = fkey(-INFINITY);
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  if (cached) {
#pragma unroll
    for (int u = 0; u < RC; ++u)
      if (rk[u] > kneg) lo = min(lo, rk[u]), hi = max(hi, rk[u]);
  } else {
    for (int i = tid; i < count; i += 1024) {
      const float x = v[i];
      if (x > -INFINITY) {
        const uint32_t key = fkey(x);
        lo = min(lo, key), hi = max(hi, key);
      }
    }
  }
  block_minmax(lo, hi, red);
  if (hi < lo) lo = hi = fkey(-INFINITY);
  const int nbits = hi > lo ? 32 - __clz(hi - lo) : 0;
  const int lsh = 32 - nbits;
  uint32_t prefix = 0;
  int need = k;
  for (int pass = 0; 8 * pass < nbits; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    if (cached) {
#pragma unroll
      for (int u = 0; u < RC; ++u) {
        const uint32_t raw = rk[u];
        const uint32_t key = (raw - lo) << lsh;
        const bool match = pass == 0 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
        if (raw != 0u && match && raw >= lo) atomicAdd(&hist[(key >> shift) & 255u], 1);
      }
    } else {
      for (int i = tid; i < count; i += 1024) {
        const uint32_t raw = fkey(v[i]);
        const uint32_t key = (raw - lo) << lsh;
        const bool match = pass == 0 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
        if (match && raw >= lo) atomicAdd(&hist[(key >> shift) & 255u], 1);
      }
    }
    __syncthreads();
    int above;
    const int b = find_bin(hist, scan, need, &above, res);
    need -= above;
    prefix |= (uint32_t)b << shift;
    __syncthreads();
  }
  const float band = fkey_inv(nbits ? (prefix >> lsh) + lo : lo) - eps2[q];  // t~_k - 2 eps_q
  // ---- b. the band's ids
  int32_t* ids = reinterpret_cast<int32_t*>(pkeys + cap2p / 2);  // [cap2p] ints in the upper half of the key array
  if (tid == 0) n_sh = 0;
  __syncthreads();
  if (cached) {
#pragma unroll
    for (int u = 0; u < RC; ++u) {
      const int i = tid + u * 1024;
      if (rk[u] > kneg && fkey_inv(rk[u]) >= band) {
        const int32_t id = ix[i];
        if (id >= 0) {
          const int p = atomicAdd(&n_sh, 1);
          if (p < cap2) ids[p] = id;
        }
      }
    }
  } else {
    for (int i = tid; i < count; i += 1024) {
      const int32_t id = ix[i];
      if (id >= 0 && v[i] >= band) {
        const int p = atomicAdd(&n_sh, 1);
        if (p < cap2) ids[p] = id;
      }
    }
  }
  __syncthreads();
  const int n_all = n_sh, n2 = n_all < cap2 ? n_all : cap2;
  if (tid == 0 && status) status[q] = (c_all > (int)cap || n_all > cap2) ? 1 : 0;
  // ---- c. fp32 scores of the band (the candidate's slot i keeps its id until its key is written: lane 0 of the owning wave does both)
  {
    constexpr int MAXP = 4;  // d <= 1024
    float4 qr[MAXP];
#pragma unroll
    for (int t = 0; t < MAXP; ++t) {
      const int col = lane * 4 + 256 * t;
      qr[t] = col < d ? *reinterpret_cast<const float4*>(Q + (int64_t)q * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float* vals = reinterpret_cast<float*>(pkeys);  // [cap2p] floats in the first quarter: disjoint from the ids
    for (int i0 = wave * 2; i0 < n2; i0 += 32) {
      const int i1 = i0 + 1 < n2 ? i0 + 1 : i0;
      const int64_t r0 = ids[i0], r1 = ids[i1];
      float4 a0[MAXP], a1[MAXP];
#pragma unroll
      for (int t = 0; t < MAXP; ++t) {
        const int col = lane * 4 + 256 * t;
        a0[t] = col < d ? *reinterpret_cast<const float4*>(D + r0 * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        a1[t] = col < d ? *reinterpret_cast<const float4*>(D + r1 * d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int t = 0; t < MAXP; ++t) {
        s0 = fmaf(qr[t].x, a0[t].x, s0), s0 = fmaf(qr[t].y, a0[t].y, s0), s0 = fmaf(qr[t].z, a0[t].z, s0), s0 = fmaf(qr[t].w, a0[t].w, s0);
        s1 = fmaf(qr[t].x, a1[t].x, s1), s1 = fmaf(qr[t].y, a1[t].y, s1), s1 = fmaf(qr[t].z, a1[t].z, s1), s1 = fmaf(qr[t].w, a1[t].w, s1);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s0 += __shfl_xor(s0, off), s1 += __shfl_xor(s1, off);
      if (lane == 0) {
        vals[i0] = s0;
        if (i1 != i0) vals[i1] = s1;
      }
    }
    __syncthreads();
    // ---- d. keys, sort, first k.  Keys are built from (vals, ids) through registers: the key array overlays both
    unsigned long long mykey[4];  // cap2p <= 4096
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 1024;
      mykey[u] = 0ull;
      if (i < n2) mykey[u] = ((unsigned long long)fkey(vals[i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)ids[i]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 1024;
      if (i < cap2p) pkeys[i] = mykey[u];
    }
    __syncthreads();
  }
  for (int size = 2; size <= cap2p; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (cap2p >> 1); t += 1024) {
        const int lo_i = (t / stride) * (stride << 1) + (t % stride), hi_i = lo_i + stride;
        const bool desc = ((lo_i & size) == 0);
        const unsigned long long a = pkeys[lo_i], b = pkeys[hi_i];
        if ((a < b) == desc) pkeys[lo_i] = b, pkeys[hi_i] = a;
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += 1024) {
    const unsigned long long key = pkeys[i];
    float val = -INFINITY;
    int32_t id = -1;
    if (i < n2 && key != 0ull) {
      val = fkey_inv((uint32_t)(key >> 32));
      id = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) + idx_offset;
    }
    out_val[(int64_t)q * k + i] = val;
    out_idx[(int64_t)q * k + i] = id;
  }
}

static int prefilter_cap2(int k) {
  int c = 1024;
  while (c < 4 * k) c <<= 1;
  return c;
}
struct PrefilterPlan {
  SimPlan p;
  int cap2;
  size_t off_q16, off_eps, total;
};
static PrefilterPlan make_prefilter_plan(int B, int64_t N, int d, int k) {
  PrefilterPlan pp{};
  pp.p = make_plan(B, N, k, false);
  pp.cap2 = prefilter_cap2(k);
  size_t o = pp.p.total;
  pp.off_q16 = o, o += align_up((size_t)B * d * 2, 256);
  pp.off_eps = o, o += align_up((size_t)B * 4, 256);
  pp.total = o;
  return pp;
}
}  // namespace gdr

extern "C" size_t gdr_sim_topk_prefilter_workspace_bytes(int B, int64_t N, int d, int k) {
  if (B <= 0 || N <= 0 || k <= 0 || d <= 0) return 0;
  return gdr::make_prefilter_plan(B, N, d, k).total;
}

extern "C" int gdr_row_norm2_max(const float* D, int64_t N, int d, float* out_dev, void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GDR_CHECK_ARG(D && out_dev && N > 0 && d > 0 && d % 4 == 0 && ((uintptr_t)D & 15) == 0, "row_norm2_max: bad arguments");
  if (hipMemsetAsync(out_dev, 0, 4, stream) != hipSuccess) {
    set_error("row_norm2_max: memset failed");
    return GDR_EHIP;
  }
  const int64_t blocks = (N + 3) / 4;
  hipLaunchKernelGGL(row_norm2_max_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, D, N, d,
                     reinterpret_cast<unsigned*>(out_dev));
  GDR_CHECK_LAUNCH("row_norm2_max_kernel");
  return GDR_OK;
}

extern "C" int gdr_sim_topk_prefilter(const float* Q, int B, const float* D, const void* D_bf16, float dnorm_max, int64_t N, int d, int k,
                                      int32_t idx_offset, float* out_val, int32_t* out_idx, int32_t* status, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B == 0) return GDR_OK;
  GDR_CHECK_ARG(Q && D && D_bf16 && out_val && out_idx && workspace, "sim_topk_prefilter: null pointer");
  GDR_CHECK_ARG(B > 0 && N > 0 && d > 0 && d % 8 == 0 && d <= 1024, "sim_topk_prefilter: bad shape B=%d N=%lld d=%d (d %% 8 == 0, d <= 1024)", B,
                (long long)N, d);
  GDR_CHECK_ARG(k >= 1 && k <= 1024 && k <= N, "sim_topk_prefilter: k=%d must be in [1, min(1024, N)]", k);
  GDR_CHECK_ARG(N < 0x7fffffffLL - 256, "sim_topk_prefilter: shard too large for int32 doc ids");
  GDR_CHECK_ARG(dnorm_max > 0.f && dnorm_max < INFINITY, "sim_topk_prefilter: dnorm_max must be the largest row norm of D (> 0, finite)");
  GDR_CHECK_ARG(((uintptr_t)Q & 15) == 0 && ((uintptr_t)D & 15) == 0 && ((uintptr_t)D_bf16 & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
                "sim_topk_prefilter: Q, D, D_bf16 must be 16-byte and workspace 256-byte aligned");
  const PrefilterPlan pp = make_prefilter_plan(B, N, d, k);
  const SimPlan& p = pp.p;
  if (workspace_bytes < pp.total) {
    set_error("sim_topk_prefilter: workspace %zu < required %zu", workspace_bytes, pp.total);
    return GDR_ENOSPC;
  }
  char* ws = static_cast<char*>(workspace);
  SimEpilogue ep{};
  ep.cand_val = reinterpret_cast<float*>(ws + p.off_val);
  ep.cand_idx = reinterpret_cast<int32_t*>(ws + p.off_idx);
  ep.cand_cnt = reinterpret_cast<int32_t*>(ws + p.off_cnt);
  float* thr = reinterpret_cast<float*>(ws + p.off_thr);
  ep.thr = thr, ep.status = nullptr, ep.cap = (int32_t)p.cap, ep.tile_stride = p.stride, ep.mode = 1;
  __bf16* Q16 = reinterpret_cast<__bf16*>(ws + pp.off_q16);
  float* eps2 = reinterpret_cast<float*>(ws + pp.off_eps);
  hipLaunchKernelGGL(sim_qprep_kernel, dim3(B), dim3(256), 0, stream, Q, d, dnorm_max, Q16, eps2);
  GDR_CHECK_LAUNCH("sim_qprep_kernel");
  const bool stream_mode = sim_stream_supported(B, d, true);  // B <= 32: the HBM-bound stream over the bf16 image
  int rc = stream_mode ? launch_sim_stream(D_bf16, N, Q16, B, d, ep, true, stream) : launch_sim_gemm(D_bf16, N, Q16, B, d, ep, true, stream);
  if (rc) return rc;
  // L - 2 eps: the filter pass keeps a superset of the band around the (yet unknown) k-th largest bf16 score
  hipLaunchKernelGGL(sim_threshold_kernel, dim3(B), dim3(1024), 0, stream, ep.cand_val, p.cap, (int)p.n_slots, k, thr, ep.cand_cnt,
                     (const float*)eps2);
  GDR_CHECK_LAUNCH("sim_threshold_kernel");
  if (p.stride > 1) {
    ep.mode = 2;
    rc = stream_mode ? launch_sim_stream(D_bf16, N, Q16, B, d, ep, true, stream) : launch_sim_gemm(D_bf16, N, Q16, B, d, ep, true, stream);
    if (rc) return rc;
  }
  // the tail in one launch: k-th largest bf16-operand score, the band around it, fp32 scores of the band, exact select
  int cap2p = 1024;
  while (cap2p < pp.cap2) cap2p <<= 1;
  if (int rc__ = ensure_dyn_lds(reinterpret_cast<const void*>(prefilter_tail_kernel<16>), cap2p * 8, "sim_topk_prefilter")) return rc__;
  hipLaunchKernelGGL(prefilter_tail_kernel<16>, dim3(B), dim3(1024), (size_t)cap2p * 8, stream, (const float*)ep.cand_val,
                     (const int32_t*)ep.cand_idx, (const int32_t*)ep.cand_cnt, p.cap, k, pp.cap2, cap2p, (const float*)eps2, Q, D, d,
                     idx_offset, out_val, out_idx, status);
  GDR_CHECK_LAUNCH("prefilter_tail_kernel");
  return GDR_OK;
}

extern "C" int gdr_topk_merge(const float* vals, const int32_t* idx, int G, int B, int k, float* out_val,
                              int32_t* out_idx, void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B == 0) return GDR_OK;
  GDR_CHECK_ARG(vals && idx && out_val && out_idx, "topk_merge: null pointer");
  GDR_CHECK_ARG(G > 0 && B > 0 && k >= 1 && k <= 1024, "topk_merge: bad shape G=%d B=%d k=%d", G, B, k);
  const int kpad = next_pow2(k);
  hipLaunchKernelGGL(topk_select_kernel<true>, dim3(B), dim3(SEL_THREADS), kpad * sizeof(unsigned long long), stream,
                     vals, idx, (const int32_t*)nullptr, (int64_t)0, G, B, k, kpad, 0, (const float*)nullptr, out_val, out_idx,
                     (int32_t*)nullptr, k, 1);
  GDR_CHECK_LAUNCH("topk_select_kernel<merge>");
  return GDR_OK;
}

namespace gdr {
// pairs[q][j] = {score bits, id} for j < k, pairs[q][k] = {0, status[q]}: one 8-byte-per-entry message per query row
__global__ void topk_pack_kernel(const float* __restrict__ vals, const int32_t* __restrict__ idx,
                                 const int32_t* __restrict__ status, int B, int k, int2* __restrict__ pairs) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * (k + 1)) return;
  const int q = (int)(e / (k + 1)), j = (int)(e - (int64_t)q * (k + 1));
  int2 o;
  if (j < k) {
    o.x = __float_as_int(vals[(int64_t)q * k + j]);
    o.y = idx[(int64_t)q * k + j];
  } else {
    o.x = 0;
    o.y = status ? status[q] : 0;
  }
  pairs[e] = o;
}
}  // namespace gdr

extern "C" int gdr_topk_pack(const float* vals, const int32_t* idx, const int32_t* status, int B, int k, void* pairs,
                             void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B == 0) return GDR_OK;
  GDR_CHECK_ARG(vals && idx && pairs, "topk_pack: null pointer");
  GDR_CHECK_ARG(B > 0 && k >= 1 && k <= 1024, "topk_pack: bad shape B=%d k=%d", B, k);
  const int64_t n = (int64_t)B * (k + 1);
  hipLaunchKernelGGL(topk_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, vals, idx, status, B, k,
                     static_cast<int2*>(pairs));
  GDR_CHECK_LAUNCH("topk_pack_kernel");
  return GDR_OK;
}

extern "C" int gdr_topk_merge_packed(const void* pairs, int G, int B, int k, float* out_val, int32_t* out_idx,
                                     int32_t* out_status, void* stream_) {
  using namespace gdr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (B == 0) return GDR_OK;
  GDR_CHECK_ARG(pairs && out_val && out_idx, "topk_merge_packed: null pointer");
  GDR_CHECK_ARG(G > 0 && B > 0 && k >= 1 && k <= 1024, "topk_merge_packed: bad shape G=%d B=%d k=%d", G, B, k);
  const int kpad = next_pow2(k);
  const float* vals = static_cast<const float*>(pairs);
  const int32_t* idx = static_cast<const int32_t*>(pairs) + 1;
  hipLaunchKernelGGL(topk_select_kernel<true>, dim3(B), dim3(SEL_THREADS), kpad * sizeof(unsigned long long), stream,
                     vals, idx, (const int32_t*)nullptr, (int64_t)0, G, B, k, kpad, 0, (const float*)nullptr, out_val, out_idx,
                     out_status, k + 1, 2);
  GDR_CHECK_LAUNCH("topk_select_kernel<merge packed>");
  return GDR_OK;
}
