// Evaluation of the reference's metric.py:36-68 (calc_recall_ndcg) for gfx950: per test user the scores
// against every item (metric.py:48), the training items' scores set to 0.0 (:50), the K best of a
// descending sort (:51-59, ties: lower item position first, as a stable sort gives), then recall@K
// (:5-7) and ndcg@K against the user's OWN sorted hit list (:23-34).  SURVEY.md 8f #4.
//
// Three launches, no (users x items) score matrix in memory:
//  1. eval_items_kmajor_kernel: the item rows as MFMA A fragments, [32-item tile][group of 8 k pairs][half][lane][4] -
//     one coalesced 16-byte load per lane per FOUR v_mfma_f32_32x32x2_f32 later.
//  2. eval_topk_kernel: a wavefront owns 32 users (the B operand: their rows, in registers at the reference's readout
//     width, in LDS otherwise) and a contiguous segment of item tiles; scores come out of the fp32 MFMA (an exact
//     k-ordered fmaf chain, 157 TF peak: 620 GFLOP on the amazon-book shape) with the USER on the lane, so a lane
//     compares its 16 scores of a tile with its user's current K-th best and only the rare survivors (about
//     K ln(n / K) per user over the whole sweep) are appended to the user's candidate buffer in LDS.  A full buffer is
//     pruned by the whole wavefront: entries that are training items are dropped (binary search in the user's sorted
//     list), every entry's place is counted on (score descending, position ascending), the K best stay and renew the
//     threshold.  One grid; the segments of a user block share their K-th best as they go.
//  3. eval_merge_kernel: one wavefront per user merges the segments' partial lists with the masked
//     training items - min(K, |train_u|) entries (0.0, lowest positions), which is all of them that can
//     rank - marks the hits (binary search in the sorted test list) and writes recall and ndcg in fp64.
//
// The order is total - (score descending, position ascending) - so the result does not depend on how
// items were cut into tiles and segments: bitwise reproducible, equal to a stable descending sort.
#include <functional>
#include <queue>
#include <vector>
#include "kgat_common.h"

namespace kgat {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kEvalCap = 64;      // candidate entries per user (one per lane of the pruning wavefront)
constexpr int kEvalMaxK = 32;
constexpr int kEvalTile = 32;     // items per MFMA tile
// (item tiles in flight per wavefront: two, one in the widest register form - NT in the kernel)
constexpr float kNegInf = -__builtin_inff();
constexpr int kIdxPad = 0x7fffffff;

__global__ __launch_bounds__(256) void eval_items_kmajor_kernel(int64_t n_items, int F, int FP2, int64_t n_tiles,
                                                                const float* __restrict__ emb, int64_t emb_stride,
                                                                const int32_t* __restrict__ item_ids,
                                                                float* __restrict__ itemT) {
  // one thread per element of itemT: coalesced stores, gathered 4-byte loads (a 17 MB one-off per call).
  // Layout (round 6): [tile][group of 8 k pairs][half of the group][lane][4 k pairs] - a lane's A operands of FOUR
  // consecutive MFMAs are one 16-byte load, a wavefront's one contiguous KB (it was [tile][k pair][lane]: one 4-byte
  // load per MFMA, and on this chip a vector-memory instruction costs the fp32 MFMA stream about as much issue time as
  // an MFMA does - NOTEBOOK 3.2)
  const int64_t total = n_tiles * FP2 * 64;
  for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (int64_t)gridDim.x * 256) {
    const int u4 = (int)(x & 3);
    const int lane = (int)(x >> 2 & 63);
    const int64_t tq = x >> 8;                       // (tile, group, half) = tile * (FP2 / 4) + quad of k pairs
    const int s = (int)(tq % (FP2 / 4)) * 4 + u4;
    const int64_t tile = tq / (FP2 / 4);
    const int64_t it = tile * kEvalTile + (lane & 31);
    const int k = 2 * s + (lane >> 5);
    float v = 0.f;
    if (it < n_items && k < F) v = emb[(size_t)item_ids[it] * emb_stride + k];
    itemT[x] = v;
  }
}

// A float as an unsigned integer with the same order (for atomicMax on a shared threshold) and back.
__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}

// "a ranks before b": score descending, position ascending
__device__ __forceinline__ bool ranks_before(float sa, int ia, float sb, int ib) {
  return sa > sb || (sa == sb && ia < ib);
}

// Descending bitonic sort of one (score, position) entry per lane over the wavefront.
__device__ __forceinline__ void wave_sort_desc(float& s, int& i, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float so = __shfl_xor(s, j, 64);
      const int io = __shfl_xor(i, j, 64);
      const bool lower = (lane & j) == 0;          // this lane is the lower index of its pair
      const bool desc = (lane & k) == 0;           // this block sorts descending (final k = 64: every lane)
      const bool mine_first = ranks_before(s, i, so, io);
      // the lower lane of a descending pair keeps the entry that ranks first
      const bool keep = (lower == desc) ? mine_first : !mine_first;
      s = keep ? s : so;
      i = keep ? i : io;
    }
  }
}

// position `it` in the ascending list a[lo, hi)?
__device__ __forceinline__ bool in_sorted(const int32_t* __restrict__ a, int32_t lo, int32_t hi, int32_t it) {
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    const int32_t v = a[mid];
    if (v == it) return true;
    if (v < it) lo = mid + 1; else hi = mid;
  }
  return false;
}

// tile boundaries of the grid's segments (a kernel argument): segment y sweeps tiles [b[y], b[y + 1])
constexpr int kEvalMaxLists = 66;
struct EvalBounds { int32_t b[kEvalMaxLists + 1]; };

struct EvalLds {
  // per wavefront: cand_s / cand_i [32 users][kEvalCap], kept [32]; then (LDS form only) the users' rows [FP2][64]
  static __host__ __device__ size_t per_wave_bytes(int FP2, bool rows_in_registers = false) {
    return (size_t)32 * kEvalCap * 8 + 32 * 4 * 2 + (rows_in_registers ? (size_t)0 : (size_t)FP2 * 64 * 4);
  }
};

// KG > 0 (round 6): the users' rows - the B operand of every MFMA of the sweep - live in REGISTERS (KG groups of 8
// k pairs = 8 KG values per lane: 88 at the reference readout's 176 columns) instead of LDS.  The LDS form spends
// 22.5 KB per wavefront on them, so a CU holds ONE workgroup = one wavefront per SIMD, and that wavefront's candidate
// handling (vector ALU) leaves the matrix pipe idle: the sweep ran at 0.57 of the fp32-MFMA peak, the launch at 0.30.
// With the rows in registers a wavefront needs 16.6 KB, two workgroups of four fit a CU, and a second wavefront per
// SIMD issues MFMAs while the first one filters; no ds_read per MFMA either.  The arithmetic is the same fmaf chain
// in the same k order: same score bits.  KG = 0: the LDS form (any width).
// Instantiated for KG = 11 (the 64 + 64 + 32 + 16 = 176-column readout of the reference's default model), 6 and 16: a
// width in (40, 96] / (96, 176] / (176, 256] is padded with zero columns to 96 / 176 / 256 and takes the register form
// of that size (70,679 x 24,915, K = 20: 160 columns 9.1 ms in the LDS form, 5.7 padded to 176; 88 columns 6.4 vs 3.1;
// 192 columns 17.1 ms in the LDS form).  Narrower rows are cheaper in the LDS form (16 columns: 2.2 ms), wider ones do
// not fit the registers.
// KG = 22 (352 columns: the 128 + 128 + 64 + 32 readout of the d = 128 model) keeps 176 row values per lane and ONE item
// tile in flight instead of two (a single accumulator chain per wavefront; the second wavefront of the SIMD fills in).
__host__ __device__ constexpr int eval_reg_kg(int F) {
  return F <= 40 ? 0 : (F <= 96 ? 6 : (F <= 176 ? 11 : (F <= 256 ? 16 : (F <= 352 ? 22 : 0))));
}

// (two wavefronts per SIMD at most - the LDS allows no more in either form - so the allocator may use 256 registers:
// left to its default budget it kept 150 and spilled the users' rows to scratch)
template <int NW, int KG>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2), amdgpu_num_vgpr(248))) void eval_topk_kernel(
    int64_t n_users, const int32_t* __restrict__ user_ids, int64_t n_items, int FP2, int F, EvalBounds bounds,
    const float* __restrict__ emb, int64_t emb_stride,
    const float* __restrict__ itemT, const int32_t* __restrict__ train_ptr, const int32_t* __restrict__ train_items,
    int K, float* __restrict__ part_s, int32_t* __restrict__ part_i, unsigned* __restrict__ tau_shared) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ul = lane & 31, half = lane >> 5;
  constexpr bool REG = KG > 0;
  constexpr int NT = KG > 16 ? 1 : 2;   // item tiles in flight: one where the users' rows leave no registers for two
  char* base = s_raw + (size_t)w * EvalLds::per_wave_bytes(FP2, REG);
  float* cand_s = reinterpret_cast<float*>(base);
  int32_t* cand_i = reinterpret_cast<int32_t*>(base + 32 * kEvalCap * 4);
  int32_t* kept = reinterpret_cast<int32_t*>(base + 32 * kEvalCap * 8);  // entries known not to be training items
  float* ub = reinterpret_cast<float*>(base + 32 * kEvalCap * 8 + 256);

  const int64_t u0 = ((int64_t)blockIdx.x * NW + w) * 32;      // the wavefront's 32 users (positions in user_ids)
  const int seg = blockIdx.y;
  if (u0 >= n_users) return;                                   // (no workgroup barrier anywhere below)
  const int64_t up = u0 + ul;
  const bool u_ok = up < n_users;
  const int64_t up_c = u_ok ? up : n_users - 1;
  // B fragments: lane (user ul, half) holds the user's elements k = 2s + half
  float breg[REG ? KG * 8 : 1];
  {
    const float* row = emb + (size_t)user_ids[up_c] * emb_stride;
    if constexpr (REG) {
#pragma unroll
      for (int s = 0; s < KG * 8; ++s) {
        const int k = 2 * s + half;
        breg[s] = (u_ok && k < F) ? row[k < F ? k : 0] : 0.f;
      }
    } else {
      for (int s = 0; s < FP2; ++s) {
        const int k = 2 * s + half;
        ub[s * 64 + lane] = (u_ok && k < F) ? row[k] : 0.f;
      }
    }
  }
  if (lane < 32) kept[lane] = 0;
  const int32_t tr_lo = train_ptr[up_c], tr_hi = train_ptr[up_c + 1];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // The segment's tiles (bounds, from the host's plan).  Workgroups are dispatched x-fastest: one segment row (all user
  // blocks) after the other, so the segments of a user block mostly run one after the other and a later one starts
  // from the K-th best the earlier ones have published (tau_shared below; any published value is a valid bound).
  // (Rounds 5-6 seeded the thresholds with a sample sweep over the first 512 items - first a launch of its own, 0.9 ms of
  // candidate handling with almost no MFMAs, then segment 0 of this grid; with the shared threshold it measures the same
  // with and without, 500 ... 300,000 users: profiles/r06_eval_scan.txt - and it is gone.)
  const int n_lists = gridDim.y;
  const int64_t t_lo = bounds.b[seg], t_hi = bounds.b[seg + 1];
  // the user's K-th best score so far: -inf while fewer than K are held (+inf on a lane without a user: nothing passes)
  float tau_s = u_ok ? kNegInf : __builtin_inff();
  // A user's buffer of kEvalCap entries is filled from BOTH ends: the lane of half 0 appends upwards from position 0,
  // the lane of half 1 downwards from kEvalCap - 1 - no coordination between the two per append (round 6: it was one
  // counter per user, a ballot per register and 64-bit shifts to order the two lanes behind each other: about twenty
  // vector instructions per register of a tile, all of them taken from the other wavefront's MFMAs).  wp = the lane's
  // next free entry (index into cand_s / cand_i).
  int wp = ul * kEvalCap + (half ? kEvalCap - 1 : 0);
  const int dir = half ? -1 : 1;
  // Round 6: the segments of a user's item range run side by side (one wavefront each), every one warming up its own
  // K-th best: they now SHARE it.  tau_shared[user] holds the best K-th score any segment has published (atomicMax on
  // order-preserving bits; zero-filled = below every float): K entries of some segment rank at or before it, so
  // nothing that scores below it can be among the user's K best - a valid bound whenever it is read, however stale.
  // A wavefront takes it over when it beats its own (ties on the score stay candidates: position "pad"), reads it
  // once per tile group - requested at the end of a check, used by the next - and publishes after every prune.  The
  // result is the exact top K either way; fewer candidates are appended and pruned on the way.
  float sh_next = kNegInf;
  if (tau_shared != nullptr)
    sh_next = from_ordered_bits(__hip_atomic_load(tau_shared + up_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  auto adopt_shared = [&]() {
    if (sh_next > tau_s) tau_s = sh_next;   // (ties on the score stay candidates: the filter is >=)
  };

  // Keep the K best entries of user `v` (wave-uniform), dropping training items among the new ones.  The K-th best
  // key is SELECTED, not ranked: a bisection over the 32 bits of the scores' order-preserving form - per bit one vector
  // compare against a scalar and a scalar population count - instead of one broadcast + compare + add per entry and
  // lane (round 6: the rank count was three quarters of a prune's vector instructions, and on this chip every vector
  // instruction of this wavefront is taken from the other wavefront's fp32 MFMAs; scalar instructions are not).  Entries
  // that tie with the K-th score are taken in position order (a second bisection over the positions, only when the tie
  // straddles the cut).  The survivors are compacted to the low end of the buffer in lane order - not sorted: the order
  // is total, so the SET is what matters; the merge launch sorts.
  auto prune = [&](int v) {
    const int base = v * kEvalCap, kp = kept[v];
    // entries [0, p_lo) and (p_hi, kEvalCap) of the buffer are in use (p_lo <= p_hi + 1)
    const int p_lo = __builtin_amdgcn_readlane(wp, v) - base, p_hi = __builtin_amdgcn_readlane(wp, v + 32) - base;
    const bool have = lane < p_lo || lane > p_hi;
    float s = kNegInf;
    int i = kIdxPad;
    if (have) { s = cand_s[base + lane]; i = cand_i[base + lane]; }
    const int32_t lo = __builtin_amdgcn_readlane(tr_lo, v), hi = __builtin_amdgcn_readlane(tr_hi, v);
    if (lane >= kp && have && in_sorted(train_items, lo, hi, i)) { s = kNegInf; i = kIdxPad; }
    const bool valid = i != kIdxPad;
    // the score as an unsigned key of the same order (s + 0 maps -0 to +0, equal for the reference's compare); 0 - below
    // the key of every float - for a lane without an entry
    const unsigned key = valid ? ordered_bits(s + 0.f) : 0u;
    const int n_valid = __popcll(__ballot(valid));
    const int keep = n_valid < K ? n_valid : K;
    unsigned T = 0u;   // the keep-th largest key: the largest T with at least `keep` keys >= T
    if (keep > 0) {
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned c = T | (1u << bit);
        if ((int)__popcll(__ballot(key >= c)) >= keep) T = c;
      }
    }
    const bool above = keep > 0 && key > T, tied = keep > 0 && key == T;   // (T > 0: an empty lane is neither)
    const int need_tied = keep - (int)__popcll(__ballot(above));                // >= 1 when keep > 0
    bool sel = above || tied;
    if ((int)__popcll(__ballot(tied)) > need_tied) {
      // the need_tied lowest positions among the tied entries: I = the need_tied-th smallest of them (positions are
      // distinct), found as the largest I with fewer than need_tied tied positions below it
      int I = 0;
      for (int bit = 30; bit >= 0; --bit) {
        const int c = I | (1 << bit);
        if ((int)__popcll(__ballot(tied && i < c)) < need_tied) I = c;
      }
      sel = above || (tied && i <= I);
    }
    const unsigned long long sel_mask = __ballot(sel);
    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(sel_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sel_mask, 0u));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                        // every lane has read its entry
    if (sel) { cand_s[base + pos] = s; cand_i[base + pos] = i; }
    const float ts = from_ordered_bits(T);                  // the K-th best score when keep == K
    if (ul == v) {  // both lanes of the user
      // (never below what is already known: a shared bound may be ahead of this segment's own K-th best)
      if (keep == K && ts > tau_s) tau_s = ts;
      wp = base + (half ? kEvalCap - 1 : keep);
    }
    if (lane == 0) {
      kept[v] = keep;
      if (keep == K && tau_shared != nullptr) atomicMax(tau_shared + (u0 + v), T);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // A fragments: one coalesced 16-byte load per lane per FOUR MFMAs (itemT's layout, above).  The sweep is a flat
  // sequence of steps (tile group, group of U k pairs); a step's loads are issued while the previous step's MFMAs run
  // (two register buffers, the loop unrolled by two so that no buffer is copied).
  constexpr int U = 8;
  typedef float floatx4 __attribute__((ext_vector_type(4)));
  const floatx4* a_base = reinterpret_cast<const floatx4*>(itemT) + lane;
  const int KGR = FP2 / U;                                     // k groups per tile group (FP2 is a multiple of U)
  struct Pos { int64_t t0; int g; };
  auto advance = [&](Pos& p) { if (++p.g == KGR) { p.g = 0; p.t0 += NT; } };
  auto issue = [&](float (&a)[U][NT], const Pos& p) {
    // (unconditional: a load under a branch makes the compiler's counted vmcnt waits conservative - it then waited
    // for the NEXT step's loads before this step's MFMAs; past the end the clamped tile is loaded again, unused)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t tt = p.t0 + t < t_hi ? p.t0 + t : t_hi - 1;   // (a clamped duplicate tile is ignored below)
      const floatx4* ap = a_base + ((size_t)tt * KGR + p.g) * (U / 4) * 64;
#pragma unroll
      for (int q = 0; q < U / 4; ++q) {
        const floatx4 v = ap[q * 64];
#pragma unroll
        for (int c = 0; c < 4; ++c) a[4 * q + c][t] = v[c];
      }
    }
  };
  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // acc[t][r]: user ul, item position 32 (t0 + t) + (r & 3) + 8 (r >> 2) + 4 half
  auto check = [&](int64_t t0) {
    adopt_shared();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t0 + t >= t_hi) break;  // wave-uniform
      const int ib = (int)((t0 + t) * kEvalTile) + 4 * half;
      if ((t0 + t + 1) * kEvalTile > n_items) {   // the padded end of the last tile (wave-uniform): never a candidate
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ib + (r & 3) + 8 * (r >> 2) >= n_items) acc[t][r] = __builtin_nanf("");
      }
      // the maximum of every four registers, then of all sixteen: a tile without a candidate costs the ten maxima and one
      // compare, a tile with one walks only the quads that hold it
      float mq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        mq[q] = fmaxf(fmaxf(acc[t][4 * q], acc[t][4 * q + 1]), fmaxf(acc[t][4 * q + 2], acc[t][4 * q + 3]));
      const float m = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
      if (__ballot(m >= tau_s) != 0ull) {
        // every score at or above the user's K-th best so far is appended (a tie on the score with a later position is
        // sorted out by the prune: the order there is total); after every four registers - at most four entries from
        // either end - a buffer with fewer than eight free entries is pruned
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (__ballot(mq[q] >= tau_s) == 0ull) continue;   // wave-uniform
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * q + rr;
            const float sc = acc[t][r];
            if (sc >= tau_s) {
              cand_s[wp] = sc;
              cand_i[wp] = ib + (r & 3) + 8 * (r >> 2);
              wp += dir;
            }
          }
          // both lanes of a user see both write positions: v_permlane32_swap hands every lane the lower half's value
          // and the upper half's
          const auto both = __builtin_amdgcn_permlane32_swap((unsigned)wp, (unsigned)wp, false, false);
          unsigned long long need = __ballot((int)both[1] - (int)both[0] + 1 < 8);
          if (need) {
            need &= 0xffffffffull;   // (one bit per user)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            do {
              const int v = __builtin_ctzll(need);
              need &= need - 1;
              prune(v);
            } while (need);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (tau_shared != nullptr)   // for the next tile group's check (an L2 round trip behind that group's MFMAs)
      sh_next = from_ordered_bits(__hip_atomic_load(tau_shared + up_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // (past the L1)
  };
  auto compute = [&](const float (&a)[U][NT], const Pos& p) {
    if (p.t0 >= t_hi) return;
    float b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = ub[(p.g * U + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u], acc[t], 0, 0, 0);
    if (p.g == KGR - 1) {
      check(p.t0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;   // (here, behind the check's branch: not a select per step)
    }
  };
  if constexpr (REG) {
    // Two tile groups per loop iteration = 4 KG half steps (tile group, k group, half of the k group; 44 at KG = 11) with
    // compile-time k groups - the register index of the B operand.  A half step is two 16-byte loads per lane (four k
    // pairs of both tiles) and eight MFMAs; its loads are issued THREE half steps ahead into a ring of four buffers (the
    // registers of two whole-step buffers: one step ahead was 1,024 MFMA cycles, about the L2's latency under this
    // load; three half steps are 1,536).  4 KG is a multiple of 4: the ring position of a half step is a compile-time
    // constant.  A group past the end re-reads the clamped last tile and is skipped.
    // (written out, not a loop: `#pragma unroll` over the steps was declined by the optimiser, and a generic lambda
    //  per step - the index as an integral_constant - sent every captured array to scratch)
    static_assert(KG <= 22 && U == 8, "the step list below is written out for up to 2 x 22 steps of 2 halves");
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float h0[4][NT], h1[4][NT], h2[4][NT], h3[4][NT];
    const size_t seg_bytes = (size_t)(t_hi - t_lo) * KG * 2048;
    const __amdgpu_buffer_rsrc_t seg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(itemT) + (size_t)t_lo * KG * 512, 0, (int)(unsigned)seg_bytes, 0x00020000);
    auto issue_half = [&](float (&a)[4][NT], int64_t tg, int g, int q) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int64_t tt = tg + t < t_hi ? tg + t : t_hi - 1;   // (a clamped duplicate tile is ignored below)
        // (a buffer load: the segment's fragments as a resource, the wave-uniform offset in a scalar register, the lane's
        //  16 bytes the only vector operand - as a global load every address was two 64-bit vector adds, six vector
        //  instructions per half step; the host checks that a segment's fragments stay below 4 GB)
        const unsigned soff = (unsigned)(((tt - t_lo) * KG + g) * 2 + q) * 1024u;
        const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(seg_rsrc, lane * 16, (int)soff, 0));
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c][t] = v[c];
      }
    };
    issue_half(h0, t_lo, 0, 0);
    issue_half(h1, t_lo, 0, 1);
    issue_half(h2, t_lo, 1, 0);
#define KGAT_EVAL_HALF(J, CUR, NXT)                                                                       \
    if constexpr (J < 4 * KG) {                                                                           \
      constexpr int j = J, g = (j / 2) % KG, q = j % 2, jn = j + 3, gn = (jn / 2) % KG, qn = jn % 2;      \
      const int64_t tt = t0 + (j / 2 / KG) * NT, tn = t0 + (jn / 2 / KG) * NT;                  \
      issue_half(NXT, tn, gn, qn);                                                                        \
      /* the loads go out HERE, ahead of this half step's MFMAs: left to itself the scheduler sinks them between */ \
      /* the MFMAs and waits for each a few instructions after issuing it                                         */ \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      if (tt < t_hi) { /* wave-uniform */                                                                 \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                     \
          _Pragma("unroll") for (int t = 0; t < NT; ++t)                                             \
            /* (a tile group's first MFMA adds to the constant 0: no clearing of 32 registers per group) */ \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(CUR[c][t], breg[g * U + 4 * q + c],             \
                                                          (g == 0 && q == 0 && c == 0) ? zero16 : acc[t], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (g == KG - 1 && q == 1) check(tt);                                                             \
      }                                                                                                   \
    }
#define KGAT_EVAL_HALF4(J) KGAT_EVAL_HALF(J, h0, h3) KGAT_EVAL_HALF(J + 1, h1, h0) KGAT_EVAL_HALF(J + 2, h2, h1) KGAT_EVAL_HALF(J + 3, h3, h2)
    for (int64_t t0 = t_lo; t0 < t_hi; t0 += 2 * NT) {
      KGAT_EVAL_HALF4(0) KGAT_EVAL_HALF4(4) KGAT_EVAL_HALF4(8) KGAT_EVAL_HALF4(12) KGAT_EVAL_HALF4(16) KGAT_EVAL_HALF4(20)
      KGAT_EVAL_HALF4(24) KGAT_EVAL_HALF4(28) KGAT_EVAL_HALF4(32) KGAT_EVAL_HALF4(36) KGAT_EVAL_HALF4(40) KGAT_EVAL_HALF4(44)
      KGAT_EVAL_HALF4(48) KGAT_EVAL_HALF4(52) KGAT_EVAL_HALF4(56) KGAT_EVAL_HALF4(60) KGAT_EVAL_HALF4(64) KGAT_EVAL_HALF4(68)
      KGAT_EVAL_HALF4(72) KGAT_EVAL_HALF4(76) KGAT_EVAL_HALF4(80) KGAT_EVAL_HALF4(84)   // (beyond 4 KG: compiled out)
    }
#undef KGAT_EVAL_HALF4
#undef KGAT_EVAL_HALF
  } else {
    float a0[U][NT], a1[U][NT];
    Pos p0{t_lo, 0}, p1{t_lo, 0};
    advance(p1);
    issue(a0, p0);
    while (p0.t0 < t_hi) {
      issue(a1, p1);
      compute(a0, p0);
      advance(p0); advance(p0);
      issue(a0, p0);
      compute(a1, p1);
      advance(p1); advance(p1);
    }
  }
  // the segment's list of every user: K entries, padded with (-inf, pad)
  for (int v = 0; v < 32; ++v) {
    if (u0 + v >= n_users) break;
    prune(v);
    const int n = __builtin_amdgcn_readlane(wp, v) - v * kEvalCap;   // (the kept entries, at the low end)
    if (lane < K) {
      const size_t o = ((size_t)(u0 + v) * n_lists + seg) * K + lane;
      part_s[o] = lane < n ? cand_s[v * kEvalCap + lane] : kNegInf;
      part_i[o] = lane < n ? cand_i[v * kEvalCap + lane] : kIdxPad;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void eval_merge_kernel(
    int64_t n_users, int n_seg, int K, const float* __restrict__ part_s, const int32_t* __restrict__ part_i,
    const int32_t* __restrict__ train_ptr, const int32_t* __restrict__ train_items,
    const int32_t* __restrict__ test_ptr, const int32_t* __restrict__ test_items, const double* __restrict__ disc,
    double* __restrict__ recall, double* __restrict__ ndcg, int32_t* __restrict__ topk) {
  const int lane = threadIdx.x & 63;
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= n_users) return;
  float s = kNegInf;
  int i = kIdxPad;
  // the masked training items (metric.py:50): score 0.0; only the K lowest positions can rank
  const int32_t tr_lo = train_ptr[u], tr_hi = train_ptr[u + 1];
  if (lane < K && tr_lo + lane < tr_hi) { s = 0.f; i = train_items[tr_lo + lane]; }
  for (int g = 0; g < n_seg; ++g) {
    if (lane >= 32 && lane - 32 < K) {
      const size_t o = ((size_t)u * n_seg + g) * K + (lane - 32);
      s = part_s[o];
      i = part_i[o];
    }
    wave_sort_desc(s, i, lane);  // the K best so far in lanes [0, K), K <= 32
  }
  if (n_seg == 0) wave_sort_desc(s, i, lane);
  const int32_t te_lo = test_ptr[u], te_hi = test_ptr[u + 1];
  const bool hit = lane < K && i != kIdxPad && in_sorted(test_items, te_lo, te_hi, i);
  const unsigned long long hits = __ballot(hit);
  if (topk && lane < K) topk[(size_t)u * K + lane] = i == kIdxPad ? -1 : i;
  if (lane == 0) {
    const int nh = __popcll(hits);
    const int n_pos = te_hi - te_lo;
    double dcg = 0.0, ideal = 0.0;
    for (int k = 0; k < K; ++k)
      if (hits >> k & 1ull) dcg += disc[k];
    for (int k = 0; k < nh; ++k) ideal += disc[k];
    recall[u] = n_pos > 0 ? (double)nh / (double)n_pos : 0.0;
    ndcg[u] = ideal > 0.0 ? dcg / ideal : 0.0;
  }
}

// k pairs of a row, padded with zeros: to the register form's size where there is one, else to the unroll of the MFMA loop
static int eval_fp2(int F) { return eval_reg_kg(F) > 0 ? eval_reg_kg(F) * 8 : ((F + 1) / 2 + 7) / 8 * 8; }

static int eval_waves_per_block(int F) {
  const int FP2 = eval_fp2(F);
  if (eval_reg_kg(F) > 0) return 4;
  // the largest workgroup whose wavefronts' LDS (candidates + the users' rows) fits a CU
  for (int nw = 4; nw >= 1; nw >>= 1)
    if (EvalLds::per_wave_bytes(FP2) * nw <= (size_t)160 * 1024) return nw;
  return 0;
}

// Launch plan.  One grid: the item tiles are split into segments so that the grid has a few workgroups per CU.
//
// Segment sizes (round 6).  A CU holds `slots` workgroups at a time and the hardware hands the grid out in order, one
// segment row (all user blocks) after the other: with equal segments the reference's shape is 2,212 workgroups on
// 512 slots - 4.3 rounds, the fifth a third full.  The plan therefore also considers rows whose LAST segments are
// shorter (the stragglers of the last round are short ones) and one or two more rows than the minimum, simulates the
// in-order hand-out of each candidate (a workgroup costs its tiles plus a fixed share for its prologue, closing prunes
// and list) and keeps the shortest.  (Worth 2 % on the chip, not the 11 % of the simulation: a CU left with one
// workgroup runs it faster.)
constexpr double kEvalFixedCost = 0.02;   // a workgroup's fixed work, in units of one user block's whole sweep
struct EvalPlanH { int nw, seg, n_lists; EvalBounds bounds; };

static double eval_makespan(int64_t blocks, int64_t slots, const double* frac, int n) {
  // in-order list scheduling on `slots` identical slots: a min-heap of the slots' finishing times
  std::priority_queue<double, std::vector<double>, std::greater<double>> h;
  for (int64_t i = 0; i < slots; ++i) h.push(0.0);
  double last = 0.0;
  for (int r = 0; r < n; ++r) {
    const double cost = frac[r] + kEvalFixedCost;
    for (int64_t x = 0; x < blocks; ++x) {
      const double t = h.top() + cost;
      h.pop();
      h.push(t);
      if (t > last) last = t;
    }
  }
  return last;
}

static EvalPlanH eval_plan(int64_t n_users, int64_t n_items, int F) {
  EvalPlanH p;
  p.nw = eval_waves_per_block(F);
  const int64_t n_tiles = (n_items + kEvalTile - 1) / kEvalTile;
  const int64_t rest = n_tiles;
  const int nw = p.nw > 0 ? p.nw : 1;
  const int64_t blocks = (n_users + 32 * nw - 1) / (32 * nw);
  const bool reg = eval_reg_kg(F) > 0;
  const int64_t slots = (int64_t)device_cu_count() * (reg ? 2 : 1);   // resident workgroups (LDS bound)
  // about two rounds of resident workgroups (measured against one and four at the shapes of the reference's three
  // datasets and at 8,000 users: profiles/r06_eval_scan.txt - every row more is another prologue, warm-up and list
  // per user, one round leaves the last workgroups of an uneven grid alone on the chip)
  const int64_t want = slots * 2;
  int64_t seg = (want + blocks - 1) / (blocks > 0 ? blocks : 1);
  const int64_t max_seg = rest / 16 > 0 ? rest / 16 : 1;     // at least 16 tiles (512 items) per segment
  if (seg > max_seg) seg = max_seg;
  if (seg > 64) seg = 64;
  if (seg < 1) seg = 1;
  // candidates: seg .. seg + 2 rows, equal or with a short tail (weights 1, .., 1, 1/2, 1/4)
  double best = -1.0, best_frac[kEvalMaxLists];
  int best_n = (int)seg;
  for (int i = 0; i < best_n; ++i) best_frac[i] = 1.0 / best_n;
  if (seg > 1 && blocks <= 16 * slots) {   // (one row, or a grid of many rounds: nothing to gain)
    for (int n = (int)seg; n <= (int)seg + 2 && n <= max_seg && n <= 64; ++n)
      for (int tail = 0; tail <= 1; ++tail) {
        if (tail && n < 3) continue;
        double w[kEvalMaxLists], sum = 0.0;
        for (int i = 0; i < n; ++i) { w[i] = !tail || i < n - 2 ? 1.0 : (i == n - 2 ? 0.5 : 0.25); sum += w[i]; }
        bool ok = true;
        for (int i = 0; i < n; ++i) { w[i] /= sum; ok = ok && w[i] * rest >= 8.0; }
        if (!ok) continue;
        const double mk = eval_makespan(blocks, slots, w, n);
        if (best < 0.0 || mk < best) { best = mk; best_n = n; for (int i = 0; i < n; ++i) best_frac[i] = w[i]; }
      }
  }
  p.seg = best_n;
  p.n_lists = p.seg;
  int y = 0;
  p.bounds.b[0] = 0;
  double acc = 0.0;
  for (int i = 0; i < p.seg; ++i) {
    acc += best_frac[i];
    int64_t e = i == p.seg - 1 ? n_tiles : (int64_t)(acc * (double)rest + 0.5);
    if (e <= p.bounds.b[y]) e = p.bounds.b[y] + 1;                      // (never empty; rest >= seg)
    if (e > n_tiles - (p.seg - 1 - i)) e = n_tiles - (p.seg - 1 - i);
    p.bounds.b[++y] = (int32_t)e;
  }
  for (int i = y + 1; i <= kEvalMaxLists; ++i) p.bounds.b[i] = (int32_t)n_tiles;
  return p;
}

}  // namespace kgat

using namespace kgat;

extern "C" {

int kgat_eval_supported(int F, int K) {
  return F >= 1 && K >= 1 && K <= kEvalMaxK && eval_waves_per_block(F) > 0;
}

int64_t kgat_eval_items_elems(int64_t n_items, int F) {
  if (n_items < 0 || F < 1) return 0;
  const int64_t n_tiles = (n_items + kEvalTile - 1) / kEvalTile;
  return n_tiles * eval_fp2(F) * 64;
}

int kgat_eval_items_kmajor_f32(int64_t n_items, int F, const float* emb, int64_t emb_stride, const int32_t* item_ids,
                               float* itemT, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_items >= 0 && F >= 1 && emb_stride >= F, "eval_items_kmajor: bad sizes");
  if (n_items == 0) return KGAT_OK;
  KGAT_CHECK_ARG(emb && item_ids && itemT, "eval_items_kmajor: null pointer");
  const int FP2 = eval_fp2(F);
  const int64_t n_tiles = (n_items + kEvalTile - 1) / kEvalTile;
  const int64_t total = n_tiles * FP2 * 64;
  int64_t grid = (total + 255) / 256;
  if (grid > 65536) grid = 65536;
  hipLaunchKernelGGL(eval_items_kmajor_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), n_items, F, FP2,
                     n_tiles, emb, emb_stride, item_ids, itemT);
  KGAT_CHECK_LAUNCH("eval_items_kmajor");
  return KGAT_OK;
}

size_t kgat_eval_workspace_bytes(int64_t n_users, int64_t n_items, int F, int K) {
  if (n_users <= 0 || n_items <= 0 || !kgat_eval_supported(F, K)) return 256;
  const EvalPlanH pl = eval_plan(n_users, n_items, F);
  return 2 * align_up((size_t)n_users * pl.n_lists * K * 4, 256) + align_up((size_t)n_users * 4, 256) + 256;
}

int kgat_eval_recall_ndcg_f32(int64_t n_users, const int32_t* user_ids, int64_t n_items, int F, const float* emb,
                              int64_t emb_stride, const float* itemT, const int32_t* train_ptr,
                              const int32_t* train_items, const int32_t* test_ptr, const int32_t* test_items, int K,
                              const double* disc, void* workspace, size_t workspace_bytes, double* recall_out,
                              double* ndcg_out, int32_t* topk_out, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_users >= 0 && n_items >= 0 && F >= 1 && emb_stride >= F, "eval_recall_ndcg: bad sizes");
  if (!kgat_eval_supported(F, K)) {
    set_error("eval_recall_ndcg: K = %d (1..%d) or F = %d not supported", K, kEvalMaxK, F);
    return KGAT_E_UNSUPPORTED;
  }
  if (n_users == 0) return KGAT_OK;
  KGAT_CHECK_ARG(n_items >= K, "eval_recall_ndcg: fewer items (%lld) than K (%d): the reference indexes rank K - 1",
                 (long long)n_items, K);
  KGAT_CHECK_ARG(user_ids && emb && itemT && train_ptr && test_ptr && disc && recall_out && ndcg_out && workspace,
                 "eval_recall_ndcg: null pointer");
  if (workspace_bytes < kgat_eval_workspace_bytes(n_users, n_items, F, K)) {
    set_error("eval_recall_ndcg: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  const int FP2 = eval_fp2(F);
  const EvalPlanH pl = eval_plan(n_users, n_items, F);
  const int nw = pl.nw;
  for (int y = 0; y < pl.n_lists; ++y)   // (the kernel addresses a segment's fragments with 32-bit offsets)
    KGAT_CHECK_ARG((int64_t)(pl.bounds.b[y + 1] - pl.bounds.b[y]) * FP2 * 256 < ((int64_t)1 << 32),
                   "eval_recall_ndcg: a segment of %d tiles is beyond 4 GB of fragments", pl.bounds.b[y + 1] - pl.bounds.b[y]);
  Carver cv(workspace);
  float* part_s = cv.take<float>((size_t)n_users * pl.n_lists * K);
  int32_t* part_i = cv.take<int32_t>((size_t)n_users * pl.n_lists * K);
  unsigned* tau_shared = cv.take<unsigned>((size_t)n_users);   // shared K-th best per user (order-preserving bits)
  const int kg = eval_reg_kg(F);
  const bool reg = kg > 0;
  const size_t lds = EvalLds::per_wave_bytes(FP2, reg) * nw;
  const unsigned gx = (unsigned)((n_users + 32 * nw - 1) / (32 * nw));
  hipStream_t st = as_stream(stream);
#define KGAT_EVAL_LAUNCH(NW, KG_)                                                                                     \
  do {                                                                                                                \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(eval_topk_kernel<NW, KG_>),                                 \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {                    \
      set_error("eval_recall_ndcg: cannot reserve %zu bytes of LDS", lds);                                            \
      return KGAT_E_HIP;                                                                                              \
    }                                                                                                                 \
    hipLaunchKernelGGL((eval_topk_kernel<NW, KG_>), dim3(gx, (unsigned)pl.n_lists), dim3(NW * 64), lds, st, n_users,  \
                       user_ids, n_items, FP2, F, pl.bounds, emb, emb_stride, itemT,                                  \
                       train_ptr, train_items, K, part_s, part_i, pl.n_lists > 1 ? tau_shared : nullptr);             \
  } while (0)
  if (pl.n_lists > 1 && hipMemsetAsync(tau_shared, 0, (size_t)n_users * 4, st) != hipSuccess) {
    set_error("eval_recall_ndcg: cannot clear the shared thresholds");
    return KGAT_E_HIP;
  }
  // ONE launch over (user blocks) x (item segments)
  if (kg == 6) KGAT_EVAL_LAUNCH(4, 6);
  else if (kg == 11) KGAT_EVAL_LAUNCH(4, 11);
  else if (kg == 16) KGAT_EVAL_LAUNCH(4, 16);
  else if (kg == 22) KGAT_EVAL_LAUNCH(4, 22);
  else if (nw == 4) KGAT_EVAL_LAUNCH(4, 0);
  else if (nw == 2) KGAT_EVAL_LAUNCH(2, 0);
  else KGAT_EVAL_LAUNCH(1, 0);
#undef KGAT_EVAL_LAUNCH
  KGAT_CHECK_LAUNCH("eval_topk");
  hipLaunchKernelGGL(eval_merge_kernel, dim3((unsigned)((n_users + 3) / 4)), dim3(256), 0, st, n_users, pl.n_lists, K,
                     part_s, part_i, train_ptr, train_items, test_ptr, test_items, disc, recall_out, ndcg_out,
                     topk_out);
  KGAT_CHECK_LAUNCH("eval_merge");
  return KGAT_OK;
}

}  // extern "C"
