// Edge softmax over the incoming edges of each destination, for gfx950.  Row A3 of
// SURVEY.md 8a.  Replaces dgl.nn.pytorch.softmax.edge_softmax (call site reference
// models.py:153; DGL 0.4.x runs it as copy_reduce(max) / sub / exp / copy_reduce(sum) / div,
// five E-sized temporaries):
//   a[e] = exp(s[e] - M[dst e]) / Z[dst e],  M[v] = max_{e->v} s[e],  Z[v] = sum_{e->v} exp(s[e]-M[v])
//
// Design (HBM bound; algorithmic bytes 12E + 4N): ONE sweep over the destination-sorted edge
// array plus a short second launch for the rows the sweep's ranges cut (default,
// kgat_edge_softmax_f32: softmax_local_kernel + softmax_cut_rows_kernel, described above those
// kernels).  No atomics, fixed combination order, no bound on a row's length.
//
// The operator's first implementation - three streaming passes (integer-ordered atomic row max;
// row sums in 2^-40 fixed point with 64-bit integer atomic adds, associative and therefore order
// independent; normalise), one lane per CSR position with segmented wave scans - is kept below as
// kgat_edge_softmax_3pass_f32: an independently written second implementation the tests
// cross-check the sweep against.
#include <math.h>

#include "kgat_common.h"

// Cache policy of the sweep's streams: the position map (read once per step) as non-temporal loads - step 0.4178 ->
// 0.4132 ms, it no longer displaces reused rows from the Infinity Cache.  Measured and not taken
// (profiles/r04_step_ab_cache_policy.txt): the logits too (SLOWER, 0.4251 - they are gathered through the map and
// their lines are hit several times), the destination rows.
#define SM_LD_LOGIT(p) (*(p))
#define SM_LD_MAP(p) __builtin_nontemporal_load(p)
namespace kgat {

constexpr float kFixScale = 1099511627776.0f;       // 2^40
constexpr float kFixInv = 1.0f / 1099511627776.0f;  // 2^-40

__global__ void softmax_init_kernel(int64_t n, float* __restrict__ M,
                                    unsigned long long* __restrict__ Z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    M[i] = -INFINITY;
    Z[i] = 0ull;
  }
}

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  // total order of IEEE floats through their integer images (no NaN inputs expected)
  if (v == 0.f) v = 0.f;  // -0 -> +0: the integer image of -0 would sort below every negative
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// Lanes of one destination form a contiguous segment of the wavefront (edges are sorted by
// destination).  Segment bounds come from one neighbour exchange + one ballot; the reductions
// below then need only the value shuffles (6 steps), not a second shuffle of the row ids.
struct WaveSeg {
  int start;  // first lane of this lane's segment
  bool last;  // this lane is the segment's last lane
};
__device__ __forceinline__ WaveSeg wave_segment(int32_t r, int lane) {
  const int32_t rprev = __shfl_up(r, 1, kWave);
  const unsigned long long heads = __ballot(lane == 0 || rprev != r);
  const unsigned long long upto = (lane == kWave - 1) ? ~0ull : ((2ull << lane) - 1ull);  // bits 0..lane
  WaveSeg s;
  s.start = 63 - __clzll(heads & upto);
  s.last = (lane == kWave - 1) || ((heads >> (lane + 1)) & 1ull);
  return s;
}

// Every wavefront handles kItems consecutive 64-edge chunks; the loads of all chunks are issued
// before the first reduction so that the two dependent memory latencies of a chunk (row id ->
// row statistic) are paid once per wavefront, not once per chunk.
constexpr int kItems = 4;
constexpr int kBlockEdges = 256 * kItems;

__device__ __forceinline__ int64_t chunk_pos(int64_t e0, int k) {
  // block tile = 4 waves x kItems chunks of 64 edges; wave w owns chunks [w*kItems, (w+1)*kItems)
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  return e0 + (int64_t)blockIdx.x * kBlockEdges + (int64_t)(wave * kItems + k) * kWave + lane;
}

template <bool IN_CSR>
__global__ __launch_bounds__(256) void softmax_max_kernel(int64_t e0, int64_t e1,
                                                          const int32_t* __restrict__ row_of,
                                                          const int32_t* __restrict__ eid,
                                                          const float* __restrict__ logits,
                                                          float* __restrict__ M) {
  const int lane = threadIdx.x & (kWave - 1);
  int32_t r[kItems];
  float v[kItems];
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t p = chunk_pos(e0, k);
    const bool valid = p < e1;
    r[k] = valid ? row_of[p] : -1;
    v[k] = valid ? (IN_CSR ? logits[p] : logits[eid[p]]) : -INFINITY;
  }
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const WaveSeg sg = wave_segment(r[k], lane);
    float x = v[k];
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const float up = __shfl_up(x, off, kWave);
      if (lane - off >= sg.start) x = fmaxf(x, up);
    }
    if (r[k] >= 0 && sg.last) atomic_max_f32(&M[r[k]], x);
  }
}

template <bool IN_CSR>
__global__ __launch_bounds__(256) void softmax_sum_kernel(int64_t e0, int64_t e1,
                                                          const int32_t* __restrict__ row_of,
                                                          const int32_t* __restrict__ eid,
                                                          const float* __restrict__ logits,
                                                          const float* __restrict__ M,
                                                          unsigned long long* __restrict__ Z) {
  const int lane = threadIdx.x & (kWave - 1);
  int32_t r[kItems];
  float s[kItems], m[kItems];
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int64_t p = chunk_pos(e0, k);
    const bool valid = p < e1;
    r[k] = valid ? row_of[p] : -1;
    s[k] = valid ? (IN_CSR ? logits[p] : logits[eid[p]]) : 0.f;
  }
#pragma unroll
  for (int k = 0; k < kItems; ++k) m[k] = r[k] >= 0 ? M[r[k]] : 0.f;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    float x = r[k] >= 0 ? expf(s[k] - m[k]) : 0.f;
    const WaveSeg sg = wave_segment(r[k], lane);
    // in-wave partial in fp32 (fixed scan order, <= 64 terms each <= 1); only the cross-wave
    // combination has to be order independent, so the partial is added in 2^-40 fixed point
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const float up = __shfl_up(x, off, kWave);
      if (lane - off >= sg.start) x += up;
    }
    if (r[k] >= 0 && sg.last) atomicAdd(&Z[r[k]], __float2ull_rn(x * kFixScale));
  }
}

template <bool IN_CSR>
__global__ __launch_bounds__(256) void softmax_norm_kernel(
    int64_t e0, int64_t e1, const int32_t* __restrict__ row_of, const int32_t* __restrict__ eid,
    const float* __restrict__ logits, const float* __restrict__ M,
    const unsigned long long* __restrict__ Z, float* __restrict__ out,
    float* __restrict__ out_csr) {
  int32_t r[kItems], e[kItems];
  float s[kItems], m[kItems];
  unsigned long long z[kItems];
  int64_t p[kItems];
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    p[k] = chunk_pos(e0, k);
    const bool valid = p[k] < e1;
    r[k] = valid ? row_of[p[k]] : -1;
    e[k] = valid ? (eid ? eid[p[k]] : (int32_t)p[k]) : 0;
  }
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    s[k] = r[k] >= 0 ? (IN_CSR ? logits[p[k]] : logits[e[k]]) : 0.f;
    m[k] = r[k] >= 0 ? M[r[k]] : 0.f;
    z[k] = r[k] >= 0 ? Z[r[k]] : 1ull;
  }
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    if (r[k] < 0) continue;
    const float a = expf(s[k] - m[k]) / ((float)z[k] * kFixInv);
    if (out_csr) out_csr[p[k]] = a;
    if (out) out[e[k]] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// One-sweep form (default).  A wavefront takes 512 (1,024) consecutive CSR positions, 8 (16) per lane in
// BLOCKED order - lane l holds positions 8l .. 8l+7, so row ids, index map, CSR-ordered logits and results
// move as 16-byte loads / stores and nothing is transposed - and finishes every destination row that lies
// completely inside its range:
//   * inside a lane, runs of equal row id are reduced with forward sweeps over the registers (static
//     indexing) and the finished values travel back with a copy sweep;
//   * across lanes, the row of a lane's LAST run is a segmented inclusive scan over the lanes (DPP row
//     shifts inside a 16-lane row, row broadcasts across rows; which lane may take which neighbour is
//     decided once from a ballot of the segment heads and reused by both passes), and the total of a row
//     is fetched from the lane in which the row ends with ONE ds_bpermute (that lane again from a ballot);
//   * two passes of that: the row maxima first, so that every position is exponentiated ONCE against the
//     maximum of its row inside the range (no rescaling of partial sums, no exp inside a scan), then the
//     row sums;
//   * the row cut by the start / the end of the wavefront's range is not finished here: its partial
//     (m, s) goes to a carry entry together with the indices of the first and the last wavefront the row
//     touches (from indptr), and its positions are stored PROVISIONALLY as exp(x - m).
//     softmax_cut_rows_kernel then lets every wavefront combine, for each of its (at most two) cut rows,
//     the carries of the row's whole chain - head entry first, then the followers in blocks of 64 with an
//     ordered tree, the same sequence in every wavefront of the chain, hence the same bits - and rescale
//     its own provisional positions of that row by exp(m - M) / S.  It reads neither logits nor indices.
// No atomics, fixed combination order: bitwise reproducible, and no bound on a row's length.
// Round 3 (second half): the sweep had been ~1,200 instructions per wavefront (two LDS transposes, 64-bit
// address arithmetic per load, thirteen ds_bpermute scan steps carrying (m, s) pairs with an exp in every
// combine, a compare pair per stored position) at 7 wavefronts per SIMD - bound by instruction issue.  This
// form is ~350.
// positions per lane: 16 (1,024 per wavefront), or 8 for launches over fewer than kSmSmallEdges
// positions, where the sweep is a latency chain rather than a stream and twice the wavefronts with half
// the chain each finish sooner.
constexpr int kSmMaxEPL = 16;
constexpr int64_t kSmSmallEdges = 8 << 20;
template <int EPL> struct SmGeom {
  static constexpr int EPW = kWave * EPL;   // positions per wavefront
};
constexpr float kSmNegBig = -3.0e38f;     // stands in for -inf (keeps the combine NaN-free)

struct __attribute__((aligned(16))) SmCarry {
  int32_t row;    // -1: no entry
  int32_t count;  // positions of this wavefront that belong to the row
  float m, s;     // partial max, partial sum of exp(x - m)
  int32_t head;   // wavefront in which the row starts (its `b` entry heads the chain)
  int32_t last;   // wavefront in which the row ends
  int32_t pad0, pad1;
};

// exp(x) for x <= 0 on the hardware exp2: the product x*log2(e) is split into its rounded value
// and the rounding remainder (fma), the remainder enters as a first-order factor - about 1.5 ulp,
// against ~|x| ulp for exp2(x * log2e) alone.  Results below the normal range flush to zero.
__device__ __forceinline__ float sm_exp(float x) {
  x = fmaxf(x, -128.0f);  // exp2(-184) is already 0; keeps the split finite for the -3e38 stand-in
  const float t = x * 1.44269504088896341f;
  const float lo = fmaf(x, 1.44269504088896341f, -t) + x * 1.92596299112661746e-8f;
  return __builtin_amdgcn_exp2f(t) * fmaf(lo, 0.693147180559945309f, 1.0f);
}

// (m, s) <- (m, s) (+) (m2, s2): the smaller maximum's sum is rescaled by exp(-|m - m2|).  Branch-free
// (one exp, selects).  Used where partial results of different wavefronts meet (the chains of cut rows).
__device__ __forceinline__ void sm_combine(float& m, float& s, float m2, float s2) {
  const bool keep = m >= m2;
  const float e = sm_exp(keep ? m2 - m : m - m2);
  const float big = keep ? s : s2, small = keep ? s2 : s;
  s = fmaf(small, e, big);
  m = keep ? m : m2;
}

// index of the wavefront in the launch, as a scalar (the compiler cannot see that threadIdx.x / 64 is
// uniform; with it in an SGPR the range base is one, and loads take the base + 32-bit lane offset form)
__device__ __forceinline__ int64_t sm_wave_index() {
  return (int64_t)blockIdx.x * (256 / kWave) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
}

// DPP lane moves (gfx9 controls): row_shr:n = 0x110 + n, wave_shl:1 = 0x130, wave_shr:1 = 0x138,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143.  A lane without a valid source keeps `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sm_dpp(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int32_t sm_dpp_i(int32_t old, int32_t src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false);
}

// Lane predicates are kept as 64-bit lane masks in scalar registers (ballot -> empty asm that pins the value
// -> inverse ballot at the use): left to itself the compiler re-issued the compare in front of every select
// (a VALU write of an SGPR pair costs two wait states before a VALU may read it) and turned chains of
// selects over the same predicates into indexed-choice code.
typedef unsigned long long sm_mask;
__device__ __forceinline__ sm_mask sm_pin(bool c) {
  sm_mask m = __builtin_amdgcn_ballot_w64(c);
  asm volatile("" : "+s"(m));
  return m;
}
__device__ __forceinline__ float sm_sel(sm_mask m, float a, float b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b; }
// max without the quieting v_max x, x the compiler puts in front of every fmaxf operand it did not produce
// itself (loads, selects).  Plain VALU asm; its results reach DPP moves only through selects.
__device__ __forceinline__ float sm_vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// the value lane `l` holds, as a scalar (the builtin moves integers: bit casts, not conversions)
__device__ __forceinline__ float sm_lane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// How the runs at the two ends of a lane's block of positions continue into the neighbouring lanes; computed
// once per wavefront, used by the maximum pass and by the sum pass.
struct SmLinks {
  sm_mask multi;                   // the block holds more than one row (first run != last run)
  sm_mask link, link_n;            // first run continues lane-1's last run / last run continues into lane+1
  sm_mask p1, p2, p4, p8, p15, p31;  // scan steps in which the lane takes the value offered from its left
  int zaddr;                       // 4 x the lane in which the row of the lane's FIRST run ends
};

__device__ __forceinline__ SmLinks sm_links(int32_t key_f, int32_t key_l, int lane) {
  SmLinks k;
  const bool multi = key_f != key_l;
  const int32_t prev_l = sm_dpp_i<0x138>(key_l, key_l), next_f = sm_dpp_i<0x130>(key_f, key_f);
  const bool link = lane > 0 && prev_l == key_f;
  const bool link_n = lane < kWave - 1 && key_l == next_f;
  // the scan runs over the lanes' LAST runs; a lane whose last run starts inside it (multi) or does not
  // continue its left neighbour's heads a segment
  const unsigned long long heads = __ballot(multi || !link);
  const unsigned long long upto = (lane == kWave - 1) ? ~0ull : ((2ull << lane) - 1ull);  // bits 0..lane
  const int start = 63 - __clzll(heads & upto);
  const int in_row = lane & 15;
  k.multi = sm_pin(multi);
  k.link = sm_pin(link);
  k.link_n = sm_pin(link_n);
  k.p1 = sm_pin(in_row >= 1 && lane - 1 >= start);
  k.p2 = sm_pin(in_row >= 2 && lane - 2 >= start);
  k.p4 = sm_pin(in_row >= 4 && lane - 4 >= start);
  k.p8 = sm_pin(in_row >= 8 && lane - 8 >= start);
  k.p15 = sm_pin((lane & 16) != 0 && start < (lane & 48));   // rows 1 and 3 take lane 15 / 47
  k.p31 = sm_pin(lane >= 32 && start < 32);                  // rows 2 and 3 take lane 31
  // a row handed from lane to lane ends in the first lane at or after this one that holds a second run
  // (the row ends inside it, as its first run) or whose run does not continue (the row ends with it)
  const unsigned long long stops = __ballot(multi || !link_n);  // (lane 63 always stops)
  k.zaddr = 4 * (lane + (int)__builtin_ctzll(stops >> lane));
  return k;
}

template <bool IS_MAX>
__device__ __forceinline__ float sm_op(float left, float right) { return IS_MAX ? fmaxf(left, right) : left + right; }

// a_f / a_l: aggregate of the lane's first / last run (the same number when the lane holds one run).
// t_f / t_l: aggregate of those runs' ROWS over the whole range of the wavefront; incl: the inclusive scan
// value (the lane's last row from its start inside the range to the end of the lane).
template <bool IS_MAX>
__device__ __forceinline__ void sm_rows_across_lanes(const SmLinks& k, float a_f, float a_l, float& t_f, float& t_l,
                                                     float& incl) {
  float v = a_l, t;
  t = sm_dpp<0x111, 0xf>(v, v); v = sm_sel(k.p1, sm_op<IS_MAX>(t, v), v);
  t = sm_dpp<0x112, 0xf>(v, v); v = sm_sel(k.p2, sm_op<IS_MAX>(t, v), v);
  t = sm_dpp<0x114, 0xf>(v, v); v = sm_sel(k.p4, sm_op<IS_MAX>(t, v), v);
  t = sm_dpp<0x118, 0xf>(v, v); v = sm_sel(k.p8, sm_op<IS_MAX>(t, v), v);
  t = sm_dpp<0x142, 0xa>(v, v); v = sm_sel(k.p15, sm_op<IS_MAX>(t, v), v);
  t = sm_dpp<0x143, 0xc>(v, v); v = sm_sel(k.p31, sm_op<IS_MAX>(t, v), v);
  incl = v;
  const float prev = sm_dpp<0x138, 0xf>(v, v);                          // lane-1's scan value
  const float fin_f = sm_sel(k.link, sm_op<IS_MAX>(prev, a_f), a_f);   // the first run's row, if it ends in this lane
  const float own = sm_sel(k.multi, fin_f, v);                          // what this lane can offer as a finished total
  const float e = __int_as_float(__builtin_amdgcn_ds_bpermute(k.zaddr, __float_as_int(own)));
  const float nxt = sm_dpp<0x130, 0xf>(e, e);                           // lane+1's finished first-run row
  t_f = e;
  t_l = sm_sel(k.multi, sm_sel(k.link_n, nxt, v), e);
}

template <bool IN_CSR, int kSmEPL, bool FAST>
__device__ __forceinline__ void sm_sweep(int64_t e0, int64_t e1, int64_t w, int64_t base, int64_t end,
                                         const int32_t* __restrict__ indptr, const int32_t* __restrict__ row_of,
                                         const int32_t* __restrict__ eid, const float* __restrict__ logits,
                                         float* __restrict__ out, float* __restrict__ out_csr,
                                         SmCarry* __restrict__ carry) {
  constexpr int kSmEPW = SmGeom<kSmEPL>::EPW;
  const int lane = threadIdx.x % kWave;
  const int n_valid = (int)(end - base);
  const int32_t* ro = row_of + base;
  // ---- loads: lane l takes positions kSmEPL * l .. + kSmEPL - 1 of the range.  FAST (a full range,
  // 16-byte aligned): dwordx4.  Otherwise per position, clamped to the last valid one (whose row the
  // positions past the end then repeat, with a huge negative logit) - nothing under a per-position branch.
  int32_t r[kSmEPL], gi[kSmEPL];
  float x[kSmEPL];
  const bool need_gi = !IN_CSR || out != nullptr;
  if (FAST) {
#pragma unroll
    for (int v = 0; v < kSmEPL / 4; ++v) {
      const int4 b = *reinterpret_cast<const int4*>(ro + lane * kSmEPL + 4 * v);
      r[4 * v] = b.x; r[4 * v + 1] = b.y; r[4 * v + 2] = b.z; r[4 * v + 3] = b.w;
    }
    if (need_gi) {
      const int32_t* ei = eid + base;
#pragma unroll
      for (int v = 0; v < kSmEPL / 4; ++v) {
        typedef int i4v __attribute__((ext_vector_type(4)));
        const i4v b = __builtin_nontemporal_load(reinterpret_cast<const i4v*>(ei + lane * kSmEPL + 4 * v));
        gi[4 * v] = b.x; gi[4 * v + 1] = b.y; gi[4 * v + 2] = b.z; gi[4 * v + 3] = b.w;
      }
    }
    if (IN_CSR) {
      const float* lg = logits + base;
#pragma unroll
      for (int v = 0; v < kSmEPL / 4; ++v) {
        const float4 a = *reinterpret_cast<const float4*>(lg + lane * kSmEPL + 4 * v);
        x[4 * v] = a.x; x[4 * v + 1] = a.y; x[4 * v + 2] = a.z; x[4 * v + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < kSmEPL; ++i) x[i] = SM_LD_LOGIT(logits + gi[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < kSmEPL; ++i) {
      const int q = lane * kSmEPL + i;
      const int qc = q < n_valid ? q : n_valid - 1;
      r[i] = ro[qc];
      gi[i] = need_gi ? SM_LD_MAP(eid + base + qc) : 0;
    }
#pragma unroll
    for (int i = 0; i < kSmEPL; ++i) {
      const int q = lane * kSmEPL + i;
      const int qc = q < n_valid ? q : n_valid - 1;
      const float v = IN_CSR ? SM_LD_LOGIT(logits + base + qc) : SM_LD_LOGIT(logits + gi[i]);
      x[i] = q < n_valid ? v : kSmNegBig;
    }
  }
  // ---- the rows at the two ends of the range: cut by it?
  const int32_t R0 = __builtin_amdgcn_readfirstlane(r[0]);
  const int32_t RL = __builtin_amdgcn_readlane(r[kSmEPL - 1], kWave - 1);
  const bool cut_start = base > e0 && row_of[base - 1] == R0;
  const bool cut_end = end < e1 && row_of[end] == RL;
  // ---- runs inside the lane: eq[i] = position i continues the run of position i-1
  sm_mask eq[kSmEPL];
  eq[0] = 0;
#pragma unroll
  for (int i = 1; i < kSmEPL; ++i) eq[i] = sm_pin(r[i] == r[i - 1]);
  const int32_t key_f = r[0], key_l = r[kSmEPL - 1];
  const SmLinks k = sm_links(key_f, key_l, lane);
  // ---- pass 1: row maxima.  Forward max sweep, backward copy sweep (every position then holds the maximum
  // of its run; [0] / [last] those of the lane's first / last run), the rows of those two runs across the
  // lanes, and the two results handed back through their runs by one more copy sweep each way.
  float M[kSmEPL];
  M[0] = x[0];
#pragma unroll
  for (int i = 1; i < kSmEPL; ++i) M[i] = sm_sel(eq[i], sm_vmax(M[i - 1], x[i]), x[i]);
#pragma unroll
  for (int i = kSmEPL - 2; i >= 0; --i) M[i] = sm_sel(eq[i + 1], M[i + 1], M[i]);
  float mf, ml, mi;
  sm_rows_across_lanes<true>(k, M[0], M[kSmEPL - 1], mf, ml, mi);
  M[kSmEPL - 1] = ml;
#pragma unroll
  for (int i = kSmEPL - 2; i >= 0; --i) M[i] = sm_sel(eq[i + 1], M[i + 1], M[i]);
  M[0] = mf;
#pragma unroll
  for (int i = 1; i < kSmEPL; ++i) M[i] = sm_sel(eq[i], M[i - 1], M[i]);
  // ---- pass 2: x <- exp(x - row max inside the range), row sums the same way
#pragma unroll
  for (int i = 0; i < kSmEPL; ++i) x[i] = sm_exp(x[i] - M[i]);
  float S[kSmEPL];
  S[0] = x[0];
#pragma unroll
  for (int i = 1; i < kSmEPL; ++i) S[i] = sm_sel(eq[i], S[i - 1] + x[i], x[i]);
#pragma unroll
  for (int i = kSmEPL - 2; i >= 0; --i) S[i] = sm_sel(eq[i + 1], S[i + 1], S[i]);
  float sf, sl, si;
  sm_rows_across_lanes<false>(k, S[0], S[kSmEPL - 1], sf, sl, si);
  // ---- carries of the cut rows (first row within the range: lane 0's finished first-run row; last row
  // within the range: lane 63's scan value); their positions inside the range follow from indptr
  if (cut_start || cut_end) {  // (wave-uniform)
    const float t0m = sm_lane_f(mf, 0), t0s = sm_lane_f(sf, 0);
    const float tlm = sm_lane_f(mi, kWave - 1), tls = sm_lane_f(si, kWave - 1);
    SmCarry a, b;
    a.row = b.row = -1;
    a.count = b.count = 0; a.m = b.m = kSmNegBig; a.s = b.s = 0.f;
    a.head = b.head = a.last = b.last = 0; a.pad0 = a.pad1 = b.pad0 = b.pad1 = 0;
    if (cut_start) {  // a follower entry of the first row's chain (the row may also fill the whole range)
      const int64_t first_beg = indptr[R0], first_end = indptr[R0 + 1];
      const int64_t fb = first_beg > e0 ? first_beg : e0, fe = first_end < e1 ? first_end : e1;
      a.row = R0; a.count = (int)((first_end < end ? first_end : end) - base); a.m = t0m; a.s = t0s;
      a.head = (int32_t)((fb - e0) / kSmEPW); a.last = (int32_t)((fe - 1 - e0) / kSmEPW);
    }
    if (cut_end && !(cut_start && R0 == RL)) {  // the head entry of the last row's chain
      const int64_t last_beg = indptr[RL], last_end = indptr[RL + 1];
      const int64_t lb = last_beg > e0 ? last_beg : e0, le = last_end < e1 ? last_end : e1;
      b.row = RL; b.count = (int)(end - (last_beg > base ? last_beg : base)); b.m = tlm; b.s = tls;
      b.head = (int32_t)((lb - e0) / kSmEPW); b.last = (int32_t)((le - 1 - e0) / kSmEPW);
    }
    if (lane == 0) {
      carry[2 * w] = a;
      carry[2 * w + 1] = b;
    }
    // positions of a cut row leave as exp(x - m), m the carry's: a "sum" of one
    const bool of = (cut_start && key_f == R0) || (cut_end && key_f == RL);
    const bool ol = (cut_start && key_l == R0) || (cut_end && key_l == RL);
    sf = of ? 1.0f : sf;
    sl = ol ? 1.0f : sl;
  } else if (lane == 0) {
    SmCarry a;
    a.row = -1; a.count = 0; a.m = kSmNegBig; a.s = 0.f; a.head = a.last = 0; a.pad0 = a.pad1 = 0;
    carry[2 * w] = a;
    carry[2 * w + 1] = a;
  }
  S[kSmEPL - 1] = sl;
#pragma unroll
  for (int i = kSmEPL - 2; i >= 0; --i) S[i] = sm_sel(eq[i + 1], S[i + 1], S[i]);
  S[0] = sf;
#pragma unroll
  for (int i = 1; i < kSmEPL; ++i) S[i] = sm_sel(eq[i], S[i - 1], S[i]);
  // ---- normalise (hardware reciprocal, 1 ulp) and store
#pragma unroll
  for (int i = 0; i < kSmEPL; ++i) x[i] *= __builtin_amdgcn_rcpf(S[i]);
  if (FAST) {
    if (out_csr) {
      float* oc = out_csr + base;
#pragma unroll
      for (int v = 0; v < kSmEPL / 4; ++v) {
        float4 a;
        a.x = x[4 * v]; a.y = x[4 * v + 1]; a.z = x[4 * v + 2]; a.w = x[4 * v + 3];
        *reinterpret_cast<float4*>(oc + lane * kSmEPL + 4 * v) = a;
      }
    }
    if (out) {
#pragma unroll
      for (int i = 0; i < kSmEPL; ++i) out[gi[i]] = x[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < kSmEPL; ++i) {
      const int q = lane * kSmEPL + i;
      if (q < n_valid) {
        if (out_csr) out_csr[base + q] = x[i];
        if (out) out[gi[i]] = x[i];
      }
    }
  }
}

template <bool IN_CSR, int kSmEPL>
__global__ __launch_bounds__(256) void softmax_local_kernel(
    int64_t e0, int64_t e1, int aligned, const int32_t* __restrict__ indptr, const int32_t* __restrict__ row_of,
    const int32_t* __restrict__ eid, const float* __restrict__ logits, float* __restrict__ out,
    float* __restrict__ out_csr, SmCarry* __restrict__ carry) {
  constexpr int kSmEPW = SmGeom<kSmEPL>::EPW;
  const int64_t w = sm_wave_index();
  const int64_t base = e0 + w * kSmEPW;
  if (base >= e1) return;
  const int64_t end = base + kSmEPW < e1 ? base + kSmEPW : e1;
  if (aligned && end - base == kSmEPW)
    sm_sweep<IN_CSR, kSmEPL, true>(e0, e1, w, base, end, indptr, row_of, eid, logits, out, out_csr, carry);
  else
    sm_sweep<IN_CSR, kSmEPL, false>(e0, e1, w, base, end, indptr, row_of, eid, logits, out, out_csr, carry);
}

// (M, S) of a cut row from the carries of its chain: the head wavefront's `b` entry, then the `a`
// entries of the followers head+1 .. last in blocks of 64 (ordered tree inside a block).  Every
// wavefront of the chain runs exactly this sequence, so all of them normalise with the same bits.
__device__ __forceinline__ void sm_chain_total(const SmCarry* __restrict__ carry, int32_t head, int32_t last,
                                               int lane, float& M, float& S) {
  const SmCarry h = carry[2 * (int64_t)head + 1];
  float m = h.m, s = h.s;
  for (int64_t j0 = (int64_t)head + 1; j0 <= last; j0 += kWave) {
    const int64_t j = j0 + lane;
    float pm = kSmNegBig, ps = 0.f;
    if (j <= last) {
      const SmCarry f = carry[2 * j];
      pm = f.m; ps = f.s;
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const float m2 = __shfl_down(pm, d, kWave), s2 = __shfl_down(ps, d, kWave);
      if ((lane & (2 * d - 1)) == 0) sm_combine(pm, ps, m2, s2);
    }
    pm = __shfl(pm, 0, kWave); ps = __shfl(ps, 0, kWave);
    sm_combine(m, s, pm, ps);
  }
  M = m; S = s;
}

// provisional values of positions [lo, hi) of one row times f, 8 x 64 per step so that the loads of a
// step are in flight together
__device__ __forceinline__ void sm_rescale_span(int64_t lo, int64_t hi, int lane, float f,
                                                const int32_t* __restrict__ eid, float* __restrict__ out,
                                                float* __restrict__ out_csr) {
  constexpr int U = 8;
  for (int64_t p0 = lo; p0 < hi; p0 += U * kWave) {
    int64_t e[U];
    float x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * kWave + lane;
      const int64_t pc = p < hi ? p : hi - 1;
      e[u] = out ? (int64_t)eid[pc] : pc;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * kWave + lane;
      const int64_t pc = p < hi ? p : hi - 1;
      x[u] = out_csr ? out_csr[pc] : out[e[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * kWave + lane;
      if (p < hi) {
        const float a = x[u] * f;
        if (out_csr) out_csr[p] = a;
        if (out) out[e[u]] = a;
      }
    }
  }
}

// Second launch: the rows the ranges cut.  The sweep left their positions as exp(x - m) with m the
// range's partial maximum of the row (the carry's); here they are multiplied by exp(m - M) / S of the
// whole row.  The common chain is two ranges long (a row crossing one boundary): its two entries are
// this wavefront's own and its neighbour's, requested together with the first / last 64 provisional
// values of the range - one memory round trip; longer chains (hubs) walk their entries.
template <int kSmEPL>
__global__ __launch_bounds__(256) void softmax_cut_rows_kernel(int64_t e0, int64_t e1,
                                                               const int32_t* __restrict__ eid,
                                                               float* __restrict__ out, float* __restrict__ out_csr,
                                                               const SmCarry* __restrict__ carry) {
  constexpr int kSmEPW = SmGeom<kSmEPL>::EPW;
  const int lane = threadIdx.x % kWave;
  const int64_t w = sm_wave_index();
  const int64_t base = e0 + w * kSmEPW;
  if (base >= e1) return;
  const int64_t end = base + kSmEPW < e1 ? base + kSmEPW : e1;
  const int64_t n_waves = (e1 - e0 + kSmEPW - 1) / kSmEPW;
  const SmCarry ca = carry[2 * w], cb = carry[2 * w + 1];
  const SmCarry pv = carry[2 * (w > 0 ? w - 1 : 0) + 1], nx = carry[2 * (w + 1 < n_waves ? w + 1 : w)];
  const int64_t pa = base + lane < end ? base + lane : end - 1;
  const int64_t pb = end - 1 - lane >= base ? end - 1 - lane : base;
  const int64_t ea = out ? (int64_t)eid[pa] : pa, eb = out ? (int64_t)eid[pb] : pb;
  const float xa = out_csr ? out_csr[pa] : out[ea], xb = out_csr ? out_csr[pb] : out[eb];
  if (ca.row >= 0) {
    float M, S;
    if (ca.head == w - 1 && ca.last == w) {  // (same operands, same order as the walk)
      M = pv.m; S = pv.s;
      sm_combine(M, S, ca.m, ca.s);
    } else {
      sm_chain_total(carry, ca.head, ca.last, lane, M, S);
    }
    const float f = sm_exp(ca.m - M) / S;
    if (lane < ca.count) {
      const float a = xa * f;
      if (out_csr) out_csr[pa] = a;
      if (out) out[ea] = a;
    }
    sm_rescale_span(base + kWave, base + ca.count, lane, f, eid, out, out_csr);
  }
  if (cb.row >= 0) {  // (the head entry of its chain: the row starts in this range)
    float M, S;
    if (cb.last == w + 1) {
      M = cb.m; S = cb.s;
      sm_combine(M, S, nx.m, nx.s);
    } else {
      sm_chain_total(carry, cb.head, cb.last, lane, M, S);
    }
    const float f = sm_exp(cb.m - M) / S;
    if (lane < cb.count) {
      const float a = xb * f;
      if (out_csr) out_csr[pb] = a;
      if (out) out[eb] = a;
    }
    sm_rescale_span(end - cb.count, end - kWave, lane, f, eid, out, out_csr);
  }
}

// Backward (DGL 0.4.x EdgeSoftmax.backward): grad_s = a*g - a * sum_row(a*g).
// Row sums of a*g are unbounded floats, so this path uses one subgroup per row in CSR order
// (fixed summation order); it is not on the reference's training path (attention is computed
// under no_grad, kgat.py:142-144) and is provided for operator completeness.
template <bool HAS_EID>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int32_t n_rows, int32_t row0,
                                                          const int32_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ eid,
                                                          const float* __restrict__ a,
                                                          const float* __restrict__ g,
                                                          float* __restrict__ gs) {
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int64_t v = (int64_t)blockIdx.x * (256 / kWave) + wave;
  if (v >= n_rows) return;
  const int32_t beg = indptr[row0 + v], end = indptr[row0 + v + 1];
  float acc = 0.f;
  for (int32_t p = beg + lane; p < end; p += kWave) {
    const int32_t e = HAS_EID ? eid[p] : p;
    acc = fmaf(a[e], g[e], acc);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  for (int32_t p = beg + lane; p < end; p += kWave) {
    const int32_t e = HAS_EID ? eid[p] : p;
    gs[e] = a[e] * g[e] - a[e] * acc;
  }
}

}  // namespace kgat

using namespace kgat;

template <bool IN_CSR, int EPL>
static void launch_softmax_sweep(int64_t e_begin, int64_t e_end, const int32_t* indptr, const int32_t* row_of,
                                 const int32_t* eid, const float* logits, float* out, float* out_csr, SmCarry* carry,
                                 hipStream_t st) {
  const int64_t n_waves = (e_end - e_begin + SmGeom<EPL>::EPW - 1) / SmGeom<EPL>::EPW;
  const unsigned blocks = (unsigned)((n_waves + 3) / 4);
  // the 16-byte loads / stores of the sweep's full ranges need every array it touches that way aligned
  const uintptr_t bits = (uintptr_t)row_of | (uintptr_t)eid | (uintptr_t)out_csr | (IN_CSR ? (uintptr_t)logits : 0);
  const int aligned = (bits % 16 == 0 && e_begin % 4 == 0) ? 1 : 0;
  hipLaunchKernelGGL((softmax_local_kernel<IN_CSR, EPL>), dim3(blocks), dim3(256), 0, st, e_begin, e_end, aligned, indptr,
                     row_of, eid, logits, out, out_csr, carry);
  hipLaunchKernelGGL((softmax_cut_rows_kernel<EPL>), dim3(blocks), dim3(256), 0, st, e_begin, e_end, eid, out, out_csr,
                     (const SmCarry*)carry);
}

extern "C" {

size_t kgat_edge_softmax_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  (void)n_nodes;
  constexpr int kMinEPW = SmGeom<8>::EPW;  // the shortest range any launch uses: the most carry entries
  const size_t n_waves = (size_t)((n_edges > 0 ? n_edges : 0) + kMinEPW - 1) / kMinEPW;
  return align_up((2 * n_waves + 2) * sizeof(SmCarry), 256);
}

int kgat_edge_softmax_f32(int64_t n_nodes, int64_t e_begin, int64_t e_end, const int32_t* indptr,
                          const int32_t* row_of, const int32_t* eid, const float* logits,
                          int logits_in_csr_order, float* out, float* out_csr, void* workspace,
                          size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && e_begin >= 0 && e_end >= e_begin && e_end < INT32_MAX,
                 "edge_softmax: bad size");
  if (e_end == e_begin) return KGAT_OK;
  KGAT_CHECK_ARG(indptr && row_of && logits && workspace, "edge_softmax: null pointer");
  KGAT_CHECK_ARG(out || out_csr, "edge_softmax: no output requested");
  KGAT_CHECK_ARG(eid != nullptr || (logits_in_csr_order && out == nullptr),
                 "edge_softmax: edge-id ordered input/output needs eid");
  const int64_t ne = e_end - e_begin;
  if (workspace_bytes < kgat_edge_softmax_workspace_bytes(n_nodes, ne)) {
    set_error("edge_softmax: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  SmCarry* carry = static_cast<SmCarry*>(workspace);
  const bool small = ne < kSmSmallEdges;
  if (logits_in_csr_order) {
    if (small) launch_softmax_sweep<true, 8>(e_begin, e_end, indptr, row_of, eid, logits, out, out_csr, carry, st);
    else launch_softmax_sweep<true, 16>(e_begin, e_end, indptr, row_of, eid, logits, out, out_csr, carry, st);
  } else {
    if (small) launch_softmax_sweep<false, 8>(e_begin, e_end, indptr, row_of, eid, logits, out, out_csr, carry, st);
    else launch_softmax_sweep<false, 16>(e_begin, e_end, indptr, row_of, eid, logits, out, out_csr, carry, st);
  }
  KGAT_CHECK_LAUNCH("edge_softmax");
  return KGAT_OK;
}

// The three-pass form (atomic row max / fixed-point atomic row sum / normalise) this file started
// with; kept as an independent implementation for A/B measurements and cross-checks in the tests.
size_t kgat_edge_softmax_3pass_workspace_bytes(int64_t n_nodes) {
  const size_t n = (size_t)(n_nodes > 0 ? n_nodes : 1);
  return align_up(n * sizeof(float), 256) + align_up(n * sizeof(unsigned long long), 256);
}

int kgat_edge_softmax_3pass_f32(int64_t n_nodes, int64_t e_begin, int64_t e_end,
                                const int32_t* row_of, const int32_t* eid, const float* logits,
                                int logits_in_csr_order, float* out, float* out_csr, void* workspace,
                                size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && e_begin >= 0 && e_end >= e_begin && e_end < INT32_MAX,
                 "edge_softmax_3pass: bad size");
  if (e_end == e_begin) return KGAT_OK;
  KGAT_CHECK_ARG(row_of && logits && workspace, "edge_softmax_3pass: null pointer");
  KGAT_CHECK_ARG(out || out_csr, "edge_softmax_3pass: no output requested");
  KGAT_CHECK_ARG(eid != nullptr || (logits_in_csr_order && out == nullptr),
                 "edge_softmax_3pass: edge-id ordered input/output needs eid");
  if (workspace_bytes < kgat_edge_softmax_3pass_workspace_bytes(n_nodes)) {
    set_error("edge_softmax_3pass: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver cv(workspace);
  float* M = cv.take<float>((size_t)n_nodes);
  unsigned long long* Z = cv.take<unsigned long long>((size_t)n_nodes);
  const int64_t ne = e_end - e_begin;
  const unsigned eb = (unsigned)((ne + kBlockEdges - 1) / kBlockEdges);
  hipLaunchKernelGGL(softmax_init_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0,
                     st, n_nodes, M, Z);
  KGAT_CHECK_LAUNCH("softmax_init");
  if (logits_in_csr_order) {
    hipLaunchKernelGGL(softmax_max_kernel<true>, dim3(eb), dim3(256), 0, st, e_begin, e_end, row_of,
                       eid, logits, M);
    hipLaunchKernelGGL(softmax_sum_kernel<true>, dim3(eb), dim3(256), 0, st, e_begin, e_end, row_of,
                       eid, logits, (const float*)M, Z);
    hipLaunchKernelGGL(softmax_norm_kernel<true>, dim3(eb), dim3(256), 0, st, e_begin, e_end,
                       row_of, eid, logits, (const float*)M, (const unsigned long long*)Z, out,
                       out_csr);
  } else {
    hipLaunchKernelGGL(softmax_max_kernel<false>, dim3(eb), dim3(256), 0, st, e_begin, e_end,
                       row_of, eid, logits, M);
    hipLaunchKernelGGL(softmax_sum_kernel<false>, dim3(eb), dim3(256), 0, st, e_begin, e_end,
                       row_of, eid, logits, (const float*)M, Z);
    hipLaunchKernelGGL(softmax_norm_kernel<false>, dim3(eb), dim3(256), 0, st, e_begin, e_end,
                       row_of, eid, logits, (const float*)M, (const unsigned long long*)Z, out,
                       out_csr);
  }
  KGAT_CHECK_LAUNCH("edge_softmax_3pass");
  return KGAT_OK;
}

int kgat_edge_softmax_bwd_f32(int64_t n_rows, int64_t row0, const int32_t* indptr,
                              const int32_t* eid, const float* a, const float* grad_a,
                              float* grad_logits, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && row0 >= 0 && row0 + n_rows < INT32_MAX, "edge_softmax_bwd: bad size");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(indptr && a && grad_a && grad_logits, "edge_softmax_bwd: null pointer");
  const unsigned blocks = (unsigned)((n_rows + 3) / 4);
  if (eid)
    hipLaunchKernelGGL(softmax_bwd_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream),
                       (int32_t)n_rows, (int32_t)row0, indptr, eid, a, grad_a, grad_logits);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream),
                       (int32_t)n_rows, (int32_t)row0, indptr, eid, a, grad_a, grad_logits);
  KGAT_CHECK_LAUNCH("edge_softmax_bwd");
  return KGAT_OK;
}

}  // extern "C"
