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
// One-sweep form (default).  A wavefront takes 1,024 consecutive CSR positions, 16 per lane
// (coalesced loads, transposed through a wave-private LDS patch), and finishes every destination
// row that lies completely inside its range:
//   * inside a lane, runs of equal row id are reduced with forward/backward sweeps over the 16
//     registers (static indexing): run max, then run sum of exp(x - max);
//   * across lanes, the aggregate (m, s) of a row spanning several lanes is a segmented
//     forward scan (combine: s <- s1 e^{m1-m} + s2 e^{m2-m}) and the finished total travels
//     back to the lanes of the row with a segmented copy scan, so that all positions of a row
//     are normalised with the same (M, S);
//   * the row cut by the start / the end of the wavefront's range is not finished here: its
//     partial (m, s) goes to a carry entry together with the indices of the first and the last
//     wavefront the row touches (from indptr).  softmax_cut_rows_kernel then lets every
//     wavefront combine, for each of its (at most two) cut rows, the carries of the row's whole
//     chain - head entry first, then the followers in blocks of 64 with an ordered tree, the same
//     sequence in every wavefront of the chain, hence the same bits - and normalise its own
//     positions of that row.  (Until round 2 a separate one-wave-per-boundary chain launch wrote
//     the totals back into the entries; three dependent launches of 5-23 us were latency.)
// No atomics, fixed combination order: bitwise reproducible, and no bound on a row's length.
// LDS: one 5 KB patch per wavefront, used for the logits, then the row ids, then the results
// (40 KB per workgroup with two patches limited a CU to 3 workgroups = 768 of the benchmark
// graph's 895: a second, almost empty round doubled the kernel's time).
// positions per lane: 16 (1,024 per wavefront), or 8 for launches over fewer than kSmSmallEdges
// positions, where the sweep is a latency chain (load, transpose, scans, store) rather than a
// stream and twice the wavefronts with half the chain each finish sooner (amazon-book graph: sweep
// 22 -> 17 us, cut rows 13 -> 12 us; 4 per lane was slower again: 36 us for both)
constexpr int kSmMaxEPL = 16;
#ifndef KGAT_SM_SMALL_EDGES
#define KGAT_SM_SMALL_EDGES (8 << 20)
#endif
constexpr int64_t kSmSmallEdges = KGAT_SM_SMALL_EDGES;  // (a macro for A/B builds: scripts/micro/softmax_ab.py)
template <int EPL> struct SmGeom {
  static constexpr int EPW = kWave * EPL;   // positions per wavefront
  static constexpr int Pad = EPL + 4;       // floats per lane in the LDS patch (conflict-free b128 for 8 and 16)
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
// (one exp, selects): the scans below call it in every lane at every step, and a divergent if / else
// cost two exec-mask switches and both exp paths per call.  Same arithmetic as the branchy form.
__device__ __forceinline__ void sm_combine(float& m, float& s, float m2, float s2) {
  const bool keep = m >= m2;
  const float e = sm_exp(keep ? m2 - m : m - m2);
  const float big = keep ? s : s2, small = keep ? s2 : s;
  s = fmaf(small, e, big);
  m = keep ? m : m2;
}

__device__ __forceinline__ int64_t sm_wave_index() {
  return (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
}

__device__ __forceinline__ void sm_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <bool IN_CSR, int kSmEPL>
__global__ __launch_bounds__(256) void softmax_local_kernel(
    int64_t e0, int64_t e1, const int32_t* __restrict__ indptr, const int32_t* __restrict__ row_of,
    const int32_t* __restrict__ eid, const float* __restrict__ logits, float* __restrict__ out,
    float* __restrict__ out_csr, SmCarry* __restrict__ carry) {
  constexpr int kSmEPW = SmGeom<kSmEPL>::EPW, kSmPad = SmGeom<kSmEPL>::Pad;
  __shared__ __attribute__((aligned(16))) float s_patch[256 / kWave][kWave * kSmPad];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int64_t w = sm_wave_index();
  const int64_t base = e0 + w * kSmEPW;
  if (base >= e1) return;
  const int64_t end = base + kSmEPW < e1 ? base + kSmEPW : e1;
  float* px = s_patch[wv];
  int32_t* pr = reinterpret_cast<int32_t*>(s_patch[wv]);
  // striped, coalesced loads; positions past `end` repeat the last row with a huge negative logit.
  // Branch-free (clamped positions instead of predicated loads): all row ids and input indices are
  // requested together, then all logits - with a load under a condition per position the compiler
  // waited for every index before it issued the dependent gather, sixteen serial round trips per
  // wavefront on the indexed input path (round 3: that path now carries the grouped-order logits).
  const int32_t r_last = row_of[end - 1];
  int32_t rs[kSmEPL], gi[kSmEPL];
  float xs[kSmEPL];
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    const int64_t p = base + j * kWave + lane;
    const int64_t pc = p < end ? p : end - 1;
    rs[j] = row_of[pc];  // (= r_last past the end)
    gi[j] = IN_CSR ? (int32_t)(pc - e0) : eid[pc];
  }
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    const int64_t p = base + j * kWave + lane;
    const float x = IN_CSR ? logits[p < end ? p : end - 1] : logits[gi[j]];
    xs[j] = p < end ? x : kSmNegBig;
  }
  // neighbours of the range (is the first / last row cut?) and the extent of the two rows at its ends
  const int32_t r_first = row_of[base];
  const bool cut_start = base > e0 && row_of[base - 1] == r_first;
  const bool cut_end = end < e1 && row_of[end] == r_last;
  const int64_t first_beg = indptr[r_first], first_end = indptr[r_first + 1];
  const int64_t last_beg = indptr[r_last], last_end = indptr[r_last + 1];
  float x[kSmEPL], M[kSmEPL], S[kSmEPL];
  int32_t r[kSmEPL];
  // transpose through the wave's patch: the logits, then (same patch) the row ids
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    const int idx = j * kWave + lane;  // owner lane idx / EPL, slot idx % EPL
    px[(idx / kSmEPL) * kSmPad + (idx % kSmEPL)] = xs[j];
  }
  sm_wave_sync();
#pragma unroll
  for (int v = 0; v < kSmEPL / 4; ++v) {
    const float4 a = *reinterpret_cast<const float4*>(px + lane * kSmPad + 4 * v);
    x[4 * v] = a.x; x[4 * v + 1] = a.y; x[4 * v + 2] = a.z; x[4 * v + 3] = a.w;
  }
  sm_wave_sync();
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    const int idx = j * kWave + lane;
    pr[(idx / kSmEPL) * kSmPad + (idx % kSmEPL)] = rs[j];
  }
  sm_wave_sync();
#pragma unroll
  for (int v = 0; v < kSmEPL / 4; ++v) {
    const int4 b = *reinterpret_cast<const int4*>(pr + lane * kSmPad + 4 * v);
    r[4 * v] = b.x; r[4 * v + 1] = b.y; r[4 * v + 2] = b.z; r[4 * v + 3] = b.w;
  }
  // ---- lane-local runs
  M[0] = x[0];
#pragma unroll
  for (int i = 1; i < kSmEPL; ++i) M[i] = r[i] == r[i - 1] ? fmaxf(M[i - 1], x[i]) : x[i];
#pragma unroll
  for (int i = kSmEPL - 2; i >= 0; --i) M[i] = r[i] == r[i + 1] ? M[i + 1] : M[i];
  // x[i] <- exp(x[i] - lane-local run max): the run sums use it, and so does the result
  // (rescaled by exp(run max - row max) where the row reaches beyond the lane)
#pragma unroll
  for (int i = 0; i < kSmEPL; ++i) x[i] = sm_exp(x[i] - M[i]);
  S[0] = x[0];
#pragma unroll
  for (int i = 1; i < kSmEPL; ++i) S[i] = r[i] == r[i - 1] ? S[i - 1] + x[i] : x[i];
#pragma unroll
  for (int i = kSmEPL - 2; i >= 0; --i) S[i] = r[i] == r[i + 1] ? S[i + 1] : S[i];
  // ---- rows spanning lanes
  const int32_t key_f = r[0], key_l = r[kSmEPL - 1];
  const bool multi = key_f != key_l;  // the lane's last run starts (and its first run ends) inside the lane
  const int32_t prev_l = __shfl_up(key_l, 1, kWave), next_f = __shfl_down(key_f, 1, kWave);
  const bool link = lane > 0 && prev_l == key_f;             // first run continues lane-1's last run
  const bool link_n = lane < kWave - 1 && key_l == next_f;   // last run continues into lane+1
  // I = (m, s) of the lane's last row from the row's start (inside the wavefront) to the lane's end
  float im = M[kSmEPL - 1], is = S[kSmEPL - 1];
  {
    int stop = (multi || !link) ? 1 : 0;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const float m2 = __shfl_up(im, d, kWave), s2 = __shfl_up(is, d, kWave);
      const int f2 = __shfl_up(stop, d, kWave);
      const bool act = lane >= d && !stop;  // (selects, not a branch: every lane runs the combine)
      float cm = m2, cs = s2;
      sm_combine(cm, cs, im, is);
      im = act ? cm : im;
      is = act ? cs : is;
      stop = act ? f2 : stop;
    }
  }
  // the row that is the lane's FIRST run: finished total if it ends in this lane
  float fm = M[0], fs = S[0];
  {
    const float pm = __shfl_up(im, 1, kWave), ps = __shfl_up(is, 1, kWave);
    float cm = pm, cs = ps;
    sm_combine(cm, cs, fm, fs);
    fm = (multi && link) ? cm : fm;
    fs = (multi && link) ? cs : fs;
  }
  // E = finished total of the row that is the lane's first run (copy scan from the lane where it ends)
  float em = multi ? fm : im, es = multi ? fs : is;
  {
    int stop = (multi || !link_n) ? 1 : 0;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const float m2 = __shfl_down(em, d, kWave), s2 = __shfl_down(es, d, kWave);
      const int f2 = __shfl_down(stop, d, kWave);
      const bool act = lane + d < kWave && !stop;
      em = act ? m2 : em;
      es = act ? s2 : es;
      stop = act ? f2 : stop;
    }
  }
  const float nm = __shfl_down(em, 1, kWave), ns = __shfl_down(es, 1, kWave);
  const float lm = multi ? (link_n ? nm : im) : em, ls = multi ? (link_n ? ns : is) : es;  // last run's row
  // ---- the rows cut by the range: their positions inside it follow from indptr
  const int32_t R0 = r_first, RL = r_last;
  const int c0 = (int)((first_end < end ? first_end : end) - base);
  const int cl = (int)(end - (last_beg > base ? last_beg : base));
  const float t0m = __shfl(em, 0, kWave), t0s = __shfl(es, 0, kWave);                 // first row, within the range
  const float tlm = __shfl(im, kWave - 1, kWave), tls = __shfl(is, kWave - 1, kWave);  // last row, within the range
  if (lane == 0) {
    SmCarry a, b;
    a.row = b.row = -1;
    a.count = b.count = 0; a.m = b.m = kSmNegBig; a.s = b.s = 0.f;
    a.head = b.head = a.last = b.last = 0; a.pad0 = a.pad1 = b.pad0 = b.pad1 = 0;
    const int64_t fb = first_beg > e0 ? first_beg : e0, fe = first_end < e1 ? first_end : e1;
    const int64_t lb = last_beg > e0 ? last_beg : e0, le = last_end < e1 ? last_end : e1;
    if (cut_start) {  // a follower entry of the first row's chain (the row may also fill the whole range)
      a.row = R0; a.count = c0; a.m = R0 == RL ? tlm : t0m; a.s = R0 == RL ? tls : t0s;
      a.head = (int32_t)((fb - e0) / kSmEPW); a.last = (int32_t)((fe - 1 - e0) / kSmEPW);
    }
    if (cut_end && !(cut_start && R0 == RL)) {  // the head entry of the last row's chain
      b.row = RL; b.count = cl; b.m = tlm; b.s = tls;
      b.head = (int32_t)((lb - e0) / kSmEPW); b.last = (int32_t)((le - 1 - e0) / kSmEPW);
    }
    carry[2 * w] = a;
    carry[2 * w + 1] = b;
  }
  // ---- normalise the finished rows; positions of cut rows are left to softmax_cut_rows_kernel
  const float scale_f = sm_exp(M[0] - em) / es, scale_l = sm_exp(M[kSmEPL - 1] - lm) / ls;
#pragma unroll
  for (int i = 0; i < kSmEPL; ++i) {
    const bool first = r[i] == key_f, last = r[i] == key_l;
    // (hardware reciprocal, 1 ulp: two instructions per position where the IEEE division took ten)
    x[i] = x[i] * ((first || last) ? (first ? scale_f : scale_l) : __builtin_amdgcn_rcpf(S[i]));
  }
  sm_wave_sync();
#pragma unroll
  for (int v = 0; v < kSmEPL / 4; ++v) {
    float4 a;
    a.x = x[4 * v]; a.y = x[4 * v + 1]; a.z = x[4 * v + 2]; a.w = x[4 * v + 3];
    *reinterpret_cast<float4*>(px + lane * kSmPad + 4 * v) = a;
  }
  sm_wave_sync();
  const int64_t skip_lo = cut_start ? base + c0 : base;     // [base, skip_lo) belongs to the cut first row
  const int64_t skip_hi = cut_end ? end - cl : end;         // [skip_hi, end) belongs to the cut last row
  float av[kSmEPL];
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    const int idx = j * kWave + lane;
    av[j] = px[(idx / kSmEPL) * kSmPad + (idx % kSmEPL)];
  }
  if (out_csr) {  // (the wave-uniform tests outside the per-position loops)
#pragma unroll
    for (int j = 0; j < kSmEPL; ++j) {
      const int64_t p = base + j * kWave + lane;
      if (p >= skip_lo && p < skip_hi) out_csr[p] = av[j];
    }
  }
  if (out) {
#pragma unroll
    for (int j = 0; j < kSmEPL; ++j) {
      const int64_t p = base + j * kWave + lane;
      if (p >= skip_lo && p < skip_hi) out[eid ? eid[p] : p] = av[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Round 3 experiment, NOT the shipped sweep (built only with -DKGAT_SOFTMAX_SLOTS; kept as the record of
// a negative result and as a third implementation for cross-checks): the sweep without scans.  The scan
// form above spends ~1,500 instructions per wavefront (two LDS transposes, lane-local run loops,
// thirteen cross-lane scan steps with an exp each) and is bound by instruction issue, not by memory:
// 7 waves per SIMD x 1,500 instructions ~ 25 us.  This form has ~1,150 instructions and no scans - and
// is slower: 48.6 us against 32.4 us stand-alone on the amazon-book graph (51.6 against 35.0 in the
// step).  Its twenty LDS atomics per wavefront mostly hit a handful of addresses (a row is ~23
// consecutive positions, striped over neighbouring lanes), and the LDS serialises same-address
// read-modify-writes lane by lane: with 28 resident wavefronts per CU the LDS pipe, not the issue
// port, becomes the bound.  What it does:
// a wavefront keeps one LDS slot per destination row of its range (row - first row; at most one
// row per position) and lets the LDS do the reductions:
//   1. striped coalesced loads (no transpose), ds_max_f32 of every logit into its row's slot;
//   2. e = exp(x - slot max), added into the slot in 2^-40 fixed point with ds_add_u64 - integer
//      adds are associative, so the sum does not depend on the order the LDS serialises the lanes
//      in: bitwise reproducible by construction (the 3-pass form does the same in global memory);
//   3. one reciprocal per slot; 4. a = e * slot scale, coalesced stores.
// ~350 instructions per wavefront.  Rows cut by the range boundaries: every wavefront also reads a
// halo of 64 positions on either side into two extra accumulators.  A cut row that lies entirely
// inside range + halo is finished here (own positions normalised with the merged statistics); its
// carry entry is still written - a neighbour may need it for a row it cannot finish - with pad0 = 1,
// which tells softmax_cut_rows_kernel to skip it.  Only rows reaching further out (hubs) are left to
// the second launch, whose wavefronts otherwise return after reading their two entries.
constexpr int kSmHalo = 64;
constexpr float kSmFix = 1099511627776.0f;         // 2^40
constexpr float kSmFixInv = 1.0f / 1099511627776.0f;

#ifdef KGAT_SOFTMAX_SLOTS
template <bool IN_CSR, int kSmEPL>
__global__ __launch_bounds__(256) void softmax_slots_kernel(
    int64_t e0, int64_t e1, const int32_t* __restrict__ indptr, const int32_t* __restrict__ row_of,
    const int32_t* __restrict__ eid, const float* __restrict__ logits, float* __restrict__ out,
    float* __restrict__ out_csr, SmCarry* __restrict__ carry) {
  constexpr int EPW = kWave * kSmEPL, NS = EPW + 3;  // slots: rows of the range, halo before, halo after, discard
  __shared__ float s_m[256 / kWave][NS];
  __shared__ unsigned long long s_s[256 / kWave][NS];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int64_t w = sm_wave_index();
  const int64_t base = e0 + w * EPW;
  if (base >= e1) return;
  const int64_t end = base + EPW < e1 ? base + EPW : e1;
  float* pm = s_m[wv];
  unsigned long long* ps = s_s[wv];
  // ---- loads: row ids and input indices of the own positions and of the two halos, then the logits
  const int32_t R0 = row_of[base], RL = row_of[end - 1];
  int32_t slot[kSmEPL], gi[kSmEPL];
  float x[kSmEPL];
  const int64_t pb = base - 1 - lane, pa = end + lane;
  const bool hb = pb >= e0, ha = pa < e1;
  const int64_t pbc = hb ? pb : base, pac = ha ? pa : end - 1;
  const int32_t rb = row_of[pbc], ra = row_of[pac];
  const int32_t gb = IN_CSR ? 0 : eid[pbc], ga = IN_CSR ? 0 : eid[pac];
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {  // (clamped positions, unconditional loads: nothing here may sit under a branch)
    const int64_t p = base + j * kWave + lane;
    const int64_t pc = p < end ? p : end - 1;
    slot[j] = row_of[pc];
    gi[j] = IN_CSR ? 0 : eid[pc];
  }
  const int64_t first_beg = indptr[R0], first_end = indptr[R0 + 1];
  const int64_t last_beg = indptr[RL], last_end = indptr[RL + 1];
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    const int64_t p = base + j * kWave + lane;
    x[j] = IN_CSR ? logits[p < end ? p : end - 1] : logits[gi[j]];
  }
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j)  // positions past the end go to the discard slot: no branches in the reductions
    slot[j] = base + j * kWave + lane < end ? slot[j] - R0 : EPW + 2;
  const float xb = IN_CSR ? logits[pbc] : logits[gb], xa = IN_CSR ? logits[pac] : logits[ga];
  for (int k = lane; k < NS; k += kWave) {
    pm[k] = kSmNegBig;
    ps[k] = 0ull;
  }
  sm_wave_sync();
  // ---- 1. row maxima
  const int sb = (hb && rb == R0) ? EPW : EPW + 2, sa = (ha && ra == RL) ? EPW + 1 : EPW + 2;
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j)
    __hip_atomic_fetch_max(&pm[slot[j]], x[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __hip_atomic_fetch_max(&pm[sb], xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  __hip_atomic_fetch_max(&pm[sa], xa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  sm_wave_sync();
  // ---- 2. exp(x - row max) and the row sums, in 2^-40 fixed point (exact integer adds)
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    x[j] = sm_exp(x[j] - pm[slot[j]]);
    __hip_atomic_fetch_add(&ps[slot[j]], (unsigned long long)(x[j] * kSmFix), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __hip_atomic_fetch_add(&ps[sb], (unsigned long long)(sm_exp(xb - pm[sb]) * kSmFix), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WAVEFRONT);
  __hip_atomic_fetch_add(&ps[sa], (unsigned long long)(sm_exp(xa - pm[sa]) * kSmFix), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WAVEFRONT);
  sm_wave_sync();
  // ---- 3. the rows cut by the range (lane 0; statistics read before the slots turn into scales)
  const int64_t fb = first_beg > e0 ? first_beg : e0, fe = first_end < e1 ? first_end : e1;
  const int64_t lb = last_beg > e0 ? last_beg : e0, le = last_end < e1 ? last_end : e1;
  const bool cut_start = fb < base, cut_end = le > end;
  const int c0 = (int)((fe < end ? fe : end) - base);
  const int cl = (int)(end - (lb > base ? lb : base));
  const int sl = RL - R0;
  // can this wavefront see the whole row?  (first row: starts inside the halo before; if it is also the
  // last row and continues, it must end inside the halo after)
  const bool whole0 = fb >= base - kSmHalo && fe <= end + kSmHalo;
  const bool wholeL = lb >= base - kSmHalo && le <= end + kSmHalo;
  float sc0 = 0.f, scL = 0.f;  // scales of the first / last row's own positions when finished here
  if (lane == 0) {
    const float m0 = pm[0], s0 = (float)ps[0] * kSmFixInv;
    const float mL = pm[sl], sL = (float)ps[sl] * kSmFixInv;
    const float mB = pm[EPW], sB = (float)ps[EPW] * kSmFixInv;
    const float mA = pm[EPW + 1], sA = (float)ps[EPW + 1] * kSmFixInv;
    SmCarry a, b;
    a.row = b.row = -1;
    a.count = b.count = 0; a.m = b.m = kSmNegBig; a.s = b.s = 0.f;
    a.head = b.head = a.last = b.last = 0; a.pad0 = a.pad1 = b.pad0 = b.pad1 = 0;
    if (cut_start) {  // a follower entry of the first row's chain (the row may also fill the whole range)
      a.row = R0; a.count = c0; a.m = m0; a.s = s0;
      a.head = (int32_t)((fb - e0) / EPW); a.last = (int32_t)((fe - 1 - e0) / EPW);
      a.pad0 = whole0 ? 1 : 0;
    }
    if (cut_end && !(cut_start && R0 == RL)) {  // the head entry of the last row's chain
      b.row = RL; b.count = cl; b.m = mL; b.s = sL;
      b.head = (int32_t)((lb - e0) / EPW); b.last = (int32_t)((le - 1 - e0) / EPW);
      b.pad0 = wholeL ? 1 : 0;
    }
    carry[2 * w] = a;
    carry[2 * w + 1] = b;
    if ((cut_start || (cut_end && R0 == RL)) && whole0) {  // first row: own positions + halo before (+ halo after)
      float M = m0, S = s0;
      if (fb < base) sm_combine(M, S, mB, sB);
      if (R0 == RL && fe > end) sm_combine(M, S, mA, sA);
      sc0 = sm_exp(m0 - M) / S;
    }
    if (cut_end && R0 != RL && wholeL) {  // last row: own positions + halo after
      float M = mL, S = sL;
      sm_combine(M, S, mA, sA);
      scL = sm_exp(mL - M) / S;
    }
  }
  sc0 = __shfl(sc0, 0, kWave);
  scL = __shfl(scL, 0, kWave);
  sm_wave_sync();
  // slot -> scale (1 / row sum), in place of the maxima
  for (int k = lane; k < EPW; k += kWave) {
    const float S = (float)ps[k] * kSmFixInv;
    pm[k] = S > 0.f ? 1.0f / S : 0.f;
  }
  sm_wave_sync();
  // ---- 4. normalise and store; positions of cut rows that were not finished here are left to the
  // second launch
  const bool cut0 = cut_start || (cut_end && R0 == RL), cutL = cut_end && R0 != RL;
  const bool skip0 = cut0 && !whole0, skipL = cutL && !wholeL;
#pragma unroll
  for (int j = 0; j < kSmEPL; ++j) {
    const int64_t p = base + j * kWave + lane;
    if (p >= end) continue;
    const bool in0 = slot[j] == 0, inL = slot[j] == sl;
    if ((in0 && skip0) || (inL && skipL)) continue;
    float scale = pm[slot[j]];
    if (in0 && cut0) scale = sc0;
    if (inL && cutL) scale = scL;
    const float a = x[j] * scale;
    if (out_csr) out_csr[p] = a;
    if (out) out[eid ? eid[p] : p] = a;
  }
}

#endif  // KGAT_SOFTMAX_SLOTS

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

// positions [lo, hi) of one row, 8 x 64 per step so that the loads of a step are in flight together
template <bool IN_CSR>
__device__ __forceinline__ void fix_span(int64_t lo, int64_t hi, int lane, float M, float S,
                                         const int32_t* __restrict__ eid, const float* __restrict__ logits,
                                         float* __restrict__ out, float* __restrict__ out_csr) {
  constexpr int U = 8;
  for (int64_t p0 = lo; p0 < hi; p0 += U * kWave) {
    int64_t e[U];
    float x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // all indices first, then all logits (see softmax_local_kernel)
      const int64_t p = p0 + u * kWave + lane;
      const int64_t pc = p < hi ? p : hi - 1;
      e[u] = eid ? eid[pc] : pc;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * kWave + lane;
      const int64_t pc = p < hi ? p : hi - 1;
      x[u] = IN_CSR ? logits[pc] : logits[e[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * kWave + lane;
      if (p < hi) {
        const float a = sm_exp(x[u] - M) / S;
        if (out_csr) out_csr[p] = a;
        if (out) out[e[u]] = a;
      }
    }
  }
}

template <bool IN_CSR, int kSmEPL>
__global__ __launch_bounds__(256) void softmax_cut_rows_kernel(
    int64_t e0, int64_t e1, const int32_t* __restrict__ eid, const float* __restrict__ logits,
    float* __restrict__ out, float* __restrict__ out_csr, const SmCarry* __restrict__ carry) {
  constexpr int kSmEPW = SmGeom<kSmEPL>::EPW;
  const int lane = threadIdx.x % kWave;
  const int64_t w = sm_wave_index();
  const int64_t base = e0 + w * kSmEPW;
  if (base >= e1) return;
  const int64_t end = base + kSmEPW < e1 ? base + kSmEPW : e1;
  // the cut rows sit at the two ends of the range: their first 64 positions are requested together
  // with the carry entries (one memory round trip for the common short case)
  const SmCarry ca = carry[2 * w], cb = carry[2 * w + 1];
  const int64_t pa = base + lane < end ? base + lane : end - 1;
  const int64_t pb = end - 1 - lane >= base ? end - 1 - lane : base;
  const int64_t ea = eid ? eid[pa] : pa, eb = eid ? eid[pb] : pb;
  const float xa = IN_CSR ? logits[pa] : logits[ea], xb = IN_CSR ? logits[pb] : logits[eb];
  if (ca.row >= 0 && !ca.pad0) {  // (pad0: the sweep saw the whole row through its halo and finished its own positions)
    float M, S;
    sm_chain_total(carry, ca.head, ca.last, lane, M, S);
    if (lane < ca.count) {
      const float a = sm_exp(xa - M) / S;
      if (out_csr) out_csr[pa] = a;
      if (out) out[ea] = a;
    }
    fix_span<IN_CSR>(base + kWave, base + ca.count, lane, M, S, eid, logits, out, out_csr);
  }
  if (cb.row >= 0 && !cb.pad0) {
    float M, S;
    sm_chain_total(carry, cb.head, cb.last, lane, M, S);
    if (lane < cb.count) {
      const float a = sm_exp(xb - M) / S;
      if (out_csr) out_csr[pb] = a;
      if (out) out[eb] = a;
    }
    fix_span<IN_CSR>(end - cb.count, end - kWave, lane, M, S, eid, logits, out, out_csr);
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
#ifdef KGAT_SOFTMAX_SLOTS  // A/B builds: the LDS-atomic sweep (slower, see softmax_slots_kernel)
  hipLaunchKernelGGL((softmax_slots_kernel<IN_CSR, EPL>), dim3(blocks), dim3(256), 0, st, e_begin, e_end, indptr, row_of,
                     eid, logits, out, out_csr, carry);
#else
  hipLaunchKernelGGL((softmax_local_kernel<IN_CSR, EPL>), dim3(blocks), dim3(256), 0, st, e_begin, e_end, indptr, row_of,
                     eid, logits, out, out_csr, carry);
#endif
  hipLaunchKernelGGL((softmax_cut_rows_kernel<IN_CSR, EPL>), dim3(blocks), dim3(256), 0, st, e_begin, e_end, eid, logits,
                     out, out_csr, (const SmCarry*)carry);
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
