// Edge softmax over the incoming edges of each destination, for gfx950.  Row A3 of
// SURVEY.md 8a.  Replaces dgl.nn.pytorch.softmax.edge_softmax (call site reference
// models.py:153; DGL 0.4.x runs it as copy_reduce(max) / sub / exp / copy_reduce(sum) / div,
// five E-sized temporaries):
//   a[e] = exp(s[e] - M[dst e]) / Z[dst e],  M[v] = max_{e->v} s[e],  Z[v] = sum_{e->v} exp(s[e]-M[v])
//
// Design (HBM bound, ~24-36 B/edge over three streaming passes, no per-row launch shape):
//  * Work is split by EDGE over the destination-sorted edge array, one lane per CSR
//    position, so hub destinations (10^5..10^6 in-edges) cost the same per edge as leaves.
//  * Inside a wavefront the lanes of one destination form a contiguous segment; a segmented
//    shuffle scan reduces them and only the last lane of each segment touches global memory.
//  * Pass 1: row max via integer-ordered atomic max (order independent).
//    Pass 2: row sum of exp(s - M): per-wave partials in fp32 (fixed order), combined across
//            waves in 2^-40 fixed point with 64-bit integer atomic adds - integer addition is
//            associative, so the sum does not depend on arrival order and the result is bitwise
//            reproducible (a float atomic add would not be).  A partial is <= 64, so 2^18
//            waves (2^24 edges) per destination fit in 64 bits.
//    Pass 3: normalise, write in CSR order (consumed by the SpMM) and/or edge-id order.
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

extern "C" {

size_t kgat_edge_softmax_workspace_bytes(int64_t n_nodes) {
  const size_t n = (size_t)(n_nodes > 0 ? n_nodes : 1);
  return align_up(n * sizeof(float), 256) + align_up(n * sizeof(unsigned long long), 256);
}

int kgat_edge_softmax_f32(int64_t n_nodes, int64_t e_begin, int64_t e_end,
                          const int32_t* row_of, const int32_t* eid, const float* logits,
                          int logits_in_csr_order, float* out, float* out_csr, void* workspace,
                          size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && e_begin >= 0 && e_end >= e_begin && e_end < INT32_MAX,
                 "edge_softmax: bad size");
  if (e_end == e_begin) return KGAT_OK;
  KGAT_CHECK_ARG(row_of && logits && workspace, "edge_softmax: null pointer");
  KGAT_CHECK_ARG(out || out_csr, "edge_softmax: no output requested");
  KGAT_CHECK_ARG(eid != nullptr || (logits_in_csr_order && out == nullptr),
                 "edge_softmax: edge-id ordered input/output needs eid");
  if (workspace_bytes < kgat_edge_softmax_workspace_bytes(n_nodes)) {
    set_error("edge_softmax: workspace too small");
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
  KGAT_CHECK_LAUNCH("edge_softmax");
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
