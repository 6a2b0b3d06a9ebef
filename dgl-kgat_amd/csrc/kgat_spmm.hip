// u_mul_e -> sum aggregation (SpMM) for gfx950.  Row S1 / S1b of SURVEY.md 8a.
//
// Replaces g.update_all(fn.u_mul_e('h','w','m'), fn.sum('m','h_neighbor')) of reference
// models.py:63 (DGL binary_reduce(sum, mul, SRC, EDGE), (N,D) x (E,1) broadcast):
//   out[v,:] = sum_{p in row v} w_p * X[col[p],:]
//
// Design (HBM/Infinity-Cache gather bound, 0.5 FLOP/B - no MFMA):
//  * Edge-balanced ("merge-path") decomposition over the destination-sorted edge array:
//    every 256-thread workgroup owns a tile of TE consecutive CSR positions, whatever rows
//    they belong to, so a power-law in-degree distribution cannot unbalance the launch.
//  * A row of X is D floats; LPR = D/4 lanes read it with one 16-byte load each (a full
//    256-B row per 16 lanes at D = 64, coalesced).  A wavefront therefore works on 64/LPR
//    edges per load instruction; each lane group ("subgroup") walks its own run of C
//    consecutive edges with U loads in flight, accumulating in registers and flushing when
//    the destination row changes (the row id of every CSR position is a graph-static array).
//  * The (col, w, row id) triples reach the lane groups in one of two ways: the second form
//    (spmm_merge2_kernel, weights in CSR order - the path the KGAT layer uses) stages a tile's
//    triples once into LDS as 16-byte records and reads one record per edge with a
//    ds_read_b128 broadcast; the first form (spmm_merge_kernel, also serves weights given in
//    edge-id order through eid) reads them coalesced per lane group and hands them around with
//    wavefront shuffles (ds_bpermute).
//  * Rows that end inside a run are stored straight to `out`.  A run's first and last row
//    may continue in a neighbour run: those partial sums are combined through LDS in run
//    order by the workgroup; only the first/last row of the whole tile goes to a small
//    global partial buffer, which the finish kernel sums in tile order.  No float atomics:
//    the summation order is fixed, results are bitwise reproducible.
//  * The finish kernel also writes the zero rows (destinations without in-edges).
#define KGAT_SPMM_MAIN_TU 1
#include "kgat_spmm_impl.h"

namespace kgat {

// SDDMM: grad_w[e] = <X[src e], G[dst e]>.  One subgroup of 16 lanes per edge.
__global__ __launch_bounds__(256) void sddmm_dot_kernel(int64_t n_edges, int D,
                                                        const int32_t* __restrict__ src,
                                                        const int32_t* __restrict__ dst,
                                                        const float* __restrict__ X,
                                                        const float* __restrict__ G,
                                                        float* __restrict__ out) {
  constexpr int L = 16;
  const int sub = threadIdx.x / L, sl = threadIdx.x % L;
  const int64_t e = (int64_t)blockIdx.x * (256 / L) + sub;
  if (e >= n_edges) return;
  const float* x = X + (size_t)src[e] * D;
  const float* g = G + (size_t)dst[e] * D;
  float acc = 0.f;
  for (int d = sl; d < D; d += L) acc = fmaf(x[d], g[d], acc);
#pragma unroll
  for (int off = L / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, L);
  if (sl == 0) out[e] = acc;
}

// Measurement aid (bench.py's roofline.gather_ceiling): reads the rows X[col[p], :] of CSR positions [0, n_edges) and
// nothing else - no weights, no row ids, no output (a running sum keeps the loads alive; `sink` is never written for
// finite data).  LPR lanes x one float4 per row, U rows in flight per lane group: the access pattern of the
// aggregation's edge loop without the aggregation.  What this launch takes is the floor of any kernel that has to
// fetch those rows through the cache hierarchy, whatever serves them (L2, Infinity Cache, HBM).
template <int LPR, int U>
__global__ __launch_bounds__(256) void gather_probe_kernel(int64_t n_edges, const int32_t* __restrict__ col,
                                                           const float4* __restrict__ X, float4* __restrict__ sink) {
  constexpr int EPS = 64 / LPR;  // edges per wavefront step
  const int lane = threadIdx.x & 63, sl = lane % LPR, sub = lane / LPR;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  constexpr int64_t kPerWave = 512;
  const int64_t base = wave * kPerWave;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t p0 = base; p0 < base + kPerWave && p0 < n_edges; p0 += EPS * U) {
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + EPS * u + sub;
      c[u] = col[p < n_edges ? p : n_edges - 1];
    }
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = X[(size_t)c[u] * LPR + sl];
#pragma unroll
    for (int u = 0; u < U; ++u) acc = add4(acc, v[u]);
  }
  if (acc.x == 12345.678f && acc.y == -8765.4321f) sink[wave] = acc;
}

}  // namespace kgat

using namespace kgat;

extern "C" {


size_t kgat_spmm_workspace_bytes(int64_t n_edges, int D) {
  const int lpr = lpr_for(D);
  if (lpr == 0 || n_edges <= 0) return 256;
  const int nsub = spmm_threads(lpr) / lpr;
  const int64_t te = (int64_t)nsub * run_len(lpr), te_s = (int64_t)nsub * short_run_len(lpr);
  const int64_t te_m = (int64_t)nsub * mid_run_len(lpr);
  int64_t tiles = (n_edges + te - 1) / te;
  const int64_t tiles_s = (n_edges + te_s - 1) / te_s, tiles_m = (n_edges + te_m - 1) / te_m;
  if (tiles_m <= kMidRunTileLimit && tiles_m > tiles) tiles = tiles_m;    // the launch takes the half-length runs
  if (tiles_s <= kShortRunTileLimit && tiles_s > tiles) tiles = tiles_s;  // the launch takes the short runs
  const int64_t te_f = (int64_t)nsub * fused_run_len(lpr), tiles_f = (n_edges + te_f - 1) / te_f;
  if (tiles_f > tiles) tiles = tiles_f;                                   // kgat_spmm_bi_fused_f32's run length
  return align_up((size_t)tiles * 2 * lpr * sizeof(float4), 256) + 256;
}

int kgat_spmm_tile_edges(int64_t n_edges, int D) {
  if (n_edges < 0) return 0;
  int te = 0;
  switch (D) {
    case 16: te = merge_tile_edges<4>(n_edges); break;
    case 32: te = merge_tile_edges<8>(n_edges); break;
    case 64: te = merge_tile_edges<16>(n_edges); break;
    case 128: te = merge_tile_edges<32>(n_edges); break;
    default: return 0;
  }
  return (te & (te - 1)) == 0 ? te : 0;  // (the consumer shifts; every shipped geometry is a power of two)
}

int kgat_spmm_umule_sum_f32(int64_t n_rows, int64_t row0, int64_t e_begin, int64_t e_end, int D,
                            const int32_t* indptr, const int32_t* col, const int32_t* row_of,
                            const int32_t* eid, const float* X, const float* w, float* out,
                            const int32_t* order, void* workspace, size_t workspace_bytes,
                            unsigned flags, int algo, float* self_out, int64_t self_stride,
                            kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && row0 >= 0 && D > 0, "spmm: bad size (n_rows=%lld row0=%lld D=%d)",
                 (long long)n_rows, (long long)row0, D);
  KGAT_CHECK_ARG(row0 + n_rows < INT32_MAX, "spmm: row range exceeds int32");
  KGAT_CHECK_ARG(e_begin >= 0 && e_end >= e_begin && e_end < INT32_MAX, "spmm: bad edge range");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(indptr && X && out, "spmm: null pointer");
  KGAT_CHECK_ARG(e_end == e_begin || (col && w), "spmm: null col/w");
  KGAT_CHECK_ARG((flags & ~(unsigned)(KGAT_SPMM_MUL_SELF | KGAT_SPMM_DEFER_FINISH)) == 0, "spmm: unknown flags 0x%x", flags);
  KGAT_CHECK_ARG(algo >= KGAT_SPMM_ALGO_AUTO && algo <= KGAT_SPMM_ALGO_MERGE1,
                 "spmm: unknown algo %d", algo);
  if (algo == KGAT_SPMM_ALGO_AUTO)
    algo = (lpr_for(D) && (row_of || e_end == e_begin)) ? KGAT_SPMM_ALGO_MERGE
                                  : (lpr_for(D) ? KGAT_SPMM_ALGO_ROWS : KGAT_SPMM_ALGO_GENERIC);
  if (lpr_for(D) == 0) algo = KGAT_SPMM_ALGO_GENERIC;
  KGAT_CHECK_ARG((algo != KGAT_SPMM_ALGO_MERGE && algo != KGAT_SPMM_ALGO_MERGE1) || row_of != nullptr || e_end == e_begin,
                 "spmm: merge algorithm needs row_of");
  KGAT_CHECK_ARG(order == nullptr || algo == KGAT_SPMM_ALGO_ROWS,
                 "spmm: a row order only applies to the rows algorithm");
  if (flags & KGAT_SPMM_DEFER_FINISH) {
    KGAT_CHECK_ARG(!(flags & KGAT_SPMM_MUL_SELF) && eid == nullptr && algo == KGAT_SPMM_ALGO_MERGE &&
                       kgat_spmm_tile_edges(e_end - e_begin, D) > 0,
                   "spmm: KGAT_SPMM_DEFER_FINISH goes with the plain operator, CSR-ordered weights, the merge algorithm "
                   "and D in {16, 32, 64, 128}");
  }
  if (self_out != nullptr) {
    KGAT_CHECK_ARG((flags & KGAT_SPMM_MUL_SELF) && eid == nullptr && algo == KGAT_SPMM_ALGO_MERGE,
                   "spmm: self_out goes with KGAT_SPMM_MUL_SELF, CSR-ordered weights and the merge algorithm");
    KGAT_CHECK_ARG(self_stride >= D && self_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(self_out) & 15u) == 0,
                   "spmm: self_out must be 16-byte aligned with a row stride that is a multiple of 4 floats >= D");
  }
  SpmmArgs a;
  a.self_out = self_out; a.self_stride = self_stride;
  a.n_rows = n_rows; a.row0 = row0; a.D = D;
  a.indptr = indptr; a.col = col; a.row_of = row_of; a.eid = eid; a.order = order;
  a.X = X; a.w = w; a.out = out; a.ws = workspace; a.ws_bytes = workspace_bytes;
  a.flags = flags; a.algo = algo;
  a.e0_host = (int32_t)e_begin; a.e1_host = (int32_t)e_end;
  a.st = as_stream(stream);
  const bool mul = flags & KGAT_SPMM_MUL_SELF;
  if (mul) return eid ? dispatch_width<true, true>(a) : dispatch_width<true, false>(a);
  return eid ? dispatch_width<false, true>(a) : dispatch_width<false, false>(a);
}

constexpr int kProbeU = 8;  // rows in flight per lane group (A/B builds)
int kgat_gather_probe_f32(int64_t n_edges, int D, const int32_t* col, const float* X, float* sink, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_edges >= 0, "gather_probe: bad size");
  if (n_edges == 0) return KGAT_OK;
  KGAT_CHECK_ARG(col && X && sink, "gather_probe: null pointer");
  const unsigned blocks = (unsigned)((n_edges + 2047) / 2048);
  const float4* X4 = reinterpret_cast<const float4*>(X);
  float4* s4 = reinterpret_cast<float4*>(sink);
  switch (D) {
    case 16: hipLaunchKernelGGL((gather_probe_kernel<4, 8>), dim3(blocks), dim3(256), 0, as_stream(stream), n_edges, col, X4, s4); break;
    case 32: hipLaunchKernelGGL((gather_probe_kernel<8, 8>), dim3(blocks), dim3(256), 0, as_stream(stream), n_edges, col, X4, s4); break;
    case 64: hipLaunchKernelGGL((gather_probe_kernel<16, kProbeU>), dim3(blocks), dim3(256), 0, as_stream(stream), n_edges, col, X4, s4); break;
    case 128: hipLaunchKernelGGL((gather_probe_kernel<32, 8>), dim3(blocks), dim3(256), 0, as_stream(stream), n_edges, col, X4, s4); break;
    default:
      set_error("gather_probe: D must be 16, 32, 64 or 128 (got %d)", D);
      return KGAT_E_UNSUPPORTED;
  }
  KGAT_CHECK_LAUNCH("gather_probe");
  return KGAT_OK;
}

int kgat_sddmm_dot_f32(int64_t n_edges, int D, const int32_t* src, const int32_t* dst,
                       const float* X, const float* grad_out, float* grad_w,
                       kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_edges >= 0 && D > 0, "sddmm: bad size");
  if (n_edges == 0) return KGAT_OK;
  KGAT_CHECK_ARG(src && dst && X && grad_out && grad_w, "sddmm: null pointer");
  hipLaunchKernelGGL(sddmm_dot_kernel, dim3((unsigned)((n_edges + 15) / 16)), dim3(256), 0,
                     as_stream(stream), n_edges, D, src, dst, X, grad_out, grad_w);
  KGAT_CHECK_LAUNCH("sddmm_dot");
  return KGAT_OK;
}

}  // extern "C"
