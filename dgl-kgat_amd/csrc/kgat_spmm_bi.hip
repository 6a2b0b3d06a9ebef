// One KGATConv forward as one pass for gfx950: the u_mul_e -> sum aggregation (kgat_spmm_impl.h) with the
// bi-interaction's dense part behind it in the same launch.  Rows S1 + B1 + B2 of SURVEY.md 8a; replaces the
// sequence of reference models.py:63-66 + :165 (update_all(u_mul_e, sum), th.mul, res_fc_2, LeakyReLU,
// F.normalize) for the no-grad forward.
//
// Why: the two-launch form writes P = h * h_N (N x D fp32) and reads it back - 82 MB of cache-fabric traffic
// per 64-wide layer on the amazon-book-shaped CKG plus a launch boundary, ~12 % of the step.  Here a row the
// aggregation completes goes into an LDS row buffer of its workgroup (slot = row - first row of the tile);
// when the tile's edge loop and combine are done, the edge records' LDS is reused for W2 in MFMA fragment
// order and the workgroup's four wavefronts run 16-row blocks of Z = P W2^T on v_mfma_f32_16x16x4_f32 (exact
// fp32, the products and summation order of kgat_bi_interaction_f32: same bits), LeakyReLU, the row norm, and
// store h_out / norm_out.  A tile's first and last row (which may continue in a neighbour tile) go through the
// partial buffer as before; the finish kernel sums them and runs the same 16-row block on them.
// Rows of a tile beyond the buffer's capacity spill their P row to a global scratch and are read back by the
// same workgroup (L2-local).  The buffer's size is set by occupancy, not by the typical tile (44 rows):
// profiles/r04_spmm_lds_ballast_ab.txt.
#include "kgat_spmm_impl.h"

using namespace kgat;

namespace {

template <int DI, int DO>
int launch_fused(const SpmmArgs& a) {
  return launch_merge<DI / 4, true, false, DO>(a);
}

}  // namespace

extern "C" {


int kgat_spmm_bi_fused_supported(int d_in, int d_out) {
  auto ok = [](int d) { return d == 16 || d == 32 || d == 64; };
  return ok(d_in) && ok(d_out) && d_out <= d_in;
}

int kgat_spmm_bi_fused_f32(int64_t n_rows, int64_t row0, int64_t e_begin, int64_t e_end, int d_in, int d_out,
                           const int32_t* indptr, const int32_t* col, const int32_t* row_of,
                           const float* X, const float* w, const float* W2, float negative_slope,
                           float* h_out, float* norm_out, int64_t norm_stride, float* scratch,
                           void* workspace, size_t workspace_bytes, float* self_out, int64_t self_stride,
                           kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && row0 >= 0, "spmm_bi_fused: bad size (n_rows=%lld row0=%lld)", (long long)n_rows,
                 (long long)row0);
  KGAT_CHECK_ARG(row0 + n_rows < INT32_MAX, "spmm_bi_fused: row range exceeds int32");
  KGAT_CHECK_ARG(e_begin >= 0 && e_end >= e_begin && e_end < INT32_MAX, "spmm_bi_fused: bad edge range");
  if (!kgat_spmm_bi_fused_supported(d_in, d_out)) {
    set_error("spmm_bi_fused: unsupported widths %d -> %d", d_in, d_out);
    return KGAT_E_UNSUPPORTED;
  }
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(indptr && X && W2 && scratch && (h_out || norm_out), "spmm_bi_fused: null pointer");
  KGAT_CHECK_ARG(e_end == e_begin || (col && w && row_of), "spmm_bi_fused: null col / w / row_of");
  KGAT_CHECK_ARG(norm_out == nullptr || (norm_stride >= d_out && norm_stride % 4 == 0 &&
                                         (reinterpret_cast<uintptr_t>(norm_out) & 15u) == 0),
                 "spmm_bi_fused: norm_out must be 16-byte aligned with a row stride that is a multiple of 4 floats >= d_out");
  if (self_out != nullptr)
    KGAT_CHECK_ARG(self_stride >= d_in && self_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(self_out) & 15u) == 0,
                   "spmm_bi_fused: self_out must be 16-byte aligned with a row stride that is a multiple of 4 floats >= d_in");
  SpmmArgs a;
  a.self_out = self_out; a.self_stride = self_stride;
  a.n_rows = n_rows; a.row0 = row0; a.D = d_in;
  a.indptr = indptr; a.col = col; a.row_of = row_of; a.eid = nullptr; a.order = nullptr;
  a.X = X; a.w = w; a.out = scratch; a.ws = workspace; a.ws_bytes = workspace_bytes;
  a.flags = KGAT_SPMM_MUL_SELF; a.algo = KGAT_SPMM_ALGO_MERGE;
  a.e0_host = (int32_t)e_begin; a.e1_host = (int32_t)e_end;
  a.st = as_stream(stream);
  a.bi.W2 = W2; a.bi.slope = negative_slope; a.bi.h_out = h_out; a.bi.norm_out = norm_out;
  a.bi.norm_stride = norm_stride; a.bi.indptr = indptr;
#define KGAT_FUSED_CASE(DI, DO) if (d_in == DI && d_out == DO) return launch_fused<DI, DO>(a);
  KGAT_FUSED_CASE(64, 64) KGAT_FUSED_CASE(64, 32) KGAT_FUSED_CASE(64, 16)
  KGAT_FUSED_CASE(32, 32) KGAT_FUSED_CASE(32, 16) KGAT_FUSED_CASE(16, 16)
#undef KGAT_FUSED_CASE
  set_error("spmm_bi_fused: unsupported widths %d -> %d", d_in, d_out);
  return KGAT_E_UNSUPPORTED;
}

}  // extern "C"
