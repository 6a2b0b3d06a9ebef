// Bi-interaction dense part of a KGAT propagation layer, for gfx950.  Rows B1 + B2 of
// SURVEY.md 8a.  Replaces, for the forward (no-grad) path, the torch sequence of reference
// models.py:66 + :165-166
//     out = F.leaky_relu(res_fc_2(h * h_N));  cache.append(F.normalize(out, p=2, dim=1))
// by one kernel: Z = P @ W2^T (P = h * h_N already formed in the SpMM epilogue), LeakyReLU,
// the un-normalised rows for the next layer and the L2-normalised rows written straight into
// their column slice of the concatenated output (models.py:167).
//
// Design: N x D_in x D_out with D <= 128 is a skinny GEMM that streams P once (HBM bound:
// 4*(D_in + 2*D_out) bytes per row against 2*D_in*D_out FLOP).  One wavefront owns 16-row
// tiles; W2 (<= 32 KB) is staged once per workgroup through LDS and then lives in registers as
// fp32 MFMA B fragments for the whole launch
// (v_mfma_f32_16x16x4_f32, exact fp32); the rows of a tile are contiguous, so the A-fragment
// loads are fully coalesced float4 reads; the row norm is a DPP reduction over the 16 lanes
// that hold one row's columns.
#include <math.h>

#include "kgat_common.h"

namespace kgat {

typedef float floatx4_d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row16_sum_d(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}

template <int DI, int DO>
__global__ __launch_bounds__(256) void bi_interaction_kernel(
    int32_t n_rows, const float* __restrict__ P, const float* __restrict__ W2, float slope,
    float* __restrict__ h_out, float* __restrict__ norm_out, int64_t norm_stride) {
  constexpr int KS = DI / 4, KT = DO / 16;
  // W2 is staged once per workgroup through LDS (coalesced 16-byte reads of the whole matrix),
  // laid out in B-fragment order so that every wave then pulls its fragments with
  // conflict-free ds_read_b32: s_w[(s*KT + c)*64 + q*16 + i] = W2[16c + i][16*(s>>2) + 4q + (s&3)]
  __shared__ float s_w[KS * KT * kWave];
  for (int idx = threadIdx.x * 4; idx < DO * DI; idx += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(W2 + idx);
    const int j = idx / DI, k0 = idx % DI;  // four consecutive k of output column j
    const int c = j >> 4, i = j & 15;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = k0 + t;
      const int s = (k >> 4) * 4 + (k & 3), q = (k >> 2) & 3;
      s_w[(s * KT + c) * kWave + q * 16 + i] = vv[t];
    }
  }
  __syncthreads();

  const int lane = threadIdx.x % kWave;
  const int i = lane & 15, q = lane >> 4;
  const int64_t n_waves = (int64_t)gridDim.x * (256 / kWave);
  const int64_t wv = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
  const int32_t n_tiles = (n_rows + 15) >> 4;
  const int32_t t_begin = (int32_t)((int64_t)n_tiles * wv / n_waves);
  const int32_t t_end = (int32_t)((int64_t)n_tiles * (wv + 1) / n_waves);
  if (t_begin >= t_end) return;

  float wreg[KS][KT];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int c = 0; c < KT; ++c) wreg[s][c] = s_w[(s * KT + c) * kWave + lane];

  auto load_a = [&](int32_t t, float (&a)[KS]) {
    int32_t ra = (t << 4) + i;
    ra = ra < n_rows ? ra : n_rows - 1;
    const float4* pa = reinterpret_cast<const float4*>(P + (size_t)ra * DI) + q;
#pragma unroll
    for (int m = 0; m < DI / 16; ++m) {
      const float4 v = pa[m * 4];
      a[4 * m + 0] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
    }
  };
  auto tile = [&](int32_t t, const float (&a)[KS]) {
    const int32_t row0 = t << 4;
    floatx4_d acc[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[c] = (floatx4_d){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < KT; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wreg[s][c], acc[c], 0, 0, 0);
    // acc[c][j] = Z[row0 + 4q + j][16c + i]
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = acc[c][j];
        z = z >= 0.f ? z : z * slope;
        acc[c][j] = z;
        ss[j] = fmaf(z, z, ss[j]);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int32_t row = row0 + 4 * q + j;
      const float nrm = fmaxf(sqrtf(row16_sum_d(ss[j])), 1e-12f);
      if (row < n_rows) {
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const float z = acc[c][j];
          if (h_out) h_out[(size_t)row * DO + 16 * c + i] = z;
          if (norm_out) norm_out[(size_t)row * norm_stride + 16 * c + i] = z / nrm;
        }
      }
    }
  };

  // rows of the next tile are requested before the current tile is computed
  float a0[KS], a1[KS];
  load_a(t_begin, a0);
  for (int32_t t = t_begin; t < t_end; t += 2) {
    load_a(t + 1 < t_end ? t + 1 : t, a1);
    tile(t, a0);
    if (t + 1 >= t_end) break;
    load_a(t + 2 < t_end ? t + 2 : t + 1, a0);
    tile(t + 1, a1);
  }
}

// F.normalize(x, p=2, dim=1, eps=1e-12) of contiguous rows into a strided destination (a column
// slice of the concatenated readout, reference models.py:165-167).  One 16-lane group per row.
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(int64_t n_rows, int d,
                                                                const float* __restrict__ x,
                                                                float* __restrict__ out,
                                                                int64_t out_stride) {
  const int sub = threadIdx.x >> 4, sl = threadIdx.x & 15;
  for (int64_t row = (int64_t)blockIdx.x * 16 + sub; row < n_rows; row += (int64_t)gridDim.x * 16) {
    const float* xr = x + (size_t)row * d;
    float ss = 0.f;
    for (int c = sl; c < d; c += 16) ss = fmaf(xr[c], xr[c], ss);
    const float nrm = fmaxf(sqrtf(row16_sum_d(ss)), 1e-12f);
    for (int c = sl; c < d; c += 16) out[(size_t)row * out_stride + c] = xr[c] / nrm;
  }
}

template <int DI, int DO>
static int launch_bi(int64_t n_rows, const float* P, const float* W2, float slope, float* h_out,
                     float* norm_out, int64_t norm_stride, hipStream_t st) {
  const int64_t tiles = (n_rows + 15) / 16;
  int64_t blocks = (tiles + 3) / 4;  // at least one tile per wave ...
  if (blocks > 2048) blocks = 2048;  // ... at most 8 blocks per CU, contiguous tile ranges
  hipLaunchKernelGGL((bi_interaction_kernel<DI, DO>), dim3((unsigned)blocks), dim3(256), 0, st,
                     (int32_t)n_rows, P, W2, slope, h_out, norm_out, norm_stride);
  KGAT_CHECK_LAUNCH("bi_interaction");
  return KGAT_OK;
}

}  // namespace kgat

using namespace kgat;

extern "C" {

int kgat_l2_normalize_rows_f32(int64_t n_rows, int d, const float* x, float* out, int64_t out_stride,
                               kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && d > 0 && out_stride >= d, "l2_normalize_rows: bad size");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(x && out, "l2_normalize_rows: null pointer");
  int64_t blocks = (n_rows + 15) / 16;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n_rows, d,
                     x, out, out_stride);
  KGAT_CHECK_LAUNCH("l2_normalize_rows");
  return KGAT_OK;
}

int kgat_bi_interaction_supported(int d_in, int d_out) {
  auto ok = [](int d) { return d == 16 || d == 32 || d == 64 || d == 128; };
  return ok(d_in) && ok(d_out) && (d_in / 4) * (d_out / 16) <= 128;
}

int kgat_bi_interaction_f32(int64_t n_rows, int d_in, int d_out, const float* P, const float* W2,
                            float negative_slope, float* h_out, float* norm_out,
                            int64_t norm_stride, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_rows < INT32_MAX, "bi_interaction: bad row count");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(P && W2 && (h_out || norm_out), "bi_interaction: null pointer");
  KGAT_CHECK_ARG(norm_out == nullptr || norm_stride >= d_out, "bi_interaction: bad norm_stride");
  if (!kgat_bi_interaction_supported(d_in, d_out)) {
    set_error("bi_interaction: unsupported widths %d -> %d", d_in, d_out);
    return KGAT_E_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
#define KGAT_BI_CASE(DI, DO) \
  if (d_in == DI && d_out == DO) return launch_bi<DI, DO>(n_rows, P, W2, negative_slope, h_out, norm_out, norm_stride, st);
  KGAT_BI_CASE(16, 16) KGAT_BI_CASE(16, 32) KGAT_BI_CASE(16, 64) KGAT_BI_CASE(16, 128)
  KGAT_BI_CASE(32, 16) KGAT_BI_CASE(32, 32) KGAT_BI_CASE(32, 64) KGAT_BI_CASE(32, 128)
  KGAT_BI_CASE(64, 16) KGAT_BI_CASE(64, 32) KGAT_BI_CASE(64, 64) KGAT_BI_CASE(64, 128)
  KGAT_BI_CASE(128, 16) KGAT_BI_CASE(128, 32) KGAT_BI_CASE(128, 64)
#undef KGAT_BI_CASE
  set_error("bi_interaction: unsupported widths %d -> %d", d_in, d_out);
  return KGAT_E_UNSUPPORTED;
}

}  // extern "C"
