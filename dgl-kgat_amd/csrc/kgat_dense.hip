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
// (v_mfma_f32_16x16x4_f32, exact fp32); the rows of a tile are contiguous, so the row-fragment
// loads are fully coalesced float4 reads; the product is issued with the operands swapped so
// that a lane ends up with four consecutive columns of one row (16-byte stores), and the row
// norm is a sum over a lane's values plus two cross-lane adds.
#include <math.h>
#include <stdint.h>

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

#ifndef KGAT_BI_NT_LOADS
#define KGAT_BI_NT_LOADS 0   // A/B builds: the row streams as non-temporal loads (measured 25-70 % SLOWER: profiles/r04_bi_probe.txt)
#endif
#ifndef KGAT_BI_NT_STORES
#define KGAT_BI_NT_STORES 0  // A/B builds: the normalised slice and the ego block as non-temporal stores (no change)
#endif
__device__ __forceinline__ float4 ld_row4(const float4* p) {
#if KGAT_BI_NT_LOADS
  const floatx4_d v = __builtin_nontemporal_load(reinterpret_cast<const floatx4_d*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_final4(float4* p, const float4& v) {
#if KGAT_BI_NT_STORES
  const floatx4_d x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<floatx4_d*>(p));
#else
  *p = v;
#endif
}

// Dropout mask of the training form: a counter-based hash of (seed, element index), so the backward
// pass recomputes the mask instead of storing it.  keep <=> hash >= p * 2^32.
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t index, uint32_t threshold) {
  uint32_t x = (index * 0x9E3779B1u) ^ seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x >= threshold;
}

// MODE 0: the A operand is P as given.  MODE 1 (kgat_bi_interaction_mul_f32): A = H * HN formed while loading
// (P = H, second factor HN) - the th.mul of reference models.py:66, which the aggregation used to form in its
// epilogue at the price of a dependent X[v] load inside its edge loop (91 vs 78 us, profiles/
// r04_spmm_epilogue_probe.txt) and which costs this kernel one more coalesced stream; the rows of H can also be
// copied into a column slice of the readout on the way (ego block, models.py:159,168).  MODE 2 (training form): as
// 1, and the LeakyReLU output goes through dropout (mess_drop of reference models.py:70) before it is written
// and normalised.
struct EgoCopy {
  float* out;      // nullptr: off
  int64_t stride;  // row stride in floats
};
template <int DI, int DO, int MODE, bool VEC_NORM>
__global__ __launch_bounds__(256) void bi_interaction_kernel(
    int32_t n_rows, const float* __restrict__ P, const float* __restrict__ HN, const float* __restrict__ W2,
    float slope, uint32_t drop_threshold, float keep_scale, uint32_t seed, uint32_t index0,
    float* __restrict__ h_out, float* __restrict__ norm_out, int64_t norm_stride, const EgoCopy ego) {
  constexpr bool TRAIN = MODE == 2;
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

  // W2's fragments live in registers for the whole launch - except at 128 x 128, where they would
  // need 256 VGPRs: there every MFMA takes its fragment from the LDS copy (one conflict-free
  // ds_read_b32 each)
#ifndef KGAT_BI_W_LDS_ABOVE
#define KGAT_BI_W_LDS_ABOVE 128  // (A/B builds: 0 = fragments always from LDS, fewer registers, more wavefronts per SIMD)
#endif
  constexpr bool W_IN_LDS = KS * KT > KGAT_BI_W_LDS_ABOVE;
  float wreg[W_IN_LDS ? 1 : KS][W_IN_LDS ? 1 : KT];
  if (!W_IN_LDS) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < KT; ++c) wreg[W_IN_LDS ? 0 : s][W_IN_LDS ? 0 : c] = s_w[(s * KT + c) * kWave + lane];
  }

  auto load_a = [&](int32_t t, float (&a)[KS]) {
    int32_t ra = (t << 4) + i;
    ra = ra < n_rows ? ra : n_rows - 1;
    const float4* pa = reinterpret_cast<const float4*>(P + (size_t)ra * DI) + q;
#pragma unroll
    for (int m = 0; m < DI / 16; ++m) {
      const float4 v = ld_row4(pa + m * 4);
      a[4 * m + 0] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
    }
    if (MODE >= 1) {
      if (MODE == 1 && ego.out != nullptr && (t << 4) + i < n_rows) {
        float4* pe = reinterpret_cast<float4*>(ego.out + (size_t)ra * ego.stride) + q;
#pragma unroll
        for (int m = 0; m < DI / 16; ++m) st_final4(pe + m * 4, make_float4(a[4 * m + 0], a[4 * m + 1], a[4 * m + 2], a[4 * m + 3]));
      }
      const float4* pb = reinterpret_cast<const float4*>(HN + (size_t)ra * DI) + q;
#pragma unroll
      for (int m = 0; m < DI / 16; ++m) {
        const float4 v = ld_row4(pb + m * 4);
        a[4 * m + 0] *= v.x; a[4 * m + 1] *= v.y; a[4 * m + 2] *= v.z; a[4 * m + 3] *= v.w;
      }
    }
  };
  auto tile = [&](int32_t t, const float (&a)[KS]) {
    const int32_t row0 = t << 4;
    floatx4_d acc[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[c] = (floatx4_d){0.f, 0.f, 0.f, 0.f};
    // operands swapped (A = W2 fragment, B = the tile's rows): the accumulators hold Z^T, i.e.
    // acc[c][j] = Z[row0 + i][16c + 4q + j] - four consecutive columns per lane, so the results
    // leave as 16-byte stores (a quarter of the store instructions of the row-major result)
#ifdef KGAT_BI_STRIP_MFMA  // A/B builds (WRONG results): the launch without its matrix work - what the row streams alone take
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[c] = (floatx4_d){a[(4 * c) % KS], a[(4 * c + 1) % KS], a[(4 * c + 2) % KS], a[(4 * c + 3) % KS]};
#else
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < KT; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            W_IN_LDS ? s_w[(s * KT + c) * kWave + lane] : wreg[W_IN_LDS ? 0 : s][W_IN_LDS ? 0 : c], a[s], acc[c], 0, 0, 0);
#endif
    const int32_t row = row0 + i;
    // row norm: per 16-column tile the sum of squares over the row's four lanes (i, q = 0..3), then the tiles'
    // partials in tile order - the order of the fused aggregation + dense launch (kgat_spmm_impl.h: tile_ssq),
    // whose wavefronts each own one column tile, so the two paths give the same bits
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = acc[c][j];
        z = z >= 0.f ? z : z * slope;
        if (TRAIN)
          z = drop_keep(seed, index0 + (uint32_t)row * (uint32_t)DO + (uint32_t)(16 * c + 4 * q + j), drop_threshold)
                  ? z * keep_scale : 0.f;
        acc[c][j] = z;
        part = j == 0 ? z * z : fmaf(z, z, part);
      }
      part += __shfl_xor(part, 16, kWave);
      part += __shfl_xor(part, 32, kWave);
      ss = c == 0 ? part : ss + part;
    }
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // one division per row; the 4 x KT values are scaled by it
    if (row < n_rows) {
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const float z0 = acc[c][0], z1 = acc[c][1], z2 = acc[c][2], z3 = acc[c][3];
        if (h_out) *reinterpret_cast<float4*>(h_out + (size_t)row * DO + 16 * c + 4 * q) = make_float4(z0, z1, z2, z3);
        if (norm_out) {
          float* dst = norm_out + (size_t)row * norm_stride + 16 * c + 4 * q;
          if (VEC_NORM) {
            st_final4(reinterpret_cast<float4*>(dst), make_float4(z0 * inv, z1 * inv, z2 * inv, z3 * inv));
          } else {
            dst[0] = z0 * inv; dst[1] = z1 * inv; dst[2] = z2 * inv; dst[3] = z3 * inv;
          }
        }
      }
    }
  };

  // The rows of the next PF - 1 tiles are requested before a tile is computed.  Round 4 tried a deeper ring (PF = 4,
  // the grid of two workgroups per CU leaves the registers free), W2's fragments from LDS instead of registers
  // with 4 and 8 workgroups per CU, and a 1,024-block grid: 30.7-35.7 us against 31.0 at 64 -> 64, 18.1-20.1
  // against 19.0 at 64 -> 32, 13.7-14.0 against 13.9 at 32 -> 16 (profiles/r04_bi_probe.txt) - the launch is not
  // bound by the latency of its row fetches, nor by resident wavefronts; dropping either output saves 4-5 us.
#ifndef KGAT_BI_PREFETCH
#define KGAT_BI_PREFETCH 2
#endif
  constexpr int PF = KS * KGAT_BI_PREFETCH <= 64 ? KGAT_BI_PREFETCH : (64 / KS >= 2 ? 64 / KS : 2);  // <= 64 VGPRs of rows in flight
  float a[PF][KS];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (t_begin + p < t_end) load_a(t_begin + p, a[p]);
  for (int32_t t = t_begin; t < t_end; t += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      if (t + p < t_end) {
        tile(t + p, a[p]);
        if (t + p + PF < t_end) load_a(t + p + PF, a[p]);
      }
    }
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

// The concatenated readout of Model.gnn (reference models.py:159-168: [h0 | normalize(h1) | ...])
// from separately held blocks in one pass: a 16-lane group per row walks the blocks, 16 bytes per
// lane, normalising where asked.  Used where the layers' rows come back from the multi-GPU exchange
// (one launch instead of one normalisation per layer plus the copy of the ego block).
constexpr int kMaxReadoutBlocks = 8;
struct ReadoutBlocks {
  const float* ptr[kMaxReadoutBlocks];
  int width[kMaxReadoutBlocks];
  int normalize[kMaxReadoutBlocks];
  int n;
};
__global__ __launch_bounds__(256) void readout_concat_kernel(int64_t n_rows, ReadoutBlocks b, float* __restrict__ out,
                                                             int64_t out_stride) {
  const int sub = threadIdx.x >> 4, sl = threadIdx.x & 15;
  for (int64_t row = (int64_t)blockIdx.x * 16 + sub; row < n_rows; row += (int64_t)gridDim.x * 16) {
    float* o = out + (size_t)row * out_stride;
    for (int k = 0; k < b.n; ++k) {
      const int w = b.width[k];
      const float4* xr = reinterpret_cast<const float4*>(b.ptr[k] + (size_t)row * w);
      float4 v[2];
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {  // widths up to 128: two float4 per lane
        const int c = sl + 16 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * c < w) v[j] = xr[c];
        ss = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, ss))));
      }
      float inv = 1.f;
      if (b.normalize[k]) inv = 1.0f / fmaxf(sqrtf(row16_sum_d(ss)), 1e-12f);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = sl + 16 * j;
        if (4 * c < w) {
          float4 r = v[j];
          if (b.normalize[k]) r = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
          *reinterpret_cast<float4*>(o + 4 * c) = r;
        }
      }
      o += w;
    }
  }
}

// Narrow widths (d_in or d_out below the 16 columns of an MFMA tile: configs[0]'s 8 -> 8 layer): one LANE per
// row, W2 (<= 1,024 floats) broadcast from LDS, the row's d_out results in registers.  Same epilogue as the MFMA
// kernel (LeakyReLU, hash dropout in the training form, un-normalised rows + L2-normalised copy).  Replaces the
// torch sequence Linear / leaky_relu / norm / cat (seven launches, ~35 us on the last-fm graph) by one of ~5 us.
template <int DI, int DO, int MODE>
__global__ __launch_bounds__(256) void bi_interaction_small_kernel(
    int32_t n_rows, const float* __restrict__ P, const float* __restrict__ HN, const float* __restrict__ W2,
    float slope, uint32_t drop_threshold, float keep_scale, uint32_t seed, uint32_t index0,
    float* __restrict__ h_out, float* __restrict__ norm_out, int64_t norm_stride, const EgoCopy ego) {
  constexpr bool TRAIN = MODE == 2;
  __shared__ float s_w[DO * DI];
  for (int idx = threadIdx.x; idx < DO * DI; idx += 256) s_w[idx] = W2[idx];
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n_rows; row += (int64_t)gridDim.x * 256) {
    float a[DI];
    const float4* pa = reinterpret_cast<const float4*>(P + (size_t)row * DI);
#pragma unroll
    for (int m = 0; m < DI / 4; ++m) {
      const float4 v = pa[m];
      a[4 * m] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
    }
    if (MODE >= 1) {
      if (MODE == 1 && ego.out != nullptr) {
        float4* pe = reinterpret_cast<float4*>(ego.out + (size_t)row * ego.stride);
#pragma unroll
        for (int m = 0; m < DI / 4; ++m) pe[m] = make_float4(a[4 * m], a[4 * m + 1], a[4 * m + 2], a[4 * m + 3]);
      }
      const float4* pb = reinterpret_cast<const float4*>(HN + (size_t)row * DI);
#pragma unroll
      for (int m = 0; m < DI / 4; ++m) {
        const float4 v = pb[m];
        a[4 * m] *= v.x; a[4 * m + 1] *= v.y; a[4 * m + 2] *= v.z; a[4 * m + 3] *= v.w;
      }
    }
    float z[DO];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < DO; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < DI; ++k) acc = fmaf(a[k], s_w[j * DI + k], acc);
      acc = acc >= 0.f ? acc : acc * slope;
      if (TRAIN) acc = drop_keep(seed, index0 + (uint32_t)row * (uint32_t)DO + (uint32_t)j, drop_threshold) ? acc * keep_scale : 0.f;
      z[j] = acc;
      ss = fmaf(acc, acc, ss);
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(|x|, eps)
    if (h_out) {
      float4* ph = reinterpret_cast<float4*>(h_out + (size_t)row * DO);
#pragma unroll
      for (int m = 0; m < DO / 4; ++m) ph[m] = make_float4(z[4 * m], z[4 * m + 1], z[4 * m + 2], z[4 * m + 3]);
    }
    if (norm_out) {
      float* pn = norm_out + (size_t)row * norm_stride;
#pragma unroll
      for (int j = 0; j < DO; ++j) pn[j] = z[j] * inv;
    }
  }
}

struct DropArgs {
  uint32_t threshold = 0;
  float keep_scale = 1.f;
  uint32_t seed = 0;
  uint32_t index0 = 0;  // element index of (row 0, column 0): row0 * d_out for a row range of a larger matrix
};

static DropArgs drop_args(float p, uint64_t seed, int64_t row0, int d_out) {
  DropArgs a;
  a.index0 = (uint32_t)((uint64_t)row0 * (uint64_t)d_out);
  double t = (double)p * 4294967296.0;
  a.threshold = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  a.keep_scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  a.seed = (uint32_t)(seed ^ (seed >> 32));
  return a;
}

template <int DI, int DO>
static int launch_bi(int64_t n_rows, const float* P, const float* HN, const float* W2, float slope,
                     const DropArgs& dr, float* h_out, float* norm_out, int64_t norm_stride, hipStream_t st,
                     int mode, const EgoCopy ego) {
  const int64_t tiles = (n_rows + 15) / 16;
  int64_t blocks = (tiles + 3) / 4;  // at least one tile per wave ...
#ifndef KGAT_BI_MAX_BLOCKS
#define KGAT_BI_MAX_BLOCKS 512
#endif
  if (blocks > KGAT_BI_MAX_BLOCKS) blocks = KGAT_BI_MAX_BLOCKS;    // ... two workgroups per CU (each stages W2 once; measured 256: 22.7, 512: 21.2, 1024: 22.1, 2048: 24.1 us avg)
  // 16-byte stores into the normalised copy need its slice 16-byte aligned with a row stride that keeps it so
  const bool vec = norm_out == nullptr ||
                   ((reinterpret_cast<uintptr_t>(norm_out) & 15u) == 0 && norm_stride % 4 == 0);
#define KGAT_BI_LAUNCH(MD, VEC)                                                                                     \
  hipLaunchKernelGGL((bi_interaction_kernel<DI, DO, MD, VEC>), dim3((unsigned)blocks), dim3(256), 0, st,            \
                     (int32_t)n_rows, P, HN, W2, slope, dr.threshold, dr.keep_scale, dr.seed, dr.index0, h_out,     \
                     norm_out, norm_stride, ego)
  if (mode == 2) {
    if (vec) KGAT_BI_LAUNCH(2, true); else KGAT_BI_LAUNCH(2, false);
  } else if (mode == 1) {
    if (vec) KGAT_BI_LAUNCH(1, true); else KGAT_BI_LAUNCH(1, false);
  } else {
    if (vec) KGAT_BI_LAUNCH(0, true); else KGAT_BI_LAUNCH(0, false);
  }
#undef KGAT_BI_LAUNCH
  KGAT_CHECK_LAUNCH("bi_interaction");
  return KGAT_OK;
}

template <int DI, int DO>
static int launch_bi_small(int64_t n_rows, const float* P, const float* HN, const float* W2, float slope,
                           const DropArgs& dr, float* h_out, float* norm_out, int64_t norm_stride, hipStream_t st,
                           int mode, const EgoCopy ego) {
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
#define KGAT_BI_SMALL_LAUNCH(MD)                                                                                          \
  hipLaunchKernelGGL((bi_interaction_small_kernel<DI, DO, MD>), dim3((unsigned)blocks), dim3(256), 0, st, (int32_t)n_rows, \
                     P, HN, W2, slope, dr.threshold, dr.keep_scale, dr.seed, dr.index0, h_out, norm_out, norm_stride, ego)
  if (mode == 2) KGAT_BI_SMALL_LAUNCH(2);
  else if (mode == 1) KGAT_BI_SMALL_LAUNCH(1);
  else KGAT_BI_SMALL_LAUNCH(0);
#undef KGAT_BI_SMALL_LAUNCH
  KGAT_CHECK_LAUNCH("bi_interaction_small");
  return KGAT_OK;
}

// Backward head of the training layer, one 16-lane group per row.  y = the layer's (dropped)
// output, saved by the forward.  dZ = [gA + gB + normalize_bwd(g_norm; y)] * keep/(1-p) *
// LeakyReLU'(Z), with sign(Z) = sign(y) where kept and the mask recomputed from the hash.
__global__ __launch_bounds__(256) void bi_bwd_pre_kernel(int64_t n_rows, int d, const float* __restrict__ y,
                                                         const float* __restrict__ gA, const float* __restrict__ gB,
                                                         const float* __restrict__ g_norm, int64_t g_norm_stride,
                                                         float slope, uint32_t drop_threshold, float keep_scale,
                                                         uint32_t seed, uint32_t index0, float* __restrict__ dZ) {
  const int sub = threadIdx.x >> 4, sl = threadIdx.x & 15;
  for (int64_t row = (int64_t)blockIdx.x * 16 + sub; row < n_rows; row += (int64_t)gridDim.x * 16) {
    const float* yr = y + (size_t)row * d;
    float yv[8], gn[8];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = sl + 16 * c;
      yv[c] = i < d ? yr[i] : 0.f;
      gn[c] = (g_norm && i < d) ? g_norm[(size_t)row * g_norm_stride + i] : 0.f;
      ss = fmaf(yv[c], yv[c], ss);
      dot = fmaf(yv[c], gn[c], dot);
    }
    ss = row16_sum_d(ss);
    dot = row16_sum_d(dot);
    const float nrm = sqrtf(ss);
    const bool clamped = nrm < 1e-12f;  // F.normalize: x / max(|x|, eps)
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    const float proj = clamped ? 0.f : dot * inv * inv;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = sl + 16 * c;
      if (i < d) {
        float g = (gn[c] - yv[c] * proj) * inv;
        if (gA) g += gA[(size_t)row * d + i];
        if (gB) g += gB[(size_t)row * d + i];
        const bool keep = drop_keep(seed, index0 + (uint32_t)row * (uint32_t)d + (uint32_t)i, drop_threshold);
        dZ[(size_t)row * d + i] = keep ? g * keep_scale * (yv[c] > 0.f ? 1.f : slope) : 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(256) void mul2_kernel(int64_t n4, const float4* __restrict__ A, const float4* __restrict__ B,
                                                   const float4* __restrict__ C, float4* __restrict__ AB,
                                                   float4* __restrict__ AC) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = A[i], b = B[i], c = C[i];
    AB[i] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
    AC[i] = make_float4(a.x * c.x, a.y * c.y, a.z * c.z, a.w * c.w);
  }
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

int kgat_readout_concat_f32(int64_t n_rows, int n_blocks, const float* const* blocks, const int* widths,
                            const int* normalize, float* out, int64_t out_stride, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_blocks > 0 && n_blocks <= kMaxReadoutBlocks, "readout_concat: bad size");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(blocks && widths && normalize && out, "readout_concat: null pointer");
  ReadoutBlocks b;
  int64_t total = 0;
  for (int k = 0; k < n_blocks; ++k) {
    KGAT_CHECK_ARG(blocks[k] && widths[k] > 0 && widths[k] <= 128 && widths[k] % 4 == 0,
                   "readout_concat: block widths must be multiples of 4 up to 128");
    KGAT_CHECK_ARG((reinterpret_cast<uintptr_t>(blocks[k]) & 15u) == 0, "readout_concat: blocks must be 16-byte aligned");
    b.ptr[k] = blocks[k]; b.width[k] = widths[k]; b.normalize[k] = normalize[k];
    total += widths[k];
  }
  b.n = n_blocks;
  KGAT_CHECK_ARG(out_stride >= total && out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0,
                 "readout_concat: out must be 16-byte aligned with a row stride that is a multiple of 4 floats");
  int64_t nb = (n_rows + 15) / 16;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(readout_concat_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), n_rows, b, out,
                     out_stride);
  KGAT_CHECK_LAUNCH("readout_concat");
  return KGAT_OK;
}

int kgat_bi_interaction_supported(int d_in, int d_out) {
  auto ok = [](int d) { return d == 16 || d == 32 || d == 64 || d == 128; };
  auto narrow = [](int d) { return d == 4 || d == 8; };
  auto small = [](int d) { return d == 4 || d == 8 || d == 16 || d == 32; };
  return (ok(d_in) && ok(d_out)) || (narrow(d_in) && small(d_out)) || (small(d_in) && narrow(d_out));
}

static int bi_dispatch(int64_t n_rows, int d_in, int d_out, const float* P, const float* HN, const float* W2,
                       float negative_slope, const DropArgs& dr, float* h_out, float* norm_out,
                       int64_t norm_stride, hipStream_t st, int mode, const EgoCopy ego = EgoCopy{nullptr, 0}) {
#define KGAT_BI_CASE(DI, DO) \
  if (d_in == DI && d_out == DO) \
    return launch_bi<DI, DO>(n_rows, P, HN, W2, negative_slope, dr, h_out, norm_out, norm_stride, st, mode, ego);
  KGAT_BI_CASE(16, 16) KGAT_BI_CASE(16, 32) KGAT_BI_CASE(16, 64) KGAT_BI_CASE(16, 128)
  KGAT_BI_CASE(32, 16) KGAT_BI_CASE(32, 32) KGAT_BI_CASE(32, 64) KGAT_BI_CASE(32, 128)
  KGAT_BI_CASE(64, 16) KGAT_BI_CASE(64, 32) KGAT_BI_CASE(64, 64) KGAT_BI_CASE(64, 128)
  KGAT_BI_CASE(128, 16) KGAT_BI_CASE(128, 32) KGAT_BI_CASE(128, 64) KGAT_BI_CASE(128, 128)
#undef KGAT_BI_CASE
#define KGAT_BI_SMALL(DI, DO) \
  if (d_in == DI && d_out == DO) \
    return launch_bi_small<DI, DO>(n_rows, P, HN, W2, negative_slope, dr, h_out, norm_out, norm_stride, st, mode, ego);
  KGAT_BI_SMALL(4, 4) KGAT_BI_SMALL(4, 8) KGAT_BI_SMALL(4, 16) KGAT_BI_SMALL(4, 32)
  KGAT_BI_SMALL(8, 4) KGAT_BI_SMALL(8, 8) KGAT_BI_SMALL(8, 16) KGAT_BI_SMALL(8, 32)
  KGAT_BI_SMALL(16, 4) KGAT_BI_SMALL(16, 8) KGAT_BI_SMALL(32, 4) KGAT_BI_SMALL(32, 8)
#undef KGAT_BI_SMALL
  set_error("bi_interaction: unsupported widths %d -> %d", d_in, d_out);
  return KGAT_E_UNSUPPORTED;
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
  return bi_dispatch(n_rows, d_in, d_out, P, nullptr, W2, negative_slope, DropArgs(), h_out, norm_out, norm_stride,
                     as_stream(stream), 0);
}

int kgat_bi_interaction_mul_f32(int64_t n_rows, int d_in, int d_out, const float* H, const float* HN, const float* W2,
                                float negative_slope, float* h_out, float* norm_out, int64_t norm_stride,
                                float* self_out, int64_t self_stride, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_rows < INT32_MAX, "bi_interaction_mul: bad row count");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(H && HN && W2 && (h_out || norm_out), "bi_interaction_mul: null pointer");
  KGAT_CHECK_ARG(norm_out == nullptr || norm_stride >= d_out, "bi_interaction_mul: bad norm_stride");
  KGAT_CHECK_ARG(self_out == nullptr || (self_stride >= d_in && self_stride % 4 == 0 &&
                                         (reinterpret_cast<uintptr_t>(self_out) & 15u) == 0),
                 "bi_interaction_mul: self_out must be 16-byte aligned with a row stride that is a multiple of 4 floats >= d_in");
  if (!kgat_bi_interaction_supported(d_in, d_out)) {
    set_error("bi_interaction_mul: unsupported widths %d -> %d", d_in, d_out);
    return KGAT_E_UNSUPPORTED;
  }
  return bi_dispatch(n_rows, d_in, d_out, H, HN, W2, negative_slope, DropArgs(), h_out, norm_out, norm_stride,
                     as_stream(stream), 1, EgoCopy{self_out, self_stride});
}

int kgat_bi_interaction_train_f32(int64_t n_rows, int d_in, int d_out, const float* H, const float* HN,
                                  const float* W2, float negative_slope, float drop_p, uint64_t seed, int64_t row0,
                                  float* h_out, float* norm_out, int64_t norm_stride, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_rows < INT32_MAX && row0 >= 0 &&
                     (uint64_t)(row0 + n_rows) * (uint64_t)d_out < (1ull << 32),
                 "bi_interaction_train: bad row count");
  KGAT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "bi_interaction_train: dropout probability outside [0, 1)");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(H && HN && W2 && h_out, "bi_interaction_train: null pointer");
  KGAT_CHECK_ARG(norm_out == nullptr || norm_stride >= d_out, "bi_interaction_train: bad norm_stride");
  if (!kgat_bi_interaction_supported(d_in, d_out)) {
    set_error("bi_interaction_train: unsupported widths %d -> %d", d_in, d_out);
    return KGAT_E_UNSUPPORTED;
  }
  return bi_dispatch(n_rows, d_in, d_out, H, HN, W2, negative_slope, drop_args(drop_p, seed, row0, d_out), h_out,
                     norm_out, norm_stride, as_stream(stream), 2);
}

int kgat_bi_interaction_bwd_pre_f32(int64_t n_rows, int d_out, const float* h_out, const float* grad_a,
                                    const float* grad_b, const float* grad_norm, int64_t grad_norm_stride,
                                    float negative_slope, float drop_p, uint64_t seed, int64_t row0, float* grad_z,
                                    kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && d_out > 0 && d_out <= 128 && row0 >= 0 &&
                     (uint64_t)(row0 + n_rows) * (uint64_t)d_out < (1ull << 32),
                 "bi_interaction_bwd_pre: bad size");
  KGAT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "bi_interaction_bwd_pre: dropout probability outside [0, 1)");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(h_out && grad_z, "bi_interaction_bwd_pre: null pointer");
  KGAT_CHECK_ARG(grad_norm == nullptr || grad_norm_stride >= d_out, "bi_interaction_bwd_pre: bad stride");
  const DropArgs dr = drop_args(drop_p, seed, row0, d_out);
  int64_t blocks = (n_rows + 15) / 16;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bi_bwd_pre_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n_rows, d_out, h_out,
                     grad_a, grad_b, grad_norm, grad_norm_stride, negative_slope, dr.threshold, dr.keep_scale, dr.seed,
                     dr.index0, grad_z);
  KGAT_CHECK_LAUNCH("bi_bwd_pre");
  return KGAT_OK;
}

int kgat_mul2_f32(int64_t n, const float* a, const float* b, const float* c, float* ab, float* ac,
                  kgat_stream_t stream) {
  KGAT_CHECK_ARG(n >= 0 && n % 4 == 0, "mul2: length must be a multiple of 4");
  if (n == 0) return KGAT_OK;
  KGAT_CHECK_ARG(a && b && c && ab && ac, "mul2: null pointer");
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mul2_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n / 4,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                     reinterpret_cast<const float4*>(c), reinterpret_cast<float4*>(ab), reinterpret_cast<float4*>(ac));
  KGAT_CHECK_LAUNCH("mul2");
  return KGAT_OK;
}

}  // extern "C"
