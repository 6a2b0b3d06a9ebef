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

constexpr bool kBiMulAtLoad = false;   // (true: the first form of MODE >= 1, h * h_N multiplied in the load step - round 4's A/B)
constexpr int kBiNtLoads = 0;   // A/B builds, bit mask: 1 = the rows of H (P), 2 = the rows of HN as non-temporal loads
template <bool NT>
__device__ __forceinline__ float4 ld_row4(const float4* p) {
  if constexpr (NT) {
    const floatx4_d v = __builtin_nontemporal_load(reinterpret_cast<const floatx4_d*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
  } else {
    return *p;
  }
}
__device__ __forceinline__ void st_final4(float4* p, const float4& v) {
  const floatx4_d x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<floatx4_d*>(p));
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
// DEFER (MODE 1, kgat_bi_interaction_mul_deferred_f32): HN comes from an aggregation launched with
// KGAT_SPMM_DEFER_FINISH - its second launch (kgat_spmm_impl.h: spmm_finish_kernel) did not run, so the rows that
// are the first or the last row of one of its edge tiles, and the rows without in-edges, are not in HN: this
// kernel forms them on the way, from the row offsets and the tiles' boundary partials in the aggregation's
// workspace, in the finish launch's order of additions (same bits).  Stand-alone the second launch costs the
// aggregation 5-6 us and forming its rows costs this kernel 2.5-3.4 (profiles/r04_defer_probe.txt); in the step the pair
// saves 7 us of the 13 the three launches cost (profiles/r04_deferred_finish_step_ab.txt).
// (Requesting a chain's first follower in the load step as well - one stage instead of two at d_in >= 64 for the
// registers - changes nothing: 0.4093 vs 0.4088 ms per step.)
struct DeferredRows {
  const int32_t* indptr;  // row offsets of THIS call's row 0 .. n_rows (CSR positions)
  const float4* bpart;    // per tile two partial rows (first row's, last row's), LPR = d_in / 4 float4 each
  int32_t e0, e1;         // CSR positions of the row range
  int32_t te_shift;       // log2(edges per tile)
};
constexpr int kDeferLongChain = 8;  // = spmm_finish_kernel's kLongChain
template <int DI, int DO, int MODE, bool VEC_NORM, bool DEFER = false>
__global__ __launch_bounds__(256) void bi_interaction_kernel(
    int32_t n_rows, const float* __restrict__ P, const float* __restrict__ HN, const float* __restrict__ W2,
    float slope, uint32_t drop_threshold, float keep_scale, uint32_t seed, uint32_t index0,
    float* __restrict__ h_out, float* __restrict__ norm_out, int64_t norm_stride, const EgoCopy ego,
    const DeferredRows df) {
  constexpr bool TRAIN = MODE == 2;
  static_assert(!DEFER || (MODE == 1 && !kBiMulAtLoad), "the deferred rows go with the late product");
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
  // (the wavefront's index through readfirstlane: its tile range is then held in SGPRs and the tile loop's branches
  // are scalar - as a per-lane value the loop was compiled as divergent control flow, exec-masked block by block)
  const int64_t wv = (int64_t)blockIdx.x * (256 / kWave) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int32_t n_tiles = (n_rows + 15) >> 4;
  const int32_t t_begin = (int32_t)((int64_t)n_tiles * wv / n_waves);
  const int32_t t_end = (int32_t)((int64_t)n_tiles * (wv + 1) / n_waves);
  if (t_begin >= t_end) return;

  // W2's fragments live in registers for the whole launch - except at 128 x 128, where they would
  // need 256 VGPRs: there every MFMA takes its fragment from the LDS copy (one conflict-free
  // ds_read_b32 each)
constexpr int kBiWLdsAbove = 128;  // (A/B builds: 0 = fragments always from LDS, fewer registers, more wavefronts per SIMD)
  constexpr bool W_IN_LDS = KS * KT > kBiWLdsAbove;
  float wreg[W_IN_LDS ? 1 : KS][W_IN_LDS ? 1 : KT];
  if (!W_IN_LDS) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < KT; ++c) wreg[W_IN_LDS ? 0 : s][W_IN_LDS ? 0 : c] = s_w[(s * KT + c) * kWave + lane];
  }

  // MODE >= 1: the rows of H and of HN are REQUESTED here and multiplied when the tile is computed (round 4, second
  // form).  The first form multiplied here, which put the waits for both row sets - one after the other, the ego
  // copy in between - into the load step: two exposed memory round trips per tile with nothing else of the
  // wavefront in flight, and no overlap with the previous tile's matrix work whatever kBiPrefetch said
  // (kBiMulAtLoad = true restores that form.)
  constexpr bool LATE_MUL = MODE >= 1 && !kBiMulAtLoad;
constexpr int kBiPrefetch = 2;
  constexpr int PF = KS * kBiPrefetch <= 64 ? kBiPrefetch : (64 / KS >= 2 ? 64 / KS : 2);  // <= 64 VGPRs of rows in flight (x 2 with HN)
  // DEFER: per stage, the offsets of the row this lane loads NEXT (requested one load step ahead), and what the load
  // step found out for the tile step: nf = followers of the row's chain of tile partials (-1: a row without in-edges,
  // 0: an ordinary row or a one-partial chain), bh = the chain's head tile
  constexpr int LPR = DI / 4;
  struct Defer { int32_t rb, re, nf, bh, slot; };
  auto row_offsets = [&](int32_t t, Defer& d) {
    int32_t ra = (t << 4) + i;
    ra = ra < n_rows ? ra : n_rows - 1;
    d.rb = df.indptr[ra];
    d.re = df.indptr[ra + 1];
  };
  auto load_a = [&](int32_t t, float (&a)[KS], float (&b)[LATE_MUL ? KS : 1], Defer& d) {
    int32_t ra = (t << 4) + i;
    ra = ra < n_rows ? ra : n_rows - 1;
    const float4* pa = reinterpret_cast<const float4*>(P + (size_t)ra * DI) + q;
#pragma unroll
    for (int m = 0; m < DI / 16; ++m) {
      const float4 v = ld_row4<(kBiNtLoads & 1) != 0>(pa + m * 4);
      a[4 * m + 0] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
    }
    if constexpr (LATE_MUL) {
      const float4* pb = reinterpret_cast<const float4*>(HN + (size_t)ra * DI) + q;
      if constexpr (DEFER) {
        // a row the aggregation's tiles left as partials: the head partial has an HN row's layout - it is requested
        // in the row's place, the followers when the tile is computed
        const int32_t rb = d.rb - df.e0, re = d.re - df.e0;
        const int32_t te_mask = (1 << df.te_shift) - 1;
        const int32_t bh = rb >> df.te_shift, bl = (re - 1) >> df.te_shift;
        const bool empty = rb == re;
        const bool lo_al = (rb & te_mask) == 0;
        const bool partial = !empty && (bh != bl || lo_al || (re & te_mask) == 0 || d.re == df.e1);
        if (partial) pb = df.bpart + ((size_t)bh * 2 + (lo_al ? 0 : 1)) * LPR + q;
        d.nf = empty ? -1 : (partial ? bl - bh : 0);
        d.bh = bh;
        d.slot = lo_al ? 0 : 1;
      }
#pragma unroll
      for (int m = 0; m < DI / 16; ++m) {
        const float4 v = ld_row4<(kBiNtLoads & 2) != 0>(pb + m * 4);
        b[4 * m + 0] = v.x; b[4 * m + 1] = v.y; b[4 * m + 2] = v.z; b[4 * m + 3] = v.w;
      }
      if constexpr (DEFER) row_offsets(t + PF, d);  // (clamped to the last row past the end)
    } else if (MODE >= 1) {
      if (MODE >= 1 && ego.out != nullptr && (t << 4) + i < n_rows) {
        float4* pe = reinterpret_cast<float4*>(ego.out + (size_t)ra * ego.stride) + q;
#pragma unroll
        for (int m = 0; m < DI / 16; ++m) st_final4(pe + m * 4, make_float4(a[4 * m + 0], a[4 * m + 1], a[4 * m + 2], a[4 * m + 3]));
      }
      const float4* pb = reinterpret_cast<const float4*>(HN + (size_t)ra * DI) + q;
#pragma unroll
      for (int m = 0; m < DI / 16; ++m) {
        const float4 v = ld_row4<(kBiNtLoads & 2) != 0>(pb + m * 4);
        a[4 * m + 0] *= v.x; a[4 * m + 1] *= v.y; a[4 * m + 2] *= v.z; a[4 * m + 3] *= v.w;
      }
    }
  };
  auto tile = [&](int32_t t, float (&a)[KS], float (&b)[LATE_MUL ? KS : 1], const int32_t nf, const int32_t bh, const int32_t slot) {
    const int32_t row0 = t << 4;
    if constexpr (DEFER) {
      if (__builtin_amdgcn_ballot_w64(nf != 0) != 0ull) {  // (uniform: about a third of the 16-row tiles)
        const bool is_long = nf >= kDeferLongChain;
        if (nf < 0) {
#pragma unroll
          for (int s = 0; s < KS; ++s) b[s] = 0.f;
        } else if (nf > 0 && !is_long) {
          // acc = head partial; acc += the followers' first-row partials, in tile order (spmm_finish_kernel)
          const float4* pf = df.bpart + ((size_t)(bh + 1) * 2) * LPR + q;
          for (int32_t k = 0; k < nf; ++k, pf += 2 * LPR) {
#pragma unroll
            for (int m = 0; m < DI / 16; ++m) {
              const float4 v = pf[m * 4];
              b[4 * m + 0] += v.x; b[4 * m + 1] += v.y; b[4 * m + 2] += v.z; b[4 * m + 3] += v.w;
            }
          }
        }
        // hub rows (a chain of more than kDeferLongChain tiles): the whole wavefront sums one row, as the finish
        // launch does - lane group g = lane / LPR takes the tiles head + g, head + g + SPW, ..., a fixed shuffle
        // tree adds the groups' sums - and hands the row to its four lanes
        unsigned long long todo = __builtin_amdgcn_ballot_w64(is_long && q == 0);
        if (todo) {
          constexpr int SPW = kWave / LPR;
          const int g = lane / LPR, sl = lane % LPR;
          while (todo) {
            const int src = __ffsll((long long)todo) - 1;  // (q == 0: the lane index is the row's i)
            todo &= todo - 1;
            const int32_t bo = __shfl(bh, src, kWave);
            const int32_t bl = bo + __shfl(nf, src, kWave);
            const int so = __shfl(slot, src, kWave);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int32_t bb = bo + g;
            constexpr int U = 8;
            for (; bb + (U - 1) * SPW <= bl; bb += U * SPW) {
              float4 v[U];
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int32_t tt = bb + u * SPW;
                v[u] = df.bpart[((size_t)tt * 2 + ((tt == bo) ? so : 0)) * LPR + sl];
              }
#pragma unroll
              for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; bb <= bl; bb += SPW) {
              const float4 v = df.bpart[((size_t)bb * 2 + ((bb == bo) ? so : 0)) * LPR + sl];
              acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
#pragma unroll
            for (int off = LPR; off < kWave; off <<= 1) {
              acc.x += __shfl_xor(acc.x, off, kWave); acc.y += __shfl_xor(acc.y, off, kWave);
              acc.z += __shfl_xor(acc.z, off, kWave); acc.w += __shfl_xor(acc.w, off, kWave);
            }
            // every lane group now holds the row (lane sl: columns 4 sl .. 4 sl + 3); lane (i, q) takes 4 (4m + q)..
#pragma unroll
            for (int m = 0; m < DI / 16; ++m) {
              const float x = __shfl(acc.x, 4 * m + q, kWave), y = __shfl(acc.y, 4 * m + q, kWave);
              const float z = __shfl(acc.z, 4 * m + q, kWave), w = __shfl(acc.w, 4 * m + q, kWave);
              if (i == src) { b[4 * m + 0] = x; b[4 * m + 1] = y; b[4 * m + 2] = z; b[4 * m + 3] = w; }
            }
          }
        }
      }
    }
    if constexpr (LATE_MUL) {
      if (MODE >= 1 && ego.out != nullptr && row0 + i < n_rows) {
        float4* pe = reinterpret_cast<float4*>(ego.out + (size_t)(row0 + i) * ego.stride) + q;
#pragma unroll
        for (int m = 0; m < DI / 16; ++m) st_final4(pe + m * 4, make_float4(a[4 * m + 0], a[4 * m + 1], a[4 * m + 2], a[4 * m + 3]));
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) a[s] *= b[s];
    }
    floatx4_d acc[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[c] = (floatx4_d){0.f, 0.f, 0.f, 0.f};
    // operands swapped (A = W2 fragment, B = the tile's rows): the accumulators hold Z^T, i.e.
    // acc[c][j] = Z[row0 + i][16c + 4q + j] - four consecutive columns per lane, so the results
    // leave as 16-byte stores (a quarter of the store instructions of the row-major result)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < KT; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            W_IN_LDS ? s_w[(s * KT + c) * kWave + lane] : wreg[W_IN_LDS ? 0 : s][W_IN_LDS ? 0 : c], a[s], acc[c], 0, 0, 0);
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

  // Ring of PF stages: the rows of tile t + PF are requested once tile t is computed.  (Round 4 measured an explicit
  // two-buffer loop, load t + 1 / compute t, with unconditional load steps against it: stand-alone faster at 64 -> 64
  // (37.2 vs 38.9 us), slower at 32 -> 16 (14.8 vs 14.0); inside the step the ring wins, 0.4374 vs 0.4395 ms:
  // profiles/r04_bi_late_mul_ab.txt; that form is in the history, commit c6eba96 and before.  Round 4 had
  // also tried PF = 4, W2's fragments from LDS with 4 and 8 workgroups per CU and a 1,024-block grid:
  // profiles/r04_bi_probe.txt - the launch runs at the rate of a device copy of its bytes.)
  float a[PF][KS], b[PF][LATE_MUL ? KS : 1];
  Defer d[PF];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    d[p] = Defer{0, 0, 0, 0, 0};
    if (DEFER) row_offsets(t_begin + p, d[p]);
  }
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (t_begin + p < t_end) load_a(t_begin + p, a[p], b[p], d[p]);
  for (int32_t t = t_begin; t < t_end; t += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      if (t + p < t_end) {
        // (nf / bh by value: the load step below overwrites the stage's record)
        tile(t + p, a[p], b[p], d[p].nf, d[p].bh, d[p].slot);
        if (t + p + PF < t_end) load_a(t + p + PF, a[p], b[p], d[p]);
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
          st_final4(reinterpret_cast<float4*>(o + 4 * c), r);  // (the readout: nothing reads it again in the step)
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
      if (MODE >= 1 && ego.out != nullptr) {
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
                     int mode, const EgoCopy ego, const DeferredRows* defer = nullptr) {
  const int64_t tiles = (n_rows + 15) / 16;
  int64_t blocks = (tiles + 3) / 4;  // at least one tile per wave ...
constexpr int kBiMaxBlocks = 512;
  if (blocks > kBiMaxBlocks) blocks = kBiMaxBlocks;    // ... two workgroups per CU (each stages W2 once; measured 256: 22.7, 512: 21.2, 1024: 22.1, 2048: 24.1 us avg)
  // 16-byte stores into the normalised copy need its slice 16-byte aligned with a row stride that keeps it so
  const bool vec = norm_out == nullptr ||
                   ((reinterpret_cast<uintptr_t>(norm_out) & 15u) == 0 && norm_stride % 4 == 0);
  const DeferredRows no_defer{nullptr, nullptr, 0, 0, 0};
#define KGAT_BI_LAUNCH(MD, VEC)                                                                                     \
  hipLaunchKernelGGL((bi_interaction_kernel<DI, DO, MD, VEC>), dim3((unsigned)blocks), dim3(256), 0, st,            \
                     (int32_t)n_rows, P, HN, W2, slope, dr.threshold, dr.keep_scale, dr.seed, dr.index0, h_out,     \
                     norm_out, norm_stride, ego, no_defer)
  if (mode == 1 && defer != nullptr) {
    if (vec)
      hipLaunchKernelGGL((bi_interaction_kernel<DI, DO, 1, true, true>), dim3((unsigned)blocks), dim3(256), 0, st,
                         (int32_t)n_rows, P, HN, W2, slope, dr.threshold, dr.keep_scale, dr.seed, dr.index0, h_out,
                         norm_out, norm_stride, ego, *defer);
    else
      hipLaunchKernelGGL((bi_interaction_kernel<DI, DO, 1, false, true>), dim3((unsigned)blocks), dim3(256), 0, st,
                         (int32_t)n_rows, P, HN, W2, slope, dr.threshold, dr.keep_scale, dr.seed, dr.index0, h_out,
                         norm_out, norm_stride, ego, *defer);
  } else if (mode == 2) {
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

// Backward of the dense part towards the layer input (round 4, second half): grad_P = grad_z W2, then the two products
// the rest of the backward wants - T = grad_P * H (aggregated over the reversed CSR: the gradient through h_N) and
// GB = grad_P * HN (the gradient through the row's own features) - in one pass.  Replaces a library GEMM + kgat_mul2_f32:
// grad_P (N x d_in) is never written.  The forward kernel's structure with the weight read transposed:
// B fragments s_w[(s*KT + c)*64 + q*16 + i] = W2[k][16c + i], k = 16 (s >> 2) + 4q + (s & 3) the contraction index (a
// column of grad_z), 16c + i the output column; a lane ends with four consecutive columns of one row, where it also
// holds H and HN (requested with the grad_z rows).
template <int DK, int DN>
__global__ __launch_bounds__(256) void bi_bwd_input_kernel(int32_t n_rows, const float* __restrict__ GZ,
                                                          const float* __restrict__ W2, const float* __restrict__ H,
                                                          const float* __restrict__ HN, float* __restrict__ T,
                                                          float* __restrict__ GB) {
  constexpr int KS = DK / 4, KT = DN / 16;
  __shared__ float s_w[KS * KT * kWave];
  for (int idx = threadIdx.x * 4; idx < DK * DN; idx += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(W2 + idx);  // W2[k][j0 .. j0 + 3]
    const int k = idx / DN, j0 = idx % DN;
    const int s = (k >> 4) * 4 + (k & 3), q = (k >> 2) & 3;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = j0 + t;
      s_w[(s * KT + (j >> 4)) * kWave + q * 16 + (j & 15)] = vv[t];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x % kWave;
  const int i = lane & 15, q = lane >> 4;
  const int64_t n_waves = (int64_t)gridDim.x * (256 / kWave);
  const int64_t wv = (int64_t)blockIdx.x * (256 / kWave) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int32_t n_tiles = (n_rows + 15) >> 4;
  const int32_t t_begin = (int32_t)((int64_t)n_tiles * wv / n_waves);
  const int32_t t_end = (int32_t)((int64_t)n_tiles * (wv + 1) / n_waves);
  if (t_begin >= t_end) return;
  constexpr bool W_IN_LDS = KS * KT > 128;
  float wreg[W_IN_LDS ? 1 : KS][W_IN_LDS ? 1 : KT];
  if (!W_IN_LDS) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < KT; ++c) wreg[W_IN_LDS ? 0 : s][W_IN_LDS ? 0 : c] = s_w[(s * KT + c) * kWave + lane];
  }
  struct Rows { float g[KS]; float4 h[KT], hn[KT]; };
  auto load_rows = [&](int32_t t, Rows& r) {
    int32_t ra = (t << 4) + i;
    ra = ra < n_rows ? ra : n_rows - 1;
    const float4* pg = reinterpret_cast<const float4*>(GZ + (size_t)ra * DK) + q;
#pragma unroll
    for (int m = 0; m < DK / 16; ++m) {
      const float4 v = pg[m * 4];
      r.g[4 * m + 0] = v.x; r.g[4 * m + 1] = v.y; r.g[4 * m + 2] = v.z; r.g[4 * m + 3] = v.w;
    }
    const float4* ph = reinterpret_cast<const float4*>(H + (size_t)ra * DN) + q;
    const float4* pn = reinterpret_cast<const float4*>(HN + (size_t)ra * DN) + q;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      r.h[c] = ph[c * 4];
      r.hn[c] = pn[c * 4];
    }
  };
  auto tile = [&](int32_t t, const Rows& r) {
    floatx4_d acc[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[c] = (floatx4_d){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < KT; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            W_IN_LDS ? s_w[(s * KT + c) * kWave + lane] : wreg[W_IN_LDS ? 0 : s][W_IN_LDS ? 0 : c], r.g[s], acc[c], 0, 0, 0);
    const int32_t row = (t << 4) + i;
    if (row < n_rows) {
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const size_t off = (size_t)row * DN + 16 * c + 4 * q;
        *reinterpret_cast<float4*>(T + off) =
            make_float4(acc[c][0] * r.h[c].x, acc[c][1] * r.h[c].y, acc[c][2] * r.h[c].z, acc[c][3] * r.h[c].w);
        *reinterpret_cast<float4*>(GB + off) =
            make_float4(acc[c][0] * r.hn[c].x, acc[c][1] * r.hn[c].y, acc[c][2] * r.hn[c].z, acc[c][3] * r.hn[c].w);
      }
    }
  };
  Rows r0, r1;
  load_rows(t_begin, r0);
  for (int32_t t = t_begin; t < t_end; t += 2) {
    if (t + 1 < t_end) load_rows(t + 1, r1);
    tile(t, r0);
    if (t + 1 >= t_end) break;
    if (t + 2 < t_end) load_rows(t + 2, r0);
    tile(t + 1, r1);
  }
}

// Weight gradient of the dense part (round 4, second half): grad_W2 = grad_z^T (H * HN), a (d_out x d_in) result reduced
// over all N rows.  Replaces torch's H * HN (an N x d_in round trip) + a batched library GEMM over row slabs.  A
// workgroup walks 64-row slabs: the rows of grad_z and the product H * HN (formed on the way) are staged in LDS, row
// stride D + 16 floats so that the four rows of a k-step sit 16 banks apart; wavefront w owns the output tiles
// w, w + 4, ... (at 64 x 64: one column tile, all four row tiles - one B read per four MFMAs); the contraction index
// of v_mfma_f32_16x16x4_f32 is the ROW: A[m][k] = grad_z[r0 + k][16 cm + m], B[k][n] = P[r0 + k][16 cn + n].  Every
// workgroup writes its partial (d_out x d_in); the caller sums the partials (fixed order: reproducible).
template <int DO, int DI>
__global__ __launch_bounds__(256) void bi_bwd_weight_kernel(int32_t n_rows, const float* __restrict__ GZ,
                                                           const float* __restrict__ H, const float* __restrict__ HN,
                                                           float* __restrict__ partial) {
  constexpr int SLAB = 64;
  constexpr int LG = DO == 16 ? 16 : DO + 16, LP = DI == 16 ? 16 : DI + 16;
  constexpr int TM = DO / 16, TN = DI / 16, TT = TM * TN;
  constexpr int TPW = (TT + 3) / 4;                 // output tiles per wavefront
  constexpr int G4 = SLAB * DO / 4 / 256 > 0 ? SLAB * DO / 4 / 256 : 1;  // float4 of grad_z per thread per slab
  constexpr int P4 = SLAB * DI / 4 / 256 > 0 ? SLAB * DI / 4 / 256 : 1;
  __shared__ float s_g[SLAB * LG];
  __shared__ float s_p[SLAB * LP];
  const int tid = threadIdx.x, lane = tid % kWave, w = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int i = lane & 15, q = lane >> 4;
  const int32_t n_slabs = (n_rows + SLAB - 1) / SLAB;
  floatx4_d acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = (floatx4_d){0.f, 0.f, 0.f, 0.f};
  float4 g[G4], hh[P4], hn[P4];
  auto request = [&](int32_t slab) {  // the slab's rows into registers (rows past the end: zeros)
    const int32_t r0 = slab * SLAB;
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int e = (u * 256 + tid) * 4;            // element index inside the SLAB x DO block
      const int32_t r = r0 + e / DO;
      g[u] = (e < SLAB * DO && r < n_rows) ? *reinterpret_cast<const float4*>(GZ + (size_t)r * DO + e % DO)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < P4; ++u) {
      const int e = (u * 256 + tid) * 4;
      const int32_t r = r0 + e / DI;
      const bool in = e < SLAB * DI && r < n_rows;
      hh[u] = in ? *reinterpret_cast<const float4*>(H + (size_t)r * DI + e % DI) : make_float4(0.f, 0.f, 0.f, 0.f);
      hn[u] = in ? *reinterpret_cast<const float4*>(HN + (size_t)r * DI + e % DI) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int32_t slab = blockIdx.x;
  if (slab < n_slabs) request(slab);
  for (; slab < n_slabs; slab += gridDim.x) {
    __syncthreads();  // the previous slab's fragments have been read
#pragma unroll
    for (int u = 0; u < G4; ++u) {
      const int e = (u * 256 + tid) * 4;
      if (e < SLAB * DO) *reinterpret_cast<float4*>(&s_g[(e / DO) * LG + e % DO]) = g[u];
    }
#pragma unroll
    for (int u = 0; u < P4; ++u) {
      const int e = (u * 256 + tid) * 4;
      if (e < SLAB * DI)
        *reinterpret_cast<float4*>(&s_p[(e / DI) * LP + e % DI]) =
            make_float4(hh[u].x * hn[u].x, hh[u].y * hn[u].y, hh[u].z * hn[u].z, hh[u].w * hn[u].w);
    }
    __syncthreads();
    if (slab + (int32_t)gridDim.x < n_slabs) request(slab + gridDim.x);  // in flight while this slab is multiplied
    if (w < TT) {  // (narrow results have fewer tiles than wavefronts)
#pragma unroll 4
      for (int r0 = 0; r0 < SLAB; r0 += 4) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int tile = w + 4 * t;
          if (tile < TT) {
            const int cm = tile / TN, cn = tile % TN;
            const float a = s_g[(r0 + q) * LG + 16 * cm + i];
            const float b = s_p[(r0 + q) * LP + 16 * cn + i];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  // C[m = 4q + j][n = i] of tile (cm, cn) -> partial[block][16 cm + 4q + j][16 cn + i]
  float* out = partial + (size_t)blockIdx.x * DO * DI;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = w + 4 * t;
    if (tile < TT) {
      const int cm = tile / TN, cn = tile % TN;
#pragma unroll
      for (int j = 0; j < 4; ++j) out[(size_t)(16 * cm + 4 * q + j) * DI + 16 * cn + i] = acc[t][j];
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
                       int64_t norm_stride, hipStream_t st, int mode, const EgoCopy ego = EgoCopy{nullptr, 0},
                       const DeferredRows* defer = nullptr) {
#define KGAT_BI_CASE(DI, DO) \
  if (d_in == DI && d_out == DO) \
    return launch_bi<DI, DO>(n_rows, P, HN, W2, negative_slope, dr, h_out, norm_out, norm_stride, st, mode, ego, defer);
  KGAT_BI_CASE(16, 16) KGAT_BI_CASE(16, 32) KGAT_BI_CASE(16, 64) KGAT_BI_CASE(16, 128)
  KGAT_BI_CASE(32, 16) KGAT_BI_CASE(32, 32) KGAT_BI_CASE(32, 64) KGAT_BI_CASE(32, 128)
  KGAT_BI_CASE(64, 16) KGAT_BI_CASE(64, 32) KGAT_BI_CASE(64, 64) KGAT_BI_CASE(64, 128)
  KGAT_BI_CASE(128, 16) KGAT_BI_CASE(128, 32) KGAT_BI_CASE(128, 64) KGAT_BI_CASE(128, 128)
#undef KGAT_BI_CASE
  if (defer != nullptr) {
    set_error("bi_interaction_mul_deferred: widths %d -> %d are outside the MFMA kernel's {16, 32, 64, 128}", d_in, d_out);
    return KGAT_E_UNSUPPORTED;
  }
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

int kgat_bi_interaction_mul_deferred_f32(int64_t n_rows, int d_in, int d_out, const float* H, const float* HN,
                                         const float* W2, float negative_slope, float* h_out, float* norm_out,
                                         int64_t norm_stride, float* self_out, int64_t self_stride,
                                         const int32_t* indptr_rows, int64_t e_begin, int64_t e_end,
                                         const void* spmm_workspace, int tile_edges, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_rows < INT32_MAX, "bi_interaction_mul_deferred: bad row count");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(H && HN && W2 && (h_out || norm_out) && indptr_rows, "bi_interaction_mul_deferred: null pointer");
  KGAT_CHECK_ARG(e_begin >= 0 && e_end >= e_begin && e_end < INT32_MAX, "bi_interaction_mul_deferred: bad edge range");
  KGAT_CHECK_ARG(e_end == e_begin || spmm_workspace != nullptr, "bi_interaction_mul_deferred: null workspace");
  KGAT_CHECK_ARG(tile_edges > 0 && (tile_edges & (tile_edges - 1)) == 0,
                 "bi_interaction_mul_deferred: tile_edges must be kgat_spmm_tile_edges() of the aggregation (a power of two)");
  KGAT_CHECK_ARG(norm_out == nullptr || norm_stride >= d_out, "bi_interaction_mul_deferred: bad norm_stride");
  KGAT_CHECK_ARG(self_out == nullptr || (self_stride >= d_in && self_stride % 4 == 0 &&
                                         (reinterpret_cast<uintptr_t>(self_out) & 15u) == 0),
                 "bi_interaction_mul_deferred: self_out must be 16-byte aligned with a row stride that is a multiple of 4 floats >= d_in");
  int shift = 0;
  while ((1 << shift) < tile_edges) ++shift;
  const DeferredRows df{indptr_rows, static_cast<const float4*>(spmm_workspace), (int32_t)e_begin, (int32_t)e_end, shift};
  return bi_dispatch(n_rows, d_in, d_out, H, HN, W2, negative_slope, DropArgs(), h_out, norm_out, norm_stride,
                     as_stream(stream), 1, EgoCopy{self_out, self_stride}, &df);
}

int kgat_bi_interaction_train_f32(int64_t n_rows, int d_in, int d_out, const float* H, const float* HN,
                                  const float* W2, float negative_slope, float drop_p, uint64_t seed, int64_t row0,
                                  float* h_out, float* norm_out, int64_t norm_stride, float* self_out,
                                  int64_t self_stride, kgat_stream_t stream) {
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
  KGAT_CHECK_ARG(self_out == nullptr || (self_stride >= d_in && self_stride % 4 == 0 &&
                                         (reinterpret_cast<uintptr_t>(self_out) & 15u) == 0),
                 "bi_interaction_train: self_out must be 16-byte aligned with a row stride that is a multiple of 4 floats >= d_in");
  return bi_dispatch(n_rows, d_in, d_out, H, HN, W2, negative_slope, drop_args(drop_p, seed, row0, d_out), h_out,
                     norm_out, norm_stride, as_stream(stream), 2, EgoCopy{self_out, self_stride});
}

// out = a + b + c over n_rows x d (a: rows of a_stride floats - a column slice of a wider matrix; b, c, out contiguous)
__global__ __launch_bounds__(256) void add3_rows_kernel(int64_t n4, int d4, int64_t a_stride4, const float4* __restrict__ a,
                                                        const float4* __restrict__ b, const float4* __restrict__ c,
                                                        float4* __restrict__ out) {
  for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < n4; x += (int64_t)gridDim.x * 256) {
    const int64_t row = x / d4;
    const float4 va = a[row * a_stride4 + (x - row * d4)], vb = b[x], vc = c[x];
    // (a + b) + c, element by element: the order of `grad_out[:, :d] + g_a` followed by `+= g_b`
    out[x] = make_float4((va.x + vb.x) + vc.x, (va.y + vb.y) + vc.y, (va.z + vb.z) + vc.z, (va.w + vb.w) + vc.w);
  }
}

int kgat_add3_rows_f32(int64_t n_rows, int d, const float* a, int64_t a_stride, const float* b, const float* c, float* out,
                       kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && d > 0 && d % 4 == 0 && a_stride >= d && a_stride % 4 == 0, "add3_rows: bad sizes");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(a && b && c && out, "add3_rows: null pointer");
  KGAT_CHECK_ARG(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                   reinterpret_cast<uintptr_t>(out)) & 15u) == 0, "add3_rows: pointers must be 16-byte aligned");
  const int64_t n4 = n_rows * (d / 4);
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(add3_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), n4, d / 4, a_stride / 4,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                     reinterpret_cast<const float4*>(c), reinterpret_cast<float4*>(out));
  KGAT_CHECK_LAUNCH("add3_rows");
  return KGAT_OK;
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

int kgat_bi_interaction_bwd_input_supported(int d_in, int d_out) {
  auto ok = [](int d) { return d == 16 || d == 32 || d == 64 || d == 128; };
  return ok(d_in) && ok(d_out);
}

int kgat_bi_interaction_bwd_input_f32(int64_t n_rows, int d_in, int d_out, const float* grad_z, const float* W2,
                                      const float* H, const float* HN, float* grad_hn_times_h, float* grad_h_direct,
                                      kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_rows < INT32_MAX, "bi_interaction_bwd_input: bad row count");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(grad_z && W2 && H && HN && grad_hn_times_h && grad_h_direct, "bi_interaction_bwd_input: null pointer");
  if (!kgat_bi_interaction_bwd_input_supported(d_in, d_out)) {
    set_error("bi_interaction_bwd_input: unsupported widths %d -> %d", d_in, d_out);
    return KGAT_E_UNSUPPORTED;
  }
  const int64_t tiles = (n_rows + 15) / 16;
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 512) blocks = 512;
#define KGAT_BWD_CASE(DK, DN)                                                                                         \
  if (d_out == DK && d_in == DN) {                                                                                    \
    hipLaunchKernelGGL((bi_bwd_input_kernel<DK, DN>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),        \
                       (int32_t)n_rows, grad_z, W2, H, HN, grad_hn_times_h, grad_h_direct);                           \
    KGAT_CHECK_LAUNCH("bi_bwd_input");                                                                                \
    return KGAT_OK;                                                                                                   \
  }
  KGAT_BWD_CASE(16, 16) KGAT_BWD_CASE(16, 32) KGAT_BWD_CASE(16, 64) KGAT_BWD_CASE(16, 128)
  KGAT_BWD_CASE(32, 16) KGAT_BWD_CASE(32, 32) KGAT_BWD_CASE(32, 64) KGAT_BWD_CASE(32, 128)
  KGAT_BWD_CASE(64, 16) KGAT_BWD_CASE(64, 32) KGAT_BWD_CASE(64, 64) KGAT_BWD_CASE(64, 128)
  KGAT_BWD_CASE(128, 16) KGAT_BWD_CASE(128, 32) KGAT_BWD_CASE(128, 64) KGAT_BWD_CASE(128, 128)
#undef KGAT_BWD_CASE
  set_error("bi_interaction_bwd_input: unsupported widths %d -> %d", d_in, d_out);
  return KGAT_E_UNSUPPORTED;
}

int64_t kgat_bi_interaction_bwd_weight_partials(int64_t n_rows) {
  int64_t nb = (n_rows + 63) / 64;
  if (nb > 768) nb = 768;   // three workgroups per CU; each partial is d_out x d_in floats
  return nb < 1 ? 1 : nb;
}

int kgat_bi_interaction_bwd_weight_f32(int64_t n_rows, int d_in, int d_out, const float* grad_z, const float* H,
                                       const float* HN, float* partials, int64_t n_partials, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_rows < INT32_MAX, "bi_interaction_bwd_weight: bad row count");
  KGAT_CHECK_ARG(n_partials == kgat_bi_interaction_bwd_weight_partials(n_rows),
                 "bi_interaction_bwd_weight: n_partials must be kgat_bi_interaction_bwd_weight_partials(n_rows)");
  KGAT_CHECK_ARG(partials != nullptr && (n_rows == 0 || (grad_z && H && HN)), "bi_interaction_bwd_weight: null pointer");
  if (!kgat_bi_interaction_bwd_input_supported(d_in, d_out)) {
    set_error("bi_interaction_bwd_weight: unsupported widths %d -> %d", d_in, d_out);
    return KGAT_E_UNSUPPORTED;
  }
#define KGAT_BWW_CASE(DO_, DI_)                                                                                        \
  if (d_out == DO_ && d_in == DI_) {                                                                                  \
    hipLaunchKernelGGL((bi_bwd_weight_kernel<DO_, DI_>), dim3((unsigned)n_partials), dim3(256), 0, as_stream(stream), \
                       (int32_t)n_rows, grad_z, H, HN, partials);                                                     \
    KGAT_CHECK_LAUNCH("bi_bwd_weight");                                                                               \
    return KGAT_OK;                                                                                                   \
  }
  KGAT_BWW_CASE(16, 16) KGAT_BWW_CASE(16, 32) KGAT_BWW_CASE(16, 64) KGAT_BWW_CASE(16, 128)
  KGAT_BWW_CASE(32, 16) KGAT_BWW_CASE(32, 32) KGAT_BWW_CASE(32, 64) KGAT_BWW_CASE(32, 128)
  KGAT_BWW_CASE(64, 16) KGAT_BWW_CASE(64, 32) KGAT_BWW_CASE(64, 64) KGAT_BWW_CASE(64, 128)
  KGAT_BWW_CASE(128, 16) KGAT_BWW_CASE(128, 32) KGAT_BWW_CASE(128, 64) KGAT_BWW_CASE(128, 128)
#undef KGAT_BWW_CASE
  set_error("bi_interaction_bwd_weight: unsupported widths %d -> %d", d_in, d_out);
  return KGAT_E_UNSUPPORTED;
}

// out[s][e] = sum over the partials of set s, for up to four sets in ONE launch (the weight gradients of a stack's
// layers: three torch reductions of 11 us each before).  Sixteen lanes share an output float4: lane j adds partials
// j, j + 16, ... (eight loads in flight), then the sixteen sums are added in lane order by a fixed shuffle tree -
// a fixed order of additions: bitwise reproducible.
struct SumSets {
  const float4* part[4];
  float4* out[4];
  int32_t n_part[4], n4[4];     // partials per set, float4 elements per partial
  int32_t first_group[5];       // 16-lane groups of set s: [first_group[s], first_group[s + 1])
  int n_sets;
};
__global__ __launch_bounds__(256) void sum_partials_kernel(SumSets a) {
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int j = threadIdx.x & 15;
  if (grp >= a.first_group[a.n_sets]) return;
  int s = 0;
  while (s + 1 < a.n_sets && grp >= a.first_group[s + 1]) ++s;
  const int32_t e = (int32_t)(grp - a.first_group[s]);
  const float4* __restrict__ p = a.part[s] + e;
  const int32_t np = a.n_part[s], n4 = a.n4[s];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int32_t q0 = j; q0 < np; q0 += 16 * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int32_t q = q0 + 16 * u;
      v[u] = q < np ? p[(size_t)q * n4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    acc.x += __shfl_down(acc.x, off, 16); acc.y += __shfl_down(acc.y, off, 16);
    acc.z += __shfl_down(acc.z, off, 16); acc.w += __shfl_down(acc.w, off, 16);
  }
  if (j == 0) a.out[s][e] = acc;
}

int kgat_sum_partials_f32(int n_sets, const float* const* partials_host, float* const* out_host,
                          const int64_t* n_partials_host, const int64_t* n_elems_host, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_sets >= 0 && n_sets <= 4, "sum_partials: at most four sets per call");
  if (n_sets == 0) return KGAT_OK;
  KGAT_CHECK_ARG(partials_host && out_host && n_partials_host && n_elems_host, "sum_partials: null pointer");
  SumSets a;
  a.n_sets = 0;
  int64_t groups = 0;
  for (int s = 0; s < n_sets; ++s) {
    KGAT_CHECK_ARG(n_partials_host[s] >= 1 && n_partials_host[s] < INT32_MAX && n_elems_host[s] >= 0 &&
                       n_elems_host[s] % 4 == 0 && n_elems_host[s] / 4 < INT32_MAX, "sum_partials: set %d: bad sizes", s);
    if (n_elems_host[s] == 0) continue;
    KGAT_CHECK_ARG(partials_host[s] && out_host[s] &&
                       ((reinterpret_cast<uintptr_t>(partials_host[s]) | reinterpret_cast<uintptr_t>(out_host[s])) & 15u) == 0,
                   "sum_partials: set %d: null or misaligned pointer", s);
    const int c = a.n_sets++;
    a.part[c] = reinterpret_cast<const float4*>(partials_host[s]);
    a.out[c] = reinterpret_cast<float4*>(out_host[s]);
    a.n_part[c] = (int32_t)n_partials_host[s];
    a.n4[c] = (int32_t)(n_elems_host[s] / 4);
    a.first_group[c] = (int32_t)groups;
    groups += n_elems_host[s] / 4;
  }
  if (a.n_sets == 0) return KGAT_OK;
  a.first_group[a.n_sets] = (int32_t)groups;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0, as_stream(stream), a);
  KGAT_CHECK_LAUNCH("sum_partials");
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
