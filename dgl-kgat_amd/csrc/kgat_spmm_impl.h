// Kernel templates and launchers of the u_mul_e -> sum aggregation (included by kgat_spmm.hip, which
// instantiates the plain operator, and by kgat_spmm_bi.hip, which instantiates the forms with the
// bi-interaction fused behind the aggregation).  Not part of the ABI.
#pragma once
#include "kgat_common.h"

namespace kgat {

__device__ __forceinline__ float4 fma4(float a, const float4& x, const float4& c) {
  return make_float4(fmaf(a, x.x, c.x), fmaf(a, x.y, c.y), fmaf(a, x.z, c.z), fmaf(a, x.w, c.w));
}
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 mul4(const float4& a, const float4& b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// (128-thread workgroups with half-size tiles, round 3: D = 64 0.0993 vs 0.0987 ms, D = 128 0.187 vs 0.179, D = 8 0.058 vs 0.066)
constexpr int kSpmmThreads = 256;

// Threads per workgroup of the kernels that are laid out in lane groups of LPR lanes.  D <= 8 (two lanes per
// row and fewer): 128 - a 256-thread workgroup holds 128 runs there, i.e. 256 run partials to combine per
// tile; halving the workgroup (not the run length, which was tried and lost) took the D = 8 launch on the
// last-fm graph from 0.066 to 0.058 ms (round 3, AB_FLAG=-DKGAT_SPMM_THREADS=128).  Round 6 re-scanned workgroup size
// x run length for the narrow rows (scripts/micro/spmm_narrow_scan.sh, profiles/r06_spmm_narrow_scan.txt): 128 threads
// also at D = 16 / 32 - 16 runs and 32 run partials per tile, as a D = 64 tile has - with the half-length runs:
// D = 32 49.4 -> 47.6 us, D = 16 47.8 -> 44.7 (full-length runs 50.9 / 46.1, quarter-length 54.2 / 51.8); D >= 64 flat.
constexpr int spmm_threads(int lpr) { return (lpr <= 8 && kSpmmThreads == 256) ? 128 : kSpmmThreads; }

template <int LPR>
struct SpmmGeom {
  static constexpr int THREADS = spmm_threads(LPR);
  static constexpr int NSUB = THREADS / LPR;  // subgroups per workgroup
  static constexpr int U = LPR >= 4 ? 4 : LPR;     // X-row loads in flight per subgroup
};

// Where a launch also copies the rows' own features (KGAT_SPMM_MUL_SELF reads X[v] anyway): the ego
// block of Model.gnn's readout, out[:, :d] = h0 (models.py:159,168), written from the register that
// holds X[v] instead of by a separate 2 x N x d x 4-byte copy pass.
struct SelfCopy {
  float4* out;      // nullptr: off
  int64_t stride4;  // row stride in float4 units
};

// The dense part of a KGAT layer fused behind the aggregation (kernels instantiated with DO > 0; reference
// models.py:63-66 + :165-166 as one pass): a row that the aggregation completes becomes P = h * h_N in LDS
// (or, beyond the row buffer's capacity, in the global scratch `out`), and the workgroup that completed it
// runs Z = P W2^T (fp32 MFMA, the very products and summation order of kgat_bi_interaction_f32: same bits),
// LeakyReLU, the un-normalised row for the next layer and the L2-normalised row into its slice of the readout.
struct BiArgs {
  const float* W2 = nullptr;     // (DO, DI) row-major
  float slope = 0.f;
  float* h_out = nullptr;        // n_rows x DO, or nullptr
  float* norm_out = nullptr;     // 16-byte aligned, row stride a multiple of 4 floats, or nullptr
  int64_t norm_stride = 0;
  const int32_t* indptr = nullptr;
};

typedef float floatx4_s __attribute__((ext_vector_type(4)));

// LDS geometry of the fused form: P rows padded by one float4 (conflict-free column reads of 16 rows); the
// capacity keeps a workgroup's LDS within a fifth of the CU's 160 KB at the fused run length below
// (profiles/r04_spmm_lds_ballast_ab.txt: 4 resident workgroups per CU instead of 5 cost the D = 64 launch
// 1 % as LDS ballast alone, but 25-30 % once the workgroups also spend a sixth of their time in the dense tail).
constexpr int kFusedCap16 = 17;  // (A/B builds: 49 = four workgroups per CU at the full run length)
template <int LPR>
struct FusedGeom {
  static constexpr int PS4 = LPR + 1;                                    // row stride in float4
  // rows of the tile held in LDS: slot 0 (the tile's first row, never used) + whole 16-row blocks, so that a
  // block of the dense tail comes either from LDS or from the spill scratch, never from both
  static constexpr int CAP = LPR >= 16 ? kFusedCap16 : 49;
  static_assert((CAP - 1) % 16 == 0, "the row buffer holds whole 16-row blocks");
};

// Sum of squares of one 16-column tile of a row (lane (i, q) holds 4 of its values), reduced over the row's four
// lanes.  The row norm is the sum of the tiles' partials in tile order - the same order in every kernel that
// normalises (kgat_bi_interaction_f32, the fused launch, its finish launch), so their results agree bit for bit.
__device__ __forceinline__ float tile_ssq(const floatx4_s& z) {
  float s = z[0] * z[0];
  s = fmaf(z[1], z[1], s);
  s = fmaf(z[2], z[2], s);
  s = fmaf(z[3], z[3], s);
  s += __shfl_xor(s, 16, kWave);
  s += __shfl_xor(s, 32, kWave);
  return s;
}

// One column tile (16 output columns, index c) of a 16-row block of the dense part: lane (i = lane & 15,
// q = lane >> 4) brings a[s] = P[row_i][16 (s >> 2) + 4q + (s & 3)] and wf[s] = W2[16c + i][the same column]
// (MFMA operands swapped: the accumulator holds Z[row_i][16c + 4q .. + 3]).  Returns LeakyReLU(Z).
template <int KS>
__device__ __forceinline__ floatx4_s bi_col_tile(const float (&a)[KS], const float (&wf)[KS], float slope) {
  floatx4_s acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], a[s], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = acc[j] >= 0.f ? acc[j] : acc[j] * slope;
  return acc;
}

// W2's fragments for column tile c straight from global memory: lane (i, q) needs W2[16c + i][16m + 4q .. + 3]
// for fragments s = 4m .. 4m + 3.
template <int DI>
__device__ __forceinline__ void load_w2_col_tile(const float* __restrict__ W2, int c, int i, int q, float (&wf)[DI / 4]) {
#pragma unroll
  for (int m = 0; m < DI / 16; ++m) {
    const float4 v = *reinterpret_cast<const float4*>(W2 + (size_t)(16 * c + i) * DI + 16 * m + 4 * q);
    wf[4 * m + 0] = v.x; wf[4 * m + 1] = v.y; wf[4 * m + 2] = v.z; wf[4 * m + 3] = v.w;
  }
}

template <int DO>
__device__ __forceinline__ void store_col_tile(const BiArgs& bi, size_t r, int c, int q, const floatx4_s& z, float inv) {
  if (bi.h_out) *reinterpret_cast<float4*>(bi.h_out + r * DO + 16 * c + 4 * q) = make_float4(z[0], z[1], z[2], z[3]);
  if (bi.norm_out)
    *reinterpret_cast<float4*>(bi.norm_out + r * bi.norm_stride + 16 * c + 4 * q) =
        make_float4(z[0] * inv, z[1] * inv, z[2] * inv, z[3] * inv);
}

// Final store of a complete row.
template <int LPR, bool MUL_SELF, bool COPY_SELF = false>
__device__ __forceinline__ void store_row(float4* __restrict__ out, const float4* __restrict__ X,
                                          int32_t row, int32_t row0, int sl, float4 v,
                                          const SelfCopy sc = SelfCopy{nullptr, 0}) {
  if (MUL_SELF) {
    const float4 x = X[(size_t)row * LPR + sl];
    v = mul4(v, x);
    if (COPY_SELF) sc.out[(size_t)(row - row0) * sc.stride4 + sl] = x;
  }
  // (round 4 tried write-through `sc1` stores here, which do not keep the output rows' lines in the XCD's L2: no change,
  // 0.0867 vs 0.0868 ms - profiles/r04_spmm_cache_policy_ab.txt)
  out[(size_t)(row - row0) * LPR + sl] = v;
}

template <int LPR, int C, bool MUL_SELF, bool HAS_EID>
__global__ __launch_bounds__(SpmmGeom<LPR>::THREADS) void spmm_merge_kernel(
    int64_t e0, int64_t e1, int32_t row0, const int32_t* __restrict__ col,
    const int32_t* __restrict__ row_of, const int32_t* __restrict__ eid,
    const float4* __restrict__ X, const float* __restrict__ w, float4* __restrict__ out,
    float4* __restrict__ bpart) {
  constexpr int NSUB = SpmmGeom<LPR>::NSUB;
  constexpr int U = SpmmGeom<LPR>::U;
  constexpr int TE = NSUB * C;
  __shared__ float4 s_part[NSUB][2][LPR];
  __shared__ int32_t s_row[NSUB][2];

  const int tid = threadIdx.x;
  const int sub = tid / LPR, sl = tid % LPR;
  const int64_t tile0 = e0 + (int64_t)blockIdx.x * TE;
  const int64_t tile1 = (tile0 + TE < e1) ? tile0 + TE : e1;
  const int64_t p0 = tile0 + (int64_t)sub * C;
  const int64_t p1 = (p0 + C < tile1) ? p0 + C : tile1;

  int32_t cur_row = -1;
  bool head_done = false;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int64_t base = p0; base < p1; base += LPR) {
    const int64_t my = base + sl;
    const bool valid = my < p1;
    const int32_t c = valid ? col[my] : 0;
    const int32_t r = valid ? row_of[my] : -1;
    float wv = 0.f;
    if (valid) wv = HAS_EID ? w[eid[my]] : w[my];
    const int n = (p1 - base < LPR) ? (int)(p1 - base) : LPR;
    for (int j = 0; j < n; j += U) {
      int32_t cj[U], rj[U];
      float wj[U];
      float4 x[U];
#pragma unroll
      for (int i = 0; i < U; ++i) {
        cj[i] = __shfl(c, j + i, LPR);
        rj[i] = __shfl(r, j + i, LPR);
        wj[i] = __shfl(wv, j + i, LPR);
      }
#pragma unroll
      for (int i = 0; i < U; ++i) x[i] = X[(size_t)cj[i] * LPR + sl];  // padding lanes: row 0, w = 0
#pragma unroll
      for (int i = 0; i < U; ++i) {
        if (j + i < n) {
          if (rj[i] != cur_row) {
            if (cur_row >= 0) {
              if (!head_done) {
                s_part[sub][0][sl] = acc;
                if (sl == 0) s_row[sub][0] = cur_row;
                head_done = true;
              } else {
                store_row<LPR, MUL_SELF>(out, X, cur_row, row0, sl, acc);
              }
            }
            cur_row = rj[i];
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          acc = fma4(wj[i], x[i], acc);
        }
      }
    }
  }
  // the run's last open row: head slot if the run never changed row, else tail slot
  if (!head_done) {
    s_part[sub][0][sl] = acc;
    if (sl == 0) {
      s_row[sub][0] = cur_row;  // -1 for an empty run
      s_row[sub][1] = -1;
    }
  } else {
    s_part[sub][1][sl] = acc;
    if (sl == 0) s_row[sub][1] = cur_row;
  }
  __syncthreads();

  // In-order combine of the run-boundary partials by subgroup 0.
  if (sub == 0) {
    const int32_t first_row = s_row[0][0];
    const int32_t last_row = row_of[tile1 - 1];
    float4* bp = bpart + (size_t)blockIdx.x * 2 * LPR;
    int32_t crow = -1;
    float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto emit = [&](int32_t rr, const float4& v) {
      if (rr < 0) return;
      if (rr == first_row) bp[sl] = v;
      else if (rr == last_row) bp[LPR + sl] = v;
      else store_row<LPR, MUL_SELF>(out, X, rr, row0, sl, v);
    };
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int32_t rr = s_row[s][t];
        if (rr < 0) continue;
        const float4 v = s_part[s][t][sl];
        if (rr == crow) {
          cacc = add4(cacc, v);
        } else {
          emit(crow, cacc);
          crow = rr;
          cacc = v;
        }
      }
    }
    emit(crow, cacc);
  }
}

// ---------------------------------------------------------------------------------------------
// Second form of the merge kernel (weights in CSR order).  Same decomposition, combine and
// summation order as spmm_merge_kernel; what changes is the instruction count per edge, which
// - not the gather - bounds the first form (with every gather hitting L1 it still ran at 70 %
// of its time): the tile's (col, row, w) triples are staged once into LDS as 16-byte records
// (coalesced dword loads, one ds_write_b128 per edge), so an edge costs one ds_read_b128
// broadcast instead of three ds_bpermute; a group of four edges takes a wave-level "no lane
// group changes row" fast path (one compare + ballot instead of a divergent branch per edge);
// the FMAs are packed (v_pk_fma_f32); the next group's records and X rows are requested before
// the current group is consumed.
typedef float float2v __attribute__((ext_vector_type(2)));


typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
struct alignas(16) EdgeRec {
  // One 16-byte record per edge, moved as ONE vector (ds_write_b128 / ds_read_b128).  v[0]: the source row's position
  // in X in 16-byte units (row * LPR) as a 64-bit value - after the read it sits in an aligned register pair and the
  // gather's address is ONE v_lshl_add_u64 ((cq << 4) + the lane's pointer; round 6: the record held the 32-bit row and
  // every gather paid a sign extension, a copy, a 64-bit shift and a 64-bit add, four vector instructions beside the
  // two packed FMAs of an edge).  v[1]: destination row (low word), weight (high word).
  u64x2v v;
  __device__ __forceinline__ uint64_t cq() const { return v[0]; }
  __device__ __forceinline__ int32_t r() const { return (int32_t)(uint32_t)v[1]; }
  __device__ __forceinline__ float w() const { return __uint_as_float((uint32_t)(v[1] >> 32)); }
  static __device__ __forceinline__ EdgeRec make(uint64_t cq, int32_t r, float w) {
    EdgeRec e;
    e.v[0] = cq;
    e.v[1] = ((uint64_t)__float_as_uint(w) << 32) | (uint32_t)r;
    return e;
  }
};

template <int LPR, int C, bool MUL_SELF, bool COPY_SELF = false, int DO = 0>
__global__ __launch_bounds__(SpmmGeom<LPR>::THREADS) void spmm_merge2_kernel(
    int64_t e0, int64_t e1, int32_t row0, const int32_t* __restrict__ col,
    const int32_t* __restrict__ row_of, const float4* __restrict__ X, const float* __restrict__ w,
    float4* __restrict__ out, float4* __restrict__ bpart, const SelfCopy sc, const BiArgs bi) {
  constexpr int NSUB = SpmmGeom<LPR>::NSUB;
  constexpr int TE = NSUB * C;
  constexpr bool FUSED = DO > 0;
  constexpr int DI = 4 * LPR;
  static_assert(!FUSED || MUL_SELF, "the fused dense part consumes h * h_N");
constexpr int kSpmmGroup = 4;
  constexpr int G = (C % kSpmmGroup == 0) ? kSpmmGroup : 4;  // edges per group
  static_assert(C % G == 0, "run length must be a multiple of the group size");
  // the edge records and the run partials share one block of LDS: once the combine is done the fused form
  // reuses all of it as the staging area of spilled P rows (below)
  constexpr int kRecBytes = TE * (int)sizeof(EdgeRec), kPartBytes = NSUB * 2 * LPR * (int)sizeof(float4);
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kRecBytes + kPartBytes];
  EdgeRec* const s_rec = reinterpret_cast<EdgeRec*>(s_raw);
  float4 (*const s_part)[2][LPR] = reinterpret_cast<float4 (*)[2][LPR]>(s_raw + kRecBytes);
  __shared__ __attribute__((aligned(16))) int32_t s_row[NSUB][2];
  constexpr int CAP = FusedGeom<LPR>::CAP, PS4 = FusedGeom<LPR>::PS4;
  constexpr int KS = DI / 4, KT = FUSED ? DO / 16 : 1, WAVES = SpmmGeom<LPR>::THREADS / kWave;
  constexpr int GROUPS = WAVES / KT >= 1 ? WAVES / KT : 1;  // independent 16-row block streams of the dense part
  static_assert(!FUSED || (KT <= WAVES && WAVES % KT == 0), "one wavefront per column tile");
  __shared__ float4 s_P[FUSED ? CAP * PS4 : 1];
  __shared__ float s_ss[FUSED ? 2 * GROUPS * KT * 16 : 1];

  // KGAT_SPMM_XCD_REMAP=1 (A/B builds): every XCD takes a contiguous eighth of the tiles instead of
  // every eighth tile.  Measured slower on both CKG shapes (round 3, scripts/micro/spmm_runlen_ab.py with
  // AB_FLAG=-DKGAT_SPMM_XCD_REMAP=1; D = 64: 0.110 vs 0.105 ms, D = 128: 0.200 vs 0.181 ms, D = 32: 0.073 vs
  // 0.066 ms): with the round-robin placement the eight L2s work on neighbouring destination ranges at
  // the same time and miss on the same source rows together - one fetch from the Infinity Cache serves
  // requests that are in flight in several XCDs -, a contiguous eighth per XCD spreads the misses in time.
constexpr int kSpmmXcdRemap = 0;
  const int tid = threadIdx.x;
  const int sub = tid / LPR, sl = tid % LPR;
  const unsigned tile = kSpmmXcdRemap ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t tile0 = e0 + (int64_t)tile * TE;
  const int64_t tile1 = (tile0 + TE < e1) ? tile0 + TE : e1;
  const int n_tile = (int)(tile1 - tile0);
  for (int k = tid; k < TE; k += SpmmGeom<LPR>::THREADS) {
    EdgeRec rec = EdgeRec::make(0, -1, 0.f);
    if (k < n_tile) {
      const int64_t p = tile0 + k;
      // (non-temporal: col / row_of / w are read once and should not push source rows out of the caches - the launches
      //  gain 3 % (D = 64) / 7 % (D = 32), profiles/r04_spmm_cache_policy_ab.txt, the step 6.6 us: 0.4244 -> 0.4178 ms,
      //  profiles/r04_step_ab_cache_policy.txt)
      rec = EdgeRec::make((uint64_t)(uint32_t)__builtin_nontemporal_load(col + p) * LPR,
                          __builtin_nontemporal_load(row_of + p), __builtin_nontemporal_load(w + p));
    }
    s_rec[k] = rec;
  }
  if (FUSED) {  // rows of the tile's range without in-edges are never written: they must read as zero
    for (int k = tid; k < CAP * PS4; k += SpmmGeom<LPR>::THREADS) s_P[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  // (wave-uniform values read from LDS: pinned in SGPRs, the edge loop is at its VGPR limit)
  const int32_t first_row = __builtin_amdgcn_readfirstlane(s_rec[0].r());
  const int32_t last_row = __builtin_amdgcn_readfirstlane(s_rec[n_tile - 1].r());
  if constexpr (FUSED) {
    // interior rows beyond the row buffer's capacity that have no in-edges: nobody writes their P row, the dense
    // tail reads it from the scratch - zero it here (one lane per row tests the row offsets; such rows are rare)
    for (int32_t r = first_row + CAP + tid; r < last_row; r += SpmmGeom<LPR>::THREADS) {
      if (bi.indptr[r] == bi.indptr[r + 1]) {
        for (int c = 0; c < LPR; ++c) out[(size_t)(r - row0) * LPR + c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  // a complete row of the fused form: P = h * h_N into the LDS row buffer (slot = row - first row of the
  // tile), or into the global scratch when the tile spans more rows than the buffer holds
  // (h_N only: the product with the row's own features - and the ego copy - is formed by the dense tail, which loads
  // X[v] for whole 16-row blocks; loading it HERE is a dependent round trip inside the edge loop, the very epilogue
  // that cost the plain operator 14 us per launch - profiles/r04_spmm_epilogue_probe.txt)
  auto put_row = [&](int32_t row, const float4& acc) {
    const int32_t slot = row - first_row;
    if (slot < CAP) s_P[slot * PS4 + sl] = acc;
    else out[(size_t)(row - row0) * LPR + sl] = acc;
  };

  const EdgeRec* run = s_rec + sub * C;
  const float4* const Xl = X + sl;   // the lane's column of every source row
  int n_run = n_tile - sub * C;
  n_run = n_run < 0 ? 0 : (n_run > C ? C : n_run);
  const int ng = n_run / G;

  int32_t cur_row = n_run > 0 ? run[0].r() : -1;
  bool head_done = false;
  float2v a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
  // KGAT_SPMM_MUL_SELF loads X[row] when a row ENDS: a dependent round trip inside the edge loop - the wavefront's
  // other lane groups and its gathers in flight wait for it (benchmark graph: 91.4 us with that epilogue against
  // 77.5 us for the plain operator, profiles/r04_spmm_epilogue_probe.txt).  A/B arm KGAT_SPMM_SELF_PREFETCH=1:
  // request it when the row OPENS instead.  The layer now forms h * h_N in the bi-interaction kernel
  // (kgat_bi_interaction_mul_f32) and calls the plain operator.
constexpr int kSpmmSelfPrefetch = 0;  // measured SLOWER (100.7 vs 91.4 us at D = 64): the conditional load makes every later wait of the loop conservative
  constexpr bool SELF_PF = MUL_SELF && !FUSED && kSpmmSelfPrefetch != 0;
  float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
  auto open_row = [&](int32_t row) {
    cur_row = row;
    if (SELF_PF) xs = X[(size_t)row * LPR + sl];
  };

  auto flush = [&]() {  // the open row ends here
    const float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
    if (!head_done) {
      s_part[sub][0][sl] = acc;
      if (sl == 0) s_row[sub][0] = cur_row;
      head_done = true;
    } else if (FUSED) {
      put_row(cur_row, acc);
    } else if (SELF_PF) {
      if (COPY_SELF) sc.out[(size_t)(cur_row - row0) * sc.stride4 + sl] = xs;
      out[(size_t)(cur_row - row0) * LPR + sl] = mul4(acc, xs);
    } else {
      store_row<LPR, MUL_SELF, COPY_SELF>(out, X, cur_row, row0, sl, acc, sc);
    }
    a01 = (float2v){0.f, 0.f};
    a23 = (float2v){0.f, 0.f};
  };
  auto accum = [&](float wv, const float4& x) {
    const float2v ww = {wv, wv};
    a01 = __builtin_elementwise_fma(ww, (float2v){x.x, x.y}, a01);
    a23 = __builtin_elementwise_fma(ww, (float2v){x.z, x.w}, a23);
  };
  auto load_group = [&](int g, EdgeRec (&rec)[G], float4 (&x)[G]) {
#pragma unroll
    for (int i = 0; i < G; ++i) rec[i] = run[g * G + i];
#pragma unroll
    for (int i = 0; i < G; ++i) x[i] = Xl[rec[i].cq()];
  };
  auto consume = [&](const EdgeRec (&rec)[G], const float4 (&x)[G]) {
    // rows are sorted: the group stays inside the open row iff its last edge does
    if (__ballot(rec[G - 1].r() != cur_row) == 0ull) {
#pragma unroll
      for (int i = 0; i < G; ++i) accum(rec[i].w(), x[i]);
    } else {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (rec[i].r() != cur_row) {
          flush();
          open_row(rec[i].r());
        }
        accum(rec[i].w(), x[i]);
      }
    }
  };

  EdgeRec ra[G], rb[G];
  float4 xa[G], xb[G];
  // The next group's records and rows are requested UNCONDITIONALLY (past the run's end: its last group again, rows the
  // L1 has just seen, never consumed).  Round 6: they were requested under `if (g + 1 < ng)`, and a load under a branch
  // makes the compiler's counted waits conservative - it cannot assume the four new loads are pending, so the current
  // group's first row was waited for with vmcnt(3) instead of vmcnt(7): the wavefront sat out most of the round trip
  // of the loads it had just issued, and the double buffering bought nothing.
  if (ng > 0) load_group(0, ra, xa);
  for (int g = 0; g < ng; g += 2) {
    load_group(g + 1 < ng ? g + 1 : ng - 1, rb, xb);
    consume(ra, xa);
    load_group(g + 2 < ng ? g + 2 : ng - 1, ra, xa);
    if (g + 1 < ng) consume(rb, xb);
  }
  for (int j = ng * G; j < n_run; ++j) {  // only the last run of the edge range is ragged
    const EdgeRec rec = run[j];
    const float4 x = Xl[rec.cq()];
    if (rec.r() != cur_row) {
      flush();
      open_row(rec.r());
    }
    accum(rec.w(), x);
  }
  // the run's last open row: head slot if the run never changed row, else tail slot
  {
    const float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
    if (!head_done) {
      s_part[sub][0][sl] = acc;
      if (sl == 0) {
        s_row[sub][0] = cur_row;  // -1 for an empty run
        s_row[sub][1] = -1;
      }
    } else {
      s_part[sub][1][sl] = acc;
      if (sl == 0) s_row[sub][1] = cur_row;
    }
  }
  __syncthreads();

  // Combine of the run-boundary partials, in parallel: the 2*NSUB partials are in run order and
  // the partials of one row are consecutive.  Lane group s looks at its own two entries; an
  // entry that starts a row segment (its row differs from the previous valid entry's) sums
  // the segment in entry order and emits it.  Typical segments have two entries (tail of a run
  // + head of the next).  A segment of more than kShortSeg entries - a hub row covering many runs
  // of the tile; at narrow widths a tile has up to 512 entries - is summed by the whole wavefront
  // instead: its lane groups stride over the segment's entries and a fixed shuffle tree adds
  // their sums.  (One lane group walking a long segment alone was the largest phase of a D = 8
  // tile: median 16.7 k of 34.9 k cycles, 90 k on hub tiles.)
  // fused form: this wavefront's W2 fragments (column tile c_tile) are requested now and land while the combine runs
  const int wave_e = __builtin_amdgcn_readfirstlane(tid / kWave), lane_e = tid % kWave;  // (SGPR: uniform branches below)
  const int c_tile = wave_e % KT, grp = wave_e / KT;
  float wf[FUSED ? KS : 1];
  if constexpr (FUSED) load_w2_col_tile<DI>(bi.W2, c_tile, lane_e & 15, lane_e >> 4, wf);
  {
    float4* bp = bpart + (size_t)tile * 2 * LPR;
    constexpr int NE = 2 * NSUB;
    constexpr int SPW = kWave / LPR >= 1 ? kWave / LPR : 1;  // lane groups per wavefront
    constexpr int kShortSeg = 8;
    const int lane = tid % kWave;
    const int q = (LPR < kWave) ? lane / LPR : 0;
    auto emit = [&](int32_t rr, const float4& v) {
      if (rr == first_row) bp[sl] = v;
      else if (rr == last_row) bp[LPR + sl] = v;
      else if (FUSED) put_row(rr, v);
      else store_row<LPR, MUL_SELF, COPY_SELF>(out, X, rr, row0, sl, v, sc);
    };
    // Round 4: the rows of the entries around this lane group's two (2 sub - 2 .. 2 sub + 5) and the partials of
    // entries 2 sub .. 2 sub + 4 are requested TOGETHER, up front - one LDS round trip - and the common segments
    // (up to four entries) are summed from registers.  (The first form read s_row / s_part entry by entry, each read
    // depending on the comparison before it: 5-8 LDS round trips per entry, 4.6-5.8 k of a tile's 28-45 k ticks;
    // profiles/r04_gather_vs_spmm_widths.txt.)  Same entries, same order of additions: same bits.
    constexpr int WIN = 8;
    int32_t rw[WIN];   // rw[j] = row of entry 2 sub - 2 + j; -1: unused slot; -2: no such entry
    {
      const int32_t* srow = &s_row[0][0];
#pragma unroll
      for (int j = 0; j < WIN; j += 2) {
        const int e = 2 * sub - 2 + j;
        const bool in = e >= 0 && e < NE;
        const int2 v2 = *reinterpret_cast<const int2*>(srow + (in ? e : 0));
        rw[j] = in ? v2.x : -2;
        rw[j + 1] = in ? v2.y : -2;
      }
    }
    float4 pw[5];      // pw[j] = partial of entry 2 sub + j
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int e = 2 * sub + j;
      pw[j] = s_part[(e < NE ? e : 0) >> 1][e & 1][sl];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k = 2 * sub + t;
      const int wi = t + 2;
      const int32_t rr = rw[wi];
      // previous valid entry: entry k-1, or k-2 when k-1 is an unused tail slot
      const int32_t prev = rw[wi - 1] == -1 ? rw[wi - 2] : rw[wi - 1];
      const bool starts = rr >= 0 && prev != rr;
      bool is_long = false;
      {
        float4 v = pw[t];
        int taken = 1;
        bool open = starts;   // the segment may still continue
#pragma unroll
        for (int j = 1; j <= 3; ++j) {
          const int32_t r2 = rw[wi + j];
          if (open && r2 == rr) {
            v = add4(v, pw[t + j]);
            ++taken;
          } else if (r2 != -1) {
            open = false;     // another row, or the end of the table
          }
        }
        if (open) {           // longer than the window: walk on, entry by entry
          for (int k2 = k + 4; k2 < NE; ++k2) {
            const int32_t r2 = s_row[k2 >> 1][k2 & 1];
            if (r2 < 0) continue;
            if (r2 != rr) break;
            if (taken == kShortSeg) { is_long = true; break; }
            v = add4(v, s_part[k2 >> 1][k2 & 1][sl]);
            ++taken;
          }
        }
        if (starts && !is_long) emit(rr, v);
      }
      if (LPR < kWave) {  // (one lane group per wavefront: the walk above is all there is)
        unsigned long long todo = __ballot(is_long && sl == 0);
        while (todo) {
          const int src = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int k0 = __shfl(k, src, kWave);
          const int32_t r0 = __shfl(rr, src, kWave);
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int k2 = k0 + q; k2 < NE; k2 += SPW) {
            const int32_t r2 = s_row[k2 >> 1][k2 & 1];
            if (r2 < 0) continue;
            if (r2 != r0) break;
            acc = add4(acc, s_part[k2 >> 1][k2 & 1][sl]);
          }
#pragma unroll
          for (int off = LPR; off < kWave; off <<= 1) {
            float4 o;
            o.x = __shfl_xor(acc.x, off, kWave);
            o.y = __shfl_xor(acc.y, off, kWave);
            o.z = __shfl_xor(acc.z, off, kWave);
            o.w = __shfl_xor(acc.w, off, kWave);
            acc = add4(acc, o);
          }
          if (q == 0) emit(r0, acc);
        }
      } else if (is_long) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k2 = k; k2 < NE; ++k2) {
          const int32_t r2 = s_row[k2 >> 1][k2 & 1];
          if (r2 < 0) continue;
          if (r2 != rr) break;
          v = add4(v, s_part[k2 >> 1][k2 & 1][sl]);
        }
        emit(rr, v);
      }
    }
  }
  if constexpr (FUSED) {
    // the dense part on the tile's interior rows (first_row, last_row): all complete, their P rows in s_P
    // (slots 1 ..) or, beyond its capacity, in `out`.  Wavefront w owns column tile w % KT of the 16-row blocks
    // of stream w / KT; the row norms meet through LDS (one barrier per block, partials double-buffered).
    __syncthreads();  // every P row is written
    const int32_t n_int = last_row - first_row - 1;
    const int32_t n_blocks = n_int > 0 ? (n_int + 15) >> 4 : 0;
    const int i = lane_e & 15, q = lane_e >> 4;
    // Rows with slot < CAP sit in the row buffer.  The spilled ones (rows without in-edges among them were
    // zero-filled at the start of the tile) come back from global memory (L2: this workgroup wrote them) in
    // chunks of CH rows, all threads loading side by side into the LDS that held the edge records and the run
    // partials: one round trip per chunk, not one per 16-row block.
    constexpr int CH = ((kRecBytes + kPartBytes) / (PS4 * 16)) / 16 * 16;
    static_assert(CH >= 16, "the staging area holds at least one block");
    float4* const s_chunk = reinterpret_cast<float4*>(s_raw);
    int par = 0;  // parity of the norm partials' double buffer (uniform over the workgroup)
    // 16-row blocks [0, nblk) of `nrows` rows whose P rows start at `src` (LDS, row stride PS4), the first of
    // them destination row `rbase`
    auto dense_rows = [&](const float4* src, int32_t nrows, int32_t rbase) {
      const int32_t nblk = (nrows + 15) >> 4;
      for (int32_t itc = 0; itc * GROUPS < nblk; ++itc, par ^= 1) {
        const int32_t bb = itc * GROUPS + grp;
        const int32_t lr = 16 * bb + i;
        const bool valid = bb < nblk && lr < nrows;
        float a[KS];
        // the rows' own features (requested first: they come from L2 / the Infinity Cache while the LDS reads run)
        const float4* px = X + (size_t)(rbase + (valid ? lr : 0)) * LPR + q;
        float4 xv[DI / 16];
#pragma unroll
        for (int m = 0; m < DI / 16; ++m) xv[m] = px[4 * m];
#pragma unroll
        for (int m = 0; m < DI / 16; ++m) {
          const float4 v = src[(valid ? lr : 0) * PS4 + 4 * m + q];
          a[4 * m + 0] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
        }
        if (COPY_SELF && c_tile == 0 && valid) {   // ego block of the readout: one wavefront of the four writes it
          float4* pe = sc.out + (size_t)(rbase + lr - row0) * sc.stride4 + q;
#pragma unroll
          for (int m = 0; m < DI / 16; ++m) pe[4 * m] = xv[m];
        }
#pragma unroll
        for (int m = 0; m < DI / 16; ++m) {
          a[4 * m + 0] *= xv[m].x; a[4 * m + 1] *= xv[m].y; a[4 * m + 2] *= xv[m].z; a[4 * m + 3] *= xv[m].w;
        }
        const floatx4_s z = bi_col_tile<KS>(a, wf, bi.slope);
        float tot = tile_ssq(z);
        if constexpr (KT > 1) {
          float* ss = s_ss + (par * GROUPS + grp) * KT * 16;
          if (q == 0) ss[c_tile * 16 + i] = tot;
          // (LDS-only rendezvous: __syncthreads() would also wait for the global stores in flight)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
          tot = ss[i];
#pragma unroll
          for (int cc = 1; cc < KT; ++cc) tot += ss[cc * 16 + i];
        }
        const float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
        if (valid) store_col_tile<DO>(bi, (size_t)(rbase + lr - row0), c_tile, q, z, inv);
      }
    };
    const int32_t n_buf = n_int < CAP - 1 ? n_int : CAP - 1;  // interior rows held by the row buffer (slots 1 ..)
    if (n_buf > 0) dense_rows(s_P + PS4, n_buf, first_row + 1);
    for (int32_t slot0 = CAP; slot0 <= n_int; slot0 += CH) {
      const int32_t nrows = (n_int - slot0 + 1) < CH ? (n_int - slot0 + 1) : CH;
      if (slot0 > CAP) __syncthreads();  // the previous chunk has been consumed
      const float4* g = out + (size_t)(first_row + slot0 - row0) * LPR;
      for (int f = tid; f < nrows * LPR; f += SpmmGeom<LPR>::THREADS) s_chunk[(f / LPR) * PS4 + (f % LPR)] = g[f];
      __syncthreads();
      dense_rows(s_chunk, nrows, first_row + slot0);
    }
  } else {
  }
}

// Finish: (a) rows that are first/last in some tile: sum their tile partials in tile order;
// (b) rows without in-edges: write zeros.  For (a) every lane group (LPR lanes) examines one
// (tile, slot) item - is this tile the first one of the slot's row, i.e. its owner? - and sums a
// short chain of partials (the common case: the tail of one tile + the head of the next) by
// itself, so the items of a wavefront proceed in parallel; the chains of hub rows (8 tiles and
// more) are then summed one after the other by the whole wavefront, its lane groups striding over
// the chain's tiles.  Both orders are fixed by the graph alone.  (One wavefront per item, the
// first form, spent 0.3 ms of a 200 M-edge launch on waves that returned at once; one item per
// LANE, tried next, serialised up to 64 latency-bound chains in a wave: 9 -> 30 us on the
// amazon-book graph.)
template <int LPR, int C, bool MUL_SELF, bool COPY_SELF = false, int DO = 0>
__global__ __launch_bounds__(SpmmGeom<LPR>::THREADS) void spmm_finish_kernel(
    int64_t e0, int64_t e1, int32_t row0, int32_t n_rows, int32_t n_tiles,
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ row_of,
    const float4* __restrict__ X, float4* __restrict__ out, const float4* __restrict__ bpart,
    int32_t fix_blocks, const SelfCopy sc, const BiArgs bi) {
  constexpr int NSUB = SpmmGeom<LPR>::NSUB;
  constexpr int TE = NSUB * C;
  constexpr int SPW = kWave / LPR >= 1 ? kWave / LPR : 1;  // subgroups per wave
  constexpr int WPB = SpmmGeom<LPR>::THREADS / kWave;
  constexpr int kLongChain = 8;
  constexpr bool FUSED = DO > 0;
  constexpr int DI = 4 * LPR;
  // fused form: the boundary rows this workgroup finishes (one per lane group item) are collected as P rows
  // in LDS and take the dense part in 16-row blocks, one wavefront per block
  constexpr int ITEMS = WPB * SPW, PS4 = FusedGeom<LPR>::PS4;
  constexpr int ROWS_F = ITEMS < 16 ? 16 : ITEMS;
  __shared__ float4 s_Pf[FUSED ? ROWS_F * PS4 : 1];
  __shared__ int32_t s_rowf[FUSED ? ROWS_F : 1];
  const int tid = threadIdx.x;
  // a finished row: plain form -> out (store_row); fused form -> P row + row id into the item's LDS slot
  auto finish_row = [&](int32_t row, int sl_, const float4& acc, int slot) {
    if constexpr (FUSED) {
      const float4 x = X[(size_t)row * LPR + sl_];
      if (COPY_SELF) sc.out[(size_t)(row - row0) * sc.stride4 + sl_] = x;
      s_Pf[slot * PS4 + sl_] = mul4(acc, x);
      if (sl_ == 0) s_rowf[slot] = row;
    } else {
      store_row<LPR, MUL_SELF, COPY_SELF>(out, X, row, row0, sl_, acc, sc);
    }
  };
  if ((int32_t)blockIdx.x < fix_blocks) {
    if (LPR > kWave) return;  // not instantiated
    const int wave = tid / kWave, lane = tid % kWave;
    const int q = lane / LPR, sl = lane % LPR;
    if (FUSED) {
      for (int k = tid; k < ROWS_F; k += SpmmGeom<LPR>::THREADS) s_rowf[k] = -1;
      __syncthreads();
    }
    const int64_t item = ((int64_t)blockIdx.x * WPB + wave) * SPW + q;
    const int32_t b = (int32_t)(item >> 1);
    const int s = (int)(item & 1);
    int32_t my_row = -1, my_bl = 0;
    if (b < n_tiles) {
      const int64_t t0 = e0 + (int64_t)b * TE;
      const int64_t t1 = (t0 + TE < e1) ? t0 + TE : e1;
      const int32_t fr = row_of[t0], lr = row_of[t1 - 1];
      if (!(s == 1 && lr == fr)) {
        const int32_t r = s == 0 ? fr : lr;
        const int64_t rb = indptr[r], re = indptr[r + 1];
        if ((int32_t)((rb - e0) / TE) == b) {  // this tile owns the row's fix-up
          my_row = r;
          my_bl = (int32_t)((re - 1 - e0) / TE);
        }
      }
    }
    const bool is_long = my_row >= 0 && my_bl - b >= kLongChain;
    if (my_row >= 0 && !is_long) {
      float4 acc = bpart[((size_t)b * 2 + s) * LPR + sl];
      for (int32_t bb = b + 1; bb <= my_bl; ++bb) acc = add4(acc, bpart[((size_t)bb * 2) * LPR + sl]);
      finish_row(my_row, sl, acc, wave * SPW + q);
    }
    unsigned long long todo = __ballot(is_long && sl == 0);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int32_t r = __shfl(my_row, src, kWave);
      const int32_t bl = __shfl(my_bl, src, kWave);
      const int32_t bo = __shfl(b, src, kWave);
      const int so = __shfl(s, src, kWave);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int32_t bb = bo + q;
      constexpr int U = 8;  // partials requested together (the chain of a 10^5-edge row has hundreds)
      for (; bb + (U - 1) * SPW <= bl; bb += U * SPW) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t t = bb + u * SPW;
          v[u] = bpart[((size_t)t * 2 + ((t == bo) ? so : 0)) * LPR + sl];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = add4(acc, v[u]);
      }
      for (; bb <= bl; bb += SPW) {
        const int slot = (bb == bo) ? so : 0;
        acc = add4(acc, bpart[((size_t)bb * 2 + slot) * LPR + sl]);
      }
      // fixed-order reduction across the wave's subgroups
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1) {
        float4 o;
        o.x = __shfl_xor(acc.x, off, kWave);
        o.y = __shfl_xor(acc.y, off, kWave);
        o.z = __shfl_xor(acc.z, off, kWave);
        o.w = __shfl_xor(acc.w, off, kWave);
        acc = add4(acc, o);
      }
      if (q == 0) finish_row(r, sl, acc, wave * SPW + src / LPR);
    }
    if constexpr (FUSED) {
      constexpr int KS = DI / 4, KT = DO / 16;
      __syncthreads();
      if (wave * 16 < ITEMS) {
        const int i = lane & 15, qq = lane >> 4;
        const int32_t row = s_rowf[wave * 16 + i];
        if (__ballot(row >= 0) != 0ull) {
          float a[KS];
#pragma unroll
          for (int s_ = 0; s_ < KS; ++s_) a[s_] = 0.f;
          if (row >= 0) {
#pragma unroll
            for (int m = 0; m < DI / 16; ++m) {
              const float4 v = s_Pf[(wave * 16 + i) * PS4 + 4 * m + qq];
              a[4 * m + 0] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
            }
          }
          floatx4_s z[KT];
          float tot = 0.f;
#pragma unroll
          for (int c = 0; c < KT; ++c) {
            float wf[KS];
            load_w2_col_tile<DI>(bi.W2, c, i, qq, wf);
            z[c] = bi_col_tile<KS>(a, wf, bi.slope);
            const float part = tile_ssq(z[c]);
            tot = c == 0 ? part : tot + part;
          }
          const float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
          if (row >= 0) {
#pragma unroll
            for (int c = 0; c < KT; ++c) store_col_tile<DO>(bi, (size_t)(row - row0), c, qq, z[c], inv);
          }
        }
      }
    }
  } else {
    // rows without in-edges: one LANE tests one row (coalesced indptr loads, 64 rows per step); the
    // rows found - few - are zeroed by the wavefront's lane groups in turn.  (A lane group per row
    // walked 10 M rows in 305 dependent steps per group: 0.2 of the finish's 0.29 ms on that graph.)
    const int lane = tid % kWave;
    const int q = (LPR < kWave) ? lane / LPR : 0, sl = tid % LPR;
    const int64_t n_waves = (int64_t)(gridDim.x - fix_blocks) * WPB;
    const int64_t wave = (int64_t)(blockIdx.x - fix_blocks) * WPB + tid / kWave;
    for (int64_t v0 = wave * kWave; v0 < n_rows; v0 += n_waves * kWave) {
      const int64_t v = v0 + lane;
      bool empty = false;
      if (v < n_rows) {
        const int32_t row = row0 + (int32_t)v;
        empty = indptr[row] == indptr[row + 1];
      }
      unsigned long long m = __ballot(empty);
      int turn = 0;
      while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (turn == q) {
          const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (FUSED) {  // LeakyReLU(0 W2^T) = 0, normalised: 0 / max(0, eps) = 0
            for (int c = sl; c < DO / 4; c += LPR) {
              if (bi.h_out) reinterpret_cast<float4*>(bi.h_out)[(size_t)(v0 + b) * (DO / 4) + c] = zero;
              if (bi.norm_out) *reinterpret_cast<float4*>(bi.norm_out + (size_t)(v0 + b) * bi.norm_stride + 4 * c) = zero;
            }
          } else {
            out[(size_t)(v0 + b) * LPR + sl] = zero;
          }
          if (COPY_SELF) sc.out[(size_t)(v0 + b) * sc.stride4 + sl] = X[(size_t)(row0 + v0 + b) * LPR + sl];
        }
        turn = turn + 1 == SPW ? 0 : turn + 1;
      }
    }
  }
}

// Row-per-subgroup kernel (optionally in a degree-sorted order).  Kept as the simple
// reference formulation on the device and as an A/B arm for the merge kernel; long rows
// serialise on one subgroup.
template <int LPR, bool MUL_SELF, bool HAS_EID>
__global__ __launch_bounds__(SpmmGeom<LPR>::THREADS) void spmm_rows_kernel(
    int32_t n_rows, int32_t row0, const int32_t* __restrict__ indptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
    const int32_t* __restrict__ order, const float4* __restrict__ X, const float* __restrict__ w,
    float4* __restrict__ out) {
  constexpr int NSUB = SpmmGeom<LPR>::NSUB;
  constexpr int U = SpmmGeom<LPR>::U;
  const int tid = threadIdx.x;
  const int sub = tid / LPR, sl = tid % LPR;
  const int64_t g = (int64_t)blockIdx.x * NSUB + sub;
  if (g >= n_rows) return;
  const int32_t v = order ? order[g] : (int32_t)g;
  const int32_t row = row0 + v;
  const int32_t beg = indptr[row], end = indptr[row + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int32_t base = beg; base < end; base += LPR) {
    const int32_t my = base + sl;
    const bool valid = my < end;
    const int32_t c = valid ? col[my] : 0;
    float wv = 0.f;
    if (valid) wv = HAS_EID ? w[eid[my]] : w[my];
    const int n = (end - base < LPR) ? (end - base) : LPR;
    for (int j = 0; j < n; j += U) {
      int32_t cj[U];
      float wj[U];
      float4 x[U];
#pragma unroll
      for (int i = 0; i < U; ++i) {
        cj[i] = __shfl(c, j + i, LPR);
        wj[i] = __shfl(wv, j + i, LPR);
      }
#pragma unroll
      for (int i = 0; i < U; ++i) x[i] = X[(size_t)cj[i] * LPR + sl];
#pragma unroll
      for (int i = 0; i < U; ++i)
        if (j + i < n) acc = fma4(wj[i], x[i], acc);
    }
  }
  store_row<LPR, MUL_SELF>(out, X, row, row0, sl, acc);
}

// Any feature width: one wavefront per row, lane j covers columns j, j+64, ...
template <bool MUL_SELF, bool HAS_EID>
__global__ __launch_bounds__(kSpmmThreads) void spmm_rows_generic_kernel(
    int32_t n_rows, int32_t row0, int D, const int32_t* __restrict__ indptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
    const float* __restrict__ X, const float* __restrict__ w, float* __restrict__ out) {
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int64_t v = (int64_t)blockIdx.x * (kSpmmThreads / kWave) + wave;
  if (v >= n_rows) return;
  const int32_t row = row0 + (int32_t)v;
  const int32_t beg = indptr[row], end = indptr[row + 1];
  for (int d0 = 0; d0 < D; d0 += kWave) {
    const int d = d0 + lane;
    float acc = 0.f;
    if (d < D) {
      for (int32_t p = beg; p < end; ++p) {
        const float wv = HAS_EID ? w[eid[p]] : w[p];
        acc = fmaf(wv, X[(size_t)col[p] * D + d], acc);
      }
      if (MUL_SELF) acc *= X[(size_t)row * D + d];
      out[(size_t)v * D + d] = acc;
    }
  }
}

struct SpmmArgs {
  int64_t n_rows, row0;
  int D;
  const int32_t *indptr, *col, *row_of, *eid, *order;
  const float *X, *w;
  float* out;
  void* ws;
  size_t ws_bytes;
  unsigned flags;
  int algo;
  int32_t e0_host, e1_host;  // CSR position range of the row range
  hipStream_t st;
  float* self_out = nullptr;  // MUL_SELF launches of the merge algorithm: also copy X[v] here (SelfCopy)
  int64_t self_stride = 0;    // row stride of self_out in floats (a multiple of 4)
  BiArgs bi;                  // fused dense part (kernels with DO > 0)
};

// C: edges per lane-group run in the merge kernels; tiles are NSUB * C <= 2048 edges (the
// LDS record stage of the second form holds one tile).  A launch over few edges - a destination
// shard of a multi-GPU run holds E/P of them - takes the short run length: a tile is walked
// serially by its lane groups, so a launch cannot be shorter than one tile's time (~25 us at
// C = 64, which is what a 458 k-edge shard's launch took: a third of the whole graph's time for an
// eighth of its edges); a quarter of the run length gives four times the tiles, each a quarter as long.
constexpr int run_len(int lpr) { return lpr >= 8 ? 64 : (lpr == 4 ? 32 : (lpr == 2 ? 16 : 8)); }
constexpr int short_run_len(int lpr) { return run_len(lpr) / 4 >= 4 ? run_len(lpr) / 4 : 4; }
constexpr int64_t kShortRunTileLimit = 4096;  // use the short runs while they give at most this many tiles
// Between the two, for rows of 64 and 128 bytes (LPR = 4, 8: a tile is 2,048 edges there): half
// the run length while that gives at most kMidRunTileLimit tiles.  A launch of a few thousand
// full-length tiles ends with its last, longest tiles running on a nearly empty chip (tile times
// spread 14 k - 52 k cycles; 1,789 tiles on 1,280 workgroup slots at D = 32 on the amazon-book
// graph), and half-length tiles halve that tail: D = 32 0.079 -> 0.066 ms, D = 16 0.066 -> 0.060
// (quarter length: 0.076; at D = 64 / 128, 16-KB tiles of 1,024 edges, half length changes nothing,
// at D = 8 it costs 6-10 %: scripts/micro/spmm_runlen_ab.py).  The two macros exist for that A/B build only.
constexpr int kSpmmMidDiv = 2;
constexpr int mid_run_len(int lpr) {
  return (lpr == 8 || lpr == 4) ? run_len(lpr) / kSpmmMidDiv : run_len(lpr);
}
constexpr int kSpmmMidLimit = 16384;
constexpr int64_t kMidRunTileLimit = kSpmmMidLimit;

template <int LPR, int C>
static int64_t merge_tiles_c(int64_t n_edges) {
  constexpr int TE = SpmmGeom<LPR>::NSUB * C;
  return (n_edges + TE - 1) / TE;
}

template <int LPR>
static bool use_short_runs(int64_t n_edges) {
  return merge_tiles_c<LPR, short_run_len(LPR)>(n_edges) <= kShortRunTileLimit;
}
template <int LPR>
static bool use_mid_runs(int64_t n_edges) {
  return merge_tiles_c<LPR, mid_run_len(LPR)>(n_edges) <= kMidRunTileLimit;
}

template <int LPR, int C, bool MUL_SELF, bool HAS_EID, bool COPY_SELF = false, int DO = 0>
static int launch_merge_c(const SpmmArgs& a) {
  if (MUL_SELF && !HAS_EID && !COPY_SELF && a.self_out != nullptr) return launch_merge_c<LPR, C, MUL_SELF, HAS_EID, MUL_SELF && !HAS_EID, DO>(a);
  const SelfCopy sc{reinterpret_cast<float4*>(a.self_out), a.self_stride / 4};
  const int64_t e0 = a.e0_host, e1 = a.e1_host;
  const int64_t tiles = merge_tiles_c<LPR, C>(e1 - e0);
  const size_t need = (size_t)tiles * 2 * LPR * sizeof(float4);
  if (tiles > 0 && (a.ws == nullptr || a.ws_bytes < need)) {
    set_error("spmm: workspace too small (%zu < %zu)", a.ws_bytes, need);
    return KGAT_E_WORKSPACE;
  }
  float4* bpart = static_cast<float4*>(a.ws);
  if (tiles > 0) {
    if (!HAS_EID && a.algo != KGAT_SPMM_ALGO_MERGE1) {
      hipLaunchKernelGGL((spmm_merge2_kernel<LPR, C, MUL_SELF, COPY_SELF, DO>), dim3((unsigned)tiles),
                         dim3(SpmmGeom<LPR>::THREADS), 0, a.st, e0, e1, (int32_t)a.row0, a.col, a.row_of,
                         (const float4*)a.X, a.w, (float4*)a.out, bpart, sc, a.bi);
    } else if constexpr (DO > 0) {
      set_error("spmm: the fused form takes CSR-ordered weights and the merge algorithm");
      return KGAT_E_UNSUPPORTED;
    } else {
      hipLaunchKernelGGL((spmm_merge_kernel<LPR, C, MUL_SELF, HAS_EID>), dim3((unsigned)tiles),
                         dim3(SpmmGeom<LPR>::THREADS), 0, a.st, e0, e1, (int32_t)a.row0, a.col, a.row_of,
                         a.eid, (const float4*)a.X, a.w, (float4*)a.out, bpart);
    }
    KGAT_CHECK_LAUNCH("spmm_merge");
  }
  constexpr int kThreads = SpmmGeom<LPR>::THREADS;
  constexpr int kItemsPerBlock = (kThreads / kWave) * (kWave / LPR >= 1 ? kWave / LPR : 1);  // one per lane group
  const int32_t fix_blocks = (int32_t)((tiles * 2 + kItemsPerBlock - 1) / kItemsPerBlock);
  int64_t nz_blocks = (a.n_rows + kThreads - 1) / kThreads;  // one lane per row
  if (nz_blocks > 2048) nz_blocks = 2048;
  if (nz_blocks < 1) nz_blocks = 1;
  // KGAT_SPMM_DEFER_FINISH: the consumer (kgat_bi_interaction_mul_deferred_f32) forms the tiles' first / last rows
  // from the partials in the workspace, and the rows without in-edges, itself
  if (DO == 0 && (a.flags & KGAT_SPMM_DEFER_FINISH)) return KGAT_OK;
  hipLaunchKernelGGL((spmm_finish_kernel<LPR, C, MUL_SELF, COPY_SELF, DO>),
                     dim3((unsigned)(fix_blocks + nz_blocks)), dim3(kThreads), 0, a.st, e0, e1,
                     (int32_t)a.row0, (int32_t)a.n_rows, (int32_t)tiles, a.indptr, a.row_of,
                     (const float4*)a.X, (float4*)a.out, (const float4*)bpart, fix_blocks, sc, a.bi);
  KGAT_CHECK_LAUNCH("spmm_finish");
  return KGAT_OK;
}

// Run length of the fused form (DO > 0): the plain operator's, so that the aggregation's summation order - and
// with it every bit of the result - is that of the two-launch sequence.  (A/B builds, KGAT_FUSED_HALF_RUNS=1: half
// the run length at D = 64 - 512-edge tiles hold half the rows and leave LDS for a 56-row buffer at five
// workgroups per CU; measured slower, profiles/r04_fused_bi_ab.txt: the per-tile phases around the edge loop
// (stage, partials, combine, the dense tail) do not shrink with the tile.)
constexpr int kFusedHalfRuns = 0;
constexpr int fused_run_len(int lpr) { return (kFusedHalfRuns && lpr >= 16) ? run_len(lpr) / 2 : run_len(lpr); }

template <int LPR, bool MUL_SELF, bool HAS_EID, int DO = 0>
static int launch_merge(const SpmmArgs& a) {
  if (use_short_runs<LPR>((int64_t)a.e1_host - a.e0_host))
    return launch_merge_c<LPR, short_run_len(LPR), MUL_SELF, HAS_EID, false, DO>(a);
  if (use_mid_runs<LPR>((int64_t)a.e1_host - a.e0_host))
    return launch_merge_c<LPR, mid_run_len(LPR), MUL_SELF, HAS_EID, false, DO>(a);
  return launch_merge_c<LPR, (DO > 0 ? fused_run_len(LPR) : run_len(LPR)), MUL_SELF, HAS_EID, false, DO>(a);
}

// edges per tile launch_merge picks for a launch over n_edges positions (kgat_spmm_tile_edges)
template <int LPR>
static int merge_tile_edges(int64_t n_edges) {
  const int c = use_short_runs<LPR>(n_edges) ? short_run_len(LPR) : (use_mid_runs<LPR>(n_edges) ? mid_run_len(LPR) : run_len(LPR));
  return SpmmGeom<LPR>::NSUB * c;
}

template <int LPR, bool MUL_SELF, bool HAS_EID>
static int launch_rows(const SpmmArgs& a) {
  const int64_t blocks = (a.n_rows + SpmmGeom<LPR>::NSUB - 1) / SpmmGeom<LPR>::NSUB;
  hipLaunchKernelGGL((spmm_rows_kernel<LPR, MUL_SELF, HAS_EID>), dim3((unsigned)blocks),
                     dim3(SpmmGeom<LPR>::THREADS), 0, a.st, (int32_t)a.n_rows, (int32_t)a.row0, a.indptr,
                     a.col, a.eid, a.order, (const float4*)a.X, a.w, (float4*)a.out);
  KGAT_CHECK_LAUNCH("spmm_rows");
  return KGAT_OK;
}

template <bool MUL_SELF, bool HAS_EID>
static int launch_generic(const SpmmArgs& a) {
  const int64_t blocks = (a.n_rows + 3) / 4;
  hipLaunchKernelGGL((spmm_rows_generic_kernel<MUL_SELF, HAS_EID>), dim3((unsigned)blocks),
                     dim3(kSpmmThreads), 0, a.st, (int32_t)a.n_rows, (int32_t)a.row0, a.D,
                     a.indptr, a.col, a.eid, a.X, a.w, a.out);
  KGAT_CHECK_LAUNCH("spmm_generic");
  return KGAT_OK;
}

template <int LPR, bool MUL_SELF, bool HAS_EID>
static int dispatch_algo(const SpmmArgs& a) {
  if (a.algo == KGAT_SPMM_ALGO_ROWS) return launch_rows<LPR, MUL_SELF, HAS_EID>(a);
  return launch_merge<LPR, MUL_SELF, HAS_EID>(a);
}

template <bool MUL_SELF, bool HAS_EID>
static int dispatch_width(const SpmmArgs& a) {
  if (a.algo == KGAT_SPMM_ALGO_GENERIC) return launch_generic<MUL_SELF, HAS_EID>(a);
  switch (a.D) {
    case 4: return dispatch_algo<1, MUL_SELF, HAS_EID>(a);
    case 8: return dispatch_algo<2, MUL_SELF, HAS_EID>(a);
    case 16: return dispatch_algo<4, MUL_SELF, HAS_EID>(a);
    case 32: return dispatch_algo<8, MUL_SELF, HAS_EID>(a);
    case 64: return dispatch_algo<16, MUL_SELF, HAS_EID>(a);
    case 128: return dispatch_algo<32, MUL_SELF, HAS_EID>(a);
    case 256: return dispatch_algo<64, MUL_SELF, HAS_EID>(a);
    default: return launch_generic<MUL_SELF, HAS_EID>(a);
  }
}

static int lpr_for(int D) {
  switch (D) {
    case 4: case 8: case 16: case 32: case 64: case 128: case 256: return D / 4;
    default: return 0;
  }
}

}  // namespace kgat

