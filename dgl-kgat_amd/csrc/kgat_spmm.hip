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

#ifndef KGAT_SPMM_THREADS
#define KGAT_SPMM_THREADS 256  // (A/B builds.  128-thread workgroups with half-size tiles, round 3: D = 64 0.0993 vs 0.0987 ms, D = 128 0.187 vs 0.179, D = 8 0.058 vs 0.066)
#endif
constexpr int kSpmmThreads = KGAT_SPMM_THREADS;

// Threads per workgroup of the kernels that are laid out in lane groups of LPR lanes.  D <= 8 (two lanes per
// row and fewer): 128 - a 256-thread workgroup holds 128 runs there, i.e. 256 run partials to combine per
// tile; halving the workgroup (not the run length, which was tried and lost) took the D = 8 launch on the
// last-fm graph from 0.066 to 0.058 ms (round 3, AB_FLAG=-DKGAT_SPMM_THREADS=128).  Wider rows: no gain.
constexpr int spmm_threads(int lpr) { return (lpr <= 2 && kSpmmThreads == 256) ? 128 : kSpmmThreads; }

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

#ifdef KGAT_SPMM_STAMPS
// Diagnostic build only (-DKGAT_SPMM_STAMPS): per-tile phase stamps of the merge kernel, read
// back by scripts/micro/spmm_stamps.py.  Never compiled into the shipped library.
__device__ unsigned long long* g_spmm_stamps = nullptr;
#define KGAT_STAMP(k)                                                                  \
  do {                                                                                 \
    if (g_spmm_stamps && threadIdx.x == 0)                                             \
      g_spmm_stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();      \
  } while (0)
#else
#define KGAT_STAMP(k) do { } while (0)
#endif

struct alignas(16) EdgeRec {
  int32_t c;  // source row
  int32_t r;  // destination row
  float w;
  int32_t pad;
};

template <int LPR, int C, bool MUL_SELF, bool COPY_SELF = false>
__global__ __launch_bounds__(SpmmGeom<LPR>::THREADS) void spmm_merge2_kernel(
    int64_t e0, int64_t e1, int32_t row0, const int32_t* __restrict__ col,
    const int32_t* __restrict__ row_of, const float4* __restrict__ X, const float* __restrict__ w,
    float4* __restrict__ out, float4* __restrict__ bpart, const SelfCopy sc) {
  constexpr int NSUB = SpmmGeom<LPR>::NSUB;
  constexpr int TE = NSUB * C;
#ifndef KGAT_SPMM_GROUP
#define KGAT_SPMM_GROUP 4
#endif
#ifndef KGAT_SPMM_PREFETCH
#define KGAT_SPMM_PREFETCH 1
#endif
  constexpr int G = (C % KGAT_SPMM_GROUP == 0) ? KGAT_SPMM_GROUP : 4;  // edges per group
  static_assert(C % G == 0, "run length must be a multiple of the group size");
  __shared__ EdgeRec s_rec[TE];
  __shared__ float4 s_part[NSUB][2][LPR];
  __shared__ int32_t s_row[NSUB][2];

  // KGAT_SPMM_XCD_REMAP=1 (A/B builds): every XCD takes a contiguous eighth of the tiles instead of
  // every eighth tile.  Measured slower on both CKG shapes (round 3, scripts/micro/spmm_runlen_ab.py with
  // AB_FLAG=-DKGAT_SPMM_XCD_REMAP=1; D = 64: 0.110 vs 0.105 ms, D = 128: 0.200 vs 0.181 ms, D = 32: 0.073 vs
  // 0.066 ms): with the round-robin placement the eight L2s work on neighbouring destination ranges at
  // the same time and miss on the same source rows together - one fetch from the Infinity Cache serves
  // requests that are in flight in several XCDs -, a contiguous eighth per XCD spreads the misses in time.
#ifndef KGAT_SPMM_XCD_REMAP
#define KGAT_SPMM_XCD_REMAP 0
#endif
  const int tid = threadIdx.x;
  const int sub = tid / LPR, sl = tid % LPR;
  const unsigned tile = KGAT_SPMM_XCD_REMAP ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t tile0 = e0 + (int64_t)tile * TE;
  const int64_t tile1 = (tile0 + TE < e1) ? tile0 + TE : e1;
  const int n_tile = (int)(tile1 - tile0);
  KGAT_STAMP(0);

  for (int k = tid; k < TE; k += SpmmGeom<LPR>::THREADS) {
    EdgeRec rec;
    if (k < n_tile) {
      const int64_t p = tile0 + k;
      rec.c = col[p];
      rec.r = row_of[p];
      rec.w = w[p];
    } else {
      rec.c = 0;
      rec.r = -1;
      rec.w = 0.f;
    }
    rec.pad = 0;
    s_rec[k] = rec;
  }
  __syncthreads();
  KGAT_STAMP(1);

  const EdgeRec* run = s_rec + sub * C;
  int n_run = n_tile - sub * C;
  n_run = n_run < 0 ? 0 : (n_run > C ? C : n_run);
  const int ng = n_run / G;

  int32_t cur_row = n_run > 0 ? run[0].r : -1;
  bool head_done = false;
  float2v a01 = {0.f, 0.f}, a23 = {0.f, 0.f};

  auto flush = [&]() {  // the open row ends here
    const float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
    if (!head_done) {
      s_part[sub][0][sl] = acc;
      if (sl == 0) s_row[sub][0] = cur_row;
      head_done = true;
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
    for (int i = 0; i < G; ++i) x[i] = X[(size_t)rec[i].c * LPR + sl];
  };
  auto consume = [&](const EdgeRec (&rec)[G], const float4 (&x)[G]) {
    // rows are sorted: the group stays inside the open row iff its last edge does
    if (__ballot(rec[G - 1].r != cur_row) == 0ull) {
#pragma unroll
      for (int i = 0; i < G; ++i) accum(rec[i].w, x[i]);
    } else {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (rec[i].r != cur_row) {
          flush();
          cur_row = rec[i].r;
        }
        accum(rec[i].w, x[i]);
      }
    }
  };

#if KGAT_SPMM_PREFETCH
  EdgeRec ra[G], rb[G];
  float4 xa[G], xb[G];
  if (ng > 0) load_group(0, ra, xa);
  for (int g = 0; g < ng; g += 2) {
    if (g + 1 < ng) load_group(g + 1, rb, xb);
    consume(ra, xa);
    if (g + 1 >= ng) break;
    if (g + 2 < ng) load_group(g + 2, ra, xa);
    consume(rb, xb);
  }
#else
  for (int g = 0; g < ng; ++g) {
    EdgeRec ra[G];
    float4 xa[G];
    load_group(g, ra, xa);
    consume(ra, xa);
  }
#endif
  for (int j = ng * G; j < n_run; ++j) {  // only the last run of the edge range is ragged
    const EdgeRec rec = run[j];
    const float4 x = X[(size_t)rec.c * LPR + sl];
    if (rec.r != cur_row) {
      flush();
      cur_row = rec.r;
    }
    accum(rec.w, x);
  }

  KGAT_STAMP(2);
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
  KGAT_STAMP(3);
  {
    const int32_t first_row = s_rec[0].r;
    const int32_t last_row = s_rec[n_tile - 1].r;
    float4* bp = bpart + (size_t)tile * 2 * LPR;
    constexpr int NE = 2 * NSUB;
    constexpr int SPW = kWave / LPR >= 1 ? kWave / LPR : 1;  // lane groups per wavefront
    constexpr int kShortSeg = 8;
    const int lane = tid % kWave;
    const int q = (LPR < kWave) ? lane / LPR : 0;
    auto emit = [&](int32_t rr, const float4& v) {
      if (rr == first_row) bp[sl] = v;
      else if (rr == last_row) bp[LPR + sl] = v;
      else store_row<LPR, MUL_SELF, COPY_SELF>(out, X, rr, row0, sl, v, sc);
    };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k = 2 * sub + t;
      const int32_t rr = s_row[sub][t];
      bool starts = rr >= 0;
      if (starts && k > 0) {
        // previous valid entry: entry k-1, or k-2 when k-1 is an unused tail slot
        int32_t prev = s_row[(k - 1) >> 1][(k - 1) & 1];
        if (prev < 0 && k > 1) prev = s_row[(k - 2) >> 1][(k - 2) & 1];
        starts = prev != rr;
      }
      bool is_long = false;
      if (starts) {
        float4 v = s_part[sub][t][sl];
        int taken = 1;
        for (int k2 = k + 1; k2 < NE; ++k2) {
          const int32_t r2 = s_row[k2 >> 1][k2 & 1];
          if (r2 < 0) continue;
          if (r2 != rr) break;
          if (taken == kShortSeg) { is_long = true; break; }
          v = add4(v, s_part[k2 >> 1][k2 & 1][sl]);
          ++taken;
        }
        if (!is_long) emit(rr, v);
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
  KGAT_STAMP(4);
  KGAT_STAMP(5);
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
template <int LPR, int C, bool MUL_SELF, bool COPY_SELF = false>
__global__ __launch_bounds__(SpmmGeom<LPR>::THREADS) void spmm_finish_kernel(
    int64_t e0, int64_t e1, int32_t row0, int32_t n_rows, int32_t n_tiles,
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ row_of,
    const float4* __restrict__ X, float4* __restrict__ out, const float4* __restrict__ bpart,
    int32_t fix_blocks, const SelfCopy sc) {
  constexpr int NSUB = SpmmGeom<LPR>::NSUB;
  constexpr int TE = NSUB * C;
  constexpr int SPW = kWave / LPR >= 1 ? kWave / LPR : 1;  // subgroups per wave
  constexpr int WPB = SpmmGeom<LPR>::THREADS / kWave;
  constexpr int kLongChain = 8;
  const int tid = threadIdx.x;
  if ((int32_t)blockIdx.x < fix_blocks) {
    if (LPR > kWave) return;  // not instantiated
    const int wave = tid / kWave, lane = tid % kWave;
    const int q = lane / LPR, sl = lane % LPR;
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
      store_row<LPR, MUL_SELF, COPY_SELF>(out, X, my_row, row0, sl, acc, sc);
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
      if (q == 0) store_row<LPR, MUL_SELF, COPY_SELF>(out, X, r, row0, sl, acc, sc);
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
          out[(size_t)(v0 + b) * LPR + sl] = make_float4(0.f, 0.f, 0.f, 0.f);
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
#ifndef KGAT_SPMM_MID_DIV
#define KGAT_SPMM_MID_DIV 2
#endif
constexpr int mid_run_len(int lpr) {
  return (lpr == 8 || lpr == 4) ? run_len(lpr) / KGAT_SPMM_MID_DIV : run_len(lpr);
}
#ifndef KGAT_SPMM_MID_LIMIT
#define KGAT_SPMM_MID_LIMIT 16384
#endif
constexpr int64_t kMidRunTileLimit = KGAT_SPMM_MID_LIMIT;

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

template <int LPR, int C, bool MUL_SELF, bool HAS_EID, bool COPY_SELF = false>
static int launch_merge_c(const SpmmArgs& a) {
  if (MUL_SELF && !HAS_EID && !COPY_SELF && a.self_out != nullptr) return launch_merge_c<LPR, C, MUL_SELF, HAS_EID, MUL_SELF && !HAS_EID>(a);
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
      hipLaunchKernelGGL((spmm_merge2_kernel<LPR, C, MUL_SELF, COPY_SELF>), dim3((unsigned)tiles),
                         dim3(SpmmGeom<LPR>::THREADS), 0, a.st, e0, e1, (int32_t)a.row0, a.col, a.row_of,
                         (const float4*)a.X, a.w, (float4*)a.out, bpart, sc);
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
  hipLaunchKernelGGL((spmm_finish_kernel<LPR, C, MUL_SELF, COPY_SELF>),
                     dim3((unsigned)(fix_blocks + nz_blocks)), dim3(kThreads), 0, a.st, e0, e1,
                     (int32_t)a.row0, (int32_t)a.n_rows, (int32_t)tiles, a.indptr, a.row_of,
                     (const float4*)a.X, (float4*)a.out, (const float4*)bpart, fix_blocks, sc);
  KGAT_CHECK_LAUNCH("spmm_finish");
  return KGAT_OK;
}

template <int LPR, bool MUL_SELF, bool HAS_EID>
static int launch_merge(const SpmmArgs& a) {
  if (use_short_runs<LPR>((int64_t)a.e1_host - a.e0_host))
    return launch_merge_c<LPR, short_run_len(LPR), MUL_SELF, HAS_EID>(a);
  if (use_mid_runs<LPR>((int64_t)a.e1_host - a.e0_host))
    return launch_merge_c<LPR, mid_run_len(LPR), MUL_SELF, HAS_EID>(a);
  return launch_merge_c<LPR, run_len(LPR), MUL_SELF, HAS_EID>(a);
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

}  // namespace kgat

using namespace kgat;

extern "C" {

#ifdef KGAT_SPMM_STAMPS
int kgat_debug_set_spmm_stamps(void* dev_ptr) {
  return hipMemcpyToSymbol(HIP_SYMBOL(kgat::g_spmm_stamps), &dev_ptr, sizeof(void*)) == hipSuccess ? 0 : -4;
}
#endif

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
  return align_up((size_t)tiles * 2 * lpr * sizeof(float4), 256) + 256;
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
  KGAT_CHECK_ARG((flags & ~(unsigned)KGAT_SPMM_MUL_SELF) == 0, "spmm: unknown flags 0x%x", flags);
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
