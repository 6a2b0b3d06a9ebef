// TransR-style attention logits for gfx950.  Rows A1 + A2 of SURVEY.md 8a.
//
// Replaces the per-relation loop of reference models.py:146-152
//     for i in range(R): eids = g.filter_edges(type == i); g.apply_edges(_att_score, eids)
// and the UDF models.py:135-144
//     t_r = ent[src] @ W_r ; h_r = ent[dst] @ W_r ; att = sum_j t_r[j] * tanh(h_r[j] + rel_r[j])
// by ONE launch over relation-grouped edges.
//
// Design: this is the only dense contraction on the path (4*d*k FLOP per edge against
// ~8*d bytes of gathered rows, AI ~ 31 FLOP/B at d = k = 64 > the fp32 ridge), so it is
// bound by the fp32 matrix pipe: v_mfma_f32_16x16x4_f32 (exact fp32, a k-ordered fma chain).
//  * A workgroup (4 wavefronts) owns a chunk of one relation's edges; W_r (d x k) is staged
//    once per chunk into LDS in MFMA B-fragment order, so every B fetch is one conflict-free
//    ds_read_b32 per lane, shared by 2*TILES MFMAs (t and h projections of TILES 16-edge tiles).
//  * A fragments come straight from global memory: lane (edge i = lane&15, slot q = lane>>4)
//    loads float4 pieces of its edge's embedding row so that the 4 lanes of an edge read 64
//    contiguous bytes per instruction; the contraction index is permuted consistently in A
//    and B (k-step s, slot q -> element 16*(s>>2) + 4*q + (s&3)), which MFMA does not care about.
//  * The 16 x k projection tiles stay in the accumulators; tanh, the product and the row sum
//    run on the VALU from there (row sum = 4 DPP-width shuffles over the 16 lanes of a slot).
//  * Relation-grouped src/dst arrays make the per-tile index reads coalesced; logits are
//    scattered back to edge-id order (and optionally to CSR position order for the softmax).
#include <math.h>

#include "kgat_common.h"

namespace kgat {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kAttThreads = 256;
constexpr int kAttChunk = 1024;  // edges of one relation per workgroup

// Locate (relation, chunk) for a block: blocks are laid out relation by relation,
// ceil(E_r / kAttChunk) blocks each.  Returns false past the end.
__device__ __forceinline__ bool att_locate(const int32_t* __restrict__ rel_ptr, int n_rel,
                                           int block, int& r_out, int32_t& beg, int32_t& end) {
  int acc = 0;
  for (int r = 0; r < n_rel; ++r) {
    const int32_t b = rel_ptr[r], e = rel_ptr[r + 1];
    const int nb = (e - b + kAttChunk - 1) / kAttChunk;
    if (block < acc + nb) {
      r_out = r;
      beg = b + (block - acc) * kAttChunk;
      end = (beg + kAttChunk < e) ? beg + kAttChunk : e;
      return true;
    }
    acc += nb;
  }
  return false;
}

// tanh for the epilogue.  ACCURATE = 0: 1 - 2/(exp(2x)+1) with the hardware exp2/rcp
// (absolute error ~1e-7, which is what the logit sum_j t_j*tanh(.) is sensitive to; the
// relative error near 0 is not preserved).  ACCURATE = 1: the device library's tanhf.
template <int ACCURATE>
__device__ __forceinline__ float att_tanh(float x) {
  if (ACCURATE) return tanhf(x);
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // exp(2x) = 2^(2x*log2 e)
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// tanh(x) given y = x * 2*log2(e) (the scale is folded into the caller's fma)
__device__ __forceinline__ float att_tanh_scaled(float y) {
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y) + 1.0f), 1.0f);
}
constexpr float kTwoLog2e = 2.8853900817779268f;

// Sum over the 16 lanes of a DPP row (= one slot q of the MFMA layout); every lane gets the
// total.  Four v_add_f32 with DPP operands, no LDS round trip.
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

template <int D_, int TILES>
struct AFrag {
  float t[TILES][D_ / 4], h[TILES][D_ / 4];
};

// Gather the A fragments (tail and head embedding rows) of TILES 16-edge tiles.
template <int D_, int TILES>
__device__ __forceinline__ void att_load_a(AFrag<D_, TILES>& f, const float* __restrict__ ent,
                                           const int32_t (&rs)[TILES], const int32_t (&rd)[TILES],
                                           int q) {
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const float4* ps = reinterpret_cast<const float4*>(ent + (size_t)rs[t] * D_) + q;
    const float4* pd = reinterpret_cast<const float4*>(ent + (size_t)rd[t] * D_) + q;
#pragma unroll
    for (int m = 0; m < D_ / 16; ++m) {
      const float4 a = ps[m * 4];
      const float4 b = pd[m * 4];
      f.t[t][4 * m + 0] = a.x; f.t[t][4 * m + 1] = a.y; f.t[t][4 * m + 2] = a.z; f.t[t][4 * m + 3] = a.w;
      f.h[t][4 * m + 0] = b.x; f.h[t][4 * m + 1] = b.y; f.h[t][4 * m + 2] = b.z; f.h[t][4 * m + 3] = b.w;
    }
  }
}

template <int D_, int K_, int TILES, int ACC_TANH>
__global__ __launch_bounds__(kAttThreads) void att_score_mfma_kernel(
    int n_rel, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ src_g, const int32_t* __restrict__ dst_g,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ logits, float* __restrict__ logits_csr,
    const int32_t* __restrict__ pos_g) {
  constexpr int KS = D_ / 4;   // k-steps (4 contraction elements each)
  constexpr int KT = K_ / 16;  // 16-wide column tiles of the projection
  constexpr int EPW = 16 * TILES;
  constexpr int STEP = (kAttThreads / kWave) * EPW;
  __shared__ float s_w[KS * KT * kWave];

  int r;
  int32_t cbeg, cend;
  if (!att_locate(rel_ptr, n_rel, blockIdx.x, r, cbeg, cend)) return;

  const int tid = threadIdx.x;
  // ---- stage W_r into LDS in B-fragment order: s_w[(s*KT + c)*64 + q*16 + n] =
  //      W_r[16*(s>>2) + 4*q + (s&3)][16*c + n]
  {
    const float* W = W_R + (size_t)r * D_ * K_;
    for (int idx = tid; idx < D_ * K_; idx += kAttThreads) {
      const int row = idx / K_, colx = idx % K_;
      const int s = (row >> 4) * 4 + (row & 3);
      const int q = (row >> 2) & 3;
      const int c = colx >> 4, n = colx & 15;
      s_w[(s * KT + c) * kWave + q * 16 + n] = W[idx];
    }
  }
  const int wave = tid / kWave, lane = tid % kWave;
  const int i = lane & 15, q = lane >> 4;
  float relv[KT];
#pragma unroll
  for (int c = 0; c < KT; ++c) relv[c] = rel[(size_t)r * K_ + 16 * c + i];

  // Edge indices are clamped to the chunk (no divergent loads); results of padding lanes are
  // simply not written.  The next step's rows are requested before this step's epilogue so
  // that the gather latency hides behind the tanh / reduction work.
  auto load_idx = [&](int32_t t0, int32_t (&rs)[TILES], int32_t (&rd)[TILES]) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      int32_t pe = t0 + t * 16 + i;
      pe = pe < cend ? pe : cend - 1;
      rs[t] = src_g[pe];
      rd[t] = dst_g[pe];
    }
  };
  int32_t t0 = cbeg + wave * EPW;
  int32_t rs[TILES], rd[TILES];
  AFrag<D_, TILES> fa;
  if (t0 < cend) {
    load_idx(t0, rs, rd);
    att_load_a<D_, TILES>(fa, ent, rs, rd, q);
  }
  __syncthreads();

  for (; t0 < cend; t0 += STEP) {
    const int32_t tn = t0 + STEP;
    const bool more = tn < cend;
    if (more) load_idx(tn, rs, rd);
    floatx4 accT[TILES][KT], accH[TILES][KT];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        accT[t][c] = (floatx4){0.f, 0.f, 0.f, 0.f};
        accH[t][c] = (floatx4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const float b = s_w[(s * KT + c) * kWave + lane];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          accT[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.t[t][s], b, accT[t][c], 0, 0, 0);
          accH[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.h[t][s], b, accH[t][c], 0, 0, 0);
        }
      }
    }
    if (more) att_load_a<D_, TILES>(fa, ent, rs, rd, q);
    // accX[t][c][j] = projection[edge 4*q + j of tile t][column 16*c + i]
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          part[j] = fmaf(accT[t][c][j], att_tanh<ACC_TANH>(accH[t][c][j] + relv[c]), part[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) part[j] = row16_sum(part[j]);
      // lanes i = 0..3 of slot q write edges 4*q + i
      const float v = i == 0 ? part[0] : (i == 1 ? part[1] : (i == 2 ? part[2] : part[3]));
      const int32_t pe = t0 + t * 16 + 4 * q + i;
      if (i < 4 && pe < cend) {
        const int32_t e = perm[pe];
        logits[e] = v;
        if (logits_csr) logits_csr[pos_g[pe]] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Persistent-wavefront form (d == k <= 64): W_r lives in registers as MFMA B fragments (64
// VGPRs at d = 64), every wavefront owns a contiguous, equally sized range of 16-edge tiles of
// the relation-grouped edge list (so the launch cannot end on a partly filled round of
// workgroups), A fragments are double buffered and requested one tile ahead, edge indices two
// tiles ahead.  No LDS traffic and no barrier inside the tile loop; W_r is re-read from L2 only
// when a wave's range crosses into the next relation.
constexpr int kAttMaxRelLds = 4096;

template <int D_, int ACC_TANH>
__global__ __launch_bounds__(kAttThreads) void att_score_persistent_kernel(
    int n_rel, int64_t n_edges, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ src_g, const int32_t* __restrict__ dst_g,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ logits, float* __restrict__ logits_csr,
    const int32_t* __restrict__ pos_g) {
  constexpr int K_ = D_;
  constexpr int KS = D_ / 4, KT = K_ / 16;
  __shared__ int32_t s_tptr[kAttMaxRelLds + 1];  // tile prefix per relation
  const int tid = threadIdx.x;
  for (int r = tid; r < n_rel; r += kAttThreads)
    s_tptr[r + 1] = (rel_ptr[r + 1] - rel_ptr[r] + 15) >> 4;
  __syncthreads();
  if (tid == 0) {
    int32_t run = 0;
    s_tptr[0] = 0;
    for (int r = 0; r < n_rel; ++r) {
      run += s_tptr[r + 1];
      s_tptr[r + 1] = run;
    }
  }
  __syncthreads();
  const int32_t n_tiles = s_tptr[n_rel];
  const int lane = tid % kWave;
  const int i = lane & 15, q = lane >> 4;
  const int64_t n_waves = (int64_t)gridDim.x * (kAttThreads / kWave);
  const int64_t wv = (int64_t)blockIdx.x * (kAttThreads / kWave) + tid / kWave;
  const int32_t t_begin = (int32_t)((int64_t)n_tiles * wv / n_waves);
  const int32_t t_end = (int32_t)((int64_t)n_tiles * (wv + 1) / n_waves);

  // Edges whose type is outside [0, R) sit after rel_ptr[R] in perm: logit 0 (DGL's
  // zero-initialised column); each wave clears its slice of that tail.
  {
    const int64_t tail0 = rel_ptr[n_rel];
    const int64_t n_tail = n_edges - tail0;
    for (int64_t p = tail0 + n_tail * wv / n_waves + lane; p < tail0 + n_tail * (wv + 1) / n_waves; p += kWave) {
      logits[perm[p]] = 0.f;
      if (logits_csr) logits_csr[pos_g[p]] = 0.f;
    }
  }
  if (t_begin >= t_end) return;

  float wreg[KS][KT];
  float relv[KT];

  // Relation segments of this wave's tile range; all cursor values are wave-uniform scalars.
  int32_t t = t_begin;
  while (t < t_end) {
    int lo = 0, hi = n_rel;  // relation of tile t: largest r with s_tptr[r] <= t
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_tptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int r = __builtin_amdgcn_readfirstlane(lo);
    const int32_t rbeg = __builtin_amdgcn_readfirstlane(rel_ptr[r]);
    const int32_t rend = __builtin_amdgcn_readfirstlane(rel_ptr[r + 1]);
    const int32_t tfirst = __builtin_amdgcn_readfirstlane(s_tptr[r]);
    int32_t seg_end = __builtin_amdgcn_readfirstlane(s_tptr[r + 1]);
    seg_end = seg_end < t_end ? seg_end : t_end;
    const int32_t n_seg = seg_end - t;               // tiles of relation r owned by this wave
    const int32_t pe0 = rbeg + ((t - tfirst) << 4);  // first edge of the first tile

    {  // W_r as B fragments: wreg[s][c] = W_r[16*(s>>2) + 4*q + (s&3)][16*c + i]
      const float* W = W_R + (size_t)r * D_ * K_;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int krow = 16 * (s >> 2) + 4 * q + (s & 3);
#pragma unroll
        for (int c = 0; c < KT; ++c) wreg[s][c] = W[krow * K_ + 16 * c + i];
      }
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        relv[c] = rel[(size_t)r * K_ + 16 * c + i];
        if (!ACC_TANH) relv[c] *= kTwoLog2e;  // tanh argument scale folded into one fma
      }
    }

    // Tile n of the segment covers edges pe0 + 16 n ...; indices past the segment are clamped
    // (redundant but branch-free prefetches: every step issues the same number of loads, so
    // the counted waits the compiler places never drain the prefetch of the following tile).
    // (the output slots of lanes i = 0..3 of slot q - edge id and CSR position of edge 4q + i -
    // travel with the indices, so the store block issues no load of its own: any load there
    // would need an in-order vmcnt(0) that also drains the A prefetch)
    struct Idx { int32_t rs, rd, oe, op; };
    auto load_idx = [&](int32_t n, Idx& x) {
      n = n < n_seg ? n : n_seg - 1;
      const int32_t base = pe0 + (n << 4);
      int32_t pe = base + i;
      pe = pe < rend ? pe : rend - 1;
      x.rs = src_g[pe];
      x.rd = dst_g[pe];
      int32_t po = base + 4 * q + (i & 3);
      po = po < rend ? po : rend - 1;
      x.oe = perm[po];
      x.op = logits_csr ? pos_g[po] : 0;
    };
    // 32-bit byte offsets from the table base (the launcher guarantees N*d*4 < 4 GiB): one
    // scalar base + one VGPR offset per row instead of 64-bit address arithmetic per load
    auto load_a = [&](AFrag<D_, 1>& f, int32_t rs, int32_t rd) {
      const char* base = reinterpret_cast<const char*>(ent);
      const uint32_t os = (uint32_t)rs * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
      const uint32_t od = (uint32_t)rd * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
#pragma unroll
      for (int m = 0; m < D_ / 16; ++m) {
        const float4 a = *reinterpret_cast<const float4*>(base + os + m * 64);
        const float4 b = *reinterpret_cast<const float4*>(base + od + m * 64);
        f.t[0][4 * m + 0] = a.x; f.t[0][4 * m + 1] = a.y; f.t[0][4 * m + 2] = a.z; f.t[0][4 * m + 3] = a.w;
        f.h[0][4 * m + 0] = b.x; f.h[0][4 * m + 1] = b.y; f.h[0][4 * m + 2] = b.z; f.h[0][4 * m + 3] = b.w;
      }
    };
    auto tile = [&](int32_t n, AFrag<D_, 1>& fa, const Idx& x) {
      floatx4 accT[KT], accH[KT];
#pragma unroll
      for (int cc = 0; cc < KT; ++cc) {
        accT[cc] = (floatx4){0.f, 0.f, 0.f, 0.f};
        accH[cc] = (floatx4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int cc = 0; cc < KT; ++cc) {
          accT[cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.t[0][s], wreg[s][cc], accT[cc], 0, 0, 0);
          accH[cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.h[0][s], wreg[s][cc], accH[cc], 0, 0, 0);
        }
      }
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < KT; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          part[j] = fmaf(accT[cc][j],
                         ACC_TANH ? tanhf(accH[cc][j] + relv[cc])
                                  : att_tanh_scaled(fmaf(accH[cc][j], kTwoLog2e, relv[cc])),
                         part[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) part[j] = row16_sum(part[j]);
      const float v = i == 0 ? part[0] : (i == 1 ? part[1] : (i == 2 ? part[2] : part[3]));
      const int32_t pe = pe0 + (n << 4) + 4 * q + i;  // lanes i = 0..3 of slot q: edges 4q + i
      if (i < 4 && pe < rend) {
        logits[x.oe] = v;
        if (logits_csr) logits_csr[x.op] = v;
      }
    };

    // software pipeline: indices two tiles ahead, A fragments one tile ahead (double buffer).
    // sched_barrier pins "issue the prefetch, then compute": without it the scheduler hoists
    // the next address computation above the MFMA phase and its wait drains the prefetch.
    AFrag<D_, 1> fa, fb;
    Idx x0, x1, x2;
    load_idx(0, x0);
    load_idx(1, x1);
    load_a(fa, x0.rs, x0.rd);
    for (int32_t n = 0; n < n_seg; n += 2) {
      load_a(fb, x1.rs, x1.rd);
      load_idx(n + 2, x2);
      __builtin_amdgcn_sched_barrier(0);
      tile(n, fa, x0);
      __builtin_amdgcn_sched_barrier(0);
      if (n + 1 >= n_seg) break;
      load_a(fa, x2.rs, x2.rd);
      load_idx(n + 3, x0);
      __builtin_amdgcn_sched_barrier(0);
      tile(n + 1, fb, x1);
      __builtin_amdgcn_sched_barrier(0);
      x1 = x0;   // indices of tile n + 3
      x0 = x2;   // indices of tile n + 2
    }
    t = seg_end;
  }
}

// Any (d, k): VALU kernel, one wavefront per 64-edge step, W_r column-sliced through LDS.
// Correctness path for widths the MFMA kernel does not cover (e.g. the d = k = 8 config).
__global__ __launch_bounds__(kAttThreads) void att_score_generic_kernel(
    int d, int k, int n_rel, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ src_g, const int32_t* __restrict__ dst_g,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ logits, float* __restrict__ logits_csr,
    const int32_t* __restrict__ pos_g) {
  extern __shared__ float s_wg[];  // d*k floats of W_r
  int r;
  int32_t cbeg, cend;
  if (!att_locate(rel_ptr, n_rel, blockIdx.x, r, cbeg, cend)) return;
  const float* W = W_R + (size_t)r * d * k;
  for (int idx = threadIdx.x; idx < d * k; idx += kAttThreads) s_wg[idx] = W[idx];
  __syncthreads();
  const float* er = rel + (size_t)r * k;
  for (int32_t pe = cbeg + threadIdx.x; pe < cend; pe += kAttThreads) {
    const float* xt = ent + (size_t)src_g[pe] * d;
    const float* xh = ent + (size_t)dst_g[pe] * d;
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
      float tr = 0.f, hr = 0.f;
      for (int a = 0; a < d; ++a) {
        const float wv = s_wg[a * k + j];
        tr = fmaf(xt[a], wv, tr);
        hr = fmaf(xh[a], wv, hr);
      }
      acc = fmaf(tr, tanhf(hr + er[j]), acc);
    }
    const int32_t e = perm[pe];
    logits[e] = acc;
    if (logits_csr) logits_csr[pos_g[pe]] = acc;
  }
}

struct AttArgs {
  unsigned grid;
  hipStream_t st;
  int n_rel;
  const int32_t *rel_ptr, *perm, *src_g, *dst_g;
  const float *ent, *W_R, *rel;
  float *logits, *logits_csr;
  const int32_t* pos_g;
  int waves_per_simd = 0;
  unsigned long long table_bytes = 0;
  int64_t n_edges = 0;
  bool needs_memset = true;
};

template <int D_, int K_, int TILES, int ACC_TANH>
static int launch_att_mfma(const AttArgs& a) {
  hipLaunchKernelGGL((att_score_mfma_kernel<D_, K_, TILES, ACC_TANH>), dim3(a.grid), dim3(kAttThreads),
                     0, a.st, a.n_rel, a.rel_ptr, a.perm, a.src_g, a.dst_g, a.ent, a.W_R, a.rel,
                     a.logits, a.logits_csr, a.pos_g);
  KGAT_CHECK_LAUNCH("att_score_mfma");
  return KGAT_OK;
}

template <int D_, int ACC_TANH>
static int launch_att_persistent(const AttArgs& a) {
  // one resident workgroup per CU slot the kernel's register budget admits: the tile ranges
  // are split evenly over exactly the wavefronts that run concurrently
  static int blocks_per_cu = 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  if (blocks_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, att_score_persistent_kernel<D_, ACC_TANH>,
                                                     kAttThreads, 0) != hipSuccess || nb < 1)
      nb = 1;
    blocks_per_cu = nb > 8 ? 8 : nb;
  }
  const int per_cu = a.waves_per_simd > 0 ? a.waves_per_simd : blocks_per_cu;
  const unsigned grid = (unsigned)(cus * per_cu);  // 4 waves per block, one per SIMD
  hipLaunchKernelGGL((att_score_persistent_kernel<D_, ACC_TANH>), dim3(grid), dim3(kAttThreads), 0, a.st,
                     a.n_rel, a.n_edges, a.rel_ptr, a.perm, a.src_g, a.dst_g, a.ent, a.W_R, a.rel, a.logits,
                     a.logits_csr, a.pos_g);
  KGAT_CHECK_LAUNCH("att_score_persistent");
  return KGAT_OK;
}

// variant bits (A/B tuning): bit 0 = one 16-edge tile per wave step of the chunk kernel
// (default two, except d = 128), bit 1 = device-library tanhf instead of the exp2/rcp form,
// bit 2 = workgroup-chunk kernel (W_r in LDS) instead of the persistent-wavefront kernel,
// bit 3 = force one persistent wave per SIMD (default: what the occupancy query admits).
template <int D_>
static int dispatch_att_variant(AttArgs a, int variant) {
  const bool one_tile = (variant & 1) || D_ >= 128;
  const bool acc = variant & 2;
  const bool chunk = (variant & 4) || D_ >= 128 || a.n_rel > kAttMaxRelLds || a.table_bytes >= (1ull << 32);
  a.waves_per_simd = (variant & 8) ? 1 : 0;  // 0 = as many as are resident
  if constexpr (D_ <= 64) {
    if (!chunk) return acc ? launch_att_persistent<D_, 1>(a) : launch_att_persistent<D_, 0>(a);
  }
  if (one_tile) return acc ? launch_att_mfma<D_, D_, 1, 1>(a) : launch_att_mfma<D_, D_, 1, 0>(a);
  if constexpr (D_ < 128) return acc ? launch_att_mfma<D_, D_, 2, 1>(a) : launch_att_mfma<D_, D_, 2, 0>(a);
  return KGAT_E_UNSUPPORTED;
}

}  // namespace kgat

using namespace kgat;

extern "C" {

int kgat_att_score_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                       const int32_t* rel_ptr, const int32_t* perm, const int32_t* src_g,
                       const int32_t* dst_g, const float* ent, const float* W_R, const float* rel,
                       float* logits, float* logits_csr, const int32_t* pos_g, int algo,
                       kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && d > 0 && k > 0 && n_rel >= 0,
                 "att_score: bad size");
  KGAT_CHECK_ARG(n_edges < INT32_MAX, "att_score: size exceeds int32");
  if (n_edges == 0) return KGAT_OK;
  KGAT_CHECK_ARG(logits != nullptr, "att_score: logits is null");
  KGAT_CHECK_ARG(logits_csr == nullptr || pos_g != nullptr, "att_score: logits_csr needs pos_g");
  KGAT_CHECK_ARG((algo >= KGAT_ATT_ALGO_AUTO && algo <= KGAT_ATT_ALGO_GENERIC) ||
                     (algo >= KGAT_ATT_ALGO_VARIANT_BASE && algo < KGAT_ATT_ALGO_VARIANT_BASE + 16),
                 "att_score: bad algo");
  hipStream_t st = as_stream(stream);
  // edges whose type is outside [0, R) keep logit 0 (DGL zero-initialised column): the
  // persistent kernel clears that tail itself, the other kernels rely on a memset
  const bool persistent = (d == k) && (d == 16 || d == 32 || d == 64) && n_rel > 0 &&
                          n_rel <= kAttMaxRelLds && (unsigned long long)n_nodes * d * 4ull < (1ull << 32) &&
                          (algo == KGAT_ATT_ALGO_AUTO || algo == KGAT_ATT_ALGO_MFMA ||
                           (algo >= KGAT_ATT_ALGO_VARIANT_BASE && !((algo - KGAT_ATT_ALGO_VARIANT_BASE) & 4)));
  if (!persistent) {
    hipError_t e = hipMemsetAsync(logits, 0, sizeof(float) * (size_t)n_edges, st);
    if (e == hipSuccess && logits_csr) e = hipMemsetAsync(logits_csr, 0, sizeof(float) * (size_t)n_edges, st);
    if (e != hipSuccess) {
      set_error("att_score: memset failed: %s", hipGetErrorString(e));
      return KGAT_E_HIP;
    }
  }
  if (n_rel == 0) return KGAT_OK;
  KGAT_CHECK_ARG(rel_ptr && perm && src_g && dst_g && ent && W_R && rel, "att_score: null pointer");
  AttArgs a;
  a.grid = (unsigned)((n_edges + kAttChunk - 1) / kAttChunk + n_rel);
  a.st = st; a.n_rel = n_rel; a.rel_ptr = rel_ptr; a.perm = perm; a.src_g = src_g; a.dst_g = dst_g;
  a.ent = ent; a.W_R = W_R; a.rel = rel; a.logits = logits; a.logits_csr = logits_csr;
  a.pos_g = pos_g;
  a.table_bytes = (unsigned long long)n_nodes * (unsigned long long)d * 4ull;
  a.n_edges = n_edges;
  const unsigned grid = a.grid;
  const bool mfma_ok = (d == k) && (d == 16 || d == 32 || d == 64 || d == 128);
  int variant = 0;
  if (algo >= KGAT_ATT_ALGO_VARIANT_BASE) {
    variant = algo - KGAT_ATT_ALGO_VARIANT_BASE;
    algo = KGAT_ATT_ALGO_MFMA;
  }
  if (algo == KGAT_ATT_ALGO_MFMA && !mfma_ok) {
    set_error("att_score: the MFMA kernel covers d == k in {16,32,64,128}, got d=%d k=%d", d, k);
    return KGAT_E_UNSUPPORTED;
  }
  if (mfma_ok && algo != KGAT_ATT_ALGO_GENERIC) {
    switch (d) {
      case 16: return dispatch_att_variant<16>(a, variant);
      case 32: return dispatch_att_variant<32>(a, variant);
      case 64: return dispatch_att_variant<64>(a, variant);
      default: return dispatch_att_variant<128>(a, variant);
    }
  }
  const size_t lds = sizeof(float) * (size_t)d * k;
  if (lds > 64 * 1024) {
    set_error("att_score: d*k = %d exceeds the generic kernel's LDS tile", d * k);
    return KGAT_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(att_score_generic_kernel, dim3(grid), dim3(kAttThreads), lds, st, d, k, n_rel,
                     rel_ptr, perm, src_g, dst_g, ent, W_R, rel, logits, logits_csr, pos_g);
  KGAT_CHECK_LAUNCH("att_score_generic");
  return KGAT_OK;
}

}  // extern "C"
