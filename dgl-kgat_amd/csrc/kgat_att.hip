#include <stdlib.h>
// TransR-style attention logits for gfx950.  Rows A1 + A2 of SURVEY.md 8a.
//
// Replaces the per-relation loop of reference models.py:146-152
//     for i in range(R): eids = g.filter_edges(type == i); g.apply_edges(_att_score, eids)
// and the UDF models.py:135-144
//     t_r = ent[src] @ W_r ; h_r = ent[dst] @ W_r ; att = sum_j t_r[j] * tanh(h_r[j] + rel_r[j])
// by launches over relation-grouped edges.  Three kernel families, chosen by the host:
//   * kgat_att_score_split_f32 (kgat_att_persistent.hip, d = k <= 64, default when the (head,
//     relation) groups share work): head projections + tanh once per group, tail projection
//     + dot product per edge;
//   * kgat_att_score_f32 -> persistent one-kernel form (kgat_att_persistent.hip, d = k <= 64);
//   * kgat_att_score_f32 -> the workgroup-chunk kernel below (d = k = 128, or forced for A/B)
//     and the VALU kernel for any other (d, k).
//
// Design: this is the only dense contraction on the path (4*d*k FLOP per edge against
// ~8*d bytes of gathered rows, AI ~ 31 FLOP/B at d = k = 64 > the fp32 ridge), so it is
// bound by the fp32 matrix pipe: v_mfma_f32_16x16x4_f32 (exact fp32, a k-ordered fma chain).
// Common to the MFMA kernels:
//  * A fragments come straight from global memory: lane (edge i = lane&15, slot q = lane>>4)
//    loads float4 pieces of its edge's embedding row so that the 4 lanes of an edge read 64
//    contiguous bytes per instruction; the contraction index is permuted consistently in A
//    and B (k-step s, slot q -> element 16*(s>>2) + 4*q + (s&3)), which MFMA does not care about.
//  * The 16 x k projection tiles stay in registers; tanh, the product and the row sum run on
//    the VALU from there (row sum = 4 DPP adds over the 16 lanes of a slot).
//  * Relation-grouped endpoint arrays make the per-tile index reads coalesced; logits are
//    scattered to CSR position order for the softmax (and optionally to edge-id order).
// The chunk kernel of this file: a workgroup (4 wavefronts) owns a chunk of one relation's
// edges; W_r (d x k) is staged once per chunk into LDS in MFMA B-fragment order, so every B
// fetch is one conflict-free ds_read_b32 per lane, shared by 2*TILES MFMAs.
#include "kgat_att_common.h"

namespace kgat {

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

template <int D_, int K_, int TILES, int ACC_TANH>
static int launch_att_mfma(const AttArgs& a) {
  hipLaunchKernelGGL((att_score_mfma_kernel<D_, K_, TILES, ACC_TANH>), dim3(a.grid), dim3(kAttThreads),
                     0, a.st, a.n_rel, a.rel_ptr, a.perm, a.src_g, a.dst_g, a.ent, a.W_R, a.rel,
                     a.logits, a.logits_csr, a.pos_g);
  KGAT_CHECK_LAUNCH("att_score_mfma");
  return KGAT_OK;
}

// The MFMA forms of the one-kernel attention: persistent wavefronts with W_r in registers (d <= 64), or - at d = 128,
// beyond kAttMaxRelLds relations, for tables of 4 GiB and more, and when asked for (KGAT_ATT_ALGO_MFMA_CHUNK) - the
// workgroup-chunk kernel with W_r in LDS (two 16-edge tiles per wave step; one at d = 128).
template <int D_>
static int dispatch_att_mfma(const AttArgs& a, bool force_chunk) {
  const bool chunk = force_chunk || D_ >= 128 || a.n_rel > kAttMaxRelLds || a.table_bytes >= (1ull << 32);
  if (D_ <= 64 && !chunk) return launch_att_persistent_any(D_, a);
  if constexpr (D_ >= 128) return launch_att_mfma<D_, D_, 1, 0>(a);
  else return launch_att_mfma<D_, D_, 2, 0>(a);
}

// Per-edge half of the folded form (head half: att_fold_head_kernel): logit = e_t . V[group].
// A gather-dot, bound by the tail-row gather like the SpMM.  LPE = d/4 lanes hold one edge's
// rows as float4; a wavefront takes 64 consecutive grouped positions, reads their indices with
// one coalesced load each and hands them to the lane groups through ds_bpermute; lane group u
// owns positions p0 + u*LPE .. + LPE-1 and walks them in LPE steps.  The reduced dot products
// are collected so that lane l ends up with position p0 + l, and the results leave with one
// scattered store per output array.
template <int D_, bool LOGITS_EID>
__global__ __launch_bounds__(256) void att_fold_tail_kernel(
    int n_rel, const int32_t* __restrict__ rel_ptr, int64_t n_edges, const int32_t* __restrict__ src_g,
    const int32_t* __restrict__ gid,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ pos_g, const float* __restrict__ ent,
    const float* __restrict__ V_tab, float* __restrict__ logits, float* __restrict__ logits_csr) {
  constexpr int LPE = kTailLanesPerEdge<D_>();  // lanes per edge
  constexpr int VPL = D_ / (4 * LPE);           // float4 pieces per lane
  const int64_t n_scored = rel_ptr[n_rel];
  const int lane = threadIdx.x % kWave;
  const int li = lane % LPE;
  const int64_t wv = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
  const int64_t p0 = wv * kWave;
  if (p0 >= n_edges) return;
  const int64_t p = p0 + lane;
  if (p0 >= n_scored) {  // relation ids outside [0, R): logit 0
    if (p < n_edges) {
      if (LOGITS_EID) logits[perm[p]] = 0.f;
      if (logits_csr) logits_csr[pos_g[p]] = 0.f;
    }
    return;
  }
  const int64_t pc = p < n_scored ? p : n_scored - 1;
  const int32_t my_row = src_g[pc], my_g = gid[pc];
  const int32_t oe = LOGITS_EID ? perm[p < n_edges ? p : n_edges - 1] : 0;
  const int32_t op = logits_csr ? pos_g[p < n_edges ? p : n_edges - 1] : 0;
  const uint32_t e_off = (uint32_t)my_row * (uint32_t)(D_ * 4), v_off_lo = (uint32_t)li * 16u;
  const char* eb = reinterpret_cast<const char*>(ent) + v_off_lo;
  const char* vb = reinterpret_cast<const char*>(V_tab) + v_off_lo;
  float mine = 0.f;
  constexpr int U = (8 / VPL < 1 ? 1 : 8 / VPL) < LPE ? (8 / VPL < 1 ? 1 : 8 / VPL) : LPE;  // steps whose loads (<= 16 float4 per lane) are in flight together
#pragma unroll
  for (int s0 = 0; s0 < LPE; s0 += U) {
    float4 a[U][VPL], b[U][VPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int from = (lane - li + s0 + u) << 2;  // lane (group base + step)
      const uint32_t eo = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)e_off);
      const int32_t g = __builtin_amdgcn_ds_bpermute(from, my_g);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        a[u][v] = *reinterpret_cast<const float4*>(eb + eo + v * (LPE * 16));
        b[u][v] = *reinterpret_cast<const float4*>(vb + (size_t)g * (D_ * 4) + v * (LPE * 16));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = a[u][0].x * b[u][0].x;
      d = fmaf(a[u][0].y, b[u][0].y, d);
      d = fmaf(a[u][0].z, b[u][0].z, d);
      d = fmaf(a[u][0].w, b[u][0].w, d);
#pragma unroll
      for (int v = 1; v < VPL; ++v) {
        d = fmaf(a[u][v].x, b[u][v].x, d);
        d = fmaf(a[u][v].y, b[u][v].y, d);
        d = fmaf(a[u][v].z, b[u][v].z, d);
        d = fmaf(a[u][v].w, b[u][v].w, d);
      }
      if (LPE >= 2) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xF, 0xF, true));
      if (LPE >= 4) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xF, 0xF, true));
      if (LPE >= 8) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xF, 0xF, true));
      if (LPE >= 16) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x140, 0xF, 0xF, true));
      mine = li == s0 + u ? d : mine;
    }
  }
  if (p < n_scored) {
    if (LOGITS_EID) logits[oe] = mine;
    if (logits_csr) logits_csr[op] = mine;
  } else if (p < n_edges) {
    if (LOGITS_EID) logits[oe] = 0.f;
    if (logits_csr) logits_csr[op] = 0.f;
  }
}

// Folded head kernel for small widths without an MFMA tile shape (d, k <= 32, d and k independent;
// BASELINE configs[0] runs d = k = 8): one THREAD per (head, relation) group, W_r (<= 4 KB) in LDS,
// V[g] = W_r tanh(W_r^T e_h + e_r) in two register loops.  A workgroup owns a contiguous range of
// groups and reloads W_r when its range crosses into the next relation.
constexpr int kSmallMaxDim = 32;

__global__ __launch_bounds__(256) void att_fold_head_small_kernel(
    int d, int k, int n_rel, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ V_tab) {
  __shared__ float s_w[kSmallMaxDim * kSmallMaxDim];
  __shared__ float s_r[kSmallMaxDim];
  const int32_t n_groups = gptr[n_rel];
  const int32_t g_begin = (int32_t)((int64_t)n_groups * blockIdx.x / gridDim.x);
  const int32_t g_end = (int32_t)((int64_t)n_groups * (blockIdx.x + 1) / gridDim.x);
  int32_t g0 = g_begin;
  while (g0 < g_end) {  // workgroup-uniform loop over relation segments
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (gptr[mid] <= g0) lo = mid; else hi = mid;
    }
    // (empty relations share their gptr value with the next one: take the last of them)
    while (lo + 1 < n_rel && gptr[lo + 1] <= g0) ++lo;
    const int r = lo;
    const int32_t seg_end = gptr[r + 1] < g_end ? gptr[r + 1] : g_end;
    __syncthreads();
    for (int t = threadIdx.x; t < d * k; t += 256) s_w[t] = W_R[(size_t)r * d * k + t];
    for (int t = threadIdx.x; t < k; t += 256) s_r[t] = rel[(size_t)r * k + t];
    __syncthreads();
    for (int32_t g = g0 + threadIdx.x; g < seg_end; g += 256) {
      const float* eh = ent + (size_t)g_node[g] * d;
      float x[kSmallMaxDim], tt[kSmallMaxDim];
#pragma unroll
      for (int i = 0; i < kSmallMaxDim; ++i) x[i] = i < d ? eh[i] : 0.f;
#pragma unroll
      for (int j = 0; j < kSmallMaxDim; ++j) {
        float acc = 0.f;
        if (j < k) {
#pragma unroll
          for (int i = 0; i < kSmallMaxDim; ++i)
            if (i < d) acc = fmaf(x[i], s_w[i * k + j], acc);
          acc = att_tanh<0>(acc + s_r[j]);
        }
        tt[j] = acc;
      }
#pragma unroll
      for (int i = 0; i < kSmallMaxDim; ++i) {
        if (i < d) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < kSmallMaxDim; ++j)
            if (j < k) acc = fmaf(s_w[i * k + j], tt[j], acc);
          V_tab[(size_t)g * d + i] = acc;
        }
      }
    }
    g0 = seg_end;
  }
}

// The same for d == k == D known at compile time (configs[0]'s d = k = 8): the two D x D register loops
// without the runtime width guards of the general kernel (whose 32 x 32 unrolled loops are mostly predicated
// away at d = 8), the head row read and the V row written as 16-byte accesses.  Same arithmetic, same order.
template <int D>
__global__ __launch_bounds__(256) void att_fold_head_small_dk_kernel(
    int n_rel, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node, const float* __restrict__ ent,
    const float* __restrict__ W_R, const float* __restrict__ rel, float* __restrict__ V_tab) {
  __shared__ float s_w[D * D];
  __shared__ float s_r[D];
  const int32_t n_groups = gptr[n_rel];
  const int32_t g_begin = (int32_t)((int64_t)n_groups * blockIdx.x / gridDim.x);
  const int32_t g_end = (int32_t)((int64_t)n_groups * (blockIdx.x + 1) / gridDim.x);
  int32_t g0 = g_begin;
  while (g0 < g_end) {  // workgroup-uniform loop over relation segments
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (gptr[mid] <= g0) lo = mid; else hi = mid;
    }
    while (lo + 1 < n_rel && gptr[lo + 1] <= g0) ++lo;
    const int r = lo;
    const int32_t seg_end = gptr[r + 1] < g_end ? gptr[r + 1] : g_end;
    __syncthreads();
    for (int t = threadIdx.x; t < D * D; t += 256) s_w[t] = W_R[(size_t)r * D * D + t];
    if (threadIdx.x < D) s_r[threadIdx.x] = rel[(size_t)r * D + threadIdx.x];
    __syncthreads();
    for (int32_t g = g0 + threadIdx.x; g < seg_end; g += 256) {
      const float4* eh = reinterpret_cast<const float4*>(ent + (size_t)g_node[g] * D);
      float x[D], tt[D];
#pragma unroll
      for (int m = 0; m < D / 4; ++m) {
        const float4 v = eh[m];
        x[4 * m] = v.x; x[4 * m + 1] = v.y; x[4 * m + 2] = v.z; x[4 * m + 3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < D; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) acc = fmaf(x[i], s_w[i * D + j], acc);
        tt[j] = att_tanh<0>(acc + s_r[j]);
      }
      float4* vo = reinterpret_cast<float4*>(V_tab + (size_t)g * D);
#pragma unroll
      for (int m = 0; m < D / 4; ++m) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) acc = fmaf(s_w[(4 * m + c) * D + j], tt[j], acc);
          o[c] = acc;
        }
        vo[m] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    g0 = seg_end;
  }
}

static int launch_att_fold_head_small(int d, int k, const AttArgs& a, int64_t n_groups) {
  int64_t blocks = (n_groups + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (d == k && (d == 8 || d == 4) && (reinterpret_cast<uintptr_t>(a.ent) & 15u) == 0 &&
      (reinterpret_cast<uintptr_t>(a.G_tab) & 15u) == 0) {
    if (d == 8)
      hipLaunchKernelGGL(att_fold_head_small_dk_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, a.st, a.n_rel, a.gptr,
                         a.g_node, a.ent, a.W_R, a.rel, a.G_tab);
    else
      hipLaunchKernelGGL(att_fold_head_small_dk_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, a.st, a.n_rel, a.gptr,
                         a.g_node, a.ent, a.W_R, a.rel, a.G_tab);
    KGAT_CHECK_LAUNCH("att_fold_head_small_dk");
    return KGAT_OK;
  }
  hipLaunchKernelGGL(att_fold_head_small_kernel, dim3((unsigned)blocks), dim3(256), 0, a.st, d, k, a.n_rel, a.gptr,
                     a.g_node, a.ent, a.W_R, a.rel, a.G_tab);
  KGAT_CHECK_LAUNCH("att_fold_head_small");
  return KGAT_OK;
}

template <int D_>
static int launch_att_fold_tail(const AttArgs& a) {
  const int64_t waves = (a.n_edges + kWave - 1) / kWave;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  if (a.logits)
    hipLaunchKernelGGL((att_fold_tail_kernel<D_, true>), dim3(blocks), dim3(256), 0, a.st, a.n_rel, a.rel_ptr,
                       a.n_edges, a.src_g, a.gid, a.perm, a.pos_g, a.ent, a.G_tab, a.logits, a.logits_csr);
  else
    hipLaunchKernelGGL((att_fold_tail_kernel<D_, false>), dim3(blocks), dim3(256), 0, a.st, a.n_rel, a.rel_ptr,
                       a.n_edges, a.src_g, a.gid, a.perm, a.pos_g, a.ent, a.G_tab, a.logits, a.logits_csr);
  KGAT_CHECK_LAUNCH("att_fold_tail");
  return KGAT_OK;
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
  KGAT_CHECK_ARG(algo >= KGAT_ATT_ALGO_AUTO && algo <= KGAT_ATT_ALGO_MFMA_CHUNK, "att_score: bad algo");
  hipStream_t st = as_stream(stream);
  // edges whose type is outside [0, R) keep logit 0 (DGL zero-initialised column): the
  // persistent kernel clears that tail itself, the other kernels rely on a memset
  const bool persistent = (d == k) && (d == 16 || d == 32 || d == 64) && n_rel > 0 &&
                          n_rel <= kAttMaxRelLds && (unsigned long long)n_nodes * d * 4ull < (1ull << 32) &&
                          (algo == KGAT_ATT_ALGO_AUTO || algo == KGAT_ATT_ALGO_MFMA);
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
  const bool force_chunk = algo == KGAT_ATT_ALGO_MFMA_CHUNK;
  if (force_chunk) algo = KGAT_ATT_ALGO_MFMA;
  if (algo == KGAT_ATT_ALGO_MFMA && !mfma_ok) {
    set_error("att_score: the MFMA kernel covers d == k in {16,32,64,128}, got d=%d k=%d", d, k);
    return KGAT_E_UNSUPPORTED;
  }
  if (mfma_ok && algo != KGAT_ATT_ALGO_GENERIC) {
    switch (d) {
      case 16: return dispatch_att_mfma<16>(a, force_chunk);
      case 32: return dispatch_att_mfma<32>(a, force_chunk);
      case 64: return dispatch_att_mfma<64>(a, force_chunk);
      default: return dispatch_att_mfma<128>(a, force_chunk);
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

int kgat_att_score_split_supported(int64_t n_nodes, int d, int k, int n_rel) {
  return d == k && (d == 16 || d == 32 || d == 64) && n_rel > 0 && n_rel <= kAttMaxRelLds &&
         (unsigned long long)n_nodes * (unsigned long long)d * 4ull < (1ull << 32);
}

int kgat_att_score_split_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                             const int32_t* rel_ptr, const int32_t* perm, const int32_t* src_g,
                             const int32_t* pos_g, const int32_t* gid, const int32_t* gptr,
                             const int32_t* g_node, int64_t n_groups, const float* ent,
                             const float* W_R, const float* rel, float* G_tab, float* logits,
                             float* logits_csr, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_groups >= 0 && n_edges < INT32_MAX,
                 "att_score_split: bad size");
  if (n_edges == 0) return KGAT_OK;
  if (!kgat_att_score_split_supported(n_nodes, d, k, n_rel)) {
    set_error("att_score_split: needs d == k in {16,32,64}, 0 < R <= %d, N*d*4 < 4 GiB (d=%d k=%d R=%d)",
              kAttMaxRelLds, d, k, n_rel);
    return KGAT_E_UNSUPPORTED;
  }
  KGAT_CHECK_ARG(rel_ptr && perm && src_g && gid && gptr && ent && W_R && rel, "att_score_split: null pointer");
  KGAT_CHECK_ARG(logits || logits_csr, "att_score_split: no output requested");
  KGAT_CHECK_ARG(n_groups == 0 || (g_node && G_tab), "att_score_split: null group table");
  KGAT_CHECK_ARG(logits_csr == nullptr || pos_g != nullptr, "att_score_split: logits_csr needs pos_g");
  AttArgs a;
  a.grid = 0;
  a.st = as_stream(stream);
  a.n_rel = n_rel; a.rel_ptr = rel_ptr; a.perm = perm; a.src_g = src_g; a.dst_g = nullptr;
  a.ent = ent; a.W_R = W_R; a.rel = rel; a.logits = logits; a.logits_csr = logits_csr;
  a.pos_g = pos_g;
  a.table_bytes = (unsigned long long)n_nodes * (unsigned long long)d * 4ull;
  a.n_edges = n_edges;
  a.gid = gid; a.gptr = gptr; a.g_node = g_node; a.G_tab = G_tab;
  return launch_att_split_any(d, a);
}

static bool fold_small(int d, int k) {
  return (d == 8 || d == 16 || d == 32) && k >= 1 && k <= kSmallMaxDim && !(d == k && d >= 16);
}

int kgat_att_score_folded_supported(int64_t n_nodes, int d, int k, int n_rel) {
  const bool mfma = d == k && (d == 16 || d == 32 || d == 64 || d == 128);
  return (mfma || fold_small(d, k)) && n_rel > 0 && n_rel <= kAttMaxRelLds &&
         (unsigned long long)n_nodes * (unsigned long long)d * 4ull < (1ull << 32);
}

int kgat_att_score_folded_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                              const int32_t* rel_ptr, const int32_t* perm, const int32_t* src_g,
                              const int32_t* pos_g, const int32_t* gid, const int32_t* gptr,
                              const int32_t* g_node, int64_t n_groups, const float* ent,
                              const float* W_R, const float* rel, float* V_tab, float* logits,
                              float* logits_csr, int flags, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_groups >= 0 && n_edges < INT32_MAX,
                 "att_score_folded: bad size");
  KGAT_CHECK_ARG((flags & ~KGAT_ATT_F32_PRODUCTS) == 0, "att_score_folded: unknown flag");
  if (n_edges == 0) return KGAT_OK;
  if (!kgat_att_score_folded_supported(n_nodes, d, k, n_rel)) {
    set_error("att_score_folded: needs d == k in {16,32,64,128} or d in {8,16,32} with k <= 32, 0 < R <= %d, N*d*4 < 4 GiB (d=%d k=%d R=%d)",
              kAttMaxRelLds, d, k, n_rel);
    return KGAT_E_UNSUPPORTED;
  }
  KGAT_CHECK_ARG(rel_ptr && perm && src_g && gid && gptr && ent && W_R && rel, "att_score_folded: null pointer");
  KGAT_CHECK_ARG(logits || logits_csr, "att_score_folded: no output requested");
  KGAT_CHECK_ARG(n_groups == 0 || (g_node && V_tab), "att_score_folded: null group table");
  KGAT_CHECK_ARG(logits_csr == nullptr || pos_g != nullptr, "att_score_folded: logits_csr needs pos_g");
  KGAT_CHECK_ARG((unsigned long long)n_groups * (unsigned long long)d * 4ull < (1ull << 40),
                 "att_score_folded: group table too large");
  AttArgs a;
  a.grid = 0;
  a.st = as_stream(stream);
  a.n_rel = n_rel; a.rel_ptr = rel_ptr; a.perm = perm; a.src_g = src_g; a.dst_g = nullptr;
  a.ent = ent; a.W_R = W_R; a.rel = rel; a.logits = logits; a.logits_csr = logits_csr;
  a.pos_g = pos_g;
  a.n_edges = n_edges;
  a.gid = gid; a.gptr = gptr; a.g_node = g_node; a.G_tab = V_tab;
  a.f32_products = (flags & KGAT_ATT_F32_PRODUCTS) != 0;
  if (n_groups > 0) {
    const int rc = fold_small(d, k) ? launch_att_fold_head_small(d, k, a, n_groups) : launch_att_fold_head_any(d, a);
    if (rc != KGAT_OK) return rc;
  }
  switch (d) {
    case 8: return launch_att_fold_tail<8>(a);
    case 16: return launch_att_fold_tail<16>(a);
    case 32: return launch_att_fold_tail<32>(a);
    case 64: return launch_att_fold_tail<64>(a);
    default: return launch_att_fold_tail<128>(a);
  }
}

int kgat_att_score_fused_supported(int64_t n_nodes, int d, int k, int n_rel) {
  // d = k = 128 (round 3): bf16-piece products only (KGAT_ATT_F32_PRODUCTS at that width is
  // KGAT_E_UNSUPPORTED here; the two-launch folded form has it)
  return d == k && (d == 16 || d == 32 || d == 64 || d == 128) && n_rel > 0 && n_rel <= kAttMaxRelLds &&
         n_nodes <= (1ll << 28) && (unsigned long long)n_nodes * (unsigned long long)d * 4ull < (1ull << 32);
}

static int att_score_fused_run(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                               const int32_t* rel_ptr, const int32_t* perm, const int32_t* rec_g,
                               const int32_t* pos_g, const int32_t* gptr,
                               const int32_t* g_node, const int32_t* tiles, const int32_t* rel_tptr,
                               const int32_t* part_tptr, int n_parts,
                               const float* ent, const float* W_R, const float* rel, float* logits,
                               float* logits_csr, float* logits_g, int flags, long long* part_clocks,
                               kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < INT32_MAX, "att_score_fused: bad size");
  KGAT_CHECK_ARG((flags & ~(KGAT_ATT_F32_PRODUCTS | KGAT_ATT_TILES32)) == 0, "att_score_fused: unknown flag");
  if ((flags & KGAT_ATT_TILES32) && (d != 64 || k != 64 || (flags & KGAT_ATT_F32_PRODUCTS))) {
    set_error("att_score_fused: 32-group tiles exist at d = k = 64 with the piece products only");
    return KGAT_E_UNSUPPORTED;
  }
  if (n_edges == 0) return KGAT_OK;
  if (!kgat_att_score_fused_supported(n_nodes, d, k, n_rel)) {
    set_error("att_score_fused: needs d == k in {16,32,64,128}, 0 < R <= %d, N*d*4 < 4 GiB (d=%d k=%d R=%d)",
              kAttMaxRelLds, d, k, n_rel);
    return KGAT_E_UNSUPPORTED;
  }
  if (d == 128 && (flags & KGAT_ATT_F32_PRODUCTS)) {
    set_error("att_score_fused: d = 128 runs the bf16-piece products only (the folded form has KGAT_ATT_F32_PRODUCTS)");
    return KGAT_E_UNSUPPORTED;
  }
  KGAT_CHECK_ARG(n_nodes <= ((flags & KGAT_ATT_TILES32) ? (1ll << 27) : (1ll << 28)),
                 "att_score_fused: packed records hold node ids below 2^28 (2^27 with 32-group tiles)");
  KGAT_CHECK_ARG(rel_ptr && rec_g && gptr && g_node && tiles && rel_tptr && ent && W_R && rel,
                 "att_score_fused: null pointer");
  KGAT_CHECK_ARG(logits || logits_csr || logits_g, "att_score_fused: no output requested");
  KGAT_CHECK_ARG(logits == nullptr || perm != nullptr, "att_score_fused: edge-id ordered logits need perm");
  KGAT_CHECK_ARG(logits_csr == nullptr || pos_g != nullptr, "att_score_fused: logits_csr needs pos_g");
  KGAT_CHECK_ARG((part_tptr == nullptr) == (n_parts == 0) && n_parts >= 0 && n_parts <= 65536,
                 "att_score_fused: part_tptr and n_parts go together");
  AttArgs a;
  a.grid = (unsigned)n_parts;
  a.part_tptr = part_tptr;
  a.f32_products = (flags & KGAT_ATT_F32_PRODUCTS) != 0;
  a.st = as_stream(stream);
  a.n_rel = n_rel; a.rel_ptr = rel_ptr; a.perm = perm; a.src_g = nullptr; a.dst_g = nullptr;
  a.ent = ent; a.W_R = W_R; a.rel = rel; a.logits = logits; a.logits_csr = logits_csr;
  a.pos_g = pos_g;
  a.n_edges = n_edges;
  a.gid = nullptr; a.gptr = gptr; a.g_node = g_node;
  a.rec_g = rec_g; a.logits_g = logits_g;
  a.part_clocks = part_clocks;
  if (flags & KGAT_ATT_TILES32) {
    KGAT_CHECK_ARG(part_clocks == nullptr, "att_score_fused_timed: not with 32-group tiles");
    return launch_att_fold_fused32(a, rel_tptr, tiles);
  }
  return launch_att_fold_fused_any(d, a, rel_tptr, tiles);
}

int kgat_att_score_fused_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                             const int32_t* rel_ptr, const int32_t* perm, const int32_t* rec_g,
                             const int32_t* pos_g, const int32_t* gptr,
                             const int32_t* g_node, const int32_t* tiles, const int32_t* rel_tptr,
                             const int32_t* part_tptr, int n_parts,
                             const float* ent, const float* W_R, const float* rel, float* logits,
                             float* logits_csr, float* logits_g, int flags, kgat_stream_t stream) {
  return att_score_fused_run(n_nodes, n_edges, d, k, n_rel, rel_ptr, perm, rec_g, pos_g, gptr, g_node, tiles, rel_tptr,
                             part_tptr, n_parts, ent, W_R, rel, logits, logits_csr, logits_g, flags, nullptr, stream);
}

int kgat_att_score_fused_timed_f32(int64_t n_nodes, int64_t n_edges, int d, int k, int n_rel,
                                   const int32_t* rel_ptr, const int32_t* perm, const int32_t* rec_g,
                                   const int32_t* pos_g, const int32_t* gptr,
                                   const int32_t* g_node, const int32_t* tiles, const int32_t* rel_tptr,
                                   const int32_t* part_tptr, int n_parts,
                                   const float* ent, const float* W_R, const float* rel, float* logits,
                                   float* logits_csr, float* logits_g, int flags, long long* part_clocks,
                                   kgat_stream_t stream) {
  KGAT_CHECK_ARG(part_clocks != nullptr && part_tptr != nullptr && n_parts > 0,
                 "att_score_fused_timed: needs part_tptr and 2 * n_parts clock slots");
  return att_score_fused_run(n_nodes, n_edges, d, k, n_rel, rel_ptr, perm, rec_g, pos_g, gptr, g_node, tiles, rel_tptr,
                             part_tptr, n_parts, ent, W_R, rel, logits, logits_csr, logits_g, flags, part_clocks, stream);
}

}  // extern "C"
