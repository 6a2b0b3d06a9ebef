// Persistent-wavefront attention-logit kernels (d == k <= 64): the one-kernel form
// (att_score_persistent_kernel) and the split form (att_split_kernel, MODE_HEAD / MODE_TAIL);
// overview in kgat_att.hip, measurements and the cost model in DESIGN.md 3.2 / NOTEBOOK.md 3.2.
// Separate translation unit because it is built with -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA
// results in VGPRs, no v_accvgpr_read copies in the epilogue), which the LDS-staged chunk
// kernels of kgat_att.hip do not want (it costs them occupancy).

#include "kgat_att_common.h"

namespace kgat {

// ---------------------------------------------------------------------------------------------
// Persistent-wavefront form (d == k <= 64): W_r lives in registers as MFMA B fragments (64
// VGPRs at d = 64), every wavefront owns a contiguous, equally sized range of 16-edge tiles of
// the relation-grouped edge list (so the launch cannot end on a partly filled round of
// workgroups), A fragments are double buffered and requested one tile ahead, edge indices two
// tiles ahead.  No LDS traffic and no barrier inside the tile loop; W_r is re-read from L2 only
// when a wave's range crosses into the next relation.

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
  const int32_t* __restrict__ pos_or_perm = logits_csr ? pos_g : perm;
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
      x.op = pos_or_perm[po];  // unconditional: a branch around a load makes the counted waits conservative
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

template <int D_, int ACC_TANH>
static int launch_att_persistent(const AttArgs& a) {
  // one resident workgroup per CU slot the kernel's register budget admits: the tile ranges
  // are split evenly over exactly the wavefronts that run concurrently
  static int blocks_per_cu = 0;
  const int cus = device_cu_count();
  if (blocks_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, att_score_persistent_kernel<D_, ACC_TANH>,
                                                     kAttThreads, 0) != hipSuccess || nb < 1)
      nb = 1;
    blocks_per_cu = nb > 8 ? 8 : nb;
  }
  const int per_cu = blocks_per_cu;
  const unsigned grid = (unsigned)(cus * per_cu);  // 4 waves per block, one per SIMD
  hipLaunchKernelGGL((att_score_persistent_kernel<D_, ACC_TANH>), dim3(grid), dim3(kAttThreads), 0, a.st,
                     a.n_rel, a.n_edges, a.rel_ptr, a.perm, a.src_g, a.dst_g, a.ent, a.W_R, a.rel, a.logits,
                     a.logits_csr, a.pos_g);
  KGAT_CHECK_LAUNCH("att_score_persistent");
  return KGAT_OK;
}


// ---------------------------------------------------------------------------------------------
// Split form: the head projection tanh(e_h W_r + e_r) depends only on (head, relation).  With
// every relation's edges sorted by destination, the edges of one (head, relation) pair are
// consecutive ("head group"), so the projection is computed once per group (MODE_HEAD, writes
// the G table) and the per-edge kernel (MODE_TAIL) only projects the tail and takes the dot
// product with its group's G row - half the MFMAs and none of the tanh work per edge.  The
// arithmetic per edge is unchanged (same fma chains, same reduction order), so results are
// bit-identical to the one-kernel form.
constexpr int MODE_HEAD = 1, MODE_TAIL = 2;

template <int D_, int MODE, bool LOGITS_EID>
__global__ __launch_bounds__(kAttThreads) void att_split_kernel(
    int n_rel, int64_t n_edges, const int32_t* __restrict__ seg_ptr /* gptr | rel_ptr */,
    const int32_t* __restrict__ row_idx /* g_node | src_g */, const int32_t* __restrict__ gid,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ pos_g,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ G_tab, float* __restrict__ logits, float* __restrict__ logits_csr) {
  constexpr int K_ = D_;
  constexpr int KS = D_ / 4, KT = K_ / 16;
  __shared__ int32_t s_tptr[kAttMaxRelLds + 1];  // tile prefix per relation
  const int tid = threadIdx.x;
  for (int r = tid; r < n_rel; r += kAttThreads)
    s_tptr[r + 1] = (seg_ptr[r + 1] - seg_ptr[r] + 15) >> 4;
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
  const int32_t* __restrict__ pos_or_perm = logits_csr ? pos_g : perm;
  const int64_t n_waves = (int64_t)gridDim.x * (kAttThreads / kWave);
  const int64_t wv = (int64_t)blockIdx.x * (kAttThreads / kWave) + tid / kWave;
  const int32_t t_begin = (int32_t)((int64_t)n_tiles * wv / n_waves);
  const int32_t t_end = (int32_t)((int64_t)n_tiles * (wv + 1) / n_waves);

  if (MODE == MODE_TAIL) {  // unscored tail of the edge list: logit 0
    const int64_t tail0 = seg_ptr[n_rel];
    const int64_t n_tail = n_edges - tail0;
    for (int64_t p = tail0 + n_tail * wv / n_waves + lane; p < tail0 + n_tail * (wv + 1) / n_waves; p += kWave) {
      if (LOGITS_EID) logits[perm[p]] = 0.f;
      if (logits_csr) logits_csr[pos_g[p]] = 0.f;
    }
  }
  if (t_begin >= t_end) return;

  float wreg[KS][KT];
  float relv[KT];
  int32_t t = t_begin;
  while (t < t_end) {
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_tptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int r = __builtin_amdgcn_readfirstlane(lo);
    const int32_t rbeg = __builtin_amdgcn_readfirstlane(seg_ptr[r]);
    const int32_t rend = __builtin_amdgcn_readfirstlane(seg_ptr[r + 1]);
    const int32_t tfirst = __builtin_amdgcn_readfirstlane(s_tptr[r]);
    int32_t seg_end = __builtin_amdgcn_readfirstlane(s_tptr[r + 1]);
    seg_end = seg_end < t_end ? seg_end : t_end;
    const int32_t n_seg = seg_end - t;
    const int32_t pe0 = rbeg + ((t - tfirst) << 4);
    {
      const float* W = W_R + (size_t)r * D_ * K_;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int krow = 16 * (s >> 2) + 4 * q + (s & 3);
#pragma unroll
        for (int c = 0; c < KT; ++c) wreg[s][c] = W[krow * K_ + 16 * c + i];
      }
      if (MODE == MODE_HEAD) {
#pragma unroll
        for (int c = 0; c < KT; ++c) relv[c] = rel[(size_t)r * K_ + 16 * c + i] * kTwoLog2e;
      }
    }

    // Per-tile index sets.  L = what the row/G loads of a tile need (source row, group ids);
    // O = where the tile's results go (edge id, CSR position).  Both are clamped past the
    // segment end so that every step issues the same loads.
    struct alignas(KT * 4) GVec { float v[KT]; };  // G table row layout: [column slot i][tile c]
    struct __attribute__((packed, aligned(4))) Gid4 { int32_t v[4]; };
    struct LIdx { int32_t row, g[4]; };
    struct OIdx { int32_t oe, op; };
    auto load_l = [&](int32_t n, LIdx& x) {
      n = n < n_seg ? n : n_seg - 1;
      const int32_t base = pe0 + (n << 4);
      int32_t pe = base + i;
      pe = pe < rend ? pe : rend - 1;
      x.row = row_idx[pe];
      if (MODE == MODE_TAIL) {
        // group ids of edges 4q .. 4q+3: one 16-byte load (positions past the relation's end
        // read the next relation's ids or the array's padding; their results are never stored)
        const Gid4 g4 = *reinterpret_cast<const Gid4*>(gid + base + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) x.g[j] = g4.v[j];
      }
    };
    auto load_o = [&](int32_t n, OIdx& x) {
      if (MODE == MODE_TAIL) {
        n = n < n_seg ? n : n_seg - 1;
        int32_t po = pe0 + (n << 4) + 4 * q + (i & 3);
        po = po < rend ? po : rend - 1;
        x.oe = LOGITS_EID ? perm[po] : 0;
        x.op = pos_or_perm[po];  // unconditional: a branch around a load makes the counted waits conservative
      }
    };
    struct Buf { float a[KS]; float g[KT][4]; };
    auto load_rows = [&](Buf& f, const LIdx& x) {
      const char* base = reinterpret_cast<const char*>(ent);
      const uint32_t o = (uint32_t)x.row * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
#pragma unroll
      for (int m = 0; m < D_ / 16; ++m) {
        const float4 v = *reinterpret_cast<const float4*>(base + o + m * 64);
        f.a[4 * m + 0] = v.x; f.a[4 * m + 1] = v.y; f.a[4 * m + 2] = v.z; f.a[4 * m + 3] = v.w;
      }
      if (MODE == MODE_TAIL) {  // the KT values of column slot i of group g sit together: one load per edge
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const GVec v = *reinterpret_cast<const GVec*>(G_tab + (size_t)x.g[j] * K_ + i * KT);
#pragma unroll
          for (int c = 0; c < KT; ++c) f.g[c][j] = v.v[c];
        }
      }
    };
    auto tile = [&](int32_t n, const Buf& f, const OIdx& x) {
      floatx4 acc[KT];
#pragma unroll
      for (int cc = 0; cc < KT; ++cc) acc[cc] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int cc = 0; cc < KT; ++cc)
          acc[cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s], wreg[s][cc], acc[cc], 0, 0, 0);
      const int32_t item0 = pe0 + (n << 4) + 4 * q;  // this lane's rows: item0 + j
      if (MODE == MODE_HEAD) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (item0 + j < rend) {
            GVec v;
#pragma unroll
            for (int cc = 0; cc < KT; ++cc) v.v[cc] = att_tanh_scaled(fmaf(acc[cc][j], kTwoLog2e, relv[cc]));
            *reinterpret_cast<GVec*>(G_tab + (size_t)(item0 + j) * K_ + i * KT) = v;
          }
      } else {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < KT; ++cc)
#pragma unroll
          for (int j = 0; j < 4; ++j) part[j] = fmaf(acc[cc][j], f.g[cc][j], part[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) part[j] = row16_sum(part[j]);
        const float v = i == 0 ? part[0] : (i == 1 ? part[1] : (i == 2 ? part[2] : part[3]));
        if (i < 4 && item0 + i < rend) {
          if (LOGITS_EID) logits[x.oe] = v;
          if (logits_csr) logits_csr[x.op] = v;
        }
      }
    };

    // Three-deep ring: step n computes tile n while the rows of tiles n+1 and n+2 are in
    // flight (a tile is ~2,700 cycles of issue here, less than the gather latency under load,
    // so one tile of look-ahead is not enough).  Inside a step the small index loads are
    // issued BEFORE the row loads: vmcnt retires in order, and the next step's wait for
    // those indices must not have to wait for this step's row loads behind them.
    Buf b0, b1, b2;
    LIdx lc, ln;
    OIdx oc, on;
    {
      LIdx l0, l1;
      load_l(0, l0);
      load_l(1, l1);
      load_l(2, lc);
      load_o(0, oc);
      load_rows(b0, l0);
      load_rows(b1, l1);
    }
#define KGAT_SPLIT_STEP(BCUR, BFILL)                 \
    {                                                \
      load_l(n + 3, ln);                             \
      load_o(n + 1, on);                             \
      load_rows(BFILL, lc);                          \
      __builtin_amdgcn_sched_barrier(0);             \
      tile(n, BCUR, oc);                             \
      __builtin_amdgcn_sched_barrier(0);             \
      lc = ln;                                       \
      oc = on;                                       \
      ++n;                                           \
    }
    int32_t n = 0;
    while (n < n_seg) {
      KGAT_SPLIT_STEP(b0, b2)
      if (n >= n_seg) break;
      KGAT_SPLIT_STEP(b1, b0)
      if (n >= n_seg) break;
      KGAT_SPLIT_STEP(b2, b1)
    }
#undef KGAT_SPLIT_STEP
    t = seg_end;
  }
}

template <int D_, int MODE, bool LOGITS_EID>
static int launch_att_split(const AttArgs& a, const int32_t* seg_ptr, const int32_t* row_idx) {
  static int blocks_per_cu = 0;
  const int cus = device_cu_count();
  if (blocks_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, att_split_kernel<D_, MODE, LOGITS_EID>, kAttThreads,
                                                     0) != hipSuccess || nb < 1)
      nb = 1;
    blocks_per_cu = nb > 8 ? 8 : nb;
  }
  hipLaunchKernelGGL((att_split_kernel<D_, MODE, LOGITS_EID>), dim3((unsigned)(cus * blocks_per_cu)),
                     dim3(kAttThreads), 0, a.st, a.n_rel, a.n_edges, seg_ptr, row_idx, a.gid, a.perm, a.pos_g,
                     a.ent, a.W_R, a.rel, a.G_tab, a.logits, a.logits_csr);
  KGAT_CHECK_LAUNCH("att_split");
  return KGAT_OK;
}

template <int D_>
static int launch_att_split_d(const AttArgs& a) {
  int rc = launch_att_split<D_, MODE_HEAD, false>(a, a.gptr, a.g_node);
  if (rc != KGAT_OK) return rc;
  return a.logits ? launch_att_split<D_, MODE_TAIL, true>(a, a.rel_ptr, a.src_g)
                  : launch_att_split<D_, MODE_TAIL, false>(a, a.rel_ptr, a.src_g);
}

// ---------------------------------------------------------------------------------------------
// Folded form.  The logit is bilinear in the tail row:
//     sum_j (e_t W_r)_j * T_j  =  e_t . (W_r T),        T = tanh(e_h W_r + e_r),
// so the whole relation-space work can be done once per (head, relation) group:
// V[g] = W_r tanh(W_r^T e_h + e_r) (a d-vector), after which an edge costs one d-length dot
// product with its tail row (att_fold_tail_kernel in kgat_att.hip) instead of a d x k
// projection.  The contraction order differs from the reference's (sum over k first there,
// over d last here): results agree to fp32 rounding (~1e-6 relative), not bit for bit.
//
// One wavefront owns 16-group tiles.  Both products run on the MFMA unit without a transpose
// in between: the first one is issued with the operands swapped (A = W_r fragment, B = head
// rows), which leaves G^T in the accumulators - lane (i, q), register (c, j) holds
// G[group i][column 16c + 4q + j] - and that is exactly a B fragment of the second product if
// its contraction steps are taken in the order (c, j), with W_r's A fragments loaded to match.
template <int D_>
__global__ __launch_bounds__(kAttThreads) void att_fold_head_kernel(
    int n_rel, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ V_tab) {
  constexpr int K_ = D_;
  constexpr int KS = D_ / 4, KT = K_ / 16;
  __shared__ int32_t s_tptr[kAttMaxRelLds + 1];  // tile prefix per relation
  const int tid = threadIdx.x;
  for (int r = tid; r < n_rel; r += kAttThreads) s_tptr[r + 1] = (gptr[r + 1] - gptr[r] + 15) >> 4;
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
  if (t_begin >= t_end) return;

  float w1[KS][KT];      // W_r[krow(s, q)][16c + i]        (first product, as in the other kernels)
  float w2[KT][KT][4];   // W_r[16c' + i][16c + 4q + j]     (second product, contraction step (c, j))
  float relv[KT][4];     // e_r[16c + 4q + j] * 2 log2(e)
  int32_t t = t_begin;
  while (t < t_end) {
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_tptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int r = __builtin_amdgcn_readfirstlane(lo);
    const int32_t rbeg = __builtin_amdgcn_readfirstlane(gptr[r]);
    const int32_t rend = __builtin_amdgcn_readfirstlane(gptr[r + 1]);
    const int32_t tfirst = __builtin_amdgcn_readfirstlane(s_tptr[r]);
    int32_t seg_end = __builtin_amdgcn_readfirstlane(s_tptr[r + 1]);
    seg_end = seg_end < t_end ? seg_end : t_end;
    const int32_t n_seg = seg_end - t;
    const int32_t g0 = rbeg + ((t - tfirst) << 4);
    {
      const float* W = W_R + (size_t)r * D_ * K_;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int krow = 16 * (s >> 2) + 4 * q + (s & 3);
#pragma unroll
        for (int c = 0; c < KT; ++c) w1[s][c] = W[krow * K_ + 16 * c + i];
      }
#pragma unroll
      for (int c2 = 0; c2 < KT; ++c2)
#pragma unroll
        for (int c = 0; c < KT; ++c) {
          const float4 v = *reinterpret_cast<const float4*>(W + (16 * c2 + i) * K_ + 16 * c + 4 * q);
          w2[c2][c][0] = v.x; w2[c2][c][1] = v.y; w2[c2][c][2] = v.z; w2[c2][c][3] = v.w;
        }
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(rel + (size_t)r * K_ + 16 * c + 4 * q);
        relv[c][0] = v.x * kTwoLog2e; relv[c][1] = v.y * kTwoLog2e;
        relv[c][2] = v.z * kTwoLog2e; relv[c][3] = v.w * kTwoLog2e;
      }
    }

    auto load_idx = [&](int32_t n) -> int32_t {
      n = n < n_seg ? n : n_seg - 1;
      int32_t g = g0 + (n << 4) + i;
      g = g < rend ? g : rend - 1;
      return g_node[g];
    };
    struct Buf { float a[KS]; };
    auto load_rows = [&](Buf& f, int32_t row) {
      const char* base = reinterpret_cast<const char*>(ent);
      const uint32_t o = (uint32_t)row * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
#pragma unroll
      for (int m = 0; m < D_ / 16; ++m) {
        const float4 v = *reinterpret_cast<const float4*>(base + o + m * 64);
        f.a[4 * m + 0] = v.x; f.a[4 * m + 1] = v.y; f.a[4 * m + 2] = v.z; f.a[4 * m + 3] = v.w;
      }
    };
    auto tile = [&](int32_t n, const Buf& f) {
      floatx4 acc[KT];
#pragma unroll
      for (int c = 0; c < KT; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int c = 0; c < KT; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s][c], f.a[s], acc[c], 0, 0, 0);
      // acc[c][j] = (e_h W_r)[group i][16c + 4q + j]
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = att_tanh_scaled(fmaf(acc[c][j], kTwoLog2e, relv[c][j]));
      floatx4 v[KT];
#pragma unroll
      for (int c2 = 0; c2 < KT; ++c2) v[c2] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c2 = 0; c2 < KT; ++c2)
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[c2][c][j], acc[c][j], v[c2], 0, 0, 0);
      // v[c2][j] = V[group i][16c2 + 4q + j]
      const int32_t g = g0 + (n << 4) + i;
      if (g < rend) {
#pragma unroll
        for (int c2 = 0; c2 < KT; ++c2) {
          float4 o;
          o.x = v[c2][0]; o.y = v[c2][1]; o.z = v[c2][2]; o.w = v[c2][3];
          *reinterpret_cast<float4*>(V_tab + (size_t)g * D_ + 16 * c2 + 4 * q) = o;
        }
      }
    };

    // three-deep ring, indices requested before rows (see att_split_kernel)
    Buf b0, b1, b2;
    int32_t rc;
    {
      const int32_t r0 = load_idx(0), r1 = load_idx(1);
      rc = load_idx(2);
      load_rows(b0, r0);
      load_rows(b1, r1);
    }
#define KGAT_FOLD_STEP(BCUR, BFILL)                  \
    {                                                \
      const int32_t rn = load_idx(n + 3);            \
      load_rows(BFILL, rc);                          \
      __builtin_amdgcn_sched_barrier(0);             \
      tile(n, BCUR);                                 \
      __builtin_amdgcn_sched_barrier(0);             \
      rc = rn;                                       \
      ++n;                                           \
    }
    int32_t n = 0;
    while (n < n_seg) {
      KGAT_FOLD_STEP(b0, b2)
      if (n >= n_seg) break;
      KGAT_FOLD_STEP(b1, b0)
      if (n >= n_seg) break;
      KGAT_FOLD_STEP(b2, b1)
    }
#undef KGAT_FOLD_STEP
    t = seg_end;
  }
}

template <int D_>
static int launch_att_fold_head(const AttArgs& a) {
  static int blocks_per_cu = 0;
  const int cus = device_cu_count();
  if (blocks_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, att_fold_head_kernel<D_>, kAttThreads, 0) != hipSuccess ||
        nb < 1)
      nb = 1;
    blocks_per_cu = nb > 8 ? 8 : nb;
  }
  hipLaunchKernelGGL((att_fold_head_kernel<D_>), dim3((unsigned)(cus * blocks_per_cu)), dim3(kAttThreads), 0, a.st,
                     a.n_rel, a.gptr, a.g_node, a.ent, a.W_R, a.rel, a.G_tab);
  KGAT_CHECK_LAUNCH("att_fold_head");
  return KGAT_OK;
}


// Products at fp32 accuracy on the bf16 matrix pipe (X3 = true in the kernels below; d = k a
// multiple of 32).  Every fp32 operand x is cut into three bf16 pieces x = h + m + l by rounding to
// nearest (v_cvt_pk_bf16_f32): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m).  Both subtractions
// are exact, |m| <= 2^-8 |x| (half an ulp of 8 significand bits), |x - h - m| <= 2^-16 |x| is a
// multiple of ulp_24(x) below 2^8 ulps, so l holds it exactly: the three pieces sum to x bit for
// bit, and every piece keeps the fp32 exponent range.  A product a*b is taken as the six piece
// products of weight >= 2^-16 (l*h, h*l, m*m, m*h, h*m, h*h), each exact in the fp32 accumulator of
// v_mfma_f32_16x16x32_bf16 and accumulated smallest first; the three dropped ones (m*l, l*m, l*l)
// are together <= 2 * 2^-8 * 2^-16 + 2^-32 < 2^-23 + 2^-30 of |a*b| - one fp32 ulp, of either sign
// (the remainders of a round-to-nearest cut are signed, so the dropped terms do not bias a sum; the
// truncating cut of round 2 left them one-signed and up to 2^-21).  Six MFMAs of 16 cycles for a
// 16x16x32 block against eight fp32 MFMAs of 32 cycles: 2.7 x fewer matrix-pipe cycles.  W_r's
// pieces are cut once per relation segment into LDS; the head rows and the tanh values are cut in
// registers (11 VALU instructions per pair of values).  Non-finite operands give NaN (Inf - Inf in
// the cut), where the fp32 products follow IEEE and can give +-Inf.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pair (a in the low half), round to nearest even.  Inline asm so that
// the two unpacks below read this one register (the compiler's own lowering of the cast re-converts
// the low element alone before shifting it).
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void split_bf16x3(const float (&x)[8], uintx4& h, uintx4& m, uintx4& l) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float x0 = x[2 * t], x1 = x[2 * t + 1];
    const unsigned hh = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hh << 16);
    const float r1 = x1 - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mm << 16);
    const float s1 = r1 - __uint_as_float(mm & 0xffff0000u);
    h[t] = hh;
    m[t] = mm;
    l[t] = cvt_pk_bf16(s0, s1);
  }
}

__device__ __forceinline__ floatx4 mfma_bf16(const uintx4& a, const uintx4& b, const floatx4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

// fp16 pieces for the second product of the d <= 64 fused kernel (kAttF16Second, default 1 since round 4;
// built and measured in round 3: W_r * 2^shift as three fp16 pieces, the tanh values * 2^14 as two, five piece
// products).  Measured on the amazon-book graph, d = 64: stand-alone 0.1555 vs 0.1614 ms, inside the step 137.3 vs
// 140.6 us; against fp64 max error 9.98e-7 / mean 1.09e-7 (three-bf16-piece form: 9.4e-7 / 1.23e-7; fp32 MFMA:
// 1.48e-6 / 1.43e-7), every GPU test unchanged (tests/test_gpu_parity.py::test_att_fused_product_forms, incl. the
// "small T" case that the 2^14 scale exists for).
// A value inside fp16's range is h + m + l with three round-to-nearest fp16 pieces EXACTLY (3 x 11 significand
// bits), and h + m with a residual <= 2^-22 |x|.  What the format buys is the cut: the remainder x - float(h) is
// ONE v_fma_mix_f32 per value (f16 operand taken straight from either half of the packed register), so a pair
// costs 4 instructions for two pieces and 7 for three, against 11 for three bf16 pieces (v_fma_mix_f32_bf16 is
// not a gfx950 instruction).  What it costs is the exponent range: operands are scaled by a power of two first.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sub_f16_lo(float x, unsigned hh) {  // x - float(low half of hh)
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(x));
  return r;
}
__device__ __forceinline__ float sub_f16_hi(float x, unsigned hh) {  // x - float(high half of hh)
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(x));
  return r;
}
__device__ __forceinline__ void split_f16x2(const float (&x)[8], uintx4& h, uintx4& m) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned hh = cvt_pk_f16(x[2 * t], x[2 * t + 1]);
    h[t] = hh;
    m[t] = cvt_pk_f16(sub_f16_lo(x[2 * t], hh), sub_f16_hi(x[2 * t + 1], hh));
  }
}
__device__ __forceinline__ void split_f16x3(const float (&x)[8], uintx4& h, uintx4& m, uintx4& l) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned hh = cvt_pk_f16(x[2 * t], x[2 * t + 1]);
    const float r0 = sub_f16_lo(x[2 * t], hh), r1 = sub_f16_hi(x[2 * t + 1], hh);
    const unsigned mm = cvt_pk_f16(r0, r1);
    h[t] = hh;
    m[t] = mm;
    l[t] = cvt_pk_f16(sub_f16_lo(r0, mm), sub_f16_hi(r1, mm));
  }
}
__device__ __forceinline__ floatx4 mfma_f16(const uintx4& a, const uintx4& b, const floatx4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// power of two that brings a block's largest magnitude into [2^13, 2^14) (0 for a zero / non-finite maximum)
__device__ __forceinline__ int f16_block_shift(unsigned max_bits) {
  const int ex = (int)(max_bits >> 23);
  return (ex == 0 || ex == 255) ? 0 : 13 - (ex - 127);
}

// Folded head kernel for widths whose W_r does not fit the register file (d = k = 128: 64 KB).
// One 512-thread workgroup per CU keeps the current relation's W_r in LDS, row-major with a
// 4-float pad (bank-conflict-free for both fragment shapes: the first product reads
// W[f(s,q)][16c+i] as ds_read_b32, the second W[16c'+i][16c+4q .. +3] as ds_read_b128), and its
// eight wavefronts take the 16-group tiles of the workgroup's contiguous tile range round
// robin.  A tile is 2 * (d/4) * (d/16) MFMAs (512 at d = 128, ~16k cycles), so one tile of
// row look-ahead hides the gather.
constexpr int kFoldLdsThreads = 512;

template <int D_, bool X3>
__global__ __launch_bounds__(kFoldLdsThreads) void att_fold_head_lds_kernel(
    int n_rel, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ V_tab) {
  constexpr int K_ = D_;
  constexpr int KS = D_ / 4, KT = K_ / 16, LD = K_ + 4, NW = kFoldLdsThreads / kWave;
  // X3: one image of W_r per bf16 piece, [row d][column k] with 256-byte rows whose 16-byte
  // chunks are swizzled (chunk ^ (2 (row % 8) + (row / 8) % 2)): the first product contracts over
  // W_r's row index and takes its A fragments with ds_read_b64_tr_b16 (a 16-lane group reads 4
  // rows x 16 columns and every lane receives one column: the eight rows of a 32-lane half land
  // in eight different 32-byte slots), the second contracts over the column index and reads
  // 8-byte row pieces (sixteen rows, sixteen different chunks) - both free of bank conflicts on
  // the one image.
  constexpr int S3 = D_ / 32, ROWB = D_ * 2, IMG = D_ * ROWB;
  static_assert(!X3 || D_ == 128, "image swizzle is written for 256-byte rows");
  __shared__ int32_t s_tptr[kAttMaxRelLds + 1];
  __shared__ __attribute__((aligned(16))) float s_w[X3 ? 4 : D_ * LD];
  __shared__ __attribute__((aligned(16))) unsigned char s_img[X3 ? 3 * IMG : 16];
  __shared__ __attribute__((aligned(16))) float s_rel[K_];
  const int tid = threadIdx.x;
  for (int r = tid; r < n_rel; r += kFoldLdsThreads) s_tptr[r + 1] = (gptr[r + 1] - gptr[r] + 15) >> 4;
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
  const int lane = tid % kWave, w = tid / kWave;
  const int i = lane & 15, q = lane >> 4;
  const int32_t t_begin = (int32_t)((int64_t)n_tiles * blockIdx.x / gridDim.x);
  const int32_t t_end = (int32_t)((int64_t)n_tiles * (blockIdx.x + 1) / gridDim.x);

  struct Buf { float a[KS]; };
  auto load_rows = [&](Buf& f, int32_t row) {
    const char* base = reinterpret_cast<const char*>(ent);
    const uint32_t o = (uint32_t)row * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
#pragma unroll
    for (int m = 0; m < D_ / 16; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(base + o + m * 64);
      f.a[4 * m + 0] = v.x; f.a[4 * m + 1] = v.y; f.a[4 * m + 2] = v.z; f.a[4 * m + 3] = v.w;
    }
  };

  int32_t t = t_begin;
  while (t < t_end) {  // workgroup-uniform loop over relation segments
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_tptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int r = lo;
    const int32_t rbeg = gptr[r], rend = gptr[r + 1];
    const int32_t tfirst = s_tptr[r];
    int32_t seg_end = s_tptr[r + 1];
    seg_end = seg_end < t_end ? seg_end : t_end;
    __syncthreads();  // every wave is done with the previous relation's W_r
    if (X3) {
      const float* W = W_R + (size_t)r * D_ * K_;
      for (int u = tid; u < D_ * (K_ / 8); u += kFoldLdsThreads) {
        const int row = u / (K_ / 8), ch = u % (K_ / 8);
        const float4 w0 = *reinterpret_cast<const float4*>(W + row * K_ + 8 * ch);
        const float4 w1 = *reinterpret_cast<const float4*>(W + row * K_ + 8 * ch + 4);
        const float x[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        uintx4 h, m, l;
        split_bf16x3(x, h, m, l);
        const int off = ROWB * row + 16 * (ch ^ (((row & 7) << 1) | ((row >> 3) & 1)));
        *reinterpret_cast<uintx4*>(s_img + off) = h;
        *reinterpret_cast<uintx4*>(s_img + IMG + off) = m;
        *reinterpret_cast<uintx4*>(s_img + 2 * IMG + off) = l;
      }
    } else {
      const float* W = W_R + (size_t)r * D_ * K_;
      for (int idx = tid * 4; idx < D_ * K_; idx += kFoldLdsThreads * 4) {
        const float4 v = *reinterpret_cast<const float4*>(W + idx);
        *reinterpret_cast<float4*>(s_w + (idx / K_) * LD + (idx % K_)) = v;
      }
    }
    // e_r * 2 log2(e), staged in LDS: at d = 128 holding the lane's 32 values in registers across the
    // tile pushed the bf16-piece form over the 256-register budget (36 bytes of scratch in round 2)
    for (int u = tid; u < K_; u += kFoldLdsThreads) s_rel[u] = rel[(size_t)r * K_ + u] * kTwoLog2e;
    __syncthreads();
    auto row_of_tile = [&](int32_t n) -> int32_t {
      n = n < seg_end ? n : seg_end - 1;
      int32_t g = rbeg + ((n - tfirst) << 4) + i;
      g = g < rend ? g : rend - 1;
      return g_node[g];
    };
    auto tile = [&](int32_t n, const Buf& f) {
      floatx4 acc[KT];
#pragma unroll
      for (int c = 0; c < KT; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
      floatx4 v[KT];
      if (X3) {
        // First product, P^T = W^T E^T.  Element jj of k-step s on lane group q is contraction
        // index 32s + 16 (jj / 4) + 4q + jj % 4 (the lane's own pieces of the head row); lane
        // 4qq + p of the group addresses row (32s + 16t + 4q) + qq, columns 16c + 4p .. + 3.
        // The fragments of step (s, c) are requested one step ahead of their six MFMAs.
        typedef short shortx4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) shortx4 lds_shortx4;
        const int qq = i >> 2, p = i & 3;
        const int base1 = ROWB * (4 * q + qq) + 8 * (p & 1);
        const int sw1 = ((4 * (q & 1) + qq) << 1) | (q >> 1);
        auto frag1 = [&](int n, uintx4 (&ap)[3]) {
          const int s = n / KT, c = n % KT;
          const int o = base1 + 16 * ((2 * c + (p >> 1)) ^ sw1) + ROWB * 32 * s;
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            const shortx4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(s_img + pc * IMG + o));
            const shortx4 hi =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(s_img + pc * IMG + o + ROWB * 16));
            ap[pc] = __builtin_bit_cast(uintx4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        };
        auto pieces1 = [&](int s, uintx4 (&b)[3]) {
          float x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = f.a[8 * s + jj];
          split_bf16x3(x, b[0], b[1], b[2]);
        };
        uintx4 fa[2][3], fb[2][3];
        frag1(0, fa[0]);
        pieces1(0, fb[0]);
#pragma unroll
        for (int n = 0; n < S3 * KT; ++n) {
          const int s = n / KT, c = n % KT;
          if (n + 1 < S3 * KT) frag1(n + 1, fa[(n + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (c == KT - 1 && s + 1 < S3) pieces1(s + 1, fb[(s + 1) & 1]);
          const uintx4(&ap)[3] = fa[n & 1];
          const uintx4(&bp)[3] = fb[s & 1];
          acc[c] = mfma_bf16(ap[2], bp[0], acc[c]);
          acc[c] = mfma_bf16(ap[0], bp[2], acc[c]);
          acc[c] = mfma_bf16(ap[1], bp[1], acc[c]);
          acc[c] = mfma_bf16(ap[1], bp[0], acc[c]);
          acc[c] = mfma_bf16(ap[0], bp[1], acc[c]);
          acc[c] = mfma_bf16(ap[0], bp[0], acc[c]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        const float* w1 = s_w + (4 * q) * LD + i;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const float* ws = w1 + (16 * (s >> 2) + (s & 3)) * LD;
#pragma unroll
          for (int c = 0; c < KT; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[16 * c], f.a[s], acc[c], 0, 0, 0);
        }
      }
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const float4 rv = *reinterpret_cast<const float4*>(s_rel + 16 * c + 4 * q);
        acc[c][0] = att_tanh_scaled(fmaf(acc[c][0], kTwoLog2e, rv.x));
        acc[c][1] = att_tanh_scaled(fmaf(acc[c][1], kTwoLog2e, rv.y));
        acc[c][2] = att_tanh_scaled(fmaf(acc[c][2], kTwoLog2e, rv.z));
        acc[c][3] = att_tanh_scaled(fmaf(acc[c][3], kTwoLog2e, rv.w));
      }
#pragma unroll
      for (int c2 = 0; c2 < KT; ++c2) v[c2] = (floatx4){0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_sched_barrier(0);
      if (X3) {
        // Second product, V^T = W T: row 16c2 + i of the image, contraction index
        // 32s + 16 (jj / 4) + 4q + jj % 4 = accumulator tile 2s + jj / 4, register jj % 4.
        const int base2 = ROWB * i + 8 * (q & 1);
        const int sw2 = ((i & 7) << 1) | (i >> 3);
        auto frag2 = [&](int n, uintx4 (&ap)[3]) {
          const int s = n / KT, c2 = n % KT;
          const int o0 = base2 + 16 * (((4 * s) | (q >> 1)) ^ sw2) + ROWB * 16 * c2;
          const int o1 = base2 + 16 * (((4 * s + 2) | (q >> 1)) ^ sw2) + ROWB * 16 * c2;
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            const uintx2 l2 = *reinterpret_cast<const uintx2*>(s_img + pc * IMG + o0);
            const uintx2 h2 = *reinterpret_cast<const uintx2*>(s_img + pc * IMG + o1);
            ap[pc] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3);
          }
        };
        auto pieces2 = [&](int s, uintx4 (&b)[3]) {
          float x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = acc[2 * s + (jj >> 2)][jj & 3];
          split_bf16x3(x, b[0], b[1], b[2]);
        };
        uintx4 fa[2][3], fb[2][3];
        frag2(0, fa[0]);
        pieces2(0, fb[0]);
#pragma unroll
        for (int n = 0; n < S3 * KT; ++n) {
          const int s = n / KT, c2 = n % KT;
          if (n + 1 < S3 * KT) frag2(n + 1, fa[(n + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (c2 == KT - 1 && s + 1 < S3) pieces2(s + 1, fb[(s + 1) & 1]);
          const uintx4(&ap)[3] = fa[n & 1];
          const uintx4(&bp)[3] = fb[s & 1];
          v[c2] = mfma_bf16(ap[2], bp[0], v[c2]);
          v[c2] = mfma_bf16(ap[0], bp[2], v[c2]);
          v[c2] = mfma_bf16(ap[1], bp[1], v[c2]);
          v[c2] = mfma_bf16(ap[1], bp[0], v[c2]);
          v[c2] = mfma_bf16(ap[0], bp[1], v[c2]);
          v[c2] = mfma_bf16(ap[0], bp[0], v[c2]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        const float* w2 = s_w + i * LD + 4 * q;
#pragma unroll
        for (int c = 0; c < KT; ++c)
#pragma unroll
          for (int c2 = 0; c2 < KT; ++c2) {
            const float4 wv = *reinterpret_cast<const float4*>(w2 + (16 * c2) * LD + 16 * c);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, acc[c][0], v[c2], 0, 0, 0);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, acc[c][1], v[c2], 0, 0, 0);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, acc[c][2], v[c2], 0, 0, 0);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, acc[c][3], v[c2], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      const int32_t g = rbeg + ((n - tfirst) << 4) + i;
      if (g < rend) {
#pragma unroll
        for (int c2 = 0; c2 < KT; ++c2) {
          float4 o;
          o.x = v[c2][0]; o.y = v[c2][1]; o.z = v[c2][2]; o.w = v[c2][3];
          *reinterpret_cast<float4*>(V_tab + (size_t)g * D_ + 16 * c2 + 4 * q) = o;
        }
      }
    };
    int32_t n = t + w;
    if (n < seg_end) {
      Buf b0, b1;
      int32_t rn = row_of_tile(n + NW);
      load_rows(b0, row_of_tile(n));
      while (true) {
        load_rows(b1, rn);
        rn = row_of_tile(n + 2 * NW);
        __builtin_amdgcn_sched_barrier(0);
        tile(n, b0);
        __builtin_amdgcn_sched_barrier(0);
        n += NW;
        if (n >= seg_end) break;
        load_rows(b0, rn);
        rn = row_of_tile(n + 2 * NW);
        __builtin_amdgcn_sched_barrier(0);
        tile(n, b1);
        __builtin_amdgcn_sched_barrier(0);
        n += NW;
        if (n >= seg_end) break;
      }
    }
    t = seg_end;
  }
}

template <int D_>
static int launch_att_fold_head_lds(const AttArgs& a) {
  const int cus = device_cu_count();
  if (a.f32_products)
    hipLaunchKernelGGL((att_fold_head_lds_kernel<D_, false>), dim3((unsigned)cus), dim3(kFoldLdsThreads), 0, a.st,
                       a.n_rel, a.gptr, a.g_node, a.ent, a.W_R, a.rel, a.G_tab);
  else
    hipLaunchKernelGGL((att_fold_head_lds_kernel<D_, true>), dim3((unsigned)cus), dim3(kFoldLdsThreads), 0, a.st,
                       a.n_rel, a.gptr, a.g_node, a.ent, a.W_R, a.rel, a.G_tab);
  KGAT_CHECK_LAUNCH("att_fold_head_lds");
  return KGAT_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused folded form: the V rows never leave the CU.  Work comes as tiles (kgat_fold_tiles): at
// most 16 consecutive head groups of one relation and at most `cap` grouped positions.  A wave
// computes the tile's V rows (two MFMA products, as att_fold_head_lds_kernel, W_r in LDS), parks
// them in its own 16 x d LDS patch and walks the tile's positions itself: d/4 lanes per edge,
// tail row from global memory, V row from LDS, DPP reduction, lane l ends with position pb + l.
// The first 64 positions' tail rows are requested BEFORE the MFMA phase (they arrive while it
// runs); descriptors and indices are requested one and two tiles ahead so that every load of
// the steady state is a prefetch.  Positions past the first 64 of a tile (<= cap) are walked
// with their indices requested one chunk ahead; their tail rows are not (the row buffer is 64
// registers), so tiles are capped and hub groups recompute V per `cap` positions instead.
constexpr int kFusedThreads = 512;
// kAttXcdRemap: every XCD a contiguous eighth of the workgroups' tile ranges (xcd_contiguous, kgat_common.h)
// - neighbouring ranges, which share head nodes and relation tables, behind one L2.  Stand-alone within the noise at
// d = 64 (0.172 vs 0.176 ms, round 3); in the step 0.4116 -> 0.4098 and 0.4121 -> 0.4107 ms on two boxes
// (profiles/r04_step_ab_cache_policy.txt): default since round 4 for d <= 64.  2 % slower at d = 128
// (kAtt128XcdRemap, off).
constexpr int kAttXcdRemap = 1;
constexpr int kAtt128XcdRemap = 0;
// (Round 4 also measured the packed position records - 4 bytes per edge, read once per step - and the head-group
// node ids as non-temporal loads: with round-robin tile ranges the records gained 2.5 us per step (0.4178 -> 0.4153 ms),
// with the XCD-contiguous ranges above they LOSE 1.2-1.6 us (0.4108 vs 0.4096, 0.4121 vs 0.4105:
// profiles/r04_step_ab_cache_policy.txt): plain loads.)

// X3: the two products as bf16-piece products (above); W_r's pieces sit in LDS already in fragment
// order, one 16-byte read per lane per fragment.
// A grouped position p comes as ONE packed record rec_g[p] = source node | (group slot << 28) (slot =
// the position's head group inside its 16-group block; kgat_att_pack_records): one 4-byte index load
// per position instead of three.  OUT selects where the logits go: 0 = grouped order only (logits_g,
// coalesced 256-byte stores; the softmax reads them through the inverse map), 1 = also CSR order
// (4-byte scatter through pos_g), 2 = also edge-id order (scatter through perm).
template <int D_, int OUT, bool X3>
__global__ __launch_bounds__(kFusedThreads) void att_fold_fused_kernel(
    int n_rel, int64_t n_edges, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel_tptr,
    const int4* __restrict__ tiles, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const int32_t* __restrict__ rec_g, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ pos_g, const float* __restrict__ ent, const float* __restrict__ W_R,
    const float* __restrict__ rel, float* __restrict__ logits, float* __restrict__ logits_csr,
    float* __restrict__ logits_g, const int32_t* __restrict__ part_tptr, long long* __restrict__ clocks) {
  constexpr bool LOGITS_EID = OUT == 2;
  constexpr int ROW_SHIFT = D_ == 64 ? 8 : (D_ == 32 ? 7 : 6);  // log2 of a row's bytes
  static_assert((D_ * 4) == (1 << ROW_SHIFT), "row bytes");
  constexpr int K_ = D_;
  constexpr int KS = D_ / 4, KT = K_ / 16, LD = K_ + 4, NW = kFusedThreads / kWave;
  constexpr int LPE = kFusedLanesPerEdge<D_>();    // lanes per edge in the edge phase
  constexpr int VPL = D_ / (4 * LPE);             // float4 pieces of a row per lane
  constexpr int LDV = D_ + 4;
  static_assert(D_ <= 64, "one float4 per lane per row");
  static_assert(!X3 || D_ % 32 == 0, "bf16 pieces: k-steps of 32");
  constexpr int S3 = X3 ? D_ / 32 : 1;             // k-steps of the bf16 MFMA
  constexpr int NFRAG = KT * S3 * kWave;           // fragments of one piece of one product
  __shared__ __attribute__((aligned(16))) float s_w[X3 ? 4 : D_ * LD];
  // X3: W_r's pieces as A fragments, [product][piece h,m,l][column tile][k-step][lane]
  __shared__ uintx4 s_a[X3 ? 2 * 3 * NFRAG : 1];
  __shared__ __attribute__((aligned(16))) float s_v[NW][16 * LDV];
  __shared__ int32_t s_next;  // next unclaimed tile of the current relation segment
constexpr int kAttF16Second = 1;  // default since round 4 (-3.3 us in the step at equal error; 0: three bf16 pieces on both products)
  constexpr bool F16B = X3 && kAttF16Second != 0;  // second product on fp16 pieces: W_r three, tanh values two
  __shared__ unsigned s_wmax;                             // F16B: bits of max |W_r| of the current relation
  const int tid = threadIdx.x;
  const int lane = tid % kWave, w = tid / kWave;
  const int i = lane & 15, q = lane >> 4;
  const int li = lane % LPE;

  {  // relation ids outside [0, R): logit 0
    const int64_t n_scored = rel_ptr[n_rel];
    for (int64_t p = n_scored + (int64_t)blockIdx.x * kFusedThreads + tid; p < n_edges;
         p += (int64_t)gridDim.x * kFusedThreads) {
      if (LOGITS_EID) logits[perm[p]] = 0.f;
      if (OUT >= 1 && logits_csr) logits_csr[pos_g[p]] = 0.f;
      if (logits_g) logits_g[p] = 0.f;
    }
  }
  const int32_t n_tiles = rel_tptr[n_rel];
  // the workgroup's contiguous tile range: cost balanced (kgat_fold_tile_parts) or equal counts
  const unsigned part = kAttXcdRemap ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int32_t t_begin = part_tptr ? part_tptr[part] : (int32_t)((int64_t)n_tiles * part / gridDim.x);
  const int32_t t_end = part_tptr ? part_tptr[part + 1] : (int32_t)((int64_t)n_tiles * (part + 1) / gridDim.x);
  float* vrow = s_v[w];
  // measurement aid (kgat_att_score_fused_timed_f32): when this workgroup started and ended, in 100 MHz ticks
  if (clocks && tid == 0) clocks[2 * part] = (long long)__builtin_amdgcn_s_memrealtime();

  int32_t t = t_begin;
  while (t < t_end) {  // workgroup-uniform loop over relation segments
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rel_tptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int r = lo;
    const int32_t rend = gptr[r + 1];
    int32_t seg_end = rel_tptr[r + 1];
    seg_end = seg_end < t_end ? seg_end : t_end;
    if (F16B && tid == 0) s_wmax = 0u;  // (every wave has read the previous segment's value by now)
    __syncthreads();  // every wave is done with the previous relation's W_r
    if (tid == 0) s_next = t;
    if (X3) {
      // The contraction index of a k-step is permuted the same way in A and B so that a lane's
      // eight B elements are registers it already holds: element jj of k-step s on lane group q
      // is index 16 (2s + jj/4) + 4q + jj%4 (two float4 pieces of a head row / two accumulator
      // tiles of the first product).
      const float* W = W_R + (size_t)r * D_ * K_;
      for (int f = tid; f < NFRAG; f += kFusedThreads) {
        const int fl = f % kWave, fs = (f / kWave) % S3, fc = f / (kWave * S3);
        const int fi = fl & 15, fq = fl >> 4;
        float x[8];
        uintx4 h, m, l;
        // first product, P^T = W^T E^T: A[row 16c + i][kk] = W[kk][16c + i]
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          x[jj] = W[(16 * (2 * fs + (jj >> 2)) + 4 * fq + (jj & 3)) * K_ + 16 * fc + fi];
        split_bf16x3(x, h, m, l);
        s_a[0 * NFRAG + f] = h; s_a[1 * NFRAG + f] = m; s_a[2 * NFRAG + f] = l;
        // second product, V^T = W T: A[row 16c2 + i][kk] = W[16c2 + i][kk]
        const float4 w0 = *reinterpret_cast<const float4*>(W + (16 * fc + fi) * K_ + 32 * fs + 4 * fq);
        const float4 w1 = *reinterpret_cast<const float4*>(W + (16 * fc + fi) * K_ + 32 * fs + 16 + 4 * fq);
        x[0] = w0.x; x[1] = w0.y; x[2] = w0.z; x[3] = w0.w;
        x[4] = w1.x; x[5] = w1.y; x[6] = w1.z; x[7] = w1.w;
        if (F16B) {  // the fp16 images need the relation's scale first: second pass below
          float mx = 0.f;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) mx = fmaxf(mx, fabsf(x[jj]));
          atomicMax(&s_wmax, __float_as_uint(mx));
        } else {
          split_bf16x3(x, h, m, l);
          s_a[3 * NFRAG + f] = h; s_a[4 * NFRAG + f] = m; s_a[5 * NFRAG + f] = l;
        }
      }
    } else {
      const float* W = W_R + (size_t)r * D_ * K_;
      for (int idx = tid * 4; idx < D_ * K_; idx += kFusedThreads * 4) {
        const float4 v = *reinterpret_cast<const float4*>(W + idx);
        *reinterpret_cast<float4*>(s_w + (idx / K_) * LD + (idx % K_)) = v;
      }
    }
    __syncthreads();
    int w_shift = 0;  // F16B: the second product runs on W_r * 2^w_shift; the logits are scaled back
    if (F16B) {
      w_shift = __builtin_amdgcn_readfirstlane(f16_block_shift(s_wmax));
      const float* W = W_R + (size_t)r * D_ * K_;
      for (int f = tid; f < NFRAG; f += kFusedThreads) {
        const int fl = f % kWave, fs = (f / kWave) % S3, fc = f / (kWave * S3);
        const int fi = fl & 15, fq = fl >> 4;
        const float4 w0 = *reinterpret_cast<const float4*>(W + (16 * fc + fi) * K_ + 32 * fs + 4 * fq);
        const float4 w1 = *reinterpret_cast<const float4*>(W + (16 * fc + fi) * K_ + 32 * fs + 16 + 4 * fq);
        const float x[8] = {ldexpf(w0.x, w_shift), ldexpf(w0.y, w_shift), ldexpf(w0.z, w_shift), ldexpf(w0.w, w_shift),
                            ldexpf(w1.x, w_shift), ldexpf(w1.y, w_shift), ldexpf(w1.z, w_shift), ldexpf(w1.w, w_shift)};
        uintx4 h, m, l;
        split_f16x3(x, h, m, l);
        s_a[3 * NFRAG + f] = h; s_a[4 * NFRAG + f] = m; s_a[5 * NFRAG + f] = l;
      }
      __syncthreads();
    }
    float relv[KT][4];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(rel + (size_t)r * K_ + 16 * c + 4 * q);
      relv[c][0] = v.x * kTwoLog2e; relv[c][1] = v.y * kTwoLog2e;
      relv[c][2] = v.z * kTwoLog2e; relv[c][3] = v.w * kTwoLog2e;
    }

    // descriptor of the wave's k-th tile of this segment (clamped: past the end it repeats the
    // last one, whose loads are then harmless duplicates)
    auto desc_of = [&](int32_t n) -> int4 {
      n = n < seg_end ? n : seg_end - 1;
      return tiles[n];
    };
    struct CIdx { int32_t row_off, lg, oe, op; };
    auto head_idx = [&](const int4& d) -> int32_t {
      int32_t g = d.y + i;
      g = g < rend ? g : rend - 1;
      return g_node[g];
    };
    auto chunk_idx = [&](const int4& d, int32_t p0) -> CIdx {
      int32_t p = p0 + lane;
      p = p < d.w ? p : d.w - 1;
      CIdx c;
      const uint32_t rec = (uint32_t)rec_g[p];
      // byte offset of the source row: N * d * 4 < 4 GiB (checked by the caller), so the node id ends
      // below bit 32 - ROW_SHIFT and the shift drops exactly the slot bits
      c.row_off = (int32_t)(rec << ROW_SHIFT);
      c.lg = (int32_t)(rec >> 28);
      c.oe = LOGITS_EID ? perm[p] : 0;
      c.op = (OUT >= 1 && logits_csr) ? pos_g[p] : 0;
      return c;
    };
    struct HBuf { float a[KS]; };
    auto load_head = [&](HBuf& f, int32_t row) {
      const char* base = reinterpret_cast<const char*>(ent);
      const uint32_t o = (uint32_t)row * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
#pragma unroll
      for (int m = 0; m < D_ / 16; ++m) {
        const float4 v = *reinterpret_cast<const float4*>(base + o + m * 64);
        f.a[4 * m + 0] = v.x; f.a[4 * m + 1] = v.y; f.a[4 * m + 2] = v.z; f.a[4 * m + 3] = v.w;
      }
    };
    struct EBuf { float4 r[LPE][VPL]; };
    const char* eb = reinterpret_cast<const char*>(ent) + li * 16;
    auto load_edges = [&](EBuf& e, const CIdx& c) {
#pragma unroll
      for (int s = 0; s < LPE; ++s) {
        const uint32_t eo = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - li + s) << 2, c.row_off);
#pragma unroll
        for (int v = 0; v < VPL; ++v) e.r[s][v] = *reinterpret_cast<const float4*>(eb + eo + v * (LPE * 16));
      }
    };
    auto mfma_phase = [&](const HBuf& f) {
      floatx4 acc[KT];
#pragma unroll
      for (int c = 0; c < KT; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
      floatx4 v[KT];
#pragma unroll
      for (int c2 = 0; c2 < KT; ++c2) v[c2] = (floatx4){0.f, 0.f, 0.f, 0.f};
      if (X3) {
        const uintx4* fa = s_a + lane;
#pragma unroll
        for (int s = 0; s < S3; ++s) {
          float x[8];
          uintx4 bh, bm, bl;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = f.a[8 * s + jj];
          split_bf16x3(x, bh, bm, bl);
#pragma unroll
          for (int c = 0; c < KT; ++c) {  // smallest piece products first
            const uintx4 ah = fa[0 * NFRAG + (c * S3 + s) * kWave];
            const uintx4 am = fa[1 * NFRAG + (c * S3 + s) * kWave];
            const uintx4 al = fa[2 * NFRAG + (c * S3 + s) * kWave];
            acc[c] = mfma_bf16(al, bh, acc[c]);
            acc[c] = mfma_bf16(ah, bl, acc[c]);
            acc[c] = mfma_bf16(am, bm, acc[c]);
            acc[c] = mfma_bf16(am, bh, acc[c]);
            acc[c] = mfma_bf16(ah, bm, acc[c]);
            acc[c] = mfma_bf16(ah, bh, acc[c]);
          }
        }
      } else {
        const float* w1 = s_w + (4 * q) * LD + i;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const float* ws = w1 + (16 * (s >> 2) + (s & 3)) * LD;
#pragma unroll
          for (int c = 0; c < KT; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[16 * c], f.a[s], acc[c], 0, 0, 0);
        }
      }
#pragma unroll
      for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // F16B: the tanh values leave as T * 2^14 (the same rounding, scaled: a power of two), so that the low
          // piece of a small |T| is still a normal fp16 number (unscaled, |T| ~ 1e-3 keeps 14 bits, not 22)
          const float y = fmaf(acc[c][j], kTwoLog2e, relv[c][j]);
          acc[c][j] = F16B ? fmaf(-32768.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y) + 1.0f), 16384.0f)
                           : att_tanh_scaled(y);
        }
      if (X3) {
        const uintx4* fa = s_a + 3 * NFRAG + lane;
#pragma unroll
        for (int s = 0; s < S3; ++s) {
          float x[8];
          uintx4 bh, bm, bl;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = acc[2 * s + (jj >> 2)][jj & 3];
          if (F16B) {
            split_f16x2(x, bh, bm);  // |tanh| * 2^14 < 16,384: inside fp16's range
#pragma unroll
            for (int c2 = 0; c2 < KT; ++c2) {  // five piece products, smallest first
              const uintx4 ah = fa[0 * NFRAG + (c2 * S3 + s) * kWave];
              const uintx4 am = fa[1 * NFRAG + (c2 * S3 + s) * kWave];
              const uintx4 al = fa[2 * NFRAG + (c2 * S3 + s) * kWave];
              v[c2] = mfma_f16(al, bh, v[c2]);
              v[c2] = mfma_f16(am, bm, v[c2]);
              v[c2] = mfma_f16(am, bh, v[c2]);
              v[c2] = mfma_f16(ah, bm, v[c2]);
              v[c2] = mfma_f16(ah, bh, v[c2]);
            }
            continue;
          }
          split_bf16x3(x, bh, bm, bl);
#pragma unroll
          for (int c2 = 0; c2 < KT; ++c2) {
            const uintx4 ah = fa[0 * NFRAG + (c2 * S3 + s) * kWave];
            const uintx4 am = fa[1 * NFRAG + (c2 * S3 + s) * kWave];
            const uintx4 al = fa[2 * NFRAG + (c2 * S3 + s) * kWave];
            v[c2] = mfma_bf16(al, bh, v[c2]);
            v[c2] = mfma_bf16(ah, bl, v[c2]);
            v[c2] = mfma_bf16(am, bm, v[c2]);
            v[c2] = mfma_bf16(am, bh, v[c2]);
            v[c2] = mfma_bf16(ah, bm, v[c2]);
            v[c2] = mfma_bf16(ah, bh, v[c2]);
          }
        }
      } else {
        const float* w2 = s_w + i * LD + 4 * q;
#pragma unroll
        for (int c = 0; c < KT; ++c)
#pragma unroll
          for (int c2 = 0; c2 < KT; ++c2) {
            const float4 wv = *reinterpret_cast<const float4*>(w2 + (16 * c2) * LD + 16 * c);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, acc[c][0], v[c2], 0, 0, 0);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, acc[c][1], v[c2], 0, 0, 0);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, acc[c][2], v[c2], 0, 0, 0);
            v[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, acc[c][3], v[c2], 0, 0, 0);
          }
      }
      // v[c2][j] = V[group i][16c2 + 4q + j] -> the wave's LDS patch, row = group
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous tile's reads are done
#pragma unroll
      for (int c2 = 0; c2 < KT; ++c2) {
        float4 o;
        o.x = v[c2][0]; o.y = v[c2][1]; o.z = v[c2][2]; o.w = v[c2][3];
        *reinterpret_cast<float4*>(vrow + i * LDV + 16 * c2 + 4 * q) = o;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    auto edge_phase = [&](const EBuf& e, const CIdx& c, int32_t p0, int32_t pe) {
      float mine = 0.f;
#pragma unroll
      for (int s = 0; s < LPE; ++s) {
        const int32_t lg = __builtin_amdgcn_ds_bpermute((lane - li + s) << 2, c.lg);
        float d = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const float4 b = *reinterpret_cast<const float4*>(vrow + lg * LDV + 4 * li + v * (LPE * 4));
          d = v == 0 ? e.r[s][v].x * b.x : fmaf(e.r[s][v].x, b.x, d);
          d = fmaf(e.r[s][v].y, b.y, d);
          d = fmaf(e.r[s][v].z, b.z, d);
          d = fmaf(e.r[s][v].w, b.w, d);
        }
        d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xF, 0xF, true));
        if (LPE >= 4) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xF, 0xF, true));
        if (LPE >= 8) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xF, 0xF, true));
        if (LPE >= 16) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x140, 0xF, 0xF, true));
        mine = li == s ? d : mine;
      }
      if (F16B) mine = ldexpf(mine, -w_shift - 14);  // (V rows are those of (W_r * 2^w_shift) (T * 2^14))
      if (p0 + lane < pe) {
        if (LOGITS_EID) logits[c.oe] = mine;
        if (OUT >= 1 && logits_csr) logits_csr[c.op] = mine;
        if (logits_g) logits_g[p0 + lane] = mine;
      }
    };

    // Waves claim tiles one at a time from the segment (LDS counter), three tiles before they
    // compute them - the look-ahead of the descriptor/index prefetch - so a wave that meets
    // heavy tiles simply claims fewer.
    auto claim = [&]() -> int32_t {
      int32_t g = 0;
      if (lane == 0) g = atomicAdd(&s_next, 1);
      return __builtin_amdgcn_readfirstlane(g);
    };
    int32_t n = claim();
    if (n < seg_end) {
      int32_t n1 = claim(), n2 = claim();
      // prologue: descriptors of the wave's next three tiles; head row index of two; chunk indices of one
      int4 d0 = desc_of(n), d1 = desc_of(n1), d2 = desc_of(n2);
      int32_t h0 = head_idx(d0), h1 = head_idx(d1);
      CIdx c0 = chunk_idx(d0, d0.z);
      HBuf hb0, hb1;
      load_head(hb0, h0);
#define KGAT_FUSED_STEP(HCUR, HNEXT)                                                   \
      {                                                                                \
        const int32_t n3 = claim();                                                    \
        const int4 d3 = desc_of(n3);                                                   \
        const int32_t h2 = head_idx(d2);                                               \
        const CIdx c1 = chunk_idx(d1, d1.z);                                           \
        /* indices of the second to fourth chunk (a tile of the default cap has at most */ \
        /* four): requested before the MFMA phase, which hides their trip from HBM - in */ \
        /* the step the index arrays were last read a whole pass ago                    */ \
        CIdx cx = chunk_idx(d0, d0.z + kWave);                                         \
        const CIdx cy = chunk_idx(d0, d0.z + 2 * kWave);                               \
        const CIdx cz = chunk_idx(d0, d0.z + 3 * kWave);                               \
        EBuf eb0;                                                                      \
        load_edges(eb0, c0);                                                           \
        load_head(HNEXT, h1);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                             \
        mfma_phase(HCUR);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        edge_phase(eb0, c0, d0.z, d0.w);                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (d0.z + kWave < d0.w) {                                                     \
          load_edges(eb0, cx);                                                         \
          edge_phase(eb0, cx, d0.z + kWave, d0.w);                                     \
          if (d0.z + 2 * kWave < d0.w) {                                               \
            load_edges(eb0, cy);                                                       \
            cx = chunk_idx(d0, d0.z + 4 * kWave);  /* fifth chunk (caps above 256) */  \
            edge_phase(eb0, cy, d0.z + 2 * kWave, d0.w);                               \
            if (d0.z + 3 * kWave < d0.w) {                                             \
              load_edges(eb0, cz);                                                     \
              edge_phase(eb0, cz, d0.z + 3 * kWave, d0.w);                             \
              for (int32_t p0 = d0.z + 4 * kWave; p0 < d0.w; p0 += kWave) {            \
                load_edges(eb0, cx);                                                   \
                const CIdx cn = chunk_idx(d0, p0 + kWave);                             \
                edge_phase(eb0, cx, p0, d0.w);                                         \
                cx = cn;                                                               \
              }                                                                        \
            }                                                                          \
          }                                                                            \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        d0 = d1; d1 = d2; d2 = d3;                                                     \
        h1 = h2;                                                                       \
        c0 = c1;                                                                       \
        n = n1; n1 = n2; n2 = n3;                                                      \
      }
      while (true) {
        KGAT_FUSED_STEP(hb0, hb1)
        if (n >= seg_end) break;
        KGAT_FUSED_STEP(hb1, hb0)
        if (n >= seg_end) break;
      }
#undef KGAT_FUSED_STEP
    }
    t = seg_end;
  }
  if (clocks) {
    __syncthreads();
    if (tid == 0) clocks[2 * part + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

// ---------------------------------------------------------------------------------------------
// Fused folded form on 32-GROUP tiles (d = k = 64, round 5; KGAT_ATT_TILES32, opt-in).  The same algorithm as
// att_fold_fused_kernel - per tile the V rows of its head groups from two chained products at fp32 accuracy, parked
// in the wavefront's LDS patch, then a gather-dot over the tile's positions - re-cut to test what rounds 3-4 took for
// that kernel's bound, the instructions a SIMD has to issue per head group (478 vector instructions against 88 MFMAs
// per 16-group tile, the matrix pipe 40 % busy):
//  * v_mfma_f32_32x32x16_f16 on 32 groups: half as many matrix instructions per group, half as many fragment reads
//    of W_r's pieces per group;
//  * BOTH products on fp16 pieces: W_r 2^shift as three (exact), the head rows as two after a power-of-two scale PER
//    ROW that brings the row's largest magnitude to [2^13, 2^14) (the scale is folded into the converts -
//    v_fma_mixlo/mixhi_f16 - and leaves through the per-lane factor of the tanh argument's fma: lane (group, half)
//    holds only its own group's columns), the tanh values 2^14 as two: five piece products per product (3 x 2, the
//    2^-33 one dropped), cuts of 5 / 4 vector instructions per pair of values instead of 11;
//  * twice the positions per tile: fewer partly filled 64-position chunks (69.7 k on the benchmark graph against
//    89.0 k), half as many tile descriptors, claims and waits per group.
// An fp32 operand as h + l with fp16 pieces carries 22 bits: |x s - h - l| <= 2^-23 |x s|, one more rounding of the
// operand at fp32's own level, of either sign - what the 16-group kernel's second product has done to the tanh values
// since round 4; error against fp64 no larger than the fp32-MFMA form's (test_att_fused_product_forms, which runs this
// kernel too; test_att_fused32_tiles_and_logits: rows over 70 orders of magnitude, tiny and zero rows).
// MEASURED (profiles/r05_att32_experiments.txt): 22.5 M vector + 2.5 M matrix instructions per launch against
// 30.2 M + 5.6 M - and the SAME run time, 126-132 us in the step against 125-133.  So the instruction count is not
// what bounds either kernel; nor is the gather (every row from a 1,024-row cache-resident slice: -5 us), nor the
// residency (12 wavefronts per CU without the cross-phase row prefetch: the same), nor the split over workgroups
// (max / mean of their end times 1.14, finer static splits slower).  What does bound both kernels was found by ablation
// afterwards (profiles/r05_att_bounds.txt): with BOTH products and the tanh removed the launch takes
// 124.9 of 127.9 us, with the edge phases removed 86.3 - the launch is its gather-dot side (the tail row of every edge
// against the group's V row), which runs at 1.36-1.39 x the bare gather of the same rows; the products hide behind it.
// (The "478 vector instructions per tile / pipe 40 % busy => issue-bound" reading of rounds 3-4, quoted above as the
// hypothesis this kernel was built to test, is refuted by this kernel's own timing.)  Not the default: equal speed does
// not pay for a second arithmetic statement on the path.
constexpr int kFused32Threads = 512;
constexpr int kF32Groups = 32;
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ floatx16 mfma32_f16(const uintx4& a, const uintx4& b, const floatx16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (a s, b s) as a packed fp16 pair, round to nearest even; s a power of two: the product is exact, one rounding
__device__ __forceinline__ unsigned cvt_scaled_pk_f16(float a, float b, float s) {
  unsigned r;
  asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
      : "=&v"(r) : "v"(a), "v"(b), "v"(s));
  return r;
}
__device__ __forceinline__ float rem_scaled_lo(float x, float s, unsigned hh) {  // x s - float(low half of hh)
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(hh));
  return r;
}
__device__ __forceinline__ float rem_scaled_hi(float x, float s, unsigned hh) {  // x s - float(high half of hh)
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(hh));
  return r;
}
__device__ __forceinline__ void split_scaled_f16x2(const float (&x)[8], float s, uintx4& h, uintx4& m) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned hh = cvt_scaled_pk_f16(x[2 * t], x[2 * t + 1], s);
    h[t] = hh;
    m[t] = cvt_pk_f16(rem_scaled_lo(x[2 * t], s, hh), rem_scaled_hi(x[2 * t + 1], s, hh));
  }
}

template <int OUT>
__global__ __launch_bounds__(kFused32Threads) void att_fold_fused32_kernel(
    int n_rel, int64_t n_edges, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel_tptr,
    const int4* __restrict__ tiles, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const int32_t* __restrict__ rec_g, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ pos_g, const float* __restrict__ ent, const float* __restrict__ W_R,
    const float* __restrict__ rel, float* __restrict__ logits, float* __restrict__ logits_csr,
    float* __restrict__ logits_g, const int32_t* __restrict__ part_tptr) {
  constexpr int D_ = 64, K_ = 64;
  constexpr bool LOGITS_EID = OUT == 2;
  constexpr int ROW_SHIFT = 8;                         // log2 of a row's bytes
  constexpr int NW = kFused32Threads / kWave;
  constexpr int LPE = 4, VPL = 4;                      // lanes per edge, float4 pieces of a row per lane
  constexpr int LDV = D_ + 4;
  constexpr int NFRAG = 2 * 4 * kWave;                 // fragments of one piece of one product: [row tile][k-step][lane]
  // W_r 2^shift as fp16 pieces in A-fragment order, [product][piece h,m,l][row tile][k-step][lane]
  __shared__ uintx4 s_a[2 * 3 * NFRAG];
  __shared__ __attribute__((aligned(16))) float s_v[NW][kF32Groups * LDV];
  __shared__ __attribute__((aligned(16))) float s_rel[K_];   // e_r 2 log2(e)
  __shared__ int32_t s_next;                                  // next unclaimed tile of the current relation segment
  __shared__ unsigned s_wmax;                                 // bits of max |W_r| of the current relation
  const int tid = threadIdx.x;
  const int lane = tid % kWave, w = tid / kWave;
  const int gr = lane & 31, hf = lane >> 5;                   // MFMA column (head group of the tile), k half
  const int li = lane % LPE;

  {  // relation ids outside [0, R): logit 0
    const int64_t n_scored = rel_ptr[n_rel];
    for (int64_t p = n_scored + (int64_t)blockIdx.x * kFused32Threads + tid; p < n_edges;
         p += (int64_t)gridDim.x * kFused32Threads) {
      if (LOGITS_EID) logits[perm[p]] = 0.f;
      if (OUT >= 1 && logits_csr) logits_csr[pos_g[p]] = 0.f;
      if (logits_g) logits_g[p] = 0.f;
    }
  }
  const int32_t n_tiles = rel_tptr[n_rel];
  const unsigned part = xcd_contiguous(blockIdx.x, gridDim.x);
  const int32_t t_begin = part_tptr ? part_tptr[part] : (int32_t)((int64_t)n_tiles * part / gridDim.x);
  const int32_t t_end = part_tptr ? part_tptr[part + 1] : (int32_t)((int64_t)n_tiles * (part + 1) / gridDim.x);
  float* vrow = s_v[w];

  int32_t t = t_begin;
  while (t < t_end) {  // workgroup-uniform loop over relation segments
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rel_tptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int r = lo;
    const int32_t rend = gptr[r + 1];
    int32_t seg_end = rel_tptr[r + 1];
    seg_end = seg_end < t_end ? seg_end : t_end;
    if (tid == 0) s_wmax = 0u;  // (every wave has read the previous segment's value by now)
    __syncthreads();            // every wave is done with the previous relation's W_r
    if (tid == 0) s_next = t;
    const float* W = W_R + (size_t)r * D_ * K_;
    {
      float mx = 0.f;
      for (int idx = tid * 4; idx < D_ * K_; idx += kFused32Threads * 4) {
        const float4 v = *reinterpret_cast<const float4*>(W + idx);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
      atomicMax(&s_wmax, __float_as_uint(mx));
      if (tid < K_) s_rel[tid] = rel[(size_t)r * K_ + tid] * kTwoLog2e;
    }
    __syncthreads();
    const int w_shift = __builtin_amdgcn_readfirstlane(f16_block_shift(s_wmax));
    for (int f = tid; f < 2 * NFRAG; f += kFused32Threads) {
      const int prod = f / NFRAG, g = f % NFRAG;
      const int fl = g % kWave, fs = (g / kWave) % 4, ft = g / (kWave * 4);
      const int fr = fl & 31, fh = fl >> 5;
      float x[8];
      if (prod == 0) {
        // first product, G^T = W^T E^T: A[row 32 ft + fr][k = 16 fs + 8 fh + jj] = W[k][row]
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) x[jj] = ldexpf(W[(16 * fs + 8 * fh + jj) * K_ + 32 * ft + fr], w_shift);
      } else {
        // second product, V^T = W T: the contraction index of k-step fs = 2 tt + u, slot (fh, jj) is the G column the
        // lane's accumulator register 8 u + jj of row tile tt holds: 32 tt + 8 (2 u + jj / 4) + 4 fh + jj % 4
        const int tt = fs >> 1, u = fs & 1;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          x[jj] = ldexpf(W[(32 * ft + fr) * K_ + 32 * tt + 8 * (2 * u + (jj >> 2)) + 4 * fh + (jj & 3)], w_shift);
      }
      uintx4 h, m, l;
      split_f16x3(x, h, m, l);
      s_a[(prod * 3 + 0) * NFRAG + g] = h; s_a[(prod * 3 + 1) * NFRAG + g] = m; s_a[(prod * 3 + 2) * NFRAG + g] = l;
    }
    __syncthreads();

    auto desc_of = [&](int32_t n) -> int4 {
      n = n < seg_end ? n : seg_end - 1;
      return tiles[n];
    };
    struct CIdx { int32_t row_off, lg, oe, op; };
    auto head_idx = [&](const int4& d) -> int32_t {
      int32_t g = d.y + gr;
      g = g < rend ? g : rend - 1;
      return g_node[g];
    };
    auto chunk_idx = [&](const int4& d, int32_t p0) -> CIdx {
      int32_t p = p0 + lane;
      p = p < d.w ? p : d.w - 1;
      CIdx c;
      const uint32_t rec = (uint32_t)rec_g[p];
      c.row_off = (int32_t)(rec << ROW_SHIFT);   // (node ids below 2^24: the shift drops the five slot bits)
      c.lg = (int32_t)(rec >> 27);
      c.oe = LOGITS_EID ? perm[p] : 0;
      c.op = (OUT >= 1 && logits_csr) ? pos_g[p] : 0;
      return c;
    };
    struct HBuf { float a[32]; };
    // lane (group gr, half hf) takes its group's head row elements 16 s + 8 hf .. + 7 of every k-step s
    auto load_head = [&](HBuf& f, int32_t row) {
      const char* base = reinterpret_cast<const char*>(ent);
      const uint32_t o = (uint32_t)row * (uint32_t)(D_ * 4) + (uint32_t)(hf * 32);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 v0 = *reinterpret_cast<const float4*>(base + o + s * 64);
        const float4 v1 = *reinterpret_cast<const float4*>(base + o + s * 64 + 16);
        f.a[8 * s + 0] = v0.x; f.a[8 * s + 1] = v0.y; f.a[8 * s + 2] = v0.z; f.a[8 * s + 3] = v0.w;
        f.a[8 * s + 4] = v1.x; f.a[8 * s + 5] = v1.y; f.a[8 * s + 6] = v1.z; f.a[8 * s + 7] = v1.w;
      }
    };
    struct EBuf { float4 r[LPE][VPL]; };
    const char* eb = reinterpret_cast<const char*>(ent) + li * 16;
    auto load_edges = [&](EBuf& e, const CIdx& c) {
#pragma unroll
      for (int s = 0; s < LPE; ++s) {
        const uint32_t eo = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - li + s) << 2, c.row_off);
#pragma unroll
        for (int v = 0; v < VPL; ++v) e.r[s][v] = *reinterpret_cast<const float4*>(eb + eo + v * (LPE * 16));
      }
    };
    // V rows of the tile's 32 groups into the wave's patch; `f` is free again after the first product: the next
    // tile's head rows are requested into it there and arrive during the rest of the step
    auto mfma_phase = [&](HBuf& f, int32_t next_row) {
      float mx = fabsf(f.a[0]);
#pragma unroll
      for (int i = 1; i < 32; ++i) mx = fmaxf(mx, fabsf(f.a[i]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));            // the other half of the row
      const int ex = (int)(__float_as_uint(mx) >> 23);   // (mx >= 0)
      int sh = (ex == 0 || ex == 255) ? 0 : 140 - ex;     // brings the row's largest magnitude into [2^13, 2^14)
      sh = sh > 126 ? 126 : sh;
      const float sc = __uint_as_float((unsigned)(sh + 127) << 23);
      const float cl = ldexpf(kTwoLog2e, -sh - w_shift);  // G = G' 2^-(sh + w_shift); tanh argument scale 2 log2(e)
      floatx16 acc[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tt][i] = 0.f;
      {
        const uintx4* fa = s_a + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float x[8];
          uintx4 bh, bl;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = f.a[8 * s + jj];
          split_scaled_f16x2(x, sc, bh, bl);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {  // five piece products, smallest first
            const uintx4 ah = fa[0 * NFRAG + (tt * 4 + s) * kWave];
            const uintx4 am = fa[1 * NFRAG + (tt * 4 + s) * kWave];
            const uintx4 al = fa[2 * NFRAG + (tt * 4 + s) * kWave];
            acc[tt] = mfma32_f16(al, bh, acc[tt]);
            acc[tt] = mfma32_f16(am, bl, acc[tt]);
            acc[tt] = mfma32_f16(am, bh, acc[tt]);
            acc[tt] = mfma32_f16(ah, bl, acc[tt]);
            acc[tt] = mfma32_f16(ah, bh, acc[tt]);
          }
        }
      }
      load_head(f, next_row);
      // acc[tt][4 q + i] = G'[group gr][column 32 tt + 8 q + 4 hf + i]: tanh, scaled by 2^14 for the fp16 cut
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 rv = *reinterpret_cast<const float4*>(s_rel + 32 * tt + 8 * q + 4 * hf);
          const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float y = fmaf(acc[tt][4 * q + i], cl, rr[i]);
            acc[tt][4 * q + i] = fmaf(-32768.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y) + 1.0f), 16384.0f);
          }
        }
      floatx16 v[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[tt][i] = 0.f;
      {
        const uintx4* fa = s_a + 3 * NFRAG + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // k-step s = 2 tt + u: registers 8 u .. 8 u + 7 of row tile tt
          float x[8];
          uintx4 bh, bl;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = acc[s >> 1][8 * (s & 1) + jj];
          split_f16x2(x, bh, bl);        // |tanh| 2^14 < 16,384: inside fp16's range
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const uintx4 ah = fa[0 * NFRAG + (t2 * 4 + s) * kWave];
            const uintx4 am = fa[1 * NFRAG + (t2 * 4 + s) * kWave];
            const uintx4 al = fa[2 * NFRAG + (t2 * 4 + s) * kWave];
            v[t2] = mfma32_f16(al, bh, v[t2]);
            v[t2] = mfma32_f16(am, bl, v[t2]);
            v[t2] = mfma32_f16(am, bh, v[t2]);
            v[t2] = mfma32_f16(ah, bl, v[t2]);
            v[t2] = mfma32_f16(ah, bh, v[t2]);
          }
        }
      }
      // v[t2][4 q + i] = V'[group gr][32 t2 + 8 q + 4 hf + i] -> the wave's LDS patch, row = group
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous tile's reads are done
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 o;
          o.x = v[t2][4 * q + 0]; o.y = v[t2][4 * q + 1]; o.z = v[t2][4 * q + 2]; o.w = v[t2][4 * q + 3];
          *reinterpret_cast<float4*>(vrow + gr * LDV + 32 * t2 + 8 * q + 4 * hf) = o;
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    auto edge_phase = [&](const EBuf& e, const CIdx& c, int32_t p0, int32_t pe) {
      float mine = 0.f;
#pragma unroll
      for (int s = 0; s < LPE; ++s) {
        const int32_t lg = __builtin_amdgcn_ds_bpermute((lane - li + s) << 2, c.lg);
        float d = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const float4 b = *reinterpret_cast<const float4*>(vrow + lg * LDV + 4 * li + v * (LPE * 4));
          d = v == 0 ? e.r[s][v].x * b.x : fmaf(e.r[s][v].x, b.x, d);
          d = fmaf(e.r[s][v].y, b.y, d);
          d = fmaf(e.r[s][v].z, b.z, d);
          d = fmaf(e.r[s][v].w, b.w, d);
        }
        d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xF, 0xF, true));
        d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xF, 0xF, true));
        mine = li == s ? d : mine;
      }
      mine = ldexpf(mine, -w_shift - 14);  // (V rows are those of (W_r 2^w_shift) (T 2^14))
      if (p0 + lane < pe) {
        if (LOGITS_EID) logits[c.oe] = mine;
        if (OUT >= 1 && logits_csr) logits_csr[c.op] = mine;
        if (logits_g) logits_g[p0 + lane] = mine;
      }
    };

    auto claim = [&]() -> int32_t {
      int32_t g = 0;
      if (lane == 0) g = atomicAdd(&s_next, 1);
      return __builtin_amdgcn_readfirstlane(g);
    };
    int32_t n = claim();
    if (n < seg_end) {
      int32_t n1 = claim(), n2 = claim();
      int4 d0 = desc_of(n), d1 = desc_of(n1), d2 = desc_of(n2);
      int32_t h1 = head_idx(d1);
      CIdx c0 = chunk_idx(d0, d0.z);
      HBuf hb;
      load_head(hb, head_idx(d0));
      while (true) {
        const int32_t n3 = claim();
        const int4 d3 = desc_of(n3);
        const int32_t h2 = head_idx(d2);
        const CIdx c1 = chunk_idx(d1, d1.z);
        // indices of the second to fourth chunk: requested before the MFMA phase, which hides their trip from HBM
        CIdx cx = chunk_idx(d0, d0.z + kWave);
        const CIdx cy = chunk_idx(d0, d0.z + 2 * kWave);
        const CIdx cz = chunk_idx(d0, d0.z + 3 * kWave);
        EBuf eb0;
        load_edges(eb0, c0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(hb, h1);
        __builtin_amdgcn_sched_barrier(0);
        edge_phase(eb0, c0, d0.z, d0.w);
        __builtin_amdgcn_sched_barrier(0);
        if (d0.z + kWave < d0.w) {
          load_edges(eb0, cx);
          edge_phase(eb0, cx, d0.z + kWave, d0.w);
          if (d0.z + 2 * kWave < d0.w) {
            load_edges(eb0, cy);
            cx = chunk_idx(d0, d0.z + 4 * kWave);
            edge_phase(eb0, cy, d0.z + 2 * kWave, d0.w);
            if (d0.z + 3 * kWave < d0.w) {
              load_edges(eb0, cz);
              edge_phase(eb0, cz, d0.z + 3 * kWave, d0.w);
              for (int32_t p0 = d0.z + 4 * kWave; p0 < d0.w; p0 += kWave) {
                load_edges(eb0, cx);
                const CIdx cn = chunk_idx(d0, p0 + kWave);
                edge_phase(eb0, cx, p0, d0.w);
                cx = cn;
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        d0 = d1; d1 = d2; d2 = d3;
        h1 = h2;
        c0 = c1;
        n = n1; n1 = n2; n2 = n3;
        if (n >= seg_end) break;
      }
    }
    t = seg_end;
  }
}

template <int OUT>
static void launch_att_fold_fused32_form(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  const unsigned grid = a.part_tptr ? a.grid : (unsigned)device_cu_count();
  hipLaunchKernelGGL((att_fold_fused32_kernel<OUT>), dim3(grid), dim3(kFused32Threads), 0, a.st, a.n_rel, a.n_edges,
                     a.rel_ptr, rel_tptr, reinterpret_cast<const int4*>(tiles), a.gptr, a.g_node, a.rec_g, a.perm,
                     a.pos_g, a.ent, a.W_R, a.rel, a.logits, a.logits_csr, a.logits_g, a.part_tptr);
}

int launch_att_fold_fused32(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  if (a.logits) launch_att_fold_fused32_form<2>(a, rel_tptr, tiles);
  else if (a.logits_csr) launch_att_fold_fused32_form<1>(a, rel_tptr, tiles);
  else launch_att_fold_fused32_form<0>(a, rel_tptr, tiles);
  KGAT_CHECK_LAUNCH("att_fold_fused32");
  return KGAT_OK;
}

template <int D_, int OUT, bool X3>
static void launch_att_fold_fused_form(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  const unsigned grid = a.part_tptr ? a.grid : (unsigned)device_cu_count();
  hipLaunchKernelGGL((att_fold_fused_kernel<D_, OUT, X3>), dim3(grid), dim3(kFusedThreads), 0, a.st, a.n_rel,
                     a.n_edges, a.rel_ptr, rel_tptr, reinterpret_cast<const int4*>(tiles), a.gptr, a.g_node, a.rec_g,
                     a.perm, a.pos_g, a.ent, a.W_R, a.rel, a.logits, a.logits_csr, a.logits_g, a.part_tptr, a.part_clocks);
}

template <int D_, bool X3>
static void launch_att_fold_fused_out(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  if (a.logits) launch_att_fold_fused_form<D_, 2, X3>(a, rel_tptr, tiles);
  else if (a.logits_csr) launch_att_fold_fused_form<D_, 1, X3>(a, rel_tptr, tiles);
  else launch_att_fold_fused_form<D_, 0, X3>(a, rel_tptr, tiles);
}


template <int D_>
static int launch_att_fold_fused(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  constexpr bool kCanSplit = D_ % 32 == 0;
  if (kCanSplit && !a.f32_products) launch_att_fold_fused_out<D_, kCanSplit>(a, rel_tptr, tiles);
  else launch_att_fold_fused_out<D_, false>(a, rel_tptr, tiles);
  KGAT_CHECK_LAUNCH("att_fold_fused");
  return KGAT_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused folded form at d = k = 128 (round 3; products re-cut in round 5): what att_fold_fused_kernel is at d <= 64.
// One 512-thread workgroup per CU; per relation segment W_r 2^shift is cut into TWO fp16 piece images in LDS (64 KB;
// one row-major image per piece serves both products: the first contracts over the row index and takes its A
// fragments with ds_read_b64_tr_b16, att_fold_head_lds_kernel's layout); a wavefront takes the tiles t + w,
// t + w + 8, ... of the workgroup's segment, computes the 16 V rows of a tile (two chained piece products), parks
// them in its own 16 x 128 LDS patch (8 KB) and walks the tile's positions itself: 8 lanes per edge, tail row from
// global memory (512 B), V row from the patch, lane l ends with position p0 + l, logits stored in grouped order
// (coalesced).  No V table: the two-launch form wrote n_groups x 512 B and read it back per edge (0.5 GB each way on
// the benchmark graph).  The register file is full during the MFMA phase, so a chunk's rows (64 positions = 128
// registers) are gathered after it, not ahead of it as at d <= 64; the two waves of a SIMD run out of phase, one in
// its MFMA phase while the other waits for rows.
// PRODUCTS (round 5): every operand as h + l with fp16 pieces after a power-of-two scale (W_r per relation, the head
// rows per row - folded into the converts, v_fma_mixlo/mixhi_f16 -, the tanh values by 2^14), and THREE of the four
// piece products, l h, h l, h h, in the fp32 accumulator of v_mfma_f32_16x16x32_f16: 192 MFMAs per tile against the
// 384 of three bf16 pieces x six products (rounds 3-4), cuts of 5 / 4 vector instructions per pair of values instead
// of 11.  A two-piece operand carries 22 bits (|x s - h - l| <= 2^-23 |x s|) and the dropped l l is <= 2^-22 of its
// term: up to ~2^-21 of a TERM, where the exact fp32 fma chain has 2^-24 of a partial SUM per step - smaller than the
// chain's error on d-term sums (measured, test_att_fused_product_forms[128], every case incl. rows over 70 orders of
// magnitude: max error 0.76-0.97 x and mean error 0.74-0.95 x the fp32 MFMA form's; the bar of the test is unchanged).
// At this width the matrix pipe was the longest of the kernel's resources: 0.427 -> 0.382 ms on the amazon-book graph
// (profiles/r05_att_bounds.txt).  At d = 64 the same cut was measured on the 32-group kernel and buys nothing
// (126.5 vs 126.4 us: that launch is not bound by its products, below) while its rows with one dominant term came
// out at 2.2 x the fp32 form's error, over the bar: d <= 64 keeps W_r exact in three pieces.
constexpr int kFused128Threads = 512;

template <int OUT>
__global__ __launch_bounds__(kFused128Threads) void att_fold_fused128_kernel(
    int n_rel, int64_t n_edges, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel_tptr,
    const int4* __restrict__ tiles, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const int32_t* __restrict__ rec_g, const int32_t* __restrict__ perm, const int32_t* __restrict__ pos_g,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel,
    float* __restrict__ logits, float* __restrict__ logits_csr, float* __restrict__ logits_g,
    const int32_t* __restrict__ part_tptr, long long* __restrict__ clocks) {
  constexpr int D_ = 128, K_ = 128, KS = D_ / 4, KT = K_ / 16, NW = kFused128Threads / kWave;
  constexpr int S3 = D_ / 32, ROWB = D_ * 2, IMG = D_ * ROWB;
  constexpr int LPE = 8, VPL = D_ / (4 * LPE);
constexpr int kF128Passes = 2;
  constexpr int NPASS = kF128Passes, HALF = LPE / NPASS;  // row-gather passes per 64-position chunk
  constexpr bool LOGITS_EID = OUT == 2;
  constexpr int ROW_SHIFT = 9;  // 512-byte rows
  __shared__ __attribute__((aligned(16))) unsigned char s_img[2 * IMG];   // W_r 2^shift: fp16 pieces h, l
  __shared__ __attribute__((aligned(16))) float s_v[NW][16 * D_];
  __shared__ __attribute__((aligned(16))) float s_rel[K_];                  // e_r 2 log2(e)
  __shared__ unsigned s_wmax;                                               // bits of max |W_r| of the current relation
  const int tid = threadIdx.x;
  const int lane = tid % kWave, w = tid / kWave;
  const int i = lane & 15, q = lane >> 4;
  const int li = lane % LPE;

  {  // relation ids outside [0, R): logit 0
    const int64_t n_scored = rel_ptr[n_rel];
    for (int64_t p = n_scored + (int64_t)blockIdx.x * kFused128Threads + tid; p < n_edges;
         p += (int64_t)gridDim.x * kFused128Threads) {
      if (LOGITS_EID) logits[perm[p]] = 0.f;
      if (OUT >= 1 && logits_csr) logits_csr[pos_g[p]] = 0.f;
      if (logits_g) logits_g[p] = 0.f;
    }
  }
  const int32_t n_tiles = rel_tptr[n_rel];
  const unsigned part = kAtt128XcdRemap ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int32_t t_begin = part_tptr ? part_tptr[part] : (int32_t)((int64_t)n_tiles * part / gridDim.x);
  const int32_t t_end = part_tptr ? part_tptr[part + 1] : (int32_t)((int64_t)n_tiles * (part + 1) / gridDim.x);
  float* vrow = s_v[w];
  if (clocks && tid == 0) clocks[2 * part] = (long long)__builtin_amdgcn_s_memrealtime();   // (measurement aid)

  struct HBuf { float a[KS]; };
  auto load_head = [&](HBuf& f, int32_t row) {
    const char* base = reinterpret_cast<const char*>(ent);
    const uint32_t o = (uint32_t)row * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
#pragma unroll
    for (int m = 0; m < D_ / 16; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(base + o + m * 64);
      f.a[4 * m + 0] = v.x; f.a[4 * m + 1] = v.y; f.a[4 * m + 2] = v.z; f.a[4 * m + 3] = v.w;
    }
  };

  int32_t t = t_begin;
  while (t < t_end) {  // workgroup-uniform loop over relation segments
    int lo = 0, hi = n_rel;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rel_tptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int r = lo;
    const int32_t rend = gptr[r + 1];
    int32_t seg_end = rel_tptr[r + 1];
    seg_end = seg_end < t_end ? seg_end : t_end;
    if (tid == 0) s_wmax = 0u;  // (every wave has read the previous segment's value by now)
    __syncthreads();            // every wave is done with the previous relation's images
    const float* W = W_R + (size_t)r * D_ * K_;
    {
      float mx = 0.f;
      for (int idx = tid * 4; idx < D_ * K_; idx += kFused128Threads * 4) {
        const float4 v = *reinterpret_cast<const float4*>(W + idx);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
      atomicMax(&s_wmax, __float_as_uint(mx));
      if (tid < K_) s_rel[tid] = rel[(size_t)r * K_ + tid] * kTwoLog2e;
    }
    __syncthreads();
    const int w_shift = __builtin_amdgcn_readfirstlane(f16_block_shift(s_wmax));
    {
      int u0 = tid;
      asm volatile("" : "+v"(u0));  // recompute the per-thread offsets per segment: hoisted out of the segment loop they were kept (spilled) across the whole tile loop
      for (int u = u0; u < D_ * (K_ / 8); u += kFused128Threads) {
        const int row = u / (K_ / 8), ch = u % (K_ / 8);
        const float4 w0 = *reinterpret_cast<const float4*>(W + row * K_ + 8 * ch);
        const float4 w1 = *reinterpret_cast<const float4*>(W + row * K_ + 8 * ch + 4);
        const float x[8] = {ldexpf(w0.x, w_shift), ldexpf(w0.y, w_shift), ldexpf(w0.z, w_shift), ldexpf(w0.w, w_shift),
                            ldexpf(w1.x, w_shift), ldexpf(w1.y, w_shift), ldexpf(w1.z, w_shift), ldexpf(w1.w, w_shift)};
        uintx4 h, l;
        split_f16x2(x, h, l);
        const int off = ROWB * row + 16 * (ch ^ (((row & 7) << 1) | ((row >> 3) & 1)));
        *reinterpret_cast<uintx4*>(s_img + off) = h;
        *reinterpret_cast<uintx4*>(s_img + IMG + off) = l;
      }
    }
    __syncthreads();

    auto desc_of = [&](int32_t n) -> int4 {  // wave-uniform: kept in scalar registers
      n = n < seg_end ? n : seg_end - 1;
      const int4 d = tiles[n];
      return make_int4(__builtin_amdgcn_readfirstlane(d.x), __builtin_amdgcn_readfirstlane(d.y),
                       __builtin_amdgcn_readfirstlane(d.z), __builtin_amdgcn_readfirstlane(d.w));
    };
    auto head_idx = [&](const int4& d) -> int32_t {
      int32_t g = d.y + i;
      g = g < rend ? g : rend - 1;
      return g_node[g];
    };
    struct CIdx { int32_t row_off, lg, oe, op; };
    auto chunk_idx = [&](const int4& d, int32_t p0) -> CIdx {
      int32_t p = p0 + lane;
      p = p < d.w ? p : d.w - 1;
      CIdx c;
      const uint32_t rec = (uint32_t)rec_g[p];
      c.row_off = (int32_t)(rec << ROW_SHIFT);
      c.lg = (int32_t)(rec >> 28);
      c.oe = LOGITS_EID ? perm[p] : 0;
      c.op = (OUT >= 1 && logits_csr) ? pos_g[p] : 0;
      return c;
    };

    // the two chained products of one tile: V rows of the tile's 16 groups -> the wave's LDS patch
    auto mfma_phase = [&](const HBuf& f) {
      // power-of-two scale of the lane's head row (group i): the row's largest magnitude into [2^13, 2^14)
      float mx = fabsf(f.a[0]);
#pragma unroll
      for (int e = 1; e < KS; ++e) mx = fmaxf(mx, fabsf(f.a[e]));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const int ex = (int)(__float_as_uint(mx) >> 23);   // (mx >= 0)
      int sh = (ex == 0 || ex == 255) ? 0 : 140 - ex;
      sh = sh > 126 ? 126 : sh;
      const float sc = __uint_as_float((unsigned)(sh + 127) << 23);
      const float cl = ldexpf(kTwoLog2e, -sh - w_shift);  // G = G' 2^-(sh + w_shift); tanh argument scale 2 log2(e)
      floatx4 acc[KT];
#pragma unroll
      for (int c = 0; c < KT; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
      floatx4 v[KT];
      {
        typedef short shortx4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) shortx4 lds_shortx4;
        const int qq = i >> 2, p = i & 3;
        const int base1 = ROWB * (4 * q + qq) + 8 * (p & 1);
        const int sw1 = ((4 * (q & 1) + qq) << 1) | (q >> 1);
        auto frag1 = [&](int n, uintx4 (&ap)[2]) {
          const int s = n / KT, c = n % KT;
          const int o = base1 + 16 * ((2 * c + (p >> 1)) ^ sw1) + ROWB * 32 * s;
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            const shortx4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(s_img + pc * IMG + o));
            const shortx4 hi4 =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(s_img + pc * IMG + o + ROWB * 16));
            ap[pc] = __builtin_bit_cast(uintx4, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        };
        auto pieces1 = [&](int s, uintx4 (&b)[2]) {
          float x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = f.a[8 * s + jj];
          split_scaled_f16x2(x, sc, b[0], b[1]);
        };
        uintx4 fa[2][2], fb[2][2];
        frag1(0, fa[0]);
        pieces1(0, fb[0]);
#pragma unroll
        for (int n = 0; n < S3 * KT; ++n) {
          const int s = n / KT, c = n % KT;
          if (n + 1 < S3 * KT) frag1(n + 1, fa[(n + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (c == KT - 1 && s + 1 < S3) pieces1(s + 1, fb[(s + 1) & 1]);
          const uintx4(&ap)[2] = fa[n & 1];
          const uintx4(&bp)[2] = fb[s & 1];
          acc[c] = mfma_f16(ap[1], bp[0], acc[c]);   // three piece products (l h, h l, h h; l l < 2^-22 dropped)
          acc[c] = mfma_f16(ap[0], bp[1], acc[c]);
          acc[c] = mfma_f16(ap[0], bp[0], acc[c]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // tanh(G + e_r), leaving as T 2^14 for the fp16 cut (a small |T| keeps 22 bits in two pieces)
#pragma unroll
      for (int c = 0; c < KT; ++c) {
        const float4 rv = *reinterpret_cast<const float4*>(s_rel + 16 * c + 4 * q);
        const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y = fmaf(acc[c][j], cl, rr[j]);
          acc[c][j] = fmaf(-32768.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y) + 1.0f), 16384.0f);
        }
      }
#pragma unroll
      for (int c2 = 0; c2 < KT; ++c2) v[c2] = (floatx4){0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_sched_barrier(0);
      {
        const int base2 = ROWB * i + 8 * (q & 1);
        const int sw2 = ((i & 7) << 1) | (i >> 3);
        auto frag2 = [&](int n, uintx4 (&ap)[2]) {
          const int s = n / KT, c2 = n % KT;
          const int o0 = base2 + 16 * (((4 * s) | (q >> 1)) ^ sw2) + ROWB * 16 * c2;
          const int o1 = base2 + 16 * (((4 * s + 2) | (q >> 1)) ^ sw2) + ROWB * 16 * c2;
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            const uintx2 l2 = *reinterpret_cast<const uintx2*>(s_img + pc * IMG + o0);
            const uintx2 h2 = *reinterpret_cast<const uintx2*>(s_img + pc * IMG + o1);
            ap[pc] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3);
          }
        };
        auto pieces2 = [&](int s, uintx4 (&b)[2]) {
          float x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = acc[2 * s + (jj >> 2)][jj & 3];
          split_f16x2(x, b[0], b[1]);   // |T| 2^14 < 16,384: inside fp16's range
        };
        uintx4 fa[2][2], fb[2][2];
        frag2(0, fa[0]);
        pieces2(0, fb[0]);
#pragma unroll
        for (int n = 0; n < S3 * KT; ++n) {
          const int s = n / KT, c2 = n % KT;
          if (n + 1 < S3 * KT) frag2(n + 1, fa[(n + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (c2 == KT - 1 && s + 1 < S3) pieces2(s + 1, fb[(s + 1) & 1]);
          const uintx4(&ap)[2] = fa[n & 1];
          const uintx4(&bp)[2] = fb[s & 1];
          v[c2] = mfma_f16(ap[1], bp[0], v[c2]);
          v[c2] = mfma_f16(ap[0], bp[1], v[c2]);
          v[c2] = mfma_f16(ap[0], bp[0], v[c2]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // v[c2][j] = V[group i][16c2 + 4q + j] -> the wave's patch, row = group slot; the 16-byte chunks
      // of a row are swizzled by the slot's parity (chunk ^ 8) so that the edge phase's reads of two
      // different slots by one 16-lane group fall into different bank halves
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous tile's reads are done
#pragma unroll
      for (int c2 = 0; c2 < KT; ++c2) {
        float4 o;
        o.x = v[c2][0]; o.y = v[c2][1]; o.z = v[c2][2]; o.w = v[c2][3];
        *reinterpret_cast<float4*>(vrow + i * D_ + 4 * ((4 * c2 + q) ^ ((i & 1) << 3))) = o;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };

    // 64 positions starting at p0, their rows gathered in NPASS passes (1: all 8 steps of the 8-lane
    // groups at once, 128 registers - free during the edge phase, the MFMA phase's are dead by then)
    const char* eb = reinterpret_cast<const char*>(ent) + li * 16;
    auto edge_chunk = [&](const CIdx& c, int32_t p0, int32_t pe) {
      float mine = 0.f;
#pragma unroll
      for (int h = 0; h < NPASS; ++h) {
        float4 rowv[HALF][VPL];
#pragma unroll
        for (int s = 0; s < HALF; ++s) {
          const uint32_t eo = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - li + h * HALF + s) << 2, c.row_off);
#pragma unroll
          for (int v = 0; v < VPL; ++v) rowv[s][v] = *reinterpret_cast<const float4*>(eb + eo + v * (LPE * 16));
        }
#pragma unroll
        for (int s = 0; s < HALF; ++s) {
          const int32_t lg = __builtin_amdgcn_ds_bpermute((lane - li + h * HALF + s) << 2, c.lg);
          const float* vr = vrow + lg * D_;
          const int sw = (lg & 1) << 3;
          float d = 0.f;
#pragma unroll
          for (int v = 0; v < VPL; ++v) {
            const float4 b = *reinterpret_cast<const float4*>(vr + 4 * ((li + LPE * v) ^ sw));
            d = v == 0 ? rowv[s][v].x * b.x : fmaf(rowv[s][v].x, b.x, d);
            d = fmaf(rowv[s][v].y, b.y, d);
            d = fmaf(rowv[s][v].z, b.z, d);
            d = fmaf(rowv[s][v].w, b.w, d);
          }
          d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xF, 0xF, true));
          d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xF, 0xF, true));
          d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xF, 0xF, true));
          mine = li == h * HALF + s ? d : mine;
        }
      }
      mine = ldexpf(mine, -w_shift - 14);  // (V rows are those of (W_r 2^w_shift) (T 2^14))
      if (p0 + lane < pe) {
        if (LOGITS_EID) logits[c.oe] = mine;
        if (OUT >= 1 && logits_csr) logits_csr[c.op] = mine;
        if (logits_g) logits_g[p0 + lane] = mine;
      }
    };

    int32_t n = t + w;
    if (n < seg_end) {
      // One head-row buffer: the next tile's rows, descriptor and first-chunk records are requested right
      // AFTER the current tile's MFMA phase (which is done with the buffer by then) and arrive during its
      // edge phase; nothing is in flight across an MFMA phase, whose registers are all its own.
      int4 d0 = desc_of(n), d1 = desc_of(n + NW);
      HBuf hb;
      load_head(hb, head_idx(d0));
      int32_t h1 = head_idx(d1);
      CIdx c0 = chunk_idx(d0, d0.z);
      while (true) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase(hb);
        __builtin_amdgcn_sched_barrier(0);
        const int4 d2 = desc_of(n + 2 * NW);
        const CIdx c1 = chunk_idx(d1, d1.z);
        load_head(hb, h1);
        __builtin_amdgcn_sched_barrier(0);
        edge_chunk(c0, d0.z, d0.w);
        for (int32_t p0 = d0.z + kWave; p0 < d0.w; p0 += kWave) {
          const CIdx cx = chunk_idx(d0, p0);
          edge_chunk(cx, p0, d0.w);
        }
        __builtin_amdgcn_sched_barrier(0);
        h1 = head_idx(d2);
        d0 = d1; d1 = d2;
        c0 = c1;
        n += NW;
        if (n >= seg_end) break;
      }
    }
    t = seg_end;
  }
  if (clocks) {
    __syncthreads();
    if (tid == 0) clocks[2 * part + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

// Round 3's wave-role experiments (producer / consumer waves; measured, not adopted - notes in the file)

template <int OUT>
static void launch_att_fold_fused128_form(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  const unsigned grid = a.part_tptr ? a.grid : (unsigned)device_cu_count();
  hipLaunchKernelGGL((att_fold_fused128_kernel<OUT>), dim3(grid), dim3(kFused128Threads), 0, a.st, a.n_rel, a.n_edges,
                     a.rel_ptr, rel_tptr, reinterpret_cast<const int4*>(tiles), a.gptr, a.g_node, a.rec_g, a.perm,
                     a.pos_g, a.ent, a.W_R, a.rel, a.logits, a.logits_csr, a.logits_g, a.part_tptr, a.part_clocks);
}

static int launch_att_fold_fused128(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  if (a.logits) launch_att_fold_fused128_form<2>(a, rel_tptr, tiles);
  else if (a.logits_csr) launch_att_fold_fused128_form<1>(a, rel_tptr, tiles);
  else launch_att_fold_fused128_form<0>(a, rel_tptr, tiles);
  KGAT_CHECK_LAUNCH("att_fold_fused128");
  return KGAT_OK;
}


int launch_att_fold_fused_any(int d, const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  switch (d) {
    case 16: return launch_att_fold_fused<16>(a, rel_tptr, tiles);
    case 32: return launch_att_fold_fused<32>(a, rel_tptr, tiles);
    case 64: return launch_att_fold_fused<64>(a, rel_tptr, tiles);
    case 128: return a.f32_products ? KGAT_E_UNSUPPORTED : launch_att_fold_fused128(a, rel_tptr, tiles);
    default: return KGAT_E_UNSUPPORTED;
  }
}

int launch_att_fold_head_any(int d, const AttArgs& a) {
  switch (d) {
    case 16: return launch_att_fold_head<16>(a);
    case 32: return launch_att_fold_head<32>(a);
    case 64: return launch_att_fold_head<64>(a);
    case 128: return launch_att_fold_head_lds<128>(a);
    default: return KGAT_E_UNSUPPORTED;
  }
}

int launch_att_split_any(int d, const AttArgs& a) {
  switch (d) {
    case 16: return launch_att_split_d<16>(a);
    case 32: return launch_att_split_d<32>(a);
    case 64: return launch_att_split_d<64>(a);
    default: return KGAT_E_UNSUPPORTED;
  }
}

int launch_att_persistent_any(int d, const AttArgs& a) {
  switch (d) {
    case 16: return launch_att_persistent<16, 0>(a);
    case 32: return launch_att_persistent<32, 0>(a);
    case 64: return launch_att_persistent<64, 0>(a);
    default: return KGAT_E_UNSUPPORTED;
  }
}

}  // namespace kgat
