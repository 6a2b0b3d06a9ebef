// Persistent-wavefront attention-logit kernel (d == k <= 64); design notes in kgat_att.hip.
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


int launch_att_persistent_any(int d, bool acc, const AttArgs& a) {
  switch (d) {
    case 16: return acc ? launch_att_persistent<16, 1>(a) : launch_att_persistent<16, 0>(a);
    case 32: return acc ? launch_att_persistent<32, 1>(a) : launch_att_persistent<32, 0>(a);
    case 64: return acc ? launch_att_persistent<64, 1>(a) : launch_att_persistent<64, 0>(a);
    default: return KGAT_E_UNSUPPORTED;
  }
}

}  // namespace kgat
