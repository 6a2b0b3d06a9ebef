// Wave-role forms of the two fused attention kernels: EXPERIMENTS of round 3, not part of the shipped
// library.  Included by kgat_att_persistent.hip only under -DKGAT_ATT_WAVE_ROLES (scripts/micro/
// att_variants_ab.py, att_ws_phases.py); uses that file's helpers (split_bf16x3, mfma_bf16, the stamp macros).
// Kept as the record of what was measured - NOTEBOOK.md 3.2, profiles/r03_att_wave_roles.txt.
#pragma once
// (included inside namespace kgat)

// ---------------------------------------------------------------------------------------------
// Fused folded form with wave roles: an experiment of round 3, NOT the shipped kernel (compiled only with
// -DKGAT_ATT_WAVE_ROLES; scripts/micro/att_variants_ab.py, att_ws_phases.py).  Measured on the amazon-book
// graph, d = 64: stand-alone 140 us against 155 us for att_fold_fused_kernel, inside the step 141-142 us
// against 140-141 us - no gain where it counts.  What the stamps and the ablations say:
//   * consumers alone (producers publish zeros at once): 112 us with 8 or with 12 consumer waves - the
//     tail-row gather at the fabric's ~10.6 TB/s, the same floor the aggregation sits on;
//   * producers alone: 132 us with 8 (two per SIMD), 164 us with 4; per tile 1,528 + 1,410 cycles for the two
//     products (48 MFMAs of 16 cycles each: the two producers of a SIMD share its matrix pipe), ~1,000 for
//     the tanh and the cut between them, ~950 waiting for the consumer to free a slot;
//   * with sixteen busy waves per CU the shader clock (delta s_memtime over the launch) falls to ~1.5 GHz,
//     against ~2.1 GHz for the two-waves-per-SIMD kernel: a third fewer cycles, each a third longer.
// Kept as the record of that result.
// Design notes as written for the experiment:
// Fused folded form with wave roles (round 3, second half; d = k = 64, bf16-piece products, grouped-order
// logits).  att_fold_fused_kernel above runs every wave through all of a tile's phases - issue the loads,
// two chained MFMA products, the tile's edges - in 231 VGPRs, i.e. two waves per SIMD, and its stamps put a
// wave at 7.7 k cycles per tile against ~1.6 k of matrix-pipe or vector-issue time: the waves wait on each
// other's dependency chains, and two of them per SIMD cannot cover that.  Here a 1,024-thread workgroup
// holds sixteen waves, four per SIMD, in two roles of at most 128 VGPRs each:
//   * PRODUCER j (waves 0..7): tiles j, j+8, ... of the relation segment - head rows (requested a tile
//     ahead), the two products, V rows into slot (count & 1) of pair j's two 16 x d LDS patches;
//   * CONSUMER j (waves 8..15): the same tiles' edges - records two tiles ahead, the first chunk's tail
//     rows one tile ahead (requested as soon as the previous tile's first chunk has left the row buffer),
//     V rows from the slot.
// A pair talks through two LDS words per slot: full = count + 1 once the V rows are written, free =
// count + 1 once the consumer has read them; the count runs over the whole launch, so the words only
// ever grow.  Producers never wait for consumers of other pairs, consumers never for other producers;
// the only workgroup barriers are the two around a relation's W_r staging, as before.  Polls are
// bounded (a lost wake-up would then show as wrong logits in the tests, not as a hung GPU).
#ifndef KGAT_WS_PRODUCERS
#define KGAT_WS_PRODUCERS 8
#endif
#ifndef KGAT_WS_ABLATE
#define KGAT_WS_ABLATE 0  // diagnostic builds: 1 = producers publish at once and compute nothing, 2 = consumers read nothing
#endif
constexpr int kWsThreads = 1024;
constexpr int kWsNP = KGAT_WS_PRODUCERS, kWsNC = kWsThreads / kWave - kWsNP;  // producers (waves 0..NP-1), consumers
static_assert(kWsNC % kWsNP == 0, "a producer serves a whole number of consumers");
constexpr int kWsSpinMax = 1 << 22;

__device__ __forceinline__ void ws_wait_ge(const int32_t* flag, int32_t want) {
  for (int spin = 0; spin < kWsSpinMax; ++spin) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Tile n of a relation segment starting at tile t0: consumer (n - t0) % NC takes it as its ((n - t0) / NC)-th
// tile of the segment, from slot (count & 1) of its two V patches; producer (n - t0) % NP - the same number for
// all tiles of a consumer, NP dividing NC - computes it.  Counts and flag words restart with every segment
// (between the two barriers of the W_r staging).
template <int D_>
__global__ __launch_bounds__(kWsThreads) void att_fold_ws_kernel(
    int n_rel, int64_t n_edges, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel_tptr,
    const int4* __restrict__ tiles, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const int32_t* __restrict__ rec_g, const float* __restrict__ ent, const float* __restrict__ W_R,
    const float* __restrict__ rel, float* __restrict__ logits_g, const int32_t* __restrict__ part_tptr) {
  constexpr int ROW_SHIFT = D_ == 64 ? 8 : 7;
  static_assert(D_ == 64 || D_ == 32, "bf16 pieces: k-steps of 32; one float4 per lane per row");
  constexpr int K_ = D_, KT = K_ / 16;
  constexpr int NP = kWsNP, NC = kWsNC;
  constexpr int LPE = kFusedLanesPerEdge<D_>(), VPL = D_ / (4 * LPE), LDV = D_ + 4;
  constexpr int S3 = D_ / 32, NFRAG = KT * S3 * kWave;
  __shared__ uintx4 s_a[2 * 3 * NFRAG];  // W_r's pieces as A fragments, [product][piece h,m,l][column tile][k-step][lane]
  __shared__ __attribute__((aligned(16))) float s_v[NC][2][16 * LDV];
  __shared__ __attribute__((aligned(16))) float s_rel[K_];  // e_r * 2 log2(e)
  __shared__ int32_t s_full[NC][2], s_free[NC][2];
  const int tid = threadIdx.x;
  const int lane = tid % kWave;
  const int w = __builtin_amdgcn_readfirstlane(tid / kWave);
  const bool producer = w < NP;
  const int i = lane & 15, q = lane >> 4;
  const int li = lane % LPE;

  {  // relation ids outside [0, R): logit 0
    const int64_t n_scored = rel_ptr[n_rel];
    for (int64_t p = n_scored + (int64_t)blockIdx.x * kWsThreads + tid; p < n_edges; p += (int64_t)gridDim.x * kWsThreads)
      logits_g[p] = 0.f;
  }
  const int32_t n_tiles = rel_tptr[n_rel];
  const int32_t t_begin = part_tptr ? part_tptr[blockIdx.x] : (int32_t)((int64_t)n_tiles * blockIdx.x / gridDim.x);
  const int32_t t_end = part_tptr ? part_tptr[blockIdx.x + 1] : (int32_t)((int64_t)n_tiles * (blockIdx.x + 1) / gridDim.x);

#ifdef KGAT_ATT_STAMPS
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
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
    __syncthreads();  // every wave is done with the previous segment: its W_r, its patches, its flag words
    {
      // fragment images of W_r's pieces (see att_fold_fused_kernel): the first product's by threads
      // 0 .. NFRAG-1, the second product's by the next NFRAG
      const float* W = W_R + (size_t)r * D_ * K_;
      for (int f2 = tid; f2 < 2 * NFRAG; f2 += kWsThreads) {
        const int f = f2 % NFRAG;
        const int fl = f % kWave, fs = (f / kWave) % S3, fc = f / (kWave * S3);
        const int fi = fl & 15, fq = fl >> 4;
        float x[8];
        uintx4 h, m, l;
        if (f2 < NFRAG) {  // P^T = W^T E^T: A[row 16c + i][kk] = W[kk][16c + i]
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = W[(16 * (2 * fs + (jj >> 2)) + 4 * fq + (jj & 3)) * K_ + 16 * fc + fi];
          split_bf16x3(x, h, m, l);
          s_a[0 * NFRAG + f] = h; s_a[1 * NFRAG + f] = m; s_a[2 * NFRAG + f] = l;
        } else {           // V^T = W T: A[row 16c2 + i][kk] = W[16c2 + i][kk]
          const float4 w0 = *reinterpret_cast<const float4*>(W + (16 * fc + fi) * K_ + 32 * fs + 4 * fq);
          const float4 w1 = *reinterpret_cast<const float4*>(W + (16 * fc + fi) * K_ + 32 * fs + 16 + 4 * fq);
          x[0] = w0.x; x[1] = w0.y; x[2] = w0.z; x[3] = w0.w;
          x[4] = w1.x; x[5] = w1.y; x[6] = w1.z; x[7] = w1.w;
          split_bf16x3(x, h, m, l);
          s_a[3 * NFRAG + f] = h; s_a[4 * NFRAG + f] = m; s_a[5 * NFRAG + f] = l;
        }
      }
      if (tid < K_) s_rel[tid] = rel[(size_t)r * K_ + tid] * kTwoLog2e;
      if (tid >= kWsThreads - 2 * NC) {
        const int f = kWsThreads - 1 - tid;
        s_full[f >> 1][f & 1] = 0;
        s_free[f >> 1][f & 1] = 0;
      }
    }
    __syncthreads();

    auto desc_of = [&](int32_t n) -> int4 {  // (clamped: past the segment's end the last tile again, harmless duplicates)
      n = n < seg_end ? n : seg_end - 1;
      return tiles[n];
    };
    if (producer) {
      // ------------------------------------------------------------------ producer
      int32_t n = t + w;
      auto head_idx = [&](const int4& d) -> int32_t {
        int32_t g = d.y + i;
        g = g < rend ? g : rend - 1;
        return g_node[g];
      };
      struct HBuf { float a[D_ / 4]; };
      auto load_head = [&](HBuf& f, int32_t row) {
        const char* base = reinterpret_cast<const char*>(ent);
        const uint32_t o = (uint32_t)row * (uint32_t)(D_ * 4) + (uint32_t)(q * 16);
#pragma unroll
        for (int m = 0; m < D_ / 16; ++m) {
          const float4 v = *reinterpret_cast<const float4*>(base + o + m * 64);
          f.a[4 * m + 0] = v.x; f.a[4 * m + 1] = v.y; f.a[4 * m + 2] = v.z; f.a[4 * m + 3] = v.w;
        }
      };
      auto tile_v = [&](const HBuf& f, int32_t cons, int32_t c) {
        float* vrow = s_v[cons][c & 1];
        floatx4 acc[KT], v[KT];
#pragma unroll
        for (int c1 = 0; c1 < KT; ++c1) {
          acc[c1] = (floatx4){0.f, 0.f, 0.f, 0.f};
          v[c1] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }
#if KGAT_WS_ABLATE != 1
        // The A fragments of step n = (k-step s, column tile c) are requested one step ahead of their six
        // MFMAs, across the two products (at 128 registers the compiler, left alone, put every read right
        // in front of its MFMAs: an LDS round trip sixteen times per tile).
        KGAT_ATT_PHASE_T(pa);
        auto frag = [&](int prod, int n2, uintx4 (&ap)[3]) {
          const uintx4* fa = s_a + prod * 3 * NFRAG + ((n2 % KT) * S3 + n2 / KT) * kWave + lane;
          ap[0] = fa[0]; ap[1] = fa[NFRAG]; ap[2] = fa[2 * NFRAG];  // h, m, l
        };
        uintx4 fa[2][3], fb[2][3];
        frag(0, 0, fa[0]);
        {
          float x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = f.a[jj];
          split_bf16x3(x, fb[0][0], fb[0][1], fb[0][2]);
        }
        __builtin_amdgcn_sched_barrier(0);
        KGAT_ATT_PHASE_T(pb);
#pragma unroll
        for (int n2 = 0; n2 < S3 * KT; ++n2) {
          const int s = n2 / KT, c1 = n2 % KT;
          if (n2 + 1 < S3 * KT) frag(0, n2 + 1, fa[(n2 + 1) & 1]);
          else frag(1, 0, fa[(n2 + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (c1 == KT - 1 && s + 1 < S3) {
            float x[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) x[jj] = f.a[8 * (s + 1) + jj];
            split_bf16x3(x, fb[(s + 1) & 1][0], fb[(s + 1) & 1][1], fb[(s + 1) & 1][2]);
          }
          const uintx4(&ap)[3] = fa[n2 & 1];
          const uintx4(&bp)[3] = fb[s & 1];
          acc[c1] = mfma_bf16(ap[2], bp[0], acc[c1]);  // smallest piece products first
          acc[c1] = mfma_bf16(ap[0], bp[2], acc[c1]);
          acc[c1] = mfma_bf16(ap[1], bp[1], acc[c1]);
          acc[c1] = mfma_bf16(ap[1], bp[0], acc[c1]);
          acc[c1] = mfma_bf16(ap[0], bp[1], acc[c1]);
          acc[c1] = mfma_bf16(ap[0], bp[0], acc[c1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        KGAT_ATT_PHASE_T(pc);
#pragma unroll
        for (int c1 = 0; c1 < KT; ++c1) {  // (the arithmetic of att_fold_fused_kernel: the same bits)
          const float4 e = *reinterpret_cast<const float4*>(s_rel + 16 * c1 + 4 * q);
          acc[c1][0] = att_tanh_scaled(fmaf(acc[c1][0], kTwoLog2e, e.x));
          acc[c1][1] = att_tanh_scaled(fmaf(acc[c1][1], kTwoLog2e, e.y));
          acc[c1][2] = att_tanh_scaled(fmaf(acc[c1][2], kTwoLog2e, e.z));
          acc[c1][3] = att_tanh_scaled(fmaf(acc[c1][3], kTwoLog2e, e.w));
        }
        {
          float x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = acc[jj >> 2][jj & 3];
          split_bf16x3(x, fb[0][0], fb[0][1], fb[0][2]);
        }
        __builtin_amdgcn_sched_barrier(0);
        KGAT_ATT_PHASE_T(pd);
#pragma unroll
        for (int n2 = 0; n2 < S3 * KT; ++n2) {
          const int s = n2 / KT, c2 = n2 % KT;
          if (n2 + 1 < S3 * KT) frag(1, n2 + 1, fa[(n2 + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (c2 == KT - 1 && s + 1 < S3) {
            float x[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) x[jj] = acc[2 * (s + 1) + (jj >> 2)][jj & 3];
            split_bf16x3(x, fb[(s + 1) & 1][0], fb[(s + 1) & 1][1], fb[(s + 1) & 1][2]);
          }
          const uintx4(&ap)[3] = fa[n2 & 1];
          const uintx4(&bp)[3] = fb[s & 1];
          v[c2] = mfma_bf16(ap[2], bp[0], v[c2]);
          v[c2] = mfma_bf16(ap[0], bp[2], v[c2]);
          v[c2] = mfma_bf16(ap[1], bp[1], v[c2]);
          v[c2] = mfma_bf16(ap[1], bp[0], v[c2]);
          v[c2] = mfma_bf16(ap[0], bp[1], v[c2]);
          v[c2] = mfma_bf16(ap[0], bp[0], v[c2]);
          __builtin_amdgcn_sched_barrier(0);
        }
        KGAT_ATT_PHASE_T(pe);
        KGAT_ATT_PHASE_ADD(1, pa, pb); KGAT_ATT_PHASE_ADD(2, pb, pc); KGAT_ATT_PHASE_ADD(3, pc, pd); KGAT_ATT_PHASE_ADD(4, pd, pe);
#else
        (void)f;
#endif
        // the slot is free once the consumer is through its tile before last
        KGAT_ATT_PHASE_T(pf);
        if (c >= 2) ws_wait_ge(&s_free[cons][c & 1], c - 1);
        KGAT_ATT_PHASE_T(pg);
        KGAT_ATT_PHASE_ADD(5, pf, pg);
#pragma unroll
        for (int c2 = 0; c2 < KT; ++c2) {  // v[c2][j] = V[group i][16 c2 + 4q + j]
          float4 o;
          o.x = v[c2][0]; o.y = v[c2][1]; o.z = v[c2][2]; o.w = v[c2][3];
          *reinterpret_cast<float4*>(vrow + i * LDV + 16 * c2 + 4 * q) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_full[cons][c & 1], c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        KGAT_ATT_PHASE_T(ph2);
        KGAT_ATT_PHASE_ADD(6, pg, ph2);
        KGAT_ATT_PHASE_ADD(7, 0, 1);
      };
      if (n < seg_end) {
        int4 d1 = desc_of(n + NP), d2 = desc_of(n + 2 * NP);
        int32_t h1 = head_idx(d1);
        HBuf hb0, hb1;
        load_head(hb0, head_idx(desc_of(n)));
#define KGAT_WS_PSTEP(HCUR, HNEXT)                                   \
        {                                                            \
          KGAT_ATT_PHASE_T(p0);                                      \
          const int4 d3 = desc_of(n + 3 * NP);                       \
          const int32_t h2 = head_idx(d2);                           \
          load_head(HNEXT, h1);                                      \
          __builtin_amdgcn_sched_barrier(0);                         \
          KGAT_ATT_PHASE_T(p1);                                      \
          KGAT_ATT_PHASE_ADD(0, p0, p1);                             \
          tile_v(HCUR, (n - t) % NC, (n - t) / NC);                  \
          __builtin_amdgcn_sched_barrier(0);                         \
          d1 = d2; d2 = d3; h1 = h2;                                 \
          n += NP;                                                   \
        }
        while (true) {
          KGAT_WS_PSTEP(hb0, hb1)
          if (n >= seg_end) break;
          KGAT_WS_PSTEP(hb1, hb0)
          if (n >= seg_end) break;
        }
#undef KGAT_WS_PSTEP
      }
    } else {
      // ------------------------------------------------------------------ consumer
      const int cons = w - NP;
      int32_t n = t + cons, cnt = 0;
      struct CIdx { int32_t row_off, lg; };
      auto chunk_idx = [&](const int4& d, int32_t p0) -> CIdx {
        int32_t p = p0 + lane;
        p = p < d.w ? p : d.w - 1;
        CIdx c;
        const uint32_t rec = (uint32_t)rec_g[p];
        c.row_off = (int32_t)(rec << ROW_SHIFT);  // (the slot bits fall off the top: N * d * 4 < 4 GiB)
        c.lg = (int32_t)(rec >> 28);
        return c;
      };
      struct EBuf { float4 r[LPE][VPL]; };
      const char* eb = reinterpret_cast<const char*>(ent) + li * 16;
      auto load_edges = [&](EBuf& e, const CIdx& c) {
#if KGAT_WS_ABLATE != 2
#pragma unroll
        for (int s = 0; s < LPE; ++s) {
          const uint32_t eo = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - li + s) << 2, c.row_off);
#pragma unroll
          for (int v = 0; v < VPL; ++v) e.r[s][v] = *reinterpret_cast<const float4*>(eb + eo + v * (LPE * 16));
        }
#else
#pragma unroll
        for (int s = 0; s < LPE; ++s)
#pragma unroll
          for (int v = 0; v < VPL; ++v) e.r[s][v] = make_float4(1.f, 1.f, 1.f, (float)c.row_off);
#endif
      };
      auto edge_phase = [&](const EBuf& e, const CIdx& c, const float* vrow, int32_t p0, int32_t pe) {
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
          if (LPE >= 8) d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xF, 0xF, true));
          mine = li == s ? d : mine;
        }
        if (p0 + lane < pe) logits_g[p0 + lane] = mine;
      };
      if (n < seg_end) {
        int4 d0 = desc_of(n), d1 = desc_of(n + NC), d2 = desc_of(n + 2 * NC);
        CIdx c0 = chunk_idx(d0, d0.z), c1 = chunk_idx(d1, d1.z);
        CIdx cx = chunk_idx(d0, d0.z + kWave), cy = chunk_idx(d0, d0.z + 2 * kWave), cz = chunk_idx(d0, d0.z + 3 * kWave);
        EBuf eb0;
        load_edges(eb0, c0);
        while (true) {
          // records two tiles ahead; the later chunks' of the next tile
          KGAT_ATT_PHASE_T(q0);
          const int4 d3 = desc_of(n + 3 * NC);
          const CIdx c2 = chunk_idx(d2, d2.z);
          const CIdx nx = chunk_idx(d1, d1.z + kWave), ny = chunk_idx(d1, d1.z + 2 * kWave), nz = chunk_idx(d1, d1.z + 3 * kWave);
          __builtin_amdgcn_sched_barrier(0);
          const float* vrow = s_v[cons][cnt & 1];
          KGAT_ATT_PHASE_T(q1);
          ws_wait_ge(&s_full[cons][cnt & 1], cnt + 1);
          KGAT_ATT_PHASE_T(q2);
          edge_phase(eb0, c0, vrow, d0.z, d0.w);
          __builtin_amdgcn_sched_barrier(0);
          KGAT_ATT_PHASE_T(q3);
          if (d0.z + kWave < d0.w) {
            load_edges(eb0, cx);
            edge_phase(eb0, cx, vrow, d0.z + kWave, d0.w);
            if (d0.z + 2 * kWave < d0.w) {
              load_edges(eb0, cy);
              cx = chunk_idx(d0, d0.z + 4 * kWave);  // fifth chunk (caps above 256)
              edge_phase(eb0, cy, vrow, d0.z + 2 * kWave, d0.w);
              if (d0.z + 3 * kWave < d0.w) {
                load_edges(eb0, cz);
                edge_phase(eb0, cz, vrow, d0.z + 3 * kWave, d0.w);
                for (int32_t p0 = d0.z + 4 * kWave; p0 < d0.w; p0 += kWave) {
                  load_edges(eb0, cx);
                  const CIdx cn = chunk_idx(d0, p0 + kWave);
                  edge_phase(eb0, cx, vrow, p0, d0.w);
                  cx = cn;
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          KGAT_ATT_PHASE_T(q4);
          KGAT_ATT_PHASE_ADD(0, q0, q1); KGAT_ATT_PHASE_ADD(1, q1, q2); KGAT_ATT_PHASE_ADD(2, q2, q3); KGAT_ATT_PHASE_ADD(3, q3, q4);
          KGAT_ATT_PHASE_ADD(7, 0, 1);
          // the V rows of this tile are read: hand the slot back, then ask for the next tile's first rows
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_store(&s_free[cons][cnt & 1], cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          n += NC; ++cnt;
          if (n >= seg_end) break;
          load_edges(eb0, c1);
          d0 = d1; d1 = d2; d2 = d3;
          c0 = c1; c1 = c2;
          cx = nx; cy = ny; cz = nz;
        }
      }
    }
    t = seg_end;
  }
#ifdef KGAT_ATT_STAMPS
  if (g_att_phases && lane == 0)
    for (int k2 = 0; k2 < 8; ++k2) g_att_phases[((size_t)blockIdx.x * (kWsThreads / kWave) + w) * 8 + k2] = ph[k2];
#endif
}

template <int D_>
static void launch_att_fold_ws(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  const unsigned grid = a.part_tptr ? a.grid : (unsigned)device_cu_count();
  hipLaunchKernelGGL((att_fold_ws_kernel<D_>), dim3(grid), dim3(kWsThreads), 0, a.st, a.n_rel, a.n_edges, a.rel_ptr, rel_tptr,
                     reinterpret_cast<const int4*>(tiles), a.gptr, a.g_node, a.rec_g, a.ent, a.W_R, a.rel, a.logits_g,
                     a.part_tptr);
}


// ---------------------------------------------------------------------------------------------
// d = k = 128 with wave roles: an experiment of round 3, NOT the shipped kernel (compiled only with
// -DKGAT_ATT_WAVE_ROLES; scripts/micro/att_ws_phases.py 128).  0.575-0.59 ms against 0.47-0.53 ms for
// att_fold_fused128_kernel on the same box.  The stamps: a producer's tile (two products, tanh, cuts) takes
// 16.3 k cycles with the consumer wave on its SIMD - 384 MFMAs of 16 cycles, ~700 vector instructions of its
// own and the consumer's ~1,200 all go through the SIMD's one issue port, and their costs ADD (6.1 k + 2.8 k
// + 4.8 k) -, the consumers wait 10 k cycles per tile for V rows.  The same sum bounds the one-role kernel
// (two waves per SIMD, ~15 k cycles of issue per tile per SIMD): no arrangement of the waves changes it, only
// fewer instructions would.  The d = 64 kernel is in the same place (96 x 16 + ~420 x 4 + LDS ~ 3.5 k of its
// measured 3.85 k cycles per tile per SIMD).  Kept as the record.
// Design notes as written for the experiment:
// d = k = 128 with wave roles (round 3, second half; grouped-order logits, the form the propagation path
// takes).  att_fold_fused128_kernel runs a tile's two products (384 MFMAs, ~6.1 k matrix-pipe cycles) and
// then its edges (rows gathered on demand: nothing fits in flight across the MFMA phase) in ONE wave; its
// partner wave on the SIMD covers part of either phase, and the launch takes 0.44 ms against ~0.23 ms of
// matrix-pipe time and ~0.23 ms of row gather.  Here the eight waves of the workgroup split the work:
//   * PRODUCER p (waves 0..3, one per SIMD): tiles p, p+4, ... of the relation segment - head rows of the
//     next tile in flight across the MFMA phase (a second head-row buffer: the registers the edge phase
//     needed are free in this role), the two products, V rows into the pair's 16 x 128 LDS patch;
//   * CONSUMER p (waves 4..7): the same tiles' edges - records a tile ahead, the first chunk's 64 tail rows
//     (128 registers) requested as soon as the previous tile's edges are done, i.e. in flight while the
//     producer computes; V rows from the patch.
// One patch per pair (W_r's three images take 96 of the 160 KB): full = count + 1 when the V rows are in,
// free = count + 1 when the consumer has read them; the producer computes the next tile meanwhile and
// waits only before it parks.  Counts and flag words restart with every relation segment.  The d = 64 form
// of this idea (KGAT_ATT_WAVE_ROLES above) bought nothing - that kernel sits at 1.25 x its gather floor
// already; here the two halves are equally long and used to run one after the other.
constexpr int kRoles128Threads = 512, kRoles128Pairs = 4;
constexpr int kRoles128SpinMax = 1 << 22;

__device__ __forceinline__ void ws128_wait_ge(const int32_t* flag, int32_t want) {
  for (int spin = 0; spin < kRoles128SpinMax; ++spin) {  // (bounded: a lost wake-up shows as wrong logits, not as a hung GPU)
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __launch_bounds__(kRoles128Threads) void att_fold_roles128_kernel(
    int n_rel, int64_t n_edges, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel_tptr,
    const int4* __restrict__ tiles, const int32_t* __restrict__ gptr, const int32_t* __restrict__ g_node,
    const int32_t* __restrict__ rec_g, const float* __restrict__ ent, const float* __restrict__ W_R,
    const float* __restrict__ rel, float* __restrict__ logits_g, const int32_t* __restrict__ part_tptr) {
  constexpr int D_ = 128, K_ = 128, KS = D_ / 4, KT = K_ / 16, NP = kRoles128Pairs;
  constexpr int S3 = D_ / 32, ROWB = D_ * 2, IMG = D_ * ROWB;
  constexpr int LPE = 8, VPL = D_ / (4 * LPE);
  constexpr int ROW_SHIFT = 9;  // 512-byte rows
  __shared__ __attribute__((aligned(16))) unsigned char s_img[3 * IMG];
  __shared__ __attribute__((aligned(16))) float s_v[NP][16 * D_];
  __shared__ int32_t s_full[NP], s_free[NP];
  const int tid = threadIdx.x;
  const int lane = tid % kWave;
  const int w = __builtin_amdgcn_readfirstlane(tid / kWave);
  const bool producer = w < NP;
  const int pr = w % NP;
  const int i = lane & 15, q = lane >> 4;
  const int li = lane % LPE;

  {  // relation ids outside [0, R): logit 0
    const int64_t n_scored = rel_ptr[n_rel];
    for (int64_t p = n_scored + (int64_t)blockIdx.x * kRoles128Threads + tid; p < n_edges;
         p += (int64_t)gridDim.x * kRoles128Threads)
      logits_g[p] = 0.f;
  }
  const int32_t n_tiles = rel_tptr[n_rel];
  const int32_t t_begin = part_tptr ? part_tptr[blockIdx.x] : (int32_t)((int64_t)n_tiles * blockIdx.x / gridDim.x);
  const int32_t t_end = part_tptr ? part_tptr[blockIdx.x + 1] : (int32_t)((int64_t)n_tiles * (blockIdx.x + 1) / gridDim.x);
  float* vrow = s_v[pr];

#ifdef KGAT_ATT_STAMPS
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
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
    __syncthreads();  // every wave is done with the previous segment: its images, its patches, its flag words
    {
      const float* W = W_R + (size_t)r * D_ * K_;
      int u0 = tid;
      asm volatile("" : "+v"(u0));  // (per-thread offsets recomputed per segment, not kept across the tile loops)
      for (int u = u0; u < D_ * (K_ / 8); u += kRoles128Threads) {
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
      if (tid < NP) {
        s_full[tid] = 0;
        s_free[tid] = 0;
      }
    }
    __syncthreads();

    auto desc_of = [&](int32_t n) -> int4 {  // wave-uniform: kept in scalar registers
      n = n < seg_end ? n : seg_end - 1;
      const int4 d = tiles[n];
      return make_int4(__builtin_amdgcn_readfirstlane(d.x), __builtin_amdgcn_readfirstlane(d.y),
                       __builtin_amdgcn_readfirstlane(d.z), __builtin_amdgcn_readfirstlane(d.w));
    };
    int32_t n = t + pr;
    if (producer) {
      // ------------------------------------------------------------------ producer
      const float* relr = rel + (size_t)r * K_ + 4 * q;
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
      auto head_idx = [&](const int4& d) -> int32_t {
        int32_t g = d.y + i;
        g = g < rend ? g : rend - 1;
        return g_node[g];
      };
        // the two chained products of one tile (att_fold_fused128_kernel's, the same bits): V rows of the tile's
        // 16 groups -> the pair's LDS patch
      auto mfma_phase = [&](const HBuf& f, int32_t cnt) {
        floatx4 acc[KT];
  #pragma unroll
        for (int c = 0; c < KT; ++c) {  // e_r is where the accumulation starts: acc = e_r + e_h W_r
          const float4 rv = *reinterpret_cast<const float4*>(relr + 16 * c);
          acc[c] = (floatx4){rv.x, rv.y, rv.z, rv.w};
        }
        floatx4 v[KT];
        {
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
              const shortx4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(s_img + pc * IMG + o));
              const shortx4 hi4 =
                  __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_shortx4*)(s_img + pc * IMG + o + ROWB * 16));
              ap[pc] = __builtin_bit_cast(uintx4, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
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
        }
  #pragma unroll
        for (int c = 0; c < KT; ++c)
  #pragma unroll
          for (int j = 0; j < 4; ++j) acc[c][j] = att_tanh_scaled(acc[c][j] * kTwoLog2e);
  #pragma unroll
        for (int c2 = 0; c2 < KT; ++c2) v[c2] = (floatx4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        {
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
        }
        // v[c2][j] = V[group i][16c2 + 4q + j] -> the wave's patch, row = group slot; the 16-byte chunks
        // of a row are swizzled by the slot's parity (chunk ^ 8) so that the edge phase's reads of two
        // different slots by one 16-lane group fall into different bank halves
        // one patch per pair: it is free once the consumer is through the pair's previous tile
        KGAT_ATT_PHASE_T(pf);
        if (cnt >= 1) ws128_wait_ge(&s_free[pr], cnt);
        KGAT_ATT_PHASE_T(pg);
        KGAT_ATT_PHASE_ADD(5, pf, pg);
  #pragma unroll
        for (int c2 = 0; c2 < KT; ++c2) {
          float4 o;
          o.x = v[c2][0]; o.y = v[c2][1]; o.z = v[c2][2]; o.w = v[c2][3];
          *reinterpret_cast<float4*>(vrow + i * D_ + 4 * ((4 * c2 + q) ^ ((i & 1) << 3))) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_full[pr], cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };

      if (n < seg_end) {
        int4 d1 = desc_of(n + NP), d2 = desc_of(n + 2 * NP);
        int32_t h1 = head_idx(d1);
        HBuf hb0, hb1;
        load_head(hb0, head_idx(desc_of(n)));
        int32_t cnt = 0;
#define KGAT_R128_PSTEP(HCUR, HNEXT)                       \
        {                                                  \
          const int4 d3 = desc_of(n + 3 * NP);             \
          const int32_t h2 = head_idx(d2);                 \
          load_head(HNEXT, h1);                            \
          __builtin_amdgcn_sched_barrier(0);               \
          KGAT_ATT_PHASE_T(p1);                            \
          mfma_phase(HCUR, cnt);                           \
          __builtin_amdgcn_sched_barrier(0);               \
          KGAT_ATT_PHASE_T(p2);                            \
          KGAT_ATT_PHASE_ADD(1, p1, p2);                   \
          KGAT_ATT_PHASE_ADD(7, 0, 1);                     \
          d1 = d2; d2 = d3; h1 = h2;                       \
          n += NP; ++cnt;                                  \
        }
        while (true) {
          KGAT_R128_PSTEP(hb0, hb1)
          if (n >= seg_end) break;
          KGAT_R128_PSTEP(hb1, hb0)
          if (n >= seg_end) break;
        }
#undef KGAT_R128_PSTEP
      }
    } else {
      // ------------------------------------------------------------------ consumer
      struct CIdx { int32_t row_off, lg; };
      auto chunk_idx = [&](const int4& d, int32_t p0) -> CIdx {
        int32_t p = p0 + lane;
        p = p < d.w ? p : d.w - 1;
        CIdx c;
        const uint32_t rec = (uint32_t)rec_g[p];
        c.row_off = (int32_t)(rec << ROW_SHIFT);
        c.lg = (int32_t)(rec >> 28);
        return c;
      };
      struct EBuf { float4 r[LPE][VPL]; };  // the tail rows of 64 positions: 128 registers
      const char* eb = reinterpret_cast<const char*>(ent) + li * 16;
      auto load_rows = [&](EBuf& e, const CIdx& c) {
#pragma unroll
        for (int s = 0; s < LPE; ++s) {
          const uint32_t eo = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - li + s) << 2, c.row_off);
#pragma unroll
          for (int v = 0; v < VPL; ++v) e.r[s][v] = *reinterpret_cast<const float4*>(eb + eo + v * (LPE * 16));
        }
      };
      auto edge_chunk = [&](const EBuf& e, const CIdx& c, int32_t p0, int32_t pe) {
        float mine = 0.f;
#pragma unroll
        for (int s = 0; s < LPE; ++s) {
          const int32_t lg = __builtin_amdgcn_ds_bpermute((lane - li + s) << 2, c.lg);
          const float* vr = vrow + lg * D_;
          const int sw = (lg & 1) << 3;
          float d = 0.f;
#pragma unroll
          for (int v = 0; v < VPL; ++v) {
            const float4 b = *reinterpret_cast<const float4*>(vr + 4 * ((li + LPE * v) ^ sw));
            d = v == 0 ? e.r[s][v].x * b.x : fmaf(e.r[s][v].x, b.x, d);
            d = fmaf(e.r[s][v].y, b.y, d);
            d = fmaf(e.r[s][v].z, b.z, d);
            d = fmaf(e.r[s][v].w, b.w, d);
          }
          d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xF, 0xF, true));
          d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xF, 0xF, true));
          d += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xF, 0xF, true));
          mine = li == s ? d : mine;
        }
        if (p0 + lane < pe) logits_g[p0 + lane] = mine;
      };
      if (n < seg_end) {
        int4 d0 = desc_of(n), d1 = desc_of(n + NP);
        CIdx c0 = chunk_idx(d0, d0.z), c1 = chunk_idx(d1, d1.z);
        EBuf eb0;
        load_rows(eb0, c0);
        int32_t cnt = 0;
        while (true) {
          const int4 d2 = desc_of(n + 2 * NP);
          const CIdx c2 = chunk_idx(d2, d2.z);
          CIdx cx = chunk_idx(d0, d0.z + kWave);  // the tile's second chunk, if it has one
          __builtin_amdgcn_sched_barrier(0);
          KGAT_ATT_PHASE_T(q1);
          ws128_wait_ge(&s_full[pr], cnt + 1);
          KGAT_ATT_PHASE_T(q2);
          edge_chunk(eb0, c0, d0.z, d0.w);
          __builtin_amdgcn_sched_barrier(0);
          KGAT_ATT_PHASE_T(q3);
          for (int32_t p0 = d0.z + kWave; p0 < d0.w; p0 += kWave) {
            load_rows(eb0, cx);
            const CIdx cn = chunk_idx(d0, p0 + kWave);
            edge_chunk(eb0, cx, p0, d0.w);
            cx = cn;
          }
          __builtin_amdgcn_sched_barrier(0);
          KGAT_ATT_PHASE_T(q4);
          KGAT_ATT_PHASE_ADD(1, q1, q2); KGAT_ATT_PHASE_ADD(2, q2, q3); KGAT_ATT_PHASE_ADD(3, q3, q4);
          KGAT_ATT_PHASE_ADD(7, 0, 1);
          // the V rows are read: hand the patch back, then ask for the next tile's first rows
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_store(&s_free[pr], cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          n += NP; ++cnt;
          if (n >= seg_end) break;
          load_rows(eb0, c1);
          d0 = d1; d1 = d2;
          c0 = c1; c1 = c2;
        }
      }
    }
    t = seg_end;
  }
#ifdef KGAT_ATT_STAMPS
  if (g_att_phases && lane == 0)
    for (int k2 = 0; k2 < 8; ++k2) g_att_phases[((size_t)blockIdx.x * (kRoles128Threads / kWave) + w) * 8 + k2] = ph[k2];
#endif
}

static void launch_att_fold_roles128(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles) {
  const unsigned grid = a.part_tptr ? a.grid : (unsigned)device_cu_count();
  hipLaunchKernelGGL(att_fold_roles128_kernel, dim3(grid), dim3(kRoles128Threads), 0, a.st, a.n_rel, a.n_edges, a.rel_ptr,
                     rel_tptr, reinterpret_cast<const int4*>(tiles), a.gptr, a.g_node, a.rec_g, a.ent, a.W_R, a.rel,
                     a.logits_g, a.part_tptr);
}


