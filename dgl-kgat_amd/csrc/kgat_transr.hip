// TransR KG-embedding step, for gfx950: loss AND gradients of reference
// models.py:114-133 (`transR`, with `bmm_maybe_select` :13-47 and `_L2_loss_mean` :9-11) for one
// batch of triplets in a handful of launches:
//     a_x = e_x W_r                    (x = head, positive tail, negative tail)
//     u_x = a_x / max(|a_x|, 1e-12),   u_r = rel_r / max(|rel_r|, 1e-12)
//     pos = |u_h + u_r - u_p|^2,       neg = |u_h + u_r - u_n|^2
//     loss = mean softplus(pos - neg) + lambda * sum_{v in h,r,p,n} mean(|u_v|^2 / 2)
// The reference runs this through ~100 small torch kernels per step (gather of W_R[r] =
// B x d x k floats per operand, three bmm, four normalisations, their backward, an atomic
// index_add into W_R's gradient and four sort-based embedding backwards): 1.6 ms per step on
// MI355X at B = 2048, and ~1,800 steps per epoch on amazon-book - the larger half of the epoch.
//
// Here: the batch is sorted by relation once (one-workgroup bitonic sort); one wavefront per
// sample does the three projections, the whole loss head, its backward and the three
// d-vectors grad_a W_r^T; W_R's gradient is a per-relation sum of outer products (partial sums
// per 64-sample chunk, then an ordered reduction); the entity gradient rows are added into the
// dense gradient in sorted-id order.  No atomics: every sum has a fixed order, so the step is
// bitwise reproducible.  All arithmetic fp32 on the vector ALU - the step is ~75 MFLOP, launch
// and latency bound, not MFMA work.
#include <math.h>

#include "kgat_adam_common.h"

namespace kgat {

constexpr int kTrMaxDim = 128;    // d, k <= 128
constexpr int kTrChunk = 64;      // samples per partial W-gradient block
constexpr int kTrSmallSort = 8192;
constexpr int kTrMaxRel = 4096;   // relations (chunk table built by one workgroup, 4 keys per thread)
constexpr int kTrSlotBits = 14;   // row_slot word = call tag << 14 | (first sorted position of the row's run + 1); 3 * batch <= 8192 < 2^14

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// ---- one-workgroup stable sort of n <= 8192 keys (< 2^19), payload = position in the input,
// packed as key << 13 | position.  LSD radix sort, 9-bit digits (8 with the 64-bit packing), both buffers and the
// counters in LDS: each of the 16 wavefronts owns a contiguous slice of the current order and its own column
// of the 512 x 16 counter table, counts its digits (LDS atomics - counts do not depend on order),
// a block scan turns the table into offsets in (digit, wavefront) order, and the wavefront walks
// its slice 64 keys at a time ranking equal digits by lane with ballots - so equal keys keep
// their order.  Also emits, for keys in [0, n_keys], offsets[v] = first sorted position whose key
// is >= v (offsets == NULL: skipped).
// keys_b / keys_c non-NULL: the key list is the interleaving (keys[i], keys_b[i], keys_c[i]) of
// three arrays of n/3 entries (the three entity ids of every sample).
constexpr int kSsWaves = 16;

// PT: the packed (key, position) type - uint32_t for keys below 2^19 (64 KB of LDS for the two buffers), uint64_t for
// any int32 key (128 KB: entity ids of graphs beyond 524,288 nodes, e.g. BASELINE configs[4]'s 10 M).
struct SortJob {
  int32_t n;
  int key_bits;
  const int32_t *keys, *keys_b, *keys_c;
  int32_t *order, *sorted_keys;
  int32_t n_keys;
  int32_t *offsets, *chunk_ptr;
  int2* chunks;
  // (the KG phase's presort only) inv[o] = sorted position of input position o; run_len[p] = length of the run of
  // equal keys that STARTS at sorted position p, 0 where p is not the first of its run
  int32_t *inv = nullptr, *run_len = nullptr;
};

template <typename PT>
__device__ __forceinline__ void small_sort_body(const SortJob& job) {
  const int32_t n = job.n, n_keys = job.n_keys;
  const int key_bits = job.key_bits;
  const int32_t* __restrict__ keys = job.keys;
  const int32_t* __restrict__ keys_b = job.keys_b;
  const int32_t* __restrict__ keys_c = job.keys_c;
  int32_t* __restrict__ order = job.order;
  int32_t* __restrict__ sorted_keys = job.sorted_keys;
  int32_t* __restrict__ offsets = job.offsets;
  int32_t* __restrict__ chunk_ptr = job.chunk_ptr;
  int2* __restrict__ chunks = job.chunks;
  // digits of 9 bits with the 32-bit packing (eighteen-bit entity ids: two passes instead of three), 8 with the
  // 64-bit one (its buffers leave 32 KB of LDS: 256 x 16 counters)
  constexpr int DB = sizeof(PT) == 4 ? 9 : 8;
  constexpr int NDIG = 1 << DB, CPT = NDIG * kSsWaves / 1024;   // counters per thread in the scan
  __shared__ PT s_buf[2][kTrSmallSort];
  __shared__ int32_t s_cnt[NDIG * kSsWaves];
  __shared__ int32_t s_wsum[kSsWaves];
  const int tid = threadIdx.x, lane = tid % kWave, w = tid / kWave;
  for (int32_t i = tid; i < n; i += 1024) {
    int32_t key;
    if (keys_b) {
      const int32_t q = i / 3, v = i - 3 * q;
      key = v == 0 ? keys[q] : (v == 1 ? keys_b[q] : keys_c[q]);
    } else {
      key = keys[i];
    }
    s_buf[0][i] = ((PT)(uint32_t)key << 13) | (PT)i;
  }
  const int32_t slice = ((n + kSsWaves * kWave - 1) / (kSsWaves * kWave)) * kWave;  // per wavefront, multiple of 64
  const int32_t lo = w * slice, hi = lo + slice < n ? lo + slice : n;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int cur = 0;
  const int passes = (key_bits + DB - 1) / DB > 0 ? (key_bits + DB - 1) / DB : 1;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 13 + DB * pass;
    for (int i = tid; i < NDIG * kSsWaves; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    for (int32_t i = lo + lane; i < hi; i += kWave) atomicAdd(&s_cnt[(uint32_t)((s_buf[cur][i] >> shift) & (PT)(NDIG - 1)) * kSsWaves + w], 1);
    __syncthreads();
    {  // exclusive scan of the counters in (digit, wavefront) order, CPT per thread
      int32_t c[CPT];
      int32_t mine = 0;
#pragma unroll
      for (int q = 0; q < CPT; ++q) { c[q] = s_cnt[CPT * tid + q]; mine += c[q]; }
      int32_t inc = mine;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int32_t up = __shfl_up(inc, d, kWave);
        if (lane >= d) inc += up;
      }
      if (lane == kWave - 1) s_wsum[w] = inc;
      __syncthreads();
      int32_t base = inc - mine;
      for (int q = 0; q < w; ++q) base += s_wsum[q];
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        s_cnt[CPT * tid + q] = base;
        base += c[q];
      }
    }
    __syncthreads();
    for (int32_t i0 = lo; i0 < hi; i0 += kWave) {
      const int32_t i = i0 + lane;
      const bool valid = i < hi;
      const PT v = valid ? s_buf[cur][i] : (PT)0;
      const uint32_t dgt = (uint32_t)((v >> shift) & (PT)(NDIG - 1));
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < DB; ++bit) {
        const bool on = (dgt >> bit) & 1u;
        const uint64_t bal = __ballot(on);
        peers &= on ? bal : ~bal;
      }
      const int rank = __popcll(peers & lt_mask);
      const int32_t basep = valid ? s_cnt[dgt * kSsWaves + w] : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane has its base before a leader moves it
      if (valid) {
        s_buf[cur ^ 1][basep + rank] = v;
        if (rank == 0) s_cnt[dgt * kSsWaves + w] = basep + __popcll(peers);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    cur ^= 1;
  }
  const PT* s = s_buf[cur];
  for (int32_t p = tid; p <= n; p += 1024) {
    if (p < n) {
      order[p] = (int32_t)(s[p] & 8191u);
      if (sorted_keys) sorted_keys[p] = (int32_t)(s[p] >> 13);
      if (job.inv) job.inv[(int32_t)(s[p] & 8191u)] = p;
      if (job.run_len) {
        const PT key = s[p] >> 13;
        int32_t len = 0;
        if (p == 0 || (s[p - 1] >> 13) != key) {
          len = 1;
          while (p + len < n && (s[p + len] >> 13) == key) ++len;
        }
        job.run_len[p] = len;
      }
    }
    if (offsets) {
      int32_t prev = p == 0 ? -1 : (int32_t)(s[p - 1] >> 13);
      int32_t curk = p == n ? n_keys : (int32_t)(s[p] >> 13);
      curk = curk > n_keys ? n_keys : curk;
      prev = prev > n_keys ? n_keys : prev;
      for (int32_t v = prev + 1; v <= curk; ++v) offsets[v] = p;
    }
  }
  if (chunk_ptr == nullptr) return;
  // chunk table of the per-relation gradient sums (n_keys <= kTrMaxRel): chunk_ptr[v] = first chunk
  // of key v, chunks[c] = (key, first sorted position); a chunk is <= kTrChunk positions of one key
  int32_t* s_off = reinterpret_cast<int32_t*>(s_buf[cur ^ 1]);  // the other buffer is free now
  __syncthreads();
  for (int32_t p = tid; p <= n; p += 1024) {
    int32_t prev = p == 0 ? -1 : (int32_t)(s[p - 1] >> 13);
    int32_t curk = p == n ? n_keys : (int32_t)(s[p] >> 13);
    curk = curk > n_keys ? n_keys : curk;
    prev = prev > n_keys ? n_keys : prev;
    for (int32_t v = prev + 1; v <= curk; ++v) s_off[v] = p;
  }
  __syncthreads();
  {
    int32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int32_t v = 4 * tid + q;
      c[q] = v < n_keys ? (s_off[v + 1] - s_off[v] + kTrChunk - 1) / kTrChunk : 0;
    }
    const int32_t mine = c[0] + c[1] + c[2] + c[3];
    int32_t inc = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int32_t up = __shfl_up(inc, d, kWave);
      if (lane >= d) inc += up;
    }
    if (lane == kWave - 1) s_wsum[w] = inc;
    __syncthreads();
    int32_t base = inc - mine;
    for (int q = 0; q < w; ++q) base += s_wsum[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int32_t v = 4 * tid + q;
      if (v <= n_keys) chunk_ptr[v] = base;
      if (v < n_keys)
        for (int32_t j = 0; j < c[q]; ++j) chunks[base + j] = make_int2(v, s_off[v] + j * kTrChunk);
      base += c[q];
    }
  }
}

// The step's sorts are independent of everything but the batch ids, and so is the zero fill of the dense entity
// gradient: ONE launch - workgroup 0 (and 1) sort, the others clear `zero` (n_zero floats, a multiple of 4).  (Round 4
// ran relation sort -> ... -> memset -> id sort in stream order: two one-workgroup launches of ~18 us and a 41 MB
// memset on an otherwise idle chip.)
template <typename PT>
__global__ __launch_bounds__(1024) void small_sort_kernel(SortJob a, SortJob b, int n_jobs, float* __restrict__ zero,
                                                          int64_t n_zero) {
  if ((int)blockIdx.x < n_jobs) {
    small_sort_body<PT>(blockIdx.x == 0 ? a : b);
    return;
  }
  const int64_t n4 = n_zero / 4;
  const int64_t stride = (int64_t)(gridDim.x - n_jobs) * 1024;
  float4* z = reinterpret_cast<float4*>(zero);
  for (int64_t i = (int64_t)(blockIdx.x - n_jobs) * 1024 + threadIdx.x; i < n4; i += stride)
    z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- per-sample kernel: projections, loss head, its backward, grad_a W_r^T
// One wavefront per (relation-sorted) sample s; b = order[s] is the sample's place in the batch.
// Row 3 b + {0,1,2} of DX belongs to (head, positive tail, negative tail) of batch sample b; GA, GR
// and XS (the gathered entity rows) are written in relation-sorted order (rows 3 s + v, s) for the
// W-gradient kernel, which then reads contiguous memory only.
// WLDS (d, k <= 64, the reference's sizes): the wavefront copies its relation's W_r into LDS first (16 KB, rows
// padded by one float so that both the row-wise and the column-wise walk are free of bank conflicts) - one round trip
// of sixteen 16-byte loads per lane instead of a load per step of the two 64-step loops (the first walked W_r row by
// row with a dependent global load in every step, the second read a 256-byte-strided column per lane: 64 cache lines
// per load instruction): 22 -> ~10 us for the launch.  The arithmetic and its order are unchanged: same bits.
constexpr int kTrWLds = 64;
struct TrStepMarks {
  const int32_t* inv_pos = nullptr;     // non-NULL: the KG phase's form
  const int32_t* sorted_ids = nullptr;
  const int32_t* run_len = nullptr;
  unsigned long long* row_slot = nullptr;
  unsigned long long tag = 0ull;
  int sample_blocks = 0;
};
template <bool GRAD, bool WLDS>
__global__ __launch_bounds__(256) void transr_sample_kernel(
    int32_t batch, int d, int k, const int32_t* __restrict__ order, const int32_t* __restrict__ h,
    const int32_t* __restrict__ r, const int32_t* __restrict__ pt, const int32_t* __restrict__ nt,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel, float lambda,
    float* __restrict__ losses, float* __restrict__ GA, float* __restrict__ GR, float* __restrict__ DX,
    float* __restrict__ XS, const TrStepMarks mk = TrStepMarks{}) {
  // the KG phase's form (mk.inv_pos given): workgroups behind the samples' tag the batch's entity rows in row_slot -
  // one thread per sorted position; the first of a run writes call tag << 14 | position + 1 - and the samples' DX rows
  // go to their SORTED positions (a row's contributions are then consecutive: the Adam launch sums them itself)
  if (mk.inv_pos != nullptr && (int)blockIdx.x >= mk.sample_blocks) {
    const int32_t q = ((int)blockIdx.x - mk.sample_blocks) * 256 + (int)threadIdx.x;
    if (q < 3 * batch && mk.run_len[q] > 0)
      mk.row_slot[mk.sorted_ids[q]] = (mk.tag << kTrSlotBits) | (unsigned long long)(q + 1);
    return;
  }
  __shared__ float s_x[256 / kWave][3][kTrMaxDim];
  __shared__ __attribute__((aligned(16))) float s_w[WLDS ? 256 / kWave : 1][WLDS ? kTrWLds * (kTrWLds + 1) : 1];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int32_t s = blockIdx.x * (256 / kWave) + wv;
  if (s >= batch) return;
  const int32_t b = order[s];
  const int32_t rr = r[b];
  const int32_t ids[3] = {h[b], pt[b], nt[b]};
  // rows of DX: 3 b + v, or (KG phase) the sorted position of that entry of the id list
  int32_t dxr[3] = {3 * b, 3 * b + 1, 3 * b + 2};
  if (mk.inv_pos != nullptr) { dxr[0] = mk.inv_pos[3 * b]; dxr[1] = mk.inv_pos[3 * b + 1]; dxr[2] = mk.inv_pos[3 * b + 2]; }
  const float* W = W_R + (size_t)rr * d * k;
  float(*sx)[kTrMaxDim] = s_x[wv];
  float* sw = s_w[wv];
  const int ks = k + 1;   // LDS row stride of W_r
  if (WLDS) {
    const int n4 = d * k / 4, k4 = k / 4;   // (k is a multiple of 4)
    constexpr int UB = kTrWLds * kTrWLds / 4 / kWave;   // 16 loads per lane at 64 x 64
    float4 wv4[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int f = lane + kWave * u;
      wv4[u] = reinterpret_cast<const float4*>(W)[f < n4 ? f : n4 - 1];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int f = lane + kWave * u;
      if (f < n4) {
        const int row = f / k4, col = 4 * (f - row * k4);
        float* q = sw + row * ks + col;
        q[0] = wv4[u].x; q[1] = wv4[u].y; q[2] = wv4[u].z; q[3] = wv4[u].w;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 3; ++v)
    for (int i = lane; i < d; i += kWave) {
      const float x = ent[(size_t)ids[v] * d + i];
      sx[v][i] = x;
      if (GRAD) XS[((size_t)3 * s + v) * d + i] = x;  // the rows again, in relation-sorted order, for the W gradient
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  constexpr int JV = kTrMaxDim / kWave;  // column slots per lane
  float a[3][JV], er[JV];
#pragma unroll
  for (int c = 0; c < JV; ++c) {
    a[0][c] = a[1][c] = a[2][c] = 0.f;
    const int j = lane + c * kWave;
    er[c] = j < k ? rel[(size_t)rr * k + j] : 0.f;
  }
  if (WLDS) {   // (k <= 64: one column slot per lane)
    const int jj = lane < k ? lane : 0;
    for (int i0 = 0; i0 < d; i0 += 4) {   // d is a multiple of 4: four steps' LDS reads in flight together
      float wq[4], xq[3][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wq[t] = sw[(i0 + t) * ks + jj];
        xq[0][t] = sx[0][i0 + t]; xq[1][t] = sx[1][i0 + t]; xq[2][t] = sx[2][i0 + t];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float w = lane < k ? wq[t] : 0.f;
        a[0][0] = fmaf(xq[0][t], w, a[0][0]);
        a[1][0] = fmaf(xq[1][t], w, a[1][0]);
        a[2][0] = fmaf(xq[2][t], w, a[2][0]);
      }
    }
  } else {
  for (int i = 0; i < d; ++i) {
    const float x0 = sx[0][i], x1 = sx[1][i], x2 = sx[2][i];
#pragma unroll
    for (int c = 0; c < JV; ++c) {
      const int j = lane + c * kWave;
      const float w = j < k ? W[(size_t)i * k + j] : 0.f;
      a[0][c] = fmaf(x0, w, a[0][c]);
      a[1][c] = fmaf(x1, w, a[1][c]);
      a[2][c] = fmaf(x2, w, a[2][c]);
    }
  }
  }
  // F.normalize(p=2, dim=1, eps=1e-12) of the three projections and the relation row
  float nrm[4], u[4][JV];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < JV; ++c) {
      const float t = v < 3 ? a[v][c] : er[c];
      ss = fmaf(t, t, ss);
    }
    nrm[v] = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
    for (int c = 0; c < JV; ++c) u[v][c] = (v < 3 ? a[v][c] : er[c]) / nrm[v];
  }
  float dp[JV], dn[JV], pos = 0.f, neg = 0.f, reg = 0.f;
#pragma unroll
  for (int c = 0; c < JV; ++c) {
    dp[c] = u[0][c] + u[3][c] - u[1][c];
    dn[c] = u[0][c] + u[3][c] - u[2][c];
    pos = fmaf(dp[c], dp[c], pos);
    neg = fmaf(dn[c], dn[c], neg);
#pragma unroll
    for (int v = 0; v < 4; ++v) reg = fmaf(u[v][c], u[v][c], reg);
  }
  pos = wave_sum(pos);
  neg = wave_sum(neg);
  reg = wave_sum(reg);
  const float z = pos - neg;  // -logsigmoid(neg - pos) = softplus(z)
  if (lane == 0) losses[s] = fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))) + lambda * 0.5f * reg;
  if (!GRAD) return;
  const float sg = 1.f / (1.f + expf(-z));  // d softplus / dz
  const float c2 = 2.f * sg / (float)batch, lb = lambda / (float)batch;
  // gradients with respect to the normalised vectors, then through the normalisation
  float g[4][JV];
#pragma unroll
  for (int c = 0; c < JV; ++c) {
    const float dd = c2 * (dp[c] - dn[c]);
    g[0][c] = dd + lb * u[0][c];
    g[3][c] = dd + lb * u[3][c];
    g[1][c] = -c2 * dp[c] + lb * u[1][c];
    g[2][c] = c2 * dn[c] + lb * u[2][c];
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < JV; ++c) dot = fmaf(u[v][c], g[v][c], dot);
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < JV; ++c) g[v][c] = (g[v][c] - u[v][c] * dot) / nrm[v];
  }
  // grad wrt the projections (rows 3s..3s+2 of GA) and the relation row (row s of GR), relation-sorted
#pragma unroll
  for (int c = 0; c < JV; ++c) {
    const int j = lane + c * kWave;
    if (j < k) {
      GA[((size_t)3 * s + 0) * k + j] = g[0][c];
      GA[((size_t)3 * s + 1) * k + j] = g[1][c];
      GA[((size_t)3 * s + 2) * k + j] = g[2][c];
      GR[(size_t)s * k + j] = g[3][c];
    }
  }
  // grad wrt the entity rows: dx[i] = sum_j ga[j] W[i][j]; ga goes through the wave's LDS patch
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int v = 0; v < 3; ++v)
#pragma unroll
    for (int c = 0; c < JV; ++c) {
      const int j = lane + c * kWave;
      if (j < k) sx[v][j] = g[v][c];
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (WLDS) {   // (d <= 64: one row per lane; lane i walks row i of the LDS copy, stride k + 1: conflict-free)
    const int i = lane < d ? lane : 0;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    for (int j0 = 0; j0 < k; j0 += 4) {
      float wq[4], gq[3][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wq[t] = sw[i * ks + j0 + t];
        gq[0][t] = sx[0][j0 + t]; gq[1][t] = sx[1][j0 + t]; gq[2][t] = sx[2][j0 + t];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        x0 = fmaf(gq[0][t], wq[t], x0);
        x1 = fmaf(gq[1][t], wq[t], x1);
        x2 = fmaf(gq[2][t], wq[t], x2);
      }
    }
    if (lane < d) {
      DX[(size_t)dxr[0] * d + lane] = x0;
      DX[(size_t)dxr[1] * d + lane] = x1;
      DX[(size_t)dxr[2] * d + lane] = x2;
    }
    return;
  }
  for (int i = lane; i < d; i += kWave) {
    const float* wr = W + (size_t)i * k;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    for (int j = 0; j < k; ++j) {
      const float w = wr[j];
      x0 = fmaf(sx[0][j], w, x0);
      x1 = fmaf(sx[1][j], w, x1);
      x2 = fmaf(sx[2][j], w, x2);
    }
    DX[(size_t)dxr[0] * d + i] = x0;
    DX[(size_t)dxr[1] * d + i] = x1;
    DX[(size_t)dxr[2] * d + i] = x2;
  }
}

// ---- W_R gradient: partial sums of x^T ga per chunk of <= 64 relation-sorted samples

// Thread t owns 4 x 4 blocks of the d x k outer-product sum: block index t, t + 256, ... over
// (d/4) x (k/4) blocks, so a sample costs two 16-byte LDS reads per 16 fmas.  d, k multiples of 4.
// The staging area of the partial kernels (dynamic LDS): `st` samples of x (3 rows of d), g (3 rows of k) and the
// relation row (k) each, row strides sxs / sgs = the width rounded up to 32 floats + 16 (three rows apart - the four
// sample slots of an MFMA operand - then fall 16 banks apart instead of on the same ones).
struct TrStageGeom {
  int st, sxs, sgs;
};
static TrStageGeom transr_stage_geom(int d, int k) {
  TrStageGeom g;
  g.sxs = (d + 31) / 32 * 32 + 16;
  g.sgs = (k + 31) / 32 * 32 + 16;
  const int per_sample = 3 * g.sxs + 3 * g.sgs + (k + 3) / 4 * 4;
  int st = (150 * 1024 / 4) / per_sample;
  st = st > kTrChunk ? kTrChunk : st;
  g.st = st / 4 * 4;
  return g;
}
static size_t transr_stage_bytes(int d, int k) {
  const TrStageGeom g = transr_stage_geom(d, k);
  return (size_t)g.st * (3 * g.sxs + 3 * g.sgs + (k + 3) / 4 * 4) * sizeof(float);
}

// `ns` samples from s0 on into the staging area; rows past `ns` up to `fill` read as zero.  The global side is
// contiguous (rows of d / k floats back to back), so the block walks it 16 bytes per thread, UB loads in flight per
// thread before the first LDS store (a load and its store inside one loop iteration made every row a round trip of its
// own - 24 in a row, two thirds of the kernel's time; element-wise index arithmetic, three divisions by run-time widths
// per float, was most of the rest).
__device__ __forceinline__ void transr_stage_rows(float* __restrict__ dst, int stride, const float* __restrict__ src,
                                                  int width, int n_rows) {
  constexpr int UB = 6;
  const int n4 = n_rows * width / 4;   // float4 pieces (width is a multiple of 4)
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (int base = threadIdx.x; base < n4; base += 256 * UB) {
    float4 v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int f = base + 256 * u;
      v[u] = s4[f < n4 ? f : n4 - 1];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int f = base + 256 * u;
      if (f < n4) {
        const unsigned e = 4u * (unsigned)f, row = e / (unsigned)width, col = e - row * (unsigned)width;
        *reinterpret_cast<float4*>(dst + (size_t)row * stride + col) = v[u];
      }
    }
  }
}

// columns [16 ti, 16 ti + 16) of `n_rows` rows, for the row tiles ti = ts, ts + TS, ... of a split block
__device__ __forceinline__ void transr_stage_col_tiles(float* __restrict__ dst, int stride, const float* __restrict__ src,
                                                       int width, int n_rows, int ts, int TS) {
  const int n_ti = (width / 16 - ts + TS - 1) / TS;       // row tiles of this block
  const int per_row = 4 * n_ti;                            // float4 pieces per row
  const int n4 = n_rows * per_row;
  constexpr int UB = 6;
  for (int base = threadIdx.x; base < n4; base += 256 * UB) {
    float4 v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      int f = base + 256 * u;
      f = f < n4 ? f : n4 - 1;
      const int row = f / per_row, pc = f - row * per_row;
      v[u] = *reinterpret_cast<const float4*>(src + (size_t)row * width + 16 * (ts + TS * (pc >> 2)) + 4 * (pc & 3));
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int f = base + 256 * u;
      if (f < n4) {
        const int row = f / per_row, pc = f - row * per_row;
        *reinterpret_cast<float4*>(dst + (size_t)row * stride + 16 * (ts + TS * (pc >> 2)) + 4 * (pc & 3)) = v[u];
      }
    }
  }
}

// ts / TS: the block computes the output tiles of the row tiles ti = ts, ts + TS, ... only (MFMA form): it stages
// those columns of x, all of g, and - split 0 alone - the relation rows
__device__ __forceinline__ void transr_stage_load(float* __restrict__ sx, float* __restrict__ sg, float* __restrict__ sr,
                                                  const TrStageGeom& g, int d, int k, int32_t s0, int ns, int fill,
                                                  const float* __restrict__ XS, const float* __restrict__ GA,
                                                  const float* __restrict__ GR, int ts = 0, int TS = 1) {
  const int kr = (k + 3) / 4 * 4;
  if (TS > 1) transr_stage_col_tiles(sx, g.sxs, XS + (size_t)3 * s0 * d, d, 3 * ns, ts, TS);
  else transr_stage_rows(sx, g.sxs, XS + (size_t)3 * s0 * d, d, 3 * ns);
  transr_stage_rows(sg, g.sgs, GA + (size_t)3 * s0 * k, k, 3 * ns);
  if (ts == 0) transr_stage_rows(sr, kr, GR + (size_t)s0 * k, k, ns);
  // zero rows behind the last sample (the MFMA form rounds the sample count up to a multiple of four)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int hl = lane >> 5, c4 = 4 * (lane & 31);
  for (int rr = 3 * ns + 2 * w + hl; rr < 3 * fill; rr += 8) {
    if (c4 < d) *reinterpret_cast<float4*>(sx + (size_t)rr * g.sxs + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < k) *reinterpret_cast<float4*>(sg + (size_t)rr * g.sgs + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__device__ __forceinline__ void transr_wgrad_partial_body(
    float* s_dyn, const TrStageGeom g, int bx, int d, int k, int n_rel, const int32_t* __restrict__ seg,
    const float* __restrict__ XS, const float* __restrict__ GA, const float* __restrict__ GR,
    const int32_t* __restrict__ chunk_ptr, const int2* __restrict__ chunks, float* __restrict__ part) {
  if ((int32_t)bx >= chunk_ptr[n_rel]) return;
  const int2 ck = chunks[bx];
  const int32_t beg = ck.y, end = beg + kTrChunk < seg[ck.x + 1] ? beg + kTrChunk : seg[ck.x + 1];
  const int kr = (k + 3) / 4 * 4;
  float* const sx = s_dyn;
  float* const sg = sx + (size_t)g.st * 3 * g.sxs;
  float* const sr = sg + (size_t)g.st * 3 * g.sgs;
  constexpr int TV = (kTrMaxDim / 4) * (kTrMaxDim / 4) / 256;  // 4 x 4 blocks per thread at the largest size
  float acc[TV][4][4], racc = 0.f;
#pragma unroll
  for (int o = 0; o < TV; ++o)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[o][a][c] = 0.f;
  const int kb = k / 4, n_blk = (d / 4) * kb;
  int bi[TV], bj[TV];
#pragma unroll
  for (int o = 0; o < TV; ++o) {
    const int t = threadIdx.x + o * 256;
    bi[o] = t < n_blk ? 4 * (t / kb) : -1;
    bj[o] = 4 * (t % kb);
  }
  for (int32_t s0 = beg; s0 < end; s0 += g.st) {
    const int ns = end - s0 < g.st ? end - s0 : g.st;
    __syncthreads();
    transr_stage_load(sx, sg, sr, g, d, k, s0, ns, ns, XS, GA, GR);
    __syncthreads();
    for (int q = 0; q < ns; ++q) {
#pragma unroll
      for (int o = 0; o < TV; ++o) {
        if (bi[o] < 0) continue;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const float4 x = *reinterpret_cast<const float4*>(sx + (size_t)(3 * q + v) * g.sxs + bi[o]);
          const float4 gg = *reinterpret_cast<const float4*>(sg + (size_t)(3 * q + v) * g.sgs + bj[o]);
          const float xs[4] = {x.x, x.y, x.z, x.w}, gs[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[o][a][c] = fmaf(xs[a], gs[c], acc[o][a][c]);
        }
      }
      if (threadIdx.x < k) racc += sr[(size_t)q * kr + threadIdx.x];
    }
  }
  const int dk = d * k;
  float* out = part + (size_t)bx * (dk + k);
#pragma unroll
  for (int o = 0; o < TV; ++o) {
    if (bi[o] < 0) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float4 v4;
      v4.x = acc[o][a][0]; v4.y = acc[o][a][1]; v4.z = acc[o][a][2]; v4.w = acc[o][a][3];
      *reinterpret_cast<float4*>(out + (size_t)(bi[o] + a) * k + bj[o]) = v4;
    }
  }
  if (threadIdx.x < k) out[dk + threadIdx.x] = racc;
}

// The same partial sums on the fp32 matrix pipe (d, k multiples of 16): the block's four wavefronts share the
// (d/16) x (k/16) output tiles round robin; v_mfma_f32_16x16x4_f32 takes four samples of one operand kind per
// instruction (A = x[sample][16 ti + i], B = g[sample][16 tj + j]: exact fp32 products, fp32 accumulate).  The whole
// chunk is staged in ONE round trip where it fits (64 samples at d = k = 64: 139 KB), its tail zero-filled to a
// multiple of 16 samples (zero rows add nothing); the operand pairs of four MFMA groups are requested from LDS together,
// ahead of their twelve instructions.  32 us (vector form, element-wise staging of 16 samples at a time) -> ~10.  The
// order of additions differs from the vector form's (four samples inside an instruction), fixed all the same.
typedef float trx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void transr_wgrad_mfma_body(
    float* s_dyn, const TrStageGeom g, int bx, int d, int k, int n_rel, const int32_t* __restrict__ seg,
    const float* __restrict__ XS, const float* __restrict__ GA, const float* __restrict__ GR,
    const int32_t* __restrict__ chunk_ptr, const int2* __restrict__ chunks, float* __restrict__ part, int ts = 0,
    int TS = 1) {
  if ((int32_t)bx >= chunk_ptr[n_rel]) return;
  const int2 ck = chunks[bx];
  const int32_t beg = ck.y, end = beg + kTrChunk < seg[ck.x + 1] ? beg + kTrChunk : seg[ck.x + 1];
  const int kr = (k + 3) / 4 * 4;
  float* const sx = s_dyn;
  float* const sg = sx + (size_t)g.st * 3 * g.sxs;
  float* const sr = sg + (size_t)g.st * 3 * g.sgs;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, lq = lane >> 4;
  // (a split block: the row tiles ti = ts, ts + TS, ... and every column tile)
  const int tk = k / 16, n_tiles = ((d / 16 - ts + TS - 1) / TS) * tk;
  constexpr int MAXT = (kTrMaxDim / 16) * (kTrMaxDim / 16) / 4;  // tiles per wavefront at the largest size
  trx4 acc[MAXT];
#pragma unroll
  for (int o = 0; o < MAXT; ++o) acc[o] = (trx4){0.f, 0.f, 0.f, 0.f};
  float racc = 0.f;
  for (int32_t s0 = beg; s0 < end; s0 += g.st) {
    const int ns = end - s0 < g.st ? end - s0 : g.st;
    const int ng = (ns + 3) >> 2;                    // groups of four samples
    const int fill = 4 * ng < g.st ? 4 * ng : g.st;  // (g.st is a multiple of 4)
    __syncthreads();
    transr_stage_load(sx, sg, sr, g, d, k, s0, ns, fill, XS, GA, GR, ts, TS);
    __syncthreads();
#pragma unroll
    for (int o = 0; o < MAXT; ++o) {
      const int t = w + 4 * o;
      if (t >= n_tiles) break;   // (wave-uniform)
      const int tl = t / tk, tj = t - tl * tk, ti = ts + TS * tl;
      const float* px = sx + (size_t)(3 * lq) * g.sxs + 16 * ti + li;
      const float* pg = sg + (size_t)(3 * lq) * g.sgs + 16 * tj + li;
      for (int gb = 0; gb < ng; gb += 4) {
        float av[4][3], bv[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            const int gq = gb + j < ng ? gb + j : ng - 1;   // (clamped: the products of a repeated group are skipped below)
            av[j][v] = px[(size_t)(12 * gq + v) * g.sxs];
            bv[j][v] = pg[(size_t)(12 * gq + v) * g.sgs];
          }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (gb + j < ng) {
#pragma unroll
            for (int v = 0; v < 3; ++v) acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][v], bv[j][v], acc[o], 0, 0, 0);
          }
      }
    }
    if (ts == 0 && threadIdx.x < k)
      for (int q = 0; q < ns; ++q) racc += sr[(size_t)q * kr + threadIdx.x];
  }
  const int dk = d * k;
  float* out = part + (size_t)bx * (dk + k);
#pragma unroll
  for (int o = 0; o < MAXT; ++o) {
    const int t = w + 4 * o;
    if (t >= n_tiles) break;
    const int tl = t / tk, tj = t - tl * tk, ti = ts + TS * tl;
    // acc[o][r] = sum over samples of x[16 ti + 4 lq + r] * g[16 tj + li]
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(16 * ti + 4 * lq + r) * k + 16 * tj + li] = acc[o][r];
  }
  if (ts == 0 && threadIdx.x < k) out[dk + threadIdx.x] = racc;
}

// ---- ordered reductions: block r < n_rel: dW[r] and drel[r] = sums of r's partial chunks (they are
// consecutive); block n_rel: the loss
__device__ __forceinline__ void transr_reduce_body(int bx, int by, int ny, int32_t batch, int d, int k, int n_rel,
                                                   const int32_t* __restrict__ chunk_ptr,
                                                   const float* __restrict__ part,
                                                   const float* __restrict__ losses,
                                                   float* __restrict__ grad_W, float* __restrict__ grad_rel,
                                                   float* __restrict__ loss,
                                                   const float* __restrict__ grad_scale) {
  const int r = bx;
  if (r == n_rel) {
    if (loss == nullptr || by != 0) return;
    __shared__ float s_l[256];
    float v = 0.f;
    for (int32_t s = threadIdx.x; s < batch; s += 256) v += losses[s];
    s_l[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) s_l[threadIdx.x] += s_l[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = s_l[0] / (float)batch;
    return;
  }
  if (grad_W == nullptr) return;
  const int dk = d * k;
  const int first = chunk_ptr[r], n_mine = chunk_ptr[r + 1] - first;
  const float sc = grad_scale ? grad_scale[0] : 1.f;   // the gradient arriving at the loss (device scalar)
  // gridDim.y blocks share a relation's dk + k elements (42 blocks alone left most of the chip idle: 22 us)
  for (int e = by * 256 + threadIdx.x; e < dk + k; e += 256 * ny) {
    float v = 0.f;
    for (int q = 0; q < n_mine; ++q) v += part[(size_t)(first + q) * (dk + k) + e];
    if (grad_scale) v *= sc;
    if (e < dk) grad_W[(size_t)r * dk + e] = v;
    else grad_rel[(size_t)r * k + (e - dk)] = v;
  }
}

// ---- entity gradient: rows of DX added into the (zeroed) dense gradient in sorted-id order;
// one 16-lane group per run of equal ids.  The run's length comes from 16 ids per look (one per
// lane), its rows are then added four at a time (their loads in flight together).
__device__ __forceinline__ void transr_scatter_body(int bx, int32_t n_rows, int d,
                                                    const int32_t* __restrict__ sorted_ids,
                                                    const int32_t* __restrict__ row_order,
                                                    const float* __restrict__ DX,
                                                    float* __restrict__ grad_ent,
                                                    const float* __restrict__ grad_scale) {
  const int sl = threadIdx.x & 15;
  const int32_t p = bx * 16 + (threadIdx.x >> 4);
  if (p >= n_rows) return;
  // ONE round trip tells the group everything about its run's first 64 positions: the id before it, the ids and the
  // row numbers of positions p .. p + 63 (lane sl takes positions p + sl + 16 j).  (Id, then the ids behind it, then
  // the row numbers, then the rows - four dependent round trips per look - made this a 26-us launch of 6,144 rows.)
  constexpr int LK = 4;   // looks of sixteen positions covered by the first round trip
  int32_t my_id[LK], my_row[LK];
#pragma unroll
  for (int j = 0; j < LK; ++j) {
    const int32_t qs = p + 16 * j + sl < n_rows ? p + 16 * j + sl : n_rows - 1;
    my_id[j] = sorted_ids[qs];
    my_row[j] = row_order[qs];
  }
  const int32_t prev = p > 0 ? sorted_ids[p - 1] : -1;
  // 16-lane ballots: the group's lanes are bits [16 g, 16 g + 16) of the wavefront mask
  const int gshift = (threadIdx.x & 48);
  const int32_t id = __shfl(my_id[0], 0, 16);
  if (p > 0 && prev == id) return;  // not the head of its run
  int32_t len = 0;
  bool open = true;
#pragma unroll
  for (int j = 0; j < LK; ++j) {
    const unsigned m = (unsigned)((__ballot(open && p + 16 * j + sl < n_rows && my_id[j] == id) >> gshift) & 0xFFFFull);
    if (open) {
      if (m == 0xFFFFu) len += 16;
      else { len += __builtin_ctz(~m); open = false; }
    }
  }
  while (open) {  // a run beyond the first 64 positions: sixteen more ids per step
    const int32_t q = p + len + sl;
    const bool same = q < n_rows && sorted_ids[q] == id;
    const unsigned m = (unsigned)((__ballot(same) >> gshift) & 0xFFFFull);
    if (m == 0xFFFFu) { len += 16; continue; }
    len += __builtin_ctz(~m);
    open = false;
  }
  // Lane sl owns floats 4 sl .. 4 sl + 3 (and + 64 at d > 64) of the row: one 16-byte load per row and half, SIXTEEN
  // rows in flight per look.  (Batches are drawn from the triples, so a hub entity heads dozens of samples of a batch:
  // four rows per look made its one lane group walk 49 rows in 13 dependent round trips - the launch's 26 us.)  The
  // rows are still added one after the other in sorted order: same bits.
  constexpr int HV = kTrMaxDim / 64;  // 16-byte pieces per lane and row at the largest width
  float4 acc[HV];
#pragma unroll
  for (int c = 0; c < HV; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int32_t q0 = 0; q0 < len; q0 += 16) {
    float4 v[16][HV];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int32_t qq = q0 + u < len ? q0 + u : len - 1;
      // the first 64 row numbers are in the group's registers already
      int32_t row;
      if (q0 < 16 * LK) {   // (uniform over the group)
        const int jj = q0 >> 4;
        const int32_t mine = jj == 0 ? my_row[0] : (jj == 1 ? my_row[1] : (jj == 2 ? my_row[2] : my_row[3]));
        row = __shfl(mine, qq & 15, 16);
      } else {
        row = row_order[p + qq];
      }
#pragma unroll
      for (int c = 0; c < HV; ++c) {
        const int i = 4 * sl + 64 * c;
        v[u][c] = i < d ? *reinterpret_cast<const float4*>(DX + (size_t)row * d + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (q0 + u < len)
#pragma unroll
        for (int c = 0; c < HV; ++c) {
          acc[c].x += v[u][c].x; acc[c].y += v[u][c].y; acc[c].z += v[u][c].z; acc[c].w += v[u][c].w;
        }
  }
  const float gs = grad_scale ? grad_scale[0] : 1.f;
#pragma unroll
  for (int c = 0; c < HV; ++c) {
    const int i = 4 * sl + 64 * c;
    if (i < d) {
      float4 o = acc[c];
      if (grad_scale) { o.x *= gs; o.y *= gs; o.z *= gs; o.w *= gs; }
      *reinterpret_cast<float4*>(grad_ent + (size_t)id * d + i) = o;
    }
  }
}

// The launches of the backward half.  The weight-gradient partials (or, when the forward half ran in an earlier call,
// the ordered reductions) and the entity-gradient scatter read only what the per-sample kernel and the sorts left:
// independent work, one launch - blocks [0, n_first) take the first job, the rest the scatter.
struct TrScatterArgs {
  int32_t n_rows;
  const int32_t *sorted_ids, *row_order;
  const float* DX;
  float* grad_ent;
  const float* grad_scale;
};

__global__ __launch_bounds__(256) void transr_wgrad_partial_kernel(
    int n_first, int d, int k, int n_rel, const int32_t* __restrict__ seg, const float* __restrict__ XS,
    const float* __restrict__ GA, const float* __restrict__ GR, const int32_t* __restrict__ chunk_ptr,
    const int2* __restrict__ chunks, float* __restrict__ part, TrStageGeom geom, TrScatterArgs sc) {
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // (the scatter's blocks do not touch it)
  if ((int)blockIdx.x < n_first) {
    if (d % 16 == 0 && k % 16 == 0) transr_wgrad_mfma_body(s_dyn, geom, (int)blockIdx.x, d, k, n_rel, seg, XS, GA, GR, chunk_ptr, chunks, part);
    else transr_wgrad_partial_body(s_dyn, geom, (int)blockIdx.x, d, k, n_rel, seg, XS, GA, GR, chunk_ptr, chunks, part);
  } else transr_scatter_body((int)blockIdx.x - n_first, sc.n_rows, d, sc.sorted_ids, sc.row_order, sc.DX, sc.grad_ent, sc.grad_scale);
}

__global__ __launch_bounds__(256) void transr_reduce_kernel(int n_first, int ny, int32_t batch, int d, int k, int n_rel,
                                                            const int32_t* __restrict__ chunk_ptr,
                                                            const float* __restrict__ part,
                                                            const float* __restrict__ losses,
                                                            float* __restrict__ grad_W, float* __restrict__ grad_rel,
                                                            float* __restrict__ loss,
                                                            const float* __restrict__ grad_scale, TrScatterArgs sc) {
  if ((int)blockIdx.x < n_first)
    transr_reduce_body((int)blockIdx.x / ny, (int)blockIdx.x % ny, ny, batch, d, k, n_rel, chunk_ptr, part, losses, grad_W,
                       grad_rel, loss, grad_scale);
  else transr_scatter_body((int)blockIdx.x - n_first, sc.n_rows, d, sc.sorted_ids, sc.row_order, sc.DX, sc.grad_ent, sc.grad_scale);
}


// ---------------------------------------------------------------------------------------------
// The KG phase as a sequence of three-launch iterations (round 6; reference kgat.py:116-136: sample a batch, transR,
// backward, optimizer.step, zero_grad - 1,641 times per epoch on the amazon-book shape, 78 % of the measured epoch).
//
// What changed against transr_run + kgat_adam_step_f32 (five launches, 0.1215 ms):
//  * the two one-workgroup sorts of a batch depend on nothing but the batch's ids.  A phase's batches are drawn up
//    front (the samplers are edge-uniform draws, dataset.py:234-323), so ALL of them are sorted by ONE launch
//    (transr_presort_kernel: two workgroups per batch, 3,282 of them on 256 CUs) instead of 13-18 us on an idle chip
//    in front of every iteration;
//  * no dense entity gradient, and no scatter launch.  The reference's optimiser is torch's dense Adam - every row of
//    the table moves in every step - but at most 3 B of the N rows have a non-zero gradient.  The per-sample kernel
//    writes its three gradient rows at their SORTED positions (the presort also emits the inverse permutation and the
//    run lengths), so the contributions of one entity are consecutive rows; workgroups riding in the same launch tag the
//    batch's entities in `row_slot` (one 64-bit word per node: call tag << 14 | first sorted position + 1; stale words
//    carry an older tag, so nothing is ever cleared).  The Adam launch streams p, m, v of every row and takes g = 0
//    unless the row's word carries this call's tag - then g = the sum of the run's rows, in sorted order (the order of
//    transr_scatter_body: same bits).  Gone: the 41 MB zero fill, the 41 MB gradient read, and the sorted-scatter
//    workgroups, which were the critical path of the middle launch (32 of its 33 us: 128 registers of rows in flight per
//    lane = one wavefront per SIMD; profiles/r06_kg_wgrad_ablation.txt).
//    Same arithmetic on the same values: the bits of torch.optim.Adam on the dense gradient;
//  * the ordered reduction of the weight-gradient partials happens where the sum is consumed: the Adam blocks of W_R
//    and of the relation table add a relation's partials in chunk order (the reduction launch's order) on the way.
// Launches per iteration: per-sample kernel (+ row tags) -> weight-gradient partials + loss -> Adam (+ gradient-row sums).

// per-batch block of the presort's output (offsets in bytes, 256-byte aligned)
struct TrSortedLayout {
  size_t order, seg, chunk_ptr, chunks, sorted_ids, row_order, inv_pos, run_len, bytes;
};
static TrSortedLayout transr_sorted_layout(int64_t batch, int n_rel) {
  const size_t b = (size_t)(batch > 0 ? batch : 1);
  const size_t n_part = b / kTrChunk + (size_t)(n_rel > 0 ? n_rel : 0) + 1;
  TrSortedLayout l;
  size_t w = 0;
  l.order = w; w += align_up(b * 4, 256);
  l.seg = w; w += align_up(((size_t)n_rel + 2) * 4, 256);
  l.chunk_ptr = w; w += align_up(((size_t)n_rel + 2) * 4, 256);
  l.chunks = w; w += align_up(n_part * 8, 256);
  l.sorted_ids = w; w += align_up(3 * b * 4, 256);
  l.row_order = w; w += align_up(3 * b * 4, 256);
  l.inv_pos = w; w += align_up(3 * b * 4, 256);
  l.run_len = w; w += align_up(3 * b * 4, 256);
  l.bytes = w;
  return l;
}

struct TrPresortArgs {
  int32_t n_batches, batch;
  int rel_bits, id_bits, n_rel;
  const int32_t *h, *r, *pos_t, *neg_t;   // n_batches x batch each
  unsigned char* sorted;
  TrSortedLayout lay;
};

template <typename PT>
__global__ __launch_bounds__(1024) void transr_presort_kernel(TrPresortArgs a) {
  const int32_t b = (int32_t)(blockIdx.x >> 1);
  if (b >= a.n_batches) return;
  unsigned char* blk = a.sorted + (size_t)b * a.lay.bytes;
  const size_t o = (size_t)b * a.batch;
  SortJob job;
  if ((blockIdx.x & 1) == 0) {   // samples by relation (+ the chunk table of the weight-gradient partials)
    job = SortJob{a.batch, a.rel_bits, a.r + o, nullptr, nullptr, reinterpret_cast<int32_t*>(blk + a.lay.order), nullptr,
                  (int32_t)a.n_rel, reinterpret_cast<int32_t*>(blk + a.lay.seg),
                  reinterpret_cast<int32_t*>(blk + a.lay.chunk_ptr), reinterpret_cast<int2*>(blk + a.lay.chunks)};
  } else {                       // the 3 B entity ids (head, positive tail, negative tail of every sample)
    job = SortJob{3 * a.batch, a.id_bits, a.h + o, a.pos_t + o, a.neg_t + o,
                  reinterpret_cast<int32_t*>(blk + a.lay.row_order), reinterpret_cast<int32_t*>(blk + a.lay.sorted_ids), 0,
                  nullptr, nullptr, nullptr,
                  reinterpret_cast<int32_t*>(blk + a.lay.inv_pos), reinterpret_cast<int32_t*>(blk + a.lay.run_len)};
  }
  small_sort_body<PT>(job);
}

// second launch of an iteration: blocks [0, n_part * n_split) the weight-gradient partials - a 64-sample chunk's
// d/16 x k/16 output tiles shared by n_split workgroups, each taking every n_split-th row tile (it stages those columns
// of x only: 60 instead of 112 KB, and a quarter of the MFMAs; the chunk's staging round trip and its MFMA loop were
// 9 + 8 us of a 22-us launch that 74 of 256 CUs took part in) -, the last block the loss
__global__ __launch_bounds__(256) void transr_wgrad_step_kernel(
    int n_part, int n_split, int32_t batch, int d, int k, int n_rel, const int32_t* __restrict__ seg,
    const float* __restrict__ XS, const float* __restrict__ GA, const float* __restrict__ GR,
    const int32_t* __restrict__ chunk_ptr, const int2* __restrict__ chunks, float* __restrict__ part,
    const float* __restrict__ losses, float* __restrict__ loss, TrStageGeom geom) {
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int bx = (int)blockIdx.x;
  if (bx < n_part * n_split) {
    if (d % 16 == 0 && k % 16 == 0)
      transr_wgrad_mfma_body(s_dyn, geom, bx / n_split, d, k, n_rel, seg, XS, GA, GR, chunk_ptr, chunks, part, bx % n_split,
                             n_split);
    else transr_wgrad_partial_body(s_dyn, geom, bx, d, k, n_rel, seg, XS, GA, GR, chunk_ptr, chunks, part);
  } else {
    transr_reduce_body(n_rel, 0, 1, batch, d, k, n_rel, chunk_ptr, part, losses, nullptr, nullptr, loss, nullptr);
  }
}

// third launch: Adam over the entity table (gradient through row_slot), W_R and the relation table (gradient = the
// ordered sum of the relation's partials).  4,096 elements per workgroup, 16 bytes per lane and stream, as adam_kernel.
struct TrAdamArgs {
  float *p[3], *m[3], *v[3];   // entity table, W_R, relation table
  int64_t n[3];
  int first_block[4];
  float step_size[3], bc2_sqrt[3];
  int d, k, dk;
  const unsigned long long* row_slot;
  unsigned long long tag;
  const float* DXs;          // the samples' gradient rows at their sorted positions
  const int32_t* run_len;    // run_len[p]: rows of the run that starts at sorted position p
  const float* part;
  const int32_t* chunk_ptr;
};

__global__ __launch_bounds__(256) void transr_adam_kernel(TrAdamArgs a, float w1, float beta2, float w2, float eps) {
  const int t = (int)blockIdx.x >= a.first_block[2] ? 2 : ((int)blockIdx.x >= a.first_block[1] ? 1 : 0);
  const int64_t base = (int64_t)((int)blockIdx.x - a.first_block[t]) * 4096;
  float* __restrict__ p = a.p[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const int64_t n = a.n[t];
  const float ss = a.step_size[t], bs = a.bc2_sqrt[t];
  const int stride = a.dk + a.k;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const int64_t i = base + (int64_t)q4 * 1024 + threadIdx.x * 4;
    if (i >= n) break;   // (n is a multiple of 4: d and k are)
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t == 0) {
      const int64_t row = i / a.d;
      const unsigned long long slot = a.row_slot[row];
      if ((slot >> kTrSlotBits) == a.tag) {
        // the row's gradient: its run of the sorted gradient rows, added in sorted order from 0 (transr_scatter_body's
        // order), four rows requested per look (a hub entity heads dozens of samples of an edge-uniform batch)
        const int64_t pr = (int64_t)(slot & ((1ull << kTrSlotBits) - 1ull)) - 1;
        const int len = a.run_len[pr];
        const float* src = a.DXs + pr * a.d + (i - row * a.d);
        for (int q = 0; q < len; q += 4) {
          float4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(src + (size_t)(q + u < len ? q + u : len - 1) * a.d);
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q + u < len) { gg.x += x[u].x; gg.y += x[u].y; gg.z += x[u].z; gg.w += x[u].w; }
        }
      }
    } else {
      const int width = t == 1 ? a.dk : a.k;
      const int64_t r = i / width;
      const int e = (int)(i - r * width) + (t == 1 ? 0 : a.dk);
      const int first = a.chunk_ptr[r], n_mine = a.chunk_ptr[r + 1] - first;
      // transr_reduce_body's order: v = 0; v += partial, chunk after chunk.  Four partials are requested together (a
      // load per iteration made the sum a chain of L2 round trips - ten for the most frequent relation at 32-sample
      // chunks - at the tail of the launch); the additions stay in chunk order.
      for (int q = 0; q < n_mine; q += 4) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int qq = q + u < n_mine ? q + u : n_mine - 1;
          x[u] = *reinterpret_cast<const float4*>(a.part + (size_t)(first + qq) * stride + e);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q + u < n_mine) { gg.x += x[u].x; gg.y += x[u].y; gg.z += x[u].z; gg.w += x[u].w; }
      }
    }
    float4 pp = *reinterpret_cast<const float4*>(p + i);
    float4 mm = *reinterpret_cast<const float4*>(m + i);
    float4 vv = *reinterpret_cast<const float4*>(v + i);
    adam_one(pp.x, gg.x, mm.x, vv.x, w1, beta2, w2, ss, bs, eps);
    adam_one(pp.y, gg.y, mm.y, vv.y, w1, beta2, w2, ss, bs, eps);
    adam_one(pp.z, gg.z, mm.z, vv.z, w1, beta2, w2, ss, bs, eps);
    adam_one(pp.w, gg.w, mm.w, vv.w, w1, beta2, w2, ss, bs, eps);
    *reinterpret_cast<float4*>(p + i) = pp;
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(v + i) = vv;
  }
}

static size_t transr_step_workspace_bytes(int64_t batch, int d, int k, int n_rel) {
  const size_t b = (size_t)(batch > 0 ? batch : 1);
  const size_t n_part = b / kTrChunk + (size_t)(n_rel > 0 ? n_rel : 0) + 1;
  size_t w = 0;
  w += align_up(b * 4, 256);                             // losses
  w += align_up(3 * b * (size_t)k * 4, 256);             // GA
  w += align_up(b * (size_t)k * 4, 256);                 // GR
  w += 2 * align_up(3 * b * (size_t)d * 4, 256);         // DX (rows at their sorted positions), XS
  w += align_up(n_part * ((size_t)d * k + k) * 4, 256);  // W / relation gradient partials
  return w;
}

}  // namespace kgat

using namespace kgat;

extern "C" {

int kgat_transr_supported(int64_t n_nodes, int d, int k, int n_rel, int64_t batch) {
  return d > 0 && k > 0 && d % 4 == 0 && k % 4 == 0 && d <= kTrMaxDim && k <= kTrMaxDim && n_rel > 0 && n_rel <= kTrMaxRel && batch > 0 &&
         3 * batch <= kTrSmallSort && n_nodes > 0 && n_nodes < INT32_MAX;
}

size_t kgat_transr_workspace_bytes(int64_t batch, int d, int k, int n_rel) {
  const size_t b = (size_t)(batch > 0 ? batch : 1);
  const size_t n_part = b / kTrChunk + (size_t)(n_rel > 0 ? n_rel : 0) + 1;
  size_t w = 0;
  w += align_up(b * 4, 256);                       // order (relation-sorted samples)
  w += 2 * align_up(((size_t)n_rel + 2) * 4, 256); // seg, chunk_ptr
  w += align_up(n_part * 8, 256);                  // chunks
  w += align_up(b * 4, 256);                       // losses
  w += align_up(3 * b * (size_t)k * 4, 256);       // GA
  w += align_up(b * (size_t)k * 4, 256);           // GR
  w += 2 * align_up(3 * b * (size_t)d * 4, 256);   // DX, XS
  w += align_up(n_part * ((size_t)d * k + k) * 4, 256);  // W / relation gradient partials
  w += 2 * align_up(3 * b * 4, 256);               // sorted ids, row order
  return w;
}

// stage bits of transr_run: FORWARD = relation sort + per-sample kernel (+ the weight-gradient partials) + the loss;
// BACKWARD = the ordered reductions into grad_W / grad_rel and the sorted scatter into grad_ent, scaled by
// grad_scale[0] when given.  The workspace carries the per-sample rows from one to the other.
enum { kTrForward = 1, kTrBackward = 2 };

static int transr_run(int stage, int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h,
                      const int32_t* r, const int32_t* pos_t, const int32_t* neg_t, const float* ent,
                      const float* W_R, const float* rel, float reg_lambda, float* loss, const float* grad_scale,
                      float* grad_ent, float* grad_W, float* grad_rel, bool want_grad, void* workspace,
                      size_t workspace_bytes, kgat_stream_t stream) {
  if (!kgat_transr_supported(n_nodes, d, k, n_rel, batch)) {
    set_error("transr: needs d, k multiples of 4 and <= %d, batch <= %d, n_nodes < 2^31 (d=%d k=%d batch=%lld n_nodes=%lld)", kTrMaxDim,
              kTrSmallSort / 3, d, k, (long long)batch, (long long)n_nodes);
    return KGAT_E_UNSUPPORTED;
  }
  KGAT_CHECK_ARG(h && r && pos_t && neg_t && workspace, "transr: null pointer");
  KGAT_CHECK_ARG(!(stage & kTrForward) || (ent && W_R && rel && loss), "transr: null pointer");
  KGAT_CHECK_ARG(!(stage & kTrBackward) || !want_grad || (grad_ent && grad_W && grad_rel),
                 "transr: all three gradients or none");
  if (workspace_bytes < kgat_transr_workspace_bytes(batch, d, k, n_rel)) {
    set_error("transr: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int32_t B = (int32_t)batch;
  const int n_part = B / kTrChunk + n_rel + 1;
  Carver cv(workspace);
  int32_t* order = cv.take<int32_t>((size_t)B);
  int32_t* seg = cv.take<int32_t>((size_t)n_rel + 2);
  int32_t* chunk_ptr = cv.take<int32_t>((size_t)n_rel + 2);
  int2* chunks = cv.take<int2>((size_t)n_part);
  float* losses = cv.take<float>((size_t)B);
  float* GA = cv.take<float>((size_t)3 * B * k);
  float* GR = cv.take<float>((size_t)B * k);
  float* DX = cv.take<float>((size_t)3 * B * d);
  float* XS = cv.take<float>((size_t)3 * B * d);
  float* part = cv.take<float>((size_t)n_part * ((size_t)d * k + k));
  int32_t* sorted_ids = cv.take<int32_t>((size_t)3 * B);
  int32_t* row_order = cv.take<int32_t>((size_t)3 * B);

  int rel_bits = 1, id_bits = 1;
  while ((1 << rel_bits) < n_rel) ++rel_bits;
  while ((1ll << id_bits) < n_nodes) ++id_bits;
  const bool fwd = (stage & kTrForward) != 0, bwd = (stage & kTrBackward) != 0 && want_grad;
  const SortJob rel_job = {B, rel_bits, r, nullptr, nullptr, order, nullptr, (int32_t)n_rel, seg, chunk_ptr, chunks};
  const SortJob id_job = {3 * B, id_bits, h, pos_t, neg_t, row_order, sorted_ids, 0, nullptr, nullptr, nullptr};
  const TrScatterArgs sc = {3 * B, sorted_ids, row_order, DX, grad_ent, grad_scale};
  const TrScatterArgs no_sc = {0, nullptr, nullptr, nullptr, nullptr, nullptr};
  const unsigned scatter_blocks = (unsigned)((3 * B + 15) / 16);
  const int64_t n_zero = (int64_t)n_nodes * d;
  // first launch: the sort(s) this call needs + the zero fill of the dense entity gradient
  {
    const SortJob& first = fwd ? rel_job : id_job;
    const int n_jobs = (fwd && bwd) ? 2 : ((fwd || bwd) ? 1 : 0);
    if (n_jobs == 0) return KGAT_OK;
    const unsigned zero_blocks = bwd ? 2u * (unsigned)device_cu_count() : 0u;
    if (bwd && id_bits > 19)  // entity ids beyond 2^19: the 64-bit packing (128 KB of LDS, one more radix pass per 8 id bits)
      hipLaunchKernelGGL(small_sort_kernel<uint64_t>, dim3(n_jobs + zero_blocks), dim3(1024), 0, st, first, id_job, n_jobs,
                         grad_ent, n_zero);
    else
      hipLaunchKernelGGL(small_sort_kernel<uint32_t>, dim3(n_jobs + zero_blocks), dim3(1024), 0, st, first, id_job, n_jobs,
                         bwd ? grad_ent : (float*)nullptr, bwd ? n_zero : (int64_t)0);
    KGAT_CHECK_LAUNCH("transr_sort");
  }
  if (fwd) {
    const unsigned sb = (unsigned)((B + 3) / 4);
    if (!want_grad) {
      if (d <= kTrWLds && k <= kTrWLds)
        hipLaunchKernelGGL((transr_sample_kernel<false, true>), dim3(sb), dim3(256), 0, st, B, d, k, (const int32_t*)order, h,
                           r, pos_t, neg_t, ent, W_R, rel, reg_lambda, losses, GA, GR, DX, XS);
      else
        hipLaunchKernelGGL((transr_sample_kernel<false, false>), dim3(sb), dim3(256), 0, st, B, d, k, (const int32_t*)order, h,
                           r, pos_t, neg_t, ent, W_R, rel, reg_lambda, losses, GA, GR, DX, XS);
      KGAT_CHECK_LAUNCH("transr_sample");
      hipLaunchKernelGGL(transr_reduce_kernel, dim3(1), dim3(256), 0, st, 1, 1, B, d, k, 0, (const int32_t*)chunk_ptr,
                         (const float*)part, (const float*)losses, (float*)nullptr, (float*)nullptr, loss,
                         (const float*)nullptr, no_sc);
      KGAT_CHECK_LAUNCH("transr_reduce");
      return KGAT_OK;
    }
    if (d <= kTrWLds && k <= kTrWLds)
      hipLaunchKernelGGL((transr_sample_kernel<true, true>), dim3(sb), dim3(256), 0, st, B, d, k, (const int32_t*)order, h, r,
                         pos_t, neg_t, ent, W_R, rel, reg_lambda, losses, GA, GR, DX, XS);
    else
      hipLaunchKernelGGL((transr_sample_kernel<true, false>), dim3(sb), dim3(256), 0, st, B, d, k, (const int32_t*)order, h, r,
                         pos_t, neg_t, ent, W_R, rel, reg_lambda, losses, GA, GR, DX, XS);
    KGAT_CHECK_LAUNCH("transr_sample");
    // the weight-gradient partials, and beside them (whole step in one call) the entity-gradient scatter
    const size_t lds = transr_stage_bytes(d, k);
    // (set on every call: the attribute is per function AND per device, and a process-wide "already set" flag is
    //  neither - ADVICE round 5; the call is a host-side table write)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(transr_wgrad_partial_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("transr: cannot reserve %zu bytes of LDS", lds);
      return KGAT_E_HIP;
    }
    hipLaunchKernelGGL(transr_wgrad_partial_kernel, dim3((unsigned)n_part + (bwd ? scatter_blocks : 0u)), dim3(256), lds, st,
                       n_part, d, k, n_rel, (const int32_t*)seg, (const float*)XS, (const float*)GA, (const float*)GR,
                       (const int32_t*)chunk_ptr, (const int2*)chunks, part, transr_stage_geom(d, k), bwd ? sc : no_sc);
    KGAT_CHECK_LAUNCH("transr_wgrad_partial");
    if (!bwd) {  // the loss alone now (block n_rel of the reduction)
      hipLaunchKernelGGL(transr_reduce_kernel, dim3(1), dim3(256), 0, st, 1, 1, B, d, k, 0, (const int32_t*)chunk_ptr,
                         (const float*)part, (const float*)losses, (float*)nullptr, (float*)nullptr, loss,
                         (const float*)nullptr, no_sc);
      KGAT_CHECK_LAUNCH("transr_reduce");
      return KGAT_OK;
    }
  }
  if (!bwd) return KGAT_OK;
  // the ordered reductions into grad_W / grad_rel (+ the loss when the forward half ran here); in a backward-only call
  // the scatter rides along
  constexpr int kNy = 8;
  const unsigned red_blocks = ((unsigned)n_rel + 1) * kNy;
  hipLaunchKernelGGL(transr_reduce_kernel, dim3(red_blocks + (fwd ? 0u : scatter_blocks)), dim3(256), 0, st, (int)red_blocks,
                     kNy, B, d, k, n_rel, (const int32_t*)chunk_ptr, (const float*)part, (const float*)losses, grad_W,
                     grad_rel, fwd ? loss : (float*)nullptr, grad_scale, fwd ? no_sc : sc);
  KGAT_CHECK_LAUNCH("transr_reduce");
  return KGAT_OK;
}

int kgat_transr_loss_grad_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h,
                              const int32_t* r, const int32_t* pos_t, const int32_t* neg_t, const float* ent,
                              const float* W_R, const float* rel, float reg_lambda, float* loss, float* grad_ent,
                              float* grad_W, float* grad_rel, void* workspace, size_t workspace_bytes,
                              kgat_stream_t stream) {
  const bool want_grad = grad_ent || grad_W || grad_rel;
  KGAT_CHECK_ARG(!want_grad || (grad_ent && grad_W && grad_rel), "transr: all three gradients or none");
  return transr_run(kTrForward | kTrBackward, n_nodes, n_rel, d, k, batch, h, r, pos_t, neg_t, ent, W_R, rel, reg_lambda,
                    loss, nullptr, grad_ent, grad_W, grad_rel, want_grad, workspace, workspace_bytes, stream);
}

int kgat_transr_forward_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h, const int32_t* r,
                            const int32_t* pos_t, const int32_t* neg_t, const float* ent, const float* W_R,
                            const float* rel, float reg_lambda, float* loss, void* workspace, size_t workspace_bytes,
                            kgat_stream_t stream) {
  return transr_run(kTrForward, n_nodes, n_rel, d, k, batch, h, r, pos_t, neg_t, ent, W_R, rel, reg_lambda, loss, nullptr,
                    nullptr, nullptr, nullptr, true, workspace, workspace_bytes, stream);
}

int kgat_transr_backward_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h, const int32_t* r,
                             const int32_t* pos_t, const int32_t* neg_t, const float* grad_scale, float* grad_ent,
                             float* grad_W, float* grad_rel, void* workspace, size_t workspace_bytes,
                             kgat_stream_t stream) {
  KGAT_CHECK_ARG(grad_ent && grad_W && grad_rel, "transr_backward: null gradient pointer");
  return transr_run(kTrBackward, n_nodes, n_rel, d, k, batch, h, r, pos_t, neg_t, nullptr, nullptr, nullptr, 0.f, nullptr,
                    grad_scale, grad_ent, grad_W, grad_rel, true, workspace, workspace_bytes, stream);
}


size_t kgat_transr_sorted_bytes(int64_t batch, int n_rel) { return transr_sorted_layout(batch, n_rel).bytes; }

int kgat_transr_presort_f32(int64_t n_nodes, int n_rel, int64_t n_batches, int64_t batch, const int32_t* h,
                            const int32_t* r, const int32_t* pos_t, const int32_t* neg_t, void* sorted,
                            size_t sorted_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_batches >= 0 && n_batches < (1 << 30), "transr_presort: bad batch count");
  if (n_batches == 0) return KGAT_OK;
  if (!kgat_transr_supported(n_nodes, 4, 4, n_rel, batch)) {
    set_error("transr_presort: needs batch <= %d, n_rel <= %d, n_nodes < 2^31", kTrSmallSort / 3, kTrMaxRel);
    return KGAT_E_UNSUPPORTED;
  }
  KGAT_CHECK_ARG(h && r && pos_t && neg_t && sorted, "transr_presort: null pointer");
  const TrSortedLayout lay = transr_sorted_layout(batch, n_rel);
  if (sorted_bytes < (size_t)n_batches * lay.bytes) {
    set_error("transr_presort: output buffer too small");
    return KGAT_E_WORKSPACE;
  }
  int rel_bits = 1, id_bits = 1;
  while ((1 << rel_bits) < n_rel) ++rel_bits;
  while ((1ll << id_bits) < n_nodes) ++id_bits;
  const TrPresortArgs a = {(int32_t)n_batches, (int32_t)batch, rel_bits, id_bits, n_rel, h, r, pos_t, neg_t,
                           static_cast<unsigned char*>(sorted), lay};
  if (id_bits > 19)
    hipLaunchKernelGGL(transr_presort_kernel<uint64_t>, dim3((unsigned)(2 * n_batches)), dim3(1024), 0, as_stream(stream), a);
  else
    hipLaunchKernelGGL(transr_presort_kernel<uint32_t>, dim3((unsigned)(2 * n_batches)), dim3(1024), 0, as_stream(stream), a);
  KGAT_CHECK_LAUNCH("transr_presort");
  return KGAT_OK;
}

size_t kgat_transr_step_workspace_bytes(int64_t batch, int d, int k, int n_rel) {
  return transr_step_workspace_bytes(batch, d, k, n_rel);
}

int kgat_transr_adam_step_f32(int64_t n_nodes, int n_rel, int d, int k, int64_t batch, const int32_t* h, const int32_t* r,
                              const int32_t* pos_t, const int32_t* neg_t, const void* sorted, float* ent, float* W_R,
                              float* rel, float* const* exp_avg_host, float* const* exp_avg_sq_host,
                              const int64_t* steps_host, double lr, double beta1, double beta2, double eps,
                              float reg_lambda, float* loss, uint64_t* row_slot, uint64_t tag, void* workspace,
                              size_t workspace_bytes, kgat_stream_t stream) {
  if (!kgat_transr_supported(n_nodes, d, k, n_rel, batch)) {
    set_error("transr_adam_step: needs d, k multiples of 4 and <= %d, batch <= %d, n_nodes < 2^31 (d=%d k=%d batch=%lld n_nodes=%lld)",
              kTrMaxDim, kTrSmallSort / 3, d, k, (long long)batch, (long long)n_nodes);
    return KGAT_E_UNSUPPORTED;
  }
  KGAT_CHECK_ARG(h && r && pos_t && neg_t && sorted && ent && W_R && rel && exp_avg_host && exp_avg_sq_host && steps_host &&
                     loss && row_slot && workspace, "transr_adam_step: null pointer");
  KGAT_CHECK_ARG(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "transr_adam_step: bad hyperparameter");
  KGAT_CHECK_ARG(tag >= 1 && tag < (1ull << (64 - kTrSlotBits)), "transr_adam_step: tag must be in [1, 2^50)");
  float* const ps[3] = {ent, W_R, rel};
  for (int t = 0; t < 3; ++t) {
    KGAT_CHECK_ARG(exp_avg_host[t] && exp_avg_sq_host[t] && steps_host[t] >= 1, "transr_adam_step: tensor %d: bad state", t);
    KGAT_CHECK_ARG(((reinterpret_cast<uintptr_t>(ps[t]) | reinterpret_cast<uintptr_t>(exp_avg_host[t]) |
                     reinterpret_cast<uintptr_t>(exp_avg_sq_host[t])) & 15) == 0,
                   "transr_adam_step: tensor %d: parameters and moments must be 16-byte aligned", t);
  }
  if (workspace_bytes < transr_step_workspace_bytes(batch, d, k, n_rel)) {
    set_error("transr_adam_step: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int32_t B = (int32_t)batch;
  const int n_part = B / kTrChunk + n_rel + 1;
  const TrSortedLayout lay = transr_sorted_layout(batch, n_rel);
  const unsigned char* blk = static_cast<const unsigned char*>(sorted);
  const int32_t* order = reinterpret_cast<const int32_t*>(blk + lay.order);
  const int32_t* seg = reinterpret_cast<const int32_t*>(blk + lay.seg);
  const int32_t* chunk_ptr = reinterpret_cast<const int32_t*>(blk + lay.chunk_ptr);
  const int2* chunks = reinterpret_cast<const int2*>(blk + lay.chunks);
  const int32_t* sorted_ids = reinterpret_cast<const int32_t*>(blk + lay.sorted_ids);
  const int32_t* inv_pos = reinterpret_cast<const int32_t*>(blk + lay.inv_pos);
  const int32_t* run_len = reinterpret_cast<const int32_t*>(blk + lay.run_len);
  Carver cv(workspace);
  float* losses = cv.take<float>((size_t)B);
  float* GA = cv.take<float>((size_t)3 * B * k);
  float* GR = cv.take<float>((size_t)B * k);
  float* DX = cv.take<float>((size_t)3 * B * d);
  float* XS = cv.take<float>((size_t)3 * B * d);
  float* part = cv.take<float>((size_t)n_part * ((size_t)d * k + k));

  const unsigned sb = (unsigned)((B + 3) / 4), mark_blocks = (unsigned)((3 * B + 255) / 256);
  const TrStepMarks mk = {inv_pos, sorted_ids, run_len, reinterpret_cast<unsigned long long*>(row_slot),
                          (unsigned long long)tag, (int)sb};
  if (d <= kTrWLds && k <= kTrWLds)
    hipLaunchKernelGGL((transr_sample_kernel<true, true>), dim3(sb + mark_blocks), dim3(256), 0, st, B, d, k, order, h, r,
                       pos_t, neg_t, (const float*)ent, (const float*)W_R, (const float*)rel, reg_lambda, losses, GA, GR, DX,
                       XS, mk);
  else
    hipLaunchKernelGGL((transr_sample_kernel<true, false>), dim3(sb + mark_blocks), dim3(256), 0, st, B, d, k, order, h, r,
                       pos_t, neg_t, (const float*)ent, (const float*)W_R, (const float*)rel, reg_lambda, losses, GA, GR, DX,
                       XS, mk);
  KGAT_CHECK_LAUNCH("transr_sample");
  const size_t lds = transr_stage_bytes(d, k);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(transr_wgrad_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess) {
    set_error("transr_adam_step: cannot reserve %zu bytes of LDS", lds);
    return KGAT_E_HIP;
  }
  const int n_split = (d % 16 == 0 && k % 16 == 0) ? (d / 16 < 4 ? d / 16 : 4) : 1;
  hipLaunchKernelGGL(transr_wgrad_step_kernel, dim3((unsigned)(n_part * n_split + 1)), dim3(256), lds, st, n_part, n_split, B,
                     d, k, n_rel, seg, (const float*)XS, (const float*)GA, (const float*)GR, chunk_ptr, chunks, part,
                     (const float*)losses, loss, transr_stage_geom(d, k));
  KGAT_CHECK_LAUNCH("transr_wgrad_step");
  TrAdamArgs a;
  const int64_t sizes[3] = {(int64_t)n_nodes * d, (int64_t)n_rel * d * k, (int64_t)n_rel * k};
  int blocks = 0;
  for (int t = 0; t < 3; ++t) {
    a.p[t] = ps[t]; a.m[t] = exp_avg_host[t]; a.v[t] = exp_avg_sq_host[t];
    a.n[t] = sizes[t];
    a.first_block[t] = blocks;
    const int64_t nb = (sizes[t] + 4095) / 4096;
    KGAT_CHECK_ARG(nb + blocks < (int64_t)1 << 31, "transr_adam_step: too many elements");
    blocks += (int)nb;
    // torch.optim.adam: python floats (double), then fp32 in the kernels (as kgat_adam_step_f32)
    a.step_size[t] = (float)(lr / (1.0 - pow(beta1, (double)steps_host[t])));
    a.bc2_sqrt[t] = (float)sqrt(1.0 - pow(beta2, (double)steps_host[t]));
  }
  a.first_block[3] = blocks;
  a.d = d; a.k = k; a.dk = d * k;
  a.row_slot = reinterpret_cast<const unsigned long long*>(row_slot);
  a.tag = (unsigned long long)tag;
  a.DXs = DX; a.run_len = run_len; a.part = part; a.chunk_ptr = chunk_ptr;
  hipLaunchKernelGGL(transr_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps);
  KGAT_CHECK_LAUNCH("transr_adam");
  return KGAT_OK;
}
}  // extern "C"
