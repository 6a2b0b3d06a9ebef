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

#include "kgat_common.h"

namespace kgat {

constexpr int kTrMaxDim = 128;    // d, k <= 128
constexpr int kTrChunk = 64;      // samples per partial W-gradient block
constexpr int kTrSmallSort = 8192;
constexpr int kTrMaxRel = 4096;   // relations (chunk table built by one workgroup, 4 keys per thread)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// ---- one-workgroup stable sort of n <= 8192 keys (< 2^19), payload = position in the input,
// packed as key << 13 | position.  LSD radix sort, 8-bit digits, both buffers and the counters in
// LDS: each of the 16 wavefronts owns a contiguous slice of the current order and its own column
// of the 256 x 16 counter table, counts its digits (LDS atomics - counts do not depend on order),
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
  __shared__ PT s_buf[2][kTrSmallSort];
  __shared__ int32_t s_cnt[256 * kSsWaves];
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
  const int passes = (key_bits + 7) / 8 > 0 ? (key_bits + 7) / 8 : 1;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 13 + 8 * pass;
    for (int i = tid; i < 256 * kSsWaves; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    for (int32_t i = lo + lane; i < hi; i += kWave) atomicAdd(&s_cnt[(uint32_t)((s_buf[cur][i] >> shift) & 255u) * kSsWaves + w], 1);
    __syncthreads();
    {  // exclusive scan of the 4096 counters, 4 per thread
      int32_t c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = s_cnt[4 * tid + q];
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
        s_cnt[4 * tid + q] = base;
        base += c[q];
      }
    }
    __syncthreads();
    for (int32_t i0 = lo; i0 < hi; i0 += kWave) {
      const int32_t i = i0 + lane;
      const bool valid = i < hi;
      const PT v = valid ? s_buf[cur][i] : (PT)0;
      const uint32_t dgt = (uint32_t)((v >> shift) & 255u);
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
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
template <bool GRAD>
__global__ __launch_bounds__(256) void transr_sample_kernel(
    int32_t batch, int d, int k, const int32_t* __restrict__ order, const int32_t* __restrict__ h,
    const int32_t* __restrict__ r, const int32_t* __restrict__ pt, const int32_t* __restrict__ nt,
    const float* __restrict__ ent, const float* __restrict__ W_R, const float* __restrict__ rel, float lambda,
    float* __restrict__ losses, float* __restrict__ GA, float* __restrict__ GR, float* __restrict__ DX,
    float* __restrict__ XS) {
  __shared__ float s_x[256 / kWave][3][kTrMaxDim];
  const int lane = threadIdx.x % kWave, wv = threadIdx.x / kWave;
  const int32_t s = blockIdx.x * (256 / kWave) + wv;
  if (s >= batch) return;
  const int32_t b = order[s];
  const int32_t rr = r[b];
  const int32_t ids[3] = {h[b], pt[b], nt[b]};
  const float* W = W_R + (size_t)rr * d * k;
  float(*sx)[kTrMaxDim] = s_x[wv];
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
  for (int i = lane; i < d; i += kWave) {
    const float* wr = W + (size_t)i * k;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    for (int j = 0; j < k; ++j) {
      const float w = wr[j];
      x0 = fmaf(sx[0][j], w, x0);
      x1 = fmaf(sx[1][j], w, x1);
      x2 = fmaf(sx[2][j], w, x2);
    }
    DX[((size_t)3 * b + 0) * d + i] = x0;
    DX[((size_t)3 * b + 1) * d + i] = x1;
    DX[((size_t)3 * b + 2) * d + i] = x2;
  }
}

// ---- W_R gradient: partial sums of x^T ga per chunk of <= 64 relation-sorted samples
constexpr int kTrStage = 16;  // samples staged through LDS per step of the partial kernel

// Thread t owns 4 x 4 blocks of the d x k outer-product sum: block index t, t + 256, ... over
// (d/4) x (k/4) blocks, so a sample costs two 16-byte LDS reads per 16 fmas.  d, k multiples of 4.
__device__ __forceinline__ void transr_wgrad_partial_body(
    int bx, int d, int k, int n_rel, const int32_t* __restrict__ seg, const float* __restrict__ XS,
    const float* __restrict__ GA, const float* __restrict__ GR, const int32_t* __restrict__ chunk_ptr,
    const int2* __restrict__ chunks, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float s_x[kTrStage][3][kTrMaxDim];
  __shared__ __attribute__((aligned(16))) float s_g[kTrStage][3][kTrMaxDim];
  __shared__ float s_r[kTrStage][kTrMaxDim];
  if ((int32_t)bx >= chunk_ptr[n_rel]) return;
  const int2 ck = chunks[bx];
  const int32_t beg = ck.y, end = beg + kTrChunk < seg[ck.x + 1] ? beg + kTrChunk : seg[ck.x + 1];
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
  for (int32_t s0 = beg; s0 < end; s0 += kTrStage) {
    const int ns = end - s0 < kTrStage ? end - s0 : kTrStage;
    __syncthreads();
    for (int t = threadIdx.x; t < ns * 3 * d; t += 256)
      s_x[t / (3 * d)][(t / d) % 3][t % d] = XS[(size_t)3 * s0 * d + t];
    for (int t = threadIdx.x; t < ns * 3 * k; t += 256)
      s_g[t / (3 * k)][(t / k) % 3][t % k] = GA[(size_t)3 * s0 * k + t];
    for (int t = threadIdx.x; t < ns * k; t += 256) s_r[t / k][t % k] = GR[(size_t)s0 * k + t];
    __syncthreads();
    for (int q = 0; q < ns; ++q) {
#pragma unroll
      for (int o = 0; o < TV; ++o) {
        if (bi[o] < 0) continue;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const float4 x = *reinterpret_cast<const float4*>(&s_x[q][v][bi[o]]);
          const float4 g = *reinterpret_cast<const float4*>(&s_g[q][v][bj[o]]);
          const float xs[4] = {x.x, x.y, x.z, x.w}, gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[o][a][c] = fmaf(xs[a], gs[c], acc[o][a][c]);
        }
      }
      if (threadIdx.x < k) racc += s_r[q][threadIdx.x];
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
  const int32_t id = sorted_ids[p];
  if (p > 0 && sorted_ids[p - 1] == id) return;  // not the head of its run
  // 16-lane ballots: the group's lanes are bits [16 g, 16 g + 16) of the wavefront mask
  const int gshift = (threadIdx.x & 48);
  int32_t len = 0;
  while (true) {
    const int32_t q = p + len + sl;
    const bool same = q < n_rows && sorted_ids[q] == id;
    const unsigned m = (unsigned)((__ballot(same) >> gshift) & 0xFFFFull);
    if (m == 0xFFFFu) { len += 16; continue; }
    len += __builtin_ctz(~m);
    break;
  }
  float acc[kTrMaxDim / 16];
#pragma unroll
  for (int c = 0; c < kTrMaxDim / 16; ++c) acc[c] = 0.f;
  for (int32_t q0 = 0; q0 < len; q0 += 4) {
    int32_t rows[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rows[u] = row_order[p + (q0 + u < len ? q0 + u : len - 1)];
    float v[4][kTrMaxDim / 16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < kTrMaxDim / 16; ++c) {
        const int i = sl + 16 * c;
        v[u][c] = i < d ? DX[(size_t)rows[u] * d + i] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q0 + u < len)
#pragma unroll
        for (int c = 0; c < kTrMaxDim / 16; ++c) acc[c] += v[u][c];
  }
#pragma unroll
  for (int c = 0; c < kTrMaxDim / 16; ++c) {
    const int i = sl + 16 * c;
    if (i < d) grad_ent[(size_t)id * d + i] = grad_scale ? acc[c] * grad_scale[0] : acc[c];
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
    const int2* __restrict__ chunks, float* __restrict__ part, TrScatterArgs sc) {
  if ((int)blockIdx.x < n_first) transr_wgrad_partial_body((int)blockIdx.x, d, k, n_rel, seg, XS, GA, GR, chunk_ptr, chunks, part);
  else transr_scatter_body((int)blockIdx.x - n_first, sc.n_rows, d, sc.sorted_ids, sc.row_order, sc.DX, sc.grad_ent, sc.grad_scale);
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
      hipLaunchKernelGGL(transr_sample_kernel<false>, dim3(sb), dim3(256), 0, st, B, d, k, (const int32_t*)order, h, r,
                         pos_t, neg_t, ent, W_R, rel, reg_lambda, losses, GA, GR, DX, XS);
      KGAT_CHECK_LAUNCH("transr_sample");
      hipLaunchKernelGGL(transr_reduce_kernel, dim3(1), dim3(256), 0, st, 1, 1, B, d, k, 0, (const int32_t*)chunk_ptr,
                         (const float*)part, (const float*)losses, (float*)nullptr, (float*)nullptr, loss,
                         (const float*)nullptr, no_sc);
      KGAT_CHECK_LAUNCH("transr_reduce");
      return KGAT_OK;
    }
    hipLaunchKernelGGL(transr_sample_kernel<true>, dim3(sb), dim3(256), 0, st, B, d, k, (const int32_t*)order, h, r,
                       pos_t, neg_t, ent, W_R, rel, reg_lambda, losses, GA, GR, DX, XS);
    KGAT_CHECK_LAUNCH("transr_sample");
    // the weight-gradient partials, and beside them (whole step in one call) the entity-gradient scatter
    hipLaunchKernelGGL(transr_wgrad_partial_kernel, dim3((unsigned)n_part + (bwd ? scatter_blocks : 0u)), dim3(256), 0, st,
                       n_part, d, k, n_rel, (const int32_t*)seg, (const float*)XS, (const float*)GA, (const float*)GR,
                       (const int32_t*)chunk_ptr, (const int2*)chunks, part, bwd ? sc : no_sc);
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

}  // extern "C"
