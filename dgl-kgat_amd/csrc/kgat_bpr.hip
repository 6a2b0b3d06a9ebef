// BPR loss of the CF phase (reference models.py:170-178, get_loss; _L2_loss_mean :9-11) and its gradient with
// respect to the readout, for gfx950.  The reference's expression is ~30 torch operators forward and ~45 backward
// (gathers, batched dot products, logsigmoid, three regularisers, an index_put with a device sort): 75 launches of
// a few microseconds each, 0.7 ms of the 1.6 ms CF step, all of it launch latency.  Here:
//   forward   bpr_sample_kernel (one wavefront per sample: the three rows, five dot products) + bpr_reduce_kernel
//   backward  four launches of bounded work: bpr_sort_zero_kernel (slices of 4,096 row ids sorted in LDS beside
//             three quarters of the zero fill of the N x F gradient) + bpr_merge_kernel (every id's place among all
//             ids: a stable merge by binary searches advanced together, beside the rest of the fill; up to 65,536 ids -
//             beyond: the device radix sort of kgat_graph.hip + a memset) +
//             bpr_window_kernel (a wavefront per 32 sorted positions adds their contributions in order; rows inside
//             the window are written, pieces of rows that cross its edges left as partials) + bpr_carry_kernel (the
//             crossing rows' pieces in window order): fixed order of additions, no float atomics, bitwise
//             reproducible; scaled by the incoming gradient read from device memory (no host synchronisation, no
//             separate multiply pass)
//   loss = -mean_b logsigmoid(<s,p> - <s,n>) + lambda (mean_b |s|^2/2 + mean_b |p|^2/2 + mean_b |n|^2/2)
#include "kgat_common.h"

namespace kgat {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// per sample b: out[b] = logsigmoid(x), out[B + b] = (|s|^2 + 0)/2 ..., coef[b] = sigmoid(-x)
__global__ __launch_bounds__(256) void bpr_sample_kernel(int64_t batch, int64_t n_nodes, int F,
                                                         const float* __restrict__ emb, int64_t stride, const int32_t* __restrict__ u,
                                                         const int32_t* __restrict__ p, const int32_t* __restrict__ n,
                                                         float* __restrict__ part, float* __restrict__ coef) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= batch) return;
  if ((uint64_t)u[b] >= (uint64_t)n_nodes || (uint64_t)p[b] >= (uint64_t)n_nodes || (uint64_t)n[b] >= (uint64_t)n_nodes) {
    // an id outside [0, n_nodes) (torch would fail a device-side assertion): NaN in the loss, no access
    if (lane == 0) {
      part[b] = __builtin_nanf(""); part[batch + b] = 0.f; part[2 * batch + b] = 0.f; part[3 * batch + b] = 0.f;
      coef[b] = 0.f;
    }
    return;
  }
  const float* rs = emb + (size_t)u[b] * stride;
  const float* rp = emb + (size_t)p[b] * stride;
  const float* rn = emb + (size_t)n[b] * stride;
  float sp = 0.f, sn = 0.f, ss = 0.f, pp = 0.f, nn = 0.f;
  for (int c = lane * 4; c < F; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(rs + c);
    const float4 x = *reinterpret_cast<const float4*>(rp + c);
    const float4 y = *reinterpret_cast<const float4*>(rn + c);
    sp += a.x * x.x + a.y * x.y + a.z * x.z + a.w * x.w;
    sn += a.x * y.x + a.y * y.y + a.z * y.z + a.w * y.w;
    ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    pp += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    nn += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
  }
  sp = wave_sum(sp); sn = wave_sum(sn); ss = wave_sum(ss); pp = wave_sum(pp); nn = wave_sum(nn);
  if (lane == 0) {
    const float x = sp - sn;
    // logsigmoid(x) = min(x, 0) - log1p(exp(-|x|)); sigmoid(-x) = 1 / (1 + exp(x))
    part[b] = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
    part[batch + b] = 0.5f * ss;
    part[2 * batch + b] = 0.5f * pp;
    part[3 * batch + b] = 0.5f * nn;
    coef[b] = x >= 0.f ? expf(-x) / (1.f + expf(-x)) : 1.f / (1.f + expf(x));
  }
}

// loss = -mean(part[0:B]) + lambda (mean(part[B:2B]) + mean(part[2B:3B]) + mean(part[3B:4B])), fixed order
__global__ __launch_bounds__(1024) void bpr_reduce_kernel(int64_t batch, const float* __restrict__ part,
                                                          float reg_lambda, float* __restrict__ loss) {
  __shared__ float s_red[4][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = threadIdx.x; i < batch; i += 1024)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += part[k * batch + i];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[k] = wave_sum(acc[k]);
    if (lane == 0) s_red[k][w] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x = 0.f;
      for (int j = 0; j < 16; ++j) x += s_red[k][j];
      t[k] = x / (float)batch;
    }
    loss[0] = -t[0] + reg_lambda * (t[1] + t[2] + t[3]);
  }
}

__global__ __launch_bounds__(256) void bpr_keys_kernel(int64_t batch, const int32_t* __restrict__ u,
                                                       const int32_t* __restrict__ p, const int32_t* __restrict__ n,
                                                       int32_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * batch) return;
  const int64_t b = i % batch;
  const int role = (int)(i / batch);
  keys[i] = role == 0 ? u[b] : (role == 1 ? p[b] : n[b]);
}

// The dense gradient from the sorted ids, in two launches of bounded work (the scheme of the aggregation's merge
// tiles and finish launch).  Launch one: a wavefront per WINDOW of 32 sorted positions fetches the 32 samples' indices
// and coefficients side by side (lanes 0-31: two dependent round trips for the window), then walks the positions in
// order, the partner rows of eight positions in flight at a time; a row id whose contributions begin and end inside
// the window is written at once, a first / last piece of a run that crosses the window's edge goes to the window's
// partial slot 0 / 1.  Launch two: the wavefront of the window in which a crossing run BEGINS adds the run's partials
// in window order and writes the row.  Every sum has a fixed order (a run inside one window: sample order, as ever;
// a crossing run: its pieces in sample order, then the pieces in window order): bitwise reproducible, no atomics.
// (The first form - one wavefront per run, one sample after the other, three dependent round trips each - is fine
// for uniformly drawn ids; the reference's samplers draw interactions uniformly over EDGES, so a popular item is the
// positive of hundreds of samples of a batch: ~1 us per sample on one wavefront, +0.9 ms on a 1.0-ms step.)
constexpr int kBprWin = 32;

struct BprSample {   // what the window's lane l knows about sorted position base + l
  int32_t key, role, r1, r2;
  float cv;
  bool ok;
};

__global__ __launch_bounds__(256) void bpr_window_kernel(int64_t batch, int64_t n_nodes, int F,
                                                         const float* __restrict__ emb, int64_t stride,
                                                         const int32_t* __restrict__ u, const int32_t* __restrict__ p,
                                                         const int32_t* __restrict__ n, const float* __restrict__ coef,
                                                         float reg_lambda, const float* __restrict__ grad_scale,
                                                         const int32_t* __restrict__ sorted,
                                                         const int32_t* __restrict__ order, float* __restrict__ grad,
                                                         float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = 3 * batch;
  const int64_t base = w * kBprWin;
  if (base >= total) return;
  const int m = (int)(total - base < kBprWin ? total - base : kBprWin);
  const float scale = (grad_scale ? grad_scale[0] : 1.f) / (float)batch;
  const float lam = reg_lambda * scale;
  BprSample me;
  {
    const int64_t pos = base + (lane < m ? lane : m - 1);
    me.key = sorted[pos];
    const int64_t idx = order[pos];
    const int64_t b = idx % batch;
    me.role = (int32_t)(idx / batch);
    const int32_t ub = u[b], pb = p[b], nb = n[b];
    me.ok = (uint64_t)ub < (uint64_t)n_nodes && (uint64_t)pb < (uint64_t)n_nodes && (uint64_t)nb < (uint64_t)n_nodes &&
            (uint64_t)me.key < (uint64_t)n_nodes;
    const float cs = coef[b] * scale;   // -dL/dx
    // the partner rows: p and n for the source row (dL/ds = dL/dx (p - n)), the source row for p / n (dL/dp = dL/dx s,
    // dL/dn = -dL/dx s)
    me.r1 = me.ok ? (me.role == 0 ? pb : ub) : 0;
    me.r2 = me.ok ? (me.role == 0 ? nb : ub) : 0;
    me.cv = me.role == 0 ? cs : (me.role == 1 ? -cs : cs);
    if (!me.ok) me.key = me.key < 0 || (uint64_t)me.key >= (uint64_t)n_nodes ? (int32_t)n_nodes : me.key;
  }
  const int32_t prev_key = base > 0 ? sorted[base - 1] : -1;
  const int32_t next_key = base + m < total ? sorted[base + m] : -2;
  for (int c0 = 0; c0 < F; c0 += 256) {   // (every lane stays in the loop: the samples live in lanes 0-31)
    const bool live = c0 + lane * 4 < F;
    const int c = live ? c0 + lane * 4 : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int32_t cur = __shfl(me.key, 0, 64);
    bool first_piece = true;
    auto flush = [&](bool last_piece) {
      if ((uint64_t)cur >= (uint64_t)n_nodes) return;              // ids outside the table carry nothing
      const bool starts = !(first_piece && prev_key == cur), ends = !(last_piece && next_key == cur);
      if (!live) return;
      if (starts && ends) *reinterpret_cast<float4*>(grad + (size_t)cur * F + c) = acc;
      else *reinterpret_cast<float4*>(partials + ((size_t)w * 2 + (starts ? 1 : 0)) * F + c) = acc;
    };
    constexpr int RB = 8;   // positions whose rows are in flight together
    for (int t0 = 0; t0 < m; t0 += RB) {
      float4 x[RB], y[RB], o[RB];
      int32_t kt[RB], rl[RB];
      float cv[RB];
      bool okt[RB];
#pragma unroll
      for (int t = 0; t < RB; ++t) {
        const int tt = t0 + t < m ? t0 + t : m - 1;
        kt[t] = __shfl(me.key, tt, 64);
        rl[t] = __shfl(me.role, tt, 64);
        okt[t] = __shfl((int)me.ok, tt, 64) != 0;
        cv[t] = __shfl(me.cv, tt, 64);
        const int32_t a1 = __shfl(me.r1, tt, 64), a2 = __shfl(me.r2, tt, 64);
        x[t] = *reinterpret_cast<const float4*>(emb + (size_t)a1 * stride + c);
        y[t] = *reinterpret_cast<const float4*>(emb + (size_t)a2 * stride + c);
        o[t] = *reinterpret_cast<const float4*>(emb + (size_t)(okt[t] ? kt[t] : 0) * stride + c);
      }
#pragma unroll
      for (int t = 0; t < RB; ++t) {
        if (t0 + t >= m) continue;
        if (kt[t] != cur) {   // (wave-uniform)
          flush(false);
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
          cur = kt[t];
          first_piece = false;
        }
        if (!okt[t]) continue;   // (a sample with an id outside the table: the forward made the loss NaN)
        float4 d;
        if (rl[t] == 0) d = make_float4(-cv[t] * (x[t].x - y[t].x), -cv[t] * (x[t].y - y[t].y), -cv[t] * (x[t].z - y[t].z),
                                        -cv[t] * (x[t].w - y[t].w));
        else d = make_float4(cv[t] * x[t].x, cv[t] * x[t].y, cv[t] * x[t].z, cv[t] * x[t].w);
        acc.x += d.x + lam * o[t].x; acc.y += d.y + lam * o[t].y; acc.z += d.z + lam * o[t].z; acc.w += d.w + lam * o[t].w;
      }
    }
    flush(true);
  }
}

// the runs that cross window edges: the wavefront of the window in which such a run begins (its last piece, slot 1)
// adds the following windows' first pieces (slot 0) in window order
__global__ __launch_bounds__(256) void bpr_carry_kernel(int64_t batch, int64_t n_nodes, int F,
                                                        const int32_t* __restrict__ sorted,
                                                        const float* __restrict__ partials, float* __restrict__ grad) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = 3 * batch;
  const int64_t base = w * kBprWin;
  if (base >= total) return;
  const int m = (int)(total - base < kBprWin ? total - base : kBprWin);
  if (base + m >= total) return;                      // the last window: nothing continues
  const int32_t key = sorted[base + m - 1];
  if (sorted[base + m] != key) return;                // its last run ends with the window
  if ((uint64_t)key >= (uint64_t)n_nodes) return;
  // does that run begin in this window?  (its first position is inside it, or it fills the window from its start on
  // with another id before it)
  if (sorted[base] == key && base > 0 && sorted[base - 1] == key) return;   // a middle window of the run
  // the windows behind it that the run reaches: window w + j continues it iff its first id is the key; it is the last
  // one iff the run ends inside it (or at its end)
  int64_t n_follow = 0;
  while (true) {   // 64 windows per look
    const int64_t wj = w + 1 + n_follow + lane;
    const bool cont = wj * kBprWin < total && sorted[wj * kBprWin] == key;
    const unsigned long long mk = __ballot(cont);
    if (mk == ~0ull) { n_follow += 64; continue; }
    n_follow += __builtin_ctzll(~mk);
    break;
  }
  for (int c0 = 0; c0 < F; c0 += 256) {
    const int c = c0 + lane * 4;
    if (c >= F) continue;
    float4 acc = *reinterpret_cast<const float4*>(partials + ((size_t)w * 2 + 1) * F + c);
    for (int64_t j0 = 0; j0 < n_follow; j0 += 8) {
      float4 v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int64_t wj = w + 1 + (j0 + t < n_follow ? j0 + t : n_follow - 1);
        v[t] = *reinterpret_cast<const float4*>(partials + ((size_t)wj * 2 + 0) * F + c);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (j0 + t < n_follow) { acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w; }
    }
    *reinterpret_cast<float4*>(grad + (size_t)key * F + c) = acc;
  }
}

// ---- stable sort of the 3B row ids, role-major (i = role * B + b), beside the zero fill of the dense gradient.
// The general device sort (kgat_graph.hip) takes four launches per digit - twelve dependent launches for 30,720
// eighteen-bit keys, 0.10 of the CF step's 0.99 ms, all of it launch latency.  Here, up to 65,536 ids: launch one -
// workgroup L sorts slice L (4,096 consecutive ids) in LDS: LSD radix, 8-bit digits, per-wavefront counter columns,
// ballots for the rank inside 64 keys (equal keys keep their order: kgat_transr.hip's small sort), packed values
// key << 17 | index; the workgroups behind the slices clear `zero` (the N x F gradient) meanwhile.  Launch two - one
// thread per id finds its place among ALL ids: its rank in its own slice + what the earlier slices hold up to its
// key + what the later slices hold below it (binary searches over <= 15 sorted slices): a stable merge without a
// merge tree.  Ids outside [0, N) sort as N, behind every real row.  (One workgroup sorting all 30,720 ids from
// registers was measured first: 180 us - 1,440 wavefront-iterations of ~100 instructions on one CU.)
constexpr int kSliceSort = 4096;
constexpr int kSliceSortMaxLists = 16;
constexpr int kSliceWaves = 16;
constexpr int kSliceDigitBits = 9;   // 18-bit ids (the CKGs of the reference's datasets) in TWO passes; eight bits took three
constexpr int kSliceDigits = 1 << kSliceDigitBits;

__global__ __launch_bounds__(1024) void bpr_sort_zero_kernel(int32_t batch, int32_t n_lists, int64_t n_nodes, int key_bits,
                                                             const int32_t* __restrict__ u, const int32_t* __restrict__ p,
                                                             const int32_t* __restrict__ n, uint64_t* __restrict__ lists,
                                                             float* __restrict__ zero, int64_t n_zero) {
  if ((int32_t)blockIdx.x >= n_lists) {
    const int64_t n4 = n_zero / 4, stride = (int64_t)(gridDim.x - n_lists) * 1024;
    float4* z = reinterpret_cast<float4*>(zero);
    for (int64_t i = (int64_t)(blockIdx.x - n_lists) * 1024 + threadIdx.x; i < n4; i += stride)
      z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  __shared__ uint64_t s_buf[2][kSliceSort];
  __shared__ int32_t s_cnt[kSliceDigits * kSliceWaves];
  __shared__ int32_t s_wsum[kSliceWaves];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t total = 3 * batch;
  const int32_t g0 = (int32_t)blockIdx.x * kSliceSort;
  const int32_t cnt = total - g0 < kSliceSort ? total - g0 : kSliceSort;
  for (int32_t i = tid; i < cnt; i += 1024) {
    const int32_t gi = g0 + i;
    const int32_t role = gi / batch, b = gi - role * batch;
    const int32_t id = role == 0 ? u[b] : (role == 1 ? p[b] : n[b]);
    const uint64_t key = (uint64_t)(uint32_t)id >= (uint64_t)n_nodes ? (uint64_t)n_nodes : (uint64_t)id;
    s_buf[0][i] = (key << 17) | (uint64_t)gi;
  }
  const int32_t slice = ((cnt + kSliceWaves * 64 - 1) / (kSliceWaves * 64)) * 64;  // per wavefront, a multiple of 64
  const int32_t lo = w * slice, hi = lo + slice < cnt ? lo + slice : cnt;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int passes = (key_bits + kSliceDigitBits - 1) / kSliceDigitBits;
  int cur = 0;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 17 + kSliceDigitBits * pass;
    constexpr int CPT = kSliceDigits * kSliceWaves / 1024;   // counters per thread
    for (int i = tid; i < kSliceDigits * kSliceWaves; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    for (int32_t i = lo + lane; i < hi; i += 64)
      atomicAdd(&s_cnt[(uint32_t)((s_buf[cur][i] >> shift) & (uint32_t)(kSliceDigits - 1)) * kSliceWaves + w], 1);
    __syncthreads();
    {  // exclusive scan of the counters in (digit, wavefront) order, CPT per thread
      int32_t c[CPT];
      int32_t mine = 0;
#pragma unroll
      for (int q = 0; q < CPT; ++q) { c[q] = s_cnt[CPT * tid + q]; mine += c[q]; }
      int32_t inc = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int32_t up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
      }
      if (lane == 63) s_wsum[w] = inc;
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
    for (int32_t i0 = lo; i0 < hi; i0 += 64) {
      const int32_t i = i0 + lane;
      const bool valid = i < hi;
      const uint64_t v = valid ? s_buf[cur][i] : 0ull;
      const uint32_t dgt = (uint32_t)((v >> shift) & (uint32_t)(kSliceDigits - 1));
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < kSliceDigitBits; ++bit) {
        const bool on = (dgt >> bit) & 1u;
        const uint64_t bal = __ballot(on);
        peers &= on ? bal : ~bal;
      }
      const int rank = __popcll(peers & lt_mask);
      const int32_t basep = valid ? s_cnt[dgt * kSliceWaves + w] : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every lane has its base before a leader moves it
      if (valid) {
        s_buf[cur ^ 1][basep + rank] = v;
        if (rank == 0) s_cnt[dgt * kSliceWaves + w] = basep + __popcll(peers);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int32_t i = tid; i < cnt; i += 1024) lists[g0 + i] = s_buf[cur][i];
}

// the place of every id among all ids (see above); thread t = position t of the concatenated sorted slices
// (NL: the slices the search is unrolled over - 8 at the CF step's batch, 30,720 ids in eight slices)
template <int NL>
__global__ __launch_bounds__(256) void bpr_merge_kernel(int32_t total, int32_t n_lists, const uint64_t* __restrict__ lists,
                                                        int32_t* __restrict__ order, int32_t* __restrict__ sorted,
                                                        float* __restrict__ zero, int64_t n_zero) {
  // the blocks behind the merge's own: the second part of the gradient's zero fill (see kgat_bpr_grad_f32)
  const int32_t merge_blocks = (total + 255) / 256;
  if ((int32_t)blockIdx.x >= merge_blocks) {
    const int64_t n4 = n_zero / 4, stride = (int64_t)(gridDim.x - merge_blocks) * 256;
    float4* z = reinterpret_cast<float4*>(zero);
    for (int64_t i = (int64_t)(blockIdx.x - merge_blocks) * 256 + threadIdx.x; i < n4; i += stride)
      z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int32_t t = (int32_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const uint64_t v = lists[t];
  const uint64_t key = v >> 17;
  const int32_t mine = t / kSliceSort;
  int32_t pos = t - mine * kSliceSort;
  // One binary search per other slice - earlier slices: ids <= key come first (upper bound); later slices: ids < key
  // (lower bound) - all of them advanced TOGETHER, one probe of every slice per round: the probes of a round are
  // independent loads (round 6: slice after slice it was a chain of up to 15 x 12 dependent L2 round trips, 19.7 us
  // for the CF step's 30,720 ids).  Same positions.
  int32_t lo[NL], hi[NL];
#pragma unroll
  for (int L = 0; L < NL; ++L) {
    const int32_t g0 = L * kSliceSort;
    lo[L] = 0;
    hi[L] = (L < n_lists && L != mine) ? (total - g0 < kSliceSort ? total - g0 : kSliceSort) : 0;
  }
  for (int step = 0; step < 13; ++step) {   // 2^12 = kSliceSort entries: 13 halvings empty every range
    uint64_t probe[NL];
#pragma unroll
    for (int L = 0; L < NL; ++L) {
      // (unconditional: a load under a lane mask is waited for on the spot - sixteen round trips per round again; an
      //  empty range probes entry 0 of its slice, or of slice 0 if the slice does not exist, and ignores it)
      const int32_t mid = lo[L] < hi[L] ? (lo[L] + hi[L]) >> 1 : 0;
      probe[L] = lists[(L < n_lists ? L * kSliceSort : 0) + mid] >> 17;
    }
#pragma unroll
    for (int L = 0; L < NL; ++L) {
      const int32_t mid = (lo[L] + hi[L]) >> 1;
      const uint64_t bound = L < mine ? key + 1 : key;
      if (lo[L] < hi[L]) {
        if (probe[L] < bound) lo[L] = mid + 1; else hi[L] = mid;
      }
    }
  }
#pragma unroll
  for (int L = 0; L < NL; ++L) pos += lo[L];
  order[pos] = (int32_t)(v & 0x1FFFFull);
  sorted[pos] = (int32_t)key;
}

static int bits_for_id(int64_t max_id) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) <= max_id) ++b;
  return b;
}

}  // namespace kgat

using namespace kgat;

extern "C" {

size_t kgat_bpr_workspace_bytes(int64_t batch, int F) {
  if (batch < 1) batch = 1;
  if (F < 4) F = 4;
  // per-sample partials (4 B floats) | keys, order (3 B int32 each) | sort scratch: the device radix sort's, or the
  // one-workgroup sort's two packed buffers + its sorted keys
  const size_t one_wg = 3 * batch <= kSliceSort * kSliceSortMaxLists ? align_up((size_t)batch * 3 * 8, 256) + align_up((size_t)batch * 3 * 4, 256) : 0;
  const size_t radix = radix_sort_workspace_bytes(3 * batch);
  // + the window partials of the gradient (two rows of F floats per 32 sorted positions)
  const size_t win = align_up((size_t)((3 * batch + kBprWin - 1) / kBprWin) * 2 * (size_t)F * 4, 256);
  return align_up((size_t)batch * 4 * 4, 256) + 2 * align_up((size_t)batch * 3 * 4, 256) + win + (one_wg > radix ? one_wg : radix) + 256;
}

int kgat_bpr_loss_f32(int64_t n_nodes, int F, const float* emb, int64_t emb_stride, int64_t batch, const int32_t* u,
                      const int32_t* p, const int32_t* n, float reg_lambda, float* loss, float* coef, void* workspace,
                      size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && batch >= 1 && F >= 4 && F % 4 == 0 && emb_stride >= F && emb_stride % 4 == 0,
                 "bpr_loss: bad sizes (F and the row stride must be multiples of 4, batch >= 1)");
  KGAT_CHECK_ARG(batch < ((int64_t)1 << 29), "bpr_loss: batch too large");
  KGAT_CHECK_ARG(emb && u && p && n && loss && coef && workspace, "bpr_loss: null pointer");
  KGAT_CHECK_ARG((reinterpret_cast<uintptr_t>(emb) & 15) == 0, "bpr_loss: emb must be 16-byte aligned");
  if (workspace_bytes < kgat_bpr_workspace_bytes(batch, F)) {
    set_error("bpr_loss: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  Carver cv(workspace);
  float* part = cv.take<float>((size_t)batch * 4);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bpr_sample_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, st, batch, n_nodes, F, emb,
                     emb_stride, u, p, n, part, coef);
  KGAT_CHECK_LAUNCH("bpr_sample");
  hipLaunchKernelGGL(bpr_reduce_kernel, dim3(1), dim3(1024), 0, st, batch, (const float*)part, reg_lambda, loss);
  KGAT_CHECK_LAUNCH("bpr_reduce");
  return KGAT_OK;
}

int kgat_bpr_grad_f32(int64_t n_nodes, int F, const float* emb, int64_t emb_stride, int64_t batch, const int32_t* u,
                      const int32_t* p, const int32_t* n, const float* coef, float reg_lambda, const float* grad_scale,
                      float* grad, void* workspace, size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 1 && batch >= 1 && F >= 4 && F % 4 == 0 && emb_stride >= F && emb_stride % 4 == 0,
                 "bpr_grad: bad sizes (F and the row stride must be multiples of 4, batch >= 1)");
  KGAT_CHECK_ARG(batch < ((int64_t)1 << 29), "bpr_grad: batch too large");
  KGAT_CHECK_ARG(emb && u && p && n && coef && grad && workspace, "bpr_grad: null pointer");
  KGAT_CHECK_ARG(((reinterpret_cast<uintptr_t>(emb) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0,
                 "bpr_grad: emb and grad must be 16-byte aligned");
  if (workspace_bytes < kgat_bpr_workspace_bytes(batch, F)) {
    set_error("bpr_grad: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  Carver cv(workspace);
  cv.take<float>((size_t)batch * 4);
  int32_t* keys = cv.take<int32_t>((size_t)batch * 3);
  int32_t* order = cv.take<int32_t>((size_t)batch * 3);
  float* win_part = cv.take<float>((size_t)((3 * batch + kBprWin - 1) / kBprWin) * 2 * (size_t)F);
  void* sort_ws = cv.base + cv.off;
  hipStream_t st = as_stream(stream);
  const int32_t* sorted = nullptr;
  if (3 * batch <= kSliceSort * kSliceSortMaxLists && n_nodes < ((int64_t)1 << 30)) {
    Carver cs(sort_ws);
    uint64_t* lists = cs.take<uint64_t>((size_t)batch * 3);
    int32_t* sorted_w = cs.take<int32_t>((size_t)batch * 3);
    const int32_t total = (int32_t)(3 * batch), n_lists = (total + kSliceSort - 1) / kSliceSort;
    // The zero fill of the N x F gradient (112 MB at the amazon-book size: 25 us of stores) rides beside the two
    // launches in front of the scatter, both of them chains of round trips on a few workgroups: the sort's launch
    // (~17 us with two radix passes) takes the first 75 %, the merge's (12 us) the rest.  F is a multiple of 4, so both
    // parts are whole float4s.
    const unsigned zero_blocks = 2u * (unsigned)device_cu_count();
    const int64_t n_all = (int64_t)n_nodes * F;
    const int64_t n_first = (n_all * 75 / 100) / 4 * 4;
    hipLaunchKernelGGL(bpr_sort_zero_kernel, dim3((unsigned)n_lists + zero_blocks), dim3(1024), 0, st, (int32_t)batch, n_lists,
                       n_nodes, bits_for_id(n_nodes), u, p, n, lists, grad, n_first);
    KGAT_CHECK_LAUNCH("bpr_sort_zero");
    const unsigned merge_blocks = (unsigned)((total + 255) / 256), zero_blocks2 = 8u * (unsigned)device_cu_count();
    if (n_lists <= 8)
      hipLaunchKernelGGL(bpr_merge_kernel<8>, dim3(merge_blocks + zero_blocks2), dim3(256), 0, st, total, n_lists,
                         (const uint64_t*)lists, order, sorted_w, grad + n_first, n_all - n_first);
    else
      hipLaunchKernelGGL(bpr_merge_kernel<kSliceSortMaxLists>, dim3(merge_blocks + zero_blocks2), dim3(256), 0, st, total,
                         n_lists, (const uint64_t*)lists, order, sorted_w, grad + n_first, n_all - n_first);
    KGAT_CHECK_LAUNCH("bpr_merge");
    sorted = sorted_w;
  } else {
    hipLaunchKernelGGL(bpr_keys_kernel, dim3((unsigned)((3 * batch + 255) / 256)), dim3(256), 0, st, batch, u, p, n, keys);
    KGAT_CHECK_LAUNCH("bpr_keys");
    const int rc = radix_sort_index(keys, 3 * batch, bits_for_id(n_nodes - 1), order, &sorted, sort_ws,
                                    workspace_bytes - cv.off, st);
    if (rc != KGAT_OK) return rc;
    if (hipMemsetAsync(grad, 0, (size_t)n_nodes * F * sizeof(float), st) != hipSuccess) {
      set_error("bpr_grad: memset failed");
      return KGAT_E_HIP;
    }
  }
  {
    const int64_t n_win = (3 * batch + kBprWin - 1) / kBprWin;
    hipLaunchKernelGGL(bpr_window_kernel, dim3((unsigned)((n_win + 3) / 4)), dim3(256), 0, st, batch, n_nodes, F, emb,
                       emb_stride, u, p, n, coef, reg_lambda, grad_scale, sorted, (const int32_t*)order, grad, win_part);
    KGAT_CHECK_LAUNCH("bpr_window");
    hipLaunchKernelGGL(bpr_carry_kernel, dim3((unsigned)((n_win + 3) / 4)), dim3(256), 0, st, batch, n_nodes, F, sorted,
                       (const float*)win_part, grad);
  }
  KGAT_CHECK_LAUNCH("bpr_carry");
  return KGAT_OK;
}

}  // extern "C"
