// Developer probe (round 6; VERDICT round 5, task 1): what would ONE CSR-order pass cost that gathers every tail row
// once and does both things the step does with it today - the attention logit (e_t . V[g], V[g] = W_r tanh(W_r^T e_h +
// e_r) precomputed per (head, relation) group by a head kernel) and the layer-1 aggregation (sum of exp(s) x_t per
// destination) - instead of the relation-grouped gather-dot of att_fold_fused_kernel (131.6 us) + edge_softmax
// (22.7 us) + the edge-id-order permutation (27.7 us) + the D = 64 aggregation (69.9 us)?
//
// pass kernel:      the aggregation's merge-path decomposition (256 threads = 16 lane groups of 16 lanes, a lane group
//                   walks a run of 64 CSR positions in groups of four, records staged in LDS, the next group's rows
//                   requested before the current one is consumed), plus per edge: the V row of its group (global
//                   load, consecutive edges of a group hit the same line), a 4-FMA partial dot, a 16-lane DPP sum,
//                   one exp2, and the running denominator of the row.  NO running maximum (softmax is shift
//                   invariant; exp(s) / sum exp(s) is exact arithmetic's answer and safe while |s| < 80): the cheapest
//                   form such a pass can take - an online-softmax rescale only adds instructions.  Rows a run does
//                   not finish go to LDS partials and a serial in-order combine; rows a tile does not finish go to a
//                   two-slot partial buffer (the finish launch / the dense kernel's deferred rows are not in the probe).
// normalise kernel: w[p] = exp(s[p]) / l[row_of[p]] in CSR order AND scattered to edge-id order.
//
// Built by scripts/micro/fused_pass_probe.py (hipcc --offload-arch=gfx950 -shared); not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int kWave = 64;
constexpr int LPR = 16, THREADS = 256, NSUB = THREADS / LPR, C = 64, TE = NSUB * C, G = 4;
constexpr float kLog2e = 1.4426950408889634f;

struct alignas(16) Rec {
  int32_t c, r, g, pad;
};

__device__ __forceinline__ float dpp_add(float d, int ctrl_id) {
  switch (ctrl_id) {
    case 0: return d + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xF, 0xF, true));   // quad xor 1
    case 1: return d + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x4E, 0xF, 0xF, true));   // quad xor 2
    case 2: return d + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x141, 0xF, 0xF, true));  // row_half_mirror
    default: return d + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0x140, 0xF, 0xF, true)); // row_mirror
  }
}

template <bool WITH_DOT>
__global__ __launch_bounds__(THREADS) void pass_kernel(int64_t n_edges, const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ row_of,
                                                       const int32_t* __restrict__ gidx, const float4* __restrict__ X,
                                                       const float4* __restrict__ V, const float* __restrict__ w_in,
                                                       float4* __restrict__ out, float* __restrict__ l_out,
                                                       float* __restrict__ logit, float4* __restrict__ bpart,
                                                       float* __restrict__ bl) {
  __shared__ Rec s_rec[TE];
  __shared__ float4 s_part[NSUB][2][LPR];
  __shared__ float s_l[NSUB][2];
  __shared__ int32_t s_row[NSUB][2];
  const int tid = threadIdx.x, sub = tid / LPR, sl = tid % LPR;
  const int64_t tile0 = (int64_t)blockIdx.x * TE;
  const int64_t tile1 = tile0 + TE < n_edges ? tile0 + TE : n_edges;
  const int n_tile = (int)(tile1 - tile0);
  for (int k = tid; k < TE; k += THREADS) {
    Rec rec;
    if (k < n_tile) {
      rec.c = __builtin_nontemporal_load(col + tile0 + k);
      rec.r = __builtin_nontemporal_load(row_of + tile0 + k);
      rec.g = WITH_DOT ? __builtin_nontemporal_load(gidx + tile0 + k)
                       : __float_as_int(__builtin_nontemporal_load(w_in + tile0 + k));
    } else {
      rec.c = 0; rec.r = -1; rec.g = 0;
    }
    rec.pad = 0;
    s_rec[k] = rec;
  }
  __syncthreads();
  const int32_t first_row = __builtin_amdgcn_readfirstlane(s_rec[0].r);
  const int32_t last_row = __builtin_amdgcn_readfirstlane(s_rec[n_tile - 1].r);
  const Rec* run = s_rec + sub * C;
  int n_run = n_tile - sub * C;
  n_run = n_run < 0 ? 0 : (n_run > C ? C : n_run);
  const int ng = n_run / G;

  int32_t cur_row = n_run > 0 ? run[0].r : -1;
  bool head_done = false;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float l = 0.f;
  auto flush = [&]() {
    if (!head_done) {
      s_part[sub][0][sl] = acc;
      if (sl == 0) { s_row[sub][0] = cur_row; s_l[sub][0] = l; }
      head_done = true;
    } else if (cur_row >= 0) {
      out[(size_t)cur_row * LPR + sl] = acc;
      if (sl == 0) l_out[cur_row] = l;
    }
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    l = 0.f;
  };
  auto load_group = [&](int g, Rec (&rec)[G], float4 (&x)[G], float4 (&v)[G]) {
#pragma unroll
    for (int i = 0; i < G; ++i) rec[i] = run[g * G + i];
#pragma unroll
    for (int i = 0; i < G; ++i) x[i] = X[(size_t)rec[i].c * LPR + sl];
    if (WITH_DOT) {
#pragma unroll
      for (int i = 0; i < G; ++i) v[i] = V[(size_t)rec[i].g * LPR + sl];
    }
  };
  auto consume = [&](int g, const Rec (&rec)[G], const float4 (&x)[G], const float4 (&v)[G]) {
    float p[G];
    if (WITH_DOT) {
      float mine = 0.f;
#pragma unroll
      for (int i = 0; i < G; ++i) {
        float d = x[i].x * v[i].x;
        d = fmaf(x[i].y, v[i].y, d);
        d = fmaf(x[i].z, v[i].z, d);
        d = fmaf(x[i].w, v[i].w, d);
        d = dpp_add(d, 0); d = dpp_add(d, 1); d = dpp_add(d, 2); d = dpp_add(d, 3);
        mine = sl == i ? d : mine;
        p[i] = __builtin_amdgcn_exp2f(d * kLog2e);
      }
      const int64_t pos = tile0 + sub * C + g * G + sl;
      if (sl < G && pos < tile1) logit[pos] = mine;
    } else {
#pragma unroll
      for (int i = 0; i < G; ++i) p[i] = __int_as_float(rec[i].g);
    }
    if (__ballot(rec[G - 1].r != cur_row) == 0ull) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        acc.x = fmaf(p[i], x[i].x, acc.x); acc.y = fmaf(p[i], x[i].y, acc.y);
        acc.z = fmaf(p[i], x[i].z, acc.z); acc.w = fmaf(p[i], x[i].w, acc.w);
        if (WITH_DOT) l += p[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        if (rec[i].r != cur_row) { flush(); cur_row = rec[i].r; }
        acc.x = fmaf(p[i], x[i].x, acc.x); acc.y = fmaf(p[i], x[i].y, acc.y);
        acc.z = fmaf(p[i], x[i].z, acc.z); acc.w = fmaf(p[i], x[i].w, acc.w);
        if (WITH_DOT) l += p[i];
      }
    }
  };
  Rec ra[G], rb[G];
  float4 xa[G], xb[G], va[G], vb[G];
  if (ng > 0) load_group(0, ra, xa, va);
  for (int g = 0; g < ng; g += 2) {
    if (g + 1 < ng) load_group(g + 1, rb, xb, vb);
    consume(g, ra, xa, va);
    if (g + 1 >= ng) break;
    if (g + 2 < ng) load_group(g + 2, ra, xa, va);
    consume(g + 1, rb, xb, vb);
  }
  for (int j = ng * G; j < n_run; ++j) {  // only the last run of the edge range is ragged
    const Rec rec = run[j];
    const float4 x = X[(size_t)rec.c * LPR + sl];
    float p;
    if (WITH_DOT) {
      const float4 v = V[(size_t)rec.g * LPR + sl];
      float d = x.x * v.x;
      d = fmaf(x.y, v.y, d); d = fmaf(x.z, v.z, d); d = fmaf(x.w, v.w, d);
      d = dpp_add(d, 0); d = dpp_add(d, 1); d = dpp_add(d, 2); d = dpp_add(d, 3);
      if (sl == 0) logit[tile0 + sub * C + j] = d;
      p = __builtin_amdgcn_exp2f(d * kLog2e);
    } else {
      p = __int_as_float(rec.g);
    }
    if (rec.r != cur_row) { flush(); cur_row = rec.r; }
    acc.x = fmaf(p, x.x, acc.x); acc.y = fmaf(p, x.y, acc.y); acc.z = fmaf(p, x.z, acc.z); acc.w = fmaf(p, x.w, acc.w);
    if (WITH_DOT) l += p;
  }
  if (!head_done) {
    s_part[sub][0][sl] = acc;
    if (sl == 0) { s_row[sub][0] = cur_row; s_row[sub][1] = -1; s_l[sub][0] = l; }
  } else {
    s_part[sub][1][sl] = acc;
    if (sl == 0) { s_row[sub][1] = cur_row; s_l[sub][1] = l; }
  }
  __syncthreads();
  if (sub == 0) {   // serial in-order combine (the library's parallel combine is ~1.3 k ticks cheaper per tile)
    float4* bp = bpart + (size_t)blockIdx.x * 2 * LPR;
    int32_t crow = -1;
    float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
    float cl = 0.f;
    auto emit = [&](int32_t rr, const float4& v, float lv) {
      if (rr < 0) return;
      if (rr == first_row) { bp[sl] = v; if (sl == 0) bl[2 * blockIdx.x] = lv; }
      else if (rr == last_row) { bp[LPR + sl] = v; if (sl == 0) bl[2 * blockIdx.x + 1] = lv; }
      else { out[(size_t)rr * LPR + sl] = v; if (sl == 0) l_out[rr] = lv; }
    };
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int32_t rr = s_row[s][t];
        if (rr < 0) continue;
        const float4 v = s_part[s][t][sl];
        const float lv = s_l[s][t];
        if (rr == crow) {
          cacc.x += v.x; cacc.y += v.y; cacc.z += v.z; cacc.w += v.w; cl += lv;
        } else {
          emit(crow, cacc, cl);
          crow = rr; cacc = v; cl = lv;
        }
      }
    }
    emit(crow, cacc, cl);
  }
}

// tile-boundary rows: sum the tiles' partials in tile order (one lane group per tile slot; correctness aid for the
// probe's check only - the library folds this into the dense kernel)
__global__ void finish_kernel(int64_t n_edges, int32_t n_tiles, const int32_t* __restrict__ indptr,
                              const int32_t* __restrict__ row_of, float4* __restrict__ out, float* __restrict__ l_out,
                              const float4* __restrict__ bpart, const float* __restrict__ bl) {
  const int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  const int sl = threadIdx.x % LPR;
  const int32_t b = (int32_t)(item >> 1);
  const int s = (int)(item & 1);
  if (b >= n_tiles) return;
  const int64_t t0 = (int64_t)b * TE, t1 = t0 + TE < n_edges ? t0 + TE : n_edges;
  const int32_t fr = row_of[t0], lr = row_of[t1 - 1];
  if (s == 1 && lr == fr) return;
  const int32_t r = s == 0 ? fr : lr;
  const int64_t rb = indptr[r], re = indptr[r + 1];
  if ((int32_t)(rb / TE) != b) return;
  const int32_t last = (int32_t)((re - 1) / TE);
  float4 acc = bpart[((size_t)b * 2 + s) * LPR + sl];
  float l = bl[2 * b + s];
  for (int32_t bb = b + 1; bb <= last; ++bb) {
    const float4 v = bpart[((size_t)bb * 2) * LPR + sl];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    l += bl[2 * bb];
  }
  out[(size_t)r * LPR + sl] = acc;
  if (sl == 0) l_out[r] = l;
}

// w[p] = exp(s[p]) / l[row]: CSR order + (SCATTER) edge-id order
template <bool SCATTER>
__global__ __launch_bounds__(256) void normalise_kernel(int64_t n_edges, const float* __restrict__ logit,
                                                        const int32_t* __restrict__ row_of,
                                                        const int32_t* __restrict__ eid, const float* __restrict__ l,
                                                        float* __restrict__ w_csr, float* __restrict__ w_eid) {
  const int64_t p4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p4 + 3 < n_edges) {
    const float4 s = *reinterpret_cast<const float4*>(logit + p4);
    const int4 r = *reinterpret_cast<const int4*>(row_of + p4);
    float4 w;
    w.x = __builtin_amdgcn_exp2f(s.x * kLog2e) / l[r.x];
    w.y = __builtin_amdgcn_exp2f(s.y * kLog2e) / l[r.y];
    w.z = __builtin_amdgcn_exp2f(s.z * kLog2e) / l[r.z];
    w.w = __builtin_amdgcn_exp2f(s.w * kLog2e) / l[r.w];
    *reinterpret_cast<float4*>(w_csr + p4) = w;
    if (SCATTER) {
      const int4 e = *reinterpret_cast<const int4*>(eid + p4);
      w_eid[e.x] = w.x; w_eid[e.y] = w.y; w_eid[e.z] = w.z; w_eid[e.w] = w.w;
    }
  } else {
    for (int64_t p = p4; p < n_edges; ++p) {
      const float w = __builtin_amdgcn_exp2f(logit[p] * kLog2e) / l[row_of[p]];
      w_csr[p] = w;
      if (SCATTER) w_eid[eid[p]] = w;
    }
  }
}
}  // namespace

extern "C" {
// with_dot = 1: the fused pass (gidx, V); 0: the plain aggregation in the same kernel shape (w_in), the probe's own baseline
int probe_pass(int64_t n_edges, const int32_t* col, const int32_t* row_of, const int32_t* gidx, const float* X,
               const float* V, const float* w_in, float* out, float* l_out, float* logit, float* bpart, float* bl,
               int with_dot, void* stream) {
  const unsigned blocks = (unsigned)((n_edges + TE - 1) / TE);
  if (with_dot)
    hipLaunchKernelGGL(pass_kernel<true>, dim3(blocks), dim3(THREADS), 0, (hipStream_t)stream, n_edges, col, row_of, gidx,
                       (const float4*)X, (const float4*)V, w_in, (float4*)out, l_out, logit, (float4*)bpart, bl);
  else
    hipLaunchKernelGGL(pass_kernel<false>, dim3(blocks), dim3(THREADS), 0, (hipStream_t)stream, n_edges, col, row_of, gidx,
                       (const float4*)X, (const float4*)V, w_in, (float4*)out, l_out, logit, (float4*)bpart, bl);
  return (int)hipGetLastError();
}
int probe_finish(int64_t n_edges, const int32_t* indptr, const int32_t* row_of, float* out, float* l_out,
                 const float* bpart, const float* bl, void* stream) {
  const int32_t n_tiles = (int32_t)((n_edges + TE - 1) / TE);
  const int64_t threads = (int64_t)n_tiles * 2 * LPR;
  hipLaunchKernelGGL(finish_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_edges,
                     n_tiles, indptr, row_of, (float4*)out, l_out, (const float4*)bpart, bl);
  return (int)hipGetLastError();
}
int probe_normalise(int64_t n_edges, const float* logit, const int32_t* row_of, const int32_t* eid, const float* l,
                    float* w_csr, float* w_eid, int scatter, void* stream) {
  const unsigned blocks = (unsigned)((n_edges + 1023) / 1024);
  if (scatter)
    hipLaunchKernelGGL(normalise_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_edges, logit, row_of,
                       eid, l, w_csr, w_eid);
  else
    hipLaunchKernelGGL(normalise_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_edges, logit, row_of,
                       eid, l, w_csr, w_eid);
  return (int)hipGetLastError();
}
int probe_tile_edges(void) { return TE; }
}
