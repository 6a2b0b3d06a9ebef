// Shared pieces of the attention-logit kernels (internal; see kgat_att.hip for the design).
#pragma once
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


struct AttArgs {
  unsigned grid;
  hipStream_t st;
  int n_rel;
  const int32_t *rel_ptr, *perm, *src_g, *dst_g;
  const float *ent, *W_R, *rel;
  float *logits, *logits_csr;
  const int32_t* pos_g;
  // split form (head groups): gid per grouped position, gptr per relation, g_node per group,
  // G table (n_groups x k)
  const int32_t *gid = nullptr, *gptr = nullptr, *g_node = nullptr;
  float* G_tab = nullptr;
  unsigned long long table_bytes = 0;
  int64_t n_edges = 0;
  bool needs_memset = true;
  const int32_t* part_tptr = nullptr;  // fused form: tile range per workgroup (grid = number of parts)
  const int32_t* rec_g = nullptr;      // fused form: packed (source node | group slot << 28) per grouped position
  float* logits_g = nullptr;           // fused form: logits in grouped order
  bool f32_products = false;           // fused / folded forms: fp32 MFMA products instead of the bf16-piece products
  long long* part_clocks = nullptr;    // fused form, measurement aid: s_memrealtime at every workgroup's start and end
};


// kgat_att_persistent.hip (compiled with -amdgpu-mfma-vgpr-form: its epilogue reads the MFMA
// results from VGPRs directly); returns KGAT_E_UNSUPPORTED for widths it does not cover.
int launch_att_persistent_any(int d, const AttArgs& a);
int launch_att_split_any(int d, const AttArgs& a);
int launch_att_fold_head_any(int d, const AttArgs& a);
int launch_att_fold_fused_any(int d, const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles);
int launch_att_fold_fused32(const AttArgs& a, const int32_t* rel_tptr, const int32_t* tiles);  // d = 64, 32-group tiles  // writes V (n_groups x d) into a.G_tab
constexpr int kAttMaxRelLds = 4096;

// Lanes that share one edge in the per-edge dot product of the folded forms (a lane holds
// d / (4 * lanes) float4 pieces of the tail row and of the group's V row).  Fewer lanes per edge
// mean fewer instructions per edge (measured at d = 64, fused form: 16 lanes 0.251 ms, 8 lanes
// 0.229, 4 lanes 0.215, 2 lanes 0.217); the stand-alone per-edge launch, which lives on loads in
// flight rather than on issue slots, is fastest with 8 (0.281 ms against 0.308 with 4).  The two
// therefore sum a dot product in different orders and agree to fp32 rounding, not bit for bit.
template <int D_>
constexpr int kFusedLanesPerEdge() { return D_ / 4 < 4 ? D_ / 4 : 4; }
template <int D_>
constexpr int kTailLanesPerEdge() { return D_ / 4 < 8 ? D_ / 4 : 8; }

}  // namespace kgat
