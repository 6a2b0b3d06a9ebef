// Developer probe (round 6): cache policy and load shape of the KG iteration's Adam pass over the entity table
// (p, m, v: 3 x 40.8 MB read and written in place, every 89 us).  POLICY bits: 1 = p, 2 = m, 4 = v streamed with
// non-temporal loads AND stores; SHAPE 0 = the library's loop (load, compute, store per 16-byte piece, four pieces per
// thread), 1 = all twelve loads of a thread issued before the first is used.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <bool NT>
__device__ __forceinline__ float4 ld(const float* p) {
  if (NT) {
    float4 v;
    v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1);
    v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3);
    return v;
  }
  return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st(float* p, const float4& v) {
  if (NT) {
    __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1);
    __builtin_nontemporal_store(v.z, p + 2); __builtin_nontemporal_store(v.w, p + 3);
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float ss,
                                         float bs, float eps) {
#pragma clang fp contract(off)
  const float dm = g - m;
  m = fmaf(w1, dm, m);
  const float t = v * beta2;
  v = fmaf(w2, g * g, t);
  float d = sqrtf(v) / bs + eps;
  p = fmaf(-ss, m / d, p);
}

template <int POLICY, int SHAPE>
__global__ __launch_bounds__(256) void adam_probe(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                  int64_t n, float w1, float beta2, float w2, float ss, float bs, float eps) {
  const int64_t base = (int64_t)blockIdx.x * 4096;
  constexpr bool PN = POLICY & 1, MN = (POLICY & 2) != 0, VN = (POLICY & 4) != 0;
  if (SHAPE == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = base + q * 1024 + threadIdx.x * 4;
      if (i >= n) break;
      float4 pp = ld<PN>(p + i), mm = ld<MN>(m + i), vv = ld<VN>(v + i);
      adam_one(pp.x, 0.f, mm.x, vv.x, w1, beta2, w2, ss, bs, eps);
      adam_one(pp.y, 0.f, mm.y, vv.y, w1, beta2, w2, ss, bs, eps);
      adam_one(pp.z, 0.f, mm.z, vv.z, w1, beta2, w2, ss, bs, eps);
      adam_one(pp.w, 0.f, mm.w, vv.w, w1, beta2, w2, ss, bs, eps);
      st<PN>(p + i, pp); st<MN>(m + i, mm); st<VN>(v + i, vv);
    }
  } else {
    float4 pp[4], mm[4], vv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int64_t i = base + q * 1024 + threadIdx.x * 4;
      i = i < n ? i : n - 4;
      pp[q] = ld<PN>(p + i); mm[q] = ld<MN>(m + i); vv[q] = ld<VN>(v + i);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = base + q * 1024 + threadIdx.x * 4;
      if (i >= n) break;
      adam_one(pp[q].x, 0.f, mm[q].x, vv[q].x, w1, beta2, w2, ss, bs, eps);
      adam_one(pp[q].y, 0.f, mm[q].y, vv[q].y, w1, beta2, w2, ss, bs, eps);
      adam_one(pp[q].z, 0.f, mm[q].z, vv[q].z, w1, beta2, w2, ss, bs, eps);
      adam_one(pp[q].w, 0.f, mm[q].w, vv[q].w, w1, beta2, w2, ss, bs, eps);
      st<PN>(p + i, pp[q]); st<MN>(m + i, mm[q]); st<VN>(v + i, vv[q]);
    }
  }
}

template <int POLICY, int SHAPE>
static void launch(float* p, float* m, float* v, int64_t n, hipStream_t st_) {
  hipLaunchKernelGGL((adam_probe<POLICY, SHAPE>), dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, st_, p, m, v, n, 0.1f,
                     0.999f, 0.001f, 1e-3f, 0.5f, 1e-8f);
}

extern "C" int adam_probe_launch(int policy, int shape, float* p, float* m, float* v, int64_t n, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define C(P) case P: if (shape) launch<P, 1>(p, m, v, n, s); else launch<P, 0>(p, m, v, n, s); break;
  switch (policy) { C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) default: return -1; }
#undef C
  return (int)hipGetLastError();
}
