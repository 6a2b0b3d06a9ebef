// Developer probe (round 6): the edge-id-ordered copy of the attention weights, w_eid[i] = w_csr[pos[i]] (gather4_kernel,
// 26-28 us of the default step: 3.66 M random 4-byte reads of a 14.7 MB table that no single 4 MiB L2 holds).
// Variant: every XCD serves the reads of ONE eighth of the table (L2-resident there) - the (eid, position) pairs are
// sorted by (position eighth, eid) once per graph, workgroup b takes chunk b / 8 of eighth b % 8 (workgroups go to the
// XCDs round robin) - and writes its edge ids, which ascend inside a chunk but with gaps.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void gather_plain(int64_t n, const int32_t* __restrict__ idx, const float* __restrict__ src,
                                                    float* __restrict__ dst) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const int4 p = *reinterpret_cast<const int4*>(idx + i);
    float4 v;
    v.x = src[p.x]; v.y = src[p.y]; v.z = src[p.z]; v.w = src[p.w];
    *reinterpret_cast<float4*>(dst + i) = v;
  } else {
    for (int64_t j = i; j < n; ++j) dst[j] = src[idx[j]];
  }
}

// seg_ptr[9]: pair ranges of the eight table eighths; block b: eighth b % 8, chunk b / 8 (1,024 pairs per chunk)
__global__ __launch_bounds__(256) void gather_xcd(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ pi,
                                                  const int32_t* __restrict__ pp, const float* __restrict__ src,
                                                  float* __restrict__ dst) {
  const int x = blockIdx.x & 7;
  const int64_t lo = seg_ptr[x], hi = seg_ptr[x + 1];
  const int64_t t = lo + ((int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x) * 4;
  if (t + 3 < hi) {
    const int4 i4 = *reinterpret_cast<const int4*>(pi + t);   // (lo is a multiple of 4 by construction)
    const int4 p4 = *reinterpret_cast<const int4*>(pp + t);
    const float a = src[p4.x], b = src[p4.y], c = src[p4.z], d = src[p4.w];
    dst[i4.x] = a; dst[i4.y] = b; dst[i4.z] = c; dst[i4.w] = d;
  } else {
    for (int64_t j = t; j < hi; ++j) dst[pi[j]] = src[pp[j]];
  }
}

extern "C" {
int perm_plain(int64_t n, const int32_t* idx, const float* src, float* dst, void* st) {
  hipLaunchKernelGGL(gather_plain, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)st, n, idx, src, dst);
  return (int)hipGetLastError();
}
int perm_xcd(int64_t max_seg, const int32_t* seg_ptr, const int32_t* pi, const int32_t* pp, const float* src, float* dst,
             void* st) {
  const unsigned chunks = (unsigned)((max_seg + 1023) / 1024);
  hipLaunchKernelGGL(gather_xcd, dim3(chunks * 8), dim3(256), 0, (hipStream_t)st, seg_ptr, pi, pp, src, dst);
  return (int)hipGetLastError();
}
}
