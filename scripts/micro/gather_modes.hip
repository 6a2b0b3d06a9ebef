// Placement probe without the rest of the stack (NOTEBOOK.md 3.1): an N-row table far beyond the
// Infinity Cache, E uniformly random 256-byte row reads per pass, freshly hipMalloc'ed tables one
// after the other, for several ROW STRIDES.  If the fast / slow modes of the HBM-resident SpMM come
// from how a physical block's 256-byte rows spread over DRAM channels, a stride that is not a
// multiple of 256 B (320 B = rows padded from 64 to 80 floats) should change the picture; if they
// come from the mapping granularity (fragment size of the page tables), it should not.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/build/gather_modes scripts/micro/gather_modes.hip
// Run:   scripts/micro/build/gather_modes [rows=10000000] [reads=200000000] [tables=8]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// 16 lanes x float4 per row, U rows in flight per 16-lane group; row index = hash(position) mod n_rows
template <int U>
__global__ __launch_bounds__(256) void gather_kernel(long n_reads, unsigned n_rows, unsigned seed, long stride_f4,
                                                     const float4* __restrict__ X, float4* __restrict__ sink) {
  const int lane = threadIdx.x & 63, sl = lane & 15, sub = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long per_wave = 1024;
  const long base = wave * per_wave;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long p0 = base; p0 < base + per_wave && p0 < n_reads; p0 += 4 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned r = (unsigned)(((unsigned long long)hash32((unsigned)(p0 + 4 * u + sub) ^ seed) * n_rows) >> 32);
      v[u] = X[(size_t)r * stride_f4 + sl];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) sink[wave] = acc;
}

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 10000000, E = argc > 2 ? atol(argv[2]) : 200000000;
  const int tables = argc > 3 ? atoi(argv[3]) : 8;
  float4* sink;
  hipMalloc(&sink, 1 << 22);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const unsigned blocks = (unsigned)((E + 4095) / 4096);
  const int strides[] = {256, 320, 384, 512};
  for (int si = 0; si < 4; ++si) {
    const long stride = strides[si];
    printf("row stride %ld B (table %.2f GB):", stride, N * stride / 1e9);
    for (int t = 0; t < tables; ++t) {
      float4* X;
      if (hipMalloc(&X, (size_t)N * stride) != hipSuccess) { printf(" alloc failed"); break; }
      hipMemset(X, 0, (size_t)N * stride);
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((gather_kernel<8>), dim3(blocks), dim3(256), 0, 0, E, (unsigned)N, 17u + rep, stride / 16, X, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf(" %.2f", best);
      fflush(stdout);
      hipFree(X);
    }
    printf("  ms per pass of %ld x 256 B reads\n", E);
  }
  return 0;
}
