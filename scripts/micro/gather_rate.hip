// Ceiling probe for the SpMM gather: E rows of D floats read through random (uniform) row indices from
// an N x D table, 16 lanes x float4 per row, nothing else (a running sum keeps the loads alive).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rate scripts/micro/gather_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void gather_kernel(long n_edges, const int* __restrict__ col,
                                                     const float4* __restrict__ X, float4* __restrict__ sink) {
  const int lane = threadIdx.x & 63, sl = lane & 15, sub = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long per_wave = 256;  // edges per wavefront
  const long base = wave * per_wave;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long p0 = base; p0 < base + per_wave && p0 < n_edges; p0 += 4 * U) {
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long p = p0 + 4 * u + sub;
      c[u] = col[p < n_edges ? p : n_edges - 1];
    }
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = X[(size_t)c[u] * 16 + sl];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) sink[wave] = acc;
}

int main(int argc, char** argv) {
  const long N = 159251, E = 3663302;
  std::vector<int> col(E);
  srand(1);
  for (long i = 0; i < E; ++i) col[i] = (int)(((long)rand() * 32768 + rand()) % N);
  int* dcol; float4 *dX, *dsink;
  hipMalloc(&dcol, E * 4); hipMalloc(&dX, N * 256); hipMalloc(&dsink, 1 << 20);
  hipMemcpy(dcol, col.data(), E * 4, hipMemcpyHostToDevice);
  hipMemset(dX, 0, N * 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned blocks = (unsigned)((E + 1023) / 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gather_kernel<8>, dim3(blocks), dim3(256), 0, 0, E, dcol, dX, dsink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("U=8: %.1f us per pass, %.2f TB/s of gathered rows\n", ms / 20 * 1e3, E * 256.0 / (ms / 20 * 1e-3) / 1e12);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gather_kernel<16>, dim3(blocks), dim3(256), 0, 0, E, dcol, dX, dsink);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("U=16: %.1f us per pass, %.2f TB/s of gathered rows\n", ms / 20 * 1e3, E * 256.0 / (ms / 20 * 1e-3) / 1e12);
  }
  return 0;
}
