// Ceiling probe for the SpMM gather: E rows of 256 B read through random (uniform) row indices from an
// N x 64 float table, LPR lanes x (16/LPR) float4 per row, nothing else (a running sum keeps the loads
// alive).  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/build/gather_rate scripts/micro/gather_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int LPR, int U>
__global__ __launch_bounds__(256) void gather_kernel(long n_edges, const int* __restrict__ col,
                                                     const float4* __restrict__ X, float4* __restrict__ sink) {
  constexpr int VPL = 16 / LPR, EPS = 64 / LPR;  // float4 per lane per row, edges per wave step
  const int lane = threadIdx.x & 63, sl = lane % LPR, sub = lane / LPR;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long per_wave = 512;
  const long base = wave * per_wave;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long p0 = base; p0 < base + per_wave && p0 < n_edges; p0 += EPS * U) {
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long p = p0 + EPS * u + sub;
      c[u] = col[p < n_edges ? p : n_edges - 1];
    }
    float4 v[U][VPL];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < VPL; ++q) v[u][q] = X[(size_t)c[u] * 16 + sl + q * LPR];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < VPL; ++q) { acc.x += v[u][q].x; acc.y += v[u][q].y; acc.z += v[u][q].z; acc.w += v[u][q].w; }
  }
  if (acc.x == 12345.678f) sink[wave] = acc;
}

template <int LPR, int U>
static void run(const char* name, long E, const int* dcol, const float4* dX, float4* dsink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned blocks = (unsigned)((E + 2047) / 2048);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gather_kernel<LPR, U>), dim3(blocks), dim3(256), 0, 0, E, dcol, dX, dsink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("%s %s: %.1f us per pass, %.2f TB/s of gathered rows\n", name, "", best / 20 * 1e3, E * 256.0 / (best / 20 * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const long N = 159251, E = 3663302;
  const bool skew = argc > 1;  // any argument: Zipf-like source popularity instead of uniform
  std::vector<int> col(E);
  srand(1);
  for (long i = 0; i < E; ++i) {
    long r = ((long)rand() * 32768 + rand()) % N;
    if (skew) { double u = (rand() + 1.0) / (RAND_MAX + 2.0); r = (long)(N * u * u * u); }
    col[i] = (int)r;
  }
  int* dcol; float4 *dX, *dsink;
  hipMalloc(&dcol, E * 4); hipMalloc(&dX, N * 256); hipMalloc(&dsink, 1 << 20);
  hipMemcpy(dcol, col.data(), E * 4, hipMemcpyHostToDevice);
  hipMemset(dX, 0, N * 256);
  printf("%s source indices\n", skew ? "skewed (u^3)" : "uniform");
  run<16, 8>("16 lanes x 1 float4, 8 rows in flight", E, dcol, dX, dsink);
  run<16, 16>("16 lanes x 1 float4, 16 rows in flight", E, dcol, dX, dsink);
  run<8, 8>(" 8 lanes x 2 float4, 8 rows in flight", E, dcol, dX, dsink);
  run<4, 4>(" 4 lanes x 4 float4, 4 rows in flight", E, dcol, dX, dsink);
  run<4, 8>(" 4 lanes x 4 float4, 8 rows in flight", E, dcol, dX, dsink);
  return 0;
}
