// Round 5 probe: what the LANE SHAPE of a row gather costs (the attention kernel's question).  E random 256-byte
// rows of a 40.8 MB table (N = 159,251: the benchmark's embedding table), 64 rows in flight per wavefront in every
// shape, 16-byte loads:
//   A  16 adjacent lanes per row, one instruction covers 4 whole rows               (aggregation, gather probe)
//   B  4 adjacent lanes per row, 4 instructions at stride 64 B cover 16 rows        (attention, tail rows)
//   C  4 lanes SIXTEEN APART per row (lane = group + 16 q), else as B               (attention, head rows, 16x16 MFMA layout)
//   D  2 lanes 32 apart per row, 32 B per lane in two loads, 8 instr cover 32 rows  (head rows, 32x32 MFMA layout)
//   E  as B with the last third of every 64 rows clamped to one row                 (partly filled 64-position chunks)
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/build/gather_lane_shapes scripts/micro/gather_lane_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned row_of(long p, unsigned n_rows) {
  return (unsigned)(((unsigned long long)hash32((unsigned)p ^ 0x9e3779b9u) * n_rows) >> 32);
}

template <int SHAPE>
__global__ __launch_bounds__(256) void gather_kernel(long n_reads, unsigned n_rows, const float4* __restrict__ X,
                                                     float4* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long base = wave * 1024;  // 16 chunks of 64 rows per wavefront
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long p0 = base; p0 < base + 1024 && p0 < n_reads; p0 += 64) {
    float4 v[16];
    if (SHAPE == 0) {  // A
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = X[(size_t)row_of(p0 + 4 * u + (lane >> 4), n_rows) * 16 + (lane & 15)];
    } else if (SHAPE == 1 || SHAPE == 4) {  // B, E: instruction (s, m): row of position 4 (lane / 4) + s, piece m
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        long p = p0 + 4 * (lane >> 2) + s;
        if (SHAPE == 4 && 4 * (lane >> 2) + s >= 41) p = p0 + 40;
        const unsigned r = row_of(p, n_rows);
#pragma unroll
        for (int m = 0; m < 4; ++m) v[4 * s + m] = X[(size_t)r * 16 + 4 * m + (lane & 3)];
      }
    } else if (SHAPE == 2) {  // C: lane = i + 16 q holds pieces 4 m + q of the row of position 16 s + i
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned r = row_of(p0 + 16 * s + (lane & 15), n_rows);
#pragma unroll
        for (int m = 0; m < 4; ++m) v[4 * s + m] = X[(size_t)r * 16 + 4 * m + (lane >> 4)];
      }
    } else {  // D: lane = g + 32 h holds 32 B at 32 h + 64 m of the row of position 32 s + g
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned r = row_of(p0 + 32 * s + (lane & 31), n_rows);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          v[8 * s + 2 * m] = X[(size_t)r * 16 + 4 * m + 2 * (lane >> 5)];
          v[8 * s + 2 * m + 1] = X[(size_t)r * 16 + 4 * m + 2 * (lane >> 5) + 1];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) sink[wave] = acc;
}

// F: shape B from PERSISTENT workgroups (the fused attention kernel's launch): `grid` workgroups of `threads`, every
// wavefront walks its contiguous share of the rows in 64-row chunks, all 16 loads of a chunk waited for before the next
__global__ __launch_bounds__(1024) void gather_persistent_kernel(long n_reads, unsigned n_rows, const float4* __restrict__ X,
                                                                 float4* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long chunks = (n_reads + 63) / 64;
  const long c0 = chunks * wave / nw, c1 = chunks * (wave + 1) / nw;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long c = c0; c < c1; ++c) {
    const long p0 = c * 64;
    float4 v[16];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned r = row_of(p0 + 4 * (lane >> 2) + s, n_rows);
#pragma unroll
      for (int m = 0; m < 4; ++m) v[4 * s + m] = X[(size_t)r * 16 + 4 * m + (lane & 3)];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) sink[wave] = acc;
}

static float run_persistent(long E, unsigned N, const float4* X, float4* sink, unsigned grid, unsigned threads) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float sum = 0.f;
  for (int rep = 0; rep < 12; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(gather_persistent_kernel, dim3(grid), dim3(threads), 0, 0, E, N, X, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep >= 2) sum += ms;
  }
  return sum / 10.f * 1e3f;
}

template <int SHAPE>
static float run(long E, unsigned N, const float4* X, float4* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const unsigned blocks = (unsigned)((E + 4095) / 4096);
  float best = 1e9f, sum = 0.f;
  for (int rep = 0; rep < 12; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((gather_kernel<SHAPE>), dim3(blocks), dim3(256), 0, 0, E, N, X, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep >= 2) { sum += ms; best = ms < best ? ms : best; }
  }
  return sum / 10.f * 1e3f;
}

int main(int argc, char** argv) {
  const unsigned N = argc > 1 ? (unsigned)atol(argv[1]) : 159251u;
  const long E = argc > 2 ? atol(argv[2]) : 3663302;
  float4 *X, *sink;
  hipMalloc(&X, (size_t)N * 256);
  hipMemset(X, 0, (size_t)N * 256);
  hipMalloc(&sink, 1 << 22);
  printf("N = %u rows of 256 B (%.1f MB), E = %ld uniformly random rows per launch (%.0f MB), 64 rows in flight per wavefront\n",
         N, N * 256 / 1e6, E, E * 256 / 1e6);
  printf("A  16 adjacent lanes per row                       %.1f us\n", run<0>(E, N, X, sink));
  printf("B  4 adjacent lanes per row, 64-B pieces           %.1f us\n", run<1>(E, N, X, sink));
  printf("C  4 lanes sixteen apart per row (16x16 MFMA)      %.1f us\n", run<2>(E, N, X, sink));
  printf("D  2 lanes 32 apart, 32 B per lane (32x32 MFMA)    %.1f us\n", run<3>(E, N, X, sink));
  printf("E  as B, 23 of 64 rows clamped to one row          %.1f us (%.0f MB useful)\n", run<4>(E, N, X, sink), E * 256 / 1e6 * 41 / 64);
  for (unsigned wpc : {4u, 8u, 12u, 16u, 24u, 32u})
    printf("F  shape B, persistent: %2u wavefronts per CU (256 CUs)  %.1f us\n", wpc,
           run_persistent(E, N, X, sink, wpc <= 16 ? 256u : 512u, wpc <= 16 ? wpc * 64u : wpc * 32u));
  return 0;
}
