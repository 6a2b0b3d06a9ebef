// Micro-benchmark (developer tool, round 6): what bounds the evaluation sweep - v_mfma_f32_32x32x2_f32 with NCH
// accumulator chains per wavefront, B operands in registers, the A operands (a) constant registers, (b) 16-byte loads
// from a table that fits the L1/L2, (c) 16-byte loads streaming a 17.5 MB table (the item fragments), two wavefronts
// per SIMD as in the shipped kernel.  hipcc --offload-arch=gfx950 -O3 eval_sweep_rate.hip -o eval_sweep_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// one "half step" = NCH loads (one per chain) + 4 * NCH MFMAs; loads issued AHEAD half steps ahead
template <int NCH, int MODE, int AHEAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2), amdgpu_num_vgpr(248))) void sweep(
    const floatx4* __restrict__ tab, long n_vec, float* out, int iters, float seed) {
  floatx16 acc[NCH];
  for (int c = 0; c < NCH; ++c)
    for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
  float b[16];
  for (int s = 0; s < 16; ++s) b[s] = seed * 2 + s;
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  // every wavefront walks the table from a different start (as the segments of the sweep do)
  long pos = (wave * 7919 * 64) % n_vec;
  floatx4 ring[AHEAD + 1][NCH];
  auto issue = [&](floatx4 (&r)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (MODE == 0) r[c] = (floatx4){seed, seed + 1, seed + 2, seed + 3};
      else r[c] = tab[pos + c * 64 + lane];
    }
    pos += NCH * 64;
    if (pos + NCH * 64 > n_vec) pos = 0;
  };
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) issue(ring[a]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < AHEAD + 1; ++h) {
      issue(ring[(h + AHEAD) % (AHEAD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[h][c][u], b[(4 * h + u) & 15], acc[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int c = 0; c < NCH; ++c)
    for (int j = 0; j < 16; ++j) r += acc[c][j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NCH, int MODE, int AHEAD>
void run(const char* name, long table_bytes, int blocks) {
  float* out;
  floatx4* tab;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&tab, table_bytes);
  hipMemset(tab, 0, table_bytes);
  const long n_vec = table_bytes / 16;
  const int iters = 4000 / (AHEAD + 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  sweep<NCH, MODE, AHEAD><<<blocks, 256>>>(tab, n_vec, out, 10, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    sweep<NCH, MODE, AHEAD><<<blocks, 256>>>(tab, n_vec, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mf = (double)blocks * 4 * iters * (AHEAD + 1) * 4 * NCH;
  printf("%-64s blocks=%d  %.3f ms  %.1f TFLOP/s  loads %.2f TB/s\n", name, blocks, best, mf * 4096 / best / 1e9,
         MODE ? mf / 4 * 1024 / best / 1e9 : 0.0);
  hipFree(out); hipFree(tab);
}

int main() {
  const long small = 1 << 20, big = 17540096;
  run<2, 0, 3>("2 chains, no loads, 2 waves/SIMD", small, 512);
  run<4, 0, 3>("4 chains, no loads, 2 waves/SIMD", small, 512);
  run<2, 0, 3>("2 chains, no loads, 1 wave/SIMD", small, 256);
  run<2, 1, 3>("2 chains, loads from a 1 MB table, 3 half steps ahead", small, 512);
  run<2, 1, 3>("2 chains, loads streaming 17.5 MB, 3 half steps ahead", big, 512);
  run<2, 1, 1>("2 chains, loads streaming 17.5 MB, 1 half step ahead", big, 512);
  run<2, 1, 7>("2 chains, loads streaming 17.5 MB, 7 half steps ahead", big, 512);
  run<4, 1, 3>("4 chains, loads streaming 17.5 MB, 3 half steps ahead", big, 512);
  run<4, 1, 3>("4 chains, loads from a 1 MB table, 3 half steps ahead", small, 512);
  run<2, 1, 3>("2 chains, loads streaming 17.5 MB, 4 x 512 workgroups", big, 2048);
  return 0;
}
