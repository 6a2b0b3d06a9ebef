// Micro-benchmark (developer tool): issue rate of v_mfma_f32_16x16x4_f32 in the access pattern
// of the attention kernel (8 independent accumulator chains, operands in VGPRs), with and
// without VALU filler between the MFMAs.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int FILL>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  floatx4 acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
  for (int s = 0; s < 16; ++s) { a[s] = seed + s + threadIdx.x; b[s] = seed * 2 + s; }
  float f = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[(s + c) & 15], acc[c], 0, 0, 0);
      if (FILL) {
#pragma unroll
        for (int j = 0; j < FILL; ++j) f = fmaf(f, 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = f;
  for (int c = 0; c < 8; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// transcendental filler: TR v_exp_f32 per k-step (independent chains) - do they issue beside the MFMAs?
template <int TR>
__global__ __launch_bounds__(256) void ktr(float* out, int iters, float seed) {
  floatx4 acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
  for (int s = 0; s < 16; ++s) { a[s] = seed + s + threadIdx.x; b[s] = seed * 2 + s; }
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = seed * 0.01f * (j + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[(s + c) & 15], acc[c], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TR; ++j) f[j & 7] = __builtin_amdgcn_exp2f(f[j & 7]) * 0.25f;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int j = 0; j < 8; ++j) r += f[j];
  for (int c = 0; c < 8; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int TR>
void runtr(const char* name, int blocks) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  ktr<TR><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  ktr<TR><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 128 * (16 * 16 * 4 * 2);
  const double waves = (double)blocks / 256.0 > 1 ? (double)blocks / 256.0 : 1;
  const double cyc_per_kstep = ms * 1e-3 * 2.4e9 / (iters * 16.0) / waves;
  printf("%-40s blocks=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles per k-step per wave; 8 MFMAs alone = 256)\n", name, blocks, ms,
         flop / ms / 1e9, cyc_per_kstep);
  hipFree(out);
}

typedef float floatx16 __attribute__((ext_vector_type(16)));

// same FLOPs per k-step with the 32x32x2 form: 4 accumulator tiles of 32x32, K = 2 per MFMA
template <int FILL>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float seed) {
  floatx16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
  float a[16], b[16];
  for (int s = 0; s < 16; ++s) { a[s] = seed + s + threadIdx.x; b[s] = seed * 2 + s; }
  float f = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + c) & 15], acc[c], 0, 0, 0);
      if (FILL) {
#pragma unroll
        for (int j = 0; j < FILL; ++j) f = fmaf(f, 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = f;
  for (int c = 0; c < 4; ++c)
    for (int j = 0; j < 16; ++j) r += acc[c][j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int FILL>
void run32(const char* name, int blocks) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k32<FILL><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k32<FILL><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 64 * (32 * 32 * 2 * 2);
  printf("%-36s blocks=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
  hipFree(out);
}

template <int FILL>
void run(const char* name, int blocks) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<FILL><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<FILL><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 128 * (16 * 16 * 4 * 2);
  double cyc_per_mfma = ms * 1e-3 * 2.4e9 / (iters * 128.0) / ((double)blocks / 256.0 > 1 ? (double)blocks / 256.0 : 1);
  printf("%-28s blocks=%d  %.3f ms  %.1f TFLOP/s  (~%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", name, blocks, ms, flop / ms / 1e9, cyc_per_mfma);
  hipFree(out);
}

int main() {
  run<0>("mfma only, 1 wave/SIMD", 256);
  run<0>("mfma only, 2 waves/SIMD", 512);
  run<4>("mfma + 4 fma/kstep, 1 w/SIMD", 256);
  run<8>("mfma + 8 fma/kstep, 1 w/SIMD", 256);
  run<16>("mfma + 16 fma/kstep, 1 w/SIMD", 256);
  run<8>("mfma + 8 fma/kstep, 2 w/SIMD", 512);
  runtr<0>("mfma + 0 v_exp/kstep, 1 w/SIMD", 256);
  runtr<2>("mfma + 2 v_exp(+mul)/kstep, 1 w/SIMD", 256);
  runtr<4>("mfma + 4 v_exp(+mul)/kstep, 1 w/SIMD", 256);
  runtr<8>("mfma + 8 v_exp(+mul)/kstep, 1 w/SIMD", 256);
  runtr<4>("mfma + 4 v_exp(+mul)/kstep, 2 w/SIMD", 512);
  run32<0>("32x32x2 only, 1 wave/SIMD", 256);
  run32<8>("32x32x2 + 8 fma/kstep, 1 w/SIMD", 256);
  run32<16>("32x32x2 + 16 fma/kstep, 1 w/SIMD", 256);
  run32<8>("32x32x2 + 8 fma/kstep, 2 w/SIMD", 512);
  return 0;
}
