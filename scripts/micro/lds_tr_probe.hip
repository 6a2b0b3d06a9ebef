// Probe (developer tool): what ds_read_b64_tr_b16 hands each lane.  A 16-lane group reads a block
// of 4 rows x 16 columns of 16-bit elements; lane 4q+p of the group supplies the address of row q,
// columns 4p..4p+3; lane i receives column i of the four rows (row r in element r).
//   hipcc --offload-arch=gfx950 -O2 lds_tr_probe.hip -o build/lds_tr_probe && build/lds_tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short shortx4 __attribute__((ext_vector_type(4)));
__global__ void k(shortx4* out) {
  __shared__ short s[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) s[i] = (short)i;  // value = row * 64 + column
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
  const short* addr = s + (4 * g + q) * 64 + 4 * p;
  out[lane] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) shortx4*)addr);
}
int main() {
  shortx4* d;
  hipMalloc(&d, 64 * sizeof(shortx4));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  shortx4 h[64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) bad += h[l][r] != (short)((4 * (l >> 4) + r) * 64 + (l & 15));
  printf("lane 0: %d %d %d %d | lane 5: %d %d %d %d | lane 21: %d %d %d %d | mismatches vs [row 4g+r][col i]: %d\n", h[0][0],
         h[0][1], h[0][2], h[0][3], h[5][0], h[5][1], h[5][2], h[5][3], h[21][0], h[21][1], h[21][2], h[21][3], bad);
  return bad != 0;
}
