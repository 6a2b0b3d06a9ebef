// Internal helpers shared by the gfx950 kernels of libkgat_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/kgat_hip.h"

namespace kgat {

constexpr int kWave = 64;  // gfx950 wavefront

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(kgat_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Carves 256-byte aligned regions out of a caller-provided workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off = align_up(off + count * sizeof(T), 256);
    return p;
  }
};

// Compute units of the current device; the attribute query is made once per device ordinal.
inline int device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int v = 0;
    cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cached[dev];
}

// Workgroups are dealt round-robin over the 8 XCDs (observed placement, a speed matter only: blocks b and
// b + 8 share an L2; MI355X_MICROARCH.md, Workgroup dispatch).  The work item of block b out of n such
// that the blocks of one XCD take a CONTIGUOUS eighth of the items (bijective for any n).  Used by the fused
// attention kernel at d <= 64 (KGAT_ATT_XCD_REMAP); slower on the aggregation's tiles and on the d = 128 attention
// (notes at KGAT_SPMM_XCD_REMAP, KGAT_ATT128_XCD_REMAP: A/B builds).
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned n) {
  const unsigned q = n / 8, r = n % 8, x = b % 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

// Device-side primitives implemented in kgat_graph.hip, reused by other translation units.
size_t scan_workspace_elems(int64_t n);
// In-place exclusive scan of int32 data[n]; ws has scan_workspace_elems(n) int32.
int exclusive_scan_i32(int32_t* data, int64_t n, int32_t* ws, hipStream_t st);

size_t radix_sort_workspace_bytes(int64_t n);
// Stable LSD radix sort of (key, index) pairs on the low `key_bits` bits of keys_in[n].
// vals_out[n] receives the source index of each sorted position; *sorted_keys (optional)
// receives a pointer (inside ws) to the sorted keys.
int radix_sort_index(const int32_t* keys_in, int64_t n, int key_bits, int32_t* vals_out,
                     const int32_t** sorted_keys, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace kgat

#define KGAT_CHECK_ARG(cond, ...)   \
  do {                              \
    if (!(cond)) {                  \
      kgat::set_error(__VA_ARGS__); \
      return KGAT_E_BADARG;         \
    }                               \
  } while (0)

#define KGAT_CHECK_LAUNCH(what)                                                    \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      kgat::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));      \
      return KGAT_E_HIP;                                                           \
    }                                                                              \
  } while (0)
