// Graph-structure kernels for gfx950: stable radix sort, COO -> CSR-by-destination,
// grouping of edges by relation, permutation helpers.  Row G0 / A1 of SURVEY.md 8a.
//
// Replaces (a) DGL 0.4.x's COO -> in-CSR conversion that runs on the first kernel call on a
// graph built by reference dataset.py:112-120, and (b) the R full-graph filter_edges sweeps
// of reference models.py:149-150.  One-off, integer, HBM-bound work: coalesced key reads,
// wave-ballot ranking, no MFMA.
#include <stdarg.h>

#include <type_traits>

#include "kgat_common.h"

namespace kgat {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------ exclusive scan (int32)
constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;  // 4096 elements per block

// Per-block exclusive scan; block totals go to sums[blockIdx] (if non-null).
__global__ __launch_bounds__(kScanThreads) void scan_block_kernel(int32_t* __restrict__ data,
                                                                    int64_t n,
                                                                    int32_t* __restrict__ sums) {
  __shared__ int32_t s_wave[kScanThreads / kWave];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanItems;
  int32_t v[kScanItems];
  int32_t tsum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = (base + i < n) ? data[base + i] : 0;
    tsum += v[i];
  }
  // inclusive scan of thread sums inside the wave
  int32_t inc = tsum;
  const int lane = tid & (kWave - 1), wv = tid / kWave;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    int32_t up = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += up;
  }
  if (lane == kWave - 1) s_wave[wv] = inc;
  __syncthreads();
  int32_t wbase = 0, total = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / kWave; ++i) {
    if (i < wv) wbase += s_wave[i];
    total += s_wave[i];
  }
  int32_t run = wbase + inc - tsum;  // exclusive prefix of this thread
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) data[base + i] = run;
    run += v[i];
  }
  if (sums != nullptr && tid == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanThreads) void scan_add_kernel(int32_t* __restrict__ data,
                                                                  int64_t n,
                                                                  const int32_t* __restrict__ sums) {
  const int32_t add = sums[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  for (int i = threadIdx.x; i < kScanTile; i += kScanThreads)
    if (base + i < n) data[base + i] += add;
}

size_t scan_workspace_elems(int64_t n) {
  size_t total = 0;
  while (n > kScanTile) {
    n = (n + kScanTile - 1) / kScanTile;
    total += align_up((size_t)n, 64);
  }
  return total + 64;
}

int exclusive_scan_i32(int32_t* data, int64_t n, int32_t* ws, hipStream_t st) {
  if (n <= 0) return KGAT_OK;
  const int64_t nblk = (n + kScanTile - 1) / kScanTile;
  if (nblk == 1) {
    hipLaunchKernelGGL(scan_block_kernel, dim3(1), dim3(kScanThreads), 0, st, data, n,
                       (int32_t*)nullptr);
    KGAT_CHECK_LAUNCH("scan_block");
    return KGAT_OK;
  }
  int32_t* sums = ws;
  hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nblk), dim3(kScanThreads), 0, st, data, n,
                     sums);
  KGAT_CHECK_LAUNCH("scan_block");
  int rc = exclusive_scan_i32(sums, nblk, ws + align_up((size_t)nblk, 64), st);
  if (rc != KGAT_OK) return rc;
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nblk), dim3(kScanThreads), 0, st, data, n,
                     sums);
  KGAT_CHECK_LAUNCH("scan_add");
  return KGAT_OK;
}

// ------------------------------------------------------------------ stable LSD radix sort
// One wavefront per block, kSortTile keys per block, 8-bit digits.  The scatter pass ranks
// the 64 keys of a step with wave ballots (peers with the same digit and a lower lane), so
// equal keys keep their input order: the sort is stable, which is what makes edge ids
// ascend inside every CSR row.
constexpr int kSortTile = 4096;       // keys per block on large inputs (graph builds: millions of keys)
constexpr int kSortTileSmall = 256;   // below 2^20 keys: one wavefront walking 4,096 keys in 64 dependent steps is
                                      // latency, not work (30,720 BPR row ids: 8 blocks, 240 us for three passes)
constexpr int kDigits = 256;
static int sort_tile(int64_t n) { return n >= ((int64_t)1 << 20) ? kSortTile : kSortTileSmall; }

__global__ __launch_bounds__(kWave) void sort_hist_kernel(const int32_t* __restrict__ keys,
                                                          int64_t n, int shift,
                                                          int32_t* __restrict__ hist,
                                                          int nblk, int tile) {
  __shared__ int32_t cnt[kDigits];
  const int lane = threadIdx.x;
  for (int i = lane; i < kDigits; i += kWave) cnt[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * tile;
  for (int j = lane; j < tile; j += kWave) {
    const int64_t idx = base + j;
    if (idx < n) atomicAdd(&cnt[((uint32_t)keys[idx] >> shift) & (kDigits - 1)], 1);
  }
  __syncthreads();
  for (int i = lane; i < kDigits; i += kWave) hist[(size_t)i * nblk + blockIdx.x] = cnt[i];
}

__global__ __launch_bounds__(kWave) void sort_scatter_kernel(
    const int32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in, int64_t n,
    int shift, const int32_t* __restrict__ offs, int nblk, int tile, int32_t* __restrict__ keys_out,
    int32_t* __restrict__ vals_out) {
  __shared__ int32_t run[kDigits];
  const int lane = threadIdx.x;
  for (int i = lane; i < kDigits; i += kWave) run[i] = offs[(size_t)i * nblk + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * tile;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (int step = 0; step < tile / kWave; ++step) {
    const int64_t idx = base + (int64_t)step * kWave + lane;
    const bool valid = idx < n;
    const int32_t key = valid ? keys_in[idx] : 0;
    const uint32_t dgt = ((uint32_t)key >> shift) & (kDigits - 1);
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dgt >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const int rank = __popcll(peers & lt_mask);
    const int32_t basep = run[dgt];
    __syncthreads();  // every lane has read its base before any leader bumps it
    if (valid) {
      const int32_t pos = basep + rank;
      keys_out[pos] = key;
      vals_out[pos] = vals_in ? vals_in[idx] : (int32_t)idx;
      if (rank == 0) run[dgt] = basep + __popcll(peers);
    }
    __syncthreads();
  }
}

struct SortPlan {
  int nblk;
  size_t hist_elems, scan_elems;
};

static SortPlan sort_plan(int64_t n) {
  SortPlan p;
  p.nblk = (int)((n + sort_tile(n) - 1) / sort_tile(n));
  if (p.nblk < 1) p.nblk = 1;
  p.hist_elems = (size_t)kDigits * p.nblk;
  p.scan_elems = scan_workspace_elems((int64_t)p.hist_elems);
  return p;
}

size_t radix_sort_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  SortPlan p = sort_plan(n);
  size_t b = 0;
  b += 3 * align_up((size_t)n * 4, 256);  // keys ping, keys pong, vals pong
  b += align_up(p.hist_elems * 4, 256);
  b += align_up(p.scan_elems * 4, 256);
  return b;
}

int radix_sort_index(const int32_t* keys_in, int64_t n, int key_bits, int32_t* vals_out,
                     const int32_t** sorted_keys, void* ws, size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < radix_sort_workspace_bytes(n)) {
    set_error("radix_sort: workspace too small (%zu < %zu)", ws_bytes,
              radix_sort_workspace_bytes(n));
    return KGAT_E_WORKSPACE;
  }
  Carver cv(ws);
  int32_t* K[2] = {cv.take<int32_t>((size_t)(n > 0 ? n : 1)), cv.take<int32_t>((size_t)(n > 0 ? n : 1))};
  int32_t* V1 = cv.take<int32_t>((size_t)(n > 0 ? n : 1));
  SortPlan p = sort_plan(n);
  int32_t* hist = cv.take<int32_t>(p.hist_elems);
  int32_t* scan_ws = cv.take<int32_t>(p.scan_elems);
  if (sorted_keys) *sorted_keys = K[0];
  if (n <= 0) return KGAT_OK;
  int passes = (key_bits + 7) / 8;
  if (passes < 1) passes = 1;
  int32_t* V[2] = {vals_out, V1};
  const int32_t* kin = keys_in;
  const int32_t* vin = nullptr;  // pass 0 emits the source index itself
  for (int j = 0; j < passes; ++j) {
    const int sel = (passes - 1 - j) & 1;  // the last pass lands in K[0] / vals_out
    hipLaunchKernelGGL(sort_hist_kernel, dim3(p.nblk), dim3(kWave), 0, st, kin, n, 8 * j, hist,
                       p.nblk, sort_tile(n));
    KGAT_CHECK_LAUNCH("sort_hist");
    int rc = exclusive_scan_i32(hist, (int64_t)p.hist_elems, scan_ws, st);
    if (rc != KGAT_OK) return rc;
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(p.nblk), dim3(kWave), 0, st, kin, vin, n, 8 * j,
                       (const int32_t*)hist, p.nblk, sort_tile(n), K[sel], V[sel]);
    KGAT_CHECK_LAUNCH("sort_scatter");
    kin = K[sel];
    vin = V[sel];
  }
  return KGAT_OK;
}

// ---- cost-balanced split of the fused attention kernel's tiles over its workgroups
// cost[t] = c_tile + c_chunk * ceil(max(P - 64, 0) / 64) + (t opens its relation ? c_rel : 0), P = positions of tile t
__global__ void fold_tile_cost_kernel(int64_t t_max, int n_rel, const int32_t* __restrict__ rel_tptr,
                                      const int4* __restrict__ tiles, int c_tile, int c_chunk, int c_rel,
                                      int32_t* __restrict__ cost) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > t_max) return;
  int32_t c = 0;
  if (t < rel_tptr[n_rel]) {
    const int4 d = tiles[t];
    const int32_t P = d.w - d.z;
    const int32_t later = P > 64 ? (P - 64 + 63) >> 6 : 0;
    c = c_tile + c_chunk * later + (rel_tptr[d.x] == (int32_t)t ? c_rel : 0);
  }
  cost[t] = c;
}
// prefix = exclusive scan of cost (t_max + 1 entries: prefix[t_max] = total of the tiles in use);
// part b starts at the first tile whose prefix reaches total * b / n_parts
__global__ void fold_parts_kernel(int n_rel, int64_t t_max, const int32_t* __restrict__ rel_tptr,
                                  const int32_t* __restrict__ prefix, int n_parts, int32_t* __restrict__ part_tptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_parts) return;
  const int32_t n_tiles = rel_tptr[n_rel];
  if (b == n_parts) { part_tptr[b] = n_tiles; return; }
  const int64_t total = prefix[t_max];
  const int64_t target = total * b / n_parts;
  int32_t lo = 0, hi = n_tiles;  // first t in [0, n_tiles] with prefix[t] >= target
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if ((int64_t)prefix[mid] < target) lo = mid + 1; else hi = mid;
  }
  part_tptr[b] = lo;
}

// ------------------------------------------------------------------ small helpers
// out[i] = in[index[i]]: a permutation / gather of 4-byte items.  Four consecutive outputs per
// thread: one 16-byte index load, four independent gathers in flight, one 16-byte store (the
// one-item-per-thread form spent its time on issue slots and launch granularity, not on the
// gathered sectors).  Pointers with 16-byte alignment take the vector path.
// fp32 gathers are the permutations of per-edge weights between CSR order and edge-id / reversed-CSR order: the result is
// either never read on the path (the edge-id-ordered attention tensor) or read once by the next launch's record stream,
// the index is read once - both as non-temporal accesses (round 4's A/B: eager step 0.4431 -> 0.4386 ms,
// profiles/r04_step_ab_cache_policy.txt).  Integer gathers (structure build) stay plain.
template <typename T>
__global__ __launch_bounds__(256) void gather4_kernel(int64_t n, const int32_t* __restrict__ index,
                                                      const T* __restrict__ in, T* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    typedef int i4g __attribute__((ext_vector_type(4)));
    constexpr bool NT = std::is_same<T, float>::value;
    int4 ix;
    if constexpr (NT) {
      const i4g ix4 = __builtin_nontemporal_load(reinterpret_cast<const i4g*>(index + i));
      ix = make_int4(ix4[0], ix4[1], ix4[2], ix4[3]);
    } else {
      ix = *reinterpret_cast<const int4*>(index + i);
    }
    const T a = in[ix.x], b = in[ix.y], c = in[ix.z], d = in[ix.w];
    T v[4] = {a, b, c, d};
    if constexpr (NT) {
      const int4 o4 = *reinterpret_cast<const int4*>(v);
      const i4g ov = {o4.x, o4.y, o4.z, o4.w};
      __builtin_nontemporal_store(ov, reinterpret_cast<i4g*>(out + i));
    } else {
      *reinterpret_cast<int4*>(out + i) = *reinterpret_cast<const int4*>(v);
    }
  } else {
    for (int64_t k = i; k < n; ++k) out[k] = in[index[k]];
  }
}

template <typename T>
__global__ void gather1_kernel(int64_t n, const int32_t* __restrict__ index, const T* __restrict__ in,
                               T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[index[i]];
}

template <typename T>
static void launch_gather(int64_t n, const int32_t* index, const T* in, T* out, hipStream_t st) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(index) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (aligned)
    hipLaunchKernelGGL(gather4_kernel<T>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, n, index, in, out);
  else
    hipLaunchKernelGGL(gather1_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, index, in, out);
}

__global__ void invert_perm_kernel(int64_t n, const int32_t* __restrict__ perm,
                                   int32_t* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[perm[i]] = (int32_t)i;
}

// offsets[v] = first sorted position whose key is >= v, for v in [0, n_keys]; keys sorted.
// One thread per v, binary search: the cost does not depend on how the keys are distributed (a
// destination shard has no edges for most rows - a thread per boundary would walk those gaps).
__global__ void offsets_from_sorted_kernel(int64_t n, const int32_t* __restrict__ keys,
                                           int32_t n_keys, int32_t* __restrict__ offsets) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > n_keys) return;
  int64_t lo = 0, hi = n;  // first p in [0, n] with keys[p] >= v
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (int32_t)v) lo = mid + 1; else hi = mid;
  }
  offsets[v] = (int32_t)lo;
}

__global__ void relation_key_kernel(int64_t n, int n_rel, const int32_t* __restrict__ etype,
                                    int32_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int32_t t = etype[i];
    key[i] = (t >= 0 && t < n_rel) ? t : n_rel;
  }
}

constexpr int kDegClamp = 65535;
__global__ void degree_key_kernel(int64_t n, const int32_t* __restrict__ indptr,
                                  int32_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int32_t deg = indptr[i + 1] - indptr[i];
    if (deg > kDegClamp) deg = kDegClamp;
    key[i] = kDegClamp - deg;  // ascending key = descending degree
  }
}

// ---- head groups: runs of equal (relation, destination) in a relation-grouped edge list whose
// relations are internally sorted by destination
__global__ void group_flag_kernel(int64_t n, int64_t n_scored, const int32_t* __restrict__ dst_g,
                                  int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  flag[i] = (i < n_scored && (i == 0 || dst_g[i] != dst_g[i - 1])) ? 1 : 0;
}
__global__ void group_relstart_kernel(int n_rel, const int32_t* __restrict__ rel_ptr,
                                      int32_t* __restrict__ flag) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_rel && rel_ptr[r] < rel_ptr[r + 1]) flag[rel_ptr[r]] = 1;
}
// ex = exclusive scan of the flags (E+1 entries): ex[i+1] - ex[i] recovers flag i
// Positions past rel_ptr[n_rel] (edges whose type is never scored) get group id 0: the tail kernels
// of the split forms read the ids of a whole 16-position tile, i.e. up to 15 positions past the last
// relation's end, and index the per-group table with them before discarding the result - an id
// beyond the table (these positions used to open groups of their own) is an out-of-bounds read that
// faults when the table ends a mapped segment (found by scripts/fuzz_gpu.py, seed 5393).
__global__ void group_fill_kernel(int64_t n, int n_rel, const int32_t* __restrict__ rel_ptr,
                                  const int32_t* __restrict__ ex, const int32_t* __restrict__ dst_g,
                                  int32_t* __restrict__ gid, int32_t* __restrict__ g_node) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t inc = ex[i + 1];
  const bool scored = i < rel_ptr[n_rel];
  gid[i] = scored ? inc - 1 : 0;
  if (inc != ex[i]) g_node[inc - 1] = dst_g[i];
}
__global__ void group_ptr_kernel(int n_rel, const int32_t* __restrict__ rel_ptr,
                                 const int32_t* __restrict__ ex, int32_t* __restrict__ gptr) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= n_rel) gptr[r] = ex[rel_ptr[r]];
}

// ---- packed per-position records of the fused attention kernel (kgat_att_pack_records)
__global__ void att_pack_records_kernel(int64_t n_edges, int n_rel, const int32_t* __restrict__ rel_ptr,
                                        const int32_t* __restrict__ gptr, const int32_t* __restrict__ gid,
                                        const int32_t* __restrict__ src_g, int32_t* __restrict__ rec_g, int gpt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_edges) return;
  const int slot_shift = gpt == 32 ? 27 : 28;   // the slot of 16 / 32 groups in the top 4 / 5 bits
  uint32_t rec = (uint32_t)src_g[p];
  if (p < rel_ptr[n_rel]) {
    int lo = 0, hi = n_rel;  // relation of position p: largest r with rel_ptr[r] <= p
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rel_ptr[mid] <= p) lo = mid; else hi = mid;
    }
    rec |= (uint32_t)((gid[p] - gptr[lo]) & (gpt - 1)) << slot_shift;
  }
  rec_g[p] = (int32_t)rec;
}

// ---- work tiles of the fused folded attention kernel (see kgat_fold_tiles in the header)
__global__ void fold_gstart_kernel(int n_rel, const int32_t* __restrict__ rel_ptr, int64_t n_groups,
                                   const int32_t* __restrict__ gid, int32_t* __restrict__ gstart) {
  const int64_t n_scored = rel_ptr[n_rel];
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_scored && (p == 0 || gid[p] != gid[p - 1])) gstart[gid[p]] = (int32_t)p;
  if (p == 0) gstart[n_groups] = (int32_t)n_scored;
}
// one thread: base-tile prefix per relation (16 groups per base tile, relations kept apart)
__global__ void fold_base_ptr_kernel(int n_rel, const int32_t* __restrict__ gptr, int32_t* __restrict__ bptr, int gpt) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int32_t run = 0;
  bptr[0] = 0;
  for (int r = 0; r < n_rel; ++r) {
    run += (gptr[r + 1] - gptr[r] + gpt - 1) / gpt;
    bptr[r + 1] = run;
  }
}
__device__ __forceinline__ int fold_base_tile(int n_rel, const int32_t* __restrict__ bptr,
                                              const int32_t* __restrict__ gptr,
                                              const int32_t* __restrict__ gstart, int32_t b, int gpt,
                                              int32_t& g0, int32_t& pb, int32_t& pe) {
  int lo = 0, hi = n_rel;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (bptr[mid] <= b) lo = mid; else hi = mid;
  }
  g0 = gptr[lo] + (b - bptr[lo]) * gpt;
  const int32_t g1 = g0 + gpt < gptr[lo + 1] ? g0 + gpt : gptr[lo + 1];
  pb = gstart[g0];
  pe = gstart[g1];
  return lo;
}
// cnt[b] = number of tiles base tile b becomes (0 past the last base tile); cnt has nb_max + 1 entries
__global__ void fold_count_kernel(int n_rel, int64_t nb_max, int cap, int gpt, const int32_t* __restrict__ bptr,
                                  const int32_t* __restrict__ gptr, const int32_t* __restrict__ gstart,
                                  int32_t* __restrict__ cnt) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nb_max) return;
  int32_t c = 0;
  if (b < bptr[n_rel]) {
    int32_t g0, pb, pe;
    fold_base_tile(n_rel, bptr, gptr, gstart, (int32_t)b, gpt, g0, pb, pe);
    c = (pe - pb + cap - 1) / cap;
    c = c > 0 ? c : 1;
  }
  cnt[b] = c;
}
// off = exclusive scan of cnt; writes the tiles, the tile prefix per relation and the tile count
__global__ void fold_emit_kernel(int n_rel, int64_t nb_max, int cap, int gpt, const int32_t* __restrict__ bptr,
                                 const int32_t* __restrict__ gptr, const int32_t* __restrict__ gstart,
                                 const int32_t* __restrict__ off, int32_t* __restrict__ tiles,
                                 int32_t* __restrict__ rel_tptr) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= n_rel) rel_tptr[b] = off[bptr[b]];
  if (b >= bptr[n_rel]) return;
  int32_t g0, pb, pe;
  const int r = fold_base_tile(n_rel, bptr, gptr, gstart, (int32_t)b, gpt, g0, pb, pe);
  int32_t t = off[b];
  int32_t p = pb;
  do {
    const int32_t q = p + cap < pe ? p + cap : pe;
    reinterpret_cast<int4*>(tiles)[t++] = make_int4(r, g0, p, q);
    p = q;
  } while (p < pe);
}

static inline unsigned blocks_for(int64_t n, int threads) {
  return (unsigned)((n + threads - 1) / threads);
}

static int bits_for(int64_t max_key) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) <= max_key) ++b;
  return b;
}

}  // namespace kgat

using namespace kgat;

extern "C" {

int kgat_version(void) { return KGAT_ABI_VERSION; }

const char* kgat_last_error(void) { return kgat::g_err; }

#ifndef KGAT_BUILD_HASH
#define KGAT_BUILD_HASH "unhashed"
#endif
const char* kgat_build_hash(void) { return KGAT_BUILD_HASH; }

size_t kgat_csr_from_coo_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  (void)n_nodes;
  return radix_sort_workspace_bytes(n_edges);
}

int kgat_csr_from_coo(int64_t n_nodes, int64_t n_edges, const int32_t* src, const int32_t* dst,
                      int32_t* indptr, int32_t* col, int32_t* eid, int32_t* row_of,
                      void* workspace, size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_nodes >= 0 && n_edges >= 0, "csr_from_coo: negative size");
  KGAT_CHECK_ARG(n_nodes < INT32_MAX && n_edges < INT32_MAX, "csr_from_coo: size exceeds int32");
  KGAT_CHECK_ARG(indptr != nullptr, "csr_from_coo: indptr is null");
  KGAT_CHECK_ARG(n_edges == 0 || (src && dst && col && eid && workspace),
                 "csr_from_coo: null pointer");
  hipStream_t st = as_stream(stream);
  const int32_t* sorted = nullptr;
  if (n_edges > 0) {
    int rc = radix_sort_index(dst, n_edges, bits_for(n_nodes > 0 ? n_nodes - 1 : 0), eid, &sorted,
                              workspace, workspace_bytes, st);
    if (rc != KGAT_OK) return rc;
    launch_gather<int32_t>(n_edges, (const int32_t*)eid, src, col, st);
    KGAT_CHECK_LAUNCH("csr gather col");
    if (row_of) {
      hipError_t e = hipMemcpyAsync(row_of, sorted, sizeof(int32_t) * (size_t)n_edges,
                                    hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) {
        set_error("csr_from_coo: memcpy failed: %s", hipGetErrorString(e));
        return KGAT_E_HIP;
      }
    }
  }
  hipLaunchKernelGGL(offsets_from_sorted_kernel, dim3(blocks_for(n_nodes + 1, 256)), dim3(256), 0,
                     st, n_edges, sorted, (int32_t)n_nodes, indptr);
  KGAT_CHECK_LAUNCH("csr offsets");
  return KGAT_OK;
}

size_t kgat_group_by_relation_workspace_bytes(int64_t n_edges, int n_rel) {
  (void)n_rel;
  return radix_sort_workspace_bytes(n_edges) + align_up((size_t)(n_edges > 0 ? n_edges : 1) * 4, 256) +
         align_up(((size_t)(n_rel > 0 ? n_rel : 0) + 2) * 4, 256);
}

int kgat_group_by_relation(int64_t n_edges, int n_rel, const int32_t* etype, int32_t* rel_ptr,
                           int32_t* perm, void* workspace, size_t workspace_bytes,
                           kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_edges >= 0 && n_rel >= 0, "group_by_relation: negative size");
  KGAT_CHECK_ARG(n_edges < INT32_MAX, "group_by_relation: size exceeds int32");
  KGAT_CHECK_ARG(rel_ptr != nullptr && workspace != nullptr, "group_by_relation: null pointer");
  KGAT_CHECK_ARG(n_edges == 0 || (etype && perm), "group_by_relation: null pointer");
  if (workspace_bytes < kgat_group_by_relation_workspace_bytes(n_edges, n_rel)) {
    set_error("group_by_relation: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver cv(workspace);
  int32_t* key = cv.take<int32_t>((size_t)(n_edges > 0 ? n_edges : 1));
  int32_t* offs = cv.take<int32_t>((size_t)n_rel + 2);
  void* sort_ws = cv.base + cv.off;
  const int32_t* sorted = nullptr;
  if (n_edges > 0) {
    hipLaunchKernelGGL(relation_key_kernel, dim3(blocks_for(n_edges, 256)), dim3(256), 0, st,
                       n_edges, n_rel, etype, key);
    KGAT_CHECK_LAUNCH("relation_key");
    int rc = radix_sort_index(key, n_edges, bits_for(n_rel), perm, &sorted, sort_ws,
                              workspace_bytes - cv.off, st);
    if (rc != KGAT_OK) return rc;
  }
  // offs[v] for v in [0, n_rel+1]; rel_ptr is its first n_rel+1 entries
  hipLaunchKernelGGL(offsets_from_sorted_kernel, dim3(blocks_for(n_rel + 2, 256)), dim3(256), 0,
                     st, n_edges, sorted, (int32_t)(n_rel + 1), offs);
  KGAT_CHECK_LAUNCH("relation offsets");
  hipError_t e = hipMemcpyAsync(rel_ptr, offs, sizeof(int32_t) * ((size_t)n_rel + 1),
                                hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) {
    set_error("group_by_relation: memcpy failed: %s", hipGetErrorString(e));
    return KGAT_E_HIP;
  }
  return KGAT_OK;
}

size_t kgat_head_groups_workspace_bytes(int64_t n_edges) {
  const size_t n = (size_t)(n_edges > 0 ? n_edges : 0) + 1;
  return align_up(n * 4, 256) + align_up(scan_workspace_elems((int64_t)n) * 4, 256);
}

int kgat_head_groups(int64_t n_edges, int n_rel, const int32_t* rel_ptr, const int32_t* dst_g,
                     int32_t* gid, int32_t* gptr, int32_t* g_node, void* workspace,
                     size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_edges >= 0 && n_edges < INT32_MAX - 1 && n_rel >= 0, "head_groups: bad size");
  KGAT_CHECK_ARG(rel_ptr && gptr && workspace, "head_groups: null pointer");
  KGAT_CHECK_ARG(n_edges == 0 || (dst_g && gid && g_node), "head_groups: null pointer");
  if (workspace_bytes < kgat_head_groups_workspace_bytes(n_edges)) {
    set_error("head_groups: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver cv(workspace);
  int32_t* ex = cv.take<int32_t>((size_t)n_edges + 1);
  int32_t* scan_ws = cv.take<int32_t>(scan_workspace_elems(n_edges + 1));
  // (positions past rel_ptr[n_rel] - edges whose type is never scored - open groups of their
  // own beyond gptr[n_rel] in the scan; g_node is sized for them, their gid entries are written as 0)
  hipLaunchKernelGGL(group_flag_kernel, dim3(blocks_for(n_edges + 1, 256)), dim3(256), 0, st,
                     n_edges, n_edges, dst_g, ex);
  KGAT_CHECK_LAUNCH("group_flag");
  if (n_rel > 0) {
    hipLaunchKernelGGL(group_relstart_kernel, dim3(blocks_for(n_rel, 256)), dim3(256), 0, st, n_rel,
                       rel_ptr, ex);
    KGAT_CHECK_LAUNCH("group_relstart");
  }
  int rc = exclusive_scan_i32(ex, n_edges + 1, scan_ws, st);
  if (rc != KGAT_OK) return rc;
  if (n_edges > 0) {
    hipLaunchKernelGGL(group_fill_kernel, dim3(blocks_for(n_edges, 256)), dim3(256), 0, st, n_edges, n_rel, rel_ptr,
                       (const int32_t*)ex, dst_g, gid, g_node);
    KGAT_CHECK_LAUNCH("group_fill");
  }
  hipLaunchKernelGGL(group_ptr_kernel, dim3(blocks_for(n_rel + 1, 256)), dim3(256), 0, st, n_rel,
                     rel_ptr, (const int32_t*)ex, gptr);
  KGAT_CHECK_LAUNCH("group_ptr");
  return KGAT_OK;
}

static int64_t fold_base_max(int64_t n_groups, int n_rel) { return n_groups / 16 + n_rel + 1; }

int64_t kgat_fold_tiles_max(int64_t n_edges, int64_t n_groups, int n_rel, int cap) {
  if (n_edges < 0 || n_groups < 0 || n_rel < 0 || cap <= 0) return 0;
  return fold_base_max(n_groups, n_rel) + n_edges / cap + 1;
}

size_t kgat_fold_tiles_workspace_bytes(int64_t n_groups, int n_rel) {
  const size_t nb = (size_t)fold_base_max(n_groups > 0 ? n_groups : 0, n_rel > 0 ? n_rel : 0) + 1;
  return align_up(((size_t)(n_groups > 0 ? n_groups : 0) + 1) * 4, 256) + align_up(((size_t)n_rel + 2) * 4, 256) +
         align_up(nb * 4, 256) + align_up(scan_workspace_elems((int64_t)nb) * 4, 256);
}

int kgat_fold_tiles(int64_t n_edges, int n_rel, int64_t n_groups, const int32_t* rel_ptr, const int32_t* gid,
                    const int32_t* gptr, int cap, int groups_per_tile, int32_t* tiles, int32_t* rel_tptr,
                    void* workspace, size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(groups_per_tile == 16 || groups_per_tile == 32, "fold_tiles: 16 or 32 groups per tile");
  const int gpt = groups_per_tile;
  KGAT_CHECK_ARG(n_edges >= 0 && n_edges < INT32_MAX - 1 && n_rel > 0 && n_groups >= 0 && n_groups <= n_edges,
                 "fold_tiles: bad size");
  KGAT_CHECK_ARG(cap >= 64 && cap % 64 == 0, "fold_tiles: cap must be a positive multiple of 64");
  KGAT_CHECK_ARG(rel_ptr && gptr && tiles && rel_tptr && workspace, "fold_tiles: null pointer");
  KGAT_CHECK_ARG(n_edges == 0 || gid, "fold_tiles: null pointer");
  if (workspace_bytes < kgat_fold_tiles_workspace_bytes(n_groups, n_rel)) {
    set_error("fold_tiles: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int64_t nb_max = fold_base_max(n_groups, n_rel);
  Carver cv(workspace);
  int32_t* gstart = cv.take<int32_t>((size_t)n_groups + 1);
  int32_t* bptr = cv.take<int32_t>((size_t)n_rel + 2);
  int32_t* cnt = cv.take<int32_t>((size_t)nb_max + 1);
  int32_t* scan_ws = cv.take<int32_t>(scan_workspace_elems(nb_max + 1));
  hipLaunchKernelGGL(fold_gstart_kernel, dim3(blocks_for(n_edges > 0 ? n_edges : 1, 256)), dim3(256), 0, st, n_rel,
                     rel_ptr, n_groups, gid, gstart);
  KGAT_CHECK_LAUNCH("fold_gstart");
  hipLaunchKernelGGL(fold_base_ptr_kernel, dim3(1), dim3(64), 0, st, n_rel, gptr, bptr, gpt);
  KGAT_CHECK_LAUNCH("fold_base_ptr");
  hipLaunchKernelGGL(fold_count_kernel, dim3(blocks_for(nb_max + 1, 256)), dim3(256), 0, st, n_rel, nb_max, cap, gpt,
                     (const int32_t*)bptr, gptr, (const int32_t*)gstart, cnt);
  KGAT_CHECK_LAUNCH("fold_count");
  const int rc = exclusive_scan_i32(cnt, nb_max + 1, scan_ws, st);
  if (rc != KGAT_OK) return rc;
  const int64_t n_thr = nb_max > n_rel + 1 ? nb_max : n_rel + 1;
  hipLaunchKernelGGL(fold_emit_kernel, dim3(blocks_for(n_thr, 256)), dim3(256), 0, st, n_rel, nb_max, cap, gpt,
                     (const int32_t*)bptr, gptr, (const int32_t*)gstart, (const int32_t*)cnt, tiles, rel_tptr);
  KGAT_CHECK_LAUNCH("fold_emit");
  return KGAT_OK;
}

int kgat_att_pack_records(int64_t n_edges, int n_rel, const int32_t* rel_ptr, const int32_t* gptr,
                          const int32_t* gid, const int32_t* src_g, int groups_per_tile, int32_t* rec_g,
                          kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_edges >= 0 && n_edges < INT32_MAX && n_rel > 0, "att_pack_records: bad size");
  KGAT_CHECK_ARG(groups_per_tile == 16 || groups_per_tile == 32, "att_pack_records: 16 or 32 groups per tile");
  if (n_edges == 0) return KGAT_OK;
  KGAT_CHECK_ARG(rel_ptr && gptr && gid && src_g && rec_g, "att_pack_records: null pointer");
  hipLaunchKernelGGL(att_pack_records_kernel, dim3(blocks_for(n_edges, 256)), dim3(256), 0, as_stream(stream), n_edges,
                     n_rel, rel_ptr, gptr, gid, src_g, rec_g, groups_per_tile);
  KGAT_CHECK_LAUNCH("att_pack_records");
  return KGAT_OK;
}

size_t kgat_fold_tile_parts_workspace_bytes(int64_t t_max) {
  const size_t n = (size_t)(t_max > 0 ? t_max : 0) + 1;
  return align_up(n * 4, 256) + align_up(scan_workspace_elems((int64_t)n) * 4, 256);
}

int kgat_fold_tile_parts(int64_t t_max, int n_rel, const int32_t* tiles, const int32_t* rel_tptr, int n_parts,
                         int cost_tile, int cost_chunk, int cost_relation, int32_t* part_tptr, void* workspace,
                         size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(t_max >= 0 && t_max < INT32_MAX - 1 && n_rel > 0 && n_parts > 0, "fold_tile_parts: bad size");
  KGAT_CHECK_ARG(cost_tile > 0 && cost_chunk >= 0 && cost_relation >= 0 && cost_tile <= 256 && cost_chunk <= 256 &&
                 cost_relation <= 65536, "fold_tile_parts: cost coefficients out of range");
  KGAT_CHECK_ARG(tiles && rel_tptr && part_tptr && workspace, "fold_tile_parts: null pointer");
  if (workspace_bytes < kgat_fold_tile_parts_workspace_bytes(t_max)) {
    set_error("fold_tile_parts: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver cv(workspace);
  int32_t* cost = cv.take<int32_t>((size_t)t_max + 1);
  int32_t* scan_ws = cv.take<int32_t>(scan_workspace_elems(t_max + 1));
  hipLaunchKernelGGL(fold_tile_cost_kernel, dim3(blocks_for(t_max + 1, 256)), dim3(256), 0, st, t_max, n_rel, rel_tptr,
                     reinterpret_cast<const int4*>(tiles), cost_tile, cost_chunk, cost_relation, cost);
  KGAT_CHECK_LAUNCH("fold_tile_cost");
  const int rc = exclusive_scan_i32(cost, t_max + 1, scan_ws, st);
  if (rc != KGAT_OK) return rc;
  hipLaunchKernelGGL(fold_parts_kernel, dim3(blocks_for(n_parts + 1, 256)), dim3(256), 0, st, n_rel, t_max, rel_tptr,
                     (const int32_t*)cost, n_parts, part_tptr);
  KGAT_CHECK_LAUNCH("fold_parts");
  return KGAT_OK;
}

int kgat_invert_permutation(int64_t n, const int32_t* perm, int32_t* inv, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n >= 0, "invert_permutation: negative size");
  if (n == 0) return KGAT_OK;
  KGAT_CHECK_ARG(perm && inv, "invert_permutation: null pointer");
  hipLaunchKernelGGL(invert_perm_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream),
                     n, perm, inv);
  KGAT_CHECK_LAUNCH("invert_perm");
  return KGAT_OK;
}

size_t kgat_row_order_workspace_bytes(int64_t n_rows) {
  return radix_sort_workspace_bytes(n_rows) + align_up((size_t)(n_rows > 0 ? n_rows : 1) * 4, 256);
}

int kgat_row_order_by_degree(int64_t n_rows, const int32_t* indptr, int32_t* order,
                             void* workspace, size_t workspace_bytes, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_rows >= 0 && n_rows < INT32_MAX, "row_order: bad size");
  if (n_rows == 0) return KGAT_OK;
  KGAT_CHECK_ARG(indptr && order && workspace, "row_order: null pointer");
  if (workspace_bytes < kgat_row_order_workspace_bytes(n_rows)) {
    set_error("row_order: workspace too small");
    return KGAT_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver cv(workspace);
  int32_t* key = cv.take<int32_t>((size_t)n_rows);
  hipLaunchKernelGGL(degree_key_kernel, dim3(blocks_for(n_rows, 256)), dim3(256), 0, st, n_rows,
                     indptr, key);
  KGAT_CHECK_LAUNCH("degree_key");
  return radix_sort_index(key, n_rows, 16, order, nullptr, cv.base + cv.off,
                          workspace_bytes - cv.off, st);
}

int kgat_gather_i32(int64_t n, const int32_t* index, const int32_t* in, int32_t* out,
                    kgat_stream_t stream) {
  KGAT_CHECK_ARG(n >= 0, "gather: negative size");
  if (n == 0) return KGAT_OK;
  KGAT_CHECK_ARG(index && in && out, "gather: null pointer");
  launch_gather<int32_t>(n, index, in, out, as_stream(stream));
  KGAT_CHECK_LAUNCH("gather_i32");
  return KGAT_OK;
}

int kgat_gather_f32(int64_t n, const int32_t* index, const float* in, float* out,
                    kgat_stream_t stream) {
  KGAT_CHECK_ARG(n >= 0, "gather: negative size");
  if (n == 0) return KGAT_OK;
  KGAT_CHECK_ARG(index && in && out, "gather: null pointer");
  launch_gather<float>(n, index, in, out, as_stream(stream));
  KGAT_CHECK_LAUNCH("gather_f32");
  return KGAT_OK;
}

}  // extern "C"
