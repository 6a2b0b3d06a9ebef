/* CPU oracle (plain C + OpenMP) for KGAT's propagation path.  TEST INFRASTRUCTURE ONLY.
 *
 * A restatement, in fp32, of what the reference computes on the hot path
 * (SURVEY.md section 8a).  Used (a) by tests as a second, independently written
 * checker next to oracle/kgat_oracle.py, (b) by bench.py's `cpu_baseline` leg as
 * the "DGL-CPU-equivalent restatement" timed on the host cores (kind = "port").
 * Never linked into, loaded by, or called from the product library.
 *
 * Parity status: the dense arithmetic follows /root/reference/models.py and is
 * pinned through tests/golden (generated from that file).  The sparse operators
 * live in the un-vendored `dgl` package (0.4.x); they are restated from DGL's
 * published semantics - parity for those is UNPINNED by any reference-owned test.
 *
 * Structure mirrors DGL 0.4's CPU kernels as far as a destination-major CSR
 * allows: OpenMP parallel-for over rows, fp32 accumulation.  DGL-CPU itself
 * parallelises over source rows with `omp atomic` adds; row-major-by-destination
 * needs no atomics and has a fixed summation order (edge-id order within a row).
 *
 * Build: see oracle/Makefile  (gcc -O3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int kgat_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void kgat_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* COO -> CSR by destination, stable in edge id (reference dataset.py:112-120:
 * edge id = triplet row; DGL in-CSR: row v = edges with dst == v). */
int kgat_oracle_csr_from_coo(int64_t n, int64_t e, const int32_t* src, const int32_t* dst,
                             int32_t* indptr, int32_t* col, int32_t* eid) {
  memset(indptr, 0, sizeof(int32_t) * (size_t)(n + 1));
  for (int64_t i = 0; i < e; ++i) {
    if (dst[i] < 0 || dst[i] >= n || src[i] < 0 || src[i] >= n) return -1;
    indptr[dst[i] + 1]++;
  }
  for (int64_t v = 0; v < n; ++v) indptr[v + 1] += indptr[v];
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!cur) return -2;
  memcpy(cur, indptr, sizeof(int32_t) * (size_t)n);
  for (int64_t i = 0; i < e; ++i) {
    int32_t p = cur[dst[i]]++;
    col[p] = src[i];
    eid[p] = (int32_t)i;
  }
  free(cur);
  return 0;
}

/* Edges grouped by relation, stable (replaces the R filter_edges sweeps of
 * reference models.py:149-150).  Types outside [0,R) go after rel_ptr[R]. */
int kgat_oracle_group_by_relation(int64_t e, int n_rel, const int32_t* etype,
                                  int32_t* rel_ptr, int32_t* perm) {
  int64_t* cnt = (int64_t*)calloc((size_t)n_rel + 2, sizeof(int64_t));
  if (!cnt) return -2;
  for (int64_t i = 0; i < e; ++i) {
    int32_t t = etype[i];
    cnt[(t >= 0 && t < n_rel ? t : n_rel) + 1]++;
  }
  for (int r = 0; r <= n_rel; ++r) cnt[r + 1] += cnt[r];
  for (int r = 0; r <= n_rel; ++r) rel_ptr[r] = (int32_t)cnt[r];
  for (int64_t i = 0; i < e; ++i) {
    int32_t t = etype[i];
    perm[cnt[(t >= 0 && t < n_rel ? t : n_rel)]++] = (int32_t)i;
  }
  free(cnt);
  return 0;
}

/* Attention logits, reference models.py:135-154.
 * logits[e] = sum_j (ent[src e] W_r)_j * tanh((ent[dst e] W_r)_j + rel[r]_j), r = etype[e];
 * edges with r outside [0,R) keep 0. */
void kgat_oracle_att_score_f32(int64_t e, int d, int k, int n_rel, const int32_t* src,
                               const int32_t* dst, const int32_t* etype, const float* ent,
                               const float* W_R, const float* rel, float* logits) {
#pragma omp parallel
  {
    float* tr = (float*)malloc(sizeof(float) * (size_t)k * 2);
    float* hr = tr + k;
#pragma omp for schedule(dynamic, 1024)
    for (int64_t i = 0; i < e; ++i) {
      int32_t r = etype[i];
      if (r < 0 || r >= n_rel) { logits[i] = 0.0f; continue; }
      const float* W = W_R + (size_t)r * d * k;
      const float* xt = ent + (size_t)src[i] * d;
      const float* xh = ent + (size_t)dst[i] * d;
      for (int j = 0; j < k; ++j) { tr[j] = 0.0f; hr[j] = 0.0f; }
      for (int a = 0; a < d; ++a) {
        const float* wr = W + (size_t)a * k;
        float ta = xt[a], ha = xh[a];
        for (int j = 0; j < k; ++j) {
          tr[j] += ta * wr[j];
          hr[j] += ha * wr[j];
        }
      }
      const float* er = rel + (size_t)r * k;
      float acc = 0.0f;
      for (int j = 0; j < k; ++j) acc += tr[j] * tanhf(hr[j] + er[j]);
      logits[i] = acc;
    }
    free(tr);
  }
}

/* The same logits with the inner loops written for the vector units (bench.py's cpu_baseline
 * times this one; kgat_oracle_att_score_f32 above stays the checker and the two are compared
 * before any timing).  Same arithmetic per edge - project tail and head through W_r, add e_r,
 * tanh, dot - with (a) the projection accumulators of a k-block held in vector registers,
 * (b) tanh(x) = 1 - 2 / (exp(2x) + 1) on an inlined polynomial exp (|abs err| ~ 1e-7, the libm
 * call does not vectorise), (c) ISA clones resolved at load time (the .so is built once and
 * travels to hosts with other vector widths). */
static inline float kgat_exp_poly(float x) {
  x = x < -87.0f ? -87.0f : (x > 88.0f ? 88.0f : x);
  float t = x * 1.44269504088896341f;
  float n = (t + 12582912.0f) - 12582912.0f;          /* round to nearest integer */
  float r = (x - n * 0.693359375f) - n * -2.12194440e-4f; /* Cody-Waite: x - n ln 2 */
  float p = 1.0f / 5040.0f;
  p = p * r + 1.0f / 720.0f;
  p = p * r + 1.0f / 120.0f;
  p = p * r + 1.0f / 24.0f;
  p = p * r + 1.0f / 6.0f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  union { int32_t i; float f; } u;
  u.i = ((int32_t)n + 127) << 23;
  return p * u.f;
}

#define KGAT_KB 32 /* k-block: 2 x 32 fp32 accumulators = 4 zmm / 8 ymm */

__attribute__((target_clones("avx512f", "fma", "default")))
void kgat_oracle_att_score_fast_f32(int64_t e, int d, int k, int n_rel, const int32_t* src,
                                    const int32_t* dst, const int32_t* etype, const float* ent,
                                    const float* W_R, const float* rel, float* logits) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < e; ++i) {
    int32_t r = etype[i];
    if (r < 0 || r >= n_rel) { logits[i] = 0.0f; continue; }
    const float* W = W_R + (size_t)r * d * k;
    const float* xt = ent + (size_t)src[i] * d;
    const float* xh = ent + (size_t)dst[i] * d;
    const float* er = rel + (size_t)r * k;
    float acc = 0.0f;
    for (int j0 = 0; j0 < k; j0 += KGAT_KB) {
      int kb = k - j0 < KGAT_KB ? k - j0 : KGAT_KB;
      float tr[KGAT_KB] __attribute__((aligned(64)));
      float hr[KGAT_KB] __attribute__((aligned(64)));
      if (kb == KGAT_KB) {
#pragma omp simd
        for (int j = 0; j < KGAT_KB; ++j) { tr[j] = 0.0f; hr[j] = 0.0f; }
        for (int a = 0; a < d; ++a) {
          const float* wr = W + (size_t)a * k + j0;
          float ta = xt[a], ha = xh[a];
#pragma omp simd
          for (int j = 0; j < KGAT_KB; ++j) {
            tr[j] += ta * wr[j];
            hr[j] += ha * wr[j];
          }
        }
      } else {
        for (int j = 0; j < kb; ++j) { tr[j] = 0.0f; hr[j] = 0.0f; }
        for (int a = 0; a < d; ++a) {
          const float* wr = W + (size_t)a * k + j0;
          float ta = xt[a], ha = xh[a];
          for (int j = 0; j < kb; ++j) {
            tr[j] += ta * wr[j];
            hr[j] += ha * wr[j];
          }
        }
      }
      float part = 0.0f;
#pragma omp simd reduction(+ : part)
      for (int j = 0; j < kb; ++j) {
        float x2 = 2.0f * (hr[j] + er[j0 + j]);
        part += tr[j] * (1.0f - 2.0f / (kgat_exp_poly(x2) + 1.0f));
      }
      acc += part;
    }
    logits[i] = acc;
  }
}

/* edge_softmax over the in-edges of each destination (call site reference
 * models.py:153; DGL 0.4.x semantics: max-subtracted softmax per destination). */
void kgat_oracle_edge_softmax_f32(int64_t n, const int32_t* indptr, const int32_t* eid,
                                  const float* logits, float* out) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t v = 0; v < n; ++v) {
    int32_t b = indptr[v], en = indptr[v + 1];
    if (b == en) continue;
    float m = -INFINITY;
    for (int32_t p = b; p < en; ++p) m = fmaxf(m, logits[eid[p]]);
    float z = 0.0f;
    for (int32_t p = b; p < en; ++p) z += expf(logits[eid[p]] - m);
    for (int32_t p = b; p < en; ++p) out[eid[p]] = expf(logits[eid[p]] - m) / z;
  }
}

/* update_all(u_mul_e, sum), reference models.py:63.
 * out[v,:] = sum over in-edges (edge-id order) of w[eid] * X[col,:]; empty rows = 0.
 * `mul_self` != 0 additionally multiplies the row by X[v,:] (the h * h_neighbor
 * product of reference models.py:66). */
void kgat_oracle_spmm_f32(int64_t n, int d, const int32_t* indptr, const int32_t* col,
                          const int32_t* eid, const float* X, const float* w, float* out,
                          int mul_self) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t v = 0; v < n; ++v) {
    float* o = out + (size_t)v * d;
    for (int j = 0; j < d; ++j) o[j] = 0.0f;
    for (int32_t p = indptr[v]; p < indptr[v + 1]; ++p) {
      float we = w[eid ? eid[p] : p];
      const float* x = X + (size_t)col[p] * d;
      for (int j = 0; j < d; ++j) o[j] = fmaf(we, x[j], o[j]);
    }
    if (mul_self) {
      const float* xs = X + (size_t)v * d;
      for (int j = 0; j < d; ++j) o[j] *= xs[j];
    }
  }
}

/* KGATConv dense part, reference models.py:66: leaky_relu_{slope}((h*h_N) W2^T);
 * W2 is (d_out, d_in) row-major (nn.Linear weight), no bias. */
void kgat_oracle_bi_interaction_f32(int64_t n, int d_in, int d_out, const float* h,
                                    const float* h_n, const float* W2, float slope,
                                    float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < n; ++v) {
    const float* a = h + (size_t)v * d_in;
    const float* b = h_n + (size_t)v * d_in;
    float* o = out + (size_t)v * d_out;
    for (int j = 0; j < d_out; ++j) {
      const float* wr = W2 + (size_t)j * d_in;
      float acc = 0.0f;
      for (int i = 0; i < d_in; ++i) acc += (a[i] * b[i]) * wr[i];
      o[j] = acc >= 0.0f ? acc : acc * slope;
    }
  }
}

/* F.normalize(h, p=2, dim=1, eps=1e-12), reference models.py:165. */
void kgat_oracle_l2_normalize_f32(int64_t n, int d, const float* x, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < n; ++v) {
    const float* a = x + (size_t)v * d;
    float s = 0.0f;
    for (int j = 0; j < d; ++j) s += a[j] * a[j];
    float nrm = sqrtf(s);
    if (nrm < 1e-12f) nrm = 1e-12f;
    for (int j = 0; j < d; ++j) out[(size_t)v * d + j] = a[j] / nrm;
  }
}
