// Adam for the training loop of reference kgat.py:85 (optim.Adam(model.parameters(), lr)) as ONE launch over
// every parameter tensor that has a gradient: the reference's optimiser is torch's dense Adam, so every row of
// the 159k x 64 embedding table moves in every step (the moments decay even where the gradient is zero) - a
// streaming pass over p, g, m, v (28 bytes per element), which torch takes as ten multi-tensor launches (0.13 ms
// of the 0.31 ms KG step) and this takes as one.
//
// Per element, in fp32, the operations of torch.optim.Adam (amsgrad = False, weight_decay = 0, maximize = False),
// in torch's order:
//   m  <- m + (1 - beta1) (g - m)                    (Tensor.lerp_)
//   v  <- v beta2 + (1 - beta2) g g                  (mul_, addcmul_)
//   p  <- p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with the step-dependent scalars formed in double on the host exactly as torch forms them.
#include "kgat_adam_common.h"

namespace kgat {

constexpr int kAdamMaxTensors = 16;
constexpr int kAdamChunk = 4096;  // elements per workgroup: 256 threads x 4 float4

struct AdamArgs {
  float* p[kAdamMaxTensors];
  float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  int64_t n[kAdamMaxTensors];
  int first_block[kAdamMaxTensors + 1];
  float step_size[kAdamMaxTensors];   // lr / (1 - beta1^t)
  float bc2_sqrt[kAdamMaxTensors];    // sqrt(1 - beta2^t)
  int count;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a, float w1, float beta2, float w2, float eps,
                                                   int zero_grads) {
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
  const int64_t base = (int64_t)(blockIdx.x - a.first_block[t]) * kAdamChunk;
  float* __restrict__ p = a.p[t];
  float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const int64_t n = a.n[t];
  const float ss = a.step_size[t], bs = a.bc2_sqrt[t];
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
  for (int k = 0; k < kAdamChunk / 1024; ++k) {
    const int64_t i = base + (int64_t)k * 1024 + threadIdx.x * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      float4 pp = *reinterpret_cast<const float4*>(p + i);
      const float4 gg = *reinterpret_cast<const float4*>(g + i);
      float4 mm = *reinterpret_cast<const float4*>(m + i);
      float4 vv = *reinterpret_cast<const float4*>(v + i);
      adam_one(pp.x, gg.x, mm.x, vv.x, w1, beta2, w2, ss, bs, eps);
      adam_one(pp.y, gg.y, mm.y, vv.y, w1, beta2, w2, ss, bs, eps);
      adam_one(pp.z, gg.z, mm.z, vv.z, w1, beta2, w2, ss, bs, eps);
      adam_one(pp.w, gg.w, mm.w, vv.w, w1, beta2, w2, ss, bs, eps);
      *reinterpret_cast<float4*>(p + i) = pp;
      *reinterpret_cast<float4*>(m + i) = mm;
      *reinterpret_cast<float4*>(v + i) = vv;
      if (zero_grads) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int64_t j = i; j < i + 4 && j < n; ++j) {
        float pp = p[j], mm = m[j], vv = v[j];
        adam_one(pp, g[j], mm, vv, w1, beta2, w2, ss, bs, eps);
        p[j] = pp; m[j] = mm; v[j] = vv;
        if (zero_grads) g[j] = 0.f;
      }
    }
  }
}

}  // namespace kgat

using namespace kgat;

extern "C" {

int kgat_adam_max_tensors(void) { return kAdamMaxTensors; }

int kgat_adam_step_f32(int n_tensors, const int64_t* sizes_host, float* const* params_host, float* const* grads_host,
                       float* const* exp_avg_host, float* const* exp_avg_sq_host, const int64_t* steps_host, double lr,
                       double beta1, double beta2, double eps, int zero_grads, kgat_stream_t stream) {
  KGAT_CHECK_ARG(n_tensors >= 0 && n_tensors <= kAdamMaxTensors, "adam_step: %d tensors (at most %d per call)",
                 n_tensors, kAdamMaxTensors);
  if (n_tensors == 0) return KGAT_OK;
  KGAT_CHECK_ARG(sizes_host && params_host && grads_host && exp_avg_host && exp_avg_sq_host && steps_host,
                 "adam_step: null pointer");
  KGAT_CHECK_ARG(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "adam_step: bad hyperparameter");
  AdamArgs a;
  a.count = 0;
  int blocks = 0;
  for (int t = 0; t < n_tensors; ++t) {
    KGAT_CHECK_ARG(sizes_host[t] >= 0 && steps_host[t] >= 1, "adam_step: tensor %d: bad size or step", t);
    if (sizes_host[t] == 0) continue;
    KGAT_CHECK_ARG(params_host[t] && grads_host[t] && exp_avg_host[t] && exp_avg_sq_host[t],
                   "adam_step: tensor %d: null pointer", t);
    const int c = a.count++;
    a.p[c] = params_host[t]; a.g[c] = grads_host[t]; a.m[c] = exp_avg_host[t]; a.v[c] = exp_avg_sq_host[t];
    a.n[c] = sizes_host[t];
    a.first_block[c] = blocks;
    const int64_t nb = (sizes_host[t] + kAdamChunk - 1) / kAdamChunk;
    KGAT_CHECK_ARG(nb + blocks < (int64_t)1 << 31, "adam_step: too many elements");
    blocks += (int)nb;
    // torch.optim.adam._single_tensor_adam / _multi_tensor_adam: python floats (double), then fp32 in the kernels
    const double bc1 = 1.0 - pow(beta1, (double)steps_host[t]);
    const double bc2 = 1.0 - pow(beta2, (double)steps_host[t]);
    a.step_size[c] = (float)(lr / bc1);
    a.bc2_sqrt[c] = (float)sqrt(bc2);
  }
  if (a.count == 0) return KGAT_OK;
  a.first_block[a.count] = blocks;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, zero_grads);
  KGAT_CHECK_LAUNCH("adam_step");
  return KGAT_OK;
}

}  // extern "C"
