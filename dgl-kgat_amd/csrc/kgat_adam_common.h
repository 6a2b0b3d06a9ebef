// Adam's per-element arithmetic, shared by the one-launch optimiser (kgat_optim.hip) and the fused KG iteration
// (kgat_transr.hip).  Not part of the ABI.
#pragma once
#include "kgat_common.h"

namespace kgat {

// One element, rounding where torch's multi-tensor Adam rounds (each _foreach_* call is a kernel of its own, so its
// result is rounded to fp32 before the next one reads it; inside lerp / addcmul / addcdiv the compiler contracts
// `a + s * x` into one fma): contraction is switched off here and the fmas are written out.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2,
                                         float step_size, float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
  // _foreach_lerp_(exp_avg, grad, 1 - beta1).  torch's lerp takes `self + weight * (end - self)` for weight < 0.5 and
  // `end - (end - self) * (1 - weight)` from 0.5 on (ATen/native/Lerp.h), i.e. for beta1 <= 0.5: both arms, so that
  // "bit-identical to torch.optim.Adam" holds for every beta the constructor accepts.
  const float dm = g - m;
  m = w1 < 0.5f ? fmaf(w1, dm, m) : fmaf(-dm, 1.0f - w1, g);   // (the device compiler contracts both arms)
  const float t = v * beta2;          // _foreach_mul_(exp_avg_sq, beta2)
  const float gg = g * g;
  v = fmaf(w2, gg, t);                // _foreach_addcmul_(exp_avg_sq, grad, grad, 1 - beta2)
  float d = sqrtf(v);                 // _foreach_sqrt
  d = d / bc2_sqrt;                   // _foreach_div_(.., sqrt(1 - beta2^t))
  d = d + eps;                        // _foreach_add_(.., eps)
  const float q = m / d;
  p = fmaf(-step_size, q, p);         // _foreach_addcdiv_(param, exp_avg, denom, -lr / (1 - beta1^t))
}


}  // namespace kgat
