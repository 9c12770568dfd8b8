// pinn_adam_update.h -- one element of torch.optim.Adam (defaults, 01:939 / 954), shared by the optimizer kernels
// (pinn_optim.hip) and the slab reduction that applies the step in the same launch (pinn_train.hip, pinn_mlp_train_step_dev).
// Every multiply-add is an explicit fmaf: the two call sites must round alike whatever the compiler would contract.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace pinn {
__device__ __forceinline__ void adam_update(float& p, const float g, float& m, float& v, const float step_size, const float bc2_sqrt) {
  m = fmaf(0.1f, g - m, m);                          // exp_avg.lerp_(grad, 1 - beta1)
  v = fmaf(0.001f * g, g, v * 0.999f);               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
  const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
  p = fmaf(-step_size, m / denom, p);                // param.addcdiv_(exp_avg, denom, value=-step_size)
}
}  // namespace pinn
