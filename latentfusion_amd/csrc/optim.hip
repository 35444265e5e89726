// Batched per-sample optimiser step for the N pose hypotheses (gfx950).
// Replaces N x 3 tiny torch.optim launches per iteration (GradientPoseEstimator,
// latentfusion/pose/estimation.py:579-594,664-666).  Rows of `params` are independent optimisers
// sharing the step count; `step_size` / `lr` carry the per-row ReduceLROnPlateau state.  The
// arithmetic follows torch.optim.{Adam, AdamW} (defaults: betas 0.9/0.999, eps 1e-8) operation by
// operation so that trajectories match the reference.
#include "lf_common.h"

namespace {
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, const float* __restrict__ step_size,
                                 const float* __restrict__ lr, float bc2_sqrt, float beta1, float beta2, float eps,
                                 float weight_decay, int N, int P) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * P) return;
  const int n = idx / P;
  float pv = p[idx];
  const float gv = g[idx];
  if (weight_decay != 0.f) pv *= (1.f - lr[n] * weight_decay);          // AdamW decoupled decay
  const float mv = m[idx] + (gv - m[idx]) * (1.f - beta1);                // exp_avg.lerp_(grad, 1 - beta1)
  const float vv = v[idx] * beta2 + ((1.f - beta2) * gv) * gv;           // mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(vv) / bc2_sqrt + eps;
  pv -= step_size[n] * (mv / denom);
  p[idx] = pv; m[idx] = mv; v[idx] = vv;
}
}  // namespace

extern "C" int lf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                            const float* step_size, const float* lr, float bias_correction2_sqrt,
                            float beta1, float beta2, float eps, float weight_decay, int N, int P, void* stream) {
  lf_clear_error();
  if (N <= 0 || P <= 0) return LF_EINVAL;
  hipLaunchKernelGGL(adam_step_kernel, dim3((N * P + 127) / 128), dim3(128), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, step_size, lr, bias_correction2_sqrt, beta1, beta2, eps, weight_decay, N, P);
  return lf_launch_status();
}
