// Fused Adam over a flat fp32 parameter arena (one launch per optimiser instead of ~950 tensors).
// Semantics = torch.optim.Adam(weight_decay=wd) as used by the reference
// (train_deform_semantic.py:271-288): L2 decay folded into the gradient, bias-corrected moments,
// denom = sqrt(v)/sqrt(1-b2^t) + eps.  grad_scale folds the 1/world_size of the gradient average.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n4, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float grad_scale) {
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * grad_scale;
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    gv += wd * pv;
    mv = b1 * mv + (1.f - b1) * gv;
    vv = b2 * vv + (1.f - b2) * gv * gv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pv[j] -= step_size * (mv[j] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
}
}  // namespace

extern "C" int fs_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int step, float grad_scale, hipStream_t stream) {
  FS_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  int blocks = cdiv(n / 4, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n / 4, lr, beta1, beta2, eps, weight_decay, bc1,
                     bc2_sqrt, grad_scale);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
