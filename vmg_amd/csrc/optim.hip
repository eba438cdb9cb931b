// AdamW over a flat fp32 parameter segment (the reference trains with torch.optim.AdamW, tools/Trainer.py:86-105): one
// HBM-bound pass over (param, grad, exp_avg, exp_avg_sq) -- 28 bytes per parameter -- instead of a multi-tensor launch per
// few dozen of the model's 560 tensors.  Same arithmetic and order as torch.optim.AdamW (no amsgrad, no maximize):
//   p *= 1 - lr*wd;  m = m + (1-b1)*(g - m);  v = b2*v + (1-b2)*g*g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// The per-step scalars (lr, weight decay, bias corrections) come from DEVICE memory, so a captured hipGraph replays the
// kernel with fresh values.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, const float* __restrict__ hyper, float b1, float b2,
                                                         float eps) {
  const float lr = hyper[0], wd = hyper[1], bc1 = hyper[2], sq_bc2 = hyper[3];
  const float decay = 1.0f - lr * wd, step_size = lr / bc1;
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* G = &gg.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float q = P[e] * decay;
      const float mo = M[e] + (1.0f - b1) * (G[e] - M[e]);
      const float vo = b2 * V[e] + (1.0f - b2) * (G[e] * G[e]);
      q -= step_size * (mo / (sqrtf(vo) / sq_bc2 + eps));
      P[e] = q; M[e] = mo; V[e] = vo;
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  for (long long i = (n4 << 2) + blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float q = p[i] * decay;
    const float mo = m[i] + (1.0f - b1) * (g[i] - m[i]);
    const float vo = b2 * v[i] + (1.0f - b2) * (g[i] * g[i]);
    q -= step_size * (mo / (sqrtf(vo) / sq_bc2 + eps));
    p[i] = q; m[i] = mo; v[i] = vo;
  }
}

}  // namespace

extern "C" int vmg_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1, float beta2, float eps,
                              void* stream) {
  VMG_CHECK(p && g && m && v && hyper && n > 0, "adamw_flat: bad arguments");
  VMG_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "adamw_flat: segments must be 16-byte aligned");
  const long long work = (n + 3) / 4;
  const int blocks = (int)(cdiv64(work, 256) > 8192 ? 8192 : cdiv64(work, 256));
  hipLaunchKernelGGL(adamw_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, hyper, beta1, beta2, eps);
  VMG_LAUNCH_CHECK();
  return 0;
}
