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

// ---- clip_grad_norm_ over the flat gradient buffer (tools/Trainer.py:141-143, 166-167: torch.nn.utils.clip_grad_norm_, norm_type 2) ----
// Two launches, no atomics: per-block sums of squares over fixed contiguous chunks (fixed order inside a block: a strided walk, wave
// shuffles, the four waves through LDS), then every block of the second launch adds the partials in index order -- in double -- and
// scales its own chunk by min(1, max_norm / (norm + 1e-6)).  The result is the same bits on every run and for any gradient-arrival order.
constexpr int CLIP_BLOCKS = 1024;

__device__ __forceinline__ double block_sum_double(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ partial) {
  __shared__ double sm[4];
  const long long n4 = n >> 2;
  const long long per = (n4 + CLIP_BLOCKS - 1) / CLIP_BLOCKS;
  const long long lo = blockIdx.x * per, hi = (lo + per < n4) ? lo + per : n4;
  float acc = 0.f;
  double total = 0.0;
  int k = 0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    if (++k == 64) { total += acc; acc = 0.f; k = 0; }  // (fp32 runs of at most 256 squares, summed in double)
  }
  total += acc;
  if (blockIdx.x == CLIP_BLOCKS - 1)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) total += (double)g[i] * g[i];
  const double s = block_sum_double(total, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void grad_clip_scale_kernel(float* __restrict__ g, long long n, const double* __restrict__ partial, float max_norm,
                                                              float* __restrict__ norm_out) {
  __shared__ double sm[4];
  double t = 0.0;
  for (int i = threadIdx.x; i < CLIP_BLOCKS; i += 256) t += partial[i];
  const double total = block_sum_double(t, sm);
  const float norm = (float)sqrt(total);
  float coef = max_norm / (norm + 1e-6f);
  coef = coef < 1.0f ? coef : 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) { norm_out[0] = norm; norm_out[1] = coef; }
  if (coef >= 1.0f) return;
  const long long n4 = n >> 2;
  const long long per = (n4 + CLIP_BLOCKS - 1) / CLIP_BLOCKS;
  const long long lo = blockIdx.x * per, hi = (lo + per < n4) ? lo + per : n4;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    float4 v = reinterpret_cast<float4*>(g)[i];
    v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
    reinterpret_cast<float4*>(g)[i] = v;
  }
  if (blockIdx.x == CLIP_BLOCKS - 1)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) g[i] *= coef;
}

}  // namespace

extern "C" int64_t vmg_grad_clip_ws_bytes() { return (int64_t)CLIP_BLOCKS * sizeof(double); }

extern "C" int vmg_grad_clip_norm(float* g, int64_t n, float max_norm, void* workspace, float* norm_out, void* stream) {
  VMG_CHECK(g && workspace && norm_out && n > 0 && max_norm > 0.f, "grad_clip_norm: bad arguments");
  VMG_CHECK(((uintptr_t)g % 16 == 0) && ((uintptr_t)workspace % 8 == 0), "grad_clip_norm: the gradient buffer must be 16-byte aligned");
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, (long long)n, (double*)workspace);
  VMG_LAUNCH_CHECK();
  hipLaunchKernelGGL(grad_clip_scale_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, (long long)n, (const double*)workspace, max_norm,
                     norm_out);
  VMG_LAUNCH_CHECK();
  return 0;
}

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
