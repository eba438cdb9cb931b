// Charbonnier + edge loss of the training step (reference: utils/loss.py:22-79, CharbonnierLoss with if_aux_loss):
//   L = mean sqrt(d^2 + eps) + r * mean sqrt(lap(d)^2 + eps),   d = x - y,
//   lap(d) = d - G(D(G d)),  G = 5x5 Gaussian [.05 .25 .4 .25 .05]^2 with REPLICATE padding, D = 4x at (even, even) else 0
// (the reference builds lap(x) and lap(y); the operator is linear).  HBM-bound gathers on (planes, H, W) fp32 images:
//   forward:  a1 = 4 (G d) at even positions (quarter size);  ld = d - G(z[a1]);  per-block partial sums of both terms;
//   backward: w = ld / sqrt(ld^2 + eps);  u = 4 (G^T w) at even positions;  dx = gs1 d / sqrt(d^2 + eps) + gs2 (w - G^T(z[u])).
// G^T is the adjoint of the replicate-padded blur: an edge pixel also collects what the padding copied out of it, i.e. the
// 1-D weight of source s on target c is sum_e k(e) [clamp(s + e) == c].
#include "common.h"

namespace {

__device__ __forceinline__ float kg(int i) {  // tap i in [-2, 2]
  return i == 0 ? 0.4f : ((i == 1 || i == -1) ? 0.25f : ((i == 2 || i == -2) ? 0.05f : 0.f));
}
__device__ __forceinline__ int clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }
// adjoint weight of source s on target c for an axis of length n (n >= 3)
__device__ __forceinline__ float wadj(int c, int s, int n) {
  if (c > 0 && c < n - 1) return kg(c - s);
  float w = 0.f;
  if (c == 0) { for (int e = -2; e <= 2; ++e) if (s + e <= 0) w += kg(e); }
  else { for (int e = -2; e <= 2; ++e) if (s + e >= n - 1) w += kg(e); }
  return w;
}

// a1[q] = 4 * sum_f k(f) d(clamp(2q + f))
__global__ __launch_bounds__(256) void lap_even_blur_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ a1,
                                                            long long planes, int H, int W, int H2, int W2) {
  const long long total = planes * H2 * W2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int qx = (int)(i % W2);
    const long long r = i / W2;
    const int qy = (int)(r % H2);
    const long long p = r / H2;
    const float* xp = x + p * H * W;
    const float* yp = y ? y + p * H * W : nullptr;
    float s = 0.f;
#pragma unroll
    for (int fy = -2; fy <= 2; ++fy) {
      const int yy = clampi(2 * qy + fy, H);
      float rs = 0.f;
#pragma unroll
      for (int fx = -2; fx <= 2; ++fx) {
        const int xx = clampi(2 * qx + fx, W);
        const float v = xp[yy * W + xx] - (yp ? yp[yy * W + xx] : 0.f);
        rs += kg(fx) * v;
      }
      s += kg(fy) * rs;
    }
    a1[i] = 4.f * s;
  }
}

// ld = d - G(z[a1]); partial sums of sqrt(d^2 + eps) and sqrt(ld^2 + eps) per block
__global__ __launch_bounds__(256) void lap_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ a1,
                                                      float* __restrict__ ld, float* __restrict__ partial, long long planes, int H, int W, int H2,
                                                      int W2, float eps) {
  __shared__ float red[2 * 256];
  const long long total = planes * H * W;
  float s1 = 0.f, s2 = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int px = (int)(i % W);
    const long long r = i / W;
    const int py = (int)(r % H);
    const long long p = r / H;
    const float* ap = a1 + p * H2 * W2;
    float gz = 0.f;
#pragma unroll
    for (int ey = -2; ey <= 2; ++ey) {
      const int yy = clampi(py + ey, H);
      if (yy & 1) continue;
      float rs = 0.f;
#pragma unroll
      for (int ex = -2; ex <= 2; ++ex) {
        const int xx = clampi(px + ex, W);
        if (!(xx & 1)) rs += kg(ex) * ap[(yy >> 1) * W2 + (xx >> 1)];
      }
      gz += kg(ey) * rs;
    }
    const float d = x[i] - y[i];
    const float l = d - gz;
    ld[i] = l;
    s1 += sqrtf(d * d + eps);
    s2 += sqrtf(l * l + eps);
  }
  red[threadIdx.x] = s1;
  red[256 + threadIdx.x] = s2;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      red[threadIdx.x] += red[threadIdx.x + k];
      red[256 + threadIdx.x] += red[256 + threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0];
    partial[2 * blockIdx.x + 1] = red[256];
  }
}

// u[q] = 4 * (G^T w)(2q),  w = ld / sqrt(ld^2 + eps)
__global__ __launch_bounds__(256) void lap_bwd_even_kernel(const float* __restrict__ ld, float* __restrict__ u, long long planes, int H, int W,
                                                           int H2, int W2, float eps) {
  const long long total = planes * H2 * W2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int qx = (int)(i % W2);
    const long long r = i / W2;
    const int qy = (int)(r % H2);
    const long long p = r / H2;
    const float* lp = ld + p * H * W;
    const int cy = 2 * qy, cx = 2 * qx;
    float s = 0.f;
    for (int sy = max(0, cy - 2); sy <= min(H - 1, cy + 2); ++sy) {
      const float wy = wadj(cy, sy, H);
      float rs = 0.f;
      for (int sx = max(0, cx - 2); sx <= min(W - 1, cx + 2); ++sx) {
        const float l = lp[sy * W + sx];
        rs += wadj(cx, sx, W) * (l / sqrtf(l * l + eps));
      }
      s += wy * rs;
    }
    u[i] = 4.f * s;
  }
}

// dx = gs1 * d / sqrt(d^2 + eps) + gs2 * (w - G^T(z[u]))
__global__ __launch_bounds__(256) void lap_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ ld,
                                                      const float* __restrict__ u, float* __restrict__ dx, long long planes, int H, int W, int H2,
                                                      int W2, float eps, float gs1, float gs2) {
  const long long total = planes * H * W;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int px = (int)(i % W);
    const long long r = i / W;
    const int py = (int)(r % H);
    const long long p = r / H;
    const float* up = u + p * H2 * W2;
    float t = 0.f;
    for (int sy = max(0, py - 2); sy <= min(H - 1, py + 2); ++sy) {
      if (sy & 1) continue;
      const float wy = wadj(py, sy, H);
      float rs = 0.f;
      for (int sx = max(0, px - 2); sx <= min(W - 1, px + 2); ++sx)
        if (!(sx & 1)) rs += wadj(px, sx, W) * up[(sy >> 1) * W2 + (sx >> 1)];
      t += wy * rs;
    }
    const float d = x[i] - y[i];
    const float l = ld[i];
    dx[i] = gs1 * (d / sqrtf(d * d + eps)) + gs2 * (l / sqrtf(l * l + eps) - t);
  }
}

int grid_for(long long n) { return (int)(cdiv64(n, 256) > 16384 ? 16384 : cdiv64(n, 256)); }

}  // namespace

extern "C" int vmg_charbonnier_edge_blocks(int64_t planes, int H, int W) { return grid_for(planes * H * W); }

extern "C" int vmg_charbonnier_edge_fwd(const float* x, const float* y, float* a1, float* ld, float* partial, int64_t planes, int H, int W, float eps,
                                        void* stream) {
  VMG_CHECK(x && y && a1 && ld && partial && planes > 0 && H >= 3 && W >= 3, "charbonnier_edge_fwd: bad arguments (H, W >= 3)");
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lap_even_blur_kernel, dim3(grid_for(planes * H2 * W2)), dim3(256), 0, st, x, y, a1, (long long)planes, H, W, H2, W2);
  VMG_LAUNCH_CHECK();
  hipLaunchKernelGGL(lap_fwd_kernel, dim3(grid_for(planes * H * W)), dim3(256), 0, st, x, y, a1, ld, partial, (long long)planes, H, W, H2, W2, eps);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_charbonnier_edge_bwd(const float* x, const float* y, const float* ld, float* u, float* dx, int64_t planes, int H, int W, float eps,
                                        float gs1, float gs2, void* stream) {
  VMG_CHECK(x && y && ld && u && dx && planes > 0 && H >= 3 && W >= 3, "charbonnier_edge_bwd: bad arguments (H, W >= 3)");
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lap_bwd_even_kernel, dim3(grid_for(planes * H2 * W2)), dim3(256), 0, st, ld, u, (long long)planes, H, W, H2, W2, eps);
  VMG_LAUNCH_CHECK();
  hipLaunchKernelGGL(lap_bwd_kernel, dim3(grid_for(planes * H * W)), dim3(256), 0, st, x, y, ld, u, dx, (long long)planes, H, W, H2, W2, eps, gs1, gs2);
  VMG_LAUNCH_CHECK();
  return 0;
}
