// Small image-pyramid kernels of SPyNet (reference: models/vmg.py:39-123): 2x2 average pooling and the x2 bilinear
// (align_corners=True) up-sampling of the flow between pyramid levels, channels-last.  HBM-bound, a few hundred kB each; the
// 7x7 convolutions are vmg_conv_fwd (KS = 7), the warps vmg_warp_bilinear_fwd / _bwd.
#include "common.h"

namespace {

template <typename T>
__global__ void avgpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c) {
  // F.avg_pool2d(x, 2, 2, count_include_pad=False) on (n, h, w, c) -> (n, h/2, w/2, c)  (models/vmg.py:66-70)
  const int ho = h >> 1, wo = w >> 1;
  const long long total = (long long)n * ho * wo * c;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c);
    long long t = i / c;
    const int xo = (int)(t % wo);
    t /= wo;
    const int yo = (int)(t % ho);
    const long long nn = t / ho;
    const T* p = x + ((nn * h + 2 * yo) * w + 2 * xo) * c + cc;
    const float s = to_f32(p[0]) + to_f32(p[c]) + to_f32(p[(long long)w * c]) + to_f32(p[(long long)w * c + c]);
    y[i] = from_f32<T>(s * 0.25f);
  }
}

// torch's bilinear source index for align_corners=True: src = dst * (in - 1) / (out - 1)
__device__ __forceinline__ void ac_src(int dst, float ratio, int in, int& i0, int& i1, float& l1) {
  const float s = ratio * (float)dst;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ void upsample2x_ac_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int h, int w, int c, float scale) {
  // dst (n, 2h, 2w, c) = scale * bilinear_ac(src (n, h, w, c))
  const int ho = 2 * h, wo = 2 * w;
  const float rh = h > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f, rw = w > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
  const long long total = (long long)n * ho * wo * c;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c);
    long long t = i / c;
    const int xo = (int)(t % wo);
    t /= wo;
    const int yo = (int)(t % ho);
    const long long nn = t / ho;
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(yo, rh, h, y0, y1, ly);
    ac_src(xo, rw, w, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const long long b = nn * h;
    const float v = hy * (hx * src[((b + y0) * w + x0) * c + cc] + lx * src[((b + y0) * w + x1) * c + cc]) +
                    ly * (hx * src[((b + y1) * w + x0) * c + cc] + lx * src[((b + y1) * w + x1) * c + cc]);
    dst[i] = scale * v;
  }
}

// The adjoint as a GATHER: dst (n, h, w, c) = scale * U^T src (n, 2h, 2w, c) -- input pixel (y, x) collects, in a fixed order, from the few
// output pixels whose interpolation touches it.  No atomics (bit-reproducible) and no zero-fill: round 3's version was hipMemsetAsync +
// float atomics, and the memset NODE that a stream capture records for it did not clear buffers of a few KB on replay (SPyNet's coarse
// pyramid levels accumulated onto whatever the graph's pool had left there: NaN gradients in a replayed step, tools/dbg/replay_stress2.py).
// Output row yo touches input rows y0 = floor(yo * r) and y0 + 1, so the candidates for input row y are the yo with yo * r in (y - 1, y + 1).
__device__ __forceinline__ void ac_range(int i, float ratio, int in, int out, int& lo, int& hi) {
  if (in <= 1 || ratio <= 0.f) { lo = 0; hi = out - 1; return; }
  lo = (int)floorf((float)(i - 1) / ratio) - 1;
  hi = (int)ceilf((float)(i + 1) / ratio) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out - 1 ? out - 1 : hi;
}
__device__ __forceinline__ float ac_weight(int dst, float ratio, int in, int i) {  // weight of input index i in output index dst
  int i0, i1;
  float l1;
  ac_src(dst, ratio, in, i0, i1, l1);
  return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}
__global__ void upsample2x_ac_bwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int h, int w, int c, float scale) {
  const int ho = 2 * h, wo = 2 * w;
  const float rh = h > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f, rw = w > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
  const long long total = (long long)n * h * w * c;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c);
    long long t = i / c;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h);
    const long long nn = t / h;
    int ylo, yhi, xlo, xhi;
    ac_range(y, rh, h, ho, ylo, yhi);
    ac_range(x, rw, w, wo, xlo, xhi);
    float acc = 0.f;
    for (int yo = ylo; yo <= yhi; ++yo) {
      const float wy = ac_weight(yo, rh, h, y);
      if (wy == 0.f) continue;
      const float* row = src + ((nn * ho + yo) * wo) * (long long)c + cc;
      float racc = 0.f;
      for (int xo = xlo; xo <= xhi; ++xo) {
        const float wx = ac_weight(xo, rw, w, x);
        if (wx != 0.f) racc += wx * row[(long long)xo * c];
      }
      acc += wy * racc;
    }
    dst[i] = scale * acc;
  }
}

int blocks_for(long long total) { return (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256)); }

}  // namespace

extern "C" int vmg_avgpool2_nhwc(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "avgpool2: bad dtype");
  VMG_CHECK(x && y && n > 0 && h >= 2 && w >= 2 && c > 0, "avgpool2: bad arguments");
  const long long total = (long long)n * (h / 2) * (w / 2) * c;
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(avgpool2_kernel<bf16>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n, h, w, c);
  else
    hipLaunchKernelGGL(avgpool2_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n, h, w, c);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_upsample2x_ac_fwd(const float* x, float* y, int n, int h, int w, int c, float scale, void* stream) {
  VMG_CHECK(x && y && n > 0 && h > 0 && w > 0 && c > 0, "upsample2x_fwd: bad arguments");
  const long long total = (long long)n * 4 * h * w * c;
  hipLaunchKernelGGL(upsample2x_ac_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, n, h, w, c, scale);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_upsample2x_ac_bwd(const float* dy, float* dx, int n, int h, int w, int c, float scale, void* stream) {
  VMG_CHECK(dy && dx && n > 0 && h > 0 && w > 0 && c > 0, "upsample2x_bwd: bad arguments");
  const long long total = (long long)n * h * w * c;
  hipLaunchKernelGGL(upsample2x_ac_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, n, h, w, c, scale);
  VMG_LAUNCH_CHECK();
  return 0;
}
