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

// The glue of one pyramid level (models/vmg.py:72-85: `torch.cat([ref[level], warp(supp[level], flow_up), flow_up], 1)` and `flow = flow_up +
// basic_module(...)`) on the 8-channel pixels this port uses (RGB in channels 0..2, zeros behind): one pass builds the operand
// [ref RGB | warped RGB | flow] (a 16-byte vector per pixel in bf16), one pass splits its gradient into the warped image's (8 channels, zeros behind
// the RGB) and the flow's (fp32).  torch spelled it cast + cat forward, two slice-backward zero fills, two copies and a cast backward: 8 launches
// per level on tensors of a few hundred kB.
template <typename T>
struct alignas(8 * sizeof(T) > 16 ? 16 : 8 * sizeof(T)) Px8 { T v[8]; };

template <typename T>
__global__ void spy_operand_kernel(const Px8<T>* __restrict__ ref, const Px8<T>* __restrict__ warped, const float2* __restrict__ up,
                                   Px8<T>* __restrict__ out, long long npix) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const Px8<T> r = ref[i], w = warped[i];
    const float2 f = up[i];
    Px8<T> o;
    o.v[0] = r.v[0]; o.v[1] = r.v[1]; o.v[2] = r.v[2];
    o.v[3] = w.v[0]; o.v[4] = w.v[1]; o.v[5] = w.v[2];
    o.v[6] = from_f32<T>(f.x); o.v[7] = from_f32<T>(f.y);
    out[i] = o;
  }
}

template <typename T>
__global__ void spy_operand_bwd_kernel(const Px8<T>* __restrict__ dx8, Px8<T>* __restrict__ dwarped, float2* __restrict__ dup, long long npix) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const Px8<T> d = dx8[i];
    Px8<T> o;
    o.v[0] = d.v[3]; o.v[1] = d.v[4]; o.v[2] = d.v[5];
#pragma unroll
    for (int e = 3; e < 8; ++e) o.v[e] = from_f32<T>(0.f);
    dwarped[i] = o;
    dup[i] = float2{to_f32(d.v[6]), to_f32(d.v[7])};
  }
}

template <typename T>
__global__ void spy_flow_add_kernel(const float* __restrict__ up, const T* __restrict__ res, float* __restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = up[i] + to_f32(res[i]);
}

// SPyNet's input normalisation (models/vmg.py:104-106 `(x - mean) / std`) written straight into this port's pixel layout: (n, 3, h, w) fp32 -> (n, h, w, 8) T with zeros
// behind the RGB -- torch spelled it sub, div, permute + pad, cast: four launches per image batch.
template <typename T>
__global__ void spy_prep_kernel(const float* __restrict__ img, const float* __restrict__ mean, const float* __restrict__ stdv, Px8<T>* __restrict__ out, long long n, int hw) {
  const float m0 = mean[0], m1 = mean[1], m2 = mean[2], s0 = stdv[0], s1 = stdv[1], s2 = stdv[2];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n * hw; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / hw;
    const int p = (int)(i - b * hw);
    const float* src = img + b * 3 * hw + p;
    Px8<T> o;
    o.v[0] = from_f32<T>(__fdiv_rn(__fsub_rn(src[0], m0), s0));
    o.v[1] = from_f32<T>(__fdiv_rn(__fsub_rn(src[hw], m1), s1));
    o.v[2] = from_f32<T>(__fdiv_rn(__fsub_rn(src[2 * hw], m2), s2));
#pragma unroll
    for (int e = 3; e < 8; ++e) o.v[e] = from_f32<T>(0.f);
    out[i] = o;
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

extern "C" int vmg_spy_operand_fwd(int dtype, const void* ref, const void* warped, const float* up, void* out, int64_t npix, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "spy_operand_fwd: bad dtype");
  VMG_CHECK(ref && warped && up && out && npix > 0, "spy_operand_fwd: bad arguments");
  VMG_CHECK(((uintptr_t)ref | (uintptr_t)warped | (uintptr_t)out) % 16 == 0 && (uintptr_t)up % 8 == 0, "spy_operand_fwd: 16-byte aligned 8-channel tensors, 8-byte aligned flow");
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(spy_operand_kernel<bf16>, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, (const Px8<bf16>*)ref, (const Px8<bf16>*)warped,
                       (const float2*)up, (Px8<bf16>*)out, (long long)npix);
  else
    hipLaunchKernelGGL(spy_operand_kernel<float>, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, (const Px8<float>*)ref, (const Px8<float>*)warped,
                       (const float2*)up, (Px8<float>*)out, (long long)npix);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_spy_operand_bwd(int dtype, const void* dx8, void* dwarped, float* dup, int64_t npix, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "spy_operand_bwd: bad dtype");
  VMG_CHECK(dx8 && dwarped && dup && npix > 0, "spy_operand_bwd: bad arguments");
  VMG_CHECK(((uintptr_t)dx8 | (uintptr_t)dwarped) % 16 == 0 && (uintptr_t)dup % 8 == 0, "spy_operand_bwd: 16-byte aligned 8-channel tensors, 8-byte aligned flow gradient");
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(spy_operand_bwd_kernel<bf16>, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, (const Px8<bf16>*)dx8, (Px8<bf16>*)dwarped, (float2*)dup,
                       (long long)npix);
  else
    hipLaunchKernelGGL(spy_operand_bwd_kernel<float>, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, (const Px8<float>*)dx8, (Px8<float>*)dwarped, (float2*)dup,
                       (long long)npix);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_spy_flow_add(int dtype, const float* up, const void* res, float* out, int64_t n, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "spy_flow_add: bad dtype");
  VMG_CHECK(up && res && out && n > 0, "spy_flow_add: bad arguments");
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(spy_flow_add_kernel<bf16>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, up, (const bf16*)res, out, (long long)n);
  else
    hipLaunchKernelGGL(spy_flow_add_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, up, (const float*)res, out, (long long)n);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_spy_prep(int dtype, const float* img, const float* mean, const float* stdv, void* out, int64_t n, int h, int w, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "spy_prep: bad dtype");
  VMG_CHECK(img && mean && stdv && out && n > 0 && h > 0 && w > 0 && (uintptr_t)out % 16 == 0, "spy_prep: bad arguments");
  const long long total = (long long)n * h * w;
  if (dtype == VMG_BF16) hipLaunchKernelGGL(spy_prep_kernel<bf16>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, img, mean, stdv, (Px8<bf16>*)out, (long long)n, h * w);
  else hipLaunchKernelGGL(spy_prep_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, img, mean, stdv, (Px8<float>*)out, (long long)n, h * w);
  VMG_LAUNCH_CHECK();
  return 0;
}
