// Sliding-window inference accumulators (tools/Tester.py:107-177 of the reference): HBM-bound elementwise kernels.
//   vmg_tile_accumulate: E[region] += patch * m,  Wt[region] += m, with m the tile's border mask (rows / columns within the
//                        given margins of a side that has a neighbouring tile are dropped) -- one pass over the patch
//                        instead of the reference's four in-place zeroings, a ones tensor and two adds;
//   vmg_tile_finalize:   E / Wt, optionally clamped to [0,1], scaled by 255 and rounded half-to-even to uint8
//                        (tools/Tester.py:139, :249-250).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void tile_accumulate_kernel(const T* __restrict__ patch, float* __restrict__ E, float* __restrict__ Wt,
                                                              long long planes, int ph, int pw, int EH, int EW, int oh, int ow, int top,
                                                              int bottom, int left, int right) {
  const long long total = planes * ph * pw;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % pw);
    const long long r = i / pw;
    const int y = (int)(r % ph);
    const long long p = r / ph;
    const bool keep = y >= top && y < ph - bottom && x >= left && x < pw - right;
    if (!keep) continue;  // a dropped pixel adds 0 to both maps
    const long long o = (p * EH + (oh + y)) * EW + (ow + x);
    E[o] += to_f32(patch[i]);
    Wt[o] += 1.0f;
  }
}

__global__ __launch_bounds__(256) void tile_finalize_kernel(const float* __restrict__ E, const float* __restrict__ Wt, float* __restrict__ outf,
                                                            unsigned char* __restrict__ outu, long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float q = E[i] / Wt[i];
    if (outf) outf[i] = q;
    if (outu) {
      const float c = fminf(fmaxf(q, 0.f), 1.f);
      outu[i] = (unsigned char)rintf(c * 255.0f);  // numpy round = half to even = rintf in the default rounding mode
    }
  }
}

}  // namespace

extern "C" int vmg_tile_accumulate(int dtype, const void* patch, float* E, float* Wt, int64_t planes, int ph, int pw, int EH, int EW, int oh,
                                   int ow, int top, int bottom, int left, int right, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "tile_accumulate: bad dtype");
  VMG_CHECK(patch && E && Wt && planes > 0 && ph > 0 && pw > 0, "tile_accumulate: bad arguments");
  VMG_CHECK(oh >= 0 && ow >= 0 && oh + ph <= EH && ow + pw <= EW, "tile_accumulate: the tile [%d+%d, %d+%d] leaves the %d x %d canvas", oh, ph, ow, pw, EH, EW);
  VMG_CHECK(top >= 0 && bottom >= 0 && left >= 0 && right >= 0, "tile_accumulate: negative margin");
  const long long total = planes * ph * pw;
  const int blocks = (int)(cdiv64(total, 256) > 16384 ? 16384 : cdiv64(total, 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(tile_accumulate_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)patch, E, Wt, (long long)planes, ph, pw, EH, EW,
                       oh, ow, top, bottom, left, right);
  else
    hipLaunchKernelGGL(tile_accumulate_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)patch, E, Wt, (long long)planes, ph, pw, EH,
                       EW, oh, ow, top, bottom, left, right);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_tile_finalize(const float* E, const float* Wt, float* out_f32, unsigned char* out_u8, int64_t n, void* stream) {
  VMG_CHECK(E && Wt && n > 0 && (out_f32 || out_u8), "tile_finalize: bad arguments");
  const int blocks = (int)(cdiv64(n, 256) > 16384 ? 16384 : cdiv64(n, 256));
  hipLaunchKernelGGL(tile_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, E, Wt, out_f32, out_u8, (long long)n);
  VMG_LAUNCH_CHECK();
  return 0;
}
