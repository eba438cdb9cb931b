// What bounds a row-normalising streaming kernel at (M = 114 688, C = 144) bf16?  Variants of a copy with the LayerNorm kernel's access
// pattern.  hipcc --offload-arch=gfx950 -O3 -o ln_probe ln_probe.cpp && ./ln_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef unsigned short bf16;
struct alignas(16) V8 { bf16 v[8]; };
__device__ __forceinline__ float f(bf16 b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ bf16 t(float x) { return (bf16)(__float_as_uint(x) >> 16); }

// flat copy: thread i moves vector i
__global__ __launch_bounds__(256) void k_flat(const V8* x, V8* y, long long n) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i < n) y[i] = x[i];
}
// flat, grid-stride
__global__ __launch_bounds__(256) void k_flat_gs(const V8* x, V8* y, long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) y[i] = x[i];
}
// row groups of G lanes, NV vectors per lane (18 vectors per row): mode 0 copy, 1 + row sums (two shuffles trees), 2 + mean/rstd stores
template <int G, int NV, int MODE, int ROWS>
__global__ __launch_bounds__(256) void k_rows(const bf16* x, bf16* y, float* mean, float* rstd, long long M) {
  constexpr int C = 144, nvec = 18;
  const int gl = threadIdx.x % G;
  const long long gpb = 256 / G, stride = (long long)gridDim.x * gpb;
  for (long long row = blockIdx.x * gpb + threadIdx.x / G; row < M; row += stride * ROWS) {
    V8 b[ROWS][NV];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (gl + k * G < nvec && row + r * stride < M) b[r][k] = reinterpret_cast<const V8*>(x + (row + r * stride) * C)[gl + k * G];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (row + r * stride >= M) break;
      float mu = 0.f, rs = 1.f;
      if (MODE >= 1) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) if (gl + k * G < nvec) for (int e = 0; e < 8; ++e) s += f(b[r][k].v[e]);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mu = s / C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) if (gl + k * G < nvec) for (int e = 0; e < 8; ++e) { const float d = f(b[r][k].v[e]) - mu; q += d * d; }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        rs = rsqrtf(q / C + 1e-5f);
        if (MODE >= 2 && gl == 0) { mean[row + r * stride] = mu; rstd[row + r * stride] = rs; }
      }
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (gl + k * G < nvec) {
          V8 o;
          for (int e = 0; e < 8; ++e) o.v[e] = MODE ? t((f(b[r][k].v[e]) - mu) * rs) : b[r][k].v[e];
          reinterpret_cast<V8*>(y + (row + r * stride) * C)[gl + k * G] = o;
        }
    }
  }
}

template <int G, int NV, int R, bool AFF, bool RTC>
__global__ __launch_bounds__(256) void k_prod(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bb, bf16* __restrict__ y,
                                              float* __restrict__ mean, float* __restrict__ rstd, long long M, int Crt, float eps) {
  const int C = RTC ? Crt : 144;
  const int nvec = C / 8;
  const int gl = threadIdx.x % G;
  const long long gpb = 256 / G, stride = (long long)gridDim.x * gpb;
  float wr[NV][8], br[NV][8];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (gl + k * G) * 8 + e;
      wr[k][e] = AFF ? (c < C ? w[c] : 0.f) : 1.f;
      br[k][e] = AFF ? (c < C ? bb[c] : 0.f) : 0.f;
    }
  for (long long row = blockIdx.x * gpb + threadIdx.x / G; row < M; row += R * stride) {
    V8 b[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long rr = row + r * stride < M ? row + r * stride : M - 1;
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (gl + k * G < nvec) b[r][k] = reinterpret_cast<const V8*>(x + rr * C)[gl + k * G];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long rr = row + r * stride;
      const bool live = rr < M;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (gl + k * G < nvec) for (int e = 0; e < 8; ++e) s += f(b[r][k].v[e]);
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mu = s / C;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) if (gl + k * G < nvec) for (int e = 0; e < 8; ++e) { const float d = f(b[r][k].v[e]) - mu; q += d * d; }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      const float rs = rsqrtf(q / C + eps);
      if (live && gl == 0 && mean) { mean[rr] = mu; rstd[rr] = rs; }
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (live && gl + k * G < nvec) {
          V8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o.v[e] = t((f(b[r][k].v[e]) - mu) * rs * wr[k][e] + br[k][e]);
          reinterpret_cast<V8*>(y + rr * C)[gl + k * G] = o;
        }
    }
  }
}

int main() {
  const long long M = 28 * 64 * 64, C = 144, n = M * C / 8;
  bf16 *x, *y; float *mean, *rstd;
  hipMalloc(&x, M * C * 2); hipMalloc(&y, M * C * 2); hipMalloc(&mean, M * 4); hipMalloc(&rstd, M * 4);
  hipMemset(x, 0x3f, M * C * 2);
  if (getenv("RANDOM_DATA")) {
    std::vector<bf16> h(M * C);
    unsigned st = 12345;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (bf16)(0x3f00 + ((st >> 16) & 0xff) + ((st >> 31) << 15)); }
    hipMemcpy(x, h.data(), M * C * 2, hipMemcpyHostToDevice);
    printf("random data\n");
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %6.1f us  %5.0f GB/s\n", name, ms / 20 * 1e3, 2.0 * M * C * 2 / (ms / 20 * 1e-3) / 1e9);
  };
  run("flat copy, one vector per thread", [&] { hipLaunchKernelGGL(k_flat, dim3((n + 255) / 256), dim3(256), 0, 0, (const V8*)x, (V8*)y, n); });
  run("flat copy, grid-stride 2048 blocks", [&] { hipLaunchKernelGGL(k_flat_gs, dim3(2048), dim3(256), 0, 0, (const V8*)x, (V8*)y, n); });
#define RUN(G, NV, MODE, ROWS, BL) run("rows G=" #G " NV=" #NV " mode " #MODE " rows " #ROWS " blocks " #BL, [&] { hipLaunchKernelGGL((k_rows<G, NV, MODE, ROWS>), dim3(BL), dim3(256), 0, 0, x, y, mean, rstd, M); })
  RUN(16, 2, 0, 1, 1536); RUN(16, 2, 0, 1, 7168); RUN(16, 2, 0, 2, 1536); RUN(16, 2, 0, 4, 1536);
  RUN(32, 1, 0, 1, 1536); RUN(32, 1, 0, 1, 14336); RUN(32, 1, 0, 2, 1536); RUN(32, 1, 0, 4, 1536);
  RUN(16, 2, 1, 1, 1536); RUN(16, 2, 2, 1, 1536); RUN(16, 2, 2, 1, 7168); RUN(16, 2, 2, 2, 1536);
  RUN(32, 1, 2, 1, 1536); RUN(32, 1, 2, 1, 14336); RUN(32, 1, 2, 2, 1536); RUN(32, 1, 2, 4, 1536); RUN(32, 1, 2, 2, 3584);
  float *w, *bb; hipMalloc(&w, 4096); hipMalloc(&bb, 4096); hipMemset(w, 0, 4096); hipMemset(bb, 0, 4096);
#define RUNP(G, NV, R, AFF, RTC, BL) run("prod G=" #G " NV=" #NV " R=" #R " affine " #AFF " runtimeC " #RTC " blocks " #BL, [&] { hipLaunchKernelGGL((k_prod<G, NV, R, AFF, RTC>), dim3(BL), dim3(256), 0, 0, x, w, bb, y, mean, rstd, M, 144, 1e-5f); })
  RUNP(16, 2, 1, false, false, 1536); RUNP(16, 2, 1, false, true, 1536); RUNP(16, 2, 1, true, false, 1536); RUNP(16, 2, 1, true, true, 1536);
  RUNP(16, 2, 1, true, true, 1024); RUNP(16, 2, 2, true, true, 1024); RUNP(16, 2, 2, true, true, 768); RUNP(16, 2, 2, false, true, 1024);
  return 0;
}
