#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int SHAPE>
__global__ void k_mfma(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (threadIdx.x - j)); }
  float s = 0;
  long long t0 = 0, t1 = 0;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    t1 = clock64();
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    t1 = clock64();
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
float timeit(F f, int reps = 10) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1000.f;
}

int main() {
  float* out; (void)hipMalloc(&out, 1 << 26);
  long long* cyc; (void)hipMalloc(&cyc, 64);
  const int iters = 4000;
  for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD
    const int threads = 256, grid = 256 * wps;  // 4 waves per block, `wps` blocks per CU
    long long h;
    float us = timeit([&] { hipLaunchKernelGGL((k_mfma<4, 16>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters); });
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double n = 4.0 * iters;
    printf("16x16x32 NACC4 wps %d: %.1f us, %.1f cycles/MFMA/wave (memtime), clock ~ %.2f GHz?, %.0f TFLOP/s\n", wps, us, h / n, h / (us * 1000.0), grid * 4 * n * 16384.0 / us / 1e6);
    us = timeit([&] { hipLaunchKernelGGL((k_mfma<2, 32>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters); });
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    n = 2.0 * iters;
    printf("32x32x16 NACC2 wps %d: %.1f us, %.1f cycles/MFMA/wave (memtime), %.0f TFLOP/s\n", wps, us, h / n, grid * 4 * n * 32768.0 / us / 1e6);
  }
  return 0;
}
