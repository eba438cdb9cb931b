// Calibration microbenchmarks (diagnostics only): bare MFMA issue rate, LDS-read + MFMA loop, barrier cost.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (threadIdx.x - j)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// per "k-step": NR ds_read_b128 of distinct addresses + NM MFMAs, software-pipelined by one step
template <int NR, int NM>
__global__ __launch_bounds__(256) void k_lds_mfma(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)smem)[i] = 0.001f * i;
  __syncthreads();
  f32x4 acc[NM];
  for (int i = 0; i < NM; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const char* base = smem + (threadIdx.x & 63) * 16;
  bf16x8 f[2][NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) f[0][r] = *(const bf16x8*)(base + r * 1024);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* nb = base + (((it + h + 1) & 7) * NR) * 1024;
#pragma unroll
      for (int r = 0; r < NR; ++r) f[(h + 1) & 1][r] = *(const bf16x8*)(nb + r * 1024);
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[h][i % NR], f[h][(i + 1) % NR], acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_barrier(float* out, int iters) {
  float s = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_barrier();
    s = s * 1.0001f + 1.f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float timeit(F f, int reps = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1000.f;  // us
}

int main() {
  float* out; hipMalloc(&out, 1 << 24);
  const int grids[] = {256, 512};
  for (int g : grids) {
    const int iters = 2000;
    float us = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(g), dim3(256), 0, 0, out, iters); });
    double n = 8.0 * iters;
    printf("bare MFMA 16x16x32  grid %d: %.1f us  -> %.2f ns/MFMA/wave  (%.1f TFLOP/s)\n", g, us, us * 1000 / n, g * 4 * n * 16384.0 / us / 1e6);
    us = timeit([&] { hipLaunchKernelGGL((k_lds_mfma<6, 5>), dim3(g), dim3(256), 65536, 0, out, iters); });
    printf("LDS 6 reads + 5 MFMA grid %d: %.1f us  -> %.1f ns per k-step (5 MFMA = %.1f ns at 2.4 GHz)\n", g, us, us * 1000 / iters, 5 * 16 / 2.4);
    us = timeit([&] { hipLaunchKernelGGL((k_lds_mfma<7, 10>), dim3(g), dim3(256), 65536, 0, out, iters); });
    printf("LDS 7 reads + 10 MFMA grid %d: %.1f us -> %.1f ns per k-step (10 MFMA = %.1f ns at 2.4 GHz)\n", g, us, us * 1000 / iters, 10 * 16 / 2.4);
    us = timeit([&] { hipLaunchKernelGGL((k_lds_mfma<10, 9>), dim3(g), dim3(256), 65536, 0, out, iters); });
    printf("LDS 10 reads + 9 MFMA grid %d: %.1f us -> %.1f ns per k-step\n", g, us, us * 1000 / iters);
    us = timeit([&] { hipLaunchKernelGGL(k_barrier, dim3(g), dim3(256), 0, 0, out, 1000); });
    printf("barrier loop grid %d: %.1f us -> %.1f ns per barrier\n", g, us, us);
    us = timeit([&] { hipLaunchKernelGGL(k_barrier, dim3(g), dim3(256), 0, 0, out, 0); });
    printf("empty kernel grid %d: %.2f us\n", g, us);
  }
  return 0;
}
