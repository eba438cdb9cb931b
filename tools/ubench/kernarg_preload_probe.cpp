#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { const float* p[8]; int v[40]; };
__global__ void k_args(unsigned long long* out, const float* src, int n, int m, const Big b) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int idx = n + m;                       // first use of preloadable scalars
  asm volatile("" : "+s"(idx));
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  int w = b.v[3] + b.v[20];              // struct fields: ordinary s_load
  asm volatile("" : "+s"(w));
  unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = t1; out[2] = t2; out[3] = idx + w + (src != nullptr); }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  Big b{}; b.v[3] = 1; b.v[20] = 2;
  for (int it = 0; it < 5; ++it) {
    hipLaunchKernelGGL(k_args, dim3(256), dim3(256), 0, 0, d, (const float*)d, 3, 4, b);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("scalars ready after %.2f us, struct fields after %.2f us more\n", (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01);
  }
  return 0;
}
