// Which (lane, byte) of operand B meets a given (lane group, byte) of operand A (row 0) in v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void one(const uint8_t* a, int lb, int jb, float* out, int idx) {
  const int l = threadIdx.x;
  i32x8 av, bv = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) av[i] = ((const int*)(a + l * 32))[i];
  const int word = l == lb ? (0x38 << (8 * (jb & 3))) : 0;
  bv[0] = (jb >> 2) == 0 ? word : 0; bv[1] = (jb >> 2) == 1 ? word : 0; bv[2] = (jb >> 2) == 2 ? word : 0; bv[3] = (jb >> 2) == 3 ? word : 0;
  bv[4] = (jb >> 2) == 4 ? word : 0; bv[5] = (jb >> 2) == 5 ? word : 0; bv[6] = (jb >> 2) == 6 ? word : 0; bv[7] = (jb >> 2) == 7 ? word : 0;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 127, 0, 127);
  out[(size_t)idx * 256 + l * 4 + 0] = c[0]; out[(size_t)idx * 256 + l * 4 + 1] = c[1]; out[(size_t)idx * 256 + l * 4 + 2] = c[2]; out[(size_t)idx * 256 + l * 4 + 3] = c[3];
}
int main() {
  const int probes[][2] = {{0, 0}, {0, 15}, {0, 16}, {0, 31}, {1, 0}, {1, 16}, {2, 16}, {3, 31}, {2, 0}, {3, 0}};
  uint8_t* da; float* d;
  hipMalloc(&da, 2048); hipMalloc(&d, 2048 * 256 * 4);
  static float h[2048 * 256];
  for (auto& p : probes) {
    uint8_t ha[2048];
    memset(ha, 0, sizeof ha);
    ha[(p[0] * 16 + 0) * 32 + p[1]] = 0x38;  // A lane (group, row 0), byte j
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    for (int lb = 0; lb < 64; ++lb) for (int jb = 0; jb < 32; ++jb) hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, lb, jb, d, lb * 32 + jb);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("A lane group %d byte %2d (row 0):", p[0], p[1]);
    int n = 0;
    for (int b = 0; b < 2048; ++b)
      for (int o = 0; o < 256; ++o)
        if (h[(size_t)b * 256 + o] != 0.f) { if (++n <= 4) printf("  B lane %2d byte %2d -> C lane %2d reg %d = %g;", b >> 5, b & 31, o >> 2, o & 3, h[(size_t)b * 256 + o]); }
    printf("  [%d nonzero]\n", n);
  }
  return 0;
}
