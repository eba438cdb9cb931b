// Shared device/host helpers for libvmg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vmg_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// per-thread error text (vmg_last_error)
void vmg_set_error(const char* fmt, ...);

// per-device state (runtime.hip): everything the library remembers between calls hangs off the device's vmg_ctx
constexpr int VMG_MAX_DEVICES = 64;
int vmg_current_device();      // hipGetDevice, clamped to [0, VMG_MAX_DEVICES)
int vmg_cu_count(int device);  // compute units of the device (cached in its context)

// live timing hooks (runtime.hip); kernel classes: 1 = conv3x3 fwd/dgrad bf16 C<=160 -> C<=160, 2 = conv wgrad.
// vmg_prof_before returns false at once unless a profiler has been armed with vmg_prof_begin on the current device's context.
bool vmg_prof_before(int klass, long long pixels, hipStream_t st);
void vmg_prof_after(hipStream_t st);
#define VMG_PROF_CONV3X3 1
#define VMG_PROF_WGRAD 2
#define VMG_PROF_CONVQ8 3  // the fp8 conv3x3 C -> C (conv_fp8.hip)

#define VMG_CHECK(cond, ...)      \
  do {                            \
    if (!(cond)) {                \
      vmg_set_error(__VA_ARGS__); \
      return -1;                  \
    }                             \
  } while (0)

#define VMG_LAUNCH_CHECK()                                            \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      vmg_set_error("kernel launch failed: %s", hipGetErrorString(e_)); \
      return -2;                                                      \
    }                                                                 \
  } while (0)

template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<bf16> {
  static constexpr int ES = 2;       // bytes per element
  static constexpr int CHUNKB = 16;  // bytes per 8-element K chunk
};
template <>
struct ElemTraits<float> {
  static constexpr int ES = 4;
  static constexpr int CHUNKB = 32;
};

__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }

// 4 consecutive elements <-> 4 floats
__device__ __forceinline__ void load4(const bf16* p, float v[4]) {
  bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
__device__ __forceinline__ void load4(const float* p, float v[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void store4(bf16* p, const float v[4]) {
  bf16x4 t = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = t;
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  // d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * __expf(-0.5f * x * x) * 0.39894228040143267794f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
