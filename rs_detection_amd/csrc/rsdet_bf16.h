// rsdet_bf16.h -- bf16 storage helpers shared by the kernels that read or write bf16 activations (the autocast step):
// loads widen to fp32, stores round to nearest even; all arithmetic stays fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsdet {

typedef uint16_t bf16_t;  // storage only

// four consecutive activations <-> float4
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even; NaN stays NaN
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  uint2 r;
  r.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
  r.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = r;
}
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f2bf(v); }

}  // namespace rsdet
