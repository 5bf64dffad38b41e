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
// fp32 -> bf16, round to nearest even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; rounds 1-4
// did the same rounding with ~6 integer instructions per value -- a third of the vector instructions of the GEMM
// epilogues, which in-kernel stamps showed to be VALU-bound: profiles/r05_g1_stamps.txt).  Same result bit for bit on
// every non-NaN input; a NaN comes back as the hardware's quiet NaN instead of the payload-preserving one.
typedef __attribute__((ext_vector_type(2))) float rsdet_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 rsdet_bf16x2;
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {      // two values packed as they lie in memory
  const rsdet_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rsdet_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  uint2 r;
  r.x = f2bf2(v.x, v.y);
  r.y = f2bf2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = r;
}
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f2bf(v); }

// eight consecutive bf16 activations (16 bytes) <-> eight floats; eight consecutive fp32 parameters
struct F8 {
  float v[8];
};
__device__ __forceinline__ F8 ld8(const bf16_t* p) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  F8 o;
  o.v[0] = __uint_as_float(r.x << 16), o.v[1] = __uint_as_float(r.x & 0xffff0000u);
  o.v[2] = __uint_as_float(r.y << 16), o.v[3] = __uint_as_float(r.y & 0xffff0000u);
  o.v[4] = __uint_as_float(r.z << 16), o.v[5] = __uint_as_float(r.z & 0xffff0000u);
  o.v[6] = __uint_as_float(r.w << 16), o.v[7] = __uint_as_float(r.w & 0xffff0000u);
  return o;
}
__device__ __forceinline__ void st8(bf16_t* p, const F8& a) {
  uint4 r;
  r.x = f2bf2(a.v[0], a.v[1]);
  r.y = f2bf2(a.v[2], a.v[3]);
  r.z = f2bf2(a.v[4], a.v[5]);
  r.w = f2bf2(a.v[6], a.v[7]);
#ifdef RSDET_BN8_NT
  __builtin_nontemporal_store(r.x, reinterpret_cast<uint32_t*>(p));
  __builtin_nontemporal_store(r.y, reinterpret_cast<uint32_t*>(p) + 1);
  __builtin_nontemporal_store(r.z, reinterpret_cast<uint32_t*>(p) + 2);
  __builtin_nontemporal_store(r.w, reinterpret_cast<uint32_t*>(p) + 3);
#else
  *reinterpret_cast<uint4*>(p) = r;
#endif
}
__device__ __forceinline__ F8 ldp8(const float* p) {   // eight consecutive fp32 parameters
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  F8 o;
  o.v[0] = a.x, o.v[1] = a.y, o.v[2] = a.z, o.v[3] = a.w, o.v[4] = b.x, o.v[5] = b.y, o.v[6] = b.z, o.v[7] = b.w;
  return o;
}

}  // namespace rsdet
