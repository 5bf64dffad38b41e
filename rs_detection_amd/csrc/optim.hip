// optim.hip -- the optimizer step of the train loop as TWO launches over all parameters (gfx950 / CDNA4).
//
// Replaces, for the whole model at once, what /root/reference/python/jdet/optims/optimizer.py:24-43 does per step
// (global grad-norm clip `grad_clip=dict(max_norm=35, norm_type=2)`, then jittor.optim.SGD: weight decay, momentum,
// update) and, in the bf16 configurations, the two casts the autocast route pays per parameter and step (fp32 master
// -> bf16 for the convolution, bf16 gradient -> fp32 for the accumulation): the model holds bf16 weights, the
// optimizer keeps the fp32 masters and momenta and writes both copies in the same pass.
//
//   launch 1  mt_sqnorm   one workgroup per 16 Ki-element chunk of a gradient: sum of squares in fp32 -> partial;
//                         the last workgroup to arrive (device-scope atomics both sides) folds the partials in a fixed
//                         order (deterministic) into the squared global norm
//   launch 2  mt_sgd      same chunks: g = grad * min(1, max_norm / (norm + 1e-6)) [+ wd * p];
//                         m = momentum * m + g;  p32 -= lr * m;  p_model = (bf16 | f32) p32
// HBM-bound by construction: every gradient read twice, parameter / momentum read + written once.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

constexpr int MT_NT = 256;
constexpr int MT_CHUNK = 16384;   // elements per workgroup

struct MtTensor {       // one per parameter, in device memory (64 bytes)
  const void* grad;     // bf16 or f32 (flags bit 0: bf16)
  void* param;          // the model's copy: bf16 (flags bit 1) or f32
  float* master;        // fp32 master of a bf16 parameter; nullptr when the model's copy IS the fp32 parameter
  float* mom;           // fp32 momentum buffer
  long long n;
  int flags, pad;
  float* mom2;          // AdamW: fp32 second-moment buffer (exp_avg_sq); unused by SGD
  long long pad2;
};
static_assert(sizeof(MtTensor) == 64, "host packs 64-byte records");

__device__ __forceinline__ float mt_load(const void* p, long long i, bool bf16) {
  return bf16 ? bf2f(reinterpret_cast<const uint16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}

__global__ __launch_bounds__(MT_NT) void mt_sqnorm_kernel(const MtTensor* __restrict__ tensors,
                                                          const int2* __restrict__ chunks, int n_chunks,
                                                          float* __restrict__ partial, unsigned* __restrict__ counter,
                                                          float* __restrict__ sqnorm) {
  __shared__ float s_red[MT_NT / 64];
  __shared__ int s_last;
  const int2 c = chunks[blockIdx.x];
  const MtTensor t = tensors[c.x];
  const long long i0 = (long long)c.y * MT_CHUNK, i1 = min(i0 + MT_CHUNK, t.n);
  const bool gb = (t.flags & 1) != 0;
  float acc = 0.f;
  // four elements (16 B of fp32 / 8 B of bf16) per lane and access where the gradient's base allows it (a DDP bucket
  // view may start at any 4-byte offset): the scalar form moved 4 / 2 bytes per lane and ran at 0.9 TB/s
  const bool vec = ((uintptr_t)t.grad & (gb ? 7 : 15)) == 0;
  long long iv = i0;
  if (vec) {
    const long long nv = (i1 - i0) & ~3LL;
    for (long long i = i0 + 4 * threadIdx.x; i < i0 + nv; i += 4 * MT_NT) {
      const float4 g = gb ? ld4(reinterpret_cast<const bf16_t*>(t.grad) + i) : ld4(reinterpret_cast<const float*>(t.grad) + i);
      acc += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
    }
    iv = i0 + nv;
  }
  for (long long i = iv + threadIdx.x; i < i1; i += MT_NT) {
    const float g = mt_load(t.grad, i, gb);
    acc += g * g;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < MT_NT / 64; ++w) v += s_red[w];
    atomicExch(reinterpret_cast<unsigned*>(partial) + blockIdx.x, __float_as_uint(v));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the partial is performed before the arrival is counted
    s_last = atomicAdd(counter, 1u) == (unsigned)n_chunks - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  // last arriver: fixed-order fold of the partials (returning atomics read where the atomic stores were performed)
  float tot = 0.f;
  for (int b = threadIdx.x; b < n_chunks; b += MT_NT) tot += __uint_as_float(atomicOr(reinterpret_cast<unsigned*>(partial) + b, 0u));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < MT_NT / 64; ++w) v += s_red[w];
    sqnorm[0] = v;
    atomicExch(counter, 0u);   // ready for the next step
  }
}

__global__ __launch_bounds__(MT_NT) void mt_sgd_kernel(const MtTensor* __restrict__ tensors,
                                                       const int2* __restrict__ chunks, const float* __restrict__ sqnorm,
                                                       float max_norm, float lr, float momentum, float weight_decay) {
  const int2 c = chunks[blockIdx.x];
  const MtTensor t = tensors[c.x];
  const long long i0 = (long long)c.y * MT_CHUNK, i1 = min(i0 + MT_CHUNK, t.n);
  const bool gb = (t.flags & 1) != 0, pb = (t.flags & 2) != 0;
  float coef = 1.0f;
  if (max_norm > 0.f) coef = fminf(max_norm / (sqrtf(sqnorm[0]) + 1e-6f), 1.0f);   // torch clip_grad_norm_: clamp(max=1)
  float* p32 = t.master ? t.master : reinterpret_cast<float*>(t.param);
  const bool vec = ((uintptr_t)t.grad & (gb ? 7 : 15)) == 0 && ((uintptr_t)p32 & 15) == 0 && ((uintptr_t)t.mom & 15) == 0 &&
                   (!pb || ((uintptr_t)t.param & 7) == 0);
  long long iv = i0;
  if (vec) {   // 16-byte accesses on every stream (see mt_sqnorm_kernel)
    const long long nv = (i1 - i0) & ~3LL;
    for (long long i = i0 + 4 * threadIdx.x; i < i0 + nv; i += 4 * MT_NT) {
      const float4 p = ld4(p32 + i), mo = ld4(t.mom + i);
      float4 g = gb ? ld4(reinterpret_cast<const bf16_t*>(t.grad) + i) : ld4(reinterpret_cast<const float*>(t.grad) + i);
      g.x *= coef, g.y *= coef, g.z *= coef, g.w *= coef;
      if (weight_decay != 0.f) g.x += weight_decay * p.x, g.y += weight_decay * p.y, g.z += weight_decay * p.z, g.w += weight_decay * p.w;
      const float4 m = make_float4(momentum * mo.x + g.x, momentum * mo.y + g.y, momentum * mo.z + g.z, momentum * mo.w + g.w);
      st4(t.mom + i, m);
      const float4 pn = make_float4(p.x - lr * m.x, p.y - lr * m.y, p.z - lr * m.z, p.w - lr * m.w);
      st4(p32 + i, pn);
      if (pb) st4(reinterpret_cast<bf16_t*>(t.param) + i, pn);
    }
    iv = i0 + nv;
  }
  for (long long i = iv + threadIdx.x; i < i1; i += MT_NT) {
    const float p = p32[i];
    float g = mt_load(t.grad, i, gb) * coef;
    if (weight_decay != 0.f) g += weight_decay * p;
    const float m = momentum * t.mom[i] + g;
    t.mom[i] = m;
    const float pn = p - lr * m;
    p32[i] = pn;
    if (pb) reinterpret_cast<uint16_t*>(t.param)[i] = f2bf(pn);
  }
}

// AdamW (torch.optim.AdamW, amsgrad off), same chunks and records; `mom` = exp_avg, `mom2` = exp_avg_sq:
//   p *= 1 - lr * wd;  m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)          bc1 = 1 - beta1^t, bc2 = 1 - beta2^t (host, double)
__global__ __launch_bounds__(MT_NT) void mt_adamw_kernel(const MtTensor* __restrict__ tensors,
                                                         const int2* __restrict__ chunks, const float* __restrict__ sqnorm,
                                                         float max_norm, float decay, float w1, float beta2, float w2,
                                                         float eps, float step_size, float rsqrt_bc2) {
  const int2 c = chunks[blockIdx.x];
  const MtTensor t = tensors[c.x];
  const long long i0 = (long long)c.y * MT_CHUNK, i1 = min(i0 + MT_CHUNK, t.n);
  const bool gb = (t.flags & 1) != 0, pb = (t.flags & 2) != 0;
  float coef = 1.0f;
  if (max_norm > 0.f) coef = fminf(max_norm / (sqrtf(sqnorm[0]) + 1e-6f), 1.0f);
  float* p32 = t.master ? t.master : reinterpret_cast<float*>(t.param);
  auto upd = [&](float p, float g, float& m, float& v) {
    g *= coef;
    p *= decay;
    m = m + (g - m) * w1;
    v = v * beta2 + (w2 * g) * g;
    return p - step_size * (m / (sqrtf(v) * rsqrt_bc2 + eps));
  };
  const bool vec = ((uintptr_t)t.grad & (gb ? 7 : 15)) == 0 && ((uintptr_t)p32 & 15) == 0 && ((uintptr_t)t.mom & 15) == 0 &&
                   ((uintptr_t)t.mom2 & 15) == 0 && (!pb || ((uintptr_t)t.param & 7) == 0);
  long long iv = i0;
  if (vec) {
    const long long nv = (i1 - i0) & ~3LL;
    for (long long i = i0 + 4 * threadIdx.x; i < i0 + nv; i += 4 * MT_NT) {
      const float4 p = ld4(p32 + i);
      float4 m = ld4(t.mom + i), v = ld4(t.mom2 + i);
      const float4 g = gb ? ld4(reinterpret_cast<const bf16_t*>(t.grad) + i) : ld4(reinterpret_cast<const float*>(t.grad) + i);
      const float4 pn = make_float4(upd(p.x, g.x, m.x, v.x), upd(p.y, g.y, m.y, v.y), upd(p.z, g.z, m.z, v.z),
                                    upd(p.w, g.w, m.w, v.w));
      st4(t.mom + i, m);
      st4(t.mom2 + i, v);
      st4(p32 + i, pn);
      if (pb) st4(reinterpret_cast<bf16_t*>(t.param) + i, pn);
    }
    iv = i0 + nv;
  }
  for (long long i = iv + threadIdx.x; i < i1; i += MT_NT) {
    float m = t.mom[i], v = t.mom2[i];
    const float pn = upd(p32[i], mt_load(t.grad, i, gb), m, v);
    t.mom[i] = m;
    t.mom2[i] = v;
    p32[i] = pn;
    if (pb) reinterpret_cast<uint16_t*>(t.param)[i] = f2bf(pn);
  }
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_mt_chunk_elems(void) { return MT_CHUNK; }

extern "C" size_t rsdet_mt_sgd_state_bytes(int n_chunks) {   // zero on entry, left zero: counter | sqnorm | partials
  return n_chunks > 0 ? 256 + (((size_t)n_chunks * 4 + 255) & ~(size_t)255) : 0;
}

extern "C" int rsdet_mt_sgd_step(const void* tensors, const int* chunks, int n_chunks, float max_norm, float lr,
                                 float momentum, float weight_decay, float* sqnorm_out, void* state, size_t state_bytes,
                                 void* stream) {
  if (n_chunks < 0) return RSDET_EINVAL;
  if (n_chunks == 0) return RSDET_OK;
  if (!tensors || !chunks || !state || ((uintptr_t)state & 15) || state_bytes < rsdet_mt_sgd_state_bytes(n_chunks))
    return RSDET_EINVAL;
  unsigned* counter = (unsigned*)state;
  float* sqnorm = (float*)((char*)state + 128);
  float* partial = (float*)((char*)state + 256);
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.f || sqnorm_out) {
    hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(n_chunks), dim3(MT_NT), 0, s, (const MtTensor*)tensors,
                       (const int2*)chunks, n_chunks, partial, counter, sqnorm_out ? sqnorm_out : sqnorm);
  }
  hipLaunchKernelGGL(mt_sgd_kernel, dim3(n_chunks), dim3(MT_NT), 0, s, (const MtTensor*)tensors, (const int2*)chunks,
                     sqnorm_out ? sqnorm_out : sqnorm, max_norm, lr, momentum, weight_decay);
  return rsdet_launch_status();
}

// AdamW over the same records (`mom2` set) and chunks; `step` = this update's 1-based count (bias corrections).
// (hyper-parameters as doubles: torch forms 1 - beta, 1 - lr * wd and the bias corrections in double before rounding)
extern "C" int rsdet_mt_adamw_step(const void* tensors, const int* chunks, int n_chunks, float max_norm, double lr,
                                   double beta1, double beta2, double eps, double weight_decay, long long step,
                                   float* sqnorm_out, void* state, size_t state_bytes, void* stream) {
  if (n_chunks < 0 || step < 1) return RSDET_EINVAL;
  if (n_chunks == 0) return RSDET_OK;
  if (!tensors || !chunks || !state || ((uintptr_t)state & 15) || state_bytes < rsdet_mt_sgd_state_bytes(n_chunks))
    return RSDET_EINVAL;
  unsigned* counter = (unsigned*)state;
  float* sqnorm = (float*)((char*)state + 128);
  float* partial = (float*)((char*)state + 256);
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.f || sqnorm_out) {
    hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(n_chunks), dim3(MT_NT), 0, s, (const MtTensor*)tensors,
                       (const int2*)chunks, n_chunks, partial, counter, sqnorm_out ? sqnorm_out : sqnorm);
  }
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(mt_adamw_kernel, dim3(n_chunks), dim3(MT_NT), 0, s, (const MtTensor*)tensors, (const int2*)chunks,
                     sqnorm_out ? sqnorm_out : sqnorm, max_norm, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)));
  return rsdet_launch_status();
}
