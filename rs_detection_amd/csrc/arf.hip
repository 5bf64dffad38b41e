// arf.hip -- Active Rotating Filter forward / backward for gfx950.
//
// Replaces: arf_forward / arf_backward
//   /root/reference/python/jdet/ops/orn.py:260-280; CUDA kernels :17-72 (the
//   intended semantics -- the CPU kernels :138-211 index with uint16 and wrap
//   past 65535 weights); ORConv2d.rotate_arf :680-681.
//
// Pure data movement (S2ANet: 73 728 weights -> 589 824): one thread per source
// weight, index table (<= 9*8 bytes for 3x3) staged in LDS.  int64-safe indexing.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

constexpr int ARF_MAX_TABLE = 4096;  // nEntry * nRot bytes kept in LDS

__global__ void arf_forward_kernel(const float* __restrict__ weight,
                                   const uint8_t* __restrict__ indices, long long total, int I,
                                   int nEntry, int nRot, float* __restrict__ out) {
  __shared__ uint8_t s_idx[ARF_MAX_TABLE];
  for (int t = threadIdx.x; t < nEntry * nRot; t += blockDim.x) s_idx[t] = indices[t];
  __syncthreads();
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total;
       n += (long long)gridDim.x * blockDim.x) {
    int l = (int)(n % nEntry);
    long long oc = n / nEntry;
    int c = (int)(oc % I);
    long long o = oc / I;
    float v = weight[n];
    for (int k = 0; k < nRot; ++k) {
      int idx = (int)s_idx[l * nRot + k] - 1;
      out[((o * nRot + k) * I + c) * nEntry + idx] = v;
    }
  }
}

__global__ void arf_backward_kernel(const uint8_t* __restrict__ indices,
                                    const float* __restrict__ grad_out, long long total, int I,
                                    int nEntry, int nRot, float* __restrict__ grad_w) {
  __shared__ uint8_t s_idx[ARF_MAX_TABLE];
  for (int t = threadIdx.x; t < nEntry * nRot; t += blockDim.x) s_idx[t] = indices[t];
  __syncthreads();
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total;
       n += (long long)gridDim.x * blockDim.x) {
    int l = (int)(n % nEntry);
    long long oc = n / nEntry;
    int c = (int)(oc % I);
    long long o = oc / I;
    float acc = 0.f;
    for (int k = 0; k < nRot; ++k) {  // ascending k, as orn.py:62-69
      int idx = (int)s_idx[l * nRot + k] - 1;
      acc = acc + grad_out[((o * nRot + k) * I + c) * nEntry + idx];
    }
    grad_w[n] = acc;
  }
}

// ---- rotation-invariant encoding (ops/orn.py:283-540) -------------------------------------
// One thread per (batch, feature): first arg-max over the nOri orientations, then the cyclic shift that brings
// it to slot 0.  nOri <= 32: the group lives in registers.
__global__ void rie_forward_kernel(const float* __restrict__ feature, long long groups, int nOri,
                                   uint8_t* __restrict__ direction, float* __restrict__ aligned) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const float* src = feature + g * nOri;
  float best = -3.402823466e+38F;  // -FLT_MAX, strict '>' (orn.py:312-316)
  int d = 0;
  for (int l = 0; l < nOri; ++l) {
    const float v = src[l];
    if (v > best) {
      best = v;
      d = l;
    }
  }
  direction[g] = (uint8_t)d;
  float* dst = aligned + g * nOri;
  for (int l = 0; l < nOri; ++l) dst[(l - d + nOri) % nOri] = src[l];
}

__global__ void rie_backward_kernel(const uint8_t* __restrict__ direction, const float* __restrict__ grad_out,
                                    long long groups, int nOri, float* __restrict__ grad_in) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const int d = direction[g];
  for (int l = 0; l < nOri; ++l) grad_in[g * nOri + (l + d) % nOri] = grad_out[g * nOri + l];
}


// ---- RotationInvariantPooling (orn.py:595-617): max over the nOri orientation channels of every feature ---------------
// x (N, F*nOri, H, W) -> y (N, F, H, W), y[n,f,p] = max_k x[n, f*nOri + k, p].  One HBM pass each way instead of torch's
// generic amax over a 5-D view (145 us forward on the head's bf16 canvas; this: the copy time of the input).
// NCHW: a thread owns one (n, f, p) and walks nOri planes (coalesced over p).  NHWC: a thread owns the nOri consecutive
// channels of one (n, p, f).  Backward with torch.amax's rule: the gradient is shared equally by the tied maxima.
template <typename T>
__global__ __launch_bounds__(256) void ori_maxpool_kernel(const T* __restrict__ x, long long total, int F, int nOri,
                                                          int HW, int nhwc, T* __restrict__ y) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;   // index into y, in y's memory order
  if (e >= total) return;
  long long base, step;
  if (nhwc) {   // y (n, p, f): e = (n*HW + p)*F + f;  x (n, p, f*nOri + k)
    base = e * nOri;
    step = 1;
  } else {      // y (n, f, p): x (n, f*nOri + k, p)
    const long long nf = e / HW;
    base = nf * nOri * HW + (e - nf * HW);
    step = HW;
  }
  float m = ld1(x + base);
  for (int k = 1; k < nOri; ++k) {
    const float v = ld1(x + base + k * step);
    m = (v > m || v != v) ? v : m;          // NaN propagates like torch.amax
  }
  st1(y + e, m);
}

template <typename T>
__global__ __launch_bounds__(256) void ori_maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                              long long total, int F, int nOri, int HW, int nhwc,
                                                              T* __restrict__ gx) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  long long base, step;
  if (nhwc) {
    base = e * nOri;
    step = 1;
  } else {
    const long long nf = e / HW;
    base = nf * nOri * HW + (e - nf * HW);
    step = HW;
  }
  float m = ld1(x + base);
  for (int k = 1; k < nOri; ++k) {
    const float v = ld1(x + base + k * step);
    m = (v > m || v != v) ? v : m;
  }
  int cnt = 0;
  for (int k = 0; k < nOri; ++k) cnt += (ld1(x + base + k * step) == m) ? 1 : 0;
  const float g = cnt > 0 ? ld1(gy + e) / (float)cnt : 0.f;
  for (int k = 0; k < nOri; ++k) st1(gx + base + k * step, (ld1(x + base + k * step) == m) ? g : 0.f);
}

// channels-last, nOri == 8, F % 4 == 0 (the S2ANet head: F = 32): a thread owns FOUR features = 32 consecutive channels of
// one pixel -- 64 B (bf16) / 128 B (fp32) of loads in flight per lane instead of 16 / 32 B, and an 8- / 16-byte store
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void ori_maxpool8x4_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                             long long total4, T* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;   // index of a group of 4 outputs
  if (e >= total4) return;
  const T* xp = x + e * 32;
  float v[32];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 a = ld4(xp + 4 * j);
    v[4 * j] = a.x, v[4 * j + 1] = a.y, v[4 * j + 2] = a.z, v[4 * j + 3] = a.w;
  }
  float m[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    m[f] = v[8 * f];
#pragma unroll
    for (int k = 1; k < 8; ++k) m[f] = (v[8 * f + k] > m[f] || v[8 * f + k] != v[8 * f + k]) ? v[8 * f + k] : m[f];
  }
  if (!BWD) {
    st4(out + e * 4, make_float4(m[0], m[1], m[2], m[3]));
    return;
  }
  const float4 g4 = ld4(gy + e * 4);
  const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt += (v[8 * f + k] == m[f]) ? 1 : 0;
    const float gg = cnt > 0 ? g[f] / (float)cnt : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[8 * f + k] = (v[8 * f + k] == m[f]) ? gg : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) st4(out + e * 32 + 4 * j, make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]));
}

template <typename T>
static int ori_maxpool(const T* x, const T* gy, int N, int F, int nOri, int HW, int nhwc, T* out, hipStream_t s) {
  if (N < 0 || F < 0 || HW < 0 || nOri < 1) return RSDET_EINVAL;
  const long long total = (long long)N * F * HW;
  if (total == 0) return RSDET_OK;
  if (!x || !out) return RSDET_EINVAL;
  if (nhwc && nOri == 8 && (F & 3) == 0) {
    const long long t4 = total / 4;
    const dim3 g4((unsigned)((t4 + 255) / 256));
    if (gy)
      hipLaunchKernelGGL((ori_maxpool8x4_kernel<T, true>), g4, dim3(256), 0, s, x, gy, t4, out);
    else
      hipLaunchKernelGGL((ori_maxpool8x4_kernel<T, false>), g4, dim3(256), 0, s, x, gy, t4, out);
    return rsdet_launch_status();
  }
  const dim3 grid((unsigned)((total + 255) / 256));
  if (gy)
    hipLaunchKernelGGL(ori_maxpool_bwd_kernel<T>, grid, dim3(256), 0, s, x, gy, total, F, nOri, HW, nhwc, out);
  else
    hipLaunchKernelGGL(ori_maxpool_kernel<T>, grid, dim3(256), 0, s, x, total, F, nOri, HW, nhwc, out);
  return rsdet_launch_status();
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_rie_forward_f32(const float* feature, int nBatch, int nFeature, int nOri, uint8_t* direction,
                                     float* aligned, void* stream) {
  if (nBatch < 0 || nFeature < 0 || nOri < 1 || nOri > 255) return RSDET_EINVAL;
  const long long groups = (long long)nBatch * nFeature;
  if (groups == 0) return RSDET_OK;
  if (!feature || !direction || !aligned) return RSDET_EINVAL;
  hipLaunchKernelGGL(rie_forward_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     feature, groups, nOri, direction, aligned);
  return rsdet_launch_status();
}

extern "C" int rsdet_rie_backward_f32(const uint8_t* direction, const float* grad_out, int nBatch, int nFeature,
                                      int nOri, float* grad_in, void* stream) {
  if (nBatch < 0 || nFeature < 0 || nOri < 1 || nOri > 255) return RSDET_EINVAL;
  const long long groups = (long long)nBatch * nFeature;
  if (groups == 0) return RSDET_OK;
  if (!direction || !grad_out || !grad_in) return RSDET_EINVAL;
  hipLaunchKernelGGL(rie_backward_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     direction, grad_out, groups, nOri, grad_in);
  return rsdet_launch_status();
}

static int arf_check(int O, int I, int nOri, int kH, int kW, int nRot) {
  if (O < 0 || I < 0 || nOri < 1 || kH < 1 || kW < 1 || nRot < 1) return RSDET_EINVAL;
  if ((long long)nOri * kH * kW > 255) return RSDET_EINVAL;  // uint8 1-based index table
  if ((long long)nOri * kH * kW * nRot > ARF_MAX_TABLE) return RSDET_EINVAL;
  return RSDET_OK;
}

extern "C" int rsdet_arf_forward_f32(const float* weight, const uint8_t* indices, int O, int I,
                                     int nOri, int kH, int kW, int nRot, float* out, void* stream) {
  int rc = arf_check(O, I, nOri, kH, kW, nRot);
  if (rc) return rc;
  const int nEntry = nOri * kH * kW;
  long long total = (long long)O * I * nEntry;
  if (total == 0) return RSDET_OK;
  if (!weight || !indices || !out) return RSDET_EINVAL;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(arf_forward_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, weight,
                     indices, total, I, nEntry, nRot, out);
  return rsdet_launch_status();
}

extern "C" int rsdet_arf_backward_f32(const uint8_t* indices, const float* grad_out, int O, int I,
                                      int nOri, int kH, int kW, int nRot, float* grad_weight,
                                      void* stream) {
  int rc = arf_check(O, I, nOri, kH, kW, nRot);
  if (rc) return rc;
  const int nEntry = nOri * kH * kW;
  long long total = (long long)O * I * nEntry;
  if (total == 0) return RSDET_OK;
  if (!grad_out || !indices || !grad_weight) return RSDET_EINVAL;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(arf_backward_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, indices,
                     grad_out, total, I, nEntry, nRot, grad_weight);
  return rsdet_launch_status();
}

extern "C" int rsdet_ori_maxpool_forward(const void* x, int bf16, int N, int F, int nOri, int HW, int nhwc, void* y,
                                         void* stream) {
  return bf16 ? ori_maxpool<bf16_t>((const bf16_t*)x, nullptr, N, F, nOri, HW, nhwc, (bf16_t*)y, (hipStream_t)stream)
              : ori_maxpool<float>((const float*)x, nullptr, N, F, nOri, HW, nhwc, (float*)y, (hipStream_t)stream);
}

extern "C" int rsdet_ori_maxpool_backward(const void* x, const void* grad_y, int bf16, int N, int F, int nOri, int HW,
                                          int nhwc, void* grad_x, void* stream) {
  if (!grad_y) return RSDET_EINVAL;
  return bf16 ? ori_maxpool<bf16_t>((const bf16_t*)x, (const bf16_t*)grad_y, N, F, nOri, HW, nhwc, (bf16_t*)grad_x,
                                    (hipStream_t)stream)
              : ori_maxpool<float>((const float*)x, (const float*)grad_y, N, F, nOri, HW, nhwc, (float*)grad_x,
                                   (hipStream_t)stream);
}
