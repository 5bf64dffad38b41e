// bn_act.hip -- eval-mode BatchNorm + residual add + ReLU in one pass (forward and backward), NCHW; activations
// fp32 or bf16 (the autocast step: loads widen to fp32, the arithmetic and the parameter sums are fp32, stores round
// to nearest even), parameters always fp32.
//
// Replaces, on the backbone part of the path (a21): the three Jittor ops per Bottleneck tail
//   /root/reference/python/jdet/models/backbones/resnet.py:101-126 (bn -> (+ identity) -> relu) with
//   every BatchNorm in eval mode (norm_eval, :177-184), i.e. a per-channel affine.
// The torch route runs them as 2-3 separate HBM passes forward (MIOpenBatchNormFwdInferSpatialEst,
// add, clamp) and 3 backward (threshold, batch_norm_backward_kernel, add): ~8 ms of a 68 ms step.
// Same arithmetic order as the unfused ops: ((x - mean) * invstd) * weight + bias, then + residual,
// then max(., 0).  HBM-bound: forward reads x (+res), writes y; backward reads dy, y, x, writes dx
// (the residual gradient is the masked dy itself and shares dx's mask: written once as `g`).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

constexpr int BN_NT = 256;

template <bool RELU, bool RES, typename T>
__global__ __launch_bounds__(BN_NT) void bn_act_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ var,
                                                          const float* __restrict__ weight,
                                                          const float* __restrict__ bias, float eps, int C,
                                                          int HW, T* __restrict__ y) {
  const int plane = blockIdx.y;  // n * C + c
  const int c = plane % C;
  const float m = mean[c], is = 1.0f / sqrtf(var[c] + eps);
  const float g = weight ? weight[c] : 1.0f, b = bias ? bias[c] : 0.0f;
  const long long base = (long long)plane * HW;
  const int i = (blockIdx.x * BN_NT + threadIdx.x) * 4;
  if (i >= HW) return;
  if (((HW & 3) == 0)) {  // planes stay 16-byte aligned
    const float4 v = ld4(x + base + i);
    float4 o;
    o.x = ((v.x - m) * is) * g + b;
    o.y = ((v.y - m) * is) * g + b;
    o.z = ((v.z - m) * is) * g + b;
    o.w = ((v.w - m) * is) * g + b;
    if (RES) {
      const float4 r = ld4(res + base + i);
      o.x += r.x;
      o.y += r.y;
      o.z += r.z;
      o.w += r.w;
    }
    if (RELU) {
      o.x = fmaxf(o.x, 0.f);
      o.y = fmaxf(o.y, 0.f);
      o.z = fmaxf(o.z, 0.f);
      o.w = fmaxf(o.w, 0.f);
    }
    st4(y + base + i, o);
  } else {
    for (int k = i; k < min(i + 4, HW); ++k) {
      float o = ((ld1(x + base + k) - m) * is) * g + b;
      if (RES) o += ld1(res + base + k);
      if (RELU) o = fmaxf(o, 0.f);
      st1(y + base + k, o);
    }
  }
}

// grid (S, C): block (s, c) walks slice s of channel c's N*HW elements (float4 granules), writes
// g = dy * [y > 0] (the residual's gradient) and/or dx = g * invstd * weight, and leaves its partial
// sums of g and g * xhat in `partial` for the deterministic second stage.
template <bool RELU, typename T>
__global__ __launch_bounds__(BN_NT) void bn_act_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ weight, float eps,
    int N, int C, int HW, T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ partial) {
  const int c = blockIdx.y, S = gridDim.x, s = blockIdx.x;
  const float m = mean[c], is = 1.0f / sqrtf(var[c] + eps);
  const float scale = is * (weight ? weight[c] : 1.0f);
  const int q_per_plane = (HW + 3) >> 2;  // float4 granules per (n, c) plane
  const long long total = (long long)N * q_per_plane;
  const long long per = (total + S - 1) / S;
  const long long q0 = (long long)s * per, q1 = min(total, q0 + per);
  const bool vec = (HW & 3) == 0;
  float sum_g = 0.f, sum_gx = 0.f;
  for (long long q = q0 + threadIdx.x; q < q1; q += BN_NT) {
    const int n = (int)(q / q_per_plane), i = (int)(q - (long long)n * q_per_plane) * 4;
    const long long base = ((long long)n * C + c) * HW + i;
    if (vec) {
      float4 g = ld4(dy + base);
      if (RELU) {
        const float4 o = ld4(y + base);
        g.x = o.x > 0.f ? g.x : 0.f;
        g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f;
        g.w = o.w > 0.f ? g.w : 0.f;
      }
      if (partial) {
        sum_g += (g.x + g.y) + (g.z + g.w);
        if (x) {  // x == nullptr: only the bias gradient is wanted (bias + ReLU epilogue of a convolution)
          const float4 v = ld4(x + base);
          sum_gx += (g.x * ((v.x - m) * is) + g.y * ((v.y - m) * is)) + (g.z * ((v.z - m) * is) + g.w * ((v.w - m) * is));
        }
      }
      if (dres) st4(dres + base, g);
      if (dx) st4(dx + base, make_float4(g.x * scale, g.y * scale, g.z * scale, g.w * scale));
    } else {
      for (int k = 0; k < 4 && i + k < HW; ++k) {
        float g = ld1(dy + base + k);
        if (RELU) g = ld1(y + base + k) > 0.f ? g : 0.f;
        if (partial) {
          sum_g += g;
          if (x) sum_gx += g * ((ld1(x + base + k) - m) * is);
        }
        if (dres) st1(dres + base + k, g);
        if (dx) st1(dx + base + k, g * scale);
      }
    }
  }
  if (!partial) return;
  // block reduction: wave shuffles, then LDS across the four waves
  for (int off = 32; off > 0; off >>= 1) {
    sum_g += __shfl_down(sum_g, off);
    sum_gx += __shfl_down(sum_gx, off);
  }
  __shared__ float s_g[BN_NT / 64], s_gx[BN_NT / 64];
  if ((threadIdx.x & 63) == 0) {
    s_g[threadIdx.x >> 6] = sum_g;
    s_gx[threadIdx.x >> 6] = sum_gx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < BN_NT / 64; ++w) {
      a += s_g[w];
      b += s_gx[w];
    }
    partial[((long long)c * S + s) * 2 + 0] = a;
    partial[((long long)c * S + s) * 2 + 1] = b;
  }
}

// One wave per channel: lanes read the channel's S partial pairs in a fixed stride and fold by shuffles in a fixed
// pattern (deterministic).  (A single thread walking S partials was fine for the NCHW form's S <= 64; the NHWC form
// produces up to 4 096 slices: 166 us per call, 7 ms of a bf16 step, measured.)
__global__ __launch_bounds__(256) void bn_act_bwd_finish_kernel(const float* __restrict__ partial, int C, int S,
                                                                float* __restrict__ dweight, float* __restrict__ dbias) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
#pragma unroll 8        // (eight loads in flight; the additions keep their order)
  for (int s = lane; s < S; s += 64) {
    const float2 p = *reinterpret_cast<const float2*>(partial + ((long long)c * S + s) * 2);
    a += p.x;
    b += p.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
  }
  if (lane == 0) {
    if (dbias) dbias[c] = a;
    if (dweight) dweight[c] = b;
  }
}

// ---- channels-last (NHWC) forms ------------------------------------------------------------------------------------
// The bf16 trunk runs channels_last (MIOpen's bf16 kernels are NHWC-native: on NCHW tensors it wraps every
// convolution in layout transposes, 6.6 ms of a 28 ms step).  Same arithmetic, different indexing: a thread owns FOUR
// CONSECUTIVE CHANNELS (one 16-byte / 8-byte access), so the per-channel parameters are four values in registers and
// consecutive lanes walk a pixel's channel vector.  C % 4 == 0.
template <typename T>
__device__ __forceinline__ bool stored_pos(float v) {   // is the value positive once stored as T?
  if constexpr (sizeof(T) == 2) return bf2f(f2bf(v)) > 0.f;
  return v > 0.f;
}

template <bool RELU, bool RES, typename T>
__global__ __launch_bounds__(BN_NT) void bn_act_fwd_nhwc_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ var,
                                                               const float* __restrict__ weight,
                                                               const float* __restrict__ bias, float eps, int C,
                                                               long long total_q, T* __restrict__ y,
                                                               unsigned char* __restrict__ mask) {
  const long long q = (long long)blockIdx.x * BN_NT + threadIdx.x;
  if (q >= total_q) return;
  const int c0 = (int)((q * 4) % C);
  const float4 m = *reinterpret_cast<const float4*>(mean + c0), vr = *reinterpret_cast<const float4*>(var + c0);
  const float4 g = weight ? *reinterpret_cast<const float4*>(weight + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v = ld4(x + q * 4);
  float4 o;
  o.x = ((v.x - m.x) * (1.0f / sqrtf(vr.x + eps))) * g.x + b.x;
  o.y = ((v.y - m.y) * (1.0f / sqrtf(vr.y + eps))) * g.y + b.y;
  o.z = ((v.z - m.z) * (1.0f / sqrtf(vr.z + eps))) * g.z + b.z;
  o.w = ((v.w - m.w) * (1.0f / sqrtf(vr.w + eps))) * g.w + b.w;
  if (RES) {
    const float4 r = ld4(res + q * 4);
    o.x += r.x, o.y += r.y, o.z += r.z, o.w += r.w;
  }
  if (RELU) {
    // the backward's ReLU gate as ONE BIT per element (a byte per lane granule): it then reads these instead of y
    o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
    if (mask)   // "the STORED y > 0" (bf16: of the rounded value)
      mask[q] = (unsigned char)((int)stored_pos<T>(o.x) | ((int)stored_pos<T>(o.y) << 1) | ((int)stored_pos<T>(o.z) << 2) |
                                ((int)stored_pos<T>(o.w) << 3));
  }
  st4(y + q * 4, o);
}

// Backward: workgroup s owns rows [s * rows_per, ...) of the (N*H*W, C) matrix.  With G = C / 4 channel groups:
// G <= 256 -> 256 / G row lanes share a group column; G > 256 -> a thread owns groups t, t + 256, ... (J of them).
// Per-thread partial sums of g and g * xhat stay in registers over its rows, row lanes fold in LDS in a fixed order,
// and `partial` gets the same (C, S, 2) layout the NCHW form hands to bn_act_bwd_finish_kernel.
template <bool RELU, typename T, int J>
__global__ __launch_bounds__(BN_NT) void bn_act_bwd_nhwc_kernel(
    const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ weight, float eps, long long rows, int C, int rows_per,
    T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ partial, const unsigned char* __restrict__ mask,
    const T* __restrict__ dy2 = nullptr) {
  // dy2: a second gradient of the same output (a block's output feeds the next block's conv1 AND its identity branch:
  // autograd would add the two in a pass of its own -- read 2, write 1 of the trunk's widest tensors)
  __shared__ float s_red[BN_NT][8];
  const int G = C >> 2, S = gridDim.x, s = blockIdx.x;
  const int RL = J == 1 ? max(BN_NT / G, 1) : 1;
  const int t = threadIdx.x;
  const int rl = J == 1 ? t / G : 0;
  const bool active = J == 1 ? rl < RL : true;
  const long long r0 = (long long)s * rows_per, r1 = min(rows, r0 + rows_per);
  float acc[J][8];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[j][k] = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int grp = J == 1 ? t % G : t + j * BN_NT;
    if (!active || grp >= G) continue;
    const int c0 = grp * 4;
    const float4 m = *reinterpret_cast<const float4*>(mean + c0), vr = *reinterpret_cast<const float4*>(var + c0);
    const float4 w = weight ? *reinterpret_cast<const float4*>(weight + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float is0 = 1.0f / sqrtf(vr.x + eps), is1 = 1.0f / sqrtf(vr.y + eps), is2 = 1.0f / sqrtf(vr.z + eps),
                is3 = 1.0f / sqrtf(vr.w + eps);
#ifndef RSDET_BN4_UNROLL
#define RSDET_BN4_UNROLL 1
#endif
#pragma unroll RSDET_BN4_UNROLL
    for (long long r = r0 + rl; r < r1; r += RL) {
      const long long base = r * C + c0;
      float4 g = ld4(dy + base);
      if (dy2) {
        const float4 h = ld4(dy2 + base);
        g.x += h.x, g.y += h.y, g.z += h.z, g.w += h.w;
      }
      if (RELU) {
        if (mask) {       // one byte instead of the 16 / 8 bytes of y
          const unsigned b = mask[base >> 2];
          g.x = (b & 1u) ? g.x : 0.f, g.y = (b & 2u) ? g.y : 0.f, g.z = (b & 4u) ? g.z : 0.f, g.w = (b & 8u) ? g.w : 0.f;
        } else {
          const float4 o = ld4(y + base);
          g.x = o.x > 0.f ? g.x : 0.f, g.y = o.y > 0.f ? g.y : 0.f, g.z = o.z > 0.f ? g.z : 0.f, g.w = o.w > 0.f ? g.w : 0.f;
        }
      }
      if (partial) {
        acc[j][0] += g.x, acc[j][1] += g.y, acc[j][2] += g.z, acc[j][3] += g.w;
        if (x) {
          const float4 v = ld4(x + base);
          acc[j][4] += g.x * ((v.x - m.x) * is0), acc[j][5] += g.y * ((v.y - m.y) * is1);
          acc[j][6] += g.z * ((v.z - m.z) * is2), acc[j][7] += g.w * ((v.w - m.w) * is3);
        }
      }
      if (dres) st4(dres + base, g);
      if (dx) st4(dx + base, make_float4(g.x * (is0 * w.x), g.y * (is1 * w.y), g.z * (is2 * w.z), g.w * (is3 * w.w)));
    }
  }
  if (!partial) return;
  if (J == 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s_red[t][k] = acc[0][k];
    __syncthreads();
    if (t < G) {
      float tot[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) tot[k] = 0.f;
      for (int l = 0; l < RL; ++l)  // fixed order
#pragma unroll
        for (int k = 0; k < 8; ++k) tot[k] += s_red[l * G + t][k];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        partial[((long long)(t * 4 + k) * S + s) * 2 + 0] = tot[k];
        partial[((long long)(t * 4 + k) * S + s) * 2 + 1] = tot[4 + k];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int grp = t + j * BN_NT;
      if (grp >= G) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        partial[((long long)(grp * 4 + k) * S + s) * 2 + 0] = acc[j][k];
        partial[((long long)(grp * 4 + k) * S + s) * 2 + 1] = acc[j][4 + k];
      }
    }
  }
}

// ---- bf16 channels-last forms with EIGHT channels (16 bytes) per lane -------------------------------------------------------
// The four-channel forms above move 8 bytes per lane per access in bf16: half of what a wave instruction can carry.  With
// C % 8 == 0 and C <= 2 048 (every trunk / head map of the step) a lane owns eight consecutive channels: 16-byte loads and
// stores, and the backward needs no multi-group (J > 1) variant.  Same arithmetic, same partial layout.
template <bool RELU, bool RES>
__global__ __launch_bounds__(BN_NT) void bn_act_fwd_nhwc8_kernel(const bf16_t* __restrict__ x,
                                                                const bf16_t* __restrict__ res,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ var,
                                                                const float* __restrict__ weight,
                                                                const float* __restrict__ bias, float eps, int C,
                                                                long long total_q, bf16_t* __restrict__ y,
                                                                unsigned char* __restrict__ mask) {
  // a lane owns granule q and the granule half the tensor away (same channels: total_q / 2 is a multiple of C / 8 --
  // the host launches this form only then): two independent 16-byte load streams per lane, one set of parameters
  const long long half = total_q >> 1;
  const long long q = (long long)blockIdx.x * BN_NT + threadIdx.x;
  if (q >= half) return;
  const int c0 = (int)((q * 8) % C);
  const F8 m = ldp8(mean + c0), vr = ldp8(var + c0);
  F8 g, b, is;
  if (weight) g = ldp8(weight + c0);
  if (bias) b = ldp8(bias + c0);
#pragma unroll
  for (int k = 0; k < 8; ++k) is.v[k] = 1.0f / sqrtf(vr.v[k] + eps);
  const F8 v0 = ld8(x + q * 8), v1 = ld8(x + (q + half) * 8);
  F8 r0, r1;
  if (RES) r0 = ld8(res + q * 8), r1 = ld8(res + (q + half) * 8);
  F8 o0, o1;
  unsigned b0 = 0u, b1 = 0u;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float t0 = ((v0.v[k] - m.v[k]) * is.v[k]) * (weight ? g.v[k] : 1.0f) + (bias ? b.v[k] : 0.0f);
    float t1 = ((v1.v[k] - m.v[k]) * is.v[k]) * (weight ? g.v[k] : 1.0f) + (bias ? b.v[k] : 0.0f);
    if (RES) t0 += r0.v[k], t1 += r1.v[k];
    if (RELU) {
      t0 = fmaxf(t0, 0.f), t1 = fmaxf(t1, 0.f);
      // the gate the backward needs is "the STORED y > 0": compare the bf16-rounded value (the same conversion the
      // store below makes; a positive below the smallest bf16 rounds to +0)
      b0 |= (unsigned)(bf2f(f2bf(t0)) > 0.f) << k, b1 |= (unsigned)(bf2f(f2bf(t1)) > 0.f) << k;
    }
    o0.v[k] = t0, o1.v[k] = t1;
  }
  st8(y + q * 8, o0);
  st8(y + (q + half) * 8, o1);
  if (RELU && mask) mask[q] = (unsigned char)b0, mask[q + half] = (unsigned char)b1;
}

// CFG >= 0: which optional operands exist is a COMPILE-TIME mask (bit 0 mask, 1 x, 4 dres, 5 dx, 6 partial); CFG = -1: decided from the pointers at run time.  The run-time form branches (uniformly) six times per
// row inside the loop, which keeps the compiler from batching the loads of the four unrolled rows: two or three loads
// in flight per lane, 0.40-0.45 of the HBM roofline.  The configurations the bf16 step uses are instantiated branch-free.
constexpr int BN8_MASK = 1, BN8_X = 2, BN8_DRES = 16, BN8_DX = 32, BN8_PARTIAL = 64;
template <bool RELU, int CFG = -1>
__global__ __launch_bounds__(BN_NT) void bn_act_bwd_nhwc8_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ weight, float eps,
    long long rows, int C, int rows_per, bf16_t* __restrict__ dx, bf16_t* __restrict__ dres,
    float* __restrict__ partial, const unsigned char* __restrict__ mask) {
  // Without x (the backward of csrc/gemm1x1_mfma.hip's fused convolution + BatchNorm, whose pre-BatchNorm value never
  // reached memory) the pass forms the gated gradient and its per-channel sum only; the scale gradient then comes from
  // the weight gradient (rsdet_bn_affine_grads_finish_multi_f32 below) -- NOT from xhat = (y - beta) / gamma of the stored
  // output, whose bf16 rounding 1 / gamma would amplify (round 5 did that).
  __shared__ float s_red[BN_NT][17];   // 17: the fold below reads a column of 16 across rows
  const bool has_mask = CFG >= 0 ? (CFG & BN8_MASK) != 0 : mask != nullptr;
  const bool has_x = CFG >= 0 ? (CFG & BN8_X) != 0 : x != nullptr;
  const bool has_dres = CFG >= 0 ? (CFG & BN8_DRES) != 0 : dres != nullptr;
  const bool has_dx = CFG >= 0 ? (CFG & BN8_DX) != 0 : dx != nullptr;
  const bool has_partial = CFG >= 0 ? (CFG & BN8_PARTIAL) != 0 : partial != nullptr;
  const int G = C >> 3, S = gridDim.x, s = blockIdx.x;
  const int RL = BN_NT / G;            // G is a divisor of 256 (host-checked)
  const int t = threadIdx.x, rl = t / G, c0 = (t % G) * 8;
  const long long r0 = (long long)s * rows_per, r1 = min(rows, r0 + rows_per);
  F8 m, is, sc;
#pragma unroll
  for (int k = 0; k < 8; ++k) m.v[k] = 0.f, is.v[k] = 1.f, sc.v[k] = 1.f;
  if (has_x || has_dx) {               // (the gate-and-sum form takes no statistics: mean / var may be NULL there)
    m = ldp8(mean + c0);
    const F8 vr = ldp8(var + c0);
#pragma unroll
    for (int k = 0; k < 8; ++k) is.v[k] = 1.0f / sqrtf(vr.v[k] + eps);
    if (weight) {
      const F8 w = ldp8(weight + c0);
#pragma unroll
      for (int k = 0; k < 8; ++k) sc.v[k] = is.v[k] * w.v[k];
    } else {
      sc = is;
    }
  }
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#ifndef RSDET_BN8_UNROLL
#define RSDET_BN8_UNROLL 4
#endif
#pragma unroll RSDET_BN8_UNROLL
  for (long long r = r0 + rl; r < r1; r += RL) {
    const long long base = r * C + c0;
    F8 g = ld8(dy + base);
    F8 o;
    if (RELU && !has_mask) o = ld8(y + base);
    if (RELU) {
      if (has_mask) {         // one byte instead of the 16 bytes of y
        const unsigned b = mask[base >> 3];
#pragma unroll
        for (int k = 0; k < 8; ++k) g.v[k] = ((b >> k) & 1u) ? g.v[k] : 0.f;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) g.v[k] = o.v[k] > 0.f ? g.v[k] : 0.f;
      }
    }
    if (has_partial) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += g.v[k];
      if (has_x) {
        const F8 v = ld8(x + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[8 + k] += g.v[k] * ((v.v[k] - m.v[k]) * is.v[k]);
      }
    }
    if (has_dres) st8(dres + base, g);
    if (has_dx) {
      F8 d;
#pragma unroll
      for (int k = 0; k < 8; ++k) d.v[k] = g.v[k] * sc.v[k];
      st8(dx + base, d);
    }
  }
  if (!has_partial) return;
#pragma unroll
  for (int k = 0; k < 16; ++k) s_red[t][k] = acc[k];
  __syncthreads();
  if (t < G) {
    float tot[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) tot[k] = 0.f;
    for (int l = 0; l < RL; ++l)  // fixed order
#pragma unroll
      for (int k = 0; k < 16; ++k) tot[k] += s_red[l * G + t][k];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      partial[((long long)(t * 8 + k) * S + s) * 2 + 0] = tot[k];
      partial[((long long)(t * 8 + k) * S + s) * 2 + 1] = tot[8 + k];
    }
  }
}

// launch the instantiation that matches the operands (branch-free loop) or the run-time form
template <bool RELU>
static void bn_act_bwd_nhwc8_launch(int S, hipStream_t s, const bf16_t* dy, const bf16_t* y, const bf16_t* x,
                                    const float* mean, const float* var, const float* weight, float eps, long long rows,
                                    int C, int per, bf16_t* dx, bf16_t* dres, float* partial, const unsigned char* mask) {
  const int cfg = (mask ? BN8_MASK : 0) | (x ? BN8_X : 0) | (dres ? BN8_DRES : 0) | (dx ? BN8_DX : 0) |
                  (partial ? BN8_PARTIAL : 0);
#define BN8_CASE(CFG_)                                                                                              \
  case CFG_:                                                                                                         \
    hipLaunchKernelGGL((bn_act_bwd_nhwc8_kernel<RELU, CFG_>), dim3(S), dim3(BN_NT), 0, s, dy, y, x, mean, var, weight, \
                       eps, rows, C, per, dx, dres, partial, mask);                                                  \
    return;
  switch (cfg) {
    BN8_CASE(BN8_DRES | BN8_PARTIAL)                                     // gate + sums of the fused conv + bn nodes
    BN8_CASE(BN8_PARTIAL)                                                // ... without ReLU: the sums alone
    BN8_CASE(BN8_DX | BN8_PARTIAL)                                       // bias + ReLU of the head towers
    BN8_CASE(BN8_DX)
    BN8_CASE(BN8_MASK | BN8_X | BN8_DX | BN8_PARTIAL)                    // bn_act with the ReLU bit mask
    BN8_CASE(BN8_MASK | BN8_X | BN8_DRES | BN8_DX | BN8_PARTIAL)
    BN8_CASE(BN8_X | BN8_DX | BN8_PARTIAL)
    BN8_CASE(BN8_X | BN8_DRES | BN8_DX | BN8_PARTIAL)
    default: break;
  }
#undef BN8_CASE
  hipLaunchKernelGGL((bn_act_bwd_nhwc8_kernel<RELU, -1>), dim3(S), dim3(BN_NT), 0, s, dy, y, x, mean, var, weight, eps, rows,
                     C, per, dx, dres, partial, mask);
}

static inline bool bn_nhwc8_ok(int C) {   // eight channels per lane: C / 8 a divisor of 256
  const int G = C / 8;
  return C > 0 && (C & 7) == 0 && G <= BN_NT && BN_NT % G == 0;
}

// ---- stem tail: eval-mode BatchNorm + ReLU + 3x3 / stride 2 / padding 1 max-pool in ONE pass (channels-last, forward) ----
// resnet.py:186-189 of the reference: conv1 -> bn1 -> relu -> maxpool.  With the stem frozen (every shipped config) no
// gradient flows here, so the full-resolution activation (4 x 64 x 512 x 512: 134 MB in bf16) need not exist: a lane owns
// VEC consecutive channels of one pooled pixel, reads its <= 9 inputs (neighbouring windows overlap: those re-reads hit
// L2), applies the affine + ReLU to each and keeps the maximum (max and a monotone map do not commute with a negative
// scale, so the affine is applied per input, exactly as the unfused sequence does).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_relu_maxpool_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ var,
                                                                   const float* __restrict__ weight,
                                                                   const float* __restrict__ bias, float eps, int C, int H,
                                                                   int W, int Ho, int Wo, long long total,
                                                                   T* __restrict__ y) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;   // (n, ho, wo, channel group)
  if (q >= total) return;
  const int G = C / VEC;
  const int g = (int)(q % G);
  long long r = q / G;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const long long n = r / Ho;
  const int c0 = g * VEC;
  float m[VEC], is[VEC], gw[VEC], sh[VEC], best[VEC];
#pragma unroll
  for (int j = 0; j < VEC; j += 4) {   // parameters by 16-byte loads
    const float4 a = *reinterpret_cast<const float4*>(mean + c0 + j), b = *reinterpret_cast<const float4*>(var + c0 + j);
    const float4 w4 = weight ? *reinterpret_cast<const float4*>(weight + c0 + j) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + c0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    m[j] = a.x, m[j + 1] = a.y, m[j + 2] = a.z, m[j + 3] = a.w;
    is[j] = 1.0f / sqrtf(b.x + eps), is[j + 1] = 1.0f / sqrtf(b.y + eps), is[j + 2] = 1.0f / sqrtf(b.z + eps),
    is[j + 3] = 1.0f / sqrtf(b.w + eps);
    gw[j] = w4.x, gw[j + 1] = w4.y, gw[j + 2] = w4.z, gw[j + 3] = w4.w;
    sh[j] = b4.x, sh[j + 1] = b4.y, sh[j + 2] = b4.z, sh[j + 3] = b4.w;
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) best[k] = -INFINITY;
  const T* xn = x + n * (long long)H * W * C + c0;
  // all nine loads first (addresses clamped into the map, validity kept aside), then the arithmetic: the loop with
  // early `continue`s serialised the loads behind the border tests
  typedef typename std::conditional<VEC == 8, uint4, float4>::type Raw;
  Raw raw[9];
  bool ok[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int h = 2 * ho - 1 + t / 3, w = 2 * wo - 1 + t % 3;
    ok[t] = h >= 0 && h < H && w >= 0 && w < W;
    const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
    raw[t] = *reinterpret_cast<const Raw*>(xn + ((long long)hc * W + wc) * C);
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v[VEC];
    if constexpr (VEC == 8) {
      const uint4 r4 = raw[t];
      v[0] = __uint_as_float(r4.x << 16), v[1] = __uint_as_float(r4.x & 0xffff0000u);
      v[2] = __uint_as_float(r4.y << 16), v[3] = __uint_as_float(r4.y & 0xffff0000u);
      v[4] = __uint_as_float(r4.z << 16), v[5] = __uint_as_float(r4.z & 0xffff0000u);
      v[6] = __uint_as_float(r4.w << 16), v[7] = __uint_as_float(r4.w & 0xffff0000u);
    } else {
      v[0] = raw[t].x, v[1] = raw[t].y, v[2] = raw[t].z, v[3] = raw[t].w;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      // bn_act's own operation order for the affine (it does not commute with max: scales may be negative); ReLU and the
      // rounding to the storage type the unfused sequence applies before the pool are monotone, so they follow the max
      const float r = ((v[k] - m[k]) * is[k]) * gw[k] + sh[k];
      best[k] = ok[t] ? fmaxf(best[k], r) : best[k];     // a NaN input loses, as it does after bn_act's fmaxf ReLU
    }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) best[k] = fmaxf(best[k], 0.f);
  T* o = y + q * VEC;
  if (VEC == 8) {
    F8 b;
#pragma unroll
    for (int k = 0; k < VEC; ++k) b.v[k] = best[k];
    st8(reinterpret_cast<bf16_t*>(o), b);
  } else {
    st4(o, make_float4(best[0], best[1], best[2], best[3]));
  }
}

static inline bool bn_vec8() { return true; }   // the eight-channel bf16 kernels wherever C % 8 == 0

// rows per workgroup / number of workgroups of the NHWC backward
static inline void bn_nhwc_split(long long rows, int C, int* rows_per, int* S, int vec = 4) {
  const int G = C / vec;
  const int RL = G < BN_NT ? BN_NT / G : 1;
  long long per = (long long)RL * 8;                       // ~8 rows per thread
  long long s = (rows + per - 1) / per;
  // slices the finish kernel folds.  2 048: bf16 step 18.7 -> 18.3 ms against 4 096 (the finish reads half the
  // partials), 512 starves the main pass (19.1); fp32 step flat (profiles/r03_canvas_ab.txt).
  const long long cap = 2048;
  if (s > cap) {
    s = cap;
    per = (rows + s - 1) / s;
    per = (per + RL - 1) / RL * RL;
    s = (rows + per - 1) / per;
  }
  *rows_per = (int)per;
  *S = (int)(s < 1 ? 1 : s);
}

static inline int bn_slices(int N, int C, int HW) {
  long long granules = (long long)N * ((HW + 3) / 4);
  long long want = (4096 + C - 1) / C;                      // ~4096 workgroups in flight
  long long cap = (granules + BN_NT - 1) / BN_NT;           // at least one granule per thread
  long long s = want < cap ? want : cap;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace rsdet

using namespace rsdet;

extern "C" size_t rsdet_bn_act_backward_ws_size(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return (size_t)C * bn_slices(N, C, HW) * 2 * sizeof(float);
}

template <typename T>
static int bn_act_forward(const T* x, const T* residual, const float* running_mean, const float* running_var,
                          const float* weight, const float* bias, float eps, int N, int C, int HW, int relu, T* y,
                          void* stream) {
  if (N < 0 || C <= 0 || HW < 0) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!x || !running_mean || !running_var || !y) return RSDET_EINVAL;
  if ((long long)N * C > 65535LL * 32768LL) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((HW + BN_NT * 4 - 1) / (BN_NT * 4), N * C);
  if (grid.y > 65535u * 1024u) return RSDET_EINVAL;
#define RSDET_BN_FWD(R, A)                                                                                          \
  hipLaunchKernelGGL((bn_act_fwd_kernel<R, A, T>), grid, dim3(BN_NT), 0, s, x, residual, running_mean, running_var, \
                     weight, bias, eps, C, HW, y)
  if (relu) {
    if (residual) RSDET_BN_FWD(true, true); else RSDET_BN_FWD(true, false);
  } else {
    if (residual) RSDET_BN_FWD(false, true); else RSDET_BN_FWD(false, false);
  }
#undef RSDET_BN_FWD
  return rsdet_launch_status();
}

template <typename T>
static int bn_act_backward(const T* grad_y, const T* y, const T* x, const float* running_mean,
                           const float* running_var, const float* weight, float eps, int N, int C, int HW, int relu,
                           T* grad_x, T* grad_residual, float* grad_weight, float* grad_bias, void* ws, size_t ws_bytes,
                           void* stream) {
  if (N < 0 || C <= 0 || HW < 0) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!grad_y || !running_mean || !running_var || (relu && !y)) return RSDET_EINVAL;
  const bool need_param = grad_weight || grad_bias;
  if (need_param && ((grad_weight && !x) || !ws || ws_bytes < rsdet_bn_act_backward_ws_size(N, C, HW)))
    return RSDET_EINVAL;
  if (!grad_weight) x = nullptr;  // the kernel then skips the x stream
  hipStream_t s = (hipStream_t)stream;
  const int S = bn_slices(N, C, HW);
  float* partial = need_param ? (float*)ws : nullptr;
  if (relu)
    hipLaunchKernelGGL((bn_act_bwd_kernel<true, T>), dim3(S, C), dim3(BN_NT), 0, s, grad_y, y, x, running_mean,
                       running_var, weight, eps, N, C, HW, grad_x, grad_residual, partial);
  else
    hipLaunchKernelGGL((bn_act_bwd_kernel<false, T>), dim3(S, C), dim3(BN_NT), 0, s, grad_y, y, x, running_mean,
                       running_var, weight, eps, N, C, HW, grad_x, grad_residual, partial);
  if (need_param)
    hipLaunchKernelGGL(bn_act_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, C, S, grad_weight,
                       grad_bias);
  return rsdet_launch_status();
}

extern "C" int rsdet_bn_act_forward_f32(const float* x, const float* residual, const float* running_mean,
                                        const float* running_var, const float* weight, const float* bias, float eps,
                                        int N, int C, int HW, int relu, float* y, void* stream) {
  return bn_act_forward<float>(x, residual, running_mean, running_var, weight, bias, eps, N, C, HW, relu, y, stream);
}

extern "C" int rsdet_bn_act_backward_f32(const float* grad_y, const float* y, const float* x, const float* running_mean,
                                         const float* running_var, const float* weight, float eps, int N, int C,
                                         int HW, int relu, float* grad_x, float* grad_residual, float* grad_weight,
                                         float* grad_bias, void* ws, size_t ws_bytes, void* stream) {
  return bn_act_backward<float>(grad_y, y, x, running_mean, running_var, weight, eps, N, C, HW, relu, grad_x,
                                grad_residual, grad_weight, grad_bias, ws, ws_bytes, stream);
}

// bf16 activations (uint16 storage), fp32 parameters / parameter gradients: the autocast step
extern "C" int rsdet_bn_act_forward_bf16(const uint16_t* x, const uint16_t* residual, const float* running_mean,
                                         const float* running_var, const float* weight, const float* bias, float eps,
                                         int N, int C, int HW, int relu, uint16_t* y, void* stream) {
  return bn_act_forward<bf16_t>(x, residual, running_mean, running_var, weight, bias, eps, N, C, HW, relu, y, stream);
}

extern "C" int rsdet_bn_act_backward_bf16(const uint16_t* grad_y, const uint16_t* y, const uint16_t* x,
                                          const float* running_mean, const float* running_var, const float* weight,
                                          float eps, int N, int C, int HW, int relu, uint16_t* grad_x,
                                          uint16_t* grad_residual, float* grad_weight, float* grad_bias, void* ws,
                                          size_t ws_bytes, void* stream) {
  return bn_act_backward<bf16_t>(grad_y, y, x, running_mean, running_var, weight, eps, N, C, HW, relu, grad_x,
                                 grad_residual, grad_weight, grad_bias, ws, ws_bytes, stream);
}

// ---- channels-last entries ----------------------------------------------------------------------------------------------
static inline bool bn_nhwc_ok(int C) {
  const int G = C / 4;
  return C > 0 && (C & 3) == 0 && ((G <= BN_NT && BN_NT % G == 0) || (G > BN_NT && G <= 4 * BN_NT));
}

// ---- column sums of a (rows, C) matrix with a small C (bias gradient of the 5- / 15-channel prediction maps of a
// channels_last head: torch's reduction takes ~20 us per call there).  Thread (c, rl): column c, rows rl, rl + RL, ...
// of the workgroup's slice; the row lanes fold in LDS in a fixed order; a second launch folds the slices.
constexpr int CS_MAX_C = 64, CS_MAX_S = 512;
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long long rows, int C, int cpad,
                                                             int rows_per, float* __restrict__ partial) {
  __shared__ float s_red[256];
  const int t = threadIdx.x, c = t % cpad, rl = t / cpad, RL = 256 / cpad;
  const long long r0 = (long long)blockIdx.x * rows_per, r1 = min(rows, r0 + rows_per);
  float acc = 0.f;
  if (c < C)
    for (long long r = r0 + rl; r < r1; r += RL) acc += ld1(x + r * C + c);
  s_red[t] = acc;
  __syncthreads();
  if (rl == 0 && c < C) {
    float a = 0.f;
    for (int k = 0; k < RL; ++k) a += s_red[k * cpad + c];
    partial[(long long)blockIdx.x * C + c] = a;
  }
}
__global__ __launch_bounds__(64) void colsum_finish_kernel(const float* __restrict__ partial, int C, int S,
                                                           float* __restrict__ out) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
#pragma unroll 8        // (eight loads in flight; the additions keep their order)
  for (int s = 0; s < S; ++s) a += partial[(long long)s * C + c];
  out[c] = a;
}
static int colsum_slices(long long rows) { return (int)std::min<long long>(CS_MAX_S, std::max<long long>(1, rows / 512)); }

extern "C" size_t rsdet_colsum_ws_size(long long rows, int C) {
  return rows > 0 && C > 0 ? (size_t)colsum_slices(rows) * C * sizeof(float) : 0;
}
template <typename T>
static int colsum_launch(const T* x, long long rows, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (rows < 0 || C < 1 || C > CS_MAX_C || !out) return RSDET_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) return hipMemsetAsync(out, 0, C * sizeof(float), st) == hipSuccess ? RSDET_OK : RSDET_ELAUNCH;
  if (!x || !ws || ws_bytes < rsdet_colsum_ws_size(rows, C)) return RSDET_EINVAL;
  int cpad = 1;
  while (cpad < C) cpad <<= 1;
  const int S = colsum_slices(rows);
  const int rows_per = (int)((rows + S - 1) / S);
  hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3(S), dim3(256), 0, st, x, rows, C, cpad, rows_per, (float*)ws);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, C, S, out);
  return rsdet_launch_status();
}
extern "C" int rsdet_colsum_f32(const float* x, long long rows, int C, float* out, void* ws, size_t ws_bytes,
                                void* stream) {
  return colsum_launch<float>(x, rows, C, out, ws, ws_bytes, stream);
}
extern "C" int rsdet_colsum_bf16(const uint16_t* x, long long rows, int C, float* out, void* ws, size_t ws_bytes,
                                 void* stream) {
  return colsum_launch<bf16_t>(x, rows, C, out, ws, ws_bytes, stream);
}

extern "C" int rsdet_bn_act_nhwc_supported(int C) { return bn_nhwc_ok(C) ? 1 : 0; }

extern "C" size_t rsdet_bn_act_backward_nhwc_ws_size(int N, int C, int HW) {
  if (N <= 0 || HW <= 0 || !bn_nhwc_ok(C)) return 0;
  int per, S;
  bn_nhwc_split((long long)N * HW, C, &per, &S);
  return (size_t)C * S * 2 * sizeof(float);
}

// Bytes of the ReLU bit mask of an (N, HW, C) channels-last map: one byte per lane granule (8 channels where the bf16
// eight-channel kernels run -- forward AND backward must take them, so the forward's extra condition, an even pixel
// count, is part of it -- else 4 channels, the low nibble used).  0 = no mask form for this shape (callers keep y).
extern "C" size_t rsdet_bn_act_relu_mask_bytes(int N, int C, int HW, int bf16) {
  if (N <= 0 || HW <= 0 || !bn_nhwc_ok(C)) return 0;
  const long long rows = (long long)N * HW;
  if (bf16 && bn_nhwc8_ok(C) && bn_vec8()) return (rows & 1) ? 0 : (size_t)(rows * (C / 8));
  return (size_t)(rows * (C / 4));
}

template <typename T>
static int bn_act_forward_nhwc(const T* x, const T* residual, const float* running_mean, const float* running_var,
                               const float* weight, const float* bias, float eps, int N, int C, int HW, int relu, T* y,
                               unsigned char* mask, void* stream) {
  if (N < 0 || HW < 0 || !bn_nhwc_ok(C)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!x || !running_mean || !running_var || !y) return RSDET_EINVAL;
  if (mask && (!relu || rsdet_bn_act_relu_mask_bytes(N, C, HW, sizeof(T) == 2) == 0)) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if constexpr (sizeof(T) == 2) {
    if (bn_nhwc8_ok(C) && bn_vec8() && (((long long)N * HW) & 1) == 0) {
      const long long tq = (long long)N * HW * (C / 8);
      const dim3 g8((unsigned)((tq / 2 + BN_NT - 1) / BN_NT));
#define RSDET_BN_FWD8(R, A)                                                                                         \
  hipLaunchKernelGGL((bn_act_fwd_nhwc8_kernel<R, A>), g8, dim3(BN_NT), 0, s, x, residual, running_mean, running_var, \
                     weight, bias, eps, C, tq, y, mask)
      if (relu) {
        if (residual) RSDET_BN_FWD8(true, true); else RSDET_BN_FWD8(true, false);
      } else {
        if (residual) RSDET_BN_FWD8(false, true); else RSDET_BN_FWD8(false, false);
      }
#undef RSDET_BN_FWD8
      return rsdet_launch_status();
    }
  }
  const long long total_q = (long long)N * HW * (C / 4);
  const dim3 grid((unsigned)((total_q + BN_NT - 1) / BN_NT));
#define RSDET_BN_FWD(R, A)                                                                                      \
  hipLaunchKernelGGL((bn_act_fwd_nhwc_kernel<R, A, T>), grid, dim3(BN_NT), 0, s, x, residual, running_mean,     \
                     running_var, weight, bias, eps, C, total_q, y, mask)
  if (relu) {
    if (residual) RSDET_BN_FWD(true, true); else RSDET_BN_FWD(true, false);
  } else {
    if (residual) RSDET_BN_FWD(false, true); else RSDET_BN_FWD(false, false);
  }
#undef RSDET_BN_FWD
  return rsdet_launch_status();
}

template <typename T>
static int bn_act_backward_nhwc(const T* grad_y, const T* y, const unsigned char* mask, const T* x,
                                const float* running_mean, const float* running_var, const float* weight, float eps,
                                int N, int C, int HW, int relu, T* grad_x, T* grad_residual, float* grad_weight,
                                float* grad_bias, void* ws, size_t ws_bytes, void* stream, const T* grad_y2 = nullptr) {
  if (N < 0 || HW < 0 || !bn_nhwc_ok(C)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!grad_y || !running_mean || !running_var || (relu && !y && !mask)) return RSDET_EINVAL;
  if (grad_y2 && sizeof(T) == 2) return RSDET_EINVAL;      // (the second gradient input exists in the fp32 kernels only)
  if (mask && rsdet_bn_act_relu_mask_bytes(N, C, HW, sizeof(T) == 2) == 0) return RSDET_EINVAL;
  const bool need_param = grad_weight || grad_bias;
  if (need_param && ((grad_weight && !x) || !ws || ws_bytes < rsdet_bn_act_backward_nhwc_ws_size(N, C, HW)))
    return RSDET_EINVAL;
  if (!grad_weight) x = nullptr;
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)N * HW;
  int per, S;
  bn_nhwc_split(rows, C, &per, &S);
  float* partial = need_param ? (float*)ws : nullptr;
  if constexpr (sizeof(T) == 2) {
    if (bn_nhwc8_ok(C) && bn_vec8()) {
      bn_nhwc_split(rows, C, &per, &S, 8);   // never more slices than the four-channel split the workspace is sized for
      if (relu)
        bn_act_bwd_nhwc8_launch<true>(S, s, grad_y, y, x, running_mean, running_var, weight, eps, rows, C, per, grad_x,
                                      grad_residual, partial, mask);
      else
        bn_act_bwd_nhwc8_launch<false>(S, s, grad_y, y, x, running_mean, running_var, weight, eps, rows, C, per, grad_x,
                                       grad_residual, partial, mask);
      if (need_param)
        hipLaunchKernelGGL(bn_act_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, C, S, grad_weight,
                           grad_bias);
      return rsdet_launch_status();
    }
  }
  const int G = C / 4;
  const int J = G <= BN_NT ? 1 : (G + BN_NT - 1) / BN_NT;
#define RSDET_BN_BWD(R, JJ)                                                                                        \
  hipLaunchKernelGGL((bn_act_bwd_nhwc_kernel<R, T, JJ>), dim3(S), dim3(BN_NT), 0, s, grad_y, y, x, running_mean,   \
                     running_var, weight, eps, rows, C, per, grad_x, grad_residual, partial, mask, grad_y2)
  if (relu) {
    if (J == 1) RSDET_BN_BWD(true, 1); else if (J == 2) RSDET_BN_BWD(true, 2); else RSDET_BN_BWD(true, 4);
  } else {
    if (J == 1) RSDET_BN_BWD(false, 1); else if (J == 2) RSDET_BN_BWD(false, 2); else RSDET_BN_BWD(false, 4);
  }
#undef RSDET_BN_BWD
  if (need_param)
    hipLaunchKernelGGL(bn_act_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, C, S, grad_weight,
                       grad_bias);
  return rsdet_launch_status();
}

// Gate pass of the fused 1x1 convolution + eval BatchNorm (+ identity) + ReLU nodes (csrc/gemm1x1_mfma.hip forward, whose
// pre-BatchNorm value never reaches memory): grad_z = grad_y [y > 0] (relu != 0; relu == 0: grad_z = grad_y is not
// written, pass NULL) and the per-slice channel sums of grad_z LEFT in ws as (C, S, 2) floats ([0] the sum, [1] zero),
// S = rsdet_bn_gate_sums_nhwc_slices(N, C, HW), for rsdet_bn_affine_grads_finish_multi_f32 to fold.  grad_z is the gradient
// of the BatchNorm's OUTPUT: the BatchNorm's scale gamma / sqrt(var + eps) rides in the weights of the backward-data GEMM
// (ops/weight_prep.py) and in the fold of the weight gradient (rsdet_sum_slabs_rowscale_f32), so no scaled copy exists.
// bf16 channels-last, C / 8 a divisor of 256.  ws NULL: no sums.
extern "C" int rsdet_bn_gate_sums_nhwc_slices(int N, int C, int HW) {
  if (N <= 0 || HW <= 0 || !bn_nhwc_ok(C) || !bn_nhwc8_ok(C)) return 0;
  int per, S;
  bn_nhwc_split((long long)N * HW, C, &per, &S, 8);
  return S;
}
extern "C" int rsdet_bn_gate_sums_nhwc_bf16(const uint16_t* grad_y, const uint16_t* y, int N, int C, int HW, int relu,
                                            uint16_t* grad_z, void* ws, size_t ws_bytes, void* stream) {
  if (N < 0 || HW < 0 || !bn_nhwc_ok(C) || !bn_nhwc8_ok(C)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!grad_y || (relu && (!y || !grad_z)) || (!relu && grad_z) || (!relu && !ws)) return RSDET_EINVAL;
  if (ws && ws_bytes < rsdet_bn_act_backward_nhwc_ws_size(N, C, HW)) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)N * HW;
  int per, S;
  bn_nhwc_split(rows, C, &per, &S, 8);
  const bf16_t* none = nullptr;
  if (relu)
    bn_act_bwd_nhwc8_launch<true>(S, s, grad_y, y, none, nullptr, nullptr, nullptr, 0.f, rows, C, per, (bf16_t*)nullptr,
                                  (bf16_t*)grad_z, (float*)ws, (const unsigned char*)nullptr);
  else
    bn_act_bwd_nhwc8_launch<false>(S, s, grad_y, none, none, nullptr, nullptr, nullptr, 0.f, rows, C, per,
                                   (bf16_t*)nullptr, (bf16_t*)nullptr, (float*)ws, (const unsigned char*)nullptr);
  return rsdet_launch_status();
}

void rsdet_launch_sums_finish(const float* partial, int C, int S, float* dweight, float* dbias, hipStream_t stream) {
  hipLaunchKernelGGL(rsdet::bn_act_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, partial, C, S, dweight, dbias);
}

// The affine-parameter gradients of up to RSDET_BN_FINISH_JOBS eval-mode BatchNorms behind convolutions, one launch.
// Job j: partial[j] = (C, S, 2) per-slice sums whose [0] column is sum_p gz[p, c] (gz = the gated gradient of the
// BatchNorm's output);  rowdot[j][c] = sum_k W[c, k] U[c, k] with U = sum_p gz[p, c] patch[p, k] the UNSCALED weight
// gradient of the convolution in fp32 (the folds form it: rsdet_sum_slabs_rowscale_f32, rsdet_conv3x3_wrw_mfma_rowscale_bf16)
// -- which is sum_p gz[p, c] conv[p, c], the convolution output never having been stored.  Then
//   grad_beta[c]  = sum_s partial[c][s][0]
//   grad_gamma[c] = (rowdot[c] - mean[c] grad_beta[c]) / sqrt(var[c] + eps)          (= sum_p gz xhat, exactly)
// rowdot[j] NULL: grad_gamma[j] is not written.  One wave per channel, fixed order.
constexpr int BN_FINISH_JOBS = 4;
struct BnFinishJobs {
  const float* partial[BN_FINISH_JOBS];
  const float* rowdot[BN_FINISH_JOBS];
  const float* mean[BN_FINISH_JOBS];
  const float* var[BN_FINISH_JOBS];
  float* dweight[BN_FINISH_JOBS];
  float* dbias[BN_FINISH_JOBS];
  float eps[BN_FINISH_JOBS];
  int C[BN_FINISH_JOBS], S[BN_FINISH_JOBS], block0[BN_FINISH_JOBS + 1];
  int n;
};
__global__ __launch_bounds__(256) void bn_affine_finish_multi_kernel(BnFinishJobs jobs) {
  int j = 0;
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.block0[j + 1]) ++j;
  const int c = ((int)blockIdx.x - jobs.block0[j]) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int C = jobs.C[j], S = jobs.S[j];
  if (c >= C) return;
  const float* partial = jobs.partial[j];
  float a = 0.f;
#pragma unroll 8        // (eight loads in flight; the additions keep their order)
  for (int s = lane; s < S; s += 64) a += partial[((long long)c * S + s) * 2];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
  if (lane == 0) {
    if (jobs.dbias[j]) jobs.dbias[j][c] = a;
    if (jobs.dweight[j] && jobs.rowdot[j])
      jobs.dweight[j][c] = (jobs.rowdot[j][c] - jobs.mean[j][c] * a) * (1.0f / sqrtf(jobs.var[j][c] + jobs.eps[j]));
  }
}
extern "C" int rsdet_bn_affine_grads_finish_multi_f32(int n, const float* const* partial, const int* C, const int* S,
                                                      const float* const* rowdot, const float* const* running_mean,
                                                      const float* const* running_var, const float* eps,
                                                      float* const* grad_gamma, float* const* grad_beta, void* stream) {
  if (n < 0 || n > BN_FINISH_JOBS) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!partial || !C || !S || !rowdot || !running_mean || !running_var || !eps || !grad_gamma || !grad_beta)
    return RSDET_EINVAL;
  BnFinishJobs jobs;
  jobs.n = n;
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    if (!partial[j] || C[j] < 1 || S[j] < 1) return RSDET_EINVAL;
    if (grad_gamma[j] && rowdot[j] && (!running_mean[j] || !running_var[j])) return RSDET_EINVAL;
    jobs.partial[j] = partial[j], jobs.rowdot[j] = rowdot[j], jobs.mean[j] = running_mean[j], jobs.var[j] = running_var[j];
    jobs.dweight[j] = grad_gamma[j], jobs.dbias[j] = grad_beta[j], jobs.eps[j] = eps[j];
    jobs.C[j] = C[j], jobs.S[j] = S[j], jobs.block0[j] = blocks;
    blocks += (C[j] + 3) / 4;
  }
  jobs.block0[n] = blocks;
  hipLaunchKernelGGL(bn_affine_finish_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs);
  return rsdet_launch_status();
}

#define RSDET_BN_NHWC_ENTRIES(tag, T)                                                                                 \
  extern "C" int rsdet_bn_act_forward_nhwc_##tag(const T* x, const T* residual, const float* running_mean,            \
                                                 const float* running_var, const float* weight, const float* bias,   \
                                                 float eps, int N, int C, int HW, int relu, T* y, void* stream) {     \
    return bn_act_forward_nhwc(reinterpret_cast<const rsdet_bn_##tag##_t*>(x),                                        \
                               reinterpret_cast<const rsdet_bn_##tag##_t*>(residual), running_mean, running_var,      \
                               weight, bias, eps, N, C, HW, relu, reinterpret_cast<rsdet_bn_##tag##_t*>(y), nullptr,  \
                               stream);                                                                               \
  }                                                                                                                   \
  extern "C" int rsdet_bn_act_forward_nhwc_mask_##tag(const T* x, const T* residual, const float* running_mean,       \
                                                      const float* running_var, const float* weight,                  \
                                                      const float* bias, float eps, int N, int C, int HW, int relu,   \
                                                      T* y, uint8_t* relu_mask, void* stream) {                       \
    return bn_act_forward_nhwc(reinterpret_cast<const rsdet_bn_##tag##_t*>(x),                                        \
                               reinterpret_cast<const rsdet_bn_##tag##_t*>(residual), running_mean, running_var,      \
                               weight, bias, eps, N, C, HW, relu, reinterpret_cast<rsdet_bn_##tag##_t*>(y),           \
                               relu_mask, stream);                                                                    \
  }                                                                                                                   \
  extern "C" int rsdet_bn_act_backward_nhwc_mask_##tag(                                                               \
      const T* grad_y, const uint8_t* relu_mask, const T* x, const float* running_mean, const float* running_var,     \
      const float* weight, float eps, int N, int C, int HW, T* grad_x, T* grad_residual, float* grad_weight,          \
      float* grad_bias, void* ws, size_t ws_bytes, void* stream) {                                                    \
    typedef rsdet_bn_##tag##_t E;                                                                                     \
    if (!relu_mask) return RSDET_EINVAL;                                                                              \
    return bn_act_backward_nhwc<E>(reinterpret_cast<const E*>(grad_y), nullptr, relu_mask,                            \
                                   reinterpret_cast<const E*>(x), running_mean, running_var, weight, eps, N, C, HW,   \
                                   1, reinterpret_cast<E*>(grad_x), reinterpret_cast<E*>(grad_residual),              \
                                   grad_weight, grad_bias, ws, ws_bytes, stream);                                     \
  }                                                                                                                   \
  extern "C" int rsdet_bn_act_backward_nhwc_##tag(                                                                    \
      const T* grad_y, const T* y, const T* x, const float* running_mean, const float* running_var,                   \
      const float* weight, float eps, int N, int C, int HW, int relu, T* grad_x, T* grad_residual,                    \
      float* grad_weight, float* grad_bias, void* ws, size_t ws_bytes, void* stream) {                                \
    typedef rsdet_bn_##tag##_t E;                                                                                     \
    return bn_act_backward_nhwc(reinterpret_cast<const E*>(grad_y), reinterpret_cast<const E*>(y), nullptr,           \
                                reinterpret_cast<const E*>(x), running_mean, running_var, weight, eps, N, C, HW,      \
                                relu, reinterpret_cast<E*>(grad_x), reinterpret_cast<E*>(grad_residual), grad_weight, \
                                grad_bias, ws, ws_bytes, stream);                                                     \
  }
typedef float rsdet_bn_f32_t;
typedef bf16_t rsdet_bn_bf16_t;
RSDET_BN_NHWC_ENTRIES(f32, float)
RSDET_BN_NHWC_ENTRIES(bf16, uint16_t)

// rsdet_bn_act_backward_nhwc_mask_f32 for an output that was used TWICE (a residual block's output: the next block's first
// convolution and its identity branch): grad_y + grad_y2 is formed on the fly instead of by a pass of autograd's own.
extern "C" int rsdet_bn_act_backward_nhwc_mask2_f32(const float* grad_y, const float* grad_y2, const uint8_t* relu_mask,
                                                    const float* x, const float* running_mean, const float* running_var,
                                                    const float* weight, float eps, int N, int C, int HW, float* grad_x,
                                                    float* grad_residual, float* grad_weight, float* grad_bias, void* ws,
                                                    size_t ws_bytes, void* stream) {
  if (!relu_mask || !grad_y2) return RSDET_EINVAL;
  return bn_act_backward_nhwc<float>(grad_y, nullptr, relu_mask, x, running_mean, running_var, weight, eps, N, C, HW, 1,
                                     grad_x, grad_residual, grad_weight, grad_bias, ws, ws_bytes, stream, grad_y2);
}

// y (N, Ho, Wo, C) <- maxpool3x3/s2/p1(relu(bn(x))), x (N, H, W, C) channels-last; Ho = (H + 1) / 2, Wo = (W + 1) / 2.
// bf16: C % 8 == 0; f32: C % 4 == 0.
extern "C" int rsdet_bn_relu_maxpool_nhwc(const void* x, int bf16, const float* running_mean, const float* running_var,
                                          const float* weight, const float* bias, float eps, int N, int C, int H, int W,
                                          void* y, void* stream) {
  if (N < 0 || C <= 0 || H < 0 || W < 0 || (C & (bf16 ? 7 : 3))) return RSDET_EINVAL;
  if (N == 0 || H == 0 || W == 0) return RSDET_OK;
  if (!x || !running_mean || !running_var || !y) return RSDET_EINVAL;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  if (bf16) {
    const long long total = (long long)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL((bn_relu_maxpool_nhwc_kernel<bf16_t, 8>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const bf16_t*)x, running_mean, running_var, weight, bias, eps, C, H, W, Ho, Wo, total, (bf16_t*)y);
  } else {
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL((bn_relu_maxpool_nhwc_kernel<float, 4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const float*)x, running_mean, running_var, weight, bias, eps, C, H, W, Ho, Wo, total, (float*)y);
  }
  return rsdet_launch_status();
}
