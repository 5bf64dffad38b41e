// canvas.hip -- the pyramid canvas of the S2ANet head.
//
// The reference runs the head's thirteen convolutions once PER PYRAMID LEVEL with the same weights
// (/root/reference/python/jdet/models/roi_heads/s2anet_head.py:207-252 forward_single under multi_apply :254-255):
// 65 convolution calls forward, 130 backward, four of every five on maps of 64^2 .. 8^2 pixels that cannot fill
// 256 CUs.  Here the five maps of a batch are laid side by side in ONE (B, C, Hc, Wc) canvas with a zero gap of one
// pixel between neighbours: a 3x3 / padding-1 convolution of the canvas equals the five per-level convolutions at
// every level pixel, as long as the gap pixels of its INPUT are zero (they play the role of each level's zero
// padding).  What that takes, and what lives in this file:
//   * pyramid_copy_*: levels -> canvas (gaps zero-filled, one launch) and canvas -> levels (one launch), any mix of
//     NCHW / channels-last on either side, 2- or 4-byte elements; each is the other's backward.
//   * canvas_bias_act_*: the tower epilogue relu(conv + bias) with the gap pixels forced back to zero, so that the
//     next convolution sees zero padding again.  (Backward: the existing bias + ReLU backward of bn_act.hip gates on
//     y > 0, which the zeroed gaps already fail.)
// HBM-bound copies; the per-pixel source map (int32, Hc*Wc entries, L2-resident) is built once per geometry by the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

constexpr int CV_NT = 256;
constexpr int CV_MAX_LEVELS = 8;

struct CanvasLevels {
  void* ptr[CV_MAX_LEVELS];
  int hw[CV_MAX_LEVELS];   // pixels of level l
};

// pixmap[p]: -1 for a gap pixel, else (level << 27) | pixel index inside the level
__device__ __forceinline__ int cv_level(int m) { return m >> 27; }
__device__ __forceinline__ int cv_pixel(int m) { return m & ((1 << 27) - 1); }

// One thread per canvas element, in the canvas' memory order (coalesced on the canvas side; on the level side too
// when both share a layout).  TO_CANVAS: canvas <- levels (gaps <- 0), else levels <- canvas (gaps skipped).
template <typename E, bool TO_CANVAS>
__global__ __launch_bounds__(CV_NT) void pyramid_copy_kernel(CanvasLevels lv, E* __restrict__ canvas,
                                                             const int* __restrict__ pixmap, int B, int C, int HWc,
                                                             int canvas_nhwc, int level_nhwc) {
  const long long e = (long long)blockIdx.x * CV_NT + threadIdx.x;
  const long long total = (long long)B * C * HWc;
  if (e >= total) return;
  int b, c, p;
  if (canvas_nhwc) {
    c = (int)(e % C);
    const long long r = e / C;
    p = (int)(r % HWc);
    b = (int)(r / HWc);
  } else {
    p = (int)(e % HWc);
    const long long r = e / HWc;
    c = (int)(r % C);
    b = (int)(r / C);
  }
  const int m = pixmap[p];
  if (m < 0) {
    if (TO_CANVAS) canvas[e] = (E)0;
    return;
  }
  const int l = cv_level(m), q = cv_pixel(m), hw = lv.hw[l];
  E* lp = reinterpret_cast<E*>(lv.ptr[l]);
  const long long a = level_nhwc ? ((long long)b * hw + q) * C + c : ((long long)b * C + c) * hw + q;
  if (TO_CANVAS)
    canvas[e] = lp[a];
  else
    lp[a] = canvas[e];
}

// Both sides channels-last and a pixel's channel vector a multiple of 16 bytes: one thread per 16-byte granule.
template <bool TO_CANVAS>
__global__ __launch_bounds__(CV_NT) void pyramid_copy_nhwc16_kernel(CanvasLevels lv, uint4* __restrict__ canvas,
                                                                    const int* __restrict__ pixmap, int B, int G,
                                                                    int HWc) {
  const long long e = (long long)blockIdx.x * CV_NT + threadIdx.x;   // granule index, canvas order (b, p, g)
  const long long total = (long long)B * HWc * G;
  if (e >= total) return;
  const int g = (int)(e % G);
  const long long r = e / G;
  const int p = (int)(r % HWc), b = (int)(r / HWc);
  const int m = pixmap[p];
  if (m < 0) {
    if (TO_CANVAS) canvas[e] = make_uint4(0, 0, 0, 0);
    return;
  }
  const int l = cv_level(m), q = cv_pixel(m);
  uint4* lp = reinterpret_cast<uint4*>(lv.ptr[l]);
  const long long a = ((long long)b * lv.hw[l] + q) * G + g;
  if (TO_CANVAS)
    canvas[e] = lp[a];
  else
    lp[a] = canvas[e];
}

// relu(x + bias) with the gap pixels zeroed.  NCHW: grid (chunks of 4 pixels, planes); NHWC: a thread owns four
// consecutive channels of one pixel.
template <bool RELU, typename T>
__global__ __launch_bounds__(CV_NT) void canvas_bias_act_nchw_kernel(const T* __restrict__ x,
                                                                     const float* __restrict__ bias,
                                                                     const uint8_t* __restrict__ live, int C, int HW,
                                                                     T* __restrict__ y) {
  const int plane = blockIdx.y;
  const float b = bias[plane % C];
  const long long base = (long long)plane * HW;
  const int i = (blockIdx.x * CV_NT + threadIdx.x) * 4;
  if (i >= HW) return;
  if ((HW & 3) == 0) {
    const float4 v = ld4(x + base + i);
    const uchar4 k = *reinterpret_cast<const uchar4*>(live + i);
    float4 o = make_float4(v.x + b, v.y + b, v.z + b, v.w + b);
    if (RELU) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
    o.x = k.x ? o.x : 0.f, o.y = k.y ? o.y : 0.f, o.z = k.z ? o.z : 0.f, o.w = k.w ? o.w : 0.f;
    st4(y + base + i, o);
  } else {
    for (int k = i; k < min(i + 4, HW); ++k) {
      float o = ld1(x + base + k) + b;
      if (RELU) o = fmaxf(o, 0.f);
      st1(y + base + k, live[k] ? o : 0.f);
    }
  }
}

template <bool RELU, typename T>
__global__ __launch_bounds__(CV_NT) void canvas_bias_act_nhwc_kernel(const T* __restrict__ x,
                                                                     const float* __restrict__ bias,
                                                                     const uint8_t* __restrict__ live, int C, int HW,
                                                                     long long total_q, T* __restrict__ y) {
  const long long q = (long long)blockIdx.x * CV_NT + threadIdx.x;
  if (q >= total_q) return;
  const long long e = q * 4;
  const int c0 = (int)(e % C);
  const int p = (int)((e / C) % HW);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live[p]) {
    const float4 b = *reinterpret_cast<const float4*>(bias + c0);
    const float4 v = ld4(x + e);
    o = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    if (RELU) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
  }
  st4(y + e, o);
}

// bf16, C % 8 == 0: eight channels (16 bytes) per lane
template <bool RELU>
__global__ __launch_bounds__(CV_NT) void canvas_bias_act_nhwc8_kernel(const bf16_t* __restrict__ x,
                                                                      const float* __restrict__ bias,
                                                                      const uint8_t* __restrict__ live, int C, int HW,
                                                                      long long total_q, bf16_t* __restrict__ y) {
  const long long q = (long long)blockIdx.x * CV_NT + threadIdx.x;
  if (q >= total_q) return;
  const long long e = q * 8;
  const int c0 = (int)(e % C);
  const int p = (int)((e / C) % HW);
  F8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
  if (live[p]) {
    const F8 b = ldp8(bias + c0), v = ld8(x + e);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = v.v[k] + b.v[k];
      o.v[k] = RELU ? fmaxf(t, 0.f) : t;
    }
  }
  st8(y + e, o);
}

template <typename T>
static int canvas_bias_act(const T* x, const float* bias, const uint8_t* live, int N, int C, int HW, int relu, int nhwc,
                           T* y, hipStream_t s) {
  if (!x || !bias || !live || !y || N <= 0 || C <= 0 || HW <= 0) return RSDET_EINVAL;
  if (nhwc) {
    if (C & 3) return RSDET_EINVAL;
    if constexpr (sizeof(T) == 2) {
      if ((C & 7) == 0) {
        const long long tq = (long long)N * HW * (C / 8);
        const dim3 g8((unsigned)((tq + CV_NT - 1) / CV_NT));
        if (relu)
          hipLaunchKernelGGL((canvas_bias_act_nhwc8_kernel<true>), g8, dim3(CV_NT), 0, s, x, bias, live, C, HW, tq, y);
        else
          hipLaunchKernelGGL((canvas_bias_act_nhwc8_kernel<false>), g8, dim3(CV_NT), 0, s, x, bias, live, C, HW, tq, y);
        return rsdet_launch_status();
      }
    }
    const long long total_q = (long long)N * HW * (C / 4);
    const dim3 grid((unsigned)((total_q + CV_NT - 1) / CV_NT));
    if (relu)
      hipLaunchKernelGGL((canvas_bias_act_nhwc_kernel<true, T>), grid, dim3(CV_NT), 0, s, x, bias, live, C, HW, total_q, y);
    else
      hipLaunchKernelGGL((canvas_bias_act_nhwc_kernel<false, T>), grid, dim3(CV_NT), 0, s, x, bias, live, C, HW, total_q, y);
  } else {
    if ((long long)N * C > 65535) return RSDET_EINVAL;
    const dim3 grid((unsigned)rsdet_ceil_div((HW + 3) / 4, CV_NT), (unsigned)(N * C));
    if (relu)
      hipLaunchKernelGGL((canvas_bias_act_nchw_kernel<true, T>), grid, dim3(CV_NT), 0, s, x, bias, live, C, HW, y);
    else
      hipLaunchKernelGGL((canvas_bias_act_nchw_kernel<false, T>), grid, dim3(CV_NT), 0, s, x, bias, live, C, HW, y);
  }
  return rsdet_launch_status();
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_pyramid_copy(void* const* levels, const int* level_pixels, int n_levels, void* canvas,
                                  const int* pixmap, int B, int C, int canvas_pixels, int elem_bytes, int canvas_nhwc,
                                  int levels_nhwc, int to_canvas, void* stream) {
  if (!levels || !level_pixels || !canvas || !pixmap || n_levels < 1 || n_levels > CV_MAX_LEVELS || B <= 0 || C <= 0 ||
      canvas_pixels <= 0 || (elem_bytes != 2 && elem_bytes != 4))
    return RSDET_EINVAL;
  CanvasLevels lv;
  for (int l = 0; l < CV_MAX_LEVELS; ++l) {
    lv.ptr[l] = l < n_levels ? levels[l] : nullptr;
    lv.hw[l] = l < n_levels ? level_pixels[l] : 0;
    if (l < n_levels && (!levels[l] || level_pixels[l] <= 0 || level_pixels[l] >= (1 << 27))) return RSDET_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  const int HWc = canvas_pixels;
  if (canvas_nhwc && levels_nhwc && ((long long)C * elem_bytes) % 16 == 0) {
    const int G = C * elem_bytes / 16;
    const long long total = (long long)B * HWc * G;
    const dim3 grid((unsigned)((total + CV_NT - 1) / CV_NT));
    if (to_canvas)
      hipLaunchKernelGGL((pyramid_copy_nhwc16_kernel<true>), grid, dim3(CV_NT), 0, s, lv, (uint4*)canvas, pixmap, B, G, HWc);
    else
      hipLaunchKernelGGL((pyramid_copy_nhwc16_kernel<false>), grid, dim3(CV_NT), 0, s, lv, (uint4*)canvas, pixmap, B, G, HWc);
    return rsdet_launch_status();
  }
  const long long total = (long long)B * C * HWc;
  const dim3 grid((unsigned)((total + CV_NT - 1) / CV_NT));
#define RSDET_CV(E, T)                                                                                               \
  hipLaunchKernelGGL((pyramid_copy_kernel<E, T>), grid, dim3(CV_NT), 0, s, lv, (E*)canvas, pixmap, B, C, HWc,       \
                     canvas_nhwc, levels_nhwc)
  if (elem_bytes == 4) {
    if (to_canvas) RSDET_CV(uint32_t, true); else RSDET_CV(uint32_t, false);
  } else {
    if (to_canvas) RSDET_CV(uint16_t, true); else RSDET_CV(uint16_t, false);
  }
#undef RSDET_CV
  return rsdet_launch_status();
}

extern "C" int rsdet_canvas_bias_act_f32(const float* x, const float* bias, const uint8_t* live, int N, int C, int HW,
                                         int relu, int nhwc, float* y, void* stream) {
  return canvas_bias_act<float>(x, bias, live, N, C, HW, relu, nhwc, y, (hipStream_t)stream);
}

extern "C" int rsdet_canvas_bias_act_bf16(const uint16_t* x, const float* bias, const uint8_t* live, int N, int C,
                                          int HW, int relu, int nhwc, uint16_t* y, void* stream) {
  return canvas_bias_act<bf16_t>(x, bias, live, N, C, HW, relu, nhwc, y, (hipStream_t)stream);
}
