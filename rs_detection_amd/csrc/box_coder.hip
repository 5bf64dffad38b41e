// box_coder.hip -- DeltaXYWHA coder, fused S2ANet refine+offset, box->poly (gfx950).
//
// Replaces (all elementwise, HBM-bound; the reference runs each as dozens of
// tiny Jittor ops inside per-image Python loops):
//   bbox2delta_rotated / delta2bbox_rotated / norm_angle
//       /root/reference/python/jdet/models/boxes/box_ops.py:176-289
//   bbox_decode + AlignConv.get_offset
//       /root/reference/python/jdet/models/roi_heads/s2anet_head.py:631-654, :676-713
//   rotated_box_to_poly        box_ops.py:633-654
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_coder.h"

namespace rsdet {

__device__ __forceinline__ float s2a_ld(const float* p) { return *p; }
__device__ __forceinline__ float s2a_ld(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }

__global__ void delta2bbox_kernel(const float* __restrict__ rois, const float* __restrict__ deltas,
                                  int n, F5 mean, F5 stdv, float max_ratio,
                                  float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r[5], d[5], o[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    r[k] = rois[(long long)i * 5 + k];
    d[k] = deltas[(long long)i * 5 + k];
  }
  decode_one(r, d, mean, stdv, max_ratio, o);
#pragma unroll
  for (int k = 0; k < 5; ++k) out[(long long)i * 5 + k] = o[k];
}

__global__ void bbox2delta_kernel(const float* __restrict__ prop, const float* __restrict__ gt,
                                  int n, F5 mean, F5 stdv, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[5];
  encode_one(prop + (long long)i * 5, gt + (long long)i * 5, mean, stdv, o);
#pragma unroll
  for (int k = 0; k < 5; ++k) out[(long long)i * 5 + k] = o[k];
}

// One thread per (b, h, w).  bbox_pred is NCHW so lanes (consecutive w) read
// each of the 5 delta planes coalesced; the 2*ks*ks offset planes are written
// coalesced the same way.
// the work of one (b, h, w): TP = float or bf16 (the prediction of an autocast step, widened exactly)
template <typename TP>
__device__ __forceinline__ void s2a_refine_offset_one(const TP* __restrict__ bbox_pred, const float* __restrict__ anchors,
                                                      int H, int W, long long idx, float stride_px, int ks, const F5& mean,
                                                      const F5& stdv, float max_ratio, float* __restrict__ refined,
                                                      float* __restrict__ offset) {
  const long long HW = (long long)H * W;
  int b = (int)(idx / HW);
  int hw = (int)(idx - (long long)b * HW);
  int h = hw / W, w = hw - h * W;
  float d[5], a[5], r[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    d[k] = s2a_ld(bbox_pred + ((long long)b * 5 + k) * HW + hw);
    a[k] = anchors[(long long)hw * 5 + k];
  }
  decode_one(a, d, mean, stdv, max_ratio, r);
  if (refined) {
#pragma unroll
    for (int k = 0; k < 5; ++k) refined[idx * 5 + k] = r[k];
  }
  if (!offset) return;
  // AlignConv.get_offset, s2anet_head.py:693-713
  float xc = r[0] / stride_px, yc = r[1] / stride_px;
  float bw = r[2] / stride_px, bh = r[3] / stride_px;
  float c = cosf(r[4]), s = sinf(r[4]);
  float dw = bw / (float)ks, dh = bh / (float)ks;
  int pad = (ks - 1) / 2;
  float* op = offset + (long long)b * 2 * ks * ks * HW + hw;
  for (int ty = 0; ty < ks; ++ty) {
    float yy = (float)(ty - pad);
    for (int tx = 0; tx < ks; ++tx) {
      float xx = (float)(tx - pad);
      float x = dw * xx, y = dh * yy;
      float xr = c * x - s * y;
      float yr = s * x + c * y;
      float off_x = (xr + xc) - ((float)w + xx);
      float off_y = (yr + yc) - ((float)h + yy);
      int t = ty * ks + tx;
      op[(long long)(2 * t) * HW] = off_y;
      op[(long long)(2 * t + 1) * HW] = off_x;
    }
  }
}

__global__ void s2a_refine_offset_kernel(const float* __restrict__ bbox_pred,
                                         const float* __restrict__ anchors, int B, int H, int W,
                                         float stride_px, int ks, F5 mean, F5 stdv,
                                         float max_ratio, float* __restrict__ refined,
                                         float* __restrict__ offset) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * H * W) return;
  s2a_refine_offset_one(bbox_pred, anchors, H, W, idx, stride_px, ks, mean, stdv, max_ratio, refined, offset);
}

// all pyramid levels of a step in ONE launch (five launches of 5 us, the small levels at launch latency, and under bf16
// autocast five widening casts before them): workgroups map to levels through blk_base
struct S2aLevels {
  int n, B, ks, bf16;
  unsigned blk_base[RSDET_S2A_MAX_LEVELS + 1];
  int H[RSDET_S2A_MAX_LEVELS], W[RSDET_S2A_MAX_LEVELS];
  float stride[RSDET_S2A_MAX_LEVELS];
  const void* pred[RSDET_S2A_MAX_LEVELS];
  const float* anchors[RSDET_S2A_MAX_LEVELS];
  float* refined[RSDET_S2A_MAX_LEVELS];
  float* offset[RSDET_S2A_MAX_LEVELS];
  F5 mean, stdv;
  float max_ratio;
};

__global__ __launch_bounds__(256) void s2a_refine_offset_multi_kernel(const S2aLevels lv) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < RSDET_S2A_MAX_LEVELS; ++i)
    if (i < lv.n && blockIdx.x >= lv.blk_base[i]) l = i;
  const long long idx = (long long)(blockIdx.x - lv.blk_base[l]) * 256 + threadIdx.x;
  if (idx >= (long long)lv.B * lv.H[l] * lv.W[l]) return;
  if (lv.bf16)
    s2a_refine_offset_one((const uint16_t*)lv.pred[l], lv.anchors[l], lv.H[l], lv.W[l], idx, lv.stride[l], lv.ks, lv.mean,
                          lv.stdv, lv.max_ratio, lv.refined[l], lv.offset[l]);
  else
    s2a_refine_offset_one((const float*)lv.pred[l], lv.anchors[l], lv.H[l], lv.W[l], idx, lv.stride[l], lv.ks, lv.mean,
                          lv.stdv, lv.max_ratio, lv.refined[l], lv.offset[l]);
}

__global__ void box_to_poly_kernel(const float* __restrict__ boxes, int n, float* __restrict__ polys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = boxes + (long long)i * 5;
  float hw = r[2] / 2, hh = r[3] / 2;
  float c = cosf(r[4]), s = sinf(r[4]);
  const float xs[4] = {-hw, hw, hw, -hw};
  const float ys[4] = {-hh, -hh, hh, hh};
  float* o = polys + (long long)i * 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = c * xs[k] + (-s) * ys[k] + r[0];
    o[2 * k + 1] = s * xs[k] + c * ys[k] + r[1];
  }
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_bbox2delta_rotated_f32(const float* proposals, const float* gt, int n,
                                            const float* means_host, const float* stds_host,
                                            float* deltas, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!proposals || !gt || !deltas) return RSDET_EINVAL;
  hipLaunchKernelGGL(bbox2delta_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     proposals, gt, n, load5(means_host, 0.f), load5(stds_host, 1.f), deltas);
  return rsdet_launch_status();
}

extern "C" int rsdet_delta2bbox_rotated_f32(const float* rois, const float* deltas, int n,
                                            const float* means_host, const float* stds_host,
                                            float max_ratio, float* boxes, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!rois || !deltas || !boxes) return RSDET_EINVAL;
  hipLaunchKernelGGL(delta2bbox_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     rois, deltas, n, load5(means_host, 0.f), load5(stds_host, 1.f), max_ratio,
                     boxes);
  return rsdet_launch_status();
}

extern "C" int rsdet_s2a_refine_and_offset_f32(const float* bbox_pred, const float* anchors, int B,
                                               int H, int W, float stride_px, int ks,
                                               const float* means_host, const float* stds_host,
                                               float max_ratio, float* refined, float* offset,
                                               void* stream) {
  if (B < 0 || H < 0 || W < 0 || ks < 1 || !(stride_px > 0)) return RSDET_EINVAL;
  long long total = (long long)B * H * W;
  if (total == 0) return RSDET_OK;
  if (!bbox_pred || !anchors || (!refined && !offset)) return RSDET_EINVAL;
  hipLaunchKernelGGL(s2a_refine_offset_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, bbox_pred, anchors, B, H, W, stride_px, ks,
                     load5(means_host, 0.f), load5(stds_host, 1.f), max_ratio, refined, offset);
  return rsdet_launch_status();
}

extern "C" int rsdet_s2a_refine_and_offset_multi(const rsdet_s2a_levels* d, void* stream) {
  if (!d || d->n_levels < 1 || d->n_levels > RSDET_S2A_MAX_LEVELS || d->B < 0 || d->ks < 1) return RSDET_EINVAL;
  S2aLevels lv;
  lv.n = d->n_levels, lv.B = d->B, lv.ks = d->ks, lv.bf16 = d->pred_bf16 ? 1 : 0;
  long long blk = 0;
  for (int l = 0; l < lv.n; ++l) {
    if (d->H[l] < 0 || d->W[l] < 0 || !(d->stride[l] > 0)) return RSDET_EINVAL;
    const long long total = (long long)d->B * d->H[l] * d->W[l];
    if (total > 0 && (!d->pred[l] || !d->anchors[l] || (!d->refined[l] && !d->offset[l]))) return RSDET_EINVAL;
    lv.H[l] = d->H[l], lv.W[l] = d->W[l], lv.stride[l] = d->stride[l];
    lv.pred[l] = d->pred[l], lv.anchors[l] = d->anchors[l], lv.refined[l] = d->refined[l], lv.offset[l] = d->offset[l];
    lv.blk_base[l] = (unsigned)blk;
    blk += (total + 255) / 256;
  }
  lv.blk_base[lv.n] = (unsigned)blk;
  if (blk > 0x7fffffffLL) return RSDET_EINVAL;
  if (blk == 0) return RSDET_OK;
  lv.mean = load5(d->means, 0.f), lv.stdv = load5(d->stds, 1.f), lv.max_ratio = d->max_ratio;
  hipLaunchKernelGGL(s2a_refine_offset_multi_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, lv);
  return rsdet_launch_status();
}

extern "C" int rsdet_rotated_box_to_poly_f32(const float* boxes, int n, float* polys, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!boxes || !polys) return RSDET_EINVAL;
  hipLaunchKernelGGL(box_to_poly_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     boxes, n, polys);
  return rsdet_launch_status();
}

extern "C" int rsdet_abi_version(void) { return 1; }
