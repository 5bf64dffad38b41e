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
__global__ void s2a_refine_offset_kernel(const float* __restrict__ bbox_pred,
                                         const float* __restrict__ anchors, int B, int H, int W,
                                         float stride_px, int ks, F5 mean, F5 stdv,
                                         float max_ratio, float* __restrict__ refined,
                                         float* __restrict__ offset) {
  const long long HW = (long long)H * W;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * HW) return;
  int b = (int)(idx / HW);
  int hw = (int)(idx - (long long)b * HW);
  int h = hw / W, w = hw - h * W;
  float d[5], a[5], r[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    d[k] = bbox_pred[((long long)b * 5 + k) * HW + hw];
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

extern "C" int rsdet_rotated_box_to_poly_f32(const float* boxes, int n, float* polys, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!boxes || !polys) return RSDET_EINVAL;
  hipLaunchKernelGGL(box_to_poly_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     boxes, n, polys);
  return rsdet_launch_status();
}

extern "C" int rsdet_abi_version(void) { return 1; }
