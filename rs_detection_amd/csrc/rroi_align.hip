// rroi_align.hip -- ROIAlignRotated_v1 forward / backward for gfx950.
//
// Replaces: _RotatedROIAlign_v1.execute / .grad
//   /root/reference/python/jdet/ops/roi_align_rotated_v1.py:300-351,
//   kernels ROIAlignRotatedForward :71-147, ROIAlignBackward :193-298,
//   bilinear helpers :24-68, :149-190.
//
// One workgroup per RoI.  The RoI frame (centre, bin size, sin/cos) is derived
// once per workgroup into LDS instead of once per output element; threads then
// walk (channel, bin) pairs bin-fastest, so the 49 bins of one channel read one
// feature plane around the RoI (L2/L1-resident) and the output store is
// contiguous.  Backward uses fp32 atomics like the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"

namespace rsdet {

struct RoiFrame {
  int batch, gh, gw;
  float cw, ch, bin_h, bin_w, start_h, start_w, cs, sn;
};

__device__ __forceinline__ RoiFrame make_frame(const float* roi, float scale, int sample_num,
                                               int PH, int PW) {
  RoiFrame f;
  f.batch = (int)roi[0];
  f.cw = roi[1] * scale - 0.5f;  // :89-90 "do not round"
  f.ch = roi[2] * scale - 0.5f;
  float rw = fmaxf(roi[3] * scale, 1.f);
  float rh = fmaxf(roi[4] * scale, 1.f);
  float theta = roi[5];
  f.bin_h = rh / (float)PH;
  f.bin_w = rw / (float)PW;
  f.gh = sample_num > 0 ? sample_num : (int)ceilf(rh / PH);
  f.gw = sample_num > 0 ? sample_num : (int)ceilf(rw / PW);
  f.start_h = -rh / 2.0f;
  f.start_w = -rw / 2.0f;
  f.cs = cosf(theta);
  f.sn = sinf(theta);
  return f;
}

struct Bil {
  float w1, w2, w3, w4;
  int xl, xh, yl, yh;
};

// :24-68 / :149-190 ; yl == -1 marks "outside"
__device__ __forceinline__ Bil bilinear(int H, int W, float y, float x) {
  Bil r{0.f, 0.f, 0.f, 0.f, -1, -1, -1, -1};
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return r;
  if (y < 0) y = 0;
  if (x < 0) x = 0;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (float)yl;
  } else {
    yh = yl + 1;
  }
  if (xl >= W - 1) {
    xh = xl = W - 1;
    x = (float)xl;
  } else {
    xh = xl + 1;
  }
  float ly = y - yl, lx = x - xl;
  float hy = 1.f - ly, hx = 1.f - lx;
  r.w1 = hy * hx;
  r.w2 = hy * lx;
  r.w3 = ly * hx;
  r.w4 = ly * lx;
  r.xl = xl;
  r.xh = xh;
  r.yl = yl;
  r.yh = yh;
  return r;
}

constexpr int RROI_NT = 256;

__global__ __launch_bounds__(RROI_NT) void rroi_forward_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, int C, int H, int W, int PH,
    int PW, float scale, int sample_num, float* __restrict__ out) {
  __shared__ RoiFrame s_f;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) s_f = make_frame(rois + (long long)n * 6, scale, sample_num, PH, PW);
  __syncthreads();
  const RoiFrame f = s_f;
  const int bins = PH * PW;
  const float count = (float)max(f.gh * f.gw, 1);
  const long long HW = (long long)H * W;
  for (int e = blockIdx.y * RROI_NT + threadIdx.x; e < C * bins; e += gridDim.y * RROI_NT) {
    int c = e / bins, bin = e - c * bins;
    int ph = bin / PW, pw = bin - ph * PW;
    const float* fp = feat + ((long long)f.batch * C + c) * HW;
    float acc = 0.f;
    for (int iy = 0; iy < f.gh; ++iy) {
      float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
      for (int ix = 0; ix < f.gw; ++ix) {
        float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
        float x = xx * f.cs + yy * f.sn + f.cw;  // :133-134
        float y = yy * f.cs - xx * f.sn + f.ch;
        Bil b = bilinear(H, W, y, x);
        float v = 0.f;
        if (b.yl >= 0)
          v = b.w1 * fp[b.yl * W + b.xl] + b.w2 * fp[b.yl * W + b.xh] +
              b.w3 * fp[b.yh * W + b.xl] + b.w4 * fp[b.yh * W + b.xh];
        acc += v;
      }
    }
    out[(long long)n * C * bins + e] = acc / count;
  }
}

__global__ __launch_bounds__(RROI_NT) void rroi_backward_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, int C, int H, int W, int PH,
    int PW, float scale, int sample_num, float* __restrict__ grad_feat) {
  __shared__ RoiFrame s_f;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) s_f = make_frame(rois + (long long)n * 6, scale, sample_num, PH, PW);
  __syncthreads();
  const RoiFrame f = s_f;
  const int bins = PH * PW;
  const float count = (float)(f.gh * f.gw);  // :245
  const long long HW = (long long)H * W;
  for (int e = blockIdx.y * RROI_NT + threadIdx.x; e < C * bins; e += gridDim.y * RROI_NT) {
    int c = e / bins, bin = e - c * bins;
    int ph = bin / PW, pw = bin - ph * PW;
    float* gp = grad_feat + ((long long)f.batch * C + c) * HW;
    float top = grad_out[(long long)n * C * bins + e];
    for (int iy = 0; iy < f.gh; ++iy) {
      float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
      for (int ix = 0; ix < f.gw; ++ix) {
        float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
        float x = xx * f.cs + yy * f.sn + f.cw;
        float y = yy * f.cs - xx * f.sn + f.ch;
        Bil b = bilinear(H, W, y, x);
        if (b.yl >= 0) {
          atomicAdd(gp + b.yl * W + b.xl, top * b.w1 / count);
          atomicAdd(gp + b.yl * W + b.xh, top * b.w2 / count);
          atomicAdd(gp + b.yh * W + b.xl, top * b.w3 / count);
          atomicAdd(gp + b.yh * W + b.xh, top * b.w4 / count);
        }
      }
    }
  }
}

}  // namespace rsdet

using namespace rsdet;

static int rroi_check(int R, int C, int H, int W, int PH, int PW) {
  if (R < 0 || C < 0 || H < 1 || W < 1 || PH < 1 || PW < 1) return RSDET_EINVAL;
  if ((long long)C * PH * PW > 0x7fffffffLL) return RSDET_EINVAL;
  return RSDET_OK;
}

extern "C" int rsdet_rroi_align_v1_forward_f32(const float* feat, const float* rois, int R, int C,
                                               int H, int W, int PH, int PW, float spatial_scale,
                                               int sample_num, float* out, void* stream) {
  int rc = rroi_check(R, C, H, W, PH, PW);
  if (rc) return rc;
  if (R == 0 || C == 0) return RSDET_OK;
  if (!feat || !rois || !out) return RSDET_EINVAL;
  int per_roi = (C * PH * PW + RROI_NT - 1) / RROI_NT;
  int gy = per_roi < 8 ? per_roi : 8;  // >= 8 workgroups per RoI keeps small R busy
  hipLaunchKernelGGL(rroi_forward_kernel, dim3(R, gy), dim3(RROI_NT), 0, (hipStream_t)stream, feat,
                     rois, C, H, W, PH, PW, spatial_scale, sample_num, out);
  return rsdet_launch_status();
}

extern "C" int rsdet_rroi_align_v1_backward_f32(const float* grad_out, const float* rois, int R,
                                                int C, int H, int W, int PH, int PW,
                                                float spatial_scale, int sample_num,
                                                float* grad_feat, void* stream) {
  int rc = rroi_check(R, C, H, W, PH, PW);
  if (rc) return rc;
  if (R == 0 || C == 0) return RSDET_OK;
  if (!grad_out || !rois || !grad_feat) return RSDET_EINVAL;
  int per_roi = (C * PH * PW + RROI_NT - 1) / RROI_NT;
  int gy = per_roi < 8 ? per_roi : 8;
  hipLaunchKernelGGL(rroi_backward_kernel, dim3(R, gy), dim3(RROI_NT), 0, (hipStream_t)stream,
                     grad_out, rois, C, H, W, PH, PW, spatial_scale, sample_num, grad_feat);
  return rsdet_launch_status();
}
