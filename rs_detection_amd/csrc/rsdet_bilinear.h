// rsdet_bilinear.h -- the bilinear footprint shared by ROIAlignRotated (v0 / v1) and FeatureRefine.
// ops/roi_align_rotated_v1.py:24-68 / :149-190, ops/roi_align_rotated.py:21-57 / :128-168, ops/fr.py:18-111: the
// three copies in the reference are the same function (`y <= 0` and `y < 0` clamp identically).
#pragma once
#include <hip/hip_runtime.h>

namespace rsdet {

struct Bil {
  float w1, w2, w3, w4;
  int xl, xh, yl, yh;
};

// yl == -1 marks "outside"
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

}  // namespace rsdet
