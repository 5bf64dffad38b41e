// rsdet_coder.h -- DeltaXYWHA box coder arithmetic shared by box_coder.hip (the standalone coder entry points)
// and anchor_target.hip (targets encoded where the assignment is made).
// Reference: /root/reference/python/jdet/models/boxes/box_ops.py:176-289 (norm_angle, bbox2delta_rotated,
// delta2bbox_rotated).  One definition, so both paths produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace rsdet {

struct F5 {
  float v[5];
};

constexpr float kPi = 3.14159265358979323846f;

static inline F5 load5(const float* host, float dflt) {
  F5 f;
  for (int k = 0; k < 5; ++k) f.v[k] = host ? host[k] : dflt;
  return f;
}

// norm_angle(a,'le135') = (a + pi/4) mod pi - pi/4 with Python-style mod (box_ops.py:176-182)
__device__ __forceinline__ float norm_angle_le135(float a) {
  const float lo = -0.78539816339744830962f;
  float x = a - lo;
  float r = fmodf(x, kPi);
  if (r != 0.f && r < 0.f) r += kPi;
  return r + lo;
}

// delta2bbox_rotated (box_ops.py:233-289)
__device__ __forceinline__ void decode_one(const float* roi, const float* d, const F5& mean,
                                           const F5& stdv, float max_ratio, float* o) {
  float dx = d[0] * stdv.v[0] + mean.v[0];
  float dy = d[1] * stdv.v[1] + mean.v[1];
  float dw = d[2] * stdv.v[2] + mean.v[2];
  float dh = d[3] * stdv.v[3] + mean.v[3];
  float da = d[4] * stdv.v[4] + mean.v[4];
  dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
  dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
  float c = cosf(roi[4]), s = sinf(roi[4]);
  o[0] = dx * roi[2] * c - dy * roi[3] * s + roi[0];
  o[1] = dx * roi[2] * s + dy * roi[3] * c + roi[1];
  o[2] = roi[2] * expf(dw);
  o[3] = roi[3] * expf(dh);
  o[4] = norm_angle_le135(kPi * da + roi[4]);
}

// bbox2delta_rotated (box_ops.py:184-230): proposal p, ground truth g -> normalised deltas
__device__ __forceinline__ void encode_one(const float* p, const float* g, const F5& mean, const F5& stdv, float* o) {
  float c = cosf(p[4]), s = sinf(p[4]);
  float cx = g[0] - p[0], cy = g[1] - p[1];
  float d[5];
  d[0] = (c * cx + s * cy) / p[2];
  d[1] = (-s * cx + c * cy) / p[3];
  // jt.safe_log: log of the argument clamped to [1e-30, 1e30]
  d[2] = logf(fminf(fmaxf(g[2] / p[2], 1e-30f), 1e30f));
  d[3] = logf(fminf(fmaxf(g[3] / p[3], 1e-30f), 1e30f));
  d[4] = norm_angle_le135(g[4] - p[4]) / kPi;
#pragma unroll
  for (int k = 0; k < 5; ++k) o[k] = (d[k] - mean.v[k]) / stdv.v[k];
}

}  // namespace rsdet
