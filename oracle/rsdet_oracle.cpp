// rsdet_oracle.cpp -- CPU ORACLE for the oriented-detection hot path.
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT CODE ***
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library.  The product path (rs_detection_amd + librsdet_hip.so) never
// links, imports or falls back to anything in oracle/.
//
// This file is an independent restatement, in flat scalar C++ (no templates,
// single thread, no FMA contraction: build with -ffp-contract=off), of the
// algorithms JDet embeds as C++/CUDA source strings.  Every function cites the
// reference text (paths relative to /root/reference/python/jdet/) it follows.
// Arithmetic order and the float/double promotion points of the reference are
// kept so that results can be compared bit-for-bit with oracle/_ref (the
// reference's own CPU sources compiled by oracle/build_ref.py).
//
// Parity pinning (see DESIGN.md "Oracle"):
//   * box_iou_rotated v0/v1, nms_rotated 5/6 col, ARF fwd/bwd: pinned against
//     oracle/_ref (reference cpu_src compiled from where it lies) and against
//     tests/golden/*.npz generated from it.
//   * deform im2col/col2im/col2im_coord, ROIAlignRotated_v1 fwd/bwd: the
//     reference has CUDA text only; pinned against fixtures produced by running
//     that CUDA text on the host behind a macro shim (tests/golden/make_golden.py).
//   * assigner / coders / anchors: reference is Jittor tensor code (not
//     importable); restated here, pinned by closed-form and property tests.
//     Jittor argmax tie-break is parity-unpinned (first max index chosen).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct P2 {
  float x, y;
};

inline P2 sub(P2 a, P2 b) { return P2{a.x - b.x, a.y - b.y}; }
inline float dot2(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }
// ops/box_iou_rotated.py:48-51  cross_2d(A,B) = A.x*B.y - B.x*A.y
inline float cross2(P2 a, P2 b) { return a.x * b.y - b.x * a.y; }

struct RBox {
  float cx, cy, w, h, a;
};

// ops/box_iou_rotated.py:53-72 (version 0) and ops/box_iou_rotated_v1.py:64-77
// (version 1: x terms of pts[0]/pts[1] have the opposite sign on sin/cos).
void rect_corners(const RBox& b, int version, P2 out[4]) {
  double theta = b.a;
  float c2 = (float)std::cos(theta) * 0.5f;
  float s2 = (float)std::sin(theta) * 0.5f;
  if (version == 0) {
    out[0].x = b.cx - s2 * b.h - c2 * b.w;
    out[0].y = b.cy + c2 * b.h - s2 * b.w;
    out[1].x = b.cx + s2 * b.h - c2 * b.w;
    out[1].y = b.cy - c2 * b.h - s2 * b.w;
  } else {
    out[0].x = b.cx + s2 * b.h + c2 * b.w;
    out[0].y = b.cy + c2 * b.h - s2 * b.w;
    out[1].x = b.cx - s2 * b.h + c2 * b.w;
    out[1].y = b.cy - c2 * b.h - s2 * b.w;
  }
  out[2].x = 2 * b.cx - out[0].x;
  out[2].y = 2 * b.cy - out[0].y;
  out[3].x = 2 * b.cx - out[1].x;
  out[3].y = 2 * b.cy - out[1].y;
}

// ops/box_iou_rotated.py:74-153: 16 edge/edge solves, then 4+4 containment tests.
int collect_points(const P2 r1[4], const P2 r2[4], P2 pts[24]) {
  P2 e1[4], e2[4];
  for (int i = 0; i < 4; ++i) {
    e1[i] = sub(r1[(i + 1) & 3], r1[i]);
    e2[i] = sub(r2[(i + 1) & 3], r2[i]);
  }
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) {
      float det = cross2(e2[j], e1[i]);
      if (std::fabs((double)det) <= 1e-14) continue;  // parallel edges (:93-96)
      P2 d = sub(r2[j], r1[i]);
      float t1 = cross2(e2[j], d) / det;
      float t2 = cross2(e1[i], d) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        pts[n].x = r1[i].x + e1[i].x * t1;
        pts[n].y = r1[i].y + e1[i].y * t1;
        ++n;
      }
    }
  }
  // corners of rect1 inside rect2 (:110-129), then the reverse (:132-150)
  for (int pass = 0; pass < 2; ++pass) {
    const P2* inner = pass == 0 ? r1 : r2;
    const P2* outer = pass == 0 ? r2 : r1;
    const P2* oe = pass == 0 ? e2 : e1;
    P2 AB = oe[0], DA = oe[3];
    float ABAB = dot2(AB, AB), ADAD = dot2(DA, DA);
    for (int i = 0; i < 4; ++i) {
      P2 AP = sub(inner[i], outer[0]);
      float pab = dot2(AP, AB);
      float pad = -dot2(AP, DA);
      if (pab >= 0 && pad >= 0 && pab <= ABAB && pad <= ADAD) pts[n++] = inner[i];
    }
  }
  return n;
}

// ops/box_iou_rotated.py:155-238 with the CPU sort of :317-325 (std::sort and a
// tolerance comparator, kept verbatim in behaviour: same library sort, same
// predicate) -- shift_to_zero=true as called from :275.
int hull_shifted(const P2 p[24], int n, P2 q[24]) {
  int t = 0;
  for (int i = 1; i < n; ++i)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  P2 start = p[t];
  for (int i = 0; i < n; ++i) q[i] = sub(p[i], start);
  std::swap(q[0], q[t]);
  float dist[24];
  for (int i = 0; i < n; ++i) dist[i] = dot2(q[i], q[i]);
  std::sort(q + 1, q + n, [](const P2& A, const P2& B) -> bool {
    float c = cross2(A, B);
    if (std::fabs((double)c) < 1e-6) return dot2(A, A) < dot2(B, B);
    return c > 0;
  });
  // NB (:199-202 then :208-212): dist[] is filled BEFORE the sort and read after
  // it, so dist[k] belongs to the pre-sort occupant of slot k.  Kept as is.
  int k;
  for (k = 1; k < n; ++k)
    if ((double)dist[k] > 1e-8) break;
  if (k == n) {
    q[0] = p[t];
    return 1;
  }
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < n; ++i) {
    while (m > 1 && cross2(sub(q[i], q[m - 2]), sub(q[m - 1], q[m - 2])) >= 0) --m;
    q[m++] = q[i];
  }
  return m;
}

// ops/box_iou_rotated.py:240-252
float fan_area(const P2 q[24], int m) {
  if (m <= 2) return 0;
  float area = 0;
  for (int i = 1; i < m - 1; ++i)
    area = (float)((double)area + std::fabs((double)cross2(sub(q[i], q[0]), sub(q[i + 1], q[0]))));
  return (float)((double)area / 2.0);
}

// ops/box_iou_rotated.py:281-310 (+ label gate of ops/nms_rotated.py:283-286)
float pair_iou(const float* a, const float* b, int version) {
  double sx = (double)(a[0] + b[0]) / 2.0;
  double sy = (double)(a[1] + b[1]) / 2.0;
  RBox b1{(float)(a[0] - sx), (float)(a[1] - sy), a[2], a[3], a[4]};
  RBox b2{(float)(b[0] - sx), (float)(b[1] - sy), b[2], b[3], b[4]};
  float area1 = b1.w * b1.h, area2 = b2.w * b2.h;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;
  P2 r1[4], r2[4], pts[24], ord[24];
  rect_corners(b1, version, r1);
  rect_corners(b2, version, r2);
  int n = collect_points(r1, r2, pts);
  float inter = 0.f;
  if (n > 2) {
    int m = hull_shifted(pts, n, ord);
    inter = fan_area(ord, m);
  }
  return inter / (area1 + area2 - inter);
}

}  // namespace

extern "C" {

// ---- a1/a2: pairwise rotated IoU -------------------------------------------
// ops/box_iou_rotated.py:487-500 (double loop, row-major (n1,n2) output).
// `stride` = floats per box row (5, or 6 when a score/label column is present).
void oracle_box_iou_rotated(const float* b1, int n1, const float* b2, int n2, int stride,
                            int version, float* out) {
  for (int i = 0; i < n1; ++i)
    for (int j = 0; j < n2; ++j)
      out[(size_t)i * n2 + j] = pair_iou(b1 + (size_t)i * stride, b2 + (size_t)j * stride, version);
}

// ---- a16: greedy rotated NMS -----------------------------------------------
// ops/nms_rotated.py:414-449.  dets (n, box_len) with box_len 5 or 6 (6th =
// label: pairs with different labels have IoU 0, :285-286); `order` = indices
// by descending score; suppression test is `>=` (CPU path, :444).
void oracle_nms_rotated(const float* dets, int n, int box_len, const int* order, float thr,
                        uint8_t* keep) {
  std::vector<uint8_t> dead(n, 0);
  std::memset(keep, 0, n);
  for (int a = 0; a < n; ++a) {
    int i = order[a];
    if (dead[i]) continue;
    keep[i] = 1;
    for (int b = a + 1; b < n; ++b) {
      int j = order[b];
      if (dead[j]) continue;
      const float* bi = dets + (size_t)i * box_len;
      const float* bj = dets + (size_t)j * box_len;
      float ovr = (box_len == 6 && bi[5] != bj[5]) ? 0.0f : pair_iou(bi, bj, 0);
      if (ovr >= thr) dead[j] = 1;
    }
  }
}

// ---- a12: Active Rotating Filter -------------------------------------------
// ops/orn.py:17-43 (CUDA, the intended semantics) / :138-172 (CPU; its uint16
// flat index overflows past 65535 entries -- SURVEY q3 -- so this oracle follows
// the CUDA index arithmetic, which is what oracle/_ref also reproduces for
// O*I*nEntry <= 65535).
// weight (O, I, nOri, kH, kW); indices (nOri*kH*kW, nRot) uint8, 1-based;
// out (O*nRot, I*nOri, kH, kW).
void oracle_arf_forward(const float* weight, const uint8_t* indices, int O, int I, int nOri,
                        int kH, int kW, int nRot, float* out) {
  const int nEntry = nOri * kH * kW;
  for (int o = 0; o < O; ++o)
    for (int c = 0; c < I; ++c)
      for (int l = 0; l < nEntry; ++l) {
        float v = weight[((size_t)o * I + c) * nEntry + l];
        for (int k = 0; k < nRot; ++k) {
          int idx = (int)indices[l * nRot + k] - 1;
          out[(((size_t)o * nRot + k) * I + c) * nEntry + idx] = v;
        }
      }
}

// ops/orn.py:45-72 / :173-211: gradWeight[o,c,l] = sum_k gradOut[o,k,c,idx(l,k)]
void oracle_arf_backward(const uint8_t* indices, const float* grad_out, int O, int I, int nOri,
                         int kH, int kW, int nRot, float* grad_w) {
  const int nEntry = nOri * kH * kW;
  for (int o = 0; o < O; ++o)
    for (int c = 0; c < I; ++c)
      for (int l = 0; l < nEntry; ++l) {
        float acc = 0;
        for (int k = 0; k < nRot; ++k) {
          int idx = (int)indices[l * nRot + k] - 1;
          acc = acc + grad_out[(((size_t)o * nRot + k) * I + c) * nEntry + idx];
        }
        grad_w[((size_t)o * I + c) * nEntry + l] = acc;
      }
}

// ---- 8(f)4: rotation-invariant encoding (ops/orn.py:290-363) -----------------
// feature (nBatch, nFeature*nOri) [H = W = 1]; per (batch, feature): direction = first arg-max over the nOri
// orientations (strict '>' from -FLT_MAX), aligned[(l - direction + nOri) % nOri] = feature[l].
void oracle_rie_forward(const float* feature, int nBatch, int nFeature, int nOri, uint8_t* direction, float* aligned) {
  for (int i = 0; i < nBatch; ++i)
    for (int j = 0; j < nFeature; ++j) {
      const float* src = feature + ((size_t)i * nFeature + j) * nOri;
      float best = -3.402823466e+38F;
      uint8_t d = 0;  // the reference leaves the uint8 untouched if nothing beats -FLT_MAX; its output buffer is zeroed
      for (int l = 0; l < nOri; ++l)
        if (src[l] > best) {
          best = src[l];
          d = (uint8_t)l;
        }
      direction[(size_t)i * nFeature + j] = d;
      for (int l = 0; l < nOri; ++l) aligned[((size_t)i * nFeature + j) * nOri + (l - d + nOri) % nOri] = src[l];
    }
}

// ops/orn.py:336-363: gradInput[(l + direction) % nOri] = gradOutput[l]
void oracle_rie_backward(const uint8_t* direction, const float* grad_out, int nBatch, int nFeature, int nOri,
                         float* grad_in) {
  for (int i = 0; i < nBatch; ++i)
    for (int j = 0; j < nFeature; ++j) {
      const uint8_t d = direction[(size_t)i * nFeature + j];
      for (int l = 0; l < nOri; ++l)
        grad_in[((size_t)i * nFeature + j) * nOri + (l + d) % nOri] = grad_out[((size_t)i * nFeature + j) * nOri + l];
    }
}

// ---- a11: deformable conv v1 pieces -----------------------------------------
struct DcnGeom {
  int C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg, Ho, Wo;
};

static DcnGeom mk_geom(int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw,
                       int dh, int dw, int B, int dg) {
  DcnGeom g{C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg, 0, 0};
  g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;  // ops/dcn_v1.py:328-329
  g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  return g;
}

// ops/dcn_v1.py:25-56
static float dcn_bilinear(const float* im, int ld, int H, int W, float h, float w) {
  int hl = (int)std::floor(h), wl = (int)std::floor(w);
  int hh = hl + 1, wh = wl + 1;
  float lh = h - hl, lw = w - wl;
  float uh = 1 - lh, uw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (hl >= 0 && wl >= 0) v1 = im[hl * ld + wl];
  if (hl >= 0 && wh <= W - 1) v2 = im[hl * ld + wh];
  if (hh <= H - 1 && wl >= 0) v3 = im[hh * ld + wl];
  if (hh <= H - 1 && wh <= W - 1) v4 = im[hh * ld + wh];
  float w1 = uh * uw, w2 = uh * lw, w3 = lh * uw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

// ops/dcn_v1.py:132-184.  im (B,C,H,W); offset (B, dg*2*kh*kw, Ho, Wo);
// col (C*kh*kw, B, Ho, Wo).
void oracle_deform_im2col(const float* im, const float* offset, int C, int H, int W, int kh,
                          int kw, int ph, int pw, int sh, int sw, int dh, int dw, int B, int dg,
                          float* col) {
  DcnGeom g = mk_geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg);
  const int cpg = C / dg;
  const size_t plane = (size_t)g.Ho * g.Wo;
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < B; ++b)
      for (int ho = 0; ho < g.Ho; ++ho)
        for (int wo = 0; wo < g.Wo; ++wo) {
          const float* imp = im + ((size_t)b * C + c) * H * W;
          const float* offp = offset + ((size_t)b * dg + c / cpg) * 2 * kh * kw * plane;
          int h_in = ho * sh - ph, w_in = wo * sw - pw;
          for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
              int tap = i * kw + j;
              float oh = offp[(size_t)(2 * tap) * plane + (size_t)ho * g.Wo + wo];
              float ow = offp[(size_t)(2 * tap + 1) * plane + (size_t)ho * g.Wo + wo];
              float h_im = h_in + i * dh + oh;
              float w_im = w_in + j * dw + ow;
              float v = 0;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                v = dcn_bilinear(imp, W, H, W, h_im, w_im);
              col[(((size_t)(c * kh * kw + tap) * B + b) * g.Ho + ho) * g.Wo + wo] = v;
            }
        }
}

// ops/dcn_v1.py:58-84
static float dcn_grad_weight(float ah, float aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  int hl = (int)std::floor(ah), wl = (int)std::floor(aw);
  int hh = hl + 1, wh = wl + 1;
  float wt = 0;
  if (h == hl && w == wl) wt = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) wt = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) wt = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) wt = (ah + 1 - h) * (aw + 1 - w);
  return wt;
}

// ops/dcn_v1.py:186-241.  Scatter in ascending column-index order (the
// reference uses atomicAdd, i.e. an unordered sum; this order is the one a
// single host thread walking CUDA_KERNEL_LOOP produces).  grad_im must be
// zeroed by the caller (:405 cudaMemsetAsync).
void oracle_deform_col2im(const float* col, const float* offset, int C, int H, int W, int kh,
                          int kw, int ph, int pw, int sh, int sw, int dh, int dw, int B, int dg,
                          float* grad_im) {
  DcnGeom g = mk_geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg);
  const int cpg = C / dg;
  const size_t plane = (size_t)g.Ho * g.Wo;
  size_t index = 0;
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < kh; ++i)
      for (int j = 0; j < kw; ++j)
        for (int b = 0; b < B; ++b)
          for (int ho = 0; ho < g.Ho; ++ho)
            for (int wo = 0; wo < g.Wo; ++wo, ++index) {
              const float* offp = offset + ((size_t)b * dg + c / cpg) * 2 * kh * kw * plane;
              int tap = i * kw + j;
              float oh = offp[(size_t)(2 * tap) * plane + (size_t)ho * g.Wo + wo];
              float ow = offp[(size_t)(2 * tap + 1) * plane + (size_t)ho * g.Wo + wo];
              float fh = (ho * sh - ph) + i * dh + oh;
              float fw = (wo * sw - pw) + j * dw + ow;
              float top = col[index];
              int ch = (int)fh, cw = (int)fw;
              for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                  int y = ch + dy, x = cw + dx;
                  if (y >= 0 && y < H && x >= 0 && x < W && std::fabs(fh - y) < 1 &&
                      std::fabs(fw - x) < 1) {
                    float wt = dcn_grad_weight(fh, fw, y, x, H, W);
                    grad_im[(((size_t)b * C + c) * H + y) * W + x] += wt * top;
                  }
                }
            }
}

// ops/dcn_v1.py:86-129
static float dcn_coord_weight(float ah, float aw, int H, int W, const float* im, int ld, int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  int hl = (int)std::floor(ah), wl = (int)std::floor(aw);
  int hh = hl + 1, wh = wl + 1;
  float wt = 0;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) wt += -1 * (wl + 1 - aw) * im[hl * ld + wl];
    if (hl >= 0 && wh <= W - 1) wt += -1 * (aw - wl) * im[hl * ld + wh];
    if (hh <= H - 1 && wl >= 0) wt += (wl + 1 - aw) * im[hh * ld + wl];
    if (hh <= H - 1 && wh <= W - 1) wt += (aw - wl) * im[hh * ld + wh];
  } else {
    if (hl >= 0 && wl >= 0) wt += -1 * (hl + 1 - ah) * im[hl * ld + wl];
    if (hl >= 0 && wh <= W - 1) wt += (hl + 1 - ah) * im[hl * ld + wh];
    if (hh <= H - 1 && wl >= 0) wt += -1 * (ah - hl) * im[hh * ld + wl];
    if (hh <= H - 1 && wh <= W - 1) wt += (ah - hl) * im[hh * ld + wh];
  }
  return wt;
}

// ops/dcn_v1.py:244-306.  grad_offset (B, dg*2*kh*kw, Ho, Wo).
void oracle_deform_col2im_coord(const float* col, const float* im, const float* offset, int C,
                                int H, int W, int kh, int kw, int ph, int pw, int sh, int sw,
                                int dh, int dw, int B, int dg, float* grad_offset) {
  DcnGeom g = mk_geom(C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg);
  const size_t plane = (size_t)g.Ho * g.Wo;
  const int offc = 2 * kh * kw * dg;
  const int cpg_col = C * kh * kw / dg;  // "channel_per_deformable_group" of :365
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < offc; ++c)
      for (int h = 0; h < g.Ho; ++h)
        for (int w = 0; w < g.Wo; ++w) {
          int grp = c / (2 * kh * kw);
          const float* colp = col + (size_t)grp * cpg_col * B * plane;
          const float* imp = im + ((size_t)b * dg + grp) * (cpg_col / kh / kw) * H * W;
          const float* offp = offset + ((size_t)b * dg + grp) * 2 * kh * kw * plane;
          int oc = c - grp * 2 * kh * kw;
          float val = 0;
          int cnt = 0;
          for (int cc = oc / 2; cc < cpg_col; cc += kh * kw, ++cnt) {
            size_t pos = (((size_t)cc * B + b) * g.Ho + h) * g.Wo + w;
            int dir = oc % 2;
            int j = (int)((pos / g.Wo / g.Ho / B) % kw);
            int i = (int)((pos / g.Wo / g.Ho / B / kw) % kh);
            int tap = i * kw + j;
            float oh = offp[(size_t)(2 * tap) * plane + (size_t)h * g.Wo + w];
            float ow = offp[(size_t)(2 * tap + 1) * plane + (size_t)h * g.Wo + w];
            float ih = (h * sh - ph) + i * dh + oh;
            float iw = (w * sw - pw) + j * dw + ow;
            if (ih <= -1 || iw <= -1 || ih >= H || iw >= W) ih = iw = -2;
            float wt = dcn_coord_weight(ih, iw, H, W, imp + (size_t)cnt * H * W, W, dir);
            val += wt * colp[pos];
          }
          grad_offset[(((size_t)b * offc + c) * g.Ho + h) * g.Wo + w] = val;
        }
}

// ---- a18: ROIAlignRotated_v1 -------------------------------------------------
struct Bil {
  float w1, w2, w3, w4;
  int xl, xh, yl, yh;
  bool ok;
};

// ops/roi_align_rotated_v1.py:24-68 (value) / :149-190 (weights)
static Bil rroi_bilinear(int H, int W, float y, float x) {
  Bil r{0, 0, 0, 0, -1, -1, -1, -1, false};
  if (y < -1.0 || y > H || x < -1.0 || x > W) return r;
  if (y < 0) y = 0;
  if (x < 0) x = 0;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (float)yl;
  } else
    yh = yl + 1;
  if (xl >= W - 1) {
    xh = xl = W - 1;
    x = (float)xl;
  } else
    xh = xl + 1;
  float ly = y - yl, lx = x - xl;
  float hy = (float)(1. - ly), hx = (float)(1. - lx);
  r = Bil{hy * hx, hy * lx, ly * hx, ly * lx, xl, xh, yl, yh, true};
  return r;
}

struct RoiFrame {
  int batch;
  float cw, ch, rw, rh, bin_h, bin_w, start_h, start_w, cs, sn;
  int gh, gw;
};

// ops/roi_align_rotated_v1.py:84-118; v0 = ops/roi_align_rotated.py:72-101 (centre without the -0.5 shift :76-77,
// frame rotated the other way :116-117 -- written below as sn = -sin so that both variants share the expressions
// x = xx*cs + yy*sn, y = yy*cs - xx*sn, which are bit-identical to v0's xx*cs - yy*sin, xx*sin + yy*cs).
static RoiFrame roi_frame(const float* roi, float scale, int sample_num, int PH, int PW, int v0) {
  RoiFrame f;
  f.batch = (int)roi[0];
  f.cw = roi[1] * scale - (v0 ? 0.f : 0.5f);
  f.ch = roi[2] * scale - (v0 ? 0.f : 0.5f);
  f.rw = std::max(roi[3] * scale, 1.f);
  f.rh = std::max(roi[4] * scale, 1.f);
  float theta = roi[5];
  f.bin_h = f.rh / (float)PH;
  f.bin_w = f.rw / (float)PW;
  f.gh = sample_num > 0 ? sample_num : (int)std::ceil(f.rh / PH);
  f.gw = sample_num > 0 ? sample_num : (int)std::ceil(f.rw / PW);
  f.start_h = (float)(-f.rh / 2.0);
  f.start_w = (float)(-f.rw / 2.0);
  f.cs = std::cos(theta);  // float overloads, as scalar_t=float in the reference
  f.sn = v0 ? -std::sin(theta) : std::sin(theta);
  return f;
}

// ops/roi_align_rotated_v1.py:71-147.  feat (N,C,H,W); rois (R,6)=(batch,cx,cy,w,h,theta);
// out (R,C,PH,PW).
static void rroi_align_forward(const float* feat, const float* rois, int R, int C, int H,
                               int W, int PH, int PW, float scale, int sample_num, int v0,
                               float* out) {
  for (int n = 0; n < R; ++n) {
    RoiFrame f = roi_frame(rois + (size_t)n * 6, scale, sample_num, PH, PW, v0);
    const float count = (float)std::max(f.gh * f.gw, 1);
    for (int c = 0; c < C; ++c) {
      const float* fp = feat + ((size_t)f.batch * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float acc = 0.f;
          for (int iy = 0; iy < f.gh; ++iy) {
            float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
            for (int ix = 0; ix < f.gw; ++ix) {
              float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
              float x = xx * f.cs + yy * f.sn + f.cw;
              float y = yy * f.cs - xx * f.sn + f.ch;
              Bil b = rroi_bilinear(H, W, y, x);
              float v = 0;
              if (b.ok)
                v = b.w1 * fp[b.yl * W + b.xl] + b.w2 * fp[b.yl * W + b.xh] +
                    b.w3 * fp[b.yh * W + b.xl] + b.w4 * fp[b.yh * W + b.xh];
              acc += v;
            }
          }
          out[(((size_t)n * C + c) * PH + ph) * PW + pw] = acc / count;
        }
    }
  }
}

// ops/roi_align_rotated_v1.py:193-298.  grad_feat must be zeroed by the caller
// (:345 cudaMemsetAsync); sums run in ascending output-index order.
static void rroi_align_backward(const float* grad_out, const float* rois, int R, int C, int H,
                                int W, int PH, int PW, float scale, int sample_num, int v0,
                                float* grad_feat) {
  for (int n = 0; n < R; ++n) {
    RoiFrame f = roi_frame(rois + (size_t)n * 6, scale, sample_num, PH, PW, v0);
    const float count = (float)(f.gh * f.gw);
    for (int c = 0; c < C; ++c) {
      float* gp = grad_feat + ((size_t)f.batch * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float top = grad_out[(((size_t)n * C + c) * PH + ph) * PW + pw];
          for (int iy = 0; iy < f.gh; ++iy) {
            float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
            for (int ix = 0; ix < f.gw; ++ix) {
              float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
              float x = xx * f.cs + yy * f.sn + f.cw;
              float y = yy * f.cs - xx * f.sn + f.ch;
              Bil b = rroi_bilinear(H, W, y, x);
              float g1 = top * b.w1 / count, g2 = top * b.w2 / count;
              float g3 = top * b.w3 / count, g4 = top * b.w4 / count;
              if (b.xl >= 0 && b.xh >= 0 && b.yl >= 0 && b.yh >= 0) {
                gp[b.yl * W + b.xl] += g1;
                gp[b.yl * W + b.xh] += g2;
                gp[b.yh * W + b.xl] += g3;
                gp[b.yh * W + b.xh] += g4;
              }
            }
          }
        }
    }
  }
}

void oracle_rroi_align_v1_forward(const float* feat, const float* rois, int R, int C, int H, int W, int PH, int PW,
                                  float scale, int sample_num, float* out) {
  rroi_align_forward(feat, rois, R, C, H, W, PH, PW, scale, sample_num, 0, out);
}
void oracle_rroi_align_v1_backward(const float* grad_out, const float* rois, int R, int C, int H, int W, int PH,
                                   int PW, float scale, int sample_num, float* grad_feat) {
  rroi_align_backward(grad_out, rois, R, C, H, W, PH, PW, scale, sample_num, 0, grad_feat);
}
// f4: ROIAlignRotated (v0), ops/roi_align_rotated.py:59-126 / :170-254
void oracle_rroi_align_v0_forward(const float* feat, const float* rois, int R, int C, int H, int W, int PH, int PW,
                                  float scale, int sample_num, float* out) {
  rroi_align_forward(feat, rois, R, C, H, W, PH, PW, scale, sample_num, 1, out);
}
void oracle_rroi_align_v0_backward(const float* grad_out, const float* rois, int R, int C, int H, int W, int PH,
                                   int PW, float scale, int sample_num, float* grad_feat) {
  rroi_align_backward(grad_out, rois, R, C, H, W, PH, PW, scale, sample_num, 1, grad_feat);
}

// ---- f4: FeatureRefine (R3Det) ---------------------------------------------------
// ops/fr.py:113-173 (forward), :175-232 (backward).  feat (N,C,H,W); boxes (N,H,W,5); points in {1,5}.
// Entry 0 of the box is the ROW coordinate and entry 1 the COLUMN (:131-133), as written in the reference.
static void fr_points(const float* b, float scale, int points, float* px, float* py) {
  float roi_y = b[0] * scale, roi_x = b[1] * scale;
  for (int i = 0; i < 5; ++i) px[i] = py[i] = 0.f;
  px[0] = roi_x;
  py[0] = roi_y;
  if (points > 1) {
    float roi_w = b[2] * scale, roi_h = b[3] * scale, roi_a = b[4];
    float w_2 = roi_w / 2, h_2 = roi_h / 2;
    float cosa = cosf(roi_a), sina = sinf(roi_a);
    float wx = cosa * w_2, wy = sina * w_2;
    float hx = -sina * h_2, hy = cosa * h_2;
    px[1] = roi_x + wx + hx; py[1] = roi_y + wy + hy;
    px[2] = roi_x - wx + hx; py[2] = roi_y - wy + hy;
    px[3] = roi_x - wx - hx; py[3] = roi_y - wy - hy;
    px[4] = roi_x + wx - hx; py[4] = roi_y + wy - hy;
  }
}

void oracle_feature_refine_forward(const float* feat, const float* boxes, int N, int C, int H, int W, float scale,
                                   int points, float* out) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float* fp = feat + ((size_t)n * C + c) * H * W;
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          float px[5], py[5];
          fr_points(boxes + (((size_t)n * H + h) * W + w) * 5, scale, points, px, py);
          float v = fp[h * W + w];
          for (int i = 0; i < points; ++i) {
            Bil b = rroi_bilinear(H, W, py[i], px[i]);  // same function as fr.py:18-66
            if (b.ok)
              v += b.w1 * fp[b.yl * W + b.xl] + b.w2 * fp[b.yl * W + b.xh] + b.w3 * fp[b.yh * W + b.xl] +
                   b.w4 * fp[b.yh * W + b.xh];
          }
          out[((size_t)n * C + c) * H * W + h * W + w] = v;
        }
    }
}

// grad_in must be zeroed by the caller (fr.py:245 jt.zeros_like); sums in ascending index order.
void oracle_feature_refine_backward(const float* top, const float* boxes, int N, int C, int H, int W, float scale,
                                    int points, float* grad_in) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      float* gp = grad_in + ((size_t)n * C + c) * H * W;
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          float px[5], py[5];
          fr_points(boxes + (((size_t)n * H + h) * W + w) * 5, scale, points, px, py);
          float t = top[((size_t)n * C + c) * H * W + h * W + w];
          gp[h * W + w] += t;
          for (int i = 0; i < points; ++i) {
            Bil b = rroi_bilinear(H, W, py[i], px[i]);
            if (b.ok) {
              gp[b.yl * W + b.xl] += t * b.w1;
              gp[b.yl * W + b.xh] += t * b.w2;
              gp[b.yh * W + b.xl] += t * b.w3;
              gp[b.yh * W + b.xh] += t * b.w4;
            }
          }
        }
    }
}

// ---- f4: convex_sort (Graham scan over prepared order) ------------------------------
// ops/convex_sort.py:93-154 (CPU loop; the CUDA kernel :5-64 is the same statement sequence).
// x, y, m (nbs, npts); start (nbs); order (nbs, npts); out (nbs, index_size) prefilled with -1 by the caller.
void oracle_convex_sort_scan(const float* x, const float* y, const float* m, const int* start, const int* order,
                             int nbs, int npts, int circular, int* out) {
  const int index_size = circular ? npts + 1 : npts;
  for (int i = 0; i < nbs; ++i) {
    const float *sx = x + (size_t)i * npts, *sy = y + (size_t)i * npts, *sm = m + (size_t)i * npts;
    const int* so = order + (size_t)i * npts;
    int* hull = out + (size_t)i * index_size;
    const int s0 = start[i];
    hull[0] = s0;
    int top = 0;
    for (int k = 0; k < npts; ++k) {
      const int j = so[k];
      if (j == s0 || sm[j] < 0.5) continue;
      const float x0 = sx[j], y0 = sy[j];
      float x1 = sx[hull[top]], y1 = sy[hull[top]];
      const float d = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
      if (d < 0.000001) continue;  // double literal: the float is widened for the comparison (:112)
      if (top < 2) {
        hull[++top] = j;
        continue;
      }
      float x2 = sx[hull[top - 1]], y2 = sy[hull[top - 1]];
      for (;;) {
        const float t = (x1 - x2) * (y0 - y2) - (y1 - y2) * (x0 - x2);
        if (t >= 0) {
          hull[++top] = j;
          break;
        }
        if (top <= 1) {
          hull[top] = j;
          break;
        }
        --top;
        x1 = sx[hull[top]];
        y1 = sy[hull[top]];
        x2 = sx[hull[top - 1]];
        y2 = sy[hull[top - 1]];
      }
    }
    if (circular) hull[++top] = hull[0];
  }
}

// ---- f4: poly_nms (in-model, fp32) ----------------------------------------------------------
// ops/nms_poly.py:17-132 (devPolyIoU), :135-183 (mask), :195-207 (sweep).  Plain float arithmetic in the order the
// reference writes it (compile with -ffp-contract=off: see Makefile).  The staging ring `pp` of one quadrilateral
// pair starts from zeros (the reference's is uninitialised stack memory that is read only after a failed
// lineCross).
namespace polyf {
struct P2 {
  float x, y;
};
static inline int sig(float d) { return (d > 1e-8) - (d < -1e-8); }
static inline bool same(P2 a, P2 b) { return sig(a.x - b.x) == 0 && sig(a.y - b.y) == 0; }
static inline float cross(P2 o, P2 a, P2 b) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }
static float area(P2* ps, int n) {
  ps[n] = ps[0];
  float res = 0;
  for (int i = 0; i < n; ++i) res += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
  return (float)(res / 2.0);
}
static void cut(P2* p, int& n, P2 a, P2 b, P2* pp) {
  int m = 0;
  p[n] = p[0];
  for (int i = 0; i < n; ++i) {
    float s1 = cross(a, b, p[i]), s2 = cross(a, b, p[i + 1]);
    if (sig(s1) > 0) pp[m++] = p[i];
    if (sig(s1) != sig(s2)) {
      if (!(sig(s1) == 0 && sig(s2) == 0) && sig(s2 - s1) != 0) {
        pp[m].x = (p[i].x * s2 - p[i + 1].x * s1) / (s2 - s1);
        pp[m].y = (p[i].y * s2 - p[i + 1].y * s1) / (s2 - s1);
      }
      ++m;
    }
  }
  n = 0;
  for (int i = 0; i < m; ++i)
    if (!i || !same(pp[i], pp[i - 1])) p[n++] = pp[i];
  while (n > 1 && same(p[n - 1], p[0])) --n;
}
static float tri(P2 a, P2 b, P2 c, P2 d, P2* pp) {
  P2 o{0, 0};
  int s1 = sig(cross(o, a, b)), s2 = sig(cross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.f;
  if (s1 == -1) std::swap(a, b);
  if (s2 == -1) std::swap(c, d);
  P2 p[12] = {o, a, b};
  int n = 3;
  cut(p, n, o, c, pp);
  cut(p, n, c, d, pp);
  cut(p, n, d, o, pp);
  float res = std::fabs(area(p, n));
  if (s1 * s2 == -1) res = -res;
  return res;
}
static float quad_iou(const float* q1, const float* q2) {
  P2 a[6], b[6], pp[12];
  for (int i = 0; i < 4; ++i) {
    a[i] = P2{q1[2 * i], q1[2 * i + 1]};
    b[i] = P2{q2[2 * i], q2[2 * i + 1]};
  }
  for (int i = 0; i < 12; ++i) pp[i] = P2{0, 0};
  if (area(a, 4) < 0) std::reverse(a, a + 4);
  if (area(b, 4) < 0) std::reverse(b, b + 4);
  a[4] = a[0];
  b[4] = b[0];
  float inter = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) inter += tri(a[i], a[i + 1], b[j], b[j + 1], pp);
  float uni = std::fabs(area(a, 4)) + std::fabs(area(b, 4)) - inter;
  return uni == 0 ? (inter + 1) / (uni + 1) : inter / uni;
}
}  // namespace polyf

void oracle_poly_iou_f32(const float* p1, int n1, const float* p2, int n2, float* out) {
  for (int i = 0; i < n1; ++i)
    for (int j = 0; j < n2; ++j) out[(size_t)i * n2 + j] = polyf::quad_iou(p1 + (size_t)i * 8, p2 + (size_t)j * 8);
}

// dets_sorted (n, 9) descending score; keep[i] = 1 iff kept; suppression on iou > thr (:179), greedy (:197-206)
void oracle_poly_nms_sorted(const float* dets, int n, float thr, unsigned char* keep) {
  std::vector<unsigned char> removed(n > 0 ? n : 1, 0);
  for (int i = 0; i < n; ++i) {
    keep[i] = 0;
    if (removed[i]) continue;
    keep[i] = 1;
    for (int j = i + 1; j < n; ++j)
      if (polyf::quad_iou(dets + (size_t)i * 9, dets + (size_t)j * 9) > thr) removed[j] = 1;
  }
}

// ---- a4: MaxIoUAssigner.assign_wrt_overlaps -----------------------------------
// models/boxes/assigner.py:111-170.  overlaps (K, A) row-major.  Tie rule for
// both argmax calls: FIRST index of the maximum (Jittor's rule is unpinned).
// neg range [neg_lo, neg_hi): float neg_iou_thr => neg_lo = 0 (:139-140).
// Outputs: gt_inds (A) int32 (-1 ignore / 0 neg / k+1), max_ov (A),
// labels (A) (assigned_labels_filled where not positive) when gt_labels != NULL.
void oracle_assign_wrt_overlaps(const float* ov, int K, int A, float pos_thr, float neg_lo,
                                float neg_hi, float min_pos_iou, int match_low_quality,
                                int gt_max_assign_all, const int* gt_labels, int labels_filled,
                                int* gt_inds, float* max_ov, int* labels) {
  std::vector<int> argmax(A, 0);
  for (int j = 0; j < A; ++j) {
    float best = ov[j];
    int bi = 0;
    for (int i = 1; i < K; ++i) {
      float v = ov[(size_t)i * A + j];
      if (v > best) {
        best = v;
        bi = i;
      }
    }
    max_ov[j] = best;
    argmax[j] = bi;
    int g = -1;
    if (best >= neg_lo && best < neg_hi) g = 0;
    if (best >= pos_thr) g = bi + 1;
    gt_inds[j] = g;
  }
  if (match_low_quality) {
    for (int i = 0; i < K; ++i) {
      const float* row = ov + (size_t)i * A;
      float gmax = row[0];
      int gj = 0;
      for (int j = 1; j < A; ++j)
        if (row[j] > gmax) {
          gmax = row[j];
          gj = j;
        }
      if (gmax >= min_pos_iou) {
        if (gt_max_assign_all) {
          for (int j = 0; j < A; ++j)
            if (row[j] == gmax) gt_inds[j] = i + 1;
        } else {
          gt_inds[gj] = i + 1;
        }
      }
    }
  }
  if (gt_labels && labels) {
    for (int j = 0; j < A; ++j)
      labels[j] = gt_inds[j] > 0 ? gt_labels[gt_inds[j] - 1] : labels_filled;
  }
}

}  // extern "C"
