// rsdet_geom.h -- device-side rotated-rectangle intersection for gfx950.
//
// Computes what JDet's single_box_iou_rotated computes
// (/root/reference/python/jdet/ops/box_iou_rotated.py:53-310, CPU flavour of the
// hull sort :317-325) with the SAME fp32 operation order, so that results agree
// with the CPU path to the last bit in the generic case (this TU must be built
// with -ffp-contract=off; the reference CPU build has no FMA).
//
// MI355X-first layout instead of the reference's per-thread 24-point arrays
// (which spill to scratch on a GPU):
//   * per-box trigonometry (fp64 sincos, as :59-61) is hoisted out of the pair
//     loop into a 9-float "prepared box" (BoxPre);
//   * the <=24 candidate points of a pair live in LDS, slot-major
//     ([slot][thread]) so lane l touches bank (2*l)%64 -- conflict free for
//     ds_read/write_b64 -- and the hull is built in place (no second array).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsdet {

struct BoxPre {
  float cx, cy;      // raw centre
  float cw, sw;      // 0.5*cos(a)*w , 0.5*sin(a)*w   (cosTheta2*w, sinTheta2*w)
  float ch, sh;      // 0.5*cos(a)*h , 0.5*sin(a)*h
  float area;        // w*h
  float rad;         // conservative circumscribed radius (early-out only)
};

__device__ __forceinline__ BoxPre prepare_box(const float* __restrict__ b) {
  BoxPre p;
  float w = b[2], h = b[3];
  double theta = (double)b[4];
  double sd, cd;
  sincos(theta, &sd, &cd);
  float c2 = (float)cd * 0.5f;
  float s2 = (float)sd * 0.5f;
  p.cx = b[0];
  p.cy = b[1];
  p.cw = c2 * w;
  p.sw = s2 * w;
  p.ch = c2 * h;
  p.sh = s2 * h;
  p.area = w * h;
  // |half diagonal| <= 0.5*(|w|+|h|); padded so the test below stays conservative
  // against every rounding in the exact path.
  p.rad = 0.5f * (fabsf(w) + fabsf(h)) * 1.0001f + 1e-3f;
  return p;
}

// true  => the exact algorithm is guaranteed to find no intersection point and no
//          contained corner, i.e. the reference returns exactly 0.0f.
__device__ __forceinline__ bool surely_disjoint(const BoxPre& a, const BoxPre& b) {
  float dx = a.cx - b.cx, dy = a.cy - b.cy;
  float r = a.rad + b.rad;
  // NaN/Inf inputs fail this test and take the exact path, like the reference.
  return dx * dx + dy * dy > r * r * 1.0001f;
}

struct F2 {
  float x, y;
};
__device__ __forceinline__ F2 f2sub(F2 a, F2 b) { return F2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ float f2dot(F2 a, F2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float f2cross(F2 a, F2 b) { return a.x * b.y - b.x * a.y; }

// corners in the pair-centred frame; VERSION 0: box_iou_rotated.py:64-71,
// VERSION 1: box_iou_rotated_v1.py:69-72.
template <int VERSION>
__device__ __forceinline__ void corners(const BoxPre& p, float cx, float cy, F2 r[4]) {
  if (VERSION == 0) {
    r[0].x = cx - p.sh - p.cw;
    r[1].x = cx + p.sh - p.cw;
  } else {
    r[0].x = cx + p.sh + p.cw;
    r[1].x = cx - p.sh + p.cw;
  }
  r[0].y = cy + p.ch - p.sw;
  r[1].y = cy - p.ch - p.sw;
  r[2].x = 2 * cx - r[0].x;
  r[2].y = 2 * cy - r[0].y;
  r[3].x = 2 * cx - r[1].x;
  r[3].y = 2 * cy - r[1].y;
}

// LDS scratch accessor: slot k of this thread.
struct Scratch {
  F2* base;      // &lds[threadIdx]
  int stride;    // threads sharing the scratch (block size)
  __device__ __forceinline__ F2 get(int k) const { return base[k * stride]; }
  __device__ __forceinline__ void put(int k, F2 v) const { base[k * stride] = v; }
};

// hull-sort predicate of the reference CPU path (box_iou_rotated.py:317-325)
__device__ __forceinline__ bool hull_less(F2 A, F2 B) {
  float c = f2cross(A, B);
  if (fabs((double)c) < 1e-6) return f2dot(A, A) < f2dot(B, B);
  return c > 0;
}

// Exact pair IoU.  `a` plays box1, `b` box2 (the function is not bitwise symmetric).
template <int VERSION>
__device__ float pair_iou(const BoxPre& a, const BoxPre& b, const Scratch sc) {
  // centre shift, box_iou_rotated.py:288-291 (fp32 here; the reference's detour
  // through double is value-identical except in astronomically rare double roundings)
  float sx = (a.cx + b.cx) * 0.5f, sy = (a.cy + b.cy) * 0.5f;
  if ((double)a.area < 1e-14 || (double)b.area < 1e-14) return 0.f;

  F2 r1[4], r2[4], e1[4], e2[4];
  corners<VERSION>(a, a.cx - sx, a.cy - sy, r1);
  corners<VERSION>(b, b.cx - sx, b.cy - sy, r2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e1[i] = f2sub(r1[(i + 1) & 3], r1[i]);
    e2[i] = f2sub(r2[(i + 1) & 3], r2[i]);
  }

  int n = 0;
  // 16 edge/edge solves (:87-107)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float det = f2cross(e2[j], e1[i]);
      if (fabs((double)det) > 1e-14) {
        F2 d = f2sub(r2[j], r1[i]);
        float t1 = f2cross(e2[j], d) / det;
        float t2 = f2cross(e1[i], d) / det;
        if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
          sc.put(n, F2{r1[i].x + e1[i].x * t1, r1[i].y + e1[i].y * t1});
          ++n;
        }
      }
    }
  }
  // corners of rect1 inside rect2 (:110-129)
  {
    F2 AB = e2[0], DA = e2[3];
    float ABAB = f2dot(AB, AB), ADAD = f2dot(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      F2 AP = f2sub(r1[i], r2[0]);
      float pab = f2dot(AP, AB), pad = -f2dot(AP, DA);
      if (pab >= 0 && pad >= 0 && pab <= ABAB && pad <= ADAD) {
        sc.put(n, r1[i]);
        ++n;
      }
    }
  }
  // corners of rect2 inside rect1 (:132-150)
  {
    F2 AB = e1[0], DA = e1[3];
    float ABAB = f2dot(AB, AB), ADAD = f2dot(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      F2 AP = f2sub(r2[i], r1[0]);
      float pab = f2dot(AP, AB), pad = -f2dot(AP, DA);
      if (pab >= 0 && pad >= 0 && pab <= ABAB && pad <= ADAD) {
        sc.put(n, r2[i]);
        ++n;
      }
    }
  }

  float inter = 0.f;
  if (n > 2) {
    // Graham hull, in place (:155-238, shift_to_zero = true)
    int t = 0;
    F2 best = sc.get(0);
    for (int i = 1; i < n; ++i) {
      F2 p = sc.get(i);
      if (p.y < best.y || (p.y == best.y && p.x < best.x)) {
        best = p;
        t = i;
      }
    }
    // q[i] = p[i] - start ; swap q[0] <-> q[t] ; k = first pre-sort slot off the start
    // point (the CPU reference fills dist[] before std::sort and reads it after, :199-212)
    F2 p0 = sc.get(0);
    int k = n;
    for (int i = 1; i < n; ++i) {
      F2 q = f2sub(sc.get(i), best);
      if (i == t) q = f2sub(p0, best);
      sc.put(i, q);
      if (k == n && (double)f2dot(q, q) > 1e-8) k = i;
    }
    sc.put(0, F2{best.x - best.x, best.y - best.y});
    // insertion sort of q[1..n), libstdc++ __insertion_sort order of comparisons
    F2 first = sc.get(1);
    for (int i = 2; i < n; ++i) {
      F2 v = sc.get(i);
      if (hull_less(v, first)) {
        for (int m = i; m > 1; --m) sc.put(m, sc.get(m - 1));
        sc.put(1, v);
        first = v;
      } else {
        int m = i - 1;
        F2 u = sc.get(m);
        while (hull_less(v, u)) {
          sc.put(m + 1, u);
          --m;
          u = sc.get(m);
        }
        sc.put(m + 1, v);
      }
    }
    if (k < n) {
      // scan (:214-232)
      F2 q0 = sc.get(0);
      sc.put(1, sc.get(k));
      int m = 2;
      for (int i = k + 1; i < n; ++i) {
        F2 qi = sc.get(i);
        while (m > 1) {
          F2 qa = sc.get(m - 2), qb = sc.get(m - 1);
          if (f2cross(f2sub(qi, qa), f2sub(qb, qa)) >= 0)
            --m;
          else
            break;
        }
        sc.put(m, qi);
        ++m;
      }
      // fan area (:240-252)
      if (m > 2) {
        float area = 0.f;
        F2 prev = f2sub(sc.get(1), q0);
        for (int i = 1; i < m - 1; ++i) {
          F2 nxt = f2sub(sc.get(i + 1), q0);
          area += fabsf(f2cross(prev, nxt));
          prev = nxt;
        }
        inter = area * 0.5f;
      }
    }
  }
  return inter / (a.area + b.area - inter);
}

}  // namespace rsdet
