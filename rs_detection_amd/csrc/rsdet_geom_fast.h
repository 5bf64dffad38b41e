// rsdet_geom_fast.h -- rotated-rectangle intersection area by Green's theorem, ONE lane per pair, registers only.
//
// Tier 1 of the two-tier IoU (DESIGN.md "two-tier IoU"): the reference's clipper
// (/root/reference/python/jdet/ops/box_iou_rotated.py:281-310: 16 edge solves + 8 containment tests + Graham hull +
// fan area, ~2 800 lane-instructions per overlapping pair in rsdet_geom.h) is kept operation by operation wherever a
// decision depends on it (thresholds, ties, slivers); every other pair only needs its IoU to 1e-4 and gets this:
//
//   map both boxes into box A's frame scaled to the square [-1, 1]^2 (an affine map: areas scale by |u_A| |v_A|);
//   area(B' n square) = closed line integral over B's 4 edges of  clamp(x, -1, 1) * [|y| <= 1] dy   (Green, with
//   d/dx clamp(x) = [|x| <= 1]); per edge: clip the parameter range to |y| <= 1, then the mean of clamp(x) over a
//   linear x range in closed form.  No point lists, no sorting, no branches on geometry, no LDS: ~250 VALU
//   instructions.  The integrand is continuous in the corner coordinates, so coincident / collinear edges (identical
//   boxes, shared edge lines) are not special cases: rounding moves the result by O(1e-7) of A's area.
// Measured against the reference's own CPU source on 1.2e7 overlapping pairs (tests/test_gpu_iou_fast.py): max
// |difference| of the IoU < 3e-6; the two-tier callers budget 2e-5.
#pragma once
#include "rsdet_geom.h"

namespace rsdet {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
__host__ __device__ inline float fast_rcp(float x) { return 1.0f / x; }
#endif

// integral of clamp(x, -1, 1) dy along the edge (x0, y0) -> (x1, y1), restricted to |y| <= 1
__host__ __device__ __forceinline__ float green_edge(float x0, float y0, float x1, float y1) {
  const float dx = x1 - x0, dy = y1 - y0;
  const float r = fast_rcp(dy);
  const float ta = (-1.0f - y0) * r, tb = (1.0f - y0) * r;
  const float tlo = fmaxf(0.0f, fminf(ta, tb)), thi = fminf(1.0f, fmaxf(ta, tb));
  const bool valid = thi > tlo && fabsf(dy) > 1e-30f;  // NaN-safe: a NaN compare is false
  const float xa = __builtin_fmaf(dx, tlo, x0), xb = __builtin_fmaf(dx, thi, x0);
  const float yl = dy * (thi - tlo);
  const float lo = fminf(xa, xb), hi = fmaxf(xa, xb);
  const float L1 = fmaxf(0.0f, fminf(hi, -1.0f) - lo);       // part left of the square: clamp = -1
  const float L3 = fmaxf(0.0f, hi - fmaxf(lo, 1.0f));        // part right of it: clamp = +1
  const float m0 = fmaxf(lo, -1.0f), m1 = fminf(hi, 1.0f);
  const float Lm = fmaxf(0.0f, m1 - m0);                     // part inside: clamp = x, mean (m0 + m1) / 2
  const float num = (L3 - L1) + Lm * (m1 + m0) * 0.5f, den = L1 + L3 + Lm;
  const float g = den > 0.0f ? num * fast_rcp(den) : fminf(fmaxf(lo, -1.0f), 1.0f);
  return valid ? yl * g : 0.0f;
}

// Tolerances of the two-tier scheme.  The reference's clipper is exact arithmetic plus ABSOLUTE tolerances and one
// indexing slip: `dist[]` is filled BEFORE std::sort and read AFTER it (box_iou_rotated.py:199-212), so when the
// lowest candidate point has a near-duplicate (|d|^2 <= 1e-8: a corner of one box lying on an edge of the other, to
// 1e-4 px) the scan starts from the wrong point and real hull vertices are dropped -- the reference returns e.g. 0.0
// for a true IoU of 0.06, or 0.33 for identical boxes.  That happens for ~1e-5 of the overlapping pairs of random
// float boxes and for a large share of integer-coordinate axis-aligned ones.  The exact path (rsdet_geom.h)
// reproduces it bit for bit; the fast path cannot, so it flags every pair in which a corner of one box is within
// kFastCollinear px of an edge segment of the other (a ~100x margin around the trigger) and the caller runs the
// reference-order clipper there.
constexpr float kFastCollinear = 1e-2f;  // px between a corner and an edge segment of the other box
constexpr float kFastGap = 2e-3f;        // px of clear separation that make the reference's result exactly 0
constexpr float kFastSliver = 1e-6f;     // IoUs below this are recomputed by the reference-order clipper (exact zeros:
                                         // measured, the Green sum is exactly 0 wherever the reference's is, and never
                                         // above 1e-7 where the reference returns 0 by its `num <= 2` rule)
constexpr float kFastBudget = 2e-5f;     // |fast - reference| outside the danger zone (measured < 3e-6): decision margin

// corner (x, y) of the other box, in the frame where this box is [-1, 1]^2, within (tx, ty) of one of its edge segments
__host__ __device__ __forceinline__ bool near_square_edge(float x, float y, float tx, float ty) {
  const float ax = fabsf(x), ay = fabsf(y);
  return (fabsf(ax - 1.f) < tx && ay < 1.f + ty) || (fabsf(ay - 1.f) < ty && ax < 1.f + tx);
}

// Intersection-over-union of two prepared boxes by the Green integral; `danger` is set where the reference's own
// result may differ from the true value (see above) -- callers must use the reference-order clipper there.
// NaN in -> NaN out.  The LARGER box is the frame (its coordinates of the other box stay O(1)).
// `apart` is set when the boxes are separated by a gap of more than kFastGap px (+ 2e-5 of their size) along one of the
// four edge normals: ~20x the rounding of that very test and ~20x what the reference's own fp32 edge solves and
// containment tests can bridge, so the reference finds no candidate point and returns EXACTLY 0; so does this function.
template <int VERSION>
__host__ __device__ __forceinline__ float pair_iou_fast(const BoxPre& a0, const BoxPre& b0, bool& danger, bool& apart) {
  danger = false;
  apart = false;
  if (lt_1e14(a0.area) || lt_1e14(b0.area)) return 0.f;   // box_iou_rotated.py:288-290
  const bool swap = fabsf(b0.area) > fabsf(a0.area);
  const float acx = swap ? b0.cx : a0.cx, acy = swap ? b0.cy : a0.cy, bcx = swap ? a0.cx : b0.cx, bcy = swap ? a0.cy : b0.cy;
  const float acw = swap ? b0.cw : a0.cw, asw = swap ? b0.sw : a0.sw, ach = swap ? b0.ch : a0.ch, ash = swap ? b0.sh : a0.sh;
  const float bcw = swap ? a0.cw : b0.cw, bsw = swap ? a0.sw : b0.sw, bch = swap ? a0.ch : b0.ch, bsh = swap ? a0.sh : b0.sh;
  const float alu = swap ? b0.lu : a0.lu, alv = swap ? b0.lv : a0.lv, blu = swap ? a0.lu : b0.lu, blv = swap ? a0.lv : b0.lv;
  const float aarea = swap ? b0.area : a0.area;
  const float sg = VERSION == 0 ? 1.f : -1.f;
  const float aux = acw, auy = sg * asw, avx = -sg * ash, avy = ach;
  const float bux = bcw, buy = sg * bsw, bvx = -sg * bsh, bvy = bch;
  const float iu = fast_rcp(aux * aux + auy * auy), iv = fast_rcp(avx * avx + avy * avy);
  const float dx = bcx - acx, dy = bcy - acy;
  // the four dot products of the half-edge vectors serve both frames
  const float duu = bux * aux + buy * auy, duv = bux * avx + buy * avy;   // u_B . u_A, u_B . v_A
  const float dvu = bvx * aux + bvy * auy, dvv = bvx * avx + bvy * avy;   // v_B . u_A, v_B . v_A
  const float dau = dx * aux + dy * auy, dav = dx * avx + dy * avy;       // d . u_A, d . v_A
  const float cX = dau * iu, cY = dav * iv;
  const float uX = duu * iu, uY = duv * iv, vX = dvu * iu, vY = dvv * iv;
  const float p0x = cX - uX - vX, p0y = cY - uY - vY, p1x = cX + uX - vX, p1y = cY + uY - vY;
  const float p2x = cX + uX + vX, p2y = cY + uY + vY, p3x = cX - uX + vX, p3y = cY - uY + vY;
  const float s = (green_edge(p0x, p0y, p1x, p1y) + green_edge(p1x, p1y, p2x, p2y)) +
                  (green_edge(p2x, p2y, p3x, p3y) + green_edge(p3x, p3y, p0x, p0y));
  // the square has area 4 = A's area in its own frame; orientation of B' may be either (negative sizes): |s|
  const float inter = fabsf(s) * 0.25f * fabsf(aarea);
  // ---- danger zone, corners of B against the edges of A (frame units: 1 = |u_A| resp. |v_A| px) ...
  const float tx = kFastCollinear * alu * iu, ty = kFastCollinear * alv * iv;
  bool dz = near_square_edge(p0x, p0y, tx, ty) || near_square_edge(p1x, p1y, tx, ty) ||
            near_square_edge(p2x, p2y, tx, ty) || near_square_edge(p3x, p3y, tx, ty);
  // ... and corners of A against the edges of B, in B's frame
  const float ju = fast_rcp(bux * bux + buy * buy), jv = fast_rcp(bvx * bvx + bvy * bvy);
  const float dbu = dx * bux + dy * buy, dbv = dx * bvx + dy * bvy;       // d . u_B, d . v_B
  const float eX = -dbu * ju, eY = -dbv * jv;
  const float wX = duu * ju, wY = dvu * jv, zX = duv * ju, zY = dvv * jv;     // u_A, v_A in B's frame
  const float sx = kFastCollinear * blu * ju, sy = kFastCollinear * blv * jv;
  dz = dz || near_square_edge(eX - wX - zX, eY - wY - zY, sx, sy) || near_square_edge(eX + wX - zX, eY + wY - zY, sx, sy) ||
       near_square_edge(eX + wX + zX, eY + wY + zY, sx, sy) || near_square_edge(eX - wX + zX, eY - wY + zY, sx, sy);
  danger = dz;
  // ---- separation along the four edge normals, in px (the dot products above are projections times |axis|)
  const float gu = (fabsf(dau) - (alu * alu + fabsf(duu) + fabsf(dvu))) * alu * iu;
  const float gv = (fabsf(dav) - (alv * alv + fabsf(duv) + fabsf(dvv))) * alv * iv;
  const float hu = (fabsf(dbu) - (blu * blu + fabsf(duu) + fabsf(duv))) * blu * ju;
  const float hv = (fabsf(dbv) - (blv * blv + fabsf(dvu) + fabsf(dvv))) * blv * jv;
  const float gap = fmaxf(fmaxf(gu, gv), fmaxf(hu, hv));
  apart = gap > kFastGap + 2e-5f * ((alu + alv) + (blu + blv));
  if (apart) return 0.f;
  return inter / (a0.area + b0.area - inter);   // :307-309
}

}  // namespace rsdet
