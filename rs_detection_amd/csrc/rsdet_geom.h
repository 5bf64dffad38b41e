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
//     loop into a 10-float "prepared box" (BoxPre);
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
  float lu, lv;      // |half-width vector|, |half-height vector| (early-out only)
};                   // 40 B

__host__ __device__ __forceinline__ BoxPre prepare_box(const float* __restrict__ b) {
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
  p.lu = sqrtf(p.cw * p.cw + p.sw * p.sw);
  p.lv = sqrtf(p.ch * p.ch + p.sh * p.sh);
  return p;
}

// true  => the exact algorithm is guaranteed to find no intersection point and no
//          contained corner, i.e. the reference returns exactly 0.0f.
__host__ __device__ __forceinline__ bool surely_disjoint(const BoxPre& a, const BoxPre& b) {
  float dx = a.cx - b.cx, dy = a.cy - b.cy;
  float r = a.rad + b.rad;
  // NaN/Inf inputs fail this test and take the exact path, like the reference.
  return dx * dx + dy * dy > r * r * 1.0001f;
}

// Second-stage early-out: separating-axis test on the four edge directions with a
// safety margin (1e-4 relative + 0.01 px).  The half-edge vectors are, for VERSION 0
// (box_iou_rotated.py:64-67)  u = (cw, sw), v = (-sh, ch);  VERSION 1 (_v1.py:69-72)
// u = (cw, -sw), v = (sh, ch).  true => the rectangles are disjoint by more than the
// margin, so the reference's clipper finds nothing and returns exactly 0.0f (barring
// its own collinear-edge round-off artefacts of order 1e-10).
template <int VERSION>
__host__ __device__ __forceinline__ bool sat_disjoint(const BoxPre& a, const BoxPre& b) {
  const float sg = VERSION == 0 ? 1.f : -1.f;
  const float dx = b.cx - a.cx, dy = b.cy - a.cy;
  const float aux = a.cw, auy = sg * a.sw, avx = -sg * a.sh, avy = a.ch;
  const float bux = b.cw, buy = sg * b.sw, bvx = -sg * b.sh, bvy = b.ch;
  // axis = a.u (|n| = a.lu): own radius lu^2, other box projected
  float sep = fabsf(dx * aux + dy * auy) -
              ((a.lu * a.lu + fabsf(bux * aux + buy * auy) + fabsf(bvx * aux + bvy * auy)) * 1.0001f + 0.01f * a.lu);
  bool out = sep > 0.f;
  sep = fabsf(dx * avx + dy * avy) -
        ((a.lv * a.lv + fabsf(bux * avx + buy * avy) + fabsf(bvx * avx + bvy * avy)) * 1.0001f + 0.01f * a.lv);
  out |= sep > 0.f;
  sep = fabsf(dx * bux + dy * buy) -
        ((b.lu * b.lu + fabsf(aux * bux + auy * buy) + fabsf(avx * bux + avy * buy)) * 1.0001f + 0.01f * b.lu);
  out |= sep > 0.f;
  sep = fabsf(dx * bvx + dy * bvy) -
        ((b.lv * b.lv + fabsf(aux * bvx + auy * bvy) + fabsf(avx * bvx + avy * bvy)) * 1.0001f + 0.01f * b.lv);
  out |= sep > 0.f;
  return out;
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

// Scratch accessor for the <= 24 candidate points of one pair (LDS, `stride` F2 apart).
struct Scratch {
  F2* base;
  int stride;
  __device__ __forceinline__ F2 get(int k) const { return base[k * stride]; }
  __device__ __forceinline__ void put(int k, F2 v) const { base[k * stride] = v; }
};

// The reference compares fp32 values against double literals (`(double)x < 1e-6` ...).  None of the literals is
// representable in fp32, so each comparison is EXACTLY a fp32 comparison against a neighbouring float -- no
// v_cvt_f64_f32 / v_cmp_f64 (quarter / half rate) in the clipper:
//   (double)x <  1e-6   <=>  x <= 0x358637bd (9.99999997e-07, the largest float below 1e-6)
//   (double)x >  1e-8   <=>  x >= 0x322bcc78 (1.00000008e-08, the smallest float above 1e-8)
//   (double)x <  1e-14  <=>  x <= 0x283424dc (9.99999982e-15)
//   (double)x >  1e-14  <=>  x >= 0x283424dd (1.00000007e-14)
// (NaN fails both forms alike.)
__host__ __device__ __forceinline__ bool lt_1e6(float x) { return x <= __builtin_bit_cast(float, 0x358637bdu); }
__host__ __device__ __forceinline__ bool gt_1e8(float x) { return x >= __builtin_bit_cast(float, 0x322bcc78u); }
__host__ __device__ __forceinline__ bool lt_1e14(float x) { return x <= __builtin_bit_cast(float, 0x283424dcu); }
__host__ __device__ __forceinline__ bool gt_1e14(float x) { return x >= __builtin_bit_cast(float, 0x283424ddu); }

// hull-sort predicate of the reference CPU path (box_iou_rotated.py:317-325)
__device__ __forceinline__ bool hull_less(F2 A, F2 B) {
  float c = f2cross(A, B);
  if (lt_1e6(fabsf(c))) return f2dot(A, A) < f2dot(B, B);
  return c > 0;
}

// ---- hull + area, general path: Graham scan in place on the scratch (:155-252) ------------
__device__ __forceinline__ float hull_area_general(const Scratch sc, int n) {
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
    if (k == n && gt_1e8(f2dot(q, q))) k = i;
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
  if (k >= n) return 0.f;  // hull is a single point (:209-213)
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
  if (m <= 2) return 0.f;
  float area = 0.f;
  F2 prev = f2sub(sc.get(1), q0);
  for (int i = 1; i < m - 1; ++i) {
    F2 nxt = f2sub(sc.get(i + 1), q0);
    area += fabsf(f2cross(prev, nxt));
    prev = nxt;
  }
  return area * 0.5f;
}

// LDS hand-off between lanes of ONE wave: the LDS unit executes a wave's ds_* instructions in
// issue order, so a later ds_read sees an earlier ds_write of another lane; only the compiler
// must be kept from reordering them.  (A C++ fence here lowers to s_waitcnt vmcnt(0) and stalls
// on every outstanding global store -- measured 2-3 us per fence.)
__device__ __forceinline__ void lds_wave_order() { asm volatile("" ::: "memory"); }

// ---- 4 lanes per pair ("quad") ------------------------------------------------------------------
// Why not one thread per pair: the serial clipper is ~2500 dependent instructions (~15 us) and a
// 400 x 21 824 call has only ~1e5 overlapping pairs -- fewer than the chip has lanes -- so kernel
// time was that latency.  Why not 16 lanes per pair: the hull part is redundant across lanes, so
// throughput drops 4x for no latency gain.  A quad is the balance (measured):
//   * lane l solves edge l of box1 against the 4 edges of box2 (:87-107) and tests corner l of
//     each rectangle for containment in the other (:110-150);
//   * points are compacted into the quad's 24-slot LDS scratch in the reference's enumeration
//     order ((i, j) edge pairs, rect1 corners, rect2 corners) with wave ballots + popcounts;
//   * two rectangles in general position give n <= 8 points that are ALL vertices of the convex
//     intersection, so the Graham scan never pops: lane l ranks points l and l+4 with the
//     reference's own sort predicate (:317-325), the sorted ring goes back through LDS, the
//     scan's pop test is evaluated as a pure CHECK, and the fan terms are summed in the
//     reference's order -- bit-identical to the serial path whenever no pop occurs;
//   * anything else (n > 8: duplicated points; a pop; duplicated start point) is finished by the
//     quad's lane 0 on the untouched point list with the serial Graham scan above.
// All 4 lanes of a quad must call this with the same (a, b); the result is quad-uniform.
// No arrays of corners on purpose: an indexed corner table is demoted to private scratch memory
// by the compiler (measured 3 us per pair, and its vmcnt waits drain every older global store).
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global
// store of the wave to be acknowledged (s_waitcnt vmcnt(0)), which would stall a kernel that keeps
// a long zero-fill store stream in flight underneath its LDS phases.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kQuadSlots = 25;  // 24 used; stride 25 x 8 B = 50 dwords: the 16 quads of a wave hit distinct LDS banks

template <int VERSION>
__device__ __forceinline__ float pair_iou_quad(const BoxPre& a, const BoxPre& b, F2* __restrict__ qscr,
                                               int lane) {
  const int l = lane & 3;
  const int qsh = lane & 60;  // bit position of this quad inside a wave ballot
  float sx = (a.cx + b.cx) * 0.5f, sy = (a.cy + b.cy) * 0.5f;
  if (lt_1e14(a.area) || lt_1e14(b.area)) return 0.f;

  // corners 0, 1 (VERSION 0: box_iou_rotated.py:64-67, VERSION 1: _v1.py:69-72); 2, 3 are their
  // mirrors through the centre (:68-71)
  const float c1x = a.cx - sx, c1y = a.cy - sy, c2x = b.cx - sx, c2y = b.cy - sy;
  const F2 a0 = VERSION == 0 ? F2{c1x - a.sh - a.cw, c1y + a.ch - a.sw} : F2{c1x + a.sh + a.cw, c1y + a.ch - a.sw};
  const F2 a1 = VERSION == 0 ? F2{c1x + a.sh - a.cw, c1y - a.ch - a.sw} : F2{c1x - a.sh + a.cw, c1y - a.ch - a.sw};
  const F2 b0 = VERSION == 0 ? F2{c2x - b.sh - b.cw, c2y + b.ch - b.sw} : F2{c2x + b.sh + b.cw, c2y + b.ch - b.sw};
  const F2 b1 = VERSION == 0 ? F2{c2x + b.sh - b.cw, c2y - b.ch - b.sw} : F2{c2x - b.sh + b.cw, c2y - b.ch - b.sw};
  const F2 a2 = F2{2 * c1x - a0.x, 2 * c1y - a0.y}, a3 = F2{2 * c1x - a1.x, 2 * c1y - a1.y};
  const F2 b2 = F2{2 * c2x - b0.x, 2 * c2y - b0.y}, b3 = F2{2 * c2x - b1.x, 2 * c2y - b1.y};
  auto pick = [](F2 p0, F2 p1, F2 p2, F2 p3, int m) -> F2 {
    const bool odd = (m & 1) != 0, hi = (m & 2) != 0;
    const float ex = odd ? p1.x : p0.x, ey = odd ? p1.y : p0.y;
    const float ox = odd ? p3.x : p2.x, oy = odd ? p3.y : p2.y;
    return F2{hi ? ox : ex, hi ? oy : ey};
  };
#if defined(RSDET_AB_STAGE) && RSDET_AB_STAGE <= 1  // timing-only ablation builds (profiles/scripts/ab_build.sh)
  return a0.x + b3.y;
#endif
  // ---- edge l of box1 against edges 0..3 of box2
  const F2 P1 = pick(a0, a1, a2, a3, l), P1n = pick(a1, a2, a3, a0, l);
  const F2 v1 = f2sub(P1n, P1);
  F2 pe0{0.f, 0.f}, pe1{0.f, 0.f}, pe2{0.f, 0.f}, pe3{0.f, 0.f};
  auto solve = [&](F2 P2, F2 P2n, F2& pt) -> bool {
    const F2 v2 = f2sub(P2n, P2);
    float det = f2cross(v2, v1);
    if (!gt_1e14(fabsf(det))) return false;
    F2 d = f2sub(P2, P1);
    float c1 = f2cross(v2, d), c2 = f2cross(v1, d);
    // exact shortcut: skip the IEEE divisions when 0 <= t <= 1 is already decided with a margin
    // far above rounding (|c| > |det|(1+1e-6) => |t| > 1; opposite signs, |c| > 1e-20|det| => t < 0)
    float ad = fabsf(det), hi = ad * 1.000001f, lo = ad * 1e-20f;
    bool neg1 = (c1 < 0.f) != (det < 0.f), neg2 = (c2 < 0.f) != (det < 0.f);
    if (fabsf(c1) > hi || fabsf(c2) > hi || (neg1 && fabsf(c1) > lo) || (neg2 && fabsf(c2) > lo)) return false;
    float t1 = c1 / det, t2 = c2 / det;
    if (!(t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f)) return false;
    pt = F2{P1.x + v1.x * t1, P1.y + v1.y * t1};
    return true;
  };
  const bool h0 = solve(b0, b1, pe0), h1 = solve(b1, b2, pe1), h2 = solve(b2, b3, pe2), h3 = solve(b3, b0, pe3);
#if defined(RSDET_AB_STAGE) && RSDET_AB_STAGE == 2
  return (h0 ? pe0.x : 0.f) + (h1 ? pe1.x : 0.f) + (h2 ? pe2.y : 0.f) + (h3 ? pe3.y : 0.f);
#endif
  // ---- containment of corner l of each rectangle in the other
  auto inside = [](F2 P, F2 O0, F2 O1, F2 O3) -> bool {
    const F2 AB = f2sub(O1, O0), DA = f2sub(O0, O3);
    const float ABAB = f2dot(AB, AB), ADAD = f2dot(DA, DA);
    const F2 AP = f2sub(P, O0);
    const float pab = f2dot(AP, AB), pad = -f2dot(AP, DA);
    return pab >= 0 && pad >= 0 && pab <= ABAB && pad <= ADAD;
  };
  const F2 Pb = pick(b0, b1, b2, b3, l);
  const bool inA = inside(P1, b0, b1, b3), inB = inside(Pb, a0, a1, a3);
  // ---- compaction in the reference's enumeration order
  const unsigned below = (1u << l) - 1u;
  const unsigned n0 = (unsigned)(__ballot(h0) >> qsh) & 15u, n1 = (unsigned)(__ballot(h1) >> qsh) & 15u;
  const unsigned n2 = (unsigned)(__ballot(h2) >> qsh) & 15u, n3 = (unsigned)(__ballot(h3) >> qsh) & 15u;
  const unsigned mA = (unsigned)(__ballot(inA) >> qsh) & 15u, mB = (unsigned)(__ballot(inB) >> qsh) & 15u;
  const int E = __popc(n0) + __popc(n1) + __popc(n2) + __popc(n3);
  const int n = E + __popc(mA) + __popc(mB);
  int slot = __popc(n0 & below) + __popc(n1 & below) + __popc(n2 & below) + __popc(n3 & below);
  if (h0) qscr[slot++] = pe0;
  if (h1) qscr[slot++] = pe1;
  if (h2) qscr[slot++] = pe2;
  if (h3) qscr[slot++] = pe3;
  if (inA) qscr[E + __popc(mA & below)] = P1;
  if (inB) qscr[E + __popc(mA) + __popc(mB & below)] = Pb;
  lds_wave_order();
#if defined(RSDET_AB_STAGE) && RSDET_AB_STAGE == 3
  return (float)n;
#endif

  float inter = 0.f;
  if (n > 2) {
    bool done = false;
    if (n <= 8) {
      // all lanes: the 8 slots (stale beyond n but readable), start point, shifted list
      F2 p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = qscr[k];
      int t = 0;
      F2 best = p[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        bool lower = k < n && (p[k].y < best.y || (p[k].y == best.y && p[k].x < best.x));
        if (lower) {
          best = p[k];
          t = k;
        }
      }
      F2 q[8];
      float d[8];
      const F2 q0old = f2sub(p[0], best);
      q[0] = F2{best.x - best.x, best.y - best.y};
      d[0] = 0.f;
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        q[k] = f2sub(p[k], best);
        if (k == t) q[k] = q0old;
        d[k] = f2dot(q[k], q[k]);
      }
      if (gt_1e8(d[1])) {  // reference's k == 1 (:206-212); otherwise serial path
        F2* ring = qscr + 16;  // slots 16..23: sorted ring; 8..15: fan terms (both free when n <= 8)
        unsigned used = 0u;    // ranks handed out by this lane
        // lane l ranks elements l and l + 4 of q[1..n)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kk = l + 4 * half;
          const F2 qk = half == 0 ? pick(q[0], q[1], q[2], q[3], l) : pick(q[4], q[5], q[6], q[7], l);
          const float dk = f2dot(qk, qk);
          int rank = 0;
#pragma unroll
          for (int jx = 1; jx < 8; ++jx) {
            float c = f2cross(qk, q[jx]);
            bool tie = lt_1e6(fabsf(c));
            bool j_first = tie ? (d[jx] < dk) : (c < 0);  // less(q[jx], qk)
            bool k_first = tie ? (dk < d[jx]) : (c > 0);  // less(qk, q[jx])
            bool precede = jx < kk ? !k_first : j_first;  // stable: earlier index wins ties
            if (jx < n && jx != kk && precede) rank++;
          }
          if (kk == 0)
            ring[0] = q[0];
          else if (kk < n) {
            ring[1 + rank] = qk;
            used |= 1u << rank;
          }
        }
        // The tolerance predicate is not a strict weak order: near-collinear points can form cycles
        // (a < b < c < a).  A tournament is transitive iff its scores are all distinct, so unless the
        // ranks of the quad are exactly {0 .. n-2} the ring has a slot written twice and a stale one:
        // leave such pairs to the serial path, which replays the reference's insertion sort.
        used |= (unsigned)__builtin_amdgcn_mov_dpp((int)used, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
        used |= (unsigned)__builtin_amdgcn_mov_dpp((int)used, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
        const bool ranks_ok = used == (1u << (n - 1)) - 1u;
        lds_wave_order();
        // ring[l-2 .. l+5] covers the neighbourhoods of both ranks owned by this lane
        F2 w[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) w[m] = ring[min(max(l - 2 + m, 0), 7)];
        const F2 s0 = ring[0];
        // scan pop test for i = l and i = l + 4 (:225-229), as a check only
        bool pop = (l >= 2 && l < n && f2cross(f2sub(w[2], w[0]), f2sub(w[1], w[0])) >= 0) ||
                   (l + 4 < n && f2cross(f2sub(w[6], w[4]), f2sub(w[5], w[4])) >= 0);
        const unsigned popm = (unsigned)(__ballot(pop) >> qsh) & 15u;
        if (popm == 0u && ranks_ok) {
          float* terms = reinterpret_cast<float*>(qscr + 8);
          terms[l] = fabsf(f2cross(f2sub(w[2], s0), f2sub(w[3], s0)));
          terms[l + 4] = fabsf(f2cross(f2sub(w[6], s0), f2sub(w[7], s0)));
          lds_wave_order();
          float area = 0.f;
#pragma unroll
          for (int k = 1; k < 7; ++k) {
            float tk = terms[k];
            if (k + 1 < n) area += tk;  // i = 1 .. m-2, m = n (:246-249)
          }
          inter = area * 0.5f;
          done = true;
        }
      }
    }
#if defined(RSDET_AB_STAGE) && RSDET_AB_STAGE == 4
    done = true;
#endif
    if (!done) {
      if (l == 0) inter = hull_area_general(Scratch{qscr, 1}, n);
      inter = __shfl(inter, lane & 60);
    }
  }
  return inter / (a.area + b.area - inter);
}

}  // namespace rsdet
