// poly_iou.hip -- quadrilateral IoU and the merge-time polygon NMS, fp64.
//
// Replaces (evaluation / tile-merge side of the path, CPU + shapely + a 16-process pool in the reference):
//   /root/reference/python/jdet/ops/nms_poly.py:247-252        iou_poly (shapely Polygon.intersection().area)
//   /root/reference/python/jdet/data/devkits/result_merge.py:66-126   py_cpu_nms_poly_fast
//   /root/reference/python/jdet/data/devkits/voc_eval.py:263-304      the per-detection overlap loop of voc_eval_dota
// shapely (GEOS) is a third-party dependency that is absent here and on the GPU box: parity with it is
// UNPINNED.  The intersection is computed by Sutherland-Hodgman clipping of one quadrilateral against the
// other in double precision (exact for convex clippers; detections are rectangles, so one side of every pair
// is convex and is chosen as the clipper), area by the shoelace formula.
//   rsdet_poly_iou_f64          dense (n1, n2) IoU matrix, one thread per pair
//   rsdet_nms_poly_sorted_f64   py_cpu_nms_poly_fast on score-sorted quads: horizontal-hull gate exactly as
//                               written there (+1 in the areas, none in the overlap), polygon IoU where the gate
//                               passes, suppression on IoU > thr; 64x64 mask tiles in the sparse entry format
//                               of nms_rotated.hip + its device sweep.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"

namespace rsdet {

struct D2 {
  double x, y;
};

__device__ __forceinline__ double quad_signed_area(const D2* p) {
  double a = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const D2 u = p[i], v = p[(i + 1) & 3];
    a += u.x * v.y - v.x * u.y;
  }
  return 0.5 * a;
}

__device__ __forceinline__ bool quad_is_convex(const D2* p) {  // p counter-clockwise
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const D2 a = p[i], b = p[(i + 1) & 3], c = p[(i + 2) & 3];
    ok = ok && ((b.x - a.x) * (c.y - b.y) - (b.y - a.y) * (c.x - b.x) >= 0.0);
  }
  return ok;
}

__device__ __forceinline__ void load_quad_ccw(const double* __restrict__ q, D2* p, double& area) {
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = D2{q[2 * i], q[2 * i + 1]};
  area = quad_signed_area(p);
  if (area < 0.0) {  // clockwise: reverse
    const D2 t = p[1];
    p[1] = p[3];
    p[3] = t;
    area = -area;
  }
}

// area of (subject ∩ clipper), clipper convex and counter-clockwise
__device__ double clip_area(const D2* subject, const D2* clipper) {
  D2 cur[10], nxt[10];
  int n = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) cur[i] = subject[i];
  for (int e = 0; e < 4 && n > 0; ++e) {
    const D2 a = clipper[e], b = clipper[(e + 1) & 3];
    const double ex = b.x - a.x, ey = b.y - a.y;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const D2 p = cur[i], q = cur[i + 1 == n ? 0 : i + 1];
      const double sp = ex * (p.y - a.y) - ey * (p.x - a.x);  // >= 0: on the inner side
      const double sq = ex * (q.y - a.y) - ey * (q.x - a.x);
      if (sp >= 0.0) nxt[m++] = p;
      if ((sp >= 0.0) != (sq >= 0.0)) {
        const double t = sp / (sp - sq);
        nxt[m++] = D2{p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
      }
    }
    n = m;
    for (int i = 0; i < n; ++i) cur[i] = nxt[i];
  }
  if (n < 3) return 0.0;
  double a2 = 0.0;
  for (int i = 0; i < n; ++i) {
    const D2 u = cur[i], v = cur[i + 1 == n ? 0 : i + 1];
    a2 += u.x * v.y - v.x * u.y;
  }
  return fabs(0.5 * a2);
}

__device__ __forceinline__ double quad_iou(const double* __restrict__ q1, const double* __restrict__ q2) {
  D2 p1[4], p2[4];
  double a1, a2;
  load_quad_ccw(q1, p1, a1);
  load_quad_ccw(q2, p2, a2);
  const double inter = quad_is_convex(p2) ? clip_area(p1, p2) : clip_area(p2, p1);
  return inter / fmax(a1 + a2 - inter, 0.01);  // nms_poly.py:251
}

__global__ __launch_bounds__(256) void poly_iou_kernel(const double* __restrict__ polys1, int n1,
                                                       const double* __restrict__ polys2, int n2,
                                                       double* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)n1 * n2) return;
  const int i = (int)(idx / n2), j = (int)(idx - (long long)i * n2);
  out[idx] = quad_iou(polys1 + (long long)i * 8, polys2 + (long long)j * 8);
}

struct NmsEntry {  // same record as nms_rotated.hip
  unsigned long long bits;
  int cblock;
  int row;
};

// Tail shared by the mask kernels of this file: off-diagonal tiles append their non-zero rows to the row block's
// entry list (one returning atomic per tile), the diagonal tile is stored transposed for the sweep's fixpoint.
__device__ __forceinline__ void poly_emit_tile(unsigned long long bits, int rb, int cbk, int tid, int col_blocks,
                                               NmsEntry* __restrict__ entries, unsigned* __restrict__ blk_cnt,
                                               unsigned long long* __restrict__ diag_t,
                                               unsigned long long* s_rows) {
  if (rb != cbk) {
    const unsigned long long nz = __ballot(bits != 0ull);
    if (nz == 0ull) return;
    unsigned base = 0u;
    if (tid == 0) base = atomicAdd(blk_cnt + rb, (unsigned)__popcll(nz));
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (bits != 0ull) {
      NmsEntry e;
      e.bits = bits;
      e.cblock = cbk;
      e.row = tid;
      entries[(size_t)rb * 64 * col_blocks + base + __popcll(nz & ((1ull << tid) - 1ull))] = e;
    }
    return;
  }
  s_rows[tid] = bits;
  __syncthreads();
  unsigned long long col = 0ull;
#pragma unroll 8
  for (int i = 0; i < 64; ++i) col |= ((s_rows[i] >> tid) & 1ull) << i;
  diag_t[rb * 64 + tid] = col;
}

// one wave per 64x64 tile; lane = row of the tile
__global__ __launch_bounds__(64) void nms_poly_mask_kernel(const double* __restrict__ polys, int n, double thr,
                                                           int col_blocks, NmsEntry* __restrict__ entries,
                                                           unsigned* __restrict__ blk_cnt,
                                                           unsigned long long* __restrict__ diag_t) {
  const int rb = blockIdx.y, cbk = blockIdx.x;
  if (cbk < rb) return;
  __shared__ double s_col[64 * 8];
  __shared__ double s_box[64 * 5];  // x1, y1, x2, y2, hull area (+1)
  __shared__ unsigned long long s_rows[64];
  const int tid = threadIdx.x;
  const int cols = min(64, n - cbk * 64), rows = min(64, n - rb * 64);
  if (tid < cols) {
    const double* q = polys + (long long)(cbk * 64 + tid) * 8;
    double x1 = q[0], x2 = q[0], y1 = q[1], y2 = q[1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_col[tid * 8 + 2 * k] = q[2 * k];
      s_col[tid * 8 + 2 * k + 1] = q[2 * k + 1];
      x1 = fmin(x1, q[2 * k]);
      x2 = fmax(x2, q[2 * k]);
      y1 = fmin(y1, q[2 * k + 1]);
      y2 = fmax(y2, q[2 * k + 1]);
    }
    s_box[tid * 5 + 0] = x1;
    s_box[tid * 5 + 1] = y1;
    s_box[tid * 5 + 2] = x2;
    s_box[tid * 5 + 3] = y2;
    s_box[tid * 5 + 4] = (x2 - x1 + 1) * (y2 - y1 + 1);  // result_merge.py:73
  }
  __syncthreads();
  unsigned long long bits = 0ull;
  if (tid < rows) {
    const double* q = polys + (long long)(rb * 64 + tid) * 8;
    double x1 = q[0], x2 = q[0], y1 = q[1], y2 = q[1];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      x1 = fmin(x1, q[2 * k]);
      x2 = fmax(x2, q[2 * k]);
      y1 = fmin(y1, q[2 * k + 1]);
      y2 = fmax(y2, q[2 * k + 1]);
    }
    const double area = (x2 - x1 + 1) * (y2 - y1 + 1);
    const int start = (rb == cbk) ? tid + 1 : 0;
    for (int j = start; j < cols; ++j) {
      const double w = fmax(0.0, fmin(x2, s_box[j * 5 + 2]) - fmax(x1, s_box[j * 5 + 0]));  // :94-96: no +1 here
      const double h = fmax(0.0, fmin(y2, s_box[j * 5 + 3]) - fmax(y1, s_box[j * 5 + 1]));
      const double hbb_inter = w * h;
      double ovr = hbb_inter / (area + s_box[j * 5 + 4] - hbb_inter);
      if (ovr > 0) ovr = quad_iou(q, s_col + j * 8);  // :99-103 (box1 = the kept, higher-scoring box)
      if (!(ovr <= thr)) bits |= 1ull << j;           // :117 keeps `ovr <= thresh`; NaN suppresses, as there
    }
  }
  poly_emit_tile(bits, rb, cbk, tid, col_blocks, entries, blk_cnt, diag_t, s_rows);
}

// ---- in-model polygon NMS, fp32 (ops/nms_poly.py:135-245, caller roi_heads/gliding_head.py:181) -------------------
// The reference's quadrilateral IoU (nms_poly.py:17-132) sums SIGNED intersections of origin-anchored triangles
// (o, a_i, a_i+1) x (o, b_j, b_j+1), each cut by three half-planes, all in float.  With image coordinates (plus the
// per-class offset of multiclass_poly_nms :213-216) the triangles are ~1e6 px^2 and the cancellation noise of that
// sum is part of the reference's keep decisions, so the arithmetic is restated operation by operation (same order,
// no FMA: -ffp-contract=off) and EVERY pair is evaluated -- a horizontal-hull gate would change results.
// One wave per 64x64 tile, lane = row.  The two vertex rings of the cutter (10 + 10 float2 per lane) are indexed
// by data-dependent counters: they live in LDS in [slot][lane] layout (conflict-free, no scratch memory).
struct F2 {
  float x, y;
};

__device__ __forceinline__ int psig(float d) { return ((double)d > 1e-8) - ((double)d < -1e-8); }  // :17-19
__device__ __forceinline__ bool peq(F2 a, F2 b) { return psig(a.x - b.x) == 0 && psig(a.y - b.y) == 0; }
__device__ __forceinline__ float pcross(F2 o, F2 a, F2 b) {  // :39-41
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

constexpr int PN_SLOTS = 10;  // maxn (:14)

// ring accessors: slot k of this lane
#define PN_AT(ring, k) ring[(k) * 64 + lane]

// :42-49 -- closes the ring (slot n <- slot 0) and returns the signed area
__device__ __forceinline__ float pn_area(F2* ring, int n, int lane) {
  PN_AT(ring, n) = PN_AT(ring, 0);
  float res = 0.f;
  for (int i = 0; i < n; ++i) {
    const F2 a = PN_AT(ring, i), b = PN_AT(ring, i + 1);
    res += a.x * b.y - a.y * b.x;
  }
  return res * 0.5f;  // res / 2.0 evaluated in double and rounded back: the same value
}

// :61-76 -- keep the part of ring p (n vertices) left of a->b; pp is the staging ring
__device__ __forceinline__ void pn_cut(F2* p, int& n, F2 a, F2 b, F2* pp, int lane) {
  int m = 0;
  PN_AT(p, n) = PN_AT(p, 0);
  for (int i = 0; i < n; ++i) {
    const F2 u = PN_AT(p, i), v = PN_AT(p, i + 1);
    const float su = pcross(a, b, u), sv = pcross(a, b, v);
    if (psig(su) > 0) {
      PN_AT(pp, m) = u;
      m += m < PN_SLOTS - 1;  // the reference's rings hold maxn = 10 points and are never checked; a triangle cut
    }                         // by three lines stays below that, the clamp only keeps a pathological input in bounds
    if (psig(su) != psig(sv)) {
      // lineCross (:50-59): s1 = su, s2 = sv.  The two "no crossing" exits leave the staging slot as it is.
      if (!(psig(su) == 0 && psig(sv) == 0) && psig(sv - su) != 0) {
        F2 x;
        x.x = (u.x * sv - v.x * su) / (sv - su);
        x.y = (u.y * sv - v.y * su) / (sv - su);
        PN_AT(pp, m) = x;
      }
      m += m < PN_SLOTS - 1;
    }
  }
  n = 0;
  for (int i = 0; i < m; ++i)
    if (!i || !peq(PN_AT(pp, i), PN_AT(pp, i - 1))) {
      PN_AT(p, n) = PN_AT(pp, i);
      ++n;
    }
  while (n > 1 && peq(PN_AT(p, n - 1), PN_AT(p, 0))) --n;
}

// :80-98 -- signed intersection area of triangles (o,a,b) and (o,c,d), o = origin
__device__ __forceinline__ float pn_tri(F2 a, F2 b, F2 c, F2 d, F2* p, F2* pp, int lane) {
  const F2 o{0.f, 0.f};
  const int s1 = psig(pcross(o, a, b)), s2 = psig(pcross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.f;
  if (s1 == -1) { const F2 t = a; a = b; b = t; }
  if (s2 == -1) { const F2 t = c; c = d; d = t; }
  PN_AT(p, 0) = o;
  PN_AT(p, 1) = a;
  PN_AT(p, 2) = b;
  int n = 3;
  pn_cut(p, n, o, c, pp, lane);
  pn_cut(p, n, c, d, pp, lane);
  pn_cut(p, n, d, o, pp, lane);
  float res = fabsf(pn_area(p, n, lane));
  if (s1 * s2 == -1) res = -res;
  return res;
}

// :100-132 -- devPolyIoU of two quadrilaterals given as 4 vertices each (q1 = the row = higher score)
__device__ float pn_quad_iou(const float* __restrict__ q1, const float* __restrict__ q2, F2* p, F2* pp, int lane) {
  F2 a[5], b[5];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = F2{q1[2 * i], q1[2 * i + 1]};
    b[i] = F2{q2[2 * i], q2[2 * i + 1]};
  }
  // area() of a 4-ring in registers (same summation order as pn_area)
  auto area4 = [](const F2* r) {
    float res = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const F2 u = r[i], v = r[(i + 1) & 3];
      res += u.x * v.y - u.y * v.x;
    }
    return res * 0.5f;
  };
  if (area4(a) < 0) {  // point_reverse (:31-37): 0<->3, 1<->2
    F2 t = a[0]; a[0] = a[3]; a[3] = t;
    t = a[1]; a[1] = a[2]; a[2] = t;
  }
  if (area4(b) < 0) {
    F2 t = b[0]; b[0] = b[3]; b[3] = t;
    t = b[1]; b[1] = b[2]; b[2] = t;
  }
  a[4] = a[0];
  b[4] = b[0];
  // staging ring starts from zeros for every quad pair (the reference's is uninitialised stack memory; it is only
  // ever read after a failed lineCross, which needs cross products within 1e-8 of each other)
  for (int k = 0; k < PN_SLOTS; ++k) PN_AT(pp, k) = F2{0.f, 0.f};
  float inter = 0.f;
#pragma unroll 1
  for (int i = 0; i < 4; ++i)
#pragma unroll 1
    for (int j = 0; j < 4; ++j) inter += pn_tri(a[i], a[i + 1], b[j], b[j + 1], p, pp, lane);
  const float uni = fabsf(area4(a)) + fabsf(area4(b)) - inter;
  return uni == 0 ? (inter + 1) / (uni + 1) : inter / uni;  // :125-129
}

// PN_WAVES waves per 64x64 tile: wave w takes the columns j = w (mod PN_WAVES), every wave with its own pair of rings
// (a lone wave per tile walks 64 columns x 16 triangle pairs serially and 2000 boxes give only ~500 tiles: ~2 waves
// per CU, 4.9 ms; 8 waves per tile: see DESIGN.md).  The waves OR their bits together in LDS.
constexpr int PN_WAVES = 8;

__global__ __launch_bounds__(64 * PN_WAVES) void nms_poly_f32_mask_kernel(const float* __restrict__ polys, int stride,
                                                                         int n, float thr, int col_blocks,
                                                                         NmsEntry* __restrict__ entries,
                                                                         unsigned* __restrict__ blk_cnt,
                                                                         unsigned long long* __restrict__ diag_t) {
  const int rb = blockIdx.y, cbk = blockIdx.x;
  if (cbk < rb) return;
  __shared__ float s_col[64 * 8];
  __shared__ F2 s_ring[PN_WAVES * 2 * PN_SLOTS * 64];  // per wave: ring p, then staging ring pp
  __shared__ unsigned long long s_rows[64];
  const int tid = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cols = min(64, n - cbk * 64), rows = min(64, n - rb * 64);
  if (wave == 0) {
    s_rows[tid] = 0ull;
    if (tid < cols) {
      const float* q = polys + (long long)(cbk * 64 + tid) * stride;
#pragma unroll
      for (int k = 0; k < 8; ++k) s_col[tid * 8 + k] = q[k];
    }
  }
  __syncthreads();
  unsigned long long bits = 0ull;
  if (tid < rows) {
    float q[8];
    const float* g = polys + (long long)(rb * 64 + tid) * stride;
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = g[k];
    const int start = (rb == cbk) ? tid + 1 : 0;
    for (int j = wave; j < cols; j += PN_WAVES)
      if (j >= start && pn_quad_iou(q, s_col + j * 8, s_ring + wave * 2 * PN_SLOTS * 64,
                                   s_ring + (wave * 2 + 1) * PN_SLOTS * 64, tid) > thr) bits |= 1ull << j;  // :179
  }
  if (bits) atomicOr(&s_rows[tid], bits);
  __syncthreads();  // the last barrier of the kernel: every wave reaches it, s_rows is complete afterwards
  if (wave != 0) return;
  bits = s_rows[tid];
  if (rb != cbk) {  // off-diagonal tile: append the non-zero rows to the row block's entry list
    const unsigned long long nz = __ballot(bits != 0ull);
    if (nz == 0ull) return;
    unsigned base = 0u;
    if (tid == 0) base = atomicAdd(blk_cnt + rb, (unsigned)__popcll(nz));
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (bits != 0ull) {
      NmsEntry e;
      e.bits = bits;
      e.cblock = cbk;
      e.row = tid;
      entries[(size_t)rb * 64 * col_blocks + base + __popcll(nz & ((1ull << tid) - 1ull))] = e;
    }
    return;
  }
  unsigned long long col = 0ull;  // diagonal tile: stored transposed for the sweep's fixpoint
#pragma unroll 8
  for (int i = 0; i < 64; ++i) col |= ((s_rows[i] >> tid) & 1ull) << i;
  diag_t[rb * 64 + tid] = col;
}

// dense (n1, n2) matrix of the same IoU: one wave per 64 pairs (used by tests and by callers that want the values)
__global__ __launch_bounds__(64) void poly_iou_f32_kernel(const float* __restrict__ polys1, int n1,
                                                          const float* __restrict__ polys2, int n2,
                                                          float* __restrict__ out) {
  __shared__ F2 s_p[PN_SLOTS * 64], s_pp[PN_SLOTS * 64];
  const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
  if (idx >= (long long)n1 * n2) return;
  const int i = (int)(idx / n2), j = (int)(idx - (long long)i * n2);
  float a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = polys1[(long long)i * 8 + k];
    b[k] = polys2[(long long)j * 8 + k];
  }
  out[idx] = pn_quad_iou(a, b, s_p, s_pp, threadIdx.x);
}

__global__ void poly_iota_kernel(int* p, int n, unsigned* blk_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (n + 63) / 64) blk_cnt[i] = 0u;
  if (i < n) p[i] = i;
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_poly_iou_f64(const double* polys1, int n1, const double* polys2, int n2, double* ious,
                                  void* stream) {
  if (n1 < 0 || n2 < 0) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;
  if (!polys1 || !polys2 || !ious) return RSDET_EINVAL;
  const long long total = (long long)n1 * n2;
  hipLaunchKernelGGL(poly_iou_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, polys1,
                     n1, polys2, n2, ious);
  return rsdet_launch_status();
}

// ws layout: ident (n ints) | diag_t | blk_cnt | entries  -- sized by rsdet_nms_hbb_ws_size(n) (same sweep)
extern "C" size_t rsdet_nms_hbb_ws_size(int n);

extern "C" int rsdet_nms_poly_sorted_f64(const double* polys_sorted, int n, double thr, uint8_t* keep_sorted, void* ws,
                                         size_t ws_bytes, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!polys_sorted || !keep_sorted || !ws || ws_bytes < rsdet_nms_hbb_ws_size(n) || ((uintptr_t)ws & 15))
    return RSDET_EINVAL;
  const int cb = (n + 63) / 64;
  if (cb > 8192) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t ident_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
  const size_t diag_bytes = (size_t)cb * 64 * sizeof(unsigned long long);
  const size_t cnt_bytes = ((size_t)cb * 4 + 255) & ~(size_t)255;
  int* ident = (int*)ws;
  char* w = (char*)ws + ident_bytes;
  unsigned long long* diag_t = (unsigned long long*)w;
  unsigned* blk_cnt = (unsigned*)(w + diag_bytes);
  NmsEntry* entries = (NmsEntry*)(w + diag_bytes + cnt_bytes);
  hipLaunchKernelGGL(poly_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ident, n, blk_cnt);
  hipLaunchKernelGGL(nms_poly_mask_kernel, dim3(cb, cb), dim3(64), 0, s, polys_sorted, n, thr, cb, entries, blk_cnt,
                     diag_t);
  rsdet_launch_nms_sweep(entries, blk_cnt, diag_t, n, cb, ident, keep_sorted, s);
  return rsdet_launch_status();
}

// ---- f4: poly_nms (fp32, in-model) -------------------------------------------------------------------------
extern "C" int rsdet_poly_iou_f32(const float* polys1, int n1, const float* polys2, int n2, float* ious,
                                  void* stream) {
  if (n1 < 0 || n2 < 0) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;
  if (!polys1 || !polys2 || !ious) return RSDET_EINVAL;
  const long long total = (long long)n1 * n2;
  hipLaunchKernelGGL(poly_iou_f32_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     polys1, n1, polys2, n2, ious);
  return rsdet_launch_status();
}

extern "C" int rsdet_poly_nms_sorted_f32(const float* dets_sorted, int n, float thr, uint8_t* keep_sorted, void* ws,
                                         size_t ws_bytes, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!dets_sorted || !keep_sorted || !ws || ws_bytes < rsdet_nms_hbb_ws_size(n) || ((uintptr_t)ws & 15))
    return RSDET_EINVAL;
  const int cb = (n + 63) / 64;
  if (cb > 8192) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t ident_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
  const size_t diag_bytes = (size_t)cb * 64 * sizeof(unsigned long long);
  const size_t cnt_bytes = ((size_t)cb * 4 + 255) & ~(size_t)255;
  int* ident = (int*)ws;
  char* w = (char*)ws + ident_bytes;
  unsigned long long* diag_t = (unsigned long long*)w;
  unsigned* blk_cnt = (unsigned*)(w + diag_bytes);
  NmsEntry* entries = (NmsEntry*)(w + diag_bytes + cnt_bytes);
  hipLaunchKernelGGL(poly_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ident, n, blk_cnt);
  hipLaunchKernelGGL(nms_poly_f32_mask_kernel, dim3(cb, cb), dim3(64 * PN_WAVES), 0, s, dets_sorted, 9, n, thr, cb, entries,
                     blk_cnt, diag_t);
  rsdet_launch_nms_sweep(entries, blk_cnt, diag_t, n, cb, ident, keep_sorted, s);
  return rsdet_launch_status();
}
