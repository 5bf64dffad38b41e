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
