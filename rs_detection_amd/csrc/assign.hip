// assign.hip -- MaxIoUAssigner.assign_wrt_overlaps for a whole batch, on device.
//
// Replaces: /root/reference/python/jdet/models/boxes/assigner.py:111-170 --
// in particular the per-gt Python loop :151-160 (K launches of length A plus a
// jt.sync_all() every 100 gts) and the host syncs of :164-166.
//
// Two HBM-bound passes over the (n1, A) overlaps matrix:
//   row pass    one workgroup per gt row: max / first-argmax over A (float4 loads,
//               wave shuffle + LDS reduction)  -> ws
//   column pass one thread per (group, anchor): walks the group's rows (coalesced
//               across lanes), keeps max / first-argmax, applies the neg / pos
//               thresholds and the low-quality rule (last row whose IoU EQUALS
//               its row maximum wins, as the ascending reference loop does),
//               then gathers the label.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"

namespace rsdet {

#ifndef RSDET_ASG_NT
#define RSDET_ASG_NT 512
#endif
constexpr int ASG_NT = RSDET_ASG_NT;

__global__ __launch_bounds__(ASG_NT) void assign_row_kernel(const float* __restrict__ ov, int A,
                                                            float* __restrict__ row_max,
                                                            int* __restrict__ row_arg) {
  const int row = blockIdx.x;
  const float* p = ov + (long long)row * A;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  constexpr int U = 4;  // 4 independent loads in flight per thread, consumed in ascending j
  for (int j0 = threadIdx.x; j0 < A; j0 += U * ASG_NT) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * ASG_NT;
      v[u] = j < A ? p[j] : -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (v[u] > best) {  // ascending j per thread => first index kept on ties
        best = v[u];
        arg = j0 + u * ASG_NT;
      }
  }
  // wave reduce (value desc, index asc)
  for (int off = 32; off > 0; off >>= 1) {
    float ob = __shfl_down(best, off);
    int oa = __shfl_down(arg, off);
    if (ob > best || (ob == best && oa < arg)) {
      best = ob;
      arg = oa;
    }
  }
  __shared__ float s_b[ASG_NT / 64];
  __shared__ int s_a[ASG_NT / 64];
  if ((threadIdx.x & 63) == 0) {
    s_b[threadIdx.x >> 6] = best;
    s_a[threadIdx.x >> 6] = arg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ASG_NT / 64; ++w)
      if (s_b[w] > best || (s_b[w] == best && s_a[w] < arg)) {
        best = s_b[w];
        arg = s_a[w];
      }
    row_max[row] = best;
    row_arg[row] = arg;
  }
}

// Column pass: workgroup = 64 anchors x ASG_SLICES row slices (wave s walks rows r0+s, r0+s+SLICES, ...),
// then the slices are folded in LDS.  One thread per anchor walking all K rows was a latency chain
// (~140 dependent row steps, 344 workgroups for the whole chip: 48 us); sliced, the chain is K/8 long and
// there are 8x the waves (step shape: 63 -> 23.5 us for both passes = 4.2 TB/s of matrix reads).  The fold reproduces the serial semantics exactly: max with `>` (first row wins
// ties, NaN rows never win, a NaN FIRST row sticks -- assigner.py:133 argmax on a matrix seeded by row 0),
// low-quality match = LAST row whose IoU equals its row maximum (ascending reference loop, :151-160).
#ifndef RSDET_ASG_SLICES
#define RSDET_ASG_SLICES 8
#endif
constexpr int ASG_SLICES = RSDET_ASG_SLICES;

__global__ __launch_bounds__(64 * ASG_SLICES) void assign_col_kernel(
    const float* __restrict__ ov, int A, const int* __restrict__ row_offsets,
    const float* __restrict__ row_max, const int* __restrict__ row_arg, float pos_thr,
    float neg_lo, float neg_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
    const int* __restrict__ gt_labels, int labels_filled, int* __restrict__ gt_inds,
    float* __restrict__ max_ov, int* __restrict__ labels) {
  __shared__ float s_best[ASG_SLICES][64];
  __shared__ int s_arg[ASG_SLICES][64];
  __shared__ int s_lowq[ASG_SLICES][64];
  const int g = blockIdx.y;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int r0 = row_offsets[g], r1 = row_offsets[g + 1];
  const long long o = (long long)g * A + j;
  if (r1 <= r0) {
    if (slice == 0 && j < A) {
      gt_inds[o] = 0;
      max_ov[o] = 0.f;
      if (labels) labels[o] = labels_filled;
    }
    return;
  }
  const int jc = min(j, A - 1);  // lanes past the end read a valid column and are dropped at the store
  float best = -INFINITY;
  int arg = 0x7fffffff;
  int lowq = -1;
  constexpr int U = 4;  // independent loads in flight per thread
  for (int rb = r0 + slice; rb < r1; rb += U * ASG_SLICES) {
    float v[U], rm[U];
    int ra[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(rb + u * ASG_SLICES, r1 - 1);
      v[u] = ov[(long long)r * A + jc];
      rm[u] = row_max[r];  // wave-uniform -> scalar loads
      ra[u] = row_arg[r];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * ASG_SLICES;
      if (r < r1) {
        if (r == r0 || v[u] > best) {  // r == r0: the very first row seeds the maximum (NaN semantics of `>`)
          best = v[u];
          arg = r - r0;
        }
        if (match_low_quality && rm[u] >= min_pos_iou)
          if (gt_max_assign_all ? (v[u] == rm[u]) : (ra[u] == jc)) lowq = r - r0;
      }
    }
  }
  s_best[slice][lane] = best;
  s_arg[slice][lane] = arg;
  s_lowq[slice][lane] = lowq;
  __syncthreads();
  if (slice != 0 || j >= A) return;
  // slice 0 holds row r0: start from it, fold the others with `>` / first index
#pragma unroll
  for (int s = 1; s < ASG_SLICES; ++s) {
    const float b = s_best[s][lane];
    const int a = s_arg[s][lane];
    if (b > best || (b == best && a < arg)) {
      best = b;
      arg = a;
    }
    lowq = max(lowq, s_lowq[s][lane]);
  }
  int gi = -1;
  if (best >= neg_lo && best < neg_hi) gi = 0;   // assigner.py:138-145
  if (best >= pos_thr) gi = arg + 1;             // :147-148
  if (lowq >= 0) gi = lowq + 1;                  // :151-158
  gt_inds[o] = gi;
  max_ov[o] = best;
  if (labels) labels[o] = gi > 0 ? gt_labels[r0 + gi - 1] : labels_filled;  // :162-166
}

// ---- horizontal-box IoU / IoF matrix (the assigner's default calculator) -----------------------------------------------
// /root/reference/python/jdet/models/boxes/iou_calculator.py:164-257 (bbox_overlaps, not aligned) as ONE pass: the tensor
// form builds lt / rb / wh / overlap / union as (K, A, 2) and (K, A) intermediates -- ten launches over up to 1 GB each
// for the Oriented RPN's 400 gts x 400 000 anchors.  Same operations in the same order, fp32, no contraction
// (-ffp-contract=off): area = (x2 - x1) * (y2 - y1); overlap = max(min(x2) - max(x1), 0) * max(min(y2) - max(y1), 0);
// union = (area1 + area2) - overlap (IoU) or area1 (IoF); result = overlap / max(union, eps) -- bit-identical.
// One thread = one column box over a strip of 32 rows (row boxes and areas in LDS); a wave writes 256 B of a matrix row.
constexpr int HB_NT = 256, HB_ROWS = 32;
__global__ __launch_bounds__(HB_NT) void bbox_overlaps_kernel(const float* __restrict__ b1, int n1, int stride1,
                                                              const float* __restrict__ b2, int n2, int stride2, int iof,
                                                              float eps, float* __restrict__ out) {
  __shared__ float s_box[HB_ROWS][5];
  const int col = blockIdx.x * HB_NT + threadIdx.x, row0 = blockIdx.y * HB_ROWS;
  const int nr = min(HB_ROWS, n1 - row0);
  if (threadIdx.x < nr) {
    const float* p = b1 + (long long)(row0 + threadIdx.x) * stride1;
    const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    s_box[threadIdx.x][0] = x1, s_box[threadIdx.x][1] = y1, s_box[threadIdx.x][2] = x2, s_box[threadIdx.x][3] = y2;
    s_box[threadIdx.x][4] = (x2 - x1) * (y2 - y1);
  }
  __syncthreads();
  if (col >= n2) return;
  const float* q = b2 + (long long)col * stride2;
  const float cx1 = q[0], cy1 = q[1], cx2 = q[2], cy2 = q[3];
  const float a2 = (cx2 - cx1) * (cy2 - cy1);
  float* o = out + (long long)row0 * n2 + col;
  for (int r = 0; r < nr; ++r) {
    const float w = fmaxf(fminf(s_box[r][2], cx2) - fmaxf(s_box[r][0], cx1), 0.f);
    const float h = fmaxf(fminf(s_box[r][3], cy2) - fmaxf(s_box[r][1], cy1), 0.f);
    const float ov = w * h;
    const float uni = iof ? s_box[r][4] : (s_box[r][4] + a2) - ov;
    o[(long long)r * n2] = ov / fmaxf(uni, eps);
  }
}

}  // namespace rsdet

using namespace rsdet;

extern "C" size_t rsdet_assign_ws_size(int n1) {
  return n1 > 0 ? (((size_t)n1 * 4 + 255) & ~(size_t)255) * 2 : 0;
}

extern "C" int rsdet_assign_wrt_overlaps_f32(const float* overlaps, int n1, int A,
                                             const int* row_offsets, int n_groups,
                                             int max_rows_per_group, float pos_iou_thr,
                                             float neg_iou_lo, float neg_iou_hi, float min_pos_iou,
                                             int match_low_quality, int gt_max_assign_all,
                                             const int* gt_labels, int labels_filled, int* gt_inds,
                                             float* max_overlaps, int* labels, void* ws,
                                             size_t ws_bytes, void* stream) {
  (void)max_rows_per_group;
  if (n1 < 0 || A < 0 || n_groups < 0) return RSDET_EINVAL;
  if (A == 0 || n_groups == 0) return RSDET_OK;
  if (!row_offsets || !gt_inds || !max_overlaps) return RSDET_EINVAL;
  if (n1 > 0 && (!overlaps || !ws || ws_bytes < rsdet_assign_ws_size(n1))) return RSDET_EINVAL;
  if (labels && !gt_labels && n1 > 0) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* row_max = (float*)ws;
  int* row_arg = (int*)((char*)ws + rsdet_assign_ws_size(n1) / 2);
  if (n1 > 0)
    hipLaunchKernelGGL(assign_row_kernel, dim3(n1), dim3(ASG_NT), 0, s, overlaps, A, row_max,
                       row_arg);
  hipLaunchKernelGGL(assign_col_kernel, dim3((A + 63) / 64, n_groups), dim3(64 * ASG_SLICES), 0,
                     s, overlaps, A, row_offsets, row_max, row_arg, pos_iou_thr, neg_iou_lo,
                     neg_iou_hi, min_pos_iou, match_low_quality, gt_max_assign_all, gt_labels,
                     labels_filled, gt_inds, max_overlaps, labels);
  return rsdet_launch_status();
}

// boxes1 (n1 rows of stride1 >= 4 floats: x1 y1 x2 y2 ...), boxes2 likewise -> out (n1, n2) row-major.
extern "C" int rsdet_bbox_overlaps_f32(const float* boxes1, int n1, int stride1, const float* boxes2, int n2, int stride2,
                                       int iof, float eps, float* out, void* stream) {
  if (n1 < 0 || n2 < 0 || stride1 < 4 || stride2 < 4) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;
  if (!boxes1 || !boxes2 || !out) return RSDET_EINVAL;
  const dim3 grid((n2 + HB_NT - 1) / HB_NT, (n1 + HB_ROWS - 1) / HB_ROWS);
  if (grid.y > 65535u) return RSDET_EINVAL;
  hipLaunchKernelGGL(bbox_overlaps_kernel, grid, dim3(HB_NT), 0, (hipStream_t)stream, boxes1, n1, stride1, boxes2, n2,
                     stride2, iof, eps, out);
  return rsdet_launch_status();
}
