// box_iou_rotated.hip -- pairwise rotated IoU for gfx950 (CDNA4), C-ABI entry
// rsdet_box_iou_rotated_f32 (+ grouped form used by the batched assigner).
//
// Replaces: jdet.ops.box_iou_rotated / box_iou_rotated_v1
//   /root/reference/python/jdet/ops/box_iou_rotated.py:502-509 (jt.code seam),
//   kernel :413-461, CPU loop :487-500;  _v1.py:507-524.
//
// Design (HBM-store-bound with a sparse ALU-heavy tail):
//   tile = TI rows (boxes1) x NT columns (boxes2), one workgroup of NT threads.
//   phase A  every thread owns one column: prepares its box (fp64 sincos once),
//            walks the TI prepared rows held in LDS, applies a conservative
//            bounding-circle test and stores exact 0.0f for disjoint pairs --
//            lanes write consecutive columns => fully coalesced 256-B stores.
//            Surviving (row, col) pairs are appended to an LDS work queue with a
//            wave ballot + popcount prefix (one LDS atomic per wave per row).
//   phase B  the queue is drained densely: consecutive lanes take consecutive
//            candidate pairs, so the expensive polygon clipping runs at full
//            lane occupancy instead of diverging inside phase A.
//   The <=24 clip points per pair sit in LDS slot-major (rsdet_geom.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_geom.h"

namespace rsdet {

constexpr int IOU_NT = 256;  // columns per tile = threads per workgroup
constexpr int IOU_TI = 16;   // rows per tile

template <int VERSION>
__global__ __launch_bounds__(IOU_NT) void box_iou_rotated_kernel(
    const float* __restrict__ boxes1, int n1, int stride1, const float* __restrict__ boxes2,
    int n2, int stride2, const int* __restrict__ row_offsets, long long group_stride2,
    float* __restrict__ out) {
  __shared__ F2 s_pts[24 * IOU_NT];
  __shared__ BoxPre s_row[IOU_TI];
  __shared__ BoxPre s_col[IOU_NT];
  __shared__ unsigned short s_queue[IOU_TI * IOU_NT];
  __shared__ int s_count;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int col0 = blockIdx.x * IOU_NT;
  // plain form: row_offsets == nullptr, rows [0, n1).
  // grouped form (batched assigner): blockIdx.z = image g, its rows are
  // [row_offsets[g], row_offsets[g+1]) of boxes1/out and its columns are
  // boxes2 + g*group_stride2 (group_stride2 = 0: one column set shared by all).
  int row_begin = 0, row_end = n1;
  const float* b2 = boxes2;
  if (row_offsets) {
    row_begin = row_offsets[blockIdx.z];
    row_end = row_offsets[blockIdx.z + 1];
    b2 += (long long)blockIdx.z * group_stride2;
  }
  const int row0 = row_begin + blockIdx.y * IOU_TI;
  if (row0 >= row_end) return;
  const int nrows = min(IOU_TI, row_end - row0);

  if (tid == 0) s_count = 0;
  if (tid < nrows) s_row[tid] = prepare_box(boxes1 + (long long)(row0 + tid) * stride1);
  const int col = col0 + tid;
  const bool col_ok = col < n2;
  BoxPre mine;
  if (col_ok) {
    mine = prepare_box(b2 + (long long)col * stride2);
    s_col[tid] = mine;
  }
  __syncthreads();

  // ---- phase A: zero-fill + candidate queue
  for (int i = 0; i < nrows; ++i) {
    bool cand = false;
    if (col_ok) {
      cand = !surely_disjoint(s_row[i], mine);
      if (!cand) out[(long long)(row0 + i) * n2 + col] = 0.0f;
    }
    unsigned long long m = __ballot(cand);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(m));
      base = __shfl(base, 0);
      if (cand) {
        int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        s_queue[pos] = (unsigned short)((i << 8) | tid);
      }
    }
  }
  __syncthreads();

  // ---- phase B: dense drain
  const int total = s_count;
  Scratch sc{s_pts + tid, IOU_NT};
  for (int q = tid; q < total; q += IOU_NT) {
    unsigned e = s_queue[q];
    int i = e >> 8, j = e & 255;
    float v = pair_iou<VERSION>(s_row[i], s_col[j], sc);
    out[(long long)(row0 + i) * n2 + col0 + j] = v;
  }
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_box_iou_rotated_f32(const float* boxes1, int n1, int stride1,
                                         const float* boxes2, int n2, int stride2, int version,
                                         float* ious, void* stream) {
  if (n1 < 0 || n2 < 0 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (stride1 < 5 || stride2 < 5) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;  // box_iou_rotated.py:487-500: empty loops
  if (!boxes1 || !boxes2 || !ious) return RSDET_EINVAL;
  dim3 grid((n2 + IOU_NT - 1) / IOU_NT, (n1 + IOU_TI - 1) / IOU_TI);
  hipStream_t s = (hipStream_t)stream;
  if (version == 0)
    hipLaunchKernelGGL(box_iou_rotated_kernel<0>, grid, dim3(IOU_NT), 0, s, boxes1, n1, stride1,
                       boxes2, n2, stride2, (const int*)nullptr, 0LL, ious);
  else
    hipLaunchKernelGGL(box_iou_rotated_kernel<1>, grid, dim3(IOU_NT), 0, s, boxes1, n1, stride1,
                       boxes2, n2, stride2, (const int*)nullptr, 0LL, ious);
  return rsdet_launch_status();
}

extern "C" int rsdet_box_iou_rotated_grouped_f32(const float* boxes1, int n1, int stride1,
                                                 const int* row_offsets, int n_groups,
                                                 int max_rows_per_group, const float* boxes2,
                                                 int n2, int stride2, long long group_stride2,
                                                 int version, float* ious, void* stream) {
  if (n1 < 0 || n2 < 0 || n_groups < 0 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (stride1 < 5 || stride2 < 5) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0 || n_groups == 0 || max_rows_per_group <= 0) return RSDET_OK;
  if (!boxes1 || !boxes2 || !ious || !row_offsets) return RSDET_EINVAL;
  dim3 grid((n2 + IOU_NT - 1) / IOU_NT, (max_rows_per_group + IOU_TI - 1) / IOU_TI, n_groups);
  hipStream_t s = (hipStream_t)stream;
  if (version == 0)
    hipLaunchKernelGGL(box_iou_rotated_kernel<0>, grid, dim3(IOU_NT), 0, s, boxes1, n1, stride1,
                       boxes2, n2, stride2, row_offsets, group_stride2, ious);
  else
    hipLaunchKernelGGL(box_iou_rotated_kernel<1>, grid, dim3(IOU_NT), 0, s, boxes1, n1, stride1,
                       boxes2, n2, stride2, row_offsets, group_stride2, ious);
  return rsdet_launch_status();
}
