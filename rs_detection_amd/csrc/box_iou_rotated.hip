// box_iou_rotated.hip -- pairwise rotated IoU for gfx950 (CDNA4), C-ABI entries
// rsdet_box_iou_rotated_f32 (+ grouped form used by the batched assigner).
//
// Replaces: jdet.ops.box_iou_rotated / box_iou_rotated_v1
//   /root/reference/python/jdet/ops/box_iou_rotated.py:502-509 (jt.code seam),
//   kernel :413-461, CPU loop :487-500;  _v1.py:507-524.
//
// Shape of the problem (S2ANet: K gts x 21 824 anchors, ~1.2 % of pairs overlap): a dense
// fp32 matrix must be written (HBM-store-bound) but the arithmetic lives in a sparse,
// badly balanced tail (a 16 x 256 tile holds anything from 0 to ~420 overlapping pairs).
// Three launches on one stream, no host round trip:
//   iou_prepare  fp64 sincos once per box -> 40-B "prepared box" (workspace); zeroes the
//                global work-queue counter.
//   iou_filter   tile = 16 rows x 256 columns per workgroup.  A thread owns a column, walks
//                the prepared rows in LDS, applies a bounding-circle test then a separating-
//                axis test and stores exact 0.0f for disjoint pairs (lanes = consecutive
//                columns: coalesced 256-B stores).  Survivors are appended to an LDS list with
//                wave ballot + popcount prefix and flushed to a GLOBAL work queue (64 shards) with
//                one returning atomic per workgroup.
//   iou_clip     a fixed grid drains the global queue, FOUR LANES PER PAIR (rsdet_geom.h):
//                perfect balance whatever the tile densities were, ~1.5 us latency per pair
//                instead of ~15 us for a one-thread clipper.
// If the queue overflows (more than its capacity of overlapping pairs) the overflowing
// workgroup clips its own survivors in place -- slower, still exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_geom.h"

namespace rsdet {

constexpr int IOU_NT = 256;  // columns per tile = threads per workgroup
constexpr int IOU_TI = 16;   // rows per tile
constexpr long long IOU_QUEUE_CAP = 4LL << 20;  // pairs (16 B each)
constexpr int IOU_SHARDS = 64;  // work-queue shards, one counter per 128-B line: a single returning
                                // atomic word saturates at ~88 ops/us (MI355X_MICROARCH.md "dequeue"),
                                // i.e. ~100 us for the 8 600 workgroups of a 1600 x 21 824 call (measured)
constexpr int CLIP_NT = 256;
constexpr int CLIP_BLOCKS = 2048;  // 256 CUs x 8 workgroups, quad-stride loop

struct WorkItem {
  int row;   // row of boxes1 / ious
  int col;   // column inside the row
  int p2;    // index into the prepared boxes2 array (col + group slab)
  int pad;
};

__global__ void iou_prepare_kernel(const float* __restrict__ boxes1, long long n1, int stride1,
                                   BoxPre* __restrict__ pre1, const float* __restrict__ boxes2,
                                   long long n2, int stride2, BoxPre* __restrict__ pre2,
                                   unsigned* __restrict__ counter) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < IOU_SHARDS) counter[i * 32] = 0u;
  if (i < n1)
    pre1[i] = prepare_box(boxes1 + i * stride1);
  else if (i - n1 < n2)
    pre2[i - n1] = prepare_box(boxes2 + (i - n1) * stride2);
}

template <int VERSION>
__global__ __launch_bounds__(IOU_NT) void iou_filter_kernel(
    const BoxPre* __restrict__ pre1, int n1, const BoxPre* __restrict__ pre2, int n2,
    const int* __restrict__ row_offsets, long long group_stride2, float* __restrict__ out,
    WorkItem* __restrict__ queue, unsigned* __restrict__ counter, unsigned capacity) {
  __shared__ BoxPre s_row[IOU_TI];
  __shared__ unsigned short s_list[IOU_TI * IOU_NT];
  __shared__ F2 s_pts[kQuadSlots * 16];  // overflow path only: one wave (16 quads) clips
  __shared__ int s_count;
  __shared__ unsigned s_base;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int col0 = blockIdx.x * IOU_NT;
  // plain form: row_offsets == nullptr, rows [0, n1).
  // grouped form (batched assigner): blockIdx.z = image g, its rows are
  // [row_offsets[g], row_offsets[g+1]) of boxes1/out and its columns are the g-th slab of
  // pre2 (group_stride2 boxes apart; 0: one column set shared by all).
  int row_begin = 0, row_end = n1;
  long long slab = 0;
  if (row_offsets) {
    row_begin = row_offsets[blockIdx.z];
    row_end = row_offsets[blockIdx.z + 1];
    slab = (long long)blockIdx.z * group_stride2;
  }
  const BoxPre* p2 = pre2 + slab;
  const int row0 = row_begin + blockIdx.y * IOU_TI;
  if (row0 >= row_end) return;
  const int nrows = min(IOU_TI, row_end - row0);

  if (tid == 0) s_count = 0;
  if (tid < nrows) s_row[tid] = pre1[row0 + tid];
  const int col = col0 + tid;
  const bool col_ok = col < n2;
  BoxPre mine;
  if (col_ok) mine = p2[col];
  __syncthreads();

  // ---- zero-fill + survivor list (bounding circles, then separating axes)
  for (int i = 0; i < nrows; ++i) {
    bool cand = false;
    if (col_ok) {
      const BoxPre r = s_row[i];
      cand = !surely_disjoint(r, mine);
      if (cand) cand = !sat_disjoint<VERSION>(r, mine);
      if (!cand) out[(long long)(row0 + i) * n2 + col] = 0.0f;
    }
    unsigned long long m = __ballot(cand);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(m));
      base = __shfl(base, 0);
      if (cand) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((i << 8) | tid);
    }
  }
  __syncthreads();

  // ---- flush the survivors to the global work queue (one returning atomic per workgroup)
  const int total = s_count;
  if (total == 0) return;
  const unsigned shard = (blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u) % IOU_SHARDS;
  if (tid == 0) s_base = atomicAdd(counter + shard * 32, (unsigned)total);
  __syncthreads();
  const unsigned base = s_base;  // capacity = entries per shard
  const int fit = base >= capacity ? 0 : (int)min((unsigned)total, capacity - base);
  queue += (size_t)shard * capacity;
  for (int q = tid; q < fit; q += IOU_NT) {
    unsigned e = s_list[q];
    WorkItem w;
    w.row = row0 + (int)(e >> 8);
    w.col = col0 + (int)(e & 255);
    w.p2 = (int)(slab + w.col);
    w.pad = 0;
    queue[base + q] = w;
  }
  // ---- queue overflow: clip the rest here (same routine, worse balance)
  if (tid >= 64) return;
  const int quad = tid >> 2;
  F2* qscr = s_pts + quad * kQuadSlots;
  for (int q = fit + quad; q < total; q += 16) {
    unsigned e = s_list[q];
    const int i = (int)(e >> 8), j = (int)(e & 255);
    const BoxPre bb = p2[col0 + j];
    float v = pair_iou_quad<VERSION>(s_row[i], bb, qscr, lane);
    if ((tid & 3) == 0) out[(long long)(row0 + i) * n2 + col0 + j] = v;
  }
}

template <int VERSION>
__global__ __launch_bounds__(CLIP_NT) void iou_clip_kernel(
    const BoxPre* __restrict__ pre1, const BoxPre* __restrict__ pre2, int n2,
    const WorkItem* __restrict__ queue, const unsigned* __restrict__ counter, unsigned capacity,
    float* __restrict__ out) {
  __shared__ F2 s_pts[kQuadSlots * (CLIP_NT / 4)];
  __shared__ unsigned s_end[IOU_SHARDS];  // inclusive prefix sums of the shard fill levels
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {
    static_assert(IOU_SHARDS == 64, "one wave scans the shard counters");
    unsigned c = min(counter[tid * 32], capacity);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned o = __shfl_up(c, off);
      if (tid >= off) c += o;
    }
    s_end[tid] = c;
  }
  __syncthreads();
  const unsigned total = s_end[IOU_SHARDS - 1];
  const unsigned quads = gridDim.x * (CLIP_NT / 4);
  F2* qscr = s_pts + (tid >> 2) * kQuadSlots;
  for (unsigned q = blockIdx.x * (CLIP_NT / 4) + (tid >> 2); q < total; q += quads) {
    int shard = 0;  // first shard whose inclusive end exceeds q
#pragma unroll
    for (int step = 32; step > 0; step >>= 1)
      if (s_end[shard + step - 1] <= q) shard += step;
    const unsigned first = shard ? s_end[shard - 1] : 0u;
    const WorkItem w = queue[(size_t)shard * capacity + (q - first)];  // same 16 B in the 4 lanes of the quad
    const BoxPre a = pre1[w.row], b = pre2[w.p2];
    float v = pair_iou_quad<VERSION>(a, b, qscr, lane);
    if ((tid & 3) == 0) out[(long long)w.row * n2 + w.col] = v;
  }
}

}  // namespace rsdet

using namespace rsdet;

static inline size_t pre_bytes(long long n) { return ((size_t)n * sizeof(BoxPre) + 255) & ~(size_t)255; }
// entries per shard: a workgroup appends at most IOU_TI*IOU_NT pairs, so tiny problems still fit
static inline long long queue_cap(long long n1, long long n2) {
  long long pairs = n1 * n2;
  long long total = pairs < IOU_QUEUE_CAP ? pairs : IOU_QUEUE_CAP;
  long long per = (total + IOU_SHARDS - 1) / IOU_SHARDS;
  long long tile = (long long)IOU_TI * IOU_NT;
  return per < tile ? (pairs < tile ? pairs : tile) : per;
}

extern "C" size_t rsdet_box_iou_rotated_ws_size(int n1, long long n2_total, int n2) {
  if (n1 <= 0 || n2_total <= 0 || n2 <= 0) return 0;
  return pre_bytes(n1) + pre_bytes(n2_total) + IOU_SHARDS * 128 + (size_t)queue_cap(n1, n2) * IOU_SHARDS * sizeof(WorkItem);
}

static int iou_launch(const float* boxes1, int n1, int stride1, const int* row_offsets, int n_groups,
                      int max_rows, const float* boxes2, int n2, int stride2, long long group_stride2,
                      int version, float* ious, void* ws, size_t ws_bytes, hipStream_t s) {
  const long long n2_total = (row_offsets && group_stride2 != 0) ? (long long)n_groups * n2 : n2;
  if (!ws || ((uintptr_t)ws & 15) || ws_bytes < rsdet_box_iou_rotated_ws_size(n1, n2_total, n2))
    return RSDET_EINVAL;
  char* w = (char*)ws;
  BoxPre* pre1 = (BoxPre*)w;
  BoxPre* pre2 = (BoxPre*)(w + pre_bytes(n1));
  unsigned* counter = (unsigned*)(w + pre_bytes(n1) + pre_bytes(n2_total));
  WorkItem* queue = (WorkItem*)(w + pre_bytes(n1) + pre_bytes(n2_total) + IOU_SHARDS * 128);
  const unsigned cap = (unsigned)queue_cap(n1, n2);
  hipLaunchKernelGGL(iou_prepare_kernel, dim3((unsigned)((n1 + n2_total + 63) / 64)), dim3(64), 0, s, boxes1,
                     (long long)n1, stride1, pre1, boxes2, n2_total, stride2, pre2, counter);
  dim3 grid((n2 + IOU_NT - 1) / IOU_NT, (max_rows + IOU_TI - 1) / IOU_TI, row_offsets ? n_groups : 1);
  const long long gs = (row_offsets && group_stride2 != 0) ? (long long)n2 : 0LL;
  // enough quads for the queue, at most the resident set of the chip
  long long need = ((long long)cap * IOU_SHARDS + CLIP_NT / 4 - 1) / (CLIP_NT / 4);
  const int clip_blocks = (int)(need < CLIP_BLOCKS ? need : CLIP_BLOCKS);
  if (version == 0) {
    hipLaunchKernelGGL(iou_filter_kernel<0>, grid, dim3(IOU_NT), 0, s, pre1, n1, pre2, n2, row_offsets, gs, ious,
                       queue, counter, cap);
    hipLaunchKernelGGL(iou_clip_kernel<0>, dim3(clip_blocks), dim3(CLIP_NT), 0, s, pre1, pre2, n2, queue, counter,
                       cap, ious);
  } else {
    hipLaunchKernelGGL(iou_filter_kernel<1>, grid, dim3(IOU_NT), 0, s, pre1, n1, pre2, n2, row_offsets, gs, ious,
                       queue, counter, cap);
    hipLaunchKernelGGL(iou_clip_kernel<1>, dim3(clip_blocks), dim3(CLIP_NT), 0, s, pre1, pre2, n2, queue, counter,
                       cap, ious);
  }
  return rsdet_launch_status();
}

extern "C" int rsdet_box_iou_rotated_f32(const float* boxes1, int n1, int stride1,
                                         const float* boxes2, int n2, int stride2, int version,
                                         float* ious, void* ws, size_t ws_bytes, void* stream) {
  if (n1 < 0 || n2 < 0 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (stride1 < 5 || stride2 < 5) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;  // box_iou_rotated.py:487-500: empty loops
  if (!boxes1 || !boxes2 || !ious) return RSDET_EINVAL;
  return iou_launch(boxes1, n1, stride1, nullptr, 1, n1, boxes2, n2, stride2, 0, version, ious, ws,
                    ws_bytes, (hipStream_t)stream);
}

extern "C" int rsdet_box_iou_rotated_grouped_f32(const float* boxes1, int n1, int stride1,
                                                 const int* row_offsets, int n_groups,
                                                 int max_rows_per_group, const float* boxes2,
                                                 int n2, int stride2, long long group_stride2,
                                                 int version, float* ious, void* ws, size_t ws_bytes,
                                                 void* stream) {
  if (n1 < 0 || n2 < 0 || n_groups < 0 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (stride1 < 5 || stride2 < 5) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0 || n_groups == 0 || max_rows_per_group <= 0) return RSDET_OK;
  if (!boxes1 || !boxes2 || !ious || !row_offsets) return RSDET_EINVAL;
  if (group_stride2 != 0 && group_stride2 != (long long)n2 * stride2) return RSDET_EINVAL;  // dense (G,n2,stride2)
  return iou_launch(boxes1, n1, stride1, row_offsets, n_groups, max_rows_per_group, boxes2, n2, stride2,
                    group_stride2, version, ious, ws, ws_bytes, (hipStream_t)stream);
}
