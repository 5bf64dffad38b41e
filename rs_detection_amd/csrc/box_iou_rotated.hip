// box_iou_rotated.hip -- pairwise rotated IoU for gfx950 (CDNA4), C-ABI entries
// rsdet_box_iou_rotated_f32 (+ grouped form used by the batched assigner).
//
// Replaces: jdet.ops.box_iou_rotated / box_iou_rotated_v1
//   /root/reference/python/jdet/ops/box_iou_rotated.py:502-509 (jt.code seam),
//   kernel :413-461, CPU loop :487-500;  _v1.py:507-524.
//
// Shape of the problem (S2ANet: K gts x 21 824 anchors, ~1.2 % of pairs overlap): a dense
// fp32 matrix must be written (HBM-store-bound, ~8 us for 48 MB) but the arithmetic lives in a
// sparse tail (~2 800 lane-instructions per overlapping pair, ALU-bound) and everything in
// between is latency.  Three launches on one stream, no host round trip:
//   iou_prepare  fp64 sincos once per box -> 40-B "prepared box" (workspace) + the bounding box
//                of every 64 consecutive column circles; zeroes the queue counters.
//   iou_filter   tile = 16 rows x 256 columns per workgroup.  Issues the zero fill of the whole
//                tile first (lanes = consecutive columns: 256-B stores) and never waits for it;
//                underneath, whole (row, 64 columns) strips are culled against the column
//                bounding box, live strips get the bounding-circle test, the circle survivors
//                are compacted in LDS and get the separating-axis test on dense lanes.  What
//                is left goes to a GLOBAL work queue (64 shards, one returning atomic per
//                workgroup).
//   iou_clip     a fixed grid drains the global queue, FOUR LANES PER PAIR (rsdet_geom.h):
//                perfect balance whatever the tile densities were, ~1.5 us latency per pair
//                instead of ~15 us for a one-thread clipper; overwrites the zeros of its pairs.
// If the queue overflows (more than its capacity of overlapping pairs) the overflowing
// workgroup clips its own survivors in place -- slower, still exact.
// Tried and measured slower: one wave per 32 x 64 strip set without LDS, and the zero
// fill moved into a store-only wave of iou_clip behind a live-strip bitmap.
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

// One wave per 64 boxes.  Workgroups [0, nb1) prepare boxes1; the rest prepare boxes2, aligned to
// the 64-column words of the IoU matrix (cw words per column slab), and also reduce the padded
// bounding box of their 64 bounding circles: iou_filter rejects whole (row, 64 columns) strips
// against it before any per-pair work.
__global__ __launch_bounds__(64) void iou_prepare_kernel(
    const float* __restrict__ boxes1, int n1, int stride1, BoxPre* __restrict__ pre1, int nb1,
    const float* __restrict__ boxes2, int n2, int stride2, BoxPre* __restrict__ pre2, int cw,
    float4* __restrict__ colbox, unsigned* __restrict__ counter) {
  const int lane = threadIdx.x;
  if (blockIdx.x == 0) counter[lane * 32] = 0u;
  if ((int)blockIdx.x < nb1) {
    const int i = blockIdx.x * 64 + lane;
    if (i < n1) pre1[i] = prepare_box(boxes1 + (long long)i * stride1);
    return;
  }
  const int word = blockIdx.x - nb1;  // slab * cw + k
  const int slab = word / cw, k = word - slab * cw;
  const int col = k * 64 + lane;
  float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
  if (col < n2) {
    const long long j = (long long)slab * n2 + col;
    const BoxPre p = prepare_box(boxes2 + j * stride2);
    pre2[j] = p;
    const float pad = 1.001f * p.rad + 1e-5f * (fabsf(p.cx) + fabsf(p.cy));
    const bool finite = fabsf(p.cx) < INFINITY && fabsf(p.cy) < INFINITY && pad < INFINITY;  // false for NaN too
    x0 = finite ? p.cx - pad : -INFINITY;
    y0 = finite ? p.cy - pad : -INFINITY;
    x1 = finite ? p.cx + pad : INFINITY;
    y1 = finite ? p.cy + pad : INFINITY;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    x0 = fminf(x0, __shfl_xor(x0, off));
    y0 = fminf(y0, __shfl_xor(y0, off));
    x1 = fmaxf(x1, __shfl_xor(x1, off));
    y1 = fmaxf(y1, __shfl_xor(y1, off));
  }
  if (lane == 0) colbox[word] = make_float4(x0, y0, x1, y1);
}

#ifdef RSDET_FILTER_TRACE  // debug builds only (profiles/scripts/trace_filter.py): per-workgroup stage timestamps, 100 MHz
__device__ unsigned long long* g_trace;
#define TRACE(k)                                                                                     \
  do {                                                                                               \
    if (threadIdx.x == 0 && g_trace)                                                                 \
      g_trace[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); \
  } while (0)
#else
#define TRACE(k)
#endif

template <int VERSION>
__global__ __launch_bounds__(IOU_NT) void iou_filter_kernel(
    const BoxPre* __restrict__ pre1, int n1, const BoxPre* __restrict__ pre2, int n2,
    const float4* __restrict__ colbox, int cw, const int* __restrict__ row_offsets,
    long long group_stride2, float* __restrict__ out, WorkItem* __restrict__ queue,
    unsigned* __restrict__ counter, unsigned capacity) {
  __shared__ BoxPre s_row[IOU_TI];
  __shared__ BoxPre s_col[IOU_NT];
  // pairs whose bounding circles touch; compacted IN PLACE to the pairs the separating-axis test
  // could not reject (the survivors)
  __shared__ unsigned short s_list[IOU_TI * IOU_NT];
  __shared__ F2 s_pts[kQuadSlots * 16];  // overflow path only: one wave (16 quads) clips
  __shared__ int s_c1, s_c2;
  __shared__ unsigned s_base;

  TRACE(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int col0 = blockIdx.x * IOU_NT;
  // plain form: row_offsets == nullptr, rows [0, n1).
  // grouped form (batched assigner): blockIdx.z = image g, its rows are
  // [row_offsets[g], row_offsets[g+1]) of boxes1/out and its columns are the g-th slab of
  // pre2 (group_stride2 boxes apart; 0: one column set shared by all).
  int row_begin = 0, row_end = n1;
  long long slab = 0;
  int slab_word = 0;
  if (row_offsets) {
    row_begin = row_offsets[blockIdx.z];
    row_end = row_offsets[blockIdx.z + 1];
    slab = (long long)blockIdx.z * group_stride2;
    if (group_stride2 != 0) slab_word = blockIdx.z * cw;
  }
  const BoxPre* p2 = pre2 + slab;
  const int row0 = row_begin + blockIdx.y * IOU_TI;
  if (row0 >= row_end) return;
  const int nrows = min(IOU_TI, row_end - row0);
  const int col = col0 + tid;
  const bool col_ok = col < n2;

  // ---- stage the tile's boxes in LDS, then start the zero fill (every pair of the tile; survivors
  // are rewritten by iou_clip in the next launch).  The stores are never waited for: all barriers
  // below order LDS only (lds_barrier), so the store stream -- the HBM-bound part of the whole
  // call -- drains underneath the latency-bound phases.  vmcnt retires in order, though, and the
  // returning queue atomic would sit behind the stores: wave 0 owns that atomic and issues its
  // share of the fill after it.
  const BoxPre mine = p2[min(col, n2 - 1)];
  const int kw = (col0 >> 6) + (tid >> 6);
  float4 cb = colbox[slab_word + min(kw, cw - 1)];
  if (kw >= cw) cb = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);  // empty strip
  if (tid < nrows) s_row[tid] = pre1[row0 + tid];
  if (col_ok) s_col[tid] = mine;
  if (tid == 0) {
    s_c1 = 0;
    s_c2 = 0;
  }
  auto zero_fill = [&]() {
    if (!col_ok) return;
    float* o = out + (long long)row0 * n2 + col;
    if (nrows == IOU_TI) {
#pragma unroll
      for (int i = 0; i < IOU_TI; ++i) o[(long long)i * n2] = 0.0f;
    } else {
      for (int i = 0; i < nrows; ++i) o[(long long)i * n2] = 0.0f;
    }
  };
  // pin the strip box in registers here: a wait for it placed after the fill would be a vmcnt(0)
  asm volatile("" : "+v"(cb.x), "+v"(cb.y), "+v"(cb.z), "+v"(cb.w));
  TRACE(1);
  const bool fill_late = tid < 64;
  if (!fill_late) zero_fill();
  lds_barrier();
  TRACE(2);

  // ---- strip culling: lane i < nrows tests row i's circle against the bounding box of this wave's
  // 64 column circles (iou_prepare).  Both sides carry a 1e-3 relative pad, so a culled strip
  // satisfies surely_disjoint() for each of its pairs; NaN/Inf boxes are never culled.
  static_assert(IOU_TI <= 32, "row mask is 32 bits wide");
  bool lv = false;
  if (lane < nrows) {
    const float rx = s_row[lane].cx, ry = s_row[lane].cy;
    const float dx = fmaxf(fmaxf(cb.x - rx, rx - cb.z), 0.f), dy = fmaxf(fmaxf(cb.y - ry, ry - cb.w), 0.f);
    const float thr = 1.001f * s_row[lane].rad + 1e-5f * (fabsf(rx) + fabsf(ry));
    lv = !(dx * dx + dy * dy > thr * thr);
  }
  unsigned live = (unsigned)__ballot(lv);

  // ---- pass A: bounding circles of the live strips (a handful of instructions per pair, no
  // divergence).  The separating-axis test used to sit here: ~3.5 % of the pairs reach it, i.e.
  // nearly every wave paid for it on nearly every row.
  while (live) {
    const int i = __builtin_ctz(live);
    live &= live - 1u;
    bool cand = false;
    if (col_ok) {
      const float dx = s_row[i].cx - mine.cx, dy = s_row[i].cy - mine.cy;
      const float r = s_row[i].rad + mine.rad;
      cand = !(dx * dx + dy * dy > r * r * 1.0001f);  // == !surely_disjoint(s_row[i], mine)
    }
    unsigned long long m = __ballot(cand);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_c1, __popcll(m));
      base = __shfl(base, 0);
      if (cand) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((i << 8) | tid);
    }
  }
  lds_barrier();
  TRACE(3);

  // ---- pass B: separating axes on the compacted list (dense lanes).  In-place compaction is safe:
  // by the barrier every entry below q0 + IOU_NT has been read, and at most that many were kept.
  const int n_cand = s_c1;
  for (int q0 = 0; q0 < n_cand; q0 += IOU_NT) {
    const int q = q0 + tid;
    unsigned e = 0;
    if (q < n_cand) e = s_list[q];
    lds_barrier();
    const bool keep = q < n_cand && !sat_disjoint<VERSION>(s_row[e >> 8], s_col[e & 255]);
    unsigned long long m = __ballot(keep);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_c2, __popcll(m));
      base = __shfl(base, 0);
      if (keep) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)e;
    }
  }
  lds_barrier();
  TRACE(4);

  // ---- flush the survivors to the global work queue (one returning atomic per workgroup)
  const int total = s_c2;
  if (total == 0) {
    if (fill_late) zero_fill();
    return;
  }
  const unsigned shard = (blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u) % IOU_SHARDS;
  if (tid == 0) s_base = atomicAdd(counter + shard * 32, (unsigned)total);
  lds_barrier();
  TRACE(5);
  if (fill_late) zero_fill();
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
  // ---- queue overflow: clip the rest here (same routine, worse balance).  __syncthreads() drains
  // this workgroup's zero stores (s_waitcnt vmcnt(0) before s_barrier), so the values written
  // below land after them.
  TRACE(6);
  if (fit == total) return;
  __syncthreads();
  if (tid >= 64) return;
  const int quad = tid >> 2;
  F2* qscr = s_pts + quad * kQuadSlots;
  for (int q = fit + quad; q < total; q += 16) {
    unsigned e = s_list[q];
    const int i = (int)(e >> 8), j = (int)(e & 255);
    float v = pair_iou_quad<VERSION>(s_row[i], s_col[j], qscr, lane);
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
#ifdef RSDET_POISON_LDS  // debug builds only: stale scratch must never reach a result
  for (int k = tid; k < kQuadSlots * (CLIP_NT / 4); k += CLIP_NT) s_pts[k] = F2{RSDET_POISON_LDS, RSDET_POISON_LDS};
  __syncthreads();
#endif
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

#ifdef RSDET_FILTER_TRACE
extern "C" void rsdet_debug_set_trace(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)); }
#endif

static inline size_t pre_bytes(long long n) { return ((size_t)n * sizeof(BoxPre) + 255) & ~(size_t)255; }
// entries per shard: a workgroup appends at most IOU_TI*IOU_NT pairs, so tiny problems still fit
static inline long long queue_cap(long long n1, long long n2) {
  long long pairs = n1 * n2;
  long long total = pairs < IOU_QUEUE_CAP ? pairs : IOU_QUEUE_CAP;
  long long per = (total + IOU_SHARDS - 1) / IOU_SHARDS;
  long long tile = (long long)IOU_TI * IOU_NT;
  return per < tile ? (pairs < tile ? pairs : tile) : per;
}
static inline size_t colbox_bytes(long long n2_total, long long n2) {  // one float4 per (slab, 64 columns)
  return ((size_t)(n2_total / n2) * (size_t)((n2 + 63) / 64) * 16 + 255) & ~(size_t)255;
}

extern "C" size_t rsdet_box_iou_rotated_ws_size(int n1, long long n2_total, int n2) {
  if (n1 <= 0 || n2_total <= 0 || n2 <= 0) return 0;
  return pre_bytes(n1) + pre_bytes(n2_total) + IOU_SHARDS * 128 + colbox_bytes(n2_total, n2) +
         (size_t)queue_cap(n1, n2) * IOU_SHARDS * sizeof(WorkItem);
}

static int iou_launch(const float* boxes1, int n1, int stride1, const int* row_offsets, int n_groups,
                      int max_rows, const float* boxes2, int n2, int stride2, long long group_stride2,
                      int version, float* ious, void* ws, size_t ws_bytes, hipStream_t s) {
  const long long n2_total = (row_offsets && group_stride2 != 0) ? (long long)n_groups * n2 : n2;
  if (!ws || ((uintptr_t)ws & 15) || ws_bytes < rsdet_box_iou_rotated_ws_size(n1, n2_total, n2))
    return RSDET_EINVAL;
  char* w = (char*)ws;
  size_t off = 0;
  BoxPre* pre1 = (BoxPre*)(w + off);
  off += pre_bytes(n1);
  BoxPre* pre2 = (BoxPre*)(w + off);
  off += pre_bytes(n2_total);
  unsigned* counter = (unsigned*)(w + off);
  off += IOU_SHARDS * 128;
  float4* colbox = (float4*)(w + off);
  off += colbox_bytes(n2_total, n2);
  WorkItem* queue = (WorkItem*)(w + off);
  const int cw = (n2 + 63) / 64;
  const unsigned cap = (unsigned)queue_cap(n1, n2);
  const int nb1 = (n1 + 63) / 64, nb2 = (int)(n2_total / n2) * cw;
  hipLaunchKernelGGL(iou_prepare_kernel, dim3(nb1 + nb2), dim3(64), 0, s, boxes1, n1, stride1, pre1, nb1, boxes2,
                     n2, stride2, pre2, cw, colbox, counter);
  dim3 grid((n2 + IOU_NT - 1) / IOU_NT, (max_rows + IOU_TI - 1) / IOU_TI, row_offsets ? n_groups : 1);
  const long long gs = (row_offsets && group_stride2 != 0) ? (long long)n2 : 0LL;
  // enough quads for the queue, at most the resident set of the chip
  long long need = ((long long)cap * IOU_SHARDS + CLIP_NT / 4 - 1) / (CLIP_NT / 4);
  const int clip_blocks = (int)(need < CLIP_BLOCKS ? need : CLIP_BLOCKS);
  if (version == 0) {
    hipLaunchKernelGGL(iou_filter_kernel<0>, grid, dim3(IOU_NT), 0, s, pre1, n1, pre2, n2, colbox, cw, row_offsets, gs,
                       ious, queue, counter, cap);
    hipLaunchKernelGGL(iou_clip_kernel<0>, dim3(clip_blocks), dim3(CLIP_NT), 0, s, pre1, pre2, n2, queue, counter,
                       cap, ious);
  } else {
    hipLaunchKernelGGL(iou_filter_kernel<1>, grid, dim3(IOU_NT), 0, s, pre1, n1, pre2, n2, colbox, cw, row_offsets, gs,
                       ious, queue, counter, cap);
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
