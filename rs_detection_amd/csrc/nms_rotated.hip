// nms_rotated.hip -- greedy rotated / class-aware NMS for gfx950, fully on device.
//
// Replaces: jdet.ops.nms_rotated.{nms_rotated_cpu, nms_rotated_cuda}
//   /root/reference/python/jdet/ops/nms_rotated.py:495-512; CUDA mask kernel :353-411;
//   host bit-mask sweep on managed memory after cudaDeviceSynchronize :450-493;
//   CPU greedy loop :414-449 (the parity target: `ovr >= thr`, :444).
//
// Three launches on one stream, no host round trip:
//   1. nms_prepare : gather dets by `order`, hoist fp64 sincos -> BoxPre (+label)
//   2. nms_mask    : upper-triangular 64x64 tiles.  Cheap pass (label gate +
//                    bounding circles) fills an LDS queue via wave ballot /
//                    popcount prefix; the queue is drained densely through the
//                    exact clipper; hits set bits with LDS 64-bit atomicOr; one
//                    u64 word per (row, column block) goes to HBM.
//   3. nms_sweep   : one workgroup walks the 64-box blocks in score order.  Wave 0
//                    resolves the diagonal tile in registers (v_readlane chain),
//                    then all waves OR the kept rows' words into the LDS `removed`
//                    bitmap with coalesced reads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_geom.h"

namespace rsdet {

struct NmsBox {
  BoxPre p;
  float label;
  float pad;
};  // 48 B

constexpr int NMS_NT = 256;

__global__ void nms_prepare_kernel(const float* __restrict__ dets, int n, int box_len,
                                   const int* __restrict__ order, NmsBox* __restrict__ sorted) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float* b = dets + (long long)order[p] * box_len;
  NmsBox o;
  o.p = prepare_box(b);
  o.label = box_len == 6 ? b[5] : 0.f;
  o.pad = 0.f;
  sorted[p] = o;
}

template <bool GE>
__global__ __launch_bounds__(NMS_NT) void nms_mask_kernel(const NmsBox* __restrict__ sorted, int n,
                                                          float thr, int col_blocks,
                                                          unsigned long long* __restrict__ mask) {
  const int rb = blockIdx.y, cbk = blockIdx.x;
  if (cbk < rb) return;  // lower triangle never read by the sweep
  __shared__ F2 s_pts[kQuadSlots * (NMS_NT / 4)];
  __shared__ NmsBox s_row[64];
  __shared__ NmsBox s_col[64];
  __shared__ unsigned long long s_mask[64];
  __shared__ unsigned short s_queue[64 * 64];
  __shared__ int s_count;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = min(64, n - rb * 64), cols = min(64, n - cbk * 64);
  if (tid < 64) {
    if (tid < rows) s_row[tid] = sorted[rb * 64 + tid];
    s_mask[tid] = 0ull;
  } else if (tid < 128) {
    int c = tid - 64;
    if (c < cols) s_col[c] = sorted[cbk * 64 + c];
  }
  if (tid == 0) s_count = 0;
  __syncthreads();

  // cheap pass: wave w covers rows w, w+4, ...; lane = column
  for (int i = wave; i < rows; i += 4) {
    bool cand = false;
    if (lane < cols) {
      bool later = (cbk > rb) || (lane > i);
      cand = later && s_row[i].label == s_col[lane].label &&
             !surely_disjoint(s_row[i].p, s_col[lane].p) && !sat_disjoint<0>(s_row[i].p, s_col[lane].p);
    }
    unsigned long long m = __ballot(cand);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(m));
      base = __shfl(base, 0);
      if (cand) s_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((i << 6) | lane);
    }
  }
  __syncthreads();

  const int total = s_count;
  const int quad = tid >> 2;
  F2* qscr = s_pts + quad * kQuadSlots;
  for (int q = quad; q < total; q += NMS_NT / 4) {  // four lanes per pair (rsdet_geom.h)
    unsigned e = s_queue[q];
    int i = e >> 6, j = e & 63;
    float v = pair_iou_quad<0>(s_row[i].p, s_col[j].p, qscr, lane);  // box1 = earlier (kept) box, :443
    bool hit = GE ? (v >= thr) : (v > thr);
    if (hit && (tid & 3) == 0) atomicOr(&s_mask[i], 1ull << j);
  }
  __syncthreads();
  if (tid < rows) mask[(long long)(rb * 64 + tid) * col_blocks + cbk] = s_mask[tid];
}

// One workgroup; `removed` bitmap lives in LDS (n <= 64*NMS_MAX_BLOCKS boxes).
constexpr int NMS_MAX_BLOCKS = 8192;  // 524288 boxes, 64 KB of LDS

__global__ __launch_bounds__(NMS_NT) void nms_sweep_kernel(
    const unsigned long long* __restrict__ mask, int n, int col_blocks,
    const int* __restrict__ order, uint8_t* __restrict__ keep) {
  extern __shared__ unsigned long long s_removed[];  // col_blocks words + 1 (kept word)
  unsigned long long& s_kept = s_removed[col_blocks];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int w = tid; w < col_blocks; w += NMS_NT) s_removed[w] = 0ull;
  __syncthreads();

  for (int bk = 0; bk < col_blocks; ++bk) {
    const int rows = min(64, n - bk * 64);
    if (tid < 64) {
      unsigned long long diag = 0ull;
      if (lane < rows) diag = mask[(long long)(bk * 64 + lane) * col_blocks + bk];
      unsigned long long cur = s_removed[bk];
      unsigned long long kept = 0ull;
      unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
#pragma unroll
      for (int r = 0; r < 64; ++r) {
        // wave-uniform chain: row r survives iff no earlier kept row removed it
        unsigned long long row =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, r) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dlo, r);  // readlane returns int
        bool alive = r < rows && !((cur >> r) & 1ull);
        if (alive) {
          kept |= 1ull << r;
          cur |= row;
        }
      }
      if (lane < rows) keep[order[bk * 64 + lane]] = (uint8_t)((kept >> lane) & 1ull);
      if (lane == 0) s_kept = kept;
    }
    __syncthreads();
    unsigned long long kept = s_kept;
    // OR the kept rows into removed[bk+1 ..): thread owns words w = bk+1+tid, +NT, ...
    for (int w = bk + 1 + tid; w < col_blocks; w += NMS_NT) {
      unsigned long long acc = s_removed[w];
      unsigned long long k = kept;
      while (k) {
        int r = __builtin_ctzll(k);
        k &= k - 1;
        acc |= mask[(long long)(bk * 64 + r) * col_blocks + w];
      }
      s_removed[w] = acc;
    }
    __syncthreads();
  }
}

}  // namespace rsdet

using namespace rsdet;

static inline size_t nms_sorted_bytes(int n) { return ((size_t)n * sizeof(NmsBox) + 255) & ~(size_t)255; }

extern "C" size_t rsdet_nms_rotated_ws_size(int n) {
  if (n <= 0) return 0;
  size_t cb = ((size_t)n + 63) / 64;
  return nms_sorted_bytes(n) + (size_t)n * cb * sizeof(unsigned long long);
}

extern "C" int rsdet_nms_rotated_f32(const float* dets, int n, int box_len, const int* order,
                                     float thr, int ge, uint8_t* keep, void* ws, size_t ws_bytes,
                                     void* stream) {
  if (n < 0 || (box_len != 5 && box_len != 6)) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!dets || !order || !keep || !ws) return RSDET_EINVAL;
  if (ws_bytes < rsdet_nms_rotated_ws_size(n) || ((uintptr_t)ws & 15)) return RSDET_EINVAL;
  const int cb = (n + 63) / 64;
  if (cb > NMS_MAX_BLOCKS) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  NmsBox* sorted = (NmsBox*)ws;
  unsigned long long* mask = (unsigned long long*)((char*)ws + nms_sorted_bytes(n));
  hipLaunchKernelGGL(nms_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dets, n, box_len,
                     order, sorted);
  if (ge)
    hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(cb, cb), dim3(NMS_NT), 0, s, sorted, n, thr, cb,
                       mask);
  else
    hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(cb, cb), dim3(NMS_NT), 0, s, sorted, n, thr,
                       cb, mask);
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(NMS_NT), (size_t)(cb + 1) * 8, s, mask, n, cb, order,
                     keep);
  return rsdet_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Horizontal-box NMS (Oriented-RCNN proposal stage).  Replaces Jittor's built-in `jt.nms`
// (third-party, call sites /root/reference/python/jdet/models/roi_heads/oriented_rpn_head.py:219,
// ops/nms.py:9,44): boxes (x1,y1,x2,y2) already gathered in descending-score order, IoU with the
// legacy "+1" pixel convention when plus_one != 0, suppression on IoU > thr.  Same mask + device
// sweep structure as the rotated NMS above; keep[] is indexed by SORTED position.
namespace rsdet {

__global__ __launch_bounds__(64) void nms_hbb_mask_kernel(const float* __restrict__ boxes, int n, float thr,
                                                          float one, int col_blocks,
                                                          unsigned long long* __restrict__ mask) {
  const int rb = blockIdx.y, cbk = blockIdx.x;
  if (cbk < rb) return;
  __shared__ float s_col[64 * 4];
  const int tid = threadIdx.x;
  const int cols = min(64, n - cbk * 64), rows = min(64, n - rb * 64);
  if (tid < cols) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s_col[tid * 4 + k] = boxes[(long long)(cbk * 64 + tid) * 4 + k];
  }
  __syncthreads();
  if (tid >= rows) return;
  const float* b = boxes + (long long)(rb * 64 + tid) * 4;
  const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
  const float area = (x2 - x1 + one) * (y2 - y1 + one);
  unsigned long long bits = 0ull;
  const int start = (rb == cbk) ? tid + 1 : 0;
  for (int j = start; j < cols; ++j) {
    const float* c = s_col + j * 4;
    float w = fmaxf(0.f, fminf(x2, c[2]) - fmaxf(x1, c[0]) + one);
    float h = fmaxf(0.f, fminf(y2, c[3]) - fmaxf(y1, c[1]) + one);
    float inter = w * h;
    float carea = (c[2] - c[0] + one) * (c[3] - c[1] + one);
    if (inter / (area + carea - inter) > thr) bits |= 1ull << j;
  }
  mask[(long long)(rb * 64 + tid) * col_blocks + cbk] = bits;
}

__global__ void iota_kernel(int* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

}  // namespace rsdet

extern "C" size_t rsdet_nms_hbb_ws_size(int n) {
  if (n <= 0) return 0;
  size_t cb = ((size_t)n + 63) / 64;
  return (((size_t)n * 4 + 255) & ~(size_t)255) + (size_t)n * cb * sizeof(unsigned long long);
}

extern "C" int rsdet_nms_hbb_sorted_f32(const float* boxes_sorted, int n, float thr, int plus_one,
                                        uint8_t* keep_sorted, void* ws, size_t ws_bytes, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!boxes_sorted || !keep_sorted || !ws || ws_bytes < rsdet_nms_hbb_ws_size(n) || ((uintptr_t)ws & 15))
    return RSDET_EINVAL;
  const int cb = (n + 63) / 64;
  if (cb > rsdet::NMS_MAX_BLOCKS) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int* ident = (int*)ws;
  unsigned long long* mask = (unsigned long long*)((char*)ws + (((size_t)n * 4 + 255) & ~(size_t)255));
  hipLaunchKernelGGL(rsdet::iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ident, n);
  hipLaunchKernelGGL(rsdet::nms_hbb_mask_kernel, dim3(cb, cb), dim3(64), 0, s, boxes_sorted, n, thr,
                     plus_one ? 1.f : 0.f, cb, mask);
  hipLaunchKernelGGL(rsdet::nms_sweep_kernel, dim3(1), dim3(rsdet::NMS_NT), (size_t)(cb + 1) * 8, s, mask, n, cb,
                     ident, keep_sorted);
  return rsdet_launch_status();
}
