// convex_sort.hip -- convex_sort (ordered convex-hull indices of masked point sets) for gfx950.
//
// Replaces: convex_sort / convex_sort_gpu / convex_sort_cpu
//   /root/reference/python/jdet/ops/convex_sort.py:67-201 -- the tensor code :159-176 (masked argmin of y, cosine of
//   every point's direction from that point, descending argsort: five Jittor ops + a sort launch) AND the Graham-scan
//   kernel :5-64, in ONE launch.  Caller: models/losses/poly_iou_loss.py:23 with npts = 24 (16 edge intersections +
//   2 x 4 vertices) or 8.
//
// One lane per point set, one wave per workgroup.  A set's points are touched O(npts^2) times with data-dependent
// indices (insertion sort, hull stack pops), so they live in LDS in [point][lane] layout: lane l only ever touches
// column l, every access of a wave hits 64 consecutive banks/words -- no bank conflicts and no scratch memory.  The
// set data of a workgroup is contiguous in global memory and is staged with coalesced loads; the hull indices go
// out the same way.  Sets too large for 64 KB of LDS (npts > 56) run the same code on a global workspace.
//
// Semantics kept from the reference: `d < 0.000001` duplicate test against the current stack top, `t >= 0` keeps
// collinear points, popped stack slots are NOT reset (stale indices can follow the closing index; convex_sort.py
// never clears them and poly_iou_loss.convex_areas consumes the row as is), unused slots are -1.
// Tie rules of argmin / argsort (not pinned by the reference: Jittor): first index, stable descending.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"

namespace rsdet {

constexpr int CVX_LANES = 64;

// bytes of per-workgroup state for npts points per set (see the carve-up in the kernel)
__host__ __device__ inline size_t cvx_block_bytes(int npts) {
  return (size_t)CVX_LANES * ((size_t)npts * 16 + (size_t)(npts + 1) * 2 + 2);  // +2: keeps the total 4-byte aligned
}

template <bool USE_LDS>
__global__ __launch_bounds__(CVX_LANES) void convex_sort_kernel(const float* __restrict__ pts,
                                                                const float* __restrict__ masks, int nbs, int npts,
                                                                int circular, int* __restrict__ out,
                                                                unsigned char* __restrict__ ws) {
  extern __shared__ __align__(16) unsigned char cvx_smem[];
  unsigned char* base = USE_LDS ? cvx_smem : ws + (size_t)blockIdx.x * cvx_block_bytes(npts);
  const int L = CVX_LANES;
  float* xs = (float*)base;                       // [npts][64]
  float* ys = xs + (size_t)npts * L;              // [npts][64]
  float* key = ys + (size_t)npts * L;             // [npts][64] cosine keys, sorted in place
  uint16_t* ord = (uint16_t*)(key + (size_t)npts * L);  // [npts][64] point order
  uint16_t* valid = ord + (size_t)npts * L;       // [npts][64] mask >= 0.5
  int16_t* hull = (int16_t*)(valid + (size_t)npts * L);  // [npts + 1][64]

  const int lane = threadIdx.x;
  const long long set0 = (long long)blockIdx.x * L;
  const int nset = (int)min((long long)L, (long long)nbs - set0);
  const int index_size = circular ? npts + 1 : npts;

  // stage: coalesced reads of this workgroup's nset * npts * 2 contiguous floats, transposed into [point][lane]
  {
    const float* src = pts + set0 * npts * 2;
    const int total = nset * npts * 2;
    for (int i = lane; i < total; i += L) {
      const int s = i / (2 * npts), r = i - s * 2 * npts;
      const float v = src[i];
      ((r & 1) ? ys : xs)[(size_t)(r >> 1) * L + s] = v;
    }
  }
  __syncthreads();

  if (lane < nset) {
    // masked argmin of y (:168-169): masked_y = m * y + (1 - m) * INF, first index on ties
    const float* mrow = masks + (set0 + lane) * npts;
    float best = 0.f;
    int s0 = 0;
    for (int p = 0; p < npts; ++p) {
      const float m = mrow[p];
      valid[(size_t)p * L + lane] = m < 0.5 ? 0 : 1;  // :26 `sub_m[j] < 0.5` skips the point
      hull[(size_t)p * L + lane] = -1;
      const float my = m * ys[(size_t)p * L + lane] + (1.f - m) * 10000000.f;
      if (p == 0 || my < best) {
        best = my;
        s0 = p;
      }
    }
    hull[(size_t)npts * L + lane] = -1;
    const float sx = xs[(size_t)s0 * L + lane], sy = ys[(size_t)s0 * L + lane];
    // cosine keys (:173) + stable descending insertion sort (:174)
    for (int p = 0; p < npts; ++p) {
      const float dx = xs[(size_t)p * L + lane] - sx, dy = ys[(size_t)p * L + lane] - sy;
      const float k = dx / sqrtf(dx * dx + dy * dy + 0.000001f);
      int q = p - 1;
      while (q >= 0 && key[(size_t)q * L + lane] < k) {
        key[(size_t)(q + 1) * L + lane] = key[(size_t)q * L + lane];
        ord[(size_t)(q + 1) * L + lane] = ord[(size_t)q * L + lane];
        --q;
      }
      key[(size_t)(q + 1) * L + lane] = k;
      ord[(size_t)(q + 1) * L + lane] = (uint16_t)p;
    }
    // Graham scan (:19-63)
    if (npts > 0) {
      hull[lane] = (int16_t)s0;
      int top = 0;
      for (int kk = 0; kk < npts; ++kk) {
        const int j = ord[(size_t)kk * L + lane];
        if (j == s0 || !valid[(size_t)j * L + lane]) continue;
        const float x0 = xs[(size_t)j * L + lane], y0 = ys[(size_t)j * L + lane];
        int h1 = hull[(size_t)top * L + lane];
        float x1 = xs[(size_t)h1 * L + lane], y1 = ys[(size_t)h1 * L + lane];
        const float d = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
        if ((double)d < 0.000001) continue;  // :32 compares against a double literal
        if (top < 2) {
          hull[(size_t)(++top) * L + lane] = (int16_t)j;
          continue;
        }
        int h2 = hull[(size_t)(top - 1) * L + lane];
        float x2 = xs[(size_t)h2 * L + lane], y2 = ys[(size_t)h2 * L + lane];
        for (;;) {
          const float t = (x1 - x2) * (y0 - y2) - (y1 - y2) * (x0 - x2);
          if (t >= 0) {
            hull[(size_t)(++top) * L + lane] = (int16_t)j;
            break;
          }
          if (top <= 1) {
            hull[(size_t)top * L + lane] = (int16_t)j;
            break;
          }
          --top;
          h1 = hull[(size_t)top * L + lane];
          h2 = hull[(size_t)(top - 1) * L + lane];
          x1 = xs[(size_t)h1 * L + lane];
          y1 = ys[(size_t)h1 * L + lane];
          x2 = xs[(size_t)h2 * L + lane];
          y2 = ys[(size_t)h2 * L + lane];
        }
      }
      if (circular) hull[(size_t)(top + 1) * L + lane] = hull[lane];
    }
  }
  __syncthreads();

  // coalesced store of nset * index_size indices
  {
    int* dst = out + set0 * index_size;
    const int total = nset * index_size;
    for (int i = lane; i < total; i += L) {
      const int s = i / index_size, k = i - s * index_size;
      dst[i] = hull[(size_t)k * L + s];
    }
  }
}

}  // namespace rsdet

using namespace rsdet;

static const size_t CVX_LDS_LIMIT = 64 * 1024;

extern "C" size_t rsdet_convex_sort_ws_size(int nbs, int npts) {
  if (nbs <= 0 || npts <= 0) return 0;
  const size_t per_block = cvx_block_bytes(npts);
  if (per_block <= CVX_LDS_LIMIT) return 0;
  return per_block * (size_t)((nbs + CVX_LANES - 1) / CVX_LANES);
}

extern "C" int rsdet_convex_sort_f32(const float* pts, const float* masks, int nbs, int npts, int circular,
                                     int* convex_index, void* ws, size_t ws_bytes, void* stream) {
  if (nbs < 0 || npts < 0 || npts > 32767) return RSDET_EINVAL;
  const int index_size = circular ? npts + 1 : npts;
  if (nbs == 0 || index_size == 0) return RSDET_OK;
  if (!convex_index) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (npts == 0)  // :179-180: all -1
    return hipMemsetAsync(convex_index, 0xff, (size_t)nbs * index_size * 4, s) == hipSuccess ? RSDET_OK
                                                                                            : RSDET_ELAUNCH;
  if (!pts || !masks) return RSDET_EINVAL;
  if ((long long)nbs * index_size > 0x7fffffffLL) return RSDET_EINVAL;
  const unsigned blocks = (unsigned)((nbs + CVX_LANES - 1) / CVX_LANES);
  const size_t per_block = cvx_block_bytes(npts);
  if (per_block <= CVX_LDS_LIMIT) {
    hipLaunchKernelGGL(convex_sort_kernel<true>, dim3(blocks), dim3(CVX_LANES), per_block, s, pts, masks, nbs, npts,
                       circular, convex_index, (unsigned char*)nullptr);
  } else {
    if (!ws || ((uintptr_t)ws & 15) || ws_bytes < rsdet_convex_sort_ws_size(nbs, npts)) return RSDET_EINVAL;
    hipLaunchKernelGGL(convex_sort_kernel<false>, dim3(blocks), dim3(CVX_LANES), 0, s, pts, masks, nbs, npts,
                       circular, convex_index, (unsigned char*)ws);
  }
  return rsdet_launch_status();
}
