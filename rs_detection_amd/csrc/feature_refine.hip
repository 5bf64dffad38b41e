// feature_refine.hip -- FeatureRefine (R3Det) forward / backward for gfx950.
//
// Replaces: feature_refine_forward / feature_refine_backward
//   /root/reference/python/jdet/ops/fr.py:234-252, kernels :113-173 (forward) and :175-232 (backward).
//
//   out[n,c,h,w] = feat[n,c,h,w] + sum_{i < points} bilinear(feat[n,c], py_i, px_i)
// where the `points` (1 or 5: centre, then the four corners) come from best_bboxes[n,h,w,:] -- and are the same
// for every channel.  The reference recomputes them (2 transcendental calls, 20 corner weights) once per (n,c,h,w)
// element; here one thread owns a position, derives its footprints once into registers and walks the channels,
// so the per-element work is 4*points loads + 1 store and consecutive lanes read / write consecutive w.
//
// Reference quirk kept (fr.py:131-133): entry 0 of the box is used as the ROW coordinate and entry 1 as the COLUMN
// (`roi_y = bbox[0] * scale; roi_x = bbox[1] * scale`) although the boxes are (x_ctr, y_ctr, w, h, angle).
//
// Backward = gather form (as col2im / RROIAlign backward): the (position, point, corner) -> pixel map is inverted
// on integers, then one wave per pixel sums its terms from the channels-last gradient.  No fp32 atomics (the
// reference issues 1 + 4*points per element), grad_in written exactly once.  (The order of a pixel's terms follows
// the integer atomics of the fill stage, so the last bit of a sum can differ between runs, as with atomicAdd.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bilinear.h"

namespace rsdet {

constexpr int FR_MAX_POINTS = 5;

struct FrPoints {
  float py[FR_MAX_POINTS], px[FR_MAX_POINTS];
};

// fr.py:128-152
__device__ __forceinline__ FrPoints fr_points(const float* __restrict__ b, float scale, int points) {
  FrPoints p;
  const float roi_y = b[0] * scale;
  const float roi_x = b[1] * scale;
#pragma unroll
  for (int i = 0; i < FR_MAX_POINTS; ++i) {
    p.px[i] = 0.f;
    p.py[i] = 0.f;
  }
  p.px[0] = roi_x;
  p.py[0] = roi_y;
  if (points > 1) {
    const float roi_w = b[2] * scale, roi_h = b[3] * scale, roi_a = b[4];
    const float w_2 = roi_w / 2, h_2 = roi_h / 2;
    const float cosa = cosf(roi_a), sina = sinf(roi_a);
    const float wx = cosa * w_2, wy = sina * w_2;
    const float hx = -sina * h_2, hy = cosa * h_2;
    p.px[1] = roi_x + wx + hx; p.py[1] = roi_y + wy + hy;
    p.px[2] = roi_x - wx + hx; p.py[2] = roi_y - wy + hy;
    p.px[3] = roi_x - wx - hx; p.py[3] = roi_y - wy - hy;
    p.px[4] = roi_x + wx - hx; p.py[4] = roi_y + wy - hy;
  }
  return p;
}

// 1-D grid over (position blocks of all images) x (channel slices) in XCD bands (rsdet_xcd_band): the gathers of
// neighbouring rows meet in one L2 (386 MB of L2 misses for a 33.5 MB input with a plain (x, y, z) grid).  POINTS is 1 or 5 at compile time: footprints in
// registers.
template <int POINTS>
__global__ __launch_bounds__(256) void fr_forward_kernel(const float* __restrict__ feat,
                                                         const float* __restrict__ boxes, float scale, int C, int H,
                                                         int W, int pblocks, int npb, int slices, float* __restrict__ out) {
  const int HW = H * W;
  const RsdetBandItem it = rsdet_xcd_band(blockIdx.x, npb, slices);  // npb = N * pblocks
  if (!it.valid) return;
  const int slice = it.inner;
  const int pbn = it.outer;  // n * pblocks + position block
  const int n = pbn / pblocks;
  const int pos = (pbn - n * pblocks) * 256 + threadIdx.x;
  if (pos >= HW) return;
  const FrPoints p = fr_points(boxes + ((long long)n * HW + pos) * 5, scale, POINTS);
  int o00[POINTS], o01[POINTS], o10[POINTS], o11[POINTS];
  float w1[POINTS], w2[POINTS], w3[POINTS], w4[POINTS];
#pragma unroll
  for (int i = 0; i < POINTS; ++i) {
    const Bil b = bilinear(H, W, p.py[i], p.px[i]);
    const bool in = b.yl >= 0;  // outside: weight 0 on a valid address (fr.py:26-28 returns 0)
    o00[i] = in ? b.yl * W + b.xl : 0;
    o01[i] = in ? b.yl * W + b.xh : 0;
    o10[i] = in ? b.yh * W + b.xl : 0;
    o11[i] = in ? b.yh * W + b.xh : 0;
    w1[i] = b.w1; w2[i] = b.w2; w3[i] = b.w3; w4[i] = b.w4;
  }
  // FR_U channels per trip: all their loads are issued before the first use (the gathers are independent), which is
  // what hides the ~2 us HBM latency of a trip; the tail trips clamp the channel and drop the store.
  constexpr int FR_U = POINTS == 1 ? 8 : 4;
  const int cstep = slices;
  for (int c0 = slice; c0 < C; c0 += FR_U * cstep) {
    float v[FR_U][4 * POINTS + 1];
#pragma unroll
    for (int u = 0; u < FR_U; ++u) {
      const int c = min(c0 + u * cstep, C - 1);
      const float* fp = feat + ((long long)n * C + c) * HW;
      v[u][4 * POINTS] = fp[pos];
#pragma unroll
      for (int i = 0; i < POINTS; ++i) {
        v[u][4 * i + 0] = fp[o00[i]];
        v[u][4 * i + 1] = fp[o01[i]];
        v[u][4 * i + 2] = fp[o10[i]];
        v[u][4 * i + 3] = fp[o11[i]];
      }
    }
#pragma unroll
    for (int u = 0; u < FR_U; ++u) {
      const int c = c0 + u * cstep;
      float acc = v[u][4 * POINTS];
#pragma unroll
      for (int i = 0; i < POINTS; ++i)  // :55-63, :166-169
        acc += w1[i] * v[u][4 * i] + w2[i] * v[u][4 * i + 1] + w3[i] * v[u][4 * i + 2] + w4[i] * v[u][4 * i + 3];
      if (c < C) out[((long long)n * C + c) * HW + pos] = acc;
    }
  }
}

// ---- channels-last forward -------------------------------------------------------------------------------------------
// feat / out (N, H, W, C): a sample is a CONTIGUOUS channel vector, so the 4 * POINTS + 1 reads of a position are
// wave-wide 16-byte-per-lane loads of consecutive addresses instead of 21 gathers per channel whose lanes scatter over
// rows (the NCHW form above: L2-gather-bound at 0.04 of the HBM roofline).  LPP = min(64, C / 4) lanes own one position
// (4 channels each), 64 / LPP positions per wave pass, channel chunks of 256 for wider maps.  The footprints are computed
// redundantly by the lanes of a position (~150 VALU instructions against 21 KB of loads per 256 channels).  The four
// waves of a workgroup take NEIGHBOURING positions at the same time (their corner pixels overlap: L1 hits), workgroup
// ids are mapped so that each XCD's L2 sees one contiguous band of rows.  Same sums in the same order as the NCHW form.
template <int POINTS>
__global__ __launch_bounds__(256) void fr_forward_nhwc_kernel(const float* __restrict__ feat,
                                                              const float* __restrict__ boxes, float scale, int C, int H,
                                                              int W, long long npos, int lpp, int per_block, int nblocks,
                                                              float* __restrict__ out) {
  const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
  const int band = (nblocks + 7) >> 3;
  const int chunk = xcd * band + slot;
  if (slot >= band || chunk >= nblocks) return;
  const int tid = threadIdx.x;
  const int ppb = 256 / lpp;                          // positions the workgroup's lanes cover at once
  const int sub = tid / lpp, cl = tid - sub * lpp;     // position slot of this lane, its float4 inside a 4 * lpp chunk
  const long long HW = (long long)H * W;
  const long long p_end = min(npos, (long long)(chunk + 1) * per_block);
  for (long long p = (long long)chunk * per_block + sub; p < p_end; p += ppb) {
    const long long n = p / HW;
    const FrPoints pt = fr_points(boxes + p * 5, scale, POINTS);
    int o[POINTS][4];                                 // element offsets inside the image (H W C < 2^31: host-checked)
    float w[POINTS][4];
#pragma unroll
    for (int i = 0; i < POINTS; ++i) {
      const Bil b = bilinear(H, W, pt.py[i], pt.px[i]);
      const bool in = b.yl >= 0;                      // outside: weight 0 on a valid address (fr.py:26-28 returns 0)
      o[i][0] = (in ? b.yl * W + b.xl : 0) * C;
      o[i][1] = (in ? b.yl * W + b.xh : 0) * C;
      o[i][2] = (in ? b.yh * W + b.xl : 0) * C;
      o[i][3] = (in ? b.yh * W + b.xh : 0) * C;
      w[i][0] = b.w1, w[i][1] = b.w2, w[i][2] = b.w3, w[i][3] = b.w4;
    }
    const float* img = feat + n * HW * C;
    for (int c = cl * 4; c < C; c += lpp * 4) {
      float4 acc = *reinterpret_cast<const float4*>(feat + p * C + c);
      // (written as two groups of loads; the compiler issues all 4 POINTS + 1 vectors of a lane back to back anyway --
      // 21 KB in flight per wave, 192 registers, 2 waves per SIMD: enough requests in flight for a memory-bound pass)
      constexpr int SPLIT = POINTS > 3 ? 3 : POINTS;
      float4 v[SPLIT][4];
#pragma unroll
      for (int i = 0; i < SPLIT; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[i][k] = *reinterpret_cast<const float4*>(img + o[i][k] + c);
#pragma unroll
      for (int i = 0; i < SPLIT; ++i) {                // :55-63, :166-169 -- the NCHW form's expression per channel
        acc.x += w[i][0] * v[i][0].x + w[i][1] * v[i][1].x + w[i][2] * v[i][2].x + w[i][3] * v[i][3].x;
        acc.y += w[i][0] * v[i][0].y + w[i][1] * v[i][1].y + w[i][2] * v[i][2].y + w[i][3] * v[i][3].y;
        acc.z += w[i][0] * v[i][0].z + w[i][1] * v[i][1].z + w[i][2] * v[i][2].z + w[i][3] * v[i][3].z;
        acc.w += w[i][0] * v[i][0].w + w[i][1] * v[i][1].w + w[i][2] * v[i][2].w + w[i][3] * v[i][3].w;
      }
      if (POINTS > SPLIT) {
        float4 u[POINTS - SPLIT > 0 ? POINTS - SPLIT : 1][4];
#pragma unroll
        for (int i = SPLIT; i < POINTS; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) u[i - SPLIT][k] = *reinterpret_cast<const float4*>(img + o[i][k] + c);
#pragma unroll
        for (int i = SPLIT; i < POINTS; ++i) {
          const int j = i - SPLIT;
          acc.x += w[i][0] * u[j][0].x + w[i][1] * u[j][1].x + w[i][2] * u[j][2].x + w[i][3] * u[j][3].x;
          acc.y += w[i][0] * u[j][0].y + w[i][1] * u[j][1].y + w[i][2] * u[j][2].y + w[i][3] * u[j][3].y;
          acc.z += w[i][0] * u[j][0].z + w[i][1] * u[j][1].z + w[i][2] * u[j][2].z + w[i][3] * u[j][3].z;
          acc.w += w[i][0] * u[j][0].w + w[i][1] * u[j][1].w + w[i][2] * u[j][2].w + w[i][3] * u[j][3].w;
        }
      }
      *reinterpret_cast<float4*>(out + p * C + c) = acc;
    }
  }
}

// ---- backward: invert (position, point, corner) -> pixel ---------------------------------------------------
// item = position * (points + 1) + k; k == points is the identity term (fr.py:213 atomicAdd(bottom_diff + index)).
struct FrItem {
  long long p[4];
  float w[4];
};

__device__ __forceinline__ FrItem fr_item(const float* __restrict__ boxes, long long item, int H, int W, float scale,
                                          int points) {
  const long long pos = item / (points + 1);
  const int k = (int)(item - pos * (points + 1));
  FrItem t;
  t.p[1] = t.p[2] = t.p[3] = -1;
  t.w[1] = t.w[2] = t.w[3] = 0.f;
  if (k == points) {
    t.p[0] = pos;
    t.w[0] = 1.f;
    return t;
  }
  const FrPoints p = fr_points(boxes + pos * 5, scale, points);
  float py = p.py[0], px = p.px[0];
#pragma unroll
  for (int i = 1; i < FR_MAX_POINTS; ++i)
    if (k == i) {
      py = p.py[i];
      px = p.px[i];
    }
  const Bil b = bilinear(H, W, py, px);
  const long long base = pos / ((long long)H * W) * ((long long)H * W);
  const bool in = b.yl >= 0;
  t.p[0] = in ? base + b.yl * W + b.xl : -1;
  t.p[1] = in ? base + b.yl * W + b.xh : -1;
  t.p[2] = in ? base + b.yh * W + b.xl : -1;
  t.p[3] = in ? base + b.yh * W + b.xh : -1;
  t.w[0] = b.w1; t.w[1] = b.w2; t.w[2] = b.w3; t.w[3] = b.w4;
  return t;
}

__global__ __launch_bounds__(256) void fr_idx_count_kernel(const float* __restrict__ boxes, long long items, int H,
                                                           int W, float scale, int points, int* __restrict__ cnt) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  const FrItem t = fr_item(boxes, item, H, W, scale, points);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.p[k] >= 0) atomicAdd(cnt + t.p[k], 1);
}

__global__ __launch_bounds__(256) void fr_idx_fill_kernel(const float* __restrict__ boxes, long long items, int H,
                                                          int W, float scale, int points,
                                                          const int* __restrict__ start, int* __restrict__ fill,
                                                          int* __restrict__ ent_row, float* __restrict__ ent_w) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  const FrItem t = fr_item(boxes, item, H, W, scale, points);
  const int row = (int)(item / (points + 1));  // position = row of the channels-last gradient (N*H*W, C)
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.p[k] >= 0) {
      const int slot = start[t.p[k]] + atomicAdd(fill + t.p[k], 1);
      ent_row[slot] = row;
      ent_w[slot] = t.w[k];
    }
}

}  // namespace rsdet

using namespace rsdet;

static int fr_check(int N, int C, int H, int W, int points) {
  if (N < 0 || C < 0 || H < 1 || W < 1) return RSDET_EINVAL;
  if (points != 1 && points != 5) return RSDET_EINVAL;  // fr.py:261 assert points in [1, 5]
  if ((long long)H * W > 0x7fffffffLL || (long long)N * ((H * (long long)W + 255) / 256) > (1LL << 24)) return RSDET_EINVAL;
  return RSDET_OK;
}

extern "C" int rsdet_feature_refine_forward_f32(const float* feat, const float* best_bboxes, int N, int C, int H,
                                                int W, float spatial_scale, int points, float* out, void* stream) {
  int rc = fr_check(N, C, H, W, points);
  if (rc) return rc;
  if (N == 0 || C == 0) return RSDET_OK;
  if (!feat || !best_bboxes || !out) return RSDET_EINVAL;
  const int HW = H * W;
  const int bx = (HW + 255) / 256;
  // enough workgroups for 256 CUs, but never fewer than 8 channels per thread (the footprints are amortised over them)
  int cz = 2048 / (bx * N > 0 ? bx * N : 1);
  cz = cz < 1 ? 1 : cz;
  const int cz_max = (C + 7) / 8;
  cz = cz > cz_max ? cz_max : cz;
  const dim3 grid((unsigned)rsdet_xcd_band_grid((long long)bx * N, cz));
  if (points == 1)
    hipLaunchKernelGGL(fr_forward_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, feat, best_bboxes,
                       spatial_scale, C, H, W, bx, bx * N, cz, out);
  else
    hipLaunchKernelGGL(fr_forward_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, feat, best_bboxes,
                       spatial_scale, C, H, W, bx, bx * N, cz, out);
  return rsdet_launch_status();
}

// feat, out: (N, H, W, C) channels-last fp32; C % 4 == 0 and C / 4 a power of two up to 64 or a multiple of 64.
extern "C" int rsdet_feature_refine_forward_nhwc_supported(int C) {
  if (C < 4 || (C & 3)) return 0;
  const int q = C / 4;
  return (q >= 64) ? (q % 64 == 0) : ((q & (q - 1)) == 0);
}
extern "C" int rsdet_feature_refine_forward_nhwc_f32(const float* feat, const float* best_bboxes, int N, int C, int H,
                                                     int W, float spatial_scale, int points, float* out, void* stream) {
  int rc = fr_check(N, C, H, W, points);
  if (rc) return rc;
  if (N == 0 || C == 0) return RSDET_OK;
  if (!rsdet_feature_refine_forward_nhwc_supported(C)) return RSDET_EINVAL;
  if (!feat || !best_bboxes || !out || (((uintptr_t)feat | (uintptr_t)out) & 15)) return RSDET_EINVAL;
  if ((long long)H * W * C > 0x7fffffffLL) return RSDET_EINVAL;
  const long long npos = (long long)N * H * W;
  const int lpp = C / 4 >= 64 ? 64 : C / 4;
  // positions per workgroup: ~8 passes of its 256 / lpp position slots, but at least ~2 048 workgroups when there is work
  const int ppb = 256 / lpp;
  long long per = (long long)ppb * 8;
  while (per > ppb && (npos + per - 1) / per < 2048) per -= ppb;
  const long long nblocks = (npos + per - 1) / per;
  if (nblocks > 0x7fffff00LL) return RSDET_EINVAL;
  const unsigned grid = (unsigned)(((nblocks + 7) / 8) * 8);
  if (points == 1)
    hipLaunchKernelGGL(fr_forward_nhwc_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, best_bboxes,
                       spatial_scale, C, H, W, npos, lpp, (int)per, (int)nblocks, out);
  else
    hipLaunchKernelGGL(fr_forward_nhwc_kernel<5>, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, best_bboxes,
                       spatial_scale, C, H, W, npos, lpp, (int)per, (int)nblocks, out);
  return rsdet_launch_status();
}

static inline size_t fr_align256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t rsdet_feature_refine_backward_ws_size(int N, int H, int W, int points) {
  if (N < 1 || H < 1 || W < 1 || (points != 1 && points != 5)) return 0;
  const size_t npix = (size_t)N * H * W, ent = npix * (4 * (size_t)points + 1);
  return fr_align256((npix + 1) * 4) * 2 + fr_align256(ent * 4) * 2 + fr_align256((npix / 4096 + 1) * 4);
}

extern "C" int rsdet_feature_refine_backward_nhwc_f32(const float* grad_out_nhwc, const float* best_bboxes, int N,
                                                      int C, int H, int W, float spatial_scale, int points,
                                                      float* grad_in_nhwc, void* ws, size_t ws_bytes, void* stream) {
  int rc = fr_check(N, C, H, W, points);
  if (rc) return rc;
  if (N == 0 || C == 0) return RSDET_OK;
  if (!grad_out_nhwc || !best_bboxes || !grad_in_nhwc) return RSDET_EINVAL;
  const long long npix = (long long)N * H * W;
  const long long items = npix * (points + 1);
  if (npix * (4 * points + 1) > 0x7fffffffLL) return RSDET_EINVAL;
  if (!ws || ((uintptr_t)ws & 15) || ws_bytes < rsdet_feature_refine_backward_ws_size(N, H, W, points))
    return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t ent = (size_t)npix * (4 * points + 1);
  char* w = (char*)ws;
  int* cnt = (int*)w;
  int* start = (int*)(w + fr_align256((npix + 1) * 4));
  int* ent_row = (int*)(w + fr_align256((npix + 1) * 4) * 2);
  float* ent_w = (float*)(w + fr_align256((npix + 1) * 4) * 2 + fr_align256(ent * 4));
  int* chunk_sum = (int*)(w + fr_align256((npix + 1) * 4) * 2 + fr_align256(ent * 4) * 2);
  if (hipMemsetAsync(cnt, 0, (size_t)(npix + 1) * 4, s) != hipSuccess) return RSDET_ELAUNCH;
  const unsigned ib = (unsigned)((items + 255) / 256);
  hipLaunchKernelGGL(fr_idx_count_kernel, dim3(ib), dim3(256), 0, s, best_bboxes, items, H, W, spatial_scale, points,
                     cnt);
  rsdet_launch_index_scan(cnt, npix, chunk_sum, start, s);
  hipLaunchKernelGGL(fr_idx_fill_kernel, dim3(ib), dim3(256), 0, s, best_bboxes, items, H, W, spatial_scale, points,
                     start, cnt, ent_row, ent_w);
  rsdet_launch_pixel_gather(grad_out_nhwc, start, ent_row, ent_w, npix, C, grad_in_nhwc, s);
  return rsdet_launch_status();
}
