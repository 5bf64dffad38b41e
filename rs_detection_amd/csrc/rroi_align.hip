// rroi_align.hip -- ROIAlignRotated_v1 and ROIAlignRotated (v0) forward / backward for gfx950.
//
// Replaces: _RotatedROIAlign_v1.execute / .grad
//   /root/reference/python/jdet/ops/roi_align_rotated_v1.py:300-351,
//   kernels ROIAlignRotatedForward :71-147, ROIAlignBackward :193-298,
//   bilinear helpers :24-68, :149-190;
// and _RotatedROIAlign.execute / .grad of ops/roi_align_rotated.py:256-309 (kernels :59-126, :170-254), which
// differ from v1 in two places only: the RoI centre has no -0.5 pixel shift (:76-77 vs v1 :89-90) and the frame is
// rotated the other way (:116-117 vs v1 :133-134, i.e. sin(theta) enters with the opposite sign).  `v0` below
// selects that variant; everything else is shared.
//
// One workgroup per RoI.  The RoI frame (centre, bin size, sin/cos) is derived
// once per workgroup into LDS instead of once per output element; threads then
// walk (channel, bin) pairs bin-fastest, so the 49 bins of one channel read one
// feature plane around the RoI (L2/L1-resident) and the output store is
// contiguous.  Backward uses fp32 atomics like the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bilinear.h"

namespace rsdet {

struct RoiFrame {
  int batch, gh, gw;
  float cw, ch, bin_h, bin_w, start_h, start_w, cs, sn;
};

__device__ __forceinline__ RoiFrame make_frame(const float* roi, float scale, int sample_num,
                                               int PH, int PW, int v0) {
  RoiFrame f;
  f.batch = (int)roi[0];
  f.cw = roi[1] * scale;  // :89-90 "do not round"
  f.ch = roi[2] * scale;
  if (!v0) {
    f.cw -= 0.5f;
    f.ch -= 0.5f;
  }
  float rw = fmaxf(roi[3] * scale, 1.f);
  float rh = fmaxf(roi[4] * scale, 1.f);
  float theta = roi[5];
  f.bin_h = rh / (float)PH;
  f.bin_w = rw / (float)PW;
  f.gh = sample_num > 0 ? sample_num : (int)ceilf(rh / PH);
  f.gw = sample_num > 0 ? sample_num : (int)ceilf(rw / PW);
  f.start_h = -rh / 2.0f;
  f.start_w = -rw / 2.0f;
  f.cs = cosf(theta);
  f.sn = v0 ? -sinf(theta) : sinf(theta);  // x = xx*cs + yy*sn, y = yy*cs - xx*sn below
  return f;
}

constexpr int RROI_NT = 256;

__global__ __launch_bounds__(RROI_NT) void rroi_forward_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, int C, int H, int W, int PH,
    int PW, float scale, int sample_num, int v0, float* __restrict__ out) {
  __shared__ RoiFrame s_f;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) s_f = make_frame(rois + (long long)n * 6, scale, sample_num, PH, PW, v0);
  __syncthreads();
  const RoiFrame f = s_f;
  const int bins = PH * PW;
  const float count = (float)max(f.gh * f.gw, 1);
  const long long HW = (long long)H * W;
  for (int e = blockIdx.y * RROI_NT + threadIdx.x; e < C * bins; e += gridDim.y * RROI_NT) {
    int c = e / bins, bin = e - c * bins;
    int ph = bin / PW, pw = bin - ph * PW;
    const float* fp = feat + ((long long)f.batch * C + c) * HW;
    float acc = 0.f;
    for (int iy = 0; iy < f.gh; ++iy) {
      float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
      for (int ix = 0; ix < f.gw; ++ix) {
        float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
        float x = xx * f.cs + yy * f.sn + f.cw;  // :133-134
        float y = yy * f.cs - xx * f.sn + f.ch;
        Bil b = bilinear(H, W, y, x);
        float v = 0.f;
        if (b.yl >= 0)
          v = b.w1 * fp[b.yl * W + b.xl] + b.w2 * fp[b.yl * W + b.xh] +
              b.w3 * fp[b.yh * W + b.xl] + b.w4 * fp[b.yh * W + b.xh];
        acc += v;
      }
    }
    out[(long long)n * C * bins + e] = acc / count;
  }
}

// The forward of OrientedSingleRoIExtractor (oriented_single_level.py:91-114) in one launch: every RoI samples the map of
// ITS pyramid level (lvl[n]).  The sync-free form of the extractor ran one launch per level over ALL RoIs with the
// other levels' RoIs pushed outside the map, then added the four (R, C, PH, PW) results: 4 x 115 us + 3 adds of 51 MB.
struct RroiLevels {
  const float* feat[RSDET_RROI_MAX_LEVELS];
  int H[RSDET_RROI_MAX_LEVELS], W[RSDET_RROI_MAX_LEVELS];
  float scale[RSDET_RROI_MAX_LEVELS];
};

__global__ __launch_bounds__(RROI_NT) void rroi_forward_levels_kernel(const RroiLevels lv, const float* __restrict__ rois,
                                                                      const int* __restrict__ lvl, int n_levels, int C,
                                                                      int PH, int PW, int sample_num, int v0,
                                                                      float* __restrict__ out) {
  __shared__ RoiFrame s_f;
  const int n = blockIdx.x;
  int l = lvl[n];
  l = l < 0 ? 0 : (l >= n_levels ? n_levels - 1 : l);
  const int H = lv.H[l], W = lv.W[l];
  if (threadIdx.x == 0) s_f = make_frame(rois + (long long)n * 6, lv.scale[l], sample_num, PH, PW, v0);
  __syncthreads();
  const RoiFrame f = s_f;
  const int bins = PH * PW;
  const float count = (float)max(f.gh * f.gw, 1);
  const long long HW = (long long)H * W;
  const float* __restrict__ feat = lv.feat[l];
  for (int e = blockIdx.y * RROI_NT + threadIdx.x; e < C * bins; e += gridDim.y * RROI_NT) {
    int c = e / bins, bin = e - c * bins;
    int ph = bin / PW, pw = bin - ph * PW;
    const float* fp = feat + ((long long)f.batch * C + c) * HW;
    float acc = 0.f;
    for (int iy = 0; iy < f.gh; ++iy) {
      float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
      for (int ix = 0; ix < f.gw; ++ix) {
        float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
        float x = xx * f.cs + yy * f.sn + f.cw;  // :133-134
        float y = yy * f.cs - xx * f.sn + f.ch;
        Bil b = bilinear(H, W, y, x);
        float v = 0.f;
        if (b.yl >= 0)
          v = b.w1 * fp[b.yl * W + b.xl] + b.w2 * fp[b.yl * W + b.xh] +
              b.w3 * fp[b.yh * W + b.xl] + b.w4 * fp[b.yh * W + b.xh];
        acc += v;
      }
    }
    out[(long long)n * C * bins + e] = acc / count;
  }
}

__global__ __launch_bounds__(RROI_NT) void rroi_backward_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, int C, int H, int W, int PH,
    int PW, float scale, int sample_num, int v0, float* __restrict__ grad_feat) {
  __shared__ RoiFrame s_f;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) s_f = make_frame(rois + (long long)n * 6, scale, sample_num, PH, PW, v0);
  __syncthreads();
  const RoiFrame f = s_f;
  const int bins = PH * PW;
  const float count = (float)(f.gh * f.gw);  // :245
  const long long HW = (long long)H * W;
  for (int e = blockIdx.y * RROI_NT + threadIdx.x; e < C * bins; e += gridDim.y * RROI_NT) {
    int c = e / bins, bin = e - c * bins;
    int ph = bin / PW, pw = bin - ph * PW;
    float* gp = grad_feat + ((long long)f.batch * C + c) * HW;
    float top = grad_out[(long long)n * C * bins + e];
    for (int iy = 0; iy < f.gh; ++iy) {
      float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
      for (int ix = 0; ix < f.gw; ++ix) {
        float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
        float x = xx * f.cs + yy * f.sn + f.cw;
        float y = yy * f.cs - xx * f.sn + f.ch;
        Bil b = bilinear(H, W, y, x);
        if (b.yl >= 0) {
          atomicAdd(gp + b.yl * W + b.xl, top * b.w1 / count);
          atomicAdd(gp + b.yl * W + b.xh, top * b.w2 / count);
          atomicAdd(gp + b.yh * W + b.xl, top * b.w3 / count);
          atomicAdd(gp + b.yh * W + b.xh, top * b.w4 / count);
        }
      }
    }
  }
}

// ---- gather form of the backward (fixed sample_num): no floating-point atomics ---------------------------
// The scatter kernel above issues R*C*PH*PW*gh*gw*4 fp32 atomics, lanes of a wave aiming at the same few cache
// lines of one channel plane (120 M atomics, 3.1 ms for 600 RoIs on a 256 x 256 x 256 level).  The footprints do not
// depend on the channel: invert (roi, bin, sample, corner) -> pixel once on integers (0.47 M entries), then one
// wave per pixel sums its terms from the channels-last output gradient (R, PH*PW, C) -- 1-KiB coalesced reads,
// one store per element of the channels-last input gradient (N, H, W, C), no zero fill.
struct RroiItem {
  long long p[4];  // input pixel index (batch*H + y)*W + x of the four corners, -1 = outside
  float w[4];      // bilinear weight / samples per bin
};

__device__ __forceinline__ RroiItem rroi_item(const float* __restrict__ rois, long long item, int H, int W, int PH,
                                             int PW, float scale, int sample_num, int v0) {
  const int spb = sample_num * sample_num, bins = PH * PW;
  const long long n = item / ((long long)bins * spb);
  const int rem = (int)(item - n * bins * spb);
  const int bin = rem / spb, smp = rem - bin * spb;
  const int ph = bin / PW, pw = bin - ph * PW, iy = smp / sample_num, ix = smp - iy * sample_num;
  const RoiFrame f = make_frame(rois + n * 6, scale, sample_num, PH, PW, v0);
  const float yy = f.start_h + ph * f.bin_h + (float)(iy + .5f) * f.bin_h / (float)f.gh;
  const float xx = f.start_w + pw * f.bin_w + (float)(ix + .5f) * f.bin_w / (float)f.gw;
  const float x = xx * f.cs + yy * f.sn + f.cw;
  const float y = yy * f.cs - xx * f.sn + f.ch;
  const Bil b = bilinear(H, W, y, x);
  RroiItem t;
  const float count = (float)(f.gh * f.gw);
  const long long base = (long long)f.batch * H * W;
  const bool in = b.yl >= 0;
  t.p[0] = in ? base + b.yl * W + b.xl : -1;
  t.p[1] = in ? base + b.yl * W + b.xh : -1;
  t.p[2] = in ? base + b.yh * W + b.xl : -1;
  t.p[3] = in ? base + b.yh * W + b.xh : -1;
  t.w[0] = b.w1 / count;
  t.w[1] = b.w2 / count;
  t.w[2] = b.w3 / count;
  t.w[3] = b.w4 / count;
  return t;
}

__global__ __launch_bounds__(256) void rroi_idx_count_kernel(const float* __restrict__ rois, long long items, int H,
                                                             int W, int PH, int PW, float scale, int sample_num,
                                                             int v0, long long npix, int* __restrict__ cnt) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  const RroiItem t = rroi_item(rois, item, H, W, PH, PW, scale, sample_num, v0);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.p[k] >= 0 && t.p[k] < npix) atomicAdd(cnt + t.p[k], 1);
}

__global__ __launch_bounds__(256) void rroi_idx_fill_kernel(const float* __restrict__ rois, long long items, int H,
                                                            int W, int PH, int PW, float scale, int sample_num,
                                                            int v0, long long npix, const int* __restrict__ start,
                                                            int* __restrict__ fill, int* __restrict__ ent_row,
                                                            float* __restrict__ ent_w) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  const RroiItem t = rroi_item(rois, item, H, W, PH, PW, scale, sample_num, v0);
  const int row = (int)(item / (sample_num * sample_num));  // roi * PH*PW + bin: row of the (R, PH*PW, C) gradient
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.p[k] >= 0 && t.p[k] < npix) {
      const int slot = start[t.p[k]] + atomicAdd(fill + t.p[k], 1);
      ent_row[slot] = row;
      ent_w[slot] = t.w[k];
    }
}

// ---- the same index for the levels of an extractor at once: RoI n belongs to level lvl[n] and to no other ----
// Pixels of the levels lie end to end (pix_base); one histogram / scan / fill over every (RoI, bin, sample) -- each on the
// geometry of ITS level -- replaces a build per level over all RoIs (the other levels' RoIs moved outside the map).
struct RroiLevelGeom {
  int n;
  int H[RSDET_RROI_MAX_LEVELS], W[RSDET_RROI_MAX_LEVELS];
  float scale[RSDET_RROI_MAX_LEVELS];
  long long pix_base[RSDET_RROI_MAX_LEVELS + 1];
};

__device__ __forceinline__ RroiItem rroi_item_levels(const RroiLevelGeom& g, const float* __restrict__ rois,
                                                    const int* __restrict__ lvl, long long item, int PH, int PW,
                                                    int sample_num, int v0, long long* base, long long* end) {
  const long long n = item / ((long long)PH * PW * sample_num * sample_num);
  int l = lvl[n];
  l = l < 0 ? 0 : (l >= g.n ? g.n - 1 : l);
  *base = g.pix_base[l];
  *end = g.pix_base[l + 1] - g.pix_base[l];
  return rroi_item(rois, item, g.H[l], g.W[l], PH, PW, g.scale[l], sample_num, v0);
}

__global__ __launch_bounds__(256) void rroi_idx_count_levels_kernel(const RroiLevelGeom g, const float* __restrict__ rois,
                                                                    const int* __restrict__ lvl, long long items, int PH,
                                                                    int PW, int sample_num, int v0, int* __restrict__ cnt) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  long long base, npix;
  const RroiItem t = rroi_item_levels(g, rois, lvl, item, PH, PW, sample_num, v0, &base, &npix);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.p[k] >= 0 && t.p[k] < npix) atomicAdd(cnt + base + t.p[k], 1);
}

__global__ __launch_bounds__(256) void rroi_idx_fill_levels_kernel(const RroiLevelGeom g, const float* __restrict__ rois,
                                                                   const int* __restrict__ lvl, long long items, int PH,
                                                                   int PW, int sample_num, int v0,
                                                                   const int* __restrict__ start, int* __restrict__ fill,
                                                                   int* __restrict__ ent_row, float* __restrict__ ent_w) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  long long base, npix;
  const RroiItem t = rroi_item_levels(g, rois, lvl, item, PH, PW, sample_num, v0, &base, &npix);
  const int row = (int)(item / (sample_num * sample_num));  // roi * PH*PW + bin: row of the (R, PH*PW, C) gradient
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t.p[k] >= 0 && t.p[k] < npix) {
      const int slot = start[base + t.p[k]] + atomicAdd(fill + base + t.p[k], 1);
      ent_row[slot] = row;
      ent_w[slot] = t.w[k];
    }
}

template <bool VEC4>
__global__ __launch_bounds__(256) void rroi_gather_kernel(const float* __restrict__ go_t, const int* __restrict__ start,
                                                          const int* __restrict__ ent_row,
                                                          const float* __restrict__ ent_w, long long npix, int C,
                                                          float* __restrict__ grad_nhwc) {
  const int lane = threadIdx.x & 63;
  // XCD-contiguous workgroup order: neighbouring pixels share gradient rows, keep them on one L2
  const long long pix = (long long)rsdet_xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const int e0 = start[pix], e1 = start[pix + 1];
  if (VEC4) {
    for (int c = lane * 4; c < C; c += 256) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int e = e0;
      for (; e + 2 <= e1; e += 2) {
        const int r0 = ent_row[e], r1 = ent_row[e + 1];
        const float w0 = ent_w[e], w1 = ent_w[e + 1];
        const float4 v0 = *reinterpret_cast<const float4*>(go_t + (long long)r0 * C + c);
        const float4 v1 = *reinterpret_cast<const float4*>(go_t + (long long)r1 * C + c);
        acc.x += w0 * v0.x; acc.y += w0 * v0.y; acc.z += w0 * v0.z; acc.w += w0 * v0.w;
        acc.x += w1 * v1.x; acc.y += w1 * v1.y; acc.z += w1 * v1.z; acc.w += w1 * v1.w;
      }
      if (e < e1) {
        const int r0 = ent_row[e];
        const float w0 = ent_w[e];
        const float4 v0 = *reinterpret_cast<const float4*>(go_t + (long long)r0 * C + c);
        acc.x += w0 * v0.x; acc.y += w0 * v0.y; acc.z += w0 * v0.z; acc.w += w0 * v0.w;
      }
      *reinterpret_cast<float4*>(grad_nhwc + pix * C + c) = acc;
    }
  } else {
    for (int c = lane; c < C; c += 64) {
      float acc = 0.f;
      for (int e = e0; e < e1; ++e) acc += ent_w[e] * go_t[(long long)ent_row[e] * C + c];
      grad_nhwc[pix * C + c] = acc;
    }
  }
}

// The same gather with the result written in NCHW -- what the convolution before the RoI head wants -- so that the
// caller needs neither a (N,H,W,C) -> (N,C,H,W) transpose of the whole feature gradient (128 MB each way for a
// 2 x 256 x 256 x 256 level) nor a pre-zeroed output.
//   reads   lanes = channels: a gradient row is read in coalesced 256-byte pieces (1-2 cache lines per instruction)
//   writes  lanes = pixels: a workgroup owns 64 consecutive pixels and writes whole 256-byte runs of a channel plane
//   between a 64 pixel x 64 channel LDS tile
// A pixel's entries {row, weight} are fetched by one coalesced load (lane = entry) and broadcast by shuffles.
// NOT the default yet (ops/roi_align_rotated_v1.py keeps the channels-last gather + transposes): measured 321 us at a
// 2 x 256 x 256 x 256 level with 512 RoIs, of which 136 us remain with the row reads removed and 328 us with the
// stores removed -- it is the 64 serial (pixel, chunk) steps of a wave, one global round trip each for the entries,
// that cost the time, not the stores.  Other forms measured: serial entry walk 250 us; lanes = pixels reading 64 B of
// a row each, no LDS, 269 us (64 cache lines per load instruction); 16-pixel tiles with 64-byte writes per lane
// 377 us.  Next: fetch the entries of all 16 pixels of a wave in one batch and keep four channel chunks in flight.
constexpr int RG_PIX = 64;
__global__ __launch_bounds__(256) void rroi_gather_nchw_kernel(const float* __restrict__ go_t, const int* __restrict__ start,
                                                               const int* __restrict__ ent_row,
                                                               const float* __restrict__ ent_w, long long npix, int C,
                                                               int HW, float* __restrict__ grad_nchw) {
  __shared__ float s_tile[RG_PIX][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p0 = (long long)rsdet_xcd_contiguous(blockIdx.x, gridDim.x) * RG_PIX;
  const long long pw = p0 + lane;  // this lane's pixel in the write phase
  const long long n_w = pw / HW, hw_w = pw - n_w * HW;
  // the entry ranges of this wave's 16 pixels: one coalesced load (lanes 0..16 -> start[p .. p+16])
  const long long pb = p0 + wave * 16;
  const int my_start = (lane <= 16 && pb + lane <= npix) ? start[pb + lane] : 0;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    for (int k = 0; k < 16; ++k) {
      float acc = 0.f;
      if (pb + k < npix) {
        const int e0 = __shfl(my_start, k), e1 = __shfl(my_start, k + 1);
        for (int eb = e0; eb < e1; eb += 64) {
          const int ne = min(64, e1 - eb);
          int row = 0;
          float w = 0.f;
          if (lane < ne) {
            row = ent_row[eb + lane];
            w = ent_w[eb + lane];
          }
          for (int i = 0; i < ne; ++i) {  // independent row reads: addresses come from registers
            const int r = __shfl(row, i);
            const float wi = __shfl(w, i);
            if (c < C) acc += wi * go_t[(long long)r * C + c];
          }
        }
      }
      s_tile[wave * 16 + k][lane] = acc;
    }
    __syncthreads();
    if (pw < npix) {
      for (int k = 0; k < 16; ++k) {
        const int ch = wave * 16 + k;
        if (c0 + ch < C) grad_nchw[(n_w * C + c0 + ch) * HW + hw_w] = s_tile[lane][ch];
      }
    }
    __syncthreads();
  }
}

// Round 5 form of the NCHW-writing gather (C % 4 == 0; the default of ops/roi_align_rotated_v1.py): what the note above
// asked for.  Measured at the same 2 x 256 x 256 x 256 level, 512 RoIs: the CALL (index build + gradient-row turn + gather)
// 146.7 us against 168.8 us for the channels-last gather + the layout turn of its result, 321 us for the round-2 NCHW form
// below.  History of this kernel (profiles/r05_q_rroi_bwd_kernels.txt, r05_rroi_uncond.txt): first built with the row loads
// of a batch under `if (i + j < ne)` -- 222.8 us for the call whether 4 or 16 rows per batch, because a load under a
// (uniform) branch is followed by a wait for it: the "batch" was serial; v_readlane instead of the LDS crossbar for the
// wave-uniform lane reads: 207.6; the loads made UNCONDITIONAL (clamped entry index, unused values never added): 146.7.
// The index build (count 15 + fill 18 + scan 13 + fills 10 us) is now 40 % of the call, which stands at 0.136 of the
// HBM roofline (asked: 0.25).
//   * a wave owns 16 consecutive pixels, whose entries are ONE contiguous range of the CSR arrays: it walks that range
//     flat, 64 entries per coalesced fetch (lane = entry), RG2_DEPTH gradient rows in flight (addresses from registers), and
//     flushes the accumulator to LDS whenever the walk crosses a pixel boundary -- no per-pixel round trip for the
//     entries, no serial (pixel, chunk) steps; a pixel's terms are still added in entry order;
//   * lanes = 4 channels each (256 channels per pass): a gradient row is one 1 KB wave-wide load, a flush one
//     conflict-free 16-byte LDS write per lane (pitch 260 floats);
//   * write-out: lane = pixel; a 16-byte LDS read gives 4 channels of the lane's pixel (pitch 260: eight lanes cover
//     the 32 banks), each stored as part of a 256-byte run of its channel plane.
// The caller needs neither the (N,H,W,C) -> (N,C,H,W) turn of the whole feature gradient (128 MB each way at a
// 2 x 256 x 256 x 256 level) nor a pre-zeroed output.
constexpr int RG2_PITCH = 260;
#ifndef RG2_DEPTH
#define RG2_DEPTH 16      // gradient rows in flight per wave (16 KB)
#endif
// lane `idx` of v for a WAVE-UNIFORM idx: v_readlane (a few cycles) instead of the LDS crossbar of __shfl (~100 cycles of
// dependent latency per entry in the walk below)
__device__ __forceinline__ int rg2_lane(int v, int idx) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(idx));
}
__global__ __launch_bounds__(256) void rroi_gather_nchw_tile_kernel(const float* __restrict__ go_t,
                                                                    const int* __restrict__ start,
                                                                    const int* __restrict__ ent_row,
                                                                    const float* __restrict__ ent_w, long long npix, int C,
                                                                    int HW, float* __restrict__ grad_nchw) {
  __shared__ __attribute__((aligned(16))) float s_tile[RG_PIX * RG2_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p0 = (long long)rsdet_xcd_contiguous(blockIdx.x, gridDim.x) * RG_PIX;
  const long long pw = p0 + lane;  // this lane's pixel in the write phase
  const long long n_w = pw / HW, hw_w = pw - n_w * HW;
  const long long pb = p0 + wave * 16;
  // the 17 entry boundaries of this wave's 16 pixels: one coalesced load (pixels past the end: empty)
  const int my_start = lane <= 16 ? start[min(pb + lane, npix)] : 0;
  const int e_lo = rg2_lane(my_start, 0), e_hi = rg2_lane(my_start, 16);
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool c_ok = c < C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;                                   // the pixel the walk is in
    int bound = rg2_lane(my_start, 1);           // first entry of pixel k + 1
    float* my_rows = s_tile + (wave * 16) * RG2_PITCH + lane * 4;
    for (int eb = e_lo; eb < e_hi; eb += 64) {
      const int ne = min(64, e_hi - eb);
      int row = 0;
      float w = 0.f;
      if (lane < ne) {
        row = ent_row[eb + lane];
        w = ent_w[eb + lane];
      }
      for (int i = 0; i < ne; i += RG2_DEPTH) {
        float4 v[RG2_DEPTH];
        float wi[RG2_DEPTH];
#pragma unroll
        for (int j = 0; j < RG2_DEPTH; ++j) {
          // UNCONDITIONAL loads (entry index and channel clamped to valid ones; the unused values are never added): a
          // load under a branch is followed by a wait for it, which had serialised the whole batch
          const int ej = min(i + j, ne - 1);
          const int r = rg2_lane(row, ej);
          wi[j] = __int_as_float(rg2_lane(__float_as_int(w), ej));
          v[j] = *reinterpret_cast<const float4*>(go_t + (long long)r * C + (c_ok ? c : 0));
        }
#pragma unroll
        for (int j = 0; j < RG2_DEPTH; ++j) {
          if (i + j < ne) {                      // (wave-uniform)
            const int e = eb + i + j;
            while (e >= bound) {                 // the walk leaves pixel k: its sum is complete
              *reinterpret_cast<float4*>(my_rows + k * RG2_PITCH) = acc;
              acc = make_float4(0.f, 0.f, 0.f, 0.f);
              ++k;
              bound = rg2_lane(my_start, min(k + 1, 16));
            }
            acc.x += wi[j] * v[j].x, acc.y += wi[j] * v[j].y, acc.z += wi[j] * v[j].z, acc.w += wi[j] * v[j].w;
          }
        }
      }
    }
    for (; k < 16; ++k) {                        // the last pixel with entries, then the empty ones behind it
      *reinterpret_cast<float4*>(my_rows + k * RG2_PITCH) = acc;
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (pw < npix) {
      const float* mine = s_tile + lane * RG2_PITCH + wave * 64;
      float* dst = grad_nchw + (n_w * C + c0 + wave * 64) * HW + hw_w;
#pragma unroll 4
      for (int cc = 0; cc < 64; cc += 4) {
        const float4 v = *reinterpret_cast<const float4*>(mine + cc);
        const int ch = c0 + wave * 64 + cc;
        if (ch + 3 < C) {
          dst[(long long)(cc + 0) * HW] = v.x, dst[(long long)(cc + 1) * HW] = v.y;
          dst[(long long)(cc + 2) * HW] = v.z, dst[(long long)(cc + 3) * HW] = v.w;
        } else {
          if (ch + 0 < C) dst[(long long)(cc + 0) * HW] = v.x;
          if (ch + 1 < C) dst[(long long)(cc + 1) * HW] = v.y;
          if (ch + 2 < C) dst[(long long)(cc + 2) * HW] = v.z;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace rsdet

using namespace rsdet;

static int rroi_check(int R, int C, int H, int W, int PH, int PW) {
  if (R < 0 || C < 0 || H < 1 || W < 1 || PH < 1 || PW < 1) return RSDET_EINVAL;
  if ((long long)C * PH * PW > 0x7fffffffLL) return RSDET_EINVAL;
  return RSDET_OK;
}

static int rroi_forward(const float* feat, const float* rois, int R, int C, int H, int W, int PH, int PW,
                        float spatial_scale, int sample_num, int v0, float* out, void* stream) {
  int rc = rroi_check(R, C, H, W, PH, PW);
  if (rc) return rc;
  if (R == 0 || C == 0) return RSDET_OK;
  if (!feat || !rois || !out) return RSDET_EINVAL;
  int per_roi = (C * PH * PW + RROI_NT - 1) / RROI_NT;
  int gy = per_roi < 8 ? per_roi : 8;  // >= 8 workgroups per RoI keeps small R busy
  hipLaunchKernelGGL(rroi_forward_kernel, dim3(R, gy), dim3(RROI_NT), 0, (hipStream_t)stream, feat,
                     rois, C, H, W, PH, PW, spatial_scale, sample_num, v0, out);
  return rsdet_launch_status();
}

static int rroi_forward_levels(const rsdet_rroi_levels* d, const float* rois, const int* lvl, int R, int C, int PH, int PW,
                               int sample_num, int v0, float* out, void* stream) {
  if (!d || d->n_levels < 1 || d->n_levels > RSDET_RROI_MAX_LEVELS) return RSDET_EINVAL;
  RroiLevels lv;
  for (int l = 0; l < d->n_levels; ++l) {
    int rc = rroi_check(R, C, d->H[l], d->W[l], PH, PW);
    if (rc) return rc;
    if (!d->feat[l] || !(d->scale[l] > 0.f)) return RSDET_EINVAL;
    lv.feat[l] = d->feat[l], lv.H[l] = d->H[l], lv.W[l] = d->W[l], lv.scale[l] = d->scale[l];
  }
  if (R == 0 || C == 0) return RSDET_OK;
  if (!rois || !lvl || !out) return RSDET_EINVAL;
  int per_roi = (C * PH * PW + RROI_NT - 1) / RROI_NT;
  int gy = per_roi < 8 ? per_roi : 8;
  hipLaunchKernelGGL(rroi_forward_levels_kernel, dim3(R, gy), dim3(RROI_NT), 0, (hipStream_t)stream, lv, rois, lvl,
                     d->n_levels, C, PH, PW, sample_num, v0, out);
  return rsdet_launch_status();
}

extern "C" int rsdet_rroi_align_v1_forward_levels_f32(const rsdet_rroi_levels* levels, const float* rois, const int32_t* lvl,
                                                      int R, int C, int PH, int PW, int sample_num, float* out,
                                                      void* stream) {
  return rroi_forward_levels(levels, rois, lvl, R, C, PH, PW, sample_num, 0, out, stream);
}

extern "C" int rsdet_rroi_align_v0_forward_levels_f32(const rsdet_rroi_levels* levels, const float* rois, const int32_t* lvl,
                                                      int R, int C, int PH, int PW, int sample_num, float* out,
                                                      void* stream) {
  return rroi_forward_levels(levels, rois, lvl, R, C, PH, PW, sample_num, 1, out, stream);
}

static int rroi_backward(const float* grad_out, const float* rois, int R, int C, int H, int W, int PH, int PW,
                         float spatial_scale, int sample_num, int v0, float* grad_feat, void* stream) {
  int rc = rroi_check(R, C, H, W, PH, PW);
  if (rc) return rc;
  if (R == 0 || C == 0) return RSDET_OK;
  if (!grad_out || !rois || !grad_feat) return RSDET_EINVAL;
  int per_roi = (C * PH * PW + RROI_NT - 1) / RROI_NT;
  int gy = per_roi < 8 ? per_roi : 8;
  hipLaunchKernelGGL(rroi_backward_kernel, dim3(R, gy), dim3(RROI_NT), 0, (hipStream_t)stream,
                     grad_out, rois, C, H, W, PH, PW, spatial_scale, sample_num, v0, grad_feat);
  return rsdet_launch_status();
}

void rsdet_launch_pixel_gather(const float* rows, const int* start, const int* ent_row, const float* ent_w,
                               long long npix, int C, float* out_nhwc, hipStream_t s) {
  const bool vec4 = (C % 4 == 0) && (((uintptr_t)rows | (uintptr_t)out_nhwc) % 16 == 0);
  const unsigned gb = (unsigned)((npix + 3) / 4);
  if (vec4)
    hipLaunchKernelGGL(rroi_gather_kernel<true>, dim3(gb), dim3(256), 0, s, rows, start, ent_row, ent_w, npix, C,
                       out_nhwc);
  else
    hipLaunchKernelGGL(rroi_gather_kernel<false>, dim3(gb), dim3(256), 0, s, rows, start, ent_row, ent_w, npix, C,
                       out_nhwc);
}

static void launch_pixel_gather_nchw(const float* rows, const int* start, const int* ent_row, const float* ent_w,
                                     long long npix, int C, int HW, float* out_nchw, hipStream_t s) {
  const dim3 grid((unsigned)((npix + RG_PIX - 1) / RG_PIX));
  if (C % 4 == 0 && (((uintptr_t)rows) & 15) == 0)
    hipLaunchKernelGGL(rroi_gather_nchw_tile_kernel, grid, dim3(256), 0, s, rows, start, ent_row, ent_w, npix, C, HW,
                       out_nchw);
  else
    hipLaunchKernelGGL(rroi_gather_nchw_kernel, grid, dim3(256), 0, s, rows, start, ent_row, ent_w, npix, C, HW,
                       out_nchw);
}

static inline size_t rroi_align256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t rsdet_rroi_align_v1_backward_gather_ws_size(int R, int PH, int PW, int sample_num, int N, int H,
                                                              int W) {
  if (R <= 0 || PH < 1 || PW < 1 || sample_num < 1 || N < 1 || H < 1 || W < 1) return 0;
  const size_t npix = (size_t)N * H * W, ent = (size_t)R * PH * PW * sample_num * sample_num * 4;
  return rroi_align256((npix + 1) * 4) * 2 + rroi_align256(ent * 4) * 2 + rroi_align256((npix / 4096 + 1) * 4);
}

struct RroiWs {
  int *cnt, *start, *ent_row, *chunk_sum;
  float* ent_w;
};
static RroiWs rroi_ws(void* ws, long long npix, long long items) {
  char* w = (char*)ws;
  RroiWs o;
  o.cnt = (int*)w;
  o.start = (int*)(w + rroi_align256((npix + 1) * 4));
  o.ent_row = (int*)(w + rroi_align256((npix + 1) * 4) * 2);
  o.ent_w = (float*)(w + rroi_align256((npix + 1) * 4) * 2 + rroi_align256((size_t)items * 16));
  o.chunk_sum = (int*)(w + rroi_align256((npix + 1) * 4) * 2 + rroi_align256((size_t)items * 16) * 2);
  return o;
}

static int rroi_gather_args(int R, int C, int N, int H, int W, int PH, int PW, int sample_num, const void* ws,
                            size_t ws_bytes, long long* items) {
  int rc = rroi_check(R, C, H, W, PH, PW);
  if (rc) return rc;
  if (sample_num < 1 || N < 1) return RSDET_EINVAL;  // adaptive sampling (sample_num <= 0): use the scatter form
  *items = (long long)R * PH * PW * sample_num * sample_num;
  if (*items * 4 > 0x7fffffffLL) return RSDET_EINVAL;
  if (R > 0 && (!ws || ((uintptr_t)ws & 15) ||
                ws_bytes < rsdet_rroi_align_v1_backward_gather_ws_size(R, PH, PW, sample_num, N, H, W)))
    return RSDET_EINVAL;
  return RSDET_OK;
}

// The inverted index (pixel -> the (RoI, bin) rows that touch it, with their weights) of the gather-form backward.  It
// depends on the RoIs and the geometry only -- not on the gradient -- so the caller may build it at FORWARD time, on a
// side stream beside the forward kernel (ops/roi_align_rotated_v1.py does), and hand the workspace to the gather later.
static int rroi_backward_index(const float* rois, int R, int N, int H, int W, int PH, int PW, float spatial_scale,
                               int sample_num, int v0, void* ws, size_t ws_bytes, void* stream) {
  long long items;
  int rc = rroi_gather_args(R, 1, N, H, W, PH, PW, sample_num, ws, ws_bytes, &items);
  if (rc) return rc;
  if (R == 0) return RSDET_OK;
  if (!rois) return RSDET_EINVAL;
  const long long npix = (long long)N * H * W;
  hipStream_t s = (hipStream_t)stream;
  const RroiWs o = rroi_ws(ws, npix, items);
  if (hipMemsetAsync(o.cnt, 0, (size_t)(npix + 1) * 4, s) != hipSuccess) return RSDET_ELAUNCH;
  const unsigned ib = (unsigned)((items + 255) / 256);
  hipLaunchKernelGGL(rroi_idx_count_kernel, dim3(ib), dim3(256), 0, s, rois, items, H, W, PH, PW, spatial_scale,
                     sample_num, v0, npix, o.cnt);
  rsdet_launch_index_scan(o.cnt, npix, o.chunk_sum, o.start, s);
  hipLaunchKernelGGL(rroi_idx_fill_kernel, dim3(ib), dim3(256), 0, s, rois, items, H, W, PH, PW, spatial_scale,
                     sample_num, v0, npix, o.start, o.cnt, o.ent_row, o.ent_w);
  return rsdet_launch_status();
}

static int rroi_backward_gather_indexed(const float* grad_out_t, int R, int C, int N, int H, int W, int PH, int PW,
                                        int sample_num, float* grad_feat, const void* ws, size_t ws_bytes, void* stream,
                                        bool nchw) {
  long long items;
  int rc = rroi_gather_args(R, C, N, H, W, PH, PW, sample_num, ws, ws_bytes, &items);
  if (rc) return rc;
  const long long npix = (long long)N * H * W;
  if (C == 0) return RSDET_OK;
  if (!grad_feat || (R > 0 && !grad_out_t)) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (R == 0) return hipMemsetAsync(grad_feat, 0, (size_t)npix * C * 4, s) == hipSuccess ? RSDET_OK : RSDET_ELAUNCH;
  const RroiWs o = rroi_ws(const_cast<void*>(ws), npix, items);
  if (nchw)
    launch_pixel_gather_nchw(grad_out_t, o.start, o.ent_row, o.ent_w, npix, C, H * W, grad_feat, s);
  else
    rsdet_launch_pixel_gather(grad_out_t, o.start, o.ent_row, o.ent_w, npix, C, grad_feat, s);
  return rsdet_launch_status();
}

static int rroi_backward_gather(const float* grad_out_t, const float* rois, int R, int C, int N, int H, int W, int PH,
                                int PW, float spatial_scale, int sample_num, int v0, float* grad_feat_nhwc, void* ws,
                                size_t ws_bytes, void* stream, bool nchw = false) {
  if (C == 0) return rroi_check(R, C, H, W, PH, PW);
  int rc = rroi_backward_index(rois, R, N, H, W, PH, PW, spatial_scale, sample_num, v0, ws, ws_bytes, stream);
  if (rc) return rc;
  return rroi_backward_gather_indexed(grad_out_t, R, C, N, H, W, PH, PW, sample_num, grad_feat_nhwc, ws, ws_bytes, stream,
                                      nchw);
}

// ---- backward of the one-launch extractor forward: one index over the levels, one NCHW gather per level ----
static long long rroi_levels_npix(const rsdet_rroi_levels* d, int N, RroiLevelGeom* g) {
  long long pix = 0;
  g->n = d->n_levels;
  for (int l = 0; l < d->n_levels; ++l) {
    g->H[l] = d->H[l], g->W[l] = d->W[l], g->scale[l] = d->scale[l];
    g->pix_base[l] = pix;
    pix += (long long)N * d->H[l] * d->W[l];
  }
  g->pix_base[d->n_levels] = pix;
  return pix;
}

extern "C" size_t rsdet_rroi_align_backward_levels_ws_size(const rsdet_rroi_levels* d, int R, int PH, int PW, int sample_num,
                                                           int N) {
  if (!d || d->n_levels < 1 || d->n_levels > RSDET_RROI_MAX_LEVELS || R <= 0 || PH < 1 || PW < 1 || sample_num < 1 || N < 1)
    return 0;
  RroiLevelGeom g;
  const size_t npix = (size_t)rroi_levels_npix(d, N, &g), ent = (size_t)R * PH * PW * sample_num * sample_num * 4;
  return rroi_align256((npix + 1) * 4) * 2 + rroi_align256(ent * 4) * 2 + rroi_align256((npix / 4096 + 1) * 4);
}

static int rroi_backward_levels(const rsdet_rroi_levels* d, float* const* grad_feat, const float* grad_out_t,
                                const float* rois, const int* lvl, int R, int C, int N, int PH, int PW, int sample_num,
                                int v0, void* ws, size_t ws_bytes, void* stream) {
  if (!d || d->n_levels < 1 || d->n_levels > RSDET_RROI_MAX_LEVELS || !grad_feat || sample_num < 1 || N < 1 || R < 1 ||
      C < 1 || (C & 3))
    return RSDET_EINVAL;
  for (int l = 0; l < d->n_levels; ++l) {
    int rc = rroi_check(R, C, d->H[l], d->W[l], PH, PW);
    if (rc) return rc;
    if (!(d->scale[l] > 0.f)) return RSDET_EINVAL;
  }
  if (!grad_out_t || !rois || !lvl) return RSDET_EINVAL;
  RroiLevelGeom g;
  const long long npix = rroi_levels_npix(d, N, &g), items = (long long)R * PH * PW * sample_num * sample_num;
  if (items * 4 > 0x7fffffffLL || npix > 0x7fffffffLL) return RSDET_EINVAL;
  if (!ws || ((uintptr_t)ws & 15) || ws_bytes < rsdet_rroi_align_backward_levels_ws_size(d, R, PH, PW, sample_num, N))
    return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const RroiWs o = rroi_ws(ws, npix, items);
  if (hipMemsetAsync(o.cnt, 0, (size_t)(npix + 1) * 4, s) != hipSuccess) return RSDET_ELAUNCH;
  const unsigned ib = (unsigned)((items + 255) / 256);
  hipLaunchKernelGGL(rroi_idx_count_levels_kernel, dim3(ib), dim3(256), 0, s, g, rois, lvl, items, PH, PW, sample_num, v0,
                     o.cnt);
  rsdet_launch_index_scan(o.cnt, npix, o.chunk_sum, o.start, s);
  hipLaunchKernelGGL(rroi_idx_fill_levels_kernel, dim3(ib), dim3(256), 0, s, g, rois, lvl, items, PH, PW, sample_num, v0,
                     o.start, o.cnt, o.ent_row, o.ent_w);
  for (int l = 0; l < d->n_levels; ++l)
    if (grad_feat[l])
      launch_pixel_gather_nchw(grad_out_t, o.start + g.pix_base[l], o.ent_row, o.ent_w, (long long)N * g.H[l] * g.W[l], C,
                               g.H[l] * g.W[l], grad_feat[l], s);
  return rsdet_launch_status();
}

extern "C" int rsdet_rroi_align_v1_backward_levels_nchw_f32(const rsdet_rroi_levels* levels, float* const* grad_feat,
                                                            const float* grad_out_t, const float* rois, const int32_t* lvl,
                                                            int R, int C, int N, int PH, int PW, int sample_num, void* ws,
                                                            size_t ws_bytes, void* stream) {
  return rroi_backward_levels(levels, grad_feat, grad_out_t, rois, lvl, R, C, N, PH, PW, sample_num, 0, ws, ws_bytes, stream);
}

extern "C" int rsdet_rroi_align_v0_backward_levels_nchw_f32(const rsdet_rroi_levels* levels, float* const* grad_feat,
                                                            const float* grad_out_t, const float* rois, const int32_t* lvl,
                                                            int R, int C, int N, int PH, int PW, int sample_num, void* ws,
                                                            size_t ws_bytes, void* stream) {
  return rroi_backward_levels(levels, grad_feat, grad_out_t, rois, lvl, R, C, N, PH, PW, sample_num, 1, ws, ws_bytes, stream);
}

extern "C" int rsdet_rroi_align_backward_gather_indexed_f32(const float* grad_out_t, int R, int C, int N, int H, int W,
                                                            int PH, int PW, int sample_num, int nchw, float* grad_feat,
                                                            const void* ws, size_t ws_bytes, void* stream) {
  return rroi_backward_gather_indexed(grad_out_t, R, C, N, H, W, PH, PW, sample_num, grad_feat, ws, ws_bytes, stream,
                                      nchw != 0);
}

#define RSDET_RROI_ENTRY(tag, v0)                                                                                     \
  extern "C" int rsdet_rroi_align_##tag##_forward_f32(const float* feat, const float* rois, int R, int C, int H,      \
                                                      int W, int PH, int PW, float spatial_scale, int sample_num,    \
                                                      float* out, void* stream) {                                    \
    return rroi_forward(feat, rois, R, C, H, W, PH, PW, spatial_scale, sample_num, v0, out, stream);                 \
  }                                                                                                                   \
  extern "C" int rsdet_rroi_align_##tag##_backward_f32(const float* grad_out, const float* rois, int R, int C,        \
                                                       int H, int W, int PH, int PW, float spatial_scale,            \
                                                       int sample_num, float* grad_feat, void* stream) {             \
    return rroi_backward(grad_out, rois, R, C, H, W, PH, PW, spatial_scale, sample_num, v0, grad_feat, stream);      \
  }                                                                                                                   \
  extern "C" int rsdet_rroi_align_##tag##_backward_gather_f32(                                                        \
      const float* grad_out_t, const float* rois, int R, int C, int N, int H, int W, int PH, int PW,                  \
      float spatial_scale, int sample_num, float* grad_feat_nhwc, void* ws, size_t ws_bytes, void* stream) {         \
    return rroi_backward_gather(grad_out_t, rois, R, C, N, H, W, PH, PW, spatial_scale, sample_num, v0,              \
                                grad_feat_nhwc, ws, ws_bytes, stream);                                               \
  }                                                                                                                   \
  extern "C" int rsdet_rroi_align_##tag##_backward_index_f32(const float* rois, int R, int N, int H, int W, int PH,  \
                                                             int PW, float spatial_scale, int sample_num, void* ws, \
                                                             size_t ws_bytes, void* stream) {                       \
    return rroi_backward_index(rois, R, N, H, W, PH, PW, spatial_scale, sample_num, v0, ws, ws_bytes, stream);      \
  }                                                                                                                   \
  extern "C" int rsdet_rroi_align_##tag##_backward_gather_nchw_f32(                                                   \
      const float* grad_out_t, const float* rois, int R, int C, int N, int H, int W, int PH, int PW,                  \
      float spatial_scale, int sample_num, float* grad_feat_nchw, void* ws, size_t ws_bytes, void* stream) {         \
    return rroi_backward_gather(grad_out_t, rois, R, C, N, H, W, PH, PW, spatial_scale, sample_num, v0,              \
                                grad_feat_nchw, ws, ws_bytes, stream, true);                                         \
  }
RSDET_RROI_ENTRY(v1, 0)
RSDET_RROI_ENTRY(v0, 1)
