// deform_conv.hip -- deformable convolution v1 pieces (AlignConv) for gfx950.
//
// Replaces: deformable_im2col / deformable_col2im / deformable_col2im_coord
//   /root/reference/python/jdet/ops/dcn_v1.py:309-410, kernels :132-306,
//   bilinear :25-56, gradient weights :58-129.
//
// The reference gives one thread to every (channel, position) and so re-reads
// the offsets and re-derives the bilinear footprint C (=256) times.  Here a
// thread owns one (tap, position): it reads its two offsets once, builds the
// 4-corner footprint once and then streams over the channels -- per channel 4
// gathers that neighbouring lanes (neighbouring positions) serve from the same
// cache lines, and one fully coalesced store (im2col) / load (col2im) of the
// column row.  HBM traffic is then the algorithmic minimum: the column matrix
// once, the image once (from L2 after first touch), the offsets once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"

namespace rsdet {

struct Geom {
  int C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, B, dg, Ho, Wo;
};

struct Foot {  // bilinear footprint of one sampling point
  int o1, o2, o3, o4;      // element offsets inside an (H, W) plane, -1 = outside
  float w1, w2, w3, w4;
};

// dcn_v1.py:25-56 + the validity window of :170
__device__ __forceinline__ Foot im2col_foot(float h, float w, int H, int W) {
  Foot f{-1, -1, -1, -1, 0.f, 0.f, 0.f, 0.f};
  if (!(h > -1 && w > -1 && h < H && w < W)) return f;
  int hl = (int)floorf(h), wl = (int)floorf(w);
  int hh = hl + 1, wh = wl + 1;
  float lh = h - hl, lw = w - wl;
  float uh = 1 - lh, uw = 1 - lw;
  if (hl >= 0 && wl >= 0) f.o1 = hl * W + wl;
  if (hl >= 0 && wh <= W - 1) f.o2 = hl * W + wh;
  if (hh <= H - 1 && wl >= 0) f.o3 = hh * W + wl;
  if (hh <= H - 1 && wh <= W - 1) f.o4 = hh * W + wh;
  f.w1 = uh * uw;
  f.w2 = uh * lw;
  f.w3 = lh * uw;
  f.w4 = lh * lw;
  return f;
}

constexpr int DCN_NT = 256;

// grid: (ceil(B*Ho*Wo / NT), kh*kw, channel_chunks)
__global__ __launch_bounds__(DCN_NT) void deform_im2col_kernel(const float* __restrict__ im,
                                                               const float* __restrict__ offset,
                                                               Geom g, int c_chunk,
                                                               float* __restrict__ col) {
  const long long plane = (long long)g.Ho * g.Wo;
  const long long npos = (long long)g.B * plane;
  long long pos = (long long)blockIdx.x * DCN_NT + threadIdx.x;
  if (pos >= npos) return;
  const int tap = blockIdx.y;
  const int i = tap / g.kw, j = tap - i * g.kw;
  int b = (int)(pos / plane);
  int hw = (int)(pos - (long long)b * plane);
  int ho = hw / g.Wo, wo = hw - ho * g.Wo;
  const int cpg = g.C / g.dg;
  const int c0 = blockIdx.z * c_chunk, c1 = min(g.C, c0 + c_chunk);
  const long long HW = (long long)g.H * g.W;
  int cur_grp = -1;
  Foot f{};
  for (int c = c0; c < c1; ++c) {
    int grp = c / cpg;
    if (grp != cur_grp) {  // once per deformable group (once in S2ANet: dg = 1)
      cur_grp = grp;
      const float* offp = offset + ((long long)b * g.dg + grp) * 2 * g.kh * g.kw * plane;
      float oh = offp[(long long)(2 * tap) * plane + hw];
      float ow = offp[(long long)(2 * tap + 1) * plane + hw];
      float h_im = (ho * g.sh - g.ph) + i * g.dh + oh;
      float w_im = (wo * g.sw - g.pw) + j * g.dw + ow;
      f = im2col_foot(h_im, w_im, g.H, g.W);
    }
    const float* imp = im + ((long long)b * g.C + c) * HW;
    float v1 = f.o1 >= 0 ? imp[f.o1] : 0.f;
    float v2 = f.o2 >= 0 ? imp[f.o2] : 0.f;
    float v3 = f.o3 >= 0 ? imp[f.o3] : 0.f;
    float v4 = f.o4 >= 0 ? imp[f.o4] : 0.f;
    float val = f.w1 * v1 + f.w2 * v2 + f.w3 * v3 + f.w4 * v4;
    col[((long long)(c * g.kh * g.kw + tap)) * npos + pos] = val;
  }
}

// dcn_v1.py:58-84 evaluated at the (<= 4) pixels the reference's 5x5 window admits:
// |h - y| < 1 and |w - x| < 1 leave y in {floor(h), floor(h)+1}, x likewise.
__device__ __forceinline__ Foot col2im_foot(float h, float w, int H, int W) {
  Foot f{-1, -1, -1, -1, 0.f, 0.f, 0.f, 0.f};
  if (h <= -1 || h >= H || w <= -1 || w >= W) return f;  // get_gradient_weight early return
  int hl = (int)floorf(h), wl = (int)floorf(w);
  int hh = hl + 1, wh = wl + 1;
  bool yl = hl >= 0 && hl < H && fabsf(h - hl) < 1, yh = hh >= 0 && hh < H && fabsf(h - hh) < 1;
  bool xl = wl >= 0 && wl < W && fabsf(w - wl) < 1, xh = wh >= 0 && wh < W && fabsf(w - wh) < 1;
  if (yl && xl) { f.o1 = hl * W + wl; f.w1 = (hl + 1 - h) * (wl + 1 - w); }
  if (yl && xh) { f.o2 = hl * W + wh; f.w2 = (hl + 1 - h) * (w + 1 - wh); }
  if (yh && xl) { f.o3 = hh * W + wl; f.w3 = (h + 1 - hh) * (wl + 1 - w); }
  if (yh && xh) { f.o4 = hh * W + wh; f.w4 = (h + 1 - hh) * (w + 1 - wh); }
  return f;
}

__global__ __launch_bounds__(DCN_NT) void deform_col2im_kernel(const float* __restrict__ col,
                                                               const float* __restrict__ offset,
                                                               Geom g, int c_chunk,
                                                               float* __restrict__ grad_im) {
  const long long plane = (long long)g.Ho * g.Wo;
  const long long npos = (long long)g.B * plane;
  long long pos = (long long)blockIdx.x * DCN_NT + threadIdx.x;
  if (pos >= npos) return;
  const int tap = blockIdx.y;
  const int i = tap / g.kw, j = tap - i * g.kw;
  int b = (int)(pos / plane);
  int hw = (int)(pos - (long long)b * plane);
  int ho = hw / g.Wo, wo = hw - ho * g.Wo;
  const int cpg = g.C / g.dg;
  const int c0 = blockIdx.z * c_chunk, c1 = min(g.C, c0 + c_chunk);
  const long long HW = (long long)g.H * g.W;
  int cur_grp = -1;
  Foot f{};
  for (int c = c0; c < c1; ++c) {
    int grp = c / cpg;
    if (grp != cur_grp) {
      cur_grp = grp;
      const float* offp = offset + ((long long)b * g.dg + grp) * 2 * g.kh * g.kw * plane;
      float oh = offp[(long long)(2 * tap) * plane + hw];
      float ow = offp[(long long)(2 * tap + 1) * plane + hw];
      float fh = (ho * g.sh - g.ph) + i * g.dh + oh;
      float fw = (wo * g.sw - g.pw) + j * g.dw + ow;
      f = col2im_foot(fh, fw, g.H, g.W);
    }
    float top = col[((long long)(c * g.kh * g.kw + tap)) * npos + pos];
    float* gp = grad_im + ((long long)b * g.C + c) * HW;
    if (f.o1 >= 0) atomicAdd(gp + f.o1, f.w1 * top);
    if (f.o2 >= 0) atomicAdd(gp + f.o2, f.w2 * top);
    if (f.o3 >= 0) atomicAdd(gp + f.o3, f.w3 * top);
    if (f.o4 >= 0) atomicAdd(gp + f.o4, f.w4 * top);
  }
}

// dcn_v1.py:86-129 + :244-306.  One thread per (b, offset channel, ho, wo); the
// channel loop accumulates in ascending channel order like the reference.
__global__ __launch_bounds__(DCN_NT) void deform_col2im_coord_kernel(
    const float* __restrict__ col, const float* __restrict__ im, const float* __restrict__ offset,
    Geom g, float* __restrict__ grad_offset) {
  const long long plane = (long long)g.Ho * g.Wo;
  const long long npos = (long long)g.B * plane;
  const int offc = 2 * g.kh * g.kw * g.dg;
  long long idx = (long long)blockIdx.x * DCN_NT + threadIdx.x;
  if (idx >= (long long)g.B * offc * plane) return;
  int hw = (int)(idx % plane);
  int oc = (int)((idx / plane) % offc);
  int b = (int)(idx / plane / offc);
  int ho = hw / g.Wo, wo = hw - ho * g.Wo;
  const int taps = g.kh * g.kw;
  int grp = oc / (2 * taps);
  int lc = oc - grp * 2 * taps;
  int tap = lc / 2, dir = lc & 1;
  int i = tap / g.kw, j = tap - i * g.kw;
  const int cpg = g.C / g.dg;
  const long long HW = (long long)g.H * g.W;
  const float* offp = offset + ((long long)b * g.dg + grp) * 2 * taps * plane;
  float oh = offp[(long long)(2 * tap) * plane + hw];
  float ow = offp[(long long)(2 * tap + 1) * plane + hw];
  float ih = (ho * g.sh - g.ph) + i * g.dh + oh;
  float iw = (wo * g.sw - g.pw) + j * g.dw + ow;
  const int H = g.H, W = g.W;
  bool inside = !(ih <= -1 || iw <= -1 || ih >= H || iw >= W);
  int hl = (int)floorf(ih), wl = (int)floorf(iw);
  int hh = hl + 1, wh = wl + 1;
  // coefficient of each corner pixel (get_coordinate_weight, :107-127)
  float k1, k2, k3, k4;
  if (dir == 0) {
    k1 = -1 * (wl + 1 - iw); k2 = -1 * (iw - wl); k3 = (wl + 1 - iw); k4 = (iw - wl);
  } else {
    k1 = -1 * (hl + 1 - ih); k2 = (hl + 1 - ih); k3 = -1 * (ih - hl); k4 = (ih - hl);
  }
  bool p1 = inside && hl >= 0 && wl >= 0, p2 = inside && hl >= 0 && wh <= W - 1;
  bool p3 = inside && hh <= H - 1 && wl >= 0, p4 = inside && hh <= H - 1 && wh <= W - 1;
  float val = 0.f;
  for (int cc = 0; cc < cpg; ++cc) {
    int c = grp * cpg + cc;
    const float* imp = im + ((long long)b * g.C + c) * HW;
    float wt = 0.f;
    if (p1) wt += k1 * imp[hl * W + wl];
    if (p2) wt += k2 * imp[hl * W + wh];
    if (p3) wt += k3 * imp[hh * W + wl];
    if (p4) wt += k4 * imp[hh * W + wh];
    val += wt * col[((long long)(c * taps + tap)) * npos + (long long)b * plane + hw];
  }
  grad_offset[idx] = val;
}


// ---------------------------------------------------------------------------------
// Channels-last ("NHWC") forms -- the MI355X-first layout of the AlignConv hot path.
//
//   im      (B, H, W, C)            x as MIOpen's NHWC igemm kernels already hold it
//   colT    (B*Ho*Wo, kh*kw, C)     one row per output position, K index = tap*C + c
//   grad_im (B, H, W, C)
//
// One wave owns one output position: its 2*kh*kw offsets are wave-uniform, the
// bilinear footprint of a tap is derived once per wave, and the 64 lanes sweep the
// channel axis, so every global access of a wave-instruction is one contiguous
// 256-B (dword) or 1-KiB (dwordx4) segment -- gathers, column stores and, in
// col2im, the fp32 atomics (the shape MI355X_MICROARCH.md "Global float atomics"
// prices at the full chip-wide rate; the NCHW kernel's one-lane-per-row atomics are
// ~17x slower).  The GEMM consumes colT directly: out(npos, O) = colT @ W(kh*kw*C, O).
constexpr int DCN_WAVES = 4;  // waves (= positions in flight) per workgroup

struct TapFoot {
  long long o1, o2, o3, o4;  // element offsets of the corner pixels' channel vectors, -1 = outside
  float w1, w2, w3, w4;
};

__device__ __forceinline__ void nhwc_position(const Geom& g, long long pos, int& b, int& ho, int& wo,
                                              int& hw) {
  const long long plane = (long long)g.Ho * g.Wo;
  b = (int)(pos / plane);
  hw = (int)(pos - (long long)b * plane);
  ho = hw / g.Wo;
  wo = hw - ho * g.Wo;
}

template <bool COL2IM>
__device__ __forceinline__ TapFoot nhwc_tap_foot(const float* __restrict__ offset, const Geom& g,
                                                 int b, int grp, int ho, int wo, int hw, int tap) {
  const long long plane = (long long)g.Ho * g.Wo;
  const int i = tap / g.kw, j = tap - i * g.kw;
  const float* offp = offset + ((long long)b * g.dg + grp) * 2 * g.kh * g.kw * plane;
  float oh = offp[(long long)(2 * tap) * plane + hw];
  float ow = offp[(long long)(2 * tap + 1) * plane + hw];
  float h = (ho * g.sh - g.ph) + i * g.dh + oh;
  float w = (wo * g.sw - g.pw) + j * g.dw + ow;
  Foot f = COL2IM ? col2im_foot(h, w, g.H, g.W) : im2col_foot(h, w, g.H, g.W);
  const long long base = (long long)b * g.H * g.W;
  TapFoot t;
  t.o1 = f.o1 >= 0 ? (base + f.o1) * g.C : -1;
  t.o2 = f.o2 >= 0 ? (base + f.o2) * g.C : -1;
  t.o3 = f.o3 >= 0 ? (base + f.o3) * g.C : -1;
  t.o4 = f.o4 >= 0 ? (base + f.o4) * g.C : -1;
  t.w1 = f.w1; t.w2 = f.w2; t.w3 = f.w3; t.w4 = f.w4;
  return t;
}

// VEC4: C % 4 == 0 and one deformable group -> dwordx4 path (1 KiB per wave-instruction)
template <bool VEC4>
__global__ __launch_bounds__(64 * DCN_WAVES) void deform_im2col_nhwc_kernel(
    const float* __restrict__ im, const float* __restrict__ offset, Geom g, float* __restrict__ colT) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long npos = (long long)g.B * g.Ho * g.Wo;
  const int taps = g.kh * g.kw;
  const int cpg = g.C / g.dg;
  for (long long pos = (long long)blockIdx.x * DCN_WAVES + wave; pos < npos;
       pos += (long long)gridDim.x * DCN_WAVES) {
    int b, ho, wo, hw;
    nhwc_position(g, pos, b, ho, wo, hw);
    float* dst = colT + pos * taps * g.C;
    for (int tap = 0; tap < taps; ++tap) {
      if (VEC4) {
        TapFoot t = nhwc_tap_foot<false>(offset, g, b, 0, ho, wo, hw, tap);
        for (int c = lane * 4; c < g.C; c += 256) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          // same summation order as the reference: w1*v1 + w2*v2 + w3*v3 + w4*v4
          float4 a = t.o1 >= 0 ? *reinterpret_cast<const float4*>(im + t.o1 + c) : v;
          float4 bq = t.o2 >= 0 ? *reinterpret_cast<const float4*>(im + t.o2 + c) : v;
          float4 cq = t.o3 >= 0 ? *reinterpret_cast<const float4*>(im + t.o3 + c) : v;
          float4 d = t.o4 >= 0 ? *reinterpret_cast<const float4*>(im + t.o4 + c) : v;
          v.x = t.w1 * a.x + t.w2 * bq.x + t.w3 * cq.x + t.w4 * d.x;
          v.y = t.w1 * a.y + t.w2 * bq.y + t.w3 * cq.y + t.w4 * d.y;
          v.z = t.w1 * a.z + t.w2 * bq.z + t.w3 * cq.z + t.w4 * d.z;
          v.w = t.w1 * a.w + t.w2 * bq.w + t.w3 * cq.w + t.w4 * d.w;
          *reinterpret_cast<float4*>(dst + (long long)tap * g.C + c) = v;
        }
      } else {
        for (int c = lane; c < g.C; c += 64) {
          TapFoot t = nhwc_tap_foot<false>(offset, g, b, c / cpg, ho, wo, hw, tap);
          float a = t.o1 >= 0 ? im[t.o1 + c] : 0.f, bq = t.o2 >= 0 ? im[t.o2 + c] : 0.f;
          float cq = t.o3 >= 0 ? im[t.o3 + c] : 0.f, d = t.o4 >= 0 ? im[t.o4 + c] : 0.f;
          dst[(long long)tap * g.C + c] = t.w1 * a + t.w2 * bq + t.w3 * cq + t.w4 * d;
        }
      }
    }
  }
}

template <bool UNIFORM>
__global__ __launch_bounds__(64 * DCN_WAVES) void deform_col2im_nhwc_kernel(
    const float* __restrict__ colT, const float* __restrict__ offset, Geom g,
    float* __restrict__ grad_im) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long npos = (long long)g.B * g.Ho * g.Wo;
  const int taps = g.kh * g.kw;
  const int cpg = g.C / g.dg;
  for (long long pos = (long long)blockIdx.x * DCN_WAVES + wave; pos < npos;
       pos += (long long)gridDim.x * DCN_WAVES) {
    int b, ho, wo, hw;
    nhwc_position(g, pos, b, ho, wo, hw);
    const float* src = colT + pos * taps * g.C;
    for (int tap = 0; tap < taps; ++tap) {
      TapFoot t{};
      if (UNIFORM) t = nhwc_tap_foot<true>(offset, g, b, 0, ho, wo, hw, tap);
      // lanes stride the channel axis by 1: each atomic wave-instruction = 256 contiguous bytes
      for (int c = lane; c < g.C; c += 64) {
        if (!UNIFORM) t = nhwc_tap_foot<true>(offset, g, b, c / cpg, ho, wo, hw, tap);
        float top = src[(long long)tap * g.C + c];
        if (t.o1 >= 0) atomicAdd(grad_im + t.o1 + c, t.w1 * top);
        if (t.o2 >= 0) atomicAdd(grad_im + t.o2 + c, t.w2 * top);
        if (t.o3 >= 0) atomicAdd(grad_im + t.o3 + c, t.w3 * top);
        if (t.o4 >= 0) atomicAdd(grad_im + t.o4 + c, t.w4 * top);
      }
    }
  }
}

static int make_geom(const rsdet_dcn_geom* s, Geom* g) {
  if (!s) return RSDET_EINVAL;
  if (s->C < 0 || s->H < 0 || s->W < 0 || s->B < 0 || s->kh < 1 || s->kw < 1 || s->sh < 1 ||
      s->sw < 1 || s->dh < 1 || s->dw < 1 || s->ph < 0 || s->pw < 0 || s->dg < 1)
    return RSDET_EINVAL;
  if (s->C % s->dg) return RSDET_EINVAL;
  if (s->kh * s->kw > 65535) return RSDET_EINVAL;
  *g = Geom{s->C, s->H, s->W, s->kh, s->kw, s->ph, s->pw, s->sh, s->sw, s->dh, s->dw, s->B, s->dg,
            0, 0};
  g->Ho = (s->H + 2 * s->ph - (s->dh * (s->kh - 1) + 1)) / s->sh + 1;  // dcn_v1.py:328-329
  g->Wo = (s->W + 2 * s->pw - (s->dw * (s->kw - 1) + 1)) / s->sw + 1;
  if (g->Ho < 0 || g->Wo < 0) return RSDET_EINVAL;
  return RSDET_OK;
}

// enough workgroups for 256 CUs even on the 8x8 pyramid level: split channels
static int pick_c_chunk(const Geom& g, long long pos_blocks) {
  long long blocks = pos_blocks * g.kh * g.kw;
  int chunks = 1;
  while (blocks * chunks < 2048 && chunks * 16 < g.C) chunks *= 2;
  int cc = (g.C + chunks - 1) / chunks;
  // keep chunks inside one deformable group boundary pattern simple
  return cc < 1 ? 1 : cc;
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_deform_im2col_f32(const float* im, const float* offset,
                                       const rsdet_dcn_geom* geom, float* col, void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  long long npos = (long long)g.B * g.Ho * g.Wo;
  if (npos == 0 || g.C == 0) return RSDET_OK;
  if (!im || !offset || !col) return RSDET_EINVAL;
  long long pb = (npos + DCN_NT - 1) / DCN_NT;
  int cc = pick_c_chunk(g, pb);
  dim3 grid((unsigned)pb, g.kh * g.kw, (g.C + cc - 1) / cc);
  hipLaunchKernelGGL(deform_im2col_kernel, grid, dim3(DCN_NT), 0, (hipStream_t)stream, im, offset,
                     g, cc, col);
  return rsdet_launch_status();
}

extern "C" int rsdet_deform_col2im_f32(const float* col, const float* offset,
                                       const rsdet_dcn_geom* geom, float* grad_im, void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  long long npos = (long long)g.B * g.Ho * g.Wo;
  if (npos == 0 || g.C == 0) return RSDET_OK;
  if (!col || !offset || !grad_im) return RSDET_EINVAL;
  long long pb = (npos + DCN_NT - 1) / DCN_NT;
  int cc = pick_c_chunk(g, pb);
  dim3 grid((unsigned)pb, g.kh * g.kw, (g.C + cc - 1) / cc);
  hipLaunchKernelGGL(deform_col2im_kernel, grid, dim3(DCN_NT), 0, (hipStream_t)stream, col, offset,
                     g, cc, grad_im);
  return rsdet_launch_status();
}

extern "C" int rsdet_deform_col2im_coord_f32(const float* col, const float* im,
                                             const float* offset, const rsdet_dcn_geom* geom,
                                             float* grad_offset, void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  long long total = (long long)g.B * 2 * g.kh * g.kw * g.dg * g.Ho * g.Wo;
  if (total == 0) return RSDET_OK;
  if (!col || !im || !offset || !grad_offset) return RSDET_EINVAL;
  hipLaunchKernelGGL(deform_col2im_coord_kernel, dim3((unsigned)((total + DCN_NT - 1) / DCN_NT)),
                     dim3(DCN_NT), 0, (hipStream_t)stream, col, im, offset, g, grad_offset);
  return rsdet_launch_status();
}

static int nhwc_grid(long long npos) {
  long long blocks = (npos + DCN_WAVES - 1) / DCN_WAVES;
  const long long cap = 256LL * 8 * 4;  // 256 CUs x 8 workgroups, then grid-stride
  return (int)(blocks < cap ? blocks : cap);
}

extern "C" int rsdet_deform_im2col_nhwc_f32(const float* im, const float* offset,
                                            const rsdet_dcn_geom* geom, float* colT, void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  long long npos = (long long)g.B * g.Ho * g.Wo;
  if (npos == 0 || g.C == 0) return RSDET_OK;
  if (!im || !offset || !colT) return RSDET_EINVAL;
  bool vec4 = (g.C % 4 == 0) && g.dg == 1 && (((uintptr_t)im | (uintptr_t)colT) % 16 == 0);
  if (vec4)
    hipLaunchKernelGGL(deform_im2col_nhwc_kernel<true>, dim3(nhwc_grid(npos)), dim3(64 * DCN_WAVES), 0,
                       (hipStream_t)stream, im, offset, g, colT);
  else
    hipLaunchKernelGGL(deform_im2col_nhwc_kernel<false>, dim3(nhwc_grid(npos)), dim3(64 * DCN_WAVES),
                       0, (hipStream_t)stream, im, offset, g, colT);
  return rsdet_launch_status();
}

extern "C" int rsdet_deform_col2im_nhwc_f32(const float* colT, const float* offset,
                                            const rsdet_dcn_geom* geom, float* grad_im,
                                            void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  long long npos = (long long)g.B * g.Ho * g.Wo;
  if (npos == 0 || g.C == 0) return RSDET_OK;
  if (!colT || !offset || !grad_im) return RSDET_EINVAL;
  if (g.dg == 1)
    hipLaunchKernelGGL(deform_col2im_nhwc_kernel<true>, dim3(nhwc_grid(npos)), dim3(64 * DCN_WAVES), 0,
                       (hipStream_t)stream, colT, offset, g, grad_im);
  else
    hipLaunchKernelGGL(deform_col2im_nhwc_kernel<false>, dim3(nhwc_grid(npos)), dim3(64 * DCN_WAVES),
                       0, (hipStream_t)stream, colT, offset, g, grad_im);
  return rsdet_launch_status();
}
