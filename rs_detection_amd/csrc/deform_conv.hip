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
#include "rsdet_bf16.h"

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

// 3x3 form of the kernel above: one thread per POSITION keeps the nine bilinear footprints in registers and walks
// a chunk of channels with the nine taps innermost.  The kernel above gives every tap its own wave, so the nine
// taps of a position re-read the same neighbourhood of a channel plane from beyond L2 (PMC: >= 1 GB fetched for
// a 67 MB input); here a fetched line serves all the taps (and corners) that touch it while it is still in L1.
// Same arithmetic, same output.  grid: (ceil(B*Ho*Wo / NT), C / c_chunk), a chunk never straddles a deformable group.
// The two corners of a footprint row are neighbours in memory, so each row is ONE 8-byte load of the pixel pair
// (xs, xs + 1), xs = clamp(floor(w), 0, W - 2): half the vector-memory instructions of four scalar gathers (the
// kernel is bound by the texture-addresser rate: 16 clocks per 64-lane gather).  sel says which half of the pair
// is the left / right corner (or none: 0.f), so the sum is the reference's w1*v1 + w2*v2 + w3*v3 + w4*v4 exactly.
struct PairFoot {
  int lo, hi;        // element offset of the pixel pair in the upper / lower footprint row, -1 = row outside
  float w1, w2, w3, w4;
  int sel;           // bits 0-1: left corner 0 = none, 1 = pair.x, 2 = pair.y; bits 2-3: right corner
};

__device__ __forceinline__ PairFoot pair_foot(float h, float w, int H, int W) {
  PairFoot f{-1, -1, 0.f, 0.f, 0.f, 0.f, 0};
  if (!(h > -1 && w > -1 && h < H && w < W)) return f;
  const int hl = (int)floorf(h), wl = (int)floorf(w);
  const int hh = hl + 1, wh = wl + 1;
  const float lh = h - hl, lw = w - wl;
  const float uh = 1 - lh, uw = 1 - lw;
  const int xs = min(max(wl, 0), W - 2);
  if (hl >= 0) f.lo = hl * W + xs;
  if (hh <= H - 1) f.hi = hh * W + xs;
  const int left = wl >= 0 ? (wl == xs ? 1 : 2) : 0;        // wl in [-1, W-1]
  const int right = wh <= W - 1 ? (wh == xs + 1 ? 2 : 1) : 0;
  f.sel = left | (right << 2);
  f.w1 = uh * uw;
  f.w2 = uh * lw;
  f.w3 = lh * uw;
  f.w4 = lh * lw;
  return f;
}

template <int TAPS, typename TCOL>
__global__ __launch_bounds__(DCN_NT) void deform_im2col_taps_kernel(const float* __restrict__ im,
                                                                    const float* __restrict__ offset, Geom g,
                                                                    int c_chunk, TCOL* __restrict__ col) {
  const long long plane = (long long)g.Ho * g.Wo;
  const long long npos = (long long)g.B * plane;
  const long long pos = (long long)blockIdx.x * DCN_NT + threadIdx.x;
  if (pos >= npos) return;
  const int b = (int)(pos / plane);
  const int hw = (int)(pos - (long long)b * plane);
  const int ho = hw / g.Wo, wo = hw - ho * g.Wo;
  const int cpg = g.C / g.dg;
  const int c0 = blockIdx.y * c_chunk, c1 = min(g.C, c0 + c_chunk);
  const int grp = c0 / cpg;
  const float* offp = offset + ((long long)b * g.dg + grp) * 2 * TAPS * plane;
  PairFoot f[TAPS];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int i = tap / g.kw, j = tap - i * g.kw;
    const float oh = offp[(long long)(2 * tap) * plane + hw];
    const float ow = offp[(long long)(2 * tap + 1) * plane + hw];
    f[tap] = pair_foot((ho * g.sh - g.ph) + i * g.dh + oh, (wo * g.sw - g.pw) + j * g.dw + ow, g.H, g.W);
  }
  const long long HW = (long long)g.H * g.W;
  for (int c = c0; c < c1; ++c) {
    const float* imp = im + ((long long)b * g.C + c) * HW;
    TCOL* cp = col + (long long)c * TAPS * npos + pos;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      typedef float pair_t __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load at dword alignment
      // (unconditional loads from clamped offsets + selects: a load under a branch is followed by a wait, which
      //  serialised the 18 pair loads of a channel)
      const pair_t zero2 = {0.f, 0.f};
      pair_t a = *reinterpret_cast<const pair_t*>(imp + max(f[tap].lo, 0));
      pair_t d = *reinterpret_cast<const pair_t*>(imp + max(f[tap].hi, 0));
      a = f[tap].lo >= 0 ? a : zero2;
      d = f[tap].hi >= 0 ? d : zero2;
      const int l = f[tap].sel & 3, r = f[tap].sel >> 2;
      const float v1 = l == 1 ? a.x : (l == 2 ? a.y : 0.f), v2 = r == 2 ? a.y : (r == 1 ? a.x : 0.f);
      const float v3 = l == 1 ? d.x : (l == 2 ? d.y : 0.f), v4 = r == 2 ? d.y : (r == 1 ? d.x : 0.f);
      st1(cp + (long long)tap * npos, f[tap].w1 * v1 + f[tap].w2 * v2 + f[tap].w3 * v3 + f[tap].w4 * v4);
    }
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

// ---------------------------------------------------------------------------------
// Gather form of the channels-last col2im (one deformable group): no floating-point atomics.
//
// The scatter kernel above issues 4 corners x kh*kw taps x C channels atomic adds per position --
// 2.4 GB of atomic traffic at pyramid level 0, the chip-wide fp32 atomic floor (1.84 ms).  The
// footprints do not depend on the channel, so the scatter can be inverted once per call on
// (position, tap, corner) triples -- 2.4 M integers instead of 2.4 GB of floats:
//   dcn_idx_count   histogram of contributions per input pixel          (int atomics)
//   dcn_idx_scan    exclusive prefix sum over the pixels                (chunk sums + per-chunk scan)
//   dcn_idx_fill    each contribution takes a slot of its pixel: {colT row, bilinear weight}
//   dcn_gather      one wave per input pixel, lanes over channels: grad_im[pixel] = sum_w * colT[row]
//                   (1-KiB coalesced row reads, ONE plain store per element, no zero fill)
// The order of the terms of a pixel's sum is the arrival order of dcn_idx_fill -- as free as the
// order of the atomics it replaces (and of the reference's atomicAdd, dcn_v1.py:110-113).
struct PixFoot {
  long long p1, p2, p3, p4;  // linear input-pixel indices (b*H*W + y*W + x), -1 = outside
  float w1, w2, w3, w4;
};

__device__ __forceinline__ PixFoot pix_foot(const float* __restrict__ offset, const Geom& g, long long item) {
  const int taps = g.kh * g.kw;
  const long long pos = item / taps;
  const int tap = (int)(item - pos * taps);
  int b, ho, wo, hw;
  nhwc_position(g, pos, b, ho, wo, hw);
  const long long plane = (long long)g.Ho * g.Wo;
  const int i = tap / g.kw, j = tap - i * g.kw;
  const float* offp = offset + (long long)b * g.dg * 2 * taps * plane;
  const float h = (ho * g.sh - g.ph) + i * g.dh + offp[(long long)(2 * tap) * plane + hw];
  const float w = (wo * g.sw - g.pw) + j * g.dw + offp[(long long)(2 * tap + 1) * plane + hw];
  const Foot f = col2im_foot(h, w, g.H, g.W);
  const long long base = (long long)b * g.H * g.W;
  PixFoot t;
  t.p1 = f.o1 >= 0 ? base + f.o1 : -1;
  t.p2 = f.o2 >= 0 ? base + f.o2 : -1;
  t.p3 = f.o3 >= 0 ? base + f.o3 : -1;
  t.p4 = f.o4 >= 0 ? base + f.o4 : -1;
  t.w1 = f.w1; t.w2 = f.w2; t.w3 = f.w3; t.w4 = f.w4;
  return t;
}

__global__ __launch_bounds__(256) void dcn_idx_count_kernel(const float* __restrict__ offset, Geom g,
                                                            long long items, int* __restrict__ cnt) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  const PixFoot t = pix_foot(offset, g, item);
  if (t.p1 >= 0) atomicAdd(cnt + t.p1, 1);
  if (t.p2 >= 0) atomicAdd(cnt + t.p2, 1);
  if (t.p3 >= 0) atomicAdd(cnt + t.p3, 1);
  if (t.p4 >= 0) atomicAdd(cnt + t.p4, 1);
}

// exclusive scan of cnt[0..n) into start[0..n], start[n] = total, in two launches: per-chunk sums, then every
// workgroup adds up the sums before its chunk and scans the chunk (also clears cnt for dcn_idx_fill).  A single
// 1024-thread workgroup walking all pixels took 36 us per call.
constexpr int SCAN_NT = 256, SCAN_PER = 16, SCAN_CHUNK = SCAN_NT * SCAN_PER;  // 4096 pixels per workgroup

__device__ __forceinline__ int block_sum_256(int v, int* s_tmp) {  // all threads get the total
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  const int t = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(SCAN_NT) void dcn_idx_chunk_sum_kernel(const int* __restrict__ cnt, long long n,
                                                                    int* __restrict__ chunk_sum) {
  __shared__ int s_tmp[4];
  const long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_PER;
  int sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER; ++k)
    if (base + k < n) sum += cnt[base + k];
  const int t = block_sum_256(sum, s_tmp);
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = t;
}

__global__ __launch_bounds__(SCAN_NT) void dcn_idx_scan_kernel(int* __restrict__ cnt, long long n,
                                                               const int* __restrict__ chunk_sum, int nchunks,
                                                               int* __restrict__ start) {
  __shared__ int s_tmp[4];
  __shared__ int s_scan[SCAN_NT];
  const int tid = threadIdx.x;
  int before = 0;  // sum of the chunks in front of this one
  for (int c = tid; c < (int)blockIdx.x; c += SCAN_NT) before += chunk_sum[c];
  before = block_sum_256(before, s_tmp);
  const long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)tid * SCAN_PER;
  int v[SCAN_PER], sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER; ++k) {
    v[k] = base + k < n ? cnt[base + k] : 0;
    sum += v[k];
  }
  s_scan[tid] = sum;
  __syncthreads();
  for (int off = 1; off < SCAN_NT; off <<= 1) {  // Hillis-Steele over the 256 thread sums
    const int o = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += o;
    __syncthreads();
  }
  int run = before + s_scan[tid] - sum;
#pragma unroll
  for (int k = 0; k < SCAN_PER; ++k)
    if (base + k < n) {
      start[base + k] = run;
      cnt[base + k] = 0;
      run += v[k];
    }
  if ((int)blockIdx.x == nchunks - 1 && tid == SCAN_NT - 1) start[n] = before + s_scan[SCAN_NT - 1];
}

__global__ __launch_bounds__(256) void dcn_idx_fill_kernel(const float* __restrict__ offset, Geom g, long long items,
                                                           const int* __restrict__ start, int* __restrict__ fill,
                                                           int* __restrict__ ent_row, float* __restrict__ ent_w) {
  const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
  if (item >= items) return;
  const PixFoot t = pix_foot(offset, g, item);
#define RSDET_PUT(P, Wt)                                    \
  if (P >= 0) {                                             \
    const int slot = start[P] + atomicAdd(fill + P, 1);     \
    ent_row[slot] = (int)item;                              \
    ent_w[slot] = Wt;                                       \
  }
  RSDET_PUT(t.p1, t.w1)
  RSDET_PUT(t.p2, t.w2)
  RSDET_PUT(t.p3, t.w3)
  RSDET_PUT(t.p4, t.w4)
#undef RSDET_PUT
}

// ---- the same index for SEVERAL calls at once (the five AlignConv levels of one step) ----
// Pixels and items of the levels are laid end to end: one histogram, one scan and one fill serve all of them
// (5 launches instead of 5 per level; the small levels cost ~5 us of launch latency each on their own).  Workgroups
// map to levels through blk_base so that a workgroup never straddles two geometries; entries hold the item id INSIDE
// their level, start[] holds slots of the shared entry arrays, so a level's gather reads start + pix_base[l] with the
// shared ent_row / ent_w.
constexpr int DCN_IDX_LEVELS = RSDET_DCN_INDEX_MAX_LEVELS;

struct IdxLevels {
  int n;
  unsigned blk_base[DCN_IDX_LEVELS + 1];
  long long pix_base[DCN_IDX_LEVELS + 1];
  long long items[DCN_IDX_LEVELS];
  const float* offset[DCN_IDX_LEVELS];
  Geom g[DCN_IDX_LEVELS];
};

__device__ __forceinline__ int idx_level_of(const IdxLevels& lv, unsigned blk) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < DCN_IDX_LEVELS; ++i)
    if (i < lv.n && blk >= lv.blk_base[i]) l = i;
  return l;
}

__global__ __launch_bounds__(256) void dcn_idx_count_multi_kernel(const IdxLevels lv, int* __restrict__ cnt) {
  const int l = idx_level_of(lv, blockIdx.x);
  const long long item = (long long)(blockIdx.x - lv.blk_base[l]) * 256 + threadIdx.x;
  if (item >= lv.items[l]) return;
  const PixFoot t = pix_foot(lv.offset[l], lv.g[l], item);
  int* c = cnt + lv.pix_base[l];
  if (t.p1 >= 0) atomicAdd(c + t.p1, 1);
  if (t.p2 >= 0) atomicAdd(c + t.p2, 1);
  if (t.p3 >= 0) atomicAdd(c + t.p3, 1);
  if (t.p4 >= 0) atomicAdd(c + t.p4, 1);
}

__global__ __launch_bounds__(256) void dcn_idx_fill_multi_kernel(const IdxLevels lv, const int* __restrict__ start,
                                                                 int* __restrict__ fill, int* __restrict__ ent_row,
                                                                 float* __restrict__ ent_w) {
  const int l = idx_level_of(lv, blockIdx.x);
  const long long item = (long long)(blockIdx.x - lv.blk_base[l]) * 256 + threadIdx.x;
  if (item >= lv.items[l]) return;
  const PixFoot t = pix_foot(lv.offset[l], lv.g[l], item);
  const int* st = start + lv.pix_base[l];
  int* fl = fill + lv.pix_base[l];
#define RSDET_PUT(P, Wt)                                 \
  if (P >= 0) {                                          \
    const int slot = st[P] + atomicAdd(fl + P, 1);       \
    ent_row[slot] = (int)item;                           \
    ent_w[slot] = Wt;                                    \
  }
  RSDET_PUT(t.p1, t.w1)
  RSDET_PUT(t.p2, t.w2)
  RSDET_PUT(t.p3, t.w3)
  RSDET_PUT(t.p4, t.w4)
#undef RSDET_PUT
}

// one wave per input pixel; pixels are walked in 8x8 tiles so that the four pixels sharing a colT row
// (the corners of one sampling point) are processed close together, and the workgroup ids are renumbered so that an
// XCD owns a contiguous run of tiles (rsdet_xcd_contiguous): round-robin placement would put the four on four
// different L2s and the row would be fetched again by each
template <bool VEC4, typename TROW, typename TOUT = float>
__global__ __launch_bounds__(64 * DCN_WAVES) void dcn_gather_kernel(const TROW* __restrict__ colT,
                                                                    const int* __restrict__ start,
                                                                    const int* __restrict__ ent_row,
                                                                    const float* __restrict__ ent_w, Geom g,
                                                                    TOUT* __restrict__ grad_im) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long npix = (long long)g.B * g.H * g.W;
  long long wid = (long long)rsdet_xcd_contiguous(blockIdx.x, gridDim.x) * DCN_WAVES + wave;
  if (wid >= npix) return;
  long long pix = wid;
  if ((g.H & 7) == 0 && (g.W & 7) == 0) {  // tile swizzle (16x16 tiles measure the same)
    const long long plane = (long long)g.H * g.W;
    const int b = (int)(wid / plane);
    const int r = (int)(wid - (long long)b * plane);
    const int tile = r >> 6, in = r & 63, tpr = g.W >> 3;
    const int y = (tile / tpr) * 8 + (in >> 3), x = (tile % tpr) * 8 + (in & 7);
    pix = (long long)b * plane + (long long)y * g.W + x;
  }
  const int e0 = start[pix], e1 = start[pix + 1];
  const int rowlen = g.C;  // colT row of (position, tap) = item * C
  if (VEC4) {
    for (int c = lane * 4; c < g.C; c += 256) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int e = e0;
      for (; e + 4 <= e1; e += 4) {  // four independent row reads in flight
        const int r0 = ent_row[e], r1 = ent_row[e + 1], r2 = ent_row[e + 2], r3 = ent_row[e + 3];
        const float w0 = ent_w[e], w1 = ent_w[e + 1], w2 = ent_w[e + 2], w3 = ent_w[e + 3];
        const float4 v0 = ld4(colT + (long long)r0 * rowlen + c);
        const float4 v1 = ld4(colT + (long long)r1 * rowlen + c);
        const float4 v2 = ld4(colT + (long long)r2 * rowlen + c);
        const float4 v3 = ld4(colT + (long long)r3 * rowlen + c);
        acc.x += w0 * v0.x; acc.y += w0 * v0.y; acc.z += w0 * v0.z; acc.w += w0 * v0.w;
        acc.x += w1 * v1.x; acc.y += w1 * v1.y; acc.z += w1 * v1.z; acc.w += w1 * v1.w;
        acc.x += w2 * v2.x; acc.y += w2 * v2.y; acc.z += w2 * v2.z; acc.w += w2 * v2.w;
        acc.x += w3 * v3.x; acc.y += w3 * v3.y; acc.z += w3 * v3.z; acc.w += w3 * v3.w;
      }
      for (; e < e1; ++e) {
        const int r0 = ent_row[e];
        const float w0 = ent_w[e];
        const float4 v0 = ld4(colT + (long long)r0 * rowlen + c);
        acc.x += w0 * v0.x; acc.y += w0 * v0.y; acc.z += w0 * v0.z; acc.w += w0 * v0.w;
      }
      st4(grad_im + pix * g.C + c, acc);      // (bf16 output: rounded to nearest even, what `.to(bfloat16)` did in a pass of its own)
    }
  } else {
    for (int c = lane; c < g.C; c += 64) {
      float acc = 0.f;
      for (int e = e0; e < e1; ++e) acc += ent_w[e] * ld1(colT + (long long)ent_row[e] * rowlen + c);
      st1(grad_im + pix * g.C + c, acc);
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
  const int cpg = g.C / g.dg;
#ifndef RSDET_IM2COL_CHUNK
#define RSDET_IM2COL_CHUNK 16
#endif
  constexpr int kChunk = RSDET_IM2COL_CHUNK;  // channels per workgroup row
  if (g.kh * g.kw == 9 && cpg % kChunk == 0 && g.W >= 2) {  // the AlignConv shape: nine taps per thread
    hipLaunchKernelGGL((deform_im2col_taps_kernel<9, float>), dim3((unsigned)pb, g.C / kChunk), dim3(DCN_NT), 0,
                       (hipStream_t)stream, im, offset, g, kChunk, col);
    return rsdet_launch_status();
  }
  int cc = pick_c_chunk(g, pb);
  dim3 grid((unsigned)pb, g.kh * g.kw, (g.C + cc - 1) / cc);
  hipLaunchKernelGGL(deform_im2col_kernel, grid, dim3(DCN_NT), 0, (hipStream_t)stream, im, offset,
                     g, cc, col);
  return rsdet_launch_status();
}

// bf16 column matrix for the autocast step (the GEMMs that consume it run on bf16 MFMA): same kernel, the store rounds
// to nearest even.  Only the AlignConv geometry (3x3 taps, channels per deformable group a multiple of 16).
extern "C" int rsdet_deform_im2col_bf16col_f32(const float* im, const float* offset, const rsdet_dcn_geom* geom,
                                               uint16_t* col, void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  long long npos = (long long)g.B * g.Ho * g.Wo;
  if (npos == 0 || g.C == 0) return RSDET_OK;
  if (!im || !offset || !col) return RSDET_EINVAL;
  const long long pb = (npos + DCN_NT - 1) / DCN_NT;
  const int cpg = g.C / g.dg;
  if (!(g.kh * g.kw == 9 && cpg % RSDET_IM2COL_CHUNK == 0 && g.W >= 2)) return RSDET_EINVAL;
  hipLaunchKernelGGL((deform_im2col_taps_kernel<9, bf16_t>), dim3((unsigned)pb, g.C / RSDET_IM2COL_CHUNK), dim3(DCN_NT),
                     0, (hipStream_t)stream, im, offset, g, RSDET_IM2COL_CHUNK, col);
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

void rsdet_launch_index_scan(int* cnt, long long n, int* chunk_sum, int* start, hipStream_t stream) {
  const int nchunks = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  hipLaunchKernelGGL(dcn_idx_chunk_sum_kernel, dim3(nchunks), dim3(SCAN_NT), 0, stream, cnt, n, chunk_sum);
  hipLaunchKernelGGL(dcn_idx_scan_kernel, dim3(nchunks), dim3(SCAN_NT), 0, stream, cnt, n, chunk_sum, nchunks, start);
}

static inline size_t dcn_align256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t rsdet_deform_col2im_gather_ws_size(const rsdet_dcn_geom* geom) {
  Geom g;
  if (make_geom(geom, &g)) return 0;
  const size_t npix = (size_t)g.B * g.H * g.W, ent = (size_t)g.B * g.Ho * g.Wo * g.kh * g.kw * 4;
  return dcn_align256((npix + 1) * 4) * 2 + dcn_align256(ent * 4) * 2 +
         dcn_align256((npix / 4096 + 1) * 4);  // cnt | start | ent_row | ent_w | chunk sums
}

template <typename TROW>
static int dcn_col2im_gather(const TROW* colT, const float* offset, const rsdet_dcn_geom* geom, float* grad_im,
                             void* ws, size_t ws_bytes, void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  if (g.dg != 1) return RSDET_EINVAL;  // footprints must not depend on the channel
  const long long npos = (long long)g.B * g.Ho * g.Wo, npix = (long long)g.B * g.H * g.W;
  if (npix == 0 || g.C == 0) return RSDET_OK;
  if (!offset || !grad_im || (npos > 0 && !colT)) return RSDET_EINVAL;
  const long long items = npos * g.kh * g.kw;
  if (items * 4 > 0x7fffffffLL) return RSDET_EINVAL;  // int slots
  if (!ws || ((uintptr_t)ws & 15) || ws_bytes < rsdet_deform_col2im_gather_ws_size(geom)) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  int* cnt = (int*)w;
  int* start = (int*)(w + dcn_align256((npix + 1) * 4));
  int* ent_row = (int*)(w + dcn_align256((npix + 1) * 4) * 2);
  float* ent_w = (float*)(w + dcn_align256((npix + 1) * 4) * 2 + dcn_align256((size_t)items * 4 * 4));
  if (hipMemsetAsync(cnt, 0, (size_t)(npix + 1) * 4, s) != hipSuccess) return RSDET_ELAUNCH;
  if (items > 0)
    hipLaunchKernelGGL(dcn_idx_count_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, offset, g, items,
                       cnt);
  int* chunk_sum = (int*)(w + dcn_align256((npix + 1) * 4) * 2 + dcn_align256((size_t)items * 4 * 4) * 2);
  rsdet_launch_index_scan(cnt, npix, chunk_sum, start, s);
  if (items > 0)
    hipLaunchKernelGGL(dcn_idx_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, offset, g, items,
                       start, cnt, ent_row, ent_w);
  // four consecutive channels per lane: 16-byte (fp32) / 8-byte (bf16) row loads, 16-byte stores
  const bool vec4 = (g.C % 4 == 0) && ((uintptr_t)colT % (4 * sizeof(TROW)) == 0) && ((uintptr_t)grad_im % 16 == 0);
  const unsigned blocks = (unsigned)((npix + DCN_WAVES - 1) / DCN_WAVES);
  if (vec4)
    hipLaunchKernelGGL((dcn_gather_kernel<true, TROW>), dim3(blocks), dim3(64 * DCN_WAVES), 0, s, colT, start, ent_row,
                       ent_w, g, grad_im);
  else
    hipLaunchKernelGGL((dcn_gather_kernel<false, TROW>), dim3(blocks), dim3(64 * DCN_WAVES), 0, s, colT, start,
                       ent_row, ent_w, g, grad_im);
  return rsdet_launch_status();
}

// ---- shared index of several levels: build once, gather per level ----
struct IdxPlan {
  IdxLevels lv;
  long long npix, nitems;
  size_t off_start, off_row, off_w, off_chunk, bytes;
};

static int idx_plan(const rsdet_dcn_index_levels* s, IdxPlan* p) {
  if (!s || s->n_levels < 1 || s->n_levels > DCN_IDX_LEVELS) return RSDET_EINVAL;
  IdxLevels& lv = p->lv;
  lv.n = s->n_levels;
  long long pix = 0, items = 0, blk = 0;
  for (int l = 0; l < lv.n; ++l) {
    int rc = make_geom(&s->geom[l], &lv.g[l]);
    if (rc) return rc;
    const Geom& g = lv.g[l];
    if (g.dg != 1) return RSDET_EINVAL;
    lv.items[l] = (long long)g.B * g.Ho * g.Wo * g.kh * g.kw;
    lv.offset[l] = s->offset[l];
    if (lv.items[l] > 0 && !s->offset[l]) return RSDET_EINVAL;
    lv.pix_base[l] = pix;
    lv.blk_base[l] = (unsigned)blk;
    pix += (long long)g.B * g.H * g.W;
    items += lv.items[l];
    blk += (lv.items[l] + 255) / 256;
  }
  lv.pix_base[lv.n] = pix;
  lv.blk_base[lv.n] = (unsigned)blk;
  if (items * 4 > 0x7fffffffLL || blk > 0x7fffffffLL) return RSDET_EINVAL;  // int slots
  p->npix = pix;
  p->nitems = items;
  p->off_start = dcn_align256((size_t)(pix + 1) * 4);
  p->off_row = p->off_start * 2;
  p->off_w = p->off_row + dcn_align256((size_t)items * 4 * 4);
  p->off_chunk = p->off_w + dcn_align256((size_t)items * 4 * 4);
  p->bytes = p->off_chunk + dcn_align256(((size_t)pix / 4096 + 1) * 4);
  return RSDET_OK;
}

extern "C" size_t rsdet_deform_col2im_index_multi_ws_size(const rsdet_dcn_index_levels* levels) {
  IdxPlan p;
  return idx_plan(levels, &p) ? 0 : p.bytes;
}

extern "C" int rsdet_deform_col2im_index_multi_f32(const rsdet_dcn_index_levels* levels, void* ws, size_t ws_bytes,
                                                   long long* pix_base, size_t* ent_row_offset,
                                                   size_t* ent_w_offset, void* stream) {
  IdxPlan p;
  int rc = idx_plan(levels, &p);
  if (rc) return rc;
  if (!pix_base || !ent_row_offset || !ent_w_offset) return RSDET_EINVAL;
  if (!ws || ((uintptr_t)ws & 15) || ws_bytes < p.bytes) return RSDET_EINVAL;
  for (int l = 0; l <= p.lv.n; ++l) pix_base[l] = p.lv.pix_base[l];
  *ent_row_offset = p.off_row;
  *ent_w_offset = p.off_w;
  if (p.npix == 0) return RSDET_OK;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  int* cnt = (int*)w;
  int* start = (int*)(w + p.off_start);
  const unsigned blocks = p.lv.blk_base[p.lv.n];
  if (hipMemsetAsync(cnt, 0, (size_t)(p.npix + 1) * 4, s) != hipSuccess) return RSDET_ELAUNCH;
  if (blocks > 0) hipLaunchKernelGGL(dcn_idx_count_multi_kernel, dim3(blocks), dim3(256), 0, s, p.lv, cnt);
  rsdet_launch_index_scan(cnt, p.npix, (int*)(w + p.off_chunk), start, s);
  if (blocks > 0)
    hipLaunchKernelGGL(dcn_idx_fill_multi_kernel, dim3(blocks), dim3(256), 0, s, p.lv, start, cnt,
                       (int*)(w + p.off_row), (float*)(w + p.off_w));
  return rsdet_launch_status();
}

template <typename TROW, typename TOUT>
static int dcn_col2im_gather_indexed(const TROW* colT, const rsdet_dcn_geom* geom, const int* start,
                                     const int* ent_row, const float* ent_w, TOUT* grad_im, void* stream) {
  Geom g;
  int rc = make_geom(geom, &g);
  if (rc) return rc;
  if (g.dg != 1) return RSDET_EINVAL;
  const long long npos = (long long)g.B * g.Ho * g.Wo, npix = (long long)g.B * g.H * g.W;
  if (npix == 0 || g.C == 0) return RSDET_OK;
  if (!start || !ent_row || !ent_w || !grad_im || (npos > 0 && !colT)) return RSDET_EINVAL;
  const bool vec4 = (g.C % 4 == 0) && ((uintptr_t)colT % (4 * sizeof(TROW)) == 0) &&
                    ((uintptr_t)grad_im % (4 * sizeof(TOUT)) == 0);
  const unsigned blocks = (unsigned)((npix + DCN_WAVES - 1) / DCN_WAVES);
  hipStream_t s = (hipStream_t)stream;
  if (vec4)
    hipLaunchKernelGGL((dcn_gather_kernel<true, TROW, TOUT>), dim3(blocks), dim3(64 * DCN_WAVES), 0, s, colT, start,
                       ent_row, ent_w, g, grad_im);
  else
    hipLaunchKernelGGL((dcn_gather_kernel<false, TROW, TOUT>), dim3(blocks), dim3(64 * DCN_WAVES), 0, s, colT, start,
                       ent_row, ent_w, g, grad_im);
  return rsdet_launch_status();
}

extern "C" int rsdet_deform_col2im_gather_indexed_nhwc_f32(const float* colT, const rsdet_dcn_geom* geom,
                                                           const int* start, const int* ent_row, const float* ent_w,
                                                           float* grad_im, void* stream) {
  return dcn_col2im_gather_indexed<float, float>(colT, geom, start, ent_row, ent_w, grad_im, stream);
}

extern "C" int rsdet_deform_col2im_gather_indexed_nhwc_bf16col_f32(const uint16_t* colT, const rsdet_dcn_geom* geom,
                                                                   const int* start, const int* ent_row,
                                                                   const float* ent_w, float* grad_im, void* stream) {
  return dcn_col2im_gather_indexed<bf16_t, float>((const bf16_t*)colT, geom, start, ent_row, ent_w, grad_im, stream);
}

// ... and grad_im written as bf16 (the input of the autocast step is bf16: its gradient was cast in a pass of its own)
extern "C" int rsdet_deform_col2im_gather_indexed_nhwc_bf16col_bf16(const uint16_t* colT, const rsdet_dcn_geom* geom,
                                                                    const int* start, const int* ent_row,
                                                                    const float* ent_w, uint16_t* grad_im, void* stream) {
  return dcn_col2im_gather_indexed<bf16_t, bf16_t>((const bf16_t*)colT, geom, start, ent_row, ent_w, (bf16_t*)grad_im,
                                                   stream);
}

extern "C" int rsdet_deform_col2im_gather_nhwc_f32(const float* colT, const float* offset,
                                                   const rsdet_dcn_geom* geom, float* grad_im, void* ws,
                                                   size_t ws_bytes, void* stream) {
  return dcn_col2im_gather<float>(colT, offset, geom, grad_im, ws, ws_bytes, stream);
}

// the column gradient in bf16 (output of a bf16 GEMM in the autocast step); sums and grad_im stay fp32
extern "C" int rsdet_deform_col2im_gather_nhwc_bf16col_f32(const uint16_t* colT, const float* offset,
                                                           const rsdet_dcn_geom* geom, float* grad_im, void* ws,
                                                           size_t ws_bytes, void* stream) {
  return dcn_col2im_gather<bf16_t>(colT, offset, geom, grad_im, ws, ws_bytes, stream);
}
