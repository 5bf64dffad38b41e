// van_ops.hip -- the elementwise passes of a VAN block, fused (NCHW fp32, gfx950).
//
// The block of /root/reference/python/jdet/models/backbones/van.py:46-122 is, between its convolutions,
//   attention:  u = GELU(proj_1(x) )            a = conv1(dw7(dw5(u)))        gate = u * a
//               x <- x + ls1 * (proj_2(gate) + xn)                  (xn = norm1(x), the attention's own shortcut)
//   mlp:        x <- x + ls2 * fc2(GELU(dw3(fc1(xn2))))
// and every 1x1 convolution carries a bias.  Run as library calls that is, per block and step, 5 broadcast bias adds,
// 2 GELUs, a product, a sum and two scale-and-add passes forward, and their backward passes plus 5 strided bias
// reductions (torch: ~16 us each) backward -- 21 ms of elementwise kernels in the 83 ms Oriented R-CNN VAN-B3 step,
// most of them on maps so small (stage 3: 27 blocks of 2 x 320 x 64 x 64) that the HOST cannot issue them as fast as
// the GPU retires them.  Here the convolutions run WITHOUT their bias and each tail is one pass:
//   bias_gelu   y = GELU(x + b[c])                    backward: gx = gy * GELU'(x + b),  gb = sum gx
//   gate        y = u * (a + b[c])                    backward: gu = g * (a + b),  ga = g * u,  gb = sum ga
//   residual    y = x + ls[c] * (p + b[c] + sc)       backward: gp = ls * g (also sc's gradient), gb = ls * sum g,
//                                                               gls = sum g * (p + b + sc); x's gradient is g itself
// One workgroup = one slice of one (n, c) plane (the per-channel parameters are scalars), float4 accesses; the
// per-channel sums are two-stage and deterministic: slice partials, then one wave per channel folds them (a second,
// ~4 us launch; folding in the last workgroup to arrive through device-scope atomics was measured: it doubles the
// 15 us backward passes -- every workgroup's arrival serialises on one counter).
// GELU is the erf form of torch.nn.GELU() (what jittor.nn.GELU computes), in fp32 like torch's kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "rsdet_api_internal.h"

namespace rsdet {

constexpr int VE_NT = 256;

__device__ __forceinline__ float ve_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float ve_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

struct VeSlice {
  long long base;   // element offset of the plane
  int i0, i1, c;    // slice [i0, i1) of the plane, its channel
};

__device__ __forceinline__ VeSlice ve_slice(int C, int HW) {
  const int pc = blockIdx.y, S = gridDim.x, s = blockIdx.x;
  const int chunk = (((HW + S - 1) / S) + 3) & ~3;
  VeSlice v;
  v.base = (long long)pc * HW;
  v.c = pc % C;
  v.i0 = min(HW, s * chunk);
  v.i1 = min(HW, v.i0 + chunk);
  return v;
}

// (a, b) summed over the workgroup -> partial[(c * (N * S) + n * S + s) * 2 + {0, 1}]
__device__ __forceinline__ void ve_reduce_store(float a, float b, int C, float* __restrict__ partial) {
  __shared__ float s_red[VE_NT / 64][2];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
  }
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][0] = a, s_red[threadIdx.x >> 6][1] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int w = 0; w < VE_NT / 64; ++w) ta += s_red[w][0], tb += s_red[w][1];
    const int pc = blockIdx.y, S = gridDim.x, n = pc / C, c = pc - n * C, N = gridDim.y / C;
    float* dst = partial + ((long long)c * (N * S) + n * S + blockIdx.x) * 2;
    dst[0] = ta, dst[1] = tb;
  }
}

// body(i, vec): vec = true handles elements i .. i + 3, false the single element i
template <typename F>
__device__ __forceinline__ void ve_loop(const VeSlice& v, bool aligned, F body) {
  if (aligned) {
    for (int i = v.i0 + 4 * threadIdx.x; i + 3 < v.i1; i += 4 * VE_NT) body(i, true);
    const int tail = v.i0 + ((v.i1 - v.i0) & ~3);
    for (int i = tail + threadIdx.x; i < v.i1; i += VE_NT) body(i, false);
  } else {
    for (int i = v.i0 + threadIdx.x; i < v.i1; i += VE_NT) body(i, false);
  }
}

#define VE_LD4(p, i) (*reinterpret_cast<const float4*>((p) + (i)))
#define VE_ST4(p, i, v) (*reinterpret_cast<float4*>((p) + (i)) = (v))

__global__ __launch_bounds__(VE_NT) void van_bias_gelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                                  int C, int HW, int al, float* __restrict__ y) {
  const VeSlice v = ve_slice(C, HW);
  const float b = bias ? bias[v.c] : 0.f;
  const float* xp = x + v.base;
  float* yp = y + v.base;
  ve_loop(v, al, [&](int i, bool vec) {
    if (vec) {
      const float4 t = VE_LD4(xp, i);
      VE_ST4(yp, i, make_float4(ve_gelu(t.x + b), ve_gelu(t.y + b), ve_gelu(t.z + b), ve_gelu(t.w + b)));
    } else {
      yp[i] = ve_gelu(xp[i] + b);
    }
  });
}

__global__ __launch_bounds__(VE_NT) void van_bias_gelu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                  const float* __restrict__ bias, int C, int HW, int al,
                                                                  float* __restrict__ gx, float* __restrict__ partial) {
  const VeSlice v = ve_slice(C, HW);
  const float b = bias ? bias[v.c] : 0.f;
  const float *gp = gy + v.base, *xp = x + v.base;
  float* op = gx + v.base;
  float acc = 0.f;
  ve_loop(v, al, [&](int i, bool vec) {
    if (vec) {
      const float4 g = VE_LD4(gp, i), t = VE_LD4(xp, i);
      const float4 r = make_float4(g.x * ve_gelu_grad(t.x + b), g.y * ve_gelu_grad(t.y + b), g.z * ve_gelu_grad(t.z + b),
                                   g.w * ve_gelu_grad(t.w + b));
      VE_ST4(op, i, r);
      acc += (r.x + r.y) + (r.z + r.w);
    } else {
      const float r = gp[i] * ve_gelu_grad(xp[i] + b);
      op[i] = r;
      acc += r;
    }
  });
  if (partial) ve_reduce_store(acc, 0.f, C, partial);
}

__global__ __launch_bounds__(VE_NT) void van_gate_fwd_kernel(const float* __restrict__ u, const float* __restrict__ a,
                                                             const float* __restrict__ bias, int C, int HW, int al,
                                                             float* __restrict__ y) {
  const VeSlice v = ve_slice(C, HW);
  const float b = bias ? bias[v.c] : 0.f;
  const float *up = u + v.base, *ap = a + v.base;
  float* yp = y + v.base;
  ve_loop(v, al, [&](int i, bool vec) {
    if (vec) {
      const float4 p = VE_LD4(up, i), q = VE_LD4(ap, i);
      VE_ST4(yp, i, make_float4(p.x * (q.x + b), p.y * (q.y + b), p.z * (q.z + b), p.w * (q.w + b)));
    } else {
      yp[i] = up[i] * (ap[i] + b);
    }
  });
}

__global__ __launch_bounds__(VE_NT) void van_gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                             const float* __restrict__ a, const float* __restrict__ bias,
                                                             int C, int HW, int al, float* __restrict__ gu,
                                                             float* __restrict__ ga, float* __restrict__ partial) {
  const VeSlice v = ve_slice(C, HW);
  const float b = bias ? bias[v.c] : 0.f;
  const float *gp = g + v.base, *up = u + v.base, *ap = a + v.base;
  float *gup = gu + v.base, *gap = ga + v.base;
  float acc = 0.f;
  ve_loop(v, al, [&](int i, bool vec) {
    if (vec) {
      const float4 t = VE_LD4(gp, i), p = VE_LD4(up, i), q = VE_LD4(ap, i);
      const float4 r = make_float4(t.x * p.x, t.y * p.y, t.z * p.z, t.w * p.w);
      VE_ST4(gup, i, make_float4(t.x * (q.x + b), t.y * (q.y + b), t.z * (q.z + b), t.w * (q.w + b)));
      VE_ST4(gap, i, r);
      acc += (r.x + r.y) + (r.z + r.w);
    } else {
      const float r = gp[i] * up[i];
      gup[i] = gp[i] * (ap[i] + b);
      gap[i] = r;
      acc += r;
    }
  });
  if (partial) ve_reduce_store(acc, 0.f, C, partial);
}

__global__ __launch_bounds__(VE_NT) void van_residual_fwd_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                                                 const float* __restrict__ bias, const float* __restrict__ sc,
                                                                 const float* __restrict__ scale, int C, int HW, int al,
                                                                 float* __restrict__ y) {
  const VeSlice v = ve_slice(C, HW);
  const float b = bias ? bias[v.c] : 0.f, ls = scale[v.c];
  const float *xp = x + v.base, *pp = p + v.base, *sp = sc ? sc + v.base : nullptr;
  float* yp = y + v.base;
  ve_loop(v, al, [&](int i, bool vec) {
    if (vec) {
      const float4 t = VE_LD4(xp, i), q = VE_LD4(pp, i);
      const float4 s = sp ? VE_LD4(sp, i) : make_float4(0.f, 0.f, 0.f, 0.f);
      VE_ST4(yp, i, make_float4(t.x + ls * (q.x + b + s.x), t.y + ls * (q.y + b + s.y), t.z + ls * (q.z + b + s.z),
                                t.w + ls * (q.w + b + s.w)));
    } else {
      yp[i] = xp[i] + ls * (pp[i] + b + (sp ? sp[i] : 0.f));
    }
  });
}

__global__ __launch_bounds__(VE_NT) void van_residual_bwd_kernel(const float* __restrict__ g, const float* __restrict__ p,
                                                                 const float* __restrict__ bias, const float* __restrict__ sc,
                                                                 const float* __restrict__ scale, int C, int HW, int al,
                                                                 float* __restrict__ gpo, float* __restrict__ partial) {
  const VeSlice v = ve_slice(C, HW);
  const float b = bias ? bias[v.c] : 0.f, ls = scale[v.c];
  const float *gp = g + v.base, *pp = p + v.base, *sp = sc ? sc + v.base : nullptr;
  float* op = gpo + v.base;
  float acc_g = 0.f, acc_f = 0.f;
  ve_loop(v, al, [&](int i, bool vec) {
    if (vec) {
      const float4 t = VE_LD4(gp, i), q = VE_LD4(pp, i);
      const float4 s = sp ? VE_LD4(sp, i) : make_float4(0.f, 0.f, 0.f, 0.f);
      VE_ST4(op, i, make_float4(ls * t.x, ls * t.y, ls * t.z, ls * t.w));
      acc_g += (t.x + t.y) + (t.z + t.w);
      acc_f += (t.x * (q.x + b + s.x) + t.y * (q.y + b + s.y)) + (t.z * (q.z + b + s.z) + t.w * (q.w + b + s.w));
    } else {
      const float t = gp[i];
      op[i] = ls * t;
      acc_g += t;
      acc_f += t * (pp[i] + b + (sp ? sp[i] : 0.f));
    }
  });
  ve_reduce_store(acc_g, acc_f, C, partial);
}

// one wave per channel: out0[c] = mul0[c] * sum of the first partials, out1[c] = sum of the second ones (fixed order)
__global__ __launch_bounds__(256) void van_finish_kernel(const float* __restrict__ partial, int C, int S,
                                                         const float* __restrict__ mul0, float* __restrict__ out0,
                                                         float* __restrict__ out1) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
#pragma unroll 8        // (eight loads in flight; the additions keep their order)
  for (int s = lane; s < S; s += 64) {
    const float2 p = *reinterpret_cast<const float2*>(partial + ((long long)c * S + s) * 2);
    a += p.x;
    b += p.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
  }
  if (lane == 0) {
    if (out0) out0[c] = mul0 ? mul0[c] * a : a;
    if (out1) out1[c] = b;
  }
}

static inline int ve_slices(int N, int C, int HW) {
  const long long planes = (long long)N * C;
  long long s = (2048 + planes - 1) / planes;          // enough workgroups for 256 CUs ...
  const long long cap = HW / 2048 > 1 ? HW / 2048 : 1;  // ... of at least 2 048 elements each
  if (s > cap) s = cap;
  return (int)(s < 1 ? 1 : s);
}

static inline bool ve_ok(int N, int C, int HW) {
  return N >= 0 && C > 0 && HW >= 0 && (long long)N * C <= 65535;
}

static inline int ve_aligned(int HW, std::initializer_list<const void*> ptrs) {
  if (HW & 3) return 0;
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & 15)) return 0;
  return 1;
}


// ---- LayerNorm over the CHANNELS of an NCHW map (the norm at the end of every VAN stage, van.py:303-306) -------------
// The reference flattens the map to (B, HW, C), normalises the last axis and permutes back: as tensor operations that is a
// strided copy in, the library LayerNorm, a strided copy out, and the same again (plus two partial-sum kernels) in the
// backward -- 1.0 ms per Oriented R-CNN step for four calls.  Here a workgroup owns 64 consecutive pixels of one image and
// ALL their channels: lanes = pixels (every channel row is a coalesced 256-byte read), the four waves split the channels.
// Forward: the tile is staged in LDS (C x 64 floats, <= 128 KB), mean and variance in two passes over it (biased variance,
// rsqrt(var + eps): torch's native_layer_norm), one global read and one write of the map.  Backward: the two per-pixel sums
// (sum_c g gamma, sum_c g gamma xhat) in a first pass, the input gradient and the per-channel sums for gamma / beta in a
// second (the tile's second read comes from L2); (C, S, 2) partials for rsdet_launch_sums_finish.
constexpr int LN_PT = 64, LN_NT = 256, LN_CG = LN_NT / LN_PT;

__global__ __launch_bounds__(LN_NT) void ln_chan_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int C, int HW, float eps,
                                                            float* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd) {
  extern __shared__ float ln_tile[];  // [C][LN_PT]
  __shared__ float s_red[LN_CG][LN_PT];
  const int px = threadIdx.x & (LN_PT - 1), cg = threadIdx.x >> 6, n = blockIdx.y;
  const int p = blockIdx.x * LN_PT + px;
  const bool ok = p < HW;
  const float* xb = x + (long long)n * C * HW + (ok ? p : 0);
  float s = 0.f;
#pragma unroll 8
  for (int c = cg; c < C; c += LN_CG) {
    const float v = ok ? xb[(long long)c * HW] : 0.f;
    ln_tile[c * LN_PT + px] = v;
    s += v;
  }
  s_red[cg][px] = s;
  __syncthreads();
  const float mu = ((s_red[0][px] + s_red[1][px]) + (s_red[2][px] + s_red[3][px])) / (float)C;
  __syncthreads();
  float q = 0.f;
#pragma unroll 8
  for (int c = cg; c < C; c += LN_CG) {
    const float d = ln_tile[c * LN_PT + px] - mu;
    q += d * d;
  }
  s_red[cg][px] = q;
  __syncthreads();
  const float var = ((s_red[0][px] + s_red[1][px]) + (s_red[2][px] + s_red[3][px])) / (float)C;
  const float r = rsqrtf(var + eps);
  if (!ok) return;
  float* yb = y + (long long)n * C * HW + p;
#pragma unroll 8
  for (int c = cg; c < C; c += LN_CG)
    yb[(long long)c * HW] = (ln_tile[c * LN_PT + px] - mu) * r * gamma[c] + beta[c];
  if (cg == 0) {
    mean[(long long)n * HW + p] = mu;
    rstd[(long long)n * HW + p] = r;
  }
}

__device__ __forceinline__ float ln_row16_sum(float v) {
#define LN_DPP(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  v += LN_DPP(0xB1);   // quad_perm [1, 0, 3, 2]
  v += LN_DPP(0x4E);   // quad_perm [2, 3, 0, 1]
  v += LN_DPP(0x141);  // row_half_mirror
  v += LN_DPP(0x140);  // row_mirror
#undef LN_DPP
  return v;
}

__global__ __launch_bounds__(LN_NT) void ln_chan_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, int C, int HW,
                                                            float* __restrict__ gx, float* __restrict__ partial) {
  __shared__ float s_a[LN_CG][LN_PT], s_b[LN_CG][LN_PT];
  const int px = threadIdx.x & (LN_PT - 1), cg = threadIdx.x >> 6, n = blockIdx.y;
  const int p = blockIdx.x * LN_PT + px;
  const bool ok = p < HW;
  const long long base = (long long)n * C * HW + (ok ? p : 0);
  const float mu = ok ? mean[(long long)n * HW + p] : 0.f, r = ok ? rstd[(long long)n * HW + p] : 0.f;
  float a = 0.f, b = 0.f;
#pragma unroll 8
  for (int c = cg; c < C; c += LN_CG) {
    const float gv = ok ? g[base + (long long)c * HW] * gamma[c] : 0.f;
    const float xh = ok ? (x[base + (long long)c * HW] - mu) * r : 0.f;
    a += gv;
    b += gv * xh;
  }
  s_a[cg][px] = a, s_b[cg][px] = b;
  __syncthreads();
  a = ((s_a[0][px] + s_a[1][px]) + (s_a[2][px] + s_a[3][px])) / (float)C;
  b = ((s_b[0][px] + s_b[1][px]) + (s_b[2][px] + s_b[3][px])) / (float)C;
  const int S = gridDim.x * gridDim.y, slot = n * gridDim.x + blockIdx.x;
  for (int c = cg; c < C; c += LN_CG) {            // (a wave = one channel group: its 64 lanes are the tile's pixels)
    const float gr = ok ? g[base + (long long)c * HW] : 0.f;
    const float xh = ok ? (x[base + (long long)c * HW] - mu) * r : 0.f;
    if (ok) gx[base + (long long)c * HW] = r * (gr * gamma[c] - (a + xh * b));
    // rows of 16 lanes summed on the vector pipe (four DPP adds); the four row sums of the tile are four slots of the
    // partial table (a wave butterfly is six dependent LDS-crossbar round trips per value: most of this kernel's time)
    const float db = ln_row16_sum(gr), dg = ln_row16_sum(gr * xh);
    if ((px & 15) == 0)
      *reinterpret_cast<float2*>(partial + ((long long)c * (4 * S) + 4 * slot + (px >> 4)) * 2) = make_float2(db, dg);
  }
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_van_supported(int N, int C, int HW) { return ve_ok(N, C, HW) ? 1 : 0; }

extern "C" size_t rsdet_van_ws_size(int N, int C, int HW) {
  if (!ve_ok(N, C, HW)) return 0;
  return (size_t)C * N * ve_slices(N, C, HW) * 2 * sizeof(float);
}

extern "C" int rsdet_van_bias_gelu_fwd_f32(const float* x, const float* bias, int N, int C, int HW, float* y,
                                           void* stream) {
  if (!ve_ok(N, C, HW)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!x || !y) return RSDET_EINVAL;
  hipLaunchKernelGGL(van_bias_gelu_fwd_kernel, dim3(ve_slices(N, C, HW), N * C), dim3(VE_NT), 0, (hipStream_t)stream, x,
                     bias, C, HW, ve_aligned(HW, {x, y}), y);
  return rsdet_launch_status();
}

extern "C" int rsdet_van_bias_gelu_bwd_f32(const float* gy, const float* x, const float* bias, int N, int C, int HW,
                                           float* gx, float* gbias, void* ws, size_t ws_bytes, void* stream) {
  if (!ve_ok(N, C, HW)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) {
    if (gbias) (void)hipMemsetAsync(gbias, 0, sizeof(float) * C, (hipStream_t)stream);
    return RSDET_OK;
  }
  if (!gy || !x || !gx) return RSDET_EINVAL;
  if (gbias && (!ws || ws_bytes < rsdet_van_ws_size(N, C, HW))) return RSDET_EINVAL;
  const int S = ve_slices(N, C, HW);
  hipLaunchKernelGGL(van_bias_gelu_bwd_kernel, dim3(S, N * C), dim3(VE_NT), 0, (hipStream_t)stream, gy, x, bias, C, HW,
                     ve_aligned(HW, {gy, x, gx}), gx, gbias ? (float*)ws : nullptr);
  if (gbias)
    hipLaunchKernelGGL(van_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)ws, C,
                       N * S, (const float*)nullptr, gbias, (float*)nullptr);
  return rsdet_launch_status();
}

extern "C" int rsdet_van_gate_fwd_f32(const float* u, const float* a, const float* bias, int N, int C, int HW, float* y,
                                      void* stream) {
  if (!ve_ok(N, C, HW)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!u || !a || !y) return RSDET_EINVAL;
  hipLaunchKernelGGL(van_gate_fwd_kernel, dim3(ve_slices(N, C, HW), N * C), dim3(VE_NT), 0, (hipStream_t)stream, u, a,
                     bias, C, HW, ve_aligned(HW, {u, a, y}), y);
  return rsdet_launch_status();
}

extern "C" int rsdet_van_gate_bwd_f32(const float* g, const float* u, const float* a, const float* bias, int N, int C,
                                      int HW, float* gu, float* ga, float* gbias, void* ws, size_t ws_bytes,
                                      void* stream) {
  if (!ve_ok(N, C, HW)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) {
    if (gbias) (void)hipMemsetAsync(gbias, 0, sizeof(float) * C, (hipStream_t)stream);
    return RSDET_OK;
  }
  if (!g || !u || !a || !gu || !ga) return RSDET_EINVAL;
  if (gbias && (!ws || ws_bytes < rsdet_van_ws_size(N, C, HW))) return RSDET_EINVAL;
  const int S = ve_slices(N, C, HW);
  hipLaunchKernelGGL(van_gate_bwd_kernel, dim3(S, N * C), dim3(VE_NT), 0, (hipStream_t)stream, g, u, a, bias, C, HW,
                     ve_aligned(HW, {g, u, a, gu, ga}), gu, ga, gbias ? (float*)ws : nullptr);
  if (gbias)
    hipLaunchKernelGGL(van_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)ws, C,
                       N * S, (const float*)nullptr, gbias, (float*)nullptr);
  return rsdet_launch_status();
}

extern "C" int rsdet_van_residual_fwd_f32(const float* x, const float* p, const float* bias, const float* shortcut,
                                          const float* scale, int N, int C, int HW, float* y, void* stream) {
  if (!ve_ok(N, C, HW)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) return RSDET_OK;
  if (!x || !p || !scale || !y) return RSDET_EINVAL;
  hipLaunchKernelGGL(van_residual_fwd_kernel, dim3(ve_slices(N, C, HW), N * C), dim3(VE_NT), 0, (hipStream_t)stream, x,
                     p, bias, shortcut, scale, C, HW, ve_aligned(HW, {x, p, shortcut, y}), y);
  return rsdet_launch_status();
}

extern "C" int rsdet_van_residual_bwd_f32(const float* g, const float* p, const float* bias, const float* shortcut,
                                          const float* scale, int N, int C, int HW, float* gp, float* gbias,
                                          float* gscale, void* ws, size_t ws_bytes, void* stream) {
  if (!ve_ok(N, C, HW)) return RSDET_EINVAL;
  if (N == 0 || HW == 0) {
    if (gbias) (void)hipMemsetAsync(gbias, 0, sizeof(float) * C, (hipStream_t)stream);
    if (gscale) (void)hipMemsetAsync(gscale, 0, sizeof(float) * C, (hipStream_t)stream);
    return RSDET_OK;
  }
  if (!g || !p || !scale || !gp || !ws || ws_bytes < rsdet_van_ws_size(N, C, HW)) return RSDET_EINVAL;
  const int S = ve_slices(N, C, HW);
  hipLaunchKernelGGL(van_residual_bwd_kernel, dim3(S, N * C), dim3(VE_NT), 0, (hipStream_t)stream, g, p, bias, shortcut,
                     scale, C, HW, ve_aligned(HW, {g, p, shortcut, gp}), gp, (float*)ws);
  if (gbias || gscale)
    hipLaunchKernelGGL(van_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)ws, C,
                       N * S, scale, gbias, gscale);
  return rsdet_launch_status();
}

// ---- LayerNorm over the channels of an NCHW map -----------------------------------------------------------------------
extern "C" int rsdet_chan_layernorm_supported(int N, int C, int HW) {
  return (N >= 1 && C >= 4 && C <= 512 && HW >= 1 && (long long)N * C * HW < (1ll << 31)) ? 1 : 0;
}
extern "C" size_t rsdet_chan_layernorm_ws_size(int N, int C, int HW) {
  if (!rsdet_chan_layernorm_supported(N, C, HW)) return 0;
  return (size_t)C * ((size_t)N * ((HW + LN_PT - 1) / LN_PT)) * 4 * 2 * sizeof(float);      // four row slots per tile
}
extern "C" int rsdet_chan_layernorm_forward_f32(const float* x, const float* gamma, const float* beta, int N, int C, int HW,
                                                float eps, float* y, float* mean, float* rstd, void* stream) {
  if (!rsdet_chan_layernorm_supported(N, C, HW) || !x || !gamma || !beta || !y || !mean || !rstd) return RSDET_EINVAL;
  const size_t lds = (size_t)C * LN_PT * sizeof(float);
  static size_t lds_set = 0;
  if (lds > 65536 && lds > lds_set) {
    if (hipFuncSetAttribute((const void*)ln_chan_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * LN_PT * 4) !=
        hipSuccess)
      return RSDET_ELAUNCH;
    lds_set = 512 * LN_PT * 4;
  }
  hipLaunchKernelGGL(ln_chan_fwd_kernel, dim3((HW + LN_PT - 1) / LN_PT, N), dim3(LN_NT), lds, (hipStream_t)stream, x, gamma,
                     beta, C, HW, eps, y, mean, rstd);
  return rsdet_launch_status();
}
extern "C" int rsdet_chan_layernorm_backward_f32(const float* grad_y, const float* x, const float* mean, const float* rstd,
                                                 const float* gamma, int N, int C, int HW, float* grad_x, float* grad_gamma,
                                                 float* grad_beta, void* ws, size_t ws_bytes, void* stream) {
  if (!rsdet_chan_layernorm_supported(N, C, HW) || !grad_y || !x || !mean || !rstd || !gamma || !grad_x || !ws ||
      ws_bytes < rsdet_chan_layernorm_ws_size(N, C, HW))
    return RSDET_EINVAL;
  const int bx = (HW + LN_PT - 1) / LN_PT;
  hipLaunchKernelGGL(ln_chan_bwd_kernel, dim3(bx, N), dim3(LN_NT), 0, (hipStream_t)stream, grad_y, x, mean, rstd, gamma, C, HW,
                     grad_x, (float*)ws);
  if (grad_gamma || grad_beta)
    rsdet_launch_sums_finish((const float*)ws, C, 4 * bx * N, grad_gamma, grad_beta, (hipStream_t)stream);
  return rsdet_launch_status();
}
