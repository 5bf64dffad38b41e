// dwconv.hip -- depthwise 2-D convolution ("same" padding, stride 1) forward / backward for gfx950.
//
// Replaces, on the Oriented R-CNN + VAN path, the depthwise convolutions of the VAN backbone:
//   /root/reference/python/jdet/models/backbones/van.py:32 (DWConv 3x3 on the MLP's hidden width),
//   :56 (LKA conv0 5x5), :57 (LKA conv_spatial 7x7, dilation 3)  -- nn.Conv2d(dim, dim, k, groups=dim).
// MIOpen serves these fp32 shapes with its naive direct kernels and im2col (`naive_conv_ab_nonpacked_*`,
// `Im2d2Col_v2`: 36 of 143 ms of kernel time per VAN-B3 step), torch's own depthwise kernels take another 12 ms.
// The op is a stencil: every input element is needed K*K times by neighbouring outputs of ONE channel plane and by
// nothing else, so it is HBM-bound if a plane tile is staged once in LDS.
//
//   forward / backward-data   one workgroup per (plane, 32 x 64 output tile): the tile plus its halo (up to 9 on each
//                             side for 7x7 dilation 3) goes to LDS with coalesced loads; a thread owns one column and
//                             8 rows, walks the K kernel columns, keeps the column window in registers and reuses
//                             it for the K vertical taps.  backward-data is the same stencil with the taps
//                             mirrored (same padding => same geometry).
//   backward-weight           same staging of x; a thread holds its 8 output-gradient values and accumulates
//                             K*K (+1 for the bias) partial sums, reduced over the workgroup (wave shuffles + LDS)
//                             into a partial row per workgroup, then summed in a fixed order by a second kernel
//                             (deterministic, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"

namespace rsdet {

constexpr int DW_TH = 32, DW_TW = 64, DW_NT = 256, DW_ROWS = DW_TH / (DW_NT / DW_TW);  // 8 rows per thread

template <int K, int D>
struct DwGeom {
  static constexpr int halo = D * (K - 1) / 2;
  static constexpr int LH = DW_TH + 2 * halo, LW = DW_TW + 2 * halo;
  static constexpr int LWP = LW | 1;  // odd row pitch: rows of a column land on different banks
  static constexpr int WIN = DW_ROWS + (K - 1) * D;
  // the kernel-column loop stays rolled for 7x7: unrolled, the compiler keeps all seven 26-value column windows live
  // (235 VGPRs, two waves per SIMD); rolled it is one window at a time
  static constexpr int KW_UNROLL = K >= 5 ? 1 : K;
};

// coalesced staging of the (tile + halo) window of one plane, zero outside the plane
template <int K, int D>
__device__ __forceinline__ void dw_stage(const float* __restrict__ plane, int H, int W, int y0, int x0,
                                         float* __restrict__ s, float add = 0.f) {
  // `add`: a per-channel constant the PRODUCER of this tensor owed it (the bias of the 1x1 convolution in front of
  // the depthwise one, ops/dwconv.py in_bias): added to the in-bounds elements only, the zero padding stays zero
  using G = DwGeom<K, D>;
  // every load of the window is ISSUED before the first LDS write: the rolled form of this loop (9 .. 22 trips of
  // load -> wait -> write per thread) paid one global-memory latency per trip, which is what held the stencil kernels
  // at 0.26-0.40 of the HBM roofline (round 5: 327 instructions in the 3x3 kernel, TWO of them loads)
  constexpr int TOTAL = G::LH * G::LW, TRIPS = (TOTAL + DW_NT - 1) / DW_NT;
  float v[TRIPS];
#pragma unroll
  for (int k = 0; k < TRIPS; ++k) {
    const int i = threadIdx.x + k * DW_NT;
    const int r = i / G::LW, c = i - r * G::LW;
    const int y = y0 - G::halo + r, x = x0 - G::halo + c;
    // (an UNCONDITIONAL load from a clamped address + a select: a conditional load is a branch and a wait per trip)
    const bool in = i < TOTAL && y >= 0 && y < H && x >= 0 && x < W;
    const float t = plane[in ? (long long)y * W + x : 0];
    v[k] = in ? t + add : 0.f;
  }
#pragma unroll
  for (int k = 0; k < TRIPS; ++k) {
    const int i = threadIdx.x + k * DW_NT;
    const int r = i / G::LW, c = i - r * G::LW;
    if (i < TOTAL) s[r * G::LWP + c] = v[k];
  }
}

// grid: (N * C planes, tiles_x * tiles_y).  Neighbouring tiles of a plane are N * C workgroup ids apart -- a multiple
// of 8 for every VAN width, i.e. on the same XCD, where their shared halo rows meet in L2.  FLIP mirrors the taps
// (backward-data).
// in_bias (forward only): see dw_stage.  out_sum (backward-data only): per-workgroup sum of the tile it wrote, i.e. a
// partial of sum(grad_x) per channel = the gradient of that producer bias; out_sum[(c * nslots + slot)].
__device__ __forceinline__ float dw_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dw_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

template <int K, int D, bool FLIP>
__global__ __launch_bounds__(DW_NT) void dwconv_stencil_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ in_bias, int C, int H, int W,
                                                               int tiles_x, float* __restrict__ y,
                                                               float* __restrict__ out_sum,
                                                               float* __restrict__ y_act = nullptr,
                                                               const float* __restrict__ ep_add = nullptr,
                                                               const float* __restrict__ ep_gelu = nullptr,
                                                               int y_is_grad = 0) {
  // y_act (forward): a second output GELU(y) (Mlp.act behind Mlp.dwconv, van.py:140-175).  ep_add / ep_gelu (backward-data):
  // the value written (and summed) is (acc + ep_add) * GELU'(ep_gelu) -- the gradient through u = GELU(t1) of the two
  // branches that read u (the depthwise pair and the gate), ops/van_block.py.
  using G = DwGeom<K, D>;
  __shared__ float s[G::LH * G::LWP];
  __shared__ float s_w[K * K];
  __shared__ float s_sum[DW_NT / 64];
  const int plane = blockIdx.x, c = plane % C;
  const int ty0 = (blockIdx.y / tiles_x) * DW_TH, tx0 = (blockIdx.y % tiles_x) * DW_TW;
  if (threadIdx.x < K * K) s_w[threadIdx.x] = w[c * K * K + (FLIP ? K * K - 1 - threadIdx.x : threadIdx.x)];
  dw_stage<K, D>(x + (long long)plane * H * W, H, W, ty0, tx0, s, (!FLIP && in_bias) ? in_bias[c] : 0.f);
  __syncthreads();
  const int tx = threadIdx.x % DW_TW, tr = (threadIdx.x / DW_TW) * DW_ROWS;
  float acc[DW_ROWS];
  const float b = (!FLIP && bias) ? bias[c] : 0.f;
#pragma unroll
  for (int r = 0; r < DW_ROWS; ++r) acc[r] = b;
#pragma unroll G::KW_UNROLL
  for (int kw = 0; kw < K; ++kw) {
    float win[G::WIN];
#pragma unroll
    for (int i = 0; i < G::WIN; ++i) win[i] = s[(tr + i) * G::LWP + tx + kw * D];
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const float wv = s_w[kh * K + kw];
#pragma unroll
      for (int r = 0; r < DW_ROWS; ++r) acc[r] = __builtin_fmaf(wv, win[r + kh * D], acc[r]);   // (the TU is built -ffp-contract=off)
    }
  }
  const int ox = tx0 + tx;
  float tile_sum = 0.f;
  if (ox < W) {
    float* yp = y + (long long)plane * H * W;
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r) {
      const int oy = ty0 + tr + r;
      if (oy < H) {
        const long long o = (long long)oy * W + ox;
        float v = acc[r];
        if (FLIP && ep_gelu) v = (v + ep_add[(long long)plane * H * W + o]) * dw_gelu_grad(ep_gelu[(long long)plane * H * W + o]);
        if (!FLIP && y_act) {
          y_act[(long long)plane * H * W + o] = dw_gelu(v);
          yp[o] = y_is_grad ? dw_gelu_grad(v) : v;
        } else {
          yp[o] = v;
        }
        tile_sum += v;
      }
    }
  }
  if (FLIP && out_sum) {  // uniform branch: fixed-order reduction of the tile (wave butterflies, then four waves)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tile_sum += __shfl_xor(tile_sum, off);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = tile_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = plane / C, nslots = (gridDim.x / C) * gridDim.y;
      out_sum[(long long)c * nslots + n * gridDim.y + blockIdx.y] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    }
  }
}

// Sum over each row of 16 lanes, left in every lane of the row: four DPP adds (quad swaps, half-row and row mirrors) on the
// vector pipe.  The butterfly over the whole wave (six dependent ds_bpermute round trips per value, ~400 cycles each chain)
// was what the 7x7 weight gradient spent its time in: 49 taps x 6 of them per thread and tile.
__device__ __forceinline__ float dw_row16_sum(float v) {
#define DW_DPP(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  v += DW_DPP(0xB1);   // quad_perm [1, 0, 3, 2]
  v += DW_DPP(0x4E);   // quad_perm [2, 3, 0, 1]
  v += DW_DPP(0x141);  // row_half_mirror
  v += DW_DPP(0x140);  // row_mirror
#undef DW_DPP
  return v;
}

// partial[(c * nslots + slot) * (K*K + 1) + t], slot = n * gridDim.y + blockIdx.y; t = K*K is the bias gradient.
// A workgroup walks `tpw` tiles of ONE plane and keeps its K*K + 1 sums in REGISTERS across them (one per thread and
// tap); the cross-lane reduction (6 butterfly steps per tap) runs ONCE per workgroup at the end.  The first form
// reduced after every tile: 49 taps x 6 ds_bpermute per thread and tile made the 7x7 / 5x5 gradients of VAN's
// attention branch instruction-bound at 0.1 of the HBM rate (profiles/r04_e_bench_kernels.json).
template <int K, int D>
__global__ __launch_bounds__(DW_NT) void dwconv_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                             const float* __restrict__ in_bias, int C, int H, int W,
                                                             int tiles_x, int ntiles, int tpw, int nslots,
                                                             float* __restrict__ partial) {
  using G = DwGeom<K, D>;
  constexpr int T = K * K + 1;
  __shared__ float s[G::LH * G::LWP];
  __shared__ float s_red[DW_NT / 16][T];
  const int plane = blockIdx.x, c = plane % C, n = plane / C;
  const int tx = threadIdx.x % DW_TW, tr = (threadIdx.x / DW_TW) * DW_ROWS;
  const float add = in_bias ? in_bias[c] : 0.f;
  const float* xp = x + (long long)plane * H * W;
  const float* gp = gy + (long long)plane * H * W;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  const int t_begin = blockIdx.y * tpw, t_end = min(ntiles, t_begin + tpw);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int ty0 = (tile / tiles_x) * DW_TH, tx0 = (tile % tiles_x) * DW_TW;
    if (tile != t_begin) __syncthreads();                      // the previous tile's window reads are done
    dw_stage<K, D>(xp, H, W, ty0, tx0, s, add);
    float g[DW_ROWS];
    {
      const int ox = tx0 + tx;
#pragma unroll
      for (int r = 0; r < DW_ROWS; ++r) {
        const int oy = ty0 + tr + r;
        g[r] = (ox < W && oy < H) ? gp[(long long)oy * W + ox] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r) acc[K * K] += g[r];
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      float win[G::WIN];
#pragma unroll
      for (int i = 0; i < G::WIN; ++i) win[i] = s[(tr + i) * G::LWP + tx + kw * D];
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < DW_ROWS; ++r) v = __builtin_fmaf(g[r], win[r + kh * D], v);
        acc[kh * K + kw] += v;
      }
      if (K >= 7) asm volatile("" ::: "memory");               // one column window live at a time (register budget)
    }
  }
  // one reduction per workgroup: rows of 16 lanes on the vector pipe (dw_row16_sum), then the sixteen rows through LDS
  // in a fixed order
  const int row16 = threadIdx.x >> 4;
  const bool lead = (threadIdx.x & 15) == 0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float v = dw_row16_sum(acc[t]);
    if (lead) s_red[row16][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < T) {
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < DW_NT / 16; ++wv) v += s_red[wv][threadIdx.x];
    const int slot = n * gridDim.y + blockIdx.y;
    partial[((long long)c * nslots + slot) * T + threadIdx.x] = v;
  }
}

// The one-tile form (a workgroup = one tile, the cross-lane reduction after it): what the 7x7 / dilation-3 layers keep --
// their K*K = 49 running sums in registers (98 VGPRs, the kernel-column loop unrolled for static indices) cost more in
// occupancy than the amortised reductions return (VAN-B3 step, rocprofv3: 38.3 us per call in this form, 46.3 in the
// multi-tile one).  partial[(c * nslots + slot) * (K*K + 1) + t], slot = n * tiles + tile; t = K*K is the bias gradient
template <int K, int D>
__global__ __launch_bounds__(DW_NT) void dwconv_wgrad_tile_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                             const float* __restrict__ in_bias, int C, int H, int W,
                                                             int tiles_x, int nslots, float* __restrict__ partial) {
  using G = DwGeom<K, D>;
  constexpr int T = K * K + 1;
  __shared__ float s[G::LH * G::LWP];
  __shared__ float s_red[DW_NT / 16][T];  // one partial per row of 16 lanes
  const int plane = blockIdx.x, c = plane % C, n = plane / C;
  const int ty0 = (blockIdx.y / tiles_x) * DW_TH, tx0 = (blockIdx.y % tiles_x) * DW_TW;
  dw_stage<K, D>(x + (long long)plane * H * W, H, W, ty0, tx0, s, in_bias ? in_bias[c] : 0.f);
  const int tx = threadIdx.x % DW_TW, tr = (threadIdx.x / DW_TW) * DW_ROWS;
  float g[DW_ROWS];
  {
    const int ox = tx0 + tx;
    const float* gp = gy + (long long)plane * H * W;
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r) {
      const int oy = ty0 + tr + r;
      g[r] = (ox < W && oy < H) ? gp[(long long)oy * W + ox] : 0.f;
    }
  }
  __syncthreads();
  // per kernel column: K partial sums (one per kernel row) over this thread's 8 outputs, reduced over each row of 16
  // lanes (dw_row16_sum) and parked in LDS; the bias gradient rides along as one more value
  const int row16 = threadIdx.x >> 4;
  const bool lead = (threadIdx.x & 15) == 0;
  {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r) v += g[r];
    v = dw_row16_sum(v);
    if (lead) s_red[row16][K * K] = v;
  }
#pragma unroll 1
  for (int kw = 0; kw < K; ++kw) {
    float win[G::WIN];
#pragma unroll
    for (int i = 0; i < G::WIN; ++i) win[i] = s[(tr + i) * G::LWP + tx + kw * D];
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < DW_ROWS; ++r) v = __builtin_fmaf(g[r], win[r + kh * D], v);
      v = dw_row16_sum(v);
      if (lead) s_red[row16][kh * K + kw] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < T) {
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < DW_NT / 16; ++wv) v += s_red[wv][threadIdx.x];
    const int slot = n * gridDim.y + blockIdx.y;
    partial[((long long)c * nslots + slot) * T + threadIdx.x] = v;
  }
}

// ... the same for up to four layers in one launch (blockIdx.y = layer): the finishing passes of a block's depthwise weight
// gradients are 5 us launches with a few KB of work each; the VAN block node issues them together behind its backward
struct DwFinishJobs {
  const float* partial[4];
  float* gw[4];
  float* gb[4];
  int nslots[4], T[4], C[4];
};
__global__ __launch_bounds__(64) void dwconv_wgrad_finish_multi_kernel(DwFinishJobs j) {
  const int job = blockIdx.y, c = blockIdx.x, t = threadIdx.x, T = j.T[job], nslots = j.nslots[job];
  if (c >= j.C[job] || t >= T) return;
  const float* p = j.partial[job] + (long long)c * nslots * T + t;
  float v = 0.f;
#pragma unroll 8
  for (int sl = 0; sl < nslots; ++sl) v += p[(long long)sl * T];
  if (t < T - 1)
    j.gw[job][c * (T - 1) + t] = v;
  else if (j.gb[job])
    j.gb[job][c] = v;
}

// one wave per channel: fixed-order sum of its nslots partial rows
__global__ __launch_bounds__(64) void dwconv_wgrad_finish_kernel(const float* __restrict__ partial, int nslots, int T,
                                                                 float* __restrict__ gw, float* __restrict__ gb) {
  const int c = blockIdx.x, t = threadIdx.x;
  if (t >= T) return;
  const float* p = partial + (long long)c * nslots * T + t;
  float v = 0.f;
#pragma unroll 8        // (eight loads in flight; the additions keep their order)
  for (int sl = 0; sl < nslots; ++sl) v += p[(long long)sl * T];
  if (t < T - 1)
    gw[c * (T - 1) + t] = v;
  else if (gb)
    gb[c] = v;
}

}  // namespace rsdet

using namespace rsdet;

static int dw_check(int N, int C, int H, int W, int K, int dil) {
  if (N < 0 || C < 0 || H < 1 || W < 1) return RSDET_EINVAL;
  if (!((K == 3 && dil == 1) || (K == 5 && dil == 1) || (K == 7 && dil == 3))) return RSDET_EINVAL;
  if ((long long)N * C > 0x7fffffffLL) return RSDET_EINVAL;
  if ((long long)((W + DW_TW - 1) / DW_TW) * ((H + DW_TH - 1) / DW_TH) > 65535) return RSDET_EINVAL;  // gridDim.y
  return RSDET_OK;
}

template <bool FLIP>
static int dw_stencil(const float* x, const float* w, const float* bias, const float* in_bias, int N, int C, int H,
                      int W, int K, int dil, float* y, float* out_sum, hipStream_t s, float* y_act = nullptr,
                      const float* ep_add = nullptr, const float* ep_gelu = nullptr, int y_is_grad = 0) {
  const int tx = (W + DW_TW - 1) / DW_TW, ty = (H + DW_TH - 1) / DW_TH;
  const dim3 grid(N * C, tx * ty);
  if (K == 3)
    hipLaunchKernelGGL((dwconv_stencil_kernel<3, 1, FLIP>), grid, dim3(DW_NT), 0, s, x, w, bias, in_bias, C, H, W, tx,
                       y, out_sum, y_act, ep_add, ep_gelu, y_is_grad);
  else if (K == 5)
    hipLaunchKernelGGL((dwconv_stencil_kernel<5, 1, FLIP>), grid, dim3(DW_NT), 0, s, x, w, bias, in_bias, C, H, W, tx,
                       y, out_sum, y_act, ep_add, ep_gelu, y_is_grad);
  else
    hipLaunchKernelGGL((dwconv_stencil_kernel<7, 3, FLIP>), grid, dim3(DW_NT), 0, s, x, w, bias, in_bias, C, H, W, tx,
                       y, out_sum, y_act, ep_add, ep_gelu, y_is_grad);
  (void)dil;
  return rsdet_launch_status();
}

extern "C" int rsdet_dwconv2d_forward_f32(const float* x, const float* in_bias, const float* weight, const float* bias,
                                          int N, int C, int H, int W, int K, int dilation, float* y, void* stream) {
  int rc = dw_check(N, C, H, W, K, dilation);
  if (rc) return rc;
  if (N == 0 || C == 0) return RSDET_OK;
  if (!x || !weight || !y) return RSDET_EINVAL;
  return dw_stencil<false>(x, weight, bias, in_bias, N, C, H, W, K, dilation, y, nullptr, (hipStream_t)stream);
}

// ... with a second output y_act = GELU(y) (erf form): Mlp.dwconv + Mlp.act as one pass (van.py:169-171)
extern "C" int rsdet_dwconv2d_forward_act_f32(const float* x, const float* weight, const float* bias, int N, int C, int H,
                                              int W, int K, int dilation, int y_is_grad, float* y, float* y_act,
                                              void* stream) {
  int rc = dw_check(N, C, H, W, K, dilation);
  if (rc) return rc;
  if (N == 0 || C == 0) return RSDET_OK;
  if (!x || !weight || !y || !y_act) return RSDET_EINVAL;
  return dw_stencil<false>(x, weight, bias, nullptr, N, C, H, W, K, dilation, y, nullptr, (hipStream_t)stream, y_act, nullptr,
                           nullptr, y_is_grad);
}

// floats of workspace per output of the reductions: one partial per (plane, tile)
static inline size_t dw_slots(int N, int C, int H, int W) {
  return (size_t)C * N * ((W + DW_TW - 1) / DW_TW) * ((H + DW_TH - 1) / DW_TH);
}

extern "C" size_t rsdet_dwconv2d_backward_data_ws_size(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H < 1 || W < 1) return 0;
  return dw_slots(N, C, H, W) * sizeof(float);
}

extern "C" int rsdet_dwconv2d_backward_data_f32(const float* grad_y, const float* weight, int N, int C, int H, int W,
                                                int K, int dilation, float* grad_x, float* grad_in_bias, void* ws,
                                                size_t ws_bytes, void* stream) {
  int rc = dw_check(N, C, H, W, K, dilation);
  if (rc) return rc;
  if (C == 0) return RSDET_OK;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    if (grad_in_bias && hipMemsetAsync(grad_in_bias, 0, (size_t)C * 4, s) != hipSuccess) return RSDET_ELAUNCH;
    return RSDET_OK;
  }
  if (!grad_y || !weight || !grad_x) return RSDET_EINVAL;
  // (ws without grad_in_bias: the per-tile sums stay in ws as [c][slot] floats, slot count = ws_size / 4 / C, for a consumer
  //  that folds them itself -- rsdet_van_fold_bn_f32's gs_tab)
  if ((grad_in_bias || ws) && (!ws || ws_bytes < rsdet_dwconv2d_backward_data_ws_size(N, C, H, W))) return RSDET_EINVAL;
  rc = dw_stencil<true>(grad_y, weight, nullptr, nullptr, N, C, H, W, K, dilation, grad_x, (float*)ws, s);
  if (rc || !grad_in_bias) return rc;
  const int nslots = (int)(dw_slots(N, C, H, W) / C);
  hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3(C), dim3(64), 0, s, (const float*)ws, nslots, 1,
                     (float*)nullptr, grad_in_bias);
  return rsdet_launch_status();
}

// Backward-data whose result meets a second gradient of the same tensor and then a GELU: grad_x = (dwconv^T(grad_y) + add)
// * GELU'(gelu_arg) -- the gradient of t1 in u = GELU(t1), u read by the depthwise pair AND by the gate (SpatialAttention /
// AttentionModule, van.py:177-215); grad_sum[c] = sum over the map of grad_x (the gradient of the bias that produced t1).
extern "C" int rsdet_dwconv2d_backward_data_act_f32(const float* grad_y, const float* weight, int N, int C, int H, int W,
                                                    int K, int dilation, const float* add, const float* gelu_arg,
                                                    float* grad_x, float* grad_sum, void* ws, size_t ws_bytes,
                                                    void* stream) {
  int rc = dw_check(N, C, H, W, K, dilation);
  if (rc) return rc;
  if (C == 0 || N == 0) return RSDET_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!grad_y || !weight || !grad_x || !add || !gelu_arg) return RSDET_EINVAL;
  if ((grad_sum || ws) && (!ws || ws_bytes < rsdet_dwconv2d_backward_data_ws_size(N, C, H, W))) return RSDET_EINVAL;
  rc = dw_stencil<true>(grad_y, weight, nullptr, nullptr, N, C, H, W, K, dilation, grad_x, (float*)ws, s, nullptr, add,
                        gelu_arg);
  if (rc || !grad_sum) return rc;
  const int nslots = (int)(dw_slots(N, C, H, W) / C);
  hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3(C), dim3(64), 0, s, (const float*)ws, nslots, 1, (float*)nullptr,
                     grad_sum);
  return rsdet_launch_status();
}

extern "C" size_t rsdet_dwconv2d_backward_weight_ws_size(int N, int C, int H, int W, int K) {
  if (N <= 0 || C <= 0 || H < 1 || W < 1 || K < 1) return 0;
  return dw_slots(N, C, H, W) * (K * K + 1) * sizeof(float);
}

// the partial rows of the weight gradient: (C, nslots, K*K + 1) floats in ws; -> nslots (< 0: error)
static int dw_wgrad_partials(const float* grad_y, const float* x, const float* in_bias, int N, int C, int H, int W, int K,
                             void* ws, hipStream_t s) {
  const int tx = (W + DW_TW - 1) / DW_TW, ty = (H + DW_TH - 1) / DW_TH, ntiles = tx * ty;
  // tiles per workgroup: as many as still leave ~1 500 workgroups (6 per CU) to the launch
  long long tpw = ((long long)N * C * ntiles) / 1536;
  tpw = tpw < 1 ? 1 : (tpw > ntiles ? ntiles : tpw);
  const int groups = (int)((ntiles + tpw - 1) / tpw);
  const int nslots = K == 7 ? N * ntiles : N * groups;   // (7x7: the one-tile kernel below; <= N * ntiles either way)
  const dim3 grid(N * C, groups);
  float* partial = (float*)ws;
  if (K == 3)
    hipLaunchKernelGGL((dwconv_wgrad_kernel<3, 1>), grid, dim3(DW_NT), 0, s, grad_y, x, in_bias, C, H, W, tx, ntiles,
                       (int)tpw, nslots, partial);
  else if (K == 5)
    hipLaunchKernelGGL((dwconv_wgrad_kernel<5, 1>), grid, dim3(DW_NT), 0, s, grad_y, x, in_bias, C, H, W, tx, ntiles,
                       (int)tpw, nslots, partial);
  else
    hipLaunchKernelGGL((dwconv_wgrad_tile_kernel<7, 3>), dim3(N * C, ntiles), dim3(DW_NT), 0, s, grad_y, x, in_bias, C,
                       H, W, tx, N * ntiles, partial);
  return nslots;
}

// The weight gradient WITHOUT its finishing pass: the partial rows stay in ws (rsdet_dwconv2d_backward_weight_ws_size) for
// rsdet_dwconv2d_wgrad_finish_multi_f32, which sums the partials of up to four layers in one launch.  N >= 1.
extern "C" int rsdet_dwconv2d_backward_weight_partial_f32(const float* grad_y, const float* x, const float* in_bias, int N,
                                                          int C, int H, int W, int K, int dilation, void* ws,
                                                          size_t ws_bytes, void* stream) {
  int rc = dw_check(N, C, H, W, K, dilation);
  if (rc) return rc;
  if (N < 1 || C < 1 || !grad_y || !x || !ws || ws_bytes < rsdet_dwconv2d_backward_weight_ws_size(N, C, H, W, K))
    return RSDET_EINVAL;
  dw_wgrad_partials(grad_y, x, in_bias, N, C, H, W, K, ws, (hipStream_t)stream);
  return rsdet_launch_status();
}

extern "C" int rsdet_dwconv2d_wgrad_finish_multi_f32(int n, const void* const* ws, const int* N, const int* C, const int* H,
                                                     const int* W, const int* K, float* const* grad_weight,
                                                     float* const* grad_bias, void* stream) {
  if (n < 1 || n > 4 || !ws || !N || !C || !H || !W || !K || !grad_weight || !grad_bias) return RSDET_EINVAL;
  DwFinishJobs j;
  int cmax = 0;
  for (int i = 0; i < 4; ++i) {
    const int k = i < n ? i : 0;
    if (i < n && (N[k] < 1 || C[k] < 1 || !ws[k] || !grad_weight[k] || dw_check(N[k], C[k], H[k], W[k], K[k], K[k] == 7 ? 3 : 1)))
      return RSDET_EINVAL;
    const int tx = (W[k] + DW_TW - 1) / DW_TW, ty = (H[k] + DW_TH - 1) / DW_TH, ntiles = tx * ty;
    long long tpw = ((long long)N[k] * C[k] * ntiles) / 1536;
    tpw = tpw < 1 ? 1 : (tpw > ntiles ? ntiles : tpw);
    const int groups = (int)((ntiles + tpw - 1) / tpw);
    j.partial[i] = (const float*)ws[k];
    j.gw[i] = grad_weight[k];
    j.gb[i] = grad_bias[k];
    j.nslots[i] = K[k] == 7 ? N[k] * ntiles : N[k] * groups;
    j.T[i] = K[k] * K[k] + 1;
    j.C[i] = i < n ? C[k] : 0;
    if (i < n && C[k] > cmax) cmax = C[k];
  }
  hipLaunchKernelGGL(dwconv_wgrad_finish_multi_kernel, dim3(cmax, n), dim3(64), 0, (hipStream_t)stream, j);
  return rsdet_launch_status();
}

extern "C" int rsdet_dwconv2d_backward_weight_f32(const float* grad_y, const float* x, const float* in_bias, int N,
                                                  int C, int H, int W, int K, int dilation, float* grad_weight,
                                                  float* grad_bias, void* ws, size_t ws_bytes, void* stream) {
  int rc = dw_check(N, C, H, W, K, dilation);
  if (rc) return rc;
  if (C == 0) return RSDET_OK;
  if (!grad_weight) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    if (hipMemsetAsync(grad_weight, 0, (size_t)C * K * K * 4, s) != hipSuccess) return RSDET_ELAUNCH;
    if (grad_bias && hipMemsetAsync(grad_bias, 0, (size_t)C * 4, s) != hipSuccess) return RSDET_ELAUNCH;
    return RSDET_OK;
  }
  if (!grad_y || !x || !ws || ws_bytes < rsdet_dwconv2d_backward_weight_ws_size(N, C, H, W, K)) return RSDET_EINVAL;
  float* partial = (float*)ws;
  const int nslots = dw_wgrad_partials(grad_y, x, in_bias, N, C, H, W, K, ws, s);
  hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3(C), dim3(64), 0, s, partial, nslots, K * K + 1, grad_weight,
                     grad_bias);
  return rsdet_launch_status();
}
