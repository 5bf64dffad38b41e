// layout.hip -- (B, R, C) -> (B, C, R) transposes of fp32 tensors through an LDS tile: the NCHW <-> NHWC turns around
// the channels-last gather kernels (RROIAlign / FeatureRefine / AlignConv backward, /root/reference/python/jdet/ops/
// roi_align_rotated_v1.py:329-351, fr.py:235-260, dcn_v1.py:456-557 consume and produce NCHW).  torch's generic strided
// copy moves such a turn at ~1.5 TB/s (a 134 MB level: ~180 us); 64 x 64 tiles with 16-byte accesses on both sides run at
// the HBM rate.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

constexpr int TR_T = 64;  // tile edge

// grid: (ceil(C / 64), ceil(R / 64), B); block 256.  in[b][r][c] -> out[b][c][r]
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int R, int C, int aligned16) {
  __shared__ float tile[TR_T][TR_T + 1];
  const int t = threadIdx.x;
  const long long base_in = (long long)blockIdx.z * R * C, base_out = base_in;
  const int c0 = blockIdx.x * TR_T, r0 = blockIdx.y * TR_T;
  // 16-byte accesses need 16-byte aligned bases (a view with a storage offset need not be) and row pitches
  const bool full = aligned16 && c0 + TR_T <= C && r0 + TR_T <= R && (C & 3) == 0 && (R & 3) == 0;
  if (full) {
    const int q = (t & 15) * 4, p = t >> 4;   // 16 threads x float4 = one 256-byte tile row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = p + 16 * i;
      const float4 v = *reinterpret_cast<const float4*>(in + base_in + (long long)(r0 + r) * C + c0 + q);
      tile[q][r] = v.x; tile[q + 1][r] = v.y; tile[q + 2][r] = v.z; tile[q + 3][r] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = p + 16 * i;
      const float4 v = make_float4(tile[c][q], tile[c][q + 1], tile[c][q + 2], tile[c][q + 3]);
      *reinterpret_cast<float4*>(out + base_out + (long long)(c0 + c) * R + r0 + q) = v;
    }
  } else {   // ragged edge tiles: element by element
    for (int e = t; e < TR_T * TR_T; e += 256) {
      const int r = e / TR_T, c = e - r * TR_T;
      if (r0 + r < R && c0 + c < C) tile[c][r] = in[base_in + (long long)(r0 + r) * C + c0 + c];
    }
    __syncthreads();
    for (int e = t; e < TR_T * TR_T; e += 256) {
      const int c = e / TR_T, r = e - c * TR_T;
      if (r0 + r < R && c0 + c < C) out[base_out + (long long)(c0 + c) * R + r0 + r] = tile[c][r];
    }
  }
}


// Channels-last convolution weights (O, T, C) -> (C, T', O) with the T taps reversed (t' = T - 1 - t): the weights of the
// convolution that computes a stride-1 "same" convolution's INPUT gradient as a forward convolution of the output
// gradient (ops/conv3x3.py).  One 32 x 32 (o, c) tile per workgroup and tap through LDS; elements are copied as bits.
template <typename E>
__global__ __launch_bounds__(256) void weight_flip_transpose_kernel(const E* __restrict__ in, E* __restrict__ out, int O,
                                                                    int C, int T) {
  __shared__ E tile[32][33];
  const int t = blockIdx.z, o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o0 + ty + 8 * i, c = c0 + tx;
    if (o < O && c < C) tile[ty + 8 * i][tx] = in[((long long)o * T + t) * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, o = o0 + tx;
    if (o < O && c < C) out[((long long)c * T + (T - 1 - t)) * O + o] = tile[tx][ty + 8 * i];
  }
}

// ---- every prepared weight operand of a step in ONE launch --------------------------------------------------------------
// The backward of a bf16 step wants, per convolution weight (O, T, C) (channels_last (O, C, kh, kw), T = kh kw taps), the
// operand (C, T, O) with the taps reversed -- the flipped weights of "backward-data through the forward solver"
// (ops/conv3x3.py; T = 9) and the transposed weights of the 1x1 backward-data GEMM (ops/bottleneck.py; T = 1), the latter
// optionally with column o scaled by gamma[o] / sqrt(var[o] + eps) of the BatchNorm behind the convolution.  The weights
// change once per optimizer step, so all of them are prepared by one launch after it (ops/weight_prep.py) instead of
// one small launch per use: 37 launches -> 1 in the S2ANet-R50 step.  32 x 32 (o, c) tiles through LDS.
struct WpEntry {         // 64 bytes, packed by the host (ops/weight_prep.py)
  const bf16_t* src;
  bf16_t* dst;
  const float* var;      // NULL: no scale
  const float* gamma;    // NULL: 1
  int O, C, T;
  float eps;
  int tile0;             // first tile (workgroup) of this entry
  int tiles_c;           // ceil(C / 32)
  int tiles_o;           // ceil(O / 32)
  int pad;
};
static_assert(sizeof(WpEntry) == 64, "host packs 64-byte records");

__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const WpEntry* __restrict__ entries, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;                           // the last entry whose tile0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (entries[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WpEntry e = entries[lo];
  int l = (int)blockIdx.x - e.tile0;
  const int per_tap = e.tiles_c * e.tiles_o;
  const int t = l / per_tap;
  l -= t * per_tap;
  const int o0 = (l / e.tiles_c) * 32, c0 = (l % e.tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (o < e.O && c < e.C) {
      v = bf2f(e.src[((long long)o * e.T + t) * e.C + c]);
      if (e.var) {
        float sc = 1.0f / sqrtf(e.var[o] + e.eps);
        if (e.gamma) sc *= e.gamma[o];
        v *= sc;
      }
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, o = o0 + tx;
    if (o < e.O && c < e.C) e.dst[((long long)c * e.T + (e.T - 1 - t)) * e.O + o] = f2bf(tile[tx][ty + 8 * i]);
  }
}

}  // namespace rsdet

using namespace rsdet;

// entries: n 64-byte WpEntry records in DEVICE memory (layout above; tile0 ascending from 0, entry j owning T * tiles_c *
// tiles_o tiles), total_tiles their sum.  An unscaled entry copies bit patterns (bf16 -> fp32 -> bf16 is exact).
extern "C" int rsdet_weight_prep_multi_bf16(const void* entries, int n, int total_tiles, void* stream) {
  if (n < 0 || total_tiles < 0) return RSDET_EINVAL;
  if (n == 0 || total_tiles == 0) return RSDET_OK;
  if (!entries) return RSDET_EINVAL;
  hipLaunchKernelGGL(weight_prep_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const WpEntry*)entries, n);
  return rsdet_launch_status();
}

extern "C" int rsdet_transpose_last2_f32(const float* in, float* out, int B, int R, int C, void* stream) {
  if (B < 0 || R < 0 || C < 0) return RSDET_EINVAL;
  if (B == 0 || R == 0 || C == 0) return RSDET_OK;
  if (!in || !out || in == out) return RSDET_EINVAL;
  const long long gx = (C + TR_T - 1) / TR_T, gy = (R + TR_T - 1) / TR_T;
  if (gy > 65535 || B > 65535) return RSDET_EINVAL;
  hipLaunchKernelGGL(transpose_last2_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, in, out, R, C,
                     (((uintptr_t)in | (uintptr_t)out) & 15) == 0 ? 1 : 0);
  return rsdet_launch_status();
}

// in: (O, T, C) elements of elem_bytes (2 or 4) -- a channels_last (O, C, kh, kw) weight with T = kh * kw;
// out: (C, T, O) with the taps reversed -- the channels_last (C, O, kh, kw) weight flipped in both spatial directions.
extern "C" int rsdet_weight_flip_transpose(const void* in, void* out, int O, int C, int T, int elem_bytes, void* stream) {
  if (O < 0 || C < 0 || T < 0 || (elem_bytes != 2 && elem_bytes != 4)) return RSDET_EINVAL;
  if (O == 0 || C == 0 || T == 0) return RSDET_OK;
  if (!in || !out || in == out || T > 65535) return RSDET_EINVAL;
  const dim3 grid((C + 31) / 32, (O + 31) / 32, T);
  if (grid.y > 65535u) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (elem_bytes == 4)
    hipLaunchKernelGGL(weight_flip_transpose_kernel<uint32_t>, grid, dim3(256), 0, s, (const uint32_t*)in, (uint32_t*)out,
                       O, C, T);
  else
    hipLaunchKernelGGL(weight_flip_transpose_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)in, (uint16_t*)out,
                       O, C, T);
  return rsdet_launch_status();
}
