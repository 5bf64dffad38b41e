// losses.hip -- the S2ANet head's classification + regression losses of one module (FAM or ODM) in ONE pass each way.
//
// Replaces, for all pyramid levels of a batch at once:
//   FocalLoss / sigmoid_focal_loss / binary_cross_entropy_with_logits
//       /root/reference/python/jdet/models/losses/focal_loss.py:5-96
//   SmoothL1Loss / smooth_l1_loss      /root/reference/python/jdet/models/losses/smooth_l1_loss.py:5-54
//   the per-level reshapes of loss_fam_single / loss_odm_single
//       /root/reference/python/jdet/models/roi_heads/s2anet_head.py:430-508
// The reference (and round 1) ran 2 x 5 loss calls per module, each a chain of ~10 elementwise kernels over
// permuted copies of the prediction maps: ~200 launches and 5.6 ms of torch elementwise time per step.
//
// Here the prediction maps are read where the convolutions left them -- NCHW, level by level, no permute, no
// concatenation -- one thread per (image, anchor) walking the C class planes (lanes = consecutive positions, so
// every plane read is coalesced), fp32 arithmetic whatever the storage type (fp32 or bf16 maps under autocast).
// Reductions are deterministic: a workgroup covers one (level, image) chunk, folds with shuffles, hands its two
// partial sums over as device-scope atomics (no fence: losses.hip never needs one, see anchor_target.hip), and the
// last workgroup to arrive adds the partials of every level IN INDEX ORDER.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

constexpr int LOSS_NT = 256;
constexpr int LOSS_MAX_LEVELS = 8;

struct LossMaps {
  const void* cls[LOSS_MAX_LEVELS];   // (B, C, H, W) per level
  const void* box[LOSS_MAX_LEVELS];   // (B, 5, H, W) per level
  void* gcls[LOSS_MAX_LEVELS];        // backward outputs, same layouts
  void* gbox[LOSS_MAX_LEVELS];
  int hw[LOSS_MAX_LEVELS];            // H * W
  int a0[LOSS_MAX_LEVELS];            // first anchor of the level inside (B, A)
  int blk0[LOSS_MAX_LEVELS + 1];      // first workgroup of the level: B * ceil(hw / LOSS_NT) workgroups each
  int L, B, C, A;
};

template <typename T>
__device__ __forceinline__ float ld(const void* p, long long i) {
  return ld1(reinterpret_cast<const T*>(p) + i);
}
template <typename T>
__device__ __forceinline__ void st(void* p, long long i, float v) {
  st1(reinterpret_cast<T*>(p) + i, v);
}

struct Where {
  int level, b, hw;   // hw < 0: past the end of the chunk
};
__device__ __forceinline__ Where locate_block(const LossMaps& m) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < LOSS_MAX_LEVELS; ++k)
    if (k < m.L && (int)blockIdx.x >= m.blk0[k]) l = k;
  const int per = (m.hw[l] + LOSS_NT - 1) / LOSS_NT;
  const int r = blockIdx.x - m.blk0[l];
  Where w;
  w.level = l;
  w.b = r / per;
  const int hw = (r - w.b * per) * LOSS_NT + threadIdx.x;
  w.hw = hw < m.hw[l] ? hw : -1;
  return w;
}

// focal term of one logit (focal_loss.py:5-21 BCE in the max_val form; :37-52 the modulating factors)
__device__ __forceinline__ float focal_term(float x, bool t, float alpha, float gamma, bool gamma2) {
  const float p = 1.0f / (1.0f + __expf(-x));
  const float max_val = fmaxf(-x, 0.0f);
  const float ce = (t ? 0.0f : x) + max_val + __logf(fmaxf(__expf(-max_val) + __expf(-x - max_val), 1e-10f));
  const float q = t ? 1.0f - p : p;  // 1 - p_t
  const float mod = gamma2 ? q * q : __powf(q, gamma);
  const float at = alpha >= 0.f ? (t ? alpha : 1.0f - alpha) : 1.0f;
  return at * ce * mod;
}
// d focal_term / d x
__device__ __forceinline__ float focal_grad(float x, bool t, float alpha, float gamma, bool gamma2) {
  const float p = 1.0f / (1.0f + __expf(-x));
  const float at = alpha >= 0.f ? (t ? alpha : 1.0f - alpha) : 1.0f;
  // log p and log(1 - p) in the stable softplus form
  const float log_p = -(fmaxf(-x, 0.f) + __logf(1.f + __expf(-fabsf(x))));
  const float log_1mp = -(fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x))));
  if (t) {
    const float q = 1.f - p;
    const float mod = gamma2 ? q * q : __powf(q, gamma);
    return at * mod * (gamma * p * log_p - q);
  }
  const float mod = gamma2 ? p * p : __powf(p, gamma);
  return at * mod * (p - gamma * (1.f - p) * log_1mp);
}

__device__ __forceinline__ float block_sum(float v, float* s_part) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_part[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < LOSS_NT / 64; ++w) r += s_part[w];   // fixed order
  __syncthreads();
  return r;  // valid in thread 0
}

template <typename T>
__global__ __launch_bounds__(LOSS_NT) void s2a_loss_fwd_kernel(
    const LossMaps m, const int* __restrict__ labels, const float* __restrict__ label_w,
    const float* __restrict__ box_t, const float* __restrict__ box_w, const float* __restrict__ avg_factor,
    float alpha, float gamma, float beta, float w_cls, float w_box, float* __restrict__ partial /* 2 x nblocks */,
    unsigned* __restrict__ counter, float* __restrict__ losses /* 2 x L */) {
  __shared__ float s_part[LOSS_NT / 64];
  __shared__ int s_last;
  const Where w = locate_block(m);
  float cls = 0.f, box = 0.f;
  if (w.hw >= 0) {
    const long long o = (long long)w.b * m.A + m.a0[w.level] + w.hw;
    const int HW = m.hw[w.level];
    const float lw = label_w[o];
    if (lw != 0.f) {
      const int lab = labels[o];
      const long long base = (long long)w.b * m.C * HW + w.hw;
      const bool g2 = gamma == 2.0f;
      for (int c = 0; c < m.C; ++c)
        cls += focal_term(ld<T>(m.cls[w.level], base + (long long)c * HW), lab == c + 1, alpha, gamma, g2);
      cls *= lw;
    }
    const long long bbase = (long long)w.b * 5 * HW + w.hw;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float bw = box_w[o * 5 + k];
      if (bw != 0.f) {
        const float d = fabsf(ld<T>(m.box[w.level], bbase + (long long)k * HW) - box_t[o * 5 + k]);
        box += bw * (beta != 0.f ? (d < beta ? 0.5f * d * d / beta : d - 0.5f * beta) : d);
      }
    }
  }
  const float sc = block_sum(cls, s_part);
  const float sb = block_sum(box, s_part);
  if (threadIdx.x == 0) {
    // Hand-off in the form MI355X_MICROARCH.md lists as valid for cross-CU data ("agent atomics both sides"): the
    // partials go out as device-scope atomic stores, this wave WAITS for them (s_waitcnt vmcnt(0); the "memory" clobber
    // also pins the compiler's order), only then the arrival is counted; the last arriver reads the partials with
    // returning atomics.  (An acq_rel arrival costs a buffer_wbl2 + buffer_inv per workgroup: at_finish, which uses the
    // same pattern, went from 11.8 to 20.7 us with it.)
    atomicExch(reinterpret_cast<unsigned*>(partial) + 2 * blockIdx.x, __float_as_uint(sc));
    atomicExch(reinterpret_cast<unsigned*>(partial) + 2 * blockIdx.x + 1, __float_as_uint(sb));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(counter, 1u) == gridDim.x - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- last workgroup: one wave per (level, loss): lanes read the level's partials in a fixed stride (returning
  // atomics: read where the atomic stores were performed, never from a cache line a previous call left behind), fold
  // by shuffles in a fixed pattern -> the same bits every run; / avg_factor, x loss weight
  const int lane = threadIdx.x & 63;
  for (int task = threadIdx.x >> 6; task < 2 * m.L; task += LOSS_NT / 64) {
    const int l = task >> 1, which = task & 1;
    float acc = 0.f;
    for (int b = m.blk0[l] + lane; b < m.blk0[l + 1]; b += 64)
      acc += __uint_as_float(atomicOr(reinterpret_cast<unsigned*>(partial) + 2 * b + which, 0u));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) losses[which * m.L + l] = acc / avg_factor[0] * (which ? w_box : w_cls);
  }
  if (threadIdx.x == 0) atomicExch(counter, 0u);  // ready for the next call
}

template <typename T>
__global__ __launch_bounds__(LOSS_NT) void s2a_loss_bwd_kernel(
    const LossMaps m, const int* __restrict__ labels, const float* __restrict__ label_w,
    const float* __restrict__ box_t, const float* __restrict__ box_w, const float* __restrict__ avg_factor,
    const float* __restrict__ grad_losses /* 2 x L */, float alpha, float gamma, float beta, float w_cls,
    float w_box) {
  const Where w = locate_block(m);
  if (w.hw < 0) return;
  const long long o = (long long)w.b * m.A + m.a0[w.level] + w.hw;
  const int HW = m.hw[w.level];
  const float inv = 1.0f / avg_factor[0];
  const float gc = grad_losses[w.level] * w_cls * inv, gb = grad_losses[m.L + w.level] * w_box * inv;
  const float lw = label_w[o];
  const int lab = lw != 0.f ? labels[o] : 0;
  const long long base = (long long)w.b * m.C * HW + w.hw;
  const bool g2 = gamma == 2.0f;
  for (int c = 0; c < m.C; ++c) {
    const long long i = base + (long long)c * HW;
    float g = 0.f;
    if (lw != 0.f) g = gc * lw * focal_grad(ld<T>(m.cls[w.level], i), lab == c + 1, alpha, gamma, g2);
    st<T>(m.gcls[w.level], i, g);
  }
  const long long bbase = (long long)w.b * 5 * HW + w.hw;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const long long i = bbase + (long long)k * HW;
    const float bw = box_w[o * 5 + k];
    float g = 0.f;
    if (bw != 0.f) {
      const float d = ld<T>(m.box[w.level], i) - box_t[o * 5 + k];
      const float ad = fabsf(d);
      const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      g = gb * bw * (beta != 0.f ? (ad < beta ? d / beta : s) : s);
    }
    st<T>(m.gbox[w.level], i, g);
  }
}

}  // namespace rsdet

using namespace rsdet;

static int fill_maps(LossMaps& m, const void* const* cls, const void* const* box, void* const* gcls, void* const* gbox,
                     const int* hw, int L, int B, int C) {
  if (L < 1 || L > LOSS_MAX_LEVELS || B < 1 || C < 1) return RSDET_EINVAL;
  m.L = L, m.B = B, m.C = C;
  int a = 0, blk = 0;
  for (int l = 0; l < L; ++l) {
    if (hw[l] <= 0 || !cls[l] || !box[l]) return RSDET_EINVAL;
    m.cls[l] = cls[l], m.box[l] = box[l];
    m.gcls[l] = gcls ? gcls[l] : nullptr, m.gbox[l] = gbox ? gbox[l] : nullptr;
    m.hw[l] = hw[l], m.a0[l] = a, m.blk0[l] = blk;
    a += hw[l];
    blk += B * ((hw[l] + LOSS_NT - 1) / LOSS_NT);
  }
  for (int l = L; l < LOSS_MAX_LEVELS; ++l) m.hw[l] = 1, m.a0[l] = a, m.blk0[l] = blk;
  m.blk0[L] = blk;
  m.A = a;
  return RSDET_OK;
}

extern "C" size_t rsdet_s2a_loss_ws_size(const int* hw_host, int n_levels, int B) {
  if (!hw_host || n_levels < 1 || n_levels > LOSS_MAX_LEVELS || B < 1) return 0;
  size_t blk = 0;
  for (int l = 0; l < n_levels; ++l) blk += (size_t)B * ((hw_host[l] + LOSS_NT - 1) / LOSS_NT);
  return 256 + blk * 8;  // [0] arrival counter (zero on entry, zero again on exit) | 2 partial sums per workgroup
}

extern "C" int rsdet_s2a_loss_forward(const void* const* cls_maps, const void* const* box_maps, int bf16_maps,
                                      const int* hw_host, int n_levels, int B, int C, const int* labels,
                                      const float* label_weights, const float* bbox_targets,
                                      const float* bbox_weights, const float* avg_factor, float alpha, float gamma,
                                      float beta, float w_cls, float w_box, float* losses, void* ws, size_t ws_bytes,
                                      void* stream) {
  if (!cls_maps || !box_maps || !hw_host || !labels || !label_weights || !bbox_targets || !bbox_weights ||
      !avg_factor || !losses || !ws || ((uintptr_t)ws & 15))
    return RSDET_EINVAL;
  LossMaps m{};
  const int rc = fill_maps(m, cls_maps, box_maps, nullptr, nullptr, hw_host, n_levels, B, C);
  if (rc != RSDET_OK) return rc;
  if (ws_bytes < rsdet_s2a_loss_ws_size(hw_host, n_levels, B)) return RSDET_EINVAL;
  unsigned* counter = (unsigned*)ws;
  float* partial = (float*)((char*)ws + 256);
  const dim3 grid((unsigned)m.blk0[n_levels]);
  if (bf16_maps)
    hipLaunchKernelGGL(s2a_loss_fwd_kernel<bf16_t>, grid, dim3(LOSS_NT), 0, (hipStream_t)stream, m, labels,
                       label_weights, bbox_targets, bbox_weights, avg_factor, alpha, gamma, beta, w_cls, w_box, partial,
                       counter, losses);
  else
    hipLaunchKernelGGL(s2a_loss_fwd_kernel<float>, grid, dim3(LOSS_NT), 0, (hipStream_t)stream, m, labels,
                       label_weights, bbox_targets, bbox_weights, avg_factor, alpha, gamma, beta, w_cls, w_box, partial,
                       counter, losses);
  return rsdet_launch_status();
}

extern "C" int rsdet_s2a_loss_backward(const void* const* cls_maps, const void* const* box_maps, int bf16_maps,
                                       const int* hw_host, int n_levels, int B, int C, const int* labels,
                                       const float* label_weights, const float* bbox_targets,
                                       const float* bbox_weights, const float* avg_factor, const float* grad_losses,
                                       float alpha, float gamma, float beta, float w_cls, float w_box,
                                       void* const* grad_cls, void* const* grad_box, void* stream) {
  if (!cls_maps || !box_maps || !hw_host || !labels || !label_weights || !bbox_targets || !bbox_weights ||
      !avg_factor || !grad_losses || !grad_cls || !grad_box)
    return RSDET_EINVAL;
  LossMaps m{};
  const int rc = fill_maps(m, cls_maps, box_maps, grad_cls, grad_box, hw_host, n_levels, B, C);
  if (rc != RSDET_OK) return rc;
  for (int l = 0; l < n_levels; ++l)
    if (!grad_cls[l] || !grad_box[l]) return RSDET_EINVAL;
  const dim3 grid((unsigned)m.blk0[n_levels]);
  if (bf16_maps)
    hipLaunchKernelGGL(s2a_loss_bwd_kernel<bf16_t>, grid, dim3(LOSS_NT), 0, (hipStream_t)stream, m, labels,
                       label_weights, bbox_targets, bbox_weights, avg_factor, grad_losses, alpha, gamma, beta, w_cls,
                       w_box);
  else
    hipLaunchKernelGGL(s2a_loss_bwd_kernel<float>, grid, dim3(LOSS_NT), 0, (hipStream_t)stream, m, labels,
                       label_weights, bbox_targets, bbox_weights, avg_factor, grad_losses, alpha, gamma, beta, w_cls,
                       w_box);
  return rsdet_launch_status();
}
