// gemm1x1_mfma.hip -- the 1x1 convolutions of the bf16 ResNet trunk as ONE launch each, with the frozen-statistics
// BatchNorm, the residual add and the ReLU in the GEMM's epilogue (gfx950 matrix cores).
//
// Replaces, for a channels-last bf16 map (which IS the row-major (positions, channels) matrix):
//   conv1 / conv3 / downsample[0] + bn + (identity add) + relu of the Bottleneck,
//   /root/reference/python/jdet/models/backbones/resnet.py:57-93 (norm_eval: BatchNorm in eval mode, :177-184, so
//   bn(x) = x * gamma / sqrt(var + eps) + (beta - mean * gamma / sqrt(var + eps)) is a per-channel affine map),
// which the step ran as a library GEMM (ops/conv1x1.py) followed by a `bn_act` pass over the result (csrc/bn_act.hip:
// read conv output + residual, write y: 0.86 ms and 48 launches of the 15.9 ms bf16 step).  Here the conv output never
// reaches HBM: out[p, o] = relu((sum_c x[p, c] W[o, c]) * s[o] + t[o] + res[p, o]).
//
//   GEMM   M = positions (B*H*W), N = output channels, K = input channels; both operands K-contiguous as they lie.
//   tile   one workgroup = 224 positions x 256 output channels (the tile of csrc/conv3x3_mfma.hip: 2 x 4 waves of
//          112 x 64, v_mfma_f32_16x16x32_bf16, 28 accumulators per wave); K in steps of 64 channels;
//   LDS    A tile 224 x 128 B and B tile 256 x 128 B per step, both by LDS-DMA (global_load_lds, 16 B per lane) into a
//          two-slot ring, 16-byte chunk c of row r in slot c ^ (r & 7) (conflict-free ds_read_b128 fragments; the
//          swizzle is applied on the DMA's SOURCE address); one barrier per step: wait for tile s -> barrier -> issue
//          tile s + 1 into the slot step s - 1 read -> the 56 MFMAs of step s;
//   frags  hand-issued ds_read_b128 a sub-step ahead with counted lgkmcnt (the conv3x3 kernel's sub-step macro);
//   D = W-fragment x X-fragment: a lane ends up with four consecutive output channels of one position: 8-byte
//          stores, per-channel scale / shift in registers.
// The arithmetic is the convolution's and the BatchNorm's own: bf16 products, fp32 accumulation, the affine map and the
// residual in fp32, ONE rounding to bf16 (the two-launch form rounded the conv output to bf16 first).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

typedef __attribute__((ext_vector_type(8))) __bf16 g1_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned g1_u32x4;
typedef __attribute__((ext_vector_type(4))) float g1_f32x4;

constexpr int G1_TM = 224, G1_TN = 256, G1_NW = 8;
constexpr int G1_A_BYTES = G1_TM * 128, G1_B_BYTES = G1_TN * 128;           // 28 672, 32 768
constexpr int G1_SLOT = G1_A_BYTES + G1_B_BYTES;                             // 61 440
constexpr int G1_LDS_BYTES = 2 * G1_SLOT;                                    // 122 880
constexpr int G1_A_PIECES = G1_A_BYTES / 1024, G1_B_PIECES = G1_B_BYTES / 1024;   // 28, 32 (1 KiB = 8 rows x 128 B)
constexpr int G1_A_ITERS = (G1_A_PIECES + G1_NW - 1) / G1_NW;                // 4 (waves 4..7 issue 3)
constexpr int G1_B_OPS = G1_B_PIECES / G1_NW;                                // 4
constexpr int G1_MI = 7, G1_NI = 4, G1_WM = 112;

__device__ const uint4 g1_zero_line[8] = {};   // 128 B of zeros: rows past the end of the matrix

struct G1Geom {
  long long M;
  int N, K;
};

// epilogue parameters (all per output channel n unless said otherwise; null = absent)
struct G1Epi {
  const float* mean;     // BatchNorm running mean / var / weight / bias: scale = gamma / sqrt(var + eps),
  const float* var;      //   shift = beta - mean * scale; mean == var == null: plain convolution (scale 1)
  const float* gamma;
  const float* beta;     // with mean == null: a plain bias
  float eps;
  const bf16_t* res;     // [M][N] residual added before the activation, or null
  int relu;
};

template <int N>
__device__ __forceinline__ void g1_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void g1_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void g1_lds_read(g1_u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void g1_landed(g1_u32x4& v) { asm volatile("" : "+v"(v)); }

// grid: rsdet_xcd_band_grid(m_tiles, n_tiles); block 512.
__global__ __launch_bounds__(64 * G1_NW, 1) void gemm1x1_bn_act_mfma_bf16_kernel(
    const bf16_t* __restrict__ a, const bf16_t* __restrict__ w, G1Geom g, G1Epi e, int m_tiles, int n_tiles,
    bf16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G1_LDS_BYTES];
  const RsdetBandItem item = rsdet_xcd_band(blockIdx.x, m_tiles, n_tiles);
  if (!item.valid) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long p0 = (long long)item.outer * G1_TM;
  const int n_base = item.inner * G1_TN;
  const int steps = g.K >> 6;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g1_zero_line);
  const int prow = lane >> 3;                       // row of the lane inside a piece = (LDS row) & 7
  const int chunk = (lane & 7) ^ prow;              // the source chunk that belongs in the lane's slot
  // per-lane source offsets (elements) of the A and B pieces this wave moves; -1: a row past the matrix
  long long a_off[G1_A_ITERS], b_off[G1_B_OPS];
#pragma unroll
  for (int it = 0; it < G1_A_ITERS; ++it) {
    const long long p = p0 + (wave + it * G1_NW) * 8 + prow;
    a_off[it] = p < g.M ? p * g.K + chunk * 8 : -1;
  }
#pragma unroll
  for (int it = 0; it < G1_B_OPS; ++it) {
    const int r = (wave + it * G1_NW) * 8 + prow;
    b_off[it] = (long long)min(n_base + r, g.N - 1) * g.K + chunk * 8;
  }
  auto issue = [&](int s) {
    unsigned char* slot = lds + (s & 1) * G1_SLOT;
    const int k0 = s * 64;
#pragma unroll
    for (int it = 0; it < G1_A_ITERS; ++it) {
      const int piece = wave + it * G1_NW;
      if (piece < G1_A_PIECES) {
        const bf16_t* src = a_off[it] >= 0 ? a + a_off[it] + k0 : zero + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int it = 0; it < G1_B_OPS; ++it) {
      const int piece = wave + it * G1_NW;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + b_off[it] + k0),
                                       (__attribute__((address_space(3))) void*)(slot + G1_A_BYTES + piece * 1024), 16,
                                       0, 0);
    }
  };

  g1_f32x4 acc[G1_MI][G1_NI];
#pragma unroll
  for (int mi = 0; mi < G1_MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < G1_NI; ++ni)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[mi][ni][k] = 0.f;

  const int wm = wave >> 2, wn = wave & 3;
  const int q4 = lane >> 4, l15 = lane & 15;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // fragment addresses: position row = wm * 112 + mi * 16 + (lane & 15), weight row = wn * 64 + ni * 16 + (lane & 15);
  // 16-byte slot (4 ks + (lane >> 4)) ^ (row & 7): wm * 112, mi * 16, ni * 16 are 0 mod 8 and 4 ks is bit 2 of the slot:
  // one base per lane, XOR 64 for the second k-32 sub-step, immediates mi * 2048 / ni * 2048
  const unsigned a_lane = (unsigned)((wm * G1_WM + l15) * 128 + ((q4 ^ (l15 & 7)) << 4));
  const unsigned b_lane = (unsigned)(G1_A_BYTES + (wn * 64 + l15) * 128 + ((q4 ^ (l15 & 7)) << 4));
  g1_u32x4 fa[2][G1_MI], fb[2][G1_NI];
#define G1_RA(buf, mi, ab) g1_lds_read<(mi) * 2048>(fa[buf][mi], ab)
#define G1_RB(buf, ni, bb) g1_lds_read<(ni) * 2048>(fb[buf][ni], bb)
  // read order of a sub-step: B0, A0 .. A6, B1, B2, B3 -- the order the MFMAs (weight-fragment major) first need them
#define G1_READ(buf, idx, ab, bb)                  \
  switch (idx) {                                   \
    case 0: G1_RB(buf, 0, bb); break;              \
    case 1: G1_RA(buf, 0, ab); break;              \
    case 2: G1_RA(buf, 1, ab); break;              \
    case 3: G1_RA(buf, 2, ab); break;              \
    case 4: G1_RA(buf, 3, ab); break;              \
    case 5: G1_RA(buf, 4, ab); break;              \
    case 6: G1_RA(buf, 5, ab); break;              \
    case 7: G1_RA(buf, 6, ab); break;              \
    case 8: G1_RB(buf, 1, bb); break;              \
    case 9: G1_RB(buf, 2, bb); break;              \
    default: G1_RB(buf, 3, bb); break;             \
  }
  static_assert(G1_MI == 7 && G1_NI == 4, "read order and wait counts below are written out for 7 x 4 fragments");
  // One sub-step (k = 32): 28 MFMAs on register set CUR, whose 11 reads were all ISSUED during the previous sub-step;
  // the 11 reads of the next sub-step go out one per MFMA into set 1 - CUR.  LDS reads return in issue order: at MFMA j
  // the first 2 + j (j < 7), 9 / 10 / 11 (from j = 7 / 14 / 21) of the CURRENT set are needed.  `more` false (second
  // sub-step of a step: the next fragments lie behind the barrier): the counts run down.
#define G1_SUBSTEP(CUR, more, ab, bb)                                                                                   \
  {                                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < G1_MI * G1_NI; ++j) {                                                         \
      const int ni = j / G1_MI, mi = j - ni * G1_MI;                                                                    \
      if (more) {                                                                                                       \
        if (j < 11) G1_READ(1 - CUR, j, ab, bb);                                                                        \
        if (j <= 7) g1_wait_lgkm<10>();                                                                                 \
        else if (j == 14) g1_wait_lgkm<12>();                                                                           \
        else if (j == 21) g1_wait_lgkm<11>();                                                                           \
      } else {                                                                                                          \
        if (j == 0) g1_wait_lgkm<9>();                                                                                  \
        else if (j == 1) g1_wait_lgkm<8>();                                                                             \
        else if (j == 2) g1_wait_lgkm<7>();                                                                             \
        else if (j == 3) g1_wait_lgkm<6>();                                                                             \
        else if (j == 4) g1_wait_lgkm<5>();                                                                             \
        else if (j == 5) g1_wait_lgkm<4>();                                                                             \
        else if (j == 6) g1_wait_lgkm<3>();                                                                             \
        else if (j == 7) g1_wait_lgkm<2>();                                                                             \
        else if (j == 14) g1_wait_lgkm<1>();                                                                            \
        else if (j == 21) g1_wait_lgkm<0>();                                                                            \
      }                                                                                                                 \
      if (ni == 0) g1_landed(fa[CUR][mi]);                                                                              \
      if (mi == 0) g1_landed(fb[CUR][ni]);                                                                              \
      /* D = W-fragment x X-fragment: rows = output channels, columns = positions (4 consecutive channels per lane) */ \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g1_bf16x8, fb[CUR][ni]),                 \
                                                            __builtin_bit_cast(g1_bf16x8, fa[CUR][mi]), acc[mi][ni], 0, \
                                                            0, 0);                                                      \
    }                                                                                                                   \
  }

  // ---- schedule: tile s was issued during step s - 1 (tile 0 before the loop) and nothing younger is in flight when
  // step s starts, so `vmcnt(0)` is exactly "tile s has landed" for this wave's pieces; the barrier extends that to all
  // eight waves AND says everybody has finished reading the other slot (step s - 1), which tile s + 1 may now overwrite
  issue(0);
  for (int s = 0; s < steps; ++s) {
    g1_wait_vm<0>();
    __syncthreads();
    if (s + 1 < steps) issue(s + 1);
    const unsigned ab = lds_base + (s & 1) * G1_SLOT + a_lane, bb = lds_base + (s & 1) * G1_SLOT + b_lane;
#pragma unroll
    for (int idx = 0; idx < 11; ++idx) G1_READ(0, idx, ab, bb);
    G1_SUBSTEP(0, true, ab ^ 64u, bb ^ 64u);
    G1_SUBSTEP(1, false, ab, bb);
  }

  // ---- epilogue: lane holds channels n_base + wn * 64 + ni * 16 + 4 (lane >> 4) + 0..3 of position
  // p0 + wm * 112 + mi * 16 + (lane & 15)
  const int ob = n_base + wn * 64 + 4 * q4;
  float sc[G1_NI][4], sh[G1_NI][4];
#pragma unroll
  for (int ni = 0; ni < G1_NI; ++ni)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o = min(ob + 16 * ni + k, g.N - 1);
      float s1 = 1.f, t1 = 0.f;
      if (e.mean) {
        const float is = 1.0f / sqrtf(e.var[o] + e.eps);
        s1 = e.gamma ? is * e.gamma[o] : is;
        t1 = (e.beta ? e.beta[o] : 0.f) - e.mean[o] * s1;
      } else if (e.beta) {
        t1 = e.beta[o];
      }
      sc[ni][k] = s1, sh[ni][k] = t1;
    }
#pragma unroll
  for (int mi = 0; mi < G1_MI; ++mi) {
    const long long p = p0 + wm * G1_WM + mi * 16 + l15;
    if (p >= g.M) continue;
    bf16_t* orow = out + p * g.N;
    const bf16_t* rrow = e.res ? e.res + p * g.N : nullptr;
#pragma unroll
    for (int ni = 0; ni < G1_NI; ++ni) {
      const int o = ob + 16 * ni;
      if (o >= g.N) continue;               // (N % 32 == 0 and quads start at multiples of 4: inside or outside as a whole)
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rrow) r = ld4(rrow + o);
      float v[4];
      v[0] = acc[mi][ni][0] * sc[ni][0] + sh[ni][0] + r.x;
      v[1] = acc[mi][ni][1] * sc[ni][1] + sh[ni][1] + r.y;
      v[2] = acc[mi][ni][2] * sc[ni][2] + sh[ni][2] + r.z;
      v[3] = acc[mi][ni][3] * sc[ni][3] + sh[ni][3] + r.w;
      if (e.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      st4(orow + o, make_float4(v[0], v[1], v[2], v[3]));
    }
  }
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_gemm1x1_mfma_supported(long long M, int N, int K) {
  if (M < 1 || N < 32 || (N & 31) || K < 64 || (K & 63)) return 0;
  if (M * (long long)(N > K ? N : K) >= (1ll << 40)) return 0;
  return 1;
}

// out[p, o] = act((sum_c x[p, c] weight[o, c]) * scale[o] + shift[o] + residual[p, o]),  x (M, K) / weight (N, K) /
// residual, out (M, N) bf16 row-major; scale / shift from the BatchNorm's running statistics and affine parameters
// (fp32, any of them NULL: see G1Epi); relu != 0: max(., 0).
extern "C" int rsdet_conv1x1_bn_act_fwd_bf16(const uint16_t* x, const uint16_t* weight, long long M, int N, int K,
                                             const float* running_mean, const float* running_var, const float* gamma,
                                             const float* beta, float eps, const uint16_t* residual, int relu,
                                             uint16_t* out, void* stream) {
  if (!rsdet_gemm1x1_mfma_supported(M, N, K)) return RSDET_EINVAL;
  if (!x || !weight || !out || ((running_mean == nullptr) != (running_var == nullptr))) return RSDET_EINVAL;
  G1Geom g{M, N, K};
  G1Epi e{running_mean, running_var, gamma, beta, eps, (const bf16_t*)residual, relu};
  const int m_tiles = (int)((M + G1_TM - 1) / G1_TM), n_tiles = (N + G1_TN - 1) / G1_TN;
  const dim3 grid((unsigned)rsdet_xcd_band_grid(m_tiles, n_tiles));
  hipLaunchKernelGGL(gemm1x1_bn_act_mfma_bf16_kernel, grid, dim3(64 * G1_NW), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)weight, g, e, m_tiles, n_tiles, (bf16_t*)out);
  return rsdet_launch_status();
}
