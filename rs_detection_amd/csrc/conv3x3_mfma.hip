// conv3x3_mfma.hip -- 3x3 / stride 1 / padding 1 convolution of a channels-last bf16 map as an IMPLICIT GEMM on the
// matrix cores of gfx950, written for the shapes of the S2ANet head: the shared-weight 256 -> 256 tower convolutions
// on the pyramid canvas (4 x 128 x 196 x 256: 118 GFLOP per call, 17 calls per bf16 step forward + backward-data).
//
// Replaces (as the library kernel behind them, same arithmetic: bf16 products, fp32 accumulation, one rounding):
//   the `ConvModule` convolutions of /root/reference/python/jdet/models/roi_heads/s2anet_head.py:127-186 (fam / odm
//   reg / cls towers, or_conv) -- forward here; their backward-data is the same kernel on the flipped, transposed
//   weights (ops/conv3x3.py).
//
//   out[p, o] = sum_{t < 9, c < C} x[p + shift(t), c] * W[o, t*C + c]      p = (b, y, x): GEMM  M = B*H*W, N = O, K = 9*C
//
// Tiling for 256 CUs: ONE WORKGROUP = ONE IMAGE ROW (up to 224 positions) x 256 output channels.  The canvas has
// B*H = 512 rows of 196 positions: two full rounds of workgroups (a 256 x 256 position tile gives 392 tiles = 1.53
// rounds).  A row tile also makes the A operand trivial: tap (ki, kj) of row y is the row y + ki - 1 shifted by kj - 1
// positions -- 128-byte channel chunks that LDS-DMA copies straight from the map (a line of zeros where the shifted
// position leaves the map), no im2col, no gather arithmetic.
// K runs in steps of 64 channels of one tap, ordered as (kernel row ki, 64-channel chunk) GROUPS of three steps
// kj = 0, 1, 2: the A tile of a group is the input row y + ki - 1, positions x0 - 1 .. x0 + 224, loaded ONCE -- tap kj
// reads it shifted by kj rows.  A tiles are double-buffered and travel a whole group ahead; the B tiles (256 x 128 B,
// L2-resident: every workgroup reads the same 1.2 MB of weights) one per step, two steps ahead in a ring of three; all
// by LDS-DMA (154 KB of LDS), one barrier per step, vmcnt counted by hand (LDS-DMA completes in issue order).
// Waves: 2 (position halves of 112) x 4 (64 output channels), v_mfma_f32_16x16x32_bf16: a k-32 sub-step of a wave is
// 7 position fragments + 4 weight fragments (11 ds_read_b128, hand-counted lgkmcnt, issued a sub-step ahead) for 28
// MFMAs; 28 accumulators of 16 x 16 = 112 VGPRs.  (First form, kept as profiles/experiments/
// conv3x3_mfma_r04_32x32x16_form.hip.txt: 1 x 8 waves of 224 x 32 with the 32 x 32 x 16 MFMA -- every wave re-read the
// whole A tile, 16 KB of LDS reads per k-32 against 11 KB, and the 16 x 16 x 32 shape holds a higher clock under load:
// 137 -> 127 us.)  LDS image: 128-B rows, 16-B chunk c of row r in slot c ^ (r & 7) -- conflict-free for the 16-row
// fragment reads at every row shift kj -- the swizzle applied on the DMA's SOURCE address.
// Epilogue (optional, fused): + bias[o], ReLU, and the canvas' gap pixels forced to zero (`live` byte per position) --
// the `canvas_bias_act` pass of the towers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

typedef __attribute__((ext_vector_type(8))) __bf16 c3_bf16x8;

constexpr int C3_TM = 224, C3_TN = 256, C3_NW = 8;
constexpr int C3_A_ROWS = 232;                               // positions x0 - 1 .. x0 + 224 (226 used) in whole pieces of 8
constexpr int C3_A_BYTES = C3_A_ROWS * 128, C3_B_BYTES = C3_TN * 128;
constexpr int C3_A_STAGES = 2, C3_B_STAGES = 3;
constexpr int C3_B_OFF = C3_A_STAGES * C3_A_BYTES;
constexpr int C3_LIVE_OFF = C3_B_OFF + C3_B_STAGES * C3_B_BYTES;
constexpr int C3_LDS_BYTES = C3_LIVE_OFF + 256;              // 157 952 B
constexpr int C3_A_PIECES = C3_A_BYTES / 1024, C3_B_PIECES = C3_B_BYTES / 1024;   // 29, 32 (1 KiB = 8 rows x 128 B)
constexpr int C3_B_OPS = C3_B_PIECES / C3_NW;                // LDS-DMA operations of one B tile per wave (4)
constexpr int C3_A_OPS_MIN = C3_A_PIECES / C3_NW;            // ... of one A tile: 3 or 4 per wave

__device__ const uint4 c3_zero_line[8] = {};   // 128 B of zeros: what a shifted position outside the map points at

struct C3Geom {
  int B, H, W, C, O;
};

template <int N>
__device__ __forceinline__ void c3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void c3_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
typedef __attribute__((ext_vector_type(4))) unsigned c3_u32x4;
template <int OFF>
__device__ __forceinline__ void c3_lds_read(c3_u32x4& dst, unsigned addr) {
#ifndef C3_AB_NO_LDS
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
#else
  asm volatile("v_mov_b32 %0, %1" : "=v"(dst[0]) : "v"(addr) : "memory");   // timing-only ablation: no LDS traffic
#endif
}
__device__ __forceinline__ void c3_landed(c3_u32x4& v) { asm volatile("" : "+v"(v)); }

// grid: rsdet_xcd_band_grid(row tiles, output-channel tiles); block 512.
typedef __attribute__((ext_vector_type(4))) float c3_f32x4;
constexpr int C3V_MI = 7, C3V_NI = 4, C3V_WM = 112;

// GATE (the backward-data of the SECOND convolution of a conv + ReLU tower, run as this kernel on the flipped weights):
// the result is the gradient of the first convolution's ReLU output c1, so its own backward step -- zero where c1 <= 0,
// and the per-channel sum of what is left (the first convolution's bias gradient) -- happens here: `gate` = c1 (same
// shape as out), `partial` = (O, m_tiles, 2) floats, [0] = this row tile's sum, for rsdet_launch_sums_finish.  The gate
// pass over the tower's widest tensor (read 2, write 1) and its finish launch are gone.
template <bool GATE>
__global__ __launch_bounds__(64 * C3_NW, 1) void conv3x3_fwd_mfma_bf16_kernel(
    const bf16_t* __restrict__ im, const bf16_t* __restrict__ wt, const float* __restrict__ bias,
    const unsigned char* __restrict__ live, C3Geom g, int m_tiles, int n_tiles, int relu, bf16_t* __restrict__ out,
    const bf16_t* __restrict__ gate, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[C3_LDS_BYTES];
  const RsdetBandItem item = rsdet_xcd_band(blockIdx.x, m_tiles, n_tiles);
  if (!item.valid) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (g.W + C3_TM - 1) / C3_TM;
  const int rowid = item.outer / tiles_x, xt = item.outer - rowid * tiles_x;
  const int b = rowid / g.H, y = rowid - b * g.H, x0 = xt * C3_TM;
  const int n_base = item.inner * C3_TN;
  const int K = 9 * g.C, cchunks = g.C >> 6, groups = 3 * cchunks, steps = 3 * groups;
  unsigned char* s_live = lds + C3_LIVE_OFF;
  if (tid < C3_TM) {
    const int x = x0 + tid;
    s_live[tid] = (x < g.W) ? (live ? live[(long long)y * g.W + x] : (unsigned char)1) : (unsigned char)0;
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(c3_zero_line);
  const int prow = lane >> 3;                       // row of the lane inside a piece = (LDS row) & 7
  const int chunk = (lane & 7) ^ prow;              // the source chunk that belongs in the lane's slot
  constexpr int A_ITERS = (C3_A_PIECES + C3_NW - 1) / C3_NW;
  int a_off[A_ITERS];
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int q = (wave + it * C3_NW) * 8 + prow;
    const int xx = x0 - 1 + q;
    a_off[it] = (q <= C3_TM + 1 && xx >= 0 && xx < g.W) ? xx * g.C + chunk * 8 : -1;
  }
  long long b_off[C3_B_OPS];
#pragma unroll
  for (int it = 0; it < C3_B_OPS; ++it) {
    const int r = (wave + it * C3_NW) * 8 + prow;
    b_off[it] = (long long)min(n_base + r, g.O - 1) * K + chunk * 8;
  }
  auto issue_a = [&](int grp) {
    const int ki = grp / cchunks, cc = grp - ki * cchunks;
    unsigned char* stage = lds + (grp & 1) * C3_A_BYTES;
    const int yy = y + ki - 1;
    const bool row_ok = yy >= 0 && yy < g.H;
    const bf16_t* rowp = im + (((long long)b * g.H + (row_ok ? yy : 0)) * g.W) * g.C + cc * 64;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int piece = wave + it * C3_NW;
      if (piece < C3_A_PIECES) {
        const bf16_t* src = (row_ok && a_off[it] >= 0) ? rowp + a_off[it] : zero + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + piece * 1024), 16, 0, 0);
      }
    }
  };
  auto issue_b = [&](int step) {
    const int grp = step / 3, kj = step - grp * 3;
    const int ki = grp / cchunks, cc = grp - ki * cchunks;
    unsigned char* stageb = lds + C3_B_OFF + (step % C3_B_STAGES) * C3_B_BYTES;
    const bf16_t* wbase = wt + (long long)(ki * 3 + kj) * g.C + cc * 64;
#pragma unroll
    for (int it = 0; it < C3_B_OPS; ++it) {
      const int piece = wave + it * C3_NW;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + b_off[it]),
                                       (__attribute__((address_space(3))) void*)(stageb + piece * 1024), 16, 0, 0);
    }
  };

  c3_f32x4 acc[C3V_MI][C3V_NI];
#pragma unroll
  for (int mi = 0; mi < C3V_MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C3V_NI; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;

  const int wm = wave >> 2, wn = wave & 3;
  const int q4 = lane >> 4;
  const int rb = wn * 64 + (lane & 15);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned boff = rb * 128 + ((q4 ^ (rb & 7)) << 4);
  // fragment addresses: position row = wm * 112 + mi * 16 + (lane & 15) + kj, weight row = wn * 64 + ni * 16 + (lane & 15);
  // 16-byte slot (4 ks + (lane >> 4)) ^ (row & 7).  wm * 112, mi * 16, ni * 16 are 0 mod 8 and 4 ks is bit 2 of the
  // slot: one base per lane and step, XOR ks << 6, immediates mi * 2048 / ni * 2048.
  auto a_base = [&](int grp, int kj) {
    const int t = (lane & 15) + kj;
    return lds_base + (grp & 1) * C3_A_BYTES + (wm * C3V_WM + t) * 128 + ((q4 ^ (t & 7)) << 4);
  };
  auto b_base = [&](int step) { return lds_base + C3_B_OFF + (step % C3_B_STAGES) * C3_B_BYTES + boff; };
  c3_u32x4 fa[2][C3V_MI], fb[2][C3V_NI];
#define C3V_RA(buf, mi, ab) c3_lds_read<(mi) * 2048>(fa[buf][mi], ab)
#define C3V_RB(buf, ni, bb) c3_lds_read<(ni) * 2048>(fb[buf][ni], bb)
  // read order of a sub-step: B0, A0 .. A6, B1, B2, B3 -- the order the MFMAs (weight-fragment major) first need them
#define C3V_READ(buf, idx, ab, bb)                  \
  switch (idx) {                                    \
    case 0: C3V_RB(buf, 0, bb); break;              \
    case 1: C3V_RA(buf, 0, ab); break;              \
    case 2: C3V_RA(buf, 1, ab); break;              \
    case 3: C3V_RA(buf, 2, ab); break;              \
    case 4: C3V_RA(buf, 3, ab); break;              \
    case 5: C3V_RA(buf, 4, ab); break;              \
    case 6: C3V_RA(buf, 5, ab); break;              \
    case 7: C3V_RA(buf, 6, ab); break;              \
    case 8: C3V_RB(buf, 1, bb); break;              \
    case 9: C3V_RB(buf, 2, bb); break;              \
    default: C3V_RB(buf, 3, bb); break;             \
  }
  static_assert(C3V_MI == 7 && C3V_NI == 4, "read order and wait counts below are written out for 7 x 4 fragments");
  // One sub-step (k = 32): 28 MFMAs on the fragments in register set CUR, which were all ISSUED during the previous
  // sub-step; the 11 reads of the next sub-step go out one per MFMA into set 1 - CUR.  LDS reads return in issue
  // order: at MFMA j, 11 + min(j + 1, 11) reads have been issued and the first 2 + j (j < 7), 9 / 10 / 11 (from
  // j = 7 / 14 / 21) are needed -- lgkmcnt(10) up to j = 7, then 12, then 11.  When nothing follows (`more` false: the
  // second sub-step of a step -- the next fragments lie behind the barrier) the counts run down: 9 - j, then 2, 1, 0.
#define C3V_SUBSTEP(CUR, more, ab, bb)                                                                                  \
  {                                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < C3V_MI * C3V_NI; ++j) {                                                       \
      const int ni = j / C3V_MI, mi = j - ni * C3V_MI;                                                                  \
      if (more) {                                                                                                       \
        if (j < 11) C3V_READ(1 - CUR, j, ab, bb);                                                                       \
        if (j <= 7) c3_wait_lgkm<10>();                                                                                 \
        else if (j == 14) c3_wait_lgkm<12>();                                                                           \
        else if (j == 21) c3_wait_lgkm<11>();                                                                           \
      } else {                                                                                                          \
        if (j == 0) c3_wait_lgkm<9>();                                                                                  \
        else if (j == 1) c3_wait_lgkm<8>();                                                                             \
        else if (j == 2) c3_wait_lgkm<7>();                                                                             \
        else if (j == 3) c3_wait_lgkm<6>();                                                                             \
        else if (j == 4) c3_wait_lgkm<5>();                                                                             \
        else if (j == 5) c3_wait_lgkm<4>();                                                                             \
        else if (j == 6) c3_wait_lgkm<3>();                                                                             \
        else if (j == 7) c3_wait_lgkm<2>();                                                                             \
        else if (j == 14) c3_wait_lgkm<1>();                                                                            \
        else if (j == 21) c3_wait_lgkm<0>();                                                                            \
      }                                                                                                                 \
      if (ni == 0) c3_landed(fa[CUR][mi]);                                                                              \
      if (mi == 0) c3_landed(fb[CUR][ni]);                                                                              \
      /* D = W-fragment x X-fragment: rows = output channels, columns = positions (4 consecutive channels per lane) */ \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c3_bf16x8, fb[CUR][ni]),                 \
                                                            __builtin_bit_cast(c3_bf16x8, fa[CUR][mi]), acc[mi][ni], 0, \
                                                            0, 0);                                                      \
    }                                                                                                                   \
  }

  // ---- schedule: one barrier per step, at its start: B(s) -- and with it everything older, A(grp) included -- has
  // landed once only the operations issued after it remain: B(s + 1), and in front of that A(grp + 1) when step s - 1
  // opened a group (kj == 1; 3 or 4 operations per wave).  Then the stages step s - 1 used are refilled two steps ahead
  // (B) / a group ahead (A).  (Tried and slower, profiles/experiments/: the synchronisation point in the middle of the
  // step so that the first fragments of step s + 1 are read during step s -- the B ring of three then leaves the DMA one
  // step of lead instead of two, 127 -> 135 us; the DMA operations spread over the MFMA stream or issued half a step
  // apart by the two waves of a SIMD, 136 us.)
  issue_a(0);
  issue_b(0);
  issue_b(1);
  for (int s = 0; s < steps; ++s) {
    const int grp = s / 3, kj = s - grp * 3;
    if (s + 2 >= steps) c3_wait_vm<0>();
    else if (kj == 1) c3_wait_vm<C3_B_OPS + C3_A_OPS_MIN>();
    else c3_wait_vm<C3_B_OPS>();
#ifndef C3_AB_NO_BARRIER
    __syncthreads();
#endif
#ifndef C3_AB_NO_DMA
    if (kj == 0 && grp + 1 < groups) issue_a(grp + 1);      // into the A stage the previous group used
    if (s + 2 < steps) issue_b(s + 2);                      // into the B stage step s - 1 used
#endif
    const unsigned ab = a_base(grp, kj), bb = b_base(s);
#pragma unroll
    for (int idx = 0; idx < 11; ++idx) C3V_READ(0, idx, ab, bb);
    C3V_SUBSTEP(0, true, ab ^ 64u, bb ^ 64u);
    C3V_SUBSTEP(1, false, ab, bb);
  }

  // ---- epilogue: lane holds channels n_base + wn * 64 + ni * 16 + 4 (lane >> 4) + 0..3 of position wm * 112 + mi * 16 + (lane & 15)
  const int ob = n_base + wn * 64 + 4 * q4;
  float bq[C3V_NI][4];
#pragma unroll
  for (int ni = 0; ni < C3V_NI; ++ni)
#pragma unroll
    for (int k = 0; k < 4; ++k) bq[ni][k] = (bias && ob + 16 * ni + k < g.O) ? bias[ob + 16 * ni + k] : 0.f;
  bf16_t* orow = out + (((long long)b * g.H + y) * g.W) * g.O;
  const bf16_t* grow = GATE ? gate + (((long long)b * g.H + y) * g.W) * g.O : nullptr;
  float csum[C3V_NI][4];
#pragma unroll
  for (int ni = 0; ni < C3V_NI; ++ni)
#pragma unroll
    for (int k = 0; k < 4; ++k) csum[ni][k] = 0.f;
#pragma unroll
  for (int mi = 0; mi < C3V_MI; ++mi) {
    const int pl = wm * C3V_WM + mi * 16 + (lane & 15), x = x0 + pl;
    if (x >= g.W) continue;
    const bool lv = s_live[pl] != 0;
#pragma unroll
    for (int ni = 0; ni < C3V_NI; ++ni) {
      const int o = ob + 16 * ni;
      if (o >= g.O) continue;               // (O % 32 == 0 and quads start at multiples of 4: inside or outside as a whole)
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float tt = acc[mi][ni][k] + bq[ni][k];
        if (relu) tt = fmaxf(tt, 0.f);
        v[k] = lv ? tt : 0.f;
      }
      if (GATE) {
        const uint2 sd = *reinterpret_cast<const uint2*>(grow + (long long)x * g.O + o);
        const float s0 = __uint_as_float(sd.x << 16), s1 = __uint_as_float(sd.x & 0xffff0000u);
        const float s2 = __uint_as_float(sd.y << 16), s3 = __uint_as_float(sd.y & 0xffff0000u);
        v[0] = s0 > 0.f ? v[0] : 0.f, v[1] = s1 > 0.f ? v[1] : 0.f;
        v[2] = s2 > 0.f ? v[2] : 0.f, v[3] = s3 > 0.f ? v[3] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) csum[ni][k] += v[k];
      }
      uint2 pk;
      pk.x = f2bf2(v[0], v[1]);
      pk.y = f2bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(orow + (long long)x * g.O + o) = pk;
    }
  }
  if (GATE && partial) {
    // this row tile's channel sums: the 16 lanes of a row hold different positions of the same 4 channels -> butterfly;
    // the two position halves (wm) meet in LDS (every wave is past its last fragment read after the barrier); fixed order
#pragma unroll
    for (int ni = 0; ni < C3V_NI; ++ni)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) csum[ni][k] += __shfl_xor(csum[ni][k], off);
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);                        // [wm][C3_TN]
    if ((lane & 15) == 0) {
#pragma unroll
      for (int ni = 0; ni < C3V_NI; ++ni)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wm * C3_TN + wn * 64 + 16 * ni + 4 * q4 + k] = csum[ni][k];
    }
    __syncthreads();
    if (tid < C3_TN && n_base + tid < g.O) {
      float2 r;
      r.x = red[tid] + red[C3_TN + tid];
      r.y = 0.f;
      *reinterpret_cast<float2*>(partial + ((long long)(n_base + tid) * m_tiles + item.outer) * 2) = r;
    }
  }
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_conv3x3_mfma_supported(int B, int H, int W, int C, int O) {
  if (B < 1 || H < 1 || W < 1 || C < 64 || (C & 63) || O < 32 || (O & 31)) return 0;
  if ((long long)B * H * W * (long long)(C > O ? C : O) >= (1ll << 31)) return 0;
  return 1;
}

// x (B, H, W, C) bf16 channels-last; weight (O, 3, 3, C) bf16 = a channels_last (O, C, 3, 3) tensor's storage; bias (O)
// fp32 or NULL; live (H*W) bytes, the same for every image, or NULL (canvas gap pixels = 0 are written as zeros); out (B, H, W, O) bf16.
extern "C" int rsdet_conv3x3_fwd_mfma_bf16(const uint16_t* x, const uint16_t* weight, const float* bias,
                                           const uint8_t* live, int B, int H, int W, int C, int O, int relu,
                                           uint16_t* out, void* stream) {
  if (!rsdet_conv3x3_mfma_supported(B, H, W, C, O)) return RSDET_EINVAL;
  if (!x || !weight || !out) return RSDET_EINVAL;
  C3Geom g{B, H, W, C, O};
  const int m_tiles = B * H * ((W + C3_TM - 1) / C3_TM);
  const int n_tiles = (O + C3_TN - 1) / C3_TN;
  const dim3 grid((unsigned)rsdet_xcd_band_grid(m_tiles, n_tiles));
  hipLaunchKernelGGL(conv3x3_fwd_mfma_bf16_kernel<false>, grid, dim3(64 * C3_NW), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)weight, bias, live, g, m_tiles, n_tiles, relu, (bf16_t*)out,
                     (const bf16_t*)nullptr, (float*)nullptr);
  return rsdet_launch_status();
}

// Backward-data of the second convolution of a conv + ReLU tower with the FIRST convolution's ReLU / bias backward in the
// epilogue (kernel note above): grad_c1 = [c1 > 0] conv3x3(grad, weight_flipped), grad_bias1[o] = sum over positions of
// grad_c1 (NULL: not formed).  grad (B, H, W, C), c1 / grad_c1 (B, H, W, O) bf16 channels-last; weight_flipped (O, 3, 3, C)
// (ops/weight_prep.py).  ws: rsdet_conv3x3_dgrad_gate_ws_size bytes when grad_bias1 is wanted.
extern "C" size_t rsdet_conv3x3_dgrad_gate_ws_size(int B, int H, int W, int O) {
  if (B < 1 || H < 1 || W < 1 || O < 1) return 0;
  return (size_t)O * ((size_t)B * H * ((W + C3_TM - 1) / C3_TM)) * 2 * sizeof(float);
}
extern "C" int rsdet_conv3x3_dgrad_gate_mfma_bf16(const uint16_t* grad, const uint16_t* weight_flipped, const uint16_t* c1,
                                                  int B, int H, int W, int C, int O, uint16_t* grad_c1, float* grad_bias1,
                                                  void* ws, size_t ws_bytes, void* stream) {
  if (!rsdet_conv3x3_mfma_supported(B, H, W, C, O)) return RSDET_EINVAL;
  if (!grad || !weight_flipped || !c1 || !grad_c1) return RSDET_EINVAL;
  if (grad_bias1 && (!ws || ws_bytes < rsdet_conv3x3_dgrad_gate_ws_size(B, H, W, O))) return RSDET_EINVAL;
  C3Geom g{B, H, W, C, O};
  const int m_tiles = B * H * ((W + C3_TM - 1) / C3_TM);
  const int n_tiles = (O + C3_TN - 1) / C3_TN;
  const dim3 grid((unsigned)rsdet_xcd_band_grid(m_tiles, n_tiles));
  float* partial = grad_bias1 ? (float*)ws : nullptr;
  hipLaunchKernelGGL(conv3x3_fwd_mfma_bf16_kernel<true>, grid, dim3(64 * C3_NW), 0, (hipStream_t)stream,
                     (const bf16_t*)grad, (const bf16_t*)weight_flipped, (const float*)nullptr,
                     (const unsigned char*)nullptr, g, m_tiles, n_tiles, 0, (bf16_t*)grad_c1, (const bf16_t*)c1, partial);
  if (grad_bias1) rsdet_launch_sums_finish(partial, O, m_tiles, nullptr, grad_bias1, (hipStream_t)stream);
  return rsdet_launch_status();
}
