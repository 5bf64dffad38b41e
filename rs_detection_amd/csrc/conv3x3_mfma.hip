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
// K runs in steps of 64 channels of one tap; the A tile of a (kernel row, channel chunk) is loaded once and serves its
// three taps, the B tiles (256 x 128 B, L2-resident: every workgroup reads the same 1.2 MB of weights) one per step,
// all by LDS-DMA ahead of their use (154 KB of LDS), one barrier per step.  All eight waves consume: wave w owns output channels 32w .. 32w+31 and
// all seven 32-row position tiles (7 accumulators of 32 x 32 = 112 VGPRs), so a k-16 sub-step is 8 ds_read_b128 for
// 7 MFMAs (v_mfma_f32_32x32x16_bf16).  LDS image as in alignconv_mfma.hip: 128-B rows, 16-B chunk c of row r in slot
// c ^ ((r >> 1) & 7), the swizzle applied on the DMA's SOURCE address.
// Epilogue (optional, fused): + bias[o], ReLU, and the canvas' gap pixels forced to zero (`live` byte per position) --
// the `canvas_bias_act` pass of the towers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

typedef __attribute__((ext_vector_type(8))) __bf16 c3_bf16x8;
typedef __attribute__((ext_vector_type(16))) float c3_f32x16;

constexpr int C3_TM = 224, C3_TN = 256, C3_NW = 8, C3_MI = C3_TM / 32;
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

__device__ __forceinline__ int c3_slot(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

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
// K order: (ki, 64-channel chunk) GROUPS of three steps kj = 0, 1, 2.  The A tile of a group is the input row
// y + ki - 1, positions x0 - 1 .. x0 + 224, loaded ONCE: tap kj reads it shifted by kj rows (the swizzle is a function
// of the LDS row, so shifted fragment reads stay conflict-free).  A tiles are double-buffered and travel a whole group
// (three steps) ahead; B tiles (one per step) travel two steps ahead in a ring of three.  vmcnt is counted by hand:
// LDS-DMA operations complete in issue order, so "all but the N newest" is what a step has to wait for.
__global__ __launch_bounds__(64 * C3_NW, 1) void conv3x3_fwd_mfma_bf16_kernel(
    const bf16_t* __restrict__ im, const bf16_t* __restrict__ wt, const float* __restrict__ bias,
    const unsigned char* __restrict__ live, C3Geom g, int m_tiles, int n_tiles, int relu, bf16_t* __restrict__ out) {
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
    s_live[tid] = (x < g.W) ? (live ? live[((long long)b * g.H + y) * g.W + x] : (unsigned char)1) : (unsigned char)0;
  }

  // ---- per-lane invariants of the LDS-DMA: a lane always moves the same 16 bytes of the same piece rows
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(c3_zero_line);
  const int prow = lane >> 3;                       // row of the lane inside a piece
  constexpr int A_ITERS = (C3_A_PIECES + C3_NW - 1) / C3_NW;
  int a_off[A_ITERS];                               // element offset of the lane's chunk inside an input row, -1: outside
  int a_zoff[A_ITERS];                              // its chunk inside the line of zeros
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int q = (wave + it * C3_NW) * 8 + prow;   // LDS row: position x0 - 1 + q
    const int chunk = (lane & 7) ^ ((q >> 1) & 7);
    const int xx = x0 - 1 + q;
    a_off[it] = (q <= C3_TM + 1 && xx >= 0 && xx < g.W) ? xx * g.C + chunk * 8 : -1;
    a_zoff[it] = chunk * 8;
  }
  long long b_off[C3_B_OPS];                        // element offset of the lane's chunk of its weight rows
#pragma unroll
  for (int it = 0; it < C3_B_OPS; ++it) {
    const int r = (wave + it * C3_NW) * 8 + prow;
    b_off[it] = (long long)min(n_base + r, g.O - 1) * K + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  auto issue_a = [&](int grp) {                     // group = ki * cchunks + cc
    const int ki = grp / cchunks, cc = grp - ki * cchunks;
    unsigned char* stage = lds + (grp & 1) * C3_A_BYTES;
    const int yy = y + ki - 1;
    const bool row_ok = yy >= 0 && yy < g.H;
    const bf16_t* rowp = im + (((long long)b * g.H + (row_ok ? yy : 0)) * g.W) * g.C + cc * 64;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int piece = wave + it * C3_NW;
      if (piece < C3_A_PIECES) {
        const bf16_t* src = (row_ok && a_off[it] >= 0) ? rowp + a_off[it] : zero + a_zoff[it];
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

  c3_f32x16 acc[C3_MI];
#pragma unroll
  for (int mi = 0; mi < C3_MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;

  issue_a(0);
  issue_b(0);
  if (steps > 1) issue_b(1);
  const int rb = wave * 32 + (lane & 31);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  for (int s = 0; s < steps; ++s) {
    const int grp = s / 3, kj = s - grp * 3;
    // B(s) -- and with it everything older, A(grp) included -- has landed once only the operations issued after it remain:
    // B(s + 1), and in front of that A(grp + 1) when step s - 1 opened a group (kj == 1; 3 or 4 operations per wave)
    if (s + 2 >= steps) c3_wait_vm<0>();
    else if (kj == 1) c3_wait_vm<C3_B_OPS + C3_A_OPS_MIN>();
    else c3_wait_vm<C3_B_OPS>();
#ifndef C3_AB_NO_BARRIER
    __syncthreads();        // everybody's pieces are in; everybody is done reading the stages of step s - 1
#endif
#ifndef C3_AB_NO_DMA
    if (kj == 0 && grp + 1 < groups) issue_a(grp + 1);      // into the A stage the previous group used
    if (s + 2 < steps) issue_b(s + 2);                      // into the B stage step s - 1 used
#endif
    // ---- fragments by inline-asm ds_read_b128 with hand-counted lgkmcnt: LDS reads return in issue order, and the
    // pipeline keeps EIGHT of them in flight (one k-sub-step of fragments: B + 7 A) -- the reads of sub-step ks + 1 are
    // issued one by one between the MFMAs of sub-step ks.  Written as plain loads the compiler re-reads each fragment
    // right in front of its MFMA into ONE register set and waits lgkmcnt(0) there: 48 % MFMA utilisation.
    // Fragment addresses: row = mi * 32 + (lane & 31) + kj and 16-byte slot (2 ks + half) ^ ((row >> 1) & 7).  mi * 32
    // drops out of the swizzle (mi * 16 = 0 mod 8) and 2 ks only touches bits 1-2 of the slot, so one per-lane base per
    // step, XORed with ks << 5, plus an IMMEDIATE mi * 4096 serves all 28 A reads (address arithmetic written out per
    // read was ~200 VALU per wave and step: with two waves per SIMD that alone out-issued the 1 792 MFMA cycles).
    const int t = (lane & 31) + kj, half = lane >> 5;
    const unsigned abase = lds_base + (grp & 1) * C3_A_BYTES + t * 128 + ((half ^ ((t >> 1) & 7)) << 4);
    const unsigned bbase = lds_base + C3_B_OFF + (s % C3_B_STAGES) * C3_B_BYTES + rb * 128 + ((half ^ ((rb >> 1) & 7)) << 4);
    c3_u32x4 fa[2][C3_MI], fb[2];
#define C3_READ_A(dst, ks, mi) c3_lds_read<(mi) * 4096>(dst, abase ^ ((ks) << 5))
#define C3_READ_B(dst, ks) c3_lds_read<0>(dst, bbase ^ ((ks) << 5))
    C3_READ_B(fb[0], 0);
    C3_READ_A(fa[0][0], 0, 0); C3_READ_A(fa[0][1], 0, 1); C3_READ_A(fa[0][2], 0, 2); C3_READ_A(fa[0][3], 0, 3);
    C3_READ_A(fa[0][4], 0, 4); C3_READ_A(fa[0][5], 0, 5); C3_READ_A(fa[0][6], 0, 6);
    static_assert(C3_MI == 7, "the fragment reads above and below are written out for seven position tiles");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int mi = 0; mi < C3_MI; ++mi) {
        if (ks + 1 < 4) {
          if (mi == 0) C3_READ_B(fb[(ks + 1) & 1], ks + 1);
          switch (mi) {          // (the offset is an immediate: one case per position tile)
            case 0: C3_READ_A(fa[(ks + 1) & 1][0], ks + 1, 0); break;
            case 1: C3_READ_A(fa[(ks + 1) & 1][1], ks + 1, 1); break;
            case 2: C3_READ_A(fa[(ks + 1) & 1][2], ks + 1, 2); break;
            case 3: C3_READ_A(fa[(ks + 1) & 1][3], ks + 1, 3); break;
            case 4: C3_READ_A(fa[(ks + 1) & 1][4], ks + 1, 4); break;
            case 5: C3_READ_A(fa[(ks + 1) & 1][5], ks + 1, 5); break;
            default: C3_READ_A(fa[(ks + 1) & 1][6], ks + 1, 6); break;
          }
          c3_wait_lgkm<C3_MI + 1>();               // all but the 8 newest: fa[ks][mi] (and fb[ks]) are in
        } else {
          if (mi == 0) c3_wait_lgkm<C3_MI - 1>();
          else if (mi == 1) c3_wait_lgkm<C3_MI - 2>();
          else if (mi == 2) c3_wait_lgkm<C3_MI - 3>();
          else if (mi == 3) c3_wait_lgkm<C3_MI - 4>();
          else if (mi == 4) c3_wait_lgkm<C3_MI - 5>();
          else if (mi == 5) c3_wait_lgkm<C3_MI - 6>();
          else c3_wait_lgkm<0>();
        }
        c3_landed(fa[ks & 1][mi]);
        if (mi == 0) c3_landed(fb[ks & 1]);
#ifndef C3_AB_NO_MFMA
        // D = W-fragment x X-fragment: rows = output channels, columns = positions, so that a lane ends up with FOUR
        // CONSECUTIVE CHANNELS of one position per register quad -- 8-byte stores in the epilogue, not 2-byte ones
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fb[ks & 1]),
                                                          __builtin_bit_cast(c3_bf16x8, fa[ks & 1][mi]), acc[mi], 0, 0, 0);
#else
        acc[mi][0] += __uint_as_float(fa[ks & 1][mi][0]) + __uint_as_float(fb[ks & 1][1]);
#endif
      }
    }
  }

  // ---- epilogue: D[row = output channel][col = position]; col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5):
  // register quad j = e >> 2 of a lane holds channels 8 j + 4 (lane >> 5) + 0..3 of position mi * 32 + (lane & 31)
  const int ob = n_base + wave * 32 + 4 * (lane >> 5);
  float bq[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) bq[j][k] = (bias && ob + 8 * j + k < g.O) ? bias[ob + 8 * j + k] : 0.f;
  bf16_t* orow = out + (((long long)b * g.H + y) * g.W) * g.O;
#pragma unroll
  for (int mi = 0; mi < C3_MI; ++mi) {
    const int pl = mi * 32 + (lane & 31), x = x0 + pl;
    if (x >= g.W) continue;
    const bool lv = s_live[pl] != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = ob + 8 * j;
      if (o >= g.O) continue;               // (O % 32 == 0: a quad is inside or outside as a whole)
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = acc[mi][4 * j + k] + bq[j][k];
        if (relu) t = fmaxf(t, 0.f);
        v[k] = lv ? t : 0.f;
      }
      uint2 pk;
      pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
      pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
#ifdef C3_AB_NO_EPI
      if (v[0] == 12345.678f)
#endif
      *reinterpret_cast<uint2*>(orow + (long long)x * g.O + o) = pk;
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
// fp32 or NULL; live (B*H*W) bytes or NULL (canvas gap pixels = 0 are written as zeros); out (B, H, W, O) bf16.
extern "C" int rsdet_conv3x3_fwd_mfma_bf16(const uint16_t* x, const uint16_t* weight, const float* bias,
                                           const uint8_t* live, int B, int H, int W, int C, int O, int relu,
                                           uint16_t* out, void* stream) {
  if (!rsdet_conv3x3_mfma_supported(B, H, W, C, O)) return RSDET_EINVAL;
  if (!x || !weight || !out) return RSDET_EINVAL;
  C3Geom g{B, H, W, C, O};
  const int m_tiles = B * H * ((W + C3_TM - 1) / C3_TM);
  const int n_tiles = (O + C3_TN - 1) / C3_TN;
  const dim3 grid((unsigned)rsdet_xcd_band_grid(m_tiles, n_tiles));
  hipLaunchKernelGGL(conv3x3_fwd_mfma_bf16_kernel, grid, dim3(64 * C3_NW), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)weight, bias, live, g, m_tiles, n_tiles, relu, (bf16_t*)out);
  return rsdet_launch_status();
}
