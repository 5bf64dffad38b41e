// conv3x3_wrw_mfma.hip -- weight gradient of the 3x3 / stride 1 / padding 1 convolution of a channels-last bf16 map as a
// split-K implicit GEMM on the matrix cores of gfx950 (the forward is conv3x3_mfma.hip).
//
// Replaces (as the library kernel behind it: bf16 products, fp32 accumulation, one rounding of the result) the
// backward-weight pass of the `ConvModule(256, 256, 3)` tower convolutions of
// /root/reference/python/jdet/models/roi_heads/s2anet_head.py:127-186 on the pyramid canvas (MIOpen: an NHWC
// implicit-GEMM kernel + a zero-fill and a cast launch around it, 186 + ~10 us per call, 8 calls per bf16 step; this kernel: 155 us + a 9 us fold).
//
//   dW[o, (ki, kj), c] = sum_{b, y, x} g[b, y, x, o] * X[b, y + ki - 1, x + kj - 1, c]        GEMM: M = (kj, c), N = o,
//   K = all positions.  BOTH operands are position-major in memory (channels contiguous), i.e. K is the slow index of
//   both: the fragments come out of row-major LDS images through `ds_read_b64_tr_b16` (transposing read: a 16-lane group
//   fetches 4 positions x 16 channels and each lane receives one channel's 4 positions).
//
// Tiling: one workgroup = one kernel row ki x one 64-channel chunk of C (all three taps kj: 192 rows of the result) x
// 256 output channels, over a GROUP of image rows (split-K: 12 result tiles x 21 row groups = 252 workgroups = one
// round on 256 CUs for the 4 x 128 x 196 canvas).  K runs in chunks of 32 positions of one image row: the g tile
// (32 positions x 256 channels, 16 KB) and the X tile (positions x0 - 1 .. x0 + 32 of the input row y + ki - 1, 64
// channels: 34 rows x 128 B, read at three row offsets for the three taps) come by LDS-DMA, two chunks per step, four
// chunks ahead in a ring of six (126 KB), one barrier per step, vmcnt counted by hand.  Waves: 2 (halves of the 192
// result rows) x 4 (64 output channels), v_mfma_f32_16x16x32_bf16 with the X fragment as the first operand, so that a
// lane ends up with four consecutive c of one o (16-byte stores): 6 x 4 accumulators = 96 VGPRs.
// LDS images (swizzles found with a bank model of the transposing read and checked with SQ_LDS_BANK_CONFLICT):
//   g tile: 512-B rows, 16-byte chunk j of row r in slot j ^ (2 (r & 3) | 8 ((r >> 3) & 1))
//   X tile: 128-B rows, chunk j of row r in slot j ^ (2 ((r >> 1) & 1) | 4 ((r >> 3) & 1))
// The fp32 partial tiles of the row groups go to a workspace; a second launch folds them in a fixed order
// (deterministic) and writes the gradient in the weight's layout (O, 3, 3, C) as bf16 or fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

typedef __attribute__((ext_vector_type(8))) __bf16 w3_bf16x8;
typedef __attribute__((ext_vector_type(4))) float w3_f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned w3_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned w3_u32x4;

constexpr int W3_NW = 8;
constexpr int W3_KC = 32;                                  // positions per chunk
constexpr int W3_G_BYTES = W3_KC * 512;                    // 32 positions x 256 output channels
constexpr int W3_X_ROWS = 40;                              // 34 used (x0 - 1 .. x0 + 32), whole pieces of 8 rows
constexpr int W3_X_BYTES = W3_X_ROWS * 128;
constexpr int W3_SLOT = W3_G_BYTES + W3_X_BYTES;           // 21 504 B
constexpr int W3_SLOTS = 6;
constexpr int W3_LDS = W3_SLOT * W3_SLOTS;                 // 129 024 B
constexpr int W3_G_OPS = (W3_G_BYTES / 1024) / W3_NW;      // LDS-DMA operations of a g tile per wave: 2
constexpr int W3_X_PIECES = W3_X_BYTES / 1024;             // 5: waves 0 .. 4 move one each
constexpr int W3_TM = 192, W3_TN = 256;                    // result tile: (kj, c) x o

struct W3Geom {
  int B, H, W, C, O;
};

template <int N>
__device__ __forceinline__ void w3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void w3_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ void w3_tr_read(w3_u32x2& dst, unsigned addr) {
#ifndef W3_AB_NO_LDS
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
#else
  asm volatile("v_mov_b32 %0, %1" : "=v"(dst[0]) : "v"(addr) : "memory");   // timing-only ablation: no LDS traffic
#endif
}
__device__ __forceinline__ void w3_landed(w3_u32x2& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ int w3_fg(int row) { return (2 * (row & 3)) | (8 * ((row >> 3) & 1)); }
__device__ __forceinline__ int w3_fx(int row) { return (2 * ((row >> 1) & 1)) | (4 * ((row >> 3) & 1)); }

// grid: n_groups x (3 * C / 64) x ceil(O / 256) workgroups (logical order: result tile fastest), block 512.
// partial: [group][tile = (ki, cc, ob)][o_local 256][192] fp32.
__global__ __launch_bounds__(64 * W3_NW, 1) void conv3x3_wrw_mfma_bf16_kernel(
    const bf16_t* __restrict__ gmap, const bf16_t* __restrict__ xmap, W3Geom g, int n_groups, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[W3_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cchunks = g.C >> 6;
  // the workgroups that stream the same rows of g (one row group, all result tiles) are neighbours in the logical order,
  // and every XCD takes one contiguous range of it: they meet in one L2 (blockIdx.x, blockIdx.x + 8, ... share an XCD)
  const int n_tiles_all = (int)gridDim.x / n_groups;
  const unsigned lid = rsdet_xcd_contiguous(blockIdx.x, gridDim.x);
  const int grp = (int)lid / n_tiles_all, tile = (int)lid - grp * n_tiles_all;
  const int ob = tile / (3 * cchunks), rem = tile - ob * 3 * cchunks;
  const int ki = rem / cchunks, cc = rem - ki * cchunks;
  const int o_base = ob * W3_TN;
  const int rows = g.B * g.H;
  // rows [r0, r1) of this group: the first (rows % n_groups) groups take one more
  const int per = rows / n_groups, extra = rows - per * n_groups;
  const int r0 = grp * per + min(grp, extra), r1 = r0 + per + (grp < extra ? 1 : 0);
  const int parts = (g.W + W3_KC - 1) / W3_KC;
  const int NC = (r1 - r0) * parts;                       // chunks of this workgroup

  // ---- LDS-DMA through buffer descriptors: the per-lane part of every source address is a CONSTANT 32-bit offset
  // (tile row x row pitch + the swizzled 16-byte chunk), the per-chunk part a scalar offset -- no per-operation 64-bit
  // address arithmetic (the flat-address form of this loop spent 60 of 200 us on it) -- and a lane whose tile row lies
  // outside the map (row tail, halo, image border) gets an offset beyond the buffer: the range check returns zeros.
  constexpr unsigned OOB = 0x80000000u;                   // (buffers are < 2^31 bytes: rsdet_conv3x3_wrw_mfma_supported)
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)gmap, 0, (int)((long long)g.B * g.H * g.W * g.O * 2), 0x00020000);
  // X through a base one position BEFORE the map (tile row 0 is position x0 - 1; never dereferenced there: that lane
  // is out of range whenever x0 = 0)
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(xmap - g.C), 0, (int)(((long long)g.B * g.H * g.W + 1) * g.C * 2), 0x00020000);
  int g_row[W3_G_OPS];
  unsigned g_voff[W3_G_OPS];
#pragma unroll
  for (int it = 0; it < W3_G_OPS; ++it) {
    const int piece = wave + it * W3_NW;
    g_row[it] = piece * 2 + (lane >> 5);                   // tile row of the lane's 16 bytes
    const int j = (lane & 31) ^ w3_fg(g_row[it]);          // the source chunk that belongs in the lane's slot
    g_voff[it] = (o_base + j * 8 < g.O) ? (unsigned)(g_row[it] * g.O + o_base + j * 8) * 2u : OOB;
  }
  const int x_row = wave * 8 + (lane >> 3);               // (waves 0 .. 4 only)
  const unsigned x_voff = x_row < W3_KC + 2 ? (unsigned)(x_row * g.C + ((lane & 7) ^ w3_fx(x_row)) * 8) * 2u : OOB;
  // issue state: chunks are issued in order, so (row, part, y, slot) advance by increments
  int i_row = r0, i_part = 0, i_y = r0 % g.H, i_slot = 0;
  auto issue_next = [&]() {
    const int p0 = i_part * W3_KC;
    const int g_valid = g.W - p0;                                      // tile rows < g_valid hold positions of the row
    const unsigned g_soff = (unsigned)((i_row * g.W + p0) * g.O) * 2u;
    __attribute__((address_space(3))) unsigned char* slot =
        (__attribute__((address_space(3))) unsigned char*)lds + i_slot * W3_SLOT;
#pragma unroll
    for (int it = 0; it < W3_G_OPS; ++it) {
      const unsigned vo = g_row[it] < g_valid ? g_voff[it] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rsrc, slot + (wave + it * W3_NW) * 1024, 16, (int)vo, (int)g_soff, 0, 0);
    }
    if (wave < W3_X_PIECES) {
      const int yy = i_y + ki - 1;
      const bool row_ok = yy >= 0 && yy < g.H;
      const int lo = p0 == 0 ? 1 : 0, hi = g.W - p0 + 1;               // tile rows lo <= r < hi are positions 0 .. W - 1
      const unsigned x_soff = row_ok ? (unsigned)(((i_row + ki - 1) * g.W + p0) * g.C + cc * 64) * 2u : 0u;
      const unsigned vo = (row_ok && x_row >= lo && x_row < hi) ? x_voff : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, slot + W3_G_BYTES + wave * 1024, 16, (int)vo, (int)x_soff, 0, 0);
    }
    if (++i_part == parts) {
      i_part = 0, ++i_row;
      if (++i_y == g.H) i_y = 0;
    }
    if (++i_slot == W3_SLOTS) i_slot = 0;
  };

  // ---- fragment addresses.  Transposing read: lane 4q + p of 16-lane group gq supplies the
  // address of block row q, columns 4p .. 4p + 3; read h (0 / 1) of a fragment covers positions 8 gq + 4 h + q.
  const int wm = wave >> 2, wn = wave & 3;                 // wm: half of the 192 (kj, c) rows; wn: 64 output channels
  const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  // running LDS addresses of the current STEP's first slot (the second chunk of a step is an immediate W3_SLOT away);
  // they advance by two slots per step and wrap after three steps
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  unsigned ga[4];                                          // g fragments: ni = 0..3 (16 output channels each); + 2048 for h = 1
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int row = 8 * gq + q;                            // (h = 1: row + 4 -- the swizzle does not see bit 2)
    const int chunk = (wn * 64 + ni * 16) / 8 + (p >> 1);
    ga[ni] = lds_base + row * 512 + ((chunk ^ w3_fg(row)) << 4) + 8 * (p & 1);
  }
  unsigned xa[6][2];                                       // X fragments: mi = 0..5 -> n-tile wm * 6 + mi = (kj, c16)
#pragma unroll
  for (int mi = 0; mi < 6; ++mi) {
    const int nt = wm * 6 + mi, kj = nt >> 2, c16 = nt & 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = kj + 8 * gq + 4 * h + q;
      const int chunk = c16 * 2 + (p >> 1);
      xa[mi][h] = lds_base + W3_G_BYTES + row * 128 + ((chunk ^ w3_fx(row)) << 4) + 8 * (p & 1);
    }
  }

  w3_f32x4 acc[6][4];
#pragma unroll
  for (int mi = 0; mi < 6; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;

  // ---- fragments.  The g fragments of a chunk (4 x 2 reads, 16 VGPRs) serve all 24 MFMAs of the chunk; the X
  // fragments (2 reads, 4 VGPRs) serve the four MFMAs of one result row tile and are read one tile ahead.  Within a step
  // the second chunk's g fragments are read during the first chunk's MFMAs (two register sets), so only the first
  // chunk after the barrier waits for LDS.  LDS reads return in issue order: lgkmcnt(N) = "all but the N newest".
  w3_u32x2 fg[2][4][2], fx[2][2];
#define W3_RG(buf, ni, OFF)                                        \
  w3_tr_read<(OFF)>(fg[buf][ni][0], ga[ni]);                       \
  w3_tr_read<(OFF) + 2048>(fg[buf][ni][1], ga[ni])
#define W3_RX(xb, mi, OFF)                                         \
  w3_tr_read<(OFF)>(fx[xb][0], xa[mi][0]);                         \
  w3_tr_read<(OFF)>(fx[xb][1], xa[mi][1])
  // The 24 MFMAs of one chunk (PAR = 0 / 1: first / second chunk of the step, i.e. immediate 0 / W3_SLOT) with the g
  // fragments of register set PAR, already issued, as is X tile 0 into fx[0].  NEXT: the second chunk of the step
  // follows -- its g fragments (set 1) and its X tile 0 are issued on the way.  LDS reads return in issue order: when X
  // tile mi is needed, the reads issued after it are X tile mi + 1 (2) and the g pairs of the next chunk issued in
  // iterations mi - 1 and mi (2 each, iterations 1..4).
#define W3_CHUNK(PAR, NEXT)                                                                                            \
  _Pragma("unroll") for (int mi = 0; mi < 6; ++mi) {                                                                   \
    const int xb = mi & 1;                                                                                             \
    if (mi < 5) {                                                                                                      \
      switch (mi) {                                                                                                    \
        case 0: W3_RX(1, 1, (PAR) * W3_SLOT); break;                                                                   \
        case 1: W3_RX(0, 2, (PAR) * W3_SLOT); break;                                                                   \
        case 2: W3_RX(1, 3, (PAR) * W3_SLOT); break;                                                                   \
        case 3: W3_RX(0, 4, (PAR) * W3_SLOT); break;                                                                   \
        default: W3_RX(1, 5, (PAR) * W3_SLOT); break;                                                                  \
      }                                                                                                                \
    } else if (NEXT) {                                                                                                 \
      W3_RX(0, 0, W3_SLOT);                                                                                            \
    }                                                                                                                  \
    const bool gnow = (NEXT) && mi >= 1 && mi <= 4, gprev = (NEXT) && mi >= 2 && mi <= 5;                              \
    if (gnow) {                                                                                                        \
      switch (mi) {                                                                                                    \
        case 1: W3_RG(1, 0, W3_SLOT); break;                                                                           \
        case 2: W3_RG(1, 1, W3_SLOT); break;                                                                           \
        case 3: W3_RG(1, 2, W3_SLOT); break;                                                                           \
        default: W3_RG(1, 3, W3_SLOT); break;                                                                          \
      }                                                                                                                \
    }                                                                                                                  \
    const int newer = ((mi < 5 || (NEXT)) ? 2 : 0) + (gnow ? 2 : 0) + (gprev ? 2 : 0);                                 \
    if (newer == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");                                                 \
    else if (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                            \
    else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                            \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
    w3_landed(fx[xb][0]), w3_landed(fx[xb][1]);                                                                        \
    w3_u32x4 a;                                                                                                        \
    a[0] = fx[xb][0][0], a[1] = fx[xb][0][1], a[2] = fx[xb][1][0], a[3] = fx[xb][1][1];                                \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) {                                                                 \
      if (mi == 0) w3_landed(fg[PAR][ni][0]), w3_landed(fg[PAR][ni][1]);                                               \
      w3_u32x4 b;                                                                                                      \
      b[0] = fg[PAR][ni][0][0], b[1] = fg[PAR][ni][0][1], b[2] = fg[PAR][ni][1][0], b[3] = fg[PAR][ni][1][1];          \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(w3_bf16x8, a),                          \
                                                            __builtin_bit_cast(w3_bf16x8, b), acc[mi][ni], 0, 0, 0);   \
    }                                                                                                                  \
  }

  // ---- schedule: steps of two chunks; chunks 2s + 4, 2s + 5 are issued at the start of step s (into the slots step
  // s - 1 used), so at that point only the operations of chunks 2s + 2, 2s + 3 may still be in flight
  const bool xw = wave < W3_X_PIECES;                      // this wave also moves an X piece per chunk
  for (int c = 0; c < 4 && c < NC; ++c) issue_next();
  // Full steps (two chunks) in a branch-free body, the odd last chunk peeled behind the loop: with "two chunks or one"
  // decided inside the body the two paths' accumulators met in 96 phi copies per step (v_mov_b32: a third as many
  // vector instructions again as the step has MFMAs, issued after them -- found by counting the SQ's VALU instructions
  // against the MFMAs, profiles/r05_roofline.json: 15.1 k vs 3.6 k per wave).
#define W3_STEP_HEAD(s_)                                                                                               \
  {                                                                                                                    \
    const int c0_ = 2 * (s_), ahead = min(NC, c0_ + 4) - min(NC, c0_ + 2); /* chunks issued after this step's: 0..2 */ \
    if (ahead == 2) {                                                                                                  \
      if (xw) w3_wait_vm<2 * (W3_G_OPS + 1)>(); else w3_wait_vm<2 * W3_G_OPS>();                                       \
    } else if (ahead == 1) {                                                                                           \
      if (xw) w3_wait_vm<W3_G_OPS + 1>(); else w3_wait_vm<W3_G_OPS>();                                                 \
    } else {                                                                                                           \
      w3_wait_vm<0>();                                                                                                 \
    }                                                                                                                  \
    W3_BARRIER();                                                                                                      \
    /* (the two chunks of a step sit in consecutive slots: 2s mod 6 is even) */                                        \
    W3_RG(0, 0, 0); W3_RG(0, 1, 0); W3_RG(0, 2, 0); W3_RG(0, 3, 0);                                                    \
    W3_RX(0, 0, 0);                                                                                                    \
    /* the refill of the slots step s - 1 used goes out while those ten reads are in flight */                         \
    W3_REFILL(c0_);                                                                                                    \
  }
#ifndef W3_AB_NO_BARRIER
#define W3_BARRIER() __syncthreads()
#else
#define W3_BARRIER()
#endif
#ifndef W3_AB_NO_DMA
#define W3_REFILL(c0_)                  \
  if ((c0_) + 4 < NC) issue_next();     \
  if ((c0_) + 5 < NC) issue_next()
#else
#define W3_REFILL(c0_)
#endif
  const int full = NC >> 1;
  for (int s = 0; s < full; ++s) {
    W3_STEP_HEAD(s)
    W3_CHUNK(0, true)            // (its first wait, lgkmcnt(2), covers the ten reads above)
    W3_CHUNK(1, false)
    const int adv = ((2 * s) % W3_SLOTS) == W3_SLOTS - 2 ? -(W3_SLOTS - 2) * W3_SLOT : 2 * W3_SLOT;   // next step's first slot
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) ga[ni] += adv;
#pragma unroll
    for (int mi = 0; mi < 6; ++mi) xa[mi][0] += adv, xa[mi][1] += adv;
  }
  if (NC & 1) {
    W3_STEP_HEAD(full)
    W3_CHUNK(0, false)
  }
#undef W3_STEP_HEAD
#undef W3_BARRIER
#undef W3_REFILL

  // ---- epilogue: D[row = (kj, c) local][col = o local]: lane holds rows 4 (lane >> 4) + 0..3 of n-tile mi and column
  // lane & 15 of o-tile ni: four consecutive c of one o -> one 16-byte store into the [o][192] partial tile
  float* pt = partial + ((long long)grp * gridDim.x / n_groups + tile) * (long long)(W3_TN * W3_TM);
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int o = wn * 64 + ni * 16 + (lane & 15);
#pragma unroll
    for (int mi = 0; mi < 6; ++mi) {
      const int n = (wm * 6 + mi) * 16 + 4 * (lane >> 4);
#ifdef W3_AB_NO_EPI
      if (acc[mi][ni][0] == 12345.678f)
#endif
      *reinterpret_cast<float4*>(pt + (long long)o * W3_TM + n) =
          make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
    }
  }
}

// dW[o][t = ki * 3 + kj][c] = sum over the row groups, in group order; one thread per four consecutive c
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_wrw_fold_kernel(const float* __restrict__ partial, W3Geom g, int n_groups,
                                                               int n_tiles, T* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;       // over O * 9 * C / 4
  const int c4 = g.C >> 2;
  if (idx >= (long long)g.O * 9 * c4) return;
  const int cq = (int)(idx % c4), t = (int)((idx / c4) % 9), o = (int)(idx / ((long long)c4 * 9));
  const int c = cq * 4, ki = t / 3, kj = t - ki * 3, cc = c >> 6, cl = c & 63;
  const int cchunks = g.C >> 6;
  const int ob = o / W3_TN, ol = o - ob * W3_TN;
  const int tile = (ob * 3 + ki) * cchunks + cc;
  const float* p = partial + (long long)tile * (W3_TN * W3_TM) + (long long)ol * W3_TM + kj * 64 + cl;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8                         // (eight independent loads in flight; the additions keep their order)
  for (int gi = 0; gi < n_groups; ++gi) {
    const float4 v = *reinterpret_cast<const float4*>(p + (long long)gi * n_tiles * (W3_TN * W3_TM));
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  T* dst = out + ((long long)o * 9 + t) * g.C + c;
  if constexpr (sizeof(T) == 2) {
    uint2 pk;
    pk.x = f2bf2(acc.x, acc.y);
    pk.y = f2bf2(acc.z, acc.w);
    *reinterpret_cast<uint2*>(dst) = pk;
  } else {
    *reinterpret_cast<float4*>(dst) = acc;
  }
}

// The same fold by ROWS for a convolution whose output feeds an eval-mode BatchNorm (ops/bottleneck.py: conv2 / bn2): one
// workgroup per output channel o; row o of dW is multiplied by gamma[o] / sqrt(var[o] + eps) before the rounding (the
// gradient arriving is that of the BatchNorm's OUTPUT), and rowdot[o] = sum_{t, c} w[o][t][c] U[o][t][c] with U the
// unscaled fp32 sum -- sum_p gz[p, o] conv[p, o], what the BatchNorm's scale gradient is formed from
// (rsdet_bn_affine_grads_finish_multi_f32, csrc/bn_act.hip).  Fixed order throughout.
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_wrw_fold_rows_kernel(const float* __restrict__ partial, W3Geom g, int n_groups,
                                                                    int n_tiles, T* __restrict__ out,
                                                                    const float* __restrict__ var,
                                                                    const float* __restrict__ gamma, float eps,
                                                                    const bf16_t* __restrict__ w, float* __restrict__ rowdot) {
  __shared__ float s_dot[256];
  const int o = blockIdx.x, c4 = g.C >> 2, cchunks = g.C >> 6;
  const int ob = o / W3_TN, ol = o - ob * W3_TN;
  float sc = 1.0f / sqrtf(var[o] + eps);
  if (gamma) sc *= gamma[o];
  float d = 0.f;
  for (int idx = threadIdx.x; idx < 9 * c4; idx += 256) {
    const int cq = idx % c4, t = idx / c4;
    const int c = cq * 4, ki = t / 3, kj = t - ki * 3, cc = c >> 6, cl = c & 63;
    const int tile = (ob * 3 + ki) * cchunks + cc;
    const float* p = partial + (long long)tile * (W3_TN * W3_TM) + (long long)ol * W3_TM + kj * 64 + cl;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int gi = 0; gi < n_groups; ++gi) {
      const float4 v = *reinterpret_cast<const float4*>(p + (long long)gi * n_tiles * (W3_TN * W3_TM));
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    const long long e = ((long long)o * 9 + t) * g.C + c;
    const uint2 wq = *reinterpret_cast<const uint2*>(w + e);
    d += acc.x * __uint_as_float(wq.x << 16) + acc.y * __uint_as_float(wq.x & 0xffff0000u) +
         acc.z * __uint_as_float(wq.y << 16) + acc.w * __uint_as_float(wq.y & 0xffff0000u);
    acc.x *= sc, acc.y *= sc, acc.z *= sc, acc.w *= sc;
    T* dst = out + e;
    if constexpr (sizeof(T) == 2) {
      uint2 pk;
      pk.x = f2bf2(acc.x, acc.y);
      pk.y = f2bf2(acc.z, acc.w);
      *reinterpret_cast<uint2*>(dst) = pk;
    } else {
      *reinterpret_cast<float4*>(dst) = acc;
    }
  }
  s_dot[threadIdx.x] = d;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_dot[threadIdx.x] += s_dot[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowdot[o] = s_dot[0];
}

static inline int w3_groups(int B, int H, int C, int O) {
  const int tiles = 3 * (C / 64) * ((O + W3_TN - 1) / W3_TN);
  int n = 256 / tiles;                 // one round of workgroups on 256 CUs
  if (n < 1) n = 1;
  if (n > B * H) n = B * H;
  return n;
}

}  // namespace rsdet

using namespace rsdet;

extern "C" int rsdet_conv3x3_wrw_mfma_supported(int B, int H, int W, int C, int O) {
  if (B < 1 || H < 1 || W < 1 || C < 64 || (C & 63) || O < 8 || (O & 7)) return 0;
  if (((long long)B * H * W + 1) * (long long)(C > O ? C : O) >= (1ll << 30)) return 0;   // buffers < 2^31 bytes
  return 1;
}

extern "C" size_t rsdet_conv3x3_wrw_mfma_ws_size(int B, int H, int W, int C, int O) {
  if (!rsdet_conv3x3_wrw_mfma_supported(B, H, W, C, O)) return 0;
  const size_t tiles = (size_t)3 * (C / 64) * ((O + W3_TN - 1) / W3_TN);
  return (size_t)w3_groups(B, H, C, O) * tiles * W3_TN * W3_TM * sizeof(float);
}

// grad_out (B, H, W, O) and x (B, H, W, C): channels-last bf16; grad_weight (O, 3, 3, C) = the storage of a
// channels_last (O, C, 3, 3) tensor, bf16 (out_bf16 != 0) or fp32.
static int w3_wrw(const uint16_t* grad_out, const uint16_t* x, int B, int H, int W, int C, int O, const float* var,
                  const float* gamma, float eps, const uint16_t* weight, float* rowdot, void* grad_weight, int out_bf16,
                  void* ws, size_t ws_bytes, void* stream) {
  if (!rsdet_conv3x3_wrw_mfma_supported(B, H, W, C, O)) return RSDET_EINVAL;
  if (!grad_out || !x || !grad_weight || !ws || ws_bytes < rsdet_conv3x3_wrw_mfma_ws_size(B, H, W, C, O) ||
      ((uintptr_t)ws & 15))
    return RSDET_EINVAL;
  W3Geom g{B, H, W, C, O};
  const int n_groups = w3_groups(B, H, C, O);
  const int n_tiles = 3 * (C / 64) * ((O + W3_TN - 1) / W3_TN);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3x3_wrw_mfma_bf16_kernel, dim3(n_groups * n_tiles), dim3(64 * W3_NW), 0, s,
                     (const bf16_t*)grad_out, (const bf16_t*)x, g, n_groups, (float*)ws);
  if (var) {
    if (out_bf16)
      hipLaunchKernelGGL((conv3x3_wrw_fold_rows_kernel<bf16_t>), dim3((unsigned)O), dim3(256), 0, s, (const float*)ws, g,
                         n_groups, n_tiles, (bf16_t*)grad_weight, var, gamma, eps, (const bf16_t*)weight, rowdot);
    else
      hipLaunchKernelGGL((conv3x3_wrw_fold_rows_kernel<float>), dim3((unsigned)O), dim3(256), 0, s, (const float*)ws, g,
                         n_groups, n_tiles, (float*)grad_weight, var, gamma, eps, (const bf16_t*)weight, rowdot);
    return rsdet_launch_status();
  }
  const long long n4 = (long long)O * 9 * (C / 4);
  if (out_bf16)
    hipLaunchKernelGGL((conv3x3_wrw_fold_kernel<bf16_t>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                       (const float*)ws, g, n_groups, n_tiles, (bf16_t*)grad_weight);
  else
    hipLaunchKernelGGL((conv3x3_wrw_fold_kernel<float>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                       (const float*)ws, g, n_groups, n_tiles, (float*)grad_weight);
  return rsdet_launch_status();
}

extern "C" int rsdet_conv3x3_wrw_mfma_bf16(const uint16_t* grad_out, const uint16_t* x, int B, int H, int W, int C, int O,
                                           void* grad_weight, int out_bf16, void* ws, size_t ws_bytes, void* stream) {
  return w3_wrw(grad_out, x, B, H, W, C, O, nullptr, nullptr, 0.f, nullptr, nullptr, grad_weight, out_bf16, ws, ws_bytes,
                stream);
}

// ... of a convolution whose output feeds an eval-mode BatchNorm, from the gradient of the BatchNorm's OUTPUT: row o of the
// result times gamma[o] / sqrt(running_var[o] + eps) (gamma NULL: 1), and rowdot[o] = sum weight[o] * (the unscaled fp32
// row) for the BatchNorm's scale gradient (conv3x3_wrw_fold_rows_kernel's note).  weight: the convolution's own (O, 3, 3, C)
// bf16 weight.
extern "C" int rsdet_conv3x3_wrw_mfma_rowscale_bf16(const uint16_t* grad_out, const uint16_t* x, int B, int H, int W, int C,
                                                    int O, const float* running_var, const float* gamma, float eps,
                                                    const uint16_t* weight, float* rowdot, void* grad_weight,
                                                    int out_bf16, void* ws, size_t ws_bytes, void* stream) {
  if (!running_var || !weight || !rowdot) return RSDET_EINVAL;
  return w3_wrw(grad_out, x, B, H, W, C, O, running_var, gamma, eps, weight, rowdot, grad_weight, out_bf16, ws, ws_bytes,
                stream);
}
