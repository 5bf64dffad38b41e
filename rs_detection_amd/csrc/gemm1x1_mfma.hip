// gemm1x1_mfma.hip -- the 1x1 convolutions of the bf16 ResNet trunk as ONE launch each, with the frozen-statistics
// BatchNorm, the residual add and the ReLU in the GEMM's epilogue (gfx950 matrix cores).
//
// Replaces, for a channels-last bf16 map (which IS the row-major (positions, channels) matrix):
//   conv1 / conv3 / downsample[0] + bn + (identity add) + relu of the Bottleneck,
//   /root/reference/python/jdet/models/backbones/resnet.py:57-93 (norm_eval: BatchNorm in eval mode, :177-184, so
//   bn(x) = x * gamma / sqrt(var + eps) + (beta - mean * gamma / sqrt(var + eps)) is a per-channel affine map),
// which the step ran as a library GEMM (ops/conv1x1.py) followed by a `bn_act` pass over the result (csrc/bn_act.hip:
// read conv output + residual, write y).  Here the conv output never reaches HBM:
//   out[p, o] = relu((sum_c x[p, c] W[o, c]) * s[o] + t[o] + res[p, o]).
//
// These GEMMs are STREAMS, not matrix-core problems: M = 4 096 .. 262 144 positions against N, K <= 2 048 channels;
// the bytes of x / res / out set the time (e.g. 64 -> 256 channels at 4 x 256^2: 300 MB for 8.6 GFLOP), so the kernel
// is built to keep HBM loads in flight, not to fill the MFMA pipe:
//   * PERSISTENT workgroups (one per CU): a workgroup owns one 64 NI-channel column panel and walks its share of the
//     128-position row tiles; (tile, 64-channel K step) pairs form ONE flat sequence of steps;
//   * a ring of 3 LDS slots (A tile 128 x 128 B + W tile 64 NI x 128 B), filled by LDS-DMA (global_load_lds, 16 B per
//     lane) TWO steps ahead of the MFMAs across tile boundaries: the next tile's loads are in flight while this
//     tile's epilogue stores leave; `s_waitcnt vmcnt(N)` is counted by hand (loads, LDS-DMA and stores complete in
//     issue order), raw s_barrier (a __syncthreads would drain the DMA queue);
//   * fragments: ds_read_b128 from the XOR-swizzled image (16-byte chunk c of row r in slot c ^ (r & 7), swizzle on the
//     DMA's SOURCE address: conflict-free), issued a k-32 sub-step ahead with counted lgkmcnt;
//     v_mfma_f32_16x16x32_bf16, D = W-fragment x X-fragment (a lane holds 4 consecutive channels of one position);
//   * epilogue in registers: v_permlane16_swap pairs two fragments so that a lane owns EIGHT consecutive channels:
//     16-byte residual loads (issued before the tile's last MFMAs) and 16-byte stores, the affine map from a
//     scale / shift table in LDS.
// 8 waves = 2 (64 positions) x 4 (16 NI channels); NI = 2 (N tile 128) or 4 (N tile 256).
// The arithmetic is the convolution's and the BatchNorm's own: bf16 products, fp32 accumulation, the affine map and the
// residual in fp32, ONE rounding to bf16 (the two-launch form rounded the conv output to bf16 first).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

typedef __attribute__((ext_vector_type(8))) __bf16 g1_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned g1_u32x4;
typedef __attribute__((ext_vector_type(4))) float g1_f32x4;

constexpr int G1_TM = 128, G1_NW = 8, G1_MI = 4;
constexpr int G1_A_BYTES = G1_TM * 128;                                      // 16 384
constexpr int G1_A_OPS = G1_A_BYTES / 1024 / G1_NW;                          // LDS-DMA operations per wave and tile: 2

__device__ const uint4 g1_zero_line[8] = {};   // 128 B of zeros: rows past the end of the matrix

struct G1Geom {
  long long M;
  int N, K;
};

// epilogue parameters (all per output channel n; null = absent)
struct G1Epi {
  const float* mean;     // BatchNorm running mean / var / weight / bias: scale = gamma / sqrt(var + eps),
  const float* var;      //   shift = beta - mean * scale; mean == var == null: plain convolution (scale 1)
  const float* gamma;
  const float* beta;     // with mean == null: a plain bias
  float eps;
  const bf16_t* res;     // [M][N] side operand: EPI 1 the residual added before the activation; EPI 2 the tensor whose
                         //   sign gates the result (y > 0); EPI 3 an addend
  int relu;
  float* partial;        // EPI 2: [(n * S + slice) * 2 + {0, 1}] per-workgroup sums of the gated value and of
  int slices;            //   gated value * (y - beta) / gamma (csrc/bn_act.hip's finish kernel folds them), or null
};
// EPI: 0 = forward, no side operand; 1 = forward + residual;
//      2 = backward-data INTO a BatchNorm + ReLU: out = (acc * [side > 0]) * scale, scale = gamma / sqrt(var + eps) of
//          THAT BatchNorm (mean unused), sums for its beta / gamma gradients with xhat = (side - beta) / gamma;
//      3 = backward-data + addend: out = acc + side (the identity branch's gradient).
constexpr int G1_FWD = 0, G1_FWD_RES = 1, G1_BWD_GATE = 2, G1_BWD_ADD = 3;

template <int N>
__device__ __forceinline__ void g1_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g1_wait_vm_n(int n) {   // n is wave-uniform and one of the multiples of 2 below
  switch (n) {
#define G1_VMCASE(k) case k: g1_wait_vm<k>(); break;
    G1_VMCASE(0) G1_VMCASE(2) G1_VMCASE(4) G1_VMCASE(6) G1_VMCASE(8) G1_VMCASE(10) G1_VMCASE(12) G1_VMCASE(14)
    G1_VMCASE(16) G1_VMCASE(18) G1_VMCASE(20) G1_VMCASE(22) G1_VMCASE(24) G1_VMCASE(26) G1_VMCASE(28) G1_VMCASE(30)
    G1_VMCASE(32) G1_VMCASE(34) G1_VMCASE(36) G1_VMCASE(38) G1_VMCASE(40)
#undef G1_VMCASE
    default: g1_wait_vm<0>(); break;
  }
}
template <int N>
__device__ __forceinline__ void g1_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g1_wait_lgkm_n(int n) {   // n: a compile-time constant after unrolling
  switch (n) {
#define G1_LGCASE(k) case k: g1_wait_lgkm<k>(); break;
    G1_LGCASE(0) G1_LGCASE(1) G1_LGCASE(2) G1_LGCASE(3) G1_LGCASE(4) G1_LGCASE(5) G1_LGCASE(6) G1_LGCASE(7)
    G1_LGCASE(8) G1_LGCASE(9) G1_LGCASE(10) G1_LGCASE(11) G1_LGCASE(12) G1_LGCASE(13) G1_LGCASE(14) G1_LGCASE(15)
#undef G1_LGCASE
    default: g1_wait_lgkm<0>(); break;
  }
}
__device__ __forceinline__ void g1_lds_read(g1_u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ void g1_landed(g1_u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void g1_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void g1_load16(g1_u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// rows (16 lanes) 1 and 3 of x trade places with rows 0 and 2 of y
__device__ __forceinline__ void g1_swap16(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}

// -DG1_STAMP: a diagnostic build (scripts: profiles/scripts/gemm1x1_stamps.py) that records s_memrealtime (100 MHz) at
// the phase boundaries of the first 512 workgroups into a buffer of its own; no output value depends on a stamp.
#ifdef G1_STAMP
__device__ unsigned long long g1_stamp_buf[512 * 64];
#define G1_ST(i)                                                                                              \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 512 && (i) < 64) g1_stamp_buf[blockIdx.x * 64 + (i)] = wall_clock64(); \
  } while (0)
#else
#define G1_ST(i)
#endif

// four swaps behind ONE hazard gap (a v_permlane16_swap needs a wait state after a VALU write of its operands; the four
// pairs are independent)
__device__ __forceinline__ void g1_swap16x4(float& x0, float& y0, float& x1, float& y1, float& x2, float& y2, float& x3,
                                            float& y3) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
               "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
               : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(x3), "+v"(y3));
}
// max(v, 0) as ONE instruction (fmaxf adds a canonicalising v_max before it; a NaN input gives 0 either way)
__device__ __forceinline__ float g1_relu(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}

template <typename T, T V>
struct g1_const {
  static constexpr T value = V;
};

// persistent grid of 8 * n_tiles * mm workgroups (host: ~one per CU); workgroup id -> (XCD x, slot): column panel
// n = slot % n_tiles, row lane = (slot / n_tiles) * 8 + x -- the workgroups that stream the SAME rows sit on one XCD
// (ids x, x + 8, ...: round-robin placement; speed only) and share the rows in its L2.
template <int NI, int EPI, int G1_STAGES>
__global__ __launch_bounds__(64 * G1_NW, 1) void gemm1x1_bn_act_mfma_bf16_kernel(
    const bf16_t* __restrict__ a, const bf16_t* __restrict__ w, G1Geom g, G1Epi e, int m_tiles, int n_tiles,
    bf16_t* __restrict__ out) {
  constexpr int TN = 64 * NI, B_BYTES = TN * 128, SLOT = G1_A_BYTES + B_BYTES, B_OPS = B_BYTES / 1024 / G1_NW;
  constexpr bool RES = EPI != G1_FWD;              // a 16-byte side operand per output chunk
  constexpr int NP = NI / 2;                       // fragment pairs of a wave = 16-byte stores per position
  constexpr int E_OPS = G1_MI * NP, R_OPS = RES ? E_OPS : 0, AB_OPS = G1_A_OPS + B_OPS;
  constexpr int TAB = G1_STAGES * SLOT;            // per-channel tables behind the ring (ONE shared array: a second
  __shared__ __attribute__((aligned(1024))) unsigned char lds[TAB + TN * 16];  // one would drain the DMA queue, guide 5.4(a))
  const int tid = threadIdx.x, lane = tid & 63;
  G1_ST(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = (int)(blockIdx.x & 7u), slot_id = (int)(blockIdx.x >> 3);
  const int n_t = slot_id % n_tiles, lanes_m = (int)(gridDim.x >> 3) / n_tiles * 8;
  const int m_lane = (slot_id / n_tiles) * 8 + xcd;
  if (m_lane >= m_tiles) {                          // an idle row lane still owns a slice of the sums: zeros
    if (EPI == G1_BWD_GATE && e.partial)
      for (int c = tid; c < TN; c += 64 * G1_NW)
        if (n_t * TN + c < g.N)
          *reinterpret_cast<float2*>(e.partial + ((long long)(n_t * TN + c) * e.slices + m_lane) * 2) = make_float2(0.f, 0.f);
    return;
  }
  const int my_tiles = (m_tiles - m_lane + lanes_m - 1) / lanes_m;
  const int KS = g.K >> 6, S = my_tiles * KS;
  const int n_base = n_t * TN;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g1_zero_line);
  const int prow = lane >> 3;                       // row of the lane inside a 1 KiB piece = (LDS row) & 7
  const int chunk = (lane & 7) ^ prow;              // the source chunk that belongs in the lane's slot

  long long b_off[B_OPS];
#pragma unroll
  for (int it = 0; it < B_OPS; ++it) {
    const int r = (wave + it * G1_NW) * 8 + prow;
    b_off[it] = (long long)min(n_base + r, g.N - 1) * g.K + chunk * 8;
  }
  // issue cursor: the flat step the next DMA belongs to
  int i_tile = 0, i_k = 0;
  auto issue = [&](int s) {
    unsigned char* slot = lds + (s % G1_STAGES) * SLOT;
    const long long p0 = ((long long)m_lane + (long long)i_tile * lanes_m) * G1_TM;
    const int k0 = i_k * 64;
#pragma unroll
    for (int it = 0; it < G1_A_OPS; ++it) {
      const int piece = wave + it * G1_NW;
      const long long p = p0 + piece * 8 + prow;
      const bf16_t* src = p < g.M ? a + p * g.K + chunk * 8 + k0 : zero + chunk * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < B_OPS; ++it) {
      const int piece = wave + it * G1_NW;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + b_off[it] + k0),
                                       (__attribute__((address_space(3))) void*)(slot + G1_A_BYTES + piece * 1024), 16,
                                       0, 0);
    }
    if (++i_k == KS) i_k = 0, ++i_tile;
  };

  g1_f32x4 acc[G1_MI][NI];
#pragma unroll
  for (int mi = 0; mi < G1_MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[mi][ni][k] = 0.f;

  const int wm = wave >> 2, wn = wave & 3;
  const int q4 = lane >> 4, l15 = lane & 15;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // fragment addresses: position row = wm * 64 + mi * 16 + (lane & 15), weight row = wn * 16 NI + ni * 16 + (lane & 15);
  // 16-byte slot (4 ks + (lane >> 4)) ^ (row & 7): every row offset above is 0 mod 8, 4 ks is bit 2 of the slot: one
  // base per lane, XOR 64 for the second k-32 sub-step, + mi * 2048 / ni * 2048
  const unsigned a_lane = (unsigned)((wm * 64 + l15) * 128 + ((q4 ^ (l15 & 7)) << 4));
  const unsigned b_lane = (unsigned)(G1_A_BYTES + (wn * 16 * NI + l15) * 128 + ((q4 ^ (l15 & 7)) << 4));
  g1_u32x4 fa[2][G1_MI], fb[2][NI];
  constexpr int NR = G1_MI + NI;                    // fragment reads of a sub-step, in the order B0, A0 .. A3, B1 ..
  auto read_frag = [&](int buf, int idx, unsigned ab, unsigned bb) {
    if (idx == 0) g1_lds_read(fb[buf][0], bb);
    else if (idx <= G1_MI) g1_lds_read(fa[buf][idx - 1], ab + (idx - 1) * 2048);
    else g1_lds_read(fb[buf][idx - G1_MI], bb + (idx - G1_MI) * 2048);
  };
  // One sub-step (k = 32): MI * NI MFMAs (weight-fragment major) on register set `cur`, whose NR reads were all ISSUED
  // during the previous sub-step; with `more`, the NR reads of the next sub-step go out one per MFMA into the other set.
  // LDS reads return in issue order: MFMA j needs the first mi + 2 reads of its set (ni == 0) or the first MI + ni + 1;
  // outstanding may be NR - needed + (next-set reads issued so far).
  // FIRST: the first sub-step of a tile starts its accumulators from the inline constant 0 (the epilogue then has no
  // 64 registers per lane to clear: ~1 of its ~8 vector instructions per element, and the epilogue is VALU-bound).
  auto substep = [&](auto cur_c, auto more_c, auto first_c, unsigned ab, unsigned bb) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;
    const g1_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < G1_MI * NI; ++j) {
      const int ni = j / G1_MI, mi = j - ni * G1_MI;
      if (MORE && j < NR) read_frag(1 - CUR, j, ab, bb);
      if (ni == 0 || mi == 0) {
        const int needed = ni == 0 ? mi + 2 : G1_MI + ni + 1;
        const int issued_next = MORE ? (j + 1 < NR ? j + 1 : NR) : 0;
        g1_wait_lgkm_n(NR - needed + issued_next);
      }
      if (ni == 0) g1_landed(fa[CUR][mi]);
      if (mi == 0) g1_landed(fb[CUR][ni]);
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g1_bf16x8, fb[CUR][ni]),
                                                            __builtin_bit_cast(g1_bf16x8, fa[CUR][mi]),
                                                            FIRST ? zero4 : acc[mi][ni], 0, 0, 0);
    }
  };
  using C0 = g1_const<int, 0>;
  using C1 = g1_const<int, 1>;
  using BT = g1_const<bool, true>;
  using BF = g1_const<bool, false>;

  // ---- the flat step loop.  Vector-memory operations of a wave, in issue order, per step i:
  //   [R residual loads, if i ends a tile] -> [A + B LDS-DMA of step i + 2, if it exists] -> (MFMAs) -> [E stores, if i
  //   ends a tile].  They complete in that order, so "tile s has landed" = all but the N youngest are done, with
  //   N = (stores of step s - 2) + (residual loads + stores of step s - 1) + (DMA of step s + 1).
  constexpr int LEAD = G1_STAGES - 1;               // DMA lead in steps: step s + LEAD is issued during step s
#pragma unroll
  for (int l = 0; l < LEAD; ++l)
    if (l < S) issue(l);
  G1_ST(1);
  // (the tables are built AFTER the first tiles' DMA is in flight -- stamps: 2.8 us from kernel start to the first DMA when
  //  the parameter loads came first; their LDS writes are ordered before the first epilogue by the step barriers)
  // ---- per-channel tables (fp32) in LDS: [0] scale, [1] shift (the forward forms; the backward forms take none)
  constexpr bool TABLES = EPI == G1_FWD || EPI == G1_FWD_RES;
  if (TABLES) {
    float* tab = reinterpret_cast<float*>(lds + TAB);
    for (int c = tid; c < TN; c += 64 * G1_NW) {
      const int o = min(n_base + c, g.N - 1);
      float s1 = 1.f, t1 = 0.f;
      if (e.mean) {
        const float is = 1.0f / sqrtf(e.var[o] + e.eps);
        s1 = e.gamma ? is * e.gamma[o] : is;
        t1 = (e.beta ? e.beta[o] : 0.f) - e.mean[o] * s1;
      } else if (e.beta) {
        t1 = e.beta[o];
      }
      tab[c] = s1, tab[TN + c] = t1;
    }
  }
  g1_u32x4 rres[RES ? E_OPS : 1];
  float sum1[EPI == G1_BWD_GATE ? NP : 1][8];      // EPI 2: this lane's running sums of the gated gradient
#pragma unroll
  for (int pp = 0; pp < (EPI == G1_BWD_GATE ? NP : 1); ++pp)
#pragma unroll
    for (int k = 0; k < 8; ++k) sum1[pp][k] = 0.f;
  int t_idx = 0, k_idx = 0;                        // tile / K step of the current step
  unsigned ended = 0u;                             // bit d - 1: did step s - d end a tile?
  for (int s = 0; s < S; ++s) {
    const bool last_k = k_idx == KS - 1;
    // younger than the DMA of step s (issued during step s - LEAD): the stores of step s - LEAD, then for every step
    // j = s - LEAD + 1 .. s - 1 its residual loads, the DMA of step j + LEAD and its stores
    int young = ((ended >> (LEAD - 1)) & 1u) ? E_OPS : 0;
#pragma unroll
    for (int d = 1; d < LEAD; ++d)
      young += (((ended >> (d - 1)) & 1u) ? R_OPS + E_OPS : 0) + (s - d + LEAD < S ? AB_OPS : 0);
    g1_wait_vm_n(young);
    g1_barrier();
    G1_ST(2 + 4 * s);
    const long long p_tile = ((long long)m_lane + (long long)t_idx * lanes_m) * G1_TM;
    // epilogue geometry of this lane: position rows p_tile + wm * 64 + mi * 16 + l15; after the permlane swap pair p
    // holds channels n_base + wn * 16 NI + 16 (2 p + (q4 & 1)) + 8 (q4 >> 1) + 0..7
    const int ch0 = wn * 16 * NI + 16 * (q4 & 1) + 8 * (q4 >> 1);
    if (RES && last_k) {
#pragma unroll
      for (int mi = 0; mi < G1_MI; ++mi) {
        const long long p = min(p_tile + wm * 64 + mi * 16 + l15, g.M - 1);
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
          g1_load16(rres[RES ? mi * NP + pp : 0], e.res + p * g.N + min(n_base + ch0 + 32 * pp, g.N - 8));
      }
    }
#if !defined(G1_ABL) || G1_ABL != 2
    if (s + LEAD < S) issue(s + LEAD);
#else
    if (s + LEAD < S && s < 1) issue(s + LEAD);     // (ablation 2: compute only -- timing build, wrong values)
#endif
    const unsigned ab = lds_base + (s % G1_STAGES) * SLOT + a_lane, bb = lds_base + (s % G1_STAGES) * SLOT + b_lane;
#if !defined(G1_ABL) || G1_ABL != 1
#pragma unroll
    for (int idx = 0; idx < NR; ++idx) read_frag(0, idx, ab, bb);
    if (k_idx == 0) substep(C0{}, BT{}, BT{}, ab ^ 64u, bb ^ 64u);
    else substep(C0{}, BT{}, BF{}, ab ^ 64u, bb ^ 64u);
    substep(C1{}, BF{}, BF{}, ab, bb);
#endif                                              // (ablation 1: data movement only -- timing build, wrong values)
    G1_ST(3 + 4 * s);
    if (last_k) {
      if (RES) {
        g1_wait_vm_n(s + LEAD < S ? AB_OPS : 0);   // the residual loads are older than the DMA issued above
#pragma unroll
        for (int r = 0; r < E_OPS; ++r) g1_landed(rres[RES ? r : 0]);
      }
      G1_ST(4 + 4 * s);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const int c = ch0 + 32 * pp;
        // (table reads in inline asm: an ordinary LDS load beside outstanding LDS-DMA makes the compiler drain the queue)
        float sc[8], sh[8];
        if (TABLES) {
          g1_u32x4 s_lo, s_hi, t_lo, t_hi;
          const unsigned tb = lds_base + TAB + c * 4;
          g1_lds_read(s_lo, tb), g1_lds_read(s_hi, tb + 16), g1_lds_read(t_lo, tb + TN * 4), g1_lds_read(t_hi, tb + TN * 4 + 16);
          g1_wait_lgkm<0>();
          g1_landed(s_lo), g1_landed(s_hi), g1_landed(t_lo), g1_landed(t_hi);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sc[k] = __uint_as_float(s_lo[k]), sc[4 + k] = __uint_as_float(s_hi[k]);
            sh[k] = __uint_as_float(t_lo[k]), sh[4 + k] = __uint_as_float(t_hi[k]);
          }
        }
        const bool in_n = n_base + c < g.N;         // (N % 32 == 0 and c % 8 == 0: inside or outside as a whole)
#pragma unroll
        for (int mi = 0; mi < G1_MI; ++mi) {
          // (swapped IN PLACE: the accumulators are dead after this epilogue -- the next tile starts from the constant 0 --
          //  but the compiler cannot see that through the flat step loop and would copy all 64 of them first)
          float v[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = acc[mi][2 * pp][k], v[4 + k] = acc[mi][2 * pp + 1][k];
          g1_swap16x4(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7]);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[mi][2 * pp][k] = v[k], acc[mi][2 * pp + 1][k] = v[4 + k];   // (= "acc is dead")
          float side[8];
          if (RES) {
            const g1_u32x4 r = rres[RES ? mi * NP + pp : 0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              side[2 * k] = __uint_as_float(r[k] << 16);
              side[2 * k + 1] = __uint_as_float(r[k] & 0xffff0000u);
            }
          }
          const long long p = p_tile + wm * 64 + mi * 16 + l15;
          if (EPI == G1_BWD_GATE) {
            const bool row_ok = p < g.M;            // (rows past the end carry a clamped side operand: keep them out of the sums)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              v[k] = (side[k] > 0.f && row_ok) ? v[k] : 0.f;
              sum1[pp][k] += v[k];
            }
          } else if (EPI == G1_BWD_ADD) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += side[k];
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], sc[k], sh[k]);
            if (RES) {
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] += side[k];
            }
            if (e.relu) {
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] = g1_relu(v[k]);
            }
          }
          if (p < g.M && in_n) {
            uint4 o;
            o.x = f2bf2(v[0], v[1]);
            o.y = f2bf2(v[2], v[3]);
            o.z = f2bf2(v[4], v[5]);
            o.w = f2bf2(v[6], v[7]);
            *reinterpret_cast<uint4*>(out + p * g.N + n_base + c) = o;
          }
        }
      }
    }
    if (last_k) G1_ST(5 + 4 * s);
    ended = (ended << 1) | (last_k ? 1u : 0u);
    if (++k_idx == KS) k_idx = 0, ++t_idx;
  }
  G1_ST(63);
  if (EPI == G1_BWD_GATE && e.partial) {
    // this workgroup's sums over all its rows: lanes l15 = 0..15 of a 16-lane row hold different positions of the same
    // 8 channels -> butterfly over the row; the two position halves (wm) meet in LDS (the ring is idle now); fixed order
    g1_wait_vm<0>();
    g1_barrier();
    float* red = reinterpret_cast<float*>(lds);                        // [wm][TN]
    const int ch0 = wn * 16 * NI + 16 * (q4 & 1) + 8 * (q4 >> 1);
#pragma unroll
    for (int pp = 0; pp < (EPI == G1_BWD_GATE ? NP : 1); ++pp)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float a1 = sum1[pp][k];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) a1 += __shfl_xor(a1, off);
        if (l15 == 0) red[wm * TN + ch0 + 32 * pp + k] = a1;
      }
    __syncthreads();
    const int slice = (slot_id / n_tiles) * 8 + xcd;                   // = m_lane: the row lanes of this column panel
    for (int c = tid; c < TN; c += 64 * G1_NW)
      if (n_base + c < g.N) {
        *reinterpret_cast<float2*>(e.partial + ((long long)(n_base + c) * e.slices + slice) * 2) =
            make_float2(red[c] + red[TN + c], 0.f);
      }
  }
}

// out[i] = sum_s partial[s][i] in slab order (fp32 sums), written as fp32 or rounded ONCE to bf16: the fold of a split-K
// product (the 1x1 weight gradients of ops/conv1x1.py: S partial (O, C) products of a batched GEMM).  One thread per 4.
// With `var`: row r = i / row_len of the result is multiplied by gamma[r] / sqrt(var[r] + eps) (gamma NULL = 1) before
// the rounding -- the weight gradient of a convolution whose OUTPUT feeds an eval-mode BatchNorm, formed from the
// gradient of the BatchNorm's output (the scale commutes with the sum over positions).
template <typename T>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ partial, int S, long long n,
                                                        T* __restrict__ out, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, float eps, int row_len) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8                         // (eight independent loads in flight; the additions keep their order)
  for (int s = 0; s < S; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (long long)s * n + i);
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  if (var) {                                         // (row_len % 4 == 0: the four values share a row)
    const int r = (int)(i / row_len);
    float sc = 1.0f / sqrtf(var[r] + eps);
    if (gamma) sc *= gamma[r];
    acc.x *= sc, acc.y *= sc, acc.z *= sc, acc.w *= sc;
  }
  st4(out + i, acc);
}

// The same fold by ROWS, for the weight gradient of a convolution whose output feeds an eval-mode BatchNorm: beside the
// scaled, rounded row it leaves rowdot[r] = sum_i w[r][i] U[r][i], U the UNSCALED fp32 sum -- which equals
// sum_p gz[p, r] conv[p, r] (gz the gradient of the BatchNorm's output, conv the convolution's raw output, never stored):
// the scale gradient of that BatchNorm is (rowdot - mean * sum_p gz) / sqrt(var + eps), exact for any gamma
// (rsdet_bn_affine_grads_finish_multi_f32, csrc/bn_act.hip).  TPR threads per row (a power of two <= 256, four
// consecutive elements each, strided over the row), 256 / TPR rows per workgroup; the dot is folded over a row's
// threads by a fixed butterfly.
template <typename T>
__global__ __launch_bounds__(256) void sum_slabs_rows_kernel(const float* __restrict__ partial, int S, int rows, int row_len,
                                                             int tpr, T* __restrict__ out, const float* __restrict__ var,
                                                             const float* __restrict__ gamma, float eps,
                                                             const bf16_t* __restrict__ w, float* __restrict__ rowdot) {
  __shared__ float s_dot[256];
  const int t = threadIdx.x, rl = t / tpr, j = t - rl * tpr;
  const int r = blockIdx.x * (256 / tpr) + rl;
  const long long n = (long long)rows * row_len;
  float d = 0.f;
  if (r < rows) {
    float sc = 1.0f / sqrtf(var[r] + eps);
    if (gamma) sc *= gamma[r];
    for (int i = j * 4; i < row_len; i += tpr * 4) {
      const long long e = (long long)r * row_len + i;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int s = 0; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(partial + (long long)s * n + e);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
      }
      const uint2 wq = *reinterpret_cast<const uint2*>(w + e);
      d += acc.x * __uint_as_float(wq.x << 16) + acc.y * __uint_as_float(wq.x & 0xffff0000u) +
           acc.z * __uint_as_float(wq.y << 16) + acc.w * __uint_as_float(wq.y & 0xffff0000u);
      acc.x *= sc, acc.y *= sc, acc.z *= sc, acc.w *= sc;
      st4(out + e, acc);
    }
  }
  s_dot[t] = d;
  __syncthreads();
  for (int off = tpr >> 1; off > 0; off >>= 1) {      // (tpr is a power of two: the groups never mix)
    if (j < off) s_dot[t] += s_dot[t + off];
    __syncthreads();
  }
  if (j == 0 && r < rows) rowdot[r] = s_dot[t];
}

// out[c][o] = w[o][c] * scale[o], scale[o] = gamma[o] / sqrt(var[o] + eps) (var NULL: 1): the (C, O) operand of the
// backward-data GEMM of a 1x1 convolution (O, C) whose output feeds an eval-mode BatchNorm -- the BatchNorm's backward
// scale rides in the weights, so the GEMM reads the gradient of the BatchNorm's OUTPUT.  32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void weight_transpose_scale_kernel(const bf16_t* __restrict__ w, int O, int C,
                                                                     const float* __restrict__ var,
                                                                     const float* __restrict__ gamma, float eps,
                                                                     bf16_t* __restrict__ out) {
  __shared__ float tile[32][33];
  const int o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = o0 + ty + 8 * j, c = c0 + tx;
    float v = 0.f;
    if (o < O && c < C) {
      v = bf2f(w[(long long)o * C + c]);
      if (var) {
        float sc = 1.0f / sqrtf(var[o] + eps);
        if (gamma) sc *= gamma[o];
        v *= sc;
      }
    }
    tile[ty + 8 * j][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, o = o0 + tx;
    if (o < O && c < C) out[(long long)c * O + o] = f2bf(tile[tx][ty + 8 * j]);
  }
}

// (C, S, 2) partial sums -> dbias[c] = sum_s [0], dweight[c] = sum_s [1]; one wave per channel, fixed order (the twin of
// csrc/bn_act.hip's bn_act_bwd_finish_kernel for the sums the backward-data epilogue leaves)
__global__ __launch_bounds__(256) void g1_sums_finish_kernel(const float* __restrict__ partial, int C, int S,
                                                             float* __restrict__ dweight, float* __restrict__ dbias) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
#pragma unroll 8        // (eight loads in flight; the additions keep their order)
  for (int s = lane; s < S; s += 64) {
    const float2 p = *reinterpret_cast<const float2*>(partial + ((long long)c * S + s) * 2);
    a += p.x, b += p.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off), b += __shfl_down(b, off);
  if (lane == 0) {
    if (dbias) dbias[c] = a;
    if (dweight) dweight[c] = b;
  }
}

}  // namespace rsdet

using namespace rsdet;

static int sum_slabs_launch(const float* partial, int S, long long n, void* out, int out_bf16, const float* var,
                            const float* gamma, float eps, int row_len, const uint16_t* weight, float* rowdot,
                            void* stream) {
  if (S < 1 || n < 0 || (n & 3)) return RSDET_EINVAL;
  if (var && (row_len < 4 || (row_len & 3) || n % row_len)) return RSDET_EINVAL;
  if ((weight == nullptr) != (rowdot == nullptr) || (rowdot && !var)) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!partial || !out) return RSDET_EINVAL;
  if (rowdot) {
    if (n / row_len > 0x7fffffffll) return RSDET_EINVAL;
    const int rows = (int)(n / row_len);
    int tpr = 1;
    while (tpr < 256 && tpr * 2 <= row_len / 4) tpr *= 2;
    const dim3 grid((unsigned)((rows + 256 / tpr - 1) / (256 / tpr)));
    if (out_bf16)
      hipLaunchKernelGGL((sum_slabs_rows_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, partial, S, rows, row_len,
                         tpr, (bf16_t*)out, var, gamma, eps, (const bf16_t*)weight, rowdot);
    else
      hipLaunchKernelGGL((sum_slabs_rows_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, partial, S, rows, row_len,
                         tpr, (float*)out, var, gamma, eps, (const bf16_t*)weight, rowdot);
    return rsdet_launch_status();
  }
  const dim3 grid((unsigned)((n / 4 + 255) / 256));
  if (out_bf16)
    hipLaunchKernelGGL((sum_slabs_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, partial, S, n, (bf16_t*)out,
                       var, gamma, eps, row_len);
  else
    hipLaunchKernelGGL((sum_slabs_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, partial, S, n, (float*)out,
                       var, gamma, eps, row_len);
  return rsdet_launch_status();
}

extern "C" int rsdet_sum_slabs_f32(const float* partial, int S, long long n, void* out, int out_bf16, void* stream) {
  return sum_slabs_launch(partial, S, n, out, out_bf16, nullptr, nullptr, 0.f, 4, nullptr, nullptr, stream);
}

// the same with row r of the (n / row_len, row_len) result scaled by gamma[r] / sqrt(running_var[r] + eps).  weight (bf16,
// the result's shape) and rowdot (n / row_len floats), both or neither: rowdot[r] = sum_i weight[r][i] * (the unscaled
// fp32 sum)[r][i] -- what the BatchNorm's scale gradient is formed from (sum_slabs_rows_kernel's note).
extern "C" int rsdet_sum_slabs_rowscale_f32(const float* partial, int S, long long n, int row_len,
                                            const float* running_var, const float* gamma, float eps,
                                            const uint16_t* weight, float* rowdot, void* out, int out_bf16, void* stream) {
  if (!running_var) return RSDET_EINVAL;
  return sum_slabs_launch(partial, S, n, out, out_bf16, running_var, gamma, eps, row_len, weight, rowdot, stream);
}

// out (C, O) = transpose(weight (O, C)) with column o scaled by gamma[o] / sqrt(running_var[o] + eps) (running_var NULL:
// plain transpose) -- the weight operand of rsdet_conv1x1_dgrad_bf16
extern "C" int rsdet_weight_transpose_scale_bf16(const uint16_t* weight, int O, int C, const float* running_var,
                                                 const float* gamma, float eps, uint16_t* out, void* stream) {
  if (O < 1 || C < 1 || !weight || !out) return RSDET_EINVAL;
  hipLaunchKernelGGL(weight_transpose_scale_kernel, dim3((C + 31) / 32, (O + 31) / 32), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)weight, O, C, running_var, gamma, eps, (bf16_t*)out);
  return rsdet_launch_status();
}

extern "C" int rsdet_gemm1x1_mfma_supported(long long M, int N, int K) {
  if (M < 1 || N < 32 || (N & 31) || K < 64 || (K & 63)) return 0;
  if (M * (long long)(N > K ? N : K) >= (1ll << 40)) return 0;
  return 1;
}

// the persistent grid of one launch: 8 XCD groups x n_tiles column panels x mm row lanes per XCD, ~256 workgroups
struct G1Grid {
  int ni, m_tiles, n_tiles, mm;
  unsigned blocks() const { return (unsigned)(8 * n_tiles * mm); }
  int row_lanes() const { return 8 * mm; }
};
static G1Grid g1_grid(long long M, int N, bool narrow_only) {
  G1Grid r;
  r.m_tiles = (int)((M + G1_TM - 1) / G1_TM);
  // N tile: 256 channels where N fills it AND that still leaves ~a workgroup per CU (row tiles x panels), else 128
  r.ni = (!narrow_only && (N % 256 == 0 || N > 1024) && (long long)r.m_tiles * ((N + 255) / 256) >= 192) ? 4 : 2;
  const int tn = 64 * r.ni;
  r.n_tiles = (N + tn - 1) / tn;
  r.mm = 256 / (8 * r.n_tiles);
  if (r.mm < 1) r.mm = 1;
  const int need = (r.m_tiles + 7) / 8;
  if (r.mm > need) r.mm = need;
  return r;
}

#define G1_LAUNCH_S(NI_, EPI_, ST_)                                                                                      \
  hipLaunchKernelGGL((gemm1x1_bn_act_mfma_bf16_kernel<NI_, EPI_, ST_>), dim3(gr.blocks()), dim3(64 * G1_NW), 0,           \
                     (hipStream_t)stream, (const bf16_t*)a_ptr, (const bf16_t*)b_ptr, g, e, gr.m_tiles, gr.n_tiles,       \
                     (bf16_t*)out)
// Ring depth 3 for both tile widths.  Round 5 measured a 4-slot ring (DMA lead 3 steps) for the 128-channel tile: no
// change on any trunk shape (sum over the 12 shapes of profiles/scripts/gemm1x1_check.py 377.0 vs 376.6 us), and two
// ablation builds (-DG1_ABL=1: no fragment reads / MFMAs; -DG1_ABL=2: no DMA after the first step) BOTH run within 10 %
// of the full kernel (profiles/r05_t_gemm1x1_ablation.txt): the two halves overlap, and each is bound on its own -- the
// movement half at ~12 bytes / clock / CU, i.e. ~85 cycles per 1 KB LDS-DMA piece per CU (32 pieces per step), which is
// the guide's per-piece issue cost, not latency (hence no gain from a longer lead).  40-50 % of the pieces re-load the
// SAME weight panel from L2 for every row tile -- but a build that keeps the panel RESIDENT in LDS (K <= 128 at 256
// channels, K <= 256 at 128; a 5-slot ring of activation tiles beside it) is value-identical and no faster either:
// 128 -> 512 channels at 128^2 with identity 39.8 -> 41.8 us, 64 -> 256 at 256^2 81.2 -> 75.2, 256 -> 1 024 at 64^2 27.9 ->
// 30.8, step unchanged (profiles/r05_v_gemm1x1_resident_panel.txt, source profiles/experiments/
// gemm1x1_resident_panel_r05.hip.txt).  So neither the lead, nor the weight re-loads, nor the MFMA side alone sets
// the time of the epilogue-heavy shapes; what the three experiments leave is the tile boundary itself -- the identity
// loads are issued one step before they are needed and the 16-byte stores of a tile queue in front of the next tile's
// DMA in the in-order vector-memory queue.  In-kernel stamps around the epilogue are the next measurement.
#define G1_LAUNCH_4(EPI_) G1_LAUNCH_S(4, EPI_, 3)
#define G1_LAUNCH_2(EPI_) G1_LAUNCH_S(2, EPI_, 3)

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
  G1Epi e{running_mean, running_var, gamma, beta, eps, (const bf16_t*)residual, relu, nullptr, 0};
  const G1Grid gr = g1_grid(M, N, false);
  const void *a_ptr = x, *b_ptr = weight;
  if (gr.ni == 4) {
    if (residual) G1_LAUNCH_4(G1_FWD_RES); else G1_LAUNCH_4(G1_FWD);
  } else {
    if (residual) G1_LAUNCH_2(G1_FWD_RES); else G1_LAUNCH_2(G1_FWD);
  }
  return rsdet_launch_status();
}

// Backward-data of a 1x1 convolution as the same streaming GEMM: grad_in[p, c] = epi(sum_o grad_out[p, o] wt[c, o]),
// grad_out (M, O), wt (C, O) (rsdet_weight_transpose_scale_bf16 / ops/weight_prep.py), grad_in / side (M, C), all bf16
// row-major.
//   mode 0: epi = identity.
//   mode 2: the convolution's INPUT was side = relu(bn(.)) of an eval-mode BatchNorm: grad_in = [side > 0] acc = the gated
//           gradient of that BatchNorm's OUTPUT (its scale rides in the weights of the next backward step, like this
//           call's own wt may carry the scale of the BatchNorm behind this convolution); per-channel sums of grad_in
//           (= that BatchNorm's grad_beta) with ws: folded into grad_beta when given, else LEFT in ws as (C, S, 2) floats
//           ([0] the sum, [1] zero), S = rsdet_conv1x1_dgrad_slices(M, C, O), for rsdet_bn_affine_grads_finish_multi_f32.
//   mode 3: grad_in = acc + side (the gradient that reaches the same tensor through the identity branch).
// ws: rsdet_conv1x1_dgrad_ws_size(M, C, O) bytes (mode 2 with sums; NULL: no sums).
extern "C" int rsdet_conv1x1_dgrad_slices(long long M, int C, int O) {
  if (!rsdet_gemm1x1_mfma_supported(M, C, O)) return 0;
  return g1_grid(M, C, true).row_lanes();
}
extern "C" size_t rsdet_conv1x1_dgrad_ws_size(long long M, int C, int O) {
  return (size_t)C * rsdet_conv1x1_dgrad_slices(M, C, O) * 2 * sizeof(float);
}

extern "C" int rsdet_conv1x1_dgrad_bf16(const uint16_t* grad_out, const uint16_t* wt, long long M, int C, int O, int mode,
                                        const uint16_t* side, float* grad_beta, void* ws, size_t ws_bytes,
                                        uint16_t* grad_in, void* stream) {
  if (!rsdet_gemm1x1_mfma_supported(M, C, O)) return RSDET_EINVAL;
  if (!grad_out || !wt || !grad_in || (mode != 0 && mode != G1_BWD_GATE && mode != G1_BWD_ADD)) return RSDET_EINVAL;
  if (mode != 0 && !side) return RSDET_EINVAL;
  if (grad_beta && (mode != G1_BWD_GATE || !ws)) return RSDET_EINVAL;
  G1Geom g{M, C, O};
  const void *a_ptr = grad_out, *b_ptr = wt;
  uint16_t* out = grad_in;
  if (mode == G1_BWD_GATE) {
    const G1Grid gr = g1_grid(M, C, true);
    if (ws && ws_bytes < rsdet_conv1x1_dgrad_ws_size(M, C, O)) return RSDET_EINVAL;
    G1Epi e{nullptr, nullptr, nullptr, nullptr, 0.f, (const bf16_t*)side, 0, (float*)ws, gr.row_lanes()};
    G1_LAUNCH_2(G1_BWD_GATE);
    if (grad_beta)
      hipLaunchKernelGGL(g1_sums_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)ws, C,
                         gr.row_lanes(), (float*)nullptr, grad_beta);
    return rsdet_launch_status();
  }
  const G1Grid gr = g1_grid(M, C, false);
  G1Epi e{nullptr, nullptr, nullptr, nullptr, 0.f, (const bf16_t*)side, 0, nullptr, 0};
  if (mode == G1_BWD_ADD) {
    if (gr.ni == 4) G1_LAUNCH_4(G1_BWD_ADD); else G1_LAUNCH_2(G1_BWD_ADD);
  } else {
    if (gr.ni == 4) G1_LAUNCH_4(G1_FWD); else G1_LAUNCH_2(G1_FWD);
  }
  return rsdet_launch_status();
}
#undef G1_LAUNCH_2
#undef G1_LAUNCH_4
#undef G1_LAUNCH_S

#ifdef G1_STAMP
extern "C" int rsdet_g1_stamps_read(unsigned long long* dst, int clear) {
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(rsdet::g1_stamp_buf), sizeof(unsigned long long) * 512 * 64) != hipSuccess)
    return RSDET_ELAUNCH;
  if (clear) {
    static unsigned long long z[512 * 64];
    if (hipMemcpyToSymbol(HIP_SYMBOL(rsdet::g1_stamp_buf), z, sizeof(z)) != hipSuccess) return RSDET_ELAUNCH;
  }
  return RSDET_OK;
}
#endif
