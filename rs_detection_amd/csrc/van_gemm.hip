// van_gemm.hip -- the 1x1 convolutions of a VAN block on NCHW fp32 maps as streaming GEMMs on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fmaf chain), with the block's elementwise passes in their epilogues.
//
// The block of /root/reference/python/jdet/models/backbones/van.py:140-263 (Mlp / AttentionModule / SpatialAttention / Block)
// has five 1x1 convolutions (proj_1, conv1, proj_2, fc1, fc2); an NCHW map IS, per image, the row-major (C, H W) matrix,
// so each convolution is  out[n] (M x P) = W (M x K) . x[n] (K x P)  with the pixels contiguous in both activations and
// every per-channel quantity (bias, layer scale, BatchNorm scale / shift) a per-ROW constant of the result.  Rounds 1-5
// ran them through MIOpen -> rocBLAS small-tile kernels (22 ms of the 70 ms Oriented R-CNN step) with every tail -- bias +
// GELU, the gate product, bias + layer scale + shortcut, and the BatchNorm backward -- as a separate pass over the map.
// Here a tail is the EPILOGUE of the GEMM that produces its input (forward and backward-data; the backward-data GEMM is
// the same kernel on the transposed weights).  ops/van_block.py strings them into one autograd node per Block.
//
//   * tile: 2 x 2 waves, a wave owns (MI x 16) rows x (NI x 16) pixels: 160 x 64 (MI 5, NI 2), 128 x 64 and 64 x 128, and
//     their half-width forms (160 x 32, 128 x 32, 64 x 64) where the wide tile would give a CU fewer than two workgroups;
//   * K in chunks of 32: a ring of LDS slots (W tile TM x 128 B, x tile 32 x TN x 4 B) filled by LDS-DMA (global_load_lds,
//     16 B per lane).  The GEMM's ring has TWO slots (<= 57 KB) so that two workgroups share a CU -- one's prologue and
//     epilogue run under the other's K loop (round 6: 1280 x 320 x 8192 65 -> 57 us; with epilogues 88 -> 70) -- the
//     weight gradient keeps three slots and one workgroup per CU (its doubled split-K partials cost the folds more than
//     the overlap returned).  The swizzles sit on the DMA's SOURCE address:
//       W tile: 16-byte chunk c of row m in slot c ^ ((m >> 1) & 7)    (ds_read_b128 of 16 rows: conflict-free)
//       x tile: 16-pixel block b of row k in block b ^ ((k >> 2) & 1)  (ds_read_b32, lane halves on different rows)
//   * fragments: a lane (m = lane & 15, h = lane >> 4) reads 4 consecutive k of W (k = 16 g + 4 h + j) with one
//     ds_read_b128 and the matching x rows with ds_read_b32; MFMA j of the group sums k slots {16 g + 4 h + j}: any order
//     of k is a valid sum order as long as both operands agree.  Reads of the next 16-k group are issued before the MFMAs
//     of the current one (two register sets); ONE raw s_barrier per chunk, placed mid-chunk so that the first reads of the
//     next chunk overlap this chunk's second half; vmcnt / lgkmcnt counted by hand (a compiler-visible LDS load beside an
//     outstanding LDS-DMA drains the queue);
//   * epilogue: the accumulator tile goes through LDS once so that every lane owns 4 consecutive pixels of one row:
//     16-byte stores, per-row constants as scalars; the side operands (shortcut, gate, GELU' map) of those same pixels are
//     loaded into the lane's registers at KERNEL START (VgSide), so their latency runs under the K loop.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include <type_traits>

#include "rsdet_api_internal.h"

namespace rsdet {

typedef __attribute__((ext_vector_type(4))) float vg_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned vg_u32x4;

constexpr int VG_KC = 32, VG_STAGES = 2, VG_NT = 256;

// epilogues (r = output row = channel, p = pixel; v* per-row vectors, s* side maps in the output's layout)
constexpr int VG_NONE = 0;        // out0 = acc
constexpr int VG_BIAS = 1;        // out0 = acc + v0[r]
constexpr int VG_BIAS_GELU2 = 2;  // t = acc + v0[r]: out0 = t, out1 = GELU(t)
constexpr int VG_GATE2 = 3;       // a = acc + v0[r]: out0 = a, out1 = a * s0
constexpr int VG_AFFINE = 4;      // out0 = s0 * v0[r] + acc * v1[r] + v2[r] (+ s1 * v3[r] when s1)   (v0 NULL: 1)
constexpr int VG_MUL2 = 5;        // out0 = acc * s0, out1 = acc * s1
constexpr int VG_MUL1 = 6;        // out0 = acc * s0

struct VgArgs {
  const float* A;   // (M, K) row-major
  const float* B;   // (n_img, K, P)
  float* out0;      // (n_img, M, P)
  float* out1;
  const float* s0;
  const float* s1;
  const float* v0;
  const float* v1;
  const float* v2;
  const float* v3;
  int M, K, P, n_img;
};

__device__ __forceinline__ float vg_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float vg_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

template <int N>
__device__ __forceinline__ void vg_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void vg_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void vg_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ void vg_read128(vg_u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void vg_read32(float& dst, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// MFMA as volatile asm: the builtin is not ordered against the volatile LDS reads / waits around it, and the compiler then
// CLUSTERS them (40 MFMAs, then 13 reads + 7 LDS-DMA + the waits): a wave issues in order, so everything clustered behind
// the last MFMA of a group runs while the matrix pipe idles -- measured 73 % of the MFMA rate with the data movement
// switched off entirely.  With every instruction of the loop volatile, program order IS issue order and the reads / DMA of
// the next group sit BETWEEN this group's MFMAs.
__device__ __forceinline__ void vg_mfma(vg_f32x4& c, unsigned a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void vg_mfma(vg_f32x4& c, unsigned a, unsigned b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int I, int N, typename F>
__device__ __forceinline__ void vg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    vg_static_for<I + 1, N>(f);
  }
}

// ---- epilogue: accumulators -> LDS (row-major tile) -> 4 consecutive pixels per lane, 16-byte side loads and stores
// The MFMAs are volatile asm: the compiler does not know their results arrive LATE (it would read an accumulator right
// behind the instruction that "wrote" it).  vg_acc_fence = the wait states of the last MFMA, then every accumulator passes
// through an empty asm behind them: reads of acc cannot be scheduled above this point.
template <int MI, int NI>
__device__ __forceinline__ void vg_acc_fence(vg_f32x4 (&acc)[MI][NI]) {
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+a"(acc[mi][ni]));
}

// The epilogue's side operands (the shortcut of AFFINE, the gate's u, GELU' / the gate's two factors of the backward
// epilogues), loaded at the START of the kernel into the registers of the lanes that will consume them: their latency
// (and the 40 - 80 MB they stream per launch) runs under the K loop instead of behind it.  Taken when the side loads were
// two iterations deep in the epilogue's own loop: MUL1 on 1280 x 320 x 8192 88 us against 65 us for the plain GEMM.
template <int MI, int NI, int EPI, int NT = VG_NT>
struct VgSide {
  static constexpr int IT = (2 * MI * 16) * (2 * NI * 16 / 4) / NT;
  static_assert((2 * MI * 16) * (2 * NI * 16 / 4) % NT == 0, "whole epilogue iterations");
  static constexpr bool S0 = EPI == VG_GATE2 || EPI == VG_AFFINE || EPI == VG_MUL2 || EPI == VG_MUL1;
  static constexpr bool S1 = EPI == VG_AFFINE || EPI == VG_MUL2;
  float4 s0[S0 ? IT : 1], s1[S1 ? IT : 1];
  __device__ __forceinline__ void load(const VgArgs& a, int img, int m0, int p0, int tid) {
    constexpr int CPR = 2 * NI * 16 / 4;
    const long long obase = ((long long)img * a.M + m0) * a.P + p0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * NT, r = e / CPR, c4 = (e % CPR) * 4;
      const long long o = obase + (long long)r * a.P + c4;
      if (S0) s0[it] = *reinterpret_cast<const float4*>(a.s0 + o);
      if (S1 && a.s1) s1[it] = *reinterpret_cast<const float4*>(a.s1 + o);
    }
  }
};

template <int MI, int NI, int EPI, int NT = VG_NT>
__device__ __forceinline__ void vg_epilogue(vg_f32x4 (&acc)[MI][NI], unsigned char* lds, const VgArgs& a, int img, int m0,
                                            int p0, int wm, int wn, int l15, int h, int tid,
                                            const VgSide<MI, NI, EPI, NT>& side, bool writes_acc = true) {
  constexpr int TN = 2 * NI * 16, CPR = TN / 4, CROW = TN + 4;
  vg_wait_vm<0>();
  vg_barrier();
  float* ct = reinterpret_cast<float*>(lds);
  if (writes_acc) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          ct[((wm * MI + mi) * 16 + 4 * h + i) * CROW + (wn * NI + ni) * 16 + l15] = acc[mi][ni][i];
  }
  __syncthreads();
  const long long obase = ((long long)img * a.M + m0) * a.P + p0;
#pragma unroll
  for (int it = 0; it < VgSide<MI, NI, EPI, NT>::IT; ++it) {
    const int e = tid + it * NT, r = e / CPR, c4 = (e % CPR) * 4;
    const float4 v = *reinterpret_cast<const float4*>(ct + r * CROW + c4);
    const long long o = obase + (long long)r * a.P + c4;
    const int row = m0 + r;
    float4 y0 = v, y1;
    if (EPI == VG_BIAS || EPI == VG_BIAS_GELU2 || EPI == VG_GATE2) {
      const float b = a.v0 ? a.v0[row] : 0.f;
      y0 = make_float4(v.x + b, v.y + b, v.z + b, v.w + b);
      if (EPI == VG_BIAS_GELU2) y1 = make_float4(vg_gelu(y0.x), vg_gelu(y0.y), vg_gelu(y0.z), vg_gelu(y0.w));
      if (EPI == VG_GATE2) {
        const float4 u = side.s0[it];
        y1 = make_float4(y0.x * u.x, y0.y * u.y, y0.z * u.z, y0.w * u.w);
      }
    } else if (EPI == VG_AFFINE) {
      const float c0 = a.v0 ? a.v0[row] : 1.f, c1 = a.v1[row], c2 = a.v2[row];
      const float4 s = side.s0[it];
      y0 = make_float4(s.x * c0 + v.x * c1 + c2, s.y * c0 + v.y * c1 + c2, s.z * c0 + v.z * c1 + c2,
                       s.w * c0 + v.w * c1 + c2);
      if (a.s1) {
        const float c3 = a.v3[row];
        const float4 q = side.s1[it];
        y0.x += q.x * c3, y0.y += q.y * c3, y0.z += q.z * c3, y0.w += q.w * c3;
      }
    } else if (EPI == VG_MUL2) {
      const float4 u = side.s0[it], w = side.s1[it];
      y0 = make_float4(v.x * u.x, v.y * u.y, v.z * u.z, v.w * u.w);
      y1 = make_float4(v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w);
    } else if (EPI == VG_MUL1) {
      const float4 x = side.s0[it];
      y0 = make_float4(v.x * x.x, v.y * x.y, v.z * x.z, v.w * x.w);
    }
    *reinterpret_cast<float4*>(a.out0 + o) = y0;
    if (EPI == VG_BIAS_GELU2 || EPI == VG_GATE2 || EPI == VG_MUL2) *reinterpret_cast<float4*>(a.out1 + o) = y1;
  }
}

// one workgroup per (image, pixel tile, row tile); id -> tile so that the workgroups of one XCD (ids x, x + 8, ...) walk a
// contiguous range with the row tile fastest: the row tiles that share an x tile share it in that XCD's L2
template <int MI, int NI, int EPI, int ST>
__global__ __launch_bounds__(VG_NT, ST == 2 ? 2 : 1) void van_gemm_f32_kernel(VgArgs a, int m_tiles, int p_tiles) {
  constexpr int TM = 2 * MI * 16, TN = 2 * NI * 16;
  constexpr int A_BYTES = TM * 128, B_BYTES = VG_KC * TN * 4, SLOT = A_BYTES + B_BYTES;
  constexpr int A_OPS = TM / 8 / 4, B_OPS = B_BYTES / 1024 / 4, OPS = A_OPS + B_OPS;     // LDS-DMA operations per wave, chunk
  constexpr int CPR = TN / 4;                        // 16-byte chunks per x-tile row
  constexpr int CROW = TN + 4;                       // the epilogue's tile in LDS: row stride in floats
  static_assert(TM * CROW * 4 <= ST * SLOT, "the accumulator tile must fit in the ring");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[ST * SLOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = m_tiles * p_tiles * a.n_img;
  int t = (int)blockIdx.x;
  if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
  const int mt = t % m_tiles, pt = (t / m_tiles) % p_tiles, img = t / (m_tiles * p_tiles);
  const int m0 = mt * TM, p0 = pt * TN;
  const float* Bn = a.B + (long long)img * a.K * a.P + p0;
  const int nk = a.K / VG_KC;

  // ---- DMA source addresses of this lane (per operation), chunk 0
  const float* a_src[A_OPS];
  const float* b_src[B_OPS];
#pragma unroll
  for (int it = 0; it < A_OPS; ++it) {
    const int piece = wave + 4 * it, r = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    a_src[it] = a.A + (long long)(m0 + r) * a.K + c * 4;
  }
#pragma unroll
  for (int it = 0; it < B_OPS; ++it) {
    const int piece = wave + 4 * it, idx = piece * 64 + lane, r = idx / CPR, c = idx % CPR;
    const int blk = (c >> 2) ^ ((r >> 2) & 1);
    b_src[it] = Bn + (long long)r * a.P + blk * 16 + (c & 3) * 4;
  }
  // LDS-DMA operation k (0 .. OPS - 1) of chunk kc
  auto issue_one = [&](auto k_c, int kc) {
    constexpr int KI = decltype(k_c)::value;
    unsigned char* slot = lds + (kc % ST) * SLOT;
    if constexpr (KI < A_OPS)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[KI] + kc * VG_KC),
                                       (__attribute__((address_space(3))) void*)(slot + (wave + 4 * KI) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(b_src[KI - A_OPS] + (long long)kc * VG_KC * a.P),
          (__attribute__((address_space(3))) void*)(slot + A_BYTES + (wave + 4 * (KI - A_OPS)) * 1024), 16, 0, 0);
  };
  auto issue = [&](int kc) { vg_static_for<0, OPS>([&](auto k_c) { issue_one(k_c, kc); }); };

  VgSide<MI, NI, EPI> side;
  side.load(a, img, m0, p0, tid);
  vg_f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = vg_f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, h = lane >> 4;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // fragment addresses inside a slot (16-k group 0; group 1: W ^ 64 bytes, x + 16 rows)
  const unsigned a_lane = (unsigned)((wm * MI * 16 + l15) * 128 + ((h ^ (l15 >> 1)) << 4));
  unsigned b_lane[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    b_lane[ni] = (unsigned)(A_BYTES + 4 * h * TN * 4 + ((((wn * NI + ni) ^ (h & 1)) << 4) + l15) * 4);

  vg_u32x4 fa[2][MI];
  float fb[2][NI][4];
  constexpr int NREADS = MI + 4 * NI, NMFMA = 4 * NI * MI;
  static_assert(NREADS + OPS <= NMFMA, "the next group's reads and the LDS-DMA fit between this group's MFMAs");
  // fragment read k (0 .. NREADS - 1) of 16-k group G of a slot, into register set BUF
  auto read_one = [&](auto buf_c, auto g_c, auto k_c, unsigned slot_base) {
    constexpr int BUF = decltype(buf_c)::value, G = decltype(g_c)::value, KI = decltype(k_c)::value;
    if constexpr (KI < MI) {
      vg_read128<KI * 2048>(fa[BUF][KI], slot_base + (a_lane ^ (G ? 64u : 0u)));    // (+ mi * 2048 never touches bit 6)
    } else {
      constexpr int ni = (KI - MI) / 4, j = (KI - MI) % 4;
      vg_read32<(G * 16 + j) * TN * 4>(fb[BUF][ni][j], slot_base + b_lane[ni]);
    }
  };
  // the 4 NI MI MFMAs of register set BUF, hook(i) between MFMA i and i + 1
  auto mfma_group = [&](auto buf_c, auto&& hook) {
    constexpr int BUF = decltype(buf_c)::value;
    vg_static_for<0, NMFMA>([&](auto i_c) {
      constexpr int I = decltype(i_c)::value, j = I / (NI * MI), ni = (I / MI) % NI, mi = I % MI;
#if !defined(VG_ABL) || VG_ABL != 1              // (ablation 1: data movement only -- a timing build, wrong values)
      vg_mfma(acc[mi][ni], fa[BUF][mi][j], fb[BUF][ni][j]);
#endif
      hook(i_c);
    });
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

  // ---- prologue: chunks 0 .. 2 in flight, chunk 0 landed, its first group in registers
#pragma unroll
  for (int kc = 0; kc < ST; ++kc)
    if (kc < nk) issue(kc);
  if (ST == 3 && nk >= 3) vg_wait_vm<2 * OPS>(); else if (nk >= 2) vg_wait_vm<OPS>(); else vg_wait_vm<0>();
  vg_barrier();
  vg_static_for<0, NREADS>([&](auto k_c) { read_one(C0{}, C0{}, k_c, lds_base); });
  vg_wait_lgkm0();
  // One chunk.  DMA / MORE are compile-time in the steady state (s + 3 < nk): a run-time test around every read and
  // LDS-DMA operation is a scalar branch in EVERY gap between two MFMAs (20 per chunk) -- the tail (the last three chunks)
  // takes the run-time form.
  auto chunk = [&](int s, auto dma_c, auto more_c, bool dma, bool more) {
    constexpr bool DMA_CT = decltype(dma_c)::value != 0, MORE_CT = decltype(more_c)::value != 0;
    const unsigned slot_base = lds_base + (s % ST) * SLOT;
    const unsigned next_base = lds_base + ((s + 1) % ST) * SLOT;
#if defined(VG_ABL) && VG_ABL >= 3               // (ablation 3: MFMAs only; 4: + the barrier -- timing builds, wrong values)
    mfma_group(C0{}, [&](auto) {});
#if VG_ABL == 4
    vg_barrier();
#endif
    mfma_group(C1{}, [&](auto) {});
    (void)dma, (void)more, (void)slot_base, (void)next_base;
#else
    // first half of chunk s; the reads of its second half ride between the MFMAs
    mfma_group(C0{}, [&](auto i_c) {
      constexpr int I = decltype(i_c)::value;
      if constexpr (I < NREADS) read_one(C1{}, C1{}, i_c, slot_base);
    });
    vg_wait_lgkm0();
    if (MORE_CT || more) {
      // chunk s + 1 landed (mine: all but the operations of chunk s + 2); every wave has chunk s in registers
      // (two stages: chunk s + 2 is not issued yet -- nothing else of mine is in flight)
      if (ST == 3 && (DMA_CT || s + 2 < nk)) vg_wait_vm<OPS>(); else vg_wait_vm<0>();
      vg_barrier();
    }
    // second half; between its MFMAs: the LDS-DMA of chunk s + 3 (into the slot chunk s just left), then the first reads
    // of chunk s + 1
    mfma_group(C1{}, [&](auto i_c) {
      constexpr int I = decltype(i_c)::value;
      if constexpr (I < OPS) {
#if !defined(VG_ABL) || VG_ABL != 2              // (ablation 2: no DMA after the prologue -- a timing build, wrong values)
        if (DMA_CT || dma) issue_one(i_c, s + ST);
#endif
      } else if constexpr (I - OPS < NREADS) {
        if (MORE_CT || more) read_one(C0{}, C0{}, std::integral_constant<int, I - OPS>{}, next_base);
      }
    });
    vg_wait_lgkm0();
#endif
  };
  int s = 0;
  for (; s + ST < nk; ++s) chunk(s, C1{}, C1{}, true, true);
  for (; s < nk; ++s) chunk(s, C0{}, C0{}, false, s + 1 < nk);
  vg_acc_fence(acc);
  vg_epilogue<MI, NI, EPI>(acc, lds, a, img, m0, p0, wm, wn, l15, h, tid, side);
}

// ---- weight gradients: U[m][n] = sum over images and pixels of g[img][m][p] * x[img][n][p]  (M rows of g, N rows of x) ----
// Both operands are pixel-contiguous rows: both tiles take the W-tile layout above (rows x 32 pixels, chunk swizzle), both
// fragments are ds_read_b128.  Split-K over the (image, 32-pixel chunk) sequence: workgroup (split, tile) sums a contiguous
// range of chunks and leaves its TM x 64 partial in partial[split][m][n]; the folds below sum the splits in order.
template <int MI, int NI, int ST>
__global__ __launch_bounds__(VG_NT, ST == 2 ? 2 : 1) void van_wgrad_f32_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                 int M, int N, int P, int n_img, int m_tiles, int n_tiles,
                                                                 int splits, float* __restrict__ partial) {
  constexpr int TM = 2 * MI * 16, TN = 2 * NI * 16;
  constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, SLOT = A_BYTES + B_BYTES;
  constexpr int A_OPS = TM / 8 / 4, B_OPS = TN / 8 / 4, OPS = A_OPS + B_OPS;
  constexpr int CROW = TN + 4;
  static_assert(TM * CROW * 4 <= ST * SLOT, "the accumulator tile must fit in the ring");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[ST * SLOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = m_tiles * n_tiles, total = tiles * splits;
  int t = (int)blockIdx.x;
  if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
  const int tile = t % tiles, split = t / tiles;
  const int mt = tile % m_tiles, nt = tile / m_tiles;
  const int m0 = mt * TM, n0 = nt * TN;
  const int cpi = P / VG_KC;                                   // chunks per image
  const long long Q = (long long)cpi * n_img;
  const int q0 = (int)(Q * split / splits), q1 = (int)(Q * (split + 1) / splits), nk = q1 - q0;

  const float* a_src[A_OPS];
  const float* b_src[B_OPS];
#pragma unroll
  for (int it = 0; it < A_OPS; ++it) {
    const int piece = wave + 4 * it, r = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    a_src[it] = g + (long long)(m0 + r) * P + c * 4;
  }
#pragma unroll
  for (int it = 0; it < B_OPS; ++it) {
    const int piece = wave + 4 * it, r = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[it] = x + (long long)(n0 + r) * P + c * 4;
  }
  auto issue_one = [&](auto k_c, int kc) {
    constexpr int KI = decltype(k_c)::value;
    unsigned char* slot = lds + (kc % ST) * SLOT;
    const int q = q0 + kc, img = q / cpi, px = (q - img * cpi) * VG_KC;
    if constexpr (KI < A_OPS)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[KI] + ((long long)img * M * P + px)),
                                       (__attribute__((address_space(3))) void*)(slot + (wave + 4 * KI) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(b_src[KI - A_OPS] + ((long long)img * N * P + px)),
          (__attribute__((address_space(3))) void*)(slot + A_BYTES + (wave + 4 * (KI - A_OPS)) * 1024), 16, 0, 0);
  };
  auto issue = [&](int kc) { vg_static_for<0, OPS>([&](auto k_c) { issue_one(k_c, kc); }); };

  vg_f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = vg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, h = lane >> 4;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned a_lane = (unsigned)((wm * MI * 16 + l15) * 128 + ((h ^ (l15 >> 1)) << 4));
  const unsigned b_lane = (unsigned)(A_BYTES + (wn * NI * 16 + l15) * 128 + ((h ^ (l15 >> 1)) << 4));
  vg_u32x4 fa[2][MI], fb[2][NI];
  constexpr int NREADS = MI + NI, NMFMA = 4 * NI * MI;
  static_assert(NREADS + OPS <= NMFMA, "the next group's reads and the LDS-DMA fit between this group's MFMAs");
  auto read_one = [&](auto buf_c, auto g_c, auto k_c, unsigned slot_base) {
    constexpr int BUF = decltype(buf_c)::value, G = decltype(g_c)::value, KI = decltype(k_c)::value;
    if constexpr (KI < MI)
      vg_read128<KI * 2048>(fa[BUF][KI], slot_base + (a_lane ^ (G ? 64u : 0u)));
    else
      vg_read128<(KI - MI) * 2048>(fb[BUF][KI - MI], slot_base + (b_lane ^ (G ? 64u : 0u)));
  };
  auto mfma_group = [&](auto buf_c, auto&& hook) {
    constexpr int BUF = decltype(buf_c)::value;
    vg_static_for<0, NMFMA>([&](auto i_c) {
      constexpr int I = decltype(i_c)::value, j = I / (NI * MI), ni = (I / MI) % NI, mi = I % MI;
      vg_mfma(acc[mi][ni], fa[BUF][mi][j], fb[BUF][ni][j]);
      hook(i_c);
    });
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  if (nk > 0) {
#pragma unroll
    for (int kc = 0; kc < ST; ++kc)
      if (kc < nk) issue(kc);
    if (ST == 3 && nk >= 3) vg_wait_vm<2 * OPS>(); else if (nk >= 2) vg_wait_vm<OPS>(); else vg_wait_vm<0>();
    vg_barrier();
    vg_static_for<0, NREADS>([&](auto k_c) { read_one(C0{}, C0{}, k_c, lds_base); });
    vg_wait_lgkm0();
    auto chunk = [&](int s, auto dma_c, auto more_c, bool dma, bool more) {
      constexpr bool DMA_CT = decltype(dma_c)::value != 0, MORE_CT = decltype(more_c)::value != 0;
      const unsigned slot_base = lds_base + (s % ST) * SLOT;
      const unsigned next_base = lds_base + ((s + 1) % ST) * SLOT;
      mfma_group(C0{}, [&](auto i_c) {
        constexpr int I = decltype(i_c)::value;
        if constexpr (I < NREADS) read_one(C1{}, C1{}, i_c, slot_base);
      });
      vg_wait_lgkm0();
      if (MORE_CT || more) {
        if (ST == 3 && (DMA_CT || s + 2 < nk)) vg_wait_vm<OPS>(); else vg_wait_vm<0>();
        vg_barrier();
      }
      mfma_group(C1{}, [&](auto i_c) {
        constexpr int I = decltype(i_c)::value;
        if constexpr (I < OPS) {
          if (DMA_CT || dma) issue_one(i_c, s + ST);
        } else if constexpr (I - OPS < NREADS) {
          if (MORE_CT || more) read_one(C0{}, C0{}, std::integral_constant<int, I - OPS>{}, next_base);
        }
      });
      vg_wait_lgkm0();
    };
    int s = 0;
    for (; s + ST < nk; ++s) chunk(s, C1{}, C1{}, true, true);
    for (; s < nk; ++s) chunk(s, C0{}, C0{}, false, s + 1 < nk);
  }
  vg_acc_fence(acc);
  vg_wait_vm<0>();
  vg_barrier();
  float* ct = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        ct[((wm * MI + mi) * 16 + 4 * h + i) * CROW + (wn * NI + ni) * 16 + l15] = acc[mi][ni][i];
  __syncthreads();
  float* dst = partial + ((long long)split * M + m0) * N + n0;
  for (int e = tid; e < TM * (TN / 4); e += VG_NT) {
    const int r = e / (TN / 4), c4 = (e % (TN / 4)) * 4;
    *reinterpret_cast<float4*>(dst + (long long)r * N + c4) = *reinterpret_cast<const float4*>(ct + r * CROW + c4);
  }
}

// channel sums over a small table of slice partials: sum_j tab[(c * ns + j) * stride + which]
__device__ __forceinline__ float vg_tab_sum(const float* tab, int c, int ns, int stride, int which) {
  float a = 0.f;
  for (int j = 0; j < ns; ++j) a += tab[((long long)c * ns + j) * stride + which];
  return a;
}

// Fold of the split partials of U (M, N), one workgroup per row m (fixed order), with the row's epilogue: the weight
// gradient of a convolution whose OUTPUT (plus bias) is multiplied by a per-channel scale rs (the layer scale) before it
// meets the gradient g the partials were formed from:
//   grad_w[m][n] = rs[m] U[m][n]                                  (rs NULL: 1)
//   grad_b[m]    = rs[m] gs[m],   gs[m] = sum_p g[m, p] from gs_tab
//   grad_rs[m]   = sum_n w[m][n] U[m][n] + bias[m] gs[m] + sc[m] R2[m] + sh[m] R1[m]
//     -- sum_p g (conv + bias + xn): the convolution output is never stored, its product with g is the row dot; the last two
//     terms are the attention's shortcut xn = x sc + sh (r_tab = slice partials of (sum_p g, sum_p g x); NULL: absent).
struct VgRowsFold {
  const float* partial;   // (S, M, N)
  const float* rs;
  const float* w;         // (M, N) or NULL (no row dot)
  const float* gs_tab;    // [(m * gs_ns + j) * gs_stride] or NULL
  const float* bias;
  const float* r_tab;     // (M, r_ns, 2) or NULL
  const float* sc;
  const float* sh;
  float* grad_w;
  float* grad_b;          // or NULL
  float* grad_rs;         // or NULL
  int S, M, N, gs_ns, gs_stride, r_ns;
};
__device__ __forceinline__ void van_fold_rows_body(const VgRowsFold& f, int m);
__global__ __launch_bounds__(256) void van_fold_rows_kernel(VgRowsFold f) { van_fold_rows_body(f, blockIdx.x); }
// up to three folds in one launch (blockIdx.y = fold): the block node issues its three row folds together behind its
// backward -- none of them feeds the data chain
struct VgRowsFold3 {
  VgRowsFold j[3];
};
__global__ __launch_bounds__(256) void van_fold_rows_multi_kernel(VgRowsFold3 jobs) {
  const VgRowsFold& f = jobs.j[blockIdx.y];
  if ((int)blockIdx.x >= f.M) return;
  van_fold_rows_body(f, blockIdx.x);
}
__device__ __forceinline__ void van_fold_rows_body(const VgRowsFold& f, int m) {
  __shared__ float s_dot[256];
  __shared__ float4 s_part[256];
  const int N = f.N;
  const float sc = f.rs ? f.rs[m] : 1.f;
  float d = 0.f;
  // A row narrower than the workgroup (N / 4 < 256 float4 columns: the 320 x 320 weights) would leave most lanes idle behind
  // S dependent-latency loads each: the S partials are dealt to G = 256 / (N / 4) lane groups, whose sums meet in LDS in
  // group order (a fixed order, like the plain loop's).
  const int nc4 = N >> 2, G = nc4 < 256 ? 256 / nc4 : 1;
  const int col = G > 1 ? (int)threadIdx.x % nc4 : 0, grp = G > 1 ? (int)threadIdx.x / nc4 : 0;
  for (int i = (G > 1 ? col : (int)threadIdx.x) * 4; i < N; i += 1024) {
    const long long e = (long long)m * N + i;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (G > 1) {
      if (grp < G) {
#pragma unroll 4
        for (int s = grp; s < f.S; s += G) {
          const float4 v = *reinterpret_cast<const float4*>(f.partial + (long long)s * f.M * N + e);
          acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
        s_part[grp * nc4 + col] = acc;
      }
      __syncthreads();
      if (grp != 0) break;
      for (int g2 = 1; g2 < G; ++g2) {
        const float4 v = s_part[g2 * nc4 + col];
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
      }
    } else {
#pragma unroll 8
      for (int s = 0; s < f.S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(f.partial + (long long)s * f.M * N + e);
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
      }
    }
    if (f.w) {
      const float4 wv = *reinterpret_cast<const float4*>(f.w + e);
      d += acc.x * wv.x + acc.y * wv.y + acc.z * wv.z + acc.w * wv.w;
    }
    *reinterpret_cast<float4*>(f.grad_w + e) = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
  }
  if (!f.grad_b && !f.grad_rs) return;
  s_dot[threadIdx.x] = d;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_dot[threadIdx.x] += s_dot[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float gs = f.gs_tab ? vg_tab_sum(f.gs_tab, m, f.gs_ns, f.gs_stride, 0) : 0.f;
    if (f.grad_b) f.grad_b[m] = sc * gs;
    if (f.grad_rs) {
      float v = s_dot[0] + (f.bias ? f.bias[m] : 0.f) * gs;
      if (f.r_tab) v += f.sc[m] * vg_tab_sum(f.r_tab, m, f.r_ns, 2, 1) + f.sh[m] * vg_tab_sum(f.r_tab, m, f.r_ns, 2, 0);
      f.grad_rs[m] = v;
    }
  }
}

// Fold for a convolution whose INPUT went through a training-mode BatchNorm that was folded into its weights
// (xn = x sc[k] + sh[k], never materialised): the partials are UT[k][o] = sum_p x[k, p] g[o, p] (the TRANSPOSED unscaled
// weight gradient against the RAW input x), one workgroup per input channel k:
//   grad_w[o][k] = sc[k] UT[k][o] + sh[k] gs[o]                        (gs[o] = sum_p g[o, p], from gs_tab)
//   S1 = sum_o wt[k][o] gs[o] + e1,  S2 = sum_o wt[k][o] UT[k][o] + e2   (= sum_p gxn, sum_p gxn x over the map; e1 / e2: what
//        reaches xn beside the convolution -- the attention's own shortcut, ls[k] R1[k] / ls[k] R2[k])
//   grad_gamma[k] = rstd (S2 - mean S1),  grad_beta[k] = S1
//   the BatchNorm backward as per-channel constants of the backward-data GEMM's epilogue (rsdet_van_gemm_f32 epi 4):
//   grad_x = s0 v0 + acc v1 + v2 + x v3:  v1 = sc, v2 = sc (rstd c2 mean - c1), v3 = -sc rstd c2, v0 = 1 + sc ls (ls NULL: 1)
//   with c1 = S1 / cnt, c2 = rstd (S2 - mean S1) / cnt.
struct VgBnFold {
  const float* partial;   // (S, K, O)
  const float* wt;        // (K, O) the convolution weight, transposed
  const float* gs_tab;    // (O, gs_ns, gs_stride): slice partials of sum_p g[o, p] (column 0)
  const float* r_tab;     // (K, r_ns, 2) slice partials of (R1, R2) = (sum_p G, sum_p G x), or NULL
  const float* ls;        // (K) layer scale of the shortcut term, or NULL (with r_tab)
  const float* mean;      // (K) batch mean, rstd, sc, sh of the folded BatchNorm
  const float* rstd;
  const float* sc;
  const float* sh;
  float* grad_w;          // (O, K)
  float* grad_b;          // (O) = gs (written by workgroup 0), or NULL
  float* grad_gamma;      // (K)
  float* grad_beta;
  float* v0;              // (K) each
  float* v1;
  float* v2;
  float* v3;
  int S, K, O, gs_ns, gs_stride, r_ns;
  float cnt;
};
__global__ __launch_bounds__(256) void van_fold_bn_kernel(VgBnFold f) {
  __shared__ float s_a[256], s_b[256];
  const int k = blockIdx.x;
  const float sc = f.sc[k], sh = f.sh[k];
  float d1 = 0.f, d2 = 0.f;
  // (four channels per workgroup with 16-byte transposed stores was measured in round 6: 1.10 -> 1.18 ms per ORCNN step --
  //  the fold is bound by the latency of its O / 256 dependent trips, not by its stores)
  for (int o = threadIdx.x; o < f.O; o += 256) {
    const long long e = (long long)k * f.O + o;
    float u = 0.f;
#pragma unroll 8
    for (int s = 0; s < f.S; ++s) u += f.partial[(long long)s * f.K * f.O + e];
    const float gs = vg_tab_sum(f.gs_tab, o, f.gs_ns, f.gs_stride, 0);
    const float w = f.wt[e];
    d1 += w * gs, d2 += w * u;
    f.grad_w[(long long)o * f.K + k] = sc * u + sh * gs;
    if (k == 0 && f.grad_b) f.grad_b[o] = gs;
  }
  s_a[threadIdx.x] = d1, s_b[threadIdx.x] = d2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_a[threadIdx.x] += s_a[threadIdx.x + off], s_b[threadIdx.x] += s_b[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float S1 = s_a[0], S2 = s_b[0], ls = 0.f;
    if (f.r_tab) {
      ls = f.ls[k];
      S1 += ls * vg_tab_sum(f.r_tab, k, f.r_ns, 2, 0);
      S2 += ls * vg_tab_sum(f.r_tab, k, f.r_ns, 2, 1);
    }
    const float mean = f.mean[k], rstd = f.rstd[k];
    const float gg = rstd * (S2 - mean * S1);
    f.grad_gamma[k] = gg, f.grad_beta[k] = S1;
    const float c1 = S1 / f.cnt, c2 = gg / f.cnt;
    f.v0[k] = 1.f + sc * ls;
    f.v1[k] = sc;
    f.v2[k] = sc * (rstd * c2 * mean - c1);
    f.v3[k] = -sc * rstd * c2;
  }
}

// ---- per-channel reductions over NCHW maps: one workgroup per (plane, slice) ----
// MODE 0: (sum a, sum a b)  -> tab[(c * ns + n * S + s) * 2 + {0, 1}]   (b NULL: the second sum is 0)
// MODE 1: BatchNorm statistics of a: the slice's (mean, M2 about that mean), combined later by Chan's formula
template <int MODE>
__global__ __launch_bounds__(256) void van_chan_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b, int C,
                                                              int P, int S, float* __restrict__ tab) {
  __shared__ float s_a[4], s_b[4];
  const int plane = blockIdx.y, s = blockIdx.x, c = plane % C, n = plane / C;
  const int len = P / S, i0 = s * len;                 // (P % (4 S) == 0: host-checked)
  const float* ap = a + (long long)plane * P + i0;
  const float* bp = b ? b + (long long)plane * P + i0 : nullptr;
  float x0 = 0.f, x1 = 0.f;
  auto wg_sum = [&](float v, float* sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
  };
  if (MODE == 0) {
    for (int i = threadIdx.x * 4; i < len; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(ap + i);
      x0 += (v.x + v.y) + (v.z + v.w);
      if (bp) {
        const float4 w = *reinterpret_cast<const float4*>(bp + i);
        x1 += (v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w);
      }
    }
    x0 = wg_sum(x0, s_a);
    x1 = wg_sum(x1, s_b);
  } else {
    for (int i = threadIdx.x * 4; i < len; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(ap + i);
      x0 += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wg_sum(x0, s_a) / (float)len;
    for (int i = threadIdx.x * 4; i < len; i += 1024) {      // (the slice is in L2 / the infinity cache by now)
      const float4 v = *reinterpret_cast<const float4*>(ap + i);
      const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
      x1 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    x1 = wg_sum(x1, s_b);
    x0 = mean;
  }
  if (threadIdx.x == 0) {
    float* dst = tab + ((long long)c * (gridDim.y / C * S) + n * S + s) * 2;
    dst[0] = x0, dst[1] = x1;
  }
}

// Training-mode BatchNorm folded into the 1x1 convolution behind it (Block.norm1 -> proj_1, Block.norm2 -> fc1;
// van.py:121-122): one workgroup per output row o.  From the slice statistics (ns slices of len pixels per channel):
//   mean, var (biased) -> rstd = 1 / sqrt(var + eps), sc = gamma rstd, sh = beta - mean sc
//   w_out[o][k] = w[o][k] sc[k],   b_out[o] = b[o] + sum_k w[o][k] sh[k]
// workgroup 0 also stores mean / rstd / sc / sh (K each) for the backward and updates the running statistics the way
// nn.BatchNorm2d does (momentum, unbiased variance) and the batch counter.
struct VgBnPrep {
  const float* tab;     // (K, ns, 2): slice (mean, M2)
  const float* gamma;
  const float* beta;
  const float* w;       // (O, K)
  const float* b;       // (O) or NULL
  float* w_out;
  float* b_out;
  float* mean;
  float* rstd;
  float* sc;
  float* sh;
  float* running_mean;  // or NULL
  float* running_var;
  long long* batches;   // num_batches_tracked or NULL
  const float* ls;      // the layer scale / the bias of the LAST convolution of the half this BatchNorm opens (K each), for the
  const float* b2;      //   constants of that convolution's residual epilogue (rsdet_van_gemm_f32 epi 4), or NULL:
  float* e0;            //   e0 = 1 + ls sc (shortcut != 0, else 1),  e2 = ls (b2 + sh) (shortcut != 0, else ls b2)
  float* e2;
  int shortcut;
  int O, K, ns, len;
  float eps, momentum;
};
__global__ __launch_bounds__(256) void van_bn_prep_kernel(VgBnPrep f) {
  __shared__ float s_dot[256];
  const int o = blockIdx.x;
  float d = 0.f;
  for (int k = threadIdx.x; k < f.K; k += 256) {
    // Chan's combination of ns equal-count slices, in slice order
    float mean = 0.f, m2 = 0.f;
    for (int j = 0; j < f.ns; ++j) {
      const float mj = f.tab[((long long)k * f.ns + j) * 2], vj = f.tab[((long long)k * f.ns + j) * 2 + 1];
      const float delta = mj - mean, na = (float)j * f.len, nb = (float)f.len;
      mean += delta * (nb / (na + nb));
      m2 += vj + delta * delta * (na * nb / (na + nb));
    }
    const float cnt = (float)f.ns * f.len, var = m2 / cnt;
    const float rstd = 1.0f / sqrtf(var + f.eps);
    const float sc = f.gamma[k] * rstd, sh = f.beta[k] - mean * sc;
    const float w = f.w[(long long)o * f.K + k];
    f.w_out[(long long)o * f.K + k] = w * sc;
    d += w * sh;
    if (o == 0) {
      f.mean[k] = mean, f.rstd[k] = rstd, f.sc[k] = sc, f.sh[k] = sh;
      if (f.ls) {
        const float ls = f.ls[k], b2 = f.b2 ? f.b2[k] : 0.f;
        f.e0[k] = f.shortcut ? 1.f + ls * sc : 1.f;
        f.e2[k] = f.shortcut ? ls * (b2 + sh) : ls * b2;
      }
      if (f.running_mean) {
        f.running_mean[k] = (1.f - f.momentum) * f.running_mean[k] + f.momentum * mean;
        f.running_var[k] = (1.f - f.momentum) * f.running_var[k] + f.momentum * (m2 / (cnt - 1.f));
      }
    }
  }
  s_dot[threadIdx.x] = d;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_dot[threadIdx.x] += s_dot[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    f.b_out[o] = (f.b ? f.b[o] : 0.f) + s_dot[0];
    if (o == 0 && f.batches) *f.batches += 1;
  }
}

// up to 5 weight transposes of one launch: dst[j] (K, O) = transpose of src[j] (O, K) with row o scaled by rs[j][o] (NULL: 1)
struct VgTransposeJobs {
  const float* src[5];
  const float* rs[5];
  float* dst[5];
  int O[5], K[5], block0[6];
  int n;
};
__global__ __launch_bounds__(256) void van_transposes_kernel(VgTransposeJobs jobs) {
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.block0[j + 1]) ++j;
  const int O = jobs.O[j], K = jobs.K[j], tk = (K + 31) / 32;
  const int b = (int)blockIdx.x - jobs.block0[j], o0 = (b / tk) * 32, k0 = (b % tk) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o0 + ty + 8 * i, k = k0 + tx;
    float v = 0.f;
    if (o < O && k < K) v = jobs.src[j][(long long)o * K + k] * (jobs.rs[j] ? jobs.rs[j][o] : 1.f);
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, o = o0 + tx;
    if (o < O && k < K) jobs.dst[j][(long long)k * O + o] = tile[tx][ty + 8 * i];
  }
}

struct VgTile {
  int mi, ni;
};
// The tile for (M, P): rows must divide.  Two workgroups share a CU (two-slot rings, below), so that one's prologue and
// epilogue run under the other's K loop -- which takes at least two tiles per CU: when the wide tile gives fewer than 512,
// the tile of half the pixel width is taken (twice the tiles; the weight tile is read by twice as many workgroups, from L2).
static bool vg_pick(int M, int K, int P, int n_img, VgTile* t) {
  if (M < 64 || K < VG_KC || (K % VG_KC) || P < 64 || n_img < 1) return false;
  if (M % 160 == 0 && P % 64 == 0) *t = VgTile{5, 2};
  else if (M % 128 == 0 && P % 64 == 0) *t = VgTile{4, 2};
  else if (M % 64 == 0 && P % 128 == 0) *t = VgTile{2, 4};
  else return false;
  const long long tiles = (long long)(M / (32 * t->mi)) * (P / (32 * t->ni)) * n_img;
  if (tiles < 256 || (tiles < 512 && K <= 512)) t->ni /= 2;      // (a long K loop amortises its ends by itself)
  return true;
}

}  // namespace rsdet

using namespace rsdet;

// 1 when rsdet_van_gemm_f32 takes out (n_img, M, P) = W (M, K) . x (n_img, K, P): M a multiple of 160, 128 or 64, K of 32,
// P of 64 (128 for M % 128 != 0)
extern "C" int rsdet_van_gemm_f32_supported(int M, int K, int P, int n_img) {
  VgTile t;
  return vg_pick(M, K, P, n_img, &t) ? 1 : 0;
}

#define VG_LAUNCH(MI_, NI_, EPI_)                                                                                       \
  hipLaunchKernelGGL((van_gemm_f32_kernel<MI_, NI_, EPI_, VG_STAGES>), dim3((unsigned)(m_tiles * p_tiles * n_img)),      \
                     dim3(VG_NT), 0, (hipStream_t)stream, a, m_tiles, p_tiles)
#define VG_EPI_SWITCH(MI_, NI_)                                                                                         \
  switch (epi) {                                                                                                        \
    case VG_NONE: VG_LAUNCH(MI_, NI_, VG_NONE); break;                                                                  \
    case VG_BIAS: VG_LAUNCH(MI_, NI_, VG_BIAS); break;                                                                  \
    case VG_BIAS_GELU2: VG_LAUNCH(MI_, NI_, VG_BIAS_GELU2); break;                                                      \
    case VG_GATE2: VG_LAUNCH(MI_, NI_, VG_GATE2); break;                                                                \
    case VG_AFFINE: VG_LAUNCH(MI_, NI_, VG_AFFINE); break;                                                              \
    case VG_MUL2: VG_LAUNCH(MI_, NI_, VG_MUL2); break;                                                                  \
    default: VG_LAUNCH(MI_, NI_, VG_MUL1); break;                                                                       \
  }

// out (n_img, M, P) = epi(weight (M, K) . x (n_img, K, P)), all fp32, pixels contiguous.  epi (r = row, v* (M) vectors, s*
// maps of the output's shape; out1 only where named):
//   0 out0 = acc                         1 out0 = acc + v0[r]                  2 out0 = acc + v0[r], out1 = GELU(out0)
//   3 out0 = acc + v0[r], out1 = out0 s0 4 out0 = s0 v0[r] + acc v1[r] + v2[r] (+ s1 v3[r]; v0 NULL: 1)
//   5 out0 = acc s0, out1 = acc s1       6 out0 = acc s0
extern "C" int rsdet_van_gemm_f32(const float* weight, const float* x, int M, int K, int P, int n_img, int epi,
                                  const float* v0, const float* v1, const float* v2, const float* v3, const float* s0,
                                  const float* s1, float* out0, float* out1, void* stream) {
  VgTile t;
  if (!vg_pick(M, K, P, n_img, &t) || epi < 0 || epi > VG_MUL1) return RSDET_EINVAL;
  if (!weight || !x || !out0) return RSDET_EINVAL;
  const bool two = epi == VG_BIAS_GELU2 || epi == VG_GATE2 || epi == VG_MUL2;
  if (two && !out1) return RSDET_EINVAL;
  if ((epi == VG_GATE2 || epi == VG_AFFINE || epi == VG_MUL2 || epi == VG_MUL1) && !s0) return RSDET_EINVAL;
  if (epi == VG_MUL2 && !s1) return RSDET_EINVAL;
  if (epi == VG_AFFINE && (!v1 || !v2 || (s1 && !v3))) return RSDET_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)s0 | (uintptr_t)s1 | (uintptr_t)weight) & 15) || (P & 3))
    return RSDET_EINVAL;
  VgArgs a{weight, x, out0, out1, s0, s1, v0, v1, v2, v3, M, K, P, n_img};
  const int m_tiles = M / (32 * t.mi), p_tiles = P / (32 * t.ni);
  if ((long long)m_tiles * p_tiles * n_img > 0x7fffffffll) return RSDET_EINVAL;
  if (t.mi == 5 && t.ni == 2) {
    VG_EPI_SWITCH(5, 2)
  } else if (t.mi == 5) {
    VG_EPI_SWITCH(5, 1)
  } else if (t.mi == 4 && t.ni == 2) {
    VG_EPI_SWITCH(4, 2)
  } else if (t.mi == 4) {
    VG_EPI_SWITCH(4, 1)
  } else if (t.ni == 4) {
    VG_EPI_SWITCH(2, 4)
  } else {
    VG_EPI_SWITCH(2, 2)
  }
  return rsdet_launch_status();
}
#undef VG_EPI_SWITCH
#undef VG_LAUNCH

// ---- weight gradients and the small passes of the block node (ops/van_block.py) ----
static inline int vg_wgrad_mi(int M) { return M % 160 == 0 ? 5 : (M % 128 == 0 ? 4 : (M % 64 == 0 ? 2 : 0)); }

// 1 when rsdet_van_wgrad_f32 takes U (M, N) = sum_{img, p} g[img][m][p] x[img][n][p]: M a multiple of 160 / 128 / 64, N of 64,
// P of 32
extern "C" int rsdet_van_wgrad_f32_supported(int M, int N, int P, int n_img) {
  return (vg_wgrad_mi(M) && N >= 64 && N % 64 == 0 && P >= 32 && P % 32 == 0 && n_img >= 1) ? 1 : 0;
}
// number of split-K partials rsdet_van_wgrad_f32 leaves: <= 256 workgroups per launch, at least 4 chunks of 32 pixels each
extern "C" int rsdet_van_wgrad_f32_splits(int M, int N, int P, int n_img) {
  if (!rsdet_van_wgrad_f32_supported(M, N, P, n_img)) return 0;
  const int tiles = (M / (32 * vg_wgrad_mi(M))) * (N / 64);
  const long long Q = (long long)(P / 32) * n_img;
  // one workgroup per CU (three-slot ring, 84 KB of LDS): ONE round of workgroups.  (Two-slot rings with two workgroups per
  // CU and twice the splits, round 6: the weight-gradient kernels 7.06 -> 6.81 ms per Oriented R-CNN step, the folds that
  // sum twice the partials 1.91 -> 2.32 ms -- not taken.)
  long long S = 256 / tiles;
  if (S > Q / 4) S = Q / 4;
  if (S < 1) S = 1;
  return (int)S;
}
// g (n_img, M, P), x (n_img, N, P) -> partial (S, M, N) fp32, S = rsdet_van_wgrad_f32_splits; fold with rsdet_van_fold_rows_f32
// (or, for partials formed as (x, g): rsdet_van_fold_bn_f32)
extern "C" int rsdet_van_wgrad_f32(const float* g, const float* x, int M, int N, int P, int n_img, float* partial,
                                   void* stream) {
  if (!rsdet_van_wgrad_f32_supported(M, N, P, n_img)) return RSDET_EINVAL;
  if (!g || !x || !partial || (((uintptr_t)g | (uintptr_t)x | (uintptr_t)partial) & 15)) return RSDET_EINVAL;
  const int mi = vg_wgrad_mi(M), m_tiles = M / (32 * mi), n_tiles = N / 64;
  const int S = rsdet_van_wgrad_f32_splits(M, N, P, n_img);
  const dim3 grid((unsigned)(m_tiles * n_tiles * S));
  hipStream_t s = (hipStream_t)stream;
  // One workgroup per CU on three-slot rings.  Measured against it in round 6 (Oriented R-CNN step, same box, three runs
  // each): two-slot rings with twice the splits (kernels 7.06 -> 6.81 ms, the folds that sum twice the partials 1.91 -> 2.32
  // ms), and half-width tiles (M-tile x 32 columns) on two-slot rings at the SAME number of partials -- two workgroups per
  // CU like the GEMM -- 39.40 vs 39.23 tiles/s: the narrow tile's extra LDS reads per MFMA cost what the overlap returns.
  // RSDET_VG_WGRAD_NARROW=1 selects the latter.
  static const bool wide = [] { const char* e = getenv("RSDET_VG_WGRAD_NARROW"); return !(e && e[0] == '1'); }();
#define VG_WGRAD(MI_)                                                                                                     \
  do {                                                                                                                    \
    if (wide)                                                                                                             \
      hipLaunchKernelGGL((van_wgrad_f32_kernel<MI_, 2, 3>), grid, dim3(VG_NT), 0, s, g, x, M, N, P, n_img, m_tiles,       \
                         n_tiles, S, partial);                                                                            \
    else                                                                                                                  \
      hipLaunchKernelGGL((van_wgrad_f32_kernel<MI_, 1, 2>), dim3((unsigned)(m_tiles * 2 * n_tiles * S)), dim3(VG_NT), 0,  \
                         s, g, x, M, N, P, n_img, m_tiles, 2 * n_tiles, S, partial);                                      \
  } while (0)
  if (mi == 5) VG_WGRAD(5);
  else if (mi == 4) VG_WGRAD(4);
  else VG_WGRAD(2);
#undef VG_WGRAD
  return rsdet_launch_status();
}

static int vg_rows_fold_args(const rsdet_van_rows_fold* f, VgRowsFold* k);
extern "C" int rsdet_van_fold_rows_multi_f32(const rsdet_van_rows_fold* jobs, int n, void* stream) {
  if (!jobs || n < 1 || n > 3) return RSDET_EINVAL;
  VgRowsFold3 k;
  int mmax = 0;
  for (int i = 0; i < 3; ++i) {
    const int rc = vg_rows_fold_args(jobs + (i < n ? i : 0), &k.j[i]);
    if (rc) return rc;
    if (i >= n) k.j[i].M = 0;
    if (k.j[i].M > mmax) mmax = k.j[i].M;
  }
  hipLaunchKernelGGL(van_fold_rows_multi_kernel, dim3((unsigned)mmax, (unsigned)n), dim3(256), 0, (hipStream_t)stream, k);
  return rsdet_launch_status();
}

static int vg_rows_fold_args(const rsdet_van_rows_fold* f, VgRowsFold* k) {
  if (!f || f->S < 1 || f->M < 1 || f->N < 4 || (f->N & 3) || !f->partial || !f->grad_w) return RSDET_EINVAL;
  if ((f->grad_b || f->grad_rs) && f->gs_tab && (f->gs_ns < 1 || f->gs_stride < 1)) return RSDET_EINVAL;
  if (f->grad_rs && !f->w) return RSDET_EINVAL;
  if (f->r_tab && (f->r_ns < 1 || !f->sc || !f->sh)) return RSDET_EINVAL;
  *k = VgRowsFold{f->partial, f->row_scale, f->w, f->gs_tab, f->bias, f->r_tab, f->sc, f->sh, f->grad_w, f->grad_b, f->grad_rs,
                  f->S, f->M, f->N, f->gs_ns, f->gs_stride, f->r_ns};
  return RSDET_OK;
}
extern "C" int rsdet_van_fold_rows_f32(const rsdet_van_rows_fold* f, void* stream) {
  VgRowsFold k;
  const int rc = vg_rows_fold_args(f, &k);
  if (rc) return rc;
  hipLaunchKernelGGL(van_fold_rows_kernel, dim3((unsigned)f->M), dim3(256), 0, (hipStream_t)stream, k);
  return rsdet_launch_status();
}

extern "C" int rsdet_van_fold_bn_f32(const rsdet_van_bn_fold* f, void* stream) {
  if (!f || f->S < 1 || f->K < 1 || f->O < 1 || f->gs_ns < 1 || f->gs_stride < 1 || !(f->cnt > 0.f)) return RSDET_EINVAL;
  if (!f->partial || !f->wt || !f->gs_tab || !f->mean || !f->rstd || !f->sc || !f->sh || !f->grad_w || !f->grad_gamma ||
      !f->grad_beta || !f->v0 || !f->v1 || !f->v2 || !f->v3 || ((f->r_tab == nullptr) != (f->ls == nullptr)) ||
      (f->r_tab && f->r_ns < 1))
    return RSDET_EINVAL;
  VgBnFold k{f->partial, f->wt, f->gs_tab, f->r_tab, f->ls, f->mean, f->rstd, f->sc, f->sh, f->grad_w, f->grad_b,
             f->grad_gamma, f->grad_beta, f->v0, f->v1, f->v2, f->v3, f->S, f->K, f->O, f->gs_ns, f->gs_stride, f->r_ns,
             f->cnt};
  hipLaunchKernelGGL(van_fold_bn_kernel, dim3((unsigned)f->K), dim3(256), 0, (hipStream_t)stream, k);
  return rsdet_launch_status();
}

// slices per (image, channel) plane of the reductions below: planes of more than 8 192 pixels are cut
extern "C" int rsdet_van_chan_slices(int P) {
  int S = 1;
  while (P / S > 8192 && P % (S * 2 * 4) == 0) S *= 2;
  return S;
}
// mode 0: tab[(c * ns + j) * 2 + {0, 1}] = slice partials of (sum_p a, sum_p a b) (b NULL: 0), ns = N * rsdet_van_chan_slices(P);
// mode 1: the slice's (mean, sum of squared deviations from it) of a -- BatchNorm statistics for rsdet_van_bn_prep_f32.
// a, b: (N, C, P) fp32, P % 4 == 0.
extern "C" int rsdet_van_chan_reduce_f32(const float* a, const float* b, int N, int C, int P, int mode, float* tab,
                                         void* stream) {
  if (N < 1 || C < 1 || P < 4 || (P & 3) || !a || !tab || (mode != 0 && mode != 1) || ((uintptr_t)a & 15) ||
      ((uintptr_t)b & 15))
    return RSDET_EINVAL;
  if ((long long)N * C > 65535) return RSDET_EINVAL;
  const int S = rsdet_van_chan_slices(P);
  const dim3 grid((unsigned)S, (unsigned)(N * C));
  if (mode == 0)
    hipLaunchKernelGGL((van_chan_reduce_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, a, b, C, P, S, tab);
  else
    hipLaunchKernelGGL((van_chan_reduce_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a, b, C, P, S, tab);
  return rsdet_launch_status();
}

extern "C" int rsdet_van_bn_prep_f32(const rsdet_van_bn_prep* f, void* stream) {
  if (!f || f->O < 1 || f->K < 1 || f->ns < 1 || f->len < 1 || (long long)f->ns * f->len < 2) return RSDET_EINVAL;
  if (!f->tab || !f->gamma || !f->beta || !f->w || !f->w_out || !f->b_out || !f->mean || !f->rstd || !f->sc || !f->sh ||
      ((f->running_mean == nullptr) != (f->running_var == nullptr)))
    return RSDET_EINVAL;
  if (f->ls && (!f->e0 || !f->e2)) return RSDET_EINVAL;
  VgBnPrep k{f->tab, f->gamma, f->beta, f->w, f->b, f->w_out, f->b_out, f->mean, f->rstd, f->sc, f->sh, f->running_mean,
             f->running_var, (long long*)f->num_batches_tracked, f->ls, f->b2, f->e0, f->e2, f->shortcut, f->O, f->K, f->ns,
             f->len, f->eps, f->momentum};
  hipLaunchKernelGGL(van_bn_prep_kernel, dim3((unsigned)f->O), dim3(256), 0, (hipStream_t)stream, k);
  return rsdet_launch_status();
}

// n <= 5 transposes as one launch: dst[j] (K[j], O[j]) = transpose(src[j] (O[j], K[j])) with row o of src scaled by
// row_scale[j][o] (NULL: 1) -- the weight operands of the backward-data GEMMs of a block
extern "C" int rsdet_van_transposes_f32(int n, const float* const* src, const float* const* row_scale, float* const* dst,
                                        const int* O, const int* K, void* stream) {
  if (n < 0 || n > 5) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!src || !row_scale || !dst || !O || !K) return RSDET_EINVAL;
  VgTransposeJobs jobs;
  jobs.n = n;
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    if (!src[j] || !dst[j] || O[j] < 1 || K[j] < 1) return RSDET_EINVAL;
    jobs.src[j] = src[j], jobs.rs[j] = row_scale[j], jobs.dst[j] = dst[j], jobs.O[j] = O[j], jobs.K[j] = K[j];
    jobs.block0[j] = blocks;
    blocks += ((O[j] + 31) / 32) * ((K[j] + 31) / 32);
  }
  jobs.block0[n] = blocks;
  hipLaunchKernelGGL(van_transposes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs);
  return rsdet_launch_status();
}
