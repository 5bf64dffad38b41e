// alignconv_mfma.hip -- AlignConv (3x3 deformable convolution, one deformable group) as an IMPLICIT GEMM on the
// bf16 matrix cores of gfx950: the column matrix never exists in HBM.
//
// Replaces: DeformConvFunction.forward of /root/reference/python/jdet/ops/dcn_v1.py:412-454 (deformable_im2col
//   :309-339 with the sampling kernel :132-184 and the bilinear rule :25-56, then the product :448-452) for the
//   geometry AlignConv uses (models/roi_heads/s2anet_head.py:603-660: 3x3, stride 1, padding 1, dg = 1), in the
//   arithmetic of a bf16 autocast step: samples interpolated in fp32 from bf16 activations, rounded once to bf16,
//   products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
//   out[p, o] = sum_{t < 9, c < C}  S(p, t, c) * W[o, t*C + c]        p = (b, ho, wo): GEMM  M = B*Ho*Wo, N = O, K = 9*C
//
// One workgroup owns 128 positions (an 8 x 16 pixel tile of one image) x 256 output channels, so every sample is
// interpolated exactly once.  K runs tap-major in steps of 64 channels of one tap; a step's A tile (128 x 64 samples)
// is PRODUCED into LDS by half of the workgroup's waves: a thread keeps the corner offsets and (lh, lw) of its
// positions for the current tap in registers, loads 16 B per corner (channels-last: the 64 channels of a corner are
// one 128-B line, 8 lanes per line), interpolates, rounds, and writes 16 B into the swizzled tile; the corner loads of
// two steps are in flight.  The other half CONSUMES: 32 x 32 MFMA tiles, and the B tiles (256 x 64 weights, 32 KB,
// L2-resident: every workgroup re-reads the same 1.2 MB) by LDS-DMA with the swizzle applied on the SOURCE address,
// two steps ahead of their use in a 3-deep ring.  A tiles are double-buffered; one barrier per step.  Workgroup ids
// are banded per XCD so that neighbouring position tiles share an L2.
//
// The same kernel in exact fp32 (T = float): 32 channels per step (the same 128-B rows), samples interpolated in the
// reference's operation order without contraction (columns bit-identical to rsdet_deform_im2col_f32), products on
// v_mfma_f32_32x32x2_f32 -- 1/16 of the bf16 rate, so this form is bound by the matrix cores (77 GFLOP at level 0 of a
// 4-tile batch = 0.49 ms at the 157 TFLOP/s peak) and the gather hides under them.
//
// LDS image (both tiles): row = position (A) / output channel (B), 128 B = 8 chunks of 8 bf16; chunk c of row r sits
// in slot c ^ ((r >> 1) & 7): 16 consecutive rows of one chunk cover all 64 banks once (ds_read_b128 fragments), and a
// producer / DMA piece of 8 rows x 8 slots is 1 KiB of consecutive LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_bf16.h"

namespace rsdet {

typedef __attribute__((ext_vector_type(8))) __bf16 acm_bf16x8;
typedef __attribute__((ext_vector_type(16))) float acm_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned acm_u32x4;
typedef __attribute__((ext_vector_type(4))) float acm_f32x4;

constexpr int ACM_BM = 128, ACM_BN = 256;          // K step: 128 B of channels = 64 bf16 / 32 fp32
constexpr int ACM_TY = 8, ACM_TX = 16;            // the 128 positions of a workgroup: 8 x 16 output pixels
constexpr int ACM_A_BYTES = ACM_BM * 128;          // 16 KB
constexpr int ACM_B_BYTES = ACM_BN * 128;          // 32 KB
constexpr int ACM_B_STAGES = 3;                    // weight tiles travel two steps ahead of their use
constexpr int ACM_B_OFF = 2 * ACM_A_BYTES;         // LDS: A0 A1 | B0 B1 B2  (128 KB)
constexpr int ACM_LDS_BYTES = ACM_B_OFF + ACM_B_STAGES * ACM_B_BYTES;
// wave split (template parameters of the kernel): ACM_CW consumer waves = 2 (rows) x ACM_CW/2 (columns) wave tiles of
// 64 x (512 / ACM_CW), ACM_PW producer waves.  Measured in one box at level 0: bf16 4+4 waves 169 us, 8+8 waves 150 us
// (a second wave of each kind per SIMD hides the LDS / global latencies of the first); fp32 848 vs 867 us (bound by the
// matrix cores: the smaller wave tiles only add fragment reads)

constexpr int ACM_ZERO_BYTES = 4096;  // C <= 2048 (bf16) / 1024 (fp32)
__device__ const uint4 acm_zero_line[ACM_ZERO_BYTES / 16] = {};  // what the corners outside the map point at

#ifdef ACM_TRACE  // debug builds only (profiles/scripts/trace_alignconv.py): per-step timestamps of the first tile
__device__ unsigned long long* g_acm_trace;
#define ACM_STAMP(slot, step, k)                                                                         \
  do {                                                                                                   \
    if (acm_trace_wg && (threadIdx.x & 63) == 0 && g_acm_trace && (step) < 80)                         \
      g_acm_trace[((slot) * 80 + (step)) * 4 + (k)] = __builtin_amdgcn_s_memtime();                      \
  } while (0)
#else
#define ACM_STAMP(slot, step, k)
#endif

struct AcmGeom {
  int C, H, W, B, Ho, Wo, O;
  int ph, pw;
};

__device__ __forceinline__ int acm_slot(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// corner element offsets (from the image's first pixel; -1 = outside the map) and weights of one (position, tap):
// dcn_v1.py:25-56 and the window test of :170.  h, w: the sampling point (base + kernel + learned offset)
__device__ __forceinline__ void acm_foot(const AcmGeom& g, float h, float w, bool valid, int ibase, int (&off)[4],
                                         float (&frac)[2]) {
  const bool in = valid && (h > -1 && w > -1 && h < g.H && w < g.W);
  const float fh = floorf(h), fw = floorf(w);
  const int hl = (int)fh, wl = (int)fw;
  const int hh = hl + 1, wh = wl + 1;
  const bool t = in && hl >= 0, bt = in && hh <= g.H - 1, l = wl >= 0, r = wh <= g.W - 1;
  const int row0 = ibase + hl * g.W * g.C, row1 = row0 + g.W * g.C;
  off[0] = (t && l) ? row0 + wl * g.C : -1;
  off[1] = (t && r) ? row0 + wh * g.C : -1;
  off[2] = (bt && l) ? row1 + wl * g.C : -1;
  off[3] = (bt && r) ? row1 + wh * g.C : -1;
  // (lh, lw): the four weights are formed where they are used; a corner outside the map contributes 0 because it is
  // read from a line of zeros, whatever its weight
  frac[0] = h - fh;
  frac[1] = w - fw;
}

__device__ __forceinline__ void acm_weights(float lh, float lw, float (&wt)[4]) {  // dcn_v1.py:44-47
  const float uh = 1 - lh, uw = 1 - lw;
  wt[0] = uh * uw; wt[1] = uh * lw; wt[2] = lh * uw; wt[3] = lh * lw;
}

// The corner loads of a step stay in flight across a barrier and a loop back-edge; the compiler's wait-count pass
// answers that with vmcnt(0) at every use.  So they are issued from inline asm (the compiler does not track them) and
// waited for by hand: loads complete in issue order, so vmcnt(N) with N <= the number of vector-memory operations
// issued after a load guarantees it has landed.  acm_landed() then ties the registers to the wait (later uses cannot be
// scheduled above it).
__device__ __forceinline__ void acm_gload(acm_u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void acm_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void acm_landed(acm_u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void acm_gload(float& dst, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void acm_landed(float& v) { asm volatile("" : "+v"(v)); }

typedef __attribute__((ext_vector_type(2))) float acm_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 acm_bf16x2;

// 8 samples of one (position, tap): w1*v1 + w2*v2 + w3*v3 + w4*v4 per channel, rounded to bf16 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ acm_u32x4 acm_blend(const acm_u32x4 (&v)[4], const float (&w)[4]) {
  acm_u32x4 r;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t u = v[j][d];
      lo = __builtin_fmaf(w[j], __uint_as_float(u << 16), lo);
      hi = __builtin_fmaf(w[j], __uint_as_float(u & 0xffff0000u), hi);
    }
    const acm_f32x2 pr = {lo, hi};
    r[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, acm_bf16x2));
  }
  return r;
}

// fp32 form: 4 samples, the reference's order w1*v1 + w2*v2 + w3*v3 + w4*v4 (this file is compiled without contraction)
__device__ __forceinline__ acm_u32x4 acm_blend_f32(const acm_u32x4 (&v)[4], const float (&w)[4]) {
  acm_u32x4 r;
#pragma unroll
  for (int d = 0; d < 4; ++d)
    r[d] = __float_as_uint(w[0] * __uint_as_float(v[0][d]) + w[1] * __uint_as_float(v[1][d]) +
                           w[2] * __uint_as_float(v[2][d]) + w[3] * __uint_as_float(v[3][d]));
  return r;
}

// grid: rsdet_xcd_band_grid(position tiles, output-channel tiles); block 64 * (ACM_CW + ACM_PW); 128 KB of LDS.
// Waves [0, ACM_CW) CONSUME (MFMA tiles + the LDS-DMA of the weight tiles), waves [ACM_CW, ACM_CW + ACM_PW) PRODUCE the
// next A tile; one barrier per K step keeps the two halves in lockstep, so a step costs max(gather + interpolation,
// MFMA) instead of their sum.  A producer thread fetches the raw offsets of a tap one tap ahead and keeps the corner
// loads of TWO steps in flight.
// out: OUT_NHWC ? (B*Ho*Wo, O) : (B, O, Ho*Wo), bf16.  colT (optional): (B*Ho*Wo, 9*C) bf16, the A tiles as produced.
template <typename T, bool OUT_NHWC, int ACM_CW, int ACM_PW>
__global__ __launch_bounds__(64 * (ACM_CW + ACM_PW), 1) void alignconv_fwd_mfma_kernel(
    const T* __restrict__ im, const float* __restrict__ offset, const T* __restrict__ wt, AcmGeom g,
    int m_tiles, int n_tiles, T* __restrict__ out, T* __restrict__ colT) {
  constexpr int ACM_WN = ACM_CW / 2;                 // consumer waves along N
  constexpr int ACM_NI = ACM_BN / ACM_WN / 32;       // 32-wide MFMA tiles per wave along N (4 or 2)
  constexpr int ACM_PT = 64 * ACM_PW;                // producer threads
  constexpr int ACM_TASKS = ACM_BM * 8 / ACM_PT;     // (position, 16-byte chunk) tasks per producer thread and step
  constexpr int ACM_LOADS = ACM_TASKS * 4;           // corner loads a producer thread issues per step
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int BK = 128 / (int)sizeof(T);   // channels per K step
  constexpr int EPT = 16 / (int)sizeof(T);   // channels per 16-byte piece
  __shared__ __attribute__((aligned(1024))) unsigned char acm_lds[ACM_LDS_BYTES];
  const RsdetBandItem item = rsdet_xcd_band(blockIdx.x, m_tiles, n_tiles);
  if (!item.valid) return;
#ifdef ACM_TRACE
  const bool acm_trace_wg = item.outer == 0 && item.inner == 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long plane = (long long)g.Ho * g.Wo;
  // position tile: ACM_TY x ACM_TX output pixels of one image (neighbours in both directions share corner pixels, so
  // a step touches ~(TY+1)(TX+1) distinct lines instead of 2 x 129 for a row of 128)
  const int tiles_x = (g.Wo + ACM_TX - 1) / ACM_TX, tiles_y = (g.Ho + ACM_TY - 1) / ACM_TY;
  const int tb = item.outer / (tiles_x * tiles_y), tr = item.outer - tb * (tiles_x * tiles_y);
  const int ty0 = (tr / tiles_x) * ACM_TY, tx0 = (tr - (tr / tiles_x) * tiles_x) * ACM_TX;
  // tile-local index p -> (ty0 + p / TX, tx0 + p % TX)
  const int n_base = item.inner * ACM_BN;
  const int K = 9 * g.C;
  const int cchunks = g.C / BK;
  const int steps = 9 * cchunks;

  if (wave >= ACM_CW) {
    // ------------------------------------------------------------------ producers
    const int ptid = tid - 64 * ACM_CW;
    const int q = ptid & 7, prow = ptid >> 3;  // tile positions i*32 + prow, channel chunk q of the step's 64
    int cbase[ACM_TASKS];        // first column element of the position in colT (B*Ho*Wo*9*C < 2^31 is checked)
    float hb[ACM_TASKS], wb[ACM_TASKS];
    int ibase[ACM_TASKS], obase[ACM_TASKS];  // element offsets of the image / of the position inside the offset tensor
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < ACM_TASKS; ++i) {
      const int pl = i * (ACM_PT / 8) + prow;
      const int ho = ty0 + pl / ACM_TX, wo = tx0 + pl % ACM_TX;
      const bool ok = ho < g.Ho && wo < g.Wo;
      vmask |= ok ? (1u << i) : 0u;
      const int b = tb;
      const int hw = ok ? ho * g.Wo + wo : 0;
      cbase[i] = (int)(((long long)b * plane + hw) * K);
      hb[i] = (float)(ho - g.ph);
      wb[i] = (float)(wo - g.pw);
      ibase[i] = b * g.H * g.W * g.C;
      obase[i] = (int)((long long)b * 18 * plane + hw);
    }
    const T* zero = reinterpret_cast<const T*>(acm_zero_line);
    int foff[ACM_TASKS][4];          // corners of the tap whose loads are issued next (element offsets, -1 outside)
    float wa[ACM_TASKS][2], wb2[ACM_TASKS][2];  // (lh, lw) of the even / odd taps
    float onext[ACM_TASKS][2];       // raw offsets of the following tap, fetched a tap ahead
    acm_u32x4 pre0[ACM_TASKS][4], pre1[ACM_TASKS][4];  // corner loads of the even / odd steps in flight

    auto fetch_offsets = [&](int tap) {
#pragma unroll
      for (int i = 0; i < ACM_TASKS; ++i) {
        acm_gload(onext[i][0], offset + obase[i] + (long long)(2 * tap) * plane);
        acm_gload(onext[i][1], offset + obase[i] + (long long)(2 * tap + 1) * plane);
      }
    };
    auto feet = [&](int tap) {  // from onext (>= 16 loads were issued after them, except for tap 0); then start
      const int ki = tap / 3, kj = tap - ki * 3;  // fetching the following tap's offsets
      if (tap == 0) acm_wait_vm<0>(); else acm_wait_vm<ACM_LOADS>();
#pragma unroll
      for (int i = 0; i < ACM_TASKS; ++i) { acm_landed(onext[i][0]); acm_landed(onext[i][1]); }
#pragma unroll
      for (int i = 0; i < ACM_TASKS; ++i) {
        const float h = hb[i] + (float)ki + onext[i][0], w = wb[i] + (float)kj + onext[i][1];
        float wt[2];
        acm_foot(g, h, w, (vmask >> i) & 1u, ibase[i], foff[i], wt);
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // value selects: a pointer select would put both sets in scratch
          wa[i][j] = (tap & 1) ? wa[i][j] : wt[j];
          wb2[i][j] = (tap & 1) ? wt[j] : wb2[i][j];
        }
      }
      if (tap + 1 < 9) fetch_offsets(tap + 1);
    };
    auto issue = [&](int step, acm_u32x4 (&dst)[ACM_TASKS][4]) {
      const int tap = step / cchunks, cc = step - tap * cchunks;
      if (cc == 0) feet(tap);
      const int coff = cc * BK + q * EPT;
#pragma unroll
      for (int i = 0; i < ACM_TASKS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const T* src = foff[i][j] >= 0 ? im + foff[i][j] : zero;
          acm_gload(dst[i][j], src + coff);
        }
    };
    auto landed = [&](acm_u32x4 (&v)[ACM_TASKS][4]) {
#pragma unroll
      for (int i = 0; i < ACM_TASKS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acm_landed(v[i][j]);
    };
    auto produce = [&](int step, const acm_u32x4 (&src)[ACM_TASKS][4]) {
      unsigned char* stage = acm_lds + (step & 1) * ACM_A_BYTES;
      if (wave == ACM_CW) ACM_STAMP(1, step, 0);
      const int tap = step / cchunks, cc = step - tap * cchunks;
      const int k0 = tap * g.C + cc * BK;
#pragma unroll
      for (int i = 0; i < ACM_TASKS; ++i) {
        float wt[4];
        acm_weights((tap & 1) ? wb2[i][0] : wa[i][0], (tap & 1) ? wb2[i][1] : wa[i][1], wt);
        const acm_u32x4 r = F32 ? acm_blend_f32(src[i], wt) : acm_blend(src[i], wt);
        const int row = i * (ACM_PT / 8) + prow;
        *reinterpret_cast<acm_u32x4*>(stage + row * 128 + acm_slot(row, q) * 16) = r;
        if (colT != nullptr && ((vmask >> i) & 1u))
          *reinterpret_cast<acm_u32x4*>(colT + cbase[i] + k0 + q * EPT) = r;
      }
      if (wave == ACM_CW) ACM_STAMP(1, step, 1);
    };
    // step u travels in the register set u & 1; the weights of a tap live in the set tap & 1 (the steps in flight
    // span at most two consecutive taps).  At the top of iteration s: stage s & 1 holds step s (its register set is
    // free), the loads of step s+1 are in flight.  The iteration issues step s+2 (16 loads), waits for all but those,
    // and produces step s+1.
    fetch_offsets(0);
    issue(0, pre0);
    if (steps > 1) { issue(1, pre1); acm_wait_vm<ACM_LOADS>(); } else { acm_wait_vm<0>(); }
    landed(pre0);
    produce(0, pre0);
    __syncthreads();
    for (int s = 0; s < steps; s += 2) {
      if (s + 2 < steps) { issue(s + 2, pre0); acm_wait_vm<ACM_LOADS>(); } else { acm_wait_vm<0>(); }
      if (s + 1 < steps) { landed(pre1); produce(s + 1, pre1); }
      __syncthreads();
      if (s + 1 < steps) {
        if (s + 3 < steps) { issue(s + 3, pre1); acm_wait_vm<ACM_LOADS>(); } else { acm_wait_vm<0>(); }
        if (s + 2 < steps) { landed(pre0); produce(s + 2, pre0); }
        __syncthreads();
      }
    }
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int wm = wave / ACM_WN, wn = wave % ACM_WN;  // 2 x ACM_WN waves: 64 positions x 32*ACM_NI output channels each
  acm_f32x16 acc[2][ACM_NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < ACM_NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  auto issue_b = [&](int step) {
    const int t1 = step / cchunks, c1 = step - t1 * cchunks;
    const int k0 = t1 * g.C + c1 * BK;
    unsigned char* stage = acm_lds + ACM_B_OFF + (step % ACM_B_STAGES) * ACM_B_BYTES;
    // 32 pieces of 8 rows x 128 B; wave w moves pieces w*8 .. w*8+7
#pragma unroll
    for (int it = 0; it < 32 / ACM_CW; ++it) {
      const int piece = wave * (32 / ACM_CW) + it;
      const int row = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const int o = min(n_base + row, g.O - 1);
      const T* src = wt + (long long)o * K + k0 + chunk * EPT;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + piece * 1024),
                                       16, 0, 0);
    }
  };
  constexpr int B_OPS = 32 / ACM_CW;            // LDS-DMA operations of one tile per consumer wave
  issue_b(0);
  if (steps > 1) issue_b(1);
  if (steps > 1) acm_wait_vm<B_OPS>(); else acm_wait_vm<0>();   // tile 0 landed, tile 1 may still be in flight
  __syncthreads();

  for (int s = 0; s < steps; ++s) {
    const unsigned char* cur = acm_lds + (s & 1) * ACM_A_BYTES;
    const unsigned char* curb = acm_lds + ACM_B_OFF + (s % ACM_B_STAGES) * ACM_B_BYTES;
    if (wave == 0) ACM_STAMP(0, s, 0);
    // tile s+2 goes where tile s-1 was: every consumer wave finished reading that one before the last barrier
    if (s + 2 < steps) issue_b(s + 2);
    if constexpr (!F32) {
      // 4 k-steps of 16 on the current stage: lane half h holds k = 8h .. 8h+7 of a step, i.e. chunk 2*ks + h.
      // Fragments of two k-steps in registers: the reads of step ks+1 are issued before the MFMAs of step ks
      acm_bf16x8 a[2][2], b[2][ACM_NI];
      auto frags = [&](int ks, acm_bf16x8 (&fa)[2], acm_bf16x8 (&fb)[ACM_NI]) {
        const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row = wm * 64 + mi * 32 + (lane & 31);
          fa[mi] = *reinterpret_cast<const acm_bf16x8*>(cur + row * 128 + acm_slot(row, chunk) * 16);
        }
#pragma unroll
        for (int ni = 0; ni < ACM_NI; ++ni) {
          const int row = wn * (32 * ACM_NI) + ni * 32 + (lane & 31);
          fb[ni] = *reinterpret_cast<const acm_bf16x8*>(curb + row * 128 + acm_slot(row, chunk) * 16);
        }
      };
      frags(0, a[0], b[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < ACM_NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][mi], b[ks & 1][ni], acc[mi][ni], 0, 0, 0);
      }
    } else {
      // 16 k-steps of 2: a 16-byte chunk holds k = 4c .. 4c+3 of a row; lane half h takes k = 4c + 2*kk + h
      const int h = lane >> 5;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        acm_f32x4 a[2], b[ACM_NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row = wm * 64 + mi * 32 + (lane & 31);
          a[mi] = *reinterpret_cast<const acm_f32x4*>(cur + row * 128 + acm_slot(row, c) * 16);
        }
#pragma unroll
        for (int ni = 0; ni < ACM_NI; ++ni) {
          const int row = wn * (32 * ACM_NI) + ni * 32 + (lane & 31);
          b[ni] = *reinterpret_cast<const acm_f32x4*>(curb + row * 128 + acm_slot(row, c) * 16);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float av[2], bv[ACM_NI];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) av[mi] = h ? a[mi][2 * kk + 1] : a[mi][2 * kk];
#pragma unroll
          for (int ni = 0; ni < ACM_NI; ++ni) bv[ni] = h ? b[ni][2 * kk + 1] : b[ni][2 * kk];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < ACM_NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
      }
    }
    if (wave == 0) ACM_STAMP(0, s, 1);
    // tile s+1 must have landed before the barrier; tile s+2 (the B_OPS newest operations) stays in flight
    if (s + 2 < steps) acm_wait_vm<B_OPS>(); else acm_wait_vm<0>();
    if (wave == 0) ACM_STAMP(0, s, 2);
    __syncthreads();
    if (wave == 0) ACM_STAMP(0, s, 3);
  }

  // epilogue: D[row = position][col = output channel]; col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < ACM_NI; ++ni) {
      const int o = n_base + wn * (32 * ACM_NI) + ni * 32 + (lane & 31);
      if (o >= g.O) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pl = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int ho = ty0 + pl / ACM_TX, wo = tx0 + pl % ACM_TX;
        if (ho >= g.Ho || wo >= g.Wo) continue;
        const long long hw = (long long)ho * g.Wo + wo;
        T v;
        if constexpr (F32) v = acc[mi][ni][e]; else v = f2bf(acc[mi][ni][e]);
        if (OUT_NHWC) out[((long long)tb * plane + hw) * g.O + o] = v;
        else out[((long long)tb * g.O + o) * plane + hw] = v;
      }
    }
}

}  // namespace rsdet

using namespace rsdet;

#ifdef ACM_TRACE
extern "C" void rsdet_debug_set_acm_trace(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_acm_trace), &p, sizeof(p)); }
#endif

static int acm_supported(const rsdet_dcn_geom* s, int O, int elem_bytes) {
  if (!s) return 0;
  if (s->kh != 3 || s->kw != 3 || s->sh != 1 || s->sw != 1 || s->dh != 1 || s->dw != 1 || s->dg != 1) return 0;
  const int bk = 128 / elem_bytes;
  if (s->C < bk || s->C % bk || s->C * elem_bytes > ACM_ZERO_BYTES || O < 32 || O % 32) return 0;
  if (s->B < 1 || s->H < 1 || s->W < 1) return 0;
  if ((long long)s->B * s->H * s->W * s->C >= (1ll << 31)) return 0;
  if ((long long)s->B * (s->H + 2 * s->ph - 2) * (s->W + 2 * s->pw - 2) * 9 * s->C >= (1ll << 31)) return 0;
  return 1;
}

extern "C" int rsdet_alignconv_mfma_supported(const rsdet_dcn_geom* s, int O) { return acm_supported(s, O, 2); }
extern "C" int rsdet_alignconv_mfma_f32_supported(const rsdet_dcn_geom* s, int O) { return acm_supported(s, O, 4); }

template <typename T>
static int acm_launch(const T* im_nhwc, const float* offset, const T* weight, const rsdet_dcn_geom* geom, int O,
                      int out_nhwc, T* out, T* colT, void* stream) {
  if (!acm_supported(geom, O, (int)sizeof(T))) return RSDET_EINVAL;
  if (!im_nhwc || !offset || !weight || !out) return RSDET_EINVAL;
  AcmGeom g{geom->C, geom->H, geom->W, geom->B, 0, 0, O, geom->ph, geom->pw};
  g.Ho = geom->H + 2 * geom->ph - 2;
  g.Wo = geom->W + 2 * geom->pw - 2;
  if (g.Ho < 1 || g.Wo < 1) return RSDET_EINVAL;
  const int m_tiles = g.B * ((g.Ho + ACM_TY - 1) / ACM_TY) * ((g.Wo + ACM_TX - 1) / ACM_TX);
  const int n_tiles = (O + ACM_BN - 1) / ACM_BN;
  const dim3 grid((unsigned)rsdet_xcd_band_grid(m_tiles, n_tiles));
  hipStream_t st = (hipStream_t)stream;
  constexpr int CW = sizeof(T) == 4 ? 4 : 8, PW = CW;   // see the note on the wave split above
  if (out_nhwc)
    hipLaunchKernelGGL((alignconv_fwd_mfma_kernel<T, true, CW, PW>), grid, dim3(64 * (CW + PW)), 0, st, im_nhwc, offset,
                       weight, g, m_tiles, n_tiles, out, colT);
  else
    hipLaunchKernelGGL((alignconv_fwd_mfma_kernel<T, false, CW, PW>), grid, dim3(64 * (CW + PW)), 0, st, im_nhwc, offset,
                       weight, g, m_tiles, n_tiles, out, colT);
  return rsdet_launch_status();
}

extern "C" int rsdet_alignconv_fwd_mfma_bf16(const uint16_t* im_nhwc, const float* offset, const uint16_t* weight,
                                             const rsdet_dcn_geom* geom, int O, int out_nhwc, uint16_t* out,
                                             uint16_t* colT, void* stream) {
  return acm_launch<bf16_t>(im_nhwc, offset, weight, geom, O, out_nhwc, out, colT, stream);
}

extern "C" int rsdet_alignconv_fwd_mfma_f32(const float* im_nhwc, const float* offset, const float* weight,
                                            const rsdet_dcn_geom* geom, int O, int out_nhwc, float* out, float* colT,
                                            void* stream) {
  return acm_launch<float>(im_nhwc, offset, weight, geom, O, out_nhwc, out, colT, stream);
}
