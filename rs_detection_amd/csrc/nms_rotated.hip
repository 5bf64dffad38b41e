// nms_rotated.hip -- greedy rotated / class-aware NMS for gfx950, fully on device.
//
// Replaces: jdet.ops.nms_rotated.{nms_rotated_cpu, nms_rotated_cuda}
//   /root/reference/python/jdet/ops/nms_rotated.py:495-512; CUDA mask kernel :353-411;
//   host bit-mask sweep on managed memory after cudaDeviceSynchronize :450-493;
//   CPU greedy loop :414-449 (the parity target: `ovr >= thr`, :444).
//
// Three launches on one stream, no host round trip:
//   1. nms_prepare : gather dets by `order`, hoist fp64 sincos -> BoxPre (+label)
//   2. nms_mask    : upper-triangular 64x64 tiles.  Cheap pass (label gate +
//                    bounding circles) fills an LDS queue via wave ballot /
//                    popcount prefix; the queue is drained densely through the
//                    exact clipper; hits set bits with LDS 64-bit atomicOr.  Output is
//                    SPARSE: the non-zero (row, column block) words of a tile are appended
//                    to the row block's entry list (one returning atomic per tile) -- a
//                    64-box block suppresses into a handful of column blocks, and the sweep,
//                    which runs on ONE CU, is bound by what it has to pull through that
//                    CU's memory pipe (dense rows: 64 KB per block step, measured 1.4 us).
//                    The diagonal tile goes out transposed (diag_t) for the sweep's fixpoint.
//   3. nms_sweep   : one workgroup walks the 64-box blocks in score order.  Wave 0
//                    resolves the diagonal tile as a fixpoint over wave ballots; all waves
//                    OR the entries of the kept rows (prefetched two blocks ahead) into
//                    the LDS `removed` bitmap.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "rsdet_api_internal.h"
#include "rsdet_geom_fast.h"

namespace rsdet {

struct NmsBox {
  BoxPre p;
  float label;
  float pad;
};  // 48 B

constexpr int NMS_NT = 256;
constexpr int NMS_MAX_SEGS = 4096;  // label runs the sweep can take apart; beyond that it falls back to one sweep

struct NmsEntry {           // one non-zero 64-bit word of the suppression matrix
  unsigned long long bits;  // bit j: the row's box suppresses box 64*cblock + j
  int cblock;
  int row;                  // row inside its 64-box block
};  // 16 B

// 256 threads = four 64-box blocks.  Besides the gather + fp64 sincos, every wave reduces the label range of its
// block: nms_mask skips a tile whose row and column blocks cannot share a label -- with a label-major order
// (ops/nms_rotated.py builds one for the class-aware entry points) that is ~14 of 15 tiles of a 15-class set.
__global__ __launch_bounds__(256) void nms_prepare_kernel(const float* __restrict__ dets, int n, int box_len,
                                                          const int* __restrict__ order,
                                                          NmsBox* __restrict__ sorted, unsigned* __restrict__ blk_cnt,
                                                          float2* __restrict__ blk_label, int label_major,
                                                          int* __restrict__ segs) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < (n + 63) / 64) blk_cnt[p] = 0u;
  float lo = INFINITY, hi = -INFINITY;
  if (p < n) {
    const float* b = dets + (long long)order[p] * box_len;
    // segment table for the sweep: with a label-major order every run of equal labels is an independent NMS
    bool starts = p == 0;
    if (label_major && box_len == 6 && p > 0) starts = !(b[5] == dets[(long long)order[p - 1] * box_len + 5]);
    if (starts) {
      const int idx = atomicAdd(segs, 1);  // segs[0] was zeroed by the launcher
      if (idx < NMS_MAX_SEGS) segs[1 + idx] = p;
    }
    NmsBox o;
    o.p = prepare_box(b);
    o.label = box_len == 6 ? b[5] : 0.f;
    o.pad = 0.f;
    sorted[p] = o;
    lo = hi = o.label;
    if (!(o.label == o.label)) {  // a NaN label never equals anything: keep the block's range open
      lo = -INFINITY;
      hi = INFINITY;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
  if ((threadIdx.x & 63) == 0 && (p >> 6) < (n + 63) / 64) blk_label[p >> 6] = make_float2(lo, hi);
}

// One wave: lane = row of tile (rb, cbk); append the non-zero words to row block rb's entry list.
__device__ __forceinline__ void nms_emit_entries(unsigned long long word, int lane, int rb, int cbk, int col_blocks,
                                                 NmsEntry* __restrict__ entries, unsigned* __restrict__ blk_cnt) {
  const unsigned long long nz = __ballot(word != 0ull);
  if (nz == 0ull) return;
  unsigned base = 0u;
  if (lane == 0) base = atomicAdd(blk_cnt + rb, (unsigned)__popcll(nz));
  base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
  if (word != 0ull) {
    NmsEntry e;
    e.bits = word;
    e.cblock = cbk;
    e.row = lane;
    // capacity of a row block's list: 64 rows x col_blocks words (the dense size), never exceeded
    entries[(size_t)rb * 64 * col_blocks + base + __popcll(nz & ((1ull << lane) - 1ull))] = e;
  }
}

// TWO_TIER (round 3, the default): the suppression bit of a pair is a DECISION (IoU against one threshold), so the
// Green-integral IoU of rsdet_geom_fast.h (one lane per pair) settles every pair whose value is more than the budget away
// from the threshold; the reference-order clipper only sees the pairs inside the budget, the pairs in the reference's
// fragile zone (where its own value may be anything) and NaN boxes -- ~0.3 % of the pairs that reach this stage.  The bits,
// hence `keep`, are those of the all-exact form.
template <bool GE, bool TWO_TIER>
__global__ __launch_bounds__(NMS_NT) void nms_mask_kernel(const NmsBox* __restrict__ sorted, int n,
                                                          float thr, int col_blocks,
                                                          NmsEntry* __restrict__ entries,
                                                          unsigned* __restrict__ blk_cnt,
                                                          unsigned long long* __restrict__ diag_t,
                                                          const float2* __restrict__ blk_label, int no_cull) {
  // no_cull: ">= thr" with thr <= 0 -- the reference's CPU loop (nms_rotated.py:443-444) then suppresses on an IoU of
  // exactly 0 too, i.e. disjoint pairs count: the disjointness filters below must not drop them.
  const int rb = blockIdx.y, cbk = blockIdx.x;
  if (cbk < rb) return;  // lower triangle never read by the sweep
  if (cbk != rb) {       // label ranges apart: no pair of this tile passes the label gate (diagonal tiles always run:
    const float2 lr = blk_label[rb], lc = blk_label[cbk];  // they own diag_t)
    if (lr.x > lc.y || lc.x > lr.y) return;
  }
  __shared__ F2 s_pts[kQuadSlots * (NMS_NT / 4)];
  __shared__ NmsBox s_row[64];
  __shared__ NmsBox s_col[64];
  __shared__ unsigned long long s_mask[64];
  __shared__ unsigned short s_queue[64 * 64];
  __shared__ int s_count, s_count2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = min(64, n - rb * 64), cols = min(64, n - cbk * 64);
  if (tid < 64) {
    if (tid < rows) s_row[tid] = sorted[rb * 64 + tid];
    s_mask[tid] = 0ull;
  } else if (tid < 128) {
    int c = tid - 64;
    if (c < cols) s_col[c] = sorted[cbk * 64 + c];
  }
  if (tid == 0) {
    s_count = 0;
    s_count2 = 0;
  }
  __syncthreads();

  // pass A: wave w covers rows w, w+4, ...; lane = column.  Label gate + bounding circles only: a
  // handful of instructions per pair and no divergence (the separating-axis test used to sit here
  // and was paid by the whole wave whenever one lane reached it).
  for (int i = wave; i < rows; i += 4) {
    bool cand = false;
    if (lane < cols) {
      bool later = (cbk > rb) || (lane > i);
      cand = later && s_row[i].label == s_col[lane].label && (no_cull || !surely_disjoint(s_row[i].p, s_col[lane].p));
    }
    unsigned long long m = __ballot(cand);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(m));
      base = __shfl(base, 0);
      if (cand) s_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((i << 6) | lane);
    }
  }
  __syncthreads();
  // pass B: separating axes on the compacted list (dense lanes), compacted in place: by the barrier
  // every entry below q0 + NMS_NT has been read, and at most that many were kept.
  const int n_cand = s_count;
  for (int q0 = 0; q0 < n_cand; q0 += NMS_NT) {
    const int q = q0 + tid;
    unsigned e = 0;
    if (q < n_cand) e = s_queue[q];
    __syncthreads();
    const bool keep = q < n_cand && (no_cull || !sat_disjoint<0>(s_row[e >> 6].p, s_col[e & 63].p));
    unsigned long long m = __ballot(keep);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count2, __popcll(m));
      base = __shfl(base, 0);
      if (keep) s_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)e;
    }
  }
  __syncthreads();

  int total = s_count2;
  if (TWO_TIER) {
    // tier 1 on dense lanes; what it cannot decide is compacted in place for the clipper below (by the barrier every
    // entry below q0 + NMS_NT has been read, and at most that many were kept)
    __shared__ int s_count3;
    if (tid == 0) s_count3 = 0;
    __syncthreads();
    for (int q0 = 0; q0 < total; q0 += NMS_NT) {
      const int q = q0 + tid;
      unsigned e = 0;
      if (q < total) e = s_queue[q];
      __syncthreads();
      bool undecided = false;
      if (q < total) {
        const int i = e >> 6, j = e & 63;
        bool danger, apart;
        const float v = pair_iou_fast<0>(s_row[i].p, s_col[j].p, danger, apart);
        // `apart`: the reference returns exactly 0 (a hit only for thr <= 0 with >=, which the budget test catches)
        undecided = danger || !(fabsf(v - thr) > kFastBudget);          // NaN lands here too
        if (!undecided && (GE ? (v >= thr) : (v > thr))) atomicOr(&s_mask[i], 1ull << j);
      }
      unsigned long long m = __ballot(undecided);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_count3, __popcll(m));
        base = __shfl(base, 0);
        if (undecided) s_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)e;
      }
    }
    __syncthreads();
    total = s_count3;
  }
  const int quad = tid >> 2;
  F2* qscr = s_pts + quad * kQuadSlots;
  for (int q = quad; q < total; q += NMS_NT / 4) {  // four lanes per pair (rsdet_geom.h)
    unsigned e = s_queue[q];
    int i = e >> 6, j = e & 63;
    float v = pair_iou_quad<0>(s_row[i].p, s_col[j].p, qscr, lane);  // box1 = earlier (kept) box, :443
    bool hit = GE ? (v >= thr) : (v > thr);
    if (hit && (tid & 3) == 0) atomicOr(&s_mask[i], 1ull << j);
  }
  __syncthreads();
  if (tid >= 64) return;
  if (rb == cbk) {  // diagonal tile, transposed: bit i of word j = "box i suppresses box j"
    unsigned long long col = 0ull;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) col |= ((s_mask[i] >> tid) & 1ull) << i;
    diag_t[rb * 64 + tid] = col;
    return;
  }
  nms_emit_entries(s_mask[tid], tid, rb, cbk, col_blocks, entries, blk_cnt);
}

// One workgroup; the `removed` bitmap lives in LDS (n <= 64*NMS_MAX_BLOCKS boxes).  The walk over
// the 64-box blocks is serial by nature, so the kernel is a latency chain and is built to keep
// everything but the decision itself off that chain:
//   * the transposed diagonal word, the `order` entries and the first SWEEP_PRE entries of block
//     bk+2 are requested while block bk is being decided (three register buffers in rotation,
//     never a vmcnt(0): the barriers order LDS only, the loads are unconditional so that the
//     compiler can count them);
//   * wave 0 resolves the diagonal tile as a fixpoint over wave ballots (0-6 rounds typically)
//     instead of a 64-step serial chain;
//   * every thread applies its prefetched entries of kept rows with 64-bit LDS ds_or.
// History (M = 5 344, 84 blocks): one thread per word with one dependent load per kept row 9 us per
// block; dense rows prefetched by 16 waves 1.4 us per block (64 KB per step through one CU's memory
// pipe); sparse entries: see DESIGN.md.
#ifdef RSDET_SWEEP_TRACE  // debug builds only (profiles/scripts/trace_sweep.py): per-step stage timestamps of wave 0, 100 MHz wall clock
__device__ unsigned long long* g_sweep_trace;
#define STRACE(bk, k, v)                                                                 \
  do {                                                                                   \
    if (threadIdx.x == 0 && g_sweep_trace) g_sweep_trace[(size_t)(bk) * 8 + (k)] = (v);  \
  } while (0)
#else
#define STRACE(bk, k, v)
#endif
constexpr int NMS_MAX_BLOCKS = 8192;  // 524288 boxes, 96 KB of LDS
constexpr int SWEEP_NT = 512;
constexpr int SWEEP_EPT = 2;                      // prefetched entries per thread
constexpr int SWEEP_PRE = SWEEP_NT * SWEEP_EPT;   // entries of a block that come from the prefetch

__global__ __launch_bounds__(SWEEP_NT) void nms_sweep_kernel(
    const NmsEntry* __restrict__ entries, const unsigned* __restrict__ blk_cnt,
    const unsigned long long* __restrict__ diag_t, int n, int col_blocks,
    const int* __restrict__ order, uint8_t* __restrict__ keep, const int* __restrict__ segs) {
  extern __shared__ unsigned long long s_removed[];  // col_blocks words, 1 kept word, col_blocks counts
  unsigned long long* s_kept = s_removed + col_blocks;
  unsigned* s_cnt = reinterpret_cast<unsigned*>(s_removed + col_blocks + 1);
  __shared__ int s_end;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t seg = (size_t)64 * col_blocks;  // entries reserved per row block

  // Segments: runs of one label in a label-major order are independent NMS problems -- one workgroup each,
  // concurrently (a 15-class set: 15 short walks instead of one long one).  No table / too many runs: one walk.
  int nseg = segs ? segs[0] : 1;
  const bool whole = !segs || nseg > NMS_MAX_SEGS;
  if (whole) nseg = 1;
  for (int si = blockIdx.x; si < nseg; si += gridDim.x) {
    int s0 = 0, e0 = n;
    if (!whole) {
      s0 = segs[1 + si];
      if (tid == 0) s_end = n;
      __syncthreads();
      int mine_end = n;
      for (int k = tid; k < nseg; k += SWEEP_NT) {
        const int st = segs[1 + k];
        if (st > s0) mine_end = min(mine_end, st);
      }
      if (mine_end < n) atomicMin(&s_end, mine_end);
      __syncthreads();
      e0 = s_end;
    }
    const int b0 = s0 >> 6, b1 = (e0 - 1) >> 6;
    for (int w = b0 + tid; w <= b1; w += SWEEP_NT) {
      s_removed[w - b0] = 0ull;
      s_cnt[w - b0] = blk_cnt[w];
    }
    __syncthreads();

    struct Pre {  // everything block bk needs from memory
      uint4 ent[SWEEP_EPT];
      unsigned long long diag;
      int ord;
    };
    auto prefetch = [&](int bk, Pre& p) {
      p.diag = diag_t[bk * 64 + lane];  // padded to whole blocks
      p.ord = order[min(bk * 64 + lane, n - 1)];
      const uint4* e = reinterpret_cast<const uint4*>(entries + (size_t)bk * seg);
#pragma unroll
      for (int k = 0; k < SWEEP_EPT; ++k) p.ent[k] = e[k * SWEEP_NT + tid];  // seg >= SWEEP_PRE slots exist (ws slack)
    };
    // one block: `cur` was requested two steps ago (a step is shorter than a trip to memory), `fill`
    // is requested now for block bk + 2 and first touched two steps on.  The loop is unrolled by
    // three with the buffers rotated -- no register moves, so no wait for the loads in flight.
    auto step = [&](int bk, const Pre& cur, Pre& fill) {
      STRACE(bk, 0, wall_clock64());
      prefetch(min(bk + 2, b1), fill);
      // boxes of this block that belong to the segment (a block at a segment border is shared with its neighbour)
      const int lo = max(s0 - bk * 64, 0), hi = min(e0 - bk * 64, 64);
      const unsigned long long segmask = (hi >= 64 ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << lo) - 1ull);
      if (wave == 0) {
        const unsigned long long cur_v = s_removed[bk - b0];
        const unsigned cur_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur_v);
        const unsigned cur_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur_v >> 32));
        const unsigned long long pre = ((unsigned long long)cur_hi << 32) | cur_lo;  // wave-uniform: scalar registers
        const unsigned long long cand = ~pre & segmask;
        // Greedy NMS inside the 64-box block as a fixpoint instead of a 64-step serial chain.  Lane j
        // holds column j of the diagonal tile (bit i: box i < j suppresses box j).  K <- cand minus the
        // boxes suppressed by a member of K: after t rounds the first t decisions are final, so the
        // fixpoint is the greedy answer (unique: r is kept iff no earlier kept box suppresses it); a
        // typical block needs 0-6 rounds, the worst case 64.
        unsigned long long kept = cand;
        for (int round = 0; round < 64; ++round) {
          const unsigned long long hit = __ballot((cur.diag & kept) != 0ull);
          const unsigned long long next = cand & ~hit;
          if (next == kept) break;
          kept = next;
        }
        STRACE(bk, 1, wall_clock64());
        if ((segmask >> lane) & 1ull) keep[cur.ord] = (uint8_t)((kept >> lane) & 1ull);
        if (lane == 0) *s_kept = kept;
      }
      lds_barrier();
      STRACE(bk, 2, wall_clock64());
      const unsigned long long kept = *s_kept;
      const unsigned cnt = s_cnt[bk - b0];
#pragma unroll
      for (int k = 0; k < SWEEP_EPT; ++k) {
        const uint4 e = cur.ent[k];  // {bits lo, bits hi, column block, row}
        if ((unsigned)(k * SWEEP_NT + tid) < cnt && ((kept >> (e.w & 63u)) & 1ull) && (int)e.z <= b1)
          atomicOr(&s_removed[(int)e.z - b0], ((unsigned long long)e.y << 32) | e.x);
      }
      // a block with more than SWEEP_PRE non-zero words: the rest straight from memory
      for (unsigned q = SWEEP_PRE + tid; q < cnt; q += SWEEP_NT) {
        const NmsEntry e = entries[(size_t)bk * seg + q];
        if (((kept >> e.row) & 1ull) && e.cblock <= b1) atomicOr(&s_removed[e.cblock - b0], e.bits);
      }
      lds_barrier();
      STRACE(bk, 3, wall_clock64());
    };

    Pre A, B, C;
    prefetch(b0, A);
    prefetch(min(b0 + 1, b1), B);
    for (int bk = b0; bk <= b1; bk += 3) {
      step(bk, A, C);
      if (bk + 1 <= b1) step(bk + 1, B, A);
      if (bk + 2 <= b1) step(bk + 2, C, B);
    }
    __syncthreads();  // LDS is re-initialised for the next segment of this workgroup
  }
}

}  // namespace rsdet

using namespace rsdet;

#ifdef RSDET_SWEEP_TRACE
extern "C" void rsdet_debug_set_sweep_trace(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_trace), &p, sizeof(p)); }
#endif

void rsdet_launch_nms_sweep(const void* entries, const unsigned* blk_cnt, const unsigned long long* diag_t, int n,
                            int col_blocks, const int* order, unsigned char* keep, hipStream_t stream) {
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(SWEEP_NT), (size_t)(col_blocks + 1) * 8 + (size_t)col_blocks * 4,
                     stream, (const NmsEntry*)entries, blk_cnt, diag_t, n, col_blocks, order, keep, (const int*)nullptr);
}

static inline size_t nms_sorted_bytes(int n) { return ((size_t)n * sizeof(NmsBox) + 255) & ~(size_t)255; }

// ws layout after the kernel-specific head: diag_t (64*cb words) | blk_cnt (cb) | entries (cb lists of 64*cb)
static inline size_t nms_diag_bytes(int n) { return (((size_t)n + 63) / 64) * 64 * sizeof(unsigned long long); }
static inline size_t nms_cnt_bytes(int n) {  // per 64-box block: entry count (u32) and label range (float2)
  return (((((size_t)n + 63) / 64) * 4 + 255) & ~(size_t)255) + (((((size_t)n + 63) / 64) * 8 + 255) & ~(size_t)255);
}
static inline size_t nms_entry_bytes(int n) {
  size_t cb = ((size_t)n + 63) / 64;
  return (cb * 64 * cb + rsdet::SWEEP_PRE) * sizeof(rsdet::NmsEntry);  // + slack: unconditional prefetch of the last list
}

static inline size_t nms_seg_bytes() { return (((size_t)NMS_MAX_SEGS + 1) * 4 + 255) & ~(size_t)255; }

extern "C" size_t rsdet_nms_rotated_ws_size(int n) {
  if (n <= 0) return 0;
  return nms_sorted_bytes(n) + nms_diag_bytes(n) + nms_cnt_bytes(n) + nms_seg_bytes() + nms_entry_bytes(n);
}

extern "C" int rsdet_nms_rotated_f32(const float* dets, int n, int box_len, const int* order,
                                     float thr, int ge, uint8_t* keep, void* ws, size_t ws_bytes,
                                     void* stream) {
  if (n < 0 || (box_len != 5 && box_len != 6)) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!dets || !order || !keep || !ws) return RSDET_EINVAL;
  if (ws_bytes < rsdet_nms_rotated_ws_size(n) || ((uintptr_t)ws & 15)) return RSDET_EINVAL;
  const int cb = (n + 63) / 64;
  if (cb > NMS_MAX_BLOCKS) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  NmsBox* sorted = (NmsBox*)ws;
  char* w = (char*)ws + nms_sorted_bytes(n);
  unsigned long long* diag_t = (unsigned long long*)w;
  unsigned* blk_cnt = (unsigned*)(w + nms_diag_bytes(n));
  float2* blk_label = (float2*)(w + nms_diag_bytes(n) + ((((size_t)cb * 4) + 255) & ~(size_t)255));
  int* segs = (int*)(w + nms_diag_bytes(n) + nms_cnt_bytes(n));
  NmsEntry* entries = (NmsEntry*)(w + nms_diag_bytes(n) + nms_cnt_bytes(n) + nms_seg_bytes());
  const int label_major = (ge >> 1) & 1;  // RSDET_NMS_LABEL_MAJOR
  ge &= 1;
  if (hipMemsetAsync(segs, 0, sizeof(int), s) != hipSuccess) return RSDET_ELAUNCH;
  hipLaunchKernelGGL(nms_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dets, n, box_len,
                     order, sorted, blk_cnt, blk_label, label_major, segs);
  static const bool exact_all = [] {   // A/B switch: RSDET_NMS_EXACT=1 runs the reference-order clipper on every pair
    const char* e = getenv("RSDET_NMS_EXACT");
    return e && e[0] == '1';
  }();
#define RSDET_NMS_MASK(G, T)                                                                                       \
  hipLaunchKernelGGL((nms_mask_kernel<G, T>), dim3(cb, cb), dim3(NMS_NT), 0, s, sorted, n, thr, cb, entries, blk_cnt, \
                     diag_t, blk_label, (ge && !(thr > 0.f)) ? 1 : 0)
  if (ge) {
    if (exact_all) RSDET_NMS_MASK(true, false); else RSDET_NMS_MASK(true, true);
  } else {
    if (exact_all) RSDET_NMS_MASK(false, false); else RSDET_NMS_MASK(false, true);
  }
#undef RSDET_NMS_MASK
  // one workgroup per label run (at most 64 in flight; more runs are walked in turn); idle workgroups exit at once
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(label_major ? 64 : 1), dim3(SWEEP_NT),
                     (size_t)(cb + 1) * 8 + (size_t)cb * 4, s, entries, blk_cnt, diag_t, n, cb, order, keep, segs);
  return rsdet_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Horizontal-box NMS (Oriented-RCNN proposal stage).  Replaces Jittor's built-in `jt.nms`
// (third-party, call sites /root/reference/python/jdet/models/roi_heads/oriented_rpn_head.py:219,
// ops/nms.py:9,44): boxes (x1,y1,x2,y2) already gathered in descending-score order, IoU with the
// legacy "+1" pixel convention when plus_one != 0, suppression on IoU > thr.  Same mask + device
// sweep structure as the rotated NMS above; keep[] is indexed by SORTED position.
namespace rsdet {

__global__ __launch_bounds__(64) void nms_hbb_mask_kernel(const float* __restrict__ boxes, int n, float thr,
                                                          float one, int col_blocks,
                                                          NmsEntry* __restrict__ entries,
                                                          unsigned* __restrict__ blk_cnt,
                                                          unsigned long long* __restrict__ diag_t) {
  const int rb = blockIdx.y, cbk = blockIdx.x;
  if (cbk < rb) return;
  __shared__ float s_col[64 * 4];
  __shared__ unsigned long long s_rows[64];
  const int tid = threadIdx.x;
  const int cols = min(64, n - cbk * 64), rows = min(64, n - rb * 64);
  if (tid < cols) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s_col[tid * 4 + k] = boxes[(long long)(cbk * 64 + tid) * 4 + k];
  }
  __syncthreads();
  unsigned long long bits = 0ull;
  if (tid < rows) {
    const float* b = boxes + (long long)(rb * 64 + tid) * 4;
    const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
    const float area = (x2 - x1 + one) * (y2 - y1 + one);
    const int start = (rb == cbk) ? tid + 1 : 0;
    for (int j = start; j < cols; ++j) {
      const float* c = s_col + j * 4;
      float w = fmaxf(0.f, fminf(x2, c[2]) - fmaxf(x1, c[0]) + one);
      float h = fmaxf(0.f, fminf(y2, c[3]) - fmaxf(y1, c[1]) + one);
      float inter = w * h;
      float carea = (c[2] - c[0] + one) * (c[3] - c[1] + one);
      if (inter / (area + carea - inter) > thr) bits |= 1ull << j;
    }
  }
  if (rb != cbk) {
    nms_emit_entries(bits, tid, rb, cbk, col_blocks, entries, blk_cnt);
    return;
  }
  s_rows[tid] = bits;  // diagonal tile also transposed for the sweep (see nms_mask_kernel)
  __syncthreads();
  unsigned long long col = 0ull;
#pragma unroll 8
  for (int i = 0; i < 64; ++i) col |= ((s_rows[i] >> tid) & 1ull) << i;
  diag_t[rb * 64 + tid] = col;
}

__global__ void iota_kernel(int* p, int n, unsigned* blk_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (n + 63) / 64) blk_cnt[i] = 0u;
  if (i < n) p[i] = i;
}

}  // namespace rsdet

extern "C" size_t rsdet_nms_hbb_ws_size(int n) {
  if (n <= 0) return 0;
  return (((size_t)n * 4 + 255) & ~(size_t)255) + nms_diag_bytes(n) + nms_cnt_bytes(n) + nms_entry_bytes(n);
}

extern "C" int rsdet_nms_hbb_sorted_f32(const float* boxes_sorted, int n, float thr, int plus_one,
                                        uint8_t* keep_sorted, void* ws, size_t ws_bytes, void* stream) {
  if (n < 0) return RSDET_EINVAL;
  if (n == 0) return RSDET_OK;
  if (!boxes_sorted || !keep_sorted || !ws || ws_bytes < rsdet_nms_hbb_ws_size(n) || ((uintptr_t)ws & 15))
    return RSDET_EINVAL;
  const int cb = (n + 63) / 64;
  if (cb > rsdet::NMS_MAX_BLOCKS) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int* ident = (int*)ws;
  const size_t ident_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
  char* w = (char*)ws + ident_bytes;
  unsigned long long* diag_t = (unsigned long long*)w;
  unsigned* blk_cnt = (unsigned*)(w + nms_diag_bytes(n));
  rsdet::NmsEntry* entries = (rsdet::NmsEntry*)(w + nms_diag_bytes(n) + nms_cnt_bytes(n));
  hipLaunchKernelGGL(rsdet::iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ident, n, blk_cnt);
  hipLaunchKernelGGL(rsdet::nms_hbb_mask_kernel, dim3(cb, cb), dim3(64), 0, s, boxes_sorted, n, thr,
                     plus_one ? 1.f : 0.f, cb, entries, blk_cnt, diag_t);
  rsdet_launch_nms_sweep(entries, blk_cnt, diag_t, n, cb, ident, keep_sorted, s);
  return rsdet_launch_status();
}
