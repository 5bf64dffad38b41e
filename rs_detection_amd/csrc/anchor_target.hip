// anchor_target.hip -- rotated IoU in ONE launch (detect + clip inside the tile) and the fused,
// sparse anchor-target path built on it (gfx950 / CDNA4).
//
// Replaces, for a whole batch and without ever materialising the (K, A) overlaps matrix:
//   anchor_target_single        /root/reference/python/jdet/models/boxes/anchor_target.py:105-180
//   MaxIoUAssigner.assign       /root/reference/python/jdet/models/boxes/assigner.py:65-170
//   box_iou_rotated             /root/reference/python/jdet/ops/box_iou_rotated.py:502-509
//   PseudoSampler.sample        /root/reference/python/jdet/models/boxes/sampler.py:114-130
//   bbox2delta_rotated          /root/reference/python/jdet/models/boxes/box_ops.py:184-230
//
// Round 1 ran the IoU as three dependent launches (prepare -> filter -> clip through a global work
// queue): 36 us for the S2ANet step shape, bound by three launch / first-touch latency chains, then
// two HBM passes of the assigner over the 48 MB matrix (23 us) and ~15 torch kernels for the targets.
//
// Here a tile (16 gts x 256 anchors) is finished by the workgroup that owns it:
//   stage      gts are prepared in the workgroup (16 fp64 sincos), anchors come prepared (cacheable:
//              the FAM grid never changes), one 16-byte-per-lane coalesced copy into LDS
//   detect     strip culling against the 64-column boxes, bounding circles, separating axes on dense
//              lanes -> the tile's surviving pairs in LDS (1.2 % of the pairs at S2ANet shapes)
//   dense mode every NON-surviving position is zero-filled right here (one store per element, issued
//              before the clip and never waited for), the survivors are clipped by the tile's own 64
//              quads (rsdet_geom.h, 4 lanes per pair) and each value is stored once.  No element is
//              written twice, so there is no store-ordering hazard across XCD L2s and no fence.
//   sparse     (anchor targets) nothing dense is written at all: each survivor's IoU goes to a compact
//   mode       entry list (8 B) + a row-maximum atomic; `at_final` then owns 256 anchors of one image,
//              folds that image's entries into column max / first argmax (LDS 64-bit atomic max) and
//              the low-quality rule (last gt whose IoU EQUALS its row maximum; a gt that overlaps
//              nothing claims every anchor, assigner.py:151-160 with min_pos_iou = 0), and writes
//              labels, label weights, encoded box targets, box weights and the pos / neg counts.
// Heavy tiles first: the anchors of the top pyramid levels (the last columns) overlap nearly every gt,
// so column tiles are walked from the last one down -- the hardware dispatcher is the work queue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_coder.h"
#include "rsdet_geom_fast.h"
#include "rsdet_tile.h"

namespace rsdet {

#ifdef RSDET_TILE_TRACE  // debug builds only (profiles/scripts/trace_tiles.py): per-workgroup stage timestamps, 100 MHz
__device__ unsigned long long* g_tile_trace;
#define TTRACE_REAL(k)                                                                  \
  do {                                                                                  \
    if (threadIdx.x == 0 && g_tile_trace) g_tile_trace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); \
  } while (0)
#ifdef RSDET_TRACE_FINISH_ONLY
#define TTRACE(k)
#else
#define TTRACE(k) TTRACE_REAL(k)
#endif
#define FTRACE(k)                                                                       \
  do {                                                                                  \
    if (threadIdx.x == 0 && g_tile_trace)                                               \
      g_tile_trace[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); \
  } while (0)
#else
#define TTRACE(k)
#define FTRACE(k)
#endif

constexpr int T_NT = 256;  // columns per tile = threads per workgroup
constexpr int T_TI = 16;   // rows per tile
#ifndef RSDET_TILE_QUADS
#define RSDET_TILE_QUADS 64
#endif
constexpr int T_QUADS = RSDET_TILE_QUADS;  // quads (of 4 lanes) that clip: waves [0, T_QUADS / 16) of the workgroup
static_assert(T_QUADS % 16 == 0 && T_QUADS >= 16 && T_QUADS <= T_NT / 4, "whole waves");

struct AtEntry {
  unsigned key;  // (row inside its group << 8) | column inside the tile
  float v;
};

struct WorkItem {  // one surviving pair of the split dense mode
  int row, col;  // position in the IoU matrix
  int p2;        // index of the column box in the prepared array (col + slab)
  int pad;
};
constexpr int IOU2_SHARDS = 64;

struct TileDesc {  // one per row tile (host-built when the gt counts are host-known)
  int group, row0, nrows, group_row0;
};

// One wave per 64 boxes: prepared boxes + the padded bounding box of their 64 bounding circles.
// `n_per_group` columns per slab (strip words never straddle two images).
__global__ __launch_bounds__(64) void at_prepare_kernel(const float* __restrict__ boxes, int n_per_group, int stride,
                                                        int cw, int pitch, BoxPre* __restrict__ pre,
                                                        float4* __restrict__ colbox) {
  const int lane = threadIdx.x;
  const int word = blockIdx.x;
  const int slab = word / cw, k = word - slab * cw;
  const int col = k * 64 + lane;
  float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
  if (col < n_per_group) {
    const long long j = (long long)slab * n_per_group + col;
    const BoxPre p = prepare_box(boxes + j * stride);
    pre[(long long)slab * pitch + col] = p;   // slabs start on 16-byte boundaries (even pitch): the tile copy is float4
    const float pad = 1.001f * p.rad + 1e-5f * (fabsf(p.cx) + fabsf(p.cy));
    const bool finite = fabsf(p.cx) < INFINITY && fabsf(p.cy) < INFINITY && pad < INFINITY;  // false for NaN too
    x0 = finite ? p.cx - pad : -INFINITY;
    y0 = finite ? p.cy - pad : -INFINITY;
    x1 = finite ? p.cx + pad : INFINITY;
    y1 = finite ? p.cy + pad : INFINITY;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    x0 = fminf(x0, __shfl_xor(x0, off));
    y0 = fminf(y0, __shfl_xor(y0, off));
    x1 = fmaxf(x1, __shfl_xor(x1, off));
    y1 = fmaxf(y1, __shfl_xor(y1, off));
  }
  if (lane == 0) colbox[word] = make_float4(x0, y0, x1, y1);
}

struct TileArgs {
  const float* boxes1;  // raw gts (n1, stride1)
  const BoxPre* pre1;   // the same gts prepared (optional; nullptr: 16 lanes of the workgroup run the fp64 sincos)
  int stride1, n1;
  const BoxPre* pre2;     // prepared columns, slab g at g * pitch when per_group != 0
  const float4* colbox;   // one per (slab, 64 columns)
  int n2, cw, per_group;
  const int* row_offsets;  // n_groups + 1 (device); nullptr: one group = all rows
  const TileDesc* tiles;   // n_row_tiles descriptors, or nullptr (then row tiles = n_groups x ny)
  int n_row_tiles, ny, nx;
  int split_xt;  // column tiles >= split_xt are cut into T_SUB sub-tiles of T_TI / T_SUB rows (performance hint only)
  // dense mode
  float* out;
  // split dense mode (detect kernel -> clip + fill kernel)
  BoxPre* pre1_out;             // the prepared gts, written by the detect kernel for the clip kernel
  unsigned long long* tilemask; // T_WORDS survivor words per directory slot
  unsigned* tileflag;           // per slot: 0, or 1 + the number of survivors that did fit the queue (overflow)
  struct WorkItem* queue;       // IOU2_SHARDS shards of `capacity` items
  unsigned* counter;            // one fill counter per shard (128-byte lines), zero on entry
  unsigned capacity;
  // sparse mode
  const unsigned char* valid;  // optional (n_groups, n2): columns outside it are never candidates
  unsigned* rowmax;            // (n1) float bits, zero on entry
  AtEntry* list;               // T_SUB sub-slices of T_TI / T_SUB * T_NT entries per (column tile, row tile)
  unsigned* dir;               // (nx, n_row_tiles, T_SUB): entries in each sub-slice
  unsigned long long* colkey;  // (n_groups, n2) column max | first argmax accumulators, zero on entry (0 = no positive IoU)
};

__device__ __forceinline__ unsigned long long at_pack(float v, int row) {
  return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)row);
}

// Position of the r-th (0-based) set bit of m; r < popcount(m).
__device__ __forceinline__ int select_bit64(unsigned long long m, int r) {
  unsigned w = (unsigned)m;
  int pos = 0;
  const int c = __popc(w);
  if (r >= c) {
    r -= c;
    w = (unsigned)(m >> 32);
    pos = 32;
  }
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) {
    const unsigned low = w & ((1u << s) - 1u);
    const int c2 = __popc(low);
    if (r >= c2) {
      r -= c2;
      w >>= s;
      pos += s;
    } else {
      w = low;
    }
  }
  return pos;
}

constexpr int T_SUB = 4;            // row sub-tiles of a heavy column tile
constexpr int T_KEEP = 8;           // clip rounds whose IoU stays in registers for the row-maximum filter
constexpr int T_WORDS = T_TI * (T_NT / 64);  // one 64-bit mask word per (row, wave)
static_assert(T_WORDS == 64, "one wave scans the mask words");

// word + bit of the k-th set bit over the T_WORDS mask words (inclusive prefix counts in `end`)
__device__ __forceinline__ void locate(const unsigned long long* __restrict__ mask, const unsigned short* __restrict__ end,
                                       int k, int& word, int& bit) {
  int lo = 0;
#pragma unroll
  for (int step = T_WORDS / 2; step > 0; step >>= 1)
    if ((int)end[lo + step - 1] <= k) lo += step;
  const int before = lo ? (int)end[lo - 1] : 0;
  word = lo;
  bit = select_bit64(mask[lo], k - before);
}

__device__ __forceinline__ void scan_words(const unsigned long long* __restrict__ mask, unsigned short* __restrict__ end,
                                           int tid) {
  if (tid < 64) {  // wave 0: inclusive scan of the 64 popcounts
    int c = __popcll(mask[tid]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(c, off);
      if (tid >= off) c += o;
    }
    end[tid] = (unsigned short)c;
  }
}

struct TileId {
  int xt, rt, sub, g, row0, nrows, grow0;
  size_t slot;
  bool heavy;
};
// linear workgroup id -> tile (heavy column tiles first, each cut into T_SUB row sub-tiles)
__device__ __forceinline__ TileId decode_tile(const TileArgs& a, int b) {
  TileId t;
  const int n_heavy = (a.nx - a.split_xt) * a.n_row_tiles * T_SUB;
  t.sub = 0;
  t.heavy = b < n_heavy;
  if (t.heavy) {
    const int per = a.n_row_tiles * T_SUB;
    t.xt = a.nx - 1 - b / per;
    const int rem = b % per;
    t.rt = rem / T_SUB;
    t.sub = rem - t.rt * T_SUB;
  } else {
    const int id = b - n_heavy;
    t.xt = a.split_xt - 1 - id / a.n_row_tiles;
    t.rt = id % a.n_row_tiles;
  }
  if (a.tiles) {
    const TileDesc d = a.tiles[t.rt];
    t.g = d.group, t.row0 = d.row0, t.nrows = d.nrows, t.grow0 = d.group_row0;
  } else {
    t.g = t.rt / a.ny;
    const int y = t.rt - t.g * a.ny;
    int rb = 0, re = a.n1;
    if (a.row_offsets) {
      rb = a.row_offsets[t.g];
      re = a.row_offsets[t.g + 1];
    }
    t.grow0 = rb;
    t.row0 = rb + y * T_TI;
    t.nrows = min(T_TI, re - t.row0);
  }
  t.slot = ((size_t)t.xt * a.n_row_tiles + t.rt) * T_SUB + t.sub;
  if (t.heavy) {
    t.row0 += t.sub * (T_TI / T_SUB);
    t.nrows = min(T_TI / T_SUB, t.nrows - t.sub * (T_TI / T_SUB));
  }
  return t;
}

// ---- split dense mode, launch 1: detection only ---------------------------------------------------------------------
// The fused tile kernel above is bound by its LDS footprint (25 KB: 6 workgroups per CU) and by the ~800 VALU
// instructions of a clip round stretching every latency-bound detection step of its neighbours (measured: 37 us, the
// same as round 1's three launches).  Detection alone needs 12 KB and no clipper: 8 workgroups per CU, ~4 us per tile.
// It leaves, per tile, the survivor bit mask (the fill of launch 2 skips exactly those positions) and the survivors as
// work items in a sharded global queue (launch 2 clips them perfectly balanced), plus the prepared gts.
template <int VERSION>
__global__ __launch_bounds__(T_NT) void iou_detect_kernel(const TileArgs a) {
  __shared__ BoxPre s_row[T_TI];
  __shared__ __attribute__((aligned(16))) BoxPre s_col[T_NT];
  __shared__ unsigned long long s_cm[T_WORDS], s_sm[T_WORDS];
  __shared__ unsigned short s_cend[T_WORDS], s_send[T_WORDS];
  __shared__ unsigned s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TileId t = decode_tile(a, blockIdx.x);
  unsigned long long* mask_out = a.tilemask + t.slot * T_WORDS;
  if (t.nrows <= 0) {
    if (tid < T_WORDS) mask_out[tid] = 0ull;
    if (tid == 0) a.tileflag[t.slot] = 0u;
    return;
  }
  const long long slab = a.per_group ? (long long)t.g * ((a.n2 + 1) & ~1) : 0;
  const int slab_word = a.per_group ? t.g * a.cw : 0;
  const BoxPre* p2 = a.pre2 + slab;
  const int col0 = t.xt * T_NT, col = col0 + tid;
  const bool col_ok = col < a.n2;
  const int ncols = min(T_NT, a.n2 - col0);
  {
    const float4* src = reinterpret_cast<const float4*>(p2 + col0);
    float4* dst = reinterpret_cast<float4*>(s_col);
    const int n16 = (ncols * (int)sizeof(BoxPre)) / 16;
    for (int k = tid; k < n16; k += T_NT) dst[k] = src[k];
    if ((ncols & 1) && tid == 0) {
      const float2* s2 = reinterpret_cast<const float2*>(p2 + col0);
      reinterpret_cast<float2*>(s_col)[n16 * 2] = s2[n16 * 2];
    }
  }
  const int kw = (col0 >> 6) + wave;
  float4 cb = a.colbox[slab_word + min(kw, a.cw - 1)];
  if (kw >= a.cw) cb = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
  if (tid < t.nrows) {
    const BoxPre r = a.pre1 ? a.pre1[t.row0 + tid] : prepare_box(a.boxes1 + (long long)(t.row0 + tid) * a.stride1);
    s_row[tid] = r;
    if (t.xt == 0) a.pre1_out[t.row0 + tid] = r;  // one column tile publishes the prepared gts for launch 2
  }
  if (tid < T_WORDS) {
    s_cm[tid] = 0ull;
    s_sm[tid] = 0ull;
  }
  __syncthreads();
  const BoxPre mine = s_col[col_ok ? tid : 0];
  bool lv = false;
  if (lane < t.nrows) {
    const float rx = s_row[lane].cx, ry = s_row[lane].cy;
    const float dx = fmaxf(fmaxf(cb.x - rx, rx - cb.z), 0.f), dy = fmaxf(fmaxf(cb.y - ry, ry - cb.w), 0.f);
    const float thr = 1.001f * s_row[lane].rad + 1e-5f * (fabsf(rx) + fabsf(ry));
    lv = !(dx * dx + dy * dy > thr * thr);
  }
  unsigned live = (unsigned)__ballot(lv);
  while (live) {
    const int i = __builtin_ctz(live);
    live &= live - 1u;
    bool cand = false;
    if (col_ok) {
      const float dx = s_row[i].cx - mine.cx, dy = s_row[i].cy - mine.cy;
      const float r = s_row[i].rad + mine.rad;
      cand = !(dx * dx + dy * dy > r * r * 1.0001f);
    }
    const unsigned long long m = __ballot(cand);
    if (m && lane == 0) s_cm[i * (T_NT / 64) + wave] = m;
  }
  __syncthreads();
  scan_words(s_cm, s_cend, tid);
  __syncthreads();
  const int n_cand = s_cend[T_WORDS - 1];
  for (int k = tid; k < n_cand; k += T_NT) {
    int word, bit;
    locate(s_cm, s_cend, k, word, bit);
    if (!sat_disjoint<VERSION>(s_row[word >> 2], s_col[((word & 3) << 6) | bit])) atomicOr(&s_sm[word], 1ull << bit);
  }
  __syncthreads();
  scan_words(s_sm, s_send, tid);
  if (tid < T_WORDS) mask_out[tid] = s_sm[tid];
  __syncthreads();
  const int total = s_send[T_WORDS - 1];
  if (total == 0) {
    if (tid == 0) a.tileflag[t.slot] = 0u;
    return;
  }
  const unsigned shard = (blockIdx.x * 7u + blockIdx.x / 64u) % IOU2_SHARDS;
  if (tid == 0) s_base = atomicAdd(a.counter + shard * 32, (unsigned)total);
  __syncthreads();
  const unsigned base = s_base;
  const int fit = base >= a.capacity ? 0 : (int)min((unsigned)total, a.capacity - base);
  if (tid == 0) a.tileflag[t.slot] = fit == total ? 0u : 1u + (unsigned)fit;
  WorkItem* qd = a.queue + (size_t)shard * a.capacity + base;
  for (int q = tid; q < fit; q += T_NT) {
    int word, bit;
    locate(s_sm, s_send, q, word, bit);
    WorkItem w;
    w.row = t.row0 + (word >> 2);
    w.col = col0 + (((word & 3) << 6) | bit);
    w.p2 = (int)(slab + w.col);
    w.pad = 0;
    qd[q] = w;
  }
}

// ---- split dense mode, launch 2: zero fill (every position that is NOT a survivor) + the balanced clip ----------------
// No element of the matrix is written twice: the fill skips the survivor bits, the clip writes exactly those.  The
// stores of the fill are issued first and drain underneath the VALU-bound clip.
template <int VERSION>
__global__ __launch_bounds__(T_NT) void iou_clip_fill_kernel(const TileArgs a, int n_detect_blocks, unsigned* done) {
  __shared__ F2 s_pts[kQuadSlots * (T_NT / 4)];
  __shared__ unsigned s_end[IOU2_SHARDS];
  __shared__ unsigned long long s_sm[T_WORDS];
  __shared__ unsigned short s_send[T_WORDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  F2* qscr = s_pts + (tid >> 2) * kQuadSlots;
  // ---- fill
  for (int b = blockIdx.x; b < n_detect_blocks; b += gridDim.x) {
    const TileId t = decode_tile(a, b);
    if (t.nrows <= 0) continue;
    const int col0 = t.xt * T_NT, col = col0 + tid;
    const unsigned long long* m = a.tilemask + t.slot * T_WORDS;
    const unsigned flag = a.tileflag[t.slot];
    {
      // the 16 mask words of this wave's column strip first (independent loads), then the stores: a load per row
      // inside the store loop made the fill a chain of 16 dependent global round trips per tile (40 us for the launch)
      unsigned long long mw[T_TI];
#pragma unroll
      for (int i = 0; i < T_TI; ++i) mw[i] = i < t.nrows ? m[i * 4 + wave] : ~0ull;
      if (col < a.n2) {
        float* o = a.out + (long long)t.row0 * a.n2 + col;
        const unsigned long long bitm = 1ull << lane;
#pragma unroll
        for (int i = 0; i < T_TI; ++i)
          if (!(mw[i] & bitm)) o[(long long)i * a.n2] = 0.0f;
      }
    }
    if (flag) {  // queue overflow (rare): the survivors past the ones that fit are clipped right here
      if (tid < T_WORDS) s_sm[tid] = m[tid];
      __syncthreads();
      scan_words(s_sm, s_send, tid);
      __syncthreads();
      const int total = s_send[T_WORDS - 1];
      const long long slab = a.per_group ? (long long)t.g * ((a.n2 + 1) & ~1) : 0;
      for (int q0 = (int)flag - 1; q0 < total; q0 += T_NT / 4) {
        const int q = q0 + (tid >> 2);
        const bool on = q < total;
        int word, bit;
        locate(s_sm, s_send, on ? q : (int)flag - 1, word, bit);
        const int i = word >> 2, j = ((word & 3) << 6) | bit;
        const BoxPre ra = a.pre1_out[t.row0 + i], cbx = a.pre2[slab + col0 + j];
        const float v = pair_iou_quad<VERSION>(ra, cbx, qscr, lane);
        if (on && (tid & 3) == 0) a.out[(long long)(t.row0 + i) * a.n2 + col0 + j] = v;
        lds_wave_order();
      }
      __syncthreads();
    }
  }
  // ---- clip: the queue's pairs, a contiguous share per quad stride
  if (tid < 64) {
    static_assert(IOU2_SHARDS == 64, "one wave scans the shard counters");
    unsigned c = min(a.counter[tid * 32], a.capacity);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(c, off);
      if (tid >= off) c += o;
    }
    s_end[tid] = c;
  }
  __syncthreads();
  const unsigned total = s_end[IOU2_SHARDS - 1];
  const unsigned quads = gridDim.x * (T_NT / 4);
  for (unsigned q = blockIdx.x * (T_NT / 4) + (tid >> 2); q < total; q += quads) {
    int shard = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1)
      if (s_end[shard + step - 1] <= q) shard += step;
    const unsigned first = shard ? s_end[shard - 1] : 0u;
    const WorkItem w = a.queue[(size_t)shard * a.capacity + (q - first)];
    const BoxPre ra = a.pre1_out[w.row], cbx = a.pre2[w.p2];
    const float v = pair_iou_quad<VERSION>(ra, cbx, qscr, lane);
    if ((tid & 3) == 0) a.out[(long long)w.row * a.n2 + w.col] = v;
  }
  // ---- the last workgroup to finish puts the shard counters back to zero for the next call.  Two levels: one
  // returning atomic per workgroup on ONE word serialises at ~88 per us (2 048 workgroups: 23 us, measured as a 40 us
  // launch); 64 first-level words take 32 arrivals each, their 64 last arrivers meet on the second-level word.
  __syncthreads();
  if (tid == 0) {
    const unsigned lvl1 = blockIdx.x % IOU2_SHARDS;
    const unsigned members = (gridDim.x - lvl1 + IOU2_SHARDS - 1) / IOU2_SHARDS;
    if (atomicAdd(done + lvl1 * 32, 1u) == members - 1u) {
      atomicExch(done + lvl1 * 32, 0u);
      const unsigned groups = min((unsigned)IOU2_SHARDS, gridDim.x);
      if (atomicAdd(done + IOU2_SHARDS * 32, 1u) == groups - 1u) {
        atomicExch(done + IOU2_SHARDS * 32, 0u);
        for (int k = 0; k < IOU2_SHARDS; ++k) atomicExch(a.counter + k * 32, 0u);
      }
    }
  }
}

template <int VERSION, int SPARSE>
__global__ __launch_bounds__(T_NT) void iou_tile_kernel(const TileArgs a) {
  __shared__ BoxPre s_row[T_TI];
  __shared__ __attribute__((aligned(16))) BoxPre s_col[T_NT];
  // candidate / survivor sets as bit masks (1 KB) instead of index lists (8 KB): LDS is what bounds the number of
  // resident workgroups here, and the k-th-set-bit lookups cost ~50 instructions against ~800 for a clip round
  __shared__ unsigned long long s_cm[T_WORDS], s_sm[T_WORDS];
  __shared__ unsigned short s_cend[T_WORDS], s_send[T_WORDS];
  __shared__ F2 s_pts[kQuadSlots * T_QUADS];
  __shared__ unsigned s_rmax[T_TI];  // sparse: row maxima of this tile (float bits)
  __shared__ unsigned s_nent;        // sparse: entries written to the sub-slice

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifndef RSDET_TILE_NO_PRIO
  // detection is a chain of short, latency-bound steps; the clip at the end is ~800 VALU instructions per round.
  // With equal priority the clipping waves of the other workgroups on the SIMD stretch every detection step.
  __builtin_amdgcn_s_setprio(3);
#endif
  TTRACE(0);
  // Heavy column tiles first (see the header), each cut into T_SUB row sub-tiles: a tile of the top pyramid levels
  // holds ~10x the surviving pairs of an average one and would otherwise be the tail of the whole launch.
  const int n_heavy = (a.nx - a.split_xt) * a.n_row_tiles * T_SUB;
  int rt, xt, sub = 0;
  const bool heavy = (int)blockIdx.x < n_heavy;
  if (heavy) {
    const int per = a.n_row_tiles * T_SUB;
    xt = a.nx - 1 - (int)blockIdx.x / per;
    const int rem = (int)blockIdx.x % per;
    rt = rem / T_SUB;
    sub = rem - rt * T_SUB;
  } else {
    const int id = (int)blockIdx.x - n_heavy;
    xt = a.split_xt - 1 - id / a.n_row_tiles;
    rt = id % a.n_row_tiles;
  }
  int g, row0, nrows, grow0;
  if (a.tiles) {
    const TileDesc t = a.tiles[rt];
    g = t.group, row0 = t.row0, nrows = t.nrows, grow0 = t.group_row0;
  } else {
    g = rt / a.ny;
    const int y = rt - g * a.ny;
    int rb = 0, re = a.n1;
    if (a.row_offsets) {
      rb = a.row_offsets[g];
      re = a.row_offsets[g + 1];
    }
    grow0 = rb;
    row0 = rb + y * T_TI;
    nrows = min(T_TI, re - row0);
  }
  const size_t slot = ((size_t)xt * a.n_row_tiles + rt) * T_SUB + sub;  // sub-slice / directory slot of this workgroup
  if (heavy) {
    row0 += sub * (T_TI / T_SUB);
    nrows = min(T_TI / T_SUB, nrows - sub * (T_TI / T_SUB));
  } else if (SPARSE && tid >= 1 && tid < T_SUB) {
    a.dir[slot + tid] = 0u;  // a whole tile uses sub-slice 0 only (all T_TI x T_NT entries of it)
  }
  if (nrows <= 0) {  // row tile past the group's end (no tile table), or an empty sub-tile
    if (SPARSE && tid == 0) a.dir[slot] = 0u;
    return;
  }
  const long long slab = a.per_group ? (long long)g * ((a.n2 + 1) & ~1) : 0;
  const int slab_word = a.per_group ? g * a.cw : 0;
  const BoxPre* p2 = a.pre2 + slab;
  const int col0 = xt * T_NT;
  const int col = col0 + tid;
  bool col_ok = col < a.n2;
  const int ncols = min(T_NT, a.n2 - col0);

  // ---- stage: the tile's columns as one coalesced 16-byte-per-lane copy (BoxPre is 40 B: 2.5 float4),
  // the rows from the prepared gts (or prepared here: fp64 sincos by 16 lanes)
  {
    const float4* src = reinterpret_cast<const float4*>(p2 + col0);  // 40 * 256 * xt bytes: 16-byte aligned
    float4* dst = reinterpret_cast<float4*>(s_col);
    const int n16 = (ncols * (int)sizeof(BoxPre)) / 16;  // ncols*40/16; the tail (8 B when ncols is odd) below
    for (int k = tid; k < n16; k += T_NT) dst[k] = src[k];
    if ((ncols & 1) && tid == 0) {
      const float2* s2 = reinterpret_cast<const float2*>(p2 + col0);
      reinterpret_cast<float2*>(s_col)[n16 * 2] = s2[n16 * 2];
    }
  }
  const int kw = (col0 >> 6) + wave;
  float4 cb = a.colbox[slab_word + min(kw, a.cw - 1)];
  if (kw >= a.cw) cb = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);  // empty strip
  if (tid < nrows)
    s_row[tid] = a.pre1 ? a.pre1[row0 + tid] : prepare_box(a.boxes1 + (long long)(row0 + tid) * a.stride1);
  if (tid < T_WORDS) {
    s_cm[tid] = 0ull;
    s_sm[tid] = 0ull;
  }
  if (SPARSE) {
    if (tid < T_TI) s_rmax[tid] = 0u;
    if (tid == 0) s_nent = 0u;
    if (a.valid && col_ok) col_ok = a.valid[(long long)g * a.n2 + col] != 0;
  }
  __syncthreads();
  TTRACE(1);
  const BoxPre mine = s_col[col_ok ? tid : 0];

  // ---- strip culling: lane i < nrows tests row i's circle against the bounding box of this wave's 64
  // column circles.  Both sides carry a 1e-3 relative pad, so a culled strip satisfies
  // surely_disjoint() for each of its pairs; NaN / Inf boxes are never culled.
  static_assert(T_TI <= 32, "row mask is 32 bits wide");
  bool lv = false;
  if (lane < nrows) {
    const float rx = s_row[lane].cx, ry = s_row[lane].cy;
    const float dx = fmaxf(fmaxf(cb.x - rx, rx - cb.z), 0.f), dy = fmaxf(fmaxf(cb.y - ry, ry - cb.w), 0.f);
    const float thr = 1.001f * s_row[lane].rad + 1e-5f * (fabsf(rx) + fabsf(ry));
    lv = !(dx * dx + dy * dy > thr * thr);
  }
  unsigned live = (unsigned)__ballot(lv);

  // ---- pass A: bounding circles of the live strips; the ballot word IS the candidate set of (row, wave)
  while (live) {
    const int i = __builtin_ctz(live);
    live &= live - 1u;
    bool cand = false;
    if (col_ok) {
      const float dx = s_row[i].cx - mine.cx, dy = s_row[i].cy - mine.cy;
      const float r = s_row[i].rad + mine.rad;
      cand = !(dx * dx + dy * dy > r * r * 1.0001f);  // == !surely_disjoint(s_row[i], mine)
    }
    const unsigned long long m = __ballot(cand);
    if (m && lane == 0) s_cm[i * (T_NT / 64) + wave] = m;
  }
  __syncthreads();
  scan_words(s_cm, s_cend, tid);
  __syncthreads();
  TTRACE(2);

  // ---- pass B: separating axes on dense lanes (candidate k of the tile -> thread k mod 256)
  const int n_cand = s_cend[T_WORDS - 1];
  for (int k = tid; k < n_cand; k += T_NT) {
    int word, bit;
    locate(s_cm, s_cend, k, word, bit);
    const int i = word >> 2, j = ((word & 3) << 6) | bit;
    if (!sat_disjoint<VERSION>(s_row[i], s_col[j])) atomicOr(&s_sm[word], 1ull << bit);
  }
  __syncthreads();
  scan_words(s_sm, s_send, tid);
  __syncthreads();
  TTRACE(3);
  const int total = s_send[T_WORDS - 1];

  if (!SPARSE) {
    // ---- zero fill of every position the clipper will NOT write: issued now, drained underneath the clip
    if (col < a.n2) {
      float* o = a.out + (long long)row0 * a.n2 + col;
      const unsigned long long bitm = 1ull << lane;
      if (nrows == T_TI) {
#pragma unroll
        for (int i = 0; i < T_TI; ++i)
          if (!(s_sm[i * 4 + wave] & bitm)) o[(long long)i * a.n2] = 0.0f;
      } else {
        for (int i = 0; i < nrows; ++i)
          if (!(s_sm[i * 4 + wave] & bitm)) o[(long long)i * a.n2] = 0.0f;
      }
    }
  }
  TTRACE(4);
  if (total == 0) {
    if (SPARSE && tid == 0) a.dir[slot] = 0u;
    return;
  }
  if (!SPARSE && wave >= T_QUADS / 16) return;  // dense: the waves that do not clip are done (no barrier follows)
  static_assert(!SPARSE || T_QUADS == T_NT / 4, "the sparse epilogue has workgroup barriers: every wave clips");

  // ---- clip the survivors here: T_QUADS quads, 4 lanes per pair (rsdet_geom.h).  A wave whose 16 quads are all
  // past the end of the set skips the round (the clipper only uses wave-level ballots, no workgroup barrier).
#ifndef RSDET_TILE_NO_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  F2* qscr = s_pts + (tid >> 2) * kQuadSlots;
  const int quad = tid >> 2;
  AtEntry* slice = SPARSE ? a.list + slot * (T_TI / T_SUB * T_NT) : nullptr;
  const long long gcol0 = SPARSE ? (long long)g * a.n2 + col0 : 0;
  float vk[T_KEEP];
#pragma unroll
  for (int k = 0; k < T_KEEP; ++k) vk[k] = 0.f;
  int rounds_done = 0;
  for (int r = 0; r * T_QUADS + wave * 16 < total; ++r) {
    const int q = r * T_QUADS + quad;
    const bool on = q < total;
    int word, bit;
    locate(s_sm, s_send, on ? q : 0, word, bit);
    const int i = word >> 2, j = ((word & 3) << 6) | bit;
    // every lane of the wave takes part (ballots inside are shifted per quad); idle quads repeat pair 0
    const float v = pair_iou_quad<VERSION>(s_row[i], s_col[j], qscr, lane);
    if (!SPARSE) {
      if (on && (tid & 3) == 0) a.out[(long long)(row0 + i) * a.n2 + col0 + j] = v;
    } else {
      const bool lead = on && (tid & 3) == 0;
      const int rowg = row0 - grow0 + i;
      if (lead) {
        // column maximum / first argmax: one device-scope 64-bit atomic max per positive IoU, never waited for.
        // A NaN never wins a `>`; as the image's FIRST row it sticks (assigner.py:133 argmax seeded by row 0).
        if (v > 0.f) {
          atomicMax(a.colkey + gcol0 + j, at_pack(v, rowg));
          atomicMax(&s_rmax[i], __float_as_uint(v));
        } else if (v != v && rowg == 0) {
          atomicMax(a.colkey + gcol0 + j, ~0ull);
        }
      }
      // The low-quality rule needs the pairs whose IoU EQUALS their gt's maximum over ALL anchors; only a pair that
      // equals the maximum of its row inside this tile can.  The IoUs of the first T_KEEP rounds wait in registers
      // for the tile's row maxima; later rounds (tiles with > 512 survivors) are written out unfiltered -- a
      // superset is harmless, at_finish repeats the exact comparison against the global maximum.
      if (r < T_KEEP) {
#pragma unroll
        for (int k = 0; k < T_KEEP; ++k) vk[k] = (r == k) ? v : vk[k];
      } else if (lead && v > 0.f) {
        AtEntry en;
        en.key = ((unsigned)rowg << 8) | (unsigned)j;
        en.v = v;
        slice[atomicAdd(&s_nent, 1u)] = en;
      }
      rounds_done = r + 1;
    }
    lds_wave_order();
  }
  TTRACE(5);
  if (SPARSE) {
    __syncthreads();  // the tile's row maxima are final
    if (tid < nrows && s_rmax[tid] != 0u) atomicMax(a.rowmax + row0 + tid, s_rmax[tid]);
    const int kmax = min(rounds_done, T_KEEP);
    for (int k = 0; k < kmax; ++k) {
      const int q = k * T_QUADS + quad;
      float v = 0.f;
#pragma unroll
      for (int kk = 0; kk < T_KEEP; ++kk) v = (k == kk) ? vk[kk] : v;
      if (q < total && (tid & 3) == 0 && v > 0.f) {
        int word, bit;
        locate(s_sm, s_send, q, word, bit);
        const int i = word >> 2, j = ((word & 3) << 6) | bit;
        if (__float_as_uint(v) == s_rmax[i]) {
          AtEntry en;
          en.key = ((unsigned)(row0 - grow0 + i) << 8) | (unsigned)j;
          en.v = v;
          slice[atomicAdd(&s_nent, 1u)] = en;
        }
      }
    }
    __syncthreads();
    if (tid == 0) a.dir[slot] = s_nent;
  }
}

// ---- sparse mode with the TWO-TIER clipper (round 3) -------------------------------------------------------------------
// What the anchor targets need from the IoU values is decisions, not digits: per anchor the maximum over the gts against
// two thresholds and its FIRST argmax, per gt the anchors whose IoU EQUALS its maximum.  So every surviving pair gets the
// Green-integral IoU of rsdet_geom_fast.h (one lane per pair, ~350 instructions, |error| < 3e-6 against the reference,
// budgeted kFastBudget = 2e-5), and the reference-order clipper (~3 200 lane-instructions per pair) runs only where a
// decision could depend on the difference:
//   here      (a) pairs the fast path flags itself: the reference's fragile zone, IoU < 1e-6, NaN;
//             (b) per gt, the pairs within 2 x budget of the gt's best value inside this tile -- a superset of the pairs
//                 that can equal the gt's exact maximum; their exact values give the exact row maximum (global atomic
//                 max) and the `v == row maximum` entries of the low-quality rule (assigner.py:151-160);
//   at_finish (c) per anchor, when more than one gt comes within 2 x budget of the anchor's best value, or that value is
//                 within the budget of a threshold: those pairs only (kernel below).
// EVERY surviving pair leaves an entry {row, column, value, exact?} for (c); the column accumulators take the best value
// known here (exact where computed, fast otherwise).
constexpr unsigned AT_EXACT = 0x80000000u;   // entry key flag: the value is the reference-order clipper's
constexpr int T2_XCAP = 768;                 // pairs queued for the clipper per tile and phase (more: chunked)
constexpr int T2_QUADS = 32;                 // quads that clip: waves 0 and 1
#ifndef RSDET_T2_KEEP
#define RSDET_T2_KEEP 4
#endif
constexpr int T2_KEEP = RSDET_T2_KEEP;                   // rounds of 256 survivors whose fast values wait in registers
constexpr unsigned T2_NONE = 0xFFFFFFFFu, T2_PEND = 0x40000000u;

#ifndef RSDET_T2_WAVES
#define RSDET_T2_WAVES 8
#endif
template <int VERSION>
__global__ __launch_bounds__(T_NT) void at_tile2_kernel(const TileArgs a) {
  __shared__ BoxPre s_row[T_TI];
  __shared__ __attribute__((aligned(16))) BoxPre s_col[T_NT];
  __shared__ unsigned long long s_sm[T_WORDS];
  __shared__ unsigned short s_send[T_WORDS];
  __shared__ F2 s_pts[kQuadSlots * T2_QUADS];
  __shared__ unsigned short s_x[T2_XCAP];
  __shared__ float s_xv[T2_XCAP];          // the clipper's value of queue slot q
  __shared__ unsigned char s_xk[T2_XCAP];  // ... and what to keep of it
  __shared__ unsigned s_rapx[T_TI];   // best FAST value of the row inside the tile (float bits)
  __shared__ unsigned s_rex[T_TI];    // best EXACT value of the row inside the tile
  __shared__ unsigned s_nx, s_nent;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TTRACE(0);
  const TileId t = decode_tile(a, blockIdx.x);
  if (!t.heavy && tid >= 1 && tid < T_SUB) a.dir[t.slot + tid] = 0u;  // a whole tile uses sub-slice 0 only
  if (t.nrows <= 0) {
    if (tid == 0) a.dir[t.slot] = 0u;
    return;
  }
  const int row0 = t.row0, nrows = t.nrows, g = t.g;
  const long long slab = a.per_group ? (long long)g * ((a.n2 + 1) & ~1) : 0;
  const int slab_word = a.per_group ? g * a.cw : 0;
  const int col0 = t.xt * T_NT, col = col0 + tid;
  bool col_ok = col < a.n2;
  const int ncols = min(T_NT, a.n2 - col0);
  {
    const float4* src = reinterpret_cast<const float4*>(a.pre2 + slab + col0);
    float4* dst = reinterpret_cast<float4*>(s_col);
    const int n16 = (ncols * (int)sizeof(BoxPre)) / 16;
    for (int k = tid; k < n16; k += T_NT) dst[k] = src[k];
    if ((ncols & 1) && tid == 0) {
      const float2* s2 = reinterpret_cast<const float2*>(a.pre2 + slab + col0);
      reinterpret_cast<float2*>(s_col)[n16 * 2] = s2[n16 * 2];
    }
  }
  const int kw = (col0 >> 6) + wave;
  float4 cb = a.colbox[slab_word + min(kw, a.cw - 1)];
  if (kw >= a.cw) cb = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
  if (tid < nrows) {
    const BoxPre r = a.pre1 ? a.pre1[row0 + tid] : prepare_box(a.boxes1 + (long long)(row0 + tid) * a.stride1);
    s_row[tid] = r;
    if (t.xt == 0) a.pre1_out[row0 + tid] = r;   // one column tile publishes the prepared gts for at_finish
  }
  if (tid < T_WORDS) s_sm[tid] = 0ull;
  if (tid < T_TI) {
    s_rapx[tid] = 0u;
    s_rex[tid] = 0u;
  }
  if (tid == 0) {
    s_nx = 0u;
    s_nent = 0u;
  }
  if (a.valid && col_ok) col_ok = a.valid[(long long)g * a.n2 + col] != 0;
  __syncthreads();
  TTRACE(1);
  const BoxPre mine = s_col[col_ok ? tid : 0];

  // ---- detection in one pass: strip cull, then circles + separating axes for the live (row, strip) words
  bool lv = false;
  if (lane < nrows) {
    const float rx = s_row[lane].cx, ry = s_row[lane].cy;
    const float dx = fmaxf(fmaxf(cb.x - rx, rx - cb.z), 0.f), dy = fmaxf(fmaxf(cb.y - ry, ry - cb.w), 0.f);
    const float thr = 1.001f * s_row[lane].rad + 1e-5f * (fabsf(rx) + fabsf(ry));
    lv = !(dx * dx + dy * dy > thr * thr);
  }
  unsigned live = (unsigned)__ballot(lv);
  while (live) {
    const int i = __builtin_ctz(live);
    live &= live - 1u;
    bool surv = false;
    if (col_ok) {
      const BoxPre r = s_row[i];
      const float dx = r.cx - mine.cx, dy = r.cy - mine.cy;
      const float rr = r.rad + mine.rad;
      if (!(dx * dx + dy * dy > rr * rr * 1.0001f)) surv = !sat_disjoint<VERSION>(r, mine);
    }
    const unsigned long long m = __ballot(surv);
    if (m && lane == 0) s_sm[i * (T_NT / 64) + wave] = m;
  }
  __syncthreads();
  scan_mask_words<T_WORDS>(s_sm, s_send, tid);
  __syncthreads();
  TTRACE(2);
  const int total = s_send[T_WORDS - 1];
  if (total == 0) {
    if (tid == 0) a.dir[t.slot] = 0u;
    return;
  }
  AtEntry* slice = a.list + t.slot * (T_TI / T_SUB * T_NT);
  const long long gcol0 = (long long)g * a.n2 + col0;
  const int growb = row0 - t.grow0;   // row inside its group of tile row 0

  // Column accumulator update with the value known here; true when the pair is (so far) within 2 x budget of the
  // column's best value -- only such pairs can be candidates for the column maximum in at_finish, the others leave no
  // entry (the returned old value is a lower bound of the final maximum).
  auto to_column = [&](int i, int j, float v) -> bool {
    const unsigned long long old = atomicMax(a.colkey + gcol0 + j, at_pack(v, growb + i));
    return old == ~0ull || v >= __uint_as_float((unsigned)(old >> 32)) - 2.f * kFastBudget;
  };
  auto put_entry = [&](int i, int j, float v, bool exact) {
    AtEntry en;
    en.key = ((unsigned)(growb + i) << 8) | (unsigned)j | (exact ? AT_EXACT : 0u);
    en.v = v;
    slice[atomicAdd(&s_nent, 1u)] = en;
  };

  // ---- T2_KEEP rounds of 256 survivors per trip (one trip unless the tile has > T2_KEEP * 256 of them)
  float hv[T2_KEEP];
  unsigned hk[T2_KEEP];   // (i << 8) | j; T2_NONE = nothing held; T2_PEND: waits for room in the clipper's queue
  const int nround = (total + T_NT - 1) / T_NT;
  for (int base = 0; base < nround; base += T2_KEEP) {
    // -- A: tier 1 on every survivor.  Fast values stay in registers; what the fast path flags is queued
#pragma unroll
    for (int r = 0; r < T2_KEEP; ++r) {
      hk[r] = T2_NONE;
      hv[r] = 0.f;
      const int k = (base + r) * T_NT + tid;
      if (k < total) {
        int word, bit;
        locate_bit<T_WORDS>(s_sm, s_send, k, word, bit);
        const int i = word >> 2, j = ((word & 3) << 6) | bit;
        bool danger, apart;
        const float v = pair_iou_fast<VERSION>(s_row[i], s_col[j], danger, apart);
        if (!apart) {
          const unsigned ij = (unsigned)((i << 8) | j);
          if (danger || !(v >= kFastSliver)) {
            const unsigned at = atomicAdd(&s_nx, 1u);
            if (at < (unsigned)T2_XCAP) s_x[at] = (unsigned short)ij;
            else hk[r] = T2_PEND | ij;
          } else {
            hv[r] = v;
            hk[r] = ij;
            atomicMax(&s_rapx[i], __float_as_uint(v));   // best FAST value of the row in this tile
          }
        }
      }
    }
    __syncthreads();
    TTRACE(3);
    // -- B: a held fast value within 2 x budget of the row's best fast value may be the row's exact maximum: queued
    // too (the flagged pairs are clipped anyway; every other pair that could equal the exact maximum is within the
    // budget of its fast value, hence inside this window).  The rest is final: column accumulator, maybe an entry.
#pragma unroll
    for (int r = 0; r < T2_KEEP; ++r)
      if (hk[r] != T2_NONE && !(hk[r] & T2_PEND)) {
        const int i = (int)(hk[r] >> 8), j = (int)(hk[r] & 255u);
        if (hv[r] >= __uint_as_float(s_rapx[i]) - 2.f * kFastBudget) {
          const unsigned at = atomicAdd(&s_nx, 1u);
          if (at < (unsigned)T2_XCAP) {
            s_x[at] = (unsigned short)hk[r];
            hk[r] = T2_NONE;
          } else {
            hk[r] |= T2_PEND;
          }
        } else {
          if (to_column(i, j, hv[r])) put_entry(i, j, hv[r], false);
          hk[r] = T2_NONE;
        }
      }
    // -- C / D: the queue through the reference-order clipper, in chunks of T2_XCAP while something waits for room
    for (;;) {
      __syncthreads();
      const int nq = min((int)s_nx, T2_XCAP);
      const bool over = (int)s_nx > T2_XCAP;
      if (wave < T2_QUADS / 16) {
        F2* qscr = s_pts + (tid >> 2) * kQuadSlots;
        for (int q0 = 0; q0 < nq; q0 += T2_QUADS) {
          if (q0 + wave * 16 >= nq) break;               // wave-uniform: the clipper uses wave ballots only
          const int q = q0 + (tid >> 2);
          const bool on = q < nq;
          const unsigned e = s_x[on ? q : q0];
          const int i = (int)(e >> 8), j = (int)(e & 255u);
          const float v = pair_iou_quad<VERSION>(s_row[i], s_col[j], qscr, lane);
          if (on && (tid & 3) == 0) {
            unsigned keep = 0u;   // bit 0: positive value, bit 1: candidate for its column's maximum
            if (v > 0.f) {
              atomicMax(&s_rex[i], __float_as_uint(v));
              keep = 1u | (to_column(i, j, v) ? 2u : 0u);
            } else if (v != v && growb + i == 0) {
              atomicMax(a.colkey + gcol0 + j, ~0ull);   // a NaN in the image's first row sticks (assigner.py:133)
            }
            s_xv[q] = v;
            s_xk[q] = (unsigned char)keep;
          }
          lds_wave_order();
        }
      }
      __syncthreads();
      // entries of the clipped pairs: column candidates, and the pairs that equal the row's exact maximum so far
      // (low-quality rule, assigner.py:151-160; at_finish repeats the comparison against the global maximum)
      for (int q = tid; q < nq; q += T_NT) {
        const unsigned e = s_x[q];
        const int i = (int)(e >> 8), j = (int)(e & 255u);
        const float v = s_xv[q];
        const unsigned keep = s_xk[q];
        if ((keep & 1u) && ((keep & 2u) || __float_as_uint(v) == s_rex[i])) put_entry(i, j, v, true);
      }
      __syncthreads();
      if (tid == 0) s_nx = 0u;
      if (!over) break;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < T2_KEEP; ++r)
        if (hk[r] != T2_NONE) {      // only T2_PEND items are still held here
          const unsigned at = atomicAdd(&s_nx, 1u);
          if (at < (unsigned)T2_XCAP) {
            s_x[at] = (unsigned short)(hk[r] & 0xFFFFu);
            hk[r] = T2_NONE;
          }
        }
    }
    __syncthreads();
  }
  TTRACE(4);
  // ---- the exact row maxima of this tile, the entry count
  if (tid < nrows && s_rex[tid] != 0u) atomicMax(a.rowmax + row0 + tid, s_rex[tid]);
  if (tid == 0) a.dir[t.slot] = s_nent;
  TTRACE(5);
}

// ---- at_finish: one workgroup per (column tile, image): low-quality rule + the targets ---------------------------
struct FinishArgs {
  const float* boxes1;  // raw gts
  int stride1;
  const float* boxes2;  // raw anchors (n2, stride2), slab g at g * n2 * stride2 when per_group
  int stride2, n2, per_group;
  const int* row_offsets;
  const int* group_tile0;  // n_groups + 1: first row tile of every group (nullptr: g * ny)
  int n_row_tiles, ny, nx, n_groups, n1;
  const unsigned* rowmax;
  const AtEntry* list;
  const unsigned* dir;
  unsigned long long* colkey;  // read, then cleared for the next call
  const unsigned char* valid;
  const int* gt_labels;  // optional
  float pos_thr, neg_lo, neg_hi, min_pos_iou;
  int match_low_quality, labels_filled, reg_decoded;
  float pos_weight;  // <= 0: 1
  F5 mean, stdv;
  int* gt_inds;
  float* max_ov;
  int* labels;
  float* label_weights;
  float* bbox_targets;
  float* bbox_weights;
  float* totals;        // [0] = sum_g max(#pos_g, 1), [1] = sum_g max(#neg_g, 1)
  unsigned* state;      // 64-bit words: [0] batch, [1 + g] image g = packed {pos, neg, arrivals}: zero on entry / exit
  unsigned* rowmax_rw;  // same array as rowmax: cleared by the last workgroup
  // two-tier mode: the prepared boxes for the clipper on the undecided columns
  const BoxPre* pre1;   // prepared gts (n1), written by at_tile2_kernel
  const BoxPre* pre2;   // prepared anchors, slab g at g * pitch when per_group
};

// MODE 0: every entry is an exact value that equals its tile's row maximum (iou_tile_kernel<.., SPARSE>).
// MODE 1 / 2 (two-tier, box convention 0 / 1): every surviving pair is an entry, exact or fast (AT_EXACT); the columns
// whose decision could depend on the difference are settled here with the reference-order clipper.
template <int MODE>
__global__ __launch_bounds__(T_NT) void at_finish_kernel(const FinishArgs a) {
  constexpr int RM_CAP = 2048;  // row maxima of the image staged in LDS (larger images gather them from global)
  constexpr int FX_CAP = MODE ? 1024 : 1;   // undecided pairs queued for the clipper (drained when nearly full)
  __shared__ unsigned s_cmax[MODE ? T_NT : 1];              // the column's best value so far (float bits)
  __shared__ unsigned s_ncont[MODE ? T_NT : 1];             // entries within 2 x budget of it | 0x10000 if a fast one is among them
  __shared__ unsigned long long s_exkey[MODE ? T_NT : 1];   // exact accumulator of an undecided column
  __shared__ unsigned char s_unc[MODE ? T_NT : 1];
  __shared__ unsigned s_fx[FX_CAP];                         // (row in group << 8) | column in tile
  __shared__ unsigned s_nfx;
  __shared__ F2 s_pts[MODE ? kQuadSlots * 16 : 1];
  __shared__ int s_lowq[T_NT];
  __shared__ unsigned s_end[T_NT], s_wsum[T_NT / 64];
  __shared__ unsigned s_rm[RM_CAP];
  __shared__ int s_zmax;
  __shared__ int s_cnt[2];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int xt = blockIdx.x, g = blockIdx.y;
  const int col = xt * T_NT + tid;
  FTRACE(0);
  // The kernel is a chain of dependent global round trips on 344 workgroups (1.3 per CU): everything that does not
  // depend on the entries is requested up front -- this column's accumulator, validity flag and anchor.
  const long long o = (long long)g * a.n2 + min(col, a.n2 - 1);
  const unsigned long long key_in = a.colkey[o];  // the tile kernel's atomics; complete at the kernel boundary
  const unsigned char valid_in = a.valid ? a.valid[o] : (unsigned char)1;
  float an[5] = {0.f, 0.f, 1.f, 1.f, 0.f};
  if (a.bbox_targets && !a.reg_decoded) {
    const float* ap = a.boxes2 + ((a.per_group ? (long long)g * a.n2 : 0) + min(col, a.n2 - 1)) * a.stride2;
#pragma unroll
    for (int k = 0; k < 5; ++k) an[k] = ap[k];
  }
  const int r0 = a.row_offsets[g], K = a.row_offsets[g + 1] - r0;
  const int t0 = a.group_tile0 ? a.group_tile0[g] : g * a.ny;
  const int t1 = a.group_tile0 ? a.group_tile0[g + 1] : g * a.ny + (K + T_TI - 1) / T_TI;
  s_lowq[tid] = -1;
  if (tid == 0) {
    s_zmax = -1;
    s_cnt[0] = 0;
    s_cnt[1] = 0;
  }
  __syncthreads();
  FTRACE(1);
  if (MODE) {
    s_cmax[tid] = (key_in != 0ull && key_in != ~0ull) ? (unsigned)(key_in >> 32) : 0u;
    s_ncont[tid] = 0u;
    s_exkey[tid] = 0ull;
    s_unc[tid] = 0;
    if (tid == 0) s_nfx = 0u;
  }
  // The sub-slices {count} this column tile received from the image's row tiles: all counts are fetched at once (one
  // per thread), scanned in LDS, and the entries walked as ONE flat range of independent loads.  `visit(row, j, v,
  // exact)` per entry; `after()` once per 256 entries, by every thread (it may hold barriers).
  const size_t slot0 = ((size_t)xt * a.n_row_tiles + t0) * T_SUB;
  const int n_slots = (t1 - t0) * T_SUB;
  auto scan_entries = [&](auto visit, auto after) {
    for (int sb = 0; sb < n_slots; sb += T_NT) {
      const int ns = min(T_NT, n_slots - sb);
      unsigned inc = tid < ns ? a.dir[slot0 + sb + tid] : 0u;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o2 = __shfl_up(inc, off);
        if (lane >= off) inc += o2;
      }
      if (lane == 63) s_wsum[tid >> 6] = inc;
      __syncthreads();
      unsigned add = 0;
      for (int w = 0; w < (tid >> 6); ++w) add += s_wsum[w];
      s_end[tid] = inc + add;
      __syncthreads();
      const unsigned total = s_end[T_NT - 1];
      for (unsigned e0 = 0; e0 < total; e0 += T_NT) {
        const unsigned e = e0 + tid;
        if (e < total) {
          int lo = 0;  // first sub-slice whose inclusive end exceeds e
#pragma unroll
          for (int step = T_NT / 2; step > 0; step >>= 1)
            if (s_end[lo + step - 1] <= e) lo += step;
          const unsigned before = lo ? s_end[lo - 1] : 0u;
          const AtEntry en = a.list[(slot0 + sb + lo) * (T_TI / T_SUB * T_NT) + (e - before)];
          visit((int)((en.key & ~AT_EXACT) >> 8), (int)(en.key & 255u), en.v, MODE == 0 || (en.key & AT_EXACT) != 0u);
        }
        after();
      }
      __syncthreads();
    }
  };
  if (K > 0 && (a.match_low_quality || MODE)) {
    // gts that overlap nothing: their row maximum is 0 and every anchor "equals" it (assigner.py:155)
    const bool zero_rows = a.match_low_quality && 0.0f >= a.min_pos_iou;
    for (int r = tid; r < K; r += T_NT) {
      const unsigned rm = a.rowmax[r0 + r];
      if (r < RM_CAP) s_rm[r] = rm;
      if (zero_rows && rm == 0u) atomicMax(&s_zmax, r);
    }
    if (MODE) __syncthreads();   // s_cmax is read by other lanes below
    scan_entries(
        [&](int row, int j, float v, bool exact) {
          if (a.match_low_quality && exact) {
            const float rm = __uint_as_float(row < RM_CAP ? s_rm[row] : a.rowmax[r0 + row]);
            if (rm >= a.min_pos_iou && v == rm) atomicMax(&s_lowq[j], row);
          }
          if (MODE && v >= __uint_as_float(s_cmax[j]) - 2.f * kFastBudget) atomicAdd(&s_ncont[j], exact ? 1u : 0x10001u);
        },
        [] {});
  }
  __syncthreads();
  unsigned long long key_final = key_in;
  if (MODE && K > 0) {
    // ---- undecided columns: a fast value among the candidates for the column maximum and either a second candidate
    // (maximum / first argmax open) or a threshold within the budget of the best value (side of the threshold open)
    const unsigned nc = s_ncont[tid];
    const float m = __uint_as_float(s_cmax[tid]);
    const bool near_thr = fabsf(m - a.pos_thr) <= kFastBudget || fabsf(m - a.neg_hi) <= kFastBudget ||
                          (a.neg_lo > 0.f && fabsf(m - a.neg_lo) <= kFastBudget);
    const bool unc = col < a.n2 && valid_in && key_in != 0ull && key_in != ~0ull && (nc & 0xFFFF0000u) != 0u &&
                     ((nc & 0xFFFFu) >= 2u || near_thr);
    s_unc[tid] = unc ? 1 : 0;
    if (__syncthreads_or(unc ? 1 : 0)) {
      const BoxPre* p2 = a.pre2 + (a.per_group ? (long long)g * ((a.n2 + 1) & ~1) : 0) + (long long)xt * T_NT;
      auto drain = [&]() {   // the queued pairs through the reference-order clipper (first wave, 16 quads)
        const int n = (int)s_nfx;
        if (tid < 64) {
          F2* qscr = s_pts + (tid >> 2) * kQuadSlots;
          for (int q0 = 0; q0 < n; q0 += 16) {
            const int q = q0 + (tid >> 2);
            const bool on = q < n;
            const unsigned e = s_fx[on ? q : q0];
            const int row = (int)(e >> 8), j = (int)(e & 255u);
            const BoxPre ra = a.pre1[r0 + row], cbx = p2[j];
            const float v = MODE == 1 ? pair_iou_quad<0>(ra, cbx, qscr, lane) : pair_iou_quad<1>(ra, cbx, qscr, lane);
            if (on && (tid & 3) == 0) {
              if (v > 0.f) atomicMax(&s_exkey[j], at_pack(v, row));
              else if (v != v && row == 0) atomicMax(&s_exkey[j], ~0ull);
            }
            lds_wave_order();
          }
        }
        __syncthreads();
        if (tid == 0) s_nfx = 0u;
        __syncthreads();
      };
      scan_entries(
          [&](int row, int j, float v, bool exact) {
            if (s_unc[j] && v >= __uint_as_float(s_cmax[j]) - 2.f * kFastBudget) {
              if (exact) atomicMax(&s_exkey[j], at_pack(v, row));
              else s_fx[atomicAdd(&s_nfx, 1u)] = ((unsigned)row << 8) | (unsigned)j;
            }
          },
          [&] {
            __syncthreads();
            if ((int)s_nfx > FX_CAP - T_NT) drain();   // uniform: read after the barrier
          });
      __syncthreads();
      if (s_nfx != 0u) drain();
      if (unc) key_final = s_exkey[tid];
    }
  }
  __syncthreads();
  FTRACE(2);
  bool pos = false, neg = false;
  if (col < a.n2) {
    int gi;
    float best;
    if (K <= 0) {  // no gts: everything negative (the reference raises; the batched form keeps going)
      gi = 0;
      best = 0.f;
    } else if (!valid_in) {
      gi = -1;
      best = -1.f;
    } else {
      const unsigned long long key = key_final;
      if (key_in != 0ull) a.colkey[o] = 0ull;  // empty again for the next call
      int arg = 0;
      best = 0.f;  // key 0: no positive entry -> the column holds zeros only: maximum 0 at the first row
      if (key != 0ull) {
        best = __uint_as_float((unsigned)(key >> 32));
        arg = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      }
      gi = -1;
      if (best >= a.neg_lo && best < a.neg_hi) gi = 0;  // assigner.py:138-145
      if (best >= a.pos_thr) gi = arg + 1;              // :147-148
      const int lowq = max(s_lowq[tid], s_zmax);
      if (lowq >= 0) gi = lowq + 1;                     // :151-158
    }
    pos = gi > 0;
    neg = gi == 0;
    if (a.gt_inds) a.gt_inds[o] = gi;
    if (a.max_ov) a.max_ov[o] = best;
    if (a.labels) a.labels[o] = pos ? (a.gt_labels ? a.gt_labels[r0 + gi - 1] : 1) : a.labels_filled;
    if (a.label_weights) a.label_weights[o] = neg ? 1.0f : (pos ? (a.pos_weight <= 0.f ? 1.0f : a.pos_weight) : 0.0f);
    if (a.bbox_targets) {
      float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      if (pos) {
        const float* gt = a.boxes1 + (long long)(r0 + gi - 1) * a.stride1;
        if (a.reg_decoded) {
#pragma unroll
          for (int k = 0; k < 5; ++k) t[k] = gt[k];
        } else {
          encode_one(an, gt, a.mean, a.stdv, t);
        }
      }
      const float wv = pos ? 1.0f : 0.0f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        a.bbox_targets[o * 5 + k] = t[k];
        if (a.bbox_weights) a.bbox_weights[o * 5 + k] = wv;
      }
    }
  }
  FTRACE(3);
  // ---- counts: sum_img max(#pos, 1) and the same for negatives (anchor_target.py:79-80).  One returning 64-bit
  // atomic per workgroup carries its counts AND its arrival (pos : 24 | neg : 24 | arrivals : 16): the workgroup
  // whose add completes the image holds the image's totals in the returned value -- no second counter to order
  // against, no wait for acknowledgements.  The same again over the images (26 | 26 | 12).
  const int np = __popcll(__ballot(pos)), nn = __popcll(__ballot(neg));
  if (lane == 0) {
    if (np) atomicAdd(&s_cnt[0], np);
    if (nn) atomicAdd(&s_cnt[1], nn);
  }
  __syncthreads();
  if (tid == 0) {
    unsigned long long* words = reinterpret_cast<unsigned long long*>(a.state);
    const unsigned long long mine = ((unsigned long long)(unsigned)s_cnt[0] << 40) |
                                    ((unsigned long long)(unsigned)s_cnt[1] << 16) | 1ull;
    // Arrival.  What the LAST arriver does afterwards (clearing rowmax) only has to come after every other workgroup's
    // READS of rowmax; those reads feed `mine` (the counts depend on the low-quality matches, which depend on the row
    // maxima), so each workgroup's arrival atomic cannot issue before its reads have returned -- a data dependency, not a
    // fence.  The compiler barrier keeps the atomic below the output stores in program order.  (An agent-scope acq_rel
    // arrival -- one buffer_wbl2 + buffer_inv per workgroup -- was measured in round 3: at_finish 11.8 -> 20.7 us.)
    asm volatile("" ::: "memory");
    const unsigned long long now = atomicAdd(words + 1 + g, mine) + mine;
    int last = 0;
    if ((now & 0xFFFFull) == (unsigned long long)gridDim.x) {  // this image is complete
      atomicExch(words + 1 + g, 0ull);
      const unsigned long long p = now >> 40, n = (now >> 16) & 0xFFFFFFull;
      const unsigned long long img = ((p ? p : 1ull) << 38) | ((n ? n : 1ull) << 12) | 1ull;
      const unsigned long long all = atomicAdd(words, img) + img;
      if ((all & 0xFFFull) == (unsigned long long)gridDim.y) {  // ... and so is the batch
        atomicExch(words, 0ull);
        if (a.totals) {
          a.totals[0] = (float)(all >> 38);
          a.totals[1] = (float)((all >> 12) & 0x3FFFFFFull);
        }
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  FTRACE(4);
  if (!s_last) return;
  // ---- the last workgroup: the row maxima go back to zero for the next call (every workgroup has read them before
  // its arrival above; the counters are only ever touched by device-scope atomics, so no fence is needed anywhere)
  for (int r = tid; r < a.n1; r += T_NT) a.rowmax_rw[r] = 0u;
}

}  // namespace rsdet

using namespace rsdet;

#ifdef RSDET_TILE_TRACE
extern "C" void rsdet_debug_set_tile_trace(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace), &p, sizeof(p)); }
#endif

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
static inline long long slab_pitch(int n_per_group) { return ((long long)n_per_group + 1) & ~1LL; }

// ---- prepared column sets (cacheable by the caller: the FAM anchor grid never changes) -------------------------
extern "C" size_t rsdet_iou_prepared_bytes(long long n_total, int n_per_group) {
  if (n_total <= 0 || n_per_group <= 0) return 0;
  const long long cw = (n_per_group + 63) / 64, groups = n_total / n_per_group;
  return up256((size_t)(groups * slab_pitch(n_per_group)) * sizeof(BoxPre)) + up256((size_t)(groups * cw) * 16);
}

extern "C" int rsdet_iou_prepare_f32(const float* boxes, long long n_total, int n_per_group, int stride,
                                     void* prepared, size_t prepared_bytes, void* stream) {
  if (n_total < 0 || n_per_group <= 0 || stride < 5 || n_total % n_per_group) return RSDET_EINVAL;
  if (n_total == 0) return RSDET_OK;
  if (!boxes || !prepared || ((uintptr_t)prepared & 15) || prepared_bytes < rsdet_iou_prepared_bytes(n_total, n_per_group))
    return RSDET_EINVAL;
  const int cw = (n_per_group + 63) / 64;
  const long long groups = n_total / n_per_group;
  BoxPre* pre = (BoxPre*)prepared;
  float4* colbox = (float4*)((char*)prepared + up256((size_t)(groups * slab_pitch(n_per_group)) * sizeof(BoxPre)));
  hipLaunchKernelGGL(at_prepare_kernel, dim3((unsigned)(groups * cw)), dim3(64), 0, (hipStream_t)stream, boxes,
                     n_per_group, stride, cw, (int)slab_pitch(n_per_group), pre, colbox);
  return rsdet_launch_status();
}

static inline void split_prepared(const void* prepared, long long groups, int n2, const BoxPre** pre,
                                  const float4** colbox) {
  *pre = (const BoxPre*)prepared;
  *colbox = (const float4*)((const char*)prepared + up256((size_t)(groups * slab_pitch(n2)) * sizeof(BoxPre)));
}

// first column tile that is cut into row sub-tiles (hint: columns >= heavy_from_col hold large boxes)
static inline int split_tile(int heavy_from_col, int n2, int nx) {
  if (heavy_from_col < 0 || heavy_from_col >= n2) return nx;
  return heavy_from_col / T_NT;
}
static inline long long tile_grid(int nx, int split_xt, int nrt) {
  return (long long)split_xt * nrt + (long long)(nx - split_xt) * nrt * T_SUB;
}

// ---- dense IoU in one launch -------------------------------------------------------------------------------------
extern "C" int rsdet_box_iou_rotated_tiled_f32(const float* boxes1, int n1, int stride1, const int* row_offsets,
                                               int n_groups, int max_rows_per_group, const int* tile_table,
                                               int n_row_tiles, const void* prepared1, const void* prepared2, int n2,
                                               int per_group, int heavy_from_col, int version, float* ious,
                                               void* stream) {
  if (n1 < 0 || n2 < 0 || n_groups < 1 || stride1 < 5 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;
  if (!boxes1 || !prepared2 || !ious) return RSDET_EINVAL;
  if (!row_offsets && n_groups != 1) return RSDET_EINVAL;
  TileArgs a{};
  a.boxes1 = boxes1, a.stride1 = stride1, a.n1 = n1;
  a.pre1 = (const BoxPre*)prepared1;
  split_prepared(prepared2, per_group ? n_groups : 1, n2, &a.pre2, &a.colbox);
  a.n2 = n2, a.cw = (n2 + 63) / 64, a.per_group = per_group ? 1 : 0;
  a.row_offsets = row_offsets;
  a.tiles = (const TileDesc*)tile_table;
  a.ny = (max_rows_per_group + T_TI - 1) / T_TI;
  a.n_row_tiles = tile_table ? n_row_tiles : n_groups * a.ny;
  a.nx = (n2 + T_NT - 1) / T_NT;
  a.split_xt = split_tile(heavy_from_col, n2, a.nx);
  a.out = ious;
  if (a.n_row_tiles <= 0) return RSDET_OK;
  const dim3 grid((unsigned)tile_grid(a.nx, a.split_xt, a.n_row_tiles));
  if (version == 0)
    hipLaunchKernelGGL((iou_tile_kernel<0, 0>), grid, dim3(T_NT), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((iou_tile_kernel<1, 0>), grid, dim3(T_NT), 0, (hipStream_t)stream, a);
  return rsdet_launch_status();
}

// ---- dense IoU in two launches (detect | fill + balanced clip) -------------------------------------------------------
// state (zero on entry, zero again on exit): IOU2_SHARDS counters on 128-byte lines | done counter
// ws: pre1 (n1 prepared gts) | tile masks (slots x 64 words) | tile flags (slots) | queue (IOU2_SHARDS x cap items)
static inline long long split_queue_cap(long long n1, long long n2) {
  const long long pairs = n1 * n2, total = pairs < (4LL << 20) ? pairs : (4LL << 20);
  const long long per = (total + IOU2_SHARDS - 1) / IOU2_SHARDS, tile = (long long)T_TI * T_NT;
  return per < tile ? (pairs < tile ? pairs : tile) : per;
}

extern "C" size_t rsdet_box_iou_rotated_split_state_bytes(void) { return 2 * IOU2_SHARDS * 128 + 256; }

extern "C" size_t rsdet_box_iou_rotated_split_ws_size(int n1, int n2, int n_row_tiles) {
  if (n1 <= 0 || n2 <= 0 || n_row_tiles <= 0) return 0;
  const size_t slots = (size_t)((n2 + T_NT - 1) / T_NT) * n_row_tiles * T_SUB;
  return up256((size_t)n1 * sizeof(BoxPre)) + up256(slots * T_WORDS * 8) + up256(slots * 4) +
         (size_t)split_queue_cap(n1, n2) * IOU2_SHARDS * sizeof(WorkItem);
}

extern "C" int rsdet_box_iou_rotated_split_f32(const float* boxes1, int n1, int stride1, const int* row_offsets,
                                               int n_groups, int max_rows_per_group, const int* tile_table,
                                               int n_row_tiles, const void* prepared1, const void* prepared2, int n2,
                                               int per_group, int heavy_from_col, int version, float* ious,
                                               void* state, size_t state_bytes, void* ws, size_t ws_bytes,
                                               void* stream) {
  if (n1 < 0 || n2 < 0 || n_groups < 1 || stride1 < 5 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;
  if (!boxes1 || !prepared2 || !ious || !state || !ws || ((uintptr_t)ws & 15) || ((uintptr_t)state & 15))
    return RSDET_EINVAL;
  if (!row_offsets && n_groups != 1) return RSDET_EINVAL;
  TileArgs a{};
  a.boxes1 = boxes1, a.stride1 = stride1, a.n1 = n1;
  a.pre1 = (const BoxPre*)prepared1;
  split_prepared(prepared2, per_group ? n_groups : 1, n2, &a.pre2, &a.colbox);
  a.n2 = n2, a.cw = (n2 + 63) / 64, a.per_group = per_group ? 1 : 0;
  a.row_offsets = row_offsets;
  a.tiles = (const TileDesc*)tile_table;
  a.ny = (max_rows_per_group + T_TI - 1) / T_TI;
  a.n_row_tiles = tile_table ? n_row_tiles : n_groups * a.ny;
  a.nx = (n2 + T_NT - 1) / T_NT;
  a.split_xt = split_tile(heavy_from_col, n2, a.nx);
  a.out = ious;
  if (a.n_row_tiles <= 0) return RSDET_OK;
  if (state_bytes < rsdet_box_iou_rotated_split_state_bytes() ||
      ws_bytes < rsdet_box_iou_rotated_split_ws_size(n1, n2, a.n_row_tiles))
    return RSDET_EINVAL;
  const size_t slots = (size_t)a.nx * a.n_row_tiles * T_SUB;
  char* w = (char*)ws;
  a.pre1_out = (BoxPre*)w;
  w += up256((size_t)n1 * sizeof(BoxPre));
  a.tilemask = (unsigned long long*)w;
  w += up256(slots * T_WORDS * 8);
  a.tileflag = (unsigned*)w;
  w += up256(slots * 4);
  a.queue = (WorkItem*)w;
  a.counter = (unsigned*)state;
  a.capacity = (unsigned)split_queue_cap(n1, n2);
  unsigned* done = (unsigned*)((char*)state + IOU2_SHARDS * 128);
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (int)tile_grid(a.nx, a.split_xt, a.n_row_tiles);
  long long need = ((long long)a.capacity * IOU2_SHARDS + T_NT / 4 - 1) / (T_NT / 4);
  const int clip_blocks = (int)(need < 2048 ? (need < 1 ? 1 : need) : 2048);
  if (version == 0) {
    hipLaunchKernelGGL(iou_detect_kernel<0>, dim3(nblk), dim3(T_NT), 0, s, a);
    hipLaunchKernelGGL(iou_clip_fill_kernel<0>, dim3(clip_blocks), dim3(T_NT), 0, s, a, nblk, done);
  } else {
    hipLaunchKernelGGL(iou_detect_kernel<1>, dim3(nblk), dim3(T_NT), 0, s, a);
    hipLaunchKernelGGL(iou_clip_fill_kernel<1>, dim3(clip_blocks), dim3(T_NT), 0, s, a, nblk, done);
  }
  return rsdet_launch_status();
}

// ---- fused anchor targets ------------------------------------------------------------------------------------------
// state (zero on entry, zero again on exit): [0] finished workgroups | pos/neg counts (2 per group) |
//                                            rowmax (n1 words) | colkey (n_groups * n2 u64)
// ws (scratch, no initialisation needed):     dir (nx * n_row_tiles * T_SUB words) |
//                                            list (nx * n_row_tiles slices of T_TI * T_NT entries)
static inline size_t at_counters_bytes(int n_groups) { return up256((1 + (size_t)n_groups) * 8); }

extern "C" size_t rsdet_anchor_target_rotated_state_bytes(int n1, int n2, int n_groups) {
  if (n_groups <= 0 || n2 <= 0) return 0;
  return at_counters_bytes(n_groups) + up256((size_t)(n1 > 0 ? n1 : 1) * 4) + up256((size_t)n_groups * n2 * 8);
}

extern "C" size_t rsdet_anchor_target_rotated_ws_size(int n2, int n_groups, int n_row_tiles) {
  if (n2 <= 0 || n_row_tiles <= 0 || n_groups <= 0) return 0;
  const size_t nx = (size_t)(n2 + T_NT - 1) / T_NT;
  // directory | entry slices | prepared gts of the two-tier mode (at most T_TI rows per row tile)
  return up256(nx * (size_t)n_row_tiles * T_SUB * 4) + up256(nx * (size_t)n_row_tiles * (T_TI * T_NT) * sizeof(AtEntry)) +
         up256((size_t)n_row_tiles * T_TI * sizeof(BoxPre));
}

extern "C" int rsdet_anchor_target_rotated_f32(
    const float* gt_boxes, int n1, int stride1, const int* gt_labels, const int* row_offsets, int n_groups,
    int max_rows_per_group, const int* tile_table, int n_row_tiles, const int* group_tile0, const float* anchors,
    int n2, int stride2, int per_group, const void* prepared2, const void* prepared_gt, int heavy_from_col,
    const unsigned char* valid, int version, int two_tier, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi,
    float min_pos_iou, int match_low_quality, int labels_filled, float pos_weight, int reg_decoded_bbox,
    const float* means_host, const float* stds_host, int* gt_inds, float* max_overlaps, int* labels,
    float* label_weights, float* bbox_targets, float* bbox_weights, float* totals, void* state, size_t state_bytes,
    void* ws, size_t ws_bytes, void* stream) {
  if (n1 < 0 || n2 <= 0 || n_groups < 1 || stride1 < 5 || stride2 < 5 || (version != 0 && version != 1))
    return RSDET_EINVAL;
  // packed counters of at_finish: 24 bits of anchors per image, 16 of column tiles, 12 of images, 26 of anchors in all
  if (n2 >= (1 << 24) || n_groups >= (1 << 12) || (long long)n2 * n_groups >= (1LL << 26)) return RSDET_EINVAL;
  if (!row_offsets || !anchors || !prepared2 || !ws || ((uintptr_t)ws & 15) || !state || ((uintptr_t)state & 15))
    return RSDET_EINVAL;
  if (n1 > 0 && !gt_boxes) return RSDET_EINVAL;
  if ((tile_table == nullptr) != (group_tile0 == nullptr)) return RSDET_EINVAL;
  const int ny = (max_rows_per_group + T_TI - 1) / T_TI;
  const int nrt = tile_table ? n_row_tiles : n_groups * ny;
  if (nrt < 0) return RSDET_EINVAL;
  const int nrt1 = nrt > 0 ? nrt : 1;
  if (ws_bytes < rsdet_anchor_target_rotated_ws_size(n2, n_groups, nrt1)) return RSDET_EINVAL;
  if (state_bytes < rsdet_anchor_target_rotated_state_bytes(n1, n2, n_groups)) return RSDET_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nx = (n2 + T_NT - 1) / T_NT;
  unsigned* st = (unsigned*)state;
  unsigned* rowmax = (unsigned*)((char*)state + at_counters_bytes(n_groups));
  unsigned long long* colkey = (unsigned long long*)((char*)rowmax + up256((size_t)(n1 > 0 ? n1 : 1) * 4));
  unsigned* dir = (unsigned*)ws;
  AtEntry* list = (AtEntry*)((char*)ws + up256((size_t)nx * nrt1 * T_SUB * 4));
  BoxPre* pre1_out = (BoxPre*)((char*)list + up256((size_t)nx * nrt1 * (T_TI * T_NT) * sizeof(AtEntry)));
  if (n1 > (long long)nrt1 * T_TI) return RSDET_EINVAL;   // the row tiles cover every gt

  if (n1 > 0 && nrt > 0) {
    TileArgs a{};
    a.boxes1 = gt_boxes, a.stride1 = stride1, a.n1 = n1;
    a.pre1 = (const BoxPre*)prepared_gt;
    split_prepared(prepared2, per_group ? n_groups : 1, n2, &a.pre2, &a.colbox);
    a.n2 = n2, a.cw = (n2 + 63) / 64, a.per_group = per_group ? 1 : 0;
    a.row_offsets = row_offsets;
    a.tiles = (const TileDesc*)tile_table;
    a.ny = ny, a.n_row_tiles = nrt, a.nx = nx;
    a.split_xt = split_tile(heavy_from_col, n2, nx);
    a.valid = valid, a.rowmax = rowmax, a.list = list, a.dir = dir, a.colkey = colkey;
    a.pre1_out = pre1_out;
    const dim3 grid((unsigned)tile_grid(nx, a.split_xt, nrt));
    if (two_tier) {
      if (version == 0)
        hipLaunchKernelGGL(at_tile2_kernel<0>, grid, dim3(T_NT), 0, s, a);
      else
        hipLaunchKernelGGL(at_tile2_kernel<1>, grid, dim3(T_NT), 0, s, a);
    } else if (version == 0) {
      hipLaunchKernelGGL((iou_tile_kernel<0, 1>), grid, dim3(T_NT), 0, s, a);
    } else {
      hipLaunchKernelGGL((iou_tile_kernel<1, 1>), grid, dim3(T_NT), 0, s, a);
    }
  }
  FinishArgs f{};
  const float4* f_colbox_unused = nullptr;
  f.boxes1 = gt_boxes, f.stride1 = stride1, f.boxes2 = anchors, f.stride2 = stride2, f.n2 = n2;
  f.per_group = per_group ? 1 : 0, f.row_offsets = row_offsets, f.group_tile0 = group_tile0;
  f.n_row_tiles = nrt1, f.ny = ny, f.nx = nx, f.n_groups = n_groups, f.n1 = n1;
  f.rowmax = rowmax, f.list = list, f.dir = dir, f.colkey = colkey, f.valid = valid;
  f.gt_labels = gt_labels, f.pos_thr = pos_iou_thr, f.neg_lo = neg_iou_lo, f.neg_hi = neg_iou_hi;
  f.min_pos_iou = min_pos_iou, f.match_low_quality = match_low_quality, f.labels_filled = labels_filled;
  f.reg_decoded = reg_decoded_bbox, f.pos_weight = pos_weight;
  f.mean = load5(means_host, 0.f), f.stdv = load5(stds_host, 1.f);
  f.gt_inds = gt_inds, f.max_ov = max_overlaps, f.labels = labels, f.label_weights = label_weights;
  f.bbox_targets = bbox_targets, f.bbox_weights = bbox_weights, f.totals = totals;
  f.state = st, f.rowmax_rw = rowmax;
  f.pre1 = pre1_out;
  split_prepared(prepared2, per_group ? n_groups : 1, n2, &f.pre2, &f_colbox_unused);
  if (!two_tier)
    hipLaunchKernelGGL(at_finish_kernel<0>, dim3(nx, n_groups), dim3(T_NT), 0, s, f);
  else if (version == 0)
    hipLaunchKernelGGL(at_finish_kernel<1>, dim3(nx, n_groups), dim3(T_NT), 0, s, f);
  else
    hipLaunchKernelGGL(at_finish_kernel<2>, dim3(nx, n_groups), dim3(T_NT), 0, s, f);
  return rsdet_launch_status();
}
