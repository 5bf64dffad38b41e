// iou_fast.hip -- dense rotated IoU in ONE launch with a two-tier clipper (gfx950 / CDNA4).
//
// Replaces, to the north star's tolerance (|IoU - reference| <= 1e-4; measured < 3e-6), the pair loop of
//   /root/reference/python/jdet/ops/box_iou_rotated.py:487-500 (box_iou_rotated) and box_iou_rotated_v1.py:507-524.
// The bit-exact form of the same op stays in box_iou_rotated.hip (three launches, reference-order clipper on every
// overlapping pair, 37 us at the S2ANet step shape); this one is for callers that need the VALUES only:
//
//   tile       32 gts x 256 anchors per workgroup, row tiles from a host-built table (no empty workgroups)
//   fill       first thing in the kernel: the tile's zeros as 16-byte-per-lane row-contiguous stores (a wave writes
//              1 KB of one matrix row per instruction, 8 instructions per lane), never waited for while detecting
//   detect     strip cull against the 64-column boxes -> bounding circles -> separating axes on dense lanes; the
//              candidate / survivor sets are bit masks (rsdet_tile.h)
//   tier 1     every survivor: intersection area by Green's theorem, one lane per pair, registers only, ~350 VALU
//              instructions (rsdet_geom_fast.h) -- against ~3 200 lane-instructions of the reference-order clipper
//   tier 2     the survivors tier 1 flags -- a corner of one box within 0.01 px of an edge of the other (where the
//              REFERENCE leaves the true area, see rsdet_geom_fast.h), IoU < 3e-5 (exact zeros), NaN -- go through
//              the reference-order clipper (rsdet_geom.h, 4 lanes per pair) on the first wave: ~1.2 % of the survivors
//              of random boxes
// Values are stored after the workgroup's own zero stores have been acknowledged (s_waitcnt vmcnt(0) + barrier: same
// CU, same L2 channel), so no element depends on the order of two stores in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "rsdet_api_internal.h"
#include "rsdet_geom_fast.h"
#include "rsdet_tile.h"

namespace rsdet {

#ifdef RSDET_FAST_TRACE  // debug builds only (profiles/scripts/trace_fast.py): per-workgroup stage timestamps, 100 MHz
__device__ unsigned long long* g_fast_trace;
#define FTR(k)                                                                                              \
  do {                                                                                                      \
    if (threadIdx.x == 0 && g_fast_trace) g_fast_trace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64();      \
  } while (0)
#else
#define FTR(k)
#endif

constexpr int F_NT = 256;     // columns per tile = threads per workgroup
constexpr int F_R = 32;       // rows per tile
#ifndef RSDET_FAST_SUB
#define RSDET_FAST_SUB 4
#endif
constexpr int F_SUB = RSDET_FAST_SUB;  // heavy column tiles (large boxes: most rows overlap them) are cut into F_SUB row sub-tiles
constexpr int F_NW = F_R * (F_NT / 64);
constexpr int F_XCAP = 1024;  // flagged pairs kept per tile (more: the whole tile goes through tier 2)

struct FastArgs {
  const float* boxes1;   // raw rows (n1, stride1)
  const BoxPre* pre1;    // the same rows prepared (optional)
  int stride1, n1;
  const BoxPre* pre2;    // prepared columns, slab g at g * pitch when per_group
  const float4* colbox;  // bounding box of every 64 column circles
  int n2, cw, per_group;
  const int* row_offsets;  // n_groups + 1 (device); nullptr: one group = all rows
  const RowTile* tiles;    // n_row_tiles descriptors, or nullptr (then row tiles = n_groups x ny)
  int n_row_tiles, ny, nx;
  int split_xt;            // column tiles >= split_xt are cut into F_SUB sub-tiles (performance hint only)
  int vec4;                // matrix rows are 16-byte aligned (n2 % 4 == 0 and an aligned base)
  float* out;
};

template <int VERSION>
__global__ __launch_bounds__(F_NT) void iou_fast_tile_kernel(const FastArgs a) {
  __shared__ BoxPre s_row[F_R];
  __shared__ __attribute__((aligned(16))) BoxPre s_col[F_NT];
  __shared__ unsigned long long s_sm[F_NW];
  __shared__ unsigned short s_send[F_NW];
  __shared__ F2 s_pts[kQuadSlots * 16];
  __shared__ unsigned short s_xlist[F_XCAP];
  __shared__ unsigned s_nx;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  FTR(0);
  // heavy column tiles (the top pyramid levels: the last columns) first, each cut into F_SUB row sub-tiles
  const int n_heavy = (a.nx - a.split_xt) * a.n_row_tiles * F_SUB;
  const bool heavy = (int)blockIdx.x < n_heavy;
  int xt, rt, sub = 0;
  if (heavy) {
    const int per = a.n_row_tiles * F_SUB;
    xt = a.nx - 1 - (int)blockIdx.x / per;
    const int rem = (int)blockIdx.x % per;
    rt = rem / F_SUB;
    sub = rem - rt * F_SUB;
  } else {
    const int id = (int)blockIdx.x - n_heavy;
    xt = a.split_xt - 1 - id / a.n_row_tiles;
    rt = id % a.n_row_tiles;
  }
  const int col0 = xt * F_NT, col = col0 + tid;
  const int ncols = min(F_NT, a.n2 - col0);
  const bool col_ok = col < a.n2;

  // ---- LOADS FIRST: a CU serves its vector-memory queue in order, and the chip is about to be saturated by 48 MB of
  // zero stores -- loads queued behind this workgroup's own stores came back after 5 us (measured); queued ahead
  // of them they take ~1.5.  The tile's columns (40-byte prepared boxes: 2.5 float4 per column), its strip box, its rows.
  // With one column set for all groups the column loads do not depend on the tile table: they go out before it is read.
  const int n16 = (ncols * (int)sizeof(BoxPre)) / 16;
  float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0, cb = c0;
  float2 ctail = make_float2(0.f, 0.f);
  const bool has_tail = (ncols & 1) && tid == 0;   // odd column count: 8 bytes beyond the last whole float4
  const int kw = (col0 >> 6) + wave;
  auto load_cols = [&](long long slab, int slab_word) {
    const float4* src = reinterpret_cast<const float4*>(a.pre2 + slab + col0);  // 40 * 256 * xt bytes: 16-byte aligned
    if (tid < n16) c0 = src[tid];
    if (tid + F_NT < n16) c1 = src[tid + F_NT];
    if (tid + 2 * F_NT < n16) c2 = src[tid + 2 * F_NT];
    if (has_tail) ctail = reinterpret_cast<const float2*>(src)[n16 * 2];
    cb = a.colbox[slab_word + min(kw, a.cw - 1)];
  };
  if (!a.per_group) load_cols(0, 0);

  int g, row0, nrows;
  if (a.tiles) {
    const RowTile t = a.tiles[rt];
    g = t.group, row0 = t.row0, nrows = t.nrows;
  } else {
    g = rt / a.ny;
    const int y = rt - g * a.ny;
    int rb = 0, re = a.n1;
    if (a.row_offsets) {
      rb = a.row_offsets[g];
      re = a.row_offsets[g + 1];
    }
    row0 = rb + y * F_R;
    nrows = min(F_R, re - row0);
  }
  if (heavy) {
    row0 += sub * (F_R / F_SUB);
    nrows = min(F_R / F_SUB, nrows - sub * (F_R / F_SUB));
  }
  if (nrows <= 0) return;
  if (a.per_group) load_cols((long long)g * ((a.n2 + 1) & ~1), g * a.cw);
  float rraw[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) rraw[k] = 0.f;
  if (tid < nrows) {
    if (a.pre1) {
      const float* rp = reinterpret_cast<const float*>(a.pre1 + row0 + tid);
#pragma unroll
      for (int k = 0; k < 10; ++k) rraw[k] = rp[k];
    } else {
      const float* rp = a.boxes1 + (long long)(row0 + tid) * a.stride1;
#pragma unroll
      for (int k = 0; k < 5; ++k) rraw[k] = rp[k];
    }
  }

  // ---- the tile's zeros (16-byte-per-lane, a wave writes 1 KB of one matrix row per instruction), then the staging
  // of what was loaded above.  NST is a compile-time count so that the compiler waits for the loads with
  // s_waitcnt vmcnt(NST) and leaves the stores in flight (rows past the tile's end repeat its last row).
  auto stage = [&]() {
    float4* dst = reinterpret_cast<float4*>(s_col);
    if (tid < n16) dst[tid] = c0;
    if (tid + F_NT < n16) dst[tid + F_NT] = c1;
    if (tid + 2 * F_NT < n16) dst[tid + 2 * F_NT] = c2;
    if (has_tail) reinterpret_cast<float2*>(s_col)[n16 * 2] = ctail;
    if (tid < nrows) {
      if (a.pre1) {
        float* rd = reinterpret_cast<float*>(&s_row[tid]);
#pragma unroll
        for (int k = 0; k < 10; ++k) rd[k] = rraw[k];
      } else {
        s_row[tid] = prepare_box(rraw);
      }
    }
    for (int k = tid; k < F_NW; k += F_NT) s_sm[k] = 0ull;
    if (tid == 0) s_nx = 0u;
  };
  if (a.vec4) {
    const int c4 = min(col0 + 4 * lane, a.n2 - 4);      // lanes past a ragged tile's end repeat its last 16 bytes
    float* o = a.out + (long long)row0 * a.n2 + c4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (heavy) {
#pragma unroll
      for (int k = 0; k < F_R / F_SUB / 4; ++k)
        *reinterpret_cast<float4*>(o + (long long)min(wave + 4 * k, nrows - 1) * a.n2) = z;
      stage();
    } else {
#pragma unroll
      for (int k = 0; k < F_R / 4; ++k)
        *reinterpret_cast<float4*>(o + (long long)min(wave + 4 * k, nrows - 1) * a.n2) = z;
      stage();
    }
  } else {
    if (col_ok) {
      float* o = a.out + (long long)row0 * a.n2 + col;
      for (int r = 0; r < nrows; ++r) o[(long long)r * a.n2] = 0.0f;
    }
    stage();
  }
  if (kw >= a.cw) cb = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);  // empty strip
  lds_barrier();   // (LDS-only barrier: the zero stores stay in flight)
  FTR(1);
  const BoxPre mine = s_col[col_ok ? tid : 0];

  // ---- detection in one pass, no compaction: lane i < nrows tests row i's circle against the bounding box of this
  // wave's 64 column circles (strip cull); for every live (row, strip) the lanes whose circles touch run the
  // separating-axis test at once -- ~70 instructions per live strip, 2-3 live strips per wave in a sparse tile -- and
  // the ballot word IS the survivor set of (row, wave).
  static_assert(F_R <= 32, "row mask is 32 bits wide");
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
      if (!(dx * dx + dy * dy > rr * rr * 1.0001f))          // == !surely_disjoint(r, mine)
        surv = !sat_disjoint<VERSION>(r, mine);
    }
    const unsigned long long m = __ballot(surv);
    if (m && lane == 0) s_sm[i * (F_NT / 64) + wave] = m;
  }
  lds_barrier();
  FTR(2);
  scan_mask_words<F_NW>(s_sm, s_send, tid);
  lds_barrier();
  const int total = s_send[F_NW - 1];
  FTR(3);
  if (total == 0) return;

  // every zero of this workgroup is in L2 before the first value is stored
  __syncthreads();   // s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier
  FTR(4);

  // ---- tier 1: Green integral, one lane per survivor
  for (int k = tid; k < total; k += F_NT) {
    int word, bit;
    locate_bit<F_NW>(s_sm, s_send, k, word, bit);
    const int i = word >> 2, j = ((word & 3) << 6) | bit;
    bool danger, apart;
    const float v = pair_iou_fast<VERSION>(s_row[i], s_col[j], danger, apart);
    if (!apart && (danger || !(v >= kFastSliver))) {
      const unsigned at = atomicAdd(&s_nx, 1u);
      if (at < (unsigned)F_XCAP) s_xlist[at] = (unsigned short)((i << 8) | j);
    } else if (!apart) {
      a.out[(long long)(row0 + i) * a.n2 + col0 + j] = v;
    }
  }
  __syncthreads();
  FTR(5);
  // ---- tier 2: the flagged pairs through the reference-order clipper, 16 quads of the first wave
  const unsigned nflag = s_nx;
  if (nflag == 0u || wave != 0) return;
  F2* qscr = s_pts + (lane >> 2) * kQuadSlots;
  const bool overflow = nflag > (unsigned)F_XCAP;   // more than the list holds: every survivor of the tile
  const int n2nd = overflow ? total : (int)nflag;
  for (int q0 = 0; q0 < n2nd; q0 += 16) {
    const int q = q0 + (lane >> 2);
    const bool on = q < n2nd;
    int i, j;
    if (overflow) {
      int word, bit;
      locate_bit<F_NW>(s_sm, s_send, on ? q : 0, word, bit);
      i = word >> 2, j = ((word & 3) << 6) | bit;
    } else {
      const unsigned e = s_xlist[on ? q : 0];
      i = (int)(e >> 8), j = (int)(e & 255u);
    }
    const float v = pair_iou_quad<VERSION>(s_row[i], s_col[j], qscr, lane);
    if (on && (lane & 3) == 0) a.out[(long long)(row0 + i) * a.n2 + col0 + j] = v;
    lds_wave_order();
  }
  FTR(6);
}

}  // namespace rsdet

using namespace rsdet;

static inline size_t fast_up256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" int rsdet_box_iou_rotated_fast_f32(const float* boxes1, int n1, int stride1, const int* row_offsets,
                                              int n_groups, int max_rows_per_group, const int* tile_table,
                                              int n_row_tiles, const void* prepared1, const void* prepared2, int n2,
                                              int per_group, int heavy_from_col, int version, float* ious,
                                              void* stream) {
  if (n1 < 0 || n2 < 0 || n_groups < 1 || stride1 < 5 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;
  if (!boxes1 || !prepared2 || !ious) return RSDET_EINVAL;
  if (!row_offsets && n_groups != 1) return RSDET_EINVAL;
  FastArgs a{};
  a.boxes1 = boxes1, a.stride1 = stride1, a.n1 = n1;
  a.pre1 = (const BoxPre*)prepared1;
  // layout of rsdet_iou_prepare_f32: prepared boxes (slabs of an even pitch) | one float4 per 64 columns
  const long long groups = per_group ? n_groups : 1;
  a.pre2 = (const BoxPre*)prepared2;
  a.colbox = (const float4*)((const char*)prepared2 + fast_up256((size_t)(groups * ((n2 + 1) & ~1)) * sizeof(BoxPre)));
  a.n2 = n2, a.cw = (n2 + 63) / 64, a.per_group = per_group ? 1 : 0;
  a.row_offsets = row_offsets;
  a.tiles = (const RowTile*)tile_table;
  a.ny = (max_rows_per_group + F_R - 1) / F_R;
  a.n_row_tiles = tile_table ? n_row_tiles : n_groups * a.ny;
  a.nx = (n2 + F_NT - 1) / F_NT;
  a.split_xt = (heavy_from_col < 0 || heavy_from_col >= n2) ? a.nx : heavy_from_col / F_NT;
  a.vec4 = ((n2 & 3) == 0 && n2 >= 4 && ((uintptr_t)ious & 15) == 0) ? 1 : 0;
  a.out = ious;
  if (a.n_row_tiles <= 0) return RSDET_OK;
  const dim3 grid((unsigned)((long long)a.split_xt * a.n_row_tiles + (long long)(a.nx - a.split_xt) * a.n_row_tiles * F_SUB));
  if (version == 0)
    hipLaunchKernelGGL(iou_fast_tile_kernel<0>, grid, dim3(F_NT), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(iou_fast_tile_kernel<1>, grid, dim3(F_NT), 0, (hipStream_t)stream, a);
  return rsdet_launch_status();
}

extern "C" int rsdet_box_iou_rotated_fast_rows_per_tile(void) { return F_R; }

#ifdef RSDET_FAST_TRACE
extern "C" void rsdet_debug_set_fast_trace(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fast_trace), &p, sizeof(p)); }
#endif
