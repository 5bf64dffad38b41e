// iou_grid.hip -- dense rotated IoU against a PYRAMID-GRID column set, ONE launch, no pair-wise detection (gfx950).
//
// Same values as rsdet_box_iou_rotated_fast_f32 (csrc/iou_fast.hip: two-tier clipper, |IoU - reference| < 3e-6,
// exact zeros, the reference-order clipper wherever the reference itself is fragile) for the column sets the S2ANet
// heads produce: one anchor per cell of a few regular grids (strides 8 .. 128), exactly on the grid (FAM call,
// /root/reference/python/jdet/models/boxes/anchor_generator.py:7-91) or moved by a bounded refinement (ODM call,
// roi_heads/s2anet_head.py:631-654).  Replaces the pair loop of ops/box_iou_rotated.py:487-500 for those callers.
//
// What changes against the tile kernel, and why (profiles/experiments/iou_fast_r04_store_once.md: 52 % of its VALU
// instructions were DETECTION -- every (32 rows x 256 columns) tile tests row circles against strip boxes, then
// circles, then separating axes -- and its tail was the compute chain of the last workgroup):
//   * the anchors a gt can touch are a closed-form cell window per level: |centre distance| <= r_gt + r_level (+ the
//     level's maximum displacement), i.e. a rectangle of cells; no strip cull, no per-pair circle pass over the
//     1.2e7 pairs -- only the ~400 cells of the windows of a gt are ever looked at;
//   * ONE WORKGROUP = ONE MATRIX ROW x ONE COLUMN CHUNK (4 096 columns, 16 KB): the chunk is composed in LDS and every
//     element leaves exactly once, as 16-byte-per-lane row-contiguous stores.  Segments (1 KB) no window reaches --
//     most of the matrix -- are stored as zeros FIRST, before the gt's fp64 sincos, so the 48 MB zero stream is in
//     flight while the few candidates are computed; the segments a window touches follow from LDS;
//   * no workgroup depends on another one and none carries more than one row: the tail is one row's chain.
// Per candidate cell: the column's prepared box (40 B, L2-resident), the SAME tests in the SAME order as the tile
// kernel (surely_disjoint -> sat_disjoint -> pair_iou_fast -> flagged pairs through pair_iou_quad), so the two entry
// points agree bit for bit (tests/test_gpu_iou_grid.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsdet_api_internal.h"
#include "rsdet_geom_fast.h"
#include "rsdet_tile.h"

namespace rsdet {

constexpr int GR_NT = 256;
#ifndef RSDET_GRID_CHUNK
#define RSDET_GRID_CHUNK 4096
#endif
constexpr int GR_CHUNK = RSDET_GRID_CHUNK;          // columns per workgroup
constexpr int GR_SEG = 256;                         // columns per store segment (1 KB = one wave instruction)
constexpr int GR_NSEG = GR_CHUNK / GR_SEG;          // 16
constexpr int GR_WORDS = GR_CHUNK / 64 < 64 ? 64 : GR_CHUNK / 64;   // flag words (whole words per lane of the scanning wave)
constexpr int GR_MAXL = RSDET_GRID_MAX_LEVELS;      // 8
static_assert(GR_NSEG <= 32 && GR_WORDS % 64 == 0 && GR_WORDS <= 256, "segment mask is 32 bits; flag words scanned by one wave");

struct GridArgs {
  const float* boxes1;
  const BoxPre* pre1;
  int stride1, n1;
  int n2, chunks;
  RsdetGridLevel lv[GR_MAXL];
  int nl;
  int vec4;
  float* out;
};

// The prepared form of the GENERATED anchor of cell (i, j) is prepare_box() (rsdet_geom.h) of (cx, cy, w, h, theta = 0),
// operation by operation -- cos(0) = 1, sin(0) = 0 exactly, so the fp64 sincos drops out and the result is bit-identical.
// Everything but the centre is a constant of the level: kept once per workgroup in LDS.
struct GrLevelPre {
  int col0, W;
  float x0, y0, stride;
  float cw, sw, ch, sh, area, rad, lu, lv;
};
__device__ __forceinline__ GrLevelPre grid_level_pre(const RsdetGridLevel& L) {
  GrLevelPre p;
  const float c2 = 0.5f, s2 = 0.0f;
  p.col0 = L.col0, p.W = L.W, p.x0 = L.x0, p.y0 = L.y0, p.stride = L.stride;
  p.cw = c2 * L.box_w;
  p.sw = s2 * L.box_w;
  p.ch = c2 * L.box_h;
  p.sh = s2 * L.box_h;
  p.area = L.box_w * L.box_h;
  p.rad = 0.5f * (fabsf(L.box_w) + fabsf(L.box_h)) * 1.0001f + 1e-3f;
  p.lu = sqrtf(p.cw * p.cw + p.sw * p.sw);
  p.lv = sqrtf(p.ch * p.ch + p.sh * p.sh);
  return p;
}
__device__ __forceinline__ BoxPre grid_cell_box(const GrLevelPre& L, int i, int j) {
  BoxPre p;
  p.cx = L.x0 + (float)j * L.stride;
  p.cy = L.y0 + (float)i * L.stride;
  p.cw = L.cw, p.sw = L.sw, p.ch = L.ch, p.sh = L.sh, p.area = L.area, p.rad = L.rad, p.lu = L.lu, p.lv = L.lv;
  return p;
}

struct GrWin {
  int i0, j0, nj, cnt;      // candidate k of the level: cell (i0 + k / nj, j0 + k % nj)
};

template <int VERSION>
__global__ __launch_bounds__(GR_NT) void iou_grid_kernel(const GridArgs a) {
  __shared__ __attribute__((aligned(16))) float s_val[GR_CHUNK];
  __shared__ unsigned long long s_flag[GR_WORDS];
  __shared__ unsigned short s_end[GR_WORDS];
  __shared__ F2 s_pts[kQuadSlots * 16];
  __shared__ GrLevelPre s_lv[GR_MAXL];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // chunk-major, last chunk first: the chunks of the upper pyramid levels (every gt has candidates there) start first,
  // the level-0 chunks -- most of them untouched by a given gt -- fill in behind them
  const int ch = a.chunks - 1 - (int)blockIdx.x / a.n1, row = (int)blockIdx.x % a.n1;
  const int c0 = ch * GR_CHUNK, c1 = min(a.n2, c0 + GR_CHUNK), ncol = c1 - c0;

  // ---- the row: raw values first (centre + sizes are all the windows need)
  float rraw[10];
  if (a.pre1) {
    const float* rp = reinterpret_cast<const float*>(a.pre1 + row);
#pragma unroll
    for (int k = 0; k < 10; ++k) rraw[k] = rp[k];
  } else {
    const float* rp = a.boxes1 + (long long)row * a.stride1;
#pragma unroll
    for (int k = 0; k < 5; ++k) rraw[k] = rp[k];
  }
  const float rcx = rraw[0], rcy = rraw[1];
  const float rrad = a.pre1 ? rraw[7] : 0.5f * (fabsf(rraw[2]) + fabsf(rraw[3])) * 1.0001f + 1e-3f;   // == prepare_box
  const bool finite = fabsf(rcx) < 1e30f && fabsf(rcy) < 1e30f && rrad < 1e30f;   // NaN / Inf rows: every cell is a candidate

  // ---- per level: the cell window inside this chunk and the 1 KB segments it can reach.  Lane l of every wave works
  // out level l; the results travel to all lanes as wave-uniform values (v_readlane): no LDS round trip, no barrier
  GrWin mywin{0, 0, 1, 0};
  unsigned mydirty = 0u;
  if (lane < a.nl) {
    const RsdetGridLevel L = a.lv[lane];
    const int lo = max(c0 - L.col0, 0), hi = min(c1 - L.col0, L.H * L.W) - 1;    // the level's columns in [c0, c1)
    int i0 = lo / L.W, i1 = hi / L.W, j0 = 0, j1 = L.W - 1;
    bool any = hi >= lo;
    if (any && finite) {
      // surely_disjoint keeps pairs with |d| <= (r_gt + r_cell) * sqrt(1.0001); the window is that radius widened by
      // far more than the rounding of these few operations
      const float lrad = 0.5f * (fabsf(L.box_w) + fabsf(L.box_h)) * 1.0001f + 1e-3f;
      const float R = (rrad + lrad) * 1.0002f + 0.02f + 1e-5f * (fabsf(rcx) + fabsf(rcy));
      const float inv = 1.0f / L.stride;
      const float fj0 = floorf((rcx - R - L.x0) * inv), fj1 = ceilf((rcx + R - L.x0) * inv);
      const float fi0 = floorf((rcy - R - L.y0) * inv), fi1 = ceilf((rcy + R - L.y0) * inv);
      // (clamped as floats: the quotient of a far-away gt does not fit an int)
      const float cj0 = fmaxf(fj0, 0.f), cj1 = fminf(fj1, (float)(L.W - 1));
      const float ci0 = fmaxf(fi0, (float)i0), ci1 = fminf(fi1, (float)i1);
      any = cj1 >= cj0 && ci1 >= ci0;
      if (any) j0 = (int)cj0, j1 = (int)cj1, i0 = (int)ci0, i1 = (int)ci1;
    }
    if (any) {
      mywin.i0 = i0, mywin.j0 = j0, mywin.nj = j1 - j0 + 1, mywin.cnt = (i1 - i0 + 1) * mywin.nj;
      const int first = max(L.col0 + i0 * L.W + j0, c0) - c0, last = min(L.col0 + i1 * L.W + j1, c1 - 1) - c0;
      const int s0 = first / GR_SEG, s1 = last / GR_SEG;
      mydirty = (s1 >= 31 ? 0xffffffffu : ((2u << s1) - 1u)) & ~((1u << s0) - 1u);
    }
  }
  GrWin win[GR_MAXL];
  unsigned dirty = 0u;
  int total = 0;
#pragma unroll
  for (int l = 0; l < GR_MAXL; ++l) {
    win[l].i0 = __builtin_amdgcn_readlane(mywin.i0, l);
    win[l].j0 = __builtin_amdgcn_readlane(mywin.j0, l);
    win[l].nj = __builtin_amdgcn_readlane(mywin.nj, l);
    win[l].cnt = __builtin_amdgcn_readlane(mywin.cnt, l);
    dirty |= (unsigned)__builtin_amdgcn_readlane((int)mydirty, l);
    total += win[l].cnt;
  }
  const int nseg = (ncol + GR_SEG - 1) / GR_SEG;
  float* orow = a.out + (long long)row * a.n2 + c0;

  // ---- zeros of the untouched segments: 16 bytes per lane, a wave writes one segment per instruction, never waited for
  if (a.vec4) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < GR_NSEG / 4; ++k) {
      const int s = wave + 4 * k;
      if (s < nseg && !((dirty >> s) & 1u) && s * GR_SEG + 4 * lane < ncol)
        *reinterpret_cast<float4*>(orow + s * GR_SEG + 4 * lane) = z;
    }
  } else {
    for (int s = 0; s < nseg; ++s)
      if (!((dirty >> s) & 1u) && s * GR_SEG + tid < ncol) orow[s * GR_SEG + tid] = 0.f;
  }
  if (dirty == 0u) return;

  // ---- the dirty segments start as zeros in LDS
  {
    float4* v4 = reinterpret_cast<float4*>(s_val);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < GR_CHUNK / 4 / GR_NT; ++k) v4[tid + k * GR_NT] = z;
    if (tid < GR_WORDS) s_flag[tid] = 0ull;
    if (tid < a.nl) s_lv[tid] = grid_level_pre(a.lv[tid]);
  }
  // the row, prepared (fp64 sincos: the same BoxPre the other entry points use), every lane for itself -- uniform
  BoxPre rb;
  if (a.pre1) {
    float* rd = reinterpret_cast<float*>(&rb);
#pragma unroll
    for (int k = 0; k < 10; ++k) rd[k] = rraw[k];
  } else {
    rb = prepare_box(rraw);
  }
  lds_barrier();

  // ---- candidates: the cells of the windows; the tile kernel's tests in the tile kernel's order
  for (int k = tid; k < total; k += GR_NT) {
    int kk = k, l = 0;
#pragma unroll
    for (int q = 0; q < GR_MAXL - 1; ++q) {
      if (l == q && kk >= win[q].cnt) {
        kk -= win[q].cnt;
        l = q + 1;
      }
    }
    int wi0 = win[0].i0, wj0 = win[0].j0, wnj = win[0].nj;
#pragma unroll
    for (int q = 1; q < GR_MAXL; ++q)
      if (l == q) wi0 = win[q].i0, wj0 = win[q].j0, wnj = win[q].nj;
    const int di = kk / wnj, dj = kk - di * wnj;
    const int ci = wi0 + di, cj = wj0 + dj;
    const int col = s_lv[l].col0 + ci * s_lv[l].W + cj;
    if (col < c0 || col >= c1) continue;
    const BoxPre cb = grid_cell_box(s_lv[l], ci, cj);
    const float dx = rb.cx - cb.cx, dy = rb.cy - cb.cy, rr = rb.rad + cb.rad;
    if (dx * dx + dy * dy > rr * rr * 1.0001f) continue;                    // == surely_disjoint(rb, cb)
    if (sat_disjoint<VERSION>(rb, cb)) continue;
    bool danger, apart;
    const float v = pair_iou_fast<VERSION>(rb, cb, danger, apart);
    if (apart) continue;
    const int at = col - c0;
    if (danger || !(v >= kFastSliver))
      atomicOr(&s_flag[at >> 6], 1ull << (at & 63));
    else
      s_val[at] = v;
  }
  lds_barrier();

  // ---- tier 2: flagged pairs through the reference-order clipper, 16 quads of the first wave
  scan_mask_words<GR_WORDS>(s_flag, s_end, tid);
  lds_barrier();
  const int nflag = s_end[GR_WORDS - 1];
  if (nflag && wave == 0) {
    F2* qscr = s_pts + (lane >> 2) * kQuadSlots;
    for (int q0 = 0; q0 < nflag; q0 += 16) {
      const int q = q0 + (lane >> 2);
      const bool on = q < nflag;
      int word, bit;
      locate_bit<GR_WORDS>(s_flag, s_end, on ? q : 0, word, bit);
      const int at = (word << 6) | bit, col = c0 + at;
      int l = 0;
      for (int qq = 1; qq < a.nl; ++qq)
        if (col >= s_lv[qq].col0) l = qq;
      const int rel = col - s_lv[l].col0, ci = rel / s_lv[l].W, cj = rel - ci * s_lv[l].W;
      const BoxPre cb = grid_cell_box(s_lv[l], ci, cj);
      const float v = pair_iou_quad<VERSION>(rb, cb, qscr, lane);
      if (on && (lane & 3) == 0) s_val[at] = v;
      lds_wave_order();
    }
  }
  if (nflag) lds_barrier();

  // ---- the dirty segments leave from LDS
  if (a.vec4) {
    const float4* v4 = reinterpret_cast<const float4*>(s_val);
#pragma unroll
    for (int k = 0; k < GR_NSEG / 4; ++k) {
      const int s = wave + 4 * k;
      if (s < nseg && ((dirty >> s) & 1u) && s * GR_SEG + 4 * lane < ncol)
        *reinterpret_cast<float4*>(orow + s * GR_SEG + 4 * lane) = v4[s * (GR_SEG / 4) + lane];
    }
  } else {
    for (int s = 0; s < nseg; ++s)
      if (((dirty >> s) & 1u) && s * GR_SEG + tid < ncol) orow[s * GR_SEG + tid] = s_val[s * GR_SEG + tid];
  }
}

}  // namespace rsdet

using namespace rsdet;

static int grid_levels_ok(const RsdetGridLevel* levels, int n_levels, int n2) {
  if (!levels || n_levels < 1 || n_levels > GR_MAXL) return 0;
  long long at = 0;
  for (int l = 0; l < n_levels; ++l) {        // the levels tile [0, n2) in order
    if (levels[l].col0 != at || levels[l].H < 1 || levels[l].W < 1 || !(levels[l].stride > 0.f)) return 0;
    at += (long long)levels[l].H * levels[l].W;
  }
  return at == n2;
}

extern "C" int rsdet_box_iou_rotated_grid_chunk(void) { return GR_CHUNK; }

extern "C" int rsdet_box_iou_rotated_grid_f32(const float* boxes1, int n1, int stride1, const void* prepared1, int n2,
                                              const RsdetGridLevel* levels, int n_levels, int version, float* ious,
                                              void* stream) {
  if (n1 < 0 || n2 < 0 || stride1 < 5 || (version != 0 && version != 1)) return RSDET_EINVAL;
  if (n1 == 0 || n2 == 0) return RSDET_OK;
  if (!boxes1 || !ious || !grid_levels_ok(levels, n_levels, n2)) return RSDET_EINVAL;
  GridArgs a{};
  a.boxes1 = boxes1, a.stride1 = stride1, a.n1 = n1;
  a.pre1 = (const BoxPre*)prepared1;
  a.n2 = n2, a.chunks = (n2 + GR_CHUNK - 1) / GR_CHUNK;
  for (int l = 0; l < n_levels; ++l) a.lv[l] = levels[l];
  a.nl = n_levels;
  a.vec4 = ((n2 & 3) == 0 && ((uintptr_t)ious & 15) == 0) ? 1 : 0;
  a.out = ious;
  const long long grid = (long long)n1 * a.chunks;
  if (grid > 0x7fffffffLL) return RSDET_EINVAL;
  if (version == 0)
    hipLaunchKernelGGL(iou_grid_kernel<0>, dim3((unsigned)grid), dim3(GR_NT), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(iou_grid_kernel<1>, dim3((unsigned)grid), dim3(GR_NT), 0, (hipStream_t)stream, a);
  return rsdet_launch_status();
}
