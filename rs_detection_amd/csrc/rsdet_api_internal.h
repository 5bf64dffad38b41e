// Shared by every .hip translation unit of librsdet_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rsdet.h"

// Status of the launch just enqueued (no sync): maps a HIP launch error to
// RSDET_ELAUNCH.  Never throws; the C ABI returns ints only.
static inline int rsdet_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RSDET_OK : RSDET_ELAUNCH;
}

static inline int rsdet_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// XCD-aware renumbering of a 1-D grid (speed only, never correctness).  MI355X deals consecutive workgroup ids
// round-robin over its 8 XCDs, each with a private 4 MiB L2 (MI355X_MICROARCH.md, Workgroup dispatch): ids b and
// b + 8 share an L2, b and b + 1 do not.  This bijection on [0, total) hands every XCD one CONTIGUOUS range of
// logical ids, so that work items that re-read each other's cache lines (neighbouring image rows, the offsets of one
// position block, the rows of one column-gradient tile) can be made neighbours in L2 by making them neighbours in
// the logical order.
__device__ __forceinline__ unsigned rsdet_xcd_contiguous(unsigned id, unsigned total) {
  const unsigned q = total >> 3, r = total & 7u, xcd = id & 7u, slot = id >> 3;
  return xcd * q + (xcd < r ? xcd : r) + slot;
}

// Two-level form for (outer x inner) work, e.g. position blocks x channel slices: the outer range is cut into 8
// contiguous bands, one per XCD; inside a band the OUTER index runs fastest, so the ~256 workgroups an XCD holds at a
// time are neighbouring outer items of a few inner slices (their shared cache lines meet in that L2, and the working
// set is a few slices deep instead of all of them).  Launch rsdet_xcd_band_grid(n_outer, n_inner) workgroups; ids
// past a band's end return valid = false.
struct RsdetBandItem {
  int outer, inner;
  bool valid;
};
__host__ __device__ inline long long rsdet_xcd_band_grid(long long n_outer, long long n_inner) {
  return 8 * ((n_outer + 7) / 8) * n_inner;
}
__device__ __forceinline__ RsdetBandItem rsdet_xcd_band(unsigned id, int n_outer, int n_inner) {
  const int xcd = (int)(id & 7u), slot = (int)(id >> 3);
  const int o0 = (int)((long long)n_outer * xcd / 8), o1 = (int)((long long)n_outer * (xcd + 1) / 8);
  const int len = o1 - o0;
  RsdetBandItem it{0, 0, false};
  if (len <= 0 || slot >= len * n_inner) return it;
  it.inner = slot / len;
  it.outer = o0 + (slot - it.inner * len);
  it.valid = true;
  return it;
}

// Shared by the rotated / horizontal / polygon NMS entry points (defined in nms_rotated.hip): the device sweep over
// the sparse suppression entries.  `entries` holds col_blocks lists of 64*col_blocks 16-byte records
// {u64 bits, int column block, int row}, `blk_cnt` their lengths, `diag_t` the transposed diagonal tiles.
void rsdet_launch_nms_sweep(const void* entries, const unsigned* blk_cnt, const unsigned long long* diag_t, int n,
                            int col_blocks, const int* order, unsigned char* keep, hipStream_t stream);

// Exclusive prefix sum of n int counters (defined in deform_conv.hip, used by the gather-form backward kernels):
// start[0..n] <- scan(cnt[0..n)), start[n] = total; cnt is cleared; chunk_sum needs n / 4096 + 1 ints.
void rsdet_launch_index_scan(int* cnt, long long n, int* chunk_sum, int* start, hipStream_t stream);

// Gather stage shared by the gather-form backward kernels (defined in rroi_align.hip): one wave per output pixel
// sums  out[pix, :] = sum_{e in [start[pix], start[pix+1])} ent_w[e] * rows[ent_row[e], :]  over C channels
// (rows and out channels-last); every element of out is written exactly once.
// csrc/bn_act.hip: (C, S, 2) per-slice partial sums -> dbias[c] = sum_s [0], dweight[c] = sum_s [1] (either NULL: skipped)
void rsdet_launch_sums_finish(const float* partial, int C, int S, float* dweight, float* dbias, hipStream_t stream);
void rsdet_launch_pixel_gather(const float* rows, const int* start, const int* ent_row, const float* ent_w,
                               long long npix, int C, float* out_nhwc, hipStream_t stream);
