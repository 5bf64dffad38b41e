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
void rsdet_launch_pixel_gather(const float* rows, const int* start, const int* ent_row, const float* ent_w,
                               long long npix, int C, float* out_nhwc, hipStream_t stream);
