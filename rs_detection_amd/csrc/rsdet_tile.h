// rsdet_tile.h -- bit-mask bookkeeping of the IoU tile kernels (candidate / survivor sets as 64-bit words per
// (row, 64-column strip) instead of index lists: 1-2 KB of LDS instead of 8-16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsdet {

struct RowTile {  // one per row tile (host-built when the gt counts are host-known); same layout as TileDesc
  int group, row0, nrows, group_row0;
};

// Position of the r-th (0-based) set bit of m; r < popcount(m).
__device__ __forceinline__ int nth_set_bit64(unsigned long long m, int r) {
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

// word + bit of the k-th set bit over NW mask words (inclusive prefix counts in `end`); NW a power of two
template <int NW>
__device__ __forceinline__ void locate_bit(const unsigned long long* __restrict__ mask,
                                           const unsigned short* __restrict__ end, int k, int& word, int& bit) {
  int lo = 0;
#pragma unroll
  for (int step = NW / 2; step > 0; step >>= 1)
    if ((int)end[lo + step - 1] <= k) lo += step;
  const int before = lo ? (int)end[lo - 1] : 0;
  word = lo;
  bit = nth_set_bit64(mask[lo], k - before);
}

// wave 0 (tid < 64): inclusive scan of the NW popcounts, NW / 64 consecutive words per lane
template <int NW>
__device__ __forceinline__ void scan_mask_words(const unsigned long long* __restrict__ mask,
                                                unsigned short* __restrict__ end, int tid) {
  static_assert(NW % 64 == 0 && NW <= 256, "whole words per lane");
  constexpr int PER = NW / 64;
  if (tid < 64) {
    int c[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      sum += __popcll(mask[tid * PER + k]);
      c[k] = sum;
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (tid >= off) incl += o;
    }
    const int before = incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) end[tid * PER + k] = (unsigned short)(before + c[k]);
  }
}

}  // namespace rsdet
