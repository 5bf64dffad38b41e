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
