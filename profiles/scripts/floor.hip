// store-bandwidth floor probes
#include <hip/hip_runtime.h>
extern "C" __global__ void fill4(float4* p, long long n4) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) p[i] = make_float4(0, 0, 0, 0);
}
extern "C" __global__ void fill4_tile(float4* p, long long n4, int per_block) {  // each block a contiguous chunk
  long long base = (long long)blockIdx.x * per_block;
  for (int k = threadIdx.x; k < per_block; k += blockDim.x) {
    long long i = base + k;
    if (i < n4) p[i] = make_float4(0, 0, 0, 0);
  }
}
extern "C" __global__ void fill1(float* p, long long n) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0.f;
}
extern "C" __global__ void nullk(int* p) { if (p && threadIdx.x == 9999) *p = 0; }
extern "C" void run_fill4(void* p, long long n4, int blocks, void* s) { hipLaunchKernelGGL(fill4, dim3(blocks), dim3(256), 0, (hipStream_t)s, (float4*)p, n4); }
extern "C" void run_fill4_tile(void* p, long long n4, int per_block, void* s) { int blocks = (int)((n4 + per_block - 1) / per_block); hipLaunchKernelGGL(fill4_tile, dim3(blocks), dim3(256), 0, (hipStream_t)s, (float4*)p, n4, per_block); }
extern "C" void run_fill1(void* p, long long n, int blocks, void* s) { hipLaunchKernelGGL(fill1, dim3(blocks), dim3(256), 0, (hipStream_t)s, (float*)p, n); }
extern "C" void run_null(void* s) { hipLaunchKernelGGL(nullk, dim3(1), dim3(64), 0, (hipStream_t)s, (int*)nullptr); }
extern "C" void run_memset(void* p, long long bytes, void* s) { hipMemsetAsync(p, 0, bytes, (hipStream_t)s); }
