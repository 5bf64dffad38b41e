// LD_PRELOAD shim: counts every kernel launch (and async fill / copy) a process makes through the HIP runtime,
// whoever makes it (torch, MIOpen, rocBLAS / hipBLASLt, librsdet_hip.so).  Test infrastructure for
// tests/test_gpu_guards.py -- the "launches per step" regression guard without a profiler in the loop.
//   g++ -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include launch_counter.cpp -o liblaunch_counter.so -ldl
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <link.h>
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static std::atomic<long long> g_launches{0}, g_fills{0}, g_copies{0};

extern "C" {
long long rsdet_lc_launches() { return g_launches.load(); }
long long rsdet_lc_fills() { return g_fills.load(); }
long long rsdet_lc_copies() { return g_copies.load(); }
}

// The real entry point.  RTLD_NEXT only sees the GLOBAL lookup scope; Python loads torch's libraries RTLD_LOCAL, so
// libamdhip64 is usually not in it (the preloaded shim still wins the lookup for the callers: LD_PRELOAD objects are
// searched first for every object).  Fallback: find the loaded libamdhip64 by walking the link maps.
static int find_hip(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) {
    *static_cast<void**>(out) = dlopen(info->dlpi_name, RTLD_NOW | RTLD_NOLOAD);
    return 1;
  }
  return 0;
}
static void* real_sym(const char* name) {
  if (void* p = dlsym(RTLD_NEXT, name)) return p;
  static void* hip = nullptr;
  if (!hip) dl_iterate_phdr(find_hip, &hip);
  void* p = hip ? dlsym(hip, name) : nullptr;
  if (!p) {
    fprintf(stderr, "launch_counter: cannot resolve %s\n", name);
    abort();
  }
  return p;
}
template <class F> static F next_sym(const char* name) { return reinterpret_cast<F>(real_sym(name)); }
#define NEXT(name) static auto real = next_sym<decltype(&name)>(#name)

extern "C" {
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** a, size_t sh, hipStream_t s) {
  NEXT(hipLaunchKernel); ++g_launches; return real(f, g, b, a, sh, s);
}
hipError_t hipExtLaunchKernel(const void* f, dim3 g, dim3 b, void** a, size_t sh, hipStream_t s, hipEvent_t e0,
                              hipEvent_t e1, int fl) {
  NEXT(hipExtLaunchKernel); ++g_launches; return real(f, g, b, a, sh, s, e0, e1, fl);
}
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by,
                                 unsigned bz, unsigned sh, hipStream_t s, void** p, void** ex) {
  NEXT(hipModuleLaunchKernel); ++g_launches; return real(f, gx, gy, gz, bx, by, bz, sh, s, p, ex);
}
hipError_t hipExtModuleLaunchKernel(hipFunction_t f, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                                    uint32_t bz, size_t sh, hipStream_t s, void** p, void** ex, hipEvent_t e0,
                                    hipEvent_t e1, uint32_t fl) {
  NEXT(hipExtModuleLaunchKernel); ++g_launches; return real(f, gx, gy, gz, bx, by, bz, sh, s, p, ex, e0, e1, fl);
}
hipError_t hipLaunchCooperativeKernel(const void* f, dim3 g, dim3 b, void** a, unsigned sh, hipStream_t s) {
  typedef hipError_t (*fn_t)(const void*, dim3, dim3, void**, unsigned, hipStream_t);   // (the header also has a template overload)
  static fn_t real = next_sym<fn_t>("hipLaunchCooperativeKernel"); ++g_launches; return real(f, g, b, a, sh, s);
}
hipError_t hipModuleLaunchCooperativeKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx,
                                            unsigned by, unsigned bz, unsigned sh, hipStream_t s, void** p) {
  NEXT(hipModuleLaunchCooperativeKernel); ++g_launches; return real(f, gx, gy, gz, bx, by, bz, sh, s, p);
}
hipError_t hipLaunchKernelExC(const hipLaunchConfig_t* c, const void* f, void** a) {
  NEXT(hipLaunchKernelExC); ++g_launches; return real(c, f, a);
}
hipError_t hipDrvLaunchKernelEx(const HIP_LAUNCH_CONFIG* c, hipFunction_t f, void** p, void** ex) {
  NEXT(hipDrvLaunchKernelEx); ++g_launches; return real(c, f, p, ex);
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t s) {
  NEXT(hipMemsetAsync); ++g_fills; return real(d, v, n, s);
}
hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t n, hipStream_t s) {
  NEXT(hipMemsetD32Async); ++g_fills; return real(d, v, n, s);
}
hipError_t hipMemsetD8Async(hipDeviceptr_t d, unsigned char v, size_t n, hipStream_t s) {
  NEXT(hipMemsetD8Async); ++g_fills; return real(d, v, n, s);
}
hipError_t hipMemcpyAsync(void* d, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) {
  NEXT(hipMemcpyAsync); ++g_copies; return real(d, src, n, k, s);
}
hipError_t hipMemcpyWithStream(void* d, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) {
  NEXT(hipMemcpyWithStream); ++g_copies; return real(d, src, n, k, s);
}
}
