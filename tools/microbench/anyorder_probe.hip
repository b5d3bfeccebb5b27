// Does hipExtAnyOrderLaunch (hip_ext.h: "not supported on AMD GFX9xx boards") let a kernel START before its stream predecessor has
// finished on gfx950?  Kernel A spins ~30 us and records its end time, kernel B records its start time (s_memrealtime, 100 MHz).
//   hipcc -O2 --offload-arch=gfx950 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__global__ void spin_kernel(unsigned long long* end, int ticks) {
  const unsigned long long t0 = now();
  while (now() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) end[blockIdx.x] = now();
}
__global__ void mark_kernel(unsigned long long* start) { if (threadIdx.x == 0) start[blockIdx.x] = now(); }
int main() {
  const int NB = 256;
  unsigned long long *a, *b;
  CK(hipMalloc(&a, NB * 8)); CK(hipMalloc(&b, NB * 8));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<unsigned long long> ha(NB), hb(NB);
  for (int flags = 0; flags <= 1; ++flags)
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, s, a, 3000);
      hipExtLaunchKernelGGL(mark_kernel, dim3(NB), dim3(256), 0, s, nullptr, nullptr, flags, b);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(ha.data(), a, NB * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, NB * 8, hipMemcpyDeviceToHost));
      const long long last_end = (long long)*std::max_element(ha.begin(), ha.end()), first_end = (long long)*std::min_element(ha.begin(), ha.end());
      const long long first_start = (long long)*std::min_element(hb.begin(), hb.end());
      printf("flags %d: successor's first wave starts %+.2f us after the predecessor's LAST workgroup ended (%+.2f us after its first ended)\n", flags,
             0.01 * (first_start - last_end), 0.01 * (first_start - first_end));
    }
  return 0;
}
