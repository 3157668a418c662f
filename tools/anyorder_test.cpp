// Does hipExtAnyOrderLaunch let consecutive independent kernels of ONE stream overlap on gfx950?  (hip_ext.h says not on GFX9xx.)
// Kernels of 300 workgroups x 30 us that fit two per CU: a ragged single round each; overlap would fill the tails.
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/anyorder_test tools/anyorder_test.cpp
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
__global__ __launch_bounds__(512) void busy_kernel(long ticks, unsigned* sink) {
  extern __shared__ char lds[];
  const unsigned long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long)(ticks + (blockIdx.x & 7) * 100)) acc += 1;      // 30-37 us: a ragged tail
  if (acc == 0xFFFFFFFFu) { *sink = acc; lds[0] = 1; }
}
int main() {
  unsigned* sink; CK(hipMalloc(&sink, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  CK(hipFuncSetAttribute((const void*)busy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024));
  for (int rep = 0; rep < 2; ++rep)
  for (int flags = 0; flags < 2; ++flags) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, st));
    for (int i = 0; i < 200; ++i)
      hipExtLaunchKernelGGL(busy_kernel, dim3(300), dim3(512), 66 * 1024, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, 3000L, sink);
    CK(hipEventRecord(t1, st)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    if (rep) printf("flags = %s: %.2f us per kernel (300 workgroups of 30-37 us, two fit per CU)\n", flags ? "hipExtAnyOrderLaunch" : "0", ms * 1e3 / 200);
  }
  return 0;
}
