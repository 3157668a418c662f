#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(int* out) {
  extern __shared__ char sm[];
  if (threadIdx.x == 0) { unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); out[blockIdx.x] = (int)(x & 15); if (sm[0] == 77) out[0] = -1; }
}
static void test(int grid, int threads, int lds) {
  int* d; hipMalloc(&d, grid * 4);
  hipFuncSetAttribute((const void*)who, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(who, dim3(grid), dim3(threads), lds, 0, d);
  std::vector<int> h(grid); hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < grid; ++i) bad += (h[i] != (i & 7));
  printf("grid %5d x %4d threads, %6d B LDS: XCC id == blockIdx %% 8 for %d of %d workgroups; first 16:", grid, threads, lds, grid - bad, grid);
  for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
  printf("\n"); hipFree(d);
}
int main() { test(1024, 256, 0); test(256, 512, 110 * 1024); test(256, 512, 75 * 1024); test(640, 256, 64 * 1024); test(4096, 256, 0); return 0; }
