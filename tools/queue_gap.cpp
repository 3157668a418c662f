// What a kernel boundary costs a stream while ANOTHER stream is dispatching too, by which of the process's streams the two are
// (HIP streams map onto a few hardware queues / pipes).  Main stream: N short kernels back to back; partner stream: longer kernels on
// a quarter of the CUs, back to back, for the whole time.  Prints the main stream's time per kernel alone and beside each partner.
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/queue_gap tools/queue_gap.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
__global__ void busy_kernel(long ticks, unsigned* sink) {      // every wave spins for `ticks` of the 100 MHz clock
  const unsigned long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long)ticks) acc += 1;
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
int main(int argc, char** argv) {
  const int nstreams = argc > 1 ? atoi(argv[1]) : 8;
  const int nk = 400;
  unsigned* sink; CK(hipMalloc(&sink, 4));
  std::vector<hipStream_t> st(nstreams);
  for (auto& s : st) CK(hipStreamCreate(&s));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto run_main = [&](hipStream_t m) {
    CK(hipEventRecord(t0, m));
    for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(busy_kernel, dim3(256), dim3(256), 0, m, 1000L, sink);      // 10 us on one workgroup per CU
    CK(hipEventRecord(t1, m)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    return ms * 1e3 / nk;
  };
  for (int mi = 0; mi < 2 && mi < nstreams; ++mi) {
    CK(hipDeviceSynchronize());
    run_main(st[mi]);
    printf("main = stream %d alone: %.2f us per 10-us kernel\n", mi, run_main(st[mi]));
    for (int pi = 0; pi < nstreams; ++pi) {
      if (pi == mi) continue;
      CK(hipDeviceSynchronize());
      // partner: 64 workgroups x 256 threads for 50 us each, enough of them to outlast the main stream's run
      for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(256), 0, st[pi], 5000L, sink);
      const double us = run_main(st[mi]);
      CK(hipDeviceSynchronize());
      printf("   beside stream %d (50-us kernels back to back): %.2f us per kernel\n", pi, us);
    }
  }
  // the same with the partner's kernels SHORT (10 us): every boundary of the partner is a dispatch too
  CK(hipDeviceSynchronize());
  for (int pi = 1; pi < nstreams; ++pi) {
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(256), 0, st[pi], 1000L, sink);
    const double us = run_main(st[0]);
    CK(hipDeviceSynchronize());
    printf("main = stream 0 beside stream %d (10-us kernels back to back): %.2f us per kernel\n", pi, us);
  }
  return 0;
}
