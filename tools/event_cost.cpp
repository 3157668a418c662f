// What an event record between two kernels of one stream costs that stream, by event flags (the executor forks a parameter-gradient
// launch behind almost every data-gradient kernel: one hipEventRecord on the chain stream + one hipStreamWaitEvent on the branch).
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/event_cost tools/event_cost.cpp
//   run:   tools/event_cost [MB per kernel] [kernels]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
__global__ void stream_kernel(const uint4* __restrict__ s, uint4* __restrict__ d, long n16) {      // a ~10-20 us memory-bound stand-in
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) { uint4 v = s[i]; v.x += 1; d[i] = v; }
}
__global__ void small_kernel(const unsigned* __restrict__ s, unsigned* d) { if (threadIdx.x == 0 && blockIdx.x == 0) d[0] = s[0] + 1; }
int main(int argc, char** argv) {
  const long mb = argc > 1 ? atol(argv[1]) : 20; const int nk = argc > 2 ? atoi(argv[2]) : 400;
  const long n16 = mb * (1L << 20) / 16;
  const int nbuf = 24;
  std::vector<uint4*> buf(nbuf); for (auto& b : buf) { CK(hipMalloc(&b, n16 * 16)); CK(hipMemset(b, 1, n16 * 16)); }
  unsigned* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  unsigned* sig; CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); unsigned sigval = 0;
  hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  struct Case { const char* name; int mode; unsigned flags; };
  const Case cases[] = {
    {"kernels back to back, no events", 0, 0},
    {"+ record (default flags)", 1, hipEventDefault},
    {"+ record (DisableTiming)", 1, hipEventDisableTiming},
    {"+ record (DisableTiming | ReleaseToDevice)", 1, hipEventDisableTiming | hipEventReleaseToDevice},
    {"+ record (DisableTiming | DisableSystemFence)", 1, hipEventDisableTiming | hipEventDisableSystemFence},
    {"+ record (DisableTiming) + wait and a small kernel on a 2nd stream", 2, hipEventDisableTiming},
    {"+ record (DisableTiming | ReleaseToDevice) + wait and small kernel on 2nd stream", 2, hipEventDisableTiming | hipEventReleaseToDevice},
    {"+ record (DisableTiming | DisableSystemFence) + wait and small kernel on 2nd stream", 2, hipEventDisableTiming | hipEventDisableSystemFence},
    {"kernel launched with hipExtLaunchKernelGGL(stopEvent) (DisableTiming | DisableSystemFence), nobody waits", 6, hipEventDisableTiming | hipEventDisableSystemFence},
    {"kernel launched with hipExtLaunchKernelGGL(stopEvent) + wait and small kernel on 2nd stream", 7, hipEventDisableTiming | hipEventDisableSystemFence},
    {"kernel launched with hipExtLaunchKernelGGL(stopEvent, DisableTiming only) + wait and small kernel on 2nd stream", 7, hipEventDisableTiming},
    {"+ hipStreamWriteValue32 (no event)", 4, 0},
    {"+ hipStreamWriteValue32 + hipStreamWaitValue32 (>=) and a small kernel on a 2nd stream", 5, 0},
    {"+ record + 2nd-stream kernel, and the chain waits for the 2nd stream every 8th kernel (DisableTiming)", 3, hipEventDisableTiming},
    {"+ record + 2nd-stream kernel, and the chain waits for the 2nd stream every 8th kernel (ReleaseToDevice)", 3, hipEventDisableTiming | hipEventReleaseToDevice},
  };
  for (int rep = 0; rep < 2; ++rep)
  for (const Case& c : cases) {
    std::vector<hipEvent_t> ev(nk), ev2(nk);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, c.flags ? c.flags : hipEventDisableTiming));
    for (auto& e : ev2) CK(hipEventCreateWithFlags(&e, c.flags ? c.flags : hipEventDisableTiming));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, a));
    for (int i = 0; i < nk; ++i) {
      if (c.mode == 6 || c.mode == 7) {
        hipExtLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, a, nullptr, ev[i], 0, (const uint4*)buf[i % nbuf], buf[(i + 7) % nbuf], n16);
        if (c.mode == 7) { CK(hipStreamWaitEvent(b, ev[i], 0)); hipLaunchKernelGGL(small_kernel, dim3(1), dim3(64), 0, b, flag, flag + 8); }
        continue;
      }
      hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, a, buf[i % nbuf], buf[(i + 7) % nbuf], n16);
      if (c.mode == 4 || c.mode == 5) {
        ++sigval;
        CK(hipStreamWriteValue32(a, sig, sigval, 0));
        if (c.mode == 5) { CK(hipStreamWaitValue32(b, sig, sigval, hipStreamWaitValueGte, 0xFFFFFFFFu)); hipLaunchKernelGGL(small_kernel, dim3(1), dim3(64), 0, b, flag, flag + 8); }
        continue;
      }
      if (c.mode >= 1) CK(hipEventRecord(ev[i], a));
      if (c.mode >= 2) { CK(hipStreamWaitEvent(b, ev[i], 0)); hipLaunchKernelGGL(small_kernel, dim3(1), dim3(64), 0, b, flag, flag + 8); }
      if (c.mode >= 3 && (i & 7) == 7) { CK(hipEventRecord(ev2[i], b)); CK(hipStreamWaitEvent(a, ev2[i], 0)); }
    }
    CK(hipEventRecord(t1, a)); CK(hipEventSynchronize(t1)); CK(hipStreamSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    if (rep) printf("  %-105s %7.2f us per kernel\n", c.name, ms * 1e3 / nk);
    for (auto& e : ev) CK(hipEventDestroy(e));
    for (auto& e : ev2) CK(hipEventDestroy(e));
  }
  return 0;
}
