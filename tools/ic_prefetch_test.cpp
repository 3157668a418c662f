// Premise test for a weight prefetch into the Infinity Cache: NT products C[M,N] = A[M,K] . W[N,K]^T over rotating operand sets
// (> 600 MB: HBM-cold, as the step's weights are), with
//   a: the activation operand A written by a producer kernel right before the product (as in the step) or left cold
//   w: the NEXT product's weight operand read by a small kernel on a second stream while the current product runs
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/ic_prefetch_test tools/ic_prefetch_test.cpp -ldl
//   run:   tools/ic_prefetch_test lib.so M N K
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef int (*gemm_fn)(int, int, int, int, int, const void*, long, const void*, long, void*, long, const void*, const void*, int, long,
                       const void*, long, int, int, void*, long, void*);
__global__ void fill_kernel(unsigned short* p, long n, unsigned seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    float f = ((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;
    p[i] = (unsigned short)(__float_as_uint(f) >> 16);
  }
}
// producer: dst = src (16-byte pieces), the stand-in for the kernel that writes the activation in front of the product
__global__ void copy_kernel(const uint4* __restrict__ s, uint4* __restrict__ d, long n16) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) d[i] = s[i];
}
// prefetch: read every 128-byte line once (one dword per line), keep nothing
__global__ void touch_kernel(const unsigned* __restrict__ p, long lines, unsigned* sink) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < lines; i += stride) acc ^= p[i * 32];
  if (acc == 0x9E3779B9u) *sink = acc;
}
int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: ic_prefetch_test lib.so M N K [pfwg] [group]\n"); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  gemm_fn gemm = (gemm_fn)dlsym(h, "az_gemm_bf16");
  const long M = atol(argv[2]), N = atol(argv[3]), K = atol(argv[4]);
  const int pfwg = argc > 5 ? atoi(argv[5]) : 16;
  const int G = argc > 6 ? atoi(argv[6]) : 8;      // products per prefetch group
  const long WS = 64L << 20; void* ws; CK(hipMalloc(&ws, WS));
  const int nset = (int)std::min(64L, std::max(4L, (long)(900e6 / ((M * K + N * K + M * N) * 2)) + 1));
  std::vector<void*> A(nset), B(nset), C(nset); void* Asrc; unsigned* sink;
  CK(hipMalloc(&Asrc, M * K * 2)); CK(hipMalloc(&sink, 4));
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned short*)Asrc, M * K, 5u, 1.f);
  for (int s = 0; s < nset; ++s) {
    CK(hipMalloc(&A[s], M * K * 2)); CK(hipMalloc(&B[s], N * K * 2)); CK(hipMalloc(&C[s], M * N * 2));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned short*)A[s], M * K, 11u + s, 1.f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned short*)B[s], N * K, 777u + s, 0.05f);
  }
  hipStream_t st, pf; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&pf));
  std::vector<hipEvent_t> ev(nset); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[6] = {"A cold, W cold", "A produced in front, W cold", "A produced, W of the next product touched on a 2nd stream",
                          "A cold, next W touched", "A cold, next A and W touched", "producer alone (to subtract)"};
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 6; ++mode) {
    CK(hipDeviceSynchronize());
    const int rounds = 3;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < rounds; ++r)
      for (int s = 0; s < nset; ++s) {
        const bool produce = (mode == 1 || mode == 2 || mode == 5);
        if (produce) hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st, (const uint4*)Asrc, (uint4*)A[s], M * K / 8);
        if (mode == 5) continue;
        if (mode >= 2 && s % G == 0) {      // the touch of the next group's operands starts when this group starts
          CK(hipEventRecord(ev[s], st)); CK(hipStreamWaitEvent(pf, ev[s], 0));
          for (int g = 0; g < G; ++g) {
            const int nx = (s + G + g) % nset;
            hipLaunchKernelGGL(touch_kernel, dim3(pfwg), dim3(256), 0, pf, (const unsigned*)B[nx], N * K * 2 / 128, sink);
            if (mode == 4) hipLaunchKernelGGL(touch_kernel, dim3(pfwg), dim3(256), 0, pf, (const unsigned*)A[nx], M * K * 2 / 128, sink);
          }
        }
        gemm(0, 1, M, N, K, A[s], K, B[s], K, C[s], N, nullptr, nullptr, 0, 0, nullptr, 0, 0, 1, ws, WS, st);
      }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipStreamSynchronize(pf));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("  %-62s %7.2f us per product (%d sets)\n", names[mode], ms * 1e3 / (rounds * nset), nset);
  }
  return 0;
}
