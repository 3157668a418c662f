// Anatomy of one GEMM launch: where a workgroup of a short-k product spends its lifetime.  Needs an anatomy
// build of the library (bash tools/build_exp.sh anatomy -DAZ_ANATOMY): thread 0 of every workgroup stamps s_memtime at kernel
// entry, operand addresses ready, first DMA issued, first k-tile landed, loop left, epilogue stores issued / acknowledged.
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/gemm_anatomy tools/gemm_anatomy.cpp -ldl
//   run:   tools/gemm_anatomy aozora_sdxl_training_amd/lib_exp_anatomy.so M N K [tile:bm:bn:waves] [excl] [opt:NAME=V] [sets:N]
// Prints, over all workgroups of the LAST of several back-to-back launches (cold operand sets rotate): median / p10 / p90 of
// each segment in shader cycles and in microseconds (clock from s_memtime vs the 100 MHz s_memrealtime), the spread of kernel
// entry times across workgroups (launch ramp) and of exit times (tail), and the launch-to-launch period from HIP events.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef int (*gemm_fn)(int, int, int, int, int, const void*, long, const void*, long, void*, long, const void*, const void*, int, long,
                       const void*, long, int, int, void*, long, void*);
typedef int (*settile_fn)(int, int, int);
typedef int (*setopt_fn)(const char*, int);
__global__ void fill_kernel(unsigned short* p, long n, unsigned seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    float f = ((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;
    unsigned u = __float_as_uint(f);
    p[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
}
static void* rnd(long elems, unsigned seed, float scale) {
  void* p; CK(hipMalloc(&p, elems * 2));
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned short*)p, elems, seed, scale);
  return p;
}
static double pct(std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; }
int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: gemm_anatomy lib.so M N K [tile:bm:bn:w] [excl]\n"); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  gemm_fn gemm = (gemm_fn)dlsym(h, "az_gemm_bf16"); settile_fn settile = (settile_fn)dlsym(h, "az_gemm_set_tile_ex"); setopt_fn setopt = (setopt_fn)dlsym(h, "az_set_option");
  const long M = atol(argv[2]), N = atol(argv[3]), K = atol(argv[4]);
  int force_sets = 0, fix_a = 0, fix_b = 0;
  for (int i = 5; i < argc; ++i) {
    if (!strncmp(argv[i], "tile:", 5)) { int bm, bn, w; sscanf(argv[i] + 5, "%d:%d:%d", &bm, &bn, &w); settile(bm, bn, w); }
    if (!strcmp(argv[i], "excl")) setopt("LDS_EXCLUSIVE", 1);
    if (!strncmp(argv[i], "opt:", 4)) { char nm[64]; int v; if (sscanf(argv[i] + 4, "%63[^=]=%d", nm, &v) == 2) setopt(nm, v); }
    if (!strncmp(argv[i], "sets:", 5)) force_sets = atoi(argv[i] + 5);
    if (!strcmp(argv[i], "fixA")) fix_a = 1;      // the activation operand stays the same (warm) set, only the weights rotate
    if (!strcmp(argv[i], "fixB")) fix_b = 1;      // ... and the other way round      // sets:1 = the same (warm) operands every launch
  }
  const long WS = 64L << 20;
  void* ws; CK(hipMalloc(&ws, WS)); CK(hipMemset(ws, 0, WS));
  const int nset = force_sets > 0 ? force_sets : (int)std::min(48L, std::max(2L, (long)(600e6 / ((M * K + N * K + M * N) * 2)) + 1));
  std::vector<void*> A(nset), B(nset), C(nset);
  for (int s = 0; s < nset; ++s) { A[s] = rnd(M * K, 11 + s, 1.f); B[s] = rnd(N * K, 777 + s, 0.05f); CK(hipMalloc(&C[s], M * N * 2)); }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int s = 0; s < nset; ++s) { int rc = gemm(0, 1, M, N, K, A[fix_a ? 0 : s], K, B[fix_b ? 0 : s], K, C[s], N, nullptr, nullptr, 0, 0, nullptr, 0, 0, 1, ws, WS, st); if (rc) { printf("rc %d\n", rc); return 1; } }
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < 3; ++r) for (int s = 0; s < nset; ++s) gemm(0, 1, M, N, K, A[fix_a ? 0 : s], K, B[fix_b ? 0 : s], K, C[s], N, nullptr, nullptr, 0, 0, nullptr, 0, 0, 1, ws, WS, st);
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double period_us = ms * 1e3 / (3 * nset);
  // the workspace now holds the stamps of the LAST launch; find the workgroup count (entries with a non-zero entry stamp)
  std::vector<unsigned long> hbuf(WS / 8);
  CK(hipMemcpy(hbuf.data(), ws, WS, hipMemcpyDeviceToHost));
  size_t nwg = 0; while (nwg < WS / 128 && hbuf[nwg * 16] != 0) ++nwg;
  if (!nwg) { printf("no stamps: not an anatomy build?\n"); return 1; }
  // s_memtime is a per-XCD counter (different bases): segments are differences INSIDE a workgroup; everything across workgroups
  // (entry ramp, exit tail, span) uses the chip-wide 100 MHz s_memrealtime stamps [8] / [9]
  unsigned long rt_min = ~0ul, rt_max = 0;
  for (size_t w = 0; w < nwg; ++w) { rt_min = std::min(rt_min, hbuf[w * 16 + 8]); rt_max = std::max(rt_max, hbuf[w * 16 + 9]); }
  std::vector<double> seg[7], entry, exit_, clk;
  for (size_t w = 0; w < nwg; ++w) {
    const unsigned long* s = &hbuf[w * 16];
    for (int i = 0; i < 6; ++i) seg[i].push_back((double)(s[i + 1] - s[i]));
    seg[6].push_back((double)(s[6] - s[0]));
    entry.push_back((double)(s[8] - rt_min) / 100.0); exit_.push_back((double)(rt_max - s[9]) / 100.0);
    if (s[9] > s[8]) clk.push_back((double)(s[6] - s[0]) / ((double)(s[9] - s[8]) * 10.0));      // cycles per ns = GHz
  }
  const double ghz = pct(clk, 0.5), life = pct(seg[6], 0.5);
  const double span_us = (rt_max - rt_min) / 100.0;
  printf("product %ldx%ldx%ld: %zu workgroups; launch-to-launch period %.1f us (HIP events, %d cold operand sets); kernel span first entry -> last exit %.2f us; shader clock %.2f GHz (median over workgroups)\n",
         M, N, K, nwg, period_us, nset, span_us, ghz);
  const char* names[7] = {"entry -> operand addresses ready", "-> first k-tile's DMA issued", "-> first k-tile landed (wait + barrier)", "-> k-loop left (all MFMAs issued)",
                          "-> epilogue stores issued", "-> stores acknowledged (vmcnt 0)", "workgroup lifetime"};
  for (int i = 0; i < 7; ++i) printf("  %-42s median %8.0f cyc = %6.2f us   p10 %8.0f   p90 %8.0f   (%4.1f %% of the lifetime)\n", names[i], pct(seg[i], 0.5), pct(seg[i], 0.5) / ghz / 1e3,
                                     pct(seg[i], 0.1), pct(seg[i], 0.9), 100.0 * pct(seg[i], 0.5) / life);
  const long ktiles = (K + 63) / 64;
  printf("  k-loop: %.0f cycles per 64-deep k-tile\n", pct(seg[3], 0.5) / (double)std::max(1L, ktiles * (long)(nwg > 0 ? 1 : 1)));
  printf("  %-42s median %6.2f us   p90 %6.2f us   max %6.2f us\n", "kernel entry after the first workgroup's", pct(entry, 0.5), pct(entry, 0.9), pct(entry, 1.0));
  printf("  %-42s median %6.2f us   p90 %6.2f us   max %6.2f us\n", "exit before the last workgroup's", pct(exit_, 0.5), pct(exit_, 0.9), pct(exit_, 1.0));
  printf("  per launch outside the span (dispatch of the next launch + end-of-kernel): %.1f us\n", period_us - span_us);
  return 0;
}
