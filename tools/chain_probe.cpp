// Producer -> LayerNorm -> consumer chain of the 1280-wide transformer level through the C ABI, launched from C (no interpreter, no
// torch in the process): out-projection + bias + residual (4096x1280x1280), LayerNorm, q|k|v projection (4096x3840x1280); 8 rotating
// buffer sets.  Launch-to-launch time of the chain, and of each kernel kind alone in the same rotation, per value of the options given.
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/chain_probe tools/chain_probe.cpp -ldl
//   run:   tools/chain_probe <lib.so> [OPTION=V ...]      (each OPTION=V is one more variant after the defaults)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef int (*gemm_fn)(int, int, int, int, int, const void*, long, const void*, long, void*, long, const void*, const void*, int, long, const void*, long, int, int, void*, long, void*);
typedef int (*ln_fn)(int, int, float, const void*, long, const void*, const void*, void*, long, void*, void*);
typedef int (*opt_fn)(const char*, int);
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: chain_probe lib.so [OPTION=V ...]\n"); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW); if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  gemm_fn gemm = (gemm_fn)dlsym(h, "az_gemm_bf16"); ln_fn ln = (ln_fn)dlsym(h, "az_layernorm_fwd"); opt_fn setopt = (opt_fn)dlsym(h, "az_set_option");
  const int M = 4096, C = 1280, NS = 8, REPS = 400;
  struct Set { unsigned short *a, *r, *x, *y, *z; float* st; } s[NS];
  auto alloc = [](size_t n) { unsigned short* p; CK(hipMalloc(&p, n * 2)); CK(hipMemset(p, 0x3c, n * 2)); return p; };
  for (int i = 0; i < NS; ++i) { s[i].a = alloc((size_t)M * C); s[i].r = alloc((size_t)M * C); s[i].x = alloc((size_t)M * C); s[i].y = alloc((size_t)M * C); s[i].z = alloc((size_t)M * 3 * C); CK(hipMalloc(&s[i].st, M * 8)); }
  unsigned short *w1 = alloc((size_t)C * C), *w2 = alloc((size_t)3 * C * C), *bias = alloc(C), *g = alloc(C), *b = alloc(C);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](int what) {      // 0: whole chain, 1 / 2 / 3: one kernel kind alone
    float best = 1e9f;
    for (int w = 0; w < 3; ++w) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < REPS; ++i) {
        Set& t = s[i % NS];
        int rc = 0;
        if (what == 0 || what == 1) rc |= gemm(0, 1, M, C, C, t.a, C, w1, C, t.x, C, bias, nullptr, 0, 0, t.r, C, 0, 1, nullptr, 0, nullptr);
        if (what == 0 || what == 2) rc |= ln(M, C, 1e-5f, t.x, C, g, b, t.y, C, t.st, nullptr);
        if (what == 0 || what == 3) rc |= gemm(0, 1, M, 3 * C, C, t.y, C, w2, C, t.z, 3 * C, nullptr, nullptr, 0, 0, nullptr, 0, 0, 1, nullptr, 0, nullptr);
        if (rc) { fprintf(stderr, "entry point failed: %d\n", rc); exit(1); }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    return best * 1e3 / REPS;
  };
  for (int v = 1; v <= argc - 1; ++v) {
    std::string name = "defaults";
    if (v >= 2) { name = argv[v]; size_t eq = name.find('='); if (setopt(name.substr(0, eq).c_str(), atoi(name.c_str() + eq + 1))) { fprintf(stderr, "unknown option %s\n", name.c_str()); return 2; } }
    const double c = timeit(0), k1 = timeit(1), k2 = timeit(2), k3 = timeit(3);
    printf("%-14s chain %6.2f us   (alone, same rotation: out-projection %6.2f   LayerNorm %6.2f   q|k|v projection %6.2f   sum %6.2f)\n", name.c_str(), c, k1, k2, k3, k1 + k2 + k3);
  }
  return 0;
}
