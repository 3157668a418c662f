// In-process A/B of GEMM / conv products between one or two builds of libaozora_hip.so (cdna_hip_programming.md 5.4 rule 24:
// interleaved rounds in ONE process), through the C ABI with no Python between launches.
//   build:  hipcc -O2 --offload-arch=gfx950 -o tools/gemm_ab tools/gemm_ab.cpp -ldl
//   run:    tools/gemm_ab <libA.so> [<libB.so>] -- <case> [<case> ...]
//   case:   nt:M:N:K[:split]         C[M,N] = A[M,K] . W[N,K]^T             (az_gemm_bf16 0,1)
//           tn:M:N:K[:split[:b]]     dW[M,N] = dY[K,M]^T . X[K,N] (+bias grad with :b)
//           gg:M:H:K                 fused GEGLU forward: proj[M,2H] = X[M,K] . W[2H,K]^T + b, out[M,H] = value * gelu(gate)
//           cf|cd|cw:B:H:W:Cin:Cout  3x3 stride-1 conv forward / dgrad (W^T form) / wgrad (+bias grad)
//           opt:NAME=V               az_set_option on every library from here on
//           tile:bm:bn:waves         az_gemm_set_tile_ex on every library from here on (0:0:0 = back to the heuristic)
// Operands cycle through enough distinct buffer sets (> 600 MB) that they come from HBM / the Infinity Cache as in the step.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef int (*gemm_fn)(int, int, int, int, int, const void*, long, const void*, long, void*, long, const void*, const void*, int, long,
                       const void*, long, int, int, void*, long, void*);
typedef int (*wgrad_fn)(int, int, int, const void*, long, const void*, long, void*, long, int, int, void*, long, void*, int, void*);
typedef int (*conv_fn)(int, int, int, int, int, int, int, int, int, int, int, int, const void*, long, const void*, const void*, long, void*,
                       long, const void*, const void*, long, const void*, long, int, int, void*, long, void*);
typedef int (*convwg_fn)(int, int, int, int, int, int, int, int, int, int, const void*, long, const void*, long, void*, int, int, void*, long,
                         void*, void*, void*);
typedef int (*geglu_fn)(int, int, int, const void*, long, const void*, long, const void*, void*, long, void*, long, void*);
typedef int (*setopt_fn)(const char*, int);
typedef int (*settile_fn)(int, int, int);

struct Lib { std::string name; geglu_fn geglu; gemm_fn gemm; wgrad_fn wgrad; conv_fn conv; convwg_fn convwg; setopt_fn setopt; settile_fn settile; };

static Lib load(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(2); }
  Lib l; l.name = path;
  l.gemm = (gemm_fn)dlsym(h, "az_gemm_bf16"); l.wgrad = (wgrad_fn)dlsym(h, "az_gemm_wgrad_bias_bf16");
  l.conv = (conv_fn)dlsym(h, "az_conv2d_bf16"); l.convwg = (convwg_fn)dlsym(h, "az_conv2d_wgrad_bias_bf16");
  l.geglu = (geglu_fn)dlsym(h, "az_gemm_geglu_fwd_bf16");
  l.setopt = (setopt_fn)dlsym(h, "az_set_option"); l.settile = (settile_fn)dlsym(h, "az_gemm_set_tile_ex");
  if (!l.gemm || !l.wgrad || !l.conv || !l.convwg || !l.setopt) { fprintf(stderr, "missing symbols in %s\n", path); exit(2); }
  return l;
}

__global__ void fill_kernel(unsigned short* p, long n, unsigned seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    float f = ((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;     // uniform [-scale, scale)
    unsigned u = __float_as_uint(f);
    p[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
}

static void* dalloc(long bytes) { void* p; CK(hipMalloc(&p, bytes)); return p; }
static void* rnd(long elems, unsigned seed, float scale) {
  void* p = dalloc(elems * 2);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (unsigned short*)p, elems, seed, scale);
  return p;
}

struct Case { std::string kind; long v[8]; int nv; };

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  std::vector<Lib> libs; int i = 1;
  for (; i < argc && strcmp(argv[i], "--"); ++i) libs.push_back(load(argv[i]));
  if (libs.empty() || i >= argc) { fprintf(stderr, "usage: gemm_ab libA.so [libB.so] -- cases...\n"); return 2; }
  const long WS = 512L << 20;
  std::vector<void*> wss;
  for (size_t l = 0; l < libs.size(); ++l) { void* w = dalloc(WS); CK(hipMemset(w, 0, WS)); wss.push_back(w); }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (++i; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("opt:", 0) == 0) {
      std::string kv = a.substr(4); size_t eq = kv.find('=');
      for (auto& l : libs) { int rc = l.setopt(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1)); if (rc) printf("  (option %s unknown to %s: %d)\n", kv.c_str(), l.name.c_str(), rc); }
      printf("option %s\n", kv.c_str());
      continue;
    }
    if (a.rfind("tile:", 0) == 0) {      // tile:bm:bn:waves -> az_gemm_set_tile_ex on every library (0:0:0 = heuristic)
      int bm = 0, bn = 0, w = 0; sscanf(a.c_str() + 5, "%d:%d:%d", &bm, &bn, &w);
      for (auto& l : libs) { int rc = l.settile(bm, bn, w); if (rc) printf("  (tile %d x %d / %d refused by %s: %d)\n", bm, bn, w, l.name.c_str(), rc); }
      printf("forced tile %d x %d / %d\n", bm, bn, w);
      continue;
    }
    Case c; c.nv = 0; size_t pos = a.find(':'); c.kind = a.substr(0, pos);
    bool with_b = false;
    while (pos != std::string::npos) {
      size_t nx = a.find(':', pos + 1);
      std::string tok = a.substr(pos + 1, nx == std::string::npos ? std::string::npos : nx - pos - 1);
      if (tok == "b") with_b = true; else c.v[c.nv++] = atol(tok.c_str());
      pos = nx;
    }
    double flops = 0; long set_bytes = 0;
    long M = 0, N = 0, K = 0; int split = 0;
    long B = 0, H = 0, W = 0, Cin = 0, Cout = 0;
    const bool is_conv = c.kind[0] == 'c';
    if (c.kind == "gg") { c.v[1] *= 2; }      // N = 2H
    if (!is_conv) { M = c.v[0]; N = c.v[1]; K = c.v[2]; split = c.nv > 3 ? (int)c.v[3] : (c.kind == "tn" ? 0 : 1); if (c.kind == "gg") split = 1; flops = 2.0 * M * N * K; set_bytes = (M * K + N * K + M * N) * 2; }
    else { B = c.v[0]; H = c.v[1]; W = c.v[2]; Cin = c.v[3]; Cout = c.v[4]; flops = 2.0 * B * H * W * Cout * 9 * Cin; set_bytes = (B * H * W * (Cin + Cout) + 9 * Cin * Cout) * 2; }
    int nset = (int)std::max(2L, (long)(600e6 / set_bytes) + 1); if (nset > 64) nset = 64;
    std::vector<void*> X(nset), Y(nset), Z(nset), BG(nset);
    for (int s = 0; s < nset; ++s) {
      if (c.kind == "gg") { X[s] = rnd(M * K, 11 + s, 1.f); Y[s] = rnd(N * K, 777 + s, 0.05f); Z[s] = dalloc(M * N * 2); BG[s] = dalloc(M * (N / 2) * 2); }
      else if (c.kind == "nt") { X[s] = rnd(M * K, 11 + s, 1.f); Y[s] = rnd(N * K, 777 + s, 0.05f); Z[s] = dalloc(M * N * 2); }
      else if (c.kind == "tn") { X[s] = rnd(K * M, 11 + s, 1.f); Y[s] = rnd(K * N, 777 + s, 1.f); Z[s] = dalloc(M * N * 2); CK(hipMemset(Z[s], 0, M * N * 2)); BG[s] = dalloc(M * 2); CK(hipMemset(BG[s], 0, M * 2)); }
      else {
        X[s] = rnd(B * H * W * Cin, 11 + s, 1.f);          // activations
        Y[s] = rnd(9 * Cin * Cout, 777 + s, 0.05f);        // weights (or dW target for cw)
        Z[s] = rnd(B * H * W * Cout, 999 + s, 1.f);        // Y / dY
        BG[s] = dalloc(Cout * 2); CK(hipMemset(BG[s], 0, Cout * 2));
      }
    }
    CK(hipDeviceSynchronize());
    auto call = [&](size_t li, int s) -> int {
      Lib& l = libs[li]; void* ws = wss[li];
      if (c.kind == "gg") return l.geglu ? l.geglu((int)M, (int)(N / 2), (int)K, X[s], K, Y[s], K, nullptr, Z[s], N, BG[s], N / 2, st) : -1;
      if (c.kind == "nt") return l.gemm(0, 1, (int)M, (int)N, (int)K, X[s], K, Y[s], K, Z[s], N, nullptr, nullptr, 0, 0, nullptr, 0, 0, split, ws, WS, st);
      if (c.kind == "tn") {
        if (with_b) return l.wgrad((int)M, (int)N, (int)K, X[s], M, Y[s], N, Z[s], N, 1, split, ws, WS, BG[s], (int)M, st);
        return l.gemm(1, 0, (int)M, (int)N, (int)K, X[s], M, Y[s], N, Z[s], N, nullptr, nullptr, 0, 0, nullptr, 0, 1, split, ws, WS, st);
      }
      if (c.kind == "cf") return l.conv(0, (int)B, (int)H, (int)W, (int)Cin, (int)H, (int)W, (int)Cout, 3, 1, 1, 0, X[s], Cin, Y[s], nullptr, 0, Z[s], Cout, nullptr, nullptr, 0, nullptr, 0, 0, 1, ws, WS, st);
      if (c.kind == "cd") return l.conv(3, (int)B, (int)H, (int)W, (int)Cin, (int)H, (int)W, (int)Cout, 3, 1, 1, 0, nullptr, 0, Y[s], Z[s], Cout, X[s], Cin, nullptr, nullptr, 0, nullptr, 0, 0, 1, ws, WS, st);
      if (c.kind == "cw") return l.convwg((int)B, (int)H, (int)W, (int)Cin, (int)H, (int)W, (int)Cout, 3, 1, 1, X[s], Cin, Z[s], Cout, Y[s], 1, 0, ws, WS, BG[s], nullptr, st);
      return -1;
    };
    const int ROUNDS = 5, REPS = 3;
    std::vector<std::vector<double>> us(libs.size());
    bool failed = false;
    for (size_t li = 0; li < libs.size() && !failed; ++li)
      for (int s = 0; s < nset; ++s) { int rc = call(li, s); if (rc) { printf("%s: %s returned %d\n", a.c_str(), libs[li].name.c_str(), rc); failed = true; break; } }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < ROUNDS && !failed; ++r)
      for (size_t li = 0; li < libs.size(); ++li) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < REPS; ++k) for (int s = 0; s < nset; ++s) call(li, s);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[li].push_back(ms * 1e3 / (REPS * nset));
      }
    if (!failed) {
      printf("%-34s", a.c_str());
      for (size_t li = 0; li < libs.size(); ++li) {
        std::sort(us[li].begin(), us[li].end());
        const double med = us[li][us[li].size() / 2];
        printf(" | %s: %8.1f us (min %8.1f) %7.1f TF/s", libs[li].name.c_str(), med, us[li][0], flops / med / 1e6);
      }
      printf("\n"); fflush(stdout);
    }
    for (int s = 0; s < nset; ++s) { hipFree(X[s]); hipFree(Y[s]); hipFree(Z[s]); if (BG[s]) hipFree(BG[s]); }
  }
  return 0;
}
