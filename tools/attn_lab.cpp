// In-process A/B of the attention kernels under different runtime options (cdna_hip_programming.md 5.4 rule 24: interleaved
// rounds in ONE process), through the C ABI with no Python between launches, on uniform random data, rotating buffer sets.
//   build:  hipcc -O2 --offload-arch=gfx950 -o tools/attn_lab tools/attn_lab.cpp -ldl
//   run:    tools/attn_lab <lib.so> -- [rounds:N] var:NAME=V[,NAME=V] [var:...] shape:B:heads:Tq:Tk [shape:...]
// The first variant is the reference the others are compared with (relative Frobenius / max abs difference of O, lse, dQ, dK, dV);
// the kernels themselves are checked against fp32 torch by tests/test_kernels_gpu.py and tests/test_realshape_gpu.py.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef int (*fwd_fn)(int, int, int, int, float, const void*, long, long, const void*, long, long, const void*, long, long, void*, long, long, void*, void*);
typedef int (*bwd_fn)(int, int, int, int, float, const void*, long, long, const void*, long, long, const void*, long, long, const void*, long, long,
                      const void*, long, long, const void*, void*, void*, long, long, void*, long, long, void*, long, long, void*, long, int, void*);
typedef int (*setopt_fn)(const char*, int);

__global__ void fill_kernel(unsigned short* p, long n, unsigned seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    float f = ((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;
    unsigned u = __float_as_uint(f);
    p[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
}
// one key row of one head made large: its score dominates from that key tile on (forces the deferred-maximum rescale branch)
__global__ void spike_kernel(unsigned short* k, long ld, long sb, int batch, int heads, int row, float v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * heads * 64) return;
  int d = i & 63, h = (i >> 6) % heads, b = i / (64 * heads);
  unsigned u = __float_as_uint((d & 1) ? v : -v);
  k[b * sb + (long)row * ld + h * 64 + d] = (unsigned short)(u >> 16);
}

static void* dalloc(long bytes) { void* p; CK(hipMalloc(&p, bytes)); return p; }
static unsigned short* rnd(long elems, unsigned seed, float scale) {
  unsigned short* p = (unsigned short*)dalloc(elems * 2);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p, elems, seed, scale);
  return p;
}
static float bf2f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

struct Set {
  unsigned short *q, *k, *v, *o, *dout, *dq, *dk, *dv; float *lse, *delta;
  long ldq, sq, ldk, sk, ldo, so;
};
struct Var { std::string name; std::vector<std::pair<std::string, int>> opts; };

static void diff_bf16(const char* what, const unsigned short* a_d, const unsigned short* b_d, long rows, long cols, long ld) {
  const long n = (rows - 1) * ld + cols;      // the operand may be a column slice of a wider buffer
  std::vector<unsigned short> a(n), b(n);
  CK(hipMemcpy(a.data(), a_d, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), b_d, n * 2, hipMemcpyDeviceToHost));
  double num = 0, den = 0, mx = 0; long bad = 0;
  for (long r = 0; r < rows; ++r) for (long c = 0; c < cols; ++c) {
    float x = bf2f(a[r * ld + c]), y = bf2f(b[r * ld + c]);
    if (!(std::isfinite(x) && std::isfinite(y))) { ++bad; continue; }
    num += (double)(x - y) * (x - y); den += (double)x * x; mx = std::max(mx, (double)fabsf(x - y));
  }
  printf("    %-4s rel-fro %.2e  max-abs %.2e%s\n", what, sqrt(num / (den + 1e-30)), mx, bad ? "  NON-FINITE VALUES" : "");
}
static void diff_f32(const char* what, const float* a_d, const float* b_d, long n) {
  std::vector<float> a(n), b(n);
  CK(hipMemcpy(a.data(), a_d, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), b_d, n * 4, hipMemcpyDeviceToHost));
  double mx = 0; long bad = 0;
  for (long i = 0; i < n; ++i) { if (!std::isfinite(a[i]) || !std::isfinite(b[i])) { ++bad; continue; } mx = std::max(mx, (double)fabsf(a[i] - b[i])); }
  printf("    %-4s max-abs %.2e%s\n", what, mx, bad ? "  NON-FINITE VALUES" : "");
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  if (argc < 4) { fprintf(stderr, "usage: attn_lab lib.so -- var:... shape:...\n"); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  fwd_fn fwd = (fwd_fn)dlsym(h, "az_attn_fwd"); bwd_fn bwd = (bwd_fn)dlsym(h, "az_attn_bwd"); setopt_fn setopt = (setopt_fn)dlsym(h, "az_set_option");
  if (!fwd || !bwd || !setopt) { fprintf(stderr, "missing symbols\n"); return 2; }
  std::vector<Var> vars; std::vector<std::vector<long>> shapes; int rounds = 7; int spike = 0;
  for (int i = 3; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("var:", 0) == 0) {
      Var v; v.name = a.substr(4); std::string rest = v.name;
      while (!rest.empty()) {
        size_t c = rest.find(','); std::string one = rest.substr(0, c); rest = c == std::string::npos ? "" : rest.substr(c + 1);
        size_t e = one.find('='); v.opts.push_back({one.substr(0, e), atoi(one.substr(e + 1).c_str())});
      }
      vars.push_back(v);
    } else if (a.rfind("shape:", 0) == 0) {
      std::vector<long> s; std::string rest = a.substr(6);
      while (!rest.empty()) { size_t c = rest.find(':'); s.push_back(atol(rest.substr(0, c).c_str())); rest = c == std::string::npos ? "" : rest.substr(c + 1); }
      shapes.push_back(s);
    } else if (a.rfind("rounds:", 0) == 0) rounds = atoi(a.c_str() + 7);
    else if (a.rfind("spike:", 0) == 0) spike = atoi(a.c_str() + 6);
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long WS = 256L << 20; void* ws = dalloc(WS);
  for (auto& sh : shapes) {
    const int B = sh[0], H = sh[1], Tq = sh[2], Tk = sh[3]; const long C = H * 64;
    const int nset = 6;
    std::vector<Set> sets(nset);
    for (int i = 0; i < nset; ++i) {
      Set& s = sets[i];
      if (Tq == Tk) {
        unsigned short* qkv = rnd((long)B * Tq * 3 * C, 1000 + i, 1.7f); unsigned short* dqkv = (unsigned short*)dalloc((long)B * Tq * 3 * C * 2);
        s.q = qkv; s.k = qkv + C; s.v = qkv + 2 * C; s.ldq = s.ldk = 3 * C; s.sq = s.sk = (long)Tq * 3 * C;
        s.dq = dqkv; s.dk = dqkv + C; s.dv = dqkv + 2 * C;
      } else {
        s.q = rnd((long)B * Tq * C, 2000 + i, 1.7f); unsigned short* kv = rnd((long)B * Tk * 2 * C, 3000 + i, 1.7f);
        s.k = kv; s.v = kv + C; s.ldq = C; s.sq = (long)Tq * C; s.ldk = 2 * C; s.sk = (long)Tk * 2 * C;
        s.dq = (unsigned short*)dalloc((long)B * Tq * C * 2); unsigned short* dkv = (unsigned short*)dalloc((long)B * Tk * 2 * C * 2);
        s.dk = dkv; s.dv = dkv + C;
      }
      if (spike > 0 && spike < Tk)
        hipLaunchKernelGGL(spike_kernel, dim3((B * H * 64 + 255) / 256), dim3(256), 0, 0, s.k, s.ldk, s.sk, B, H, spike, 6.0f);
      s.o = (unsigned short*)dalloc((long)B * Tq * C * 2); s.ldo = C; s.so = (long)Tq * C;
      s.dout = rnd((long)B * Tq * C, 4000 + i, 1.0f);
      s.lse = (float*)dalloc((long)B * H * Tq * 4); s.delta = (float*)dalloc((long)B * H * Tq * 4);
    }
    CK(hipDeviceSynchronize());
    auto apply = [&](const Var& v) { for (auto& o : v.opts) if (setopt(o.first.c_str(), o.second)) { fprintf(stderr, "bad option %s\n", o.first.c_str()); exit(2); } };
    auto run_fwd = [&](Set& s) { int rc = fwd(B, H, Tq, Tk, 0.125f, s.q, s.ldq, s.sq, s.k, s.ldk, s.sk, s.v, s.ldk, s.sk, s.o, s.ldo, s.so, s.lse, st);
                                 if (rc) { fprintf(stderr, "az_attn_fwd rc %d\n", rc); exit(2); } };
    auto run_bwd = [&](Set& s) { int rc = bwd(B, H, Tq, Tk, 0.125f, s.q, s.ldq, s.sq, s.k, s.ldk, s.sk, s.v, s.ldk, s.sk, s.o, s.ldo, s.so, s.dout, s.ldo, s.so,
                                               s.lse, s.delta, s.dq, s.ldq, s.sq, s.dk, s.ldk, s.sk, s.dv, s.ldk, s.sk, ws, WS, 0, st);
                                 if (rc) { fprintf(stderr, "az_attn_bwd rc %d\n", rc); exit(2); } };
    printf("shape %d x %d heads, Tq %d, Tk %d%s\n", B, H, Tq, Tk, spike ? "  (spiked key row)" : "");
    // correctness: variant 0 on set 0, every other variant on set 1 with set 0's inputs copied
    {
      Set& a = sets[0]; Set& b = sets[1];
      const long qbytes = (Tq == Tk) ? (long)B * Tq * 3 * C * 2 : (long)B * Tq * C * 2;
      CK(hipMemcpy(b.q, a.q, qbytes, hipMemcpyDeviceToDevice));
      if (Tq != Tk) CK(hipMemcpy(b.k, a.k, (long)B * Tk * 2 * C * 2, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(b.dout, a.dout, (long)B * Tq * C * 2, hipMemcpyDeviceToDevice));
      apply(vars[0]); run_fwd(a); run_bwd(a); CK(hipStreamSynchronize(st));
      for (size_t v = 1; v < vars.size(); ++v) {
        apply(vars[v]);
        CK(hipMemsetAsync(b.o, 0xFF, (long)B * Tq * C * 2, st));
        run_fwd(b); run_bwd(b); CK(hipStreamSynchronize(st));
        printf("  %s vs %s\n", vars[v].name.c_str(), vars[0].name.c_str());
        diff_bf16("O", a.o, b.o, (long)B * Tq, C, C);
        diff_f32("lse", a.lse, b.lse, (long)B * H * Tq);
        diff_bf16("dQ", a.dq, b.dq, (long)B * Tq, C, a.ldq);
        diff_bf16("dK", a.dk, b.dk, (long)B * Tk, C, a.ldk);
        diff_bf16("dV", a.dv, b.dv, (long)B * Tk, C, a.ldk);
      }
      CK(hipMemcpy(b.q, sets[2].q, qbytes, hipMemcpyDeviceToDevice));   // de-duplicate the sets again
    }
    // timing
    const double fl = 4.0 * B * H * (double)Tq * Tk * 64;
    std::vector<std::vector<float>> tf(vars.size()), tb(vars.size());
    for (int r = 0; r < rounds + 1; ++r)
      for (size_t v = 0; v < vars.size(); ++v) {
        apply(vars[v]);
        for (int pass = 0; pass < 2; ++pass) {
          CK(hipEventRecord(e0, st));
          for (int rep = 0; rep < 2; ++rep) for (auto& s : sets) { if (pass == 0) run_fwd(s); else run_bwd(s); }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (r > 0) (pass == 0 ? tf : tb)[v].push_back(ms * 1e3f / (2 * nset));
        }
      }
    for (size_t v = 0; v < vars.size(); ++v) {
      std::sort(tf[v].begin(), tf[v].end()); std::sort(tb[v].begin(), tb[v].end());
      const float f = tf[v][tf[v].size() / 2], b = tb[v][tb[v].size() / 2];
      printf("  %-28s fwd %7.1f us (min %7.1f) %6.0f TF/s   bwd %7.1f us (min %7.1f) %6.0f TF/s\n", vars[v].name.c_str(), f, tf[v][0], fl / f / 1e6,
             b, tb[v][0], 2.5 * fl / b / 1e6);
    }
    for (auto& s : sets) { /* leak: the process ends after the last shape */ (void)s; }
  }
  return 0;
}
