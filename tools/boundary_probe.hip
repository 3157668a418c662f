// What a kernel boundary costs on one in-order stream: N back-to-back launches of (a) an empty kernel, (b) a kernel whose every
// workgroup reads and writes a slice of a buffer (so each launch has dirty lines to write back and its successor misses), at the
// grid sizes of the step's chain (256 / 1024 workgroups of 256 lanes).  Launch-to-launch period from HIP events over the whole run.
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/boundary_probe tools/boundary_probe.hip ; run on the GPU box, no arguments
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
// every lane moves `per_lane` 16-byte pieces of its workgroup's slice from src to dst (+1), coalesced
__global__ __launch_bounds__(256) void touch_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int per_lane) {
  const long base = ((long)blockIdx.x * per_lane) * 256 + threadIdx.x;
  for (int i = 0; i < per_lane; ++i) { uint4 v = src[base + (long)i * 256]; v.x += 1; dst[base + (long)i * 256] = v; }
}

// LayerNorm-shaped traffic: one wave per row of C = 1280 bf16 (160 pieces of 16 B: lanes 0-63 twice, lanes 0-31 a third time), read
// and written back; REDUCE: two dependent wave reductions between the loads and the stores, as a LayerNorm has
template <int REDUCE, int ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void row_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int M, int shift) {
  const int lane = threadIdx.x & 63;
  const int blk = (blockIdx.x + shift) % gridDim.x;          // shift != 0: the rows a workgroup (and its XCD) handles move from launch to launch
  const int row0 = (blk * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
#pragma unroll
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int row = row0 + rr;
    if (row >= M) return;
    const uint4* s = src + (long)row * 160; uint4* d = dst + (long)row * 160;
    uint4 v0 = s[lane], v1 = s[lane + 64], v2 = lane < 32 ? s[lane + 128] : make_uint4(0, 0, 0, 0);
    if (REDUCE >= 2) {      // every wave also reads the SAME two 2.5-KB vectors (a LayerNorm's gamma and beta), from the end of the source buffer
      const uint4* gsh = src + (long)M * 160;
      const uint4 g0 = gsh[lane], g1 = gsh[lane + 64], b0 = gsh[160 + lane], b1 = gsh[160 + lane + 64];
      v0.y += g0.x ^ b0.x; v1.y += g1.x ^ b1.x;
      if (lane < 32) { const uint4 g2 = gsh[lane + 128], b2 = gsh[160 + lane + 128]; v2.y += g2.x ^ b2.x; }
    }
    if (REDUCE) {
      float f = __uint_as_float(v0.x) + __uint_as_float(v1.y) + __uint_as_float(v2.z);
      for (int o = 32; o > 0; o >>= 1) f += __shfl_xor(f, o, 64);
      float g = (__uint_as_float(v0.y) - f) * (__uint_as_float(v1.x) - f);
      for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
      v0.x += (unsigned)(g != 0.f);
    }
    v0.w += 1;
    d[lane] = v0; d[lane + 64] = v1; if (lane < 32) d[lane + 128] = v2;
  }
}
template <int REDUCE, int RPW> static double run_rows(int n, int M, uint4* a, uint4* b, int step = 0) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (M / RPW + 3) / 4;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((row_kernel<REDUCE, RPW>), dim3(grid), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, M, (i * step) % grid);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / n;
}

// A producer that, like a GEMM, streams other bytes through the L2 before it writes its rows: every workgroup first reads `junk16` 16-byte
// pieces per lane from its own slice of a junk buffer (summed into a value that goes out with the rows), then copies its 4 rows.
__global__ __launch_bounds__(256) void producer_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const uint4* __restrict__ junk, int junk16, int M) {
  const int lane = threadIdx.x & 63;
  unsigned acc = 0;
  const uint4* j = junk + (long)blockIdx.x * junk16 * 256 + threadIdx.x;
  for (int i = 0; i < junk16; ++i) acc += j[(long)i * 256].x;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const uint4* s = src + (long)row * 160; uint4* d = dst + (long)row * 160;
  uint4 v0 = s[lane], v1 = s[lane + 64]; v0.w += acc;
  d[lane] = v0; d[lane + 64] = v1; if (lane < 32) d[lane + 128] = s[lane + 128];
}
// the same pair, but the producer writes into one of 8 rotating buffers whose lines were last touched 7 pairs (84 MB of writes) ago:
// its stores MISS in the L2 (in run_pair they hit lines the previous consumer launch had read)
static double run_pair_fresh(int n, int M, uint4* a, uint4* fresh8, int shift) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (M + 3) / 4;
  const long stride = (long)M * 160 + 1024;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) {
      uint4* dst = fresh8 + (i & 7) * stride;
      hipLaunchKernelGGL(producer_kernel, dim3(grid), dim3(256), 0, 0, a, dst, a, 0, M);
      hipLaunchKernelGGL((row_kernel<2, 1>), dim3(grid), dim3(256), 0, 0, dst, a, M, shift);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / n;
}
static double run_pair(int n, int M, uint4* a, uint4* b, const uint4* junk, int junk16, int shift) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (M + 3) / 4;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) {
      hipLaunchKernelGGL(producer_kernel, dim3(grid), dim3(256), 0, 0, a, b, junk, junk16, M);
      hipLaunchKernelGGL((row_kernel<2, 1>), dim3(grid), dim3(256), 0, 0, b, a, M, shift);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / n;
}

static double run(int n, int grid, int per_lane, uint4* a, uint4* b) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) {
      if (per_lane == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, (int*)nullptr);
      else hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, per_lane);   // each launch reads what its predecessor wrote
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / n;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'p') {      // "pinned": the process holds pinned host memory and has copied from it, as a torch process does
    void* hp; CK(hipHostMalloc(&hp, 64u << 20)); void* dp; CK(hipMalloc(&dp, 64u << 20));
    CK(hipMemcpyAsync(dp, hp, 64u << 20, hipMemcpyHostToDevice, 0)); CK(hipDeviceSynchronize());
    printf("(process holds 64 MiB of pinned host memory)\n");
  }
  const size_t bytes = 256u << 20;
  uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  const int n = 2000;
  for (int grid : {256, 1024}) {
    printf("grid %4d x 256 lanes, %d dependent launches on one stream:\n", grid, n);
    printf("  empty kernel                               %6.2f us per launch\n", run(n, grid, 0, a, b));
    for (int per_lane : {1, 8, 40}) {
      const double mb = (double)grid * 256 * per_lane * 16 / 1e6;
      const double us = run(n, grid, per_lane, a, b);
      printf("  moves %6.1f MB in and %6.1f MB out         %6.2f us per launch  (%5.2f TB/s)\n", mb, mb, us, 2 * mb / us);
    }
  }
  printf("LayerNorm-shaped rows (4096 x 1280 bf16 = 10.5 MB in, 10.5 MB out), one wave per row unless said:\n");
  printf("  copy only                                   %6.2f us per launch\n", run_rows<0, 1>(n, 4096, a, b));
  printf("  + two dependent wave reductions             %6.2f us per launch\n", run_rows<1, 1>(n, 4096, a, b));
  printf("  + reductions + the same 2 x 2.5 KB read by every wave %6.2f us per launch\n", run_rows<2, 1>(n, 4096, a, b));
  printf("  copy only, 2 rows per wave                  %6.2f us per launch\n", run_rows<0, 2>(n, 4096, a, b));
  printf("  + reductions, 2 rows per wave               %6.2f us per launch\n", run_rows<1, 2>(n, 4096, a, b));
  printf("  ... rows move to another XCD every launch (what a consumer of another kernel's output sees):\n");
  printf("  copy only                                   %6.2f us per launch\n", run_rows<0, 1>(n, 4096, a, b, 3));
  printf("  + two dependent wave reductions             %6.2f us per launch\n", run_rows<1, 1>(n, 4096, a, b, 3));
  printf("  + reductions, 2 rows per wave               %6.2f us per launch\n", run_rows<1, 2>(n, 4096, a, b, 3));
  printf("  + reductions + the same 2 x 2.5 KB read by every wave %6.2f us per launch\n", run_rows<2, 1>(n, 4096, a, b, 3));
  printf("  copy only, 4 rows per wave                  %6.2f us per launch\n", run_rows<0, 4>(n, 4096, a, b));
  printf("  + reductions, 4 rows per wave               %6.2f us per launch\n", run_rows<1, 4>(n, 4096, a, b));
  uint4* junk; CK(hipMalloc(&junk, 512u << 20)); CK(hipMemset(junk, 1, 512u << 20));
  printf("producer (streams junk through the L2, then writes its 4 rows) -> LayerNorm-shaped consumer of the same rows; us per PAIR of launches:\n");
  for (int junk16 : {0, 2, 8, 16}) {       // per workgroup: junk16 x 4 KiB; per XCD (128 workgroups): junk16 x 0.5 MiB
    printf("  junk %4.1f MiB per XCD:  consumer on the producer's XCD %6.2f   on another XCD %6.2f\n", junk16 * 0.5,
           run_pair(n, 4096, a, b, junk, junk16, 0), run_pair(n, 4096, a, b, junk, junk16, 3));
  }
  printf("  producer writes into a buffer it has not seen for 7 pairs (its stores miss the L2): consumer on the producer's XCD %6.2f   on another XCD %6.2f\n",
         run_pair_fresh(n, 4096, a, junk, 0), run_pair_fresh(n, 4096, a, junk, 3));
  // the library's own LayerNorm forward in the same ping-pong (input in the local L2 every launch), launched from C
  if (void* h = dlopen("aozora_sdxl_training_amd/libaozora_hip.so", RTLD_NOW)) {
    typedef int (*ln_fn)(int, int, float, const void*, long, const void*, const void*, void*, long, void*, void*);
    ln_fn ln = (ln_fn)dlsym(h, "az_layernorm_fwd");
    float* stats; CK(hipMalloc(&stats, 4096 * 2 * 4));
    unsigned short* gb; CK(hipMalloc(&gb, 2 * 1280 * 2)); CK(hipMemset(gb, 0, 2 * 1280 * 2));
    CK(hipMemset(a, 0, 4096 * 2560));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < n; ++i) {
        int rc = ln(4096, 1280, 1e-5f, (i & 1) ? b : a, 1280, gb, gb + 1280, (i & 1) ? a : b, 1280, stats, nullptr);
        if (rc) { printf("az_layernorm_fwd failed: %d\n", rc); return 1; }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("az_layernorm_fwd (4096 x 1280) in the same ping-pong, launched from C:  %6.2f us per launch\n", ms * 1e3 / n);
  }
  return 0;
}