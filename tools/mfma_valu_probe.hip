// Does vector-ALU work hide in the shadow of MFMAs, and does it depend on WHERE the MFMA's accumulator lives (architectural VGPR
// vs accumulator VGPR) and on who issues the vector work (the same wave between its MFMAs, or the partner wave of the SIMD)?
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/mfma_valu_probe tools/mfma_valu_probe.hip ; run on the GPU box, no arguments
// Every variant runs NMF dependent v_mfma_f32_32x32x16_bf16 per loop iteration with FILL vector instructions behind each of them
// (KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pk_bf16_f32, 3: mix 2 fma + 2 exp + 1 cvt), all inline asm so that the stream is what
// is written here.  Reported: shader cycles per MFMA (s_memtime around the loop, wave 0 of block 0) at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND, int n> __device__ __forceinline__ void fill5(float& a, float& b, float& c, float& d, unsigned& e) {
  // up to 5 independent vector instructions on private registers
  if (KIND == 0) {
    if constexpr (n > 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
    if constexpr (n > 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b));
    if constexpr (n > 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c));
    if constexpr (n > 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d));
    if constexpr (n > 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
    if constexpr (n > 5) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b));
    if constexpr (n > 6) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c));
    if constexpr (n > 7) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d));
  } else if (KIND == 1) {
    if constexpr (n > 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    if constexpr (n > 1) asm volatile("v_exp_f32 %0, %0" : "+v"(b));
    if constexpr (n > 2) asm volatile("v_exp_f32 %0, %0" : "+v"(c));
    if constexpr (n > 3) asm volatile("v_exp_f32 %0, %0" : "+v"(d));
  } else if (KIND == 2) {
    if constexpr (n > 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e) : "v"(a), "v"(b));
    if constexpr (n > 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e) : "v"(c), "v"(d));
    if constexpr (n > 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e) : "v"(a), "v"(c));
    if constexpr (n > 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e) : "v"(b), "v"(d));
  } else {
    // the softmax mix per MFMA gap: n = 5 -> 2 fma, 2 exp, 1 cvt; n = 8 -> 3 fma 3 exp 2 cvt
    asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
    asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b));
    if constexpr (n > 5) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c));
    asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    asm volatile("v_exp_f32 %0, %0" : "+v"(b));
    if constexpr (n > 5) asm volatile("v_exp_f32 %0, %0" : "+v"(c));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e) : "v"(a), "v"(b));
    if constexpr (n > 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e) : "v"(c), "v"(d));
  }
}

// ACC: 0 = accumulator in architectural VGPRs, 1 = in AGPRs.  ROLE 0: every wave runs MFMA + fill; ROLE 1: waves 0-3 of an 8-wave
// block run the MFMAs only and waves 4-7 the fill only (the partner wave of the same SIMD); ROLE 2: waves 0-3 MFMAs only, waves
// 4-7 idle (exit); ROLE 3: waves 4-7 fill only, waves 0-3 exit.  out[2w], out[2w+1] = start / end stamp of wave 0 (w=0) and 4 (w=1)
template <int ACC, int KIND, int ROLE, int NF>
__global__ __launch_bounds__(512, 1) void probe(int iters, long long* out, float* sink) {
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + threadIdx.x % 7); b[j] = (short)(0x3C00 + j); }
  float f0 = 0.5f + threadIdx.x * 1e-3f, f1 = 0.25f, f2 = 0.125f, f3 = 0.0625f; unsigned e = 0;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (ROLE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (ACC == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        fill5<KIND, NF>(f0, f1, f2, f3, e);
      }
    }
  } else if (wave < 4) {
    if (ROLE == 1 || ROLE == 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (ACC == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        }
      }
  } else {
    if (ROLE == 1 || ROLE == 3)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) fill5<KIND, NF>(f0, f1, f2, f3, e);
      }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) { out[2 * (threadIdx.x >> 8)] = t0; out[2 * (threadIdx.x >> 8) + 1] = t1; }
  float s = f0 + f1 + f2 + f3 + (float)e;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int ACC, int KIND, int ROLE, int NF> static void run(const char* name, int threads, long long* d_out, float* sink) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<ACC, KIND, ROLE, NF>), dim3(256), dim3(threads), 0, 0, iters, d_out, sink);
  hipLaunchKernelGGL((probe<ACC, KIND, ROLE, NF>), dim3(256), dim3(threads), 0, 0, iters, d_out, sink);
  CK(hipDeviceSynchronize());
  long long c[4]; CK(hipMemcpy(c, d_out, 32, hipMemcpyDeviceToHost));
  const double n = iters * 16.0;
  if (threads == 256) printf("%-40s 1 wave/SIMD  fill %d : %6.1f cycles per (MFMA + fill)\n", name, NF, (c[1] - c[0]) / n);
  else printf("%-40s 2 waves/SIMD fill %d : wave 0 %6.1f  wave 4 %6.1f  both done after %6.1f cycles per step\n", name, NF, (c[1] - c[0]) / n, (c[3] - c[2]) / n,
              (double)(std::max(c[1], c[3]) - std::min(c[0], c[2])) / n);
}

template <int KIND, int ROLE, int NF> static void both(const char* what, int th, long long* d_out, float* sink) {
  char n0[96], n1[96]; snprintf(n0, 96, "acc VGPR, %s", what); snprintf(n1, 96, "acc AGPR, %s", what);
  run<0, KIND, ROLE, NF>(n0, th, d_out, sink); run<1, KIND, ROLE, NF>(n1, th, d_out, sink);
}
int main() {
  long long* d_out; float* sink; CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&sink, 4096));
  for (int th : {256, 512}) {
    both<0, 0, 0>("v_fma fill, same wave", th, d_out, sink); both<0, 0, 2>("v_fma fill, same wave", th, d_out, sink);
    both<0, 0, 4>("v_fma fill, same wave", th, d_out, sink); both<0, 0, 5>("v_fma fill, same wave", th, d_out, sink);
    both<0, 0, 6>("v_fma fill, same wave", th, d_out, sink); both<0, 0, 8>("v_fma fill, same wave", th, d_out, sink);
    both<1, 0, 1>("v_exp fill, same wave", th, d_out, sink); both<1, 0, 2>("v_exp fill, same wave", th, d_out, sink);
    both<1, 0, 3>("v_exp fill, same wave", th, d_out, sink); both<1, 0, 4>("v_exp fill, same wave", th, d_out, sink);
    both<2, 0, 2>("v_cvt_pk fill, same wave", th, d_out, sink); both<2, 0, 4>("v_cvt_pk fill, same wave", th, d_out, sink);
    both<3, 0, 5>("softmax mix, same wave", th, d_out, sink); both<3, 0, 8>("softmax mix, same wave", th, d_out, sink);
  }
  both<0, 2, 0>("MFMA waves alone (partner exits)", 512, d_out, sink);
  both<0, 3, 4>("v_fma fill alone (MFMA waves exit)", 512, d_out, sink); both<0, 3, 8>("v_fma fill alone (MFMA waves exit)", 512, d_out, sink);
  both<3, 3, 5>("softmax mix alone", 512, d_out, sink); both<3, 3, 8>("softmax mix alone", 512, d_out, sink);
  both<0, 1, 2>("v_fma fill, PARTNER wave", 512, d_out, sink); both<0, 1, 4>("v_fma fill, PARTNER wave", 512, d_out, sink);
  both<0, 1, 6>("v_fma fill, PARTNER wave", 512, d_out, sink); both<0, 1, 8>("v_fma fill, PARTNER wave", 512, d_out, sink);
  both<1, 1, 2>("v_exp fill, PARTNER wave", 512, d_out, sink); both<1, 1, 4>("v_exp fill, PARTNER wave", 512, d_out, sink);
  both<3, 1, 5>("softmax mix, PARTNER wave", 512, d_out, sink); both<3, 1, 8>("softmax mix, PARTNER wave", 512, d_out, sink);
  return 0;
}
