// Probe of gfx950 primitives the kernels rely on: MFMA fragment maps and ds_read_b64_tr_b16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline short f2bf(float f) { uint32_t u = __float_as_uint(f); return (short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

// A[i][k] = i*64+k style exact ints; we compute D = A*B with A = one-hot rows to read maps.
__global__ void k_mfma16(const short* A /*16x32 row-major*/, const short* B /*32x16 (k,n) row-major*/, float* D /*16x16*/) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_mfma32(const short* A /*32x16*/, const short* B /*16x32 (k,n)*/, float* D /*32x32*/) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// tr read: LDS image T[row][col] of shorts with value row*256+col, pitch P shorts. lane supplies address of
// (row = rbase + (i>>2), col = 4*(i&3)) within its 16-lane group; group g uses rbase = 4*g.
__global__ void k_tr(int* out /*64 lanes x 4*/) {
  __shared__ __attribute__((aligned(16))) short T[64 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 64 * 64; i += 64) T[i] = (short)(((i / 64) << 8) | (i % 64));
  __syncthreads();
  int i = l & 15, g = l >> 4;
  int row = 4 * g + (i >> 2), col = 16 + 4 * (i & 3);   // block: rows 4g..4g+3, cols 16..31
  uint32_t addr = (uint32_t)(uintptr_t)(&T[row * 64 + col]);
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = (int)(unsigned short)v[r];
}
int main() {
  // mfma16: A = exact small ints, B asymmetric
  std::vector<short> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16);
  auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (short)(u >> 16); };
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { float v = (float)((i * 3 + k) % 7 - 3); Af[i * 32 + k] = v; A[i * 32 + k] = bf(v); }
  for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) { float v = (float)((k * 5 + n * 2) % 9 - 4); Bf[k * 16 + n] = v; B[k * 16 + n] = bf(v); }
  short *dA, *dB; float* dD; int* dO;
  hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dD, 32 * 32 * 4); hipMalloc(&dO, 64 * 4 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k_mfma16<<<1, 64>>>(dA, dB, dD);
  std::vector<float> D(32 * 32); hipMemcpy(D.data(), dD, 16 * 16 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float r = 0; for (int k = 0; k < 32; ++k) r += Af[i * 32 + k] * Bf[k * 16 + n]; if (r != D[i * 16 + n]) ++bad; }
  printf("mfma16x16x32 map mismatches: %d\n", bad);
  // mfma32
  std::vector<short> A2(32 * 16), B2(16 * 32); std::vector<float> A2f(32 * 16), B2f(16 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) { float v = (float)((i * 3 + k) % 7 - 3); A2f[i * 16 + k] = v; A2[i * 16 + k] = bf(v); }
  for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) { float v = (float)((k * 5 + n * 2) % 9 - 4); B2f[k * 32 + n] = v; B2[k * 32 + n] = bf(v); }
  hipMemcpy(dA, A2.data(), A2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B2.data(), B2.size() * 2, hipMemcpyHostToDevice);
  k_mfma32<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) { float r = 0; for (int k = 0; k < 16; ++k) r += A2f[i * 16 + k] * B2f[k * 32 + n]; if (r != D[i * 32 + n]) ++bad; }
  printf("mfma32x32x16 map mismatches: %d\n", bad);
  k_tr<<<1, 64>>>(dO);
  std::vector<int> O(256); hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
  // expectation: lane (g,i) gets column 16+i of rows 4g..4g+3 -> value ((4g+r)<<8)|(16+i)
  bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { int g = l >> 4, i = l & 15; int want = ((4 * g + r) << 8) | (16 + i); if (O[l * 4 + r] != want) ++bad; }
  printf("ds_read_b64_tr_b16 expectation mismatches: %d\n", bad);
  for (int l = 0; l < 64; l += 5) printf("lane %2d: %04x %04x %04x %04x\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]);
  hipError_t e = hipDeviceSynchronize(); printf("status %s\n", hipGetErrorString(e));
  return 0;
}
