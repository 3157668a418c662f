// Probe: buffer_load_dwordx4 ... lds (LDS-DMA): placement, OOB behaviour.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const unsigned* src, unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned lds[1024];   // 4 KB
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7FFFFFFF, 0x00020000);
  // lane l reads 16 B at src chunk (63 - l) (reversed), lanes 5 and 40 out of bounds
  unsigned off = (unsigned)(63 - l) * 16u;
  if (l == 5 || l == 40) off = 0x80000000u;
  if (mode == 0) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(lds + 256), 16, off, 0, 0, 0);   // into bytes [1024, 2048)
  } else {
    const unsigned* gp = src + (63 - l) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (lds_void*)(lds + 256), 16, 0, 0);
  }
  __syncthreads();
  for (int i = l; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(64 * 4); for (int i = 0; i < 256; ++i) h[i] = 0x1000u * (i / 4) + (i % 4);
  unsigned *ds, *dout; hipMalloc(&ds, 1024); hipMalloc(&dout, 4096);
  hipMemcpy(ds, h.data(), 1024, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(ds, dout, mode);
    std::vector<unsigned> o(1024); hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    int untouched_outside = 0, placed = 0;
    for (int i = 0; i < 1024; ++i) if ((i < 256 || i >= 512) && o[i] == 0xDEADBEEFu) ++untouched_outside;
    for (int l = 0; l < 64; ++l) { bool ok = true; for (int e = 0; e < 4; ++e) ok &= (o[256 + l * 4 + e] == 0x1000u * (63 - l) + e); placed += ok; }
    printf("mode %d: outside untouched %d/768, lanes placed at base+16*lane with own data: %d/64\n", mode, untouched_outside, placed);
    printf("  lane5 slot: %08x %08x %08x %08x   lane40 slot: %08x %08x\n", o[256 + 20], o[256 + 21], o[256 + 22], o[256 + 23], o[256 + 160], o[256 + 161]);
  }
  printf("status %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
