// Shared device helpers for the aozora HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <tuple>
#include <utility>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define AZ_OK 0
#define AZ_ERR_ARG(k) (-1000 - (k))

#define AZ_CHECK_LAUNCH()                          \
  do {                                             \
    hipError_t _e = hipGetLastError();             \
    if (_e != hipSuccess) return -(int)_e;         \
  } while (0)

// Every kernel of the library is launched through az_launch.  While this thread has a stop event set (az_set_launch_stop_event:
// the launch tape's peephole for "kernel, then hipEventRecord on the same stream"), the launch carries the event as its OWN
// completion signal (hipExtLaunchKernel) -- no separate record packet behind the kernel on the recording stream
// (tools/event_cost.cpp: +0.2 us against +1.3 us for a record without the system-scope fence, +2.8 us with torch's flags).  An entry
// point that launches several kernels re-records the event with each: what a later hipStreamWaitEvent sees is the last one.
hipEvent_t az_stop_event_of_this_thread();      // az_runtime.hip (a function, not a variable: the device pass parses this header too)
template <typename... P, size_t... I>
inline void az_launch_ev_(void (*k)(P...), dim3 g, dim3 b, size_t sh, hipStream_t st, hipEvent_t ev, std::tuple<P...>& t, std::index_sequence<I...>) {
  void* args[] = {(void*)&std::get<I>(t)..., nullptr};
  (void)hipExtLaunchKernel((const void*)k, g, b, args, sh, st, nullptr, ev, 0);
}
template <typename... P, typename... A>
inline void az_launch(void (*k)(P...), dim3 g, dim3 b, size_t sh, hipStream_t st, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
  const hipEvent_t ev = az_stop_event_of_this_thread();
  if (!ev) { hipLaunchKernelGGL(k, g, b, sh, st, std::forward<A>(a)...); return; }
  std::tuple<P...> t{static_cast<P>(std::forward<A>(a))...};      // the kernel's formal parameter types, as <<< >>> would convert them
  az_launch_ev_(k, g, b, sh, st, ev, t, std::index_sequence_for<P...>{});
}

#define AZ_HIP(x)                                  \
  do {                                             \
    hipError_t _e = (x);                           \
    if (_e != hipSuccess) return -(int)_e;         \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserving (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// sigmoid through the hardware reciprocal (1 ulp): an IEEE fp32 division is ~10 VALU instructions per element, and these run inside
// the GroupNorm passes that share their CUs with the GEMMs of the other stream
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}

// erf GELU and its derivative: the GEGLU gate (diffusers GEGLU.gelu -> F.gelu, no tanh approximation).
// Phi(g) = 0.5 erfc(-g / sqrt 2) with erfc(x) = t (a1 + t (a2 + ... a5 t)) e^{-x^2}, t = 1 / (1 + p x), x >= 0 (Abramowitz & Stegun
// 7.1.26): |error| <= 1.5e-7 on erfc, <= 4.3e-7 absolute on gelu and its derivative over all g (checked against float64) -- four
// orders of magnitude below a bf16 ulp of the values stored -- at ONE exponential (the same e^{-g^2/2} serves the density term of
// the derivative), one reciprocal and a dozen FMAs, where libdevice's erff + expf cost ~100 VALU instructions per element and made
// the GEGLU backward kernel VALU-co-limited (45.6 -> 39.4 us at [4096][2 x 5120]).  The negative tail is formed from erfc directly
// (no 1 - erf cancellation).
__device__ __forceinline__ void gelu_pair(float g, float& gelu, float& dgelu) {
  const float x = fabsf(g) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  const float e = __expf(-x * x);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float hc = 0.5f * poly * e;                  // 0.5 erfc(|g| / sqrt 2)
  const float phi = g >= 0.f ? 1.0f - hc : hc;
  gelu = g * phi;
  dgelu = fmaf(g * 0.3989422804014327f, e, phi);
}
__device__ __forceinline__ float gelu_erf(float g) { float a, b; gelu_pair(g, a, b); return a; }
__device__ __forceinline__ float dgelu_erf(float g) { float a, b; gelu_pair(g, a, b); return b; }

// i = q * d + r for a flat element index: 32-bit division whenever the index fits (a 64-bit one is ~150 VALU instructions,
// several times the arithmetic of the 8 elements it addresses in the element-wise kernels)
__device__ __forceinline__ void divmod(long i, int d, long& q, int& r) {
  if ((unsigned long)i <= 0xFFFFFFFFul) {
    const unsigned qi = (unsigned)i / (unsigned)d;
    q = (long)qi; r = (int)((unsigned)i - qi * (unsigned)d);
  } else {
    q = i / d; r = (int)(i - q * d);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024; `sh` must hold 16 floats. Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// hardware transposed LDS read (guide T10): per 16-lane group, lane i supplies the address of
// block row (i>>2), columns 4*(i&3)..+3; it receives column i of the 4 rows.
__device__ __forceinline__ bf16x4 lds_read_tr16_b64(uint32_t lds_byte_addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_byte_addr) : "memory");
  return v;
}

// ---- runtime options (az_set_option / az_get_option, az_runtime.hip) ----------------------------------------------------------
// One process-wide table of integer knobs (tile policy, split-K heuristics, stream exclusivity ...), each an atomic: read at
// launch time by the host-side launchers, settable between launches from any thread.  Initial values: the defaults below,
// overridden once by the environment variable AZ_<NAME> if present.  They steer SPEED only (NORM_STAT_BF16 excepted: it selects whose saved statistics the norm backward follows) -- every setting computes the same
// mathematical result (split-K changes the fp32 summation order).
enum AzOption {
  AZ_OPT_TILE_POLICY = 0,     // 4: 8-wave 128x128 / 128x160 tiles, 256x256 where the grid fills (see az_gemm.hip choose_tile)
  AZ_OPT_BIG_FILL,            // tenths of whole waves of 256 CUs from which the 256x256 tile is taken (5)
  AZ_OPT_SPLIT_SLOTS,         // workgroup slots a split-K grid is sized for (384: beside the other stream a 100-tile weight gradient runs best in 3 slabs, not 5)
  AZ_OPT_NOSPLIT_TILES,       // grids of at least this many tiles are never split (256 = one per CU; was 384 before the weight-gradient loop
                              //    stopped stalling on its own DMA: 3840x1280x4096 66 us unsplit vs 76 us in three slabs, micro-step -0.8 ms)
  AZ_OPT_LDS_EXCLUSIVE,       // 1 while the data chain has the CUs to itself (forward pass): 3-stage tiles allowed
  AZ_OPT_NT_SPLIT_BIG,        // k-heavy few-tile linear products (M*N = 80 tiles of 256x256) beside the weight-gradient stream: 256-row 8-wave tiles with up to
                              //    this many k-splits (3; 0 = off).  Same-process A/B, 6 rounds: 117.75 -> 117.49 ms per micro-step with NT_SPLIT_MINK 5120
                              //    (only the ff.net.0 data gradient, K = 10240: 171 -> 111 us alone), 118.03 with 3840 (the q|k|v data gradient too)
  AZ_OPT_NT_SPLIT_MINK,       // ... from this K on (5120)
  AZ_OPT_ATTN_SPLIT_TARGET,   // workgroups the cross-attention dK/dV query split aims at (384)
  AZ_OPT_LN_RPB,              // LayerNorm backward (fused form) rows per block (32; was 8 until round 5: in the step fewer, longer-lived blocks win -- 113.1 vs 114.0 ms)
  AZ_OPT_INKERNEL_FINISH,     // 1: split-K slabs and fused column sums are finished by the tile's last-arriving workgroup (no reduce /
                              //    finish launch).  Default 0: correct and bitwise equal to the separate launches, but +3 ms per micro-step
                              //    in the two-stream step (same-process A/B, tools/ab_opts.py; +10 ms in the release-fence form)
  AZ_OPT_NT_SPLIT2_MINK,      // k-contiguous products on <= 256 tiles of 128x160 (one workgroup per CU) outside the exclusive forward pass: split k in two
                              //    from this K on (0 = off, the default: the chain alone gains 2.5 ms per micro-step at 3840, the two-stream step loses
                              //    1.5 ms -- the extra workgroups take the weight-gradient stream's place on the CUs; same-process A/B, tools/chain_time.py)
  AZ_OPT_GEMM_ABLATE,         // diagnostic, timing only (WRONG RESULTS): 1 = GEMM kernels skip fragment reads + MFMAs, 2 = skip operand DMA after the first k-tile, 4 = no split-K reduce / column-sum finish launches
  AZ_OPT_GEMM8,               // 1: the 256-row tile of plain k-contiguous products runs on the 8-wave ping-pong kernel (az_gemm8.inc); 0: the 16-wave one-barrier tile
  AZ_OPT_NT_SPLIT_FWD,        // the same split of few-tile k-heavy products onto 256-row tiles while LDS_EXCLUSIVE is set (forward pass): largest split count (0 = off)
  AZ_OPT_ATTN_PIPE,           // attention (az_attn.hip): bit 0 = self-attention forward in its software-pipelined LDS-DMA form, bit 1 = dQ and
                              //    dK/dV workgroups of a self-attention backward in ONE launch when both grids are short, bit 2 = the backward of a short key
                              //    axis (cross-attention, Tk <= 128) as ONE kernel when the caller asks for all of dQ, dK, dV (7), bit 3 = the dQ and dK/dV
                              //    bodies of self-attention shapes in their LDS-DMA form (unpadded swizzled tile images); 0 = the plain kernels
  AZ_OPT_XCD_SPLIT,           // 1: split-K weight gradients (linear and convolution) deal their k-SPLITS to the XCDs: the tiles of one k-range run
                              //    behind one L2 (az_gemm.hip gemm_kernel); 0: the tiles of every split are dealt to the XCDs
  AZ_OPT_ATTN_XCD,            // attention workgroup order (az_attn.hip attn_block): bit 0 forward, bit 1 the dQ and dK/dV kernels, bit 3 the merged launch (LDS-DMA form), bit 2 the
                              //    short-key one-kernel backward: all blocks (and roles) of a (batch, head) behind one XCD's L2; 0 = plain x-fastest order
  AZ_OPT_NORM_STAT_BF16,      // 1: the GroupNorm / LayerNorm BACKWARD reads mean and rstd rounded to bf16, as the reference's dataflow saves them
                              //    (az_norm.hip stat_round); 0: fp32 statistics.  The one option that changes RESULTS rather than speed.
  AZ_OPT_GN_RPT,              // GroupNorm forward: rows a thread covers per chunk at least (4); sets the chunk (= block) count of the two row passes
  AZ_OPT_GN_RPT_BWD,          // ... the same for the backward's passes.  More rows = fewer, longer-lived blocks: beside the weight-gradient stream a
                              //    light kernel pays for every block slot it has to wait for (profiles/r05_dilation_in_step.txt)
  AZ_OPT_GN_RPT_APPLY,        // ... and for the backward's element-wise pass (dx), whose chunks are independent of the partial sums'
  AZ_OPT_COUNT
};
int az_opt(int id);           // host side
