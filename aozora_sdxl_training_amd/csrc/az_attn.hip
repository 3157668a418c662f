// Flash-style scaled-dot-product attention for head_dim 64, forward and backward (gfx950).
// Stands in for F.scaled_dot_product_attention as called by diffusers' AttnProcessor2_0 inside
// BasicTransformerBlock.attn1/attn2 (reference selects it at train.py:204-228; executed at
// train.py:2760 fwd / 2765 bwd).  No mask, dropout 0, scale 1/sqrt(64); SURVEY.md 2.3 K12/K13.
//
// MFMA: v_mfma_f32_32x32x16_bf16.  Conventions (guide section 3): A-operand lane l holds
// A[row l&31][k 8(l>>5)+j], B-operand lane l holds B[k 8(l>>5)+j][col l&31]; the accumulator holds
// D[row (r&3)+8(r>>2)+4(l>>5)][col l&31].  All score tiles are computed TRANSPOSED (S^T = K.Q^T,
// key on the register axis, query on the lane) so that softmax statistics are per-lane scalars and
// the bf16-converted accumulator is directly the B operand of the next product ("accumulator as
// operand": k-slot j of lane half h = row 16s + 8(j>>2) + 4h + (j&3)); the other operand of that
// product is fetched from LDS with ds_read_b64_tr_b16 using the same row permutation.
//
// K / V / Q / dO tiles live in LDS as [row][64] bf16 with a 144-byte pitch: ds_read_b128 of
// 16 rows x 16 B is bank-conflict-free and each tr-read address is 8-byte aligned.
#include "az_common.h"
#include "aozora_hip.h"
#include <math.h>

namespace {

constexpr int D = 64;
constexpr int PITCH = D * 2 + 16;       // 144 B
constexpr int TILE = 64;                 // rows per staged tile
constexpr int TILE_BYTES = TILE * PITCH; // 9216
constexpr float LOG2E = 1.4426950408889634f;

typedef __attribute__((address_space(3))) bf16x4 lds_v4;

struct AttnPtr {
  const bf16_t* p; long ld, sb;   // row stride (elements), batch stride (elements)
};
struct AttnOut {
  bf16_t* p; long ld, sb;
};

// raw v_exp_f32: exp2f() expands to a denormal-safe sequence (compare, select, add, ldexp: +5 VALU instructions per
// score) that a softmax never needs -- results below 2^-126 may flush to zero
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ bf16x8 cvt8(const f32x16& a, int s) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(a[8 * s + j]);
  return v;
}

// A-operand style fragment (row = rowbase + (l&31), k = 16*s + 8*(l>>5) + j) from a [row][64] image
__device__ __forceinline__ bf16x8 frag_rowmajor(const char* img, int rowbase, int s, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + (rowbase + (lane & 31)) * PITCH + (16 * s + 8 * (lane >> 5)) * 2);
}
// transposed fragment: lane row = column index c = colbase + (l&31) of the image, k-slots = image rows
// rowbase + 16*ks + 8*(j>>2) + 4*(l>>5) + (j&3)
__device__ __forceinline__ bf16x8 frag_transposed(const char* img, int rowbase, int ks, int colbase, int lane) {
  const int g = lane >> 4, i = lane & 15, h = lane >> 5;
  const char* base = img + (rowbase + 16 * ks + 4 * h + (i >> 2)) * PITCH + (colbase + 16 * (g & 1) + 4 * (i & 3)) * 2;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base + 8 * PITCH));
  bf16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return v;
}

// global -> registers for one [64][64] tile: 2 chunks of 16 B per thread (256 threads)
template <bool FULL = false>
__device__ __forceinline__ void tile_load(const bf16_t* base, long ld, int row0, int nrows, int t, uint4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int c = t + 256 * i;
    int row = c >> 3, dc = c & 7;
    r[i] = (FULL || row0 + row < nrows) ? *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * ld + dc * 8) : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void tile_store(char* img, int t, const uint4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int c = t + 256 * i;
    *reinterpret_cast<uint4*>(img + (c >> 3) * PITCH + (c & 7) * 16) = r[i];
  }
}

// operand-B style per-wave resident fragments: lane (row = l&31 of the wave's 32 rows, d = 16s+8h+j)
template <bool FULL = false>
__device__ __forceinline__ void load_row_frags(const bf16_t* base, long ld, int row, int nrows, int lane, bf16x8 (&f)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (FULL || row < nrows) f[s] = *reinterpret_cast<const bf16x8*>(base + (long)row * ld + 16 * s + 8 * (lane >> 5));
    else f[s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Workgroup order (round 5).  Every attention launch is ONE-dimensional; a workgroup finds (x, y, z) -- x = query / key block,
// y = (batch, head), z = role or query split -- from its id.  The dispatcher deals consecutive workgroup ids to the 8 XCDs in
// turn (MI355X_MICROARCH.md "Workgroup dispatch"), so with x fastest the 8 (T = 1024) or 32 (T = 4096) query blocks of one head
// ran behind eight different L2s and each of them fetched that head's K / V from the fabric: 4.5-5.2 x the algorithmic bytes
// (profiles/r04_d_pmc_fetch.json).  With `xcd` set the ids that share id % 8 -- one XCD's workgroups -- are mapped to a contiguous
// range of (y, z, x): all blocks of a (batch, head) run behind ONE L2, in dispatch order (guide 5.5 T1, the bijective form for
// n % 8 != 0).  Placement changes speed only.  Measured (tools/attn_lab, same process, profiles/r05_attn_order_lab.txt): forward
// (4,20,1024,1024) 43.2 -> 40.4 us, but (4,10,4096,4096) 210 -> 218: the 32 blocks of a head then ask ONE L2 for the same lines at
// the same moment, channel after channel -- hence the rotated key order of attn_fwd_dma_kernel (38.8 / 202.3 us with both).  Backward
// as two kernels (T = 4096) 655 -> 633 us; the merged one-launch backward (T = 1024) loses IN ISOLATION (register-staged bodies 103 ->
// 110 us, 107 with rotated tiles, 112 with the roles apart; LDS-DMA bodies 96.7 -> 103.7) but wins IN THE STEP, where the 5.2 x
// fabric traffic of the plain order is taken from the other stream: micro-step 114.6 -> 114.0 ms (profiles/r05_attn_merged_xcd_ab.txt);
// remapped in its LDS-DMA form (bit 3), plain in the register-staged one.  The short-key kernel is indifferent.
struct AttnGrid { int nx, ny, nz, xcd; };
struct AttnBlk { int x, y, z; };
__device__ __forceinline__ AttnBlk attn_block_of(const int nx, const int ny, const int nz, const int xcd) {
  int id = blockIdx.x;
  if (xcd) {
    const int n = nx * ny * nz, q = n >> 3, r = n & 7, x = id & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  const int rest = id / nx;
  const int bx = id - rest * nx;
  int by, bz;
  if (xcd) { by = rest / nz; bz = rest - by * nz; }     // query splits of one head next to each other
  else { bz = rest / ny; by = rest - bz * ny; }         // the plain (x, y, z) order
  // wave-uniform by construction; said so explicitly because the divisions run on the vector ALU and the results feed scalar
  // operands (buffer descriptors, LDS-DMA offsets)
  return AttnBlk{__builtin_amdgcn_readfirstlane(bx), __builtin_amdgcn_readfirstlane(by), __builtin_amdgcn_readfirstlane(bz)};
}
// (values, not references: with int& outputs every kernel of this file carried 12 bytes of scratch per lane -- a private segment to
//  set up at each of ~400 launches per micro-step -- which cost the step about as much as the new workgroup order gained)
#define attn_block(G, bx, by, bz) do { const AttnBlk blk_ = attn_block_of((G).nx, (G).ny, (G).nz, (G).xcd); bx = blk_.x; by = blk_.y; bz = blk_.z; } while (0)

// =============================== forward ======================================================
// FULL: Tq % 128 == 0 and Tk % 64 == 0 (every self-attention of the UNet): no row / key range tests, no half-tile skips.  The
// general form's key mask was if-converted into 32 compares + 32 selects per key tile and its skippable second half kept the
// score accumulators zero-initialised by 32 moves -- 110 of the 283 vector instructions of a key tile, in a loop that is bound by
// the vector ALU (20 MFMAs = 640 cycles against ~1500 cycles of VALU issue per wave and tile).
template <bool FULL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                          AttnOut O, float* __restrict__ lse2) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // K0 V0 K1 V1
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int bx, bh, bz_; attn_block(G, bx, bh, bz_);
  const int b = bh / heads, h = bh - b * heads;
  const int q0 = bx * 128 + wave * 32;
  const bf16_t* Qb = Q.p + b * Q.sb + h * D;
  const bf16_t* Kb = K.p + b * K.sb + h * D;
  const bf16_t* Vb = V.p + b * V.sb + h * D;
  const float c = scale * LOG2E;

  bf16x8 qf[4];
  load_row_frags<FULL>(Qb, Q.ld, q0 + (lane & 31), Tq, lane, qf);

  f32x16 o[2] = {zero16(), zero16()};
  // softmax denominator on the matrix pipe (the VALU is the saturated pipe here): lsum = ones[32][keys] . P^T[keys][q],
  // every row of the accumulator holds the same per-query sum of the bf16 probabilities that also feed P.V
  f32x16 lsum = zero16();
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  float m = -INFINITY;

  const int ntiles = (Tk + TILE - 1) / TILE;
  uint4 rk[2], rv[2];
  tile_load<FULL>(Kb, K.ld, 0, Tk, t, rk);
  tile_load<FULL>(Vb, V.ld, 0, Tk, t, rv);
  tile_store(smem, t, rk);
  tile_store(smem + TILE_BYTES, t, rv);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    const char* kimg = smem + cur * 2 * TILE_BYTES;
    const char* vimg = kimg + TILE_BYTES;
    const bool more = kt + 1 < ntiles;
    if (more) {
      tile_load<FULL>(Kb, K.ld, (kt + 1) * TILE, Tk, t, rk);
      tile_load<FULL>(Vb, V.ld, (kt + 1) * TILE, Tk, t, rv);
    }
    // S^T[key][q] for the two 32-key halves; a half that lies entirely beyond Tk is skipped everywhere below (wave-uniform:
    // cross-attention has 77 keys = 2.4 halves)
    const int kbase = kt * TILE;
    const bool h1 = FULL || kbase + 32 < Tk;
    f32x16 st[2] = {zero16(), zero16()};
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      if (!FULL && kh == 1 && !h1) break;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        st[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(kimg, 32 * kh, s, lane), qf[s], st[kh], 0, 0, 0);
    }
    // mask keys beyond Tk (only the last tile can be partial)
    if (!FULL && kbase + TILE > Tk) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + 32 * kh + acc_row(r, lane) >= Tk) st[kh][r] = -INFINITY;
    }
    float mx = st[0][0];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      if (!FULL && kh == 1 && !h1) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kh][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float mc = m_new * c;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      if (!FULL && kh == 1 && !h1) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kh][r] = fast_exp2(fmaf(st[kh][r], c, -mc));   // one fma + one exp per score
    }
    if (__any(m_new != m)) {                 // wave-uniform: the running max moved for some query -> rescale O and l
      const float alpha = fast_exp2((m - m_new) * c);
      lsum[0] *= alpha;                      // only element 0 is read back; MFMA accumulates element-wise
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      m = m_new;
    }
    // O^T[d][q] += V^T[d][key] . P^T[key][q]
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      if (!FULL && kh == 1 && !h1) break;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf = cvt8(st[kh], s);
        lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lsum, 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(vimg, 32 * kh, s, 32 * dt, lane), pf, o[dt], 0, 0, 0);
      }
    }
    if (more) {
      tile_store(smem + (cur ^ 1) * 2 * TILE_BYTES, t, rk);
      tile_store(smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, t, rv);
    }
    __syncthreads();
  }

  const int q = q0 + (lane & 31);
  if (FULL || q < Tq) {
    const float l = lsum[0];
    const float inv = 1.0f / l;
    bf16_t* op = O.p + b * O.sb + (long)q * O.ld + h * D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 u;
        u.x = pack2bf(o[dt][4 * rr] * inv, o[dt][4 * rr + 1] * inv);
        u.y = pack2bf(o[dt][4 * rr + 2] * inv, o[dt][4 * rr + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * dt + 8 * rr + 4 * (lane >> 5)) = u;
      }
    if (lane < 32) lse2[((long)bh) * Tq + q] = m * c + log2f(l);
  }
}

// =============================== forward, software-pipelined (self-attention shapes) ============
// Same arithmetic as attn_fwd_kernel<true>, other schedule.  In the plain kernel a wave runs K.Q^T, then the softmax, then P.V
// one after the other, and PMC shows the SIMD's time as the SUM of its matrix time (640 cycles per key tile) and its vector time
// (~750): co-resident waves do the same thing at the same time.  Here one wave carries TWO key tiles at different stages: while
// the matrix pipe computes S(t+1) = K(t+1).Q^T the vector ALU exponentiates the first key half of S(t); P.V of that half runs
// beside the exponentials of the second half; P.V of the second half beside the row maximum of S(t+1) (guide T15
// "compute[cur] || finish[prev]"; tools/mfma_valu_probe.hip: up to 5 plain / 2 transcendental vector instructions hide behind one
// v_mfma_f32_32x32x16_bf16, from the same wave or from the SIMD's other wave, accumulator in VGPRs or AGPRs alike).  The K
// fragments of tile t+1 are read right behind the barrier that publishes the tile and the V fragments of tile t at the top of the
// iteration, so the MFMA blocks are register-only.
// Deferred maximum (guide T13): the running maximum m is raised -- and O, l rescaled -- only when some query's tile maximum
// exceeds it by more than 2^THR2 in the exponent (a rarely taken wave-uniform branch OUTSIDE the scheduled block); until then
// probabilities are formed against the stale m and are bounded by 2^THR2, which neither the bf16 P nor the fp32 sums notice
// (floating point: the relative rounding error does not depend on the scale).  tests/test_kernels_gpu.py forces the branch with
// a spiked key row (guide 5.4 rule 26).
// K tiles are staged two tiles ahead, V tiles one: ring of two LDS buffers each, ONE barrier per tile.
// Measured (tools/attn_lab, same process, random data; profiles/r04_attn_fwd_*.txt): (4,10,4096,4096) 229.8 -> 210.8 us,
// (16,20,1024,1024) 141 -> 134 us, (4,20,1024,1024) 42.7 -> 41.9 us (640 workgroups of 16 tiles on 512 slots: quantisation and
// the prologue, not the loop).  Timing-only ablations of the loop at T = 4096: MFMAs alone 141 us (20 MFMAs x 32 cycles per
// tile: the chip holds ~1.45 GHz under them), softmax arithmetic alone 98 us, both from registers 181 us, + tile staging, LDS
// reads and barrier 211 us.
constexpr float ATTN_THR2 = 4.0f;

// cross-half maximum without an LDS round trip: v_permlane32_swap exchanges the upper half of one operand with the lower half of
// the other, so max(r[0], r[1]) is max(own, partner) in every lane
__device__ __forceinline__ float max_xhalf(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// max of the 16 accumulator registers as two independent serial chains (each folds into v_max3_f32; a tree of fmaxf calls makes
// the compiler canonicalise every MFMA output with a v_max x, x first)
__device__ __forceinline__ float max16(const f32x16& a) {
  float x = a[0], y = a[8];
#pragma unroll
  for (int r = 1; r < 8; ++r) { x = fmaxf(x, a[r]); y = fmaxf(y, a[8 + r]); }
  return fmaxf(x, y);
}

// =============================== forward, pipelined, tiles staged by LDS-DMA ===================
// K / V tiles move global -> LDS by buffer_load ... lds (no staging registers, no
// ds_write, no per-tile address arithmetic: the tile advance rides in the scalar offset).  An LDS-DMA piece is lane-linear (lane l's
// 16 bytes land at base + 16 l), so the image is [64 rows][128 B] UNPADDED and the bank spread comes from a swizzle applied to the
// per-lane SOURCE address and to the fragment reads (guide 5.4 rule 21): 16-byte chunk c of row r lives at position c ^ swz(r),
// swz(r) = ((r>>1)&1)<<2 | (r>>2)&3.  With two 128-byte rows per 256-byte bank row this makes both read kinds conflict-free: a
// ds_read_b128 lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31} of one chunk column) hits 16 distinct slots, and the four rows of a
// ds_read_b64_tr_b16 half (64 contiguous bytes each) land in the four different 16-bank quarters.  The swizzle depends on bits 1-3
// of the row only, so the fragment addresses of the other key half (+32 rows), k-step (+16 rows) and ring buffer are IMMEDIATE
// offsets of eight per-lane base addresses; the tile loop is unrolled by two for that.
constexpr int DT_BYTES = 64 * 128;
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ unsigned lds_addr_of(const char* p) { return (unsigned)(unsigned long)(lds_void_t*)const_cast<char*>(p); }
__device__ __forceinline__ u32x4 rsrc_words(const void* p) {
  const unsigned long a = (unsigned long)p;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
}
// one 1-KiB piece; issued from inline asm so that hipcc's wait bookkeeping does not drain it in front of unrelated LDS reads
// (az_gemm.hip dma16s); the kernel waits for its own DMA (vmcnt(0)) right before the barrier that publishes a tile
__device__ __forceinline__ void attn_dma16(u32x4 r, unsigned voff, unsigned soff, unsigned dst_wave_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(dst_wave_uniform), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// tile stream of one operand for a 4-wave workgroup: wave w moves pieces 2w, 2w+1 = rows 16w .. 16w+15 of every tile
struct DmaStream {
  u32x4 rs; unsigned voff[2]; unsigned tile_bytes;
  __device__ __forceinline__ void init(const bf16_t* base, long ld, int lane, int wave) {
    rs = rsrc_words(base);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 16 * wave + 8 * j + (lane >> 3);
      voff[j] = (unsigned)(row * ld * 2 + (((lane & 7) ^ swz(row)) << 4));
    }
    tile_bytes = (unsigned)(64 * ld * 2);
  }
  __device__ __forceinline__ void issue(int tile, unsigned dst, int wave_u) const {
    const unsigned so = (unsigned)tile * tile_bytes;
    attn_dma16(rs, voff[0], so, dst + (unsigned)(2 * wave_u) * 1024u);
    attn_dma16(rs, voff[1], so, dst + (unsigned)(2 * wave_u + 1) * 1024u);
  }
};
template <int N> struct IC { static constexpr int value = N; };

__global__ __launch_bounds__(256, 2) void attn_fwd_dma_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                              AttnOut O, float* __restrict__ lse2) {
  __shared__ __attribute__((aligned(1024))) char smem[4 * DT_BYTES];   // K ring [2], V ring [2]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int bx, bh, bz_; attn_block(G, bx, bh, bz_);
  const int b = bh / heads, h = bh - b * heads;
  const int q0 = bx * 128 + wave * 32;
  const float c = scale * LOG2E;
  const unsigned kring = lds_addr_of(smem), vring = kring + 2 * DT_BYTES;
  DmaStream ks, vs;
  ks.init(K.p + b * K.sb + h * D, K.ld, lane, wave);
  vs.init(V.p + b * V.sb + h * D, V.ld, lane, wave);
  const int ntiles = Tk / TILE;                        // even, >= 2 (host-checked)
  // key tiles are visited in a rotated order that depends on the query block: the blocks of one head run side by side behind one
  // L2 (attn_block) and would otherwise ask it for the same lines at the same moment, channel after channel.  Unconditional: the
  // order of the online softmax is part of the result (to rounding), the workgroup placement option is not
  const int rot = (bx * ntiles) / G.nx;
  auto tw = [&](int i) { const int x = i + rot; return x >= ntiles ? x - ntiles : x; };
  ks.issue(tw(0), kring, wave); vs.issue(tw(0), vring, wave); ks.issue(tw(1), kring + DT_BYTES, wave);

  bf16x8 qf[4];
  load_row_frags<true>(Q.p + b * Q.sb + h * D, Q.ld, q0 + (lane & 31), Tq, lane, qf);
  // per-lane fragment addresses inside a tile image
  const int hh = lane >> 5, r32 = lane & 31, i16 = lane & 15, g1 = (lane >> 4) & 1;
  const char* ka[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) ka[s] = smem + r32 * 128 + (((2 * s + hh) ^ swz(r32)) << 4);
  const int rl = 4 * hh + (i16 >> 2);
  const char *va_lo[2], *va_hi[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int chunk = 4 * dt + 2 * g1 + ((i16 & 3) >> 1);
    va_lo[dt] = smem + 2 * DT_BYTES + rl * 128 + ((chunk ^ swz(rl)) << 4) + 8 * (i16 & 1);
    va_hi[dt] = smem + 2 * DT_BYTES + (rl + 8) * 128 + ((chunk ^ swz(rl + 8)) << 4) + 8 * (i16 & 1);
  }
  auto kfrag = [&](int buf, int kh, int s) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(ka[s] + buf * DT_BYTES + kh * 32 * 128); };
  auto vfrag = [&](int buf, int kh, int s, int dt) -> bf16x8 {
    const int off = buf * DT_BYTES + (32 * kh + 16 * s) * 128;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(va_lo[dt] + off));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(va_hi[dt] + off));
    bf16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 o[2] = {zero16(), zero16()};
  f32x16 lsum = zero16();
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  f32x16 st[2] = {zero16(), zero16()};
#pragma unroll
  for (int kh = 0; kh < 2; ++kh)
#pragma unroll
    for (int s = 0; s < 4; ++s) st[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(0, kh, s), qf[s], st[kh], 0, 0, 0);
  float m = max_xhalf(fmaxf(max16(st[0]), max16(st[1])));
  bf16x8 kf[2][4];                                     // K(kt+1) fragments: read right behind the barrier that publishes the tile
#pragma unroll
  for (int kh = 0; kh < 2; ++kh)
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[kh][s] = kfrag(1, kh, s);
  __syncthreads();            // iteration 0 overwrites K(0)'s buffer: every wave must have read it (a wave whose Q arrives late has not)

  auto body = [&](auto CURC, const int kt) {
    constexpr int cur = decltype(CURC)::value;
    const bool last = kt + 1 == ntiles;
    if (kt + 2 < ntiles) ks.issue(tw(kt + 2), kring + cur * DT_BYTES, wave);        // K(kt+2) replaces K(kt)
    if (!last) vs.issue(tw(kt + 1), vring + (cur ^ 1) * DT_BYTES, wave);            // V(kt+1) replaces V(kt-1)
    bf16x8 vf[2][2][2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[kh][s][dt] = vfrag(cur, kh, s, dt);
    const float mc = m * c;
    f32x16 sn[2] = {zero16(), zero16()};
    bf16x8 pf[2][2];
    // stage A: exponentials of key half 0 beside K(t+1).Q^T (the last iteration multiplies stale fragments; result unused)
#pragma unroll
    for (int r = 0; r < 16; ++r) st[0][r] = fast_exp2(fmaf(st[0][r], c, -mc));
    pf[0][0] = cvt8(st[0], 0); pf[0][1] = cvt8(st[0], 1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int s = 0; s < 4; ++s) sn[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kh][s], qf[s], sn[kh], 0, 0, 0);
    // stage B: P.V of half 0 beside the exponentials of half 1
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[0][s], lsum, 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][s][dt], pf[0][s], o[dt], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) st[1][r] = fast_exp2(fmaf(st[1][r], c, -mc));
    pf[1][0] = cvt8(st[1], 0); pf[1][1] = cvt8(st[1], 1);
    // stage C: P.V of half 1 beside the row maximum of S(t+1)
    float mx = fmaxf(max16(sn[0]), max16(sn[1]));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[1][s], lsum, 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][s][dt], pf[1][s], o[dt], 0, 0, 0);
    }
    mx = max_xhalf(mx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of K(kt+2), V(kt+1) have landed
    __syncthreads();
    if (!last) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[kh][s] = kfrag(cur, kh, s);      // K(kt+2), for the next iteration
      if (__any((mx - m) * c > ATTN_THR2)) {           // wave-uniform, rare after the first tiles
        const float m_new = fmaxf(m, mx);
        const float alpha = fast_exp2((m - m_new) * c);
        lsum[0] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        m = m_new;
      }
      st[0] = sn[0]; st[1] = sn[1];
    }
  };
  for (int kt = 0; kt < ntiles; kt += 2) { body(IC<0>{}, kt); body(IC<1>{}, kt + 1); }

  const int q = q0 + (lane & 31);
  const float l = lsum[0];
  const float inv = 1.0f / l;
  bf16_t* op = O.p + b * O.sb + (long)q * O.ld + h * D;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      uint2 u;
      u.x = pack2bf(o[dt][4 * rr] * inv, o[dt][4 * rr + 1] * inv);
      u.y = pack2bf(o[dt][4 * rr + 2] * inv, o[dt][4 * rr + 3] * inv);
      *reinterpret_cast<uint2*>(op + 32 * dt + 8 * rr + 4 * (lane >> 5)) = u;
    }
  if (lane < 32) lse2[((long)bh) * Tq + q] = m * c + log2f(l);
}

// =============================== delta = rowsum(dO * O) ======================================
__global__ void attn_delta_kernel(int heads, int Tq, AttnPtr O, AttnPtr dO, float* __restrict__ delta, int batch) {
  // eight lanes per (query, head) row of 64 elements, one 16-byte piece each: a wave's load instruction covers 1 KiB of contiguous
  // bytes (one lane per row read eight pieces 128 B apart: 16 instructions that each touched 64 lines); the eight partial sums meet
  // in a fixed order through three row_shr steps
  const long n = (long)batch * Tq * heads;
  const int sub = threadIdx.x & 7;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += ((long)gridDim.x * blockDim.x) >> 3) {
    int h = (int)(i % heads); long bq = i / heads; int q = (int)(bq % Tq); int b = (int)(bq / Tq);
    const uint4 a = *reinterpret_cast<const uint4*>(O.p + b * O.sb + (long)q * O.ld + h * D + sub * 8);
    const uint4 d = *reinterpret_cast<const uint4*>(dO.p + b * dO.sb + (long)q * dO.ld + h * D + sub * 8);
    const uint32_t* aw = reinterpret_cast<const uint32_t*>(&a); const uint32_t* dw = reinterpret_cast<const uint32_t*>(&d);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s += __uint_as_float(aw[e] << 16) * __uint_as_float(dw[e] << 16);
      s += __uint_as_float(aw[e] & 0xFFFF0000u) * __uint_as_float(dw[e] & 0xFFFF0000u);
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (sub == 0) delta[((long)(b * heads + h)) * Tq + q] = s;
  }
}

// =============================== backward: dQ ================================================
// FUSE_DELTA: delta = rowsum(dO * O) is computed here from the wave's resident dO fragments (and written out for the
// dK/dV kernel) instead of by a separate pass over O and dO.
template <bool FUSE_DELTA, bool FULL>
__device__ __forceinline__ void attn_bwd_dq_body(char* smem, const int bx, const int bh, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                 AttnPtr dO, AttnPtr O, const float* __restrict__ lse2,
                                                 float* __restrict__ delta, AttnOut dQ) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = bh / heads, h = bh - b * heads;
  const int q0 = bx * 128 + wave * 32;
  const bf16_t* Kb = K.p + b * K.sb + h * D;
  const bf16_t* Vb = V.p + b * V.sb + h * D;
  const float c = scale * LOG2E;
  const int q = q0 + (lane & 31);

  bf16x8 qf[4], dof[4];
  load_row_frags<FULL>(Q.p + b * Q.sb + h * D, Q.ld, q, Tq, lane, qf);
  load_row_frags<FULL>(dO.p + b * dO.sb + h * D, dO.ld, q, Tq, lane, dof);
  const float my_lse = (FULL || q < Tq) ? lse2[(long)bh * Tq + q] : INFINITY;
  float my_delta;
  if constexpr (FUSE_DELTA) {
    bf16x8 of[4];
    load_row_frags<FULL>(O.p + b * O.sb + h * D, O.ld, q, Tq, lane, of);
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) part = fmaf(bf2f((bf16_t)of[s][j]), bf2f((bf16_t)dof[s][j]), part);
    my_delta = part + __shfl_xor(part, 32);          // the other 32 head-dim elements of row q live in lane ^ 32
    if (lane < 32 && (FULL || q < Tq)) delta[(long)bh * Tq + q] = my_delta;
  } else {
    my_delta = (FULL || q < Tq) ? delta[(long)bh * Tq + q] : 0.f;
  }
  f32x16 negd;                 // C operand of the first dP MFMA: dP - delta comes out of the matrix pipe
#pragma unroll
  for (int r = 0; r < 16; ++r) negd[r] = -my_delta;

  f32x16 dq[2] = {zero16(), zero16()};
  const int ntiles = (Tk + TILE - 1) / TILE;
  uint4 rk[2], rv[2];
  tile_load<FULL>(Kb, K.ld, 0, Tk, t, rk);
  tile_load<FULL>(Vb, V.ld, 0, Tk, t, rv);
  tile_store(smem, t, rk);
  tile_store(smem + TILE_BYTES, t, rv);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    const char* kimg = smem + cur * 2 * TILE_BYTES;
    const char* vimg = kimg + TILE_BYTES;
    const bool more = kt + 1 < ntiles;
    if (more) {
      tile_load<FULL>(Kb, K.ld, (kt + 1) * TILE, Tk, t, rk);
      tile_load<FULL>(Vb, V.ld, (kt + 1) * TILE, Tk, t, rv);
    }
    const int kbase = kt * TILE;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      if (!FULL && kbase + 32 * kh >= Tk) break;      // this half lies entirely beyond Tk (wave-uniform)
      f32x16 st = zero16(), dp = negd;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(kimg, 32 * kh, s, lane), qf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(vimg, 32 * kh, s, lane), dof[s], dp, 0, 0, 0);
      }
      if (!FULL && kbase + TILE > Tk) {      // only the last key tile can be partial (uniform branch)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + 32 * kh + acc_row(r, lane) >= Tk) st[r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = fast_exp2(fmaf(st[r], c, -my_lse)) * dp[r];    // dS^T = P (dP - delta); the softmax scale multiplies dQ once
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 df = cvt8(st, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kimg, 32 * kh, s, 32 * dt, lane), df, dq[dt], 0, 0, 0);
      }
    }
    if (more) {
      tile_store(smem + (cur ^ 1) * 2 * TILE_BYTES, t, rk);
      tile_store(smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, t, rv);
    }
    __syncthreads();
  }
  if (FULL || q < Tq) {
    bf16_t* op = dQ.p + b * dQ.sb + (long)q * dQ.ld + h * D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 u;
        u.x = pack2bf(dq[dt][4 * rr] * scale, dq[dt][4 * rr + 1] * scale);
        u.y = pack2bf(dq[dt][4 * rr + 2] * scale, dq[dt][4 * rr + 3] * scale);
        *reinterpret_cast<uint2*>(op + 32 * dt + 8 * rr + 4 * (lane >> 5)) = u;
      }
  }
}

// =============================== backward: dK, dV ============================================
// workgroup = 128 keys (wave = 32 keys, K/V fragments resident); loop over 64-query tiles of Q, dO.
constexpr int DKV_SMEM = 4 * TILE_BYTES + 2 * 2 * TILE * 4;   // Q0 dO0 Q1 dO1, lse/delta x2
__device__ __forceinline__ void attn_bwd_dkv_body(char* smem, const int bx, const int bh, const int bz, const int nbx, const int nbh, const int nbz,
                                                  int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                  AttnPtr dO, const float* __restrict__ lse2,
                                                  const float* __restrict__ delta, AttnOut dK, AttnOut dV,
                                                  int tiles_per_split, float* __restrict__ part) {
  float* stat = reinterpret_cast<float*>(smem + 4 * TILE_BYTES);   // [buf][2][64]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = bh / heads, h = bh - b * heads;
  const int k0 = bx * 128 + wave * 32;
  const bf16_t* Qb = Q.p + b * Q.sb + h * D;
  const bf16_t* dOb = dO.p + b * dO.sb + h * D;
  const float c = scale * LOG2E;
  const int key = k0 + (lane & 31);

  bf16x8 kf[4], vf[4];
  load_row_frags(K.p + b * K.sb + h * D, K.ld, key, Tk, lane, kf);
  load_row_frags(V.p + b * V.sb + h * D, V.ld, key, Tk, lane, vf);

  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  const int ntiles_all = (Tq + TILE - 1) / TILE;
  const int qt_begin = bz * tiles_per_split;
  int qt_end = qt_begin + tiles_per_split; if (qt_end > ntiles_all) qt_end = ntiles_all;
  uint4 rq[2], rd[2];
  // per-query statistics of the next tile: thread t < 64 carries lse[q], thread 64 <= t < 128 carries -delta[q] (waves 0 / 1:
  // wave-uniform).  ONE plain register, loaded by a select on the source pointer: the earlier form (two values captured by
  // reference in divergent lambdas) was put on the stack by the compiler, and its scratch store behind the load carried an
  // s_waitcnt vmcnt(0) that also drained the Q / dO tile prefetch issued just before it -- every iteration waited for the
  // next tile's global loads BEFORE multiplying the current one.
  float rstat = 0.f;
  const float* stat_src = (t < 64) ? lse2 : delta;
  auto stat_load = [&](int qt) -> float {                     // the RAW value: nothing may consume it before stat_store (a use
    if (t >= 128) return 0.f;                                  // here would put the wait for the load in front of the MFMAs)
    int qq = qt * TILE + (t & 63);
    if (qq >= Tq) qq = Tq - 1;
    return stat_src[(long)bh * Tq + qq];
  };
  auto stat_store = [&](int buf, int qt, float v) {
    if (t < 128) {
      const bool inside = qt * TILE + (t & 63) < Tq;
      stat[buf * 128 + t] = (t < 64) ? (inside ? v : INFINITY) : (inside ? -v : 0.f);      // [0, 64): lse, [64, 128): -delta
    }
  };
  tile_load(Qb, Q.ld, qt_begin * TILE, Tq, t, rq);
  tile_load(dOb, dO.ld, qt_begin * TILE, Tq, t, rd);
  rstat = stat_load(qt_begin);
  tile_store(smem, t, rq);
  tile_store(smem + TILE_BYTES, t, rd);
  stat_store(0, qt_begin, rstat);
  __syncthreads();

  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const int cur = (qt - qt_begin) & 1;
    const char* qimg = smem + cur * 2 * TILE_BYTES;
    const char* doimg = qimg + TILE_BYTES;
    const float* lsev = stat + cur * 128;
    const float* delv = lsev + 64;
    const bool more = qt + 1 < qt_end;
    if (more) {
      tile_load(Qb, Q.ld, (qt + 1) * TILE, Tq, t, rq);
      tile_load(dOb, dO.ld, (qt + 1) * TILE, Tq, t, rd);
      rstat = stat_load(qt + 1);
    }
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      if (k0 >= Tk) break;                   // this wave's 32 keys lie entirely beyond Tk (cross-attention: 77 keys, wave 3 idles)
      // S[q][key], dP[q][key]  (rows = query on the register axis, key on the lane)
      f32x16 sa = zero16(), dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = delv[32 * qh + acc_row(r, lane)];     // -delta: dP - delta comes out of the matrix pipe
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(qimg, 32 * qh, s, lane), kf[s], sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(doimg, 32 * qh, s, lane), vf[s], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = 32 * qh + acc_row(r, lane);
        const float p = fast_exp2(fmaf(sa[r], c, -lsev[qr]));
        sa[r] = p;                                          // P
        dp[r] = p * dp[r];                                  // dS = P (dP - delta); the softmax scale multiplies dK once
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf = cvt8(sa, s), df = cvt8(dp, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_transposed(doimg, 32 * qh, s, 32 * dt, lane), dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, frag_transposed(qimg, 32 * qh, s, 32 * dt, lane), dk[dt], 0, 0, 0);
        }
      }
    }
    if (more) {
      tile_store(smem + (cur ^ 1) * 2 * TILE_BYTES, t, rq);
      tile_store(smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, t, rd);
      stat_store(cur ^ 1, qt + 1, rstat);
    }
    __syncthreads();
  }
  // accumulators: row = key (register axis), col = d (lane)
  if (nbz == 1) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = k0 + acc_row(r, lane);
        if (kk < Tk) {
          const int d = 32 * dt + (lane & 31);
          dK.p[b * dK.sb + (long)kk * dK.ld + h * D + d] = f2bf(dk[dt][r] * scale);
          dV.p[b * dV.sb + (long)kk * dV.ld + h * D + d] = f2bf(dv[dt][r]);
        }
      }
  } else {
    // fp32 partials part[z][bh][kpad][2][64]; summed in split order by attn_dkv_reduce_kernel
    const int kpad = nbx * 128;
    float* base = part + (((long)bz * nbh + bh) * kpad) * 128;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = k0 + acc_row(r, lane);
        const int d = 32 * dt + (lane & 31);
        base[(long)kk * 128 + d] = dk[dt][r] * scale;
        base[(long)kk * 128 + 64 + d] = dv[dt][r];
      }
  }
}

template <bool FUSE_DELTA, bool FULL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                             AttnPtr dO, AttnPtr O, const float* __restrict__ lse2,
                                                             float* __restrict__ delta, AttnOut dQ) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  int bx, bh, bz_; attn_block(G, bx, bh, bz_);
  attn_bwd_dq_body<FUSE_DELTA, FULL>(smem, bx, bh, heads, Tq, Tk, scale, Q, K, V, dO, O, lse2, delta, dQ);
}
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                              AttnPtr dO, const float* __restrict__ lse2,
                                                              const float* __restrict__ delta, AttnOut dK, AttnOut dV,
                                                              int tiles_per_split, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) char smem[DKV_SMEM];
  int bx, bh, bz; attn_block(G, bx, bh, bz);
  attn_bwd_dkv_body(smem, bx, bh, bz, G.nx, G.ny, G.nz, heads, Tq, Tk, scale, Q, K, V, dO, lse2, delta, dK, dV,
                    tiles_per_split, part);
}
// dQ and dK/dV workgroups of one self-attention in ONE launch (blockIdx.z: 0 = dQ role, 1 = dK/dV role; delta from its own
// small kernel in front).  Two launches of 640 (T = 1024) or 1280 (T = 4096) equal workgroups on 512 slots each run 2 or 3
// rounds with the last one a quarter or half full; 1280 / 2560 mixed workgroups fill 2.5 / 5 rounds -- the other role's
// workgroups are the filler of each role's tail.
__global__ __launch_bounds__(256, 1) void attn_bwd_merged_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                                 AttnPtr dO, AttnPtr O, const float* __restrict__ lse2,
                                                                 float* __restrict__ delta, AttnOut dQ, AttnOut dK, AttnOut dV) {
  __shared__ __attribute__((aligned(16))) char smem[DKV_SMEM];
  int bx, bh, role; attn_block(G, bx, bh, role);
  if (role == 0) attn_bwd_dq_body<false, true>(smem, bx, bh, heads, Tq, Tk, scale, Q, K, V, dO, O, lse2, delta, dQ);
  else attn_bwd_dkv_body(smem, bx, bh, 0, G.nx, G.ny, 1, heads, Tq, Tk, scale, Q, K, V, dO, lse2, delta, dK, dV, Tq / TILE, nullptr);
}

// =============================== backward, tiles staged by LDS-DMA (round 5) ===================
// The dQ and dK/dV bodies above with the forward kernel's tile transport: K / V (dQ role) or Q / dO (dK/dV role) tiles move
// global -> LDS by buffer_load ... lds into rings of two UNPADDED [64][128 B] images, XOR-swizzled on the source address and on the
// fragment reads (chunk c of row r at position c ^ swz(r)): no staging registers, no ds_write pass, and both read kinds of a tile
// -- ds_read_b128 rows for S / dP, ds_read_b64_tr_b16 for dQ / dV / dK -- are bank-conflict free on ONE image (the padded 144-byte
// pitch left the transposed reads 2-way: 21 % of the backward's LDS cycles in profiles/r04_d_pmc_sq.json).  Same MFMAs in the same
// order on the same values: bit-identical to the register-staged bodies (tests/test_kernels_gpu.py toggles ATTN_PIPE bit 3).
// Shapes: Tq % 128 == 0 and Tk % 128 == 0 (every self-attention of the UNet); everything else keeps the plain kernels.
struct DmaFragAddr {
  const char* row[4];        // row-major fragment, k-step s: rows l&31 (+32 per half), 16-byte chunk 2s + (l>>5)
  const char* tlo[2];        // transposed fragment, column half dt: image rows 4(l>>5) + ((l&15)>>2) (+16 per k-step, +32 per half)
  const char* thi[2];        //   ... and 8 rows further down
  __device__ __forceinline__ void init(const char* img, int lane) {
    const int hh = lane >> 5, r32 = lane & 31, i16 = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) row[s_] = img + r32 * 128 + (((2 * s_ + hh) ^ swz(r32)) << 4);
    const int rl = 4 * hh + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int chunk = 4 * dt + 2 * g1 + ((i16 & 3) >> 1);
      tlo[dt] = img + rl * 128 + ((chunk ^ swz(rl)) << 4) + 8 * (i16 & 1);
      thi[dt] = img + (rl + 8) * 128 + ((chunk ^ swz(rl + 8)) << 4) + 8 * (i16 & 1);
    }
  }
  __device__ __forceinline__ bf16x8 rowfrag(int off, int half, int s_) const {
    return *reinterpret_cast<const bf16x8*>(row[s_] + off + half * 32 * 128);
  }
  __device__ __forceinline__ bf16x8 trfrag(int off, int half, int ks, int dt) const {
    const int o = off + (32 * half + 16 * ks) * 128;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(tlo[dt] + o));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(thi[dt] + o));
    bf16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
  }
};

template <bool FUSE_DELTA>
__device__ __forceinline__ void attn_bwd_dq_dma_body(char* smem, const int bx, const int bh, int heads, int Tq, int Tk, float scale, AttnPtr Q,
                                                     AttnPtr K, AttnPtr V, AttnPtr dO, AttnPtr O, const float* __restrict__ lse2,
                                                     float* __restrict__ delta, AttnOut dQ) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = bh / heads, h = bh - b * heads;
  const int q0 = bx * 128 + wave * 32;
  const float c = scale * LOG2E;
  const int q = q0 + (lane & 31);
  const unsigned kring = lds_addr_of(smem), vring = kring + 2 * DT_BYTES;      // K ring [2], V ring [2]
  DmaStream ks, vs;
  ks.init(K.p + b * K.sb + h * D, K.ld, lane, wave);
  vs.init(V.p + b * V.sb + h * D, V.ld, lane, wave);
  const int ntiles = Tk / TILE;                        // even (host-checked)
  ks.issue(0, kring, wave); vs.issue(0, vring, wave);

  bf16x8 qf[4], dof[4];
  load_row_frags<true>(Q.p + b * Q.sb + h * D, Q.ld, q, Tq, lane, qf);
  load_row_frags<true>(dO.p + b * dO.sb + h * D, dO.ld, q, Tq, lane, dof);
  const float my_lse = lse2[(long)bh * Tq + q];
  float my_delta;
  if constexpr (FUSE_DELTA) {
    bf16x8 of[4];
    load_row_frags<true>(O.p + b * O.sb + h * D, O.ld, q, Tq, lane, of);
    float part = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int j = 0; j < 8; ++j) part = fmaf(bf2f((bf16_t)of[s_][j]), bf2f((bf16_t)dof[s_][j]), part);
    my_delta = part + __shfl_xor(part, 32);
    if (lane < 32) delta[(long)bh * Tq + q] = my_delta;
  } else {
    my_delta = delta[(long)bh * Tq + q];
  }
  f32x16 negd;
#pragma unroll
  for (int r = 0; r < 16; ++r) negd[r] = -my_delta;
  DmaFragAddr fa; fa.init(smem, lane);                 // offsets: K buffer u at u * DT_BYTES, V buffer u at (2 + u) * DT_BYTES
  f32x16 dq[2] = {zero16(), zero16()};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto body = [&](auto CURC, const int kt) {
    constexpr int cur = decltype(CURC)::value;
    constexpr int ko = cur * DT_BYTES, vo = (2 + cur) * DT_BYTES;
    if (kt + 1 < ntiles) {                               // the other buffers were last read in iteration kt - 1 (barrier below)
      ks.issue(kt + 1, kring + (cur ^ 1) * DT_BYTES, wave);
      vs.issue(kt + 1, vring + (cur ^ 1) * DT_BYTES, wave);
    }
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      f32x16 st = zero16(), dp = negd;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(ko, kh, s_), qf[s_], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(vo, kh, s_), dof[s_], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = fast_exp2(fmaf(st[r], c, -my_lse)) * dp[r];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        bf16x8 df = cvt8(st, s_);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(ko, kh, s_, dt), df, dq[dt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt + 1 have landed
    __syncthreads();
  };
  for (int kt = 0; kt < ntiles; kt += 2) { body(IC<0>{}, kt); body(IC<1>{}, kt + 1); }

  bf16_t* op = dQ.p + b * dQ.sb + (long)q * dQ.ld + h * D;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      uint2 u;
      u.x = pack2bf(dq[dt][4 * rr] * scale, dq[dt][4 * rr + 1] * scale);
      u.y = pack2bf(dq[dt][4 * rr + 2] * scale, dq[dt][4 * rr + 3] * scale);
      *reinterpret_cast<uint2*>(op + 32 * dt + 8 * rr + 4 * (lane >> 5)) = u;
    }
}

// dK / dV role: 128 keys per workgroup (wave = 32 keys, K / V fragments resident), Q / dO tiles of 64 queries through the rings;
// lse / -delta of a tile ride in LDS beside them ([buf][2][64] floats behind the four images), loaded a tile ahead as before.
constexpr int DKV_DMA_SMEM = 4 * DT_BYTES + 2 * 2 * TILE * 4;
__device__ __forceinline__ void attn_bwd_dkv_dma_body(char* smem, const int bx, const int bh, int heads, int Tq, int Tk, float scale, AttnPtr Q,
                                                      AttnPtr K, AttnPtr V, AttnPtr dO, const float* __restrict__ lse2,
                                                      const float* __restrict__ delta, AttnOut dK, AttnOut dV) {
  float* stat = reinterpret_cast<float*>(smem + 4 * DT_BYTES);   // [buf][2][64]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = bh / heads, h = bh - b * heads;
  const int k0 = bx * 128 + wave * 32;
  const float c = scale * LOG2E;
  const int key = k0 + (lane & 31);
  const unsigned qring = lds_addr_of(smem), dring = qring + 2 * DT_BYTES;      // Q ring [2], dO ring [2]
  DmaStream qs, ds;
  qs.init(Q.p + b * Q.sb + h * D, Q.ld, lane, wave);
  ds.init(dO.p + b * dO.sb + h * D, dO.ld, lane, wave);
  const int ntiles = Tq / TILE;                        // even (host-checked)
  qs.issue(0, qring, wave); ds.issue(0, dring, wave);

  bf16x8 kf[4], vf[4];
  load_row_frags<true>(K.p + b * K.sb + h * D, K.ld, key, Tk, lane, kf);
  load_row_frags<true>(V.p + b * V.sb + h * D, V.ld, key, Tk, lane, vf);
  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  // per-query statistics of the next tile: thread t < 64 carries lse[q], 64 <= t < 128 carries delta[q] (one plain register: see
  // attn_bwd_dkv_body); ordinary loads -- they are waited for (vmcnt(0)) together with the DMA right before the barrier
  const float* stat_src = ((t < 64) ? lse2 : delta) + (long)bh * Tq + (t & 63);
  float rstat = (t < 128) ? stat_src[0] : 0.f;
  DmaFragAddr fa; fa.init(smem, lane);                 // offsets: Q buffer u at u * DT_BYTES, dO buffer u at (2 + u) * DT_BYTES
  if (t < 128) stat[t] = (t < 64) ? rstat : -rstat;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto body = [&](auto CURC, const int qt) {
    constexpr int cur = decltype(CURC)::value;
    constexpr int qo = cur * DT_BYTES, dofs = (2 + cur) * DT_BYTES;
    const float* lsev = stat + cur * 128;
    const float* delv = lsev + 64;
    const bool more = qt + 1 < ntiles;
    if (more) {
      qs.issue(qt + 1, qring + (cur ^ 1) * DT_BYTES, wave);
      ds.issue(qt + 1, dring + (cur ^ 1) * DT_BYTES, wave);
      if (t < 128) rstat = stat_src[(qt + 1) * TILE];
    }
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      f32x16 sa = zero16(), dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = delv[32 * qh + acc_row(r, lane)];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(qo, qh, s_), kf[s_], sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(dofs, qh, s_), vf[s_], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = 32 * qh + acc_row(r, lane);
        const float p = fast_exp2(fmaf(sa[r], c, -lsev[qr]));
        sa[r] = p;
        dp[r] = p * dp[r];
      }
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        bf16x8 pf = cvt8(sa, s_), df = cvt8(dp, s_);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, fa.trfrag(dofs, qh, s_, dt), dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, fa.trfrag(qo, qh, s_, dt), dk[dt], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // DMA pieces of tile qt + 1 and its statistic have landed
    if (more && t < 128) stat[(cur ^ 1) * 128 + t] = (t < 64) ? rstat : -rstat;
    __syncthreads();
  };
  for (int qt = 0; qt < ntiles; qt += 2) { body(IC<0>{}, qt); body(IC<1>{}, qt + 1); }

#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = k0 + acc_row(r, lane);
      const int d = 32 * dt + (lane & 31);
      dK.p[b * dK.sb + (long)kk * dK.ld + h * D + d] = f2bf(dk[dt][r] * scale);
      dV.p[b * dV.sb + (long)kk * dV.ld + h * D + d] = f2bf(dv[dt][r]);
    }
}

template <bool FUSE_DELTA>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_dma_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                                 AttnPtr dO, AttnPtr O, const float* __restrict__ lse2,
                                                                 float* __restrict__ delta, AttnOut dQ) {
  __shared__ __attribute__((aligned(1024))) char smem[4 * DT_BYTES];
  int bx, bh, bz_; attn_block(G, bx, bh, bz_);
  attn_bwd_dq_dma_body<FUSE_DELTA>(smem, bx, bh, heads, Tq, Tk, scale, Q, K, V, dO, O, lse2, delta, dQ);
}
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_dma_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                                  AttnPtr dO, const float* __restrict__ lse2,
                                                                  const float* __restrict__ delta, AttnOut dK, AttnOut dV) {
  __shared__ __attribute__((aligned(1024))) char smem[DKV_DMA_SMEM];
  int bx, bh, bz_; attn_block(G, bx, bh, bz_);
  attn_bwd_dkv_dma_body(smem, bx, bh, heads, Tq, Tk, scale, Q, K, V, dO, lse2, delta, dK, dV);
}
__global__ __launch_bounds__(256, 1) void attn_bwd_merged_dma_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                                     AttnPtr dO, AttnPtr O, const float* __restrict__ lse2,
                                                                     float* __restrict__ delta, AttnOut dQ, AttnOut dK, AttnOut dV) {
  __shared__ __attribute__((aligned(1024))) char smem[DKV_DMA_SMEM];
  int bx, bh, role; attn_block(G, bx, bh, role);
  if (role == 0) attn_bwd_dq_dma_body<false>(smem, bx, bh, heads, Tq, Tk, scale, Q, K, V, dO, O, lse2, delta, dQ);
  else attn_bwd_dkv_dma_body(smem, bx, bh, heads, Tq, Tk, scale, Q, K, V, dO, lse2, delta, dK, dV);
}

// =============================== backward, short key axis (cross-attention): one kernel ======
// Tk <= 128 (the text context: 77 keys): K and V of a (batch, head) fit in LDS whole, so ONE workgroup computes dQ of its queries
// AND its share of dK / dV from one pass over Q / dO -- instead of a dQ kernel, a query-split dK/dV kernel and their 2 x re-read
// of Q and dO (the three-launch form: 14 + 20 + 8 us per layer at (4,20,1024,77) for 4 GFLOP, all of it launch ramps and
// dependent memory round trips).  Workgroup = 4 waves, one 128-query tile at a time, `tiles_per_wg` tiles of one (batch, head):
//   part A (wave w = queries 32w .. 32w+31 of the tile): S^T = K.Q^T and dP^T = V.dO^T - delta with the key on the register axis,
//     dS^T = P^T (dP^T - delta), dQ^T += K^T.dS^T -- the arithmetic of attn_bwd_dq_kernel; delta = rowsum(dO o O) from the wave's
//     own fragments, left in LDS with the row's lse for part B;
//   part B (wave w = keys 32w .. 32w+31, idle when those lie beyond Tk): S = Q.K^T, dP = dO.V^T - delta with the key on the
//     lane, dV^T += dO^T.P, dK^T += Q^T.dS -- the arithmetic of attn_bwd_dkv_kernel, accumulated over the workgroup's tiles.
// dK / dV leave as fp32 partials [z][bh][128 keys][dK 64 | dV 64] summed in split order by attn_dkv_reduce_kernel (no atomics),
// or directly when one workgroup covers all queries.  Keys beyond Tk: zero K / V rows and a -1e30 initial score accumulator (P = 0);
// queries beyond Tq: zero rows, lse = +inf (P = 0), no dQ store.
constexpr int XIMG = 128 * PITCH;                    // one [128][64] image
constexpr int X_SMEM = 4 * XIMG + 2 * 128 * 4;       // K, V, Q, dO images + lse / -delta of the tile

__device__ __forceinline__ void ximg_load(const bf16_t* base, long ld, int row0, int nrows, int t, char* img) {
  uint4 r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = t + 256 * i, row = c >> 3, dc = c & 7;
    r[i] = (row0 + row < nrows) ? *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * ld + dc * 8) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = t + 256 * i;
    *reinterpret_cast<uint4*>(img + (c >> 3) * PITCH + (c & 7) * 16) = r[i];
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_cross_kernel(AttnGrid G, int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                                AttnPtr dO, AttnPtr O, const float* __restrict__ lse2, AttnOut dQ,
                                                                AttnOut dK, AttnOut dV, int tiles_per_wg, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) char smem[X_SMEM];
  char* const kimg = smem; char* const vimg = smem + XIMG; char* const qimg = smem + 2 * XIMG; char* const doimg = smem + 3 * XIMG;
  float* const lsev = reinterpret_cast<float*>(smem + 4 * XIMG);   // [128]
  float* const delv = lsev + 128;                                   // [128], holds -delta
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int bx, bh, bz_; attn_block(G, bx, bh, bz_);
  const int b = bh / heads, h = bh - b * heads;
  const float c = scale * LOG2E;
  const int nkb = (Tk + 31) >> 5;                                   // key blocks of 32 (<= 4)
  const bf16_t* Qb = Q.p + b * Q.sb + h * D;
  const bf16_t* dOb = dO.p + b * dO.sb + h * D;
  const bf16_t* Ob = O.p + b * O.sb + h * D;
  ximg_load(K.p + b * K.sb + h * D, K.ld, 0, Tk, t, kimg);
  ximg_load(V.p + b * V.sb + h * D, V.ld, 0, Tk, t, vimg);

  const int ntiles_all = (Tq + 127) >> 7;
  const int qt_begin = bx * tiles_per_wg;
  int qt_end = qt_begin + tiles_per_wg; if (qt_end > ntiles_all) qt_end = ntiles_all;
  // part B: this wave's keys
  const int key = 32 * wave + (lane & 31);
  const bool bwave = wave < nkb;                                    // wave-uniform
  const float kmask = key < Tk ? 0.f : -1e30f;
  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};

  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const int q = qt * 128 + 32 * wave + (lane & 31);
    bf16x8 of[4];
    load_row_frags(Ob, O.ld, q, Tq, lane, of);
    const float my_lse = q < Tq ? lse2[(long)bh * Tq + q] : INFINITY;
    ximg_load(Qb, Q.ld, qt * 128, Tq, t, qimg);
    ximg_load(dOb, dO.ld, qt * 128, Tq, t, doimg);
    __syncthreads();
    // ---------------- part A: dQ of this wave's 32 queries ----------------
    {
      bf16x8 qf[4], dof[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { qf[s] = frag_rowmajor(qimg, 32 * wave, s, lane); dof[s] = frag_rowmajor(doimg, 32 * wave, s, lane); }
      float pd = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pd = fmaf(bf2f((bf16_t)of[s][j]), bf2f((bf16_t)dof[s][j]), pd);
      const float my_delta = pd + __shfl_xor(pd, 32);
      if (lane < 32) { lsev[32 * wave + lane] = my_lse; delv[32 * wave + lane] = -my_delta; }
      f32x16 dq[2] = {zero16(), zero16()};
#pragma unroll 1
      for (int kb = 0; kb < nkb; ++kb) {
        f32x16 st, dp;
        const int klim = Tk - 32 * kb;                 // -1e30 in the score rows of keys beyond Tk (last key block only): P = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = acc_row(r, lane) < klim ? 0.f : -1e30f; dp[r] = -my_delta; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(kimg, 32 * kb, s, lane), qf[s], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(vimg, 32 * kb, s, lane), dof[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = fast_exp2(fmaf(st[r], c, -my_lse)) * dp[r];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 df = cvt8(st, s);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(kimg, 32 * kb, s, 32 * dt, lane), df, dq[dt], 0, 0, 0);
        }
      }
      if (q < Tq) {
        bf16_t* op = dQ.p + b * dQ.sb + (long)q * dQ.ld + h * D;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            uint2 u;
            u.x = pack2bf(dq[dt][4 * rr] * scale, dq[dt][4 * rr + 1] * scale);
            u.y = pack2bf(dq[dt][4 * rr + 2] * scale, dq[dt][4 * rr + 3] * scale);
            *reinterpret_cast<uint2*>(op + 32 * dt + 8 * rr + 4 * (lane >> 5)) = u;
          }
      }
    }
    __syncthreads();                                   // lse / -delta of the whole tile are in LDS
    // ---------------- part B: dK / dV of this wave's 32 keys over the tile's 128 queries ----------------
    if (bwave) {
      bf16x8 kf[4], vf[4];                             // this wave's K / V rows as B operands (the images stay in LDS for the whole kernel)
#pragma unroll
      for (int s = 0; s < 4; ++s) { kf[s] = frag_rowmajor(kimg, 32 * wave, s, lane); vf[s] = frag_rowmajor(vimg, 32 * wave, s, lane); }
#pragma unroll 1
      for (int qb = 0; qb < 4; ++qb) {
        f32x16 sa, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = kmask; dp[r] = delv[32 * qb + acc_row(r, lane)]; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(qimg, 32 * qb, s, lane), kf[s], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rowmajor(doimg, 32 * qb, s, lane), vf[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(fmaf(sa[r], c, -lsev[32 * qb + acc_row(r, lane)]));
          sa[r] = p;
          dp[r] = p * dp[r];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 pf = cvt8(sa, s), df = cvt8(dp, s);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, frag_transposed(doimg, 32 * qb, s, 32 * dt, lane), dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, frag_transposed(qimg, 32 * qb, s, 32 * dt, lane), dk[dt], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();                                   // the next tile overwrites the Q / dO images and the statistics
  }
  if (!bwave) return;
  const int k0 = 32 * wave;
  if (G.nx == 1) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = k0 + acc_row(r, lane);
        if (kk < Tk) {
          const int d = 32 * dt + (lane & 31);
          dK.p[b * dK.sb + (long)kk * dK.ld + h * D + d] = f2bf(dk[dt][r] * scale);
          dV.p[b * dV.sb + (long)kk * dV.ld + h * D + d] = f2bf(dv[dt][r]);
        }
      }
  } else {
    float* base = part + (((long)bx * G.ny + bh) * 128) * 128;      // [z][bh][128 keys][dK 64 | dV 64]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = k0 + acc_row(r, lane);
        const int d = 32 * dt + (lane & 31);
        base[(long)kk * 128 + d] = dk[dt][r] * scale;
        base[(long)kk * 128 + 64 + d] = dv[dt][r];
      }
  }
}

__global__ void attn_dkv_reduce_kernel(int heads, int Tk, int kpad, int nsplit, int BH, const float* __restrict__ part, AttnOut dK, AttnOut dV) {
  long n = (long)BH * Tk * 128;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i & 127); long bhl; int kk; divmod(i >> 7, Tk, bhl, kk); int bh = (int)bhl;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += part[(((long)z * BH + bh) * kpad + kk) * 128 + c];
    int b = bh / heads, h = bh - b * heads;
    if (c < 64) dK.p[b * dK.sb + (long)kk * dK.ld + h * D + c] = f2bf(s);
    else dV.p[b * dV.sb + (long)kk * dV.ld + h * D + (c - 64)] = f2bf(s);
  }
}

// bit of option ATTN_XCD per kernel family: 0 forward, 1 dQ and dK / dV kernels, 2 the short-key one-kernel backward, 3 the merged
// backward in its LDS-DMA form
// (bit < 0: the plain order -- the merged backward)
AttnGrid attn_grid(int nx, int ny, int nz, int bit) { return AttnGrid{nx, ny, nz, bit < 0 ? 0 : (az_opt(AZ_OPT_ATTN_XCD) >> bit) & 1}; }
dim3 grid1(const AttnGrid& g) { return dim3((unsigned)(g.nx * g.ny * g.nz)); }

int check_ptr(const void* p, long ld, long sb) {
  if (((uintptr_t)p & 15) || (ld & 7) || (sb & 7)) return AZ_ERR_ARG(50);
  return AZ_OK;
}

}  // namespace

extern "C" {

int az_attn_fwd(int batch, int heads, int Tq, int Tk, float scale, const void* Q, long ldq, long sq, const void* K,
                long ldk, long sk, const void* V, long ldv, long sv, void* O, long ldo, long so, void* lse,
                void* stream) {
  if (batch <= 0 || heads <= 0 || Tq <= 0 || Tk <= 0) return AZ_ERR_ARG(51);
  int rc;
  if ((rc = check_ptr(Q, ldq, sq)) || (rc = check_ptr(K, ldk, sk)) || (rc = check_ptr(V, ldv, sv)) || (rc = check_ptr(O, ldo, so))) return rc;
  const AttnGrid G = attn_grid((Tq + 127) / 128, batch * heads, 1, 0);
  const dim3 grid = grid1(G);
  if ((az_opt(AZ_OPT_ATTN_PIPE) & 1) && (Tq % 128) == 0 && (Tk % (2 * TILE)) == 0) {
    az_launch(attn_fwd_dma_kernel, grid, dim3(256), 0, (hipStream_t)stream, G, heads, Tq, Tk, scale,
              AttnPtr{(const bf16_t*)Q, ldq, sq}, AttnPtr{(const bf16_t*)K, ldk, sk}, AttnPtr{(const bf16_t*)V, ldv, sv},
              AttnOut{(bf16_t*)O, ldo, so}, (float*)lse);
  } else if ((Tq % 128) == 0 && (Tk % TILE) == 0)
    az_launch(attn_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, G, heads, Tq, Tk, scale,
                       AttnPtr{(const bf16_t*)Q, ldq, sq}, AttnPtr{(const bf16_t*)K, ldk, sk}, AttnPtr{(const bf16_t*)V, ldv, sv},
                       AttnOut{(bf16_t*)O, ldo, so}, (float*)lse);
  else
    az_launch(attn_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, G, heads, Tq, Tk, scale,
                       AttnPtr{(const bf16_t*)Q, ldq, sq}, AttnPtr{(const bf16_t*)K, ldk, sk}, AttnPtr{(const bf16_t*)V, ldv, sv},
                       AttnOut{(bf16_t*)O, ldo, so}, (float*)lse);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_attn_bwd(int batch, int heads, int Tq, int Tk, float scale, const void* Q, long ldq, long sq, const void* K,
                long ldk, long sk, const void* V, long ldv, long sv, const void* O, long ldo, long so, const void* dO,
                long lddo, long sdo, const void* lse, void* delta, void* dQ, long lddq, long sdq, void* dK, long lddk,
                long sdk, void* dV, long lddv, long sdv, void* workspace, long workspace_bytes, int parts, void* stream) {
  if (batch <= 0 || heads <= 0 || Tq <= 0 || Tk <= 0) return AZ_ERR_ARG(52);
  if (parts == 0) parts = 7;
  int rc;
  if ((rc = check_ptr(Q, ldq, sq)) || (rc = check_ptr(K, ldk, sk)) || (rc = check_ptr(V, ldv, sv)) || (rc = check_ptr(O, ldo, so)) ||
      (rc = check_ptr(dO, lddo, sdo)) || (rc = check_ptr(dQ, lddq, sdq)) || (rc = check_ptr(dK, lddk, sdk)) || (rc = check_ptr(dV, lddv, sdv))) return rc;
  hipStream_t st = (hipStream_t)stream;
  AttnPtr q{(const bf16_t*)Q, ldq, sq}, k{(const bf16_t*)K, ldk, sk}, v{(const bf16_t*)V, ldv, sv}, o{(const bf16_t*)O, ldo, so},
      d_o{(const bf16_t*)dO, lddo, sdo};
  long n = (long)batch * Tq * heads;
  int g = (int)((n * 8 + 255) / 256); if (g > 4096) g = 4096;      // attn_delta_kernel: eight lanes per (query, head)
  // bit 3: the LDS-DMA forms of the dQ / dK-dV bodies (self-attention shapes: whole 128-row blocks on both axes)
  const bool dma = (az_opt(AZ_OPT_ATTN_PIPE) & 8) && (Tq % 128) == 0 && (Tk % 128) == 0;
  if (parts == 7 && (az_opt(AZ_OPT_ATTN_PIPE) & 2) && Tq == Tk && (Tq % 128) == 0 && (long)(Tq / 128) * batch * heads <= 768) {
    az_launch(attn_delta_kernel, dim3(g), dim3(256), 0, st, heads, Tq, o, d_o, (float*)delta, batch);
    const AttnGrid G = attn_grid(Tq / 128, batch * heads, 2, dma ? 3 : -1);
    if (dma)
      az_launch(attn_bwd_merged_dma_kernel, grid1(G), dim3(256), 0, st, G, heads, Tq, Tk, scale, q, k, v, d_o, o, (const float*)lse,
                (float*)delta, AttnOut{(bf16_t*)dQ, lddq, sdq}, AttnOut{(bf16_t*)dK, lddk, sdk}, AttnOut{(bf16_t*)dV, lddv, sdv});
    else
    az_launch(attn_bwd_merged_kernel, grid1(G), dim3(256), 0, st, G, heads, Tq, Tk, scale, q, k, v, d_o, o, (const float*)lse,
              (float*)delta, AttnOut{(bf16_t*)dQ, lddq, sdq}, AttnOut{(bf16_t*)dK, lddk, sdk}, AttnOut{(bf16_t*)dV, lddv, sdv});
    AZ_CHECK_LAUNCH();
    return AZ_OK;
  }
  if (parts == 7 && Tk <= 128 && (az_opt(AZ_OPT_ATTN_PIPE) & 4)) {
    // cross-attention: dQ, dK, dV from one kernel; the query axis is split over workgroups until the grid has ~ATTN_SPLIT_TARGET of them
    const int BHx = batch * heads, ntile = (Tq + 127) / 128;
    int nsplit = (az_opt(AZ_OPT_ATTN_SPLIT_TARGET) + BHx - 1) / BHx;
    if (nsplit > ntile) nsplit = ntile;
    if (nsplit > 1 && !workspace) nsplit = 1;
    while (nsplit > 1 && (long)nsplit * BHx * 128 * 128 * 4 > workspace_bytes) --nsplit;
    const int tpw = (ntile + nsplit - 1) / nsplit;
    nsplit = (ntile + tpw - 1) / tpw;
    AttnOut dk{(bf16_t*)dK, lddk, sdk}, dv{(bf16_t*)dV, lddv, sdv};
    const AttnGrid G = attn_grid(nsplit, BHx, 1, 2);
    az_launch(attn_bwd_cross_kernel, grid1(G), dim3(256), 0, st, G, heads, Tq, Tk, scale, q, k, v, d_o, o, (const float*)lse,
              AttnOut{(bf16_t*)dQ, lddq, sdq}, dk, dv, tpw, (float*)workspace);
    AZ_CHECK_LAUNCH();
    if (nsplit > 1) {
      long nred = (long)BHx * Tk * 128;
      int gr = (int)((nred + 255) / 256); if (gr > 2048) gr = 2048;
      az_launch(attn_dkv_reduce_kernel, dim3(gr), dim3(256), 0, st, heads, Tk, 128, nsplit, BHx, (const float*)workspace, dk, dv);
      AZ_CHECK_LAUNCH();
    }
    return AZ_OK;
  }
  if ((parts & 1) && !(parts & 2)) {
    az_launch(attn_delta_kernel, dim3(g), dim3(256), 0, st, heads, Tq, o, d_o, (float*)delta, batch);
    AZ_CHECK_LAUNCH();
  }
  if (parts & 2) {
    const bool full = (Tq % 128) == 0 && (Tk % TILE) == 0;
    const AttnGrid G = attn_grid((Tq + 127) / 128, batch * heads, 1, 1);
#define AZ_DQ(FD, FL) az_launch((attn_bwd_dq_kernel<FD, FL>), grid1(G), dim3(256), 0, st, G, heads, Tq, Tk, scale, \
                                         q, k, v, d_o, o, (const float*)lse, (float*)delta, AttnOut{(bf16_t*)dQ, lddq, sdq})
#define AZ_DQD(FD) az_launch((attn_bwd_dq_dma_kernel<FD>), grid1(G), dim3(256), 0, st, G, heads, Tq, Tk, scale, \
                                         q, k, v, d_o, o, (const float*)lse, (float*)delta, AttnOut{(bf16_t*)dQ, lddq, sdq})
    if (dma) { if (parts & 1) AZ_DQD(true); else AZ_DQD(false); }
    else if (parts & 1) { if (full) AZ_DQ(true, true); else AZ_DQ(true, false); }      // delta rides on the dQ kernel's resident dO fragments
    else { if (full) AZ_DQ(false, true); else AZ_DQ(false, false); }
#undef AZ_DQ
#undef AZ_DQD
    AZ_CHECK_LAUNCH();
  }
  if (!(parts & 4)) return AZ_OK;
  // few key blocks (cross-attention: Tk = 77): split the query range over gridDim.z to fill the chip
  const int kblocks = (Tk + 127) / 128, BH = batch * heads, qtiles = (Tq + TILE - 1) / TILE;
  int nsplit = 1;
  if (kblocks * BH < 384 && qtiles >= 4 && workspace) {
    const int target = az_opt(AZ_OPT_ATTN_SPLIT_TARGET);   // workgroups aimed at: 384 beats 768 by 0.6-1 ms per micro-step in the two-stream step (same-box A/B)
    nsplit = (target + kblocks * BH - 1) / (kblocks * BH);
    if (nsplit > qtiles / 2) nsplit = qtiles / 2;
    while (nsplit > 1 && (long)nsplit * BH * kblocks * 128 * 128 * 4 > workspace_bytes) --nsplit;
    if (nsplit < 1) nsplit = 1;
  }
  const int tps = (qtiles + nsplit - 1) / nsplit;
  nsplit = (qtiles + tps - 1) / tps;
  AttnOut dk{(bf16_t*)dK, lddk, sdk}, dv{(bf16_t*)dV, lddv, sdv};
  if (dma && nsplit == 1) {
    const AttnGrid G = attn_grid(kblocks, BH, 1, 1);
    az_launch(attn_bwd_dkv_dma_kernel, grid1(G), dim3(256), 0, st, G, heads, Tq, Tk, scale, q, k, v, d_o, (const float*)lse, (const float*)delta, dk, dv);
    AZ_CHECK_LAUNCH();
    return AZ_OK;
  }
  const AttnGrid G = attn_grid(kblocks, BH, nsplit, 1);
  az_launch(attn_bwd_dkv_kernel, grid1(G), dim3(256), 0, st, G, heads, Tq, Tk, scale, q, k, v, d_o,
                     (const float*)lse, (const float*)delta, dk, dv, tps, (float*)workspace);
  AZ_CHECK_LAUNCH();
  if (nsplit > 1) {
    long nred = (long)BH * Tk * 128;
    int gr = (int)((nred + 255) / 256); if (gr > 2048) gr = 2048;
    az_launch(attn_dkv_reduce_kernel, dim3(gr), dim3(256), 0, st, heads, Tk, kblocks * 128, nsplit, BH, (const float*)workspace, dk, dv);
    AZ_CHECK_LAUNCH();
  }
  return AZ_OK;
}

}  // extern "C"
