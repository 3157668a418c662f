// bf16 MFMA GEMM / implicit-GEMM convolution core for gfx950 (CDNA4).
//
// One templated kernel serves every dense contraction of the SDXL UNet training step
// (reference: diffusers UNet2DConditionModel called at train.py:2760-2761 and its autograd
// backward at train.py:2765):
//
//   D[M,N] = epilogue( sum_k  Aop[m][k] * Bop[k][n] )
//
//   A operand modes                                         B operand modes
//   A_ROW   A[m*lda + k]            (linear fwd / dgrad)    B_NT     B[n*ldb + k]   (W[out][in])
//   A_COL   A[k*lda + m]            (wgrad: dY^T)           B_NN     B[k*ldb + n]   (dgrad: W, wgrad: X)
//   A_CONV  im2col gather of X      (conv fwd)              B_CONVDG W[co][tap][ci] as [(tap,co)][ci]
//   A_CONVT transposed-conv gather  (conv dgrad, of dY)     B_CONVWG im2col gather of X as [pixel][(tap,ci)]
//
// Layout: activations are NHWC (B,H,W,C) == row-major [pixels][C]; conv weights [Cout][ky][kx][Cin].
// Tiles (BM x BN x 64, 8 or 16 waves): 128x128 (8 waves of 32x64, 2 workgroups / CU: every weight gradient),
// 128x160 (8 waves of 32x80; 2 stages, or 3 stages in the exclusive forward pass), 256x256 (16 waves of 64x64, 1 workgroup / CU);
// v_mfma_f32_16x16x32_bf16 with the operands swapped (D^T = W-frag x A-frag) so each lane owns 4 consecutive output columns.
// Operand tiles reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds, issued from inline asm -- see dma16s: no VGPR staging, no
// ds_write; masked chunks use an out-of-range offset and the hardware writes zeros).
// LDS images are unpadded and lane-linear per 1-KiB DMA piece; bank conflicts are removed by an XOR
// swizzle applied to the per-lane SOURCE address and to the fragment reads (guide rule 21):
//   k-contiguous operands  [rows][64 k]  (128-B rows): 16-B chunk c of row r lives at c ^ ((r>>1)&7)
//   m/n-contiguous operands [64 k][128 x]    (256-B rows): 32-B unit u of row k lives at u ^ (4*((k>>3)&1) + (k&3)),
//     read with the hardware transpose read ds_read_b64_tr_b16.
// One barrier per 64-deep k-tile; the loop waits for its own DMA (vmcnt) right before it.  Optional split-K
// writes fp32 slabs reduced in a fixed order by splitk_reduce_vec_kernel.
#include "az_common.h"
#include "aozora_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 64;
constexpr int PITCH_K = BK * 2;         // 128 B  : [row][k] image
constexpr int PITCH_X = 128 * 2;        // 256 B  : [k][x]  image (per 128-wide sub-image)
// Tile = BM x BN x 64 computed by (BM/64) x (BN/64) waves of 64x64 each: 128x128 (4 waves, 2 workgroups/CU),
// 256x128 / 128x256 (8 waves) and 256x256 (16 waves, 1 workgroup/CU, half the L2->LDS bytes per FLOP).
// An operand tile of R rows is R*128 bytes = R/8 DMA pieces; x-major tiles are split in 128-wide sub-images.

enum { A_ROW = 0, A_COL = 1, A_CONV = 2, A_CONVT = 3 };
enum { B_NT = 0, B_NN = 1, B_CONVDG = 2, B_CONVWG = 3 };

struct Geom {          // convolution geometry (all modes that gather)
  int Hin, Win, Cin;   // conv input grid / channels (X)
  int Hout, Wout, Cout;// conv output grid / channels (Y)
  int stride, pad, ks; // ks in {1,3}
  int cpad;            // channel count used to split k into (tap, c) for CONVT/CONVDG (>= real count)
  int ups;             // 1: X is stored at HALF resolution (Hin/2 x Win/2) and read through a nearest-neighbour 2x gather
                       //    (diffusers Upsample2D: F.interpolate(scale_factor=2, mode="nearest") -> conv)
};

struct Params {
  const bf16_t* A; const bf16_t* B;
  long lda, ldb;
  int lda2, ldb2;             // byte strides (checked < 2^31 on the host)
  int M, N, K;
  bf16_t* C; long ldc;
  float* ws;                 // split-K slabs [ksplit][M][N]
  const bf16_t* bias;        // [N]
  const bf16_t* rowbias;     // [M / rows_per_seg][ld_rb]
  int rows_per_seg; long ld_rb;
  const bf16_t* R; long ldr; // residual [M][N]
  int accumulate;
  int vec_epi;               // 16-byte coalesced epilogue allowed (N % 8 == 0, C / R / slab rows 16-byte aligned)
  int ksplit, ktiles_per_split;
  int wgrad_c;                // 1: C is a weight gradient (A transposed): the reduce streams it past the caches
  int xsplit;                 // 1: the k-splits are dealt to the XCDs (1-D grid, see gemm_kernel): split z runs on XCD(s) z % 8
  int tiles_m, tiles_n;
  int bm, bn, nwaves, stages, kb;      // kb: device k-tile depth (64; 32 for the deep-ring variants of the k-contiguous products)
  int use8;                   // 256-row tile worked by the 8-wave ping-pong kernel (az_gemm8.inc): bn = 256 or 320
  int ablate;                 // diagnostic (option GEMM_ABLATE, timing only -- results are wrong): 1 = no fragment reads / MFMAs, 2 = no operand DMA after the first k-tile
  // implicit-GEMM address arithmetic without per-lane integer division:
  int k_full;                 // K % 64 == 0 (every SDXL linear): the k-tile offset rides in the DMA's scalar-offset operand
  int tap_uniform;            // channel count of the k = (tap, channel) split is a multiple of BK: a k-tile lies in ONE tap
  int conv_fast_a, conv_fast_b;   // fast gather forms of the A (forward / stride-1 data gradient) and B (stride-1 weight gradient) operands
  unsigned magic_w, magic_h;      // floor(2^32 / W) + 1, floor(2^32 / H) + 1: division by multiply-high in the weight-gradient gather
  // fused column sums of the transposed A operand (A_COL products = weight gradients): sum_k A[k][m] per k-segment
  // of cs_rps elements -> cs_ws[(z * cs_nseg + seg) * M + m]; finished by colsum_finish_kernel (bias / time-emb grads)
  float* cs_ws; int cs_rps, cs_nseg;
  bf16_t* cs_seg_out; bf16_t* cs_bias; int cs_n_real;   // targets of the column-sum finish (per-sample sums, bias gradient)
  // In-kernel finish: one arrival counter per output tile (zero at launch, left zero).  The workgroup that arrives LAST at a
  // tile's counter sums the tile's split-K slabs in ascending split order and finishes the tile's column sums, so no reduce /
  // finish kernel follows the product.  nullptr: the separate finish launches are used.
  unsigned* tickets;
  // Grouped launch (az_gemm_nt_grouped_bf16): many products that share the A operand in ONE grid; tile column tn belongs to the
  // group whose [tile_start, next tile_start) range holds it, and that group supplies B, C, bias, N and the leading dimensions.
  const struct GemmGroup* groups; int ngroups;
  // Grouped weight-gradient launch (az_gemm_tn_grouped_bf16, kernel MODE 3): many independent products dW_g += dY_g^T X_g in ONE
  // grid, each over its WHOLE k-range (no split-K slabs, no reduce launch); tiles_m holds the total tile count.
  const struct TnGroup* tgroups;
  // Fused GEGLU forward (az_gemm_geglu_fwd_bf16, kernel MODE 2): B is the [2H][K] weight of ff.net.0.proj; a BM x BN tile holds BN/2
  // value columns n and the BN/2 gate columns H + n of the same n, arranged so that one lane owns value and gate of the same
  // element; the epilogue stores proj[M][2H] (C, needed by the backward pass) and out[M][H] = value * gelu(gate) (gg_y).
  int gg_H; bf16_t* gg_y; long gg_ldy;
  Geom g;
};

struct GemmGroup { const bf16_t* W; bf16_t* C; const bf16_t* bias; long N, ldb, ldc, tile_start; };
// sixteen int64 per record (built by the caller in device memory): tile ids [tile_start, tile_start + tiles_m * tiles_n) are this product's
struct TnGroup { const bf16_t* dY; const bf16_t* X; bf16_t* dW; bf16_t* bias; long M, N, K, lda, ldb, ldc, tile_start, tiles_m, tiles_n, n_real, vec_epi, pad; };

// bias[m] += sum_seg s(seg, m) ; seg_out[seg][m] = bf16(s(seg, m)) with s = sum over the splits z whose k-range meets the
// segment, in ascending z (fixed order: bitwise reproducible)
struct ColsumFinish { int S, nseg, M, kps, ktiles, rps; const float* ws; bf16_t* seg_out; bf16_t* bias; int n_real; };

__device__ __forceinline__ void colsum_finish(const ColsumFinish& c, int m) {
  if (m >= c.M) return;
  float tot = 0.f;
  for (int seg = 0; seg < c.nseg; ++seg) {
    const long lo = (long)seg * c.rps, hi = lo + c.rps;
    float a = 0.f;
    for (int z = 0; z < c.S; ++z) {
      const long k0 = (long)z * c.kps * 64;
      long k1 = (long)(z + 1) * c.kps; if (k1 > c.ktiles) k1 = c.ktiles; k1 *= 64;
      if (k0 < hi && k1 > lo) a += c.ws[((long)z * c.nseg + seg) * c.M + m];
    }
    if (c.seg_out) c.seg_out[(long)seg * c.M + m] = f2bf(a);
    tot += a;
  }
  if (c.bias && m < c.n_real) c.bias[m] = f2bf(bf2f(c.bias[m]) + tot);
}

// Weight gradients are written once per micro-step and read again only by the NEXT micro-step's accumulation (115 ms later): their
// 16-byte read-modify-write goes past the caches' retention (nontemporal), so that 2 x 5 GB per micro-step of it do not push the
// chain's activations out of the L2s and the Infinity Cache.
typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream16(const void* p) {
  const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream16(void* p, const uint4& u) {
  const u32x4_nt v = {u.x, u.y, u.z, u.w};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4_nt*>(p));
}

// Operand fetch: LDS-DMA with 32-bit byte offsets.  A chunk that is masked out (row/col/k tail, conv zero
// padding, stride-2 parity) gets the offset OOB; the hardware range check then writes zeros.
constexpr unsigned OOB = 0x80000000u;          // >= num_records (0x7FFFFFFF): every real offset is below it
typedef __attribute__((address_space(3))) void lds_void;

// k-major image geometry for a KB-deep tile: rows of KB*2 bytes; a 1-KiB DMA piece covers KRPP rows of KCPR 16-byte chunks.
// XOR swizzle of the chunk position (applied to the DMA source address and to the fragment reads, guide rule 21), chosen so
// that every ds_read_b128 lane group -- {0-3, 12-15} of one k-chunk plus {4-11} of the next -- lands on 16 distinct 16-byte
// slots of the 256-byte bank row:  KB = 64 (two rows per bank row): c ^ ((r >> 1) & 7);  KB = 32 (four rows per bank row):
// c ^ h((r >> 2) & 3) with h = {0, 3, 2, 1}.
template <int KB> __device__ __forceinline__ int kswz(int r) {
  if constexpr (KB == 64) return (r >> 1) & 7;
  else return (4 - ((r >> 2) & 3)) & 3;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
// The same descriptor as four scalar dwords (base, stride 0, num_records 0x7FFFFFFF, raw-buffer flags) for the inline-asm DMA.
__device__ __forceinline__ u32x4 make_rsrc_words(const void* p) {
  const unsigned long a = (unsigned long)p;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
}
// LDS byte address of a pointer into the dynamic shared array
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(unsigned long)(lds_void*)const_cast<char*>(p); }

// One 1-KiB piece: lane l's 16 bytes land at dst + 16*l.  The LDS-DMA is issued from INLINE ASM on purpose: hipcc counts a
// __builtin_amdgcn_raw_ptr_buffer_load_lds as a pending LDS write and drains it (s_waitcnt vmcnt(0)) in front of the next
// ds_read_b64_tr_b16 it cannot disambiguate -- in the weight-gradient kernels that put a full DMA round trip in the MIDDLE of
// every k-iteration (between the two 32-deep halves).  An asm statement is invisible to that bookkeeping
// (cdna_hip_programming.md 5.7 item 1): the loops below wait for their DMA themselves (wait_dma / an immediate vmcnt) right before
// the barrier that publishes a tile.  M0 (the LDS destination base) is written in the same statement that uses it.  M0 is NOT on the
// clobber list: it is a register hipcc RESERVES (it never keeps a value live in it across statements -- each of its own users, the
// LDS-DMA builtins, s_movrel indexing, GWS, s_sendmsg, is preceded by its own s_mov m0), and naming a reserved register there only
// draws -Winline-asm "clobber list contains reserved registers: m0 ... may lead to undefined behaviour" (round 5: built both ways,
// same step time, profiles/r05_attn_policy_step_ab.txt).
__device__ __forceinline__ void dma16s(u32x4 r, unsigned voff, unsigned soff, unsigned dst_wave_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(dst_wave_uniform), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// (the wave-uniform byte offset rides in the scalar-offset operand: per-lane offsets that do not depend on the k-tile stay
// untouched in their VGPR and the k advance costs no vector instruction; the range check is on the vector offset, so an OOB
// lane stays masked)
__device__ __forceinline__ void dma16(u32x4 r, unsigned off, unsigned dst_wave_uniform) { dma16s(r, off, 0u, dst_wave_uniform); }
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Work split: an operand tile of R rows is R/8 pieces of 1 KiB; wave w of NW issues pieces q = NP*w + j, j < NP = ceil(R/8/NW)
// (pieces q >= R/8 do not exist and are skipped: only the 160-row tile on 8 waves has such a remainder).
//   k-major image: piece q = rows 8q..8q+7 ; lane l -> row 8q + (l>>3), chunk position l&7
//   x-major image: sub-image q>>4 (128 columns), piece q&15 = k-rows 4(q&15)..+3 ; lane l -> k-row + (l>>4), chunk position l&15

// Convolution gathers, fast form (Params::conv_fast_a / conv_fast_b; every 3x3 / 1x1 convolution of the UNet except conv_in and
// the nearest-2x / stride-2 special cases): the k-tile lies in ONE filter tap (channel count % 64 == 0), so the tap is a scalar
// that the loader advances by itself from call to call (issue() is called for consecutive k-tiles), and the source address of
// a row is  pixel offset (per lane, fixed)  +  tap displacement (scalar)  -- one vector add and one mask test per DMA piece
// instead of the ~50 vector instructions (integer divisions included) of the general gather, which had made the convolution
// kernels issue 6..9 vector instructions per MFMA (profiles/r02_a_pmc_sq.json).  The zero padding is a 9-bit per-row validity mask.
__device__ __forceinline__ int tap_row(int tap, int ks) { return ks == 3 ? (tap * 11) >> 5 : 0; }     // tap / 3 for tap < 9

template <int AMODE, int R, int NW, int KB = 64>
struct ALoader {
  static constexpr int PPS = KB / 4;                       // x-major: 1-KiB pieces (4 k-rows) per 128-wide sub-image
  static constexpr int KCPR = KB / 8, KRPP = 64 / KCPR;    // k-major: 16-byte chunks per row, rows per 1-KiB piece
  static constexpr int NPIECE = (AMODE == A_COL) ? (R / 128) * PPS : R / KRPP;
  static constexpr int NP = (NPIECE + NW - 1) / NW;
  static constexpr bool EXACT = (NP * NW == NPIECE);
  u32x4 rs;
  unsigned base[NP];                 // per piece j
  int kc[NP];                        // k-major: logical k-chunk (0..7) this lane fetches for piece j
  int pix_b[NP], pix_y[NP], pix_x[NP];   // general gather: the row's pixel; fast gather: pix_b = tap validity mask
  int tap_s, c0_s;                   // fast gather: filter tap and first channel of the k-tile the next issue() fetches (wave-uniform)
  __device__ __forceinline__ void init(const Params& p, int m0, int t, int kbeg) {
    rs = make_rsrc_words(p.A);
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;   // wave id as a scalar: LDS destinations and piece indices stay on the SALU
    tap_s = 0; c0_s = 0;
    if constexpr (AMODE == A_COL) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int q = NP * w + j;
        const int krow = 4 * (q % PPS) + (l >> 4);
        const int sw = 4 * ((krow >> 3) & 1) + (krow & 3);
        const int xc = ((((l & 15) >> 1) ^ sw) << 1) | (l & 1);
        const int m = m0 + (q / PPS) * 128 + xc * 8;
        // K % 64 == 0: the lane's k-row offset is folded in here and the k-tile offset rides in the DMA's scalar operand
        base[j] = (m < p.M && (EXACT || q < NPIECE)) ? (unsigned)m * 2u + (p.k_full ? (unsigned)krow * (unsigned)p.lda2 : 0u) : OOB;
      }
    } else {
      const int Hr = (AMODE == A_CONV) ? p.g.Hout : p.g.Hin;
      const int Wr = (AMODE == A_CONV) ? p.g.Wout : p.g.Win;
      if constexpr (AMODE != A_ROW) {
        if (p.conv_fast_a) {
          const int C = (AMODE == A_CONV) ? p.g.Cin : p.g.cpad;
          tap_s = kbeg / C; c0_s = kbeg - tap_s * C;
        }
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int r = KRPP * (NP * w + j) + l / KCPR;
        kc[j] = (l % KCPR) ^ kswz<KB>(r);
        const int m = m0 + r;
        const bool ok = m < p.M && (EXACT || r < R);
        if constexpr (AMODE == A_ROW) {
          base[j] = ok ? (unsigned)m * (unsigned)p.lda2 + (unsigned)kc[j] * 16u : OOB;
        } else {
          const int mm = ok ? m : 0;
          const int b = mm / (Hr * Wr);
          const int rem = mm - b * (Hr * Wr);
          const int y = rem / Wr, x = rem - (rem / Wr) * Wr;
          pix_b[j] = b; pix_y[j] = ok ? y : -100000; pix_x[j] = x;
          if (p.conv_fast_a) {
            // source grid and the row's centre pixel in it: forward reads X (Hin, Win) around (y*stride, x*stride);
            // the stride-1 data gradient reads dY (Hout, Wout) around (y, x)
            const int Hs = (AMODE == A_CONV) ? p.g.Hin : p.g.Hout, Ws = (AMODE == A_CONV) ? p.g.Win : p.g.Wout;
            const int cy = (AMODE == A_CONV) ? y * p.g.stride : y, cx = (AMODE == A_CONV) ? x * p.g.stride : x;
            base[j] = (unsigned)((b * Hs + cy) * Ws + cx) * (unsigned)p.lda2 + (unsigned)kc[j] * 16u;
            int mask = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
              if (tap < p.g.ks * p.g.ks) {
                const int ky = tap_row(tap, p.g.ks), kx = tap - 3 * ky;
                const int sy = (AMODE == A_CONV) ? cy + ky - p.g.pad : cy + p.g.pad - ky;
                const int sx = (AMODE == A_CONV) ? cx + kx - p.g.pad : cx + p.g.pad - kx;
                if (ok && sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) mask |= 1 << tap;
              }
            }
            pix_b[j] = mask;
          }
        }
      }
    }
  }
  // pad_dst != 0 (rings of >= 3 stages): a wave whose share of the tile has fewer than NP pieces issues the missing ones out of
  // range into a 1-KiB scratch row, so that EVERY wave has exactly NP DMAs per k-tile in flight and the loop's counted wait
  // is an immediate (a wave-dependent count had to go through a 17-way branch tree in front of every barrier)
  __device__ __forceinline__ void issue_pad(unsigned pad_dst) const {
#pragma unroll
    for (int j = 0; j < NP; ++j) dma16(rs, OOB, pad_dst);
  }
  __device__ __forceinline__ void issue(const Params& p, int k0, int t, unsigned img, unsigned pad_dst) {
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;   // wave id as a scalar: LDS destinations and piece indices stay on the SALU
    if constexpr (AMODE == A_ROW || AMODE == A_COL) {
      if (p.k_full) {      // K % 64 == 0: no k-tail to mask; ONE uniform branch per tile instead of one per piece
        const unsigned so = (AMODE == A_ROW) ? (unsigned)k0 * 2u : (unsigned)k0 * (unsigned)p.lda2;
#pragma unroll
        for (int j = 0; j < NP; ++j)      // a piece this wave does not have: base[j] is out of range (init), its zeros go to the scratch row
          dma16s(rs, base[j], so, (EXACT || NP * w + j < NPIECE) ? img + (unsigned)(NP * w + j) * 1024u : pad_dst);
        return;
      }
    }
    if constexpr (AMODE == A_CONV || AMODE == A_CONVT) {
      if (p.conv_fast_a) {
        const int C = (AMODE == A_CONV) ? p.g.Cin : p.g.cpad;
        const int Ws = (AMODE == A_CONV) ? p.g.Win : p.g.Wout;
        const int ky = tap_row(tap_s, p.g.ks), kx = tap_s - 3 * ky;
        const int dpix = (AMODE == A_CONV) ? (ky - p.g.pad) * Ws + (kx - p.g.pad) : (p.g.pad - ky) * Ws + (p.g.pad - kx);
        const unsigned delta = (unsigned)(dpix * p.lda2 + c0_s * 2);         // wave-uniform, may be "negative" (wraps)
        const int bit = 1 << tap_s;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (!EXACT && NP * w + j >= NPIECE) { dma16(rs, OOB, pad_dst); continue; }
          dma16(rs, (pix_b[j] & bit) ? base[j] + delta : OOB, img + (unsigned)(NP * w + j) * 1024u);
        }
        c0_s += KB;
        if (c0_s >= C) { c0_s -= C; ++tap_s; }
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (!EXACT && NP * w + j >= NPIECE) { dma16(rs, OOB, pad_dst); continue; }
      const unsigned dst = img + (unsigned)(NP * w + j) * 1024u;
      unsigned off;
      if constexpr (AMODE == A_ROW) {
        off = (k0 + kc[j] * 8) < p.K ? base[j] + (unsigned)k0 * 2u : OOB;
      } else if constexpr (AMODE == A_COL) {
        const int k = k0 + 4 * ((NP * w + j) % PPS) + (l >> 4);
        off = k < p.K ? (unsigned)k * (unsigned)p.lda2 + base[j] : OOB;
      } else if constexpr (AMODE == A_CONV) {
        const int k = k0 + kc[j] * 8;
        int tap, ci;
        if (p.tap_uniform) { tap = k0 / p.g.Cin; ci = k - tap * p.g.Cin; }      // k0 is wave-uniform: scalar division, once
        else { tap = k / p.g.Cin; ci = k - tap * p.g.Cin; }
        const int ky = (p.g.ks == 3) ? tap / 3 : 0, kx = (p.g.ks == 3) ? tap - 3 * ky : 0;
        const int iy = pix_y[j] * p.g.stride + ky - p.g.pad, ix = pix_x[j] * p.g.stride + kx - p.g.pad;
        const bool ok = k < p.K && iy >= 0 && iy < p.g.Hin && ix >= 0 && ix < p.g.Win;
        const int u = p.g.ups;     // nearest-2x: source pixel (iy >> 1, ix >> 1) of the half-resolution tensor
        off = ok ? (unsigned)((pix_b[j] * (p.g.Hin >> u) + (iy >> u)) * (p.g.Win >> u) + (ix >> u)) * (unsigned)p.lda2 + (unsigned)ci * 2u : OOB;
      } else {  // A_CONVT : rows = conv-input pixels, source = dY (Hout,Wout,cpad)
        const int k = k0 + kc[j] * 8;
        const int tap = p.tap_uniform ? k0 / p.g.cpad : k / p.g.cpad;
        const int co = k - tap * p.g.cpad;
        const int ky = tap / 3, kx = tap - 3 * ky;
        int ty = pix_y[j] + p.g.pad - ky, tx = pix_x[j] + p.g.pad - kx;
        bool ok = k < p.K && ty >= 0 && tx >= 0;
        if (p.g.stride == 2) { ok = ok && !(ty & 1) && !(tx & 1); ty >>= 1; tx >>= 1; }
        ok = ok && ty < p.g.Hout && tx < p.g.Wout;
        off = ok ? (unsigned)((pix_b[j] * p.g.Hout + ty) * p.g.Wout + tx) * (unsigned)p.lda2 + (unsigned)co * 2u : OOB;
      }
      dma16(rs, off, dst);
    }
  }
};

template <int BMODE, int R, int NW, int KB = 64>
struct BLoader {
  static constexpr int PPS = KB / 4;
  static constexpr int KCPR = KB / 8, KRPP = 64 / KCPR;
  static constexpr int NPIECE = (BMODE != B_NT) ? (R / 128) * PPS : R / KRPP;
  static constexpr int NP = (NPIECE + NW - 1) / NW;
  static constexpr bool EXACT = (NP * NW == NPIECE);
  u32x4 rs;
  unsigned base[NP];
  int kc[NP];
  int tap_ky[NP], tap_kx[NP], ci[NP]; bool n_ok[NP];   // CONVWG per-piece n-chunk state (fast form: tap_ky / tap_kx hold ky - pad / kx - pad)
  template <bool GGF = false>
  __device__ __forceinline__ void init(const Params& p, int n0, int t) {
    rs = make_rsrc_words(p.B);
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;   // wave id as a scalar: LDS destinations and piece indices stay on the SALU
    if constexpr (BMODE == B_NT) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int r = KRPP * (NP * w + j) + l / KCPR;
        kc[j] = (l % KCPR) ^ kswz<KB>(r);
        if constexpr (GGF) {      // tile rows [0, R/2): value rows n0 + r of the weight; [R/2, R): gate rows H + n0 + (r - R/2)
          const int rr = r < R / 2 ? r : r - R / 2;
          const int n = n0 + rr + (r < R / 2 ? 0 : p.gg_H);
          base[j] = ((n0 + rr) < p.gg_H && (EXACT || r < R)) ? (unsigned)n * (unsigned)p.ldb2 + (unsigned)kc[j] * 16u : OOB;
        } else {
          const int n = n0 + r;
          base[j] = (n < p.N && (EXACT || r < R)) ? (unsigned)n * (unsigned)p.ldb2 + (unsigned)kc[j] * 16u : OOB;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int q = NP * w + j;
        const int krow = 4 * (q % PPS) + (l >> 4);
        const int sw = 4 * ((krow >> 3) & 1) + (krow & 3);
        const int xc = ((((l & 15) >> 1) ^ sw) << 1) | (l & 1);
        const int n = n0 + (q / PPS) * 128 + xc * 8;
        if constexpr (BMODE == B_CONVWG) {
          n_ok[j] = n < p.N;
          const int nn = n_ok[j] ? n : 0;
          const int tap = nn / p.g.Cin;
          ci[j] = nn - tap * p.g.Cin;
          tap_ky[j] = (p.g.ks == 3) ? tap / 3 : 0;
          tap_kx[j] = (p.g.ks == 3) ? tap - 3 * tap_ky[j] : 0;
          if (p.conv_fast_b) {       // stride 1, same-size grids: source pixel index = output pixel index + a per-lane tap displacement
            tap_ky[j] -= p.g.pad; tap_kx[j] -= p.g.pad;
            kc[j] = tap_ky[j] * p.g.Win + tap_kx[j];
            base[j] = (unsigned)ci[j] * 2u;
          }
        } else if constexpr (BMODE == B_NN) {
          base[j] = (n < p.N && (EXACT || q < NPIECE)) ? (unsigned)n * 2u + (p.k_full ? (unsigned)krow * (unsigned)p.ldb2 : 0u) : OOB;
        } else {
          base[j] = n < p.N ? (unsigned)n * 2u : OOB;
        }
      }
    }
  }
  __device__ __forceinline__ void issue_pad(unsigned pad_dst) const {
#pragma unroll
    for (int j = 0; j < NP; ++j) dma16(rs, OOB, pad_dst);
  }
  __device__ __forceinline__ void issue(const Params& p, int k0, int t, unsigned img, unsigned pad_dst) const {
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;   // wave id as a scalar: LDS destinations and piece indices stay on the SALU
    if constexpr (BMODE == B_NT || BMODE == B_NN) {
      if (p.k_full) {      // one uniform branch per tile (see ALoader::issue)
        const unsigned so = (BMODE == B_NT) ? (unsigned)k0 * 2u : (unsigned)k0 * (unsigned)p.ldb2;
#pragma unroll
        for (int j = 0; j < NP; ++j)      // a piece this wave does not have: base[j] is out of range (init), its zeros go to the scratch row
          dma16s(rs, base[j], so, (EXACT || NP * w + j < NPIECE) ? img + (unsigned)(NP * w + j) * 1024u : pad_dst);
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (!EXACT && NP * w + j >= NPIECE) { dma16(rs, OOB, pad_dst); continue; }
      const unsigned dst = img + (unsigned)(NP * w + j) * 1024u;
      unsigned off;
      if constexpr (BMODE == B_NT) {
        off = (k0 + kc[j] * 8) < p.K ? base[j] + (unsigned)k0 * 2u : OOB;
      } else if constexpr (BMODE == B_NN) {
        const int k = k0 + 4 * ((NP * w + j) % PPS) + (l >> 4);
        off = k < p.K ? (unsigned)k * (unsigned)p.ldb2 + base[j] : OOB;
      } else if constexpr (BMODE == B_CONVDG) {  // k = (tap, co) with co < cpad ; n = ci
        const int k = k0 + 4 * ((NP * w + j) % PPS) + (l >> 4);
        const int tap = p.tap_uniform ? k0 / p.g.cpad : k / p.g.cpad;
        const int co = k - tap * p.g.cpad;
        const bool ok = k < p.K && co < p.g.Cout;
        off = ok ? (unsigned)(co * 9 + tap) * (unsigned)(p.g.Cin * 2) + base[j] : OOB;
      } else if (p.conv_fast_b) {  // B_CONVWG, stride 1: output pixel m -> (row, column) by multiply-high (exact: host checks m * W < 2^32)
        const int m = k0 + 4 * ((NP * w + j) % PPS) + (l >> 4);
        const unsigned q1 = __umulhi((unsigned)m, p.magic_w);               // m / W  (rows of all samples)
        const int ox = m - (int)q1 * p.g.Win;
        const unsigned q2 = __umulhi(q1, p.magic_h);                        // sample index
        const int oy = (int)q1 - (int)q2 * p.g.Hin;
        const bool ok = n_ok[j] && m < p.K && (unsigned)(oy + tap_ky[j]) < (unsigned)p.g.Hin && (unsigned)(ox + tap_kx[j]) < (unsigned)p.g.Win;
        off = ok ? (unsigned)(m + kc[j]) * (unsigned)p.ldb2 + base[j] : OOB;
      } else {  // B_CONVWG : k = output pixel, n = (tap, ci)
        const int hw = p.g.Hout * p.g.Wout;
        const int m = k0 + 4 * ((NP * w + j) % PPS) + (l >> 4);
        bool ok = n_ok[j] && m < p.K;
        const int mm = ok ? m : 0;
        const int b = mm / hw; const int rem = mm - b * hw;
        const int oy = rem / p.g.Wout, ox = rem - oy * p.g.Wout;
        const int iy = oy * p.g.stride + tap_ky[j] - p.g.pad, ix = ox * p.g.stride + tap_kx[j] - p.g.pad;
        ok = ok && iy >= 0 && iy < p.g.Hin && ix >= 0 && ix < p.g.Win;
        const int u = p.g.ups;
        off = ok ? (unsigned)((b * (p.g.Hin >> u) + (iy >> u)) * (p.g.Win >> u) + (ix >> u)) * (unsigned)p.ldb2 + (unsigned)ci[j] * 2u : OOB;
      }
      dma16(rs, off, dst);
    }
  }
};

// fragment for rows [rowbase, rowbase+16) and k-step kk (32 deep) of the tile
template <bool XMAJOR, int KB = 64>
__device__ __forceinline__ bf16x8 read_frag(const char* img, int rowbase, int kk, int lane) {
  if constexpr (!XMAJOR) {
    const int r = rowbase + (lane & 15);
    const int c = (kk * 4 + (lane >> 4)) ^ kswz<KB>(r);
    return *reinterpret_cast<const bf16x8*>(img + r * (KB * 2) + c * 16);
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int krow = kk * 32 + 8 * g + (i >> 2);            // second read: krow + 4 (same swizzle value)
    const int sw = 4 * (g & 1) + (i >> 2);                  // = 4*((krow>>3)&1) + (krow&3)
    const char* base = img + (rowbase >> 7) * (KB * PITCH_X) + krow * PITCH_X + ((((rowbase & 127) >> 4) ^ sw) << 5) + (i & 3) * 8;
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base + 4 * PITCH_X));
    bf16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
  }
}

// Anatomy builds only (tools/build_exp.sh anatomy -DAZ_ANATOMY; the product library carries no stamp): thread 0 of every workgroup
// leaves s_memtime stamps in a buffer of its own (the otherwise unused split-K workspace of an unsplit product):
//   [0] kernel entry  [1] operand addresses ready  [2] first k-tile's DMA issued  [3] first k-tile landed + published
//   [4] last MFMA issued (loop left)  [5] epilogue stores issued  [6] ... and acknowledged  [8] / [9] s_memrealtime (100 MHz) at entry / exit
#ifdef AZ_ANATOMY
#define AZ_STAMP(i) do { if (threadIdx.x == 0 && p.ws && p.ksplit == 1) ((unsigned long*)p.ws)[(long)blockIdx.x * 16 + (i)] = (i) >= 8 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AZ_STAMP(i) do { } while (0)
#endif

// vmcnt(n) for a compile-time n that reaches here through a function argument (az_gemm8.inc; the switch folds away)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

#include "az_gemm8.inc"

template <int AMODE, int BMODE, int BM, int BN, int NWM, int NWN, int NS = 2, int KB = 64, int MODE = 0>      // MODE 1: grouped launch, 2: fused GEGLU forward epilogue
__global__ __launch_bounds__(NWM * NWN * 64) void gemm_kernel(const Params pin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool GROUPED = (MODE == 1), GGF = (MODE == 2), GTN = (MODE == 3);
  constexpr bool AX = (AMODE == A_COL);
  constexpr bool WGRAD_C = (AMODE == A_COL);      // every product with a transposed A is a weight gradient: C streams past the caches
  constexpr bool BX = (BMODE != B_NT);
  constexpr int NW = NWM * NWN;
  constexpr int WM = BM / NWM, WN = BN / NWN;        // wave tile: 64x64 (standard), 64x80 / 32x80 for the 128x160 tile
  constexpr int MI = WM / 16, NJ = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0 && (BN % 64 == 0 || BMODE == B_NT), "wave tile");
  constexpr int A_BYTES = BM * KB * 2, STAGE = (BM + BN) * KB * 2;
  static_assert(KB == 64 || KB == 32, "k-tile depth");
  static_assert(NS >= 2 && NS <= 6, "ring depth");
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / NWN, wn = wave - wm * NWN;

  // XCD-aware bijective remap of the linear tile id (guide T1): blocks b, b+8, ... share an XCD.
  const int nwg = pin.tiles_m * pin.tiles_n;
  int id = blockIdx.x, z = blockIdx.y;
  if (pin.xsplit) {
    // Split-K weight gradients: the SPLITS are dealt to the XCDs.  Blocks b, b + 8, ... share an XCD, so the tiles that sum over one
    // k-range run side by side behind one L2 and that k-range of dY and X crosses the fabric about 8 / ksplit (>= 1) times -- dealt
    // tile-wise (every XCD a patch of tiles of EVERY split) each L2 fetches the panels of its patch for all of K: 4.8x the
    // algorithmic bytes on 1280x1280x4096, 9.2x on the 128^2 convolution weight gradients (PMC, profiles/r03_d_pmc_*.json).
    // Speed only: any placement computes the same slabs.
    // The (split, tile) pairs in split-major order are cut into 8 contiguous runs, one per XCD (the bijective remap below, over
    // nwg * ksplit items): an XCD's co-resident workgroups are consecutive tiles of ONE split (two at a run boundary).
    const int W = nwg * pin.ksplit, L = blockIdx.x, xcd = L & 7;
    const int q = W >> 3, r = W & 7;
    const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    z = idx / nwg; id = idx - z * nwg;
  } else {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  Params pl;
  int tiles_m = pin.tiles_m, tiles_n = pin.tiles_n;
  if constexpr (GTN) {          // this tile's product: bisection over the products' first tile ids (wave-uniform), then its own raster
    pl = pin;
    int lo = 0, hi = pin.ngroups - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pin.tgroups[mid].tile_start <= (long)id) lo = mid; else hi = mid - 1;
    }
    const TnGroup g = pin.tgroups[lo];
    pl.A = g.dY; pl.B = g.X; pl.C = g.dW; pl.M = (int)g.M; pl.N = (int)g.N; pl.K = (int)g.K;
    pl.lda = g.lda; pl.ldb = g.ldb; pl.ldc = g.ldc; pl.lda2 = (int)(g.lda * 2); pl.ldb2 = (int)(g.ldb * 2);
    pl.k_full = (g.K % BK) == 0; pl.ktiles_per_split = (int)((g.K + BK - 1) / BK); pl.vec_epi = (int)g.vec_epi;
    pl.cs_bias = g.bias; pl.cs_n_real = (int)g.n_real;
    id -= (int)g.tile_start; tiles_m = (int)g.tiles_m; tiles_n = (int)g.tiles_n;
  }
  // grouped rasterisation: consecutive ids (= co-resident workgroups of one XCD after the remap) walk GROUP rows of
  // tiles before moving to the next column, so the ~32-64 tiles sharing an L2 form a near-square patch and both
  // operand panels are re-used from L2 instead of being re-streamed through the fabric.
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int in_grp = id - grp * per_group;
  const int tn = in_grp / gsize;
  const int tm = first_m + (in_grp - tn * gsize);
  const int m0 = tm * BM;
  int n0 = GGF ? tn * (BN / 2) : tn * BN;      // GGF: first of the tile's BN/2 value columns (its gate columns start at H + n0)
  if constexpr (GROUPED) {      // this tile column's product: bisection over the groups' first-tile indices (wave-uniform)
    pl = pin;
    int lo = 0, hi = pin.ngroups - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pin.groups[mid].tile_start <= (long)tn) lo = mid; else hi = mid - 1;
    }
    const GemmGroup g = pin.groups[lo];
    pl.B = g.W; pl.C = g.C; pl.bias = g.bias; pl.N = (int)g.N; pl.ldb = g.ldb; pl.ldb2 = (int)(g.ldb * 2); pl.ldc = g.ldc;
    n0 = (tn - (int)g.tile_start) * BN;
  }
  const Params& p = (GROUPED || GTN) ? pl : pin;
  const int ktiles = (p.K + BK - 1) / BK;
  const int kt_begin = z * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > ktiles) kt_end = ktiles;

  auto imgA = [&](int buf) -> char* { return smem + buf * STAGE; };
  auto imgB = [&](int buf) -> char* { return smem + buf * STAGE + A_BYTES; };
  const unsigned smem_lds = lds_addr(smem);          // LDS byte addresses of the same images, for the DMA destinations
  auto dstA = [&](int buf) -> unsigned { return smem_lds + (unsigned)(buf * STAGE); };
  auto dstB = [&](int buf) -> unsigned { return smem_lds + (unsigned)(buf * STAGE + A_BYTES); };

  // first B-image row of the wave's j-th 16-column sub-tile.  GGF: sub-tiles [0, NJ/2) are value columns, [NJ/2, NJ) the gate columns
  // of the SAME n (B-image rows BN/2 + ...), so accumulators j and j + NJ/2 of a lane are value and gate of one element
  auto brow = [&](int j) -> int {
    if constexpr (GGF) return (j < NJ / 2) ? wn * (WN / 2) + 16 * j : BN / 2 + wn * (WN / 2) + 16 * (j - NJ / 2);
    else return wn * WN + 16 * j;
  };
  AZ_STAMP(0); AZ_STAMP(8);
  ALoader<AMODE, BM, NW, KB> la; la.init(p, m0, t, kt_begin * BK);
  BLoader<BMODE, BN, NW, KB> lb; lb.template init<GGF>(p, n0, t);
  AZ_STAMP(1);

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // column sums of A ride on the matrix pipe: one extra MFMA per A fragment against an all-ones B fragment, issued by
  // the wn == 0 waves of the tn == 0 workgroups only (wave-uniform)
  constexpr bool CS = (AMODE == A_COL);
  // Weight-gradient products (both operands through the transposing read): with the DMA invisible to the compiler's wait
  // bookkeeping (see dma16s) the next tile is issued at the top of the iteration like everywhere else (+5..14 % over issuing it
  // behind the fragment reads), and the linear form keeps the fragments of BOTH 32-deep halves in flight before the first MFMA
  // (+2..5 % more; the convolution form has no registers to spare for that: -30 %).  tools/gemm_ab, same-process A/B.
  constexpr bool READ_ALL = (AMODE == A_COL) && (BMODE == B_NN) && (KB == 64) && (MI * NJ <= 10);      // (for the k-contiguous products: +-0, they wait on their DMA)
  const bool cs_on = CS && (GTN ? p.cs_bias != nullptr : p.cs_ws != nullptr) && tn == 0 && wn == 0;
  f32x4 accs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  // NS == 2, LDS-DMA double buffer: while tile t is multiplied out of buffer t&1 the DMA of tile t+1 fills the
  // other buffer; an explicit vmcnt(0) + __syncthreads() close the iteration.
  // NS == 3, prefetch distance 2 (grids of <= 1 workgroup per CU, where the LDS is otherwise idle): tiles t+1 and t+2
  // are in flight while tile t is multiplied; a counted vmcnt (this wave's DMA pieces of ONE tile may stay outstanding)
  // + a raw barrier open the iteration, which also frees buffer (t+2)%3 = (t-1)%3 for the next DMA.
  // host k-tiles are 64 deep (split-K bookkeeping); the device tile is KB deep
  const int kbeg = kt_begin * BK;
  int kend = kt_end * BK; if (kend > p.K) kend = p.K;
  const int nk = kend > kbeg ? (kend - kbeg + KB - 1) / KB : 0;
  // rings (NS >= 3): every wave keeps exactly PIECES DMA instructions per k-tile in flight (missing pieces and tiles past the last
  // one are issued out of range into the scratch row behind the ring), so the loop's counted wait is the immediate (NS-2)*PIECES
  constexpr int PIECES = decltype(la)::NP + decltype(lb)::NP;
  const unsigned pad_dst = smem_lds + (unsigned)(NS * STAGE);
  if (nk > 0) {
    la.issue(p, kbeg, t, dstA(0), pad_dst);
    lb.issue(p, kbeg, t, dstB(0), pad_dst);
  } else if constexpr (NS >= 3) { la.issue_pad(pad_dst); lb.issue_pad(pad_dst); }
  AZ_STAMP(2);
  if constexpr (NS >= 3) {
    // ring of NS buffers, prefetch distance NS-1: tiles 0 .. NS-2 are in flight before the loop
#pragma unroll
    for (int d = 1; d < NS - 1; ++d) {
      if (nk > d) {
        la.issue(p, kbeg + d * KB, t, dstA(d), pad_dst);
        lb.issue(p, kbeg + d * KB, t, dstB(d), pad_dst);
      } else { la.issue_pad(pad_dst); lb.issue_pad(pad_dst); }
    }
#ifdef AZ_ANATOMY
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * PIECES) : "memory"); AZ_STAMP(3);
#endif
  } else {
    wait_dma();
    __syncthreads();
    AZ_STAMP(3);
  }
  for (int it = 0; it < nk; ++it) {
    int cur;
    if constexpr (NS >= 3) {
      cur = it % NS;
      // tile `it` must have landed: this wave keeps the pieces of the NS-2 younger tiles (real or padding) outstanding; the barrier
      // then (a) publishes every wave's pieces of tile `it` and (b) frees buffer (it-1) % NS = (it+NS-1) % NS for the next DMA
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * PIECES) : "memory");
      asm volatile("s_barrier" ::: "memory");
      if (it + NS - 1 < nk && !(p.ablate & 2)) {
        const int nb = (it + NS - 1) % NS;
        la.issue(p, kbeg + (it + NS - 1) * KB, t, dstA(nb), pad_dst);
        lb.issue(p, kbeg + (it + NS - 1) * KB, t, dstB(nb), pad_dst);
      } else { la.issue_pad(pad_dst); lb.issue_pad(pad_dst); }
    } else {
      cur = it & 1;
      if (it + 1 < nk && !(p.ablate & 2)) {
        la.issue(p, kbeg + (it + 1) * KB, t, dstA(cur ^ 1), pad_dst);
        lb.issue(p, kbeg + (it + 1) * KB, t, dstB(cur ^ 1), pad_dst);
      }
    }
    if constexpr (READ_ALL) {
      if (!(p.ablate & 1)) {
        bf16x8 fa[2][MI], fb[2][NJ];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[kk][i] = read_frag<AX, KB>(imgA(cur), wm * WM + 16 * i, kk, lane);
#pragma unroll
          for (int j = 0; j < NJ; ++j) fb[kk][j] = read_frag<BX, KB>(imgB(cur), brow(j), kk, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
          if constexpr (CS) {
            if (cs_on) {
#pragma unroll
              for (int i = 0; i < MI; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[kk][i], accs[i], 0, 0, 0);
            }
          }
        }
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) {
      if (p.ablate & 1) break;
      bf16x8 fa[MI], fb[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = read_frag<AX, KB>(imgA(cur), wm * WM + 16 * i, kk, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = read_frag<BX, KB>(imgB(cur), brow(j), kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      if constexpr (CS) {
        if (cs_on) {
#pragma unroll
          for (int i = 0; i < MI; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], accs[i], 0, 0, 0);
        }
      }
    }
    }
    if constexpr (CS && GTN) {      // whole k-range in this tile: the column sums are final, this tile is the only writer of bias[m0 ..]
      if (cs_on && it + 1 == nk && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int m = m0 + wm * WM + 16 * i + (lane & 15);
          if (m < p.cs_n_real) p.cs_bias[m] = f2bf(bf2f(p.cs_bias[m]) + accs[i][0]);
        }
      }
    } else if constexpr (CS) {
      if (cs_on) {      // flush at the end of a k-segment (= one sample's pixels) and at the end of this split's range
        const int kb = kbeg + it * KB;
        const int seg = kb / p.cs_rps;
        if (it + 1 == nk || (kb + KB) / p.cs_rps != seg) {
          if ((lane >> 4) == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              const int m = m0 + wm * WM + 16 * i + (lane & 15);
              if (m < p.M) {
                float* slot = p.cs_ws + ((long)z * p.cs_nseg + seg) * p.M + m;
                if (p.tickets) __hip_atomic_store(slot, accs[i][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through: read by the tile's last arriver
                else *slot = accs[i][0];
              }
            }
          }
#pragma unroll
          for (int i = 0; i < MI; ++i) accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if constexpr (NS == 2) { wait_dma(); __syncthreads(); }      // tile it+1 has landed (this wave's pieces; the barrier covers the others')
  }
  if constexpr (NS >= 3) __syncthreads();      // the epilogue re-uses the LDS
  AZ_STAMP(4);

  // ---- epilogue ---------------------------------------------------------------------------------------------
  // MFMA layout: lane owns row m = ..+(lane&15), columns n = ..+4*(lane>>4)+{0..3} of each 16x16 sub-tile.
  const int lm = lane & 15, ln = 4 * (lane >> 4);
  const __amdgpu_buffer_rsrc_t ws_rs = make_rsrc(p.ws);       // slab stores of the in-kernel finish (32-bit offsets: checked on the host)
  if (p.vec_epi) {
    // Coalesced path: the wave's WM x WN fp32 tile goes through its private LDS window in passes of PR rows, so that
    // residual / accumulate loads and the stores are 16-byte lane pieces of full rows instead of 8-byte row-strided
    // accesses.  64-column wave tiles use a 256-byte pitch with XOR-swizzled 16-byte units, the 80-column ones a padded
    // pitch.  (The k-loop ended with a barrier: LDS is free.)
    constexpr bool XORW = (WN == 64);
    constexpr int EP = XORW ? 256 : WN * 4 + 16;
    constexpr int PR = (WM >= 32 && 32 * EP * NW <= NS * STAGE) ? 32 : 16;
    constexpr int NPASS = WM / PR, IPP = PR / 16, CH = WN / 8, ITEMS = PR * CH;
    static_assert(PR * EP * NW <= NS * STAGE, "epilogue window exceeds the LDS allocation");
    char* win = smem + wave * (PR * EP);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
      for (int ii = 0; ii < IPP; ++ii) {
        const int i = IPP * pass + ii;
        const int r = 16 * ii + lm;
        const int m = m0 + wm * WM + 16 * i + lm;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = GGF ? ((j < NJ / 2) ? n0 + wn * (WN / 2) + 16 * j + ln : p.gg_H + n0 + wn * (WN / 2) + 16 * (j - NJ / 2) + ln)
                            : n0 + wn * WN + 16 * j + ln;
          float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          if (p.ksplit == 1) {
            if (p.bias && n < p.N) {
              v.x += bf2f(p.bias[n]); v.y += bf2f(p.bias[n + 1]); v.z += bf2f(p.bias[n + 2]); v.w += bf2f(p.bias[n + 3]);
            }
            if (p.rowbias && n < p.N && m < p.M) {
              const bf16_t* rbp = p.rowbias + (long)(m / p.rows_per_seg) * p.ld_rb + n;
              v.x += bf2f(rbp[0]); v.y += bf2f(rbp[1]); v.z += bf2f(rbp[2]); v.w += bf2f(rbp[3]);
            }
          }
          const int cu = 4 * j + (lane >> 4);
          *reinterpret_cast<float4*>(win + r * EP + ((XORW ? (cu ^ (r & 15)) : cu) << 4)) = v;
        }
      }
      if constexpr (GGF) {       // window columns [0, WN/2): value, [WN/2, WN): gate of the same n; one lane takes an 8-column piece of both
        static_assert(!GGF || XORW, "fused GEGLU forward: 64-column wave tiles");
        constexpr int CHH = CH / 2;
#pragma unroll
        for (int q0 = 0; q0 < PR * CHH; q0 += 64) {
          const int q = q0 + lane;
          if ((PR * CHH) % 64 != 0 && q >= PR * CHH) break;
          const int r = q / CHH, cc = q - r * CHH;
          const int m = m0 + wm * WM + PR * pass + r;
          const int n = n0 + wn * (WN / 2) + cc * 8;
          const float4 vlo = *reinterpret_cast<const float4*>(win + r * EP + (((2 * cc) ^ (r & 15)) << 4));
          const float4 vhi = *reinterpret_cast<const float4*>(win + r * EP + (((2 * cc + 1) ^ (r & 15)) << 4));
          const float4 glo = *reinterpret_cast<const float4*>(win + r * EP + (((2 * (cc + CHH)) ^ (r & 15)) << 4));
          const float4 ghi = *reinterpret_cast<const float4*>(win + r * EP + (((2 * (cc + CHH) + 1) ^ (r & 15)) << 4));
          if (m >= p.M || n >= p.gg_H) continue;
          const float a[8] = {vlo.x, vlo.y, vlo.z, vlo.w, vhi.x, vhi.y, vhi.z, vhi.w};
          const float gt[8] = {glo.x, glo.y, glo.z, glo.w, ghi.x, ghi.y, ghi.z, ghi.w};
          uint4 ua, ug, uy;
          uint32_t* wa = reinterpret_cast<uint32_t*>(&ua);
          uint32_t* wg2 = reinterpret_cast<uint32_t*>(&ug);
          uint32_t* wy = reinterpret_cast<uint32_t*>(&uy);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            wa[e] = pack2bf(a[2 * e], a[2 * e + 1]);
            wg2[e] = pack2bf(gt[2 * e], gt[2 * e + 1]);
            // out = value * gelu(gate) on the bf16-rounded projection: the bits az_geglu_fwd computes from the stored proj
            const float a0 = __uint_as_float(wa[e] << 16), a1 = __uint_as_float(wa[e] & 0xFFFF0000u);
            const float g0 = __uint_as_float(wg2[e] << 16), g1 = __uint_as_float(wg2[e] & 0xFFFF0000u);
            wy[e] = pack2bf(a0 * gelu_erf(g0), a1 * gelu_erf(g1));
          }
          bf16_t* cp = p.C + (long)m * p.ldc + n;
          st_stream16(cp, ua);                 // the projection is read again only by the BACKWARD pass: streamed past the caches
          st_stream16(cp + p.gg_H, ug);
          *reinterpret_cast<uint4*>(p.gg_y + (long)m * p.gg_ldy + n) = uy;
        }
      } else
#pragma unroll
      for (int q0 = 0; q0 < ITEMS; q0 += 64) {
        const int q = q0 + lane;
        if (ITEMS % 64 != 0 && q >= ITEMS) break;
        const int r = q / CH, cc = q - r * CH;
        const int m = m0 + wm * WM + PR * pass + r;
        const int n = n0 + wn * WN + cc * 8;
        const float4 lo = *reinterpret_cast<const float4*>(win + r * EP + ((XORW ? ((2 * cc) ^ (r & 15)) : (2 * cc)) << 4));
        const float4 hi = *reinterpret_cast<const float4*>(win + r * EP + ((XORW ? ((2 * cc + 1) ^ (r & 15)) : (2 * cc + 1)) << 4));
        if (m >= p.M || n >= p.N) continue;
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (p.ksplit > 1) {
          float* dst = p.ws + ((long)z * p.M + m) * p.N + n;
          if (p.tickets) {        // in-kernel finish: WRITE-THROUGH (sc1) slab stores -- another workgroup reads them in this launch
            const unsigned off = (unsigned)(((long)z * p.M + m) * p.N + n) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), ws_rs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), ws_rs, off + 16u, 0, 16);
          } else {
            *reinterpret_cast<float4*>(dst) = lo;
            *reinterpret_cast<float4*>(dst + 4) = hi;
          }
          continue;
        }
        if (p.R) {
          const uint4 u = *reinterpret_cast<const uint4*>(p.R + (long)m * p.ldr + n);
          const uint32_t* w = reinterpret_cast<const uint32_t*>(&u);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u); }
        }
        bf16_t* cp = p.C + (long)m * p.ldc + n;
        if (p.accumulate) {
          const uint4 u = WGRAD_C ? ld_stream16(cp) : *reinterpret_cast<const uint4*>(cp);
          const uint32_t* w = reinterpret_cast<const uint32_t*>(&u);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u); }
        }
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        if (WGRAD_C) st_stream16(cp, o); else *reinterpret_cast<uint4*>(cp) = o;
      }
    }
  } else {
  // generic path (unaligned / narrow outputs such as conv_out's 4 channels)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WM + 16 * i + lm;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * WN + 16 * j + ln;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const bool full = (n + 3) < p.N;
      if (p.ksplit > 1) {
        float* dst = p.ws + ((long)z * p.M + m) * p.N + n;
        if (p.tickets) { for (int e = 0; e < 4 && n + e < p.N; ++e) __hip_atomic_store(dst + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // sc1 stores
        else if (full && ((p.N & 3) == 0)) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = v[e];
        continue;
      }
      if (p.bias) {
        for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += bf2f(p.bias[n + e]);
      }
      if (p.rowbias) {
        const bf16_t* rbp = p.rowbias + (long)(m / p.rows_per_seg) * p.ld_rb + n;
        for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += bf2f(rbp[e]);
      }
      if (p.R) {
        const bf16_t* rp = p.R + (long)m * p.ldr + n;
        for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += bf2f(rp[e]);
      }
      bf16_t* cp = p.C + (long)m * p.ldc + n;
      if (p.accumulate) {
        for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += bf2f(cp[e]);
      }
      if (full && ((p.ldc & 3) == 0)) {
        uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(cp) = o;
      } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) cp[e] = f2bf(v[e]);
      }
    }
  }
  }

#ifdef AZ_ANATOMY
  AZ_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AZ_STAMP(6); AZ_STAMP(9);
#endif
  // ---- in-kernel finish (p.tickets != nullptr): split-K slab sum and column-sum finish by the tile's LAST arriver ----------
  // Hand-off between workgroups as cdna_hip_programming.md section 5 ("In-launch split-K reduction", the write-through form)
  // prescribes: slabs and column-sum slots are stored WRITE-THROUGH (sc1: no release fence, whose L2 write-back per workgroup cost
  // the step +10 ms when tried), every wave drains its stores (vmcnt(0)), the workgroup meets at a barrier and ONE lane draws a
  // ticket with a relaxed agent-scope add; the workgroup that draws ksplit-1 resets the counter, takes ONE agent-scope acquire
  // (drops this CU's stale L1 lines), meets at a barrier and only then reads the other workgroups' slabs with plain loads.
  // Placement-independent; the summation order is the split index, never the arrival order.
  if (p.tickets == nullptr) return;
  const bool has_cs = CS && p.cs_ws != nullptr && tn == 0;
  if (p.ksplit == 1 && !has_cs) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (p.ksplit > 1) {
    if (t == 0) {
      unsigned* cnt = p.tickets + (tm * p.tiles_n + tn);
      const unsigned tk = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned is_last = (tk == (unsigned)(p.ksplit - 1)) ? 1u : 0u;
      if (is_last) {
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left zero for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *reinterpret_cast<volatile unsigned*>(smem) = is_last;
    }
    __syncthreads();
    if (*reinterpret_cast<volatile unsigned*>(smem) == 0u) return;
    // sum the tile's slabs in ascending split order (+ bias, + existing C)
    constexpr int NT = NW * 64;
    if (p.vec_epi) {
      constexpr int CH8 = BN / 8;
      for (int item = t; item < BM * CH8; item += NT) {
        const int r = item / CH8, cc = item - r * CH8;
        const int m = m0 + r, n = n0 + cc * 8;
        if (m >= p.M || n >= p.N) continue;
        const float* src = p.ws + (long)m * p.N + n;
        const long slab = (long)p.M * p.N;
        float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        for (int zz = 1; zz < p.ksplit; ++zz) {
          const float4 a = *reinterpret_cast<const float4*>(src + zz * slab);
          const float4 b = *reinterpret_cast<const float4*>(src + zz * slab + 4);
          lo.x += a.x; lo.y += a.y; lo.z += a.z; lo.w += a.w; hi.x += b.x; hi.y += b.y; hi.z += b.z; hi.w += b.w;
        }
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(p.bias[n + e]);
        }
        bf16_t* cp = p.C + (long)m * p.ldc + n;
        if (p.accumulate) {
          const uint4 u = WGRAD_C ? ld_stream16(cp) : *reinterpret_cast<const uint4*>(cp);
          const uint32_t* w = reinterpret_cast<const uint32_t*>(&u);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u); }
        }
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        if (WGRAD_C) st_stream16(cp, o); else *reinterpret_cast<uint4*>(cp) = o;
      }
    } else {
      for (int item = t; item < BM * BN; item += NT) {
        const int r = item / BN, cc = item - r * BN;
        const int m = m0 + r, n = n0 + cc;
        if (m >= p.M || n >= p.N) continue;
        float sum = 0.f;
        for (int zz = 0; zz < p.ksplit; ++zz) sum += p.ws[((long)zz * p.M + m) * p.N + n];
        if (p.bias) sum += bf2f(p.bias[n]);
        bf16_t* cp = p.C + (long)m * p.ldc + n;
        if (p.accumulate) sum += bf2f(*cp);
        *cp = f2bf(sum);
      }
    }
  }
  if constexpr (CS) {
    if (has_cs) {       // every split of this row block has published its column-sum slots (same ticket): finish rows m0 .. m0+BM-1
      const ColsumFinish c{p.ksplit, p.cs_nseg, p.M, p.ktiles_per_split, (p.K + BK - 1) / BK, p.cs_rps, p.cs_ws, p.cs_seg_out, p.cs_bias, p.cs_n_real};
      for (int r = t; r < BM; r += NW * 64) colsum_finish(c, m0 + r);
    }
  }
}

// out[m][n] (bf16) = sum_z ws[z][m][n] (+bias) (+out if accumulate)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int S, long MN, int N, bf16_t* out, long ldc,
                                     const bf16_t* bias, int accumulate) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < MN; i += stride) {
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += ws[(long)z * MN + i];
    long m = i / N; int n = (int)(i - m * N);
    if (bias) s += bf2f(bias[n]);
    bf16_t* o = out + m * ldc + n;
    if (accumulate) s += bf2f(*o);
    *o = f2bf(s);
  }
}

__global__ void colsum_finish_kernel(const ColsumFinish c) { colsum_finish(c, blockIdx.x * blockDim.x + threadIdx.x); }

// the same, 8 consecutive columns per thread with 16-byte accesses (N % 8 == 0, rows of `out` 16-byte aligned):
// the scalar form ran at 1.8 TB/s and cost as much as the split-K product it finishes
// Blocks >= red_blocks finish the fused column sums of the same product (bias / time-embedding gradients) instead: one launch
// less per weight gradient.  The slab loads go out four slabs at a time; the sums stay in ascending-z order.
__global__ void splitk_reduce_vec_kernel(const float* __restrict__ ws, int S, long MN, int N, bf16_t* out, long ldc,
                                         const bf16_t* bias, int accumulate, int red_blocks, const ColsumFinish cs,
                                         const bf16_t* res, long ldr, int stream_out) {
  if ((int)blockIdx.x >= red_blocks) {
    colsum_finish(cs, ((int)blockIdx.x - red_blocks) * blockDim.x + threadIdx.x);
    return;
  }
  const long n8 = MN >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)red_blocks * blockDim.x) {
    const long e = i << 3;
    float4 lo = *reinterpret_cast<const float4*>(ws + e), hi = *reinterpret_cast<const float4*>(ws + e + 4);
    int z = 1;
    for (; z + 3 < S; z += 4) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const float4*>(ws + (long)(z + u) * MN + e);
        b[u] = *reinterpret_cast<const float4*>(ws + (long)(z + u) * MN + e + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        lo.x += a[u].x; lo.y += a[u].y; lo.z += a[u].z; lo.w += a[u].w; hi.x += b[u].x; hi.y += b[u].y; hi.z += b[u].z; hi.w += b[u].w;
      }
    }
    for (; z < S; ++z) {
      const float4 a = *reinterpret_cast<const float4*>(ws + (long)z * MN + e);
      const float4 b = *reinterpret_cast<const float4*>(ws + (long)z * MN + e + 4);
      lo.x += a.x; lo.y += a.y; lo.z += a.z; lo.w += a.w; hi.x += b.x; hi.y += b.y; hi.z += b.z; hi.w += b.w;
    }
    long m; int n; divmod(e, N, m, n);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (bias) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += bf2f(bias[n + k]);
    }
    bf16_t* o = out + m * ldc + n;
    if (res) {          // out-of-place residual (same order as the unsplit epilogue: product + bias, + residual, + accumulate)
      const uint4 u = *reinterpret_cast<const uint4*>(res + m * ldr + n);
      const uint32_t* w = reinterpret_cast<const uint32_t*>(&u);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(w[k] << 16); v[2 * k + 1] += __uint_as_float(w[k] & 0xFFFF0000u); }
    }
    if (accumulate) {
      const uint4 u = stream_out ? ld_stream16(o) : *reinterpret_cast<const uint4*>(o);
      const uint32_t* w = reinterpret_cast<const uint32_t*>(&u);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(w[k] << 16); v[2 * k + 1] += __uint_as_float(w[k] & 0xFFFF0000u); }
    }
    uint4 r;
    r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]); r.z = pack2bf(v[4], v[5]); r.w = pack2bf(v[6], v[7]);
    if (stream_out) st_stream16(o, r); else *reinterpret_cast<uint4*>(o) = r;
  }
}

template <int AMODE, int BMODE, int BM, int BN, int NWM = BM / 64, int NWN = BN / 64, int NS = 2, int KB = 64, int MODE = 0>
int launch_tile(const Params& p, hipStream_t st) {
  constexpr int LDS = NS * (BM + BN) * KB * 2 + 2048;      // + the scratch row of the rings' padding DMAs
  static bool attr_set = false;
  auto kern = gemm_kernel<AMODE, BMODE, BM, BN, NWM, NWN, NS, KB, MODE>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return -(int)e;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, p.ksplit, 1);
  if (p.xsplit) grid = dim3(p.tiles_m * p.tiles_n * p.ksplit, 1, 1);
  az_launch(kern, grid, dim3(NWM * NWN * 64), LDS, st, p);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

// 16-byte coalesced epilogue allowed: N % 8 == 0, rows of C / R / the slabs 16-byte aligned
bool vec_epi_ok(const Params& p) {
  return ((p.N & 7) == 0) && ((p.ldc & 7) == 0) && (((uintptr_t)p.C & 15) == 0) &&
         (!p.R || (((p.ldr & 7) == 0) && (((uintptr_t)p.R & 15) == 0))) && (!p.ws || p.ksplit == 1 || (((uintptr_t)p.ws & 15) == 0));
}

template <int AMODE, int BMODE>
int launch(Params& p, hipStream_t st) {
  // 32-bit buffer offsets: every operand extent must stay below 2 GiB
  const long GB2 = 0x7FFFFFF0L;
  long ext_a, ext_b;
  if (AMODE == A_ROW) ext_a = ((long)p.M * p.lda + p.K) * 2;
  else if (AMODE == A_COL) ext_a = ((long)p.K * p.lda + p.M + 8) * 2;
  else if (AMODE == A_CONV) ext_a = ((long)(p.M / (p.g.Hout * p.g.Wout)) * p.g.Hin * p.g.Win) * p.lda * 2;
  else ext_a = (long)p.M / (p.g.Hin * p.g.Win) * p.g.Hout * p.g.Wout * p.lda * 2;
  if (BMODE == B_NT) ext_b = ((long)p.N * p.ldb + p.K) * 2;
  else if (BMODE == B_NN) ext_b = ((long)p.K * p.ldb + p.N + 8) * 2;
  else if (BMODE == B_CONVDG) ext_b = (long)p.g.Cout * 9 * p.g.Cin * 2;
  else ext_b = (long)(p.K / (p.g.Hout * p.g.Wout)) * p.g.Hin * p.g.Win * p.ldb * 2;
  if (ext_a >= GB2 || ext_b >= GB2 || p.lda * 2 >= GB2 || p.ldb * 2 >= GB2) return AZ_ERR_ARG(8);
  p.lda2 = (int)(p.lda * 2); p.ldb2 = (int)(p.ldb * 2);
  p.k_full = (p.K % BK) == 0;
  p.ablate = az_opt(AZ_OPT_GEMM_ABLATE);
  p.vec_epi = vec_epi_ok(p);
  if constexpr (AMODE == A_ROW && BMODE == B_NT) {
    if (p.use8) {          // 8-wave ping-pong tile (apply_gemm8 checked K % 64 == 0, the vector epilogue, no split, no row bias)
      if (p.gg_y) return launch_gemm8<256, 2>(p, st);
      if (p.bn == 320) return launch_gemm8<320, 0>(p, st);
      return launch_gemm8<256, 0>(p, st);
    }
  }
  if (p.gg_y) {          // fused GEGLU forward: plain NT product, 2 stages of 64-deep k-tiles, vector epilogue only
    if (!p.vec_epi || p.ksplit != 1) return AZ_ERR_ARG(9);
    if constexpr (AMODE == A_ROW && BMODE == B_NT) {
      if (p.bm == 256 && p.bn == 256) return launch_tile<AMODE, BMODE, 256, 256, 4, 4, 2, 64, 2>(p, st);
      return launch_tile<AMODE, BMODE, 128, 128, 4, 2, 2, 64, 2>(p, st);
    } else {
      return AZ_ERR_ARG(9);
    }
  }
#ifdef AZ_EXP_MINIMAL      // experiment builds (tools/build_exp.sh): only the default 8-wave 128x128 / 128x160 and the 256x256 tile
  if constexpr (BMODE == B_NT) {
    if (p.bm == 128 && p.bn == 160) {
      // (round 5: 4 waves of 64x80 / 64x64 on the same block tiles -- 460 instead of 717 LDS bytes read per MFMA -- were built here and
      //  removed: isolated +-2 %, 3-stage form 6 % slower, two-stream micro-step +1.7 / +0.4 ms; profiles/r05_w4_tiles.txt)
      if (p.stages == 4) return launch_tile<AMODE, BMODE, 128, 160, 4, 2, 4>(p, st);
      if (p.stages == 3) return launch_tile<AMODE, BMODE, 128, 160, 4, 2, 3>(p, st);
      return launch_tile<AMODE, BMODE, 128, 160, 4, 2>(p, st);
    }
  }
  if (p.bm == 256 && p.bn == 256) return launch_tile<AMODE, BMODE, 256, 256>(p, st);
  return launch_tile<AMODE, BMODE, 128, 128, 4, 2>(p, st);
#else
  if constexpr (BMODE == B_NT) {      // 160-wide N tiles exist for k-contiguous B only (every SDXL width is a multiple of 160)
    if constexpr (AMODE != A_COL) {     // deep rings of 32-deep k-tiles: the same LDS footprint keeps 1.5x / 2x the k-depth in flight
      if (p.kb == 32 && p.bm == 128 && p.bn == 160) {
        if (p.stages == 5) return launch_tile<AMODE, BMODE, 128, 160, 4, 2, 5, 32>(p, st);
        return launch_tile<AMODE, BMODE, 128, 160, 4, 2, 4, 32>(p, st);
      }
      if (p.kb == 32 && p.bm == 256 && p.bn == 256) return launch_tile<AMODE, BMODE, 256, 256, 4, 4, 4, 32>(p, st);
    }
    if (p.bm == 128 && p.bn == 160) {
      if (p.stages == 4) return launch_tile<AMODE, BMODE, 128, 160, 4, 2, 4>(p, st);
      if (p.stages == 3) return launch_tile<AMODE, BMODE, 128, 160, 4, 2, 3>(p, st);
      return launch_tile<AMODE, BMODE, 128, 160, 4, 2>(p, st);
    }
  } else if (p.bn == 160) {
    return AZ_ERR_ARG(9);
  }
  if (p.bm == 256 && p.bn == 256) return launch_tile<AMODE, BMODE, 256, 256>(p, st);
  return launch_tile<AMODE, BMODE, 128, 128, 4, 2>(p, st);
#endif
}

ColsumFinish colsum_args(const Params& p, void* seg_grad, void* bias_grad, int n_real) {
  return ColsumFinish{p.ksplit, p.cs_nseg, p.M, p.ktiles_per_split, (p.K + BK - 1) / BK, p.cs_rps, (const float*)p.cs_ws,
                      (bf16_t*)seg_grad, (bf16_t*)bias_grad, n_real};
}

// split-K slab reduction and (when the product carried fused column sums) their finish, in one launch where possible
int finish_product(const Params& p, hipStream_t st, void* seg_grad = nullptr, void* bias_grad = nullptr, int n_real = 0) {
  if (p.tickets) return AZ_OK;        // the product finished itself (last arriver per tile)
  if (az_opt(AZ_OPT_GEMM_ABLATE) & 4) return AZ_OK;      // diagnostic (timing only, results wrong): no reduce / finish launches
  const bool cs = p.cs_ws != nullptr;
  const ColsumFinish c = cs ? colsum_args(p, seg_grad, bias_grad, n_real) : ColsumFinish{};
  const int cs_blocks = cs ? (p.M + 255) / 256 : 0;
  long MN = (long)p.M * p.N;
  const int fused = 1;
  if (p.ksplit > 1 && p.vec_epi) {   // N % 8 == 0, ldc % 8 == 0, C and the slabs 16-byte aligned (the slab pitch M*N is then a multiple of 8 too)
    int blocks = (int)((MN / 8 + 255) / 256); if (blocks > 4096) blocks = 4096;      // (fewer, longer-lived blocks measured neutral to worse in the step: 1024 / 512 same, 256 / 128 +0.4 ms)
    az_launch(splitk_reduce_vec_kernel, dim3(blocks + (fused ? cs_blocks : 0)), dim3(256), 0, st, p.ws, p.ksplit, MN, p.N, p.C, p.ldc,
                       p.bias, p.accumulate, blocks, c, p.R, p.ldr, p.cs_ws != nullptr || p.wgrad_c);
    AZ_CHECK_LAUNCH();
    if (fused) return AZ_OK;
  } else if (p.ksplit > 1) {
    int blocks = (int)((MN + 255) / 256); if (blocks > 2048) blocks = 2048;
    az_launch(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.ws, p.ksplit, MN, p.N, p.C, p.ldc, p.bias, p.accumulate);
    AZ_CHECK_LAUNCH();
  }
  if (cs) {
    az_launch(colsum_finish_kernel, dim3(cs_blocks), dim3(256), 0, st, c);
    AZ_CHECK_LAUNCH();
  }
  return AZ_OK;
}

int g_force_bm = 0, g_force_bn = 0, g_force_nw = 0, g_force_stages = 0;
   // tuning hook (az_gemm_set_tile)

// Tile choice: bigger cooperative tiles halve the L2->LDS bytes per FLOP but run 1 workgroup / CU, so they
// only pay when the grid still covers the 256 CUs well.
// Measured on MI355X (tools/gemm_tiles.py): the 16-wave 256x256 tile wins (+20..50 %) for forward / dgrad
// products when its grid fills >= 50 % of whole waves of 256 CUs (70 % in isolation; in the two-stream step 50 % is 0.7 ms
// better, same-box A/B, AZ_BIG_FILL); it loses for the 320-tile (N = 1280)
// family and for wgrad (split-K over pixels), which stay on 128x128 at 2 workgroups / CU.
void choose_tile(Params& p, bool wgrad, bool b_kmajor) {
  p.nwaves = 0; p.stages = 2; p.kb = 64;
  if (g_force_bm == 256 && g_force_nw == 8) {      // forced 8-wave ping-pong tile: apply_gemm8 takes it where it exists, else the 16-wave 256x256 tile
    p.bm = 256; p.bn = 256; p.nwaves = 0;
    return;
  }
  if (g_force_bm) {
    const bool kb32 = ((g_force_nw >> 5) & 1) && b_kmajor, deep = (g_force_nw >> 4) & 1;
    p.bm = g_force_bm; p.bn = g_force_bn; p.nwaves = g_force_nw & 15;
    p.kb = kb32 ? 32 : 64; p.stages = kb32 ? (deep ? 5 : 4) : (deep ? 3 : 2);
    if ((g_force_nw >> 6) & 1) { p.kb = 64; p.stages = 4; }      // 72 = 8 waves, 4 stages of 64-deep k-tiles (147 KiB)
    if (p.bn == 160 && !b_kmajor) { p.bn = 128; p.nwaves = 8; p.stages = 2; }   // the forced 160-wide tile only applies where it exists
    if (((g_force_nw >> 5) & 1) && !b_kmajor && p.bm == 256) { p.nwaves = 0; p.stages = 2; }
    return;
  }
  // Option TILE_POLICY: 4 (default) = the 3-stage 128x160 variant only while LDS_EXCLUSIVE is set (forward pass); 5 = also in
  // the backward pass (right when most weights are frozen: in the full two-stream step its 108 KiB lock the weight-gradient
  // stream's workgroups out of the CU, +3 ms per micro-step in the same-process A/B of tools/ab_opts.py).
  const int policy = az_opt(AZ_OPT_TILE_POLICY);
  const bool exclusive = az_opt(AZ_OPT_LDS_EXCLUSIVE) != 0;
  p.bm = 128; p.bn = 128; p.nwaves = 8;     // 4x2 waves of 32x64: +5..15 % over 2x2 waves of 64x64 (tools/gemm_tiles.py)
  if (wgrad) return;
  const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  const long waves = (t256 + 255) / 256;
  const bool big = t256 * 10 >= waves * 256 * az_opt(AZ_OPT_BIG_FILL);     // tenths of whole waves of 256 CUs
  if (b_kmajor && (p.N % 160) == 0 && (p.N <= 640 || !big)) {
    p.bn = 160; p.nwaves = 8;
    // a grid of at most one workgroup per CU leaves LDS idle: spend it on a third stage (prefetch distance 2), which hides the
    // HBM latency of cold operands (+16..21 % on the M = 4096, N = 1280 family; tools/r2_gemm_sweep.py, tools/r2_ablate.py:
    // the 2-stage loop is DMA-latency-bound, 169 of 183 us for K = 10240)
    const long t160 = (long)((p.M + 127) / 128) * (p.N / 160);
    if ((policy >= 5 || exclusive) && t160 <= 256 && p.N > 640) p.stages = 3;
    return;
  }
  if (big) { p.bm = 256; p.bn = 256; p.nwaves = 0; }
}

int choose_split(Params& p, int want_split, long ws_bytes, bool wgrad = false, bool b_kmajor = false, bool big_split = false) {
  if (big_split) { p.bm = 256; p.bn = 256; p.nwaves = 0; p.stages = 2; p.kb = 64; }
  else choose_tile(p, wgrad, b_kmajor);
  p.tiles_m = (p.M + p.bm - 1) / p.bm;
  p.tiles_n = (p.N + p.bn - 1) / p.bn;
  const int ktiles = (p.K + BK - 1) / BK;
  int s = 1;
  if (want_split != 1 && p.ws) {
    const int tiles = p.tiles_m * p.tiles_n;
    if (want_split > 1) {
      s = want_split;
    } else {
      // measured (tools/splitk_sweep.py): in isolation best when the grid fills whole waves of 512 workgroup slots (2 / CU); beside the
      // data-gradient stream 384 slots are 0.75 ms per micro-step better (same-box A/B, 4 repetitions: a 100-tile weight gradient in 3 slabs instead of 5);
      // every extra split costs an fp32 slab round trip (~8 % each), and a split needs >= 8 k-tiles to amortise
      const int slots = az_opt(AZ_OPT_SPLIT_SLOTS);
      double best = -1.0;
      const int nosplit = az_opt(AZ_OPT_NOSPLIT_TILES);
      for (int c = 1; c <= (tiles >= nosplit ? 1 : 24); ++c) {      // grids of >= 256 tiles (one per CU) are never split
        if (c > 1 && ktiles / c < 8) break;
        const long blocks = (long)tiles * c;
        const double fill = (double)blocks / (double)(((blocks + slots - 1) / slots) * slots);
        const double score = fill / (1.0 + 0.08 * (c - 1));
        if (score > best + 1e-9) { best = score; s = c; }
      }
    }
    if (s > ktiles) s = ktiles;
    if (s > 64) s = 64;
    while (s > 1 && (long)s * p.M * p.N * 4 > ws_bytes) --s;   // (ws_bytes already excludes the column-sum slots)
    if (s < 1) s = 1;
  }
  p.ktiles_per_split = (ktiles + s - 1) / s;
  p.ksplit = (ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;
  if (p.ksplit < 1) p.ksplit = 1;
  p.xsplit = (wgrad && az_opt(AZ_OPT_XCD_SPLIT) != 0 && p.ksplit > 1) ? 1 : 0;
  p.wgrad_c = wgrad ? 1 : 0;
  return AZ_OK;
}

// The 8-wave ping-pong tile (az_gemm8.inc) for a plain k-contiguous product whose tile policy came out as the 256-row tile:
// 256x256, or 256x320 where N is a multiple of 320 and that grid fills whole waves of 256 CUs better (4096x5120: 256 tiles
// = one wave instead of 320 = 1.25; 16384x2560 and 4096x10240: 512 = two waves instead of 640 = 2.5).  Option GEMM8 = 0: the 16-wave tile.
void apply_gemm8(Params& p, int want_bn = 0) {
  p.use8 = 0;
  const bool forced = (g_force_bm == 256 && g_force_nw == 8);
  if (!forced && (g_force_bm || !az_opt(AZ_OPT_GEMM8))) return;
  if (p.bm != 256 || p.bn != 256 || (p.K % BK) || p.rowbias || !vec_epi_ok(p)) return;
  int bn = 256;
  if (forced) {
    bn = (g_force_bn == 320 && (p.N % 320) == 0) ? 320 : 256;
  } else if (want_bn) {
    bn = want_bn;
  } else if ((p.N % 320) == 0 && p.ksplit == 1) {
    auto fill = [&](int w) { const long tl = (long)((p.M + 255) / 256) * ((p.N + w - 1) / w); return (double)tl / (double)(((tl + 255) / 256) * 256); };
    if (fill(320) > fill(256) + 1e-9) bn = 320;
  }
  p.use8 = 1; p.bn = bn;
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + bn - 1) / bn;
}

}  // namespace

extern "C" {

int az_gemm_set_exclusive(int on) { return az_set_option("LDS_EXCLUSIVE", on ? 1 : 0); }

int az_gemm_set_tile(int bm, int bn) { return az_gemm_set_tile_ex(bm, bn, 0); }

int az_gemm_set_tile_ex(int bm, int bn, int waves) {
  const bool std_tile = (bm == 128 && bn == 128 && (waves == 0 || waves == 8)) ||
                        (bm == 256 && bn == 256 && (waves == 0 || waves == 32 || waves == 8)) ||   /* 32 = 4 stages of 32-deep k-tiles, 8 = the 8-wave ping-pong tile */
                        (bm == 256 && bn == 320 && waves == 8);
  const bool n160 = (bm == 128 && bn == 160 && (waves == 0 || waves == 8 || waves == 24 ||   /* 24 = 8 waves, 3 stages */
                                                waves == 40 || waves == 56 || waves == 72));   /* 40 / 56 = 4 / 5 stages of 32-deep k-tiles, 72 = 4 stages of 64-deep ones */
  if (!((bm == 0 && bn == 0) || std_tile || n160)) return AZ_ERR_ARG(9);
  g_force_bm = bm; g_force_bn = bn; g_force_nw = waves;
  return AZ_OK;
}

// carve [64 splits][nseg][M] fp32 column-sum slots off the END of the split-K workspace
// With option INKERNEL_FINISH on, the LAST 16 KiB of the caller's workspace hold the arrival counters of the in-kernel finish
// (one per output tile, <= 4096 tiles), zeroed by a memset node in front of every such launch: nothing is assumed about the
// workspace's contents in either mode.  With the option off (the default) nothing is reserved.
constexpr long TICKET_BYTES = 16384;
static void carve_tickets(Params& p, void* workspace, long& workspace_bytes) {
  p.tickets = nullptr;
  if (!workspace || workspace_bytes < TICKET_BYTES + 65536 || ((uintptr_t)workspace & 15) || (workspace_bytes & 15)) return;
  if (!az_opt(AZ_OPT_INKERNEL_FINISH)) return;          // option off: the whole workspace is slab / column-sum space, no contract on its contents
  workspace_bytes -= TICKET_BYTES;
  p.tickets = (unsigned*)((char*)workspace + workspace_bytes);
}

static int carve_colsum(Params& p, void* workspace, long& workspace_bytes, int nseg, int rps) {
  const long need = 64L * nseg * p.M * 4;
  if (!workspace || workspace_bytes < need + 4096 || ((uintptr_t)workspace & 15)) return AZ_ERR_ARG(20);
  if (nseg > 1 && (rps % BK)) return AZ_ERR_ARG(21);
  workspace_bytes = ((workspace_bytes - need) / 16) * 16;
  p.cs_ws = (float*)((char*)workspace + workspace_bytes);
  p.cs_nseg = nseg; p.cs_rps = rps;
  return AZ_OK;
}

static int gemm_impl(int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                 void* C, long ldc, const void* bias, const void* rowbias, int rows_per_seg, long ld_rowbias,
                 const void* residual, long ldr, int accumulate, int split_k, void* workspace, long workspace_bytes,
                 void* bias_grad, int n_real, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return AZ_ERR_ARG(1);
  if ((K & 7) && !transA && !transB) return AZ_ERR_ARG(2);
  if ((lda & 7) || (ldb & 7)) return AZ_ERR_ARG(3);
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return AZ_ERR_ARG(4);
  if (rowbias && rows_per_seg <= 0) return AZ_ERR_ARG(5);
  Params p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K;
  p.C = (bf16_t*)C; p.ldc = ldc; p.ws = (float*)workspace; p.bias = (const bf16_t*)bias;
  p.rowbias = (const bf16_t*)rowbias; p.rows_per_seg = rows_per_seg; p.ld_rb = ld_rowbias;
  p.R = (const bf16_t*)residual; p.ldr = ldr; p.accumulate = accumulate;
  carve_tickets(p, workspace, workspace_bytes);
  if (bias_grad) {
    if (!(transA && !transB) || n_real > M) return AZ_ERR_ARG(22);
    int rc0 = carve_colsum(p, workspace, workspace_bytes, 1, K);
    if (rc0) return rc0;
    p.cs_bias = (bf16_t*)bias_grad; p.cs_n_real = n_real; p.cs_seg_out = nullptr;
  }
  // k-heavy linear products whose output is only ~80 tiles of 256x256 (M = 4096, N = 1280 at local batch 4): one 128x160 tile
  // per CU streams 36.9 KB of operands per 64-deep k-step through a DMA path that delivers ~70 GB/s per CU (MI355X_MICROARCH,
  // "Indexed rows: gather into LDS"), i.e. it is operand-bandwidth-bound at ~1/3 of the MFMA rate; the 256x256 tile moves half
  // the bytes per FLOP, and splitting k by 3 puts 240 of them on the 256 CUs.  Measured cold (tools/r2_gemm_sweep.py):
  // 4096x1280x10240 177 -> 116 us, x3840 68 -> 61 us (the fp32 slab round trip of the split included).
  // Few-tile k-heavy products (M x N = 4096 x 1280 at local batch 4: 80 tiles of 256x256) on 256-row tiles with k split so
  // that the grid covers the chip: half the L2->LDS bytes per FLOP of the 128x160 tile they would otherwise run on, at the
  // price of an fp32 slab round trip (finished, bias / residual / accumulate included, by splitk_reduce_vec_kernel).  With the
  // 8-wave ping-pong tile the candidates are 256x256 and (N % 320 == 0) 256x320; the pair (width, splits) that fills the 256 CUs
  // best wins, fewer splits on ties.  Options: NT_SPLIT_FWD / NT_SPLIT_BIG = largest split count tried in the exclusive forward pass / beside the weight-gradient stream (0 = off), NT_SPLIT_MINK = smallest K.
  bool big_split = false; int split_bn = 256;
  {
    const int sb = az_opt(AZ_OPT_LDS_EXCLUSIVE) ? az_opt(AZ_OPT_NT_SPLIT_FWD) : az_opt(AZ_OPT_NT_SPLIT_BIG);
    if (sb > 1 && !transA && transB && !g_force_bm && !rowbias && workspace && (K % BK) == 0 && K >= az_opt(AZ_OPT_NT_SPLIT_MINK) &&
        (split_k == 0 || split_k == 1) && (!residual || (((ldr & 7) == 0) && (((uintptr_t)residual & 15) == 0))) &&
        ((N & 7) == 0) && ((ldc & 7) == 0) && (((uintptr_t)C & 15) == 0) && (((uintptr_t)workspace & 15) == 0)) {
      const int kt = K / BK;
      double best = 0.0; int bs = 0;
      for (int bn = 256; bn <= 320; bn += 64) {
        if (bn == 320 && ((N % 320) || !az_opt(AZ_OPT_GEMM8))) continue;
        const long tl = (long)((M + 255) / 256) * ((N + bn - 1) / bn);
        if (tl > 128) continue;          // a grid that fills half the chip unsplit stays unsplit
        for (int sp = 2; sp <= sb; ++sp) {
          if (kt / sp < 8 || tl * sp > 256 || (long)sp * M * N * 4 > workspace_bytes) break;
          const double score = (double)(tl * sp) / 256.0 - 0.02 * (sp - 1);
          if (score > best + 1e-9) { best = score; bs = sp; split_bn = bn; }
        }
      }
      if (bs > 1 && best >= 0.7) { big_split = true; split_k = bs; }
    }
  }
  // The same family on its 256 tiles of 128x160 (one workgroup per CU) runs at the DMA round trip of its single workgroup; with k
  // split in two, two workgroups share each CU and cover each other's waits (tools/gemm_ab: 4096x1280x10240 171 -> 127 us,
  // x5120 89 -> 74, x3840 68 -> 60, the slab reduce included; K = 1280 loses).  Not in the exclusive forward pass, where the
  // 3-stage variant does the same job without slabs.  Option NT_SPLIT2_MINK, OFF by default: in the two-stream backward pass
  // the extra workgroups displace the weight-gradient stream's (chain alone -2.5 ms, whole micro-step +1.5 ms).
  if (!big_split && !transA && transB && !g_force_bm && !rowbias && !residual && workspace && (K % BK) == 0 && (split_k == 0 || split_k == 1) &&
      !az_opt(AZ_OPT_LDS_EXCLUSIVE) && az_opt(AZ_OPT_TILE_POLICY) < 5) {
    const int mink = az_opt(AZ_OPT_NT_SPLIT2_MINK);
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long t160 = (long)((M + 127) / 128) * (N / 160);
    const bool big = t256 * 10 >= ((t256 + 255) / 256) * 256 * az_opt(AZ_OPT_BIG_FILL);
    if (mink > 0 && K >= mink && (N % 160) == 0 && N > 640 && !big && t160 <= 256 && 2L * M * N * 4 <= workspace_bytes - 65536) split_k = 2;
  }
  choose_split(p, split_k, workspace_bytes, transA != 0, !transA && transB, big_split);
  if (p.ksplit > 1 && (rowbias || (residual && !vec_epi_ok(p)))) return AZ_ERR_ARG(6);      // (the scalar reduce kernel adds no residual)
  if (!transA && transB) apply_gemm8(p, big_split ? split_bn : 0);
  // (a split product with an out-of-place residual is always finished by the reduce launch: the in-kernel finish adds no residual)
  if ((long)p.tiles_m * p.tiles_n > TICKET_BYTES / 4 || (long)p.ksplit * p.M * p.N * 4 >= 0x7FFFFFF0L || p.use8 || (p.R && p.ksplit > 1)) p.tickets = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (p.tickets) AZ_HIP(hipMemsetAsync(p.tickets, 0, (size_t)p.tiles_m * p.tiles_n * 4, st));      // the counters start from zero whatever the workspace held
  int rc;
  if (!transA && transB) rc = launch<A_ROW, B_NT>(p, st);
  else if (!transA && !transB) rc = launch<A_ROW, B_NN>(p, st);
  else if (transA && !transB) rc = launch<A_COL, B_NN>(p, st);
  else return AZ_ERR_ARG(7);
  if (rc) return rc;
  return finish_product(p, st, nullptr, bias_grad, n_real);
}

int az_gemm_bf16(int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                 void* C, long ldc, const void* bias, const void* rowbias, int rows_per_seg, long ld_rowbias,
                 const void* residual, long ldr, int accumulate, int split_k, void* workspace, long workspace_bytes,
                 void* stream) {
  return gemm_impl(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, rowbias, rows_per_seg, ld_rowbias, residual, ldr,
                   accumulate, split_k, workspace, workspace_bytes, nullptr, 0, stream);
}

int az_gemm_geglu_fwd_bf16(int M, int H, int K, const void* X, long lda, const void* W, long ldb, const void* bias, void* proj, long ldp,
                           void* out, long ldo, void* stream) {
  if (M <= 0 || H <= 0 || K <= 0 || (K & 7) || (H & 7)) return AZ_ERR_ARG(72);
  if ((lda & 7) || (ldb & 7) || (ldp & 7) || (ldo & 7)) return AZ_ERR_ARG(73);
  if (((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)proj & 15) || ((uintptr_t)out & 15) || !proj || !out) return AZ_ERR_ARG(74);
  Params p{};
  p.A = (const bf16_t*)X; p.B = (const bf16_t*)W; p.lda = lda; p.ldb = ldb; p.M = M; p.N = 2 * H; p.K = K;
  p.C = (bf16_t*)proj; p.ldc = ldp; p.bias = (const bf16_t*)bias; p.gg_H = H; p.gg_y = (bf16_t*)out; p.gg_ldy = ldo;
  // a tile covers bn/2 value columns: the 16-wave 256x256 tile when its grid over [M][H] fills whole waves of CUs like the plain product's
  // does over [M][2H] (choose_tile's rule), else the 8-wave 128x128 tile
  choose_split(p, 1, 0, false, true);
  const bool big = (p.bm == 256 && p.bn == 256);
  p.stages = 2; p.kb = 64;
  if (!big) { p.bm = 128; p.bn = 128; }
  p.nwaves = big ? 0 : 8;
  p.tiles_m = (M + p.bm - 1) / p.bm; p.tiles_n = (H + p.bn / 2 - 1) / (p.bn / 2);
  // the 256-row tile on the 8-wave ping-pong kernel (128 value + 128 gate columns per tile)
  p.use8 = (big && (K % BK) == 0 && vec_epi_ok(p) && (az_opt(AZ_OPT_GEMM8) || (g_force_bm == 256 && g_force_nw == 8)) &&
            (!g_force_bm || g_force_nw == 8)) ? 1 : 0;
  return launch<A_ROW, B_NT>(p, (hipStream_t)stream);
}

int az_gemm_nt_grouped_bf16(int M, int K, const void* A, long lda, const void* groups_dev, int ngroups, long total_tiles_n, void* stream) {
  if (M <= 0 || K <= 0 || (K & 7) || (lda & 7) || ((uintptr_t)A & 15) || !groups_dev || ngroups <= 0 || total_tiles_n <= 0 ||
      ((uintptr_t)groups_dev & 7)) return AZ_ERR_ARG(70);
  Params p{};
  p.A = (const bf16_t*)A; p.lda = lda; p.M = M; p.K = K; p.N = 0;
  p.groups = (const GemmGroup*)groups_dev; p.ngroups = ngroups;
  p.bm = 128; p.bn = 160; p.nwaves = 8; p.kb = 64;
  p.stages = az_opt(AZ_OPT_LDS_EXCLUSIVE) ? 3 : 2;
  p.tiles_m = (M + 127) / 128;
  if ((long)p.tiles_m * total_tiles_n > 0x7FFFFFF0L) return AZ_ERR_ARG(71);
  p.tiles_n = (int)total_tiles_n;
  p.ksplit = 1; p.ktiles_per_split = (K + BK - 1) / BK;
  if ((long)M * lda * 2 >= 0x7FFFFFF0L) return AZ_ERR_ARG(8);
  p.lda2 = (int)(lda * 2);
  p.k_full = (K % BK) == 0;
  p.vec_epi = 1;          // the caller guarantees N % 8 == 0, ldc % 8 == 0 and 16-byte aligned C for every group
  hipStream_t st = (hipStream_t)stream;
  if (p.stages == 3) return launch_tile<A_ROW, B_NT, 128, 160, 4, 2, 3, 64, 1>(p, st);
  return launch_tile<A_ROW, B_NT, 128, 160, 4, 2, 2, 64, 1>(p, st);
}

int az_gemm_tn_grouped_bf16(const void* groups_dev, int ngroups, long total_tiles, void* stream) {
  if (!groups_dev || ngroups <= 0 || total_tiles <= 0 || total_tiles > 0x7FFFFFF0L || ((uintptr_t)groups_dev & 7)) return AZ_ERR_ARG(75);
  Params p{};
  p.tgroups = (const TnGroup*)groups_dev; p.ngroups = ngroups;
  p.bm = 128; p.bn = 128; p.nwaves = 8; p.kb = 64; p.stages = 2;
  p.tiles_m = (int)total_tiles; p.tiles_n = 1;
  p.ksplit = 1; p.accumulate = 1;
  return launch_tile<A_COL, B_NN, 128, 128, 4, 2, 2, 64, 3>(p, (hipStream_t)stream);
}

int az_gemm_wgrad_bias_bf16(int M, int N, int K, const void* dY, long lddy, const void* X, long ldx, void* dW, long lddw,
                            int accumulate, int split_k, void* workspace, long workspace_bytes, void* bias_grad, int n_real,
                            void* stream) {
  if (!bias_grad) return AZ_ERR_ARG(22);
  return gemm_impl(1, 0, M, N, K, dY, lddy, X, ldx, dW, lddw, nullptr, nullptr, 0, 0, nullptr, 0, accumulate, split_k, workspace,
                   workspace_bytes, bias_grad, n_real, stream);
}

// mode: 0 = forward, 1 = dgrad, 2 = wgrad.  See include/aozora_hip.h for the contract.
static int conv_impl(int mode, int batch, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int ksize, int stride,
                   int pad, int cpad, const void* X, long ldx, const void* W, const void* dY, long lddy, void* out,
                   long ldo, const void* bias, const void* rowbias, long ld_rowbias, const void* residual, long ldr,
                   int accumulate, int split_k, void* workspace, long workspace_bytes, void* bias_grad, void* seg_grad,
                   void* stream) {
  const int ups = (mode >> 4) & 1;       // mode | 16: X is the half-resolution input of a nearest-2x upsample (forward / weight gradient)
  mode &= 15;
  if (ups && ((mode != 0 && mode != 2) || ksize != 3 || stride != 1 || (Hin & 1) || (Win & 1))) return AZ_ERR_ARG(23);
  if (ksize != 1 && ksize != 3) return AZ_ERR_ARG(10);
  if (stride != 1 && stride != 2) return AZ_ERR_ARG(11);
  if ((Cin & 7)) return AZ_ERR_ARG(12);
  Params p{};
  p.g = Geom{Hin, Win, Cin, Hout, Wout, Cout, stride, pad, ksize, cpad > 0 ? cpad : Cout, ups};
  const int taps = ksize * ksize;
  p.bias = (const bf16_t*)bias; p.accumulate = accumulate; p.ws = (float*)workspace;
  p.rowbias = (const bf16_t*)rowbias; p.ld_rb = ld_rowbias; p.R = (const bf16_t*)residual; p.ldr = ldr;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (mode == 0) {          // Y[pix][co] = im2col(X) . W^T
    p.A = (const bf16_t*)X; p.lda = ldx; p.B = (const bf16_t*)W; p.ldb = (long)taps * Cin;
    p.M = batch * Hout * Wout; p.N = Cout; p.K = taps * Cin; p.C = (bf16_t*)out; p.ldc = ldo;
    p.rows_per_seg = Hout * Wout;
    p.tap_uniform = (Cin % BK) == 0;
    p.conv_fast_a = p.tap_uniform && !ups;
    if ((ldx & 7)) return AZ_ERR_ARG(13);
    choose_split(p, 1, 0, false, true);
    rc = launch<A_CONV, B_NT>(p, st);
  } else if (mode == 1) {   // dX[pix][ci] = gatherT(dY) . W   (k = (tap, co<cpad))
    if (ksize != 3) return AZ_ERR_ARG(14);
    if ((p.g.cpad & 7) || (lddy & 7)) return AZ_ERR_ARG(15);
    p.A = (const bf16_t*)dY; p.lda = lddy; p.B = (const bf16_t*)W; p.ldb = 0;
    p.M = batch * Hin * Win; p.N = Cin; p.K = 9 * p.g.cpad; p.C = (bf16_t*)out; p.ldc = ldo;
    p.rows_per_seg = Hin * Win;
    p.tap_uniform = (p.g.cpad % BK) == 0;
    p.conv_fast_a = p.tap_uniform && stride == 1;
    choose_split(p, 1, 0);
    rc = launch<A_CONVT, B_CONVDG>(p, st);
  } else if (mode == 3) {   // dgrad with pre-transposed weights W'[ci][tap][co]: both operands k-contiguous (NT form)
    if (ksize != 3 || (Cout & 7)) return AZ_ERR_ARG(19);
    if ((lddy & 7)) return AZ_ERR_ARG(15);
    p.g.cpad = Cout;
    p.A = (const bf16_t*)dY; p.lda = lddy; p.B = (const bf16_t*)W; p.ldb = 9L * Cout;
    p.M = batch * Hin * Win; p.N = Cin; p.K = 9 * Cout; p.C = (bf16_t*)out; p.ldc = ldo;
    p.rows_per_seg = Hin * Win;
    p.tap_uniform = (Cout % BK) == 0;
    p.conv_fast_a = p.tap_uniform && stride == 1;
    choose_split(p, 1, 0, false, true);
    rc = launch<A_CONVT, B_NT>(p, st);
  } else if (mode == 2) {   // dW[co][(tap,ci)] = dY^T . im2col(X)     (k = output pixel)
    if ((lddy & 7) || (ldx & 7)) return AZ_ERR_ARG(16);
    p.A = (const bf16_t*)dY; p.lda = lddy; p.B = (const bf16_t*)X; p.ldb = ldx;
    p.M = Cout; p.N = taps * Cin; p.K = batch * Hout * Wout; p.C = (bf16_t*)out; p.ldc = ldo;
    if (rowbias || residual) return AZ_ERR_ARG(17);
    // stride 1 on same-size grids: source pixel = output pixel + tap displacement; rows / columns by multiply-high, exact while
    // pixel index * divisor < 2^32
    // (a 1-wide / 1-high grid has no multiply-high reciprocal: floor(2^32 / 1) + 1 wraps to 1 -- the general gather serves it)
    p.conv_fast_b = stride == 1 && !ups && Hout == Hin && Wout == Win && Win > 1 && Hin > 1 && (long)p.K * Win < (1L << 32) && (long)p.K * Hin < (1L << 32);
    p.magic_w = (unsigned)((1UL << 32) / (unsigned long)Win) + 1u;
    p.magic_h = (unsigned)((1UL << 32) / (unsigned long)Hin) + 1u;
    carve_tickets(p, workspace, workspace_bytes);
    if (bias_grad || seg_grad) {
      const int nseg = seg_grad ? batch : 1;
      int rc0 = carve_colsum(p, workspace, workspace_bytes, nseg, seg_grad ? Hout * Wout : p.K);
      if (rc0) return rc0;
      p.cs_bias = (bf16_t*)bias_grad; p.cs_n_real = Cout; p.cs_seg_out = (bf16_t*)seg_grad;
    }
    choose_split(p, split_k, workspace_bytes, true);
    if ((long)p.tiles_m * p.tiles_n > TICKET_BYTES / 4 || (long)p.ksplit * p.M * p.N * 4 >= 0x7FFFFFF0L) p.tickets = nullptr;
    if (p.tickets) AZ_HIP(hipMemsetAsync(p.tickets, 0, (size_t)p.tiles_m * p.tiles_n * 4, st));
    rc = launch<A_COL, B_CONVWG>(p, st);
  } else {
    return AZ_ERR_ARG(18);
  }
  if (rc) return rc;
  return finish_product(p, st, seg_grad, bias_grad, Cout);
}

int az_conv2d_bf16(int mode, int batch, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int ksize, int stride,
                   int pad, int cpad, const void* X, long ldx, const void* W, const void* dY, long lddy, void* out,
                   long ldo, const void* bias, const void* rowbias, long ld_rowbias, const void* residual, long ldr,
                   int accumulate, int split_k, void* workspace, long workspace_bytes, void* stream) {
  return conv_impl(mode, batch, Hin, Win, Cin, Hout, Wout, Cout, ksize, stride, pad, cpad, X, ldx, W, dY, lddy, out, ldo, bias, rowbias,
                   ld_rowbias, residual, ldr, accumulate, split_k, workspace, workspace_bytes, nullptr, nullptr, stream);
}

int az_conv2d_wgrad_bias_bf16(int batch, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int ksize, int stride, int pad,
                              const void* X, long ldx, const void* dY, long lddy, void* dW, int accumulate, int split_k,
                              void* workspace, long workspace_bytes, void* bias_grad, void* seg_grad, void* stream) {
  if (!bias_grad && !seg_grad) return AZ_ERR_ARG(22);
  return conv_impl(2 | (ksize & 16), batch, Hin, Win, Cin, Hout, Wout, Cout, ksize & 15, stride, pad, 0, X, ldx, nullptr, dY, lddy, dW,
                   (long)(ksize & 15) * (ksize & 15) * Cin, nullptr, nullptr, 0, nullptr, 0, accumulate, split_k, workspace, workspace_bytes,
                   bias_grad, seg_grad, stream);
}

}  // extern "C"
