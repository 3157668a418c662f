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

// =============================== forward ======================================================
// FULL: Tq % 128 == 0 and Tk % 64 == 0 (every self-attention of the UNet): no row / key range tests, no half-tile skips.  The
// general form's key mask was if-converted into 32 compares + 32 selects per key tile and its skippable second half kept the
// score accumulators zero-initialised by 32 moves -- 110 of the 283 vector instructions of a key tile, in a loop that is bound by
// the vector ALU (20 MFMAs = 640 cycles against ~1500 cycles of VALU issue per wave and tile).
template <bool FULL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                          AttnOut O, float* __restrict__ lse2) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // K0 V0 K1 V1
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
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

// =============================== delta = rowsum(dO * O) ======================================
__global__ void attn_delta_kernel(int heads, int Tq, AttnPtr O, AttnPtr dO, float* __restrict__ delta, int batch) {
  long n = (long)batch * Tq * heads;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int h = (int)(i % heads); long bq = i / heads; int q = (int)(bq % Tq); int b = (int)(bq / Tq);
    const bf16_t* op = O.p + b * O.sb + (long)q * O.ld + h * D;
    const bf16_t* dp = dO.p + b * dO.sb + (long)q * dO.ld + h * D;
    float s = 0.f;
#pragma unroll
    for (int cidx = 0; cidx < 8; ++cidx) {
      uint4 a = *reinterpret_cast<const uint4*>(op + cidx * 8), d = *reinterpret_cast<const uint4*>(dp + cidx * 8);
      const uint32_t* aw = reinterpret_cast<const uint32_t*>(&a); const uint32_t* dw = reinterpret_cast<const uint32_t*>(&d);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s += __uint_as_float(aw[e] << 16) * __uint_as_float(dw[e] << 16);
        s += __uint_as_float(aw[e] & 0xFFFF0000u) * __uint_as_float(dw[e] & 0xFFFF0000u);
      }
    }
    delta[((long)(b * heads + h)) * Tq + q] = s;
  }
}

// =============================== backward: dQ ================================================
// FUSE_DELTA: delta = rowsum(dO * O) is computed here from the wave's resident dO fragments (and written out for the
// dK/dV kernel) instead of by a separate pass over O and dO.
template <bool FUSE_DELTA, bool FULL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                             AttnPtr dO, AttnPtr O, const float* __restrict__ lse2,
                                                             float* __restrict__ delta, AttnOut dQ) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
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
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(int heads, int Tq, int Tk, float scale, AttnPtr Q, AttnPtr K, AttnPtr V,
                                                              AttnPtr dO, const float* __restrict__ lse2,
                                                              const float* __restrict__ delta, AttnOut dK, AttnOut dV,
                                                              int tiles_per_split, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES + 2 * 2 * TILE * 4];   // Q0 dO0 Q1 dO1, lse/delta x2
  float* stat = reinterpret_cast<float*>(smem + 4 * TILE_BYTES);   // [buf][2][64]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const bf16_t* Qb = Q.p + b * Q.sb + h * D;
  const bf16_t* dOb = dO.p + b * dO.sb + h * D;
  const float c = scale * LOG2E;
  const int key = k0 + (lane & 31);

  bf16x8 kf[4], vf[4];
  load_row_frags(K.p + b * K.sb + h * D, K.ld, key, Tk, lane, kf);
  load_row_frags(V.p + b * V.sb + h * D, V.ld, key, Tk, lane, vf);

  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  const int ntiles_all = (Tq + TILE - 1) / TILE;
  const int qt_begin = blockIdx.z * tiles_per_split;
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
  if (gridDim.z == 1) {
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
    const int kpad = gridDim.x * 128;
    float* base = part + (((long)blockIdx.z * gridDim.y + bh) * kpad) * 128;
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
  dim3 grid((Tq + 127) / 128, batch * heads);
  if ((Tq % 128) == 0 && (Tk % TILE) == 0)
    az_launch(attn_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, heads, Tq, Tk, scale,
                       AttnPtr{(const bf16_t*)Q, ldq, sq}, AttnPtr{(const bf16_t*)K, ldk, sk}, AttnPtr{(const bf16_t*)V, ldv, sv},
                       AttnOut{(bf16_t*)O, ldo, so}, (float*)lse);
  else
    az_launch(attn_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, heads, Tq, Tk, scale,
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
  int g = (int)((n + 255) / 256); if (g > 4096) g = 4096;
  if ((parts & 1) && !(parts & 2)) {
    az_launch(attn_delta_kernel, dim3(g), dim3(256), 0, st, heads, Tq, o, d_o, (float*)delta, batch);
    AZ_CHECK_LAUNCH();
  }
  if (parts & 2) {
    const bool full = (Tq % 128) == 0 && (Tk % TILE) == 0;
#define AZ_DQ(FD, FL) az_launch((attn_bwd_dq_kernel<FD, FL>), dim3((Tq + 127) / 128, batch * heads), dim3(256), 0, st, heads, Tq, Tk, scale, \
                                         q, k, v, d_o, o, (const float*)lse, (float*)delta, AttnOut{(bf16_t*)dQ, lddq, sdq})
    if (parts & 1) { if (full) AZ_DQ(true, true); else AZ_DQ(true, false); }      // delta rides on the dQ kernel's resident dO fragments
    else { if (full) AZ_DQ(false, true); else AZ_DQ(false, false); }
#undef AZ_DQ
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
  az_launch(attn_bwd_dkv_kernel, dim3(kblocks, BH, nsplit), dim3(256), 0, st, heads, Tq, Tk, scale, q, k, v, d_o,
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
