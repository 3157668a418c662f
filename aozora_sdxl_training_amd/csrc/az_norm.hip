// GroupNorm (+SiLU) and LayerNorm, forward and backward, on NHWC / [rows][C] bf16 with fp32
// statistics (diffusers ResnetBlock2D.norm1/norm2, Transformer2DModel.norm,
// BasicTransformerBlock.norm1-3, conv_norm_out -- executed at train.py:2760 fwd / 2765 bwd).
// HBM-bound: every pass reads rows fully coalesced (thread = fixed 8-channel chunk, 16 B loads).
#include "az_common.h"
#include "aozora_hip.h"

namespace {

constexpr int GN_MAX_THREADS = 256;

struct GnGeom {
  int B, HW, C, G, cpg, cchunks, py, rows_per_chunk, nchunk;
  int stat_bf16;     // the backward reads mean / rstd rounded to bf16 (option NORM_STAT_BF16, below)
};

// Statistics as the reference's BACKWARD sees them.  Under train.py:273 (bf16 autocast, bf16 parameters) torch's native_group_norm /
// native_layer_norm compute with fp32 statistics but RETURN mean and rstd in the input's dtype -- bf16 -- and autograd saves those
// for the backward (checked on the CPU build of torch 2.10: `_saved_result1/2.dtype == torch.bfloat16`).  Every (sample, group) of
// a GroupNorm backward therefore carries a coherent scale error of up to 2^-9 of random sign; over a layer's 32 groups that is
// +-3e-4 on the data gradient, and the layers' errors walk randomly along the backward chain: the reference's gradient norm sits
// up to ~1e-3 away from the same dataflow with exact statistics, sample by sample in either direction (round 5: tools/act_noise.py,
// profiles/r05_act_noise.txt -- at conv_norm_out the reference's data gradient is +3.7e-4 above what its fp32 rstd implies).
// Parity means following it: the forward normalises with the fp32 statistics, the backward kernels read them rounded.
// (The reference's other roundings around GroupNorm -- its bf16 output that SiLU reads, SiLU's bf16 gradient -- were put where torch
// has them as well, built and measured: the gradient norm does not move (1024^2: 6.02e-4 -> 5.93e-4 from the reference's) and the
// loss moves AWAY from it (1.6e-5 -> 8.7e-5): incoherent noise, not followed.  profiles/r05_parity_localisation.md)
__device__ __forceinline__ float stat_round(float v, int on) { return on ? bf2f(f2bf(v)) : v; }

// rpt: rows a thread covers per chunk at least.  0 / 1: the option of the forward / backward passes (GN_RPT, GN_RPT_BWD), clamped to
// [4, 64]; 4 is also what az_gn_scratch_floats sizes the partial sums for, so no option value can outgrow a caller's scratch.
constexpr int GN_RPT_MIN = 4;
__host__ GnGeom gn_geom(int B, int HW, int C, int G, int pass = -1) {
  GnGeom g;
  g.B = B; g.HW = HW; g.C = C; g.G = G; g.cpg = C / G; g.cchunks = C / 8;
  g.py = GN_MAX_THREADS / g.cchunks; if (g.py < 1) g.py = 1;
  int want = (HW + 1023) / 1024;               // <= 1024 chunks per sample, >= 4 rows per thread
  int rpt = pass < 0 ? GN_RPT_MIN : az_opt(pass == 2 ? AZ_OPT_GN_RPT_APPLY : pass ? AZ_OPT_GN_RPT_BWD : AZ_OPT_GN_RPT);
  if (rpt < GN_RPT_MIN) rpt = GN_RPT_MIN; if (rpt > 64) rpt = 64;
  if (want < rpt * g.py) want = rpt * g.py;
  g.rows_per_chunk = ((want + g.py - 1) / g.py) * g.py;
  g.nchunk = (HW + g.rows_per_chunk - 1) / g.rows_per_chunk;
  g.stat_bf16 = az_opt(AZ_OPT_NORM_STAT_BF16) != 0;
  return g;
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 u; u.x = pack2bf(f[0], f[1]); u.y = pack2bf(f[2], f[3]); u.z = pack2bf(f[4], f[5]); u.w = pack2bf(f[6], f[7]);
  return u;
}

// group of each of a thread's 8 consecutive channels c0 .. c0+7 with ONE integer division (cpg >= 8: the chunk meets at most two
// groups); narrower groups take the per-channel division
__device__ __forceinline__ void chunk_groups(int c0, int cpg, int (&grp)[8]) {
  if (cpg >= 8) {
    const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) grp[e] = e < split ? g0 : g0 + 1;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) grp[e] = (c0 + e) / cpg;
  }
}

// ---------------- GroupNorm forward -----------------------------------------------------------
// stage 1: partial[b][chunk][g][2] = (sum, sumsq) over the chunk's rows and the group's channels
__global__ void gn_partial_kernel(GnGeom g, const bf16_t* __restrict__ x, long ldx, float* __restrict__ partial) {
  extern __shared__ float sh[];  // [py][C][2]
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * g.rows_per_chunk;
  int r1 = r0 + g.rows_per_chunk; if (r1 > g.HW) r1 = g.HW;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  if (tx < g.cchunks) {
    const bf16_t* base = x + ((long)b * g.HW) * ldx + tx * 8;
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += g.py) {
      float f[8]; unpack8(*reinterpret_cast<const uint4*>(base + (long)r * ldx), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[((ty * g.C) + tx * 8 + e) * 2] = s[e]; sh[((ty * g.C) + tx * 8 + e) * 2 + 1] = q[e]; }
  }
  __syncthreads();
  // the block's (row slice, channel) sums -> one (sum, sumsq) per group: L lanes per group (a power of two, contiguous in a wave) stride
  // over the group's py x cpg entries and meet through shuffles.  (One THREAD per group walked them serially until round 5: 250
  // dependent LDS round trips at C = 320 while the other ~970 threads of the block waited -- a quarter of this kernel.)
  const int tid = ty * blockDim.x + tx, nth = blockDim.x * blockDim.y;
  int L = 64;
  while (L > 1 && L * g.G > nth) L >>= 1;
  const int n = g.py * g.cpg;
  for (int grp = tid / L; grp < g.G; grp += nth / L) {      // (nth / L >= G: one pass; written as a loop for tiny blocks)
    const int l = tid & (L - 1);
    float ss = 0.f, qq = 0.f;
    for (int i = l; i < n; i += L) {
      const int y = i / g.cpg, c = grp * g.cpg + (i - y * g.cpg);
      ss += sh[(y * g.C + c) * 2]; qq += sh[(y * g.C + c) * 2 + 1];
    }
    for (int o_ = L >> 1; o_ > 0; o_ >>= 1) { ss += __shfl_xor(ss, o_, 64); qq += __shfl_xor(qq, o_, 64); }
    if (l == 0) {
      float* o = partial + (((long)b * g.nchunk + chunk) * g.G + grp) * 2;
      o[0] = ss; o[1] = qq;
    }
  }
}

__global__ void gn_finalize_kernel(GnGeom g, float eps, const float* __restrict__ partial, float* __restrict__ stats) {
  const int i = blockIdx.x;            // (b, group); one wave
  const int b = i / g.G, grp = i - b * g.G;
  double s = 0.0, q = 0.0;
  for (int c = threadIdx.x; c < g.nchunk; c += 64) {
    const float2 p = *reinterpret_cast<const float2*>(partial + (((long)b * g.nchunk + c) * g.G + grp) * 2);
    s += (double)p.x; q += (double)p.y;
  }
  // butterfly over the wave in double precision (the serial sum of 64 LDS slots by lane 0 was a third of this kernel's 6.6 us)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if (threadIdx.x == 0) {
    double n = (double)g.HW * g.cpg;
    double mean = s / n;
    double var = q / n - mean * mean; if (var < 0.0) var = 0.0;
    stats[i * 2] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <bool SILU>
__global__ void gn_apply_kernel(GnGeom g, const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ gamma,
                                const bf16_t* __restrict__ beta, const float* __restrict__ stats, bf16_t* __restrict__ y, long ldy) {
  const int tx = threadIdx.x, ty = threadIdx.y;
  if (tx >= g.cchunks) return;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * g.rows_per_chunk;
  int r1 = r0 + g.rows_per_chunk; if (r1 > g.HW) r1 = g.HW;
  float sc[8], sf[8];
  {
    int grp[8]; chunk_groups(tx * 8, g.cpg, grp);
    float ga[8], be[8];
    unpack8(*reinterpret_cast<const uint4*>(gamma + tx * 8), ga);
    unpack8(*reinterpret_cast<const uint4*>(beta + tx * 8), be);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 st = *reinterpret_cast<const float2*>(stats + (b * g.G + grp[e]) * 2);
      sc[e] = st.y * ga[e]; sf[e] = be[e] - st.x * st.y * ga[e];
    }
  }
  const bf16_t* xb = x + ((long)b * g.HW) * ldx + tx * 8;
  bf16_t* yb = y + ((long)b * g.HW) * ldy + tx * 8;
#pragma unroll 4
  for (int r = r0 + ty; r < r1; r += g.py) {
    float f[8]; unpack8(*reinterpret_cast<const uint4*>(xb + (long)r * ldx), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float z = f[e] * sc[e] + sf[e];
      f[e] = SILU ? silu_f(z) : z;
    }
    *reinterpret_cast<uint4*>(yb + (long)r * ldy) = pack8(f);
  }
}

// ---------------- GroupNorm backward ----------------------------------------------------------
// stage 1: partial[b][chunk][c][2] = (sum dz, sum dz*xhat) over the chunk's rows
template <bool SILU>
__global__ void gn_bwd_partial_kernel(GnGeom g, const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ gamma,
                                      const bf16_t* __restrict__ beta, const float* __restrict__ stats,
                                      const bf16_t* __restrict__ dy, long lddy, float* __restrict__ partial) {
  extern __shared__ float sh[];  // [py][C][2]
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * g.rows_per_chunk;
  int r1 = r0 + g.rows_per_chunk; if (r1 > g.HW) r1 = g.HW;
  if (tx < g.cchunks) {
    float mean[8], rstd[8], ga[8], be[8], a[8], bb[8], zs[8], zo[8];
    int grp[8]; chunk_groups(tx * 8, g.cpg, grp);
    unpack8(*reinterpret_cast<const uint4*>(gamma + tx * 8), ga);
    unpack8(*reinterpret_cast<const uint4*>(beta + tx * 8), be);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 st = *reinterpret_cast<const float2*>(stats + (b * g.G + grp[e]) * 2);
      mean[e] = stat_round(st.x, g.stat_bf16); rstd[e] = stat_round(st.y, g.stat_bf16); a[e] = 0.f; bb[e] = 0.f;
      // SiLU' is evaluated where the FORWARD evaluated SiLU: at z formed from the fp32 statistics (torch's SiLU backward reads the
      // saved GroupNorm output); only GroupNorm's own backward formula sees the rounded pair
      zs[e] = st.y * ga[e]; zo[e] = be[e] - st.x * st.y * ga[e];
    }
    const bf16_t* xb = x + ((long)b * g.HW) * ldx + tx * 8;
    const bf16_t* db = dy + ((long)b * g.HW) * lddy + tx * 8;
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += g.py) {
      float f[8], d[8];
      unpack8(*reinterpret_cast<const uint4*>(xb + (long)r * ldx), f);
      unpack8(*reinterpret_cast<const uint4*>(db + (long)r * lddy), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xh = (f[e] - mean[e]) * rstd[e];
        float dz = d[e];
        if (SILU) dz *= dsilu_f(f[e] * zs[e] + zo[e]);
        a[e] += dz; bb[e] += dz * xh;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[((ty * g.C) + tx * 8 + e) * 2] = a[e]; sh[((ty * g.C) + tx * 8 + e) * 2 + 1] = bb[e]; }
  }
  __syncthreads();
  const int tid = ty * blockDim.x + tx, nth = blockDim.x * blockDim.y;
  for (int c = tid; c < g.C; c += nth) {
    float a = 0.f, bsum = 0.f;
    for (int y = 0; y < g.py; ++y) { a += sh[(y * g.C + c) * 2]; bsum += sh[(y * g.C + c) * 2 + 1]; }
    float* o = partial + (((long)b * g.nchunk + chunk) * g.C + c) * 2;
    o[0] = a; o[1] = bsum;
  }
}

// stage 2: block per (group, b), 64 channel lanes x 16 chunk slices: chan[b][c][2] = (sum dz, sum dz*xhat),
// gsum[b][g][2] = (s1, s2) = gamma-weighted group sums.  A slice's partial sums are fetched four chunks at a time (independent loads,
// added in chunk order: the one-load-per-iteration loop paid a dependent L2 round trip per chunk).  The parameter gradients
// dgamma / dbeta += sum_b chan[b][c] are formed by ONE block of the apply kernel (stage 3) since round 5 -- they were a launch of
// their own, 46 times per micro-step on the data-gradient chain; gn_bwd_param_kernel remains for calls without a data gradient.
__global__ void gn_bwd_finalize_kernel(GnGeom g, const bf16_t* __restrict__ gamma, const float* __restrict__ partial,
                                       float* __restrict__ chan, float* __restrict__ gsum) {
  __shared__ float red[16][128][2];
  const int grp = blockIdx.x, b = blockIdx.y, lx = threadIdx.x, sy = threadIdx.y;
  for (int lc = lx; lc < g.cpg; lc += 64) {
    const int c = grp * g.cpg + lc;
    const float* p = partial + ((long)b * g.nchunk * g.C + c) * 2;
    const long stride = (long)g.C * 2;
    float a = 0.f, bs = 0.f;
    int ch = sy;
    for (; ch + 48 < g.nchunk; ch += 64) {
      const float2 v0 = *reinterpret_cast<const float2*>(p + (long)ch * stride), v1 = *reinterpret_cast<const float2*>(p + (long)(ch + 16) * stride);
      const float2 v2 = *reinterpret_cast<const float2*>(p + (long)(ch + 32) * stride), v3 = *reinterpret_cast<const float2*>(p + (long)(ch + 48) * stride);
      a += v0.x; bs += v0.y; a += v1.x; bs += v1.y; a += v2.x; bs += v2.y; a += v3.x; bs += v3.y;
    }
    for (; ch < g.nchunk; ch += 16) { const float2 v = *reinterpret_cast<const float2*>(p + (long)ch * stride); a += v.x; bs += v.y; }
    red[sy][lc][0] = a; red[sy][lc][1] = bs;
  }
  __syncthreads();
  if (sy == 0) {
    for (int lc = lx; lc < g.cpg; lc += 64) {
      float a = 0.f, bs = 0.f;
      for (int k = 0; k < 16; ++k) { a += red[k][lc][0]; bs += red[k][lc][1]; }
      const int c = grp * g.cpg + lc;
      chan[((long)b * g.C + c) * 2] = a; chan[((long)b * g.C + c) * 2 + 1] = bs;
      const float ga = bf2f(gamma[c]);
      red[0][lc][0] = a * ga; red[0][lc][1] = bs * ga;
    }
  }
  __syncthreads();
  if (sy == 0) {            // the first wave: lane sums of the group's (<= 128) channels, then a butterfly (was a serial loop of one lane)
    float s1 = 0.f, s2 = 0.f;
    for (int i = lx; i < g.cpg; i += 64) { s1 += red[0][i][0]; s2 += red[0][i][1]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lx == 0) { gsum[(b * g.G + grp) * 2] = s1; gsum[(b * g.G + grp) * 2 + 1] = s2; }
  }
}
__global__ void gn_bwd_param_kernel(GnGeom g, const float* __restrict__ chan, bf16_t* dgamma, bf16_t* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= g.C) return;
  float a = 0.f, bs = 0.f;
  for (int b = 0; b < g.B; ++b) { a += chan[((long)b * g.C + c) * 2]; bs += chan[((long)b * g.C + c) * 2 + 1]; }
  if (dbeta) dbeta[c] = f2bf(bf2f(dbeta[c]) + a);
  if (dgamma) dgamma[c] = f2bf(bf2f(dgamma[c]) + bs);
}

template <bool SILU>
__global__ void gn_bwd_apply_kernel(GnGeom g, const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ gamma,
                                    const bf16_t* __restrict__ beta, const float* __restrict__ stats,
                                    const float* __restrict__ gsum, const bf16_t* __restrict__ dy, long lddy,
                                    bf16_t* dx, long lddx, const bf16_t* dadd, long ldadd,      // dx = (dadd ? dadd : 0) + gradient; dadd may be dx itself
                                    const float* __restrict__ chan, bf16_t* dgamma, bf16_t* dbeta) {
  const bool accumulate = dadd != nullptr;
  const int tx = threadIdx.x, ty = threadIdx.y;
  if (tx >= g.cchunks) return;
  const int b = blockIdx.y, chunk = blockIdx.x;
  if (chan && b == 0 && chunk == 0 && ty == 0) {      // the parameter gradients (gn_bwd_param_kernel's arithmetic: sums over the samples in order)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = tx * 8 + e;
      float a = 0.f, bs = 0.f;
      for (int bb = 0; bb < g.B; ++bb) { a += chan[((long)bb * g.C + c) * 2]; bs += chan[((long)bb * g.C + c) * 2 + 1]; }
      if (dbeta) dbeta[c] = f2bf(bf2f(dbeta[c]) + a);
      if (dgamma) dgamma[c] = f2bf(bf2f(dgamma[c]) + bs);
    }
  }
  const int r0 = chunk * g.rows_per_chunk;
  int r1 = r0 + g.rows_per_chunk; if (r1 > g.HW) r1 = g.HW;
  const float inv_n = 1.0f / ((float)g.HW * (float)g.cpg);
  float mean[8], rstd[8], ga[8], be[8], k1[8], k2[8], zs[8], zo[8];
  int grp[8]; chunk_groups(tx * 8, g.cpg, grp);
  unpack8(*reinterpret_cast<const uint4*>(gamma + tx * 8), ga);
  unpack8(*reinterpret_cast<const uint4*>(beta + tx * 8), be);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float2 st = *reinterpret_cast<const float2*>(stats + (b * g.G + grp[e]) * 2);
    const float2 gs = *reinterpret_cast<const float2*>(gsum + (b * g.G + grp[e]) * 2);
    mean[e] = stat_round(st.x, g.stat_bf16); rstd[e] = stat_round(st.y, g.stat_bf16); k1[e] = gs.x * inv_n; k2[e] = gs.y * inv_n;
    zs[e] = st.y * ga[e]; zo[e] = be[e] - st.x * st.y * ga[e];
  }
  const bf16_t* xb = x + ((long)b * g.HW) * ldx + tx * 8;
  const bf16_t* db = dy + ((long)b * g.HW) * lddy + tx * 8;
  bf16_t* ob = dx + ((long)b * g.HW) * lddx + tx * 8;
  const bf16_t* ab = accumulate ? dadd + ((long)b * g.HW) * ldadd + tx * 8 : nullptr;
#pragma unroll 2
  for (int r = r0 + ty; r < r1; r += g.py) {
    float f[8], d[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(xb + (long)r * ldx), f);
    unpack8(*reinterpret_cast<const uint4*>(db + (long)r * lddy), d);
    if (accumulate) unpack8(*reinterpret_cast<const uint4*>(ab + (long)r * ldadd), o);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xh = (f[e] - mean[e]) * rstd[e];
      float dz = d[e];
      if (SILU) dz *= dsilu_f(f[e] * zs[e] + zo[e]);
      float v = rstd[e] * (dz * ga[e] - k1[e] - xh * k2[e]);
      o[e] = accumulate ? o[e] + v : v;
    }
    *reinterpret_cast<uint4*>(ob + (long)r * lddx) = pack8(o);
  }
}

// ---------------- LayerNorm -------------------------------------------------------------------
constexpr int LN_MAXCH = 4;  // chunks of 8 per lane -> C <= 2048

__global__ void ln_fwd_kernel(int M, int C, float eps, const bf16_t* __restrict__ x, long ldx,
                              const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                              bf16_t* __restrict__ y, long ldy, float* __restrict__ stats, int stat_bf16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cch = C >> 3;
  float v[LN_MAXCH][8];
  uint4 ug[LN_MAXCH], ub[LN_MAXCH];      // gamma / beta fetched with the row, not after the two reductions
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
    if (cc < cch) {
      const uint4 ux = *reinterpret_cast<const uint4*>(x + (long)row * ldx + cc * 8);
      ug[i] = *reinterpret_cast<const uint4*>(gamma + cc * 8);
      ub[i] = *reinterpret_cast<const uint4*>(beta + cc * 8);
      unpack8(ux, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float invC = 1.0f / (float)C;
  const float mean = wave_sum(s) * invC;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
    if (cc < cch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(fmaf(wave_sum(q), invC, eps));
  // the row is normalised with the fp32 statistics; what is SAVED is what the backward reads (stat_round above)
  if (lane == 0) { stats[row * 2] = stat_round(mean, stat_bf16); stats[row * 2 + 1] = stat_round(rstd, stat_bf16); }
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
    if (cc < cch) {
      float g8[8], b8[8], o[8];
      unpack8(ug[i], g8);
      unpack8(ub[i], b8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g8[e] + b8[e];
      *reinterpret_cast<uint4*>(y + (long)row * ldy + cc * 8) = pack8(o);
    }
  }
}

// LayerNorm backward, two kernels:
//  (1) ln_bwd_dx_kernel: one wave per row, NCH 16-byte chunks per lane held in registers (no spills)
//  (2) ln_bwd_param_kernel: thread = fixed 8-channel chunk, block strides over rows, partial dgamma/dbeta
//      per block -> partial[blk][C][2]; colpair_finalize_kernel sums the partials in a fixed order.
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(int M, int C, const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ gamma,
                                 const float* __restrict__ stats, const bf16_t* __restrict__ dy, long lddy,
                                 bf16_t* dx, long lddx, const bf16_t* dadd, long ldadd) {
  const bool accumulate = dadd != nullptr;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cch = C >> 3;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float xh[NCH][8], dg[NCH][8];
  uint4 prev[NCH];             // accumulate target, fetched with the operands (not after the reductions: the loads would
                               // otherwise start a second exposed memory round trip per row)
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int cc = lane + 64 * i;
    if (cc < cch) {
      const uint4 ux = *reinterpret_cast<const uint4*>(x + (long)row * ldx + cc * 8);
      const uint4 ud = *reinterpret_cast<const uint4*>(dy + (long)row * lddy + cc * 8);
      const uint4 ug = *reinterpret_cast<const uint4*>(gamma + cc * 8);
      if (accumulate) prev[i] = *reinterpret_cast<const uint4*>(dadd + (long)row * ldadd + cc * 8);
      float f[8], d[8], g8[8];
      unpack8(ux, f); unpack8(ud, d); unpack8(ug, g8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[i][e] = (f[e] - mean) * rstd;
        dg[i][e] = d[e] * g8[e];
        c1 += dg[i][e]; c2 += dg[i][e] * xh[i][e];
      }
    }
  }
  const float invC = 1.0f / (float)C;
  c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int cc = lane + 64 * i;
    if (cc < cch) {
      float o[8];
      bf16_t* op = dx + (long)row * lddx + cc * 8;
      if (accumulate) unpack8(prev[i], o);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float vv = rstd * (dg[i][e] - c1 - xh[i][e] * c2);
        o[e] = accumulate ? o[e] + vv : vv;
      }
      *reinterpret_cast<uint4*>(op) = pack8(o);
    }
  }
}

// (1)+(2) in one pass when both the data gradient and the parameter gradients are wanted: a block of 4 waves takes
// `rows_per_block` rows (wave w: rows r0+w, r0+w+4, ...), each lane keeps fp32 dgamma/dbeta sums of its own channels
// over the rows of its wave; the 4 waves are summed through LDS in wave order -> partial[blk][C][2].  x and dy are read
// once instead of twice.
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(int M, int C, int rows_per_block, const bf16_t* __restrict__ x, long ldx,
                                 const bf16_t* __restrict__ gamma, const float* __restrict__ stats, const bf16_t* __restrict__ dy,
                                 long lddy, bf16_t* dx, long lddx, const bf16_t* dadd, long ldadd, float* __restrict__ partial) {
  const bool accumulate = dadd != nullptr;
  extern __shared__ float sh[];   // [4 waves][64 lanes][16]: one column group of the cross-wave sum at a time
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cch = C >> 3;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  float pg[NCH][8], pb[NCH][8];
  uint4 ug[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int cc = lane + 64 * i;
    ug[i] = cc < cch ? *reinterpret_cast<const uint4*>(gamma + cc * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[i][e] = 0.f; pb[i][e] = 0.f; }
  }
  const float invC = 1.0f / (float)C;
  // Rows are software-pipelined: the 16-byte pieces of row r + 4 (x, dy, the accumulate target, the row's statistics) are requested
  // before row r is reduced, so a wave keeps two rows in flight instead of paying one exposed memory round trip per row.
  uint4 nx[NCH], nd[NCH], np[NCH];
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](int row) {
    nmean = stats[row * 2]; nrstd = stats[row * 2 + 1];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      if (cc < cch) {
        nx[i] = *reinterpret_cast<const uint4*>(x + (long)row * ldx + cc * 8);
        nd[i] = *reinterpret_cast<const uint4*>(dy + (long)row * lddy + cc * 8);
        if (accumulate) np[i] = *reinterpret_cast<const uint4*>(dadd + (long)row * ldadd + cc * 8);
      }
    }
  };
  if (r0 + w < r1) fetch(r0 + w);
  for (int row = r0 + w; row < r1; row += 4) {
    const float mean = nmean, rstd = nrstd;
    uint4 cx[NCH], cd[NCH], prev[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { cx[i] = nx[i]; cd[i] = nd[i]; if (accumulate) prev[i] = np[i]; }
    if (row + 4 < r1) fetch(row + 4);
    float xh[NCH][8], dg[NCH][8];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      if (cc < cch) {
        float f[8], d[8], g8[8];
        unpack8(cx[i], f); unpack8(cd[i], d); unpack8(ug[i], g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = (f[e] - mean) * rstd;
          pb[i][e] += d[e]; pg[i][e] += d[e] * xh[i][e];
          dg[i][e] = d[e] * g8[e];
          c1 += dg[i][e]; c2 += dg[i][e] * xh[i][e];
        }
      }
    }
    c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      if (cc < cch) {
        float o[8];
        if (accumulate) unpack8(prev[i], o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float vv = rstd * (dg[i][e] - c1 - xh[i][e] * c2);
          o[e] = accumulate ? o[e] + vv : vv;
        }
        *reinterpret_cast<uint4*>(dx + (long)row * lddx + cc * 8) = pack8(o);
      }
    }
  }
  // The four waves' sums meet in LDS one 64-chunk column group at a time: [4 waves][64 lanes][8 channels][2] floats = 16 KiB however
  // wide the row is (it was 4 * C * 2 floats = 40 KiB at C = 1280, which kept the blocks off every CU that already held two
  // weight-gradient workgroups of the other stream: 128 + 40 > 160 KiB).  Same summation order (wave 0 + 1 + 2 + 3): same bits.
  float* out = partial + (long)blockIdx.x * C * 2;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (i) __syncthreads();
    float* d = sh + ((long)w * 64 + lane) * 16;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
      *reinterpret_cast<float4*>(d + e * 2) = make_float4(pg[i][e], pb[i][e], pg[i][e + 1], pb[i][e + 1]);
    __syncthreads();
    const int l2 = threadIdx.x >> 2, q = threadIdx.x & 3;      // thread -> (lane of the group, which float4 of its 16 floats)
    const int cc = l2 + 64 * i;
    if (cc < cch) {
      float4 a = *reinterpret_cast<const float4*>(sh + (long)l2 * 16 + q * 4);
#pragma unroll
      for (int y = 1; y < 4; ++y) {
        const float4 v = *reinterpret_cast<const float4*>(sh + ((long)y * 64 + l2) * 16 + q * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      *reinterpret_cast<float4*>(out + ((long)cc * 8 + q * 2) * 2) = a;
    }
  }
}

// grid.x = row blocks; block (bx = min(cch,256) [x gridDim.y column blocks], by)
__global__ void ln_bwd_param_kernel(int M, int C, int rows_per_block, const bf16_t* __restrict__ x, long ldx,
                                    const float* __restrict__ stats, const bf16_t* __restrict__ dy, long lddy,
                                    float* __restrict__ partial) {
  extern __shared__ float sh[];   // [by][bx*8][2]
  const int cc = blockIdx.y * blockDim.x + threadIdx.x;
  const int cch = C >> 3;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  float dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; }
  if (cc < cch) {
#pragma unroll 4
    for (int r = r0 + threadIdx.y; r < r1; r += blockDim.y) {
      const float mean = stats[r * 2], rstd = stats[r * 2 + 1];
      float f[8], d[8];
      unpack8(*reinterpret_cast<const uint4*>(x + (long)r * ldx + cc * 8), f);
      unpack8(*reinterpret_cast<const uint4*>(dy + (long)r * lddy + cc * 8), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) { db[e] += d[e]; dg[e] += d[e] * (f[e] - mean) * rstd; }
    }
  }
  const int w8 = blockDim.x * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[((threadIdx.y * w8) + threadIdx.x * 8 + e) * 2] = dg[e];
    sh[((threadIdx.y * w8) + threadIdx.x * 8 + e) * 2 + 1] = db[e];
  }
  __syncthreads();
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, nth = blockDim.x * blockDim.y;
  for (int lc = tid; lc < w8; lc += nth) {
    const int c = blockIdx.y * w8 + lc;
    if (c >= C) continue;
    float a = 0.f, bsum = 0.f;
    for (int y = 0; y < (int)blockDim.y; ++y) { a += sh[(y * w8 + lc) * 2]; bsum += sh[(y * w8 + lc) * 2 + 1]; }
    partial[((long)blockIdx.x * C + c) * 2] = a;
    partial[((long)blockIdx.x * C + c) * 2 + 1] = bsum;
  }
}

// out pairs: dgamma[c] += sum_k partial[k][c][0], dbeta[c] += sum_k partial[k][c][1]; 32 columns x 16 slices per block,
// a thread owns two adjacent columns (one 16-byte load per partial row) and keeps 8 loads in flight
__device__ __forceinline__ void colpair_finalize_block(int blk, int nparts, int C, const float* __restrict__ partial, bf16_t* dgamma, bf16_t* dbeta,
                                                       float4 (*sh)[16]) {
  const int lc = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blk * 32 + lc * 2;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float* base = partial + (long)c * 2;
    int k = sl;
    for (; k + 7 * 16 < nparts; k += 8 * 16) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long)(k + u * 16) * C * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; k < nparts; k += 16) {
      const float4 v = *reinterpret_cast<const float4*>(base + (long)k * C * 2);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  sh[sl][lc] = a;
  __syncthreads();
  if (sl == 0 && c < C) {
    a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float4 v = sh[i][lc]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (dgamma) { dgamma[c] = f2bf(bf2f(dgamma[c]) + a.x); dgamma[c + 1] = f2bf(bf2f(dgamma[c + 1]) + a.z); }
    if (dbeta) { dbeta[c] = f2bf(bf2f(dbeta[c]) + a.y); dbeta[c + 1] = f2bf(bf2f(dbeta[c + 1]) + a.w); }
  }
}

__global__ __launch_bounds__(256) void colpair_finalize_kernel(int nparts, int C, const float* __restrict__ partial, bf16_t* dgamma, bf16_t* dbeta) {
  __shared__ float4 sh[16][16];
  colpair_finalize_block(blockIdx.x, nparts, C, partial, dgamma, dbeta, sh);
}

// The same for MANY LayerNorms in one launch (the gamma / beta gradients are not needed before the end of a parameter region's
// backward, so the executor parks each LayerNorm's partial sums and finishes them together: 210 finish launches per micro-step
// leave the data-gradient chain).  Job table in device memory, six int64 per job: partial, dgamma, dbeta, nparts, C, first block.
struct LnFinishJob { const float* partial; bf16_t* dgamma; bf16_t* dbeta; long nparts, C, block_start; };
__global__ __launch_bounds__(256) void colpair_finalize_multi_kernel(const LnFinishJob* __restrict__ jobs, int njobs) {
  __shared__ float4 sh[16][16];
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const LnFinishJob j = jobs[lo];
  colpair_finalize_block((int)(blockIdx.x - j.block_start), (int)j.nparts, (int)j.C, j.partial, j.dgamma, j.dbeta, sh);
}

constexpr int LN_BWD_BLOCKS = 256;
constexpr int LN_FUSED_MAX_BLOCKS = 1024;   // partial rows of the fused data + parameter gradient pass

int gn_check(int B, int HW, int C, int G, long ld) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0) return AZ_ERR_ARG(20);
  if ((C & 7) || (C % G) || (ld & 7)) return AZ_ERR_ARG(21);
  if (C / 8 > 1024 || C / G > 128) return AZ_ERR_ARG(22);
  return AZ_OK;
}

}  // namespace

extern "C" {

long az_gn_scratch_floats(int batch, int HW, int C, int G) {
  GnGeom g = gn_geom(batch, HW, C, G);
  long fwd = (long)batch * g.nchunk * G * 2;
  long bwd = (long)batch * g.nchunk * C * 2 + (long)batch * C * 2 + (long)batch * G * 2;
  return fwd > bwd ? fwd : bwd;
}

int az_groupnorm_fwd(int batch, int HW, int C, int G, float eps, int fuse_silu, const void* x, long ldx,
                     const void* gamma, const void* beta, void* y, long ldy, void* stats, void* partial, void* stream) {
  int rc = gn_check(batch, HW, C, G, ldx); if (rc) return rc;
  if ((ldy & 7) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) || ((uintptr_t)stats & 7)) return AZ_ERR_ARG(23);
  GnGeom g = gn_geom(batch, HW, C, G, 0);
  hipStream_t st = (hipStream_t)stream;
  dim3 blk(g.cchunks, g.py), grid(g.nchunk, batch);
  size_t shb = (size_t)g.py * C * 2 * sizeof(float);
  if (shb > 64 * 1024) return AZ_ERR_ARG(24);
  az_launch(gn_partial_kernel, grid, blk, shb, st, g, (const bf16_t*)x, ldx, (float*)partial);
  AZ_CHECK_LAUNCH();
  az_launch(gn_finalize_kernel, dim3(batch * G), dim3(64), 0, st, g, eps, (const float*)partial, (float*)stats);
  AZ_CHECK_LAUNCH();
  if (fuse_silu)
    az_launch(gn_apply_kernel<true>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (bf16_t*)y, ldy);
  else
    az_launch(gn_apply_kernel<false>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (bf16_t*)y, ldy);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_groupnorm_bwd_ex(int batch, int HW, int C, int G, int fuse_silu, const void* x, long ldx, const void* gamma,
                        const void* beta, const void* stats, const void* dy, long lddy, void* dx, long lddx,
                        const void* dx_add, long ld_add, void* dgamma, void* dbeta, void* partial, void* stream) {
  int rc = gn_check(batch, HW, C, G, ldx); if (rc) return rc;
  if ((lddy & 7) || (lddx & 7) || (dx_add && (ld_add & 7)) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) || ((uintptr_t)stats & 7)) return AZ_ERR_ARG(25);
  GnGeom g = gn_geom(batch, HW, C, G, 1);
  hipStream_t st = (hipStream_t)stream;
  dim3 blk(g.cchunks, g.py), grid(g.nchunk, batch);
  size_t shb = (size_t)g.py * C * 2 * sizeof(float);
  if (shb > 64 * 1024) return AZ_ERR_ARG(24);
  float* part = (float*)partial;
  float* chan = part + (long)batch * g.nchunk * C * 2;
  float* gsum = chan + (long)batch * C * 2;
  if (fuse_silu)
    az_launch(gn_bwd_partial_kernel<true>, grid, blk, shb, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (const bf16_t*)dy, lddy, part);
  else
    az_launch(gn_bwd_partial_kernel<false>, grid, blk, shb, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (const bf16_t*)dy, lddy, part);
  AZ_CHECK_LAUNCH();
  az_launch(gn_bwd_finalize_kernel, dim3(G, batch), dim3(64, 16), 0, st, g, (const bf16_t*)gamma, (const float*)part, chan, gsum);
  AZ_CHECK_LAUNCH();
  const bool params = dgamma || dbeta;
  if (params && !dx) {      // no data gradient wanted: the parameter gradients keep their own launch
    az_launch(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, st, g, (const float*)chan, (bf16_t*)dgamma, (bf16_t*)dbeta);
    AZ_CHECK_LAUNCH();
  }
  const float* chan_for_apply = params ? chan : nullptr;
  if (dx) {
    // the element-wise pass has its own row chunks (GN_RPT_APPLY): nothing ties them to the chunks of the partial sums
    g = gn_geom(batch, HW, C, G, 2);
    blk = dim3(g.cchunks, g.py); grid = dim3(g.nchunk, batch);
    if (fuse_silu)
      az_launch(gn_bwd_apply_kernel<true>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                         (const bf16_t*)beta, (const float*)stats, (const float*)gsum, (const bf16_t*)dy, lddy,
                         (bf16_t*)dx, lddx, (const bf16_t*)dx_add, ld_add, chan_for_apply, (bf16_t*)dgamma, (bf16_t*)dbeta);
    else
      az_launch(gn_bwd_apply_kernel<false>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                         (const bf16_t*)beta, (const float*)stats, (const float*)gsum, (const bf16_t*)dy, lddy,
                         (bf16_t*)dx, lddx, (const bf16_t*)dx_add, ld_add, chan_for_apply, (bf16_t*)dgamma, (bf16_t*)dbeta);
    AZ_CHECK_LAUNCH();
  }
  return AZ_OK;
}

int az_groupnorm_bwd(int batch, int HW, int C, int G, int fuse_silu, const void* x, long ldx, const void* gamma,
                     const void* beta, const void* stats, const void* dy, long lddy, void* dx, long lddx,
                     int accumulate_dx, void* dgamma, void* dbeta, void* partial, void* stream) {
  return az_groupnorm_bwd_ex(batch, HW, C, G, fuse_silu, x, ldx, gamma, beta, stats, dy, lddy, dx, lddx, accumulate_dx ? dx : nullptr, lddx,
                             dgamma, dbeta, partial, stream);
}

long az_ln_scratch_floats(int M, int C) { (void)M; return (long)LN_FUSED_MAX_BLOCKS * C * 2; }

int az_layernorm_fwd(int M, int C, float eps, const void* x, long ldx, const void* gamma, const void* beta, void* y,
                     long ldy, void* stats, void* stream) {
  if (M <= 0 || (C & 7) || C > 64 * 8 * LN_MAXCH || (ldx & 7) || (ldy & 7)) return AZ_ERR_ARG(30);
  az_launch(ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, M, C, eps, (const bf16_t*)x, ldx,
                     (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, ldy, (float*)stats, (int)(az_opt(AZ_OPT_NORM_STAT_BF16) != 0));
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

static int ln_fused_rpb(int M);
int az_layernorm_bwd_ex(int M, int C, const void* x, long ldx, const void* gamma, const void* stats, const void* dy,
                        long lddy, void* dx, long lddx, const void* dx_add, long ld_add, void* dgamma, void* dbeta, void* partial,
                        void* stream) {
  if (M <= 0 || (C & 7) || C > 64 * 8 * LN_MAXCH || (ldx & 7) || (lddy & 7) || (dx && (lddx & 7)) || (dx_add && (ld_add & 7))) return AZ_ERR_ARG(31);
  hipStream_t st = (hipStream_t)stream;
  const int cch = C / 8;
  const int nch = (cch + 63) / 64;
  dim3 g1((M + 3) / 4), b1(256);
#define LN_DX(N) az_launch(ln_bwd_dx_kernel<N>, g1, b1, 0, st, M, C, (const bf16_t*)x, ldx, (const bf16_t*)gamma, \
                                    (const float*)stats, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, (const bf16_t*)dx_add, ld_add)
  if (dx && (dgamma || dbeta)) {      // one pass over x / dy for both
    const int nblk = (M + ln_fused_rpb(M) - 1) / ln_fused_rpb(M);
    const int rpb = ((M + nblk - 1) / nblk + 3) / 4 * 4;          // the rows per block az_layernorm_bwd_partial derives from the same block count
    const size_t shb = (size_t)4 * 64 * 16 * sizeof(float);      // one column group of the cross-wave sum (ln_bwd_fused_kernel)
#define LN_FU(N) az_launch(ln_bwd_fused_kernel<N>, dim3(nblk), b1, shb, st, M, C, rpb, (const bf16_t*)x, ldx, (const bf16_t*)gamma, \
                                    (const float*)stats, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, (const bf16_t*)dx_add, ld_add, (float*)partial)
    if (nch == 1) LN_FU(1); else if (nch == 2) LN_FU(2); else if (nch == 3) LN_FU(3); else LN_FU(4);
#undef LN_FU
    AZ_CHECK_LAUNCH();
    az_launch(colpair_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, nblk, C, (const float*)partial,
                       (bf16_t*)dgamma, (bf16_t*)dbeta);
    AZ_CHECK_LAUNCH();
    return AZ_OK;
  }
  if (dx) {              // dx == NULL: parameter gradients only (issued on the parameter-gradient stream)
    if (nch == 1) LN_DX(1); else if (nch == 2) LN_DX(2); else if (nch == 3) LN_DX(3); else LN_DX(4);
    AZ_CHECK_LAUNCH();
  }
#undef LN_DX
  if (dgamma || dbeta) {
    int bx = cch < 128 ? cch : 128; int by = 256 / bx; if (by < 1) by = 1; if (by > 16) by = 16;
    int colblocks = (cch + bx - 1) / bx;
    int rpb = (M + LN_BWD_BLOCKS - 1) / LN_BWD_BLOCKS; if (rpb < by) rpb = by;
    int nblk = (M + rpb - 1) / rpb;
    size_t shb = (size_t)by * bx * 8 * 2 * sizeof(float);
    az_launch(ln_bwd_param_kernel, dim3(nblk, colblocks), dim3(bx, by), shb, st, M, C, rpb, (const bf16_t*)x, ldx,
                       (const float*)stats, (const bf16_t*)dy, lddy, (float*)partial);
    AZ_CHECK_LAUNCH();
    az_launch(colpair_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, nblk, C, (const float*)partial,
                       (bf16_t*)dgamma, (bf16_t*)dbeta);
    AZ_CHECK_LAUNCH();
  }
  return AZ_OK;
}

static int ln_fused_rpb(int M) {
  const int rpb_env = az_opt(AZ_OPT_LN_RPB);
  int rpb = rpb_env < 4 ? 4 : (rpb_env + 3) / 4 * 4;
  while ((M + rpb - 1) / rpb > LN_FUSED_MAX_BLOCKS) rpb += 4;
  return rpb;
}

int az_ln_partial_blocks(int M) { return M > 0 ? (M + ln_fused_rpb(M) - 1) / ln_fused_rpb(M) : 0; }

int az_layernorm_bwd_partial(int M, int C, const void* x, long ldx, const void* gamma, const void* stats, const void* dy,
                             long lddy, void* dx, long lddx, const void* dx_add, long ld_add, void* partial, int nblk, void* stream) {
  if (M <= 0 || (C & 7) || C > 64 * 8 * LN_MAXCH || (ldx & 7) || (lddy & 7) || !dx || (lddx & 7) || (dx_add && (ld_add & 7)) || !partial) return AZ_ERR_ARG(32);
  // The block count is the CALLER's (the capacity of `partial` in blocks, and what its finish job sums over): the rows per block
  // are derived from it, never from the LN_RPB option as it stands at launch time -- a recorded launch replayed after the option
  // changed would otherwise write past the buffer the caller sized (and the finish would read the old count).
  if (nblk <= 0 || nblk > LN_FUSED_MAX_BLOCKS) return AZ_ERR_ARG(34);
  const int rpb = ((M + nblk - 1) / nblk + 3) / 4 * 4;
  if ((M + rpb - 1) / rpb != nblk) return AZ_ERR_ARG(34);        // not a block count az_ln_partial_blocks hands out for this M
  hipStream_t st = (hipStream_t)stream;
  const int nch = (C / 8 + 63) / 64;
  const size_t shb = (size_t)4 * 64 * 16 * sizeof(float);      // one column group of the cross-wave sum (ln_bwd_fused_kernel)
#define LN_FU(N) az_launch(ln_bwd_fused_kernel<N>, dim3(nblk), dim3(256), shb, st, M, C, rpb, (const bf16_t*)x, ldx, (const bf16_t*)gamma, \
                                    (const float*)stats, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, (const bf16_t*)dx_add, ld_add, (float*)partial)
  if (nch == 1) LN_FU(1); else if (nch == 2) LN_FU(2); else if (nch == 3) LN_FU(3); else LN_FU(4);
#undef LN_FU
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_ln_param_finish_multi(const void* jobs_dev, int njobs, long nblocks, void* stream) {
  if (!jobs_dev || njobs <= 0 || nblocks <= 0 || nblocks > 0x7FFFFFF0L || ((uintptr_t)jobs_dev & 7)) return AZ_ERR_ARG(33);
  az_launch(colpair_finalize_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const LnFinishJob*)jobs_dev, njobs);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_layernorm_bwd(int M, int C, const void* x, long ldx, const void* gamma, const void* stats, const void* dy,
                     long lddy, void* dx, long lddx, int accumulate_dx, void* dgamma, void* dbeta, void* partial,
                     void* stream) {
  return az_layernorm_bwd_ex(M, C, x, ldx, gamma, stats, dy, lddy, dx, lddx, accumulate_dx ? dx : nullptr, lddx, dgamma, dbeta, partial, stream);
}

}  // extern "C"
