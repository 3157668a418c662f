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
};

__host__ GnGeom gn_geom(int B, int HW, int C, int G) {
  GnGeom g;
  g.B = B; g.HW = HW; g.C = C; g.G = G; g.cpg = C / G; g.cchunks = C / 8;
  g.py = GN_MAX_THREADS / g.cchunks; if (g.py < 1) g.py = 1;
  int want = (HW + 127) / 128;                 // <= 128 chunks per sample
  g.rows_per_chunk = ((want + g.py - 1) / g.py) * g.py;
  g.nchunk = (HW + g.rows_per_chunk - 1) / g.rows_per_chunk;
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
    for (int r = r0 + ty; r < r1; r += g.py) {
      float f[8]; unpack8(*reinterpret_cast<const uint4*>(base + (long)r * ldx), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[((ty * g.C) + tx * 8 + e) * 2] = s[e]; sh[((ty * g.C) + tx * 8 + e) * 2 + 1] = q[e]; }
  }
  __syncthreads();
  const int tid = ty * blockDim.x + tx, nth = blockDim.x * blockDim.y;
  for (int grp = tid; grp < g.G; grp += nth) {
    float ss = 0.f, qq = 0.f;
    for (int y = 0; y < g.py; ++y)
      for (int c = grp * g.cpg; c < (grp + 1) * g.cpg; ++c) { ss += sh[(y * g.C + c) * 2]; qq += sh[(y * g.C + c) * 2 + 1]; }
    float* o = partial + (((long)b * g.nchunk + chunk) * g.G + grp) * 2;
    o[0] = ss; o[1] = qq;
  }
}

__global__ void gn_finalize_kernel(GnGeom g, float eps, const float* __restrict__ partial, float* __restrict__ stats) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.B * g.G) return;
  int b = i / g.G, grp = i - b * g.G;
  double s = 0.0, q = 0.0;
  for (int c = 0; c < g.nchunk; ++c) {
    const float* p = partial + (((long)b * g.nchunk + c) * g.G + grp) * 2;
    s += (double)p[0]; q += (double)p[1];
  }
  double n = (double)g.HW * g.cpg;
  double mean = s / n;
  double var = q / n - mean * mean; if (var < 0.0) var = 0.0;
  stats[i * 2] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
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
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = tx * 8 + e; int grp = c / g.cpg;
    float mean = stats[(b * g.G + grp) * 2], rstd = stats[(b * g.G + grp) * 2 + 1];
    float ga = bf2f(gamma[c]), be = bf2f(beta[c]);
    sc[e] = rstd * ga; sf[e] = be - mean * rstd * ga;
  }
  const bf16_t* xb = x + ((long)b * g.HW) * ldx + tx * 8;
  bf16_t* yb = y + ((long)b * g.HW) * ldy + tx * 8;
  for (int r = r0 + ty; r < r1; r += g.py) {
    float f[8]; unpack8(*reinterpret_cast<const uint4*>(xb + (long)r * ldx), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { float z = f[e] * sc[e] + sf[e]; f[e] = SILU ? silu_f(z) : z; }
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
    float mean[8], rstd[8], ga[8], be[8], a[8], bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = tx * 8 + e; int grp = c / g.cpg;
      mean[e] = stats[(b * g.G + grp) * 2]; rstd[e] = stats[(b * g.G + grp) * 2 + 1];
      ga[e] = bf2f(gamma[c]); be[e] = bf2f(beta[c]); a[e] = 0.f; bb[e] = 0.f;
    }
    const bf16_t* xb = x + ((long)b * g.HW) * ldx + tx * 8;
    const bf16_t* db = dy + ((long)b * g.HW) * lddy + tx * 8;
    for (int r = r0 + ty; r < r1; r += g.py) {
      float f[8], d[8];
      unpack8(*reinterpret_cast<const uint4*>(xb + (long)r * ldx), f);
      unpack8(*reinterpret_cast<const uint4*>(db + (long)r * lddy), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xh = (f[e] - mean[e]) * rstd[e];
        float dz = d[e];
        if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
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

// stage 2: one block per group. gsum[b][g][2] = (s1, s2); dgamma/dbeta accumulate over b.
__global__ void gn_bwd_finalize_kernel(GnGeom g, const bf16_t* __restrict__ gamma, const float* __restrict__ partial,
                                       float* __restrict__ gsum, bf16_t* dgamma, bf16_t* dbeta) {
  __shared__ float red[2][128];
  const int grp = blockIdx.x, lc = threadIdx.x;   // blockDim.x = 128 >= cpg
  const int c = grp * g.cpg + lc;
  const bool act = lc < g.cpg;
  float ga = act ? bf2f(gamma[c]) : 0.f;
  float dg = 0.f, dbt = 0.f;
  for (int b = 0; b < g.B; ++b) {
    float a = 0.f, bs = 0.f;
    if (act) {
      for (int ch = 0; ch < g.nchunk; ++ch) {
        const float* p = partial + (((long)b * g.nchunk + ch) * g.C + c) * 2;
        a += p[0]; bs += p[1];
      }
    }
    dbt += a; dg += bs;
    __syncthreads();
    red[0][lc] = a * ga; red[1][lc] = bs * ga;
    __syncthreads();
    if (lc == 0) {
      float s1 = 0.f, s2 = 0.f;
      for (int i = 0; i < g.cpg; ++i) { s1 += red[0][i]; s2 += red[1][i]; }
      gsum[(b * g.G + grp) * 2] = s1; gsum[(b * g.G + grp) * 2 + 1] = s2;
    }
  }
  if (act) {
    if (dgamma) dgamma[c] = f2bf(bf2f(dgamma[c]) + dg);
    if (dbeta) dbeta[c] = f2bf(bf2f(dbeta[c]) + dbt);
  }
}

template <bool SILU>
__global__ void gn_bwd_apply_kernel(GnGeom g, const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ gamma,
                                    const bf16_t* __restrict__ beta, const float* __restrict__ stats,
                                    const float* __restrict__ gsum, const bf16_t* __restrict__ dy, long lddy,
                                    bf16_t* dx, long lddx, int accumulate) {
  const int tx = threadIdx.x, ty = threadIdx.y;
  if (tx >= g.cchunks) return;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * g.rows_per_chunk;
  int r1 = r0 + g.rows_per_chunk; if (r1 > g.HW) r1 = g.HW;
  const float inv_n = 1.0f / ((float)g.HW * (float)g.cpg);
  float mean[8], rstd[8], ga[8], be[8], k1[8], k2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = tx * 8 + e; int grp = c / g.cpg;
    mean[e] = stats[(b * g.G + grp) * 2]; rstd[e] = stats[(b * g.G + grp) * 2 + 1];
    ga[e] = bf2f(gamma[c]); be[e] = bf2f(beta[c]);
    k1[e] = gsum[(b * g.G + grp) * 2] * inv_n; k2[e] = gsum[(b * g.G + grp) * 2 + 1] * inv_n;
  }
  const bf16_t* xb = x + ((long)b * g.HW) * ldx + tx * 8;
  const bf16_t* db = dy + ((long)b * g.HW) * lddy + tx * 8;
  bf16_t* ob = dx + ((long)b * g.HW) * lddx + tx * 8;
  for (int r = r0 + ty; r < r1; r += g.py) {
    float f[8], d[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(xb + (long)r * ldx), f);
    unpack8(*reinterpret_cast<const uint4*>(db + (long)r * lddy), d);
    if (accumulate) unpack8(*reinterpret_cast<const uint4*>(ob + (long)r * lddx), o);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xh = (f[e] - mean[e]) * rstd[e];
      float dz = d[e];
      if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
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
                              bf16_t* __restrict__ y, long ldy, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cch = C >> 3;
  float v[LN_MAXCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
    if (cc < cch) {
      unpack8(*reinterpret_cast<const uint4*>(x + (long)row * ldx + cc * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
    if (cc < cch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
    if (cc < cch) {
      float g8[8], b8[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(gamma + cc * 8), g8);
      unpack8(*reinterpret_cast<const uint4*>(beta + cc * 8), b8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g8[e] + b8[e];
      *reinterpret_cast<uint4*>(y + (long)row * ldy + cc * 8) = pack8(o);
    }
  }
}

// each wave walks rows wave_id, wave_id + nwaves, ...; block-level partial dgamma/dbeta to
// partial[block][C][2]
__global__ void ln_bwd_kernel(int M, int C, const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ gamma,
                              const float* __restrict__ stats, const bf16_t* __restrict__ dy, long lddy,
                              bf16_t* dx, long lddx, int accumulate, float* __restrict__ partial) {
  extern __shared__ float sh[];   // [waves][C][2]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int cch = C >> 3;
  float g8[LN_MAXCH][8], dg[LN_MAXCH][8], db[LN_MAXCH][8];
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; g8[i][e] = 0.f; }
    if (cc < cch) unpack8(*reinterpret_cast<const uint4*>(gamma + cc * 8), g8[i]);
  }
  const float invC = 1.0f / (float)C;
  for (int row = blockIdx.x * nw + w; row < M; row += gridDim.x * nw) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float xh[LN_MAXCH][8], d[LN_MAXCH][8];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
      int cc = lane + 64 * i;
      if (cc < cch) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + (long)row * ldx + cc * 8), f);
        unpack8(*reinterpret_cast<const uint4*>(dy + (long)row * lddy + cc * 8), d[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = (f[e] - mean) * rstd;
          db[i][e] += d[i][e]; dg[i][e] += d[i][e] * xh[i][e];
          float dxh = d[i][e] * g8[i][e];
          c1 += dxh; c2 += dxh * xh[i][e];
        }
      }
    }
    c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
      int cc = lane + 64 * i;
      if (cc < cch) {
        float o[8];
        bf16_t* op = dx + (long)row * lddx + cc * 8;
        if (accumulate) unpack8(*reinterpret_cast<const uint4*>(op), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float vv = rstd * (d[i][e] * g8[i][e] - c1 - xh[i][e] * c2);
          o[e] = accumulate ? o[e] + vv : vv;
        }
        *reinterpret_cast<uint4*>(op) = pack8(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    int cc = lane + 64 * i;
    if (cc < cch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sh[((w * C) + cc * 8 + e) * 2] = dg[i][e]; sh[((w * C) + cc * 8 + e) * 2 + 1] = db[i][e]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int ww = 0; ww < nw; ++ww) { a += sh[(ww * C + c) * 2]; b += sh[(ww * C + c) * 2 + 1]; }
    partial[((long)blockIdx.x * C + c) * 2] = a; partial[((long)blockIdx.x * C + c) * 2 + 1] = b;
  }
}

__global__ void ln_bwd_finalize_kernel(int nblk, int C, const float* __restrict__ partial, bf16_t* dgamma, bf16_t* dbeta) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int i = 0; i < nblk; ++i) { a += partial[((long)i * C + c) * 2]; b += partial[((long)i * C + c) * 2 + 1]; }
  if (dgamma) dgamma[c] = f2bf(bf2f(dgamma[c]) + a);
  if (dbeta) dbeta[c] = f2bf(bf2f(dbeta[c]) + b);
}

constexpr int LN_BWD_BLOCKS = 512;

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
  long bwd = (long)batch * g.nchunk * C * 2 + (long)batch * G * 2;
  return fwd > bwd ? fwd : bwd;
}

int az_groupnorm_fwd(int batch, int HW, int C, int G, float eps, int fuse_silu, const void* x, long ldx,
                     const void* gamma, const void* beta, void* y, long ldy, void* stats, void* partial, void* stream) {
  int rc = gn_check(batch, HW, C, G, ldx); if (rc) return rc;
  if (ldy & 7) return AZ_ERR_ARG(23);
  GnGeom g = gn_geom(batch, HW, C, G);
  hipStream_t st = (hipStream_t)stream;
  dim3 blk(g.cchunks, g.py), grid(g.nchunk, batch);
  size_t shb = (size_t)g.py * C * 2 * sizeof(float);
  if (shb > 64 * 1024) return AZ_ERR_ARG(24);
  hipLaunchKernelGGL(gn_partial_kernel, grid, blk, shb, st, g, (const bf16_t*)x, ldx, (float*)partial);
  AZ_CHECK_LAUNCH();
  int n = batch * G;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((n + 63) / 64), dim3(64), 0, st, g, eps, (const float*)partial, (float*)stats);
  AZ_CHECK_LAUNCH();
  if (fuse_silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (bf16_t*)y, ldy);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (bf16_t*)y, ldy);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_groupnorm_bwd(int batch, int HW, int C, int G, int fuse_silu, const void* x, long ldx, const void* gamma,
                     const void* beta, const void* stats, const void* dy, long lddy, void* dx, long lddx,
                     int accumulate_dx, void* dgamma, void* dbeta, void* partial, void* stream) {
  int rc = gn_check(batch, HW, C, G, ldx); if (rc) return rc;
  if ((lddy & 7) || (lddx & 7)) return AZ_ERR_ARG(25);
  GnGeom g = gn_geom(batch, HW, C, G);
  hipStream_t st = (hipStream_t)stream;
  dim3 blk(g.cchunks, g.py), grid(g.nchunk, batch);
  size_t shb = (size_t)g.py * C * 2 * sizeof(float);
  if (shb > 64 * 1024) return AZ_ERR_ARG(24);
  float* part = (float*)partial;
  float* gsum = part + (long)batch * g.nchunk * C * 2;
  if (fuse_silu)
    hipLaunchKernelGGL(gn_bwd_partial_kernel<true>, grid, blk, shb, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (const bf16_t*)dy, lddy, part);
  else
    hipLaunchKernelGGL(gn_bwd_partial_kernel<false>, grid, blk, shb, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (const float*)stats, (const bf16_t*)dy, lddy, part);
  AZ_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(G), dim3(128), 0, st, g, (const bf16_t*)gamma, (const float*)part, gsum,
                     (bf16_t*)dgamma, (bf16_t*)dbeta);
  AZ_CHECK_LAUNCH();
  if (dx) {
    if (fuse_silu)
      hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                         (const bf16_t*)beta, (const float*)stats, (const float*)gsum, (const bf16_t*)dy, lddy,
                         (bf16_t*)dx, lddx, accumulate_dx);
    else
      hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, grid, blk, 0, st, g, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                         (const bf16_t*)beta, (const float*)stats, (const float*)gsum, (const bf16_t*)dy, lddy,
                         (bf16_t*)dx, lddx, accumulate_dx);
    AZ_CHECK_LAUNCH();
  }
  return AZ_OK;
}

long az_ln_scratch_floats(int M, int C) { (void)M; return (long)LN_BWD_BLOCKS * C * 2; }

int az_layernorm_fwd(int M, int C, float eps, const void* x, long ldx, const void* gamma, const void* beta, void* y,
                     long ldy, void* stats, void* stream) {
  if (M <= 0 || (C & 7) || C > 64 * 8 * LN_MAXCH || (ldx & 7) || (ldy & 7)) return AZ_ERR_ARG(30);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, M, C, eps, (const bf16_t*)x, ldx,
                     (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, ldy, (float*)stats);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_layernorm_bwd(int M, int C, const void* x, long ldx, const void* gamma, const void* stats, const void* dy,
                     long lddy, void* dx, long lddx, int accumulate_dx, void* dgamma, void* dbeta, void* partial,
                     void* stream) {
  if (M <= 0 || (C & 7) || C > 64 * 8 * LN_MAXCH || (ldx & 7) || (lddy & 7) || (lddx & 7)) return AZ_ERR_ARG(31);
  int nblk = (M + 3) / 4; if (nblk > LN_BWD_BLOCKS) nblk = LN_BWD_BLOCKS;
  size_t shb = (size_t)4 * C * 2 * sizeof(float);
  if (shb > 64 * 1024) return AZ_ERR_ARG(32);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), shb, st, M, C, (const bf16_t*)x, ldx, (const bf16_t*)gamma,
                     (const float*)stats, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, accumulate_dx, (float*)partial);
  AZ_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, nblk, C, (const float*)partial,
                     (bf16_t*)dgamma, (bf16_t*)dbeta);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

}  // extern "C"
