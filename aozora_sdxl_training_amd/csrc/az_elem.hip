// Elementwise / reduction kernels of the SDXL train step (HBM-bound; 16-byte vector accesses).
// Reference call sites: GEGLU / SiLU / residual adds / nearest upsample inside diffusers blocks
// (train.py:2760-2765), noise mix + targets train.py:2743-2758, weighted MSE train.py:2408-2416.
#include "az_common.h"
#include "aozora_hip.h"

namespace {

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

inline int grid_for(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}


__global__ void geglu_fwd_kernel(long M, int Hc, const bf16_t* __restrict__ proj, long ldp, bf16_t* __restrict__ out, long ldo) {
  long n = M * Hc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m; int c; divmod(i, Hc, m, c);
    float a[8], g[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(proj + m * ldp + c * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(proj + m * ldp + (long)Hc * 8 + c * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] * gelu_erf(g[e]);
    *reinterpret_cast<uint4*>(out + m * ldo + c * 8) = pack8(o);
  }
}

__global__ void geglu_bwd_kernel(long M, int Hc, const bf16_t* __restrict__ proj, long ldp, const bf16_t* __restrict__ dout,
                                 long lddo, bf16_t* __restrict__ dproj, long lddp) {
  long n = M * Hc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m; int c; divmod(i, Hc, m, c);
    float a[8], g[8], d[8], da[8], dg[8];
    unpack8(*reinterpret_cast<const uint4*>(proj + m * ldp + c * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(proj + m * ldp + (long)Hc * 8 + c * 8), g);
    unpack8(*reinterpret_cast<const uint4*>(dout + m * lddo + c * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) { float ge, dge; gelu_pair(g[e], ge, dge); da[e] = d[e] * ge; dg[e] = d[e] * a[e] * dge; }
    *reinterpret_cast<uint4*>(dproj + m * lddp + c * 8) = pack8(da);
    *reinterpret_cast<uint4*>(dproj + m * lddp + (long)Hc * 8 + c * 8) = pack8(dg);
  }
}

__global__ void silu_fwd_kernel(long n, const bf16_t* __restrict__ x, bf16_t* __restrict__ y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = f2bf(silu_f(bf2f(x[i])));
}
__global__ void silu_bwd_kernel(long n, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* dx, int acc) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = bf2f(dy[i]) * dsilu_f(bf2f(x[i]));
    if (acc) v += bf2f(dx[i]);
    dx[i] = f2bf(v);
  }
}

__global__ void add_rows_kernel(long rows, int Cc, const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b,
                                long ldb, bf16_t* y, long ldy) {
  long n = rows * Cc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long r; int c; divmod(i, Cc, r, c);
    uint4 ua = *reinterpret_cast<const uint4*>(a + r * lda + c * 8);
    if (b) {
      float fa[8], fb[8];
      unpack8(ua, fa);
      unpack8(*reinterpret_cast<const uint4*>(b + r * ldb + c * 8), fb);
#pragma unroll
      for (int e = 0; e < 8; ++e) fa[e] += fb[e];
      ua = pack8(fa);
    }
    *reinterpret_cast<uint4*>(y + r * ldy + c * 8) = ua;
  }
}

__global__ void upsample_fwd_kernel(int B, int H, int W, int Cc, const bf16_t* __restrict__ x, bf16_t* __restrict__ y) {
  long n = (long)B * (2 * H) * (2 * W) * Cc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cc); long p = i / Cc;
    int ox = (int)(p % (2 * W)); p /= (2 * W);
    int oy = (int)(p % (2 * H)); int b = (int)(p / (2 * H));
    const bf16_t* src = x + (((long)b * H + (oy >> 1)) * W + (ox >> 1)) * (Cc * 8L) + c * 8;
    *reinterpret_cast<uint4*>(y + i * 8) = *reinterpret_cast<const uint4*>(src);
  }
}
__global__ void upsample_bwd_kernel(int B, int H, int W, int Cc, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx) {
  long n = (long)B * H * W * Cc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cc); long p = i / Cc;
    int x0 = (int)(p % W); p /= W;
    int y0 = (int)(p % H); int b = (int)(p / H);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + (((long)b * 2 * H + 2 * y0 + dyy) * (2 * W) + 2 * x0 + dxx) * (Cc * 8L) + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8(s);
  }
}

// grid (nchunk, nseg, column blocks); block (bx <= 128 chunks of 8 channels, by rows): the block reduces
// its by row-lanes through LDS and writes partial[seg][chunk][C] (no atomics -> bitwise reproducible);
// colsum_final_kernel sums the chunk partials in a fixed order, 64 columns x 4 slices per block.
__global__ void colsum_kernel(long rows_per_seg, int C, int rows_per_chunk, const bf16_t* __restrict__ x, long ldx, float* __restrict__ partial) {
  extern __shared__ float sh[];   // [by][bx*8]
  const int seg = blockIdx.y;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  long r1 = r0 + rows_per_chunk; if (r1 > rows_per_seg) r1 = rows_per_seg;
  const int cch = C >> 3;
  const int cc = blockIdx.z * blockDim.x + threadIdx.x;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cc < cch) {
    const bf16_t* base = x + ((long)seg * rows_per_seg) * ldx + cc * 8;
#pragma unroll 4
    for (long r = r0 + threadIdx.y; r < r1; r += blockDim.y) {
      float f[8]; unpack8(*reinterpret_cast<const uint4*>(base + r * ldx), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
  }
  const int w8 = blockDim.x * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.y * w8 + threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, nth = blockDim.x * blockDim.y;
  for (int lc = tid; lc < w8; lc += nth) {
    const int c = blockIdx.z * w8 + lc;
    if (c >= C) continue;
    float a = 0.f;
    for (int y = 0; y < (int)blockDim.y; ++y) a += sh[y * w8 + lc];
    partial[((long)seg * gridDim.x + blockIdx.x) * C + c] = a;
  }
}
__global__ void colsum_final_kernel(int nparts, int C, const float* __restrict__ partial, float* __restrict__ out) {
  __shared__ float sh[4][64];
  const int seg = blockIdx.y, lc = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  float a = 0.f;
  if (c < C)
    for (int k = sl; k < nparts; k += 4) a += partial[((long)seg * nparts + k) * C + c];
  sh[sl][lc] = a;
  __syncthreads();
  if (sl == 0 && c < C) out[(long)seg * C + c] = sh[0][lc] + sh[1][lc] + sh[2][lc] + sh[3][lc];
}

// fused finalize for gradients: per column c (32 columns x 8 partial-slices per block):
//   seg_out[seg][c] = bf16(sum_k partial[seg][k][c])   (optional; the time-embedding gradient)
//   bias[c]        += sum_seg sum_k partial[seg][k][c]  (optional; c < n_real)
__global__ void colsum_grad_final_kernel(int nseg, int nparts, int C, int n_real, const float* __restrict__ partial,
                                         bf16_t* __restrict__ seg_out, bf16_t* bias) {
  __shared__ float sh[8][32];
  const int lc = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + lc;
  float tot = 0.f;
  for (int seg = 0; seg < nseg; ++seg) {
    float a = 0.f;
    if (c < C)
      for (int k = sl; k < nparts; k += 8) a += partial[((long)seg * nparts + k) * C + c];
    sh[sl][lc] = a;
    __syncthreads();
    if (sl == 0) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += sh[i][lc];
      if (c < C && seg_out) seg_out[(long)seg * C + c] = f2bf(s);
      tot += s;
    }
    __syncthreads();
  }
  if (sl == 0 && bias && c < n_real) bias[c] = f2bf(bf2f(bias[c]) + tot);
}

// dst[c][r] = src[r][c] for a batch of [R][C] bf16 matrices (strided rows on both sides); 64x64 tiles through LDS.
// VEC: R, C, both leading dimensions and both batch strides are multiples of 8 -> 16-byte global loads and stores
// (the transposition happens in the 2-byte LDS scatter); otherwise 2-byte accesses with bounds checks.
template <bool VEC>
__global__ void transpose_kernel(int R, int C, const bf16_t* __restrict__ src, long lds_, long bs_src,
                                 bf16_t* __restrict__ dst, long ldd, long bs_dst) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];   // [c][r], row stride 144 B
  src += (long)blockIdx.z * bs_src;
  dst += (long)blockIdx.z * bs_dst;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if constexpr (VEC) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = threadIdx.x + 256 * j, r = v >> 3, ch = v & 7;
      uint4 q = make_uint4(0, 0, 0, 0);
      if (r0 + r < R && c0 + ch * 8 < C) q = *reinterpret_cast<const uint4*>(src + (long)(r0 + r) * lds_ + c0 + ch * 8);
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&q);
#pragma unroll
      for (int i = 0; i < 8; ++i) tile[ch * 8 + i][r] = e[i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = threadIdx.x + 256 * j, c = v >> 3, rc = v & 7;
      if (c0 + c < C && r0 + rc * 8 < R)
        *reinterpret_cast<uint4*>(dst + (long)(c0 + c) * ldd + r0 + rc * 8) = *reinterpret_cast<const uint4*>(&tile[c][rc * 8]);
    }
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 256 threads: 64 columns x 4 row lanes
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      tile[tx][i] = (r < R && c < C) ? src[(long)r * lds_ + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
      const int c = c0 + i, r = r0 + tx;
      if (c < C && r < R) dst[(long)c * ldd + r] = tile[i][tx];
    }
  }
}

// The same for the vector-aligned case WITHOUT LDS: a lane owns an 8x8 block (8 row loads of 16 B -> 8x8 transpose in
// registers -> 8 row stores of 16 B); the 64 lanes of a wave form an 8x8 grid of blocks = one 64x64 tile, laid out so that the
// 8 lanes that share a source row (load) or a destination row (store) touch one contiguous 128-byte line.  The 2-byte LDS
// scatter of transpose_kernel<true> ran at 0.26 TB/s (8-way bank conflicts: the eight column chunks of a row map to one bank);
// the W^T refresh of a whole SDXL UNet (5 GB read + 5 GB written per optimizer step) took 38 ms of side-stream time with it.
__device__ __forceinline__ void transpose_tile_reg(const bf16_t* __restrict__ src, long lds_, bf16_t* __restrict__ dst, long ldd, int R, int C,
                                                   int tr, int tc, int lane) {
  // load: lane (i = lane >> 3, j = lane & 7) -> rows r0 + 8i .. +7, columns c0 + 8j .. +7  (8 lanes j = one 128-B row segment)
  const int li = lane >> 3, lj = lane & 7;
  const int r0 = tr * 64 + 8 * li, c0 = tc * 64 + 8 * lj;
  uint4 in[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    in[k] = (r0 + k < R && c0 < C) ? *reinterpret_cast<const uint4*>(src + (long)(r0 + k) * lds_ + c0) : make_uint4(0, 0, 0, 0);
  // 8x8 transpose of 16-bit elements: out[c] = (in[0][c], in[1][c], ..., in[7][c])
  uint4 out[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t w[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const uint32_t a = reinterpret_cast<const uint32_t*>(&in[2 * h])[c >> 1], b = reinterpret_cast<const uint32_t*>(&in[2 * h + 1])[c >> 1];
      w[h] = (c & 1) ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16));
    }
    out[c] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // store: this lane's block lands at dst rows c0 .. c0+7, columns r0 .. r0+7.  The lanes with equal j (different i) cover one
  // dst row: they are 8 apart, i.e. 8 separate 16-B pieces of one 128-B line per wave instruction; the memory pipe merges them
  // (same line, same instruction).
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c0 + c < C && r0 < R) *reinterpret_cast<uint4*>(dst + (long)(c0 + c) * ldd + r0) = out[c];
}

__global__ __launch_bounds__(256) void transpose_reg_kernel(int R, int C, const bf16_t* __restrict__ src, long lds_, long bs_src,
                                                            bf16_t* __restrict__ dst, long ldd, long bs_dst, int tiles_c, int ntiles) {
  src += (long)blockIdx.y * bs_src;
  dst += (long)blockIdx.y * bs_dst;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;
  const int tr = tile / tiles_c, tc = tile - tr * tiles_c;
  transpose_tile_reg(src, lds_, dst, ldd, R, C, tr, tc, lane);
}

// Many transposes in ONE launch: a table of jobs in device memory (8 x int64 each: src, dst, R, C, ld_src, ld_dst, first tile,
// tiles per row), one wave per 64x64 tile, the job found by bisection over the first-tile column.  The W^T refresh of a whole SDXL
// UNet is ~1200 matrices: as separate launches (58 us each beside the forward pass, 45 ms of side-stream time per optimizer step)
// it stretched the first micro-step of every iteration by ~11 ms.
struct TJob { const bf16_t* src; bf16_t* dst; long R, C, lds, ldd, tile_start, tiles_c; };
__global__ __launch_bounds__(256) void transpose_multi_kernel(const TJob* __restrict__ jobs, int njobs, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long g = (long)blockIdx.x * 4 + wave;
  if (g >= ntiles) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile_start <= g) lo = mid; else hi = mid - 1;
  }
  const TJob j = jobs[lo];
  const long tile = g - j.tile_start;
  const int tr = (int)(tile / j.tiles_c), tc = (int)(tile - (long)tr * j.tiles_c);
  transpose_tile_reg(j.src, j.lds, j.dst, j.ldd, (int)j.R, (int)j.C, tr, tc, lane);
}

__global__ void reduce_segs_kernel(int nseg, int n, const float* __restrict__ src, bf16_t* dst, int acc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nseg; ++k) s += src[(long)k * n + i];
  if (acc) s += bf2f(dst[i]);
  dst[i] = f2bf(s);
}

__global__ void f32_to_bf16_kernel(long n, const float* __restrict__ s, bf16_t* __restrict__ d) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = f2bf(s[i]);
}

__global__ void timestep_embed_kernel(int n, int dim, const float* __restrict__ t, bf16_t* __restrict__ out, long ldo) {
  int half = dim >> 1;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half) return;
  int r = i / half, j = i - r * half;
  float expo = (-9.210340371976184f * (float)j) / (float)half;   // -ln(10000) * j / half in fp32
  float arg = t[r] * expf(expo);
  out[(long)r * ldo + j] = f2bf(cosf(arg));
  out[(long)r * ldo + half + j] = f2bf(sinf(arg));
}

__global__ void nchw_to_nhwc_pad_kernel(int B, int C, int HW, int Cpad, const void* src, int is_f32, bf16_t* dst) {
  long n = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW; long p = i - b * HW;
    for (int c = 0; c < Cpad; ++c) {
      bf16_t v = 0;
      if (c < C) {
        long si = (b * C + c) * HW + p;
        v = is_f32 ? f2bf(((const float*)src)[si]) : ((const bf16_t*)src)[si];
      }
      dst[i * Cpad + c] = v;
    }
  }
}
__global__ void nhwc_to_nchw_kernel(int B, int C, int HW, int lds, const bf16_t* __restrict__ src, void* dst, int is_f32) {
  long n = (long)B * C * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long p = i % HW; long bc = i / HW; int c = (int)(bc % C); long b = bc / C;
    bf16_t v = src[(b * HW + p) * lds + c];
    if (is_f32) ((float*)dst)[i] = bf2f(v); else ((bf16_t*)dst)[i] = v;
  }
}

// one thread per (b, pixel): reads C channel planes (coalesced along pixels), writes one NHWC row
__global__ void noise_target_kernel(int mode, int B, int C, int HW, int cpad, const bf16_t* __restrict__ lat,
                                    const float* __restrict__ noise, const float* __restrict__ ca, const float* __restrict__ cb,
                                    bf16_t* __restrict__ noisy, float* __restrict__ target) {
#pragma clang fp contract(off)   // HIP's __fmul_rn/__fadd_rn are plain operators: forbid FMA fusion here
  long n = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW; long p = i - b * HW;
    float a = ca[b], s = cb[b];
    for (int c = 0; c < cpad; ++c) {
      bf16_t o = 0;
      if (c < C) {
        long si = (b * C + c) * HW + p;
        float x = bf2f(lat[si]); float nz = noise[si];
        float xt, tg;
        // the reference evaluates these expressions as separate fp32 tensor ops and the bf16 result
        // must match bit for bit: plain operators under `fp contract(off)` (no FMA fusion)
        if (mode == 2) {                 // rectified flow: fp32 throughout (train.py:2749-2750)
          float p1 = a * x, p2 = s * nz;
          xt = p1 + p2; tg = nz - x;
        } else {
          // DDPM: coefficient (bf16) * latents (bf16) is a bf16 product in the reference dataflow
          float p1 = bf2f(f2bf(a * x)), p2 = s * nz;
          xt = p1 + p2;
          if (mode == 1) { float q1 = a * nz, q2 = bf2f(f2bf(s * x)); tg = q1 - q2; } else tg = nz;
        }
        o = f2bf(xt);
        target[si] = tg;
      }
      noisy[i * cpad + c] = o;
    }
  }
}

// per (b,pixel): squared error over C channels; writes the dpred row; one partial sum per (sample, block), summed in block order by
// mse_finalize_kernel (no atomics)
__global__ void mse_kernel(int B, int C, int HW, const bf16_t* __restrict__ pred, long ldp, const float* __restrict__ target,
                           const float* __restrict__ w, float gscale, float* per_sample, bf16_t* __restrict__ dpred, int cpad) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const float k = gscale * w[b] * 2.0f / ((float)C * (float)HW * (float)B);
  float acc = 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
    long row = (long)b * HW + p;
    for (int c = 0; c < cpad; ++c) {
      bf16_t g = 0;
      if (c < C) {
        float d = bf2f(pred[row * ldp + c]) - target[((long)b * C + c) * HW + p];
        acc += d * d;
        g = f2bf(k * d);
      }
      if (dpred) dpred[row * cpad + c] = g;
    }
  }
  float tot = block_sum(acc, sh);
  if (threadIdx.x == 0) per_sample[(long)b * gridDim.x + blockIdx.x] = tot;   // partial[b][block]
}
__global__ void mse_finalize_kernel(int B, int C, int HW, int nblk, const float* __restrict__ partial, const float* __restrict__ w,
                                    float* loss_out, float* per_sample_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      float ps = 0.f;
      for (int k = 0; k < nblk; ++k) ps += partial[(long)b * nblk + k];
      float m = ps / ((float)C * (float)HW);
      if (per_sample_out) per_sample_out[b] = m;
      s += m * w[b];
    }
    loss_out[0] = s / (float)B;
  }
}

}  // namespace

// ---- per-micro-step input staging: up to 8 device-to-device segments (4-byte words) and up to 256 host floats, ONE launch -------
// (the same work as five hipMemcpyAsync device copies + one pinned H2D copy, which the runtime runs as blit kernels with 100-400 us
//  of idle stream time around each: tools/trace_gaps.py, "__amd_rocclr_copyBuffer")
struct StageArgs {
  const unsigned* src[8]; unsigned* dst[8]; long words[8]; long first_block[9];
  float coef[256]; float* coef_dst; int nseg, ncoef;
};
__global__ __launch_bounds__(256) void stage_inputs_kernel(const StageArgs a) {
  const long b = blockIdx.x;
  if (b == a.first_block[a.nseg]) {          // the last block: the host floats (they travelled in the kernel arguments)
    for (int i = threadIdx.x; i < a.ncoef; i += 256) a.coef_dst[i] = a.coef[i];
    return;
  }
  int s = 0;
  while (s + 1 < a.nseg && b >= a.first_block[s + 1]) ++s;
  const long w0 = (b - a.first_block[s]) * 4096;      // 4096 words per block, 16 per thread
  const unsigned* src = a.src[s]; unsigned* dst = a.dst[s];
  const long nw = a.words[s];
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && w0 + 4096 <= nw) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long w = w0 + (long)(k * 256 + threadIdx.x) * 4;
      *reinterpret_cast<uint4*>(dst + w) = *reinterpret_cast<const uint4*>(src + w);
    }
  } else {
    for (long w = w0 + threadIdx.x; w < w0 + 4096 && w < nw; w += 256) dst[w] = src[w];
  }
}

extern "C" {

int az_geglu_fwd(int M, int H, const void* proj, long ldp, void* out, long ldo, void* stream) {
  if (M <= 0 || (H & 7) || (ldp & 7) || (ldo & 7)) return AZ_ERR_ARG(40);
  long n = (long)M * (H / 8);
  az_launch(geglu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long)M, H / 8, (const bf16_t*)proj, ldp, (bf16_t*)out, ldo);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_geglu_bwd(int M, int H, const void* proj, long ldp, const void* dout, long lddo, void* dproj, long lddp, void* stream) {
  if (M <= 0 || (H & 7) || (ldp & 7) || (lddo & 7) || (lddp & 7)) return AZ_ERR_ARG(41);
  long n = (long)M * (H / 8);
  az_launch(geglu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (long)M, H / 8, (const bf16_t*)proj, ldp,
                     (const bf16_t*)dout, lddo, (bf16_t*)dproj, lddp);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_silu_fwd(long n, const void* x, void* y, void* stream) {
  az_launch(silu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, (const bf16_t*)x, (bf16_t*)y);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_silu_bwd(long n, const void* x, const void* dy, void* dx, int accumulate, void* stream) {
  az_launch(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, accumulate);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_add_rows(long rows, int C, const void* a, long lda, const void* b, long ldb, void* y, long ldy, void* stream) {
  if (rows <= 0 || (C & 7) || (lda & 7) || (ldy & 7) || (b && (ldb & 7))) return AZ_ERR_ARG(42);
  long n = rows * (C / 8);
  az_launch(add_rows_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, rows, C / 8, (const bf16_t*)a, lda,
                     (const bf16_t*)b, ldb, (bf16_t*)y, ldy);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_upsample2x_fwd(int batch, int H, int W, int C, const void* x, void* y, void* stream) {
  if (C & 7) return AZ_ERR_ARG(43);
  long n = (long)batch * 4 * H * W * (C / 8);
  az_launch(upsample_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, batch, H, W, C / 8, (const bf16_t*)x, (bf16_t*)y);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_upsample2x_bwd(int batch, int H, int W, int C, const void* dy, void* dx, void* stream) {
  if (C & 7) return AZ_ERR_ARG(43);
  long n = (long)batch * H * W * (C / 8);
  az_launch(upsample_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, batch, H, W, C / 8, (const bf16_t*)dy, (bf16_t*)dx);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
namespace {
struct ColsumGeom { int bx, by, zb, rpc, nchunk, nseg; };
ColsumGeom colsum_geom(long rows, int C, int rows_per_seg) {
  ColsumGeom g;
  int cch = C / 8;
  g.bx = cch < 128 ? cch : 128; g.by = 256 / g.bx; if (g.by < 1) g.by = 1; if (g.by > 32) g.by = 32;
  g.zb = (cch + g.bx - 1) / g.bx;
  g.nseg = (int)(rows / rows_per_seg);
  long want_chunks = 2048 / ((long)g.nseg * g.zb); if (want_chunks < 1) want_chunks = 1; if (want_chunks > 256) want_chunks = 256;
  long rpc = (rows_per_seg + want_chunks - 1) / want_chunks;
  if (rpc < 4L * g.by) rpc = 4L * g.by;
  g.rpc = (int)(((rpc + g.by - 1) / g.by) * g.by);
  g.nchunk = (int)((rows_per_seg + g.rpc - 1) / g.rpc);
  return g;
}
}  // namespace
long az_colsum_scratch_floats(long rows, int C, int rows_per_seg) {
  if (rows <= 0 || rows_per_seg <= 0) return 0;
  ColsumGeom g = colsum_geom(rows, C, rows_per_seg);
  return (long)g.nseg * g.nchunk * C;
}
int az_colsum(long rows, int C, int rows_per_seg, const void* x, long ldx, void* out_f32, void* scratch_f32, void* stream) {
  if (rows <= 0 || (C & 7) || (ldx & 7) || rows_per_seg <= 0 || rows % rows_per_seg) return AZ_ERR_ARG(44);
  hipStream_t st = (hipStream_t)stream;
  ColsumGeom g = colsum_geom(rows, C, rows_per_seg);
  size_t shb = (size_t)g.by * g.bx * 8 * sizeof(float);
  az_launch(colsum_kernel, dim3(g.nchunk, g.nseg, g.zb), dim3(g.bx, g.by), shb, st, (long)rows_per_seg, C, g.rpc,
                     (const bf16_t*)x, ldx, (float*)scratch_f32);
  AZ_CHECK_LAUNCH();
  az_launch(colsum_final_kernel, dim3((C + 63) / 64, g.nseg), dim3(256), 0, st, g.nchunk, C, (const float*)scratch_f32, (float*)out_f32);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_colsum_grad(long rows, int C, int rows_per_seg, const void* x, long ldx, void* seg_out_bf16, void* bias_grad_bf16,
                   int n_real, void* scratch_f32, void* stream) {
  if (rows <= 0 || (C & 7) || (ldx & 7) || rows_per_seg <= 0 || rows % rows_per_seg || n_real > C) return AZ_ERR_ARG(48);
  hipStream_t st = (hipStream_t)stream;
  ColsumGeom g = colsum_geom(rows, C, rows_per_seg);
  size_t shb = (size_t)g.by * g.bx * 8 * sizeof(float);
  az_launch(colsum_kernel, dim3(g.nchunk, g.nseg, g.zb), dim3(g.bx, g.by), shb, st, (long)rows_per_seg, C, g.rpc,
                     (const bf16_t*)x, ldx, (float*)scratch_f32);
  AZ_CHECK_LAUNCH();
  az_launch(colsum_grad_final_kernel, dim3((C + 31) / 32), dim3(256), 0, st, g.nseg, g.nchunk, C, n_real,
                     (const float*)scratch_f32, (bf16_t*)seg_out_bf16, (bf16_t*)bias_grad_bf16);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_transpose_bf16_batched(int batch, int R, int C, const void* src, long ld_src, long bstride_src, void* dst, long ld_dst,
                              long bstride_dst, void* stream) {
  if (R <= 0 || C <= 0 || batch <= 0 || batch > 65535 || ld_src < C || ld_dst < R) return AZ_ERR_ARG(49);
  const bool vec = !((R | C | ld_src | ld_dst | bstride_src | bstride_dst) & 7) && !(((uintptr_t)src | (uintptr_t)dst) & 15);
  const dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
  if (vec) {
    const int tiles_c = (C + 63) / 64, ntiles = tiles_c * ((R + 63) / 64);
    az_launch(transpose_reg_kernel, dim3((ntiles + 3) / 4, batch), dim3(256), 0, (hipStream_t)stream, R, C, (const bf16_t*)src, ld_src,
                       bstride_src, (bf16_t*)dst, ld_dst, bstride_dst, tiles_c, ntiles);
  } else
    az_launch(transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, R, C, (const bf16_t*)src, ld_src, bstride_src,
                       (bf16_t*)dst, ld_dst, bstride_dst);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_transpose_bf16(int R, int C, const void* src, long ld_src, void* dst, long ld_dst, void* stream) {
  return az_transpose_bf16_batched(1, R, C, src, ld_src, 0, dst, ld_dst, 0, stream);
}
int az_transpose_multi_bf16(const void* jobs_dev, int njobs, long ntiles, void* stream) {
  if (!jobs_dev || njobs <= 0 || ntiles <= 0 || ntiles > 0x7FFFFFF0L || ((uintptr_t)jobs_dev & 7)) return AZ_ERR_ARG(50);
  az_launch(transpose_multi_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const TJob*)jobs_dev, njobs, ntiles);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_reduce_segs_to_bf16(int nseg, int n, const void* src_f32, void* dst, int accumulate, void* stream) {
  az_launch(reduce_segs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, nseg, n, (const float*)src_f32, (bf16_t*)dst, accumulate);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_stage_inputs(int nseg, const void* src_ptrs, const void* dst_ptrs, const void* seg_bytes, int ncoef, const void* coef_host_,
                    void* coef_dev, void* stream) {
  const void* const* src = (const void* const*)src_ptrs; void* const* dst = (void* const*)dst_ptrs;
  const long* bytes = (const long*)seg_bytes; const float* coef_host = (const float*)coef_host_;
  if (nseg < 0 || nseg > 8 || ncoef < 0 || ncoef > 256 || (nseg && (!src || !dst || !bytes)) || (ncoef && (!coef_host || !coef_dev))) return AZ_ERR_ARG(56);
  StageArgs a;
  long blocks = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!src[i] || !dst[i] || bytes[i] < 0 || (bytes[i] & 3) || (((uintptr_t)src[i] | (uintptr_t)dst[i]) & 3)) return AZ_ERR_ARG(57);
    a.src[i] = (const unsigned*)src[i]; a.dst[i] = (unsigned*)dst[i]; a.words[i] = bytes[i] / 4;
    a.first_block[i] = blocks; blocks += (a.words[i] + 4095) / 4096;
  }
  a.first_block[nseg] = blocks;
  for (int i = nseg + 1; i < 9; ++i) a.first_block[i] = blocks;
  for (int i = nseg; i < 8; ++i) { a.src[i] = nullptr; a.dst[i] = nullptr; a.words[i] = 0; }
  for (int i = 0; i < ncoef; ++i) a.coef[i] = coef_host[i];
  a.coef_dst = (float*)coef_dev; a.nseg = nseg; a.ncoef = ncoef;
  if (blocks + 1 > 0x7FFFFFF0L) return AZ_ERR_ARG(57);
  az_launch(stage_inputs_kernel, dim3((unsigned)(blocks + 1)), dim3(256), 0, (hipStream_t)stream, a);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_f32_to_bf16(long n, const void* src, void* dst, void* stream) {
  az_launch(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, (const float*)src, (bf16_t*)dst);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_timestep_embed(int n, int dim, const void* t_f32, void* out, long ldo, void* stream) {
  if (n <= 0 || (dim & 1)) return AZ_ERR_ARG(45);
  int tot = n * (dim / 2);
  az_launch(timestep_embed_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, dim, (const float*)t_f32, (bf16_t*)out, ldo);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_nchw_to_nhwc_pad(int batch, int C, int HW, int Cpad, const void* src, int src_is_f32, void* dst, void* stream) {
  long n = (long)batch * HW;
  az_launch(nchw_to_nhwc_pad_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, batch, C, HW, Cpad, src, src_is_f32, (bf16_t*)dst);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_nhwc_to_nchw(int batch, int C, int HW, int ldsrc, const void* src, void* dst, int dst_is_f32, void* stream) {
  long n = (long)batch * C * HW;
  az_launch(nhwc_to_nchw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, batch, C, HW, ldsrc, (const bf16_t*)src, dst, dst_is_f32);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_noise_target(int mode, int batch, int C, int HW, int cpad, const void* latents, const void* noise,
                    const void* coef_a, const void* coef_b, void* noisy_nhwc, void* target_f32, void* stream) {
  if (mode < 0 || mode > 2 || cpad < C) return AZ_ERR_ARG(46);
  long n = (long)batch * HW;
  az_launch(noise_target_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mode, batch, C, HW, cpad,
                     (const bf16_t*)latents, (const float*)noise, (const float*)coef_a, (const float*)coef_b,
                     (bf16_t*)noisy_nhwc, (float*)target_f32);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}
int az_mse_loss_fwd_bwd(int batch, int C, int HW, const void* pred, long ldp, const void* target_f32, const void* w,
                        float grad_scale, void* loss_out, void* per_sample_out, void* dpred, int cpad, void* scratch_f32,
                        void* stream) {
  if (batch <= 0 || batch > 4096 || !per_sample_out || !scratch_f32) return AZ_ERR_ARG(47);
  hipStream_t st = (hipStream_t)stream;
  int gx = (HW + 255) / 256; if (gx > 64) gx = 64;          // scratch: batch * 64 floats
  az_launch(mse_kernel, dim3(gx, batch), dim3(256), 0, st, batch, C, HW, (const bf16_t*)pred, ldp, (const float*)target_f32,
                     (const float*)w, grad_scale, (float*)scratch_f32, (bf16_t*)dpred, cpad);
  AZ_CHECK_LAUNCH();
  az_launch(mse_finalize_kernel, dim3(1), dim3(64), 0, st, batch, C, HW, gx, (const float*)scratch_f32, (const float*)w,
                     (float*)loss_out, (float*)per_sample_out);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

}  // extern "C"
