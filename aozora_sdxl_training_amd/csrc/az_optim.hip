// Optimizer-side kernels: global grad norm / clip (train.py:2771-2781, titan.py:162-184) and the
// Raven / Titan fused AdamW update (raven.py:96-147, titan.py:230-296) over a flat parameter range
// with first/second moments resident in PINNED HOST memory, streamed through the GPU by async copies.
#include "az_common.h"
#include "aozora_hip.h"
#include <map>
#include <mutex>

namespace {

typedef _Float16 f16_t;   // momentum_dtype torch.float16 (raven.py:37-42): IEEE half, round-to-nearest-even, overflow -> inf

constexpr int SUMSQ_BLOCKS = 1024;

template <typename T> __device__ __forceinline__ float ldf(const T* p, long i);
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }
template <> __device__ __forceinline__ float ldf<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<f16_t>(const f16_t* p, long i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void stf(T* p, long i, float v);
template <> __device__ __forceinline__ void stf<f16_t>(f16_t* p, long i, float v) { p[i] = (f16_t)v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, long i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void stf<float>(float* p, long i, float v) { p[i] = v; }

template <typename T>
__global__ void sumsq_partial_kernel(long n, const T* __restrict__ g, float* __restrict__ partial) {
  __shared__ float sh[16];
  float s = 0.f;
  if (sizeof(T) == 2) {
    const long n8 = n >> 3;
    const uint4* g8 = reinterpret_cast<const uint4*>(g);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
      uint4 u = g8[i];
      const uint32_t* w = reinterpret_cast<const uint32_t*>(&u);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = __uint_as_float(w[e] << 16), b = __uint_as_float(w[e] & 0xFFFF0000u);
        s += a * a + b * b;
      }
    }
    for (long i = (n8 << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      float a = ldf<T>(g, i); s += a * a;
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      float a = ldf<T>(g, i); s += a * a;
    }
  }
  float tot = block_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void sumsq_final_kernel(int nblk, const float* __restrict__ partial, float* out, int accumulate) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += (double)partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)sh[0];
}

__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float unscale, float* coef, float* norm) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float nrm = sqrtf(sumsq[0]) * unscale;
    norm[0] = nrm;
    float c = max_norm / (nrm + 1e-6f);
    coef[0] = (c < 1.0f ? c : 1.0f) * unscale;
  }
}

// hyper: [0] lr (unused here) [1] beta1 [2] beta2 [3] eps [4] wd_factor [5] step_size [6] sqrt_bc2
template <typename TM, typename TG>
__global__ void adamw_kernel(long n, bf16_t* __restrict__ p, const TG* __restrict__ g, TM* __restrict__ m, TM* __restrict__ v,
                             const float* __restrict__ hyper, const float* __restrict__ coef) {
  const float b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wdf = hyper[4], step = hyper[5], sbc2 = hyper[6];
  const float gc = coef ? coef[0] : 1.0f;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // The reference's operation order (raven.py:125-143, fp32 scratch tensors), rounding for rounding -- this TU's default
    // contraction would fuse differently and the update is host-link-bound, so the extra roundings cost nothing:
    //   exp_avg.mul_(b1).add_(g, alpha=1-b1)            ATen's add-with-alpha is a fused multiply-add (vec::fmadd)
    //   exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2)  self + ((value * g) * g), each product and the sum rounded
    //   p.mul_(wd_factor); denom = sqrt(v) / sqrt_bc2 + eps; p.addcdiv_(m, denom, value=-step_size)   self + ((value * m) / denom)
#pragma clang fp contract(off)
    float gr = ldf<TG>(g, i) * gc;
    if constexpr (sizeof(TG) == 2) gr = bf2f(f2bf(gr));   // = reading a gradient that was clipped in place (bf16 rounding)
    float mm = ldf<TM>(m, i) * b1; mm = __builtin_fmaf(gr, omb1, mm);
    float vv = ldf<TM>(v, i) * b2; vv = vv + ((omb2 * gr) * gr);
    float pp = bf2f(p[i]) * wdf;
    const float denom = sqrtf(vv) / sbc2 + eps;
    pp = pp + ((-step * mm) / denom);
    p[i] = f2bf(pp);
    stf<TM>(m, i, mm);
    stf<TM>(v, i, vv);
  }
}

template <typename TG>
__global__ void offload_kernel(long n, const TG* __restrict__ g, float* __restrict__ gh, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = ldf<TG>(g, i);
    gh[i] = accumulate ? gh[i] + v : v;
  }
}

// in-place clip of bf16 grads (torch.nn.utils.clip_grad_norm_ semantics: g = bf16(g * coef)); a
// coefficient of exactly 1 leaves the buffer untouched (no traffic).
__global__ void scale_bf16_kernel(long n, bf16_t* g, const float* coef) {
  const float c = coef[0];
  if (c == 1.0f) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] = f2bf(bf2f(g[i]) * c);
}

__global__ void scale_f32_kernel(long n, float* x, const float* coef) {
  const float c = coef[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= c;
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

int launch_adamw(long n, void* p, const void* g, int gdtype, void* m, void* v, int mdtype, const void* hyper, const void* coef,
                 hipStream_t st) {
  dim3 grid(grid_for(n)), blk(256);
  const float* hy = (const float*)hyper; const float* cf = (const float*)coef;
  if (mdtype == 0 && gdtype == 0)
    az_launch((adamw_kernel<bf16_t, bf16_t>), grid, blk, 0, st, n, (bf16_t*)p, (const bf16_t*)g, (bf16_t*)m, (bf16_t*)v, hy, cf);
  else if (mdtype == 1 && gdtype == 0)
    az_launch((adamw_kernel<float, bf16_t>), grid, blk, 0, st, n, (bf16_t*)p, (const bf16_t*)g, (float*)m, (float*)v, hy, cf);
  else if (mdtype == 0 && gdtype == 1)
    az_launch((adamw_kernel<bf16_t, float>), grid, blk, 0, st, n, (bf16_t*)p, (const float*)g, (bf16_t*)m, (bf16_t*)v, hy, cf);
  else if (mdtype == 1 && gdtype == 1)
    az_launch((adamw_kernel<float, float>), grid, blk, 0, st, n, (bf16_t*)p, (const float*)g, (float*)m, (float*)v, hy, cf);
  else if (mdtype == 2 && gdtype == 0)
    az_launch((adamw_kernel<f16_t, bf16_t>), grid, blk, 0, st, n, (bf16_t*)p, (const bf16_t*)g, (f16_t*)m, (f16_t*)v, hy, cf);
  else if (mdtype == 2 && gdtype == 1)
    az_launch((adamw_kernel<f16_t, float>), grid, blk, 0, st, n, (bf16_t*)p, (const float*)g, (f16_t*)m, (f16_t*)v, hy, cf);
  else
    return AZ_ERR_ARG(60);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

// Hand-off events of the chunk pipeline, one set per COMPUTE STREAM (a stream belongs to one device, so two optimizers,
// threads or devices in one process never share a set); creation is serialised by a mutex.  Calls that name the same
// compute stream must come from one host thread at a time -- the stream's own order is what sequences them.
struct EvPool {
  hipEvent_t h2d[2], comp[2], d2h[2];
};
std::mutex g_ev_mutex;
std::map<hipStream_t, EvPool> g_ev_pools;

int ev_pool_for(hipStream_t sc, EvPool** out) {
  std::lock_guard<std::mutex> lock(g_ev_mutex);
  auto it = g_ev_pools.find(sc);
  if (it == g_ev_pools.end()) {
    EvPool ep;
    for (int i = 0; i < 2; ++i) {
      AZ_HIP(hipEventCreateWithFlags(&ep.h2d[i], hipEventDisableTiming));
      AZ_HIP(hipEventCreateWithFlags(&ep.comp[i], hipEventDisableTiming));
      AZ_HIP(hipEventCreateWithFlags(&ep.d2h[i], hipEventDisableTiming));
    }
    it = g_ev_pools.emplace(sc, ep).first;
  }
  *out = &it->second;
  return AZ_OK;
}

}  // namespace

extern "C" {

int az_sumsq(long n, const void* g, int dtype, void* out_f32, int accumulate, void* scratch_f32, void* stream) {
  if (n <= 0) return AZ_ERR_ARG(61);
  hipStream_t st = (hipStream_t)stream;
  int nblk = grid_for(n); if (nblk > SUMSQ_BLOCKS) nblk = SUMSQ_BLOCKS;
  if (dtype == 0) {
    if ((uintptr_t)g & 15) return AZ_ERR_ARG(62);
    az_launch(sumsq_partial_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, n, (const bf16_t*)g, (float*)scratch_f32);
  } else {
    az_launch(sumsq_partial_kernel<float>, dim3(nblk), dim3(256), 0, st, n, (const float*)g, (float*)scratch_f32);
  }
  AZ_CHECK_LAUNCH();
  az_launch(sumsq_final_kernel, dim3(1), dim3(256), 0, st, nblk, (const float*)scratch_f32, (float*)out_f32, accumulate);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_sumsq_bf16(long n, const void* g, void* out_f32, int accumulate, void* scratch_f32, void* stream) {
  return az_sumsq(n, g, 0, out_f32, accumulate, scratch_f32, stream);
}

int az_clip_coef(const void* sumsq_f32, float max_norm, float grad_unscale, void* coef_f32, void* norm_f32, void* stream) {
  az_launch(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)sumsq_f32, max_norm, grad_unscale,
                     (float*)coef_f32, (float*)norm_f32);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_adamw_flat(long n, void* p, const void* g, void* m, void* v, int mdtype, const void* hyper, const void* coef,
                  void* stream) {
  if (n <= 0) return AZ_ERR_ARG(63);
  return launch_adamw(n, p, g, 0, m, v, mdtype, hyper, coef, (hipStream_t)stream);
}

int az_adamw_flat_ex(long n, void* p, const void* g, int gdtype, void* m, void* v, int mdtype, const void* hyper,
                     const void* coef, void* stream) {
  if (n <= 0) return AZ_ERR_ARG(63);
  return launch_adamw(n, p, g, gdtype, m, v, mdtype, hyper, coef, (hipStream_t)stream);
}

// Chunk pipeline: H2D(m,v)[c+1]  ||  adamw[c]  ||  D2H(m,v)[c-1]; staging = 2 buffers x (m,v) x chunk.
int az_raven_step_ex(long n, void* p, const void* g, int gdtype, void* m_host, void* v_host, int mdtype, const void* hyper,
                     const void* coef, void* staging, long chunk_elems, void* stream_compute, void* stream_h2d,
                     void* stream_d2h) {
  if (n <= 0 || chunk_elems <= 0 || mdtype < 0 || mdtype > 2) return AZ_ERR_ARG(64);
  hipStream_t sc = (hipStream_t)stream_compute, sh = (hipStream_t)stream_h2d, sd = (hipStream_t)stream_d2h;
  EvPool* evp = nullptr;
  { int rc0 = ev_pool_for(sc, &evp); if (rc0) return rc0; }
  EvPool& g_ev = *evp;
  const size_t esz = mdtype == 1 ? 4 : 2;
  const size_t gsz = gdtype == 0 ? 2 : 4;
  char* stg = (char*)staging;
  const long nchunk = (n + chunk_elems - 1) / chunk_elems;
  // the copy streams must not start before work already queued on the compute stream (grads, coef)
  hipEvent_t& start = g_ev.comp[0];
  AZ_HIP(hipEventRecord(start, sc));
  AZ_HIP(hipStreamWaitEvent(sh, start, 0));
  for (long c = 0; c < nchunk; ++c) {
    const int buf = (int)(c & 1);
    const long off = c * chunk_elems;
    const long len = (off + chunk_elems <= n) ? chunk_elems : (n - off);
    char* mb = stg + (size_t)buf * 2 * chunk_elems * esz;
    char* vb = mb + (size_t)chunk_elems * esz;
    if (c >= 2) AZ_HIP(hipStreamWaitEvent(sh, g_ev.d2h[buf], 0));       // staging buffer free again
    AZ_HIP(hipMemcpyAsync(mb, (char*)m_host + off * esz, len * esz, hipMemcpyHostToDevice, sh));
    AZ_HIP(hipMemcpyAsync(vb, (char*)v_host + off * esz, len * esz, hipMemcpyHostToDevice, sh));
    AZ_HIP(hipEventRecord(g_ev.h2d[buf], sh));
    AZ_HIP(hipStreamWaitEvent(sc, g_ev.h2d[buf], 0));
    int rc = launch_adamw(len, (bf16_t*)p + off, (const char*)g + off * gsz, gdtype, mb, vb, mdtype, hyper, coef, sc);
    if (rc) return rc;
    AZ_HIP(hipEventRecord(g_ev.comp[buf], sc));
    AZ_HIP(hipStreamWaitEvent(sd, g_ev.comp[buf], 0));
    AZ_HIP(hipMemcpyAsync((char*)m_host + off * esz, mb, len * esz, hipMemcpyDeviceToHost, sd));
    AZ_HIP(hipMemcpyAsync((char*)v_host + off * esz, vb, len * esz, hipMemcpyDeviceToHost, sd));
    AZ_HIP(hipEventRecord(g_ev.d2h[buf], sd));
  }
  // join: the compute stream observes the end of the last write-backs
  AZ_HIP(hipStreamWaitEvent(sc, g_ev.d2h[0], 0));
  if (nchunk > 1) AZ_HIP(hipStreamWaitEvent(sc, g_ev.d2h[1], 0));
  return AZ_OK;
}

int az_raven_step(long n, void* p, const void* g, void* m_host, void* v_host, int mdtype, const void* hyper,
                  const void* coef, void* staging, long chunk_elems, void* stream_compute, void* stream_h2d,
                  void* stream_d2h) {
  return az_raven_step_ex(n, p, g, 0, m_host, v_host, mdtype, hyper, coef, staging, chunk_elems, stream_compute, stream_h2d, stream_d2h);
}

int az_titan_offload(long n, const void* g, void* g_host_f32, void* staging_f32, int accumulate, void* stream) {
  (void)staging_f32;
  if (n <= 0) return AZ_ERR_ARG(65);
  az_launch(offload_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, (const bf16_t*)g, (float*)g_host_f32, accumulate);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_scale_bf16(long n, void* g, const void* coef_f32, void* stream) {
  if (n <= 0) return AZ_ERR_ARG(67);
  az_launch(scale_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, (bf16_t*)g, (const float*)coef_f32);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

int az_scale_f32(long n, void* x, const void* coef_f32, void* stream) {
  if (n <= 0) return AZ_ERR_ARG(66);
  az_launch(scale_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, (float*)x, (const float*)coef_f32);
  AZ_CHECK_LAUNCH();
  return AZ_OK;
}

}  // extern "C"
