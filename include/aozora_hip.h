/* aozora_hip.h -- C ABI of libaozora_hip.so (MI355X / gfx950).
 *
 * The reference (Hysocs/Aozora_SDXL_Training) is pure Python and has no FFI; every native
 * instruction on its hot path comes from PyTorch/ATen through diffusers.  This library is the
 * native layer a maintainer would bind (ctypes stub: INTEGRATION.md) to replace those calls for
 * the SDXL-UNet training step.  Each entry point cites the reference call it stands in for.
 *
 * Conventions: every pointer is a DEVICE pointer unless its name ends in `_host`
 * (pinned host memory).  Memory is caller-owned.  `stream` is a hipStream_t passed as void*.
 * bf16 tensors are raw uint16 bits.  Activations are NHWC: row-major [pixels][channels] with a
 * leading dimension (elements) per row.  Conv weights are [Cout][ky][kx][Cin] (the memory of a
 * torch (Cout,Cin,kh,kw) tensor in channels_last format).  Every function returns 0 on success,
 * -(hipError_t) for a HIP failure, or -1000-k for argument error k; nothing throws or prints.
 * All launches are asynchronous on `stream` and graph-capturable (no allocation, no sync).
 */
#ifndef AOZORA_HIP_H
#define AOZORA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ----------------------------------------------------------------------------- */
int az_version(void);
/* device properties of the current device: out[0]=CU count, out[1]=is_gfx950, out[2]=LDS/CU bytes */
int az_device_info(int* out3);
/* hipGraph capture of everything launched on `stream` between begin/end (thread-local capture) */
int az_graph_begin(void* stream);
int az_graph_end(void* stream, void** graph_exec_out);
int az_graph_launch(void* graph_exec, void* stream);
int az_graph_destroy(void* graph_exec);
/* pinned host memory for Raven/Titan state (raven.py:83-84 allocates pageable; we pin) */
int az_host_alloc(void** ptr_host, long bytes);
int az_host_free(void* ptr_host);
/* HIP events on an explicit stream (bench.py times kernels on the stream they run on) */
int az_event_create(void** ev);
int az_event_record(void* ev, void* stream);
int az_event_sync(void* ev);
int az_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int az_event_destroy(void* ev);
/* fork / join events of the two-stream executor (the reference has one stream and no such events): created with
 * hipEventDisableTiming | hipEventDisableSystemFence -- they order kernels of one device only, and without the system-scope fence a
 * record costs the recording stream about half (tools/event_cost.cpp).  az_event_record / az_event_destroy serve both kinds. */
int az_event_create_fork(void** ev);
int az_stream_wait_event(void* stream, void* ev);
/* While set (per thread; NULL clears), every kernel this thread launches through the library carries `ev` as its own completion
 * signal (hipExtLaunchKernel's stop event): az_set_launch_stop_event(ev); az_xxx(..., stream); az_set_launch_stop_event(NULL) leaves
 * `ev` in the state az_event_record(ev, stream) behind az_xxx would, without the record packet on `stream`.  Used by the launch
 * tape's peephole (tape.fuse_records) for the executor's forks. */
int az_set_launch_stop_event(void* ev);
int az_stream_sync(void* stream);
/* one idle wave for `microseconds` (<= 100 000) on `stream`: concurrency probe for the executor's stream choice (HIP streams
 * share a few hardware queues; the data-gradient chain, the weight-gradient branch and the exchange stream must not) */
int az_spin(long microseconds, void* stream);
int az_memset_async(void* ptr, int value, long bytes, void* stream);
/* kind: 0 default, 1 H2D, 2 D2H, 3 D2D */
int az_memcpy_async(void* dst, const void* src, long bytes, int kind, void* stream);

/* ---- dense contractions (torch.nn.functional.linear / conv2d and their autograd backward as
 *      executed inside diffusers' UNet, train.py:2760-2761 fwd, 2765 bwd) ------------------------ */
/* C[M,N] (+)= op(A) . op(B) (+bias[n]) (+rowbias[m / rows_per_seg][n]) (+residual[m][n])
 *   transA=0: A[m*lda+k]   transA=1: A[k*lda+m]
 *   transB=1: B[n*ldb+k]   transB=0: B[k*ldb+n]        (transA=1,transB=1 unsupported)
 *   linear fwd  : transA=0, transB=1  (B = weight[out][in])
 *   linear dgrad: transA=0, transB=0  (A = dY, B = weight)
 *   linear wgrad: transA=1, transB=0  (A = dY, B = X ; M=out, N=in, K=rows)
 * split_k: 1 = none, 0 = auto, >1 = forced (needs fp32 workspace of split*M*N*4 bytes).
 * K, lda, ldb multiples of 8; A, B 16-byte aligned. */
/* tuning hook: force the cooperative tile (128|256 x 128|256) of the GEMM/conv core; (0,0) = heuristic */
int az_gemm_set_tile(int bm, int bn);
/* the same with the 128x160 tile (k-contiguous B only; waves = 4: 64x80 per wave, 8: 32x80 per wave, 0: default) */
int az_gemm_set_tile_ex(int bm, int bn, int waves);
/* scheduling hint from the executor: 1 while no other stream has work that wants to share the CUs' LDS (the forward
 * pass) -- products on <= 256-tile grids then use the 3-stage / 108-KiB variant; 0 during the backward pass, where the
 * weight-gradient stream's workgroups must stay co-resident */
int az_gemm_set_exclusive(int on);
/* Runtime options: one process-wide table of integer knobs read by the launchers at launch time (tile policy, split-K
 * heuristics, LDS exclusivity ...; names and defaults: csrc/az_common.h AzOption / csrc/az_runtime.hip).  Each knob is an atomic --
 * setting one from any thread between launches is safe -- and starts from the environment variable AZ_<NAME> when that is set.
 * Options steer speed only: every setting computes the same mathematical result (split-K changes the fp32 summation order).
 * Unknown name: -1092.  The library keeps no other mutable state besides the contexts below and the forced tile of az_gemm_set_tile_ex (a test hook). */
/* ref: none (execution policy of this library; the reference has no counterpart) */
int az_set_option(const char* name, int value);
int az_get_option(const char* name, int* value);
/* Per-device / per-owner state (SURVEY.md section 8b `az_init(device, *handle)`): a context carries its OWN copy of the option
 * table (initialised from the process-wide one).  az_make_current(handle) binds it to the CALLING THREAD: every launcher called
 * from that thread, and az_set_option / az_get_option, then use the context's table; az_make_current(NULL) returns the thread to
 * the process-wide table.  The context holds no device memory and makes no HIP call (the caller selects the device: one process
 * -- or one thread -- per GPU); az_context_device returns the device it was created for.  az_destroy drops the owner's reference
 * (and unbinds the context from the calling thread if current); a thread that still has it current keeps its table alive until
 * that thread makes something else current or exits, so a handle destroyed on one thread never dangles on another.  SCOPE of
 * az_set_option: the calling thread's current context when it has one, else the process-wide table -- a context created later
 * copies the process-wide table, not another context's.  Errors: -1093 for a NULL handle / negative device. */
/* ref: train.py:2551 (`device = cuda if available else cpu`: the reference's one device per process) */
int az_init(int device, void** handle);
int az_make_current(void* handle);
int az_context_device(void* handle, int* device);
int az_destroy(void* handle);
/* WORKSPACE CONTRACT (az_gemm_bf16, az_gemm_wgrad_bias_bf16, az_conv2d_bf16, az_conv2d_wgrad_bias_bf16): `workspace` holds the
 * fp32 split-K slabs and the column-sum slots; give each stream that issues products concurrently its own.  Nothing is assumed
 * about its contents and nothing in it survives a call.  While option INKERNEL_FINISH is set (default 0 -- measured slower than
 * the separate reduce launch in the two-stream step) its LAST 16 KiB hold one arrival counter per output tile, zeroed by the
 * library in front of every launch: the workgroup that arrives last at a tile's counter sums the tile's slabs in ascending split
 * order and finishes its column sums (no reduce / finish kernel follows the product).  Grids of more than 4096 tiles, or a
 * workspace of < 80 KiB, use the separate finish launches. */
/* ref: train.py:2760-2761 (every torch.nn.Linear inside unet(...): time/add embedding MLPs, proj_in/out, to_q/k/v/out, ff.net.*), train.py:2765 (their autograd dgrad / wgrad) */
int az_gemm_bf16(int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                 void* C, long ldc, const void* bias, const void* rowbias, int rows_per_seg, long ld_rowbias,
                 const void* residual, long ldr, int accumulate, int split_k, void* workspace, long workspace_bytes,
                 void* stream);

/* MANY linear products that share the A operand in ONE launch: C_g[M, N_g] = A[M, K] . W_g[N_g, K]^T (+ bias_g) for every group g.
 * groups_dev: device array of ngroups records of seven int64 each -- W pointer, C pointer, bias pointer (or 0), N, ldb, ldc, index of
 * the group's first 160-wide tile column (a group takes (N + 159) / 160 columns; first-column indices ascend from 0);
 * total_tiles_n = their total.  Every group: N % 8 == 0, ldb % 8 == 0, ldc % 8 == 0, W and C 16-byte aligned, extents below
 * 2 GiB (the caller checks: the records live in device memory).  Used for the products whose A operand does not depend on the layer:
 * the cross-attention to_k|to_v projections of the text context (70 per step) and the ResnetBlock2D time_emb_proj linears (17). */
/* ref: train.py:2760-2761 (attn2.to_k / attn2.to_v and time_emb_proj inside unet(...): same input for every layer) */
int az_gemm_nt_grouped_bf16(int M, int K, const void* A, long lda, const void* groups_dev, int ngroups, long total_tiles_n, void* stream);

/* MANY independent weight-gradient products in ONE launch: dW_g[M_g, N_g] += dY_g[K_g, M_g]^T . X_g[K_g, N_g] (and, where a bias
 * gradient pointer is given, bias_g[:n_real] += column sums of dY_g) for every product g, each over its WHOLE k-range on 128x128
 * tiles -- no split-K slabs, no reduce / finish launches: the tiles of all products fill the chip together.
 * groups_dev: device array of ngroups records of sixteen int64 each -- dY, X, dW, bias-gradient pointer (or 0), M, N, K, lddy, ldx,
 * lddw, first tile id, tiles_m = ceil(M / 128), tiles_n = ceil(N / 128), n_real (<= M), vec (1: N % 8 == 0, lddw % 8 == 0 and dW
 * 16-byte aligned -> 16-byte epilogue), 0.  First tile ids ascend from 0; total_tiles = their total.  Every product: lddy % 8 == 0,
 * ldx % 8 == 0, dY / X 16-byte aligned, rows readable in 8-element chunks, operand extents below 2 GiB (the caller checks: the
 * records live in device memory).  Results are deterministic (one workgroup owns a tile and its bias-gradient rows). */
/* ref: train.py:2765 (loss.backward(): grad_weight = dY^T X, grad_bias = dY.sum(0) of every nn.Linear of a transformer block) */
int az_gemm_tn_grouped_bf16(const void* groups_dev, int ngroups, long total_tiles, void* stream);

/* The GEGLU projection with the GEGLU itself in the epilogue: proj[M, 2H] = X[M, K] . W[2H, K]^T + bias (value | gate halves, kept
 * for the backward pass) AND out[M, H] = value * gelu(gate) (erf GELU), computed from the bf16-rounded projection -- bit for bit what
 * az_gemm_bf16 followed by az_geglu_fwd produces, without reading proj back (one launch and an [M, 2H] read less per
 * transformer block).  H % 8 == 0, K % 8 == 0, leading dimensions % 8 == 0, all pointers 16-byte aligned; bias may be NULL. */
/* ref: train.py:2760-2761 (diffusers FeedForward: GEGLU.proj Linear, then hidden * gelu(gate)) */
int az_gemm_geglu_fwd_bf16(int M, int H, int K, const void* X, long lda, const void* W, long ldb, const void* bias, void* proj, long ldp,
                           void* out, long ldo, void* stream);

/* linear weight gradient dW[M=out][N=in] (+)= dY^T . X with the BIAS gradient fused into the same pass over dY
 * (torch autograd computes grad_bias = dY.sum(0) as a separate reduction): bias_grad[m] += sum_k dY[k][m] for m < n_real.
 * The column sums ride on the matrix pipe (one extra MFMA per A fragment against an all-ones fragment); split-K
 * partials are summed in a fixed order.  Needs the fp32 workspace (>= 64*M*4 bytes beyond the split-K slabs). */
/* ref: train.py:2765 (autograd of nn.Linear: grad_weight = dY^T X and grad_bias = dY.sum(0)) */
int az_gemm_wgrad_bias_bf16(int M, int N, int K, const void* dY, long lddy, const void* X, long ldx, void* dW, long lddw,
                            int accumulate, int split_k, void* workspace, long workspace_bytes, void* bias_grad, int n_real,
                            void* stream);

/* 3x3 (pad 1, stride 1|2) or 1x1 convolution on NHWC as implicit GEMM.
 *   mode 0 forward : out = Y[B,Hout,Wout,Cout] from X, W (+bias[co]) (+rowbias[b][co]) (+residual)
 *   mode 1 dgrad   : out = dX[B,Hin,Win,Cin]   from dY, W   (3x3 only; 1x1 dgrad is az_gemm_bf16)
 *   mode 2 wgrad   : out = dW[Cout][k][k][Cin] from dY, X   (accumulate / split_k as az_gemm_bf16)
 *   mode 3 dgrad   : as mode 1 but W is the pre-transposed copy W'[Cin][ky][kx][Cout] (Cout % 8 == 0): NT-form product
 * `cpad` (mode 1): channel count dY rows are padded to (>= Cout, multiple of 8; 0 => Cout).
 * mode | 16 (with mode 0 or 2, 3x3, stride 1): X is stored at HALF resolution [B][Hin/2][Win/2][Cin] and is read through a
 *   nearest-neighbour 2x gather -- diffusers Upsample2D (F.interpolate(scale_factor=2, mode="nearest") followed by the conv)
 *   without materialising the upsampled tensor; Hin / Win stay the UPSAMPLED extents.  az_conv2d_wgrad_bias_bf16 takes the same
 *   flag as ksize | 16.
 * ldx / lddy / ldo / ldr: elements between consecutive pixels.  Cin multiple of 8. */
/* ref: train.py:2760-2761 / 2765 (every nn.Conv2d of ResnetBlock2D, Downsample2D, Upsample2D, conv_in, conv_out and its backward) */
int az_conv2d_bf16(int mode, int batch, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int ksize, int stride,
                   int pad, int cpad, const void* X, long ldx, const void* W, const void* dY, long lddy, void* out,
                   long ldo, const void* bias, const void* rowbias, long ld_rowbias, const void* residual, long ldr,
                   int accumulate, int split_k, void* workspace, long workspace_bytes, void* stream);
/* conv weight gradient (mode 2 of az_conv2d_bf16) with fused bias gradient and, optionally, the per-sample column sums
 * seg_grad[b][co] = sum over the sample's pixels of dY (ResnetBlock2D: the gradient of the time-embedding projection
 * that was broadcast-added after conv1).  bias_grad (+=) and seg_grad (overwritten) are bf16; either may be NULL. */
/* ref: train.py:2765 (autograd of nn.Conv2d weight / bias; the per-sample sums are the gradient of ResnetBlock2D's time_emb_proj broadcast add) */
int az_conv2d_wgrad_bias_bf16(int batch, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int ksize, int stride, int pad,
                              const void* X, long ldx, const void* dY, long lddy, void* dW, int accumulate, int split_k,
                              void* workspace, long workspace_bytes, void* bias_grad, void* seg_grad, void* stream);

/* ---- attention (F.scaled_dot_product_attention via diffusers AttnProcessor2_0,
 *      train.py:204-228; head_dim 64, no mask, dropout 0) -------------------------------------- */
/* Q[b][tq][h*64+d] with row stride ldq (elements) and batch stride sq; same for K,V (tk), O.
 * lse[b][h][tq] fp32 = log-sum-exp of scaled scores (saved for backward). */
/* ref: train.py:204-228 (attention processor choice) -> F.scaled_dot_product_attention executed at train.py:2760-2761 */
int az_attn_fwd(int batch, int heads, int Tq, int Tk, float scale, const void* Q, long ldq, long sq, const void* K,
                long ldk, long sk, const void* V, long ldv, long sv, void* O, long ldo, long so, void* lse,
                void* stream);
/* dQ,dK,dV from dO (+ saved Q,K,V,O,lse); delta[b][h][tq] fp32 scratch. dQ/dK/dV are overwritten.
 * workspace (optional fp32 scratch): lets short-context (cross-attention) dK/dV split the query range across
 * workgroups; partials are summed in a fixed order (no atomics).
 * parts: bit0 delta = rowsum(dO*O), bit1 dQ kernel, bit2 dK/dV kernel (0 = all); dQ and dK/dV only need delta,
 * so a caller may issue them on different streams. */
/* ref: train.py:2765 (autograd of scaled_dot_product_attention) */
int az_attn_bwd(int batch, int heads, int Tq, int Tk, float scale, const void* Q, long ldq, long sq, const void* K,
                long ldk, long sk, const void* V, long ldv, long sv, const void* O, long ldo, long so, const void* dO,
                long lddo, long sdo, const void* lse, void* delta, void* dQ, long lddq, long sdq, void* dK, long lddk,
                long sdk, void* dV, long lddv, long sdv, void* workspace, long workspace_bytes, int parts, void* stream);

/* ---- normalisation (torch GroupNorm / LayerNorm inside diffusers blocks; fp32 statistics) ---- */
/* GroupNorm over NHWC x[B][HW][C] (ld = ldx), G groups, optional fused SiLU.  stats[B][G][2] fp32
 * (mean, rstd) is written by fwd and consumed by bwd.  partial: fp32 scratch >= az_gn_scratch_floats.
 * gamma / beta: 16-byte aligned (read 8 channels at a time), stats 8-byte aligned. */
/* ref: no reference counterpart (workspace size query) */
long az_gn_scratch_floats(int batch, int HW, int C, int G);
/* ref: train.py:2760-2761 (nn.GroupNorm(32) + SiLU of ResnetBlock2D, Transformer2DModel.norm, conv_norm_out) */
int az_groupnorm_fwd(int batch, int HW, int C, int G, float eps, int fuse_silu, const void* x, long ldx,
                     const void* gamma, const void* beta, void* y, long ldy, void* stats, void* partial, void* stream);
/* dx (overwritten or accumulated), dgamma/dbeta (bf16, ACCUMULATED in place) */
/* ref: train.py:2765 (its autograd) */
int az_groupnorm_bwd(int batch, int HW, int C, int G, int fuse_silu, const void* x, long ldx, const void* gamma,
                     const void* beta, const void* stats, const void* dy, long lddy, void* dx, long lddx,
                     int accumulate_dx, void* dgamma, void* dbeta, void* partial, void* stream);
/* the same with the accumulation source named separately: dx = (dx_add ? dx_add : 0) + gradient.  dx_add == dx is the in-place
 * form above; a different buffer leaves dx_add untouched, so a weight-gradient product that still reads it (the residual stream's
 * gradient is some layer's dY) need not have finished */
/* ref: train.py:2765 (autograd accumulates the residual stream's gradient out of place as well) */
int az_groupnorm_bwd_ex(int batch, int HW, int C, int G, int fuse_silu, const void* x, long ldx, const void* gamma,
                        const void* beta, const void* stats, const void* dy, long lddy, void* dx, long lddx,
                        const void* dx_add, long ld_add, void* dgamma, void* dbeta, void* partial, void* stream);
/* LayerNorm over rows of x[M][C]; stats[M][2] fp32.  partial: fp32 scratch >= az_ln_scratch_floats. */
/* ref: no reference counterpart (workspace size query) */
long az_ln_scratch_floats(int M, int C);
/* ref: train.py:2760-2761 (nn.LayerNorm norm1-3 of BasicTransformerBlock) */
int az_layernorm_fwd(int M, int C, float eps, const void* x, long ldx, const void* gamma, const void* beta, void* y,
                     long ldy, void* stats, void* stream);
/* dx may be NULL (gamma / beta gradients only); dgamma / dbeta may be NULL (data gradient only) */
/* ref: train.py:2765 (its autograd) */
int az_layernorm_bwd(int M, int C, const void* x, long ldx, const void* gamma, const void* stats, const void* dy,
                     long lddy, void* dx, long lddx, int accumulate_dx, void* dgamma, void* dbeta, void* partial,
                     void* stream);
/* the same with the accumulation source named separately (see az_groupnorm_bwd_ex) */
/* ref: train.py:2765 (same) */
int az_layernorm_bwd_ex(int M, int C, const void* x, long ldx, const void* gamma, const void* stats, const void* dy,
                        long lddy, void* dx, long lddx, const void* dx_add, long ld_add, void* dgamma, void* dbeta, void* partial,
                        void* stream);
/* The one-pass LayerNorm backward with the gamma / beta gradients left as partial sums: dx (= dx_add + gradient) is final,
 * partial[nblk][C][2] fp32 holds per-block (dgamma, dbeta) sums, to be finished later by az_ln_param_finish_multi -- the
 * parameter gradients are not needed before the end of the backward pass.  nblk: the block count the caller sized `partial`
 * (and its finish job) for, as returned by az_ln_partial_blocks(M) when it did; the launch derives its rows per block from
 * nblk and returns an argument error for a count that function cannot have returned, so a recorded launch replayed after the
 * LN_RPB option changed fails instead of writing past the buffer. */
/* ref: train.py:2765 (autograd of nn.LayerNorm; grad of weight / bias) */
int az_layernorm_bwd_partial(int M, int C, const void* x, long ldx, const void* gamma, const void* stats, const void* dy,
                             long lddy, void* dx, long lddx, const void* dx_add, long ld_add, void* partial, int nblk, void* stream);
/* ref: no reference counterpart (size query of the above: rows of the partial-sum buffer) */
int az_ln_partial_blocks(int M);
/* dgamma[c] += sum_k partial[k][c][0], dbeta[c] += sum_k partial[k][c][1] for MANY LayerNorms in one launch.  jobs_dev: device array
 * of njobs records of six int64 each -- partial, dgamma (or 0), dbeta (or 0), number of partial rows, C, index of the job's first
 * block (a job takes (C + 31) / 32 blocks; first-block indices ascend from 0); nblocks = their total. */
/* ref: train.py:2765 (same) */
int az_ln_param_finish_multi(const void* jobs_dev, int njobs, long nblocks, void* stream);

/* ---- elementwise / reductions ------------------------------------------------------------------ */
/* GEGLU (diffusers GEGLU, exact-erf GELU): proj[M][2H] -> out[M][H] = proj[:, :H] * gelu(proj[:, H:]) */
/* ref: train.py:2760-2761 (diffusers GEGLU of ff.net.0: hidden * gelu(gate), exact erf) */
int az_geglu_fwd(int M, int H, const void* proj, long ldp, void* out, long ldo, void* stream);
/* ref: train.py:2765 (its autograd) */
int az_geglu_bwd(int M, int H, const void* proj, long ldp, const void* dout, long lddo, void* dproj, long lddp,
                 void* stream);
/* ref: train.py:2760-2761 (SiLU on the summed time / text embedding before time_emb_proj) */
int az_silu_fwd(long n, const void* x, void* y, void* stream);
/* ref: train.py:2765 (its autograd) */
int az_silu_bwd(long n, const void* x, const void* dy, void* dx, int accumulate, void* stream);
/* y[rows][C] (ldy) = a (lda) + b (ldb) ; b may be null (strided copy) */
/* ref: train.py:2765 (autograd accumulation of a residual / skip gradient: grad += grad) */
int az_add_rows(long rows, int C, const void* a, long lda, const void* b, long ldb, void* y, long ldy, void* stream);
/* nearest-neighbour 2x upsample NHWC and its adjoint (2x2 sum) */
/* ref: train.py:2760-2761 (Upsample2D: F.interpolate(scale_factor=2, mode="nearest")) */
int az_upsample2x_fwd(int batch, int H, int W, int C, const void* x, void* y, void* stream);
/* ref: train.py:2765 (its autograd) */
int az_upsample2x_bwd(int batch, int H, int W, int C, const void* dy, void* dx, void* stream);
/* out[seg][C] fp32 = column sums of x[seg*rows_per_seg ...][C] (bias / time-embedding grads); two ordered
 * passes through scratch_f32 (>= az_colsum_scratch_floats), no atomics: bitwise reproducible */
/* ref: no reference counterpart (workspace size query) */
long az_colsum_scratch_floats(long rows, int C, int rows_per_seg);
/* ref: train.py:2765 (sum over rows as used by bias / broadcast gradients) */
int az_colsum(long rows, int C, int rows_per_seg, const void* x, long ldx, void* out_f32, void* scratch_f32, void* stream);
/* fused gradient form: seg_out_bf16[seg][C] = per-segment column sums (nullable: the time-embedding gradient of
 * ResnetBlock2D), bias_grad_bf16[c] += total column sums for c < n_real (nullable). Same scratch as az_colsum. */
/* ref: train.py:2765 (grad_bias of Linear / Conv2d and the gradient of the time-embedding broadcast add) */
int az_colsum_grad(long rows, int C, int rows_per_seg, const void* x, long ldx, void* seg_out_bf16, void* bias_grad_bf16,
                   int n_real, void* scratch_f32, void* stream);
/* dst_bf16[n] (+)= src_f32[seg][n] summed over nseg segments (finishes az_colsum into a bf16 grad) */
/* ref: train.py:2765 (finishes az_colsum into a bf16 .grad) */
int az_reduce_segs_to_bf16(int nseg, int n, const void* src_f32, void* dst, int accumulate, void* stream);
/* dst[c*ld_dst + r] = src[r*ld_src + c]: transposed weight copies W^T that turn every linear dgrad into the faster
 * k-contiguous (NT) product; refreshed once per optimizer step */
/* ref: no reference counterpart: W^T copies so that F.linear's dgrad (train.py:2765) reads k-contiguous operands */
int az_transpose_bf16(int R, int C, const void* src, long ld_src, void* dst, long ld_dst, void* stream);
/* the same for `batch` matrices at element strides bstride_src / bstride_dst (the 9 taps of a conv weight
 * [Cout][3][3][Cin] -> [Cin][3][3][Cout] in one launch) */
/* ref: no reference counterpart (same, all 9 taps of a conv weight) */
int az_transpose_bf16_batched(int batch, int R, int C, const void* src, long ld_src, long bstride_src, void* dst, long ld_dst,
                              long bstride_dst, void* stream);
/* many such transposes in ONE launch.  jobs_dev: device array of njobs records of eight int64 each -- src pointer, dst pointer, R, C,
 * ld_src, ld_dst, index of the job's first 64x64 tile, tiles per row ((C + 63) / 64) -- with first-tile indices ascending from 0;
 * ntiles = their total.  Every job must satisfy the 16-byte form: R, C, ld_src, ld_dst multiples of 8, pointers 16-byte aligned
 * (the caller checks; the whole W^T refresh of a UNet region is one call) */
/* ref: no reference counterpart (same, all weights of a parameter region) */
int az_transpose_multi_bf16(const void* jobs_dev, int njobs, long ntiles, void* stream);
/* fp32 [rows][C] -> bf16 */
/* ref: train.py:2760 (`.to(config.compute_dtype)` casts around the UNet call) */
int az_f32_to_bf16(long n, const void* src, void* dst, void* stream);
/* sinusoidal embedding (diffusers get_timestep_embedding, flip_sin_to_cos, shift 0):
 * out[i][0:half]=cos, [half:dim]=sin of t[i]*exp(-ln(1e4)*j/half); t fp32; out bf16 (ld = ldo) */
/* ref: train.py:2760-2761 (diffusers Timesteps / get_timestep_embedding inside unet(...)) */
int az_timestep_embed(int n, int dim, const void* t_f32, void* out, long ldo, void* stream);
/* NCHW (fp32 or bf16) <-> NHWC bf16 with channel padding (latents 4ch -> 8ch) */
/* ref: train.py:2760 (the NCHW latent handed to unet(...); layout change only) */
int az_nchw_to_nhwc_pad(int batch, int C, int HW, int Cpad, const void* src, int src_is_f32, void* dst, void* stream);
/* ref: train.py:2761 (`.sample` returned in NCHW) */
int az_nhwc_to_nchw(int batch, int C, int HW, int ldsrc, const void* src, void* dst, int dst_is_f32, void* stream);

/* ---- step glue (train.py:2743-2765) ------------------------------------------------------------- */
/* mode 0 epsilon, 1 v_prediction, 2 rectified_flow.  latents/noise NCHW [B][CHW]: latents bf16,
 * noise fp32.  coef_a/coef_b fp32 [B]: (sqrt_ab, sqrt_1m_ab) or (1-t, t).  Outputs: noisy NHWC bf16
 * padded to cpad channels (the UNet input), target fp32 NCHW. */
/* ref: train.py:2743-2758 (rectified-flow mix / scheduler.add_noise / get_velocity) */
int az_noise_target(int mode, int batch, int C, int HW, int cpad, const void* latents, const void* noise,
                    const void* coef_a, const void* coef_b, void* noisy_nhwc, void* target_f32, void* stream);
/* weighted_sdxl_mse_loss (train.py:2408-2416) forward + d(loss*scale)/dpred.
 * pred NHWC bf16 [B][HW][ldp], target fp32 NCHW, w fp32 [B] (curve[timestep]).  loss_out fp32[1]
 * is overwritten; scratch_f32 >= 64*B floats (ordered partial sums, no atomics).  dpred NHWC bf16 padded to cpad channels (zeros). */
/* ref: train.py:2408-2416 (weighted_sdxl_mse_loss), train.py:2763-2765 ((loss / GA).backward() seed) */
int az_mse_loss_fwd_bwd(int batch, int C, int HW, const void* pred, long ldp, const void* target_f32, const void* w,
                        float grad_scale, void* loss_out, void* per_sample_out, void* dpred, int cpad, void* scratch_f32,
                        void* stream);

/* ---- optimizer (raven.py:89-149, titan.py:119-131,162-184,230-296; clip train.py:2771-2781) ------- */
/* sum of squares of n bf16 grads -> out_f32[0] (accumulate=1 adds to existing value) */
/* ref: train.py:2771-2781 (torch.nn.utils.clip_grad_norm_: global L2 norm) */
int az_sumsq_bf16(long n, const void* g, void* out_f32, int accumulate, void* scratch_f32, void* stream);
/* same with dtype 0 = bf16, 1 = fp32 (fp32 may be pinned host memory: Titan's CPU-resident grads).
 * scratch_f32: >= 1024 floats */
/* ref: train.py:2771-2781, titan.py:162-184 (norm of host fp32 gradients) */
int az_sumsq(long n, const void* g, int dtype, void* out_f32, int accumulate, void* scratch_f32, void* stream);
/* clip coefficient on device: coef[0] = min(1, max_norm / (sqrt(sumsq*inv_scale2) + 1e-6)); norm[0] = sqrt(...) */
/* ref: train.py:2775-2778 (clip coefficient max_norm / (norm + 1e-6), clamped to 1) */
int az_clip_coef(const void* sumsq_f32, float max_norm, float grad_unscale, void* coef_f32, void* norm_f32, void* stream);
/* fused AdamW on a flat range: p bf16 (device), g bf16 (device), m/v (device staging copies of the
 * host state, dtype mdtype: 0 bf16, 1 fp32, 2 fp16 -- raven.py:37-42 momentum_dtype).  hyper (device fp32[8]): lr, beta1, beta2, eps,
 * wd_factor, step_size(lr/bc1), sqrt_bc2, unused.  coef (device fp32[1]) multiplies g (clip); a bf16 gradient is
 * rounded to bf16 after the multiplication, i.e. exactly the value an in-place clip_grad_norm_ would have left. */
/* ref: raven.py:89-149 (RavenAdamW.step element math) */
int az_adamw_flat(long n, void* p, const void* g, void* m, void* v, int mdtype, const void* hyper, const void* coef,
                  void* stream);
/* Raven step over a flat parameter range with m,v resident in PINNED HOST memory: chunked
 * H2D(m,v) -> az_adamw_flat -> D2H(m,v) pipelined over 3 streams with double-buffered device staging
 * (staging: device scratch >= 4 * chunk_elems * sizeof(mdtype)).  The hand-off events are kept per compute stream (created
 * on first use under a mutex), so optimizers on different streams / devices of one process are independent; calls naming the
 * SAME compute stream must not be issued from two host threads at once. */
/* ref: raven.py:103-149 (per-parameter H2D of exp_avg / exp_avg_sq, update, D2H) */
int az_raven_step(long n, void* p, const void* g, void* m_host, void* v_host, int mdtype, const void* hyper,
                  const void* coef, void* staging, long chunk_elems, void* stream_compute, void* stream_h2d,
                  void* stream_d2h);
/* _ex variants: gdtype 0 = bf16 grads (device), 1 = fp32 grads (device or pinned host: Titan) */
/* ref: raven.py:89-149, titan.py:230-296 (fp32 host gradients) */
int az_adamw_flat_ex(long n, void* p, const void* g, int gdtype, void* m, void* v, int mdtype, const void* hyper,
                     const void* coef, void* stream);
/* ref: raven.py:103-149, titan.py:230-296 */
int az_raven_step_ex(long n, void* p, const void* g, int gdtype, void* m_host, void* v_host, int mdtype, const void* hyper,
                     const void* coef, void* staging, long chunk_elems, void* stream_compute, void* stream_h2d,
                     void* stream_d2h);
/* g_bf16[i] = bf16(g[i] * coef[0]) in place -- the in-place clip of torch.nn.utils.clip_grad_norm_
 * (train.py:2775-2778); skipped entirely when coef[0] == 1 */
/* ref: train.py:2775-2778 (in-place gradient scaling of clip_grad_norm_) */
int az_scale_bf16(long n, void* g, const void* coef_f32, void* stream);
/* x_f32[i] *= coef[0]  (Titan's CPU-side clip of host grads, titan.py:177-182) */
/* ref: titan.py:177-182 */
int az_scale_f32(long n, void* x, const void* coef_f32, void* stream);
/* Per-micro-step input staging in ONE launch.  src_ptrs / dst_ptrs: HOST arrays of nseg <= 8 device pointers (4-byte aligned),
 * seg_bytes: HOST array of nseg longs (each % 4 == 0): dst[i][0 .. bytes[i]) = src[i][..]; coef_host: HOST array of ncoef <= 256
 * floats -> coef_dev[0 .. ncoef).  All four host arrays are read DURING the call (the floats travel in the kernel arguments: the
 * caller may reuse them at once -- no pinned buffer, no event).  Replaces the per-micro-step tensor placements of the reference's
 * loop body (train.py:2731 time_ids, 2733-2758 noise / timestep-derived coefficients, 2760 the UNet's inputs) when the batch is
 * already on the device: as hipMemcpyAsync copies they ran as blit kernels with 0.1-0.4 ms of idle stream time around each. */
int az_stage_inputs(int nseg, const void* src_ptrs, const void* dst_ptrs, const void* seg_bytes, int ncoef, const void* coef_host,
                    void* coef_dev, void* stream);
/* Titan: offload grad range to host fp32 (copy, or add when accumulate) via device staging */
/* ref: titan.py:93-100, 119-131 (post-accumulate hook: copy_ on the first micro-step, add_ afterwards) */
int az_titan_offload(long n, const void* g, void* g_host_f32, void* staging_f32, int accumulate, void* stream);

/* ---- native launch tape: the executor seam (SURVEY.md 8b "az_unet_step") ----------------------------------------------------
 * The step's launch sequence is static (same entry points, pointers, streams and events every step of a resolution bucket).  The
 * host executor records it once; az_tape_play re-issues it from C: entry-point calls (arguments as 64-bit words: int / long by
 * value, float by bit pattern, pointers as integers), hipEventRecord, hipStreamWaitEvent, until a BREAK (kind 3: host logic of the
 * caller runs between two plays) or the end.  kind: 0 = call of entry point `fn` (az_tape_fn_id), 1 = event record (words: event,
 * stream), 2 = stream wait (words: stream, event), 3 = break.  az_tape_play returns the index to resume from (= the number of
 * operations when the tape is done), -2 after a failing operation (az_tape_last_error names it). */
/* ref: train.py:2743-2767 (the loop body whose launch sequence is recorded and replayed) */
int az_tape_fn_id(const char* name);
/* ref: train.py:2743-2767 (same) */
int az_tape_create(void** tape);
/* ref: train.py:2743-2767 (same) */
int az_tape_destroy(void* tape);
/* ref: train.py:2743-2767 (same) */
int az_tape_add(void* tape, int kind, int fn, const void* words, int nwords);
/* ref: train.py:2743-2767 (same) */
long az_tape_play(void* tape, long start);
/* ref: train.py:2743-2767 (same) */
int az_tape_last_error(void* tape, long* index, int* rc);

#ifdef __cplusplus
}
#endif
#endif
