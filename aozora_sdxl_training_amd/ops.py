"""Tensor-level wrappers over the C ABI (include/aozora_hip.h).

PyTorch is used only for device memory and streams: every function checks operand shapes,
strides, dtypes and alignment on the host (a faulting kernel can take the whole node down), then
passes raw pointers and the current HIP stream to libaozora_hip.so.  No function here has a CPU
or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ._lib import lib, AozoraError

BF16 = torch.bfloat16
F32 = torch.float32


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(cond, msg):
    if not cond:
        raise AozoraError("operand check failed: " + msg)


def _rows(t: torch.Tensor, dtype=BF16):
    """2-D row-major view check -> (rows, cols, ld)."""
    _req(t.is_cuda and t.dtype == dtype, f"need cuda {dtype} tensor, got {t.device} {t.dtype}")
    _req(t.dim() == 2 and t.stride(1) == 1, f"need 2-D tensor with unit inner stride, got {tuple(t.shape)} {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


class Profiler:
    """Brackets every ABI launch with HIP events on the stream it runs on and aggregates by op class
    (bench.py uses it for the per-kernel roofline; tests never enable it)."""

    def __init__(self):
        self.rec = []

    def add(self, key, flops, nbytes, e0, e1):
        self.rec.append((key, flops, nbytes, e0, e1))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        ms = ctypes.c_float()
        for key, flops, nbytes, e0, e1 in self.rec:
            lib().call("az_event_elapsed_ms", e0, e1, ctypes.byref(ms))
            d = out.setdefault(key, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1; d["ms"] += ms.value; d["flops"] += flops; d["bytes"] += nbytes
            lib().call("az_event_destroy", e0); lib().call("az_event_destroy", e1)
        self.rec = []
        return out


PROFILER: Optional[Profiler] = None
PROFILE_SHAPES = False


class _prof:
    def __init__(self, key, flops=0.0, nbytes=0.0):
        self.key, self.flops, self.nbytes = key, flops, nbytes

    def __enter__(self):
        if PROFILER is not None:
            self.e0, self.e1 = ctypes.c_void_p(), ctypes.c_void_p()
            lib().call("az_event_create", ctypes.byref(self.e0)); lib().call("az_event_create", ctypes.byref(self.e1))
            lib().call("az_event_record", self.e0, _stream())
        return self

    def __exit__(self, *a):
        if PROFILER is not None:
            lib().call("az_event_record", self.e1, _stream())
            PROFILER.add(self.key, self.flops, self.nbytes, self.e0, self.e1)
        return False


class Workspace:
    """Per-device scratch owned by the caller side of the ABI (allocated once; graph-capture safe)."""

    def __init__(self, device, splitk_bytes: int = 192 << 20, scratch_floats: int = 64 << 20, attn_bytes: int = 96 << 20):
        self.device = device
        # GEMM / conv workspace: split-K slabs, column-sum slots and -- in its last 16 KiB -- the arrival counters of the
        # in-kernel finish, which must be ZERO when first handed over and are written by the library only (include/aozora_hip.h);
        # hence zeros, and a separate buffer for the attention partials
        self.splitk = torch.zeros(splitk_bytes // 4, dtype=F32, device=device)
        self.attn = torch.empty(attn_bytes // 4, dtype=F32, device=device)
        self.scratch = torch.empty(scratch_floats, dtype=F32, device=device)
        self.small = torch.zeros(4096 + 64, dtype=F32, device=device)   # [4096:] = reserved scalars


_ws = {}


_ws_slot = [0]     # 0 = main stream, 1 = the executor's forked wgrad stream (own scratch: no sharing hazards)


def set_workspace_slot(slot: int):
    _ws_slot[0] = slot


def workspace(device=None) -> Workspace:
    device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device, _ws_slot[0])
    if key not in _ws:
        _ws[key] = Workspace(device)
    return _ws[key]


# ---------------------------------------------------------------------------------------------
# GEMM / conv
# ---------------------------------------------------------------------------------------------

def gemm(a, b, out, *, trans_a=False, trans_b=True, bias=None, rowbias=None, rows_per_seg=0, residual=None,
         accumulate=False, split_k=1, bias_grad=None):
    """out[M,N] (+)= op(a) @ op(b) (+bias) (+rowbias[m // rows_per_seg]) (+residual). See az_gemm_bf16.
    bias_grad (weight-gradient form trans_a=True, trans_b=False only): bf16 [M], += column sums of `a` in the same pass."""
    ar, ac, lda = _rows(a)
    br, bc, ldb = _rows(b)
    M, K = (ac, ar) if trans_a else (ar, ac)
    N, Kb = (br, bc) if trans_b else (bc, br)
    _req(K == Kb, f"K mismatch {K} vs {Kb}")
    _req(not (trans_a and trans_b), "transA & transB unsupported")
    om, on, ldc = _rows(out)
    _req((om, on) == (M, N), f"out shape {(om, on)} != {(M, N)}")
    _req(lda % 8 == 0 and ldb % 8 == 0, "lda/ldb must be multiples of 8")
    _req(a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0, "A/B must be 16-byte aligned")
    if not trans_a and trans_b:
        _req(K % 8 == 0, "K must be a multiple of 8")
    if trans_a:
        _req(a.shape[1] % 8 == 0 or lda >= ((M + 7) // 8) * 8, "transposed A rows must be readable in 8-element chunks")
    if not trans_b:
        _req(ldb >= ((N + 7) // 8) * 8, "B rows must be readable in 8-element chunks")
    if bias is not None:
        _req(bias.dtype == BF16 and bias.numel() == N and bias.is_contiguous(), "bias must be contiguous bf16 [N]")
    ld_rb = 0
    if rowbias is not None:
        rr, rc, ld_rb = _rows(rowbias)
        _req(rc == N and rows_per_seg > 0 and rr * rows_per_seg >= M, "rowbias shape")
    ldr = 0
    if residual is not None:
        rm, rn, ldr = _rows(residual)
        _req((rm, rn) == (M, N), "residual shape")
    ws = workspace(out.device)
    if bias_grad is not None:
        _req(trans_a and not trans_b and bias is None and rowbias is None and residual is None, "bias_grad needs the wgrad form")
        _req(bias_grad.dtype == BF16 and bias_grad.is_contiguous() and bias_grad.numel() <= M, "bias_grad must be contiguous bf16 [<=M]")
        with _prof("gemm_tn" + (f" {M}x{N}x{K}+b" if PROFILE_SHAPES else ""), 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)):
            lib().call("az_gemm_wgrad_bias_bf16", M, N, K, _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc, int(accumulate), int(split_k),
                       _ptr(ws.splitk), ws.splitk.numel() * 4, _ptr(bias_grad), bias_grad.numel(), _stream())
        return out
    with _prof("gemm_" + ("tn" if trans_a else ("nt" if trans_b else "nn")) + (f" {M}x{N}x{K}" if PROFILE_SHAPES else ""), 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)):
      lib().call("az_gemm_bf16", int(trans_a), int(trans_b), M, N, K, _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc,
               _ptr(bias), _ptr(rowbias), int(rows_per_seg), ld_rb, _ptr(residual), ldr, int(accumulate), int(split_k),
               _ptr(ws.splitk), ws.splitk.numel() * 4, _stream())
    return out


def gemm_nt_grouped(a, table, ngroups: int, total_tiles_n: int, sum_n: int = 0):
    """Many products sharing the A operand in one launch (az_gemm_nt_grouped_bf16): table = device int64 [ngroups][7]
    (W, C, bias or 0, N, ldb, ldc, first 160-wide tile column); the caller built it from tensors it checked.
    sum_n: total output columns (profiling only: algorithmic FLOPs / bytes of the launch)."""
    M, K, lda = _rows(a)
    _req(K % 8 == 0 and lda % 8 == 0 and a.data_ptr() % 16 == 0, "grouped gemm: A layout")
    with _prof("gemm_nt" + (f" grouped {M}x{sum_n}x{K} ({ngroups})" if PROFILE_SHAPES else ""), 2.0 * M * sum_n * K,
               2.0 * (M * K + sum_n * K + M * sum_n)):
        lib().call("az_gemm_nt_grouped_bf16", M, K, _ptr(a), lda, _ptr(table), int(ngroups), int(total_tiles_n), _stream())


def tn_group_table(jobs, device):
    """Record table of a grouped weight-gradient launch (az_gemm_tn_grouped_bf16).  jobs: list of (dy [K, M], x [K, N], dw [M, N],
    bias_grad bf16 [n_real <= M] or None); every operand is checked here (the records live in device memory, the library cannot).
    Products with the deepest k-range come first (their tiles start first: shorter tail).  -> (table, ngroups, total_tiles, flops, bytes)"""
    recs, tiles, flops, nbytes = [], 0, 0.0, 0.0
    for dy, x, dw, bg in sorted(jobs, key=lambda j: -j[0].shape[0]):
        K, M, lda = _rows(dy)
        Kx, N, ldb = _rows(x)
        Mw, Nw, ldc = _rows(dw)
        _req(Kx == K and (Mw, Nw) == (M, N), f"grouped wgrad shapes {tuple(dy.shape)} {tuple(x.shape)} {tuple(dw.shape)}")
        _req(lda % 8 == 0 and ldb % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0, "grouped wgrad: operand alignment")
        _req((M % 8 == 0 or lda >= ((M + 7) // 8) * 8) and ldb >= ((N + 7) // 8) * 8, "grouped wgrad: rows must be readable in 8-element chunks")
        _req((K * lda + M + 8) * 2 < 0x7FFFFFF0 and (K * ldb + N + 8) * 2 < 0x7FFFFFF0, "grouped wgrad: operand extent >= 2 GiB")
        n_real = 0
        if bg is not None:
            _req(bg.dtype == BF16 and bg.is_contiguous() and bg.numel() <= M, "grouped wgrad: bias_grad must be contiguous bf16 [<= M]")
            n_real = bg.numel()
        vec = int(N % 8 == 0 and ldc % 8 == 0 and dw.data_ptr() % 16 == 0)
        tm, tn = (M + 127) // 128, (N + 127) // 128
        recs.append([dy.data_ptr(), x.data_ptr(), dw.data_ptr(), bg.data_ptr() if bg is not None else 0, M, N, K, lda, ldb, ldc, tiles, tm, tn, n_real, vec, 0])
        tiles += tm * tn
        flops += 2.0 * M * N * K
        nbytes += 2.0 * (M * K + N * K + M * N)
    return torch.tensor(recs, dtype=torch.int64, device=device), len(recs), tiles, flops, nbytes


def gemm_tn_grouped(table, ngroups: int, total_tiles: int, flops: float = 0.0, nbytes: float = 0.0):
    """dW_g += dY_g^T X_g (+ bias gradients) for every record of `table` (tn_group_table) in ONE launch, no split-K."""
    with _prof("gemm_tn" + (f" grouped ({ngroups}) {total_tiles} tiles" if PROFILE_SHAPES else ""), flops, nbytes):
        lib().call("az_gemm_tn_grouped_bf16", _ptr(table), int(ngroups), int(total_tiles), _stream())


def gemm_geglu_fwd(x, w, bias, proj, out):
    """proj[M, 2H] = x @ w^T + bias and out[M, H] = proj[:, :H] * gelu(proj[:, H:]) in one launch (az_gemm_geglu_fwd_bf16)."""
    M, K, lda = _rows(x)
    H2, Kw, ldb = _rows(w)
    Mp, H2p, ldp = _rows(proj)
    Mo, H, ldo = _rows(out)
    _req(Kw == K and H2 == 2 * H and (Mp, Mo, H2p) == (M, M, H2) and (bias is None or bias.numel() == H2), "fused GEGLU forward shapes")
    with _prof("gemm_nt" + (f" {M}x{H2}x{K}+geglu" if PROFILE_SHAPES else ""), 2.0 * M * H2 * K, 2.0 * (M * K + H2 * K + M * H2 + M * H)):
        lib().call("az_gemm_geglu_fwd_bf16", M, H, K, _ptr(x), lda, _ptr(w), ldb, _ptr(bias), _ptr(proj), ldp, _ptr(out), ldo, _stream())
    return out


def _nhwc(t: torch.Tensor):
    """(B,H,W,C) tensor whose last dim is contiguous and whose pixel stride is uniform -> ld."""
    _req(t.is_cuda and t.dtype == BF16 and t.dim() == 4, "need cuda bf16 (B,H,W,C)")
    B, H, W, C = t.shape
    ld = t.stride(2)
    _req(t.stride(3) == 1 and t.stride(1) == W * ld and t.stride(0) == H * W * ld, f"non-uniform NHWC strides {t.stride()}")
    _req(ld % 8 == 0 and t.data_ptr() % 16 == 0, "pixel stride must be a multiple of 8 and base 16-byte aligned")
    return B, H, W, C, ld


def conv_fwd(x, w, out, *, stride=1, bias=None, rowbias=None, residual=None, upsample=False):
    """x (B,H,W,Cin) ; w [Cout][k][k][Cin] contiguous ; out (B,Ho,Wo,Cout).
    upsample: x is the HALF-resolution input of a nearest-2x upsample (Upsample2D); the conv reads it through the gather."""
    B, H, W, Cin, ldx = _nhwc(x)
    if upsample:
        _req(stride == 1 and w.shape[1] == 3, "the upsample gather exists for 3x3 stride-1 convolutions")
        H, W = 2 * H, 2 * W
    _req(w.dtype == BF16 and w.is_contiguous() and w.dim() == 4 and w.shape[3] == Cin and w.shape[1] == w.shape[2], "weight layout")
    Cout, ks = w.shape[0], w.shape[1]
    pad = 1 if ks == 3 else 0
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    _req(out.is_cuda and out.dtype == BF16 and tuple(out.shape) == (B, Ho, Wo, Cout) and out.stride(3) == 1, "out shape")
    ldo = out.stride(2)
    _req(out.stride(1) == Wo * ldo and out.stride(0) == Ho * Wo * ldo, "out strides")
    ld_rb = 0
    if rowbias is not None:
        _req(rowbias.dtype == BF16 and tuple(rowbias.shape) == (B, Cout) and rowbias.stride(1) == 1, "rowbias [B][Cout]")
        ld_rb = rowbias.stride(0)
    ldr = 0
    if residual is not None:
        _req(tuple(residual.shape) == tuple(out.shape) and residual.stride(3) == 1, "residual shape")
        ldr = residual.stride(2)
    if bias is not None:
        _req(bias.dtype == BF16 and bias.numel() == Cout and bias.is_contiguous(), "bias")
    with _prof('conv_fwd' + (f' {B}x{H}x{W} {Cin}->{Cout} k{ks}s{stride}' if PROFILE_SHAPES else ''), 2.0 * B * Ho * Wo * Cout * ks * ks * Cin, 0.0):
        lib().call("az_conv2d_bf16", 16 if upsample else 0, B, H, W, Cin, Ho, Wo, Cout, ks, stride, pad, 0, _ptr(x), ldx, _ptr(w), _ptr(None), 0,
               _ptr(out), ldo, _ptr(bias), _ptr(rowbias), ld_rb, _ptr(residual), ldr, 0, 1, _ptr(None), 0, _stream())
    return out


def conv_dgrad(dy, w, dx, *, stride=1, cout_real=None, accumulate=False, residual=None):
    """dy (B,Ho,Wo,Cpad) ; w [Cout][3][3][Cin] ; dx (B,H,W,Cin) (overwritten, accumulated in place, or = residual + gradient)."""
    B, Ho, Wo, Cpad, lddy = _nhwc(dy)
    Bx, H, W, Cin, lddx = _nhwc(dx)
    Cout = w.shape[0] if cout_real is None else cout_real
    _req(w.dtype == BF16 and w.is_contiguous() and tuple(w.shape) == (Cout, 3, 3, Cin), "weight layout")
    _req(Bx == B and Cpad >= Cout and Cpad % 8 == 0, "dy/dx batch or channel padding")
    _req(Ho == (H + 2 - 3) // stride + 1 and Wo == (W + 2 - 3) // stride + 1, "geometry")
    with _prof('conv_dgrad' + (f' {B}x{H}x{W} {Cin}<-{Cout} s{stride}' if PROFILE_SHAPES else ''), 2.0 * B * Ho * Wo * Cin * 9 * Cout, 0.0):
        ldr = _nhwc(residual)[4] if residual is not None else 0
        lib().call("az_conv2d_bf16", 1, B, H, W, Cin, Ho, Wo, Cout, 3, stride, 1, Cpad, _ptr(None), 0, _ptr(w), _ptr(dy), lddy,
               _ptr(dx), lddx, _ptr(None), _ptr(None), 0, _ptr(residual), ldr, int(accumulate), 1, _ptr(None), 0, _stream())
    return dx


def conv_dgrad_wt(dy, wt, dx, *, stride=1, accumulate=False, residual=None):
    """dgrad with the pre-transposed weight copy wt [Cin][3][3][Cout]; residual (B,H,W,Cin): dx = residual + gradient."""
    B, Ho, Wo, Cout, lddy = _nhwc(dy)
    Bx, H, W, Cin, lddx = _nhwc(dx)
    _req(wt.dtype == BF16 and wt.is_contiguous() and tuple(wt.shape) == (Cin, 3, 3, Cout) and Cout % 8 == 0, "transposed weight layout")
    _req(Bx == B and Ho == (H + 2 - 3) // stride + 1 and Wo == (W + 2 - 3) // stride + 1, "geometry")
    with _prof('conv_dgrad' + (f' {B}x{H}x{W} {Cin}<-{Cout} s{stride}' if PROFILE_SHAPES else ''), 2.0 * B * Ho * Wo * Cin * 9 * Cout, 0.0):
        ldr = _nhwc(residual)[4] if residual is not None else 0
        lib().call("az_conv2d_bf16", 3, B, H, W, Cin, Ho, Wo, Cout, 3, stride, 1, Cout, _ptr(None), 0, _ptr(wt), _ptr(dy), lddy,
                   _ptr(dx), lddx, _ptr(None), _ptr(None), 0, _ptr(residual), ldr, int(accumulate), 1, _ptr(None), 0, _stream())
    return dx


def conv_wgrad(dy, x, dw, *, stride=1, cout_real=None, accumulate=True, split_k=0, bias_grad=None, seg_grad=None, upsample=False):
    """dw [Cout][k][k][Cin] (+)= dy^T . im2col(x).  bias_grad (bf16 [Cout], +=) and seg_grad (bf16 [B][Cout], overwritten:
    per-sample channel sums of dy) are produced in the same pass when given.  upsample: x is the half-resolution input of a
    nearest-2x upsample in front of the conv."""
    B, Ho, Wo, Cdy, lddy = _nhwc(dy)
    Bx, H, W, Cin, ldx = _nhwc(x)
    if upsample:
        _req(stride == 1 and dw.shape[1] == 3, "the upsample gather exists for 3x3 stride-1 convolutions")
        H, W = 2 * H, 2 * W
    Cout = Cdy if cout_real is None else cout_real
    _req(dw.dtype == BF16 and dw.is_contiguous() and dw.dim() == 4 and dw.shape[0] == Cout and dw.shape[3] == Cin, "dw layout")
    ks = dw.shape[1]
    pad = 1 if ks == 3 else 0
    _req(Bx == B and Ho == (H + 2 * pad - ks) // stride + 1, "geometry")
    _req(lddy >= ((Cout + 7) // 8) * 8, "dy rows must be readable in 8-element chunks")
    ws = workspace(dw.device)
    if bias_grad is not None or seg_grad is not None:
        if bias_grad is not None:
            _req(bias_grad.dtype == BF16 and bias_grad.is_contiguous() and bias_grad.numel() == Cout, "bias_grad must be contiguous bf16 [Cout]")
        if seg_grad is not None:
            _req(seg_grad.dtype == BF16 and seg_grad.is_contiguous() and seg_grad.numel() == B * Cout, "seg_grad must be contiguous bf16 [B][Cout]")
            _req((Ho * Wo) % 64 == 0, "per-sample sums need Hout*Wout to be a multiple of 64")
        with _prof('conv_wgrad' + (f' {B}x{H}x{W} {Cin}x{Cout} k{ks}s{stride}+b' if PROFILE_SHAPES else ''), 2.0 * B * Ho * Wo * Cout * ks * ks * Cin, 0.0):
            lib().call("az_conv2d_wgrad_bias_bf16", B, H, W, Cin, Ho, Wo, Cout, ks | (16 if upsample else 0), stride, pad, _ptr(x), ldx, _ptr(dy), lddy, _ptr(dw),
                       int(accumulate), int(split_k), _ptr(ws.splitk), ws.splitk.numel() * 4, _ptr(bias_grad), _ptr(seg_grad), _stream())
        return dw
    with _prof('conv_wgrad' + (f' {B}x{H}x{W} {Cin}x{Cout} k{ks}s{stride}' if PROFILE_SHAPES else ''), 2.0 * B * Ho * Wo * Cout * ks * ks * Cin, 0.0):
        lib().call("az_conv2d_bf16", 18 if upsample else 2, B, H, W, Cin, Ho, Wo, Cout, ks, stride, pad, 0, _ptr(x), ldx, _ptr(None), _ptr(dy), lddy,
               _ptr(dw), ks * ks * Cin, _ptr(None), _ptr(None), 0, _ptr(None), 0, int(accumulate), int(split_k),
               _ptr(ws.splitk), ws.splitk.numel() * 4, _stream())
    return dw


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------

def _attn_view(t, heads):
    """(B,T,>=heads*64) view with unit inner stride -> (B, T, ld, sb)."""
    _req(t.is_cuda and t.dtype == BF16 and t.dim() == 3 and t.stride(2) == 1, "attention operand must be (B,T,C) bf16")
    _req(t.shape[2] == heads * 64, "last dim must be heads*64")
    _req(t.stride(1) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0, "attention operand alignment")
    return t.shape[0], t.shape[1], t.stride(1), t.stride(0)


def attn_fwd(q, k, v, o, lse, heads, scale):
    B, Tq, ldq, sq = _attn_view(q, heads)
    Bk, Tk, ldk, sk = _attn_view(k, heads)
    Bv, Tv, ldv, sv = _attn_view(v, heads)
    Bo, To, ldo, so = _attn_view(o, heads)
    _req(B == Bk == Bv == Bo and Tk == Tv and To == Tq, "attention shapes")
    _req(lse.dtype == F32 and lse.is_contiguous() and lse.numel() == B * heads * Tq, "lse buffer")
    with _prof('attn_fwd' + (f' {B}x{heads} {Tq}x{Tk}' if PROFILE_SHAPES else ''), 4.0 * B * heads * Tq * Tk * 64, 0.0):
        lib().call("az_attn_fwd", B, heads, Tq, Tk, float(scale), _ptr(q), ldq, sq, _ptr(k), ldk, sk, _ptr(v), ldv, sv,
               _ptr(o), ldo, so, _ptr(lse), _stream())
    return o


def attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, heads, scale, parts=7):
    B, Tq, ldq, sq = _attn_view(q, heads)
    _, Tk, ldk, sk = _attn_view(k, heads)
    _, _, ldv, sv = _attn_view(v, heads)
    _, _, ldo, so = _attn_view(o, heads)
    _, _, lddo, sdo = _attn_view(do, heads)
    Bq, Tq2, lddq, sdq = _attn_view(dq, heads)
    Bk, Tk2, lddk, sdk = _attn_view(dk, heads)
    Bv, Tk3, lddv, sdv = _attn_view(dv, heads)
    _req(Tq2 == Tq and Tk2 == Tk and Tk3 == Tk and Bq == B and Bk == B and Bv == B, "attention bwd shapes")
    _req(lse.dtype == F32 and lse.numel() == B * heads * Tq and delta.dtype == F32 and delta.numel() >= B * heads * Tq, "lse/delta")
    with _prof('attn_bwd' + (f' {B}x{heads} {Tq}x{Tk}' if PROFILE_SHAPES else ''), 10.0 * B * heads * Tq * Tk * 64 * (((parts >> 1) & 1) * 3 + ((parts >> 2) & 1) * 4) / 7.0, 0.0):
        lib().call("az_attn_bwd", B, heads, Tq, Tk, float(scale), _ptr(q), ldq, sq, _ptr(k), ldk, sk, _ptr(v), ldv, sv,
               _ptr(o), ldo, so, _ptr(do), lddo, sdo, _ptr(lse), _ptr(delta), _ptr(dq), lddq, sdq, _ptr(dk), lddk, sdk,
               _ptr(dv), lddv, sdv, _ptr(workspace(q.device).attn), workspace(q.device).attn.numel() * 4, int(parts), _stream())


# ---------------------------------------------------------------------------------------------
# norms
# ---------------------------------------------------------------------------------------------

def gn_scratch_floats(B, HW, C, G):
    return int(lib().raw("az_gn_scratch_floats")(B, HW, C, G))


def groupnorm_fwd(x, gamma, beta, y, stats, G, eps, silu):
    """x,y (B,HW,C) bf16 with unit channel stride; stats (B,G,2) fp32."""
    _req(x.dim() == 3 and y.shape == x.shape and x.stride(2) == 1 and y.stride(2) == 1, "groupnorm operands")
    B, HW, C = x.shape
    _req(x.stride(0) == HW * x.stride(1) and y.stride(0) == HW * y.stride(1), "batch stride")
    _req(stats.dtype == F32 and stats.numel() == B * G * 2 and stats.is_contiguous(), "stats")
    ws = workspace(x.device)
    _req(gn_scratch_floats(B, HW, C, G) <= ws.scratch.numel(), "scratch too small")
    with _prof('gn_fwd', 0.0, 4.0 * B * HW * C):
        lib().call("az_groupnorm_fwd", B, HW, C, G, float(eps), int(silu), _ptr(x), x.stride(1), _ptr(gamma), _ptr(beta),
               _ptr(y), y.stride(1), _ptr(stats), _ptr(ws.scratch), _stream())
    return y


def groupnorm_bwd(x, gamma, beta, stats, dy, dx, dgamma, dbeta, G, silu, accumulate_dx=False, dx_add=None):
    """dx_add: dx = dx_add + gradient (dx_add may be dx itself = accumulate_dx=True)."""
    B, HW, C = x.shape
    if accumulate_dx and dx_add is None:
        dx_add = dx
    _req(dx_add is None or (dx is not None and dx_add.shape == x.shape and dx_add.stride(2) == 1 and dx_add.stride(0) == HW * dx_add.stride(1)), "dx_add layout")
    _req(dy.shape == x.shape and (dx is None or dx.shape == x.shape), "groupnorm bwd shapes")
    _req(x.stride(2) == 1 and dy.stride(2) == 1 and x.stride(0) == HW * x.stride(1) and dy.stride(0) == HW * dy.stride(1), "strides")
    ws = workspace(x.device)
    _req(gn_scratch_floats(B, HW, C, G) <= ws.scratch.numel(), "scratch too small")
    with _prof('gn_bwd', 0.0, 10.0 * B * HW * C):
        lib().call("az_groupnorm_bwd_ex", B, HW, C, G, int(silu), _ptr(x), x.stride(1), _ptr(gamma), _ptr(beta), _ptr(stats),
               _ptr(dy), dy.stride(1), _ptr(dx), dx.stride(1) if dx is not None else 0, _ptr(dx_add), dx_add.stride(1) if dx_add is not None else 0,
               _ptr(dgamma), _ptr(dbeta), _ptr(ws.scratch), _stream())


def layernorm_fwd(x, gamma, beta, y, stats, eps=1e-5):
    M, C, ldx = _rows(x)
    My, Cy, ldy = _rows(y)
    _req((M, C) == (My, Cy) and stats.dtype == F32 and stats.numel() == 2 * M, "layernorm operands")
    with _prof('ln_fwd', 0.0, 4.0 * M * C):
        lib().call("az_layernorm_fwd", M, C, float(eps), _ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(y), ldy, _ptr(stats), _stream())
    return y


def layernorm_bwd(x, gamma, stats, dy, dx, dgamma, dbeta, accumulate_dx=False, dx_add=None):
    """dx may be None (parameter gradients only) and dgamma/dbeta may be None (data gradient only).
    dx_add: dx = dx_add + gradient (dx_add may be dx itself = accumulate_dx=True)."""
    M, C, ldx = _rows(x)
    if accumulate_dx and dx_add is None:
        dx_add = dx
    ld_add = 0
    if dx_add is not None:
        Ma, Ca, ld_add = _rows(dx_add)
        _req(dx is not None and (Ma, Ca) == (M, C), "dx_add shape")
    _, _, lddy = _rows(dy)
    lddx = _rows(dx)[2] if dx is not None else 0
    ws = workspace(x.device)
    _req(int(lib().raw("az_ln_scratch_floats")(M, C)) <= ws.scratch.numel(), "scratch too small")
    with _prof('ln_bwd' if dx is not None else 'ln_bwd_param', 0.0, (8.0 if dx is not None else 4.0) * M * C):
        lib().call("az_layernorm_bwd_ex", M, C, _ptr(x), ldx, _ptr(gamma), _ptr(stats), _ptr(dy), lddy, _ptr(dx), lddx,
               _ptr(dx_add), ld_add, _ptr(dgamma), _ptr(dbeta), _ptr(ws.scratch), _stream())


def ln_partial_blocks(M: int) -> int:
    return int(lib().raw("az_ln_partial_blocks")(int(M)))


def layernorm_bwd_partial(x, gamma, stats, dy, dx, partial, dx_add=None, nblk=None):
    """One-pass LayerNorm backward: dx final (= dx_add + gradient), gamma / beta gradients left as partial[nblk][C][2] fp32
    (finished later, many LayerNorms at a time, by ln_param_finish_multi)."""
    M, C, ldx = _rows(x)
    _, _, lddy = _rows(dy)
    _, _, lddx = _rows(dx)
    ld_add = 0
    if dx_add is not None:
        Ma, Ca, ld_add = _rows(dx_add)
        _req((Ma, Ca) == (M, C), "dx_add shape")
    if nblk is None:
        nblk = ln_partial_blocks(M)
    _req(partial.dtype == F32 and partial.is_contiguous() and partial.numel() >= nblk * C * 2, "partial buffer")
    with _prof('ln_bwd', 0.0, 8.0 * M * C):
        lib().call("az_layernorm_bwd_partial", M, C, _ptr(x), ldx, _ptr(gamma), _ptr(stats), _ptr(dy), lddy, _ptr(dx), lddx,
                   _ptr(dx_add), ld_add, _ptr(partial), int(nblk), _stream())


def ln_param_finish_multi(table, njobs: int, nblocks: int):
    """table: device int64 [njobs][6] (partial, dgamma, dbeta, nparts, C, first block), see az_ln_param_finish_multi."""
    with _prof('ln_bwd_param', 0.0, 0.0):
        lib().call("az_ln_param_finish_multi", _ptr(table), int(njobs), int(nblocks), _stream())


# ---------------------------------------------------------------------------------------------
# elementwise
# ---------------------------------------------------------------------------------------------

def geglu_fwd(proj, out):
    M, H2, ldp = _rows(proj)
    Mo, H, ldo = _rows(out)
    _req(Mo == M and H2 == 2 * H, "geglu shapes")
    with _prof('geglu_fwd', 0.0, 6.0 * M * H):
        lib().call("az_geglu_fwd", M, H, _ptr(proj), ldp, _ptr(out), ldo, _stream())
    return out


def geglu_bwd(proj, dout, dproj):
    M, H2, ldp = _rows(proj)
    _, H, lddo = _rows(dout)
    Md, Hd, lddp = _rows(dproj)
    _req(H2 == 2 * H and (Md, Hd) == (M, H2), "geglu bwd shapes")
    with _prof('geglu_bwd', 0.0, 10.0 * M * H):
        lib().call("az_geglu_bwd", M, H, _ptr(proj), ldp, _ptr(dout), lddo, _ptr(dproj), lddp, _stream())
    return dproj


def silu_fwd(x, y):
    _req(x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel() and x.dtype == BF16, "silu operands")
    lib().call("az_silu_fwd", x.numel(), _ptr(x), _ptr(y), _stream())
    return y


def silu_bwd(x, dy, dx, accumulate=False):
    _req(x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous() and x.numel() == dy.numel() == dx.numel(), "silu bwd")
    lib().call("az_silu_bwd", x.numel(), _ptr(x), _ptr(dy), _ptr(dx), int(accumulate), _stream())
    return dx


def add_rows(a, b, y):
    """y = a + b (row-strided 2-D views) ; b None -> strided copy."""
    R, C, lda = _rows(a)
    Ry, Cy, ldy = _rows(y)
    _req((R, C) == (Ry, Cy), "add_rows shapes")
    ldb = 0
    if b is not None:
        Rb, Cb, ldb = _rows(b)
        _req((Rb, Cb) == (R, C), "add_rows b shape")
    with _prof('add_rows', 0.0, 6.0 * R * C):
        lib().call("az_add_rows", R, C, _ptr(a), lda, _ptr(b), ldb, _ptr(y), ldy, _stream())
    return y


def upsample2x_fwd(x, y):
    B, H, W, C = x.shape
    _req(x.is_contiguous() and y.is_contiguous() and tuple(y.shape) == (B, 2 * H, 2 * W, C) and x.dtype == BF16, "upsample")
    with _prof('upsample', 0.0, 10.0 * B * H * W * C):
        lib().call("az_upsample2x_fwd", B, H, W, C, _ptr(x), _ptr(y), _stream())
    return y


def upsample2x_bwd(dy, dx):
    B, H, W, C = dx.shape
    _req(dx.is_contiguous() and dy.is_contiguous() and tuple(dy.shape) == (B, 2 * H, 2 * W, C), "upsample bwd")
    with _prof('upsample', 0.0, 10.0 * B * H * W * C):
        lib().call("az_upsample2x_bwd", B, H, W, C, _ptr(dy), _ptr(dx), _stream())
    return dx


def colsum(x, rows_per_seg, out_f32):
    R, C, ldx = _rows(x)
    _req(R % rows_per_seg == 0 and out_f32.dtype == F32 and out_f32.numel() >= (R // rows_per_seg) * C, "colsum")
    ws = workspace(x.device)
    need = int(lib().raw("az_colsum_scratch_floats")(R, C, int(rows_per_seg)))
    half = ws.scratch.numel() // 2
    _req(need <= half, "colsum scratch too small")
    # partials live in the upper half of the shared scratch (out_f32 may be its lower half)
    with _prof('colsum', 0.0, 2.0 * R * C):
        lib().call("az_colsum", R, C, int(rows_per_seg), _ptr(x), ldx, _ptr(out_f32), _ptr(ws.scratch[half:]), _stream())
    return out_f32


def colsum_grad(x, rows_per_seg, seg_out, bias_grad, n_real):
    """Fused bias-gradient form of colsum: seg_out [nseg][C] bf16 (optional) and bias_grad[:n_real] += column sums."""
    R, C, ldx = _rows(x)
    _req(R % rows_per_seg == 0 and n_real <= C, "colsum_grad shapes")
    nseg = R // rows_per_seg
    if seg_out is not None:
        _req(seg_out.dtype == BF16 and seg_out.is_contiguous() and seg_out.numel() == nseg * C, "seg_out")
    if bias_grad is not None:
        _req(bias_grad.dtype == BF16 and bias_grad.is_contiguous() and bias_grad.numel() >= n_real, "bias_grad")
    ws = workspace(x.device)
    need = int(lib().raw("az_colsum_scratch_floats")(R, C, int(rows_per_seg)))
    _req(need <= ws.scratch.numel(), "colsum scratch too small")
    with _prof('colsum', 0.0, 2.0 * R * C):
        lib().call("az_colsum_grad", R, C, int(rows_per_seg), _ptr(x), ldx, _ptr(seg_out), _ptr(bias_grad), int(n_real),
                   _ptr(ws.scratch), _stream())


def reduce_segs_to_bf16(src_f32, nseg, n, dst, accumulate):
    _req(src_f32.dtype == F32 and src_f32.numel() >= nseg * n and dst.dtype == BF16 and dst.numel() == n and dst.is_contiguous(), "reduce_segs")
    lib().call("az_reduce_segs_to_bf16", nseg, n, _ptr(src_f32), _ptr(dst), int(accumulate), _stream())


def transpose(src, dst):
    """dst[c][r] = src[r][c] (2-D bf16, unit inner strides)."""
    R, C, lds = _rows(src)
    Rd, Cd, ldd = _rows(dst)
    _req((Rd, Cd) == (C, R), "transpose shapes")
    lib().call("az_transpose_bf16", R, C, _ptr(src), lds, _ptr(dst), ldd, _stream())
    return dst


def transpose_batched(src, dst):
    """dst[b][c][r] = src[b][r][c] for 3-D bf16 views (unit inner strides, arbitrary batch / row strides)."""
    _req(src.dim() == 3 and dst.dim() == 3 and src.dtype == BF16 and dst.dtype == BF16, "transpose_batched rank/dtype")
    nb, R, C = src.shape
    _req(tuple(dst.shape) == (nb, C, R) and src.stride(2) == 1 and dst.stride(2) == 1, "transpose_batched shapes")
    lib().call("az_transpose_bf16_batched", nb, R, C, _ptr(src), src.stride(1), src.stride(0), _ptr(dst), dst.stride(1),
               dst.stride(0), _stream())
    return dst


def f32_to_bf16(src, dst):
    _req(src.dtype == F32 and dst.dtype == BF16 and src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous(), "cast")
    lib().call("az_f32_to_bf16", src.numel(), _ptr(src), _ptr(dst), _stream())


def timestep_embed(t_f32, dim, out):
    n = t_f32.numel()
    _req(t_f32.dtype == F32 and t_f32.is_contiguous() and out.dtype == BF16 and out.dim() == 2 and out.shape[0] == n
         and out.shape[1] == dim and out.stride(1) == 1, "timestep_embed")
    lib().call("az_timestep_embed", n, dim, _ptr(t_f32), _ptr(out), out.stride(0), _stream())
    return out


def nchw_to_nhwc_pad(src, dst, C):
    """src (B,C,H,W) fp32|bf16 contiguous -> dst (B,H,W,Cpad) bf16 contiguous (zero padded channels)."""
    B, Cs, H, W = src.shape
    _req(Cs == C and src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape[:3]) == (B, H, W) and dst.dtype == BF16, "nchw->nhwc")
    lib().call("az_nchw_to_nhwc_pad", B, C, H * W, dst.shape[3], _ptr(src), int(src.dtype == F32), _ptr(dst), _stream())
    return dst


def nhwc_to_nchw(src, dst, C):
    B, H, W, ld = src.shape
    _req(src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape) == (B, C, H, W), "nhwc->nchw")
    lib().call("az_nhwc_to_nchw", B, C, H * W, ld, _ptr(src), _ptr(dst), int(dst.dtype == F32), _stream())
    return dst


def noise_target(mode, latents, noise, coef_a, coef_b, noisy_nhwc, target):
    B, C, H, W = latents.shape
    _req(latents.dtype == BF16 and noise.dtype == F32 and latents.is_contiguous() and noise.is_contiguous() and
         noise.shape == latents.shape and coef_a.dtype == F32 and coef_b.dtype == F32 and coef_a.numel() == B and coef_b.numel() == B and
         noisy_nhwc.is_contiguous() and tuple(noisy_nhwc.shape[:3]) == (B, H, W) and noisy_nhwc.dtype == BF16 and
         target.dtype == F32 and target.shape == latents.shape and target.is_contiguous(), "noise_target operands")
    lib().call("az_noise_target", int(mode), B, C, H * W, noisy_nhwc.shape[3], _ptr(latents), _ptr(noise), _ptr(coef_a), _ptr(coef_b),
               _ptr(noisy_nhwc), _ptr(target), _stream())


def mse_loss_fwd_bwd(pred_nhwc, target, w, grad_scale, loss_out, per_sample, dpred):
    B, H, W, ldp = pred_nhwc.shape
    C = target.shape[1]
    _req(pred_nhwc.dtype == BF16 and pred_nhwc.is_contiguous() and target.dtype == F32 and target.is_contiguous() and
         tuple(target.shape) == (B, C, H, W) and w.dtype == F32 and w.numel() == B and loss_out.dtype == F32 and
         per_sample.dtype == F32 and per_sample.numel() == B, "mse operands")
    cpad = 0
    if dpred is not None:
        _req(dpred.dtype == BF16 and dpred.is_contiguous() and tuple(dpred.shape[:3]) == (B, H, W), "dpred")
        cpad = dpred.shape[3]
    lib().call("az_mse_loss_fwd_bwd", B, C, H * W, _ptr(pred_nhwc), ldp, _ptr(target), _ptr(w), float(grad_scale), _ptr(loss_out),
               _ptr(per_sample), _ptr(dpred), cpad if dpred is not None else C, _ptr(workspace(pred_nhwc.device).scratch), _stream())


# ---------------------------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------------------------

def sumsq(g, out_f32, accumulate):
    _req(g.is_contiguous() and g.dtype in (BF16, F32), "sumsq operand")
    ws = workspace(out_f32.device)
    lib().call("az_sumsq", g.numel(), _ptr(g), int(g.dtype == F32), _ptr(out_f32), int(accumulate), _ptr(ws.scratch), _stream())


def clip_coef(sumsq_f32, max_norm, coef, norm, unscale=1.0):
    lib().call("az_clip_coef", _ptr(sumsq_f32), float(max_norm), float(unscale), _ptr(coef), _ptr(norm), _stream())
