"""Per-op parity at the BENCHMARK's own shapes (BASELINE configs[1]: SDXL-base UNet, local batch 4, 1024x1024 => latent
128x128; M = 4096 rows at the 1280-wide level, 16384 at the 640-wide one): the products that carry the step -- each one under
the tile / stage / split-K variant the model picks for it and under the alternatives that exist for it -- against fp32 PyTorch
on the CPU on identical bf16-rounded operands.  (tests/test_kernels_gpu.py covers ragged and tail shapes up to 1000x640x1280;
the grid sizes, k-depths and split-K slab counts here are the real ones.)  Tolerances as there: relative Frobenius error
<= 4e-3 (bf16 output rounding is 2^-9), max abs error <= 2e-2 of the largest reference element; 8e-3 / 3e-2 for attention
gradients.  Each case also runs twice and must reproduce itself bit for bit (ordered reductions only)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 64))
    from aozora_sdxl_training_amd import ops as _ops
    return _ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def check(out, ref, name, fro=4e-3, mx=2e-2):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape and torch.isfinite(out).all(), name
    e_fro = (out - ref).norm().item() / (ref.norm().item() + 1e-12)
    e_max = (out - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    assert e_fro <= fro and e_max <= mx, f"{name}: rel_fro={e_fro:.3e} (<= {fro}) rel_max={e_max:.3e} (<= {mx})"
    return e_fro


def _force(tile):
    from aozora_sdxl_training_amd._lib import lib
    lib().call("az_gemm_set_tile_ex", *tile)


# (M, N, K, forced tiles to run besides the heuristic (0,0,0)); the comment names the layer
NT_CASES = [
    (4096, 1280, 10240, [(128, 160, 8), (128, 160, 24), (128, 160, 40), (128, 160, 56), (128, 128, 8), (256, 256, 0), (256, 256, 32)]),   # dgrad of ff.net.0.proj (K-heavy, 256 tiles of 128x160)
    (4096, 1280, 1280, [(128, 160, 8), (128, 160, 24), (128, 160, 40), (128, 160, 56), (128, 128, 8)]),    # to_out / to_q / proj_in / proj_out: 384 launches per micro-step
    (4096, 10240, 1280, [(256, 256, 0), (256, 256, 32), (128, 160, 8), (256, 256, 8), (256, 320, 8)]),     # ff.net.0.proj forward (256-row tiles: 16-wave, 8-wave ping-pong 256 / 320 wide)
    (4096, 5120, 1280, [(256, 256, 0), (256, 256, 8), (256, 320, 8)]),                     # ff.net.2 data gradient: 256 tiles of 256x320 = one wave of CUs
    (4096, 3840, 1280, [(256, 256, 0), (256, 256, 8)]),                                    # fused q|k|v projection
    (16384, 5120, 640, [(256, 256, 0), (128, 128, 8), (256, 256, 8), (256, 320, 8)]),      # ff.net.0.proj at the 640-wide level
    (16384, 1920, 640, [(256, 256, 8)]),                                                   # q|k|v at the 640-wide level: N = 7.5 tiles of 256 (masked columns)
    (4096, 1280, 5120, [(128, 160, 8), (128, 160, 24)]),                                   # ff.net.2 forward
    (16384, 640, 640, [(128, 160, 8)]),
]


@pytest.mark.parametrize("M,N,K,tiles", NT_CASES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else (str(v) if isinstance(v, int) else ""))
def test_linear_forward_and_dgrad_products(ops, M, N, K, tiles):
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a.float() @ w.float().t() + bias.float() + res.float()
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    try:
        for tile in [(0, 0, 0)] + tiles:
            _force(tile)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            ops.gemm(ad, wd, out, trans_b=True, bias=bd, residual=rd)
            check(out, ref, f"gemm_nt {M}x{N}x{K} tile {tile}")
            out2 = torch.empty_like(out)
            ops.gemm(ad, wd, out2, trans_b=True, bias=bd, residual=rd)
            assert torch.equal(out, out2), f"gemm_nt {M}x{N}x{K} tile {tile}: not reproducible"
        # the data-gradient form accumulates into an existing gradient (residual fan-in)
        _force((0, 0, 0))
        acc = res.to(DEV).clone()
        ops.gemm(ad, wd, acc, trans_b=True, accumulate=True)
        check(acc, a.float() @ w.float().t() + res.float(), f"gemm_nt accumulate {M}x{N}x{K}")
    finally:
        _force((0, 0, 0))


@pytest.mark.parametrize("M,N,K", [(4096, 1280, 10240), (4096, 1280, 3840), (4096, 1280, 5120), (1024, 1280, 2048)])
def test_few_tile_products_on_split_256_row_tiles(ops, M, N, K):
    """Option NT_SPLIT_BIG: the N = 1280 family (80 tiles of 256x256) on 256-row 8-wave tiles with k split to cover the chip;
    bias, out-of-place residual and accumulation are applied by the slab reduce.  Checked against fp32 torch like the unsplit form,
    and the 16-wave fallback (GEMM8 = 0) likewise."""
    from aozora_sdxl_training_amd._lib import set_option, get_option
    a, w, bias, res = rnd(M, K, seed=21), rnd(N, K, scale=K ** -0.5, seed=22), rnd(N, seed=23), rnd(M, N, seed=24)
    ref = a.float() @ w.float().t() + bias.float() + res.float()
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    saved = {k: get_option(k) for k in ("NT_SPLIT_BIG", "NT_SPLIT_MINK", "GEMM8")}
    try:
        for g8 in (1, 0):
            set_option("GEMM8", g8); set_option("NT_SPLIT_BIG", 4); set_option("NT_SPLIT_MINK", 2048)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            ops.gemm(ad, wd, out, trans_b=True, bias=bd, residual=rd)
            check(out, ref, f"split 256-row tiles {M}x{N}x{K} gemm8={g8}")
            out2 = torch.empty_like(out)
            ops.gemm(ad, wd, out2, trans_b=True, bias=bd, residual=rd)
            assert torch.equal(out, out2), "not reproducible"
            acc = rd.clone()
            ops.gemm(ad, wd, acc, trans_b=True, accumulate=True)
            check(acc, a.float() @ w.float().t() + res.float(), f"split 256-row tiles, accumulate {M}x{N}x{K} gemm8={g8}")
    finally:
        for k, v in saved.items():
            set_option(k, v)


# dW[M,N] += dY[K,M]^T X[K,N]: (M, N, K, explicit split counts besides the heuristic 0)
TN_CASES = [
    (1280, 1280, 4096, [1, 3, 8]),      # 192 launches per micro-step: 100 tiles, 5 slabs by the heuristic
    (10240, 1280, 4096, [1, 2]),        # ff.net.0.proj weight gradient, fused bias gradient
    (1280, 5120, 4096, [2]),            # ff.net.2
    (3840, 1280, 4096, [1]),            # fused q|k|v projection
    (640, 640, 16384, [4, 16]),         # k = 16384 pixels: deep split-K
    (5120, 640, 16384, [1]),
    (2560, 2048, 308, [1]),             # cross-attention k|v: 308 context rows
]


@pytest.mark.parametrize("M,N,K,splits", TN_CASES, ids=lambda v: str(v) if isinstance(v, int) else "")
def test_weight_gradient_products(ops, M, N, K, splits):
    dy, x, prev, bprev = rnd(K, M, seed=5), rnd(K, N, seed=6), rnd(M, N, scale=0.1, seed=7), rnd(M, scale=0.1, seed=8)
    ref = prev.float() + dy.float().t() @ x.float()
    bref = bprev.float() + dy.float().sum(0)
    dyd, xd = dy.to(DEV), x.to(DEV)
    for split in [0] + splits:
        out = prev.to(DEV).clone()
        ops.gemm(dyd, xd, out, trans_a=True, trans_b=False, accumulate=True, split_k=split)
        check(out, ref, f"gemm_tn {M}x{N}x{K} split {split}")
        out2, bg = prev.to(DEV).clone(), bprev.to(DEV).clone()
        ops.gemm(dyd, xd, out2, trans_a=True, trans_b=False, accumulate=True, split_k=split, bias_grad=bg)
        assert torch.equal(out, out2), f"gemm_tn {M}x{N}x{K} split {split}: fused bias gradient changed dW / not reproducible"
        check(bg, bref, f"fused bias gradient {M}x{N}x{K} split {split}", fro=4e-3, mx=3e-2)


CONV_CASES = [   # B, H, W, Cin, Cout, stride : the resnet convs that dominate the conv time
    (4, 128, 128, 320, 320, 1),     # down_blocks.0 / up_blocks.2 (7 per micro-step)
    (4, 32, 32, 1280, 1280, 1),     # down_blocks.2 / mid / up_blocks.0 (10 per micro-step)
    (4, 64, 64, 640, 640, 2),       # downsampler
    (4, 32, 32, 2560, 1280, 1),     # up_blocks.0.resnets.0.conv1: the largest weight (29.5 M elements)
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", CONV_CASES, ids=lambda v: str(v))
def test_conv_products(ops, B, H, W, Cin, Cout, stride):
    x, w, b = rnd(B, H, W, Cin, seed=9), rnd(Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5, seed=10), rnd(Cout, seed=11)
    xn = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wn = w.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.conv2d(xn, wn, b.float(), stride=stride, padding=1)
    Ho, Wo = y.shape[2], y.shape[3]
    rb, res = rnd(B, Cout, seed=12), rnd(B, Ho, Wo, Cout, seed=13)
    xd, wd = x.to(DEV), w.to(DEV)
    out = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=DEV)
    ops.conv_fwd(xd, wd, out, stride=stride, bias=b.to(DEV), rowbias=rb.to(DEV), residual=res.to(DEV))
    check(out, (y + rb.float()[:, :, None, None]).permute(0, 2, 3, 1) + res.float(), f"conv_fwd {B,H,W,Cin,Cout,stride}")
    dy = rnd(B, Ho, Wo, Cout, seed=14)
    y.backward(dy.float().permute(0, 3, 1, 2))
    dyd = dy.to(DEV)
    wt = w.permute(3, 1, 2, 0).contiguous().to(DEV)            # W'[Cin][3][3][Cout]: the NT-form dgrad the model uses
    dx = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device=DEV)
    ops.conv_dgrad_wt(dyd, wt, dx, stride=stride)
    check(dx, xn.grad.permute(0, 2, 3, 1), f"conv_dgrad {B,H,W,Cin,Cout,stride}")
    prev, bprev = rnd(Cout, 3, 3, Cin, scale=0.05, seed=15), rnd(Cout, scale=0.1, seed=16)
    dw, bg = prev.to(DEV).clone(), bprev.to(DEV).clone()
    seg = torch.empty(B, Cout, dtype=torch.bfloat16, device=DEV)
    ops.conv_wgrad(dyd, xd, dw, stride=stride, accumulate=True, split_k=0, bias_grad=bg, seg_grad=seg)
    check(dw, prev.float() + wn.grad.permute(0, 2, 3, 1), f"conv_wgrad {B,H,W,Cin,Cout,stride}")
    sums = dy.float().sum((1, 2))
    check(bg, bprev.float() + sums.sum(0), "conv fused bias gradient", fro=4e-3, mx=3e-2)
    check(seg, sums, "conv fused per-sample channel sums (time-embedding gradient)", fro=4e-3, mx=3e-2)
    dw2, bg2 = prev.to(DEV).clone(), bprev.to(DEV).clone()
    ops.conv_wgrad(dyd, xd, dw2, stride=stride, accumulate=True, split_k=0, bias_grad=bg2, seg_grad=seg)
    assert torch.equal(dw, dw2) and torch.equal(bg, bg2), "conv_wgrad not reproducible"


@pytest.mark.parametrize("B,heads,Tq,Tk", [(4, 20, 1024, 1024), (4, 10, 4096, 4096), (4, 20, 1024, 77), (4, 10, 4096, 77)],
                         ids=lambda v: str(v))
def test_attention_at_model_shapes(ops, B, heads, Tq, Tk):
    C = heads * 64
    if Tq == Tk:
        qkv = rnd(B, Tq, 3 * C, seed=17)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        qkvd = qkv.to(DEV)
        qd, kd, vd = qkvd[..., :C], qkvd[..., C:2 * C], qkvd[..., 2 * C:]
    else:
        q, kv = rnd(B, Tq, C, seed=18), rnd(B, Tk, 2 * C, seed=19)
        k, v = kv[..., :C], kv[..., C:]
        qd, kvd = q.to(DEV), kv.to(DEV)
        kd, vd = kvd[..., :C], kvd[..., C:]
    do = rnd(B, Tq, C, seed=20)
    o = torch.empty(B, Tq, C, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * heads * Tq, dtype=torch.float32, device=DEV)
    ops.attn_fwd(qd, kd, vd, o, lse, heads, 0.125)
    dq = torch.empty(B, Tq, C, dtype=torch.bfloat16, device=DEV)
    dkv = torch.empty(B, Tk, 2 * C, dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(B * heads * Tq, dtype=torch.float32, device=DEV)
    ops.attn_bwd(qd, kd, vd, o, do.to(DEV), lse, delta, dq, dkv[..., :C], dkv[..., C:], heads, 0.125)
    dq2, dkv2 = torch.empty_like(dq), torch.empty_like(dkv)
    ops.attn_bwd(qd, kd, vd, o, do.to(DEV), lse, delta, dq2, dkv2[..., :C], dkv2[..., C:], heads, 0.125)
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2), "attention backward not reproducible"
    # fp32 reference one batch element at a time (the 4096^2 score matrices of one element are 1.3 GB in fp32)
    for b in range(B):
        qf = q[b].float().reshape(Tq, heads, 64).transpose(0, 1).requires_grad_(True)
        kf = k[b].float().reshape(Tk, heads, 64).transpose(0, 1).requires_grad_(True)
        vf = v[b].float().reshape(Tk, heads, 64).transpose(0, 1).requires_grad_(True)
        s = (qf @ kf.transpose(-1, -2)) * 0.125
        o_ref = torch.softmax(s, dim=-1) @ vf
        lse_ref = torch.logsumexp(s, dim=-1) * math.log2(math.e)
        o_ref.backward(do[b].float().reshape(Tq, heads, 64).transpose(0, 1))
        check(o[b], o_ref.transpose(0, 1).reshape(Tq, C), f"attn_fwd {B,heads,Tq,Tk} [{b}]")
        check(lse.view(B, heads, Tq)[b], lse_ref, "attn lse", fro=1e-4, mx=1e-3)
        check(dq[b], qf.grad.transpose(0, 1).reshape(Tq, C), f"attn dq [{b}]", fro=8e-3, mx=3e-2)
        check(dkv[b][..., :C], kf.grad.transpose(0, 1).reshape(Tk, C), f"attn dk [{b}]", fro=8e-3, mx=3e-2)
        check(dkv[b][..., C:], vf.grad.transpose(0, 1).reshape(Tk, C), f"attn dv [{b}]", fro=8e-3, mx=3e-2)
        del s, o_ref, qf, kf, vf
