"""Systematic magnitude bias of single kernels against fp32 torch on identical bf16 inputs: norm(out) / norm(ref) - 1 and the relative
L2 of the difference (a truncation or a dropped contribution shows as a norm deficit of ~1e-3 that the per-kernel Frobenius gates
do not see).  usage: python tools/norm_bias.py"""
import math, sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
DEV = 'cuda:0'
g = torch.Generator().manual_seed(3)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).bfloat16()
def rep(name, out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    print(f'{name:46s} norm ratio - 1 = {out.norm().item() / ref.norm().item() - 1:+.2e}   rel L2 = {(out - ref).norm().item() / ref.norm().item():.2e}', flush=True)

for (B, H, W, Cin, Cout) in ((1, 64, 64, 320, 320), (1, 64, 64, 640, 320), (1, 32, 32, 640, 640)):
    x, dy, w = rnd(B, H, W, Cin), rnd(B, H, W, Cout, scale=0.05), rnd(Cout, 3, 3, Cin, scale=0.03)
    xn = x.float().permute(0, 3, 1, 2).requires_grad_(True); wn = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    y = F.conv2d(xn, wn, padding=1); y.backward(dy.float().permute(0, 3, 1, 2))
    yo = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=DEV); ops.conv_fwd(x.to(DEV), w.to(DEV), yo)
    rep(f'conv_fwd {B}x{H}x{W} {Cin}->{Cout}', yo, y.permute(0, 2, 3, 1))
    dw = torch.zeros(Cout, 3, 3, Cin, dtype=torch.bfloat16, device=DEV); bg = torch.zeros(Cout, dtype=torch.bfloat16, device=DEV)
    ops.conv_wgrad(dy.to(DEV), x.to(DEV), dw, accumulate=True, split_k=0, bias_grad=bg)
    rep(f'conv_wgrad', dw, wn.grad.permute(0, 2, 3, 1)); rep('conv bias grad', bg, dy.float().sum((0, 1, 2)))
    dx = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device=DEV)
    try:
        ops.conv_dgrad(dy.to(DEV), w.to(DEV), dx)
        rep('conv_dgrad', dx, xn.grad.permute(0, 2, 3, 1))
    except Exception as e:
        print('conv_dgrad: skipped', type(e).__name__, e)
for (B, HW, C, G, silu) in ((1, 4096, 320, 32, True), (1, 4096, 640, 32, True), (1, 1024, 1280, 32, True), (1, 4096, 320, 32, False)):
    x = (rnd(B, HW, C).float() * 1.5 + 0.3).bfloat16(); gamma, beta, dy = (1 + 0.2 * rnd(C).float()).bfloat16(), rnd(C, scale=0.2), rnd(B, HW, C, scale=0.05)
    xf = x.float().permute(0, 2, 1).requires_grad_(True); gf, bfl = gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    y = F.group_norm(xf, G, gf, bfl, 1e-5); y = F.silu(y) if silu else y; y.backward(dy.float().permute(0, 2, 1))
    xd, yd = x.to(DEV), torch.empty(B, HW, C, dtype=torch.bfloat16, device=DEV); stats = torch.empty(B * G * 2, dtype=torch.float32, device=DEV)
    ops.groupnorm_fwd(xd, gamma.to(DEV), beta.to(DEV), yd, stats, G, 1e-5, silu)
    rep(f'gn_fwd {B}x{HW}x{C} silu={silu}', yd, y.permute(0, 2, 1))
    dx = torch.empty(B, HW, C, dtype=torch.bfloat16, device=DEV); dg = torch.zeros(C, dtype=torch.bfloat16, device=DEV); db = torch.zeros(C, dtype=torch.bfloat16, device=DEV)
    ops.groupnorm_bwd(xd, gamma.to(DEV), beta.to(DEV), stats, dy.to(DEV), dx, dg, db, G, silu)
    rep('gn_bwd dx', dx, xf.grad.permute(0, 2, 1)); rep('gn_bwd dgamma', dg, gf.grad); rep('gn_bwd dbeta', db, bfl.grad)
for (M, C) in ((4096, 640), (1024, 1280)):
    x = (rnd(M, C).float() * 2 - 0.5).bfloat16(); gamma, beta, dy = (1 + 0.2 * rnd(C).float()).bfloat16(), rnd(C, scale=0.2), rnd(M, C, scale=0.05)
    xf, gf, bfl = x.float().requires_grad_(True), gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    y = F.layer_norm(xf, (C,), gf, bfl, 1e-5); y.backward(dy.float())
    xd, yd = x.to(DEV), torch.empty(M, C, dtype=torch.bfloat16, device=DEV); stats = torch.empty(2 * M, dtype=torch.float32, device=DEV)
    ops.layernorm_fwd(xd, gamma.to(DEV), beta.to(DEV), yd, stats); rep(f'ln_fwd {M}x{C}', yd, y)
    dx = torch.empty(M, C, dtype=torch.bfloat16, device=DEV); dg = torch.zeros(C, dtype=torch.bfloat16, device=DEV); db = torch.zeros(C, dtype=torch.bfloat16, device=DEV)
    ops.layernorm_bwd(xd, gamma.to(DEV), stats, dy.to(DEV), dx, dg, db)
    rep('ln_bwd dx', dx, xf.grad); rep('ln_bwd dgamma', dg, gf.grad)
for (M, N, K) in ((1280, 1280, 4096), (320, 320, 4096)):
    dy, x = rnd(K, M, scale=0.05), rnd(K, N); out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dy.to(DEV), x.to(DEV), out, trans_a=True, trans_b=False, accumulate=True, split_k=0)
    rep(f'gemm_tn {M}x{N}x{K}', out, dy.float().t() @ x.float())
    a, w = rnd(K, N), rnd(M, N, scale=0.05); out = torch.empty(K, M, dtype=torch.bfloat16, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), out); rep(f'gemm_nt {K}x{M}x{N}', out, a.float() @ w.float().t())
