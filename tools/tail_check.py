"""The last three backward operations of the model on REAL data (cfg1: eps, 512x512, B=1): conv_out's data gradient, the fused
SiLU + GroupNorm backward of conv_norm_out and its parameter gradients, HIP against fp32 torch ON THE HIP PATH'S OWN INPUTS
(captured from the executor), to tell a kernel-level bias on real data from an upstream one.  usage: python tools/tail_check.py"""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_fullsize_gpu import _micro_inputs
from oracle.unet_ref import SDXL_BASE as OCFG, init_params
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
DEV = 'cuda:0'
params = {k: v.bfloat16().float() for k, v in init_params(OCFG, seed=1234).items()}
m = _micro_inputs('epsilon', 1, 64, 64, 77, 1, [417], seed=42)[0]
unet = AozoraUNet(SDXL_BASE, DEV); unet.load_state_dict(params)
cap = {}
orig = AozoraUNet.groupnorm
def patched(self, x, geom, prefix, eps, silu):
    y = orig(self, x, geom, prefix, eps, silu)
    if prefix in ('conv_norm_out', 'up_blocks.2.resnets.2.norm2', 'up_blocks.2.resnets.2.norm1'): cap[prefix] = (x, y, geom, eps)
    return y
AozoraUNet.groupnorm = patched
step = TrainStep(unet, mode='epsilon', grad_accum=1, use_graph=False)
unet.zero_grad()
step.micro_step(m[0].to(DEV), m[1].to(DEV), m[2], m[3].to(DEV), m[4].to(DEV), m[5].to(DEV), m[6])
torch.cuda.synchronize(); unet.expose_grads()
bk = list(step._buckets.values())[0]
def rep(name, out, ref):
    out, ref = out.float().flatten(), ref.float().flatten()
    print(f'{name:56s} norm ratio - 1 = {out.norm().item() / ref.norm().item() - 1:+.2e}  slope - 1 = {(out @ ref).item() / (ref @ ref).item() - 1:+.2e}  rel L2 = {(out - ref).norm().item() / ref.norm().item():.2e}', flush=True)
gr = {n: p.grad for n, p in unet.named_parameters()}
for prefix, nxt in (('conv_norm_out', 'conv_out'), ('up_blocks.2.resnets.2.norm2', 'up_blocks.2.resnets.2.conv2')):
    x, y, (B, H, W), eps = cap[prefix]
    C = x.t.shape[1]
    h = x.t.float().view(B, H, W, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    gam = params[prefix + '.weight'].to(DEV).requires_grad_(True); bet = params[prefix + '.bias'].to(DEV).requires_grad_(True)
    act = F.silu(F.group_norm(h, 32, gam, bet, eps))
    rep(f'{prefix}: fused GN+SiLU forward on the HIP input', y.t.view(B, H, W, C).permute(0, 3, 1, 2), act)
    dact_hip = y.g.float().view(B, H, W, -1).permute(0, 3, 1, 2) if y.g is not None else None
    Wn = params[nxt + '.weight'].to(DEV)
    if prefix == 'conv_norm_out':
        dpred = bk.dpred8[..., :4].float().permute(0, 3, 1, 2).contiguous()
        dact32 = F.conv_transpose2d(dpred, Wn, padding=1)
        if dact_hip is not None: rep(f'{nxt}: data gradient d(act) on the HIP d(pred)', dact_hip, dact32)
        # weight gradient of conv_out on HIP's act (bf16) and dpred
        actb = y.t.float().view(B, H, W, C).permute(0, 3, 1, 2).contiguous().requires_grad_(False)
        wv = Wn.clone().requires_grad_(True); F.conv2d(actb, wv, padding=1).backward(dpred)
        rep(f'{nxt}: weight gradient on HIP act, d(pred)', gr[nxt + '.weight'], wv.grad)
    if dact_hip is not None:
        act.backward(dact_hip.contiguous())
        rep(f'{prefix}: d(gamma) on HIP x, d(act)', gr[prefix + '.weight'], gam.grad)
        rep(f'{prefix}: d(beta) on HIP x, d(act)', gr[prefix + '.bias'], bet.grad)
        if x.g is not None: rep(f'{prefix}: dx on HIP x, d(act) (incl. whatever was added to it)', x.g.view(B, H, W, C).permute(0, 3, 1, 2), h.grad)
