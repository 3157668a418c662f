"""How much noise do the bf16 dataflows carry where GroupNorm sees it, and what does it do to the gradient?  cfg1 (eps, 512x512, B=1).
For a few GroupNorms along the network: input h, output act = SiLU(GN(h)) (where fused) and their gradients, of (a) the HIP path and
(b) the oracle in the reference's bf16-autocast arithmetic, both against the all-fp32 oracle: slope - 1 of the projection on the fp32
tensor (a COHERENT gain), relative size of the orthogonal part (incoherent noise), and the group statistics' rstd ratio.
Theory under test: incoherent noise n on h inflates the group variance, sigma'^2 = sigma^2 + sigma_n^2, so the
normalised activations -- and, through the same rstd, the gradient that passes back -- lose sigma_n^2 / (2 sigma^2) of their coherent part.
usage: python tools/act_noise.py"""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_fullsize_gpu import _micro_inputs
from oracle.unet_ref import SDXL_BASE as OCFG, init_params, RefUNet
from oracle.step_ref import RefTrainer
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
DEV = 'cuda:0'
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 64))
LAT = int(os.environ.get('LAT', '64'))
NAMES = ['down_blocks.0.resnets.0.norm1', 'down_blocks.1.resnets.1.norm2', 'mid_block.resnets.0.norm1', 'up_blocks.0.resnets.2.norm1', 'up_blocks.1.resnets.0.norm1',
         'up_blocks.2.resnets.0.norm1', 'up_blocks.2.resnets.2.norm1', 'up_blocks.2.resnets.2.norm2', 'conv_norm_out']
params = {k: v.bfloat16().float() for k, v in init_params(OCFG, seed=1234).items()}
m = _micro_inputs('epsilon', 1, LAT, LAT, 77, 1, [417], seed=42)[0]

def run_oracle(bf16):
    cap = {}
    orig = RefUNet._gn
    def patched(self, x, name, eps):
        y = orig(self, x, name, eps)
        if name in NAMES:
            rec = cap.setdefault(name, {})
            rec['h'] = x.detach().float().clone(); rec['gn'] = y.detach().float().clone()
            if x.requires_grad: x.register_hook(lambda g, rec=rec: rec.__setitem__('dh', g.detach().float().clone()))
            y.register_hook(lambda g, rec=rec: rec.__setitem__('dgn', g.detach().float().clone()))
        return y
    RefUNet._gn = patched
    try:
        tr = RefTrainer(OCFG, params, mode='epsilon', bf16=bf16, ga=1, clip=1.0)
        tr.micro_step(*m[:6], jitter=m[6])
        g = {k: v.float().clone() for k, v in tr.grads().items() if k.endswith('conv1.weight') or k.endswith('conv2.weight') or k == 'conv_out.weight'}
        del tr
    finally:
        RefUNet._gn = orig
    return cap, g
c32, g32 = run_oracle(False)
c16, g16 = run_oracle(True)
print('oracles done', flush=True)
unet = AozoraUNet(SDXL_BASE, DEV); unet.load_state_dict(params)
hcap = {}
orig = AozoraUNet.groupnorm
def patched(self, x, geom, prefix, eps, silu):
    y = orig(self, x, geom, prefix, eps, silu)
    if prefix in NAMES: hcap[prefix] = (x, y, geom, silu)
    return y
AozoraUNet.groupnorm = patched
step = TrainStep(unet, mode='epsilon', grad_accum=1, use_graph=False)
unet.zero_grad()
step.micro_step(m[0].to(DEV), m[1].to(DEV), m[2], m[3].to(DEV), m[4].to(DEV), m[5].to(DEV), m[6])
torch.cuda.synchronize(); unet.expose_grads()
def nchw(t, B, H, W): return t.float().view(B, H, W, -1).permute(0, 3, 1, 2).cpu()
def cmp(a, ref):
    a, ref = a.flatten().double(), ref.flatten().double()
    sl = float(a @ ref) / float(ref @ ref)
    return sl - 1, float((a - sl * ref).norm() / ref.norm())
def rstd_ratio(h, href, G=32):
    B, C = h.shape[:2]
    v = h.reshape(B, G, -1).var(dim=2, unbiased=False); vr = href.reshape(B, G, -1).var(dim=2, unbiased=False)
    return float(((vr + 1e-5) / (v + 1e-5)).sqrt().mean()) - 1
print(f'{"norm":34s} | {"flow":5s} | h: slope-1   noise  | rstd-1     | act: slope-1  noise | d(act): slope-1 noise | d(h): slope-1  noise')
for n in NAMES:
    x, y, (B, H, W), silu = hcap[n]
    ref = c32[n]
    act32 = F.silu(ref['gn']) if silu else ref['gn']
    dact32 = None
    if 'dgn' in ref:       # gradient wrt the GN output; through SiLU for the fused form: d(act) = dgn / silu'(gn)  -> compare d(gn) instead
        pass
    rows = []
    hh = nchw(x.t, B, H, W); ah = nchw(y.t, B, H, W)
    dah = nchw(y.g, B, H, W) if y.g is not None else None
    dhh = nchw(x.g, B, H, W)[:, :hh.shape[1]] if x.g is not None else None
    o = c16[n]
    act16 = (F.silu(o['gn'].bfloat16()).float() if silu else o['gn'])
    for flow, h, act, dh in (('HIP', hh, ah, dhh), ('bf16', o['h'], act16, o.get('dh'))):
        s_h = cmp(h, ref['h']); s_a = cmp(act, act32)
        s_dh = cmp(dh, ref['dh']) if (dh is not None and 'dh' in ref and dh.shape == ref['dh'].shape) else (float('nan'), float('nan'))
        print(f'{n:34s} | {flow:5s} | {s_h[0]:+.2e} {s_h[1]:.2e} | {rstd_ratio(h, ref["h"]):+.2e} | {s_a[0]:+.2e} {s_a[1]:.2e} | {"":21s} | {s_dh[0]:+.2e} {s_dh[1]:.2e}', flush=True)
gh = {n: p.grad.float().cpu() for n, p in unet.named_parameters() if n in g32}
print('weight gradients (slope - 1, orthogonal part) vs fp32')
for n in ['conv_out.weight', 'up_blocks.2.resnets.2.conv2.weight', 'up_blocks.2.resnets.2.conv1.weight', 'up_blocks.2.resnets.0.conv1.weight', 'up_blocks.1.resnets.0.conv1.weight',
          'mid_block.resnets.0.conv1.weight', 'down_blocks.0.resnets.0.conv1.weight']:
    a, b = cmp(gh[n], g32[n]), cmp(g16[n], g32[n])
    print(f'  {n:44s} HIP {a[0]:+.2e} {a[1]:.2e}   bf16 {b[0]:+.2e} {b[1]:.2e}')
