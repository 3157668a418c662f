"""Where does the HIP path's distance to the all-fp32 oracle come from?  cfg1 (epsilon, 512x512, B=1, t=417) on the full SDXL-base
UNet: per top-level block, the gradient norm and the relative L2 of the gradient difference of (a) the HIP path and (b) the oracle in
the reference's bf16-autocast arithmetic, both against the all-fp32 oracle.  usage: python tools/fp32_gap.py [out.json]"""
import json, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_fullsize_gpu import _micro_inputs
from oracle.unet_ref import SDXL_BASE as OCFG, init_params
from oracle.step_ref import RefTrainer
from aozora_sdxl_training_amd import _lib as _L
if os.environ.get('AZ_LIB'):
    _L.LIB_PATH = os.path.abspath(os.environ['AZ_LIB'])      # experiment builds of the library (one process each)
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep

DEV = 'cuda:0'
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 64))
params = {k: v.bfloat16().float() for k, v in init_params(OCFG, seed=1234).items()}
LAT = int(os.environ.get('LAT', '64'))                     # latent size: 64 = cfg1 (512 px), 128 = cfg2's resolution
m = _micro_inputs('epsilon', 1, LAT, LAT, 77, 1, [417], seed=42)[0]
t0 = time.time()
ref = RefTrainer(OCFG, params, mode='epsilon', bf16=False, ga=1, clip=1.0)
l32 = ref.micro_step(*m[:6], jitter=m[6]); g32 = {k: v.float().clone() for k, v in ref.grads().items()}; p32 = ref.last_pred.float().clone(); del ref
ref = RefTrainer(OCFG, params, mode='epsilon', bf16=True, ga=1, clip=1.0)
l16 = ref.micro_step(*m[:6], jitter=m[6]); g16 = {k: v.float().clone() for k, v in ref.grads().items()}; p16 = ref.last_pred.float().clone(); del ref
print(f'oracles: {time.time() - t0:.0f} s; loss fp32 {l32:.6f} bf16 {l16:.6f}', flush=True)
unet = AozoraUNet(SDXL_BASE, DEV); unet.load_state_dict(params)
step = TrainStep(unet, mode='epsilon', grad_accum=1, use_graph=False)
unet.zero_grad()
lh = step.micro_step(m[0].to(DEV), m[1].to(DEV), m[2], m[3].to(DEV), m[4].to(DEV), m[5].to(DEV), m[6]).item()
unet.expose_grads()
# the prediction and the loss residual r = pred - target (d(loss)/d(pred) is proportional to it): slope along the fp32 residual and what is orthogonal to it
from oracle.step_ref import make_noisy_and_target, ddpm_alphas_cumprod
_, target, _ = make_noisy_and_target('epsilon', m[0].float(), m[1], m[2], ddpm_alphas_cumprod())
ph = list(step._buckets.values())[0].pred.float().view(1, LAT, LAT, -1)[..., :4].permute(0, 3, 1, 2).cpu()
def resid(name, p):
    r, r0 = (p - target.float()).flatten(), (p32 - target.float()).flatten()
    sl = (r @ r0).item() / (r0 @ r0).item()
    print(f'{name}: |pred - pred32| / |pred32| = {(p - p32).norm().item() / p32.norm().item():.2e};  residual: slope - 1 = {sl - 1:+.2e}, orthogonal part / |r32| = {(r - sl * r0).norm().item() / r0.norm().item():.2e}')
resid('HIP        ', ph); resid('bf16 oracle', p16)
gh = {n: p.grad.float().cpu() for n, p in unet.named_parameters()}
def group(name):
    p = name.split('.')
    if p[0] in ('down_blocks', 'up_blocks'):
        kind = 'attn' if p[2] == 'attentions' else ('res' if p[2] == 'resnets' else p[2])
        return f'{p[0]}.{p[1]}.{kind}'
    if p[0] == 'mid_block': return 'mid_block.' + ('attn' if p[1] == 'attentions' else 'res')
    return p[0]
def kind(name):
    for k in ('attn1.to_q', 'attn1.to_k', 'attn1.to_v', 'attn1.to_out', 'attn2.to_q', 'attn2.to_k', 'attn2.to_v', 'attn2.to_out', 'ff.net.0', 'ff.net.2',
              'norm1', 'norm2', 'norm3', 'proj_in', 'proj_out', 'conv1', 'conv2', 'conv_shortcut', 'time_emb_proj'):
        if k in name: return ('T.' if 'transformer_blocks' in name or 'proj_' in name else 'R.') + k + ('.bias' if name.endswith('bias') else '')
    return 'other'
def table(keyfn):
    acc = {}
    for n in g32:
        a = acc.setdefault(keyfn(n), [0.0] * 5)
        a[0] += float(g32[n].double().pow(2).sum()); a[1] += float(gh[n].double().pow(2).sum()); a[2] += float(g16[n].double().pow(2).sum())
        a[3] += float((gh[n] - g32[n]).double().pow(2).sum()); a[4] += float((g16[n] - g32[n]).double().pow(2).sum())
    rows = {}
    for k, a in acc.items():
        n32 = math.sqrt(a[0])
        rows[k] = dict(norm_fp32=n32, hip_norm_rel=(math.sqrt(a[1]) - n32) / n32, bf16_norm_rel=(math.sqrt(a[2]) - n32) / n32,
                       hip_diff_rel_l2=math.sqrt(a[3]) / n32, bf16_diff_rel_l2=math.sqrt(a[4]) / n32, share_of_sq=a[0])
    tot = sum(r['share_of_sq'] for r in rows.values())
    for r in rows.values(): r['share_of_sq'] /= tot
    return rows
def show(title, rows):
    print(title)
    print(f'  {"group":34s} {"|g| fp32":>9s} {"share":>6s} {"HIP dnorm":>10s} {"bf16 dnorm":>10s} {"HIP dL2":>9s} {"bf16 dL2":>9s}')
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]['share_of_sq']):
        print(f'  {k:34s} {r["norm_fp32"]:9.4f} {r["share_of_sq"]:6.3f} {r["hip_norm_rel"]:+10.2e} {r["bf16_norm_rel"]:+10.2e} {r["hip_diff_rel_l2"]:9.2e} {r["bf16_diff_rel_l2"]:9.2e}')
gn = lambda g: math.sqrt(sum(float(v.double().pow(2).sum()) for v in g.values()))
n32, nh, n16 = gn(g32), gn(gh), gn(g16)
print(f'loss: HIP {lh:.6f}  global |g|: fp32 {n32:.5f}  HIP {nh:.5f} ({(nh - n32) / n32:+.2e})  bf16 oracle {n16:.5f} ({(n16 - n32) / n32:+.2e})')
by_block, by_kind = table(group), table(kind)
show('by top-level block', by_block); show('by layer kind', by_kind)
# per-tensor, in backward order along the high-resolution tail: bias gradients = channel sums of each layer's dY
tail = [n for n in g32 if n.endswith('.bias') and (n.startswith('conv_') or n.startswith('up_blocks.2') or n.startswith('up_blocks.1.resnets') or n.startswith('up_blocks.1.upsamplers'))]
order = ['conv_out.bias', 'conv_norm_out.bias'] + [f'up_blocks.2.resnets.{j}.{k}.bias' for j in (2, 1, 0) for k in ('conv_shortcut', 'conv2', 'norm2', 'time_emb_proj', 'conv1', 'norm1')] + \
        ['up_blocks.1.upsamplers.0.conv.bias'] + [f'up_blocks.1.resnets.2.{k}.bias' for k in ('conv_shortcut', 'conv2', 'norm2', 'conv1', 'norm1')]
show('per tensor, backward order (bias gradients)', table(lambda n: n if n in order else '(rest)'))
for n in order:
    if n in g32:
        a, h, b = g32[n], gh[n], g16[n]
        print(f'  {n:50s} HIP {(h.norm() / a.norm() - 1).item():+.2e} bf16 {(b.norm() / a.norm() - 1).item():+.2e}   slope HIP {(h * a).sum().item() / (a * a).sum().item() - 1:+.2e} bf16 {(b * a).sum().item() / (a * a).sum().item() - 1:+.2e}')
if len(sys.argv) > 1:
    json.dump(dict(loss=dict(fp32=l32, bf16=l16, hip=lh), gn=dict(fp32=n32, hip=nh, bf16=n16), by_block=by_block, by_kind=by_kind), open(sys.argv[1], 'w'), indent=1)
