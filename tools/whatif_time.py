"""What would the two-stream micro-step gain if a class of launches cost NOTHING?  Timing-only upper bounds (the skipped kernels'
results are simply missing): the named C-ABI entry points are dropped before they reach the library -- and therefore the launch
tape -- optionally only for some shapes.  usage: python tools/whatif_time.py <case> [<case> ...]
cases: base | tn_small (weight gradients with M*N <= 2560*2048) | tn_all | conv_wgrad | ln_bwd | ln_fwd | gn | attn_bwd | attn_fwd | geglu_bwd | reduce (no-op here)"""
import statistics, sys, time
import torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd import _lib as L_
from aozora_sdxl_training_amd.unet import AozoraUNet, ExecPolicy
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep

CASES = {
    'base': {},
    'tn_small': {'az_gemm_wgrad_bias_bf16': lambda a: a[0] * a[1] <= 2560 * 2048},
    'tn_1280sq': {'az_gemm_wgrad_bias_bf16': lambda a: a[0] == 1280 and a[1] == 1280},
    'tn_all': {'az_gemm_wgrad_bias_bf16': lambda a: True},
    'conv_wgrad': {'az_conv2d_wgrad_bias_bf16': lambda a: True},
    'ln_bwd': {'az_layernorm_bwd': lambda a: True, 'az_layernorm_bwd_ex': lambda a: True, 'az_layernorm_bwd_partial': lambda a: True},
    'ln_fwd': {'az_layernorm_fwd': lambda a: True},
    'gn': {'az_groupnorm_fwd': lambda a: True, 'az_groupnorm_bwd': lambda a: True, 'az_groupnorm_bwd_ex': lambda a: True},
    'attn_bwd': {'az_attn_bwd': lambda a: True},
    'attn_fwd': {'az_attn_fwd': lambda a: True},
    'geglu_bwd': {'az_geglu_bwd': lambda a: True},
}
dev = torch.device('cuda', 0)
lib = L_.lib()
orig_call = L_._Lib.call
skip = {}
def call(self, name, *args):
    f = skip.get(name)
    if f is not None and f(args):
        return 0
    return orig_call(self, name, *args)
L_._Lib.call = call
for case in sys.argv[1:] or ['base']:
    skip.clear(); skip.update(CASES[case])
    unet = AozoraUNet(SDXL_BASE, dev, policy=ExecPolicy()); bench.init_weights_on_device(unet)
    batch = bench.synthetic_batch(0, 0, 0, 4, dev)
    step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
    for _ in range(3):
        step.micro_step(*batch); step.synchronize()
    ts = []
    for r in range(4):
        t0 = time.perf_counter()
        for _ in range(4): step.micro_step(*batch)
        step.synchronize()
        ts.append((time.perf_counter() - t0) / 4 * 1e3)
    print(f'{case:12s}: median {statistics.median(ts):.2f} ms (min {min(ts):.2f})', flush=True)
    del step, unet
    torch.cuda.empty_cache()
