"""Two-stream micro-step time (B=4, 1024^2, all trainable) under one executor policy / option set; run the variants to compare
one after the other on the same box (spread between runs on one box: +-0.2 ms).
usage: python tools/policy_time.py "field=V,OPTION=V"      lower-case = unet.ExecPolicy field, upper-case = library option"""
import statistics, sys, time
import torch
sys.path.insert(0, '.')
import os
from aozora_sdxl_training_amd import _lib as _L
if os.environ.get('AZ_LIB'):
    _L.LIB_PATH = os.path.abspath(os.environ['AZ_LIB'])      # A/B of two builds of the library (one process each, same box)
import bench
from aozora_sdxl_training_amd._lib import set_option
from aozora_sdxl_training_amd.unet import AozoraUNet, ExecPolicy
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep

spec = dict(kv.split('=') for kv in (sys.argv[1] if len(sys.argv) > 1 else '').split(',') if kv)
pol = ExecPolicy()
for k, v in spec.items():
    if k.isupper():
        set_option(k, int(v))
    else:
        setattr(pol, k, type(getattr(pol, k))(int(v)))
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev, policy=pol); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
for _ in range(3):
    step.micro_step(*batch); step.synchronize()
ts = []
for r in range(4):
    t0 = time.perf_counter()
    for _ in range(4):
        step.micro_step(*batch)
    step.synchronize()
    ts.append((time.perf_counter() - t0) / 4 * 1e3)
print(f'{spec or "defaults"}: median {statistics.median(ts):.2f} ms (min {min(ts):.2f}) of {[round(x, 2) for x in ts]}', flush=True)
