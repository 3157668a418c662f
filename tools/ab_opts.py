"""Same-process interleaved A/B of runtime options on the two-stream micro-step (B=4, 1024^2, all trainable) -- guide rule 24.
usage: python tools/ab_opts.py [rounds] -- "NAME=V,NAME2=V2" "NAME=V" ...   (variant "" = defaults)
Prints ms per micro-step per variant and round, then medians; variants alternate inside each round."""
import statistics
import sys
import time

import torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd._lib import set_option, get_option
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep

args = sys.argv[1:]
rounds = 3
if args and args[0].isdigit():
    rounds = int(args[0]); args = args[1:]
if args and args[0] == '--':
    args = args[1:]
variants = [dict(kv.split('=') for kv in a.split(',') if kv) for a in args] or [{}]
names = sorted({k for v in variants for k in v})
defaults = {k: get_option(k) for k in names}

dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
for i in range(3):
    step.micro_step(*batch); step.synchronize()


def apply(v):
    for k in names:
        set_option(k, int(v.get(k, defaults[k])))


res = [[] for _ in variants]
for r in range(rounds):
    for i, v in enumerate(variants):
        apply(v)
        step.micro_step(*batch); step.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step.micro_step(*batch)
        step.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        res[i].append(ms)
        print(f'round {r} {v or "defaults"}: {ms:.2f} ms', flush=True)
apply({})
for v, r_ in zip(variants, res):
    print(f'MEDIAN {v or "defaults"}: {statistics.median(r_):.2f} ms  (min {min(r_):.2f})')
