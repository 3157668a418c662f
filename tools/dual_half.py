"""Premise test: does ONE micro-step of B = 4 run faster as TWO independent half-batch (B = 2) micro-steps issued side by side (two
UNet objects -> two data-gradient + two parameter-gradient streams)?  Timing only: the two objects hold separate weight / gradient
buffers.  usage: python tools/dual_half.py [B_total]"""
import statistics, sys, time
import torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet, ExecPolicy
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep

dev = torch.device('cuda', 0)
BT = int(sys.argv[1]) if len(sys.argv) > 1 else 4
def make(B):
    u = AozoraUNet(SDXL_BASE, dev, policy=ExecPolicy()); bench.init_weights_on_device(u)
    return u, TrainStep(u, mode='epsilon', grad_accum=8, use_graph=False), bench.synthetic_batch(0, 0, 0, B, dev)
def timeit(fn, sync, n=4, reps=4):
    for _ in range(3): fn()
    sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        sync()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return statistics.median(ts), ts
uA, sA, bA = make(BT // 2)
uB, sB, bB = make(BT // 2)
def both(): sA.micro_step(*bA); sB.micro_step(*bB)
def sync2(): sA.synchronize(); sB.synchronize()
m, ts = timeit(lambda: sA.micro_step(*bA), sA.synchronize)
print(f'one B={BT // 2} micro-step alone: {m:.2f} ms {[round(x, 2) for x in ts]}', flush=True)
m, ts = timeit(both, sync2)
print(f'two B={BT // 2} micro-steps side by side: {m:.2f} ms per pair {[round(x, 2) for x in ts]}', flush=True)
del uB, sB
torch.cuda.empty_cache()
u4, s4, b4 = make(BT)
m, ts = timeit(lambda: s4.micro_step(*b4), s4.synchronize)
print(f'one B={BT} micro-step: {m:.2f} ms {[round(x, 2) for x in ts]}', flush=True)
