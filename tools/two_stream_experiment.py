# Timing-only experiment (numerics of the concurrent run are NOT valid: both executors share scratch workspaces).
import sys, time, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
def mk():
    u = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(u); return u
uA, uB = mk(), mk()
b4 = bench.synthetic_batch(0, 0, 0, 4, dev)
b2a = [t[:2] if torch.is_tensor(t) else t for t in b4]
b2b = [t[2:] if torch.is_tensor(t) else t for t in b4]
s4 = TrainStep(uA, grad_accum=8, use_graph=True)
for _ in range(3): s4.micro_step(*b4)
s4.synchronize()
t0 = time.time()
for _ in range(5): s4.micro_step(*b4)
s4.synchronize(); print(f'one context, B=4: {(time.time()-t0)/5*1e3:.1f} ms per micro-step', flush=True)
sa, sb = TrainStep(uA, grad_accum=8, use_graph=True), TrainStep(uB, grad_accum=8, use_graph=True)
for _ in range(3): sa.micro_step(*b2a); sa.synchronize(); sb.micro_step(*b2b); sb.synchronize()
t0 = time.time()
for _ in range(5): sa.micro_step(*b2a)
sa.synchronize(); print(f'one context, B=2: {(time.time()-t0)/5*1e3:.1f} ms per half micro-step', flush=True)
t0 = time.time()
for _ in range(5):
    sa.micro_step(*b2a); sb.micro_step(*b2b)
sa.synchronize(); sb.synchronize(); print(f'two contexts x B=2 concurrently: {(time.time()-t0)/5*1e3:.1f} ms per (B=4-equivalent) micro-step', flush=True)
# pipelined whole micro-steps (B=4 each) on two streams
sA4, sB4 = s4, TrainStep(uB, grad_accum=8, use_graph=True)
for _ in range(3): sB4.micro_step(*b4); sB4.synchronize()
t0 = time.time()
for _ in range(4):
    sA4.micro_step(*b4); sB4.micro_step(*b4)
sA4.synchronize(); sB4.synchronize(); print(f'two contexts x B=4 concurrently: {(time.time()-t0)/8*1e3:.1f} ms per micro-step', flush=True)
