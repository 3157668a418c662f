"""Repeat the same full-size micro-step N times (two streams, launch tape) and compare loss + gradient buffer bit for bit;
any hazard between the data-gradient chain and the parameter-gradient branch would show up as a mismatch."""
import sys, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batches = [bench.synthetic_batch(0, m, 0, 4, dev) for m in range(2)]
step = TrainStep(unet, mode='epsilon', grad_accum=2, use_graph=False)
ref = None; bad = 0
for i in range(N):
    unet.zero_grad()
    l = [step.micro_step(*batches[m]) for m in range(2)]      # back-to-back, no sync in between
    torch.cuda.synchronize()
    cur = (l[1].item(), unet.gflat.clone())
    if ref is None: ref = cur
    elif cur[0] != ref[0] or not torch.equal(cur[1], ref[1]): bad += 1
print(f'{N} repetitions of a 2-micro-step window: {bad} mismatches; loss {ref[0]:.6f}')
sys.exit(1 if bad else 0)
