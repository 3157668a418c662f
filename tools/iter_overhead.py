"""Per-iteration overhead of the optimizer boundary: T(iteration) - GA x T(micro-step) at GA = 1 / 2 / 8 on one GPU (the GA = 1
case is what every rank of an 8-GPU run sees: local batch 4, one micro-step per optimizer step)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
from aozora_sdxl_training_amd.dist import ShardedRaven
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
opt = ShardedRaven(unet, lr=8e-7, clip_grad_norm=1.0)
for ga in (8, 2, 1):
    step = TrainStep(unet, mode='epsilon', grad_accum=ga, use_graph=False)
    def iteration():
        for m in range(ga):
            if m == max(0, ga - 2): opt.prefetch()
            step.micro_step(*batch)
        opt.step(); opt.zero_grad()
    for _ in range(3): step.micro_step(*batch)
    step.synchronize(); opt.zero_grad()
    t0 = time.time()
    for _ in range(4): step.micro_step(*batch)
    step.synchronize(); tm = (time.time() - t0) / 4; opt.zero_grad()
    iteration(); torch.cuda.synchronize()
    n = 3 if ga == 8 else 8
    t0 = time.time()
    for _ in range(n): iteration()
    torch.cuda.synchronize(); ti = (time.time() - t0) / n
    print(f'GA {ga}: micro-step {tm*1e3:.1f} ms, iteration {ti*1e3:.1f} ms, overhead {1e3*(ti - ga*tm):.1f} ms', flush=True)
