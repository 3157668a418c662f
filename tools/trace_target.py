"""Target for a rocprofv3 --kernel-trace run: 8 micro-steps (B=4, 1024^2, two streams, launch tape) and nothing else."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
for i in range(4): step.micro_step(*batch); step.synchronize()
t0 = time.time()
for i in range(4): step.micro_step(*batch)
step.synchronize(); print(f'micro-step {(time.time()-t0)/4*1e3:.1f} ms', flush=True)
