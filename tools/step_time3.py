import sys, time, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
for i in range(3):
    unet.zero_grad(); l = step.micro_step(*batch); step.synchronize()
for i in range(4):
    t0 = time.time(); l = step.micro_step(*batch); t1 = time.time(); step.synchronize(); t2 = time.time()
    print(f'eager: CPU issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms', flush=True)
