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
for i in range(2): step.micro_step(*batch); step.synchronize()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.time(); step.micro_step(*batch); t1 = time.time(); step.synchronize(); t2 = time.time()
    print(f'host issue {1e3*(t1-t0):.1f} ms, gpu done after {1e3*(t2-t0):.1f} ms', flush=True)
t0 = time.time()
for i in range(5): step.micro_step(*batch)
t1 = time.time(); step.synchronize(); t2 = time.time()
print(f'5 back-to-back: host {1e3*(t1-t0)/5:.1f} ms/step, total {1e3*(t2-t0)/5:.1f} ms/step')
