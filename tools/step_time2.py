import sys, time, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
for graph in (True, False):
    for conc in (False, True):
        unet.concurrent_wgrad = conc
        step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=graph)
        for i in range(3):
            unet.zero_grad(); l = step.micro_step(*batch); step.synchronize()
        t0 = time.time()
        for i in range(6): l = step.micro_step(*batch)
        step.synchronize(); dt = (time.time() - t0) / 6
        print(f'graph={graph} concurrent_wgrad={conc}: micro-step {dt*1e3:.1f} ms  loss {l.item():.5f}', flush=True)
