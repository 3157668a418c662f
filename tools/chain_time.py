import sys, time, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
def run(tag):
    step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
    for i in range(2): step.micro_step(*batch); step.synchronize()
    t0 = time.time()
    for i in range(5): step.micro_step(*batch)
    step.synchronize(); print(f'{tag}: {(time.time()-t0)/5*1e3:.1f} ms', flush=True)
run('all trainable (2 streams)')
for p in unet.parameters(): p.requires_grad = False
run('all frozen: forward + data-gradient chain only')
