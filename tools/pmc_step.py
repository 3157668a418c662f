"""One full-size micro-step (B=4, 1024^2) issued on a single stream, for rocprofv3 --pmc passes (per-kernel HBM traffic)."""
import sys, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
unet.concurrent_wgrad = False
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
step.stream = torch.cuda.current_stream()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    l = step.micro_step(*batch)
torch.cuda.synchronize()
print('loss', l.item())
