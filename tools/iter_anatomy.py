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
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
opt = ShardedRaven(unet, lr=8e-7, clip_grad_norm=1.0)
for i in range(3): step.micro_step(*batch)
step.synchronize(); opt.zero_grad()
def T(fn, name):
    torch.cuda.synchronize(); t0 = time.time(); fn(); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print(f'{name:28s} host {1e3*(t1-t0):7.1f} ms   until GPU idle {1e3*(t2-t0):7.1f} ms', flush=True)
for it in range(2):
    T(lambda: [step.micro_step(*batch) for _ in range(2)], '2 micro-steps')
    T(opt.prefetch, 'prefetch m/v (H2D)')
    T(opt.step, 'opt.step')
    T(opt.zero_grad, 'zero_grad')
    T(unet.refresh_transposed, 'refresh W^T')
    T(lambda: step.micro_step(*batch), 'micro-step after step')
