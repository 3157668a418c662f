"""Is the first micro-step of an accumulation window cheaper than the others (it writes the gradients, the others
read-modify-write them)?  zero_grad, then 4 micro-steps, each timed on the data-gradient stream; no optimizer."""
import sys, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
for _ in range(3): step.micro_step(*batch)
step.synchronize()
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(step.stream); return e
for r in range(4):
    unet.zero_grad(); torch.cuda.synchronize()
    marks = []
    for m in range(4):
        marks.append(ev()); step.micro_step(*batch)
    marks.append(ev()); torch.cuda.synchronize()
    print('after zero_grad: ' + ' '.join(f'{marks[i].elapsed_time(marks[i + 1]):.1f}' for i in range(4)), flush=True)
# Does a pause (the chip idle, or busy with HBM-bound work only) buy the next micro-step a higher clock?
import time
g = torch.empty(1 << 30, dtype=torch.bfloat16, device=dev)
for what in ('none', 'idle 9 ms', 'idle 30 ms', 'hbm-bound 9 ms'):
    ts = []
    for r in range(6):
        torch.cuda.synchronize()
        if what.startswith('idle'):
            time.sleep(0.009 if '9' in what else 0.030)
        elif what.startswith('hbm'):
            with torch.cuda.stream(step.stream):
                for _ in range(5):
                    g.mul_(1.0)          # 4 GB of traffic per call
        a = ev(); step.micro_step(*batch); b = ev(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f'pause before each micro-step: {what:16s} ' + ' '.join(f'{t:.1f}' for t in ts), flush=True)
