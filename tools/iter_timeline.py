"""GPU timeline of one bench iteration (N=1): duration of each of the 8 micro-steps and of the optimizer boundary, from
events on the data-gradient stream.  OVERLAP=0: the update of all regions and the W^T refresh on the main stream (round 3)."""
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
ga = 8
step = TrainStep(unet, mode='epsilon', grad_accum=ga, use_graph=False)
import os
opt = ShardedRaven(unet, lr=8e-7, clip_grad_norm=1.0, overlap=os.environ.get('OVERLAP', '1') == '1')
for _ in range(3): step.micro_step(*batch)
step.synchronize(); opt.zero_grad()
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(step.stream); return e
def iteration(marks):
    for m in range(ga):
        if m == ga - 2: opt.prefetch()
        marks.append(ev()); step.micro_step(*batch)
    marks.append(ev())
    with torch.cuda.stream(step.stream):
        pass
    opt.step(); marks.append(torch.cuda.Event(enable_timing=True)); marks[-1].record(torch.cuda.current_stream())
    opt.zero_grad()
iteration([]); torch.cuda.synchronize()
for it in range(int(os.environ.get('ROUNDS', '2'))):
    marks = []; t0 = time.time(); iteration(marks); marks2 = []; iteration(marks2); torch.cuda.synchronize(); t1 = time.time()
    d = [marks[i].elapsed_time(marks[i + 1]) for i in range(ga)]
    print('micro-steps (ms): ' + ' '.join(f'{x:.1f}' for x in d), flush=True)
    print(f'  last micro-step end -> optimizer kernels done on the default stream: {marks[ga].elapsed_time(marks[ga + 1]):.1f} ms;'
          f' -> first micro-step of the next iteration starts: {marks[ga].elapsed_time(marks2[0]):.1f} ms; 2 iterations wall {1e3 * (t1 - t0):.0f} ms', flush=True)
