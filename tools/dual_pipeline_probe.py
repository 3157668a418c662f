"""Upper bound for ANY scheme that overlaps work of different micro-steps (forward of m + 1 under the backward of m, deferred
branches, ...): two INDEPENDENT full training pipelines -- two UNets, each with its own data-gradient / weight-gradient streams and
activation pools -- issued alternately from one host thread, against one pipeline alone.  If two pipelines side by side do not
finish two micro-steps in clearly less than twice the time of one, the chip has no idle capacity such a schedule could use.
(The two pipelines share the per-device scratch workspaces: their numbers are garbage, their timing is not.)
usage: python tools/dual_pipeline_probe.py [EXCL]     EXCL=0: forwards on the 2-stage tiles (LDS_EXCLUSIVE off)"""
import statistics, sys, time
import torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep

dev = torch.device('cuda', 0)
nets, steps, outer = [], [], []
for i in range(2):
    u = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(u)
    nets.append(u); steps.append(TrainStep(u, mode='epsilon', grad_accum=8, use_graph=False)); outer.append(torch.cuda.Stream(dev))
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
for s, o in zip(steps, outer):
    with torch.cuda.stream(o):
        for _ in range(3):
            s.micro_step(*batch)
    torch.cuda.synchronize()


def run(which, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for i in which:
            with torch.cuda.stream(outer[i]):
                steps[i].micro_step(*batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * len(which)) * 1e3


for r in range(3):
    a = run([0], 6); b = run([1], 6); d = run([0, 1], 6)
    print(f'round {r}: pipeline 0 alone {a:.2f} ms per micro-step, pipeline 1 alone {b:.2f}, both side by side {d:.2f} ms per micro-step '
          f'(x{(a + b) / 2 / d:.3f} throughput)', flush=True)
# offset by half a micro-step: pipeline 1 starts when pipeline 0 is in its backward
torch.cuda.synchronize()
with torch.cuda.stream(outer[0]):
    steps[0].micro_step(*batch)
time.sleep(0.045)
t0 = time.perf_counter()
for _ in range(6):
    for i in (1, 0):
        with torch.cuda.stream(outer[i]):
            steps[i].micro_step(*batch)
torch.cuda.synchronize()
print(f'offset start: {(time.perf_counter() - t0) / 12 * 1e3:.2f} ms per micro-step', flush=True)
