import sys, time, torch, json
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device('cuda', 0)
t0 = time.time(); unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet); torch.cuda.synchronize()
print('init', round(time.time() - t0, 1), 's  flat numel', unet.flat_numel, flush=True)
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
step = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=(len(sys.argv) > 1 and sys.argv[1] == 'graph'))
for i in range(4):
    t0 = time.time(); l = step.micro_step(*batch); step.synchronize(); dt = time.time() - t0
    print(f'micro-step {i}: {dt*1e3:.1f} ms  loss {l.item():.5f}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB', flush=True)
ops.PROFILER = ops.Profiler()
import os
ops.PROFILE_SHAPES = bool(os.environ.get('AZ_SHAPES'))
ps = TrainStep(unet, mode='epsilon', grad_accum=8, use_graph=False)
ps.micro_step(*batch); ps.synchronize(); s = ops.PROFILER.summary(); ops.PROFILER = None
tot = sum(v['ms'] for v in s.values())
print('eager profiled micro-step: sum of kernel ms =', round(tot, 1))
for k, v in sorted(s.items(), key=lambda kv: -kv[1]['ms'])[:70]:
    tf = v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['flops'] else 0
    gb = v['bytes'] / (v['ms'] * 1e-3) / 1e9 if v['bytes'] else 0
    print(f"  {k:34s} calls {v['calls']:5d}  {v['ms']:8.2f} ms  {100*v['ms']/tot:5.1f}%  {tf:7.1f} TFLOP/s  {gb:7.0f} GB/s")
json.dump(s, open('gpurun_out/fullsize_breakdown.json', 'w'), indent=1)
