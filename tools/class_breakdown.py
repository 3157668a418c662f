import sys, time, torch, json
sys.path.insert(0, ".")
import bench
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.train_step import TrainStep
dev = torch.device("cuda", 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
unet.concurrent_wgrad = False
batch = bench.synthetic_batch(0, 0, 0, 4, dev)
ps = TrainStep(unet, mode="epsilon", grad_accum=8, use_graph=False)
ps.micro_step(*batch); ps.synchronize()
import os
ops.PROFILE_SHAPES = os.environ.get("AZ_SHAPES", "0") == "1"
ops.PROFILER = ops.Profiler()
ps.micro_step(*batch); ps.synchronize(); s = ops.PROFILER.summary(); ops.PROFILER = None
tot = sum(v["ms"] for v in s.values())
print("serial profiled micro-step: sum of kernel ms =", round(tot, 1))
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get('AZ_TOP', '34'))]:
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0
    gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["bytes"] else 0
    print("  %-36s calls %5d  %8.2f ms  %5.1f%%  %7.1f TFLOP/s  %7.0f GB/s" % (k, v["calls"], v["ms"], 100 * v["ms"] / tot, tf, gb))
