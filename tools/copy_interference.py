"""Does a large pinned-host <-> device copy on a side stream slow the compute stream down?  (The Raven step streams 20 GB of
m / v per optimizer step beside the micro-steps; rocprofv3 shows those copies as `__amd_rocclr_copyBuffer` shader kernels.)"""
import os, sys, time, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
dev = 'cuda:0'
G = 1 << 30
host = torch.empty(4 * G, dtype=torch.uint8).pin_memory()
devb = torch.empty(4 * G, dtype=torch.uint8, device=dev)
a = torch.randn(4096, 5120, device=dev).bfloat16(); w = torch.randn(5120, 5120, device=dev).bfloat16()
c = torch.empty(4096, 5120, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
def gemms(n=400):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.gemm(a, w, c, trans_b=True)
    e1.record(); return e0, e1
gemms(50); torch.cuda.synchronize()
e0, e1 = gemms(); torch.cuda.synchronize(); base = e0.elapsed_time(e1)
print(f"env HSA_ENABLE_SDMA={os.environ.get('HSA_ENABLE_SDMA')}  GEMMs alone: {base:.1f} ms", flush=True)
for name, src, dst in (("H2D", host, devb), ("D2H", devb, host)):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(side); dst.copy_(src, non_blocking=True); c1.record(side)
    e0, e1 = gemms(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1); ct = c0.elapsed_time(c1)
    print(f"  beside a 4 GiB {name} copy ({ct:.0f} ms, {4 * G / ct / 1e6:.1f} GB/s): GEMMs {t:.1f} ms  (+{100 * (t - base) / base:.1f} % over {min(ct, t):.0f} ms of overlap)", flush=True)
    torch.cuda.synchronize()
    c0.record(); dst.copy_(src, non_blocking=True); c1.record(); torch.cuda.synchronize()
    print(f"  {name} alone: {4 * G / c0.elapsed_time(c1) / 1e6:.1f} GB/s", flush=True)
