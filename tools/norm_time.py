"""HIP-event time of the normalisation entry points at the step's shapes (cold-ish: rotating buffer sets), for A/Bs of two builds
(AZ_LIB=<other .so>).  usage: python tools/norm_time.py"""
import os, sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import _lib as _L
if os.environ.get('AZ_LIB'):
    _L.LIB_PATH = os.path.abspath(os.environ['AZ_LIB'])
from aozora_sdxl_training_amd import ops
dev = 'cuda:0'
def timeit(fn, nset, reps=20):
    for i in range(nset): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        for i in range(nset): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * nset) * 1e3
out = []
for (B, HW, C, silu) in [(4, 16384, 320, True), (4, 16384, 640, True), (4, 4096, 640, True), (4, 1024, 1280, True), (4, 1024, 2560, True), (4, 4096, 640, False), (4, 1024, 1280, False)]:
    ns = 6
    xs = [torch.randn(B, HW, C, device=dev).bfloat16() for _ in range(ns)]
    ys = [torch.empty_like(x) for x in xs]; dys = [torch.randn(B, HW, C, device=dev).bfloat16() for _ in range(ns)]; dxs = [torch.empty_like(x) for x in xs]
    g = torch.ones(C, device=dev, dtype=torch.bfloat16); b_ = torch.zeros(C, device=dev, dtype=torch.bfloat16)
    st = torch.empty(B * 32 * 2, device=dev); dg = torch.zeros(C, device=dev, dtype=torch.bfloat16); db = torch.zeros(C, device=dev, dtype=torch.bfloat16)
    f = timeit(lambda i: ops.groupnorm_fwd(xs[i], g, b_, ys[i], st, 32, 1e-5, silu), ns)
    w = timeit(lambda i: ops.groupnorm_bwd(xs[i], g, b_, st, dys[i], dxs[i], dg, db, 32, silu), ns)
    out.append(f'gn {B}x{HW}x{C} silu={int(silu)}: fwd {f:6.1f} us  bwd {w:6.1f} us')
for (M, C) in [(4096, 1280), (16384, 640)]:
    ns = 8
    xs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(ns)]; ys = [torch.empty_like(x) for x in xs]
    dys = [torch.randn(M, C, device=dev).bfloat16() for _ in range(ns)]; dxs = [torch.empty_like(x) for x in xs]
    g = torch.ones(C, device=dev, dtype=torch.bfloat16); b_ = torch.zeros(C, device=dev, dtype=torch.bfloat16)
    st = torch.empty(2 * M, device=dev); dg = torch.zeros(C, device=dev, dtype=torch.bfloat16); db = torch.zeros(C, device=dev, dtype=torch.bfloat16)
    f = timeit(lambda i: ops.layernorm_fwd(xs[i], g, b_, ys[i], st), ns)
    w = timeit(lambda i: ops.layernorm_bwd(xs[i], g, st, dys[i], dxs[i], dg, db), ns)
    out.append(f'ln {M}x{C}: fwd {f:6.1f} us  bwd {w:6.1f} us')
print('\n'.join(out))
