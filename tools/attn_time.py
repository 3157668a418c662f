"""Attention forward / backward time at the step's shapes (cold-ish: rotating buffer sets), one process."""
import sys, torch
sys.path.insert(0, '.')
import os
from aozora_sdxl_training_amd import _lib
if os.environ.get('AZ_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['AZ_LIB'])      # A/B of two builds of the library (one process each, same box)
from aozora_sdxl_training_amd import ops
dev = 'cuda:0'
def bench(B, heads, Tq, Tk, nset=6, reps=5):
    C = heads * 64
    sets = []
    for _ in range(nset):
        if Tq == Tk:
            qkv = torch.randn(B, Tq, 3 * C, device=dev).bfloat16()
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            dqkv = torch.empty_like(qkv); dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        else:
            q = torch.randn(B, Tq, C, device=dev).bfloat16(); kv = torch.randn(B, Tk, 2 * C, device=dev).bfloat16()
            k, v = kv[..., :C], kv[..., C:]
            dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk, dv = dkv[..., :C], dkv[..., C:]
        o = torch.empty(B, Tq, C, device=dev, dtype=torch.bfloat16); do = torch.randn(B, Tq, C, device=dev).bfloat16()
        lse = torch.empty(B * heads * Tq, device=dev); delta = torch.empty(B * heads * Tq, device=dev)
        sets.append((q, k, v, o, do, lse, delta, dq, dk, dv))
    def run(fn):
        for s in sets: fn(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for s in sets: fn(s)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * nset) * 1e3
    fl = 4.0 * B * heads * Tq * Tk * 64
    tf = run(lambda s: ops.attn_fwd(s[0], s[1], s[2], s[3], s[5], heads, 0.125))
    tb = run(lambda s: ops.attn_bwd(*s[:3], s[3], s[4], s[5], s[6], s[7], s[8], s[9], heads, 0.125))
    print(f'attn {B}x{heads} {Tq}x{Tk}: fwd {tf:7.1f} us {fl / tf / 1e6:6.0f} TF/s   bwd {tb:7.1f} us {2.5 * fl / tb / 1e6:6.0f} TF/s', flush=True)
for shape in ((4, 20, 1024, 1024), (4, 10, 4096, 4096), (4, 20, 1024, 77), (4, 10, 4096, 77)):
    bench(*shape)
