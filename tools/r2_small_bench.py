"""Small HBM-bound kernels at the model's shapes, cold buffers (distinct sets > 512 MB): LayerNorm fwd / bwd, W^T transposes."""
import sys
import torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops

dev = 'cuda:0'


def time_sets(fns, reps=5):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns:
            f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3


for M, C in [(4096, 1280), (16384, 640)]:
    n = 24
    X = [torch.randn(M, C, device=dev).bfloat16() for _ in range(n)]
    DY = [torch.randn(M, C, device=dev).bfloat16() for _ in range(n)]
    Y = [torch.empty(M, C, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    DX = [torch.zeros(M, C, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    g = torch.ones(C, device=dev, dtype=torch.bfloat16); b = torch.zeros(C, device=dev, dtype=torch.bfloat16)
    dg = torch.zeros(C, device=dev, dtype=torch.bfloat16); db = torch.zeros(C, device=dev, dtype=torch.bfloat16)
    st = [torch.empty(2 * M, device=dev) for _ in range(n)]
    us = time_sets([(lambda i=i: ops.layernorm_fwd(X[i], g, b, Y[i], st[i])) for i in range(n)])
    print(f'ln_fwd {M}x{C}: {us:.1f} us  {4.0 * M * C / us / 1e6:.2f} TB/s')
    for acc in (False, True):
        us = time_sets([(lambda i=i: ops.layernorm_bwd(X[i], g, st[i], DY[i], DX[i], dg, db, accumulate_dx=acc)) for i in range(n)])
        print(f'ln_bwd {M}x{C} accumulate={acc}: {us:.1f} us  {(8.0 if acc else 6.0) * M * C / us / 1e6:.2f} TB/s')
for R, C in [(1280, 1280), (10240, 1280), (1280, 5120), (3840, 1280), (640, 640)]:
    n = max(2, int(600e6 // (R * C * 4)))
    S = [torch.randn(R, C, device=dev).bfloat16() for _ in range(n)]
    D = [torch.empty(C, R, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    us = time_sets([(lambda i=i: ops.transpose(S[i], D[i])) for i in range(n)])
    assert torch.equal(D[0], S[0].t())
    print(f'transpose {R}x{C}: {us:.1f} us  {4.0 * R * C / us / 1e6:.2f} TB/s')
w = torch.randn(1280, 9, 1280, device=dev).bfloat16(); wt = torch.empty(1280, 9, 1280, device=dev, dtype=torch.bfloat16)
us = time_sets([lambda: ops.transpose_batched(w.permute(1, 0, 2), wt.permute(1, 0, 2))])
assert torch.equal(wt, w.permute(2, 1, 0).contiguous())
print(f'transpose conv 1280x9x1280: {us:.1f} us  {4.0 * 1280 * 9 * 1280 / us / 1e6:.2f} TB/s')
