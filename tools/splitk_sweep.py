import sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
dev = 'cuda:0'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (Nout, Kin, M) in [(1280, 1280, 4096), (3840, 1280, 4096), (1280, 5120, 4096), (10240, 1280, 4096), (640, 640, 16384), (1920, 640, 16384), (640, 2560, 16384), (2560, 2048, 308)]:
    dy = torch.randn(M, Nout, device=dev).bfloat16(); x = torch.randn(M, Kin, device=dev).bfloat16()
    dw = torch.zeros(Nout, Kin, device=dev, dtype=torch.bfloat16)
    res = []
    for S in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        ms = timeit(lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=False, accumulate=True, split_k=S))
        res.append(f'S={S}:{2*M*Nout*Kin/ms/1e9:6.0f}')
    print(f'tn {Nout}x{Kin}x{M}  ' + '  '.join(res))
