import sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
dev = 'cuda:0'
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N) in [(4096, 1280), (4096, 10240), (16384, 640)]:
    for K in (64, 1280):
        a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); r = torch.randn(M, N, device=dev).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        t0 = timeit(lambda: ops.gemm(a, w, c, trans_b=True))
        t1 = timeit(lambda: ops.gemm(a, w, c, trans_b=True, bias=b, residual=r))
        t2 = timeit(lambda: ops.gemm(a, w, c, trans_b=True, accumulate=True))
        cp = timeit(lambda: ops.add_rows(r, None, c))
        print(f'M={M} N={N} K={K}: plain {t0:.1f} us | +bias+residual {t1:.1f} us | accumulate {t2:.1f} us | (vector copy of C: {cp:.1f} us; C = {M*N*2/1e6:.1f} MB)')
