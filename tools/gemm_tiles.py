import sys, torch, ctypes
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd._lib import lib
dev = 'cuda:0'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(4096, 4096, 4096), (4096, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120),
          (16384, 640, 640), (16384, 1920, 640), (16384, 5120, 640), (16384, 640, 2560), (65536, 320, 320), (65536, 320, 1280)]
tiles = [(128, 128, 0), (128, 128, 8), (128, 160, 8), (256, 256, 0)]
print('mode      M      N      K  ' + '  '.join(f'{a}x{b}/{c}'.rjust(9) for a, b, c in tiles) + '   (TFLOP/s; * = mismatch vs 128x128)')
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    for mode in ('nt', 'nn', 'tn'):
        res, ref = [], None
        for tl in tiles:
            lib().call('az_gemm_set_tile_ex', *tl)
            if mode == 'nt':
                c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); fn = lambda: ops.gemm(a, w, c, trans_b=True); fl = (M, N, K)
            elif mode == 'nn':
                c = torch.empty(M, K, device=dev, dtype=torch.bfloat16); fn = lambda: ops.gemm(dy, w, c, trans_b=False); fl = (M, K, N)
            else:
                c = torch.zeros(N, K, device=dev, dtype=torch.bfloat16); fn = lambda: ops.gemm(dy, a, c, trans_a=True, trans_b=False, accumulate=False, split_k=0); fl = (N, K, M)
            fn(); torch.cuda.synchronize()
            out = c.clone()
            if ref is None: ref = out
            bad = '' if torch.equal(out, ref) else '*'
            ms = timeit(fn)
            res.append(f'{2*M*N*K/ms/1e9:8.1f}{bad}')
        print(f'{mode}  {fl[0]:6d} {fl[1]:6d} {fl[2]:6d}  ' + '  '.join(r.rjust(9) for r in res))
lib().call('az_gemm_set_tile', 0, 0)
