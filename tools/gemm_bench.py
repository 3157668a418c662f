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
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120),
          (16384, 640, 640), (16384, 1920, 640), (16384, 5120, 640), (16384, 640, 2560), (308, 2560, 2048)]
print('mode   M      N      K      ms     TFLOP/s')
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(a, w, c, trans_b=True))
    print(f'nt  {M:6d} {N:6d} {K:6d} {ms:8.4f} {2*M*N*K/ms/1e9:8.1f}')
    # dgrad: dX[M,K] = dY[M,N] @ W[N,K]
    dy = torch.randn(M, N, device=dev).bfloat16(); dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(dy, w, dx, trans_b=False))
    print(f'nn  {M:6d} {K:6d} {N:6d} {ms:8.4f} {2*M*N*K/ms/1e9:8.1f}')
    # wgrad: dW[N,K] = dY[M,N]^T @ X[M,K]
    dw = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(dy, a, dw, trans_a=True, trans_b=False, accumulate=True, split_k=0))
    print(f'tn  {N:6d} {K:6d} {M:6d} {ms:8.4f} {2*M*N*K/ms/1e9:8.1f}   (auto split-k)')
    ms = timeit(lambda: ops.gemm(dy, a, dw, trans_a=True, trans_b=False, accumulate=True, split_k=1))
    print(f'tn  {N:6d} {K:6d} {M:6d} {ms:8.4f} {2*M*N*K/ms/1e9:8.1f}   (no split)')
