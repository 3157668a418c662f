"""Where does a GEMM k-iteration spend its time?  The same launch with (a) everything, (b) operand DMA only (no fragment reads,
no MFMAs), (c) fragment reads + MFMAs only (no DMA after the first k-tile) -- option GEMM_ABLATE, timing only.
If full ~ dma + compute the phases do not overlap; if full ~ max(dma, compute) the kernel sits on that component."""
import sys
import torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd._lib import lib, set_option

dev = 'cuda:0'


def time_sets(fns, reps=4):
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


def case(kind, M, N, K, tile, split=1):
    n = max(2, int(600e6 // ((M * K + N * K + M * N) * 2)) + 1)
    if kind == 'nt':
        A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(n)]; W = [torch.randn(N, K, device=dev).bfloat16() for _ in range(n)]
        C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(n)]
        fns = [(lambda i=i: ops.gemm(A[i], W[i], C[i], trans_b=True, split_k=split)) for i in range(n)]
    else:
        DY = [torch.randn(K, M, device=dev).bfloat16() for _ in range(n)]; X = [torch.randn(K, N, device=dev).bfloat16() for _ in range(n)]
        C = [torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(n)]
        fns = [(lambda i=i: ops.gemm(DY[i], X[i], C[i], trans_a=True, trans_b=False, accumulate=False, split_k=split)) for i in range(n)]
    lib().call('az_gemm_set_tile_ex', *tile)
    out = []
    for ab in (0, 1, 2, 3):
        set_option('GEMM_ABLATE', ab)
        out.append(time_sets(fns))
    set_option('GEMM_ABLATE', 0)
    lib().call('az_gemm_set_tile', 0, 0)
    print(f'{kind} {M}x{N}x{K} tile {tile} split {split}: full {out[0]:7.1f} us | DMA only {out[1]:7.1f} | reads+MFMA only {out[2]:7.1f} | neither (launch, prologue, epilogue) {out[3]:7.1f}', flush=True)


import os
if os.environ.get('AZ_ABLATE_RING'):
    for tile in [(128, 160, 8), (128, 160, 24), (128, 160, 40), (128, 160, 56)]:
        case('nt', 4096, 1280, 10240, tile)
    for tile in [(256, 256, 0), (256, 256, 32)]:
        case('nt', 4096, 10240, 1280, tile)
    sys.exit(0)
for tile in [(128, 160, 8), (128, 160, 24), (128, 160, 4), (128, 128, 8), (128, 128, 0)]:
    case('nt', 4096, 1280, 10240, tile)
case('nt', 4096, 1280, 10240, (256, 256, 0), 3)
case('nt', 4096, 1280, 10240, (256, 256, 0), 1)
for tile in [(128, 160, 8), (128, 160, 24)]:
    case('nt', 4096, 1280, 1280, tile)
case('nt', 4096, 10240, 1280, (256, 256, 0))
case('nt', 4096, 10240, 1280, (128, 160, 8))
for tile in [(128, 128, 8), (256, 256, 0)]:
    case('tn', 10240, 1280, 4096, tile, 1)
case('tn', 1280, 1280, 4096, (128, 128, 8), 5)
