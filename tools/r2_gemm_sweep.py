"""Round-2 GEMM sweep (gpurun): tile x split-K variants on the benchmark's own NT / TN shapes with COLD operands
(distinct buffer sets cycling through > 512 MB, so weights come from HBM as they do in the step).
Usage: python tools/r2_gemm_sweep.py [nt|tn|all]  -> prints one table per shape (us per launch, TFLOP/s)."""
import sys
import torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd._lib import lib

dev = 'cuda:0'
what = sys.argv[1] if len(sys.argv) > 1 else 'all'


def time_sets(fns, reps=4):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3      # us


def nsets(bytes_per_set):
    return max(2, int(600e6 // bytes_per_set) + 1)


def nt(M, N, K, variants):
    n = nsets((M * K + N * K + M * N) * 2)
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(n)]
    W = [torch.randn(N, K, device=dev).bfloat16() * K ** -0.5 for _ in range(n)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    ref = None
    print(f'NT {M}x{N}x{K}  ({n} buffer sets)')
    for tile, split in variants:
        lib().call('az_gemm_set_tile_ex', *tile)
        try:
            fns = [(lambda i=i: ops.gemm(A[i], W[i], C[i], trans_b=True, split_k=split)) for i in range(n)]
            us = time_sets(fns)
            out = C[0].float()
            if ref is None:
                ref = out.clone()
            err = ((out - ref).norm() / ref.norm()).item()
            print(f'   tile {str(tile):16s} split {split}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s   rel-vs-first {err:.1e}')
        except Exception as e:
            print(f'   tile {tile} split {split}: FAILED {e}')
    lib().call('az_gemm_set_tile', 0, 0)


def tn(M, N, K, variants):
    """dW[M,N] = dY[K,M]^T X[K,N]"""
    n = nsets((K * M + K * N + M * N) * 2)
    DY = [torch.randn(K, M, device=dev).bfloat16() for _ in range(n)]
    X = [torch.randn(K, N, device=dev).bfloat16() for _ in range(n)]
    C = [torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    BG = [torch.zeros(M, device=dev, dtype=torch.bfloat16) for _ in range(n)]
    ref = None
    print(f'TN {M}x{N}x{K}  ({n} buffer sets)')
    for tile, split in variants:
        lib().call('az_gemm_set_tile_ex', *tile)
        try:
            fns = [(lambda i=i: ops.gemm(DY[i], X[i], C[i], trans_a=True, trans_b=False, accumulate=False, split_k=split, bias_grad=BG[i])) for i in range(n)]
            us = time_sets(fns)
            out = C[0].float()
            if ref is None:
                ref = out.clone()
            err = ((out - ref).norm() / ref.norm()).item()
            print(f'   tile {str(tile):16s} split {split}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s   rel-vs-first {err:.1e}')
        except Exception as e:
            print(f'   tile {tile} split {split}: FAILED {e}')
    lib().call('az_gemm_set_tile', 0, 0)


if what == 'ring':      # deep rings of 32-deep k-tiles vs the 64-deep 2- / 3-stage forms
    v160 = [((128, 160, 8), 1), ((128, 160, 24), 1), ((128, 160, 40), 1), ((128, 160, 56), 1)]
    v256 = [((256, 256, 0), 1), ((256, 256, 32), 1)]
    for K in (10240, 5120, 3840, 1280):
        nt(4096, 1280, K, v160 + [((256, 256, 0), 3), ((256, 256, 32), 3)])
    nt(4096, 10240, 1280, v256 + [((128, 160, 8), 1)])
    nt(4096, 5120, 1280, v256)
    nt(4096, 3840, 1280, v256)
    nt(16384, 5120, 640, v256)
    nt(16384, 640, 2560, v160 + v256)
    nt(16384, 640, 640, v160)
if what in ('nt', 'all'):
    base = [((0, 0, 0), 1), ((128, 160, 8), 1), ((128, 160, 24), 1), ((128, 128, 8), 1)]
    big = [((256, 256, 0), s) for s in (1, 2, 3, 4)] + [((256, 128, 0), s) for s in (1, 2)] + [((128, 256, 0), s) for s in (1, 2)]
    for K in (10240, 5120, 3840, 1280):
        nt(4096, 1280, K, base + big)
    nt(16384, 640, 640, base + [((256, 256, 0), 1), ((256, 256, 0), 2), ((256, 128, 0), 1)])
    nt(16384, 640, 2560, base + [((256, 256, 0), 1), ((256, 256, 0), 2), ((256, 128, 0), 1)])
    nt(4096, 10240, 1280, [((0, 0, 0), 1), ((256, 256, 0), 1), ((128, 160, 8), 1)])
if what in ('tn', 'all'):
    tiles = [(0, 0, 0), (128, 128, 8), (256, 256, 0), (256, 128, 0), (128, 256, 0)]
    for (M, N, K, splits) in [(1280, 1280, 4096, (0, 1, 2, 3, 5, 8)), (10240, 1280, 4096, (0, 1, 2)), (1280, 5120, 4096, (0, 1, 2)),
                              (3840, 1280, 4096, (0, 1, 2)), (640, 640, 16384, (0, 4, 8, 16)), (5120, 640, 16384, (0, 1, 2, 4))]:
        tn(M, N, K, [(t, s) for t in tiles for s in splits])
