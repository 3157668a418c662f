"""NT GEMM tile sweep incl. the 128x160 tile, warm (one buffer set) and cold (many distinct buffer sets, > Infinity Cache)."""
import sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd._lib import lib
dev = 'cuda:0'
TILES = [(128, 128, 8), (128, 160, 8), (128, 160, 24), (256, 256, 0)]
def run(M, N, K, nset, tile, reps=4, check=None):
    lib().call('az_gemm_set_tile_ex', *tile)
    g = torch.Generator(device=dev).manual_seed(1)
    As = [torch.randn(M, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
    Ws = [torch.randn(N, K, device=dev, generator=g).bfloat16() for _ in range(nset)]
    Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    for i in range(nset): ops.gemm(As[i], Ws[i], Cs[i], trans_b=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(nset): ops.gemm(As[i], Ws[i], Cs[i], trans_b=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * nset)
    return 2 * M * N * K / ms / 1e9, Cs[0].clone()
print('M N K sets | ' + ' | '.join(f'{a}x{b}/{c}' for a, b, c in TILES) + '  (TFLOP/s, * = differs from 128x128)')
for (M, N, K) in [(4096, 1280, 1280), (4096, 1280, 3840), (4096, 1280, 5120), (4096, 1280, 10240), (4096, 3840, 1280), (4096, 5120, 1280),
                  (4096, 10240, 1280), (16384, 640, 640), (16384, 640, 2560), (16384, 1920, 640), (16384, 2560, 640), (65536, 320, 320), (65536, 320, 1280)]:
    for nset in (1, 24):
        if nset * (M * K + N * K + M * N) * 2 > 30e9: nset = max(2, int(30e9 / ((M * K + N * K + M * N) * 2)))
        out, ref = [], None
        for tile in TILES:
            tf, c = run(M, N, K, nset, tile)
            if ref is None: ref = c
            out.append('%7.0f%s' % (tf, '' if torch.equal(c, ref) else '*'))
        print(M, N, K, nset, '|', ' | '.join(out), flush=True)
lib().call('az_gemm_set_tile', 0, 0)
