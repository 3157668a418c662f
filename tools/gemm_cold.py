import sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
from aozora_sdxl_training_amd._lib import lib
dev = 'cuda:0'
def run(M, N, K, nset, bm, bn, reps=3):
    lib().call('az_gemm_set_tile', bm, bn)
    As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nset)]
    Ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(nset)]
    Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    for i in range(nset): ops.gemm(As[i], Ws[i], Cs[i], trans_b=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(nset): ops.gemm(As[i], Ws[i], Cs[i], trans_b=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * nset)
    return 2 * M * N * K / ms / 1e9
for (M, N, K) in [(4096, 1280, 1280), (4096, 10240, 1280), (4096, 1280, 5120)]:
    for nset in (1, 8, 64):
        if nset * (M * K + N * K + M * N) * 2 > 40e9: continue
        print(M, N, K, 'distinct buffer sets', nset, ' 128x128: %.0f TF/s   256x256: %.0f TF/s' % (run(M, N, K, nset, 128, 128), run(M, N, K, nset, 256, 256)))
lib().call('az_gemm_set_tile', 0, 0)
