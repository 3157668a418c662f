import sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
dev='cuda:0'
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, N, device=dev).bfloat16(); dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16); dw = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    if mode == 'nt': ops.gemm(a, w, c, trans_b=True)
    elif mode == 'nn': ops.gemm(dy, w, dx, trans_b=False)
    else: ops.gemm(dy, a, dw, trans_a=True, trans_b=False, accumulate=True, split_k=1)
torch.cuda.synchronize()
