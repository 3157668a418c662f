import sys, torch
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
dev='cuda:0'
B, heads, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
C = heads * 64
qkv = torch.randn(B, T, 3 * C, device=dev).bfloat16()
q, k, v = qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:]
o = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16); do = torch.randn(B, T, C, device=dev).bfloat16()
lse = torch.empty(B * heads * T, device=dev); delta = torch.empty(B * heads * T, device=dev)
dqkv = torch.empty_like(qkv)
for _ in range(5):
    ops.attn_fwd(q, k, v, o, lse, heads, 0.125)
    ops.attn_bwd(q, k, v, o, do, lse, delta, dqkv[..., :C], dqkv[..., C:2*C], dqkv[..., 2*C:], heads, 0.125)
torch.cuda.synchronize()
