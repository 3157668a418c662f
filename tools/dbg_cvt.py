import torch, sys
sys.path.insert(0, '.')
from aozora_sdxl_training_amd import ops
DEV='cuda:0'
# exact ties between adjacent bf16 values, both parities
base = torch.tensor([0x3B81, 0x3B82, 0x3F80, 0x3F81, 0xBB81, 0x4000], dtype=torch.int32)
f_lo = (base << 16).view(torch.float32); f_hi = ((base + 1) << 16).view(torch.float32)
ties = (f_lo + f_hi) / 2
x = torch.cat([ties, torch.tensor([-0.0039520263671875])])
d = torch.empty(x.numel(), dtype=torch.bfloat16, device=DEV)
ops.f32_to_bf16(x.to(DEV).contiguous(), d)
print('in   ', x.tolist())
print('mine ', [hex(v & 0xFFFF) for v in d.cpu().view(torch.int16).tolist()])
print('torch', [hex(v & 0xFFFF) for v in x.bfloat16().view(torch.int16).tolist()])
print('tgpu ', [hex(v & 0xFFFF) for v in x.to(DEV).bfloat16().cpu().view(torch.int16).tolist()])
