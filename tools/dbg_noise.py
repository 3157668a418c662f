import torch, sys
sys.path.insert(0, '.')
from oracle import step_ref as R
from aozora_sdxl_training_amd import schedule as S, ops
DEV='cuda:0'
B,C,H,W=3,4,8,6
g = torch.Generator().manual_seed(5)
lat = torch.randn(B,C,H,W,generator=g).bfloat16(); noise=torch.randn(B,C,H,W,generator=g); ts=torch.tensor([10,500,999]); jit=torch.rand(B,generator=g)
noisy_ref,tgt_ref,_ = R.make_noisy_and_target("epsilon", lat, noise, ts, R.ddpm_alphas_cumprod(), jit)
ta,tb = S.ddpm_coef_tables(torch.bfloat16)
ca,cb = ta[ts], tb[ts]
noisy = torch.empty(B,H,W,8,dtype=torch.bfloat16,device=DEV); tgt=torch.empty(B,C,H,W,dtype=torch.float32,device=DEV)
ops.noise_target(0, lat.to(DEV), noise.to(DEV), ca.float().contiguous().to(DEV), cb.float().contiguous().to(DEV), noisy, tgt)
got = noisy[...,:4].cpu().permute(0,3,1,2).float(); want = noisy_ref.bfloat16().float()
bad = (got!=want)
print('mismatch', bad.sum().item(), 'of', bad.numel())
idx = bad.nonzero()[:8]
for i in idx:
    i=tuple(i.tolist()); print(i, got[i].item(), want[i].item(), noisy_ref[i].item(), 'lat',lat[i].float().item(),'noise',noise[i].item(), 'a', ca[i[0]].item(), 's', cb[i[0]].item())
print('ca', ca, 'cb', cb)
