"""Seams of the reference loop as the reference calls them, on the device, against vectors captured from the
reference's own functions (tests/golden/make_golden.py, make_golden_r2.py):

  * weighted_sdxl_mse_loss(pred, target, timesteps, weights)       train.py:2408-2416, 2763
  * torch.nn.utils.clip_grad_norm_(list_of_params, max_norm)       train.py:2775
  * RavenAdamW(momentum_dtype=torch.float16)                       raven.py:37-42
"""
import json
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold_r2():
    with open(os.path.join(ROOT, "tests", "golden", "golden_r2.json")) as f:
        host = json.load(f)
    return host, torch.load(os.path.join(ROOT, "tests", "golden", "golden_r2.pt"), map_location="cpu", weights_only=True)


def test_weighted_sdxl_mse_loss_callable_matches_reference(golden_host, golden_tensors):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aozora_sdxl_training_amd.loss import weighted_sdxl_mse_loss
    seen = 0
    for c in golden_host["loss"]:
        k = c["key"]
        pred0 = golden_tensors[k + "_pred"]
        if pred0.dtype != torch.bfloat16:          # the HIP path computes on the bf16 prediction (train.py:273: bf16 / fp16 only)
            continue
        seen += 1
        curve = None if c["curve"] == "none" else golden_tensors["curve_" + c["curve"]]
        pred = pred0.clone().to(DEV).requires_grad_(True)
        loss = weighted_sdxl_mse_loss(pred, golden_tensors[k + "_tgt"].to(DEV), golden_tensors[k + "_ts"].to(DEV),
                                      curve.to(DEV) if curve is not None else None)
        want = float(golden_tensors[k + "_loss"])
        assert loss.dtype == torch.float32 and loss.dim() == 0
        assert abs(loss.item() - want) <= 2e-6 * abs(want), (k, loss.item(), want)
        (loss / 4).backward()                       # an upstream factor, as (loss / GA).backward() applies
        got = pred.grad.float().cpu()
        ref = golden_tensors[k + "_dpred"].float() / 4          # exact: a power of two
        tol = ref.abs() * 2.0 ** -7 + 1e-12                     # one bf16 ulp (the reference's gradient is bf16 too)
        assert bool(((got - ref).abs() <= tol).all()), (k, (got - ref).abs().max().item())
    assert seen >= 4
    with pytest.raises(Exception):
        weighted_sdxl_mse_loss(torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 4, 4), torch.tensor([0]))   # CPU tensor: no fallback


def test_clip_grad_norm_list_of_parameters(golden_tensors):
    """train.py:2775 passes a parameter LIST.  (i) loose device tensors against the reference's own clip fixture;
    (ii) AozoraUNet parameters: the list form over the trainable subset equals the whole-model form."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aozora_sdxl_training_amd.clip import clip_grad_norm_
    for ci, dt in enumerate([torch.bfloat16, torch.float32]):
        ps = [torch.nn.Parameter(torch.zeros(n, dtype=dt, device=DEV)) for n in (5, 64, 1000)]
        for i, p in enumerate(ps):
            p.grad = golden_tensors[f"clip{ci}_g{i}"].clone().to(DEV)
        n = clip_grad_norm_(ps + [torch.nn.Parameter(torch.zeros(3, device=DEV))], 1.0)       # a parameter without .grad is skipped
        want = float(golden_tensors[f"clip{ci}_norm"])
        assert abs(n.item() - want) <= (8e-3 if dt == torch.bfloat16 else 1e-5) * want      # torch rounds per-tensor norms to bf16
        for i, p in enumerate(ps):
            ref = golden_tensors[f"clip{ci}_c{i}"].float()
            err = (p.grad.float().cpu() - ref).abs()
            assert bool((err <= ref.abs() * (2.0 ** -6 if dt == torch.bfloat16 else 1e-5) + 1e-9).all()), (ci, i, err.max().item())
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import mini_config
    unet = AozoraUNet(mini_config(), DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    unet.gflat.copy_((torch.randn(unet.flat_numel, generator=g, device=DEV) * 0.01).bfloat16())
    names = [n for n, _ in unet.named_parameters()]
    for n_, p in unet.named_parameters():
        p.requires_grad = not n_.startswith("mid_block")
    unet.expose_grads()
    before = unet.gflat.clone()
    n_all = clip_grad_norm_(unet, 0.5).item()
    after_all = unet.gflat.clone()
    unet.gflat.copy_(before)
    n_list = clip_grad_norm_([p for p in unet.parameters() if p.requires_grad], 0.5).item()
    assert n_all == n_list and n_all > 0.5 and torch.equal(unet.gflat, after_all)
    assert not torch.equal(after_all, before)
    lo = min(unet._slots[n_][0] for n_ in names if n_.startswith("mid_block"))
    hi = max(unet._slots[n_][0] + math.prod(unet._slots[n_][1]) for n_ in names if n_.startswith("mid_block"))
    assert torch.equal(after_all[lo:hi], before[lo:hi])            # frozen range untouched


def test_raven_fp16_momentum_against_reference(gold_r2):
    """momentum_dtype=torch.float16 (raven.py:37-42; the round-1 build rejected it): parameters, m and v after each of three
    steps against the reference's RavenAdamW on the CPU."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aozora_sdxl_training_amd.optimizers import RavenAdamW
    host, tens = gold_r2
    for c in host["raven_f16"]:
        k = c["key"]
        p = torch.nn.Parameter(tens[k + "_init"].clone().to(DEV))
        o = RavenAdamW([{"params": [p], "lr_scale": 1.0}], lr=c["lr"], betas=tuple(c["betas"]), weight_decay=c["wd"], eps=c["eps"],
                       debias_strength=c["debias"], momentum_dtype=torch.float16)
        for s in range(c["steps"]):
            p.grad = tens[f"{k}_g{s}"].clone().to(DEV)
            o.step()
            torch.cuda.synchronize()
            want = tens[f"{k}_p{s}"].float()
            err = (p.detach().cpu().float() - want).abs()
            assert bool(((err <= want.abs() * 2.0 ** -7 + 1e-30) | (err <= 0.02 * 4e-3)).all()), (k, s, err.max().item())
            assert (err > 0).float().mean().item() <= 0.005 or err.numel() < 64, (k, s)      # the kernel mirrors raven.py:125-143 rounding for rounding
            st = o.state[p]
            assert st["exp_avg"].dtype == torch.float16 and st["exp_avg_sq"].dtype == torch.float16
            assert torch.allclose(st["exp_avg"].float(), tens[f"{k}_m{s}"].float(), rtol=2e-3, atol=2e-6), (k, s)
            assert torch.allclose(st["exp_avg_sq"].float(), tens[f"{k}_v{s}"].float(), rtol=2e-3, atol=1e-7), (k, s)
        saved = o.save_cpu_state()
        assert str(saved["_momentum_dtype"]) == c["momentum_dtype"] and str(saved[0]["exp_avg_cpu"].dtype) == c["state_dtype"]
