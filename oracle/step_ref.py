"""ORACLE (test infrastructure, NOT product code).

CPU restatement of the reference's train-step arithmetic around the UNet call:

  * DDPM constants / add_noise / get_velocity   train.py:2609-2628, 2753-2757
    (diffusers DDPMScheduler with the SDXL-base scheduler config: 1000 steps,
    "scaled_linear" betas 0.00085..0.012 -- third-party, PARITY UNPINNED)
  * rectified-flow branch                        train.py:2743-2752
  * weighted MSE loss                            train.py:2408-2416   (pinned: tests/golden F3)
  * global-norm clip                             train.py:2771-2781   (torch.nn.utils.clip_grad_norm_)
  * Raven / Titan AdamW element math             raven.py:96-147, titan.py:230-296 (pinned: golden F1/F2)
  * one whole micro-step + optimizer step        train.py:2719-2784

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .unet_ref import RefUNet, UNetConfig


# ---------------------------------------------------------------------------------------
# noise schedule (SDXL-base scheduler_config.json: scaled_linear, 0.00085 -> 0.012, 1000)
# ---------------------------------------------------------------------------------------

def ddpm_alphas_cumprod(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddpm_coefficients(alphas_cumprod: torch.Tensor, timesteps: torch.Tensor, dtype) -> tuple:
    """diffusers casts alphas_cumprod to the dtype of `original_samples` (bf16 latents =>
    bf16 coefficients) BEFORE the sqrt; restated here."""
    ac = alphas_cumprod.to(dtype=dtype)
    a = ac[timesteps] ** 0.5
    s = (1 - ac[timesteps]) ** 0.5
    return a.flatten(), s.flatten()


def make_noisy_and_target(mode: str, latents: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
                          alphas_cumprod: Optional[torch.Tensor] = None, jitter: Optional[torch.Tensor] = None):
    """Returns (noisy_latents, target, unet_conditioning). mode in {epsilon, v_prediction, rectified_flow}."""
    if mode == "rectified_flow":
        t = ((timesteps.float() + jitter) / 1000.0).clamp(0.0, 1.0)
        te = t.view(-1, 1, 1, 1)
        return (1 - te) * latents + te * noise, noise - latents, t * 1000.0
    a, s = ddpm_coefficients(alphas_cumprod, timesteps, latents.dtype)
    a = a.view(-1, 1, 1, 1)
    s = s.view(-1, 1, 1, 1)
    noisy = a * latents + s * noise
    target = (a * noise - s * latents) if mode == "v_prediction" else noise
    return noisy, target, timesteps


def weighted_mse_loss(pred, target, timesteps, curve=None):
    per = (pred.float() - target.float()).pow(2).flatten(1).mean(dim=1)
    if curve is None:
        w = torch.ones_like(per)
    else:
        w = curve.to(per.dtype)[timesteps.long().clamp(0, curve.shape[0] - 1)]
    return (per * w).mean()


def clip_grad_norm(grads: List[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ semantics: per-tensor L2 norms (in grad dtype), L2 of the
    stack, coef = max/(norm+1e-6) clamped to 1, grads scaled in place; returns pre-clip norm."""
    norms = torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads])
    total = torch.linalg.vector_norm(norms, 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef.to(g.dtype))
    return total


def adamw_debiased_step(p, g32, m, v, step, lr, beta1, beta2, eps, wd, debias):
    """Raven / Titan element math (raven.py:104-147). p any float dtype (updated in place via
    fp32 scratch and rounded back), m/v stored in their own dtype, math in fp32."""
    m32 = m.float().mul_(beta1).add_(g32, alpha=1.0 - beta1)
    v32 = v.float().mul_(beta2).addcmul_(g32, g32, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if debias < 1.0:
        bc1 = 1.0 - (1.0 - bc1) * debias
        bc2 = 1.0 - (1.0 - bc2) * debias
    p32 = p.float()
    if wd != 0:
        p32.mul_(1.0 - lr * wd)
    denom = v32.sqrt().div_(math.sqrt(bc2)).add_(eps)
    p32.addcdiv_(m32, denom, value=-(lr / bc1))
    p.copy_(p32)
    m.copy_(m32)
    v.copy_(v32)


def titan_accumulate(host_grad: Optional[torch.Tensor], grad: torch.Tensor) -> torch.Tensor:
    """titan.py:119-131: the first micro-step of a window copies the (bf16) gradient into the fp32 host buffer, later
    micro-steps add it in fp32."""
    g32 = grad.detach().to(torch.float32)
    return g32.clone() if host_grad is None else host_grad.add_(g32)


def titan_clip(host_grads: List[torch.Tensor], max_norm: float) -> torch.Tensor:
    """titan.py:162-184 (L2): per-tensor norms of the fp32 host gradients, norm of the stack, in-place scale by
    max_norm / (total + 1e-6) when that is below 1 (and max_norm > 0).  Returns the pre-clip norm."""
    norms = [torch.norm(g, 2.0) for g in host_grads]
    total = torch.norm(torch.stack(norms), 2.0)
    if max_norm > 0:
        coef = max_norm / (total + 1e-6)
        if coef < 1:
            for g in host_grads:
                g.mul_(coef)
    return total


class RefTrainer:
    """Whole-step oracle: the dataflow of train.py:2719-2784 on the CPU with RefUNet.
    `bf16=True`: params/grads in bf16 + autocast (the reference's only mode, train.py:273);
    `bf16=False`: everything fp32 (the 1e-3 oracle).  `ref_inputs=True` (fp32 only): the noise mix and the target are formed as
    the reference forms them for bf16 latents -- scheduler coefficients cast to bf16 BEFORE the square root (diffusers add_noise /
    get_velocity on a bf16 sample), x_t rounded to bf16 -- and only the UNet, the loss and the backward run in fp32:
    "fp32 arithmetic on the reference's own inputs".  Without it the fp32 run also differs from the reference by a COHERENT
    scale of up to 2^-9 on x_t (t = 417: +2.3e-3), which every GroupNorm's rstd hands on to the gradients (round 5)."""

    def __init__(self, cfg: UNetConfig, params: Dict[str, torch.Tensor], mode="epsilon", bf16=False,
                 ga=1, clip=1.0, lr=8e-7, betas=(0.9, 0.999), eps=1e-8, wd=0.01, debias=0.3,
                 momentum_dtype=torch.bfloat16, curve=None, frozen=(), ref_inputs=False):
        dt = torch.bfloat16 if bf16 else torch.float32
        self.cfg, self.mode, self.bf16, self.ga, self.clip = cfg, mode, bf16, ga, clip
        self.params = {k: v.detach().to(dt).clone().requires_grad_(k not in frozen) for k, v in params.items()}
        self.net = RefUNet(cfg, self.params)
        self.hyper = dict(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, wd=wd, debias=debias)
        self.mdt = momentum_dtype
        self.state: Dict[str, dict] = {}
        self.acp = ddpm_alphas_cumprod()
        self.curve = curve
        self.micro = 0
        self.ref_inputs = bool(ref_inputs) and not bf16

    def micro_step(self, latents, noise, timesteps, ctx, pooled, time_ids, jitter=None):
        dt = torch.bfloat16 if self.bf16 else torch.float32
        lat = latents.to(dt) if self.bf16 else latents.float()
        if self.ref_inputs:
            noisy, target, cond = make_noisy_and_target(self.mode, latents.bfloat16(), noise, timesteps, self.acp, jitter)
            noisy, target = noisy.bfloat16().float(), target.float()
        else:
            noisy, target, cond = make_noisy_and_target(self.mode, lat, noise, timesteps, self.acp, jitter)
        pred = self.net.forward(noisy.to(dt), cond, ctx.to(dt), pooled.to(dt), time_ids.to(dt).float()
                                if not self.bf16 else time_ids.to(dt), autocast_bf16=self.bf16)
        loss = weighted_mse_loss(pred, target, timesteps, self.curve)
        (loss / self.ga).backward()
        self.micro += 1
        self.last_pred = pred.detach()
        return float(loss.detach())

    def grads(self):
        return {k: p.grad for k, p in self.params.items() if p.grad is not None}

    def optimizer_step(self, lr=None):
        if lr is not None:
            self.hyper["lr"] = lr
        gl = [p.grad for p in self.params.values() if p.grad is not None]
        raw = clip_grad_norm(gl, self.clip if self.clip > 0 else float("inf"))
        with torch.no_grad():
            for k, p in self.params.items():
                if p.grad is None:
                    continue
                st = self.state.setdefault(k, dict(step=0, m=torch.zeros_like(p, dtype=self.mdt),
                                                   v=torch.zeros_like(p, dtype=self.mdt)))
                st["step"] += 1
                adamw_debiased_step(p, p.grad.float(), st["m"], st["v"], st["step"], **self.hyper)
                p.grad = None
        return float(raw)
