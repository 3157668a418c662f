"""weighted_sdxl_mse_loss -- the loss seam of the reference loop (train.py:2408-2416, called at train.py:2763) as a callable
on the HIP path, for callers that keep the reference's loop body verbatim:

    pred = unet(...).sample                       # (B,C,H,W) bf16, AozoraUNet.__call__
    loss = weighted_sdxl_mse_loss(pred, target, timesteps, timestep_loss_weights)
    (loss / GA).backward()

Forward and d(loss)/d(pred) come from ONE launch of az_mse_loss_fwd_bwd (the per-sample mean over (C,H,W) does not depend on
the element order, so the NCHW tensors are handed to the kernel as [B][1 channel][C*H*W]); backward scales the saved bf16
gradient by the incoming scalar with az_scale_bf16 (device-resident coefficient: no host sync).  TrainStep uses the same
kernel directly on the NHWC prediction; this wrapper exists for the drop-in seam."""
from __future__ import annotations

import ctypes

import torch

from . import ops
from ._lib import lib


class _WeightedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, w):
        B = pred.shape[0]
        n = pred[0].numel()
        dev = pred.device
        p2 = pred.contiguous().view(B, 1, n, 1)                    # "NHWC" with H*W = C*H*W rows of one channel
        t2 = target.contiguous().view(B, 1, 1, n)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        per = torch.empty(B, dtype=torch.float32, device=dev)
        dpred = torch.empty(B, 1, n, 1, dtype=torch.bfloat16, device=dev)
        ops.mse_loss_fwd_bwd(p2, t2, w, 1.0, loss, per, dpred)
        ctx.save_for_backward(dpred)
        ctx.shape = pred.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        out = dpred.clone()
        coef = g.detach().to(torch.float32).reshape(1).contiguous()
        lib().call("az_scale_bf16", out.numel(), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(coef.data_ptr()),
                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return out.view(ctx.shape), None, None


def weighted_sdxl_mse_loss(pred, target, timesteps, timestep_loss_weights=None):
    """train.py:2408-2416.  pred (B,C,H,W) bf16 on the HIP device (differentiable), target (B,C,H,W) any float dtype,
    timesteps (B,) integer, timestep_loss_weights None or a 1-D curve indexed by clamp(timestep, 0, len-1).  Returns the 0-d
    fp32 loss: mean over the batch of weight * mean over (C,H,W) of the squared error (fp32 accumulation)."""
    if not pred.is_cuda or pred.dtype != torch.bfloat16:
        raise ops.AozoraError("weighted_sdxl_mse_loss (HIP) needs the bf16 prediction on the device; there is no CPU fallback")
    B = pred.shape[0]
    dev = pred.device
    if timestep_loss_weights is None:
        w = torch.ones(B, dtype=torch.float32, device=dev)
    else:
        curve = timestep_loss_weights.to(device=dev, dtype=torch.float32)
        w = curve[timesteps.to(dev).long().clamp(0, curve.shape[0] - 1)].contiguous()
    return _WeightedMSE.apply(pred, target.to(device=dev, dtype=torch.float32), w)
