"""Global-norm gradient clip for the HIP path (train.py:2771-2781 -> torch.nn.utils.clip_grad_norm_).

For AozoraUNet the gradients are one flat bf16 buffer: one fused sum-of-squares pass over the
trainable ranges, the coefficient min(1, max_norm/(norm+1e-6)) computed on the device, and an in-place
bf16 rescale that is skipped when the coefficient is 1.  The norm is accumulated in fp32 (torch
rounds each per-tensor norm to bf16 first; stated deviation, more accurate)."""
from __future__ import annotations

import ctypes

import torch

from . import ops
from ._lib import lib


def clip_grad_norm_(unet, max_norm: float) -> torch.Tensor:
    """-> 0-d device tensor with the pre-clip global L2 norm (call .item() to read it)."""
    ws = ops.workspace(unet.device)
    ss, coef, norm = ws.small[4100:4101], ws.small[4101:4102], ws.small[4102:4103]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ranges = unet.trainable_ranges()
    for i, (a, b) in enumerate(ranges):
        ops.sumsq(unet.gflat[a:b], ss, i > 0)
    mx = float(max_norm) if max_norm and max_norm > 0 else float("inf")
    ops.clip_coef(ss, mx, coef, norm)
    for a, b in ranges:
        lib().call("az_scale_bf16", b - a, ctypes.c_void_p(unet.gflat.data_ptr() + a * 2), ctypes.c_void_p(coef.data_ptr()), st)
    return norm[0]
