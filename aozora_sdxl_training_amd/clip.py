"""Global-norm gradient clip for the HIP path (train.py:2771-2781 -> torch.nn.utils.clip_grad_norm_).

For AozoraUNet the gradients are one flat bf16 buffer: one fused sum-of-squares pass over the
trainable ranges, the coefficient min(1, max_norm/(norm+1e-6)) computed on the device, and an in-place
bf16 rescale that is skipped when the coefficient is 1.  The norm is accumulated in fp32 (torch
rounds each per-tensor norm to bf16 first; stated deviation, more accurate)."""
from __future__ import annotations

import ctypes
import math

import torch

from . import ops
from ._lib import lib, AozoraError


def _grad_ranges(target):
    """-> (owner unet or None, [(start, end)] flat element ranges, [loose gradient tensors]).  `target` is an AozoraUNet
    (all trainable ranges) or what train.py:2775 passes to torch.nn.utils.clip_grad_norm_: an iterable of parameters.
    Parameters of an AozoraUNet map to their slots of the flat gradient buffer (adjacent slots merge); parameters without a
    gradient are skipped like torch does; any other device tensor is clipped through its own .grad."""
    if hasattr(target, "trainable_ranges"):
        return target, target.trainable_ranges(), []
    if isinstance(target, torch.Tensor):
        target = [target]
    owner, spans, loose = None, [], []
    for p in target:
        o = getattr(p, "_az_owner", None)
        if o is None:
            if p.grad is not None:
                g = p.grad
                if not (g.is_cuda and g.is_contiguous() and g.dtype in (torch.bfloat16, torch.float32)):
                    raise AozoraError("clip_grad_norm_ (HIP) needs contiguous bf16 / fp32 device gradients")
                loose.append(g)
            continue
        if owner is not None and o is not owner:
            raise AozoraError("clip_grad_norm_: parameters of two different AozoraUNet objects in one call")
        owner = o
        if p.grad is None or not p.requires_grad:
            continue
        off, st, _ = o._slots[p._az_name]
        n = ((math.prod(st) + 63) // 64) * 64
        spans.append((off, off + n))
    spans.sort()
    merged = []
    for a, b in spans:
        if merged and merged[-1][1] == a:
            merged[-1][1] = b
        else:
            merged.append([a, b])
    return owner, [(a, b) for a, b in merged], loose


def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) of train.py:2775 on the HIP path: `parameters` is the list of
    parameters the loop passes (or the AozoraUNet itself).  Returns the pre-clip global L2 norm as a 0-d device tensor
    (train.py:2780 reads it with .item()); gradients are scaled in place by min(1, max_norm / (norm + 1e-6))."""
    owner, ranges, loose = _grad_ranges(parameters)
    dev = owner.device if owner is not None else (loose[0].device if loose else None)
    if dev is None:
        return torch.zeros((), dtype=torch.float32)
    ws = ops.workspace(dev)
    ss, coef, norm = ws.small[4100:4101], ws.small[4101:4102], ws.small[4102:4103]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    first = True
    for a, b in ranges:
        ops.sumsq(owner.gflat[a:b], ss, not first)
        first = False
    for g in loose:
        ops.sumsq(g.view(-1), ss, not first)
        first = False
    if first:
        ss.zero_()
    mx = float(max_norm) if max_norm and max_norm > 0 else float("inf")
    ops.clip_coef(ss, mx, coef, norm)
    for a, b in ranges:
        lib().call("az_scale_bf16", b - a, ctypes.c_void_p(owner.gflat.data_ptr() + a * 2), ctypes.c_void_p(coef.data_ptr()), st)
    for g in loose:
        lib().call("az_scale_bf16" if g.dtype == torch.bfloat16 else "az_scale_f32", g.numel(), ctypes.c_void_p(g.data_ptr()),
                   ctypes.c_void_p(coef.data_ptr()), st)
    return norm[0]
