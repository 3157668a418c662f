"""TitanAdamW -- drop-in for training_utils/optimizers/titan.py:8-296 on the HIP path.

Raven's update core (raven.py) + Titan's gradient residency: after every backward the gradients
leave the device for a pinned fp32 HOST buffer (first micro-step: copy, later micro-steps: add --
titan.py:119-131), the global-norm clip runs against those host gradients (titan.py:162-184) and
the step streams them back (titan.py:248-249).  API kept: post-accumulate hooks on ordinary
autograd parameters, `_cpu_grads`, `_cpu_grad_ready`, `zero_grad`, `clip_grad_norm`, `close`, the
single-owner RuntimeError, `save_cpu_state` / `load_cpu_state`.

AozoraUNet parameters never go through autograd hooks: the train loop calls `offload_flat()` after
each backward, which moves every trainable range of the flat gradient buffer in one launch per range
(kernel writes the pinned host buffer directly over the host link) and clears the device buffer.
"""
from __future__ import annotations

import ctypes
import math
import weakref
from typing import Dict

import torch

from .._lib import lib, AozoraError
from .raven import RavenAdamW, _storage_span


class TitanAdamW(RavenAdamW):
    _GDTYPE = 1
    _GSIZE = 4

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), weight_decay: float = 0.01, eps: float = 1e-8,
                 debias_strength: float = 1.0, momentum_dtype: torch.dtype = torch.bfloat16):
        super().__init__(params, lr=lr, betas=betas, weight_decay=weight_decay, eps=eps, debias_strength=debias_strength,
                         momentum_dtype=momentum_dtype)
        self._cpu_grads: Dict[torch.Tensor, torch.Tensor] = {}
        self._cpu_grad_ready = set()
        self._hook_handles = []
        self._closed = False
        self._ghost: Dict[object, torch.Tensor] = {}
        self._gspan: Dict[object, tuple] = {}
        for group in self.param_groups:
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                owner_ref = getattr(p, "_titan_optimizer_owner", None)
                owner = owner_ref() if callable(owner_ref) else None
                if owner is not None and owner is not self:
                    self.close()
                    raise RuntimeError("A parameter is already owned by another live TitanAdamW. "
                                       "Close the old optimizer before creating a replacement.")
                p._titan_optimizer_owner = weakref.ref(self)
                _, n, flat = _storage_span(p)
                if flat is not None:
                    uo, off = flat
                    if uo not in self._ghost:
                        self._ghost[uo] = torch.zeros(uo.flat_numel, dtype=torch.float32).pin_memory()
                    _, sshape, lshape = uo._slots[p._az_name]
                    g = self._ghost[uo][off:off + math.prod(sshape)].view(sshape)
                    if len(sshape) == 4:
                        g = g.permute(0, 3, 1, 2)[:, :lshape[1]]
                    self._gspan[p] = (uo, off, n)
                else:
                    buf = torch.zeros(n, dtype=torch.float32).pin_memory()
                    self._ghost[id(p)] = buf
                    g = buf.view(p.shape)
                    self._gspan[p] = (id(p), 0, n)
                    me = weakref.ref(self)

                    def hook(param, me=me):
                        o = me()
                        if o is not None:
                            o._offload_gradient(param)
                    self._hook_handles.append(p.register_post_accumulate_grad_hook(hook))
                self._cpu_grads[p] = g

    # -------------------------------------------------------------------------------------------
    def _host_grad_ptr(self, p):
        key, off, n = self._gspan[p]
        return self._ghost[key].data_ptr() + off * 4, n

    def _offload_gradient(self, param):
        """titan.py:119-131 for one ordinary (contiguous bf16) parameter."""
        if param.grad is None:
            return
        g = param.grad
        if g.dtype != torch.bfloat16 or not g.is_contiguous():
            raise AozoraError("TitanAdamW (HIP) needs contiguous bf16 gradients")
        hp, n = self._host_grad_ptr(param)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib().call("az_titan_offload", n, ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(hp), ctypes.c_void_p(0),
                   int(param in self._cpu_grad_ready), st)
        self._cpu_grad_ready.add(param)
        g.record_stream(torch.cuda.current_stream())
        param.grad = None

    def offload_flat(self, unet):
        """Move every trainable range of unet.gflat to the host buffer (add on later micro-steps),
        clear the device gradients, mark the parameters ready (the fused-backward form of the hooks)."""
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ps = [p for g in self.param_groups for p in g["params"] if getattr(p, "_az_owner", None) is unet and p.requires_grad]
        if not ps:
            return
        acc = int(ps[0] in self._cpu_grad_ready)
        gh = self._ghost[unet]
        for a, b in unet.trainable_ranges():
            lib().call("az_titan_offload", b - a, ctypes.c_void_p(unet.gflat.data_ptr() + a * 2),
                       ctypes.c_void_p(gh.data_ptr() + a * 4), ctypes.c_void_p(0), acc, st)
            lib().call("az_memset_async", ctypes.c_void_p(unet.gflat.data_ptr() + a * 2), 0, (b - a) * 2, st)
        for p in ps:
            self._cpu_grad_ready.add(p)
            p.grad = None

    def _grad_ptr(self, p):
        if p in self._cpu_grad_ready:
            return self._host_grad_ptr(p)[0]
        return None

    def close(self):
        if getattr(self, "_closed", True):
            return
        self._closed = True
        for h in self._hook_handles:
            h.remove()
        self._hook_handles.clear()
        for p in self._cpu_grads:
            ref = getattr(p, "_titan_optimizer_owner", None)
            if callable(ref) and ref() is self:
                delattr(p, "_titan_optimizer_owner")
        self._cpu_grad_ready.clear()
        self._cpu_grads.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def zero_grad(self, set_to_none: bool = True):
        if set_to_none:
            self._cpu_grad_ready.clear()
        else:
            torch.cuda.synchronize()
            for p in self._cpu_grad_ready:
                self._cpu_grads[p].zero_()
        super().zero_grad(set_to_none)

    def _ready_spans(self):
        spans = []
        for g in self.param_groups:
            for p in g["params"]:
                if p in self._cpu_grad_ready:
                    ptr, n = self._host_grad_ptr(p)
                    if spans and spans[-1][0] + spans[-1][1] * 4 == ptr:
                        spans[-1][1] += n
                    else:
                        spans.append([ptr, n])
        return spans

    def clip_grad_norm(self, max_norm, norm_type=2.0):
        """titan.py:162-184 (L2 only): global norm of the HOST gradients; scales them in place when
        max_norm/(norm+1e-6) < 1.  Returns the pre-clip norm as a 0-d tensor."""
        if float(norm_type) != 2.0:
            raise AozoraError("TitanAdamW (HIP) implements the L2 norm only")
        spans = self._ready_spans()
        if not spans:
            return torch.tensor(0.0)
        self._ensure_runtime()
        from .. import ops
        ws = ops.workspace(self.param_device)
        ss = ws.small[4110:4111]
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i, (ptr, n) in enumerate(spans):
            lib().call("az_sumsq", n, ctypes.c_void_p(ptr), 1, ctypes.c_void_p(ss.data_ptr()), int(i > 0),
                       ctypes.c_void_p(ws.scratch.data_ptr()), st)
        total = float(ss.item()) ** 0.5
        if max_norm > 0:
            coef = max_norm / (total + 1e-6)
            if coef < 1:
                c = ws.small[4111:4112]
                c.fill_(coef)
                for ptr, n in spans:
                    lib().call("az_scale_f32", n, ctypes.c_void_p(ptr), ctypes.c_void_p(c.data_ptr()), st)
        return torch.tensor(total)
