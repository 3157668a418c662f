"""RavenAdamW -- drop-in for training_utils/optimizers/raven.py:7-222 on the HIP path.

Same constructor, param_groups, step / zero_grad / save_cpu_state / load_cpu_state / state_dict
surface and the same element arithmetic (fp32 math, `debias_strength`, m/v stored on the HOST in
`momentum_dtype`).  What differs is the execution model (MI355X-first, not a translation):

  * host m/v live in ONE pinned buffer pair per parameter owner instead of 2*1680 pageable tensors;
    `state[p]["exp_avg"]` / `["exp_avg_sq"]` are views of it, so the reference's state layout is kept;
  * instead of a Python loop of 1680 x (2 H2D + ~10 elementwise launches + 2 D2H) with a 3*max_numel
    fp32 scratch (raven.py:104-147), adjacent parameters are merged into flat ranges and each range
    is streamed through `az_raven_step_ex`: chunked H2D(m,v) || fused AdamW kernel || D2H(m,v) on
    three HIP streams with double-buffered device staging.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List

import torch
from torch.optim import Optimizer

from .._lib import lib, AozoraError

_MD = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}
CHUNK_ELEMS = 16 << 20


def _storage_span(p: torch.Tensor):
    """(data_ptr of first storage element used, number of storage elements spanned) for a dense,
    possibly permuted view; AozoraUNet parameters report their padded flat slot."""
    owner = getattr(p, "_az_owner", None)
    if owner is not None:
        off, st, _ = owner._slots[p._az_name]
        n = ((math.prod(st) + 63) // 64) * 64
        return owner.pflat.data_ptr() + off * 2, n, (owner, off)
    if not p.is_contiguous():
        raise AozoraError("RavenAdamW (HIP) needs contiguous parameters or AozoraUNet parameters")
    return p.data_ptr(), p.numel(), None


class _HostState:
    """m / v for one contiguous device span: pinned host memory (the reference's residency), or -- state_on_device -- device memory."""

    def __init__(self, numel, dtype, device=None):
        if device is None:
            self.m = torch.zeros(numel, dtype=dtype).pin_memory()
            self.v = torch.zeros(numel, dtype=dtype).pin_memory()
        else:
            self.m = torch.zeros(numel, dtype=dtype, device=device)
            self.v = torch.zeros(numel, dtype=dtype, device=device)


class RavenAdamW(Optimizer):
    _GRAD_SOURCE = "device"

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.98), weight_decay: float = 0.06, eps: float = 1e-8,
                 debias_strength: float = 0.9, momentum_dtype: torch.dtype = torch.bfloat16, state_on_device: bool = False):
        """state_on_device (not in the reference): keep exp_avg / exp_avg_sq resident in device memory instead of pinned host memory
        streamed through staging buffers every step (raven.py:83-84, 114-117: the reference's way onto 24 GB cards; 10.3 GB for
        SDXL-base in bf16).  Same kernel arithmetic on the same values: bit-identical parameters; `state[p]["exp_avg"]` is then a
        device tensor, save_cpu_state() / load_cpu_state() still speak the reference's CPU layout."""
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        valid = [torch.float32, torch.float16, torch.bfloat16]
        if momentum_dtype not in valid:
            raise ValueError(f"momentum_dtype must be one of {valid}, got {momentum_dtype}")
        defaults = dict(lr=lr, betas=betas, weight_decay=weight_decay, eps=eps, debias_strength=debias_strength,
                        momentum_dtype=momentum_dtype)
        super().__init__(params, defaults)
        self._momentum_dtype = momentum_dtype
        self._state_on_device = bool(state_on_device)
        if self._state_on_device and self._GDTYPE != 0:
            raise AozoraError("state_on_device needs device gradients (RavenAdamW); TitanAdamW keeps its gradients in host memory -- use dist.ShardedTitan")
        self.max_numel = 0
        self.param_device = None
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    if self.param_device is None:
                        self.param_device = p.device
                    self.max_numel = max(self.max_numel, p.numel())
        if self.param_device is not None and self.param_device.type != "cuda":
            raise AozoraError("RavenAdamW (HIP) needs parameters on a HIP device; there is no CPU fallback")
        self._host: Dict[object, _HostState] = {}     # owner unet -> flat host state ; id(p) -> per-tensor
        self._spans: Dict[object, tuple] = {}
        self._hyper_ev = None
        self._staging = None
        self._hyper_host = None
        self._hyper_dev = None
        self._copy_streams = None
        self.clip_coef = None      # optional device fp32[1]; multiplies grads inside the kernel

    # -------------------------------------------------------------------------------------------
    def _ensure_runtime(self):
        if self._staging is None:
            esz = 4 if self._momentum_dtype == torch.float32 else 2
            self._staging = torch.empty(4 * CHUNK_ELEMS * esz, dtype=torch.uint8, device=self.param_device)
            self._hyper_host = torch.zeros(64, 8, dtype=torch.float32).pin_memory()
            self._hyper_dev = torch.zeros(64, 8, dtype=torch.float32, device=self.param_device)
            from ..streams import host_link_streams
            self._copy_streams = host_link_streams(self.param_device)      # one warmed pair per device and process (streams.py)
            self._one = torch.ones(1, dtype=torch.float32, device=self.param_device)

    def _state_views(self, p):
        """Create state[p] (step 0, zero m/v as views of pinned host memory) on first use."""
        st = self.state[p]
        if "step" in st:
            return st
        ptr, n, flat = _storage_span(p)
        if flat is not None:
            owner, off = flat
            if owner not in self._host:
                self._host[owner] = _HostState(owner.flat_numel, self._momentum_dtype, self.param_device if self._state_on_device else None)
            hs = self._host[owner]
            _, sshape, lshape = owner._slots[p._az_name]
            sn = math.prod(sshape)
            m, v = hs.m[off:off + sn].view(sshape), hs.v[off:off + sn].view(sshape)
            if len(sshape) == 4:
                m, v = m.permute(0, 3, 1, 2)[:, :lshape[1]], v.permute(0, 3, 1, 2)[:, :lshape[1]]
            self._spans[p] = (owner, off, n)
        else:
            hs = _HostState(n, self._momentum_dtype, self.param_device if self._state_on_device else None)
            self._host[id(p)] = hs
            m, v = hs.m.view(p.shape), hs.v.view(p.shape)
            self._spans[p] = (id(p), 0, n)
        st["step"] = 0
        st["exp_avg"], st["exp_avg_sq"] = m, v
        return st

    def zero_grad(self, set_to_none: bool = True):
        """train.py:2784 `optimizer.zero_grad(set_to_none=True)` is the loop's only gradient reset.  AozoraUNet gradients
        accumulate in the owner's flat buffer (p.grad is a view or None), so the buffer itself is cleared here --
        dropping the views alone would let the next accumulation window start from the previous window's sums."""
        owners = []
        for g in self.param_groups:
            for p in g["params"]:
                o = getattr(p, "_az_owner", None)
                if o is not None and all(o is not q for q in owners):
                    owners.append(o)
        for o in owners:
            for a, b in o.trainable_ranges():
                o.gflat[a:b].zero_()
        super().zero_grad(set_to_none)

    def _grad_ptr(self, p):
        """device bf16 gradient pointer matching the storage span of p (None => skip this param)."""
        if p.grad is None:
            return None
        owner = getattr(p, "_az_owner", None)
        if owner is not None:
            off = owner._slots[p._az_name][0]
            return owner.gflat.data_ptr() + off * 2
        g = p.grad
        if g.dtype != torch.bfloat16 or not g.is_contiguous():
            raise AozoraError("RavenAdamW (HIP) needs contiguous bf16 gradients")
        return g.data_ptr()

    _GDTYPE = 0
    _GSIZE = 2

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure_runtime()
        segs: List[list] = []   # [p_ptr, g_ptr, host_key, host_off, numel, hyper-tuple]
        for group in self.param_groups:
            lr = group["lr"]
            beta1, beta2 = group["betas"]
            wd, eps, debias = group["weight_decay"], group["eps"], group["debias_strength"]
            wd_factor = 1.0 - lr * wd if wd != 0 else 1.0
            for p in group["params"]:
                gptr = self._grad_ptr(p)
                if gptr is None:
                    continue
                if p.dtype != torch.bfloat16:
                    raise AozoraError("RavenAdamW (HIP) updates bf16 parameters (the reference's only mode, train.py:273)")
                st = self._state_views(p)
                st["step"] += 1
                step = st["step"]
                bc1 = 1.0 - beta1 ** step
                bc2 = 1.0 - beta2 ** step
                if debias < 1.0:
                    bc1 = 1.0 - (1.0 - bc1) * debias
                    bc2 = 1.0 - (1.0 - bc2) * debias
                hyper = (lr, beta1, beta2, eps, wd_factor, lr / bc1, math.sqrt(bc2), 0.0)
                pptr, n, _ = _storage_span(p)
                key, hoff, _ = self._spans[p]
                last = segs[-1] if segs else None
                if (last is not None and last[5] == hyper and last[2] is key and last[0] + last[4] * 2 == pptr
                        and last[1] + last[4] * self._GSIZE == gptr and last[3] + last[4] == hoff):
                    last[4] += n
                else:
                    segs.append([pptr, gptr, key, hoff, n, hyper])
        if not segs:
            return loss
        if len(segs) > self._hyper_host.shape[0]:
            k = len(segs)
            self._hyper_host = torch.zeros(k, 8, dtype=torch.float32).pin_memory()
            self._hyper_dev = torch.zeros(k, 8, dtype=torch.float32, device=self.param_device)
        if self._hyper_ev is not None:
            self._hyper_ev.synchronize()      # the previous step's H2D of this pinned buffer has completed
        for i, s in enumerate(segs):
            self._hyper_host[i] = torch.tensor(s[5], dtype=torch.float32)
        self._hyper_dev[:len(segs)].copy_(self._hyper_host[:len(segs)], non_blocking=True)
        sc = torch.cuda.current_stream()
        self._hyper_ev = torch.cuda.Event()
        self._hyper_ev.record(sc)
        esz = 4 if self._momentum_dtype == torch.float32 else 2
        coef = self.clip_coef if self.clip_coef is not None else None
        L = lib()
        for i, (pptr, gptr, key, hoff, n, _) in enumerate(segs):
            hs = self._host[key]
            if self._state_on_device:          # resident moments: the update kernel alone, on the compute stream
                L.call("az_adamw_flat_ex", n, ctypes.c_void_p(pptr), ctypes.c_void_p(gptr), self._GDTYPE,
                       ctypes.c_void_p(hs.m.data_ptr() + hoff * esz), ctypes.c_void_p(hs.v.data_ptr() + hoff * esz),
                       _MD[self._momentum_dtype], ctypes.c_void_p(self._hyper_dev[i].data_ptr()),
                       ctypes.c_void_p(coef.data_ptr() if coef is not None else 0), ctypes.c_void_p(sc.cuda_stream))
                continue
            L.call("az_raven_step_ex", n, ctypes.c_void_p(pptr), ctypes.c_void_p(gptr), self._GDTYPE,
                   ctypes.c_void_p(hs.m.data_ptr() + hoff * esz), ctypes.c_void_p(hs.v.data_ptr() + hoff * esz),
                   _MD[self._momentum_dtype], ctypes.c_void_p(self._hyper_dev[i].data_ptr()),
                   ctypes.c_void_p(coef.data_ptr() if coef is not None else 0), ctypes.c_void_p(self._staging.data_ptr()),
                   CHUNK_ELEMS, ctypes.c_void_p(sc.cuda_stream), ctypes.c_void_p(self._copy_streams[0].cuda_stream),
                   ctypes.c_void_p(self._copy_streams[1].cuda_stream))
        self.clip_coef = None
        for key in self._host:
            if hasattr(key, "mark_params_dirty"):
                key.mark_params_dirty()
        return loss

    # -------------------------------------------------------------------------------------------
    def state_dict(self):
        torch.cuda.synchronize()
        sd = super().state_dict()
        sd["_momentum_dtype"] = self._momentum_dtype
        return sd

    def save_cpu_state(self):
        """raven.py:156-169: {i: {step, exp_avg_cpu, exp_avg_sq_cpu}} indexed by position among
        requires_grad params, plus "_momentum_dtype"."""
        torch.cuda.synchronize()
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        out = {"_momentum_dtype": self._momentum_dtype}
        for i, p in enumerate(ps):
            if p in self.state and "step" in self.state[p]:
                st = self.state[p]
                out[i] = {"step": st.get("step", 0), "exp_avg_cpu": st.get("exp_avg").detach().cpu().clone(),
                          "exp_avg_sq_cpu": st.get("exp_avg_sq").detach().cpu().clone()}
        return out

    def load_cpu_state(self, cpu_state):
        torch.cuda.synchronize()
        saved_dtype = cpu_state.get("_momentum_dtype", self._momentum_dtype)
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        for i, p in enumerate(ps):
            if i not in cpu_state:
                continue
            saved = cpu_state[i]
            m = saved.get("exp_avg", saved.get("exp_avg_cpu"))
            v = saved.get("exp_avg_sq", saved.get("exp_avg_sq_cpu"))
            step = saved.get("step", 0)
            if torch.is_tensor(step):
                step = int(step.item())
            st = self._state_views(p)
            st["step"] = step
            if m is not None:
                st["exp_avg"].copy_(m.to(self._momentum_dtype))
            if v is not None:
                st["exp_avg_sq"].copy_(v.to(self._momentum_dtype))
        if saved_dtype != self._momentum_dtype:
            print(f"[RavenAdamW] Loaded state saved in {saved_dtype}, converted to {self._momentum_dtype}.")
