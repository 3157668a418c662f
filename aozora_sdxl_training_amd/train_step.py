"""One micro-step of the reference's hot loop (train.py:2719-2767) on the HIP path.

    noise-mix + target (eps / v_prediction / rectified_flow)   train.py:2743-2758  -> az_noise_target
    pred = unet(...)                                            train.py:2760-2761  -> AozoraUNet.forward_nhwc
    loss = weighted_sdxl_mse_loss(pred, target, ts, curve)      train.py:2763       -> az_mse_loss_fwd_bwd
    (loss / GA).backward()                                      train.py:2765       -> AozoraUNet.backward_nhwc

Inputs that the reference draws from device RNG (noise, RF jitter) are INPUTS here (drawn by the
caller on the CPU generator keyed by SEED+micro_step; SURVEY.md section 7 "RNG").  Everything that
changes per step lives in static device buffers, so the whole launch sequence of a bucket is
captured once into a hipGraph and replayed (no tracing compiler; guide section 6 G9).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import ops
from ._lib import lib, AozoraError
from .schedule import ddpm_coef_tables
from .unet import AozoraUNet
from .streams import check as stream_check
from .tape import NativeTape, fuse_records, disarm_stop_event

BF16, F32 = torch.bfloat16, torch.float32
MODES = {"epsilon": 0, "v_prediction": 1, "rectified_flow": 2}


class _Bucket:
    """Static input/output buffers + captured graph for one (B, H, W, ctx_len) shape."""

    def __init__(self, dev, B, C, H, W, L, ctx_dim, pooled_dim):
        self.lat = torch.empty(B, C, H, W, dtype=BF16, device=dev)
        self.noise = torch.empty(B, C, H, W, dtype=F32, device=dev)
        self.ctx = torch.empty(B, L, ctx_dim, dtype=BF16, device=dev)
        self.pooled = torch.empty(B, pooled_dim, dtype=BF16, device=dev)
        self.host = [torch.empty(4, B, dtype=F32).pin_memory() for _ in range(2)]   # ca, cb, tcond, w (double-buffered)
        self.host_ev = [None, None]
        self.dev = torch.empty(4, B, dtype=F32, device=dev)
        self.x8 = torch.zeros(B, H, W, 8, dtype=BF16, device=dev)
        self.target = torch.empty(B, C, H, W, dtype=F32, device=dev)
        self.dpred8 = torch.zeros(B, H, W, 8, dtype=BF16, device=dev)
        self.loss = torch.zeros(1, dtype=F32, device=dev)
        self.per_sample = torch.zeros(B, dtype=F32, device=dev)
        self.graph = None
        self.parity = 0
        self.runs = 0
        self.pred = None


class TrainStep:
    def __init__(self, unet: AozoraUNet, mode: str = "epsilon", grad_accum: int = 1, world_size: int = 1,
                 loss_curve: Optional[torch.Tensor] = None, use_graph: bool = True, latent_dtype=BF16,
                 double_buffer: bool = False):
        if mode not in MODES:
            raise ValueError(f"unknown prediction type {mode!r}")
        self.unet, self.mode, self.ga, self.world = unet, mode, int(grad_accum), int(world_size)
        self.use_graph = use_graph
        self.use_tape = unet.policy.host_tape
        self.native_tape = unet.policy.native_tape      # re-issue the tape from C (az_tape_play); False: from Python
        # double_buffer: two activation pools used alternately, so that a micro-step may leave its weight-gradient branch
        # running (micro_step(defer_join=True)) under the next micro-step's forward -- which has no parameter-gradient
        # work of its own and leaves CUs idle.  Costs a second activation pool (50.8 GiB at B=4, 1024^2).
        self.double_buffer = bool(double_buffer) and not use_graph
        self._parity = 0
        self.curve = (loss_curve.float().cpu() if loss_curve is not None else None)
        self.tab_a, self.tab_b = ddpm_coef_tables(latent_dtype)
        # One data-gradient stream per UNet, shared by all its TrainStep objects, at NORMAL priority: a high-priority stream
        # gains nothing (same-box A/B: 137.3 vs 137.3 ms per micro-step, 1152.5 vs 1152.7 ms per iteration) and the 3rd, 4th ...
        # high-priority stream a process uses can land on a hardware queue where the two-stream step thrashes (190-240 ms
        # per micro-step; streams.py).  streams.check() logs what the probes say about the pair.
        if getattr(unet, "_main_stream", None) is None:
            unet._main_stream = torch.cuda.Stream(device=unet.device, priority=unet.policy.main_priority)
            stream_check(unet._main_stream, unet._sides[0], "data-gradient stream / weight-gradient stream")
        self.stream = unet._main_stream
        self._buckets: Dict[tuple, _Bucket] = {}
        self.last_pred_nhwc = None

    # ---------------------------------------------------------------------------------------------
    def _coefficients(self, bk: _Bucket, timesteps, jitter, time_ids, weight_scale=1.0):
        ts = torch.as_tensor(timesteps).long().cpu()
        slot = bk.runs & 1
        if bk.host_ev[slot] is not None:
            bk.host_ev[slot].synchronize()      # the H2D copy that last used this pinned buffer has completed
        h = bk.host[slot]
        if self.mode == "rectified_flow":
            if jitter is None:
                raise ValueError("rectified_flow needs the jitter tensor (train.py:2744-2745)")
            tc = ((ts.float() + jitter.float().cpu()) / 1000.0).clamp(0.0, 1.0)
            h[0], h[1], h[2] = 1.0 - tc, tc, tc * 1000.0
        else:
            h[0], h[1], h[2] = self.tab_a[ts], self.tab_b[ts], ts.float()
        h[3] = 1.0 if self.curve is None else self.curve[ts.clamp(0, self.curve.shape[0] - 1)]
        if weight_scale != 1.0:
            h[3] *= float(weight_scale)
        if h.numel() <= 256:
            return h                              # rides in the arguments of the staging launch (_stage): no H2D copy, no event
        bk.dev.copy_(h, non_blocking=True)
        bk.host_ev[slot] = torch.cuda.Event()
        bk.host_ev[slot].record(torch.cuda.current_stream())
        return None

    def _stage(self, pairs, coef, coef_dev):
        """dst.copy_(src) for every (dst, src) pair + the coefficient table, as ONE launch (az_stage_inputs) for the pairs whose
        source already is a contiguous device tensor of the destination's dtype; torch copies for the rest (host sources).
        As hipMemcpyAsync copies the six placements ran as blit kernels with 0.1-0.4 ms of idle stream around each
        (tools/trace_gaps.py: ~1 ms per micro-step)."""
        segs = []
        for dst, src in pairs:
            nb = src.numel() * src.element_size()
            if (src.device == dst.device and src.dtype == dst.dtype and src.numel() == dst.numel() and src.is_contiguous()
                    and dst.is_contiguous() and nb % 4 == 0 and src.data_ptr() % 4 == 0 and dst.data_ptr() % 4 == 0 and nb > 0):
                segs.append((src, dst, nb))
            else:
                dst.copy_(src, non_blocking=True)
        if not segs and coef is None:
            return
        n = len(segs)
        srcs = (ctypes.c_void_p * max(1, n))(*[x[0].data_ptr() for x in segs])
        dsts = (ctypes.c_void_p * max(1, n))(*[x[1].data_ptr() for x in segs])
        nbs = (ctypes.c_long * max(1, n))(*[x[2] for x in segs])
        rc = lib()._fn["az_stage_inputs"](n, ctypes.cast(srcs, ctypes.c_void_p), ctypes.cast(dsts, ctypes.c_void_p), ctypes.cast(nbs, ctypes.c_void_p),
                                          0 if coef is None else coef.numel(), None if coef is None else ctypes.c_void_p(coef.data_ptr()),
                                          None if coef is None else ctypes.c_void_p(coef_dev.data_ptr()), ctypes.c_void_p(self.stream.cuda_stream))
        if rc != 0:
            raise AozoraError(f"az_stage_inputs failed with code {rc}")

    def _host_inputs_in_one_copy(self, bk: _Bucket, pairs):
        """Inputs that arrive as HOST tensors (trainer.train hands over what the DataLoader collated, train.py:2731-2741) travel as
        ONE pinned, asynchronous H2D copy into a device staging buffer; the placements into the step's static buffers then ride in
        the az_stage_inputs launch with the device-resident ones.  As five `dst.copy_(pageable host tensor)` calls they were five
        staged hipMemcpyAsync operations on the data-gradient stream behind the previous micro-step's kernels -- a pageable copy
        may hold the host until the stream reaches it, which stops the host from queueing the next micro-step while this one
        runs (bench.py's through_trainer leg: 0.946 against 1.045 it/s on one box, equal on others)."""
        host = [(i, src) for i, (dst, src) in enumerate(pairs) if src.device.type == "cpu" and src.numel() == dst.numel() and src.dtype == dst.dtype]
        if not host:
            return pairs
        offs, total = [], 0
        for _, src in host:
            offs.append(total)
            total += (src.numel() * src.element_size() + 15) // 16 * 16
        if getattr(bk, "hstage", None) is None or bk.hstage[0].numel() < total:
            bk.hstage = [torch.empty(total, dtype=torch.uint8).pin_memory() for _ in range(2)]
            bk.hstage_ev = [None, None]
            bk.dstage = torch.empty(total, dtype=torch.uint8, device=bk.lat.device)
        slot = bk.runs & 1
        if bk.hstage_ev[slot] is not None:
            bk.hstage_ev[slot].synchronize()         # the copy that last read this pinned buffer (two micro-steps ago) has completed
        hb = bk.hstage[slot]
        out = list(pairs)
        for (i, src), o in zip(host, offs):
            nb = src.numel() * src.element_size()
            hb[o:o + nb].view(src.dtype).view(src.shape).copy_(src)
            out[i] = (pairs[i][0], bk.dstage[o:o + nb].view(src.dtype).view(src.shape))
        bk.dstage[:total].copy_(hb[:total], non_blocking=True)
        bk.hstage_ev[slot] = torch.cuda.Event()
        bk.hstage_ev[slot].record(torch.cuda.current_stream())
        return out

    def _launch_sequence(self, bk: _Bucket, after_tail=None):
        u = self.unet
        B, C, H, W = bk.lat.shape
        ops.noise_target(MODES[self.mode], bk.lat, bk.noise, bk.dev[0], bk.dev[1], bk.x8, bk.target)
        u.begin_step((B, H, W, bk.ctx.shape[1], bk.parity))
        pred = u.forward_nhwc(bk.x8, bk.dev[2], bk.ctx, bk.pooled, bk.tids)
        ops.mse_loss_fwd_bwd(pred.t.view(B, H, W, C), bk.target, bk.dev[3], 1.0 / (self.ga * self.world), bk.loss,
                             bk.per_sample, bk.dpred8)
        u.backward_nhwc(pred, bk.dpred8, after_tail=after_tail)
        bk.pred = pred.t

    def _eager(self, bk, after_tail):
        """Eager issue with a host launch tape (see _lib._Lib.recorder): the first run of a bucket allocates its pools,
        the second is recorded, later runs re-issue the recorded launches.  The tape is keyed by everything that shapes
        the sequence: the freeze mask, the issue mode, and whether profiling brackets are on."""
        u, L = self.unet, lib()
        sig = (hash(tuple(p.requires_grad for p in u.parameters())), u.concurrent_wgrad, len(u._sides))
        if not self.use_tape or ops.PROFILER is not None or L.recorder is not None:
            self._launch_sequence(bk, after_tail)
            return
        tape = getattr(bk, "tape", None)
        if tape is not None and bk.tape_sig == sig:
            u._after_tail_hook = after_tail
            if self.native_tape:                     # C-side player (tape.NativeTape): the interpreter only runs the live entries
                if getattr(bk, "ntape", None) is None:
                    bk.ntape = NativeTape(tape)
                bk.ntape.play()
                return
            try:
                for fn, args in tape:
                    if fn(*args):
                        raise AozoraError(f"{getattr(fn, '__name__', fn)} failed while re-issuing the launch tape")
            except BaseException:
                disarm_stop_event()
                raise
            return
        if bk.runs < 1:
            self._launch_sequence(bk, after_tail)
            return
        L.recorder = []
        try:
            self._launch_sequence(bk, after_tail)
            bk.tape, bk.tape_sig, bk.ntape = L.recorder, sig, None
            if u.policy.fuse_records:        # fork events ride on the kernel in front of them (tape.fuse_records)
                bk.tape, bk.fused_records = fuse_records(bk.tape, self.stream.cuda_stream if u.policy.fuse_records == 1 else None)
        finally:
            L.recorder = None
        if self.native_tape and bk.tape is not None and bk.ntape is None:
            bk.ntape = NativeTape(bk.tape)           # built here, not at the first replay (one-time ~40 ms of host work)

    def micro_step(self, latents, noise, timesteps, embeds, pooled, time_ids, jitter=None, after_tail=None, defer_join=False,
                   weight_scale=1.0):
        """latents (B,4,h,w) bf16 ; noise (B,4,h,w) fp32 ; timesteps (B,) int ; embeds (B,L,ctx) ;
        pooled (B,P) ; time_ids (B,6) in the compute dtype (bf16 values).  Returns the device fp32
        scalar holding this micro-step's loss (train.py:2767 reads it with .item()).
        weight_scale multiplies the per-sample loss weights (and hence the returned loss and the gradients): data-parallel
        callers whose ranks hold UNEQUAL shares b_r of a ragged global batch GB pass b_r * world / GB, which turns the
        built-in 1/(GA*world) seed into 1/GA * b_r/GB -- each sample then weighs 1/GB as in the single-process run.
        defer_join (needs double_buffer=True): do not wait for this micro-step's parameter-gradient branch; the gradient
        buffer is complete only after the next micro_step called WITHOUT defer_join (the last one of the window)."""
        u = self.unet
        B, C, H, W = latents.shape
        L = embeds.shape[1]
        if defer_join and not self.double_buffer:
            raise AozoraError("defer_join needs TrainStep(double_buffer=True) and the eager executor")
        parity = self._parity if self.double_buffer else 0
        if self.double_buffer:
            self._parity ^= 1
        key = (B, C, H, W, L, parity)
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            u.wait_pool_free(parity)            # deferred weight-gradient work of the previous user of this pool / these buffers
            u._defer_join = bool(defer_join)
            u._pool_parity = parity
            if key not in self._buckets:
                bk = _Bucket(u.device, B, C, H, W, L, u.cfg.cross_attention_dim, u.cfg.pooled_dim)
                bk.tids = torch.empty(B, 6, dtype=F32, device=u.device)
                bk.parity = parity
                self._buckets[key] = bk
            bk = self._buckets[key]
            coef = self._coefficients(bk, timesteps, jitter, time_ids, weight_scale)
            pairs = [(bk.lat, latents.to(BF16)), (bk.noise, noise.to(bk.noise.dtype)), (bk.ctx, embeds.to(BF16)),
                     (bk.pooled, pooled.to(BF16)), (bk.tids, time_ids.float())]
            self._stage(self._host_inputs_in_one_copy(bk, pairs), coef, bk.dev)
            if self.use_graph:
                if after_tail is not None:
                    raise AozoraError("the data-parallel overlap hook needs the eager executor (use_graph=False)")
                u.wait_tail_params()        # a captured forward cannot wait mid-graph: take the all-gather up front
                u._wait_wt_ready()          # ... and whatever the optimizer left running on the parameter-gradient stream (the
                                            # gradient clear, W^T copies): the captured backward's own wait was resolved at capture time
            # W^T copies follow the parameters (no-op unless an optimizer step happened); only the backward reads them
            if self.use_graph:
                u.refresh_transposed()
            else:
                u.refresh_transposed_async()
            st = ctypes.c_void_p(self.stream.cuda_stream)
            if bk.graph is not None:
                lib().call("az_graph_launch", bk.graph, st)
            elif self.use_graph and bk.runs >= 1:
                # second run of this bucket: all pools are allocated -> capture, then replay
                lib().call("az_graph_begin", st)
                try:
                    self._launch_sequence(bk)
                finally:
                    g = ctypes.c_void_p()
                    lib().call("az_graph_end", st, ctypes.byref(g))
                bk.graph = g
                lib().call("az_graph_launch", bk.graph, st)
            else:
                self._eager(bk, after_tail)
            bk.runs += 1
            self.last_pred_nhwc = bk.pred
            self.last_bucket = bk
        torch.cuda.current_stream().wait_stream(self.stream)
        return bk.loss

    def synchronize(self):
        self.stream.synchronize()
