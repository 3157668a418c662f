"""Data-parallel optimizer step: the one exchange step of the path (not in the reference,
which is single-process -- semantics = the reference run with BATCH_SIZE = global batch).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm), launched by
torch.distributed.run.  Per optimizer step, after the last micro-step of the accumulation window:

    reduce-scatter(sum) of the flat bf16 gradient buffer   (loss was pre-scaled by 1/(GA*world) => mean)
    global grad-norm: local sum of squares of the owned shard + one scalar all-reduce; clip in place
    Raven AdamW on the OWNED shard only (each rank holds 1/world of m/v: resident in HBM, or pinned host memory streamed per step)
    all-gather of the bf16 parameters

The flat buffers are cut into three REGIONS (unet.region_bounds(); diffusers parameter order makes them contiguous):
0 = conv_in / embeddings / down_blocks.0-1 (73 M parameters), 1 = down_blocks.2 (757 M), 2 = up_blocks / mid_block /
output head (1.74 G).  Each region is sharded across the ranks on its own, so that the exchange overlaps the step on a
dedicated stream:

    backward:  gradients become final region 2 first (once the backward has passed the mid block), then region 1 (after the
               last down block), then region 0 -> reduce-scatter(2) and reduce-scatter(1) run under the rest of the backward
               (TrainStep.micro_step(after_tail=opt.reduce_tail): the executor calls the hook with the region index);
    forward:   parameters are first read region 0, then 1 (last down block), then 2 (mid block on) -> all-gather(1) and
               all-gather(2) run under the next forward (unet.set_region_params_event / wait_region_params).

Only reduce-scatter(0), the scalar all-reduce and all-gather(0) -- 3 % of the bytes -- stay exposed.

which is element-for-element the arithmetic of "all-reduce + replicated Raven" while moving 1/world
of the optimizer state over each GPU's host link ("Raven at 8 GPUs is host-bound").
With world == 1 the collectives vanish and this is the plain fused clip + Raven step.
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional, Tuple

import torch

from . import ops
from ._lib import lib

_MD = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}


# ---- backend-agnostic flat collectives (nccl = RCCL in production; gloo in tests) -----------------
def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Equal contiguous shards of a flat buffer whose length is a multiple of world*64."""
    if n % (world * 64):
        raise ValueError("flat buffer is not divisible into aligned shards")
    s = n // world
    return rank * s, (rank + 1) * s


def intersect_ranges(ranges: List[Tuple[int, int]], lo: int, hi: int) -> List[Tuple[int, int]]:
    out = []
    for a, b in ranges:
        a2, b2 = max(a, lo), min(b, hi)
        if a2 < b2:
            out.append((a2, b2))
    return out


def _gloo_fence(t: torch.Tensor):
    """gloo path (tests / rehearsal only): make the host wait for the device, so that gloo -- which moves device tensors
    through its own streams -- sees finished inputs and its outputs are complete before anything else is enqueued."""
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def reduce_scatter_flat(dist, flat: torch.Tensor, rank: int, world: int, group=None):
    """In place: afterwards flat[shard(rank)] holds the SUM over ranks of that shard (other shards are
    unspecified).  RCCL: true in-place reduce-scatter; gloo (tests): all-reduce."""
    lo, hi = shard_bounds(flat.numel(), world, rank)
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(flat[lo:hi], flat, op=dist.ReduceOp.SUM, group=group)
    elif flat.dtype == torch.bfloat16:        # gloo (tests only): reduce in fp32, round once
        tmp = flat.float()
        _gloo_fence(flat)                     # gloo's device tensors are not ordered with a side stream: fence by host
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        _gloo_fence(flat)
        flat.copy_(tmp)
    else:
        _gloo_fence(flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        _gloo_fence(flat)


def all_gather_flat(dist, flat: torch.Tensor, rank: int, world: int, group=None):
    """In place: every rank contributes flat[shard(rank)]; afterwards all ranks hold all shards."""
    lo, hi = shard_bounds(flat.numel(), world, rank)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(flat, flat[lo:hi], group=group)
    else:
        parts = [torch.empty_like(flat[lo:hi]) for _ in range(world)]
        mine = flat[lo:hi].clone()
        _gloo_fence(flat)
        dist.all_gather(parts, mine, group=group)
        _gloo_fence(flat)
        for r, part in enumerate(parts):
            a, b = shard_bounds(flat.numel(), world, r)
            flat[a:b].copy_(part)


class ShardedRaven:
    """Raven (raven.py:89-149 arithmetic) over the flat buffers of an AozoraUNet, sharded across ranks."""

    def __init__(self, unet, lr=8e-7, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3,
                 momentum_dtype=torch.bfloat16, clip_grad_norm=1.0, process_group=None, force_local=False,
                 overlap=True, regions: Optional[int] = None, force_exchange=False, state_on_host=False):
        import torch.distributed as dist
        self.unet = unet
        self.state_on_host = bool(state_on_host)
        self.dist = dist if (dist.is_available() and dist.is_initialized() and not force_local) else None
        self.pg = process_group
        self.world = self.dist.get_world_size(self.pg) if self.dist else 1
        self.rank = self.dist.get_rank(self.pg) if self.dist else 0
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, debias_strength=debias_strength,
                                  momentum_dtype=momentum_dtype, lr_scale=1.0)]
        self.clip = clip_grad_norm
        self.mdt = momentum_dtype
        self.step_count = 0
        n = unet.flat_numel                      # multiple of 4096 (unet._layout): equal shards, in-place collectives
        dev = unet.device
        # force_exchange: issue the collectives (and the overlapped three-region schedule) even in a group of ONE rank -- the
        # RCCL calls, the communication stream and the region events then run exactly as with N ranks (tests on a 1-GPU box)
        self.exchange = self.dist is not None and (self.world > 1 or force_exchange)
        if regions is None:
            regions = 3 if overlap else 1
        self.regions = list(unet.region_bounds()) if regions == 3 else [(0, n)]
        if any(b <= a for a, b in self.regions):
            self.regions = [(0, n)]
        self.overlap = overlap and len(self.regions) == 3 and self.exchange
        # One rank, no exchange: the same three regions let the UPDATE of regions 1 and 2 (97 % of the parameters: 7 of the 8.2 ms of
        # optimizer boundary at SDXL-base, plus their W^T copies) run on the parameter-gradient stream under the next forward, which
        # waits for a region's parameters where it first reads them -- the slots the all-gathers take under data parallel.
        self.update_overlap = overlap and len(self.regions) == 3 and not self.exchange
        self._update_inflight = False
        self._reduced = set()
        trainable = unet.trainable_ranges()
        self.own, self.ranges, self.range_off = [], [], []
        own_n = 0
        for (a, b) in self.regions:
            lo, hi = shard_bounds(b - a, self.world, self.rank)
            lo, hi = a + lo, a + hi
            self.own.append((lo, hi))
            rs = intersect_ranges(trainable, lo, hi)                     # frozen parameters are never touched ...
            self.ranges.append(rs)
            offs = []
            for ra, rb in rs:                                            # ... and own no optimizer state: m / v (pinned host copies, device
                offs.append(own_n)                                       # staging) hold the owned TRAINABLE elements back to back, so a freeze
                own_n += rb - ra                                         # mask shrinks the host-link traffic of every step with it
            self.range_off.append(offs)
        self.shard = own_n
        # Raven state.  The reference keeps m, v in PINNED HOST memory and streams them through reusable device buffers every step
        # (raven.py:83-84, 114-117) -- what lets a 2.6 B-parameter model train on a 24 GB card.  On this device the two moments of the
        # whole model are 10.3 GB of 288: by default (state_on_host = False) they simply LIVE in HBM, in the buffers the update kernel
        # reads and writes anyway, and the host sees them when it asks (save_cpu_state / load_cpu_state: the reference's CPU layout,
        # unchanged).  Same kernels on the same values: bit-identical parameters (tests/test_dp_gpu.py).  With state_on_host = True
        # (config RAVEN_STATE_ON_HOST) the reference's residency is kept: the owned shards are streamed through the device copy by
        # async copies on dedicated streams -- H2D prefetched under the last micro-step's compute (m, v do not depend on the
        # gradients), D2H draining under the next iteration: 20.5 GB over the host link per optimizer step at one rank.
        if self.state_on_host:
            self.m_host = torch.zeros(max(own_n, 1), dtype=momentum_dtype).pin_memory()
            self.v_host = torch.zeros(max(own_n, 1), dtype=momentum_dtype).pin_memory()
            self.m_dev = torch.empty(max(own_n, 1), dtype=momentum_dtype, device=dev)
            self.v_dev = torch.empty(max(own_n, 1), dtype=momentum_dtype, device=dev)
        else:
            self.m_host = self.v_host = None
            self.m_dev = torch.zeros(max(own_n, 1), dtype=momentum_dtype, device=dev)
            self.v_dev = torch.zeros(max(own_n, 1), dtype=momentum_dtype, device=dev)
        self._h2d_done = None
        self._d2h_done = None
        self._prefetched = False
        self.hyper_host = torch.zeros(8, dtype=torch.float32).pin_memory()
        self.hyper_dev = torch.zeros(8, dtype=torch.float32, device=dev)
        self.scal = torch.zeros(8, dtype=torch.float32, device=dev)     # [0] sumsq [1] coef [2] norm
        from .streams import host_link_streams
        self.copy_streams = host_link_streams(dev)      # bound to their SDMA engines before any RCCL communicator exists (streams.py)
        self.comm = torch.cuda.Stream(dev)       # collectives of the overlapped (tail) region are issued from here
        from .streams import check as stream_check
        for other, name in ([(getattr(unet, "_main_stream", None), "data-gradient stream")] + [(s_, "weight-gradient stream") for s_ in getattr(unet, "_sides", [])]):
            if other is not None and self.exchange:
                # (a shared hardware queue is harmless for THIS stream: it carries event waits, the W^T copies under the forward and the
                # hand-over to torch's own collective stream; rehearsed at full size with a probed stream instead: 123.0-123.4 vs 122.9 ms)
                stream_check(self.comm, other, f"exchange stream / {name}", if_bad="share a hardware queue (measured harmless for the exchange stream)")
        self._ev = None
        self._timing = None

    # ---- optional event timing of the exchange (bench.py: exposed boundary, per-region collective rates, m/v copies) --------
    def enable_timing(self, on=True):
        self._timing = [] if on else None

    class _Span:
        def __init__(self, opt, name, stream, nbytes):
            self.opt, self.name, self.stream, self.nbytes = opt, name, stream, nbytes

        def __enter__(self):
            if self.opt._timing is not None:
                self.e0 = torch.cuda.Event(enable_timing=True); self.e0.record(self.stream)
            return self

        def __exit__(self, *a):
            if self.opt._timing is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record(self.stream)
                self.opt._timing.append((self.name, self.nbytes, self.e0, e1))
            return False

    def _span(self, name, stream=None, nbytes=0):
        return ShardedRaven._Span(self, name, stream if stream is not None else torch.cuda.current_stream(), nbytes)

    def timing_summary(self):
        """-> {name: dict(calls, ms (mean per call), GBps)} since enable_timing(); synchronises the device."""
        if not self._timing:
            return {}
        torch.cuda.synchronize()
        out = {}
        for name, nbytes, e0, e1 in self._timing:
            d = out.setdefault(name, dict(calls=0, ms=0.0, bytes=0))
            d["calls"] += 1; d["ms"] += e0.elapsed_time(e1); d["bytes"] += nbytes
        for d in out.values():
            d["GBps"] = (d["bytes"] / (d["ms"] * 1e-3) / 1e9) if d["bytes"] and d["ms"] > 0 else None
            d["ms"] /= d["calls"]
            d.pop("bytes")
        self._timing = []
        return out

    # ---------------------------------------------------------------------------------------
    def _hyper(self):
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd, deb = g["lr"], g["betas"], g["eps"], g["weight_decay"], g["debias_strength"]
        t = self.step_count
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        if deb < 1.0:
            bc1, bc2 = 1.0 - (1.0 - bc1) * deb, 1.0 - (1.0 - bc2) * deb
        wdf = 1.0 - lr * wd if wd != 0 else 1.0
        if self._ev is not None:
            self._ev.synchronize()
        self.hyper_host.copy_(torch.tensor([lr, b1, b2, eps, wdf, lr / bc1, math.sqrt(bc2), 0.0], dtype=torch.float32))
        self.hyper_dev.copy_(self.hyper_host, non_blocking=True)
        self._ev = torch.cuda.Event(); self._ev.record()

    def prefetch(self):
        """Start the async H2D of the owned m/v shard (call before the last micro-step(s) of the window)."""
        if self._prefetched:
            return
        if not self.state_on_host:           # the moments are resident in HBM: nothing to fetch
            self._prefetched = True
            return
        h2d = self.copy_streams[0]
        if self._d2h_done is not None:
            h2d.wait_event(self._d2h_done)            # the previous step's write-back has landed in host memory
        with torch.cuda.stream(h2d), self._span("mv_h2d", h2d, 2 * self.m_host.numel() * self.m_host.element_size()):
            self.m_dev.copy_(self.m_host, non_blocking=True)
            self.v_dev.copy_(self.v_host, non_blocking=True)
        self._h2d_done = torch.cuda.Event(); self._h2d_done.record(h2d)
        self._prefetched = True

    # ---- region collectives ---------------------------------------------------------------------
    def _reduce_region(self, i):
        a, b = self.regions[i]
        with self._span(f"reduce_scatter_region{i}", None, (b - a) * 2):
            reduce_scatter_flat(self.dist, self.unet.gflat[a:b], self.rank, self.world, self.pg)

    def _gather_region(self, i):
        a, b = self.regions[i]
        with self._span(f"all_gather_region{i}", None, (b - a) * 2):
            all_gather_flat(self.dist, self.unet.pflat[a:b], self.rank, self.world, self.pg)

    def reduce_tail(self, k=2):
        """Hook for the LAST micro-step of the accumulation window (TrainStep.micro_step(after_tail=...)): the backward
        calls it with k = 2 right after the mid block and with k = 1 right after the last down block, i.e. when every
        gradient of region k has been issued.  Starts that region's reduce-scatter on the communication stream; the
        rest of the backward keeps running."""
        if not self.overlap or k in self._reduced:
            return
        u = self.unet
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main); self.comm.wait_event(ev)
        for side in u._sides:                  # parameter-gradient branch stream(s)
            ev = torch.cuda.Event(); ev.record(side); self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            self._reduce_region(k)
        self._reduced.add(k)

    def step(self) -> torch.Tensor:
        """reduce -> clip -> update owned shards -> gather.  Returns the pre-clip global grad norm (0-d device tensor)."""
        u = self.unet
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        main = torch.cuda.current_stream()
        boundary = self._span("optimizer_boundary_on_main_stream", main)
        boundary.__enter__()
        self._boundary = boundary
        self.step_count += 1
        # a step without a forward in between (tests): the previous step's gathers AND its region-1 / 2 updates (side / comm
        # stream, they read hyper_dev and scal[1]) must have finished before the hyper-parameters and the clip scalars are rewritten
        u.wait_tail_params()
        self._hyper()
        self.prefetch()
        if self.exchange:          # in place: rank r's reduced shard of region i lands in gflat[own[i]]
            if self.overlap:
                self.comm.wait_stream(main)
                with torch.cuda.stream(self.comm):
                    for i in (2, 1, 0):
                        if i not in self._reduced:
                            self._reduce_region(i)
                main.wait_stream(self.comm)
            else:
                for i in range(len(self.regions)):
                    self._reduce_region(i)
        self._reduced = set()
        # grad norm over owned trainable ranges (+ scalar all-reduce)
        first = True
        for rs in self.ranges:
            for a, b in rs:
                ops.sumsq(u.gflat[a:b], self.scal[0:1], not first)
                first = False
        if first:
            self.scal[0:1].zero_()
        if self.exchange:
            self.dist.all_reduce(self.scal[0:1], op=self.dist.ReduceOp.SUM, group=self.pg)
        mx = float(self.clip) if self.clip and self.clip > 0 else float("inf")
        ops.clip_coef(self.scal[0:1], mx, self.scal[1:2], self.scal[2:3])
        esz = 4 if self.mdt == torch.float32 else 2
        L = lib()
        if self._h2d_done is not None:
            main.wait_event(self._h2d_done)

        def update_region(i, stream):
            sp = ctypes.c_void_p(stream.cuda_stream)
            for k, (a, b) in enumerate(self.ranges[i]):
                hoff = self.range_off[i][k]
                L.call("az_adamw_flat", b - a, ctypes.c_void_p(u.pflat.data_ptr() + a * 2), ctypes.c_void_p(u.gflat.data_ptr() + a * 2),
                       ctypes.c_void_p(self.m_dev.data_ptr() + hoff * esz), ctypes.c_void_p(self.v_dev.data_ptr() + hoff * esz),
                       _MD[self.mdt], ctypes.c_void_p(self.hyper_dev.data_ptr()), ctypes.c_void_p(self.scal[1:2].data_ptr()), sp)   # clip coefficient applied in-kernel
        if self.update_overlap:
            # on the parameter-gradient stream: idle during a forward, and the one stream known to run well beside the main one
            # (a first use of the communication stream re-deals the hardware queues: the m / v copy streams then shared one with
            # the compute streams and the window's last micro-steps ran 124 / 161 ms instead of 117 -- measured, streams.py)
            bg = self._bg = u._sides[0]
            update_region(0, main)                     # what the forward reads first stays on the main stream (3 % of the elements)
            bg.wait_stream(main)                       # clip coefficient, hyper-parameters, m / v staging are all ordered before this point
            with torch.cuda.stream(bg):
                later = []
                for i in (1, 2):
                    with self._span(f"update_region{i}", bg, 14 * sum(b - a for a, b in self.ranges[i])):
                        update_region(i, bg)
                    ev = torch.cuda.Event(); ev.record(bg)
                    later.append((i, ev))
                upd = torch.cuda.Event(); upd.record(bg)
                with self._span("wt_refresh", bg, 4 * (self.regions[-1][1] - self.regions[0][0])):
                    for lo, hi in self.regions:        # W^T copies: read by the next BACKWARD only, so they queue behind the updates
                        u._refresh_jobs(lo, hi)
                u._wt_ready = torch.cuda.Event(); u._wt_ready.record(bg)
            for i, ev in later:
                u.set_region_params_event(i, ev)
            u.transposed_refreshed_externally()
            u._grads_busy = upd                        # whoever clears the gradients (unet.zero_grad on any stream) goes behind the update
            self._update_inflight = True
            self._write_back(upd)
            self._boundary.__exit__()
            return self.scal[2]
        if self.exchange and self.overlap:
            # data parallel, overlapped: only region 0's shard is updated on the main stream; the shards of regions 1 / 2 are updated on the
            # exchange stream in front of their all-gathers, under the next forward (which waits per region where it first reads) -- the
            # one-rank schedule above with the all-gathers added.  Rehearsed at pretend N = 8: boundary 1.74 -> 0.6 ms per iteration.
            update_region(0, main)
            upd0 = torch.cuda.Event(); upd0.record(main)       # (also orders the clip coefficient, the hyper-parameters and the m / v staging)
            self.comm.wait_event(upd0)
            with torch.cuda.stream(self.comm):
                self._gather_region(0)
                u._refresh_jobs(*self.regions[0])
                head = torch.cuda.Event(); head.record(self.comm)
                for i in (1, 2):
                    update_region(i, self.comm)
                upd = torch.cuda.Event(); upd.record(self.comm)
                later = []
                for i in (1, 2):
                    self._gather_region(i)
                    u._refresh_jobs(*self.regions[i])
                    ev = torch.cuda.Event(); ev.record(self.comm)
                    later.append((i, ev))
            main.wait_event(head)
            for i, ev in later:
                u.set_region_params_event(i, ev)
            u.mark_params_dirty()
            u.transposed_refreshed_externally()
            self._upd_ev = upd
            u._grads_busy = upd                        # a gradient clear on any stream goes behind the last update
            self._write_back(upd)
            self._boundary.__exit__()
            return self.scal[2]
        for i in range(len(self.ranges)):
            update_region(i, main)
        self._finish_step(main)
        self._boundary.__exit__()
        return self.scal[2]

    def _write_back(self, upd):
        """m / v of the owned shard back to the pinned host copies behind event `upd` (drains under the next iteration); resident
        state: nothing moves."""
        if not self.state_on_host:
            self._prefetched = False
            return
        d2h = self.copy_streams[1]
        d2h.wait_event(upd)
        with torch.cuda.stream(d2h), self._span("mv_d2h", d2h, 2 * self.m_host.numel() * self.m_host.element_size()):
            self.m_host.copy_(self.m_dev, non_blocking=True)
            self.v_host.copy_(self.v_dev, non_blocking=True)
        self._d2h_done = torch.cuda.Event(); self._d2h_done.record(d2h)
        self._prefetched = False

    def _finish_step(self, main):
        """After the owned shards were updated on `main`: m/v write-back to the pinned host copies (drains under the next
        iteration) and the all-gather of the bf16 parameters (regions 1, 2 land under the next forward)."""
        u = self.unet
        upd = torch.cuda.Event(); upd.record(main)
        self._upd_ev = upd
        self._write_back(upd)                                   # drains under the next iteration's compute
        u.mark_params_dirty()
        if self.exchange:      # in place: every rank contributes its updated shards of pflat
            if self.overlap:
                self.comm.wait_event(upd)
                with torch.cuda.stream(self.comm):   # each region: all-gather, then its W^T copies, also on this stream
                    self._gather_region(0)
                    u._refresh_jobs(*self.regions[0])
                    head = torch.cuda.Event(); head.record(self.comm)
                    later = []
                    for i in (1, 2):                 # land under the next forward (before the last down block / the mid block)
                        self._gather_region(i)
                        u._refresh_jobs(*self.regions[i])
                        ev = torch.cuda.Event(); ev.record(self.comm)
                        later.append((i, ev))
                main.wait_event(head)
                for i, ev in later:
                    u.set_region_params_event(i, ev)
                u.transposed_refreshed_externally()
            else:
                for i in range(len(self.regions)):
                    self._gather_region(i)

    def zero_grad(self, set_to_none=True):
        if self._update_inflight:
            # the update of regions 1 / 2 is still reading the gradients on the background stream: clear them there, behind it;
            # the next backward waits for that stream's W^T event before its first launch (unet.backward_nhwc), the forward writes
            # no gradient
            u = self.unet
            with torch.cuda.stream(self._bg):          # (in order behind the update on that stream: no wait needed)
                u._grads_busy = None
                u.zero_grad(set_to_none)
                u._wt_ready = torch.cuda.Event(); u._wt_ready.record(self._bg)
            self._update_inflight = False
            return
        upd, self._upd_ev = getattr(self, "_upd_ev", None), None
        if self.exchange and self.overlap and upd is not None and self.unet.concurrent_wgrad:
            # data parallel: the 5-GB clear leaves the main stream (1.5 ms in front of every forward -- at eight ranks in front of every
            # micro-step).  It runs on the parameter-gradient stream, idle during a forward, behind the update that read the gradients;
            # the next backward waits for that stream's event before its first launch (unet._wait_wt_ready), the forward writes none.
            u = self.unet
            side = u._sides[0]
            side.wait_event(upd)
            with torch.cuda.stream(side):
                u.zero_grad(set_to_none)
                u._wt_ready = torch.cuda.Event(); u._wt_ready.record(side)
            return
        self.unet.zero_grad(set_to_none)

    def synchronize_params(self):
        """Make the current stream wait for an in-flight tail all-gather (before reading parameters outside a forward)."""
        self.unet.wait_tail_params()

    def _host_state(self):
        """(m, v) of the owned shard in host memory, current: the pinned buffers themselves (state_on_host) or a snapshot of the
        resident device state (blocks until every region's update has landed)."""
        self.synchronize_state()
        if self.state_on_host:
            return self.m_host, self.v_host
        return self.m_dev.cpu(), self.v_dev.cpu()

    def _param_views(self, m_host, v_host):
        """(position among the trainable parameters, m view, v view) in the reference's state layout -- one rank, one region:
        the host buffers are then indexed by flat offset, like RavenAdamW's."""
        u = self.unet
        out, i = [], 0
        rs, offs = [], []               # owned trainable ranges of all regions, ascending; pieces that touch (a region cut inside a
        for rr, oo in zip(self.ranges, self.range_off):      # parameter) merge: their host offsets are back to back as well
            for (a, b), o in zip(rr, oo):
                if rs and rs[-1][1] == a and offs[-1] + (rs[-1][1] - rs[-1][0]) == o:
                    rs[-1] = (rs[-1][0], b)
                else:
                    rs.append((a, b)); offs.append(o)
        k = 0
        for name, p in u.named_parameters():
            if not p.requires_grad:
                continue
            off, st, shape = u._slots[name]
            n = math.prod(st)
            while k < len(rs) and rs[k][1] <= off:       # parameters and trainable ranges both ascend in flat offset
                k += 1
            if k == len(rs) or not (rs[k][0] <= off and off + n <= rs[k][1]):
                raise ValueError("the freeze mask changed after the optimizer was created")
            ho = offs[k] + (off - rs[k][0])
            m, v = m_host[ho:ho + n].view(st), v_host[ho:ho + n].view(st)
            if len(st) == 4:
                m, v = m.permute(0, 3, 1, 2)[:, :shape[1]], v.permute(0, 3, 1, 2)[:, :shape[1]]
            out.append((i, m, v))
            i += 1
        return out

    def save_cpu_state(self):
        """One rank (the single-GPU trainer): the REFERENCE's layout {i: {step, exp_avg_cpu, exp_avg_sq_cpu}, "_momentum_dtype"}
        indexed by position among the requires_grad parameters (raven.py:156-169), so either trainer resumes the other's file.
        Several ranks: this rank's shard -- m / v of the owned ranges (copied to host memory) plus the layout they belong to; a resume
        needs the same world size and freeze mask."""
        mh, vh = self._host_state()
        if self.world == 1:
            out = {"_momentum_dtype": self.mdt}
            if self.step_count > 0:
                for i, m, v in self._param_views(mh, vh):
                    out[i] = {"step": self.step_count, "exp_avg_cpu": m.clone(), "exp_avg_sq_cpu": v.clone()}
            return out
        return {"_sharded": True, "layout_version": 2, "world": self.world, "rank": self.rank, "regions": list(self.regions), "own": list(self.own),
                "ranges": [list(map(tuple, rs)) for rs in self.ranges], "step": self.step_count, "_momentum_dtype": self.mdt, "exp_avg_cpu": mh.clone(), "exp_avg_sq_cpu": vh.clone()}

    def load_cpu_state(self, st):
        if not st.get("_sharded") and self.world == 1:      # the reference's per-parameter layout
            mh, vh = self._host_state()
            step = 0
            for i, m, v in self._param_views(mh, vh):
                if i not in st:
                    continue
                e = st[i]
                em, ev_ = e.get("exp_avg", e.get("exp_avg_cpu")), e.get("exp_avg_sq", e.get("exp_avg_sq_cpu"))
                if em is None or ev_ is None:      # raven.py:175-191 accepts an entry without moments (the state then starts from zero)
                    m.zero_(); v.zero_()
                else:
                    m.copy_(em.to(self.mdt)); v.copy_(ev_.to(self.mdt))
                sv = e.get("step", 0)
                step = max(step, int(sv.item()) if torch.is_tensor(sv) else int(sv))
            self.step_count = step
            self._host_to_device(mh, vh)
            return
        if st.get("_sharded") and "ranges" not in st and st.get("layout_version", 1) < 2:
            # round-2 files held m / v at owned-range offsets (frozen elements included); since round 3 the owned TRAINABLE elements are packed
            if st["exp_avg_cpu"].numel() != self.m_dev.numel():
                raise ValueError("sharded optimizer state was written in the pre-'layout_version 2' format (m / v at owned-range offsets, frozen "
                                 "elements included); with a freeze mask it cannot be loaded -- resume from the model file and restart the optimizer state")
        if (not st.get("_sharded") or st["world"] != self.world or st["rank"] != self.rank or list(st["own"]) != list(self.own)
                or [list(map(tuple, rs)) for rs in st.get("ranges", self.ranges)] != [list(map(tuple, rs)) for rs in self.ranges]
                or st["exp_avg_cpu"].numel() != self.m_dev.numel()):
            raise ValueError("sharded optimizer state does not match this run's world size / rank / region layout / freeze mask")
        self.synchronize_state()
        if self.state_on_host:
            self.m_host.copy_(st["exp_avg_cpu"].to(self.mdt))
            self.v_host.copy_(st["exp_avg_sq_cpu"].to(self.mdt))
            self._prefetched = False
        else:
            self._host_to_device(st["exp_avg_cpu"].to(self.mdt), st["exp_avg_sq_cpu"].to(self.mdt))
        self.step_count = int(st["step"])

    def _host_to_device(self, mh, vh):
        """After a load: the pinned host copies were written in place (the next prefetch streams them in); resident state is copied
        to the device here, once."""
        if not self.state_on_host:
            self.m_dev.copy_(mh); self.v_dev.copy_(vh)
            torch.cuda.synchronize(self.unet.device)
        self._prefetched = False

    def synchronize_state(self):
        """Block until the state a checkpoint reads is current (raven.py:156-169 save_cpu_state): the host copies' write-back, or --
        resident state -- every stream that updates m / v on the device."""
        if not self.state_on_host:
            torch.cuda.synchronize(self.unet.device)
            return
        if self._d2h_done is not None:
            self._d2h_done.synchronize()


class ShardedTitan(ShardedRaven):
    """Titan (titan.py:119-131 gradient accumulation in fp32 outside the autograd buffers, 162-184 fp32 clip, 230-296 step
    on fp32 gradients) under data parallel -- BASELINE.json configs[4]: freeze keywords + Titan on 8 GPUs.

    What Titan changes against Raven is ARITHMETIC, and that is what is kept: every micro-step's bf16 gradient is added
    into an fp32 accumulator (the first micro-step of a window copies), the global norm and the clip act on the fp32 sums,
    and AdamW reads fp32 gradients.  What Titan does for MEMORY (gradients parked in host RAM, for 12 GB cards) has no
    purpose on a 288 GB device, so the fp32 accumulator is a device buffer (10.3 GB for SDXL-base) -- the option SURVEY 8e
    names "keep GA accumulation on device (fp32) and offload once"; single-GPU TitanAdamW keeps the host buffer.

    Per micro-step: accumulate() adds the local bf16 gradients into the accumulator and clears them (the flat-path form of
    the post-accumulate hooks, as TitanAdamW.offload_flat).  Per optimizer step: fp32 reduce-scatter of the accumulator
    (the sum over ranks of the per-rank fp32 sums; RCCL sums fp32 exactly enough that the order of ranks is the only
    difference to a single process), sum of squares of the OWNED shard + scalar all-reduce, clip coefficient applied inside
    the AdamW kernel (fp32 gradients are not rounded), update of the owned shard with its 1/world of m / v (resident in HBM, or pinned host memory with state_on_host),
    all-gather of the bf16 parameters (overlapped with the next forward like ShardedRaven's)."""

    def __init__(self, unet, **kw):
        super().__init__(unet, **kw)
        self.gacc = torch.zeros(unet.flat_numel, dtype=torch.float32, device=unet.device)
        self._acc_started = False
        self._trainable = unet.trainable_ranges()
        self._region_trainable = [intersect_ranges(self._trainable, a, b) for a, b in self.regions]

    def _accumulate_ranges(self, ranges, stream):
        """gacc (+)= float(gflat) over `ranges`, then gflat = 0 there, on `stream` (titan.py:119-131 per parameter)."""
        u = self.unet
        st = ctypes.c_void_p(stream.cuda_stream)
        L = lib()
        for a, b in ranges:
            L.call("az_titan_offload", b - a, ctypes.c_void_p(u.gflat.data_ptr() + a * 2), ctypes.c_void_p(self.gacc.data_ptr() + a * 4),
                   ctypes.c_void_p(0), int(self._acc_started), st)
            L.call("az_memset_async", ctypes.c_void_p(u.gflat.data_ptr() + a * 2), 0, (b - a) * 2, st)

    def accumulate(self):
        """Call after every micro-step (trainer: where the reference's hooks fired during backward).  Regions that reduce_tail
        already moved into the accumulator during this micro-step's backward (and handed to the exchange) are left alone."""
        main = torch.cuda.current_stream()
        for i, rs in enumerate(self._region_trainable):
            if i not in self._reduced:
                self._accumulate_ranges(rs, main)
        self._acc_started = True

    def reduce_tail(self, k=2):
        """Hook for the LAST micro-step of the window (TrainStep.micro_step(after_tail=...)), as ShardedRaven.reduce_tail: when the
        backward has issued every gradient of region k, that region's bf16 gradients join the fp32 accumulator and its fp32
        reduce-scatter starts -- both on the communication stream, under the rest of the backward.  This is where the
        reference's post-accumulate hooks move each gradient DURING the backward (titan.py:93-100, 119-131); without it the
        whole 10.3 GB exchange of cfg5 sat behind the last backward.  Arithmetic unchanged: the same fp32 additions in the
        same order (tests/test_dp_gpu.py: bitwise equal to the serial form)."""
        if not self.overlap or k in self._reduced:
            return
        u = self.unet
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main); self.comm.wait_event(ev)
        for side in u._sides:                  # parameter-gradient branch stream(s)
            ev = torch.cuda.Event(); ev.record(side); self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            self._accumulate_ranges(self._region_trainable[k], self.comm)
            self._reduce_region(k)
        self._reduced.add(k)

    def _reduce_region(self, i):
        a, b = self.regions[i]
        with self._span(f"reduce_scatter_fp32_region{i}", None, (b - a) * 4):
            reduce_scatter_flat(self.dist, self.gacc[a:b], self.rank, self.world, self.pg)

    def clip_grad_norm(self, max_norm):
        """TitanAdamW API (train.py:2773-2774): under data parallel the norm needs the reduced gradients, so it is
        computed inside step(); this records max_norm for it and returns None."""
        self.clip = max_norm
        return None

    def step(self) -> torch.Tensor:
        u = self.unet
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        main = torch.cuda.current_stream()
        if not self._acc_started:
            raise RuntimeError("ShardedTitan.step() without accumulate(): no gradients in the fp32 accumulator")
        self.step_count += 1
        self._hyper()
        self.prefetch()
        u.wait_tail_params()
        if self.exchange:
            if self.overlap:           # regions 2 / 1 may already be on their way (reduce_tail); the rest follows on the same stream
                self.comm.wait_stream(main)
                with torch.cuda.stream(self.comm):
                    for i in (2, 1, 0):
                        if i not in self._reduced:
                            self._reduce_region(i)
                main.wait_stream(self.comm)
            else:
                for i in range(len(self.regions)):
                    self._reduce_region(i)
        self._reduced = set()
        first = True
        for rs in self.ranges:
            for a, b in rs:
                ops.sumsq(self.gacc[a:b], self.scal[0:1], not first)
                first = False
        if first:
            self.scal[0:1].zero_()
        if self.exchange:
            self.dist.all_reduce(self.scal[0:1], op=self.dist.ReduceOp.SUM, group=self.pg)
        mx = float(self.clip) if self.clip and self.clip > 0 else float("inf")
        ops.clip_coef(self.scal[0:1], mx, self.scal[1:2], self.scal[2:3])
        esz = 4 if self.mdt == torch.float32 else 2
        L = lib()
        if self._h2d_done is not None:
            main.wait_event(self._h2d_done)
        for i, rs in enumerate(self.ranges):
            for k, (a, b) in enumerate(rs):
                hoff = self.range_off[i][k]
                L.call("az_adamw_flat_ex", b - a, ctypes.c_void_p(u.pflat.data_ptr() + a * 2), ctypes.c_void_p(self.gacc.data_ptr() + a * 4), 1,
                       ctypes.c_void_p(self.m_dev.data_ptr() + hoff * esz), ctypes.c_void_p(self.v_dev.data_ptr() + hoff * esz),
                       _MD[self.mdt], ctypes.c_void_p(self.hyper_dev.data_ptr()), ctypes.c_void_p(self.scal[1:2].data_ptr()), st)
        self._acc_started = False
        self._finish_step(main)
        return self.scal[2]

    def zero_grad(self, set_to_none=True):
        self._acc_started = False
        self._reduced = set()
        super().zero_grad(set_to_none)
