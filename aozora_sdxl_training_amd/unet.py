"""AozoraUNet -- drop-in for the `unet` object of the reference's train loop (train.py:1437-1469 load,
2660-2667 setup, 2760-2761 call), executed entirely by hand-written HIP kernels (libaozora_hip.so).

Design:
  * parameters live in ONE flat bf16 device buffer, gradients in a second one; each diffusers-named
    torch.nn.Parameter is a view of the flat buffer (conv weights are stored [Cout][kh][kw][Cin] and
    exposed with logical shape (Cout,Cin,kh,kw) -- i.e. channels_last strides); to_q/to_k/to_v are
    adjacent so the projections run as one N=3C (self) / N=2C (cross k,v) GEMM;
  * activations are NHWC == row-major [B*H*W][C] bf16, so ResnetBlock2D and Transformer2DModel share
    a layout and no permutes exist;
  * forward records a tape of backward closures (static topology => the launch sequence is identical
    every step and can be captured into a hipGraph); all activations needed by the backward are kept
    (no gradient checkpointing: 288 GB HBM);
  * every FLOP runs in libaozora_hip.so; torch supplies memory and the stream only.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import os
import torch

from . import ops
from ._lib import lib, AozoraError, ForkEvent
from .unet_spec import UNetConfig, SDXL_BASE, param_table, up_resnet_channels

@dataclass
class ExecPolicy:
    """Where the executor places launches and which fusions it uses: data of ONE AozoraUNet (`unet.policy`), not process state.
    Every setting computes bit-identical losses and gradients (tests/test_model_gpu.py::test_executor_placements_and_fusions_are_
    bitwise_neutral) except `tn_group`, which changes the fp32 summation order of the grouped weight gradients."""
    fork_events: bool = True     # fork / join events without the system-scope fence (_lib.ForkEvent) instead of torch.cuda.Event
    fuse_records: int = 2        # launch tape: a fork event recorded right behind a kernel becomes that kernel's completion signal
                                 #   (1: on the data-gradient stream only, 2: on every stream, 0: off)
    side_batch: int = 1          # parameter-gradient launches per fork at most (block ends flush earlier)
    ln_fused: bool = True        # LayerNorm backward: dx and the gamma / beta partial sums from ONE pass over x / dy
    ln_defer: bool = True        # ... the partial sums finished once per parameter region on the branch (not one launch per LayerNorm)
    hoist: bool = True           # K/V-of-context and time-embedding projections as grouped launches per parameter region
    xkv_side: bool = True        # cross-attention dK / dV on the parameter-gradient branch (nothing on the chain reads them); False: ONE kernel for
                                 #      dQ, dK, dV on the chain (az_attn.hip attn_bwd_cross_kernel).  Re-measured at the end of round 5 on three boxes
                                 #      (False with ATTN_SPLIT_TARGET 192 against True): -0.69, +0.2, -0.02 ms per micro-step -- no decision, kept
    temb_side: bool = True       # time_emb_proj data gradients on the branch behind their producer (no chain wait per ResnetBlock2D)
    geglu_fuse: bool = True      # GEGLU forward inside the epilogue of its projection (ff.net.0.proj)
    cat_inplace: bool = True     # skip concatenations written in place by their producers (K14): no copies
    xkv_group: bool = True       # the weight gradients of the cross-attention K / V projections (hoisted: one shared 77-token context, K = 308 rows,
                                 #      5 k-tiles: never split) are parked per parameter region and go out as ONE grouped launch
                                 #      (az_gemm_tn_grouped_bf16) instead of one 18-us launch per transformer block; same tiles, same bits
    tn_group: int = 0            # > 0: linear weight gradients parked until they add up to this many 128x128 tiles, then ONE grouped
                                 #      launch over whole k-ranges (az_gemm_tn_grouped_bf16: no split-K slabs, no reduce launches).
                                 #      Off by default: same-box A/B 117.4 -> 118.6 ms per micro-step (the 75-us workgroups of an unsplit
                                 #      product hold CU slots the data-gradient chain's next kernel needs)
    host_tape: bool = True       # re-issue a bucket's launch sequence from the recorded launch tape
    native_tape: bool = True     # ... played by the C side (az_tape_play); False: replayed from Python
    side_streams: int = 1        # parameter-gradient branch streams (more than one measured slower)
    side_priority: int = 0
    main_priority: int = 0


BF16 = torch.bfloat16
F32 = torch.float32
ALIGN = 64  # elements


class Act:
    """An activation [rows][C] (2-D view, unit inner stride) and its gradient buffer."""
    __slots__ = ("t", "g", "need_grad", "ready", "g_alias")

    def __init__(self, t: torch.Tensor, need_grad: bool = True):
        self.t = t
        self.g: Optional[torch.Tensor] = None
        self.need_grad = need_grad
        self.ready = None     # event: g is being WRITTEN on the side stream (wait before READING g on main)
        self.g_alias = False  # g IS some layer's dY (given by _give_grad): the parameter-gradient branch may read it at any later
                              # time, so it is never written again -- the next contribution goes to a fresh buffer (g_new = g + ...)


class _Pool:
    """Static buffer pool: the n-th allocation of a step always returns the same tensor, so the
    launch sequence (including addresses) repeats exactly and can be replayed from a hipGraph."""

    def __init__(self, device):
        self.device = device
        self.bufs: List[torch.Tensor] = []
        self.cursor = 0

    def reset(self):
        self.cursor = 0

    def get(self, shape, dtype=BF16):
        if self.cursor < len(self.bufs):
            t = self.bufs[self.cursor]
            if tuple(t.shape) != tuple(shape) or t.dtype != dtype:
                raise AozoraError("activation pool replay mismatch (topology changed between steps)")
        else:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.bufs.append(t)
        self.cursor += 1
        return t

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self.bufs)


class _UNetCall(torch.autograd.Function):
    """Autograd bridge for the reference's own loop body: `pred = unet(...).sample; loss = f(pred);
    (loss / GA).backward()` (train.py:2760-2765).  forward = HIP forward, backward = HIP backward seeded
    with d(loss)/d(pred); parameter gradients are ACCUMULATED into the flat gradient buffer and exposed
    as `.grad` views (post-accumulate hooks do not fire: Titan users call optimizer.offload_flat(unet))."""

    @staticmethod
    def forward(ctx, anchor, unet, sample, t_f32, ehs, pooled, tids_f32):
        B, C, H, W = sample.shape
        unet.begin_step((B, H, W, ehs.shape[1], "call"))
        x8 = unet._pool.get((B, H, W, 8), BF16)
        ops.nchw_to_nhwc_pad(sample.contiguous(), x8, C)
        pred = unet.forward_nhwc(x8, t_f32, ehs, pooled, tids_f32)
        out = torch.empty(B, unet.cfg.out_channels, H, W, dtype=BF16, device=sample.device)
        ops.nhwc_to_nchw(pred.t.view(B, H, W, unet.cfg.out_channels), out, unet.cfg.out_channels)
        ctx.unet, ctx.pred, ctx.geom = unet, pred, (B, H, W)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        unet, (B, H, W) = ctx.unet, ctx.geom
        d8 = unet._pool.get((B, H, W, 8), BF16)
        g = grad_out.contiguous()
        if g.dtype not in (BF16, F32):
            g = g.float()
        ops.nchw_to_nhwc_pad(g, d8, unet.cfg.out_channels)
        unet.backward_nhwc(ctx.pred, d8)
        unet.expose_grads()
        return None, None, None, None, None, None, None


class AozoraUNet:
    def __init__(self, cfg: UNetConfig = SDXL_BASE, device="cuda:0", policy: Optional[ExecPolicy] = None):
        if not torch.cuda.is_available():
            raise AozoraError("AozoraUNet needs a HIP device; there is no CPU fallback")
        self.cfg = cfg
        self.policy = policy if policy is not None else ExecPolicy()
        self.config = SimpleNamespace(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                                      sample_size=128, cross_attention_dim=cfg.cross_attention_dim)
        self.device = torch.device(device)
        # library context of this UNet (az_init): its own copy of the execution-option table, made current on the issuing thread
        # at the start of every step -- two UNets (devices) in one process do not share option state
        # (context calls go to the bound functions directly, never through lib().call: they must not land on a launch tape that
        # happens to be recording -- a garbage-collected UNet's az_destroy replayed every step would be a double free)
        self._ctx = ctypes.c_void_p()
        if lib()._fn["az_init"](self.device.index if self.device.index is not None else torch.cuda.current_device(), ctypes.byref(self._ctx)):
            raise AozoraError("az_init failed")
        lib()._fn["az_make_current"](self._ctx)
        self.training = True
        self._table = param_table(cfg)
        self._layout()
        self._pools: Dict[tuple, _Pool] = {}
        self._pool: Optional[_Pool] = None
        self._tape: List = []
        self.conv_in = True   # train.py:2694 probes hasattr(unet, 'conv_in')
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        # backward concurrency: each layer's wgrad (+ bias grad) runs on a forked stream beside its dgrad
        self.concurrent_wgrad = True
        # parameter-gradient branch streams (round-robin): independent weight-gradient products of moderate size run
        # side by side instead of each being split-K'ed to fill the chip on its own (side_priority -1, a high-priority branch: +1 ms)
        self._sides = [torch.cuda.Stream(device=self.device, priority=self.policy.side_priority)
                       for _ in range(max(1, self.policy.side_streams))]
        self._main_stream = None       # set by TrainStep: the exchange / copy streams are chosen to run beside it too
        self._side_rr = 0
        self._side_q: List = []        # queued parameter-gradient launches (see _side_defer / _flush_side)
        self._side_done = None         # completion event of the last batch issued to the branch
        self._ln_jobs: List = []       # parked LayerNorm partial sums (part, dgamma, dbeta, nblk, C)
        self._tn_jobs: List = []       # parked linear weight gradients of the current block (dY, X, dW, bias gradient)
        self._tn_tables = {}
        self._xkv_jobs: List = []      # parked weight gradients of the hoisted context projections of the current parameter region
        self._ln_tables = {}
        self._hoisted: Dict[str, Act] = {}      # forward outputs computed ahead by grouped launches (name -> Act)
        self._group_tables = {}
        self._side = self._sides[0]
        # data-parallel overlap / scheduling state (see region_bounds, wait_region_params, _end_join)
        self._regions = None
        self._region_events = {}
        self._wt_region_pending = set()
        self._wt_ready = None
        self._deferred = {}
        self._defer_join = False
        self._pool_parity = 0
        self._after_tail_hook = None
        self._tape_mark = self._tape_mark1 = 0
        self._events: list = []      # pool of fork / join events (ForkEvent, or torch.cuda.Event with policy.fork_events off)
        self._ev_cursor = 0
        for slot in (2, 1, 0):
            ops.set_workspace_slot(slot); ops.workspace(self.device)

    def __call__(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, **_ignored):
        """diffusers call signature used at train.py:2760-2761; returns an object with `.sample` (B,C,H,W) bf16."""
        if added_cond_kwargs is None or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
            raise ValueError("SDXL needs added_cond_kwargs={'text_embeds', 'time_ids'}")
        B = sample.shape[0]
        t = torch.as_tensor(timestep, device=self.device).reshape(-1).float()
        if t.numel() == 1:
            t = t.expand(B)
        pooled = added_cond_kwargs["text_embeds"].to(device=self.device, dtype=BF16).contiguous()
        tids = added_cond_kwargs["time_ids"].to(self.device).float().contiguous()
        ehs = encoder_hidden_states.to(device=self.device, dtype=BF16).contiguous()
        out = _UNetCall.apply(self._anchor, self, sample.to(device=self.device, dtype=BF16), t.contiguous(), ehs, pooled, tids)
        return SimpleNamespace(sample=out)

    # ------------------------------------------------------------------ parameters ---------------
    def _storage_shape(self, name, shape):
        if len(shape) == 4:
            O, I, kh, kw = shape
            if name == "conv_in.weight":
                I = ((I + 7) // 8) * 8
            return (O, kh, kw, I)
        return tuple(shape)

    def _layout(self):
        off = 0
        self._slots: Dict[str, Tuple[int, tuple, tuple]] = {}
        for name, shape in self._table:
            st = self._storage_shape(name, shape)
            n = math.prod(st)
            self._slots[name] = (off, st, tuple(shape))
            off += ((n + ALIGN - 1) // ALIGN) * ALIGN
        off = ((off + 4095) // 4096) * 4096     # equal shards for 1/2/4/8-way in-place reduce-scatter / all-gather
        self.flat_numel = off
        self.pflat = torch.zeros(off, dtype=BF16, device=self.device)
        self.gflat = torch.zeros(off, dtype=BF16, device=self.device)
        self._w: Dict[str, torch.Tensor] = {}      # storage-shaped views (what kernels read)
        self._gw: Dict[str, torch.Tensor] = {}
        self._params: Dict[str, torch.nn.Parameter] = {}
        self._gviews: Dict[str, torch.Tensor] = {}  # logical-shaped grad views (what .grad exposes)
        for name, (o, st, shape) in self._slots.items():
            n = math.prod(st)
            w = self.pflat[o:o + n].view(st)
            g = self.gflat[o:o + n].view(st)
            self._w[name], self._gw[name] = w, g
            if len(st) == 4:
                lv = w.permute(0, 3, 1, 2)[:, :shape[1]]
                gv = g.permute(0, 3, 1, 2)[:, :shape[1]]
            else:
                lv, gv = w, g
            prm = torch.nn.Parameter(lv, requires_grad=True)
            prm._az_owner, prm._az_name = self, name
            self._params[name] = prm
            self._gviews[name] = gv

        # transposed copies W^T of every 2-D (linear / 1x1-conv) weight, same offsets in a parallel flat buffer: the
        # data-gradient product dX = dY . W then runs in the k-contiguous (NT) form.  Fused projections
        # (to_q|to_k|to_v, to_k|to_v) are transposed as one [sum(out)][in] matrix.
        self.wtflat = torch.zeros(self.flat_numel, dtype=BF16, device=self.device)
        self._wt_tables = {}
        self._wt_jobs: List[Tuple[int, int, int]] = []       # (offset, rows N, cols K) of the stored [N][K] matrix
        names = [n for n, _ in self._table]
        skip = set()
        for name in names:
            o, st, shape = self._slots[name]
            if not name.endswith(".weight") or name in skip:
                continue
            if len(st) == 2:
                rows, cols = st
            elif len(st) == 4 and st[1] == 1 and st[2] == 1:
                rows, cols = st[0], st[3]
            else:
                continue
            if name.endswith("attn1.to_q.weight"):
                rows *= 3
                skip.update({name.replace("to_q", "to_k"), name.replace("to_q", "to_v")})
            elif name.endswith("attn2.to_k.weight"):
                rows *= 2
                skip.add(name.replace("to_k", "to_v"))
            self._wt_jobs.append((o, rows, cols))
        # 3x3 conv weights [Cout][9][Cin] -> W'[Cin][9][Cout] (nine strided transposes) for the NT-form conv dgrad
        self._wt_conv_jobs: List[Tuple[int, int, int]] = []   # (offset, Cout, Cin)
        for name in names:
            o, st, shape = self._slots[name]
            if name.endswith(".weight") and len(st) == 4 and st[1] == 3 and st[0] % 8 == 0 and name != "conv_in.weight":
                self._wt_conv_jobs.append((o, st[0], st[3]))
        self._wt_dirty = True
        self._wt_version = -1

    def mark_params_dirty(self):
        """Call after parameters were modified behind torch's back (the HIP optimizers do)."""
        self._wt_dirty = True

    def tail_offset(self) -> int:
        """Flat offset (multiple of 4096) from which every parameter belongs to up_blocks / mid_block / the output head:
        their gradients are complete once the backward has passed the mid block, and the forward does not read them
        before the mid block -- the hook points of the data-parallel overlap (dist.ShardedRaven)."""
        return self.region_bounds()[-1][0]

    def region_bounds(self):
        """Three contiguous ranges of the flat buffers (cuts are multiples of 4096) ordered as the forward first needs them
        and as the backward finishes them last-to-first -- the units of the data-parallel overlap (dist.ShardedRaven):
          0: conv_in, embeddings, down_blocks.0 .. n-2      read first by the forward, gradients complete last
          1: the last (widest) down block                     (SDXL: 757 M of the 830 M "head" parameters)
          2: up_blocks, mid_block, output head                read from the mid block on, gradients complete first"""
        if self._regions is None:
            last = len(self.cfg.block_out_channels) - 1
            up4 = lambda o: ((o + 4095) // 4096) * 4096
            c1 = up4(min(o for n, (o, _, _) in self._slots.items() if n.startswith(f"down_blocks.{last}.")))
            c2 = up4(min(o for n, (o, _, _) in self._slots.items() if n.startswith("up_blocks.")))
            self._regions = [(0, c1), (c1, c2), (c2, self.flat_numel)]
        return self._regions

    def _refresh_table(self, lo, hi):
        """Job table (device int64 [njobs][8], see az_transpose_multi_bf16) of the W^T copies whose LAST element lies in [lo, hi),
        built once per (lo, hi): the flat buffers never move."""
        key = (lo, hi)
        tab = self._wt_tables.get(key)
        if tab is not None:
            return tab
        rows_, tiles, loose = [], 0, []
        p0, t0 = self.pflat.data_ptr(), self.wtflat.data_ptr()

        def add(src_off, dst_off, R, C, ld_src, ld_dst):
            nonlocal tiles
            if (R | C | ld_src | ld_dst) & 7 or (p0 + 2 * src_off) & 15 or (t0 + 2 * dst_off) & 15:
                loose.append((src_off, dst_off, R, C, ld_src, ld_dst))      # odd shapes (test configurations): one launch each
                return
            tc = (C + 63) // 64
            rows_.append([p0 + 2 * src_off, t0 + 2 * dst_off, R, C, ld_src, ld_dst, tiles, tc])
            tiles += tc * ((R + 63) // 64)
        for o, rows, cols in self._wt_jobs:                 # [N][K] -> [K][N]
            n = rows * cols
            if lo <= o + n - 1 < hi:
                add(o, o, rows, cols, cols, rows)
        for o, co, ci in self._wt_conv_jobs:                # [Cout][9][Cin] -> [Cin][9][Cout], one job per tap
            n = co * 9 * ci
            if lo <= o + n - 1 < hi:
                for tap in range(9):
                    add(o + tap * ci, o + tap * co, co, ci, 9 * ci, 9 * co)
        tab = (torch.tensor(rows_, dtype=torch.int64, device=self.device) if rows_ else None, len(rows_), tiles, loose)
        self._wt_tables[key] = tab
        return tab

    def _refresh_jobs(self, lo, hi):
        """W^T copies of the weights whose LAST element lies in [lo, hi), in ONE launch.  Region cuts are 4096-aligned, not
        parameter-aligned: a weight straddling a cut is complete only once the later region has been all-gathered, so it
        belongs to that region's refresh."""
        tab, njobs, tiles, loose = self._refresh_table(lo, hi)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if njobs:
            lib().call("az_transpose_multi_bf16", ctypes.c_void_p(tab.data_ptr()), njobs, tiles, st)
        p0, t0 = self.pflat.data_ptr(), self.wtflat.data_ptr()
        for so, do, R, C, ls, ld in loose:
            lib().call("az_transpose_bf16", R, C, ctypes.c_void_p(p0 + 2 * so), ls, ctypes.c_void_p(t0 + 2 * do), ld, st)

    def refresh_transposed(self):
        """Refresh the W^T copies if the parameters changed.  Regions whose all-gather is still in flight
        (set_region_params_event) are skipped; wait_region_params() refreshes them when they have landed."""
        if not self._wt_dirty and self._wt_version == self.pflat._version:
            return
        ev, pend = self._region_events, self._wt_region_pending
        for k, (lo, hi) in enumerate(self.region_bounds()):
            if k in ev:
                pend.add(k)
            else:
                self._refresh_jobs(lo, hi)
        self._wt_dirty = False
        self._wt_version = self.pflat._version

    def refresh_transposed_async(self):
        """The W^T copies are only read by the BACKWARD (data-gradient products): refresh them on the parameter-gradient
        stream while the forward runs; backward_nhwc waits for the event."""
        if not self._wt_dirty and self._wt_version == self.pflat._version:
            return
        main, side = torch.cuda.current_stream(), self._sides[0]
        ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        with torch.cuda.stream(side):
            self.refresh_transposed()
        self._wt_ready = torch.cuda.Event(); self._wt_ready.record(side)

    def _wait_wt_ready(self):
        ev = self._wt_ready
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._wt_ready = None

    def transposed_refreshed_externally(self):
        """dist.ShardedRaven refreshed every region's W^T copies itself (on its communication stream, ordered behind the
        all-gathers and ahead of the region events): nothing left to do at the next micro-step."""
        self._wt_dirty = False
        self._wt_version = self.pflat._version
        self._wt_region_pending.clear()

    def set_region_params_event(self, k, ev):
        """dist.ShardedRaven: the parameters of region k are being all-gathered on another stream; `ev` fires when they
        have landed.  Nothing may read them before wait_region_params(k)."""
        self._region_events[k] = ev

    def set_tail_params_event(self, ev):
        self.set_region_params_event(2, ev)

    def wait_region_params(self, k):
        ev = self._region_events.pop(k, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        pend = self._wt_region_pending
        if k in pend:
            lo, hi = self.region_bounds()[k]
            self._refresh_jobs(lo, hi)
            pend.discard(k)

    def _wait_region1(self):
        self.wait_region_params(1)

    def _wait_region2(self):
        self.wait_region_params(2)

    def _run_region_hook1(self):
        if self._after_tail_hook is not None:
            self._after_tail_hook(1)

    # ---- host launch tape support: stream / event operations of the launch sequence go through these two so that a
    # recording (lib().recorder) captures them next to the ABI launches
    def _ev_record(self, ev, stream):
        ev.record(stream)
        rec = lib().recorder
        if rec is not None:
            rec.append((ev.record, (stream,)))

    def _st_wait(self, stream, ev):
        if isinstance(ev, ForkEvent):
            ev.wait_on(stream)
            rec = lib().recorder
            if rec is not None:
                rec.append((ev.wait_on, (stream,)))
            return
        stream.wait_event(ev)
        rec = lib().recorder
        if rec is not None:
            rec.append((stream.wait_event, (ev,)))

    def _live(self, fn):
        """Run fn now with recording suspended and put fn itself on the tape (state-dependent host logic)."""
        L = lib()
        rec, L.recorder = L.recorder, None
        try:
            fn()
        finally:
            L.recorder = rec
        if rec is not None:
            rec.append((fn, ()))

    def _end_join(self):
        """End of a backward.  Default: the data chain waits for the parameter-gradient stream(s).  With
        `_defer_join` (TrainStep(double_buffer=True), a non-final micro-step of an accumulation window) the branch is NOT
        joined: its remaining weight-gradient products keep running under the next micro-step's forward, whose launches
        write into the other activation pool; only an event is left for the next user of THIS pool."""
        main = torch.cuda.current_stream()
        evs = []
        for sd in self._sides:
            ev = torch.cuda.Event(); ev.record(sd); evs.append(ev)
        if self._defer_join:
            self._deferred[self._pool_parity] = evs
        else:
            for ev in evs:
                main.wait_event(ev)
            self._deferred.clear()     # the in-order side stream(s) are fully drained now

    def wait_pool_free(self, parity):
        """Before a micro-step writes into activation pool `parity`: the deferred weight-gradient work that still reads it."""
        for ev in self._deferred.pop(parity, []):
            torch.cuda.current_stream().wait_event(ev)

    def has_deferred(self):
        return bool(self._deferred)

    def _set_forward_exclusive(self):
        # forward: the data chain has the CUs (and their LDS) to itself (kept so even with deferred weight-gradient work
        # around: measured 0.851 vs 0.835 it/s)
        lib().call("az_gemm_set_exclusive", 1)

    def _run_after_tail(self):
        if self._after_tail_hook is not None:
            self._after_tail_hook(2)

    def wait_tail_params(self):
        """Wait for every in-flight parameter all-gather (name kept from the two-region form)."""
        for k in (0, 1, 2):
            self.wait_region_params(k)

    def _wt(self, W: torch.Tensor) -> torch.Tensor:
        """transposed copy [K][N] of a stored [N][K] weight view of pflat."""
        o = (W.data_ptr() - self.pflat.data_ptr()) // 2
        N, K = W.shape
        return self.wtflat[o:o + N * K].view(K, N)

    def named_parameters(self):
        for name, _ in self._table:
            yield name, self._params[name]

    def parameters(self):
        for name, _ in self._table:
            yield self._params[name]

    def state_dict(self):
        self.wait_tail_params()
        return {name: self._params[name].detach() for name, _ in self._table}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict=True):
        with torch.no_grad():
            for name, _ in self._table:
                if name not in sd:
                    if strict:
                        raise KeyError(name)
                    continue
                self._params[name].copy_(sd[name].to(device=self.device, dtype=BF16))
        self._wt_dirty = True
        return self

    def trainable_ranges(self) -> List[Tuple[int, int]]:
        """Merged [start, end) element ranges of the flat buffers that belong to trainable params."""
        out: List[List[int]] = []
        for name, (o, st, _) in self._slots.items():
            if not self._params[name].requires_grad:
                continue
            n = ((math.prod(st) + ALIGN - 1) // ALIGN) * ALIGN
            if out and out[-1][1] == o:
                out[-1][1] = o + n
            else:
                out.append([o, o + n])
        return [(a, b) for a, b in out]

    def expose_grads(self):
        """Make .grad of every trainable Parameter a view of the flat gradient buffer."""
        for name, p in self._params.items():
            p.grad = self._gviews[name] if p.requires_grad else None

    def zero_grad(self, set_to_none=True):
        ev, self._grads_busy = getattr(self, "_grads_busy", None), None
        if ev is not None:            # an optimizer update that runs on another stream under the next forward (dist.ShardedRaven, one rank) is
            torch.cuda.current_stream().wait_event(ev)      # still reading the gradients: clear them behind it
        self.gflat.zero_()
        if set_to_none:
            for p in self._params.values():
                p.grad = None

    # reference seams that are no-ops here ---------------------------------------------------------
    def enable_gradient_checkpointing(self):   # train.py:2660 -- not needed with 288 GB HBM
        return None

    def enable_xformers_memory_efficient_attention(self):
        return None

    def set_attn_processor(self, *_a, **_k):   # train.py:204-228 -- attention is az_attn_fwd/bwd
        return None

    def to(self, device=None, *_a, **_k):
        if device is not None and torch.device(device).type != "cuda":
            raise AozoraError("AozoraUNet lives on a HIP device only")
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def requires_grad_(self, flag=True):
        for p in self._params.values():
            p.requires_grad = flag
        return self

    # ------------------------------------------------------------------ fork / join ---------------
    def _event(self):
        if self._ev_cursor == len(self._events):
            self._events.append(ForkEvent() if self.policy.fork_events else torch.cuda.Event())
        ev = self._events[self._ev_cursor]
        self._ev_cursor += 1
        return ev

    class _Side:
        """with unet._fork(): ... launches go to the side stream (own workspace) and are joined on exit."""

        def __init__(self, u):
            self.u = u

        def __enter__(self):
            u = self.u
            self.main = torch.cuda.current_stream()
            if not u.concurrent_wgrad:
                return self
            k = u._side_rr % len(u._sides)
            u._side_rr += 1
            self.stream = u._sides[k]
            ev = u._event(); u._ev_record(ev, self.main); u._st_wait(self.stream, ev)
            u._side_used = True
            self.ctx = torch.cuda.stream(self.stream); self.ctx.__enter__()
            ops.set_workspace_slot(1 + k)
            return self

        def __exit__(self, *a):
            u = self.u
            if not u.concurrent_wgrad:
                return False
            ops.set_workspace_slot(0)
            self.ctx.__exit__(*a)
            self.done = u._event(); u._ev_record(self.done, self.stream)
            return False

        done = None

        def join(self):
            if self.u.concurrent_wgrad and self.done is not None:
                self.u._st_wait(self.main, self.done)

    def _fork(self):
        return AozoraUNet._Side(self)

    # ------------------------------------------------------------------ tape helpers --------------
    def _trainable(self, name):
        return self._params[name].requires_grad

    def _new(self, rows, C, need_grad=True, dtype=BF16) -> Act:
        return Act(self._pool.get((rows, C), dtype), need_grad)

    def _wait_ready(self, a: Act):
        if a.ready is not None:
            self._st_wait(torch.cuda.current_stream(), a.ready)
            a.ready = None

    def _gbuf(self, a: Act):
        """-> (dst, add) for writing a contribution to a's gradient: add is None (dst is overwritten), dst itself (accumulate in
        place) or another tensor (dst = add + contribution, out of place).  Out of place exactly when the current gradient
        storage is some layer's dY (g_alias): that layer's weight-gradient product, queued on the parameter-gradient branch,
        reads it at an unspecified later time, and never having to wait for it is what lets the branch lag and its forks be
        batched (one event per block instead of one per layer: the event packets cost the data-gradient stream 5-10 us each,
        ~5 ms per micro-step, tools/trace_gaps.py)."""
        if a.g is None:
            a.g = self._pool.get(tuple(a.t.shape), BF16)
            a.g_alias = False
            return a.g, None
        if a.g_alias:
            old = a.g
            a.g = self._pool.get(tuple(a.t.shape), BF16)
            a.g_alias = False
            return a.g, old
        return a.g, a.g

    def _gbuf_single(self, a: Act, what: str):
        """Gradient buffer of an activation that must have exactly one consumer (no accumulation form in the kernel)."""
        dst, add = self._gbuf(a)
        if add is not None:
            raise AozoraError(what + " must have a single consumer")
        return dst

    def _give_grad(self, a: Act, dy: torch.Tensor, alias=True):
        """a.g += dy, aliasing dy's storage when a has no gradient yet (dy is dead afterwards for the data-gradient chain).
        alias=True: dy is a layer's dY that the parameter-gradient branch still reads (see _gbuf)."""
        if not a.need_grad:
            return
        if a.g is None:
            a.g = dy
            a.g_alias = alias
        elif a.g_alias:
            old = a.g
            a.g = self._pool.get(tuple(a.t.shape), BF16)
            a.g_alias = False
            ops.add_rows(old, dy, a.g)
        else:
            ops.add_rows(a.g, dy, a.g)

    # ---- the parameter-gradient branch: weight / bias gradient launches are queued and issued in batches ----------------------
    def _side_defer(self, fn):
        if not self.concurrent_wgrad:
            fn()
        else:
            self._side_q.append(fn)
            if len(self._side_q) >= self.policy.side_batch:
                self._flush_side()

    def _finish_ln_jobs(self):
        """Queue ONE finish launch for the parked LayerNorm partial sums (layernorm.bwd) on the parameter-gradient branch."""
        jobs, self._ln_jobs = self._ln_jobs, []
        if not jobs:
            return
        key = tuple(p.data_ptr() for p, _, _, _, _ in jobs)
        tab = self._ln_tables.get(key)
        if tab is None:
            rows_, blocks = [], 0
            for part, gw, gb, nblk, C in jobs:
                rows_.append([part.data_ptr(), gw.data_ptr() if gw is not None else 0, gb.data_ptr() if gb is not None else 0, nblk, C, blocks])
                blocks += (C + 31) // 32
            tab = (torch.tensor(rows_, dtype=torch.int64, device=self.device), len(rows_), blocks)
            self._ln_tables[key] = tab
        self._side_defer(lambda: ops.ln_param_finish_multi(tab[0], tab[1], tab[2]))

    def _finish_tn_jobs(self):
        """Queue the parked linear weight gradients on the parameter-gradient branch: as ONE grouped launch (every product over its
        whole k-range on 128x128 tiles -- no fp32 slabs, no reduce launches, the tiles of all products fill the chip together) when
        they add up to a chip-filling grid, else one split-K product each as before.  dY / X stay valid until the step ends: the
        activation pool never re-uses a buffer inside a step and a buffer that is some layer's dY is never written again (_gbuf)."""
        jobs, self._tn_jobs = self._tn_jobs, []
        if not jobs:
            return
        tiles = sum(((dy.shape[1] + 127) // 128) * ((xt.shape[1] + 127) // 128) for dy, xt, _, _ in jobs)
        if tiles < self.policy.tn_group:
            for dy, xt, GW, bg in jobs:
                self._side_defer(lambda dy=dy, xt=xt, GW=GW, bg=bg: ops.gemm(dy, xt, GW, trans_a=True, trans_b=False, accumulate=True, split_k=0, bias_grad=bg))
            return
        key = tuple((dy.data_ptr(), xt.data_ptr(), GW.data_ptr(), bg.data_ptr() if bg is not None else 0) for dy, xt, GW, bg in jobs)
        tab = self._tn_tables.get(key)
        if tab is None:
            tab = ops.tn_group_table(jobs, self.device)
            self._tn_tables[key] = tab
        self._side_defer(lambda: ops.gemm_tn_grouped(*tab))

    def _finish_xkv_jobs(self):
        """The parked K / V projection weight gradients of a parameter region as ONE grouped launch on the branch (policy.xkv_group)."""
        jobs, self._xkv_jobs = self._xkv_jobs, []
        if not jobs:
            return
        if len(jobs) == 1:
            dy, xt, GW, bg = jobs[0]
            self._side_defer(lambda: ops.gemm(dy, xt, GW, trans_a=True, trans_b=False, accumulate=True, split_k=0, bias_grad=bg))
            return
        key = ("xkv",) + tuple((dy.data_ptr(), xt.data_ptr(), GW.data_ptr(), bg.data_ptr() if bg is not None else 0) for dy, xt, GW, bg in jobs)
        tab = self._tn_tables.get(key)
        if tab is None:
            tab = ops.tn_group_table(jobs, self.device)
            self._tn_tables[key] = tab
        self._side_defer(lambda: ops.gemm_tn_grouped(*tab))

    def _block_end(self):
        """Tape entry placed at the START of a block's forward (so it runs AFTER the block's backward): the block's parked
        parameter-gradient work goes out."""
        self._finish_tn_jobs()
        self._flush_side()

    def _flush_side(self):
        """Issue the queued parameter-gradient launches on the side stream behind ONE fork event; -> the completion event of
        everything issued to the branch so far (the branch is one in-order stream per fork target; with an empty queue that is
        the previous flush's event, which _side_defer may just have produced)."""
        if not self._side_q:
            return self._side_done
        q, self._side_q = self._side_q, []
        side = self._fork()
        with side:
            for fn in q:
                fn()
        self._side_done = side.done
        return side.done

    # ------------------------------------------------------------------ layers --------------------
    def _bias_grad(self, dy: torch.Tensor, bname: Optional[str], n_real: int, rows_per_seg=None, seg_out: Optional[Act] = None):
        """bias grad (first n_real columns of the column sums of dy); with `seg_out`, the per-segment
        column sums (segments of rows_per_seg rows = one sample) become seg_out's gradient [nseg][C]
        (the time-embedding add of ResnetBlock2D)."""
        rows, C = dy.shape
        if seg_out is not None and seg_out.need_grad:
            if C != n_real:
                raise AozoraError("segment sums need an unpadded gradient")
            rps = rows_per_seg
        else:
            seg_out, rps = None, rows
        bias = self._gw[bname] if (bname is not None and self._trainable(bname)) else None
        if seg_out is None and bias is None:
            return
        ops.colsum_grad(dy, rps, seg_out.g.view(-1) if seg_out is not None else None, bias, n_real)

    def linear(self, x: Act, wname: str, bname: Optional[str], residual: Optional[Act] = None,
               w_override: Optional[Tuple[torch.Tensor, torch.Tensor, bool]] = None, out: Optional[Act] = None,
               pre: Optional[Act] = None, side_dgrad: bool = False, geglu_out: Optional[Act] = None) -> Act:
        """pre: the forward product was already computed by a grouped launch (_hoist_shared_input_linears); only the backward
        closure is registered here, at the layer's own place on the tape.
        side_dgrad: the output's gradient is PRODUCED on the parameter-gradient branch (time_emb_proj: dY = the per-sample
        channel sums that ride on conv1's weight-gradient pass) and the input's gradient is read by the chain only at the very
        end of the backward pass (emb): the data gradient then runs on the branch too, in order behind its producer, instead
        of making the chain wait for the branch to catch up once per ResnetBlock2D.
        geglu_out: this linear is a GEGLU's projection: the product is issued as az_gemm_geglu_fwd_bf16, whose epilogue also
        writes value * gelu(gate) into geglu_out (geglu(..., pre=geglu_out) then only registers the backward closure)."""
        if w_override is not None:
            W, GW, w_train = w_override
        else:
            W, GW, w_train = self._w[wname], self._gw[wname], self._trainable(wname)
        N = W.shape[0]
        WT = self._wt(W) if x.need_grad else None
        rows = x.t.shape[0]
        if pre is not None:
            y = pre
        else:
            y = out if out is not None else self._new(rows, N)
            if geglu_out is not None:
                ops.gemm_geglu_fwd(x.t, W, self._w[bname] if bname else None, y.t, geglu_out.t)
            else:
                # few-row products (the K/V projections of the 77-token context: 48 tiles) split along k to cover more CUs
                ops.gemm(x.t, W, y.t, trans_b=True, bias=self._w[bname] if bname else None,
                         residual=residual.t if residual is not None else None,
                         split_k=0 if (rows <= 512 and residual is None) else 1)

        def bwd():
            dy = y.g
            if dy is None:
                return
            on_side = side_dgrad and self.policy.temb_side and self.concurrent_wgrad and len(self._sides) == 1
            if on_side:
                y.ready = None      # written on the branch, read on the branch
            else:
                self._wait_ready(y)
            b_train = bname is not None and self._trainable(bname)

            def wgrad():        # parameter gradients run as a free-running branch beside the data-gradient chain
                if w_train:     # the bias gradient (column sums of dY) rides on the same pass over dY
                    ops.gemm(dy, x.t, GW, trans_a=True, trans_b=False, accumulate=True, split_k=0,
                             bias_grad=self._gw[bname] if b_train else None)
                elif b_train:
                    self._bias_grad(dy, bname, N)
            if w_train and pre is not None and w_override is not None and rows <= 512 and self.policy.xkv_group and self.concurrent_wgrad and not on_side:
                # a hoisted context projection (K | V of a cross-attention): parked until the region's backward is through
                self._xkv_jobs.append((dy, x.t, GW, self._gw[bname] if b_train else None))
            elif w_train and self.policy.tn_group > 0 and 256 <= rows <= 8192 and not on_side:
                # parked until the parked products add up to a chip-filling grid (attn2.to_out + attn2.to_q; attn1.to_out + to_q|k|v;
                # the feed-forward ones are that large on their own), then ONE grouped launch, every product over its whole k-range
                self._tn_jobs.append((dy, x.t, GW, self._gw[bname] if b_train else None))
                if sum(((j[0].shape[1] + 127) // 128) * ((j[1].shape[1] + 127) // 128) for j in self._tn_jobs) >= self.policy.tn_group:
                    self._finish_tn_jobs()
            elif w_train or b_train:
                self._side_defer(wgrad)
            if x.need_grad:
                dx, add = self._gbuf(x)
                oop = add is not None and add is not dx

                def dgrad():
                    ops.gemm(dy, WT, dx, trans_b=True, accumulate=add is dx, residual=add if oop else None,      # dX = dY . W  as  dY . (W^T)^T
                             split_k=0 if (rows <= 512 and not oop) else 1)
                if on_side:
                    self._side_defer(dgrad)
                    x.ready = self._flush_side()
                else:
                    dgrad()
            if residual is not None:   # dy becomes the residual's gradient (never written again: _gbuf)
                self._give_grad(residual, dy)
        self._tape.append(bwd)
        return y

    @staticmethod
    def _as4(t: torch.Tensor, B, H, W_):
        """[B*H*W][C] row view (any row stride) -> (B,H,W,C) view."""
        ld = t.stride(0)
        return t.as_strided((B, H, W_, t.shape[1]), (H * W_ * ld, W_ * ld, ld, 1))

    def conv(self, x: Act, geom, wname, bname, stride=1, rowbias: Optional[Act] = None, residual: Optional[Act] = None,
             upsample=False, out: Optional[Act] = None) -> Tuple[Act, tuple]:
        """3x3 conv (pad 1). The gradient handed to this op may carry more (zero) channels than Cout
        (conv_out: dpred is padded 4 -> 8 so that rows stay 16-byte chunks).
        upsample (Upsample2D): `geom` is x's own (half) resolution; forward and weight gradient read x through a
        nearest-2x gather inside the operand fetch, so the upsampled activation never exists; the data gradient comes out at
        the upsampled resolution into a scratch gradient and is folded 2x2 -> 1 by az_upsample2x_bwd."""
        B, Hs, Ws = geom
        H, W_ = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
        Wt = self._w[wname]
        Cout, ks, _, Cin = Wt.shape
        Ho = (H + 2 - 3) // stride + 1
        Wo = (W_ + 2 - 3) // stride + 1
        y = out if out is not None else self._new(B * Ho * Wo, Cout)      # out: a [rows][Cout] view with any row stride (one half of a skip concat)
        if tuple(y.t.shape) != (B * Ho * Wo, Cout):
            raise AozoraError("conv destination shape")
        x4 = self._as4(x.t, B, Hs, Ws)
        ops.conv_fwd(x4, Wt, self._as4(y.t, B, Ho, Wo), stride=stride, bias=self._w[bname],
                     rowbias=rowbias.t if rowbias is not None else None,
                     residual=self._as4(residual.t, B, Ho, Wo) if residual is not None else None, upsample=upsample)

        def bwd():
            dy = y.g
            if dy is None:
                return
            dy4 = self._as4(dy, B, Ho, Wo)
            need_seg = rowbias is not None and rowbias.need_grad
            if need_seg:
                self._gbuf_single(rowbias, "segment-sum target")      # allocate on the main path (pool order is stream-agnostic)
            self._wait_ready(y)

            def wgrad():
                b_train = self._trainable(bname)
                fuse = self._trainable(wname) and (b_train or need_seg) and (not need_seg or ((Ho * Wo) % 64 == 0 and dy.shape[1] == Cout))
                if fuse:        # bias / time-embedding gradients (channel sums of dY) ride on the weight-gradient pass
                    ops.conv_wgrad(dy4, x4, self._gw[wname], stride=stride, cout_real=Cout, accumulate=True, split_k=0,
                                   bias_grad=self._gw[bname] if b_train else None,
                                   seg_grad=rowbias.g.view(-1) if need_seg else None, upsample=upsample)
                else:
                    if rowbias is not None or b_train:
                        self._bias_grad(dy, bname, Cout, rows_per_seg=Ho * Wo, seg_out=rowbias)
                    if self._trainable(wname):
                        ops.conv_wgrad(dy4, x4, self._gw[wname], stride=stride, cout_real=Cout, accumulate=True, split_k=0, upsample=upsample)
            self._side_defer(wgrad)
            if need_seg:       # the time-embedding gradient is produced on the side stream and read by the chain soon: issue now
                rowbias.ready = self._flush_side()
            if x.need_grad:
                dx, add = self._gbuf(x)
                if upsample:            # d(upsampled x) into a scratch buffer, then the 2x2 -> 1 fold into dx
                    if add is not None:
                        raise AozoraError("upsample input must have a single consumer")
                    dxs = dx
                    dx = self._pool.get((B * H * W_, Cin), BF16)
                oop = add is not None and add is not dx
                res4 = self._as4(add, B, H, W_) if oop else None
                if Cout % 8 == 0 and dy.shape[1] == Cout:
                    o_w = (Wt.data_ptr() - self.pflat.data_ptr()) // 2
                    wt = self.wtflat[o_w:o_w + Wt.numel()].view(Cin, 3, 3, Cout)
                    ops.conv_dgrad_wt(dy4, wt, self._as4(dx, B, H, W_), stride=stride, accumulate=add is dx, residual=res4)
                else:
                    ops.conv_dgrad(dy4, Wt, self._as4(dx, B, H, W_), stride=stride, cout_real=Cout, accumulate=add is dx, residual=res4)
                if upsample:
                    ops.upsample2x_bwd(dx.view(B, H, W_, Cin), dxs.view(B, Hs, Ws, Cin))
            if residual is not None:
                self._give_grad(residual, dy)
        self._tape.append(bwd)
        return y, (B, Ho, Wo)

    def groupnorm(self, x: Act, geom, prefix, eps, silu) -> Act:
        B, H, W_ = geom
        C = x.t.shape[1]
        G = self.cfg.norm_groups
        y = self._new(B * H * W_, C)
        stats = self._pool.get((B * G * 2,), F32)
        gam, bet = self._w[prefix + ".weight"], self._w[prefix + ".bias"]
        x3 = x.t.as_strided((B, H * W_, C), (H * W_ * x.t.stride(0), x.t.stride(0), 1))
        ops.groupnorm_fwd(x3, gam, bet, y.t.view(B, H * W_, C), stats, G, eps, silu)

        def bwd():
            dy = y.g
            if dy is None:
                return
            tg, tb = self._trainable(prefix + ".weight"), self._trainable(prefix + ".bias")
            dy3 = dy.as_strided((B, H * W_, C), (H * W_ * dy.stride(0), dy.stride(0), 1))
            dx3, add3 = None, None
            if x.need_grad:
                dx, add = self._gbuf(x)
                dx3 = dx.as_strided((B, H * W_, C), (H * W_ * dx.stride(0), dx.stride(0), 1))
                if add is not None:
                    add3 = add.as_strided((B, H * W_, C), (H * W_ * add.stride(0), add.stride(0), 1))
            ops.groupnorm_bwd(x3, gam, bet, stats, dy3, dx3, self._gw[prefix + ".weight"] if tg else None,
                              self._gw[prefix + ".bias"] if tb else None, G, silu, dx_add=add3)
        self._tape.append(bwd)
        return y

    def layernorm(self, x: Act, prefix) -> Act:
        rows, C = x.t.shape
        y = self._new(rows, C)
        stats = self._pool.get((2 * rows,), F32)
        gam, bet = self._w[prefix + ".weight"], self._w[prefix + ".bias"]
        ops.layernorm_fwd(x.t, gam, bet, y.t, stats, 1e-5)

        def bwd():
            dy = y.g
            if dy is None:
                return
            dx, add = self._gbuf(x)
            gw = self._gw[prefix + ".weight"] if self._trainable(prefix + ".weight") else None
            gb = self._gw[prefix + ".bias"] if self._trainable(prefix + ".bias") else None
            self._wait_ready(y)
            if self.policy.ln_fused:                              # one pass over x / dy for dx and the gamma / beta gradients
                if self.policy.ln_defer:
                    # ... whose per-block partial sums are parked: all LayerNorms of a parameter region are finished by ONE
                    # launch on the parameter-gradient branch (_finish_ln_jobs) instead of one finish launch each on the chain.
                    # (The buffer is taken whether or not the parameters are frozen: the pool's allocation order may not
                    # depend on the freeze mask.)
                    nblk = ops.ln_partial_blocks(rows)
                    part = self._pool.get((nblk * C * 2,), F32)
                if self.policy.ln_defer and (gw is not None or gb is not None):
                    ops.layernorm_bwd_partial(x.t, gam, stats, dy, dx, part, dx_add=add, nblk=nblk)
                    self._ln_jobs.append((part, gw, gb, nblk, C))
                    return
                ops.layernorm_bwd(x.t, gam, stats, dy, dx, gw, gb, dx_add=add)
                return
            if gw is not None or gb is not None:       # gamma / beta gradients leave the data-gradient chain
                self._side_defer(lambda: ops.layernorm_bwd(x.t, gam, stats, dy, None, gw, gb))
            ops.layernorm_bwd(x.t, gam, stats, dy, dx, None, None, dx_add=add)
        self._tape.append(bwd)
        return y

    def silu(self, x: Act) -> Act:
        y = self._new(*x.t.shape)
        ops.silu_fwd(x.t, y.t)

        def bwd():
            if y.g is None:
                return
            self._wait_ready(y)
            dx, add = self._gbuf(x)
            if add is not None and add is not dx:
                raise AozoraError("SiLU input gradient cannot be accumulated out of place")
            ops.silu_bwd(x.t, y.g, dx, accumulate=add is dx)
        self._tape.append(bwd)
        return y

    def _fused_w(self, names: List[str]):
        """storage of adjacent parameters as one [sum(out)][in] matrix (to_q|to_k|to_v)."""
        o0, st0, _ = self._slots[names[0]]
        rows, cols = 0, st0[1]
        off = o0
        for n in names:
            o, st, _ = self._slots[n]
            if o != off or st[1] != cols or math.prod(st) % ALIGN:
                raise AozoraError(f"parameters {names} are not adjacent in the flat buffer")
            rows += st[0]
            off += math.prod(st)
        W = self.pflat[o0:off].view(rows, cols)
        G = self.gflat[o0:off].view(rows, cols)
        return W, G, any(self._trainable(n) for n in names)

    def attention(self, x: Act, B, T, prefix, ctx: Optional[Act], ctx_len, residual: Act) -> Act:
        C = x.t.shape[1]
        heads = C // self.cfg.head_dim
        scale = 1.0 / math.sqrt(self.cfg.head_dim)
        if ctx is None:
            qkv = self.linear(x, None, None, w_override=self._fused_w([prefix + ".to_q.weight", prefix + ".to_k.weight", prefix + ".to_v.weight"]))
            q3 = qkv.t.view(B, T, 3 * C)[..., :C]
            k3 = qkv.t.view(B, T, 3 * C)[..., C:2 * C]
            v3 = qkv.t.view(B, T, 3 * C)[..., 2 * C:]
            Tk, kv_train = T, True
        else:
            q = self.linear(x, prefix + ".to_q.weight", None)
            kv_w = self._fused_w([prefix + ".to_k.weight", prefix + ".to_v.weight"])
            kv_train = kv_w[2]
            kv = self.linear(ctx, None, None, w_override=kv_w, pre=self._hoisted.pop(prefix + ".kv", None))
            q3 = q.t.view(B, T, C)
            k3 = kv.t.view(B, ctx_len, 2 * C)[..., :C]
            v3 = kv.t.view(B, ctx_len, 2 * C)[..., C:]
            Tk = ctx_len
        o = self._new(B * T, C)
        lse = self._pool.get((B * heads * T,), F32)
        ops.attn_fwd(q3, k3, v3, o.t.view(B, T, C), lse, heads, scale)

        def bwd():
            do = o.g
            if do is None:
                return
            delta = self._pool.get((B * heads * T,), F32)
            if ctx is None:
                dqkv = self._gbuf_single(qkv, "attention projections")
                d3 = dqkv.view(B, T, 3 * C)
                dq3, dk3, dv3 = d3[..., :C], d3[..., C:2 * C], d3[..., 2 * C:]
            else:
                dq = self._gbuf_single(q, "attention projections")
                dkv = self._gbuf_single(kv, "attention projections")
                dq3 = dq.view(B, T, C)
                dk3, dv3 = dkv.view(B, ctx_len, 2 * C)[..., :C], dkv.view(B, ctx_len, 2 * C)[..., C:]
            o3, do3 = o.t.view(B, T, C), do.view(B, T, C)
            if ctx is not None and not ctx.need_grad and self.policy.xkv_side:
                # dK / dV of a cross-attention feed only the to_k | to_v weight gradients (the text context has no gradient):
                # the data-gradient chain needs delta + dQ alone; the dK / dV kernel and its ordered reduce go to the
                # parameter-gradient branch in front of that weight gradient (queued later on the same in-order stream), and
                # are not issued at all while to_k | to_v are frozen
                ops.attn_bwd(q3, k3, v3, o3, do3, lse, delta, dq3, dk3, dv3, heads, scale, parts=3)
                if kv_train:
                    self._side_defer(lambda: ops.attn_bwd(q3, k3, v3, o3, do3, lse, delta, dq3, dk3, dv3, heads, scale, parts=4))
            else:
                ops.attn_bwd(q3, k3, v3, o3, do3, lse, delta, dq3, dk3, dv3, heads, scale)
        self._tape.append(bwd)
        return self.linear(o, prefix + ".to_out.0.weight", prefix + ".to_out.0.bias", residual=residual)

    def geglu(self, proj: Act, pre: Optional[Act] = None) -> Act:
        """pre: the output was already written by the projection's fused epilogue (linear(..., geglu_out=pre))."""
        rows, H2 = proj.t.shape
        if pre is not None:
            y = pre
        else:
            y = self._new(rows, H2 // 2)
            ops.geglu_fwd(proj.t, y.t)

        def bwd():
            if y.g is None:
                return
            dp = self._gbuf_single(proj, "GEGLU projection")
            ops.geglu_bwd(proj.t, y.g, dp)
        self._tape.append(bwd)
        return y

    def _hoist_shared_input_linears(self, blocks, ctx_a: Act, emb_s: Act):
        """The products whose input does not depend on the layer -- attn2.to_k|to_v of the text context (K/V of every
        cross-attention) and time_emb_proj of every ResnetBlock2D -- for the given blocks, as TWO grouped launches instead of
        one small (20 us, launch-bound) GEMM per layer on the chain.  Called once per parameter region, right where the region's
        parameters are known to have landed (data parallel: behind wait_region_params).  Outputs land in self._hoisted."""
        if not self.policy.hoist:
            return
        kv_jobs, te_jobs = [], []
        for kind, pre, n_layers in blocks:
            if kind == "resnet":
                te_jobs.append((pre + ".temb", self._w[pre + ".time_emb_proj.weight"], self._w[pre + ".time_emb_proj.bias"]))
            else:
                for i in range(n_layers):
                    ap = f"{pre}.transformer_blocks.{i}.attn2"
                    kv_jobs.append((ap + ".kv", self._fused_w([ap + ".to_k.weight", ap + ".to_v.weight"])[0], None))
        for a, jobs in ((ctx_a, kv_jobs), (emb_s, te_jobs)):
            if not jobs:
                continue
            rows, K = a.t.shape
            outs = [self._new(rows, W.shape[0]) for _, W, _ in jobs]
            key = (a.t.data_ptr(),) + tuple(o.t.data_ptr() for o in outs)
            tab = self._group_tables.get(key)
            if tab is None:
                recs, tiles = [], 0
                for (name, W, b), o in zip(jobs, outs):
                    N = W.shape[0]
                    if W.shape[1] != K or N % 8 or W.stride(0) % 8 or o.t.stride(0) % 8 or W.data_ptr() % 16 or o.t.data_ptr() % 16:
                        raise AozoraError("grouped projection: operand layout")
                    recs.append([W.data_ptr(), o.t.data_ptr(), b.data_ptr() if b is not None else 0, N, W.stride(0), o.t.stride(0), tiles])
                    tiles += (N + 159) // 160
                tab = (torch.tensor(recs, dtype=torch.int64, device=self.device), len(recs), tiles, sum(r[3] for r in recs))
                self._group_tables[key] = tab
            ops.gemm_nt_grouped(a.t, tab[0], tab[1], tab[2], tab[3])
            for (name, _, _), o in zip(jobs, outs):
                self._hoisted[name] = o

    def _region_blocks(self, region):
        """(kind, prefix, transformer layers) of the blocks whose parameters lie in parameter region 0 / 1 / 2, forward order."""
        cfg = self.cfg
        nlev = len(cfg.block_out_channels)
        out = []

        def add_level(pre, n_res, tl):
            for j in range(n_res):
                out.append(("resnet", f"{pre}.resnets.{j}", 0))
                if tl > 0:
                    out.append(("attn", f"{pre}.attentions.{j}", tl))
        if region == 0:
            for i in range(nlev - 1):
                add_level(f"down_blocks.{i}", cfg.layers_per_block, cfg.transformer_layers[i])
        elif region == 1:
            add_level(f"down_blocks.{nlev - 1}", cfg.layers_per_block, cfg.transformer_layers[nlev - 1])
        else:
            out.append(("resnet", "mid_block.resnets.0", 0))
            out.append(("attn", "mid_block.attentions.0", cfg.transformer_layers[-1]))
            out.append(("resnet", "mid_block.resnets.1", 0))
            for i in range(nlev):
                add_level(f"up_blocks.{i}", cfg.layers_per_block + 1, cfg.transformer_layers[nlev - 1 - i])
        return out

    def tblock(self, h: Act, B, T, ctx: Act, ctx_len, pre) -> Act:
        self._tape.append(self._block_end)       # runs AFTER this block's backward: its parameter gradients go out as one batch
        n = self.layernorm(h, pre + ".norm1")
        h = self.attention(n, B, T, pre + ".attn1", None, 0, residual=h)
        n = self.layernorm(h, pre + ".norm2")
        h = self.attention(n, B, T, pre + ".attn2", ctx, ctx_len, residual=h)
        n = self.layernorm(h, pre + ".norm3")
        if self.policy.geglu_fuse:       # the GEGLU rides in the epilogue of its projection (pool order: projection first, output second, as unfused)
            W0 = self._w[pre + ".ff.net.0.proj.weight"]
            p_out = self._new(n.t.shape[0], W0.shape[0])
            g_out = self._new(n.t.shape[0], W0.shape[0] // 2)
            p = self.linear(n, pre + ".ff.net.0.proj.weight", pre + ".ff.net.0.proj.bias", out=p_out, geglu_out=g_out)
            g = self.geglu(p, pre=g_out)
        else:
            p = self.linear(n, pre + ".ff.net.0.proj.weight", pre + ".ff.net.0.proj.bias")
            g = self.geglu(p)
        return self.linear(g, pre + ".ff.net.2.weight", pre + ".ff.net.2.bias", residual=h)

    def transformer(self, x: Act, geom, ctx: Act, ctx_len, pre, n_layers, out: Optional[Act] = None) -> Act:
        B, H, W_ = geom
        self._tape.append(self._block_end)
        n = self.groupnorm(x, geom, pre + ".norm", 1e-6, False)
        h = self.linear(n, pre + ".proj_in.weight", pre + ".proj_in.bias")
        for i in range(n_layers):
            h = self.tblock(h, B, H * W_, ctx, ctx_len, f"{pre}.transformer_blocks.{i}")
        return self.linear(h, pre + ".proj_out.weight", pre + ".proj_out.bias", residual=x, out=out)

    def resnet(self, x: Act, geom, emb_s: Act, pre, out: Optional[Act] = None) -> Act:
        self._tape.append(self._block_end)
        n1 = self.groupnorm(x, geom, pre + ".norm1", 1e-5, True)
        t = self.linear(emb_s, pre + ".time_emb_proj.weight", pre + ".time_emb_proj.bias", pre=self._hoisted.pop(pre + ".temb", None), side_dgrad=True)
        h, _ = self.conv(n1, geom, pre + ".conv1.weight", pre + ".conv1.bias", rowbias=t)
        n2 = self.groupnorm(h, geom, pre + ".norm2", 1e-5, True)
        if (pre + ".conv_shortcut.weight") in self._w:
            wsc = self._w[pre + ".conv_shortcut.weight"]
            gsc = self._gw[pre + ".conv_shortcut.weight"]
            Cout, Cin = wsc.shape[0], wsc.shape[3]
            sc = self.linear(x, None, pre + ".conv_shortcut.bias",
                             w_override=(wsc.view(Cout, Cin), gsc.view(Cout, Cin), self._trainable(pre + ".conv_shortcut.weight")))
        else:
            sc = x
        y, _ = self.conv(n2, geom, pre + ".conv2.weight", pre + ".conv2.bias", residual=sc, out=out)
        return y

    def concat(self, a: Act, b: Act, cat: Optional[Act] = None) -> Act:
        """torch.cat([a, b], dim=channels).  cat: the concatenation's buffer when both halves were WRITTEN IN PLACE by
        their producers (`out=` views handed out by forward(): the up path's activation into the left columns, the down path's
        skip tensor -- which its own consumers read through the row stride -- into the right ones): nothing is copied, the
        concatenation never exists as a separate pass.  Without it (policy.cat_inplace False) the two halves are copied."""
        rows, C1 = a.t.shape
        C2 = b.t.shape[1]
        if cat is not None:
            y = cat
            ld = y.t.stride(0)
            if (tuple(y.t.shape) != (rows, C1 + C2) or a.t.data_ptr() != y.t.data_ptr() or b.t.data_ptr() != y.t.data_ptr() + 2 * C1
                    or a.t.stride(0) != ld or b.t.stride(0) != ld):
                raise AozoraError("in-place concat: the halves are not views of the destination")
        else:
            y = self._new(rows, C1 + C2)
            ops.add_rows(a.t, None, y.t[:, :C1])
            ops.add_rows(b.t, None, y.t[:, C1:])

        def bwd():
            if y.g is None:
                return
            self._give_grad(a, y.g[:, :C1], alias=y.g_alias)      # slices of the concat's gradient (nobody's dY unless y.g is)
            self._give_grad(b, y.g[:, C1:], alias=y.g_alias)
        self._tape.append(bwd)
        return y

    def upsample(self, x: Act, geom) -> Tuple[Act, tuple]:
        B, H, W_ = geom
        C = x.t.shape[1]
        y = self._new(B * 4 * H * W_, C)
        ops.upsample2x_fwd(x.t.view(B, H, W_, C), y.t.view(B, 2 * H, 2 * W_, C))

        def bwd():
            if y.g is None:
                return
            dx = self._gbuf_single(x, "upsample input")
            if not y.g.is_contiguous():
                raise AozoraError("upsample gradient must be contiguous")
            ops.upsample2x_bwd(y.g.view(B, 2 * H, 2 * W_, C), dx.view(B, H, W_, C))
        self._tape.append(bwd)
        return y, (B, 2 * H, 2 * W_)

    # ------------------------------------------------------------------ whole model ---------------
    def __del__(self):
        try:
            for ev in getattr(self, "_events", []):
                if isinstance(ev, ForkEvent):
                    ev.destroy()
            if getattr(self, "_ctx", None):
                lib()._fn["az_destroy"](self._ctx)
                self._ctx = None
        except Exception:
            pass

    def begin_step(self, key):
        lib()._fn["az_make_current"](self._ctx)      # this thread's launches read THIS UNet's option table from here on
        if key not in self._pools:
            self._pools[key] = _Pool(self.device)
        self.refresh_transposed()
        self._pool = self._pools[key]
        self._pool.reset()
        self._tape = []
        self._ev_cursor = 0
        self._side_used = False
        self._side_rr = 0
        self._side_q = []
        self._side_done = None
        self._ln_jobs = []
        self._tn_jobs = []
        self._xkv_jobs = []
        self._hoisted = {}

    def forward_nhwc(self, x8: torch.Tensor, t_f32: torch.Tensor, ctx: torch.Tensor, pooled: torch.Tensor,
                     time_ids_f32: torch.Tensor) -> Act:
        """x8 (B,H,W,8) bf16 (channels 4..7 zero) ; t_f32 (B,) ; ctx (B,L,ctx_dim) bf16 ; pooled (B,P) bf16 ;
        time_ids_f32 (B,6) fp32 (values already rounded through bf16, train.py:2731). -> pred Act [B*H*W][4]"""
        cfg = self.cfg
        B, H, W_, _ = x8.shape
        L = ctx.shape[1]
        ch = cfg.block_out_channels
        nlev = len(ch)
        T = cfg.time_embed_dim
        self._live(self._set_forward_exclusive)
        # ---- embeddings (a7.1) ----
        tsin = self._new(B, ch[0], need_grad=False)
        ops.timestep_embed(t_f32, ch[0], tsin.t)
        e = self.linear(tsin, "time_embedding.linear_1.weight", "time_embedding.linear_1.bias")
        e = self.silu(e)
        emb_t = self.linear(e, "time_embedding.linear_2.weight", "time_embedding.linear_2.bias")
        idsin = self._new(B * 6, cfg.addition_time_embed_dim, need_grad=False)
        ops.timestep_embed(time_ids_f32.reshape(-1), cfg.addition_time_embed_dim, idsin.t)
        addin = self._new(B, cfg.add_in_dim, need_grad=False)
        ops.add_rows(pooled, None, addin.t[:, :cfg.pooled_dim])
        ops.add_rows(idsin.t.view(B, 6 * cfg.addition_time_embed_dim), None, addin.t[:, cfg.pooled_dim:])
        a = self.linear(addin, "add_embedding.linear_1.weight", "add_embedding.linear_1.bias")
        a = self.silu(a)
        emb = self.linear(a, "add_embedding.linear_2.weight", "add_embedding.linear_2.bias", residual=emb_t)
        emb_s = self.silu(emb)
        ctx_a = Act(ctx.reshape(B * L, ctx.shape[2]), need_grad=False)
        # ---- down ----
        xin = Act(x8.view(B * H * W_, x8.shape[3]), need_grad=False)
        geom = (B, H, W_)
        # Skip concatenations without copies (K14): the s-th skip tensor (push order) is consumed by up-path resnet u = S-1-s, whose
        # other input has C1(u) channels; the [rows][C1 + C2] buffer of that concatenation is allocated when the skip tensor is
        # produced, the producer writes into its right columns, and the up path's producer later writes into the left ones.
        lpb = cfg.layers_per_block
        n_skips = 1 + nlev * lpb + (nlev - 1)
        cats: Dict[int, Act] = {}

        def c1_of(u):
            i, j = divmod(u, lpb + 1)
            lev = nlev - 1 - i
            return ch[lev] if j > 0 else (ch[nlev - 1] if i == 0 else ch[lev + 1])

        def skip_dest(rows, C2):
            """destination view for the next skip tensor (None: plain allocation, copied by concat later)"""
            if not self.policy.cat_inplace:
                return None
            u = n_skips - 1 - len(skips)
            cat = self._new(rows, c1_of(u) + C2)
            cats[u] = cat
            return Act(cat.t[:, c1_of(u):])

        def up_dest(u):
            """destination view for the up-path activation that up-path resnet u concatenates with its skip tensor"""
            cat = cats.get(u)
            return Act(cat.t[:, :c1_of(u)]) if cat is not None else None

        skips: List[Act] = []
        h, _ = self.conv(xin, geom, "conv_in.weight", "conv_in.bias", out=skip_dest(B * H * W_, ch[0]))
        self._hoist_shared_input_linears(self._region_blocks(0), ctx_a, emb_s)
        skips.append(h)
        for i in range(nlev):
            pre = f"down_blocks.{i}"
            if i == nlev - 1:
                self._tape_mark1 = len(self._tape)     # backward entries in [mark1, mark) belong to the last down block (region 1)
                self._live(self._wait_region1)         # DP overlap: region 1's all-gather must have landed by now
                self._hoist_shared_input_linears(self._region_blocks(1), ctx_a, emb_s)
            rows = geom[0] * geom[1] * geom[2]
            for j in range(lpb):
                if cfg.transformer_layers[i] > 0:
                    h = self.resnet(h, geom, emb_s, f"{pre}.resnets.{j}")
                    h = self.transformer(h, geom, ctx_a, L, f"{pre}.attentions.{j}", cfg.transformer_layers[i], out=skip_dest(rows, ch[i]))
                else:
                    h = self.resnet(h, geom, emb_s, f"{pre}.resnets.{j}", out=skip_dest(rows, ch[i]))
                skips.append(h)
            if i < nlev - 1:
                h, geom = self.conv(h, geom, f"{pre}.downsamplers.0.conv.weight", f"{pre}.downsamplers.0.conv.bias", stride=2,
                                    out=skip_dest(rows // 4, ch[i]))
                skips.append(h)
        # ---- mid ----
        self._tape_mark = len(self._tape)       # backward entries >= mark belong to mid / up / head-out (the "tail" region)
        self._live(self._wait_region2)          # DP overlap: the tail parameters' all-gather must have landed by now
        self._hoist_shared_input_linears(self._region_blocks(2), ctx_a, emb_s)
        h = self.resnet(h, geom, emb_s, "mid_block.resnets.0")
        h = self.transformer(h, geom, ctx_a, L, "mid_block.attentions.0", cfg.transformer_layers[-1])
        h = self.resnet(h, geom, emb_s, "mid_block.resnets.1", out=up_dest(0))
        # ---- up ----
        u = 0
        for i in range(nlev):
            lev = nlev - 1 - i
            pre = f"up_blocks.{i}"
            for j in range(lpb + 1):
                h = self.concat(h, skips.pop(), cats.get(u))
                u += 1
                nxt = up_dest(u) if j < lpb else None       # the block's last output feeds the upsampler (or conv_norm_out), not a concat
                if cfg.transformer_layers[lev] > 0:
                    h = self.resnet(h, geom, emb_s, f"{pre}.resnets.{j}")
                    h = self.transformer(h, geom, ctx_a, L, f"{pre}.attentions.{j}", cfg.transformer_layers[lev], out=nxt)
                else:
                    h = self.resnet(h, geom, emb_s, f"{pre}.resnets.{j}", out=nxt)
            if i < nlev - 1:       # Upsample2D: nearest-2x folded into the conv's operand gather
                h, geom = self.conv(h, geom, f"{pre}.upsamplers.0.conv.weight", f"{pre}.upsamplers.0.conv.bias", upsample=True, out=up_dest(u))
        n = self.groupnorm(h, geom, "conv_norm_out", 1e-5, True)
        pred, _ = self.conv(n, geom, "conv_out.weight", "conv_out.bias")
        return pred

    def backward_nhwc(self, pred: Act, dpred8: torch.Tensor, after_tail=None):
        """dpred8 (B,H,W,8) bf16: d(loss)/d(pred), channels >= out_channels zero. Gradients are
        ACCUMULATED into the flat gradient buffer."""
        lib().call("az_gemm_set_exclusive", 0)       # the parameter-gradient stream shares the CUs from here on
        self._live(self._wait_wt_ready)              # an asynchronous W^T refresh must have finished before the first dgrad
        B, H, W_, Cp = dpred8.shape
        pred.g = dpred8.view(B * H * W_, Cp)
        mark, mark1 = self._tape_mark, self._tape_mark1
        self._after_tail_hook = after_tail
        for idx in range(len(self._tape) - 1, -1, -1):
            if idx == mark - 1:
                self._finish_ln_jobs()
                self._finish_tn_jobs()
                self._finish_xkv_jobs()
                self._flush_side()
                self._live(self._run_after_tail)   # every gradient of region 2 has been issued (main + side stream)
            if idx == mark1 - 1:
                self._finish_ln_jobs()
                self._finish_tn_jobs()
                self._finish_xkv_jobs()
                self._flush_side()
                self._live(self._run_region_hook1) # ... and now those of region 1 (the last down block)
            self._tape[idx]()
        self._finish_ln_jobs()
        self._finish_tn_jobs()
        self._finish_xkv_jobs()
        self._flush_side()
        self._tape = []
        if self.concurrent_wgrad and self._side_used:      # join the parameter-gradient branches (or let them run on)
            self._live(self._end_join)
            self._side_used = False
