"""Native launch tape (include/aozora_hip.h az_tape_*): the recorded launch sequence of a resolution bucket -- C-ABI calls,
event records, stream waits -- compiled into a C-side tape and re-issued by az_tape_play without the interpreter; the host
logic the executor marked as live (data-parallel region waits, scheduling hints, the end-of-backward join) stays Python and runs
at the tape's BREAK operations.  SURVEY.md 8b names this seam `az_unet_step`; the reference issues the same sequence from
Python / autograd every step (train.py:2743-2767)."""
from __future__ import annotations

import ctypes
import struct

import torch

from ._lib import lib, AozoraError, ForkEvent

OP_CALL, OP_EVENT_RECORD, OP_STREAM_WAIT, OP_BREAK = 0, 1, 2, 3


def _word(ty: str, a) -> int:
    if ty == "float":
        return struct.unpack("<I", struct.pack("<f", float(a)))[0]
    if hasattr(a, "value"):            # ctypes scalar / c_void_p
        a = a.value
    if a is None:
        return 0
    return int(a)


# Entry points whose LAST operation is a kernel launched on `stream` through the library's launch wrapper (az_launch): only these
# may carry a fork event as the completion signal of that kernel.  An ALLOW-list: an entry point missing here is merely not fused
# (its record packet stays), while a wrongly fused one -- last operation a memcpy, a memset, a stream wait -- would leave its event
# unrecorded.  tests/test_abi_cpu.py checks that every entry point with a stream argument is classified in exactly one of the two.
_KERNEL_ENTRIES = frozenset({
    "az_spin", "az_gemm_bf16", "az_gemm_nt_grouped_bf16", "az_gemm_tn_grouped_bf16", "az_gemm_geglu_fwd_bf16", "az_gemm_wgrad_bias_bf16",
    "az_conv2d_bf16", "az_conv2d_wgrad_bias_bf16", "az_attn_fwd", "az_attn_bwd", "az_groupnorm_fwd", "az_groupnorm_bwd",
    "az_groupnorm_bwd_ex", "az_layernorm_fwd", "az_layernorm_bwd", "az_layernorm_bwd_ex", "az_layernorm_bwd_partial",
    "az_ln_param_finish_multi", "az_geglu_fwd", "az_geglu_bwd", "az_silu_fwd", "az_silu_bwd", "az_add_rows", "az_upsample2x_fwd",
    "az_upsample2x_bwd", "az_colsum", "az_colsum_grad", "az_reduce_segs_to_bf16", "az_transpose_bf16", "az_transpose_bf16_batched",
    "az_transpose_multi_bf16", "az_f32_to_bf16", "az_timestep_embed", "az_nchw_to_nhwc_pad", "az_nhwc_to_nchw", "az_noise_target",
    "az_mse_loss_fwd_bwd", "az_sumsq_bf16", "az_sumsq", "az_clip_coef", "az_adamw_flat", "az_adamw_flat_ex", "az_scale_bf16",
    "az_scale_f32", "az_stage_inputs"})
# ... and the ones with a stream argument that end in something else (graph capture / launch, event and stream calls, copies)
_NOT_KERNEL_ENTRIES = frozenset({"az_graph_begin", "az_graph_end", "az_graph_launch", "az_event_record", "az_stream_wait_event",
                                 "az_stream_sync", "az_memset_async", "az_memcpy_async", "az_titan_offload"})


def fuse_records(recorded, only_stream=None):
    """Peephole over a recorded launch sequence: `ABI call X on stream S` directly followed (on S) by `ForkEvent.record(S)` becomes
    `az_set_launch_stop_event(ev); X; az_set_launch_stop_event(NULL)` -- the event rides on X's last kernel as its completion
    signal and the record packet disappears from S (include/aozora_hip.h az_set_launch_stop_event; tools/event_cost.cpp).
    A second record on S with nothing queued on S since the first marks the same point: it is dropped and the waits on its event
    are re-pointed to the first event.  Anything else that touches S in between (another call, a wait, host logic) keeps the record.
    only_stream: fuse records on that stream only.  -> (new sequence, number of records fused)."""
    L = lib()
    by_fn = {id(fn): name for name, fn in L._fn.items()}
    stream_arg = {name: [i for i, (_, an) in enumerate(args) if an == "stream"] for name, (_, args) in L.protos.items()}
    set_ev = L._fn["az_set_launch_stop_event"]
    out, fused = [], 0          # out: [entry, event handle or None]
    last = {}                   # stream handle -> index in out of the latest kernel-launching call on it, while nothing else touched the stream
    last_rec = {}               # stream handle -> the ForkEvent recorded on it last, while nothing else touched the stream since
    alias = {}                  # id(ForkEvent) -> the earlier ForkEvent that marks the same point of its stream (its record is dropped)
    for fn, args in recorded:
        name = by_fn.get(id(fn))
        owner = getattr(fn, "__self__", None)
        fname = getattr(fn, "__name__", "")
        if name is not None:
            idx = stream_arg.get(name) or []
            out.append([(fn, args), None])
            if not idx or len(L.protos[name][1]) != len(args):
                last.clear(); last_rec.clear()      # option changes, context calls ...: do not reason across them
            else:
                st = _word("void*", args[idx[0]])
                last_rec.pop(st, None)
                if name not in _KERNEL_ENTRIES:
                    last.pop(st, None)
                else:
                    last[st] = len(out) - 1
            continue
        if isinstance(owner, ForkEvent) and fname == "record":
            st = args[0].cuda_stream
            if only_stream is None or st == only_stream:
                if st in last_rec:           # nothing was queued on the stream since the previous record: the same point of the stream
                    alias[id(owner)] = last_rec[st]
                    fused += 1
                    continue
                k = last.pop(st, None)
                if k is not None and out[k][1] is None:
                    out[k][1] = owner.cuda_event
                    last_rec[st] = owner
                    fused += 1
                    continue
            last.pop(st, None)
            last_rec[st] = owner
            out.append([(fn, args), None])
            continue
        if isinstance(owner, ForkEvent) and fname == "wait_on":
            if id(owner) in alias:
                fn = alias[id(owner)].wait_on
            out.append([(fn, args), None])
            last.pop(args[0].cuda_stream, None); last_rec.pop(args[0].cuda_stream, None)
            continue
        out.append([(fn, args), None])
        if isinstance(owner, torch.cuda.Stream) and fname == "wait_event":
            last.pop(owner.cuda_stream, None); last_rec.pop(owner.cuda_stream, None)
        elif isinstance(owner, torch.cuda.Event) and fname == "record":
            last.pop(args[0].cuda_stream, None); last_rec.pop(args[0].cuda_stream, None)
        else:
            last.clear(); last_rec.clear()   # host logic: anything may happen inside
    flat = []
    for entry, ev in out:
        if ev is None:
            flat.append(entry)
        else:
            flat += [(set_ev, (ctypes.c_void_p(ev),)), entry, (set_ev, (None,))]
    return flat, fused


def disarm_stop_event():
    """Clear this thread's launch stop event after a failed replay (az_set_launch_stop_event(NULL); its 'never carried' error is
    the very state being cleaned up and is ignored)."""
    lib()._fn["az_set_launch_stop_event"](None)


class NativeTape:
    def __init__(self, recorded):
        L = lib()
        self._L = L
        self.handle = ctypes.c_void_p()
        if L._fn["az_tape_create"](ctypes.byref(self.handle)):      # not through L.call: nothing here may land on a recording tape
            raise AozoraError("az_tape_create failed")
        by_fn = {id(fn): name for name, fn in L._fn.items()}
        self.callbacks = {}            # op index of a BREAK -> (callable, args)
        self.n = 0
        self.n_calls = 0
        add = L._fn["az_tape_add"]

        def push(kind, fn_id, words):
            arr = (ctypes.c_long * max(1, len(words)))(*[w if w < (1 << 63) else w - (1 << 64) for w in words])
            rc = add(self.handle, kind, fn_id, ctypes.cast(arr, ctypes.c_void_p), len(words))
            if rc:
                raise AozoraError(f"az_tape_add failed with code {rc}")
            self.n += 1

        for fn, args in recorded:
            name = by_fn.get(id(fn))
            if name is not None:
                fid = L._fn["az_tape_fn_id"](name.encode())
                protos = L.protos[name][1]
                if fid >= 0 and len(protos) == len(args):
                    push(OP_CALL, fid, [_word(t, a) for (t, _), a in zip(protos, args)])
                    self.n_calls += 1
                    continue
            owner = getattr(fn, "__self__", None)
            if isinstance(owner, (torch.cuda.Event, ForkEvent)) and getattr(fn, "__name__", "") == "record":
                push(OP_EVENT_RECORD, 0, [owner.cuda_event, args[0].cuda_stream])
                continue
            if isinstance(owner, torch.cuda.Stream) and getattr(fn, "__name__", "") == "wait_event":
                push(OP_STREAM_WAIT, 0, [owner.cuda_stream, args[0].cuda_event])
                continue
            if isinstance(owner, ForkEvent) and getattr(fn, "__name__", "") == "wait_on":
                push(OP_STREAM_WAIT, 0, [args[0].cuda_stream, owner.cuda_event])
                continue
            self.callbacks[self.n] = (fn, args)       # host logic: runs in the caller at a BREAK
            push(OP_BREAK, 0, [])
        self._play = L._fn["az_tape_play"]

    def play(self):
        try:
            self._play_all()
        except BaseException:
            disarm_stop_event()        # a failing entry / callback between `set` and `clear` must not leave the event armed
            raise

    def _play_all(self):
        i = 0
        while i < self.n:
            nxt = self._play(self.handle, i)
            if nxt < 0:
                idx, rc = ctypes.c_long(), ctypes.c_int()
                self._L._fn["az_tape_last_error"](self.handle, ctypes.byref(idx), ctypes.byref(rc))
                raise AozoraError(f"native launch tape: operation {idx.value} failed with code {rc.value}")
            if nxt <= i:
                raise AozoraError("native launch tape made no progress")
            cb = self.callbacks.get(nxt - 1)
            if cb is not None:
                fn, args = cb
                if fn(*args):
                    raise AozoraError(f"{getattr(fn, '__name__', fn)} failed while re-issuing the launch tape")
            i = nxt

    def __del__(self):
        try:
            if self.handle:
                self._L._fn["az_tape_destroy"](self.handle)
        except Exception:
            pass
