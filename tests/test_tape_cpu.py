"""Launch-tape peephole (tape.fuse_records) and the stop-event slot of the launch wrapper (include/aozora_hip.h
az_set_launch_stop_event): host logic only, no device needed."""
import ctypes

import torch

from aozora_sdxl_training_amd._lib import lib, ForkEvent
from aozora_sdxl_training_amd.tape import fuse_records


class _Stream:
    def __init__(self, h):
        self.cuda_stream = h


def _ev(h):
    e = ForkEvent.__new__(ForkEvent)      # no HIP call: only the handle and the bound methods matter to the peephole
    e.cuda_event = h
    return e


def _call(name, stream, L):
    proto = L.protos[name][1]
    args = [0] * len(proto)
    args[[i for i, (_, an) in enumerate(proto) if an == "stream"][0]] = ctypes.c_void_p(stream.cuda_stream)
    return (L._fn[name], tuple(args))


def _shape(seq, L):
    by_fn = {id(fn): name for name, fn in L._fn.items()}
    out = []
    for fn, args in seq:
        name = by_fn.get(id(fn))
        if name == "az_set_launch_stop_event":
            a = args[0]
            out.append(("set", a.value if hasattr(a, "value") else a))
        elif name is not None:
            out.append(name)
        else:
            owner = getattr(fn, "__self__", None)
            out.append((fn.__name__, getattr(owner, "cuda_event", None)))
    return out


def test_record_behind_a_kernel_rides_on_it_and_a_repeated_record_is_an_alias():
    L = lib()
    A, B = _Stream(0x1000), _Stream(0x2000)
    e1, e2, e3 = _ev(11), _ev(22), _ev(33)
    seq = [_call("az_add_rows", A, L), (e1.record, (A,)), (e1.wait_on, (B,)), _call("az_silu_bwd", B, L), (e3.record, (B,)),
           (e2.record, (A,)), (e2.wait_on, (B,)), _call("az_geglu_bwd", B, L)]
    out, fused = fuse_records(seq)
    assert fused == 3
    assert _shape(out, L) == [("set", 11), "az_add_rows", ("set", None), ("wait_on", 11), ("set", 33), "az_silu_bwd", ("set", None),
                              ("wait_on", 11), "az_geglu_bwd"]          # e2 marks the same point of A as e1: dropped, its wait re-pointed
    out, fused = fuse_records(seq, only_stream=A.cuda_stream)
    assert fused == 2 and ("record", 33) in _shape(out, L) and ("set", 33) not in _shape(out, L)


def test_anything_between_the_kernel_and_the_record_keeps_the_record():
    L = lib()
    A, B = _Stream(0x1000), _Stream(0x2000)
    for between in ([(_ev(5).wait_on, (A,))],                                  # the stream waits for something first
                    [(lambda: 0, ())],                                         # host logic
                    [_call("az_memset_async", A, L)],                          # not a kernel of the launch wrapper
                    [(L._fn["az_set_option"], (b"GEMM8", 1))]):                # no stream argument: do not reason across it
        e = _ev(7)
        seq = [_call("az_add_rows", A, L)] + between + [(e.record, (A,)), (e.wait_on, (B,))]
        out, fused = fuse_records(seq)
        assert fused == 0 and ("record", 7) in _shape(out, L)
    e = _ev(9)
    t = torch.cuda.Event.__new__(torch.cuda.Event) if False else None      # torch events are never fused: only ForkEvent records are looked at
    seq = [_call("az_add_rows", A, L), _call("az_add_rows", B, L), (e.record, (A,))]      # a launch on ANOTHER stream in between does not matter
    out, fused = fuse_records(seq)
    assert fused == 1 and _shape(out, L)[:3] == [("set", 9), "az_add_rows", ("set", None)]


def test_clearing_a_stop_event_no_launch_carried_is_an_error():
    L = lib()
    f = L._fn["az_set_launch_stop_event"]
    assert f(None) == 0
    assert f(ctypes.c_void_p(0x1234)) == 0
    assert f(None) != 0                   # nothing was launched while it was set: it would stay unrecorded
    assert f(None) == 0
