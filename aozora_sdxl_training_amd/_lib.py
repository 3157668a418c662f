"""ctypes binding of libaozora_hip.so (the C ABI declared in include/aozora_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, this
module raises.  Prototypes are parsed from the header so the binding cannot drift from it.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "aozora_hip.h")
LIB_PATH = os.path.join(_HERE, "libaozora_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "void**": ctypes.POINTER(ctypes.c_void_p), "int*": ctypes.POINTER(ctypes.c_int),
    "float*": ctypes.POINTER(ctypes.c_float), "const char*": ctypes.c_char_p, "long*": ctypes.POINTER(ctypes.c_long),
}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """-> {name: (return_type, [(arg_type, arg_name), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(az_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty = mm.group(1).strip().replace(" *", "*")
                alist.append((ty, mm.group(2)))
        protos[name] = (ret, alist)
    return protos


class AozoraError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise AozoraError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self._fn = {}
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)   # AttributeError if the symbol is not exported
            fn.restype = _CTYPES[ret]
            fn.argtypes = [_CTYPES[t] for t, _ in args]
            self._fn[name] = fn
        # Host launch tape (train_step.TrainStep): while `recorder` is a list every ABI call is appended to it as
        # (bound C function, argument tuple).  The executor's launch sequence is static -- same entry points, same
        # pointers, same streams every step -- so later steps re-issue the tape directly and skip the Python that
        # derived the arguments (operand checks, view arithmetic, tape of closures): ~25 us -> ~3 us of host time per
        # launch, which otherwise throttles the GPU where kernels are short.
        self.recorder = None

    def call(self, name: str, *args):
        fn = self._fn[name]
        rc = fn(*args)
        if rc != 0:
            raise AozoraError(f"{name} failed with code {rc}" + (" (argument error)" if rc <= -1000 else " (HIP error)"))
        if self.recorder is not None:
            self.recorder.append((fn, args))
        return rc

    def raw(self, name: str):
        return getattr(self.cdll, name)


def set_option(name: str, value: int):
    """Runtime option of the library (include/aozora_hip.h az_set_option); never recorded on a launch tape."""
    L = lib()
    rc = L._fn["az_set_option"](name.encode(), int(value))
    if rc != 0:
        raise AozoraError(f"az_set_option({name!r}) failed with code {rc}")


def get_option(name: str) -> int:
    v = ctypes.c_int()
    rc = lib()._fn["az_get_option"](name.encode(), ctypes.byref(v))
    if rc != 0:
        raise AozoraError(f"az_get_option({name!r}) failed with code {rc}")
    return v.value


class ForkEvent:
    """A fork / join event of the two-stream executor (include/aozora_hip.h az_event_create_fork): orders kernels of one device,
    is never timed or read by the host, and is created without the system-scope fence torch.cuda.Event records with -- half the
    cost to the recording stream.  `cuda_event` / record() mirror torch.cuda.Event so that launch tapes treat both alike;
    wait_on(stream) is stream.wait_event(event).  The calls bypass _Lib.call: the executor puts them on a recording tape itself."""
    __slots__ = ("cuda_event",)

    def __init__(self):
        h = ctypes.c_void_p()
        rc = lib()._fn["az_event_create_fork"](ctypes.byref(h))
        if rc != 0:
            raise AozoraError(f"az_event_create_fork failed with code {rc}")
        self.cuda_event = h.value

    def record(self, stream):
        rc = lib()._fn["az_event_record"](self.cuda_event, stream.cuda_stream)
        if rc != 0:
            raise AozoraError(f"az_event_record failed with code {rc}")

    def wait_on(self, stream):
        rc = lib()._fn["az_stream_wait_event"](stream.cuda_stream, self.cuda_event)
        if rc != 0:
            raise AozoraError(f"az_stream_wait_event failed with code {rc}")

    def destroy(self):
        if self.cuda_event:
            lib()._fn["az_event_destroy"](self.cuda_event)
            self.cuda_event = None


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
