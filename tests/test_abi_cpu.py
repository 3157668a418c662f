"""The C-ABI boundary on a machine WITHOUT a GPU: libaozora_hip.so loads, exports every symbol include/aozora_hip.h declares
(and nothing the header does not declare under the az_ prefix), the ctypes binding is derived from that header, every
prototype block cites the reference call it replaces, and the product path fails loudly -- never falls back -- when the
library is missing.  No kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from aozora_sdxl_training_amd import _lib as L        # noqa: E402


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.LIB_PATH


def test_every_declared_symbol_is_exported(built):
    protos = L.parse_header()
    assert len(protos) >= 60
    dll = ctypes.CDLL(built)
    for name, (ret, args) in protos.items():
        assert hasattr(dll, name), f"{name} is declared in include/aozora_hip.h but not exported"
        assert ret in ("int", "long") and all(t in L._CTYPES for t, _ in args), (name, ret, args)
    # and the other way round: exported az_* symbols are all declared (the header is the whole boundary)
    import shutil
    nm = shutil.which("nm")
    out = subprocess.run([nm, "-D", "--defined-only", built], capture_output=True, text=True) if nm else None
    if out is not None and out.returncode == 0:
        exported = {ln.split()[-1] for ln in out.stdout.splitlines() if ln.split() and ln.split()[-1].startswith("az_")}
        assert exported == set(protos), (exported ^ set(protos))


def test_binding_is_derived_from_the_header(built):
    lib = L.lib()
    for name, (ret, args) in lib.protos.items():
        fn = lib.raw(name)
        assert len(fn.argtypes) == len(args) and fn.restype is L._CTYPES[ret]
    assert lib.raw("az_version")() >= 1            # pure host function: proves calls go through without a device


def test_header_cites_the_reference():
    src = open(L.HEADER).read()
    cites = re.findall(r"(train\.py|raven\.py|titan\.py):\d+", src)
    assert len(cites) >= 25, "every group of entry points must say which reference call it replaces (file:line)"
    assert "torch" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S).lower()      # no torch types in the signatures


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.AozoraError, match="no CPU fallback"):
        L._Lib()


def test_oracle_is_not_imported_by_the_product():
    pkg = os.path.join(ROOT, "aozora_sdxl_training_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"


def test_tape_dispatch_table_is_current():
    """csrc/az_tape_dispatch.inc is generated from the header (tools/gen_tape_dispatch.py): the committed file must match."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_tape_dispatch", os.path.join(ROOT, "tools", "gen_tape_dispatch.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(ROOT, "aozora_sdxl_training_amd", "csrc", "az_tape_dispatch.inc")).read()
    assert committed == gen.generate()


def test_native_tape_plays_breaks_and_validates_operations():
    """The C-side tape player without any GPU work: BREAK operations hand control back in order, malformed operations are refused,
    every tape-able entry point has an id (no compute calls: CPU suite)."""
    import ctypes
    lib = L._Lib()
    t = ctypes.c_void_p()
    assert lib._fn["az_tape_create"](ctypes.byref(t)) == 0
    add, play = lib._fn["az_tape_add"], lib._fn["az_tape_play"]
    for _ in range(3):
        assert add(t, 3, 0, None, 0) == 0                       # three breaks
    assert [play(t, 0), play(t, 1), play(t, 2), play(t, 3)] == [1, 2, 3, 3]
    assert add(t, 0, 10 ** 6, None, 0) != 0                      # unknown entry point
    fid = lib._fn["az_tape_fn_id"](b"az_gemm_bf16")
    assert fid >= 0 and lib._fn["az_tape_fn_id"](b"az_graph_end") == -1      # pointer-to-pointer signature: not tape-able
    words = (ctypes.c_long * 3)(1, 2, 3)
    assert add(t, 0, fid, ctypes.cast(words, ctypes.c_void_p), 3) != 0        # wrong argument count for az_gemm_bf16
    assert add(t, 1, 0, ctypes.cast(words, ctypes.c_void_p), 3) != 0          # event record takes two words
    assert lib._fn["az_tape_destroy"](t) == 0


def test_contexts_carry_their_own_option_table(built):
    """az_init / az_make_current / az_destroy: a context owns a copy of the option table; while it is current on a
    thread the launchers and az_set_option / az_get_option of THAT thread use it, other threads and the process-wide table are
    untouched.  Host-only calls: runs without a GPU."""
    import threading
    lib = L.lib()
    v = ctypes.c_int()
    get = lambda name: (lib.call("az_get_option", name.encode(), ctypes.byref(v)), v.value)[1]
    base = get("SPLIT_SLOTS")
    h1, h2 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.call("az_init", 0, ctypes.byref(h1)); lib.call("az_init", 3, ctypes.byref(h2))
    dev = ctypes.c_int()
    lib.call("az_context_device", h2, ctypes.byref(dev))
    assert dev.value == 3
    try:
        lib.call("az_make_current", h1)
        assert get("SPLIT_SLOTS") == base                       # a fresh context starts from the process-wide values
        lib.call("az_set_option", b"SPLIT_SLOTS", base + 128)
        assert get("SPLIT_SLOTS") == base + 128
        lib.call("az_make_current", h2)
        assert get("SPLIT_SLOTS") == base                       # the other context is its own table
        seen = {}
        t = threading.Thread(target=lambda: seen.setdefault("other_thread", get("SPLIT_SLOTS")))
        t.start(); t.join()
        assert seen["other_thread"] == base                     # binding is per thread
        lib.call("az_make_current", None)
        assert get("SPLIT_SLOTS") == base                       # the process-wide table never changed
        lib.call("az_make_current", h1)
        assert get("SPLIT_SLOTS") == base + 128
    finally:
        lib.call("az_make_current", None)
        lib.call("az_destroy", h1); lib.call("az_destroy", h2)
    with pytest.raises(L.AozoraError):
        lib.call("az_destroy", None)


def test_context_destroyed_on_one_thread_stays_valid_where_it_is_current(built):
    """A context's owner may be collected on any thread (AozoraUNet.__del__ runs where the GC fires): a thread that still has the
    context current keeps reading ITS table -- the memory goes with the last reference -- and returns to the process-wide table
    once it makes something else current."""
    import threading
    lib = L.lib()
    v = ctypes.c_int()
    get = lambda name: (lib.call("az_get_option", name.encode(), ctypes.byref(v)), v.value)[1]
    base = get("SPLIT_SLOTS")
    h = ctypes.c_void_p()
    lib.call("az_init", 0, ctypes.byref(h))
    bound, destroyed, seen = threading.Event(), threading.Event(), {}

    def user():
        w = ctypes.c_int()
        g = lambda: (lib.call("az_get_option", b"SPLIT_SLOTS", ctypes.byref(w)), w.value)[1]
        lib.call("az_make_current", h)
        lib.call("az_set_option", b"SPLIT_SLOTS", base + 64)
        bound.set(); destroyed.wait(10)
        seen["after_destroy"] = g()                 # the handle is gone for its owner, not for this thread
        lib.call("az_set_option", b"SPLIT_SLOTS", base + 65)
        seen["after_set"] = g()
        lib.call("az_make_current", None)           # last reference: freed here
        seen["unbound"] = g()

    t = threading.Thread(target=user)
    t.start(); bound.wait(10)
    lib.call("az_destroy", h)                       # from a different thread than the one that has it current
    destroyed.set(); t.join()
    assert seen == {"after_destroy": base + 64, "after_set": base + 65, "unbound": base}
    assert get("SPLIT_SLOTS") == base


def test_every_stream_entry_point_is_classified_for_event_fusion(built):
    """tape.fuse_records lets a fork event ride on the last kernel of an entry point only for names on an ALLOW-list; a new entry
    point with a stream argument must be put on one of the two lists before this passes (an unclassified one is simply not fused)."""
    from aozora_sdxl_training_amd import tape
    lib = L.lib()
    with_stream = {n for n, (_, args) in lib.protos.items() if any(an == "stream" for _, an in args)}
    assert not (tape._KERNEL_ENTRIES & tape._NOT_KERNEL_ENTRIES)
    assert with_stream == (tape._KERNEL_ENTRIES | tape._NOT_KERNEL_ENTRIES), sorted(with_stream ^ (tape._KERNEL_ENTRIES | tape._NOT_KERNEL_ENTRIES))


def test_failed_tape_replay_leaves_no_stop_event_armed(built):
    """A tape whose entry fails between `set stop event` and `clear stop event` (az_tape_play aborts there) must not leave the
    event armed on the thread: the next az_set_launch_stop_event(NULL) would otherwise report an event nobody carried (-1058),
    and every launch in between would record a stale event.  Host-only: the failing entry rejects its arguments before any
    HIP call."""
    lib = L.lib()
    t = ctypes.c_void_p()
    assert lib._fn["az_tape_create"](ctypes.byref(t)) == 0
    add = lib._fn["az_tape_add"]
    def call(name, words):
        fid = lib._fn["az_tape_fn_id"](name.encode())
        assert fid >= 0, name
        arr = (ctypes.c_long * len(words))(*words)
        assert add(t, 0, fid, ctypes.cast(arr, ctypes.c_void_p), len(words)) == 0
    call("az_set_launch_stop_event", [0x1234])                 # arm (the handle is never used: nothing launches)
    nargs = len(lib.protos["az_layernorm_fwd"][1])
    call("az_layernorm_fwd", [0] * nargs)                      # M = 0: rejected with an argument error before any launch
    call("az_set_launch_stop_event", [0])
    assert lib._fn["az_tape_play"](t, 0) == -2
    idx, rc = ctypes.c_long(), ctypes.c_int()
    lib._fn["az_tape_last_error"](t, ctypes.byref(idx), ctypes.byref(rc))
    assert idx.value == 1 and rc.value != 0
    assert lib._fn["az_set_launch_stop_event"](None) == 0      # nothing armed any more
    assert lib._fn["az_tape_destroy"](t) == 0
