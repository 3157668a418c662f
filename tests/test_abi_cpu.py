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
