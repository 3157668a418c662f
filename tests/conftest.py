import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads():
    """The CPU oracle runs beside every GPU parity test: with torch's default (one thread per host CPU: 256 on a GPU box) its
    small ops oversubscribe and the mini-UNet oracle steps take 4x longer than with 64 threads."""
    import torch
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 64))
    yield


def pytest_collection_modifyitems(config, items):
    """Order of the GPU suite: the multi-process data-parallel tests first (their rank processes share the box's one GPU with
    this process: run behind the full-size tests, whose 2.567 B-parameter model and activation pools stay cached here, the same
    test took 200 s instead of 13 s), the full-size parity tests last.  Stable within each class: file order is kept."""
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return 0 if "dp_gpu" in name else (2 if "fullsize" in name else 1)
    items.sort(key=rank)


@pytest.fixture(scope="session")
def golden_host():
    with open(os.path.join(GOLDEN, "golden_host.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_tensors():
    import torch
    return torch.load(os.path.join(GOLDEN, "golden_tensors.pt"), map_location="cpu", weights_only=True)
