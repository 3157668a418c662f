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


@pytest.fixture(scope="session")
def golden_host():
    with open(os.path.join(GOLDEN, "golden_host.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_tensors():
    import torch
    return torch.load(os.path.join(GOLDEN, "golden_tensors.pt"), map_location="cpu", weights_only=True)
