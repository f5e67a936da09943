import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _torch_opens_the_device_first():
    """Tests hand torch device tensors to the library; torch must open the device before the library's own
    first HIP call does (the other order leaves torch without a device in this image), whatever test runs first."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
