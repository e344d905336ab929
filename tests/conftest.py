import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tspn():
    import tspn_mi355x
    return tspn_mi355x


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def _device_status_is_left_clean(request):
    """A GPU test that raises a device fault on purpose (or fails while one stands) must not poison the tests behind it:
    every later launch entry would return TSPN_EDEVICE."""
    yield
    if "gpu" not in request.keywords:
        return
    mod = sys.modules.get("tspn_mi355x")
    if mod is None:
        return
    for words in list(mod.ops._status_blocks.values()):
        words[mod._abi.STATUS_FAULT_INFO] = 0
        words[mod._abi.STATUS_FAULT] = 0
