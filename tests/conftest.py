import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import vpr_amd  # noqa: F401
    from vpr_amd import _lib
    _lib.lib()      # fail loudly if the HIP extension is missing on a GPU box
    return torch.device("cuda:0")


@pytest.fixture
def tune():
    """Set one of the library's A/B switches for the duration of a test: tune("VPR_KNN_VARIANT", 6).  (The library reads
    the VPR_* environment once at load, so monkeypatch.setenv would change nothing.)"""
    from vpr_amd import _lib
    old = {}

    def set_(name, value):
        if name not in old:
            old[name] = _lib.tuning_get(name)
        _lib.tuning_set(name, None if value is None else int(value))

    yield set_
    for name, value in old.items():
        _lib.tuning_set(name, value)
