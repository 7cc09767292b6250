import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(params=["mock", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """'mock': host logic on the numpy emulation of the C-ABI device calls (CPU container);
    'gpu': the real HIP kernels on an MI355X."""
    from tenpy_amd.linalg import np_conserved as npc
    npc.clear_device_caches()
    if request.param == "mock":
        import mock_device
        mock_device.install(monkeypatch)
    else:
        from tenpy_amd import _lib
        _lib.require_gpu()
    yield request.param
    npc.clear_device_caches()
