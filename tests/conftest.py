import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


CPU_WORKERS = int(os.environ.get('TPA_TEST_WORKERS', '4'))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """Without a GPU (the build container) the suite is host logic on the numpy emulation of the device: independent, CPU-bound
    tests, 8.5 min one after the other.  If pytest-xdist is there and the caller did not choose, run them on ``CPU_WORKERS`` worker
    processes (``TPA_TEST_WORKERS=0`` or ``-n 0`` / ``-p no:xdist``: serial).  With a GPU nothing changes: one process, one device."""
    opt = config.option
    if CPU_WORKERS < 2 or not hasattr(opt, 'numprocesses') or opt.numprocesses is not None or getattr(opt, 'collectonly', False):
        return None
    if getattr(opt, 'usepdb', False) or os.path.exists('/dev/kfd') or os.environ.get('PYTEST_XDIST_WORKER'):
        return None
    opt.numprocesses = CPU_WORKERS
    if getattr(opt, 'dist', 'no') == 'no':
        opt.dist = 'load'
    if not getattr(opt, 'tx', None):
        opt.tx = ['popen'] * CPU_WORKERS
    return None


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
