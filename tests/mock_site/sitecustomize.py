"""TEST INFRASTRUCTURE ONLY (tests/test_bench_contract.py::test_bench_gpus_flag_launches_ranks): a fresh Python process started with
``PYTHONPATH=tests/mock_site`` and ``TPA_TEST_MOCK_DEVICE=1`` runs the host logic on the numpy emulation of the C-ABI device entry
points (tests/mock_device.py), so that ``python bench.py --gpus 2`` -- which launches its own ranks through torch.distributed.run
-- can be exercised end to end in the CPU container.  Nothing under tenpy_amd/ or bench.py knows about this file."""
import os
import sys

if os.environ.get('TPA_TEST_MOCK_DEVICE') == '1':
    # never rebuild the library from inside this hook: the compiler driver starts Python children, which would come through here
    # again and wait for the build lock their parent holds (round 5: a header edit made every such process hang)
    os.environ['TPA_NO_AUTOBUILD'] = '1'
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path[:0] = [_root, os.path.join(_root, 'tests')]
    from _pytest.monkeypatch import MonkeyPatch
    import mock_device
    _mp = MonkeyPatch()
    mock_device.install(_mp)
    import torch
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
