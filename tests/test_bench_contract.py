"""bench.py end to end on the numpy emulation of the device (tiny chain): the JSON line must keep the driver's contract
(keys, types, the `roofline` and `cpu_baseline` objects) and the engine wiring bench.py relies on must not rot.  The
numbers mean nothing here -- the kernels are emulated; on the MI355X the same code path produces the round's bench line."""
import contextlib
import io
import json
import sys

import pytest


def test_bench_json_contract(monkeypatch):
    import mock_device
    mock_device.install(monkeypatch)
    import torch
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda *a, **k: None)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--L', '12', '--chi', '16', '--steps', '2', '--warmup', '1', '--cpu-sample-bonds', '1'])
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1, "exactly ONE JSON line on stdout"
    out = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in out, k
    assert out['n_gpus'] == 1 and out['steps'] == 2 and out['warmup'] == 1
    assert out['higher_is_better'] is False and out['vs_baseline'] is None and out['dtype'] == 'f64'
    assert out['unit'] == 's/sweep' and out['value'] > 0 and abs(out['ms_per_step'] - 1e3 * out['value']) < 1e-6
    assert 'workload' in out['config'] and 'model' not in out['config']
    r = out['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] > 0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    c = out['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['value'] > 0
    # parity at scale rides on the same line ("report parity at scale in the bench line")
    assert out['matvec_max_rel_err'] < 1e-10 and out['sv_max_rel_err'] < 1e-10 and out['E0_rel_err'] < 1e-10
    assert 'energy_err' in out and 'roofline_gemm' in out
    assert 'svd' in r['kernel'] and r['launches'] > 0
    assert abs(out['E'] - (-5.142090632841)) < 1e-6          # XXZ Jz=1, L=12 ground state energy (exact: -5.1420906328)


def _bench_worker(rank, world, port, ret):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      TPA_BENCH_BACKEND='gloo')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    try:
        from _pytest.monkeypatch import MonkeyPatch
        import mock_device
        mp = MonkeyPatch()
        mock_device.install(mp)
        import torch
        mp.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
        mp.setattr(torch.cuda, 'set_device', lambda *a, **k: None)
        sys.argv = ['bench.py', '--gpus', str(world), '--L', '12', '--chi', '16', '--steps', '1', '--warmup', '1'] + os.environ.get('TPA_TEST_BENCH_ARGS', '').split()
        import bench
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        ret[rank] = buf.getvalue()
    except Exception:  # pragma: no cover
        import traceback
        ret[rank] = 'FAIL: ' + traceback.format_exc()


def test_bench_two_ranks_gloo():
    """The N>1 control flow of bench.py (rendezvous, sharded engine, barriers, max over ranks, rank 0 prints ONE line)
    with world_size 2 over gloo on the emulated device."""
    import os
    import torch.multiprocessing as tmp
    world = 2
    port = 29700 + (os.getpid() % 200)
    mgr = tmp.Manager()
    ret = mgr.dict()
    tmp.spawn(_bench_worker, args=(world, port, ret), nprocs=world, join=True)
    assert not any(str(ret.get(r, 'FAIL')).startswith('FAIL') for r in range(world)), dict(ret)
    lines = [l for l in ret[0].splitlines() if l.strip()]
    assert len(lines) == 1 and ret[1].strip() == ''
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and 'cpu_baseline' not in out
    assert 'sharded' in out['config']['parallelism'] and out['value'] > 0
    assert abs(out['E'] - (-5.142090632841)) < 1e-6


def test_bench_two_ranks_gloo_tebd(monkeypatch):
    """``bench.py --config tebd1024 --gpus 2`` (tiny chain, emulated device): the bond-sharded TEBD engine under the bench's N > 1
    control flow; the line reports the state every rank holds."""
    import os
    import torch.multiprocessing as tmp
    monkeypatch.setenv('TPA_TEST_BENCH_ARGS', '--config tebd1024 --no-cpu-baseline')
    world = 2
    port = 29300 + (os.getpid() % 200)
    mgr = tmp.Manager()
    ret = mgr.dict()
    tmp.spawn(_bench_worker, args=(world, port, ret), nprocs=world, join=True)
    assert not any(str(ret.get(r, 'FAIL')).startswith('FAIL') for r in range(world)), dict(ret)
    out = json.loads([l for l in ret[0].splitlines() if l.strip()][0])
    assert out['n_gpus'] == 2 and out['unit'] == 's/step' and 'dealt over 2 GPUs' in out['config']['parallelism']
    assert out['S_mid_entropy'] > 0 and out['value'] > 0


def test_smoke_runs_on_the_emulated_device(monkeypatch, capsys):
    """__graft_entry__.smoke() (DMRG energy vs the exact value, matvec and SVD vs the oracle) with the device entry points
    emulated: guards the wiring the driver runs on the MI355X at round end."""
    import mock_device
    mock_device.install(monkeypatch)
    import torch
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda *a, **k: None)
    import __graft_entry__ as g
    g.smoke()
    assert 'smoke ok' in capsys.readouterr().out
