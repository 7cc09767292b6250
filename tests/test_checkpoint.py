"""Checkpoint / resume of a DMRG run through pickle: device Arrays, legs, pipes and the MPS survive the
host round trip bit for bit, and a resumed run continues like the uninterrupted one."""
import os
import pickle

import numpy as np

from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
from tenpy_amd.networks.mps import MPS

OPTS = {'trunc_params': {'chi_max': 16, 'svd_min': 1.e-10}, 'lanczos_params': {}}


def _fresh(L=10):
    H = xxz_chain_mpo(L, 1., 0.8, 0.1)
    _, p = spin_half_leg('Sz')
    return MPS.from_product_state([p] * L, [1, 0] * (L // 2)), H


def test_array_pickle_round_trip(backend):
    psi, H = _fresh()
    eng = TwoSiteDMRGEngine(psi, H, OPTS)
    eng.sweep()
    th = psi.get_theta(4, n=2).combine_legs([['vL', 'p0'], ['p1', 'vR']])
    th2 = pickle.loads(pickle.dumps(th, protocol=4))
    th2.test_sanity()
    assert th2.get_leg_labels() == th.get_leg_labels()
    np.testing.assert_array_equal(th2._qdata, th._qdata)
    np.testing.assert_array_equal(th2.to_ndarray(), th.to_ndarray())
    for a, b in zip(th2.legs, th.legs):
        a.test_equal(b)
    assert abs(npc.norm(th2 - th)) == 0.


def test_resume_continues_the_run(backend, tmp_path):
    psi, H = _fresh()
    ref = TwoSiteDMRGEngine(psi, H, OPTS)
    for _ in range(4):
        ref.sweep()
    psi2, H2 = _fresh()
    eng = TwoSiteDMRGEngine(psi2, H2, OPTS)
    eng.sweep()
    eng.sweep()
    fn = os.path.join(str(tmp_path), 'ckpt.pkl')
    eng.save_checkpoint(fn)
    del eng, psi2
    res = TwoSiteDMRGEngine.from_checkpoint(fn, H2, OPTS)
    assert res.sweeps == 2 and len(res.sweep_stats['E']) == 2
    res.sweep()
    res.sweep()
    np.testing.assert_allclose(res.sweep_stats['E'], ref.sweep_stats['E'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(res.update_stats['E_total'], ref.update_stats['E_total'], rtol=1e-11, atol=1e-11)
    for i in range(1, res.psi.L):
        np.testing.assert_allclose(res.psi.get_SL(i), ref.psi.get_SL(i), rtol=0, atol=1e-10)
