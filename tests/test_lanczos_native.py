"""``LanczosGroundState.run`` through ONE C-ABI call (``tpa_lanczos_run`` + ``tpa_krylov_combine``: the matvec replayed as a launch
program of cached plans, the recurrence with device-resident scalars, the reference's host logic in a callback) against the
step-by-step Python loop over the same kernels: same iteration count N, same E0, same vector (incl. its sign) for the reference's
default stopping rule (krylov_based.py:678-700), a forced N_min = N_max, an energy shift, both forms of the effective Hamiltonian
(fused ``LHeff . theta . RHeff`` and factored) and real / complex vectors."""
import numpy as np
import pytest

from tenpy_amd.algorithms import mps_common
from tenpy_amd.linalg import krylov_based as kb
from tenpy_amd.linalg import np_conserved as npc
from test_heff import _engine


def _overlap(a, b):
    return npc.inner(a, b, axes='range', do_conj=True)


@pytest.mark.parametrize("model", ['xxz', 'hubbard'])
@pytest.mark.parametrize("factored", [False, True])
@pytest.mark.parametrize("options", [{}, {'N_min': 8, 'N_max': 8}, {'E_shift': -3.5, 'N_max': 12}, {'E_tol': 1e-6, 'P_tol': 1e-8}],
                         ids=['default', 'forced8', 'shift', 'loose'])
def test_native_run_equals_python_loop(backend, monkeypatch, model, factored, options):
    eng = _engine(model)
    L = eng.psi.L
    ran_native = 0
    for i0 in (1, L // 2 - 1, L - 3):
        tensors = (eng.env.get_LP(i0), eng.env.get_RP(i0 + 1), eng.H.get_W(i0), eng.H.get_W(i0 + 1))
        H = mps_common.TwoSiteH(None, i0, tensors=tensors, factored=factored)
        if factored and not H.factored:
            pytest.skip("MPO blocks are not single numbers: no factored form")
        theta = H.combine_theta(eng.psi.get_theta(i0, n=2))
        res = {}
        for native in (True, False):
            monkeypatch.setattr(kb, 'NATIVE', native)
            runs = kb.stats['runs']
            lz = kb.LanczosGroundState(H, theta, dict(options))
            if native:
                ran_native += lz._native_program() is not None
            res[native] = lz.run()
            assert kb.stats['runs'] == runs + 1
        (E1, v1, N1), (E0, v0, N0) = res[True], res[False]
        assert N1 == N0
        assert abs(E1 - E0) <= 1e-12 * max(1., abs(E0))
        assert v1.get_leg_labels() == v0.get_leg_labels()
        assert abs(npc.norm(v1) - 1.) < 1e-12
        assert abs(_overlap(v0, v1) - 1.) < 1e-10          # same vector, same sign
    assert ran_native > 0, "the native path must apply to the bonds of a converged state"


def test_native_run_complex_and_small_norm(backend, monkeypatch):
    eng = _engine('xxz')
    i0 = eng.psi.L // 2 - 1
    tensors = (eng.env.get_LP(i0), eng.env.get_RP(i0 + 1), eng.H.get_W(i0), eng.H.get_W(i0 + 1))
    H = mps_common.TwoSiteH(None, i0, tensors=tensors, factored=False)
    theta = H.combine_theta(eng.psi.get_theta(i0, n=2))
    zt = theta.astype(np.complex128) * np.exp(0.3j)
    res = {}
    for native in (True, False):
        monkeypatch.setattr(kb, 'NATIVE', native)
        res[native] = kb.LanczosGroundState(H, zt, {'N_min': 4, 'N_max': 10}).run()
    # a complex vector with a real operator: mixed dtypes, the operator declines the launch program -> both go the Python way
    assert res[True][2] == res[False][2] and abs(res[True][0] - res[False][0]) < 1e-12
    monkeypatch.setattr(kb, 'NATIVE', True)
    with pytest.raises(ValueError):
        kb.LanczosGroundState(H, theta * 1e-300, {}).run()
