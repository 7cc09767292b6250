"""BASELINE config 4 in small: Fermi-Hubbard ladder, U(1) x U(1) charges (N, 2Sz), MPO bond dimension 10.
(1) the hand-built MPO against an explicit Jordan-Wigner exact diagonalisation, (2) two-site DMRG against the
per-sweep energies of the reference's FermiHubbardModel run (golden)."""
import numpy as np
import pytest

from helpers import golden
from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
from tenpy_amd.models.hubbard import hubbard_ladder_mpo, hubbard_ops, spinful_fermion_leg
from tenpy_amd.networks.mps import MPS


def _ed_ground_energy(Lx, t, U):
    o = hubbard_ops()
    N = 2 * Lx
    JW = o['JW']

    def string_op(ops):
        r = np.eye(1)
        for l in range(N):
            r = np.kron(r, ops.get(l, np.eye(4)))
        return r

    def c(s, spin):
        ops = {l: JW for l in range(s)}
        ops[s] = o['Cu'] if spin == 0 else o['Cd']
        return string_op(ops)
    bonds = []
    for x in range(Lx):
        bonds.append((2 * x, 2 * x + 1))
        if x + 1 < Lx:
            bonds += [(2 * x, 2 * x + 2), (2 * x + 1, 2 * x + 3)]
    H = sum(U * string_op({s: o['NuNd']}) for s in range(N))
    for i, j in bonds:
        for spin in (0, 1):
            ci, cj = c(i, spin), c(j, spin)
            H = H - t * (ci.T @ cj + cj.T @ ci)
    dn = sum(string_op({s: o['Ntot']}) for s in range(N)).diagonal()
    ds = sum(string_op({s: o['Nu'] - o['Nd']}) for s in range(N)).diagonal()
    idx = np.where((np.abs(dn - N) < 1e-9) & (np.abs(ds) < 1e-9))[0]
    return np.linalg.eigvalsh(H[np.ix_(idx, idx)])[0]


def test_hubbard_ladder_vs_ed(backend):
    Lx = 2
    H = hubbard_ladder_mpo(Lx, 1., 8., 0.)
    assert max(H.chi) == 10
    _, p = spinful_fermion_leg()
    psi = MPS.from_product_state([p] * (2 * Lx), [1, 2] * Lx)
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': 64, 'svd_min': 1e-12}})
    for _ in range(3):
        eng.sweep()
    assert abs(eng.sweep_stats['E'][-1] - _ed_ground_energy(Lx, 1., 8.)) < 1e-11


@pytest.mark.parametrize("name", ['hubbard_ladder_2x3', 'hubbard_ladder_2x4'])
def test_hubbard_ladder_vs_reference(backend, name):
    rec = [r for r in golden('hubbard.pkl') if r['name'] == name][0]
    if backend == 'mock' and rec['Lx'] > 3:
        pytest.skip("larger ladder only on the GPU")
    Lx = rec['Lx']
    H = hubbard_ladder_mpo(Lx, rec['t'], rec['U'], rec['mu'])
    assert max(H.chi) == rec['D_mpo']
    _, p = spinful_fermion_leg()
    psi = MPS.from_product_state([p] * (2 * Lx), [1, 2] * Lx)
    eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1e-10}})
    for s in range(len(rec['E_sweeps'])):
        eng.sweep()
        # the two MPOs differ by a gauge of the virtual index, so the Lanczos trajectories are not identical:
        # early sweeps agree to ~1e-6, converged sweeps to 1e-9
        tol = 1e-5 if s < 2 else 1e-9
        assert abs(eng.sweep_stats['E'][-1] - rec['E_sweeps'][s]) <= tol * abs(rec['E_sweeps'][s]), (s, eng.sweep_stats['E'][-1], rec['E_sweeps'][s])
    assert eng.sweep_stats['max_chi'][-1] == rec['chi_sweeps'][-1]
    np.testing.assert_allclose(psi.entanglement_entropy(), rec['S_ent'], rtol=0, atol=1e-6)
