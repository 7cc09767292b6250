"""Per-call parity with the REFERENCE at mid size (SURVEY 8(c) "extra fixtures"): operands on the legs of
real converged XXZ DMRG states at chi = 64 and chi = 512 (centre bond, fused (vL.p0) x (p1.vR) of size 128 / 1024, LHeff /
RHeff with 5 MPO indices), filled from a seeded RNG exactly as ``tests/golden/make_golden.py:gen_percall`` did on the reference
side; results compared through the fingerprints stored there (integer bookkeeping exact; norm, sum and 96 probed entries of
the stored blocks to 1e-12 relative; singular values to 1e-10 relative to the largest).  Calls: the two tensordots of
``TwoSiteH.matvec`` (mps_common.py:1336-1337), ``inner``, ``iadd_prefactor_other``, ``norm``, ``svd``, ``split_legs``,
``combine_legs``, ``transpose``."""
import numpy as np
import pytest

from helpers import golden, load_leg
from tenpy_amd.linalg import np_conserved as npc


def seeded_array(legs, seed, labels, cplx=False):
    r = np.random.RandomState(seed)

    def f(size):
        x = r.standard_normal(size)
        return x + 1.j * r.standard_normal(size) if cplx else x
    a = npc.Array.from_func(f, legs, dtype=np.complex128 if cplx else np.float64, qtotal=None, shape_kw='size')
    return a.iset_leg_labels(labels)


def check(a, fp, tol=1e-12):
    a = a.copy(deep=True)
    a.test_sanity()
    a.isort_qdata()
    a._repack()
    np.testing.assert_array_equal(a._qdata, fp['qdata'])
    np.testing.assert_array_equal(a.qtotal, fp['qtotal'])
    assert a.get_leg_labels() == fp['labels']
    blocks = a._data
    assert [tuple(b.shape) for b in blocks] == [tuple(s) for s in fp['shapes']]
    scale = fp['norm'] / np.sqrt(max(sum(b.size for b in blocks), 1))         # typical entry
    assert abs(npc.norm(a) - fp['norm']) <= tol * fp['norm']
    assert abs(sum(np.sum(b) for b in blocks) - fp['sum']) <= tol * fp['norm'] * np.sqrt(len(fp['probe_val']) + 1) * 1e3
    got = np.array([blocks[b].reshape(-1)[o] for b, o in zip(fp['probe_block'], fp['probe_pos'])])
    np.testing.assert_allclose(got, fp['probe_val'], rtol=0, atol=50 * tol * max(scale, np.abs(fp['probe_val']).max()))


@pytest.mark.parametrize("idx", [0, 1], ids=['chi64', 'chi512'])
def test_percall_vs_reference(backend, idx):
    _percall(golden('percall.pkl')[idx])


def test_percall_hubbard_chi256(backend):
    """BASELINE config 4 regime: legs of a real Fermi-Hubbard-ladder state (charges (N, Sz)) at chi = 256: 23 bond
    sectors, ~300 blocks per H_eff half, ~10^3 GEMMs per tensordot."""
    _percall(golden('percall_hubbard.pkl')[0])


@pytest.mark.gpu
def test_percall_hubbard_chi1024_structure():
    """The same charge sectors 4 times as wide (block structure of the chi = 1024 ladder, fused theta 4096 x 4096)."""
    from tenpy_amd import _lib
    _lib.require_gpu()
    npc.clear_device_caches()
    _percall(golden('percall_hubbard.pkl')[1])
    npc.clear_device_caches()


def test_percall_tebd_complex_chi64(backend):
    """BASELINE config 5 (complex128): one TEBD bond update on seeded complex operands, legs of a real quench state."""
    _percall_tebd(golden('percall_tebd.pkl')[0])


@pytest.mark.gpu
def test_percall_tebd_complex_chi1024_structure():
    """Bond sectors 16 times as wide: two ~800 x 800 complex blocks per theta (the block-SVD-bound case of config 5)."""
    from tenpy_amd import _lib
    _lib.require_gpu()
    npc.clear_device_caches()
    _percall_tebd(golden('percall_tebd.pkl')[1])
    npc.clear_device_caches()


def _percall_tebd(rec):
    theta = seeded_array([load_leg(l) for l in rec['legs_theta']], 31, rec['labels_theta'], cplx=True)
    gate = seeded_array([load_leg(l) for l in rec['legs_gate']], 32, rec['labels_gate'], cplx=True)
    check(theta, rec['operands']['theta'], tol=1e-15)
    check(gate, rec['operands']['gate'], tol=1e-15)
    t = npc.tensordot(gate, theta, axes=(['p0*', 'p1*'], ['p0', 'p1']))
    check(t, rec['gate_theta'])
    tc = t.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])
    check(tc, rec['combined'])
    assert abs(npc.norm(tc) - rec['norm']) <= 1e-13 * rec['norm']
    U, S, VH = npc.svd(tc, inner_labels=['vR', 'vL'])
    np.testing.assert_array_equal(U._qdata, rec['svd_U_qdata'])
    np.testing.assert_array_equal(VH._qdata, rec['svd_VH_qdata'])
    assert len(S) == len(rec['svd_S'])
    np.testing.assert_allclose(S, rec['svd_S'], rtol=0, atol=1e-10 * rec['svd_S'].max())
    rebuilt = npc.tensordot(U.scale_axis(S, 'vR'), VH, axes=['vR', 'vL'])
    assert npc.norm(rebuilt - tc) <= 1e-12 * rec['norm']
    assert abs(npc.inner(tc, tc, axes='range', do_conj=True) - rec['inner']) <= 1e-12 * abs(rec['inner'])
    Bn = npc.tensordot(theta.conj(), t, axes=(['vL*', 'p0*'], ['vL', 'p0']))
    check(Bn, rec['conj_contract'])


@pytest.mark.gpu
def test_percall_vs_reference_full_size():
    """BASELINE's full size: chi = 2048 Heisenberg centre bond (bond sectors [2, 24, 122, 334, 542, 542, 334, 122, 24, 2]; fused theta
    4096 x 4096 in blocks up to 1084 x 1084, LHeff / RHeff 131 MB each) against fingerprints the REFERENCE computed on the same seeded
    operands (``make_golden.py:gen_percall2048``).  GPU only: the numpy emulation would spend minutes in Python tile loops."""
    from tenpy_amd import _lib
    from tenpy_amd.linalg import np_conserved as npc
    _lib.require_gpu()
    npc.clear_device_caches()
    _percall(golden('percall2048.pkl')[0])
    npc.clear_device_caches()


def _percall(rec):
    LH = seeded_array([load_leg(l) for l in rec['legs_LHeff']], 11, rec['labels_LHeff'])
    RH = seeded_array([load_leg(l) for l in rec['legs_RHeff']], 12, rec['labels_RHeff'])
    th = seeded_array([load_leg(l) for l in rec['legs_theta']], 13, rec['labels_theta'])
    th2 = seeded_array([load_leg(l) for l in rec['legs_theta']], 14, rec['labels_theta'])
    for a, key in ((LH, 'LH'), (RH, 'RH'), (th, 'th')):          # the operands themselves are the reference's, bit for bit
        check(a, rec['operands'][key], tol=1e-15)
    t1 = npc.tensordot(LH, th, axes=['(vR.p0*)', '(vL.p0)'])
    check(t1, rec['step1'])
    t2 = npc.tensordot(t1, RH, axes=[['wR', '(p1.vR)'], ['wL', '(p1*.vL)']])
    check(t2, rec['step2'])
    ov = npc.inner(th, t2.replace_labels(['(vR*.p0)', '(p1.vL*)'], ['(vL.p0)', '(p1.vR)']), axes='labels', do_conj=True)
    assert abs(ov - rec['inner']) <= 1e-12 * rec['step2']['norm'] * rec['norm']
    w = th.copy(deep=True)
    w.iadd_prefactor_other(-0.375, th2)
    check(w, rec['axpy'])
    assert abs(npc.norm(th) - rec['norm']) <= 1e-13 * rec['norm']
    U, S, VH = npc.svd(th, inner_labels=['vR', 'vL'])
    np.testing.assert_array_equal(U._qdata, rec['svd_U_qdata'])
    np.testing.assert_array_equal(VH._qdata, rec['svd_VH_qdata'])
    assert len(S) == len(rec['svd_S'])
    np.testing.assert_allclose(S, rec['svd_S'], rtol=0, atol=1e-10 * rec['svd_S'].max())
    rebuilt = npc.tensordot(U.scale_axis(S, 'vR'), VH, axes=['vR', 'vL'])
    assert npc.norm(rebuilt - th) <= 1e-12 * rec['norm']
    sp = th.split_legs()
    check(sp, rec['split'], tol=1e-15)
    check(sp.combine_legs([['vL', 'p0'], ['p1', 'vR']], qconj=[+1, -1]), rec['recombined'], tol=1e-15)
    check(sp.transpose(['p1', 'vR', 'vL', 'p0']), rec['transposed'], tol=1e-15)
