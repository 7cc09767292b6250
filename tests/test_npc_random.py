"""Randomised property tests in the style of the reference's tests/test_np_conserved.py (dense numpy as the oracle,
fixed seeds): every result is compared with the same operation on ``to_ndarray()`` and must pass ``test_sanity``
(Appendix A invariants: qdata dtype / order, block shapes, charge rule).  Runs on the emulated device and on the GPU."""
import itertools

import numpy as np
import pytest

from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge


def _leg(rng, ch, n, qconj):
    q = np.stack([rng.randint(0, m, size=n) if m > 1 else rng.randint(-2, 3, size=n) for m in ch.mod], axis=1) \
        if ch.qnumber else np.zeros((n, 0), int)
    leg = LegCharge.from_qflat(ch, q, qconj)
    return leg.bunch()[1] if rng.rand() < 0.7 else leg


def _rand(rng, legs, cplx, qtotal=None):
    def f(shape):
        x = rng.standard_normal(shape)
        return x + 1.j * rng.standard_normal(shape) if cplx else x
    return npc.Array.from_func(f, legs, dtype=np.complex128 if cplx else np.float64, qtotal=qtotal)


CASES = [([1], False), ([2], True), ([1, 3], False), ([], True)]


@pytest.mark.parametrize("mod,cplx", CASES)
def test_tensordot_inner_outer_random(backend, mod, cplx):
    rng = np.random.RandomState(100 + len(mod) + int(cplx))
    ch = ChargeInfo(mod)
    for trial in range(6):
        la, lb, lc, ld = (_leg(rng, ch, n, q) for n, q in ((5, 1), (4, -1), (6, 1), (3, -1)))
        a = _rand(rng, [la, lb, lc], cplx).iset_leg_labels(['a', 'b', 'c'])
        b = _rand(rng, [lc.conj(), ld, lb.conj()], cplx and trial % 2 == 0).iset_leg_labels(['c*', 'd', 'b*'])
        A, B = a.to_ndarray(), b.to_ndarray()
        r = npc.tensordot(a, b, axes=(['c', 'b'], ['c*', 'b*']))
        r.test_sanity()
        np.testing.assert_allclose(r.to_ndarray(), np.tensordot(A, B, axes=([2, 1], [0, 2])), rtol=0, atol=1e-12)
        assert r.get_leg_labels() == ['a', 'd']
        r1 = npc.tensordot(a, b, axes=['c', 'c*'])
        r1.test_sanity()
        np.testing.assert_allclose(r1.to_ndarray(), np.tensordot(A, B, axes=([2], [0])), rtol=0, atol=1e-12)
        np.testing.assert_array_equal(a.to_ndarray(), A)           # operands untouched
        np.testing.assert_array_equal(b.to_ndarray(), B)
        c = _rand(rng, [la, lb, lc], cplx, qtotal=a.qtotal)
        np.testing.assert_allclose(npc.inner(a, c, axes='range', do_conj=True), np.vdot(A, c.to_ndarray()), rtol=0, atol=1e-11)
        o = npc.outer(_rand(rng, [la], cplx), _rand(rng, [ld], False))
        o.test_sanity()
        assert abs(npc.norm(a) - np.linalg.norm(A)) < 1e-12
        for o in (np.inf, 1, 3, 0, -np.inf):
            assert abs(npc.norm(a, o) - np.linalg.norm(A.reshape(-1), o)) < 1e-11 * max(1., np.linalg.norm(A.reshape(-1), o))


@pytest.mark.parametrize("mod,cplx", CASES)
def test_reshape_random(backend, mod, cplx):
    rng = np.random.RandomState(200 + len(mod) + int(cplx))
    ch = ChargeInfo(mod)
    for trial in range(5):
        legs = [_leg(rng, ch, n, q) for n, q in ((4, 1), (3, -1), (5, 1), (2, -1))]
        a = _rand(rng, legs, cplx).iset_leg_labels(['a', 'b', 'c', 'd'])
        A = a.to_ndarray()
        perm = list(rng.permutation(4))
        t = a.transpose(perm)
        t.test_sanity()
        np.testing.assert_array_equal(t.to_ndarray(), A.transpose(perm))
        groups = [['a', 'c'], ['d', 'b']] if trial % 2 else [['b', 'a', 'd']]
        c = a.combine_legs(groups)
        c.test_sanity()
        s = c.split_legs()
        s.test_sanity()
        np.testing.assert_array_equal(s.transpose(['a', 'b', 'c', 'd']).to_ndarray(), A)      # bit-exact round trip
        sc = rng.standard_normal(legs[2].ind_len)
        np.testing.assert_allclose(a.scale_axis(sc, 'c').to_ndarray(), A * sc[None, None, :, None], rtol=0, atol=1e-14)
        i = int(rng.randint(legs[1].ind_len))
        np.testing.assert_array_equal(a.take_slice(i, 'b').to_ndarray(), A[:, i])
        mask = rng.rand(legs[0].ind_len) > 0.4
        mask[0] = True
        pr = a.copy(deep=True)
        pr.iproject(mask, 'a')
        pr.test_sanity()
        np.testing.assert_array_equal(pr.to_ndarray(), A[mask])
        b = _rand(rng, legs, False, qtotal=a.qtotal)
        np.testing.assert_allclose((a + b * 0.5 - a * 2.).to_ndarray(), A + 0.5 * b.to_ndarray() - 2. * A, rtol=0, atol=1e-13)


@pytest.mark.parametrize("mod,cplx", CASES)
def test_decompositions_random(backend, mod, cplx):
    rng = np.random.RandomState(300 + len(mod) + int(cplx))
    ch = ChargeInfo(mod)
    for trial in range(4):
        l0, l1 = _leg(rng, ch, 9, 1), _leg(rng, ch, 7, -1)
        a = _rand(rng, [l0, l1], cplx).iset_leg_labels(['l', 'r'])
        A = a.to_ndarray()
        if a.stored_blocks == 0:
            continue
        U, S, VH = npc.svd(a, inner_labels=['i', 'i*'])
        U.test_sanity()
        VH.test_sanity()
        np.testing.assert_allclose((U.to_ndarray() * S) @ VH.to_ndarray(), A, rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.sort(S)[::-1], np.linalg.svd(A, compute_uv=False)[:len(S)], rtol=0, atol=1e-12)
        Q, R = npc.qr(a, inner_labels=['i', 'i*'])
        Q.test_sanity()
        np.testing.assert_allclose(Q.to_ndarray() @ R.to_ndarray(), A, rtol=0, atol=1e-12)
        Qd = Q.to_ndarray()
        np.testing.assert_allclose(Qd.conj().T @ Qd, np.eye(Qd.shape[1]), rtol=0, atol=1e-12)
        h = _rand(rng, [l0, l0.conj()], cplx)
        h = h + h.conj().itranspose()
        Hd = h.to_ndarray()
        W, V = npc.eigh(h)
        Vd = V.to_ndarray()
        np.testing.assert_allclose(Vd @ np.diag(W) @ Vd.conj().T, Hd, rtol=0, atol=1e-11)
        np.testing.assert_allclose(np.sort(W), np.linalg.eigvalsh(Hd), rtol=0, atol=1e-11)
        import scipy.linalg
        e = npc.expm(h * 0.3)
        np.testing.assert_allclose(e.to_ndarray(), scipy.linalg.expm(0.3 * Hd), rtol=0, atol=1e-11)
        P = npc.pinv(a)
        np.testing.assert_allclose(P.to_ndarray(), np.linalg.pinv(A), rtol=0, atol=1e-10)
