"""The stopping rule of the device Jacobi SVD, pinned on the CPU with a numpy emulation of the iteration
(tests/jacobi_emulation.py), and the same matrices through ``npc.svd`` on both backends.

The fixture ``golden/svd_rankdef.npz`` holds blocks from a single-site DMRG run with the subspace expansion: exactly
rank-deficient (zero columns), so a null row has no orthogonal complement to park its rounding noise in and shrinks by
~eps per sweep.  Without the null-row cut the iteration hangs once |row|^2 is denormal (seen on the MI355X:
"no convergence in 80 sweeps")."""
import os

import numpy as np
import pytest

import jacobi_emulation as je
from tenpy_amd.linalg import np_conserved as npc

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'svd_rankdef.npz')


def _mats():
    with np.load(FIX) as z:
        return [z[k] for k in sorted(z.files)]


def test_rule_terminates_on_exactly_rank_deficient_blocks():
    for A in _mats():
        sweeps, W = je.jacobi(A)
        assert 0 < sweeps <= 12
        s = np.sort(np.linalg.norm(W, axis=1))[::-1]
        ref = np.linalg.svd(A, compute_uv=False)
        assert np.abs(s - ref).max() <= 1e-14 * ref.max()


def test_rule_without_cut_hangs(monkeypatch):
    """Documents the failure mode the cut removes (so that nobody 'simplifies' it away)."""
    monkeypatch.setattr(je, 'NULL_ROW_CUT', 0.)
    with np.errstate(over='ignore'):
        hung = [je.jacobi(A, max_sweeps=40)[0] < 0 for A in _mats()]
    assert any(hung)


def test_rule_unchanged_on_generic_blocks():
    rng = np.random.RandomState(11)
    for m, n, r in ((24, 24, 24), (20, 33, 9), (40, 17, 17)):
        A = rng.standard_normal((m, r)) @ np.diag(np.logspace(0, -12, r)) @ rng.standard_normal((r, n))
        sweeps, W = je.jacobi(A)
        assert 0 < sweeps <= 20
        s = np.sort(np.linalg.norm(W, axis=1))[::-1]
        ref = np.linalg.svd(A, compute_uv=False)
        assert np.abs(s - ref[:len(s)]).max() <= 1e-13 * ref.max()


def test_npc_svd_of_rank_deficient_blocks(backend):
    from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
    ch = ChargeInfo([1])
    for A in _mats():
        m, n = A.shape
        a = npc.Array.from_ndarray(A, [LegCharge.from_qflat(ch, np.zeros((m, 1), int)), LegCharge.from_qflat(ch, np.zeros((n, 1), int), -1)])
        U, S, VH = npc.svd(a)
        ref = np.linalg.svd(A, compute_uv=False)
        np.testing.assert_allclose(np.sort(S)[::-1], ref[:len(S)], rtol=0, atol=1e-13 * ref.max())
        rec = (U.to_ndarray() * S) @ VH.to_ndarray()
        np.testing.assert_allclose(rec, A, rtol=0, atol=1e-13 * ref.max())


def test_predicted_convergence_saves_the_verification_sweep():
    """`tpa_svd_set_algorithm` bit 10 (off by default): stop after a sweep without rotations of scaled cosine > 1e-7."""
    rng = np.random.RandomState(3)
    saved = 0
    for m, n, r in ((48, 48, 48), (40, 90, 25), (64, 64, 30)):
        A = rng.standard_normal((m, r)) @ np.diag(np.logspace(0, -10, r)) @ rng.standard_normal((r, n))
        s0, W0 = je.jacobi(A)
        s1, W1 = je.jacobi(A, predict=1e-7)
        assert 0 < s1 <= s0
        saved += s0 - s1
        ref = np.linalg.svd(A, compute_uv=False)
        for W in (W0, W1):
            nr = np.linalg.norm(W, axis=1)
            assert np.abs(np.sort(nr)[::-1] - ref[:len(nr)]).max() <= 1e-13 * ref[0]
            big = nr > 1e-6 * np.linalg.norm(A)
            Wn = W[big] / nr[big, None]
            assert np.abs(Wn @ Wn.T - np.eye(big.sum())).max() < 1e-12
    assert saved >= 2


def test_npc_svd_with_predicted_convergence(backend):
    """Predicted convergence (default since round 2; `tpa_svd_set_algorithm(1024)` switches it OFF): the device iteration may
    stop after a sweep without big rotations.  Results must be the same as with the verification sweep (singular values,
    reconstruction, orthogonality), with no more sweeps."""
    from tenpy_amd import _lib
    from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
    rng = np.random.RandomState(21)
    ch = ChargeInfo([1])
    mats = []
    for m, n, r in ((96, 96, 50), (70, 130, 70), (20, 24, 11)):
        mats.append(rng.standard_normal((m, r)) @ np.diag(np.logspace(0, -9, r)) @ rng.standard_normal((r, n)))
    lib = _lib.load()
    results = {}
    try:
        for alg in (0, 1024):
            lib.tpa_svd_set_algorithm(alg)
            npc.svd_stats.update(calls=0, sweeps=0, max_block=0)
            out = []
            for A in mats:
                m, n = A.shape
                a = npc.Array.from_ndarray(A, [LegCharge.from_qflat(ch, np.zeros((m, 1), int)), LegCharge.from_qflat(ch, np.zeros((n, 1), int), -1)])
                U, S, VH = npc.svd(a)
                out.append((U.to_ndarray(), S, VH.to_ndarray()))
            results[alg] = (out, npc.svd_stats['sweeps'])
    finally:
        lib.tpa_svd_set_algorithm(0)
    assert results[0][1] <= results[1024][1]
    for A, (U0, S0, V0), (U1, S1, V1) in zip(mats, results[0][0], results[1024][0]):
        ref = np.linalg.svd(A, compute_uv=False)
        for U, S, V in ((U0, S0, V0), (U1, S1, V1)):
            np.testing.assert_allclose(np.sort(S)[::-1], ref[:len(S)], rtol=0, atol=1e-13 * ref[0])
            np.testing.assert_allclose((U * S) @ V, A, rtol=0, atol=1e-12 * ref[0])
            big = S > 1e-6 * ref[0]
            np.testing.assert_allclose(U[:, big].T @ U[:, big], np.eye(big.sum()), rtol=0, atol=1e-11)
            np.testing.assert_allclose(V[big] @ V[big].T, np.eye(big.sum()), rtol=0, atol=1e-11)


def test_isometries_on_graded_spectrum(backend, monkeypatch):
    """with the default absolute floor (1e-6) the vectors of sigma > 1e-6 sigma_max are orthonormal to machine
    precision; with ``SVD_ABS_FLOOR = 0`` EVERY returned vector is (graded spectrum over 13 decades)."""
    from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
    rng = np.random.RandomState(4)
    n = 160
    qu, _ = np.linalg.qr(rng.standard_normal((n, n)))
    qv, _ = np.linalg.qr(rng.standard_normal((n, n)))
    sig = np.logspace(0, -13, n)
    A = (qu * sig) @ qv.T
    ch = ChargeInfo([1])
    a = npc.Array.from_ndarray(A, [LegCharge.from_qflat(ch, np.zeros((n, 1), int)), LegCharge.from_qflat(ch, np.zeros((n, 1), int), -1)])
    for floor, cut in ((npc.SVD_ABS_FLOOR, 1e-6), (0., 1e-12)):
        monkeypatch.setattr(npc, 'SVD_ABS_FLOOR', floor)
        npc.svd_engine_floor = True          # the engines' opt-in (one call); without it the generic floor 0 applies
        U, S, VH = npc.svd(a)
        assert npc.svd_engine_floor is False
        u, v = U.to_ndarray(), VH.to_ndarray()
        np.testing.assert_allclose(np.sort(S)[::-1][:n], sig[:len(S)], rtol=0, atol=1e-13)
        keep = S > cut * S.max()
        assert keep.sum() >= 70
        np.testing.assert_allclose(u[:, keep].T @ u[:, keep], np.eye(keep.sum()), rtol=0, atol=2e-12)
        np.testing.assert_allclose(v[keep] @ v[keep].T, np.eye(keep.sum()), rtol=0, atol=2e-12)


def test_generic_svd_calls_run_without_the_floor(backend, monkeypatch):
    """Round 4: ``npc.svd`` called by anything but the DMRG / TEBD drivers (no hint, no ``svd_engine_floor``) runs the purely relative
    stopping rule: the floor handed to the device is 0 and nothing is post-processed; a marked call gets the engines' floor."""
    from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
    rng = np.random.RandomState(5)
    A = rng.standard_normal((40, 40))
    ch = ChargeInfo([1])
    a = npc.Array.from_ndarray(A, [LegCharge.from_qflat(ch, np.zeros((40, 1), int)), LegCharge.from_qflat(ch, np.zeros((40, 1), int), -1)])
    npc.svd(a)
    assert npc._svd_floor_now[0] == npc.SVD_ABS_FLOOR_GENERIC == 0.
    npc.svd_engine_floor = True
    npc.svd(a)
    assert npc._svd_floor_now[0] == npc.SVD_ABS_FLOOR
    npc.svd_hint = (('test', 0), 'R')
    npc.svd(a)
    assert npc._svd_floor_now[0] == npc.SVD_ABS_FLOOR and npc.svd_hint is None


def test_predicted_convergence_sees_pairs_below_the_floor():
    """Round 5 (found on the MI355X by tests/test_svd_configs_gpu.py): the "no big rotation -> stop" rule of rounds 2-4 measured a rotation
    against the FLOOR of the stopping rule, so a pair of rows below the floor was never big and the iteration could end on cosines of
    O(0.1) among such rows once the large rows were done -- harmless after a pivoted QR, wrong on a warm / sketch start of a block graded
    down to rounding level.  The numpy emulation of the device iteration (same predicate, operation for operation) on such a start:
    the old rule leaves a defect the first-order clean-up cannot repair, the corrected one (`svd_big_rotation`: every pair against its
    OWN stopping rule) a defect of ~1e-5 that three Loewdin iterations take to rounding level."""
    import jacobi_emulation as je
    rng = np.random.RandomState(1020)
    m = n = 120
    r = 60
    u, _ = np.linalg.qr(rng.standard_normal((m, r)))
    v, _ = np.linalg.qr(rng.standard_normal((n, r)))
    A = (u * np.logspace(0, -14.5, r)) @ v.T
    Bq = np.linalg.svd(A)[2][:r]                       # the basis of the previous visit ...
    k1, k2 = rng.standard_normal((m, m)) / np.sqrt(m), rng.standard_normal((n, n)) / np.sqrt(n)
    X = A + 1e-9 * ((k1 - k1.T) @ A + A @ (k2 - k2.T))   # ... and the state after a drift of 1e-9
    W = Bq @ X.T                                        # what the plain warm start hands to the iteration
    res = {}
    for new in (False, True):
        je.NEW_BIG_RULE = new
        try:
            sweeps, Wr = je.jacobi(W, rho=1e-4, predict=1e-7)
        finally:
            je.NEW_BIG_RULE = True
        s = np.linalg.norm(Wr, axis=1)
        keep = s > 1e-14 * s.max()
        Vn = Wr[keep] / s[keep, None]
        res[new] = (sweeps, float(np.abs(Vn @ Vn.T - np.eye(keep.sum())).max()))
    assert res[False][1] > 1e-3, res          # the flaw: the iteration stopped on large cosines below the floor
    assert res[True][1] < 1e-4 and res[True][0] > res[False][0], res
    T = Vn.copy()                              # the clean-up the product runs afterwards: V <- (3 - V V^T) V / 2
    for _ in range(3):
        T = 1.5 * T - 0.5 * (T @ T.T) @ T
    assert np.abs(T @ T.T - np.eye(len(T))).max() < 1e-13
