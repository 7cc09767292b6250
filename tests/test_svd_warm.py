"""Warm-started block SVD (tenpy_amd/linalg/_svd_warm.py) and the Loewdin clean-up of the small singular vectors:
results against numpy's LAPACK SVD per charge block (reference semantics: np_conserved.py:3676-3760, svd_flat :4970),
on the emulated device and on the GPU."""
import numpy as np
import pytest

from tenpy_amd.linalg import _svd_warm
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge


def _blocked(rng, sizes_l, sizes_r, rank_frac=0.6, cplx=False, decay=12.):
    """Block-diagonal matrix with graded singular values (like a DMRG theta): sector q of the left leg pairs with sector q
    of the right leg."""
    ch = ChargeInfo([1])
    legL = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(sizes_l)]), np.arange(len(sizes_l))[:, None], 1)
    legR = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(sizes_r)]), np.arange(len(sizes_r))[:, None], -1)
    dense = np.zeros((sum(sizes_l), sum(sizes_r)), dtype=complex if cplx else float)
    for q, (m, n) in enumerate(zip(sizes_l, sizes_r)):
        r = max(1, int(rank_frac * min(m, n)))
        def rnd(*sh):
            x = rng.standard_normal(sh)
            return x + 1j * rng.standard_normal(sh) if cplx else x
        u, _ = np.linalg.qr(rnd(m, r))
        v, _ = np.linalg.qr(rnd(n, r))
        dense[legL.slices[q]:legL.slices[q + 1], legR.slices[q]:legR.slices[q + 1]] = (u * np.logspace(0, -decay, r)) @ v.conj().T
    return dense, legL, legR


def _check(a_dense, legL, legR, U, S, VH, tol=2e-13):
    Ud, Vd = U.to_ndarray(), VH.to_ndarray()
    rec = (Ud * S) @ Vd
    scale = np.abs(a_dense).max()
    assert np.abs(rec - a_dense).max() <= tol * scale
    # per block singular values
    off = 0
    for q in range(legL.block_number):
        blk = a_dense[legL.slices[q]:legL.slices[q + 1], legR.slices[q]:legR.slices[q + 1]]
        ref = np.linalg.svd(blk, compute_uv=False)
        k = min(blk.shape)
        np.testing.assert_allclose(S[off:off + k], ref, rtol=0, atol=1e-13 * ref.max())
        off += k
    keep = S > 1e-14 * S.max()
    assert np.abs(Ud[:, keep].conj().T @ Ud[:, keep] - np.eye(keep.sum())).max() < 1e-12
    assert np.abs(Vd[keep] @ Vd[keep].conj().T - np.eye(keep.sum())).max() < 1e-12


@pytest.mark.parametrize("sketch", [True, False])
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("side", ['R', 'L'])
def test_warm_start_matches_lapack(backend, side, cplx, sketch, monkeypatch):
    monkeypatch.setattr(_svd_warm, 'SKETCH', sketch)       # round 5: a stale basis is used as a sketch (off: the round-4 behaviour)
    rng = np.random.RandomState(5)
    sizes_l, sizes_r = [40, 70, 9, 33], [52, 70, 17, 20]
    dense, legL, legR = _blocked(rng, sizes_l, sizes_r, cplx=cplx)
    a = npc.Array.from_ndarray(dense, [legL, legR])
    _svd_warm.cache_clear()
    for k in _svd_warm.stats:
        _svd_warm.stats[k] = 0
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(a)
    _check(dense, legL, legR, U, S, VH)
    assert _svd_warm.stats['cold_calls'] == 1 and _svd_warm.stats['warm_calls'] == 0
    # (1) tiny perturbation inside the old row / column space: basis complete, no E handling
    d2 = dense.copy()
    for q in range(legL.block_number):
        sl = (slice(legL.slices[q], legL.slices[q + 1]), slice(legR.slices[q], legR.slices[q + 1]))
        blk = dense[sl]
        mix = np.eye(blk.shape[0]) + 1e-6 * rng.standard_normal((blk.shape[0],) * 2)
        mixr = np.eye(blk.shape[1]) + 1e-6 * rng.standard_normal((blk.shape[1],) * 2)
        d2[sl] = (mix @ blk) if side == 'R' else (blk @ mixr)
    a2 = npc.Array.from_ndarray(d2, [legL, legR])
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(a2)
    _check(d2, legL, legR, U, S, VH)
    assert _svd_warm.stats['warm_calls'] == 1 and _svd_warm.stats['e_handled'] == 0
    # (2) perturbation that leaves the span of the basis (new directions at 1e-7): stale basis -> cold path, still exact
    d3 = d2.copy()
    for q in range(legL.block_number):
        sl = (slice(legL.slices[q], legL.slices[q + 1]), slice(legR.slices[q], legR.slices[q + 1]))
        m, n = d2[sl].shape
        x = rng.standard_normal((m, 2)) @ rng.standard_normal((2, n))
        d3[sl] = d2[sl] + 1e-7 * x / np.abs(x).max()
    a3 = npc.Array.from_ndarray(d3, [legL, legR])
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(a3)
    _check(d3, legL, legR, U, S, VH)
    assert _svd_warm.stats['warm_calls'] == 1 and _svd_warm.stats['fb_stale'] > 0
    if sketch:      # the stale basis + 32 random rows still sketch the column space: no pivoted QR, no cold call
        assert _svd_warm.stats['sketch_calls'] == 1 and _svd_warm.stats['fallbacks'] == 0 and _svd_warm.stats['cold_calls'] == 1
    else:
        assert _svd_warm.stats['fallbacks'] == 1
        assert _svd_warm.cooldown.get('bond', 0) > 0       # the next visits do not even try
    _svd_warm.cooldown.clear()
    # the decomposition of d3 is the new basis: d3 again starts warm
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(a3)
    _check(d3, legL, legR, U, S, VH)
    assert _svd_warm.stats['warm_calls'] == 2
    # (3) a different matrix under the same key: the basis is useless -> falls back to the cold path, still exact
    d4, _, _ = _blocked(np.random.RandomState(77), sizes_l, sizes_r, rank_frac=1.0, cplx=cplx, decay=3.)
    a4 = npc.Array.from_ndarray(d4, [legL, legR])
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(a4)
    _check(d4, legL, legR, U, S, VH)
    # (with the sketch: the plain attempt misses by O(1), far beyond SKETCH_MAX_E -- a state that is being rebuilt, not drifting --
    # so the sketch is not even tried and the call goes cold)
    assert (_svd_warm.stats['sketch_calls'] == 1 and _svd_warm.stats['fallbacks'] == 1) if sketch else _svd_warm.stats['fallbacks'] == 2
    _svd_warm.cooldown.clear()
    # (4) other leg structure under the same key: no basis
    d5, l5, r5 = _blocked(rng, [30, 21], [30, 40], cplx=cplx)
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(npc.Array.from_ndarray(d5, [l5, r5]))
    _check(d5, l5, r5, U, S, VH)
    assert npc.svd_hint is None
    _svd_warm.cache_clear()


@pytest.mark.parametrize("side", ['R', 'L'])
def test_sketch_path(backend, side, monkeypatch):
    """Round 5: a STALE basis as a sketch of the column space (range finder + unpivoted QR + Jacobi on the small factor).  Every
    old vector tilted by 1e-8 and a handful of new directions: the plain warm attempt fails its residual test, the sketch passes
    it at rounding level and the result equals LAPACK's; when the rank outgrows basis + extra rows the residual test of the sketch
    sends the call to the cold path."""
    monkeypatch.setattr(_svd_warm, 'SKETCH', True)
    rng = np.random.RandomState(11)
    sizes_l, sizes_r = [150, 96, 40], [140, 120, 33]
    dense, legL, legR = _blocked(rng, sizes_l, sizes_r, rank_frac=0.5)
    _svd_warm.cache_clear()
    for k in list(_svd_warm.stats):
        _svd_warm.stats[k] = 0
    npc.svd_hint = ('bond', side)
    npc.svd(npc.Array.from_ndarray(dense, [legL, legR]))
    assert _svd_warm.stats['cold_calls'] == 1
    d2 = dense.copy()
    for q in range(legL.block_number):
        sl = (slice(legL.slices[q], legL.slices[q + 1]), slice(legR.slices[q], legR.slices[q + 1]))
        m, n = dense[sl].shape
        rotL, _ = np.linalg.qr(np.eye(m) + 1e-8 * rng.standard_normal((m, m)))
        rotR, _ = np.linalg.qr(np.eye(n) + 1e-8 * rng.standard_normal((n, n)))
        rotL, rotR = rotL * np.sign(np.diag(rotL)), rotR * np.sign(np.diag(rotR))       # (rotations NEAR the identity: qr may flip signs)
        x = rng.standard_normal((m, 5)) @ rng.standard_normal((5, n))
        d2[sl] = rotL @ dense[sl] @ rotR + 1e-9 * x / np.abs(x).max()
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(npc.Array.from_ndarray(d2, [legL, legR]))
    _check(d2, legL, legR, U, S, VH)
    st = _svd_warm.stats
    assert st['sketch_calls'] == 1 and st['warm_calls'] == 0 and st['cold_calls'] == 1 and st['fb_stale'] > 0
    assert st['sk_e_rel_last'] < 1e-14                     # the range finder is exact to rounding, not to the tilt
    # the result of the sketch call is the next basis: the same matrix again starts warm
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(npc.Array.from_ndarray(d2, [legL, legR]))
    _check(d2, legL, legR, U, S, VH)
    assert st['warm_calls'] == 1
    # rank growth beyond basis + extra rows: full-rank blocks under the same key -> residual test of the sketch fails -> cold
    d3, _, _ = _blocked(np.random.RandomState(2), sizes_l, sizes_r, rank_frac=1.0, decay=3.)
    monkeypatch.setattr(_svd_warm, 'SKETCH_MAX_E', 10.)      # (the gate on the plain attempt's miss would not even try)
    npc.svd_hint = ('bond', side)
    U, S, VH = npc.svd(npc.Array.from_ndarray(d3, [legL, legR]))
    _check(d3, legL, legR, U, S, VH)
    assert st.get('sk_residual', 0) == 1 and st['cold_calls'] == 2 and st['sketch_calls'] == 1
    _svd_warm.cache_clear()


def test_lowdin_rows(backend):
    from tenpy_amd.linalg import _device as dev
    rng = np.random.RandomState(3)
    for cplx in (False, True):
        dt = np.complex128 if cplx else np.float64
        q, _ = np.linalg.qr(rng.standard_normal((60, 24)) + (1j * rng.standard_normal((60, 24)) if cplx else 0))
        V = q.T.copy()                                   # 24 orthonormal rows of length 60
        V = V + 1e-5 * (rng.standard_normal(V.shape))    # defect ~1e-5
        flat = np.concatenate([np.zeros(7, dt), V.reshape(-1).astype(dt)])
        arena = dev.to_device(flat)
        _svd_warm.lowdin_rows(dt, arena, [7], [24], [60], [60], [1], iterations=2)
        out = dev.to_host(arena)[7:].reshape(24, 60)
        assert np.abs(out @ out.conj().T - np.eye(24)).max() < 1e-14
        assert np.abs(out - V).max() < 1e-4
        # the same vectors stored as columns (stride 1 between vectors)
        arena = dev.to_device(np.ascontiguousarray(V.T).reshape(-1).astype(dt))
        _svd_warm.lowdin_rows(dt, arena, [0], [24], [60], [1], [24], iterations=2)
        out = dev.to_host(arena).reshape(60, 24)
        assert np.abs(out.conj().T @ out - np.eye(24)).max() < 1e-14


def test_ordered_rows(backend):
    """Round 6: the ORDERED clean-up (``tpa_tri_lower_batch``) -- every vector is orthonormalised against the vectors BEFORE it, so the
    leading vectors move by the SQUARE of the defect only, however large their cosines with later vectors are (the symmetric step
    moves both vectors of a pair by half the cosine); the result equals Gram-Schmidt in that order = Q of a QR factorisation."""
    from tenpy_amd.linalg import _device as dev
    rng = np.random.RandomState(4)
    for cplx in (False, True):
        dt = np.complex128 if cplx else np.float64
        q, _ = np.linalg.qr(rng.standard_normal((60, 24)) + (1j * rng.standard_normal((60, 24)) if cplx else 0))
        V0 = q.T.copy()
        # later vectors tilted towards earlier ones by up to 3e-2 (what the stopping rule with the floor on the smaller row leaves
        # between a tiny row and a large one), unit norm again
        mix = np.tril(rng.standard_normal((24, 24)), -1) * 3e-2 / np.sqrt(24)
        V = V0 + mix @ V0
        V /= np.linalg.norm(V, axis=1)[:, None]
        want = np.linalg.qr(V.conj().T)[0].conj().T            # Gram-Schmidt of the rows in order (up to phases)
        want *= (np.sum(want * V.conj(), axis=1) / np.abs(np.sum(want * V.conj(), axis=1)))[:, None].conj() if cplx else np.sign(np.sum(want * V, axis=1))[:, None]
        arena = dev.to_device(np.concatenate([np.zeros(5, dt), V.reshape(-1).astype(dt)]))
        _svd_warm.ordered_rows(dt, arena, [5], [24], [60], [60], [1], iterations=4)
        out = dev.to_host(arena)[5:].reshape(24, 60)
        assert np.abs(out @ out.conj().T - np.eye(24)).max() < 1e-14
        assert np.abs(out - want).max() < 1e-13
        assert np.abs(out[0] - V[0]).max() < 1e-15            # the first vector is not touched at all
        # the same vectors stored as columns
        arena = dev.to_device(np.ascontiguousarray(V.T).reshape(-1).astype(dt))
        _svd_warm.ordered_rows(dt, arena, [0], [24], [60], [1], [24], iterations=4)
        out = dev.to_host(arena).reshape(60, 24)
        assert np.abs(out.conj().T @ out - np.eye(24)).max() < 1e-14
        assert np.abs(out.T - want).max() < 1e-13


def test_warm_cache_is_bounded_by_bytes_and_tied_to_its_owner(backend, monkeypatch):
    """the cache of warm-start bases is capped by device BYTES (LRU), keyed by a token that is never reused, and the
    bases of an engine disappear with the engine."""
    import gc
    from tenpy_amd.linalg import _svd_warm as sw
    from tenpy_amd.linalg import _device as dev

    class Owner:
        pass
    sw.cache_clear()
    a, b = Owner(), Owner()
    ta, tb = sw.owner_token(a), sw.owner_token(b)
    assert ta != tb and sw.owner_token(a) == ta
    arena = dev.zeros(1000, np.float64)            # 8000 bytes per basis
    mk = lambda: sw.Basis(arena, np.zeros(1, np.int64), np.ones(1, np.int64), np.ones(1, np.int64), [(0, 1)], np.float64)
    monkeypatch.setattr(sw, 'CACHE_MAX_BYTES', 3 * 8000)
    for i in range(5):
        sw.cache_put((ta, i), 'R', mk())
    assert len(sw._cache) == 3 and sw._cache_bytes[0] == 3 * 8000           # the two oldest were evicted
    assert sw.cache_get((ta, 0), 'R') is None and sw.cache_get((ta, 4), 'R') is not None
    sw.cache_put((tb, 0), 'L', mk())
    sw.cooldown[(ta, 4)] = 2
    del a
    gc.collect()
    assert all(k[0][0] != ta for k in sw._cache) and (ta, 4) not in sw.cooldown     # a's bases went with a
    assert sw.cache_get((tb, 0), 'L') is not None
    c = Owner()
    assert sw.owner_token(c) not in (ta, tb)        # tokens are never handed out twice
    sw.cache_clear()


@pytest.mark.gpu
@pytest.mark.parametrize("side", ['R', 'L'])
@pytest.mark.parametrize("sizes", [([40, 70, 9, 33], [52, 70, 17, 20]), ([146, 292, 77, 5], [146, 200, 160, 5]), ([1086, 329], [1068, 658])])
def test_native_svd_theta_against_the_python_route(side, sizes, monkeypatch):
    """``tpa_svd_theta`` (csrc/tpa_svd_theta.hip: the warm route and the basis store as native calls) against the Python-driven route of
    the same algorithm: same routes taken, singular values / reconstruction / isometries of both against LAPACK, and against each other."""
    from tenpy_amd import _lib
    _lib.require_gpu()
    npc.clear_device_caches()
    sizes_l, sizes_r = sizes
    res, routes = {}, {}
    for native in (True, False):
        monkeypatch.setattr(npc, 'SVD_THETA_NATIVE', native)
        rng = np.random.RandomState(11)
        dense, legL, legR = _blocked(rng, sizes_l, sizes_r, decay=14.)
        _svd_warm.cache_clear()
        for k in _svd_warm.stats:
            _svd_warm.stats[k] = 0
        a = npc.Array.from_ndarray(dense, [legL, legR])
        npc.svd_hint = ('bondN', side)
        U, S, VH = npc.svd(a)
        _check(dense, legL, legR, U, S, VH)
        out = []
        cur = dense
        for step in range(3):          # three warm generations: each starts from the basis the previous one stored
            d2 = cur.copy()
            for q in range(legL.block_number):
                sl = (slice(legL.slices[q], legL.slices[q + 1]), slice(legR.slices[q], legR.slices[q + 1]))
                blk = cur[sl]
                mix = np.eye(blk.shape[0]) + 1e-7 * rng.standard_normal((blk.shape[0],) * 2)
                mixr = np.eye(blk.shape[1]) + 1e-7 * rng.standard_normal((blk.shape[1],) * 2)
                d2[sl] = (mix @ blk) if side == 'R' else (blk @ mixr)
            a2 = npc.Array.from_ndarray(d2, [legL, legR])
            npc.svd_hint = ('bondN', side)
            U, S, VH = npc.svd(a2)
            _check(d2, legL, legR, U, S, VH)
            out.append((S.copy(), U.to_ndarray(), VH.to_ndarray()))
            cur = d2
        # (a generation whose residual lands just above E_TOL -- wide blocks of the largest case, side L: 1.5e-13 -- goes through the
        #  sketch route on BOTH routes: the residuals agree to the last digit)
        routes[native] = {k: _svd_warm.stats[k] for k in ('warm_calls', 'sketch_calls', 'cold_calls', 'fb_stale')}
        assert _svd_warm.stats['warm_calls'] >= 2 and _svd_warm.stats['warm_calls'] + _svd_warm.stats['sketch_calls'] == 3, dict(_svd_warm.stats)
        assert _svd_warm.stats['cold_calls'] == 1
        assert (_svd_warm.stats.get('native_calls', 0) == 3) == native
        # a stale basis: the native call reports it and the sketch route takes over
        d3 = cur.copy()
        for q in range(legL.block_number):
            sl = (slice(legL.slices[q], legL.slices[q + 1]), slice(legR.slices[q], legR.slices[q + 1]))
            m, n = cur[sl].shape
            x = rng.standard_normal((m, 2)) @ rng.standard_normal((2, n))
            d3[sl] = cur[sl] + 1e-7 * x / np.abs(x).max()
        npc.svd_hint = ('bondN', side)
        U, S, VH = npc.svd(npc.Array.from_ndarray(d3, [legL, legR]))
        _check(d3, legL, legR, U, S, VH)
        assert _svd_warm.stats['fb_stale'] > routes[native]['fb_stale'] and _svd_warm.stats['sketch_calls'] == routes[native]['sketch_calls'] + 1, dict(_svd_warm.stats)
        res[native] = out
        npc.clear_device_caches()
    assert routes[True] == routes[False], routes
    for (S1, U1, V1), (S0, U0, V0) in zip(res[True], res[False]):
        assert np.abs(S1 - S0).max() <= 1e-14 * S0.max()
        keep = S0 > 1e-9 * S0.max()          # (vectors of well separated values agree up to sign)
        assert np.abs(np.abs(np.sum(U1[:, keep] * U0[:, keep], axis=0)) - 1.).max() < 1e-6
