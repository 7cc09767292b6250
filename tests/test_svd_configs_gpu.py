"""Block SVD parity per BASELINE configuration on GRADED, RANK-DEFICIENT blocks -- what a DMRG theta looks like (the per-call
goldens use full-rank random blocks) -- through the three paths a bond sees in a run: cold (rank-revealing pivoted QR + Jacobi), the
stale basis as a sketch (round 5: range finder + unpivoted QR + Jacobi on the small factor) and the warm start; every result against
LAPACK per charge block (reference semantics: np_conserved.py:3676-3760, svd_flat :4970).  Bar: singular values to 1e-13 of
sigma_max (north_star: 1e-10), reconstruction to rounding, isometries of everything DMRG keeps orthonormal to 1e-11."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (rows, columns) of the charge blocks of theta [(vL.p0), (p1.vR)], fraction of min(m, n) that is the numerical rank
    'xxz512': ([292, 230, 230, 108, 108, 26, 26], [292, 230, 230, 108, 108, 26, 26], 0.5),
    'heis2048': ([22, 143, 456, 870, 1072, 872, 461, 155, 32, 2], [22, 142, 455, 868, 1068, 869, 460, 154, 36, 7], 0.53),
    'hubbard1024': ([int(x) for x in np.r_[np.linspace(8, 300, 23), np.linspace(290, 6, 22)]],
                    [int(x) for x in np.r_[np.linspace(6, 290, 23), np.linspace(300, 8, 22)]], 0.5),
}


def _graded_blocks(rng, ms, ns, rank_frac, decades=14.5):
    out = []
    for m, n in zip(ms, ns):
        r = max(1, int(rank_frac * min(m, n)))
        u, _ = np.linalg.qr(rng.standard_normal((m, r)))
        v, _ = np.linalg.qr(rng.standard_normal((n, r)))
        out.append((u * np.logspace(0, -decades, r)) @ v.T)
    return out


def _array(npc, legL, legR, blocks):
    dense = np.zeros((legL.ind_len, legR.ind_len))
    for q, b in enumerate(blocks):
        dense[legL.slices[q]:legL.slices[q + 1], legR.slices[q]:legR.slices[q + 1]] = b
    return npc.Array.from_ndarray(dense, [legL, legR]), dense


def _check(what, blocks, legL, legR, dense, U, S, VH):
    """Errors of one decomposition (all relative to sigma_max = the 2-norm of the block matrix): reconstruction (max entry), singular
    values per block against LAPACK, isometry defects of everything above 1e-14 sigma_max."""
    Ud, Vd = U.to_ndarray(), VH.to_ndarray()
    smax = S.max()
    rec = np.abs((Ud * S) @ Vd - dense).max() / smax
    off, sv = 0, 0.
    for b in blocks:
        k = min(b.shape)
        ref = np.linalg.svd(b, compute_uv=False)
        sv = max(sv, np.abs(np.sort(S[off:off + k])[::-1] - ref).max() / smax)
        off += k
    keep = S > 1e-14 * smax
    iso_u = np.abs(Ud[:, keep].T @ Ud[:, keep] - np.eye(keep.sum())).max()
    iso_v = np.abs(Vd[keep] @ Vd[keep].T - np.eye(keep.sum())).max()
    msg = "%s: reconstruction %.1e, singular values %.1e, |U^T U - 1| %.1e, |V V^T - 1| %.1e" % (what, rec, sv, iso_u, iso_v)
    print(msg)
    # bars: north_star asks 1e-10 for the singular values; a backward-stable SVD of a 1000 x 1000 block reconstructs to ~1e-13 sigma_max
    assert rec <= 1e-12 and sv <= 1e-13 and iso_u < 1e-11 and iso_v < 1e-11, msg


@pytest.mark.parametrize("side", ['R', 'L'])
@pytest.mark.parametrize("config", sorted(CONFIGS))
def test_block_svd_cold_sketch_warm(config, side):
    from tenpy_amd import _lib
    _lib.require_gpu()
    from tenpy_amd.linalg import _svd_warm
    from tenpy_amd.linalg import np_conserved as npc
    from tenpy_amd.linalg.charges import ChargeInfo, LegCharge
    ms, ns, frac = CONFIGS[config]
    rng = np.random.RandomState(sum(ms))
    ch = ChargeInfo([1])
    legL = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(ms)]), np.arange(len(ms))[:, None], 1)
    legR = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(ns)]), np.arange(len(ns))[:, None], -1)
    blocks = _graded_blocks(rng, ms, ns, frac)
    _svd_warm.cache_clear()
    for k in list(_svd_warm.stats):
        _svd_warm.stats[k] = 0
    st = _svd_warm.stats
    key = ('test_svd_configs', config, side)
    # (1) first visit of the bond: no basis -> cold path
    a, dense = _array(npc, legL, legR, blocks)
    npc.svd_hint = (key, side)
    _check('cold', blocks, legL, legR, dense, *npc.svd(a))
    assert st['cold_calls'] == 1 and st['warm_calls'] == 0 and st['sketch_calls'] == 0
    # (2) the state drifted since (every singular vector tilted by ~1e-9, a few new directions): the old basis is stale, but sketches
    #     the column space -- no pivoted QR
    drifted = []
    for b in blocks:
        m, n = b.shape
        k1, k2 = rng.standard_normal((m, m)) / np.sqrt(m), rng.standard_normal((n, n)) / np.sqrt(n)
        x = rng.standard_normal((m, 3)) @ rng.standard_normal((3, n))
        drifted.append(b + 1e-9 * ((k1 - k1.T) @ b + b @ (k2 - k2.T)) + 1e-10 * np.linalg.norm(b) / np.linalg.norm(x) * x)
    a, dense = _array(npc, legL, legR, drifted)
    npc.svd_hint = (key, side)
    _check('sketch', drifted, legL, legR, dense, *npc.svd(a))
    assert st['sketch_calls'] == 1 and st['cold_calls'] == 1 and st.get('sk_residual', 0) == 0, dict(st)
    assert st['sk_e_rel_last'] < 1e-13
    # (3) the same wave function again (a converged state): plain warm start, no QR at all
    npc.svd_hint = (key, side)
    _check('warm', drifted, legL, legR, dense, *npc.svd(a))
    assert st['warm_calls'] == 1 and st['cold_calls'] == 1 and st['sketch_calls'] == 1, dict(st)
    # (4) ... and without a hint (a generic npc.svd: purely relative stopping rule, no floor, no clean-up)
    _check('generic (no hint, no floor)', drifted, legL, legR, dense, *npc.svd(a))
    _svd_warm.cache_clear()
