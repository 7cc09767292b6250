"""Fuzz of npc.eigh / eigh_batched on the device against LAPACK: sizes around the path switches (8-row generation < 96 <= 32-row blocks),
both dtypes, spectra that stress the iterations (zero matrix, multiples of the identity, exactly repeated eigenvalues, +/- pairs, negative
definite, 30 decades, rank one, nearly diagonal).  python scripts/eigh_fuzz.py [n_trials] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ch = ChargeInfo([1])


def spectrum(kind, n):
    if kind == 'zero':
        return np.zeros(n)
    if kind == 'ident':
        return np.full(n, 3.5)
    if kind == 'repeat':
        return np.repeat(rng.standard_normal(max(1, n // 7 + 1)), 7)[:n]
    if kind == 'pairs':
        h = np.abs(rng.standard_normal(n // 2 + 1)) + 0.1
        return np.concatenate([h, -h])[:n]
    if kind == 'negdef':
        return -np.logspace(0, -10, n)
    if kind == 'wide':
        return np.logspace(10, -20, n)
    if kind == 'rank1':
        w = np.zeros(n)
        w[0] = 2.
        return w
    return rng.standard_normal(n)


worst = {}
t0 = time.time()
for trial in range(n_trials):
    cplx = bool(rng.integers(2))
    nblk = int(rng.integers(1, 5))
    sizes = [int(rng.choice([1, 2, 7, 31, 33, 64, 95, 96, 97, 130, 200, 257, 400])) for _ in range(nblk)]
    kinds = [str(rng.choice(['zero', 'ident', 'repeat', 'pairs', 'negdef', 'wide', 'rank1', 'normal', 'neardiag'])) for _ in range(nblk)]
    leg = LegCharge.from_qind(ch, np.concatenate([[0], np.cumsum(sizes)]), np.arange(nblk).reshape(-1, 1))
    dense = np.zeros((sum(sizes), sum(sizes)), dtype=np.complex128 if cplx else np.float64)
    o = 0
    mats = []
    for n, kind in zip(sizes, kinds):
        if kind == 'neardiag':
            x = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
            h = np.diag(np.logspace(0, -12, n)) + 1e-6 * (x + x.conj().T)
        else:
            x = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
            q, _ = np.linalg.qr(x)
            h = (q * spectrum(kind, n)) @ q.conj().T
        h = 0.5 * (h + h.conj().T)
        dense[o:o + n, o:o + n] = h
        mats.append(h)
        o += n
    a = npc.Array.from_ndarray(dense, [leg, leg.conj()], cutoff=0.)
    for batched in (False, True):
        if batched:
            (w, v), (w_b, v_b) = npc.eigh_batched([a, a])
            assert np.array_equal(w, w_b)
        else:
            w, v = npc.eigh(a)
        vd = v.to_ndarray()
        o = 0
        for n, kind, h in zip(sizes, kinds, mats):
            sl = slice(o, o + n)
            o += n
            nrm = max(np.linalg.norm(h, 2), 1e-300)
            ref = np.linalg.eigvalsh(h)
            e_w = np.abs(w[sl] - ref).max() / nrm
            e_r = np.abs(h @ vd[sl, sl] - vd[sl, sl] * w[sl][None, :]).max() / nrm
            e_o = np.abs(vd[sl, sl].conj().T @ vd[sl, sl] - np.eye(n)).max()
            asc = bool(np.all(np.diff(w[sl]) >= 0))
            key = (kind, 'c' if cplx else 'r', '>=96' if max(sizes) >= 96 else '<96')
            cur = worst.get(key, (0., 0., 0., True))
            worst[key] = (max(cur[0], e_w), max(cur[1], e_r), max(cur[2], e_o), cur[3] and asc)
            if not (e_w < 1e-11 * max(n, 8) and e_r < 1e-11 * max(n, 8) and e_o < 1e-10 and asc):
                print("BAD trial %d batched %s cplx %s sizes %s kinds %s block n=%d kind=%s: w %.1e res %.1e orth %.1e asc %s" %
                      (trial, batched, cplx, sizes, kinds, n, kind, e_w, e_r, e_o, asc), flush=True)
print("%d trials in %.1f s; eigh_stats %s" % (n_trials, time.time() - t0, npc.eigh_stats))
for key in sorted(worst):
    print("%-10s %s %-5s  |w - w_lapack|/|A| %.1e  |Av - vw|/|A| %.1e  |v^H v - 1| %.1e  ascending %s" % (key + worst[key]))
