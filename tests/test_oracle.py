"""Pin the CPU oracle (oracle/npc_oracle.py) against golden vectors generated from the reference."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import golden  # noqa: E402
from oracle import npc_oracle as orc  # noqa: E402


def oleg(d):
    return orc.OLeg(d['slices'], d['charges'], d['qconj'], d['mod'])


def otensor(d):
    return orc.OTensor([oleg(x) for x in d['legs']], d['qtotal'], d['qdata'], d['blocks'])


def check(t, d, tol=1e-13):
    for leg, ld in zip(t.legs, d['legs']):
        np.testing.assert_array_equal(leg.slices, ld['slices'])
        np.testing.assert_array_equal(leg.charges, ld['charges'])
        assert leg.qconj == ld['qconj']
    np.testing.assert_array_equal(t.qtotal, d['qtotal'])
    gq = np.asarray(d['qdata']).reshape(-1, t.rank)
    ts = t.sorted()
    order = np.lexsort(gq.T) if len(gq) > 1 else np.arange(len(gq))
    np.testing.assert_array_equal(ts.qdata, gq[order])
    if d['qdata_sorted']:
        np.testing.assert_array_equal(t.qdata, gq)     # the oracle also produces the reference's ORDER
    scale = max(np.max(np.abs(d['dense'])), 1e-300)
    np.testing.assert_allclose(t.to_dense(), d['dense'], rtol=0, atol=tol * scale * 10)


def idx(t_dict, labels):
    return [t_dict['labels'].index(l) if isinstance(l, str) else int(l) for l in labels]


def test_oracle_tensordot_inner():
    n = 0
    for rec in golden('tensordot.pkl'):
        a, b = otensor(rec['a']), otensor(rec['b'])
        if rec['op'] == 'tensordot':
            r = orc.tensordot(a, b, idx(rec['a'], rec['axes'][0]), idx(rec['b'], rec['axes'][1]))
            check(r, rec['res'])
        elif rec['op'] == 'outer':
            check(orc.tensordot(a, b, [], []), rec['res'])
        elif rec['op'] == 'inner':
            if rec.get('axes', 'range') == 'labels':
                conj_lab = [l[:-1] if l.endswith('*') else l + '*' for l in rec['a']['labels']]
                b = orc.transpose(b, [rec['b']['labels'].index(l) for l in conj_lab])
            v = orc.inner(a, b, rec['do_conj'])
            assert abs(v - rec['res']) <= 1e-13 * max(1., abs(rec['res'])) * 300
        n += 1
    assert n > 30


def test_oracle_combine_split():
    for rec in golden('reshape.pkl'):
        a = otensor(rec['a'])
        if rec['op'] == 'combine':
            groups = [idx(rec['a'], g) for g in rec['combine_legs']]
            if rec['new_axes'] is not None:
                continue        # explicit new_axes is bookkeeping of the product, not of the oracle
            qconjs = [a.legs[g[0]].qconj for g in groups]
            c, _ = orc.combine_legs(a, groups, qconjs)
            check(c, rec['res'])
        elif rec['op'] == 'transpose':
            check(orc.transpose(a, rec['perm']), rec['res'])


def test_oracle_linalg():
    for rec in golden('linalg.pkl'):
        if rec['op'] == 'svd':
            d = rec['a']
            a = otensor(d)
            blocked = all(len({tuple(c) for c in l['charges'].tolist()}) == len(l['charges']) for l in d['legs'])
            if not blocked:
                # block it first like npc.svd does (as_completely_blocked = 1-leg pipes)
                a, _ = orc.combine_legs(a, [[0], [1]], [a.legs[0].qconj, a.legs[1].qconj])
            U, S, VH = orc.svd(a)
            np.testing.assert_allclose(np.sort(S), np.sort(rec['S']), rtol=0, atol=1e-12 * np.max(rec['S']))
            rec_a = orc.tensordot(orc.OTensor(U.legs, U.qtotal, U.qdata, [u * s for u, s in zip(U.blocks, np.split(S, np.cumsum([b.shape[1] for b in U.blocks])[:-1]))]), VH, [1], [0])
            np.testing.assert_allclose(rec_a.to_dense(), a.to_dense(), atol=1e-12)
            if blocked:
                np.testing.assert_allclose(S, rec['S'], rtol=0, atol=1e-12 * np.max(rec['S']))   # same block order
                np.testing.assert_array_equal(VH.legs[0].charges, rec['VH']['legs'][0]['charges'])
        elif rec['op'] == 'axpy':
            z = orc.axpy(otensor(rec['a']), rec['prefactor'], otensor(rec['b']))
            check(z, rec['res'])
            assert abs(orc.norm(z) - rec['norm']) < 1e-13 * rec['norm']


def test_oracle_truncate():
    for rec in golden('truncate.pkl'):
        o = rec['options']
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            mask, norm_new, eps = orc.truncate(rec['S'], chi_max=o.get('chi_max', 100), chi_min=o.get('chi_min'),
                                               degeneracy_tol=o.get('degeneracy_tol'), svd_min=o.get('svd_min', 1e-14),
                                               trunc_cut=o.get('trunc_cut', 1e-14))
        np.testing.assert_array_equal(mask, rec['mask'])
        assert abs(norm_new - rec['norm_new']) < 1e-15 and abs(eps - rec['eps']) < 1e-18


def test_oracle_lanczos():
    for rec in golden('lanczos.pkl'):
        H, psi0 = otensor(rec['H']), otensor(rec['psi0'])
        o = rec['options']
        if 'N_cache' in o:
            continue    # cache size only changes how the result vector is re-assembled
        E0, psi, N = orc.lanczos_gs(lambda v: orc.tensordot(H, v, [1], [0]), psi0, N_min=o.get('N_min', 2),
                                    N_max=o.get('N_max', 20), P_tol=o.get('P_tol', 1e-14))
        assert N == rec['N'] and abs(E0 - rec['E0']) < 1e-12 * max(1., abs(rec['E0']))
        assert abs(abs(np.vdot(rec['psi']['dense'], psi.to_dense())) - 1.) < 1e-10
