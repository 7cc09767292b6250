"""tenpy_amd.linalg.np_conserved vs golden results produced by the reference (tests/golden/make_golden.py).

Integer bookkeeping (legs, qtotal, _qdata and its order) must match bit-exactly; block data to 1e-13
relative.  Runs on the mock device (host logic, CPU) and on the real GPU (``-m gpu``)."""
import numpy as np
import pytest

from helpers import golden, load_array, load_leg, assert_array_matches, assert_leg_equal
from tenpy_amd.linalg import np_conserved as npc


def test_tensordot_golden(backend):
    n = 0
    for rec in golden('tensordot.pkl'):
        a, b = load_array(rec['a']), load_array(rec['b'])
        if rec['op'] == 'tensordot':
            a0, b0 = a.to_ndarray(), b.to_ndarray()
            r = npc.tensordot(a, b, axes=rec['axes'])
            assert_array_matches(r, rec['res'])
            assert r._qdata_sorted
            # operands untouched
            np.testing.assert_array_equal(a.to_ndarray(), a0)
            np.testing.assert_array_equal(b.to_ndarray(), b0)
        elif rec['op'] == 'outer':
            assert_array_matches(npc.outer(a, b), rec['res'])
        elif rec['op'] == 'inner':
            axes = rec.get('axes', 'range')
            v = npc.inner(a, b, axes=axes, do_conj=rec['do_conj'])
            assert abs(v - rec['res']) <= 1e-13 * max(1., abs(rec['res'])) * a.size
        n += 1
    assert n >= 30


def test_reshape_golden(backend):
    for rec in golden('reshape.pkl'):
        a = load_array(rec['a'])
        if rec['op'] == 'combine':
            c = a.combine_legs(rec['combine_legs'], new_axes=rec['new_axes'])
            assert_array_matches(c, rec['res'])
            s = c.split_legs()
            assert_array_matches(s, rec['split'])
        elif rec['op'] == 'transpose':
            assert_array_matches(a.transpose(rec['perm']), rec['res'])
        elif rec['op'] == 'scale_axis':
            assert_array_matches(a.scale_axis(rec['s'], rec['axis']), rec['res'])
        elif rec['op'] == 'project':
            p = a.copy(deep=True)
            p.iproject(rec['mask'], rec['axis'])
            assert_array_matches(p, rec['res'])


def test_linalg_golden(backend):
    for rec in golden('linalg.pkl'):
        a = load_array(rec['a'])
        if rec['op'] == 'svd':
            U, S, VH = npc.svd(a, inner_labels=['vR', 'vL'])
            # integer structure exact
            assert_array_matches(U, rec['U'], data=False)
            assert_array_matches(VH, rec['VH'], data=False)
            # singular values: 1e-10 relative to the largest (north_star), in fact ~1e-14
            np.testing.assert_allclose(S, rec['S'], rtol=0, atol=1e-12 * np.max(rec['S']))
            rec_a = npc.tensordot(U.scale_axis(S, 1), VH, axes=1)
            np.testing.assert_allclose(rec_a.to_ndarray(), rec['a']['dense'], rtol=0, atol=1e-12 * np.max(rec['S']))
            UU = npc.tensordot(U.conj(), U, axes=[[0], [0]]).to_ndarray()
            np.testing.assert_allclose(UU, np.eye(len(S)), atol=1e-12)
        elif rec['op'] == 'qr':
            Q, R = npc.qr(a, inner_labels=['q', 'r'], pos_diag_R=True)
            assert_array_matches(Q, rec['Q'], rtol=1e-11)
            assert_array_matches(R, rec['R'], rtol=1e-11)
        elif rec['op'] == 'eigh':
            W, V = npc.eigh(a)
            np.testing.assert_allclose(W, rec['W'], rtol=0, atol=1e-12 * np.max(np.abs(rec['W'])))
            assert_array_matches(V, rec['V'], data=False)
            ad, vd = a.to_ndarray(), V.to_ndarray()
            np.testing.assert_allclose(ad @ vd, vd * W, atol=1e-11)
        elif rec['op'] == 'axpy':
            b = load_array(rec['b'])
            z = a.copy(deep=True)
            z.iadd_prefactor_other(rec['prefactor'], b)
            assert_array_matches(z, rec['res'])
            assert abs(npc.norm(z) - rec['norm']) < 1e-13 * rec['norm']


def test_lanczos_golden(backend):
    from tenpy_amd.linalg.krylov_based import LanczosGroundState
    for rec in golden('lanczos.pkl'):
        H, psi0 = load_array(rec['H']), load_array(rec['psi0'])

        class Op:
            def matvec(self, v):
                return npc.tensordot(H, v, axes=['a*', 'a'])
        E0, psi, N = LanczosGroundState(Op(), psi0, dict(rec['options'])).run()
        assert N == rec['N']
        assert abs(E0 - rec['E0']) < 1e-12 * max(1., abs(rec['E0']))
        ref = rec['psi']['dense']
        ov = abs(np.vdot(ref, psi.to_ndarray()))
        assert abs(ov - 1.) < 1e-10


def test_krylov2_golden(backend):
    """LanczosEvolution and gram_schmidt vs the reference (tests/golden/make_golden.py:gen_krylov2)."""
    from tenpy_amd.linalg.krylov_based import LanczosEvolution, gram_schmidt
    for rec in golden('krylov2.pkl'):
        if rec['kind'] == 'evolution':
            H, psi0 = load_array(rec['H']), load_array(rec['psi0'])

            class Op:
                def matvec(self, v):
                    return npc.tensordot(H, v, axes=['a*', 'a'])
            psi, N = LanczosEvolution(Op(), psi0, dict(rec['options'])).run(rec['delta'], rec['normalize'])
            assert N == rec['N']
            ref = rec['psi']['dense']
            np.testing.assert_allclose(psi.to_ndarray(), ref, rtol=0, atol=1e-11 * max(1., np.abs(ref).max()))
            np.testing.assert_array_equal(psi0.to_ndarray(), rec['psi0']['dense'])      # start vector untouched
        else:
            vecs = [load_array(v) for v in rec['vecs']]
            res = gram_schmidt(vecs)
            assert len(res) == len(rec['res'])
            for v, r in zip(res, rec['res']):
                np.testing.assert_allclose(v.to_ndarray(), r['dense'], rtol=0, atol=1e-9)


def test_api2_golden(backend):
    """take_slice / concatenate / expm / pinv / polar / unary_blockwise / ones / Array.matvec vs the reference
    (tests/golden/make_golden.py:gen_api2): integer bookkeeping exact, data within tolerance."""
    seen = set()
    for rec in golden('api2.pkl'):
        op = rec['op']
        seen.add(op)
        if op == 'take_slice':
            a = load_array(rec['a'])
            assert_array_matches(a.take_slice(rec['indices'], rec['axes']), rec['res'])
        elif op == 'concatenate':
            arrs = [load_array(x) for x in rec['arrays']]
            res = npc.concatenate(arrs, axis=rec['axis'])
            assert not res._qdata_sorted
            assert_array_matches(res, rec['res'])
            for x, d in zip(arrs, rec['arrays']):
                np.testing.assert_array_equal(x.to_ndarray(), d['dense'])
        elif op == 'expm':
            a = load_array(rec['a'])
            res = npc.expm(a)
            ref = rec['res']['dense']
            assert_array_matches(res, rec['res'], data=False)
            # (1e-11: the ||a|| ~ 40 case is squared 7 times; scipy's Pade result carries an error of the same size)
            np.testing.assert_allclose(res.to_ndarray(), ref, rtol=0, atol=1e-11 * max(1., np.abs(ref).max()))
            np.testing.assert_array_equal(a.to_ndarray(), rec['a']['dense'])
        elif op == 'pinv':
            a = load_array(rec['a'])
            assert_array_matches(npc.pinv(a), rec['res'], rtol=1e-11)
        elif op == 'polar':
            a = load_array(rec['a'])
            u, p, s = npc.polar(a, left=rec['left'])
            assert_array_matches(u, rec['u'], rtol=1e-11)
            assert_array_matches(p, rec['p'], rtol=1e-11)
            np.testing.assert_allclose(np.sort(s), np.sort(rec['s']), rtol=1e-11)
        elif op == 'unary':
            a = load_array(rec['a'])
            assert_array_matches(a.unary_blockwise(getattr(np, rec['func'])), rec['res'])
            np.testing.assert_array_equal(a.to_ndarray(), rec['a']['dense'])
        elif op == 'ones':
            from helpers import load_leg
            legs = [load_leg(x) for x in rec['legs']]
            assert_array_matches(npc.ones(legs, qtotal=rec['qtotal']), rec['res'])
        elif op == 'matvec':
            a, v = load_array(rec['a']), load_array(rec['v'])
            assert_array_matches(a.matvec(v), rec['res'], rtol=1e-12)
    assert seen == {'take_slice', 'concatenate', 'expm', 'pinv', 'polar', 'unary', 'ones', 'matvec'}
    with pytest.raises(NotImplementedError):
        load_array(golden('api2.pkl')[0]['a']).unary_blockwise(lambda x: x + 1)
