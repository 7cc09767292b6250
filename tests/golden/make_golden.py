"""Generate golden fixtures from the REFERENCE implementation (TeNPy, pure-Python path).

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    TENPY_NO_CYTHON=1 PYTHONPATH=/root/reference python tests/golden/make_golden.py

Outputs (committed): ``tests/golden/*.pkl`` -- plain dict / list / numpy content only (no tenpy classes),
so the tests can load them anywhere.  What is pinned:

* ``charges.pkl``    LegCharge.sort/bunch/project and LegPipe (slices, charges, q_map, q_map_slices)
* ``tensordot.pkl``  operands + results of npc.tensordot / inner / outer for several axes / dtypes
* ``reshape.pkl``    combine_legs / split_legs / transpose results
* ``linalg.pkl``     svd (S per block, U/VH qdata + legs), qr, eigh, iadd_prefactor_other with union of blocks, norm
* ``lanczos.pkl``    LanczosGroundState on a small TwoSiteH-like dense-backed operator: E0, N, alpha/beta
* ``truncate.pkl``   truncation.truncate on assorted spectra / options
* ``dmrg.pkl``       per-sweep energies, chi, centre-bond Schmidt values of reference two-site DMRG runs
                     (XXZ Sz-conserving L=16 chi=32, TFI L=32 chi=30 = examples/d_dmrg.py config,
                     TFI parity L=12) with Lanczos always used (max_N_for_ED=0, mixer off)
"""
import os
import pickle
import sys
import warnings

import numpy as np

sys.path.insert(0, '/root/reference')
os.environ.setdefault('TENPY_NO_CYTHON', '1')
import tenpy  # noqa: E402
import tenpy.linalg.np_conserved as npc  # noqa: E402
from tenpy.linalg import charges, krylov_based, truncation  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(20260925)


def dump_leg(leg):
    d = dict(slices=np.array(leg.slices), charges=np.array(leg.charges), qconj=int(leg.qconj),
             mod=np.array(leg.chinfo.mod), sorted=bool(leg.sorted), bunched=bool(leg.bunched))
    if isinstance(leg, charges.LegPipe):
        d['pipe'] = dict(legs=[dump_leg(l) for l in leg.legs], q_map=np.array(leg.q_map),
                         q_map_slices=np.array(leg.q_map_slices))
    return d


def dump_array(a):
    return dict(legs=[dump_leg(l) for l in a.legs], qtotal=np.array(a.qtotal), qdata=np.array(a._qdata),
                qdata_sorted=bool(a._qdata_sorted), blocks=[np.array(b) for b in a._data], labels=list(a._labels),
                dtype=str(a.dtype), dense=a.to_ndarray())


def rand_leg(chinfo, n, qconj=1, bunch=True):
    qflat = []
    for mod in chinfo.mod:
        if mod > 1:
            qflat.append(rng.randint(0, mod, size=n))
        else:
            r = max(2, n // 4)
            qflat.append(rng.randint(-r, r + 1, size=n))
    qflat = np.array(qflat, dtype=np.int64).T.reshape(n, chinfo.qnumber)
    leg = charges.LegCharge.from_qflat(chinfo, qflat, qconj)
    return leg.bunch()[1] if bunch else leg


def rand_array(legs, qtotal=None, cplx=False, labels=None):
    def f(size):
        x = rng.standard_normal(size)
        if cplx:
            x = x + 1.j * rng.standard_normal(size)
        return x
    a = npc.Array.from_func(f, legs, dtype=np.complex128 if cplx else np.float64, qtotal=qtotal, shape_kw='size')
    if labels is not None:
        a.iset_leg_labels(labels)
    return a


def save(name, obj):
    with open(os.path.join(HERE, name), 'wb') as f:
        pickle.dump(obj, f, protocol=4)
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


# ---------------------------------------------------------------------------------------------------
def gen_charges():
    out = []
    for mod in ([1], [2], [1, 3], [3, 1, 2]):
        ch = charges.ChargeInfo(mod)
        for n in (1, 5, 12):
            leg = rand_leg(ch, n, qconj=rng.choice([-1, 1]), bunch=False)
            perm, sl = leg.sort(bunch=True)
            idx, bl = leg.bunch()
            mask = rng.rand(n) > 0.4
            if not mask.any():
                mask[0] = True
            map_qind, block_masks, pl = leg.project(mask)
            out.append(dict(kind='leg', leg=dump_leg(leg), sort_perm=np.array(perm), sorted=dump_leg(sl), bunch_idx=np.array(idx),
                            bunched=dump_leg(bl), mask=mask, map_qind=np.array(map_qind), projected=dump_leg(pl),
                            qflat=leg.to_qflat()))
        for shape in ((3, 4), (5, 2, 3), (1, 1), (6,)):
            legs = [rand_leg(ch, n, qconj=rng.choice([-1, 1])) for n in shape]
            for sort, bunch in ((True, True), (False, True), (True, False), (False, False)):
                for qconj in (1, -1):
                    pipe = charges.LegPipe(legs, qconj=qconj, sort=sort, bunch=bunch)
                    out.append(dict(kind='pipe', legs=[dump_leg(l) for l in legs], qconj=qconj, sort=sort, bunch=bunch,
                                    pipe=dump_leg(pipe),
                                    flat_map=[pipe.map_incoming_flat([rng.randint(0, l.ind_len) for l in legs]) for _ in range(0)]))
    save('charges.pkl', out)


def gen_tensordot():
    out = []
    for mod, cplx in (([1], False), ([1], True), ([2], False), ([1, 3], False), ([], False)):
        ch = charges.ChargeInfo(mod)
        la, lb, lc, ld, le = [rand_leg(ch, n) for n in (7, 6, 5, 4, 8)]
        a = rand_array([la, lb, lc.conj()], labels=['a', 'b', 'c'], cplx=cplx)
        qt = [1] * len(mod) if len(mod) else None
        b = rand_array([lc, lb.conj(), ld], qtotal=qt, labels=['c*', 'b*', 'd'], cplx=cplx)
        for axes in ((['c'], ['c*']), (['b', 'c'], ['b*', 'c*']), (['b'], ['b*']), ([2, 1], [0, 1])):
            r = npc.tensordot(a, b, axes=axes)
            out.append(dict(op='tensordot', a=dump_array(a), b=dump_array(b), axes=axes, res=dump_array(r)))
        r = npc.outer(a, b)
        out.append(dict(op='outer', a=dump_array(a), b=dump_array(b), res=dump_array(r)))
        a2 = rand_array([la, lb, lc.conj()], labels=['a', 'b', 'c'], cplx=cplx)
        out.append(dict(op='inner', a=dump_array(a), b=dump_array(a2), do_conj=True,
                        res=npc.inner(a, a2, axes='range', do_conj=True)))
        bc = rand_array([lc, lb.conj(), la.conj()], labels=['c*', 'b*', 'a*'], cplx=cplx)
        out.append(dict(op='inner', a=dump_array(a), b=dump_array(bc), do_conj=False, axes='labels',
                        res=npc.inner(a, bc, axes='labels', do_conj=False)))
        if cplx:
            # mixed real x complex
            ar = rand_array([la, lb, lc.conj()], labels=['a', 'b', 'c'])
            r = npc.tensordot(ar, b, axes=(['b', 'c'], ['b*', 'c*']))
            out.append(dict(op='tensordot', a=dump_array(ar), b=dump_array(b), axes=(['b', 'c'], ['b*', 'c*']), res=dump_array(r)))
    save('tensordot.pkl', out)


def gen_reshape():
    out = []
    for mod, cplx in (([1], False), ([2, 1], True)):
        ch = charges.ChargeInfo(mod)
        legs = [rand_leg(ch, n, qconj=q) for n, q in ((4, 1), (5, -1), (3, 1), (6, -1))]
        a = rand_array(legs, labels=['a', 'b', 'c', 'd'], cplx=cplx)
        for cl, new_axes in (([['a', 'b']], None), ([['a', 'c'], ['d', 'b']], None), ([['c', 'a']], [1]),
                             ([['b', 'd'], ['a']], [1, 0]), ([['a', 'b', 'c', 'd']], None)):
            c = a.combine_legs(cl, new_axes=new_axes)
            s = c.split_legs()
            out.append(dict(op='combine', a=dump_array(a), combine_legs=cl, new_axes=new_axes, res=dump_array(c),
                            split=dump_array(s)))
        for perm in ([3, 1, 0, 2], [1, 0, 2, 3]):
            t = a.transpose(perm)
            out.append(dict(op='transpose', a=dump_array(a), perm=perm, res=dump_array(t)))
        s = rng.standard_normal(legs[1].ind_len)
        out.append(dict(op='scale_axis', a=dump_array(a), s=s, axis=1, res=dump_array(a.scale_axis(s, 1))))
        mask = rng.rand(legs[3].ind_len) > 0.5
        mask[0] = True
        p = a.copy(deep=True)
        p.iproject(mask, 3)
        out.append(dict(op='project', a=dump_array(a), mask=mask, axis=3, res=dump_array(p)))
    save('reshape.pkl', out)


def gen_linalg():
    out = []
    for mod, cplx in (([1], False), ([1], True), ([3], False)):
        ch = charges.ChargeInfo(mod)
        for m, n in ((6, 6), (9, 5), (4, 11)):
            l0, l1 = rand_leg(ch, m), rand_leg(ch, n, qconj=-1)
            a = rand_array([l0, l1], labels=['L', 'R'], cplx=cplx, qtotal=[1] * len(mod))
            U, S, VH = npc.svd(a, inner_labels=['vR', 'vL'])
            out.append(dict(op='svd', a=dump_array(a), U=dump_array(U), S=np.array(S), VH=dump_array(VH)))
            Q, R = npc.qr(a, inner_labels=['q', 'r'], pos_diag_R=True)
            out.append(dict(op='qr', a=dump_array(a), Q=dump_array(Q), R=dump_array(R)))
        l0 = rand_leg(ch, 9)
        h = rand_array([l0, l0.conj()], labels=['p', 'p*'], cplx=cplx)
        h = h + h.conj().itranspose()
        W, V = npc.eigh(h)
        out.append(dict(op='eigh', a=dump_array(h), W=np.array(W), V=dump_array(V)))
        # axpy with different sparsity patterns
        la, lb = rand_leg(ch, 6), rand_leg(ch, 7, qconj=-1)
        x = rand_array([la, lb], cplx=cplx)
        y = rand_array([la, lb], cplx=cplx)
        x._data = x._data[::2]
        x._qdata = x._qdata[::2]
        y._data = y._data[1:]
        y._qdata = y._qdata[1:]
        z = x.copy(deep=True)
        z.iadd_prefactor_other(-0.7, y)
        out.append(dict(op='axpy', a=dump_array(x), b=dump_array(y), prefactor=-0.7, res=dump_array(z),
                        norm=npc.norm(z)))
    save('linalg.pkl', out)


def gen_truncate():
    out = []
    for n in (1, 5, 40):
        S = np.abs(rng.standard_normal(n)) * np.exp(-rng.rand(n) * 12)
        S = S / np.linalg.norm(S)
        for opts in ({'chi_max': 10, 'svd_min': 1e-8}, {'chi_max': 3, 'chi_min': 2}, {'chi_max': None, 'trunc_cut': 1e-3},
                     {'chi_max': 7, 'degeneracy_tol': 0.5, 'svd_min': None}, {}):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                mask, norm_new, err = truncation.truncate(S, dict(opts))
            out.append(dict(S=S, options=opts, mask=np.array(mask), norm_new=float(norm_new), eps=float(err.eps)))
    save('truncate.pkl', out)


def gen_dmrg():
    from tenpy.algorithms import dmrg
    from tenpy.models.tf_ising import TFIChain
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []

    def run(model, psi, chi, n_sweeps, name, extra):
        eng = dmrg.TwoSiteDMRGEngine(psi, model, {
            'mixer': None, 'combine': True, 'max_N_for_ED': 0,
            'trunc_params': {'chi_max': chi, 'svd_min': 1.e-10},
            'lanczos_params': {}, 'max_sweeps': n_sweeps})
        Es, chis = [], []
        for s in range(n_sweeps):
            eng.sweep()
            Es.append(float(eng.sweep_stats['E'][-1]) if len(eng.sweep_stats['E']) else float(eng.update_stats['E_total'][-1]))
            chis.append(int(max(psi.chi)))
        Es = [float(e) for e in Es]
        upd_E = [float(e) for e in eng.update_stats['E_total']]
        mid = psi.L // 2
        rec = dict(name=name, E_sweeps=Es, chi_sweeps=chis, E_updates=upd_E, N_lanczos=list(eng.update_stats['N_lanczos']),
                   S_mid=np.array(psi.get_SL(mid)), S_ent=np.array(psi.entanglement_entropy()), chi=chi, n_sweeps=n_sweeps,
                   E_mpo=float(np.real(model.H_MPO.expectation_value(psi))) if hasattr(model, 'H_MPO') else None)
        rec.update(extra)
        out.append(rec)
        print(name, Es[-1], chis[-1])

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 16
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        run(M, psi, 32, 6, 'xxz_L16_chi32', dict(L=L, Jxx=1., Jz=1., hz=0., conserve='Sz', init='neel_updown'))
        L = 12
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 0.5, 'hz': 0.1, 'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        run(M, psi, 20, 5, 'xxz_L12_chi20_hz', dict(L=L, Jxx=1., Jz=0.5, hz=0.1, conserve='Sz', init='neel_updown'))
        L = 32
        M = TFIChain({'L': L, 'J': 1., 'g': 1., 'bc_MPS': 'finite', 'conserve': None})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
        run(M, psi, 30, 5, 'tfi_L32_chi30', dict(L=L, J=1., g=1., conserve=None, init='all_up'))
        L = 12
        M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'finite', 'conserve': 'parity', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
        run(M, psi, 16, 5, 'tfi_parity_L12_chi16', dict(L=L, J=1., g=1.5, conserve='parity', init='all_up'))
    save('dmrg.pkl', out)


def gen_lanczos():
    """Lanczos on a random hermitian block matrix with Z2 charge, like tests/test_krylov_based.py:32."""
    from tenpy.linalg.sparse import FlatLinearOperator
    out = []
    ch = charges.ChargeInfo([2])
    for n, cplx in ((20, False), (30, True)):
        leg = charges.LegCharge.from_qflat(ch, rng.randint(0, 2, size=n).reshape(n, 1)).bunch()[1]
        H = rand_array([leg, leg.conj()], cplx=cplx, labels=['a', 'a*'])
        H = H + H.conj().itranspose()
        psi0 = rand_array([leg], qtotal=[rng.randint(2)], cplx=cplx, labels=['a'])

        class Op:
            def matvec(self, v):
                return npc.tensordot(H, v, axes=['a*', 'a'])
        for opts in ({}, {'N_min': 5, 'N_max': 5}, {'N_max': 40, 'N_cache': 3, 'P_tol': 1e-20}):
            E0, psi, N = krylov_based.LanczosGroundState(Op(), psi0, dict(opts)).run()
            out.append(dict(H=dump_array(H), psi0=dump_array(psi0), options=opts, E0=float(E0), N=int(N), psi=dump_array(psi)))
    save('lanczos.pkl', out)




def gen_tebd():
    """Real-time TEBD after a global quench (config 5 in small): TFI chain, parity conserved, order 2."""
    from tenpy.algorithms import tebd
    from tenpy.models.tf_ising import TFIChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for conserve in ('parity', None):
            L = 10
            M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'finite', 'conserve': conserve, 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
            eng = tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.05, 'N_steps': 1,
                                           'trunc_params': {'chi_max': 16, 'svd_min': 1.e-10}})
            S_t, chi_t, sz_t = [], [], []
            for step in range(12):
                eng.run()
                S_t.append(np.array(psi.entanglement_entropy()))
                chi_t.append(int(max(psi.chi)))
                sz_t.append(np.array(psi.expectation_value('Sigmaz')))
            psi2 = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
            eng2 = tebd.QRBasedTEBDEngine(psi2, M, {'order': 2, 'dt': 0.05, 'N_steps': 1, 'cbe_expand': 0.5,
                                                    'trunc_params': {'chi_max': 16, 'svd_min': 1.e-10}})
            S_qr, chi_qr = [], []
            for step in range(12):
                eng2.run()
                S_qr.append(np.array(psi2.entanglement_entropy()))
                chi_qr.append(int(max(psi2.chi)))
            out.append(dict(name='tfi_quench_L10_%s' % conserve, S_qr=np.array(S_qr), chi_qr=chi_qr, L=L, J=1., g=1.5, conserve=conserve, dt=0.05, chi=16,
                            S_t=np.array(S_t), chi_t=chi_t, sigmaz_t=np.array(sz_t), S_mid=np.array(psi.get_SL(L // 2)),
                            h_bond=[None if h is None else h.transpose(['p0', 'p1', 'p0*', 'p1*']).to_ndarray() for h in M.H_bond],
                            state_labels=list(M.lat.mps_sites()[0].state_labels.items())))
            print(out[-1]['name'], chi_t[-1], S_t[-1][L // 2 - 1])
    save('tebd.pkl', out)




def gen_qr_theta():
    """decompose_theta_qr_based on a two-site wave function of a small converged DMRG state, all flag combos."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 12
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        dmrg.run(psi, M, {'mixer': None, 'trunc_params': {'chi_max': 24, 'svd_min': 1e-10}, 'max_sweeps': 4})
        i0 = L // 2 - 1
        theta = psi.get_theta(i0, n=2).combine_legs([['vL', 'p0'], ['p1', 'vR']], qconj=[+1, -1])
        pert = rand_array(theta.legs, qtotal=theta.qtotal, labels=theta.get_leg_labels())
        theta = theta + 1e-3 * pert
        B_L, B_R = psi.get_B(i0, 'B'), psi.get_B(i0 + 1, 'B')
        for move_right in (True, False):
            for eig in (False, True):
                for both in (False, True):
                    T_Lc, S, T_Rc, form, err, renorm = truncation.decompose_theta_qr_based(
                        B_L.qtotal, B_R.qtotal, B_R.get_leg('vL'), theta, move_right, 0.1, 1, eig,
                        {'chi_max': 20, 'svd_min': 1e-10}, both, both)
                    out.append(dict(theta=dump_array(theta), old_qtotal_L=np.array(B_L.qtotal), old_qtotal_R=np.array(B_R.qtotal),
                                    old_bond_leg=dump_leg(B_R.get_leg('vL')), move_right=move_right, eig=eig, both=both,
                                    T_Lc=None if T_Lc is None else dump_array(T_Lc), T_Rc=None if T_Rc is None else dump_array(T_Rc),
                                    S=np.array(S), form=list(form), eps=float(err.eps), renorm=float(renorm)))
    save('qr_theta.pkl', out)


def gen_hubbard():
    """BASELINE config 4 in small: Fermi-Hubbard 2 x 4 ladder, charges (N, 2Sz), two-site DMRG chi=64."""
    from tenpy.algorithms import dmrg
    from tenpy.models.hubbard import FermiHubbardModel
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for Lx, chi in ((3, 40), (4, 64)):
            M = FermiHubbardModel({'lattice': 'Ladder', 'L': Lx, 't': 1., 'U': 8., 'mu': 0., 'cons_N': 'N', 'cons_Sz': 'Sz',
                                   'bc_MPS': 'finite', 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * Lx, bc='finite')
            eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                                  'trunc_params': {'chi_max': chi, 'svd_min': 1.e-10}})
            Es, chis = [], []
            for s in range(6):
                eng.sweep()
                Es.append(float(eng.update_stats['E_total'][-1]))
                chis.append(int(max(psi.chi)))
            out.append(dict(name='hubbard_ladder_2x%d' % Lx, Lx=Lx, chi=chi, t=1., U=8., mu=0., E_sweeps=Es, chi_sweeps=chis,
                            S_ent=np.array(psi.entanglement_entropy()), D_mpo=max(M.H_MPO.chi)))
            print(out[-1]['name'], Es[-1], chis[-1], 'MPO D', max(M.H_MPO.chi))
    save('hubbard.pkl', out)


def gen_krylov2():
    """LanczosEvolution (real / imaginary / complex time step, normalised or not, short cache) and gram_schmidt."""
    out = []
    ch = charges.ChargeInfo([2])
    r2 = np.random.RandomState(777)
    for n, cplx in ((20, False), (30, True)):
        leg = charges.LegCharge.from_qflat(ch, r2.randint(0, 2, size=n).reshape(n, 1)).bunch()[1]

        def rnd(legs, qtotal=None, labels=None):
            def f(size):
                x = r2.standard_normal(size)
                return x + 1.j * r2.standard_normal(size) if cplx else x
            a = npc.Array.from_func(f, legs, dtype=np.complex128 if cplx else np.float64, qtotal=qtotal, shape_kw='size')
            a.iset_leg_labels(labels)
            return a
        H = rnd([leg, leg.conj()], labels=['a', 'a*'])
        H = H + H.conj().itranspose()
        psi0 = rnd([leg], qtotal=[1], labels=['a'])

        class Op:
            def matvec(self, v):
                return npc.tensordot(H, v, axes=['a*', 'a'])
        for delta, opts, normalize in ((-0.1j, {}, None), (-0.05, {}, None), (0.02 - 0.03j, {'N_max': 30, 'N_cache': 3}, True),
                                       (-0.2j, {'N_min': 4, 'N_max': 4}, False)):
            psi, N = krylov_based.LanczosEvolution(Op(), psi0, dict(opts)).run(delta, normalize)
            out.append(dict(kind='evolution', H=dump_array(H), psi0=dump_array(psi0), options=opts, delta=delta, normalize=normalize,
                            N=int(N), psi=dump_array(psi)))
        vecs = [rnd([leg], qtotal=[1], labels=['a']) for _ in range(4)]
        vecs.insert(2, vecs[0] * 2. - vecs[1] * 0.5)       # linearly dependent: must be dropped
        dumped = [dump_array(v) for v in vecs]
        res = krylov_based.gram_schmidt([v.copy(deep=True) for v in vecs])
        out.append(dict(kind='gram_schmidt', vecs=dumped, res=[dump_array(v) for v in res]))
    save('krylov2.pkl', out)


def gen_api2():
    """More of the Array API: take_slice, concatenate, expm, pinv, polar, unary_blockwise, ones, Array.matvec."""
    out = []
    r3 = np.random.RandomState(4242)

    def rleg(ch, n, qconj=1):
        qflat = np.stack([r3.randint(0, m, size=n) if m > 1 else r3.randint(-1, 2, size=n) for m in ch.mod], axis=1)
        return charges.LegCharge.from_qflat(ch, qflat, qconj).bunch()[1]

    def rarr(legs, cplx, qtotal=None, labels=None):
        def f(size):
            x = r3.standard_normal(size)
            return x + 1.j * r3.standard_normal(size) if cplx else x
        a = npc.Array.from_func(f, legs, dtype=np.complex128 if cplx else np.float64, qtotal=qtotal, shape_kw='size')
        if labels is not None:
            a.iset_leg_labels(labels)
        return a
    for mod, cplx in (([1], False), ([3, 1], True)):
        ch = charges.ChargeInfo(mod)
        legs = [rleg(ch, n, q) for n, q in ((5, 1), (4, -1), (6, 1))]
        a = rarr(legs, cplx, labels=['a', 'b', 'c'])
        for idx, axes in (([2], ['b']), ([4, 0], ['a', 'c']), ([3], [2])):
            out.append(dict(op='take_slice', a=dump_array(a), indices=idx, axes=axes, res=dump_array(a.take_slice(idx, axes))))
        # concatenate along the middle leg: operands with different middle legs (one with opposite qconj)
        b = rarr([legs[0], rleg(ch, 3, -1), legs[2]], cplx, labels=['a', 'b', 'c'])
        c = rarr([legs[0], rleg(ch, 2, -1), legs[2]], False, labels=['x', 'y', 'z'])
        out.append(dict(op='concatenate', arrays=[dump_array(x) for x in (a, b, c)], axis='b',
                        res=dump_array(npc.concatenate([a, b, c], axis='b'))))
        out.append(dict(op='concatenate', arrays=[dump_array(x) for x in (b, a)], axis=1,
                        res=dump_array(npc.concatenate([b, a], axis=1))))
        # square matrices
        leg = rleg(ch, 9)
        m = rarr([leg, leg.conj()], cplx, labels=['p', 'p*'])
        out.append(dict(op='expm', a=dump_array(m), res=dump_array(npc.expm(m))))
        big = m * 7.3                                   # needs several squarings
        out.append(dict(op='expm', a=dump_array(big), res=dump_array(npc.expm(big))))
        pipe = charges.LegPipe([legs[0], legs[1].conj()], qconj=+1)
        mp = rarr([pipe, pipe.conj()], cplx, labels=['(a.b)', '(a*.b*)']) * 0.3
        out.append(dict(op='expm', a=dump_array(mp), res=dump_array(npc.expm(mp)), pipes=True))
        herm = (m + m.conj().itranspose()) * (-0.2j if cplx else -0.2)     # a TEBD-like gate  exp(-i dt H)
        out.append(dict(op='expm', a=dump_array(herm), res=dump_array(npc.expm(herm))))
        r = rarr([rleg(ch, 8), rleg(ch, 5, -1)], cplx, labels=['l', 'r'])
        out.append(dict(op='pinv', a=dump_array(r), res=dump_array(npc.pinv(r))))
        for left in (False, True):
            u, pm, sv = npc.polar(r, left=left)
            out.append(dict(op='polar', a=dump_array(r), left=left, u=dump_array(u), p=dump_array(pm), s=np.array(sv)))
        for fn in ('real', 'imag', 'abs', 'conj', 'angle'):
            out.append(dict(op='unary', a=dump_array(a), func=fn, res=dump_array(a.unary_blockwise(getattr(np, fn)))))
        out.append(dict(op='ones', legs=[dump_leg(l) for l in legs], qtotal=np.array(a.qtotal),
                        res=dump_array(npc.ones(legs, qtotal=a.qtotal))))
        v = rarr([leg], cplx, qtotal=ch.make_valid(np.ones(len(mod), dtype=int)), labels=['v'])
        out.append(dict(op='matvec', a=dump_array(m), v=dump_array(v), res=dump_array(m.matvec(v))))
    save('api2.pkl', out)



def probe_array(a, seed, n=96):
    """Size-independent fingerprint of a result Array: exact integer bookkeeping (qdata, block shapes), its 2-norm, the sum of
    its entries and the values at `n` seeded random positions of the STORED blocks (block index, flat offset in the block)."""
    a = a.copy(deep=True)
    a.isort_qdata()
    r = np.random.RandomState(seed)
    sizes = np.array([b.size for b in a._data], dtype=np.int64)
    blk = r.randint(0, len(sizes), size=n) if len(sizes) else np.zeros(0, int)
    pos = np.array([r.randint(0, sizes[b]) for b in blk], dtype=np.int64)
    vals = np.array([a._data[b].reshape(-1)[o] for b, o in zip(blk, pos)])
    return dict(qdata=np.array(a._qdata), shapes=[tuple(b.shape) for b in a._data], qtotal=np.array(a.qtotal),
                labels=list(a._labels), norm=float(npc.norm(a)), sum=complex(sum(np.sum(b) for b in a._data)),
                probe_block=blk, probe_pos=pos, probe_val=vals, dtype=str(a.dtype))


def seeded_array(legs, seed, cplx=False, labels=None):
    """Operand with every charge-allowed block filled from RandomState(seed): blocks are drawn in lexicographic qindex order
    (last leg most significant), each block C-ordered -- the test regenerates the identical tensor from the legs + seed."""
    r = np.random.RandomState(seed)

    def f(size):
        x = r.standard_normal(size)
        return x + 1.j * r.standard_normal(size) if cplx else x
    a = npc.Array.from_func(f, legs, dtype=np.complex128 if cplx else np.float64, qtotal=None, shape_kw='size')
    if labels is not None:
        a.iset_leg_labels(labels)
    return a


def _percall_record(eff, theta0, L, chi):
    """The hot-path calls of one bond update on seeded operands with the legs of `eff` / `theta0`; results as fingerprints."""
    LH = seeded_array(eff.LHeff.legs, 11, labels=eff.LHeff.get_leg_labels())
    RH = seeded_array(eff.RHeff.legs, 12, labels=eff.RHeff.get_leg_labels())
    th = seeded_array(theta0.legs, 13, labels=theta0.get_leg_labels())
    th2 = seeded_array(theta0.legs, 14, labels=theta0.get_leg_labels())
    rec = dict(L=L, chi=chi, legs_LHeff=[dump_leg(l) for l in eff.LHeff.legs], labels_LHeff=eff.LHeff.get_leg_labels(),
               legs_RHeff=[dump_leg(l) for l in eff.RHeff.legs], labels_RHeff=eff.RHeff.get_leg_labels(),
               legs_theta=[dump_leg(l) for l in theta0.legs], labels_theta=theta0.get_leg_labels(),
               operands=dict(LH=probe_array(LH, 1), RH=probe_array(RH, 2), th=probe_array(th, 3)))
    t1 = npc.tensordot(LH, th, axes=['(vR.p0*)', '(vL.p0)'])                     # mps_common.py:1336
    t2 = npc.tensordot(t1, RH, axes=[['wR', '(p1.vR)'], ['wL', '(p1*.vL)']])     # :1337
    rec['step1'] = probe_array(t1, 4)
    rec['step2'] = probe_array(t2, 5)
    rec['inner'] = complex(npc.inner(th, t2.replace_labels(['(vR*.p0)', '(p1.vL*)'], ['(vL.p0)', '(p1.vR)']), axes='labels', do_conj=True))
    w = th.copy(deep=True)
    w.iadd_prefactor_other(-0.375, th2)
    rec['axpy'] = probe_array(w, 6)
    rec['norm'] = float(npc.norm(th))
    U, S, VH = npc.svd(th, inner_labels=['vR', 'vL'])
    rec['svd_S'] = np.array(S)
    rec['svd_U_qdata'], rec['svd_VH_qdata'] = np.array(U._qdata), np.array(VH._qdata)
    sp = th.split_legs()
    rec['split'] = probe_array(sp, 7)
    rec['recombined'] = probe_array(sp.combine_legs([['vL', 'p0'], ['p1', 'vR']], qconj=[+1, -1]), 8)
    rec['transposed'] = probe_array(sp.transpose(['p1', 'vR', 'vL', 'p0']), 9)
    print('percall', L, rec['chi'], [l['slices'][-1] for l in rec['legs_theta']], len(rec['svd_S']))
    return rec


def gen_percall2048():
    """The same calls at BASELINE's full size: legs of a chi = 2048 Heisenberg centre bond (10 bond sectors
    [2, 24, 122, 334, 542, 542, 334, 122, 24, 2], the block structure of the real chi = 2048 state, cf. scripts/cpu_reference_baseline.py),
    fused theta 4096 x 4096 in blocks up to 1084 x 1084, LHeff / RHeff of 131 MB each.  Seeded operands, fingerprints only (a few KB)."""
    from tenpy.algorithms.mps_common import TwoSiteH
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mpo import MPOEnvironment
    L, sectors = 100, [2, 24, 122, 334, 542, 542, 334, 122, 24, 2]
    M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
    i0 = L // 2 - 1
    site = M.lat.mps_sites()[i0]
    q = np.arange(-(len(sectors) - 1), len(sectors), 2)
    vL = npc.LegCharge.from_qind(site.leg.chinfo, np.concatenate([[0], np.cumsum(sectors)]), q[:, None], qconj=+1)
    vR = vL.conj()
    W0, W1 = M.H_MPO.get_W(i0), M.H_MPO.get_W(i0 + 1)
    LP = seeded_array([vL, W0.get_leg('wL').conj(), vL.conj()], 21, labels=['vR*', 'wR', 'vR'])
    RP = seeded_array([vR.conj(), W1.get_leg('wR').conj(), vR], 22, labels=['vL', 'wL', 'vL*'])
    theta = seeded_array([vL, site.leg, site.leg, vR], 23, labels=['vL', 'p0', 'p1', 'vR'])

    class Env:
        H = M.H_MPO
        get_LP = staticmethod(lambda i, store=True: LP)
        get_RP = staticmethod(lambda i, store=True: RP)
    Env._contract_LHeff = lambda i, label_p='p0', pipe=None: MPOEnvironment._contract_LHeff(Env, i, label_p, pipe)
    Env._contract_RHeff = lambda i, label_p='p1', pipe=None: MPOEnvironment._contract_RHeff(Env, i, label_p, pipe)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        eff = TwoSiteH(Env, i0, combine=True)
        save('percall2048.pkl', [_percall_record(eff, eff.combine_theta(theta), L, 2048)])


def gen_percall():
    """SURVEY 8(c) "extra fixtures": per-call results of the reference on operands with the block structure of REAL DMRG
    states at chi = 64 and chi = 512 (XXZ, Sz): the two tensordots of TwoSiteH.matvec, inner, iadd_prefactor_other, norm,
    svd (singular values), combine_legs / split_legs.  Operands are seeded random tensors on the legs of the converged
    run (block data of chi = 512 environments would be 8 MB per tensor); results are stored as fingerprints (probe_array)."""
    from tenpy.algorithms import dmrg
    from tenpy.algorithms.mps_common import TwoSiteH
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    for L, chi, n_sw in ((24, 64, 5), (48, 512, 6)):
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'conserve': 'Sz', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                              'trunc_params': {'chi_max': chi, 'svd_min': 1.e-12},
                                              'lanczos_params': {'N_min': 2, 'N_max': 6}})
        for _ in range(n_sw):
            eng.sweep()
        i0 = L // 2 - 1
        eff = TwoSiteH(eng.env, i0, combine=True)
        theta0 = eff.combine_theta(psi.get_theta(i0, n=2))
        out.append(_percall_record(eff, theta0, L, int(max(psi.chi))))
    save('percall.pkl', out)



def gen_midsize():
    """Mid-size end-to-end runs of the reference (minutes, not hours): XXZ L=32 chi=128 two-site DMRG
    (per-sweep energies, centre Schmidt values), Hubbard ladder 2x6 chi=128, TFI-parity real-time TEBD L=16 chi=64 (complex)."""
    from tenpy.algorithms import dmrg, tebd
    from tenpy.models.hubbard import FermiHubbardModel
    from tenpy.models.tf_ising import TFIChain
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 32
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                              'trunc_params': {'chi_max': 128, 'svd_min': 1.e-10}, 'lanczos_params': {}})
        Es = []
        for _ in range(6):
            eng.sweep()
            Es.append(float(eng.update_stats['E_total'][-1]))
        out['xxz_L32_chi128'] = dict(L=L, chi=128, E_sweeps=Es, chi_final=int(max(psi.chi)), S_mid=np.array(psi.get_SL(L // 2)),
                                     S_ent=np.array(psi.entanglement_entropy()))
        print('xxz_L32_chi128', Es[-1], max(psi.chi))
        Lx = 6
        M = FermiHubbardModel({'lattice': 'Ladder', 'L': Lx, 't': 1., 'U': 8., 'mu': 0., 'cons_N': 'N', 'cons_Sz': 'Sz',
                               'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * Lx, bc='finite')
        eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                              'trunc_params': {'chi_max': 128, 'svd_min': 1.e-10}, 'lanczos_params': {}})
        Es = []
        for _ in range(7):
            eng.sweep()
            Es.append(float(eng.update_stats['E_total'][-1]))
        out['hubbard_2x6_chi128'] = dict(Lx=Lx, chi=128, t=1., U=8., mu=0., E_sweeps=Es, chi_final=int(max(psi.chi)),
                                         S_ent=np.array(psi.entanglement_entropy()))
        print('hubbard_2x6_chi128', Es[-1], max(psi.chi))
        L = 16
        M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'finite', 'conserve': 'parity', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
        eng = tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.05, 'N_steps': 1, 'trunc_params': {'chi_max': 64, 'svd_min': 1.e-10}})
        S_t, chi_t = [], []
        for step in range(40):
            eng.run()
            if step % 4 == 3:
                S_t.append(np.array(psi.entanglement_entropy()))
                chi_t.append(int(max(psi.chi)))
        out['tebd_tfi_parity_L16_chi64'] = dict(L=L, J=1., g=1.5, dt=0.05, chi=64, every=4, S_t=np.array(S_t), chi_t=chi_t,
                                                S_mid=np.array(psi.get_SL(L // 2)),
                                                h_bond=[None if h is None else h.transpose(['p0', 'p1', 'p0*', 'p1*']).to_ndarray() for h in M.H_bond],
                                                state_labels=list(M.lat.mps_sites()[0].state_labels.items()))
        print('tebd', chi_t[-1], S_t[-1][L // 2 - 1])
    save('midsize.pkl', out)


def _scaled_leg(leg, factor):
    """Same charge sectors, every sector `factor` times as wide (structure of a larger bond dimension)."""
    sizes = np.diff(leg.slices) * factor
    return npc.LegCharge.from_qind(leg.chinfo, np.concatenate([[0], np.cumsum(sizes)]), np.array(leg.charges), qconj=leg.qconj)


def gen_percall_hubbard():
    """per-call fixtures in the many-small-blocks regime (BASELINE config 4): Fermi-Hubbard ladder, charges (N, Sz).
    Record 0: legs of a REAL two-site DMRG state of a 2 x 8 ladder at chi = 256 (centre bond); record 1: the same charge
    sectors with every bond sector 4 times as wide = the block structure of chi = 1024.  Seeded operands, fingerprints."""
    from tenpy.algorithms import dmrg
    from tenpy.algorithms.mps_common import TwoSiteH
    from tenpy.models.hubbard import FermiHubbardModel
    from tenpy.networks.mpo import MPOEnvironment
    from tenpy.networks.mps import MPS
    Lx = 8
    M = FermiHubbardModel({'lattice': 'Ladder', 'L': Lx, 't': 1., 'U': 8., 'mu': 0., 'cons_N': 'N', 'cons_Sz': 'Sz',
                           'bc_MPS': 'finite', 'sort_charge': True})
    psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * Lx, bc='finite')
    eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                          'trunc_params': {'chi_max': 256, 'svd_min': 1.e-12}, 'lanczos_params': {'N_min': 2, 'N_max': 6}})
    for _ in range(5):
        eng.sweep()
    L = 2 * Lx
    i0 = L // 2 - 1
    out = []
    eff = TwoSiteH(eng.env, i0, combine=True)
    theta0 = eff.combine_theta(psi.get_theta(i0, n=2))
    out.append(_percall_record(eff, theta0, L, int(max(psi.chi))))
    # chi = 1024 structure: widen the bond legs of LP / RP / theta, rebuild the effective Hamiltonian on seeded environments
    vL = _scaled_leg(eng.env.get_LP(i0).get_leg('vR*'), 4)
    vR = _scaled_leg(eng.env.get_RP(i0 + 1).get_leg('vL*'), 4)
    W0, W1 = M.H_MPO.get_W(i0), M.H_MPO.get_W(i0 + 1)
    site0, site1 = M.lat.mps_sites()[i0], M.lat.mps_sites()[i0 + 1]
    LP = seeded_array([vL, W0.get_leg('wL').conj(), vL.conj()], 21, labels=['vR*', 'wR', 'vR'])
    RP = seeded_array([vR.conj(), W1.get_leg('wR').conj(), vR], 22, labels=['vL', 'wL', 'vL*'])
    theta = seeded_array([vL, site0.leg, site1.leg, vR], 23, labels=['vL', 'p0', 'p1', 'vR'])

    class Env:
        H = M.H_MPO
        get_LP = staticmethod(lambda i, store=True: LP)
        get_RP = staticmethod(lambda i, store=True: RP)
    Env._contract_LHeff = lambda i, label_p='p0', pipe=None: MPOEnvironment._contract_LHeff(Env, i, label_p, pipe)
    Env._contract_RHeff = lambda i, label_p='p1', pipe=None: MPOEnvironment._contract_RHeff(Env, i, label_p, pipe)
    eff = TwoSiteH(Env, i0, combine=True)
    out.append(_percall_record(eff, eff.combine_theta(theta), L, int(vL.ind_len)))
    save('percall_hubbard.pkl', out)


def gen_percall_tebd():
    """complex blocks of BASELINE config 5 (real-time TEBD, TFI chain with parity): the calls of one bond update
    of ``TEBDEngine.update_bond`` (algorithms/tebd.py:416-470) on seeded complex operands -- gate application
    ``tensordot(U_bond, theta)``, ``combine_legs``, complex block ``svd`` (singular values), the new-B contraction --
    record 0 on the legs of a REAL quench state at chi = 64, record 1 with every bond sector 16 times as wide (chi = 1024:
    two 1024 x 1024 complex blocks, the block-SVD-bound case)."""
    from tenpy.algorithms import tebd
    from tenpy.models.tf_ising import TFIChain
    from tenpy.networks.mps import MPS
    L = 16
    M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'finite', 'conserve': 'parity', 'sort_charge': True})
    psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
    eng = tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.05, 'N_steps': 30, 'trunc_params': {'chi_max': 64, 'svd_min': 1.e-12}})
    eng.run()
    i = L // 2
    th_real = psi.get_theta(i - 1, n=2)                      # vL, p0, p1, vR
    site = M.lat.mps_sites()[i]
    out = []
    for factor in (1, 16):
        vL = _scaled_leg(th_real.get_leg('vL'), factor)
        vR = _scaled_leg(th_real.get_leg('vR'), factor)
        theta = seeded_array([vL, site.leg, site.leg, vR], 31, cplx=True, labels=['vL', 'p0', 'p1', 'vR'])
        gate = seeded_array([site.leg, site.leg, site.leg.conj(), site.leg.conj()], 32, cplx=True, labels=['p0', 'p1', 'p0*', 'p1*'])
        rec = dict(chi=int(vL.ind_len), legs_theta=[dump_leg(l) for l in theta.legs], labels_theta=theta.get_leg_labels(),
                   legs_gate=[dump_leg(l) for l in gate.legs], labels_gate=gate.get_leg_labels(),
                   operands=dict(theta=probe_array(theta, 1), gate=probe_array(gate, 2)))
        t = npc.tensordot(gate, theta, axes=(['p0*', 'p1*'], ['p0', 'p1']))          # tebd.py:441
        rec['gate_theta'] = probe_array(t, 3)
        tc = t.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])              # tebd.py:442
        rec['combined'] = probe_array(tc, 4)
        rec['norm'] = float(npc.norm(tc))
        U, S, VH = npc.svd(tc, inner_labels=['vR', 'vL'])
        rec['svd_S'] = np.array(S)
        rec['svd_U_qdata'], rec['svd_VH_qdata'] = np.array(U._qdata), np.array(VH._qdata)
        rec['inner'] = complex(npc.inner(tc, tc, axes='range', do_conj=True))
        Bn = npc.tensordot(theta.conj(), t, axes=(['vL*', 'p0*'], ['vL', 'p0']))       # a contraction with conj(), as in tebd.py:463
        rec['conj_contract'] = probe_array(Bn, 5)
        print('percall_tebd', rec['chi'], [l['slices'][-1] for l in rec['legs_theta']], len(S))
        out.append(rec)
    save('percall_tebd.pkl', out)


def gen_eig_svd():
    """``truncation._eig_based_svd`` (linalg/truncation.py:473-530, the reference's "performs better on GPU" route used by the
    QR-based TEBD, tebd.py:685): singular values, truncation error, renormalisation and the projectors U U^dagger / Vd^dagger Vd
    (the phases of U / Vd are arbitrary) for real and complex block matrices, with and without truncation."""
    out = []
    ch = charges.ChargeInfo([1], ['2Sz'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for cplx in (False, True):
            for seed, (m, n) in enumerate(((24, 30), (40, 17), (33, 33))):
                lL, lR = rand_leg(ch, m, 1), rand_leg(ch, n, -1)
                A = rand_array([lL, lR], cplx=cplx, labels=['a', 'b'])
                # (need_U = need_Vd = False contracts the wrong axes in the reference, truncation.py:511/:513, and raises: not pinned)
                for need_U, need_Vd in ((True, False), (False, True)):
                    for tp in (None, {'chi_max': 9, 'svd_min': 1.e-8}):
                        U, S, Vd, err, ren = truncation._eig_based_svd(A, need_U=need_U, need_Vd=need_Vd, inner_labels=['x', 'y'],
                                                                       trunc_params=tp)
                        rec = dict(A=dump_array(A), need_U=need_U, need_Vd=need_Vd, trunc=tp, S=np.array(S), eps=float(err.eps),
                                   renormalize=float(ren))
                        if U is not None:
                            rec['UUh'] = npc.tensordot(U, U.conj(), axes=['x', 'x*']).to_ndarray()
                            rec['U_labels'] = U.get_leg_labels()
                        if Vd is not None:
                            rec['VhV'] = npc.tensordot(Vd.conj(), Vd, axes=['y*', 'y']).to_ndarray()
                            rec['Vd_labels'] = Vd.get_leg_labels()
                        out.append(rec)
    save('eig_svd.pkl', out)


GENERATORS = dict(percall_hubbard=gen_percall_hubbard, percall_tebd=gen_percall_tebd, eig_svd=gen_eig_svd, midsize=gen_midsize, percall=gen_percall, percall2048=gen_percall2048, api2=gen_api2, krylov2=gen_krylov2, hubbard=gen_hubbard, charges=gen_charges, tensordot=gen_tensordot,
                  reshape=gen_reshape, linalg=gen_linalg, truncate=gen_truncate, lanczos=gen_lanczos, dmrg=gen_dmrg, tebd=gen_tebd,
                  qr_theta=gen_qr_theta)

if __name__ == '__main__':
    print("reference:", tenpy.__version__, tenpy.__file__)
    only = os.environ.get('ONLY')
    for name, fn in GENERATORS.items():
        if only is None or name in only.split(','):
            fn()
