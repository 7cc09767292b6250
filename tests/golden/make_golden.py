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


def gen_mixer():
    """DensityMatrixMixer.mix_rho / svd_from_rho of the reference on a dumped bond (LP, RP, W0, W1, theta)."""
    from tenpy.algorithms import dmrg, mps_common
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 10
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1.3, 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': True, 'mixer_params': {'amplitude': 1.e-2, 'decay': 1., 'disable_after': 100},
                                              'trunc_params': {'chi_max': 12, 'svd_min': 1e-10}, 'combine': True})
        eng.sweep()
        eng.sweep()
        i0 = L // 2 - 1
        eng.i0 = i0
        eng.move_right = True
        eng.make_eff_H()
        theta = eng.eff_H.combine_theta(psi.get_theta(i0, n=2))
        theta = theta + 1e-2 * rand_array(theta.legs, qtotal=theta.qtotal, labels=theta.get_leg_labels())
        H = M.H_MPO
        for amplitude in (1e-2, 0.3):
            mixer = mps_common.DensityMatrixMixer({'amplitude': amplitude, 'decay': 1., 'disable_after': 100}, 0)
            for ml, mr in ((True, True), (True, False), (False, True)):
                rho_L, rho_R = mixer.mix_rho(eng, theta, i0, ml, mr)
                qLR = [psi.get_B(i0, None).qtotal, None]
                U, S, VH, err, S_a = mixer.svd_from_rho(eng, rho_L.copy(deep=True), rho_R.copy(deep=True), theta, qLR)
                out.append(dict(LP=dump_array(eng.env.get_LP(i0)), RP=dump_array(eng.env.get_RP(i0 + 1)),
                                W0=dump_array(H.get_W(i0)), W1=dump_array(H.get_W(i0 + 1)), theta=dump_array(theta),
                                IdL=H.get_IdL(i0 + 1), IdR=H.get_IdR(i0), amplitude=amplitude, mix_left=ml, mix_right=mr,
                                rho_L=dump_array(rho_L), rho_R=dump_array(rho_R), S=dump_array(S), S_a=np.array(S_a),
                                qtotal_L=np.array(qLR[0]), eps=float(err.eps), U=dump_array(U), VH=dump_array(VH), chi_max=12))
    save('mixer.pkl', out)


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


def gen_dmrg_mixer():
    """Two-site DMRG WITH the density-matrix mixer (mixer on for the first sweeps, then decays / is disabled), then
    mixer_cleanup: energies of every update, truncation errors, final Schmidt spectra."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for (L, Jz, hz, chi, amp, decay, dis, n_sweeps) in ((12, 1.3, 0., 16, 1.e-3, 2., 4, 7), (10, 0.7, 0.2, 12, 1.e-2, 1.5, 3, 6)):
            M = XXZChain({'L': L, 'Jxx': 1., 'Jz': Jz, 'hz': hz, 'bc_MPS': 'finite', 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
            eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': True, 'mixer_params': {'amplitude': amp, 'decay': decay, 'disable_after': dis},
                                                  'combine': True, 'max_N_for_ED': 0, 'trunc_params': {'chi_max': chi, 'svd_min': 1.e-6}})
            # svd_min = 1e-6 on purpose: the mixed density matrices are rank deficient while the bond dimension is
            # still growing; with svd_min = 1e-10 eigenvalues that are pure rounding noise (1e-17 -> S = 3e-9) decide
            # which "state" is kept and the trajectory is not reproducible across LAPACK builds
            eng.mixer_activate()
            Es, mixer_on = [], []
            for s in range(n_sweeps):
                mixer_on.append(eng.mixer is not None)
                eng.sweep()
                Es.append(float(eng.update_stats['E_total'][-1]))
            eng.mixer_cleanup()
            out.append(dict(L=L, Jxx=1., Jz=Jz, hz=hz, chi=chi, amplitude=amp, decay=decay, disable_after=dis, n_sweeps=n_sweeps,
                            E_sweeps=Es, mixer_on=mixer_on, E_updates=[float(e) for e in eng.update_stats['E_total']],
                            err_updates=[float(e.eps) for e in eng.update_stats['err']],
                            S=[np.array(psi.get_SL(i)) for i in range(1, L)], S_ent=np.array(psi.entanglement_entropy()),
                            E_mpo=float(np.real(M.H_MPO.expectation_value(psi))), svd_min=1.e-6))
            print('dmrg_mixer', L, Es, mixer_on)
    save('dmrg_mixer.pkl', out)


def gen_dmrg_single():
    """Single-site DMRG (OneSiteH, combine=True): (a) from a product state with the SubspaceExpansion mixer,
    (b) two two-site sweeps first, then single-site sweeps without any mixer.  Energies of every update, truncation
    errors, mixer schedule, final Schmidt spectra after mixer_cleanup."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for (L, Jz, hz, chi, amp, decay, dis, n_sweeps, pre2) in ((12, 1.3, 0., 16, 1.e-3, 2., 4, 7, 0), (10, 0.7, 0.2, 12, 1.e-2, 1.5, 3, 6, 0),
                                                                  (12, 1.0, 0., 14, None, None, None, 3, 2)):
            M = XXZChain({'L': L, 'Jxx': 1., 'Jz': Jz, 'hz': hz, 'bc_MPS': 'finite', 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
            E_pre = []
            if pre2:
                e2 = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0,
                                                     'trunc_params': {'chi_max': chi, 'svd_min': 1.e-10}})
                for s in range(pre2):
                    e2.sweep()
                    E_pre.append(float(e2.update_stats['E_total'][-1]))
            opts = {'combine': True, 'max_N_for_ED': 0, 'trunc_params': {'chi_max': chi, 'svd_min': 1.e-6 if amp else 1.e-10}}
            if amp:
                opts.update(mixer=True, mixer_params={'amplitude': amp, 'decay': decay, 'disable_after': dis})
            else:
                opts.update(mixer=None)
            eng = dmrg.SingleSiteDMRGEngine(psi, M, opts)
            eng.mixer_activate()
            Es, mixer_on = [], []
            for s in range(n_sweeps):
                mixer_on.append(eng.mixer is not None)
                eng.sweep()
                Es.append(float(eng.update_stats['E_total'][-1]))
            eng.mixer_cleanup()
            out.append(dict(L=L, Jxx=1., Jz=Jz, hz=hz, chi=chi, amplitude=amp, decay=decay, disable_after=dis, n_sweeps=n_sweeps,
                            pre_two_site_sweeps=pre2, E_pre=E_pre, E_sweeps=Es, mixer_on=mixer_on,
                            i0_updates=[int(i) for i in eng.update_stats['i0']],
                            E_updates=[float(e) for e in eng.update_stats['E_total']],
                            err_updates=[float(e.eps) for e in eng.update_stats['err']],
                            chi_final=[int(c) for c in psi.chi],
                            S=[np.array(psi.get_SL(i)) for i in range(1, L)], S_ent=np.array(psi.entanglement_entropy()),
                            E_mpo=float(np.real(M.H_MPO.expectation_value(psi))), svd_min=opts['trunc_params']['svd_min']))
            print('dmrg_single', L, Es, mixer_on, psi.chi)
    save('dmrg_single.pkl', out)


def gen_dmrg_two_site_subspace():
    """Two-site DMRG with mixer='SubspaceExpansion' (Mixer.mix_and_decompose_2site falling back to the one-site
    decomposition)."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for (L, Jz, hz, chi, amp, decay, dis, n_sweeps) in ((12, 1.3, 0., 16, 1.e-3, 2., 3, 5), ):
            M = XXZChain({'L': L, 'Jxx': 1., 'Jz': Jz, 'hz': hz, 'bc_MPS': 'finite', 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
            eng = dmrg.TwoSiteDMRGEngine(psi, M, {'mixer': 'SubspaceExpansion', 'mixer_params': {'amplitude': amp, 'decay': decay, 'disable_after': dis},
                                                  'combine': True, 'max_N_for_ED': 0, 'trunc_params': {'chi_max': chi, 'svd_min': 1.e-6}})
            eng.mixer_activate()
            Es, mixer_on = [], []
            for s in range(n_sweeps):
                mixer_on.append(eng.mixer is not None)
                eng.sweep()
                Es.append(float(eng.update_stats['E_total'][-1]))
            eng.mixer_cleanup()
            out.append(dict(L=L, Jxx=1., Jz=Jz, hz=hz, chi=chi, amplitude=amp, decay=decay, disable_after=dis, n_sweeps=n_sweeps,
                            E_sweeps=Es, mixer_on=mixer_on, E_updates=[float(e) for e in eng.update_stats['E_total']],
                            err_updates=[float(e.eps) for e in eng.update_stats['err']],
                            S=[np.array(psi.get_SL(i)) for i in range(1, L)], svd_min=1.e-6))
            print('dmrg_two_site_subspace', L, Es, mixer_on)
    save('dmrg_two_site_subspace.pkl', out)


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


def gen_tebd2():
    """TEBD with the other Suzuki-Trotter orders (1, 4, '4_opt'; 3 merged steps per evolve call) and imaginary time."""
    from tenpy.algorithms import tebd
    from tenpy.models.tf_ising import TFIChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 8
        M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'finite', 'conserve': 'parity', 'sort_charge': True})
        h_bond = [None if h is None else h.transpose(['p0', 'p1', 'p0*', 'p1*']).to_ndarray() for h in M.H_bond]
        labels = list(M.lat.mps_sites()[0].state_labels.items())
        for order, type_evo in ((1, 'real'), (2, 'real'), (4, 'real'), ('4_opt', 'real'), (2, 'imag'), (4, 'imag')):
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
            eng = tebd.TEBDEngine(psi, M, {'order': order, 'dt': 0.05, 'N_steps': 3, 'trunc_params': {'chi_max': 12, 'svd_min': 1.e-10}})
            eng.calc_U(order, 0.05, type_evo=type_evo)
            S_t, chi_t, errs = [], [], []
            for rep in range(4):
                err = eng.evolve(3, 0.05)
                S_t.append(np.array(psi.entanglement_entropy()))
                chi_t.append(int(max(psi.chi)))
                errs.append(float(err.eps))
            out.append(dict(order=order, type_evo=type_evo, L=L, dt=0.05, chi=12, N_steps=3, S_t=np.array(S_t), chi_t=chi_t, err_t=errs,
                            evolved_time=complex(eng.evolved_time), decomposition=list(tebd.TEBDEngine.suzuki_trotter_decomposition(order, 3)),
                            time_steps=list(tebd.TEBDEngine.suzuki_trotter_time_steps(order)),
                            S_mid=np.array(psi.get_SL(L // 2)), h_bond=h_bond, state_labels=labels, conserve='parity'))
            print('tebd2', order, type_evo, chi_t, S_t[-1][L // 2 - 1])
    save('tebd2.pkl', out)


def gen_dmrg_ortho():
    """Excited state by DMRG with ``orthogonal_to=[ground state]`` (two-site and single-site engines)."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L, chi = 10, 24
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1.2, 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
        psi0 = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        e0 = dmrg.TwoSiteDMRGEngine(psi0, M, {'mixer': None, 'combine': True, 'max_N_for_ED': 0, 'trunc_params': {'chi_max': chi, 'svd_min': 1.e-10}})
        E0s = []
        for s_ in range(5):
            e0.sweep()
            E0s.append(float(e0.update_stats['E_total'][-1]))
        for engine, n_sw in (('two', 6), ('single', 6)):
            psi1 = MPS.from_product_state(M.lat.mps_sites(), ['down', 'up'] * (L // 2), bc='finite')
            cls = dmrg.TwoSiteDMRGEngine if engine == 'two' else dmrg.SingleSiteDMRGEngine
            opts = {'combine': True, 'max_N_for_ED': 0, 'trunc_params': {'chi_max': chi, 'svd_min': 1.e-6 if engine == 'single' else 1.e-10}}
            if engine == 'single':
                opts.update(mixer=True, mixer_params={'amplitude': 1.e-3, 'decay': 2., 'disable_after': 3})
            else:
                opts.update(mixer=None)
            e1 = cls(psi1, M, opts, orthogonal_to=[psi0])
            e1.mixer_activate()
            Es = []
            for s_ in range(n_sw):
                e1.sweep()
                Es.append(float(e1.update_stats['E_total'][-1]))
            e1.mixer_cleanup()
            out.append(dict(engine=engine, L=L, Jxx=1., Jz=1.2, hz=0., chi=chi, E0_sweeps=E0s, E1_sweeps=Es,
                            E1_updates=[float(e) for e in e1.update_stats['E_total']], overlap=complex(psi0.overlap(psi1)),
                            E1_mpo=float(np.real(M.H_MPO.expectation_value(psi1))), svd_min=opts['trunc_params']['svd_min']))
            print('dmrg_ortho', engine, E0s[-1], Es, abs(psi0.overlap(psi1)))
    save('dmrg_ortho.pkl', out)


def gen_dmrg_default_diag():
    """Two-site DMRG with the reference's DEFAULT diagonalisation (ED below max_N_for_ED=400, Lanczos above) and with
    'ED_block' everywhere (small chi)."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for (L, chi, method, engine) in ((14, 24, 'default', 'two'), (10, 8, 'ED_block', 'two'), (12, 12, 'ED_block', 'single')):
            M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 0.8, 'hz': 0.1, 'bc_MPS': 'finite', 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
            opts = {'combine': True, 'diag_method': method, 'trunc_params': {'chi_max': chi, 'svd_min': 1.e-10 if engine == 'two' else 1.e-6}}
            if engine == 'two':
                opts['mixer'] = None
                eng = dmrg.TwoSiteDMRGEngine(psi, M, opts)
            else:
                opts.update(mixer=True, mixer_params={'amplitude': 1.e-3, 'decay': 2., 'disable_after': 3})
                eng = dmrg.SingleSiteDMRGEngine(psi, M, opts)
            eng.mixer_activate()
            Es = []
            for s_ in range(5):
                eng.sweep()
                Es.append(float(eng.update_stats['E_total'][-1]))
            eng.mixer_cleanup()
            out.append(dict(L=L, Jxx=1., Jz=0.8, hz=0.1, chi=chi, diag_method=method, engine=engine, E_sweeps=Es,
                            E_updates=[float(e) for e in eng.update_stats['E_total']], N_lanczos=[int(n) for n in eng.update_stats['N_lanczos']],
                            svd_min=opts['trunc_params']['svd_min'], S=[np.array(psi.get_SL(i)) for i in range(1, L)]))
            print('dmrg_default_diag', L, chi, method, engine, Es[-1], sorted(set(out[-1]['N_lanczos']))[:5])
    save('dmrg_default_diag.pkl', out)


def gen_dmrg_run():
    """The reference's ``DMRGEngine.run()`` main loop with its default options (examples/d_dmrg.py parameters: TFI chain,
    chi_max=30, svd_min=1e-10, max_E_err=1e-10, no mixer, combine, DEFAULT diag_method and Lanczos tolerances that follow
    the truncation error), and an XXZ run with the mixer on by default parameters."""
    from tenpy.algorithms import dmrg
    from tenpy.models.tf_ising import TFIChain
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for case in ('tfi_d_dmrg', 'xxz_mixer'):
            if case == 'tfi_d_dmrg':
                L = 16
                M = TFIChain({'L': L, 'J': 1., 'g': 1., 'bc_MPS': 'finite', 'conserve': None})
                psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
                opts = {'mixer': None, 'max_E_err': 1.e-10, 'trunc_params': {'chi_max': 30, 'svd_min': 1.e-10}, 'combine': True}
                extra = dict(L=L, J=1., g=1., conserve=None)
            else:
                L = 12
                M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
                psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
                opts = {'mixer': True, 'mixer_params': {'amplitude': 1.e-4, 'decay': 2., 'disable_after': 4}, 'max_E_err': 1.e-9,
                        'trunc_params': {'chi_max': 20, 'svd_min': 1.e-6}, 'combine': True, 'max_N_for_ED': 0}
                extra = dict(L=L, Jxx=1., Jz=1., hz=0., conserve='Sz')
            import copy
            opts_plain = copy.deepcopy(opts)
            eng = dmrg.TwoSiteDMRGEngine(psi, M, opts)
            E, _ = eng.run()
            st = eng.sweep_stats
            rec = dict(case=case, E=float(E), sweeps=int(eng.sweeps), P_tol_final=float(eng.lanczos_params['P_tol']),
                       N_lanczos=[int(n) for n in eng.update_stats['N_lanczos']], E_updates=[float(e) for e in eng.update_stats['E_total']],
                       E_trunc=[None if e is None else float(e) for e in eng.update_stats['E_trunc']],
                       sweep_stats={k: [float(x) for x in st[k]] for k in ('sweep', 'N_updates', 'E', 'Delta_E', 'S', 'Delta_S', 'max_S',
                                                                              'max_trunc_err', 'max_E_trunc', 'max_chi')},
                       options=opts_plain)
            rec.update(extra)
            out.append(rec)
            print('dmrg_run', case, E, eng.sweeps, eng.lanczos_params['P_tol'])
    save('dmrg_run.pkl', out)


def gen_tdvp():
    """Real-time TDVP after a quench from the Neel state (XXZ, Sz conserved): two-site engine growing the bond dimension,
    then the single-site engine continuing on the same state."""
    from tenpy.algorithms import tdvp
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 8
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 0.7, 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        opts = {'dt': 0.05, 'N_steps': 2, 'trunc_params': {'chi_max': 12, 'svd_min': 1.e-10}, 'lanczos_params': {'N_min': 2, 'N_max': 20}}
        eng = tdvp.TwoSiteTDVPEngine(psi, M, dict(opts))
        recs = []
        for rep in range(4):
            eng.run()
            recs.append(dict(engine='two', S=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi],
                             Sz=np.array(psi.expectation_value('Sz')), norm=float(psi.norm), t=float(eng.evolved_time),
                             trunc_err=float(eng.trunc_err.eps)))
        eng1 = tdvp.SingleSiteTDVPEngine(psi, M, dict(opts))
        for rep in range(4):
            eng1.run()
            recs.append(dict(engine='single', S=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi],
                             Sz=np.array(psi.expectation_value('Sz')), norm=float(psi.norm), t=float(eng1.evolved_time),
                             trunc_err=float(eng1.trunc_err.eps)))
        out.append(dict(L=L, Jxx=1., Jz=0.7, hz=0., options=opts, steps=recs,
                        E=float(np.real(M.H_MPO.expectation_value(psi)))))
        print('tdvp', [r['chi'] for r in recs][-1], recs[-1]['S'], recs[-1]['norm'])
    save('tdvp.pkl', out)


def gen_idmrg():
    """Infinite DMRG (two-site, Lanczos always): XXZ chain with a 2-site unit cell and TFI with parity, through the
    reference's run() loop (N_sweeps_check, environment sweeps, energy per site from the growth of the system)."""
    from tenpy.algorithms import dmrg
    from tenpy.models.tf_ising import TFIChain
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    import copy
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for case in ('xxz', 'tfi', 'xxz_mixer'):
            if case.startswith('xxz'):
                L = 2
                M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1.5, 'hz': 0., 'bc_MPS': 'infinite', 'sort_charge': True})
                psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'], bc='infinite')
                extra = dict(L=L, Jxx=1., Jz=1.5, hz=0., conserve='Sz')
            else:
                L = 2
                M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': 'infinite', 'conserve': 'parity', 'sort_charge': True})
                psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'up'], bc='infinite')
                extra = dict(L=L, J=1., g=1.5, conserve='parity')
            opts = {'mixer': None, 'combine': True, 'max_N_for_ED': 0, 'max_E_err': 1.e-10, 'max_sweeps': 40, 'N_sweeps_check': 5,
                    'trunc_params': {'chi_max': 16, 'svd_min': 1.e-10}}
            if case == 'xxz_mixer':
                opts.update(mixer=True, mixer_params={'amplitude': 1.e-4, 'decay': 1.5, 'disable_after': 6})
                opts['trunc_params']['svd_min'] = 1.e-6
            opts_plain = copy.deepcopy(opts)
            eng = dmrg.TwoSiteDMRGEngine(psi, M, opts)
            E, _ = eng.run()
            st = eng.sweep_stats
            rec = dict(case=case, E=float(E), sweeps=int(eng.sweeps), options=opts_plain,
                       E_updates=[float(e) for e in eng.update_stats['E_total']], age=[int(a) for a in eng.update_stats['age']],
                       i0=[int(i) for i in eng.update_stats['i0']],
                       sweep_stats={k: [float(x) for x in st[k]] for k in ('sweep', 'N_updates', 'E', 'Delta_E', 'S', 'Delta_S', 'max_S',
                                                                              'max_trunc_err', 'max_E_trunc', 'max_chi')},
                       S=[np.array(psi.get_SL(i)) for i in range(L)], chi=[int(c) for c in psi.chi])
            rec.update(extra)
            out.append(rec)
            print('idmrg', case, E, eng.sweeps, psi.chi)
    save('idmrg.pkl', out)


def gen_idmrg_bench():
    """The reference's iDMRG benchmark (tests/benchmark/dmrg_infinite.py) in small: spin-2 chain with D=0.3, Sz conserved,
    Lanczos N_min=N_max=10, optimisation sweeps alternating with environment sweeps."""
    from tenpy.algorithms import dmrg
    from tenpy.models.spins import SpinChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L, chi = 4, 14
        M = SpinChain(dict(L=L, S=2., D=0.3, bc_MPS='infinite', conserve='Sz', sort_charge=True))
        psi = MPS.from_product_state(M.lat.mps_sites(), (['up', 'down'] * L)[:L], bc='infinite')
        eng = dmrg.TwoSiteDMRGEngine(psi, M, {'trunc_params': {'chi_max': chi, 'svd_min': 1.e-10}, 'lanczos_params': {'N_min': 10, 'N_max': 10},
                                              'max_N_for_ED': 0, 'combine': True})
        eng.diag_method = 'lanczos'
        for i in range(6):
            eng.sweep(meas_E_trunc=False)
            eng.sweep(optimize=False, meas_E_trunc=False)
        Es, ages = eng.update_stats['E_total'], eng.update_stats['age']
        out.append(dict(L=L, chi=chi, S=2., D=0.3, E_updates=[float(e) for e in Es], age=[int(a) for a in ages],
                        i0=[int(i) for i in eng.update_stats['i0']], chi_final=[int(c) for c in psi.chi],
                        S_ent=np.array(psi.entanglement_entropy()), state_labels=list(M.lat.mps_sites()[0].state_labels.items())))
        print('idmrg_bench', psi.chi, (Es[-1] - Es[-9]) / (ages[-1] - ages[-9]))
    save('idmrg_bench.pkl', out)


def gen_tebd_infinite():
    """The reference's TEBD benchmark (tests/benchmark/tebd_infinite.py) in small: infinite spin-2 chain, order 2."""
    from tenpy.algorithms import tebd
    from tenpy.models.spins import SpinChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L, chi = 2, 12
        M = SpinChain(dict(L=L, S=2., D=0.3, bc_MPS='infinite', conserve='Sz', sort_charge=True))
        psi = MPS.from_product_state(M.lat.mps_sites(), (['up', 'down'] * L)[:L], bc='infinite')
        eng = tebd.TEBDEngine(psi, M, {'trunc_params': {'chi_max': chi, 'svd_min': 1.e-10}, 'order': 2, 'N_steps': 2, 'dt': 0.05})
        S_t, chi_t = [], []
        for rep in range(5):
            eng.run()
            S_t.append(np.array(psi.entanglement_entropy()))
            chi_t.append([int(c) for c in psi.chi])
        out.append(dict(L=L, chi=chi, S_t=np.array(S_t), chi_t=chi_t, trunc_err=float(eng.trunc_err.eps), t=float(eng.evolved_time),
                        h_bond=[h.transpose(['p0', 'p1', 'p0*', 'p1*']).to_ndarray() for h in M.H_bond],
                        S0=np.array(psi.get_SL(0)), S1=np.array(psi.get_SL(1))))
        print('tebd_infinite', chi_t[-1], S_t[-1])
    save('tebd_infinite.pkl', out)


def gen_canonical_form():
    """MPS.canonical_form of a finite MPS made of random (non-canonical) tensors and of a slightly perturbed canonical one."""
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    r4 = np.random.RandomState(99)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 6
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1., 'hz': 0., 'bc_MPS': 'finite', 'sort_charge': True})
        for cplx in (False, True):
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite', dtype=np.complex128 if cplx else np.float64)
            from tenpy.algorithms import tebd
            eng = tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.1, 'N_steps': 4, 'trunc_params': {'chi_max': 8, 'svd_min': 1.e-10}})
            eng.run()
            # perturb every tensor: the form labels stay 'B' but the state is no longer canonical
            Bs_in = []
            for i in range(L):
                B = psi.get_B(i, 'B')
                noise = npc.Array.from_func(lambda size: 0.05 * (r4.standard_normal(size) + (1.j * r4.standard_normal(size) if cplx else 0.)),
                                            B.legs, dtype=B.dtype, qtotal=B.qtotal, shape_kw='size')
                noise.iset_leg_labels(B.get_leg_labels())
                Bn = B + noise
                psi.set_B(i, Bn, form='B')
                Bs_in.append(dump_array(Bn.transpose(['vL', 'p', 'vR'])))
            S_in = [np.array(psi.get_SL(i)) for i in range(L)] + [np.array(psi.get_SR(L - 1))]
            for renorm in (True, False):
                p2 = psi.copy()
                p2.canonical_form(renormalize=renorm)
                out.append(dict(L=L, cplx=cplx, renormalize=renorm, B_in=Bs_in, S_in=S_in, norm=float(p2.norm),
                                S_out=[np.array(p2.get_SL(i)) for i in range(L)] + [np.array(p2.get_SR(L - 1))], S_ent=np.array(p2.entanglement_entropy()),
                                overlap=complex(p2.overlap(psi)), chi=[int(c) for c in p2.chi]))
            print('canonical_form', cplx, out[-1]['norm'], out[-1]['chi'])
    save('canonical_form.pkl', out)


def gen_canonical_form_infinite():
    """canonical_form of an infinite MPS that is slightly out of canonical form (after real-time iTEBD with truncation
    and an extra perturbation of the tensors)."""
    from tenpy.algorithms import tebd
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    r5 = np.random.RandomState(123)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 2
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1.5, 'hz': 0., 'bc_MPS': 'infinite', 'sort_charge': True})
        for cplx in (False, True):
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'], bc='infinite')
            eng = tebd.TEBDEngine(psi, M, {'order': 2, 'dt': 0.1, 'N_steps': 6, 'trunc_params': {'chi_max': 8, 'svd_min': 1.e-10}})
            if cplx:
                eng.run()
            else:
                eng.calc_U(2, 0.1, type_evo='imag')
                eng.evolve(6, 0.1)
            Bs_in = []
            for i in range(L):
                B = psi.get_B(i, 'B')
                noise = npc.Array.from_func(lambda size: 0.03 * (r5.standard_normal(size) + (1.j * r5.standard_normal(size) if cplx else 0.)),
                                            B.legs, dtype=B.dtype, qtotal=B.qtotal, shape_kw='size')
                noise.iset_leg_labels(B.get_leg_labels())
                Bn = B + noise
                psi.set_B(i, Bn, form='B')
                Bs_in.append(dump_array(Bn.transpose(['vL', 'p', 'vR'])))
            S_in = [np.array(psi.get_SL(i)) for i in range(L)]
            err_in = np.array(psi.norm_test())
            psi_before = psi.copy()
            psi.canonical_form()
            import warnings as _w
            ov_self = complex(psi.overlap(psi, understood_infinite=True))
            other = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'], bc='infinite')
            ov_prod = complex(psi.overlap(other, understood_infinite=True))
            out.append(dict(L=L, cplx=cplx, ov_self=ov_self, ov_prod=ov_prod, norm=float(psi.norm), B_in=Bs_in, S_in=S_in, err_in=err_in, err_out=np.array(psi.norm_test()),
                            S_out=[np.array(psi.get_SL(i)) for i in range(L)], S_ent=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi],
                            Sz=np.array(psi.expectation_value('Sz'))))
            print('canonical_form_infinite', cplx, np.linalg.norm(err_in), np.linalg.norm(out[-1]['err_out']), out[-1]['chi'], out[-1]['S_ent'])
    save('canonical_form_infinite.pkl', out)


def gen_tebd_gs():
    """TEBDEngine.run_GS (imaginary time, decreasing steps) for a finite chain (update_imag sweeps) and an infinite one."""
    from tenpy.algorithms import tebd
    from tenpy.models.tf_ising import TFIChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for bc in ('finite', 'infinite'):
            L = 8 if bc == 'finite' else 2
            M = TFIChain({'L': L, 'J': 1., 'g': 1.5, 'bc_MPS': bc, 'conserve': 'parity', 'sort_charge': True})
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc=bc)
            opts = {'order': 2, 'delta_tau_list': [0.1, 0.01, 0.001], 'N_steps': 10, 'max_error_E': 1.e-9, 'trunc_params': {'chi_max': 12, 'svd_min': 1.e-10}}
            eng = tebd.TEBDEngine(psi, M, dict(opts))
            eng.run_GS()
            out.append(dict(bc=bc, L=L, options=opts, E_bonds=np.array(M.bond_energies(psi)), S=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi],
                            beta=float(-np.imag(eng.evolved_time)), h_bond=[None if h is None else h.transpose(['p0', 'p1', 'p0*', 'p1*']).to_ndarray() for h in M.H_bond],
                            state_labels=list(M.lat.mps_sites()[0].state_labels.items())))
            print('tebd_gs', bc, np.mean(out[-1]['E_bonds']), out[-1]['beta'], out[-1]['chi'])
    save('tebd_gs.pkl', out)


def gen_nocharge():
    """The newer engines WITHOUT charge conservation (TFI chain, conserve=None: one block per tensor) and with Z2 parity:
    single-site DMRG with subspace expansion, TDVP (two-site then single-site), iDMRG with the default run loop."""
    from tenpy.algorithms import dmrg, tdvp
    from tenpy.models.tf_ising import TFIChain
    from tenpy.networks.mps import MPS
    import copy
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for conserve in (None, 'parity'):
            L = 8
            M = TFIChain({'L': L, 'J': 1., 'g': 1.3, 'bc_MPS': 'finite', 'conserve': conserve, 'sort_charge': True})
            labels = list(M.lat.mps_sites()[0].state_labels.items())
            # single-site DMRG
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
            eng = dmrg.SingleSiteDMRGEngine(psi, M, {'combine': True, 'max_N_for_ED': 0, 'mixer': True,
                                                     'mixer_params': {'amplitude': 1.e-3, 'decay': 2., 'disable_after': 3},
                                                     'trunc_params': {'chi_max': 10, 'svd_min': 1.e-6}})
            eng.mixer_activate()
            Es = []
            for s_ in range(5):
                eng.sweep()
                Es.append(float(eng.update_stats['E_total'][-1]))
            eng.mixer_cleanup()
            rec = dict(conserve=conserve, L=L, J=1., g=1.3, state_labels=labels, single_E_sweeps=Es,
                       single_E_updates=[float(e) for e in eng.update_stats['E_total']], single_S=np.array(psi.entanglement_entropy()))
            # TDVP from the all-up product state
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up'] * L, bc='finite')
            topts = {'dt': 0.05, 'N_steps': 2, 'trunc_params': {'chi_max': 10, 'svd_min': 1.e-10}, 'lanczos_params': {'N_min': 2, 'N_max': 20}}
            e2 = tdvp.TwoSiteTDVPEngine(psi, M, copy.deepcopy(topts))
            steps = []
            for rep in range(3):
                e2.run()
                steps.append(dict(engine='two', S=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi], sz=np.array(psi.expectation_value('Sigmaz'))))
            e1 = tdvp.SingleSiteTDVPEngine(psi, M, copy.deepcopy(topts))
            for rep in range(2):
                e1.run()
                steps.append(dict(engine='single', S=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi], sz=np.array(psi.expectation_value('Sigmaz'))))
            rec.update(tdvp_options=topts, tdvp_steps=steps)
            # iDMRG
            Mi = TFIChain({'L': 2, 'J': 1., 'g': 1.3, 'bc_MPS': 'infinite', 'conserve': conserve, 'sort_charge': True})
            psi = MPS.from_product_state(Mi.lat.mps_sites(), ['up'] * 2, bc='infinite')
            iopts = {'mixer': None, 'max_E_err': 1.e-10, 'max_sweeps': 30, 'N_sweeps_check': 5, 'trunc_params': {'chi_max': 12, 'svd_min': 1.e-10}}
            ei = dmrg.TwoSiteDMRGEngine(psi, Mi, dict(copy.deepcopy(iopts), combine=True, max_N_for_ED=0))
            E, _ = ei.run()
            rec.update(idmrg_options=iopts, idmrg_E=float(E), idmrg_sweeps=int(ei.sweeps), idmrg_E_updates=[float(e) for e in ei.update_stats['E_total']],
                       idmrg_S=[np.array(psi.get_SL(i)) for i in range(2)])
            out.append(rec)
            print('nocharge', conserve, Es[-1], steps[-1]['chi'], E, ei.sweeps)
    save('nocharge.pkl', out)


def gen_correlations():
    """Correlation functions of a DMRG ground state (XXZ chain): Sz-Sz, S+ S-, and the state itself (B tensors) so that the
    device code is checked on identical input."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 8
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 0.6, 'hz': 0.05, 'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
        dmrg.run(psi, M, {'mixer': None, 'trunc_params': {'chi_max': 12, 'svd_min': 1.e-10}, 'max_sweeps': 6})
        site = M.lat.mps_sites()[0]
        out.append(dict(L=L, B=[dump_array(psi.get_B(i, 'B').transpose(['vL', 'p', 'vR'])) for i in range(L)],
                        S=[np.array(psi.get_SL(i)) for i in range(L)] + [np.array(psi.get_SR(L - 1))],
                        SzSz=np.array(psi.correlation_function('Sz', 'Sz')), SpSm=np.array(psi.correlation_function('Sp', 'Sm')),
                        SzSz_sub=np.array(psi.correlation_function('Sz', 'Sz', sites1=[1, 4], sites2=[0, 4, 7])),
                        Sz=site.Sz.to_ndarray(), Sp=site.Sp.to_ndarray(), Sm=site.Sm.to_ndarray(), exp_Sz=np.array(psi.expectation_value('Sz'))))
        print('correlations', out[-1]['SzSz'][0, :3])
    save('correlations.pkl', out)


def gen_mpo_evolution():
    """ExpMPOEvolution (W_II, orders 1 and 2, SVD compression) after a quench from the Neel state, plus the W_II tensors."""
    from tenpy.algorithms import mpo_evolution
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 6
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 0.8, 'hz': 0.1, 'bc_MPS': 'finite', 'sort_charge': True})
        for order in (1, 2):
            psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * (L // 2), bc='finite')
            opts = {'dt': 0.05, 'N_steps': 2, 'order': order, 'approximation': 'II', 'compression_method': 'SVD',
                    'trunc_params': {'chi_max': 10, 'svd_min': 1.e-10}}
            eng = mpo_evolution.ExpMPOEvolution(psi, M, dict(opts))
            steps = []
            for rep in range(4):
                eng.run()
                steps.append(dict(S=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi], Sz=np.array(psi.expectation_value('Sz')),
                                  norm=float(psi.norm)))
            U = M.H_MPO.make_U_II(-0.05j)
            out.append(dict(L=L, Jxx=1., Jz=0.8, hz=0.1, order=order, options=opts, steps=steps, trunc_err=float(eng.trunc_err.eps),
                            W_II=[U.get_W(i).transpose(['wL', 'wR', 'p', 'p*']).to_ndarray() for i in range(L)]))
            print('mpo_evolution', order, steps[-1]['chi'], steps[-1]['S'][L // 2 - 1], steps[-1]['norm'])
    save('mpo_evolution.pkl', out)


def gen_idmrg_single():
    """Single-site infinite DMRG with the subspace expansion (reference defaults for the run loop)."""
    from tenpy.algorithms import dmrg
    from tenpy.models.xxz_chain import XXZChain
    from tenpy.networks.mps import MPS
    import copy
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        L = 2
        M = XXZChain({'L': L, 'Jxx': 1., 'Jz': 1.5, 'hz': 0., 'bc_MPS': 'infinite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'], bc='infinite')
        opts = {'mixer': True, 'mixer_params': {'amplitude': 1.e-3, 'decay': 1.5, 'disable_after': 8}, 'combine': True, 'max_N_for_ED': 0,
                'max_E_err': 1.e-9, 'max_sweeps': 40, 'N_sweeps_check': 5, 'trunc_params': {'chi_max': 12, 'svd_min': 1.e-6}}
        opts_plain = copy.deepcopy(opts)
        eng = dmrg.SingleSiteDMRGEngine(psi, M, opts)
        E, _ = eng.run()
        st = eng.sweep_stats
        out.append(dict(E=float(E), sweeps=int(eng.sweeps), options=opts_plain, E_updates=[float(e) for e in eng.update_stats['E_total']],
                        age=[int(a) for a in eng.update_stats['age']], i0=[int(i) for i in eng.update_stats['i0']],
                        sweep_stats={k: [float(x) for x in st[k]] for k in ('sweep', 'E', 'Delta_E', 'S', 'max_chi')},
                        S=[np.array(psi.get_SL(i)) for i in range(L)], chi=[int(c) for c in psi.chi], L=L, Jxx=1., Jz=1.5, hz=0.))
        print('idmrg_single', E, eng.sweeps, psi.chi)
    save('idmrg_single.pkl', out)


def gen_hubbard2():
    """Fermi-Hubbard 2 x 3 ladder (two U(1) charges, fermionic MPO with D = 10): single-site DMRG with subspace expansion and
    two-site TDVP after a quench."""
    from tenpy.algorithms import dmrg, tdvp
    from tenpy.models.hubbard import FermiHubbardModel
    from tenpy.networks.mps import MPS
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Lx = 3
        M = FermiHubbardModel({'lattice': 'Ladder', 'L': Lx, 't': 1., 'U': 8., 'mu': 0., 'cons_N': 'N', 'cons_Sz': 'Sz',
                               'bc_MPS': 'finite', 'sort_charge': True})
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * Lx, bc='finite')
        eng = dmrg.SingleSiteDMRGEngine(psi, M, {'combine': True, 'max_N_for_ED': 0, 'mixer': True,
                                                 'mixer_params': {'amplitude': 1.e-2, 'decay': 2., 'disable_after': 4},
                                                 'trunc_params': {'chi_max': 24, 'svd_min': 1.e-6}})
        eng.mixer_activate()
        Es = []
        for s_ in range(7):
            eng.sweep()
            Es.append(float(eng.update_stats['E_total'][-1]))
        eng.mixer_cleanup()
        rec = dict(Lx=Lx, t=1., U=8., mu=0., single_E_sweeps=Es, single_S=np.array(psi.entanglement_entropy()), single_chi=[int(c) for c in psi.chi])
        psi = MPS.from_product_state(M.lat.mps_sites(), ['up', 'down'] * Lx, bc='finite')
        topts = {'dt': 0.02, 'N_steps': 2, 'trunc_params': {'chi_max': 20, 'svd_min': 1.e-10}, 'lanczos_params': {'N_min': 2, 'N_max': 20}}
        e2 = tdvp.TwoSiteTDVPEngine(psi, M, dict(topts))
        steps = []
        for rep in range(3):
            e2.run()
            steps.append(dict(S=np.array(psi.entanglement_entropy()), chi=[int(c) for c in psi.chi], n=np.array(psi.expectation_value('Ntot'))))
        rec.update(tdvp_options=topts, tdvp_steps=steps)
        out.append(rec)
        print('hubbard2', Es[-1], rec['single_chi'], steps[-1]['chi'])
    save('hubbard2.pkl', out)


GENERATORS = dict(hubbard2=gen_hubbard2, idmrg_single=gen_idmrg_single, mpo_evolution=gen_mpo_evolution, correlations=gen_correlations, nocharge=gen_nocharge, tebd_gs=gen_tebd_gs, canonical_form_infinite=gen_canonical_form_infinite, canonical_form=gen_canonical_form, tebd_infinite=gen_tebd_infinite, idmrg_bench=gen_idmrg_bench, idmrg=gen_idmrg, tdvp=gen_tdvp, dmrg_run=gen_dmrg_run, dmrg_default_diag=gen_dmrg_default_diag, dmrg_ortho=gen_dmrg_ortho, tebd2=gen_tebd2, api2=gen_api2, krylov2=gen_krylov2, dmrg_two_site_subspace=gen_dmrg_two_site_subspace, dmrg_single=gen_dmrg_single, dmrg_mixer=gen_dmrg_mixer, hubbard=gen_hubbard, mixer=gen_mixer, charges=gen_charges, tensordot=gen_tensordot, reshape=gen_reshape, linalg=gen_linalg,
                  truncate=gen_truncate, lanczos=gen_lanczos, dmrg=gen_dmrg, tebd=gen_tebd, qr_theta=gen_qr_theta)

if __name__ == '__main__':
    print("reference:", tenpy.__version__, tenpy.__file__)
    only = os.environ.get('ONLY')
    for name, fn in GENERATORS.items():
        if only is None or name in only.split(','):
            fn()
