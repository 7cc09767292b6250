"""Second-order TEBD bond update -- stand-alone harness for BASELINE config 5 (complex128, block-SVD bound) on boxes
without TeNPy; with TeNPy installed, its own ``TEBDEngine`` / ``QRBasedTEBDEngine`` (all orders, imaginary time, infinite
chains) run unchanged on the mirror (``tenpy_amd/install.py``).

Mirrors ``TEBDEngine.update_bond`` (``tenpy/algorithms/tebd.py:416-483``) and the order-2 Suzuki-Trotter
step of ``evolve_step`` (:374): the same npc call sequence (``get_theta(formL=0)`` -> ``tensordot`` with the
bond gate -> ``scale_axis`` -> ``combine_legs`` -> ``svd_theta`` -> ``split_legs`` -> ``tensordot`` with
``V.conj()``), so that no inverse Schmidt values are needed.  Bond gates ``exp(-i dt h)`` are d^2 x d^2
matrices built once on the host (``_calc_U_bond`` :585 does the same through ``npc.expm``).
"""
import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.truncation import svd_theta, svd_theta_batched, TruncationError, decompose_theta_qr_based, decompose_theta_qr_based_batched

__all__ = ['TEBDEngine', 'QRBasedTEBDEngine', 'bond_gate']


def bond_gate(h_bond_dense, p_leg, dt, imaginary=False):
    """``U = exp(-i dt h)`` (or ``exp(-dt h)``) as an npc Array with labels p0, p1, p0*, p1*.
    ``h_bond_dense`` has shape (d, d, d, d) = [p0, p1, p0*, p1*]."""
    d = p_leg.ind_len
    h = np.asarray(h_bond_dense).reshape(d * d, d * d)
    lam, V = np.linalg.eigh(h)      # d^2 x d^2 hermitian bond Hamiltonian: host setup, like the tridiagonal eigh of Lanczos
    U = ((V * np.exp((-dt if imaginary else -1.j * dt) * lam)) @ V.conj().T).reshape(d, d, d, d)
    return npc.Array.from_ndarray(U, [p_leg, p_leg, p_leg.conj(), p_leg.conj()], dtype=U.dtype,
                                  labels=['p0', 'p1', 'p0*', 'p1*'], cutoff=1e-14, raise_wrong_sector=True)


BATCH_BYTES_CAP = None       # TPA_TEBD_BATCH_GB, else 1/3 of the device's memory (96 GB on an MI355X)


def batch_bytes_cap():
    from ..linalg import _device as dev
    return BATCH_BYTES_CAP if BATCH_BYTES_CAP is not None else dev.memory_budget('TPA_TEBD_BATCH_GB', 1. / 3., 96.)


def batch_group_size(psi, bonds, itemsize=16, cap=None):
    """How many bonds of a half-step go into one batched decomposition: all of them unless their work areas would not fit.  The block
    SVD needs ~24x the bytes of theta (two images of [W | G], Gram matrix, accumulated transform, split-K partials, outputs); theta is
    bounded by its dense size (d chi_L) x (d chi_R) -- charge conservation only makes it smaller.  ``TPA_TEBD_BATCH_GB`` (default: a third of the device's memory) caps the sum."""
    cap = batch_bytes_cap() if cap is None else cap
    if not bonds:
        return 1
    worst = 0
    for i in bonds:
        d0, d1 = psi.p_legs[i - 1].ind_len, psi.p_legs[i].ind_len
        chi_l, chi_r = len(psi.get_SL(i - 1)), len(psi.get_SR(i))
        worst = max(worst, 24 * itemsize * (d0 * chi_l) * (d1 * chi_r))
    return int(max(1, min(len(bonds), cap // max(worst, 1))))


class TEBDEngine:
    def __init__(self, psi, h_bonds_dense, options):
        """``h_bonds_dense[i]`` couples sites (i-1, i) (``None`` for i = 0), shape (d, d, d, d)."""
        self.psi = psi
        self.h_bonds = h_bonds_dense
        self.options = dict(options)
        self.trunc_params = dict(self.options.get('trunc_params', {}))
        self.dt = self.options.get('dt', 0.1)
        self.trunc_err = TruncationError()
        self.norm = 1.
        self._U = {}
        self.evolved_time = 0.

    def _gates(self, frac):
        key = round(frac, 12)
        if key not in self._U:
            p = self.psi.p_legs[0]
            self._U[key] = [None if h is None else bond_gate(h, p, self.dt * frac) for h in self.h_bonds]
        return self._U[key]

    def update_bond(self, i, U_bond):
        i0, i1 = i - 1, i
        psi = self.psi
        C = psi.get_theta(i0, n=2, formL=0.)
        C = npc.tensordot(U_bond, C, axes=(['p0*', 'p1*'], ['p0', 'p1']))
        C.itranspose(['vL', 'p0', 'p1', 'vR'])
        theta = C.scale_axis(psi.get_SL(i0), 'vL')
        theta = theta.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])
        npc.svd_engine_floor = True      # a TEBD bond update truncates right after the SVD: the engines' absolute floor applies (np_conserved.SVD_ABS_FLOOR)
        U, S, V, err, renorm = svd_theta(theta, self.trunc_params, [psi.get_B(i0, None).qtotal, None],
                                         inner_labels=['vR', 'vL'])
        B_R = V.split_legs(1).ireplace_label('p1', 'p')
        B_L = npc.tensordot(C.combine_legs(('p1', 'vR'), pipes=theta.legs[1]), V.conj(),
                            axes=['(p1.vR)', '(p1*.vR*)'])
        B_L.ireplace_labels(['vL*', 'p0'], ['vR', 'p'])
        B_L.iscale_prefactor(1. / renorm)
        self.norm *= renorm
        psi.set_SR(i0, S)
        psi.set_B(i0, B_L, form='B')
        psi.set_B(i1, B_R, form='B')
        self.trunc_err = self.trunc_err + err
        return err

    # The bonds of one half-step do not share a site (reference tebd.py:374-414 loops over ``np.arange(int(odd) % 2, L, 2)``), so their
    # block SVDs can be decomposed in ONE batched device call (``np_conserved.svd_batched``; ``options['batch_bonds'] = True``; same
    # numbers bond by bond, tests/test_tebd_golden.py; an integer k = groups of k bonds).  Measured on the MI355X (round 4, TFI L = 64,
    # chi = 1024 complex: 2 blocks of 1024 x 1024 per bond), s per time step:
    #   8-row-block complex rounds (round 3):   bond by bond 5.37, whole half-step 6.60 (bandwidth-bound: every round streams the block)
    #   Gram-only sweeps on 32-row blocks (csrc/tpa_svd_b32c.inc): bond by bond 4.31, groups of 4 / 8 / 16: 2.62 / 2.49 / 2.44,
    #   whole half-step (32 bonds, 64 blocks in one tpa_svd_batch): 2.34 -- the solve of a round occupies 32 CUs per bond and the
    #   pivoted-QR chain is latency-bound, both are shared by the whole group; the MFMA tile updates scale with the group.
    # Same truncation error and entropies to the last digit (bench line `tebd_parity`).  On by default.
    batch_bonds_default = True

    def update_bonds_batched(self, bonds, U):
        self._commit_bonds(self._decompose_bonds(bonds, U))

    def _decompose_bonds(self, bonds, U):
        """New tensors of the given (independent) bonds from the CURRENT state, nothing written back:
        ``[(i, S, B_L, B_R, renormalization, trunc_err), ...]`` (``algorithms/sharded.ShardedTEBDEngine`` deals the bonds of a
        half-step over the ranks with this)."""
        psi = self.psi
        Cs, thetas, qLRs = [], [], []
        for i in bonds:
            i0 = i - 1
            C = psi.get_theta(i0, n=2, formL=0.)
            C = npc.tensordot(U[i], C, axes=(['p0*', 'p1*'], ['p0', 'p1']))
            C.itranspose(['vL', 'p0', 'p1', 'vR'])
            theta = C.scale_axis(psi.get_SL(i0), 'vL')
            theta = theta.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])
            Cs.append(C)
            thetas.append(theta)
            qLRs.append([psi.get_B(i0, None).qtotal, None])
        if not bonds:
            return []
        npc.svd_engine_floor = True
        res = svd_theta_batched(thetas, self.trunc_params, qLRs, inner_labels=['vR', 'vL'])
        out = []
        for i, C, theta, (Um, S, V, err, renorm) in zip(bonds, Cs, thetas, res):
            B_R = V.split_legs(1).ireplace_label('p1', 'p')
            B_L = npc.tensordot(C.combine_legs(('p1', 'vR'), pipes=theta.legs[1]), V.conj(), axes=['(p1.vR)', '(p1*.vR*)'])
            B_L.ireplace_labels(['vL*', 'p0'], ['vR', 'p'])
            B_L.iscale_prefactor(1. / renorm)
            out.append((i, S, B_L, B_R, renorm, err))
        return out

    def _commit_bonds(self, results):
        psi = self.psi
        for i, S, B_L, B_R, renorm, err in results:
            i0, i1 = i - 1, i
            self.norm *= renorm
            psi.set_SR(i0, S)
            psi.set_B(i0, B_L, form='B')
            psi.set_B(i1, B_R, form='B')
            self.trunc_err = self.trunc_err + err

    def evolve_step_order2(self):
        """One time step dt: half step on even bonds, full step on odd bonds, half step on even bonds."""
        L = self.psi.L
        batch = self.options.get('batch_bonds', self.batch_bonds_default)
        for frac, parity in ((0.5, 0), (1.0, 1), (0.5, 0)):
            U = self._gates(frac)
            bonds = [i for i in range(1, L) if i % 2 == (1 - parity)]        # bond (i-1, i) with even i-1 <=> parity 0
            if batch:
                # True: the whole half-step (as far as the work areas fit, batch_group_size); k: k bonds per device call
                group = batch_group_size(self.psi, bonds) if batch is True else max(int(batch), 1)
                for g0 in range(0, len(bonds), group):
                    self.update_bonds_batched(bonds[g0:g0 + group], U)
            else:
                for i in bonds:
                    self.update_bond(i, U[i])
        self.evolved_time += self.dt

    def run(self, n_steps):
        for _ in range(n_steps):
            self.evolve_step_order2()


class QRBasedTEBDEngine(TEBDEngine):
    """TEBD with the QR-based truncation (reference ``QRBasedTEBDEngine.update_bond``, tebd.py:685-738; options
    ``cbe_expand`` (0.1), ``cbe_expand_0``, ``cbe_min_block_increase`` (1), ``use_eig_based_svd``, ``compute_err``)."""
    # the bond matrices Xi of a half-step are decomposed in one batched block SVD, or -- `use_eig_based_svd` -- one batched Hermitian
    # eigen-decomposition (truncation.decompose_theta_qr_based_batched)
    batch_bonds_default = True

    def update_bonds_batched(self, bonds, U):
        psi = self.psi
        Cs, items = [], []
        for i in bonds:
            i0, i1 = i - 1, i
            expand = self._expansion_rate(i)
            C = psi.get_theta(i0, n=2, formL=0.)
            C = npc.tensordot(U[i], C, axes=(['p0*', 'p1*'], ['p0', 'p1']))
            C.itranspose(['vL', 'p0', 'p1', 'vR'])
            theta = C.scale_axis(psi.get_SL(i0), 'vL')
            theta = theta.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])
            old_B_L, old_B_R = psi.get_B(i0, 'B'), psi.get_B(i1, 'B')
            Cs.append(C)
            items.append((old_B_L.qtotal, old_B_R.qtotal, old_B_R.get_leg('vL'), theta, False, expand,
                          self.options.get('cbe_min_block_increase', 1)))
        res = decompose_theta_qr_based_batched(items, self.trunc_params, self.options.get('compute_err', True), False,
                                               use_eig_based_svd=self.options.get('use_eig_based_svd', False))
        for i, C, it, (_, S, B_R, form, err, renorm) in zip(bonds, Cs, items, res):
            i0, i1 = i - 1, i
            assert form[1] == 'B'
            B_L = npc.tensordot(C.combine_legs(('p1', 'vR'), pipes=it[3].legs[1]), B_R.conj(), axes=[['(p1.vR)'], ['(p*.vR*)']])
            B_L.iscale_prefactor(1. / renorm)
            B_L.ireplace_labels(['p0', 'vL*'], ['p', 'vR'])
            B_R = B_R.split_legs(1)
            self.norm *= renorm
            psi.set_B(i0, B_L, form='B')
            psi.set_SL(i1, S)
            psi.set_B(i1, B_R, form='B')
            self.trunc_err = self.trunc_err + err

    def _expansion_rate(self, i):
        expand = self.options.get('cbe_expand', 0.1)
        expand_0 = self.options.get('cbe_expand_0', None)
        if expand_0 is None or expand_0 == expand:
            return expand
        chi_max = self.trunc_params.get('chi_max', None)
        if chi_max is None:
            raise ValueError('Need to specify trunc_params["chi_max"] in order to use cbe_expand_0.')
        chi = len(self.psi.get_SL(i))
        return max(expand_0 - chi / chi_max * (expand_0 - expand), expand)

    def update_bond(self, i, U_bond):
        i0, i1 = i - 1, i
        psi = self.psi
        expand = self._expansion_rate(i)
        C = psi.get_theta(i0, n=2, formL=0.)
        C = npc.tensordot(U_bond, C, axes=(['p0*', 'p1*'], ['p0', 'p1']))
        C.itranspose(['vL', 'p0', 'p1', 'vR'])
        theta = C.scale_axis(psi.get_SL(i0), 'vL')
        theta = theta.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])
        old_B_L, old_B_R = psi.get_B(i0, 'B'), psi.get_B(i1, 'B')
        _, S, B_R, form, err, renorm = decompose_theta_qr_based(
            old_qtotal_L=old_B_L.qtotal, old_qtotal_R=old_B_R.qtotal, old_bond_leg=old_B_R.get_leg('vL'), theta=theta,
            move_right=False, expand=expand, min_block_increase=self.options.get('cbe_min_block_increase', 1),
            use_eig_based_svd=self.options.get('use_eig_based_svd', False), trunc_params=self.trunc_params,
            compute_err=self.options.get('compute_err', True), return_both_T=False)
        assert form[1] == 'B'
        B_L = npc.tensordot(C.combine_legs(('p1', 'vR'), pipes=theta.legs[1]), B_R.conj(), axes=[['(p1.vR)'], ['(p*.vR*)']])
        B_L.iscale_prefactor(1. / renorm)
        B_L.ireplace_labels(['p0', 'vL*'], ['p', 'vR'])
        B_R = B_R.split_legs(1)
        self.norm *= renorm
        psi.set_B(i0, B_L, form='B')
        psi.set_SL(i1, S)
        psi.set_B(i1, B_R, form='B')
        self.trunc_err = self.trunc_err + err
        return err
