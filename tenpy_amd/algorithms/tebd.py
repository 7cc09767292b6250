"""Second-order TEBD bond update -- harness for BASELINE config 5 (complex128, block-SVD bound).

Mirrors ``TEBDEngine.update_bond`` (``tenpy/algorithms/tebd.py:416-483``) and the order-2 Suzuki-Trotter
step of ``evolve_step`` (:374): the same npc call sequence (``get_theta(formL=0)`` -> ``tensordot`` with the
bond gate -> ``scale_axis`` -> ``combine_legs`` -> ``svd_theta`` -> ``split_legs`` -> ``tensordot`` with
``V.conj()``), so that no inverse Schmidt values are needed.  Bond gates ``exp(-i dt h)`` are d^2 x d^2
matrices built once on the host (``_calc_U_bond`` :585 does the same through ``npc.expm``).
"""
import numpy as np

from ..linalg import np_conserved as npc
from ..linalg.truncation import svd_theta, TruncationError, decompose_theta_qr_based

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

    # ---- general Suzuki-Trotter evolution (reference tebd.py:183-414) -------------------------------------------------
    @staticmethod
    def suzuki_trotter_time_steps(order):
        """Fractions of dt for which bond gates are needed (reference :183)."""
        if order == 1:
            return [1.]
        if order == 2:
            return [0.5, 1.]
        if order == 4:
            t1 = 1. / (4. - 4.**(1 / 3.))
            t3 = 1. - 4. * t1
            return [t1 / 2., t1, (t1 + t3) / 2., t3]
        if order == '4_opt':            # Eq. (30a) of Barthel & Zhang 2020
            a1, b1 = 0.095848502741203681182, 0.42652466131587616168
            a2, b2 = -0.078111158921637922695, -0.12039526945509726545
            return [a1, b1, a2, b2, 0.5 - a1 - a2, 1. - 2 * (b1 + b2), 2 * a1]
        raise ValueError("Unknown order %r for Suzuki Trotter decomposition" % (order,))

    @staticmethod
    def suzuki_trotter_decomposition(order, N_steps):
        """List of (index into the time steps, 0 = even / 1 = odd bonds) for ``N_steps`` steps, with the last layer of
        one step merged into the first layer of the next (reference :219)."""
        even, odd = 0, 1
        if N_steps == 0:
            return []
        if order == 1:
            return [(0, odd), (0, even)] * N_steps
        if order == 2:
            a, a2, b = (0, odd), (1, odd), (1, even)
            return [a, b] + [a2, b] * (N_steps - 1) + [a]
        if order == 4:
            a, a2, b, c, d = (0, odd), (1, odd), (1, even), (2, odd), (3, even)
            steps = [a, b, a2, b, c, d, c, b, a2, b]
            return steps + [a2, b, a2, b, c, d, c, b, a2, b] * (N_steps - 1) + [a]
        if order == '4_opt':
            a1, b1, a2, b2, a3, b3, a1_twice = (0, odd), (1, even), (2, odd), (3, even), (4, odd), (5, even), (6, odd)
            steps = [a1, b1, a2, b2, a3, b3, a3, b2, a2, b1]
            return steps + [a1_twice, b1, a2, b2, a3, b3, a3, b2, a2, b1] * (N_steps - 1) + [a1]
        raise ValueError("Unknown order %r for Suzuki Trotter decomposition" % (order,))

    def calc_U(self, order, delta_t, type_evo='real'):
        """Bond gates ``exp(-i delta_t f h)`` (``type_evo='real'``) or ``exp(-delta_t f h)`` (``'imag'``) for every fraction
        f of the decomposition (reference :297); kept until the parameters change."""
        if type_evo not in ('real', 'imag'):
            raise ValueError("Invalid value for `type_evo`: " + repr(type_evo))
        param = dict(order=order, delta_t=delta_t, type_evo=type_evo)
        if getattr(self, '_U_param', None) is not None and all(self._U_param.get(k) == v for k, v in param.items()):
            return
        param['tau'] = delta_t if type_evo == 'real' else -1.j * delta_t
        self._U_param = param
        p = self.psi.p_legs[0]
        self._U_list = [[None if h is None else bond_gate(h, p, delta_t * f, imaginary=(type_evo == 'imag')) for h in self.h_bonds]
                        for f in self.suzuki_trotter_time_steps(order)]

    def evolve(self, N_steps, dt=None):
        """``N_steps`` time steps with the gates prepared by :meth:`calc_U` (reference :346).  Returns the truncation error."""
        if dt is not None:
            assert dt == self._U_param['delta_t']
        trunc_err = TruncationError()
        for U_idx_dt, odd in self.suzuki_trotter_decomposition(self._U_param['order'], N_steps):
            trunc_err = trunc_err + self.evolve_step(U_idx_dt, odd)
        self.evolved_time = self.evolved_time + N_steps * self._U_param['tau']
        return trunc_err

    def evolve_step(self, U_idx_dt, odd):
        """One layer: all even (``odd=0``) or odd bonds ``(i-1, i)`` (reference :374)."""
        Us = self._U_list[U_idx_dt]
        trunc_err = TruncationError()
        for i_bond in range(int(odd) % 2, self.psi.L, 2):
            if Us[i_bond] is None:
                continue
            trunc_err = trunc_err + self.update_bond(i_bond, Us[i_bond])
        return trunc_err

    # ---- imaginary time evolution towards the ground state (reference :113-180, :485-583) -------------------------------
    def update_bond_imag(self, i, U_bond):
        """Bond update that keeps the A - S - B form (no old Schmidt values are used), for sweeping left and right with
        non-unitary gates (reference :545)."""
        i0, i1 = i - 1, i
        psi = self.psi
        theta = psi.get_theta(i0, n=2)
        theta = npc.tensordot(U_bond, theta, axes=(['p0*', 'p1*'], ['p0', 'p1']))
        theta.itranspose(['vL', 'p0', 'p1', 'vR'])
        theta = theta.combine_legs([('vL', 'p0'), ('p1', 'vR')], qconj=[+1, -1])
        U, S, V, err, renorm = svd_theta(theta, self.trunc_params, inner_labels=['vR', 'vL'])
        self.norm *= renorm
        psi.set_SR(i0, S)
        psi.set_B(i0, U.split_legs(0).ireplace_label('p0', 'p'), form='A')
        psi.set_B(i1, V.split_legs(1).ireplace_label('p1', 'p'), form='B')
        self.trunc_err = self.trunc_err + err
        return err

    def update_imag(self, N_steps, call_canonical_form=True):
        """``N_steps`` second-order imaginary time steps of a finite chain as sweeps right and left with the half-step gates
        (reference :485)."""
        if self._U_param['order'] != 2 or not self.psi.finite:
            raise NotImplementedError("Use DMRG instead...")
        Us = self._U_list[0]            # gates for dt / 2
        trunc_err = TruncationError()
        for _ in range(N_steps):
            for i_bond in list(range(self.psi.L)) + list(range(self.psi.L - 1, -1, -1)):
                if Us[i_bond] is not None:
                    trunc_err = trunc_err + self.update_bond_imag(i_bond, Us[i_bond])
        self.evolved_time = self.evolved_time + N_steps * self._U_param['tau']
        if call_canonical_form:
            self.psi.canonical_form()
        return trunc_err

    def bond_energies(self):
        """``<h_bond>`` of every bond term; entry i is the bond (i, i+1) (reference ``NearestNeighborModel.bond_energies``)."""
        psi = self.psi
        res = []
        order = list(range(1, psi.L)) + ([0] if not psi.finite else [])
        for i in order:
            h = self.h_bonds[i]
            if h is None:
                continue
            p0, p1 = psi.p_legs[(i - 1) % psi.L], psi.p_legs[i % psi.L]
            H2 = npc.Array.from_ndarray(np.asarray(h), [p0, p1, p0.conj(), p1.conj()], labels=['p0', 'p1', 'p0*', 'p1*'], cutoff=1e-14)
            theta = psi.get_theta(i - 1, n=2)
            C = npc.tensordot(H2, theta, axes=(['p0*', 'p1*'], ['p0', 'p1']))
            res.append(np.real(npc.inner(theta, C, axes='labels', do_conj=True)))
        return np.array(res)

    def run_GS(self):
        """Imaginary time evolution with decreasing time steps until the mean bond energy stops changing (reference :113).
        Options: ``delta_tau_list``, ``max_error_E`` (1e-13), ``N_steps`` (10), ``order`` (2)."""
        opt = self.options
        delta_tau_list = opt.get('delta_tau_list', [0.1, 0.01, 0.001, 1.e-4, 1.e-5, 1.e-6, 1.e-7, 1.e-8, 1.e-9, 1.e-10, 1.e-11, 0.])
        max_error_E = opt.get('max_error_E', 1.e-13)
        N_steps = opt.get('N_steps', 10)
        order = opt.get('order', 2)
        Eold = float(np.mean(self.bond_energies()))
        for delta_tau in delta_tau_list:
            self.calc_U(order, delta_tau, type_evo='imag')
            DeltaE = 2 * max_error_E
            while DeltaE > max_error_E:
                if self.psi.finite and order == 2:
                    self.update_imag(N_steps, call_canonical_form=False)
                else:
                    self.evolve(N_steps, delta_tau)
                E = float(np.mean(self.bond_energies()))
                DeltaE = abs(Eold - E)
                Eold = E
        return Eold

    def run_evolution(self):
        """What the reference's ``TEBDEngine.run()`` does: ``options['N_steps']`` (1) steps of ``options['dt']`` at
        ``options['order']`` (2), real time."""
        self.calc_U(self.options.get('order', 2), self.dt, 'real')
        return self.evolve(self.options.get('N_steps', 1), self.dt)

    def evolve_step_order2(self):
        """One time step dt: half step on even bonds, full step on odd bonds, half step on even bonds."""
        L = self.psi.L
        for frac, parity in ((0.5, 0), (1.0, 1), (0.5, 0)):
            U = self._gates(frac)
            for i in range(1, L):
                if i % 2 == (1 - parity):        # bond (i-1, i) with even i-1 <=> parity 0
                    self.update_bond(i, U[i])
        self.evolved_time += self.dt

    def run(self, n_steps):
        for _ in range(n_steps):
            self.evolve_step_order2()


class QRBasedTEBDEngine(TEBDEngine):
    """TEBD with the QR-based truncation (reference ``QRBasedTEBDEngine.update_bond``, tebd.py:685-738; options
    ``cbe_expand`` (0.1), ``cbe_expand_0``, ``cbe_min_block_increase`` (1), ``use_eig_based_svd``, ``compute_err``)."""

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
