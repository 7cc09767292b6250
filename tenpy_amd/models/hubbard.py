"""Fermi-Hubbard ladder MPO (BASELINE config 4: 2 x Lx ladder, U(1) x U(1) charges (N, 2 Sz), MPO bond dimension
10, many small charge blocks), built directly as a finite-state machine -- setup code; the reference goes through
``FermiHubbardModel`` / ``CouplingMPOModel`` (models/hubbard.py:132).

    H = -t sum_{<ij>, s} (c^dag_{i s} c_{j s} + h.c.) + U sum_i n_{i up} n_{i down} - mu sum_i n_i

Ladder site (x, y) -> chain index 2 x + y: rungs couple chain neighbours (2x, 2x+1), legs couple (s, s+2).
Jordan-Wigner: c_{j s} = (prod_{l<j} F_l) a_{j s} with site-local a_up = a (x) 1, a_down = F_up (x) a.
"""
import numpy as np

from ..linalg.charges import ChargeInfo, LegCharge
from ..networks.mpo import mpo_from_dense

__all__ = ['spinful_fermion_leg', 'hubbard_ladder_mpo', 'hubbard_ops']


def hubbard_ops():
    """Local operators in the basis (empty, up, down, full); ``full = c^dag_up c^dag_down |0>``."""
    a = np.array([[0., 1.], [0., 0.]])          # annihilator of one mode in basis (0, 1)
    F1 = np.diag([1., -1.])
    I2 = np.eye(2)
    # mode order: up (x) down ; basis index = 2*n_up + n_down -> (00, 01, 10, 11) = (empty, down, up, full)
    Cu = np.kron(a, I2)
    Cd = np.kron(F1, a)
    perm = [0, 2, 1, 3]                           # reorder to (empty, up, down, full)
    def P(op):
        return op[np.ix_(perm, perm)]
    Cu, Cd = P(Cu), P(Cd)
    ops = dict(Cu=Cu, Cd=Cd, Cdu=Cu.T.copy(), Cdd=Cd.T.copy(), Id=np.eye(4))
    ops['Nu'], ops['Nd'] = ops['Cdu'] @ Cu, ops['Cdd'] @ Cd
    ops['Ntot'] = ops['Nu'] + ops['Nd']
    ops['NuNd'] = ops['Nu'] @ ops['Nd']
    ops['JW'] = np.diag(np.exp(1j * np.pi * np.diag(ops['Ntot'])).real)
    return ops


def spinful_fermion_leg():
    """Physical leg with charges (N, 2Sz) for (empty, up, down, full)."""
    chinfo = ChargeInfo([1, 1], ['N', '2*Sz'])
    leg = LegCharge.from_qflat(chinfo, [[0, 0], [1, 1], [1, -1], [2, 0]])
    return chinfo, leg


def hubbard_ladder_mpo(Lx, t=1., U=8., mu=0., Ly=2):
    """MPO of the Ly=2 ladder of length Lx (2 Lx chain sites), D = 10."""
    assert Ly == 2
    chinfo, p = spinful_fermion_leg()
    o = hubbard_ops()
    N = 2 * Lx
    JW = o['JW']
    # hopping terms  c^dag_i c_j = (a^dag_i F_i) F.. a_j   and   c^dag_j c_i = (F_i a_i) F.. a^dag_j   (i < j)
    first = [o['Cdu'] @ JW, JW @ o['Cu'], o['Cdd'] @ JW, JW @ o['Cd']]
    second = [o['Cu'], o['Cdu'], o['Cd'], o['Cdd']]
    D = 10          # 0: IdL, 1-4: term k waiting for its partner, 5-8: term k after one JW site, 9: IdR
    Ws = []
    for s in range(N):
        W = np.zeros((D, D, 4, 4))
        W[0, 0] = o['Id']
        W[9, 9] = o['Id']
        W[0, 9] = U * o['NuNd'] - mu * o['Ntot']
        for k in range(4):
            W[0, 1 + k] = first[k]
            W[1 + k, 5 + k] = JW
            if s % 2 == 1:                      # rung: partner of the operator placed on site s-1 (even)
                W[1 + k, 9] = -t * second[k]
            W[5 + k, 9] = -t * second[k]        # leg: partner of the operator placed on site s-2
        Ws.append(W)
    # an operator placed on an odd site must not find a rung partner on the next (even) site: handled above
    # (rung entry only on odd s).  Open boundaries: first / last tensors are the IdL row / IdR column.
    Ws[0] = Ws[0][0:1]
    Ws[-1] = Ws[-1][:, 9:10]
    H = mpo_from_dense(Ws, [p] * N, chinfo)
    H.IdL, H.IdR = 0, -1
    return H
