"""MPOs of the BASELINE configs, built directly as dense W tensors (setup code, not performance
relevant; the reference builds them through CouplingMPOModel, models/xxz_chain.py:53, tf_ising.py).

* XXZ / Heisenberg chain, Sz conserved (configs 2-3):
  ``H = sum_i Jxx (Sx_i Sx_{i+1} + Sy_i Sy_{i+1}) + Jz Sz_i Sz_{i+1} - hz sum_i Sz_i``, MPO bond dimension 5.
* Transverse field Ising chain (configs 1, 5): ``H = -J sum sigma^x_i sigma^x_{i+1} - g sum sigma^z_i``,
  conserve None or 'parity', MPO bond dimension 3.
"""
import numpy as np

from ..linalg.charges import ChargeInfo, LegCharge
from ..networks.mpo import mpo_from_dense

__all__ = ['xxz_chain_mpo', 'tfi_chain_mpo', 'spin_half_leg']


def spin_half_leg(conserve='Sz'):
    """Physical leg of a spin-1/2 site; index 0 = 'down', 1 = 'up' (sorted by charge like the reference's
    SpinHalfSite after its charge sort)."""
    if conserve == 'Sz':
        chinfo = ChargeInfo([1], ['2*Sz'])
        leg = LegCharge.from_qflat(chinfo, [[-1], [1]])
    elif conserve == 'parity':
        chinfo = ChargeInfo([2], ['parity_Sz'])
        leg = LegCharge.from_qflat(chinfo, [[0], [1]])   # down: 0, up: 1
    else:
        chinfo = ChargeInfo()
        leg = LegCharge.from_trivial(2, chinfo)
    return chinfo, leg


def xxz_chain_mpo(L, Jxx=1., Jz=1., hz=0., conserve='Sz'):
    chinfo, p = spin_half_leg(conserve)
    Sp = np.array([[0., 0.], [1., 0.]])    # |up><down| in (down, up) basis: Sp[1,0] = 1
    Sm = Sp.T.copy()
    Sz = np.diag([-0.5, 0.5])
    Id = np.eye(2)
    D = 5
    W = np.zeros((D, D, 2, 2))
    W[0, 0] = Id
    W[0, 1] = Sp
    W[0, 2] = Sm
    W[0, 3] = Sz
    W[0, 4] = -hz * Sz
    W[1, 4] = 0.5 * Jxx * Sm
    W[2, 4] = 0.5 * Jxx * Sp
    W[3, 4] = Jz * Sz
    W[4, 4] = Id
    Ws = [W[0:1] if i == 0 else (W[:, 4:5] if i == L - 1 else W) for i in range(L)]
    H = mpo_from_dense(Ws, [p] * L, chinfo)
    H.IdL, H.IdR = 0, -1
    return H


def tfi_chain_mpo(L, J=1., g=1., conserve=None):
    chinfo, p = spin_half_leg('parity' if conserve == 'parity' else None)
    sx = np.array([[0., 1.], [1., 0.]])
    sz = np.diag([-1., 1.])      # (down, up)
    Id = np.eye(2)
    D = 3
    W = np.zeros((D, D, 2, 2))
    W[0, 0] = Id
    W[0, 1] = sx
    W[0, 2] = -g * sz
    W[1, 2] = -J * sx
    W[2, 2] = Id
    Ws = [W[0:1] if i == 0 else (W[:, 2:3] if i == L - 1 else W) for i in range(L)]
    H = mpo_from_dense(Ws, [p] * L, chinfo)
    H.IdL, H.IdR = 0, -1
    return H
