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


def xxz_chain_mpo(L, Jxx=1., Jz=1., hz=0., conserve='Sz', bc='finite'):
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
    if bc == 'infinite':
        return mpo_from_dense([W] * L, [p] * L, chinfo, IdL=0, IdR=-1, bc='infinite')
    Ws = [W[0:1] if i == 0 else (W[:, 4:5] if i == L - 1 else W) for i in range(L)]
    H = mpo_from_dense(Ws, [p] * L, chinfo)
    H.IdL, H.IdR = 0, -1
    return H


def tfi_chain_mpo(L, J=1., g=1., conserve=None, bc='finite'):
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
    if bc == 'infinite':
        return mpo_from_dense([W] * L, [p] * L, chinfo, IdL=0, IdR=-1, bc='infinite')
    Ws = [W[0:1] if i == 0 else (W[:, 2:3] if i == L - 1 else W) for i in range(L)]
    H = mpo_from_dense(Ws, [p] * L, chinfo)
    H.IdL, H.IdR = 0, -1
    return H


def spin_S_leg(S, conserve='Sz'):
    """Physical leg of a spin-S site, states ordered m = -S ... S (charge 2 m for ``conserve='Sz'``)."""
    d = int(round(2 * S + 1))
    if conserve == 'Sz':
        chinfo = ChargeInfo([1], ['2*Sz'])
        leg = LegCharge.from_qflat(chinfo, [[int(round(2 * (-S + k)))] for k in range(d)])
    elif conserve == 'parity':
        chinfo = ChargeInfo([2], ['parity_Sz'])
        leg = LegCharge.from_qflat(chinfo, [[k % 2] for k in range(d)])
    else:
        chinfo = ChargeInfo()
        leg = LegCharge.from_trivial(d, chinfo)
    return chinfo, leg


def spin_chain_mpo(L, S=0.5, Jx=1., Jy=1., Jz=1., D=0., hz=0., conserve='Sz', bc='finite'):
    """Spin-S chain ``sum_i Jx SxSx + Jy SySy + Jz SzSz + D (Sz)^2 - hz Sz`` -- the model of the reference's iDMRG
    benchmark (``tests/benchmark/dmrg_infinite.py:22``: ``SpinChain(S=2, D=0.3, bc_MPS='infinite')``, models/spins.py).
    ``conserve='Sz'`` needs Jx = Jy."""
    chinfo, p = spin_S_leg(S, conserve)
    d = p.ind_len
    m = -S + np.arange(d)
    Sz = np.diag(m)
    Sp = np.zeros((d, d))
    for k in range(d - 1):
        Sp[k + 1, k] = np.sqrt(S * (S + 1) - m[k] * (m[k] + 1))
    Sm = Sp.T.copy()
    Id = np.eye(d)
    if conserve == 'Sz' and Jx != Jy:
        raise ValueError("Sz conservation needs Jx == Jy")
    Jpm, Jpp = 0.25 * (Jx + Jy), 0.25 * (Jx - Jy)          # Jx SxSx + Jy SySy = Jpm (S+S- + S-S+) + Jpp (S+S+ + S-S-)
    Dm = 5
    W = np.zeros((Dm, Dm, d, d))
    W[0, 0] = Id
    W[0, 1] = Sp
    W[0, 2] = Sm
    W[0, 3] = Sz
    W[0, 4] = D * (Sz @ Sz) - hz * Sz
    W[1, 4] = Jpm * Sm + Jpp * Sp
    W[2, 4] = Jpm * Sp + Jpp * Sm
    W[3, 4] = Jz * Sz
    W[4, 4] = Id
    if bc == 'infinite':
        return mpo_from_dense([W] * L, [p] * L, chinfo, IdL=0, IdR=-1, bc='infinite')
    Ws = [W[0:1] if i == 0 else (W[:, 4:5] if i == L - 1 else W) for i in range(L)]
    return mpo_from_dense(Ws, [p] * L, chinfo, IdL=0, IdR=-1)


def spin_chain_h_bonds(L, S=0.5, Jx=1., Jy=1., Jz=1., D=0., hz=0., bc='finite'):
    """Two-site terms ``h[i]`` coupling sites (i-1, i) of :func:`spin_chain_mpo` as dense (d, d, d, d) arrays
    [p0, p1, p0*, p1*] for TEBD; ``h[0]`` is ``None`` for a finite chain.  On-site terms are shared half / half between
    the two bonds of a site, a boundary site of a finite chain gives all of it to its only bond (reference
    ``NearestNeighborModel.calc_H_bond``, models/model.py)."""
    d = int(round(2 * S + 1))
    m = -S + np.arange(d)
    Sz = np.diag(m)
    Sp = np.zeros((d, d))
    for k in range(d - 1):
        Sp[k + 1, k] = np.sqrt(S * (S + 1) - m[k] * (m[k] + 1))
    Sm = Sp.T.copy()
    Id = np.eye(d)
    Sx, Sy = 0.5 * (Sp + Sm), -0.5j * (Sp - Sm)
    onsite = D * (Sz @ Sz) - hz * Sz
    two = Jx * np.kron(Sx, Sx) + np.real(Jy * np.kron(Sy, Sy)) + Jz * np.kron(Sz, Sz)
    res = []
    for i in range(L):
        if i == 0 and bc == 'finite':
            res.append(None)
            continue
        j = (i - 1) % L
        wl = 1. if (bc == 'finite' and j == 0) else 0.5
        wr = 1. if (bc == 'finite' and i == L - 1) else 0.5
        h = two + wl * np.kron(onsite, Id) + wr * np.kron(Id, onsite)
        res.append(np.real_if_close(h).reshape(d, d, d, d))
    return res
