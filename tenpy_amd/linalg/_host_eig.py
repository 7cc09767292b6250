"""General (non-hermitian) eigenproblems of the ``np_conserved`` interface: ``eig``, ``eigvals``, ``speigs``.

NOT part of the accelerated path.  SURVEY.md 8(a) lists what a DMRG / TEBD bond update executes -- tensordot, inner,
axpy / norm, svd, qr, eigh, combine / split -- and all of that runs in the HIP kernels behind ``include/tenpy_amd.h``.
A non-symmetric eigen-solver (Hessenberg reduction + shifted QR, or Arnoldi with implicit restarts) is a different
kernel family that none of those callers needs; TeNPy itself reaches these three functions only from diagnostics of
infinite MPS (``npc.eigvals`` / ``npc.speigs`` of small transfer-matrix-like operators).  They are provided so that the
mirror is a complete stand-in for ``tenpy.linalg.np_conserved`` (``tenpy_amd/install.py``), and they do what the
reference does (np_conserved.py:3937, :4000, :4024, worker :5041-5096): LAPACK ``geev`` / ARPACK on one charge block at a
time -- here on a host copy of the block, with the eigenvectors uploaded again.  Like every other entry point they need
the device library (no GPU -> ``BackendError``): there is no CPU-only mode of the backend.
"""
import numpy as np

from . import _device as dev
from . import np_conserved as npc
from .charges import LegPipe


def _prepare(a):
    if a.rank != 2 or a.shape[0] != a.shape[1]:
        raise ValueError("expect a square matrix!")
    a.legs[0].test_contractible(a.legs[1])
    if np.any(a.qtotal != a.chinfo.make_valid()):
        raise ValueError("Non-trivial qtotal -> Nilpotent. Not diagonizable!?")
    dev.lib()                   # fail loudly without the device library / a GPU
    return a.as_completely_blocked()


def eigvals(a, sort=None):
    _, a = _prepare(a)
    leg = a.legs[0]
    w = np.zeros(a.shape[0], dtype=np.complex128)
    for row, blk in zip(a._qdata, a._data):
        wb = np.linalg.eigvals(blk)
        if sort is not None:
            wb = wb[npc._argsort(wb, sort)]
        w[leg.get_slice(row[0])] = wb
    return w


def eig(a, sort=None):
    labels = a._labels
    piped_axes, a = _prepare(a)
    leg = a.legs[0]
    w = np.zeros(a.shape[0], dtype=np.complex128)
    sizes = leg.get_block_sizes()
    vecs = {q: np.eye(int(sizes[q]), dtype=np.complex128) for q in range(leg.block_number)}
    for row, blk in zip(a._qdata, a._data):
        wb, vb = np.linalg.eig(blk)
        if sort is not None:
            p = npc._argsort(wb, sort)
            wb, vb = wb[p], vb[:, p]
        w[leg.get_slice(row[0])] = wb
        vecs[int(row[0])] = vb
    right = leg.to_LegCharge().conj() if isinstance(leg, LegPipe) else leg.conj()
    V = npc.Array([leg, right], np.complex128)
    if leg.block_number:
        qd = np.arange(leg.block_number, dtype=np.intp)
        flat = np.concatenate([vecs[q].reshape(-1) for q in range(leg.block_number)])
        V._set_blocks(np.stack([qd, qd], axis=1), arena=dev.to_device(flat), qdata_sorted=True)
    if len(piped_axes) > 0:
        V = V.split_legs(0)
    V.iset_leg_labels([labels[0], 'eig'])
    return w, V


def speigs(a, charge_sector, k, *args, **kwargs):
    import scipy.sparse.linalg
    charge_sector = a.chinfo.make_valid(charge_sector).reshape((a.chinfo.qnumber,))
    ret_eigv = kwargs.get('return_eigenvectors', args[7] if len(args) > 7 else True)
    piped_axes, a = _prepare(a)
    leg = a.legs[0]
    sector = [q for q in range(leg.block_number) if np.all(a.chinfo.make_valid(leg.get_charge(q)) == charge_sector)]
    if len(sector) == 0:
        raise ValueError("desired charge sector not present in the leg of `a`")
    qi = sector[0]
    size = int(leg.get_block_sizes()[qi])
    blk = a.get_block(np.array([qi, qi]))
    if blk is None:                     # the sector of `a` is zero: eigenvalue 0, unit vectors
        k = min(size, k)
        W = np.zeros(k, a.dtype)
        V_flat = np.eye(size, k, dtype=a.dtype)
    else:
        if k >= size - 1:               # ARPACK needs k < n - 1: small blocks are diagonalised completely
            W, V_flat = np.linalg.eig(blk)
            order = np.argsort(-np.abs(W)) if kwargs.get('which', 'LM') == 'LM' else np.arange(len(W))
            W, V_flat = W[order][:k], V_flat[:, order][:, :k]
        else:
            res = scipy.sparse.linalg.eigs(blk, k, *args, **kwargs)
            W, V_flat = res if ret_eigv else (res, None)
    if not ret_eigv:
        return W
    V = []
    for j in range(V_flat.shape[1]):
        vec = npc.Array([leg], dtype=np.promote_types(a.dtype, V_flat.dtype), qtotal=charge_sector)
        vec._set_blocks(np.array([[qi]], dtype=np.intp), arena=dev.to_device(np.ascontiguousarray(V_flat[:, j])),
                        qdata_sorted=True)
        if len(piped_axes) > 0:
            vec = vec.split_legs(0)
        V.append(vec)
    return W, V
