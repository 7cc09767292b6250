"""Plug the device path into an UNMODIFIED TeNPy: ``import tenpy; import tenpy_amd.plugin as p; p.install(tenpy)``.

This is the "recommended plugin form" of SURVEY.md 8(b): nothing in the reference tree is edited; after import the
module attributes that the two-site DMRG driver resolves at call time are replaced by device versions,

* ``tenpy.linalg.krylov_based.LanczosGroundState.run`` (constructed at algorithms/dmrg.py:740) -- when the operator
  is a ``TwoSiteH`` with ``combine=True`` (mps_common.py:1226) the environment tensors LP, RP, W0, W1 are uploaded once
  per bond, LHeff / RHeff are fused on the device, and the whole Krylov recurrence (2 GEMM launches per matvec + the
  fused vector update) runs there; only theta goes in and the ground state comes back;
* ``tenpy.linalg.np_conserved.svd`` (np_conserved.py:3676, used by ``truncation.svd_theta`` :258) -- block SVD on the
  device for ``full_matrices=False, compute_uv=True``;
* ``tenpy.linalg.np_conserved.eigh`` (:3899, the density-matrix mixer) and ``qr`` (:4139, ``mode='reduced'``: QR-based
  truncation, canonical forms) -- block decompositions on the device.

Everything else (models, MPS bookkeeping, truncation masks, environment update) stays the reference's own code.  The
conversion functions :func:`to_device` / :func:`to_reference` carry legs (incl. nested pipes), ``_qdata``, ``qtotal``
and labels over unchanged; block data is copied block by block.  The faster way to run is the API mirror
``tenpy_amd.linalg.np_conserved`` (operands stay resident between calls); this module exists so that a
TeNPy user can switch the hot path with two lines and so that the boundary is exercised against the real reference
(``tests/test_plugin_reference.py``).
"""
import numpy as np

from .algorithms.mps_common import TwoSiteH
from .linalg import np_conserved as npc
from .linalg import _device as dev
from .linalg.charges import ChargeInfo, LegCharge, LegPipe
from .linalg.krylov_based import LanczosGroundState

__all__ = ['to_device', 'to_reference', 'install', 'uninstall']

_chinfo_cache = {}
_patched = {}


def _chinfo_to_device(ci):
    key = (tuple(int(m) for m in ci.mod), tuple(ci.names))
    if key not in _chinfo_cache:
        _chinfo_cache[key] = ChargeInfo(list(key[0]), list(key[1]))
    return _chinfo_cache[key]


def leg_to_device(leg, memo=None):
    """Reference ``LegCharge`` / ``LegPipe`` -> ours (same slices, charges, qconj; pipes are rebuilt from their sub-legs
    with the same ``sort`` / ``bunch`` flags, which reproduces ``q_map`` bit for bit -- tests/test_charges.py)."""
    memo = {} if memo is None else memo
    if id(leg) in memo:
        return memo[id(leg)]
    ci = _chinfo_to_device(leg.chinfo)
    if hasattr(leg, 'legs') and hasattr(leg, 'q_map'):
        sub = [leg_to_device(l, memo) for l in leg.legs]
        res = LegPipe(sub, qconj=int(leg.qconj), sort=bool(leg.sorted), bunch=bool(leg.bunched))
        if not np.array_equal(res.q_map, leg.q_map):
            raise ValueError("LegPipe conversion does not reproduce q_map")
    else:
        res = LegCharge.from_qind(ci, np.asarray(leg.slices), np.asarray(leg.charges), int(leg.qconj))
    memo[id(leg)] = res
    return res


def leg_to_reference(leg, tenpy, memo=None):
    memo = {} if memo is None else memo
    if id(leg) in memo:
        return memo[id(leg)]
    rc = tenpy.linalg.charges
    ci = rc.ChargeInfo(list(leg.chinfo.mod), list(leg.chinfo.names))
    if isinstance(leg, LegPipe):
        sub = [leg_to_reference(l, tenpy, memo) for l in leg.legs]
        res = rc.LegPipe(sub, qconj=int(leg.qconj), sort=bool(leg.sorted), bunch=bool(leg.bunched))
    else:
        res = rc.LegCharge.from_qind(ci, np.asarray(leg.slices), np.asarray(leg.charges), int(leg.qconj))
    memo[id(leg)] = res
    return res


def to_device(a, legs=None):
    """Reference ``Array`` (list of numpy blocks) -> device ``Array`` (one arena).  ``legs``: already converted legs to
    share (e.g. the pipes of an effective Hamiltonian)."""
    if legs is None:
        memo = {}
        legs = [leg_to_device(l, memo) for l in a.legs]
    res = npc.Array(legs, a.dtype, np.asarray(a.qtotal), list(a._labels))
    blocks = a._data
    if len(blocks):
        host = np.concatenate([np.ascontiguousarray(b, dtype=res.dtype).reshape(-1) for b in blocks])
    else:
        host = np.zeros(0, dtype=res.dtype)
    res._set_blocks(np.asarray(a._qdata), arena=dev.to_device(host), qdata_sorted=bool(a._qdata_sorted))
    return res


def to_reference(a, tenpy, legs=None):
    """Device ``Array`` -> reference ``Array``; ``legs``: reference legs to reuse (they must describe the same blocks)."""
    rnpc = tenpy.linalg.np_conserved
    if legs is None:
        memo = {}
        legs = [leg_to_reference(l, tenpy, memo) for l in a.legs]
    res = rnpc.Array(legs, a.dtype, np.asarray(a.qtotal), list(a._labels))
    res._data = [np.array(b) for b in a._data]
    res._qdata = np.array(a._qdata, dtype=np.intp, order='C').reshape(-1, a.rank)
    res._qdata_sorted = bool(a._qdata_sorted)
    return res


# ---------------------------------------------------------------------------------------------------------------------
def _device_two_site_h(H):
    """Device twin of a reference ``TwoSiteH`` (cached on the object for the lifetime of the bond update)."""
    twin = getattr(H, '_tpa_twin', None)
    if twin is None:
        LP, RP, W0, W1 = (to_device(x) for x in (H.LP, H.RP, H.W0, H.W1))
        W0 = W0.replace_labels(['p0', 'p0*'], ['p', 'p*']) if 'p0' in W0.get_leg_labels() else W0
        W1 = W1.replace_labels(['p1', 'p1*'], ['p', 'p*']) if 'p1' in W1.get_leg_labels() else W1
        twin = TwoSiteH(None, H.i0, tensors=(LP, RP, W0, W1))
        H._tpa_twin = twin
    return twin


def _lanczos_run(orig_run, tenpy):
    ref_TwoSiteH = tenpy.algorithms.mps_common.TwoSiteH

    def run(self):
        H = self.H
        if not (isinstance(H, ref_TwoSiteH) and getattr(H, 'combine', False) and hasattr(H, 'LHeff')):
            return orig_run(self)          # shifted / projected / one-site operators: the reference's own loop
        twin = _device_two_site_h(H)
        psi0 = self.psi0
        theta = to_device(psi0, legs=[twin.pipeL, twin.pipeR])
        theta.iset_leg_labels(['(vL.p0)', '(p1.vR)'])
        opts = dict(N_min=self.N_min, N_max=self.N_max, P_tol=self.P_tol, min_gap=self.min_gap, reortho=self.reortho,
                    E_tol=getattr(self, 'E_tol', np.inf), cutoff=self._cutoff)
        E0, th, N = LanczosGroundState(twin, theta, opts).run()
        res = to_reference(th, tenpy, legs=list(psi0.legs))
        res.iset_leg_labels(list(psi0.get_leg_labels()))
        return E0, res, N
    return run


def _svd(orig_svd, tenpy):
    def svd(a, full_matrices=False, compute_uv=True, cutoff=None, qtotal_LR=[None, None], inner_labels=[None, None],
            inner_qconj=+1):
        if full_matrices or not compute_uv or a.rank != 2:
            return orig_svd(a, full_matrices, compute_uv, cutoff, qtotal_LR, inner_labels, inner_qconj)
        U, S, VH = npc.svd(to_device(a), full_matrices=False, compute_uv=True, cutoff=cutoff, qtotal_LR=list(qtotal_LR),
                           inner_labels=list(inner_labels), inner_qconj=inner_qconj)
        memo = {}
        Ur = to_reference(U, tenpy, legs=[a.legs[0], leg_to_reference(U.legs[1], tenpy, memo)])
        VHr = to_reference(VH, tenpy, legs=[Ur.legs[1].conj(), a.legs[1]])
        return Ur, np.asarray(S), VHr
    return svd


def _eigh(orig_eigh, tenpy):
    def eigh(a, UPLO='L', sort=None):
        if a.rank != 2:
            return orig_eigh(a, UPLO, sort)
        W, V = npc.eigh(to_device(a), UPLO=UPLO, sort=sort)
        return np.asarray(W), to_reference(V, tenpy, legs=[a.legs[0], a.legs[0].conj()])
    return eigh


def _qr(orig_qr, tenpy):
    def qr(a, mode='reduced', inner_labels=[None, None], cutoff=None, pos_diag_R=False, qtotal_Q=None, inner_qconj=+1):
        if mode != 'reduced' or cutoff is not None or a.rank != 2:
            return orig_qr(a, mode, inner_labels, cutoff, pos_diag_R, qtotal_Q, inner_qconj)
        Q, R = npc.qr(to_device(a), mode='reduced', inner_labels=list(inner_labels), pos_diag_R=pos_diag_R, qtotal_Q=qtotal_Q,
                      inner_qconj=inner_qconj)
        memo = {}
        Qr = to_reference(Q, tenpy, legs=[a.legs[0], leg_to_reference(Q.legs[1], tenpy, memo)])
        Rr = to_reference(R, tenpy, legs=[Qr.legs[1].conj(), a.legs[1]])
        return Qr, Rr
    return qr


def install(tenpy):
    """Patch the imported TeNPy package object ``tenpy``.  Raises ``BackendError`` right away without a GPU."""
    dev.lib()                                  # fail loudly if the HIP library / device is missing
    import importlib
    kb = importlib.import_module(tenpy.__name__ + '.linalg.krylov_based')
    rnpc = importlib.import_module(tenpy.__name__ + '.linalg.np_conserved')
    importlib.import_module(tenpy.__name__ + '.algorithms.mps_common')
    if 'lanczos' in _patched:
        return
    _patched['lanczos'] = (kb.LanczosGroundState, kb.LanczosGroundState.run)
    kb.LanczosGroundState.run = _lanczos_run(kb.LanczosGroundState.run, tenpy)
    _patched['svd'] = (rnpc, rnpc.svd)
    rnpc.svd = _svd(rnpc.svd, tenpy)
    _patched['eigh'] = (rnpc, rnpc.eigh)           # density-matrix mixer (mps_common.py:2047/2055)
    rnpc.eigh = _eigh(rnpc.eigh, tenpy)
    _patched['qr'] = (rnpc, rnpc.qr)               # QR-based truncation / canonical forms (truncation.py:533, mps.py:4556)
    rnpc.qr = _qr(rnpc.qr, tenpy)


def uninstall():
    if 'lanczos' in _patched:
        cls, run = _patched.pop('lanczos')
        cls.run = run
    for name in ('svd', 'eigh', 'qr'):
        if name in _patched:
            mod, fn = _patched.pop(name)
            setattr(mod, name, fn)
