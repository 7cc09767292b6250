"""ctypes binding of the C-ABI in ``include/tenpy_amd.h`` (the drop-in boundary).

There is NO CPU fallback: if the shared library is missing or no MI355X is visible, every compute
entry point raises.  ``torch`` is imported first so that the process-wide HIP runtime
(``libamdhip64.so.7``) is the one torch already loaded; torch is used only for device memory and
streams.
"""
import ctypes
import os

import numpy as np

from . import _build

F64, C128 = 0, 1
E_BADARG, E_NOCONV, E_NAN, E_NOMEM, E_RANKCAP = -1, -2, -3, -4, -5

_lib = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
LANCZOS_CALLBACK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p)
COLLECTIVE_CALLBACK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_void_p)
_SIGS = {
    "tpa_version": (ctypes.c_int, []),
    "tpa_last_error": (ctypes.c_char_p, []),
    "tpa_device_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _i64p]),
    "tpa_gemm_tile_shape": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "tpa_gemm_set_variant": (ctypes.c_int, [ctypes.c_int]),
    "tpa_gemm_chain": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "tpa_axpy": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp]),
    "tpa_scal": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "tpa_dot": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "tpa_nrm2sq": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    "tpa_lanczos_update": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_double, ctypes.c_double, _vp,
                                          ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _vp]),
    "tpa_lanczos_step": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tpa_lanczos_run": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_int, ctypes.c_double, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp]),
    "tpa_lanczos_set_collective": (ctypes.c_int, [COLLECTIVE_CALLBACK, _vp]),
    "tpa_krylov_combine": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tpa_copy_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp]),
    "tpa_lincomb_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _vp, ctypes.c_int64, _vp, _vp, _vp]),
    "tpa_svd_dyn_stats": (ctypes.c_int, [_i64p, ctypes.c_int]),
    "tpa_tri_lower_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int64, _vp, _vp]),
    "tpa_scale_axis_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int64, _vp, _vp, ctypes.c_int, _vp]),
    "tpa_gather_axis_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    "tpa_axis_sqnorm_batch": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "tpa_convert": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64, _vp, _vp, ctypes.c_int, _vp]),
    "tpa_fill_zero": (ctypes.c_int, [_vp, ctypes.c_int64, _vp]),
    "tpa_svd_worksize": (ctypes.c_int64, [ctypes.c_int, _vp, ctypes.c_int]),
    "tpa_svd_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64,
                                     ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int), _vp]),
    "tpa_svd_set_algorithm": (ctypes.c_int, [ctypes.c_int]),
    "tpa_svd_theta": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64,
                                     _vp, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, ctypes.POINTER(ctypes.c_int), _vp, _vp]),
    "tpa_svd_theta_store": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "tpa_svd_set_rank_cap": (ctypes.c_int, [ctypes.c_int]),
    "tpa_svd_call_log": (ctypes.c_int64, [_i64p, ctypes.c_int64, ctypes.c_int]),
    "tpa_qr_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "tpa_qr_set_algorithm": (ctypes.c_int, [ctypes.c_int]),
    "tpa_eigh_worksize": (ctypes.c_int64, [ctypes.c_int, _vp, ctypes.c_int]),
    "tpa_eigh_batch": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int64,
                                      ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int), _vp]),
    "tpa_eigh_set_direct": (ctypes.c_int, [ctypes.c_int]),
    "tpa_eigh_from_svd": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tpa_plan_tensordot": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int64, _i64p, _vp,
                                          ctypes.c_int64, _i64p]),
}


class BackendError(RuntimeError):
    """The HIP extension is missing or a HIP runtime call failed."""


def load(build_if_missing=True):
    """Load ``libtenpy_amd.so`` (building it in-tree if needed) and declare all signatures."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads libamdhip64.so.7 first; device memory + streams come from torch)
    path = _build.LIB_PATH
    alt = os.environ.get('TPA_LIB_PATH')          # dev aid: an explicitly named build variant (e.g. compiled with -DTPA_B32_TIMING)
    if alt:
        if not os.path.exists(alt):
            raise BackendError("tenpy_amd: TPA_LIB_PATH=%s does not exist" % alt)
        path, build_if_missing = alt, False
    if not os.path.exists(path):
        if not build_if_missing:
            raise BackendError("tenpy_amd: %s not built; run `python -m tenpy_amd._build`" % path)
        _build.build()
    elif build_if_missing and os.path.exists(_build.HIPCC) and os.path.isdir(_build.CSRC) and not os.environ.get('TPA_NO_AUTOBUILD'):
        # no-op unless a source under csrc/ is newer than its object: a stale .so is never loaded silently.  One process
        # at a time (N ranks of a torchrun launch all come through here).
        import fcntl
        try:
            with open(os.path.join(_build.OUT_DIR, '.build.lock'), 'w') as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    _build.build()
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        except OSError:
            pass        # read-only install (site-packages, container image): load the library that is there
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        f = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what=""):
    """Map a C-ABI status to the reference's exception types (SURVEY 8(b) 'Errors')."""
    if rc == 0:
        return
    msg = load().tpa_last_error()
    msg = msg.decode() if msg else ""
    if rc == E_BADARG:
        raise ValueError("tenpy_amd %s: %s" % (what, msg))
    if rc == E_NOCONV:
        raise np.linalg.LinAlgError("tenpy_amd %s: %s" % (what, msg))
    if rc == E_NAN:
        raise ValueError("tenpy_amd %s: NaN/Inf encountered: %s" % (what, msg))
    raise BackendError("tenpy_amd %s: HIP error %d: %s" % (what, rc, msg))


_gpu_ok = False


def require_gpu():
    """Raise loudly unless an AMD GPU is usable (no CPU fallback exists).  The positive answer is cached: this sits in front of
    every device call, and `torch.cuda.is_available()` costs microseconds each time (0.07 s per chi=2048 sweep before the cache)."""
    global _gpu_ok
    if _gpu_ok:
        return
    import torch
    if not torch.cuda.is_available():
        raise BackendError("tenpy_amd: no GPU visible (torch.cuda.is_available() is False); "
                           "this backend has no CPU fallback")
    load()
    _gpu_ok = True
