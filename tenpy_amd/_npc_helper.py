"""Drop-in for ``tenpy.linalg._npc_helper`` -- the module TeNPy's own hook ``tenpy.tools.optimization.use_cython``
(tools/optimization.py:262-358) looks native replacements up in.

This is the *fine-grained* form of the boundary ("What a replacement must export"): the reference keeps its
own ``np_conserved.Array`` (a list of numpy blocks) and only the 16 worker functions it decorates with ``@use_cython``
are served from here.  Every floating-point worker uploads its operands, runs the HIP entry point behind
``include/tenpy_amd.h`` and downloads the result blocks -- one PCIe round trip per call, so this form is for
*compatibility* (any TeNPy version, any caller, nothing to re-import) rather than speed; the fast form keeps the tensors
resident and replaces the whole module (``tenpy_amd/install.py``).  The integer workers (charge arithmetic, pipe fusion
maps, strides) are the same host code as in ``tenpy_amd/linalg/charges.py``.

    import tenpy_amd._npc_helper as h; h.register()     # BEFORE ``import tenpy``
    import tenpy                                        # tenpy.show_config(): "compiled without HAVE_MKL"-style line

``register()`` puts this module into ``sys.modules['tenpy.linalg._npc_helper']``.  ``use_cython`` insists that a
replacement carries the *same docstring* as the Python function it replaces (optimization.py:346-357, a guard against
version skew); the docstrings are therefore read from the installed TeNPy's sources with ``ast`` at registration time --
they are the reference's text, not ours, and are not stored in this repository.

Exported names and the reference twin each one replaces (file:line in /root/reference/tenpy/linalg):

=============================  ==========================================  =====================================
name here                      Python twin                                 Cython original (_npc_helper.pyx)
=============================  ==========================================  =====================================
``ChargeInfo_make_valid``      charges.py:266                              :478 (``_make_valid_charges_*`` :443)
``ChargeInfo_check_valid``     charges.py:288                              :512
``LegPipe__init_from_legs``    charges.py:1779                             :545
``_find_row_differences``      charges.py:1922                             :635
``_map_blocks``                charges.py:1945                             :732
``_sliced_copy``               charges.py:1956                             :754 (``_sliced_strided_copy`` :368)
``_make_stride``               charges.py:1997                             :129
``Array_itranspose``           np_conserved.py:2056                        :813
``Array_iadd_prefactor_other`` np_conserved.py:2372                        :860 (``_blas_inpl_add`` :316)
``Array_iscale_prefactor``     np_conserved.py:2385                        :964 (``_blas_inpl_scale`` :339)
``Array__imake_contiguous``    np_conserved.py:2929                        :1000
``_combine_legs_worker``       np_conserved.py:4404                        :1013
``_split_legs_worker``         np_conserved.py:4483                        :1136
``_inner_worker``              np_conserved.py:4614                        :1791
``_tensordot_transpose_axes``  np_conserved.py:4666                        :1260
``_tensordot_worker``          np_conserved.py:4846                        :1498 (``CblasGemmBatch`` :151-273)
=============================  ==========================================  =====================================
plus the attributes ``QTYPE``, ``compiled_with_MKL``, ``_float_complex_are_64_bit`` and the back-references ``_charges`` /
``_np_conserved`` that ``linalg/__init__.py:47-53`` injects.
"""
import ast
import importlib.util
import os
import sys

import numpy as np

from .linalg import _device as dev
from .linalg import charges as mch
from .linalg import np_conserved as mnpc

QTYPE = np.int64                    # _npc_helper.pyx:77
compiled_with_MKL = False           # :75 -- the BLAS of this backend is the MI355X
backend = "tenpy_amd (MI355X, HIP)"
_charges = None                     # set by tenpy/linalg/__init__.py:_patch_cython
_np_conserved = None

__all__ = ['ChargeInfo_make_valid', 'ChargeInfo_check_valid', 'LegPipe__init_from_legs', '_find_row_differences',
           '_map_blocks', '_sliced_copy', '_make_stride', 'Array_itranspose', 'Array_iadd_prefactor_other',
           'Array_iscale_prefactor', 'Array__imake_contiguous', '_combine_legs_worker', '_split_legs_worker', '_inner_worker',
           '_tensordot_transpose_axes', '_tensordot_worker', 'register', 'unregister']


def _float_complex_are_64_bit(dtype_float, dtype_complex):
    """_npc_helper.pyx:427."""
    return np.dtype(dtype_float) == np.float64 and np.dtype(dtype_complex) == np.complex128


# ======================================================================================================================
# reference objects <-> device mirror
# ======================================================================================================================
_leg_memo = {}      # id(reference leg) -> (reference leg kept alive, mirror leg); legs are immutable and shared


def _chinfo(ci):
    key = ('ci', tuple(int(m) for m in ci.mod), tuple(ci.names))
    if key not in _leg_memo:
        _leg_memo[key] = (None, mch.ChargeInfo(list(key[1]), list(key[2])))
    return _leg_memo[key][1]


def _leg(leg):
    hit = _leg_memo.get(id(leg))
    if hit is not None and hit[0] is leg:
        return hit[1]
    ci = _chinfo(leg.chinfo)
    if hasattr(leg, 'q_map') and hasattr(leg, 'legs'):
        res = mch.LegPipe([_leg(l) for l in leg.legs], qconj=int(leg.qconj), sort=bool(leg.sorted), bunch=bool(leg.bunched))
        if not np.array_equal(res.q_map, leg.q_map) or not np.array_equal(res.slices, leg.slices):
            # e.g. a pipe whose outer charges were gauged afterwards: keep its own tables
            res = res.copy()
            res.q_map, res.q_map_slices = np.asarray(leg.q_map), np.asarray(leg.q_map_slices)
            res._set_charges(np.asarray(leg.charges))
            res._set_slices(np.asarray(leg.slices, dtype=np.intp))
            res._perm = None if leg._perm is None else np.asarray(leg._perm)
            res._strides = np.asarray(leg._strides)
    else:
        res = mch.LegCharge.from_qind(ci, np.asarray(leg.slices), np.asarray(leg.charges), int(leg.qconj))
    if len(_leg_memo) > 4096:
        _leg_memo.clear()
    _leg_memo[id(leg)] = (leg, res)
    return res


def _up(a):
    """reference Array (numpy blocks) -> device Array (one arena): one H2D copy."""
    res = mnpc.Array([_leg(l) for l in a.legs], a.dtype, np.asarray(a.qtotal))
    blocks = a._data
    if len(blocks):
        host = np.concatenate([np.ascontiguousarray(b, dtype=res.dtype).reshape(-1) for b in blocks])
        res._set_blocks(np.asarray(a._qdata), arena=dev.to_device(host), qdata_sorted=bool(a._qdata_sorted))
    return res


def _down_into(dst, src):
    """Blocks / qdata of the device Array ``src`` -> the reference Array ``dst`` (one D2H copy)."""
    dst._data = [np.array(b) for b in src._data]
    dst._qdata = np.array(src._qdata, dtype=np.intp, order='C').reshape(-1, src.rank)
    dst._qdata_sorted = bool(src._qdata_sorted)
    return dst


def _calc_dtype(*dtypes):
    return np.dtype(np.complex128) if np.result_type(*dtypes, np.float64).kind == 'c' else np.dtype(np.float64)


# ======================================================================================================================
# integer workers (host)
# ======================================================================================================================

def ChargeInfo_make_valid(self, charges=None):
    if charges is None:
        return np.zeros((self._qnumber,), dtype=QTYPE)
    charges = np.asarray(charges, dtype=QTYPE)
    mask = self._mask
    if np.any(mask):
        charges[..., mask] = np.mod(charges[..., mask], self._mod_masked)
    return charges


def ChargeInfo_check_valid(self, charges):
    charges = np.asarray(charges, dtype=QTYPE)[..., self._mask]
    return bool(np.all((charges >= 0) & (charges < self._mod_masked)))


def LegPipe__init_from_legs(self, sort=True, bunch=True):
    fused = mch.LegPipe([_leg(l) for l in self.legs], qconj=int(self.qconj), sort=sort, bunch=bunch)
    self._perm = fused._perm
    self._strides = fused._strides
    self._set_charges(fused.charges)
    self._set_slices(fused.slices)
    self.sorted, self.bunched = fused.sorted, fused.bunched
    self.q_map, self.q_map_slices = fused.q_map, fused.q_map_slices


def _find_row_differences(qflat):
    return mch._find_row_differences(qflat)


def _map_blocks(blocksizes):
    return mch._map_blocks(blocksizes)


def _sliced_copy(dest, dest_beg, src, src_beg, slice_shape):
    return mch._sliced_copy(dest, dest_beg, src, src_beg, slice_shape)


def _make_stride(shape, cstyle=True):
    return mch._make_stride(shape, cstyle)


# ======================================================================================================================
# floating-point workers (device)
# ======================================================================================================================

def Array_itranspose(self, axes=None):
    if axes is None:
        axes = tuple(reversed(range(self.rank)))
    else:
        axes = tuple(self.get_leg_indices(axes))
        if len(axes) != self.rank or len(set(axes)) != self.rank:
            raise ValueError("axes has wrong length: " + str(axes))
        if axes == tuple(range(self.rank)):
            return self
    moved = _up(self).itranspose(list(axes))
    self.legs = [self.legs[a] for a in axes]
    self._set_shape()
    self._labels = [self._labels[a] for a in axes]
    return _down_into(self, moved)


def Array_iadd_prefactor_other(self, prefactor, other):
    other = other._transpose_same_labels(self._labels)
    if self.rank != other.rank:
        raise ValueError("different rank!")
    for sl, ol in zip(self.legs, other.legs):
        sl.test_equal(ol)
    if np.any(self.qtotal != other.qtotal):
        raise ValueError("Arrays can't have different `qtotal`!")
    summed = _up(self).iadd_prefactor_other(prefactor, _up(other))
    self.dtype = summed.dtype
    return _down_into(self, summed)


def Array_iscale_prefactor(self, prefactor):
    if not np.isscalar(prefactor):
        raise ValueError("prefactor is not scalar: {0!r}".format(type(prefactor)))
    if prefactor == 0.:
        self._data, self._qdata = [], np.empty((0, self.rank), dtype=np.intp)
        self._qdata_sorted = True
        return self
    scaled = _up(self).iscale_prefactor(prefactor)
    self.dtype = scaled.dtype
    return _down_into(self, scaled)


def Array__imake_contiguous(self):
    self._data = [np.ascontiguousarray(b) for b in self._data]
    return self


def _combine_legs_worker(self, res, combine_legs, non_combined_legs, new_axes, non_new_axes, pipes):
    fused = _up(self).combine_legs([list(map(int, cl)) for cl in combine_legs], new_axes=[int(a) for a in new_axes],
                                   pipes=[_leg(p) for p in pipes])
    _down_into(res, fused)
    res._qdata_sorted = True


def _split_legs_worker(self, split_axes, cutoff):
    split = _up(self).split_legs([int(a) for a in split_axes], cutoff)
    legs = list(self.legs)
    for ax in sorted(split_axes, reverse=True):
        legs[ax:ax + 1] = self.legs[ax].legs
    res = _np_conserved.Array(legs, self.dtype, self.qtotal)
    return _down_into(res, split)


def _inner_worker(a, b, do_conj):
    val = mnpc.inner(_up(a), _up(b), axes='range', do_conj=bool(do_conj))
    return _calc_dtype(a.dtype, b.dtype).type(val)


def _listify(x):
    return list(x) if hasattr(x, '__iter__') and not isinstance(x, str) else [x]


def _tensordot_transpose_axes(a, b, axes):
    if a.chinfo != b.chinfo:
        raise ValueError("Different ChargeInfo")
    if isinstance(axes, (int, np.integer)):
        n = int(axes)            # the last `n` legs of a meet the first `n` legs of b: nothing to move
    else:
        axes_a, axes_b = axes
        axes_a, axes_b = a.get_leg_indices(_listify(axes_a)), b.get_leg_indices(_listify(axes_b))
        if len(axes_a) != len(axes_b):
            raise ValueError("different lens of axes for a, b: " + repr(axes))
        a, b = a.copy(deep=False), b.copy(deep=False)       # shallow copies: the callers' Arrays keep their leg order
        a.itranspose([i for i in range(a.rank) if i not in axes_a] + list(axes_a))
        b.itranspose(list(axes_b) + [i for i in range(b.rank) if i not in axes_b])
        n = len(axes_a)
    for la, lb in zip(a.legs[a.rank - n:], b.legs[:n]):
        la.test_contractible(lb)
    return a, b, n


def _tensordot_worker(a, b, axes):
    cut_a = a.rank - axes
    chinfo = a.chinfo
    dtype = _calc_dtype(a.dtype, b.dtype)
    res = _np_conserved.Array(a.legs[:cut_a] + b.legs[axes:], dtype, chinfo.make_valid(a.qtotal + b.qtotal))
    if a.stored_blocks == 0 or b.stored_blocks == 0:
        return res
    prod = mnpc.tensordot(_up(a), _up(b), axes=int(axes))
    return _down_into(res, prod)


# ======================================================================================================================
# registration
# ======================================================================================================================
_TWINS = {'charges.py': ('ChargeInfo_make_valid', 'ChargeInfo_check_valid', 'LegPipe__init_from_legs', '_find_row_differences',
                         '_map_blocks', '_sliced_copy', '_make_stride'),
          'np_conserved.py': ('Array_itranspose', 'Array_iadd_prefactor_other', 'Array_iscale_prefactor', 'Array__imake_contiguous',
                              '_combine_legs_worker', '_split_legs_worker', '_inner_worker', '_tensordot_transpose_axes',
                              '_tensordot_worker')}


def _decorated_docstrings(path):
    """{replacement name: docstring} of every function in the file that carries an ``@use_cython`` decorator."""
    tree = ast.parse(open(path).read())
    out = {}
    for node in ast.walk(tree):
        if not isinstance(node, ast.FunctionDef):
            continue
        for dec in node.decorator_list:
            target = dec.func if isinstance(dec, ast.Call) else dec
            if getattr(target, 'id', getattr(target, 'attr', None)) != 'use_cython':
                continue
            name = node.name
            if isinstance(dec, ast.Call):
                for kw in dec.keywords:
                    if kw.arg == 'replacement':
                        name = kw.value.value
            out[name] = ast.get_docstring(node, clean=False)
    return out


def register(tenpy_path=None):
    """Make this module TeNPy's ``_npc_helper``.  Call before ``import tenpy``.  ``tenpy_path``: directory of the tenpy
    package (default: wherever ``import tenpy`` would find it)."""
    if 'tenpy' in sys.modules:
        raise RuntimeError("tenpy is already imported; register() must run first")
    dev.lib()                       # no GPU / no library -> BackendError here, not deep inside a DMRG run
    if tenpy_path is None:
        spec = importlib.util.find_spec('tenpy')
        if spec is None:
            raise ImportError("tenpy not found on sys.path")
        tenpy_path = os.path.dirname(spec.origin)
    me = sys.modules[__name__]
    for fname, names in _TWINS.items():
        docs = _decorated_docstrings(os.path.join(tenpy_path, 'linalg', fname))
        for name in names:
            if name not in docs:
                raise ImportError("this TeNPy does not decorate %s with use_cython: version not supported" % name)
            getattr(me, name).__doc__ = docs[name]
    sys.modules['tenpy.linalg._npc_helper'] = me
    os.environ.pop('TENPY_NO_CYTHON', None)
    return me


def unregister():
    sys.modules.pop('tenpy.linalg._npc_helper', None)
    for name in [n for n in sys.modules if n == 'tenpy' or n.startswith('tenpy.')]:
        del sys.modules[name]
