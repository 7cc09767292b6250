"""Device-resident block-sparse tensors with abelian charge conservation.

Mirrors the interface of ``tenpy/linalg/np_conserved.py`` (reference ``Array`` :154, ``tensordot`` :3612,
``inner`` :3540, ``svd`` :3676, ``qr`` :4139, ``eigh`` :3899, ``combine_legs`` :1561, ``split_legs`` :1707)
for the hot path of DMRG/TEBD, but with an MI355X-first data layout:

* all blocks of an :class:`Array` are packed back to back in ONE arena in HBM (``_arena``); the host keeps
  only the integer bookkeeping: ``_qdata`` (qindices of the stored blocks, exactly the reference's field),
  ``_offsets`` (element offset of each block in the arena) and the shared ``legs``;
* a contraction is planned once on the host (integer work, C++ ``tpa_plan_tensordot``) into device-side
  task/link/tile tables that are cached and replayed; all flops run in the grouped chained MFMA GEMM;
* BLAS-1 work on two Arrays with equal block structure is a single flat pass over the arenas;
* SVD / QR / eigh of all charge blocks run as one batched device call.

Floating point data never goes through numpy on the product path; there is no CPU fallback.
"""
import functools
import threading
import warnings
from collections import OrderedDict

import os

import numpy as np
from hashlib import blake2b as _blake2b

from . import _device as dev
from .charges import QTYPE, ChargeInfo, DipolarChargeInfo, LegCharge, LegPipe, _find_row_differences, _partial_qtotal

__all__ = ['QCUTOFF', 'ChargeInfo', 'DipolarChargeInfo', 'LegCharge', 'LegPipe', 'Array', 'zeros', 'ones', 'eye_like', 'diag',
           'concatenate', 'grid_concat', 'grid_outer', 'detect_grid_outer_legcharge', 'detect_qtotal', 'detect_legcharge',
           'trace', 'outer', 'inner', 'tensordot', 'svd', 'pinv', 'norm', 'eigh', 'eig', 'eigvalsh', 'eigvals', 'speigs',
           'expm', 'qr', 'lq', 'polar', 'orthogonal_columns', 'to_iterable_arrays', 'TensordotPlan']

QCUTOFF = np.finfo(np.float64).eps * 10

COPY_MAXDIM = 6


@functools.lru_cache(maxsize=256)
def _calc_dtype(*dtypes):
    """float64 or complex128, like ``_find_calc_dtype`` (np_conserved.py:4396).  (Memoised: called ~50 times per bond.)"""
    res = np.result_type(*dtypes, np.float64)
    return np.dtype(np.complex128) if res.kind == 'c' else np.dtype(np.float64)


def _is_iterable(x):
    return not isinstance(x, str) and hasattr(x, '__iter__')


def _to_iterable(x):
    return list(x) if _is_iterable(x) else [x]


class _HostBlockList(list):
    """What ``Array._data`` returns: a list of host blocks that remembers its Array, so that *mutating* the list
    (``append``, item assignment, ...) registers the blocks for upload; reading it costs nothing extra."""

    def __init__(self, owner, blocks, dirty=False):
        list.__init__(self, blocks)
        self._owner = owner
        if dirty:
            owner.__dict__['_host_blocks'] = self

    def _touch(self):
        self._owner.__dict__['_host_blocks'] = self
        self._owner._skey = None

    def append(self, blk):
        list.append(self, blk)
        self._touch()

    def extend(self, blks):
        list.extend(self, blks)
        self._touch()

    def insert(self, i, blk):
        list.insert(self, i, blk)
        self._touch()

    def __setitem__(self, i, blk):
        list.__setitem__(self, i, blk)
        self._touch()

    def __reduce__(self):
        return (list, (list(self),))


class Array:
    """Block-sparse tensor with charge conservation; blocks live in a single HBM arena.

    Same public attributes as the reference ``Array`` (np_conserved.py:154-205): ``rank``, ``shape``,
    ``dtype``, ``chinfo``, ``qtotal``, ``legs``, ``_qdata`` (intp, ``stored_blocks x rank``),
    ``_qdata_sorted``, ``_labels``.  Instead of ``_data`` (list of numpy blocks) there is ``_arena``
    (1-D device tensor) and ``_offsets`` (int64, start of each block).  ``_data`` is available as a
    read-only property that copies the blocks to the host (debugging / tests only).
    """

    def __init__(self, legcharges, dtype=np.float64, qtotal=None, labels=None):
        self.legs = list(legcharges)
        if len(self.legs) == 0:
            raise ValueError("can't have 0-rank tensors")
        self._set_shape()
        self.dtype = _calc_dtype(dtype)
        self.chinfo = self.legs[0].chinfo
        self.qtotal = self.chinfo.make_valid(qtotal)
        self._labels = [None] * self.rank
        if labels is not None:
            self.iset_leg_labels(labels)
        self._qdata = np.empty((0, self.rank), dtype=np.intp)
        self._offsets = np.zeros(0, dtype=np.int64)
        self._arena = None
        self._qdata_sorted = True
        self._skey = None

    # ---- internal: structure -----------------------------------------------------------------------
    def _set_shape(self):
        self.shape = tuple(leg.ind_len for leg in self.legs)
        self.rank = len(self.legs)

    def _block_shapes(self, qdata=None):
        """(stored_blocks, rank) array of block shapes."""
        if qdata is None:
            qdata = self._qdata
        res = np.empty(qdata.shape, dtype=np.int64)
        for a, leg in enumerate(self.legs):
            res[:, a] = leg.get_block_sizes()[qdata[:, a]]
        return res

    def _set_blocks(self, qdata, arena=None, zero=False, qdata_sorted=False):
        """Install a block list; offsets are the packed layout in row order of ``qdata``."""
        qdata = np.ascontiguousarray(qdata, dtype=np.intp).reshape(-1, self.rank)
        shapes = self._block_shapes(qdata)
        sizes = np.prod(shapes, axis=1) if len(shapes) else np.zeros(0, np.int64)
        offs = np.zeros(len(sizes) + 1, dtype=np.int64)
        np.cumsum(sizes, out=offs[1:])
        self._qdata = qdata
        self._offsets = offs[:-1].copy()
        total = int(offs[-1])
        if arena is None:
            arena = dev.zeros(total, self.dtype) if zero else dev.empty(total, self.dtype)
        self._arena = arena
        self._qdata_sorted = qdata_sorted
        self._skey = None
        return sizes

    def _adopt_blocks(self, qdata, offsets, arena, qdata_sorted):
        """Install a planned block list (``qdata`` / ``offsets`` are shared, read-only arrays of a reshaping plan)."""
        self._qdata = qdata
        self._offsets = offsets
        self._arena = arena
        self._qdata_sorted = qdata_sorted
        self._skey = None

    def _block_sizes_flat(self):
        """Number of elements of every stored block.  Memoised on the identity of ``_qdata`` and of the legs (both are
        replaced, never modified in place): the Lanczos vector kernels ask for it ~60 times per bond."""
        c = self.__dict__.get('_sz_cache')
        if c is not None and c[0] is self._qdata and len(c[1]) == len(self.legs) and all(x is y for x, y in zip(c[1], self.legs)):
            return c[2]
        shapes = self._block_shapes()
        sizes = np.prod(shapes, axis=1) if len(shapes) else np.zeros(0, np.int64)
        self._sz_cache = (self._qdata, tuple(self.legs), sizes)
        return sizes

    def _is_packed(self):
        """True if blocks are packed back to back in qdata order (then the arena is a flat vector).  Memoised like
        ``_block_sizes_flat`` (plus the identity of ``_offsets`` and the arena size)."""
        sizes = self._block_sizes_flat()
        if len(sizes) == 0:
            return True
        numel = self._arena.numel()
        c = self.__dict__.get('_pk_cache')
        if c is not None and c[0] is sizes and c[1] is self._offsets and c[2] == numel:
            return c[3]
        exp = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        res = bool(np.array_equal(exp, self._offsets) and numel == int(np.sum(sizes)))
        self._pk_cache = (sizes, self._offsets, numel, res)
        return res

    def _struct_key(self):
        """Hashable key of everything a contraction plan depends on (blocks, offsets, leg block sizes)."""
        if self._skey is None:
            # a 128-bit digest of the bytes, not Python's 64-bit hash(): cached plans are replayed on a key hit without a second look
            # (ADVICE r3), so the identity has to be collision-free for all practical purposes
            h = _blake2b(digest_size=16)
            h.update(np.array([self.rank, len(self._qdata)], dtype=np.int64).tobytes())
            h.update(np.ascontiguousarray(self._qdata).tobytes())
            h.update(np.ascontiguousarray(self._offsets).tobytes())
            for leg in self.legs:
                h.update(np.array([leg.ind_len, leg.block_number], dtype=np.int64).tobytes())
                h.update(np.ascontiguousarray(leg.slices).tobytes())
            self._skey = h.digest()
        return self._skey

    def _same_structure(self, other):
        return (self._qdata.shape == other._qdata.shape and np.array_equal(self._qdata, other._qdata)
                and np.array_equal(self._offsets, other._offsets))

    # ---- block storage: one arena in HBM; ``_data`` is the reference's host-side view of it ---------------------------
    @property
    def _arena(self):
        """1-D device tensor holding all stored blocks.  If host blocks were assigned through ``_data`` (see below) they
        are uploaded here, on first use."""
        if self.__dict__.get('_host_blocks') is not None:
            self._upload_host_blocks()
        return self.__dict__.get('_arena_t')

    @_arena.setter
    def _arena(self, tensor):
        self.__dict__['_arena_t'] = tensor
        self.__dict__['_host_blocks'] = None
        self._drop_cow()

    def _drop_cow(self):
        token = self.__dict__.get('_cow')
        if token is not None:
            token[0] -= 1
            self.__dict__['_cow'] = None

    def _own_arena(self):
        """Copy-on-write.  The reference's in-place methods (``iscale_axis``, ``iconj``, ``iscale_prefactor``, ...) REBIND the
        block list (np_conserved.py:2130, :2213, :2398), so a shallow copy (``copy(deep=False)``, ``replace_label``,
        ``gauge_total_charge``, ...) never sees them; here they write into the arena, hence an arena that is shared with a
        shallow copy is cloned before the first write."""
        token = self.__dict__.get('_cow')
        if token is not None:
            t = self._arena
            shared = token[0] > 1           # ``_cow`` = [number of Arrays that were given this arena]; collected ones still count
            self._drop_cow()
            if shared and t is not None:
                self.__dict__['_arena_t'] = dev.clone(t)

    @property
    def _offsets(self):
        """int64 start of every stored block inside the arena (row order of ``_qdata``)."""
        if self.__dict__.get('_host_blocks') is not None:
            self._upload_host_blocks()
        return self.__dict__['_offsets_v']

    @_offsets.setter
    def _offsets(self, offs):
        self.__dict__['_offsets_v'] = offs

    @property
    def _data(self):
        """The reference's ``_data``: list of the stored blocks as numpy arrays, in the order of ``_qdata`` -- here host
        COPIES of the arena (tests, ``__iter__``, HDF5, and the few reference callers that read blocks directly:
        ``linalg/sparse.py:549-560``, ``linalg/truncation.py:456``).  Assigning a list of host arrays (``a._data = [...]``,
        ``linalg/sparse.py:511``) or appending to the list of an empty Array (``sparse.py:523``) is honoured too: the
        blocks are uploaded when the arena is next needed, laid out in the order of the ``_qdata`` valid at that time."""
        pending = self.__dict__.get('_host_blocks')
        if pending is not None:
            return pending
        if '_data_placeholder' in self.__dict__:        # tools/cache.py:541 parks the block count here while on disk
            return self.__dict__['_data_placeholder']
        if self.stored_blocks == 0 or self.__dict__.get('_arena_t') is None:
            return _HostBlockList(self, [])
        host = dev.to_host(self._arena)
        shapes = self._block_shapes()
        sizes = np.prod(shapes, axis=1)
        # write-through views (`_npc_cold.HostBlock`): the reference's tests and a few callers modify a block in place through
        # `_data` (``b._data[-1][0, -1] += 1e-13``, test_np_conserved.py:507); with numpy blocks that IS the tensor
        from ._npc_cold import HostBlock
        return _HostBlockList(self, [HostBlock(host[o:o + s].reshape(tuple(sh)), self, o) for o, s, sh in zip(self._offsets, sizes, shapes)])

    @_data.setter
    def _data(self, blocks):
        self.__dict__.pop('_data_placeholder', None)
        if not _is_iterable(blocks):                    # ``value._data = len(data)`` (tools/cache.py:541): the blocks are gone
            self._arena = None
            self.__dict__['_data_placeholder'] = blocks
            return
        self.__dict__['_host_blocks'] = _HostBlockList(self, list(blocks), dirty=True)
        self._skey = None

    def _upload_host_blocks(self):
        blocks = list(self.__dict__['_host_blocks'])
        self.__dict__['_host_blocks'] = None
        if len(blocks) != self._qdata.shape[0]:
            raise ValueError("len(_data) = %d does not match the %d rows of _qdata" % (len(blocks), self._qdata.shape[0]))
        if any(np.asarray(b).dtype.kind == 'c' for b in blocks) and self.dtype.kind != 'c':
            self.dtype = np.dtype(np.complex128)
        sizes = np.array([np.asarray(b).size for b in blocks], dtype=np.int64)
        self._offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(blocks) else np.zeros(0, np.int64)
        flat = np.concatenate([np.ascontiguousarray(b, dtype=self.dtype).reshape(-1) for b in blocks]) if len(blocks) \
            else np.zeros(0, self.dtype)
        self.__dict__['_arena_t'] = dev.to_device(flat)
        self._drop_cow()
        self._skey = None
        self.__dict__.pop('_sz_cache', None)
        self.__dict__.pop('_pk_cache', None)

    # ---- sanity ------------------------------------------------------------------------------------
    def test_sanity(self):
        """Check the invariants of SURVEY Appendix A (reference test_sanity :223-270)."""
        if len(self.legs) == 0 or self.rank != len(self.legs):
            raise ValueError("wrong rank")
        for leg in self.legs:
            if leg.chinfo != self.chinfo:
                raise ValueError("leg has different ChargeInfo")
            leg.test_sanity()
        if self._qdata.shape != (self.stored_blocks, self.rank) or self._qdata.dtype != np.intp:
            raise ValueError("wrong _qdata")
        if not self._qdata.flags['C_CONTIGUOUS']:
            raise ValueError("_qdata not contiguous")
        if self.stored_blocks:
            if np.any(self._qdata < 0) or np.any(self._qdata >= [l.block_number for l in self.legs]):
                raise ValueError("invalid qind in _qdata")
            if not np.array_equal(_partial_qtotal(self.chinfo, self.legs, self._qdata), np.tile(self.qtotal, (self.stored_blocks, 1))):
                raise ValueError("some row of _qdata is incompatible with total charge")
            sizes = self._block_sizes_flat()
            if self._arena is None or np.any(self._offsets + sizes > self._arena.numel()):
                raise ValueError("block outside of arena")
            if self._arena.dtype != dev.tdtype(self.dtype):
                raise ValueError("arena dtype mismatch")
        if self._qdata_sorted and self.stored_blocks > 1:
            perm = np.lexsort(self._qdata.T)
            if np.any(perm != np.arange(len(perm))):
                raise ValueError("_qdata_sorted == True, but _qdata is not sorted")

    # ---- copies ------------------------------------------------------------------------------------
    def copy(self, deep=True):
        res = Array.__new__(Array)
        self._arena                 # (uploads blocks assigned through `_data`, if any, before the fields are shared)
        res.__dict__.update(self.__dict__)
        res.__dict__['_cow'] = None
        res.legs = list(self.legs)
        res._labels = list(self._labels)
        if deep:
            res._qdata = self._qdata.copy()
            res._offsets = self._offsets.copy()
            res.qtotal = self.qtotal.copy()
            res._arena = None if self._arena is None else dev.clone(self._arena)
        else:
            token = self.__dict__.get('_cow')                           # see _own_arena
            if token is None:
                token = self.__dict__['_cow'] = [1]
            token[0] += 1
            res.__dict__['_cow'] = token
        return res

    def zeros_like(self):
        return Array(self.legs, self.dtype, self.qtotal, self._labels)

    # ---- constructors --------------------------------------------------------------------------------
    @classmethod
    def from_ndarray_trivial(cls, data_flat, dtype=None, labels=None):
        data_flat = np.asarray(data_flat)
        chinfo = ChargeInfo()
        legs = [LegCharge.from_trivial(s, chinfo) for s in data_flat.shape]
        if dtype is None:
            dtype = data_flat.dtype
        res = cls(legs, dtype, labels=labels)            # one block, kept also if it is zero (reference :443-446)
        res._qdata = np.zeros((1, res.rank), np.intp)
        res._data = [data_flat.astype(res.dtype, copy=False)]
        res._qdata_sorted = True
        return res

    @classmethod
    def from_ndarray(cls, data_flat, legcharges, dtype=None, qtotal=None, cutoff=None, labels=None,
                     raise_wrong_sector=True, warn_wrong_sector=True):
        """Upload a dense host array, keeping only the charge-allowed blocks that are non-zero."""
        if cutoff is None:
            cutoff = QCUTOFF
        data_flat = np.asarray(data_flat)
        if dtype is None:
            dtype = data_flat.dtype
        dtype = _calc_dtype(dtype)
        res = cls(legcharges, dtype, qtotal, labels)
        if res.shape != data_flat.shape:
            raise ValueError("Incompatible shapes: legcharges {0!s} vs flat {1!s} ".format(res.shape, data_flat.shape))
        data_flat = data_flat.astype(dtype, copy=False)
        if qtotal is None:
            res.qtotal = qtotal = detect_qtotal(data_flat, res.legs, cutoff)
        qdata = res._allowed_qdata()
        keep, blocks = [], []
        for row in qdata:
            sl = tuple(leg.get_slice(q) for leg, q in zip(res.legs, row))
            blk = data_flat[sl]
            if blk.size and np.any(np.abs(blk) > cutoff):
                keep.append(row)
                blocks.append(np.ascontiguousarray(blk).reshape(-1))
        if warn_wrong_sector or raise_wrong_sector:
            total = np.sum(np.abs(data_flat) > cutoff)
            kept = sum(int(np.sum(np.abs(b) > cutoff)) for b in blocks)
            if total != kept:
                msg = "flat array has non-zero entries in blocks incompatible with charge"
                if raise_wrong_sector:
                    raise ValueError(msg)
                warnings.warn(msg, stacklevel=2)
        if keep:
            res._set_blocks(np.array(keep, np.intp), arena=dev.to_device(np.concatenate(blocks)), qdata_sorted=True)
        return res

    def _allowed_qdata(self):
        """All qindex tuples compatible with ``qtotal``, lexsorted (last leg most significant)."""
        nb = [leg.block_number for leg in self.legs]
        n_tot = int(np.prod(nb))
        if n_tot == 0:
            return np.empty((0, self.rank), np.intp)
        # column `a` varies fastest for a = 0 -> rows come out lexsorted with the last leg most significant
        grid = np.empty((n_tot, self.rank), dtype=np.intp)
        rep = 1
        for a in range(self.rank):
            grid[:, a] = (np.arange(n_tot) // rep) % nb[a]
            rep *= nb[a]
        ch = _partial_qtotal(self.chinfo, self.legs, grid)
        ok = np.all(ch == self.qtotal[np.newaxis, :], axis=1) if self.chinfo.qnumber else np.ones(n_tot, bool)
        return grid[ok]

    @classmethod
    def from_func(cls, func, legcharges, dtype=None, qtotal=None, func_args=(), func_kwargs={}, shape_kw=None,
                  labels=None):
        """Fill every charge-allowed block with ``func(shape, ...)`` evaluated on the host, then upload."""
        res = cls(legcharges, np.float64 if dtype is None else dtype, qtotal, labels)
        qdata = res._allowed_qdata()
        shapes = res._block_shapes(qdata)
        blocks = []
        for sh in shapes:
            sh = tuple(int(s) for s in sh)
            blk = func(sh, *func_args, **func_kwargs) if shape_kw is None else \
                func(*func_args, **{**func_kwargs, shape_kw: sh})
            blocks.append(np.asarray(blk).reshape(-1))
        if dtype is None and len(blocks):            # the dtype is what `func` returns (reference :528-604)
            res.dtype = _calc_dtype(*[b.dtype for b in blocks])
        if len(blocks):
            res._set_blocks(qdata, arena=dev.to_device(np.concatenate(blocks).astype(res.dtype, copy=False)), qdata_sorted=True)
        return res

    @classmethod
    def from_func_square(cls, func, leg, dtype=None, func_args=(), func_kwargs={}, shape_kw=None, labels=None):
        blocked = leg.is_blocked()
        if not blocked:
            pipe = LegPipe([leg])
            leg_use = pipe
        else:
            leg_use = leg
        res = cls.from_func(func, [leg_use, leg_use.conj()], dtype, None, func_args, func_kwargs, shape_kw, labels)
        if not blocked:
            res = res.split_legs()
        return res

    # ---- properties ------------------------------------------------------------------------------------
    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def stored_blocks(self):
        return self._qdata.shape[0]

    @property
    def ndim(self):
        return self.rank

    # ---- labels --------------------------------------------------------------------------------------
    def get_leg_index(self, label):
        """Leg index for a leg index (negative counts from the end) or a label; anything that is not an integer is looked
        up among the labels, so ``None`` finds the first unlabeled leg (reference :683)."""
        if isinstance(label, (int, np.integer)) and not isinstance(label, bool):
            idx = int(label)
            if idx < 0:
                idx += self.rank
            if idx < 0 or idx >= self.rank:
                raise ValueError("axis {0:d} out of rank {1:d}".format(int(label), self.rank))
            return idx
        try:
            return self._labels.index(label)
        except ValueError:
            raise KeyError("label not found: " + repr(label) + ", current labels" + repr(self._labels)) from None

    def get_leg_indices(self, labels):
        return [self.get_leg_index(l) for l in labels]

    def iset_leg_labels(self, labels):
        labels = list(labels)
        if len(labels) != self.rank:
            raise ValueError("Need one leg label for each of the legs.")
        for i, l in enumerate(labels):
            if l == '':
                raise ValueError("use `None` for empty labels")
            if l is not None and l in labels[i + 1:]:
                raise ValueError("Duplicate label entry in {0!r}".format(labels))
        self._labels = labels
        return self

    def get_leg_labels(self):
        return list(self._labels)

    def has_label(self, label):
        return label in self._labels

    def get_leg(self, label):
        return self.legs[self.get_leg_index(label)]

    def ireplace_label(self, old_label, new_label):
        idx = self.get_leg_index(old_label)
        labels = list(self._labels)
        labels[idx] = None
        if new_label in labels and new_label is not None:
            raise ValueError("Duplicate label: trying to set {0!r} in {1!r}".format(new_label, labels))
        labels[idx] = new_label
        self._labels = labels
        return self

    def replace_label(self, old_label, new_label):
        return self.copy(deep=False).ireplace_label(old_label, new_label)

    def ireplace_labels(self, old_labels, new_labels):
        idx = self.get_leg_indices(old_labels)
        labels = list(self._labels)
        for i in idx:
            labels[i] = None
        for i, new in zip(idx, new_labels):
            if new is not None and new in labels:
                raise ValueError("Duplicate label: trying to set {0!r} in {1!r}".format(new, labels))
            labels[i] = new
        self._labels = labels
        return self

    def replace_labels(self, old_labels, new_labels):
        return self.copy(deep=False).ireplace_labels(old_labels, new_labels)

    def idrop_labels(self, old_labels=None):
        if old_labels is None:
            self._labels = [None] * self.rank
        else:
            for i in self.get_leg_indices(old_labels):
                self._labels[i] = None
        return self

    def __repr__(self):
        return "<npc.Array(device) shape={0!s} labels={1!r} blocks={2:d}>".format(self.shape, self._labels, self.stored_blocks)

    def sparse_stats(self):
        sizes = self._block_sizes_flat()
        nblocks = self.stored_blocks
        stored = int(np.sum(sizes))
        return "{0:d} of {1:d} entries (={2:g}) stored in {3:d} blocks".format(stored, self.size, stored / max(self.size, 1), nblocks)

    # ---- host transfer -----------------------------------------------------------------------------------
    def to_ndarray(self):
        """Dense host copy (D2H of the arena + scatter)."""
        res = np.zeros(self.shape, dtype=self.dtype)
        if self.stored_blocks == 0:
            return res
        host = dev.to_host(self._arena)
        shapes = self._block_shapes()
        sizes = np.prod(shapes, axis=1)
        for row, o, s, sh in zip(self._qdata, self._offsets, sizes, shapes):
            sl = tuple(leg.get_slice(q) for leg, q in zip(self.legs, row))
            res[sl] = host[o:o + s].reshape(tuple(sh))
        return res

    def get_block(self, qindices):
        """Host copy of one block (``None`` if not stored)."""
        qindices = np.asarray(qindices, dtype=np.intp)
        hit = np.nonzero(np.all(self._qdata == qindices[np.newaxis, :], axis=1))[0]
        if len(hit) == 0:
            return None
        i = int(hit[0])
        sh = tuple(int(s) for s in self._block_shapes()[i])
        n = int(np.prod(sh))
        return dev.to_host(self._arena[int(self._offsets[i]):int(self._offsets[i]) + n]).reshape(sh)

    def __getstate__(self):
        arena = self._arena
        state = dict(self.__dict__)
        state.pop('_arena_t', None)
        state['_offsets'] = state.pop('_offsets_v')
        state.pop('_host_blocks', None)
        state.pop('_cow', None)
        state['_arena'] = None if arena is None else dev.to_host(arena)
        state['_skey'] = None
        state.pop('_sz_cache', None)
        state.pop('_pk_cache', None)
        return state

    def __setstate__(self, state):
        arena = state.pop('_arena')
        offs = state.pop('_offsets')
        self.__dict__.update(state)
        self._offsets = offs
        self._arena = None if arena is None else dev.to_device(arena)

    # ---- leg structure ---------------------------------------------------------------------------------------
    def is_completely_blocked(self):
        return all(leg.is_blocked() for leg in self.legs)

    def make_pipe(self, axes, **kwargs):
        axes = self.get_leg_indices(axes)
        return LegPipe([self.legs[a] for a in axes], **kwargs)

    def isort_qdata(self):
        """Bring ``_qdata`` into lexsorted order (bookkeeping only: offsets are permuted, no data moves)."""
        if self._qdata_sorted:
            return
        if self.stored_blocks < 2:
            self._qdata_sorted = True
            return
        perm = np.lexsort(self._qdata.T)
        self._qdata = np.ascontiguousarray(self._qdata[perm])
        self._offsets = self._offsets[perm]
        self._qdata_sorted = True
        self._skey = None

    def _repack(self):
        """Physically reorder the arena so that blocks are packed in ``_qdata`` order."""
        if self.stored_blocks == 0 or self._is_packed():
            return self
        sizes = self._block_sizes_flat()
        new_offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        jobs = _copy_jobs_contiguous(new_offs, self._offsets, sizes)
        new_arena = dev.empty(int(np.sum(sizes)), self.dtype)
        _run_copy(self.dtype, jobs, int(np.max(sizes)), self._arena, new_arena)
        self._arena = new_arena
        self._offsets = new_offs
        self._skey = None
        return self

    def as_completely_blocked(self):
        """Return ``(piped_axes, blocked_self)``: non-blocked legs are wrapped into 1-leg pipes."""
        piped = [a for a, leg in enumerate(self.legs) if not leg.is_blocked()]
        if len(piped) == 0:
            return piped, self
        res = self.combine_legs([[a] for a in piped], new_axes=piped)
        return piped, res

    def sort_legcharge(self, sort=True, bunch=True):
        """Sort/bunch legs; returns ``(perms, result)`` where perms[a] is the flat index permutation of leg a."""
        if sort is False or sort is True:
            sort = [sort] * self.rank
        if bunch is False or bunch is True:
            bunch = [bunch] * self.rank
        perms = [np.arange(n, dtype=np.intp) for n in self.shape]       # identity where nothing moves (reference :1405)
        res = self.copy(deep=False)
        res._qdata = self._qdata.copy()
        bunch_axes = []
        for a in range(self.rank):
            leg = self.legs[a]
            if sort[a] is not False and sort[a] is not None and not leg.sorted:
                p_qind, newleg = leg.sort(bunch=False)
                perms[a] = leg.perm_flat_from_perm_qind(p_qind)
                res.legs[a] = newleg
                inv = np.empty(len(p_qind), np.intp)
                inv[p_qind] = np.arange(len(p_qind), dtype=np.intp)
                res._qdata[:, a] = inv[res._qdata[:, a]]
                res._qdata_sorted = False
            if bunch[a] and not res.legs[a].bunched:
                bunch_axes.append(a)
        res._skey = None
        if bunch_axes:
            res = res._bunch(bunch_axes)
        return perms, res

    def _bunch(self, axes):
        """Merge neighbouring equal-charge blocks of the given legs (like reference :2529)."""
        # implemented through 1-leg pipes without sorting, which produce exactly the bunched leg
        pipes = []
        for a in axes:
            pipe = LegPipe([self.legs[a]], qconj=self.legs[a].qconj, sort=False, bunch=True)
            pipes.append(pipe)
        res = self.combine_legs([[a] for a in axes], new_axes=list(axes), pipes=pipes)
        for a in axes:
            res.legs[a] = res.legs[a].to_LegCharge()
        res._labels = list(self._labels)
        res._skey = None
        return res

    # ---- combine / split ---------------------------------------------------------------------------------------
    def combine_legs(self, combine_legs, new_axes=None, pipes=None, qconj=None):
        """Reshape: fuse groups of legs into pipes (reference :1561; worker _npc_helper.pyx:1013).

        The transpose needed to bring combined legs next to each other is folded into the device copy
        plan, so the data moves exactly once (old block -> slice of the zero-initialised fused block).
        """
        combine_legs = list(combine_legs)
        if not _is_iterable(combine_legs[0]):
            combine_legs = [combine_legs]
            if new_axes is not None:
                new_axes = _to_iterable(new_axes)
            if pipes is not None:
                pipes = _to_iterable(pipes)
        pipes = self._combine_legs_make_pipes(combine_legs, pipes, qconj)
        combine_legs = [np.asarray(self.get_leg_indices(cl), dtype=np.intp) for cl in combine_legs]
        all_cl = np.concatenate(combine_legs)
        if len(set(all_cl)) != len(all_cl):
            raise ValueError("got a leg multiple times: " + str(combine_legs))
        new_axes, transp = self._combine_legs_new_axes(combine_legs, new_axes)
        order = np.argsort(new_axes)
        combine_legs = [combine_legs[p] for p in order]
        pipes = [pipes[p] for p in order]
        new_axes = [int(new_axes[p]) for p in order]
        labels = [(l if l is not None else '?' + str(i)) for i, l in enumerate(self._labels)]
        non_combined = [a for a in range(self.rank) if a not in all_cl]
        # new legs / labels; src_axes[new_axis] = list of old axes feeding it (in order)
        legs = [self.legs[a] for a in non_combined]
        src_axes = [[a] for a in non_combined]
        new_labels = [labels[a] for a in non_combined]
        for na, p, cl in zip(new_axes, pipes, combine_legs):
            legs.insert(na, p)
            src_axes.insert(na, [int(c) for c in cl])
            new_labels.insert(na, '(' + '.'.join(labels[c] for c in cl) + ')')
        res = Array(legs, self.dtype, self.qtotal, new_labels)
        if self.stored_blocks == 0:
            return res
        # the bookkeeping below (new block list, offsets, copy jobs) depends on the block structure of ``self``, the leg groups and
        # the fusion tables of the pipes only: planned once, replayed for every Array of that structure (one per bond and sweep)
        pkey = ('comb', self._struct_key(), tuple(tuple(int(c) for c in cl) for cl in combine_legs), tuple(new_axes),
                tuple(p._content_key() for p in pipes))
        plan = _reshape_plan_get(pkey)
        if plan is not None:
            res._adopt_blocks(plan[0], plan[1], dev.zeros(plan[2], self.dtype), True)
            if plan[3] is not None:
                dev.check(dev.lib().tpa_copy_batch(dev.code(self.dtype), plan[3].data_ptr(), plan[4], plan[5], self._arena.data_ptr(),
                                                   res._arena.data_ptr(), dev.stream()), "copy_batch")
            return res
        nold = self.stored_blocks
        # --- new qdata and the slice start inside the fused block, per old block
        qdata = np.empty((nold, res.rank), dtype=np.intp)
        start = np.zeros((nold, res.rank), dtype=np.int64)
        for na, src in enumerate(src_axes):
            leg = legs[na]
            if na in new_axes:
                rows = leg._map_incoming_qind(self._qdata[:, src])
                qm = leg.q_map[rows]
                qdata[:, na] = qm[:, 2]
                start[:, na] = qm[:, 0]
            else:
                qdata[:, na] = self._qdata[:, src[0]]
        sort = np.lexsort(qdata.T)
        qdata_s = qdata[sort]
        diffs = _find_row_differences(qdata_s)
        new_qdata = qdata_s[diffs[:-1]]
        new_index_sorted = np.repeat(np.arange(len(diffs) - 1), np.diff(diffs))
        new_index = np.empty(nold, dtype=np.int64)
        new_index[sort] = new_index_sorted
        res._set_blocks(new_qdata, zero=True, qdata_sorted=True)
        new_shapes = res._block_shapes()
        # --- copy jobs: one per old block, ndim = old rank, iterating old axes in NEW order
        old_shapes = self._block_shapes()
        old_strides = _c_strides(old_shapes)
        # copy dims: one per source leg, except that a group of source legs that are neighbours in ascending order is
        # contiguous in the old block AND in the slice of the fused block -> a single dim
        dims = []           # (new axis, [source legs forming one dim])
        for na, src in enumerate(src_axes):
            if all(src[i + 1] == src[i] + 1 for i in range(len(src) - 1)):
                dims.append((na, list(src)))
            else:
                dims.extend((na, [a]) for a in src)
        nd = len(dims)
        if nd > COPY_MAXDIM and res.rank <= COPY_MAXDIM:
            return self._combine_legs_via_transpose(combine_legs, new_axes, pipes, transp)
        if nd > COPY_MAXDIM:
            # a result of rank > 6 only occurs in set-up utilities (MPS.from_full of >= 5 sites, networks/mps.py:2439): host copy
            host = dev.to_host(self._arena)
            out = np.zeros(int(res._arena.numel()), dtype=self.dtype)
            perm = [a for src in src_axes for a in src]
            for b in range(nold):
                blk = host[self._offsets[b]:self._offsets[b] + int(np.prod(old_shapes[b]))].reshape(tuple(old_shapes[b]))
                nb = int(new_index[b])
                shape_in = tuple(int(np.prod(old_shapes[b, src])) for src in src_axes)
                view = out[res._offsets[nb]:res._offsets[nb] + int(np.prod(new_shapes[nb]))].reshape(tuple(new_shapes[nb]))
                sl = tuple(slice(int(start[b, na]), int(start[b, na]) + shape_in[na]) for na in range(res.rank))
                view[sl] = np.transpose(blk, perm).reshape(shape_in)
            res._arena = dev.to_device(out)
            return res
        new_strides = _c_strides(new_shapes)[new_index]  # (nold, res.rank)
        jobs = np.zeros((nold, 4 + 3 * COPY_MAXDIM), dtype=np.int64)
        jobs[:, 2] = nd
        dst_off = res._offsets[new_index].copy()
        for na in range(res.rank):
            dst_off += start[:, na] * new_strides[:, na]
        for pos in range(nd - 1, -1, -1):
            na, legs_d = dims[pos]
            # C-order stride inside the slice of new axis `na`: product of the sizes of the later source legs of that axis
            later = [a for (na2, l2) in dims[pos + 1:] if na2 == na for a in l2]
            inner = np.prod(old_shapes[:, later], axis=1) if later else np.ones(nold, dtype=np.int64)
            jobs[:, 4 + pos] = np.prod(old_shapes[:, legs_d], axis=1)
            jobs[:, 4 + COPY_MAXDIM + pos] = inner * new_strides[:, na]
            jobs[:, 4 + 2 * COPY_MAXDIM + pos] = old_strides[:, legs_d[-1]]
        jobs[:, 0] = dst_off
        jobs[:, 1] = self._offsets
        sizes = np.prod(old_shapes, axis=1)
        _run_copy(self.dtype, jobs, int(np.max(sizes)), self._arena, res._arena)
        if 0 < len(jobs) <= 60000 and int(np.max(sizes)) > 0:
            _reshape_plan_put(pkey, (res._qdata, res._offsets, int(res._arena.numel()), dev.table(jobs), len(jobs), int(np.max(sizes))))
        return res

    def _combine_legs_via_transpose(self, combine_legs, new_axes, pipes, transp):
        """More than COPY_MAXDIM copy dims: bring the legs of every group next to each other first (``transp``: the final
        leg order with each pipe expanded into its legs), then every new axis is a single copy dim."""
        tr = self.transpose(transp)
        pos, groups = 0, []
        sizes = {na: len(cl) for na, cl in zip(new_axes, combine_legs)}
        n_new = self.rank - sum(sizes.values()) + len(sizes)
        for na in range(n_new):
            w = sizes.get(na, 1)
            if na in sizes:
                groups.append(list(range(pos, pos + w)))
            pos += w
        return tr.combine_legs(groups, new_axes=list(new_axes), pipes=list(pipes))

    def _combine_legs_make_pipes(self, combine_legs, pipes, qconj):
        npipes = len(combine_legs)
        if pipes is None:
            pipes = [None] * npipes
        elif len(pipes) != npipes:
            raise ValueError("wrong len of `pipes`")
        qconj = list(_to_iterable(qconj))
        if len(qconj) == 1 and 1 < npipes:
            qconj = [qconj[0]] * npipes
        if len(qconj) != npipes:
            raise ValueError("wrong len of `qconj`")
        pipes = list(pipes)
        for i, pipe in enumerate(pipes):
            if pipe is None:
                qc = qconj[i]
                if qc is None:
                    qc = self.get_leg(combine_legs[i][0]).qconj
                pipes[i] = self.make_pipe(axes=combine_legs[i], qconj=qc)
            else:
                legs = [self.get_leg(a) for a in combine_legs[i]]
                if pipe.nlegs != len(legs):
                    raise ValueError("pipe has wrong number of legs")
                if legs[0].qconj != pipe.legs[0].qconj:
                    pipes[i] = pipe = pipe.conj()
                for self_leg, pipe_leg in zip(legs, pipe.legs):
                    self_leg.test_equal(pipe_leg)
        return pipes

    def _combine_legs_new_axes(self, combine_legs, new_axes):
        all_cl = np.concatenate(combine_legs)
        non_combined = np.array([a for a in range(self.rank) if a not in all_cl], dtype=np.intp)
        if new_axes is None:
            first = np.array([cl[0] for cl in combine_legs])
            new_axes = [int(np.sum(non_combined < a) + np.sum(first < a)) for a in first]
        else:
            new_axes = list(new_axes)
            if len(new_axes) != len(combine_legs):
                raise ValueError("wrong len of `new_axes`")
            new_rank = len(combine_legs) + len(non_combined)
            for i, a in enumerate(new_axes):
                if a < 0:
                    new_axes[i] = a + new_rank
                elif a >= new_rank:
                    raise ValueError("new_axis larger than the new number of legs")
        transp = [[a] for a in non_combined]
        for s in np.argsort(new_axes):
            transp.insert(new_axes[s], list(combine_legs[s]))
        return new_axes, tuple(int(a) for a in sum(transp, []))

    def split_legs(self, axes=None, cutoff=0.):
        """Inverse of :meth:`combine_legs` (reference :1707; worker _npc_helper.pyx:1136)."""
        if axes is None:
            axes = [i for i, l in enumerate(self.legs) if isinstance(l, LegPipe)]
        else:
            axes = self.get_leg_indices(_to_iterable(axes))
            if len(set(axes)) != len(axes):
                raise ValueError("can't split a leg multiple times!")
        for ax in axes:
            if not isinstance(self.legs[ax], LegPipe):
                raise ValueError("can't split leg {ax:d} which is not a LegPipe".format(ax=ax))
        if len(axes) == 0:
            return self.copy(deep=True)
        # new legs / labels and the map old axis -> list of new axes
        res_legs, res_labels, new_of_old = [], [], []
        for a in range(self.rank):
            if a in axes:
                pipe = self.legs[a]
                new_of_old.append(list(range(len(res_legs), len(res_legs) + pipe.nlegs)))
                res_legs.extend(pipe.legs)
                res_labels.extend(self._split_leg_label(self._labels[a], pipe.nlegs))
            else:
                new_of_old.append([len(res_legs)])
                res_legs.append(self.legs[a])
                res_labels.append(self._labels[a])
        res = Array(res_legs, self.dtype, self.qtotal, res_labels)
        if self.stored_blocks == 0:
            return res
        pkey = ('split', self._struct_key(), tuple(int(a) for a in axes), tuple(self.legs[a]._content_key() for a in sorted(axes)))
        plan = _reshape_plan_get(pkey) if self.rank <= COPY_MAXDIM else None
        if plan is not None:
            res._adopt_blocks(plan[0], plan[1], dev.empty(plan[2], self.dtype), False)
            dev.check(dev.lib().tpa_copy_batch(dev.code(self.dtype), plan[3].data_ptr(), plan[4], plan[5], self._arena.data_ptr(),
                                               res._arena.data_ptr(), dev.stream()), "copy_batch")
            if cutoff > 0.:
                res.ipurge_zeros(cutoff)
            return res
        nold = self.stored_blocks
        split_axes = [a for a in range(self.rank) if a in axes]
        nsplit = len(split_axes)
        beg = np.zeros((nold, nsplit), dtype=np.intp)
        cnt = np.zeros((nold, nsplit), dtype=np.intp)
        for j, a in enumerate(split_axes):
            qms = self.legs[a].q_map_slices
            q = self._qdata[:, a]
            beg[:, j] = qms[q]
            cnt[:, j] = qms[q + 1] - qms[q]
        per_old = np.prod(cnt, axis=1)
        old_idx = np.repeat(np.arange(nold), per_old)
        nnew = len(old_idx)
        # enumerate q_map rows: C-order over split axes inside each old block
        local = np.arange(nnew) - np.repeat(np.cumsum(per_old) - per_old, per_old)
        rows = np.empty((nnew, nsplit), dtype=np.intp)
        rem = local.copy()
        for j in range(nsplit - 1, -1, -1):
            c = cnt[old_idx, j]
            rows[:, j] = beg[old_idx, j] + rem % c
            rem //= c
        new_qdata = np.empty((nnew, res.rank), dtype=np.intp)
        old_start = np.zeros((nnew, self.rank), dtype=np.int64)
        for a in range(self.rank):
            if a in split_axes:
                j = split_axes.index(a)
                qm = self.legs[a].q_map[rows[:, j]]
                new_qdata[:, new_of_old[a]] = qm[:, 3:]
                old_start[:, a] = qm[:, 0]
            else:
                new_qdata[:, new_of_old[a][0]] = self._qdata[old_idx, a]
        res._set_blocks(new_qdata, qdata_sorted=False)
        new_shapes = res._block_shapes()
        # copy dims: ONE per old axis -- the legs a pipe is split into subdivide a contiguous index range of the old
        # block in C order and are neighbours in the new block, so they move as a single dim
        if self.rank > COPY_MAXDIM:         # set-up utilities only (see combine_legs): host copy
            host = dev.to_host(self._arena)
            out = np.empty(int(res._arena.numel()), dtype=self.dtype)
            shapes_old = self._block_shapes()
            for k in range(nnew):
                b = int(old_idx[k])
                blk = host[self._offsets[b]:self._offsets[b] + int(np.prod(shapes_old[b]))].reshape(tuple(shapes_old[b]))
                ext = [int(np.prod(new_shapes[k, new_of_old[a]])) for a in range(self.rank)]
                sl = tuple(slice(int(old_start[k, a]), int(old_start[k, a]) + ext[a]) for a in range(self.rank))
                n = int(np.prod(new_shapes[k]))
                out[res._offsets[k]:res._offsets[k] + n] = blk[sl].reshape(-1)
            res._arena = dev.to_device(out)
            if cutoff > 0.:
                res.ipurge_zeros(cutoff)
            return res
        old_shapes = self._block_shapes()[old_idx]
        old_strides = _c_strides(old_shapes)
        new_strides = _c_strides(new_shapes)
        jobs = np.zeros((nnew, 4 + 3 * COPY_MAXDIM), dtype=np.int64)
        jobs[:, 2] = self.rank
        src_off = self._offsets[old_idx].copy()
        for a in range(self.rank):
            src_off += old_start[:, a] * old_strides[:, a]
            group = list(new_of_old[a])
            jobs[:, 4 + a] = np.prod(new_shapes[:, group], axis=1)
            jobs[:, 4 + COPY_MAXDIM + a] = new_strides[:, group[-1]]
            jobs[:, 4 + 2 * COPY_MAXDIM + a] = old_strides[:, a]
        jobs[:, 0] = res._offsets
        jobs[:, 1] = src_off
        sizes = np.prod(new_shapes, axis=1)
        _run_copy(self.dtype, jobs, int(np.max(sizes)) if nnew else 0, self._arena, res._arena)
        if 0 < nnew <= 60000 and int(np.max(sizes)) > 0:
            _reshape_plan_put(pkey, (res._qdata, res._offsets, int(res._arena.numel()), dev.table(jobs), nnew, int(np.max(sizes))))
        if cutoff > 0.:
            res.ipurge_zeros(cutoff)
        return res

    @staticmethod
    def _combine_leg_labels(labels):
        """``'(a.b.(c.d))'`` for ``['a', 'b', '(c.d)']`` (reference :2852)."""
        return '(' + '.'.join(labels) + ')'

    @staticmethod
    def _split_leg_label(label, count):
        if label is None:
            return [None] * count
        if label[0] != '(' or label[-1] != ')':
            warnings.warn("split leg with label not in Form '(...)': " + repr(label), stacklevel=3)
            return [None] * count
        depth, beg, res = 0, 1, []
        for i in range(1, len(label) - 1):
            c = label[i]
            if c == '(':
                depth += 1
            elif c == ')':
                depth -= 1
            elif c == '.' and depth == 0:
                res.append(label[beg:i])
                beg = i + 1
        res.append(label[beg:len(label) - 1])
        if len(res) != count:
            raise ValueError("wrong number of splitted labels.")
        return [None if r[0] == '?' else r for r in res]

    @staticmethod
    def _conj_leg_label(label):
        """'a' -> 'a*', 'a*' -> 'a', '(a.(b*.c))' -> '(a*.(b.c*))'."""
        if label is None:
            return None
        out, i, n = [], 0, len(label)
        while i < n:
            c = label[i]
            if c in '().':
                out.append(c)
                i += 1
                continue
            j = i
            while j < n and label[j] not in '().':
                j += 1
            name = label[i:j]
            out.append(name[:-1] if name.endswith('*') else name + '*')
            i = j
        return ''.join(out)

    def ipurge_zeros(self, cutoff=QCUTOFF, norm_order=None):
        """Drop blocks whose 2-norm is below ``cutoff`` (host decision on per-block norms)."""
        if self.stored_blocks == 0 or cutoff <= 0:
            return self
        n2 = self._block_norms_sq()
        keep = np.sqrt(n2) > cutoff
        if not np.all(keep):
            self._qdata = np.ascontiguousarray(self._qdata[keep])
            self._offsets = self._offsets[keep]
            self._skey = None
        return self

    def _block_norms_sq(self):
        sizes = self._block_sizes_flat()
        out, scr = dev.reduction_buffers()
        res = np.empty(self.stored_blocks)
        L = dev.lib()
        base = self._arena.data_ptr()
        esz = self._arena.element_size()
        for i, (o, s) in enumerate(zip(self._offsets, sizes)):
            dev.check(L.tpa_nrm2sq(dev.code(self.dtype), int(s), base + int(o) * esz, out.data_ptr(), scr.data_ptr(), dev.stream()))
            res[i] = dev.read_scalar(out, False)
        return res

    # ---- transpose -----------------------------------------------------------------------------------------------
    def itranspose(self, axes=None):
        """Permute legs.  Blocks are physically transposed by one batched device copy (K8)."""
        if axes is None:
            axes = tuple(reversed(range(self.rank)))
        else:
            axes = tuple(self.get_leg_indices(axes))
            if len(axes) != self.rank or len(set(axes)) != self.rank:
                raise ValueError("axes has wrong length: " + str(axes))
        if axes == tuple(range(self.rank)):
            return self
        axes_arr = np.array(axes, dtype=np.intp)
        pkey = ('transp', self._struct_key(), axes) if self.stored_blocks and self.rank <= COPY_MAXDIM else None
        plan = _reshape_plan_get(pkey) if pkey is not None else None
        if plan is not None:        # (see combine_legs: the bookkeeping is planned once per block structure)
            self.legs = [self.legs[a] for a in axes]
            self._set_shape()
            self._labels = [self._labels[a] for a in axes]
            new_arena = dev.empty(plan[2], self.dtype)
            dev.check(dev.lib().tpa_copy_batch(dev.code(self.dtype), plan[3].data_ptr(), plan[4], plan[5], self._arena.data_ptr(),
                                               new_arena.data_ptr(), dev.stream()), "copy_batch")
            self._adopt_blocks(plan[0], plan[1], new_arena, False)
            return self
        old_shapes = self._block_shapes()
        old_strides = _c_strides(old_shapes)
        self.legs = [self.legs[a] for a in axes]
        self._set_shape()
        self._labels = [self._labels[a] for a in axes]
        self._qdata = np.ascontiguousarray(self._qdata[:, axes_arr])
        self._qdata_sorted = False
        self._skey = None
        if self.stored_blocks == 0:
            return self
        new_shapes = old_shapes[:, axes_arr]
        sizes = np.prod(new_shapes, axis=1)
        # blocks whose memory order does not change need no copy; still repack everything into a new arena
        new_offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        # legs that stay neighbours in the same order move as one copy dim
        runs = [[0]]
        for i in range(1, self.rank):
            if axes[i] == axes[i - 1] + 1:
                runs[-1].append(i)
            else:
                runs.append([i])
        nd = len(runs)
        if nd > COPY_MAXDIM:        # tensors of rank > 6 only occur in set-up utilities (grouping sites, H conversions)
            host = dev.to_host(self._arena)
            out = np.empty(int(np.sum(sizes)), dtype=self.dtype)
            for b in range(self.stored_blocks):
                blk = host[self._offsets[b]:self._offsets[b] + sizes[b]].reshape(tuple(old_shapes[b]))
                out[new_offs[b]:new_offs[b] + sizes[b]] = np.transpose(blk, axes).reshape(-1)
            self._arena = dev.to_device(out)
            self._offsets = new_offs
            return self
        new_str_full = _c_strides(new_shapes)
        old_str_perm = old_strides[:, axes_arr]
        jobs = np.zeros((self.stored_blocks, 4 + 3 * COPY_MAXDIM), dtype=np.int64)
        jobs[:, 0] = new_offs
        jobs[:, 1] = self._offsets
        jobs[:, 2] = nd
        for d, run in enumerate(runs):
            jobs[:, 4 + d] = np.prod(new_shapes[:, run], axis=1)
            jobs[:, 4 + COPY_MAXDIM + d] = new_str_full[:, run[-1]]
            jobs[:, 4 + 2 * COPY_MAXDIM + d] = old_str_perm[:, run[-1]]
        new_arena = dev.empty(int(np.sum(sizes)), self.dtype)
        _run_copy(self.dtype, jobs, int(np.max(sizes)), self._arena, new_arena)
        self._arena = new_arena
        self._offsets = new_offs
        if pkey is not None and 0 < len(jobs) <= 60000 and int(np.max(sizes)) > 0:
            _reshape_plan_put(pkey, (self._qdata, self._offsets, int(np.sum(sizes)), dev.table(jobs), len(jobs), int(np.max(sizes))))
        return self

    def transpose(self, axes=None):
        """``itranspose`` on a copy (reference :2084); the permuted blocks go to a new arena, an identity permutation
        leaves the arena shared copy-on-write."""
        res = self.copy(deep=False)
        res._qdata = self._qdata.copy()
        return res.itranspose(axes)

    def iswapaxes(self, axis1, axis2):
        axis1, axis2 = self.get_leg_index(axis1), self.get_leg_index(axis2)
        if axis1 == axis2:
            return self
        axes = list(range(self.rank))
        axes[axis1], axes[axis2] = axis2, axis1
        return self.itranspose(axes)

    def _transpose_same_labels(self, other_labels):
        """``self`` with its legs in the order of ``other_labels`` if both carry the same complete set of labels."""
        other_labels = list(other_labels)
        if self._labels == other_labels or None in self._labels or None in other_labels:
            return self
        if set(self._labels) == set(other_labels):
            return self.transpose(other_labels)
        return self

    # ---- elementwise / scaling -------------------------------------------------------------------------------------
    def iscale_axis(self, s, axis=-1):
        """Multiply by ``s[i]`` along ``axis`` (reference :2108); ``s`` is a host 1-D array."""
        axis = self.get_leg_index(axis)
        s = np.asarray(s)
        if s.shape != (self.shape[axis],):
            raise ValueError("s has wrong shape: " + str(s.shape) + " instead of " + str(self.shape[axis]))
        if s.dtype.kind == 'c' and self.dtype.kind != 'c':
            self._become(self.astype(np.complex128))
        if self.stored_blocks == 0:
            return self
        self._own_arena()
        s_cplx = s.dtype.kind == 'c'
        s_dev = dev.to_device(s.astype(np.complex128 if s_cplx else np.float64))
        shapes = self._block_shapes()
        jobs = np.zeros((self.stored_blocks, 6), dtype=np.int64)
        jobs[:, 0] = self._offsets
        jobs[:, 1] = np.prod(shapes[:, :axis], axis=1)
        jobs[:, 2] = shapes[:, axis]
        jobs[:, 3] = np.prod(shapes[:, axis + 1:], axis=1)
        jobs[:, 4] = self.legs[axis].slices[self._qdata[:, axis]]
        jd = dev.table(jobs)
        dev.check(dev.lib().tpa_scale_axis_batch(dev.code(self.dtype), jd.data_ptr(), len(jobs),
                                                 int(np.max(np.prod(shapes, axis=1))), self._arena.data_ptr(),
                                                 s_dev.data_ptr(), int(s_cplx), dev.stream()), "scale_axis")
        return self

    def scale_axis(self, s, axis=-1):
        return self.copy(deep=True).iscale_axis(s, axis)

    def _become(self, other):
        other._arena
        self._drop_cow()
        self.__dict__.update(other.__dict__)

    def astype(self, dtype, copy=True):
        dtype = _calc_dtype(dtype)
        if dtype == self.dtype:               # (reference :1882: a new Array object also for copy=False)
            return self.copy(deep=bool(copy))
        res = self.copy(deep=False)
        res._qdata = self._qdata.copy()
        res._offsets = self._offsets.copy()
        res.dtype = dtype
        res._skey = None
        if self._arena is not None:
            n = self._arena.numel()
            res._arena = dev.empty(n, dtype)
            dev.check(dev.lib().tpa_convert(dev.code(self.dtype), dev.code(dtype), n, self._arena.data_ptr(),
                                            res._arena.data_ptr(), 0, dev.stream()), "convert")
        return res

    def iconj(self, complex_conj=True):
        return self.conj(complex_conj, inplace=True)

    def conj(self, complex_conj=True, inplace=False):
        """Conjugate: complex conjugate data, conjugate all legs, negate qtotal, toggle '*' on labels."""
        res = self if inplace else self.copy(deep=True)
        if complex_conj and res.dtype.kind == 'c' and res._arena is not None:
            res._own_arena()
            n = res._arena.numel()
            dev.check(dev.lib().tpa_convert(1, 1, n, res._arena.data_ptr(), res._arena.data_ptr(), 1, dev.stream()), "conj")
        res.qtotal = res.chinfo.make_valid(-res.qtotal)
        res.legs = [leg.conj() for leg in res.legs]
        res._labels = [Array._conj_leg_label(l) for l in res._labels]
        return res

    def complex_conj(self):
        res = self.copy(deep=True)
        if res.dtype.kind == 'c' and res._arena is not None:
            n = res._arena.numel()
            dev.check(dev.lib().tpa_convert(1, 1, n, res._arena.data_ptr(), res._arena.data_ptr(), 1, dev.stream()), "conj")
        return res

    def norm(self, ord=None, convert_to_float=True):
        """``np.linalg.norm(self.to_ndarray().flatten(), ord)`` (reference :2241): the 2-norm in one fused pass of the reduction
        kernel; the other orders (inf, -inf, 0, 1, any p) as one element-wise reduction over the arena.  The zero entries
        outside the stored blocks only matter for ``ord=-inf`` (minimum is 0 unless every entry is stored)."""
        if ord not in (None, 2, 'fro'):
            # the other orders (1, inf, -inf, 0, p) are a host reduction over a download of the stored entries, like the reference's
            # `np.linalg.norm` over the block list (np_conserved.py:2252): off every hot path, and no vendor kernel in the product
            if self.stored_blocks == 0:
                return 0.
            flat = self._arena if self._is_packed() else self.copy(deep=True)._repack()._arena
            val = float(np.linalg.norm(np.asarray(dev.to_host(flat)).reshape(-1), ord=ord))
            if ord == -np.inf and flat.numel() < self.size:
                val = 0.
            return val
        if self.stored_blocks == 0:
            return 0.
        if self._is_packed():
            out, scr = dev.reduction_buffers()
            dev.check(dev.lib().tpa_nrm2sq(dev.code(self.dtype), self._arena.numel(), self._arena.data_ptr(),
                                           out.data_ptr(), scr.data_ptr(), dev.stream()), "nrm2")
            return float(np.sqrt(dev.read_scalar(out, False)))
        return float(np.sqrt(np.sum(self._block_norms_sq())))

    def __neg__(self):
        return self.copy(deep=True).iscale_prefactor(-1.)

    def axis_sqnorms(self, axis):
        """Host array ``out[j] = sum |self[..., j, ...]|^2`` over everything but ``axis`` (one device pass)."""
        axis = self.get_leg_index(axis)
        res = np.zeros(self.shape[axis])
        if self.stored_blocks == 0:
            return res
        shapes = self._block_shapes()
        nb = self.stored_blocks
        lens = shapes[:, axis]
        o_offs = np.concatenate([[0], np.cumsum(lens)])
        jobs = np.zeros((nb, 6), dtype=np.int64)
        jobs[:, 0] = self._offsets
        jobs[:, 1] = np.prod(shapes[:, :axis], axis=1)
        jobs[:, 2] = lens
        jobs[:, 3] = np.prod(shapes[:, axis + 1:], axis=1)
        jobs[:, 4] = o_offs[:-1]
        rows = np.stack([np.repeat(np.arange(nb), lens), np.arange(int(o_offs[-1])) - np.repeat(o_offs[:-1], lens)], axis=1)
        pad = (-len(rows)) % 4
        if pad:
            rows = np.concatenate([rows, np.full((pad, 2), -1)], axis=0)
        out = dev.zeros(int(o_offs[-1]), np.float64)
        jd, rd = dev.table(jobs), dev.table(rows.astype(np.int32))
        dev.check(dev.lib().tpa_axis_sqnorm_batch(dev.code(self.dtype), jd.data_ptr(), rd.data_ptr(), len(rows),
                                                  self._arena.data_ptr(), out.data_ptr(), dev.stream()), "axis_sqnorm")
        host = dev.to_host(out)
        leg = self.legs[axis]
        for b in range(nb):
            q = self._qdata[b, axis]
            res[leg.slices[q]:leg.slices[q + 1]] += host[o_offs[b]:o_offs[b + 1]]
        return res

    def iscale_prefactor(self, prefactor):
        """``self *= prefactor`` (reference :2385 / _npc_helper.pyx:964); 0 drops all blocks."""
        if not np.isscalar(prefactor) and not isinstance(prefactor, (np.generic,)):
            raise ValueError("prefactor is not scalar: {0!r}".format(type(prefactor)))
        if prefactor == 0.:
            self._qdata = np.empty((0, self.rank), np.intp)
            self._offsets = np.zeros(0, np.int64)
            self._arena = None
            self._qdata_sorted = True
            self._skey = None
            return self
        if isinstance(prefactor, complex) or np.iscomplexobj(prefactor):
            if self.dtype.kind != 'c':
                self._become(self.astype(np.complex128))
        if self.stored_blocks == 0:
            return self
        p = complex(prefactor)
        self._repack()
        self._own_arena()
        dev.check(dev.lib().tpa_scal(dev.code(self.dtype), self._arena.numel(), p.real, p.imag,
                                     self._arena.data_ptr(), dev.stream()), "scal")
        return self

    def iadd_prefactor_other(self, prefactor, other):
        """``self += prefactor * other`` (reference :2372 / _npc_helper.pyx:860)."""
        other = other._transpose_same_labels(self._labels)
        if self.rank != other.rank:
            raise ValueError("different rank!")
        for self_leg, other_leg in zip(self.legs, other.legs):
            if self_leg is not other_leg:                   # (Krylov vectors share their leg objects: nothing to compare)
                self_leg.test_equal(other_leg)
        if np.any(self.qtotal != other.qtotal):
            raise ValueError("Arrays can't have different `qtotal`!")
        if self.legs[0].chinfo is not other.legs[0].chinfo and self.legs[0].chinfo != other.legs[0].chinfo:
            raise ValueError("Arrays have different ChargeInfo")
        if prefactor == 0. or other.stored_blocks == 0:
            return self
        calc = _calc_dtype(self.dtype, other.dtype, type(prefactor))
        if self.dtype != calc:
            self._become(self.astype(calc))
        if other.dtype != calc:
            other = other.astype(calc)
        p = complex(prefactor)
        L = dev.lib()
        if self.stored_blocks and self._same_structure(other) and self._is_packed():
            self._own_arena()
            dev.check(L.tpa_axpy(dev.code(calc), self._arena.numel(), p.real, p.imag, other._arena.data_ptr(),
                                 self._arena.data_ptr(), dev.stream()), "axpy")
            return self
        # different sparsity patterns: union layout, then flat axpy
        allq = np.concatenate([self._qdata, other._qdata], axis=0)
        uq = np.unique(allq, axis=0)
        uq = uq[np.lexsort(uq.T)]
        lay = Array(self.legs, calc, self.qtotal, self._labels)
        lay._set_blocks(uq, zero=True, qdata_sorted=True)
        tmp = dev.zeros(lay._arena.numel(), calc)
        _scatter_blocks(self, lay, lay._arena)
        _scatter_blocks(other, lay, tmp)
        dev.check(L.tpa_axpy(dev.code(calc), lay._arena.numel(), p.real, p.imag, tmp.data_ptr(),
                             lay._arena.data_ptr(), dev.stream()), "axpy")
        self._qdata, self._offsets, self._arena = lay._qdata, lay._offsets, lay._arena
        self._qdata_sorted = True
        self._skey = None
        return self

    def __add__(self, other):
        if isinstance(other, Array):
            return self.copy(deep=True).iadd_prefactor_other(1., other)
        return NotImplemented

    def __iadd__(self, other):
        if isinstance(other, Array):
            return self.iadd_prefactor_other(1., other)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, Array):
            return self.copy(deep=True).iadd_prefactor_other(-1., other)
        return NotImplemented

    def __isub__(self, other):
        if isinstance(other, Array):
            return self.iadd_prefactor_other(-1., other)
        return NotImplemented

    def __mul__(self, other):
        if np.isscalar(other) or isinstance(other, np.generic):
            return self.copy(deep=True).iscale_prefactor(other)
        return NotImplemented

    __rmul__ = __mul__

    def __imul__(self, other):
        if np.isscalar(other) or isinstance(other, np.generic):
            return self.iscale_prefactor(other)
        return NotImplemented

    def __truediv__(self, other):
        if np.isscalar(other) or isinstance(other, np.generic):
            if other == 0.:
                raise ZeroDivisionError("a/b for b=0. Types: {0!s}, {1!s}".format(type(self), type(other)))
            return self.copy(deep=True).iscale_prefactor(1. / other)
        return NotImplemented

    def __itruediv__(self, other):
        if np.isscalar(other) or isinstance(other, np.generic):
            if other == 0.:
                raise ZeroDivisionError("a/b for b=0. Types: {0!s}, {1!s}".format(type(self), type(other)))
            return self.iscale_prefactor(1. / other)
        return NotImplemented

    # ---- projection ---------------------------------------------------------------------------------------------------
    def iproject(self, mask, axes):
        """Keep only the indices selected by ``mask`` (bool or index array) on the given ``axes``
        (reference :1914).  Returns ``(map_qind, block_masks)`` lists per axis; ``self`` is modified."""
        if not _is_iterable(axes):          # a single axis goes with a single mask (reference :1946)
            mask = [mask]
        axes = self.get_leg_indices(_to_iterable(axes))
        mask = list(mask)
        if len(axes) == 0:
            return [], []
        if len(axes) != len(mask):
            raise ValueError("len(axes) != len(mask)")
        masks = []
        for m, a in zip(mask, axes):
            m = np.asarray(m)
            if m.dtype != np.bool_:
                mm = np.zeros(self.shape[a], dtype=np.bool_)
                np.put(mm, m, True)
                m = mm
            if m.shape != (self.shape[a],):
                raise ValueError("mask has wrong length")
            masks.append(m)
        map_qinds, all_block_masks = [], []
        cur = self
        for m, a in zip(masks, axes):
            map_qind, block_masks, new_leg = cur.legs[a].project(m)
            cur = cur._project_axis(a, m, map_qind, new_leg)
            map_qinds.append(map_qind)
            all_block_masks.append(block_masks)
        self._become(cur)
        return map_qinds, all_block_masks

    def _project_axis(self, axis, mask, map_qind, new_leg):
        res = self.copy(deep=False)
        res.legs = list(self.legs)
        res.legs[axis] = new_leg
        res._set_shape()
        res._skey = None
        if self.stored_blocks == 0:
            return res
        pkey = ('proj', self._struct_key(), int(axis), _blake2b(np.ascontiguousarray(mask).tobytes(), digest_size=16).digest())
        plan = _reshape_plan_get(pkey)
        if plan is not None:
            res._adopt_blocks(plan[0], plan[1], dev.empty(plan[2], self.dtype), self._qdata_sorted)
            if plan[3] is not None:
                dev.check(dev.lib().tpa_gather_axis_batch(dev.code(self.dtype), plan[3].data_ptr(), plan[5], plan[6], plan[4].data_ptr(),
                                                          self._arena.data_ptr(), res._arena.data_ptr(), dev.stream()), "gather")
            return res
        keep = map_qind[self._qdata[:, axis]] >= 0
        old_q = self._qdata[keep]
        old_off = self._offsets[keep]
        old_shapes = self._block_shapes()[keep]
        new_q = old_q.copy()
        new_q[:, axis] = map_qind[old_q[:, axis]]
        old_leg = self.legs[axis]
        res._set_blocks(new_q, qdata_sorted=self._qdata_sorted)
        if len(new_q) == 0:
            _reshape_plan_put(pkey, (res._qdata, res._offsets, 0, None, None, 0, 0))
            return res
        new_shapes = res._block_shapes()
        # per old qindex of this leg: list of kept local indices, packed into one index array
        idx_chunks, idx_off = [], {}
        at = 0
        for q in np.unique(old_q[:, axis]):
            loc = np.nonzero(mask[old_leg.slices[q]:old_leg.slices[q + 1]])[0].astype(np.int64)
            idx_off[int(q)] = at
            idx_chunks.append(loc)
            at += len(loc)
        idx = np.concatenate(idx_chunks) if idx_chunks else np.zeros(0, np.int64)
        jobs = np.zeros((len(new_q), 8), dtype=np.int64)
        jobs[:, 0] = res._offsets
        jobs[:, 1] = old_off
        jobs[:, 2] = np.prod(old_shapes[:, :axis], axis=1)
        jobs[:, 3] = old_shapes[:, axis]
        jobs[:, 4] = new_shapes[:, axis]
        jobs[:, 5] = np.prod(old_shapes[:, axis + 1:], axis=1)
        jobs[:, 6] = [idx_off[int(q)] for q in old_q[:, axis]]
        jd, idd = dev.table(jobs), dev.table(idx if len(idx) else np.zeros(1, np.int64))
        dev.check(dev.lib().tpa_gather_axis_batch(dev.code(self.dtype), jd.data_ptr(), len(jobs),
                                                  int(np.max(np.prod(new_shapes, axis=1))), idd.data_ptr(),
                                                  self._arena.data_ptr(), res._arena.data_ptr(), dev.stream()), "gather")
        _reshape_plan_put(pkey, (res._qdata, res._offsets, int(res._arena.numel()), jd, idd, len(jobs), int(np.max(np.prod(new_shapes, axis=1)))))
        return res

    def take_slice(self, indices, axes):
        """Copy of ``self`` with the (flat) ``indices`` fixed on the legs ``axes``, which are dropped: for rank 4,
        ``A.take_slice([i, j], [1, 2]) == A[:, i, j, :]`` (reference :1037).  Per axis: the gather kernel of ``iproject``
        with a one-entry mask, then the leg of length 1 is removed from the bookkeeping (the memory layout of a block does
        not change when a dimension of size 1 is dropped)."""
        axes = self.get_leg_indices(_to_iterable(axes))
        indices = np.asarray(_to_iterable(indices), dtype=np.intp)
        if len(axes) != len(indices):
            raise ValueError("len(axes) != len(indices)")
        if indices.ndim != 1:
            raise ValueError("indices may only contain ints")
        if len(axes) == 0:
            return self.copy(deep=True)
        if len(axes) == self.rank:
            raise ValueError("can't have 0-rank tensors")       # (the reference returns a scalar via __getitem__ for this)
        cur = self
        qtotal = self.qtotal.copy()
        for a, i in zip(axes, indices):
            leg = self.legs[a]
            if not -leg.ind_len <= i < leg.ind_len:
                raise IndexError("flat index %d out of bounds for leg of length %d" % (i, leg.ind_len))
            i = int(i) % leg.ind_len
            qi, _ = leg.get_qindex(i)
            qtotal = qtotal - leg.get_charge(qi)
            m = np.zeros(leg.ind_len, dtype=np.bool_)
            m[i] = True
            map_qind, _, new_leg = leg.project(m)
            cur = cur._project_axis(a, m, map_qind, new_leg)
        if cur is self or cur._arena is self._arena:
            cur = cur.copy(deep=True)
        keep_axes = [a for a in range(self.rank) if a not in axes]
        res = cur.copy(deep=False)
        res.legs = [cur.legs[a] for a in keep_axes]
        res._set_shape()
        res._labels = [self._labels[a] for a in keep_axes]
        res.qtotal = self.chinfo.make_valid(qtotal)
        res._qdata = np.ascontiguousarray(cur._qdata[:, keep_axes])
        res._skey = None
        return res

    # ---- element-wise functions --------------------------------------------------------------------------------------------
    def iunary_blockwise(self, func, *args, **kwargs):
        """``block = func(block, *args, **kwargs)`` on every stored block (reference :2152).  On the device the blocks
        are one arena, so ``func`` is applied ONCE to the flat arena by the element-wise function of the same name
        (``np.conj, np.real, np.imag, np.abs, np.angle, np.sqrt, np.exp, np.log, np.square, np.negative, np.sign``,
        ...: everything that maps 0 to 0 or not is the caller's business, like in the reference).  There is no host
        fallback: a Python callable without a device counterpart raises ``NotImplementedError``."""
        name = getattr(func, '__name__', None)
        name = {'conjugate': 'conj', 'absolute': 'abs'}.get(name, name)
        import torch as t
        f = getattr(t, name, None) if name else None
        if f is None or name not in _UNARY_OK:
            raise NotImplementedError("tenpy_amd: no device implementation of the element-wise function %r" % (func,))
        if self._arena is not None and self._arena.numel() > 0:
            if name == 'imag' and not self._arena.is_complex():
                out = t.zeros_like(self._arena)
            elif name == 'angle' and not self._arena.is_complex():
                out = t.angle(self._arena)          # 0 or pi
            else:
                out = f(self._arena, *args, **kwargs)
            if out.is_conj():
                out = out.resolve_conj()
            if out.dtype not in (t.float64, t.complex128):
                out = out.to(t.complex128 if out.is_complex() else t.float64)
            self._arena = out.contiguous()
            self.dtype = np.dtype(np.complex128 if out.is_complex() else np.float64)
        elif name in ('real', 'imag', 'abs', 'angle'):
            self.dtype = np.dtype(np.float64)
        self._skey = None
        return self

    def unary_blockwise(self, func, *args, **kwargs):
        """Copy of ``self`` with ``func`` applied to every stored block (reference :2196)."""
        res = self.copy(deep=True)
        return res.iunary_blockwise(func, *args, **kwargs)

    def matvec(self, other):
        """``tensordot(self, other, axes=1)``: rank-2 matrix times rank-1 vector, the interface the Krylov solvers
        call (reference :2364)."""
        return tensordot(self, other, axes=1)

    # ---- misc ------------------------------------------------------------------------------------------------------------
    def gauge_total_charge(self, axis, newqtotal=None, new_qconj=None):
        """Change ``qtotal`` by shifting the charges of one leg (reference :1198)."""
        res = self.copy(deep=False)
        ax = self.get_leg_index(axis)
        old = self.legs[ax]
        if new_qconj is None:
            new_qconj = old.qconj
        if new_qconj not in (-1, +1):
            raise ValueError("invalid new_qconj")
        res.qtotal = self.chinfo.make_valid(newqtotal).copy()
        shifted = old.charges + old.qconj * (res.qtotal - self.qtotal)
        if new_qconj != old.qconj:
            shifted = -shifted
        res.legs = list(self.legs)
        res.legs[ax] = LegCharge.from_qind(self.chinfo, old.slices, self.chinfo.make_valid(shifted), new_qconj)
        res._skey = None
        return res

    def add_trivial_leg(self, axis=0, label=None, qconj=1):
        if axis < 0:
            axis += self.rank + 1
        res = self.copy(deep=True)
        leg = LegCharge.from_trivial(1, self.chinfo, qconj)
        res.legs.insert(axis, leg)
        res._labels.insert(axis, label)
        res._set_shape()
        res._qdata = np.ascontiguousarray(np.insert(self._qdata, axis, 0, axis=1)).astype(np.intp)
        res._skey = None
        return res

    def add_leg(self, leg, i, axis=0, label=None):
        """Insert ``leg`` at ``axis`` with the data of ``self`` at its flat index ``i`` and zeros elsewhere -- the inverse of
        ``take_slice`` (reference :1123).  Bookkeeping only when the charge sector of ``i`` has one entry (MPO legs); a wider
        sector gets its blocks embedded by one strided copy launch."""
        if axis < 0:
            axis += self.rank
        if label is not None and label in self._labels:
            raise ValueError("label already exists")
        qi, pos = leg.get_qindex(int(i) % leg.ind_len if -leg.ind_len <= i < leg.ind_len else int(i))
        width = int(leg.get_block_sizes()[qi])
        legs = list(self.legs)
        legs.insert(axis, leg)
        labels = list(self._labels)
        labels.insert(axis, label)
        res = Array(legs, self.dtype, self.chinfo.make_valid(self.qtotal + leg.get_charge(qi)), labels)
        qdata = np.ascontiguousarray(np.insert(self._qdata, axis, qi, axis=1)).astype(np.intp)
        if self.stored_blocks == 0:
            return res
        if width == 1:
            src = self.copy(deep=True)
            src._repack()
            res._set_blocks(qdata, arena=src._arena, qdata_sorted=False)
            return res
        res._set_blocks(qdata, zero=True, qdata_sorted=False)
        shapes = self._block_shapes()
        outer = np.prod(shapes[:, :axis], axis=1)
        inner = np.prod(shapes[:, axis:], axis=1)
        # block [outer, width, inner] <- self block [outer, inner] at position pos of the new axis: `outer` contiguous runs
        dst_off, src_off, sizes = [], [], []
        for b in range(self.stored_blocks):
            o = np.arange(outer[b], dtype=np.int64)
            dst_off.append(res._offsets[b] + (o * width + pos) * inner[b])
            src_off.append(self._offsets[b] + o * inner[b])
            sizes.append(np.full(outer[b], inner[b], dtype=np.int64))
        dst_off, src_off, sizes = np.concatenate(dst_off), np.concatenate(src_off), np.concatenate(sizes)
        _run_copy(self.dtype, _copy_jobs_contiguous(dst_off, src_off, sizes), int(np.max(sizes)), self._arena, res._arena)
        return res

    def squeeze(self, axes=None):
        """Remove length-1 legs (only the trivial case with zero charge on the removed leg's single block is
        handled without charge compensation, like the reference does via qtotal adjustment)."""
        if axes is None:
            axes = [a for a in range(self.rank) if self.shape[a] == 1]
        else:
            axes = self.get_leg_indices(_to_iterable(axes))
        for a in axes:
            if self.shape[a] != 1:
                raise ValueError("Tried to squeeze non-unit leg")
        keep = [a for a in range(self.rank) if a not in axes]
        if len(keep) == 0:
            v = self.to_ndarray()
            return v.reshape(-1)[0]
        res = self.copy(deep=True)
        res.legs = [self.legs[a] for a in keep]
        res._labels = [self._labels[a] for a in keep]
        res._set_shape()
        res._qdata = np.ascontiguousarray(self._qdata[:, keep])
        for a in axes:
            res.qtotal = res.qtotal - self.legs[a].get_charge(0)
        res.qtotal = self.chinfo.make_valid(res.qtotal)
        res._skey = None
        return res


# ======================================================================================================
# helpers on copy jobs
# ======================================================================================================

_reshape_plans = OrderedDict()
_RESHAPE_PLANS_MAX = 16384


def clear_device_caches():
    """Forget every cached device table / plan / scratch buffer (tests that switch between the emulated and the real device)."""
    _plan_cache.clear()
    _reshape_plans.clear()
    dev._pool.clear()
    if dev._table_cache is not None:
        dev._table_cache.clear()
        dev._table_bytes = 0
    _svd_warm._tables.clear()
    _svd_warm._plans.clear()
    _svd_warm.cache_clear()
    try:
        from ..algorithms import mps_common as _mc
        _mc._heff_plans.clear()
        _mc.MpoApplyPlan._cache.clear()
    except ImportError:      # (the linalg package is importable on its own)
        pass


def _reshape_plan_get(key):
    pl = _reshape_plans.get(key)
    if pl is not None:
        try:
            _reshape_plans.move_to_end(key)
        except KeyError:      # evicted by another thread in between (the reference's `+ h.c.` worker contracts concurrently)
            pass
    return pl


def _reshape_plan_put(key, plan):
    for arr in plan[:2]:        # block list and offsets are handed to every Array that replays the plan: nobody may write into them
        if isinstance(arr, np.ndarray):
            arr.setflags(write=False)
    _reshape_plans[key] = plan
    if len(_reshape_plans) > _RESHAPE_PLANS_MAX:
        _reshape_plans.popitem(last=False)


def _c_strides(shapes):
    """C-order strides for each row of a (n, rank) array of shapes."""
    shapes = np.asarray(shapes, dtype=np.int64)
    n, r = shapes.shape
    st = np.ones((n, r), dtype=np.int64)
    for a in range(r - 2, -1, -1):
        st[:, a] = st[:, a + 1] * shapes[:, a + 1]
    return st


def _copy_jobs_contiguous(dst_offs, src_offs, sizes):
    n = len(sizes)
    jobs = np.zeros((n, 4 + 3 * COPY_MAXDIM), dtype=np.int64)
    jobs[:, 0] = dst_offs
    jobs[:, 1] = src_offs
    jobs[:, 2] = 1
    jobs[:, 4] = sizes
    jobs[:, 4 + COPY_MAXDIM] = 1
    jobs[:, 4 + 2 * COPY_MAXDIM] = 1
    return jobs


def _run_copy(dtype, jobs, max_elems, src_arena, dst_arena):
    if len(jobs) == 0 or max_elems == 0:
        return
    L = dev.lib()
    for s in range(0, len(jobs), 60000):
        chunk = jobs[s:s + 60000]
        jd = dev.table(chunk)
        dev.check(L.tpa_copy_batch(dev.code(dtype), jd.data_ptr(), len(chunk), int(max_elems), src_arena.data_ptr(),
                                   dst_arena.data_ptr(), dev.stream()), "copy_batch")


def _scatter_blocks(src, layout, dst_arena):
    """Copy the blocks of ``src`` into ``dst_arena`` laid out like ``layout`` (a superset of blocks)."""
    if src.stored_blocks == 0:
        return
    key = {tuple(r): i for i, r in enumerate(layout._qdata)}
    idx = np.array([key[tuple(r)] for r in src._qdata], dtype=np.int64)
    sizes = src._block_sizes_flat()
    jobs = _copy_jobs_contiguous(layout._offsets[idx], src._offsets, sizes)
    _run_copy(layout.dtype, jobs, int(np.max(sizes)), src._arena, dst_arena)


# ======================================================================================================
# creation functions
# ======================================================================================================

def zeros(legcharges, dtype=np.float64, qtotal=None, labels=None):
    return Array(legcharges, dtype, qtotal, labels)


def eye_like(a, axis=0, labels=None):
    """Identity with legs ``(a.legs[axis], a.legs[axis].conj())``."""
    return diag(1., a.get_leg(axis), labels=labels)


def diag(s, leg, dtype=None, labels=None):
    """Diagonal matrix with entries ``s`` (scalar or host 1-D array) on ``(leg, leg.conj())``."""
    s = np.asarray(s, dtype)
    scalar = (s.ndim == 0)
    if not scalar and len(s) != leg.ind_len:
        raise ValueError("len(s)={0:d} not equal to leg.ind_len={1:d}".format(len(s), leg.ind_len))
    res = Array((leg, leg.conj()), s.dtype, labels=labels)
    # equal-charge blocks of a non-bunched leg also couple: only the (q, q) blocks are stored here,
    # which is what the reference does for a blocked leg (np_conserved.py:2984-3024)
    qdata = np.arange(leg.block_number, dtype=np.intp)[:, np.newaxis] * np.ones(2, dtype=np.intp)[np.newaxis, :]
    blocks = []
    for q in range(leg.block_number):
        sl = leg.get_slice(q)
        d = np.full(sl.stop - sl.start, s) if scalar else s[sl]
        blocks.append(np.diag(d.astype(res.dtype)).reshape(-1))
    keep = [i for i, b in enumerate(blocks) if b.size]
    if keep:
        res._set_blocks(qdata[keep], arena=dev.to_device(np.concatenate([blocks[i] for i in keep])), qdata_sorted=True)
    return res


_UNARY_OK = {'conj', 'real', 'imag', 'abs', 'angle', 'sqrt', 'exp', 'log', 'square', 'negative', 'sign', 'sin', 'cos', 'tanh',
             'reciprocal'}


def ones(legcharges, dtype=np.float64, qtotal=None, labels=None):
    """All charge-allowed blocks filled with 1 (reference :2959)."""
    res = Array(legcharges, dtype, qtotal, labels)
    qdata = res._allowed_qdata()
    if len(qdata):
        res._set_blocks(qdata, qdata_sorted=True)
        res._arena.fill_(1.)
    return res


def concatenate(arrays, axis=0, copy=True):
    """Stack Arrays along ``axis`` like ``np.concatenate`` (reference :3027): the sectors of that leg are appended
    without sorting / bunching, so every block keeps its shape and the result arena is the operands' arenas one after
    the other (one contiguous block copy per operand block, a single launch per operand).  ``copy`` is accepted for
    signature compatibility; the device result always owns a new arena."""
    arrays = list(arrays)
    first = arrays[0]
    axis = first.get_leg_index(axis)
    not_axis = [a for a in range(first.rank) if a != axis]
    for a in arrays:
        if a.shape[:axis] != first.shape[:axis] or a.shape[axis + 1:] != first.shape[axis + 1:]:
            raise ValueError("wrong shape to fit " + repr(a.shape) + " into " + repr(first.shape))
        if a.chinfo != first.chinfo:
            raise ValueError("wrong ChargeInfo")
        if np.any(a.qtotal != first.qtotal):
            raise ValueError("wrong qtotal")
        for l in not_axis:
            a.legs[l].test_equal(first.legs[l])
    dtype = _calc_dtype(*[a.dtype for a in arrays])
    axis_qconj = first.legs[axis].qconj
    bl_sizes, charges, qdatas = [], [], []
    shift = 0
    for a in arrays:
        leg = a.legs[axis]
        bl_sizes.extend(leg.get_block_sizes())
        charges.append(leg.charges if leg.qconj == axis_qconj else first.chinfo.make_valid(-leg.charges))
        q = a._qdata.copy()
        q[:, axis] += shift
        qdatas.append(q)
        shift += leg.block_number
    legs = list(first.legs)
    legs[axis] = LegCharge.from_qind(first.chinfo, np.append([0], np.cumsum(bl_sizes)), np.concatenate(charges, axis=0), axis_qconj)
    res = Array(legs, dtype, first.qtotal, list(first._labels))
    res._set_blocks(np.concatenate(qdatas, axis=0), qdata_sorted=False)
    at = 0
    for a in arrays:
        if a.stored_blocks == 0:
            continue
        src = a if a.dtype == dtype else a.astype(dtype)
        sizes = src._block_sizes_flat()
        n = len(sizes)
        jobs = _copy_jobs_contiguous(res._offsets[at:at + n], src._offsets, sizes)
        _run_copy(dtype, jobs, int(np.max(sizes)), src._arena, res._arena)
        at += n
    return res


def expm(a):
    """Matrix exponential of a square block-diagonal matrix (reference :4103, which runs scipy's Pade ``expm`` block by block on the host).

    Device version: scaling and squaring with a Taylor polynomial, all in block GEMMs -- ``X = a / 2**s`` with
    ``||X||_F <= 2``, ``T = sum_{k<=28} X**k / k!`` by Horner's rule (truncation error 2**29 / 29! < 1e-22),
    then ``T <- T T`` s times (few squarings: each one doubles the rounding error that scipy's Pade route does not have).  Sectors without a stored block get the identity, like the reference."""
    if a.rank != 2 or a.shape[0] != a.shape[1]:
        raise ValueError("expect a square matrix!")
    a.legs[0].test_contractible(a.legs[1])
    if np.any(a.qtotal != a.chinfo.make_valid()):
        raise NotImplementedError("A*A has different qtotal than A; nilpotent matrix")
    labels = list(a._labels)
    piped_axes, a = a.as_completely_blocked()
    res_dtype = _calc_dtype(a.dtype)
    eye = diag(1., a.legs[0], dtype=res_dtype)
    nrm = float(norm(a)) if a.stored_blocks else 0.
    if nrm == 0.:
        T = eye
    else:
        if not np.isfinite(nrm):
            raise ValueError("expm of a matrix with non-finite entries")
        s_pow = max(0, int(np.ceil(np.log2(nrm / 2.))))
        X = a.astype(res_dtype, copy=True)
        X.idrop_labels()
        X.iscale_prefactor(0.5**s_pow)
        order = 28
        T = eye.copy(deep=True)
        for k in range(order, 0, -1):
            T = tensordot(X, T, axes=1)
            T.iscale_prefactor(1. / k)
            T.iadd_prefactor_other(1., eye)
        for _ in range(s_pow):
            T = tensordot(T, T, axes=1)
    if len(piped_axes) > 0:
        T = T.split_legs(piped_axes)
    T.iset_leg_labels(labels)
    return T


def pinv(a, cutoff=1.e-15):
    """Moore-Penrose pseudo-inverse through the block SVD (reference :3821)."""
    if cutoff <= 0.:
        raise ValueError("invalid cutoff")
    U, S, VH = svd(a, cutoff=cutoff)
    X = VH.itranspose().iconj().iscale_axis(1. / S, axis=-1)
    Z = U.itranspose().iconj()
    return tensordot(X, Z, axes=1)


def polar(a, cutoff=1.e-16, left=False, inner_labels=[None, None]):
    """Polar decomposition ``a = u p`` (``left=False``) or ``a = p u`` through the block SVD (reference :3762).
    Returns ``(u, p, s)``."""
    if a.rank != 2:
        raise ValueError("Polar is only defined for a 2D matrix. Use LegPipes!")
    if cutoff < 0.:
        raise ValueError("invalid cutoff")
    W, s, VH = svd(a, cutoff=cutoff, inner_labels=inner_labels)
    u = tensordot(W, VH, axes=([1, 0]))
    if not left:
        labels = VH.conj().get_leg_labels()[1], VH.get_leg_labels()[1]
        p = tensordot(VH.conj().itranspose().iscale_axis(s), VH, axes=([1, 0])).iset_leg_labels(labels)
    else:
        labels = u.get_leg_labels()[0], u.conj().get_leg_labels()[0]
        p = tensordot(W.iscale_axis(s), W.conj().itranspose(), axes=([1, 0])).iset_leg_labels(labels)
    return u, p, s


def detect_qtotal(flat_array, legcharges, cutoff=None):
    """Total charge of the entry of largest magnitude of a dense array (reference :3346-3379)."""
    if cutoff is None:
        cutoff = QCUTOFF
    chinfo = legcharges[0].chinfo
    inds = np.unravel_index(np.argmax(np.abs(flat_array)), flat_array.shape)
    if abs(flat_array[inds]) < cutoff:
        warnings.warn("can't detect total charge: no entry larger than cutoff. Return 0 charge.", stacklevel=2)
        return chinfo.make_valid()
    q = chinfo.make_valid()
    for leg, i in zip(legcharges, inds):
        qi, _ = leg.get_qindex(int(i))
        q = q + leg.get_charge(qi)
    return chinfo.make_valid(q)


def to_iterable_arrays(array_list):
    """A single Array / string / scalar becomes a one-element list; other iterables are left alone (reference :4383)."""
    if isinstance(array_list, Array):
        return [array_list]
    return array_list if _is_iterable(array_list) else [array_list]


# ======================================================================================================
# contraction
# ======================================================================================================

class TensordotPlan:
    """A planned block-sparse contraction: host bookkeeping + device-resident task/link/tile tables.

    Built once from the *structure* of the operands (qdata, offsets, leg block sizes) by
    :func:`plan_tensordot`; :meth:`apply` replays it on any pair of arrays with that structure with a
    single kernel launch.  This is how the Lanczos loop avoids all per-matvec host planning.
    """

    def __init__(self):
        self.empty = True
        self.sk = None

    def apply(self, a, b, out_arena=None, launch=True):
        """``launch=False``: only the result's bookkeeping and allocation (callers that replay the plan themselves,
        ``TwoSiteH.matvec_program``)."""
        # legs / labels / qtotal come from the actual operands: the plan only depends on block structure
        legs = [a.legs[x] for x in self.keep_a] + [b.legs[x] for x in self.keep_b]
        la, lb = [a._labels[x] for x in self.keep_a], [b._labels[x] for x in self.keep_b]
        labels = [(l if (l is None or l not in lb) else None) for l in la] + \
            [(l if (l is None or l not in la) else None) for l in lb]
        res = Array(legs, self.dtype, a.chinfo.make_valid(a.qtotal + b.qtotal), labels)
        if self.empty:
            return res
        res._qdata = self.res_qdata
        res._offsets = self.res_offsets
        res._qdata_sorted = True
        res._skey = self.res_skey
        if out_arena is None:
            out_arena = dev.empty(self.res_total, self.dtype)
        res._arena = out_arena
        if not launch:
            return res
        a_arena, b_arena = a._arena, b._arena
        if a.dtype != self.dtype:
            a_arena = a.astype(self.dtype)._arena
        if b.dtype != self.dtype:
            b_arena = b.astype(self.dtype)._arena
        ev = gemm_timer.begin()
        sk = self.sk
        if sk is None:
            dev.check(dev.lib().tpa_gemm_chain(dev.code(self.dtype), self.cfg, self.tasks_dev.data_ptr(), self.links_dev.data_ptr(),
                                               self.tiles_dev.data_ptr(), self.n_tiles, a_arena.data_ptr(),
                                               b_arena.data_ptr(), out_arena.data_ptr(), dev.stream()), "gemm_chain")
        else:       # split-K: partial blocks into a scratch arena, then one deterministic reduction pass (see `_split_k`)
            part = dev.scratch('gemm_split_k', sk.total, self.dtype)
            dev.check(dev.lib().tpa_gemm_chain(dev.code(self.dtype), self.cfg, sk.tasks_dev.data_ptr(), sk.links_dev.data_ptr(),
                                               sk.tiles_dev.data_ptr(), sk.n_tiles, a_arena.data_ptr(),
                                               b_arena.data_ptr(), part.data_ptr(), dev.stream()), "gemm_chain")
            dev.check(dev.lib().tpa_lincomb_batch(dev.code(self.dtype), sk.jobs_dev.data_ptr(), sk.n_jobs, sk.terms_dev.data_ptr(),
                                                  sk.max_elems, part.data_ptr(), out_arena.data_ptr(), dev.stream()), "lincomb_batch")
        gemm_timer.end(ev, self)
        return res


_plan_cache = OrderedDict()
_PLAN_CACHE_SIZE = 8192     # a sweep over L=100 touches ~3000 distinct (bond, contraction) structures: an LRU smaller than that
                            # never hits; the device tables of a plan are ~100 KB (<= 1 GB in total, HBM is 288 GB)


class KernelTimer:
    """Measures the grouped-GEMM launches with HIP events recorded on the stream the kernel is launched
    on (bench.py's roofline figure).  Disabled by default: two event records per launch otherwise."""

    def __init__(self):
        self.enabled = False
        self.reset()

    def reset(self):
        self.pending = []
        self.records = []          # (ms, tag) per timed call when `keep` is set (diagnostic: bench.py TPA_BENCH_SVD_RECORDS)
        self.keep = getattr(self, 'keep', False)
        self.n_launch = 0
        self.flops = 0
        self.bytes_min = 0
        self.ms = 0.

    def sample(self):
        """True for every ``stride``-th request while enabled.  An event record between two kernels costs the device ~5.6 us (a barrier
        packet; profiles/r06_bench_trace_excerpt.txt): four per Lanczos matvec inside the timed region were 1.7 % of the chi = 2048
        sweep.  The GEMM timer therefore samples (``stride`` 8 in bench.py): the rates are averages over the sampled launches."""
        if not self.enabled:
            return False
        self._tick = getattr(self, '_tick', 0) + 1
        return (self._tick - 1) % max(int(getattr(self, 'stride', 1)), 1) == 0

    def begin(self):
        if not self.sample():
            return None
        ev = dev.torch().cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, ev0, plan, tag=None):
        if ev0 is None:
            return
        ev1 = dev.torch().cuda.Event(enable_timing=True)
        ev1.record()
        self.pending.append((ev0, ev1, tag))
        self.n_launch += 1
        self.flops += plan.flops
        self.bytes_min += plan.bytes_min
        if len(self.pending) > 4000:
            self.collect()

    def collect(self):
        """Resolve pending event pairs (synchronises)."""
        if self.pending:
            self.pending[-1][1].synchronize()
            ts = [a.elapsed_time(b) for a, b, _ in self.pending]
            self.ms += sum(ts)
            if self.keep:
                self.records.extend((t, p[2]) for t, p in zip(ts, self.pending))
            self.pending = []
        return self.ms


gemm_timer = KernelTimer()


class _Work:
    """What KernelTimer.end accumulates (algorithmic flops / bytes of one device call)."""
    __slots__ = ('flops', 'bytes_min')

    def __init__(self, flops, bytes_min):
        self.flops, self.bytes_min = flops, bytes_min


# same for every `tpa_svd_batch` call (the dominant kernel family of a DMRG sweep).  Algorithmic work as SURVEY 8(d) defines
# it for an m x n block with k = min(m, n): flops = 4 m^2 n + 8 m n^2 + 9 n^3 for m >= n (gesdd with vectors; x4 complex),
# bytes_min = itemsize (m n + m k + k n + k).
svd_timer = KernelTimer()


# `tpa_eigh_batch` (the `_eig_based_svd` route of the QR-based TEBD, the density-matrix mixer): 9 n^3 flops per Hermitian n x n block
# (tridiagonalisation 4/3 n^3 + vectors; the symmetric share of the gesdd model above), x4 complex; bytes_min = itemsize (2 n^2 + n).
eigh_timer = KernelTimer()


def eigh_work(ns, itemsize, cplx):
    n = np.asarray(ns, dtype=np.float64)
    return _Work(float(np.sum(9. * n ** 3)) * (4. if cplx else 1.), float(itemsize * np.sum(2. * n * n + n)))


def svd_work(ms, ns, itemsize, cplx):
    big, small = np.maximum(ms, ns).astype(np.float64), np.minimum(ms, ns).astype(np.float64)
    flops = np.sum(4. * big * big * small + 8. * big * small * small + 9. * small ** 3) * (4. if cplx else 1.)
    nbytes = itemsize * np.sum(big * small * 3. + small)
    return _Work(float(flops), float(nbytes))


_tile_shapes = {}


def _gemm_tile(dtype, cfg):
    c = (dev.code(dtype), cfg)
    if c not in _tile_shapes:
        bm, bn = dev.c_int(), dev.c_int()
        dev.lib().tpa_gemm_tile_shape(c[0], cfg, dev.byref(bm), dev.byref(bn))
        _tile_shapes[c] = (bm.value, bn.value)
    return _tile_shapes[c]


FORCE_GEMM_CFG = None           # tuning hook
GEMM_MIN_LARGE_TILES = 2048    # below this many large (cfg 0) tiles the 64x64 configuration fills the 256 CUs better
GEMM_LARGE_TILE_PAD = 1.02     # ... and above this ratio of the padded work (large tiles / small tiles) it wastes less arithmetic


def _plan_host(a_qdata, b_qdata, ncontr, contr_nblocks):
    """Call the C++ planner: returns (res_qdata, gemm[(res, ia, ib)])."""
    from .. import _lib
    L = _lib.load()
    na, ra = a_qdata.shape
    nb, rb = b_qdata.shape
    aq = np.ascontiguousarray(a_qdata, dtype=np.int64)
    bq = np.ascontiguousarray(b_qdata, dtype=np.int64)
    cn = np.ascontiguousarray(contr_nblocks, dtype=np.int64)
    rr = (ra - ncontr) + (rb - ncontr)
    cap_res, cap_gemm = max(na * 2, 16), max(na * 4, nb * 4, 64)
    import ctypes
    while True:
        res_q = np.empty((cap_res, max(rr, 1)), dtype=np.int64)
        ra_first = np.empty(cap_res, dtype=np.int64)
        rb_first = np.empty(cap_res, dtype=np.int64)
        gemm = np.empty((cap_gemm, 3), dtype=np.int64)
        n_res, n_gemm = ctypes.c_int64(), ctypes.c_int64()
        rc = L.tpa_plan_tensordot(aq.ctypes.data, na, ra, bq.ctypes.data, nb, rb, ncontr, cn.ctypes.data,
                                  res_q.ctypes.data, ra_first.ctypes.data, rb_first.ctypes.data, cap_res,
                                  ctypes.byref(n_res), gemm.ctypes.data, cap_gemm, ctypes.byref(n_gemm))
        if rc == 0:
            break
        if n_res.value > cap_res or n_gemm.value > cap_gemm:
            cap_res, cap_gemm = max(cap_res, n_res.value), max(cap_gemm, n_gemm.value)
            continue
        _lib.check(rc, "plan_tensordot")
    nr, ng = n_res.value, n_gemm.value
    return res_q[:nr, :rr].reshape(nr, rr), gemm[:ng]


def _normalize_axes(a, b, axes):
    if isinstance(axes, (int, np.integer)):
        n = int(axes)
        axes_a = list(range(a.rank - n, a.rank))
        axes_b = list(range(n))
    else:
        axes_a, axes_b = axes
        axes_a = a.get_leg_indices(_to_iterable(axes_a))
        axes_b = b.get_leg_indices(_to_iterable(axes_b))
    if len(axes_a) != len(axes_b):
        raise ValueError("different lens of axes for a, b: " + repr(axes))
    if len(set(axes_a)) != len(axes_a) or len(set(axes_b)) != len(axes_b):
        raise ValueError("repeated axis")
    return axes_a, axes_b


def _matrix_form(rank, caxes):
    """If the contracted axes ``caxes`` (in contraction order) are the leading or trailing axes of a
    C-contiguous block in ascending order return 'lead' / 'trail', else None."""
    n = len(caxes)
    if n == 0:
        return 'trail'
    if list(caxes) == list(range(rank - n, rank)):
        return 'trail'
    if list(caxes) == list(range(n)):
        return 'lead'
    return None


def plan_tensordot(a, b, axes=2):
    """Plan ``tensordot(a, b, axes)`` for the block structure of ``a`` and ``b``.

    Returns ``(plan, a_use, b_use)`` where ``a_use``/``b_use`` are ``a``/``b`` or physically transposed
    copies if the contracted legs were neither leading nor trailing (in matching order).
    """
    axes_a, axes_b = _normalize_axes(a, b, axes)
    for la, lb in zip(axes_a, axes_b):
        a.legs[la].test_contractible(b.legs[lb])
    if a.chinfo != b.chinfo:
        raise ValueError("Different ChargeInfo")
    nc = len(axes_a)
    # choose an order of the contracted legs so that `a` needs no data movement if possible
    order = np.argsort(axes_a)
    ca = [axes_a[i] for i in order]
    cb = [axes_b[i] for i in order]
    fa = _matrix_form(a.rank, ca)
    a_use, b_use = a, b
    if fa is None:
        keep_a = [x for x in range(a.rank) if x not in axes_a]
        a_use = a.transpose(keep_a + list(axes_a))
        ca = list(range(a.rank - nc, a.rank))
        cb = list(axes_b)
        fa = 'trail'
    fb = _matrix_form(b.rank, cb)
    if fb is None:
        keep_b = [x for x in range(b.rank) if x not in cb]
        b_use = b.transpose(cb + keep_b)
        cb = list(range(nc))
        fb = 'lead'
    key = (a_use._struct_key(), b_use._struct_key(), tuple(ca), tuple(cb), a.dtype.str, b.dtype.str)
    plan = _plan_cache.get(key)
    if plan is not None:
        try:
            _plan_cache.move_to_end(key)
        except KeyError:      # evicted by another thread in between (the reference's `+ h.c.` worker contracts concurrently)
            pass
        return plan, a_use, b_use
    plan = _build_plan(a_use, b_use, ca, cb, fa, fb)
    _plan_cache[key] = plan
    if len(_plan_cache) > _PLAN_CACHE_SIZE:
        _plan_cache.popitem(last=False)
    return plan, a_use, b_use


def _build_plan(a, b, ca, cb, fa, fb):
    nc = len(ca)
    keep_a = [x for x in range(a.rank) if x not in ca]
    keep_b = [x for x in range(b.rank) if x not in cb]
    plan = TensordotPlan()
    plan.dtype = _calc_dtype(a.dtype, b.dtype)
    plan.keep_a, plan.keep_b = keep_a, keep_b
    if len(keep_a) + len(keep_b) == 0:
        raise ValueError("full contraction: use inner()")
    if a.stored_blocks == 0 or b.stored_blocks == 0:
        return plan
    aq = a._qdata[:, keep_a + list(ca)]
    bq = b._qdata[:, list(cb) + keep_b]
    contr_nblocks = [a.legs[x].block_number for x in ca]
    res_q, gemm = _plan_host(aq, bq, nc, contr_nblocks)
    if len(res_q) == 0:
        return plan
    plan.empty = False
    a_shapes, b_shapes = a._block_shapes(), b._block_shapes()
    M = np.prod(a_shapes[:, keep_a], axis=1).astype(np.int64) if keep_a else np.ones(a.stored_blocks, np.int64)
    K = np.prod(a_shapes[:, ca], axis=1).astype(np.int64) if nc else np.ones(a.stored_blocks, np.int64)
    N = np.prod(b_shapes[:, keep_b], axis=1).astype(np.int64) if keep_b else np.ones(b.stored_blocks, np.int64)
    nres = len(res_q)
    gi, ga, gb = gemm[:, 0], gemm[:, 1], gemm[:, 2]
    first = np.concatenate([[0], np.nonzero(np.diff(gi))[0] + 1])
    counts = np.diff(np.concatenate([first, [len(gi)]]))
    m_res, n_res = M[ga[first]], N[gb[first]]
    sizes = m_res * n_res
    offs = np.concatenate([[0], np.cumsum(sizes)])
    plan.res_qdata = np.ascontiguousarray(res_q, dtype=np.intp)
    plan.res_offsets = offs[:-1].astype(np.int64)
    plan.res_total = int(offs[-1])
    tmp = Array([a.legs[x] for x in keep_a] + [b.legs[x] for x in keep_b], plan.dtype)
    tmp._qdata, tmp._offsets = plan.res_qdata, plan.res_offsets
    plan.res_skey = tmp._struct_key()
    # links
    links = np.zeros((len(gi), 8), dtype=np.int64)
    links[:, 0] = a._offsets[ga]
    links[:, 1] = b._offsets[gb]
    links[:, 2] = K[ga]
    if fa == 'trail':
        links[:, 3], links[:, 4] = K[ga], 1
    else:
        links[:, 3], links[:, 4] = 1, M[ga]
    if fb == 'lead':
        links[:, 5], links[:, 6] = N[gb], 1
    else:
        links[:, 5], links[:, 6] = 1, K[ga]
    tasks = np.zeros((nres, 8), dtype=np.int64)
    tasks[:, 0] = plan.res_offsets
    tasks[:, 1], tasks[:, 2], tasks[:, 3] = m_res, n_res, n_res
    tasks[:, 4], tasks[:, 5] = first, counts
    # tiles, heaviest chains first
    ksum = np.add.reduceat(K[ga], first)
    # large tiles only where they fill the device AND the block edges pad (almost) as little as on the small tile: the sector sizes of
    # a DMRG theta (871, 450, 148 ... rows) lose 3 % more on 128 x 64 than on 64 x 64 tiles and are faster on the small one
    pad = {}
    for cfg in (1, 0):
        bm, bn = _gemm_tile(plan.dtype, cfg)
        tm, tn = (m_res + bm - 1) // bm, (n_res + bn - 1) // bn
        ntile = tm * tn
        plan.cfg = cfg
        pad[cfg] = float(np.sum(ntile * ksum)) * bm * bn
    if FORCE_GEMM_CFG != 0 and (int(np.sum(ntile)) < GEMM_MIN_LARGE_TILES or (plan.dtype.kind != 'c' and pad[0] > GEMM_LARGE_TILE_PAD * pad[1])):
        plan.cfg = 1
        bm, bn = _gemm_tile(plan.dtype, 1)
        tm, tn = (m_res + bm - 1) // bm, (n_res + bn - 1) // bn
        ntile = tm * tn
    if FORCE_GEMM_CFG is not None and plan.cfg != FORCE_GEMM_CFG:
        plan.cfg = FORCE_GEMM_CFG
        bm, bn = _gemm_tile(plan.dtype, plan.cfg)
        tm, tn = (m_res + bm - 1) // bm, (n_res + bn - 1) // bn
        ntile = tm * tn
    t_task = np.repeat(np.arange(nres), ntile)
    local = np.arange(int(np.sum(ntile))) - np.repeat(np.cumsum(ntile) - ntile, ntile)
    t_row = local // np.repeat(tn, ntile)
    t_col = local % np.repeat(tn, ntile)
    work = np.repeat(ksum, ntile)
    order = _xcd_tile_order(t_task, t_row, t_col, work, np.repeat(tm, ntile), np.repeat(tn, ntile))
    tiles = np.zeros((len(t_task), 4), dtype=np.int32)
    tiles[:, 0], tiles[:, 1], tiles[:, 2] = t_task[order], t_row[order], t_col[order]
    plan.n_tiles = len(tiles)
    plan.tasks_host, plan.links_host = tasks, links
    plan.tasks_dev, plan.links_dev, plan.tiles_dev = dev.to_device_packed(tasks, links, tiles)
    cmul = 8 if plan.dtype.kind == 'c' else 2
    plan.flops = int(cmul * np.sum(M[ga] * K[ga] * N[gb]))
    esz = 16 if plan.dtype.kind == 'c' else 8
    ua, ub = np.unique(ga), np.unique(gb)
    plan.bytes_min = int(esz * (np.sum(M[ua] * K[ua]) + np.sum(K_of_b(b_shapes, cb, ub) * N[ub]) + plan.res_total))
    plan.n_gemm = len(gi)
    plan.gemm_shapes = np.stack([M[ga], K[ga], N[gb]], axis=1)
    plan.sk = _split_k(plan, tasks, links, tm, tn)
    return plan


# Split-K for launches with few tiles and long chains (round 5).  A launch of T tiles keeps at most T workgroups busy: the 69 tiles of
# matvec step 2 at chi = 512 use a quarter of the 256 CUs, the 831 tiles at chi = 2048 sit on the CUs as 3 or 4 workgroups each (81 %
# balance).  With the knob on, the chain of every C block is cut into up to S parts of whole k-tiles; part p of block t is its own task
# writing a PARTIAL block into a scratch arena, and one `tpa_lincomb_batch` launch adds the parts up into C in a fixed order
# (deterministic: no atomics).  No kernel change -- the cut links are ordinary links with shifted offsets.
# Measured on MI355X (profiles/r05_gemm_split_k.txt): GEMM fraction of the f64 MFMA peak 0.452 -> 0.461 (Heisenberg chi = 2048, step 2 of
# the matvec: 930 tiles -> 3572), 0.279 -> 0.324 (Hubbard chi = 1024), 0.105 -> 0.132 (XXZ chi = 512); sweeps 1.2 % / 2.2 % / ~2 % shorter.
# Launches that already have more than ~4 workgroups per CU lose (chi = 1024 step 1, 1336 tiles: 0.130 -> 0.145 ms per matvec when split).
GEMM_SPLIT_K = tuple(int(x) for x in os.environ.get('TPA_GEMM_SPLIT_K', '4,4096,128,1024').split(','))     # (max parts, target tiles, min K per part, max tiles of a launch that is split); '0' = off
GEMM_K_TILE = 16


class _SplitK:
    __slots__ = ('tasks_dev', 'links_dev', 'tiles_dev', 'n_tiles', 'total', 'jobs_dev', 'terms_dev', 'n_jobs', 'max_elems',
                 'tasks_host', 'links_host', 'jobs_host', 'terms_host', 'parts')


def _split_k(plan, tasks, links, tm, tn, knob=None):
    knob = tuple(knob or GEMM_SPLIT_K)
    s_max, target, min_k, max_tiles = knob + (4096, 128, 1024)[len(knob) - 1:]
    if s_max < 2 or plan.n_tiles * 2 > target or plan.n_tiles > max_tiles or len(tasks) > 65535:
        return None
    s_plan = min(s_max, target // plan.n_tiles)
    nt = len(tasks)
    first, counts = tasks[:, 4], tasks[:, 5]
    assert np.array_equal(first, np.cumsum(counts) - counts) and first[-1] + counts[-1] == len(links)      # chains stored back to back
    K = links[:, 2]
    g_end = np.cumsum(K)                    # the contracted indices of all chains laid end to end: link i covers [g_start, g_end)
    g_start = g_end - K
    t_start, t_end = g_start[first], g_end[first + counts - 1]
    ktot = t_end - t_start
    s_t = np.clip(ktot // max(min_k, 1), 1, s_plan)
    if np.all(s_t == 1):
        return None
    # nominal cuts ktot j / s, snapped to a whole k-tile of the link they fall into (or to that link's end)
    n_cut = s_t - 1
    ct = np.repeat(np.arange(nt), n_cut)
    cj = np.arange(int(np.sum(n_cut))) - np.repeat(np.cumsum(n_cut) - n_cut, n_cut) + 1
    pos = t_start[ct] + ktot[ct] * cj // s_t[ct]
    li = np.searchsorted(g_start, pos, side='right') - 1
    pos = g_start[li] + np.minimum((pos - g_start[li] + GEMM_K_TILE // 2) // GEMM_K_TILE * GEMM_K_TILE, K[li])
    # boundaries of the parts, task by task: start, cuts, end; parts of length zero (cuts snapped onto each other) are dropped
    bt = np.concatenate([np.arange(nt), ct, np.arange(nt)])
    bp = np.concatenate([t_start, pos, t_end])
    order = np.lexsort((bp, bt))
    bt, bp = bt[order], bp[order]
    keep = (bt[:-1] == bt[1:]) & (bp[1:] > bp[:-1])
    p_task, p0, p1 = bt[:-1][keep], bp[:-1][keep], bp[1:][keep]
    n_new = len(p_task)
    if n_new == nt:
        return None
    l0 = np.searchsorted(g_start, p0, side='right') - 1
    l1 = np.searchsorted(g_start, p1, side='left')
    nl = l1 - l0
    lb = np.cumsum(nl) - nl
    idx = np.repeat(l0, nl) + np.arange(int(np.sum(nl))) - np.repeat(lb, nl)
    new_links = links[idx].copy()
    c0 = np.maximum(np.repeat(p0, nl), g_start[idx]) - g_start[idx]
    c1 = np.minimum(np.repeat(p1, nl), g_end[idx]) - g_start[idx]
    new_links[:, 0] += c0 * new_links[:, 4]
    new_links[:, 1] += c0 * new_links[:, 5]
    new_links[:, 2] = c1 - c0
    m, n = tasks[p_task, 1], tasks[p_task, 2]
    offs = np.cumsum(m * n) - m * n
    new_tasks = np.zeros((n_new, 8), dtype=np.int64)
    new_tasks[:, 0], new_tasks[:, 1], new_tasks[:, 2], new_tasks[:, 3], new_tasks[:, 4], new_tasks[:, 5] = offs, m, n, n, lb, nl
    terms = np.zeros((n_new, 4), dtype=np.int64)
    terms[:, 0], terms[:, 1], terms[:, 2] = offs, n, np.array(1.0).view(np.int64)
    parts = np.bincount(p_task, minlength=nt)
    jobs = np.zeros((nt, 8), dtype=np.int64)
    jobs[:, :4] = tasks[:, :4]
    jobs[:, 4], jobs[:, 5] = np.cumsum(parts) - parts, parts
    works = p1 - p0
    off = int(np.sum(m * n))
    sk = _SplitK()
    sk.parts = parts
    ntile = np.repeat(tm * tn, sk.parts)
    t_task = np.repeat(np.arange(n_new), ntile)
    local = np.arange(int(np.sum(ntile))) - np.repeat(np.cumsum(ntile) - ntile, ntile)
    tn_new, tm_new = np.repeat(tn, sk.parts), np.repeat(tm, sk.parts)
    t_row, t_col = local // np.repeat(tn_new, ntile), local % np.repeat(tn_new, ntile)
    order = _xcd_tile_order(t_task, t_row, t_col, np.repeat(works, ntile), np.repeat(tm_new, ntile),
                            np.repeat(tn_new, ntile))
    tiles = np.zeros((len(t_task), 4), dtype=np.int32)
    tiles[:, 0], tiles[:, 1], tiles[:, 2] = t_task[order], t_row[order], t_col[order]
    sk.tasks_host, sk.links_host, sk.jobs_host, sk.terms_host = new_tasks, new_links, jobs, terms
    sk.tasks_dev, sk.links_dev, sk.tiles_dev, sk.jobs_dev, sk.terms_dev = dev.to_device_packed(sk.tasks_host, sk.links_host, tiles, sk.jobs_host,
                                                                                            sk.terms_host)
    sk.n_tiles, sk.total, sk.n_jobs = len(tiles), off, len(jobs)
    sk.max_elems = int(np.max(tasks[:, 1] * tasks[:, 2]))
    return sk


N_XCD = 8
XCD_TILE_ORDER = True     # tuning hook


def _xcd_tile_order(t_task, t_row, t_col, work, tm, tn):
    """Order of the tile table.  Workgroup b runs on XCD b % 8 (observed dispatch rule; used for speed only)
    and every XCD has a private L2, so each XCD gets a compact band of tile ROWS of every C block: the A row
    panels of a band are then fetched by one XCD only, and neighbouring column tiles (which share the A panel)
    are resident on that XCD at the same time.  Inside an XCD the heaviest chains go first (LPT)."""
    n = len(t_task)
    if not XCD_TILE_ORDER or n < 2 * N_XCD:
        return np.argsort(-work, kind='stable')
    xcd = np.where(tm >= N_XCD, (t_row * N_XCD) // np.maximum(tm, 1),
                   ((t_row * tn + t_col) * N_XCD) // np.maximum(tm * tn, 1)).astype(np.int64)
    # tasks with fewer than 8 tiles would all land on the low XCDs: rotate by task index
    xcd = (xcd + t_task) % N_XCD
    key = np.lexsort((t_col, t_row, t_task, -work, xcd))          # by xcd, then heavy first, then locality
    lists = [key[xcd[key] == x] for x in range(N_XCD)]
    # balance the list lengths (the table is consumed round-robin): move tail tiles of long lists to short ones
    target = -(-n // N_XCD)
    spill = []
    for x in range(N_XCD):
        if len(lists[x]) > target:
            spill.extend(lists[x][target:])
            lists[x] = lists[x][:target]
    for x in range(N_XCD):
        need = target - len(lists[x])
        if need > 0 and spill:
            lists[x] = np.concatenate([lists[x], np.array(spill[:need], dtype=np.int64)])
            spill = spill[need:]
    out = np.full(target * N_XCD, -1, dtype=np.int64)
    for x in range(N_XCD):
        out[x:x + N_XCD * len(lists[x]):N_XCD] = lists[x]
    out = out[out >= 0]
    assert len(out) == n
    return out


def K_of_b(b_shapes, cb, idx):
    return np.prod(b_shapes[idx][:, cb], axis=1) if len(cb) else np.ones(len(idx), np.int64)


def tensordot(a, b, axes=2):
    """Block-sparse ``np.tensordot`` (reference np_conserved.py:3612).  Result ``_qdata`` is lexsorted."""
    axes_a, axes_b = _normalize_axes(a, b, axes)
    if len(axes_a) == a.rank and len(axes_b) == b.rank:
        # full contraction -> scalar
        return inner(a, b, axes=(axes_a, axes_b), do_conj=False)
    plan, a_use, b_use = plan_tensordot(a, b, (axes_a, axes_b))
    return plan.apply(a_use, b_use)


def outer(a, b):
    """Outer product ``res[i.., j..] = a[i..] * b[j..]`` (reference :3494) -- a K=1 grouped GEMM."""
    return tensordot(a, b, axes=0)


def inner(a, b, axes='labels', do_conj=False):
    """Contract all legs of ``a`` with all legs of ``b`` -> scalar (reference :3540, worker :4614).

    ``axes='labels'``: same labels (``do_conj=True``) or conjugated labels (``do_conj=False``);
    ``axes='range'``: leg i with leg i; or explicit ``(axes_a, axes_b)``.
    """
    if isinstance(a, list) and isinstance(b, list):
        return np.sum([inner(w, v, axes=axes, do_conj=do_conj) for w, v in zip(a, b)])
    if a.rank != b.rank:
        raise ValueError("different rank!")
    if not (isinstance(axes, str) and axes == 'range'):
        if isinstance(axes, str) and axes == 'labels':
            a_labels = a.get_leg_labels()
            axes = (a_labels, a_labels) if do_conj else (a_labels, [Array._conj_leg_label(l) for l in a_labels])
        axes_a, axes_b = axes
        axes_a = a.get_leg_indices(_to_iterable(axes_a))
        axes_b = b.get_leg_indices(_to_iterable(axes_b))
        if len(axes_a) != a.rank or len(axes_b) != b.rank:
            raise ValueError("no full contraction. Use tensordot instead!")
        order = np.argsort(axes_b)
        axes_a = [axes_a[i] for i in order]
        if tuple(axes_a) != tuple(range(a.rank)):
            a = a.transpose(axes_a)
    if a.chinfo != b.chinfo:
        raise ValueError("different ChargeInfo")
    for lega, legb in zip(a.legs, b.legs):
        if do_conj:
            lega.test_equal(legb)
        else:
            lega.test_contractible(legb)
    # charge check: result non-zero only if total charges compensate
    if do_conj:
        if np.any(a.qtotal != b.qtotal):
            return _calc_dtype(a.dtype, b.dtype).type(0)
    else:
        if np.any(a.chinfo.make_valid(a.qtotal + b.qtotal) != 0):
            return _calc_dtype(a.dtype, b.dtype).type(0)
    calc = _calc_dtype(a.dtype, b.dtype)
    if a.stored_blocks == 0 or b.stored_blocks == 0:
        return calc.type(0)
    if a.dtype != calc:
        a = a.astype(calc)
    if b.dtype != calc:
        b = b.astype(calc)
    out, scr = dev.reduction_buffers()
    L = dev.lib()
    if a._same_structure(b) and a._is_packed():
        xa, xb, n = a._arena, b._arena, a._arena.numel()
    else:
        # gather the common blocks of both into two packed temporaries
        keyb = {tuple(r): i for i, r in enumerate(b._qdata)}
        ia = [i for i, r in enumerate(a._qdata) if tuple(r) in keyb]
        if len(ia) == 0:
            return calc.type(0)
        ib = [keyb[tuple(a._qdata[i])] for i in ia]
        sizes = a._block_sizes_flat()[ia]
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        n = int(np.sum(sizes))
        xa, xb = dev.empty(n, calc), dev.empty(n, calc)
        _run_copy(calc, _copy_jobs_contiguous(offs, a._offsets[ia], sizes), int(np.max(sizes)), a._arena, xa)
        _run_copy(calc, _copy_jobs_contiguous(offs, b._offsets[ib], sizes), int(np.max(sizes)), b._arena, xb)
    dev.check(L.tpa_dot(dev.code(calc), n, xa.data_ptr(), xb.data_ptr(), int(bool(do_conj)), out.data_ptr(),
                        scr.data_ptr(), dev.stream()), "dot")
    val = dev.read_scalar(out, calc.kind == 'c')
    return calc.type(val)


def norm(a, ord=None, convert_to_float=True):
    if isinstance(a, Array):
        return a.norm(ord, convert_to_float)
    return np.linalg.norm(np.asarray(a).reshape(-1), ord)


def trace(a, leg1=0, leg2=1):
    """Trace over two legs (reference :3441): contraction with an identity on the paired legs."""
    ax1, ax2 = a.get_leg_indices([leg1, leg2])
    a.legs[ax1].test_contractible(a.legs[ax2])
    eye = diag(1., a.legs[ax2], dtype=a.dtype)   # legs (l2, l2*), l2* == l1-compatible
    if a.rank == 2:
        return inner(a, eye, axes=([ax1, ax2], [0, 1]), do_conj=False)
    return tensordot(a, eye, axes=([ax1, ax2], [0, 1]))


# ======================================================================================================
# decompositions
# ======================================================================================================

# Absolute floor rho of the Jacobi stopping rule (include/tenpy_amd.h, tpa_svd_batch `tol`): row pairs whose larger norm is
# below rho*||block||_F are judged against rho*||block||_F instead of their own norm, i.e. the part of the spectrum below
# rho*||A|| is not rotated against itself to the last bit (measured on the chi = 2048 theta: rho = 1e-6 / 1e-8 / 0 ->
# 6 / 7 / 8 sweeps).  Singular VALUES keep absolute accuracy eps*||A|| either way; the singular VECTORS of sigma < rho*||A||
# come out mutually orthogonal only to eps*sqrt(L)*rho*||A||/sigma (5e-7 for rho = 1e-6).  Round 3: those vectors -- and only
# those -- are re-orthonormalised afterwards by two first-order Loewdin steps V <- (3 - V V^H) V / 2 on the matrix cores
# (`_svd_warm.lowdin_rows`; changes U S VH by <= eps*rho*||A||), so that every returned vector is orthonormal to machine
# precision like LAPACK's (VERDICT r2 "What's weak") at the sweep count of the floor.  Set SVD_ABS_FLOOR = 0. for
# the purely relative Hestenes criterion (no clean-up needed).
# Round 5: 1e-6 -> 1e-2, together with a CORRECTED predicted-convergence rule (csrc/tpa_svd.hip::svd_big_rotation).
# (a) What the floor trades is Jacobi's RELATIVE accuracy of the vectors of tiny singular values for the ABSOLUTE accuracy class that
#     LAPACK's gesdd (the reference's svd_robust) has anyway: a pair below the floor stops at |cos| <= eps sqrt(L) rho |A| / sigma, while
#     gesdd's vectors carry angle errors ~ eps |A| / gap, i.e. ~ 10 eps |A| / sigma for gaps of sigma / 10 -- at rho = 1e-2 the rule is
#     still ~30x stricter than that.  Singular VALUES are second order in the remaining cosines and keep eps |A| either way.
# (b) The rule of rounds 2-4 let the iteration END while pairs below the floor still had cosines of O(0.1) (its "big rotation" test
#     was scaled by the floor, so such pairs never counted).  On pivoted-QR starts that is harmless; on warm / sketch starts of blocks
#     graded down to rounding level tests/test_svd_configs_gpu.py measured isometry defects up to 3e-3 AFTER the clean-up.  With the
#     corrected rule every pair is predicted against its own stopping rule; the sweeps the old rule "saved" are back:
#     driver protocol (3 + 4 sweeps, profiles/r05_abs_floor.txt), s per sweep / Jacobi sweeps per warm, sketch, cold call:
#       old rule:  rho 1e-6  2.78 / 4.89 3.96 5.77      rho 1e-4  2.66 / 4.35 3.44 5.27      (round 4's 3.01 s was measured with this rule)
#       new rule:  rho 1e-6  3.01 / 6.42 4.81 6.36      rho 1e-4  2.87 / 5.63 4.49 5.92      rho 1e-3  2.80 / 5.19 4.23 5.84
#                  rho 1e-2  2.74 / 4.93 4.09 5.79
#     Singular values (1.7 - 2.7e-15 of sigma_max), sweep energies (3 - 8e-15 against TeNPy's), matvec / Lanczos parity, the isometry of
#     the sampled decompositions (1.5e-13) and of the MPS tensors the sweeps stored (7e-15) are the same in every row of the new rule.
SVD_ABS_FLOOR = float(os.environ.get('TPA_SVD_ABS_FLOOR', '1e-2'))      # (the environment variable: measurement knob)
# Round 6: the floor acts on the SMALLER row of a pair (include/tenpy_amd.h, tpa_svd_batch `tol` < 0) and the clean-up of the vectors
# below it is ORDERED (every vector against the ones of larger singular value, `_svd_warm.ordered_rows`) instead of symmetric: a pair
# of rows of very different size may then stop at a cosine of eps rho |A| / sigma_min instead of eps rho |A| / sigma_max, because the
# clean-up no longer pushes half of the remaining cosine into the LARGE vector.  Same accuracy class (reconstruction 1 - 3e-16 |A|,
# singular values 2 - 4e-16 sigma_max on the Jacobi inputs of the chi = 2048 sweeps, profiles/r06_stopping_rule_emulation.txt), a third
# of the block pairs active.  TPA_SVD_FLOOR_ON_MIN=0 restores the rule of rounds 1 - 5 (the ordered clean-up is valid for both).
SVD_FLOOR_ON_MIN = os.environ.get('TPA_SVD_FLOOR_ON_MIN', '1') != '0'


def _svd_tol_arg():
    """The `tol` argument of tpa_svd_batch / tpa_eigh_batch for the call in progress (sign = which row the floor acts on)."""
    f = _svd_floor_now[0]
    return -f if (SVD_FLOOR_ON_MIN and f > 0.) else f
# Round 4 (VERDICT r3 task 7): the floor is an opt-in of the callers that can afford it -- the DMRG / TEBD drivers, which
# truncate right afterwards and mark their call (``svd_hint`` of the engines, or ``svd_engine_floor = True`` for one call).  Every
# other ``npc.svd`` (an unmodified TeNPy module calling it for its own purposes) runs the purely relative criterion: every returned
# vector converged, no post-processing.  ``TENPY_AMD_SVD_FLOOR`` overrides the floor of such generic calls.
SVD_ABS_FLOOR_GENERIC = float(os.environ.get('TENPY_AMD_SVD_FLOOR', '0'))
svd_engine_floor = False


class _PerThreadFloor(threading.local):
    """Floor of the ``npc.svd`` call in progress, per thread (``DMRGThreadPlusHC`` and ``tests/test_threads.py`` run ``svd`` from a
    second thread).  Indexed like the one-element list it replaces."""

    def __init__(self):
        self.v = SVD_ABS_FLOOR_GENERIC

    def __getitem__(self, i):
        return self.v

    def __setitem__(self, i, value):
        self.v = value


_svd_floor_now = _PerThreadFloor()      # read by the helpers below
# Round 6: with the floor on the smaller row the cosines left among the tiniest rows are larger and MANY (2-norm of the cosine matrix
# 0.6 - 0.8 on the Jacobi inputs of the chi = 2048 sweeps against 0.25 under the old rule): the ordered clean-up needs 6 instead of 4 - 5
# iterations to reach 1e-15 (emulation on dumped inputs: 1e-1, 2e-2, 1e-3, 8e-6, 4e-10, 9e-16).
SVD_LOWDIN_ITERATIONS = int(os.environ.get('TPA_SVD_LOWDIN_ITERATIONS', '6' if os.environ.get('TPA_SVD_FLOOR_ON_MIN', '1') != '0' else '4'))      # (round 5: 2 -> 4 with the higher floor: the tiniest kept rows stop at cosines ~7e-3, and first-order Loewdin squares the defect per iteration)
# Warm start (`_svd_warm`): a caller that knows which bond it is decomposing sets ``svd_hint = (key, side)`` right before
# ``svd`` / ``svd_theta`` (side 'R': the right singular vectors of the previous decomposition under ``key`` are a good basis,
# 'L': the left ones); the hint is consumed by the next call.  Without a hint, or when the cached basis does not fit the block
# structure / misses too much of the matrix, the cold path (rank-revealing QR + Jacobi) runs.  Results are exact either way.
SVD_WARM = os.environ.get('TPA_SVD_WARM', '1') != '0'
svd_hint = None
SVD_PROFILE = bool(os.environ.get('TPA_SVD_PROFILE'))      # diagnostic: synchronise around the stages of every svd() and time them


def _svd_tick(name, t0=None):
    import time
    dev.torch().cuda.synchronize()
    now = time.time()
    if name is not None:
        _svd_warm.stats[name] = _svd_warm.stats.get(name, 0.) + (now - t0)
    return now


svd_stats = {'calls': 0, 'sweeps': 0, 'max_block': 0}


def _blocked_matrix_jobs(a):
    """Per stored block of a rank-2 Array: (offset, m, n)."""
    sh = a._block_shapes()
    return a._offsets.astype(np.int64), sh[:, 0].astype(np.int64), sh[:, 1].astype(np.int64)


SVD_DIST_GROUP = None     # (torch.distributed group, rank, world) while a multi-GPU engine is active: charge blocks are independent


def svd_block_owners(ms, ns, world):
    """Longest-processing-time assignment of the charge blocks to ``world`` ranks (cost ~ m n min(m, n)); identical on all ranks."""
    cost = ms.astype(np.float64) * ns * np.minimum(ms, ns)
    owner = np.zeros(len(ms), dtype=np.int64)
    load = np.zeros(world)
    for b in np.argsort(-cost, kind='stable'):
        r = int(np.argmin(load))
        owner[b] = r
        load[r] += cost[b]
    return owner


def _svd_distributed(L, code, a, jobs, ms, ns, ks, U_arena, S_dev, V_arena, sweeps):
    """Every rank decomposes its share of the (independent) charge blocks, then ONE all-gather of the packed
    [U blocks | S | VH blocks] of each rank makes the three result arenas complete and bit-identical everywhere
    ("SVD: blocks distributed by LPT ... gather U/S/VH panels afterwards")."""
    import torch
    import torch.distributed as dist
    group, rank, world = SVD_DIST_GROUP
    owner = svd_block_owners(ms, ns, world)
    mine = np.nonzero(owner == rank)[0]
    failed, err = 0., None
    flag = dev.zeros(1, np.float64)       # (allocated BEFORE the try: an allocation failure must not skip the collective below)
    if len(mine):
        lj = np.ascontiguousarray(jobs[mine])
        ls_off = np.concatenate([[0], np.cumsum(ks[mine])])
        try:      # same fallback chain as the single-GPU path; a failure is agreed on below, BEFORE the collective
            S_loc = dev.empty(int(ls_off[-1]), np.float64)
            lj2 = lj.copy()
            lj2[:, 4] = ls_off[:-1]
            _svd_batch_robust(L, code, lj2, len(mine), a._arena, U_arena, S_loc, V_arena, sweeps)
            for t, b in enumerate(mine):
                S_dev[int(jobs[b, 4]):int(jobs[b, 4]) + int(ks[b])].copy_(S_loc[int(ls_off[t]):int(ls_off[t + 1])])
        except Exception as e:        # ANY failure (LinAlgError, ValueError, a HIP error, out of memory) is agreed on before the gather:
            failed, err = 1., e       # a rank that raised alone would leave the others waiting in the collective forever
    flag.fill_(failed)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if float(flag.item()) > 0.:       # every rank raises, none is left waiting in the all-gather
        raise err if err is not None else np.linalg.LinAlgError("tenpy_amd svd: the block SVD failed on another rank")
    cplx = 2 if np.dtype(a.dtype).kind == 'c' else 1
    Uf = U_arena.view(torch.float64) if cplx == 2 else U_arena     # interleaved (re, im)
    Vf = V_arena.view(torch.float64) if cplx == 2 else V_arena
    # packed layout per rank: for each of its blocks (in block order)  U block, S block, VH block  as float64
    sizes = cplx * (ms * ks + ks * ns) + ks
    per_rank = [int(np.sum(sizes[owner == r])) for r in range(world)]
    maxlen = max(max(per_rank), 1)
    send = dev.zeros(maxlen, np.float64)

    def pieces(b):
        return ((Uf, cplx * int(jobs[b, 3]), cplx * int(ms[b] * ks[b])), (S_dev, int(jobs[b, 4]), int(ks[b])),
                (Vf, cplx * int(jobs[b, 5]), cplx * int(ks[b] * ns[b])))
    at = 0
    for b in mine:
        for buf, off, n in pieces(b):
            send[at:at + n].copy_(buf[off:off + n])
            at += n
    recv = dev.empty(maxlen * world, np.float64)
    dist.all_gather_into_tensor(recv, send, group=group)
    for r in range(world):
        if r == rank:
            continue
        at = r * maxlen
        for b in np.nonzero(owner == r)[0]:
            for buf, off, n in pieces(b):
                buf[off:off + n].copy_(recv[at:at + n])
                at += n
    sw = dev.zeros(1, np.float64)
    sw.fill_(float(sweeps.value))
    dist.all_reduce(sw, op=dist.ReduceOp.MAX, group=group)
    sweeps.value = int(sw.item())


# Algorithm chain of the block SVD, the device counterpart of svd_robust.py:65-75 (gesdd, on LinAlgError gesvd) and of the
# NaN re-try of np_conserved.py:4970-4982: every entry is a `tpa_svd_set_algorithm` code that is tried when the previous
# one returned TPA_E_NOCONV or produced NaNs: default (pivoted-QR preconditioner + fused block Jacobi), block Jacobi
# without the preconditioner and with two-kernel rounds, then the plain one-wavefront-per-row-pair Jacobi.
# Round 4: the default (code 0) runs Gram-only sweeps on 32-row blocks (csrc/tpa_svd_b32.inc).  (The end game by simultaneous
# rotations of round 4, TPA_SVD_REFINE, was removed in round 5: slower over a whole sweep, never on by default.)
SVD_ALG0 = int(os.environ.get('TPA_SVD_ALG0', '0'))             # measurement knob: extra bits for the head of the chain (e.g. 1048576 = no Gram-only sweeps)
SVD_ALGORITHM_CHAIN = (SVD_ALG0, 512 | 2, 1 | 512)
# warm-started and sketch calls (`_svd_warm`): no pivoted QR (bit 9)
SVD_ALGORITHM_CHAIN_WARM = (512 | SVD_ALG0, 512 | 2, 1 | 512)
SVD_MAX_SWEEPS = 80
svd_robust_stats = {'retries': 0, 'last_chain': ()}
_svd_worksize_cache = {}
_svd_alg_initialised = False


def _svd_batch_robust(L, code, jobs, nblk, a_arena, U_arena, S_dev, V_arena, sweeps, chain=None):
    """One batched device SVD with the fallback chain; returns the singular values on the host."""
    global _svd_alg_initialised
    if not _svd_alg_initialised:          # the library starts in state 0; a knob may have changed the head of the chain
        _svd_alg_initialised = True
        if SVD_ALGORITHM_CHAIN[0] != 0:
            L.tpa_svd_set_algorithm(SVD_ALGORITHM_CHAIN[0])
    chain = SVD_ALGORITHM_CHAIN if chain is None else chain
    wkey = (code, jobs.tobytes())
    wb = _svd_worksize_cache.get(wkey)
    if wb is None:              # (the layout of the work area: ~30 us of host time per call for what depends on the block sizes only)
        if len(_svd_worksize_cache) > 4096:
            _svd_worksize_cache.clear()
        wb = _svd_worksize_cache[wkey] = L.tpa_svd_worksize(code, jobs.ctypes.data, nblk)
    work = dev.scratch('svd_work', int(wb), np.uint8)
    if _svd_warm.PROFILE:
        _svd_warm._tick('t_svd_worksize_scratch')
    tried = []
    last_err = None
    for hop, alg in enumerate(chain):
        if hop == 0 and alg != SVD_ALGORITHM_CHAIN[0]:
            L.tpa_svd_set_algorithm(alg)
        if hop:
            warnings.warn("tenpy_amd: block SVD (algorithm %s) gave %s. Try again with algorithm %d"
                          % (tried[-1], last_err, alg), stacklevel=3)
            svd_robust_stats['retries'] += 1
            L.tpa_svd_set_algorithm(alg)
        tried.append(alg)
        try:
            rc = L.tpa_svd_batch(code, jobs.ctypes.data, nblk, a_arena.data_ptr(), U_arena.data_ptr(), S_dev.data_ptr(),
                                 V_arena.data_ptr(), work.data_ptr(), int(wb), SVD_MAX_SWEEPS, _svd_tol_arg(),
                                 dev.byref(sweeps), dev.stream())
        finally:
            if hop or alg != SVD_ALGORITHM_CHAIN[0]:
                L.tpa_svd_set_algorithm(SVD_ALGORITHM_CHAIN[0])
            if _svd_warm.PROFILE:
                _svd_warm._tick('t_svd_batch_call')
        svd_robust_stats['last_chain'] = tuple(tried)
        if rc == dev.E_NOCONV:
            last_err = "no convergence"
            continue
        dev.check(rc, "svd_batch")          # bad arguments, NaN / Inf in the INPUT, HIP errors: no second try helps
        S_host = dev.to_host(S_dev)
        if np.any(np.isnan(S_host)):
            last_err = "NaNs"
            continue
        return S_host
    if last_err == "NaNs":
        raise ValueError("NaN in S: " + str(int(np.sum(np.isnan(S_host)))))
    raise np.linalg.LinAlgError("tenpy_amd svd_batch: no convergence with any of the algorithms " + repr(tuple(tried)))


from . import _svd_warm  # noqa: E402


def _svd_sig_counts(S_host, ks, s_offs, rel):
    """Per block: number of singular values above ``rel * |S_b|_2`` (they are sorted descending inside a block)."""
    ks = np.asarray(ks, dtype=np.int64)
    nb = len(ks)
    if nb == 0:
        return np.zeros(0, dtype=np.int64)
    starts = np.asarray(s_offs[:nb], dtype=np.int64)
    if np.all(ks > 0) and np.array_equal(starts[1:], starts[:-1] + ks[:-1]) and starts[-1] + ks[-1] <= len(S_host):
        seg = S_host[starts[0]:starts[-1] + ks[-1]]        # the blocks lie back to back: one segmented reduction
        rel_starts = starts - starts[0]
        fro = np.sqrt(np.add.reduceat(seg * seg, rel_starts))
        above = seg > np.repeat(rel * fro, ks)
        return np.where(fro > 0., np.add.reduceat(above.astype(np.int64), rel_starts), 0)
    out = np.zeros(nb, dtype=np.int64)
    for b in range(nb):
        sb = S_host[s_offs[b]:s_offs[b] + ks[b]]
        fro = float(np.sqrt(np.sum(sb * sb)))
        out[b] = int(np.sum(sb > rel * fro)) if fro > 0. else 0
    return out


def _svd_y_side_cold(ms, ns, cplx):
    """Which factor of the cold device SVD holds the NORMALISED rows of the Jacobi iteration (the other one is a product of
    plane rotations / Householder reflections and orthonormal by construction): True -> VH, False -> U.  Mirrors
    ``tpa_svd_batch`` (csrc/tpa_svd.hip): with the rank-revealing QR (blocks >= 32, the default chain entry) X = A (m >= n) or
    A^T is factorised and the rows of R are orthogonalised; without it the rows of A (m < n) or of A^T."""
    if SVD_ALGORITHM_CHAIN[0] & (1 | 512):      # head of the chain without the pivoted QR (measurement knobs): the predicate below does
        return None                              # not describe that call -- both factors get the clean-up
    rmax_pad = int(np.max((np.minimum(ms, ns) + 1) // 2 * 2))
    dim_max = int(max(np.max(ms), np.max(ns)))
    if rmax_pad >= 32 and dim_max <= (2048 if cplx else 8192):
        return ms >= ns
    return ms < ns


def _svd_clean_small(dtype, U_arena, V_arena, S_host, ms, ns, ks, u_offs, s_offs, v_offs, y_is_vh=None):
    """Loewdin re-orthonormalisation of the singular vectors whose singular value lies below the absolute floor of the
    stopping rule (see ``SVD_ABS_FLOOR``): columns ``[k0, k1)`` of ``U_b`` or rows ``[k0, k1)`` of ``VH_b``, ``k0`` = number of
    values >= floor, ``k1`` = number of non-negligible values (zero-padded vectors beyond the numerical rank are left alone).
    ``y_is_vh[b]`` says which factor holds the normalised Jacobi rows (only that one can be off); None: both are treated."""
    k0 = _svd_sig_counts(S_host, ks, s_offs, _svd_floor_now[0])
    k1 = _svd_sig_counts(S_host, ks, s_offs, 1.e-15)
    if not np.any(k1 - k0 > 0):
        return
    # Round 6: ORDERED clean-up over ALL significant vectors [0, k1) -- with the floor on the smaller row a vector below the floor may
    # keep a cosine of eps rho |A| / sigma_small with a vector ABOVE it, which only the ordered step removes without touching the large one
    # (the vectors above the floor are mutually converged to eps sqrt(L): the step leaves them alone to rounding).
    # (rounded up to whole 32-row groups, at most the block's rank: the count of significant values moves by a few between two
    #  visits of a bond, and every new count would be a new plan -- tables built and uploaded with the device idle, 170 us per call in
    #  profiles/r06_bench_trace_excerpt.txt; vectors beyond the count are unit rows of rounding noise or exact zeros, harmless here)
    n_all = np.where(k1 - k0 > 0, np.minimum((k1 + 31) // 32 * 32, ks), 0)
    one = np.ones(len(ks), dtype=np.int64)
    nv = n_all if y_is_vh is None else np.where(y_is_vh, n_all, 0)
    nu = n_all if y_is_vh is None else np.where(y_is_vh, 0, n_all)
    _svd_warm.ordered_rows(dtype, V_arena, v_offs[:-1], nv, ns, ns, one, iterations=SVD_LOWDIN_ITERATIONS)
    _svd_warm.ordered_rows(dtype, U_arena, u_offs[:-1], nu, ms, one, ks, iterations=SVD_LOWDIN_ITERATIONS)


def _leg_sector_keys(leg, qinds):
    return [tuple(int(x) for x in leg.charges[q]) + (int(leg.slices[q + 1] - leg.slices[q]),) for q in qinds]


def _svd_warm_store(a, key, U_arena, V_arena, S_host, ms, ns, ks, u_offs, s_offs, v_offs):
    """Remember the singular vectors of this decomposition as warm-start bases: rows of VH (side 'R') and columns of U
    (side 'L', stored transposed), the significant ones only, in arenas owned by the cache."""
    ksig = _svd_sig_counts(S_host, ks, s_offs, _svd_warm.E_RANK_TOL)
    if np.any(ksig <= 0):
        return
    ms, ns, ks = (np.ascontiguousarray(x, dtype=np.int64) for x in (ms, ns, ks))
    L = dev.lib()
    if SVD_THETA_NATIVE and a.dtype.kind != 'c' and hasattr(L, 'tpa_svd_theta_store'):
        r_off = np.concatenate([[0], np.cumsum(ksig * ns)])
        l_off = np.concatenate([[0], np.cumsum(ksig * ms)])
        sect = a.__dict__.get('_tpa_sector_keys')
        if sect is None or sect[0] is not a._qdata:
            sect = a.__dict__['_tpa_sector_keys'] = (a._qdata, _leg_sector_keys(a.legs[0], a._qdata[:, 0]), _leg_sector_keys(a.legs[1], a._qdata[:, 1]))
        Rb, Lb = dev.empty(int(r_off[-1]), a.dtype), dev.empty(int(l_off[-1]), a.dtype)
        zero = np.zeros(len(ms), dtype=np.int64)
        blocks = np.ascontiguousarray(np.stack([zero, ms, ns, np.asarray(u_offs[:-1]), zero, np.asarray(v_offs[:-1]), zero, zero], axis=1), dtype=np.int64)
        kc = np.ascontiguousarray(ksig, dtype=np.int64)
        dev.check(L.tpa_svd_theta_store(dev.code(a.dtype), blocks.ctypes.data, kc.ctypes.data, len(ms), U_arena.data_ptr(), V_arena.data_ptr(),
                                        Rb.data_ptr(), Lb.data_ptr(), dev.stream()), "svd_theta_store")
        _svd_warm.cache_put(key, 'R', _svd_warm.Basis(Rb, r_off[:-1].copy(), kc.copy(), ns, sect[2], a.dtype))
        _svd_warm.cache_put(key, 'L', _svd_warm.Basis(Lb, l_off[:-1].copy(), kc.copy(), ms, sect[1], a.dtype))
        return
    pkey = _svd_warm._key('store', ksig, ms, ns, ks, np.ascontiguousarray(u_offs, dtype=np.int64), np.ascontiguousarray(v_offs, dtype=np.int64))
    pl = _svd_warm._plan_get(pkey)
    if pl is None:
        one = np.ones(len(ks), dtype=np.int64)
        r_off = np.concatenate([[0], np.cumsum(ksig * ns)])     # side R: k x n row-major = the first rows of VH_b
        l_off = np.concatenate([[0], np.cumsum(ksig * ms)])     # side L: k x m row-major = U_b^T (the rows span the row space of theta^T)
        pl = _svd_warm._plan_put(pkey, dict(
            r_off=r_off[:-1].copy(), nR=int(r_off[-1]), l_off=l_off[:-1].copy(), nL=int(l_off[-1]), ksig=ksig.copy(),
            copy_R=_svd_warm.copy_table(_svd_warm.copy_jobs_2d(r_off[:-1], ns, one, v_offs[:-1], ns, one, ksig, ns)),
            copy_L=_svd_warm.copy_table(_svd_warm.copy_jobs_2d(l_off[:-1], ms, one, u_offs[:-1], one, ks, ksig, ms))))
    sect = a.__dict__.get('_tpa_sector_keys')
    if sect is None or sect[0] is not a._qdata:
        sect = a.__dict__['_tpa_sector_keys'] = (a._qdata, _leg_sector_keys(a.legs[0], a._qdata[:, 0]), _leg_sector_keys(a.legs[1], a._qdata[:, 1]))
    Rb = dev.empty(pl['nR'], a.dtype)
    _svd_warm.run_copy(a.dtype, pl['copy_R'], V_arena, Rb)
    _svd_warm.cache_put(key, 'R', _svd_warm.Basis(Rb, pl['r_off'], pl['ksig'], ns, sect[2], a.dtype))
    Lb = dev.empty(pl['nL'], a.dtype)
    _svd_warm.run_copy(a.dtype, pl['copy_L'], U_arena, Lb)
    _svd_warm.cache_put(key, 'L', _svd_warm.Basis(Lb, pl['l_off'], pl['ksig'], ms, sect[1], a.dtype))


# A warm-started call may leave some charge blocks to the cold path (no basis, stale basis, full-rank edge blocks).  They are
# decomposed by a second, ordinary batch if they are a small part of the work; otherwise the whole call takes the cold path.
# Measured on the chi = 2048 theta (round 3): the second batch is a second dependent chain of Jacobi rounds whose length is set by
# the ROWS of its largest block (~3 ms for blocks of ~150 rows), so mixed calls lose; default: all or nothing.
SVD_WARM_MAX_COLD_FRACTION = 0.0
SVD_THETA_NATIVE = os.environ.get('TPA_SVD_THETA_NATIVE', '1') != '0'      # real data: the warm route and the basis store as native calls (tpa_svd_theta)
# Skipping 1-3 visits after a stale attempt was right while a stale attempt cost 9.3 ms (round 3 / early round 4: it decomposed the blocks
# that had passed before it gave up); at 0.5 ms per failed attempt against ~7 ms saved by a hit it only throws hits away.  Measured at the
# end of round 4 (5 + 6 sweeps at chi = 2048, profiles/r04_bench_heis2048_no_cooldown_5_6.json): 65 % instead of 43 % of the calls go
# warm, 2.91 s per sweep (with the cool-down: 3.01 on 5 + 20, 3.05 on 3 + 3 sweeps).  Off by default; TPA_SVD_WARM_COOLDOWN=1 restores it.
SVD_WARM_COOLDOWN = os.environ.get('TPA_SVD_WARM_COOLDOWN', '0') != '0'


def _svd_warm_try(L, code, a, hint, jobs, offs, ms, ns, ks, u_offs, s_offs, v_offs, sweeps):
    """Warm-started decomposition if the cache holds a basis for ``hint``; None -> cold path for the whole call."""
    key, side = hint
    wait = _svd_warm.cooldown.get(key, 0) if SVD_WARM_COOLDOWN else 0
    if wait > 0:                          # the last attempt under this key found a stale basis: theta is still changing
        _svd_warm.cooldown[key] = wait - 1
        _svd_warm.stats['skipped'] = _svd_warm.stats.get('skipped', 0) + 1
        return None
    basis = _svd_warm.cache_get(key, side)
    if basis is None or basis.dtype != a.dtype:
        return None
    sect = a.__dict__.get('_tpa_sector_keys')
    if sect is None or sect[0] is not a._qdata:
        sect = a.__dict__['_tpa_sector_keys'] = (a._qdata, _leg_sector_keys(a.legs[0], a._qdata[:, 0]), _leg_sector_keys(a.legs[1], a._qdata[:, 1]))
    want = sect[2] if side == 'R' else sect[1]
    have = {k: i for i, k in enumerate(basis.sectors)}
    order = np.array([have.get(k, -1) for k in want], dtype=np.int64)
    found = order >= 0
    if not np.any(found):
        _svd_warm.stats['fb_nomatch'] += 1
        return None                       # the leg changed (different sectors / sizes): no basis for any block
    nblk = len(ms)
    b_off = np.where(found, np.asarray(basis.off)[order], 0)
    b_k = np.where(found, np.asarray(basis.k)[order], 0)
    b_len = np.where(found, np.asarray(basis.length)[order], 0)
    weight = ms.astype(np.float64) * ns * ks
    if np.sum(weight[~found]) > SVD_WARM_MAX_COLD_FRACTION * np.sum(weight):
        _svd_warm.stats['fb_nomatch'] += 1
        return None
    total_sweeps = [0]
    age = _svd_warm.ages.get(key, 0) + 1
    native_stale = False
    U_arena = V_arena = None
    if SVD_THETA_NATIVE and a.dtype.kind != 'c' and np.all(found) and hasattr(L, 'tpa_svd_theta') \
            and np.all((b_k > 0) & (b_k <= ks) & (b_len == (ns if side == 'R' else ms))) \
            and np.array_equal(offs, np.concatenate([[0], np.cumsum(ms * ns)[:-1]])) and int(np.sum(ms * ns)) == int(a._arena.numel()):
        # the whole warm route as ONE native call (csrc/tpa_svd_theta.hip; include/tenpy_amd.h): tables built in C++, one staged upload
        # per stage, residual test, Jacobi, accumulated basis, result copies, singular values, ordered clean-up
        U_arena, V_arena = dev.empty(int(u_offs[-1]), a.dtype), dev.empty(int(v_offs[-1]), a.dtype)
        blocks = np.ascontiguousarray(np.stack([offs, ms, ns, u_offs[:-1], s_offs[:-1], v_offs[:-1], b_off, b_k], axis=1), dtype=np.int64)
        S_host = np.zeros(int(s_offs[-1]), dtype=np.float64)
        info = np.zeros(4, dtype=np.float64)
        clean_it = SVD_LOWDIN_ITERATIONS if _svd_floor_now[0] > 0. else 0
        rc = L.tpa_svd_theta(code, 0 if side == 'R' else 1, blocks.ctypes.data, nblk, int(a._arena.numel()), a._arena.data_ptr(),
                             basis.arena.data_ptr(), U_arena.data_ptr(), int(u_offs[-1]), V_arena.data_ptr(), int(v_offs[-1]),
                             S_host.ctypes.data, _svd_warm.E_TOL, int(age % 8 == 0), clean_it, float(_svd_floor_now[0]),
                             SVD_ALGORITHM_CHAIN_WARM[0], SVD_ALGORITHM_CHAIN[0], SVD_MAX_SWEEPS, _svd_tol_arg(), dev.byref(sweeps),
                             info.ctypes.data, dev.stream())
        _svd_warm.stats['native_calls'] = _svd_warm.stats.get('native_calls', 0) + 1
        if rc == 0:
            _svd_warm.ages[key] = age
            _svd_warm.stats['warm_calls'] += 1
            _svd_warm.stats['warm_sweeps'] += sweeps.value
            _svd_warm.stats['e_rel_last'] = float(info[0])
            _svd_warm.stats['e_rel_max'] = max(_svd_warm.stats['e_rel_max'], float(info[0]))
            _svd_warm.last_kind = 'warm'
            return U_arena, None, V_arena, S_host
        if rc == 1:            # stale basis: on to the sketch / cold route below (the result arenas come back cleared)
            _svd_warm.stats['fb_stale'] += int(info[1])
            _svd_warm.stats['e_rel_last'] = float(info[0])
            _svd_warm.stats['e_rel_max'] = max(_svd_warm.stats['e_rel_max'], float(info[0]))
            native_stale = True
        elif rc in (dev.E_NOCONV, getattr(dev, 'E_NAN', -3)):
            _svd_warm.stats['fb_svd'] += nblk
            _svd_warm.stats['fallbacks'] += 1
            return None
        else:
            dev.check(rc, "svd_theta")
    if U_arena is None:
        U_arena = dev.zeros(int(u_offs[-1]), a.dtype)
        V_arena = dev.zeros(int(v_offs[-1]), a.dtype)

    def run_svd(j, arena, U, S, VH, qrp):
        sw = dev.c_int()
        j = np.ascontiguousarray(j)
        try:
            S_h = _svd_batch_robust(L, code, j, len(j), arena, U, S, VH, sw, chain=None if qrp else SVD_ALGORITHM_CHAIN_WARM)
        except (np.linalg.LinAlgError, ValueError):
            return None
        total_sweeps[0] += sw.value
        return S_h

    # the accumulated basis U'^H Bc drifts from orthonormality by ~eps per warm generation: one Loewdin step every 8th keeps it there
    if native_stale:
        done, S_blocks = np.zeros(nblk, dtype=bool), [None] * nblk
    else:
        done, S_blocks = _svd_warm.svd_blocks_warm(a.dtype, a._arena, offs, ms, ns, basis.arena, b_off, b_k, b_len, side, run_svd,
                                                   (U_arena, V_arena, u_offs[:-1], v_offs[:-1]),
                                                   lowdin_basis=(age % 8 == 0), need_all=(SVD_WARM_MAX_COLD_FRACTION <= 0.))
    cold = np.nonzero(~done)[0]
    sk_wait = _svd_warm.sketch_cooldown.get(key, 0)
    if sk_wait > 0:
        _svd_warm.sketch_cooldown[key] = sk_wait - 1
    if len(cold) and _svd_warm.SKETCH and np.all(found) and not np.any(done) and sk_wait == 0 \
            and _svd_warm.stats.get('e_rel_last', 1.) <= _svd_warm.SKETCH_MAX_E:
        # stale basis (the state moved since the bond's previous visit): it still is an excellent SKETCH of the column space --
        # range finder + unpivoted QR + Jacobi on the small factor instead of the pivoted QR of the cold path (round 5, _svd_warm.py)
        S_blocks = _svd_warm.svd_blocks_sketch(a.dtype, a._arena, offs, ms, ns, basis.arena, b_off, b_k, b_len, side, run_svd,
                                               (U_arena, V_arena, u_offs[:-1], v_offs[:-1]))
        if S_blocks is not None:
            S_host = np.concatenate(S_blocks) if len(S_blocks) else np.zeros(0)
            S_dev = dev.to_device(S_host)
            sweeps.value = total_sweeps[0]
            _svd_warm.ages[key] = 0
            _svd_warm.stats['sketch_calls'] = _svd_warm.stats.get('sketch_calls', 0) + 1
            _svd_warm.stats['sketch_sweeps'] = _svd_warm.stats.get('sketch_sweeps', 0) + total_sweeps[0]
            _svd_warm.last_kind = 'sketch'
            if _svd_floor_now[0] > 0. and SVD_LOWDIN_ITERATIONS > 0:
                # the normalised Jacobi rows VH' are VH (side 'R') or the columns of U (side 'L')
                _svd_clean_small(a.dtype, U_arena, V_arena, S_host, ms, ns, ks, u_offs, s_offs, v_offs, np.full(nblk, side == 'R'))
            return U_arena, S_dev, V_arena, S_host
        # the rank outgrew basis + extra rows (the bond is still growing): the next visits of this bond do not try
        _svd_warm.sketch_cooldown[key] = _svd_warm.SKETCH_COOLDOWN
        total_sweeps[0] = 0
    if len(cold) and np.sum(weight[cold]) > SVD_WARM_MAX_COLD_FRACTION * np.sum(weight):
        _svd_warm.stats['fallbacks'] += 1
        # try again after a few visits: the residual of a converging state shrinks by roughly a decade per sweep
        e = _svd_warm.stats.get('e_rel_last', 1.)
        _svd_warm.cooldown[key] = int(min(3, max(0, np.ceil(np.log10(max(e, 1e-300) / _svd_warm.E_TOL) / 2.5) - 1)))
        return None
    S_host = np.zeros(int(s_offs[-1]), dtype=np.float64)
    for b in np.nonzero(done)[0]:
        S_host[s_offs[b]:s_offs[b + 1]] = S_blocks[b]
    if len(cold):
        # the remaining (small) blocks: ordinary batch into the same result arenas
        _svd_warm.stats['mixed_calls'] += 1
        cj = np.ascontiguousarray(jobs[cold])
        cs_off = np.concatenate([[0], np.cumsum(ks[cold])])
        cj[:, 4] = cs_off[:-1]
        S_c = dev.empty(int(cs_off[-1]), np.float64)
        sw = dev.c_int()
        S_ch = _svd_batch_robust(L, code, cj, len(cold), a._arena, U_arena, S_c, V_arena, sw)
        total_sweeps[0] += sw.value
        for t, b in enumerate(cold):
            S_host[s_offs[b]:s_offs[b + 1]] = S_ch[cs_off[t]:cs_off[t + 1]]
    S_dev = dev.to_device(S_host)
    sweeps.value = total_sweeps[0]
    _svd_warm.ages[key] = age
    _svd_warm.stats['warm_calls'] += 1
    _svd_warm.stats['warm_sweeps'] += total_sweeps[0]
    _svd_warm.last_kind = 'warm'
    if _svd_floor_now[0] > 0. and SVD_LOWDIN_ITERATIONS > 0:
        # the normalised Jacobi rows VH' end up in U (side 'R') or VH (side 'L'); mixed calls: treat both factors
        y_vh = None if len(cold) else np.full(nblk, side == 'L')
        _svd_clean_small(a.dtype, U_arena, V_arena, S_host, ms, ns, ks, u_offs, s_offs, v_offs, y_vh)
    return U_arena, S_dev, V_arena, S_host


def _copy_jobs_2d(dst_off, dst_ld, src_off, src_ld, rows, cols, src_transposed=False, conj=False):
    """Strided-copy jobs ``dst[r, c] = src[r, c]`` (or ``src[c, r]``, optionally conjugated) for row-major sub-matrices."""
    n = len(rows)
    jobs = np.zeros((n, 4 + 3 * COPY_MAXDIM), dtype=np.int64)
    jobs[:, 0], jobs[:, 1], jobs[:, 2], jobs[:, 3] = dst_off, src_off, 2, int(bool(conj))
    jobs[:, 4], jobs[:, 5] = rows, cols
    jobs[:, 4 + COPY_MAXDIM], jobs[:, 5 + COPY_MAXDIM] = dst_ld, 1
    if src_transposed:
        jobs[:, 4 + 2 * COPY_MAXDIM], jobs[:, 5 + 2 * COPY_MAXDIM] = 1, src_ld
    else:
        jobs[:, 4 + 2 * COPY_MAXDIM], jobs[:, 5 + 2 * COPY_MAXDIM] = src_ld, 1
    return jobs


def _complement_blocks(Q):
    """Orthonormal completion of the isometry ``Q`` (legs ``[left, inner]``, blocked): a device Array ``C`` with legs
    ``[left, c]`` whose blocks span the orthogonal complement of the column space of ``Q`` inside every sector of
    ``left``; ``None`` if ``Q`` is already unitary.  See ``_npc_cold.complement_columns`` (SVD of ``1 - Q Q^dagger``)."""
    from . import _npc_cold
    return _npc_cold.complement_columns(Q)


def _svd_result_arrays(a, qtotal_L, qtotal_R, U_arena, V_arena, u_offs, v_offs, s_offs, inner_qconj):
    """``U`` (legs [a.legs[0], new inner leg]) and ``VH`` ([new inner leg, a.legs[1]]) around the result arenas of a block SVD of the
    completely blocked matrix ``a``: the new inner leg has one sector per block, in the order of the blocks (reference :3744-3754)."""
    chinfo = a.chinfo
    nblk = a.stored_blocks
    qi_L, qi_R = a._qdata[:, 0], a._qdata[:, 1]
    new_charges = chinfo.make_valid((qtotal_R - a.legs[1].get_charge(qi_R)) * inner_qconj)
    new_leg_R = LegCharge.from_qind(chinfo, s_offs, new_charges, inner_qconj)
    new_leg_L = new_leg_R.conj()
    qi_C = np.arange(nblk, dtype=np.intp)
    U = Array([a.legs[0], new_leg_L], a.dtype, qtotal_L)
    VH = Array([new_leg_R, a.legs[1]], a.dtype, qtotal_R)
    U._qdata = np.ascontiguousarray(np.stack([qi_L, qi_C], axis=1), dtype=np.intp)
    U._offsets = u_offs[:-1].astype(np.int64)
    U._arena = U_arena
    U._qdata_sorted = a._qdata_sorted
    VH._qdata = np.ascontiguousarray(np.stack([qi_C, qi_R], axis=1), dtype=np.intp)
    VH._offsets = v_offs[:-1].astype(np.int64)
    VH._arena = V_arena
    VH._qdata_sorted = a._qdata_sorted
    return U, VH


def _svd_qtotals(a, qtotal_LR):
    qtotal_L, qtotal_R = qtotal_LR
    if qtotal_L is None and qtotal_R is None:
        qtotal_R = a.qtotal
    if qtotal_L is None:
        qtotal_L = a.chinfo.make_valid(a.qtotal - qtotal_R)
    elif qtotal_R is None:
        qtotal_R = a.chinfo.make_valid(a.qtotal - qtotal_L)
    elif np.any(a.qtotal != a.chinfo.make_valid(np.asarray(qtotal_L) + np.asarray(qtotal_R))):
        raise ValueError("The entries of `qtotal_LR` have to add up to ``a.qtotal``!")
    return a.chinfo.make_valid(qtotal_L), a.chinfo.make_valid(qtotal_R)


def svd_batched(arrays, qtotal_LRs=None, inner_labels=[None, None], inner_qconj=+1):
    """``[svd(a, qtotal_LR=q, inner_labels=inner_labels) for a, q in zip(arrays, qtotal_LRs)]`` for INDEPENDENT matrices -- the
    two-site wave functions of all even (or odd) bonds of a Trotter half-step, reference ``algorithms/tebd.py:374-414`` -- as ONE
    batched device call over the charge blocks of all of them (VERDICT r3 task 3).  The block SVD is a chain of dependent rounds that
    keeps a few CUs busy per matrix; independent matrices share the launches of that chain instead of queueing behind each other.
    Per matrix the result is bit-identical to :func:`svd` on the cold path (the blocks never interact)."""
    n = len(arrays)
    if qtotal_LRs is None:
        qtotal_LRs = [[None, None]] * n
    if n == 0:
        return []
    global svd_hint, svd_engine_floor
    svd_hint = None
    _svd_floor_now[0] = SVD_ABS_FLOOR if svd_engine_floor else SVD_ABS_FLOOR_GENERIC
    svd_engine_floor = False
    dtype = arrays[0].dtype
    preps = []
    a_base = u_base = v_base = s_base = 0
    for a, qLR in zip(arrays, qtotal_LRs):
        if a.rank != 2:
            raise ValueError("SVD is only defined for a 2D matrix. Use LegPipes!")
        if a.dtype != dtype:
            raise ValueError("svd_batched: mixed dtypes")
        a_labels = a._labels
        piped_axes, ab = a.as_completely_blocked()
        qL, qR = _svd_qtotals(ab, qLR)
        if ab.stored_blocks == 0:
            raise RuntimeError("SVD found no singular values")
        if not ab._is_packed():
            ab = ab.copy(deep=True)._repack()
        offs, ms, ns = _blocked_matrix_jobs(ab)
        ks = np.minimum(ms, ns)
        u_offs = np.concatenate([[0], np.cumsum(ms * ks)])
        v_offs = np.concatenate([[0], np.cumsum(ks * ns)])
        s_offs = np.concatenate([[0], np.cumsum(ks)])
        preps.append(dict(a=ab, labels=a_labels, piped=piped_axes, qL=qL, qR=qR, offs=offs, ms=ms, ns=ns, ks=ks, u_offs=u_offs,
                          v_offs=v_offs, s_offs=s_offs, a_base=a_base, u_base=u_base, v_base=v_base, s_base=s_base,
                          n_a=int(ab._arena.numel())))
        a_base += int(ab._arena.numel())
        u_base += int(u_offs[-1])
        v_base += int(v_offs[-1])
        s_base += int(s_offs[-1])
    big = dev.scratch('svd_batched_in', a_base, dtype)
    jobs = []
    for pr in preps:
        big[pr['a_base']:pr['a_base'] + pr['n_a']].copy_(pr['a']._arena)
        j = np.zeros((len(pr['ms']), 8), dtype=np.int64)
        j[:, 0], j[:, 1], j[:, 2] = pr['offs'] + pr['a_base'], pr['ms'], pr['ns']
        j[:, 3], j[:, 4], j[:, 5] = pr['u_offs'][:-1] + pr['u_base'], pr['s_offs'][:-1] + pr['s_base'], pr['v_offs'][:-1] + pr['v_base']
        jobs.append(j)
    jobs = np.ascontiguousarray(np.concatenate(jobs))
    U_big, V_big = dev.empty(u_base, dtype), dev.empty(v_base, dtype)
    S_dev = dev.empty(s_base, np.float64)
    L, code = dev.lib(), dev.code(dtype)
    sweeps = dev.c_int()
    ms_all, ns_all = jobs[:, 1].copy(), jobs[:, 2].copy()
    ev = svd_timer.begin()
    S_host = _svd_batch_robust(L, code, jobs, len(jobs), big, U_big, S_dev, V_big, sweeps)
    if _svd_floor_now[0] > 0. and SVD_LOWDIN_ITERATIONS > 0:
        ks_all = np.minimum(ms_all, ns_all)
        first_try = len(svd_robust_stats['last_chain']) <= 1
        y_vh = _svd_y_side_cold(ms_all, ns_all, np.dtype(dtype).kind == 'c') if first_try else None
        _svd_clean_small(dtype, U_big, V_big, S_host, ms_all, ns_all, ks_all, np.concatenate([jobs[:, 3], [u_base]]),
                         np.concatenate([jobs[:, 4], [s_base]]), np.concatenate([jobs[:, 5], [v_base]]), y_vh)
    if ev is not None:
        svd_timer.end(ev, svd_work(ms_all, ns_all, np.dtype(dtype).itemsize, np.dtype(dtype).kind == 'c'), ('batched', sweeps.value, int(np.max(np.minimum(ms_all, ns_all)))))
    svd_stats['calls'] += n
    svd_stats['sweeps'] += sweeps.value * n
    svd_stats['max_block'] = max(svd_stats['max_block'], int(np.max(np.minimum(ms_all, ns_all))))
    _svd_warm.stats['cold_calls'] += n
    _svd_warm.stats['cold_sweeps'] += sweeps.value * n
    out = []
    labL, labR = inner_labels
    for pr in preps:
        ab = pr['a']
        Ua = U_big[pr['u_base']:pr['u_base'] + int(pr['u_offs'][-1])]
        Va = V_big[pr['v_base']:pr['v_base'] + int(pr['v_offs'][-1])]
        U, VH = _svd_result_arrays(ab, pr['qL'], pr['qR'], Ua, Va, pr['u_offs'], pr['v_offs'], pr['s_offs'], inner_qconj)
        S = S_host[pr['s_base']:pr['s_base'] + int(pr['s_offs'][-1])].copy()
        if 0 in pr['piped']:
            U = U.split_legs(0)
        if 1 in pr['piped']:
            VH = VH.split_legs(1)
        U.iset_leg_labels([pr['labels'][0], labL])
        VH.iset_leg_labels([labR, pr['labels'][1]])
        out.append((U, S, VH))
    return out


def svd(a, full_matrices=False, compute_uv=True, cutoff=None, qtotal_LR=[None, None], inner_labels=[None, None],
        inner_qconj=+1):
    """Block-wise SVD ``a = U diag(S) VH`` (reference np_conserved.py:3676, worker :4950).

    All charge blocks are decomposed in one batched device call (rank-revealing QR + one-sided block Jacobi, with
    the fallback chain ``SVD_ALGORITHM_CHAIN``); ``S`` is returned on the host (1-D ndarray, concatenated block by
    block, not globally sorted) because the truncation decision (``truncation.truncate``) is host logic.
    ``full_matrices=True``: ``U`` and ``VH`` are completed to square unitary blocks on the device.
    """
    # the one-call marks of the engines are consumed FIRST: a call that raises in the validation below must not leave them
    # behind for the next, unrelated npc.svd
    global svd_hint, svd_engine_floor
    hint, svd_hint = svd_hint, None
    _svd_floor_now[0] = SVD_ABS_FLOOR if (hint is not None or svd_engine_floor) else SVD_ABS_FLOOR_GENERIC
    svd_engine_floor = False
    if a.rank != 2:
        raise ValueError("SVD is only defined for a 2D matrix. Use LegPipes!")
    labL, labR = inner_labels
    a_labels = a._labels
    piped_axes, a = a.as_completely_blocked()
    qtotal_L, qtotal_R = qtotal_LR
    if qtotal_L is None and qtotal_R is None:
        qtotal_R = a.qtotal
    if qtotal_L is None:
        qtotal_L = a.chinfo.make_valid(a.qtotal - qtotal_R)
    elif qtotal_R is None:
        qtotal_R = a.chinfo.make_valid(a.qtotal - qtotal_L)
    elif np.any(a.qtotal != a.chinfo.make_valid(np.asarray(qtotal_L) + np.asarray(qtotal_R))):
        raise ValueError("The entries of `qtotal_LR` have to add up to ``a.qtotal``!")
    qtotal_L, qtotal_R = a.chinfo.make_valid(qtotal_L), a.chinfo.make_valid(qtotal_R)
    if a.stored_blocks == 0:
        raise RuntimeError("SVD found no singular values")
    if full_matrices and cutoff is not None:
        raise ValueError("full_matrices=True with a cutoff is not defined")     # the reference asserts (:4995)
    offs, ms, ns = _blocked_matrix_jobs(a)
    ks = np.minimum(ms, ns)
    nblk = len(ms)
    u_offs = np.concatenate([[0], np.cumsum(ms * ks)])
    v_offs = np.concatenate([[0], np.cumsum(ks * ns)])
    s_offs = np.concatenate([[0], np.cumsum(ks)])
    jobs = np.zeros((nblk, 8), dtype=np.int64)
    jobs[:, 0], jobs[:, 1], jobs[:, 2] = offs, ms, ns
    jobs[:, 3], jobs[:, 4], jobs[:, 5] = u_offs[:-1], s_offs[:-1], v_offs[:-1]
    L = dev.lib()
    code = dev.code(a.dtype)
    sweeps = dev.c_int()
    warm = None
    tick = _svd_tick if SVD_PROFILE else (lambda name, t0=None: None)
    t0 = tick(None)
    # ONE timer entry per npc.svd (bench.py `roofline.avg_launch_ms`): the failed warm attempt of a call that then goes the cold way,
    # the Loewdin clean-up and the copy of the bases into the warm-start cache are inside it (the round-3 entries left
    # them out and counted a failed attempt as a launch of its own)
    ev_svd = svd_timer.begin()
    tried_warm = False
    # (with a distribution group the warm attempt runs REPLICATED on every rank: its kernels are deterministic, so the decision
    # "stale or not" and the results are bit-identical everywhere and need no exchange; only the cold path deals the blocks out)
    if hint is not None and SVD_WARM and not full_matrices:
        tried_warm = True
        warm = _svd_warm_try(L, code, a, hint, jobs, offs, ms, ns, ks, u_offs, s_offs, v_offs, sweeps)
        t0 = tick('t_warm_ok' if warm is not None else 't_warm_failed', t0)
    if warm is not None:
        U_arena, S_dev, V_arena, S_host = warm
    else:
        U_arena = dev.empty(int(u_offs[-1]), a.dtype)
        V_arena = dev.empty(int(v_offs[-1]), a.dtype)
        S_dev = dev.empty(int(s_offs[-1]), np.float64)
        if SVD_DIST_GROUP is not None and nblk > 1:
            wb = L.tpa_svd_worksize(code, jobs.ctypes.data, nblk)
            _svd_distributed(L, code, a, jobs, ms, ns, ks, U_arena, S_dev, V_arena, sweeps)
            S_host = dev.to_host(S_dev)
            if np.any(np.isnan(S_host)):
                raise ValueError("NaN in S: " + str(np.sum(np.isnan(S_host))))
        else:
            S_host = _svd_batch_robust(L, code, jobs, nblk, a._arena, U_arena, S_dev, V_arena, sweeps)
        _svd_warm.stats['cold_calls'] += 1
        _svd_warm.stats['cold_sweeps'] += sweeps.value
        if hint is not None:
            _svd_warm.ages[hint[0]] = 0
        t0 = tick('t_cold', t0)
        if compute_uv and _svd_floor_now[0] > 0. and SVD_LOWDIN_ITERATIONS > 0:
            first_try = len(svd_robust_stats['last_chain']) <= 1 and SVD_DIST_GROUP is None
            y_vh = _svd_y_side_cold(ms, ns, a.dtype.kind == 'c') if first_try else None
            _svd_clean_small(a.dtype, U_arena, V_arena, S_host, ms, ns, ks, u_offs, s_offs, v_offs, y_vh)
            t0 = tick('t_clean', t0)
    if hint is not None and SVD_WARM and compute_uv:
        _svd_warm_store(a, hint[0], U_arena, V_arena, S_host, ms, ns, ks, u_offs, s_offs, v_offs)
        t0 = tick('t_store', t0)
    if ev_svd is not None:
        svd_timer.end(ev_svd, svd_work(ms, ns, a.dtype.itemsize, a.dtype.kind == 'c'),
                      (_svd_warm.last_kind if warm is not None else ('cold after a stale warm attempt' if tried_warm else 'cold'), sweeps.value, int(np.max(ks))))
    svd_stats['calls'] += 1
    svd_stats['sweeps'] += sweeps.value
    svd_stats['max_block'] = max(svd_stats['max_block'], int(np.max(ks)))
    if not compute_uv:
        if cutoff is not None:
            S_host = S_host[S_host > cutoff]
        if len(S_host) == 0:
            raise RuntimeError("SVD found no singular values")
        return S_host
    U, VH = _svd_result_arrays(a, qtotal_L, qtotal_R, U_arena, V_arena, u_offs, v_offs, s_offs, inner_qconj)
    S = S_host
    if full_matrices:
        U, VH = _svd_full_matrices(a, U, VH, ms, ns, ks, qtotal_L, qtotal_R)
    elif cutoff is not None:
        keep = S > cutoff
        if not np.any(keep):
            raise RuntimeError("SVD found no singular values")
        if not np.all(keep):
            U.iproject(keep, 1)
            VH.iproject(keep, 0)
            S = S[keep]
    if 0 in piped_axes:
        U = U.split_legs(0)
    if 1 in piped_axes:
        VH = VH.split_legs(1)
    U.iset_leg_labels([a_labels[0], labL])
    VH.iset_leg_labels([labR, a_labels[1]])
    return U, S, VH


def _widen_isometry(thin, ms, ks, sectors, leg, transposed_rows=False):
    """Square unitary blocks ``[thin | complement]`` for the sectors ``sectors`` (qindices of ``leg``) of an isometry
    ``thin`` = Array ``[leg, inner]`` whose b-th block is ``ms[b] x ks[b]``.  Returns ``(arena, offsets)`` of the
    ``ms[b] x ms[b]`` blocks; with ``transposed_rows`` the blocks are stored as the conjugate transpose (rows = vectors)."""
    comp = _complement_blocks(thin) if np.any(ms > ks) else None
    nblk = len(ms)
    offs = np.concatenate([[0], np.cumsum(ms * ms)]).astype(np.int64)
    arena = dev.empty(int(offs[-1]), thin.dtype)
    cplx = thin.dtype.kind == 'c'

    def put(src, src_offs, widths, col0):
        sel = np.nonzero(widths > 0)[0]
        if len(sel) == 0:
            return
        m, w, c0 = ms[sel], widths[sel], col0[sel]
        if not transposed_rows:     # dst[r, c0 + c] = src[r, c]
            jobs = _copy_jobs_2d(offs[sel] + c0, m, src_offs[sel], w, m, w)
        else:                       # dst[c0 + c, r] = conj(src[r, c])
            jobs = _copy_jobs_2d(offs[sel] + c0 * m, m, src_offs[sel], w, w, m, src_transposed=True, conj=cplx)
        _run_copy(thin.dtype, jobs, int(np.max(m * w)), src._arena, arena)

    put(thin, thin._offsets, ks.astype(np.int64), np.zeros(nblk, np.int64))
    if comp is not None and comp.stored_blocks:
        where = {int(q): i for i, q in enumerate(comp._qdata[:, 0])}
        c_offs = np.array([comp._offsets[where[int(q)]] if int(q) in where else 0 for q in sectors], dtype=np.int64)
        put(comp, c_offs, (ms - ks).astype(np.int64), ks.astype(np.int64))
    return arena, offs[:-1]


def _svd_full_matrices(a, U, VH, ms, ns, ks, qtotal_L, qtotal_R):
    """``U`` -> ``m x m`` and ``VH`` -> ``n x n`` unitary blocks (reference :5012-5017: legs ``[leg0, leg0*]`` and
    ``[leg1*, leg1]``, block (q, q) for every stored block of ``a``)."""
    qi_L, qi_R = a._qdata[:, 0], a._qdata[:, 1]
    Uf = Array([a.legs[0], a.legs[0].conj()], a.dtype, qtotal_L)
    Uf._arena, Uf._offsets = _widen_isometry(U, ms, ks, qi_L, a.legs[0])
    Uf._qdata = np.ascontiguousarray(np.stack([qi_L, qi_L], axis=1), dtype=np.intp)
    Uf._qdata_sorted = bool(np.all(qi_L[:-1] < qi_L[1:]))
    V = VH.conj().itranspose()           # columns = right singular vectors, legs [leg1*, inner]
    V._repack()
    Vf = Array([a.legs[1].conj(), a.legs[1]], a.dtype, qtotal_R)
    Vf._arena, Vf._offsets = _widen_isometry(V, ns, ks, qi_R, V.legs[0], transposed_rows=True)
    Vf._qdata = np.ascontiguousarray(np.stack([qi_R, qi_R], axis=1), dtype=np.intp)
    Vf._qdata_sorted = a._qdata_sorted
    return Uf, Vf


def _qr_device(a):
    """Reduced block QR of a blocked matrix on the device: returns ``(Q_arena, q_offs, R_arena, r_offs, ms, ns, ks)``."""
    offs, ms, ns = _blocked_matrix_jobs(a)
    ks = np.minimum(ms, ns)
    nblk = len(ms)
    q_offs = np.concatenate([[0], np.cumsum(ms * ks)])
    r_offs = np.concatenate([[0], np.cumsum(ks * ns)])
    jobs = np.zeros((nblk, 8), dtype=np.int64)
    jobs[:, 0], jobs[:, 1], jobs[:, 2], jobs[:, 3], jobs[:, 4] = offs, ms, ns, q_offs[:-1], r_offs[:-1]
    Q_arena = dev.empty(int(q_offs[-1]), a.dtype)
    R_arena = dev.empty(int(r_offs[-1]), a.dtype)
    dev.check(dev.lib().tpa_qr_batch(dev.code(a.dtype), jobs.ctypes.data, nblk, a._arena.data_ptr(), Q_arena.data_ptr(),
                                     R_arena.data_ptr(), dev.stream()), "qr_batch")
    return Q_arena, q_offs, R_arena, r_offs, ms, ns, ks


def _qr_rank_cut(a, cutoff):
    """``qr(a, cutoff=...)``: discard (numerically) linearly dependent directions, reference :4191 -> ``tools/math.py:255``
    ``qr_li`` (pivoted QR, drop ``|R_ii| <= cutoff``, un-pivot, second QR).  Device route: the rank-revealing block SVD
    plays the part of the pivoted QR -- ``a = U_k (S_k VH_k)`` with the singular values ``> cutoff`` kept (``sigma_i`` and
    the ``|R_ii|`` of a pivoted QR bound each other, so the same directions go) -- and one more block QR of
    ``S_k VH_k`` makes ``R`` upper triangular: ``Q = U_k q``.  Returns blocked ``(Q, R)`` with a fresh inner leg."""
    U, S, VH = svd(a, cutoff=cutoff, inner_labels=[None, None])
    VH.iscale_axis(S, 0)
    q, R = qr(VH, mode='reduced')
    return tensordot(U, q, axes=1), R


def qr_batched(arrays, inner_labels=[None, None], inner_qconj=+1):
    """``[qr(a, inner_labels=inner_labels, inner_qconj=inner_qconj) for a in arrays]`` (reduced mode) for INDEPENDENT matrices -- the
    bonds of one Trotter half-step of the QR-based TEBD, reference ``algorithms/tebd.py:374-414`` / ``truncation.py:611-640`` -- with
    the charge blocks of all of them in ONE ``tpa_qr_batch`` call: the blocked Householder QR is a chain of panel / update launches
    that keeps a few CUs busy per matrix, independent matrices share the chain.  The inputs stay where they are (job offsets are
    taken relative to the lowest arena address); per matrix the result is bit-identical to :func:`qr`."""
    n = len(arrays)
    if n == 0:
        return []
    dtype = arrays[0].dtype
    blocked = [a.as_completely_blocked() for a in arrays]
    if n == 1 or any(ab.stored_blocks == 0 or ab.dtype != dtype for _, ab in blocked):
        return [qr(a, inner_labels=inner_labels, inner_qconj=inner_qconj) for a in arrays]
    isz = np.dtype(dtype).itemsize
    base = min(ab._arena.data_ptr() for _, ab in blocked)
    jobs_all, per = [], []
    q_base = r_base = 0
    for _, ab in blocked:
        offs, ms, ns = _blocked_matrix_jobs(ab)
        rel = ab._arena.data_ptr() - base
        if rel % isz:            # (cannot happen with torch's 256-B aligned allocations; views into arenas keep element alignment)
            return [qr(a, inner_labels=inner_labels, inner_qconj=inner_qconj) for a in arrays]
        ks = np.minimum(ms, ns)
        q_offs = np.concatenate([[0], np.cumsum(ms * ks)])
        r_offs = np.concatenate([[0], np.cumsum(ks * ns)])
        jobs = np.zeros((len(ms), 8), dtype=np.int64)
        jobs[:, 0], jobs[:, 1], jobs[:, 2] = offs + rel // isz, ms, ns
        jobs[:, 3], jobs[:, 4] = q_base + q_offs[:-1], r_base + r_offs[:-1]
        jobs_all.append(jobs)
        per.append((q_base, q_offs, r_base, r_offs, ms, ns, ks))
        q_base += int(q_offs[-1])
        r_base += int(r_offs[-1])
    jobs_all = np.ascontiguousarray(np.concatenate(jobs_all, axis=0))
    Q_big, R_big = dev.empty(q_base, dtype), dev.empty(r_base, dtype)
    dev.check(dev.lib().tpa_qr_batch(dev.code(dtype), jobs_all.ctypes.data, len(jobs_all), base, Q_big.data_ptr(), R_big.data_ptr(),
                                     dev.stream()), "qr_batch")
    out = []
    for a, blk, (qb, q_offs, rb, r_offs, ms, ns, ks) in zip(arrays, blocked, per):
        res = (Q_big[qb:qb + int(q_offs[-1])], q_offs, R_big[rb:rb + int(r_offs[-1])], r_offs, ms, ns, ks)
        out.append(qr(a, inner_labels=inner_labels, inner_qconj=inner_qconj, _blocked=blk, _device=res))
    return out


def qr(a, mode='reduced', inner_labels=[None, None], cutoff=None, pos_diag_R=False, qtotal_Q=None, inner_qconj=+1, _blocked=None,
       _device=None):
    """Block-wise QR ``a = Q R`` (reference np_conserved.py:4139): Householder panels / compact-WY updates on the device
    for all charge blocks at once.  ``mode='complete'``: ``Q`` is completed to square unitary blocks (plus identity
    blocks for the sectors of the first leg in which ``a`` vanishes, reference :4244-4262) and ``R`` padded with zero
    rows.  ``cutoff``: see :func:`_qr_rank_cut`."""
    if a.rank != 2:
        raise ValueError("expect a matrix!")
    if mode not in ('reduced', 'complete'):
        raise ValueError("unknown mode " + repr(mode))
    a_labels = a._labels
    label_Q, label_R = inner_labels
    piped_axes, a = a.as_completely_blocked() if _blocked is None else _blocked
    chinfo = a.chinfo
    qtotal_Q_given = qtotal_Q is not None
    qtotal_Q = chinfo.make_valid(qtotal_Q)
    qtotal_R = chinfo.make_valid(a.qtotal - qtotal_Q)
    a_leg0 = a.legs[0]
    complete = (mode == 'complete')
    if a.stored_blocks == 0 and not complete:
        raise ValueError("QR of an Array without blocks")
    qi_L, qi_R = a._qdata[:, 0], a._qdata[:, 1]
    if cutoff is not None:
        if complete:
            raise NotImplementedError("tenpy_amd: qr with both cutoff and mode='complete'")
        Qc, Rc = _qr_rank_cut(a, cutoff)
        Qc._repack()
        Rc._repack()
        order_q = np.argsort(Qc._qdata[:, 0], kind='stable')
        # blocks of `a` that survive, in the order of a._qdata; ks = kept rank per block
        rank_of = {int(q): int(Qc.legs[1].get_block_sizes()[c]) for q, c in Qc._qdata}
        ks_all = np.array([rank_of.get(int(q), 0) for q in qi_L], dtype=np.int64)
        keep_blocks = ks_all > 0
        qi_L, qi_R = qi_L[keep_blocks], qi_R[keep_blocks]
        ks = ks_all[keep_blocks]
        q_pos = {int(q): i for i, q in enumerate(Qc._qdata[:, 0])}
        r_pos = {int(q): i for i, q in enumerate(Rc._qdata[:, 1])}
        Q_arena, R_arena = Qc._arena, Rc._arena
        q_offs_b = np.array([Qc._offsets[q_pos[int(q)]] for q in qi_L], dtype=np.int64)
        r_offs_b = np.array([Rc._offsets[r_pos[int(q)]] for q in qi_R], dtype=np.int64)
        ms = a_leg0.get_block_sizes()[qi_L].astype(np.int64)
        ns = a.legs[1].get_block_sizes()[qi_R].astype(np.int64)
        del order_q
    elif a.stored_blocks:
        Q_arena, q_offs, R_arena, r_offs, ms, ns, ks = _qr_device(a) if _device is None else _device      # (_device: qr_batched)
        q_offs_b, r_offs_b = q_offs[:-1].astype(np.int64), r_offs[:-1].astype(np.int64)
    else:
        ms = ns = ks = q_offs_b = r_offs_b = np.zeros(0, np.int64)
        Q_arena = R_arena = dev.empty(0, a.dtype)
    nblk = len(ms)
    # ---- inner leg (reference :4186-4232): leg 0 restricted to the first k indices of every block row ('reduced')
    inner_leg = a_leg0.to_LegCharge() if isinstance(a_leg0, LegPipe) else a_leg0.copy()
    if not complete:
        mask = np.zeros(a_leg0.ind_len, dtype=np.bool_)
        for q1, k in zip(qi_L, ks):
            i0 = a_leg0.slices[q1]
            mask[i0:i0 + k] = True
        map_qind, _, inner_leg = inner_leg.project(mask)
        qi_C = map_qind[qi_L]
    else:
        qi_C = qi_L
    if qtotal_Q_given:
        inner_leg.charges = chinfo.make_valid(inner_leg.charges - inner_leg.qconj * qtotal_Q)
        inner_leg.sorted = False
    if inner_leg.qconj != inner_qconj:
        inner_leg.charges = chinfo.make_valid(-inner_leg.charges)
        inner_leg.sorted = False
        inner_leg.qconj = inner_qconj
    Q = Array([a_leg0, inner_leg.conj()], a.dtype, qtotal_Q)
    R = Array([inner_leg, a.legs[1]], a.dtype, qtotal_R)
    Q._qdata = np.ascontiguousarray(np.stack([qi_L, qi_C], axis=1), dtype=np.intp).reshape(nblk, 2)
    Q._offsets, Q._arena, Q._qdata_sorted = q_offs_b, Q_arena, False
    R._qdata = np.ascontiguousarray(np.stack([qi_C, qi_R], axis=1), dtype=np.intp).reshape(nblk, 2)
    R._offsets, R._arena, R._qdata_sorted = r_offs_b, R_arena, False
    if pos_diag_R and nblk:
        # phases of diag(R) per block: tiny D2H of the diagonals, then two axis scalings on the device
        k_offs = np.concatenate([[0], np.cumsum(ks)])
        diag_idx = np.concatenate([r_offs_b[b] + np.arange(ks[b]) * (ns[b] + 1) for b in range(nblk)])
        d = dev.to_host(dev.take(R_arena, diag_idx))
        phb = np.where(np.abs(d) > 0, d / np.where(np.abs(d) > 0, np.abs(d), 1.), 1.)
        if a.dtype.kind != 'c':
            phb = phb.real
        # block order -> flat index order of the (reduced) inner leg
        red_slices = np.concatenate([[0], np.cumsum(ks)]) if complete else None
        ph = np.ones(int(np.sum(ks)) if complete else inner_leg.ind_len, dtype=phb.dtype)
        for b in range(nblk):
            i0 = red_slices[b] if complete else inner_leg.slices[qi_C[b]]
            ph[i0:i0 + ks[b]] = phb[k_offs[b]:k_offs[b + 1]]
        if complete:        # scale the thin factors through temporary views with the reduced inner leg
            thin_leg = LegCharge.from_qind(chinfo, red_slices, inner_leg.charges[qi_C], inner_leg.qconj)
            Qt = Array([a_leg0, thin_leg.conj()], a.dtype, qtotal_Q)
            Rt = Array([thin_leg, a.legs[1]], a.dtype, qtotal_R)
            seq = np.arange(nblk, dtype=np.intp)
            Qt._qdata, Qt._offsets, Qt._arena = np.ascontiguousarray(np.stack([qi_L, seq], axis=1)), q_offs_b, Q_arena
            Rt._qdata, Rt._offsets, Rt._arena = np.ascontiguousarray(np.stack([seq, qi_R], axis=1)), r_offs_b, R_arena
            Qt.iscale_axis(ph, 1)
            Rt.iscale_axis(np.conj(ph), 0)
        else:
            Q.iscale_axis(ph, 1)
            R.iscale_axis(np.conj(ph), 0)
    if complete:
        Q, R = _qr_complete(a, Q, R, ms, ns, ks, inner_leg, qtotal_Q, qtotal_R)
    if 0 in piped_axes:
        Q = Q.split_legs(0)
    if 1 in piped_axes:
        R = R.split_legs(-1)
    Q.iset_leg_labels([a_labels[0], label_Q])
    R.iset_leg_labels([label_R, a_labels[1]])
    return Q, R


def _qr_complete(a, Q, R, ms, ns, ks, inner_leg, qtotal_Q, qtotal_R):
    """Square ``Q`` blocks ``[Q_thin | complement]``, identity for the sectors of leg 0 without a block of ``a``, and
    ``R`` blocks ``m x n`` with zero rows below the ``k x n`` triangle (reference :4244-4262)."""
    a_leg0 = a.legs[0]
    chinfo = a.chinfo
    nblk = len(ms)
    qi_L, qi_R = a._qdata[:, 0], a._qdata[:, 1]
    # thin Q with a plain sector-per-block inner leg, for the completion routine
    thin_leg = LegCharge.from_qind(chinfo, np.concatenate([[0], np.cumsum(ks)]), np.zeros((nblk, chinfo.qnumber), QTYPE), -a_leg0.qconj)
    Qt = Array.__new__(Array)
    Q._arena
    Qt.__dict__.update(Q.__dict__)
    thin_charges = chinfo.make_valid(-a_leg0.charges[qi_L] * a_leg0.qconj * thin_leg.qconj)
    thin_leg = LegCharge.from_qind(chinfo, thin_leg.slices, thin_charges, thin_leg.qconj)
    Qt.legs = [a_leg0, thin_leg]
    Qt.qtotal = chinfo.make_valid()
    Qt._labels = [None, None]
    Qt._qdata = np.ascontiguousarray(np.stack([qi_L, np.arange(nblk)], axis=1), dtype=np.intp).reshape(nblk, 2)
    Qt._set_shape()
    Qt._skey = None
    Qt.__dict__.pop('_sz_cache', None)
    Qt.__dict__.pop('_pk_cache', None)
    all_q = np.arange(a_leg0.block_number, dtype=np.intp)
    sizes_all = a_leg0.get_block_sizes().astype(np.int64)
    have = np.zeros(a_leg0.block_number, dtype=bool)
    have[qi_L] = True
    missing = all_q[~have]
    # square blocks for the sectors that have data
    if nblk:
        arena_have, offs_have = _widen_isometry(Qt, ms, ks, qi_L, a_leg0)
    else:
        arena_have, offs_have = dev.empty(0, a.dtype), np.zeros(0, np.int64)
    qdata = [np.stack([qi_L, qi_L], axis=1)] if nblk else []
    if len(missing):
        eye = np.concatenate([np.eye(int(n), dtype=a.dtype).reshape(-1) for n in sizes_all[missing]])
        n_have = int(arena_have.numel())
        full = dev.empty(n_have + len(eye), a.dtype)
        if n_have:
            _run_copy(a.dtype, _copy_jobs_contiguous(np.zeros(1, np.int64), np.zeros(1, np.int64), np.array([n_have])), n_have,
                      arena_have, full)
        eye_dev = dev.to_device(eye)
        _run_copy(a.dtype, _copy_jobs_contiguous(np.array([n_have], np.int64), np.zeros(1, np.int64), np.array([len(eye)])),
                  len(eye), eye_dev, full)
        m_offs = n_have + np.concatenate([[0], np.cumsum(sizes_all[missing] ** 2)])[:-1]
        arena_have, offs_have = full, np.concatenate([offs_have, m_offs]).astype(np.int64)
        qdata.append(np.stack([missing, missing], axis=1))
    Qf = Array([a_leg0, inner_leg.conj()], a.dtype, qtotal_Q)
    Qf._qdata = np.ascontiguousarray(np.concatenate(qdata, axis=0), dtype=np.intp)
    Qf._offsets, Qf._arena, Qf._qdata_sorted = offs_have, arena_have, False
    # R: zero-padded m x n blocks
    Rf = Array([inner_leg, a.legs[1]], a.dtype, qtotal_R)
    if nblk:
        Rf._set_blocks(np.stack([qi_L, qi_R], axis=1), zero=True, qdata_sorted=False)
        _run_copy(a.dtype, _copy_jobs_2d(Rf._offsets, ns, R._offsets, ns, ks, ns), int(np.max(ks * ns)), R._arena, Rf._arena)
    return Qf, Rf


def orthogonal_columns(a, new_label=None):
    """Isometry whose columns are orthonormal and orthogonal to all columns of the full-rank ``M x N`` matrix ``a``,
    ``M >= N`` (reference np_conserved.py:4291): the block QR gives ``range(a)``, the completion of that isometry on
    the device (``_complement_blocks``) the rest; sectors in which ``a`` vanishes contribute identity blocks."""
    if a.rank != 2:
        raise ValueError("expect a matrix!")
    M, N = a.shape
    a_labels = a._labels
    if new_label is None:
        new_label = a_labels[1]
    if M < N:
        raise ValueError("orthogonal_columns with M={0:d} < N{1:d}: overcomplete! ".format(M, N))
    right_qconj = a.legs[1].qconj
    if M == N:
        warnings.warn("orthogonal_columns(a) for square `a` yields zero matrix!")
        right_leg = LegCharge(a.chinfo, [0], np.zeros([0, a.chinfo.qnumber], dtype=QTYPE), right_qconj)
        return Array([a.legs[0], right_leg], a.dtype, a.qtotal, [a_labels[0], new_label])
    piped_axes, a = a.as_completely_blocked()
    left = a.legs[0]
    if a.stored_blocks:
        Qa, _ = qr(a.copy(deep=False).idrop_labels(), mode='reduced')
        comp = _complement_blocks(Qa)
    else:
        comp = diag(1., left, dtype=a.dtype)
    comp._repack()
    order = np.argsort(comp._qdata[:, 0], kind='stable')
    kept = comp._qdata[order, 0]
    widths = comp.legs[1].get_block_sizes()[comp._qdata[order, 1]]
    right_charges = a.chinfo.make_valid(right_qconj * (a.qtotal - left.get_charge(kept)))
    right_leg = LegCharge(a.chinfo, np.concatenate([[0], np.cumsum(widths)]), right_charges, right_qconj)
    ortho = Array([left, right_leg], a.dtype, a.qtotal)
    ortho._qdata = np.ascontiguousarray(np.stack([kept, np.arange(len(kept))], axis=1), dtype=np.intp)
    ortho._offsets, ortho._arena, ortho._qdata_sorted = comp._offsets[order].astype(np.int64), comp._arena, True
    if 0 in piped_axes:
        ortho = ortho.split_legs(0)
    ortho.iset_leg_labels([a_labels[0], new_label])
    return ortho


def _permute_within_blocks(a, perm_flat, axis):
    """``a`` with the flat indices of ``axis`` permuted inside each charge block: res[.., j, ..] = a[.., perm[j], ..]."""
    res = a.copy(deep=False)
    res._qdata, res._offsets = a._qdata.copy(), a._offsets.copy()
    if a.stored_blocks == 0:
        return res
    leg = a.legs[axis]
    shapes = a._block_shapes()
    sizes = np.prod(shapes, axis=1)
    res._arena = dev.empty(a._arena.numel(), a.dtype)
    jobs = np.zeros((a.stored_blocks, 8), dtype=np.int64)
    jobs[:, 0] = jobs[:, 1] = a._offsets
    jobs[:, 2] = np.prod(shapes[:, :axis], axis=1)
    jobs[:, 3] = jobs[:, 4] = shapes[:, axis]
    jobs[:, 5] = np.prod(shapes[:, axis + 1:], axis=1)
    starts = leg.slices[a._qdata[:, axis]]
    jobs[:, 6] = starts
    local = np.asarray(perm_flat, dtype=np.int64).copy()
    for q in range(leg.block_number):
        local[leg.slices[q]:leg.slices[q + 1]] -= leg.slices[q]
    jd, idd = dev.table(jobs), dev.table(local)
    dev.check(dev.lib().tpa_gather_axis_batch(dev.code(a.dtype), jd.data_ptr(), len(jobs), int(np.max(sizes)), idd.data_ptr(),
                                              a._arena.data_ptr(), res._arena.data_ptr(), dev.stream()), "gather")
    res._skey = None
    return res


def lq(a, mode='reduced', inner_labels=[None, None], cutoff=None, pos_diag_L=False, qtotal_Q=None, inner_qconj=+1):
    """L-Q decomposition through :func:`qr` of the transpose (reference np_conserved.py:4273)."""
    q, r = qr(a.transpose(), mode=mode, inner_labels=inner_labels[::-1], cutoff=cutoff, pos_diag_R=pos_diag_L,
              qtotal_Q=qtotal_Q, inner_qconj=inner_qconj)
    return r.transpose(), q.transpose()


def _argsort(w, sort):
    """tools.misc.argsort of the reference: 'm>' / 'm<' by magnitude, '>' / '<' by real part."""
    if sort is None or sort == '<':
        return np.argsort(w, kind='stable')
    if sort == '>':
        return np.argsort(-w, kind='stable')
    if sort == 'm<':
        return np.argsort(np.abs(w), kind='stable')
    if sort == 'm>':
        return np.argsort(-np.abs(w), kind='stable')
    raise ValueError("unknown sort option " + repr(sort))


def eigvalsh(a, UPLO='L', sort=None):
    """Eigenvalues of a hermitian block matrix (reference :3972)."""
    return eigh(a, UPLO, sort)[0]


# Real Hermitian blocks: eigenpairs out of the block SVD (pivoted QR + one-sided Jacobi on the rank-r factor), CHECKED on the device
# (`tpa_eigh_from_svd`: |A u_i - lambda_i u_i| per vector); the two-sided Jacobi iteration of `tpa_eigh_batch` takes over if the check fails
# (+/- lambda pairs, input that is not Hermitian in its upper triangle -- the reference reads the lower one only).  Why: the density matrices of
# the mixer (`mix_rho`, mps_common.py:1972-2079) are graded over 14+ decades and rank deficient; Jacobi on the matrix itself converges
# linearly on such clusters (34 - 39 sweeps at 1086 rows), the SVD path needs ~7 on half the rows.  Complex data (the flat-spectrum bond
# matrices of the TEBD eig route) keeps the two-sided iteration: no QR, no Gram products, 15 sweeps against 12 + QR.
EIGH_VIA_SVD = os.environ.get('TPA_EIGH_VIA_SVD', '1') != '0'
EIGH_VIA_SVD_MIN_ROWS = 96
EIGH_VIA_SVD_TOL = 5.e-13      # x sqrt(n) |A_b|_2: the GATE on max_i |A u_i - lambda_i u_i| (measured: a few eps sqrt(n) |A_b| on accepted blocks; O(|A_b|) on +/- pairs)
eigh_stats = {'svd_calls': 0, 'svd_rejected': 0, 'jacobi_calls': 0, 'last_err_rel': 0.}


def _eigh_via_svd(L, code, dtype, jobs, a_arena, W_dev, V_big):
    """Eigenvalues (host, per block in the order of the singular values) with the eigenvectors written to the blocks of ``V_big``, or
    ``None`` if the SVD of some block is not an eigendecomposition to rounding level (nothing is lost: the caller runs ``tpa_eigh_batch``)."""
    nblk = len(jobs)
    ns = jobs[:, 1]
    vh_offs = np.concatenate([[0], np.cumsum(ns * ns)])
    sj = np.zeros((nblk, 8), dtype=np.int64)
    sj[:, 0], sj[:, 1], sj[:, 2] = jobs[:, 0], ns, ns
    sj[:, 3], sj[:, 4], sj[:, 5] = jobs[:, 3], jobs[:, 2], vh_offs[:-1]      # U -> the blocks of V_big, S -> W_dev
    VH = dev.scratch('eigh_vh', int(vh_offs[-1]), dtype)
    S_dev = dev.scratch('eigh_s', int(W_dev.numel()), np.float64)
    sweeps = dev.c_int()
    floor_was = _svd_floor_now[0]
    _svd_floor_now[0] = SVD_ABS_FLOOR_GENERIC          # every vector converged by the relative rule (this is a library call, not an engine's svd_theta)
    try:
        S_host = _svd_batch_robust(L, code, sj, nblk, a_arena, V_big, S_dev, VH, sweeps)
    except (np.linalg.LinAlgError, ValueError):
        return None
    finally:
        _svd_floor_now[0] = floor_was
    ej = np.zeros((nblk, 8), dtype=np.int64)
    ej[:, 0], ej[:, 1], ej[:, 2], ej[:, 3], ej[:, 4] = jobs[:, 3], ns, jobs[:, 2], vh_offs[:-1], jobs[:, 2]
    err_dev = dev.empty(nblk, np.float64)
    dev.check(L.tpa_eigh_from_svd(code, ej.ctypes.data, nblk, V_big.data_ptr(), S_dev.data_ptr(), VH.data_ptr(), W_dev.data_ptr(),
                                  err_dev.data_ptr(), dev.stream()), "eigh_from_svd")
    lam = dev.to_host(W_dev)
    err = dev.to_host(err_dev)
    smax = np.array([S_host[o] if n else 0. for o, n in zip(jobs[:, 2], ns)])      # (descending inside a block)
    tol = EIGH_VIA_SVD_TOL * np.sqrt(ns) * smax
    eigh_stats['last_err_rel'] = float(np.max(err / np.maximum(smax, 1e-300)))
    if not np.all(np.isfinite(err)) or np.any(err > tol):
        eigh_stats['svd_rejected'] += 1
        return None
    # Blocks of numerical rank r < n: the block SVD returns r vectors (the pivoted QR stops at the rank; S, U, VH are zero-padded behind
    # it).  The eigenvectors of the null space = ANY orthonormal completion: Householder QR of the zero-padded n x n block of U gives a
    # full orthogonal Q whose first r columns are +/- the r vectors (they are orthonormal: R is diagonal there), the others the complement.
    ranks = np.array([int(np.count_nonzero(S_host[o:o + n])) for o, n in zip(jobs[:, 2], ns)], dtype=np.int64)
    sel = np.nonzero(ranks < ns)[0]
    if len(sel):
        n_s, r_s = ns[sel], ranks[sel]
        q_offs = np.concatenate([[0], np.cumsum(n_s * n_s)])
        qj = np.zeros((len(sel), 8), dtype=np.int64)
        qj[:, 0], qj[:, 1], qj[:, 2], qj[:, 3], qj[:, 4] = jobs[sel, 3], n_s, n_s, q_offs[:-1], q_offs[:-1]
        Q_tmp = dev.scratch('eigh_q', int(q_offs[-1]), dtype)
        R_tmp = dev.scratch('eigh_r', int(q_offs[-1]), dtype)
        dev.check(L.tpa_qr_batch(code, qj.ctypes.data, len(sel), V_big.data_ptr(), Q_tmp.data_ptr(), R_tmp.data_ptr(), dev.stream()), "qr_batch")
        cj = _copy_jobs_2d(jobs[sel, 3] + r_s, n_s, q_offs[:-1] + r_s, n_s, n_s, n_s - r_s)
        _run_copy(dtype, cj, int(np.max(n_s * (n_s - r_s))), Q_tmp, V_big)
    eigh_stats['svd_calls'] += 1
    return lam


def eigh(a, UPLO='L', sort=None):
    """Block-wise Hermitian eigendecomposition (reference np_conserved.py:3899, worker :5041).

    Returns ``(W, V)``: eigenvalues as host 1-D array (ascending inside each charge block) and the
    eigenvectors as columns of the device Array ``V``.
    """
    return eigh_batched([a], UPLO, sort)[0]


def eigh_batched(arrays, UPLO='L', sort=None):
    """``[eigh(a, UPLO, sort) for a in arrays]`` for INDEPENDENT Hermitian matrices -- the bond matrices ``Xi^dagger Xi`` of all bonds of
    a Trotter half-step on the ``use_eig_based_svd`` route (reference ``truncation.py:473-530`` called per bond from
    ``algorithms/tebd.py:685-738``) -- as ONE ``tpa_eigh_batch`` call over the charge blocks of all of them: the Jacobi rounds of the
    eigensolver are a latency chain that fills a few CUs per matrix, independent matrices share its launches.  Per matrix the result is
    bit-identical to :func:`eigh` (the blocks never interact)."""
    if len(arrays) == 0:
        return []
    dtype = arrays[0].dtype
    preps = []
    a_base = v_base = w_base = 0
    for a in arrays:
        if a.rank != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("expect a square matrix!")
        if a.dtype != dtype:
            raise ValueError("eigh_batched: mixed dtypes")
        a.legs[0].test_contractible(a.legs[1])
        if np.any(a.qtotal != a.chinfo.make_valid()):
            raise ValueError("Non-trivial qtotal -> Nilpotent. Not diagonizable!?")
        a_labels = a._labels
        piped_axes, ab = a.as_completely_blocked()
        leg = ab.legs[0]
        n_all = leg.get_block_sizes().astype(np.int64)
        v_offs = np.concatenate([[0], np.cumsum(n_all * n_all)])
        pr = dict(a=ab, labels=a_labels, piped=piped_axes, leg=leg, n_all=n_all, v_offs=v_offs, a_base=a_base, v_base=v_base,
                  w_base=w_base, jobs=None)
        if ab.stored_blocks:
            offs, ms, ns = _blocked_matrix_jobs(ab)
            w_offs = np.concatenate([[0], np.cumsum(ms)])
            jobs = np.zeros((len(ms), 8), dtype=np.int64)
            jobs[:, 0], jobs[:, 1], jobs[:, 2] = offs + a_base, ms, w_offs[:-1] + w_base
            jobs[:, 3] = v_offs[:-1][ab._qdata[:, 0]] + v_base
            pr['jobs'], pr['w_offs'], pr['ms'] = jobs, w_offs, ms
            a_base += int(ab._arena.numel())
            w_base += int(w_offs[-1])
        v_base += int(v_offs[-1])
        preps.append(pr)
    # V starts as identity on every sector; sectors with a stored block get the eigenvectors
    ident = [np.eye(int(n)).reshape(-1) for pr in preps for n in pr['n_all']]
    V_big = dev.to_device((np.concatenate(ident) if ident else np.zeros(0)).astype(dtype))
    with_blocks = [pr for pr in preps if pr['jobs'] is not None]
    W_host = None
    if with_blocks:
        if len(with_blocks) == 1:
            big = with_blocks[0]['a']._arena
        else:
            big = dev.scratch('eigh_batched_in', a_base, dtype)
            for pr in with_blocks:
                n_a = int(pr['a']._arena.numel())
                big[pr['a_base']:pr['a_base'] + n_a].copy_(pr['a']._arena)
        jobs = np.ascontiguousarray(np.concatenate([pr['jobs'] for pr in with_blocks]))
        nblk = len(jobs)
        L = dev.lib()
        code = dev.code(dtype)
        W_dev = dev.empty(w_base, np.float64)
        ev = eigh_timer.begin()
        W_host = None
        if EIGH_VIA_SVD and np.dtype(dtype).kind != 'c' and int(np.max(jobs[:, 1])) >= EIGH_VIA_SVD_MIN_ROWS:
            W_host = _eigh_via_svd(L, code, dtype, jobs, big, W_dev, V_big)
        if W_host is None:
            wb = L.tpa_eigh_worksize(code, jobs.ctypes.data, nblk)
            work = dev.torch().empty(int(wb), dtype=dev.torch().uint8, device='cuda')
            sweeps = dev.c_int()
            dev.check(L.tpa_eigh_batch(code, jobs.ctypes.data, nblk, big.data_ptr(), W_dev.data_ptr(), V_big.data_ptr(),
                                       work.data_ptr(), int(wb), 60, 0.0, dev.byref(sweeps), dev.stream()), "eigh_batch")
            W_host = dev.to_host(W_dev)
            eigh_stats['jacobi_calls'] += 1
        eigh_timer.end(ev, eigh_work(jobs[:, 1], np.dtype(dtype).itemsize, np.dtype(dtype).kind == 'c'))
    out = []
    for pr in preps:
        ab, leg, v_offs = pr['a'], pr['leg'], pr['v_offs']
        nq = leg.block_number
        resw = np.zeros(ab.shape[0], dtype=np.float64)
        V = Array([leg if not isinstance(leg, LegPipe) else leg, leg.conj() if not isinstance(leg, LegPipe) else leg.to_LegCharge().conj()], dtype)
        V._qdata = np.ascontiguousarray(np.stack([np.arange(nq), np.arange(nq)], axis=1), dtype=np.intp)
        V._offsets = v_offs[:-1].astype(np.int64)
        V._arena = V_big if len(preps) == 1 else V_big[pr['v_base']:pr['v_base'] + int(v_offs[-1])]
        V._qdata_sorted = True
        if pr['jobs'] is not None:
            w_offs = pr['w_offs']
            perm_full = np.arange(ab.shape[0], dtype=np.int64)
            need_perm = False
            for b in range(len(pr['ms'])):
                qi = ab._qdata[b, 0]
                w = W_host[pr['w_base'] + w_offs[b]:pr['w_base'] + w_offs[b + 1]]
                # (tpa_eigh_batch returns ascending values; the SVD route returns them by descending magnitude)
                if (sort is not None and sort != '<') or (len(w) > 1 and np.any(w[1:] < w[:-1])):
                    pb = _argsort(w, sort)
                    w = w[pb]
                    sl = leg.get_slice(qi)
                    perm_full[sl] = sl.start + pb
                    need_perm = True
                resw[leg.get_slice(qi)] = w
            if need_perm:
                V = _permute_within_blocks(V, perm_full, 1)
        if len(pr['piped']) > 0:
            V = V.split_legs(0)
        V.iset_leg_labels([pr['labels'][0], 'eig'])
        out.append((resw, V))
    return out


def eig(a, sort=None):
    """General (non-hermitian) eigendecomposition ``a V = V diag(W)`` (reference :3937) -- not on the DMRG / TEBD path;
    see :mod:`tenpy_amd.linalg._host_eig`."""
    from ._host_eig import eig as _eig
    return _eig(a, sort)


def eigvals(a, sort=None):
    """Eigenvalues of a general square matrix (reference :4000); see :mod:`tenpy_amd.linalg._host_eig`."""
    from ._host_eig import eigvals as _eigvals
    return _eigvals(a, sort)


def speigs(a, charge_sector, k, *args, **kwargs):
    """``k`` eigenpairs of one charge sector by ARPACK (reference :4024); see :mod:`tenpy_amd.linalg._host_eig`."""
    from ._host_eig import speigs as _speigs
    return _speigs(a, charge_sector, k, *args, **kwargs)


from . import _npc_cold  # noqa: E402   (needs Array and the functions above)
_npc_cold.attach(__import__(__name__, fromlist=['Array']))
