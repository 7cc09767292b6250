"""The parts of the ``np_conserved`` interface that TeNPy uses OUTSIDE the DMRG / TEBD inner loop: element access,
assembly of tensors from many small pieces (``grid_outer`` for MPOs, ``grid_concat``), changes of the charge structure,
orthonormal completions (``qr(mode='complete')``, ``svd(full_matrices=True)``, ``orthogonal_columns``), HDF5 export.

They exist so that an unmodified TeNPy can be imported on top of the device ``Array`` (``tenpy_amd/install.py``): models,
sites and MPOs are built once, on the host, from operators with a handful of entries.  Design rule used here:

* whatever is *assembled from or taken apart into small host pieces* (``__getitem__`` / ``__setitem__`` with numbers or
  numpy arrays, ``grid_outer``, ``add_charge``, ``permute`` across charge sectors, ``__iter__``, HDF5) goes through host
  copies of the blocks and ONE upload -- these are boundary crossings by definition (the reference's callers use them at
  set-up time, never per bond update);
* whatever is *linear algebra on resident data* (completions of isometries, ``eigh``-like calls) is expressed through the
  device entry points that already exist (block GEMM, block SVD, strided copies) -- no host LAPACK.

Reference line numbers are those of ``tenpy/linalg/np_conserved.py``.
"""
import itertools
import warnings

import numpy as np

from . import _device as dev
from .charges import QTYPE, ChargeInfo, LegCharge, LegPipe

npc = None      # the np_conserved module, set by attach()


# ======================================================================================================================
# host <-> device helpers
# ======================================================================================================================

def _array_from_host_blocks(legs, dtype, qtotal, labels, blocks):
    """``blocks``: dict qindex-tuple -> host ndarray.  One upload; ``_qdata`` comes out lexsorted."""
    res = npc.Array(legs, dtype, qtotal, labels)
    if len(blocks) == 0:
        return res
    qdata = np.array(list(blocks.keys()), dtype=np.intp).reshape(len(blocks), res.rank)
    order = np.lexsort(qdata.T)
    qdata = qdata[order]
    vals = list(blocks.values())
    flat = np.concatenate([np.ascontiguousarray(vals[i], dtype=res.dtype).reshape(-1) for i in order])
    res._set_blocks(qdata, arena=dev.to_device(flat), qdata_sorted=True)
    return res


def _array_from_dense(dense, legs, dtype, qtotal, labels, what="array"):
    """Dense host array -> Array holding every charge-allowed block with a non-zero entry (exact test, no cutoff);
    non-zero entries outside the allowed blocks raise like the reference's charge checks do."""
    res = npc.Array(legs, dtype, qtotal, labels)
    blocks, seen = {}, 0
    for row in res._allowed_qdata():
        sl = tuple(leg.get_slice(q) for leg, q in zip(res.legs, row))
        blk = dense[sl]
        nz = np.count_nonzero(blk)
        if nz:
            blocks[tuple(int(q) for q in row)] = blk
            seen += nz
    if seen != np.count_nonzero(dense):
        raise ValueError("wrong charge: " + what + " has non-zero entries in blocks incompatible with the charges")
    return _array_from_host_blocks(res.legs, res.dtype, res.qtotal, res._labels, blocks)


def _host_block_dict(a):
    return {tuple(int(q) for q in row): blk for row, blk in zip(a._qdata, a._data)}


# ======================================================================================================================
# Array methods (attached to the class by attach())
# ======================================================================================================================

def _str(self):
    """Multi-line description; small tensors are printed densely (reference :840)."""
    head = "<npc.Array(device) shape={0!s} labels={1!r}".format(self.shape, self._labels)
    lines = [head, "charge=" + str(self.chinfo)]
    lines.extend("leg {0:d}: {1!s}".format(i, leg).replace("\n", " | ") for i, leg in enumerate(self.legs))
    if self.size < 100:
        lines.append(str(self.to_ndarray()))
    return "\n".join(lines) + "\n>"


def _iter(self):
    """Iterate over the stored blocks: ``(host copy of the block, slices, charges per leg, qindices)`` (reference :897)."""
    for blk, row in zip(self._data, self._qdata):
        yield (blk, tuple(leg.get_slice(q) for leg, q in zip(self.legs, row)),
               [leg.get_charge(q) for leg, q in zip(self.legs, row)], row)


def _iter_all_blocks(self):
    """All qindex tuples in lexicographic order, last leg most significant (reference :2498)."""
    for rev in itertools.product(*[range(leg.block_number) for leg in reversed(self.legs)]):
        yield tuple(rev[::-1])


def _get_block_charge(self, qindices):
    q = self.chinfo.make_valid()
    for leg, qi in zip(self.legs, qindices):
        q = q + leg.get_charge(qi)
    return self.chinfo.make_valid(q)


def _get_block_slices(self, qindices):
    return tuple(leg.get_slice(qi) for leg, qi in zip(self.legs, qindices))


def _get_block_shape(self, qindices):
    return tuple(int(leg.slices[qi + 1] - leg.slices[qi]) for leg, qi in zip(self.legs, qindices))


class HostBlock(np.ndarray):
    """Host copy of ONE charge block, as handed out by ``Array.get_block``: a numpy array whose in-place modifications
    (``blk[...] = x``, ``blk += x``, ...) are written through to the block's place in the device arena.

    The reference's callers use ``get_block`` for exactly that (e.g. ``full_diag_effH``, algorithms/dmrg.py:1208:
    ``theta.get_block(qi, insert=True)[:] = V[:, 0]`` after a dense ``eigh`` of a small effective Hamiltonian on the host);
    with numpy blocks the write lands in ``_data`` directly, here it is one small H2D copy."""

    def __new__(cls, data, owner, offset):
        obj = np.ascontiguousarray(data).view(cls)
        obj._root, obj._owner, obj._offset = obj, owner, int(offset)
        return obj

    def __array_finalize__(self, parent):
        self._root = getattr(parent, '_root', None)
        self._owner = getattr(parent, '_owner', None)
        self._offset = getattr(parent, '_offset', 0)

    def _write_through(self):
        root = self._root
        if root is None or self._owner is None or not np.shares_memory(self, root):
            return
        self._owner._own_arena()
        arena = self._owner._arena
        flat = np.asarray(root).reshape(-1)
        arena[self._offset:self._offset + flat.size].copy_(dev.to_device(flat.astype(self._owner.dtype, copy=False)))

    def __setitem__(self, key, value):
        np.ndarray.__setitem__(self, key, value)
        self._write_through()


def _inplace(name):
    base = getattr(np.ndarray, name)

    def op(self, other):
        res = base(self, other)
        self._write_through()
        return res
    op.__name__ = name
    return op


for _n in ('__iadd__', '__isub__', '__imul__', '__itruediv__'):
    setattr(HostBlock, _n, _inplace(_n))


def _get_block(self, qindices, insert=False):
    """Host view of the block with the given qindices (``None`` if it is not stored and ``insert`` is False; with
    ``insert`` a zero block is appended to the arena first).  See :class:`HostBlock` (reference :999)."""
    qindices = np.asarray(qindices, dtype=np.intp).reshape(-1)
    if not np.all(self._get_block_charge(qindices) == self.qtotal):
        raise IndexError("trying to get block for qindices incompatible with charges")
    hit = np.nonzero(np.all(self._qdata == qindices[np.newaxis, :], axis=1))[0] if self.stored_blocks else []
    shape = self._get_block_shape(qindices)
    n = int(np.prod(shape))
    if len(hit) == 0:
        if not insert:
            return None
        old_n = 0 if self._arena is None else int(self._arena.numel())
        arena = dev.zeros(old_n + n, self.dtype)
        if old_n:
            arena[:old_n].copy_(self._arena)
        self._arena = arena
        self._qdata = np.ascontiguousarray(np.concatenate([self._qdata, qindices[np.newaxis, :]], axis=0), dtype=np.intp)
        self._offsets = np.concatenate([self._offsets, [old_n]]).astype(np.int64)
        self._qdata_sorted = False
        self._skey = None
        off = old_n
    else:
        off = int(self._offsets[int(hit[0])])
    host = np.array(dev.to_host(self._arena[off:off + n]), copy=True).reshape(shape)     # never a view of the arena itself
    return HostBlock(host, self, off)


def _pre_indexing(self, inds):
    """Normalise ``self[inds]``: returns ``(all_integer, tuple with one entry per leg)`` (reference :2600)."""
    if type(inds) is not tuple:
        inds = (inds,)
    n_ell = sum(1 for i in inds if i is Ellipsis)
    if n_ell > 1:
        raise IndexError("an index can only have a single ellipsis ('...')")
    if n_ell == 0 and len(inds) < self.rank:
        inds = inds + (Ellipsis,)
        n_ell = 1
    if n_ell:
        at = next(k for k, i in enumerate(inds) if i is Ellipsis)
        inds = inds[:at] + (slice(None),) * (self.rank - len(inds) + 1) + inds[at + 1:]
    if len(inds) > self.rank:
        raise IndexError("too many indices for Array")
    all_int = all(isinstance(i, (int, np.integer)) for i in inds)
    return all_int, inds


def _locate(self, inds):
    """(qindices, positions inside the block) of one element."""
    pos = [leg.get_qindex(int(i)) for i, leg in zip(inds, self.legs)]
    return np.array([p[0] for p in pos], dtype=np.intp), tuple(p[1] for p in pos)


def _getitem(self, inds):
    """``self[inds]`` (reference :920): all integers -> one number (a single-element D2H; 0 for a block that is not
    stored); otherwise integers fix legs (``take_slice``), slices / masks / index arrays project legs (``iproject``) and
    unsorted index arrays permute afterwards."""
    all_int, inds = self._pre_indexing(inds)
    if all_int:
        qind, within = _locate(self, inds)
        hit = np.nonzero(np.all(self._qdata == qind[np.newaxis, :], axis=1))[0] if self.stored_blocks else []
        if len(hit) == 0:
            return self.dtype.type(0)
        b = int(hit[0])
        off = int(self._offsets[b] + np.ravel_multi_index(within, self._get_block_shape(qind)))
        return self.dtype.type(dev.to_host(self._arena[off:off + 1])[0])
    return _advanced_getitem(self, inds)


def _classify_indices(self, inds):
    """Split per-leg indices into fixed ones, projections (boolean masks) and permutations applied after projecting."""
    fixed_idx, fixed_axes, masks, mask_axes, perms = [], [], [], [], []
    for a, i in enumerate(inds):
        if isinstance(i, slice):
            if i == slice(None):
                continue
            m = np.zeros(self.shape[a], dtype=np.bool_)
            m[i] = True
            masks.append(m)
            mask_axes.append(a)
            if i.step is not None and i.step < 0:
                perms.append((a, np.arange(int(np.count_nonzero(m)), dtype=np.intp)[::-1]))
        elif np.ndim(i) == 0 and not isinstance(i, (list, tuple)):
            fixed_idx.append(int(i))
            fixed_axes.append(a)
        else:
            i = np.asarray(i)
            mask_axes.append(a)
            if i.dtype == np.bool_:
                masks.append(i)
            else:
                m = np.zeros(self.shape[a], dtype=np.bool_)
                m[i] = True
                masks.append(m)
                order = np.argsort(i, kind='stable')
                if np.any(order != np.arange(len(order))):
                    inv = np.empty(len(order), dtype=np.intp)
                    inv[order] = np.arange(len(order), dtype=np.intp)
                    perms.append((a, inv))
    return fixed_idx, fixed_axes, masks, mask_axes, perms


def _advanced_getitem(self, inds, permute=True):
    fixed_idx, fixed_axes, masks, mask_axes, perms = _classify_indices(self, inds)
    res = self.take_slice(fixed_idx, fixed_axes)
    new_axis = np.cumsum([a not in fixed_axes for a in range(self.rank)]) - 1
    if masks:
        res.iproject(masks, [int(new_axis[a]) for a in mask_axes])
    if permute:
        for a, p in perms:
            res = res.permute(p, int(new_axis[a]))
    return res


def _setitem(self, inds, other):
    """``self[inds] = other`` (reference :971).  A boundary operation: the tensor is edited in a dense host copy and
    uploaded again with the same legs and total charge; values that violate the charge rule raise ``ValueError``."""
    all_int, inds = self._pre_indexing(inds)
    dense = self.to_ndarray()
    if all_int:
        if not np.all(self._get_block_charge(_locate(self, inds)[0]) == self.qtotal):
            raise IndexError("trying to set an entry of a block incompatible with the charges")
        dense[tuple(int(i) for i in inds)] = other     # (cast to self.dtype like the reference's block assignment, :986)
    else:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            part = _advanced_getitem(self.zeros_like(), inds, permute=False)     # legs / qtotal of self[inds]
        fixed_idx, fixed_axes, masks, mask_axes, perms = _classify_indices(self, inds)
        if isinstance(other, npc.Array):
            if other.rank != part.rank:
                raise IndexError("wrong number of indices")
            if np.any(other.qtotal != part.qtotal):
                raise ValueError("wrong charge for assigning self[inds] = other")
            # legs are compared after undoing the permutations that index arrays imply
            chk = other
            new_axis = np.cumsum([a not in fixed_axes for a in range(self.rank)]) - 1
            for a, p in perms:
                inv = np.empty(len(p), dtype=np.intp)
                inv[p] = np.arange(len(p), dtype=np.intp)
                chk = chk.permute(inv, int(new_axis[a]))
            for pl, ol in zip(part.legs, chk.legs):
                pl.test_contractible(ol.conj())
            val = other.to_ndarray()
        else:
            val = np.asarray(other)
        # a complex `other` assigned into a real Array keeps the real part, with numpy's ComplexWarning (reference :2790;
        # networks/mpo.py:3509-3512 relies on it)
        # numpy index with the reference's "outer product" meaning of several index arrays
        basic = [slice(None)] * self.rank
        for a, i in zip(fixed_axes, fixed_idx):
            basic[a] = i
        view = dense[tuple(basic)]
        open_axes = [a for a in range(self.rank) if a not in fixed_axes]
        sel = []
        for a in open_axes:
            i = inds[a]
            if isinstance(i, slice):
                sel.append(np.arange(self.shape[a])[i])
            else:
                i = np.asarray(i)
                sel.append(np.nonzero(i)[0] if i.dtype == np.bool_ else i)
        view[np.ix_(*sel)] = val
    new = _array_from_dense(dense, self.legs, dense.dtype, self.qtotal, self._labels, what="assigned value")
    self._become(new)


def _permute(self, perm, axis):
    """``res[.., i, ..] = self[.., perm[i], ..]`` along ``axis`` (reference :1987).  A permutation may move indices
    between charge sectors, so the new leg has one sector per run of equal charges; done on a dense host copy
    ("quite slow, and usually not needed" also in the reference)."""
    axis = self.get_leg_index(axis)
    perm = np.asarray(perm, dtype=np.intp)
    old = self.legs[axis]
    if len(perm) != old.ind_len:
        raise ValueError("permutation has wrong length")
    new_leg = LegCharge.from_qflat(self.chinfo, old.to_qflat()[perm], old.qconj).bunch()[1]
    legs = list(self.legs)
    legs[axis] = new_leg
    dense = np.take(self.to_ndarray(), perm, axis=axis)
    res = _array_from_dense(dense, legs, self.dtype, self.qtotal, self._labels)
    res._qdata_sorted = res._qdata_sorted and res.stored_blocks > 0
    return res


def _extend(self, axis, extra):
    """Copy with the leg ``axis`` made longer by ``extra`` (a LegCharge, or a number of zero-charge indices); the new
    entries are zero, i.e. no block is stored for them (reference :1172)."""
    ax = self.get_leg_index(axis)
    res = self.copy(deep=True)
    res.legs[ax] = res.legs[ax].extend(extra)
    res._set_shape()
    res._skey = None
    return res


def _rebuild_with_legs(self, legs, qtotal):
    """Same entries, different charge structure: through the host blocks, placed by their index ranges."""
    dense_legs = list(legs)
    res = npc.Array(dense_legs, self.dtype, qtotal, self._labels)
    if self.stored_blocks == 0:
        return res
    dense = self.to_ndarray()
    return _array_from_dense(dense, dense_legs, self.dtype, res.qtotal, self._labels)


def _add_charge(self, add_legs, chinfo=None, qtotal=None):
    """Attach further, independent charges given by one extra LegCharge per leg (reference :1244)."""
    add_legs = list(add_legs)
    if len(add_legs) != self.rank:
        raise ValueError("wrong number of legs in `add_legs`")
    both = ChargeInfo.add([self.chinfo, add_legs[0].chinfo])
    if chinfo is not None:
        assert chinfo == both
    else:
        chinfo = both
    legs = [LegCharge.from_add_charge([l0, l1], chinfo) for l0, l1 in zip(self.legs, add_legs)]
    if qtotal is None:
        if self.stored_blocks == 0:
            raise ValueError("no non-zero entry: can't detect qtotal")
        extra = npc.detect_qtotal(self.to_ndarray(), add_legs)
    else:
        extra = np.array(qtotal, dtype=QTYPE).reshape(-1)
    return _rebuild_with_legs(self, legs, np.concatenate([self.qtotal, extra]))


def _drop_charge(self, charge=None, chinfo=None):
    """Forget one charge (or all, ``charge=None``), reference :1291."""
    new_ci = ChargeInfo.drop(self.chinfo, charge)
    if chinfo is not None:
        assert chinfo == new_ci
        new_ci = chinfo
    legs = [LegCharge.from_drop_charge(leg, charge, new_ci) for leg in self.legs]
    if charge is None:
        res = _rebuild_with_legs(self, legs, None)
    else:       # block structure unchanged: only the leg objects and qtotal lose a column
        idx = self.chinfo.names.index(charge) if isinstance(charge, str) else int(charge)
        res = self.copy(deep=True)
        res.chinfo = new_ci
        res.legs = legs
        res.qtotal = np.delete(self.qtotal, idx, 0)
        res._skey = None
    res.test_sanity()
    return res


def _change_charge(self, charge, new_qmod, new_name='', chinfo=None):
    """Take one charge modulo ``new_qmod`` from now on (reference :1335); block structure unchanged."""
    new_ci = ChargeInfo.change(self.chinfo, charge, new_qmod, new_name)
    if chinfo is not None:
        assert chinfo == new_ci
        new_ci = chinfo
    res = self.copy(deep=True)
    res.chinfo = new_ci
    res.legs = [LegCharge.from_change_charge(leg, charge, new_qmod, new_name, new_ci) for leg in self.legs]
    res.qtotal = new_ci.make_valid(self.qtotal)
    res._skey = None
    res.test_sanity()
    return res


def _apply_charge_mapping(self, map_func, func_args=(), func_kwargs={}, inplace=False):
    """Relabel the charge values of all legs and of ``qtotal`` by ``map_func`` (reference :1446); data untouched."""
    res = self if inplace else self.copy(deep=False)
    res.legs = [leg.apply_charge_mapping(map_func, func_args, func_kwargs) for leg in self.legs]
    res.qtotal = map_func(self.qtotal, *func_args, **func_kwargs)
    res._skey = None
    return res


def _shift_charges(self, dx, inplace=False):
    if self.chinfo.trivial_shift or np.all(np.equal(dx, 0)):
        return self
    return self.apply_charge_mapping(self.chinfo.shift_charges, func_kwargs=dict(dx=dx), inplace=inplace)


def _shift_charges_horizontal(self, dx_0, inplace=False):
    if self.chinfo.trivial_shift or dx_0 == 0:
        return self
    return self.apply_charge_mapping(self.chinfo.shift_charges_horizontal, func_kwargs=dict(dx_0=dx_0), inplace=inplace)


_BINARY_DEVICE = {'add': 1., 'subtract': -1.}


def _ibinary_blockwise(self, func, other, *args, **kwargs):
    """``self = func(self, other)`` block by block (reference :2261).  ``np.add`` / ``np.subtract`` are the axpy kernel;
    any other callable is user code and therefore runs where user code runs -- on host copies of the blocks, with
    blocks missing on one side taken as zeros, followed by one upload."""
    other = other._transpose_same_labels(self._labels)
    if self.rank != other.rank:
        raise ValueError("different rank!")
    for sl, ol in zip(self.legs, other.legs):
        sl.test_equal(ol)
    if np.any(self.qtotal != other.qtotal):
        raise ValueError("Arrays can't have different `qtotal`!")
    name = getattr(func, '__name__', None)
    if name in _BINARY_DEVICE and not args and not kwargs:
        return self.iadd_prefactor_other(_BINARY_DEVICE[name], other)
    mine, theirs = _host_block_dict(self), _host_block_dict(other)
    out = {}
    for key in sorted(set(mine) | set(theirs), key=lambda k: k[::-1]):
        x, y = mine.get(key), theirs.get(key)
        if x is None:
            x = np.zeros_like(y)
        if y is None:
            y = np.zeros_like(x)
        out[key] = np.asarray(func(x, y, *args, **kwargs))
    dtype = np.result_type(*[v.dtype for v in out.values()]) if out else self.dtype
    self._become(_array_from_host_blocks(self.legs, dtype, self.qtotal, self._labels, out))
    return self


def _binary_blockwise(self, func, other, *args, **kwargs):
    return self.copy(deep=True).ibinary_blockwise(func, other, *args, **kwargs)


def _rmul(self, other):
    if np.isscalar(other) or (isinstance(other, np.ndarray) and other.ndim == 0):
        return self.__mul__(other)
    return NotImplemented


def _eq(self, other, eps=1.e-14):
    """Entries equal up to ``eps`` (reference :2466)."""
    if self is other:
        return True
    if not isinstance(other, npc.Array):
        return NotImplemented
    if other.chinfo != self.chinfo:
        raise ValueError("other array has different charges!")
    other = other._transpose_same_labels(self._labels)
    if np.any(self.qtotal != other.qtotal):
        return False
    return bool((self - other).norm(np.inf) < eps)


def _save_hdf5(self, hdf5_saver, h5gr, subpath):
    """HDF5 layout of the reference (:350): ``chinfo, legs, dtype, total_charge, labels, blocks, block_inds`` and the
    attributes ``block_inds_sorted, rank, shape``; the blocks are downloaded for it.  The class is recorded under the REFERENCE's
    module name (``hdf5_io`` writes ``obj.__class__.__module__`` before calling this): a file written on the device must load
    in a plain TeNPy and vice versa (tests/test_hdf5_format.py)."""
    h5gr.attrs['module'] = 'tenpy.linalg.np_conserved'
    hdf5_saver.save(self.chinfo, subpath + 'chinfo')
    hdf5_saver.save(list(self.legs), subpath + 'legs')
    hdf5_saver.save(self.dtype, subpath + 'dtype')
    hdf5_saver.save(self.qtotal, subpath + 'total_charge')
    hdf5_saver.save(list(self._labels), subpath + 'labels')
    hdf5_saver.save([np.array(b) for b in self._data], subpath + 'blocks')      # plain ndarrays (not write-through views)
    hdf5_saver.save(self._qdata, subpath + 'block_inds')
    h5gr.attrs['block_inds_sorted'] = bool(self._qdata_sorted)
    h5gr.attrs['rank'] = self.rank
    h5gr.attrs['shape'] = np.array(self.shape, np.intp)


def _from_hdf5(cls, hdf5_loader, h5gr, subpath):
    obj = cls.__new__(cls)
    hdf5_loader.memorize_load(h5gr, obj)
    legs = hdf5_loader.load(subpath + 'legs')
    dtype = hdf5_loader.load(subpath + 'dtype')
    qtotal = hdf5_loader.load(subpath + 'total_charge')
    labels = hdf5_loader.load(subpath + 'labels')
    blocks = hdf5_loader.load(subpath + 'blocks')
    qdata = np.asarray(hdf5_loader.load(subpath + 'block_inds'), dtype=np.intp)
    cls.__init__(obj, legs, dtype, qtotal, labels)
    if len(blocks):
        flat = np.concatenate([np.asarray(b, dtype=obj.dtype).reshape(-1) for b in blocks])
        obj._set_blocks(qdata, arena=dev.to_device(flat),
                        qdata_sorted=bool(hdf5_loader.get_attr(h5gr, 'block_inds_sorted')))
    obj.test_sanity()
    return obj


# ======================================================================================================================
# module-level functions
# ======================================================================================================================

def _grid_entries(grid):
    grid = np.asarray(grid, dtype=object)
    entries = [(idx, e) for idx, e in np.ndenumerate(grid) if e is not None]
    if len(entries) == 0:
        raise ValueError("No non-trivial entries in grid")
    return grid.shape, entries


def grid_outer(grid, grid_legs, qtotal=None, grid_labels=None):
    """Tensor ``res[i, j, ..., :] = grid[i, j, ...]`` from a grid of equally shaped Arrays, ``None`` = zero
    (reference :3206; how TeNPy builds MPO tensors).  Assembled on the host from the (tiny) entries, one upload."""
    shape, entries = _grid_entries(grid)
    grid_legs = list(grid_legs)
    if len(shape) != len(grid_legs):
        raise ValueError("wrong number of grid_legs")
    if shape != tuple(l.ind_len for l in grid_legs):
        raise ValueError("grid shape incompatible with grid_legs")
    idx0, first = entries[0]
    chinfo = first.chinfo
    dtype = np.result_type(*[e.dtype for _, e in entries])
    labels = ([None] * len(shape) if grid_labels is None else list(grid_labels)) + list(first._labels)
    if qtotal is None:
        q = first.qtotal.copy()
        for i, leg in zip(idx0, grid_legs):
            q = q + leg.get_charge(leg.get_qindex(i)[0])
        qtotal = q
    qtotal = chinfo.make_valid(qtotal)
    legs = grid_legs + list(first.legs)
    res = npc.Array(legs, dtype, qtotal, labels)
    ng = len(shape)
    blocks = {}
    for idx, entry in entries:
        if entry.rank != first.rank:
            raise ValueError("grid entries of different rank")
        for el, fl in zip(entry.legs, first.legs):
            el.test_equal(fl)
        gpos = [leg.get_qindex(i) for i, leg in zip(idx, grid_legs)]
        gq = tuple(p[0] for p in gpos)
        gw = tuple(p[1] for p in gpos)
        for row, blk in zip(entry._qdata, entry._data):
            key = gq + tuple(int(q) for q in row)
            if not np.all(res._get_block_charge(key) == qtotal):
                raise ValueError("wrong charge for assigning self[inds] = other")
            if key not in blocks:
                blocks[key] = np.zeros(res._get_block_shape(key), dtype=dtype)
            blocks[key][gw] = blk
    blocks = {k: v for k, v in blocks.items() if np.any(v)}
    res = _array_from_host_blocks(legs, dtype, qtotal, labels, blocks)
    res.test_sanity()
    return res


def detect_grid_outer_legcharge(grid, grid_legs, qtotal=None, qconj=1, bunch=False):
    """Deduce the one grid leg given as ``None`` from the total charges of the grid entries (reference :3292)."""
    shape, entries = _grid_entries(grid)
    grid_legs = list(grid_legs)
    if len(shape) != len(grid_legs):
        raise ValueError("wrong number of grid_legs")
    if any(s != l.ind_len for s, l in zip(shape, grid_legs) if l is not None):
        raise ValueError("grid shape incompatible with grid_legs")
    chinfo = entries[0][1].chinfo
    unknown = [a for a, l in enumerate(grid_legs) if l is None]
    if len(unknown) != 1:
        raise ValueError("can only derive one grid_leg")
    axis = unknown[0]
    qtotal = chinfo.make_valid(qtotal)
    qflat = [None] * shape[axis]
    for idx, entry in entries:
        q = qtotal - entry.qtotal
        for a, (i, leg) in enumerate(zip(idx, grid_legs)):
            if a != axis:
                q = q - leg.get_charge(leg.get_qindex(i)[0])
        q = chinfo.make_valid(q)
        if qflat[idx[axis]] is None:
            qflat[idx[axis]] = q
        elif np.any(qflat[idx[axis]] != q):
            raise ValueError("different grid entries lead to different charges at index " + str(idx[axis]))
    if any(q is None for q in qflat):
        raise ValueError("can't derive flat charge for all indices:" + str(qflat))
    new = LegCharge.from_qflat(chinfo, chinfo.make_valid(qconj * np.array(qflat)), qconj)
    grid_legs[axis] = new.bunch()[1] if bunch else new
    return grid_legs


def detect_legcharge(flat_array, chargeinfo, legcharges, qtotal=None, qconj=+1, cutoff=None):
    """Deduce the one leg given as ``None`` from the non-zero pattern of a dense host array (reference :3382)."""
    flat_array = np.asarray(flat_array)
    legs = list(legcharges)
    if cutoff is None:
        cutoff = npc.QCUTOFF
    if flat_array.ndim != len(legs):
        raise ValueError("wrong number of grid_legs")
    if any(s != l.ind_len for s, l in zip(flat_array.shape, legs) if l is not None):
        raise ValueError("array shape incompatible with legcharges")
    unknown = [a for a, l in enumerate(legs) if l is None]
    if len(unknown) != 1:
        raise ValueError("can only derive charges for one leg.")
    axis = unknown[0]
    n = flat_array.shape[axis]
    if chargeinfo.qnumber == 0:
        legs[axis] = LegCharge.from_trivial(n, chargeinfo, qconj=qconj)
        return legs
    qtotal = chargeinfo.make_valid(qtotal)
    known = legs[:axis] + legs[axis + 1:]
    qflat = np.empty((n, chargeinfo.qnumber), dtype=QTYPE)
    for i in range(n):
        qflat[i] = npc.detect_qtotal(np.take(flat_array, i, axis=axis), known, cutoff)
    qflat = chargeinfo.make_valid((qtotal - qflat) * qconj)
    legs[axis] = LegCharge.from_qflat(chargeinfo, qflat, qconj).bunch()[1]
    return legs


def grid_concat(grid, axes, copy=True):
    """Multi-dimensional concatenation (like ``np.block`` with uniform blocking) of a grid of Arrays along the legs
    ``axes``; ``None`` entries stand for zeros (reference :3099).  Recursion over the grid dimensions on top of the device
    :func:`concatenate`."""
    grid = np.asarray(grid, dtype=object)
    if grid.ndim < 1 or grid.ndim != len(axes):
        raise ValueError("grid has wrong dimension")
    if grid.ndim == 1:
        if any(g is None for g in grid):
            raise ValueError("`None` entry in 1D grid")
        return npc.concatenate(list(grid), axes[0], copy)
    if any(g is None for g in grid.flat):
        grid = grid.copy()
        template = next(g for g in grid.flat if g is not None)
        ax_idx = template.get_leg_indices(axes)
        # the leg a `None` entry must have along grid dimension d at position i: that of any present entry there
        legs_along = []
        for d in range(grid.ndim):
            per_pos = []
            for i in range(grid.shape[d]):
                present = next((g for g in np.take(grid, i, axis=d).flat if g is not None), None)
                if present is None:
                    raise ValueError("Full row/column with only `None` entries")
                per_pos.append(present.get_leg(axes[d]))
            legs_along.append(per_pos)
        for idx, entry in np.ndenumerate(grid):
            if entry is None:
                legs = list(template.legs)
                for d, i in enumerate(idx):
                    legs[ax_idx[d]] = legs_along[d][i]
                grid[idx] = npc.zeros(legs, template.dtype, template.qtotal, template.get_leg_labels())
        axes = ax_idx
    return _grid_concat_rec(grid, list(axes), copy)


def _grid_concat_rec(grid, axes, copy):
    if grid.ndim == 1:
        return npc.concatenate(list(grid), axes[0], copy)
    return npc.concatenate([_grid_concat_rec(sub, axes[1:], copy) for sub in grid], axes[0], copy=False)


# ---- orthonormal completion on the device ------------------------------------------------------------------------------

def complement_columns(Q):
    """For an isometry ``Q`` (blocked matrix, orthonormal columns inside every charge block): an isometry whose columns
    span the orthogonal complement of ``range(Q)``, sector by sector.

    Device route with existing entry points only: the complementary projector ``P = 1 - Q Q^dagger`` (block GEMM) has
    singular values 1 (multiplicity m - k) and 0; its block SVD returns them sorted, so the left singular vectors of the
    unit singular values are the completion.  Sectors of the first leg in which ``Q`` has no block get the identity,
    as in the reference (np_conserved.py:4244-4262, :4326-4366).

    Returns ``(C, kept_qind)``: ``C`` with legs ``[Q.legs[0], new leg]`` (``qtotal`` = that of ``Q``; new leg: one sector
    per non-empty complement, ``qconj = -1``-side handled by the caller) and the qindices of ``Q.legs[0]`` they belong to.
    """
    left = Q.legs[0]
    eye = npc.diag(1., left, dtype=Q.dtype)
    P = eye
    if Q.stored_blocks:
        Qn = Q.copy(deep=False).idrop_labels()
        QQ = npc.tensordot(Qn, Qn.conj(), axes=[1, 1])
        P = eye.copy(deep=True).iadd_prefactor_other(-1., QQ)
    U, S, _ = npc.svd(P)
    keep = S > 0.5
    if not np.any(keep):
        return None
    U.iproject(keep, 1)
    return U


def attach(module):
    """Called at the end of ``np_conserved``: bind the methods above to ``Array`` and export the functions."""
    global npc
    npc = module
    A = module.Array
    A.__str__ = _str
    A.__iter__ = _iter
    A.__getitem__ = _getitem
    A.__setitem__ = _setitem
    A.__rmul__ = _rmul
    A.__eq__ = _eq
    A.__hash__ = None
    A._iter_all_blocks = _iter_all_blocks
    A._get_block_charge = _get_block_charge
    A._get_block_slices = _get_block_slices
    A._get_block_shape = _get_block_shape
    A.get_block = _get_block
    A._pre_indexing = _pre_indexing
    A._advanced_getitem = _advanced_getitem
    A.permute = _permute
    A.extend = _extend
    A.add_charge = _add_charge
    A.drop_charge = _drop_charge
    A.change_charge = _change_charge
    A.apply_charge_mapping = _apply_charge_mapping
    A.shift_charges = _shift_charges
    A.shift_charges_horizontal = _shift_charges_horizontal
    A.ibinary_blockwise = _ibinary_blockwise
    A.binary_blockwise = _binary_blockwise
    A.save_hdf5 = _save_hdf5
    A.from_hdf5 = classmethod(_from_hdf5)
    for f in (grid_outer, grid_concat, detect_grid_outer_legcharge, detect_legcharge):
        setattr(module, f.__name__, f)
