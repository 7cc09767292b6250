"""Charge bookkeeping on the host: :class:`ChargeInfo`, :class:`LegCharge`, :class:`LegPipe`.

Mirrors the public interface of ``tenpy/linalg/charges.py`` (reference ``ChargeInfo`` :39, ``LegCharge``
:552, ``LegPipe`` :1444).  Everything here is integer (int64) host work and is kept bit-exact with the
reference: same ``slices`` / ``charges`` / ``q_map`` arrays for the same inputs (checked in
``tests/test_charges.py`` against golden fixtures generated from the reference).
No block data lives here; see :mod:`tenpy_amd.linalg.np_conserved` for the device-resident ``Array``.
"""
import copy as _copy

import numpy as np

__all__ = ['ChargeInfo', 'DipolarChargeInfo', 'LegCharge', 'LegPipe', 'QTYPE']

QTYPE = np.int64  # reference charges.py:36 / _npc_helper.pyx:77


def _inverse_permutation(perm):
    inv = np.empty(len(perm), dtype=np.intp)
    inv[perm] = np.arange(len(perm), dtype=np.intp)
    return inv


class ChargeInfo:
    """Meta-data of the charges: number of charges and the modulus of each (``1`` means U(1)).

    Same semantics as the reference (charges.py:39-372): ``make_valid`` reduces charges with a
    non-negative (python-style) modulo where ``mod != 1``; ``check_valid`` tests that.
    """

    trivial_shift = True       # translations act trivially on these charges (reference charges.py:82)

    def __init__(self, mod=[], names=None):
        mod = np.array(mod, dtype=QTYPE).reshape(-1)
        self._mod = mod
        self._qnumber = len(mod)
        self._mask = (mod != 1)
        self._mod_masked = mod[self._mask].copy()
        if names is None:
            names = [''] * self._qnumber
        self.names = [str(n) for n in names]
        self.test_sanity()

    @classmethod
    def add(cls, chinfos):
        """Concatenate the charges of several ChargeInfo."""
        mod = np.concatenate([ci.mod for ci in chinfos]) if len(chinfos) else []
        names = sum([list(ci.names) for ci in chinfos], [])
        return cls(mod, names)

    @classmethod
    def drop(cls, chinfo, charge=None):
        """Remove one or more (or all, for ``charge=None``) charges."""
        if charge is None:
            return cls()
        if isinstance(charge, (str, int, np.integer)):
            charge = [charge]
        drop = [chinfo.names.index(c) if isinstance(c, str) else int(c) for c in charge]
        keep = [i for i in range(chinfo.qnumber) if i not in drop]
        return cls([chinfo.mod[i] for i in keep], [chinfo.names[i] for i in keep])

    @classmethod
    def change(cls, chinfo, charge, new_qmod, new_name=''):
        """Same charges, but charge number / name ``charge`` gets modulus ``new_qmod`` (reference :215)."""
        idx = chinfo.names.index(charge) if isinstance(charge, str) else int(charge)
        mod = np.array(chinfo.mod, dtype=QTYPE)
        mod[idx] = new_qmod
        names = list(chinfo.names)
        names[idx] = new_name
        return cls(mod, names)

    def shift_charges(self, charges, dx):
        """Action of a lattice translation by ``dx`` on charge values: none for ordinary charges (reference :306)."""
        return charges

    def shift_charges_horizontal(self, charges, dx_0):
        return charges

    def save_hdf5(self, hdf5_saver, h5gr, subpath):
        """HDF5 layout of the reference (charges.py:111): attribute ``num_charges``, datasets ``U1_ZN`` and ``names``."""
        h5gr.attrs['module'] = 'tenpy.linalg.charges'     # the file names the reference's module (hdf5_io ATTR_MODULE), not the mirror
        h5gr.attrs['num_charges'] = self._qnumber
        hdf5_saver.save(self._mod, subpath + 'U1_ZN')
        hdf5_saver.save(self.names, subpath + 'names')

    @classmethod
    def from_hdf5(cls, hdf5_loader, h5gr, subpath):
        obj = cls.__new__(cls)
        hdf5_loader.memorize_load(h5gr, obj)
        mod = np.asarray(hdf5_loader.load(subpath + 'U1_ZN'), dtype=QTYPE)
        names = hdf5_loader.load(subpath + 'names') if 'names' in h5gr else [''] * len(mod)
        obj.__setstate__((len(mod), mod, names))
        return obj

    def test_sanity(self):
        if self._mod.ndim != 1 or len(self.names) != self._qnumber:
            raise ValueError("mod has wrong shape / names wrong length")
        if np.any(self._mod <= 0):
            raise ValueError("mod should be > 0")

    @property
    def qnumber(self):
        return self._qnumber

    @property
    def mod(self):
        return self._mod

    def make_valid(self, charges=None):
        """Charges taken modulo ``mod`` (``mod == 1``: unchanged); ``None`` gives the zero charge."""
        if charges is None:
            return np.zeros((self._qnumber,), dtype=QTYPE)
        charges = np.array(charges, dtype=QTYPE)  # copy
        if self._mod_masked.size:
            charges[..., self._mask] = np.mod(charges[..., self._mask], self._mod_masked)
        return charges

    def check_valid(self, charges):
        charges = np.asarray(charges, dtype=QTYPE)[..., self._mask]
        return bool(np.all(np.logical_and(0 <= charges, charges < self._mod_masked)))

    def __repr__(self):
        return "ChargeInfo({0!s}, {1!s})".format(list(self.mod), self.names)

    def __eq__(self, other):
        if self is other:
            return True
        if not isinstance(other, ChargeInfo):
            return NotImplemented
        if not np.array_equal(self.mod, other.mod):
            return False
        for l, r in zip(self.names, other.names):
            if r != l and l != '' and r != '':
                return False
        return True

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __getstate__(self):
        return (self._qnumber, self._mod, self.names)

    def __setstate__(self, state):
        qnumber, mod, names = state
        ChargeInfo.__init__(self, mod, names)


def _is_subgroup_by_qmod(qmod1, qmod2):
    """Whether the group with modulus ``qmod1`` (1 = U(1)) is a subgroup of the one with ``qmod2`` (reference :1887)."""
    if qmod2 == 1:
        return True
    return qmod1 != 1 and qmod2 % qmod1 == 0


class DipolarChargeInfo(ChargeInfo):
    """ChargeInfo in which some charges are dipole moments ``p = x * q`` of other charges (reference :375-549).

    Host-side integer bookkeeping only; what differs from :class:`ChargeInfo` is how charge values move under a
    lattice translation: ``p -> p + dx[dim] * q`` for every (charge, dipole, dim) triple.
    """

    trivial_shift = False

    def __init__(self, mod=[], names=None, charge_idcs=[], dipole_idcs=[], dipole_dims=None):
        charge_idcs, dipole_idcs = list(charge_idcs), list(dipole_idcs)
        dipole_dims = [0] * len(dipole_idcs) if dipole_dims is None else list(dipole_dims)
        nq = len(mod)
        for what, idcs in (('charge_idcs', charge_idcs), ('dipole_idcs', dipole_idcs)):
            for n, i in enumerate(idcs):
                if not 0 <= i < nq:
                    raise ValueError("{0}[{1:d}] out of bounds".format(what, n))
        if set(charge_idcs) & set(dipole_idcs):
            raise ValueError("dipole_idcs and charge_idcs must be disjoint.")
        for ci, di, dim in zip(charge_idcs, dipole_idcs, dipole_dims):
            if dim > 0 and mod[di] == 1:
                raise ValueError("Can not conserve U(1) dipole charge (qmod==1) along dipole_dim > 0.")
            if not _is_subgroup_by_qmod(mod[di], mod[ci]):
                raise ValueError("Dipole charge can not have qmod={0} if underlying charge has qmod={1}. "
                                 "(Not a subgroup)".format(mod[di], mod[ci]))
        self._charge_idcs, self._dipole_idcs, self._dipole_dims = charge_idcs, dipole_idcs, dipole_dims
        ChargeInfo.__init__(self, mod, names)

    def _triples(self):
        return zip(self._charge_idcs, self._dipole_idcs, self._dipole_dims)

    def shift_charges(self, charges, dx):
        if dx[-1] != 0:
            raise NotImplementedError("translation between different sites of the unit cell")
        res = np.array(charges, dtype=QTYPE)
        for ci, di, dim in self._triples():
            res[..., di] += dx[dim] * res[..., ci]
        return self.make_valid(res)

    def shift_charges_horizontal(self, charges, dx_0):
        res = np.array(charges, dtype=QTYPE)
        for ci, di, dim in self._triples():
            if dim == 0:
                res[..., di] += dx_0 * res[..., ci]
        return self.make_valid(res)

    def test_sanity(self):
        n = len(self._charge_idcs)
        if len(self._dipole_idcs) != n or len(self._dipole_dims) != n:
            raise ValueError("dipole_idcs / dipole_dims have wrong length")
        if len(set(self._dipole_idcs)) != n:
            raise ValueError("duplicates in dipole_idcs")
        ChargeInfo.test_sanity(self)

    def __repr__(self):
        return "DipolarChargeInfo({0!s}, {1!s}, {2!s}, {3!s}, {4!s})".format(
            list(self.mod), self.names, self._charge_idcs, self._dipole_idcs, self._dipole_dims)

    def __eq__(self, other):
        if self is other:
            return True
        if not isinstance(other, DipolarChargeInfo):
            return False
        return ChargeInfo.__eq__(self, other) is True and \
            (self._charge_idcs, self._dipole_idcs, self._dipole_dims) == \
            (other._charge_idcs, other._dipole_idcs, other._dipole_dims)

    def __getstate__(self):
        return (ChargeInfo.__getstate__(self), (self._charge_idcs, self._dipole_idcs, self._dipole_dims))

    def __setstate__(self, state):
        base, (self._charge_idcs, self._dipole_idcs, self._dipole_dims) = state
        ChargeInfo.__setstate__(self, base)

    def save_hdf5(self, hdf5_saver, h5gr, subpath):
        ChargeInfo.save_hdf5(self, hdf5_saver, h5gr, subpath)
        for key in ('charge_idcs', 'dipole_idcs', 'dipole_dims'):
            hdf5_saver.save(getattr(self, '_' + key), subpath + key)

    @classmethod
    def from_hdf5(cls, hdf5_loader, h5gr, subpath):
        obj = cls.__new__(cls)
        hdf5_loader.memorize_load(h5gr, obj)
        mod = np.asarray(hdf5_loader.load(subpath + 'U1_ZN'), dtype=QTYPE)
        names = hdf5_loader.load(subpath + 'names') if 'names' in h5gr else [''] * len(mod)
        rest = tuple(hdf5_loader.load(subpath + key) for key in ('charge_idcs', 'dipole_idcs', 'dipole_dims'))
        obj.__setstate__(((len(mod), mod, names), rest))
        obj.test_sanity()
        return obj


class LegCharge:
    """Charge data of one tensor leg: contiguous blocks ``slices[q]:slices[q+1]`` carry ``charges[q]``.

    Attributes and methods follow the reference (charges.py:552-1440): ``ind_len``, ``block_number``,
    ``chinfo``, ``slices`` (intp), ``charges`` (int64, block_number x qnumber), ``qconj``,
    ``sorted``, ``bunched``.  Instances are treated as immutable and shared between Arrays.
    """

    def __init__(self, chargeinfo, slices, charges, qconj=1):
        self.chinfo = chargeinfo
        self.slices = np.array(slices, dtype=np.intp)
        self.charges = np.array(charges, dtype=QTYPE).reshape(len(self.slices) - 1, chargeinfo.qnumber)
        self.qconj = int(qconj)
        self.sorted = False
        self.bunched = False
        self.ind_len = int(self.slices[-1])
        self.block_number = self.charges.shape[0]
        self._bsizes = None
        LegCharge.test_sanity(self)

    def copy(self):
        res = _copy.copy(self)
        res._bsizes = None
        return res

    # ---- constructors ----------------------------------------------------------------------------
    @classmethod
    def from_trivial(cls, ind_len, chargeinfo=None, qconj=1):
        if chargeinfo is None:
            chargeinfo = ChargeInfo()
            charges = [[]]
        else:
            charges = [[0] * chargeinfo.qnumber]
        res = cls(chargeinfo, [0, ind_len], charges, qconj)
        res.sorted = res.bunched = True
        return res

    @classmethod
    def from_qflat(cls, chargeinfo, qflat, qconj=1):
        """One block per *index* (neither sorted nor bunched, like the reference :768)."""
        qflat = np.array(qflat, dtype=QTYPE)
        if qflat.ndim == 1 and chargeinfo.qnumber == 1:
            qflat = qflat.reshape(-1, 1)
        ind_len = qflat.shape[0]
        qflat = qflat.reshape(ind_len, chargeinfo.qnumber)
        res = cls(chargeinfo, np.arange(ind_len + 1), qflat, qconj)
        res.sorted = res.is_sorted()
        res.bunched = res.is_bunched()
        return res

    @classmethod
    def from_qind(cls, chargeinfo, slices, charges, qconj=1):
        res = cls(chargeinfo, slices, charges, qconj)
        res.sorted = res.is_sorted()
        res.bunched = res.is_bunched()
        return res

    @classmethod
    def from_qdict(cls, chargeinfo, qdict, qconj=1):
        items = sorted(((sl.start, sl.stop, ch) for ch, sl in qdict.items()), key=lambda t: t[0])
        slices = [it[0] for it in items] + [items[-1][1]]
        charges = [it[2] for it in items]
        res = cls(chargeinfo, slices, charges, qconj)
        res.sorted = True
        res.bunched = res.is_bunched()
        return res

    @classmethod
    def from_add_charge(cls, legs, chargeinfo=None):
        """Leg carrying the charges of all ``legs`` side by side (reference :843): the block boundaries are the union
        of the boundaries of the given legs; neither sorted nor bunched."""
        legs = list(legs)
        chinfo = ChargeInfo.add([leg.chinfo for leg in legs])
        if chargeinfo is not None:
            assert chinfo == chargeinfo
            chinfo = chargeinfo
        if any(leg.ind_len != legs[0].ind_len for leg in legs):
            raise ValueError("different length")
        if any(leg.qconj != legs[0].qconj for leg in legs):
            raise ValueError("different qconj")
        ind_len = legs[0].ind_len
        ptr = [0] * len(legs)                       # current block of every leg
        cuts, rows = [0], []
        while True:
            rows.append(np.concatenate([leg.charges[q] for leg, q in zip(legs, ptr)]) if legs[0].chinfo is not None else [])
            ends = [int(leg.slices[q + 1]) for leg, q in zip(legs, ptr)]
            cut = min(ends)
            if cut >= ind_len:
                break
            ptr = [q + 1 if e == cut else q for q, e in zip(ptr, ends)]
            cuts.append(cut)
        cuts.append(ind_len)
        return cls.from_qind(chinfo, cuts, np.array(rows, dtype=QTYPE).reshape(len(rows), chinfo.qnumber), legs[0].qconj)

    @classmethod
    def from_drop_charge(cls, leg, charge=None, chargeinfo=None):
        """Leg without charge number / name ``charge`` (``None``: without any charge), reference :896."""
        if charge is None:
            return cls.from_trivial(leg.ind_len, chargeinfo, leg.qconj)
        chinfo = ChargeInfo.drop(leg.chinfo, charge)
        if chargeinfo is not None:
            assert chinfo == chargeinfo
            chinfo = chargeinfo
        idx = leg.chinfo.names.index(charge) if isinstance(charge, str) else int(charge)
        return cls.from_qind(chinfo, leg.slices, np.delete(leg.charges, idx, axis=1), leg.qconj)

    @classmethod
    def from_change_charge(cls, leg, charge, new_qmod, new_name='', chargeinfo=None):
        """Leg whose charge ``charge`` is taken modulo ``new_qmod`` instead (reference :926)."""
        chinfo = ChargeInfo.change(leg.chinfo, charge, new_qmod, new_name)
        if chargeinfo is not None:
            assert chinfo == chargeinfo
            chinfo = chargeinfo
        return cls.from_qind(chinfo, leg.slices, chinfo.make_valid(leg.charges), leg.qconj)

    def apply_charge_mapping(self, map_func, func_args=(), func_kwargs={}):
        """Shallow copy with ``charges = map_func(charges, ...)`` (reference :1010)."""
        res = self.copy()
        res.charges = map_func(self.charges, *func_args, **func_kwargs)
        res.sorted = res.bunched = False
        return res

    def _slice_start_stop(self):
        return zip(self.slices[:-1], self.slices[1:])

    # ---- HDF5 (layout of the reference, charges.py:649-755) -------------------------------------------------
    def save_hdf5(self, hdf5_saver, h5gr, subpath):
        fmt = hdf5_saver.format_selection.get('LegCharge', 'blocks')
        h5gr.attrs['module'] = 'tenpy.linalg.charges'
        h5gr.attrs['format'] = fmt
        h5gr.attrs['ind_len'] = self.ind_len
        h5gr.attrs['qconj'] = self.qconj
        hdf5_saver.save(self.chinfo, subpath + 'chinfo')
        if fmt in ('blocks', 'compact'):
            for key in ('block_number', 'sorted', 'bunched'):
                h5gr.attrs[key] = getattr(self, key)
            if fmt == 'blocks':
                hdf5_saver.save(self.slices, subpath + 'slices')
                hdf5_saver.save(self.charges, subpath + 'charges')
            else:
                table = np.hstack([self.slices[:-1, np.newaxis], self.slices[1:, np.newaxis], self.charges])
                hdf5_saver.save(table, subpath + 'blockcharges')
        elif fmt == 'flat':
            hdf5_saver.save(self.to_qflat(), subpath + 'charges')
        else:
            raise ValueError("Unknown format")

    @classmethod
    def from_hdf5(cls, hdf5_loader, h5gr, subpath):
        obj = cls.__new__(cls)
        hdf5_loader.memorize_load(h5gr, obj)
        fmt = hdf5_loader.get_attr(h5gr, 'format')
        ind_len = hdf5_loader.get_attr(h5gr, 'ind_len')
        qconj = hdf5_loader.get_attr(h5gr, 'qconj')
        chinfo = hdf5_loader.load(subpath + 'chinfo')
        if fmt == 'flat':
            charges = np.asarray(hdf5_loader.load(subpath + 'charges'), dtype=QTYPE)
            obj.__setstate__((ind_len, ind_len, chinfo, np.arange(ind_len + 1, dtype=np.intp), charges, qconj, False, False))
            obj.sorted, obj.bunched = obj.is_sorted(), obj.is_bunched()
        elif fmt in ('blocks', 'compact'):
            nblk = hdf5_loader.get_attr(h5gr, 'block_number')
            flags = [hdf5_loader.get_attr(h5gr, key) for key in ('sorted', 'bunched')]
            if fmt == 'blocks':
                slices = np.asarray(hdf5_loader.load(subpath + 'slices'), dtype=np.intp)
                charges = np.asarray(hdf5_loader.load(subpath + 'charges'), dtype=QTYPE)
            else:
                table = hdf5_loader.load(subpath + 'blockcharges')
                slices = np.concatenate([table[:, 0], table[-1:, 1]]).astype(np.intp)
                charges = np.ascontiguousarray(table[:, 2:], dtype=QTYPE)
            obj.__setstate__((ind_len, nblk, chinfo, slices, charges.reshape(nblk, -1), qconj, flags[0], flags[1]))
        else:
            raise ValueError("Unknown format")
        obj.test_sanity()
        return obj

    # ---- checks ------------------------------------------------------------------------------------
    def test_sanity(self):
        sl, ch = self.slices, self.charges
        if sl.ndim != 1 or sl.shape[0] != self.block_number + 1:
            raise ValueError("wrong len of `slices`")
        if sl[0] != 0 or np.any(sl[1:] < sl[:-1]):
            raise ValueError("slices should start at 0 and be non-decreasing")
        if ch.shape != (self.block_number, self.chinfo.qnumber):
            raise ValueError("charges have wrong shape")
        if not self.chinfo.check_valid(ch):
            raise ValueError("charges invalid for " + str(self.chinfo) + "\n" + str(self))
        if self.qconj not in (-1, 1):
            raise ValueError("qconj has invalid value != +-1 :" + str(self.qconj))

    def conj(self):
        """Shallow copy with opposite ``qconj`` (charges unchanged)."""
        res = _copy.copy(self)
        res.qconj = -self.qconj
        return res

    def flip_charges_qconj(self):
        res = _copy.copy(self)
        res.qconj = -self.qconj
        res.charges = self.chinfo.make_valid(-self.charges)
        res.sorted = res.is_sorted() if self.sorted else False
        return res

    def to_qflat(self):
        qflat = np.empty((self.ind_len, self.chinfo.qnumber), dtype=QTYPE)
        for q in range(self.block_number):
            qflat[self.slices[q]:self.slices[q + 1]] = self.charges[q]
        return qflat

    def to_qdict(self):
        return {tuple(int(c) for c in ch): slice(int(b), int(e))
                for ch, b, e in zip(self.charges, self.slices[:-1], self.slices[1:])}

    def is_blocked(self):
        """True if qindices map 1:1 to charge values (reference :1053): sorted+bunched, or all charges distinct."""
        if self.sorted and self.bunched:
            return True
        return len({tuple(c) for c in self.charges.tolist()}) == self.block_number

    def is_sorted(self):
        if self.chinfo.qnumber == 0:
            return True
        ch = self.charges
        return bool(np.all(np.lexsort(ch.T) == np.arange(len(ch))))

    def is_bunched(self):
        return len(_find_row_differences(self.charges)) == self.block_number + 1

    def _same_charges(self, other, sign):
        """``self.charges * self.qconj == sign * other.charges * other.qconj`` modulo the moduli of the charges."""
        if self.charges is other.charges and self.qconj == sign * other.qconj and \
                (self.slices is other.slices or np.array_equal(self.slices, other.slices)):
            return True
        if self.block_number != other.block_number or not np.array_equal(self.slices, other.slices):
            return False
        mv = self.chinfo.make_valid
        return bool(np.array_equal(mv(self.charges * self.qconj), mv(other.charges * (sign * other.qconj))))

    def test_contractible(self, other):
        """Raise ValueError unless ``self`` can be contracted with ``other``: equal ChargeInfo and slices, charges equal
        up to the opposite sign convention (reference :1071)."""
        if self.chinfo is not other.chinfo and self.chinfo != other.chinfo:
            raise ValueError(''.join(["incompatible ChargeInfo\n", str(self.chinfo), str(other.chinfo)]))
        if not self._same_charges(other, -1):
            raise ValueError("incompatible LegCharge\nself\n" + str(self) + "\nother (conjugated)\n" + str(other.conj()))

    def test_equal(self, other):
        """Raise ValueError unless slices and charges (with ``qconj``, modulo the charge moduli) agree (reference :1114)."""
        if self is other:
            return
        if self.chinfo is not other.chinfo and self.chinfo != other.chinfo:
            raise ValueError(''.join(["incompatible ChargeInfo\n", str(self.chinfo), str(other.chinfo)]))
        if not self._same_charges(other, +1):
            raise ValueError("incompatible LegCharge\nself\n" + str(self) + "\nother\n" + str(other))

    def __eq__(self, other):
        if self is other:
            return True
        if not isinstance(other, LegCharge):
            return NotImplemented
        try:
            self.test_equal(other)
        except ValueError:
            return False
        return True

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    # ---- access ------------------------------------------------------------------------------------
    def get_block_sizes(self):
        if self._bsizes is None:
            self._bsizes = self.slices[1:] - self.slices[:-1]
        return self._bsizes

    def get_slice(self, qindex):
        return slice(int(self.slices[qindex]), int(self.slices[qindex + 1]))

    def get_qindex(self, flat_index):
        if flat_index < 0:
            flat_index += self.ind_len
            if flat_index < 0:
                raise IndexError("flat index {0:d} too negative for leg with ind_len {1:d}".format(
                    flat_index - self.ind_len, self.ind_len))
        elif flat_index >= self.ind_len:
            raise IndexError("flat index {0:d} too large for leg with ind_len {1:d}".format(flat_index, self.ind_len))
        qind = int(np.searchsorted(self.slices, flat_index, side='right')) - 1
        return qind, int(flat_index - self.slices[qind])

    def get_qindex_of_charges(self, charges):
        charges = self.chinfo.make_valid(charges)
        hit = np.nonzero(np.all(self.charges == charges, axis=1))[0]
        if len(hit) != 1:
            raise ValueError("charges not found exactly once (leg not blocked?)")
        return int(hit[0])

    def get_charge(self, qindex):
        return self.charges[qindex] * self.qconj

    def charge_sectors(self):
        """Unique rows of ``charges`` in the ``np.lexsort(charges.T)`` order of a sorted leg (reference :1368; the LAST charge is
        the primary key, unlike ``np.unique(axis=0)``)."""
        charges = self.charges
        if self.block_number == 0:
            return charges.copy()
        if charges.shape[1] == 0:               # no conserved charge: one (empty) sector
            return charges[:1].copy()
        charges = charges[np.lexsort(charges.T), :]
        keep = np.concatenate([[True], np.any(charges[1:] != charges[:-1], axis=1)])
        return charges[keep]

    # ---- sort / bunch / project ----------------------------------------------------------------------
    def sort(self, bunch=True):
        """Return ``(perm_qind, sorted_leg)`` with blocks ordered by ``np.lexsort(charges.T)``."""
        if self.sorted and ((not bunch) or self.bunched):
            return np.arange(self.block_number, dtype=np.intp), self
        perm = np.lexsort(self.charges.T).astype(np.intp)
        res = _copy.copy(self)
        res._bsizes = None
        res.charges = self.charges[perm]
        bs = self.get_block_sizes()[perm]
        res.slices = np.concatenate([[0], np.cumsum(bs)]).astype(np.intp)
        res.sorted = True
        res.bunched = False
        if bunch:
            _, res = res.bunch()
        return perm, res

    def bunch(self):
        """Merge neighbouring blocks with equal charge; returns ``(idx, bunched_leg)``."""
        if self.bunched:
            return np.arange(self.block_number + 1, dtype=np.intp), self
        idx = _find_row_differences(self.charges)
        res = _copy.copy(self)
        res._bsizes = None
        res.charges = self.charges[idx[:-1]]
        res.slices = self.slices[idx]
        res.block_number = len(idx) - 1
        res.bunched = True
        return idx, res

    def project(self, mask):
        """Keep only the indices selected by the boolean ``mask``: returns
        ``(map_qind, block_masks, projected_leg)`` (reference :1304)."""
        mask = np.asarray(mask, dtype=np.bool_)
        res = _copy.copy(self)
        res._bsizes = None
        block_masks = [mask[b:e] for b, e in zip(self.slices[:-1], self.slices[1:])]
        cs = np.concatenate([[0], np.cumsum(mask, dtype=np.intp)])       # (one pass instead of a np.sum per sector: 0.2 - 0.6 ms per call on a ladder's bond leg)
        new_sizes = cs[self.slices[1:]] - cs[self.slices[:-1]]
        keep = new_sizes > 0
        block_masks = [bm for bm, k in zip(block_masks, keep) if k]
        new_sizes = new_sizes[keep]
        map_qind = np.full(self.block_number, -1, dtype=np.intp)
        map_qind[keep] = np.arange(len(block_masks), dtype=np.intp)
        res.charges = self.charges[keep]
        res.slices = np.concatenate([[0], np.cumsum(new_sizes)]).astype(np.intp)
        res.block_number = len(block_masks)
        res.ind_len = int(res.slices[-1])
        return map_qind, block_masks, res

    def extend(self, extra):
        if not isinstance(extra, LegCharge):
            extra = LegCharge.from_trivial(int(extra), self.chinfo, self.qconj)
        elif extra.chinfo != self.chinfo or extra.qconj != self.qconj:
            raise ValueError("incompatible `extra` leg")
        return LegCharge(self.chinfo, np.concatenate([self.slices, self.ind_len + extra.slices[1:]]),
                         np.concatenate([self.charges, extra.charges]), self.qconj)

    def perm_flat_from_perm_qind(self, perm_qind):
        begs, ends = self.slices[:-1][perm_qind], self.slices[1:][perm_qind]
        parts = [np.arange(b, e, dtype=np.intp) for b, e in zip(begs, ends)]
        return np.concatenate(parts) if parts else np.zeros(0, np.intp)

    def perm_qind_from_perm_flat(self, perm_flat):
        """Block permutation belonging to a flat permutation that only moves whole blocks."""
        perm_flat = np.asarray(perm_flat)
        sizes = self.get_block_sizes()
        res, pos = [], 0
        while pos < self.ind_len:
            q = int(np.searchsorted(self.slices, perm_flat[pos], side='right')) - 1
            if perm_flat[pos] != self.slices[q] or \
                    not np.array_equal(perm_flat[pos:pos + sizes[q]], np.arange(self.slices[q], self.slices[q + 1])):
                raise ValueError("Permutation mixes qind")
            res.append(q)
            pos += int(sizes[q])
        return np.array(res, dtype=np.intp)

    def __str__(self):
        qconj = " {0:+d}\n".format(self.qconj)
        slices = '\n'.join(str(s) for s in self.slices)
        return qconj + slices + "\n" + str(self.charges)

    def __repr__(self):
        return "LegCharge({0!r}, qconj={1:+d},\n{2!r}, {3!r})".format(self.chinfo, self.qconj, self.slices, self.charges)

    def _set_charges(self, charges):
        self.charges = charges
        self.block_number = charges.shape[0]

    def _set_slices(self, slices):
        self.slices = slices
        self.ind_len = int(slices[-1])
        self._bsizes = None

    def _set_block_sizes(self, block_sizes):
        self._set_slices(np.concatenate([[0], np.cumsum(block_sizes)]).astype(np.intp))

    def __getstate__(self):
        return (self.ind_len, self.block_number, self.chinfo, self.slices, self.charges, self.qconj, self.sorted,
                self.bunched)

    def __setstate__(self, state):
        (self.ind_len, self.block_number, self.chinfo, self.slices, self.charges, self.qconj, self.sorted,
         self.bunched) = state
        self._bsizes = None


class LegPipe(LegCharge):
    """Fusion of several legs into one (reference charges.py:1444-1885).

    ``q_map`` rows are ``[b_j, b_{j+1}, I_s, i_1, ..., i_nlegs]``: the block of the incoming qindex tuple
    ``(i_1..i_n)`` is stored in rows/cols ``b_j:b_{j+1}`` of the fused block ``I_s`` (C-order inside).
    ``q_map`` is sorted by ``I_s`` first (when ``sort``), then by the ``i`` (first leg slowest).
    """

    def __init__(self, legs, qconj=1, sort=True, bunch=True):
        legs = tuple(legs)
        chinfo = legs[0].chinfo
        LegCharge.__init__(self, chinfo, [0, 1], [[0] * chinfo.qnumber], qconj)
        self.legs = legs
        self.nlegs = len(legs)
        self.subshape = tuple(l.ind_len for l in legs)
        self.subqshape = tuple(l.block_number for l in legs)
        self.q_map = None
        self.q_map_slices = None
        self._fuse(sort, bunch)
        self.test_sanity()

    def _content_key(self):
        """Hashable key of everything a reshaping plan takes from this pipe (``q_map``, block structure); memoised: pipes are
        never modified after construction (``conj`` etc. make new ones)."""
        k = self.__dict__.get('_ckey')
        if k is None or k[0] is not self.q_map:
            from hashlib import blake2b
            h = blake2b(digest_size=16)      # (128-bit digest instead of Python's 64-bit hash(): the key is the plan's only identity)
            for part in (np.ascontiguousarray(self.q_map), np.ascontiguousarray(self.q_map_slices), np.ascontiguousarray(self.slices),
                         np.array((int(self.qconj),) + tuple(self.subqshape) + tuple(self.subshape), dtype=np.int64)):
                h.update(np.array(part.shape, dtype=np.int64).tobytes())
                h.update(part.tobytes())
            k = self.__dict__['_ckey'] = (self.q_map, h.digest())
        return k[1]

    def _fuse(self, sort, bunch):
        nlegs, qnumber = self.nlegs, self.chinfo.qnumber
        nq = self.subqshape
        nblocks = int(np.prod(nq))
        # C-order strides over the grid of incoming qindices
        strides = np.ones(nlegs, dtype=np.intp)
        for a in range(nlegs - 2, -1, -1):
            strides[a] = strides[a + 1] * nq[a + 1]
        self._strides = strides
        q_map = np.empty((nblocks, 3 + nlegs), dtype=np.intp)
        flat = np.arange(nblocks, dtype=np.intp)
        sizes = np.ones(nblocks, dtype=np.intp)
        charges = np.zeros((nblocks, qnumber), dtype=QTYPE)
        for a, leg in enumerate(self.legs):
            qi = (flat // strides[a]) % nq[a]
            q_map[:, 3 + a] = qi
            sizes *= leg.get_block_sizes()[qi]
            if qnumber:
                charges += (self.qconj * leg.qconj) * leg.charges[qi]
        if qnumber:
            charges = self.chinfo.make_valid(charges)
        if nq == (1,) * nlegs:
            # single-block legs: nothing to sort or bunch
            self._perm = None
            self._strides = np.zeros(nlegs, dtype=np.intp)
            self._set_charges(charges)
            self._set_block_sizes(sizes)
            q_map[:, 0], q_map[:, 1], q_map[:, 2] = 0, self.ind_len, 0
            self.q_map = q_map
            self.q_map_slices = np.array([0, 1], np.intp)
            self.sorted = self.bunched = True
            return
        if sort and qnumber > 0:
            perm = np.lexsort(charges.T)
            q_map, charges, sizes = q_map[perm], charges[perm], sizes[perm]
            self._perm = _inverse_permutation(perm)
        else:
            self._perm = None
        self._set_charges(charges)
        self.sorted = bool(sort) or (qnumber == 0)
        self.bunched = False
        self._set_block_sizes(sizes)
        starts = self.slices[:-1].copy()
        stops = self.slices[1:].copy()
        if bunch:
            idx = _find_row_differences(charges)
            Qi = np.zeros(nblocks, dtype=np.intp)
            Qi[idx[1:-1]] = 1
            Qi = np.cumsum(Qi)
            self._set_charges(charges[idx[:-1]])
            self._set_slices(self.slices[idx])
            self.bunched = True
        else:
            idx = np.arange(nblocks + 1, dtype=np.intp)
            Qi = np.arange(nblocks, dtype=np.intp)
        q_map[:, 2] = Qi
        q_map[:, 0] = starts - self.slices[Qi]
        q_map[:, 1] = stops - self.slices[Qi]
        self.q_map = q_map
        self.q_map_slices = idx

    def test_sanity(self):
        LegCharge.test_sanity(self)
        if self.q_map is None:
            return
        if self.q_map.shape != (int(np.prod(self.subqshape)), 3 + self.nlegs):
            raise ValueError("q_map has wrong shape")
        if any(l.chinfo != self.chinfo for l in self.legs):
            raise ValueError("leg with different ChargeInfo")

    def copy(self):
        res = _copy.copy(self)
        res._bsizes = None
        return res

    def apply_charge_mapping(self, map_func, func_args=(), func_kwargs={}):
        res = self.copy()
        res.legs = tuple(l.apply_charge_mapping(map_func, func_args, func_kwargs) for l in self.legs)
        res.charges = map_func(self.charges, *func_args, **func_kwargs)
        res.sorted = res.bunched = False
        return res

    def save_hdf5(self, hdf5_saver, h5gr, subpath):
        """The reference's layout (charges.py:1598): the LegCharge fields plus the incoming ``legs``."""
        LegCharge.save_hdf5(self, hdf5_saver, h5gr, subpath)
        hdf5_saver.save(list(self.legs), subpath + 'legs')

    @classmethod
    def from_hdf5(cls, hdf5_loader, h5gr, subpath):
        flags = [hdf5_loader.get_attr(h5gr, key) for key in ('sorted', 'bunched')]
        obj = cls(hdf5_loader.load(subpath + 'legs'), hdf5_loader.get_attr(h5gr, 'qconj'), flags[0], flags[1])
        hdf5_loader.memorize_load(h5gr, obj)
        return obj

    def to_LegCharge(self):
        res = LegCharge(self.chinfo, self.slices, self.charges, self.qconj)
        res.sorted, res.bunched = self.sorted, self.bunched
        return res

    def conj(self):
        """Pipe with opposite ``qconj`` and conjugated incoming legs (shares ``q_map``)."""
        res = _copy.copy(self)
        res.qconj = -self.qconj
        res.legs = tuple(l.conj() for l in self.legs)
        return res

    def outer_conj(self):
        res = _copy.copy(self)
        res.qconj = -self.qconj
        res.charges = self.chinfo.make_valid(-self.charges)
        res.sorted = False
        return res

    def sort(self, *args, **kwargs):
        return self.to_LegCharge().sort(*args, **kwargs)

    def bunch(self, *args, **kwargs):
        return self.to_LegCharge().bunch(*args, **kwargs)

    def project(self, *args, **kwargs):
        return self.to_LegCharge().project(*args, **kwargs)

    def map_incoming_flat(self, incoming_indices):
        """Flat index of the pipe for one flat index per incoming leg."""
        if len(incoming_indices) != self.nlegs:
            raise ValueError("wrong len of flat_ind_incomming")
        qind_in = np.empty((1, self.nlegs), dtype=np.intp)
        inner = []
        for a, (leg, i) in enumerate(zip(self.legs, incoming_indices)):
            qi, within = leg.get_qindex(i)
            qind_in[0, a] = qi
            inner.append(within)
        row = self.q_map[self._map_incoming_qind(qind_in)[0]]
        sizes = [int(leg.get_block_sizes()[q]) for leg, q in zip(self.legs, row[3:])]
        off = 0
        for s, w in zip(sizes, inner):
            off = off * s + w
        return int(self.slices[row[2]] + row[0] + off)

    def _map_incoming_qind(self, qind_incoming):
        """Row index into ``q_map`` for each incoming qindex tuple."""
        inds = np.sum(np.asarray(qind_incoming, dtype=np.intp) * self._strides[np.newaxis, :], axis=1)
        if self._perm is None:
            return inds
        return self._perm[inds]

    def __str__(self):
        return "\n".join(["LegPipe(shape {0!s}->{1:d}, ".format(self.subshape, self.ind_len),
                          "    qconj {0}->{1:+1};".format('(' + ', '.join("%+d" % l.qconj for l in self.legs) + ')',
                                                          self.qconj),
                          "    block numbers {0!s}->{1:d})".format(self.subqshape, self.block_number),
                          LegCharge.__str__(self), ")"])

    def __repr__(self):
        return "LegPipe({legs},\nqconj={qconj:+d}, sort={s!r}, bunch={b!r})".format(
            legs='[' + ',\n'.join(repr(l) for l in self.legs) + ']', qconj=self.qconj, s=self.sorted, b=self.bunched)

    def __getstate__(self):
        return (LegCharge.__getstate__(self), self.nlegs, self.legs, self.subshape, self.subqshape, self.q_map,
                self.q_map_slices, self._perm, self._strides)

    def __setstate__(self, state):
        (base, self.nlegs, self.legs, self.subshape, self.subqshape, self.q_map, self.q_map_slices, self._perm,
         self._strides) = state
        LegCharge.__setstate__(self, base)


# ---- module-level helpers (names as in the reference, charges.py:1899-2015) --------------------------

def _find_row_differences(qflat):
    """Indices where consecutive rows of a 2D array differ, including ``0`` and ``len`` (reference :1922)."""
    qflat = np.asarray(qflat)
    n = qflat.shape[0]
    if n < 2:
        return np.array([0, n], dtype=np.intp)
    changed = np.any(qflat[1:] != qflat[:-1], axis=1) if qflat.ndim == 2 and qflat.shape[1] > 0 \
        else np.zeros(n - 1, dtype=bool)
    return np.concatenate([[0], np.nonzero(changed)[0] + 1, [n]]).astype(np.intp)


def _partial_qtotal(chinfo, legs, qdata, qconj=1, add_qtotal=None):
    """``sum_legs qconj_leg * charges[qdata[:, leg]]`` (times ``qconj``), made valid (reference :1899)."""
    qdata = np.asarray(qdata)
    res = np.zeros((qdata.shape[0], chinfo.qnumber), dtype=QTYPE)
    for a, leg in enumerate(legs):
        res += (qconj * leg.qconj) * leg.charges[qdata[:, a]]
    if add_qtotal is not None:
        res += add_qtotal
    return chinfo.make_valid(res)


def _make_stride(shape, cstyle=True):
    L = len(shape)
    stride = 1
    res = np.empty([L], np.intp)
    order = range(L - 1, -1, -1) if cstyle else range(L)
    for a in order:
        res[a] = stride
        stride *= shape[a]
    return res


def _map_blocks(blocksizes):
    """For blocks of the given sizes laid out one after the other: the block number of every index (reference :1945)."""
    return np.repeat(np.arange(len(blocksizes), dtype=np.intp), np.asarray(blocksizes, dtype=np.intp))


def _sliced_copy(dest, dest_beg, src, src_beg, slice_shape):
    """``dest[dest_beg : dest_beg + slice_shape] = src[src_beg : src_beg + slice_shape]`` on host arrays
    (reference :1956; the device twin is ``tpa_copy_batch``)."""
    nd = dest.ndim
    dest_beg = [0] * nd if dest_beg is None else dest_beg
    src_beg = [0] * nd if src_beg is None else src_beg
    assert src.ndim == nd == len(dest_beg) == len(src_beg) == len(slice_shape)
    dest[tuple(slice(b, b + n) for b, n in zip(dest_beg, slice_shape))] = \
        src[tuple(slice(b, b + n) for b, n in zip(src_beg, slice_shape))]
