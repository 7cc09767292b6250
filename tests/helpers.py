"""Shared test helpers: load golden fixtures into tenpy_amd objects, compare results bit-exactly
(integers) / within tolerance (floating point)."""
import os
import pickle

import numpy as np

from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge, LegPipe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_chinfos = {}


def golden(name):
    with open(os.path.join(GOLDEN, name), 'rb') as f:
        return pickle.load(f)


def chinfo_of(mod):
    key = tuple(int(m) for m in mod)
    if key not in _chinfos:
        _chinfos[key] = ChargeInfo(list(key))
    return _chinfos[key]


def load_leg(d):
    ch = chinfo_of(d['mod'])
    if 'pipe' in d:
        sub = [load_leg(x) for x in d['pipe']['legs']]
        pipe = LegPipe(sub, qconj=d['qconj'], sort=d['sorted'], bunch=d['bunched'])
        return pipe
    return LegCharge.from_qind(ch, d['slices'], d['charges'], d['qconj'])


def load_array(d):
    legs = [load_leg(x) for x in d['legs']]
    a = npc.Array.from_ndarray(d['dense'], legs, dtype=np.dtype(d['dtype']), qtotal=d['qtotal'], labels=d['labels'],
                               cutoff=0.)
    return a


def assert_leg_equal(leg, d, check_pipe=True):
    np.testing.assert_array_equal(leg.slices, d['slices'])
    np.testing.assert_array_equal(leg.charges, d['charges'])
    assert leg.qconj == d['qconj']
    assert leg.slices.dtype == np.intp and leg.charges.dtype == np.int64
    if check_pipe and 'pipe' in d:
        assert isinstance(leg, LegPipe)
        np.testing.assert_array_equal(leg.q_map, d['pipe']['q_map'])
        np.testing.assert_array_equal(leg.q_map_slices, d['pipe']['q_map_slices'])
        for sub, sd in zip(leg.legs, d['pipe']['legs']):
            assert_leg_equal(sub, sd)


def assert_array_matches(a, d, rtol=1e-13, check_labels=True, data=True):
    """integer bookkeeping exact; block data within rtol * max|entry|."""
    a.test_sanity()
    assert a.rank == len(d['legs'])
    for leg, ld in zip(a.legs, d['legs']):
        assert_leg_equal(leg, ld)
    np.testing.assert_array_equal(a.qtotal, d['qtotal'])
    assert str(a.dtype) == d['dtype']
    if check_labels:
        assert list(a._labels) == list(d['labels'])
    gq = np.asarray(d['qdata']).reshape(-1, a.rank)
    mine = a.copy(deep=False)
    mine._qdata, mine._offsets = a._qdata.copy(), a._offsets.copy()
    if d['qdata_sorted']:
        # the reference's result is flagged sorted: OUR rows must be in that (lexsort) order as they are -- no sorting here --
        # and carry the flag (`stored_blocks < 2`: trivially sorted whatever the flag says)
        np.testing.assert_array_equal(a._qdata, gq)
        assert a._qdata_sorted or a.stored_blocks < 2
        order = np.arange(len(gq))
    else:
        # same set of blocks
        mine._qdata_sorted = False
        mine.isort_qdata()
        order = np.lexsort(gq.T) if len(gq) > 1 else np.arange(len(gq))
        np.testing.assert_array_equal(mine._qdata, gq[order])
    if data:
        blocks = mine._data
        scale = max([np.max(np.abs(b)) if b.size else 0. for b in d['blocks']] + [1e-300])
        for i, j in enumerate(order):
            np.testing.assert_allclose(blocks[i], d['blocks'][j], rtol=0, atol=rtol * scale * max(1, blocks[i].shape[-1]))
        np.testing.assert_allclose(a.to_ndarray(), d['dense'], rtol=0, atol=rtol * scale * 10)
