"""The planned reshaping bookkeeping (round 3: ``combine_legs`` / ``split_legs`` / ``itranspose`` / ``iproject`` replay cached block
lists, offsets and device tables per block structure; ``_device.table``; the fused H_eff construction) must (i) give the same result
on a cache hit as on the miss that made the plan, for different data of the same structure, (ii) NOT hit for inputs that share the
block sizes but differ in what the plan really depends on (the charges behind a pipe's fusion table, the projection mask), and
(iii) hand out read-only shared bookkeeping arrays.  Oracle: dense numpy, as in the reference's tests/test_np_conserved.py."""
import numpy as np
import pytest

from tenpy_amd.linalg import _device as dev
from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge


def _rand(rng, legs, labels):
    return npc.Array.from_func(lambda s: rng.standard_normal(s), legs, dtype=np.float64, labels=labels)


def _legs(ch, shift):
    a = LegCharge.from_qflat(ch, [[0], [1], [1], [2], [2], [2]], 1)
    b = LegCharge.from_qflat(ch, [[0 + shift], [1 + shift], [1 + shift], [2 + shift]], 1)
    c = LegCharge.from_qflat(ch, [[-1], [0], [0], [1], [1], [2]], -1)
    return [x.bunch()[1] for x in (a, b, c)]


def test_reshaping_plans_hit_and_do_not_collide(backend):
    rng = np.random.RandomState(7)
    ch = ChargeInfo([1])
    npc._reshape_plans.clear()
    for shift in (0, 1, 0, 1):          # same block SIZES of leg b, different charges -> different fusion tables -> different plans
        la, lb, lc = _legs(ch, shift)
        for rep in range(2):            # rep 0 may plan, rep 1 replays: different data, same structure
            t = _rand(rng, [la, lb, lc], ['a', 'b', 'c'])
            T = t.to_ndarray()
            n0 = len(npc._reshape_plans)
            f = t.combine_legs(['a', 'b'], qconj=+1)
            f.test_sanity()
            assert f.get_leg_labels() == ['(a.b)', 'c']
            back = f.split_legs()
            back.test_sanity()
            np.testing.assert_array_equal(back.to_ndarray(), T)
            tr = t.transpose(['c', 'a', 'b'])
            tr.test_sanity()
            np.testing.assert_array_equal(tr.to_ndarray(), T.transpose(2, 0, 1))
            mask = np.zeros(la.ind_len, dtype=bool)
            mask[[0, 2, 3, 5]] = True
            pr = t.copy(deep=True)
            pr.iproject(mask, 'a')
            pr.test_sanity()
            np.testing.assert_array_equal(pr.to_ndarray(), T[mask])
            mask2 = ~mask
            pr2 = t.copy(deep=True)
            pr2.iproject(mask2, 'a')
            np.testing.assert_array_equal(pr2.to_ndarray(), T[mask2])
            np.testing.assert_array_equal(t.to_ndarray(), T)          # operands untouched
            if rep == 1:
                assert len(npc._reshape_plans) == n0, "the second pass over the same structure must replay the plans"
    # shared bookkeeping arrays are read-only
    plan = next(iter(npc._reshape_plans.values()))
    with pytest.raises(ValueError):
        plan[0][0, 0] = 5


def test_table_cache_is_by_content(backend):
    a = np.arange(12, dtype=np.int64).reshape(3, 4)
    t1, t2 = dev.table(a), dev.table(a.copy())
    assert t1 is t2
    b = a.copy()
    b[1, 2] += 1
    assert dev.table(b) is not t1
    assert dev.table(a.astype(np.int32)) is not t1
    np.testing.assert_array_equal(dev.to_host(t1), a)


def test_clear_device_caches(backend):
    dev.table(np.arange(5))
    dev.scratch('some_test_buffer', 10, np.float64)
    npc.clear_device_caches()
    assert len(dev._pool) == 0 and len(dev._table_cache) == 0 and len(npc._reshape_plans) == 0 and len(npc._plan_cache) == 0


@pytest.mark.parametrize("model", ['xxz', 'hubbard'])
def test_fused_heff_plan_replay(backend, model, monkeypatch):
    """Second construction of LHeff / RHeff from an environment of the same structure but different entries: the replayed plan must
    give what the generic tensordot + combine_legs construction gives."""
    from tenpy_amd.algorithms import mps_common
    from test_heff import _engine
    eng = _engine(model)
    i0 = eng.psi.L // 2 - 1
    mps_common._heff_plans.clear()
    for rep in range(2):
        LP, RP = eng.env.get_LP(i0).copy(deep=True), eng.env.get_RP(i0 + 1).copy(deep=True)
        if rep == 1:                   # new numbers, same block structure
            LP.iscale_prefactor(1.7)
            RP.iscale_prefactor(-0.3)
        tensors = (LP, RP, eng.H.get_W(i0), eng.H.get_W(i0 + 1))
        monkeypatch.setattr(mps_common, 'FUSED_HEFF', True)
        n0 = len(mps_common._heff_plans)
        fast = mps_common.TwoSiteH(None, i0, tensors=tensors, factored=False)
        if rep == 1:
            assert len(mps_common._heff_plans) == n0, "replay expected"
        monkeypatch.setattr(mps_common, 'FUSED_HEFF', False)
        slow = mps_common.TwoSiteH(None, i0, tensors=tensors, factored=False)
        for name in ('LHeff', 'RHeff'):
            x, y = getattr(fast, name), getattr(slow, name)
            x.test_sanity()
            np.testing.assert_allclose(x.to_ndarray(), y.to_ndarray(), rtol=0, atol=1e-13 * max(1., np.max(np.abs(y.to_ndarray()))))
