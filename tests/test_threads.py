"""Concurrent callers (SURVEY 8(b) "Threading"; reference ``algorithms/dmrg_parallel.py:57-90`` ``DMRGThreadPlusHC``: a worker thread
contracts the ``+ h.c.`` half of the effective Hamiltonian while the main thread contracts the other half): two Python threads run
``tensordot`` / ``combine_legs`` / ``inner`` / axpy through the shared plan and table caches at the same time while a third keeps
setting the module-global ``np_conserved.svd_hint``.  Every result must equal the single-threaded one bit for bit (the kernels are
deterministic and every call owns its output arena), the caches must survive concurrent eviction, and a hint left behind by another
thread must only cost the next ``svd`` a fallback to the cold path.  Runs on the emulated device and on the MI355X."""
import threading

import numpy as np

from tenpy_amd.linalg import np_conserved as npc
from tenpy_amd.linalg.charges import ChargeInfo, LegCharge


def _leg(rng, ch, n, qconj):
    return LegCharge.from_qflat(ch, rng.randint(-2, 3, size=(n, 1)), qconj).bunch()[1]


def _rand(rng, legs, labels):
    return npc.Array.from_func(rng.standard_normal, legs, dtype=np.float64).iset_leg_labels(labels)


def _work(a, b, th):
    """A slice of a bond update: two tensordots, a leg fusion, a scalar product, an axpy."""
    r = npc.tensordot(a, b, axes=(['c', 'b'], ['c*', 'b*']))
    r2 = npc.tensordot(r, th, axes=(['d'], ['d*']))
    f = r2.combine_legs([['a', 'e']])
    n = npc.inner(f, f, axes='range', do_conj=True)
    g = f.copy(deep=True)
    g.iadd_prefactor_other(0.25, f)
    return r2.to_ndarray(), f.to_ndarray(), float(np.real(n)), g.to_ndarray()


def test_two_contraction_threads_and_a_hint_writer(backend, monkeypatch):
    rng = np.random.RandomState(11)
    ch = ChargeInfo([1])
    jobs = []
    for k in range(6):
        la, lb, lc, ld, le = (_leg(rng, ch, n, q) for n, q in ((12 + k, 1), (9, -1), (10, 1), (8 + k, -1), (7, 1)))
        a = _rand(rng, [la, lb, lc], ['a', 'b', 'c'])
        b = _rand(rng, [lc.conj(), ld, lb.conj()], ['c*', 'd', 'b*'])
        th = _rand(rng, [ld.conj(), le], ['d*', 'e'])
        jobs.append((a, b, th))
    want = [_work(*j) for j in jobs]
    monkeypatch.setattr(npc, '_PLAN_CACHE_SIZE', 4)       # tiny LRU: the threads evict each other's plans all the time
    errors, stop = [], threading.Event()

    def contract(offset):
        try:
            for rep in range(6):
                for i in range(len(jobs)):
                    k = (i + offset) % len(jobs)
                    got = _work(*jobs[k])
                    for x, y in zip(got, want[k]):
                        np.testing.assert_array_equal(x, y)
        except BaseException as e:      # noqa: BLE001  (reported in the main thread)
            errors.append(e)

    def hint_writer():
        i = 0
        while not stop.is_set():
            npc.svd_hint = (('other thread', i), 'R' if i % 2 else 'L')
            i += 1
            npc.svd_hint = None

    ts = [threading.Thread(target=contract, args=(0,)), threading.Thread(target=contract, args=(3,))]
    hw = threading.Thread(target=hint_writer)
    hw.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    hw.join()
    assert not errors, errors[:1]
    # a hint that some other thread left behind: no basis under that key -> the cold path, the correct result
    m = _rand(rng, [_leg(rng, ch, 30, 1), _leg(rng, ch, 30, -1)], ['x', 'y'])
    npc.svd_hint = (('other thread', 12345), 'R')
    U, S, VH = npc.svd(m)
    assert npc.svd_hint is None
    ref = np.linalg.svd(m.to_ndarray(), compute_uv=False)
    np.testing.assert_allclose(np.sort(S)[::-1], ref[:len(S)], rtol=0, atol=1e-13 * ref[0])
    np.testing.assert_allclose((U.to_ndarray() * S) @ VH.to_ndarray(), m.to_ndarray(), rtol=0, atol=1e-12 * ref[0])
