"""N>1 path on CPU: world_size-2 ``gloo`` processes, mock device.  The sharded matvec (row-panel split of
both tensordots + one all-gather) must reproduce the unsharded matvec bit-for-bit on every rank, and a DMRG
run with ``shard_matvec=True`` must give the golden energies of the reference."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from _pytest.monkeypatch import MonkeyPatch
        import mock_device
        mp = MonkeyPatch()
        mock_device.install(mp)
        from helpers import golden
        from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
        from tenpy_amd.algorithms.mps_common import TwoSiteH
        from tenpy_amd.algorithms.sharded import ShardedTwoSiteH, row_partition
        from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
        from tenpy_amd.networks.mps import MPS
        def same(got, want):
            """Sharded == unsharded: bit for bit at world 2; with more, smaller row panels the emulation's BLAS (micro-kernels chosen by the
            panel height) differs in the last bit -- the device kernel sums every element in one fixed order whatever the partition is."""
            if world == 2:
                np.testing.assert_array_equal(got, want)
            else:
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-14 * max(1., float(np.abs(want).max())))
        rec = [r for r in golden('dmrg.pkl') if r['name'] == 'xxz_L12_chi20_hz'][0]
        L = rec['L']
        H = xxz_chain_mpo(L, rec['Jxx'], rec['Jz'], rec['hz'])
        _, p = spin_half_leg('Sz')
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}, 'shard_matvec': True})
        from tenpy_amd.linalg import krylov_based as kb
        from tenpy_amd.linalg import _svd_warm
        for s in range(3):
            eng.sweep()
            assert abs(eng.sweep_stats['E'][-1] - rec['E_sweeps'][s]) <= 1e-10 * abs(rec['E_sweeps'][s])
        # round 4 (VERDICT r3 task 5): N > 1 runs the single-GPU fast path -- the Lanczos recurrence as ONE tpa_lanczos_run call with
        # the all-gather as the launch program's collective op, and the warm-started block SVD (replicated on every rank)
        assert kb.stats.get('n_native_sharded', 0) > 50, kb.stats
        assert _svd_warm.stats['warm_calls'] > 0, _svd_warm.stats
        # ... and the native sharded run gives the numbers of the unsharded one: same N, E to rounding, the same vector
        from tenpy_amd.linalg.krylov_based import LanczosGroundState
        from tenpy_amd.linalg import np_conserved as npc0
        for i_b in (L // 2 - 1, 2):
            hs, hr = ShardedTwoSiteH(eng.env, i_b), TwoSiteH(eng.env, i_b)
            th = hr.combine_theta(psi.get_theta(i_b, n=2))
            n0 = kb.stats.get('n_native_sharded', 0)
            E_s, v_s, N_s = LanczosGroundState(hs, th, {'N_min': 2, 'N_max': 20, 'P_tol': 1.e-14}).run()
            assert kb.stats.get('n_native_sharded', 0) == n0 + 1
            E_r, v_r, N_r = LanczosGroundState(hr, th, {'N_min': 2, 'N_max': 20, 'P_tol': 1.e-14}).run()
            assert N_s == N_r and abs(E_s - E_r) <= 1e-13 * abs(E_r)
            same(hs.prepare_svd(v_s).to_ndarray(), hr.prepare_svd(v_r).to_ndarray())
        # matvec: sharded == unsharded on this rank
        i0 = L // 2 - 1
        from tenpy_amd.algorithms import mps_common
        mps_common.FACTORED_MIN_SECTOR = 0                   # (small test blocks: do not fall back to the fused form)
        for factored in (True, False):
            mps_common.FACTORED_MATVEC = factored            # both forms of the operator, sharded vs unsharded
            ref_H = TwoSiteH(eng.env, i0)
            sh_H = ShardedTwoSiteH(eng.env, i0)
            assert ref_H.factored == factored and sh_H.factored == factored
            theta = ref_H.combine_theta(psi.get_theta(i0, n=2))
            a, b = ref_H.matvec(theta), sh_H.matvec(theta)
            a2, b2 = ref_H.matvec(theta), sh_H.matvec(theta)    # cached plans
            np.testing.assert_array_equal(a._qdata, b._qdata)
            same(b.to_ndarray(), a.to_ndarray())
            same(b2.to_ndarray(), a2.to_ndarray())
            bounds = sh_H._sharded['bounds']
            n_rows = (ref_H._LPf if factored else ref_H.LHeff).legs[0].ind_len
            assert bounds[0] == 0 and bounds[-1] == n_rows and np.all(np.diff(bounds) >= 0)
        mps_common.FACTORED_MATVEC = True
        # block SVD distributed over the ranks (LPT by block cost + one all-gather) == the local SVD, bit for bit
        from tenpy_amd.linalg import np_conserved as npc
        th2 = ref_H.prepare_svd(ref_H.combine_theta(psi.get_theta(i0, n=2)))
        # (the engine distributes the SVD only inside its own bond updates: the module switch is back to None afterwards, so
        # a later SVD on one rank alone -- diagnostics, post-processing -- cannot dead-lock)
        assert npc.SVD_DIST_GROUP is None and eng._svd_group is not None and eng._svd_group[2] == world
        npc.SVD_DIST_GROUP = eng._svd_group
        try:
            Ud, Sd, Vd = npc.svd(th2, inner_labels=['vR', 'vL'])
        finally:
            npc.SVD_DIST_GROUP = None
        Ul, Sl, Vl = npc.svd(th2, inner_labels=['vR', 'vL'])
        same(Sd, Sl)
        same(Ud.to_ndarray(), Ul.to_ndarray())
        same(Vd.to_ndarray(), Vl.to_ndarray())
        owners = npc.svd_block_owners(np.array([9, 5, 5, 2, 2, 1]), np.array([9, 5, 5, 2, 2, 1]), world)
        assert set(owners.tolist()) == set(range(min(world, 6)))      # LPT with fewer blocks than ranks: some ranks own nothing
        # edge bonds have fewer rows than ranks: ranks with EMPTY row ranges run the same program (VERDICT r5 task 6)
        e_H = ShardedTwoSiteH(eng.env, 0)
        e_th = e_H.combine_theta(psi.get_theta(0, n=2))
        e_ref = TwoSiteH(eng.env, 0).matvec(e_th)
        same(e_H.matvec(e_th).to_ndarray(), e_ref.to_ndarray())
        if e_H._sharded is not None:
            e_b = e_H._sharded['bounds']
            assert len(e_b) == world + 1 and (world <= 2 or np.any(np.diff(e_b) == 0) or e_b[-1] >= world)
        # every element of theta' is produced by exactly one rank
        segs = sh_H._sharded['segs']
        cover = np.zeros(sh_H._sharded['p2'].res_total, dtype=int)
        for r in range(world):
            for off, n in segs[r]:
                cover[off:off + n] += 1
        assert np.all(cover == 1)
        # north_star's variant: Krylov vectors as row panels, alpha / beta by all-reduce -- same Krylov space, same numbers
        from tenpy_amd.algorithms.sharded import lanczos_row_panels
        from tenpy_amd.linalg.krylov_based import LanczosGroundState
        for factored in (True, False):
            mps_common.FACTORED_MATVEC = factored
            sh_H = ShardedTwoSiteH(eng.env, i0)
            theta = sh_H.combine_theta(psi.get_theta(i0, n=2))
            opts = {'N_min': 2, 'N_max': 20, 'P_tol': 1.e-14}
            E_a, th_a, N_a = LanczosGroundState(TwoSiteH(eng.env, i0), theta, opts).run()
            E_b, th_b, N_b = lanczos_row_panels(sh_H, theta, opts)
            assert N_a == N_b and abs(E_a - E_b) < 1e-12 * abs(E_a)
            ov = abs(npc.inner(th_a, th_b, axes='range', do_conj=True))
            assert abs(ov - 1.) < 1e-10, ov
        mps_common.FACTORED_MATVEC = True
        psi2 = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        eng2 = TwoSiteDMRGEngine(psi2, H, {'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}, 'shard_matvec': True,
                                           'krylov_row_panels': True})
        for s_ in range(3):
            eng2.sweep()
            assert abs(eng2.sweep_stats['E'][-1] - rec['E_sweeps'][s_]) <= 1e-10 * abs(rec['E_sweeps'][s_]), (s_, eng2.sweep_stats['E'][-1], rec['E_sweeps'][s_], eng.sweep_stats['E'][s_])
        ret[rank] = 'ok'
    except Exception as e:  # pragma: no cover
        import traceback
        ret[rank] = 'FAIL: ' + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_matvec_gloo(world):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 1000) + 7 * world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) == 'ok' for r in range(world)), dict(ret)


def _tebd_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from _pytest.monkeypatch import MonkeyPatch
        import mock_device
        mp = MonkeyPatch()
        mock_device.install(mp)
        from helpers import golden
        from tenpy_amd.algorithms.sharded import ShardedTEBDEngine
        from tenpy_amd.algorithms.tebd import TEBDEngine
        from tenpy_amd.models.spin_chains import spin_half_leg
        from tenpy_amd.networks.mps import MPS
        rec = [r for r in golden('tebd.pkl') if r['name'] == 'tfi_quench_L10_parity'][0]
        L = rec['L']
        _, p = spin_half_leg(rec['conserve'])
        up = dict(rec['state_labels'])['up']
        opts = {'dt': rec['dt'], 'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}}
        psi_s = MPS.from_product_state([p] * L, [up] * L, dtype=np.complex128)
        psi_r = MPS.from_product_state([p] * L, [up] * L, dtype=np.complex128)
        eng_s, eng_r = ShardedTEBDEngine(psi_s, rec['h_bond'], dict(opts)), TEBDEngine(psi_r, rec['h_bond'], dict(opts))
        calls = [0]
        orig = eng_s._decompose_bonds
        eng_s._decompose_bonds = lambda bonds, U: (calls.__setitem__(0, calls[0] + len(bonds)), orig(bonds, U))[1]
        for step in range(6):
            eng_s.evolve_step_order2()
            eng_r.evolve_step_order2()
            np.testing.assert_array_equal(psi_s.entanglement_entropy(), psi_r.entanglement_entropy())     # bit-identical on every rank
            assert eng_s.norm == eng_r.norm and eng_s.trunc_err.eps == eng_r.trunc_err.eps
        for i in range(L):
            np.testing.assert_array_equal(psi_s.get_B(i, 'B').to_ndarray(), psi_r.get_B(i, 'B').to_ndarray())
        # every rank decomposed only its share: 6 steps x (5 + 4 + 5) bonds, dealt r, r + N, ...
        per_step = sum(len(list(range(1, L))[q::2][rank::world]) for q in (0, 1, 0))
        assert calls[0] == 6 * per_step, (calls[0], per_step)
        ret[rank] = 'ok'
    except Exception:  # pragma: no cover
        import traceback
        ret[rank] = 'FAIL: ' + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_tebd_gloo(world):
    """Bond-sharded TEBD (``algorithms/sharded.ShardedTEBDEngine``): the bonds of every half-step dealt over 2 ``gloo`` ranks, tensors
    broadcast from their owners -- the state on every rank bit-identical to the single-process engine, each rank decomposing half of
    the bonds (world 4: its share of the bonds, r, r + N, ...)."""
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 1000) + 7 * world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tebd_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) == 'ok' for r in range(world)), dict(ret)


def test_row_partition_balanced():
    sys.path.insert(0, ROOT)
    from tenpy_amd.algorithms.sharded import row_partition
    w = np.exp(-np.linspace(-3, 3, 1000)**2)        # Gaussian sector weights
    for world in (1, 2, 4, 8):
        b = row_partition(w, world)
        assert b[0] == 0 and b[-1] == 1000 and len(b) == world + 1
        shares = [w[b[i]:b[i + 1]].sum() for i in range(world)]
        assert max(shares) <= w.sum() / world * 1.05 + w.max()


def test_flop_share_per_rank():
    """The row partition of the sharded matvec gives every rank <= 1.1 / N of the GEMM flops of BOTH steps (VERDICT r4 task 9): charge
    sectors are split by rows, not dealt out whole -- the centre sector alone carries ~45 % of the flops."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _pytest.monkeypatch import MonkeyPatch
    import mock_device
    mp = MonkeyPatch()
    try:
        mock_device.install(mp)
        from tenpy_amd.algorithms import mps_common
        from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
        from tenpy_amd.algorithms.sharded import ShardedTwoSiteH
        from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
        from tenpy_amd.networks.mps import MPS
        L = 16
        H = xxz_chain_mpo(L, 1., 1., 0.)
        _, p = spin_half_leg('Sz')
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': 48, 'svd_min': 1.e-12}})
        for _ in range(2):
            eng.sweep()
        mp.setattr(mps_common, 'FACTORED_MIN_SECTOR', 0)
        i0 = L // 2 - 1
        for factored in (True, False):
            mp.setattr(mps_common, 'FACTORED_MATVEC', factored)
            for world in (2, 4, 8):
                shares = []
                for rank in range(world):
                    sh = ShardedTwoSiteH.__new__(ShardedTwoSiteH)
                    mps_common.TwoSiteH.__init__(sh, eng.env, i0, True, True)
                    sh.group, sh.world, sh.rank, sh._sharded = None, world, rank, None
                    theta = sh.combine_theta(psi.get_theta(i0, n=2))
                    s = sh._build_sharded_factored(theta) if sh.factored else sh._build_sharded(theta)
                    assert sh.factored == factored and s is not None
                    fl = 0.
                    for sp in (s['sp1'], s['sp2']):
                        if sp.local_empty:
                            continue
                        tasks, links = np.asarray(sp.tasks_dev.cpu()), np.asarray(sp.links_dev.cpu())
                        for t in tasks:
                            fl += 2. * float(t[1]) * float(t[2]) * float(np.sum(links[int(t[4]):int(t[4]) + int(t[5]), 2]))
                    shares.append(fl)
                    total = s['p1'].flops + s['p2'].flops
                assert abs(sum(shares) - total) <= 1e-9 * total, (sum(shares), total)
                # one row of the heaviest sector is the granularity of the cut
                assert max(shares) <= 1.1 * total / world + total / theta.legs[0].ind_len * 2, (factored, world, shares, total)
    finally:
        mp.undo()


def test_row_restricted_plan_writes_its_rows_only():
    """`restrict_plan_rows` copied the split-K tables of the FULL plan, and `TensordotPlan.apply` prefers them -- a half-row
    sub-plan then ran the whole GEMM.  A sub-plan never carries the parent's `sk` (round 6: it gets a split of its OWN row panel), and
    `apply` leaves every element outside its segments untouched."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _pytest.monkeypatch import MonkeyPatch
    import mock_device
    mp = MonkeyPatch()
    try:
        mock_device.install(mp)
        from tenpy_amd.algorithms import mps_common
        from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
        from tenpy_amd.algorithms.sharded import restrict_plan_rows, row_partition
        from tenpy_amd.linalg import _device as dev
        from tenpy_amd.linalg import np_conserved as npc
        from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
        from tenpy_amd.networks.mps import MPS
        mp.setattr(npc, 'GEMM_SPLIT_K', (4, 4096, 2, 1024))         # split even the short chains of a small test state
        mp.setattr(npc, 'GEMM_K_TILE', 2)                            # (cuts snap to whole k-tiles: 16 would swallow chains of <= 10)
        npc.clear_device_caches()
        L = 16
        H = xxz_chain_mpo(L, 1., 1., 0.)
        _, p = spin_half_leg('Sz')
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': 48, 'svd_min': 1.e-12}})
        for _ in range(2):
            eng.sweep()
        i0 = L // 2 - 1
        mp.setattr(mps_common, 'FACTORED_MATVEC', False)
        Heff = mps_common.TwoSiteH(eng.env, i0)
        theta = Heff.combine_theta(psi.get_theta(i0, n=2))
        plan, _, _ = npc.plan_tensordot(Heff.LHeff, theta, axes=['(vR.p0*)', '(vL.p0)'])
        assert plan.sk is not None, "the test needs a split-K parent plan"
        full = plan.apply(Heff.LHeff, theta)
        leg0 = Heff.LHeff.legs[0]
        bounds = row_partition(np.ones(leg0.ind_len), 2)
        covered = np.zeros(plan.res_total, dtype=int)
        n_own_split = 0
        for r in range(2):
            sub = restrict_plan_rows(plan, leg0, int(bounds[r]), int(bounds[r + 1]))
            assert sub.sk is not plan.sk
            n_own_split += sub.sk is not None
            out = dev.empty(plan.res_total, plan.dtype)
            out.fill_(777.)
            res = sub.apply(Heff.LHeff, theta, out_arena=out)
            got, want = np.asarray(res._arena), np.asarray(full._arena)
            mine = np.zeros(plan.res_total, dtype=bool)
            for off, n in sub.segments:
                mine[off:off + n] = True
            np.testing.assert_array_equal(got[~mine], 777.)                    # rows of the other rank: not written
            np.testing.assert_allclose(got[mine], want[mine], rtol=0, atol=1e-13 * np.abs(want).max())
            covered += mine
        assert np.all(covered == 1)
        assert n_own_split > 0, "the row panels of this plan are few-tile launches: they get a split of their own"
    finally:
        mp.undo()
        from tenpy_amd.linalg import np_conserved as npc
        npc.clear_device_caches()


def _failure_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from _pytest.monkeypatch import MonkeyPatch
        import mock_device
        mp = MonkeyPatch()
        mock_device.install(mp)
        from helpers import golden
        from tenpy_amd.algorithms.sharded import ShardedTEBDEngine, ShardedTwoSiteH
        from tenpy_amd.models.spin_chains import spin_half_leg, xxz_chain_mpo
        from tenpy_amd.networks.mps import MPS
        # ---- TEBD: one rank's decomposition fails -> EVERY rank raises before the broadcasts
        rec = [r for r in golden('tebd.pkl') if r['name'] == 'tfi_quench_L10_parity'][0]
        L = rec['L']
        _, p = spin_half_leg(rec['conserve'])
        up = dict(rec['state_labels'])['up']
        psi = MPS.from_product_state([p] * L, [up] * L, dtype=np.complex128)
        eng = ShardedTEBDEngine(psi, rec['h_bond'], {'dt': rec['dt'], 'trunc_params': {'chi_max': rec['chi'], 'svd_min': 1.e-10}})
        eng.evolve_step_order2()
        if rank == 1:
            def boom(bonds, U):
                raise np.linalg.LinAlgError("injected")
            eng._decompose_bonds = boom
        try:
            eng.evolve_step_order2()
            raise AssertionError("rank %d went on after the failure of rank 1" % rank)
        except np.linalg.LinAlgError as e:
            assert ("injected" in str(e)) == (rank == 1), str(e)
        # ---- Lanczos: the native run fails on one rank after its last collective -> every rank raises before the next one
        from tenpy_amd.algorithms.dmrg import TwoSiteDMRGEngine
        from tenpy_amd.linalg import _device as dev
        from tenpy_amd.linalg import krylov_based as kb
        L = 12
        H = xxz_chain_mpo(L, 1., 1., 0.)
        _, p = spin_half_leg('Sz')
        psi = MPS.from_product_state([p] * L, [1, 0] * (L // 2))
        eng = TwoSiteDMRGEngine(psi, H, {'trunc_params': {'chi_max': 20, 'svd_min': 1.e-10}, 'shard_matvec': True})
        eng.sweep()
        hs = ShardedTwoSiteH(eng.env, L // 2 - 1)
        th = hs.combine_theta(psi.get_theta(L // 2 - 1, n=2))
        n0 = kb.stats.get('n_native_sharded', 0)
        kb.LanczosGroundState(hs, th, {'N_min': 2, 'N_max': 20}).run()
        assert kb.stats.get('n_native_sharded', 0) == n0 + 1
        if rank == 0:
            orig = dev.check

            def check(rc, what=""):
                if what == "lanczos_run":
                    raise RuntimeError("injected HIP error")
                return orig(rc, what)
            mp.setattr(dev, 'check', check)
        try:
            kb.LanczosGroundState(hs, th, {'N_min': 2, 'N_max': 20}).run()
            raise AssertionError("rank %d went on after the failure of rank 0" % rank)
        except RuntimeError as e:
            assert ("injected" in str(e)) == (rank == 0) and ("another rank" in str(e)) == (rank != 0), str(e)
        mp.undo()
        ret[rank] = 'ok'
    except Exception:  # pragma: no cover
        import traceback
        ret[rank] = 'FAIL: ' + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_failures_are_agreed_on_before_the_next_collective(world):
    """A rank that raises alone must not leave the others waiting in a collective: the bond-sharded TEBD half-step and the sharded
    native Lanczos run all-reduce a failure flag and raise on EVERY rank (``_svd_distributed`` has done so since round 3)."""
    import torch.multiprocessing as mp
    port = 29900 + (os.getpid() % 1000) + 7 * world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_failure_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) == 'ok' for r in range(world)), dict(ret)


@pytest.mark.gpu
def test_rccl_collective_world1(tmp_path):
    """The collective of the N > 1 path on RCCL: ``bench.py --force-dist`` on ONE MI355X initialises the nccl (= RCCL) process group with a
    single rank and forces the row-sharded operator, so ``all_gather_into_tensor`` runs from the collective callback of
    ``tpa_lanczos_run`` on the launch stream, and the SVD all-gather / all-reduce of ``_svd_distributed`` run too (the RCCL
    path had never executed anywhere).  Own process + timeout: a hung collective must not hang the suite."""
    import json
    import subprocess
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--L', '16', '--chi', '48', '--steps', '1', '--warmup', '1',
                         '--no-cpu-baseline', '--no-extras'], capture_output=True, text=True, timeout=420, cwd=str(tmp_path),
                        env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'TPA_BENCH_BACKEND')})
    assert pr.returncode == 0, pr.stderr[-3000:]
    out = json.loads([l for l in pr.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 1 and 'RCCL' in out['config']['parallelism']
    assert out['lanczos_stats']['n_native_sharded'] > 20
    # the same chain unsharded: the collective of a one-rank group is the identity, so the energies agree to rounding
    pr2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--L', '16', '--chi', '48', '--steps', '1', '--warmup', '1',
                          '--no-cpu-baseline', '--no-extras'], capture_output=True, text=True, timeout=420, cwd=str(tmp_path))
    assert pr2.returncode == 0, pr2.stderr[-3000:]
    out2 = json.loads([l for l in pr2.stdout.splitlines() if l.startswith('{')][-1])
    assert abs(out['E'] - out2['E']) < 1e-11 * abs(out2['E'])
