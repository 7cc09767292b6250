"""Multi-GPU sharding of the Lanczos matvec over the rows of theta' (SURVEY 8(e)).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` in the CPU tests).
The reference has no distributed code at all (its only parallelism precedent is the ``+h.c.`` worker thread
of ``algorithms/dmrg_parallel.py:30``); this is the MI355X-native addition.

Partition: the flat index range of the fused leg ``(vL.p0)`` (= the rows of theta') is cut into
``world_size`` contiguous ranges of equal *flop* weight -- charge sectors are Gaussian-sized, one sector
carries ~45 % of the flops, so sectors are split by rows, not assigned whole.  Rank r

  1. computes T_r = LHeff[rows_r] . theta            (row panel of LHeff, replicated theta),
  2. computes theta'_r = T_r . RHeff                 (RHeff replicated; no exchange between the steps),
  3. takes part in ONE all-gather of the row panels  (the only collective on the data path).

With the factored operator (``TwoSiteH.factored``, the default for charge-resolved MPOs) the same row partition is applied
to its three steps: ``LP[rows_r] . theta`` (row panel of LP), the blockwise MPO application restricted to the rows of the
panel, ``T3[rows_r] . RP``; every step is row-local, so there is again exactly one all-gather per matvec.

Every rank then holds the full theta' and executes the (HBM-bound, cheap) Lanczos vector kernels and the
tridiagonal eigen-solve redundantly on identical data, so alpha/beta need no scalar all-reduce and the
control flow is identical on all ranks by construction.  SVD and environment update are replicated.
"""
import numpy as np

from ..linalg import _device as dev
from ..linalg import np_conserved as npc
from .mps_common import TwoSiteH, _gemm_ops

__all__ = ['ShardedTwoSiteH', 'row_partition', 'restrict_plan_rows', 'lanczos_row_panels', 'RowPanelOps', 'ShardedTEBDEngine']


def _dist():
    import torch.distributed as dist
    return dist


def _agree_failure(failed, group=None):
    """All-reduce (MAX) of a failure flag: True on every rank if ANY rank failed.  Every rank must call it at the same point; it is
    what keeps a rank that raised alone from leaving the others waiting in the next collective (same rule as ``np_conserved._svd_distributed``)."""
    flag = dev.zeros(1, np.float64)
    flag.fill_(1. if failed else 0.)
    _dist().all_reduce(flag, op=_dist().ReduceOp.MAX, group=group)
    return float(flag.item()) > 0.


def _forced():
    """``TPA_SHARD_FORCE=1``: a ONE-rank group still takes the sharded path (row-restricted plans, pack, all-gather, unpack), so
    that the collective of the N > 1 path executes on a one-GPU box (RCCL with a single member; ``bench.py --force-dist``,
    ``tests/test_sharded.py::test_rccl_collective_world1``).  Off: a one-rank group runs the unsharded operator."""
    import os
    return bool(os.environ.get('TPA_SHARD_FORCE'))


def row_partition(row_weights, world):
    """Cut ``range(len(row_weights))`` into ``world`` contiguous ranges of (nearly) equal total weight.
    Returns the ``world + 1`` boundaries."""
    w = np.asarray(row_weights, dtype=np.float64)
    n = len(w)
    csum = np.concatenate([[0.], np.cumsum(w)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(csum, target, side='left'))
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return np.array(bounds, dtype=np.int64)


def restrict_plan_rows(plan, res_leg0, lo, hi):
    """Copy of a TensordotPlan whose tasks only compute the result rows with flat leg-0 index in
    ``[lo, hi)``.  Works on the host tables: ``c_off += r0*ldc``, ``m = r1-r0``, ``a_off += r0*a_rs``."""
    sub = npc.TensordotPlan()
    sub.__dict__.update(plan.__dict__)
    # the split-K tables of the FULL plan cover all rows: a sub-plan that kept them would run the whole GEMM (and its reduction) on
    # every rank through `TensordotPlan.apply`, which prefers `sk`.  The row panels are launches of their own size.
    sub.sk = None
    if plan.empty:
        return sub
    tasks, links = plan.tasks_host, plan.links_host
    new_tasks, new_links, segs = [], [], []
    for t in range(len(tasks)):
        c_off, m, n, ldc, lb, lc = (int(x) for x in tasks[t][:6])
        q = int(plan.res_qdata[t, 0])
        s0 = int(res_leg0.slices[q])
        lead = int(res_leg0.slices[q + 1]) - s0        # rows of the leading leg inside this block
        inner = m // lead                               # further kept legs of `a` fused into m
        r0, r1 = max(lo, s0) - s0, min(hi, s0 + lead) - s0
        if r1 <= r0:
            continue
        r0m, r1m = r0 * inner, r1 * inner
        row = tasks[t].copy()
        row[0] = c_off + r0m * ldc
        row[1] = r1m - r0m
        row[4] = len(new_links)
        row[5] = lc
        for l in range(lb, lb + lc):
            lk = links[l].copy()
            lk[0] += r0m * lk[3]
            new_links.append(lk)
        new_tasks.append(row)
        segs.append((c_off + r0m * ldc, (r1m - r0m) * ldc))
    sub.segments = segs                                 # contiguous pieces of the result arena this plan writes
    if not new_tasks:
        sub.n_tiles = 0
        sub.local_empty = True
        return sub
    sub.local_empty = False
    new_tasks = np.array(new_tasks, dtype=np.int64)
    new_links = np.array(new_links, dtype=np.int64)
    bm, bn = npc._gemm_tile(plan.dtype, plan.cfg)
    tm, tn = (new_tasks[:, 1] + bm - 1) // bm, (new_tasks[:, 2] + bn - 1) // bn
    ntile = tm * tn
    t_task = np.repeat(np.arange(len(new_tasks)), ntile)
    local = np.arange(int(np.sum(ntile))) - np.repeat(np.cumsum(ntile) - ntile, ntile)
    tiles = np.zeros((len(t_task), 4), dtype=np.int32)
    tiles[:, 0], tiles[:, 1], tiles[:, 2] = t_task, local // np.repeat(tn, ntile), local % np.repeat(tn, ntile)
    sub.n_tiles = len(tiles)
    sub.tasks_host, sub.links_host = new_tasks, new_links
    sub.tasks_dev, sub.links_dev, sub.tiles_dev = dev.to_device_packed(new_tasks, new_links, tiles)
    # split-K of its OWN (round 6): a row panel has 1 / world of the tiles of the full launch -- the few-tiles / long-chains case the
    # plan-level split was made for (the sharded path at world 1 ran step 2 of the matvec unsplit: 0.16 s per chi = 2048 sweep)
    sub.sk = npc._split_k(sub, new_tasks, new_links, tm, tn)
    return sub


class ShardedTwoSiteH(TwoSiteH):
    """TwoSiteH whose matvec is sharded over ``torch.distributed`` ranks by rows of theta'."""

    def __init__(self, env, i0, combine=True, move_right=True, group=None):
        super().__init__(env, i0, combine, move_right)      # factored when the MPO allows it, else row panels of LHeff
        d = _dist()
        self.group = group
        self.world = d.get_world_size(group)
        self.rank = d.get_rank(group)
        self._sharded = None

    def _build_sharded(self, theta):
        p1, _, _ = npc.plan_tensordot(self.LHeff, theta, axes=['(vR.p0*)', '(vL.p0)'])
        tmp_full = npc.Array([self.LHeff.legs[0], self.LHeff.legs[1], theta.legs[1]], p1.dtype)
        if p1.empty:
            return None
        tmp_full._qdata, tmp_full._offsets = p1.res_qdata, p1.res_offsets
        tmp_full._arena = dev.empty(p1.res_total, p1.dtype)
        tmp_full.iset_leg_labels(['(vR*.p0)', 'wR', '(p1.vR)'])
        p2, _, _ = npc.plan_tensordot(tmp_full, self.RHeff, axes=(['wR', '(p1.vR)'], ['wL', '(p1*.vL)']))
        if p2.empty:
            return None
        leg0 = self.LHeff.legs[0]
        # flop weight per flat row of leg 0: sum over tasks of (chain K) * n per row, both steps
        weights = np.zeros(leg0.ind_len)
        for plan in (p1, p2):
            tasks, links = plan.tasks_host, plan.links_host
            for t in range(len(tasks)):
                q = int(plan.res_qdata[t, 0])
                s0, s1 = int(leg0.slices[q]), int(leg0.slices[q + 1])
                ksum = float(np.sum(links[int(tasks[t][4]):int(tasks[t][4]) + int(tasks[t][5]), 2]))
                inner = int(tasks[t][1]) // (s1 - s0)
                weights[s0:s1] += ksum * float(tasks[t][2]) * inner
        bounds = row_partition(weights, self.world)
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        sp1 = restrict_plan_rows(p1, leg0, lo, hi)
        sp2 = restrict_plan_rows(p2, leg0, lo, hi)
        # segments of theta' written by each rank (needed for the gather)
        all_segs = []
        for r in range(self.world):
            all_segs.append(restrict_plan_rows_segments(p2, leg0, int(bounds[r]), int(bounds[r + 1])))
        maxlen = max(sum(n for _, n in segs) for segs in all_segs)
        return dict(p1=p1, p2=p2, sp1=sp1, sp2=sp2, segs=all_segs, maxlen=max(maxlen, 1), tmp=tmp_full,
                    key=(theta._struct_key(), theta.dtype), bounds=bounds)

    # ---- factored operator: LP[rows] . theta -> (W0 W1) blockwise on the panel rows -> T3[rows] . RP ---------------
    def _build_sharded_factored(self, theta):
        from .mps_common import MpoApplyPlan
        p1, _, _ = npc.plan_tensordot(self._LPf, theta, axes=['vR', 'vL'])
        if p1.empty:
            return None
        T1 = p1.apply(self._LPf, theta)                     # full product once per bond: fixes the block structure
        a01 = MpoApplyPlan.get(T1, self.W0, 'wR', 'p0', 'wL', 'wR', 'p0', 'p0*', ('vR*', 'p0', 'p1', 'wR', 'vR'),
                               W2=self.W1, x_p2='p1', p2_out='p1', p2_in='p1*')
        if a01.empty:
            return None
        T3 = a01.apply(T1)
        p2, _, _ = npc.plan_tensordot(T3, self._RPf, axes=(['wR', 'vR'], ['wL', 'vL']))
        if p2.empty:
            return None
        leg0 = self._LPf.legs[0]
        weights = np.zeros(leg0.ind_len)
        for plan in (p1, p2):
            tasks, links = plan.tasks_host, plan.links_host
            for t in range(len(tasks)):
                q = int(plan.res_qdata[t, 0])
                s0, s1 = int(leg0.slices[q]), int(leg0.slices[q + 1])
                ksum = float(np.sum(links[int(tasks[t][4]):int(tasks[t][4]) + int(tasks[t][5]), 2]))
                weights[s0:s1] += ksum * float(tasks[t][2]) * (int(tasks[t][1]) // (s1 - s0))
        bounds = row_partition(weights, self.world)
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        sp1 = restrict_plan_rows(p1, leg0, lo, hi)
        sp2 = restrict_plan_rows(p2, leg0, lo, hi)
        # the MPO application restricted to the panel rows: blocks are (n_a x ... x n_b) with the bond index a slowest
        jobs, terms = a01.jobs_host.copy(), a01.terms_host.copy()
        blk_of_job = np.searchsorted(a01.offsets, jobs[:, 0], side='right') - 1
        keep = np.zeros(len(jobs), dtype=bool)
        for jdx in range(len(jobs)):
            b = int(blk_of_job[jdx])
            q = int(a01.qdata[b, 0])
            s0 = int(leg0.slices[q])
            na = int(leg0.slices[q + 1]) - s0
            rowlen = int(a01.sizes[b]) // na
            r0, r1 = max(lo, s0) - s0, min(hi, s0 + na) - s0
            if r1 <= r0:
                continue
            keep[jdx] = True
            jobs[jdx, 0] += r0 * rowlen
            jobs[jdx, 2] = jobs[jdx, 3] = (r1 - r0) * rowlen
            tb, tc = int(jobs[jdx, 4]), int(jobs[jdx, 5])
            terms[tb:tb + tc, 0] += r0 * rowlen
            terms[tb:tb + tc, 1] = (r1 - r0) * rowlen
        jobs = np.ascontiguousarray(jobs[keep])
        all_segs = [restrict_plan_rows_segments(p2, leg0, int(bounds[r]), int(bounds[r + 1])) for r in range(self.world)]
        maxlen = max(sum(n for _, n in segs) for segs in all_segs)
        nT1, nT3 = int(T1._arena.numel()), int(T3._arena.numel())
        T1._arena = T3._arena = None      # only their block structure is kept; the data live in scratch (`_mid`)
        return dict(p1=p1, p2=p2, sp1=sp1, sp2=sp2, a01=a01, T1=T1, T3=T3, nT1=nT1, nT3=nT3, segs=all_segs, maxlen=max(maxlen, 1),
                    key=(theta._struct_key(), theta.dtype), lkey=self._LPf._struct_key(), rkey=self._RPf._struct_key(),
                    world=(self.world, self.rank), bounds=bounds, n_lin=len(jobs),
                    lin_jobs=dev.to_device(jobs) if len(jobs) else None, lin_terms=dev.to_device(terms),
                    lin_max=int(np.max(jobs[:, 2])) if len(jobs) else 0)

    @staticmethod
    def _mid(s):
        """The two intermediates of the factored matvec as scratch arenas (the tables of a bond are kept across sweeps, its
        2 x 150 MB of intermediates are not); the Arrays ``T1`` / ``T3`` of the plan dictionary are re-pointed to them."""
        dt = s['p2'].dtype
        t1, t3 = dev.scratch('shard_t1', s['nT1'], dt), dev.scratch('shard_t3', s['nT3'], dt)
        s['T1']._arena, s['T3']._arena = t1, t3
        return t1, t3

    def _matvec_sharded_factored(self, theta):
        if self._sharded is None or self._sharded['key'] != (theta._struct_key(), theta.dtype):
            self._sharded = self._build_sharded_factored(theta)
            if self._sharded is None:
                return super().matvec(theta)
            s = self._sharded
            self.flops_per_matvec = s['p1'].flops + s['p2'].flops
            self.bytes_per_matvec = s['p1'].bytes_min + s['p2'].bytes_min + s['a01'].bytes
        s = self._sharded
        self._mid(s)
        T1, T3 = s['T1'], s['T3']
        if not s['sp1'].local_empty:
            s['sp1'].apply(self._LPf, theta, out_arena=T1._arena)
        if s['n_lin']:
            dev.check(dev.lib().tpa_lincomb_batch(dev.code(T3.dtype), s['lin_jobs'].data_ptr(), s['n_lin'], s['lin_terms'].data_ptr(),
                                                  s['lin_max'], T1._arena.data_ptr(), T3._arena.data_ptr(), dev.stream()), "lincomb")
        out_arena = dev.empty(s['p2'].res_total, s['p2'].dtype)
        res = None
        if not s['sp2'].local_empty:
            res = s['sp2'].apply(T3, self._RPf, out_arena=out_arena)
        if res is None:
            res = npc.Array(list(theta.legs), s['p2'].dtype, theta.qtotal)
            res._qdata, res._offsets, res._arena = s['p2'].res_qdata, s['p2'].res_offsets, out_arena
            res._qdata_sorted = True
        self._gather_rows(s, out_arena)
        res.iset_leg_labels(['vL', 'p0', 'p1', 'vR'])
        return res

    def _gather_jobs(self, s):
        """Pack / unpack tables of the all-gather as batched-copy jobs (built once per plan, kept on the device): packing
        this rank's segments and scattering the other ranks' segments are ONE `tpa_copy_batch` launch each instead of
        one small copy per (block, rank)."""
        if 'pack_dev' not in s:
            segs, maxlen = s['segs'], s['maxlen']
            mine = segs[self.rank]
            n_mine = np.array([n for _, n in mine], dtype=np.int64)
            at = np.concatenate([[0], np.cumsum(n_mine)[:-1]]) if len(mine) else np.zeros(0, np.int64)
            pack = npc._copy_jobs_contiguous(at, np.array([o for o, _ in mine], dtype=np.int64), n_mine)
            dst, src, nn = [], [], []
            for r in range(self.world):
                if r == self.rank:
                    continue
                a = r * maxlen
                for off, n in segs[r]:
                    dst.append(off)
                    src.append(a)
                    nn.append(n)
                    a += n
            unpack = npc._copy_jobs_contiguous(np.array(dst, dtype=np.int64), np.array(src, dtype=np.int64), np.array(nn, dtype=np.int64))
            s['pack_dev'] = (dev.to_device(pack) if len(pack) else None, len(pack), int(n_mine.max()) if len(mine) else 0)
            s['unpack_dev'] = (dev.to_device(unpack) if len(unpack) else None, len(unpack), int(max(nn)) if nn else 0)
        return s['pack_dev'], s['unpack_dev']

    def _gather_rows(self, s, out_arena):
        """The one collective of a matvec: all-gather of the row panels (padded to the largest share)."""
        (pk, n_pk, mx_pk), (up, n_up, mx_up) = self._gather_jobs(s)
        L = dev.lib()
        code = dev.code(s['p2'].dtype)
        send = dev.empty(s['maxlen'], s['p2'].dtype)
        if n_pk:
            dev.check(L.tpa_copy_batch(code, pk.data_ptr(), n_pk, mx_pk, out_arena.data_ptr(), send.data_ptr(), dev.stream()), "pack")
        recv = dev.empty(s['maxlen'] * self.world, s['p2'].dtype)
        if np.dtype(s['p2'].dtype).kind == 'c':       # RCCL has no complex type: ship interleaved (re, im) doubles
            import torch
            f64 = torch.float64
            _dist().all_gather_into_tensor(recv.view(f64), send.view(f64), group=self.group)
        else:
            _dist().all_gather_into_tensor(recv, send, group=self.group)
        if n_up:
            dev.check(L.tpa_copy_batch(code, up.data_ptr(), n_up, mx_up, recv.data_ptr(), out_arena.data_ptr(), dev.stream()), "unpack")

    def matvec_program(self, theta):
        """The sharded matvec as a launch program for ``tpa_lanczos_run`` (round 4, VERDICT r3 task 5): the row-restricted GEMM plans of
        this rank, the pack / unpack copies of the row panels (op kind 2) and the all-gather as the program's collective (op kind 3,
        enqueued from the host callback on the launch stream), so that N > 1 runs the same single-call Lanczos as N = 1.  Returns
        ``(ops, bufs, gemm_plans, collective)`` or ``None`` (the step-by-step loop with :meth:`matvec` is the fallback)."""
        if self.world == 1 and not _forced():
            return super().matvec_program(theta)
        want = ['vL', 'p0', 'p1', 'vR'] if self.factored else ['(vL.p0)', '(p1.vR)']
        if list(theta.get_leg_labels()) != want or theta.stored_blocks == 0 or not theta._is_packed():
            return None
        key = (theta._struct_key(), theta.dtype)
        prog = self.__dict__.get('_program')
        if prog is not None and prog[0] == key:
            return prog[1]
        if self._sharded is None or self._sharded['key'] != key:
            # tables of this bond's previous visits (`_sharded_cache`: a dictionary owned by the engine, keyed by the block structure of
            # the vector -- a bond whose theta has to be embedded in the structure of H theta asks twice per visit), valid while the
            # environments keep their shape
            cache = self.__dict__.get('_sharded_cache') if self.factored else None
            s = cache.get(key) if cache is not None else None
            if s is not None and (s.get('lkey') != self._LPf._struct_key() or s.get('rkey') != self._RPf._struct_key()
                                  or s.get('world') != (self.world, self.rank)):
                s = None
            if s is None:
                s = self._build_sharded_factored(theta) if self.factored else self._build_sharded(theta)
                if cache is not None and s is not None:
                    if len(cache) >= 4:
                        cache.clear()
                    cache[key] = s
            self._sharded = s
            if self._sharded is not None:
                s = self._sharded
                self.flops_per_matvec = s['p1'].flops + s['p2'].flops
                self.bytes_per_matvec = s['p1'].bytes_min + s['p2'].bytes_min + (s['a01'].bytes if self.factored else 0)
        s = self._sharded
        res = None
        self.__dict__['_program_out'] = None
        if s is not None:
            p2 = s['p2']
            same = (p2.res_total == theta._arena.numel() and np.array_equal(p2.res_qdata, theta._qdata)
                    and np.array_equal(p2.res_offsets, theta._offsets))
            left, right = (self._LPf, self._RPf) if self.factored else (self.LHeff, self.RHeff)
            if not same:
                self.__dict__['_program_out'] = (key, p2.res_qdata, p2.res_offsets, p2.res_total)
            elif p2.dtype == theta.dtype == left.dtype == right.dtype:
                (pk, n_pk, mx_pk), (up, n_up, mx_up) = self._gather_jobs(s)
                dt = p2.dtype
                send = dev.scratch('shard_send', s['maxlen'], dt)
                recv = dev.scratch('shard_recv', s['maxlen'] * self.world, dt)
                mid = list(self._mid(s)) if self.factored else [s['tmp']._arena]
                bufs = [left._arena, right._arena] + mid + [send, recv]
                i_send, i_recv = len(bufs) - 2, len(bufs) - 1
                ops = []
                sp1, sp2 = s['sp1'], s['sp2']
                if not sp1.local_empty:
                    ops += _gemm_ops(sp1, 0, -1, 2, bufs, 'shard_sk1')
                last_mid = 2
                if self.factored:
                    if s['n_lin']:
                        ops.append([1, 0, s['lin_jobs'].data_ptr(), s['lin_terms'].data_ptr(), 0, s['n_lin'], 2, 0, 3, s['lin_max'], 0, 0])
                    last_mid = 3
                if not sp2.local_empty:
                    ops += _gemm_ops(sp2, last_mid, 1, -2, bufs, 'shard_sk2')
                if n_pk:
                    ops.append([2, 0, pk.data_ptr(), 0, 0, n_pk, -2, 0, i_send, mx_pk, 0, 0])
                ops.append([3, 0, 0, 0, 0, 0, i_send, 0, i_recv, 0, 0, 0])
                if n_up:
                    ops.append([2, 0, up.data_ptr(), 0, 0, n_up, i_recv, 0, -2, mx_up, 0, 0])
                group, cplx = self.group, np.dtype(dt).kind == 'c'

                def collective(which, user, send=send, recv=recv, group=group, cplx=cplx):
                    try:
                        if cplx:       # RCCL has no complex type: ship interleaved (re, im) doubles
                            import torch
                            _dist().all_gather_into_tensor(recv.view(torch.float64), send.view(torch.float64), group=group)
                        else:
                            _dist().all_gather_into_tensor(recv, send, group=group)
                        return 0
                    except BaseException:      # must not unwind through the C frame
                        import traceback
                        traceback.print_exc()
                        return 1
                collective.agree = lambda failed, group=group: _agree_failure(failed, group)
                res = (np.array(ops, dtype=np.int64), bufs, (), collective)
        self.__dict__['_program'] = (key, res)
        return res

    def matvec(self, theta):
        if self.world == 1 and not _forced():
            return super().matvec(theta)
        if self.factored:
            if theta.rank == 2:
                return self.prepare_svd(self._matvec_sharded_factored(self.combine_theta(theta)))
            return self._matvec_sharded_factored(theta)
        if self._sharded is None or self._sharded['key'] != (theta._struct_key(), theta.dtype):
            self._sharded = self._build_sharded(theta)
            if self._sharded is None:
                return super().matvec(theta)
            s = self._sharded
            self.flops_per_matvec = s['p1'].flops + s['p2'].flops
            self.bytes_per_matvec = s['p1'].bytes_min + s['p2'].bytes_min
        s = self._sharded
        tmp = s['tmp']
        if not s['sp1'].local_empty:
            s['sp1'].apply(self.LHeff, theta, out_arena=tmp._arena)
        out_arena = dev.empty(s['p2'].res_total, s['p2'].dtype)
        res = None
        if not s['sp2'].local_empty:
            res = s['sp2'].apply(tmp, self.RHeff, out_arena=out_arena)
        if res is None:
            res = npc.Array([self.LHeff.legs[0], self.RHeff.legs[2]], s['p2'].dtype, theta.qtotal)
            res._qdata, res._offsets, res._arena = s['p2'].res_qdata, s['p2'].res_offsets, out_arena
            res._qdata_sorted = True
        self._gather_rows(s, out_arena)
        res.iset_leg_labels(['(vL.p0)', '(p1.vR)'])
        return res


def restrict_plan_rows_segments(plan, res_leg0, lo, hi):
    """Only the (offset, length) segments that ``restrict_plan_rows(plan, leg, lo, hi)`` would write."""
    segs = []
    tasks = plan.tasks_host
    for t in range(len(tasks)):
        c_off, m, n, ldc = (int(x) for x in tasks[t][:4])
        q = int(plan.res_qdata[t, 0])
        s0 = int(res_leg0.slices[q])
        lead = int(res_leg0.slices[q + 1]) - s0
        inner = m // lead
        r0, r1 = max(lo, s0) - s0, min(hi, s0 + lead) - s0
        if r1 > r0:
            segs.append((c_off + r0 * inner * ldc, (r1 - r0) * inner * ldc))
    return segs


# ======================================================================================================================
# north_star's variant: Krylov vectors stored as ROW PANELS, scalars by all-reduce
# ======================================================================================================================

def _panel_tables(H, s):
    """Copy-job tables between a full theta-structured arena and the per-rank row panels (built once per plan)."""
    if 'panel' in s:
        return s['panel']
    segs, maxlen, world, rank = s['segs'], s['maxlen'], H.world, H.rank
    mine = segs[rank]
    n_mine = np.array([n for _, n in mine], dtype=np.int64)
    at = (np.concatenate([[0], np.cumsum(n_mine)[:-1]]) if len(mine) else np.zeros(0)).astype(np.int64)
    offs_mine = np.array([o for o, _ in mine], dtype=np.int64)
    pack = npc._copy_jobs_contiguous(at, offs_mine, n_mine)              # full arena -> my panel
    dst, src, nn = [], [], []
    for r in range(world):                                               # gathered panels -> full arena (all ranks)
        a = r * maxlen
        for off, n in segs[r]:
            dst.append(off)
            src.append(a)
            nn.append(n)
            a += n
    unpack = npc._copy_jobs_contiguous(np.array(dst, dtype=np.int64), np.array(src, dtype=np.int64), np.array(nn, dtype=np.int64))
    s['panel'] = dict(pack=(dev.to_device(pack) if len(pack) else None, len(pack), int(n_mine.max()) if len(mine) else 0),
                      unpack=(dev.to_device(unpack) if len(unpack) else None, len(unpack), int(max(nn)) if nn else 0),
                      n_local=int(n_mine.sum()))
    return s['panel']


class RowPanelOps:
    """What ``RowPanelLanczos`` needs from a :class:`ShardedTwoSiteH`: vectors are 1-D device tensors holding only the rows of
    theta this rank owns (the segments of the result arena its row-restricted plans write)."""

    def __init__(self, H, theta0):
        self.H = H
        # the plans of H are built for the block structure of the vector they are first applied to; the Krylov vectors must
        # all share ONE structure, that of the images of H: apply H until the structure stops growing (at once, except for
        # the first updates from a product state)
        cur = theta0
        for _ in range(6):
            full = H.matvec(cur)
            if full._same_structure(cur) and full.dtype == cur.dtype:
                break
            cur = full
        else:
            raise ValueError("block structure of H^n theta does not settle")
        self.s = H._sharded
        if self.s is None:
            raise ValueError("operator has no sharded plans (empty contraction)")
        self.template = full
        self.dtype = full.dtype
        self.code = dev.code(self.dtype)
        self.t = _panel_tables(H, self.s)

    def pack(self, arena):
        pk, n_pk, mx = self.t['pack']
        panel = dev.zeros(max(self.t['n_local'], 1), self.dtype)
        if n_pk:
            dev.check(dev.lib().tpa_copy_batch(self.code, pk.data_ptr(), n_pk, mx, arena.data_ptr(), panel.data_ptr(), dev.stream()), "pack")
        return panel

    def embed(self, theta):
        """Arena with the block structure of the operator's images holding ``theta`` (missing blocks zero)."""
        tpl = self.template
        if theta._same_structure(tpl) and theta.dtype == self.dtype:
            return theta._arena
        lay = npc.Array(list(tpl.legs), self.dtype, tpl.qtotal)
        lay._qdata, lay._offsets = tpl._qdata, tpl._offsets
        arena = dev.zeros(self.s['p2'].res_total, self.dtype)
        npc._scatter_blocks(theta if theta.dtype == self.dtype else theta.astype(self.dtype), lay, arena)
        return arena

    def gather(self, panel):
        """ONE all-gather: the full theta-structured Array from everybody's row panels."""
        s, H = self.s, self.H
        send = dev.zeros(s['maxlen'], self.dtype)
        n = self.t['n_local']
        if n:
            send[:n].copy_(panel[:n])
        recv = dev.empty(s['maxlen'] * H.world, self.dtype)
        if np.dtype(self.dtype).kind == 'c':
            import torch
            _dist().all_gather_into_tensor(recv.view(torch.float64), send.view(torch.float64), group=H.group)
        else:
            _dist().all_gather_into_tensor(recv, send, group=H.group)
        arena = dev.zeros(s['p2'].res_total, self.dtype)
        up, n_up, mx = self.t['unpack']
        if n_up:
            dev.check(dev.lib().tpa_copy_batch(self.code, up.data_ptr(), n_up, mx, recv.data_ptr(), arena.data_ptr(), dev.stream()), "unpack")
        tpl = self.template
        res = npc.Array(list(tpl.legs), self.dtype, tpl.qtotal, list(tpl._labels))
        res._qdata, res._offsets, res._arena, res._qdata_sorted = tpl._qdata, tpl._offsets, arena, tpl._qdata_sorted
        return res

    def matvec(self, panel):
        """Row panel of ``H v`` from the row panel of ``v``: all-gather of v, then the row-local GEMM steps."""
        H, s = self.H, self.s
        theta = self.gather(panel)
        out_arena = dev.zeros(s['p2'].res_total, self.dtype)
        if H.factored:
            H._mid(s)
            T1, T3 = s['T1'], s['T3']
            if not s['sp1'].local_empty:
                s['sp1'].apply(H._LPf, theta, out_arena=T1._arena)
            if s['n_lin']:
                dev.check(dev.lib().tpa_lincomb_batch(dev.code(T3.dtype), s['lin_jobs'].data_ptr(), s['n_lin'], s['lin_terms'].data_ptr(),
                                                      s['lin_max'], T1._arena.data_ptr(), T3._arena.data_ptr(), dev.stream()), "lincomb")
            if not s['sp2'].local_empty:
                s['sp2'].apply(T3, H._RPf, out_arena=out_arena)
        else:
            tmp = s['tmp']
            if not s['sp1'].local_empty:
                s['sp1'].apply(H.LHeff, theta, out_arena=tmp._arena)
            if not s['sp2'].local_empty:
                s['sp2'].apply(tmp, H.RHeff, out_arena=out_arena)
        return self.pack(out_arena)

    def allreduce(self, values):
        """Sum of a few doubles over the ranks (the scalar all-reduce of a Lanczos step)."""
        t = dev.to_device(np.asarray(values, dtype=np.float64))
        _dist().all_reduce(t, group=self.H.group)
        return dev.to_host(t)

    def dots(self, w, others):
        """[Re <w|o> for o in others] summed over the ranks in ONE all-reduce."""
        L = dev.lib()
        n = self.t['n_local']
        out, scr = dev.reduction_buffers()
        loc = []
        for o in others:
            if n:
                dev.check(L.tpa_dot(self.code, n, w.data_ptr(), o.data_ptr(), 1, out.data_ptr(), scr.data_ptr(), dev.stream()), "dot")
                loc.append(float(dev.to_host(out[:1])[0]))
            else:
                loc.append(0.)
        return self.allreduce(loc)

    def axpy(self, y, alpha, x):
        n = self.t['n_local']
        if n:
            dev.check(dev.lib().tpa_axpy(self.code, n, float(alpha), 0., x.data_ptr(), y.data_ptr(), dev.stream()), "axpy")

    def scal(self, x, alpha):
        n = self.t['n_local']
        if n:
            dev.check(dev.lib().tpa_scal(self.code, n, float(alpha), 0., x.data_ptr(), dev.stream()), "scal")


def lanczos_row_panels(H, theta0, options):
    """Lanczos ground state of a :class:`ShardedTwoSiteH` with the Krylov vectors stored as row panels (``north_star``: "RCCL
    all-reduce over xGMI for the Lanczos inner product"): per step ONE all-gather (inside the matvec) and TWO scalar
    all-reduces, ``alpha = <w|v_k>`` and, after the local update, ``beta^2 = |w|^2``.  (The one-reduction variant
    ``beta^2 = <w|w> - alpha^2 - beta_prev^2`` was tried first and is unstable: the new Krylov vector is then normalised
    with a computed instead of its actual norm, the error feeds back through the recurrence and 10-step runs diverged --
    energies of -11 and -2634 instead of -4.96 on the L = 12 test chain.)  Same options, stopping rule and returned triple ``(E0, theta, N)`` as ``LanczosGroundState`` (krylov_based.py:584-716); every rank takes the same
    decisions because every decision is made on all-reduced numbers."""
    from ..linalg.krylov_based import LanczosGroundState
    ctl = LanczosGroundState(H, theta0, options)         # option parsing, tridiagonal bookkeeping, stopping rule
    if ctl.E_shift is not None or ctl.reortho or ctl.N_cache < ctl.N_max:
        # this variant keeps every Krylov panel and implements the plain three-term recurrence (do not ignore silently)
        raise ValueError("lanczos_row_panels does not support the options E_shift, reortho, N_cache < N_max")
    ops = RowPanelOps(H, theta0)
    v = ops.pack(ops.embed(theta0))
    (n0,) = ops.dots(v, [v])
    beta = float(np.sqrt(n0))
    if beta < ctl._cutoff:
        raise ValueError("Norm of self.psi0 too small: {0}".format(beta))
    ops.scal(v, 1. / beta)
    cache, h = [], ctl._h_krylov
    k = 0
    for k in range(ctl.N_max):
        cache.append(v)
        w = ops.matvec(v)
        (alpha,) = ops.dots(w, [v])
        beta_prev = beta
        ops.axpy(w, -alpha, v)
        if k > 0:
            ops.axpy(w, -beta_prev, cache[-2])
        (bsq,) = ops.dots(w, [w])
        beta = float(np.sqrt(max(bsq, 0.)))
        h[k, k] = alpha
        ctl._calc_result_krylov(k)
        h[k, k + 1] = h[k + 1, k] = beta
        if abs(beta) < ctl._cutoff or (k + 1 >= ctl.N_min and ctl._converged(k)):
            break
        ops.scal(w, 1. / beta)
        v = w
    N = k + 1
    E0 = ctl.Es[N - 1, 0]
    coeff = ctl._result_krylov
    psi = dev.zeros(max(ops.t['n_local'], 1), ops.dtype)
    for c, vec in zip(coeff[:N], cache[:N]):
        ops.axpy(psi, c, vec)
    (nn,) = ops.dots(psi, [psi])
    ops.scal(psi, 1. / np.sqrt(nn))
    res = ops.gather(psi)
    res.iset_leg_labels(list(theta0.get_leg_labels()) if theta0.rank == res.rank else list(res.get_leg_labels()))
    return E0, res, N


# ---- TEBD: the bonds of a half-step over the ranks (round 4) ---------------------------------------------------------------------------
from .tebd import TEBDEngine  # noqa: E402


def _broadcast_array(arr, src, group):
    """One npc Array from rank ``src`` to every rank: the integer bookkeeping (legs, ``_qdata``, offsets, labels, ``qtotal``) as a
    pickled shell without data, the arena as ONE device-to-device broadcast (RCCL; ``gloo`` in the CPU tests)."""
    dist = _dist()
    rank = dist.get_rank(group)
    box = [None]
    if rank == src:
        shell = arr.copy(deep=False)
        n = 0 if arr._arena is None else int(arr._arena.numel())
        shell._arena = None
        box[0] = (shell, n)
    dist.broadcast_object_list(box, src=src, group=group)
    shell, n = box[0]
    if rank == src:
        arena = arr._arena
    else:
        arena = dev.empty(n, shell.dtype) if n else None
    if n:
        t = arena
        import torch
        dist.broadcast(torch.view_as_real(t) if t.is_complex() else t, src=src, group=group)
    if rank == src:
        return arr
    shell._arena = arena
    return shell


class ShardedTEBDEngine(TEBDEngine):
    """Order-2 TEBD with the bonds of every half-step dealt over the ranks (the one workload of the path that shards
    without an Amdahl wall: reference ``algorithms/tebd.py:374-414`` loops over independent bonds).  Rank r decomposes bonds
    r, r + N, ... of the half-step in ONE batched device call (``TEBDEngine._decompose_bonds``), then every new ``B_L``, ``B_R`` and the
    Schmidt values travel once from their owner to all ranks (one broadcast per tensor; the deterministic kernels make the state
    bit-identical everywhere) and are committed in bond order, so norms and truncation errors accumulate in the same order as on
    one GPU."""

    def __init__(self, psi, h_bonds_dense, options, group=None):
        super().__init__(psi, h_bonds_dense, options)
        dist = _dist()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def update_bonds_batched(self, bonds, U):
        dist = _dist()
        bonds = list(bonds)
        mine = bonds[self.rank::self.world]
        failure, local = None, {}
        try:
            local = {r[0]: r for r in self._decompose_bonds(mine, U)}
        except Exception as e:       # LinAlgError after the fallback chain, out of memory, a HIP error ... on THIS rank only
            failure = e
        # agreed on BEFORE the fixed sequence of broadcasts below: a rank that raised alone would leave the others waiting forever
        if _agree_failure(failure is not None, self.group):
            raise failure if failure is not None else np.linalg.LinAlgError(
                "tenpy_amd TEBD: the decomposition of a bond failed on another rank")
        results = []
        for k, i in enumerate(bonds):
            src = k % self.world
            r = local.get(i)
            head = [None]
            if self.rank == src:
                head[0] = (np.asarray(r[1]), float(r[4]), r[5])
            dist.broadcast_object_list(head, src=src, group=self.group)       # Schmidt values, renormalisation, truncation error (host data)
            S, renorm, err = head[0]
            B_L = _broadcast_array(r[2] if r is not None else None, src, self.group)
            B_R = _broadcast_array(r[3] if r is not None else None, src, self.group)
            results.append((i, S, B_L, B_R, renorm, err))
        self._commit_bonds(results)

    def evolve_step_order2(self):
        self.options['batch_bonds'] = True        # the sharding IS the batching: whole half-steps
        super().evolve_step_order2()
