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
from .mps_common import TwoSiteH

__all__ = ['ShardedTwoSiteH', 'row_partition', 'restrict_plan_rows']


def _dist():
    import torch.distributed as dist
    return dist


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
    sub.tasks_dev, sub.links_dev, sub.tiles_dev = dev.to_device(new_tasks), dev.to_device(new_links), dev.to_device(tiles)
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
        return dict(p1=p1, p2=p2, sp1=sp1, sp2=sp2, a01=a01, T1=T1, T3=T3, segs=all_segs, maxlen=max(maxlen, 1),
                    key=(theta._struct_key(), theta.dtype), bounds=bounds, n_lin=len(jobs),
                    lin_jobs=dev.to_device(jobs) if len(jobs) else None, lin_terms=dev.to_device(terms),
                    lin_max=int(np.max(jobs[:, 2])) if len(jobs) else 0)

    def _matvec_sharded_factored(self, theta):
        if self._sharded is None or self._sharded['key'] != (theta._struct_key(), theta.dtype):
            self._sharded = self._build_sharded_factored(theta)
            if self._sharded is None:
                return super().matvec(theta)
            s = self._sharded
            self.flops_per_matvec = s['p1'].flops + s['p2'].flops
            self.bytes_per_matvec = s['p1'].bytes_min + s['p2'].bytes_min + s['a01'].bytes
        s = self._sharded
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

    def matvec(self, theta):
        if self.world == 1:
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
