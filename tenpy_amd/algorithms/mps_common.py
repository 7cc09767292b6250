"""Effective two-site Hamiltonian: the Lanczos matvec of the hot path.

Mirrors ``TwoSiteH`` of ``tenpy/algorithms/mps_common.py`` (:1245; ``matvec`` :1321-1348, ``combine_Heff``
:1350, ``combine_theta`` :1374, ``update_LP`` :1421, ``update_RP`` :1430) for ``combine=True``:

    |theta'> = LHeff . theta . RHeff          (two block-sparse tensordots)

MI355X-first differences:
* the leg order of ``RHeff`` is chosen so that NEITHER tensordot of the matvec needs a transpose
  (the reference calls ``itranspose`` inside ``matvec`` :1339): LHeff = [(vR*.p0), wR, (vR.p0*)],
  RHeff = [wL, (p1*.vL), (p1.vL*)];
* both contraction plans are built once per bond and replayed for every Lanczos step: one matvec is
  exactly two grouped-GEMM launches, no host planning, no allocation besides the result arena;
* ``factored`` mode (default when every block of W0, W1 is a single number): the same operator applied as
  ``LP . theta . (W0 W1) . RP`` -- two GEMM launches with a factor d fewer flops (the physical legs are not fused into
  the contracted index) plus one block-level linear combination for the two MPO tensors; LHeff / RHeff are then only
  built on demand (mixer, tests).  theta stays in its un-fused form (vL, p0, p1, vR) during the Lanczos iteration and
  is fused once for the SVD (``prepare_svd``).  Same result as the fused path to rounding (tests/test_heff.py).
"""
import numpy as np

from ..linalg import _device as dev
from ..linalg import np_conserved as npc

__all__ = ['TwoSiteH']


FUSED_HEFF = True     # tuning / test hook: False forces the generic tensordot + combine_legs construction
import os as _os
FACTORED_MATVEC = _os.environ.get('TPA_FACTORED_MATVEC', '1') != '0'   # tuning / test hook: False = always LHeff . theta . RHeff (the reference's combine=True form)
# The factored form trades d times fewer flops for more and smaller GEMMs plus one extra pass: it pays when the GEMMs are
# compute bound (Heisenberg chi=2048: 0.80 vs 1.37 ms per matvec) and loses when they are launch bound (chi=256: 0.75 vs
# 0.54 s per sweep; Hubbard ladder chi=1024 with sectors <= 100 wide: 3.7 vs 1.5 s).  Automatic choice: largest bond sector.
FACTORED_MIN_SECTOR = 200

# ---------------------------------------------------------------------------------------------------------------------
# Fused construction of LHeff / RHeff:  LP.W0 (resp. W1.RP) + combine_legs in ONE kernel launch.
# When every charge block of the MPO tensor is a single number (all charges resolved: spin-1/2 with Sz, Hubbard with
# (N, Sz), parity-conserving TFI ...), ``tensordot(LP, W0, 'wR'-'wL')`` is a sum of scaled copies of the LP[:, w, :]
# blocks -- as a GEMM it has inner dimension 1 per link (64072 almost empty 64 x 64 tiles for a chi = 2048 Heisenberg
# bond, 1.4 ms) and is followed by a 131 MB repacking copy (combine_legs, 0.2 ms).  ``tpa_lincomb_batch`` writes every
# (w', p, p*) slab of the fused tensor directly at its place inside the pipe blocks.
def _env_set_LP(env, i, LP):
    """``env.set_LP`` with the age (number of physical sites in the part) of the reference (mps_common.py:1427)."""
    if hasattr(env, 'get_LP_age'):
        env.set_LP(i, LP, age=(env.get_LP_age(i - 1) or 0) + 1)
    else:
        env.set_LP(i, LP)


def _env_set_RP(env, i, RP):
    if hasattr(env, 'get_RP_age'):
        env.set_RP(i, RP, age=(env.get_RP_age(i + 1) or 0) + 1)
    else:
        env.set_RP(i, RP)


def _mpo_entries(W):
    """(qdata, values) of an MPO tensor whose stored blocks are all 1 x 1 x 1 x 1, else ``None`` (cached on W)."""
    ent = getattr(W, '_tpa_entries', False)
    if ent is False:
        if all(np.all(leg.get_block_sizes() == 1) for leg in W.legs) and W.stored_blocks > 0:
            vals = np.array([np.asarray(b).reshape(-1)[0] for b in W._data])
            ent = (np.array(W._qdata), vals)
        else:
            ent = None
        W._tpa_entries = ent
    return ent


def _lincomb_into(dst, src, jobs, terms, max_elems):
    if len(jobs) == 0:
        return
    L = dev.lib()
    terms_d = dev.table(terms)
    for s0 in range(0, len(jobs), 60000):
        jd = dev.table(jobs[s0:s0 + 60000])
        dev.check(L.tpa_lincomb_batch(dev.code(dst.dtype), jd.data_ptr(), len(jobs[s0:s0 + 60000]), terms_d.data_ptr(),
                                      int(max_elems), src._arena.data_ptr(), dst._arena.data_ptr(), dev.stream()), "lincomb")


from collections import OrderedDict as _OrderedDict
_heff_plans = _OrderedDict()


def _fused_heff(env_t, W, left):
    """``left``: LHeff [(vR*.p0), wR, (vR.p0*)] from LP [vR*, wR, vR] and W0 [wL, wR, p0, p0*];
    else RHeff [wL, (p1*.vL), (p1.vL*)] from RP [wL, vL, vL*] and W1 [wL, wR, p1, p1*].
    Returns (Heff, pipe) or ``None`` when the fast path does not apply (generic tensordot + combine_legs then)."""
    from ..linalg.charges import LegPipe
    ent = _mpo_entries(W)
    pl = 'p0' if left else 'p1'
    want_env = ['vR*', 'wR', 'vR'] if left else ['wL', 'vL', 'vL*']
    if ent is None or list(env_t.get_leg_labels()) != want_env or list(W.get_leg_labels()) != ['wL', 'wR', pl, pl + '*'] \
            or env_t.dtype != np.result_type(env_t.dtype, W.dtype) or env_t.stored_blocks == 0:
        return None
    wq, wv = ent
    w_axis_env = 1 if left else 0
    if not np.all(env_t.legs[w_axis_env].get_block_sizes() == 1):
        return None
    # Everything below except the two launches is integer bookkeeping that depends on the block structure and the sector charges of
    # the environment tensor and on the MPO tensor: planned once per (bond, direction), replayed on every later visit.
    pkey = (left, env_t._struct_key(), env_t.dtype.str, id(ent), env_t.qtotal.tobytes(),
            tuple((l.charges.tobytes(), int(l.qconj)) for l in env_t.legs))
    plan = _heff_plans.get(pkey)
    if plan is not None and plan['ent'] is ent:
        try:
            _heff_plans.move_to_end(pkey)
        except KeyError:      # evicted by another thread in between (the reference's `+ h.c.` worker contracts concurrently)
            pass
        res = npc.Array(plan['legs'], env_t.dtype, plan['qtotal'], plan['labels'])
        res._adopt_blocks(plan['qdata'], plan['offsets'], dev.zeros(plan['total'], env_t.dtype), True)
        dev.check(dev.lib().tpa_lincomb_batch(dev.code(env_t.dtype), plan['jobs'].data_ptr(), plan['n_jobs'], plan['terms'].data_ptr(),
                                              plan['max_elems'], env_t._arena.data_ptr(), res._arena.data_ptr(), dev.stream()), "lincomb")
        return res, plan['pipe']
    eq = env_t._qdata
    shapes = env_t._block_shapes()
    if left:
        pipe = LegPipe([env_t.get_leg('vR*'), W.get_leg('p0')], qconj=+1)
        legs = [pipe, W.get_leg('wR'), pipe.conj()]
        labels = ['(vR*.p0)', 'wR', '(vR.p0*)']
    else:
        pipe = LegPipe([W.get_leg('p1'), env_t.get_leg('vL*')], qconj=-1)
        legs = [W.get_leg('wL'), pipe.conj(), pipe]
        labels = ['wL', '(p1*.vL)', '(p1.vL*)']
    qm = pipe.q_map
    n0, n1 = pipe.legs[0].block_number, pipe.legs[1].block_number
    row_of = np.full((n0, n1), -1, dtype=np.int64)
    row_of[qm[:, 3], qm[:, 4]] = np.arange(len(qm))
    psz = pipe.get_block_sizes()
    # all (env block, W entry) pairs that share the contracted MPO index
    wc = wq[:, 0] if left else wq[:, 1]            # W index contracted with the environment
    ie, iw = np.nonzero(eq[:, w_axis_env][:, None] == wc[None, :])
    if len(ie) == 0:
        return None
    if left:      # rows (vR*, p0), cols (vR, p0*)
        r_row = row_of[eq[ie, 0], wq[iw, 2]]
        c_row = row_of[eq[ie, 2], wq[iw, 3]]
        w_out = wq[iw, 1]
        rows, cols = shapes[ie, 0], shapes[ie, 2]
    else:         # rows (p1*, vL), cols (p1, vL*)
        r_row = row_of[wq[iw, 3], eq[ie, 1]]
        c_row = row_of[wq[iw, 2], eq[ie, 2]]
        w_out = wq[iw, 0]
        rows, cols = shapes[ie, 1], shapes[ie, 2]
    Qr, Qc = qm[r_row, 2], qm[c_row, 2]
    r0, c0 = qm[r_row, 0], qm[c_row, 0]
    bq = np.stack([Qr, w_out, Qc], axis=1) if left else np.stack([w_out, Qr, Qc], axis=1)
    ubq, inv = np.unique(bq, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    order = np.lexsort(ubq.T)                       # last leg most significant, like every sorted _qdata
    rank_of = np.empty(len(order), dtype=np.int64)
    rank_of[order] = np.arange(len(order))
    ubq = ubq[order]
    blk = rank_of[inv]
    res = npc.Array(legs, env_t.dtype, env_t.chinfo.make_valid(env_t.qtotal + W.qtotal), labels)
    sizes = res._set_blocks(ubq, zero=True, qdata_sorted=True)
    ld = psz[Qc]
    dst_off = res._offsets[blk] + r0 * ld + c0
    # one job per destination slab, its terms = all contributions landing there (deterministic order)
    key = np.stack([dst_off, np.arange(len(dst_off))], axis=1)
    perm = np.lexsort((key[:, 1], key[:, 0]))
    d_sorted = dst_off[perm]
    first = np.concatenate([[True], d_sorted[1:] != d_sorted[:-1]])
    starts = np.nonzero(first)[0]
    counts = np.diff(np.concatenate([starts, [len(perm)]]))
    jobs = np.zeros((len(starts), 8), dtype=np.int64)
    pf = perm[starts]
    jobs[:, 0], jobs[:, 1], jobs[:, 2], jobs[:, 3] = d_sorted[starts], rows[pf], cols[pf], ld[pf]
    jobs[:, 4], jobs[:, 5] = starts, counts
    terms = np.zeros((len(perm), 4), dtype=np.int64)
    terms[:, 0] = env_t._offsets[ie[perm]]
    terms[:, 1] = shapes[ie[perm], 2]
    alpha = np.asarray(wv[iw[perm]], dtype=np.complex128)
    terms[:, 2] = np.ascontiguousarray(alpha.real).view(np.int64)
    terms[:, 3] = np.ascontiguousarray(alpha.imag).view(np.int64)
    _lincomb_into(res, env_t, jobs, terms, int(np.max(rows * cols)))
    if 0 < len(jobs) <= 60000:
        for arr in (res._qdata, res._offsets):
            arr.setflags(write=False)
        _heff_plans[pkey] = dict(ent=ent, legs=legs, labels=labels, qtotal=res.qtotal.copy(), qdata=res._qdata, offsets=res._offsets,
                                 total=int(res._arena.numel()), jobs=dev.table(jobs), terms=dev.table(terms), n_jobs=len(jobs),
                                 max_elems=int(np.max(rows * cols)), pipe=pipe)
        if len(_heff_plans) > 8192:
            _heff_plans.popitem(last=False)
    return res, pipe


def _reverse_columns(U):
    """Rank-2 ``U`` with the columns of every block in reversed order (one strided slab copy per column, one launch)."""
    res = npc.Array(U.legs, U.dtype, U.qtotal, list(U._labels))
    res._set_blocks(U._qdata, zero=False, qdata_sorted=U._qdata_sorted)
    if U.stored_blocks == 0:
        return res
    shapes = U._block_shapes()
    m, k = shapes[:, 0], shapes[:, 1]
    blk = np.repeat(np.arange(len(k)), k)
    col = np.arange(len(blk)) - np.repeat(np.cumsum(k) - k, k)
    jobs = np.zeros((len(blk), 8), dtype=np.int64)
    jobs[:, 0] = res._offsets[blk] + col
    jobs[:, 1], jobs[:, 2], jobs[:, 3] = m[blk], 1, k[blk]
    jobs[:, 4], jobs[:, 5] = np.arange(len(blk)), 1
    terms = np.zeros((len(blk), 4), dtype=np.int64)
    terms[:, 0] = U._offsets[blk] + (k[blk] - 1 - col)
    terms[:, 1] = k[blk]
    terms[:, 2] = np.array([1.0]).view(np.int64)[0]
    _lincomb_into(res, U, jobs, terms, int(np.max(m)))
    return res


def _relabel_view(A, labels):
    """``A`` with its legs in the order ``labels`` WITHOUT moving data: allowed when only legs with 1-wide blocks change
    their position relative to the others (the memory layout of every block is then unchanged); bookkeeping only."""
    perm = [A.get_leg_index(l) for l in labels]
    if perm == list(range(A.rank)):
        return A
    big = [a for a in range(A.rank) if not np.all(A.legs[a].get_block_sizes() == 1)]
    assert [a for a in perm if a in big] == big, "only legs with 1-wide blocks may be moved"
    B = A.copy(deep=False)
    B.legs = [A.legs[a] for a in perm]
    B._set_shape()
    B._labels = [A._labels[a] for a in perm]
    B._qdata = np.ascontiguousarray(A._qdata[:, perm])
    B._offsets = A._offsets.copy()
    B._qdata_sorted = False
    B._skey = None
    B.isort_qdata()
    return B


class MpoApplyPlan:
    """``Y[.., w', p, ..] = sum_{w, p'} W[w, w', p, p'] X[.., w, p', ..]`` for an MPO tensor whose blocks are single
    numbers: every block of Y is a linear combination of whole blocks of X (the w and p legs of X have 1-wide blocks, so
    a block of X and the blocks of Y it feeds have the same memory layout).  One ``tpa_lincomb_batch`` launch; the job
    tables are built once from the block structure of X and replayed.  With ``W2`` the two neighbouring MPO tensors
    W W2 (summed over their common bond) are applied in the same single pass.

    ``x_w``, ``x_p`` (, ``x_p2``) : labels of X's MPO leg and physical leg(s) to be contracted.
    ``w_in`` / ``w_out``          : labels of the MPO legs contracted with X / left open ('wL', 'wR' when going left to right).
    ``p_out`` / ``p_in``          : labels of W's physical legs left open / contracted ('p0', 'p0*'); ``p2_out`` / ``p2_in`` for W2.
    ``out_labels``                : order of the legs of Y (must keep the relative order of X's non-trivial legs).
    """
    _cache = {}

    @classmethod
    def get(cls, X, W, *args, W2=None, **kwargs):
        """Cached constructor: the plan depends on the block structure of X and on the MPO tensors only."""
        e1, e2 = _mpo_entries(W), (None if W2 is None else _mpo_entries(W2))
        key = (X._struct_key(), str(X.dtype), id(e1), None if e2 is None else id(e2), args,
               tuple(sorted(kwargs.items())), tuple(X.get_leg_labels()))
        plan = cls._cache.get(key)
        if plan is None or plan._entries_alive[0] is not e1 or plan._entries_alive[1] is not e2:
            if len(cls._cache) > 4096:
                cls._cache.clear()
            plan = cls._cache[key] = cls(X, W, *args, W2=W2, **kwargs)
            # the key holds id()s: keep the entry tables alive as long as the plan is cached, so that the id of a dead
            # MPO's table can never be handed to a new one (a stale plan would apply the OLD couplings)
            plan._entries_alive = (e1, e2)
        return plan

    def __init__(self, X, W, x_w, x_p, w_in, w_out, p_out, p_in, out_labels, W2=None, x_p2=None, p2_out=None, p2_in=None):
        ent = _mpo_entries(W)
        assert ent is not None
        wq, wv = ent
        wl = list(W.get_leg_labels())
        ci, co, po, pi = wl.index(w_in), wl.index(w_out), wl.index(p_out), wl.index(p_in)
        xa_w, xa_p = X.get_leg_index(x_w), X.get_leg_index(x_p)
        assert np.all(X.legs[xa_w].get_block_sizes() == 1) and np.all(X.legs[xa_p].get_block_sizes() == 1)
        # entry table: (w_in, w_out, p_out, p_in [, p2_out, p2_in]) -> value
        e_in_w, e_out_w, e_po, e_pi, e_val = wq[:, ci], wq[:, co], wq[:, po], wq[:, pi], np.asarray(wv)
        e_p2o = e_p2i = None
        w_leg_out, xa_p2 = W.get_leg(w_out), None
        if W2 is not None:
            wq2, wv2 = _mpo_entries(W2)
            wl2 = list(W2.get_leg_labels())
            ci2, co2, po2, pi2 = wl2.index(w_in), wl2.index(w_out), wl2.index(p2_out), wl2.index(p2_in)
            xa_p2 = X.get_leg_index(x_p2)
            assert np.all(X.legs[xa_p2].get_block_sizes() == 1)
            i1, i2 = np.nonzero(e_out_w[:, None] == wq2[:, ci2][None, :])        # join over the common MPO bond
            keys = np.stack([e_in_w[i1], wq2[i2, co2], e_po[i1], e_pi[i1], wq2[i2, po2], wq2[i2, pi2]], axis=1)
            vals = e_val[i1] * np.asarray(wv2)[i2]
            uk, inv = np.unique(keys, axis=0, return_inverse=True)
            tot = np.zeros(len(uk), dtype=vals.dtype)
            np.add.at(tot, inv.reshape(-1), vals)
            keep = tot != 0
            uk, tot = uk[keep], tot[keep]
            e_in_w, e_out_w, e_po, e_pi, e_p2o, e_p2i, e_val = uk[:, 0], uk[:, 1], uk[:, 2], uk[:, 3], uk[:, 4], uk[:, 5], tot
            w_leg_out = W2.get_leg(w_out)
        xq = X._qdata
        match = (xq[:, xa_w][:, None] == e_in_w[None, :]) & (xq[:, xa_p][:, None] == e_pi[None, :])
        if W2 is not None:
            match &= (xq[:, xa_p2][:, None] == e_p2i[None, :])
        ib, ie = np.nonzero(match)
        oq = xq[ib].copy()
        oq[:, xa_w] = e_out_w[ie]
        oq[:, xa_p] = e_po[ie]
        legs = list(X.legs)
        legs[xa_w] = w_leg_out
        legs[xa_p] = W.get_leg(p_out)
        labels = list(X.get_leg_labels())
        labels[xa_w], labels[xa_p] = w_out, p_out
        if W2 is not None:
            oq[:, xa_p2] = e_p2o[ie]
            legs[xa_p2] = W2.get_leg(p2_out)
            labels[xa_p2] = p2_out
        wv = e_val
        perm = [labels.index(l) for l in out_labels]
        big = [a for a in range(X.rank) if not np.all(legs[a].get_block_sizes() == 1)]
        assert [a for a in perm if a in big] == big, "out_labels must keep the order of the non-trivial legs"
        oq = oq[:, perm]
        self.legs = [legs[a] for a in perm]
        self.labels = list(out_labels)
        self.dtype = np.result_type(X.dtype, W.dtype) if W2 is None else np.result_type(X.dtype, W.dtype, W2.dtype)
        self.qtotal = X.chinfo.make_valid(X.qtotal + W.qtotal + (0 if W2 is None else W2.qtotal))
        self.empty = len(ib) == 0
        if self.empty:
            return
        uq, inv = np.unique(oq, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        order = np.lexsort(uq.T)
        rank_of = np.empty(len(order), dtype=np.int64)
        rank_of[order] = np.arange(len(order))
        self.qdata = np.ascontiguousarray(uq[order], dtype=np.intp)
        blk = rank_of[inv]
        sizes_x = X._block_sizes_flat()
        proto = npc.Array(self.legs, self.dtype, self.qtotal, self.labels)
        sizes = proto._set_blocks(self.qdata, arena=dev.empty(0, self.dtype), qdata_sorted=True)
        self.offsets = proto._offsets
        self.total = int(np.sum(sizes))
        key = np.lexsort((np.arange(len(blk)), blk))
        b_sorted = blk[key]
        first = np.concatenate([[True], b_sorted[1:] != b_sorted[:-1]])
        starts = np.nonzero(first)[0]
        counts = np.diff(np.concatenate([starts, [len(key)]]))
        jobs = np.zeros((len(starts), 8), dtype=np.int64)
        sz = sizes[b_sorted[starts]]
        jobs[:, 0], jobs[:, 1], jobs[:, 2], jobs[:, 3] = self.offsets[b_sorted[starts]], 1, sz, sz
        jobs[:, 4], jobs[:, 5] = starts, counts
        terms = np.zeros((len(key), 4), dtype=np.int64)
        terms[:, 0] = X._offsets[ib[key]]
        terms[:, 1] = sizes_x[ib[key]]
        alpha = np.asarray(wv[ie[key]], dtype=np.complex128)
        terms[:, 2] = np.ascontiguousarray(alpha.real).view(np.int64)
        terms[:, 3] = np.ascontiguousarray(alpha.imag).view(np.int64)
        assert len(jobs) <= 60000
        self.jobs_host, self.terms_host, self.sizes = jobs, terms, sizes      # (row-restricted copies: algorithms/sharded.py)
        self.jobs_dev, self.terms_dev = dev.to_device(jobs), dev.to_device(terms)
        self.n_jobs, self.max_elems = len(jobs), int(np.max(sz))
        self.x_key = X._struct_key()
        self.bytes = 8 * (2 if self.dtype.kind == 'c' else 1) * (self.total + int(np.sum(sizes_x[ib])))

    def apply(self, X, launch=True):
        res = npc.Array(self.legs, self.dtype, self.qtotal, self.labels)
        if self.empty:
            return res
        if X.dtype != self.dtype:
            X = X.astype(self.dtype)
        res._set_blocks(self.qdata, arena=dev.empty(self.total, self.dtype), qdata_sorted=True)
        if not launch:
            return res
        dev.check(dev.lib().tpa_lincomb_batch(dev.code(self.dtype), self.jobs_dev.data_ptr(), self.n_jobs, self.terms_dev.data_ptr(),
                                              self.max_elems, X._arena.data_ptr(), res._arena.data_ptr(), dev.stream()), "lincomb")
        return res


def factored_matvec_possible(LP, RP, W0, W1):
    """Whether ``LP . theta . (W0 W1) . RP`` can run as GEMM / block linear combination / GEMM on the tensors as they are stored:
    every MPO block a single number, MPO bond legs of the environments resolved into 1-wide blocks, leg orders that need no
    transposed copy.  ``W0`` / ``W1`` with labels ``p`` or ``p0`` / ``p1``."""
    lp, rp = list(LP.get_leg_labels()), list(RP.get_leg_labels())
    w0, w1 = list(W0.get_leg_labels()), list(W1.get_leg_labels())
    return (_mpo_entries(W0) is not None and _mpo_entries(W1) is not None and
            sorted(lp) == sorted(['vR*', 'wR', 'vR']) and lp.index('vR*') < lp.index('vR') and
            sorted(rp) == sorted(['wL', 'vL', 'vL*']) and rp.index('vL') < rp.index('vL*') and
            bool(np.all(LP.get_leg('wR').get_block_sizes() == 1)) and bool(np.all(RP.get_leg('wL').get_block_sizes() == 1)) and
            w0 in (['wL', 'wR', 'p0', 'p0*'], ['wL', 'wR', 'p', 'p*']) and w1 in (['wL', 'wR', 'p1', 'p1*'], ['wL', 'wR', 'p', 'p*']))


def _gemm_ops(plan, a_slot, b_slot, c_slot, bufs, scratch_name):
    """Rows of a `tpa_lanczos_run` program for one planned contraction: the grouped GEMM, or -- split-K plans, `npc._split_k` --
    the GEMM of the partial blocks into a scratch arena (appended to ``bufs``) followed by their reduction into ``c_slot``
    (kind 1 with cfg = 1: timed together with the GEMM it completes)."""
    sk = plan.sk
    if sk is None:
        return [[0, plan.cfg, plan.tasks_dev.data_ptr(), plan.links_dev.data_ptr(), plan.tiles_dev.data_ptr(), plan.n_tiles,
                 a_slot, b_slot, c_slot, 0, 0, 0]]
    bufs.append(dev.scratch(scratch_name, sk.total, plan.dtype))
    part = len(bufs) - 1
    return [[0, plan.cfg, sk.tasks_dev.data_ptr(), sk.links_dev.data_ptr(), sk.tiles_dev.data_ptr(), sk.n_tiles, a_slot, b_slot, part, 0, 0, 0],
            [1, 1, sk.jobs_dev.data_ptr(), sk.terms_dev.data_ptr(), 0, sk.n_jobs, part, 0, c_slot, sk.max_elems, 0, 0]]


class TwoSiteH:
    length = 2
    acts_on = ['(vL.p0)', '(p1.vR)']

    def __init__(self, env, i0, combine=True, move_right=True, tensors=None, factored=None):
        if not combine:     # (TeNPy's own TwoSiteH runs combine=False on the mirror: tenpy_amd/install.py)
            raise NotImplementedError("tenpy_amd.TwoSiteH (stand-alone driver): only combine=True")
        self.i0 = i0
        self.combine = combine
        self.move_right = move_right
        if tensors is not None:      # explicit (LP, RP, W0, W1), e.g. replaying a dumped bond
            self.LP, self.RP, W0, W1 = tensors
            self.dtype = W0.dtype
        else:
            self.LP = env.get_LP(i0)
            self.RP = env.get_RP(i0 + 1)
            W0, W1 = env.H.get_W(i0), env.H.get_W(i0 + 1)
            self.dtype = env.H.dtype
        self.W0 = W0.replace_labels(['p', 'p*'], ['p0', 'p0*'])
        self.W1 = W1.replace_labels(['p', 'p*'], ['p1', 'p1*'])
        # the host copy of the MPO entries is cached on the MPO's own tensors (one D2H read per site for the whole run)
        self.W0._tpa_entries, self.W1._tpa_entries = _mpo_entries(W0), _mpo_entries(W1)
        self._env = env if tensors is None else None
        self._LHeff = self._RHeff = None
        self._plans = None
        self._fplans = None
        if factored is None:
            want = FACTORED_MATVEC and int(np.max(self.LP.get_leg('vR').get_block_sizes())) >= FACTORED_MIN_SECTOR
        else:
            want = factored
        self.factored = bool(want) and self._factored_possible()
        if self.factored:       # pipes only (host bookkeeping); LHeff / RHeff are built on first access
            from ..linalg.charges import LegPipe
            self._LPf = _relabel_view(self.LP, ['vR*', 'wR', 'vR'])      # leg orders the two GEMM steps want (views)
            self._RPf = _relabel_view(self.RP, ['wL', 'vL', 'vL*'])
            self.pipeL = LegPipe([self.LP.get_leg('vR*'), self.W0.get_leg('p0')], qconj=+1)
            self.pipeR = LegPipe([self.W1.get_leg('p1'), self.RP.get_leg('vL*')], qconj=-1)
            self._LHeff = self._RHeff = None
            self.acts_on = ['vL', 'p0', 'p1', 'vR']
        else:
            self.combine_Heff(self._env)
        self.N = self.pipeL.ind_len * self.pipeR.ind_len
        self.flops_per_matvec = None
        self.bytes_per_matvec = None

    def _factored_possible(self):
        return factored_matvec_possible(self.LP, self.RP, self.W0, self.W1)

    # LHeff / RHeff: attributes in the fused mode, built lazily in the factored mode (mixer, tests)
    @property
    def LHeff(self):
        if self._LHeff is None:
            self._build_heff_lazily()
        return self._LHeff

    @LHeff.setter
    def LHeff(self, value):
        self._LHeff = value

    @property
    def RHeff(self):
        if self._RHeff is None:
            self._build_heff_lazily()
        return self._RHeff

    @RHeff.setter
    def RHeff(self, value):
        self._RHeff = value

    def _build_heff_lazily(self):
        pL, pR = self.pipeL, self.pipeR
        self.combine_Heff(self._env)
        pL.test_equal(self.pipeL)
        pR.test_equal(self.pipeR)
        self.pipeL, self.pipeR = pL, pR      # keep the pipe objects theta was fused with

    def combine_Heff(self, env=None):
        """LHeff = LP.W0 and RHeff = W1.RP with the (virtual, physical) legs fused into pipes.

        Reference: ``TwoSiteH.combine_Heff`` (mps_common.py:1350).  With an environment, the fused tensors are kept in
        ``env._heff_cache`` keyed by (side, site) and reused as long as the environment tensor they were built from is
        still the stored one: in a right-moving half sweep RHeff of a bond is exactly the one built when the previous
        left-moving half sweep optimised that bond (and vice versa), so only one of the two is rebuilt per update.
        All cached tensors stay in HBM (2 x 131 MB per bond at chi=2048, < 26 GB for L=100)."""
        cache = getattr(env, '_heff_cache', None) if env is not None else None
        hit = cache.get(('L', self.i0)) if cache is not None else None
        if hit is not None and hit[0] is self.LP:
            _, self.LHeff, self.pipeL = hit
        else:
            fused = _fused_heff(self.LP, self.W0, True) if FUSED_HEFF else None
            if fused is not None:
                self.LHeff, self.pipeL = fused
            else:
                LHeff = npc.tensordot(self.LP, self.W0, axes=['wR', 'wL'])        # vR*, vR, wR, p0, p0*
                self.pipeL = pipeL = LHeff.make_pipe(['vR*', 'p0'], qconj=+1)
                self.LHeff = LHeff.combine_legs([['vR*', 'p0'], ['vR', 'p0*']], pipes=[pipeL, pipeL.conj()],
                                                new_axes=[0, 2])                   # (vR*.p0), wR, (vR.p0*)
            if cache is not None:
                cache[('L', self.i0)] = (self.LP, self.LHeff, self.pipeL)
        hit = cache.get(('R', self.i0 + 1)) if cache is not None else None
        if hit is not None and hit[0] is self.RP:
            _, self.RHeff, self.pipeR = hit
        else:
            fused = _fused_heff(self.RP, self.W1, False) if FUSED_HEFF else None
            if fused is not None:
                self.RHeff, self.pipeR = fused
            else:
                RHeff = npc.tensordot(self.W1, self.RP, axes=['wR', 'wL'])        # wL, p1, p1*, vL, vL*
                self.pipeR = pipeR = RHeff.make_pipe(['p1', 'vL*'], qconj=-1)
                self.RHeff = RHeff.combine_legs([['p1*', 'vL'], ['p1', 'vL*']], pipes=[pipeR.conj(), pipeR],
                                                new_axes=[1, 2])                   # wL, (p1*.vL), (p1.vL*)
            if cache is not None:
                cache[('R', self.i0 + 1)] = (self.RP, self.RHeff, self.pipeR)

    def combine_theta(self, theta):
        """theta (vL, p0, p1, vR) -> the form the matvec acts on: the matrix [(vL.p0), (p1.vR)] (pipes of Heff), or --
        factored mode -- the un-fused tensor itself (an already fused theta is split)."""
        if self.factored:
            if theta.rank == 2:
                theta = theta.split_legs()
            return theta if list(theta.get_leg_labels()) == ['vL', 'p0', 'p1', 'vR'] else theta.transpose(['vL', 'p0', 'p1', 'vR'])
        return theta.combine_legs([['vL', 'p0'], ['p1', 'vR']], pipes=[self.pipeL, self.pipeR])

    def prepare_svd(self, theta):
        """The matrix [(vL.p0), (p1.vR)] that svd_theta / the mixer expect (fused once per bond in the factored mode)."""
        if theta.rank == 2:
            return theta
        return theta.combine_legs([['vL', 'p0'], ['p1', 'vR']], pipes=[self.pipeL, self.pipeR])

    def _matvec_factored(self, theta):
        """theta [vL, p0, p1, vR] -> LP . theta . W0 W1 . RP with the same legs:
        T1 = LP . theta (GEMM, contracted index = chi, not d chi);  T3 = MPO tensors applied blockwise (lincomb);
        theta' = T3 . RP (GEMM, chain over wR and the bond sector)."""
        fp = self._fplans
        if fp is None or fp['key'] != theta._struct_key() or fp['dtype'] != theta.dtype:
            p1, l_use, t_use = npc.plan_tensordot(self._LPf, theta, axes=['vR', 'vL'])
            assert l_use is self._LPf and t_use is theta, "factored matvec step 1 must not need a transpose"
            T1 = p1.apply(self._LPf, theta)                                    # vR*, wR, p0, p1, vR
            a01 = MpoApplyPlan.get(T1, self.W0, 'wR', 'p0', 'wL', 'wR', 'p0', 'p0*', ('vR*', 'p0', 'p1', 'wR', 'vR'),
                                   W2=self.W1, x_p2='p1', p2_out='p1', p2_in='p1*')   # both MPO tensors in ONE pass
            T3 = a01.apply(T1)
            p2, t3_use, r_use = npc.plan_tensordot(T3, self._RPf, axes=(['wR', 'vR'], ['wL', 'vL']))
            assert t3_use is T3 and r_use is self._RPf, "factored matvec step 2 must not need a transpose"
            self._fplans = dict(key=theta._struct_key(), dtype=theta.dtype, p1=p1, a01=a01, p2=p2)
            if not p1.empty and not p2.empty:
                self.flops_per_matvec = p1.flops + p2.flops
                self.bytes_per_matvec = p1.bytes_min + p2.bytes_min + a01.bytes
                self.gemm_shapes = (p1.gemm_shapes, p2.gemm_shapes)
            res = p2.apply(T3, self._RPf)
        else:
            T1 = fp['p1'].apply(self._LPf, theta)
            T3 = fp['a01'].apply(T1)
            res = fp['p2'].apply(T3, self._RPf)
        res.iset_leg_labels(['vL', 'p0', 'p1', 'vR'])
        return res

    def matvec(self, theta):
        """theta [(vL.p0), (p1.vR)] (fused mode) or [vL, p0, p1, vR] (factored mode) -> H_eff theta, same legs and labels."""
        if self.factored:
            if theta.rank == 2:      # a fused vector handed to a factored operator (tests, parity checks)
                return self.prepare_svd(self._matvec_factored(self.combine_theta(theta)))
            return self._matvec_factored(theta)
        if self._plans is None or not self._plan_matches(theta):
            p1, l_use, t_use = npc.plan_tensordot(self.LHeff, theta, axes=['(vR.p0*)', '(vL.p0)'])
            assert l_use is self.LHeff and t_use is theta, "matvec step 1 must not need a transpose"
            tmp = p1.apply(self.LHeff, theta)
            p2, t2_use, r_use = npc.plan_tensordot(tmp, self.RHeff, axes=(['wR', '(p1.vR)'], ['wL', '(p1*.vL)']))
            assert t2_use is tmp and r_use is self.RHeff, "matvec step 2 must not need a transpose"
            self._plans = (p1, p2, theta._struct_key(), theta.dtype)
            if not p1.empty and not p2.empty:
                self.flops_per_matvec = p1.flops + p2.flops
                self.bytes_per_matvec = p1.bytes_min + p2.bytes_min
                self.gemm_shapes = (p1.gemm_shapes, p2.gemm_shapes)
            res = p2.apply(tmp, self.RHeff)
        else:
            p1, p2 = self._plans[0], self._plans[1]
            tmp = p1.apply(self.LHeff, theta)
            res = p2.apply(tmp, self.RHeff)
        res.iset_leg_labels(['(vL.p0)', '(p1.vR)'])
        return res

    def matvec_program(self, theta):
        """The matvec as a replayable launch program for ``tpa_lanczos_run`` (include/tenpy_amd.h): ``(ops, bufs)`` with ``ops``
        int64 ``[n_ops, 12]`` rows ``(kind, cfg, p0, p1, p2, count, a_slot, b_slot, c_slot, max_elems, 0, 0)`` and ``bufs`` the
        list of device tensors the non-negative slots refer to (operands first, then temporaries from the scratch pool);
        slot -1 = input vector, -2 = output vector.  ``None`` when the matvec cannot be replayed on raw arenas: vector not
        in the operator's own leg order, mixed dtypes, or an output block structure different from the input's (the first
        steps from a product state, where H creates blocks)."""
        want = ['vL', 'p0', 'p1', 'vR'] if self.factored else ['(vL.p0)', '(p1.vR)']
        if list(theta.get_leg_labels()) != want or theta.stored_blocks == 0 or not theta._is_packed():
            return None
        key = (theta._struct_key(), theta.dtype)
        prog = self.__dict__.get('_program')
        if prog is not None and prog[0] == key:
            return prog[1]
        res = None
        if self.factored:
            fp = self._fplans
            if fp is not None and 'lkey' in fp and (fp['lkey'] != self._LPf._struct_key() or fp['rkey'] != self._RPf._struct_key()):
                fp = None           # (plans handed over from the previous visit of this bond, `plan_cache`: the environments changed shape)
            if fp is None or fp['key'] != theta._struct_key() or fp['dtype'] != theta.dtype:
                p1, l_use, t_use = npc.plan_tensordot(self._LPf, theta, axes=['vR', 'vL'])
                if l_use is not self._LPf or t_use is not theta or p1.empty:
                    return None
                T1 = p1.apply(self._LPf, theta, launch=False)
                a01 = MpoApplyPlan.get(T1, self.W0, 'wR', 'p0', 'wL', 'wR', 'p0', 'p0*', ('vR*', 'p0', 'p1', 'wR', 'vR'),
                                       W2=self.W1, x_p2='p1', p2_out='p1', p2_in='p1*')
                if a01.empty:
                    return None
                T3 = a01.apply(T1, launch=False)
                p2, t3_use, r_use = npc.plan_tensordot(T3, self._RPf, axes=(['wR', 'vR'], ['wL', 'vL']))
                if t3_use is not T3 or r_use is not self._RPf or p2.empty:
                    return None
                self._fplans = fp = dict(key=theta._struct_key(), dtype=theta.dtype, p1=p1, a01=a01, p2=p2,
                                         lkey=self._LPf._struct_key(), rkey=self._RPf._struct_key())
                self.flops_per_matvec = p1.flops + p2.flops
                self.bytes_per_matvec = p1.bytes_min + p2.bytes_min + a01.bytes
                self.gemm_shapes = (p1.gemm_shapes, p2.gemm_shapes)
            p1, a01, p2 = fp['p1'], fp['a01'], fp['p2']
            ok = (p1.dtype == a01.dtype == p2.dtype == theta.dtype == self._LPf.dtype == self._RPf.dtype and not p1.empty and
                  not p2.empty and not a01.empty)
            if ok:
                bufs = [self._LPf._arena, self._RPf._arena, dev.scratch('lanczos_t1', p1.res_total, p1.dtype),
                        dev.scratch('lanczos_t3', a01.total, a01.dtype)]
                ops = _gemm_ops(p1, 0, -1, 2, bufs, 'lanczos_sk1')
                ops.append([1, 0, a01.jobs_dev.data_ptr(), a01.terms_dev.data_ptr(), 0, a01.n_jobs, 2, 0, 3, a01.max_elems, 0, 0])
                ops += _gemm_ops(p2, 3, 1, -2, bufs, 'lanczos_sk2')
                res = (np.array(ops, dtype=np.int64), bufs, (p1, p2))
                last = p2
        else:
            if self._plans is None or not self._plan_matches(theta):
                p1, l_use, t_use = npc.plan_tensordot(self.LHeff, theta, axes=['(vR.p0*)', '(vL.p0)'])
                if l_use is not self.LHeff or t_use is not theta or p1.empty:
                    return None
                tmp = p1.apply(self.LHeff, theta, launch=False)
                p2, t2_use, r_use = npc.plan_tensordot(tmp, self.RHeff, axes=(['wR', '(p1.vR)'], ['wL', '(p1*.vL)']))
                if t2_use is not tmp or r_use is not self.RHeff or p2.empty:
                    return None
                self._plans = (p1, p2, theta._struct_key(), theta.dtype)
                self.flops_per_matvec = p1.flops + p2.flops
                self.bytes_per_matvec = p1.bytes_min + p2.bytes_min
                self.gemm_shapes = (p1.gemm_shapes, p2.gemm_shapes)
            p1, p2 = self._plans[0], self._plans[1]
            if p1.dtype == p2.dtype == theta.dtype == self.LHeff.dtype == self.RHeff.dtype and not p1.empty and not p2.empty:
                bufs = [self.LHeff._arena, self.RHeff._arena, dev.scratch('lanczos_t1', p1.res_total, p1.dtype)]
                ops = _gemm_ops(p1, 0, -1, 2, bufs, 'lanczos_sk1') + _gemm_ops(p2, 2, 1, -2, bufs, 'lanczos_sk2')
                res = (np.array(ops, dtype=np.int64), bufs, (p1, p2))
                last = p2
        self.__dict__['_program_out'] = None
        if res is not None and not (last.res_total == theta._arena.numel() and np.array_equal(last.res_qdata, theta._qdata)
                                    and np.array_equal(last.res_offsets, theta._offsets)):
            self.__dict__['_program_out'] = (key, last.res_qdata, last.res_offsets, last.res_total)
            res = None
        self.__dict__['_program'] = (key, res)
        return res

    def native_input(self, theta, max_pad=2):
        """``(vector, program)`` for ``tpa_lanczos_run``: ``theta`` itself, or -- when H_eff creates blocks that ``theta`` does not
        store (tiny extreme charge sectors of a state grown from a product state: the first Krylov step of the step-by-step
        loop goes the generic way for them) -- ``theta`` embedded with zero blocks in the block structure of ``H_eff theta``,
        provided that structure is closed under another application.  ``None`` if no replayable program exists."""
        vec = theta
        for _ in range(max_pad + 1):
            prog = self.matvec_program(vec)
            if prog is not None:
                return vec, prog
            out = self.__dict__.get('_program_out')
            if out is None or out[0] != (vec._struct_key(), vec.dtype):
                return None
            _, qdata, offsets, total = out
            have = {tuple(r) for r in qdata.tolist()}
            if not all(tuple(r) in have for r in vec._qdata.tolist()):
                return None                      # H_eff theta lacks blocks of theta: not an embedding
            pad = npc.Array(vec.legs, vec.dtype, vec.qtotal, vec.get_leg_labels())
            pad._set_blocks(qdata, arena=dev.zeros(total, vec.dtype), qdata_sorted=True)
            if not np.array_equal(pad._offsets, offsets):
                return None
            npc._scatter_blocks(vec, pad, pad._arena)
            vec = pad
        return None

    def _plan_matches(self, theta):
        return self._plans[2] == theta._struct_key() and self._plans[3] == theta.dtype

    def update_LP(self, env, i, U=None):
        """LP(i0+1) = U^dagger LHeff U   (reference :1421)."""
        assert i == self.i0 + 1
        if self.factored:       # LP' = A^dagger (LP . A) W0 without LHeff (reference MPOEnvironment._contract_LP, mpo.py:3087)
            A = U.split_legs(['(vL.p0)'])                                        # vL, p0, vR
            X = npc.tensordot(self.LP, A, axes=['vR', 'vL'])                     # vR*, wR, p0, vR
            X = MpoApplyPlan.get(X, self.W0, 'wR', 'p0', 'wL', 'wR', 'p0', 'p0*', ('vR*', 'p0', 'wR', 'vR')).apply(X)
            LP = npc.tensordot(A.conj(), X, axes=(['vL*', 'p0*'], ['vR*', 'p0']))   # vR*, wR, vR
            LP = _relabel_view(LP, list(self.LP.get_leg_labels()))
            _env_set_LP(env, i, LP)
            return LP
        LP = npc.tensordot(self.LHeff, U, axes=['(vR.p0*)', '(vL.p0)'])
        LP = npc.tensordot(U.conj(), LP, axes=['(vL*.p0*)', '(vR*.p0)'])      # vR*, wR, vR
        _env_set_LP(env, i, LP)
        return LP

    def update_RP(self, env, i, VH=None):
        """RP(i0) = RHeff VH VH^dagger   (reference :1430)."""
        assert i == self.i0
        if self.factored:       # RP' = (B . RP) W1 B^dagger without RHeff (reference _contract_RP, mpo.py:3097)
            B = VH.split_legs(['(p1.vR)'])                                       # vL, p1, vR
            X = npc.tensordot(B, self.RP, axes=['vR', 'vL'])                     # vL, p1, wL, vL*
            X = MpoApplyPlan.get(X, self.W1, 'wL', 'p1', 'wR', 'wL', 'p1', 'p1*', ('vL', 'wL', 'p1', 'vL*')).apply(X)
            RP = npc.tensordot(X, B.conj(), axes=(['p1', 'vL*'], ['p1*', 'vR*']))   # vL, wL, vL*
            RP = _relabel_view(RP, list(self.RP.get_leg_labels()))
            _env_set_RP(env, i, RP)
            return RP
        RP = npc.tensordot(VH, self.RHeff, axes=['(p1.vR)', '(p1*.vL)'])      # vL, wL, (p1.vL*)
        RP = npc.tensordot(RP, VH.conj(), axes=['(p1.vL*)', '(p1*.vR*)'])     # vL, wL, vL*
        _env_set_RP(env, i, RP)
        return RP

    def to_matrix_array(self):
        """The effective Hamiltonian contracted to a device matrix with legs [(vL.p0.p1.vR)-like pipe, its conj]
        (reference ``TwoSiteH.to_matrix`` :1392) -- for the exact diagonalisation of small bonds (``full_diag_effH``)."""
        contr = npc.tensordot(self.LHeff, self.RHeff, axes=['wR', 'wL'])
        return contr.combine_legs([['(vR*.p0)', '(p1.vL*)'], ['(vR.p0*)', '(p1*.vL)']], qconj=[+1, -1])

    def to_matrix(self):
        """Dense effective Hamiltonian on the host (tests only)."""
        L, R = self.LHeff.to_ndarray(), self.RHeff.to_ndarray()
        full = np.tensordot(L, R, axes=([1], [0]))           # (vR*.p0), (vR.p0*), (p1*.vL), (p1.vL*)
        full = full.transpose(0, 3, 1, 2)                    # out_L, out_R, in_L, in_R
        n = full.shape[0] * full.shape[1]
        return full.reshape(n, n)
