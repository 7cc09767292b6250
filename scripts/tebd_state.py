"""The synthetic state of BASELINE config 5 (``bench.py --config tebd1024``): a random MPS in right-canonical ('B') form at the full
bond dimension, so that the timed TEBD step runs at saturated chi from the first step.  Written against the ``np_conserved`` /
``charges`` API only, so the SAME code builds the state on the device mirror (bench.py) and on the reference's numpy modules
(scripts/cpu_reference_tebd.py: TeNPy's own TEBDEngine on the same state is the reference of the bench line's parity fields).
Implementation independence: the entries come from one seeded numpy generator in the block order of ``Array.from_func``
(np_conserved.py:557), and the isometry of a site is the Q of an LQ decomposition with ``pos_diag_L=True``, which makes it
unique (an LQ / QR factor is otherwise only defined up to a phase per row)."""
import numpy as np


def random_right_canonical_tensors(npc, LegCharge, LegPipe, p, L, chi, dtype, seed):
    """Bonds have min(chi, d**i, d**(L-i)) states spread evenly over the charge sectors that the fusion rules allow; built from
    the right edge, site tensor = the isometric factor of an LQ decomposition of a random (vL) x (p.vR) block matrix; the
    "Schmidt values" are random positive numbers (normalised).  Returns ``(Bs, Ss)`` with labels vL, p, vR."""
    rng = np.random.default_rng(seed)
    chinfo = p.chinfo
    d = p.ind_len
    cplx = np.dtype(dtype).kind == 'c'

    def rnd(size):
        x = rng.standard_normal(size)
        return x + 1.j * rng.standard_normal(size) if cplx else x
    vR = LegCharge.from_qflat(chinfo, [chinfo.make_valid()], qconj=-1)
    Bs, Ss = [None] * L, [None] * (L + 1)
    Ss[L] = np.ones(1)
    for i in reversed(range(L)):
        pipe = LegPipe([p, vR], qconj=-1)
        n_q = pipe.get_block_sizes()
        total = int(min(chi, d ** min(i, 30), int(np.sum(n_q))))
        sizes = np.minimum(n_q, total // len(n_q))
        for q in np.argsort(-n_q, kind='stable'):                  # hand the remainder to the sectors that still have room
            room = min(int(n_q[q] - sizes[q]), total - int(np.sum(sizes)))
            sizes[q] += max(room, 0)
        keep = sizes > 0
        vL = LegCharge.from_qind(chinfo, np.concatenate([[0], np.cumsum(sizes[keep])]), pipe.charges[keep], qconj=+1)
        M = npc.Array.from_func(rnd, [vL, pipe], dtype=dtype, qtotal=None, shape_kw='size', labels=['vL', '(p.vR)'])
        _, Q = npc.lq(M, inner_labels=['vR', 'vL'], pos_diag_L=True)
        Bs[i] = Q.split_legs(1)
        vR = Bs[i].get_leg('vL').conj()
        s = np.abs(rng.standard_normal(vR.ind_len)) + 0.1
        Ss[i] = s / np.linalg.norm(s)
    return Bs, Ss
