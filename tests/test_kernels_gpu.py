"""Raw C-ABI kernel numerics vs a plain torch fp64 reference of the same op (floating-point kernels)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from tenpy_amd import _lib
    _lib.require_gpu()
    lib = _lib.load()
    return torch, lib, _lib


def _dev(torch, arr):
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def _run_gemm(torch, lib, _lib, dtype, tasks, links, A, B, C, cfg=0):
    bm, bn = ctypes.c_int(), ctypes.c_int()
    lib.tpa_gemm_tile_shape(dtype, cfg, ctypes.byref(bm), ctypes.byref(bn))
    tiles = []
    for t, tk in enumerate(tasks):
        m, n = tk[1], tk[2]
        for i in range((m + bm.value - 1) // bm.value):
            for j in range((n + bn.value - 1) // bn.value):
                tiles.append([t, i, j, 0])
    tasks_d = _dev(torch, np.array(tasks, np.int64))
    links_d = _dev(torch, np.array(links, np.int64))
    tiles_d = _dev(torch, np.array(tiles, np.int32))
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.tpa_gemm_chain(dtype, cfg, tasks_d.data_ptr(), links_d.data_ptr(), tiles_d.data_ptr(), len(tiles),
                                  A.data_ptr(), B.data_ptr(), C.data_ptr(), st), "gemm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("cfg", [0, 1])
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 5, 7), (16, 16, 16), (37, 129, 65), (128, 128, 64), (130, 131, 17),
                                   (300, 257, 290), (543, 545, 1086)])
def test_gemm_single(env, cplx, shape, cfg):
    torch, lib, _lib = env
    m, n, k = shape
    g = torch.Generator(device="cpu").manual_seed(m * 1000 + n * 10 + k)
    dt = torch.complex128 if cplx else torch.float64
    A = torch.randn(m, k, dtype=dt, generator=g).cuda()
    B = torch.randn(k, n, dtype=dt, generator=g).cuda()
    C = torch.full((m, n), float("nan"), dtype=dt).cuda()
    links = [[0, 0, k, k, 1, n, 1, 0]]
    tasks = [[0, m, n, n, 0, 1, 0, 0]]
    _run_gemm(torch, lib, _lib, int(cplx), tasks, links, A, B, C, cfg)
    ref = A @ B
    err = (C - ref).abs().max().item()
    assert err <= 1e-13 * k * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("cfg", [0, 1])
@pytest.mark.parametrize("cplx", [False, True])
def test_gemm_strided_chain_accumulate(env, cplx, cfg):
    """transposed operands (m-fast A, k-fast B), conj flags, a 3-link chain and accumulate=1."""
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(7)
    dt = torch.complex128 if cplx else torch.float64
    m, n = 70, 45
    ks = [33, 4, 129]
    # arena A holds A0 (m x k0 row-major), A1^T (k1 x m, i.e. m-fast), A2 (row-major)
    A0 = torch.randn(m, ks[0], dtype=dt, generator=g)
    A1T = torch.randn(ks[1], m, dtype=dt, generator=g)
    A2 = torch.randn(m, ks[2], dtype=dt, generator=g)
    B0 = torch.randn(ks[0], n, dtype=dt, generator=g)
    B1T = torch.randn(n, ks[1], dtype=dt, generator=g)  # k-fast B
    B2 = torch.randn(ks[2], n, dtype=dt, generator=g)
    Aar = torch.cat([A0.reshape(-1), A1T.reshape(-1), A2.reshape(-1)]).cuda()
    Bar = torch.cat([B0.reshape(-1), B1T.reshape(-1), B2.reshape(-1)]).cuda()
    C0 = torch.randn(m, n, dtype=dt, generator=g)
    C = C0.clone().cuda()
    a_offs = [0, A0.numel(), A0.numel() + A1T.numel()]
    b_offs = [0, B0.numel(), B0.numel() + B1T.numel()]
    fl = [1, 2, 3] if cplx else [0, 0, 0]
    links = [[a_offs[0], b_offs[0], ks[0], ks[0], 1, n, 1, fl[0]],
             [a_offs[1], b_offs[1], ks[1], 1, m, 1, ks[1], fl[1]],
             [a_offs[2], b_offs[2], ks[2], ks[2], 1, n, 1, fl[2]]]
    tasks = [[0, m, n, n, 0, 3, 1, 0]]
    _run_gemm(torch, lib, _lib, int(cplx), tasks, links, Aar, Bar, C, cfg)

    def cj(x, f):
        return x.conj() if f else x
    ref = (C0 + cj(A0, fl[0] & 1) @ cj(B0, fl[0] & 2) + cj(A1T.T, fl[1] & 1) @ cj(B1T.T, fl[1] & 2)
           + cj(A2, fl[2] & 1) @ cj(B2, fl[2] & 2)).cuda()
    assert (C - ref).abs().max().item() < 1e-11


def test_gemm_row_stride_beyond_2_31_elements(env):
    """Element offsets inside ONE operand block above 2^31 (a view with a huge row stride: rows 2^29 elements apart, 21 GB arena): the
    per-thread offsets of the round-6 loop are 64-bit."""
    torch, lib, _lib = env
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * (1 << 30):
        pytest.skip("needs 22 GB of device memory")
    m, n, k, rs = 5, 70, 100, 1 << 29
    g = torch.Generator(device="cpu").manual_seed(3)
    rows = torch.randn(m, k, dtype=torch.float64, generator=g).cuda()
    A = torch.empty((m - 1) * rs + k, dtype=torch.float64, device='cuda')
    for i in range(m):
        A[i * rs:i * rs + k] = rows[i]
    B = torch.randn(k, n, dtype=torch.float64, generator=g).cuda()
    for cfg in (0, 1):
        C = torch.full((m, n), float("nan"), dtype=torch.float64).cuda()
        _run_gemm(torch, lib, _lib, 0, [[0, m, n, n, 0, 1, 0, 0]], [[0, 0, k, rs, 1, n, 1, 0]], A, B, C, cfg)
        assert (C - rows @ B).abs().max().item() < 1e-11
    del A


def test_gemm_many_tasks(env):
    torch, lib, _lib = env
    rng = np.random.default_rng(3)
    sizes = [(int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 200))) for _ in range(40)]
    As, Bs, a_off, b_off, c_off = [], [], [0], [0], [0]
    for (m, n, k) in sizes:
        As.append(rng.standard_normal((m, k)))
        Bs.append(rng.standard_normal((k, n)))
        a_off.append(a_off[-1] + m * k)
        b_off.append(b_off[-1] + k * n)
        c_off.append(c_off[-1] + m * n)
    A = _dev(torch, np.concatenate([x.ravel() for x in As]))
    B = _dev(torch, np.concatenate([x.ravel() for x in Bs]))
    C = torch.zeros(c_off[-1], dtype=torch.float64).cuda()
    links = [[a_off[i], b_off[i], k, k, 1, n, 1, 0] for i, (m, n, k) in enumerate(sizes)]
    tasks = [[c_off[i], m, n, n, i, 1, 0, 0] for i, (m, n, k) in enumerate(sizes)]
    _run_gemm(torch, lib, _lib, 0, tasks, links, A, B, C)
    Ch = C.cpu().numpy()
    for i, (m, n, k) in enumerate(sizes):
        ref = As[i] @ Bs[i]
        np.testing.assert_allclose(Ch[c_off[i]:c_off[i + 1]].reshape(m, n), ref, rtol=0, atol=1e-12 * k)


@pytest.mark.parametrize("cplx", [False, True])
def test_vec_kernels(env, cplx):
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(11)
    dt = torch.complex128 if cplx else torch.float64
    n = 100003
    x = torch.randn(n, dtype=dt, generator=g).cuda()
    y = torch.randn(n, dtype=dt, generator=g).cuda()
    z = torch.randn(n, dtype=dt, generator=g).cuda()
    out = torch.zeros(2, dtype=torch.float64).cuda()
    scr = torch.zeros(4096, dtype=torch.float64).cuda()
    st = torch.cuda.current_stream().cuda_stream
    d = int(cplx)
    for do_conj in ([0, 1] if cplx else [0]):
        _lib.check(lib.tpa_dot(d, n, x.data_ptr(), y.data_ptr(), do_conj, out.data_ptr(), scr.data_ptr(), st))
        ref = ((x.conj() if do_conj else x) * y).sum()
        got = complex(out[0].item(), out[1].item())
        assert abs(got - complex(ref.item())) < 1e-10
    _lib.check(lib.tpa_nrm2sq(d, n, x.data_ptr(), out.data_ptr(), scr.data_ptr(), st))
    assert abs(out[0].item() - (x.abs() ** 2).sum().item()) < 1e-9
    al = complex(0.3, -0.7) if cplx else 0.3
    y0 = y.clone()
    _lib.check(lib.tpa_axpy(d, n, al.real, al.imag if cplx else 0.0, x.data_ptr(), y.data_ptr(), st))
    assert (y - (y0 + al * x)).abs().max().item() < 1e-14
    _lib.check(lib.tpa_scal(d, n, al.real, al.imag if cplx else 0.0, y.data_ptr(), st))
    assert (y - al * (y0 + al * x)).abs().max().item() < 1e-14
    be = complex(-0.2, 0.1) if cplx else -0.2
    w0 = z.clone()
    _lib.check(lib.tpa_lanczos_update(d, n, z.data_ptr(), al.real, al.imag if cplx else 0.0, x.data_ptr(),
                                      be.real, be.imag if cplx else 0.0, y0.data_ptr(), out.data_ptr(),
                                      scr.data_ptr(), st))
    ref = w0 - al * x - be * y0
    assert (z - ref).abs().max().item() < 1e-14
    assert abs(out[0].item() - (ref.abs() ** 2).sum().item()) < 1e-9


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("shapes", [[(1, 1)], [(5, 3), (3, 5), (4, 4)], [(40, 17), (17, 40), (64, 64), (2, 9)],
                                    [(130, 200), (257, 129)]])
def test_svd_batch(env, cplx, shapes):
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(5 + len(shapes))
    dt = torch.complex128 if cplx else torch.float64
    mats = [torch.randn(m, n, dtype=dt, generator=g) for (m, n) in shapes]
    # make one block rank deficient / graded
    if len(mats) > 1:
        mats[1][:, 0] = mats[1][:, -1]
    jobs, a_off, u_off, s_off, v_off = [], 0, 0, 0, 0
    for (m, n) in shapes:
        k = min(m, n)
        jobs.append([a_off, m, n, u_off, s_off, v_off, 0, 0])
        a_off += m * n
        u_off += m * k
        s_off += k
        v_off += k * n
    A = torch.cat([x.reshape(-1) for x in mats]).cuda()
    U = torch.zeros(u_off, dtype=dt).cuda()
    S = torch.zeros(s_off, dtype=torch.float64).cuda()
    VH = torch.zeros(v_off, dtype=dt).cuda()
    jh = np.array(jobs, np.int64)
    wb = lib.tpa_svd_worksize(int(cplx), jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8).cuda()
    sw = ctypes.c_int()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.tpa_svd_batch(int(cplx), jh.ctypes.data, len(jobs), A.data_ptr(), U.data_ptr(), S.data_ptr(),
                                 VH.data_ptr(), work.data_ptr(), wb, 40, 0.0, ctypes.byref(sw), st), "svd")
    torch.cuda.synchronize()
    for b, (m, n) in enumerate(shapes):
        k = min(m, n)
        j = jobs[b]
        u = U[j[3]:j[3] + m * k].reshape(m, k).cpu()
        s = S[j[4]:j[4] + k].cpu()
        vh = VH[j[5]:j[5] + k * n].reshape(k, n).cpu()
        ref_s = torch.linalg.svdvals(mats[b])
        assert (s - ref_s).abs().max().item() <= 1e-13 * ref_s[0].item() * max(m, n)
        rec = (u * s.to(dt)) @ vh
        assert (rec - mats[b]).abs().max().item() < 1e-12 * ref_s[0].item() * max(m, n)
        nz = s > 1e-10 * s[0]
        un = u[:, nz]
        assert (un.conj().T @ un - torch.eye(int(nz.sum()), dtype=dt)).abs().max().item() < 1e-12
        vn = vh[nz, :]
        assert (vn @ vn.conj().T - torch.eye(int(nz.sum()), dtype=dt)).abs().max().item() < 1e-12
        assert bool((s[:-1] >= s[1:]).all())


def _svd_call(torch, lib, mats, max_sweeps=60, rho=1e-6):
    """tpa_svd_batch on a list of host matrices (all real or all complex) -> list of (u, s, vh) host tensors, return code."""
    cplx = mats[0].is_complex()
    dt = torch.complex128 if cplx else torch.float64
    jobs, a_off, u_off, s_off, v_off = [], 0, 0, 0, 0
    for x in mats:
        m, n = x.shape
        k = min(m, n)
        jobs.append([a_off, m, n, u_off, s_off, v_off, 0, 0])
        a_off, u_off, s_off, v_off = a_off + m * n, u_off + m * k, s_off + k, v_off + k * n
    A = torch.cat([x.reshape(-1) for x in mats]).cuda()
    U = torch.zeros(u_off, dtype=dt).cuda()
    S = torch.zeros(s_off, dtype=torch.float64).cuda()
    VH = torch.zeros(v_off, dtype=dt).cuda()
    jh = np.array(jobs, np.int64)
    wb = lib.tpa_svd_worksize(int(cplx), jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8).cuda()
    sw = ctypes.c_int()
    rc = lib.tpa_svd_batch(int(cplx), jh.ctypes.data, len(jobs), A.data_ptr(), U.data_ptr(), S.data_ptr(), VH.data_ptr(),
                           work.data_ptr(), wb, max_sweeps, rho, ctypes.byref(sw), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = []
    for x, j in zip(mats, jobs):
        m, n = x.shape
        k = min(m, n)
        out.append((U[j[3]:j[3] + m * k].reshape(m, k).cpu(), S[j[4]:j[4] + k].cpu(), VH[j[5]:j[5] + k * n].reshape(k, n).cpu()))
    return out, rc, sw.value


@pytest.mark.parametrize("cplx", [False, True])
def test_svd_rank_revealing_path(env, cplx):
    """Rank-deficient, graded blocks like DMRG wave functions (pivoted-QR preconditioner + Jacobi on the r x n factor)
    against torch's LAPACK SVD and against the same call with the preconditioner switched off."""
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(11)
    mats = []
    for (m, n, r) in [(300, 300, 160), (200, 333, 90), (333, 200, 200), (70, 40, 1), (1, 1, 1), (5, 90, 5), (64, 64, 0)]:
        dt = torch.complex128 if cplx else torch.float64
        u, _ = torch.linalg.qr(torch.randn(m, max(r, 1), dtype=dt, generator=g))
        v, _ = torch.linalg.qr(torch.randn(n, max(r, 1), dtype=dt, generator=g))
        sv = torch.logspace(0, -9, max(r, 1), dtype=torch.float64) * (1. if r > 0 else 0.)
        mats.append((u * sv.to(dt)) @ v.conj().T)
    res, rc, sweeps = _svd_call(torch, lib, mats)
    assert rc == 0
    lib.tpa_svd_set_algorithm(512)        # same kernels without the pivoted-QR preconditioner
    try:
        res_plain, rc_plain, sweeps_plain = _svd_call(torch, lib, mats)
    finally:
        lib.tpa_svd_set_algorithm(0)
    assert rc_plain == 0
    assert sweeps < sweeps_plain, "the preconditioner must reduce the number of Jacobi sweeps"
    for x, (u, s, vh), (u2, s2, vh2) in zip(mats, res, res_plain):
        m, n = x.shape
        ref = torch.linalg.svdvals(x)
        scale = max(ref[0].item(), 1e-300)
        assert (s - ref).abs().max().item() <= 1e-13 * scale * max(m, n)          # absolute accuracy eps ||A||
        assert (s - s2).abs().max().item() <= 1e-13 * scale * max(m, n)
        assert bool((s[:-1] >= s[1:]).all())
        assert ((u * s.to(u.dtype)) @ vh - x).abs().max().item() <= 1e-12 * scale * max(m, n)
        # vectors of sigma >= rho ||A|| (rho = 1e-6, the absolute floor of the stopping rule) are orthonormal to
        # working precision, the ones below it to ~ eps rho ||A|| / sigma  
        for thresh, tol in ((1e-6, 1e-12), (1e-12, 1e-9)):
            nz = s > thresh * scale
            k = int(nz.sum())
            if k:
                assert (u[:, nz].conj().T @ u[:, nz] - torch.eye(k, dtype=u.dtype)).abs().max().item() < tol
                assert (vh[nz] @ vh[nz].conj().T - torch.eye(k, dtype=u.dtype)).abs().max().item() < tol
        # beyond the numerical rank: exact zeros, zero vectors (documented in include/tenpy_amd.h)
        dead = s == 0
        assert u[:, dead].abs().max().item() == 0.0 if bool(dead.any()) else True


@pytest.mark.parametrize("cplx", [False, True])
def test_svd_gram_only_sweeps(env, cplx):
    """Round 4: Gram-only sweeps of the 32-row-block iteration (one exact Gram matrix per sweep, rounds on it alone, one product
    with the accumulated transform at the end) against torch's LAPACK SVD and against the same call with the round-3 rounds
    (bit 20), with and without the pivoted-QR preconditioner: same singular values, reconstruction, orthonormality, and no more
    sweeps than the rounds that touch the data.  Complex data (csrc/tpa_svd_b32c.inc): the Gram-only sweep on 32-row blocks against
    the 8-row-block rounds on the data (bit 20 switches it off)."""
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(17)
    dt = torch.complex128 if cplx else torch.float64
    mats = []
    for (m, n, r) in [(300, 300, 160), (200, 333, 90), (420, 390, 390), (70, 40, 1), (128, 257, 128), (5, 90, 5), (97, 97, 97)]:
        u, _ = torch.linalg.qr(torch.randn(m, r, dtype=dt, generator=g))
        v, _ = torch.linalg.qr(torch.randn(n, r, dtype=dt, generator=g))
        sv = torch.logspace(0, -9, r, dtype=torch.float64)
        if r > 8:
            sv[r // 3] = sv[r // 3 + 1] = sv[r // 3 + 2]
            sv[5] = sv[4] * (1 - 1e-9)
        mats.append((u * sv.to(dt)) @ v.conj().T)
    # fused rounds (default) / two launches per round (bit 23); complex data: one stream (default) / the tiles the
    # next solve does not read on a second stream (bit 14)
    for base in ((0, 16384) if cplx else (0, 512, 8388608, 512 | 8388608)):
        try:
            lib.tpa_svd_set_algorithm(base)
            res, rc, sweeps = _svd_call(torch, lib, mats)
            lib.tpa_svd_set_algorithm(base | 1048576)
            res_off, rc_off, sweeps_off = _svd_call(torch, lib, mats)
        finally:
            lib.tpa_svd_set_algorithm(0)
        assert rc == 0 and rc_off == 0
        # (round 6: with the activity-driven schedule a sweep visits only the block pairs that were active on the exact Gram matrix of
        #  its start -- a pair that other rotations activate in between waits for the next sweep, which costs a few rounds, not 17)
        assert sweeps <= sweeps_off + max(2, sweeps_off // 8), (sweeps, sweeps_off)
        for x, (u, s, vh), (u2, s2, vh2) in zip(mats, res, res_off):
            m, n = x.shape
            ref = torch.linalg.svdvals(x)
            scale = ref[0].item()
            assert (s - ref).abs().max().item() <= 1e-13 * scale * max(m, n)
            assert (s - s2).abs().max().item() <= 1e-13 * scale * max(m, n)
            assert bool((s[:-1] >= s[1:]).all())
            assert ((u * s.to(dt)) @ vh - x).abs().max().item() <= 1e-12 * scale * max(m, n)
            for thresh, tol in ((1e-6, 1e-12), (1e-12, 1e-9)):
                nz = s > thresh * scale
                k = int(nz.sum())
                if k:
                    assert (u[:, nz].conj().T @ u[:, nz] - torch.eye(k, dtype=dt)).abs().max().item() < tol
                    assert (vh[nz] @ vh[nz].conj().T - torch.eye(k, dtype=dt)).abs().max().item() < tol


def test_svd_nan_input_is_an_error(env):
    """NaN in a block: TPA_E_NAN -> ValueError like np_conserved.py:4978-4982, on both SVD paths."""
    torch, lib, _lib = env
    for n in (8, 96):                     # plain Jacobi path / pivoted-QR path
        x = torch.randn(n, n, dtype=torch.float64)
        x[n // 2, n // 3] = float("nan")
        _, rc, _ = _svd_call(torch, lib, [x, torch.randn(n, n, dtype=torch.float64)])
        assert rc == -3
        with pytest.raises(ValueError):
            _lib.check(rc, "svd")


@pytest.mark.parametrize("wy", [True, False])
@pytest.mark.parametrize("cplx", [False, True])
def test_qr_batch(env, cplx, wy):
    """wy: blocked compact-WY QR on the matrix cores (whole batch, because min(m,n) >= 32 for some block) / the
    one-workgroup Householder kernel; both follow LAPACK's reflector convention, so R agrees between them."""
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(9)
    dt = torch.complex128 if cplx else torch.float64
    shapes = [(1, 1), (7, 3), (3, 7), (50, 50), (130, 40), (33, 90), (300, 200), (200, 300), (257, 64)]
    mats = [torch.randn(m, n, dtype=dt, generator=g) for (m, n) in shapes]
    mats[3][:, 1] = 0  # zero column
    mats[6][:, 5] = mats[6][:, 4]  # linearly dependent columns
    jobs, a_off, q_off, r_off = [], 0, 0, 0
    for (m, n) in shapes:
        k = min(m, n)
        jobs.append([a_off, m, n, q_off, r_off, 0, 0, 0])
        a_off += m * n
        q_off += m * k
        r_off += k * n
    A = torch.cat([x.reshape(-1) for x in mats]).cuda()
    jh = np.array(jobs, np.int64)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in ([0, 1] if wy else [1]):
        lib.tpa_qr_set_algorithm(mode)
        try:
            Q = torch.zeros(q_off, dtype=dt).cuda()
            R = torch.zeros(r_off, dtype=dt).cuda()
            _lib.check(lib.tpa_qr_batch(int(cplx), jh.ctypes.data, len(jobs), A.data_ptr(), Q.data_ptr(), R.data_ptr(), st))
            torch.cuda.synchronize()
        finally:
            lib.tpa_qr_set_algorithm(0)
        res[mode] = (Q.cpu(), R.cpu())
    Qc, Rc = res[0 if wy else 1]
    for b, (m, n) in enumerate(shapes):
        k = min(m, n)
        j = jobs[b]
        q = Qc[j[3]:j[3] + m * k].reshape(m, k)
        r = Rc[j[4]:j[4] + k * n].reshape(k, n)
        assert (q @ r - mats[b]).abs().max().item() < 1e-12 * max(m, n)
        assert (q.conj().T @ q - torch.eye(k, dtype=dt)).abs().max().item() < 1e-13 * max(m, n)
        assert torch.tril(r, -1).abs().max().item() == 0.0
        assert r.diagonal().imag.abs().max().item() == 0.0 if cplx else True
        if wy and b not in (3, 6):      # (rank-deficient blocks: R beyond the rank is rounding noise in both)
            r_old = res[1][1][j[4]:j[4] + k * n].reshape(k, n)
            assert (r - r_old).abs().max().item() < 1e-12 * max(m, n)


def test_qr_lookahead(env):
    """Round 5: one launch per panel for tall real blocks (csrc/tpa_qr_la.inc: workgroup 0 updates and factorises the next panel while
    the others apply the current one) against the two-launches-per-panel path of rounds 2-4 and against the definition."""
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(12)
    shapes = [(50, 50), (130, 40), (300, 200), (257, 64), (1100, 600), (64, 64), (40, 33), (2048, 100), (9, 8), (35, 35)]
    mats = [torch.randn(m, n, dtype=torch.float64, generator=g) for (m, n) in shapes]
    mats[2][:, 5] = mats[2][:, 4]           # linearly dependent columns
    mats[4] = mats[4] * torch.logspace(0, -14, 600, dtype=torch.float64)        # graded columns, like the sketch of a DMRG theta
    jobs, a_off, q_off, r_off = [], 0, 0, 0
    for (m, n) in shapes:
        jobs.append([a_off, m, n, q_off, r_off, 0, 0, 0])
        a_off, q_off, r_off = a_off + m * n, q_off + m * n, r_off + n * n
    A = torch.cat([x.reshape(-1) for x in mats]).cuda()
    jh = np.array(jobs, np.int64)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in (0, 2):
        lib.tpa_qr_set_algorithm(mode)
        try:
            Q = torch.zeros(q_off, dtype=torch.float64).cuda()
            R = torch.zeros(r_off, dtype=torch.float64).cuda()
            _lib.check(lib.tpa_qr_batch(0, jh.ctypes.data, len(jobs), A.data_ptr(), Q.data_ptr(), R.data_ptr(), st))
            torch.cuda.synchronize()
        finally:
            lib.tpa_qr_set_algorithm(0)
        res[mode] = (Q.cpu(), R.cpu())
    for b, (m, n) in enumerate(shapes):
        j = jobs[b]
        for mode in (0, 2):
            q = res[mode][0][j[3]:j[3] + m * n].reshape(m, n)
            r = res[mode][1][j[4]:j[4] + n * n].reshape(n, n)
            scale = mats[b].abs().max().item()
            assert (q @ r - mats[b]).abs().max().item() < 1e-13 * max(m, n) * scale, (mode, b)
            assert (q.T @ q - torch.eye(n, dtype=torch.float64)).abs().max().item() < 1e-13 * max(m, n), (mode, b)
            assert torch.tril(r, -1).abs().max().item() == 0.0
        if b != 2:          # same arithmetic per column: R agrees to rounding (beyond the rank of block 2 it is noise in both)
            r0 = res[0][1][j[4]:j[4] + n * n].reshape(n, n)
            r2 = res[2][1][j[4]:j[4] + n * n].reshape(n, n)
            assert (r0 - r2).abs().max().item() < 1e-12 * max(m, n) * mats[b].abs().max().item(), b


@pytest.mark.parametrize("cplx", [False, True])
def test_eigh_batch(env, cplx):
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(13)
    dt = torch.complex128 if cplx else torch.float64
    ns = [1, 2, 9, 64, 150]
    mats = []
    for n in ns:
        x = torch.randn(n, n, dtype=dt, generator=g)
        mats.append(x + x.conj().T)
    mats[2] = torch.diag(torch.tensor([1., -1., 1., -1., 2., -2., 0., 0., 3.], dtype=torch.float64)).to(dt)
    jobs, a_off, w_off = [], 0, 0
    for n in ns:
        jobs.append([a_off, n, w_off, a_off, 0, 0, 0, 0])
        a_off += n * n
        w_off += n
    A = torch.cat([x.reshape(-1) for x in mats]).cuda()
    W = torch.zeros(w_off, dtype=torch.float64).cuda()
    V = torch.zeros(a_off, dtype=dt).cuda()
    jh = np.array(jobs, np.int64)
    wb = lib.tpa_eigh_worksize(int(cplx), jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8).cuda()
    sw = ctypes.c_int()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.tpa_eigh_batch(int(cplx), jh.ctypes.data, len(jobs), A.data_ptr(), W.data_ptr(), V.data_ptr(),
                                  work.data_ptr(), wb, 40, 0.0, ctypes.byref(sw), st), "eigh")
    torch.cuda.synchronize()
    for b, n in enumerate(ns):
        j = jobs[b]
        w = W[j[2]:j[2] + n].cpu()
        v = V[j[3]:j[3] + n * n].reshape(n, n).cpu()
        ref = torch.linalg.eigvalsh(mats[b])
        nrm = max(1.0, mats[b].abs().max().item()) * n
        assert (w - ref).abs().max().item() < 1e-13 * nrm
        assert (mats[b] @ v - v * w.to(dt)).abs().max().item() < 1e-12 * nrm
        assert (v.conj().T @ v - torch.eye(n, dtype=dt)).abs().max().item() < 1e-12


def test_eigh_batch_mixer_blocks(env):
    """``tpa_eigh_batch`` at the sizes the density-matrix mixer diagonalises at chi = 2048 (``mix_rho``, reference
    mps_common.py:1972-2079, through ``eigh`` np_conserved.py:3899 / ``_eig_worker`` :5041): graded positive semi-definite blocks of
    570 and 1086 rows (rho = theta theta^dagger of a wave function with 14 decades of Schmidt values), real data, against LAPACK."""
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(29)
    ns = [570, 1086]
    mats = []
    for n in ns:
        q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, generator=g))
        lam = torch.logspace(0, -14, n, dtype=torch.float64)
        x = (q * lam) @ q.T
        mats.append(0.5 * (x + x.T))
    jobs, a_off, w_off = [], 0, 0
    for n in ns:
        jobs.append([a_off, n, w_off, a_off, 0, 0, 0, 0])
        a_off += n * n
        w_off += n
    A = torch.cat([x.reshape(-1) for x in mats]).cuda()
    W = torch.zeros(w_off, dtype=torch.float64).cuda()
    V = torch.zeros(a_off, dtype=torch.float64).cuda()
    jh = np.array(jobs, np.int64)
    wb = lib.tpa_eigh_worksize(0, jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8).cuda()
    sw = ctypes.c_int()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.tpa_eigh_batch(0, jh.ctypes.data, len(jobs), A.data_ptr(), W.data_ptr(), V.data_ptr(),
                                  work.data_ptr(), wb, 60, 0.0, ctypes.byref(sw), st), "eigh")
    torch.cuda.synchronize()
    for b, n in enumerate(ns):
        j = jobs[b]
        w = W[j[2]:j[2] + n].cpu()
        v = V[j[3]:j[3] + n * n].reshape(n, n).cpu()
        ref = torch.linalg.eigvalsh(mats[b])
        # eigenvalues: absolute accuracy eps ||A|| like LAPACK's (both orders), residual and orthonormality of ALL vectors
        assert (torch.sort(w).values - ref).abs().max().item() < 1e-13 * n
        assert (v.T @ v - torch.eye(n, dtype=torch.float64)).abs().max().item() < 1e-11
        # Rounds 1 - 5 (eigh = shift + one-sided SVD) resolved eigenvalues closer than ~1e-8 |A| to each other as a SUBSPACE only (vectors
        # individually eigenvectors to 7.7e-9 |A|).  The two-sided iteration on the matrix itself (round 6) rotates until no
        # |S_ij| > eps sqrt(n) mu is left: every vector is an eigenvector to rounding level.
        res = (mats[b] @ v - v * w).abs()
        assert res.max().item() < 1e-11 * n
        big = w > 1e-6                     # ... and to rounding level where the eigenvalues are separated
        assert res[:, big].max().item() < 1e-11 * n


def test_copy_scale_gather(env):
    torch, lib, _lib = env
    rng = np.random.default_rng(0)
    src = rng.standard_normal((6, 5, 4))
    s_d = _dev(torch, src.ravel())
    dst = torch.zeros(4 * 6 * 5 + 10, dtype=torch.float64).cuda()
    # job: transpose (6,5,4)->(4,6,5) written at offset 10
    MAXD = 6
    job = np.zeros(4 + 3 * MAXD, np.int64)
    job[0], job[1], job[2], job[3] = 10, 0, 3, 0
    job[4:7] = [4, 6, 5]
    job[4 + MAXD:4 + MAXD + 3] = [30, 5, 1]
    job[4 + 2 * MAXD:4 + 2 * MAXD + 3] = [1, 20, 4]
    jd = _dev(torch, job.reshape(1, -1))
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.tpa_copy_batch(0, jd.data_ptr(), 1, 120, s_d.data_ptr(), dst.data_ptr(), st))
    got = dst.cpu().numpy()[10:].reshape(4, 6, 5)
    np.testing.assert_array_equal(got, src.transpose(2, 0, 1))
    # scale axis 1 of (6,5,4)
    sv = rng.standard_normal(7)
    sj = _dev(torch, np.array([[0, 6, 5, 4, 2, 0]], np.int64))
    x = s_d.clone()
    _lib.check(lib.tpa_scale_axis_batch(0, sj.data_ptr(), 1, 120, x.data_ptr(), _dev(torch, sv).data_ptr(), 0, st))
    np.testing.assert_allclose(x.cpu().numpy().reshape(6, 5, 4), src * sv[2:7][None, :, None], rtol=1e-15)
    # gather along axis 1
    idx = np.array([4, 0, 2], np.int64)
    gj = _dev(torch, np.array([[0, 0, 6, 5, 3, 4, 0, 0]], np.int64))
    out = torch.zeros(6 * 3 * 4, dtype=torch.float64).cuda()
    _lib.check(lib.tpa_gather_axis_batch(0, gj.data_ptr(), 1, 72, _dev(torch, idx).data_ptr(), s_d.data_ptr(),
                                         out.data_ptr(), st))
    np.testing.assert_array_equal(out.cpu().numpy().reshape(6, 3, 4), src[:, idx, :])


@pytest.mark.gpu
@pytest.mark.parametrize("cplx", [False, True])
def test_eigh_direct_two_sided_iteration(env, cplx):
    """Round 6: ``tpa_eigh_batch`` as a two-sided block Jacobi iteration on the Hermitian matrix itself (blocks of >= 96 rows; no Gram
    GEMMs, accumulated transform + one Newton-Schulz step) against LAPACK and against the shift + one-sided route it replaces
    (``tpa_eigh_set_direct(0)``): indefinite, flat positive definite (``Xi^dagger Xi`` of a TEBD bond matrix), graded rank-deficient
    PSD (the mixer's density matrix) and +/- pairs, blocks of different sizes sharing the launches (incl. one below 96 rows)."""
    torch, lib, _lib = env
    g = torch.Generator(device="cpu").manual_seed(31)
    dt = torch.complex128 if cplx else torch.float64
    ns = [130, 300, 40, 257, 96]
    mats = []
    for k, n in enumerate(ns):
        x = torch.randn(n, n, dtype=dt, generator=g)
        if k == 0:
            h = x + x.conj().T                                           # indefinite
        elif k == 1:
            h = x.conj().T @ x / n                                       # flat, positive definite
        elif k == 2:
            h = x + x.conj().T
        elif k == 3:
            y = x[:, :n // 2] * torch.logspace(0, -8, n // 2, dtype=torch.float64).to(dt)
            h = y @ y.conj().T                                           # graded, rank n / 2
        else:
            q, _ = torch.linalg.qr(x)
            lam = torch.cat([torch.linspace(1, 2, n // 2, dtype=torch.float64), -torch.linspace(1, 2, n - n // 2, dtype=torch.float64)])
            h = (q * lam.to(dt)) @ q.conj().T                            # +/- lambda pairs
        mats.append(0.5 * (h + h.conj().T))
    jobs, a_off, w_off = [], 0, 0
    for n in ns:
        jobs.append([a_off, n, w_off, a_off, 0, 0, 0, 0])
        a_off += n * n
        w_off += n
    A = torch.cat([x.reshape(-1) for x in mats]).cuda()
    jh = np.array(jobs, np.int64)
    wb = lib.tpa_eigh_worksize(int(cplx), jh.ctypes.data, len(jobs))
    work = torch.empty(wb, dtype=torch.uint8).cuda()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    try:
        for direct in (1, 0):
            lib.tpa_eigh_set_direct(direct)
            W = torch.zeros(w_off, dtype=torch.float64).cuda()
            V = torch.zeros(a_off, dtype=dt).cuda()
            sw = ctypes.c_int()
            _lib.check(lib.tpa_eigh_batch(int(cplx), jh.ctypes.data, len(jobs), A.data_ptr(), W.data_ptr(), V.data_ptr(),
                                          work.data_ptr(), wb, 60, 0.0, ctypes.byref(sw), st), "eigh")
            torch.cuda.synchronize()
            out[direct] = (W.cpu(), V.cpu(), sw.value)
    finally:
        lib.tpa_eigh_set_direct(1)
    for direct in (1, 0):
        W, V, sweeps = out[direct]
        for b, n in enumerate(ns):
            j = jobs[b]
            w = W[j[2]:j[2] + n]
            v = V[j[3]:j[3] + n * n].reshape(n, n)
            ref = torch.linalg.eigvalsh(mats[b])
            nrm = torch.linalg.norm(mats[b]).item()
            assert torch.all(w[1:] >= w[:-1])                                                    # ascending, like LAPACK
            assert (w - ref).abs().max().item() < 2e-14 * n * nrm, (direct, b)
            assert (mats[b] @ v - v * w.to(dt)).abs().max().item() < 1e-13 * n * nrm, (direct, b)
            assert (v.conj().T @ v - torch.eye(n, dtype=dt)).abs().max().item() < (1e-13 if direct else 1e-12), (direct, b)
    # same eigenvalues by both routes
    assert (out[1][0] - out[0][0]).abs().max().item() < 4e-14 * max(ns) * max(torch.linalg.norm(m).item() for m in mats)


@pytest.mark.gpu
@pytest.mark.parametrize("cplx", [False, True])
def test_eigh_from_svd_kernel(env, cplx):
    """``tpa_eigh_from_svd`` against its definition: from LAPACK's SVD ``A = U S VH`` of Hermitian blocks, ``lam_i = d_i S_i`` with
    ``d_i = sign(Re u_i^H v_i)`` and ``err = max_i S_i |v_i - d_i u_i|`` -- at rounding level for an indefinite block with distinct
    |lambda|, O(|A|) for a block with +/- lambda pairs (whose singular subspaces mix), both dtypes, sizes off the 64-column tiles."""
    torch, lib, _lib = env
    rng = np.random.default_rng(3)
    dt = np.complex128 if cplx else np.float64
    ns = [70, 129, 64]
    mats = []
    for k, n in enumerate(ns):
        x = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
        q, _ = np.linalg.qr(x)
        if k == 1:      # +/- pairs
            lam = np.concatenate([np.linspace(1, 2, n // 2), -np.linspace(1, 2, n - n // 2)])
        else:
            lam = np.linspace(-1, 2, n) + 0.01
        h = (q * lam) @ q.conj().T
        mats.append(0.5 * (h + h.conj().T))
    U, S, VH, jobs = [], [], [], []
    uo = so = 0
    for n, h in zip(ns, mats):
        u, sg, vh = np.linalg.svd(h)
        U.append(u.reshape(-1)), S.append(sg), VH.append(vh.reshape(-1))
        jobs.append([uo, n, so, uo, so, 0, 0, 0])
        uo += n * n
        so += n
    Ud = torch.from_numpy(np.concatenate(U).astype(dt)).cuda()
    Sd = torch.from_numpy(np.concatenate(S)).cuda()
    Vd = torch.from_numpy(np.concatenate(VH).astype(dt)).cuda()
    lam_d = torch.zeros(so, dtype=torch.float64).cuda()
    err_d = torch.zeros(len(ns), dtype=torch.float64).cuda()
    jh = np.array(jobs, np.int64)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.tpa_eigh_from_svd(int(cplx), jh.ctypes.data, len(ns), Ud.data_ptr(), Sd.data_ptr(), Vd.data_ptr(), lam_d.data_ptr(),
                                     err_d.data_ptr(), st), "eigh_from_svd")
    torch.cuda.synchronize()
    lam_h, err_h = lam_d.cpu().numpy(), err_d.cpu().numpy()
    o = 0
    for b, (n, h) in enumerate(zip(ns, mats)):
        u, sg, vh = np.linalg.svd(h)
        v = vh.conj().T
        d = np.where(np.real(np.sum(u.conj() * v, axis=0)) < 0, -1., 1.)
        ref_err = np.max(sg * np.linalg.norm(v - u * d[None, :], axis=0))
        np.testing.assert_allclose(lam_h[o:o + n], d * sg, rtol=0, atol=1e-14)
        assert abs(err_h[b] - ref_err) <= 1e-13 + 1e-10 * ref_err
        if b == 1:
            assert err_h[b] > 1e-3                    # +/- pairs: the left singular vectors are NOT eigenvectors
        else:
            assert err_h[b] < 1e-12
            np.testing.assert_allclose(np.sort(lam_h[o:o + n]), np.linalg.eigvalsh(h), rtol=0, atol=1e-13)
        o += n

