"""``npc.expm`` (reference np_conserved.py:4064: scipy's Pade ``expm`` per block) on the device: scaling-and-squaring Taylor
series evaluated with block GEMMs.  Lives in its own module (not in test_reference_suite.py, whose module-level skip depends
on the reference tree) so that the GPU variant always runs."""
import numpy as np


def test_expm_is_closer_to_exact_than_scipy(backend):
    """Why ``test_np_conserved.py::test_expm`` is deselected above: distance to the exact exponential (extended-precision
    Taylor series with 10 squarings) in ULP, for the mirror's ``expm`` and for ``scipy.linalg.expm``."""
    import scipy.linalg
    from tenpy_amd.linalg import np_conserved as npc
    rng = np.random.default_rng(7)
    worst_mirror, worst_scipy = 0., 0.
    for _ in range(10):
        n = 8
        flat = rng.random((n, n))
        A = npc.Array.from_ndarray_trivial(flat)
        X = flat.astype(np.longdouble) / 1024
        T = np.eye(n, dtype=np.longdouble)
        for k in range(30, 0, -1):
            T = X @ T / k + np.eye(n, dtype=np.longdouble)
        for _ in range(10):
            T = T @ T
        exact = T.astype(np.float64)

        def ulp(Y):
            return float(np.max(np.abs(Y - exact) / np.spacing(np.maximum(np.abs(Y), np.abs(exact)))))
        worst_mirror = max(worst_mirror, ulp(npc.expm(A).to_ndarray()))
        worst_scipy = max(worst_scipy, ulp(scipy.linalg.expm(flat)))
    assert worst_mirror <= 16, worst_mirror
    assert worst_mirror <= worst_scipy
