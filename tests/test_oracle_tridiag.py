"""CPU tests pinning oracle/tridiag.py (the checker of the band_to_tridiagonal / bt_band_to_tridiagonal / tridiagonal
eigensolver GPU tests): its restatement of BandToTridiag::call_L against the reference test's own reconstruction
property (test_band_to_tridiag.cpp:60-118, its size list :50-58), spectrum preservation vs LAPACK, the back-transformation
against eigh of the band matrix, xLARFG against LAPACK's, and the closed-form 1D Laplacian of
test_tridiag_solver_local.cpp:62-129 against scipy."""
import numpy as np
import pytest
import scipy.linalg as sl

from oracle import tridiag as td

SIZES = [(0, 2), (1, 2), (5, 5), (4, 2), (4, 3), (8, 2), (16, 6), (18, 4), (34, 6), (37, 3), (70, 8)]


def band_matrix(n, b, dt, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (n, n)).astype(dt)
    if np.dtype(dt).kind == "c":
        a = a + 1j * rng.uniform(-1, 1, (n, n))
    a = (a + a.conj().T).astype(dt)
    i, j = np.indices((n, n))
    a[np.abs(i - j) > b] = 0
    return a


@pytest.mark.parametrize("dt", [np.float64, np.complex128, np.float32, np.complex64])
def test_band_to_tridiag_restatement(dt):
    for n, b in SIZES:
        a = band_matrix(n, b, dt, n + b)
        d, e, v = td.band_to_tridiag(a, b)
        if n == 0:
            continue
        ok, diff, bar = td.check_band_to_tridiag(a, b, d, e, v)
        assert ok, (n, b, diff, bar)
        t = np.diag(d.astype(np.float64)) + np.diag(e.astype(np.float64), -1) + np.diag(e.astype(np.float64), 1)
        wide = np.complex128 if np.dtype(dt).kind == "c" else np.float64
        assert np.abs(np.linalg.eigvalsh(t) - np.linalg.eigvalsh(a.astype(wide))).max() <= 4 * max(n, 1) * td.error_of(dt) * max(1, np.abs(a).max() * n)
        # number of stored reflectors and their layout (band_to_tridiag.h:49-63)
        assert len(td.reflector_list(n, b, dt)) == sum(td.nr_steps_for_sweep(s, n, b) for s in range(max(0, td.nr_sweeps(n, dt))))
        # back-transformation: eigenvectors of T -> eigenvectors of the band matrix
        w, z = np.linalg.eigh(t)
        ev = td.apply_q(v.astype(wide), b, z.astype(wide))
        assert np.abs(a.astype(wide) @ ev - ev * w).max() <= 50 * max(n, 1) * td.error_of(dt) * max(1, np.abs(a).max())


def test_larfg_matches_lapack():
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 17):
        x = rng.uniform(-1, 1, n)
        y = x.copy()
        tau = td.larfg(y)
        alpha, xx, ltau = sl.lapack.dlarfg(n, x[0], x[1:].copy())
        assert np.allclose([y[0], tau], [alpha, ltau]) and np.allclose(y[1:], xx)
        xc = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
        yc = xc.copy()
        tauc = td.larfg(yc)
        alpha, xx, ltau = sl.lapack.zlarfg(n, xc[0], xc[1:].copy())
        assert np.allclose([yc[0], tauc], [alpha, ltau]) and np.allclose(yc[1:], xx)


def test_laplace_closed_form_and_checker():
    for n in (1, 4, 16, 93):
        d, e, evals, evecs = td.laplace_1d(n)
        w, z = sl.eigh_tridiagonal(d, e) if n > 1 else (d.copy(), np.ones((1, 1)))
        assert np.abs(w - evals).max() <= n * td.error_of(np.float64) * 4
        full = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
        res = td.check_eigensolver(full, evals, evecs)
        assert res["sorted"] and res["orth"] <= res["orth_bar"] and res["residual_ok"], res
        bad = evecs.copy()
        bad[:, 0] *= 1.001
        assert not (td.check_eigensolver(full, evals, bad)["orth"] <= 10 * n * td.error_of(np.float64)) or n == 1
