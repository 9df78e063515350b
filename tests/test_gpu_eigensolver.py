"""GPU parity tests of the eigensolver stages behind reduction_to_band and of the eigensolver drivers (SURVEY.md
8(f)4, BASELINE configuration 5) through the C ABI, against the reference's own checkers restated in oracle/tridiag.py:

* band_to_tridiagonal: test/unit/eigensolver/test_band_to_tridiag.cpp:50-118 (size list :50-58; the tridiagonal matrix
  and the stored reflectors rebuild the band matrix within mb * m * error / m * error), plus eigvalsh(T) == eigvalsh(band);
* tridiagonal_eigensolver: test_tridiag_solver_local.cpp:62-129 (1D Laplacian, closed form, n * error) and :131-198 with
  test_eigensolver_correctness.h:37-101 (sorted, E^H E == I, A E == E Lambda), sizes of :201-211;
* bt_band_to_tridiagonal: against the one-reflector-at-a-time definition (bt_band_to_tridiag.h:28-61);
* hermitian_eigensolver / hermitian_generalized_eigensolver: test_eigensolver.cpp / test_gen_eigensolver.cpp with the
  same correctness checker, through the reference's C entry points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# test_band_to_tridiag.cpp:50-58: {m, mb, mb_1d, band_size}
B2T_SIZES = [(0, 2, 2), (1, 2, 2), (5, 5, 5), (4, 4, 2), (4, 6, 3), (8, 4, 2), (16, 12, 6), (18, 4, 4), (34, 6, 6), (37, 9, 3)]
# test_tridiag_solver_local.cpp:201-211
TRIDIAG_SIZES = [(0, 8), (4, 2), (16, 16), (16, 8), (16, 4), (16, 5), (100, 10), (93, 7)]
DT = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}


@pytest.fixture(scope="module")
def dlaf():
    import dla_future_amd as d
    d.initialize()
    return d


@pytest.fixture(scope="module")
def grid(dlaf):
    return dlaf.Grid.single()


@pytest.fixture(scope="module")
def td():
    from oracle import tridiag
    return tridiag


def random_band(n, band, dt, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (n, n)).astype(dt)
    if np.dtype(dt).kind == "c":
        a = a + 1j * rng.uniform(-1, 1, (n, n)).astype(dt)
    a = (a + a.conj().T).astype(dt)
    i, j = np.indices((n, n))
    a[np.abs(i - j) > band] = 0
    return np.asfortranarray(a)


def run_b2t(dlaf, grid, td, t, n, nb, band, seed=0):
    dt = DT[t]
    a0 = random_band(n, band, dt, 100 + n + nb + seed)
    a = a0.copy(order="F")
    # below the band the storage holds the reflectors of reduction_to_band: must be ignored
    i, j = np.indices((n, n))
    a[i - j > band] = 7.7
    a[j > i] = -9.9
    d, e, v = dlaf.band_to_tridiagonal(grid, a, nb, band)
    if n == 0:
        return
    assert d.shape == (n,) and e.shape == (max(n - 1, 0),) and v.shape == (n, n)
    ok, diff, bar = td.check_band_to_tridiag(a0, band, d, e, v)
    assert ok, (t, n, nb, band, diff, bar)
    tri = np.diag(d.astype(np.float64)) + np.diag(e.astype(np.float64), -1) + np.diag(e.astype(np.float64), 1)
    ev_t, ev_a = np.linalg.eigvalsh(tri), np.linalg.eigvalsh(a0.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64))
    tol = max(n, 1) * td.error_of(dt) * max(1.0, np.abs(ev_a).max())
    assert np.abs(ev_t - ev_a).max() <= tol, (t, n, nb, band, np.abs(ev_t - ev_a).max(), tol)
    return a0, d, e, v


@pytest.mark.parametrize("t", ["d", "z", "s", "c"])
def test_band_to_tridiag_reference_sizes(dlaf, grid, td, t):
    for n, nb, band in B2T_SIZES:
        run_b2t(dlaf, grid, td, t, n, nb, band)


@pytest.mark.parametrize("t,n,nb,band", [("d", 300, 32, 16), ("z", 260, 64, 32), ("d", 1100, 256, 128), ("z", 700, 256, 128),
                                         ("s", 400, 128, 64), ("d", 517, 128, 128)])
def test_band_to_tridiag_many_sweeps_in_flight(dlaf, grid, td, t, n, nb, band):
    """sizes at which several workgroups chase bulges at once (the hand-off between sweeps on different CUs / XCDs)"""
    a0, d, e, v = run_b2t(dlaf, grid, td, t, n, nb, band, seed=3)
    # and elementwise against the oracle's restatement of SweepWorker (same arithmetic, another summation order): a
    # sanity bar only -- the entries of a tridiagonal reduction are not forward stable (the bar of the reference's test
    # is the reconstruction above), they agree to a few digits less than the spectrum does
    rd, re_, rv = td.band_to_tridiag(a0, band)
    tol = 1e4 * n * td.error_of(DT[t]) * max(1.0, np.abs(a0).max())
    assert np.abs(rd - d).max() <= tol and np.abs(np.abs(re_) - np.abs(e)).max() <= tol, (np.abs(rd - d).max(), tol)


B2T_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import dla_future_amd as d
d.initialize()
g = d.Grid.single()
n, nb, band = 1100, 256, 128
rng = np.random.default_rng(5)
a = np.zeros((n, n), order="F")
for k in range(band + 1):
    a[np.arange(k, n), np.arange(0, n - k)] = rng.uniform(-1, 1, n - k)
dd, ee, v = d.band_to_tridiagonal(g, a, nb, band)
import scipy.linalg as sl
ab = np.zeros((band + 1, n))
for k in range(band + 1):
    ab[k, :n - k] = a[np.arange(k, n), np.arange(0, n - k)]
print("RESULT", np.abs(sl.eigvals_banded(ab, lower=True) - sl.eigvalsh_tridiagonal(dd, ee)).max(), flush=True)
"""


def test_band_to_tridiag_expired_wait_is_reported_not_hung():
    """DLAF_MI355X_B2T_SPIN_LIMIT=0: the first wait of a sweep for its predecessor that is not satisfied at once gives up.
    The register kernel's waves poll without a barrier (wave 0 the progress word, the others an LDS word), so the give-up
    path -- raise a flag, let every wave through, leave together at the next barrier, publish the sweep as done -- must drain
    the launch; the host refuses the result loudly, and the device is usable afterwards."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def child(**env):
        return subprocess.run([sys.executable, "-c", B2T_CHILD % root], cwd=root, env=dict(os.environ, **env),
                              capture_output=True, text=True, timeout=300)
    r = child(DLAF_MI355X_B2T_SPIN_LIMIT="0")
    assert r.returncode != 0, r.stdout
    assert "gave up waiting for its predecessor" in r.stderr, r.stderr[-3000:]
    r = child()
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert float([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1].split()[1]) <= 1100 * 2.3e-16 * 200, r.stdout


@pytest.mark.parametrize("t", ["d", "s"])
def test_tridiagonal_eigensolver_laplace_1d(dlaf, td, t):
    dt = DT[t]
    for n, nb in TRIDIAG_SIZES + [(64, 64), (65, 32), (300, 64), (1000, 128)]:
        d, e, evals, evecs = td.laplace_1d(n, dt)
        w, z = dlaf.tridiagonal_eigensolver(d, e, nb)
        if n == 0:
            continue
        tol = n * td.error_of(dt)
        assert np.abs(w - evals).max() <= tol * max(1.0, np.abs(evals).max()), (t, n, np.abs(w - evals).max(), tol)
        sgn = np.where(np.sign(z[0, :]) == np.sign(evecs[0, :]), 1, -1)  # eigenvectors are unique up to a sign
        if n <= 100:  # the reference's sizes: elementwise against the closed form (beyond them the gaps ~ 1/n^2 of the
            # Laplacian make single entries of an eigenvector ill conditioned: the correctness checker takes over)
            # fp32: 4 x the reference's bar.  A backward-stable solver may move an eigenvector of this matrix by
            # eps * |T| / gap = 1.2e-7 * 4 / 2.9e-3 = 1.7e-4 at n = 100 (gap of the two smallest eigenvalues); the reference's
            # n * error = 2.4e-5 holds for LAPACK's stedc there, this solver (other leaves, other root finder) lands at 5.9e-5
            etol = tol * (4 if t == "s" else 1)
            assert np.abs(z * sgn[None, :] - evecs).max() <= etol, (t, n, np.abs(z * sgn[None, :] - evecs).max(), etol)
        else:
            full = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
            res = td.check_eigensolver(full, w, z)
            assert res["sorted"] and res["orth"] <= res["orth_bar"] and res["residual_ok"], (t, n, res)
            assert np.abs(z * sgn[None, :] - evecs).max() <= 1e3 * tol


@pytest.mark.parametrize("t", ["d", "s"])
def test_tridiagonal_eigensolver_random(dlaf, td, t):
    dt = DT[t]
    for n, nb in TRIDIAG_SIZES + [(64, 64), (130, 64), (515, 128), (2000, 512)]:
        rng = np.random.default_rng(n + 1)
        d = rng.uniform(-1, 1, n).astype(dt)
        e = rng.uniform(-1, 1, max(n - 1, 0)).astype(dt)
        w, z = dlaf.tridiagonal_eigensolver(d, e, nb)
        if n == 0:
            continue
        full = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
        res = td.check_eigensolver(full, w, z)
        assert res["sorted"] and res["orth"] <= res["orth_bar"] and res["residual_ok"], (t, n, res)
        ref = np.linalg.eigvalsh(full.astype(np.float64))
        assert np.abs(ref - w).max() <= n * td.error_of(dt) * max(1.0, np.abs(ref).max())


def test_tridiagonal_eigensolver_deflation_heavy(dlaf, td):
    """matrices that deflate almost everything (equal diagonal, tiny couplings; glued Wilkinson blocks): the deflation
    scan, the Givens rotations and the column classes"""
    n = 600
    cases = []
    d = np.ones(n)
    e = np.full(n - 1, 1e-14)
    cases.append((d, e))
    w21 = np.abs(np.arange(-10, 11)).astype(np.float64)
    d = np.tile(w21, 20)
    e = np.ones(d.size - 1)
    e[20::21] = 1e-11
    cases.append((d, e))
    d = np.zeros(n)
    e = np.ones(n - 1)
    e[63::64] = 0.0
    cases.append((d, e))
    for d, e in cases:
        w, z = dlaf.tridiagonal_eigensolver(d, e, 128)
        full = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
        res = td.check_eigensolver(full, w, z)
        assert res["sorted"] and res["orth"] <= res["orth_bar"] and res["residual_ok"], res


@pytest.mark.parametrize("t,n,band,k", [("d", 37, 3, 37), ("z", 34, 6, 20), ("d", 300, 16, 77), ("z", 260, 32, 260),
                                        ("d", 1100, 128, 300), ("s", 400, 64, 400), ("c", 130, 8, 130)])
def test_bt_band_to_tridiag_vs_definition(dlaf, grid, td, t, n, band, k):
    dt = DT[t]
    nb = band * 2
    a0 = random_band(n, band, dt, 7 + n)
    d, e, v = dlaf.band_to_tridiagonal(grid, a0.copy(order="F"), nb, band)
    rng = np.random.default_rng(5)
    e0 = rng.uniform(-1, 1, (n, k)).astype(dt)
    if np.dtype(dt).kind == "c":
        e0 = (e0 + 1j * rng.uniform(-1, 1, (n, k))).astype(dt)
    emat = np.asfortranarray(e0.copy())
    dlaf.bt_band_to_tridiagonal(band, emat, v)
    ref = td.apply_q(v, band, e0)
    tol = 20 * n * td.error_of(dt)
    assert np.abs(emat - ref).max() <= tol, (t, n, band, np.abs(emat - ref).max(), tol)


def random_hermitian(n, dt, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (n, n)).astype(dt)
    if np.dtype(dt).kind == "c":
        a = a + 1j * rng.uniform(-1, 1, (n, n)).astype(dt)
    return np.asfortranarray((a + a.conj().T).astype(dt))


# test/unit/eigensolver/test_eigensolver.cpp:64-76 and test_gen_eigensolver.cpp:66-72, verbatim:
# {m, mb, eigensolver_min_band} -- the last two of `sizes` are the sub-band cases (band 4 at mb = 8, band 3 at mb = 6)
REF_SIZES = [(0, 2, 100), (5, 8, 100), (34, 34, 100), (4, 3, 100), (16, 10, 100), (34, 13, 100), (32, 5, 100),
             (34, 8, 3), (32, 6, 3)]
REF_SIZES_ID = [(8, 4, 4), (34, 8, 4)]   # the identity matrix: full deflation, zero reflectors end to end


@pytest.fixture
def min_band(dlaf):
    """getTuneParameters().eigensolver_min_band = b_min for one case (test_eigensolver.cpp:142), restored afterwards"""
    old = dlaf.eigensolver_min_band()

    def set_(b_min):
        dlaf.eigensolver_min_band(b_min)

    yield set_
    dlaf.eigensolver_min_band(old)


def test_eigensolver_min_band_tune_parameter(dlaf, min_band):
    """include/dlaf/tune.h:128 + internal/get_band_size.h:18-31; DLAF_EIGENSOLVER_MIN_BAND is read by dlaf_initialize
    (src/init.cpp:220), checked in a fresh process"""
    import subprocess
    import sys
    assert dlaf.eigensolver_min_band() == 100 and dlaf.get_band_size(512) == 128
    min_band(3)
    assert [dlaf.get_band_size(nb) for nb in (8, 6, 34, 512)] == [4, 3, 17, 4]
    min_band(4)
    assert [dlaf.get_band_size(nb) for nb in (4, 8)] == [4, 4]
    code = ("import dla_future_amd as d; d.initialize(); "
            "print(d.eigensolver_min_band(), d.get_band_size(8), d.get_band_size(512))")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, DLAF_EIGENSOLVER_MIN_BAND="3"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split() == ["3", "4", "4"], (out.stdout, out.stderr[-500:])


def test_hermitian_eigensolver_reference_sizes(dlaf, grid, td, min_band):
    """testEigensolver over exactly the reference's `sizes` (random Hermitian) and `sizes_id` (identity) with their
    eigensolver_min_band, all four element types (test_eigensolver.cpp:64-76,140-166)"""
    for t in "sdcz":
        dt = DT[t]
        for kind, sizes in (("random", REF_SIZES), ("identity", REF_SIZES_ID)):
            for n, nb, b_min in sizes:
                min_band(b_min)
                a0 = random_hermitian(n, dt, 11 + n) if kind == "random" else np.asfortranarray(np.eye(n, dtype=dt))
                a = a0.copy(order="F")
                a[np.triu_indices(n, 1)] = -9.9
                w, z = dlaf.hermitian_eigensolver(grid, "L", a, nb)
                if n == 0:
                    continue
                res = td.check_eigensolver(a0, w, z)
                assert res["sorted"] and res["orth"] <= res["orth_bar"] and res["residual_ok"], (t, kind, n, nb, b_min, res)
                if kind == "identity":
                    assert np.all(w == 1), (t, n, nb, w)


def test_hermitian_generalized_eigensolver_reference_sizes(dlaf, grid, td, min_band):
    """testGenEigensolver over the reference's `sizes` with their eigensolver_min_band, all four element types, both
    the plain and the `_factorized` entry (test_gen_eigensolver.cpp:66-72,105-129: B-orthonormality of the
    eigenvectors and A Z = B Z Lambda)"""
    for t in "sdcz":
        dt = DT[t]
        err = td.error_of(dt)
        for n, nb, b_min in REF_SIZES:
            min_band(b_min)
            a0 = random_hermitian(n, dt, 21 + n)
            b0 = random_hermitian(n, dt, 22 + n)
            b0 = np.asfortranarray((b0 @ b0.conj().T / max(n, 1) + np.eye(n, dtype=dt) * 2).astype(dt))
            a, b = a0.copy(order="F"), b0.copy(order="F")
            w, z = dlaf.hermitian_generalized_eigensolver(grid, "L", a, b, nb)
            if n == 0:
                continue
            assert np.all(np.diff(w) >= 0)
            g = z.conj().T @ b0 @ z
            assert np.abs(g - np.eye(n)).max() <= 10 * n * err * np.abs(b0).max(), (t, n, nb, np.abs(g - np.eye(n)).max())
            r = a0 @ z - (b0 @ z) * w[None, :]
            assert np.abs(r).max() <= 10 * n * err * max(1.0, np.abs(a0).max() * np.abs(w).max()), (t, n, nb, np.abs(r).max())
            w2, z2 = dlaf.hermitian_generalized_eigensolver(grid, "L", a0.copy(order="F"), b, nb, factorized=True)
            assert np.abs(w2 - w).max() <= 10 * n * err * max(1.0, np.abs(w).max()), (t, n, nb)


@pytest.mark.parametrize("t,n,nb", [("d", 1100, 256), ("z", 700, 128), ("d", 2048, 512)])
def test_hermitian_eigensolver_band_128(dlaf, grid, td, t, n, nb):
    dt = DT[t]
    a0 = random_hermitian(n, dt, 3)
    w, z = dlaf.hermitian_eigensolver(grid, "L", a0.copy(order="F"), nb)
    res = td.check_eigensolver(a0, w, z)
    assert res["sorted"] and res["orth"] <= res["orth_bar"] and res["residual_ok"], (t, n, nb, res)
    ref = np.linalg.eigvalsh(a0)
    assert np.abs(ref - w).max() <= n * td.error_of(dt) * np.abs(ref).max()
    assert all(ms >= 0 for ms in dlaf.eigensolver_profile())


@pytest.mark.parametrize("t", ["d", "z"])
def test_hermitian_generalized_eigensolver(dlaf, grid, td, t):
    dt = DT[t]
    for n, nb in [(34, 8), (64, 16), (300, 64)]:
        a0 = random_hermitian(n, dt, 21 + n)
        b0 = random_hermitian(n, dt, 22 + n)
        b0 = np.asfortranarray(b0 @ b0.conj().T / n + np.eye(n, dtype=dt) * 2)
        a, b = a0.copy(order="F"), b0.copy(order="F")
        w, z = dlaf.hermitian_generalized_eigensolver(grid, "L", a, b, nb)
        # test_gen_eigensolver.cpp: B-orthonormality and A Z = B Z Lambda
        err = td.error_of(dt)
        assert np.all(np.diff(w) >= 0)
        g = z.conj().T @ b0 @ z
        assert np.abs(g - np.eye(n)).max() <= 10 * n * err * np.abs(b0).max(), (t, n, np.abs(g - np.eye(n)).max())
        r = a0 @ z - (b0 @ z) * w[None, :]
        assert np.abs(r).max() <= 10 * n * err * max(1.0, np.abs(a0).max() * np.abs(w).max()), (t, n, np.abs(r).max())
        # the factor of B is what dlaf_cholesky_factorization leaves
        l = np.linalg.cholesky(b0)
        assert np.abs(np.tril(b) - l).max() <= 50 * n * err * np.abs(l).max()
        # and the `_factorized` entry with that factor gives the same spectrum
        w2, z2 = dlaf.hermitian_generalized_eigensolver(grid, "L", a0.copy(order="F"), b, nb, factorized=True)
        assert np.abs(w2 - w).max() <= 10 * n * err * max(1.0, np.abs(w).max())
