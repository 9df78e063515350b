"""GPU parity tests of the factorization through the reference's entry points
(test/unit/factorization/test_cholesky.cpp:54-120, test/unit/c_api/factorization/
test_cholesky_c_api.cpp:62-155), against the analytic answer, the oracle and the miniapp's
residual bar (miniapp/miniapp_cholesky.cpp:432-442)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPES = ["d", "z", "s", "c"]
CHOLESKY_SIZES = [(0, 2), (5, 8), (34, 34), (4, 3), (16, 10), (34, 13), (32, 5)]


@pytest.fixture(scope="module")
def dlaf():
    import dla_future_amd as d
    d.initialize()
    return d


@pytest.fixture(scope="module")
def grid(dlaf):
    return dlaf.Grid.single()


def err_of(orc, t):
    return (8 if t in "cz" else 2) * orc.eps_of(orc.DTYPES[t])


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_cholesky_local_analytic(dlaf, grid, oracle, t, uplo):
    for m, mb in CHOLESKY_SIZES + [(150, 64), (200, 70), (333, 128)]:
        a, l = oracle.cholesky_setters(uplo, m, oracle.DTYPES[t])
        store = np.full((max(1, m) + 3, max(1, m)), 4.4, dtype=a.dtype, order="F")
        store[:m, :m] = a
        assert dlaf.cholesky_factorization(grid, uplo, store[:m, :m], mb) == 0
        tol = 4 * (m + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(l, store[:m, :m], tol, tol)  # includes the untouched -9.9 triangle
        assert ok, (m, mb, md)
        assert (store[m:, :] == 4.4).all()


@pytest.fixture(params=["classic", "early", "sidecar", "pairs"])
def schedule(request, monkeypatch):
    """The issue orders of the tile DAG (runtime.cpp: classic / pairs = one-process defaults for large /
    small blocks, sidecar = the round-1 small-block order, early diagonal = process-grid default);
    DLAF_MI355X_SCHEDULE is read at every factorization."""
    monkeypatch.setenv("DLAF_MI355X_SCHEDULE", request.param)
    return request.param


def _random_vs_oracle(dlaf, grid, oracle, t, uplo, sizes):
    for n, nb in sizes:
        dt = oracle.DTYPES[t]
        a0 = oracle.set_random_hpd(n, nb, dt)
        ref = a0.copy(order="F")
        assert oracle.cholesky_local(uplo, ref, nb) == 0
        got = a0.copy(order="F")
        assert dlaf.cholesky_factorization(grid, uplo, got, nb) == 0
        tol = 4 * (n + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(oracle.tri(uplo, ref), oracle.tri(uplo, got), tol, tol)
        assert ok, (n, nb, md)
        other = np.triu(got, 1) if uplo == "L" else np.tril(got, -1)
        other0 = np.triu(a0, 1) if uplo == "L" else np.tril(a0, -1)
        assert np.array_equal(other, other0)
        assert oracle.cholesky_residual(uplo, a0, got) <= n * oracle.eps_of(dt)


# Covering set (round 3: the full product schedule x type x uplo x 7 sizes took a third of the GPU suite): every
# issue order x {d, z} x {L, U} on three sizes (ragged small block, 16-aligned fast path, one-tile edge), and
# s / c -- which share every code path but the element type -- on the default order over the full size list.
@pytest.mark.parametrize("t", ["d", "z"])
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_cholesky_random_vs_oracle(dlaf, grid, oracle, t, uplo, schedule):
    _random_vs_oracle(dlaf, grid, oracle, t, uplo, [(515, 128), (1024, 256), (129, 64)])


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_cholesky_random_vs_oracle_default_schedule(dlaf, grid, oracle, t, uplo):
    _random_vs_oracle(dlaf, grid, oracle, t, uplo, [(300, 64), (515, 128), (64, 64), (100, 64), (200, 32)])


def test_generator_matches_oracle(dlaf, grid, oracle):
    for t in TYPES:
        dt = oracle.DTYPES[t]
        for n, nb in [(34, 13), (100, 32)]:
            a = np.zeros((n, n), dtype=dt, order="F")
            dlaf.set_random_hermitian_positive_definite(grid, a, n, nb)
            o = oracle.set_random_hpd(n, nb, dt)
            if t in "sd":
                assert np.array_equal(a, o), (t, n, nb)
            else:
                # polar() goes through libm's sin/cos: the product (clang) and the oracle (gcc) may pick
                # sincos vs sin+cos, which differ in the last bit on some hosts
                assert np.abs(a - o).max() <= 4 * oracle.eps_of(dt), (t, n, nb)
                assert (a.real == o.real).mean() > 0.99


def test_pdpotrf_scalapack_entry(dlaf, grid, oracle):
    # test_cholesky_c_api.cpp:108-155: same matrices through dlaf_p?potrf
    for t in TYPES:
        n, nb = 130, 32
        a0 = oracle.set_random_hpd(n, nb, oracle.DTYPES[t])
        a = a0.copy(order="F")
        desca = [1, grid.context, n, n, nb, nb, 0, 0, n]
        assert dlaf.pxpotrf("L", n, a, 1, 1, desca) == 0
        ref = a0.copy(order="F")
        oracle.cholesky_local("L", ref, nb)
        tol = 4 * (n + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(np.tril(ref), np.tril(a), tol, tol)
        assert ok, (t, md)


def test_non_spd_reports_lapack_info(dlaf, grid, oracle):
    n, nb = 400, 128
    a = oracle.set_random_hpd(n, nb, np.float64)
    a[300, 300] = -1.0
    assert dlaf.cholesky_factorization(grid, "L", a.copy(order="F"), nb) == 301
    assert dlaf.cholesky_factorization(grid, "U", a.copy(order="F"), nb) == 301


def test_device_resident_repeatable(dlaf, grid, oracle):
    """miniapp pattern: upload once, copy the pristine matrix, factorize, repeat -> identical bits."""
    n, nb = 1024, 256
    a0 = oracle.set_random_hpd(n, nb, np.float64)
    ref = dlaf.DeviceMatrix(grid, np.float64, "L", n, nb)
    work = dlaf.DeviceMatrix(grid, np.float64, "L", n, nb)
    ref.upload(a0)
    outs = []
    for _ in range(2):
        work.copy_from(ref)
        assert work.factorize() == 0
        out = a0.copy(order="F")
        work.download(out)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])
    assert oracle.cholesky_residual("L", a0, outs[0]) <= n * np.finfo(np.float64).eps


@pytest.mark.parametrize("t", ["d", "z"])
def test_larger_property_checks(dlaf, grid, oracle, t):
    """Size-independent properties at a size the oracle cannot factor in seconds: residual bar of the
    miniapp, positivity of the diagonal, untouched opposite triangle."""
    n, nb = (4096, 512) if t == "d" else (2048, 256)
    dt = oracle.DTYPES[t]
    a0 = np.zeros((n, n), dtype=dt, order="F")
    dlaf.set_random_hermitian_positive_definite(grid, a0, n, nb)
    a = a0.copy(order="F")
    assert dlaf.cholesky_factorization(grid, "L", a, nb) == 0
    assert (np.diag(a).real > 0).all() and (np.diag(a).imag == 0).all()
    assert np.array_equal(np.triu(a, 1), np.triu(a0, 1))
    assert oracle.cholesky_residual("L", a0, a) <= n * oracle.eps_of(dt)


@pytest.mark.parametrize("t", ["d", "z", "s"])
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_device_residual_checker_matches_host(dlaf, grid, oracle, t, uplo):
    """dlaf_mi355x_cholesky_residual (the miniapp's check_cholesky on the device) against the same
    quantity computed on the host from the downloaded factor."""
    n, nb = 700, 128
    dt = oracle.DTYPES[t]
    a0 = oracle.set_random_hpd(n, nb, dt)
    orig = dlaf.DeviceMatrix(grid, dt, uplo, n, nb)
    fact = dlaf.DeviceMatrix(grid, dt, uplo, n, nb)
    orig.upload(a0)
    fact.copy_from(orig)
    assert fact.factorize() == 0
    out = a0.copy(order="F")
    fact.download(out)
    diff, norm_a = orig.residual_against(fact)
    tri0 = oracle.tri(uplo, a0)
    assert norm_a == pytest.approx(np.abs(tri0).max(), rel=1e-6)
    host = oracle.cholesky_residual(uplo, a0, out)
    eps = oracle.eps_of(dt)
    assert diff / norm_a <= n * eps                      # miniapp bar (miniapp_cholesky.cpp:432-442)
    assert abs(diff / norm_a - host) <= 8 * eps          # same quantity up to summation order
    # a wrong factor must be flagged
    bad = out.copy(order="F")
    idx = (n // 2, n // 3) if uplo == "L" else (n // 3, n // 2)
    bad[idx] += dt(0.5)
    fact.upload(bad)
    orig.upload(a0)
    diff2, _ = orig.residual_against(fact)
    assert diff2 / norm_a > 100 * n * eps


def test_full_size_residual_on_device(dlaf, grid, oracle):
    """Property check at a size the host checker cannot reach: N = 16384, nb = 1024 (the headline tile
    size), residual computed entirely on the GPU."""
    n, nb = 16384, 1024
    a = np.zeros((n, n), dtype=np.float64, order="F")
    dlaf.set_random_hermitian_positive_definite(grid, a, n, nb)
    orig = dlaf.DeviceMatrix(grid, np.float64, "L", n, nb)
    fact = dlaf.DeviceMatrix(grid, np.float64, "L", n, nb)
    orig.upload(a)
    del a
    fact.copy_from(orig)
    assert fact.factorize() == 0
    diff, norm_a = orig.residual_against(fact)
    assert norm_a > 2 * n - 1.001 and norm_a < 2 * n + 1.001
    assert diff / norm_a <= n * np.finfo(np.float64).eps
