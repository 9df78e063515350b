"""GPU parity tests of the four tile kernels through the C ABI against the oracle and the
reference's closed-form answers (test/unit/test_lapack_tile/test_potrf.h:33-77,
test/unit/test_blas_tile/test_{trsm,herk,gemm}.h), with the reference's tolerances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPES = ["d", "z", "s", "c"]


@pytest.fixture(scope="module")
def dlaf():
    import dla_future_amd as d
    d.initialize()
    return d


def err_of(orc, t):
    return (8 if t in "cz" else 2) * orc.eps_of(orc.DTYPES[t])


def rnd(rng, shape, dt):
    a = rng.uniform(-1, 1, shape)
    if np.issubdtype(dt, np.complexfloating):
        a = a + 1j * rng.uniform(-1, 1, shape)
    return np.asfortranarray(a.astype(dt))


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_potrf_analytic(dlaf, oracle, t, uplo):
    # test_potrf.h:33-57; sizes test_lapack_tile.cpp:142-145 plus multi-block ones (inner block 64)
    for n, extra in [(0, 0), (1, 0), (11, 2), (12, 0), (17, 3), (64, 0), (65, 1), (128, 0), (200, 1), (333, 0)]:
        a, l = oracle.cholesky_setters(uplo, n, oracle.DTYPES[t])
        store = np.full((max(1, n) + extra, max(1, n)), 3.3, dtype=a.dtype, order="F")
        store[:n, :n] = a
        assert dlaf.tile_potrf(uplo, store[:n, :n]) == 0
        tol = 4 * (n + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(l, store[:n, :n], tol, tol)
        assert ok, (n, md)
        assert (store[n:, :] == 3.3).all()


@pytest.mark.parametrize("t", TYPES)
def test_potrf_random_vs_oracle(dlaf, oracle, t):
    for n in (100, 257):
        a0 = oracle.set_random_hpd(n, 64, oracle.DTYPES[t])
        for uplo in "LU":
            ref = a0.copy(order="F")
            assert oracle.potrf(uplo, ref) == 0
            got = a0.copy(order="F")
            assert dlaf.tile_potrf(uplo, got) == 0
            tol = 4 * (n + 1) * err_of(oracle, t)
            ok, md = oracle.check_near(ref, got, tol, tol)
            assert ok, (n, uplo, md)


@pytest.mark.parametrize("t", TYPES)
def test_potrf_non_spd_info(dlaf, oracle, t):
    # test_potrf.h:59-77: null matrix -> info == 1; plus a failure in a later block
    for uplo in "LU":
        a = np.zeros((5, 5), dtype=oracle.DTYPES[t], order="F")
        assert dlaf.tile_potrf(uplo, a) == 1
        b = oracle.set_random_hpd(150, 64, oracle.DTYPES[t])
        b[100, 100] = -5
        assert dlaf.tile_potrf(uplo, b) == 101


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_trsm_vs_oracle(dlaf, oracle, t, uplo):
    rng = np.random.default_rng(3)
    dt = oracle.DTYPES[t]
    for m, n in [(0, 5), (3, 0), (3, 5), (17, 13), (13, 17), (64, 64), (130, 70), (70, 130), (300, 200)]:
        na = n if uplo == "L" else m
        tri = rnd(rng, (na, na), dt) * dt(0.1)
        tri[np.arange(na), np.arange(na)] = (np.abs(tri.diagonal()) + 1.5).astype(dt)  # real positive diagonal
        a = np.asfortranarray(np.tril(tri) if uplo == "L" else np.triu(tri))
        junk = dt(-9.9)
        a = np.asfortranarray(a + (np.triu(np.full((na, na), junk), 1) if uplo == "L" else np.tril(np.full((na, na), junk), -1)))
        b0 = rnd(rng, (m, n), dt)
        ref = b0.copy(order="F")
        if m and n:
            if uplo == "L":
                oracle.trsm("R", "L", "C", "N", 1.0, a, ref)
            else:
                oracle.trsm("L", "U", "C", "N", 1.0, a, ref)
        got = b0.copy(order="F")
        dlaf.tile_trsm(uplo, a, got)
        tol = 10 * (max(m, n) + 1) * err_of(oracle, t)  # test_trsm.h:61
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (m, n, md)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_herk_vs_oracle(dlaf, oracle, t, uplo):
    rng = np.random.default_rng(5)
    dt = oracle.DTYPES[t]
    for n, k in [(0, 2), (5, 0), (5, 3), (9, 16), (13, 13), (128, 64), (200, 130), (257, 19)]:
        a = rnd(rng, (n, k) if uplo == "L" else (k, n), dt)
        c0 = rnd(rng, (n, n), dt)
        ref = c0.copy(order="F")
        if n and k:
            oracle.herk(uplo, "N" if uplo == "L" else "C", -1.0, a, 1.0, ref, k=k)
        # K = 0 is never issued by the factorization (kb >= 1); the library treats it as a no-op
        got = c0.copy(order="F")
        dlaf.tile_herk(uplo, a, got)
        tol = (k + 1) * err_of(oracle, t)  # test_herk.h:88
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (n, k, md)
        # other triangle untouched, bit for bit
        other = np.triu(got, 1) if uplo == "L" else np.tril(got, -1)
        other0 = np.triu(c0, 1) if uplo == "L" else np.tril(c0, -1)
        assert np.array_equal(other, other0)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_gemm_vs_oracle(dlaf, oracle, t, uplo):
    rng = np.random.default_rng(7)
    dt = oracle.DTYPES[t]
    for m, n, k in [(3, 5, 7), (13, 32, 8), (12, 12, 5), (128, 128, 128), (130, 70, 33), (70, 130, 64), (256, 200, 300)]:
        if uplo == "L":
            a, b = rnd(rng, (m, k), dt), rnd(rng, (n, k), dt)
        else:
            a, b = rnd(rng, (k, m), dt), rnd(rng, (k, n), dt)
        c0 = rnd(rng, (m, n), dt)
        ref = c0.copy(order="F")
        if uplo == "L":
            oracle.gemm("N", "C", -1.0, a, b, 1.0, ref)
        else:
            oracle.gemm("C", "N", -1.0, a, b, 1.0, ref)
        got = c0.copy(order="F")
        dlaf.tile_gemm(uplo, a, b, got)
        tol = 2 * (k + 1) * err_of(oracle, t)  # test_gemm.h:68
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (m, n, k, md)


def test_gemm_asymmetric_layout_probe(dlaf, oracle):
    """A = I with an asymmetric B catches a swapped/shifted MFMA accumulator map outright."""
    n = 64
    a = np.asfortranarray(np.eye(n))
    b = np.asfortranarray(np.arange(n * n, dtype=np.float64).reshape(n, n) / 7.0)
    c = np.zeros((n, n), order="F")
    dlaf.tile_gemm("L", a, b, c)  # C -= I * B^T
    assert np.array_equal(c, -b.T)
