"""GPU parity tests of generalized_to_standard through the C ABI: the reference's own test
(test/unit/eigensolver/test_gen_to_std.cpp:54-83: analytic operands, abs tolerance 10 (m+1) error, factor
untouched) plus random operands against the oracle restatement of GenToStd::call_L."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPES = ["d", "z", "s", "c"]
SIZES = [(0, 2), (5, 8), (34, 34), (4, 3), (16, 10), (34, 13), (32, 5)]  # test_gen_to_std.cpp:54-58


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
def test_gen_to_std_local_analytic(dlaf, grid, oracle, t, uplo):
    dt = oracle.DTYPES[t]
    for m, mb in SIZES + [(150, 64), (200, 70), (300, 128)]:
        tmat, a, b = oracle.gen_to_std_setters(uplo, m, dt)
        got = a.copy(order="F")
        fac = tmat.copy(order="F")
        assert dlaf.generalized_to_standard(grid, uplo, got, fac, mb) == 0
        ok, md = oracle.check_near(b, got, 0, 10 * (m + 1) * err_of(oracle, t))  # includes the -9.9 triangle
        assert ok, (m, mb, md)
        assert np.array_equal(fac, tmat)   # CHECK_MATRIX_NEAR(el_t, mat_th, 0, error) of the distributed test


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_gen_to_std_random_vs_oracle(dlaf, grid, oracle, t, uplo):
    dt = oracle.DTYPES[t]
    # (the numpy restatement of GenToStd::call_L is what takes the time: 1536 instead of 2048 for the nb = 512 case)
    for n, nb in [(300, 64), (515, 128), (1024, 256), (129, 64), (1536 if t == "d" else 512, 512 if t == "d" else 256)]:
        b0 = oracle.set_random_hpd(n, nb, dt)
        a0 = (oracle.set_random_hpd(n, nb, dt) * dt(1.0 / n)).astype(dt)
        fac = b0.copy(order="F")
        assert oracle.cholesky_local(uplo, fac, nb) == 0
        ref = a0.copy(order="F")
        oracle.gen_to_std_local(uplo, ref, fac, nb)
        got = a0.copy(order="F")
        assert dlaf.generalized_to_standard(grid, uplo, got, fac, nb) == 0
        scale = np.abs(oracle.tri(uplo, ref)).max()
        tol = 10 * (n + 1) * err_of(oracle, t) * max(1.0, scale)
        ok, md = oracle.check_near(oracle.tri(uplo, ref), oracle.tri(uplo, got), 0, tol)
        assert ok, (n, nb, md, tol)
        other = np.triu(got, 1) if uplo == "L" else np.tril(got, -1)
        assert np.array_equal(other, np.triu(a0, 1) if uplo == "L" else np.tril(a0, -1))


def test_pdhegst_scalapack_entry_and_device_handles(dlaf, grid, oracle):
    n, nb = 260, 64
    for t in ("d", "z"):
        dt = oracle.DTYPES[t]
        b0 = oracle.set_random_hpd(n, nb, dt)
        a0 = (oracle.set_random_hpd(n, nb, dt) * dt(1.0 / n)).astype(dt)
        fac = b0.copy(order="F")
        assert dlaf.pxpotrf("L", n, fac, 1, 1, [1, grid.context, n, n, nb, nb, 0, 0, n]) == 0
        ref = a0.copy(order="F")
        oracle.gen_to_std_local("L", ref, fac, nb)
        got = a0.copy(order="F")
        desc = [1, grid.context, n, n, nb, nb, 0, 0, n]
        scale, info = dlaf.pxhegst(1, "L", n, got, 1, 1, desc, fac, 1, 1, desc)
        assert (scale, info) == (1.0, 0)
        tol = 10 * (n + 1) * err_of(oracle, t) * max(1.0, np.abs(np.tril(ref)).max())
        assert oracle.check_near(np.tril(ref), np.tril(got), 0, tol)[0]
        # p?potrf -> gen_to_std chained on device-resident matrices
        am = dlaf.DeviceMatrix(grid, dt, "L", n, nb)
        bm = dlaf.DeviceMatrix(grid, dt, "L", n, nb)
        am.upload(a0)
        bm.upload(b0)
        assert bm.factorize() == 0
        assert am.generalized_to_standard(bm) == 0
        out = a0.copy(order="F")
        am.download(out)
        assert oracle.check_near(np.tril(ref), np.tril(out), 0, tol)[0]
        am.close()
        bm.close()
