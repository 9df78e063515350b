"""Triangular solver on the GPU against the reference's analytic systems
(test/unit/solver/test_triangular.cpp:59-102 with getTriangularSystem, util_generic_blas.h:258-373) and
against the oracle's generic trsm on random operands; tolerances are the reference's."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPES = ["s", "d", "c", "z"]
# (m, n, nb): the reference's sizes with mb == nb, plus sizes whose tiles span several 64-wide blocks of the TRSM
# kernel and several 128-row strips; RECT_SIZES (m, n, mb, nb): the reference's sizes with mb != nb
# (test_triangular.cpp:59-65) plus larger ones -- A's block is mb for side = Left, nb for side = Right
SIZES = [(0, 0, 1), (0, 2, 2), (7, 0, 2), (2, 2, 5), (10, 10, 3), (7, 7, 2), (3, 2, 7), (12, 3, 5), (15, 7, 3),
         (2, 3, 7), (4, 13, 5), (19, 25, 6), (150, 70, 32), (130, 257, 64), (200, 300, 128), (333, 129, 100)]
RECT_SIZES = [(10, 10, 2, 3), (7, 7, 3, 2), (7, 6, 3, 2), (15, 7, 3, 5), (7, 8, 2, 9), (19, 25, 6, 5), (0, 2, 1, 2),
              (7, 0, 2, 1), (150, 70, 32, 48), (130, 257, 64, 40), (200, 131, 24, 128)]


@pytest.fixture(scope="module")
def dlaf():
    import dla_future_amd as d
    d.initialize()
    return d


@pytest.fixture(scope="module")
def grid(dlaf):
    return dlaf.Grid.single()


def err_of(orc, t):
    return (8 if t in "cz" else 2) * orc.eps_of(orc.DTYPES[t])  # TypeUtilities<T>::error (util_types.h:40)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("side", ["L", "R"])
def test_triangular_solver_analytic(dlaf, grid, oracle, t, side):
    dt = oracle.DTYPES[t]
    alpha = dt(complex(-1.2, .7)) if t in "cz" else dt(-1.2)
    # the analytic operands grow like (i+1)/(k+.5) (complex: with rotating phases): beyond the reference's own
    # sizes (<= 25) the systems leave its tolerance through conditioning alone -- the CPU oracle's substitution
    # does too for c/z -- so the multi-tile sizes run the analytic check in real double precision only and
    # every type is checked against the oracle on well-conditioned operands below
    sizes = SIZES if t == "d" else [sz for sz in SIZES if max(sz[:2]) <= 25]
    for (m, n, nb), uplo, op, diag in itertools.product(sizes, "LU", "NTC", "NU"):
        a, b, x = oracle.triangular_system(side, uplo, op, diag, alpha, m, n, dt)
        # padded leading dimensions, as a caller's ScaLAPACK-style local arrays have
        sa = np.full((a.shape[0] + 3, max(1, a.shape[1])), 5.5, dtype=dt, order="F")
        sb = np.full((m + 2, max(1, n)), 6.5, dtype=dt, order="F")
        sa[:a.shape[0], :a.shape[1]] = a
        sb[:m, :n] = b
        dlaf.triangular_solver(grid, side, uplo, op, diag, alpha, sa[:a.shape[0], :a.shape[1]], sb[:m, :n], nb)
        tol = 40 * (m + 1) * err_of(oracle, t)   # test_triangular.cpp:101-102
        ok, md = oracle.check_near(x, sb[:m, :n], tol, tol)
        assert ok, (md, tol, m, n, nb, side, uplo, op, diag)
        assert (sb[m:, :] == 6.5).all() and np.array_equal(sa[:a.shape[0], :a.shape[1]], a)


@pytest.mark.parametrize("t", TYPES)
def test_triangular_solver_rectangular_blocks(dlaf, grid, oracle, t):
    """B with MB x NB blocks, MB != NB (solver/triangular.h:41-60: only the block along the triangular dimension is
    tied to A's): the reference's size list, analytic systems for the small ones, the oracle's trsm on random
    operands for the multi-tile ones."""
    dt = oracle.DTYPES[t]
    cx = t in "cz"
    alpha = dt(complex(-1.2, .7)) if cx else dt(-1.2)
    rng = np.random.default_rng(5)
    for (m, n, mb, nb), side, uplo, op, diag in itertools.product(RECT_SIZES, "LR", "LU", "NTC", "NU"):
        na, nba = (m, mb) if side == "L" else (n, nb)
        if max(m, n) <= 25:
            a, b, x = oracle.triangular_system(side, uplo, op, diag, alpha, m, n, dt)
        else:
            a = rng.uniform(-1, 1, (na, na)) + (1j * rng.uniform(-1, 1, (na, na)) if cx else 0)
            a = np.asfortranarray((a / na + 2 * np.eye(na)).astype(dt))
            b = np.asfortranarray((rng.uniform(-1, 1, (m, n)) + (1j * rng.uniform(-1, 1, (m, n)) if cx else 0)).astype(dt))
            x = b.copy(order="F")
            oracle.trsm(side, uplo, op, diag, alpha, a, x)
        sb = np.full((m + 2, max(1, n)), 6.5, dtype=dt, order="F")
        sb[:m, :n] = b
        dlaf.triangular_solver(grid, side, uplo, op, diag, alpha, np.asfortranarray(a), sb[:m, :n], nba, b_block=(mb, nb))
        tol = 40 * (m + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(x, sb[:m, :n], tol, tol)
        assert ok, (md, tol, m, n, mb, nb, side, uplo, op, diag)
        assert (sb[m:, :] == 6.5).all()


@pytest.mark.parametrize("t", TYPES)
def test_triangular_solver_random_vs_oracle(dlaf, grid, oracle, t):
    """Random (well-conditioned) operands: the result is the oracle's generic trsm (blas/tile.h:359-366)."""
    dt = oracle.DTYPES[t]
    rng = np.random.default_rng(7)
    cx = t in "cz"
    # (the last two sizes are whole tiles along the triangular dimension: the variants that sweep an upper triangular
    # matrix run on its reversed image there, csrc/host/solver.cpp)
    for (m, n, nb), side, uplo, op, diag in itertools.product([(260, 200, 64), (190, 321, 128), (256, 192, 64), (130, 384, 128)],
                                                              "LR", "LU", "NTC", "NU"):
        na = m if side == "L" else n
        a = rng.uniform(-1, 1, (na, na)) + (1j * rng.uniform(-1, 1, (na, na)) if cx else 0)
        a = np.asfortranarray((a / na + 2 * np.eye(na)).astype(dt))
        b = np.asfortranarray((rng.uniform(-1, 1, (m, n)) + (1j * rng.uniform(-1, 1, (m, n)) if cx else 0)).astype(dt))
        alpha = dt(complex(.7, -.4)) if cx else dt(.7)
        ref = b.copy(order="F")
        oracle.trsm(side, uplo, op, diag, alpha, a, ref)
        got = b.copy(order="F")
        dlaf.triangular_solver(grid, side, uplo, op, diag, alpha, a, got, nb)
        tol = 40 * (m + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (md, tol, m, n, nb, side, uplo, op, diag)


def test_pdtrsm_after_pdpotrf_solves_the_system(dlaf, grid, oracle):
    """The use the solver exists for: A x = b with the Cholesky factor, through the ScaLAPACK-style entries."""
    n, nrhs, nb = 300, 70, 64
    a0 = oracle.set_random_hpd(n, nb, np.float64)
    rng = np.random.default_rng(3)
    xs = np.asfortranarray(rng.uniform(-1, 1, (n, nrhs)))
    rhs = np.asfortranarray(a0 @ xs)
    fact = a0.copy(order="F")
    desca = [1, grid.context, n, n, nb, nb, 0, 0, n]
    descb = [1, grid.context, n, nrhs, nb, nb, 0, 0, n]
    assert dlaf.pxpotrf("L", n, fact, 1, 1, desca) == 0
    dlaf.pxtrsm("L", "L", "N", "N", n, nrhs, 1.0, fact, 1, 1, desca, rhs, 1, 1, descb)
    dlaf.pxtrsm("L", "L", "C", "N", n, nrhs, 1.0, fact, 1, 1, desca, rhs, 1, 1, descb)
    assert np.abs(rhs - xs).max() <= 100 * n * oracle.eps_of(np.float64)
    # the same through p?potrs, upper factor, complex
    a0 = oracle.set_random_hpd(n, nb, np.complex128)
    xs = np.asfortranarray(rng.uniform(-1, 1, (n, nrhs)) + 1j * rng.uniform(-1, 1, (n, nrhs)))
    rhs = np.asfortranarray(a0 @ xs)
    fact = a0.copy(order="F")
    assert dlaf.pxpotrf("U", n, fact, 1, 1, desca) == 0
    assert dlaf.pxpotrs("U", n, nrhs, fact, 1, 1, desca, rhs, 1, 1, descb) == 0
    assert np.abs(rhs - xs).max() <= 100 * n * oracle.eps_of(np.float64)


@pytest.mark.parametrize("t", TYPES)
def test_triangular_solver_on_resident_operands(dlaf, grid, oracle, t):
    """dlaf_mi355x_triangular_solver_device: the triangular matrix in the uplo triangle of a resident matrix, the
    right-hand sides in a resident general matrix, every side / uplo / op / diag against the oracle's trsm; the
    other triangle of A holds junk that must not be read."""
    dt = oracle.DTYPES[t]
    rng = np.random.default_rng(11)
    cx = t in "cz"
    for (m, n, nb), side, uplo, op, diag in itertools.product([(260, 200, 64), (129, 257, 128)], "LR", "LU", "NTC", "NU"):
        na = m if side == "L" else n
        a = rng.uniform(-1, 1, (na, na)) + (1j * rng.uniform(-1, 1, (na, na)) if cx else 0)
        a = (a / na + 2 * np.eye(na)).astype(dt)
        tri = np.tril(a) if uplo == "L" else np.triu(a)
        junk = np.full((na, na), dt(-9.9))
        a_in = np.asfortranarray(tri + (np.triu(junk, 1) if uplo == "L" else np.tril(junk, -1)))
        b = np.asfortranarray((rng.uniform(-1, 1, (m, n)) + (1j * rng.uniform(-1, 1, (m, n)) if cx else 0)).astype(dt))
        alpha = dt(complex(.7, -.4)) if cx else dt(.7)
        ref = b.copy(order="F")
        oracle.trsm(side, uplo, op, diag, alpha, np.asfortranarray(tri), ref)
        am = dlaf.DeviceMatrix(grid, dt, uplo, na, nb)
        am.upload(a_in)
        bm = dlaf.GeneralDeviceMatrix(grid, dt, m, n, nb)
        bm.upload(b)
        dlaf.triangular_solver_device(side, uplo, op, diag, alpha, am, bm)
        got = np.zeros((m, n), dtype=dt, order="F")
        bm.download(got)
        tol = 40 * (m + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (md, tol, m, n, nb, side, uplo, op, diag)
        am.close()
        bm.close()


@pytest.mark.parametrize("t,uplo", [("d", "L"), ("d", "U"), ("z", "L"), ("z", "U")])
def test_potrf_then_potrs_without_leaving_hbm(dlaf, grid, oracle, t, uplo):
    """p?potrf -> p?potrs chained on resident matrices: upload A and B once, factor, solve, download X."""
    dt = oracle.DTYPES[t]
    n, nrhs, nb = 600, 200, 128
    a0 = oracle.set_random_hpd(n, nb, dt)
    rng = np.random.default_rng(5)
    xs = rng.uniform(-1, 1, (n, nrhs)) + (1j * rng.uniform(-1, 1, (n, nrhs)) if t == "z" else 0)
    xs = np.asfortranarray(xs.astype(dt))
    rhs = np.asfortranarray(a0 @ xs)
    am = dlaf.DeviceMatrix(grid, dt, uplo, n, nb)
    am.upload(a0)
    bm = dlaf.GeneralDeviceMatrix(grid, dt, n, nrhs, nb)
    bm.upload(rhs)
    assert am.factorize() == 0
    dlaf.potrs_device(uplo, am, bm)
    got = np.zeros((n, nrhs), dtype=dt, order="F")
    bm.download(got)
    assert np.abs(got - xs).max() <= 100 * n * oracle.eps_of(dt)
    am.close()
    bm.close()
