"""CPU tests that PIN the oracle (oracle/) against the reference's own known answers.

Mirrors, with the reference's tolerances:
  test/unit/matrix/test_util_distribution.cpp:41-95      index math golden rows
  test/unit/matrix/test_util_matrix.cpp:182-212          SPD generator properties
  test/unit/test_lapack_tile/test_potrf.h:33-77          tile potrf (+ non-SPD info)
  test/unit/test_blas_tile/test_{trsm,herk,gemm}.h       tile BLAS closed forms
  test/unit/factorization/test_cholesky.cpp:54-120       local + distributed factorization
"""
import ctypes as C
import itertools
import json
import os

import numpy as np
import pytest

TYPES = ["s", "d", "c", "z"]
# test/unit/factorization/test_cholesky.cpp:54-58
CHOLESKY_SIZES = [(0, 2), (5, 8), (34, 34), (4, 3), (16, 10), (34, 13), (32, 5)]
# test/include/dlaf_test/comm_grids/grids_6_ranks.h:26-71 -> 3x2, 2x3 and the split {3x1, 1x2, 1x1}
GRIDS = [(3, 2), (2, 3), (3, 1), (1, 2), (1, 1)]


def err_of(orc, t):
    # test/include/dlaf_test/util_types.h:40,62: 2 eps real, 8 eps complex
    return (8 if t in "cz" else 2) * orc.eps_of(orc.DTYPES[t])


# ------------------------------------------------------------------ index math
def test_distribution_golden_rows(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "distribution_rows.json")))
    L = oracle.lib()
    assert len(g["rows"]) == 32
    for row in g["rows"]:
        p = dict(zip(g["columns"], row))
        assert L.orc_tile_from_element(p["global_element"], p["tile_size"], p["tile_element_offset"]) == p["global_tile"]
        assert L.orc_tile_element_from_element(p["global_element"], p["tile_size"],
                                               p["tile_element_offset"]) == p["tile_element"]
        assert L.orc_element_from_tile_and_tile_element(p["global_tile"], p["tile_element"], p["tile_size"],
                                                        p["tile_element_offset"]) == p["global_element"]
        assert L.orc_rank_global_tile(p["global_tile"], p["tiles_per_block"], p["grid_size"], p["src_rank"],
                                      p["tile_offset"]) == p["rank_tile"]
        assert L.orc_local_tile_from_global_tile(p["global_tile"], p["tiles_per_block"], p["grid_size"], p["rank"],
                                                 p["src_rank"], p["tile_offset"]) == p["local_tile"]
        assert L.orc_next_local_tile_from_global_tile(p["global_tile"], p["tiles_per_block"], p["grid_size"],
                                                      p["rank"], p["src_rank"], p["tile_offset"]) == p["local_tile_next"]
        if p["local_tile"] >= 0:
            assert L.orc_global_tile_from_local_tile(p["local_tile"], p["tiles_per_block"], p["grid_size"], p["rank"],
                                                     p["src_rank"], p["tile_offset"]) == p["global_tile"]


def test_distribution_against_reference_header(oracle):
    """oracle/_ref/libref_distribution.so = the reference's own util_distribution.h compiled
    where it lies (oracle/Makefile `ref`); exhaustive sweep of the restatement against it."""
    path = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libref_distribution.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference not present at build time)")
    R = C.CDLL(path)
    L = oracle.lib()
    for f in ("ref_local_tile_from_global_tile", "ref_next_local_tile_from_global_tile",
              "ref_global_tile_from_local_tile", "ref_tile_from_element", "ref_tile_element_from_element",
              "ref_element_from_tile_and_tile_element"):
        getattr(R, f).restype = C.c_long
        getattr(R, f).argtypes = None
    lg = C.c_long
    n = 0
    for gs in (1, 2, 3, 5):
        for tpb in (1, 2, 4):
            for src in range(gs):
                for toff in range(tpb):
                    for rank in range(gs):
                        for gt in range(0, 41, 1):
                            a = (lg(gt), lg(tpb), gs, src, lg(toff))
                            assert R.ref_rank_global_tile(*a) == L.orc_rank_global_tile(gt, tpb, gs, src, toff)
                            b = (lg(gt), lg(tpb), gs, rank, src, lg(toff))
                            lt = L.orc_local_tile_from_global_tile(gt, tpb, gs, rank, src, toff)
                            assert R.ref_local_tile_from_global_tile(*b) == lt
                            assert R.ref_next_local_tile_from_global_tile(*b) == \
                                L.orc_next_local_tile_from_global_tile(gt, tpb, gs, rank, src, toff)
                            if lt >= 0:
                                assert R.ref_global_tile_from_local_tile(lg(lt), lg(tpb), gs, rank, src, lg(toff)) == gt
                                assert L.orc_global_tile_from_local_tile(lt, tpb, gs, rank, src, toff) == gt
                            n += 1
    for ts in (1, 7, 10):
        for off in range(ts):
            for e in range(0, 50):
                assert R.ref_tile_from_element(lg(e), lg(ts), lg(off)) == L.orc_tile_from_element(e, ts, off)
                assert R.ref_tile_element_from_element(lg(e), lg(ts), lg(off)) == \
                    L.orc_tile_element_from_element(e, ts, off)
    assert n > 1000


def test_local_sizes_sum_to_global(oracle):
    for n, nb in [(0, 2), (5, 8), (34, 13), (32, 5), (100, 7)]:
        for grid in (1, 2, 3, 4):
            for src in range(grid):
                assert sum(oracle.local_size(n, nb, grid, r, src) for r in range(grid)) == n
                nt = (n + nb - 1) // nb if n else 0
                assert sum(oracle.local_nr_tiles(n, nb, grid, r, src) for r in range(grid)) == nt


# ------------------------------------------------------------------ RNG / generator
def test_rng_stream_matches_libstdcxx(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "rng_stream.json")))
    for seed, rec in g.items():
        m = oracle.MT19937_64(int(seed))
        assert [m.raw() for _ in rec["raw"]] == rec["raw"]
        m = oracle.MT19937_64(int(seed))
        assert [m.uniform_d() for _ in rec["d"]] == rec["d"]
        m = oracle.MT19937_64(int(seed))
        got = np.array([m.uniform_s() for _ in rec["s"]], dtype=np.float32)
        assert (got == np.array(rec["s"], dtype=np.float32)).all()


def test_complex_sample_order_matches_gxx(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "rng_stream.json")))
    # seed 0 = diagonal tile (0,0): first complex draw lands at (0,0) as real(draw)+2n, second at (0,1)
    for t in "zc":
        rec = g["0"][t]
        n = 20
        tile = oracle.random_hpd_tile(n, 8, 0, 0, oracle.DTYPES[t])
        rt = oracle.REAL_OF[t]
        assert tile[0, 0].real == rt(rt(rec[0]) + rt(2 * n)) and tile[0, 0].imag == 0
        assert tile[0, 1] == oracle.DTYPES[t](complex(rec[2], rec[3]))
        assert tile[1, 0] == np.conj(tile[0, 1])


@pytest.mark.parametrize("t", TYPES)
def test_random_hpd_properties(oracle, t):
    # test/unit/matrix/test_util_matrix.cpp:182-212: hermitian and |A - 2N I| <= 1 elementwise
    for n, nb in [(34, 13), (32, 5), (16, 16), (7, 9)]:
        a = oracle.set_random_hpd(n, nb, oracle.DTYPES[t])
        assert np.array_equal(a, a.conj().T)
        assert (np.abs(a - 2 * n * np.eye(n)) <= 1 + 1e-6).all()
        assert (np.diag(a).imag == 0).all()
        # tiles are independent of the surrounding matrix layout: regenerate one tile alone
        tile = oracle.random_hpd_tile(n, nb, (n - 1) // nb, 0, oracle.DTYPES[t])
        assert np.array_equal(tile, a[((n - 1) // nb) * nb:, :min(nb, n)])


def test_random_hpd_golden_samples(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "hpd_34_13.json")))
    for t in TYPES:
        a = oracle.set_random_hpd(g["n"], g["nb"], oracle.DTYPES[t])
        s = g["samples"][t]

        def dec(e):
            v = np.array(e["re"])
            return v + 1j * np.array(e["im"]) if "im" in e else v
        assert np.array_equal(a[:, 0].astype(np.complex128), dec(s["col0"]).astype(np.complex128))
        assert np.array_equal(np.diag(a).astype(np.complex128), dec(s["diag"]).astype(np.complex128))


# ------------------------------------------------------------------ tile kernels
@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_tile_potrf_analytic(oracle, t, uplo):
    # test_potrf.h:33-57, sizes test_lapack_tile.cpp:142-145
    for n, extra in itertools.product([0, 1, 11, 12, 17, 128], [0, 3]):
        a, l = oracle.cholesky_setters(uplo, n, oracle.DTYPES[t])
        store = np.full((max(1, n) + extra, max(1, n)), 3.3, dtype=a.dtype, order="F")
        store[:n, :n] = a
        assert oracle.potrf(uplo, store[:n, :n]) == 0
        tol = 4 * (n + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(l, store[:n, :n], tol, tol)
        assert ok, (n, md)
        assert (store[n:, :] == 3.3).all()


@pytest.mark.parametrize("t", TYPES)
def test_tile_potrf_non_spd(oracle, t):
    # test_potrf.h:59-77: null matrix -> info == 1
    for uplo in "LU":
        a = np.zeros((5, 5), dtype=oracle.DTYPES[t], order="F")
        assert oracle.potrf(uplo, a) == 1


def _polar(orc, t, r, theta):
    if t in "sd":
        return orc.DTYPES[t](r)
    rt = orc.REAL_OF[t]
    return orc.DTYPES[t](complex(rt(r) * np.cos(rt(theta)), rt(r) * np.sin(rt(theta))))


@pytest.mark.parametrize("t", TYPES)
def test_tile_gemm_closed_form(oracle, t):
    # test_gemm.h:33-69 with getMatrixMatrixMultiplication (util_generic_blas.h:54-94)
    dt = oracle.DTYPES[t]
    alpha = dt(complex(-1.2, .7)) if t in "cz" else dt(-1.2)
    beta = dt(complex(1.1, .4)) if t in "cz" else dt(1.1)
    for (m, n, k) in [(0, 0, 0), (3, 0, 2), (7, 5, 0), (3, 5, 7), (13, 32, 8), (12, 12, 5)]:
        for opa, opb in itertools.product("NTC", "NTC"):
            opA = np.array([[_polar(oracle, t, .9 * (i + 1) / (kk + .5), 2 * i - kk) for kk in range(k)]
                            for i in range(m)], dtype=dt).reshape(m, k)
            opB = np.array([[_polar(oracle, t, .8 * (kk + .5) / (j + 2), kk + j) for j in range(n)]
                            for kk in range(k)], dtype=dt).reshape(k, n)
            c = np.array([[_polar(oracle, t, 1.2 * i / (j + 1), -i + j) for j in range(n)] for i in range(m)],
                         dtype=dt).reshape(m, n)
            gamma = dt(.72 * k) * alpha
            res = np.array([[beta * c[i, j] + gamma * _polar(oracle, t, (i + 1) / (j + 2), 2 * i + j)
                             for j in range(n)] for i in range(m)], dtype=dt).reshape(m, n)
            unop = {"N": lambda x: x, "T": lambda x: x.T, "C": lambda x: x.conj().T}
            a = np.asfortranarray(unop[opa](opA))
            b = np.asfortranarray(unop[opb](opB))
            cc = np.asfortranarray(c.copy())
            if m and n:
                oracle.gemm(opa, opb, alpha, a if a.size else np.zeros((1, 1), dt, order="F"),
                            b if b.size else np.zeros((1, 1), dt, order="F"), beta, cc, k=k)
            tol = 2 * (k + 1) * err_of(oracle, t)
            ok, md = oracle.check_near(res, cc, tol, tol)
            assert ok, (m, n, k, opa, opb, md)


@pytest.mark.parametrize("t", TYPES)
def test_tile_herk_closed_form(oracle, t):
    # test_herk.h:33-89; the hot path uses (L,N) and (U,C)
    dt = oracle.DTYPES[t]
    alpha, beta = -1.2, 1.1
    for n, k in [(0, 0), (0, 2), (5, 0), (5, 3), (9, 16), (13, 13)]:
        for uplo, op in itertools.product("LU", "NC"):
            opA = np.array([[_polar(oracle, t, .9 * (i + 1) / (kk + .5), i - kk) for kk in range(k)]
                            for i in range(n)], dtype=dt).reshape(n, k)

            def el_c(i, j):
                if (uplo == "L" and i < j) or (uplo == "U" and i > j):
                    return dt(-1)
                return _polar(oracle, t, 1.2 * i / (j + 1), -i + j)
            c = np.array([[el_c(i, j) for j in range(n)] for i in range(n)], dtype=dt).reshape(n, n)
            res = c.copy()
            for i in range(n):
                for j in range(n):
                    if (uplo == "L" and i < j) or (uplo == "U" and i > j):
                        continue
                    tmp = dt(0)
                    for kk in range(k):
                        tmp += opA[i, kk] * np.conj(opA[j, kk])
                    res[i, j] = dt(beta) * c[i, j] + dt(alpha) * tmp
            a = np.asfortranarray(opA if op == "N" else opA.conj().T)
            cc = np.asfortranarray(c.copy())
            if n:
                oracle.herk(uplo, op, alpha, a if a.size else np.zeros((1, 1), dt, order="F"), beta, cc, k=k)
            tol = (k + 1) * err_of(oracle, t)
            # herk forces a real diagonal; the closed form has one up to rounding
            ok, md = oracle.check_near(res, cc, tol, tol)
            assert ok, (n, k, uplo, op, md)
            if t in "cz" and n:
                assert (np.diag(cc).imag == 0).all()


@pytest.mark.parametrize("t", TYPES)
def test_tile_trsm_closed_form(oracle, t):
    # test_trsm.h:35-62 with getTriangularSystem (util_generic_blas.h:258-373): all variants
    dt = oracle.DTYPES[t]
    alpha = dt(complex(-1.2, .7)) if t in "cz" else dt(-1.2)
    for (m, n) in [(0, 0), (3, 0), (0, 5), (3, 5), (17, 13), (13, 17)]:
        for side, uplo, op, diag in itertools.product("LR", "LU", "NTC", "NU"):
            op_a_lower = (uplo == "L" and op == "N") or (uplo == "U" and op != "N")
            na = m if side == "L" else n
            opA = np.full((na, na), -9.9, dtype=dt)
            x = np.zeros((m, n), dtype=dt)
            b = np.zeros((m, n), dtype=dt)
            for i in range(na):
                for kk in range(na):
                    if (op_a_lower and i < kk) or (not op_a_lower and i > kk) or (diag == "U" and i == kk):
                        continue
                    if side == "L":
                        opA[i, kk] = _polar(oracle, t, (i + 1) / (kk + .5), 2 * i - kk)
                    else:
                        opA[i, kk] = _polar(oracle, t, (kk + 1) / (i + .5), 2 * kk - i)
            for i in range(m):
                for j in range(n):
                    if side == "L":
                        x[i, j] = _polar(oracle, t, (i + .5) / (j + 2), i + j)
                        kk = (i + 1) if op_a_lower else (m - i)
                        gamma = _polar(oracle, t, (i + 1) / (j + 2), 2 * i + j)
                    else:
                        x[i, j] = _polar(oracle, t, (j + .5) / (i + 2), i + j)
                        kk = (n - j) if op_a_lower else (j + 1)
                        gamma = _polar(oracle, t, (j + 1) / (i + 2), i + 2 * j)
                    b[i, j] = ((kk - 1) * gamma + x[i, j]) / alpha if diag == "U" else kk * gamma / alpha
            unop = {"N": lambda z: z, "T": lambda z: z.T, "C": lambda z: z.conj().T}
            a = np.asfortranarray(unop[op](opA)) if na else np.zeros((1, 1), dt, order="F")
            bb = np.asfortranarray(b.copy())
            if m and n:
                oracle.trsm(side, uplo, op, diag, alpha, a, bb)
            tol = 10 * (max(m, n) + 1) * err_of(oracle, t)
            ok, md = oracle.check_near(x, bb, tol, tol)
            assert ok, (m, n, side, uplo, op, diag, md)


@pytest.mark.parametrize("t", TYPES)
def test_triangular_system_generator(oracle, t):
    # oracle.triangular_system is the vectorised getTriangularSystem the solver tests use: it must agree with
    # the element-by-element restatement above, i.e. the generic trsm maps its (A, B) onto its X
    dt = oracle.DTYPES[t]
    alpha = dt(complex(-1.2, .7)) if t in "cz" else dt(-1.2)
    for (m, n) in [(3, 5), (17, 13), (13, 17)]:
        for side, uplo, op, diag in itertools.product("LR", "LU", "NTC", "NU"):
            a, b, x = oracle.triangular_system(side, uplo, op, diag, alpha, m, n, dt)
            bb = b.copy(order="F")
            oracle.trsm(side, uplo, op, diag, alpha, a, bb)
            tol = 10 * (max(m, n) + 1) * err_of(oracle, t)
            ok, md = oracle.check_near(x, bb, tol, tol)
            assert ok, (m, n, side, uplo, op, diag, md)


# ------------------------------------------------------------------ factorization
@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_cholesky_local_analytic(oracle, t, uplo):
    # test_cholesky.cpp:60-77 (CorrectnessLocal)
    for m, mb in CHOLESKY_SIZES:
        a, l = oracle.cholesky_setters(uplo, m, oracle.DTYPES[t])
        assert oracle.cholesky_local(uplo, a, mb) == 0
        tol = 4 * (m + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(l, a, tol, tol)
        assert ok, (m, mb, md)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_cholesky_distributed_analytic(oracle, t, uplo):
    # test_cholesky.cpp:79-120 (CorrectnessDistributed): non-zero source rank (:85)
    for pr, pc in GRIDS:
        sr, sc = max(0, pr - 1), min(1, pc - 1)
        for m, mb in CHOLESKY_SIZES:
            a, l = oracle.cholesky_setters(uplo, m, oracle.DTYPES[t])
            locs = oracle.scatter(a, mb, pr, pc, sr, sc, extra_ld=2)
            assert oracle.cholesky_dist(uplo, locs, m, mb, pr, pc, sr, sc) == 0
            out = oracle.gather(locs, m, mb, pr, pc, sr, sc)
            tol = 4 * (m + 1) * err_of(oracle, t)
            ok, md = oracle.check_near(l, out, tol, tol)
            assert ok, (pr, pc, m, mb, md)


@pytest.mark.parametrize("t", TYPES)
def test_cholesky_dist_equals_local_bitwise(oracle, t):
    """Same tile kernels, same order per tile -> the distributed schedule must reproduce the
    local factorization bit for bit on random SPD input."""
    n, nb = 45, 8
    a0 = oracle.set_random_hpd(n, nb, oracle.DTYPES[t])
    for uplo in "LU":
        ref = a0.copy(order="F")
        assert oracle.cholesky_local(uplo, ref, nb) == 0
        for pr, pc in [(2, 3), (3, 2), (2, 2)]:
            locs = oracle.scatter(a0, nb, pr, pc, 1 % pr, 1 % pc)
            assert oracle.cholesky_dist(uplo, locs, n, nb, pr, pc, 1 % pr, 1 % pc) == 0
            out = oracle.gather(locs, n, nb, pr, pc, 1 % pr, 1 % pc)
            assert np.array_equal(out, ref)


@pytest.mark.parametrize("t", TYPES)
def test_cholesky_vs_lapack_golden(oracle, t, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "potrf_lapack.json")))
    n, nb = g["n"], g["nb"]
    a0 = oracle.set_random_hpd(n, nb, oracle.DTYPES[t])
    for uplo in "LU":
        e = g["factors"][f"{t}{uplo}"]
        exp = np.array(e["re"]).reshape(e["shape"], order="F")
        if "im" in e:
            exp = exp + 1j * np.array(e["im"]).reshape(e["shape"], order="F")
        a = a0.copy(order="F")
        assert oracle.cholesky_local(uplo, a, nb) == 0
        got = oracle.tri(uplo, a)
        tol = 4 * (n + 1) * err_of(oracle, t)
        ok, md = oracle.check_near(exp, got, tol, tol)
        assert ok, md
        # other triangle untouched
        other = np.triu(a, 1) if uplo == "L" else np.tril(a, -1)
        other0 = np.triu(a0, 1) if uplo == "L" else np.tril(a0, -1)
        assert np.array_equal(other, other0)
        # miniapp checker bar: residual <= n*eps
        assert oracle.cholesky_residual(uplo, a0, a) <= n * oracle.eps_of(oracle.DTYPES[t])


def test_non_spd_reports_global_index(oracle):
    n, nb = 20, 6
    a = oracle.set_random_hpd(n, nb, np.float64)
    a[13, 13] = -1.0
    assert oracle.cholesky_local("L", a.copy(order="F"), nb) == 14
    locs = oracle.scatter(a, nb, 2, 2)
    assert oracle.cholesky_dist("L", locs, n, nb, 2, 2) == 14


def test_baseline_matches_oracle(oracle):
    n, nb = 200, 48
    a0 = oracle.set_random_hpd(n, nb, np.float64)
    ref = a0.copy(order="F")
    assert oracle.cholesky_local("L", ref, nb) == 0
    got = a0.copy(order="F")
    assert oracle.baseline_cholesky_d(got, nb, 4) == 0
    assert np.allclose(np.tril(got), np.tril(ref), rtol=1e-13, atol=1e-13)
    assert np.array_equal(np.triu(got, 1), np.triu(a0, 1))


# ---- gen_to_std: the restatement against the reference's known answers and LAPACK ---------
GEN_TO_STD_SIZES = [(0, 2), (5, 8), (34, 34), (4, 3), (16, 10), (34, 13), (32, 5)]  # test_gen_to_std.cpp:54-58


@pytest.mark.parametrize("t", ["s", "d", "c", "z"])
@pytest.mark.parametrize("uplo", ["L", "U"])
def test_gen_to_std_oracle_reproduces_the_reference_known_answers(oracle, t, uplo):
    """getGenToStdElementSetters (util_generic_lapack.h:96-150) with the parameters and the tolerance of
    test_gen_to_std.cpp:68-83: abs 10 (m + 1) error, other triangle untouched (-9.9)."""
    dt = oracle.DTYPES[t]
    err = (8 if t in "cz" else 2) * oracle.eps_of(dt)
    for m, mb in GEN_TO_STD_SIZES:
        tmat, a, b = oracle.gen_to_std_setters(uplo, m, dt)
        got = a.copy(order="F")
        oracle.gen_to_std_local(uplo, got, tmat, mb)
        ok, md = oracle.check_near(b, got, 0, 10 * (m + 1) * err)
        assert ok, (m, mb, md)


def test_gen_to_std_oracle_matches_lapack_hegst(oracle):
    """Independent numeric answer: LAPACK xSYGST / xHEGST (itype 1) through scipy on random operands."""
    import scipy.linalg.lapack as la
    for t, fn in (("d", la.dsygst), ("z", la.zhegst)):
        dt = oracle.DTYPES[t]
        n, nb = 70, 16
        b0 = oracle.set_random_hpd(n, nb, dt)
        a0 = oracle.set_random_hpd(n, nb, dt) * dt(0.01)
        for uplo in "LU":
            fac = b0.copy(order="F")
            assert oracle.cholesky_local(uplo, fac, nb) == 0
            got = a0.copy(order="F")
            oracle.gen_to_std_local(uplo, got, fac, nb)
            ref, info = fn(a0.copy(order="F"), fac, itype=1, lower=1 if uplo == "L" else 0)
            assert info == 0
            assert np.abs(oracle.tri(uplo, ref) - oracle.tri(uplo, got)).max() < 50 * n * oracle.eps_of(dt)
            other = np.triu(got, 1) if uplo == "L" else np.tril(got, -1)
            assert np.array_equal(other, np.triu(a0, 1) if uplo == "L" else np.tril(a0, -1))


def test_rectangular_block_scatter_gather_round_trip():
    """scatter_rect / gather_rect (MB x NB blocks, matrix.h block size != square -- the right-hand sides of the solver
    tests): the square-block case equals scatter(), every case round-trips, every element lands exactly once."""
    from oracle import oracle
    a = np.arange(19 * 25, dtype=np.float64).reshape(19, 25)
    for pr, pc, sr, sc in [(1, 1, 0, 0), (2, 3, 1, 1), (3, 2, 2, 0), (1, 3, 0, 2)]:
        for mb, nb in [(6, 5), (3, 9), (5, 5), (19, 1)]:
            locs = oracle.scatter_rect(a, mb, nb, pr, pc, sr, sc, extra_ld=2)
            assert sum(v.size for v in locs.values()) == a.size
            assert np.array_equal(oracle.gather_rect(locs, 19, 25, mb, nb, pr, pc, sr, sc), a)
        sq = oracle.scatter(a, 5, pr, pc, sr, sc)
        rc = oracle.scatter_rect(a, 5, 5, pr, pc, sr, sc)
        assert all(np.array_equal(sq[k], rc[k]) for k in sq)
