"""The BASELINE.json configurations that fit one MI355X, at their stated sizes, plus elementwise parity at the
headline block size nb = 1024 and the tile kernels at full fast-path shapes.

  C0  d N=4096  nb=256  (miniapp_cholesky plumbing config): dlaf_pdpotrf vs the oracle elementwise + the miniapp
  C1  d N=32768 nb=512  one GPU
  C2  d N=65536 nb=1024 (the metric's configuration; its 1x1 workload)
  C3  z N=32768 nb=512  (the pzpotrf type) on one GPU
At the sizes the oracle cannot factor in seconds the checks are the size-independent ones of the reference:
residual max|A - L L^H| / max|A| <= n eps (miniapp/miniapp_cholesky.cpp:432-442) computed on the device, a real
positive diagonal, and the opposite triangle untouched (test/unit/factorization/test_cholesky.cpp:54-120 fills it
with a sentinel) -- sampled tile by tile (dlaf_mi355x_matrix_fetch_tile), never a 32 GiB download."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


# ---------------------------------------------------------------------------------------------- C0
def test_C0_pdpotrf_n4096_nb256_vs_oracle(dlaf, grid, oracle):
    n, nb = 4096, 256
    a0 = oracle.set_random_hpd(n, nb, np.float64)
    a = a0.copy(order="F")
    assert dlaf.pxpotrf("L", n, a, 1, 1, [1, grid.context, n, n, nb, nb, 0, 0, n]) == 0
    ref = a0.copy(order="F")
    # the oracle's tile DAG with its tile tasks spread over the host cores (same tile kernels, same order of the
    # sums inside a tile as cholesky_local: 245 s single-threaded at this size, a third of the GPU suite in round 2)
    assert oracle.baseline_cholesky_d(ref, nb, max(1, min(16, os.cpu_count() or 1))) == 0
    tol = 4 * (n + 1) * err_of(oracle, "d")  # test_cholesky.cpp:76-77
    ok, md = oracle.check_near(np.tril(ref), np.tril(a), tol, tol)
    assert ok, md
    assert np.array_equal(np.triu(a, 1), np.triu(a0, 1))
    assert oracle.cholesky_residual("L", a0, a) <= n * np.finfo(np.float64).eps


def test_C0_miniapp_cholesky_n4096_nb256(dlaf):
    from test_cpp_api import build_miniapp, check_miniapp_output
    exe = build_miniapp()
    r = subprocess.run([exe, "--matrix-size", "4096", "--block-size", "256", "--type", "d", "--nruns", "2", "--nwarmups", "1",
                        "--check-result", "last", "--csv"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DLAF_MI355X_DEVICE="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    check_miniapp_output(r.stdout, 2, 1)
    assert "(4096, 4096) (256, 256) (1, 1)" in r.stdout


# ------------------------------------------------------------------------------------ C1, C2, C3-type
def sample_tiles(nt):
    """tile coordinates to sample: the corners and middle of the lower triangle, the diagonal ends, and their
    mirror images in the opposite triangle"""
    mid = nt // 2
    lower = {(nt - 1, 0), (mid, 0), (nt - 1, mid), (mid, mid - 1), (1, 0), (nt - 1, nt - 2)}
    diag = {(0, 0), (mid, mid), (nt - 1, nt - 1), (1, 1)}
    return sorted(lower), sorted(diag)


@pytest.mark.parametrize("t,n,nb,uplo", [("d", 32768, 512, "L"), ("d", 65536, 1024, "L"), ("z", 32768, 512, "L"),
                                         ("d", 16384, 512, "U")],
                         ids=["C1_d_N32768_nb512", "C2_d_N65536_nb1024_1x1", "C3type_z_N32768_nb512", "C1type_d_N16384_nb512_U"])
def test_baseline_config_on_one_gpu(dlaf, grid, oracle, t, n, nb, uplo):
    dt = oracle.DTYPES[t]
    eps = oracle.eps_of(dt)
    nt = n // nb
    a = np.zeros((n, n), dtype=dt, order="F")
    dlaf.set_random_hermitian_positive_definite(grid, a, n, nb)
    orig = dlaf.DeviceMatrix(grid, dt, uplo, n, nb)
    fact = dlaf.DeviceMatrix(grid, dt, uplo, n, nb)
    orig.upload(a)
    lower, diag = sample_tiles(nt)
    in_tri = (lambda i, j: (i, j)) if uplo == "L" else (lambda i, j: (j, i))
    host = {(i, j): a[i * nb:(i + 1) * nb, j * nb:(j + 1) * nb].copy() for (i, j) in (in_tri(*c) for c in lower + diag)}
    del a
    fact.copy_from(orig)
    # the opposite triangle on the device before the run (diagonal tiles carry their other half; the tiles of
    # the opposite triangle are never uploaded, whatever they hold must still be there afterwards)
    before = {c: fact.fetch_tile(*c) for c in [in_tri(j, i) for (i, j) in lower] + [in_tri(*c) for c in diag]}
    assert fact.factorize() == 0
    for c in [in_tri(j, i) for (i, j) in lower]:
        # (never-uploaded memory may hold NaN patterns: compare bytes)
        assert before[c].tobytes() == fact.fetch_tile(*c).tobytes(), ("opposite-triangle tile changed", c)
    # (what the other half of a diagonal tile holds on the device is the library's business -- it never goes back
    # to the caller -- but the factorization must not write there: compare with the bytes before the run)
    other = (lambda x: np.triu(x, 1)) if uplo == "L" else (lambda x: np.tril(x, -1))
    for c in (in_tri(*c) for c in diag):
        d = fact.fetch_tile(*c)
        assert other(d).tobytes() == other(before[c]).tobytes(), ("other half of a diagonal tile changed", c)
        dg = np.diag(d)
        assert (dg.real > 0).all() and (dg.imag == 0).all(), c
        # a factor's diagonal: sqrt of something in [n, 3n] minus what the earlier columns took
        assert dg.real.max() <= np.sqrt(3.0 * n + 1) and dg.real.min() >= np.sqrt(0.5 * n), c
    for c in (in_tri(*c) for c in lower):
        x = fact.fetch_tile(*c)
        assert np.isfinite(x).all() and not np.array_equal(x, host[c]), c   # solved, not left as uploaded
    # first tile column against the closed form  L(:,0) = A(:,0) L00^-H  (independent of the device TRSM: numpy)
    l00 = np.tril(fact.fetch_tile(0, 0)) if uplo == "L" else np.triu(fact.fetch_tile(0, 0)).conj().T
    a00 = host[(0, 0)]
    a00 = np.tril(a00) + np.tril(a00, -1).conj().T if uplo == "L" else np.triu(a00) + np.triu(a00, 1).conj().T
    assert np.abs(l00 @ l00.conj().T - a00).max() <= 4 * (nb + 1) * eps * np.abs(a00).max()
    ci, cj = in_tri(nt - 1, 0)
    x = fact.fetch_tile(ci, cj)
    x = x if uplo == "L" else x.conj().T
    a_n0 = host[(ci, cj)] if uplo == "L" else host[(ci, cj)].conj().T
    assert np.abs(x @ l00.conj().T - a_n0).max() <= 10 * (nb + 1) * eps * max(1.0, np.abs(x).max() * np.abs(l00).max())
    # whole-matrix residual on the device (overwrites orig; zeroes the strict other half of fact's diagonal tiles)
    diff, norm_a = orig.residual_against(fact)
    assert 2 * n - 1.001 < norm_a < 2 * n + 1.001
    assert diff / norm_a <= n * eps, (diff, norm_a)
    orig.close()
    fact.close()


# ------------------------------------------------------------------- elementwise at the headline block size
def test_baseline_config5_eigensolver_on_one_gpu(dlaf, grid):
    """BASELINE configs[4] at its stated size on one GPU: fp64 symmetric eigensolver N = 20480, nb = 512 (band 128: every
    stage on its fast path, the fused back-transformation included) through the reference's C entry.  Checked by
    size-independent properties: eigenvalues sorted, sum(w) = trace(A) and sum(w^2) = ||A||_F^2 (invariants of the
    similarity transformations), residual and orthogonality of a sample of eigenpairs within the bars of the
    reference's testEigensolverCorrectness (test_eigensolver.cpp: 2 n eps |w|max, 10 n eps)."""
    n, nb = 20480, 512
    rng = np.random.default_rng(20480)
    a0 = np.empty((n, n), dtype=np.float64, order="F")
    for j0 in range(0, n, 2048):
        a0[:, j0:j0 + 2048] = rng.uniform(-1, 1, (n, min(2048, n - j0)))
    low = np.tril(a0)                      # the lower triangle is the matrix
    tr = float(np.trace(low))
    fro2 = float(2.0 * np.sum(low * low) - np.sum(np.diag(low) ** 2))
    a = a0.copy(order="F")
    a[np.triu_indices(n, 1)] = -9.9        # must not be read
    w, z = dlaf.hermitian_eigensolver(grid, "L", a, nb)
    del a
    eps = np.finfo(np.float64).eps
    assert np.all(np.diff(w) >= 0)
    wmax = float(np.abs(w).max())
    assert abs(float(np.sum(w)) - tr) <= 10 * n * eps * wmax * np.sqrt(n)
    assert abs(float(np.sum(w * w)) - fro2) <= 100 * n * eps * fro2
    cols = np.unique(np.concatenate([np.arange(0, n, n // 16), [n - 1]]))
    zc = z[:, cols]
    az = low @ zc + np.tril(a0, -1).T @ zc
    assert np.abs(az - zc * w[cols][None, :]).max() <= 2 * n * 2 * eps * wmax
    assert np.abs(zc.T @ zc - np.eye(len(cols))).max() <= 10 * n * 2 * eps
    # orthogonality against a second sample (different columns)
    other = z[:, cols[:-1] + 1]
    assert np.abs(zc.T @ other).max() <= 10 * n * 2 * eps
    assert all(ms > 0 for ms in dlaf.eigensolver_profile())


@pytest.mark.parametrize("t,uplo,n", [("d", "L", 2048), ("d", "U", 2048), ("z", "L", 2048), ("z", "U", 2048),
                                      ("d", "L", 3072 + 17), ("z", "U", 3072 + 17)])
def test_elementwise_vs_oracle_at_nb1024(dlaf, grid, oracle, t, uplo, n):
    """HIP vs the oracle element by element with nb = 1024 (the block size the headline number is quoted on),
    full tiles and a ragged last tile; tolerance 4 (n+1) err as test_cholesky.cpp:76-77."""
    nb = 1024
    dt = oracle.DTYPES[t]
    a0 = oracle.set_random_hpd(n, nb, dt)
    ref = a0.copy(order="F")
    assert oracle.cholesky_local(uplo, ref, nb) == 0
    got = a0.copy(order="F")
    assert dlaf.cholesky_factorization(grid, uplo, got, nb) == 0
    tol = 4 * (n + 1) * err_of(oracle, t)
    ok, md = oracle.check_near(oracle.tri(uplo, ref), oracle.tri(uplo, got), tol, tol)
    assert ok, (t, uplo, n, md)
    other = np.triu(got, 1) if uplo == "L" else np.tril(got, -1)
    other0 = np.triu(a0, 1) if uplo == "L" else np.tril(a0, -1)
    assert np.array_equal(other, other0)
    assert oracle.cholesky_residual(uplo, a0, got) <= n * oracle.eps_of(dt)


# ------------------------------------------------------------------------ tile kernels at fast-path shapes
def rnd(rng, shape, dt):
    a = rng.uniform(-1, 1, shape)
    if np.issubdtype(dt, np.complexfloating):
        a = a + 1j * rng.uniform(-1, 1, shape)
    return np.asfortranarray(a.astype(dt))


FAST_SHAPES = [("d", 512, 512), ("z", 512, 512), ("d", 1024, 1024), ("z", 1024, 1024), ("s", 512, 512), ("c", 512, 512),
               ("d", 1024, 1000), ("z", 512, 500), ("d", 512, 1024), ("d", 1024, 520)]


@pytest.mark.parametrize("t,n,k", FAST_SHAPES)
def test_tile_gemm_fast_path_shapes(dlaf, oracle, t, n, k):
    """Whole 128x128 blocks, 16-byte aligned operands, K % 16 == 0: the direct-to-LDS pipeline the
    factorization runs (and K % 16 != 0 / K != n for the register-staged edge path), test_gemm.h:33-69."""
    rng = np.random.default_rng(11)
    dt = oracle.DTYPES[t]
    for uplo in "LU":
        a = rnd(rng, (n, k) if uplo == "L" else (k, n), dt)
        b = rnd(rng, (n, k) if uplo == "L" else (k, n), dt)
        c0 = rnd(rng, (n, n), dt)
        ref = c0.copy(order="F")
        if uplo == "L":
            oracle.gemm("N", "C", -1.0, a, b, 1.0, ref)
        else:
            oracle.gemm("C", "N", -1.0, a, b, 1.0, ref)
        got = c0.copy(order="F")
        dlaf.tile_gemm(uplo, a, b, got)
        tol = 2 * (k + 1) * err_of(oracle, t)  # test_gemm.h:68
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (uplo, md)


@pytest.mark.parametrize("t,n,k", FAST_SHAPES)
def test_tile_herk_fast_path_shapes(dlaf, oracle, t, n, k):
    rng = np.random.default_rng(13)
    dt = oracle.DTYPES[t]
    for uplo in "LU":
        a = rnd(rng, (n, k) if uplo == "L" else (k, n), dt)
        c0 = rnd(rng, (n, n), dt)
        ref = c0.copy(order="F")
        oracle.herk(uplo, "N" if uplo == "L" else "C", -1.0, a, 1.0, ref, k=k)
        got = c0.copy(order="F")
        dlaf.tile_herk(uplo, a, got)
        tol = (k + 1) * err_of(oracle, t)  # test_herk.h:88
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (uplo, md)
        other = np.triu(got, 1) if uplo == "L" else np.tril(got, -1)
        other0 = np.triu(c0, 1) if uplo == "L" else np.tril(c0, -1)
        assert np.array_equal(other, other0)


@pytest.mark.parametrize("t,m,n", [("d", 512, 512), ("z", 512, 512), ("d", 1024, 1024), ("z", 1024, 1024), ("s", 512, 512),
                                   ("c", 512, 512), ("d", 1024, 1000), ("d", 1000, 1024), ("z", 384, 520),
                                   ("d", 256, 128), ("d", 640, 384)])   # (widths of whole 128s: the 128-column macro block)
def test_tile_trsm_fast_path_shapes(dlaf, oracle, t, m, n):
    """Whole 128-row strips and 64-column blocks of a full panel tile (test_trsm.h:35-62 argument sets)."""
    rng = np.random.default_rng(17)
    dt = oracle.DTYPES[t]
    for uplo in "LU":
        na = n if uplo == "L" else m
        tri = rnd(rng, (na, na), dt) * dt(0.5 / np.sqrt(na))
        tri[np.arange(na), np.arange(na)] = (np.abs(tri.diagonal()) + 1.5).astype(dt)
        junk = np.full((na, na), dt(-9.9))
        a = np.asfortranarray(np.tril(tri) + np.triu(junk, 1) if uplo == "L" else np.triu(tri) + np.tril(junk, -1))
        b0 = rnd(rng, (m, n), dt)
        ref = b0.copy(order="F")
        if uplo == "L":
            oracle.trsm("R", "L", "C", "N", 1.0, a, ref)
        else:
            oracle.trsm("L", "U", "C", "N", 1.0, a, ref)
        got = b0.copy(order="F")
        dlaf.tile_trsm(uplo, a, got)
        tol = 10 * (max(m, n) + 1) * err_of(oracle, t)  # test_trsm.h:61
        ok, md = oracle.check_near(ref, got, tol, tol)
        assert ok, (uplo, md)


def test_tile_potrf_full_tile_sizes(dlaf, oracle):
    """The diagonal-tile kernel at the tile sizes of the BASELINE configs (256, 512, 1024) against LAPACK-style
    unblocked potf2 of the oracle."""
    for t, n in [("d", 256), ("d", 512), ("d", 1024), ("z", 512), ("z", 1024)]:
        dt = oracle.DTYPES[t]
        a0 = oracle.set_random_hpd(n, n, dt)
        for uplo in "LU":
            ref = a0.copy(order="F")
            assert oracle.potrf(uplo, ref) == 0
            got = a0.copy(order="F")
            assert dlaf.tile_potrf(uplo, got) == 0
            tol = 4 * (n + 1) * err_of(oracle, t)
            ok, md = oracle.check_near(ref, got, tol, tol)
            assert ok, (t, n, uplo, md)


def test_a_live_grid_survives_the_free_of_another(dlaf, oracle):
    """create A, create B, free A, create C: B keeps its context and still factorizes."""
    a = dlaf.Grid.single()
    b = dlaf.Grid.single()
    a.free()
    c = dlaf.Grid.single()
    assert c.context != b.context
    n, nb = 300, 64
    a0 = oracle.set_random_hpd(n, nb, np.float64)
    m = dlaf.DeviceMatrix(b, np.float64, "L", n, nb)
    m.upload(a0)
    c.free()
    assert m.factorize() == 0
    out = a0.copy(order="F")
    m.download(out)
    assert oracle.cholesky_residual("L", a0, out) <= n * np.finfo(np.float64).eps
    m.close()
    b.free()
