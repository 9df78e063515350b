"""GPU parity tests of reduction_to_band / bt_reduction_to_band (first stage of the eigensolver)
through the C ABI, against the reference's own checker restated in oracle/red2band.py
(test/unit/eigensolver/test_reduction_to_band.cpp:270-310: Q B Q^H == A within n^2 * error; :252-268: the upper
triangle untouched) over the reference's size lists (:83-110), and elementwise against the oracle's restatement of
ReductionToBand::call.  Reflector signs / taus follow xLARFG exactly as the reference does, so band, reflectors
and taus are comparable element by element (tolerance: the checker's n^2 * error, absolute)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPES = ["d", "z", "s", "c"]
# test_reduction_to_band.cpp:83-90 (full-tile band) and :92-110 (sub-band): (n, nb, band)
CONFIGS = [(0, 3, 3), (3, 3, 3), (12, 3, 3), (13, 3, 3), (24, 3, 3), (40, 5, 5)]
CONFIGS_SUBBAND = [(0, 6, 2), (4, 4, 2), (12, 4, 2), (42, 6, 3), (13, 6, 3), (27, 9, 3), (42, 12, 4), (29, 9, 3)]


@pytest.fixture(scope="module")
def dlaf():
    import dla_future_amd as d
    d.initialize()
    return d


@pytest.fixture(scope="module")
def grid(dlaf):
    return dlaf.Grid.single()


@pytest.fixture(scope="module")
def rb():
    from oracle import red2band
    return red2band


def run_and_check(dlaf, grid, rb, oracle, t, n, nb, band, banded, elementwise=True):
    dt = oracle.DTYPES[t]
    a0 = rb.random_hermitian(n, dt, seed=1000 + n + nb, banded=(band - 1) if banded else None)
    store = np.full((max(1, n) + 2, max(1, n)), 5.5, dtype=dt, order="F")
    a = store[:n, :n]
    a[...] = a0
    up = np.triu_indices(n, 1)
    a[up] = -9.9                                   # the strict upper triangle must be neither read nor written
    taus = dlaf.reduction_to_band(grid, a, nb, band)
    assert taus.shape == (max(0, n - band - 1),)
    assert (a[up] == dt(-9.9)).all(), "upper triangle changed"
    assert (store[n:, :] == 5.5).all()
    ok, diff, tol = rb.check_result(a0, a, taus, band)      # the reference's checkResult
    if not ok:                                              # say where it went wrong (first bad reflector / column)
        ref = a0.copy(order="F")
        rtaus = rb.reduction_to_band(ref, nb, band)
        dm = np.abs(np.tril(ref) - np.tril(a))
        raise AssertionError((t, n, nb, band, banded, diff, tol, "bad taus", np.nonzero(np.abs(rtaus - taus) > tol)[0][:4],
                              "bad columns", np.nonzero(dm.max(axis=0) > tol)[0][:6], "bad rows",
                              np.nonzero(dm.max(axis=1) > tol)[0][:6]))
    if elementwise and n:
        ref = a0.copy(order="F")
        rtaus = rb.reduction_to_band(ref, nb, band)
        assert np.abs(np.tril(ref) - np.tril(a)).max() <= tol, (t, n, nb, band, np.abs(np.tril(ref) - np.tril(a)).max(), tol)
        if len(taus):
            assert np.abs(rtaus - taus).max() <= tol
    return a0, a, taus


@pytest.mark.parametrize("t", TYPES)
def test_reduction_to_band_reference_configs(dlaf, grid, rb, oracle, t):
    for n, nb, band in CONFIGS + CONFIGS_SUBBAND:
        for banded in (False, True):
            run_and_check(dlaf, grid, rb, oracle, t, n, nb, band, banded)


@pytest.mark.parametrize("t,n,nb,band", [("d", 300, 64, 32), ("d", 515, 128, 64), ("z", 260, 64, 32), ("s", 300, 64, 16),
                                         ("c", 200, 64, 64), ("d", 1100, 256, 128), ("z", 700, 256, 128),
                                         ("d", 2048, 512, 128), ("z", 1024, 512, 128)])
def test_reduction_to_band_fast_path_sizes(dlaf, grid, rb, oracle, t, n, nb, band):
    """block sizes / bands that take the vectorised operand paths (multiples of 16), ragged last tiles, and the
    BASELINE configuration-5 shape nb = 512, band = get_band_size(512) = 128 at a size the oracle finishes in seconds"""
    a0, a, taus = run_and_check(dlaf, grid, rb, oracle, t, n, nb, band, False)
    # spectrum of the band == spectrum of the input (what the stage is for)
    dt = oracle.DTYPES[t]
    b = rb.split_band(a, band)
    ev0 = np.linalg.eigvalsh(a0.astype(np.complex128))
    ev1 = np.linalg.eigvalsh(b.astype(np.complex128))
    assert np.abs(ev0 - ev1).max() <= n * n * rb.error_of(dt)


def test_reduction_to_band_blocked_panels_and_their_gate(dlaf, grid, rb, oracle):
    """Double-precision panels (real and complex) of band 64 / 128 take the blocked factorization (CholeskyQR2 + Householder reconstruction,
    csrc/device/kernels_hr.hip) and must give xGEQR2's reflectors, taus and R element by element (run_and_check compares
    with the oracle's reflector-by-reflector restatement); panels whose Gram matrix is singular or badly conditioned are
    handed back to the reflector-by-reflector kernel by the gate -- same checks."""
    for t, n, nb, band in [("d", 2048, 512, 128), ("d", 1100, 256, 128), ("d", 700, 128, 64), ("z", 1100, 256, 128),
                           ("z", 700, 128, 64)]:
        run_and_check(dlaf, grid, rb, oracle, t, n, nb, band, False)
        blocked, fallback = dlaf.red2band_panel_stats()
        assert blocked > 0 and fallback == 0, (t, n, nb, band, blocked, fallback)
    # the identity matrix: every panel is zero below the band (test_eigensolver.cpp:72-76 runs it end to end)
    n, nb, band = 1024, 256, 128
    a = np.asfortranarray(np.eye(n))
    taus = dlaf.reduction_to_band(grid, a, nb, band)
    blocked, fallback = dlaf.red2band_panel_stats()
    assert blocked == 0 and fallback > 0, (blocked, fallback)
    assert np.array_equal(a, np.eye(n)) and not taus.any()
    # two columns of every panel agree to 1e-9: cond ~ 1e9, far beyond what CholeskyQR2 may be trusted with
    dt = np.float64
    a0 = rb.random_hermitian(n, dt, seed=77)
    for c in range(0, n - 1, band):
        a0[:, c + 1] = a0[:, c] * (1 + 1e-9)
        a0[c + 1, :] = a0[:, c + 1]
    a0 = np.asfortranarray((a0 + a0.T) / 2)
    a = a0.copy(order="F")
    taus = dlaf.reduction_to_band(grid, a, nb, band)
    blocked, fallback = dlaf.red2band_panel_stats()
    assert fallback > 0, (blocked, fallback)
    ok, diff, tol = rb.check_result(a0, a, taus, band)
    assert ok, (diff, tol)


def test_band_size_rule(dlaf):
    # get_band_size.h:20-31
    assert [dlaf.get_band_size(nb) for nb in (512, 1024, 256, 64, 100, 200, 300)] == [128, 128, 128, 64, 100, 100, 100]


@pytest.mark.parametrize("t", TYPES)
def test_bt_reduction_to_band_reference_configs(dlaf, grid, rb, oracle, t):
    """test_bt_reduction_to_band.cpp: C <- Q C against the oracle's restatement, k columns (ragged), the reflectors
    produced by the oracle so that the two stages are tested independently."""
    dt = oracle.DTYPES[t]
    rng = np.random.default_rng(7)
    for n, nb, band, k in [(0, 3, 3, 2), (3, 3, 3, 3), (12, 3, 3, 7), (13, 3, 3, 13), (42, 6, 3, 11), (29, 9, 3, 20),
                           (42, 12, 4, 42), (300, 64, 32, 130), (515, 128, 64, 260), (1100, 512, 128, 700)]:
        v = rb.random_hermitian(n, dt, seed=n)
        taus = rb.reduction_to_band(v, nb, band)
        c0 = rng.uniform(-1, 1, (n, k))
        if dt in (np.complex64, np.complex128):
            c0 = c0 + 1j * rng.uniform(-1, 1, (n, k))
        c0 = np.asfortranarray(c0.astype(dt))
        ref = c0.copy(order="F")
        rb.bt_reduction_to_band(ref, v, taus, nb, band)
        got = c0.copy(order="F")
        v_in = v.copy(order="F")
        dlaf.bt_reduction_to_band(grid, band, got, v_in, taus, nb)
        assert np.array_equal(v_in, v)
        tol = max(1, n) * max(1, k) * rb.error_of(dt)
        if got.size:
            assert np.abs(got - ref).max() <= tol, (t, n, nb, band, k, np.abs(got - ref).max(), tol)


@pytest.mark.parametrize("t", ["d", "z"])
def test_red2band_then_bt_recovers_the_eigenvectors(dlaf, grid, rb, oracle, t):
    """Two stages on resident operands: A -> band B (+ Q); eigenvectors Z of B from LAPACK; Q Z must diagonalise A."""
    dt = oracle.DTYPES[t]
    n, nb, band = 768, 256, 128
    a0 = rb.random_hermitian(n, dt, seed=5)
    A = dlaf.DeviceMatrix(grid, dt, "L", n, nb)
    A.upload(a0)
    taus = dlaf.reduction_to_band_device(A, band)
    out = a0.copy(order="F")
    A.download(out)
    b = rb.split_band(out, band)
    w, z = np.linalg.eigh(b)
    Cm = dlaf.GeneralDeviceMatrix(grid, dt, n, n, nb)
    zc = np.asfortranarray(z.astype(dt))
    Cm.upload(zc)
    dlaf.bt_reduction_to_band_device(band, Cm, A, taus)
    Cm.download(zc)
    resid = np.abs(a0 @ zc - zc * w[None, :]).max()
    assert resid <= n * n * rb.error_of(dt), resid
    assert np.abs(zc.conj().T @ zc - np.eye(n)).max() <= n * n * rb.error_of(dt)
    ms, flops = dlaf.red2band_profile()
    assert ms > 0 and flops > 0
