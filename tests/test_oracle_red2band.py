"""CPU tests pinning oracle/red2band.py (the restatement of ReductionToBand::call and of the back-transformation) before
anything is compared against it: the reference's own property checker (test_reduction_to_band.cpp:270-310) over the
reference's size lists (:83-110), LAPACK ?geqrf on a panel (same reflectors, taus and R: impl.h:106-140 is xLARFG), the
compact WY identity for the T factor (t_factor_impl.h:60-131 is xLARFT forward / columnwise), LAPACK ?ormqr for the
back-transformation, spectrum preservation, and the band-size rule (get_band_size.h:20-31)."""
import numpy as np
import pytest

from oracle import red2band as rb

TYPES = [np.float64, np.complex128, np.float32, np.complex64]
CONFIGS = [(0, 3, 3), (3, 3, 3), (12, 3, 3), (13, 3, 3), (24, 3, 3), (40, 5, 5)]
CONFIGS_SUBBAND = [(0, 6, 2), (4, 4, 2), (12, 4, 2), (42, 6, 3), (13, 6, 3), (27, 9, 3), (42, 12, 4), (29, 9, 3)]


@pytest.mark.parametrize("dt", TYPES)
def test_oracle_passes_the_references_checker(dt):
    for n, nb, band in CONFIGS + CONFIGS_SUBBAND + [(130, 32, 16)]:
        for banded in (None, band - 1):
            a0 = rb.random_hermitian(n, dt, seed=n + nb, banded=banded)
            a = a0.copy(order="F")
            up = np.triu_indices(n, 1)
            a[up] = 7.25
            taus = rb.reduction_to_band(a, nb, band)
            assert len(taus) == rb.nr_reflectors(n, band)
            assert (a[up] == 7.25).all()                      # checkUpperPartUnchanged (:252-268)
            ok, diff, tol = rb.check_result(a0, a, taus, band)
            assert ok, (dt, n, nb, band, diff, tol)
            if n:
                ev0 = np.linalg.eigvalsh(a0.astype(np.complex128))
                ev1 = np.linalg.eigvalsh(rb.split_band(a, band).astype(np.complex128))
                assert np.abs(ev0 - ev1).max() <= 2 * tol


@pytest.mark.parametrize("dt", TYPES)
def test_panel_reflectors_are_lapack_geqrf(dt):
    from scipy.linalg import lapack
    rng = np.random.default_rng(3)
    for m, b in [(17, 5), (40, 8), (9, 9), (6, 4)]:
        p0 = rng.uniform(-1, 1, (m, b))
        if np.dtype(dt).kind == "c":
            p0 = p0 + 1j * rng.uniform(-1, 1, (m, b))
        p0 = np.asfortranarray(p0.astype(dt))
        p = p0.copy(order="F")
        nr = min(b, m - 1)
        taus = rb.compute_panel_reflectors(p, nr)
        geqrf = getattr(lapack, {"f": "sgeqrf", "d": "dgeqrf", "F": "cgeqrf", "D": "zgeqrf"}[np.dtype(dt).char])
        qr, tau, _, info = geqrf(p0.copy(order="F"))
        assert info == 0
        tol = 50 * m * rb.error_of(dt)
        assert np.abs(taus - tau[:nr]).max() <= tol
        # R and the reflectors, column by column up to the last reflector the reference computes
        assert np.abs(np.triu(p)[:nr, :] - np.triu(qr)[:nr, :]).max() <= tol
        assert np.abs(np.tril(p, -1)[:, :nr] - np.tril(qr, -1)[:, :nr]).max() <= tol


@pytest.mark.parametrize("dt", TYPES)
def test_t_factor_is_the_compact_wy_form(dt):
    rng = np.random.default_rng(4)
    m, k = 30, 7
    p = rng.uniform(-1, 1, (m, k))
    if np.dtype(dt).kind == "c":
        p = p + 1j * rng.uniform(-1, 1, (m, k))
    p = np.asfortranarray(p.astype(dt))
    taus = rb.compute_panel_reflectors(p, k)
    v = rb.well_formed_v(p, k)
    t = rb.compute_t_factor(v, taus)
    assert np.array_equal(np.tril(t, -1), np.zeros_like(t))
    q = np.eye(m, dtype=dt)
    for j in range(k):                                # H_0 H_1 ... H_{k-1}
        q = q @ (np.eye(m, dtype=dt) - taus[j] * np.outer(v[:, j], v[:, j].conj()))
    assert np.abs(q - (np.eye(m, dtype=dt) - v @ t @ v.conj().T)).max() <= 50 * m * rb.error_of(dt)


@pytest.mark.parametrize("dt", TYPES)
def test_back_transformation_is_q_times_c(dt):
    rng = np.random.default_rng(5)
    for n, nb, band, k in [(13, 3, 3, 5), (42, 12, 4, 17), (29, 9, 3, 29), (130, 32, 16, 40)]:
        a = rb.random_hermitian(n, dt, seed=n)
        taus = rb.reduction_to_band(a, nb, band)
        c0 = rng.uniform(-1, 1, (n, k))
        if np.dtype(dt).kind == "c":
            c0 = c0 + 1j * rng.uniform(-1, 1, (n, k))
        c0 = np.asfortranarray(c0.astype(dt))
        c = c0.copy(order="F")
        rb.bt_reduction_to_band(c, a, taus, nb, band)
        ref = rb.apply_q(a, taus, band, c0, "L", False)    # LAPACK ?ormqr / ?unmqr with the same reflectors
        assert np.abs(c - ref).max() <= n * k * rb.error_of(dt)


def test_band_size_rule():
    assert [rb.get_band_size(nb) for nb in (512, 1024, 256, 64, 100, 200, 300, 99)] == [128, 128, 128, 64, 100, 100, 100, 99]


@pytest.mark.parametrize("dt", [np.float64, np.complex128])
def test_blocked_panel_factorization_reproduces_geqr2(dt):
    """CholeskyQR2 + Householder reconstruction (oracle.red2band.panel_reflectors_blocked: the numpy restatement of what
    csrc/device/kernels_hr.hip does on the GPU) gives xGEQR2's reflectors, taus, R -- LAPACK's sign convention included --
    and the T factor of xLARFT, element by element: against this oracle's reflector-by-reflector restatement
    (impl.h:297-361) and against LAPACK ?geqrf itself."""
    import scipy.linalg as sl
    rng = np.random.default_rng(3)
    for m, b in [(300, 16), (500, 128), (2000, 128), (260, 64), (130, 128)]:
        p0 = rng.uniform(-1, 1, (m, b)).astype(dt)
        if np.dtype(dt).kind == "c":
            p0 = p0 + 1j * rng.uniform(-1, 1, (m, b))
        out, taus, t = rb.panel_reflectors_blocked(p0.copy())
        ref = np.asfortranarray(p0.copy())
        rtaus = rb.compute_panel_reflectors(ref, b)
        tol = 50 * m * rb.error_of(dt)
        assert np.abs(out - ref).max() <= tol and np.abs(taus - rtaus).max() <= tol, (m, b, np.abs(out - ref).max())
        qr, tl = (sl.lapack.zgeqrf if np.dtype(dt).kind == "c" else sl.lapack.dgeqrf)(p0)[:2]
        assert np.abs(out - qr).max() <= tol and np.abs(taus - tl).max() <= tol
        v = rb.well_formed_v(out, b)
        assert np.abs(t - rb.compute_t_factor(v, taus)).max() <= tol
