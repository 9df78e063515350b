"""First stage of the Hermitian eigensolver: reduction to band and its back-transformation.

Mirrors dlaf::eigensolver::internal::reduction_to_band (include/dlaf/eigensolver/reduction_to_band.h:40-122) and
bt_reduction_to_band (include/dlaf/eigensolver/bt_reduction_to_band.h) over the C ABI
(dlaf_mi355x_reduction_to_band_*, dlaf_mi355x_bt_reduction_to_band_*).  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .capi import DLAFDescriptor, lib, type_char
from .cholesky import DeviceMatrix, GeneralDeviceMatrix, Grid, _ld_of, _ptr, make_descriptor


def get_band_size(nb: int) -> int:
    """include/dlaf/eigensolver/internal/get_band_size.h:20-31 with the tune parameter eigensolver_min_band."""
    return lib().dlaf_mi355x_get_band_size(nb)


def red2band_panel_stats():
    """(blocked, fallback): panels of this process's last reduction_to_band factored by the blocked path / handed back to
    the reflector-by-reflector kernel."""
    a, b = C.c_long(0), C.c_long(0)
    lib().dlaf_mi355x_red2band_panel_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def eigensolver_min_band(b_min: int | None = None) -> int:
    """getTuneParameters().eigensolver_min_band (include/dlaf/tune.h:128; DLAF_EIGENSOLVER_MIN_BAND, src/init.cpp:220):
    returns the current value, after setting it when `b_min` is given."""
    if b_min is not None:
        lib().dlaf_mi355x_set_eigensolver_min_band(int(b_min))
    return lib().dlaf_mi355x_get_eigensolver_min_band()


def reduction_to_band(grid: Grid, a: np.ndarray, nb: int, band_size: int, isrc: int = 0, jsrc: int = 0,
                      n: int | None = None) -> np.ndarray:
    """reduction_to_band(grid, mat_a, band_size): `a` (this process's local part; lower triangle referenced) is
    overwritten with the band and the Householder reflectors below it; returns taus (n - band_size - 1 values, all
    of them on every process)."""
    t = type_char(a.dtype)
    if n is None:
        if grid.nranks != 1:
            raise ValueError("the global size n is required on a distributed grid")
        n = a.shape[0]
    taus = np.zeros(max(0, n - band_size - 1), dtype=a.dtype)
    da = make_descriptor(n, nb, _ld_of(a), isrc, jsrc)
    r = getattr(lib(), f"dlaf_mi355x_reduction_to_band_{t}")(grid.context, _ptr(a), da, band_size, _ptr(taus))
    if r != 0:
        raise RuntimeError(f"dlaf_mi355x_reduction_to_band_{t} returned {r}")
    return taus


def bt_reduction_to_band(grid: Grid, band_size: int, c: np.ndarray, v: np.ndarray, taus: np.ndarray, nb: int, isrc: int = 0,
                         jsrc: int = 0, c_jsrc: int = 0, n: int | None = None, k: int | None = None) -> None:
    """bt_reduction_to_band(grid, band_size, mat_c, mat_v, taus): c (local part of the n x k matrix) <- Q c."""
    t = type_char(c.dtype)
    if n is None or k is None:
        if grid.nranks != 1:
            raise ValueError("global sizes n, k are required on a distributed grid")
        n, k = v.shape[0], c.shape[1]
    dc = DLAFDescriptor(n, k, nb, nb, isrc, c_jsrc, 0, 0, _ld_of(c))
    dv = make_descriptor(n, nb, _ld_of(v), isrc, jsrc)
    r = getattr(lib(), f"dlaf_mi355x_bt_reduction_to_band_{t}")(grid.context, band_size, _ptr(c), dc, _ptr(v), dv, _ptr(taus))
    if r != 0:
        raise RuntimeError(f"dlaf_mi355x_bt_reduction_to_band_{t} returned {r}")


def reduction_to_band_device(a: DeviceMatrix, band_size: int) -> np.ndarray:
    """The same on a resident matrix (uplo 'L'); returns taus."""
    taus = np.zeros(max(0, a.n - band_size - 1), dtype=a.dtype)
    r = lib().dlaf_mi355x_reduction_to_band_device(a._h, band_size, _ptr(taus))
    if r != 0:
        raise RuntimeError(f"dlaf_mi355x_reduction_to_band_device returned {r}")
    return taus


def bt_reduction_to_band_device(band_size: int, c: GeneralDeviceMatrix, v: DeviceMatrix, taus: np.ndarray) -> None:
    r = lib().dlaf_mi355x_bt_reduction_to_band_device(band_size, c._h, v._h, _ptr(taus))
    if r != 0:
        raise RuntimeError(f"dlaf_mi355x_bt_reduction_to_band_device returned {r}")


def red2band_profile():
    """(device ms, whole-grid algorithmic flops) of the last reduction_to_band / bt_reduction_to_band here."""
    ms, fl = C.c_double(0), C.c_double(0)
    lib().dlaf_mi355x_red2band_profile(C.byref(ms), C.byref(fl))
    return ms.value, fl.value


# ---- the stages behind reduction_to_band and the eigensolver drivers --------------------------------------------
def _real_dtype(dtype):
    return np.zeros(0, dtype=dtype).real.dtype


def band_to_tridiagonal(grid: Grid, a: np.ndarray, nb: int, band_size: int, isrc: int = 0, jsrc: int = 0,
                        n: int | None = None):
    """band_to_tridiagonal<Backend::MC>(grid, Lower, band_size, mat_a) (include/dlaf/eigensolver/band_to_tridiag.h:74-176):
    `a` holds a Hermitian band matrix in the lower band of its local part.  Returns (d, e, v): the diagonal (n), the
    off-diagonal (n - 1, real) and the n x n matrix of compact Householder reflectors (tau in the place of the leading
    1, band_to_tridiag.h:40-72) -- complete on every process."""
    t = type_char(a.dtype)
    if n is None:
        if grid.nranks != 1:
            raise ValueError("the global size n is required on a distributed grid")
        n = a.shape[0]
    rt = _real_dtype(a.dtype)
    d = np.zeros(n, dtype=rt)
    e = np.zeros(n, dtype=rt)
    v = np.zeros((n, n), dtype=a.dtype, order="F")
    da = make_descriptor(n, nb, _ld_of(a), isrc, jsrc)
    r = getattr(lib(), f"dlaf_mi355x_band_to_tridiagonal_{t}")(grid.context, _ptr(a), da, band_size, _ptr(d), _ptr(e),
                                                               _ptr(v), max(1, n))
    if r != 0:
        raise RuntimeError(f"dlaf_mi355x_band_to_tridiagonal_{t} returned {r}")
    return d, e[:max(n - 1, 0)].copy(), v


def bt_band_to_tridiagonal(band_size: int, e: np.ndarray, v: np.ndarray) -> None:
    """bt_band_to_tridiagonal(band_size, mat_e, mat_hh) (include/dlaf/eigensolver/bt_band_to_tridiag.h:28-61), local:
    e (n x k, column-major) <- Q e."""
    t = type_char(e.dtype)
    n, k = e.shape
    r = getattr(lib(), f"dlaf_mi355x_bt_band_to_tridiagonal_{t}")(band_size, n, k, _ptr(v), _ld_of(v), _ptr(e), _ld_of(e))
    if r != 0:
        raise RuntimeError(f"dlaf_mi355x_bt_band_to_tridiagonal_{t} returned {r}")


def tridiagonal_eigensolver(d: np.ndarray, e: np.ndarray, nb: int = 512):
    """tridiagonal_eigensolver(tridiag, evals, evecs) (include/dlaf/eigensolver/tridiag_solver.h:30-60), local:
    returns (w ascending, z column-major n x n)."""
    t = type_char(d.dtype)
    n = d.shape[0]
    d = np.ascontiguousarray(d)
    e = np.ascontiguousarray(e, dtype=d.dtype)
    w = np.zeros(n, dtype=d.dtype)
    z = np.zeros((n, n), dtype=d.dtype, order="F")
    r = getattr(lib(), f"dlaf_mi355x_tridiagonal_eigensolver_{t}")(n, nb, _ptr(d), _ptr(e), _ptr(w), _ptr(z), max(1, n))
    if r != 0:
        raise RuntimeError(f"dlaf_mi355x_tridiagonal_eigensolver_{t} returned {r}")
    return w, z


def _eig_name(t: str) -> str:
    return "symmetric" if t in "sd" else "hermitian"


def hermitian_eigensolver(grid: Grid, uplo: str, a: np.ndarray, nb: int, isrc: int = 0, jsrc: int = 0,
                          n: int | None = None, z_jsrc: int | None = None, z_shape=None):
    """dlaf::hermitian_eigensolver(grid, uplo, mat_a, evals, evecs) through the reference's C entry
    dlaf_{symmetric,hermitian}_eigensolver_* (include/dlaf_c/eigensolver/eigensolver.h:39-58).  `a` (local part, the uplo
    triangle referenced) is destroyed.  Returns (w, z): all eigenvalues (ascending) and the local part of the
    eigenvector matrix."""
    t = type_char(a.dtype)
    if n is None:
        if grid.nranks != 1:
            raise ValueError("the global size n is required on a distributed grid")
        n = a.shape[0]
    if z_jsrc is None:
        z_jsrc = jsrc
    w = np.zeros(n, dtype=_real_dtype(a.dtype))
    z = np.zeros(z_shape if z_shape is not None else a.shape, dtype=a.dtype, order="F")
    da = make_descriptor(n, nb, _ld_of(a), isrc, jsrc)
    dz = make_descriptor(n, nb, _ld_of(z), isrc, z_jsrc)
    r = getattr(lib(), f"dlaf_{_eig_name(t)}_eigensolver_{t}")(grid.context, uplo.encode()[0:1], _ptr(a), da, _ptr(w),
                                                              _ptr(z), dz)
    if r != 0:
        raise RuntimeError(f"dlaf_{_eig_name(t)}_eigensolver_{t} returned {r}")
    return w, z


def hermitian_generalized_eigensolver(grid: Grid, uplo: str, a: np.ndarray, b: np.ndarray, nb: int, isrc: int = 0,
                                      jsrc: int = 0, n: int | None = None, factorized: bool = False):
    """dlaf::hermitian_generalized_eigensolver(grid, uplo, mat_a, mat_b, evals, evecs) through the reference's C entry
    (include/dlaf_c/eigensolver/gen_eigensolver.h:44-135).  a is destroyed, b ends up holding its Cholesky factor."""
    t = type_char(a.dtype)
    if n is None:
        if grid.nranks != 1:
            raise ValueError("the global size n is required on a distributed grid")
        n = a.shape[0]
    w = np.zeros(n, dtype=_real_dtype(a.dtype))
    z = np.zeros(a.shape, dtype=a.dtype, order="F")
    da = make_descriptor(n, nb, _ld_of(a), isrc, jsrc)
    db = make_descriptor(n, nb, _ld_of(b), isrc, jsrc)
    dz = make_descriptor(n, nb, _ld_of(z), isrc, jsrc)
    name = f"dlaf_{_eig_name(t)}_generalized_eigensolver{'_factorized' if factorized else ''}_{t}"
    r = getattr(lib(), name)(grid.context, uplo.encode()[0:1], _ptr(a), da, _ptr(b), db, _ptr(w), _ptr(z), dz)
    if r != 0:
        raise RuntimeError(f"{name} returned {r}")
    return w, z


def eigensolver_profile():
    """device ms per stage of the last eigensolver call: reduction_to_band, band_to_tridiagonal, tridiagonal_eigensolver,
    bt_band_to_tridiagonal, bt_reduction_to_band."""
    ms = (C.c_double * 5)()
    lib().dlaf_mi355x_eigensolver_profile(ms)
    return list(ms)
