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
    """include/dlaf/eigensolver/internal/get_band_size.h:20-31 (eigensolver_min_band = 100)."""
    return lib().dlaf_mi355x_get_band_size(nb)


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
