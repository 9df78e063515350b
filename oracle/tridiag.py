"""tridiag.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's eigensolver stages behind reduction_to_band:
band -> tridiagonal, the tridiagonal eigensolver's checks, and the back-transformation band <- tridiagonal.
May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; the product never imports it.

What it follows (paths relative to /root/reference):
  * BandToTridiag::call_L, local            include/dlaf/eigensolver/band_to_tridiag/mc.h:681-867
      - HH_reflector (xLARFG)               :55-68
      - apply_HH_left_right_herm            :70-86   (xHEMV, w += -1/2 tau (w^H v) v, xHER2)
      - apply_HH_left / apply_HH_right      :88-118
      - SweepWorker::start_sweep / do_step  :503-531
      - nrSweeps / nrStepsForSweep          include/dlaf/eigensolver/band_to_tridiag/api.h:24-34
      - layout of the compact reflectors    include/dlaf/eigensolver/band_to_tridiag.h:40-72, mc.h:762-768
  * the checker of the reference's own test test/unit/eigensolver/test_band_to_tridiag.cpp:60-118
  * bt_band_to_tridiagonal (definition)     include/dlaf/eigensolver/bt_band_to_tridiag.h:28-61: E <- Q E with
                                            Q = HHT(0,0) HHT(0,1) ... HHT(1,0) ... (band_to_tridiag.h:49-53)
  * the checkers of test_tridiag_solver_local.cpp:62-129 (1D Laplacian, closed form) and
    test/include/dlaf_test/eigensolver/test_eigensolver_correctness.h:37-101 (orthogonality, A E = E Lambda)

The arithmetic of HH_reflector is LAPACK's xLARFG (a third-party dependency of the reference, lapackpp >= 2022.05
-> the system LAPACK), restated from its published algorithm without the rescaling loop for subnormal norms.

Pinned by: the reference test's reconstruction property (applying the stored reflectors to the tridiagonal matrix
gives back the band matrix, tolerance of test_band_to_tridiag.cpp:117), LAPACK ?sbtrd-independent spectrum
preservation (eigvalsh(tridiagonal) == eigvalsh(band)), scipy's eigh_tridiagonal and the closed-form 1D Laplacian.
"""
from __future__ import annotations

import numpy as np

from .red2band import error_of


def is_complex(dtype) -> bool:
    return np.dtype(dtype).kind == "c"


def nr_sweeps(n: int, dtype) -> int:
    """api.h:24-28."""
    return n - 1 if is_complex(dtype) else n - 2


def nr_steps_for_sweep(sweep: int, n: int, band: int) -> int:
    """api.h:30-34."""
    return 1 if sweep == n - 2 else -((n - sweep - 2) // -band)


def larfg(x: np.ndarray):
    """xLARFG on x (in place): x[0] <- beta, x[1:] <- v[1:]; returns tau."""
    n = x.shape[0]
    cx = is_complex(x.dtype)
    if n <= 0:
        return x.dtype.type(0)
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:]) if n > 1 else 0.0
    if xnorm == 0 and (not cx or alpha.imag == 0):
        return x.dtype.type(0)
    if cx:
        beta = -np.copysign(np.sqrt(alpha.real ** 2 + alpha.imag ** 2 + xnorm ** 2), alpha.real)
        tau = complex((beta - alpha.real) / beta, -alpha.imag / beta)
    else:
        beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
        tau = (beta - alpha) / beta
    x[1:] *= 1 / (alpha - beta)
    x[0] = beta
    return x.dtype.type(tau)


def band_to_tridiag(a: np.ndarray, band: int):
    """BandToTridiag::call_L on the dense Hermitian band matrix a (lower triangle referenced, bandwidth `band`).
    Returns (d, e, v): diagonal, off-diagonal (length n - 1, real) and the n x n matrix of compact reflectors
    (tau in the place of the leading 1), laid out as band_to_tridiag.h:56-63 says."""
    n = a.shape[0]
    dt = a.dtype
    b = band
    # full Hermitian working copy (the reference works on the lower band only; same arithmetic per element)
    w = np.tril(a).astype(dt)
    w = w + np.tril(w, -1).conj().T
    v_out = np.zeros((n, n), dtype=dt)
    if n == 0:
        return np.zeros(0, dtype=w.real.dtype), np.zeros(0, dtype=w.real.dtype), v_out
    for sweep in range(max(0, nr_sweeps(n, dt))):
        # start_sweep (mc.h:503-508): reflector of column `sweep`, rows sweep+1 ...
        nn = min(n - sweep - 1, b)
        x = w[sweep + 1:sweep + 1 + nn, sweep].copy()
        tau = larfg(x)
        v = x.copy()
        v[0] = 1
        w[sweep + 1, sweep] = x[0]
        w[sweep + 2:sweep + 1 + nn, sweep] = 0
        w[sweep, sweep + 1:sweep + 1 + nn] = w[sweep + 1:sweep + 1 + nn, sweep].conj()
        for step in range(nr_steps_for_sweep(sweep, n, b)):
            j = 1 + sweep + step * b
            nh = min(b, n - j)
            # compact_copy_to_tile (mc.h:492-497, :766-768)
            pos = (sweep // b + step) * b
            v_out[pos, sweep] = tau
            v_out[pos + 1:pos + nh, sweep] = v[1:nh]
            m = min(b, n - b - j)
            # apply_HH_left_right_herm on the nh x nh diagonal block
            d = w[j:j + nh, j:j + nh]
            ww = tau * (d @ v[:nh])
            ww = ww + (-np.vdot(ww, v[:nh]) * tau / 2) * v[:nh]
            d -= np.outer(ww, v[:nh].conj()) + np.outer(v[:nh], ww.conj())
            if m > 0:
                # apply_HH_right on the m x nh block below
                blk = w[j + nh:j + nh + m, j:j + nh]
                wr = blk @ v[:nh]
                blk -= tau * np.outer(wr, v[:nh].conj())
                w[j:j + nh, j + nh:j + nh + m] = blk.conj().T
            if m > 1:
                x = w[j + nh:j + nh + m, j].copy()
                tau = larfg(x)
                v = x.copy()
                v[0] = 1
                w[j + nh, j] = x[0]
                w[j + nh + 1:j + nh + m, j] = 0
                blk = w[j + nh:j + nh + m, j + 1:j + nh]
                wl = blk.conj().T @ v[:m]
                blk -= np.conj(tau) * np.outer(v[:m], wl.conj())
                w[j:j + nh, j + nh:j + nh + m] = w[j + nh:j + nh + m, j:j + nh].conj().T
    d = np.real(np.diag(w)).copy()
    e = np.real(np.diag(w, -1)).copy()
    return d, e, v_out


def reflector_list(n: int, band: int, dtype):
    """(sweep, step, first_row, size, pos) of every reflector in the order of Q = HHT(0,0) HHT(0,1) ...
    (band_to_tridiag.h:49-63)."""
    out = []
    for sweep in range(max(0, nr_sweeps(n, dtype))):
        for step in range(nr_steps_for_sweep(sweep, n, band)):
            first = 1 + sweep + step * band
            size = min(band, n - first)
            out.append((sweep, step, first, size, (sweep // band + step) * band))
    return out


def apply_q(v: np.ndarray, band: int, e: np.ndarray, adjoint: bool = False) -> np.ndarray:
    """E <- Q E (bt_band_to_tridiagonal) or Q^H E, one reflector at a time -- the definition."""
    n = v.shape[0]
    e = e.copy()
    refl = reflector_list(n, band, v.dtype)
    order = refl if adjoint else reversed(refl)
    for sweep, step, first, size, pos in order:
        vec = v[pos:pos + size, sweep].copy()
        tau = vec[0]
        vec[0] = 1
        if adjoint:
            tau = np.conj(tau)
        rows = e[first:first + size]
        rows -= tau * np.outer(vec, vec.conj() @ rows)
    return e


def check_band_to_tridiag(a: np.ndarray, band: int, d: np.ndarray, e: np.ndarray, v: np.ndarray):
    """test_band_to_tridiag.cpp:60-118: rebuild the band matrix from the tridiagonal one and the stored reflectors,
    compare the lower band with the input.  Returns (ok, max abs diff, bar)."""
    n = a.shape[0]
    dt = a.dtype
    t = np.zeros((n, n), dtype=dt)
    t[np.arange(n), np.arange(n)] = d
    if n > 1:
        t[np.arange(1, n), np.arange(n - 1)] = e[:n - 1]
        t[np.arange(n - 1), np.arange(1, n)] = e[:n - 1]
    # A = Q T Q^H
    q_t = apply_q(v, band, t)
    full = apply_q(v, band, q_t.conj().T).conj().T
    mask = np.tril(np.ones((n, n), dtype=bool)) & ~np.tril(np.ones((n, n), dtype=bool), -(band + 1))
    want = np.where(mask, a, 0)
    got = np.where(mask, full, 0)
    nb_like = max(band, 1)
    err = error_of(dt)
    diff = np.abs(got - want)
    # CHECK_MATRIX_NEAR(res, mat_a_h, mb * m * error, m * error): relative or absolute
    rel_ok = diff <= nb_like * n * err * np.maximum(np.abs(want), np.finfo(want.real.dtype).tiny)
    abs_ok = diff <= max(n, 1) * err * max(1.0, float(np.abs(want).max(initial=0)))
    return bool(np.all(rel_ok | abs_ok)), float(diff.max(initial=0)), float(n * err)


def laplace_1d(n: int, dtype=np.float64):
    """test_tridiag_solver_local.cpp:62-129: (d, e, eigenvalues, eigenvectors) of the 1D Laplacian."""
    d = np.full(n, 2, dtype=dtype)
    e = np.full(max(n - 1, 0), -1, dtype=dtype)
    i = np.arange(1, n + 1)
    evals = 2 * (1 - np.cos(np.pi * i / (n + 1)))
    evecs = np.sqrt(2.0 / (n + 1)) * np.sin(np.outer(i, i) * np.pi / (n + 1))
    return d, e, evals.astype(dtype), evecs.astype(dtype)


def check_eigensolver(a_full: np.ndarray, evals: np.ndarray, evecs: np.ndarray):
    """test_eigensolver_correctness.h:37-101 on the full Hermitian matrix a_full: eigenvalues sorted, E^H E == I
    (m * error relative / 10 m error absolute), A E == E Lambda (2 m error).  Returns a dict of the three findings."""
    m = a_full.shape[0]
    err = error_of(evecs.dtype)
    srt = bool(np.all(np.diff(evals) >= 0))
    g = evecs.conj().T @ evecs
    orth = float(np.abs(g - np.eye(m)).max(initial=0))
    ae = a_full @ evecs
    el = evecs * evals[None, :]
    diff = np.abs(ae - el)
    tol = 2 * m * err
    res_ok = bool(np.all((diff <= tol) | (diff <= tol * np.abs(el))))
    return {"sorted": srt, "orth": orth, "orth_bar": 10 * m * err, "residual": float(diff.max(initial=0)),
            "residual_bar": tol, "residual_ok": res_ok}
