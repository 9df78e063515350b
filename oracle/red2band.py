"""red2band.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's reduction of a Hermitian matrix to band form and of the matching
back-transformation (first stage).  May be imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only; the product never imports it.

What it follows, line by line (paths relative to /root/reference):
  * ReductionToBand::call, local            include/dlaf/eigensolver/reduction_to_band/impl.h:968-1110
      - computePanelReflectors              :297-361  (computeX0AndSquares :77-103, computeReflectorAndTau :106-140,
                                                        computeWTrailingPanel :143-185, updateTrailingPanel :188-228)
      - setupReflectorPanelV                :364-422
      - computeTFactor                      include/dlaf/factorization/qr/t_factor_impl.h:60-131 (gemvColumnT, trmvUpdateColumn)
      - trmmComputeW (W = V T)              :425-445
      - hemmComputeX (X = At W)             :465-517
      - gemmComputeW2, gemmUpdateX          :520-542, :448-462
      - her2kUpdateTrailingMatrix           :545-585
  * BackTransformationReductionToBand::call include/dlaf/eigensolver/bt_reduction_to_band/impl.h:132-236
  * the checker of the reference's own test (checkResult, splitReflectorsAndBand, setupHermitianBand,
    checkUpperPartUnchanged)                test/unit/eigensolver/test_reduction_to_band.cpp:141-310
  * band size of a block size               include/dlaf/eigensolver/internal/get_band_size.h:20-31

The distributed algorithm (impl.h:1113-1462) computes the same quantities with its sums split over ranks;
this restatement is the local one, the distributed HIP runs are compared against it within the tolerance of
the reference's test.

Pinned by: the reference's own property test restated here (Q B Q^H == A with tolerance n^2 * error,
test_reduction_to_band.cpp:270-310; the Q application goes through LAPACK ?ormqr/?unmqr as there), LAPACK
?sytrd/?hetrd-independent spectrum preservation (eigvalsh(band) == eigvalsh(A)), and LAPACK ?geqrf on the first
panel (the reflectors of a panel are LAPACK's xGEQR2 reflectors -- same sign and tau conventions,
impl.h:106-140 == xLARFG).
"""
from __future__ import annotations

import numpy as np

ERR = {np.dtype(np.float32): 2 * np.finfo(np.float32).eps, np.dtype(np.float64): 2 * np.finfo(np.float64).eps,
       np.dtype(np.complex64): 8 * np.finfo(np.float32).eps, np.dtype(np.complex128): 8 * np.finfo(np.float64).eps}


def error_of(dtype) -> float:
    """TypeUtilities<T>::error (test/include/dlaf_test/util_types.h:40,62)."""
    return float(ERR[np.dtype(dtype)])


def get_band_size(nb: int, min_band: int = 100) -> int:
    """get_band_size.h:20-31 with the default eigensolver_min_band = 100 (tune.h): the smallest divisor of nb
    that is >= min_band (nb itself when there is none)."""
    for div in range(nb // min_band, 1, -1):
        if nb % div == 0:
            return nb // div
    return nb


def nr_reflectors(n: int, band: int) -> int:
    return max(0, n - band - 1)


# ------------------------------------------------------------------------------------------ panel
def compute_panel_reflectors(p: np.ndarray, nrefls: int) -> np.ndarray:
    """impl.h:297-361 on the m x b panel p (in place): for reflector j the head is p[j, j]; returns taus."""
    m, cols = p.shape
    dt = p.dtype
    taus = np.zeros(nrefls, dtype=dt)
    for j in range(nrefls):
        # computeX0AndSquares (:77-103)
        x0 = p[j, j]
        squares = np.vdot(p[j:, j], p[j:, j])
        # computeReflectorAndTau (:106-140)
        if squares == 0:
            tau = dt.type(0)
        else:
            norm = np.sqrt(squares)           # (T-typed sqrt of a real non-negative number)
            y = norm if np.signbit(np.real(x0)) else -norm
            tau = (y - x0) / y
            p[j, j] = y
            if j + 1 < m:
                p[j + 1:, j] *= dt.type(1) / (x0 - y)
        taus[j] = tau
        pt_cols = cols - (j + 1)
        if pt_cols == 0:
            break
        # computeWTrailingPanel (:143-185): w = Pt^H v with v = [1; p[j+1:, j]]
        w = np.conj(p[j, j + 1:])
        if j + 1 < m:
            w = w + p[j + 1:, j + 1:].conj().T @ p[j + 1:, j]
        # updateTrailingPanel (:188-228): Pt -= conj(tau) v w^H
        p[j, j + 1:] -= np.conj(tau) * np.conj(w)
        if j + 1 < m:
            p[j + 1:, j + 1:] -= np.conj(tau) * np.outer(p[j + 1:, j], np.conj(w))
    return taus


def panel_reflectors_blocked(p: np.ndarray):
    """The BLOCKED panel factorization of the MI355X build (csrc/device/kernels_hr.hip, red2band.cpp), restated in numpy
    so that it can be pinned on the CPU against compute_panel_reflectors / LAPACK ?geqrf: CholeskyQR2, then the
    Householder representation of Q reconstructed by an LU factorization without pivoting of Q - [S; 0] with
    S = diag(-sign(Re q_jj)) chosen during the elimination (Ballard et al., "Reconstructing Householder vectors from
    TSQR", IPDPS 2014).  p: m x b, m >= b, full column rank.  Returns (out, taus, t): `out` in xGEQR2's layout (S R on and
    above the diagonal of the top block, the reflectors below), taus = -u_jj s_j, t = -U S V1^-H (the T factor)."""
    m, b = p.shape
    g = p.conj().T @ p
    l1 = np.linalg.cholesky(g)
    q = np.linalg.solve(l1, p.conj().T).conj().T          # P L1^-H
    l2 = np.linalg.cholesky(q.conj().T @ q)
    q = np.linalg.solve(l2, q.conj().T).conj().T
    r = l2.conj().T @ l1.conj().T
    w = q[:b, :].copy()
    sgn = np.zeros(b)
    for j in range(b):
        sgn[j] = -1.0 if w[j, j].real >= 0 else 1.0
        w[j, j] -= sgn[j]
        w[j + 1:, j] /= w[j, j]
        w[j + 1:, j + 1:] -= np.outer(w[j + 1:, j], w[j, j + 1:])
    y1 = np.tril(w, -1) + np.eye(b, dtype=p.dtype)
    u = np.triu(w)
    y2 = np.linalg.solve(u.T, q[b:, :].T).T               # Q2 U^-1
    taus = (-np.diag(u) * sgn).astype(p.dtype)
    t = np.linalg.solve(y1.conj(), (-u * sgn[None, :]).T).T   # (-U S) V1^-H
    out = np.vstack([np.tril(y1, -1) + np.triu(sgn[:, None] * r), y2]).astype(p.dtype)
    return out, taus, t.astype(p.dtype)


def well_formed_v(p: np.ndarray, nrefls: int) -> np.ndarray:
    """setupReflectorPanelV (:364-422): the first nrefls columns as a unit lower trapezoidal matrix."""
    v = np.tril(p[:, :nrefls], -1).copy()
    for j in range(nrefls):
        v[j, j] = 1
    return v


def compute_t_factor(v: np.ndarray, taus: np.ndarray) -> np.ndarray:
    """t_factor_impl.h:60-131: T(0:j, j) = -tau_j V(j:, 0:j)^H V(j:, j), T(j, j) = tau_j, then column by column
    t_j = T(0:j, 0:j) t_j (xTRMV)."""
    k = v.shape[1]
    t = np.zeros((k, k), dtype=v.dtype)
    for j in range(k):
        t[j, j] = taus[j]
        if j:
            t[:j, j] += -taus[j] * (v[j:, :j].conj().T @ v[j:, j])
    for j in range(k):
        if j:
            t[:j, j] = np.triu(t[:j, :j]) @ t[:j, j]
    return t


def reduction_to_band(a: np.ndarray, nb: int, band: int):
    """ReductionToBand::call (impl.h:968-1110) on a dense column-major matrix whose LOWER triangle holds the
    Hermitian input; in place; the strict upper triangle is neither read nor written.  Returns taus."""
    n = a.shape[0]
    assert a.shape == (n, n) and band >= 2 and nb % band == 0
    dt = a.dtype
    nrefls = nr_reflectors(n, band)
    taus = np.zeros(nrefls, dtype=dt)
    if nrefls == 0:
        return taus
    ntiles = (nrefls - 1) // band + 1
    for j_sub in range(ntiles):
        r0, c0 = (j_sub + 1) * band, j_sub * band
        nrefls_tile = min(band, nrefls - j_sub * band)
        panel = a[r0:, c0:c0 + band]                       # SubPanelView (views.h:128-171): a view, in place
        taus[c0:c0 + nrefls_tile] = compute_panel_reflectors(panel, nrefls_tile)
        v = well_formed_v(panel, nrefls_tile)
        t = compute_t_factor(v, taus[c0:c0 + nrefls_tile])
        at0 = r0                                            # at_offset = (r0, c0 + band) = (r0, r0)
        if at0 >= n:
            break
        at = a[at0:, at0:]
        w = v @ np.triu(t)                                  # trmmComputeW
        full = np.tril(at) + np.tril(at, -1).conj().T       # hemmComputeX: only the lower part is referenced
        np.fill_diagonal(full, np.real(np.diag(at)))        # (xHEMM ignores the imaginary part of the diagonal)
        x = full @ w
        w2 = w.conj().T @ x                                 # gemmComputeW2
        x = x - dt.type(0.5) * (v @ w2)                     # gemmUpdateX
        upd = x @ v.conj().T + v @ x.conj().T               # her2kUpdateTrailingMatrix: lower part only
        m = at.shape[0]
        il = np.tril_indices(m)
        at[il] -= upd[il]
        if np.iscomplexobj(at):                             # xHER2K leaves a real diagonal
            d = np.arange(m)
            at[d, d] = np.real(at[d, d])
    return taus


# ------------------------------------------------------------------------------------------ back-transform
def bt_reduction_to_band(c: np.ndarray, v_mat: np.ndarray, taus: np.ndarray, nb: int, band: int) -> None:
    """BackTransformationReductionToBand::call (bt_reduction_to_band/impl.h:132-236): C <- Q C in place with
    Q = H_0 H_1 ... applied in blocks of nb reflectors, last block first: W = V T^H, W2 = W^H C, C -= V W2."""
    n = v_mat.shape[0]
    total = n - band - 1
    if total <= 0 or c.size == 0:
        return
    nblocks = (total - 1) // nb + 1
    for k in range(nblocks - 1, -1, -1):
        nrefl = min(nb, total - k * nb)
        r0, c0 = k * nb + band, k * nb
        v = well_formed_v(v_mat[r0:, c0:c0 + nrefl], nrefl)
        t = compute_t_factor(v, taus[c0:c0 + nrefl])
        w = v @ np.triu(t).conj().T
        w2 = w.conj().T @ c[r0:, :]
        c[r0:, :] -= v @ w2


# ------------------------------------------------------------------------------------------ the reference's checker
def split_band(a_out: np.ndarray, band: int) -> np.ndarray:
    """splitReflectorsAndBand + setupHermitianBand (test_reduction_to_band.cpp:141-206): the Hermitian band
    matrix B held by the lower part of the result (main diagonal + `band` sub-diagonals)."""
    n = a_out.shape[0]
    b = np.zeros_like(a_out)
    for d in range(0, band + 1):
        idx = np.arange(n - d)
        b[idx + d, idx] = a_out[idx + d, idx]
    b = b + np.tril(b, -1).conj().T
    return b


def apply_q(a_out: np.ndarray, taus: np.ndarray, band: int, mat: np.ndarray, side: str, adjoint: bool) -> np.ndarray:
    """LAPACK ?ormqr / ?unmqr with the reflectors stored below the band (the call of checkResult,
    test_reduction_to_band.cpp:289-298): mat[band:, :] <- op(Q) mat[band:, :] (side L) or mat[:, band:] <-
    mat[:, band:] op(Q) (side R)."""
    from scipy.linalg import lapack
    k = len(taus)
    if k == 0:
        return mat
    dt = a_out.dtype
    n = a_out.shape[0]
    vpart = np.asfortranarray(a_out[band:, :k])
    tag = {np.dtype(np.float32): "sormqr", np.dtype(np.float64): "dormqr", np.dtype(np.complex64): "cunmqr",
           np.dtype(np.complex128): "zunmqr"}[np.dtype(dt)]
    fn = getattr(lapack, tag)
    trans = ("C" if np.iscomplexobj(a_out) else "T") if adjoint else "N"
    mat = np.asfortranarray(mat.copy())
    if side == "L":
        sub = np.asfortranarray(mat[band:, :])
        lwork = max(1, 64 * sub.shape[1])
        out, _, info = fn("L", trans, vpart, taus, sub, lwork)
        assert info == 0
        mat[band:, :] = out
    else:
        sub = np.asfortranarray(mat[:, band:])
        lwork = max(1, 64 * sub.shape[0])
        out, _, info = fn("R", trans, vpart, taus, sub, lwork)
        assert info == 0
        mat[:, band:] = out
    return mat


def check_result(reference_full: np.ndarray, a_out: np.ndarray, taus: np.ndarray, band: int):
    """checkResult (test_reduction_to_band.cpp:270-310): Q B Q^H must equal the input; absolute tolerance
    max(1, n^2) * error.  reference_full: the full Hermitian input.  Returns (ok, max abs difference, tolerance)."""
    n = a_out.shape[0]
    b = split_band(a_out, band)
    if n > band and len(taus):
        b = apply_q(a_out, taus, band, b, "L", False)
        b = apply_q(a_out, taus, band, b, "R", True)
    tol = max(1, n * n) * error_of(a_out.dtype)
    diff = float(np.abs(b - reference_full).max()) if n else 0.0
    return diff <= tol, diff, tol


def random_hermitian(n: int, dtype, seed: int = 0, banded: int | None = None) -> np.ndarray:
    """set_random_hermitian / set_random_hermitian_banded (util_matrix.h): entries in [-1, 1], real diagonal;
    `banded`: zero outside that many off-diagonals (the reference's 'banded' input structure, band - 1)."""
    rng = np.random.default_rng(seed)
    dt = np.dtype(dtype)
    a = rng.uniform(-1, 1, (n, n))
    if dt.kind == "c":
        a = a + 1j * rng.uniform(-1, 1, (n, n))
    a = np.tril(a) + np.tril(a, -1).conj().T
    a[np.arange(n), np.arange(n)] = np.real(np.diag(a))
    if banded is not None:
        i, j = np.indices((n, n))
        a[np.abs(i - j) > banded] = 0
    return np.asfortranarray(a.astype(dt))
