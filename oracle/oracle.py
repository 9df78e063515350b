"""oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy doorway onto oracle/liboracle.so, the CPU restatement of the reference's tiled
Cholesky path (see the header of oracle/dlaf_oracle.c for what it restates and how it is
pinned).  May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only; the product package dla_future_amd never imports it.

All matrices are numpy arrays in Fortran (column-major) order.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

DTYPES = {"s": np.float32, "d": np.float64, "c": np.complex64, "z": np.complex128}
REAL_OF = {"s": np.float32, "d": np.float64, "c": np.float32, "z": np.float64}


def type_char(dtype) -> str:
    dtype = np.dtype(dtype)
    for k, v in DTYPES.items():
        if np.dtype(v) == dtype:
            return k
    raise TypeError(f"unsupported dtype {dtype}")


def build(force: bool = False) -> str:
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref)."""
    srcs = [os.path.join(_HERE, f) for f in ("dlaf_oracle.c", "dlaf_oracle.h", "oracle_kernels.inc")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/include/dlaf"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


class _C64(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class _C128(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


_SCALAR = {"s": C.c_float, "d": C.c_double, "c": _C64, "z": _C128}
_REAL = {"s": C.c_float, "d": C.c_double, "c": C.c_float, "z": C.c_double}


def _scalar(t: str, v):
    if t in "sd":
        return _SCALAR[t](float(np.real(v)))
    v = complex(v)
    return _SCALAR[t](v.real, v.imag)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    for name, res, args in [
        ("orc_tile_from_element", C.c_long, [C.c_long] * 3),
        ("orc_tile_element_from_element", C.c_long, [C.c_long] * 3),
        ("orc_element_from_tile_and_tile_element", C.c_long, [C.c_long] * 4),
        ("orc_rank_global_tile", C.c_int, [C.c_long, C.c_long, C.c_int, C.c_int, C.c_long]),
        ("orc_local_tile_from_global_tile", C.c_long, [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_long]),
        ("orc_next_local_tile_from_global_tile", C.c_long,
         [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_long]),
        ("orc_global_tile_from_local_tile", C.c_long, [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_long]),
        ("orc_local_nr_tiles", C.c_long, [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int]),
        ("orc_local_size", C.c_long, [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int]),
        ("orc_mt_seed", None, [vp, C.c_uint64]),
        ("orc_mt_next", C.c_uint64, [vp]),
        ("orc_uniform_pm1_d", C.c_double, [vp]),
        ("orc_uniform_pm1_s", C.c_float, [vp]),
        ("orc_baseline_cholesky_d", C.c_int, [C.c_long, C.c_int, vp, C.c_long, C.c_int]),
        ("orc_omp_max_threads", C.c_int, []),
    ]:
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    for t in "sdcz":
        S, R = _SCALAR[t], _REAL[t]
        sig = {
            f"orc_potrf_{t}": (C.c_int, [C.c_char, C.c_int, vp, C.c_int]),
            f"orc_trsm_{t}": (None, [C.c_char] * 4 + [C.c_int, C.c_int, S, vp, C.c_int, vp, C.c_int]),
            f"orc_herk_{t}": (None, [C.c_char] * 2 + [C.c_int, C.c_int, R, vp, C.c_int, R, vp, C.c_int]),
            f"orc_gemm_{t}": (None, [C.c_char] * 2 + [C.c_int] * 3 + [S, vp, C.c_int, vp, C.c_int, S, vp, C.c_int]),
            f"orc_chol_el_a_{t}": (S, [C.c_char, C.c_long, C.c_long]),
            f"orc_chol_el_l_{t}": (S, [C.c_char, C.c_long, C.c_long]),
            f"orc_set_random_hpd_tile_{t}": (None, [C.c_long, C.c_int, C.c_long, C.c_long, R, vp, C.c_int]),
            f"orc_set_random_hpd_{t}": (None, [C.c_long, C.c_int, vp, C.c_long]),
            f"orc_cholesky_local_{t}": (C.c_int, [C.c_char, C.c_long, C.c_int, vp, C.c_long]),
            f"orc_cholesky_dist_{t}": (C.c_int, [C.c_char, C.c_long, C.c_int] + [C.c_int] * 4
                                       + [C.POINTER(vp), C.POINTER(C.c_long)]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
    _lib = L
    return L


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _ld(a: np.ndarray) -> int:
    assert a.flags.f_contiguous or a.ndim == 2 and a.strides[0] == a.itemsize
    return max(1, a.strides[1] // a.itemsize) if a.shape[1] > 1 else max(1, a.shape[0])


def _b(ch: str) -> bytes:
    return ch.encode("ascii")


# ----------------------------------------------------------------------------- tile ops
def potrf(uplo: str, a: np.ndarray) -> int:
    t = type_char(a.dtype)
    return getattr(lib(), f"orc_potrf_{t}")(_b(uplo), a.shape[0], _ptr(a), _ld(a))


def trsm(side, uplo, op, diag, alpha, a: np.ndarray, b: np.ndarray) -> None:
    t = type_char(b.dtype)
    getattr(lib(), f"orc_trsm_{t}")(_b(side), _b(uplo), _b(op), _b(diag), b.shape[0], b.shape[1],
                                    _scalar(t, alpha), _ptr(a), _ld(a), _ptr(b), _ld(b))


def herk(uplo, op, alpha, a: np.ndarray, beta, c: np.ndarray, k=None) -> None:
    t = type_char(c.dtype)
    n = c.shape[0]
    if k is None:
        k = a.shape[1] if op in "Nn" else a.shape[0]
    getattr(lib(), f"orc_herk_{t}")(_b(uplo), _b(op), n, k, _REAL[t](alpha), _ptr(a), _ld(a),
                                    _REAL[t](beta), _ptr(c), _ld(c))


def gemm(opa, opb, alpha, a: np.ndarray, b: np.ndarray, beta, c: np.ndarray, k=None) -> None:
    t = type_char(c.dtype)
    m, n = c.shape
    if k is None:
        k = a.shape[1] if opa in "Nn" else a.shape[0]
    getattr(lib(), f"orc_gemm_{t}")(_b(opa), _b(opb), m, n, k, _scalar(t, alpha), _ptr(a), _ld(a),
                                    _ptr(b), _ld(b), _scalar(t, beta), _ptr(c), _ld(c))


# ----------------------------------------------------------------------------- generators
def _from_scalar(t, v):
    if t in "sd":
        return v
    return complex(v.re, v.im)


def cholesky_setters(uplo: str, n: int, dtype):
    """(A, L) of getCholeskySetters (util_generic_lapack.h:39-68) as full n x n arrays."""
    t = type_char(dtype)
    fa = getattr(lib(), f"orc_chol_el_a_{t}")
    fl = getattr(lib(), f"orc_chol_el_l_{t}")
    a = np.empty((n, n), dtype=dtype, order="F")
    l = np.empty((n, n), dtype=dtype, order="F")
    u = _b(uplo)
    for j in range(n):
        for i in range(n):
            a[i, j] = _from_scalar(t, fa(u, i, j))
            l[i, j] = _from_scalar(t, fl(u, i, j))
    return a, l


def set_random_hpd(n: int, nb: int, dtype) -> np.ndarray:
    """set_random_hermitian_positive_definite (util_matrix.h:498-501): full global matrix."""
    t = type_char(dtype)
    a = np.zeros((n, n), dtype=dtype, order="F")
    if n:
        getattr(lib(), f"orc_set_random_hpd_{t}")(n, nb, _ptr(a), max(1, n))
    return a


def random_hpd_tile(n: int, nb: int, ti: int, tj: int, dtype) -> np.ndarray:
    t = type_char(dtype)
    rows = min(nb, n - ti * nb)
    cols = min(nb, n - tj * nb)
    tile = np.zeros((rows, cols), dtype=dtype, order="F")
    getattr(lib(), f"orc_set_random_hpd_tile_{t}")(n, nb, ti, tj, _REAL[t](2 * n), _ptr(tile), max(1, rows))
    return tile


class MT19937_64:
    def __init__(self, seed: int):
        self._buf = (C.c_uint64 * 313)()
        lib().orc_mt_seed(C.cast(self._buf, C.c_void_p), seed)

    def raw(self) -> int:
        return lib().orc_mt_next(C.cast(self._buf, C.c_void_p))

    def uniform_d(self) -> float:
        return lib().orc_uniform_pm1_d(C.cast(self._buf, C.c_void_p))

    def uniform_s(self) -> float:
        return lib().orc_uniform_pm1_s(C.cast(self._buf, C.c_void_p))


# ----------------------------------------------------------------------------- distribution
def rank_global_tile(gt, grid, src, tpb=1, toff=0):
    return lib().orc_rank_global_tile(gt, tpb, grid, src, toff)


def local_tile_from_global_tile(gt, grid, rank, src, tpb=1, toff=0):
    return lib().orc_local_tile_from_global_tile(gt, tpb, grid, rank, src, toff)


def next_local_tile_from_global_tile(gt, grid, rank, src, tpb=1, toff=0):
    return lib().orc_next_local_tile_from_global_tile(gt, tpb, grid, rank, src, toff)


def global_tile_from_local_tile(lt, grid, rank, src, tpb=1, toff=0):
    return lib().orc_global_tile_from_local_tile(lt, tpb, grid, rank, src, toff)


def local_size(n, nb, grid, rank, src):
    return lib().orc_local_size(n, nb, grid, rank, src)


def local_nr_tiles(n, nb, grid, rank, src):
    return lib().orc_local_nr_tiles(n, nb, grid, rank, src)


def scatter(a: np.ndarray, nb: int, pr: int, pc: int, sr: int = 0, sc: int = 0, extra_ld: int = 0):
    """Global m x n matrix -> dict {(r,c): local column-major array} (2-D block-cyclic,
    ScaLAPACK local layout; misc/matrix_distribution.md).  Arrays have lld = max(1, rows)+extra_ld;
    the returned array is the (rows x cols) view."""
    m, n = a.shape
    mt = (m + nb - 1) // nb if m else 0
    nt = (n + nb - 1) // nb if n else 0
    out = {}
    for r in range(pr):
        for c in range(pc):
            rows = local_size(m, nb, pr, r, sr)
            cols = local_size(n, nb, pc, c, sc)
            store = np.full((max(1, rows) + extra_ld, max(cols, 1)), -77.0, dtype=a.dtype, order="F")
            loc = store[:rows, :cols]
            for gj in range(nt):
                if rank_global_tile(gj, pc, sc) != c:
                    continue
                lj = local_tile_from_global_tile(gj, pc, c, sc)
                for gi in range(mt):
                    if rank_global_tile(gi, pr, sr) != r:
                        continue
                    li = local_tile_from_global_tile(gi, pr, r, sr)
                    blk = a[gi * nb:(gi + 1) * nb, gj * nb:(gj + 1) * nb]
                    loc[li * nb:li * nb + blk.shape[0], lj * nb:lj * nb + blk.shape[1]] = blk
            out[(r, c)] = loc
    return out


def gather(locs, n: int, nb: int, pr: int, pc: int, sr: int = 0, sc: int = 0, dtype=None, m=None) -> np.ndarray:
    """Inverse of scatter; the global matrix is m x n (m defaults to n)."""
    dtype = dtype or next(iter(locs.values())).dtype
    m = n if m is None else m
    a = np.zeros((m, n), dtype=dtype, order="F")
    mt = (m + nb - 1) // nb if m else 0
    nt = (n + nb - 1) // nb if n else 0
    for gj in range(nt):
        c = rank_global_tile(gj, pc, sc)
        lj = local_tile_from_global_tile(gj, pc, c, sc)
        for gi in range(mt):
            r = rank_global_tile(gi, pr, sr)
            li = local_tile_from_global_tile(gi, pr, r, sr)
            rows = min(nb, m - gi * nb)
            cols = min(nb, n - gj * nb)
            a[gi * nb:gi * nb + rows, gj * nb:gj * nb + cols] = \
                locs[(r, c)][li * nb:li * nb + rows, lj * nb:lj * nb + cols]
    return a


def _axis_members(n: int, nb: int, p: int, src: int):
    """global element indices each process of a block-cyclic axis holds, in local order"""
    g = np.arange(n)
    owner = ((g // nb) + src) % p
    return [g[owner == r] for r in range(p)]


def scatter_rect(a: np.ndarray, mb: int, nb: int, pr: int, pc: int, sr: int = 0, sc: int = 0, extra_ld: int = 0):
    """scatter() for MB x NB blocks (matrix.h: block size != square), element-index based."""
    m, n = a.shape
    ri, ci = _axis_members(m, mb, pr, sr), _axis_members(n, nb, pc, sc)
    out = {}
    for r in range(pr):
        for c in range(pc):
            rows, cols = len(ri[r]), len(ci[c])
            store = np.full((max(1, rows) + extra_ld, max(cols, 1)), -77.0, dtype=a.dtype, order="F")
            loc = store[:rows, :cols]
            if rows and cols:
                loc[:, :] = a[np.ix_(ri[r], ci[c])]
            out[(r, c)] = loc
    return out


def gather_rect(locs, m: int, n: int, mb: int, nb: int, pr: int, pc: int, sr: int = 0, sc: int = 0, dtype=None) -> np.ndarray:
    """Inverse of scatter_rect."""
    dtype = dtype or next(iter(locs.values())).dtype
    a = np.zeros((m, n), dtype=dtype, order="F")
    ri, ci = _axis_members(m, mb, pr, sr), _axis_members(n, nb, pc, sc)
    for r in range(pr):
        for c in range(pc):
            if len(ri[r]) and len(ci[c]):
                a[np.ix_(ri[r], ci[c])] = locs[(r, c)][:len(ri[r]), :len(ci[c])]
    return a


# ----------------------------------------------------------------------------- algorithms
def cholesky_local(uplo: str, a: np.ndarray, nb: int) -> int:
    """cholesky/impl.h:150-189 / :316-348 in place on a full (local) matrix."""
    t = type_char(a.dtype)
    n = a.shape[0]
    return getattr(lib(), f"orc_cholesky_local_{t}")(_b(uplo), n, nb, _ptr(a), _ld(a))


def cholesky_dist(uplo: str, locs, n: int, nb: int, pr: int, pc: int, sr: int = 0, sc: int = 0) -> int:
    """cholesky/impl.h:192-313 / :351-452 on the per-rank local arrays (all ranks in-process)."""
    t = type_char(next(iter(locs.values())).dtype)
    ptrs = (C.c_void_p * (pr * pc))()
    llds = (C.c_long * (pr * pc))()
    for r in range(pr):
        for c in range(pc):
            loc = locs[(r, c)]
            ptrs[r + c * pr] = loc.ctypes.data
            llds[r + c * pr] = _ld(loc) if loc.size else max(1, loc.strides[1] // loc.itemsize if loc.ndim == 2 and loc.shape[1] > 0 else 1)
    return getattr(lib(), f"orc_cholesky_dist_{t}")(_b(uplo), n, nb, pr, pc, sr, sc, ptrs, llds)


def baseline_cholesky_d(a: np.ndarray, nb: int, nthreads: int) -> int:
    assert a.dtype == np.float64 and a.flags.f_contiguous
    return lib().orc_baseline_cholesky_d(a.shape[0], nb, _ptr(a), max(1, a.shape[0]), nthreads)


def omp_max_threads() -> int:
    return lib().orc_omp_max_threads()


# ----------------------------------------------------------------------------- checker
def _polar(dt, r, theta):
    """TypeUtilities<T>::polar (test/include/dlaf_test/util_types.h): r for real types, r e^{i theta} for complex"""
    if np.dtype(dt).kind == "c":
        return np.dtype(dt).type(r * np.cos(theta) + 1j * r * np.sin(theta))
    return np.dtype(dt).type(r)


def triangular_system(side: str, uplo: str, op: str, diag: str, alpha, m: int, n: int, dtype):
    """getTriangularSystem (test/include/dlaf_test/matrix/util_generic_blas.h:258-373): returns (A, B, X) with
    op(A) X = alpha B (side L) or X op(A) = alpha B (side R).  A holds -9.9 outside the referenced triangle and
    on a unit diagonal, exactly as the reference's generator stores it."""
    dt = np.dtype(dtype).type
    alpha = dt(alpha)
    op_a_lower = (uplo == "L" and op == "N") or (uplo == "U" and op != "N")
    na = m if side == "L" else n
    i = np.arange(na, dtype=np.float64)[:, None]
    k = np.arange(na, dtype=np.float64)[None, :]
    if side == "L":
        r, th = (i + 1) / (k + .5), 2 * i - k
    else:
        r, th = (k + 1) / (i + .5), 2 * k - i
    cx = np.dtype(dtype).kind == "c"
    op_a = (r * np.exp(1j * th)).astype(dtype) if cx else r.astype(dtype)
    skip = (i < k) if op_a_lower else (i > k)
    if diag == "U":
        skip = skip | (i == k)
    op_a = np.where(skip, dt(-9.9), op_a)
    ii = np.arange(m, dtype=np.float64)[:, None]
    jj = np.arange(n, dtype=np.float64)[None, :]
    if side == "L":
        xr, xt = (ii + .5) / (jj + 2), ii + jj
        kk = (ii + 1) if op_a_lower else (m - ii)
        gr, gt = (ii + 1) / (jj + 2), 2 * ii + jj
    else:
        xr, xt = (jj + .5) / (ii + 2), ii + jj
        kk = (n - jj) if op_a_lower else (jj + 1)
        gr, gt = (jj + 1) / (ii + 2), ii + 2 * jj
    kk = np.broadcast_to(kk, (m, n))
    x = (xr * np.exp(1j * xt)).astype(dtype) if cx else np.broadcast_to(xr, (m, n)).astype(dtype)
    gamma = (gr * np.exp(1j * gt)) if cx else np.broadcast_to(gr, (m, n))
    b = (((kk - 1) * gamma + x) / alpha) if diag == "U" else (kk * gamma / alpha)
    unop = {"N": lambda z: z, "T": lambda z: z.T, "C": lambda z: z.conj().T}
    a = np.asfortranarray(unop[op](op_a).astype(dtype))
    return a, np.asfortranarray(b.astype(dtype)), np.asfortranarray(np.broadcast_to(x, (m, n)).astype(dtype))


def eps_of(dtype) -> float:
    return float(np.finfo(REAL_OF[type_char(dtype)]).eps)


def tri(uplo: str, a: np.ndarray) -> np.ndarray:
    return np.tril(a) if uplo in "Ll" else np.triu(a)


def cholesky_residual(uplo: str, a_orig: np.ndarray, fact: np.ndarray) -> float:
    """miniapp/miniapp_cholesky.cpp:408-443: max|A - L L^H| / max|A| over the uplo triangle."""
    n = a_orig.shape[0]
    if n == 0:
        return 0.0
    wide = np.complex128 if np.iscomplexobj(a_orig) else np.float64
    f = tri(uplo, fact).astype(wide)
    prod = f @ f.conj().T if uplo in "Ll" else f.conj().T @ f
    d = tri(uplo, np.abs(a_orig.astype(wide) - prod))
    return float(d.max() / np.abs(tri(uplo, a_orig)).max())


def check_near(expected: np.ndarray, actual: np.ndarray, rel: float, abs_: float):
    """CHECK_MATRIX_NEAR / CHECK_TILE_NEAR semantics
    (test/include/dlaf_test/matrix/util_matrix.h:256-281): an element passes when
    diff < abs_err OR diff / max(|expected|, |value|) < rel_err.  Returns (ok, max diff)."""
    diff = np.abs(expected - actual)
    abs_max = np.maximum(np.abs(expected), np.abs(actual))
    with np.errstate(divide="ignore", invalid="ignore"):
        relok = np.where(abs_max > 0, diff / np.where(abs_max > 0, abs_max, 1), np.inf) < rel
    ok = (diff < abs_) | relok
    return bool(ok.all()), (float(diff.max()) if diff.size else 0.0)


# ---- generalized -> standard eigenproblem ---------------------------------------------
# Reference: GenToStd::call_L (include/dlaf/eigensolver/gen_to_std/impl.h:222-283), tile::hegst
# (include/dlaf/lapack/tile.h:209-218 -> lapack::hegst = LAPACK xHEGST/xHEGS2 itype 1), known answers
# getGenToStdElementSetters (test/include/dlaf_test/matrix/util_generic_lapack.h:96-150).
def gen_to_std_setters(uplo: str, n: int, dtype, alpha: float = -2.0, beta: float = 1.5, gamma: float = 0.95):
    """(T, A, B) of getGenToStdElementSetters, itype 1 (test_gen_to_std.cpp:68-70 uses alpha -2, beta 1.5,
    gamma .95): T the triangular factor, A the input, B = inv(L) A inv(L^H) (uplo L) / inv(U^H) A inv(U) (uplo U);
    -9.9 in the triangle that must not be touched."""
    dt = np.dtype(dtype)
    cx = np.issubdtype(dt, np.complexfloating)
    i, j = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    ph = np.exp(1j * alpha * (i - j)) if cx else np.ones_like(i)
    t = (beta / np.exp2(np.abs(i - j))) * ph
    a = ((i + 1) * (j + 1) * (beta * beta * gamma) / np.exp2(i + j)) * ph
    b = (gamma / np.exp2(i + j)) * ph
    other = (i < j) if uplo in "Ll" else (i > j)
    out = []
    for m in (t, a, b):
        m = np.where(other, -9.9, m)
        out.append(np.asfortranarray(m.astype(dt)))
    return tuple(out)


def hegst_tile(uplo: str, a: np.ndarray, b: np.ndarray) -> None:
    """tile::hegst(itype 1, uplo, a, b) = LAPACK xHEGS2: a <- inv(L) a inv(L^H) (uplo L, b = L) or inv(U^H) a inv(U)
    (uplo U, b = U), unblocked, in place on the uplo triangle of a (the other triangle is never referenced).
    The upper variant is the lower algorithm on the transposed storage (B^T = L'^-1 A^T L'^-H with L' = U^T)."""
    n = a.shape[0]
    if uplo in "Uu":
        at = np.asfortranarray(a.T.copy())
        hegst_tile("L", at, np.asfortranarray(b.T.copy()))
        iu = np.triu_indices(n)
        a[iu] = at.T[iu]
        return
    real = a.real.dtype.type
    for k in range(n):
        bkk = real(b[k, k].real)
        akk = real(a[k, k].real) / (bkk * bkk)
        a[k, k] = akk
        if k < n - 1:
            x = a[k + 1:, k]
            lcol = b[k + 1:, k]
            x *= real(1) / bkk
            ct = real(-0.5) * akk
            x += ct * lcol
            # her2, lower: A22 -= x l^H + l x^H
            upd = np.outer(x, lcol.conj()) + np.outer(lcol, x.conj())
            sub = a[k + 1:, k + 1:]
            il = np.tril_indices(n - k - 1)
            sub[il] -= upd[il]
            d = np.arange(n - k - 1)
            sub[d, d] = sub[d, d].real
            x += ct * lcol
            # trsv: x <- inv(L22) x (lower, no transpose, non-unit)
            l22 = b[k + 1:, k + 1:]
            for r in range(n - k - 1):
                x[r] = (x[r] - np.dot(l22[r, :r], x[:r])) / l22[r, r]


def gen_to_std_local(uplo: str, a: np.ndarray, l: np.ndarray, nb: int) -> None:
    """GenToStd::call_L (impl.h:222-283) tile by tile, in place on the uplo triangle of `a`; `l` holds the Cholesky
    factor in the same triangle and is only read.  uplo U (call_U, the mirrored loop) runs as the lower algorithm
    on the transposed storage."""
    n = a.shape[0]
    if n == 0:
        return
    if uplo in "Uu":
        at = np.asfortranarray(a.T.copy())
        gen_to_std_local("L", at, np.asfortranarray(l.T.copy()), nb)
        iu = np.triu_indices(n)
        a[iu] = at.T[iu]
        return
    nt = (n + nb - 1) // nb

    def tl(m, i, j):
        return m[i * nb:min(n, (i + 1) * nb), j * nb:min(n, (j + 1) * nb)]

    def herm(t):
        h = np.tril(t) + np.tril(t, -1).conj().T
        d = np.arange(h.shape[0])
        h[d, d] = h[d, d].real
        return h

    half = a.real.dtype.type(0.5)
    for k in range(nt):
        akk, lkk = tl(a, k, k), tl(l, k, k)
        tmp = np.asfortranarray(akk.copy())
        hegst_tile("L", tmp, np.asfortranarray(lkk))                       # hegstDiagTile
        il = np.tril_indices(tmp.shape[0])
        akk[il] = tmp[il]
        if k == nt - 1:
            continue
        akk_full = herm(akk)
        lkk_f = np.asfortranarray(np.tril(lkk))
        for i in range(k + 1, nt):
            aik = np.asfortranarray(tl(a, i, k).copy())
            trsm("R", "L", "C", "N", 1.0, lkk_f, aik)                       # trsmPanelTile
            aik -= half * (tl(l, i, k) @ akk_full)                          # hemmPanelTile
            tl(a, i, k)[...] = aik
        for j in range(k + 1, nt):
            ajk, ljk = tl(a, j, k), tl(l, j, k)
            ajj = tl(a, j, j)
            upd = ajk @ ljk.conj().T + ljk @ ajk.conj().T                   # her2kTrailingDiagTile
            ilj = np.tril_indices(ajj.shape[0])
            ajj[ilj] -= upd[ilj]
            d = np.arange(ajj.shape[0])
            ajj[d, d] = ajj[d, d].real
            for i in range(j + 1, nt):                                      # gemmTrailingMatrixTile x 2
                tl(a, i, j)[...] -= tl(a, i, k) @ ljk.conj().T + tl(l, i, k) @ ajk.conj().T
        for i in range(k + 1, nt):
            tl(a, i, k)[...] -= half * (tl(l, i, k) @ akk_full)             # hemmPanelTile
        for j in range(k + 1, nt):
            ajk = np.asfortranarray(tl(a, j, k).copy())
            trsm("L", "L", "N", "N", 1.0, np.asfortranarray(np.tril(tl(l, j, j))), ajk)   # trsmPanelUpdateTile
            tl(a, j, k)[...] = ajk
            for i in range(j + 1, nt):
                tl(a, i, k)[...] -= tl(l, i, j) @ ajk                       # gemmPanelUpdateTile
