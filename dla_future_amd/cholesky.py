"""Host-side mirror of the reference's Cholesky interface over the C ABI.

Names and argument meaning follow the reference (file:line in the docstrings); arrays are numpy,
column-major ("F" order) like every DLA-Future local matrix.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .capi import BARRIER_FN, BCAST_FN, DLAFDescriptor, lib, type_char


def initialize(print_config: bool = False) -> None:
    """dlaf_initialize (include/dlaf_c/init.h:27).  Idempotent."""
    args = [b"dlaf"] + ([b"--dlaf:print-config"] if print_config else [])
    argv = (C.c_char_p * len(args))(*args)
    lib().dlaf_initialize(0, None, len(args), argv)


def finalize() -> None:
    """dlaf_finalize (include/dlaf_c/init.h:35).  Idempotent; frees every grid."""
    lib().dlaf_finalize()


def make_descriptor(n: int, nb: int, ld: int, isrc: int = 0, jsrc: int = 0) -> DLAFDescriptor:
    return DLAFDescriptor(n, n, nb, nb, isrc, jsrc, 0, 0, max(1, ld))


def _ld_of(a: np.ndarray) -> int:
    if a.ndim != 2:
        raise ValueError("expected a 2-D array")
    if a.size and a.strides[0] != a.itemsize:
        raise ValueError("local matrices must be column-major (Fortran order)")
    return max(1, a.strides[1] // a.itemsize) if a.shape[1] > 1 else max(1, a.shape[0])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Grid:
    """A process grid (reference: comm::CommunicatorGrid, communication/communicator_grid.h:37-153;
    C side dlaf_create_grid, include/dlaf_c/grid.h:31).  Build one with

      Grid.single()                         1x1, no communication
      Grid.rccl(unique_id, ...)             RCCL over xGMI (production multi-GPU)
      Grid.from_torch(nprow, npcol, order)  RCCL, unique id shipped through torch.distributed
      Grid.host(..., bcast, barrier)        host-staged broadcasts supplied by the caller
    """

    def __init__(self, context: int, nranks: int, rank: int, keep=None):
        if context < 0:
            raise ValueError("grid creation failed (bad shape/rank?)")
        self.context = context
        self.nranks = nranks
        self.rank = rank
        self._keep = keep  # callbacks must outlive the grid
        r = [C.c_int() for _ in range(4)]
        lib().dlaf_mi355x_grid_info(context, *(C.byref(x) for x in r))
        self.nprow, self.npcol, self.myrow, self.mycol = (x.value for x in r)

    @classmethod
    def single(cls) -> "Grid":
        return cls(lib().dlaf_mi355x_create_grid_single(), 1, 0)

    @classmethod
    def rccl(cls, unique_id: bytes, nranks: int, rank: int, nprow: int, npcol: int, order: str = "R") -> "Grid":
        buf = C.create_string_buffer(bytes(unique_id), 128)
        ctx = lib().dlaf_mi355x_create_grid_rccl(buf, nranks, rank, nprow, npcol, order.encode())
        return cls(ctx, nranks, rank)

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        lib().dlaf_mi355x_rccl_unique_id(buf)
        return buf.raw

    @classmethod
    def from_torch(cls, nprow: int, npcol: int, order: str = "R") -> "Grid":
        """One process per GPU launched by torch.distributed.run: rank 0 makes the RCCL unique id
        and broadcasts it through the default process group."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        if world == 1:
            return cls.single()
        box = [cls.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.rccl(box[0], world, rank, nprow, npcol, order)

    @classmethod
    def host(cls, nranks: int, rank: int, nprow: int, npcol: int, order: str, bcast, barrier=None) -> "Grid":
        """bcast(axis, root, memoryview) must broadcast the buffer in place along this process's
        row (axis 0, root = process column) or column (axis 1, root = process row) communicator."""

        def _bcast(_user, axis, root, buf, nbytes):
            try:
                mv = (C.c_char * nbytes).from_address(buf)
                bcast(axis, root, memoryview(mv).cast("B"))
                return 0
            except Exception as e:  # pragma: no cover - surfaced as a fatal error by the library
                print(f"[dla_future_amd] host broadcast failed: {e!r}")
                return 1

        def _barrier(_user):
            if barrier is not None:
                barrier()
            return 0

        cb, cbar = BCAST_FN(_bcast), BARRIER_FN(_barrier)
        ctx = lib().dlaf_mi355x_create_grid_host(nranks, rank, nprow, npcol, order.encode(), cb, cbar, None)
        return cls(ctx, nranks, rank, keep=(cb, cbar))

    def barrier(self) -> None:
        lib().dlaf_mi355x_grid_barrier(self.context)

    def selftest(self, nbytes: int = 1 << 20) -> int:
        """Collective wiring check of the row / column communicators; number of failed checks (0 = good)."""
        return int(lib().dlaf_mi355x_grid_selftest(self.context, nbytes))

    def comm_log(self, enable: bool = True) -> None:
        """Start (and clear) / stop the communication log of this grid (dlaf_mi355x_grid_comm_log)."""
        lib().dlaf_mi355x_grid_comm_log(self.context, 1 if enable else 0)

    def comm_log_events(self):
        """[(kind, root, bytes, grouped), ...] recorded since comm_log(True): kind 0 row broadcast, 1 column
        broadcast, 2 step marker (root = step), 3 barrier, 4 all-reduce."""
        n = lib().dlaf_mi355x_grid_comm_log_read(self.context, None, 0)
        buf = (C.c_long * (4 * max(1, n)))()
        lib().dlaf_mi355x_grid_comm_log_read(self.context, buf, n)
        return [tuple(buf[4 * i:4 * i + 4]) for i in range(n)]

    def free(self) -> None:
        """dlaf_free_grid (include/dlaf_c/grid.h:39)."""
        if self.context >= 0:
            lib().dlaf_free_grid(self.context)
            self.context = -1

    def local_shape(self, n: int, nb: int, isrc: int = 0, jsrc: int = 0):
        from . import distribution as d
        return (d.local_size(n, nb, self.nprow, self.myrow, isrc), d.local_size(n, nb, self.npcol, self.mycol, jsrc))


def cholesky_factorization(grid: Grid, uplo: str, a: np.ndarray, nb: int, isrc: int = 0, jsrc: int = 0,
                           n: int | None = None) -> int:
    """dlaf_cholesky_factorization_{s,d,c,z} (include/dlaf_c/factorization/cholesky.h:32-47) ==
    dlaf::cholesky_factorization(grid, uplo, matrix) (include/dlaf/factorization/cholesky.h:67-79).

    `a` is this process's local column-major part (host memory); factored in place in the `uplo`
    triangle.  Returns 0, or the LAPACK info of a non-positive-definite leading minor."""
    t = type_char(a.dtype)
    if n is None:
        if grid.nranks != 1:
            raise ValueError("the global size n is required on a distributed grid")
        n = a.shape[0]
    desc = make_descriptor(n, nb, _ld_of(a), isrc, jsrc)
    fn = getattr(lib(), f"dlaf_cholesky_factorization_{t}")
    return fn(grid.context, uplo.encode(), _ptr(a), desc)


def pxpotrf(uplo: str, n: int, a: np.ndarray, ia: int, ja: int, desca) -> int:
    """dlaf_p{s,d,c,z}potrf (include/dlaf_c/factorization/cholesky.h:74-87); desca is the 9-int
    ScaLAPACK descriptor {1, ctxt, M, N, MB, NB, RSRC, CSRC, LLD}.  Returns info."""
    t = type_char(a.dtype)
    d = (C.c_int * 9)(*[int(x) for x in desca])
    info = C.c_int(-999)
    getattr(lib(), f"dlaf_p{t}potrf")(uplo.encode(), n, _ptr(a), ia, ja, d, C.byref(info))
    return info.value


def triangular_solver(grid: Grid, side: str, uplo: str, op: str, diag: str, alpha, a: np.ndarray, b: np.ndarray,
                      nb: int, m: int | None = None, n: int | None = None, a_src=(0, 0), b_src=(0, 0),
                      b_block: tuple[int, int] | None = None) -> None:
    """dlaf::triangular_solver(grid, side, uplo, op, diag, alpha, A, B)
    (include/dlaf/solver/triangular.h:41-177) == dlaf_mi355x_triangular_solver_{s,d,c,z}:
    side 'L': op(A) X = alpha B, side 'R': X op(A) = alpha B; `b` (this process's local column-major part of
    the m x n right-hand sides) is overwritten by X.  `a`: local part of the triangular matrix.
    `nb`: the square block of A; `b_block` = (MB, NB) of B when they differ (triangular.h:41-60: B's block along the
    triangular dimension must be A's, the other one is free)."""
    t = type_char(b.dtype)
    if a.dtype != b.dtype:
        raise ValueError("A and B must have the same element type")
    if m is None or n is None:
        if grid.nranks != 1:
            raise ValueError("the global size m x n of B is required on a distributed grid")
        m, n = b.shape
    na = m if side.upper() == "L" else n
    da = DLAFDescriptor(na, na, nb, nb, a_src[0], a_src[1], 0, 0, _ld_of(a))
    mb_b, nb_b = b_block if b_block is not None else (nb, nb)
    db = DLAFDescriptor(m, n, mb_b, nb_b, b_src[0], b_src[1], 0, 0, _ld_of(b))
    al = np.array([alpha], dtype=b.dtype)
    fn = getattr(lib(), f"dlaf_mi355x_triangular_solver_{t}")
    r = fn(grid.context, side.encode(), uplo.encode(), op.encode(), diag.encode(), _ptr(al), _ptr(a), da, _ptr(b), db)
    if r != 0:
        raise ValueError(f"dlaf_mi355x_triangular_solver_{t} failed with {r}")


def pxpotrs(uplo: str, n: int, nrhs: int, a: np.ndarray, ia: int, ja: int, desca, b: np.ndarray, ib: int, jb: int,
            descb) -> int:
    """dlaf_mi355x_p{s,d,c,z}potrs: A X = B with the factor p?potrf left in `a`; returns info."""
    t = type_char(b.dtype)
    da = (C.c_int * 9)(*[int(x) for x in desca])
    db = (C.c_int * 9)(*[int(x) for x in descb])
    info = C.c_int(-999)
    getattr(lib(), f"dlaf_mi355x_p{t}potrs")(uplo.encode(), n, nrhs, _ptr(a), ia, ja, da, _ptr(b), ib, jb, db,
                                             C.byref(info))
    return info.value


def generalized_to_standard(grid: Grid, uplo: str, a: np.ndarray, b: np.ndarray, nb: int, isrc: int = 0, jsrc: int = 0,
                            n: int | None = None) -> int:
    """dlaf::eigensolver::internal::generalized_to_standard(grid, uplo, mat_a, mat_b)
    (include/dlaf/eigensolver/gen_to_std.h:50, :101) == dlaf_mi355x_generalized_to_standard_{s,d,c,z}:
    `a` (this process's local part of the Hermitian A) is overwritten by inv(L) A inv(L^H) (uplo 'L') or
    inv(U^H) A inv(U) (uplo 'U') in its uplo triangle; `b` holds the Cholesky factor of B in the same triangle."""
    t = type_char(a.dtype)
    if a.dtype != b.dtype:
        raise ValueError("A and B must have the same element type")
    if n is None:
        if grid.nranks != 1:
            raise ValueError("the global size n is required on a distributed grid")
        n = a.shape[0]
    da = make_descriptor(n, nb, _ld_of(a), isrc, jsrc)
    db = make_descriptor(n, nb, _ld_of(b), isrc, jsrc)
    return getattr(lib(), f"dlaf_mi355x_generalized_to_standard_{t}")(grid.context, uplo.encode(), _ptr(a), da, _ptr(b), db)


def pxhegst(ibtype: int, uplo: str, n: int, a: np.ndarray, ia: int, ja: int, desca, b: np.ndarray, ib: int, jb: int,
            descb):
    """dlaf_mi355x_p{s,d,c,z}hegst: ScaLAPACK's p?sygst / p?hegst argument list; returns (scale, info)."""
    t = type_char(a.dtype)
    da = (C.c_int * 9)(*[int(x) for x in desca])
    db = (C.c_int * 9)(*[int(x) for x in descb])
    scale = (C.c_float if t in "sc" else C.c_double)(-1)
    info = C.c_int(-999)
    getattr(lib(), f"dlaf_mi355x_p{t}hegst")(ibtype, uplo.encode(), n, _ptr(a), ia, ja, da, _ptr(b), ib, jb, db,
                                             C.byref(scale), C.byref(info))
    return scale.value, info.value


def solver_profile():
    """(ms, flops) of the sweep of the last triangular solve on this process (device time, no staging)."""
    ms, fl = C.c_double(0), C.c_double(0)
    lib().dlaf_mi355x_solver_profile(C.byref(ms), C.byref(fl))
    return ms.value, fl.value


def pxtrsm(side: str, uplo: str, op: str, diag: str, m: int, n: int, alpha, a: np.ndarray, ia: int, ja: int, desca,
           b: np.ndarray, ib: int, jb: int, descb) -> None:
    """dlaf_mi355x_p{s,d,c,z}trsm: ScaLAPACK's p?trsm argument list (9-int descriptors)."""
    t = type_char(b.dtype)
    da = (C.c_int * 9)(*[int(x) for x in desca])
    db = (C.c_int * 9)(*[int(x) for x in descb])
    al = np.array([alpha], dtype=b.dtype)
    getattr(lib(), f"dlaf_mi355x_p{t}trsm")(side.encode(), uplo.encode(), op.encode(), diag.encode(), m, n, _ptr(al),
                                            _ptr(a), ia, ja, da, _ptr(b), ib, jb, db)


def set_random_hermitian_positive_definite(grid: Grid, a: np.ndarray, n: int, nb: int, isrc: int = 0,
                                           jsrc: int = 0, nthreads: int = 0) -> None:
    """matrix::util::set_random_hermitian_positive_definite (include/dlaf/util_matrix.h:498-501) on
    this process's local array."""
    desc = make_descriptor(n, nb, _ld_of(a), isrc, jsrc)
    r = lib().dlaf_mi355x_set_random_hpd(grid.context, type_char(a.dtype).encode(), _ptr(a), desc, nthreads)
    if r != 0:
        raise ValueError(f"dlaf_mi355x_set_random_hpd failed with {r}")


def potrf_trace():
    """Diagnosis hook (DLAF_MI355X_POTRF_TRACE=1): the 32 words the last first-diagonal-tile POTRF of this process
    recorded (kernels_potrf_coop.hip), or None when tracing is off."""
    out = (C.c_ulonglong * 32)()
    return list(out) if lib().dlaf_mi355x_potrf_trace(out) == 0 else None


def release_workspace_pool() -> int:
    """Give the idle blocks of the workspace pool back to the driver (dlaf_mi355x_workspace_pool_release); returns the bytes
    that were held."""
    return int(lib().dlaf_mi355x_workspace_pool_release())


def update_launch_stats():
    """(persistent, exclusive): trailing-update launches of this process so far in persistent form / with exclusive
    compute units."""
    a, b = C.c_long(0), C.c_long(0)
    lib().dlaf_mi355x_update_launch_stats(C.byref(a), C.byref(b))
    return a.value, b.value


class DeviceMatrix:
    """Matrix<T, Device::GPU> of the MI355X build: the local part of a block-cyclic matrix resident
    in HBM in tile layout (reference: matrix/matrix.h:57-357 + MatrixMirror, matrix_mirror.h:137-173).
    A driver uploads once and times `factorize` alone, like miniapp_cholesky.cpp:133-155."""

    def __init__(self, grid: Grid, dtype, uplo: str, n: int, nb: int, isrc: int = 0, jsrc: int = 0):
        self.grid, self.dtype, self.uplo, self.n, self.nb = grid, np.dtype(dtype), uplo, n, nb
        self._h = C.c_void_p()
        desc = make_descriptor(n, nb, 1, isrc, jsrc)
        r = lib().dlaf_mi355x_matrix_create(grid.context, type_char(dtype).encode(), uplo.encode(), desc,
                                            C.byref(self._h))
        if r != 0:
            raise ValueError(f"dlaf_mi355x_matrix_create failed with {r}")

    def upload(self, a: np.ndarray) -> None:
        assert a.dtype == self.dtype
        lib().dlaf_mi355x_matrix_upload(self._h, _ptr(a), _ld_of(a))

    def download(self, a: np.ndarray) -> None:
        assert a.dtype == self.dtype
        lib().dlaf_mi355x_matrix_download(self._h, _ptr(a), _ld_of(a))

    def fetch_tile(self, gi: int, gj: int):
        """Global tile (gi, gj) of the device copy as a dense column-major array, or None when this process
        does not own it (a checker samples a large factor tile by tile instead of downloading it whole)."""
        rows = min(self.nb, self.n - gi * self.nb)
        cols = min(self.nb, self.n - gj * self.nb)
        out = np.zeros((rows, cols), dtype=self.dtype, order="F")
        r = lib().dlaf_mi355x_matrix_fetch_tile(self._h, gi, gj, _ptr(out), max(1, rows))
        if r < 0:
            raise ValueError(f"dlaf_mi355x_matrix_fetch_tile failed with {r}")
        return out if r == 0 else None

    def copy_from(self, other: "DeviceMatrix") -> None:
        r = lib().dlaf_mi355x_matrix_copy(self._h, other._h)
        if r != 0:
            raise ValueError("matrices are not conformable")

    def factorize(self) -> int:
        """dlaf::cholesky_factorization<Backend::GPU, Device::GPU, T> on the resident matrix; blocking."""
        return lib().dlaf_mi355x_cholesky_factorization_device(self._h)

    def generalized_to_standard(self, factor_of_b: "DeviceMatrix") -> int:
        """`self` (Hermitian A, resident) <- inv(L) A inv(L^H) with the resident Cholesky factor of B."""
        return lib().dlaf_mi355x_generalized_to_standard_device(self._h, factor_of_b._h)

    def start(self) -> None:
        lib().dlaf_mi355x_cholesky_start(self._h)

    def wait(self) -> int:
        return lib().dlaf_mi355x_cholesky_wait(self._h)

    def local_info(self) -> int:
        """This process's own device status word of the last factorization, before the grid agreed on one value."""
        return lib().dlaf_mi355x_matrix_local_info(self._h)

    def residual_against(self, factor: "DeviceMatrix"):
        """check_cholesky of the miniapp (miniapp_cholesky.cpp:408-443) on the device.  `self` must hold the
        ORIGINAL matrix and is overwritten with A - L L^H.  Returns (max|A - L L^H|, max|A|) over the grid."""
        d, a = C.c_double(), C.c_double()
        r = lib().dlaf_mi355x_cholesky_residual(self._h, factor._h, C.byref(d), C.byref(a))
        if r != 0:
            raise ValueError("matrices are not conformable")
        return d.value, a.value

    PROFILE_KINDS = {"update_bulk": 0, "update_lookahead": 1, "trsm_panel": 2, "potrf_tile": 3}

    def profile(self, kind: str) -> dict:
        """HIP-event timing of one launch class of the last factorization (after wait())."""
        ms, n, fl, by = C.c_double(), C.c_long(), C.c_double(), C.c_double()
        lib().dlaf_mi355x_matrix_profile(self._h, self.PROFILE_KINDS[kind], C.byref(ms), C.byref(n), C.byref(fl),
                                         C.byref(by))
        return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}

    def trsm_profile(self, reps: int = 5) -> dict:
        """The panel TRSM of step 0 alone on the device (after factorize(); one-process grids): HIP-event time per
        launch and the algorithmic work of one launch."""
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        lib().dlaf_mi355x_matrix_trsm_profile(self._h, reps, C.byref(ms), C.byref(fl), C.byref(by))
        return {"ms": ms.value, "flops": fl.value, "bytes": by.value}

    def close(self) -> None:
        if self._h:
            lib().dlaf_mi355x_matrix_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GeneralDeviceMatrix:
    """A general m x n matrix resident in HBM in tile layout (square blocks): the right-hand sides of a
    device-resident solve (dlaf_mi355x_gmatrix_*)."""

    def __init__(self, grid: Grid, dtype, m: int, n: int, nb: int, isrc: int = 0, jsrc: int = 0):
        self.grid, self.dtype, self.m, self.n, self.nb = grid, np.dtype(dtype), m, n, nb
        self._h = C.c_void_p()
        desc = DLAFDescriptor(m, n, nb, nb, isrc, jsrc, 0, 0, 1)
        r = lib().dlaf_mi355x_gmatrix_create(grid.context, type_char(dtype).encode(), desc, C.byref(self._h))
        if r != 0:
            raise ValueError(f"dlaf_mi355x_gmatrix_create failed with {r}")

    def upload(self, a: np.ndarray) -> None:
        assert a.dtype == self.dtype
        lib().dlaf_mi355x_gmatrix_upload(self._h, _ptr(a), _ld_of(a))

    def download(self, a: np.ndarray) -> None:
        assert a.dtype == self.dtype
        lib().dlaf_mi355x_gmatrix_download(self._h, _ptr(a), _ld_of(a))

    def close(self) -> None:
        if self._h:
            lib().dlaf_mi355x_gmatrix_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def triangular_solver_device(side: str, uplo: str, op: str, diag: str, alpha, a: DeviceMatrix, b: GeneralDeviceMatrix) -> None:
    """dlaf::triangular_solver on resident operands: `a` holds the triangular matrix in its uplo triangle (e.g. the
    factor a.factorize() left there), `b` is overwritten by the solution; no PCIe traffic."""
    al = np.array([alpha], dtype=b.dtype)
    r = lib().dlaf_mi355x_triangular_solver_device(side.encode(), uplo.encode(), op.encode(), diag.encode(), _ptr(al),
                                                   a._h, b._h)
    if r != 0:
        raise ValueError(f"dlaf_mi355x_triangular_solver_device failed with {r}")


def potrs_device(uplo: str, factor: DeviceMatrix, b: GeneralDeviceMatrix) -> None:
    """A X = B from the resident Cholesky factor (the two solves of p?potrs, all in HBM)."""
    r = lib().dlaf_mi355x_potrs_device(uplo.encode(), factor._h, b._h)
    if r != 0:
        raise ValueError(f"dlaf_mi355x_potrs_device failed with {r}")


# ---- tile operations with the argument sets the factorization issues ----------------------------
def tile_potrf(uplo: str, a: np.ndarray) -> int:
    """tile::potrf (include/dlaf/lapack/tile.h:362-378); returns info."""
    return lib().dlaf_mi355x_tile_potrf(type_char(a.dtype).encode(), uplo.encode(), a.shape[0], _ptr(a), _ld_of(a))


def tile_trsm(uplo: str, a: np.ndarray, b: np.ndarray) -> None:
    """tile::trsm as cholesky/impl.h:56-67 (L: B <- B A^-H) / :110-121 (U: B <- A^-H B) call it."""
    m, n = b.shape
    lib().dlaf_mi355x_tile_trsm(type_char(b.dtype).encode(), uplo.encode(), m, n, _ptr(a), _ld_of(a), _ptr(b),
                                _ld_of(b))


def tile_herk(uplo: str, a: np.ndarray, c: np.ndarray) -> None:
    """tile::herk as impl.h:70-80 (L: C -= A A^H) / :124-134 (U: C -= A^H A) call it."""
    n = c.shape[0]
    k = a.shape[1] if uplo in "Ll" else a.shape[0]
    lib().dlaf_mi355x_tile_herk(type_char(c.dtype).encode(), uplo.encode(), n, k, _ptr(a), _ld_of(a), _ptr(c),
                                _ld_of(c))


def tile_gemm(uplo: str, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> None:
    """tile::gemm as impl.h:83-94 (L: C -= A B^H) / :137-147 (U: C -= A^H B) call it."""
    m, n = c.shape
    k = a.shape[1] if uplo in "Ll" else a.shape[0]
    lib().dlaf_mi355x_tile_gemm(type_char(c.dtype).encode(), uplo.encode(), m, n, k, _ptr(a), _ld_of(a), _ptr(b),
                                _ld_of(b), _ptr(c), _ld_of(c))
