"""dla_future_amd -- MI355X-native tiled distributed Cholesky behind the DLA-Future interface.

The product is the C-ABI shared library ``dla_future_amd/lib/libdlaf_mi355x.so`` (hand-written
gfx950 HIP kernels + a thin C++ host over HIP streams/events and RCCL).  This package is the
Python doorway onto that ABI, mirroring the reference's operator surface for the path:

    dlaf_initialize / dlaf_finalize                    include/dlaf_c/init.h
    dlaf_create_grid* / dlaf_free_grid                 include/dlaf_c/grid.h
    dlaf_cholesky_factorization_{s,d,c,z}              include/dlaf_c/factorization/cholesky.h:32-47
    dlaf_p{s,d,c,z}potrf                               include/dlaf_c/factorization/cholesky.h:74-87
    dlaf::cholesky_factorization(uplo, Matrix&)        include/dlaf/factorization/cholesky.h:39-79
    dlaf::triangular_solver(side, uplo, op, diag, ...)  include/dlaf/solver/triangular.h:41-177 (next row)
    generalized_to_standard(grid, uplo, A, B)          include/dlaf/eigensolver/gen_to_std.h:50,:101
    reduction_to_band / bt_reduction_to_band           include/dlaf/eigensolver/reduction_to_band.h:40-122, bt_reduction_to_band.h

There is no CPU fallback: importing works anywhere, every compute call needs the HIP library
and a GPU and fails loudly otherwise.
"""
from .capi import (DLAFDescriptor, LibraryNotBuilt, lib, lib_path, type_char, version)  # noqa: F401
from .cholesky import (DeviceMatrix, GeneralDeviceMatrix, Grid, cholesky_factorization, finalize, generalized_to_standard,  # noqa: F401
                       initialize, make_descriptor, potrf_trace, pxhegst, pxpotrf, pxpotrs, pxtrsm, set_random_hermitian_positive_definite, tile_gemm, tile_herk, tile_potrf,
                       tile_trsm, triangular_solver, triangular_solver_device, potrs_device, release_workspace_pool, solver_profile,
                       update_launch_stats)
from . import distribution  # noqa: F401
from .eigensolver import (band_to_tridiagonal, bt_band_to_tridiagonal, bt_reduction_to_band,  # noqa: F401
                          bt_reduction_to_band_device, eigensolver_min_band, eigensolver_profile, get_band_size, hermitian_eigensolver,
                          hermitian_generalized_eigensolver, red2band_panel_stats, red2band_profile, reduction_to_band,
                          reduction_to_band_device, tridiagonal_eigensolver)

__all__ = ["band_to_tridiagonal", "bt_band_to_tridiagonal", "eigensolver_profile", "hermitian_eigensolver",
           "hermitian_generalized_eigensolver", "tridiagonal_eigensolver", "bt_reduction_to_band", "bt_reduction_to_band_device", "get_band_size", "eigensolver_min_band", "red2band_profile",
           "reduction_to_band", "reduction_to_band_device", "DLAFDescriptor", "DeviceMatrix", "GeneralDeviceMatrix", "Grid", "LibraryNotBuilt", "cholesky_factorization", "distribution",
           "finalize", "generalized_to_standard", "initialize", "lib", "lib_path", "make_descriptor", "pxhegst", "pxpotrf", "pxpotrs", "pxtrsm",
           "set_random_hermitian_positive_definite", "solver_profile", "tile_gemm", "tile_herk", "tile_potrf", "tile_trsm",
           "triangular_solver", "triangular_solver_device", "potrs_device", "type_char", "version"]
