"""ctypes binding of libdlaf_mi355x.so (include/dlaf_c/*.h + include/dlaf_mi355x/dlaf_mi355x.h)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class LibraryNotBuilt(RuntimeError):
    pass


def lib_path() -> str:
    # DLAF_MI355X_LIB: another build of the same library (diagnosis builds, e.g. tools/build_b2t_prof.sh)
    return os.environ.get("DLAF_MI355X_LIB") or os.path.join(_HERE, "lib", "libdlaf_mi355x.so")


class DLAFDescriptor(C.Structure):
    """struct DLAF_descriptor (include/dlaf_c/desc.h; reference desc.h:16-26)."""
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("mb", C.c_int), ("nb", C.c_int), ("isrc", C.c_int),
                ("jsrc", C.c_int), ("i", C.c_int), ("j", C.c_int), ("ld", C.c_int)]


BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t)
BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)

_TYPE_CHARS = {np.dtype(np.float32): "s", np.dtype(np.float64): "d", np.dtype(np.complex64): "c",
               np.dtype(np.complex128): "z"}


def type_char(dtype) -> str:
    try:
        return _TYPE_CHARS[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"unsupported element type {dtype}: expected float32/64 or complex64/128") from None


# every symbol the headers declare: name -> (restype, argtypes)
_vp, _i, _l, _ch = C.c_void_p, C.c_int, C.c_long, C.c_char
_IP = C.POINTER(C.c_int)
SIGNATURES = {
    # include/dlaf_c/init.h
    "dlaf_initialize": (None, [_i, C.POINTER(C.c_char_p), _i, C.POINTER(C.c_char_p)]),
    "dlaf_finalize": (None, []),
    # include/dlaf_c/grid.h
    "dlaf_free_grid": (None, [_i]),
    # include/dlaf_c/utils.h
    "make_dlaf_descriptor": (DLAFDescriptor, [_i, _i, _i, _i, _IP]),
    # include/dlaf_c/factorization/cholesky.h
    "dlaf_cholesky_factorization_s": (_i, [_i, _ch, _vp, DLAFDescriptor]),
    "dlaf_cholesky_factorization_d": (_i, [_i, _ch, _vp, DLAFDescriptor]),
    "dlaf_cholesky_factorization_c": (_i, [_i, _ch, _vp, DLAFDescriptor]),
    "dlaf_cholesky_factorization_z": (_i, [_i, _ch, _vp, DLAFDescriptor]),
    "dlaf_pspotrf": (None, [_ch, _i, _vp, _i, _i, _IP, _IP]),
    "dlaf_pdpotrf": (None, [_ch, _i, _vp, _i, _i, _IP, _IP]),
    "dlaf_pcpotrf": (None, [_ch, _i, _vp, _i, _i, _IP, _IP]),
    "dlaf_pzpotrf": (None, [_ch, _i, _vp, _i, _i, _IP, _IP]),
    # include/dlaf_mi355x/dlaf_mi355x.h
    "dlaf_mi355x_version": (C.c_char_p, []),
    "dlaf_mi355x_create_grid_single": (_i, []),
    "dlaf_mi355x_rccl_unique_id": (None, [_vp]),
    "dlaf_mi355x_create_grid_rccl": (_i, [_vp, _i, _i, _i, _i, _ch]),
    "dlaf_mi355x_create_grid_host": (_i, [_i, _i, _i, _i, _ch, BCAST_FN, BARRIER_FN, _vp]),
    "dlaf_mi355x_grid_host_bcast": (_i, [_i, _i, _i, _vp, C.c_size_t]),
    "dlaf_mi355x_grid_info": (_i, [_i, _IP, _IP, _IP, _IP]),
    "dlaf_mi355x_grid_barrier": (_i, [_i]),
    "dlaf_mi355x_grid_selftest": (_i, [_i, C.c_size_t]),
    "dlaf_mi355x_matrix_create": (_i, [_i, _ch, _ch, DLAFDescriptor, C.POINTER(_vp)]),
    "dlaf_mi355x_matrix_destroy": (None, [_vp]),
    "dlaf_mi355x_matrix_upload": (_i, [_vp, _vp, _i]),
    "dlaf_mi355x_matrix_download": (_i, [_vp, _vp, _i]),
    "dlaf_mi355x_matrix_copy": (_i, [_vp, _vp]),
    "dlaf_mi355x_matrix_fetch_tile": (_i, [_vp, _l, _l, _vp, _i]),
    "dlaf_mi355x_grid_on_free": (_i, [_i, _vp, _vp]),
    "dlaf_mi355x_grid_rekey": (_i, [_i, _i]),
    "dlaf_mi355x_grid_comm_log": (_i, [_i, _i]),
    "dlaf_mi355x_grid_comm_log_read": (_l, [_i, C.POINTER(C.c_long), _l]),
    "dlaf_mi355x_matrix_local_info": (_i, [_vp]),
    "dlaf_mi355x_potrf_trace": (_i, [C.POINTER(C.c_ulonglong)]),
    "dlaf_mi355x_update_launch_stats": (_i, [C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "dlaf_mi355x_cholesky_start": (_i, [_vp]),
    "dlaf_mi355x_cholesky_wait": (_i, [_vp]),
    "dlaf_mi355x_cholesky_factorization_device": (_i, [_vp]),
    "dlaf_mi355x_cholesky_residual": (_i, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dlaf_mi355x_matrix_trsm_profile": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dlaf_mi355x_matrix_profile": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_long),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dlaf_mi355x_triangular_solver_s": (_i, [_i, _ch, _ch, _ch, _ch, _vp, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_triangular_solver_d": (_i, [_i, _ch, _ch, _ch, _ch, _vp, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_triangular_solver_c": (_i, [_i, _ch, _ch, _ch, _ch, _vp, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_triangular_solver_z": (_i, [_i, _ch, _ch, _ch, _ch, _vp, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_pstrsm": (None, [_ch, _ch, _ch, _ch, _i, _i, _vp, _vp, _i, _i, _IP, _vp, _i, _i, _IP]),
    "dlaf_mi355x_pdtrsm": (None, [_ch, _ch, _ch, _ch, _i, _i, _vp, _vp, _i, _i, _IP, _vp, _i, _i, _IP]),
    "dlaf_mi355x_pctrsm": (None, [_ch, _ch, _ch, _ch, _i, _i, _vp, _vp, _i, _i, _IP, _vp, _i, _i, _IP]),
    "dlaf_mi355x_pztrsm": (None, [_ch, _ch, _ch, _ch, _i, _i, _vp, _vp, _i, _i, _IP, _vp, _i, _i, _IP]),
    "dlaf_mi355x_pspotrs": (None, [_ch, _i, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _IP]),
    "dlaf_mi355x_pdpotrs": (None, [_ch, _i, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _IP]),
    "dlaf_mi355x_pcpotrs": (None, [_ch, _i, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _IP]),
    "dlaf_mi355x_pzpotrs": (None, [_ch, _i, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _IP]),
    "dlaf_mi355x_solver_profile": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dlaf_mi355x_gmatrix_create": (_i, [_i, _ch, DLAFDescriptor, C.POINTER(_vp)]),
    "dlaf_mi355x_gmatrix_destroy": (None, [_vp]),
    "dlaf_mi355x_gmatrix_upload": (_i, [_vp, _vp, _i]),
    "dlaf_mi355x_gmatrix_download": (_i, [_vp, _vp, _i]),
    "dlaf_mi355x_triangular_solver_device": (_i, [_ch, _ch, _ch, _ch, _vp, _vp, _vp]),
    "dlaf_mi355x_potrs_device": (_i, [_ch, _vp, _vp]),
    "dlaf_mi355x_generalized_to_standard_s": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_generalized_to_standard_d": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_generalized_to_standard_c": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_generalized_to_standard_z": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor]),
    "dlaf_mi355x_pshegst": (None, [_i, _ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _IP]),
    "dlaf_mi355x_pdhegst": (None, [_i, _ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _IP]),
    "dlaf_mi355x_pchegst": (None, [_i, _ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _IP]),
    "dlaf_mi355x_pzhegst": (None, [_i, _ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _IP]),
    "dlaf_mi355x_generalized_to_standard_device": (_i, [_vp, _vp]),
    "dlaf_mi355x_reduction_to_band_s": (_i, [_i, _vp, DLAFDescriptor, _i, _vp]),
    "dlaf_mi355x_reduction_to_band_d": (_i, [_i, _vp, DLAFDescriptor, _i, _vp]),
    "dlaf_mi355x_reduction_to_band_c": (_i, [_i, _vp, DLAFDescriptor, _i, _vp]),
    "dlaf_mi355x_reduction_to_band_z": (_i, [_i, _vp, DLAFDescriptor, _i, _vp]),
    "dlaf_mi355x_reduction_to_band_device": (_i, [_vp, _i, _vp]),
    "dlaf_mi355x_bt_reduction_to_band_s": (_i, [_i, _i, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp]),
    "dlaf_mi355x_bt_reduction_to_band_d": (_i, [_i, _i, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp]),
    "dlaf_mi355x_bt_reduction_to_band_c": (_i, [_i, _i, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp]),
    "dlaf_mi355x_bt_reduction_to_band_z": (_i, [_i, _i, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp]),
    "dlaf_mi355x_bt_reduction_to_band_device": (_i, [_i, _vp, _vp, _vp]),
    "dlaf_mi355x_band_to_tridiagonal_s": (_i, [_i, _vp, DLAFDescriptor, _i, _vp, _vp, _vp, _i]),
    "dlaf_mi355x_bt_band_to_tridiagonal_s": (_i, [_i, _i, _i, _vp, _i, _vp, _i]),
    "dlaf_mi355x_band_to_tridiagonal_d": (_i, [_i, _vp, DLAFDescriptor, _i, _vp, _vp, _vp, _i]),
    "dlaf_mi355x_bt_band_to_tridiagonal_d": (_i, [_i, _i, _i, _vp, _i, _vp, _i]),
    "dlaf_mi355x_band_to_tridiagonal_c": (_i, [_i, _vp, DLAFDescriptor, _i, _vp, _vp, _vp, _i]),
    "dlaf_mi355x_bt_band_to_tridiagonal_c": (_i, [_i, _i, _i, _vp, _i, _vp, _i]),
    "dlaf_mi355x_band_to_tridiagonal_z": (_i, [_i, _vp, DLAFDescriptor, _i, _vp, _vp, _vp, _i]),
    "dlaf_mi355x_bt_band_to_tridiagonal_z": (_i, [_i, _i, _i, _vp, _i, _vp, _i]),
    "dlaf_symmetric_eigensolver_s": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_symmetric_generalized_eigensolver_s": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_symmetric_generalized_eigensolver_factorized_s": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_symmetric_eigensolver_d": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_symmetric_generalized_eigensolver_d": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_symmetric_generalized_eigensolver_factorized_d": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_hermitian_eigensolver_c": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_hermitian_generalized_eigensolver_c": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_hermitian_generalized_eigensolver_factorized_c": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_hermitian_eigensolver_z": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_hermitian_generalized_eigensolver_z": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_hermitian_generalized_eigensolver_factorized_z": (_i, [_i, _ch, _vp, DLAFDescriptor, _vp, DLAFDescriptor, _vp, _vp, DLAFDescriptor]),
    "dlaf_pssyevd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pdsyevd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pcheevd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pzheevd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pssygvd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pssygvd_factorized": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pdsygvd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pdsygvd_factorized": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pchegvd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pchegvd_factorized": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pzhegvd": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_pzhegvd_factorized": (None, [_ch, _i, _vp, _i, _i, _IP, _vp, _i, _i, _IP, _vp, _vp, _i, _i, _IP, _IP]),
    "dlaf_mi355x_tridiagonal_eigensolver_s": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i]),
    "dlaf_mi355x_tridiagonal_eigensolver_d": (_i, [_i, _i, _vp, _vp, _vp, _vp, _i]),
    "dlaf_mi355x_eigensolver_profile": (_i, [C.POINTER(C.c_double)]),
    "dlaf_mi355x_get_band_size": (_i, [_i]),
    "dlaf_mi355x_red2band_panel_stats": (_i, [C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "dlaf_mi355x_workspace_pool_release": (C.c_long, []),
    "dlaf_mi355x_get_eigensolver_min_band": (_i, []),
    "dlaf_mi355x_set_eigensolver_min_band": (None, [_i]),
    "dlaf_mi355x_red2band_profile": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dlaf_mi355x_set_random_hpd": (_i, [_i, _ch, _vp, DLAFDescriptor, _i]),
    "dlaf_mi355x_tile_potrf": (_i, [_ch, _ch, _i, _vp, _i]),
    "dlaf_mi355x_tile_trsm": (_i, [_ch, _ch, _i, _i, _vp, _i, _vp, _i]),
    "dlaf_mi355x_tile_herk": (_i, [_ch, _ch, _i, _i, _vp, _i, _vp, _i]),
    "dlaf_mi355x_tile_gemm": (_i, [_ch, _ch, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i]),
    "dlaf_mi355x_dist_owner": (_i, [_l, _i, _i]),
    "dlaf_mi355x_dist_local_tile": (_l, [_l, _i, _i, _i]),
    "dlaf_mi355x_dist_next_local_tile": (_l, [_l, _i, _i, _i]),
    "dlaf_mi355x_dist_global_tile": (_l, [_l, _i, _i, _i]),
    "dlaf_mi355x_dist_local_size": (_l, [_l, _i, _i, _i, _i]),
    "dlaf_mi355x_dist_local_tiles": (_l, [_l, _i, _i, _i, _i]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libdlaf_mi355x.so and type every entry point.  Raises LibraryNotBuilt when the HIP
    library is missing: there is deliberately no other implementation to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64 / librccl (same SONAMEs as /opt/rocm's).  Two copies in one
    # process abort at exit, so when torch is installed it must be loaded FIRST: our library then binds
    # to the runtime torch brought in.  DLAF_MI355X_NO_TORCH=1 skips this (pure C / numpy users).
    if os.environ.get("DLAF_MI355X_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise LibraryNotBuilt(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); dla_future_amd has no CPU fallback")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)  # AttributeError here = header/library drift, which must be loud
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def version() -> str:
    return lib().dlaf_mi355x_version().decode()
