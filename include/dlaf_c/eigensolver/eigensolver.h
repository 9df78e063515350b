/* dlaf_c/eigensolver/eigensolver.h -- Hermitian (symmetric) eigensolver, C interface.
 * Drop-in for the reference's include/dlaf_c/eigensolver/eigensolver.h:39-134 (src/c_api/eigensolver/eigensolver.h:33-124).
 *
 * a: this process's local part of the Hermitian matrix (HOST, column-major, ld = desca.ld); only the `uplo` triangle is
 * referenced, and it is destroyed (like upstream it ends up holding the band matrix and the Householder reflectors of the
 * reduction to band).  w: all n eigenvalues, ascending, on every process.  z: local part of the n x n eigenvector matrix
 * (column i belongs to w[i]), distributed as descz says (same block size and row source as a).  uplo must be 'L'
 * (upstream: DLAF_UNIMPLEMENTED for 'U', eigensolver/impl.h:43-45).  Collective over the grid and blocking; all stages
 * run on the GPU.  Returns 0. */
#pragma once
#include <dlaf_c/desc.h>
#include <dlaf_c/utils.h>

/* reference: eigensolver.h:39-58 */
DLAF_EXTERN_C int dlaf_symmetric_eigensolver_s(const int dlaf_context, const char uplo, float* a,
        const struct DLAF_descriptor dlaf_desca, float* w, float* z, const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_symmetric_eigensolver_d(const int dlaf_context, const char uplo, double* a,
        const struct DLAF_descriptor dlaf_desca, double* w, double* z, const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_hermitian_eigensolver_c(const int dlaf_context, const char uplo, dlaf_complex_c* a,
        const struct DLAF_descriptor dlaf_desca, float* w, dlaf_complex_c* z, const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_hermitian_eigensolver_z(const int dlaf_context, const char uplo, dlaf_complex_z* a,
        const struct DLAF_descriptor dlaf_desca, double* w, dlaf_complex_z* z, const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;

/* ScaLAPACK-style entry points, reference: eigensolver.h:117-134 (p?syevd / p?heevd argument order without the workspace
 * arguments).  desca[1] is the context, ia == ja == iz == jz == 1. */
DLAF_EXTERN_C void dlaf_pssyevd(const char uplo, const int n, float* a, const int ia, const int ja, const int desca[9],
        float* w, float* z, const int iz, const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pdsyevd(const char uplo, const int n, double* a, const int ia, const int ja, const int desca[9],
        double* w, double* z, const int iz, const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pcheevd(const char uplo, const int n, dlaf_complex_c* a, const int ia, const int ja, const int desca[9],
        float* w, dlaf_complex_c* z, const int iz, const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pzheevd(const char uplo, const int n, dlaf_complex_z* a, const int ia, const int ja, const int desca[9],
        double* w, dlaf_complex_z* z, const int iz, const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
