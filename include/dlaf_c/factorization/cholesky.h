/* dlaf_c/factorization/cholesky.h -- Cholesky factorization, C interface.
 * Drop-in for the reference's include/dlaf_c/factorization/cholesky.h:32-87
 * (src/c_api/factorization/cholesky.h:33-75, cholesky.cpp).
 *
 * `a` is this process's local part of the 2-D block-cyclic matrix: HOST memory, column-major,
 * leading dimension desc.ld; only the `uplo` triangle is read and overwritten with the factor,
 * the other triangle is left untouched.  The call is collective over the grid and blocking.
 * Upload to HBM, factorization on the GPU and download happen inside the call.
 *
 * Return value / info: 0 on success.  Improvement over upstream (which aborts on a non-SPD
 * matrix): a non-positive pivot returns the LAPACK-style index k > 0 of the failing leading
 * minor, the content of the triangle is then unspecified. */
#pragma once
#include <dlaf_c/desc.h>
#include <dlaf_c/utils.h>

/* reference: cholesky.h:32-47 */
DLAF_EXTERN_C int dlaf_cholesky_factorization_s(const int dlaf_context, const char uplo, float* a,
                                                const struct DLAF_descriptor dlaf_desca) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_cholesky_factorization_d(const int dlaf_context, const char uplo, double* a,
                                                const struct DLAF_descriptor dlaf_desca) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_cholesky_factorization_c(const int dlaf_context, const char uplo, dlaf_complex_c* a,
                                                const struct DLAF_descriptor dlaf_desca) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_cholesky_factorization_z(const int dlaf_context, const char uplo, dlaf_complex_z* a,
                                                const struct DLAF_descriptor dlaf_desca) DLAF_NOEXCEPT;

/* ScaLAPACK-style entry points, reference: cholesky.h:74-87.  desca[1] is the context
 * (a DLA-Future context from dlaf_create_grid* in this build), ia == ja == 1, desca[0] == 1. */
DLAF_EXTERN_C void dlaf_pspotrf(const char uplo, const int n, float* a, const int ia, const int ja,
                                const int desca[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pdpotrf(const char uplo, const int n, double* a, const int ia, const int ja,
                                const int desca[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pcpotrf(const char uplo, const int n, dlaf_complex_c* a, const int ia, const int ja,
                                const int desca[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pzpotrf(const char uplo, const int n, dlaf_complex_z* a, const int ia, const int ja,
                                const int desca[9], int* info) DLAF_NOEXCEPT;
