/* dlaf_c/eigensolver/gen_eigensolver.h -- Hermitian generalized eigensolver A x = lambda B x, C interface.
 * Drop-in for the reference's include/dlaf_c/eigensolver/gen_eigensolver.h:44-242
 * (src/c_api/eigensolver/gen_eigensolver.h).
 *
 * a, b: local parts of the Hermitian matrix A and the Hermitian positive definite matrix B (HOST, column-major); only the
 * `uplo` triangles are referenced.  On return b holds its Cholesky factor (the `_factorized` entries take the factor in
 * b and leave it alone), a is destroyed.  w: all n eigenvalues, ascending; z: local part of the eigenvector matrix,
 * B-orthonormal (Z^H B Z = I).  uplo must be 'L' (as upstream).  Stages: dlaf_cholesky_factorization, generalized_to_standard,
 * the eigensolver, one triangular solve (gen_eigensolver/impl.h:33-92).  Returns 0 (the LAPACK info of the
 * factorization when B is not positive definite; upstream aborts). */
#pragma once
#include <dlaf_c/desc.h>
#include <dlaf_c/utils.h>

/* reference: gen_eigensolver.h:44-69, :110-135 */
DLAF_EXTERN_C int dlaf_symmetric_generalized_eigensolver_s(const int dlaf_context, const char uplo, float* a,
        const struct DLAF_descriptor dlaf_desca, float* b, const struct DLAF_descriptor dlaf_descb, float* w, float* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_symmetric_generalized_eigensolver_d(const int dlaf_context, const char uplo, double* a,
        const struct DLAF_descriptor dlaf_desca, double* b, const struct DLAF_descriptor dlaf_descb, double* w, double* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_hermitian_generalized_eigensolver_c(const int dlaf_context, const char uplo, dlaf_complex_c* a,
        const struct DLAF_descriptor dlaf_desca, dlaf_complex_c* b, const struct DLAF_descriptor dlaf_descb, float* w, dlaf_complex_c* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_hermitian_generalized_eigensolver_z(const int dlaf_context, const char uplo, dlaf_complex_z* a,
        const struct DLAF_descriptor dlaf_desca, dlaf_complex_z* b, const struct DLAF_descriptor dlaf_descb, double* w, dlaf_complex_z* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_symmetric_generalized_eigensolver_factorized_s(const int dlaf_context, const char uplo, float* a,
        const struct DLAF_descriptor dlaf_desca, float* b, const struct DLAF_descriptor dlaf_descb, float* w, float* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_symmetric_generalized_eigensolver_factorized_d(const int dlaf_context, const char uplo, double* a,
        const struct DLAF_descriptor dlaf_desca, double* b, const struct DLAF_descriptor dlaf_descb, double* w, double* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_hermitian_generalized_eigensolver_factorized_c(const int dlaf_context, const char uplo, dlaf_complex_c* a,
        const struct DLAF_descriptor dlaf_desca, dlaf_complex_c* b, const struct DLAF_descriptor dlaf_descb, float* w, dlaf_complex_c* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_hermitian_generalized_eigensolver_factorized_z(const int dlaf_context, const char uplo, dlaf_complex_z* a,
        const struct DLAF_descriptor dlaf_desca, dlaf_complex_z* b, const struct DLAF_descriptor dlaf_descb, double* w, dlaf_complex_z* z,
        const struct DLAF_descriptor dlaf_descz) DLAF_NOEXCEPT;

/* ScaLAPACK-style entry points, reference: gen_eigensolver.h:185-242 */
DLAF_EXTERN_C void dlaf_pssygvd(const char uplo, const int n, float* a, const int ia, const int ja,
        const int desca[9], float* b, const int ib, const int jb, const int descb[9], float* w, float* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pdsygvd(const char uplo, const int n, double* a, const int ia, const int ja,
        const int desca[9], double* b, const int ib, const int jb, const int descb[9], double* w, double* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pchegvd(const char uplo, const int n, dlaf_complex_c* a, const int ia, const int ja,
        const int desca[9], dlaf_complex_c* b, const int ib, const int jb, const int descb[9], float* w, dlaf_complex_c* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pzhegvd(const char uplo, const int n, dlaf_complex_z* a, const int ia, const int ja,
        const int desca[9], dlaf_complex_z* b, const int ib, const int jb, const int descb[9], double* w, dlaf_complex_z* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pssygvd_factorized(const char uplo, const int n, float* a, const int ia, const int ja,
        const int desca[9], float* b, const int ib, const int jb, const int descb[9], float* w, float* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pdsygvd_factorized(const char uplo, const int n, double* a, const int ia, const int ja,
        const int desca[9], double* b, const int ib, const int jb, const int descb[9], double* w, double* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pchegvd_factorized(const char uplo, const int n, dlaf_complex_c* a, const int ia, const int ja,
        const int desca[9], dlaf_complex_c* b, const int ib, const int jb, const int descb[9], float* w, dlaf_complex_c* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_pzhegvd_factorized(const char uplo, const int n, dlaf_complex_z* a, const int ia, const int ja,
        const int desca[9], dlaf_complex_z* b, const int ib, const int jb, const int descb[9], double* w, dlaf_complex_z* z, const int iz,
        const int jz, const int descz[9], int* info) DLAF_NOEXCEPT;
