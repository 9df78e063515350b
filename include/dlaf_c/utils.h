/* dlaf_c/utils.h -- language glue and the ScaLAPACK-descriptor helper.
 * Drop-in for the reference's include/dlaf_c/utils.h:13-45. */
#pragma once

#ifdef __cplusplus
#define DLAF_EXTERN_C extern "C"
#define DLAF_NOEXCEPT noexcept
#include <complex>
using dlaf_complex_c = std::complex<float>;
using dlaf_complex_z = std::complex<double>;
#else
#define DLAF_EXTERN_C
#define DLAF_NOEXCEPT
#include <complex.h>
typedef float complex dlaf_complex_c;
typedef double complex dlaf_complex_z;
#endif

#include <dlaf_c/desc.h>

/* ScaLAPACK desc[9] = {dtype, ctxt, M, N, MB, NB, RSRC, CSRC, LLD} -> DLAF_descriptor for the
 * m x n sub-matrix starting at (i, j), 1-based; only i == j == 1 is accepted
 * (reference: include/dlaf_c/utils.h:43, src/c_api/utils.cpp:25-33). */
DLAF_EXTERN_C struct DLAF_descriptor make_dlaf_descriptor(const int m, const int n, const int i, const int j,
                                                          const int desc[9]) DLAF_NOEXCEPT;
