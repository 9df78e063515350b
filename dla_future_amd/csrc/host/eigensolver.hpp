// eigensolver.hpp -- the stages of the Hermitian eigensolver behind reduction_to_band (eigensolver.cpp, tridiag_dc.cpp):
// band -> tridiagonal, tridiagonal eigensolver, back-transformation band <- tridiagonal, and the drivers
// (include/dlaf/eigensolver/eigensolver/impl.h:38-55).
#pragma once
#include "red2band.hpp"
#include "runtime.hpp"
#include "tile_matrix.hpp"

namespace dlaf_mi355x {

// band_to_tridiagonal (include/dlaf/eigensolver/band_to_tridiag.h:74-97, :155-176): A (uplo L, tile layout) holds a
// Hermitian band matrix in its lower band (what reduction_to_band leaves).  Device outputs: d, e (n reals each, e[n-1]
// = 0), v (n x n, ldv): the compact Householder reflectors with tau in the place of the leading 1.  On a process grid
// every rank ends up with the whole result (the band is summed over the grid, the chase runs replicated).
template <class T>
int band_to_tridiag_device(DeviceMatrix<T>& a, int band, real_t<T>* d, real_t<T>* e, T* v, long ldv);
template <class T>
int band_to_tridiag_host(Grid* g, const T* a, long lda, long n, int nb, int isrc, int jsrc, int band, real_t<T>* d,
                         real_t<T>* e, T* v, long ldv);

// bt_band_to_tridiagonal (include/dlaf/eigensolver/bt_band_to_tridiag.h:28-61): E <- Q E on the rows of a
// column-major device array e (n x ncols, lde); v as band_to_tridiag_device left it.
template <class T>
int bt_band_to_tridiag_device(long n, int band, const T* v, long ldv, T* e, long lde, long ncols, hipStream_t s);
template <class T>
int bt_band_to_tridiag_host(long n, int band, const T* v, long ldv, T* e, long lde, long ncols);

// tridiagonal_eigensolver (include/dlaf/eigensolver/tridiag_solver.h:30-60): d, e (device, n; e[n-1] unused) ->
// eigenvalues w (device, n, ascending) and eigenvectors z (device, column-major n x n, ldz).  nb = the leaf size of the
// divide & conquer tree (the block size of the reference's distribution, tridiag_solver/impl.h:198-262).
template <class R>
int tridiag_solver_device(long n, int nb, R* d, R* e, R* w, R* z, long ldz, hipStream_t s, Transport* tr = nullptr);
template <class R>
int tridiag_solver_host(long n, int nb, const R* d, const R* e, R* w, R* z, long ldz);

// Hermitian eigensolver, Eigensolver::call (eigensolver/impl.h:38-55, :57-95): the local parts of A (lower triangle
// referenced, destroyed), eigenvalues w (all n on every rank), eigenvectors z (distributed like a general n x n matrix
// with A's block size and z's own source rank)
template <class T>
int hermitian_eigensolver_host(Grid* g, char uplo, T* a, long lda, long n, int nb, int isrc, int jsrc, real_t<T>* w, T* z,
                               long ldz, int z_isrc, int z_jsrc);
// Hermitian generalized eigensolver A x = lambda B x, GenEigensolver::call (gen_eigensolver/impl.h:33-92)
template <class T>
int hermitian_gen_eigensolver_host(Grid* g, char uplo, T* a, long lda, T* b, long ldb, long n, int nb, int a_isrc,
                                   int a_jsrc, int b_isrc, int b_jsrc, real_t<T>* w, T* z, long ldz, int z_isrc,
                                   int z_jsrc, bool b_factorized);

// per-stage device times (ms) of the last eigensolver call on this process:
// 0 reduction_to_band, 1 band_to_tridiagonal, 2 tridiagonal solver, 3 bt_band_to_tridiagonal, 4 bt_reduction_to_band
void eigensolver_last_profile(double ms[5]);

}  // namespace dlaf_mi355x
