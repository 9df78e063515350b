// red2band.hpp -- reduction to band + back-transformation (red2band.cpp), SURVEY.md section 8(f) item 4
#pragma once
#include "runtime.hpp"
#include "tile_matrix.hpp"

namespace dlaf_mi355x {

template <class T>
inline T make_host_el(double re) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{(real_t<T>) re, (real_t<T>) 0};
  else
    return (T) re;
}

// eigensolver/internal/get_band_size.h:20-31 with the tune parameter eigensolver_min_band (tune.h:128, default 100)
int get_band_size(int nb);
// panels of this process's last reduction_to_band that took the blocked factorization (CholeskyQR2 + Householder
// reconstruction) / that it handed back to the reflector-by-reflector kernel
void red2band_last_panels(long* blocked, long* fallback);
int eigensolver_min_band();
void set_eigensolver_min_band(int b_min);

// A (uplo L, tile layout) <- band + reflectors; taus_host: n - band - 1 values (all of them on every rank), may be null
template <class T>
int reduction_to_band_device(DeviceMatrix<T>& a, int band, T* taus_host);
template <class T>
int reduction_to_band_host(Grid* g, T* a, long lda, long n, int nb, int isrc, int jsrc, int band, T* taus);

// C <- Q C
template <class T>
int bt_reduction_to_band_device(int band, TileMatrix<T>& c, DeviceMatrix<T>& a, const T* taus_host);
template <class T>
int bt_reduction_to_band_host(Grid* g, int band, T* c, long ldc, long ncols_c, int c_jsrc, const T* a, long lda, long n,
                              int nb, int isrc, int jsrc, const T* taus);

// device time (ms) and algorithmic flops (whole grid) of the last reduction_to_band / bt_reduction_to_band on this process
void red2band_last_profile(double* ms, double* flops);

}  // namespace dlaf_mi355x
