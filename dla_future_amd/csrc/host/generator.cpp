// generator.cpp -- the miniapp's synthetic input: random Hermitian positive definite matrix.
//
// Mirrors set_random_hermitian_positive_definite (include/dlaf/util_matrix.h:498-501 ->
// :399-442 seeds, :323-380 tile setters, :148-179 getter_random) on this process's local part of
// the block-cyclic matrix.  Like the reference it draws from std::mt19937_64 through
// std::uniform_real_distribution<T>(-1, 1): the values are therefore those of the C++ standard
// library this file is built with (libstdc++ here, as upstream).  The complex sample
// polar(|draw|, pi*draw) has its two draws pinned to g++'s order (see Getter<std::complex<T>>).
#include <cmath>
#include <complex>
#include <random>
#include <thread>
#include <vector>

#include "distribution.hpp"

namespace dlaf_mi355x {

namespace {
template <class T>
class Getter {
public:
  explicit Getter(long seed) : engine_(static_cast<std::size_t>(seed)) {}
  T operator()() { return sampler_(engine_); }

private:
  std::mt19937_64 engine_;
  std::uniform_real_distribution<T> sampler_{-1, 1};
};

template <class T>
class Getter<std::complex<T>> : private Getter<T> {
public:
  using Getter<T>::Getter;
  std::complex<T> operator()() {
    // upstream passes both draws as function arguments (unspecified order).  g++ evaluates them right
    // to left -- first draw = angle, second = magnitude -- and that order is fixed here so that every
    // compiler (this file is built with hipcc/clang) generates the same matrix as the test oracle.
    const T angle = Getter<T>::operator()();
    const T magnitude = Getter<T>::operator()();
    return std::polar<T>(std::abs(magnitude), static_cast<T>(M_PI) * angle);
  }
};

template <class T>
T conj_of(T v) {
  return v;
}
template <class T>
std::complex<T> conj_of(std::complex<T> v) {
  return std::conj(v);
}

template <class T>
void fill_tile(T* tile, long ld, long n, int nb, long gi, long gj) {
  using R = decltype(std::real(T{}));
  const long row0 = gi * nb, col0 = gj * nb;
  const int rows = (int) std::min<long>(nb, n - row0), cols = (int) std::min<long>(nb, n - col0);
  const long seed = (gi >= gj) ? col0 + row0 * n : row0 + col0 * n;
  Getter<T> rnd(seed);
  if (gi == gj) {
    for (int j = 0; j < cols; ++j) {
      for (int i = 0; i < j; ++i) {
        const T v = rnd();
        tile[i + j * ld] = v;
        tile[j + i * ld] = conj_of(v);
      }
      tile[j + j * ld] = T(std::real(rnd()) + R(2 * n));
    }
    return;
  }
  for (int j = 0; j < nb; ++j)
    for (int i = 0; i < nb; ++i) {
      const T v = rnd();
      if (gi > gj) {
        if (i < rows && j < cols)
          tile[i + j * ld] = v;
      }
      else if (j < rows && i < cols) {
        tile[j + i * ld] = conj_of(v);
      }
    }
}
}  // namespace

template <class T>
void set_random_hpd_local(T* a, long ld, long n, int nb, const Axis& rows, const Axis& cols, int nthreads) {
  const long ltr = rows.local_tiles(), ltc = cols.local_tiles();
  const long total = ltr * ltc;
  if (total == 0)
    return;
  if (nthreads <= 0)
    nthreads = (int) std::max(1u, std::thread::hardware_concurrency());
  nthreads = (int) std::min<long>(nthreads, total);
  auto work = [&](int tid) {
    for (long t = tid; t < total; t += nthreads) {
      const long il = t % ltr, jl = t / ltr;
      fill_tile(a + il * nb + jl * nb * ld, ld, n, nb, rows.global_of(il), cols.global_of(jl));
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthreads; ++t)
    pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool)
    th.join();
}

template void set_random_hpd_local<float>(float*, long, long, int, const Axis&, const Axis&, int);
template void set_random_hpd_local<double>(double*, long, long, int, const Axis&, const Axis&, int);
template void set_random_hpd_local<std::complex<float>>(std::complex<float>*, long, long, int, const Axis&,
                                                        const Axis&, int);
template void set_random_hpd_local<std::complex<double>>(std::complex<double>*, long, long, int, const Axis&,
                                                         const Axis&, int);

}  // namespace dlaf_mi355x
