// test_cholesky_cpp.cpp -- the reference's C++ tests of the path, written against the facade
// include/dlaf_mi355x/dlaf.hpp with the reference's names: test/unit/factorization/test_cholesky.cpp:54-120
// (sizes, getCholeskySetters, MatrixMirror scope, tolerance 4 (m+1) error) and one system of
// test/unit/solver/test_triangular.cpp.  One process, one GPU.
#include <cmath>
#include <complex>
#include <cstdio>
#include <limits>
#include <tuple>
#include <vector>

#include <dlaf_mi355x/dlaf.hpp>

using namespace dlaf;

template <class T>
struct TypeUtilities {  // test/include/dlaf_test/util_types.h
  using R = T;
  static T element(double r, double) { return (T) r; }
  static T polar(double r, double) { return (T) r; }
  static constexpr R error = 2 * std::numeric_limits<T>::epsilon();
};
template <class R_>
struct TypeUtilities<std::complex<R_>> {
  using R = R_;
  static std::complex<R_> element(double r, double i) { return {(R_) r, (R_) i}; }
  static std::complex<R_> polar(double r, double theta) { return {(R_) (r * std::cos(theta)), (R_) (r * std::sin(theta))}; }
  static constexpr R error = 8 * std::numeric_limits<R_>::epsilon();
};

// test/include/dlaf_test/matrix/util_generic_lapack.h:39-68
template <class T>
auto getCholeskySetters(blas::Uplo uplo) {
  auto el_a = [uplo](const GlobalElementIndex& index) {
    if ((uplo == blas::Uplo::Lower && index.row() < index.col()) || (uplo == blas::Uplo::Upper && index.row() > index.col()))
      return TypeUtilities<T>::element(-9.9, 0.0);
    const double i = index.row(), j = index.col();
    return TypeUtilities<T>::polar(std::exp2(-(i + j)) / 3 * (std::exp2(2 * (std::min(i, j) + 1)) - 1), -i + j);
  };
  auto el_l = [uplo](const GlobalElementIndex& index) {
    if ((uplo == blas::Uplo::Lower && index.row() < index.col()) || (uplo == blas::Uplo::Upper && index.row() > index.col()))
      return TypeUtilities<T>::element(-9.9, 0.0);
    const double i = index.row(), j = index.col();
    return TypeUtilities<T>::polar(std::exp2(-std::abs(i - j)), -i + j);
  };
  return std::make_tuple(el_a, el_l);
}

static int failures = 0;

// CHECK_MATRIX_NEAR (test/include/dlaf_test/matrix/util_matrix.h): relative OR absolute
template <class T, class Expected>
void check_matrix_near(Expected&& expected, Matrix<T, Device::CPU>& m, double rel, double abs_err, const char* what) {
  const auto ls = m.distribution().local_size();
  for (SizeType j = 0; j < ls.cols(); ++j)
    for (SizeType i = 0; i < ls.rows(); ++i) {
      const T e = expected(m.distribution().global_element_index(LocalElementIndex(i, j)));
      const T g = m(LocalElementIndex(i, j));
      const double diff = std::abs(e - g), mx = std::max(std::abs(e), std::abs(g));
      if (!(diff < abs_err || diff / mx < rel)) {
        if (failures < 10)
          std::fprintf(stderr, "%s: element (%ld,%ld) differs by %g\n", what, (long) i, (long) j, diff);
        ++failures;
        return;
      }
    }
}

const std::vector<std::tuple<SizeType, SizeType>> sizes = {{0, 2}, {5, 8}, {34, 34}, {4, 3}, {16, 10}, {34, 13}, {32, 5}, {150, 64}};

template <class T>
void testCholesky(const blas::Uplo uplo, const SizeType m, const SizeType mb) {
  const LocalElementSize size(m, m);
  const TileElementSize block_size(mb, mb);
  Matrix<T, Device::CPU> mat_h(size, block_size);
  auto [el, res] = getCholeskySetters<T>(uplo);
  matrix::util::set(mat_h, el);
  cholesky_factorization<Backend::GPU, Device::CPU, T>(uplo, mat_h);
  check_matrix_near<T>(res, mat_h, 4 * (m + 1) * TypeUtilities<T>::error, 4 * (m + 1) * TypeUtilities<T>::error, "local");
}

template <class T>
void testCholesky(comm::CommunicatorGrid& grid, const blas::Uplo uplo, const SizeType m, const SizeType mb) {
  const GlobalElementSize size(m, m);
  const TileElementSize block_size(mb, mb);
  comm::Index2D src_rank_index(std::max<SizeType>(0, grid.size().rows() - 1), std::min<SizeType>(1, grid.size().cols() - 1));
  matrix::Distribution distribution(size, block_size, grid.size(), grid.rank(), src_rank_index);
  Matrix<T, Device::CPU> mat_h(std::move(distribution));
  auto [el, res] = getCholeskySetters<T>(uplo);
  matrix::util::set(mat_h, el);
  {
    matrix::MatrixMirror<T, Device::GPU, Device::CPU> mat(grid, mat_h, uplo);
    cholesky_factorization<Backend::GPU, Device::GPU, T>(grid, uplo, mat.get());
  }
  check_matrix_near<T>(res, mat_h, 4 * (m + 1) * TypeUtilities<T>::error, 4 * (m + 1) * TypeUtilities<T>::error, "grid");
}

// one system of getLeftTriangularSystem (util_generic_blas.h:258-296): Left, Lower, NoTrans, NonUnit
template <class T>
void testTriangularSolver(comm::CommunicatorGrid& grid, SizeType m, SizeType n, SizeType mb) {
  const T alpha = TypeUtilities<T>::element(-1.2, .7);
  matrix::Distribution da(GlobalElementSize(m, m), TileElementSize(mb, mb), grid.size(), grid.rank(), comm::Index2D(0, 0));
  matrix::Distribution db(GlobalElementSize(m, n), TileElementSize(mb, mb), grid.size(), grid.rank(), comm::Index2D(0, 0));
  Matrix<T, Device::CPU> mat_a(da), mat_b(db);
  auto el_a = [](const GlobalElementIndex& index) {
    if (index.row() < index.col())
      return TypeUtilities<T>::element(-9.9, 0);
    const double i = index.row(), k = index.col();
    return TypeUtilities<T>::polar((i + 1) / (k + .5), 2 * i - k);
  };
  auto el_x = [](const GlobalElementIndex& index) {
    const double k = index.row(), j = index.col();
    return TypeUtilities<T>::polar((k + .5) / (j + 2), k + j);
  };
  auto el_b = [alpha](const GlobalElementIndex& index) {
    const double i = index.row(), j = index.col();
    const T gamma = TypeUtilities<T>::polar((i + 1) / (j + 2), 2 * i + j);
    return (typename TypeUtilities<T>::R)(i + 1) * gamma / alpha;
  };
  matrix::util::set(mat_a, el_a);
  matrix::util::set(mat_b, el_b);
  triangular_solver<Backend::GPU, Device::CPU, T>(grid, blas::Side::Left, blas::Uplo::Lower, blas::Op::NoTrans,
                                                  blas::Diag::NonUnit, alpha, mat_a, mat_b);
  check_matrix_near<T>(el_x, mat_b, 40 * (m + 1) * TypeUtilities<T>::error, 40 * (m + 1) * TypeUtilities<T>::error, "trsm");
}

// test/unit/eigensolver/test_gen_to_std.cpp:60-83 (getGenToStdElementSetters, util_generic_lapack.h:96-150, itype 1)
template <class T>
void testGenToStd(comm::CommunicatorGrid& grid, blas::Uplo uplo, SizeType m, SizeType mb) {
  using R = typename TypeUtilities<T>::R;
  const double alpha = -2.0, beta = 1.5, gamma = .95;
  matrix::Distribution d(GlobalElementSize(m, m), TileElementSize(mb, mb), grid.size(), grid.rank(), comm::Index2D(0, 0));
  Matrix<T, Device::CPU> mat_a(d), mat_t(d);
  auto other = [uplo](const GlobalElementIndex& index) {
    return (uplo == blas::Uplo::Lower && index.row() < index.col()) || (uplo == blas::Uplo::Upper && index.row() > index.col());
  };
  auto el_t = [=](const GlobalElementIndex& index) {
    if (other(index))
      return TypeUtilities<T>::element(-9.9, 0);
    const double i = index.row(), j = index.col();
    return TypeUtilities<T>::polar(beta / std::exp2(std::abs(i - j)), alpha * (i - j));
  };
  auto el_a = [=](const GlobalElementIndex& index) {
    if (other(index))
      return TypeUtilities<T>::element(-9.9, 0);
    const double i = index.row(), j = index.col();
    return TypeUtilities<T>::polar((i + 1) * (j + 1) * (beta * beta * gamma) / std::exp2(i + j), alpha * (i - j));
  };
  auto res_a = [=](const GlobalElementIndex& index) {
    if (other(index))
      return TypeUtilities<T>::element(-9.9, 0);
    const double i = index.row(), j = index.col();
    return TypeUtilities<T>::polar(gamma / std::exp2(i + j), alpha * (i - j));
  };
  matrix::util::set(mat_a, el_a);
  matrix::util::set(mat_t, el_t);
  eigensolver::internal::generalized_to_standard<Backend::GPU, T>(grid, uplo, mat_a, mat_t);
  check_matrix_near<T>(res_a, mat_a, 0, 10 * (m + 1) * TypeUtilities<T>::error, "gen_to_std");
  check_matrix_near<T>(el_t, mat_t, 0, (R) TypeUtilities<T>::error, "gen_to_std factor untouched");
}

template <class T>
void run_type(comm::CommunicatorGrid& grid) {
  for (auto uplo : {blas::Uplo::Lower, blas::Uplo::Upper})
    for (const auto& [m, mb] : sizes) {
      testCholesky<T>(uplo, m, mb);
      testCholesky<T>(grid, uplo, m, mb);
    }
  testTriangularSolver<T>(grid, 19, 25, 6);
  for (auto uplo : {blas::Uplo::Lower, blas::Uplo::Upper})
    for (const auto& [m, mb] : sizes)
      testGenToStd<T>(grid, uplo, m, mb);
}

int main() {
  dlaf::initialize();
  {
    comm::CommunicatorGrid grid = comm::CommunicatorGrid::single();
    run_type<float>(grid);
    run_type<double>(grid);
    run_type<std::complex<float>>(grid);
    run_type<std::complex<double>>(grid);
    // the miniapp's input on the facade: generator + device-resident factorization
    Matrix<double, Device::CPU> a(LocalElementSize(1000, 1000), TileElementSize(128, 128));
    matrix::util::set_random_hermitian_positive_definite(grid, a);
    if (!(a(LocalElementIndex(7, 7)) > 1999.0 && a(LocalElementIndex(7, 7)) < 2001.0))
      ++failures;
    cholesky_factorization<Backend::GPU, Device::CPU, double>(grid, blas::Uplo::Lower, a);
    if (!(a(LocalElementIndex(7, 7)) > 44.0 && a(LocalElementIndex(7, 7)) < 45.5))
      ++failures;
  }
  dlaf::finalize();
  std::printf("CPP_API_TEST %s\n", failures ? "FAILED" : "OK");
  return failures ? 1 : 0;
}
