// miniapp_triangular_solver.cpp -- the reference's miniapp/miniapp_triangular_solver.cpp for the MI355X library
// on the C++ facade: same options (--m --n --mb --nb --side --uplo --op --diag --grid-rows --grid-cols --nruns
// --nwarmups --type), alpha = 2, flop model n m k / 2 adds + n m k / 2 muls (k = m for side L, n for side R,
// :143-144) and result line.  Two deliberate differences: the operands are well conditioned (upstream fills the
// triangle with uniform random numbers, whose solves overflow at benchmark sizes, and does not check; here
// A = R / k + 2 I and the residual of the last run is printed), and the reported time is the device time of the
// sweep (dlaf_mi355x_solver_profile) -- the operands of this entry start on the host, the wall time with PCIe
// staging is printed next to it.
//   g++ -std=c++17 -O2 -I include miniapp/miniapp_triangular_solver.cpp -L dla_future_amd/lib -ldlaf_mi355x -o miniapp_triangular_solver
#ifdef DLAF_MI355X_WITH_MPI
#include <mpi.h>
#endif

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include <dlaf_mi355x/dlaf.hpp>

using namespace dlaf;

struct Options {
  SizeType m = 4096, n = 512, mb = 256, nb = 256;
  int grid_rows = 1, grid_cols = 1;
  int64_t nruns = 1, nwarmups = 1;
  char type = 'd';
  blas::Side side = blas::Side::Left;
  blas::Uplo uplo = blas::Uplo::Lower;
  blas::Op op = blas::Op::NoTrans;
  blas::Diag diag = blas::Diag::NonUnit;
};

static Options parse(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], v;
    const auto eq = a.find('=');
    if (eq != std::string::npos) {
      v = a.substr(eq + 1);
      a = a.substr(0, eq);
    }
    auto val = [&]() -> std::string {
      if (!v.empty())
        return v;
      if (i + 1 >= argc) {
        std::cerr << "missing value for " << a << std::endl;
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "--m") o.m = std::stoll(val());
    else if (a == "--n") o.n = std::stoll(val());
    else if (a == "--mb") o.mb = std::stoll(val());
    else if (a == "--nb") o.nb = std::stoll(val());
    else if (a == "--grid-rows") o.grid_rows = std::stoi(val());
    else if (a == "--grid-cols") o.grid_cols = std::stoi(val());
    else if (a == "--nruns") o.nruns = std::stoll(val());
    else if (a == "--nwarmups") o.nwarmups = std::stoll(val());
    else if (a == "--type") o.type = (char) std::tolower(val()[0]);
    else if (a == "--side") o.side = std::toupper(val()[0]) == 'R' ? blas::Side::Right : blas::Side::Left;
    else if (a == "--uplo") o.uplo = std::toupper(val()[0]) == 'U' ? blas::Uplo::Upper : blas::Uplo::Lower;
    else if (a == "--op") {
      const char c = (char) std::toupper(val()[0]);
      o.op = c == 'T' ? blas::Op::Trans : c == 'C' ? blas::Op::ConjTrans : blas::Op::NoTrans;
    }
    else if (a == "--diag") o.diag = std::toupper(val()[0]) == 'U' ? blas::Diag::Unit : blas::Diag::NonUnit;
    else if (a == "--check-result" || a == "--backend") (void) val();
    else if (a == "--csv" || a == "--local" || a.rfind("--pika:", 0) == 0 || a.rfind("--dlaf:", 0) == 0) {}
    else {
      std::cerr << "unknown option " << a << std::endl;
      std::exit(2);
    }
  }
  if (o.m <= 0 || o.n <= 0 || o.mb <= 0 || o.nb <= 0 || o.nruns < 1 || std::strchr("sdcz", o.type) == nullptr) {
    std::cerr << "invalid option value" << std::endl;
    std::exit(2);
  }
  if (o.mb != o.nb) {
    std::cerr << "this build needs square blocks: --mb == --nb" << std::endl;
    std::exit(2);
  }
  return o;
}

// counter-based generator: uniform in [-1, 1) from a hash of the global element index (every grid sees the
// same global matrices; cheap enough for benchmark sizes)
static inline double uniform_pm1(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;  // splitmix64
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double) (x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
template <class T>
struct Rand {
  static T make(uint64_t key) { return (T) uniform_pm1(key); }
};
template <class R>
struct Rand<std::complex<R>> {
  static std::complex<R> make(uint64_t key) { return {(R) uniform_pm1(2 * key), (R) uniform_pm1(2 * key + 1)}; }
};

template <class T>
static void run(const Options& opts, comm::CommunicatorGrid& grid, int world_rank) {
  using Base = decltype(std::abs(T{}));
  constexpr bool complex = !std::is_same<T, Base>::value;
  const SizeType k = opts.side == blas::Side::Left ? opts.m : opts.n;
  matrix::Distribution da(GlobalElementSize(k, k), TileElementSize(opts.mb, opts.mb), grid.size(), grid.rank(), comm::Index2D(0, 0));
  matrix::Distribution db(GlobalElementSize(opts.m, opts.n), TileElementSize(opts.mb, opts.nb), grid.size(), grid.rank(), comm::Index2D(0, 0));
  Matrix<T, Device::CPU> ah(da), b_ref(db), bh(db);
  // per-element generators seeded by the global index: every grid sees the same global matrices
  matrix::util::set(ah, [k](const GlobalElementIndex& i) {
    T v = Rand<T>::make((uint64_t) i.row() * (uint64_t) k + (uint64_t) i.col()) / (Base) k;
    return i.row() == i.col() ? v + T(2) : v;
  });
  const uint64_t ncols = (uint64_t) opts.n;
  matrix::util::set(b_ref, [ncols](const GlobalElementIndex& i) {
    return Rand<T>::make(0x5851F42D4C957F2Dull + (uint64_t) i.row() * ncols + (uint64_t) i.col());
  });
  const T alpha = 2.0;
  const auto ls = db.local_size();
  for (int64_t run_index = -opts.nwarmups; run_index < opts.nruns; ++run_index) {
    if (0 == world_rank && run_index >= 0)
      std::cout << "[" << run_index << "]" << std::endl;
    for (SizeType j = 0; j < ls.cols(); ++j)
      for (SizeType i = 0; i < ls.rows(); ++i)
        bh(LocalElementIndex(i, j)) = b_ref(LocalElementIndex(i, j));
    grid.wait_all_communicators();
    const auto t0 = std::chrono::steady_clock::now();
    triangular_solver<Backend::GPU, Device::CPU, T>(grid, opts.side, opts.uplo, opts.op, opts.diag, alpha, ah, bh);
    grid.wait_all_communicators();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double ms = 0, fl = 0;
    dlaf_mi355x_solver_profile(&ms, &fl);
    const double elapsed_time = ms * 1e-3;
    const double add_mul = (double) opts.n * (double) opts.m * (double) k / 2;
    const double gigaflops = (complex ? 2 * add_mul + 6 * add_mul : 2 * add_mul) / elapsed_time / 1e9;
    if (0 == world_rank && run_index >= 0)
      std::cout << "[" << run_index << "] " << elapsed_time << "s " << gigaflops << "GFlop/s " << opts.type
                << (char) opts.side << (char) opts.uplo << (char) opts.op << (char) opts.diag << " (" << opts.m << ", "
                << opts.n << ") (" << opts.mb << ", " << opts.nb << ") (" << grid.size().rows() << ", "
                << grid.size().cols() << ") 1 GPU   [wall with PCIe staging " << wall << "s]" << std::endl;
  }
  // one-process check of the last run on the first rows / columns: op(A) X - alpha B (side L)
  if (grid.size().rows() * grid.size().cols() == 1 && opts.side == blas::Side::Left && opts.op == blas::Op::NoTrans &&
      opts.diag == blas::Diag::NonUnit) {
    double worst = 0;
    const SizeType rows = std::min<SizeType>(opts.m, 64), cols = std::min<SizeType>(opts.n, 8);
    for (SizeType j = 0; j < cols; ++j)
      for (SizeType i = 0; i < rows; ++i) {
        T s{};
        const SizeType p0 = opts.uplo == blas::Uplo::Lower ? 0 : i, p1 = opts.uplo == blas::Uplo::Lower ? i + 1 : opts.m;
        for (SizeType p = p0; p < p1; ++p)
          s += ah(LocalElementIndex(i, p)) * bh(LocalElementIndex(p, j));
        worst = std::max<double>(worst, std::abs(s - alpha * b_ref(LocalElementIndex(i, j))));
      }
    if (world_rank == 0)
      std::cout << "Max |op(A) X - alpha B| on a " << rows << " x " << cols << " corner: " << worst << std::endl;
  }
}

int main(int argc, char** argv) {
  const Options opts = parse(argc, argv);
  int world_rank = 0, world_size = 1;
#ifdef DLAF_MI355X_WITH_MPI
  int provided = 0;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_MULTIPLE, &provided);
  MPI_Comm_rank(MPI_COMM_WORLD, &world_rank);
  MPI_Comm_size(MPI_COMM_WORLD, &world_size);
#endif
  if (opts.grid_rows * opts.grid_cols != world_size) {
    if (world_rank == 0)
      std::cerr << "grid " << opts.grid_rows << " x " << opts.grid_cols << " needs " << opts.grid_rows * opts.grid_cols
                << " processes, got " << world_size << std::endl;
    return 2;
  }
  dlaf::initialize();
  {
#ifdef DLAF_MI355X_WITH_MPI
    comm::CommunicatorGrid grid(MPI_COMM_WORLD, opts.grid_rows, opts.grid_cols, common::Ordering::ColumnMajor);
#else
    comm::CommunicatorGrid grid = comm::CommunicatorGrid::single();
#endif
    switch (opts.type) {
      case 's': run<float>(opts, grid, world_rank); break;
      case 'd': run<double>(opts, grid, world_rank); break;
      case 'c': run<std::complex<float>>(opts, grid, world_rank); break;
      default: run<std::complex<double>>(opts, grid, world_rank); break;
    }
  }
  dlaf::finalize();
#ifdef DLAF_MI355X_WITH_MPI
  MPI_Finalize();
#endif
  return 0;
}
