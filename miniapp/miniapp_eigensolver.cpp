// miniapp_eigensolver.cpp -- the reference's miniapp/miniapp_eigensolver.cpp for the MI355X library on the C++ facade:
// same options (--matrix-size --block-size --grid-rows --grid-cols --nruns --nwarmups --type --uplo --check-result --csv),
// same result lines ("[run] time s type (m, m) (mb, mb) (grid) threads backend", miniapp_eigensolver.cpp:186-200), the
// same checker (--check-result: orthogonality and residual of the eigenpairs, :243-330, on a sample of columns when the
// matrix is large).  The timed window holds the call of the reference's C entry on host arrays (upload, the five stages,
// download) -- upstream times the call on device-resident mirrors; the device time of the five stages alone is printed
// beside it (dlaf_mi355x_eigensolver_profile).  Input as upstream: a random Hermitian matrix (set_random_hermitian, :136-144).
//   g++ -std=c++17 -O2 -I include miniapp/miniapp_eigensolver.cpp -L dla_future_amd/lib -ldlaf_mi355x -o miniapp_eigensolver
#ifdef DLAF_MI355X_WITH_MPI
#include <mpi.h>
#endif

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <complex>
#include <cstring>
#include <vector>
#include <iostream>
#include <limits>
#include <string>

#include <dlaf_mi355x/dlaf.hpp>

using namespace dlaf;

struct Options {
  SizeType m = 4096, mb = 256;
  int grid_rows = 1, grid_cols = 1;
  int64_t nruns = 1, nwarmups = 1;
  char type = 'd';
  SizeType b = -1;  // --band-size (default: the block size, miniapp_reduction_to_band.cpp:62-63)
  std::string check = "none";
  bool csv = false;
  std::string info;
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
    if (a == "--matrix-size")
      o.m = std::stoll(val());
    else if (a == "--block-size")
      o.mb = std::stoll(val());
    else if (a == "--grid-rows")
      o.grid_rows = std::stoi(val());
    else if (a == "--grid-cols")
      o.grid_cols = std::stoi(val());
    else if (a == "--nruns")
      o.nruns = std::stoll(val());
    else if (a == "--nwarmups")
      o.nwarmups = std::stoll(val());
    else if (a == "--type")
      o.type = (char) std::tolower(val()[0]);
    else if (a == "--band-size")
      o.b = std::stoll(val());
    else if (a == "--check-result")
      o.check = val();
    else if (a == "--csv")
      o.csv = true;
    else if (a == "--pp-info")
      o.info = val();
    else if (a == "--local" || a == "--backend" || a.rfind("--pika:", 0) == 0 || a.rfind("--dlaf:", 0) == 0) {
      if (a == "--backend")
        (void) val();  // there is one backend
    }
    else {
      std::cerr << "unknown option " << a << "\nusage: miniapp_eigensolver --matrix-size N --block-size NB [--band-size B] [--grid-rows R "
                   "--grid-cols C] [--nruns K] [--nwarmups W] [--type s|d|c|z] [--uplo L|U] "
                   "[--check-result none|last|all] [--csv]" << std::endl;
      std::exit(2);
    }
  }
  if (o.m < 0 || o.mb < 1 || o.nruns < 1 || o.nwarmups < 0 || std::strchr("sdcz", o.type) == nullptr ||
      (o.check != "none" && o.check != "last" && o.check != "all")) {
    std::cerr << "invalid option value" << std::endl;
    std::exit(2);
  }
  return o;
}

template <class T>
static void run(const Options& opts, comm::CommunicatorGrid& comm_grid, int world_rank) {
  using Base = BaseType<T>;
  GlobalElementSize matrix_size(opts.m, opts.m);
  TileElementSize block_size(opts.mb, opts.mb);
  matrix::Distribution dist(matrix_size, block_size, comm_grid.size(), comm_grid.rank(), comm::Index2D(0, 0));

  Matrix<T, Device::CPU> matrix_ref(dist);
  {
    // set_random_hermitian = the generator of set_random_hermitian_positive_definite without the 2 n on the diagonal
    // (util_matrix.h:460-462, :499-501)
    matrix::util::set_random_hermitian_positive_definite(comm_grid, matrix_ref);
    const LocalElementSize ls = dist.local_size();
    for (SizeType j = 0; j < ls.cols(); ++j)
      for (SizeType i = 0; i < ls.rows(); ++i) {
        const GlobalElementIndex g = dist.global_element_index(LocalElementIndex(i, j));
        if (g.row() == g.col())
          matrix_ref(LocalElementIndex(i, j)) -= T(2 * (Base) opts.m);
      }
  }
  Matrix<T, Device::CPU> matrix(dist), evecs(dist);
  std::vector<Base> evals;

  for (int64_t run_index = -opts.nwarmups; run_index < opts.nruns; ++run_index) {
    if (0 == world_rank && run_index >= 0)
      std::cout << "[" << run_index << "]" << std::endl;
    {
      const LocalElementSize ls = dist.local_size();
      for (SizeType j = 0; j < ls.cols(); ++j)
        std::memcpy(matrix.ptr() + j * matrix.ld(), matrix_ref.ptr() + j * matrix_ref.ld(), (size_t) ls.rows() * sizeof(T));
    }
    comm_grid.wait_all_communicators();
    const auto t0 = std::chrono::steady_clock::now();
    hermitian_eigensolver<Backend::GPU, T>(comm_grid, blas::Uplo::Lower, matrix, evals, evecs);
    comm_grid.wait_all_communicators();
    const double elapsed_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double ms[5];
    dlaf_mi355x_eigensolver_profile(ms);
    if (0 == world_rank && run_index >= 0) {
      std::cout << "[" << run_index << "] " << elapsed_time << "s " << opts.type << " (" << opts.m << ", " << opts.m << ") ("
                << opts.mb << ", " << opts.mb << ") (" << comm_grid.size().rows() << ", " << comm_grid.size().cols()
                << ") 1 GPU" << std::endl;
      std::cout << "    device stages [ms]: reduction_to_band " << ms[0] << " band_to_tridiagonal " << ms[1]
                << " tridiagonal_eigensolver " << ms[2] << " bt_band_to_tridiagonal " << ms[3] << " bt_reduction_to_band "
                << ms[4] << " sum " << ms[0] + ms[1] + ms[2] + ms[3] + ms[4] << std::endl;
      if (opts.csv)
        std::cout << "CSVData-2, run, " << run_index << ", time, " << elapsed_time << ", type, " << opts.type
                  << ", matrixsize, " << opts.m << ", blocksize, " << opts.mb << ", comm_rows, " << comm_grid.size().rows()
                  << ", comm_cols, " << comm_grid.size().cols() << ", threads, 1, backend, GPU, " << opts.info << std::endl;
    }
    const bool check = opts.check == "all" || (opts.check == "last" && run_index == opts.nruns - 1);
    if (check && comm_grid.size().rows() * comm_grid.size().cols() == 1) {
      // miniapp_eigensolver.cpp:243-330 on one process: |E^H E - I|_max and |A E - E Lambda|_max over a column sample
      const SizeType n = opts.m;
      const SizeType stride = std::max<SizeType>(1, n / 32);
      double orth = 0, res = 0, amax = 0;
      for (SizeType c = 0; c < n; c += stride) {
        for (SizeType c2 = 0; c2 < n; c2 += stride) {
          std::complex<double> dot = 0;
          for (SizeType i = 0; i < n; ++i)
            dot += std::conj(std::complex<double>(evecs(LocalElementIndex(i, c)))) * std::complex<double>(evecs(LocalElementIndex(i, c2)));
          orth = std::max(orth, std::abs(dot - (c == c2 ? 1.0 : 0.0)));
        }
        for (SizeType i = 0; i < n; ++i) {
          std::complex<double> s = 0;
          for (SizeType k = 0; k < n; ++k) {
            const std::complex<double> aik = k <= i ? std::complex<double>(matrix_ref(LocalElementIndex(i, k)))
                                                    : std::conj(std::complex<double>(matrix_ref(LocalElementIndex(k, i))));
            s += aik * std::complex<double>(evecs(LocalElementIndex(k, c)));
            amax = std::max(amax, std::abs(aik));
          }
          res = std::max(res, std::abs(s - (double) evals[(size_t) c] * std::complex<double>(evecs(LocalElementIndex(i, c)))));
        }
      }
      const double eps = std::numeric_limits<Base>::epsilon();
      const bool sorted = std::is_sorted(evals.begin(), evals.end());
      const bool ok = sorted && orth <= 10 * n * eps && res <= 10 * n * eps * std::max(1.0, amax * n);
      std::cout << (ok ? "Check: OK" : "Check: ERROR") << "  sorted " << sorted << "  |E^H E - I|max " << orth
                << "  |A E - E L|max " << res << "  (column sample, stride " << stride << ")" << std::endl;
    }
    else if (check && world_rank == 0)
      std::cerr << "Warning! Result checking on a process grid: use tests/dist_worker.py (gathers the eigenvectors)." << std::endl;
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
  if (std::getenv("LOCAL_RANK") == nullptr) {
    MPI_Comm node;
    MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, world_rank, MPI_INFO_NULL, &node);
    int local = 0;
    MPI_Comm_rank(node, &local);
    setenv("LOCAL_RANK", std::to_string(local).c_str(), 0);  // the library picks its GPU from it
    MPI_Comm_free(&node);
  }
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
    comm::CommunicatorGrid comm_grid(MPI_COMM_WORLD, opts.grid_rows, opts.grid_cols, common::Ordering::ColumnMajor);
#else
    comm::CommunicatorGrid comm_grid = comm::CommunicatorGrid::single();
#endif
    switch (opts.type) {
      case 's': run<float>(opts, comm_grid, world_rank); break;
      case 'd': run<double>(opts, comm_grid, world_rank); break;
      case 'c': run<std::complex<float>>(opts, comm_grid, world_rank); break;
      default: run<std::complex<double>>(opts, comm_grid, world_rank); break;
    }
  }
  dlaf::finalize();
#ifdef DLAF_MI355X_WITH_MPI
  MPI_Finalize();
#endif
  return 0;
}
