// miniapp_reduction_to_band.cpp -- the reference's miniapp/miniapp_reduction_to_band.cpp for the MI355X library on
// the C++ facade: same options (--matrix-size --block-size --band-size --grid-rows --grid-cols --nruns --nwarmups --type
// --csv), same timed window (the matrix resident on the device, barrier - reduction_to_band - barrier, :141-160), same
// flop model (add_mul = 2/3 n^3 - n^2 nb, :163-168), same result lines; as upstream, --check-result is accepted and
// not implemented (:84-87) -- parity is covered by tests/test_gpu_red2band.py.  Input as upstream: a random Hermitian
// matrix (set_random_hermitian, :120-126).
//   g++ -std=c++17 -O2 -I include miniapp/miniapp_reduction_to_band.cpp -L dla_future_amd/lib -ldlaf_mi355x -o miniapp_reduction_to_band
#ifdef DLAF_MI355X_WITH_MPI
#include <mpi.h>
#endif

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
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
      std::cerr << "unknown option " << a << "\nusage: miniapp_reduction_to_band --matrix-size N --block-size NB [--band-size B] [--grid-rows R "
                   "--grid-cols C] [--nruns K] [--nwarmups W] [--type s|d|c|z] [--uplo L|U] "
                   "[--check-result none|last|all] [--csv]" << std::endl;
      std::exit(2);
    }
  }
  if (o.b < 0)
    o.b = o.mb;
  if (o.b < 2 || o.mb % o.b != 0) {
    std::cerr << "band size " << o.b << " must be >= 2 and divide the block size " << o.mb << std::endl;
    std::exit(2);
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
  using Base = typename std::conditional<std::is_same<T, float>::value || std::is_same<T, std::complex<float>>::value,
                                         float, double>::type;
  constexpr bool complex = !std::is_same<T, Base>::value;
  GlobalElementSize matrix_size(opts.m, opts.m);
  TileElementSize block_size(opts.mb, opts.mb);
  matrix::Distribution dist(matrix_size, block_size, comm_grid.size(), comm_grid.rank(), comm::Index2D(0, 0));

  Matrix<T, Device::GPU> matrix_ref(comm_grid, dist, blas::Uplo::Lower);
  {
    Matrix<T, Device::CPU> host(dist);
    // set_random_hermitian = the generator of set_random_hermitian_positive_definite without the 2 n on the diagonal
    // (util_matrix.h:460-462, :499-501)
    matrix::util::set_random_hermitian_positive_definite(comm_grid, host);
    const LocalElementSize ls = dist.local_size();
    for (SizeType j = 0; j < ls.cols(); ++j)
      for (SizeType i = 0; i < ls.rows(); ++i) {
        const GlobalElementIndex g = dist.global_element_index(LocalElementIndex(i, j));
        if (g.row() == g.col())
          host(LocalElementIndex(i, j)) -= T(2 * (Base) opts.m);
      }
    dlaf_mi355x_matrix_upload(matrix_ref.handle(), host.ptr(), (int) host.ld());
  }
  Matrix<T, Device::GPU> matrix(comm_grid, dist, blas::Uplo::Lower);

  for (int64_t run_index = -opts.nwarmups; run_index < opts.nruns; ++run_index) {
    if (0 == world_rank && run_index >= 0)
      std::cout << "[" << run_index << "]" << std::endl;
    dlaf_mi355x_matrix_copy(matrix.handle(), matrix_ref.handle());  // a fresh copy outside the timer (:133-139)
    comm_grid.wait_all_communicators();
    const auto t0 = std::chrono::steady_clock::now();
    auto taus = eigensolver::internal::reduction_to_band<Backend::GPU, T>(comm_grid, matrix, opts.b);
    comm_grid.wait_all_communicators();
    const double elapsed_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void) taus;

    const double n = (double) opts.m, nbd = (double) opts.mb;
    const double add_mul = 2. / 3. * n * n * n - n * n * nbd;
    const double gigaflops = (complex ? 2 * add_mul + 6 * add_mul : 2 * add_mul) / elapsed_time / 1e9;
    if (0 == world_rank && run_index >= 0) {
      std::cout << "[" << run_index << "] " << elapsed_time << "s " << gigaflops << "GFlop/s " << opts.type << " ("
                << opts.m << ", " << opts.m << ") (" << opts.mb << ", " << opts.mb << ") " << opts.b << " ("
                << comm_grid.size().rows() << ", " << comm_grid.size().cols() << ") 1 GPU" << std::endl;
      if (opts.csv)
        std::cout << "CSVData-2, run, " << run_index << ", time, " << elapsed_time << ", GFlops, " << gigaflops
                  << ", type, " << opts.type << ", matrixsize, " << opts.m << ", blocksize, " << opts.mb
                  << ", band_size, " << opts.b << ", comm_rows, " << comm_grid.size().rows() << ", comm_cols, "
                  << comm_grid.size().cols() << ", threads, 1, backend, GPU, " << opts.info << std::endl;
    }
  }
  if (opts.check != "none" && world_rank == 0)
    std::cerr << "Warning! At the moment result checking it is not implemented." << std::endl;
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
