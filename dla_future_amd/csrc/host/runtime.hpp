// runtime.hpp -- the thin C++ host of the MI355X Cholesky: grid + transport, device tile-layout
// matrix, and the executor that issues the right-looking tile DAG onto HIP streams/events.
//
// It stands where the reference has pika senders + async_rw_mutex tile pipelines
// (sender/transform.h:55-103, matrix/internal/tile_pipeline.h:36-51), the MPI communicator grid
// (communication/communicator_grid.h:37-153) and Panel workspaces (matrix/panel.h:42-631).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "../device/device_api.hpp"
#include "distribution.hpp"

namespace dlaf_mi355x {

[[noreturn]] void fatal(const char* fmt, ...);

#define DLAF_HIP_CHECK(expr)                                                                     \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      ::dlaf_mi355x::fatal("[dlaf_mi355x] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, \
                           __LINE__, #expr);                                                     \
  } while (0)

// ------------------------------------------------------------------------------------------------
// Workspace pool of the widenings (eigensolver stages): the reference takes its device temporaries from Umpire pools
// (src/memory/memory_chunk.cpp, init.cpp:100-140); here hipMalloc / hipFree of the multi-gigabyte temporaries of
// bt_band_to_tridiagonal and the divide & conquer solver cost 0.25-0.5 s per solve on some boxes (measured:
// profiles/r04_eigensolver_alloc_phases.txt), so blocks of 4 MiB and more are kept for the next call instead of going
// back to the driver.  pool_free waits for the device first (what hipFree does implicitly).  DLAF_MI355X_POOL_GB caps
// what is kept (default 64, 0 = no pool); pool_release() gives everything back (dlaf_finalize).
hipError_t pool_malloc(void** p, size_t bytes);
hipError_t pool_free(void* p);
void pool_release();
size_t pool_idle_bytes();

// ------------------------------------------------------------------------------------------------
// Transport: the broadcast primitive along a process row / column (communication/kernels/
// internal/broadcast.h:36-119 in the reference: MPI_Ibcast).  Two implementations:
//   * RCCL over xGMI: ncclBroadcast on device buffers, stream-ordered (the production path);
//   * host callbacks: device -> pinned host -> user callback -> device.  This is the analogue of
//     the reference's non-GPU-aware-MPI staging (sender/with_temporary_tile.h:87-204) and what the
//     gloo-based tests and single-GPU multi-process runs use.
enum class CommAxis : int { Row = 0, Col = 1 };

// root is the rank INSIDE the row/column communicator (= process column / row index)
typedef int (*dlaf_host_bcast_fn)(void* user, int axis, int root, void* host_buf, size_t bytes);
typedef int (*dlaf_host_barrier_fn)(void* user);

class Transport {
public:
  virtual ~Transport() = default;
  virtual bool device_side() const = 0;
  // device-side transports: enqueue on stream.  host-side: stream is synchronised first.
  virtual void bcast(CommAxis axis, int root, int my_index, const void* send, void* recv, size_t bytes,
                     hipStream_t stream) = 0;
  virtual void group_begin() {}
  virtual void group_end() {}
  virtual void barrier(hipStream_t stream) = 0;
  // max over every rank of the grid of n host doubles (the reference's sync::reduce with MPI_MAX in
  // max_norm, include/dlaf/auxiliary/norm/mc.h); result on every rank
  virtual void allreduce_max(double* host_vals, int n, int nprow, int npcol, int myrow, int mycol) = 0;
  // element-wise SUM of a device buffer over the ranks of a communicator, in place, result on every member
  // (the reference's schedule_all_reduce_in_place / schedule_reduce_* with MPI_SUM, communication/kernels/
  // all_reduce.h, reduce.h).  type: s, d, c, z; scope: 'A' the whole grid, 'R' my process row, 'C' my process column.
  // Every member ends up with the same bits (the host transport sums in root order, RCCL's ring does too).
  virtual void allreduce_sum(void* dev_buf, size_t count, char type, char scope, hipStream_t stream) = 0;
  // step marker of the executor (a no-op for real transports; the recording wrapper logs it)
  virtual void mark(long /*step*/) {}
  // the grid this transport serves (set by grid_transport)
  int nprow = 1, npcol = 1, myrow = 0, mycol = 0;
};

// One communication event as the recording transport logs it (dlaf_mi355x_grid_comm_log_*): what the
// reference's CommunicatorPipeline token serialises (sender/transform_mpi.h:60-75) -- every member of a
// communicator must issue the same sequence.
struct CommEvent {
  long kind;   // 0 row broadcast, 1 column broadcast, 2 step marker, 3 barrier, 4 allreduce
  long root;   // broadcast: root index inside the communicator; marker: step
  long bytes;
  long group;  // 1: issued inside group_begin / group_end
};

struct Grid {
  int nprow = 1, npcol = 1;
  int myrow = 0, mycol = 0;
  int rank = 0, nranks = 1;
  char order = 'R';
  std::unique_ptr<Transport> transport;  // null for a 1x1 grid (host grids: created lazily)
  dlaf_host_bcast_fn host_bcast = nullptr;
  dlaf_host_barrier_fn host_barrier = nullptr;
  void* host_user = nullptr;
  // called once when the grid is freed (dlaf_free_grid / dlaf_finalize): releases what the creator of a
  // host grid keeps alive for the callbacks (the MPI shim's communicators)
  void (*on_free)(void*) = nullptr;
  void* on_free_user = nullptr;
  // communication log (tests): when on, `transport` is wrapped by a recorder that appends here
  bool comm_log_on = false;
  std::vector<CommEvent> comm_log;
  ~Grid() {
    transport.reset();
    if (on_free)
      on_free(on_free_user);
  }
};

std::unique_ptr<Transport> make_rccl_transport(const void* unique_id, int nranks, int rank, int nprow,
                                               int npcol, int myrow, int mycol);
void rccl_get_unique_id(void* out128);
// creates the lazily-built host transport of a grid and, when the grid's communication log is on, wraps the
// transport with the recorder; idempotent.  Returns the transport (null for a 1x1 grid without communicators).
Transport* grid_transport(Grid& g);
std::unique_ptr<Transport> make_host_transport(dlaf_host_bcast_fn bcast, dlaf_host_barrier_fn barrier,
                                               void* user);
// host-driven peer copies (hipIpc mappings + interprocess events), control messages over the host callbacks
std::unique_ptr<Transport> make_peer_transport(dlaf_host_bcast_fn bcast, dlaf_host_barrier_fn barrier,
                                               void* user);
std::unique_ptr<Transport> make_callback_transport(dlaf_host_bcast_fn bcast, dlaf_host_barrier_fn barrier,
                                                   void* user);

// ------------------------------------------------------------------------------------------------
// Device matrix in tile layout + the per-factorization workspaces.
struct MatrixBase {
  virtual ~MatrixBase() = default;
  char type = 'd';
};

template <class T>
struct DeviceMatrix : MatrixBase {
  Grid* grid = nullptr;
  char uplo = 'L';
  bool transposed = false;  // uplo == 'U': the device holds the transposed view, factored as lower
  Axis rows, cols;          // distribution of the VIEW
  long n = 0;
  int nb = 1;
  long nt = 0;
  long ltr = 0, ltc = 0;    // local tiles of the view
  size_t tile_elems = 0;    // nb*nb

  T* tiles = nullptr;       // ltr*ltc tiles
  T* diag_ws = nullptr;     // 2 x (nb*nb + ceil(nb/64)*64*64): received diagonal tile + its inverse blocks
  T* winv = nullptr;        // 2 x ceil(nb/64) * 64*64 inverted diagonal blocks (owner side)
  T* panel[2] = {nullptr, nullptr};   // ltr tiles each (received column panel)
  T* panelT[2] = {nullptr, nullptr};  // ltc tiles each (transposed panel)
  T* staging = nullptr;     // column-major staging for upload/download
  size_t staging_elems = 0;
  unsigned* coop_sync = nullptr;  // flags / counters: [update slices | one slice per diagonal tile], see the constructor
  size_t coop_sync_words = 0, coop_sync_update_slices = 0, coop_sync_potrf_words = 0;
  int* info = nullptr;      // device flag
  int* info_host = nullptr; // pinned

  hipStream_t s_high = nullptr, s_low = nullptr, s_comm = nullptr;
  // per step: panel solved, lookahead columns updated, diagonal tile ready, panels received, head tile
  // solved / received; ev_start / ev_done fence a factorization on the three streams
  std::vector<hipEvent_t> ev_panel, ev_high, ev_diag, ev_bcast, ev_head, ev_headb, ev_start, ev_done;

  // live timing of the launch classes with HIP events on the stream each class runs on
  // (kind 0: trailing bulk update, 1: lookahead-column update, 2: panel TRSM, 3: tile POTRF chain)
  struct ProfSlot {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
    double flops = 0, bytes = 0;   // algorithmic, summed over the launches of the last run
    double ms = 0;                 // filled by wait()
    long launches = 0;
  };
  ProfSlot prof[4];
  long bulk_slots = 0;  // resident workgroups of the bulk update kernel on this GPU (CUs x blocks per CU)
  bool profiling = true;
  void prof_begin(int kind, hipStream_t s);
  void prof_end(int kind, hipStream_t s, double flops, double bytes);

  // Transposed panel of a step down the process columns (communication/broadcast_panel.h:125-210), ONE
  // broadcast per root process row: the local tile columns jl >= jl_n fall into `period` classes by the
  // process row that owns global tile row global_of(jl); class c = tiles jl_n + c, jl_n + c + period, ...
  // lands contiguously at dst + c * ts2 (the root packs its -- strided -- tiles there first).  a_base: column
  // panel, tile of local row il at a_base + (il - il_n) * tile_elems.  The tile of the last global row is
  // never a gemm operand and stays out (:186-191).  Returns the number of broadcasts issued.
  int bcast_transposed_panel(Transport* tr, CommAxis ax_col, const T* a_base, long il_n, long jl_n, T* dst,
                             hipStream_t s, int& period, long& ts2);

  T* tile(long il, long jl) const { return tiles + (size_t) (il + jl * ltr) * tile_elems; }
  size_t winv_elems() const { return (size_t) ((nb + kDiagBlock - 1) / kDiagBlock) * kDiagBlock * kDiagBlock; }

  void create(Grid* g, char uplo_, long n_, int nb_, int isrc, int jsrc);
  void destroy();
  ~DeviceMatrix() override { destroy(); }

  void upload(const T* host, long ld);     // caller's local column-major array -> tiles
  // tiles -> caller's array (uplo triangle only).  staging_is_current: the staging copy was filled by upload()
  // from this same array and the caller has not run since (saves re-staging the diagonal tiles)
  void download(T* host, long ld, bool staging_is_current = false);
  void copy_from(const DeviceMatrix<T>& other);
  // Checker (miniapp/miniapp_cholesky.cpp:408-443): `this` holds the ORIGINAL matrix and is overwritten
  // with A - L L^H on the uplo triangle; returns max|A - L L^H| and max|A| over the whole grid.
  void residual_of(DeviceMatrix<T>& factor, double* max_diff, double* max_a);
  int factorize();                         // blocking; returns LAPACK-style info
  // factorize() of a matrix that upload() just filled from `host`, with the download of every finished
  // tile column overlapped with the rest of the factorization (a helper thread follows ev_panel[k]); on
  // return `host` holds the factor.  The blocking host entry points (dlaf_p?potrf ...) use it.
  int factorize_and_download(T* host, long ld);
  std::atomic<long> panels_issued{-1};     // last step whose ev_panel has been recorded by factorize_async
  void factorize_async();                  // enqueue only
  // drain + info.  On a process grid the LAPACK info is made the same on every rank (MAX over the grid of
  // the device flags: every kernel that saw the failing diagonal tile stored the same index, the others 0;
  // the scheduling-failure code wins over everything) -- collective, like the factorization itself.
  int wait();
  int local_info = 0;  // this rank's own device status word of the last wait(), before the grid agreed on one
  // one tile of the device copy (global tile indices of the caller's matrix, not of the transposed view)
  // to / from a dense host array; returns false when this rank does not own the tile
  bool fetch_tile(long gi, long gj, T* host, long ld);
  // Measurement hook (bench.py): the panel TRSM of step 0 ALONE on the device -- on a copy of the factored first
  // tile column (local rows below the diagonal) with the factored diagonal tile, `reps` launches between two HIP
  // events.  In the factorization the panel solves run beside the bulk update on a few free workgroup slots, so
  // their in-situ durations say nothing about the kernel; this does.  One-process grids, after factorize().
  // Returns the average ms per launch; *flops / *bytes: algorithmic work of one launch.
  double trsm_profile(int reps, double* flops, double* bytes);
};

// single-tile operations with host operands (tests of the tile kernels through the C ABI)
template <class T>
int tile_potrf(char uplo, int n, T* a, int lda);
template <class T>
void tile_trsm(char uplo, int m, int n, const T* a, int lda, T* b, int ldb);
template <class T>
void tile_herk(char uplo, int n, int k, const T* a, int lda, T* c, int ldc);
template <class T>
void tile_gemm(char uplo, int m, int n, int k, const T* a, int lda, const T* b, int ldb, T* c, int ldc);

// Communication self-test of a grid: every member of every row / column communicator broadcasts a
// coordinate-dependent pattern in turn (in place, out of place and grouped, the three forms the
// factorization issues), then barrier + max-allreduce.  Returns the number of mismatching checks.
int grid_selftest(Grid& g, size_t bytes);

// op(A) X = alpha B (side L) / X op(A) = alpha B (side R) on the grid, host operands (solver.cpp);
// m x n: size of B, nb: the square block of A = B's block along the triangular dimension, nb_free: B's block along
// the other dimension (<= 0: nb)
template <class T>
int triangular_solver_host(Grid* g, char side, char uplo, char op, char diag, T alpha, const T* a, long lda, int a_isrc,
                           int a_jsrc, T* b, long ldb, long m, long n, int nb, int b_isrc, int b_jsrc, int nb_free = 0);

void solver_last_profile(double* ms, double* flops);

// Device-resident operands for the solver (solver.cpp): a general m x n matrix in tile layout behind a MatrixBase
// handle, and dlaf::triangular_solver on a resident triangular matrix (a DeviceMatrix, e.g. the factor p?potrf left
// there) and a resident right-hand side -- no PCIe traffic.
MatrixBase* general_matrix_create(Grid* g, char type, long m, long n, int nb, int isrc, int jsrc);
void general_matrix_transfer(MatrixBase* h, void* host, long ld, bool upload);
int triangular_solver_device(char side, char uplo, char op, char diag, const void* alpha, MatrixBase* a, MatrixBase* b);

// A <- L^-1 A L^-H (uplo L) / U^-H A U^-1 (uplo U) with the Cholesky factor held in the same uplo triangle of
// `l` (gen_to_std.cpp; dlaf::eigensolver::internal::generalized_to_standard); device-resident and host forms
template <class T>
int gen_to_std_device(DeviceMatrix<T>& a, DeviceMatrix<T>& l);
template <class T>
int gen_to_std_host(Grid* g, char uplo, T* a, long lda, const T* l, long ldl, long n, int nb, int isrc, int jsrc);

void runtime_init();
void runtime_finalize();
bool runtime_initialized();

}  // namespace dlaf_mi355x
