/* dlaf_mi355x.h -- extensions of the C interface that the MI355X build adds next to the
 * reference's dlaf_c/ headers.  Plain C ABI: pointers, sizes, chars; no C++ or torch types.
 *
 *   - MPI-free grids over RCCL/xGMI or over caller-supplied host broadcasts,
 *   - a device-resident matrix handle so a driver can time the factorization with the matrix
 *     already in HBM (what the reference's miniapp does with MatrixMirror,
 *     miniapp/miniapp_cholesky.cpp:133-155),
 *   - the synthetic input of the miniapp (include/dlaf/util_matrix.h:498-501),
 *   - the four tile operations with exactly the argument sets the factorization issues
 *     (include/dlaf/factorization/cholesky/impl.h:45-147), for known-answer tests,
 *   - the 2-D block-cyclic index helpers (include/dlaf/matrix/util_distribution.h:82-196).
 *
 * type is one of 's' 'd' 'c' 'z'; uplo 'L' or 'U'.  Functions returning int return 0 on success,
 * a negative value for an invalid argument, or a positive LAPACK info. */
#pragma once
#include <stddef.h>

#include <dlaf_c/desc.h>
#include <dlaf_c/utils.h>

/* ---- grids --------------------------------------------------------------------------------- */
/* 1x1 grid in this process (no communication).  Returns a context. */
DLAF_EXTERN_C int dlaf_mi355x_create_grid_single(void) DLAF_NOEXCEPT;

/* RCCL: rank 0 calls dlaf_mi355x_rccl_unique_id and ships the 128 bytes to every rank by any
 * means (torch.distributed, MPI, a file); then every rank calls dlaf_mi355x_create_grid_rccl.
 * order 'R': rank = myrow*npcol + mycol, 'C': rank = mycol*nprow + myrow
 * (reference: common/index2d.h:345-355, dlaf_create_grid's order argument). */
#define DLAF_MI355X_UNIQUE_ID_BYTES 128
DLAF_EXTERN_C void dlaf_mi355x_rccl_unique_id(void* out_128_bytes) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_create_grid_rccl(const void* unique_id_128_bytes, int nranks, int rank, int nprow,
                                               int npcol, char order) DLAF_NOEXCEPT;

/* Host-staged transport: the library hands a pinned HOST buffer to `bcast`, which must broadcast
 * it from member `root` of this process's row (axis 0) or column (axis 1) communicator -- root is
 * the process column index for axis 0 and the process row index for axis 1 -- and return 0. */
typedef int (*dlaf_mi355x_bcast_fn)(void* user, int axis, int root, void* host_buf, size_t bytes);
typedef int (*dlaf_mi355x_barrier_fn)(void* user);
DLAF_EXTERN_C int dlaf_mi355x_create_grid_host(int nranks, int rank, int nprow, int npcol, char order,
                                               dlaf_mi355x_bcast_fn bcast, dlaf_mi355x_barrier_fn barrier,
                                               void* user) DLAF_NOEXCEPT;

/* Runs the grid's broadcast callback on a caller-owned HOST buffer (no GPU work): a wiring check of
 * the row (axis 0) / column (axis 1) communicators of a host grid for CPU-only tests. */
DLAF_EXTERN_C int dlaf_mi355x_grid_host_bcast(int context, int axis, int root, void* host_buf, size_t bytes) DLAF_NOEXCEPT;

/* fn(user) is called once when the grid is freed (dlaf_free_grid / dlaf_finalize): the creator of a host
 * grid releases there what its callbacks use (the MPI shim frees its communicators).  -1: unknown context. */
DLAF_EXTERN_C int dlaf_mi355x_grid_on_free(int context, void (*fn)(void*), void* user) DLAF_NOEXCEPT;
/* Moves a registered grid to the context number `new_ctx` (0: done, -1: unknown ctx, -2: new_ctx is taken).  The
 * MPI shim uses it for dlaf_create_grid_from_blacs, whose grids are looked up by the caller's BLACS context
 * (reference: src/c_api/grid.cpp:73-92 registers the grid under blacs_ctxt). */
DLAF_EXTERN_C int dlaf_mi355x_grid_rekey(int ctx, int new_ctx) DLAF_NOEXCEPT;

/* Communication log of a grid (test instrument): while enabled, every broadcast / barrier / all-reduce the
 * library issues on this grid and a marker per factorization step are appended to a per-process list of
 * events {kind, root, bytes, grouped}: kind 0 row broadcast, 1 column broadcast (root = index inside that
 * communicator), 2 step marker (root = step), 3 barrier, 4 all-reduce.  Every member of a communicator must
 * log the same sequence for it -- the property the reference's communicator pipeline enforces
 * (sender/transform_mpi.h:60-75).  enable != 0 clears the list and starts recording.  _read copies up to
 * cap_events events (4 longs each) and returns the number recorded. */
DLAF_EXTERN_C int dlaf_mi355x_grid_comm_log(int context, int enable) DLAF_NOEXCEPT;
DLAF_EXTERN_C long dlaf_mi355x_grid_comm_log_read(int context, long* out_4_longs_per_event,
                                                  long cap_events) DLAF_NOEXCEPT;

/* my coordinates in a grid; returns 0, or -1 for an unknown context */
DLAF_EXTERN_C int dlaf_mi355x_grid_info(int context, int* nprow, int* npcol, int* myrow, int* mycol) DLAF_NOEXCEPT;

/* ---- device-resident matrices -------------------------------------------------------------- */
typedef struct dlaf_mi355x_matrix_s* dlaf_mi355x_matrix_t;

/* desc.ld is ignored here (the device copy is in tile layout) */
DLAF_EXTERN_C int dlaf_mi355x_matrix_create(int context, char type, char uplo, struct DLAF_descriptor desc,
                                            dlaf_mi355x_matrix_t* out) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_matrix_destroy(dlaf_mi355x_matrix_t m) DLAF_NOEXCEPT;
/* host_local: this process's local column-major array with leading dimension ld */
DLAF_EXTERN_C int dlaf_mi355x_matrix_upload(dlaf_mi355x_matrix_t m, const void* host_local, int ld) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_matrix_download(dlaf_mi355x_matrix_t m, void* host_local, int ld) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_matrix_copy(dlaf_mi355x_matrix_t dst, dlaf_mi355x_matrix_t src) DLAF_NOEXCEPT;
/* one global tile (gi, gj) of the device copy into a dense host array (column-major, ld >= tile rows), as the
 * caller's matrix stores it.  Returns 0, 1 when this process does not own the tile, < 0 on a bad handle.
 * Lets a checker sample an N = 65536 factor without moving 32 GiB. */
DLAF_EXTERN_C int dlaf_mi355x_matrix_fetch_tile(dlaf_mi355x_matrix_t m, long gi, long gj, void* host,
                                                int ld) DLAF_NOEXCEPT;
/* enqueue the factorization / wait for it (returns info) / both */
DLAF_EXTERN_C int dlaf_mi355x_cholesky_start(dlaf_mi355x_matrix_t m) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_cholesky_wait(dlaf_mi355x_matrix_t m) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_cholesky_factorization_device(dlaf_mi355x_matrix_t m) DLAF_NOEXCEPT;
/* This process's OWN device status word of the last wait / factorization, before the grid agreed on one value
 * (the reference aborts every rank, src/cusolver/assert_info.cu:35-45; here the smallest positive LAPACK index over
 * the grid is returned everywhere).  A test uses it to assert that the owner of a non-SPD diagonal tile flagged it. */
DLAF_EXTERN_C int dlaf_mi355x_matrix_local_info(dlaf_mi355x_matrix_t m) DLAF_NOEXCEPT;
/* Trailing-update launches of this process so far that ran in persistent form (work items pulled from queues, workgroup
 * slots left to the kernels beside them) and, of those, the ones that vacated exclusive compute units -- lets a test
 * assert that a reservation took the path it was meant to take. */
DLAF_EXTERN_C int dlaf_mi355x_update_launch_stats(long* persistent, long* exclusive) DLAF_NOEXCEPT;
/* Diagnosis hook (DLAF_MI355X_POTRF_TRACE=1): what the first two strips of the last first-diagonal-tile POTRF of
 * this process saw (32 words, kernels_potrf_coop.hip); returns 0, or 1 when tracing is off. */
DLAF_EXTERN_C int dlaf_mi355x_potrf_trace(unsigned long long* out_32_words) DLAF_NOEXCEPT;
/* Residual checker of the miniapp (miniapp/miniapp_cholesky.cpp:408-443 check_cholesky with
 * auxiliary/norm max_norm) on the device: `original` holds the input matrix and is OVERWRITTEN with
 * A - L L^H (uplo triangle); *max_diff = max|A - L L^H|, *max_a = max|A| over the whole grid (MAX-reduced
 * through the grid's transport).  The strict upper part of the factor's diagonal tiles is zeroed on the
 * device (it is never written back to the caller).  Collective. */
DLAF_EXTERN_C int dlaf_mi355x_cholesky_residual(dlaf_mi355x_matrix_t original, dlaf_mi355x_matrix_t factor,
                                                double* max_diff, double* max_a) DLAF_NOEXCEPT;
/* Live timing of the last factorization, measured with HIP events on the stream each launch class
 * runs on.  kind 0: grouped trailing update (herk+gemm of columns > k+1), 1: lookahead-column update,
 * 2: panel TRSM, 3: diagonal-tile POTRF chain.  ms = summed launch durations, flops / bytes = summed
 * ALGORITHMIC work of those launches (BASELINE.md roofline table).  Call after *_wait. */
DLAF_EXTERN_C int dlaf_mi355x_matrix_profile(dlaf_mi355x_matrix_t m, int kind, double* ms, long* launches,
                                             double* flops, double* bytes) DLAF_NOEXCEPT;
/* The panel TRSM of step 0 ALONE on the device: `reps` launches on a copy of the factored first tile column with
 * the factored diagonal tile, timed with HIP events (in the factorization the panel solves run beside the bulk
 * update on a few free workgroup slots; their in-situ durations do not describe the kernel).  One-process grids,
 * after a factorization.  *ms = average per launch, *flops / *bytes = algorithmic work of one launch. */
DLAF_EXTERN_C int dlaf_mi355x_matrix_trsm_profile(dlaf_mi355x_matrix_t m, int reps, double* ms, double* flops,
                                                  double* bytes) DLAF_NOEXCEPT;
/* barrier over the matrix's grid (RCCL all-reduce / host callback) */
DLAF_EXTERN_C int dlaf_mi355x_grid_barrier(int context) DLAF_NOEXCEPT;
/* Collective communication self-test of a grid (what test/unit/communication/test_broadcast*.cpp do for
 * the reference's MPI layer): every member of every row / column communicator broadcasts `bytes` of a
 * coordinate-dependent pattern in place, out of place and grouped, then barrier + max-allreduce.
 * Returns the number of failed checks on this process (0 = good), -1 for an unknown context.
 * DLAF_MI355X_RCCL_SINGLE=1 makes create_grid_rccl build communicators for a 1-process grid too. */
DLAF_EXTERN_C int dlaf_mi355x_grid_selftest(int context, size_t bytes) DLAF_NOEXCEPT;

/* ---- triangular solver ------------------------------------------------------- */
/* dlaf::triangular_solver(grid, side, uplo, op, diag, alpha, A, B), include/dlaf/solver/triangular.h:41-177
 * (the reference has no C entry for it; the p?trsm names take ScaLAPACK's argument list):
 *   side 'L': op(A) X = alpha B,  side 'R': X op(A) = alpha B;  B (m x n) is overwritten by X.
 * A: na x na triangular (na = m for 'L', n for 'R'), only the uplo triangle is read (diag 'U': its diagonal
 * is taken as 1), op in N/T/C, alpha passed by address.  a, b: local column-major parts on the grid of
 * `context`.  Requirements of this build: square blocks, B's block = A's block, no sub-matrix offsets, A and
 * B share the source process along the triangular dimension.  Returns 0; bad arguments terminate like the
 * reference's DLAF_ASSERTs. */
DLAF_EXTERN_C int dlaf_mi355x_triangular_solver_s(int context, char side, char uplo, char op, char diag,
                                                  const float* alpha, const float* a, struct DLAF_descriptor desca,
                                                  float* b, struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_triangular_solver_d(int context, char side, char uplo, char op, char diag,
                                                  const double* alpha, const double* a, struct DLAF_descriptor desca,
                                                  double* b, struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_triangular_solver_c(int context, char side, char uplo, char op, char diag,
                                                  const dlaf_complex_c* alpha, const dlaf_complex_c* a,
                                                  struct DLAF_descriptor desca, dlaf_complex_c* b,
                                                  struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_triangular_solver_z(int context, char side, char uplo, char op, char diag,
                                                  const dlaf_complex_z* alpha, const dlaf_complex_z* a,
                                                  struct DLAF_descriptor desca, dlaf_complex_z* b,
                                                  struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pstrsm(char side, char uplo, char op, char diag, int m, int n, const float* alpha,
                                      const float* a, int ia, int ja, const int desca[9], float* b, int ib, int jb,
                                      const int descb[9]) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pdtrsm(char side, char uplo, char op, char diag, int m, int n, const double* alpha,
                                      const double* a, int ia, int ja, const int desca[9], double* b, int ib, int jb,
                                      const int descb[9]) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pctrsm(char side, char uplo, char op, char diag, int m, int n,
                                      const dlaf_complex_c* alpha, const dlaf_complex_c* a, int ia, int ja,
                                      const int desca[9], dlaf_complex_c* b, int ib, int jb,
                                      const int descb[9]) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pztrsm(char side, char uplo, char op, char diag, int m, int n,
                                      const dlaf_complex_z* alpha, const dlaf_complex_z* a, int ia, int ja,
                                      const int desca[9], dlaf_complex_z* b, int ib, int jb,
                                      const int descb[9]) DLAF_NOEXCEPT;

/* ScaLAPACK p?potrs: A X = B with the factor dlaf_p?potrf left in a (two triangular solves; b is overwritten) */
DLAF_EXTERN_C void dlaf_mi355x_pspotrs(char uplo, int n, int nrhs, const float* a, int ia, int ja, const int desca[9],
                                       float* b, int ib, int jb, const int descb[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pdpotrs(char uplo, int n, int nrhs, const double* a, int ia, int ja, const int desca[9],
                                       double* b, int ib, int jb, const int descb[9], int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pcpotrs(char uplo, int n, int nrhs, const dlaf_complex_c* a, int ia, int ja,
                                       const int desca[9], dlaf_complex_c* b, int ib, int jb, const int descb[9],
                                       int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pzpotrs(char uplo, int n, int nrhs, const dlaf_complex_z* a, int ia, int ja,
                                       const int desca[9], dlaf_complex_z* b, int ib, int jb, const int descb[9],
                                       int* info) DLAF_NOEXCEPT;

/* ---- generalized -> standard eigenproblem ------------------------------------------- */
/* dlaf::eigensolver::internal::generalized_to_standard(grid, uplo, A, B), include/dlaf/eigensolver/gen_to_std.h:50,
 * :101 (LAPACK xHEGST itype 1; the reference has no C entry for it):  A <- inv(L) A inv(L^H) (uplo 'L') or
 * inv(U^H) A inv(U) (uplo 'U'), where b holds the Cholesky factor of B in its uplo triangle (dlaf_p?potrf output).
 * Only the uplo triangles are read / written; b is not modified.  A and B must be distributed alike (size, square
 * block, source process).  Returns 0.  The p?hegst names take ScaLAPACK's p?sygst / p?hegst argument list
 * (ibtype must be 1; *scale is set to 1). */
DLAF_EXTERN_C int dlaf_mi355x_generalized_to_standard_s(int context, char uplo, float* a, struct DLAF_descriptor desca,
                                                        const float* b, struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_generalized_to_standard_d(int context, char uplo, double* a, struct DLAF_descriptor desca,
                                                        const double* b, struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_generalized_to_standard_c(int context, char uplo, dlaf_complex_c* a,
                                                        struct DLAF_descriptor desca, const dlaf_complex_c* b,
                                                        struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_generalized_to_standard_z(int context, char uplo, dlaf_complex_z* a,
                                                        struct DLAF_descriptor desca, const dlaf_complex_z* b,
                                                        struct DLAF_descriptor descb) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pshegst(int ibtype, char uplo, int n, float* a, int ia, int ja, const int desca[9],
                                       const float* b, int ib, int jb, const int descb[9], float* scale,
                                       int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pdhegst(int ibtype, char uplo, int n, double* a, int ia, int ja, const int desca[9],
                                       const double* b, int ib, int jb, const int descb[9], double* scale,
                                       int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pchegst(int ibtype, char uplo, int n, dlaf_complex_c* a, int ia, int ja,
                                       const int desca[9], const dlaf_complex_c* b, int ib, int jb,
                                       const int descb[9], float* scale, int* info) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_pzhegst(int ibtype, char uplo, int n, dlaf_complex_z* a, int ia, int ja,
                                       const int desca[9], const dlaf_complex_z* b, int ib, int jb,
                                       const int descb[9], double* scale, int* info) DLAF_NOEXCEPT;
/* the same on device-resident matrices (both created with the same uplo on the same grid) */
DLAF_EXTERN_C int dlaf_mi355x_generalized_to_standard_device(dlaf_mi355x_matrix_t a,
                                                             dlaf_mi355x_matrix_t cholesky_factor_of_b) DLAF_NOEXCEPT;

/* ---- reduction to band + back-transformation (first stage of the eigensolver) ----------- */
/* dlaf::eigensolver::internal::reduction_to_band<B, D, T>(grid, mat_a, band_size)
 * (include/dlaf/eigensolver/reduction_to_band.h:40-122; the reference has no C entry for the stage alone):
 * Q^H A Q = B with B Hermitian band (main diagonal + band_size sub-diagonals).  a: this process's local
 * column-major part of the Hermitian matrix; only its LOWER triangle is referenced and overwritten -- with the band
 * and, below it, the Householder reflectors (the layout of reduction_to_band.h:77-96); the strict upper triangle is
 * untouched.  band_size >= 2 must divide the (square) block size.  taus: n - band_size - 1 scalar factors, ALL of
 * them on every process (the reference returns them distributed over the process columns, replicated over the
 * rows; entry j belongs to the reflector in global column j).  Returns 0. */
DLAF_EXTERN_C int dlaf_mi355x_reduction_to_band_s(int context, float* a, struct DLAF_descriptor desca, int band_size,
                                                  float* taus) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_reduction_to_band_d(int context, double* a, struct DLAF_descriptor desca, int band_size,
                                                  double* taus) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_reduction_to_band_c(int context, dlaf_complex_c* a, struct DLAF_descriptor desca,
                                                  int band_size, dlaf_complex_c* taus) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_reduction_to_band_z(int context, dlaf_complex_z* a, struct DLAF_descriptor desca,
                                                  int band_size, dlaf_complex_z* taus) DLAF_NOEXCEPT;
/* the same on a device-resident matrix (created with uplo 'L'); taus: host array as above (may be NULL) */
DLAF_EXTERN_C int dlaf_mi355x_reduction_to_band_device(dlaf_mi355x_matrix_t a, int band_size, void* taus) DLAF_NOEXCEPT;
/* dlaf::eigensolver::internal::bt_reduction_to_band(grid, band_size, mat_c, mat_v, mat_taus)
 * (include/dlaf/eigensolver/bt_reduction_to_band.h): C <- Q C with the reflectors reduction_to_band left in v (lower
 * triangle, below the band) and its taus.  c: n x k general matrix with v's block size and row distribution. */
DLAF_EXTERN_C int dlaf_mi355x_bt_reduction_to_band_s(int context, int band_size, float* c, struct DLAF_descriptor descc,
                                                     const float* v, struct DLAF_descriptor descv,
                                                     const float* taus) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_bt_reduction_to_band_d(int context, int band_size, double* c, struct DLAF_descriptor descc,
                                                     const double* v, struct DLAF_descriptor descv,
                                                     const double* taus) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_bt_reduction_to_band_c(int context, int band_size, dlaf_complex_c* c,
                                                     struct DLAF_descriptor descc, const dlaf_complex_c* v,
                                                     struct DLAF_descriptor descv, const dlaf_complex_c* taus) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_bt_reduction_to_band_z(int context, int band_size, dlaf_complex_z* c,
                                                     struct DLAF_descriptor descc, const dlaf_complex_z* v,
                                                     struct DLAF_descriptor descv, const dlaf_complex_z* taus) DLAF_NOEXCEPT;
/* Panels of this process's last reduction_to_band factored by the blocked path (CholeskyQR2 + Householder
 * reconstruction, csrc/device/kernels_hr.hip) / handed back to the reflector-by-reflector kernel by its gate. */
DLAF_EXTERN_C int dlaf_mi355x_red2band_panel_stats(long* blocked, long* fallback) DLAF_NOEXCEPT;
/* Gives the idle blocks of the workspace pool back to the driver (the stages' temporaries of 4 MiB and more are kept between
 * calls, up to DLAF_MI355X_POOL_GB = 64 GiB; the analogue of the reference's Umpire pools, src/memory/memory_chunk.cpp,
 * which dlaf::finalize releases).  Returns the bytes that were held. */
DLAF_EXTERN_C long dlaf_mi355x_workspace_pool_release(void) DLAF_NOEXCEPT;
/* Tune parameter eigensolver_min_band (include/dlaf/tune.h:71-75,128; default 100).  dlaf_initialize reads
 * DLAF_EIGENSOLVER_MIN_BAND and --dlaf:eigensolver-min-band like src/init.cpp:220; the setter is what the reference's
 * tests do with getTuneParameters().eigensolver_min_band (test/unit/eigensolver/test_eigensolver.cpp:142). b_min >= 2. */
DLAF_EXTERN_C int dlaf_mi355x_get_eigensolver_min_band(void) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_set_eigensolver_min_band(int b_min) DLAF_NOEXCEPT;
/* get_band_size (include/dlaf/eigensolver/internal/get_band_size.h:20-31, tune parameter eigensolver_min_band): the band the
 * reference's eigensolver picks for a block size (128 for nb = 512) */
DLAF_EXTERN_C int dlaf_mi355x_get_band_size(int nb) DLAF_NOEXCEPT;
/* device time (ms, HIP events) and whole-grid flops in the reference miniapps' models (2 (2/3 n^3 - n^2 nb), resp.
 * 2 (n - band)^2 k; x4 complex: miniapp_reduction_to_band.cpp:163-168, miniapp_bt_reduction_to_band.cpp:160-164) of the last
 * reduction_to_band / bt_reduction_to_band on this process */
DLAF_EXTERN_C int dlaf_mi355x_red2band_profile(double* ms, double* flops) DLAF_NOEXCEPT;

/* ---- the eigensolver stages behind reduction_to_band -------------------------------------- */
/* dlaf::eigensolver::internal::band_to_tridiagonal<Backend::MC>(grid, uplo = Lower, band_size, mat_a)
 * (include/dlaf/eigensolver/band_to_tridiag.h:74-97, :155-176; the reference has no C entry for the stage): a holds a
 * Hermitian band matrix in the lower band of its local part (what reduction_to_band leaves; everything below the band
 * is ignored).  d: n diagonal entries, e: n - 1 off-diagonal entries (real), v: n x n (ld ldv) compact Householder
 * reflectors with tau in the place of the leading 1, laid out as band_to_tridiag.h:40-72 says.  The outputs are complete
 * on every process. */
DLAF_EXTERN_C int dlaf_mi355x_band_to_tridiagonal_s(int context, const float* a, struct DLAF_descriptor desca, int band_size,
                                                  float* d, float* e, float* v, int ldv) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_band_to_tridiagonal_d(int context, const double* a, struct DLAF_descriptor desca, int band_size,
                                                  double* d, double* e, double* v, int ldv) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_band_to_tridiagonal_c(int context, const dlaf_complex_c* a, struct DLAF_descriptor desca, int band_size,
                                                  float* d, float* e, dlaf_complex_c* v, int ldv) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_band_to_tridiagonal_z(int context, const dlaf_complex_z* a, struct DLAF_descriptor desca, int band_size,
                                                  double* d, double* e, dlaf_complex_z* v, int ldv) DLAF_NOEXCEPT;
/* dlaf::eigensolver::internal::bt_band_to_tridiagonal(band_size, mat_e, mat_hh)
 * (include/dlaf/eigensolver/bt_band_to_tridiag.h:28-61), local form: e (n x ncols, column-major, ld lde) <- Q e with the
 * reflectors v as band_to_tridiagonal returned them */
DLAF_EXTERN_C int dlaf_mi355x_bt_band_to_tridiagonal_s(int band_size, int n, int ncols, const float* v, int ldv, float* e,
                                                     int lde) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_bt_band_to_tridiagonal_d(int band_size, int n, int ncols, const double* v, int ldv, double* e,
                                                     int lde) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_bt_band_to_tridiagonal_c(int band_size, int n, int ncols, const dlaf_complex_c* v, int ldv, dlaf_complex_c* e,
                                                     int lde) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_bt_band_to_tridiagonal_z(int band_size, int n, int ncols, const dlaf_complex_z* v, int ldv, dlaf_complex_z* e,
                                                     int lde) DLAF_NOEXCEPT;
/* dlaf::eigensolver::internal::tridiagonal_eigensolver (include/dlaf/eigensolver/tridiag_solver.h:30-60), local form:
 * d (n), e (n - 1) -> eigenvalues w (ascending) and eigenvectors z (n x n, column-major, ld ldz) by divide & conquer on
 * the GPU.  nb: the reference's block size (unused by the device algorithm, kept for the call shape). */
DLAF_EXTERN_C int dlaf_mi355x_tridiagonal_eigensolver_s(int n, int nb, const float* d, const float* e, float* w, float* z,
                                                        int ldz) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_tridiagonal_eigensolver_d(int n, int nb, const double* d, const double* e, double* w,
                                                        double* z, int ldz) DLAF_NOEXCEPT;
/* device time (ms) per stage of the last eigensolver call on this process: reduction_to_band, band_to_tridiagonal,
 * tridiagonal_eigensolver, bt_band_to_tridiagonal, bt_reduction_to_band */
DLAF_EXTERN_C int dlaf_mi355x_eigensolver_profile(double ms[5]) DLAF_NOEXCEPT;

/* Device-resident operands: a general m x n matrix in HBM (tile layout; square blocks) as the right-hand side,
 * a dlaf_mi355x_matrix_t (the uplo triangle of a resident matrix, e.g. the factor dlaf_mi355x_cholesky_* left there)
 * as the triangular matrix.  b is overwritten by the solution; nothing crosses PCIe.  potrs_device = the two solves of
 * A X = B from the resident Cholesky factor.  Same requirements as the host entry. */
typedef struct dlaf_mi355x_gmatrix_s* dlaf_mi355x_gmatrix_t;
DLAF_EXTERN_C int dlaf_mi355x_gmatrix_create(int context, char type, struct DLAF_descriptor desc,
                                             dlaf_mi355x_gmatrix_t* out) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_mi355x_gmatrix_destroy(dlaf_mi355x_gmatrix_t m) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_gmatrix_upload(dlaf_mi355x_gmatrix_t m, const void* host_local, int ld) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_gmatrix_download(dlaf_mi355x_gmatrix_t m, void* host_local, int ld) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_triangular_solver_device(char side, char uplo, char op, char diag, const void* alpha,
                                                       dlaf_mi355x_matrix_t a, dlaf_mi355x_gmatrix_t b) DLAF_NOEXCEPT;
DLAF_EXTERN_C int dlaf_mi355x_potrs_device(char uplo, dlaf_mi355x_matrix_t factor, dlaf_mi355x_gmatrix_t b) DLAF_NOEXCEPT;
/* bt_reduction_to_band on resident operands: c (general matrix) <- Q c, reflectors in v (uplo 'L'), taus on the host */
DLAF_EXTERN_C int dlaf_mi355x_bt_reduction_to_band_device(int band_size, dlaf_mi355x_gmatrix_t c, dlaf_mi355x_matrix_t v,
                                                          const void* taus) DLAF_NOEXCEPT;

/* Device time (ms, HIP events on the compute stream) of the sweep of the last triangular solve on this process
 * -- relayout and PCIe staging excluded -- and the whole-grid algorithmic flops it stands for (m n^2 for side
 * R, m^2 n for side L; x4 complex). */
DLAF_EXTERN_C int dlaf_mi355x_solver_profile(double* ms, double* flops) DLAF_NOEXCEPT;

/* ---- synthetic input ------------------------------------------------------------------------ */
/* Fills this process's local array (column-major, ld) of the n x n matrix with the reference's
 * random Hermitian positive definite matrix: per global tile a std::mt19937_64 seeded with the
 * tile's origin, uniform(-1,1) off the diagonal, 2n added on it (util_matrix.h:323-442,498-501).
 * nthreads <= 0: all hardware threads. */
DLAF_EXTERN_C int dlaf_mi355x_set_random_hpd(int context, char type, void* host_local,
                                             struct DLAF_descriptor desc, int nthreads) DLAF_NOEXCEPT;

/* ---- tile operations (host operands, column-major) -------------------------------------------- */
/* potrf: returns LAPACK info (reference: lapack/tile.h:362-378) */
DLAF_EXTERN_C int dlaf_mi355x_tile_potrf(char type, char uplo, int n, void* a, int lda) DLAF_NOEXCEPT;
/* trsm: uplo L: B(m x n) <- B A^-H, A n x n lower;  uplo U: B <- A^-H B, A m x m upper
 * (Right/Lower/ConjTrans and Left/Upper/ConjTrans, NonUnit, alpha 1: impl.h:56-67,:110-121) */
DLAF_EXTERN_C int dlaf_mi355x_tile_trsm(char type, char uplo, int m, int n, const void* a, int lda, void* b,
                                        int ldb) DLAF_NOEXCEPT;
/* herk: uplo L: lower(C) -= A A^H, A n x k;  uplo U: upper(C) -= A^H A, A k x n (impl.h:70-80,:124-134) */
DLAF_EXTERN_C int dlaf_mi355x_tile_herk(char type, char uplo, int n, int k, const void* a, int lda, void* c,
                                        int ldc) DLAF_NOEXCEPT;
/* gemm: uplo L: C(m x n) -= A B^H, A m x k, B n x k;  uplo U: C -= A^H B, A k x m, B k x n
 * (impl.h:83-94,:137-147) */
DLAF_EXTERN_C int dlaf_mi355x_tile_gemm(char type, char uplo, int m, int n, int k, const void* a, int lda,
                                        const void* b, int ldb, void* c, int ldc) DLAF_NOEXCEPT;

/* ---- index helpers (no GPU needed) -------------------------------------------------------------- */
DLAF_EXTERN_C int dlaf_mi355x_dist_owner(long global_tile, int grid_size, int src_rank) DLAF_NOEXCEPT;
DLAF_EXTERN_C long dlaf_mi355x_dist_local_tile(long global_tile, int grid_size, int rank, int src_rank) DLAF_NOEXCEPT;
DLAF_EXTERN_C long dlaf_mi355x_dist_next_local_tile(long global_tile, int grid_size, int rank,
                                                    int src_rank) DLAF_NOEXCEPT;
DLAF_EXTERN_C long dlaf_mi355x_dist_global_tile(long local_tile, int grid_size, int rank, int src_rank) DLAF_NOEXCEPT;
DLAF_EXTERN_C long dlaf_mi355x_dist_local_size(long n, int nb, int grid_size, int rank, int src_rank) DLAF_NOEXCEPT;
DLAF_EXTERN_C long dlaf_mi355x_dist_local_tiles(long n, int nb, int grid_size, int rank, int src_rank) DLAF_NOEXCEPT;

/* library identification: "dlaf_mi355x <version> gfx950" */
DLAF_EXTERN_C const char* dlaf_mi355x_version(void) DLAF_NOEXCEPT;
