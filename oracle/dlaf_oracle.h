/* dlaf_oracle.h -- TEST INFRASTRUCTURE ONLY: C interface of the CPU oracle (see dlaf_oracle.c). */
#pragma once
#include <complex.h>
#include <stdint.h>

/* index math: include/dlaf/matrix/util_distribution.h:29-196 */
long orc_tile_from_element(long element, long tile_size, long tile_el_offset);
long orc_tile_element_from_element(long element, long tile_size, long tile_el_offset);
long orc_element_from_tile_and_tile_element(long tile, long tile_element, long tile_size, long tile_el_offset);
int orc_rank_global_tile(long global_tile, long tiles_per_block, int grid_size, int src_rank, long tile_offset);
long orc_local_tile_from_global_tile(long global_tile, long tiles_per_block, int grid_size, int rank,
                                     int src_rank, long tile_offset);
long orc_next_local_tile_from_global_tile(long global_tile, long tiles_per_block, int grid_size, int rank,
                                          int src_rank, long tile_offset);
long orc_global_tile_from_local_tile(long local_tile, long tiles_per_block, int grid_size, int rank,
                                     int src_rank, long tile_offset);
long orc_local_nr_tiles(long n, long nb, int grid_size, int rank, int src_rank);
long orc_local_size(long n, long nb, int grid_size, int rank, int src_rank);

/* RNG: std::mt19937_64 + libstdc++ uniform_real_distribution(-1,1) */
typedef struct {
  uint64_t mt[312];
  int idx;
} orc_mt19937_64;
void orc_mt_seed(orc_mt19937_64* g, uint64_t seed);
uint64_t orc_mt_next(orc_mt19937_64* g);
double orc_uniform_pm1_d(orc_mt19937_64* g);
float orc_uniform_pm1_s(orc_mt19937_64* g);

#define ORC_DECL(T, R, S)                                                                                  \
  int orc_potrf_##S(char uplo, int n, T* a, int lda);                                                      \
  void orc_trsm_##S(char side, char uplo, char op, char diag, int m, int n, T alpha, const T* a, int lda, \
                    T* b, int ldb);                                                                        \
  void orc_herk_##S(char uplo, char op, int n, int k, R alpha, const T* a, int lda, R beta, T* c, int ldc); \
  void orc_gemm_##S(char opa, char opb, int m, int n, int k, T alpha, const T* a, int lda, const T* b,    \
                    int ldb, T beta, T* c, int ldc);                                                       \
  T orc_chol_el_a_##S(char uplo, long i, long j);                                                          \
  T orc_chol_el_l_##S(char uplo, long i, long j);                                                          \
  void orc_set_random_hpd_tile_##S(long n, int nb, long ti, long tj, R offset, T* tile, int ldt);          \
  void orc_set_random_hpd_##S(long n, int nb, T* a, long lda);                                             \
  int orc_cholesky_local_##S(char uplo, long n, int nb, T* a, long lda);                                   \
  int orc_cholesky_dist_##S(char uplo, long n, int nb, int pr, int pc, int sr, int sc, T** loc,           \
                            const long* lld);

ORC_DECL(float, float, s)
ORC_DECL(double, double, d)
ORC_DECL(float complex, float, c)
ORC_DECL(double complex, double, z)
#undef ORC_DECL

/* multithreaded fp64 tile-DAG baseline (lower) for bench.py's cpu_baseline leg */
int orc_baseline_cholesky_d(long n, int nb, double* a, long lda, int nthreads);
int orc_omp_max_threads(void);
