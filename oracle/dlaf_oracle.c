/*
 * dlaf_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's tiled Cholesky hot path
 * (aurianer/DLA-Future v0.6.0).  It exists to CHECK the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * library (dla_future_amd/lib/libdlaf_mi355x.so) never links, loads or calls anything
 * in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py) against
 *   - the 32 golden rows of test/unit/matrix/test_util_distribution.cpp:45-62,
 *   - the reference's own util_distribution.h compiled where it lies (oracle/_ref),
 *   - the analytic known-answer matrices of getCholeskySetters
 *     (test/include/dlaf_test/matrix/util_generic_lapack.h:39-68) with the tolerances of
 *     test/unit/factorization/test_cholesky.cpp:76-77,
 *   - the closed forms of test_blas_tile/test_{gemm,herk,trsm}.h,
 *   - libstdc++'s mt19937_64 / uniform_real_distribution stream (tests/golden/rng_stream.json),
 *   - LAPACK ?potrf through scipy (independent numeric cross-check).
 * The arithmetic of the reference's CPU path lives in blaspp/lapackpp -> vendor
 * BLAS/LAPACK (spack pins blaspp@2022.05.00:, lapackpp@2022.05.00:,
 * spack/packages/dla-future/package.py:72-78), absent from /root/reference; the
 * published BLAS/LAPACK reference algorithms (xPOTF2, xTRSM, xHERK/xSYRK, xGEMM) are
 * restated in oracle_kernels.inc.
 *
 * Citations "file:line" are relative to /root/reference.
 */
#define _GNU_SOURCE
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dlaf_oracle.h"

/* ====================================================================== index math
 * include/dlaf/matrix/util_distribution.h:29-196.  SizeType = ptrdiff_t -> long.   */
long orc_tile_from_element(long element, long tile_size, long tile_el_offset) {
  return (element + tile_el_offset) / tile_size; /* :29-34 */
}

long orc_tile_element_from_element(long element, long tile_size, long tile_el_offset) {
  element += tile_el_offset; /* :42-53 */
  long tile_element = element % tile_size;
  if (element < tile_size)
    tile_element -= tile_el_offset;
  return tile_element;
}

long orc_element_from_tile_and_tile_element(long tile, long tile_element, long tile_size,
                                            long tile_el_offset) {
  return tile * tile_size + tile_element - (tile > 0 ? tile_el_offset : 0); /* :64-73 */
}

int orc_rank_global_tile(long global_tile, long tiles_per_block, int grid_size, int src_rank,
                         long tile_offset) {
  long global_block = (global_tile + tile_offset) / tiles_per_block; /* :82-92 */
  return (int) ((global_block + src_rank) % grid_size);
}

long orc_local_tile_from_global_tile(long global_tile, long tiles_per_block, int grid_size, int rank,
                                     int src_rank, long tile_offset) {
  if (rank != orc_rank_global_tile(global_tile, tiles_per_block, grid_size, src_rank, tile_offset))
    return -1; /* :103-127 */
  global_tile += tile_offset;
  long local_block = global_tile / tiles_per_block / grid_size;
  int partial_first = (rank == src_rank);
  return local_block * tiles_per_block + global_tile % tiles_per_block - (partial_first ? tile_offset : 0);
}

long orc_next_local_tile_from_global_tile(long global_tile, long tiles_per_block, int grid_size, int rank,
                                          int src_rank, long tile_offset) {
  int rank_to_src = (rank + grid_size - src_rank) % grid_size; /* :139-167 */
  global_tile += tile_offset;
  long global_block = global_tile / tiles_per_block;
  long owner_to_src = global_block % grid_size;
  long local_block = global_block / grid_size;
  int partial_first = (rank == src_rank);
  if (rank_to_src == owner_to_src)
    return local_block * tiles_per_block + global_tile % tiles_per_block - (partial_first ? tile_offset : 0);
  if (rank_to_src < owner_to_src)
    ++local_block;
  return local_block * tiles_per_block - (partial_first ? tile_offset : 0);
}

long orc_global_tile_from_local_tile(long local_tile, long tiles_per_block, int grid_size, int rank,
                                     int src_rank, long tile_offset) {
  int partial_first = (rank == src_rank); /* :178-196 */
  if (partial_first)
    local_tile += tile_offset;
  int rank_to_src = (rank + grid_size - src_rank) % grid_size;
  long local_block = local_tile / tiles_per_block;
  return (grid_size * local_block + rank_to_src) * tiles_per_block + local_tile % tiles_per_block - tile_offset;
}

/* Local extent along one axis of an n-element axis cut in nb-tiles over grid_size ranks.
 * src/matrix/distribution.cpp:118-151 (compute_local_nr_tiles_and_local_size /
 * compute_local_size) specialised to offset 0, tiles_per_block 1 -- the only case
 * the Cholesky path accepts (cholesky.h:39-79 preconditions).                       */
long orc_local_nr_tiles(long n, long nb, int grid_size, int rank, int src_rank) {
  long nt = n > 0 ? (n + nb - 1) / nb : 0;
  return orc_next_local_tile_from_global_tile(nt, 1, grid_size, rank, src_rank, 0);
}

long orc_local_size(long n, long nb, int grid_size, int rank, int src_rank) {
  long nt = n > 0 ? (n + nb - 1) / nb : 0;
  long lnt = orc_local_nr_tiles(n, nb, grid_size, rank, src_rank);
  if (lnt == 0)
    return 0;
  long ret = lnt * nb;
  if (rank == orc_rank_global_tile(nt - 1, 1, grid_size, src_rank, 0))
    ret -= nt * nb - n;
  return ret;
}

/* ====================================================================== RNG
 * std::mt19937_64 (ISO C++ [rand.predef]: w=64 n=312 m=156 r=31
 * a=0xb5026f5aa96619e9 u=29 d=0x5555555555555555 s=17 b=0x71d67fffeda60000 t=37
 * c=0xfff7eee000000000 l=43 f=6364136223846793005), used by getter_random
 * (include/dlaf/util_matrix.h:148-166).                                             */
void orc_mt_seed(orc_mt19937_64* g, uint64_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 312; ++i)
    g->mt[i] = 6364136223846793005ULL * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (uint64_t) i;
  g->idx = 312;
}

uint64_t orc_mt_next(orc_mt19937_64* g) {
  if (g->idx >= 312) {
    const uint64_t UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL, A = 0xB5026F5AA96619E9ULL;
    for (int i = 0; i < 312; ++i) {
      uint64_t x = (g->mt[i] & UM) | (g->mt[(i + 1) % 312] & LM);
      g->mt[i] = g->mt[(i + 156) % 312] ^ (x >> 1) ^ ((x & 1ULL) ? A : 0ULL);
    }
    g->idx = 0;
  }
  uint64_t x = g->mt[g->idx++];
  x ^= (x >> 29) & 0x5555555555555555ULL;
  x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
  x ^= (x << 37) & 0xFFF7EEE000000000ULL;
  x ^= (x >> 43);
  return x;
}

/* std::uniform_real_distribution<T>(-1, 1) as libstdc++ implements it
 * (bits/random.tcc generate_canonical: one 64-bit draw, sum = T(draw), ret = sum / T(2^64),
 * ret >= 1 -> nextafter(1, 0); then ret * (b - a) + a).  libstdc++-specific by nature.  */
double orc_uniform_pm1_d(orc_mt19937_64* g) {
  double ret = (double) orc_mt_next(g) / 18446744073709551616.0;
  if (ret >= 1.0)
    ret = nextafter(1.0, 0.0);
  return ret * 2.0 + -1.0;
}

float orc_uniform_pm1_s(orc_mt19937_64* g) {
  float ret = (float) orc_mt_next(g) / 18446744073709551616.0f;
  if (ret >= 1.0f)
    ret = nextafterf(1.0f, 0.0f);
  return ret * 2.0f + -1.0f;
}

static inline double orc_uniform_pm1_z(orc_mt19937_64* g) {
  return orc_uniform_pm1_d(g);
}
static inline float orc_uniform_pm1_c(orc_mt19937_64* g) {
  return orc_uniform_pm1_s(g);
}

/* ====================================================================== instantiate
 */
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

/* float */
#define T float
#define R float
#define SFX(name) CAT(name, _s)
#define IS_COMPLEX 0
#define CONJ(x) (x)
#define REALP(x) (x)
#define IMAGP(x) (0.0f)
#define MK(re, im) ((float) (re))
#define RSQRT(x) sqrtf(x)
#define RABS(x) fabsf(x)
#define RCOS(x) cosf(x)
#define RSIN(x) sinf(x)
#include "oracle_kernels.inc"
#undef T
#undef R
#undef SFX
#undef IS_COMPLEX
#undef CONJ
#undef REALP
#undef IMAGP
#undef MK
#undef RSQRT
#undef RABS
#undef RCOS
#undef RSIN

/* double */
#define T double
#define R double
#define SFX(name) CAT(name, _d)
#define IS_COMPLEX 0
#define CONJ(x) (x)
#define REALP(x) (x)
#define IMAGP(x) (0.0)
#define MK(re, im) ((double) (re))
#define RSQRT(x) sqrt(x)
#define RABS(x) fabs(x)
#define RCOS(x) cos(x)
#define RSIN(x) sin(x)
#include "oracle_kernels.inc"
#undef T
#undef R
#undef SFX
#undef IS_COMPLEX
#undef CONJ
#undef REALP
#undef IMAGP
#undef MK
#undef RSQRT
#undef RABS
#undef RCOS
#undef RSIN

/* float complex */
#define T float complex
#define R float
#define SFX(name) CAT(name, _c)
#define IS_COMPLEX 1
#define CONJ(x) conjf(x)
#define REALP(x) crealf(x)
#define IMAGP(x) cimagf(x)
#define MK(re, im) CMPLXF((float) (re), (float) (im))
#define RSQRT(x) sqrtf(x)
#define RABS(x) fabsf(x)
#define RCOS(x) cosf(x)
#define RSIN(x) sinf(x)
#include "oracle_kernels.inc"
#undef T
#undef R
#undef SFX
#undef IS_COMPLEX
#undef CONJ
#undef REALP
#undef IMAGP
#undef MK
#undef RSQRT
#undef RABS
#undef RCOS
#undef RSIN

/* double complex */
#define T double complex
#define R double
#define SFX(name) CAT(name, _z)
#define IS_COMPLEX 1
#define CONJ(x) conj(x)
#define REALP(x) creal(x)
#define IMAGP(x) cimag(x)
#define MK(re, im) CMPLX((double) (re), (double) (im))
#define RSQRT(x) sqrt(x)
#define RABS(x) fabs(x)
#define RCOS(x) cos(x)
#define RSIN(x) sin(x)
#include "oracle_kernels.inc"
#undef T
#undef R
#undef SFX
#undef IS_COMPLEX
#undef CONJ
#undef REALP
#undef IMAGP
#undef MK
#undef RSQRT
#undef RABS
#undef RCOS
#undef RSIN

/* ====================================================================== CPU baseline (fp64)
 * The SAME right-looking tile DAG as cholesky/impl.h:150-189, one tile task per host
 * thread with single-threaded tile kernels -- the analogue of the reference's pika
 * default pool with SingleThreadedBlasScope per tile (include/dlaf/blas/tile.h:300,
 * scripts/miniapps.py:219 sets OMP_NUM_THREADS=1 for the BLAS).  The DAG edges the
 * reference gets from per-tile async_rw_mutex (matrix/internal/tile_pipeline.h:36-51)
 * are expressed as OpenMP task dependences on the tile's first element.  Tile kernels
 * are register-blocked C (GCC vector extensions, 4 doubles wide), not a vendor BLAS:
 * this is a "port" baseline, reported as such by bench.py.  Lower, fp64 only.        */
typedef double v4d __attribute__((vector_size(32), aligned(8)));

/* C(m x n) -= A(m x k) * B(n x k)^T ; lower != 0: only i >= j (square case, herk) */
static void base_gemm_nt(int m, int n, int k, const double* restrict a, int lda, const double* restrict b,
                         int ldb, double* restrict c, int ldc, int lower) {
  int j = 0;
  for (; j + 4 <= n; j += 4) {
    int i = lower ? (j & ~7) : 0;
    for (; i + 8 <= m; i += 8) {
      v4d c00 = {0, 0, 0, 0}, c01 = c00, c02 = c00, c03 = c00, c10 = c00, c11 = c00, c12 = c00, c13 = c00;
      const double* ap = a + i;
      const double* bp = b + j;
      for (int l = 0; l < k; ++l) {
        v4d a0 = *(const v4d*) (ap + (size_t) l * lda);
        v4d a1 = *(const v4d*) (ap + (size_t) l * lda + 4);
        const double* bl = bp + (size_t) l * ldb;
        v4d b0 = {bl[0], bl[0], bl[0], bl[0]}, b1 = {bl[1], bl[1], bl[1], bl[1]};
        v4d b2 = {bl[2], bl[2], bl[2], bl[2]}, b3 = {bl[3], bl[3], bl[3], bl[3]};
        c00 += a0 * b0;
        c10 += a1 * b0;
        c01 += a0 * b1;
        c11 += a1 * b1;
        c02 += a0 * b2;
        c12 += a1 * b2;
        c03 += a0 * b3;
        c13 += a1 * b3;
      }
      v4d* acc[4][2] = {{&c00, &c10}, {&c01, &c11}, {&c02, &c12}, {&c03, &c13}};
      for (int jj = 0; jj < 4; ++jj)
        for (int h = 0; h < 2; ++h)
          for (int ii = 0; ii < 4; ++ii) {
            int gi = i + h * 4 + ii, gj = j + jj;
            if (!lower || gi >= gj)
              c[gi + (size_t) gj * ldc] -= (*acc[jj][h])[ii];
          }
    }
    for (; i < m; ++i)
      for (int jj = 0; jj < 4; ++jj) {
        if (lower && i < j + jj)
          continue;
        double s = 0;
        for (int l = 0; l < k; ++l)
          s += a[i + (size_t) l * lda] * b[j + jj + (size_t) l * ldb];
        c[i + (size_t) (j + jj) * ldc] -= s;
      }
  }
  for (; j < n; ++j)
    for (int i = lower ? j : 0; i < m; ++i) {
      double s = 0;
      for (int l = 0; l < k; ++l)
        s += a[i + (size_t) l * lda] * b[j + (size_t) l * ldb];
      c[i + (size_t) j * ldc] -= s;
    }
}

/* blocked lower potrf of one tile: 32-wide panels, trailing update with base_gemm_nt */
static int base_potrf_l(int n, double* a, int lda) {
  const int ib = 32;
  for (int j = 0; j < n; j += ib) {
    int jb = n - j < ib ? n - j : ib;
    int info = orc_potrf_d('L', jb, a + j + (size_t) j * lda, lda);
    if (info)
      return j + info;
    if (j + jb < n) {
      int mrem = n - j - jb;
      orc_trsm_d('R', 'L', 'C', 'N', mrem, jb, 1.0, a + j + (size_t) j * lda, lda,
                 a + j + jb + (size_t) j * lda, lda);
      base_gemm_nt(mrem, mrem, jb, a + j + jb + (size_t) j * lda, lda, a + j + jb + (size_t) j * lda, lda,
                   a + j + jb + (size_t) (j + jb) * lda, lda, 1);
    }
  }
  return 0;
}

/* B(m x n) <- B L^-T, blocked: column blocks of 32 solved with the unblocked kernel,
 * left-looking update with base_gemm_nt */
static void base_trsm_rltn(int m, int n, const double* l, int ldl, double* b, int ldb) {
  const int ib = 32;
  for (int j = 0; j < n; j += ib) {
    int jb = n - j < ib ? n - j : ib;
    if (j > 0)
      base_gemm_nt(m, jb, j, b, ldb, l + j, ldl, b + (size_t) j * ldb, ldb, 0);
    orc_trsm_d('R', 'L', 'C', 'N', m, jb, 1.0, l + j + (size_t) j * ldl, ldl, b + (size_t) j * ldb, ldb);
  }
}

int orc_baseline_cholesky_d(long n, int nb, double* a, long lda, int nthreads) {
  const long nt = n > 0 ? (n + nb - 1) / nb : 0;
  int result = 0;
  (void) nthreads;
#define TSZ(t) ((int) (((t) + 1) * (long) nb <= n ? nb : n - (t) * (long) nb))
#define TPT(i, j) (a + (i) * (long) nb + (j) * (long) nb * lda)
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#pragma omp single
#endif
  {
    for (long k = 0; k < nt; ++k) {
      double* kk = TPT(k, k);
      const int kb = TSZ(k);
#pragma omp task depend(inout : kk[0]) shared(result) priority(2)
      {
        int info = base_potrf_l(kb, kk, (int) lda);
        if (info && !result)
          result = (int) (k * nb + info);
      }
      for (long i = k + 1; i < nt; ++i) {
        double* ik = TPT(i, k);
        const int ib = TSZ(i);
#pragma omp task depend(in : kk[0]) depend(inout : ik[0]) priority(2)
        base_trsm_rltn(ib, kb, kk, (int) lda, ik, (int) lda);
      }
      for (long j = k + 1; j < nt; ++j) {
        double* jk = TPT(j, k);
        double* jj = TPT(j, j);
        const int jb = TSZ(j);
        const int prio = (j == k + 1) ? 1 : 0; /* lookahead rule, impl.h:172-173 */
#pragma omp task depend(in : jk[0]) depend(inout : jj[0]) priority(prio)
        base_gemm_nt(jb, jb, kb, jk, (int) lda, jk, (int) lda, jj, (int) lda, 1);
        for (long i = j + 1; i < nt; ++i) {
          double* ik = TPT(i, k);
          double* ij = TPT(i, j);
          const int ib = TSZ(i);
#pragma omp task depend(in : ik[0], jk[0]) depend(inout : ij[0]) priority(prio)
          base_gemm_nt(ib, jb, kb, ik, (int) lda, jk, (int) lda, ij, (int) lda, 0);
        }
      }
    }
  }
#undef TSZ
#undef TPT
  return result;
}

int orc_omp_max_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
