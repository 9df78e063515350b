/* test_pdpotrf.c -- a plain C + MPI caller of the reference's C interface, the way
 * test/unit/c_api/factorization/test_cholesky_c_api.cpp drives dlaf_p?potrf through C wrappers
 * (test_cholesky_c_api_wrapper.c): dlaf_initialize, dlaf_create_grid(MPI_Comm...), dlaf_pdpotrf and
 * dlaf_pzpotrf on the analytic known-answer matrix (util_generic_lapack.h:39-68), block-cyclic local
 * arrays built here in C, result checked with the reference's tolerance (test_cholesky.cpp:76-77).
 *
 *   mpicc test_pdpotrf.c -I<repo>/include -DDLAF_MI355X_WITH_MPI -L<repo>/dla_future_amd/lib \
 *         -ldlaf_mi355x_mpi -ldlaf_mi355x -lm -o test_pdpotrf
 *   mpiexec -n 4 ./test_pdpotrf 2 2 R
 */
#include <complex.h>
#include <float.h>
#include <math.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>

#include <dlaf_c/factorization/cholesky.h>
#include <dlaf_c/grid.h>
#include <dlaf_c/init.h>
#include <dlaf_mi355x/dlaf_mi355x.h>

static int numroc(int n, int nb, int iproc, int isrc, int nprocs) {
  /* ScaLAPACK NUMROC */
  int mydist = (nprocs + iproc - isrc) % nprocs;
  int nblocks = n / nb;
  int r = (nblocks / nprocs) * nb;
  int extra = nblocks % nprocs;
  if (mydist < extra)
    r += nb;
  else if (mydist == extra)
    r += n % nb;
  return r;
}

static double complex el_a(char uplo, int i, int j, int cx) {
  if ((uplo == 'L' && i < j) || (uplo == 'U' && i > j))
    return -9.9;
  double mn = i < j ? i : j;
  double r = exp2(-(double) (i + j)) / 3 * (exp2(2 * (mn + 1)) - 1);
  return cx ? r * cexp(I * (double) (j - i)) : r;
}
static double complex el_l(char uplo, int i, int j, int cx) {
  if ((uplo == 'L' && i < j) || (uplo == 'U' && i > j))
    return -9.9;
  double r = exp2(-fabs((double) (i - j)));
  return cx ? r * cexp(I * (double) (j - i)) : r;
}

static int run(int ctx, int nprow, int npcol, int myrow, int mycol, char uplo, int n, int nb, int cx) {
  const int isrc = nprow - 1, jsrc = npcol > 1 ? 1 : 0; /* non-zero source rank, test_cholesky.cpp:85 */
  const int mloc = numroc(n, nb, myrow, isrc, nprow), nloc = numroc(n, nb, mycol, jsrc, npcol);
  const int lld = (mloc > 0 ? mloc : 1) + 3;
  double complex* az = NULL;
  double* ad = NULL;
  size_t elems = (size_t) lld * (nloc > 0 ? nloc : 1);
  if (cx)
    az = malloc(sizeof(double complex) * elems);
  else
    ad = malloc(sizeof(double) * elems);
  /* local (il, jl) -> global: block-cyclic with source rank */
  for (int jl = 0; jl < nloc; ++jl) {
    int gj = ((jl / nb) * npcol + (npcol + mycol - jsrc) % npcol) * nb + jl % nb;
    for (int il = 0; il < mloc; ++il) {
      int gi = ((il / nb) * nprow + (nprow + myrow - isrc) % nprow) * nb + il % nb;
      double complex v = el_a(uplo, gi, gj, cx);
      if (cx)
        az[il + (size_t) jl * lld] = v;
      else
        ad[il + (size_t) jl * lld] = creal(v);
    }
  }
  int desca[9] = {1, ctx, n, n, nb, nb, isrc, jsrc, lld};
  int info = -1;
  if (cx)
    dlaf_pzpotrf(uplo, n, (dlaf_complex_z*) az, 1, 1, desca, &info);
  else
    dlaf_pdpotrf(uplo, n, ad, 1, 1, desca, &info);
  int bad = info != 0;
  /* the next row of the path (p?trsm, dlaf_mi355x.h): with the factor, solve A X = A for X = I */
  if (!bad && n > 0) {
    double complex* bz = cx ? malloc(sizeof(double complex) * elems) : NULL;
    double* bd = cx ? NULL : malloc(sizeof(double) * elems);
    for (int jl = 0; jl < nloc; ++jl) {
      int gj = ((jl / nb) * npcol + (npcol + mycol - jsrc) % npcol) * nb + jl % nb;
      for (int il = 0; il < mloc; ++il) {
        int gi = ((il / nb) * nprow + (nprow + myrow - isrc) % nprow) * nb + il % nb;
        /* full Hermitian A from its stored triangle */
        int in_tri = (uplo == 'L') ? gi >= gj : gi <= gj;
        double complex v = in_tri ? el_a(uplo, gi, gj, cx) : conj(el_a(uplo, gj, gi, cx));
        if (cx)
          bz[il + (size_t) jl * lld] = v;
        else
          bd[il + (size_t) jl * lld] = creal(v);
      }
    }
    const double one_d = 1.0;
    const double complex one_z = 1.0;
    /* uplo L: L Y = B, L^H X = Y;   uplo U: U^H Y = B, U X = Y */
    const char op1 = uplo == 'L' ? 'N' : 'C', op2 = uplo == 'L' ? 'C' : 'N';
    if (cx) {
      dlaf_mi355x_pztrsm('L', uplo, op1, 'N', n, n, (const dlaf_complex_z*) &one_z, (const dlaf_complex_z*) az, 1, 1,
                         desca, (dlaf_complex_z*) bz, 1, 1, desca);
      dlaf_mi355x_pztrsm('L', uplo, op2, 'N', n, n, (const dlaf_complex_z*) &one_z, (const dlaf_complex_z*) az, 1, 1,
                         desca, (dlaf_complex_z*) bz, 1, 1, desca);
    }
    else {
      dlaf_mi355x_pdtrsm('L', uplo, op1, 'N', n, n, &one_d, ad, 1, 1, desca, bd, 1, 1, desca);
      dlaf_mi355x_pdtrsm('L', uplo, op2, 'N', n, n, &one_d, ad, 1, 1, desca, bd, 1, 1, desca);
    }
    const double stol = 100.0 * (n + 1) * (cx ? 8 : 2) * DBL_EPSILON;
    for (int jl = 0; jl < nloc && !bad; ++jl) {
      int gj = ((jl / nb) * npcol + (npcol + mycol - jsrc) % npcol) * nb + jl % nb;
      for (int il = 0; il < mloc; ++il) {
        int gi = ((il / nb) * nprow + (nprow + myrow - isrc) % nprow) * nb + il % nb;
        double complex g = cx ? bz[il + (size_t) jl * lld] : bd[il + (size_t) jl * lld];
        if (cabs(g - (gi == gj ? 1.0 : 0.0)) > stol) {
          fprintf(stderr, "rank (%d,%d) %c%c n=%d nb=%d: solve (%d,%d) got %g%+gi\n", myrow, mycol, cx ? 'z' : 'd',
                  uplo, n, nb, gi, gj, creal(g), cimag(g));
          bad = 1;
          break;
        }
      }
    }
    free(bz);
    free(bd);
  }
  const double tol = 4.0 * (n + 1) * (cx ? 8 : 2) * DBL_EPSILON;
  for (int jl = 0; jl < nloc && !bad; ++jl) {
    int gj = ((jl / nb) * npcol + (npcol + mycol - jsrc) % npcol) * nb + jl % nb;
    for (int il = 0; il < mloc; ++il) {
      int gi = ((il / nb) * nprow + (nprow + myrow - isrc) % nprow) * nb + il % nb;
      double complex e = el_l(uplo, gi, gj, cx);
      double complex g = cx ? az[il + (size_t) jl * lld] : ad[il + (size_t) jl * lld];
      double diff = cabs(e - g), mx = fmax(cabs(e), cabs(g));
      if (!(diff < tol || diff / mx < tol)) {
        fprintf(stderr, "rank (%d,%d) %c%c n=%d nb=%d: (%d,%d) expected %g%+gi got %g%+gi\n", myrow, mycol,
                cx ? 'z' : 'd', uplo, n, nb, gi, gj, creal(e), cimag(e), creal(g), cimag(g));
        bad = 1;
        break;
      }
    }
  }
  free(az);
  free(ad);
  return bad;
}

int main(int argc, char** argv) {
  int provided;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_MULTIPLE, &provided);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  int nprow = argc > 1 ? atoi(argv[1]) : 1, npcol = argc > 2 ? atoi(argv[2]) : size;
  char order = argc > 3 ? argv[3][0] : 'R';
  if (nprow * npcol != size) {
    if (rank == 0)
      fprintf(stderr, "usage: mpiexec -n P test_pdpotrf nprow npcol [R|C] with nprow*npcol == P\n");
    MPI_Finalize();
    return 2;
  }
  const char* pika_argv[] = {"dlaf", "--pika:print-bind"};
  const char* dlaf_argv[] = {"dlaf"};
  dlaf_initialize(1, pika_argv, 1, dlaf_argv);
  dlaf_initialize(1, pika_argv, 1, dlaf_argv); /* idempotent like upstream */
  int ctx = dlaf_create_grid(MPI_COMM_WORLD, nprow, npcol, order);
  int myrow = order == 'C' ? rank % nprow : rank / npcol;
  int mycol = order == 'C' ? rank / nprow : rank % npcol;
  char ord = grid_ordering(MPI_COMM_WORLD, nprow, npcol, myrow, mycol);
  int bad = (size > 1 && ord != order && !(nprow == 1 || npcol == 1));
  /* sizes of test/unit/factorization/test_cholesky.cpp:54-58 */
  const int sizes[][2] = {{0, 2}, {5, 8}, {34, 34}, {4, 3}, {16, 10}, {34, 13}, {32, 5}, {150, 64}};
  for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); ++s)
    for (int cx = 0; cx < 2; ++cx)
      for (int u = 0; u < 2; ++u)
        bad |= run(ctx, nprow, npcol, myrow, mycol, u ? 'U' : 'L', sizes[s][0], sizes[s][1], cx);
  int anybad = 0;
  MPI_Allreduce(&bad, &anybad, 1, MPI_INT, MPI_LOR, MPI_COMM_WORLD);
  dlaf_free_grid(ctx);
  dlaf_finalize();
  dlaf_finalize();
  if (rank == 0)
    printf("C_API_TEST %s (%d x %d grid, order %c, %d ranks)\n", anybad ? "FAILED" : "OK", nprow, npcol, order, size);
  MPI_Finalize();
  return anybad;
}
