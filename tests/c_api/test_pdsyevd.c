/* test_pdsyevd.c -- a plain C + MPI caller of the reference's eigensolver C interface, the way
 * test/unit/c_api/eigensolver/test_eigensolver_c_api.cpp drives it through C wrappers
 * (test_eigensolver_c_api_wrapper.c:17-60, test_gen_eigensolver_c_api_wrapper.c): the DLAF_descriptor entries
 * dlaf_symmetric_eigensolver_d / dlaf_hermitian_eigensolver_z, the ScaLAPACK-style entries dlaf_pdsyevd / dlaf_pzheevd
 * and the generalized ones dlaf_pdsygvd / dlaf_pzhegvd, on the reference's own size list
 * (test/unit/eigensolver/test_eigensolver.cpp:64-76 with its eigensolver_min_band), block-cyclic local arrays built here
 * in C with a non-zero source rank, results checked with testEigensolverCorrectness
 * (test/include/dlaf_test/eigensolver/test_eigensolver_correctness.h:37-101: sorted, Z^H Z == I within 10 m error,
 * A Z == Z Lambda within 2 m error; generalized: Z^H B Z == I, A Z == B Z Lambda).
 *
 *   gcc -std=c11 test_pdsyevd.c -I<repo>/include -I<mpi>/include -DDLAF_MI355X_WITH_MPI -L<repo>/dla_future_amd/lib \
 *       -ldlaf_mi355x_mpi -ldlaf_mi355x -lmpi -lm -o test_pdsyevd
 *   mpiexec -n 4 ./test_pdsyevd 2 2 C
 */
#include <complex.h>
#include <float.h>
#include <math.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlaf_c/eigensolver/eigensolver.h>
#include <dlaf_c/eigensolver/gen_eigensolver.h>
#include <dlaf_c/grid.h>
#include <dlaf_c/init.h>
#include <dlaf_c/utils.h>
#include <dlaf_mi355x/dlaf_mi355x.h>

typedef double complex zc;

static int numroc(int n, int nb, int iproc, int isrc, int nprocs) {
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
static int l2g(int l, int nb, int iproc, int isrc, int nprocs) {
  return ((l / nb) * nprocs + (nprocs + iproc - isrc) % nprocs) * nb + l % nb;
}

/* a fixed Hermitian matrix with a spread spectrum, and a Hermitian positive definite one */
static zc el_a(int i, int j, int cx) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const double r = cos(0.7 * lo + 1.3 * hi) + (i == j ? 0.25 * i : 0.0);
  if (!cx || i == j)
    return r;
  const double im = sin(0.3 * lo - 0.9 * hi);
  return i > j ? r + I * im : r - I * im;
}
static zc el_b(int i, int j, int cx, int n) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  if (i == j)
    return 2.0 + n;
  const double r = 0.5 * sin(1.1 * lo + 0.4 * hi);
  if (!cx)
    return r;
  const double im = 0.5 * cos(0.2 * lo + 0.6 * hi);
  return i > j ? r + I * im : r - I * im;
}

/* kind 0: DLAF_descriptor entry, 1: ScaLAPACK-style entry, 2: generalized ScaLAPACK-style entry */
static int run(int ctx, int nprow, int npcol, int myrow, int mycol, int n, int nb, int cx, int kind) {
  const int isrc = nprow - 1, jsrc = npcol > 1 ? 1 : 0;
  const int mloc = numroc(n, nb, myrow, isrc, nprow), nloc = numroc(n, nb, mycol, jsrc, npcol);
  const int lld = (mloc > 0 ? mloc : 1) + 1;
  const size_t elems = (size_t) lld * (nloc > 0 ? nloc : 1);
  const size_t esz = cx ? sizeof(zc) : sizeof(double);
  void *a = calloc(elems, esz), *b = calloc(elems, esz), *z = calloc(elems, esz);
  double* w = calloc((size_t) (n > 0 ? n : 1), sizeof(double));
  for (int jl = 0; jl < nloc; ++jl) {
    const int gj = l2g(jl, nb, mycol, jsrc, npcol);
    for (int il = 0; il < mloc; ++il) {
      const int gi = l2g(il, nb, myrow, isrc, nprow);
      /* only the lower triangle is referenced (eigensolver.h:39-58): the upper one gets a sentinel */
      const zc va = gi >= gj ? el_a(gi, gj, cx) : -9.9, vb = gi >= gj ? el_b(gi, gj, cx, n) : -9.9;
      if (cx) {
        ((zc*) a)[il + (size_t) jl * lld] = va;
        ((zc*) b)[il + (size_t) jl * lld] = vb;
      }
      else {
        ((double*) a)[il + (size_t) jl * lld] = creal(va);
        ((double*) b)[il + (size_t) jl * lld] = creal(vb);
      }
    }
  }
  int desc[9] = {1, ctx, n, n, nb, nb, isrc, jsrc, lld};
  int info = 0;
  if (kind == 0) {
    const struct DLAF_descriptor d = make_dlaf_descriptor(n, n, 1, 1, desc);
    info = cx ? dlaf_hermitian_eigensolver_z(ctx, 'L', (dlaf_complex_z*) a, d, w, (dlaf_complex_z*) z, d)
              : dlaf_symmetric_eigensolver_d(ctx, 'L', (double*) a, d, w, (double*) z, d);
  }
  else if (kind == 1) {
    info = -1;
    if (cx)
      dlaf_pzheevd('L', n, (dlaf_complex_z*) a, 1, 1, desc, w, (dlaf_complex_z*) z, 1, 1, desc, &info);
    else
      dlaf_pdsyevd('L', n, (double*) a, 1, 1, desc, w, (double*) z, 1, 1, desc, &info);
  }
  else {
    info = -1;
    if (cx)
      dlaf_pzhegvd('L', n, (dlaf_complex_z*) a, 1, 1, desc, (dlaf_complex_z*) b, 1, 1, desc, w, (dlaf_complex_z*) z, 1, 1,
                   desc, &info);
    else
      dlaf_pdsygvd('L', n, (double*) a, 1, 1, desc, (double*) b, 1, 1, desc, w, (double*) z, 1, 1, desc, &info);
  }
  int bad = info != 0;
  if (bad)
    fprintf(stderr, "rank (%d,%d) kind %d %c n=%d nb=%d: info %d\n", myrow, mycol, kind, cx ? 'z' : 'd', n, nb, info);
  if (n > 0 && !bad) {
    /* every rank assembles the whole eigenvector matrix (n <= 34) and checks it */
    zc* zg = calloc((size_t) n * n, sizeof(zc));
    for (int jl = 0; jl < nloc; ++jl)
      for (int il = 0; il < mloc; ++il)
        zg[l2g(il, nb, myrow, isrc, nprow) + (size_t) l2g(jl, nb, mycol, jsrc, npcol) * n] =
            cx ? ((zc*) z)[il + (size_t) jl * lld] : ((double*) z)[il + (size_t) jl * lld];
    MPI_Allreduce(MPI_IN_PLACE, zg, 2 * n * n, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    const double err = (cx ? 8 : 2) * DBL_EPSILON; /* TypeUtilities<T>::error, util_types.h:40,62 */
    double wmax = 0, amax = 0;
    for (int i = 0; i < n; ++i) {
      wmax = fmax(wmax, fabs(w[i]));
      if (i > 0 && w[i] < w[i - 1])
        bad = 1;
    }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        amax = fmax(amax, cabs(el_a(i, j, cx)));
    /* M Z with M = B (generalized) or I */
    zc* mz = calloc((size_t) n * n, sizeof(zc));
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        zc sum = 0;
        if (kind == 2)
          for (int k = 0; k < n; ++k)
            sum += el_b(i, k, cx, n) * zg[k + (size_t) j * n];
        else
          sum = zg[i + (size_t) j * n];
        mz[i + (size_t) j * n] = sum;
      }
    double orth = 0, res = 0;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        zc g = 0, az = 0;
        for (int k = 0; k < n; ++k) {
          g += conj(zg[k + (size_t) i * n]) * mz[k + (size_t) j * n];
          az += el_a(i, k, cx) * zg[k + (size_t) j * n];
        }
        orth = fmax(orth, cabs(g - (i == j ? 1.0 : 0.0)));
        res = fmax(res, cabs(az - w[j] * mz[i + (size_t) j * n]));
      }
    const double bscale = kind == 2 ? 2.0 + n : 1.0;
    const double orth_bar = 10.0 * n * err * bscale, res_bar = (kind == 2 ? 10.0 : 2.0) * n * err * fmax(1.0, amax * fmax(1.0, wmax)) * bscale;
    if (!(orth <= orth_bar) || !(res <= res_bar)) {
      if (myrow == 0 && mycol == 0)
        fprintf(stderr, "kind %d %c n=%d nb=%d: |Z^H M Z - I| = %g (bar %g), |A Z - M Z L| = %g (bar %g)\n", kind,
                cx ? 'z' : 'd', n, nb, orth, orth_bar, res, res_bar);
      bad = 1;
    }
    free(mz);
    free(zg);
  }
  free(a);
  free(b);
  free(z);
  free(w);
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
      fprintf(stderr, "usage: mpiexec -n P test_pdsyevd nprow npcol [R|C] with nprow*npcol == P\n");
    MPI_Finalize();
    return 2;
  }
  const char* pika_argv[] = {"dlaf"};
  const char* dlaf_argv[] = {"dlaf"};
  dlaf_initialize(1, pika_argv, 1, dlaf_argv);
  int ctx = dlaf_create_grid(MPI_COMM_WORLD, nprow, npcol, order);
  int myrow = order == 'C' ? rank % nprow : rank / npcol;
  int mycol = order == 'C' ? rank / nprow : rank % npcol;
  /* {m, mb, eigensolver_min_band}: test_eigensolver.cpp:64-72 */
  const int sizes[][3] = {{0, 2, 100},  {5, 8, 100},  {34, 34, 100}, {4, 3, 100}, {16, 10, 100},
                          {34, 13, 100}, {32, 5, 100}, {34, 8, 3},    {32, 6, 3}};
  int bad = 0;
  /* a one-process grid runs the whole list, several ranks every second entry (the Python worker runs the list on
   * the six-rank grids as well) */
  for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); s += (size > 1 ? 2 : 1)) {
    dlaf_mi355x_set_eigensolver_min_band(sizes[s][2]);
    for (int cx = 0; cx < 2; ++cx)
      for (int kind = 0; kind < 3; ++kind) {
        if (rank == 0 && getenv("C_API_TEST_VERBOSE")) {
          fprintf(stderr, "case n=%d nb=%d b_min=%d %c kind %d\n", sizes[s][0], sizes[s][1], sizes[s][2], cx ? 'z' : 'd', kind);
          fflush(stderr);
        }
        bad |= run(ctx, nprow, npcol, myrow, mycol, sizes[s][0], sizes[s][1], cx, kind);
      }
  }
  dlaf_mi355x_set_eigensolver_min_band(100);
  int anybad = 0;
  MPI_Allreduce(&bad, &anybad, 1, MPI_INT, MPI_LOR, MPI_COMM_WORLD);
  dlaf_free_grid(ctx);
  dlaf_finalize();
  if (rank == 0)
    printf("C_API_EIG_TEST %s (%d x %d grid, order %c, %d ranks)\n", anybad ? "FAILED" : "OK", nprow, npcol, order, size);
  MPI_Finalize();
  return anybad;
}
