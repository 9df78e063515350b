/* test_blacs_grid.c -- dlaf_create_grid_from_blacs (include/dlaf_c/grid.h; reference grid.h:71) the way a
 * ScaLAPACK application uses it: the BLACS grid exists first, DLA-Future's grid is registered under the BLACS
 * context, dlaf_pdpotrf finds it through desca[1].  Mirrors the reference's C-API tests with a BLACS context
 * (test/unit/c_api/factorization/test_cholesky_c_api.cpp:62-155 under DLAF_WITH_SCALAPACK).
 *   mpiexec -n 4 ./test_blacs_grid 2 2 R [factorize]
 * Without "factorize" nothing touches a GPU (grid wiring only). */
#include <math.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlaf_c/factorization/cholesky.h>
#include <dlaf_c/grid.h>
#include <dlaf_c/init.h>
#include <dlaf_mi355x/dlaf_mi355x.h>

void Cblacs_get(int, int, int*);
void Cblacs_gridinit(int*, const char*, int, int);
void Cblacs_gridinfo(int, int*, int*, int*, int*);
void Cblacs_gridexit(int);

static int numroc(int n, int nb, int iproc, int isrc, int nprocs) {
  const int mydist = (nprocs + iproc - isrc) % nprocs, nblocks = n / nb;
  int r = (nblocks / nprocs) * nb;
  const int extra = nblocks % nprocs;
  if (mydist < extra)
    r += nb;
  else if (mydist == extra)
    r += n % nb;
  return r;
}

int main(int argc, char** argv) {
  int provided = 0;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_MULTIPLE, &provided);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  const int nprow = atoi(argv[1]), npcol = atoi(argv[2]);
  const char order[2] = {argv[3][0], 0};
  const int factorize = argc > 4 && strcmp(argv[4], "factorize") == 0;
  int bad = 0;

  int ictxt = 0;
  Cblacs_get(0, 0, &ictxt);
  Cblacs_gridinit(&ictxt, order, nprow, npcol);
  int np, nq, myrow, mycol;
  Cblacs_gridinfo(ictxt, &np, &nq, &myrow, &mycol);

  if (factorize)
    dlaf_initialize(0, NULL, 0, NULL);
  dlaf_create_grid_from_blacs(ictxt);
  dlaf_create_grid_from_blacs(ictxt); /* a second call is a no-op, as upstream's try_emplace */
  int pr, pc, r, c;
  bad |= dlaf_mi355x_grid_info(ictxt, &pr, &pc, &r, &c) != 0 || pr != nprow || pc != npcol || r != myrow || c != mycol;

  if (factorize) {
    /* A = tridiagonal-ish SPD matrix with a closed-form check: A = M + n I, M(i,j) = 1 / (1 + |i - j|) */
    const int n = 700, nb = 64;
    const int lr = numroc(n, nb, myrow, 0, nprow), lc = numroc(n, nb, mycol, 0, npcol), lld = lr > 0 ? lr : 1;
    double* a = (double*) malloc(sizeof(double) * (size_t) lld * (size_t) (lc > 0 ? lc : 1));
    double* a0 = (double*) malloc(sizeof(double) * (size_t) lld * (size_t) (lc > 0 ? lc : 1));
    for (int jl = 0; jl < lc; ++jl)
      for (int il = 0; il < lr; ++il) {
        const int gi = ((il / nb) * nprow + myrow) * nb + il % nb, gj = ((jl / nb) * npcol + mycol) * nb + jl % nb;
        a[il + (size_t) jl * lld] = a0[il + (size_t) jl * lld] = 1.0 / (1.0 + abs(gi - gj)) + (gi == gj ? n : 0.0);
      }
    int desca[9] = {1, ictxt, n, n, nb, nb, 0, 0, lld}, info = -1;
    dlaf_pdpotrf('L', n, a, 1, 1, desca, &info);
    bad |= info != 0;
    /* local check that needs no communication: the first block column of L satisfies L(:,0:nb) L(0:nb,0:nb)^T =
     * A(:,0:nb); the owner column of global block column 0 verifies its rows against its copy of L00 (rank row 0
     * holds L00; the others only check that the strictly upper part and the diagonal sign are sane) */
    if (mycol == 0 && myrow == 0 && lr > 0) {
      for (int j = 0; j < nb && j < n; ++j) {
        double s = 0;
        for (int k = 0; k <= j; ++k)
          s += a[j + (size_t) k * lld] * a[j + (size_t) k * lld];
        bad |= fabs(s - a0[j + (size_t) j * lld]) > 1e-9 * n;
        bad |= !(a[j + (size_t) j * lld] > 0);
        for (int i = 0; i < j; ++i) /* the upper triangle of the diagonal tile is untouched */
          bad |= a[i + (size_t) j * lld] != a0[i + (size_t) j * lld];
      }
    }
    free(a);
    free(a0);
  }

  int any = 0;
  MPI_Allreduce(&bad, &any, 1, MPI_INT, MPI_LOR, MPI_COMM_WORLD);
  dlaf_free_grid(ictxt);
  bad = dlaf_mi355x_grid_info(ictxt, &pr, &pc, &r, &c) == 0; /* gone */
  MPI_Allreduce(MPI_IN_PLACE, &any, 1, MPI_INT, MPI_LOR, MPI_COMM_WORLD);
  any |= bad;
  if (factorize)
    dlaf_finalize();
  Cblacs_gridexit(ictxt);
  if (rank == 0)
    printf("BLACS_GRID_TEST %s\n", any ? "FAILED" : "OK");
  MPI_Finalize();
  return any;
}
