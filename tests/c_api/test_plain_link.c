/* test_plain_link.c -- the call sequence of the reference's test/unit/c_api/factorization/test_cholesky_c_api_wrapper.c
 * (dlaf_initialize, dlaf_create_grid(MPI_Comm ...), dlaf_pdpotrf, dlaf_free_grid, dlaf_finalize) compiled WITHOUT any
 * macro of this repository and linked against -ldlaf_mi355x alone (+ the MPI the caller uses anyway): the headers
 * declare the MPI-typed entries because <mpi.h> is on the include path, the core library forwards them to
 * libdlaf_mi355x_mpi.so.
 *   gcc test_plain_link.c -I<repo>/include -I<mpi>/include -L<repo>/dla_future_amd/lib -ldlaf_mi355x -lmpi -lm
 *   mpiexec -n 4 ./test_plain_link 2 2 C      (argument "cpu": grid entry points only, no GPU needed)
 */
#include <math.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlaf_c/factorization/cholesky.h>
#include <dlaf_c/grid.h>
#include <dlaf_c/init.h>

static int numroc(int n, int nb, int iproc, int isrc, int nprocs) {
  int mydist = (nprocs + iproc - isrc) % nprocs, nblocks = n / nb, r = (nblocks / nprocs) * nb, extra = nblocks % nprocs;
  if (mydist < extra)
    r += nb;
  else if (mydist == extra)
    r += n % nb;
  return r;
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  const int nprow = argc > 1 ? atoi(argv[1]) : 1, npcol = argc > 2 ? atoi(argv[2]) : 1;
  const char order = argc > 3 ? argv[3][0] : 'R';
  const int cpu_only = argc > 4 && strcmp(argv[4], "cpu") == 0;
  if (nprow * npcol != size) {
    fprintf(stderr, "grid %d x %d needs %d ranks\n", nprow, npcol, nprow * npcol);
    MPI_Abort(MPI_COMM_WORLD, 2);
  }
  const char* pika_argv[] = {"plain", NULL};
  const char* dlaf_argv[] = {"plain", NULL};
  if (!cpu_only)
    dlaf_initialize(1, pika_argv, 1, dlaf_argv);
  const int ctx = dlaf_create_grid(MPI_COMM_WORLD, nprow, npcol, order);
  const int myrow = order == 'C' ? rank % nprow : rank / npcol, mycol = order == 'C' ? rank / nprow : rank % npcol;
  int ok = grid_ordering(MPI_COMM_WORLD, nprow, npcol, myrow, mycol) == order || nprow == 1 || npcol == 1;
  if (!cpu_only) {
    /* the analytic known-answer matrix of util_generic_lapack.h:39-68 through dlaf_pdpotrf */
    const int n = 50, nb = 8;
    const int mloc = numroc(n, nb, myrow, 0, nprow), nloc = numroc(n, nb, mycol, 0, npcol);
    const int lld = mloc > 0 ? mloc : 1;
    double* a = malloc(sizeof(double) * (size_t) lld * (nloc > 0 ? nloc : 1));
    for (int jl = 0; jl < nloc; ++jl)
      for (int il = 0; il < mloc; ++il) {
        const int i = (il / nb * nprow + myrow) * nb + il % nb, j = (jl / nb * npcol + mycol) * nb + jl % nb;
        const double mn = i < j ? i : j;
        a[il + (size_t) jl * lld] = i < j ? -9.9 : exp2(-(double) (i + j)) / 3 * (exp2(2 * (mn + 1)) - 1);
      }
    int desca[9] = {1, ctx, n, n, nb, nb, 0, 0, lld}, info = -1;
    dlaf_pdpotrf('L', n, a, 1, 1, desca, &info);
    ok = ok && info == 0;
    for (int jl = 0; jl < nloc; ++jl)
      for (int il = 0; il < mloc; ++il) {
        const int i = (il / nb * nprow + myrow) * nb + il % nb, j = (jl / nb * npcol + mycol) * nb + jl % nb;
        const double expect = i < j ? -9.9 : exp2(-fabs((double) (i - j)));
        if (fabs(a[il + (size_t) jl * lld] - expect) > 4 * (n + 1) * 2 * 2.220446049250313e-16 * (fabs(expect) > 1 ? fabs(expect) : 1))
          ok = 0;
      }
    free(a);
  }
  dlaf_free_grid(ctx);
  if (!cpu_only)
    dlaf_finalize();
  int all = 0;
  MPI_Allreduce(&ok, &all, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
  if (rank == 0)
    printf(all ? "PLAIN_LINK_TEST OK\n" : "PLAIN_LINK_TEST FAILED\n");
  MPI_Finalize();
  return all ? 0 : 1;
}
