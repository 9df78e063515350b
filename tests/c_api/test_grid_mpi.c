/* test_grid_mpi.c -- CPU-only check of the MPI-typed grid entry points (no GPU work):
 * dlaf_create_grid with the host-staged transport, grid_ordering, and the row / column communicator
 * wiring through dlaf_mi355x_grid_host_bcast.   mpiexec -n 6 ./test_grid_mpi 3 2 R */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>

#include <dlaf_c/grid.h>
#include <dlaf_mi355x/dlaf_mi355x.h>

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  int nprow = atoi(argv[1]), npcol = atoi(argv[2]);
  char order = argv[3][0];
  int ctx = dlaf_create_grid(MPI_COMM_WORLD, nprow, npcol, order);
  int pr, pc, myrow, mycol, bad = 0;
  bad |= dlaf_mi355x_grid_info(ctx, &pr, &pc, &myrow, &mycol) != 0 || pr != nprow || pc != npcol;
  int er = order == 'C' ? rank % nprow : rank / npcol, ec = order == 'C' ? rank / nprow : rank % npcol;
  bad |= (myrow != er || mycol != ec);
  if (nprow > 1 && npcol > 1)
    bad |= grid_ordering(MPI_COMM_WORLD, nprow, npcol, myrow, mycol) != order;
  for (int axis = 0; axis < 2; ++axis) {
    int members = axis == 0 ? npcol : nprow, me = axis == 0 ? mycol : myrow, fixed = axis == 0 ? myrow : mycol;
    for (int root = 0; root < members; ++root) {
      long buf[5];
      for (int i = 0; i < 5; ++i)
        buf[i] = (me == root) ? 1000 * axis + 100 * root + fixed : -1;
      bad |= dlaf_mi355x_grid_host_bcast(ctx, axis, root, buf, sizeof buf) != 0;
      for (int i = 0; i < 5; ++i)
        bad |= buf[i] != 1000 * axis + 100 * root + fixed;
    }
  }
  int any = 0;
  MPI_Allreduce(&bad, &any, 1, MPI_INT, MPI_LOR, MPI_COMM_WORLD);
  dlaf_free_grid(ctx);
  if (rank == 0)
    printf("GRID_MPI_TEST %s\n", any ? "FAILED" : "OK");
  MPI_Finalize();
  return any;
}
