/* dlaf_c/grid.h -- process grids.
 * Mirrors the reference's include/dlaf_c/grid.h:31-71 (src/c_api/grid.cpp:26-92).  Contexts are
 * handed out downwards from INT_MAX exactly like upstream (grid.cpp:31) so they cannot collide
 * with BLACS contexts.
 *
 * The MPI-typed entry points are declared, as upstream declares them unconditionally (grid.h:31,54,71),
 * whenever <mpi.h> can be found -- no macro needed; DLAF_MI355X_WITH_MPI forces them on, DLAF_MI355X_NO_MPI off.
 * They are implemented by libdlaf_mi355x_mpi.so (built when an MPI is installed); a program that links only
 * -ldlaf_mi355x reaches them too: the core library exports the same three symbols as forwarders that load the
 * shim from the library's own directory on first use (csrc/host/c_api.cpp).  The MI355X build talks RCCL over
 * xGMI, so a grid can also be made without MPI through include/dlaf_mi355x/dlaf_mi355x.h
 * (dlaf_mi355x_create_grid_rccl). */
#pragma once
#include <dlaf_c/utils.h>

#if !defined(DLAF_MI355X_WITH_MPI) && !defined(DLAF_MI355X_NO_MPI) && defined(__has_include)
#if __has_include(<mpi.h>)
#define DLAF_MI355X_WITH_MPI 1
#endif
#endif

#ifdef DLAF_MI355X_WITH_MPI
#include <mpi.h>
/* reference: grid.h:31 -- order 'R' (row-major) or 'C' (column-major) rank placement */
DLAF_EXTERN_C int dlaf_create_grid(MPI_Comm comm, int nprow, int npcol, char order) DLAF_NOEXCEPT;
/* reference: grid.h:54 */
DLAF_EXTERN_C char grid_ordering(MPI_Comm comm, int nprow, int npcol, int myprow, int mypcol) DLAF_NOEXCEPT;
/* reference: grid.h:71 (there under DLAF_WITH_SCALAPACK) -- the grid of an existing BLACS context, registered
 * under that context number, so that dlaf_p?potrf finds it through desca[1] like a ScaLAPACK routine would.  The
 * BLACS entry points (Cblacs_get, Cblacs2sys_handle, Cblacs_gridinfo) are taken from the calling program at run
 * time: a caller that has a BLACS context has a BLACS library loaded. */
DLAF_EXTERN_C void dlaf_create_grid_from_blacs(int blacs_ctxt) DLAF_NOEXCEPT;
#endif

/* reference: grid.h:39 */
DLAF_EXTERN_C void dlaf_free_grid(int context) DLAF_NOEXCEPT;
