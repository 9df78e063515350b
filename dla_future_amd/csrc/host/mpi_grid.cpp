// mpi_grid.cpp -- the MPI-typed grid entry points of the reference's C API, built into the optional
// libdlaf_mi355x_mpi.so (only when mpi.h / libmpi exist at build time; the core library stays MPI-free).
//
//   dlaf_create_grid(MPI_Comm, nprow, npcol, order)     include/dlaf_c/grid.h:31, src/c_api/grid.cpp:28-39
//   grid_ordering(MPI_Comm, nprow, npcol, myprow, mypcol)  include/dlaf_c/grid.h:54, src/c_api/grid.cpp:45-68
//   dlaf_create_grid_from_blacs(blacs_ctxt)              include/dlaf_c/grid.h:71, src/c_api/grid.cpp:73-92
//
// MPI is used for what the reference uses it for at this point -- the barrier and the process
// layout -- plus the bootstrap of the device transport:
//   DLAF_MI355X_MPI_TRANSPORT=rccl (default): rank 0's RCCL unique id is MPI_Bcast to the grid, the
//       factorization then talks RCCL over xGMI only;
//   DLAF_MI355X_MPI_TRANSPORT=host: panel broadcasts are staged through host memory and sent with MPI_Bcast
//       on row / column communicators from MPI_Comm_split -- the reference's non-GPU-aware-MPI path
//       (communication/kernels/internal/broadcast.h:62-70); lets several ranks share one GPU.
#define DLAF_MI355X_WITH_MPI 1
#include <mpi.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include <dlaf_c/grid.h>
#include <dlaf_mi355x/dlaf_mi355x.h>

namespace {
struct MpiComms {
  MPI_Comm world = MPI_COMM_NULL, row = MPI_COMM_NULL, col = MPI_COMM_NULL;
};
std::map<int, std::unique_ptr<MpiComms>> g_comms;  // kept alive as long as the grid
// Set when this library's statics are being torn down (constructed after g_comms, so destroyed before it).  The core
// library -- a dependency of this one, hence destroyed LATER -- frees the grids a program never freed from its own
// static destructors and calls release_comms for each: by then g_comms and the MpiComms it owned are gone.
bool g_shutdown = false;
struct ShutdownFlag {
  ~ShutdownFlag() { g_shutdown = true; }
} g_shutdown_flag;

int host_bcast(void* user, int axis, int root, void* buf, size_t bytes) {
  auto* c = static_cast<MpiComms*>(user);
  MPI_Comm comm = axis == 0 ? c->row : c->col;
  size_t off = 0;
  while (off < bytes) {  // MPI counts are ints
    const int chunk = (int) std::min<size_t>(bytes - off, (size_t) 1 << 30);
    if (MPI_Bcast(static_cast<char*>(buf) + off, chunk, MPI_BYTE, root, comm) != MPI_SUCCESS)
      return 1;
    off += (size_t) chunk;
  }
  return 0;
}

int host_barrier(void* user) {
  return MPI_Barrier(static_cast<MpiComms*>(user)->world) == MPI_SUCCESS ? 0 : 1;
}

// dlaf_free_grid / dlaf_finalize of the core library end here: drop this grid's communicators
void release_comms(void* user) {
  if (g_shutdown)
    return;  // process exit: the communicators went with g_comms (or with MPI_Finalize)
  auto* c = static_cast<MpiComms*>(user);
  int finalized = 0;
  MPI_Finalized(&finalized);
  for (auto it = g_comms.begin(); it != g_comms.end(); ++it)
    if (it->second.get() == c) {
      if (!finalized)
        for (MPI_Comm* m : {&c->world, &c->row, &c->col})
          if (*m != MPI_COMM_NULL)
            MPI_Comm_free(m);
      g_comms.erase(it);
      return;
    }
}
}  // namespace

extern "C" int dlaf_create_grid(MPI_Comm comm, int nprow, int npcol, char order) noexcept {
  int rank = 0, size = 1;
  MPI_Barrier(comm);  // reference: grid.cpp:35
  MPI_Comm_rank(comm, &rank);
  MPI_Comm_size(comm, &size);
  if (nprow * npcol != size) {
    std::fprintf(stderr, "[dlaf_mi355x] dlaf_create_grid: %d x %d grid on a communicator of %d ranks\n", nprow, npcol,
                 size);
    std::abort();
  }
  if (size == 1)
    return dlaf_mi355x_create_grid_single();
  const char* tr = std::getenv("DLAF_MI355X_MPI_TRANSPORT");
  if (tr && std::strcmp(tr, "host") == 0) {
    auto c = std::make_unique<MpiComms>();
    const bool colmajor = (order == 'C' || order == 'c');
    const int myrow = colmajor ? rank % nprow : rank / npcol;
    const int mycol = colmajor ? rank / nprow : rank % npcol;
    MPI_Comm_dup(comm, &c->world);
    // row communicator ranked by process column, column communicator ranked by process row
    // (src/communication/communicator_grid.cpp:41-43)
    MPI_Comm_split(comm, myrow, mycol, &c->row);
    MPI_Comm_split(comm, mycol, myrow, &c->col);
    const int ctx = dlaf_mi355x_create_grid_host(size, rank, nprow, npcol, order, host_bcast, host_barrier, c.get());
    dlaf_mi355x_grid_on_free(ctx, release_comms, c.get());
    g_comms.emplace(ctx, std::move(c));  // contexts are unique among live grids: never replaces an entry
    return ctx;
  }
  char uid[DLAF_MI355X_UNIQUE_ID_BYTES];
  if (rank == 0)
    dlaf_mi355x_rccl_unique_id(uid);
  MPI_Bcast(uid, DLAF_MI355X_UNIQUE_ID_BYTES, MPI_BYTE, 0, comm);
  return dlaf_mi355x_create_grid_rccl(uid, size, rank, nprow, npcol, order);
}

extern "C" char grid_ordering(MPI_Comm comm, int nprow, int npcol, int myprow, int mypcol) noexcept {
  int rank = 0;
  MPI_Comm_rank(comm, &rank);
  int mine[2] = {rank == myprow * npcol + mypcol ? 1 : 0, rank == mypcol * nprow + myprow ? 1 : 0};
  int all[2] = {0, 0};
  MPI_Allreduce(mine, all, 2, MPI_INT, MPI_LAND, comm);
  if (!all[0] && !all[1]) {
    std::fprintf(stderr, "Grid layout must be row major or column major.\n");
    std::exit(-1);
  }
  return all[1] ? 'C' : 'R';
}

// reference: src/c_api/grid.cpp:73-92 (compiled there under DLAF_WITH_SCALAPACK against src/c_api/blacs.h).  The
// three BLACS functions are resolved in the running program: nothing here links a ScaLAPACK.
extern "C" void dlaf_create_grid_from_blacs(int blacs_ctxt) noexcept {
  using get_fn = void (*)(int, int, int*);
  using handle_fn = MPI_Comm (*)(int);
  using info_fn = void (*)(int, int*, int*, int*, int*);
  auto blacs_get = reinterpret_cast<get_fn>(dlsym(RTLD_DEFAULT, "Cblacs_get"));
  auto blacs2sys = reinterpret_cast<handle_fn>(dlsym(RTLD_DEFAULT, "Cblacs2sys_handle"));
  auto blacs_info = reinterpret_cast<info_fn>(dlsym(RTLD_DEFAULT, "Cblacs_gridinfo"));
  if (!blacs_get || !blacs2sys || !blacs_info) {
    std::fprintf(stderr, "[dlaf_mi355x] dlaf_create_grid_from_blacs: no BLACS (Cblacs_get / Cblacs2sys_handle / "
                         "Cblacs_gridinfo) in this program\n");
    std::abort();
  }
  int probe[4];
  if (dlaf_mi355x_grid_info(blacs_ctxt, &probe[0], &probe[1], &probe[2], &probe[3]) == 0)
    return;  // already registered: try_emplace semantics of the reference
  int system_ctxt = 0;
  blacs_get(blacs_ctxt, 10, &system_ctxt);  // SGET_BLACSCONTXT == 10
  MPI_Comm communicator = blacs2sys(system_ctxt);
  int dims[2] = {0, 0}, coords[2] = {-1, -1};
  blacs_info(blacs_ctxt, &dims[0], &dims[1], &coords[0], &coords[1]);
  const char order = grid_ordering(communicator, dims[0], dims[1], coords[0], coords[1]);
  const int ctx = dlaf_create_grid(communicator, dims[0], dims[1], order);
  if (dlaf_mi355x_grid_rekey(ctx, blacs_ctxt) != 0) {
    std::fprintf(stderr, "[dlaf_mi355x] dlaf_create_grid_from_blacs: context %d cannot be registered\n", blacs_ctxt);
    std::abort();
  }
  auto it = g_comms.find(ctx);  // (host transport: the communicators follow the grid to its new number)
  if (it != g_comms.end()) {
    std::unique_ptr<MpiComms> c = std::move(it->second);
    g_comms.erase(it);
    g_comms.emplace(blacs_ctxt, std::move(c));
  }
}
