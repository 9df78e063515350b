// c_api.cpp -- the extern "C" surface: the reference's dlaf_c entry points for the Cholesky path
// (include/dlaf_c/{init,grid,utils}.h, include/dlaf_c/factorization/cholesky.h; implemented upstream
// in src/c_api/{init,grid,utils}.cpp and src/c_api/factorization/cholesky.{h,cpp}) plus the
// MI355X extensions declared in include/dlaf_mi355x/dlaf_mi355x.h.
// (the core never sees <mpi.h>: the MPI-typed prototypes of dlaf_c/grid.h stay out of this translation unit, which
// defines forwarders with a pointer-sized first parameter under the same names)
#define DLAF_MI355X_NO_MPI 1
#include <dlfcn.h>

#include <climits>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>

#include <dlaf_c/eigensolver/eigensolver.h>
#include <dlaf_c/eigensolver/gen_eigensolver.h>
#include <dlaf_c/factorization/cholesky.h>
#include <dlaf_c/grid.h>
#include <dlaf_c/init.h>
#include <dlaf_mi355x/dlaf_mi355x.h>

#include "eigensolver.hpp"
#include "red2band.hpp"
#include "runtime.hpp"
#include "tile_matrix.hpp"

using namespace dlaf_mi355x;

namespace dlaf_mi355x {
template <class T>
void set_random_hpd_local(T* a, long ld, long n, int nb, const Axis& rows, const Axis& cols, int nthreads);
}

namespace {
// grid registry: contexts count down from INT_MAX (reference: src/c_api/grid.cpp:26-31)
std::unordered_map<int, std::unique_ptr<Grid>> g_grids;

// A context is never handed out twice while its grid is alive: the reference numbers them
// INT_MAX - size() (grid.cpp:31) and relies on try_emplace never replacing a live entry; here the
// search simply continues downwards past contexts that are still registered (create A, create B,
// free A, create C must not give C the context of B).
int register_grid(std::unique_ptr<Grid> g) {
  int ctx = INT_MAX - (int) g_grids.size();
  while (g_grids.find(ctx) != g_grids.end())
    --ctx;
  g_grids.emplace(ctx, std::move(g));
  return ctx;
}

Grid& grid_from_context(int ctx) {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    // reference: src/c_api/utils.cpp:55-68 prints this and terminates
    fatal("[ERROR] No DLA-Future grid for context %d. Did you forget to call dlaf_create_grid()?\n", ctx);
  return *it->second;
}

bool coords_from_rank(int rank, int nprow, int npcol, char order, int& myrow, int& mycol) {
  if (order == 'C' || order == 'c') {  // reference: common/index2d.h:345-355
    myrow = rank % nprow;
    mycol = rank / nprow;
  }
  else {
    myrow = rank / npcol;
    mycol = rank % npcol;
  }
  return myrow < nprow && mycol < npcol;
}

std::unique_ptr<Grid> make_grid(int nranks, int rank, int nprow, int npcol, char order) {
  if (nprow < 1 || npcol < 1 || nprow * npcol != nranks || rank < 0 || rank >= nranks)
    return nullptr;
  auto g = std::make_unique<Grid>();
  g->nprow = nprow;
  g->npcol = npcol;
  g->rank = rank;
  g->nranks = nranks;
  g->order = (order == 'C' || order == 'c') ? 'C' : 'R';
  coords_from_rank(rank, nprow, npcol, g->order, g->myrow, g->mycol);
  return g;
}

template <class T>
struct DevType;
template <>
struct DevType<float> {
  using type = float;
};
template <>
struct DevType<double> {
  using type = double;
};
template <>
struct DevType<std::complex<float>> {
  using type = cfloat;
};
template <>
struct DevType<std::complex<double>> {
  using type = cdouble;
};

void check_cholesky_desc(const DLAF_descriptor& d) {
  // preconditions of dlaf::cholesky_factorization (include/dlaf/factorization/cholesky.h:39-79) and of
  // the C wrapper (src/c_api/factorization/cholesky.h:37-38); upstream asserts -> terminate
  if (d.i != 0 || d.j != 0)
    fatal("[dlaf_mi355x] sub-matrices are not supported: i = %d, j = %d must be 0\n", d.i, d.j);
  if (d.m != d.n)
    fatal("[dlaf_mi355x] Cholesky needs a square matrix: %d x %d\n", d.m, d.n);
  if (d.mb != d.nb || d.nb < 1)
    fatal("[dlaf_mi355x] Cholesky needs square blocks: %d x %d\n", d.mb, d.nb);
  if (d.m < 0)
    fatal("[dlaf_mi355x] negative matrix size %d\n", d.m);
}

template <class HT>
int cholesky_host(int ctx, char uplo, HT* a, const DLAF_descriptor& d) {
  using DT = typename DevType<HT>::type;
  check_cholesky_desc(d);
  if (!(uplo == 'L' || uplo == 'l' || uplo == 'U' || uplo == 'u'))
    fatal("[dlaf_mi355x] uplo must be 'L' or 'U', got '%c'\n", uplo);
  Grid& g = grid_from_context(ctx);
  if (d.isrc < 0 || d.isrc >= g.nprow || d.jsrc < 0 || d.jsrc >= g.npcol)
    fatal("[dlaf_mi355x] source rank (%d,%d) outside the %d x %d grid\n", d.isrc, d.jsrc, g.nprow, g.npcol);
  DeviceMatrix<DT> m;
  m.create(&g, uplo, d.m, d.nb, d.isrc, d.jsrc);
  m.upload(reinterpret_cast<const DT*>(a), d.ld);
  // (a matrix that is not positive definite comes back partially overwritten, as from LAPACK)
  return m.factorize_and_download(reinterpret_cast<DT*>(a), d.ld);
}

template <class HT>
void pxpotrf(char uplo, int n, HT* a, int ia, int ja, const int desca[9], int* info) {
  // reference: src/c_api/factorization/cholesky.h:65-75
  if (desca[0] != 1)
    fatal("[dlaf_mi355x] desca[0] (dtype) must be 1, got %d\n", desca[0]);
  if (ia != 1 || ja != 1)
    fatal("[dlaf_mi355x] ia = %d, ja = %d must be 1\n", ia, ja);
  const DLAF_descriptor d = make_dlaf_descriptor(n, n, ia, ja, desca);
  const int r = cholesky_host<HT>(desca[1], uplo, a, d);
  if (info)
    *info = r;
}

// dlaf::triangular_solver (include/dlaf/solver/triangular.h:41-177) through descriptors
template <class HT>
int triangular_solver_c(int ctx, char side, char uplo, char op, char diag, const HT* alpha, const HT* a,
                        const DLAF_descriptor& da, HT* b, const DLAF_descriptor& db) {
  using DT = typename DevType<HT>::type;
  auto is = [](char c, const char* set) { return c != 0 && std::strchr(set, c) != nullptr; };
  if (!is(side, "LlRr") || !is(uplo, "LlUu") || !is(op, "NnTtCc") || !is(diag, "NnUu"))
    fatal("[dlaf_mi355x] triangular solver: bad side/uplo/op/diag '%c' '%c' '%c' '%c'\n", side, uplo, op, diag);
  if (da.i != 0 || da.j != 0 || db.i != 0 || db.j != 0)
    fatal("[dlaf_mi355x] sub-matrices are not supported: offsets must be 0\n");
  // preconditions of triangular.h:43-48, :57 / :93-98: A square with square blocks, op(A) and B multipliable
  if (da.m != da.n || da.mb != da.nb || da.nb < 1)
    fatal("[dlaf_mi355x] triangular solver: A must be square with square blocks (%d x %d, %d x %d)\n", da.m, da.n,
          da.mb, da.nb);
  const bool left = (side == 'L' || side == 'l');
  if (db.m < 0 || db.n < 0 || da.m != (left ? db.m : db.n))
    fatal("[dlaf_mi355x] triangular solver: A is %d x %d, B is %d x %d (side %c)\n", da.m, da.n, db.m, db.n, side);
  // multipliable (util_matrix.h:123-143): B's block along the triangular dimension is A's; the other one is free
  // (the device tiles stay square: the free dimension is re-cut locally, tile_matrix.hpp create_rhs)
  if ((left ? db.mb : db.nb) != da.nb || db.mb < 1 || db.nb < 1)
    fatal("[dlaf_mi355x] triangular solver: B's blocks (%d x %d) do not match A's (%d x %d) for side %c\n", db.mb, db.nb,
          da.mb, da.nb, side);
  Grid& g = grid_from_context(ctx);
  for (const DLAF_descriptor* d : {&da, &db})
    if (d->isrc < 0 || d->isrc >= g.nprow || d->jsrc < 0 || d->jsrc >= g.npcol)
      fatal("[dlaf_mi355x] source rank (%d,%d) outside the %d x %d grid\n", d->isrc, d->jsrc, g.nprow, g.npcol);
  DT al;
  std::memcpy(&al, alpha, sizeof(DT));
  return triangular_solver_host<DT>(&g, side, uplo, op, diag, al, reinterpret_cast<const DT*>(a), da.ld, da.isrc,
                                    da.jsrc, reinterpret_cast<DT*>(b), db.ld, db.m, db.n, da.nb, db.isrc, db.jsrc, left ? db.nb : db.mb);
}

// ScaLAPACK p?trsm argument list
template <class HT>
void pxtrsm(char side, char uplo, char op, char diag, int m, int n, const HT* alpha, const HT* a, int ia, int ja,
            const int desca[9], HT* b, int ib, int jb, const int descb[9]) {
  if (desca[0] != 1 || descb[0] != 1)
    fatal("[dlaf_mi355x] desc[0] (dtype) must be 1\n");
  if (ia != 1 || ja != 1 || ib != 1 || jb != 1)
    fatal("[dlaf_mi355x] ia, ja, ib, jb must be 1\n");
  if (desca[1] != descb[1])
    fatal("[dlaf_mi355x] A and B live on different contexts (%d, %d)\n", desca[1], descb[1]);
  const int na = (side == 'L' || side == 'l') ? m : n;
  const DLAF_descriptor da = make_dlaf_descriptor(na, na, ia, ja, desca);
  const DLAF_descriptor db = make_dlaf_descriptor(m, n, ib, jb, descb);
  (void) triangular_solver_c<HT>(desca[1], side, uplo, op, diag, alpha, a, da, b, db);
}

// ScaLAPACK p?potrs: solve A X = B with the factor p?potrf left in a (uplo L: L L^H, uplo U: U^H U)
template <class HT>
void pxpotrs(char uplo, int n, int nrhs, const HT* a, int ia, int ja, const int desca[9], HT* b, int ib, int jb,
             const int descb[9], int* info) {
  const HT one(1);
  const bool lower = (uplo == 'L' || uplo == 'l');
  pxtrsm<HT>('L', uplo, lower ? 'N' : 'C', 'N', n, nrhs, &one, a, ia, ja, desca, b, ib, jb, descb);
  pxtrsm<HT>('L', uplo, lower ? 'C' : 'N', 'N', n, nrhs, &one, a, ia, ja, desca, b, ib, jb, descb);
  if (info)
    *info = 0;
}

// dlaf::eigensolver::internal::generalized_to_standard (include/dlaf/eigensolver/gen_to_std.h:50,:101) through
// descriptors: preconditions of gen_to_std.h:52-60 / :103-113 (square, square blocks, same size and blocks)
template <class HT>
int gen_to_std_c(int ctx, char uplo, HT* a, const DLAF_descriptor& da, const HT* b, const DLAF_descriptor& db) {
  using DT = typename DevType<HT>::type;
  check_cholesky_desc(da);
  check_cholesky_desc(db);
  if (!(uplo == 'L' || uplo == 'l' || uplo == 'U' || uplo == 'u'))
    fatal("[dlaf_mi355x] uplo must be 'L' or 'U', got '%c'\n", uplo);
  if (da.m != db.m || da.nb != db.nb || da.isrc != db.isrc || da.jsrc != db.jsrc)
    fatal("[dlaf_mi355x] gen_to_std: A (%d, block %d, source %d,%d) and B (%d, block %d, source %d,%d) must be "
          "distributed alike\n", da.m, da.nb, da.isrc, da.jsrc, db.m, db.nb, db.isrc, db.jsrc);
  Grid& g = grid_from_context(ctx);
  if (da.isrc < 0 || da.isrc >= g.nprow || da.jsrc < 0 || da.jsrc >= g.npcol)
    fatal("[dlaf_mi355x] source rank (%d,%d) outside the %d x %d grid\n", da.isrc, da.jsrc, g.nprow, g.npcol);
  return gen_to_std_host<DT>(&g, uplo, reinterpret_cast<DT*>(a), da.ld, reinterpret_cast<const DT*>(b), db.ld, da.m,
                             da.nb, da.isrc, da.jsrc);
}

// ScaLAPACK p?sygst / p?hegst argument list (ibtype 1 only: inv(L) A inv(L^H) / inv(U^H) A inv(U))
template <class HT, class RT>
void pxhegst(int ibtype, char uplo, int n, HT* a, int ia, int ja, const int desca[9], const HT* b, int ib, int jb,
             const int descb[9], RT* scale, int* info) {
  if (ibtype != 1)
    fatal("[dlaf_mi355x] p?hegst: only ibtype = 1 is built (got %d)\n", ibtype);
  if (desca[0] != 1 || descb[0] != 1)
    fatal("[dlaf_mi355x] desc[0] (dtype) must be 1\n");
  if (ia != 1 || ja != 1 || ib != 1 || jb != 1)
    fatal("[dlaf_mi355x] ia, ja, ib, jb must be 1\n");
  if (desca[1] != descb[1])
    fatal("[dlaf_mi355x] A and B live on different contexts (%d, %d)\n", desca[1], descb[1]);
  const DLAF_descriptor da = make_dlaf_descriptor(n, n, ia, ja, desca);
  const DLAF_descriptor db = make_dlaf_descriptor(n, n, ib, jb, descb);
  const int r = gen_to_std_c<HT>(desca[1], uplo, a, da, b, db);
  if (scale)
    *scale = RT(1);
  if (info)
    *info = r;
}

// dlaf::eigensolver::internal::reduction_to_band (include/dlaf/eigensolver/reduction_to_band.h:101-122) through the
// reference's descriptor conventions
template <class HT>
int red2band_c(int ctx, HT* a, const DLAF_descriptor& da, int band, HT* taus) {
  using DT = typename DevType<HT>::type;
  check_cholesky_desc(da);  // square matrix, square block, no offsets: reduction_to_band.h:103-106
  Grid& g = grid_from_context(ctx);
  if (da.isrc < 0 || da.isrc >= g.nprow || da.jsrc < 0 || da.jsrc >= g.npcol)
    fatal("[dlaf_mi355x] source rank (%d,%d) outside the %d x %d grid\n", da.isrc, da.jsrc, g.nprow, g.npcol);
  if (band < 2 || da.nb % band != 0)
    fatal("[dlaf_mi355x] reduction_to_band: band_size %d must be >= 2 and divide the block size %d "
          "(reduction_to_band.h:108-109)\n", band, da.nb);
  return reduction_to_band_host<DT>(&g, reinterpret_cast<DT*>(a), da.ld, da.m, da.nb, da.isrc, da.jsrc, band,
                                    reinterpret_cast<DT*>(taus));
}

template <class HT>
int bt_red2band_c(int ctx, int band, HT* c, const DLAF_descriptor& dc, const HT* v, const DLAF_descriptor& dv,
                  const HT* taus) {
  using DT = typename DevType<HT>::type;
  check_cholesky_desc(dv);
  Grid& g = grid_from_context(ctx);
  if (dc.i != 0 || dc.j != 0 || dc.mb != dc.nb || dc.nb != dv.nb || dc.m != dv.m || dc.isrc != dv.isrc)
    fatal("[dlaf_mi355x] bt_reduction_to_band: C (%d x %d, block %d x %d, row source %d) must have V's block size %d, "
          "row count %d and row source %d, and no offsets\n", dc.m, dc.n, dc.mb, dc.nb, dc.isrc, dv.nb, dv.m, dv.isrc);
  if (dc.jsrc < 0 || dc.jsrc >= g.npcol || dv.isrc < 0 || dv.isrc >= g.nprow || dv.jsrc < 0 || dv.jsrc >= g.npcol)
    fatal("[dlaf_mi355x] source rank outside the %d x %d grid\n", g.nprow, g.npcol);
  if (band < 2 || dv.nb % band != 0)
    fatal("[dlaf_mi355x] bt_reduction_to_band: band_size %d must be >= 2 and divide the block size %d\n", band, dv.nb);
  return bt_reduction_to_band_host<DT>(&g, band, reinterpret_cast<DT*>(c), dc.ld, dc.n, dc.jsrc,
                                       reinterpret_cast<const DT*>(v), dv.ld, dv.m, dv.nb, dv.isrc, dv.jsrc,
                                       reinterpret_cast<const DT*>(taus));
}


// ---- the eigensolver stages and drivers ------------------------------------------------------
template <class HT>
struct RealOf {
  using type = HT;
};
template <class R>
struct RealOf<std::complex<R>> {
  using type = R;
};

template <class HT>
int band_to_tridiag_c(int ctx, const HT* a, const DLAF_descriptor& da, int band, typename RealOf<HT>::type* d,
                      typename RealOf<HT>::type* e, HT* v, int ldv) {
  using DT = typename DevType<HT>::type;
  check_cholesky_desc(da);  // square matrix, square block, no offsets: band_to_tridiag.h:76-80
  Grid& g = grid_from_context(ctx);
  if (da.isrc < 0 || da.isrc >= g.nprow || da.jsrc < 0 || da.jsrc >= g.npcol)
    fatal("[dlaf_mi355x] source rank (%d,%d) outside the %d x %d grid\n", da.isrc, da.jsrc, g.nprow, g.npcol);
  if (band < 2 || da.nb % band != 0)
    fatal("[dlaf_mi355x] band_to_tridiagonal: band_size %d must be >= 2 and divide the block size %d "
          "(band_to_tridiag.h:78-80)\n", band, da.nb);
  if (ldv < std::max(1, da.m))
    fatal("[dlaf_mi355x] band_to_tridiagonal: ldv = %d < n = %d\n", ldv, da.m);
  return band_to_tridiag_host<DT>(&g, reinterpret_cast<const DT*>(a), da.ld, da.m, da.nb, da.isrc, da.jsrc, band, d, e,
                                  reinterpret_cast<DT*>(v), ldv);
}

// Eigensolver entry (src/c_api/eigensolver/eigensolver.h:33-75): descriptors of A and of the eigenvector matrix
template <class HT>
int eigensolver_c(int ctx, char uplo, HT* a, const DLAF_descriptor& da, typename RealOf<HT>::type* w, HT* z,
                  const DLAF_descriptor& dz) {
  using DT = typename DevType<HT>::type;
  check_cholesky_desc(da);
  Grid& g = grid_from_context(ctx);
  if (dz.i != 0 || dz.j != 0)
    fatal("[dlaf_mi355x] eigensolver: sub-matrix offsets of Z must be 0 (eigensolver.h:44-45 upstream)\n");
  if (dz.m != da.m || dz.n != da.n || dz.mb != da.mb || dz.nb != da.nb)
    fatal("[dlaf_mi355x] eigensolver: Z (%d x %d, block %d x %d) must have A's size %d x %d and block %d x %d\n", dz.m, dz.n,
          dz.mb, dz.nb, da.m, da.n, da.mb, da.nb);
  for (const DLAF_descriptor* d : {&da, &dz})
    if (d->isrc < 0 || d->isrc >= g.nprow || d->jsrc < 0 || d->jsrc >= g.npcol)
      fatal("[dlaf_mi355x] source rank (%d,%d) outside the %d x %d grid\n", d->isrc, d->jsrc, g.nprow, g.npcol);
  return hermitian_eigensolver_host<DT>(&g, uplo, reinterpret_cast<DT*>(a), da.ld, da.m, da.nb, da.isrc, da.jsrc, w,
                                        reinterpret_cast<DT*>(z), dz.ld, dz.isrc, dz.jsrc);
}

template <class HT>
int gen_eigensolver_c(int ctx, char uplo, HT* a, const DLAF_descriptor& da, HT* b, const DLAF_descriptor& db,
                      typename RealOf<HT>::type* w, HT* z, const DLAF_descriptor& dz, bool factorized) {
  using DT = typename DevType<HT>::type;
  check_cholesky_desc(da);
  check_cholesky_desc(db);
  Grid& g = grid_from_context(ctx);
  if (dz.i != 0 || dz.j != 0)
    fatal("[dlaf_mi355x] gen_eigensolver: sub-matrix offsets of Z must be 0\n");
  for (const DLAF_descriptor* d : {&db, &dz})
    if (d->m != da.m || d->n != da.n || d->mb != da.mb || d->nb != da.nb)
      fatal("[dlaf_mi355x] gen_eigensolver: B and Z must have A's size %d x %d and block %d x %d\n", da.m, da.n, da.mb,
            da.nb);
  for (const DLAF_descriptor* d : {&da, &db, &dz})
    if (d->isrc < 0 || d->isrc >= g.nprow || d->jsrc < 0 || d->jsrc >= g.npcol)
      fatal("[dlaf_mi355x] source rank (%d,%d) outside the %d x %d grid\n", d->isrc, d->jsrc, g.nprow, g.npcol);
  return hermitian_gen_eigensolver_host<DT>(&g, uplo, reinterpret_cast<DT*>(a), da.ld, reinterpret_cast<DT*>(b), db.ld,
                                            da.m, da.nb, da.isrc, da.jsrc, db.isrc, db.jsrc, w, reinterpret_cast<DT*>(z),
                                            dz.ld, dz.isrc, dz.jsrc, factorized);
}

// p?syevd / p?heevd and p?sygvd / p?hegvd argument lists (src/c_api/eigensolver/eigensolver.h:79-124)
template <class HT>
void pxheevd(char uplo, int n, HT* a, int ia, int ja, const int desca[9], typename RealOf<HT>::type* w, HT* z, int iz,
             int jz, const int descz[9], int* info) {
  if (desca[0] != 1 || descz[0] != 1)
    fatal("[dlaf_mi355x] desc[0] (dtype) must be 1\n");
  if (ia != 1 || ja != 1 || iz != 1 || jz != 1)
    fatal("[dlaf_mi355x] ia, ja, iz, jz must be 1\n");
  if (desca[1] != descz[1])
    fatal("[dlaf_mi355x] A and Z live on different contexts (%d, %d)\n", desca[1], descz[1]);
  const DLAF_descriptor da = make_dlaf_descriptor(n, n, ia, ja, desca);
  const DLAF_descriptor dz = make_dlaf_descriptor(n, n, iz, jz, descz);
  const int r = eigensolver_c<HT>(desca[1], uplo, a, da, w, z, dz);
  if (info)
    *info = r;
}
template <class HT>
void pxhegvd(char uplo, int n, HT* a, int ia, int ja, const int desca[9], HT* b, int ib, int jb, const int descb[9],
             typename RealOf<HT>::type* w, HT* z, int iz, int jz, const int descz[9], int* info, bool factorized) {
  if (desca[0] != 1 || descb[0] != 1 || descz[0] != 1)
    fatal("[dlaf_mi355x] desc[0] (dtype) must be 1\n");
  if (ia != 1 || ja != 1 || ib != 1 || jb != 1 || iz != 1 || jz != 1)
    fatal("[dlaf_mi355x] ia, ja, ib, jb, iz, jz must be 1\n");
  if (desca[1] != descb[1] || desca[1] != descz[1])
    fatal("[dlaf_mi355x] A, B and Z live on different contexts\n");
  const DLAF_descriptor da = make_dlaf_descriptor(n, n, ia, ja, desca);
  const DLAF_descriptor db = make_dlaf_descriptor(n, n, ib, jb, descb);
  const DLAF_descriptor dz = make_dlaf_descriptor(n, n, iz, jz, descz);
  const int r = gen_eigensolver_c<HT>(desca[1], uplo, a, da, b, db, w, z, dz, factorized);
  if (info)
    *info = r;
}

struct MatrixHandle {
  std::unique_ptr<MatrixBase> m;
  char type;
  int ctx;
};

template <class F>
int dispatch_type(char type, F&& f) {
  switch (type) {
    case 's': return f((float*) nullptr);
    case 'd': return f((double*) nullptr);
    case 'c': return f((cfloat*) nullptr);
    case 'z': return f((cdouble*) nullptr);
    default: return -2;
  }
}
}  // namespace

struct dlaf_mi355x_matrix_s : MatrixHandle {};

namespace {
static void* mpi_shim_handle() {
  static void* handle = []() -> void* {
    Dl_info info;
    std::string path = "libdlaf_mi355x_mpi.so";
    if (dladdr(reinterpret_cast<void*>(&dlaf_free_grid), &info) != 0 && info.dli_fname != nullptr) {
      std::string self = info.dli_fname;
      const size_t slash = self.rfind('/');
      if (slash != std::string::npos)
        path = self.substr(0, slash + 1) + path;
    }
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr)
      fatal("[dlaf_mi355x] the MPI-typed grid entry points need %s (built by `make -C dla_future_amd/csrc mpi` when an "
            "MPI is installed): %s\n", path.c_str(), dlerror());
    return h;
  }();
  return handle;
}
template <class F>
static F mpi_shim_symbol(const char* name, F self) {
  F f = reinterpret_cast<F>(dlsym(mpi_shim_handle(), name));
  if (f == nullptr || f == self)
    fatal("[dlaf_mi355x] libdlaf_mi355x_mpi.so does not define %s\n", name);
  return f;
}
}  // namespace

// =================================================================================== dlaf_c
extern "C" {

void dlaf_initialize(int, const char**, int argc_dlaf, const char** argv_dlaf) noexcept {
  const bool first = !runtime_initialized();
  runtime_init();
  // tune parameters (src/init.cpp:211-221: the environment variable first, the command-line option overrides it)
  if (first) {
    if (const char* e = std::getenv("DLAF_EIGENSOLVER_MIN_BAND"))
      set_eigensolver_min_band(std::atoi(e));
    for (int i = 0; argv_dlaf && i < argc_dlaf; ++i) {
      static const char opt[] = "--dlaf:eigensolver-min-band";
      if (!argv_dlaf[i] || std::strncmp(argv_dlaf[i], opt, sizeof(opt) - 1) != 0)
        continue;
      const char* v = argv_dlaf[i] + sizeof(opt) - 1;
      if (*v == '=')
        set_eigensolver_min_band(std::atoi(v + 1));
      else if (*v == 0 && i + 1 < argc_dlaf && argv_dlaf[i + 1])
        set_eigensolver_min_band(std::atoi(argv_dlaf[++i]));
    }
  }
  for (int i = 0; first && argv_dlaf && i < argc_dlaf; ++i)
    if (argv_dlaf[i] && std::strcmp(argv_dlaf[i], "--dlaf:print-config") == 0) {
      int dev = -1;
      (void) hipGetDevice(&dev);
      hipDeviceProp_t p;
      (void) hipGetDeviceProperties(&p, dev);
      std::printf("DLA-Future MI355X build: device %d (%s, %d CUs), diag block %d\n", dev, p.gcnArchName,
                  p.multiProcessorCount, kDiagBlock);
    }
}

void dlaf_finalize(void) noexcept {
  g_grids.clear();
  runtime_finalize();
}

void dlaf_free_grid(int context) noexcept {
  g_grids.erase(context);
}

DLAF_descriptor make_dlaf_descriptor(const int m, const int n, const int i, const int j, const int desc[9]) noexcept {
  if (i != 1 || j != 1)
    fatal("[dlaf_mi355x] make_dlaf_descriptor: i = %d, j = %d must be 1\n", i, j);
  DLAF_descriptor d = {m, n, desc[4], desc[5], desc[6], desc[7], i - 1, j - 1, desc[8]};
  return d;
}

int dlaf_cholesky_factorization_s(const int ctx, const char uplo, float* a, const DLAF_descriptor d) noexcept {
  return cholesky_host<float>(ctx, uplo, a, d);
}
int dlaf_cholesky_factorization_d(const int ctx, const char uplo, double* a, const DLAF_descriptor d) noexcept {
  return cholesky_host<double>(ctx, uplo, a, d);
}
int dlaf_cholesky_factorization_c(const int ctx, const char uplo, dlaf_complex_c* a, const DLAF_descriptor d) noexcept {
  return cholesky_host<std::complex<float>>(ctx, uplo, a, d);
}
int dlaf_cholesky_factorization_z(const int ctx, const char uplo, dlaf_complex_z* a, const DLAF_descriptor d) noexcept {
  return cholesky_host<std::complex<double>>(ctx, uplo, a, d);
}

void dlaf_pspotrf(const char uplo, const int n, float* a, const int ia, const int ja, const int desca[9],
                  int* info) noexcept {
  pxpotrf<float>(uplo, n, a, ia, ja, desca, info);
}
void dlaf_pdpotrf(const char uplo, const int n, double* a, const int ia, const int ja, const int desca[9],
                  int* info) noexcept {
  pxpotrf<double>(uplo, n, a, ia, ja, desca, info);
}
void dlaf_pcpotrf(const char uplo, const int n, dlaf_complex_c* a, const int ia, const int ja, const int desca[9],
                  int* info) noexcept {
  pxpotrf<std::complex<float>>(uplo, n, a, ia, ja, desca, info);
}
void dlaf_pzpotrf(const char uplo, const int n, dlaf_complex_z* a, const int ia, const int ja, const int desca[9],
                  int* info) noexcept {
  pxpotrf<std::complex<double>>(uplo, n, a, ia, ja, desca, info);
}

// =================================================================================== extensions
const char* dlaf_mi355x_version(void) noexcept {
  return "dlaf_mi355x 0.1 gfx950";
}

int dlaf_mi355x_create_grid_single(void) noexcept {
  return register_grid(make_grid(1, 0, 1, 1, 'R'));
}

void dlaf_mi355x_rccl_unique_id(void* out) noexcept {
  runtime_init();
  rccl_get_unique_id(out);
}

int dlaf_mi355x_create_grid_rccl(const void* uid, int nranks, int rank, int nprow, int npcol, char order) noexcept {
  auto g = make_grid(nranks, rank, nprow, npcol, order);
  if (!g)
    return -1;
  runtime_init();
  // a 1-process grid needs no communicator; DLAF_MI355X_RCCL_SINGLE=1 creates them anyway (communicator
  // init / split / teardown exercised on a one-GPU box)
  const char* force = std::getenv("DLAF_MI355X_RCCL_SINGLE");
  if (nranks > 1 || (force && force[0] == '1'))
    g->transport = make_rccl_transport(uid, nranks, rank, nprow, npcol, g->myrow, g->mycol);
  return register_grid(std::move(g));
}

int dlaf_mi355x_create_grid_host(int nranks, int rank, int nprow, int npcol, char order, dlaf_mi355x_bcast_fn bcast,
                                 dlaf_mi355x_barrier_fn barrier, void* user) noexcept {
  auto g = make_grid(nranks, rank, nprow, npcol, order);
  if (!g || (nranks > 1 && !bcast))
    return -1;
  if (nranks > 1) {
    g->host_bcast = bcast;
    g->host_user = user;
    g->host_barrier = barrier;
    // the transport itself (pinned staging buffer, HIP) is created on first use by a matrix
  }
  return register_grid(std::move(g));
}

int dlaf_mi355x_grid_rekey(int ctx, int new_ctx) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  if (ctx == new_ctx)
    return 0;
  if (g_grids.find(new_ctx) != g_grids.end())
    return -2;
  std::unique_ptr<Grid> g = std::move(it->second);
  g_grids.erase(it);
  g_grids.emplace(new_ctx, std::move(g));
  return 0;
}

int dlaf_mi355x_grid_on_free(int ctx, void (*fn)(void*), void* user) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  it->second->on_free = fn;
  it->second->on_free_user = user;
  return 0;
}

int dlaf_mi355x_grid_comm_log(int ctx, int enable) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  Grid& g = *it->second;
  g.comm_log_on = enable != 0;
  g.comm_log.clear();
  if (g.comm_log_on && runtime_initialized())
    (void) grid_transport(g);
  return 0;
}

long dlaf_mi355x_grid_comm_log_read(int ctx, long* out, long cap_events) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  const auto& log = it->second->comm_log;
  for (long i = 0; out && i < cap_events && i < (long) log.size(); ++i) {
    out[4 * i + 0] = log[(size_t) i].kind;
    out[4 * i + 1] = log[(size_t) i].root;
    out[4 * i + 2] = log[(size_t) i].bytes;
    out[4 * i + 3] = log[(size_t) i].group;
  }
  return (long) log.size();
}

int dlaf_mi355x_grid_info(int ctx, int* nprow, int* npcol, int* myrow, int* mycol) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  const Grid& g = *it->second;
  if (nprow) *nprow = g.nprow;
  if (npcol) *npcol = g.npcol;
  if (myrow) *myrow = g.myrow;
  if (mycol) *mycol = g.mycol;
  return 0;
}

int dlaf_mi355x_grid_barrier(int ctx) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  Grid& g = *it->second;
  if (g.transport)
    g.transport->barrier(nullptr);
  else if (g.host_barrier)
    return g.host_barrier(g.host_user);
  else if (runtime_initialized())
    (void) hipDeviceSynchronize();
  return 0;
}

int dlaf_mi355x_grid_selftest(int ctx, size_t bytes) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  return grid_selftest(*it->second, bytes);
}

int dlaf_mi355x_grid_host_bcast(int ctx, int axis, int root, void* host_buf, size_t bytes) noexcept {
  // exercises the grid's broadcast callback with a caller-owned HOST buffer (no GPU involved):
  // lets CPU-only multi-process tests check the row/column communicator wiring of a host grid
  auto it = g_grids.find(ctx);
  if (it == g_grids.end() || (axis != 0 && axis != 1))
    return -1;
  Grid& g = *it->second;
  if (!g.host_bcast)
    return g.nranks == 1 ? 0 : -2;
  return g.host_bcast(g.host_user, axis, root, host_buf, bytes);
}

#define DLAF_MI355X_TRSM_ENTRY(S, HT, CT)                                                                        \
  int dlaf_mi355x_triangular_solver_##S(int ctx, char side, char uplo, char op, char diag, const CT* alpha,      \
                                        const CT* a, DLAF_descriptor desca, CT* b, DLAF_descriptor descb) noexcept { \
    return triangular_solver_c<HT>(ctx, side, uplo, op, diag, reinterpret_cast<const HT*>(alpha),               \
                                   reinterpret_cast<const HT*>(a), desca, reinterpret_cast<HT*>(b), descb);     \
  }                                                                                                             \
  void dlaf_mi355x_p##S##trsm(char side, char uplo, char op, char diag, int m, int n, const CT* alpha, const CT* a, \
                              int ia, int ja, const int desca[9], CT* b, int ib, int jb, const int descb[9]) noexcept { \
    pxtrsm<HT>(side, uplo, op, diag, m, n, reinterpret_cast<const HT*>(alpha), reinterpret_cast<const HT*>(a), ia, \
               ja, desca, reinterpret_cast<HT*>(b), ib, jb, descb);                                              \
  }
#define DLAF_MI355X_POTRS_ENTRY(S, HT, CT)                                                                       \
  void dlaf_mi355x_p##S##potrs(char uplo, int n, int nrhs, const CT* a, int ia, int ja, const int desca[9], CT* b,  \
                               int ib, int jb, const int descb[9], int* info) noexcept {                        \
    pxpotrs<HT>(uplo, n, nrhs, reinterpret_cast<const HT*>(a), ia, ja, desca, reinterpret_cast<HT*>(b), ib, jb,  \
                descb, info);                                                                                   \
  }
DLAF_MI355X_POTRS_ENTRY(s, float, float)
DLAF_MI355X_POTRS_ENTRY(d, double, double)
DLAF_MI355X_POTRS_ENTRY(c, std::complex<float>, dlaf_complex_c)
DLAF_MI355X_POTRS_ENTRY(z, std::complex<double>, dlaf_complex_z)
#undef DLAF_MI355X_POTRS_ENTRY
DLAF_MI355X_TRSM_ENTRY(s, float, float)
DLAF_MI355X_TRSM_ENTRY(d, double, double)
DLAF_MI355X_TRSM_ENTRY(c, std::complex<float>, dlaf_complex_c)
DLAF_MI355X_TRSM_ENTRY(z, std::complex<double>, dlaf_complex_z)
#undef DLAF_MI355X_TRSM_ENTRY

#define DLAF_MI355X_HEGST_ENTRY(S, HT, CT, RT)                                                                    \
  int dlaf_mi355x_generalized_to_standard_##S(int ctx, char uplo, CT* a, DLAF_descriptor desca, const CT* b,       \
                                              DLAF_descriptor descb) noexcept {                                  \
    return gen_to_std_c<HT>(ctx, uplo, reinterpret_cast<HT*>(a), desca, reinterpret_cast<const HT*>(b), descb);    \
  }                                                                                                              \
  void dlaf_mi355x_p##S##hegst(int ibtype, char uplo, int n, CT* a, int ia, int ja, const int desca[9], const CT* b, \
                               int ib, int jb, const int descb[9], RT* scale, int* info) noexcept {               \
    pxhegst<HT, RT>(ibtype, uplo, n, reinterpret_cast<HT*>(a), ia, ja, desca, reinterpret_cast<const HT*>(b), ib, \
                    jb, descb, scale, info);                                                                     \
  }
DLAF_MI355X_HEGST_ENTRY(s, float, float, float)
DLAF_MI355X_HEGST_ENTRY(d, double, double, double)
DLAF_MI355X_HEGST_ENTRY(c, std::complex<float>, dlaf_complex_c, float)
DLAF_MI355X_HEGST_ENTRY(z, std::complex<double>, dlaf_complex_z, double)
#undef DLAF_MI355X_HEGST_ENTRY

#define DLAF_MI355X_R2B_ENTRY(S, HT, CT)                                                                          \
  int dlaf_mi355x_reduction_to_band_##S(int ctx, CT* a, DLAF_descriptor desca, int band, CT* taus) noexcept {     \
    return red2band_c<HT>(ctx, reinterpret_cast<HT*>(a), desca, band, reinterpret_cast<HT*>(taus));               \
  }                                                                                                              \
  int dlaf_mi355x_bt_reduction_to_band_##S(int ctx, int band, CT* c, DLAF_descriptor descc, const CT* v,          \
                                           DLAF_descriptor descv, const CT* taus) noexcept {                     \
    return bt_red2band_c<HT>(ctx, band, reinterpret_cast<HT*>(c), descc, reinterpret_cast<const HT*>(v), descv,   \
                             reinterpret_cast<const HT*>(taus));                                                 \
  }
DLAF_MI355X_R2B_ENTRY(s, float, float)
DLAF_MI355X_R2B_ENTRY(d, double, double)
DLAF_MI355X_R2B_ENTRY(c, std::complex<float>, dlaf_complex_c)
DLAF_MI355X_R2B_ENTRY(z, std::complex<double>, dlaf_complex_z)
#undef DLAF_MI355X_R2B_ENTRY


#define DLAF_MI355X_EIG_ENTRY(S, KIND, HT, CT, RT, PEV, PGV)                                                          \
  int dlaf_mi355x_band_to_tridiagonal_##S(int ctx, const CT* a, DLAF_descriptor desca, int band, RT* d, RT* e, CT* v, \
                                          int ldv) noexcept {                                                       \
    return band_to_tridiag_c<HT>(ctx, reinterpret_cast<const HT*>(a), desca, band, d, e, reinterpret_cast<HT*>(v), ldv); \
  }                                                                                                                  \
  int dlaf_mi355x_bt_band_to_tridiagonal_##S(int band, int n, int ncols, const CT* v, int ldv, CT* e, int lde) noexcept { \
    using DT = typename DevType<HT>::type;                                                                           \
    runtime_init();                                                                                                  \
    if (band < 2 || ldv < std::max(1, n) || lde < std::max(1, n))                                                    \
      fatal("[dlaf_mi355x] bt_band_to_tridiagonal: bad band_size / leading dimensions\n");                           \
    return bt_band_to_tridiag_host<DT>(n, band, reinterpret_cast<const DT*>(v), ldv, reinterpret_cast<DT*>(e), lde,   \
                                       ncols);                                                                       \
  }                                                                                                                  \
  int dlaf_##KIND##_eigensolver_##S(const int ctx, const char uplo, CT* a, const DLAF_descriptor desca, RT* w, CT* z, \
                                    const DLAF_descriptor descz) noexcept {                                          \
    return eigensolver_c<HT>(ctx, uplo, reinterpret_cast<HT*>(a), desca, w, reinterpret_cast<HT*>(z), descz);        \
  }                                                                                                                  \
  int dlaf_##KIND##_generalized_eigensolver_##S(const int ctx, const char uplo, CT* a, const DLAF_descriptor desca,   \
                                                CT* b, const DLAF_descriptor descb, RT* w, CT* z,                    \
                                                const DLAF_descriptor descz) noexcept {                              \
    return gen_eigensolver_c<HT>(ctx, uplo, reinterpret_cast<HT*>(a), desca, reinterpret_cast<HT*>(b), descb, w,      \
                                 reinterpret_cast<HT*>(z), descz, false);                                            \
  }                                                                                                                  \
  int dlaf_##KIND##_generalized_eigensolver_factorized_##S(const int ctx, const char uplo, CT* a,                     \
                                                           const DLAF_descriptor desca, CT* b,                       \
                                                           const DLAF_descriptor descb, RT* w, CT* z,                \
                                                           const DLAF_descriptor descz) noexcept {                   \
    return gen_eigensolver_c<HT>(ctx, uplo, reinterpret_cast<HT*>(a), desca, reinterpret_cast<HT*>(b), descb, w,      \
                                 reinterpret_cast<HT*>(z), descz, true);                                             \
  }                                                                                                                  \
  void dlaf_##PEV(const char uplo, const int n, CT* a, const int ia, const int ja, const int desca[9], RT* w, CT* z,  \
                  const int iz, const int jz, const int descz[9], int* info) noexcept {                              \
    pxheevd<HT>(uplo, n, reinterpret_cast<HT*>(a), ia, ja, desca, w, reinterpret_cast<HT*>(z), iz, jz, descz, info);  \
  }                                                                                                                  \
  void dlaf_##PGV(const char uplo, const int n, CT* a, const int ia, const int ja, const int desca[9], CT* b,         \
                  const int ib, const int jb, const int descb[9], RT* w, CT* z, const int iz, const int jz,          \
                  const int descz[9], int* info) noexcept {                                                          \
    pxhegvd<HT>(uplo, n, reinterpret_cast<HT*>(a), ia, ja, desca, reinterpret_cast<HT*>(b), ib, jb, descb, w,         \
                reinterpret_cast<HT*>(z), iz, jz, descz, info, false);                                               \
  }                                                                                                                  \
  void dlaf_##PGV##_factorized(const char uplo, const int n, CT* a, const int ia, const int ja, const int desca[9],   \
                               CT* b, const int ib, const int jb, const int descb[9], RT* w, CT* z, const int iz,    \
                               const int jz, const int descz[9], int* info) noexcept {                               \
    pxhegvd<HT>(uplo, n, reinterpret_cast<HT*>(a), ia, ja, desca, reinterpret_cast<HT*>(b), ib, jb, descb, w,         \
                reinterpret_cast<HT*>(z), iz, jz, descz, info, true);                                                \
  }
DLAF_MI355X_EIG_ENTRY(s, symmetric, float, float, float, pssyevd, pssygvd)
DLAF_MI355X_EIG_ENTRY(d, symmetric, double, double, double, pdsyevd, pdsygvd)
DLAF_MI355X_EIG_ENTRY(c, hermitian, std::complex<float>, dlaf_complex_c, float, pcheevd, pchegvd)
DLAF_MI355X_EIG_ENTRY(z, hermitian, std::complex<double>, dlaf_complex_z, double, pzheevd, pzhegvd)
#undef DLAF_MI355X_EIG_ENTRY

int dlaf_mi355x_tridiagonal_eigensolver_s(int n, int nb, const float* d, const float* e, float* w, float* z,
                                          int ldz) noexcept {
  runtime_init();
  return tridiag_solver_host<float>(n, nb, d, e, w, z, ldz);
}
int dlaf_mi355x_tridiagonal_eigensolver_d(int n, int nb, const double* d, const double* e, double* w, double* z,
                                          int ldz) noexcept {
  runtime_init();
  return tridiag_solver_host<double>(n, nb, d, e, w, z, ldz);
}
int dlaf_mi355x_eigensolver_profile(double ms[5]) noexcept {
  eigensolver_last_profile(ms);
  return 0;
}

int dlaf_mi355x_red2band_panel_stats(long* blocked, long* fallback) noexcept {
  red2band_last_panels(blocked, fallback);
  return 0;
}
long dlaf_mi355x_workspace_pool_release(void) noexcept {
  const long held = (long) pool_idle_bytes();
  pool_release();
  return held;
}
int dlaf_mi355x_get_eigensolver_min_band(void) noexcept {
  return eigensolver_min_band();
}
void dlaf_mi355x_set_eigensolver_min_band(int b_min) noexcept {
  set_eigensolver_min_band(b_min);
}
int dlaf_mi355x_get_band_size(int nb) noexcept {
  return get_band_size(nb);
}
int dlaf_mi355x_red2band_profile(double* ms, double* flops) noexcept {
  red2band_last_profile(ms, flops);
  return 0;
}

// ---- device-resident operands for the solver ----------------------------------------------------------------
struct dlaf_mi355x_gmatrix_s {
  std::unique_ptr<MatrixBase> m;
  int ctx;
};

int dlaf_mi355x_gmatrix_create(int ctx, char type, DLAF_descriptor d, dlaf_mi355x_gmatrix_t* out) noexcept {
  if (!out)
    return -1;
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  Grid* g = it->second.get();
  if (d.i != 0 || d.j != 0 || d.m < 0 || d.n < 0 || d.mb != d.nb || d.nb < 1)
    return -3;  // square blocks only in this build
  if (d.isrc < 0 || d.isrc >= g->nprow || d.jsrc < 0 || d.jsrc >= g->npcol)
    return -3;
  MatrixBase* m = general_matrix_create(g, type, d.m, d.n, d.nb, d.isrc, d.jsrc);
  if (!m)
    return -2;
  auto* h = new dlaf_mi355x_gmatrix_s;
  h->m.reset(m);
  h->ctx = ctx;
  *out = h;
  return 0;
}
void dlaf_mi355x_gmatrix_destroy(dlaf_mi355x_gmatrix_t h) noexcept {
  delete h;
}
int dlaf_mi355x_gmatrix_upload(dlaf_mi355x_gmatrix_t h, const void* host, int ld) noexcept {
  if (!h || !h->m)
    return -1;
  general_matrix_transfer(h->m.get(), const_cast<void*>(host), ld, true);
  return 0;
}
int dlaf_mi355x_gmatrix_download(dlaf_mi355x_gmatrix_t h, void* host, int ld) noexcept {
  if (!h || !h->m)
    return -1;
  general_matrix_transfer(h->m.get(), host, ld, false);
  return 0;
}
int dlaf_mi355x_triangular_solver_device(char side, char uplo, char op, char diag, const void* alpha,
                                         dlaf_mi355x_matrix_t a, dlaf_mi355x_gmatrix_t b) noexcept {
  if (!a || !a->m || !b || !b->m || a->ctx != b->ctx)
    return -1;
  auto is = [](char c, const char* set) { return c != 0 && std::strchr(set, c) != nullptr; };
  if (!is(side, "LlRr") || !is(uplo, "LlUu") || !is(op, "NnTtCc") || !is(diag, "NnUu"))
    fatal("[dlaf_mi355x] triangular solver: bad side/uplo/op/diag '%c' '%c' '%c' '%c'\n", side, uplo, op, diag);
  return triangular_solver_device(side, uplo, op, diag, alpha, a->m.get(), b->m.get());
}
// A X = B from the resident factor (p?potrs without leaving HBM): two solves, nothing crosses PCIe in between
int dlaf_mi355x_potrs_device(char uplo, dlaf_mi355x_matrix_t factor, dlaf_mi355x_gmatrix_t b) noexcept {
  if (!factor || !factor->m || !b || !b->m)
    return -1;
  const bool lower = (uplo == 'L' || uplo == 'l');
  const double one_d[2] = {1.0, 0.0};
  const float one_f[2] = {1.0f, 0.0f};
  const void* one = (factor->type == 's' || factor->type == 'c') ? (const void*) one_f : (const void*) one_d;
  int r = dlaf_mi355x_triangular_solver_device('L', uplo, lower ? 'N' : 'C', 'N', one, factor, b);
  if (r != 0)
    return r;
  return dlaf_mi355x_triangular_solver_device('L', uplo, lower ? 'C' : 'N', 'N', one, factor, b);
}

int dlaf_mi355x_solver_profile(double* ms, double* flops) noexcept {
  solver_last_profile(ms, flops);
  return 0;
}

int dlaf_mi355x_matrix_create(int ctx, char type, char uplo, DLAF_descriptor d, dlaf_mi355x_matrix_t* out) noexcept {
  if (!out)
    return -1;
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  Grid* g = it->second.get();
  if (d.i != 0 || d.j != 0 || d.m != d.n || d.mb != d.nb || d.nb < 1 || d.m < 0)
    return -3;
  if (d.isrc < 0 || d.isrc >= g->nprow || d.jsrc < 0 || d.jsrc >= g->npcol)
    return -3;
  if (!(uplo == 'L' || uplo == 'l' || uplo == 'U' || uplo == 'u'))
    return -4;
  auto* h = new dlaf_mi355x_matrix_s;
  h->type = type;
  h->ctx = ctx;
  const int r = dispatch_type(type, [&](auto* tag) {
    using DT = std::remove_pointer_t<decltype(tag)>;
    auto m = std::make_unique<DeviceMatrix<DT>>();
    m->create(g, uplo, d.m, d.nb, d.isrc, d.jsrc);
    h->m = std::move(m);
    return 0;
  });
  if (r != 0) {
    delete h;
    return r;
  }
  *out = h;
  return 0;
}

void dlaf_mi355x_matrix_destroy(dlaf_mi355x_matrix_t m) noexcept {
  delete m;
}

#define WITH_MATRIX(handle, body)                                       \
  if (!(handle) || !(handle)->m)                                        \
    return -1;                                                          \
  return dispatch_type((handle)->type, [&](auto* tag) -> int {         \
    using DT = std::remove_pointer_t<decltype(tag)>;                    \
    auto& M = static_cast<DeviceMatrix<DT>&>(*(handle)->m);             \
    body                                                                \
  });

int dlaf_mi355x_matrix_upload(dlaf_mi355x_matrix_t h, const void* host, int ld) noexcept {
  WITH_MATRIX(h, M.upload(static_cast<const DT*>(host), ld); return 0;)
}
int dlaf_mi355x_matrix_download(dlaf_mi355x_matrix_t h, void* host, int ld) noexcept {
  WITH_MATRIX(h, M.download(static_cast<DT*>(host), ld); return 0;)
}
int dlaf_mi355x_matrix_copy(dlaf_mi355x_matrix_t dst, dlaf_mi355x_matrix_t src) noexcept {
  if (!src || !dst || src->type != dst->type)
    return -1;
  WITH_MATRIX(dst, M.copy_from(static_cast<DeviceMatrix<DT>&>(*src->m)); return 0;)
}
int dlaf_mi355x_matrix_fetch_tile(dlaf_mi355x_matrix_t h, long gi, long gj, void* host, int ld) noexcept {
  WITH_MATRIX(h, return M.fetch_tile(gi, gj, static_cast<DT*>(host), ld) ? 0 : 1;)
}
int dlaf_mi355x_cholesky_start(dlaf_mi355x_matrix_t h) noexcept {
  WITH_MATRIX(h, M.factorize_async(); return 0;)
}
int dlaf_mi355x_cholesky_wait(dlaf_mi355x_matrix_t h) noexcept {
  WITH_MATRIX(h, return M.wait();)
}
int dlaf_mi355x_cholesky_factorization_device(dlaf_mi355x_matrix_t h) noexcept {
  WITH_MATRIX(h, return M.factorize();)
}
int dlaf_mi355x_matrix_local_info(dlaf_mi355x_matrix_t h) noexcept {
  WITH_MATRIX(h, return M.local_info;)
}
int dlaf_mi355x_update_launch_stats(long* persistent, long* exclusive) noexcept {
  long a = 0, b = 0;
  update_launch_stats(&a, &b);
  if (persistent)
    *persistent = a;
  if (exclusive)
    *exclusive = b;
  return 0;
}
int dlaf_mi355x_potrf_trace(unsigned long long* out) noexcept {
  unsigned long long* tb = potrf_coop_trace_buffer();
  if (tb == nullptr || out == nullptr)
    return 1;
  (void) hipDeviceSynchronize();
  return hipMemcpy(out, tb, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

int dlaf_mi355x_cholesky_residual(dlaf_mi355x_matrix_t original, dlaf_mi355x_matrix_t factor, double* max_diff,
                                  double* max_a) noexcept {
  if (!original || !factor || original->type != factor->type)
    return -1;
  WITH_MATRIX(original, M.residual_of(static_cast<DeviceMatrix<DT>&>(*factor->m), max_diff, max_a); return 0;)
}

int dlaf_mi355x_generalized_to_standard_device(dlaf_mi355x_matrix_t a, dlaf_mi355x_matrix_t l) noexcept {
  if (!a || !l || a->type != l->type)
    return -1;
  WITH_MATRIX(a, return gen_to_std_device(M, static_cast<DeviceMatrix<DT>&>(*l->m));)
}

int dlaf_mi355x_reduction_to_band_device(dlaf_mi355x_matrix_t a, int band, void* taus) noexcept {
  WITH_MATRIX(a, return reduction_to_band_device(M, band, static_cast<DT*>(taus));)
}
int dlaf_mi355x_bt_reduction_to_band_device(int band, dlaf_mi355x_gmatrix_t c, dlaf_mi355x_matrix_t v,
                                            const void* taus) noexcept {
  if (!c || !c->m || !v || !v->m || c->m->type != v->type)
    return -1;
  WITH_MATRIX(v, return bt_reduction_to_band_device(band, static_cast<GeneralMatrix<DT>&>(*c->m).m, M,
                                                    static_cast<const DT*>(taus));)
}

int dlaf_mi355x_matrix_trsm_profile(dlaf_mi355x_matrix_t h, int reps, double* ms, double* flops, double* bytes) noexcept {
  WITH_MATRIX(h, const double t = M.trsm_profile(reps, flops, bytes); if (ms) *ms = t; return 0;)
}

int dlaf_mi355x_matrix_profile(dlaf_mi355x_matrix_t h, int kind, double* ms, long* launches, double* flops,
                               double* bytes) noexcept {
  if (kind < 0 || kind > 3)
    return -2;
  WITH_MATRIX(h, const auto& p = M.prof[kind]; if (ms) *ms = p.ms; if (launches) *launches = p.launches;
              if (flops) *flops = p.flops; if (bytes) *bytes = p.bytes; return 0;)
}

int dlaf_mi355x_set_random_hpd(int ctx, char type, void* host, DLAF_descriptor d, int nthreads) noexcept {
  auto it = g_grids.find(ctx);
  if (it == g_grids.end())
    return -1;
  const Grid& g = *it->second;
  if (d.m != d.n || d.mb != d.nb || d.nb < 1)
    return -3;
  Axis rows{d.m, d.nb, g.nprow, g.myrow, d.isrc}, cols{d.n, d.nb, g.npcol, g.mycol, d.jsrc};
  switch (type) {
    case 's': set_random_hpd_local(static_cast<float*>(host), d.ld, d.m, d.nb, rows, cols, nthreads); break;
    case 'd': set_random_hpd_local(static_cast<double*>(host), d.ld, d.m, d.nb, rows, cols, nthreads); break;
    case 'c': set_random_hpd_local(static_cast<std::complex<float>*>(host), d.ld, d.m, d.nb, rows, cols, nthreads); break;
    case 'z': set_random_hpd_local(static_cast<std::complex<double>*>(host), d.ld, d.m, d.nb, rows, cols, nthreads); break;
    default: return -2;
  }
  return 0;
}

int dlaf_mi355x_tile_potrf(char type, char uplo, int n, void* a, int lda) noexcept {
  return dispatch_type(type, [&](auto* tag) {
    using DT = std::remove_pointer_t<decltype(tag)>;
    return tile_potrf<DT>(uplo, n, static_cast<DT*>(a), lda);
  });
}
int dlaf_mi355x_tile_trsm(char type, char uplo, int m, int n, const void* a, int lda, void* b, int ldb) noexcept {
  return dispatch_type(type, [&](auto* tag) {
    using DT = std::remove_pointer_t<decltype(tag)>;
    tile_trsm<DT>(uplo, m, n, static_cast<const DT*>(a), lda, static_cast<DT*>(b), ldb);
    return 0;
  });
}
int dlaf_mi355x_tile_herk(char type, char uplo, int n, int k, const void* a, int lda, void* c, int ldc) noexcept {
  return dispatch_type(type, [&](auto* tag) {
    using DT = std::remove_pointer_t<decltype(tag)>;
    tile_herk<DT>(uplo, n, k, static_cast<const DT*>(a), lda, static_cast<DT*>(c), ldc);
    return 0;
  });
}
int dlaf_mi355x_tile_gemm(char type, char uplo, int m, int n, int k, const void* a, int lda, const void* b, int ldb,
                          void* c, int ldc) noexcept {
  return dispatch_type(type, [&](auto* tag) {
    using DT = std::remove_pointer_t<decltype(tag)>;
    tile_gemm<DT>(uplo, m, n, k, static_cast<const DT*>(a), lda, static_cast<const DT*>(b), ldb, static_cast<DT*>(c), ldc);
    return 0;
  });
}

int dlaf_mi355x_dist_owner(long gt, int gs, int src) noexcept {
  return Axis{0, 1, gs, 0, src}.owner(gt);
}
long dlaf_mi355x_dist_local_tile(long gt, int gs, int rank, int src) noexcept {
  return Axis{0, 1, gs, rank, src}.local_of(gt);
}
long dlaf_mi355x_dist_next_local_tile(long gt, int gs, int rank, int src) noexcept {
  return Axis{0, 1, gs, rank, src}.next_local(gt);
}
long dlaf_mi355x_dist_global_tile(long lt, int gs, int rank, int src) noexcept {
  return Axis{0, 1, gs, rank, src}.global_of(lt);
}
long dlaf_mi355x_dist_local_size(long n, int nb, int gs, int rank, int src) noexcept {
  return Axis{n, nb, gs, rank, src}.local_size();
}
long dlaf_mi355x_dist_local_tiles(long n, int nb, int gs, int rank, int src) noexcept {
  return Axis{n, nb, gs, rank, src}.local_tiles();
}


// ---- MPI-typed entries of the reference's grid.h (dlaf_create_grid, grid_ordering, dlaf_create_grid_from_blacs) -----
// They live in the optional shim libdlaf_mi355x_mpi.so (mpi_grid.cpp), the only object of this build that sees
// <mpi.h>.  So that a caller of the reference's interface can link -ldlaf_mi355x alone, as it links -ldlaf upstream,
// the core exports forwarders under the same names: the first call loads the shim from this library's directory and
// jumps to its definition.  MPI_Comm is an int (MPICH ABI) or a pointer (Open MPI): either way the first integer
// argument register of the x86-64 psABI, which a forwarder declared with a pointer-sized first parameter passes on
// untouched.  A program that links the shim itself binds to the shim's definitions directly (it comes first in the
// link order) and never gets here.
int dlaf_create_grid(void* comm, int nprow, int npcol, char order) noexcept {
  using fn = int (*)(void*, int, int, char);
  static const fn f = mpi_shim_symbol<fn>("dlaf_create_grid", &dlaf_create_grid);
  return f(comm, nprow, npcol, order);
}
char grid_ordering(void* comm, int nprow, int npcol, int myprow, int mypcol) noexcept {
  using fn = char (*)(void*, int, int, int, int);
  static const fn f = mpi_shim_symbol<fn>("grid_ordering", &grid_ordering);
  return f(comm, nprow, npcol, myprow, mypcol);
}
void dlaf_create_grid_from_blacs(int blacs_ctxt) noexcept {
  using fn = void (*)(int);
  static const fn f = mpi_shim_symbol<fn>("dlaf_create_grid_from_blacs", &dlaf_create_grid_from_blacs);
  f(blacs_ctxt);
}

}  // extern "C"
