// tile_matrix.hpp -- a general (rectangular) block-cyclic matrix in the device tile layout: the right-hand side of
// the triangular solver (solver.cpp) and the matrix the back-transformations act on (red2band.cpp).
// Reference: matrix/matrix.h with a tileLayout (matrix/layout_info.h:140-159).
#pragma once
#include "runtime.hpp"

namespace dlaf_mi355x {

template <class T>
inline T* tm_dev_alloc(size_t elems) {
  T* p = nullptr;
  if (elems == 0)
    elems = 1;
  if (hipMalloc(reinterpret_cast<void**>(&p), elems * sizeof(T)) != hipSuccess) {
    (void) hipGetLastError();
    pool_release();  // (the workspace pool of the eigensolver stages may be holding what this allocation needs)
    DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), elems * sizeof(T)));
  }
  return p;
}

// General block-cyclic matrix in device tile layout (nb x nb tiles, tile (il,jl) at (il + jl*ltr) nb^2).
template <class T>
struct TileMatrix {
  Grid* grid = nullptr;
  bool transposed = false;  // the view is the (conjugate-)transpose of the caller's matrix
  Axis rows, cols;          // axes of the VIEW
  int nb = 1;
  long ltr = 0, ltc = 0;
  size_t tile_elems = 0;
  T* tiles = nullptr;
  T* staging = nullptr;
  bool owns = true;  // false: `tiles` belongs to somebody else (a view over a resident matrix)
  // grid dimension the view's rows are spread over: processes, my coordinate, global extent.  They equal
  // rows.P / rows.rank / rows.n unless the row axis has been localized (create_rhs).
  int row_P = 1, row_rank = 0;
  long rows_global = 0;
  // upload / download take the view's rows / columns from the far end (one process, whole tiles only: the ragged
  // tile must stay the last one)
  bool rev_rows = false, rev_cols = false;

  // m_src x n_src: global size of the caller's matrix, (isrc, jsrc) its source process
  void create(Grid* g, bool transposed_, long m_src, long n_src, int nb_, int isrc, int jsrc, T* borrow = nullptr) {
    grid = g;
    transposed = transposed_;
    nb = nb_;
    Axis srow{m_src, nb, g->nprow, g->myrow, isrc};
    Axis scol{n_src, nb, g->npcol, g->mycol, jsrc};
    rows = transposed ? scol : srow;
    cols = transposed ? srow : scol;
    ltr = rows.local_tiles();
    ltc = cols.local_tiles();
    tile_elems = (size_t) nb * nb;
    owns = borrow == nullptr;
    tiles = owns ? tm_dev_alloc<T>((size_t) ltr * ltc * tile_elems) : borrow;
    row_P = rows.P;
    row_rank = rows.rank;
    rows_global = rows.n;
  }
  // Right-hand sides of the triangular solver (view: rows = the free dimension, columns = the triangular one) with
  // the reference's MB x NB blocks (solver/triangular.h:41-60: B's block along the triangular dimension is A's, the
  // other one is free).  The rows of X T^H = B never interact, so WHICH rows a process holds is irrelevant to the
  // sweep: the local rows -- whatever block size and source process spread them -- are re-cut into nb-row tiles of
  // a one-process axis, and the device tiles stay square.
  // m_src x n_src: the caller's matrix, mb_src x nb_src its blocks; nb_ = the block along the triangular dimension.
  void create_rhs(Grid* g, bool transposed_, long m_src, long n_src, int mb_src, int nb_src, int isrc, int jsrc) {
    grid = g;
    transposed = transposed_;
    Axis srow{m_src, mb_src, g->nprow, g->myrow, isrc};
    Axis scol{n_src, nb_src, g->npcol, g->mycol, jsrc};
    const Axis free_axis = transposed ? scol : srow;
    cols = transposed ? srow : scol;
    nb = cols.nb;
    row_P = free_axis.P;
    row_rank = free_axis.rank;
    rows_global = free_axis.n;
    rows = Axis{free_axis.local_size(), nb, 1, 0, 0};
    ltr = rows.local_tiles();
    ltc = cols.local_tiles();
    tile_elems = (size_t) nb * nb;
    owns = true;
    tiles = tm_dev_alloc<T>((size_t) ltr * ltc * tile_elems);
  }
  ~TileMatrix() {
    if (tiles && owns)
      (void) hipFree(tiles);
    if (staging)
      (void) hipFree(staging);
  }
  T* tile(long il, long jl) const { return tiles + (size_t) (il + jl * ltr) * tile_elems; }
  // physical grid dimension (0: process rows, 1: process columns) the view's rows / columns are spread over
  int row_dim() const { return transposed ? 1 : 0; }
  int col_dim() const { return transposed ? 0 : 1; }

  LayoutArgs<T> layout(T* cm, long ld) const {
    LayoutArgs<T> a;
    a.tiles = tiles;
    a.cm = cm;
    a.ld_cm = ld;
    a.ltr = (int) ltr;
    a.ltc = (int) ltc;
    a.nb = nb;
    a.rows = rows.local_size();
    a.cols = cols.local_size();
    a.pr = rows.P;
    a.ri = rows.shift();
    a.pc = cols.P;
    a.ci = cols.shift();
    a.transpose = transposed ? 1 : 0;
    a.full = 1;
    a.rev_rows = rev_rows ? 1 : 0;
    a.rev_cols = rev_cols ? 1 : 0;
    return a;
  }
  void source_extents(long& srows, long& scols) const {
    srows = transposed ? cols.local_size() : rows.local_size();
    scols = transposed ? rows.local_size() : cols.local_size();
  }
  void upload(const T* host, long ld, bool conj, bool scale, T alpha, hipStream_t s) {
    long srows, scols;
    source_extents(srows, scols);
    if (srows == 0 || scols == 0)
      return;
    if (!staging)
      staging = tm_dev_alloc<T>((size_t) srows * scols);
    DLAF_HIP_CHECK(hipMemcpy2DAsync(staging, (size_t) srows * sizeof(T), host, (size_t) ld * sizeof(T),
                                    (size_t) srows * sizeof(T), (size_t) scols, hipMemcpyHostToDevice, s));
    LayoutArgs<T> a = layout(staging, srows);
    a.conj = conj ? 1 : 0;
    a.scale = scale ? 1 : 0;
    a.alpha = alpha;
    launch_to_tiles(a, s);
  }
  void download(T* host, long ld, bool conj, hipStream_t s) {
    long srows, scols;
    source_extents(srows, scols);
    if (srows == 0 || scols == 0)
      return;
    if (!staging)
      staging = tm_dev_alloc<T>((size_t) srows * scols);
    LayoutArgs<T> a = layout(staging, srows);
    a.conj = conj ? 1 : 0;
    launch_from_tiles(a, s);
    DLAF_HIP_CHECK(hipMemcpy2DAsync(host, (size_t) ld * sizeof(T), staging, (size_t) srows * sizeof(T),
                                    (size_t) srows * sizeof(T), (size_t) scols, hipMemcpyDeviceToHost, s));
  }
};


// the C-ABI handle of a resident general matrix (dlaf_mi355x_gmatrix_t)
template <class T>
struct GeneralMatrix : MatrixBase {
  TileMatrix<T> m;
  long rows_g = 0, cols_g = 0;
  int isrc = 0, jsrc = 0;
};

}  // namespace dlaf_mi355x
