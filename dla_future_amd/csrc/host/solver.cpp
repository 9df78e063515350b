// solver.cpp -- distributed triangular solve  op(A) X = alpha B  /  X op(A) = alpha B  on the tile kernels
// of the Cholesky path.
//
// Reference: dlaf::triangular_solver (include/dlaf/solver/triangular.h:41-177) and its eight hand-written
// variants call_{L,R}{L,U}{N,T} (solver/triangular/impl.h), each with its own tile loop and communication
// pattern (broadcasts for the NoTrans variants, reductions for the transposed ones).
//
// MI355X design: ONE device algorithm, the one the Cholesky kernels already implement,
//
//        X * T^H = B        T triangular (lower: swept forward, upper: swept backward)
//
//   step k:  X(:,k)  = B(:,k) * T_kk^-H                  panel TRSM kernel (inverted diagonal blocks)
//            B(:,j) -= X(:,k) * T(j,k)^H   for j beyond k   grouped NT update kernel, rectangular mode
//
// and every side / uplo / op / diag / alpha combination is mapped onto it when the operands are laid out
// on the device (the relayout kernel transposes, conjugates and scales on the way):
//
//   Right, op = C :  X A^H = aB            T = A            B_dev = a B
//   Right, op = N :  X A   = aB            T = A^H          B_dev = a B
//   Right, op = T :  X A^T = aB            T = conj(A)      B_dev = a B
//   Left,  op = N :  A X   = aB  <=>  X^H A^H   = (aB)^H    T = A            B_dev = (a B)^H
//   Left,  op = C :  A^H X = aB  <=>  X^H A     = (aB)^H    T = A^H          B_dev = (a B)^H
//   Left,  op = T :  A^T X = aB  <=>  X^H conj(A) = (aB)^H  T = A^T          B_dev = (a B)^H
//
// A transposed view of a block-cyclic matrix needs no communication: tile (i,j) of the view is tile (j,i)
// of the caller's matrix and stays on the same process, with the roles of process rows and columns swapped
// (the same device the Cholesky uses for uplo = U).  Two communication shapes remain, depending on whether
// T's rows are spread over the same grid dimension as B_dev's columns ("aligned": one broadcast of T's
// column panel) or over the other one ("crossed": the Cholesky's panel + transposed-panel pair).  T's
// panels depend on A only, so they are broadcast one step ahead of the sweep.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "runtime.hpp"
#include "tile_matrix.hpp"

namespace dlaf_mi355x {

namespace {

template <class T>
T* dev_alloc(size_t elems) {
  T* p = nullptr;
  if (elems == 0)
    elems = 1;
  DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), elems * sizeof(T)));
  return p;
}

template <class T>
T conj_el(T v) {
  if constexpr (TypeInfo<T>::is_complex)
    v.im = -v.im;
  return v;
}
template <class T>
double re_of_host(const T& v) {
  if constexpr (TypeInfo<T>::is_complex)
    return (double) v.re;
  else
    return (double) v;
}
template <class T>
double im_of_host(const T& v) {
  if constexpr (TypeInfo<T>::is_complex)
    return (double) v.im;
  else
    return 0.0;
}

// device time of the last sweep on this process (HIP events on the compute stream; relayout and PCIe excluded)
static double g_last_sweep_ms = 0;
static double g_last_sweep_flops = 0;

struct Events {
  std::vector<hipEvent_t> v;
  explicit Events(size_t n) : v(n) {
    for (auto& e : v)
      DLAF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ~Events() {
    for (auto e : v)
      (void) hipEventDestroy(e);
  }
  hipEvent_t operator[](size_t i) const { return v[i]; }
};

// X T^H = B in place on Bd; Td lower (swept forward) or upper (swept backward) triangular, unit: its
// diagonal is taken as 1.  Both operands on the same grid; Td's index distribution along the grid dimension
// it shares with Bd's columns must be the one of Bd's columns (checked by the caller).
template <class T>
void solve_canonical(TileMatrix<T>& Td, TileMatrix<T>& Bd, bool upper, bool unit) {
  Grid* g = Bd.grid;
  Transport* tr = grid_transport(*g);
  const bool dist = g->nranks > 1;
  const int nb = Bd.nb;
  const long nt = Bd.cols.nt();  // tiles along n
  if (nt == 0 || Bd.rows_global == 0)
    return;
  const size_t tile_elems = Bd.tile_elems, tile_bytes = tile_elems * sizeof(T);
  const size_t winv_elems = (size_t) ((nb + kDiagBlock - 1) / kDiagBlock) * kDiagBlock * kDiagBlock;
  const size_t diag_elems = tile_elems + winv_elems;

  // communicators seen from Bd's view: along its rows (members differ in view-column coordinate) and
  // along its columns
  const CommAxis along_row = Bd.transposed ? CommAxis::Col : CommAxis::Row;
  const CommAxis along_col = Bd.transposed ? CommAxis::Row : CommAxis::Col;
  // aligned: Td's rows live on the grid dimension of Bd's columns; crossed: on the one of Bd's rows
  const bool aligned = Td.row_dim() == Bd.col_dim();
  const Axis& t_match = aligned ? Td.rows : Td.cols;  // Td axis that shares Bd.cols' dimension
  const Axis& t_other = aligned ? Td.cols : Td.rows;  // ... and the one that shares Bd.rows' dimension
  if (t_match.P != Bd.cols.P || t_match.src != Bd.cols.src || t_match.n != Bd.cols.n)
    fatal("[dlaf_mi355x] triangular solver: A and B are not aligned along the triangular dimension "
          "(source process %d vs %d)\n", t_match.src, Bd.cols.src);

  hipStream_t s_main = nullptr, s_comm = nullptr;
  int lo = 0, hi = 0;
  DLAF_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  DLAF_HIP_CHECK(hipStreamCreateWithPriority(&s_main, hipStreamNonBlocking, lo));
  DLAF_HIP_CHECK(hipStreamCreateWithPriority(&s_comm, hipStreamNonBlocking, hi));
  Events ev_t((size_t) nt), ev_x((size_t) nt), ev_xb((size_t) nt), ev_free((size_t) nt);

  int* info = nullptr;
  DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&info), sizeof(int)));
  DLAF_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), s_main));

  // ---- preparation: inverted 64 x 64 diagonal blocks of every local diagonal tile of Td ---------------
  std::vector<long> my_diag;  // global indices of the diagonal tiles I own
  for (long k = 0; k < nt; ++k)
    if (Td.rows.mine(k) && Td.cols.mine(k))
      my_diag.push_back(k);
  T* winv_all = dev_alloc<T>(my_diag.size() * winv_elems);
  DLAF_HIP_CHECK(hipMemsetAsync(winv_all, 0, std::max<size_t>(1, my_diag.size() * winv_elems) * sizeof(T), s_main));
  for (size_t q = 0; q < my_diag.size(); ++q) {
    const long k = my_diag[q];
    launch_invert_diag_blocks(Td.tile(Td.rows.local_of(k), Td.cols.local_of(k)), nb, Td.rows.tile_extent(k),
                              winv_all + q * winv_elems, info, s_main, upper, unit);
  }
  hipEvent_t ev_prep;
  DLAF_HIP_CHECK(hipEventCreateWithFlags(&ev_prep, hipEventDisableTiming));
  DLAF_HIP_CHECK(hipEventRecord(ev_prep, s_main));
  DLAF_HIP_CHECK(hipStreamWaitEvent(s_comm, ev_prep, 0));

  // workspaces, kBuf of each (step s uses s % kBuf; the T operands of step s+1 are fetched while step s-1 is
  // still being applied): [T_kk | W_k], the T column panel as the update's second operand (one tile per
  // local column of Bd), its staging for the crossed shape, the X panel on non-owners
  constexpr int kBuf = 3;
  T* diag_ws[kBuf] = {nullptr, nullptr, nullptr};
  T* tpanel[kBuf] = {nullptr, nullptr, nullptr};
  T* tstage[kBuf] = {nullptr, nullptr, nullptr};
  T* xpanel[kBuf] = {nullptr, nullptr, nullptr};
  for (int b = 0; b < kBuf; ++b) {
    diag_ws[b] = dev_alloc<T>(diag_elems);
    if (dist) {
      tpanel[b] = dev_alloc<T>((size_t) Bd.ltc * tile_elems);
      if (!aligned)
        tstage[b] = dev_alloc<T>((size_t) Td.ltr * tile_elems);
      xpanel[b] = dev_alloc<T>((size_t) Bd.ltr * tile_elems);
    }
  }

  // global step order and the set of columns "beyond" step k
  auto step_k = [&](long s) { return upper ? nt - 1 - s : s; };

  struct TOperand {
    const T* diag = nullptr;  // T_kk
    const T* winv = nullptr;  // its inverted diagonal blocks
    const T* base = nullptr;  // T(j,k) for local column jl of Bd at base + (jl - jl0) * ts
    long ts = 0;
    long jl0 = 0, jl1 = 0;    // local columns of Bd beyond step k
  };
  std::vector<TOperand> top((size_t) nt);

  // ---- s_comm: everything the step needs of T (depends on A only: issued ahead of the sweep) ----------
  auto fetch_t = [&](long s) {
    const long k = step_k(s);
    const int buf = (int) (s % kBuf);
    TOperand& o = top[(size_t) s];
    // local columns of Bd beyond k
    o.jl0 = upper ? 0 : Bd.cols.next_local(k + 1);
    o.jl1 = upper ? Bd.cols.next_local(k) : Bd.ltc;
    // these buffers were last read by the kernels of step s - kBuf (event recorded before this call is made)
    if (s >= kBuf)
      DLAF_HIP_CHECK(hipStreamWaitEvent(s_comm, ev_free[(size_t) (s - kBuf)], 0));

    // (1) T_kk and its inverted diagonal blocks to every process holding column k of Bd
    const bool own_diag = Td.rows.mine(k) && Td.cols.mine(k);
    const bool need_diag = Bd.cols.mine(k);
    const T* tkk = nullptr;
    const T* wk = nullptr;
    if (own_diag) {
      const size_t q = (size_t) (std::find(my_diag.begin(), my_diag.end(), k) - my_diag.begin());
      tkk = Td.tile(Td.rows.local_of(k), Td.cols.local_of(k));
      wk = winv_all + q * winv_elems;
    }
    if (need_diag && Bd.row_P > 1) {
      // (own_diag implies need_diag: the owner sits in Bd's column k by the alignment requirement)
      if (own_diag) {
        DLAF_HIP_CHECK(hipMemcpyAsync(diag_ws[buf], tkk, tile_bytes, hipMemcpyDeviceToDevice, s_comm));
        DLAF_HIP_CHECK(hipMemcpyAsync(diag_ws[buf] + tile_elems, wk, winv_elems * sizeof(T), hipMemcpyDeviceToDevice, s_comm));
      }
      tr->bcast(along_col, t_other.owner(k), Bd.row_rank, diag_ws[buf], diag_ws[buf], diag_elems * sizeof(T), s_comm);
      tkk = diag_ws[buf];
      wk = diag_ws[buf] + tile_elems;
    }
    o.diag = tkk;
    o.winv = wk;

    // (2) T(j,k) for the local columns j of Bd beyond k
    const long ncols = o.jl1 - o.jl0;
    if (!dist) {
      // one process: Td's local row index of global j is Bd's local column index
      o.base = Td.tile(o.jl0 < Td.ltr ? o.jl0 : 0, k);
      o.ts = (long) tile_elems;
    }
    else if (aligned) {
      // Td's rows are spread like Bd's columns: the tiles sit on the process of the same Bd-column
      // coordinate whose Bd-row coordinate owns Td's column k -> one broadcast along Bd's columns
      const bool have = t_other.mine(k);
      T* dst = tpanel[buf];
      if (ncols > 0) {
        const T* src = have ? Td.tile(o.jl0, Td.cols.local_of(k)) : nullptr;
        if (Bd.row_P > 1)
          tr->bcast(along_col, t_other.owner(k), Bd.row_rank, src, dst, (size_t) ncols * tile_bytes, s_comm);
        else
          dst = const_cast<T*>(src);
      }
      o.base = dst;
      o.ts = (long) tile_elems;
    }
    else {
      // crossed: Td's rows are spread like Bd's ROWS.  Column panel k of Td along Bd's rows first, then
      // tile j down Bd's columns from the Bd-row coordinate that owns Td's row j (broadcast_panel.h:125-210)
      const long il0 = upper ? 0 : Td.rows.next_local(k + 1);
      const long il1 = upper ? Td.rows.next_local(k) : Td.ltr;
      const bool have = Td.cols.mine(k);
      const T* colp = nullptr;  // my rows [il0, il1) of Td's column k
      if (il1 > il0) {
        if (Bd.cols.P > 1) {
          const T* src = have ? Td.tile(il0, Td.cols.local_of(k)) : nullptr;
          tr->bcast(along_row, Td.cols.owner(k), Bd.cols.rank, src, tstage[buf], (size_t) (il1 - il0) * tile_bytes, s_comm);
          colp = tstage[buf];
        }
        else {
          colp = Td.tile(il0, Td.cols.local_of(k));
        }
      }
      if (Bd.row_P > 1) {
        tr->group_begin();
        for (long jl = o.jl0; jl < o.jl1; ++jl) {
          const long gj = Bd.cols.global_of(jl);
          const int root = Td.rows.owner(gj);
          const T* src = (Td.rows.rank == root) ? colp + (size_t) (Td.rows.local_of(gj) - il0) * tile_elems : nullptr;
          tr->bcast(along_col, root, Bd.row_rank, src, tpanel[buf] + (size_t) (jl - o.jl0) * tile_elems, tile_bytes, s_comm);
        }
        tr->group_end();
        o.base = tpanel[buf];
        o.ts = (long) tile_elems;
      }
      else {
        // I hold every row of Td's column k: tile gj sits at local row gj
        o.base = colp ? colp + (size_t) (Bd.cols.global_of(o.jl0 < Bd.ltc ? o.jl0 : 0) - il0) * tile_elems : nullptr;
        o.ts = (long) tile_elems * Bd.cols.P;
      }
    }
    DLAF_HIP_CHECK(hipEventRecord(ev_t[(size_t) s], s_comm));
  };

  auto update = [&](long s, const T* xp, long j0, long j1) {
    const TOperand& o = top[(size_t) s];
    j0 = std::max(j0, o.jl0);
    j1 = std::min(j1, o.jl1);
    if (j0 >= j1 || Bd.ltr == 0)
      return;
    const long k = step_k(s);
    UpdateArgs<T> ua;
    ua.c = Bd.tiles;
    ua.c_tsr = (long) tile_elems;
    ua.c_tsc = (long) (tile_elems * Bd.ltr);
    ua.ldc = nb;
    ua.a = xp;
    ua.a_ts = (long) tile_elems;
    ua.lda = nb;
    ua.b = o.base + (j0 - o.jl0) * o.ts;
    ua.b_ts = o.ts;
    ua.ldb = nb;
    ua.il0 = 0;
    ua.il1 = (int) Bd.ltr;
    ua.jl0 = (int) j0;
    ua.jl1 = (int) j1;
    ua.nb = nb;
    ua.K = Bd.cols.tile_extent(k);
    ua.pr = Bd.rows.P;
    ua.ri = Bd.rows.shift();
    ua.pc = Bd.cols.P;
    ua.ci = Bd.cols.shift();
    ua.nt = (int) Bd.rows.nt();
    ua.last_rows = Bd.rows.last_extent();
    ua.rect = 1;
    ua.nt_c = (int) nt;
    ua.last_cols = Bd.cols.last_extent();
    ua.info = info;
    launch_update(ua, s_main, 3);
  };

  hipEvent_t ev_t0, ev_t1;
  DLAF_HIP_CHECK(hipEventCreate(&ev_t0));
  DLAF_HIP_CHECK(hipEventCreate(&ev_t1));
  DLAF_HIP_CHECK(hipEventRecord(ev_t0, s_main));

  // ---- the sweep --------------------------------------------------------------------------------------
  // s_main: TRSM(s) . U(s, next column) . TRSM(s+1) . U(s, the rest) . U(s+1, next column) ...  so that the
  // X panel of step s+1 is on the wire under the bulk of step s; T operands arrive one step ahead on s_comm
  fetch_t(0);
  const T* xp_prev = nullptr;
  for (long s = 0; s < nt; ++s) {
    const long k = step_k(s);
    const int buf = (int) (s % kBuf);
    if (s + 1 < nt)
      fetch_t(s + 1);
    DLAF_HIP_CHECK(hipStreamWaitEvent(s_main, ev_t[(size_t) s], 0));

    // column k of Bd: X(:,k) = B(:,k) T_kk^-H
    const bool in_col = Bd.cols.mine(k);
    const long klc = in_col ? Bd.cols.local_of(k) : -1;
    if (in_col && Bd.ltr > 0) {
      TrsmArgs<T> ta;
      ta.b = Bd.tile(0, klc);
      ta.b_ts = (long) tile_elems;
      ta.ldb = nb;
      ta.il0 = 0;
      ta.il1 = (int) Bd.ltr;
      ta.pr = Bd.rows.P;
      ta.ri = Bd.rows.shift();
      ta.nb = nb;
      ta.nt = (int) Bd.rows.nt();
      ta.last_rows = Bd.rows.last_extent();
      ta.l = top[(size_t) s].diag;
      ta.ldl = nb;
      ta.winv = top[(size_t) s].winv;
      ta.n = Bd.cols.tile_extent(k);
      ta.info = info;
      ta.upper = upper ? 1 : 0;
      launch_trsm(ta, s_main);
    }
    DLAF_HIP_CHECK(hipEventRecord(ev_x[(size_t) s], s_main));

    // the solved panel to the other members of my Bd row
    const T* xp = in_col ? Bd.tile(0, klc) : nullptr;
    if (Bd.cols.P > 1) {
      DLAF_HIP_CHECK(hipStreamWaitEvent(s_comm, ev_x[(size_t) s], 0));
      T* dst = in_col ? Bd.tile(0, klc) : xpanel[buf];
      if (Bd.ltr > 0)
        tr->bcast(along_row, Bd.cols.owner(k), Bd.cols.rank, dst, dst, (size_t) Bd.ltr * tile_bytes, s_comm);
      xp = dst;
      DLAF_HIP_CHECK(hipEventRecord(ev_xb[(size_t) s], s_comm));
    }

    // what is left of the previous step's update runs under that broadcast
    if (s >= 1) {
      const long kn = k;  // the "next column" of step s-1 is this step's column
      const long jn = Bd.cols.mine(kn) ? Bd.cols.local_of(kn) : -1;
      if (jn >= 0) {
        update(s - 1, xp_prev, 0, jn);
        update(s - 1, xp_prev, jn + 1, Bd.ltc);
      }
      else {
        update(s - 1, xp_prev, 0, Bd.ltc);
      }
      DLAF_HIP_CHECK(hipEventRecord(ev_free[(size_t) (s - 1)], s_main));
    }
    if (Bd.cols.P > 1)
      DLAF_HIP_CHECK(hipStreamWaitEvent(s_main, ev_xb[(size_t) s], 0));

    // lookahead: the column the next step solves
    if (s + 1 < nt) {
      const long kn = step_k(s + 1);
      if (Bd.cols.mine(kn)) {
        const long jn = Bd.cols.local_of(kn);
        update(s, xp, jn, jn + 1);
      }
    }
    xp_prev = xp;
  }
  // the last step has nothing beyond it; its event only releases the buffers
  DLAF_HIP_CHECK(hipEventRecord(ev_free[(size_t) (nt - 1)], s_main));

  DLAF_HIP_CHECK(hipEventRecord(ev_t1, s_main));
  DLAF_HIP_CHECK(hipStreamSynchronize(s_comm));
  DLAF_HIP_CHECK(hipStreamSynchronize(s_main));
  {
    float ms = 0;
    DLAF_HIP_CHECK(hipEventElapsedTime(&ms, ev_t0, ev_t1));
    g_last_sweep_ms = ms;
    // whole-grid algorithmic flops: rows x n^2 (x4 complex)
    g_last_sweep_flops = (TypeInfo<T>::is_complex ? 4.0 : 1.0) * (double) Bd.rows_global * (double) Bd.cols.n * (double) Bd.cols.n;
  }
  (void) hipEventDestroy(ev_t0);
  (void) hipEventDestroy(ev_t1);
  (void) hipEventDestroy(ev_prep);
  (void) hipStreamDestroy(s_main);
  (void) hipStreamDestroy(s_comm);
  (void) hipFree(info);
  (void) hipFree(winv_all);
  for (int b = 0; b < kBuf; ++b) {
    (void) hipFree(diag_ws[b]);
    if (tpanel[b])
      (void) hipFree(tpanel[b]);
    if (tstage[b])
      (void) hipFree(tstage[b]);
    if (xpanel[b])
      (void) hipFree(xpanel[b]);
  }
}

}  // namespace

void solver_last_profile(double* ms, double* flops) {
  if (ms)
    *ms = g_last_sweep_ms;
  if (flops)
    *flops = g_last_sweep_flops;
}

// Host entry: a (local part of the triangular matrix, column-major lda), b (local part of the m x n right-hand
// sides, ldb) on the grid; b is overwritten by the solution.
template <class T>
int triangular_solver_host(Grid* g, char side, char uplo, char op, char diag, T alpha, const T* a, long lda, int a_isrc,
                           int a_jsrc, T* b, long ldb, long m, long n, int nb, int b_isrc, int b_jsrc, int nb_free) {
  runtime_init();
  (void) grid_transport(*g);
  if (g->nranks > 1 && !g->transport)
    fatal("[dlaf_mi355x] grid with %d ranks has no transport\n", g->nranks);
  if (m == 0 || n == 0)
    return 0;
  const bool left = (side == 'L' || side == 'l');
  const bool a_upper = (uplo == 'U' || uplo == 'u');
  const char o = (op == 'n') ? 'N' : (op == 't') ? 'T' : (op == 'c') ? 'C' : op;
  const bool unit = (diag == 'U' || diag == 'u');
  const long na = left ? m : n;
  if (left ? (a_isrc != b_isrc) : (a_jsrc != b_jsrc))
    fatal("[dlaf_mi355x] triangular solver: A and B must share the source process along the triangular "
          "dimension\n");

  // T = A (Right C / Left N), A^H (Right N / Left C), conj(A) (Right T), A^T (Left T)
  bool t_transposed, t_conj;
  if (left) {
    t_transposed = (o != 'N');
    t_conj = (o == 'C');
  }
  else {
    t_transposed = (o == 'N');
    t_conj = (o == 'N' || o == 'T');
  }
  const bool t_upper = a_upper != t_transposed;

  hipStream_t s = nullptr;
  DLAF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  {
    TileMatrix<T> Td, Bd;
    Td.create(g, t_transposed, na, na, nb, a_isrc, a_jsrc);
    // B's blocks: nb along the triangular dimension (its rows for side = Left), nb_free along the other
    if (nb_free <= 0)
      nb_free = nb;
    Bd.create_rhs(g, left, m, n, left ? nb : nb_free, left ? nb_free : nb, b_isrc, b_jsrc);
    // One process, whole tiles: an upper triangular T (swept backward by the round-1 strips kernel) is laid out
    // REVERSED -- T'(i, j) = T(na-1-i, na-1-j) is lower triangular, X'(:, j) = X(:, na-1-j) solves X' T'^H = B' -- and
    // the forward sweep of the row-owner kernel does the work; the download reverses back.
    // DLAF_MI355X_SOLVER_REVERSE=0: the backward sweep (A/B).
    static const bool reverse_on = [] {
      const char* e = std::getenv("DLAF_MI355X_SOLVER_REVERSE");
      return e ? std::atoi(e) != 0 : true;
    }();
    const bool reversed = reverse_on && t_upper && g->nranks == 1 && na % nb == 0;
    if (reversed) {
      Td.rev_rows = Td.rev_cols = true;
      Bd.rev_cols = true;  // the view's columns are the triangular dimension
    }
    Td.upload(a, lda, t_conj, false, T{}, s);
    // Left: B_dev = (alpha B)^H = conj(alpha) B^H (the relayout conjugates first, then scales)
    Bd.upload(b, ldb, left, true, left ? conj_el(alpha) : alpha, s);
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    solve_canonical(Td, Bd, reversed ? false : t_upper, unit);
    Bd.download(b, ldb, left, s);
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
  }
  DLAF_HIP_CHECK(hipStreamDestroy(s));
  return 0;
}

// ================================================================================ device-resident operands
// A general m x n matrix resident in HBM in tile layout (the right-hand sides of a solve), behind an opaque handle.

MatrixBase* general_matrix_create(Grid* g, char type, long m, long n, int nb, int isrc, int jsrc) {
  runtime_init();
  (void) grid_transport(*g);
  auto make = [&](auto* tag) -> MatrixBase* {
    using T = std::remove_pointer_t<decltype(tag)>;
    auto* gm = new GeneralMatrix<T>;
    gm->type = type;
    gm->rows_g = m;
    gm->cols_g = n;
    gm->isrc = isrc;
    gm->jsrc = jsrc;
    gm->m.create(g, false, m, n, nb, isrc, jsrc);
    return gm;
  };
  switch (type) {
    case 's': return make((float*) nullptr);
    case 'd': return make((double*) nullptr);
    case 'c': return make((cfloat*) nullptr);
    case 'z': return make((cdouble*) nullptr);
    default: return nullptr;
  }
}

template <class T>
static void gm_transfer(GeneralMatrix<T>& gm, void* host, long ld, bool up) {
  hipStream_t s = nullptr;
  DLAF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (up)
    gm.m.upload(static_cast<const T*>(host), ld, false, false, T{}, s);
  else
    gm.m.download(static_cast<T*>(host), ld, false, s);
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  DLAF_HIP_CHECK(hipStreamDestroy(s));
}

void general_matrix_transfer(MatrixBase* h, void* host, long ld, bool upload) {
  switch (h->type) {
    case 's': gm_transfer(static_cast<GeneralMatrix<float>&>(*h), host, ld, upload); break;
    case 'd': gm_transfer(static_cast<GeneralMatrix<double>&>(*h), host, ld, upload); break;
    case 'c': gm_transfer(static_cast<GeneralMatrix<cfloat>&>(*h), host, ld, upload); break;
    case 'z': gm_transfer(static_cast<GeneralMatrix<cdouble>&>(*h), host, ld, upload); break;
    default: fatal("[dlaf_mi355x] bad matrix type\n");
  }
}

// dst view tile (il, jl) = alpha * op(src tile): src (il, jl) for the copying ops (4 conjugate, 5 copy), src (jl, il)
// for the transposing ones (0 adjoint, 3 transpose).  A transposed view of a block-cyclic matrix stays on the
// same process (tile (i,j) of the view is tile (j,i) of the source), so this is local work on every rank.
template <class T>
static void xform_tiles(T* dst, long dltr, long dltc, const T* src, long sltr, size_t te, int nb, int mode, T alpha,
                        bool use_alpha, hipStream_t s) {
  const bool tr = (mode == 0 || mode == 3);
  for (long jl = 0; jl < dltc; ++jl) {
    // view column jl, all view rows: src tiles (il, jl) [stride te] or (jl, il) [stride sltr * te]
    const T* sp = tr ? src + (size_t) jl * te : src + (size_t) jl * sltr * te;
    launch_tile_xform_alpha(dst + (size_t) jl * dltr * te, (long) nb, (long) te, sp, (long) nb,
                            tr ? (long) (sltr * te) : (long) te, nb, nb, (int) dltr, mode, alpha, use_alpha, s);
  }
}

// dlaf::triangular_solver on RESIDENT operands: A = a DeviceMatrix (its uplo triangle: a Cholesky factor, or any
// triangular matrix uploaded as such), B = a general resident matrix, overwritten by the solution.  Nothing crosses
// PCIe: the operand views of the one device algorithm (X T^H = B) are made by tile transforms on the device.
template <class T>
static int solver_device(char side, char uplo, char op, char diag, T alpha, DeviceMatrix<T>& A, GeneralMatrix<T>& B) {
  Grid* g = A.grid;
  if (B.m.grid != g)
    fatal("[dlaf_mi355x] triangular solver: A and B live on different grids\n");
  const bool left = (side == 'L' || side == 'l');
  const bool a_upper = (uplo == 'U' || uplo == 'u');
  if (a_upper != A.transposed)
    fatal("[dlaf_mi355x] triangular solver: uplo '%c' but the resident matrix holds its '%c' triangle\n", uplo, A.uplo);
  const char o = (op == 'n') ? 'N' : (op == 't') ? 'T' : (op == 'c') ? 'C' : op;
  const bool unit = (diag == 'U' || diag == 'u');
  const long m = B.rows_g, n = B.cols_g, na = left ? m : n;
  const int nb = A.nb;
  if (A.n != na || B.m.nb != nb)
    fatal("[dlaf_mi355x] triangular solver: A is %ld x %ld (block %d), B is %ld x %ld (block %d), side %c\n", A.n, A.n,
          nb, m, n, B.m.nb, side);
  if (m == 0 || n == 0)
    return 0;
  // the caller's A and its source process (the DeviceMatrix holds the transposed view for uplo U)
  const Axis& a_rows = A.transposed ? A.cols : A.rows;
  const Axis& a_cols = A.transposed ? A.rows : A.cols;
  if (left ? (a_rows.src != B.isrc) : (a_cols.src != B.jsrc))
    fatal("[dlaf_mi355x] triangular solver: A and B must share the source process along the triangular dimension\n");
  // T = A (Right C / Left N), A^H (Right N / Left C), conj(A) (Right T), A^T (Left T) -- as in triangular_solver_host
  bool t_transposed, t_conj;
  if (left) {
    t_transposed = (o != 'N');
    t_conj = (o == 'C');
  }
  else {
    t_transposed = (o == 'N');
    t_conj = (o == 'N' || o == 'T');
  }
  const bool t_upper = a_upper != t_transposed;
  // in terms of the STORED tiles S (S = A for uplo L, S = A^T for uplo U): T = op(S)
  const bool s_transpose = t_transposed != A.transposed;
  const int t_mode = s_transpose ? (t_conj ? 0 : 3) : (t_conj ? 4 : 5);

  hipStream_t s = nullptr;
  DLAF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  {
    TileMatrix<T> Td, Bd;
    const size_t te = A.tile_elems;
    if (t_mode == 5) {
      Td.create(g, t_transposed, na, na, nb, a_rows.src, a_cols.src, A.tiles);  // T is the stored matrix itself
    }
    else {
      Td.create(g, t_transposed, na, na, nb, a_rows.src, a_cols.src);
      xform_tiles(Td.tiles, Td.ltr, Td.ltc, A.tiles, A.ltr, te, nb, t_mode, T{}, false, s);
    }
    const bool scale = !(re_of_host(alpha) == 1 && im_of_host(alpha) == 0);
    if (left) {
      // B_dev = (alpha B)^H = conj(alpha) B^H
      Bd.create(g, true, m, n, nb, B.isrc, B.jsrc);
      xform_tiles(Bd.tiles, Bd.ltr, Bd.ltc, B.m.tiles, B.m.ltr, te, nb, 0, conj_el(alpha), scale, s);
    }
    else {
      Bd.create(g, false, m, n, nb, B.isrc, B.jsrc, B.m.tiles);  // in place
      if (scale)
        xform_tiles(Bd.tiles, Bd.ltr, Bd.ltc, B.m.tiles, B.m.ltr, te, nb, 5, alpha, true, s);
    }
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    solve_canonical(Td, Bd, t_upper, unit);
    if (left) {
      xform_tiles(B.m.tiles, B.m.ltr, B.m.ltc, Bd.tiles, Bd.ltr, te, nb, 0, T{}, false, s);
      DLAF_HIP_CHECK(hipStreamSynchronize(s));
    }
  }
  DLAF_HIP_CHECK(hipStreamDestroy(s));
  return 0;
}

int triangular_solver_device(char side, char uplo, char op, char diag, const void* alpha, MatrixBase* a, MatrixBase* b) {
  if (!a || !b || a->type != b->type)
    fatal("[dlaf_mi355x] triangular solver: operands of different element types\n");
  switch (a->type) {
    case 's': return solver_device<float>(side, uplo, op, diag, *static_cast<const float*>(alpha),
                                          static_cast<DeviceMatrix<float>&>(*a), static_cast<GeneralMatrix<float>&>(*b));
    case 'd': return solver_device<double>(side, uplo, op, diag, *static_cast<const double*>(alpha),
                                           static_cast<DeviceMatrix<double>&>(*a), static_cast<GeneralMatrix<double>&>(*b));
    case 'c': return solver_device<cfloat>(side, uplo, op, diag, *static_cast<const cfloat*>(alpha),
                                           static_cast<DeviceMatrix<cfloat>&>(*a), static_cast<GeneralMatrix<cfloat>&>(*b));
    case 'z': return solver_device<cdouble>(side, uplo, op, diag, *static_cast<const cdouble*>(alpha),
                                            static_cast<DeviceMatrix<cdouble>&>(*a), static_cast<GeneralMatrix<cdouble>&>(*b));
    default: fatal("[dlaf_mi355x] bad matrix type\n");
  }
}

template int triangular_solver_host<float>(Grid*, char, char, char, char, float, const float*, long, int, int, float*,
                                           long, long, long, int, int, int, int);
template int triangular_solver_host<double>(Grid*, char, char, char, char, double, const double*, long, int, int,
                                            double*, long, long, long, int, int, int, int);
template int triangular_solver_host<cfloat>(Grid*, char, char, char, char, cfloat, const cfloat*, long, int, int,
                                            cfloat*, long, long, long, int, int, int, int);
template int triangular_solver_host<cdouble>(Grid*, char, char, char, char, cdouble, const cdouble*, long, int, int,
                                             cdouble*, long, long, long, int, int, int, int);

}  // namespace dlaf_mi355x
