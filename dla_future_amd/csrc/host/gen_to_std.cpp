// gen_to_std.cpp -- reduction of the Hermitian generalized eigenproblem to standard form,
//        A  <-  L^-1 A L^-H   (uplo = L)         A  <-  U^-H A U^-1   (uplo = U),
// with the Cholesky factor of B (LAPACK xHEGST itype 1).
//
// Reference: dlaf::eigensolver::internal::generalized_to_standard (include/dlaf/eigensolver/gen_to_std.h:50,
// :101), GenToStd::call_L local (eigensolver/gen_to_std/impl.h:222-283) and distributed (:286-...), tile ops
// hegst (lapack/tile.h:209-218), trsm, hemm, her2k, gemm (blas/tile.h).  Per step k the reference issues
//   (1) hegst of the diagonal tile, (2) panel trsm + hemm, (3) her2k / 2 gemm on every trailing tile,
//   (4) hemm on the panel again, (5) a left triangular solve of the panel with the trailing part of L,
// one tile task at a time.
//
// MI355X design -- the kernels of the Cholesky path, grouped launches, two phases:
//   Phase I, step k = (1)-(4):
//     * the diagonal tile goes through two passes of the panel-TRSM kernel on its full Hermitian image
//       (D <- D L^-H, then the same on D^H: L^-1 D L^-H = ((D L^-H)^H L^-H)^H);
//     * panel: TRSM kernel, then "hemm" as one rectangular update launch against 0.5 * D (full image);
//     * trailing matrix: ONE two-segment update launch per step,
//           C -= [A_ik | L_ik] [L_jk | A_jk]^H  =  A_ik L_jk^H + L_ik A_jk^H      (K = 2 nb),
//       the her2k of the diagonal tiles being the same product under the triangle mask;
//   Phase II = (5) for ALL columns at once.  Step (5) of column k reads only L and column k after its step
//     (4), and nothing reads column k afterwards, so it can be deferred; deferred, it is the block forward
//     substitution  L X = strictly-block-lower(A)  swept by tile ROWS:
//           R_j <- L_jj^-1 R_j (R_j = tile row j left of the diagonal),   A(i, :) -= L_ij R_j  for i > j,
//     i.e. per row j one panel TRSM on the transposed tiles of R_j and one rectangular update over all the
//     rows below -- the launch shapes of the Cholesky trailing update instead of a chain of tile solves.
//   uplo = U runs on the transposed view like the Cholesky (B^T = L'^-1 A^T L'^-H with L' = U^T).
// Communication per step (process grids): L_kk (+ inverse blocks, + 0.5 D) down the owning process column,
// the A and L column panels along process rows, their transposed panels down process columns (one broadcast
// per root row each); Phase II: L_jj along the process row, the solved row down process columns, the L
// column panel along process rows.  One process: the diagonal tile / panel chain of step k+1 (Phase II: the solve of
// row j+1) runs on a side stream beside the trailing update of step k, which takes the next column (row) first; on a
// grid everything is stream-ordered on one stream.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "runtime.hpp"

namespace dlaf_mi355x {

namespace {
template <class T>
T* dalloc(size_t elems) {
  T* p = nullptr;
  DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(elems, 1) * sizeof(T)));
  return p;
}
}  // namespace

template <class T>
int gen_to_std_device(DeviceMatrix<T>& A, DeviceMatrix<T>& L) {
  if (A.grid != L.grid || A.n != L.n || A.nb != L.nb || A.ltr != L.ltr || A.ltc != L.ltc || A.transposed != L.transposed)
    fatal("[dlaf_mi355x] gen_to_std: A and the Cholesky factor differ in shape, uplo or distribution\n");
  Grid* grid = A.grid;
  Transport* tr = grid_transport(*grid);
  const bool dist = grid->nranks > 1;
  if (dist && !tr)
    fatal("[dlaf_mi355x] grid with %d ranks has no transport\n", grid->nranks);
  const Axis& rows = A.rows;
  const Axis& cols = A.cols;
  const long nt = A.nt, ltr = A.ltr, ltc = A.ltc;
  const int nb = A.nb;
  const size_t te = A.tile_elems, wel = A.winv_elems();
  const CommAxis ax_row = A.transposed ? CommAxis::Col : CommAxis::Row;
  const CommAxis ax_col = A.transposed ? CommAxis::Row : CommAxis::Col;
  hipStream_t s = A.s_high;
  int* info = A.info;
  // work enqueued on the factor's streams (a cholesky_start without a wait) must be done before it is read here
  for (hipStream_t ls : {L.s_high, L.s_low, L.s_comm})
    if (ls != nullptr && ls != s)
      DLAF_HIP_CHECK(hipStreamSynchronize(ls));
  DLAF_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), s));
  if (nt == 0)
    return 0;

  // workspaces: [L_kk | inverse diagonal blocks | 0.5 D] travel together down the process column; two of them and
  // two row buffers because the panel work of step k+1 runs beside the trailing update of step k (one process)
  T* dws2[2] = {dalloc<T>(2 * te + wel), dalloc<T>(2 * te + wel)};
  T* dfull = dalloc<T>(te);
  T* xt = dalloc<T>(te);
  T* pA = dalloc<T>((size_t) ltr * te);
  T* pL = dalloc<T>((size_t) ltr * te);
  T* pAT = dalloc<T>((size_t) (ltc + rows.P) * te);
  T* pLT = dalloc<T>((size_t) (ltc + rows.P) * te);
  T* Tw2[2] = {dalloc<T>((size_t) std::max<long>(ltc, 1) * te), dalloc<T>((size_t) std::max<long>(ltc, 1) * te)};
  unsigned* counters = nullptr;
  DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&counters), 16 * sizeof(unsigned)));

  // One process: lookahead.  The chain diagonal tile -> panel of step k+1 runs on a side stream beside the trailing
  // update of step k, which takes column k+1 first and leaves `side_slots` workgroup slots free afterwards (the
  // Cholesky's "sidecar" order, runtime.cpp).  On a grid everything stays on one stream (the single panel workspaces
  // are reused step by step).  Measured on MI355X (tools/run_hegst.sh, profiles/r03_gen_to_std_lookahead_ab.txt): NO
  // gain -- fp64 N=16384 nb=512 44.8 (lookahead) vs 46.9 TFlop/s (one stream), N=32768 nb=1024 60.5 vs 61.3, z
  // N=16384 nb=512 53.4 vs 52.7: on 32 slots the chain (two single-tile solves, panel solve, hemm) takes about as
  // long as the trailing update it runs beside, and the update loses the slots.  The stage is bound by the nb = 512
  // update rate and by 126 latency-bound solves of ~86 us per run (rocprofv3: update 84 %, TRSM 11 %).  So: opt-in,
  // DLAF_MI355X_HEGST_LOOKAHEAD=1.
  const bool lookahead = !dist && [] {
    const char* e = std::getenv("DLAF_MI355X_HEGST_LOOKAHEAD");
    return e ? std::atoi(e) != 0 : false;
  }();
  hipStream_t sp = s;
  if (lookahead) {
    int lo = 0, hi = 0;
    DLAF_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    DLAF_HIP_CHECK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi));
  }
  const long side_slots = lookahead ? 32 : 0;
  std::vector<hipEvent_t> ev_panel((size_t) nt + 1), ev_la((size_t) nt + 1);
  for (auto* v : {&ev_panel, &ev_la})
    for (auto& e : *v)
      DLAF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto after = [&](hipStream_t waiter, hipEvent_t ev, hipStream_t recorder) {
    DLAF_HIP_CHECK(hipEventRecord(ev, recorder));
    if (waiter != recorder)
      DLAF_HIP_CHECK(hipStreamWaitEvent(waiter, ev, 0));
  };

  auto trsm_tiles = [&](T* b, long ntiles, int rows_each, const T* l, const T* w, int n, hipStream_t st) {
    // X = B l^-H for `ntiles` tiles of rows_each x n at b, b + te, ...
    if (ntiles <= 0)
      return;
    TrsmArgs<T> ta;
    ta.b = b;
    ta.b_ts = (long) te;
    ta.ldb = nb;
    ta.il0 = 0;
    ta.il1 = (int) ntiles;
    ta.pr = 1;
    ta.ri = 0;
    ta.nb = rows_each;
    ta.nt = (int) ntiles + 1;  // no tile of this batch is "the last global tile"
    ta.last_rows = rows_each;
    ta.l = l;
    ta.ldl = nb;
    ta.winv = w;
    ta.n = n;
    ta.info = info;
    ta.prio = (st != s) ? 1 : 0;
    launch_trsm(ta, st);
  };
  // C(il, jl) -= a(il) b(jl)^H over local tile rows [il0, il1) x local tile columns [jl0, jl1), every tile;
  // reserve > 0: persistent form that leaves that many workgroup slots to the side stream
  auto rect_update = [&](long il0, long il1, long jl0, long jl1, const T* a, const T* b, long b_ts, int K, hipStream_t st,
                         long reserve = 0) {
    if (il0 >= il1 || jl0 >= jl1 || K <= 0)
      return;
    UpdateArgs<T> ua;
    ua.c = A.tiles;
    ua.c_tsr = (long) te;
    ua.c_tsc = (long) (te * ltr);
    ua.ldc = nb;
    ua.a = a;
    ua.a_ts = (long) te;
    ua.lda = nb;
    ua.b = b;
    ua.b_ts = b_ts;
    ua.ldb = nb;
    ua.il0 = (int) il0;
    ua.il1 = (int) il1;
    ua.jl0 = (int) jl0;
    ua.jl1 = (int) jl1;
    ua.nb = nb;
    ua.K = K;
    ua.pr = rows.P;
    ua.ri = rows.shift();
    ua.pc = cols.P;
    ua.ci = cols.shift();
    ua.nt = (int) nt;
    ua.last_rows = rows.last_extent();
    ua.info = info;
    ua.rect = 1;
    ua.nt_c = (int) nt;
    ua.last_cols = cols.last_extent();
    if (reserve > 0)
      launch_update(ua, st, 3, std::max<long>(8, A.bulk_slots - reserve), counters, false);
    else
      launch_update(ua, st, 3);
  };

  // ================================================================================== Phase I
  // (1) diagonal tile: D <- L_kk^-1 D L_kk^-H  (lapack/tile.h:209-218 hegst); [L_kk | inverse blocks | 0.5 D] of
  // step k live in dws2[k & 1]
  auto diag_step = [&](long k, hipStream_t st) {
    const int kb = rows.tile_extent(k);
    T* dws = dws2[k & 1];
    T *Lkk = dws, *Wkk = dws + te, *Hs = dws + te + wel;
    if (rows.rank == rows.owner(k) && cols.rank == cols.owner(k)) {
      const long klc = cols.local_of(k);
      T* akk = A.tile(rows.local_of(k), klc);
      DLAF_HIP_CHECK(hipMemcpyAsync(Lkk, L.tile(rows.local_of(k), klc), te * sizeof(T), hipMemcpyDeviceToDevice, st));
      launch_invert_diag_blocks(Lkk, nb, kb, Wkk, info, st, false, false);
      launch_tile_xform(dfull, (long) nb, 0, akk, (long) nb, 0, kb, kb, 1, 1, 1.0, st);  // full Hermitian image
      trsm_tiles(dfull, 1, kb, Lkk, Wkk, kb, st);                                        // D L^-H
      launch_tile_xform(xt, (long) nb, 0, dfull, (long) nb, 0, kb, kb, 1, 0, 1.0, st);   // (D L^-H)^H
      trsm_tiles(xt, 1, kb, Lkk, Wkk, kb, st);                                           // (L^-1 D L^-H)^H
      launch_tile_xform(akk, (long) nb, 0, xt, (long) nb, 0, kb, kb, 1, 2, 1.0, st);     // lower triangle back
      launch_tile_xform(Hs, (long) nb, 0, akk, (long) nb, 0, kb, kb, 1, 1, 0.5, st);     // 0.5 D, full image
    }
  };
  // (2) panel: A_ik <- A_ik L_kk^-H, then A_ik -= 0.5 L_ik D  (impl.h:236-240)
  auto panel_step = [&](long k, hipStream_t st) {
    const int kb = rows.tile_extent(k);
    T* dws = dws2[k & 1];
    T *Lkk = dws, *Wkk = dws + te, *Hs = dws + te + wel;
    const bool in_col = cols.rank == cols.owner(k);
    const long il_n = rows.next_local(k + 1);
    if (in_col && rows.P > 1)
      tr->bcast(ax_col, rows.owner(k), rows.rank, dws, dws, (2 * te + wel) * sizeof(T), st);
    if (in_col && il_n < ltr) {
      const long klc = cols.local_of(k);
      TrsmArgs<T> ta;
      ta.b = A.tile(il_n, klc);
      ta.b_ts = (long) te;
      ta.ldb = nb;
      ta.il0 = (int) il_n;
      ta.il1 = (int) ltr;
      ta.pr = rows.P;
      ta.ri = rows.shift();
      ta.nb = nb;
      ta.nt = (int) nt;
      ta.last_rows = rows.last_extent();
      ta.l = Lkk;
      ta.ldl = nb;
      ta.winv = Wkk;
      ta.n = kb;
      ta.info = info;
      ta.prio = (st != s) ? 1 : 0;
      launch_trsm(ta, st);
      rect_update(il_n, ltr, klc, klc + 1, L.tile(il_n, klc), Hs, 0, kb, st);
    }
  };

  diag_step(0, sp);
  if (nt > 1)
    panel_step(0, sp);
  after(s, ev_panel[0], sp);
  for (long k = 0; k + 1 < nt; ++k) {
    if (tr)
      tr->mark(k);
    const int kb = rows.tile_extent(k);
    const int own_c = cols.owner(k);
    const bool in_col = cols.rank == own_c;
    const long il_n = rows.next_local(k + 1), jl_n = cols.next_local(k + 1);
    const long klc = in_col ? cols.local_of(k) : -1;
    T* Hs = dws2[k & 1] + te + wel;
    // ---- panels of A and L along process rows, their transposes down process columns --------------------
    T* colA = in_col ? A.tile(il_n < ltr ? il_n : 0, klc) : pA;
    T* colL = in_col ? L.tile(il_n < ltr ? il_n : 0, klc) : pL;
    if (cols.P > 1 && il_n < ltr) {
      tr->bcast(ax_row, own_c, cols.rank, colA, colA, (size_t) (ltr - il_n) * te * sizeof(T), s);
      tr->bcast(ax_row, own_c, cols.rank, colL, colL, (size_t) (ltr - il_n) * te * sizeof(T), s);
    }
    const T *rowA, *rowL;  // transposed panels: tile of local column jl
    long b_ts = (long) te, b_ts2 = 0;
    int b_period = 1;
    if (rows.P > 1) {
      A.bcast_transposed_panel(tr, ax_col, colA, il_n, jl_n, pAT, s, b_period, b_ts2);
      A.bcast_transposed_panel(tr, ax_col, colL, il_n, jl_n, pLT, s, b_period, b_ts2);
      rowA = pAT;
      rowL = pLT;
    }
    else {
      const long off = (cols.global_of(jl_n) - il_n) * (long) te;
      rowA = colA + off;
      rowL = colL + off;
      b_ts = (long) te * cols.P;
    }
    // ---- (3) trailing matrix: C -= A_ik L_jk^H + L_ik A_jk^H  (her2k on the diagonal tiles) -----------
    auto trailing = [&](long j0, long j1, long reserve) {
      if (il_n >= ltr || j0 >= j1)
        return;
      const long il0 = std::max(il_n, rows.next_local(cols.global_of(j0)));
      if (il0 >= ltr)
        return;
      UpdateArgs<T> ua;
      ua.c = A.tiles;
      ua.c_tsr = (long) te;
      ua.c_tsc = (long) (te * ltr);
      ua.ldc = nb;
      ua.a = colA + (size_t) (il0 - il_n) * te;
      ua.a2 = colL + (size_t) (il0 - il_n) * te;
      ua.a_ts = (long) te;
      ua.lda = nb;
      ua.b = rowL;
      ua.b2 = rowA;
      ua.b_ts = b_ts;
      ua.b_period = b_period;
      ua.b_ts2 = b_ts2;
      ua.b_jl0 = (int) jl_n;
      ua.ldb = nb;
      ua.il0 = (int) il0;
      ua.il1 = (int) ltr;
      ua.jl0 = (int) j0;
      ua.jl1 = (int) j1;
      ua.nb = nb;
      ua.K1 = kb;
      ua.K = 2 * kb;
      ua.her2k = 1;
      ua.pr = rows.P;
      ua.ri = rows.shift();
      ua.pc = cols.P;
      ua.ci = cols.shift();
      ua.nt = (int) nt;
      ua.last_rows = rows.last_extent();
      ua.info = info;
      if (reserve > 0)
        launch_update(ua, s, 3, std::max<long>(8, A.bulk_slots - reserve), counters, false);
      else
        launch_update(ua, s, 3);
    };
    // column k+1 first: what the diagonal tile and the panel of step k+1 need
    const long j_la = cols.mine(k + 1) ? jl_n + 1 : jl_n;
    trailing(jl_n, j_la, 0);
    after(sp, ev_la[k], s);
    diag_step(k + 1, sp);
    if (k + 2 < nt)
      panel_step(k + 1, sp);
    trailing(j_la, ltc, side_slots);
    // ---- (4) panel again: A_ik -= 0.5 L_ik D  (impl.h:263-266) -----------------------------------------
    if (in_col && il_n < ltr)
      rect_update(il_n, ltr, klc, klc + 1, L.tile(il_n, klc), Hs, 0, kb, s);
    after(s, ev_panel[k + 1], sp);
    if (dist)
      DLAF_HIP_CHECK(hipStreamSynchronize(s));  // single panel workspaces: reused by the next step
  }

  // ================================================================================== Phase II
  // (5) for every column at once: L X = strictly-block-lower(A), swept by tile rows (impl.h:268-280).  Row j:
  // R_j <- L_jj^-1 R_j through the adjoint tiles Tw2[j & 1]; the rows below take it from there.
  auto row_step = [&](long j, hipStream_t st) {
    const int kbj = rows.tile_extent(j);
    const int own_r = rows.owner(j), own_c = cols.owner(j);
    const bool in_row = rows.rank == own_r, in_col = cols.rank == own_c;
    const long ncl = cols.next_local(j);  // local tile columns left of the diagonal
    T* dws = dws2[j & 1];
    T *Lkk = dws, *Wkk = dws + te;
    T* Tw = Tw2[j & 1];
    if (in_row && in_col) {
      DLAF_HIP_CHECK(hipMemcpyAsync(Lkk, L.tile(rows.local_of(j), cols.local_of(j)), te * sizeof(T),
                                    hipMemcpyDeviceToDevice, st));
      launch_invert_diag_blocks(Lkk, nb, kbj, Wkk, info, st, false, false);
    }
    if (in_row && cols.P > 1)
      tr->bcast(ax_row, own_c, cols.rank, dws, dws, (te + wel) * sizeof(T), st);
    if (in_row && ncl > 0) {
      const long lr = rows.local_of(j);
      // R_j^H tile by tile: T_c = A(j, c)^H (nb x kbj);  T_c <- T_c L_jj^-H;  A(j, c) = T_c^H
      launch_tile_xform(Tw, (long) nb, (long) te, A.tile(lr, 0), (long) nb, (long) (te * ltr), kbj, nb, (int) ncl, 0, 1.0, st);
      trsm_tiles(Tw, ncl, nb, Lkk, Wkk, kbj, st);
      launch_tile_xform(A.tile(lr, 0), (long) nb, (long) (te * ltr), Tw, (long) nb, (long) te, nb, kbj, (int) ncl, 0, 1.0, st);
    }
  };
  if (nt > 1) {
    after(sp, ev_la[nt - 1], s);  // Phase I is complete on both streams
    row_step(1, sp);
    after(s, ev_panel[nt], sp);
  }
  for (long j = 1; j < nt; ++j) {
    if (tr)
      tr->mark(nt + j);
    const int kbj = rows.tile_extent(j);
    const int own_r = rows.owner(j), own_c = cols.owner(j);
    const bool in_col = cols.rank == own_c;
    const long il_n = rows.next_local(j + 1);
    const long ncl = cols.next_local(j);
    T* Tw = Tw2[j & 1];
    if (rows.P > 1 && ncl > 0)
      tr->bcast(ax_col, own_r, rows.rank, Tw, Tw, (size_t) ncl * te * sizeof(T), s);
    T* colL = in_col ? L.tile(il_n < ltr ? il_n : 0, cols.local_of(j)) : pL;
    if (cols.P > 1 && il_n < ltr)
      tr->bcast(ax_row, own_c, cols.rank, colL, colL, (size_t) (ltr - il_n) * te * sizeof(T), s);
    // A(i, c) -= L_ij T_c^H = L_ij A(j, c)  for the rows below j, the columns left of j: row j+1 first (the next
    // row to be solved), the others beside that solve
    const long il_la = (j + 1 < nt && rows.mine(j + 1)) ? il_n + 1 : il_n;
    rect_update(il_n, il_la, 0, ncl, colL, Tw, (long) te, kbj, s);
    if (j + 1 < nt) {
      after(sp, ev_la[j], s);
      row_step(j + 1, sp);
    }
    rect_update(il_la, ltr, 0, ncl, colL + (size_t) (il_la - il_n) * te, Tw, (long) te, kbj, s, side_slots);
    if (j + 1 < nt)
      after(s, ev_panel[j + 1], sp);
    if (dist)
      DLAF_HIP_CHECK(hipStreamSynchronize(s));
  }

  int h = 0;
  DLAF_HIP_CHECK(hipMemcpyAsync(&h, info, sizeof(int), hipMemcpyDeviceToHost, s));
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  if (lookahead) {
    DLAF_HIP_CHECK(hipStreamSynchronize(sp));
    DLAF_HIP_CHECK(hipStreamDestroy(sp));
  }
  for (auto* v : {&ev_panel, &ev_la})
    for (auto& e : *v)
      (void) hipEventDestroy(e);
  for (T* p : {dws2[0], dws2[1], dfull, xt, pA, pL, pAT, pLT, Tw2[0], Tw2[1]})
    DLAF_HIP_CHECK(hipFree(p));
  DLAF_HIP_CHECK(hipFree(counters));
  if (dist) {
    // the same value on every rank, as DeviceMatrix::wait() makes it for the factorization (MIN of the positive flags)
    constexpr double kTop = 2147483648.0;
    double v[2] = {h > 0 ? kTop - (double) h : 0.0, h == kInfoSchedulingFailure ? 1.0 : 0.0};
    tr->allreduce_max(v, 2, grid->nprow, grid->npcol, grid->myrow, grid->mycol);
    h = v[1] > 0 ? kInfoSchedulingFailure : (v[0] > 0 ? (int) (kTop - v[0]) : 0);
  }
  return h;
}

// Host entry: a, l = this process's local column-major parts of A and of the Cholesky factor of B
template <class T>
int gen_to_std_host(Grid* g, char uplo, T* a, long lda, const T* l, long ldl, long n, int nb, int isrc, int jsrc) {
  DeviceMatrix<T> A, Lm;
  A.create(g, uplo, n, nb, isrc, jsrc);
  Lm.create(g, uplo, n, nb, isrc, jsrc);
  A.upload(a, lda);
  Lm.upload(l, ldl);
  const int r = gen_to_std_device(A, Lm);
  A.download(a, lda, true);
  return r;
}

#define INST(T)                                                     \
  template int gen_to_std_device<T>(DeviceMatrix<T>&, DeviceMatrix<T>&); \
  template int gen_to_std_host<T>(Grid*, char, T*, long, const T*, long, long, int, int, int);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
