// red2band.cpp -- reduction of a Hermitian matrix to band form, Q^H A Q = B, and the matching
// back-transformation C <- Q C (first stage of the eigensolver of BASELINE
// configuration 5).
//
// Reference: dlaf::eigensolver::internal::reduction_to_band (include/dlaf/eigensolver/reduction_to_band.h:40-122),
// ReductionToBand::call local (eigensolver/reduction_to_band/impl.h:968-1110) and distributed (:1113-1462);
// bt_reduction_to_band (eigensolver/bt_reduction_to_band.h, impl.h:132-370).  Per panel of `band` columns the
// reference: copies the panel to the HOST and computes its Householder reflectors there with a thread team
// (:881-961, :297-361), forms T (t_factor_impl.h), W = V T, X = A W tile by tile with partial sums reduced
// over both communicators (:692-807), W2 = W^H X, X -= 1/2 V W2, broadcasts x / v row- and column-wise and
// updates the trailing matrix with her2k / gemm tile tasks (:810-852).
//
// MI355X design.  The matrix stays in the tile layout of the Cholesky path (lower tiles), everything else is
// organised around REPLICATED, ZERO-EXTENDED PANELS: V, W and X are column-major arrays that cover the global
// rows [e0, n), e0 = the origin of the tile row that holds the first row r0 of the panel, with zeros in
// [e0, r0).  With that
//   * the trailing update A -= X V^H + V X^H is ONE launch of the Cholesky path's two-segment her2k update
//     kernel over whole tiles (the rows / columns in front of r0 subtract exact zeros): no sub-tile views;
//   * X = A W is one launch of tile_panel_kernel over the lower tiles (each tile used straight and
//     conjugate-transposed) + one reduction of the partial layers;
//   * on a process grid every rank holds the whole panels, so the reference's six row / column panel
//     broadcasts and two tile-wise reductions per step collapse into: an all-gather of the panel's tile rows
//     inside the owning process column, one row broadcast of the factored panel (+ taus), one all-reduce of X.
//     T, W, W2 and the X update are recomputed by every rank from the replicated panels (b x b work).  The panel
//     itself is factored redundantly by the ranks of the owning process column from the gathered copy: ONE
//     cooperative kernel (panel_qr_kernel) instead of a per-reflector reduction over the column communicator.
// One stream, no lookahead yet (DESIGN.md says what that costs).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../device/band_api.hpp"
#include "red2band.hpp"
#include "runtime.hpp"
#include "tile_matrix.hpp"

namespace dlaf_mi355x {

namespace {
template <class T>
T* dalloc(size_t elems) {
  T* p = nullptr;
  DLAF_HIP_CHECK(pool_malloc(reinterpret_cast<void**>(&p), std::max<size_t>(elems, 1) * sizeof(T)));
  return p;
}
template <class T>
T scalar(double re) {
  return make_host_el<T>(re);
}

double g_last_ms = 0, g_last_flops = 0;
long g_last_panels[2] = {0, 0};  // panels of the last reduction factored blocked / sent back to the reflector kernel
}  // namespace

void red2band_last_panels(long* blocked, long* fallback) {
  if (blocked)
    *blocked = g_last_panels[0];
  if (fallback)
    *fallback = g_last_panels[1];
}

void red2band_last_profile(double* ms, double* flops) {
  if (ms)
    *ms = g_last_ms;
  if (flops)
    *flops = g_last_flops;
}

// tune parameter eigensolver_min_band (include/dlaf/tune.h:71-75,128; default 100): set from DLAF_EIGENSOLVER_MIN_BAND /
// --dlaf:eigensolver-min-band by dlaf_initialize (src/init.cpp:220) or by dlaf_mi355x_set_eigensolver_min_band
static int g_eigensolver_min_band = 100;
int eigensolver_min_band() {
  return g_eigensolver_min_band;
}
void set_eigensolver_min_band(int b_min) {
  if (b_min < 2)
    fatal("[dlaf_mi355x] eigensolver_min_band = %d must be >= 2 (get_band_size.h:24)\n", b_min);
  g_eigensolver_min_band = b_min;
}

int get_band_size(int nb) {
  // eigensolver/internal/get_band_size.h:20-31: the smallest divisor of nb that is >= eigensolver_min_band, else nb
  const int min_band = g_eigensolver_min_band;
  for (int div = nb / min_band; div >= 2; --div)
    if (nb % div == 0)
      return nb / div;
  return nb;
}

template <class T>
int reduction_to_band_device(DeviceMatrix<T>& A, int band, T* taus_host) {
  if (A.transposed)
    fatal("[dlaf_mi355x] reduction_to_band: the matrix must be held as uplo = L (the reference references the lower "
          "triangle only, reduction_to_band.h:66-68)\n");
  if (band < 2 || A.nb % band != 0)
    fatal("[dlaf_mi355x] reduction_to_band: band_size %d must be >= 2 and divide the block size %d\n", band, A.nb);
  Grid* grid = A.grid;
  Transport* tr = grid_transport(*grid);
  const bool dist = grid->nranks > 1;
  if (dist && !tr)
    fatal("[dlaf_mi355x] grid with %d ranks has no transport\n", grid->nranks);
  const Axis& rows = A.rows;
  const Axis& cols = A.cols;
  const long n = A.n, nt = A.nt, ltr = A.ltr, ltc = A.ltc;
  const int nb = A.nb, b = band;
  const size_t te = A.tile_elems;
  hipStream_t s = A.s_high;
  int* info = A.info;
  DLAF_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), s));
  const long nrefls = std::max<long>(0, n - b - 1);
  if (nrefls == 0) {
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
  }
  const long npanels = (nrefls - 1) / b + 1;
  const long ldp = ((n + 15) / 16) * 16;

  // ---- workspaces --------------------------------------------------------------------------------------------
  T* qt = dalloc<T>((size_t) b * (size_t) n);
  // V and X alternate between two buffers: the bulk of the trailing update of panel p reads them on the second
  // stream while the panel chain of p + 1 is already writing the next ones
  T* Vb[2] = {dalloc<T>((size_t) ldp * b), dalloc<T>((size_t) ldp * b)};
  T* W = dalloc<T>((size_t) ldp * b);
  T* Xb[2] = {dalloc<T>((size_t) ldp * b), dalloc<T>((size_t) ldp * b)};
  T* S = dalloc<T>((size_t) b * b);
  T* Tm = dalloc<T>((size_t) b * b);
  T* W2 = dalloc<T>((size_t) b * b);
  T* taus = dalloc<T>((size_t) nrefls + 1);
  for (int q = 0; q < 2; ++q) {
    DLAF_HIP_CHECK(hipMemsetAsync(Vb[q], 0, (size_t) ldp * b * sizeof(T), s));
    DLAF_HIP_CHECK(hipMemsetAsync(Xb[q], 0, (size_t) ldp * b * sizeof(T), s));
  }
  DLAF_HIP_CHECK(hipMemsetAsync(W, 0, (size_t) ldp * b * sizeof(T), s));
  DLAF_HIP_CHECK(hipMemsetAsync(taus, 0, ((size_t) nrefls + 1) * sizeof(T), s));
  const int ksplit_max = std::max(1, gemm_pick_ksplit<T>(b, b, n));
  T* gpart = dalloc<T>(gemm_partial_elems<T>(b, b, ksplit_max));
  // partial layers of the xHEMM: (layers + 1) * out tiles at its maximum over the panels (trailing matrices of
  // ot x ot local tiles: the local tile counts shrink together)
  long cap_s = 2, cap_t = 2;
  for (long k = 0; k <= std::max(ltr, ltc); ++k) {
    const long otr = std::max<long>(ltr - k, 0), otc = std::max<long>(ltc - k, 0);
    if (otr == 0 || otc == 0)
      break;
    const int cs = tile_panel_pick_chunk(otr, nb, b, otc, true, sizeof(T));
    const int ct = tile_panel_pick_chunk(otc, nb, b, otr, true, sizeof(T));
    cap_s = std::max<long>(cap_s, (long) (tile_panel_layers(otc, cs) + 1) * otr);
    cap_t = std::max<long>(cap_t, (long) (tile_panel_layers(otr, ct) + 1) * otc);
    // (rows and columns of a process grid need not shrink in step: one more tile either way)
    cap_s = std::max<long>(cap_s, (long) (tile_panel_layers(otc + 1, cs) + 1) * (otr + 1));
    cap_t = std::max<long>(cap_t, (long) (tile_panel_layers(otr + 1, ct) + 1) * (otc + 1));
  }
  T* part_s = dalloc<T>((size_t) cap_s * nb * (size_t) b);
  T* part_t = dalloc<T>((size_t) cap_t * nb * (size_t) b);
  void* qr_scratch = nullptr;
  DLAF_HIP_CHECK(hipMalloc(&qr_scratch, panel_qr_scratch_bytes(b, sizeof(T))));
  // blocked panel factorization (kernels_hr.hip): the panel column-major, two Gram / Cholesky factors, U^T, V1, the
  // inverted diagonal blocks of the three triangular solves, the flag that sends a panel to the reflector-by-reflector
  // kernel, the cooperative POTRF's flags
  const bool blocked_any = panel_qr_blocked_supported(b, std::max<long>(n - b, 0), b, sizeof(T), TypeInfo<T>::is_complex);
  T *Pcm = nullptr, *hr_g = nullptr, *hr_l2 = nullptr, *hr_r = nullptr, *hr_lu = nullptr, *hr_y1 = nullptr, *hr_winv = nullptr;
  int* hr_flag = nullptr;
  unsigned* hr_sync = nullptr;
  const size_t hr_wblk = (size_t) ((b + kDiagBlock - 1) / kDiagBlock) * kDiagBlock * kDiagBlock;
  if (blocked_any) {
    Pcm = dalloc<T>((size_t) ldp * b);
    hr_g = dalloc<T>((size_t) b * b);
    hr_l2 = dalloc<T>((size_t) b * b);
    hr_r = dalloc<T>((size_t) b * b);
    hr_lu = dalloc<T>((size_t) b * b);
    hr_y1 = dalloc<T>((size_t) b * b);
    hr_winv = dalloc<T>(4 * hr_wblk);
    DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&hr_flag), 2 * sizeof(int)));  // [0] failure, [1] second pass skipped
    DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&hr_sync), sizeof(unsigned) * potrf_coop_sync_words(b)));
  }
  int* hr_flag_host = nullptr;
  DLAF_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hr_flag_host), sizeof(int), hipHostMallocDefault));
  long panels_blocked = 0, panels_fallback = 0;

  hipEvent_t ev0, ev1;
  DLAF_HIP_CHECK(hipEventCreate(&ev0));
  DLAF_HIP_CHECK(hipEventCreate(&ev1));
  DLAF_HIP_CHECK(hipEventRecord(ev0, s));
  // Lookahead (one process): the trailing update of panel p is issued in two launches -- the tile column that
  // holds panel p + 1 on the panel stream, everything else on a second stream -- so that the latency-bound panel
  // kernel of p + 1 runs beside the bulk of the update of p; X = A W of p + 1 waits for both.
  // DLAF_MI355X_R2B_LOOKAHEAD=1 turns it on (off by default, see below).
  // DLAF_MI355X_R2B_LOOKAHEAD=0/1 forces it; default: on where the panel is factored blocked (a chain of small kernels
  // that finds room beside the update: 27.9 -> 30.0 TFlop/s at N = 20480, nb = 512, profiles/r04_red2band_lookahead_ab.txt),
  // off with the reflector-by-reflector kernel (measured in round 3: no gain, its 1024-thread workgroups need whole CUs)
  static const int want_lookahead = [] {
    const char* e = std::getenv("DLAF_MI355X_R2B_LOOKAHEAD");
    return e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
  }();
  const bool blocked_panels = panel_qr_blocked_supported(b, std::max<long>(n - b, 0), b, sizeof(T), TypeInfo<T>::is_complex);
  const bool lookahead = (want_lookahead < 0 ? blocked_panels : want_lookahead != 0) && !dist && A.s_low != A.s_high;
  // workgroup slots the bulk of the trailing update leaves to the panel chain beside it (DLAF_MI355X_R2B_SLOTS; a plain
  // launch's queued workgroups are not overtaken by the side stream's kernels)
  static const long la_slots = [] {
    const char* e = std::getenv("DLAF_MI355X_R2B_SLOTS");
    return e ? std::max(0L, std::atol(e)) : 0L;  // measured: 0 (plain launch) 30.0, 64 28.5, 96 28.7, 160 29.7, 256 29.6 TFlop/s
  }();
  unsigned* la_counters = nullptr;
  if (lookahead && la_slots > 0)
    DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&la_counters), 32 * sizeof(unsigned)));
  hipStream_t s2 = lookahead ? A.s_low : s;
  hipEvent_t ev_x[2], ev_rest[2];
  for (int q = 0; q < 2; ++q) {
    DLAF_HIP_CHECK(hipEventCreateWithFlags(&ev_x[q], hipEventDisableTiming));
    DLAF_HIP_CHECK(hipEventCreateWithFlags(&ev_rest[q], hipEventDisableTiming));
  }
  if (lookahead) {
    DLAF_HIP_CHECK(hipEventRecord(ev_x[0], s));
    DLAF_HIP_CHECK(hipStreamWaitEvent(s2, ev_x[0], 0));  // the workspaces are ready
  }

  const CommAxis ax_row = CommAxis::Row, ax_col = CommAxis::Col;
  for (long p = 0; p < npanels; ++p) {
    if (tr)
      tr->mark(p);
    T* V = Vb[p & 1];
    T* X = Xb[p & 1];
    const long r0 = (p + 1) * b, c0 = p * b;
    const int nr = (int) std::min<long>(b, nrefls - c0);
    if (nr <= 0)
      break;
    const long I0 = r0 / nb, e0 = I0 * nb, J0 = c0 / nb;
    const long me = n - e0, m = n - r0, o = r0 - e0;
    const int cc = (int) (c0 % nb);
    const long il0 = rows.next_local(I0), jl0 = cols.next_local(I0);
    const int pcol = cols.owner(J0);
    const bool in_pcol = cols.rank == pcol;
    // ---- 1. the panel, transposed, gathered inside the owning process column ------------------------------------
    bool panel_t_ready = false;  // the blocked factorization left the T factor in Tm
    if (in_pcol) {
      const long jlp = cols.local_of(J0);
      launch_panel_move(A.tiles, ltr, nb, (int) il0, (int) ltr, (int) jlp, rows.P, rows.shift(), (int) nt,
                        rows.last_extent(), cc, b, qt, e0, r0, true, s);
      if (rows.P > 1) {
        tr->group_begin();
        for (long I = I0; I < nt; ++I)
          tr->bcast(ax_col, rows.owner(I), rows.rank, qt + (size_t) (I * nb - e0) * b, qt + (size_t) (I * nb - e0) * b,
                    (size_t) rows.tile_extent(I) * b * sizeof(T), s);
        tr->group_end();
      }
      // ---- 2. reflectors (xGEQR2 without the size-1 reflector) ---------------------------------------------------
      bool t_ready = false;
      if constexpr (!std::is_same_v<T, cfloat>) {  // (the small kernels are instantiated for float, double, cdouble)
        if (blocked_any && panel_qr_blocked_supported(b, m, nr, sizeof(T), TypeInfo<T>::is_complex)) {
          // CholeskyQR2 on the column-major copy, then the Householder reconstruction (kernels_hr.hip).  Every kernel
          // behind the first factorization looks at hr_flag and does nothing once it is raised.
          T* P = Pcm;
          T* qtp = qt + (size_t) o * b;
          const int nrt = (int) ((m + b - 1) / b);
          DLAF_HIP_CHECK(hipMemsetAsync(hr_flag, 0, 2 * sizeof(int), s));
          launch_hr_transpose(qtp, b, m, P, ldp, true, nullptr, s);
          auto gram = [&](T* out) {
            GemmArgs<T> g;
            g.M = b;
            g.N = b;
            g.K = (int) m;
            g.a = P;
            g.lda = ldp;
            g.opa = 'C';
            g.b = P;
            g.ldb = ldp;
            g.opb = 'N';
            g.c = out;
            g.ldc = b;
            g.alpha = scalar<T>(1.0);
            g.beta = scalar<T>(0.0);
            g.ksplit = std::min(ksplit_max, gemm_pick_ksplit<T>(b, b, m));
            g.partial = gpart;
            launch_gemm(g, s);
          };
          // X L^-T in place on the rows [row0, m) of P (tiles of b rows)
          auto solve = [&](long row0, const T* L, const T* winv_blocks, const int* status = nullptr) {
            TrsmArgs<T> ta;
            ta.b = P + row0;
            ta.b_ts = b;
            ta.ldb = (int) ldp;
            ta.il0 = 0;
            ta.il1 = (int) ((m - row0 + b - 1) / b);
            ta.pr = 1;
            ta.ri = 0;
            ta.nb = b;
            ta.nt = ta.il1;
            ta.last_rows = (int) ((m - row0) - (long) (ta.il1 - 1) * b);
            ta.l = L;
            ta.ldl = b;
            ta.winv = winv_blocks;
            ta.n = b;
            ta.info = status ? status : hr_flag;
            launch_trsm(ta, s);
          };
          (void) nrt;
          gram(hr_g);
          launch_potrf_coop(hr_g, b, b, hr_winv, hr_flag, 0, hr_sync, s, false, false);
          // cond(P) >= max / min of the factor's diagonal; CholeskyQR2 is safe far beyond this gate (cond^2 eps << 1)
          launch_hr_gate(hr_g, b, b, 1.0e4, hr_flag, s);
          solve(0, hr_g, hr_winv);
          gram(hr_l2);
          // second pass -- unless the first one left Q orthonormal already (|Q1^T Q1 - I| <= 2e-13: hr_skip is raised,
          // hr_l2 becomes the identity, and the three launches below, which take hr_skip as their status word, return
          // at once: 19 ms of the 380 at N = 20480)
          launch_hr_orth(hr_l2, b, b, 2.0e-13, hr_flag + 1, hr_flag, s);
          launch_potrf_coop(hr_l2, b, b, hr_winv + hr_wblk, hr_flag + 1, 0, hr_sync, s, false, false);
          launch_hr_gate(hr_l2, b, b, 0.0, hr_flag + 1, s);  // (no gate: zeroes the strict upper triangle)
          solve(0, hr_l2, hr_winv + hr_wblk, hr_flag + 1);
          {
            GemmArgs<T> g;  // R = L2^T L1^T
            g.M = b;
            g.N = b;
            g.K = b;
            g.a = hr_l2;
            g.lda = b;
            g.opa = 'C';
            g.b = hr_g;
            g.ldb = b;
            g.opb = 'C';
            g.c = hr_r;
            g.ldc = b;
            g.alpha = scalar<T>(1.0);
            g.beta = scalar<T>(0.0);
            launch_gemm(g, s);
          }
          // reconstruction: top block, V2 = Q2 U^-1, T = -U S V1^-T
          launch_hr_lu(P, ldp, b, hr_r, hr_lu, hr_y1, Tm, taus + c0, hr_flag, s);
          launch_invert_diag_blocks(hr_lu, b, b, hr_winv + 2 * hr_wblk, hr_flag, s, false, false);
          solve(b, hr_lu, hr_winv + 2 * hr_wblk);
          launch_invert_diag_blocks(hr_y1, b, b, hr_winv + 3 * hr_wblk, hr_flag, s, false, true);
          {
            TrsmArgs<T> ta;
            ta.b = Tm;
            ta.b_ts = b;
            ta.ldb = b;
            ta.il0 = 0;
            ta.il1 = 1;
            ta.pr = 1;
            ta.ri = 0;
            ta.nb = b;
            ta.nt = 1;
            ta.last_rows = b;
            ta.l = hr_y1;
            ta.ldl = b;
            ta.winv = hr_winv + 3 * hr_wblk;
            ta.n = b;
            ta.info = hr_flag;
            launch_trsm(ta, s);
          }
          launch_hr_transpose(qtp, b, m, P, ldp, false, hr_flag, s);
          DLAF_HIP_CHECK(hipMemcpyAsync(hr_flag_host, hr_flag, sizeof(int), hipMemcpyDeviceToHost, s));
          DLAF_HIP_CHECK(hipStreamSynchronize(s));
          t_ready = (*hr_flag_host == 0);
          ++(t_ready ? panels_blocked : panels_fallback);
        }
      }
      if (!t_ready)
        launch_panel_qr(qt + (size_t) o * b, m, b, nr, taus + c0, qr_scratch, info, s);
      panel_t_ready = t_ready;
      launch_panel_move(A.tiles, ltr, nb, (int) il0, (int) ltr, (int) jlp, rows.P, rows.shift(), (int) nt,
                        rows.last_extent(), cc, b, qt, e0, r0, false, s);
    }
    if (cols.P > 1) {
      tr->bcast(ax_row, pcol, cols.rank, qt, qt, (size_t) me * b * sizeof(T), s);
      tr->bcast(ax_row, pcol, cols.rank, taus + c0, taus + c0, (size_t) nr * sizeof(T), s);
    }
    // ---- 3. well-formed V, T factor, W = V T  (every rank, from the replicated panel) ------------------------
    launch_make_v(qt, b, nr, e0, r0, n, V, ldp, s);
    // (the blocked factorization delivers T = -U S V1^-T with the reflectors; on a grid with several process columns
    // the ranks outside the panel's column have only the reflectors, and every rank forms T the same way)
    if (!(panel_t_ready && cols.P == 1)) {
      GemmArgs<T> g;
      g.M = b;
      g.N = b;
      g.K = (int) me;
      g.a = V;
      g.lda = ldp;
      g.opa = 'C';
      g.b = V;
      g.ldb = ldp;
      g.opb = 'N';
      g.c = S;
      g.ldc = b;
      g.alpha = scalar<T>(1.0);
      g.beta = scalar<T>(0.0);
      g.ksplit = std::min(ksplit_max, gemm_pick_ksplit<T>(b, b, me));
      g.partial = gpart;
      launch_gemm(g, s);
      DLAF_HIP_CHECK(hipMemsetAsync(Tm, 0, (size_t) b * b * sizeof(T), s));
      launch_tfactor(S, (long) b, taus + c0, nr, Tm, (long) b, s);
    }
    {
      GemmArgs<T> g;
      g.M = (int) me;
      g.N = b;
      g.K = b;
      g.a = V;
      g.lda = ldp;
      g.opa = 'N';
      g.b = Tm;
      g.ldb = b;
      g.opb = 'N';
      g.c = W;
      g.ldc = ldp;
      g.alpha = scalar<T>(1.0);
      g.beta = scalar<T>(0.0);
      launch_gemm(g, s);
    }
    // ---- 4. X = A_t W (xHEMM on the lower tiles), summed over the grid -------------------------------------------
    TilePanelArgs<T> h;
    h.tiles = A.tiles;
    h.ltr = ltr;
    h.nb = nb;
    h.il0 = (int) il0;
    h.il1 = (int) ltr;
    h.jl0 = (int) jl0;
    h.jl1 = (int) ltc;
    h.pr = rows.P;
    h.ri = rows.shift();
    h.pc = cols.P;
    h.ci = cols.shift();
    h.nt_r = (int) nt;
    h.last_rows = rows.last_extent();
    h.nt_c = (int) nt;
    h.last_cols = cols.last_extent();
    h.herm = 1;
    h.w = W;
    h.ldw = ldp;
    h.e0 = e0;
    h.ncols = b;
    // (a rank without local trailing rows or columns has no work items at all: nothing to sum either)
    h.kinds = (il0 < ltr && jl0 < ltc) ? 3 : 0;
    h.chunk_s = tile_panel_pick_chunk(ltr - il0, nb, b, std::max<long>(ltc - jl0, 1), true, sizeof(T));
    h.chunk_t = tile_panel_pick_chunk(ltc - jl0, nb, b, std::max<long>(ltr - il0, 1), true, sizeof(T));
    h.layers_s = tile_panel_layers(std::max<long>(ltc - jl0, 1), h.chunk_s);
    h.layers_t = tile_panel_layers(std::max<long>(ltr - il0, 1), h.chunk_t);
    if ((size_t) (h.layers_s + 1) * (size_t) (ltr - il0) > (size_t) cap_s || (size_t) (h.layers_t + 1) * (size_t) (ltc - jl0) > (size_t) cap_t)
      fatal("[dlaf_mi355x] reduction_to_band: partial-layer workspace too small (%d x %ld, %d x %ld; caps %ld, %ld)\n",
            h.layers_s + 1, ltr - il0, h.layers_t + 1, ltc - jl0, cap_s, cap_t);
    h.part_s = part_s;
    h.part_t = part_t;
    if (lookahead && p > 0)
      DLAF_HIP_CHECK(hipStreamWaitEvent(s, ev_rest[(p - 1) & 1], 0));  // the bulk of the previous trailing update
    launch_tile_panel(h, s);
    launch_hemm_reduce(h, r0, X, ldp, s);
    if (dist)
      tr->allreduce_sum(X, (size_t) ldp * b, TypeInfo<T>::tag, 'A', s);
    // ---- 5. W2 = W^H X,  X -= 1/2 V W2 -----------------------------------------------------------------------------
    {
      GemmArgs<T> g;
      g.M = b;
      g.N = b;
      g.K = (int) me;
      g.a = W;
      g.lda = ldp;
      g.opa = 'C';
      g.b = X;
      g.ldb = ldp;
      g.opb = 'N';
      g.c = W2;
      g.ldc = b;
      g.alpha = scalar<T>(1.0);
      g.beta = scalar<T>(0.0);
      g.ksplit = std::min(ksplit_max, gemm_pick_ksplit<T>(b, b, me));
      g.partial = gpart;
      launch_gemm(g, s);
    }
    {
      GemmArgs<T> g;
      g.M = (int) me;
      g.N = b;
      g.K = b;
      g.a = V;
      g.lda = ldp;
      g.opa = 'N';
      g.b = W2;
      g.ldb = b;
      g.opb = 'N';
      g.c = X;
      g.ldc = ldp;
      g.alpha = scalar<T>(-0.5);
      g.beta = scalar<T>(1.0);
      launch_gemm(g, s);
    }
    // ---- 6. A_t -= X V^H + V X^H on the lower tiles (tile::her2k / 2 x tile::gemm, impl.h:545-585) ----------------
    // reserve > 0: persistent form that leaves that many workgroup slots to the panel chain running beside it
    auto her2k = [&](long ja, long jb, hipStream_t st, long reserve = 0) {
      if (il0 >= ltr || ja >= jb)
        return;
      UpdateArgs<T> ua;
      ua.c = A.tiles;
      ua.c_tsr = (long) te;
      ua.c_tsc = (long) (te * ltr);
      ua.ldc = nb;
      ua.a = X + (rows.global_of(il0) * nb - e0);
      ua.a2 = V + (rows.global_of(il0) * nb - e0);
      ua.a_ts = (long) rows.P * nb;
      ua.lda = (int) ldp;
      ua.b = V + (cols.global_of(ja) * nb - e0);
      ua.b2 = X + (cols.global_of(ja) * nb - e0);
      ua.b_ts = (long) cols.P * nb;
      ua.ldb = (int) ldp;
      ua.il0 = (int) il0;
      ua.il1 = (int) ltr;
      ua.jl0 = (int) ja;
      ua.jl1 = (int) jb;
      ua.nb = nb;
      ua.K1 = b;
      ua.K = 2 * b;
      ua.her2k = 1;
      ua.pr = rows.P;
      ua.ri = rows.shift();
      ua.pc = cols.P;
      ua.ci = cols.shift();
      ua.nt = (int) nt;
      ua.last_rows = rows.last_extent();
      ua.info = info;
      if (reserve > 0 && la_counters != nullptr) {
        unsigned* cnt = la_counters + 16 * (size_t) (p & 1);
        launch_update(ua, st, 3, std::max<long>(64, A.bulk_slots - reserve), cnt, false);
      }
      else
        launch_update(ua, st, 3);
    };
    if (lookahead) {
      // the tile column of the next panel first, on the panel stream; the rest beside the next panel chain
      const long jsplit = std::min<long>(ltc, jl0 + 1);
      DLAF_HIP_CHECK(hipEventRecord(ev_x[p & 1], s));
      her2k(jl0, jsplit, s);
      DLAF_HIP_CHECK(hipStreamWaitEvent(s2, ev_x[p & 1], 0));
      her2k(jsplit, ltc, s2, la_slots);
      DLAF_HIP_CHECK(hipEventRecord(ev_rest[p & 1], s2));
    }
    else
      her2k(jl0, ltc, s);
    if (dist)
      DLAF_HIP_CHECK(hipStreamSynchronize(s));
  }
  if (lookahead) {
    DLAF_HIP_CHECK(hipEventRecord(ev_rest[0], s2));
    DLAF_HIP_CHECK(hipStreamWaitEvent(s, ev_rest[0], 0));
  }
  DLAF_HIP_CHECK(hipEventRecord(ev1, s));
  int h_info = 0;
  DLAF_HIP_CHECK(hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (taus_host)
    DLAF_HIP_CHECK(hipMemcpyAsync(taus_host, taus, (size_t) nrefls * sizeof(T), hipMemcpyDeviceToHost, s));
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0;
  DLAF_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
  g_last_ms = ms;
  // miniapp_reduction_to_band.cpp:163-168: add_mul = 2/3 n^3 - n^2 nb, one add + one mul each (x4 complex)
  g_last_flops = (TypeInfo<T>::is_complex ? 4.0 : 1.0) * 2.0 * (2.0 / 3.0 * (double) n * n * n - (double) n * n * nb);
  DLAF_HIP_CHECK(hipEventDestroy(ev0));
  DLAF_HIP_CHECK(hipEventDestroy(ev1));
  for (int q = 0; q < 2; ++q) {
    DLAF_HIP_CHECK(hipEventDestroy(ev_x[q]));
    DLAF_HIP_CHECK(hipEventDestroy(ev_rest[q]));
  }
  for (T* q : {qt, Vb[0], Vb[1], W, Xb[0], Xb[1], S, Tm, W2, taus, gpart, part_s, part_t})
    DLAF_HIP_CHECK(pool_free(q));
  DLAF_HIP_CHECK(pool_free(qr_scratch));
  if (blocked_any) {
    for (T* q : {Pcm, hr_g, hr_l2, hr_r, hr_lu, hr_y1, hr_winv})
      DLAF_HIP_CHECK(pool_free(q));
    DLAF_HIP_CHECK(pool_free(hr_flag));
    DLAF_HIP_CHECK(pool_free(hr_sync));
  }
  DLAF_HIP_CHECK(hipHostFree(hr_flag_host));
  if (la_counters)
    DLAF_HIP_CHECK(hipFree(la_counters));
  g_last_panels[0] = panels_blocked;
  g_last_panels[1] = panels_fallback;
  if (h_info == kInfoSchedulingFailure)
    fatal("[dlaf_mi355x] reduction_to_band: the cooperative panel kernel could not make progress (its workgroups were "
          "not co-resident)\n");
  return h_info;
}

// Host entry: a = this process's local column-major part of the Hermitian matrix (lower triangle referenced),
// overwritten with the band + reflectors; taus: n - band - 1 values, all of them on every rank
template <class T>
int reduction_to_band_host(Grid* g, T* a, long lda, long n, int nb, int isrc, int jsrc, int band, T* taus) {
  DeviceMatrix<T> A;
  A.create(g, 'L', n, nb, isrc, jsrc);
  A.upload(a, lda);
  const int r = reduction_to_band_device(A, band, taus);
  A.download(a, lda, true);
  return r;
}


// ------------------------------------------------------------------------------------------------ back-transformation
// C <- Q C with Q = H_0 H_1 ... (the reflectors reduction_to_band left below the band of A), applied in blocks of nb
// reflectors, last block first (bt_reduction_to_band/impl.h:132-236 local, :239-370 distributed):
//     W = V T^H,   W2 = W^H C,   C -= V W2.
// C: n x k general matrix in tile layout with A's block size and row distribution.  Same replicated, zero-extended
// panels as above; W2 = W^H C runs as the adjoint kind of tile_panel_kernel over the tiles of C (its result W2^H is
// summed over the process column), C -= V W2 as one rectangular launch of the Cholesky update kernel.
template <class T>
int bt_reduction_to_band_device(int band, TileMatrix<T>& C, DeviceMatrix<T>& A, const T* taus_host) {
  if (A.transposed)
    fatal("[dlaf_mi355x] bt_reduction_to_band: the reflectors must be held as uplo = L\n");
  if (C.grid != A.grid || C.nb != A.nb || C.rows.n != A.n || C.rows.src != A.rows.src || C.transposed)
    fatal("[dlaf_mi355x] bt_reduction_to_band: C must have A's block size, row count and row source rank\n");
  Grid* grid = A.grid;
  Transport* tr = grid_transport(*grid);
  const bool dist = grid->nranks > 1;
  if (dist && !tr)
    fatal("[dlaf_mi355x] grid with %d ranks has no transport\n", grid->nranks);
  const Axis& rows = A.rows;
  const Axis& cols = A.cols;
  const Axis& ccols = C.cols;
  const long n = A.n, nt = A.nt, ltr = A.ltr;
  const int nb = A.nb, b = band;
  const size_t te = A.tile_elems;
  hipStream_t s = A.s_high;
  int* info = A.info;
  DLAF_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), s));
  const long total = n - b - 1;
  const long cltc = C.ltc, cltr = C.ltr;
  if (total <= 0 || ccols.n == 0) {
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
  }
  const long nblocks = (total - 1) / nb + 1;
  const long ldp = ((n + 15) / 16) * 16;
  const long ldw2 = std::max<long>(cltc * nb, 1);

  T* qt = dalloc<T>((size_t) nb * (size_t) n);
  T* V = dalloc<T>((size_t) ldp * nb);
  T* W = dalloc<T>((size_t) ldp * nb);
  T* S = dalloc<T>((size_t) nb * nb);
  T* Tm = dalloc<T>((size_t) nb * nb);
  T* W2H = dalloc<T>((size_t) ldw2 * nb);
  T* taus = dalloc<T>((size_t) total + 1);
  DLAF_HIP_CHECK(hipMemcpyAsync(taus, taus_host, (size_t) total * sizeof(T), hipMemcpyHostToDevice, s));
  DLAF_HIP_CHECK(hipMemsetAsync(V, 0, (size_t) ldp * nb * sizeof(T), s));
  DLAF_HIP_CHECK(hipMemsetAsync(W, 0, (size_t) ldp * nb * sizeof(T), s));
  const int ksplit_max = std::max(1, gemm_pick_ksplit<T>(nb, nb, n));
  T* gpart = dalloc<T>(gemm_partial_elems<T>(nb, nb, ksplit_max));
  long lay_cap = 2;
  for (long src = 1; src <= std::max<long>(cltr, 1); ++src) {
    const int ch = tile_panel_pick_chunk(std::max<long>(cltc, 1), nb, nb, src, false, sizeof(T));
    lay_cap = std::max<long>(lay_cap, (long) (tile_panel_layers(src, ch) + 1) * std::max<long>(cltc, 1));
  }
  T* part_t = dalloc<T>((size_t) lay_cap * nb * (size_t) nb);

  hipEvent_t ev0, ev1;
  DLAF_HIP_CHECK(hipEventCreate(&ev0));
  DLAF_HIP_CHECK(hipEventCreate(&ev1));
  DLAF_HIP_CHECK(hipEventRecord(ev0, s));

  const CommAxis ax_row = CommAxis::Row, ax_col = CommAxis::Col;
  for (long k = nblocks - 1; k >= 0; --k) {
    if (tr)
      tr->mark(k);
    const int nrefl = (int) std::min<long>(nb, total - k * nb);
    const long r0 = k * nb + b, c0 = k * nb;
    const long I0 = r0 / nb, e0 = I0 * nb, J0 = k;
    const long me = n - e0;
    const long il0 = rows.next_local(I0);
    const int pcol = cols.owner(J0);
    const bool in_pcol = cols.rank == pcol;
    if (in_pcol) {
      const long jlp = cols.local_of(J0);
      launch_panel_move(A.tiles, ltr, nb, (int) il0, (int) ltr, (int) jlp, rows.P, rows.shift(), (int) nt,
                        rows.last_extent(), 0, nrefl, qt, e0, r0, true, s);
      if (rows.P > 1) {
        tr->group_begin();
        for (long I = I0; I < nt; ++I)
          tr->bcast(ax_col, rows.owner(I), rows.rank, qt + (size_t) (I * nb - e0) * nrefl,
                    qt + (size_t) (I * nb - e0) * nrefl, (size_t) rows.tile_extent(I) * nrefl * sizeof(T), s);
        tr->group_end();
      }
    }
    if (cols.P > 1)
      tr->bcast(ax_row, pcol, cols.rank, qt, qt, (size_t) me * nrefl * sizeof(T), s);
    launch_make_v(qt, nrefl, nrefl, e0, r0, n, V, ldp, s);
    {
      GemmArgs<T> g;
      g.M = nrefl;
      g.N = nrefl;
      g.K = (int) me;
      g.a = V;
      g.lda = ldp;
      g.opa = 'C';
      g.b = V;
      g.ldb = ldp;
      g.opb = 'N';
      g.c = S;
      g.ldc = nb;
      g.alpha = scalar<T>(1.0);
      g.beta = scalar<T>(0.0);
      g.ksplit = std::min(ksplit_max, gemm_pick_ksplit<T>(nrefl, nrefl, me));
      g.partial = gpart;
      launch_gemm(g, s);
    }
    launch_tfactor(S, (long) nb, taus + c0, nrefl, Tm, (long) nb, s);
    {
      GemmArgs<T> g;  // W = V T^H
      g.M = (int) me;
      g.N = nrefl;
      g.K = nrefl;
      g.a = V;
      g.lda = ldp;
      g.opa = 'N';
      g.b = Tm;
      g.ldb = nb;
      g.opb = 'C';
      g.c = W;
      g.ldc = ldp;
      g.alpha = scalar<T>(1.0);
      g.beta = scalar<T>(0.0);
      launch_gemm(g, s);
    }
    if (cltc > 0) {
      // W2^H = C^H W over the local tiles of C with global tile row >= I0
      TilePanelArgs<T> h;
      h.tiles = C.tiles;
      h.ltr = cltr;
      h.nb = nb;
      h.il0 = (int) il0;
      h.il1 = (int) cltr;
      h.jl0 = 0;
      h.jl1 = (int) cltc;
      h.pr = rows.P;
      h.ri = rows.shift();
      h.pc = ccols.P;
      h.ci = ccols.shift();
      h.nt_r = (int) nt;
      h.last_rows = rows.last_extent();
      h.nt_c = (int) ccols.nt();
      h.last_cols = ccols.last_extent();
      h.herm = 0;
      h.w = W;
      h.ldw = ldp;
      h.e0 = e0;
      h.ncols = nrefl;
      h.kinds = 2;
      h.chunk_t = tile_panel_pick_chunk(cltc, nb, nrefl, std::max<long>(cltr - il0, 1), false, sizeof(T));
      h.layers_t = tile_panel_layers(std::max<long>(cltr - il0, 1), h.chunk_t);
      h.part_t = part_t;
      if (il0 < cltr) {
        launch_tile_panel(h, s);
        // (every tile column of C has the same cltr - il0 sources: all layers_t runs exist, + the extra layer)
        launch_layers_reduce(part_t, h.layers_t + h.split, cltc * nb, nrefl, W2H, ldw2, s);
      }
      else {
        DLAF_HIP_CHECK(hipMemsetAsync(W2H, 0, (size_t) ldw2 * nrefl * sizeof(T), s));
      }
      if (rows.P > 1)
        tr->allreduce_sum(W2H, (size_t) ldw2 * nrefl, TypeInfo<T>::tag, 'C', s);
      // C -= V W2 = V (W2^H)^H
      if (il0 < cltr) {
        UpdateArgs<T> ua;
        ua.c = C.tiles;
        ua.c_tsr = (long) te;
        ua.c_tsc = (long) (te * cltr);
        ua.ldc = nb;
        ua.a = V + (rows.global_of(il0) * nb - e0);
        ua.a_ts = (long) rows.P * nb;
        ua.lda = (int) ldp;
        ua.b = W2H;
        ua.b_ts = (long) nb;
        ua.ldb = (int) ldw2;
        ua.il0 = (int) il0;
        ua.il1 = (int) cltr;
        ua.jl0 = 0;
        ua.jl1 = (int) cltc;
        ua.nb = nb;
        ua.K = nrefl;
        ua.pr = rows.P;
        ua.ri = rows.shift();
        ua.pc = ccols.P;
        ua.ci = ccols.shift();
        ua.nt = (int) nt;
        ua.last_rows = rows.last_extent();
        ua.info = info;
        ua.rect = 1;
        ua.nt_c = (int) ccols.nt();
        ua.last_cols = ccols.last_extent();
        launch_update(ua, s, 3);
      }
    }
    if (dist)
      DLAF_HIP_CHECK(hipStreamSynchronize(s));
  }
  DLAF_HIP_CHECK(hipEventRecord(ev1, s));
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0;
  DLAF_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
  g_last_ms = ms;
  // miniapp_bt_reduction_to_band.cpp:160-164: add_mul = (m - b)^2 n
  g_last_flops = (TypeInfo<T>::is_complex ? 4.0 : 1.0) * 2.0 * (double) (n - b) * (double) (n - b) * (double) ccols.n;
  DLAF_HIP_CHECK(hipEventDestroy(ev0));
  DLAF_HIP_CHECK(hipEventDestroy(ev1));
  for (T* q : {qt, V, W, S, Tm, W2H, taus, gpart, part_t})
    DLAF_HIP_CHECK(pool_free(q));
  return 0;
}

template <class T>
int bt_reduction_to_band_host(Grid* g, int band, T* c, long ldc, long ncols_c, int c_jsrc, const T* a, long lda, long n,
                              int nb, int isrc, int jsrc, const T* taus) {
  DeviceMatrix<T> A;
  A.create(g, 'L', n, nb, isrc, jsrc);
  A.upload(a, lda);
  TileMatrix<T> C;
  C.create(g, false, n, ncols_c, nb, isrc, c_jsrc);
  C.upload(c, ldc, false, false, T{}, A.s_high);
  const int r = bt_reduction_to_band_device(band, C, A, taus);
  C.download(c, ldc, false, A.s_high);
  DLAF_HIP_CHECK(hipStreamSynchronize(A.s_high));
  return r;
}

#define INST(T)                                                           \
  template int reduction_to_band_device<T>(DeviceMatrix<T>&, int, T*);    \
  template int reduction_to_band_host<T>(Grid*, T*, long, long, int, int, int, int, T*); \
  template int bt_reduction_to_band_device<T>(int, TileMatrix<T>&, DeviceMatrix<T>&, const T*); \
  template int bt_reduction_to_band_host<T>(Grid*, int, T*, long, long, int, const T*, long, long, int, int, int, const T*);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
