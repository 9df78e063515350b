// eigensolver.cpp -- band -> tridiagonal, the back-transformation band <- tridiagonal, and the drivers of the
// Hermitian (generalized) eigensolver (BASELINE configuration 5).
//
// Reference: Eigensolver::call (include/dlaf/eigensolver/eigensolver/impl.h:38-55 local, :57-95 distributed):
//   reduction_to_band -> band_to_tridiagonal<Backend::MC> -> tridiagonal_eigensolver -> bt_band_to_tridiagonal ->
//   bt_reduction_to_band;  GenEigensolver::call (gen_eigensolver/impl.h:33-92): cholesky_factorization of B,
//   generalized_to_standard, the eigensolver, triangular_solver (Left, uplo, ConjTrans) on the eigenvectors.
//   band_to_tridiagonal: band_to_tridiag/mc.h:681-867 (local), :1016-1558 (distributed, a pipeline of sweeps over
//   1-D blocks of the band with point-to-point messages); bt_band_to_tridiagonal: bt_band_to_tridiag/impl.h:609-736.
//
// MI355X design.  The band (2 b n elements: 42 MB at n = 20480) and the tridiagonal problem are small next to one
// GPU's memory and bandwidth, so a process grid does not pipeline them over ranks: the band is summed over the grid
// (one all-reduce), the bulge chase runs replicated, ONE launch per rank (kernels_tridiag.hip); what is distributed
// is what costs n^3: the eigenvector matrix of the back-transformations.
//   bt_band_to_tridiagonal: the b x b blocks of compact reflectors become well-formed 2b x b blocks V with their
//   T factors (all blocks at once: one expand launch, three strided-batch launches), then the blocks are applied in
//   WAVEFRONTS -- block (jb, step) touches rows [1 + (jb + step) b, +2b) and must follow (jb, step - 1) and the blocks
//   of sweep group jb + 1 that overlap it, so all blocks with the same step - jb are independent: W2 = V^H E and
//   E -= (V T) W2 run as two strided-batch GEMM launches per wavefront instead of two per block.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../device/band_api.hpp"
#include "../device/tridiag_api.hpp"
#include "eigensolver.hpp"

namespace dlaf_mi355x {

namespace {
template <class T>
T* ealloc(size_t elems) {
  T* p = nullptr;
  DLAF_HIP_CHECK(pool_malloc(reinterpret_cast<void**>(&p), std::max<size_t>(elems, 1) * sizeof(T)));
  return p;
}
double g_stage_ms[5] = {0, 0, 0, 0, 0};

struct StageTimer {
  hipEvent_t a, b;
  hipStream_t s;
  explicit StageTimer(hipStream_t st) : s(st) {
    DLAF_HIP_CHECK(hipEventCreate(&a));
    DLAF_HIP_CHECK(hipEventCreate(&b));
    DLAF_HIP_CHECK(hipEventRecord(a, s));
  }
  double stop() {
    DLAF_HIP_CHECK(hipEventRecord(b, s));
    DLAF_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    DLAF_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    DLAF_HIP_CHECK(hipEventDestroy(a));
    DLAF_HIP_CHECK(hipEventDestroy(b));
    return ms;
  }
};
}  // namespace

// E of the back-transformation band <- tridiagonal: leading dimension a multiple of 16 bytes, element 1 (not 0) on a
// 16-byte boundary
long bt_aligned_ld(long n) {
  return ((n + 1) / 2) * 2;
}
template <class T>
T* bt_aligned_base(T* alloc) {
  // hipMalloc returns 256-byte aligned memory: one element in front for 8-byte types, none for 16-byte ones, three for
  // 4-byte ones
  if (sizeof(T) == 16)
    return alloc;
  return alloc + (16 / sizeof(T) - 1);
}

void eigensolver_last_profile(double ms[5]) {
  for (int i = 0; i < 5; ++i)
    ms[i] = g_stage_ms[i];
}
void eigensolver_set_stage_ms(int stage, double ms) {
  g_stage_ms[stage] = ms;
}

// ------------------------------------------------------------------------------------------------ band -> tridiagonal
template <class T>
int band_to_tridiag_device(DeviceMatrix<T>& A, int band, real_t<T>* d, real_t<T>* e, T* v, long ldv) {
  if (A.transposed)
    fatal("[dlaf_mi355x] band_to_tridiagonal: the matrix must be held as uplo = L (band_to_tridiag.h:82-92: Upper is "
          "not implemented upstream either)\n");
  if (band < 2 || A.nb % band != 0)
    fatal("[dlaf_mi355x] band_to_tridiagonal: band_size %d must be >= 2 and divide the block size %d "
          "(band_to_tridiag.h:78-80)\n", band, A.nb);
  if (band > b2t_max_band())
    fatal("[dlaf_mi355x] band_to_tridiagonal: band_size %d above the supported %d\n", band, b2t_max_band());
  Grid* grid = A.grid;
  Transport* tr = grid_transport(*grid);
  const bool dist = grid->nranks > 1;
  if (dist && !tr)
    fatal("[dlaf_mi355x] grid with %d ranks has no transport\n", grid->nranks);
  const long n = A.n;
  hipStream_t s = A.s_high;
  DLAF_HIP_CHECK(hipMemsetAsync(A.info, 0, sizeof(int), s));
  if (n == 0)
    return 0;
  StageTimer timer(s);
  T* bandm = ealloc<T>((size_t) (n + 2) * 2 * band);
  // (the two columns of slack behind the matrix are read, never used: they must hold numbers)
  DLAF_HIP_CHECK(hipMemsetAsync(bandm + (size_t) n * 2 * band, 0, (size_t) 2 * 2 * band * sizeof(T), s));
  unsigned* sync = ealloc<unsigned>(b2t_sync_words(n));
  launch_band_extract(A.tiles, A.ltr, A.nb, A.rows.P, A.rows.shift(), A.cols.P, A.cols.shift(), n, band, bandm, s);
  if (dist)
    tr->allreduce_sum(bandm, (size_t) n * 2 * band, TypeInfo<T>::tag, 'A', s);
  DLAF_HIP_CHECK(hipMemset2DAsync(v, (size_t) ldv * sizeof(T), 0, (size_t) n * sizeof(T), (size_t) n, s));
  launch_band_to_tridiag(bandm, n, band, v, ldv, sync, A.info, s);
  launch_tridiag_extract(bandm, n, band, d, e, s);
  int h_info = 0;
  DLAF_HIP_CHECK(hipMemcpyAsync(&h_info, A.info, sizeof(int), hipMemcpyDeviceToHost, s));
  g_stage_ms[1] = timer.stop();
  DLAF_HIP_CHECK(pool_free(bandm));
  DLAF_HIP_CHECK(pool_free(sync));
  if (h_info == kInfoSchedulingFailure)
    fatal("[dlaf_mi355x] band_to_tridiagonal: a sweep gave up waiting for its predecessor (the workgroup that owns it "
          "made no progress)\n");
  return h_info;
}

template <class T>
int band_to_tridiag_host(Grid* g, const T* a, long lda, long n, int nb, int isrc, int jsrc, int band, real_t<T>* d,
                         real_t<T>* e, T* v, long ldv) {
  using R = real_t<T>;
  DeviceMatrix<T> A;
  A.create(g, 'L', n, nb, isrc, jsrc);
  A.upload(a, lda);
  R* dd = ealloc<R>((size_t) n);
  R* de = ealloc<R>((size_t) n);
  T* dv = ealloc<T>((size_t) n * n);
  const int r = band_to_tridiag_device(A, band, dd, de, dv, n);
  if (n > 0) {
    DLAF_HIP_CHECK(hipMemcpy(d, dd, (size_t) n * sizeof(R), hipMemcpyDeviceToHost));
    DLAF_HIP_CHECK(hipMemcpy(e, de, (size_t) n * sizeof(R), hipMemcpyDeviceToHost));
    DLAF_HIP_CHECK(hipMemcpy2D(v, (size_t) ldv * sizeof(T), dv, (size_t) n * sizeof(T), (size_t) n * sizeof(T), (size_t) n,
                               hipMemcpyDeviceToHost));
  }
  DLAF_HIP_CHECK(pool_free(dd));
  DLAF_HIP_CHECK(pool_free(de));
  DLAF_HIP_CHECK(pool_free(dv));
  return r;
}

// ------------------------------------------------------------------------------------------------ band <- tridiagonal
template <class T>
int bt_band_to_tridiag_device(long n, int band, const T* v, long ldv, T* e, long lde, long ncols, hipStream_t s) {
  const int b = band;
  const long nsweeps = TypeInfo<T>::is_complex ? n - 1 : n - 2;
  if (nsweeps <= 0 || ncols <= 0)
    return 0;
  if (ncols > 0x7fffffffL || n > 0x7fffffffL)
    fatal("[dlaf_mi355x] bt_band_to_tridiagonal: sizes beyond 32-bit tile counts\n");
  const long nblk = (n + b - 1) / b;
  const size_t vblk = (size_t) 2 * b * b;
  const size_t nb2 = (size_t) nblk * nblk;
  // DLAF_MI355X_BT_VERBOSE=1: wall time of the phases of this stage (synchronising: diagnosis only)
  static const bool verbose = [] {
    const char* e = std::getenv("DLAF_MI355X_BT_VERBOSE");
    return e && std::atoi(e) != 0;
  }();
  auto t_last = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!verbose)
      return;
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[dlaf_mi355x] bt_band_to_tridiagonal: %-28s %8.2f ms\n", what,
                 std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  static const bool dbg = [] {
    const char* e_ = std::getenv("DLAF_MI355X_EIG_DEBUG");
    return e_ && std::atoi(e_) != 0;
  }();
  auto checksum = [&](const char* what, const void* dev, size_t bytes) {
    if (!dbg)
      return;
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((bytes + 7) / 8, 0ull);
    DLAF_HIP_CHECK(hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost));
    unsigned long long acc = 1469598103934665603ull;
    for (unsigned long long x : h)
      acc = (acc ^ x) * 1099511628211ull;
    std::fprintf(stderr, "[dlaf_mi355x] bt debug %-22s %016llx\n", what, acc);
  };
  T* vx = ealloc<T>(nb2 * vblk);
  T* wx = ealloc<T>(nb2 * vblk);
  T* sm = ealloc<T>(nb2 * (size_t) b * b);
  T* tm = ealloc<T>(nb2 * (size_t) b * b);
  T* taus = ealloc<T>(nb2 * (size_t) b);
  // (no memsets: the expansion writes every block (jb <= ib) in full, zeros included, and nothing reads the others)
  phase("allocations");
  // fp64, band 128: the fused kernel (kernels_bt.hip) applies a block to a column strip in one go, on E transposed, and
  // streams V^T: the expansion writes that image directly (S = V^H V and W = V T take it as their operand);
  // DLAF_MI355X_BT_FUSED=0: the two strided-batch products per wavefront (every type, every band)
  static const bool fused_on = [] {
    const char* e = std::getenv("DLAF_MI355X_BT_FUSED");
    return e ? std::atoi(e) != 0 : true;
  }();
  const bool fused = fused_on && bt_fused_supported(b, sizeof(T), TypeInfo<T>::is_complex);
  launch_b2t_expand(v, ldv, n, b, vx, taus, s, fused);
  const T one = make_host_el<T>(1.0), zero = make_host_el<T>(0.0), mone = make_host_el<T>(-1.0);
  // Block (jb, ib) -- sweep group jb, rows 1 + ib b -- exists for jb <= ib only (ib = jb + step): the strided-batch
  // launches below take one row of the block grid at a time, the blocks ib = jb .. nblk - 1 of sweep group jb, instead
  // of all nblk^2 slots (half of which hold zeros: 28 ms of T factors alone at n = 20480)
  const long jb_end = std::min<long>(nblk, (nsweeps - 1) / b + 1);
  for (long jb = 0; jb < jb_end; ++jb) {
    const size_t q0 = (size_t) jb * nblk + (size_t) jb;
    const int cnt = (int) (nblk - jb);
    {
      GemmArgs<T> g;  // S = V^H V  (fused: vx holds V^T, S = V^T (V^T)^H for the real types of that path)
      g.M = b;
      g.N = b;
      g.K = 2 * b;
      g.a = vx + q0 * vblk;
      g.lda = fused ? b : 2 * b;
      g.opa = fused ? 'N' : 'C';
      g.b = vx + q0 * vblk;
      g.ldb = fused ? b : 2 * b;
      g.opb = fused ? 'C' : 'N';
      g.c = sm + q0 * (size_t) b * b;
      g.ldc = b;
      g.alpha = one;
      g.beta = zero;
      g.batch = cnt;
      g.sa = (long) vblk;
      g.sb = (long) vblk;
      g.sc = (long) b * b;
      launch_gemm(g, s);
    }
    launch_tfactor(sm + q0 * (size_t) b * b, (long) b, taus + q0 * (size_t) b, b, tm + q0 * (size_t) b * b, (long) b, s, cnt,
                   (long) b * b, (long) b, (long) b * b);
    {
      GemmArgs<T> g;  // W = V T
      g.M = 2 * b;
      g.N = b;
      g.K = b;
      g.a = vx + q0 * vblk;
      g.lda = fused ? b : 2 * b;
      g.opa = fused ? 'C' : 'N';
      g.b = tm + q0 * (size_t) b * b;
      g.ldb = b;
      g.opb = 'N';
      g.c = wx + q0 * vblk;
      g.ldc = 2 * b;
      g.alpha = one;
      g.beta = zero;
      g.batch = cnt;
      g.sa = (long) vblk;
      g.sb = (long) b * b;
      g.sc = (long) vblk;
      launch_gemm(g, s);
    }
  }
  if (dbg) {
    // existing blocks only (the others are never written): block row jb, blocks jb .. nblk - 1
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    for (const char* nm : {"vx", "wx", "tm"}) {
      unsigned long long acc = 1469598103934665603ull;
      for (long jb = 0; jb < jb_end; ++jb) {
        const size_t q0 = (size_t) jb * nblk + (size_t) jb, cnt = (size_t) (nblk - jb);
        const bool is_t = nm[0] == 't';
        const size_t per = is_t ? (size_t) b * b : vblk;
        const T* base = (nm[0] == 'v' ? vx : (nm[0] == 'w' ? wx : tm)) + q0 * per;
        std::vector<unsigned long long> h(cnt * per * sizeof(T) / 8);
        DLAF_HIP_CHECK(hipMemcpy(h.data(), base, cnt * per * sizeof(T), hipMemcpyDeviceToHost));
        for (unsigned long long x : h)
          acc = (acc ^ x) * 1099511628211ull;
      }
      std::fprintf(stderr, "[dlaf_mi355x] bt debug %-22s %016llx\n", nm, acc);
    }
  }
  phase("expand, S, T factors, W");
  double* et = nullptr;
  const long ldet = (ncols + 63) / 64 * 64;
  if constexpr (std::is_same_v<T, double>) {
    if (fused) {
      et = ealloc<double>((size_t) ldet * n);
      launch_bt_transpose(e, lde, n, ncols, et, ldet, s);
    }
  }
  if (et)
    checksum("et after transpose", et, (size_t) ldet * n * sizeof(double));
  phase("transposition");
  // steps of sweep group jb (its first sweep has the most)
  auto steps_of = [&](long jb) -> long {
    const long sw = jb * b;
    if (sw >= nsweeps)
      return 0;
    if (TypeInfo<T>::is_complex && sw == n - 2)
      return 1;
    return (n - sw - 2 + b - 1) / b;
  };
  const long jb_last = (nsweeps - 1) / b;
  const long max_batch = nblk / 2 + 2;
  T* w2 = ealloc<T>((size_t) max_batch * b * (size_t) ncols);
  // wavefront t = step - jb, from the last sweep group's first step to the first group's last step
  const long t_lo = -jb_last, t_hi = steps_of(0) - 1;
  for (long t = t_lo; t <= t_hi; ++t) {
    const long jb0 = std::max<long>(0, -t);
    // blocks (jb, st = t + jb), jb = jb0 ...: valid while st < steps_of(jb)
    long cnt = 0, full = 0;
    for (long jb = jb0; jb <= jb_last; ++jb) {
      const long st = t + jb;
      if (st >= steps_of(jb))
        break;
      const long r0 = 1 + (jb + st) * b;
      if (r0 >= n)
        break;
      ++cnt;
      if (r0 + 2 * b <= n)
        ++full;
    }
    if (cnt == 0)
      continue;
    if (cnt > max_batch)
      fatal("[dlaf_mi355x] bt_band_to_tridiagonal: wavefront of %ld blocks exceeds the workspace (%ld)\n", cnt, max_batch);
    // the blocks of a wavefront: block index (jb, ib = 2 jb + t) -> stride nblk + 2 blocks; rows 1 + (2 jb + t) b ->
    // stride 2 b rows of E
    auto run = [&](long first, long count, long rows) {
      if (count <= 0)
        return;
      const long jb = jb0 + first;
      const long ib = 2 * jb + t;
      const long blk = jb * nblk + ib;
      const long r0 = 1 + ib * b;
      if (fused) {
        if constexpr (std::is_same_v<T, double>)
          launch_bt_apply(vx + (size_t) blk * vblk, wx + (size_t) blk * vblk, (long) ((nblk + 2) * (long) vblk), (int) count, et,
                          ldet, ncols, r0, (int) rows, s);
        return;
      }
      GemmArgs<T> g;  // W2 = V^H E
      g.M = b;
      g.N = (int) ncols;
      g.K = (int) rows;
      g.a = vx + (size_t) blk * vblk;
      g.lda = 2 * b;
      g.opa = 'C';
      g.b = e + r0;
      g.ldb = lde;
      g.opb = 'N';
      g.c = w2;
      g.ldc = b;
      g.alpha = one;
      g.beta = zero;
      g.batch = (int) count;
      g.sa = (long) ((nblk + 2) * (long) vblk);
      g.sb = 2L * b;
      g.sc = (long) b * ncols;
      launch_gemm(g, s);
      GemmArgs<T> u;  // E -= W W2
      u.M = (int) rows;
      u.N = (int) ncols;
      u.K = b;
      u.a = wx + (size_t) blk * vblk;
      u.lda = 2 * b;
      u.opa = 'N';
      u.b = w2;
      u.ldb = b;
      u.opb = 'N';
      u.c = e + r0;
      u.ldc = lde;
      u.alpha = mone;
      u.beta = one;
      u.batch = (int) count;
      u.sa = (long) ((nblk + 2) * (long) vblk);
      u.sb = (long) b * ncols;
      u.sc = 2L * b;
      launch_gemm(u, s);
    };
    run(0, full, 2L * b);
    for (long q = full; q < cnt; ++q) {
      const long ib = 2 * (jb0 + q) + t;
      run(q, 1, n - (1 + ib * b));
    }
  }
  if (et)
    checksum("et after wavefronts", et, (size_t) ldet * n * sizeof(double));
  phase("wavefronts");
  if constexpr (std::is_same_v<T, double>) {
    if (fused)
      launch_bt_transpose(et, ldet, ncols, n, e, lde, s);
  }
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  phase("transposition back");
  for (T* q : {vx, wx, sm, tm, taus, w2})
    DLAF_HIP_CHECK(pool_free(q));
  if (et)
    DLAF_HIP_CHECK(pool_free(et));
  phase("frees");
  return 0;
}

template <class T>
int bt_band_to_tridiag_host(long n, int band, const T* v, long ldv, T* e, long lde, long ncols) {
  if (n <= 0 || ncols <= 0)
    return 0;
  hipStream_t s;
  DLAF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  T* dv = ealloc<T>((size_t) n * n);
  // row 1 of E on a 16-byte boundary, even leading dimension: the row blocks the reflector blocks act on start at the
  // rows 1 + i b, and the k-contiguous operand loader of the general product wants them aligned
  const long ldd = bt_aligned_ld(n);
  T* de_alloc = ealloc<T>((size_t) ldd * ncols + 4);
  T* de = bt_aligned_base(de_alloc);
  DLAF_HIP_CHECK(hipMemcpy2DAsync(dv, (size_t) n * sizeof(T), v, (size_t) ldv * sizeof(T), (size_t) n * sizeof(T), (size_t) n,
                                  hipMemcpyHostToDevice, s));
  DLAF_HIP_CHECK(hipMemcpy2DAsync(de, (size_t) ldd * sizeof(T), e, (size_t) lde * sizeof(T), (size_t) n * sizeof(T),
                                  (size_t) ncols, hipMemcpyHostToDevice, s));
  StageTimer timer(s);
  const int r = bt_band_to_tridiag_device(n, band, dv, n, de, ldd, ncols, s);
  g_stage_ms[3] = timer.stop();
  DLAF_HIP_CHECK(hipMemcpy2DAsync(e, (size_t) lde * sizeof(T), de, (size_t) ldd * sizeof(T), (size_t) n * sizeof(T),
                                  (size_t) ncols, hipMemcpyDeviceToHost, s));
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  DLAF_HIP_CHECK(pool_free(dv));
  DLAF_HIP_CHECK(pool_free(de_alloc));
  DLAF_HIP_CHECK(hipStreamDestroy(s));
  return r;
}

// ------------------------------------------------------------------------------------------------ drivers
// Eigensolver::call on resident operands: A (uplo L; destroyed: band + reflectors), eigenvalues to the host array w
// (all n on every rank), eigenvectors into the resident general matrix Z (n x n, A's block size and row source).
template <class T>
int hermitian_eigensolver_device(DeviceMatrix<T>& A, real_t<T>* w_host, GeneralMatrix<T>& Z) {
  using R = real_t<T>;
  const long n = A.n;
  const int nb = A.nb;
  if (n == 0)
    return 0;
  Grid* g = A.grid;
  TileMatrix<T>& C = Z.m;
  if (C.grid != g || C.nb != nb || C.rows.n != n || C.cols.n != n || C.rows.src != A.rows.src || C.transposed)
    fatal("[dlaf_mi355x] eigensolver: the eigenvector matrix must be n x n with A's block size and row source rank\n");
  const int band = get_band_size(nb);  // eigensolver/impl.h:41
  hipStream_t s = A.s_high;
  // a stage's status is made the same on every rank before anybody acts on it: a rank that left alone would leave
  // the others inside the next collective
  auto agreed = [&](int info) {
    if (g->nranks > 1 && g->transport) {
      double v[2] = {info > 0 ? (double) info : 0.0, info < 0 ? 1.0 : 0.0};
      g->transport->allreduce_max(v, 2, g->nprow, g->npcol, g->myrow, g->mycol);
      info = v[1] > 0 ? kInfoSchedulingFailure : (int) v[0];
    }
    return info;
  };
  // stages 2 and 3 run replicated: every rank holds the n x n reflector matrix (T), the n x n real eigenvector matrix
  // and the divide & conquer workspaces (4 n^2 real) -- say so instead of failing inside an allocation
  {
    size_t free_b = 0, total_b = 0;
    DLAF_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const double need = (double) n * n * (sizeof(T) + 5.0 * sizeof(R)) + (double) n * bt_aligned_ld(n) / g->npcol * sizeof(T);
    if (need > 0.95 * (double) free_b)
      fatal("[dlaf_mi355x] eigensolver: n = %ld needs %.1f GiB per rank for the replicated stages (band_to_tridiagonal, "
            "tridiagonal_eigensolver), %.1f GiB are free\n", n, need / 1073741824.0, (double) free_b / 1073741824.0);
  }
  // DLAF_MI355X_EIG_DEBUG=1: a checksum of every stage's output per rank (diagnosis: each stage is deterministic)
  static const bool dbg = [] {
    const char* e_ = std::getenv("DLAF_MI355X_EIG_DEBUG");
    return e_ && std::atoi(e_) != 0;
  }();
  auto checksum = [&](const char* what, const void* dev, size_t bytes) {
    if (!dbg)
      return;
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((bytes + 7) / 8, 0ull);
    DLAF_HIP_CHECK(hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost));
    unsigned long long acc = 1469598103934665603ull;
    for (unsigned long long x : h)
      acc = (acc ^ x) * 1099511628211ull;
    std::fprintf(stderr, "[dlaf_mi355x] eig debug rank (%d,%d) %-24s %016llx\n", g->myrow, g->mycol, what, acc);
  };
  std::vector<T> taus((size_t) std::max<long>(0, n - band - 1) + 1);
  int info = agreed(reduction_to_band_device(A, band, taus.data()));
  checksum("A after red2band", A.tiles, (size_t) A.ltr * A.ltc * A.tile_elems * sizeof(T));
  {
    double ms = 0, fl = 0;
    red2band_last_profile(&ms, &fl);
    g_stage_ms[0] = ms;
  }
  if (info != 0)
    return info;
  R* d = ealloc<R>((size_t) n);
  R* e = ealloc<R>((size_t) n);
  T* v = ealloc<T>((size_t) n * n);
  info = agreed(band_to_tridiag_device(A, band, d, e, v, n));
  if (info != 0) {
    for (R* q : {d, e})
      DLAF_HIP_CHECK(pool_free(q));
    DLAF_HIP_CHECK(pool_free(v));
    return info;
  }
  checksum("d", d, (size_t) n * sizeof(R));
  checksum("e", e, (size_t) (n - 1) * sizeof(R));
  checksum("v", v, (size_t) n * n * sizeof(T));
  R* wd = ealloc<R>((size_t) n);
  R* zr = ealloc<R>((size_t) n * n);
  {
    StageTimer t(s);
    tridiag_solver_device<R>(n, nb, d, e, wd, zr, n, s, g->nranks > 1 ? grid_transport(*g) : nullptr);
    g_stage_ms[2] = t.stop();
  }
  checksum("w", wd, (size_t) n * sizeof(R));
  checksum("z (tridiagonal)", zr, (size_t) n * n * sizeof(R));
  DLAF_HIP_CHECK(hipMemcpyAsync(w_host, wd, (size_t) n * sizeof(R), hipMemcpyDeviceToHost, s));
  // the columns of this process column, all rows (the back-transformation mixes rows, never columns)
  const long ncl = C.cols.local_size();
  const long lde = bt_aligned_ld(n);
  T* el_alloc = ealloc<T>((size_t) lde * std::max<long>(ncl, 1) + 4);
  T* el = bt_aligned_base(el_alloc);
  launch_cols_gather_cast<R, T>(zr, n, n, nb, C.cols.P, C.cols.shift(), ncl, el, lde, s);
  {
    StageTimer t(s);
    bt_band_to_tridiag_device(n, band, v, n, el, lde, ncl, s);
    g_stage_ms[3] = t.stop();
  }
  checksum("E after bt_b2t", el, (size_t) lde * std::max<long>(ncl, 1) * sizeof(T));
  launch_rows_to_tiles(el, lde, n, ncl, nb, C.rows.P, C.rows.shift(), C.ltr, C.ltc, C.tiles, s);
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  for (R* q : {d, e, wd, zr})
    DLAF_HIP_CHECK(pool_free(q));
  DLAF_HIP_CHECK(pool_free(v));
  DLAF_HIP_CHECK(pool_free(el_alloc));
  checksum("C before bt_red2band", C.tiles, (size_t) C.ltr * C.ltc * C.tile_elems * sizeof(T));
  info = agreed(bt_reduction_to_band_device(band, C, A, taus.data()));
  checksum("C after bt_red2band", C.tiles, (size_t) C.ltr * C.ltc * C.tile_elems * sizeof(T));
  {
    double ms = 0, fl = 0;
    red2band_last_profile(&ms, &fl);
    g_stage_ms[4] = ms;
  }
  return info;
}

template <class T>
int hermitian_eigensolver_host(Grid* g, char uplo, T* a, long lda, long n, int nb, int isrc, int jsrc, real_t<T>* w, T* z,
                               long ldz, int z_isrc, int z_jsrc) {
  if (uplo != 'L' && uplo != 'l')
    fatal("[dlaf_mi355x] eigensolver: uplo = %c is not implemented (neither upstream: eigensolver/impl.h:43-45)\n", uplo);
  if (z_isrc != isrc)
    fatal("[dlaf_mi355x] eigensolver: the eigenvector matrix must share A's row source rank (%d != %d)\n", z_isrc, isrc);
  DeviceMatrix<T> A;
  A.create(g, 'L', n, nb, isrc, jsrc);
  A.upload(a, lda);
  std::unique_ptr<MatrixBase> zh(general_matrix_create(g, TypeInfo<T>::tag, n, n, nb, z_isrc, z_jsrc));
  GeneralMatrix<T>& Z = static_cast<GeneralMatrix<T>&>(*zh);
  const int r = hermitian_eigensolver_device(A, w, Z);
  A.download(a, lda, true);  // (upstream leaves the band + reflectors in A as well)
  general_matrix_transfer(zh.get(), z, ldz, false);
  return r;
}

template <class T>
int hermitian_gen_eigensolver_host(Grid* g, char uplo, T* a, long lda, T* b, long ldb, long n, int nb, int a_isrc,
                                   int a_jsrc, int b_isrc, int b_jsrc, real_t<T>* w, T* z, long ldz, int z_isrc,
                                   int z_jsrc, bool b_factorized) {
  if (uplo != 'L' && uplo != 'l')
    fatal("[dlaf_mi355x] gen_eigensolver: uplo = %c is not implemented (neither upstream: eigensolver/impl.h:43-45)\n", uplo);
  if (a_isrc != b_isrc || a_jsrc != b_jsrc || z_isrc != a_isrc)
    fatal("[dlaf_mi355x] gen_eigensolver: A, B and the eigenvector matrix must share their source rank\n");
  DeviceMatrix<T> A, B;
  A.create(g, 'L', n, nb, a_isrc, a_jsrc);
  B.create(g, 'L', n, nb, b_isrc, b_jsrc);
  A.upload(a, lda);
  B.upload(b, ldb);
  // gen_eigensolver/impl.h:33-60: cholesky_factorization(uplo, B) unless the caller passes the factor,
  // generalized_to_standard(uplo, A, B), the eigensolver, triangular_solver(Left, uplo, ConjTrans, NonUnit, 1, B, Z)
  int info = 0;
  if (!b_factorized) {
    info = B.factorize();
    if (info != 0)
      return info;
  }
  info = gen_to_std_device(A, B);
  if (info != 0)
    return info;
  std::unique_ptr<MatrixBase> zh(general_matrix_create(g, TypeInfo<T>::tag, n, n, nb, z_isrc, z_jsrc));
  GeneralMatrix<T>& Z = static_cast<GeneralMatrix<T>&>(*zh);
  info = hermitian_eigensolver_device(A, w, Z);
  if (info != 0)
    return info;
  const T one = make_host_el<T>(1.0);
  B.type = TypeInfo<T>::tag;
  info = triangular_solver_device('L', 'L', 'C', 'N', &one, &B, zh.get());
  A.download(a, lda, true);
  if (!b_factorized)
    B.download(b, ldb, true);
  general_matrix_transfer(zh.get(), z, ldz, false);
  return info;
}

#define INST(T)                                                                                                          \
  template int hermitian_eigensolver_host<T>(Grid*, char, T*, long, long, int, int, int, real_t<T>*, T*, long, int, int); \
  template int hermitian_gen_eigensolver_host<T>(Grid*, char, T*, long, T*, long, long, int, int, int, int, int,         \
                                                 real_t<T>*, T*, long, int, int, bool);                                  \
  template int band_to_tridiag_device<T>(DeviceMatrix<T>&, int, real_t<T>*, real_t<T>*, T*, long);                       \
  template int band_to_tridiag_host<T>(Grid*, const T*, long, long, int, int, int, int, real_t<T>*, real_t<T>*, T*, long); \
  template int bt_band_to_tridiag_device<T>(long, int, const T*, long, T*, long, long, hipStream_t);                     \
  template int bt_band_to_tridiag_host<T>(long, int, const T*, long, T*, long, long);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
