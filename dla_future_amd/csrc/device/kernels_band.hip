// kernels_band.hip -- gfx950 kernels of the first eigensolver stage: reduction of a Hermitian matrix to band
// form and its back-transformation.
//
// Reference: eigensolver/reduction_to_band/impl.h (computePanelReflectors :297-361 -- which even the
// reference's GPU backend runs on the CPU after copying the panel to the host, :881-961 --, hemmComputeX
// :465-517, gemmComputeW2 :520-542, gemmUpdateX :448-462), factorization/qr/t_factor_impl.h:60-131,
// eigensolver/bt_reduction_to_band/impl.h:91-129.  The reference issues one BLAS call per tile; here:
//   * gemm_kernel / gemm_reduce_kernel: C = alpha op(A) op(B) + beta C on column-major operands with
//     optional split-K (the b x b Gram-type products whose inner dimension is the matrix size);
//   * tile_panel_kernel: the xHEMM X = A W (lower-stored Hermitian A in tile layout, tall panel W) and the
//     back-transformation's W2 = W^H C as one launch over (output block, source layer) work items: every lower
//     tile is read for its own row block (A_ij W_j) and, conjugate-transposed through the register stage of the
//     slab loader, for the mirrored one (A_ij^H W_i) -- no mirrored copy of the matrix, no second operand
//     layout;
//   * panel_qr_kernel: the Householder panel as ONE cooperative launch -- the panel is held transposed so
//     that a thread owns a column, a workgroup a range of rows; norm and P^H x of a reflector come out of the
//     SAME grid-wide exchange of partial sums, so a reflector costs one inter-workgroup hand-off;
//   * the small movers: panel <-> tiles (transposing through LDS), well-formed V, T factor.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "band_api.hpp"
#include "device_api.hpp"
#include "gemm_general.hpp"
#include "lane_ops.hpp"

namespace dlaf_mi355x {

namespace {

template <class T>
__device__ __forceinline__ T el_mul(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
  else
    return a * b;
}
template <class T>
__device__ __forceinline__ T el_add(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re + b.re, a.im + b.im};
  else
    return a + b;
}
template <class T>
__device__ __forceinline__ T el_sub(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re - b.re, a.im - b.im};
  else
    return a - b;
}
template <class T>
__device__ __forceinline__ T el_conj(const T& a) {
  return conj_if(a, true);
}
template <class T>
__device__ __forceinline__ bool el_is_zero(const T& a) {
  return re_of(a) == real_t<T>(0) && im_of(a) == real_t<T>(0);
}
template <class T>
__device__ __forceinline__ T el_neg(const T& a) {
  return make_el<T>(-re_of(a), -im_of(a));
}
template <class T>
__device__ __forceinline__ T el_scale(const T& a, real_t<T> s) {
  return make_el<T>(re_of(a) * s, im_of(a) * s);
}
// 1 / a
template <class T>
__device__ __forceinline__ T el_inv(const T& a) {
  using R = real_t<T>;
  if constexpr (TypeInfo<T>::is_complex) {
    const R d = a.re * a.re + a.im * a.im;
    return T{a.re / d, -a.im / d};
  }
  else
    return R(1) / a;
}

// ======================================================================================= general product
struct GemmMap {
  int MB, NB, KS;
  int kchunk;
};

// MA / MB >= 0: the slab modes of the operands fixed at compile time (the launcher checked alignment and K % BK): only
// that loader is compiled in, k-contiguous operands go through the conflict-free transposed LDS image
template <class T, int MA = -1, int MB = -1>
__global__ __launch_bounds__(GenCfg<T>::type::THREADS, 2) void gemm_kernel(GemmArgs<T> p, GemmMap mp) {
  using Cfg = typename GenCfg<T>::type;
  using R = real_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);
  const int bid = blockIdx.x;
  if (p.batch > 1) {
    p.a += (long) blockIdx.y * p.sa;
    p.b += (long) blockIdx.y * p.sb;
    p.c += (long) blockIdx.y * p.sc;
  }
  const int ks = bid / (mp.MB * mp.NB);
  const int rem = bid % (mp.MB * mp.NB);
  const int bm = rem % mp.MB, bn = rem / mp.MB;
  const int m0 = bm * Cfg::BM, n0 = bn * Cfg::BN;
  const int mrows = min(Cfg::BM, p.M - m0), ncols = min(Cfg::BN, p.N - n0);
  const int k0 = ks * mp.kchunk;
  const int kk = min(p.K - k0, mp.kchunk);
  OpDesc<T> da, db;
  if (p.opa == 'N') {
    da.p = p.a + m0 + (long) k0 * p.lda;
    da.rs = 1;
    da.ks = p.lda;
  }
  else {
    da.p = p.a + k0 + (long) m0 * p.lda;
    da.rs = p.lda;
    da.ks = 1;
    da.conj = 1;
  }
  if (p.opb == 'N') {
    db.p = p.b + k0 + (long) n0 * p.ldb;
    db.rs = p.ldb;
    db.ks = 1;
    db.conj = 1;
  }
  else {
    db.p = p.b + n0 + (long) k0 * p.ldb;
    db.rs = 1;
    db.ks = p.ldb;
  }
  Acc<Cfg> acc;
  acc.clear();
  if (kk > 0)
    gemm_acc<Cfg, T, MA, MB>(da, mrows, db, ncols, kk, lds, acc);
  if (mp.KS == 1) {
    const bool has_beta = !el_is_zero(p.beta);
    acc_foreach<Cfg, T>(acc, [&](int m, int n, const T& v) {
      if (m < mrows && n < ncols) {
        T* c = p.c + (m0 + m) + (long) (n0 + n) * p.ldc;
        T r = el_mul(p.alpha, v);
        if (has_beta)
          r = el_add(r, el_mul(p.beta, *c));
        *c = r;
      }
    });
  }
  else {
    T* blk = p.partial + ((size_t) (ks * mp.NB + bn) * mp.MB + bm) * (size_t) (Cfg::BM * Cfg::BN);
    acc_foreach<Cfg, T>(acc, [&](int m, int n, const T& v) { blk[m + n * Cfg::BM] = v; });
  }
}

// grid: x = chunk of 4 * kThreads elements of a block, y = output block
template <class T>
__global__ __launch_bounds__(kThreads) void gemm_reduce_kernel(GemmArgs<T> p, GemmMap mp) {
  using Cfg = typename GenCfg<T>::type;
  const int bm = blockIdx.y % mp.MB, bn = blockIdx.y / mp.MB;
  const int m0 = bm * Cfg::BM, n0 = bn * Cfg::BN;
  const int mrows = min(Cfg::BM, p.M - m0), ncols = min(Cfg::BN, p.N - n0);
  const bool has_beta = !el_is_zero(p.beta);
  const size_t blk = (size_t) (Cfg::BM * Cfg::BN);
  for (int e = blockIdx.x * kThreads + threadIdx.x; e < Cfg::BM * Cfg::BN; e += gridDim.x * kThreads) {
    const int m = e % Cfg::BM, n = e / Cfg::BM;
    if (m >= mrows || n >= ncols)
      continue;
    // (fixed order; four partial sums in flight)
    T s0 = zero_el<T>(), s1 = zero_el<T>(), s2 = zero_el<T>(), s3 = zero_el<T>();
    int ks = 0;
    for (; ks + 3 < mp.KS; ks += 4) {
      const size_t o = ((size_t) (ks * mp.NB + bn) * mp.MB + bm) * blk + e;
      const size_t st = (size_t) mp.NB * mp.MB * blk;
      s0 = el_add(s0, p.partial[o]);
      s1 = el_add(s1, p.partial[o + st]);
      s2 = el_add(s2, p.partial[o + 2 * st]);
      s3 = el_add(s3, p.partial[o + 3 * st]);
    }
    for (; ks < mp.KS; ++ks)
      s0 = el_add(s0, p.partial[((size_t) (ks * mp.NB + bn) * mp.MB + bm) * blk + e]);
    const T sum = el_add(el_add(s0, s1), el_add(s2, s3));
    T* c = p.c + (m0 + m) + (long) (n0 + n) * p.ldc;
    T r = el_mul(p.alpha, sum);
    if (has_beta)
      r = el_add(r, el_mul(p.beta, *c));
    *c = r;
  }
}

// ======================================================================================= tile x panel
struct TilePanelMap {
  int sub;     // BM-row blocks per tile
  int cbn;     // BN-column blocks of the panel
  long cnt_s;  // work items of kind S (they come first)
  long cnt_t;
  int chunk_s, chunk_t;    // source tiles per run (one run = one workgroup)
  int layers_s, layers_t;  // runs enumerated per output tile
  int base_s, base_t;      // first layer it writes
  int only_special;        // KIND 2: take only the sources the fixed-mode kernels leave out
};

// sources of output tile t: local tile columns [first, end) of tile row il0 + t (kind S) / local tile rows [first, end)
// of tile column jl0 + t (kind T); go = the output tile's global index
template <class T>
__device__ __forceinline__ void tile_panel_sources(const TilePanelArgs<T>& p, bool kind_t, int t, int& first, int& end) {
  if (!kind_t) {
    const int gi = (p.il0 + t) * p.pr + p.ri;
    first = p.jl0;
    end = p.jl1;
    if (p.herm)  // tiles on and left of the diagonal: gj = jl pc + ci <= gi
      end = gi >= p.ci ? min(p.jl1, (gi - p.ci) / p.pc + 1) : p.jl0;
  }
  else {
    const int gj = (p.jl0 + t) * p.pc + p.ci;
    first = p.il0;
    end = p.il1;
    if (p.herm && gj >= p.ri)  // tiles below the diagonal: gi = il pr + ri > gj
      first = max(p.il0, (gj - p.ri) / p.pr + 1);
  }
  if (end < first)
    end = first;
}

// KIND 0: kind S over the full off-diagonal tiles (A rows contiguous, W k-contiguous: slab modes 0 / 1 fixed)
// KIND 1: kind T over the full tiles (A k-contiguous, conjugated: modes 1 / 1 fixed)
// KIND 2: both kinds, any tile (run-time slab modes: Hermitian diagonal tiles, ragged extents, unaligned operands)
template <class T, int KIND>
__global__ __launch_bounds__(GenCfg<T>::type::THREADS, KIND == 2 ? 1 : 2) void tile_panel_kernel(TilePanelArgs<T> p,
                                                                                                   TilePanelMap mp) {
  using Cfg = typename GenCfg<T>::type;
  using R = real_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);
  long id = blockIdx.x;
  const bool kind_t = KIND == 1 || (KIND == 2 && id >= mp.cnt_s);
  if (KIND == 2 && kind_t)
    id -= mp.cnt_s;
  const int cb = (int) (id % mp.cbn);
  id /= mp.cbn;
  const int s = (int) (id % mp.sub);
  id /= mp.sub;
  const int ot = kind_t ? (p.jl1 - p.jl0) : (p.il1 - p.il0);
  const int t = (int) (id % ot);
  const int q = (int) (id / ot);
  const int chunk = kind_t ? mp.chunk_t : mp.chunk_s;
  const long te = (long) p.nb * p.nb;
  const int n0 = cb * Cfg::BN;
  const int ncols = min(Cfg::BN, p.ncols - n0);
  Acc<Cfg> acc;
  acc.clear();
  // the output tile: local tile row il (kind S) / local tile column jl (kind T); run q of its sources
  const int go = kind_t ? (p.jl0 + t) * p.pc + p.ci : (p.il0 + t) * p.pr + p.ri;
  const int ext_o = kind_t ? ((go == p.nt_c - 1) ? p.last_cols : p.nb) : ((go == p.nt_r - 1) ? p.last_rows : p.nb);
  const int mrows = min(Cfg::BM, ext_o - s * Cfg::BM);
  int first, end;
  tile_panel_sources(p, kind_t, t, first, end);
  if (mp.only_special == 0 || KIND != 2) {
    // (the general kernel in split mode keeps one run: all the special sources of an output tile)
    first += q * chunk;
    end = min(end, first + chunk);
  }
  if (mrows <= 0 || first >= end)
    return;  // no such run: the reduction does not read this layer of this tile
  {
    for (int src = first; src < end; ++src) {
      const int il = kind_t ? src : p.il0 + t;
      const int jl = kind_t ? p.jl0 + t : src;
      const int gi = il * p.pr + p.ri, gj = jl * p.pc + p.ci;
      // the tile the panel rows come from: global tile column gj (kind S) / global tile row gi (kind T)
      const int gk = kind_t ? gi : gj;
      const int kext = kind_t ? ((gi == p.nt_r - 1) ? p.last_rows : p.nb) : ((gj == p.nt_c - 1) ? p.last_cols : p.nb);
      // "special" sources: what the fixed-mode kernels cannot take
      const bool special = (p.herm && gi == gj) || kext != p.nb || ext_o != p.nb;
      if (KIND != 2 && special)
        continue;
      if (KIND == 2 && mp.only_special && !special)
        continue;
      const T* tile = p.tiles + ((long) il + (long) jl * p.ltr) * te;
      OpDesc<T> da, db;
      db.p = p.w + ((long) gk * p.nb - p.e0) + (long) n0 * p.ldw;
      db.rs = p.ldw;
      db.ks = 1;
      db.conj = 1;
      if constexpr (KIND == 0) {
        da.p = tile + s * Cfg::BM;
        da.rs = 1;
        da.ks = p.nb;
        gemm_acc<Cfg, T, 0, 1>(da, mrows, db, ncols, kext, lds, acc);
      }
      else if constexpr (KIND == 1) {
        da.p = tile + (long) (s * Cfg::BM) * p.nb;
        da.rs = p.nb;
        da.ks = 1;
        da.conj = 1;
        gemm_acc<Cfg, T, 1, 1>(da, mrows, db, ncols, kext, lds, acc);
      }
      else {
        if (kind_t) {
          da.p = tile + (long) (s * Cfg::BM) * p.nb;
          da.rs = p.nb;
          da.ks = 1;
          da.conj = 1;
        }
        else if (p.herm && gj == gi) {
          da.p = tile;
          da.rs = 1;
          da.ks = p.nb;
          da.herm = 1;
          da.roff = s * Cfg::BM;
        }
        else {
          da.p = tile + s * Cfg::BM;
          da.rs = 1;
          da.ks = p.nb;
        }
        gemm_acc<Cfg, T>(da, mrows, db, ncols, kext, lds, acc);
      }
    }
  }
  const long ldp = (long) ot * p.nb;
  const int layer = q + (kind_t ? mp.base_t : mp.base_s);
  T* part = (kind_t ? p.part_t : p.part_s) + (size_t) layer * (size_t) p.ncols * (size_t) ldp + (long) t * p.nb + s * Cfg::BM;
  acc_foreach<Cfg, T>(acc, [&](int m, int n, const T& v) {
    if (m < mrows && n < ncols)
      part[m + (long) (n0 + n) * ldp] = v;
  });
}

template <class T>
__global__ __launch_bounds__(kThreads) void hemm_reduce_kernel(TilePanelArgs<T> p, long r0, long n, T* x, long ldx) {
  // every row of the ldx x ncols array is written (zeros beyond the matrix: the array travels through an all-reduce)
  const long total = ldx * p.ncols;
  const long lds_ = (long) (p.il1 - p.il0) * p.nb, ldt = (long) (p.jl1 - p.jl0) * p.nb;
  for (long e = (long) blockIdx.x * kThreads + threadIdx.x; e < total; e += (long) gridDim.x * kThreads) {
    const long row = e % ldx;
    const int c = (int) (e / ldx);
    const long g = p.e0 + row;
    T v = zero_el<T>();
    if (g >= r0 && g < n) {
      const int gt = (int) (g / p.nb), r = (int) (g % p.nb);
      if ((p.kinds & 1) && gt >= p.ri && (gt - p.ri) % p.pr == 0) {
        const int il = (gt - p.ri) / p.pr;
        if (il >= p.il0 && il < p.il1) {
          int first, end;
          tile_panel_sources(p, false, il - p.il0, first, end);
          if (end > first) {
            const int runs = (end - first + p.chunk_s - 1) / p.chunk_s;
            for (int q = 0; q < runs; ++q)
              v = el_add(v, p.part_s[((size_t) q * p.ncols + c) * (size_t) lds_ + (long) (il - p.il0) * p.nb + r]);
            if (p.split)
              v = el_add(v, p.part_s[((size_t) p.layers_s * p.ncols + c) * (size_t) lds_ + (long) (il - p.il0) * p.nb + r]);
          }
        }
      }
      if ((p.kinds & 2) && gt >= p.ci && (gt - p.ci) % p.pc == 0) {
        const int jl = (gt - p.ci) / p.pc;
        if (jl >= p.jl0 && jl < p.jl1) {
          int first, end;
          tile_panel_sources(p, true, jl - p.jl0, first, end);
          if (end > first) {
            const int runs = (end - first + p.chunk_t - 1) / p.chunk_t;
            for (int q = 0; q < runs; ++q)
              v = el_add(v, p.part_t[((size_t) q * p.ncols + c) * (size_t) ldt + (long) (jl - p.jl0) * p.nb + r]);
            if (p.split)
              v = el_add(v, p.part_t[((size_t) p.layers_t * p.ncols + c) * (size_t) ldt + (long) (jl - p.jl0) * p.nb + r]);
          }
        }
      }
    }
    x[row + (long) c * ldx] = v;
  }
}

template <class T>
__global__ __launch_bounds__(kThreads) void layers_reduce_kernel(const T* part, int layers, long rows, int ncols, T* out,
                                                                 long ldo) {
  const long total = rows * ncols;
  for (long e = (long) blockIdx.x * kThreads + threadIdx.x; e < total; e += (long) gridDim.x * kThreads) {
    const long r = e % rows;
    const long c = e / rows;
    T v = zero_el<T>();
    for (int q = 0; q < layers; ++q)
      v = el_add(v, part[((size_t) q * ncols + c) * (size_t) rows + r]);
    out[r + c * ldo] = v;
  }
}

// ======================================================================================= movers
constexpr int kTR = 32;

// grid: x = 32-column chunk of the panel, y = 32-row chunk of a tile, z = local tile row (il - il0)
template <class T>
__global__ __launch_bounds__(kThreads) void panel_move_kernel(T* tiles, long ltr, int nb, int il0, int jl, int pr, int ri,
                                                              int nt, int last_rows, int cc, int b, T* qt, long e0,
                                                              long r0, int to_panel) {
  __shared__ T buf[kTR][kTR + 1];
  const int il = il0 + blockIdx.z;
  const int gi = il * pr + ri;
  const int rows_t = (gi == nt - 1) ? last_rows : nb;
  const int rr0 = blockIdx.y * kTR, c0 = blockIdx.x * kTR;
  if (rr0 >= rows_t)
    return;
  const long g0 = (long) gi * nb + rr0;
  if (g0 + kTR <= r0)
    return;
  T* tile = tiles + ((long) il + (long) jl * ltr) * (long) nb * nb;
  const int tx = threadIdx.x % kTR, ty = threadIdx.x / kTR;
  if (to_panel) {
    for (int c = ty; c < kTR; c += kThreads / kTR) {
      const int r = rr0 + tx;
      if (r < rows_t && c0 + c < b)
        buf[c][tx] = tile[r + (long) (cc + c0 + c) * nb];
    }
    __syncthreads();
    for (int r = ty; r < kTR; r += kThreads / kTR) {
      const long g = g0 + r;
      if (rr0 + r < rows_t && g >= r0 && c0 + tx < b)
        qt[(c0 + tx) + (g - e0) * b] = buf[tx][r];
    }
  }
  else {
    for (int r = ty; r < kTR; r += kThreads / kTR) {
      const long g = g0 + r;
      if (rr0 + r < rows_t && g >= r0 && c0 + tx < b)
        buf[tx][r] = qt[(c0 + tx) + (g - e0) * b];
    }
    __syncthreads();
    for (int c = ty; c < kTR; c += kThreads / kTR) {
      const int r = rr0 + tx;
      if (r < rows_t && g0 + tx >= r0 && c0 + c < b)
        tile[r + (long) (cc + c0 + c) * nb] = buf[c][tx];
    }
  }
}

// grid: x = 32-column chunk, y = 32-row chunk of the extended panel rows [e0, n)
template <class T>
__global__ __launch_bounds__(kThreads) void make_v_kernel(const T* qt, int b, int nr, long e0, long r0, long n, T* v,
                                                          long ldv) {
  __shared__ T buf[kTR][kTR + 1];
  const long row0 = (long) blockIdx.y * kTR;  // relative to e0
  const int c0 = blockIdx.x * kTR;
  const int tx = threadIdx.x % kTR, ty = threadIdx.x / kTR;
  const long me = n - e0;
  for (int r = ty; r < kTR; r += kThreads / kTR) {
    const long e = row0 + r;
    const int j = c0 + tx;
    T val = zero_el<T>();
    if (e < me && j < nr) {
      const long g = e0 + e;
      if (g == r0 + j)
        val = make_el<T>(real_t<T>(1), real_t<T>(0));
      else if (g > r0 + j)
        val = qt[j + e * b];
    }
    buf[tx][r] = val;
  }
  __syncthreads();
  for (int c = ty; c < kTR; c += kThreads / kTR) {
    const long e = row0 + tx;
    if (e < me && c0 + c < b)
      v[e + (long) (c0 + c) * ldv] = buf[c][tx];
  }
}

template <class T>
__global__ __launch_bounds__(kThreads) void zero_rows_kernel(T* x, long ldx, long nrows, int ncols) {
  const long total = nrows * ncols;
  for (long e = (long) blockIdx.x * kThreads + threadIdx.x; e < total; e += (long) gridDim.x * kThreads)
    x[e % nrows + (e / nrows) * ldx] = zero_el<T>();
}

// ======================================================================================= T factor
constexpr int kTfMax = 1024;  // 4 rows of 1024 complex doubles = 64 KiB of LDS

// T(j, j) = tau_j,  t_j = T(0:j, 0:j) (-tau_j S(0:j, j))  (t_factor_impl.h:60-131), written ROW-wise: row i of T needs
// only itself, S and taus --  T(i, j) = -tau_j sum_{l=i}^{j-1} T(i, l) S(l, j)  for j > i -- so the k rows are k
// independent waves (4 per workgroup), each a chain of k - i dot products whose operand columns of S are fetched a
// chunk ahead.  The row and a copy of the taus live in LDS.  (A thread-per-row variant without any cross-lane
// reduction -- scalar loads of S, the row in LDS -- was built and measured slower: a dependent chain of k^2 / 2
// LDS-fed FMAs per thread, 0.7 instead of 0.4 ms for k = 128.)

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
  return wave_sum_fast(v);  // (lane_ops.hpp: DPP + permlane swaps -- the T-factor kernel does one per step of its chain)
}

// kTfU: registers per lane and column (columns of up to 64 * kTfU rows), kTfD: columns in flight
template <class T, int kTfU, int kTfD>
__global__ __launch_bounds__(kThreads) void tfactor_kernel(const T* s, long lds_, const T* taus, int k, T* t, long ldt,
                                                           long bs, long btau, long bt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tf_raw[];
  s += (long) blockIdx.y * bs;  // strided batch
  taus += (long) blockIdx.y * btau;
  t += (long) blockIdx.y * bt;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (kThreads / 64) + wave;
  if (i >= k)
    return;  // (no workgroup barrier below: waves are independent)
  // per wave: the row of T being built and a copy of the taus (a global load of taus[j] inside the sequential loop
  // costs a memory latency per step)
  T* row = reinterpret_cast<T*>(tf_raw) + (size_t) wave * 2 * k;
  T* tau_s = row + k;
  for (int j = lane; j < i; j += 64)
    t[i + (long) j * ldt] = zero_el<T>();
  for (int j = i + lane; j < k; j += 64)
    tau_s[j] = taus[j];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane == 0)
    row[i] = tau_s[i];
  const bool pre = (k - i) <= 64 * kTfU;
  if (pre) {
    // columns of S in chunks of kTfD, the next chunk in flight while this one is consumed: a step then costs a wave
    // reduction and an LDS round trip instead of a global-memory latency (0.5 ms -> tens of microseconds for k = 128)
    T cur[kTfD][kTfU], nxt[kTfD][kTfU];
    auto fetch = [&](int j0, T (&dst)[kTfD][kTfU]) {
#pragma unroll
      for (int d = 0; d < kTfD; ++d)
#pragma unroll
        for (int u = 0; u < kTfU; ++u) {
          const int j = j0 + d, l = i + lane + 64 * u;
          dst[d][u] = (j < k && l < j) ? s[l + (long) j * lds_] : zero_el<T>();
        }
    };
    fetch(i + 1, nxt);
    for (int j0 = i + 1; j0 < k; j0 += kTfD) {
#pragma unroll
      for (int d = 0; d < kTfD; ++d)
#pragma unroll
        for (int u = 0; u < kTfU; ++u)
          cur[d][u] = nxt[d][u];
      if (j0 + kTfD < k)
        fetch(j0 + kTfD, nxt);
#pragma unroll
      for (int d = 0; d < kTfD; ++d) {
        const int j = j0 + d;
        if (j < k) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          T acc = zero_el<T>();
#pragma unroll
          for (int u = 0; u < kTfU; ++u) {
            const int l = i + lane + 64 * u;
            if (l < j)
              acc = el_add(acc, el_mul(row[l], cur[d][u]));
          }
          acc = wave_sum(acc);
          if (lane == 0)
            row[j] = el_neg(el_mul(tau_s[j], acc));
        }
      }
    }
  }
  else {
    for (int j = i + 1; j < k; ++j) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      T acc = zero_el<T>();
      for (int l = i + lane; l < j; l += 64)
        acc = el_add(acc, el_mul(row[l], s[l + (long) j * lds_]));
      acc = wave_sum(acc);
      if (lane == 0)
        row[j] = el_neg(el_mul(tau_s[j], acc));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int j = i + lane; j < k; j += 64)
    t[i + (long) j * ldt] = row[j];
}

// ======================================================================================= panel QR
constexpr int kQrMaxWg = 128;
constexpr int kQrDefaultWg = 128;  // most workgroups of a panel (DLAF_MI355X_QR_MAXWG): more = less work per exchange but G^2 partial-sum traffic
constexpr int kQrMaxColsPerThread = 2;  // b <= 512 (columns per thread: kernel template parameter CPT)
template <class T, int NT>
constexpr int qr_batch() {  // rows of a thread whose loads are in flight together (registers: batch x element size)
  return (sizeof(T) >= 16 ? 8 : 16) / (NT > 512 ? 2 : 1);
}
constexpr long kQrSpinLimit = 20000000;

struct QrMap {
  int nwg;
  int rows_per_wg;
  int ct;      // column threads (power of two)
  int wide;    // != 0: the partial-sum vectors are read in 16-byte words
  // Hand-off buffers.  Nothing another workgroup reads is ever read twice from the same 128-byte line in one launch:
  // the partial sums of step j live in slot j % nslots (nslots = the number of reflectors whenever that fits the
  // scratch budget), the head row of step j in its own line-padded row of `heads` -- a line re-read after another
  // XCD rewrote it may come out of the reader's L2 as it was (per-XCD L2s are not coherent; the acquire only drops L1).
  int nslots;
  long slot_elems;  // elements per slot (nwg vectors, padded to whole lines)
  long head_elems;  // elements per head row (padded to whole lines)
  long spin_limit;
};

template <class T>
__device__ __forceinline__ void qr_store_wt(T* p, const T& v) {
  if constexpr (sizeof(T) == 4) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  else if constexpr (sizeof(T) == 8) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  else {
    struct Two {
      unsigned long long a, b;
    };
    const Two tw = __builtin_bit_cast(Two, v);
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, tw.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, tw.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// qt[c + r * b]: element (r, c) of the m x b panel.  Workgroup g owns rows [g R, (g+1) R).  Step j:
//   every workgroup has published its share of d_c = sum_{r >= j} conj(P[r, c]) P[r, j], c >= j (partial[j & 1][g][c]);
//   after the exchange everybody forms (d_j = |x|^2, x0 = P[j, j], h_c = P[j, c]):
//       y = -sign(re x0) |x|,  tau = (y - x0) / y,  scale = 1 / (x0 - y)                 (xLARFG, impl.h:106-140)
//       w_c = conj(h_c) + scale (d_c - conj(h_c) x0)        (= P_t^H v with v = [1; scale x],  impl.h:143-185)
//   and updates its rows:  P[r, c] -= conj(tau) v_r conj(w_c)  (impl.h:188-228), P[r, j] = v_r, P[j, j] = y,
//   accumulating the partial sums of step j + 1 in the same pass.  A thread owns a column (CPT of them when b > 256):
//   its element of a row is a coalesced global access, the two values of the row every column needs (x_r and the old
//   P[r, j+1]) are staged in LDS before any of them is overwritten.  The row loop is branch-light on purpose: a first
//   version with a branch per row and 64-bit row arithmetic spent 12 of its 23 us per reflector issuing instructions.
// Hand-offs: write-through stores + drained counter / relaxed poll + one acquire (the protocol of the tile POTRF).
template <class T, int CPT, int NT>
__global__ __launch_bounds__(NT) void panel_qr_kernel(T* qt, long m, int b, int nr, T* taus, T* partial, T* heads,
                                                            unsigned* counters, int* info, QrMap mp) {
  using R = real_t<T>;
  constexpr int kB = qr_batch<T, NT>();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int nrg = NT / mp.ct;
  const int redrows = max(max(nrg, 8), mp.wide ? NT / (int) ((size_t) b * sizeof(T) / 16) : 0);
  T* red = reinterpret_cast<T*>(lds_raw);          // [redrows][b]: row-group / load-group partial sums, then d | h
  T* wv = red + (size_t) redrows * b;              // [b]
  T* sc = wv + b;                                  // conj(tau), scale, y, pad
  // x[parity][0/1][rows_per_wg]: columns j and j + 1 of my rows as they are BEFORE reflector j is applied (indexed by
  // r - row_lo).  A pass reads the buffers of its parity and -- from the threads that own columns j + 1 and j + 2,
  // which have just computed them -- fills those of the next step: no reload from memory, no extra round trip.
  T* xbuf = sc + 4;
  __shared__ unsigned flag_slot;
  const int g = blockIdx.x;
  // (32-bit row arithmetic: the launcher keeps m * b below 2^31)
  const int row_lo = g * mp.rows_per_wg;
  const int row_hi = min((int) m, row_lo + mp.rows_per_wg);
  const int tc = threadIdx.x % mp.ct, rg = threadIdx.x / mp.ct;
#ifdef DLAF_QR_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = wall_clock64();
#define QR_STAMP(k)                                  \
  do {                                               \
    const unsigned long long tn_ = wall_clock64();   \
    st[k] += tn_ - tprev;                            \
    tprev = tn_;                                     \
  } while (0)
#else
#define QR_STAMP(k)
#endif
  // one loop, one copy of every phase: j = -1 is the opening pass that only sums the dots of column 0
  for (int j = -1; j < nr; ++j) {
    T ctau = zero_el<T>(), scale = zero_el<T>(), y = zero_el<T>();
    if (j >= 0) {
      // ---- exchange: wait for every workgroup's share of column j ----------------------------------------------
      if (threadIdx.x == 0) {
        unsigned v;
        long spins = 0;
        while ((v = __hip_atomic_load(&counters[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned) mp.nwg) {
          __builtin_amdgcn_s_sleep(1);
          ++spins;
          if (spins > mp.spin_limit ||
              ((spins & 1023) == 0 && __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            v = 0xFFFFFFFFu;
            break;
          }
        }
        QR_STAMP(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flag_slot = v;
      }
      __syncthreads();
      QR_STAMP(1);
      if (flag_slot == 0xFFFFFFFFu) {
        if (threadIdx.x == 0)
          atomicCAS(info, 0, kInfoSchedulingFailure);
        return;
      }
      // ---- totals: every workgroup adds the shares in the same order, so all of them form the same reflector ----
      const T* part = partial + (size_t) (j % mp.nslots) * (size_t) mp.slot_elems;
      const T* head = (j == 0) ? qt : heads + (size_t) j * (size_t) mp.head_elems;  // (row 0 comes from the caller)
      T hrow[CPT];
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int c = tc + u * mp.ct;
        hrow[u] = (rg == 0 && c < b && c >= j) ? head[c] : zero_el<T>();
      }
      int G;  // partial-sum rows in `red`
      if (mp.wide) {
        // 16-byte words: thread t takes word t % W of the vectors q = t / W, t / W + G, ...; eight loads in flight
        typedef R rw __attribute__((ext_vector_type(16 / sizeof(R))));
        constexpr int NR = 16 / (int) sizeof(R);
        const int W = (int) ((size_t) b * sizeof(T) / 16);
        G = NT / W;
        const int w = threadIdx.x % W, grp = threadIdx.x / W;
        rw acc0;
#pragma unroll
        for (int e = 0; e < NR; ++e)
          acc0[e] = R(0);
        const rw* base = reinterpret_cast<const rw*>(part) + w;
        for (int q = grp; q < mp.nwg; q += 8 * G) {
          rw v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int qq = min(q + e * G, mp.nwg - 1);
            v[e] = base[(size_t) qq * W];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (q + e * G < mp.nwg)
              acc0 += v[e];
        }
        reinterpret_cast<rw*>(red)[(size_t) grp * W + w] = acc0;
      }
      else {
        G = nrg;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const int c = tc + u * mp.ct;
          if (c < b) {
            T a0 = zero_el<T>();
            if (c >= j)
              for (int q = rg; q < mp.nwg; q += nrg)
                a0 = el_add(a0, part[(size_t) q * b + c]);
            red[(size_t) rg * b + c] = a0;
          }
        }
      }
      __syncthreads();
      T dtot[CPT];
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int c = tc + u * mp.ct;
        dtot[u] = zero_el<T>();
        if (rg == 0 && c < b) {
          dtot[u] = red[c];
          for (int q2 = 1; q2 < G; ++q2)
            dtot[u] = el_add(dtot[u], red[(size_t) q2 * b + c]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int c = tc + u * mp.ct;
        if (rg == 0 && c < b) {
          red[c] = dtot[u];      // d_c
          red[b + c] = hrow[u];  // h_c
        }
      }
      __syncthreads();
      QR_STAMP(2);
      // ---- the reflector's scalars, w ---------------------------------------------------------------------------
      if (threadIdx.x == 0) {
        const T x0 = red[b + j];
        const R nrm2 = re_of(red[j]);
        // (a reflector with |x| == 0 is the identity: tau = 0 and the column stays as it is, impl.h:108-109 -- with
        // scale 1 and conj(tau) 0 the pass below rewrites every element with its own value)
        T tau = zero_el<T>(), sc1 = make_el<T>(R(1), R(0)), yy = x0;
        if (nrm2 != R(0)) {
          const R nrm = sqrt(nrm2);
          const R yr = __builtin_signbit(re_of(x0)) ? nrm : -nrm;
          yy = make_el<T>(yr, R(0));
          tau = el_scale(el_sub(yy, x0), R(1) / yr);
          sc1 = el_inv(el_sub(x0, yy));
        }
        sc[0] = el_conj(tau);
        sc[1] = sc1;
        sc[2] = yy;
        if (g == 0)
          taus[j] = tau;
      }
      __syncthreads();
      ctau = sc[0];
      scale = sc[1];
      y = sc[2];
      {
        const T x0 = red[b + j];
        for (int c = threadIdx.x; c < b; c += NT) {
          if (c <= j)
            continue;
          const T hc = el_conj(red[b + c]);
          wv[c] = el_add(hc, el_mul(scale, el_sub(red[c], el_mul(hc, x0))));
        }
      }
      __syncthreads();
      QR_STAMP(3);
    }
    // ---- pass over my rows: apply reflector j, sum the dots of column j + 1 -------------------------------------
    const int jn = j + 1;
    const bool dots = jn < nr;
    const bool upd = j >= 0;
    const int first = max(row_lo, max(j, 0));
    const int nrows = max(0, row_hi - first);
    const int par = (j + 1) & 1;
    const T* xs = xbuf + (size_t) (2 * par) * mp.rows_per_wg + (first - row_lo);       // column j
    const T* xo = xbuf + (size_t) (2 * par + 1) * mp.rows_per_wg + (first - row_lo);   // column j + 1
    T* xs_next = xbuf + (size_t) (2 * (par ^ 1)) * mp.rows_per_wg + (first - row_lo);      // column j + 1, updated
    T* xo_next = xbuf + (size_t) (2 * (par ^ 1) + 1) * mp.rows_per_wg + (first - row_lo);  // column j + 2, updated
    if (j < 0) {
      // opening pass: column 0 from memory
      T* xo_w = xbuf + (size_t) (2 * par + 1) * mp.rows_per_wg + (first - row_lo);
      for (int rr = threadIdx.x; rr < nrows; rr += NT)
        xo_w[rr] = dots ? qt[(first + rr) * b + jn] : zero_el<T>();
      __syncthreads();
    }
    const T wn = (upd && dots) ? el_conj(wv[jn]) : zero_el<T>();
    T wc[CPT], dnext[CPT], head_v[CPT];
    bool colact[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int c = tc + u * mp.ct;
      wc[u] = (upd && c < b && c > j) ? el_conj(wv[c]) : zero_el<T>();
      dnext[u] = zero_el<T>();
      head_v[u] = zero_el<T>();
      colact[u] = c < b && c >= j;
    }
    bool head_hit = false;
    const int rlast = max(nrows - 1, 0);
    for (int rb = rg; rb < nrows; rb += nrg * kB) {
      // the batch's elements first (independent loads, all in flight; rows past the end re-read the last one)
      T old[kB][CPT];
#pragma unroll
      for (int e = 0; e < kB; ++e) {
        const int r = first + min(rb + e * nrg, rlast);
#pragma unroll
        for (int u = 0; u < CPT; ++u)
          old[e][u] = colact[u] ? qt[r * b + tc + u * mp.ct] : zero_el<T>();
      }
#pragma unroll
      for (int e = 0; e < kB; ++e) {
        const int rr = rb + e * nrg;
        const int rrc = min(rr, rlast);
        const int r = first + rr;
        const bool rowact = rr < nrows;
        // (opening pass: nothing is applied and xs holds nothing yet -- 0 * garbage must not reach the sums)
        const T vr = !upd ? zero_el<T>() : (r == j) ? make_el<T>(R(1), R(0)) : el_mul(xs[rrc], scale);
        const T xov = dots ? xo[rrc] : zero_el<T>();
        const T xn = el_sub(xov, el_mul(ctau, el_mul(vr, wn)));
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const int c = tc + u * mp.ct;
          const T nvu = el_sub(old[e][u], el_mul(ctau, el_mul(vr, wc[u])));
          const T nv = (upd && c == j) ? ((r == j) ? y : vr) : nvu;
          if (rowact && colact[u] && upd && r != jn)
            qt[r * b + c] = nv;
          if (rowact && c == jn)
            xs_next[rr] = nv;
          if (rowact && c == jn + 1 && c < b)
            xo_next[rr] = nv;
          const bool is_head = rowact && colact[u] && r == jn;
          head_v[u] = is_head ? nv : head_v[u];
          head_hit = head_hit || is_head;
          const bool dact = rowact && dots && c >= jn && c < b && r >= jn;
          dnext[u] = dact ? el_add(dnext[u], el_mul(el_conj(nv), xn)) : dnext[u];
        }
      }
    }
    // the next head row is read by every workgroup after the next exchange: write-through
    if (head_hit && upd) {
#pragma unroll
      for (int u = 0; u < CPT; ++u)
        if (colact[u]) {
          qt[jn * b + tc + u * mp.ct] = head_v[u];
          qr_store_wt(&heads[(size_t) jn * (size_t) mp.head_elems + tc + u * mp.ct], head_v[u]);
        }
    }
    QR_STAMP(4);
    // ---- publish my share of column j + 1 ---------------------------------------------------------------------------
    if (dots) {
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int c = tc + u * mp.ct;
        if (c < b)
          red[(size_t) rg * b + c] = dnext[u];
      }
      __syncthreads();
      T* mine = partial + (size_t) (jn % mp.nslots) * (size_t) mp.slot_elems + (size_t) g * (size_t) b;
      for (int c = threadIdx.x; c < b; c += NT) {
        T sum = red[c];
        for (int q = 1; q < nrg; ++q)
          sum = el_add(sum, red[(size_t) q * b + c]);
        qr_store_wt(&mine[c], c >= jn ? sum : zero_el<T>());
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_fetch_add(&counters[jn], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else {
      // (the stores of the last pass need no hand-off; the barrier keeps the LDS staging of a further pass away)
      __syncthreads();
    }
    QR_STAMP(5);
  }
#ifdef DLAF_QR_STAMPS
  if (threadIdx.x == 0 && (g == 0 || g == mp.nwg - 1) && nr > 0)
    printf("[qr stamps] wg %d of %d rows/wg %d nr %d: us per step: spin %.2f fence+bar %.2f totals %.2f scalars+w %.2f pass %.2f publish %.2f\n",
           g, mp.nwg, mp.rows_per_wg, nr, st[0] * 0.01 / nr, st[1] * 0.01 / nr, st[2] * 0.01 / nr, st[3] * 0.01 / nr,
           st[4] * 0.01 / nr, st[5] * 0.01 / nr);
#endif
#undef QR_STAMP
}

}  // namespace

// ======================================================================================= launchers
template <class T>
size_t gemm_partial_elems(int M, int N, int ksplit) {
  using Cfg = typename GenCfg<T>::type;
  if (ksplit <= 1)
    return 0;
  const size_t MB = (size_t) (M + Cfg::BM - 1) / Cfg::BM, NB = (size_t) (N + Cfg::BN - 1) / Cfg::BN;
  return MB * NB * (size_t) ksplit * (size_t) (Cfg::BM * Cfg::BN);
}

template <class T>
int gemm_pick_ksplit(int M, int N, long K) {
  using Cfg = typename GenCfg<T>::type;
  const long blocks = (long) ((M + Cfg::BM - 1) / Cfg::BM) * ((N + Cfg::BN - 1) / Cfg::BN);
  // shortest K chunk of a split product (DLAF_MI355X_GEMM_KCHUNK): 512 left the tall-skinny b x b products of
  // reduction_to_band (b = 128: two output blocks) on 80 of the 256 compute units at K = 20480:
  // 85 us per Gram matrix, 44 us with 256, 40 with 128 -- profiles/r04_red2band_*
  static const long kmin = [] {
    const char* e = std::getenv("DLAF_MI355X_GEMM_KCHUNK");
    return e ? std::max(64L, std::atol(e)) : 128L;
  }();
  if (blocks >= 256 || K < 4 * kmin)
    return 1;
  long ks = std::min<long>((1024 + blocks - 1) / blocks, K / kmin);
  return (int) std::max<long>(1, std::min<long>(ks, 256));
}

template <class T>
void launch_gemm(const GemmArgs<T>& a, hipStream_t stream) {
  using Cfg = typename GenCfg<T>::type;
  if (a.M <= 0 || a.N <= 0)
    return;
  GemmMap mp;
  mp.MB = (a.M + Cfg::BM - 1) / Cfg::BM;
  mp.NB = (a.N + Cfg::BN - 1) / Cfg::BN;
  mp.KS = (a.ksplit > 1 && a.partial != nullptr && a.K > 0 && a.batch <= 1) ? a.ksplit : 1;
  int kchunk = (a.K + mp.KS - 1) / mp.KS;
  kchunk = ((kchunk + Cfg::BK - 1) / Cfg::BK) * Cfg::BK;
  if (kchunk <= 0)
    kchunk = Cfg::BK;
  mp.kchunk = kchunk;
  if (mp.KS > 1)
    mp.KS = (a.K + kchunk - 1) / kchunk;
  // fixed-mode instantiation when every block of the product may take the 16-byte loaders
  constexpr int VE = (16 / (int) sizeof(T)) > 0 ? (16 / (int) sizeof(T)) : 1;
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  auto ld16 = [](long ld) { return (ld * (long) sizeof(T)) % 16 == 0; };
  static const bool allow_fixed = [] {
    const char* e = std::getenv("DLAF_MI355X_GEMM_FIXED");
    return e ? std::atoi(e) != 0 : true;
  }();
  const bool kfull = a.K > 0 && a.K % Cfg::BK == 0 && kchunk % Cfg::BK == 0;
  const bool batch_ok = a.batch <= 1 || (ld16(a.sa) && ld16(a.sb));
  const bool a_ok = al16(a.a) && ld16(a.lda) && (a.opa != 'N' || a.M % VE == 0);
  const bool b_ok = al16(a.b) && ld16(a.ldb) && (a.opb == 'N' || a.N % VE == 0);
  const dim3 grid((unsigned) (mp.MB * mp.NB * mp.KS), (unsigned) std::max(a.batch, 1));
  if (allow_fixed && kfull && batch_ok && a_ok && b_ok) {
    const int ma = a.opa == 'N' ? 0 : 1, mb = a.opb == 'N' ? 1 : 0;
    if (ma == 0 && mb == 0)
      hipLaunchKernelGGL((gemm_kernel<T, 0, 0>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
    else if (ma == 0 && mb == 1)
      hipLaunchKernelGGL((gemm_kernel<T, 0, 1>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
    else if (ma == 1 && mb == 0)
      hipLaunchKernelGGL((gemm_kernel<T, 1, 0>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
    else
      hipLaunchKernelGGL((gemm_kernel<T, 1, 1>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
  }
  else {
    hipLaunchKernelGGL((gemm_kernel<T>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
  }
  if (mp.KS > 1)
    hipLaunchKernelGGL((gemm_reduce_kernel<T>), dim3((unsigned) ((Cfg::BM * Cfg::BN + kThreads - 1) / kThreads), (unsigned) (mp.MB * mp.NB)),
                       dim3(kThreads), 0, stream, a, mp);
}

int tile_panel_pick_chunk(long out_tiles, int nb, int ncols, long max_src, bool triangle, size_t elem_size) {
  if (out_tiles <= 0 || max_src <= 1)
    return 1;
  const int bm = 128, bn = elem_size == 16 ? 64 : 128;
  const long blocks = out_tiles * ((nb + bm - 1) / bm) * ((ncols + bn - 1) / bn);
  // source tiles in all: a triangle has about half of out_tiles x max_src
  const long tiles = blocks * max_src / (triangle ? 2 : 1);
  static const long target = [] {
    const char* e = std::getenv("DLAF_MI355X_TILE_PANEL_ITEMS");  // workgroups per kind to aim at
    return e ? std::max(1L, std::atol(e)) : 1024L;
  }();
  long chunk = (tiles + target - 1) / target;
  chunk = std::max<long>(1, std::min<long>(chunk, max_src));
  // at most 64 layers
  chunk = std::max<long>(chunk, (max_src + 63) / 64);
  return (int) chunk;
}

template <class T>
void launch_tile_panel(TilePanelArgs<T>& a, hipStream_t stream) {
  using Cfg = typename GenCfg<T>::type;
  a.split = 0;
  if (a.il1 <= a.il0 || a.jl1 <= a.jl0 || a.ncols <= 0)
    return;
  TilePanelMap mp;
  mp.sub = (a.nb + Cfg::BM - 1) / Cfg::BM;
  mp.cbn = (a.ncols + Cfg::BN - 1) / Cfg::BN;
  const long per_s = (long) (a.il1 - a.il0) * mp.sub * mp.cbn, per_t = (long) (a.jl1 - a.jl0) * mp.sub * mp.cbn;
  // fixed-mode kernels: whole slabs (nb a multiple of the slab depth), 16-byte aligned operands
  constexpr int VE = (16 / (int) sizeof(T)) > 0 ? (16 / (int) sizeof(T)) : 1;
  static const bool allow_split = [] {
    const char* e = std::getenv("DLAF_MI355X_TILE_PANEL_SPLIT");
    return e ? std::atoi(e) != 0 : true;
  }();
  const bool fast = allow_split && a.nb % Cfg::BK == 0 && a.nb % VE == 0 && a.ldw % VE == 0 &&
                    reinterpret_cast<uintptr_t>(a.w) % 16 == 0 && reinterpret_cast<uintptr_t>(a.tiles) % 16 == 0 &&
                    (a.e0 % VE) == 0 && (a.nt_r > 2 || a.nt_c > 2);
  mp.chunk_s = a.chunk_s;
  mp.chunk_t = a.chunk_t;
  if (fast) {
    a.split = 1;
    mp.only_special = 1;
    mp.layers_s = a.layers_s;
    mp.layers_t = a.layers_t;
    mp.base_s = mp.base_t = 0;
    if (a.kinds & 1) {
      mp.cnt_s = per_s * a.layers_s;
      mp.cnt_t = 0;
      hipLaunchKernelGGL((tile_panel_kernel<T, 0>), dim3((unsigned) mp.cnt_s), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
    }
    if (a.kinds & 2) {
      mp.cnt_s = 0;
      mp.cnt_t = per_t * a.layers_t;
      hipLaunchKernelGGL((tile_panel_kernel<T, 1>), dim3((unsigned) mp.cnt_t), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
    }
    // the rest: one extra layer per kind
    mp.base_s = a.layers_s;
    mp.base_t = a.layers_t;
    mp.layers_s = mp.layers_t = 1;
    mp.cnt_s = (a.kinds & 1) ? per_s : 0;
    mp.cnt_t = (a.kinds & 2) ? per_t : 0;
    if (mp.cnt_s + mp.cnt_t > 0)
      hipLaunchKernelGGL((tile_panel_kernel<T, 2>), dim3((unsigned) (mp.cnt_s + mp.cnt_t)), dim3(Cfg::THREADS), Cfg::LDS_BYTES,
                         stream, a, mp);
    return;
  }
  mp.only_special = 0;
  mp.layers_s = a.layers_s;
  mp.layers_t = a.layers_t;
  mp.base_s = mp.base_t = 0;
  mp.cnt_s = (a.kinds & 1) ? per_s * a.layers_s : 0;
  mp.cnt_t = (a.kinds & 2) ? per_t * a.layers_t : 0;
  if (mp.cnt_s + mp.cnt_t == 0)
    return;
  hipLaunchKernelGGL((tile_panel_kernel<T, 2>), dim3((unsigned) (mp.cnt_s + mp.cnt_t)), dim3(Cfg::THREADS), Cfg::LDS_BYTES,
                     stream, a, mp);
}

template <class T>
void launch_hemm_reduce(const TilePanelArgs<T>& a, long r0, T* x, long ldx, hipStream_t stream) {
  const long n = (long) (a.nt_r - 1) * a.nb + a.last_rows;
  const long total = ldx * a.ncols;
  if (total <= 0)
    return;
  const unsigned grid = (unsigned) std::min<long>((total + kThreads - 1) / kThreads, 4096);
  hipLaunchKernelGGL((hemm_reduce_kernel<T>), dim3(grid), dim3(kThreads), 0, stream, a, r0, n, x, ldx);
}

template <class T>
void launch_layers_reduce(const T* part, int layers, long rows, int ncols, T* out, long ldo, hipStream_t stream) {
  const long total = rows * ncols;
  if (total <= 0)
    return;
  const unsigned grid = (unsigned) std::min<long>((total + kThreads - 1) / kThreads, 4096);
  hipLaunchKernelGGL((layers_reduce_kernel<T>), dim3(grid), dim3(kThreads), 0, stream, part, layers, rows, ncols, out, ldo);
}

static long qr_spin_limit() {
  static const long v = [] {
    const char* e = std::getenv("DLAF_MI355X_QR_SPIN_LIMIT");
    return e ? std::atol(e) : kQrSpinLimit;
  }();
  return v;
}

// scratch layout: [counters: b + 1 words, padded to 256 B | head rows: b line-padded rows | partial-sum slots]
static size_t qr_line_pad(size_t bytes) {
  return (bytes + 127) / 128 * 128;
}
static constexpr size_t kQrPartialBudget = 96u << 20;  // bytes of partial-sum slots
static size_t qr_counter_bytes(int b) {
  return (((size_t) (b + 1) * sizeof(unsigned)) + 255) / 256 * 256;
}
size_t panel_qr_scratch_bytes(int b, size_t elem_size) {
  const size_t head = (size_t) b * qr_line_pad((size_t) b * elem_size);
  const size_t one_slot = qr_line_pad((size_t) kQrMaxWg * (size_t) b * elem_size);
  const size_t slots = std::max<size_t>(2, std::min<size_t>((size_t) b, kQrPartialBudget / one_slot));
  return qr_counter_bytes(b) + head + slots * one_slot + 256;
}

template <class T>
void launch_panel_qr(T* qt, long m, int b, int nr, T* taus, void* scratch, int* info, hipStream_t stream) {
  if (m <= 0 || b <= 0 || nr <= 0)
    return;
  if (b > kQrMaxColsPerThread * kThreads) {
    fprintf(stderr, "[dlaf_mi355x] panel QR: band size %d exceeds the supported %d\n", b, kQrMaxColsPerThread * kThreads);
    abort();
  }
  unsigned* counters = static_cast<unsigned*>(scratch);
  T* heads = reinterpret_cast<T*>(static_cast<char*>(scratch) + qr_counter_bytes(b));
  const size_t head_bytes = qr_line_pad((size_t) b * sizeof(T));
  T* partial = reinterpret_cast<T*>(reinterpret_cast<char*>(heads) + (size_t) b * head_bytes);
  (void) hipMemsetAsync(counters, 0, qr_counter_bytes(b), stream);
  const size_t vec_bytes = (size_t) b * sizeof(T);
  static const int max_wg = [] {
    const char* e = std::getenv("DLAF_MI355X_QR_MAXWG");  // tuning / debugging: 1 = the whole panel in one workgroup
    const int v = e ? std::atoi(e) : kQrDefaultWg;
    return std::max(1, std::min(v, kQrMaxWg));
  }();
  // 1024-thread workgroups for real panels (eight row groups at b = 128: a thread then has few rows, and the pass is
  // bound by the latency of its one or two batches of loads), 256 threads for narrow ones
  const bool big = b >= 64;
  const int nt = big ? 1024 : kThreads;
  QrMap mp;
  int ct = 1;
  while (ct < b && ct < nt)
    ct *= 2;
  if (big && ct > 256)
    ct = 256;  // two columns per thread beyond 256
  mp.ct = ct;
  static const long rows_target = [] {
    const char* e = std::getenv("DLAF_MI355X_QR_ROWS");  // rows per workgroup of a large panel
    return e ? std::max(16L, std::atol(e)) : 128L;
  }();
  long rows_per_wg = std::max<long>(rows_target, (m + max_wg - 1) / max_wg);
  rows_per_wg = ((rows_per_wg + 15) / 16) * 16;
  mp.rows_per_wg = (int) rows_per_wg;
  mp.nwg = (int) ((m + rows_per_wg - 1) / rows_per_wg);
  mp.spin_limit = qr_spin_limit();
  {
    const size_t slot_bytes = qr_line_pad((size_t) mp.nwg * vec_bytes);
    const size_t one_slot_max = qr_line_pad((size_t) kQrMaxWg * vec_bytes);
    const size_t slots_cap = std::max<size_t>(2, std::min<size_t>((size_t) b, kQrPartialBudget / one_slot_max));
    mp.nslots = (int) std::min<size_t>((size_t) std::max(nr, 2), std::max<size_t>(2, slots_cap * one_slot_max / slot_bytes));
    mp.slot_elems = (long) (slot_bytes / sizeof(T));
    mp.head_elems = (long) (head_bytes / sizeof(T));
  }
  // 16-byte reads of the partial-sum vectors: whole words per vector, a whole number of vectors per load round
  const int W = (int) (vec_bytes / 16);
  const int nrg = nt / ct;
  mp.wide = (vec_bytes % 16 == 0 && W >= 1 && W <= nt && nt % W == 0 && nt / W <= std::max(nrg, 8) * (b > ct ? 1 : 1) &&
             nt / W <= 16)
                ? 1
                : 0;
  const int redrows = std::max(std::max(nrg, 8), mp.wide ? nt / W : 0);
  const size_t lds = ((size_t) redrows * b + b + 4 + 4 * (size_t) rows_per_wg) * sizeof(T);
  if (lds > 150 * 1024) {
    fprintf(stderr, "[dlaf_mi355x] panel QR: %zu bytes of LDS for a %ld x %d panel\n", lds, m, b);
    abort();
  }
  if (m * (long) b >= (1L << 31)) {
    fprintf(stderr, "[dlaf_mi355x] panel QR: a %ld x %d panel exceeds the 32-bit element range of the kernel\n", m, b);
    abort();
  }
  auto go = [&](auto cpt, auto ntag) {
    constexpr int C = decltype(cpt)::value;
    constexpr int N = decltype(ntag)::value;
    hipLaunchKernelGGL((panel_qr_kernel<T, C, N>), dim3((unsigned) mp.nwg), dim3(N), lds, stream, qt, m, b, nr, taus, partial,
                       heads, counters, info, mp);
  };
  if (big) {
    if (b <= ct)
      go(std::integral_constant<int, 1>{}, std::integral_constant<int, 1024>{});
    else
      go(std::integral_constant<int, 2>{}, std::integral_constant<int, 1024>{});
  }
  else
    go(std::integral_constant<int, 1>{}, std::integral_constant<int, kThreads>{});
}

template <class T>
void launch_panel_move(T* tiles, long ltr, int nb, int il0, int il1, int jl, int pr, int ri, int nt, int last_rows, int cc,
                       int b, T* qt, long e0, long r0, bool to_panel, hipStream_t stream) {
  if (il1 <= il0 || b <= 0)
    return;
  dim3 grid((unsigned) ((b + kTR - 1) / kTR), (unsigned) ((nb + kTR - 1) / kTR), (unsigned) (il1 - il0));
  hipLaunchKernelGGL((panel_move_kernel<T>), grid, dim3(kThreads), 0, stream, tiles, ltr, nb, il0, jl, pr, ri, nt, last_rows,
                     cc, b, qt, e0, r0, to_panel ? 1 : 0);
}

template <class T>
void launch_make_v(const T* qt, int b, int nr, long e0, long r0, long n, T* v, long ldv, hipStream_t stream) {
  const long me = n - e0;
  if (me <= 0 || b <= 0)
    return;
  dim3 grid((unsigned) ((b + kTR - 1) / kTR), (unsigned) ((me + kTR - 1) / kTR), 1);
  hipLaunchKernelGGL((make_v_kernel<T>), grid, dim3(kThreads), 0, stream, qt, b, nr, e0, r0, n, v, ldv);
}

template <class T>
void launch_tfactor(const T* s, long lds_, const T* taus, int k, T* t, long ldt, hipStream_t stream, int batch, long bs,
                    long btau, long bt) {
  if (k <= 0 || batch <= 0)
    return;
  if (k > kTfMax) {
    fprintf(stderr, "[dlaf_mi355x] T factor: %d reflectors exceed the supported %d\n", k, kTfMax);
    abort();
  }
  const int rows_per_wg = kThreads / 64;
  auto go = [&](auto utag, auto dtag) {
    hipLaunchKernelGGL((tfactor_kernel<T, decltype(utag)::value, decltype(dtag)::value>),
                       dim3((unsigned) ((k + rows_per_wg - 1) / rows_per_wg), (unsigned) batch), dim3(kThreads),
                       (size_t) rows_per_wg * 2 * k * sizeof(T), stream, s, lds_, taus, k, t, ldt, bs, btau, bt);
  };
  using std::integral_constant;
  if (k <= 64)
    go(integral_constant<int, 1>{}, integral_constant<int, 8>{});
  else if (k <= 128)
    go(integral_constant<int, 2>{}, integral_constant<int, 8>{});
  else if (k <= 256)
    go(integral_constant<int, 4>{}, integral_constant<int, 4>{});
  else
    go(integral_constant<int, 8>{}, integral_constant<int, 2>{});
}

template <class T>
void launch_zero_rows(T* x, long ldx, long nrows, int ncols, hipStream_t stream) {
  const long total = nrows * ncols;
  if (total <= 0)
    return;
  const unsigned grid = (unsigned) std::min<long>((total + kThreads - 1) / kThreads, 2048);
  hipLaunchKernelGGL((zero_rows_kernel<T>), dim3(grid), dim3(kThreads), 0, stream, x, ldx, nrows, ncols);
}

template <class T>
static void band_init_one() {
  using Cfg = typename GenCfg<T>::type;
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_panel_kernel<T, 0>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_panel_kernel<T, 1>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_panel_kernel<T, 2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&panel_qr_kernel<T, 1, kThreads>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&panel_qr_kernel<T, 1, 1024>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&panel_qr_kernel<T, 2, 1024>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

void band_kernels_init() {
  band_init_one<float>();
  band_init_one<double>();
  band_init_one<cfloat>();
  band_init_one<cdouble>();
}

#define INST(T)                                                                                                        \
  template void launch_gemm<T>(const GemmArgs<T>&, hipStream_t);                                                       \
  template size_t gemm_partial_elems<T>(int, int, int);                                                                \
  template int gemm_pick_ksplit<T>(int, int, long);                                                                    \
  template void launch_tile_panel<T>(TilePanelArgs<T>&, hipStream_t);                                            \
  template void launch_hemm_reduce<T>(const TilePanelArgs<T>&, long, T*, long, hipStream_t);                           \
  template void launch_layers_reduce<T>(const T*, int, long, int, T*, long, hipStream_t);                              \
  template void launch_panel_qr<T>(T*, long, int, int, T*, void*, int*, hipStream_t);                                  \
  template void launch_panel_move<T>(T*, long, int, int, int, int, int, int, int, int, int, int, T*, long, long, bool, \
                                     hipStream_t);                                                                     \
  template void launch_make_v<T>(const T*, int, int, long, long, long, T*, long, hipStream_t);                         \
  template void launch_tfactor<T>(const T*, long, const T*, int, T*, long, hipStream_t, int, long, long, long);        \
  template void launch_zero_rows<T>(T*, long, long, int, hipStream_t);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
