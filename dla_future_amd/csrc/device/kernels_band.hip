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

#include "band_api.hpp"
#include "device_api.hpp"
#include "gemm_general.hpp"

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

template <class T>
__global__ __launch_bounds__(GenCfg<T>::type::THREADS, 2) void gemm_kernel(GemmArgs<T> p, GemmMap mp) {
  using Cfg = typename GenCfg<T>::type;
  using R = real_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);
  const int bid = blockIdx.x;
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
    gemm_acc<Cfg, T>(da, mrows, db, ncols, kk, lds, acc);
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

template <class T>
__global__ __launch_bounds__(kThreads) void gemm_reduce_kernel(GemmArgs<T> p, GemmMap mp) {
  using Cfg = typename GenCfg<T>::type;
  const int bm = blockIdx.x % mp.MB, bn = blockIdx.x / mp.MB;
  const int m0 = bm * Cfg::BM, n0 = bn * Cfg::BN;
  const int mrows = min(Cfg::BM, p.M - m0), ncols = min(Cfg::BN, p.N - n0);
  const bool has_beta = !el_is_zero(p.beta);
  for (int e = threadIdx.x; e < Cfg::BM * Cfg::BN; e += kThreads) {
    const int m = e % Cfg::BM, n = e / Cfg::BM;
    if (m >= mrows || n >= ncols)
      continue;
    T s = zero_el<T>();
    for (int ks = 0; ks < mp.KS; ++ks)
      s = el_add(s, p.partial[((size_t) (ks * mp.NB + bn) * mp.MB + bm) * (size_t) (Cfg::BM * Cfg::BN) + e]);
    T* c = p.c + (m0 + m) + (long) (n0 + n) * p.ldc;
    T r = el_mul(p.alpha, s);
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
};

template <class T>
__global__ __launch_bounds__(GenCfg<T>::type::THREADS, 1) void tile_panel_kernel(TilePanelArgs<T> p, TilePanelMap mp) {
  using Cfg = typename GenCfg<T>::type;
  using R = real_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);
  long id = blockIdx.x;
  const bool kind_t = id >= mp.cnt_s;
  if (kind_t)
    id -= mp.cnt_s;
  const int cb = (int) (id % mp.cbn);
  id /= mp.cbn;
  const int s = (int) (id % mp.sub);
  id /= mp.sub;
  const int ot = kind_t ? (p.jl1 - p.jl0) : (p.il1 - p.il0);
  const int t = (int) (id % ot);
  const int q = (int) (id / ot);
  const int layers = kind_t ? p.layers_t : p.layers_s;
  const long te = (long) p.nb * p.nb;
  const int n0 = cb * Cfg::BN;
  const int ncols = min(Cfg::BN, p.ncols - n0);
  Acc<Cfg> acc;
  acc.clear();
  // the output tile: local tile row il (kind S) / local tile column jl (kind T); its sources: the tiles of that
  // tile row left of (and on) the diagonal / of that tile column below the diagonal, dealt out to the layers
  const int go = kind_t ? (p.jl0 + t) * p.pc + p.ci : (p.il0 + t) * p.pr + p.ri;
  const int ext_o = kind_t ? ((go == p.nt_c - 1) ? p.last_cols : p.nb) : ((go == p.nt_r - 1) ? p.last_rows : p.nb);
  const int mrows = min(Cfg::BM, ext_o - s * Cfg::BM);
  if (mrows > 0) {
    int first = kind_t ? p.il0 : p.jl0;
    const int end = kind_t ? p.il1 : p.jl1;
    if (kind_t && p.herm && go >= p.ri)
      first = max(p.il0, (go - p.ri) / p.pr + 1);
    for (int src = first + q; src < end; src += layers) {
      const int il = kind_t ? src : p.il0 + t;
      const int jl = kind_t ? p.jl0 + t : src;
      const int gi = il * p.pr + p.ri, gj = jl * p.pc + p.ci;
      if (!kind_t && p.herm && gj > gi)
        break;
      // the tile the panel rows come from: global tile column gj (kind S) / global tile row gi (kind T)
      const int gk = kind_t ? gi : gj;
      const int kext = kind_t ? ((gi == p.nt_r - 1) ? p.last_rows : p.nb) : ((gj == p.nt_c - 1) ? p.last_cols : p.nb);
      const T* tile = p.tiles + ((long) il + (long) jl * p.ltr) * te;
      OpDesc<T> da, db;
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
      db.p = p.w + ((long) gk * p.nb - p.e0) + (long) n0 * p.ldw;
      db.rs = p.ldw;
      db.ks = 1;
      db.conj = 1;
      gemm_acc<Cfg, T>(da, mrows, db, ncols, kext, lds, acc);
    }
  }
  if (mrows <= 0)
    return;
  const long ldp = (long) ot * p.nb;
  T* part = (kind_t ? p.part_t : p.part_s) + (size_t) q * (size_t) p.ncols * (size_t) ldp + (long) t * p.nb + s * Cfg::BM;
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
        if (il >= p.il0 && il < p.il1)
          for (int q = 0; q < p.layers_s; ++q)
            v = el_add(v, p.part_s[((size_t) q * p.ncols + c) * (size_t) lds_ + (long) (il - p.il0) * p.nb + r]);
      }
      if ((p.kinds & 2) && gt >= p.ci && (gt - p.ci) % p.pc == 0) {
        const int jl = (gt - p.ci) / p.pc;
        if (jl >= p.jl0 && jl < p.jl1)
          for (int q = 0; q < p.layers_t; ++q)
            v = el_add(v, p.part_t[((size_t) q * p.ncols + c) * (size_t) ldt + (long) (jl - p.jl0) * p.nb + r]);
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
constexpr int kTfMax = 1024;

// one workgroup; t_j = T(0:j, 0:j) * (-tau_j S(0:j, j)), T(j, j) = tau_j (t_factor_impl.h:60-131)
template <class T>
__global__ __launch_bounds__(kThreads) void tfactor_kernel(const T* s, long lds_, const T* taus, int k, T* t, long ldt) {
  __shared__ T tmp[kTfMax];
  for (int j = 0; j < k; ++j) {
    const T tau = taus[j];
    for (int i = threadIdx.x; i < j; i += kThreads)
      tmp[i] = el_neg(el_mul(tau, s[i + (long) j * lds_]));
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += kThreads) {
      T a = zero_el<T>();
      if (i < j) {
        for (int l = i; l < j; ++l)
          a = el_add(a, el_mul(t[i + (long) l * ldt], tmp[l]));
      }
      else if (i == j)
        a = tau;
      t[i + (long) j * ldt] = a;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ======================================================================================= panel QR
constexpr int kQrMaxWg = 128;
constexpr int kQrMaxColsPerThread = 2;  // b <= 512
constexpr long kQrSpinLimit = 20000000;

struct QrMap {
  int nwg;
  int rows_per_wg;
  int ct;      // column threads (power of two)
  int chunk;   // rows per LDS chunk
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
//   accumulating the partial sums of step j + 1 in the same pass.
// Hand-offs: write-through stores + drained flag / relaxed poll + one acquire (the protocol of the tile POTRF).
template <class T>
__global__ __launch_bounds__(kThreads) void panel_qr_kernel(T* qt, long m, int b, int nr, T* taus, T* partial,
                                                            unsigned* counters, int* info, QrMap mp) {
  using R = real_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T* chunk = reinterpret_cast<T*>(lds_raw);                 // [chunk rows][b]
  T* red = chunk + (size_t) mp.chunk * b;                   // [rg][b]
  T* wv = red + (size_t) max(kThreads / mp.ct, 2) * b;    // [b]  (red doubles as [d | h] in the exchange)
  T* sc = wv + b;                                           // tau, scale, y
  __shared__ unsigned flag_slot;
  const int g = blockIdx.x;
  const long row_lo = (long) g * mp.rows_per_wg;
  const long row_hi = min(m, row_lo + mp.rows_per_wg);
  const int tc = threadIdx.x % mp.ct, rg = threadIdx.x / mp.ct, nrg = kThreads / mp.ct;
  T dnext[kQrMaxColsPerThread];

  // one pass over the rows of this workgroup: apply reflector j (j >= 0) and sum the dots of column j + 1
  auto pass = [&](int j, const T ctau, const T scale, const T y) {
#pragma unroll
    for (int u = 0; u < kQrMaxColsPerThread; ++u)
      dnext[u] = zero_el<T>();
    const int jn = j + 1;
    const bool dots = jn < nr;
    const long first = max(row_lo, (long) max(j, 0));
    for (long rc = first; rc < row_hi; rc += mp.chunk) {
      const int nrows = (int) min((long) mp.chunk, row_hi - rc);
      // old values of the chunk -> LDS (columns >= max(j, 0))
      for (int e = threadIdx.x; e < nrows * b; e += kThreads)
        chunk[e] = qt[rc * b + e];
      __syncthreads();
      for (int rr = rg; rr < nrows; rr += nrg) {
        const long r = rc + rr;
        const T* row = chunk + (size_t) rr * b;
        T vr = zero_el<T>();
        T xn = zero_el<T>();
        if (j >= 0) {
          vr = (r == j) ? make_el<T>(R(1), R(0)) : el_mul(row[j], scale);
          if (dots)
            xn = el_sub(row[jn], el_mul(ctau, el_mul(vr, el_conj(wv[jn]))));
        }
        else if (dots)
          xn = row[jn];
#pragma unroll
        for (int u = 0; u < kQrMaxColsPerThread; ++u) {
          const int c = tc + u * mp.ct;
          if (c >= b)
            break;
          if (j >= 0 && c < j)
            continue;
          T nv = row[c];
          if (j >= 0) {
            if (c == j)
              nv = (r == j) ? y : vr;
            else
              nv = el_sub(nv, el_mul(ctau, el_mul(vr, el_conj(wv[c]))));
            // the next head row is read by every workgroup after the next exchange
            if (r == jn)
              qr_store_wt(&qt[r * b + c], nv);
            else
              qt[r * b + c] = nv;
          }
          if (dots && c >= jn && r >= jn)
            dnext[u] = el_add(dnext[u], el_mul(el_conj(nv), xn));
        }
      }
      __syncthreads();
    }
  };

  // combine the row groups and publish this workgroup's partial sums of column jn
  auto publish = [&](int jn) {
#pragma unroll
    for (int u = 0; u < kQrMaxColsPerThread; ++u) {
      const int c = tc + u * mp.ct;
      if (c < b)
        red[(size_t) rg * b + c] = dnext[u];
    }
    __syncthreads();
    T* mine = partial + ((size_t) (jn & 1) * mp.nwg + g) * (size_t) b;
    for (int c = threadIdx.x; c < b; c += kThreads) {
      if (c < jn)
        continue;
      T sum = red[c];
      for (int q = 1; q < nrg; ++q)
        sum = el_add(sum, red[(size_t) q * b + c]);
      qr_store_wt(&mine[c], sum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add(&counters[jn], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  pass(-1, zero_el<T>(), zero_el<T>(), zero_el<T>());
  if (nr > 0)
    publish(0);
  for (int j = 0; j < nr; ++j) {
    // ---- exchange: wait for every workgroup's share of column j ------------------------------------------
    if (threadIdx.x == 0) {
      unsigned v;
      long spins = 0;
      while ((v = __hip_atomic_load(&counters[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned) mp.nwg) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > mp.spin_limit || __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          v = 0xFFFFFFFFu;
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flag_slot = v;
    }
    __syncthreads();
    if (flag_slot == 0xFFFFFFFFu) {
      if (threadIdx.x == 0)
        atomicCAS(info, 0, kInfoSchedulingFailure);
      return;
    }
    // ---- totals, head row, the reflector's scalars, w ---------------------------------------------------------
    const T* part = partial + (size_t) (j & 1) * mp.nwg * (size_t) b;
    for (int c = threadIdx.x; c < b; c += kThreads) {
      if (c < j)
        continue;
      T d = zero_el<T>();
      for (int q = 0; q < mp.nwg; ++q)
        d = el_add(d, part[(size_t) q * b + c]);
      red[c] = d;                   // d_c
      red[b + c] = qt[(long) j * b + c];  // h_c  (nrg >= 1 rows of `red`: needs 2 b elements, see launcher)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const T x0 = red[b + j];
      const R nrm2 = re_of(red[j]);
      T tau = zero_el<T>(), scale = zero_el<T>(), y = x0;
      if (nrm2 != R(0)) {
        const R nrm = sqrt(nrm2);
        const R yr = __builtin_signbit(re_of(x0)) ? nrm : -nrm;
        y = make_el<T>(yr, R(0));
        tau = el_scale(el_sub(y, x0), R(1) / yr);
        scale = el_inv(el_sub(x0, y));
      }
      sc[0] = tau;
      sc[1] = scale;
      sc[2] = y;
      if (g == 0)
        taus[j] = tau;
    }
    __syncthreads();
    const T tau = sc[0], scale = sc[1], y = sc[2];
    {
      const T x0 = red[b + j];
      for (int c = threadIdx.x; c < b; c += kThreads) {
        if (c <= j)
          continue;
        const T hc = el_conj(red[b + c]);
        wv[c] = el_add(hc, el_mul(scale, el_sub(red[c], el_mul(hc, x0))));
      }
    }
    __syncthreads();
    // (a reflector with tau == 0 is the identity: the reference leaves the column as it is, impl.h:108-109)
    const bool ident = el_is_zero(tau);
    if (!ident || j + 1 < nr) {
      if (ident) {
        // no update, but the dots of the next column are still needed: a pass that changes nothing
        pass(j, zero_el<T>(), make_el<T>(R(1), R(0)), red[b + j]);
      }
      else
        pass(j, el_conj(tau), scale, y);
    }
    if (j + 1 < nr)
      publish(j + 1);
  }
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
  if (blocks >= 256 || K < 4 * 512)
    return 1;
  long ks = std::min<long>((512 + blocks - 1) / blocks, K / 512);
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
  mp.KS = (a.ksplit > 1 && a.partial != nullptr && a.K > 0) ? a.ksplit : 1;
  int kchunk = (a.K + mp.KS - 1) / mp.KS;
  kchunk = ((kchunk + Cfg::BK - 1) / Cfg::BK) * Cfg::BK;
  if (kchunk <= 0)
    kchunk = Cfg::BK;
  mp.kchunk = kchunk;
  if (mp.KS > 1)
    mp.KS = (a.K + kchunk - 1) / kchunk;
  hipLaunchKernelGGL((gemm_kernel<T>), dim3((unsigned) (mp.MB * mp.NB * mp.KS)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream,
                     a, mp);
  if (mp.KS > 1)
    hipLaunchKernelGGL((gemm_reduce_kernel<T>), dim3((unsigned) (mp.MB * mp.NB)), dim3(kThreads), 0, stream, a, mp);
}

int tile_panel_pick_layers(long out_tiles, int nb, int ncols, long max_src, size_t elem_size) {
  if (out_tiles <= 0 || max_src <= 1)
    return 1;
  const int bm = 128, bn = elem_size == 16 ? 64 : 128;
  const long per_layer = out_tiles * ((nb + bm - 1) / bm) * ((ncols + bn - 1) / bn);
  long layers = (768 + per_layer - 1) / per_layer;
  layers = std::max<long>(1, std::min<long>(layers, max_src));
  return (int) std::min<long>(layers, 64);
}

template <class T>
void launch_tile_panel(const TilePanelArgs<T>& a, hipStream_t stream) {
  using Cfg = typename GenCfg<T>::type;
  if (a.il1 <= a.il0 || a.jl1 <= a.jl0 || a.ncols <= 0)
    return;
  TilePanelMap mp;
  mp.sub = (a.nb + Cfg::BM - 1) / Cfg::BM;
  mp.cbn = (a.ncols + Cfg::BN - 1) / Cfg::BN;
  mp.cnt_s = (a.kinds & 1) ? (long) a.layers_s * (a.il1 - a.il0) * mp.sub * mp.cbn : 0;
  mp.cnt_t = (a.kinds & 2) ? (long) a.layers_t * (a.jl1 - a.jl0) * mp.sub * mp.cbn : 0;
  if (mp.cnt_s + mp.cnt_t == 0)
    return;
  hipLaunchKernelGGL((tile_panel_kernel<T>), dim3((unsigned) (mp.cnt_s + mp.cnt_t)), dim3(Cfg::THREADS), Cfg::LDS_BYTES,
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

size_t panel_qr_scratch_bytes(int b, size_t elem_size) {
  // partial sums [2][kQrMaxWg][b] + one counter per reflector
  return 2 * (size_t) kQrMaxWg * (size_t) b * elem_size + (size_t) (b + 1) * sizeof(unsigned) + 64;
}

template <class T>
void launch_panel_qr(T* qt, long m, int b, int nr, T* taus, void* scratch, int* info, hipStream_t stream) {
  if (m <= 0 || b <= 0 || nr <= 0)
    return;
  if (b > kQrMaxColsPerThread * kThreads) {
    fprintf(stderr, "[dlaf_mi355x] panel QR: band size %d exceeds the supported %d\n", b, kQrMaxColsPerThread * kThreads);
    abort();
  }
  QrMap mp;
  int ct = 1;
  while (ct < b && ct < kThreads)
    ct *= 2;
  mp.ct = ct;
  // LDS chunk of about 32 KiB
  int chunk = (int) std::max<size_t>(1, (32 * 1024) / ((size_t) b * sizeof(T)));
  chunk = std::min(chunk, 64);
  const int nrg = kThreads / ct;
  chunk = std::max(chunk, nrg);
  mp.chunk = chunk;
  static const int max_wg = [] {
    const char* e = std::getenv("DLAF_MI355X_QR_MAXWG");  // tuning / debugging: 1 = the whole panel in one workgroup
    const int v = e ? std::atoi(e) : kQrMaxWg;
    return std::max(1, std::min(v, kQrMaxWg));
  }();
  long rows_per_wg = std::max<long>(64, (m + max_wg - 1) / max_wg);
  rows_per_wg = ((rows_per_wg + chunk - 1) / chunk) * chunk;
  mp.rows_per_wg = (int) rows_per_wg;
  mp.nwg = (int) ((m + rows_per_wg - 1) / rows_per_wg);
  mp.spin_limit = qr_spin_limit();
  T* partial = static_cast<T*>(scratch);
  unsigned* counters = reinterpret_cast<unsigned*>(static_cast<char*>(scratch) + 2 * (size_t) kQrMaxWg * (size_t) b * sizeof(T));
  (void) hipMemsetAsync(counters, 0, (size_t) (b + 1) * sizeof(unsigned), stream);
  // LDS: chunk + red (max(nrg, 2) rows of b) + w + 4 scalars
  const size_t lds = ((size_t) chunk * b + (size_t) std::max(nrg, 2) * b + b + 4) * sizeof(T);
  hipLaunchKernelGGL((panel_qr_kernel<T>), dim3((unsigned) mp.nwg), dim3(kThreads), lds, stream, qt, m, b, nr, taus, partial,
                     counters, info, mp);
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
void launch_tfactor(const T* s, long lds_, const T* taus, int k, T* t, long ldt, hipStream_t stream) {
  if (k <= 0)
    return;
  if (k > kTfMax) {
    fprintf(stderr, "[dlaf_mi355x] T factor: %d reflectors exceed the supported %d\n", k, kTfMax);
    abort();
  }
  hipLaunchKernelGGL((tfactor_kernel<T>), dim3(1), dim3(kThreads), 0, stream, s, lds_, taus, k, t, ldt);
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
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_panel_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&panel_qr_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             96 * 1024);
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
  template void launch_tile_panel<T>(const TilePanelArgs<T>&, hipStream_t);                                            \
  template void launch_hemm_reduce<T>(const TilePanelArgs<T>&, long, T*, long, hipStream_t);                           \
  template void launch_layers_reduce<T>(const T*, int, long, int, T*, long, hipStream_t);                              \
  template void launch_panel_qr<T>(T*, long, int, int, T*, void*, int*, hipStream_t);                                  \
  template void launch_panel_move<T>(T*, long, int, int, int, int, int, int, int, int, int, int, T*, long, long, bool, \
                                     hipStream_t);                                                                     \
  template void launch_make_v<T>(const T*, int, int, long, long, long, T*, long, hipStream_t);                         \
  template void launch_tfactor<T>(const T*, long, const T*, int, T*, long, hipStream_t);                               \
  template void launch_zero_rows<T>(T*, long, long, int, hipStream_t);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
