// kernels_hr.hip -- the small kernels of the BLOCKED panel factorization of reduction_to_band (band_api.hpp,
// "panel QR, blocked"): the reflectors of an m x b panel from three tall-skinny passes instead of b grid-wide
// exchanges.
//
// Reference: computePanelReflectors (include/dlaf/eigensolver/reduction_to_band/impl.h:297-361: xLARFG + a rank-one
// update per column, on the CPU even for the GPU backend).  Here, for real panels that are well conditioned:
//   1. Q R = P by CholeskyQR2 (G = P^T P, Cholesky, triangular solve; twice) on the Cholesky path's own kernels
//      (gemm with split K, the cooperative tile POTRF, the panel TRSM);
//   2. the Householder representation of that Q is RECONSTRUCTED (Ballard, Demmel, Grigori, Jacquelin, Nguyen,
//      Solomonik, "Reconstructing Householder vectors from TSQR", IPDPS 2014): with S = diag(-sign(q_jj)) chosen
//      during the elimination, the unit-lower factor of the LU factorization WITHOUT pivoting of Q - [S; 0] is exactly
//      the matrix V of xGEQR2's reflectors, tau_j = -u_jj s_j, T = -U S V1^-H, and R_geqr2 = S R -- the same numbers
//      LAPACK produces (sign convention beta = -sign(alpha) |x| included), to rounding (oracle: tests/test_gpu_red2band.py,
//      tests/test_oracle_red2band.py).  The pivots of that LU have magnitude >= 1 by construction: no growth.
// A panel whose Gram matrix is not safely positive definite (rank-deficient panels of an identity matrix, condition
// numbers beyond the gate) raises a flag and the caller runs the reflector-by-reflector kernel instead.
#include <cstdio>
#include <cstdlib>

#include "band_api.hpp"

namespace dlaf_mi355x {

namespace {

constexpr int kHrTile = 32;

// element arithmetic for real and complex T (cplx<R>: .re, .im)
template <class T>
__device__ __forceinline__ T h_mul(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
  else
    return a * b;
}
template <class T>
__device__ __forceinline__ T h_sub(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re - b.re, a.im - b.im};
  else
    return a - b;
}
template <class T>
__device__ __forceinline__ T h_div(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex) {
    const real_t<T> den = b.re * b.re + b.im * b.im;
    return T{(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
  }
  else
    return a / b;
}
// 1 / b for |b| >= 1 (the pivots of the reconstruction: |w_jj - s| >= 1): the hardware reciprocal and two Newton steps
// instead of the division's ~40 dependent instructions per thread and step -- within an ulp or two of the quotient
__device__ __forceinline__ double h_rcp_real(double p) {
  double r = __builtin_amdgcn_rcp(p);
  r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float h_rcp_real(float p) {
  float r = __builtin_amdgcn_rcpf(p);
  r = __builtin_fmaf(__builtin_fmaf(-p, r, 1.0f), r, r);
  return r;
}
template <class T>
__device__ __forceinline__ T h_rcp(const T& b) {
  if constexpr (TypeInfo<T>::is_complex) {
    const real_t<T> rd = h_rcp_real(b.re * b.re + b.im * b.im);
    return T{b.re * rd, -b.im * rd};
  }
  else
    return h_rcp_real(b);
}
template <class T>
__device__ __forceinline__ T h_conj(const T& a) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re, -a.im};
  else
    return a;
}
template <class T>
__device__ __forceinline__ T h_scale(const T& a, real_t<T> f) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re * f, a.im * f};
  else
    return a * f;
}

// dst[r + c * ldd] = src[c + r * b]  (to_cm != 0: transposed panel -> column-major), or back.  `flag` (may be null):
// nothing is written when *flag != 0.
template <class T>
__global__ __launch_bounds__(256) void hr_transpose_kernel(T* qt, int b, long m, T* cm, long ld, int to_cm, const int* flag) {
  __shared__ T tile[kHrTile][kHrTile + 1];
  if (flag != nullptr && *flag != 0)
    return;
  const long r0 = (long) blockIdx.x * kHrTile;
  const int c0 = (int) blockIdx.y * kHrTile;
  const int tx = threadIdx.x % kHrTile, ty = threadIdx.x / kHrTile;  // 32 x 8
  if (to_cm) {
    for (int i = ty; i < kHrTile; i += 8) {
      const long r = r0 + i;
      const int c = c0 + tx;
      if (r < m && c < b)
        tile[i][tx] = qt[c + r * b];
    }
    __syncthreads();
    for (int i = ty; i < kHrTile; i += 8) {
      const long r = r0 + tx;
      const int c = c0 + i;
      if (r < m && c < b)
        cm[r + (long) c * ld] = tile[tx][i];
    }
  }
  else {
    for (int i = ty; i < kHrTile; i += 8) {
      const long r = r0 + tx;
      const int c = c0 + i;
      if (r < m && c < b)
        tile[tx][i] = cm[r + (long) c * ld];
    }
    __syncthreads();
    for (int i = ty; i < kHrTile; i += 8) {
      const long r = r0 + i;
      const int c = c0 + tx;
      if (r < m && c < b)
        qt[c + r * b] = tile[i][tx];
    }
  }
}

// The gate between the first Cholesky factorization and everything built on it: the diagonal of the factor L of
// G = P^T P bounds the condition number of P from below (max / min of |l_jj|); CholeskyQR2 delivers a Q that is
// orthonormal to rounding while cond(P)^2 eps << 1.  Beyond `limit` (<= 0: no gate), or for a factor that is not
// finite, the flag is raised.  The strict upper triangle of l -- which the factorization leaves as it found it -- is
// zeroed, so that R = L2^T L1^T can be formed by the general product.  One workgroup.
template <class T>
__global__ __launch_bounds__(256) void hr_gate_kernel(T* l, int ld, int b, real_t<T> limit, int* flag) {
  using R = real_t<T>;
  if (*flag != 0)
    return;
  for (int idx = threadIdx.x; idx < b * b; idx += 256) {
    const int r = idx % b, c = idx / b;
    if (r < c)
      l[r + (long) c * ld] = zero_el<T>();
  }
  if (threadIdx.x >= 64)
    return;
  R mx = 0, mn = 0;
  bool bad = false, first = true;
  for (int j = threadIdx.x; j < b; j += 64) {
    const R d = re_of(l[j + (long) j * ld]);
    if (!(d > R(0)) || !(d < R(1e300)))
      bad = true;
    mx = first ? d : (d > mx ? d : mx);
    mn = first ? d : (d < mn ? d : mn);
    first = false;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const R omx = __shfl_xor(mx, off), omn = __shfl_xor(mn, off);
    const int ofirst = __shfl_xor(first ? 1 : 0, off);
    if (!ofirst) {
      mx = first ? omx : (omx > mx ? omx : mx);
      mn = first ? omn : (omn < mn ? omn : mn);
      first = false;
    }
    bad = bad || (__shfl_xor(bad ? 1 : 0, off) != 0);
  }
  if (threadIdx.x == 0 && (bad || (limit > R(0) && !(mx <= limit * mn))))
    atomicCAS(flag, 0, 1);
}

// After the first CholeskyQR pass: g2 = Q1^T Q1 (b x b, ld).  When it equals the identity to `tol` (max norm) the
// second pass would change nothing above rounding -- the panels of a random matrix have condition numbers close to 1
// and come out of ONE pass orthonormal to a few 1e-15 -- so g2 is replaced by the identity (its "Cholesky factor":
// R = L2^T L1^T = L1^T) and *skip is raised: the second factorization and solve, which take `skip` as their status
// word, return at once.  A g2 that is far from the identity (> 0.1) means the first pass failed outright: *flag.
template <class T>
__global__ __launch_bounds__(256) void hr_orth_kernel(T* g2, int ld, int b, real_t<T> tol, int* skip, int* flag) {
  using R = real_t<T>;
  __shared__ R red[256];
  if (*flag != 0)
    return;
  R mx = 0;
  for (int idx = threadIdx.x; idx < b * b; idx += 256) {
    const int r = idx % b, c = idx / b;
    if (r >= c) {
      const T e = g2[r + (long) c * ld];
      R v = re_of(e) - (r == c ? R(1) : R(0));
      v = v < 0 ? -v : v;
      R vi = im_of(e);
      vi = vi < 0 ? -vi : vi;
      v = (vi > v || !(vi == vi)) ? vi : v;
      mx = (v > mx || !(v == v)) ? v : mx;  // (a NaN wins)
    }
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int) threadIdx.x < off) {
      const R o = red[threadIdx.x + off];
      if (o > red[threadIdx.x] || !(o == o))
        red[threadIdx.x] = o;
    }
    __syncthreads();
  }
  const R m = red[0];
  if (!(m <= R(0.1))) {
    if (threadIdx.x == 0)
      atomicCAS(flag, 0, 2);
    return;
  }
  if (m <= tol) {
    for (int idx = threadIdx.x; idx < b * b; idx += 256) {
      const int r = idx % b, c = idx / b;
      g2[r + (long) c * ld] = make_el<T>((r == c) ? R(1) : R(0), R(0));
    }
    if (threadIdx.x == 0)
      *skip = 1;
  }
}

// One workgroup of NT threads.  In: the top b x b block of Q (column-major, ldq), R = L2^T L1^T (b x b, ld b, upper).
// Out:
//   top block of Q  <-  xGEQR2's output: S R on and above the diagonal, V1 strictly below it
//   lu   (b x b, ld b, lower)  = U^T: the solve of the rows below, V2 = Q2 U^-1, runs as X (U^T)^T = Q2 on the panel TRSM
//   y1   (b x b, ld b, lower, unit diagonal stored)  = V1
//   tb   (b x b, ld b, upper, strict lower part zero) = -U S: T = tb V1^-T by one more TRSM
//   taus[j] = -u_jj s_j
// The elimination keeps W = Q1 in REGISTERS: thread (column c = t % b, row group g = t / b) holds the rows g + G k of
// its column (G = NT / b row groups, b / G rows each).  Step j needs column j (the multipliers, unscaled) and row j of
// the current matrix: their owners put them into two small LDS vectors right after they have updated them in step
// j - 1 (double-buffered), so a step is ONE barrier, b / G + 1 broadcast reads and as many fused multiply-adds.
// Nothing is written into row j / column j during or after step j: the column is scaled at the end, and every thread
// forms the pivot u_jj = w_jj - s_j for itself.
template <class T, int NT, int RPT>
__global__ __launch_bounds__(NT) void hr_lu_kernel(T* q, long ldq, int b, const T* rmat, T* lu, T* y1, T* tb, T* taus,
                                                   const int* flag) {
  using R = real_t<T>;
  __shared__ T colbuf[2][128];
  __shared__ T rowbuf[2][128];
  __shared__ R sg[128];
  __shared__ T pv[128];
  if (*flag != 0)
    return;
  const int t = threadIdx.x;
  const int c = t % b, g = t / b, G = NT / b;  // (RPT == b / G)
  const int gs = __builtin_amdgcn_readfirstlane(g);                          // wave-uniform copies
  const int cmax = __builtin_amdgcn_readfirstlane(((t & ~63) % b) + 63);    // highest column of this wave
  T w[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k)
    w[k] = q[(g + G * k) + (long) c * ldq];
  if (c == 0) {
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      colbuf[0][g + G * k] = w[k];
  }
  if (g == 0)
    rowbuf[0][c] = w[0];
  __syncthreads();
  for (int j = 0; j < b; ++j) {
    const int par = j & 1;
    const T wjj = rowbuf[par][j];
    // (complex panels: the REAL sign -sign(Re w_jj) reproduces xLARFG's real beta, and |w_jj - s| >= 1 still holds)
    const R s = (re_of(wjj) >= R(0)) ? R(-1) : R(1);
    const T piv = make_el<T>(re_of(wjj) - s, im_of(wjj));
    if (t == 0) {
      sg[j] = s;
      pv[j] = piv;
    }
    // rows g + G k <= j and columns <= j are done.  Both tests are made WAVE-UNIFORM (a wave's 64 consecutive threads
    // share the row group, b >= 64, and a wave whose highest column is done has nothing to do at all), so they are
    // scalar branches around the reads and multiply-adds instead of per-lane selects on every element
    if (cmax > j) {
      const T u = (c > j) ? h_mul(rowbuf[par][c], h_rcp(piv)) : zero_el<T>();  // u_jc / u_jj
      const int k0 = (j >= gs) ? (j - gs) / G + 1 : 0;                  // first row index of the thread that is still active
      // in chunks of up to 8 (complex: 4) rows: the chunk's multipliers are read together (one LDS latency per chunk), chunks whose
      // rows are all done are skipped by a scalar branch, the chunk that straddles the boundary masks per element
      constexpr int CHW = sizeof(T) == 16 ? 4 : 8;  // (16 complex rows are half the register file of a 1024-thread group)
      constexpr int CH = RPT < CHW ? RPT : CHW;
#pragma unroll
      for (int kk = 0; kk < RPT; kk += CH) {
        if (kk + CH > k0) {
          T l[CH];
#pragma unroll
          for (int e = 0; e < CH; ++e)
            l[e] = colbuf[par][gs + G * (kk + e)];
#pragma unroll
          for (int e = 0; e < CH; ++e) {
            const bool on = kk + e >= k0;
            const T lm = make_el<T>(on ? re_of(l[e]) : R(0), on ? im_of(l[e]) : R(0));
            if constexpr (TypeInfo<T>::is_complex)
              w[kk + e] = T{__builtin_fma(lm.im, u.im, __builtin_fma(-lm.re, u.re, w[kk + e].re)),
                            __builtin_fma(-lm.im, u.re, __builtin_fma(-lm.re, u.im, w[kk + e].im))};
            else
              w[kk + e] = __builtin_fma(-lm, u, w[kk + e]);
          }
        }
      }
    }
    // hand column j + 1 and row j + 1 to the next step
    const int jn = j + 1;
    if (jn < b) {
      if (c == jn) {
#pragma unroll
        for (int k = 0; k < RPT; ++k)
          colbuf[par ^ 1][g + G * k] = w[k];
      }
      if (g == jn % G) {
        const int kn = jn / G;
        T v = w[0];
#pragma unroll
        for (int k = 1; k < RPT; ++k)
          v = make_el<T>((k == kn) ? re_of(w[k]) : re_of(v), (k == kn) ? im_of(w[k]) : im_of(v));
        rowbuf[par ^ 1][c] = v;
      }
    }
    __syncthreads();
  }
  // ---- outputs ----------------------------------------------------------------------------------------------------------
  const T pc = pv[c];
  const R sc = sg[c];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int r = g + G * k;
    T yv = zero_el<T>(), uv = zero_el<T>();
    if (r > c)
      yv = h_div(w[k], pc);  // V1, strictly lower
    else if (r == c)
      uv = pc;
    else
      uv = w[k];             // U, strictly upper
    y1[r + (long) c * b] = r > c ? yv : make_el<T>(r == c ? R(1) : R(0), R(0));
    lu[c + (long) r * b] = r <= c ? h_conj(uv) : zero_el<T>();           // element (c, r) of U^H
    tb[r + (long) c * b] = r <= c ? h_scale(uv, -sc) : zero_el<T>();     // -U S: column c scaled by s_c
    q[r + (long) c * ldq] = r > c ? yv : h_scale(rmat[r + (long) c * b], sg[r]);
    if (r == c)
      taus[c] = h_scale(pc, -sc);
  }
}

}  // namespace

bool panel_qr_blocked_supported(int b, long m, int nr, size_t elem_size, bool is_complex) {
  static const bool on = [] {
    const char* e = std::getenv("DLAF_MI355X_QR_BLOCKED");
    return e ? std::atoi(e) != 0 : true;
  }();
  // double-precision panels (real and complex) of band 64 / 128 with a full set of reflectors and at least 2 b rows;
  // everything else (single precision, the last panels of a matrix, other bands) keeps the reflector-by-reflector kernel
  return on && ((!is_complex && elem_size == 8) || (is_complex && elem_size == 16)) && (b == 64 || b == 128) && nr == b && m >= 2L * b;
}

template <class T>
void launch_hr_transpose(T* qt, int b, long m, T* cm, long ld, bool to_cm, const int* flag, hipStream_t stream) {
  if (m <= 0 || b <= 0)
    return;
  dim3 grid((unsigned) ((m + kHrTile - 1) / kHrTile), (unsigned) ((b + kHrTile - 1) / kHrTile), 1);
  hipLaunchKernelGGL((hr_transpose_kernel<T>), grid, dim3(256), 0, stream, qt, b, m, cm, ld, to_cm ? 1 : 0, flag);
}

template <class T>
void launch_hr_gate(T* l, int ld, int b, double limit, int* flag, hipStream_t stream) {
  hipLaunchKernelGGL((hr_gate_kernel<T>), dim3(1), dim3(256), 0, stream, l, ld, b, (real_t<T>) limit, flag);
}

template <class T>
void launch_hr_orth(T* g2, int ld, int b, double tol, int* skip, int* flag, hipStream_t stream) {
  hipLaunchKernelGGL((hr_orth_kernel<T>), dim3(1), dim3(256), 0, stream, g2, ld, b, (real_t<T>) tol, skip, flag);
}

template <class T>
void launch_hr_lu(T* q, long ldq, int b, const T* rmat, T* lu, T* y1, T* tb, T* taus, const int* flag, hipStream_t stream) {
  constexpr int NT = 1024;
  if (b == 128) {
    // (complex double: 16 rows of a column are 64 registers, too many beside the multipliers in a 1024-thread group's
    //  128-register budget; 512 threads with 32 rows each have 256)
    if constexpr (sizeof(T) == 16)
      hipLaunchKernelGGL((hr_lu_kernel<T, 512, 32>), dim3(1), dim3(512), 0, stream, q, ldq, b, rmat, lu, y1, tb, taus, flag);
    else
      hipLaunchKernelGGL((hr_lu_kernel<T, NT, 16>), dim3(1), dim3(NT), 0, stream, q, ldq, b, rmat, lu, y1, tb, taus, flag);
  }
  else if (b == 64)
    hipLaunchKernelGGL((hr_lu_kernel<T, NT, 4>), dim3(1), dim3(NT), 0, stream, q, ldq, b, rmat, lu, y1, tb, taus, flag);
  else {
    fprintf(stderr, "[dlaf_mi355x] Householder reconstruction: band %d is not supported\n", b);
    abort();
  }
}

void hr_kernels_init() {}

#define INST(T)                                                                                            \
  template void launch_hr_transpose<T>(T*, int, long, T*, long, bool, const int*, hipStream_t);            \
  template void launch_hr_gate<T>(T*, int, int, double, int*, hipStream_t);                                \
  template void launch_hr_orth<T>(T*, int, int, double, int*, int*, hipStream_t);                          \
  template void launch_hr_lu<T>(T*, long, int, const T*, T*, T*, T*, T*, const int*, hipStream_t);
INST(float)
INST(double)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
