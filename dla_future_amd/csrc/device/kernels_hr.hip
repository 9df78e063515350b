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
// orthonormal to rounding while cond(P)^2 eps << 1.  Beyond `limit`, or for a factor that is not finite, the flag is
// raised.  One wave.
template <class T>
__global__ __launch_bounds__(64) void hr_gate_kernel(const T* l, int ld, int b, T limit, int* flag) {
  if (*flag != 0)
    return;
  T mx = 0, mn = 0;
  bool bad = false, first = true;
  for (int j = threadIdx.x; j < b; j += 64) {
    const T d = l[j + (long) j * ld];
    if (!(d > T(0)) || !(d < T(1e300)))
      bad = true;
    mx = first ? d : (d > mx ? d : mx);
    mn = first ? d : (d < mn ? d : mn);
    first = false;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const T omx = __shfl_xor(mx, off), omn = __shfl_xor(mn, off);
    const int ofirst = __shfl_xor(first ? 1 : 0, off);
    if (!ofirst) {
      mx = first ? omx : (omx > mx ? omx : mx);
      mn = first ? omn : (omn < mn ? omn : mn);
      first = false;
    }
    bad = bad || (__shfl_xor(bad ? 1 : 0, off) != 0);
  }
  if (threadIdx.x == 0 && (bad || !(mx <= limit * mn)))
    atomicCAS(flag, 0, 1);
}

// One workgroup of NT threads.  In: the top b x b block of Q (column-major, ldq), the two Cholesky factors L1, L2
// (lower, ld b: G = L1 L1^T, G2 = L2 L2^T).  Out:
//   top block of Q  <-  xGEQR2's output: S R on and above the diagonal (R = L2^T L1^T), V1 strictly below it
//   lu   (b x b, ld b, lower)  = U^T: the solve of the rows below, V2 = Q2 U^-1, runs as X (U^T)^T = Q2 on the panel TRSM
//   y1   (b x b, ld b, lower, unit diagonal stored)  = V1
//   tb   (b x b, ld b, upper, strict lower part zero) = -U S: T = tb V1^-T by one more TRSM
//   taus[j] = -u_jj s_j
// The elimination keeps W = Q1 in LDS ([row][col], stride b + 1) and writes NOTHING into row j / column j during
// step j (the column is scaled at the end, every thread forms the pivot u_jj = w_jj - s_j for itself): one barrier per
// step, b steps.
template <class T, int NT>
__global__ __launch_bounds__(NT) void hr_lu_kernel(T* q, long ldq, int b, const T* l1, const T* l2, T* lu, T* y1, T* tb, T* taus,
                                                   const int* flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T* w = reinterpret_cast<T*>(lds_raw);  // [b][b + 1]
  T* sg = w + (size_t) b * (b + 1);      // [b] signs
  T* pv = sg + b;                        // [b] pivots u_jj
  if (*flag != 0)
    return;
  const int t = threadIdx.x;
  const int st = b + 1;
  for (int idx = t; idx < b * b; idx += NT) {
    const int r = idx % b, c = idx / b;
    w[r * st + c] = q[r + (long) c * ldq];
  }
  __syncthreads();
  // ---- R = L2^T L1^T, rows scaled by S later: kept in registers until the signs are known --------------------------
  // element e = t + k NT of the upper triangle enumeration (j <= c), column-major over c
  constexpr int kMaxPer = 16;  // b <= 128 with NT = 1024: 8256 / 1024 -> 9
  T racc[kMaxPer];
  const int nup = b * (b + 1) / 2;
#pragma unroll
  for (int k = 0; k < kMaxPer; ++k) {
    racc[k] = 0;
    const int e = t + k * NT;
    if (e < nup) {
      // e -> (j, c), j <= c: c = largest with c (c + 1) / 2 <= e
      int c = (int) ((sqrtf(8.0f * (float) e + 1.0f) - 1.0f) * 0.5f);
      while ((c + 1) * (c + 2) / 2 <= e)
        ++c;
      while (c * (c + 1) / 2 > e)
        --c;
      const int j = e - c * (c + 1) / 2;
      T acc = 0;
      for (int kk = j; kk <= c; ++kk)
        acc += l2[kk + (long) j * b] * l1[c + (long) kk * b];
      racc[k] = acc;
    }
  }
  // ---- elimination --------------------------------------------------------------------------------------------------
  const int c_own = t % b;          // (b a power of two or not: the mapping only needs NT >= b)
  const int r_first = t / b, r_step = NT / b;
  for (int j = 0; j < b; ++j) {
    const T wjj = w[j * st + j];
    const T s = (wjj >= T(0)) ? T(-1) : T(1);
    const T piv = wjj - s;
    if (c_own > j) {
      const T u = w[j * st + c_own] / piv;  // (u_jc / u_jj: one division per thread and step)
      for (int r = j + 1 + r_first; r < b; r += r_step)
        w[r * st + c_own] -= w[r * st + j] * u;
    }
    if (t == 0) {
      sg[j] = s;
      pv[j] = piv;
    }
    __syncthreads();
  }
  // ---- outputs ----------------------------------------------------------------------------------------------------------
  for (int idx = t; idx < b * b; idx += NT) {
    const int r = idx % b, c = idx / b;
    T yv = 0, uv = 0;
    if (r > c)
      yv = w[r * st + c] / pv[c];          // V1, strictly lower
    else if (r == c)
      uv = pv[c];
    else
      uv = w[r * st + c];                  // U, strictly upper
    // y1: lower with the unit diagonal stored
    y1[r + (long) c * b] = r > c ? yv : (r == c ? T(1) : T(0));
    // lu = U^T: element (c, r) of lu is u_rc
    lu[c + (long) r * b] = r <= c ? uv : T(0);
    // tb = -U S: column c scaled by s_c
    tb[r + (long) c * b] = r <= c ? -uv * sg[c] : T(0);
    if (r > c)
      q[r + (long) c * ldq] = yv;
    if (r == c)
      taus[c] = -pv[c] * sg[c];
  }
#pragma unroll
  for (int k = 0; k < kMaxPer; ++k) {
    const int e = t + k * NT;
    if (e < nup) {
      int c = (int) ((sqrtf(8.0f * (float) e + 1.0f) - 1.0f) * 0.5f);
      while ((c + 1) * (c + 2) / 2 <= e)
        ++c;
      while (c * (c + 1) / 2 > e)
        --c;
      const int j = e - c * (c + 1) / 2;
      q[j + (long) c * ldq] = sg[j] * racc[k];
    }
  }
}

}  // namespace

bool panel_qr_blocked_supported(int b, long m, int nr, size_t elem_size, bool is_complex) {
  static const bool on = [] {
    const char* e = std::getenv("DLAF_MI355X_QR_BLOCKED");
    return e ? std::atoi(e) != 0 : true;
  }();
  // fp64 panels with whole 64-column blocks, a full set of reflectors and at least 2 b rows; everything else
  // (complex and single precision, the last panels of a matrix, narrow bands) keeps the reflector-by-reflector kernel
  return on && !is_complex && elem_size == 8 && b >= 64 && b <= 128 && b % 64 == 0 && nr == b && m >= 2L * b;
}

template <class T>
void launch_hr_transpose(T* qt, int b, long m, T* cm, long ld, bool to_cm, const int* flag, hipStream_t stream) {
  if (m <= 0 || b <= 0)
    return;
  dim3 grid((unsigned) ((m + kHrTile - 1) / kHrTile), (unsigned) ((b + kHrTile - 1) / kHrTile), 1);
  hipLaunchKernelGGL((hr_transpose_kernel<T>), grid, dim3(256), 0, stream, qt, b, m, cm, ld, to_cm ? 1 : 0, flag);
}

template <class T>
void launch_hr_gate(const T* l, int ld, int b, double limit, int* flag, hipStream_t stream) {
  hipLaunchKernelGGL((hr_gate_kernel<T>), dim3(1), dim3(64), 0, stream, l, ld, b, (T) limit, flag);
}

template <class T>
void launch_hr_lu(T* q, long ldq, int b, const T* l1, const T* l2, T* lu, T* y1, T* tb, T* taus, const int* flag,
                  hipStream_t stream) {
  constexpr int NT = 1024;
  const size_t lds = ((size_t) b * (b + 1) + 2 * (size_t) b) * sizeof(T);
  if (b > 128 || lds > 150 * 1024) {
    fprintf(stderr, "[dlaf_mi355x] Householder reconstruction: band %d is not supported\n", b);
    abort();
  }
  hipLaunchKernelGGL((hr_lu_kernel<T, NT>), dim3(1), dim3(NT), lds, stream, q, ldq, b, l1, l2, lu, y1, tb, taus, flag);
}

void hr_kernels_init() {
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&hr_lu_kernel<double, 1024>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&hr_lu_kernel<float, 1024>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

#define INST(T)                                                                                            \
  template void launch_hr_transpose<T>(T*, int, long, T*, long, bool, const int*, hipStream_t);            \
  template void launch_hr_gate<T>(const T*, int, int, double, int*, hipStream_t);                          \
  template void launch_hr_lu<T>(T*, long, int, const T*, const T*, T*, T*, T*, T*, const int*, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace dlaf_mi355x
