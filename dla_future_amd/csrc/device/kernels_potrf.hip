// kernels_potrf.hip -- factorization + inversion of one <=64 x <=64 diagonal block by a single
// workgroup, entirely in LDS.
//
// Reference: tile::potrf (lapack/tile.h:577-606: rocsolver_*potrf + the assert_info kernel,
// src/cusolver/assert_info.cu:35-45).  The tile-level POTRF of the MI355X build is blocked with
// inner block 64 (host loop in executor.cpp): this kernel is the unblocked leaf; the sub-panel
// solve and the in-tile trailing update reuse the TRSM and update kernels.  Besides L it emits
// inv(L), which turns every later triangular solve into MFMA GEMMs.
// A non-positive (or NaN) pivot stores the LAPACK-style global index into *info (first failure
// wins) -- the device-side replacement of assert_info's printf+trap.
#include "device_api.hpp"

namespace dlaf_mi355x {

constexpr int kPD = kDiagBlock;  // 64
constexpr int kPDLd = kPD + 1;   // LDS leading dimension (bank spread for row access)

template <class T>
__global__ __launch_bounds__(kThreads) void potrf_diag_kernel(T* __restrict__ a, int lda, int jb,
                                                               T* __restrict__ winv, int* info, int info_base,
                                                               int factor) {
  using R = real_t<T>;
  constexpr bool CX = TypeInfo<T>::is_complex;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* Lre = reinterpret_cast<R*>(lds_raw);          // [col][row], ld kPDLd
  R* Lim = Lre + kPD * kPDLd;                      // (complex only)
  R* Wre = Lre + (CX ? 2 : 1) * kPD * kPDLd;
  R* Wim = Wre + kPD * kPDLd;
  const int t = threadIdx.x;

  if (*info != 0)
    return;

  // ---- load the lower triangle, zero elsewhere -------------------------------------------------
  for (int idx = t; idx < kPD * kPD; idx += kThreads) {
    const int r = idx % kPD, c = idx / kPD;
    R re = 0, im = 0;
    if (r < jb && c < jb && r >= c) {
      const T v = a[r + (long) c * lda];
      re = re_of(v);
      im = im_of(v);
    }
    if (r == c && r >= jb)
      re = 1;  // pad to a 64x64 identity tail so the padded inverse stays finite
    Lre[c * kPDLd + r] = re;
    if constexpr (CX)
      Lim[c * kPDLd + r] = (r == c && factor) ? R(0) : im;
  }
  __syncthreads();

  // ---- right-looking unblocked Cholesky (xPOTF2 order) -------------------------------------------
  const int r = t & (kPD - 1), jg = t >> 6;
  for (int c = 0; factor && c < jb; ++c) {
    const R d = Lre[c * kPDLd + c];
    if (!(d > R(0))) {  // uniform: every thread reads the same LDS word
      if (t == 0)
        atomicCAS(info, 0, info_base + c + 1);
      return;
    }
    const R s = sqrt(d);
    __syncthreads();  // everyone has read d before it is overwritten
    if (jg == 0) {
      if (r == c)
        Lre[c * kPDLd + c] = s;
      else if (r > c && r < jb) {
        Lre[c * kPDLd + r] = Lre[c * kPDLd + r] / s;
        if constexpr (CX)
          Lim[c * kPDLd + r] = Lim[c * kPDLd + r] / s;
      }
    }
    __syncthreads();
    // trailing rank-1 update: A[r][j] -= L[r][c] * conj(L[j][c]) for c < j <= r
    if (r > c && r < jb) {
      const R lr_re = Lre[c * kPDLd + r];
      R lr_im = 0;
      if constexpr (CX)
        lr_im = Lim[c * kPDLd + r];
      for (int j = c + 1 + jg; j <= r; j += kThreads / kPD) {
        const R lj_re = Lre[c * kPDLd + j];
        if constexpr (CX) {
          const R lj_im = Lim[c * kPDLd + j];
          // (lr_re + i lr_im)(lj_re - i lj_im)
          Lre[j * kPDLd + r] -= lr_re * lj_re + lr_im * lj_im;
          if (j != r)
            Lim[j * kPDLd + r] -= lr_im * lj_re - lr_re * lj_im;
        }
        else {
          Lre[j * kPDLd + r] -= lr_re * lj_re;
        }
      }
    }
    __syncthreads();
  }

  // ---- W = inv(L): one thread per column, forward substitution -----------------------------------
  for (int idx = t; idx < kPD * kPDLd; idx += kThreads) {
    Wre[idx] = 0;
    if constexpr (CX)
      Wim[idx] = 0;
  }
  __syncthreads();
  if (t < jb) {
    const int c = t;
    for (int i = c; i < jb; ++i) {
      R s_re = (i == c) ? R(1) : R(0), s_im = 0;
      for (int k = c; k < i; ++k) {
        const R l_re = Lre[k * kPDLd + i];
        const R w_re = Wre[c * kPDLd + k];
        if constexpr (CX) {
          const R l_im = Lim[k * kPDLd + i];
          const R w_im = Wim[c * kPDLd + k];
          s_re -= l_re * w_re - l_im * w_im;
          s_im -= l_re * w_im + l_im * w_re;
        }
        else {
          s_re -= l_re * w_re;
        }
      }
      const R dd = Lre[i * kPDLd + i];
      if constexpr (CX) {
        const R di = Lim[i * kPDLd + i];  // zero for a Cholesky factor, general in invert-only mode
        if (di == R(0)) {
          Wre[c * kPDLd + i] = s_re / dd;
          Wim[c * kPDLd + i] = s_im / dd;
        }
        else {
          const R den = dd * dd + di * di;
          Wre[c * kPDLd + i] = (s_re * dd + s_im * di) / den;
          Wim[c * kPDLd + i] = (s_im * dd - s_re * di) / den;
        }
      }
      else {
        Wre[c * kPDLd + i] = s_re / dd;
      }
    }
  }
  __syncthreads();

  // ---- write back: lower triangle of L, dense 64x64 W --------------------------------------------
  for (int idx = t; idx < kPD * kPD; idx += kThreads) {
    const int rr = idx % kPD, cc = idx / kPD;
    if (factor && rr < jb && cc < jb && rr >= cc) {
      R im = 0;
      if constexpr (CX)
        im = Lim[cc * kPDLd + rr];
      a[rr + (long) cc * lda] = make_el<T>(Lre[cc * kPDLd + rr], im);
    }
    R wre = 0, wim = 0;
    if (rr < jb && cc < jb) {
      wre = Wre[cc * kPDLd + rr];
      if constexpr (CX)
        wim = Wim[cc * kPDLd + rr];
    }
    winv[rr + (long) cc * kPD] = make_el<T>(wre, wim);
  }
}

template <class T>
static constexpr int potrf_lds_bytes() {
  return (TypeInfo<T>::is_complex ? 4 : 2) * kPD * kPDLd * (int) sizeof(real_t<T>);
}

template <class T>
void launch_potrf_diag(T* a, int lda, int jb, T* winv_block, int* info, int info_base, hipStream_t stream,
                       bool factor) {
  if (jb <= 0)
    return;
  hipLaunchKernelGGL((potrf_diag_kernel<T>), dim3(1), dim3(kThreads), potrf_lds_bytes<T>(), stream, a, lda, jb,
                     winv_block, info, info_base, factor ? 1 : 0);
}

template <class T>
static void potrf_init_one() {
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_kernel<T>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, potrf_lds_bytes<T>());
}

void potrf_kernels_init() {
  potrf_init_one<float>();
  potrf_init_one<double>();
  potrf_init_one<cfloat>();
  potrf_init_one<cdouble>();
}

template void launch_potrf_diag<float>(float*, int, int, float*, int*, int, hipStream_t, bool);
template void launch_potrf_diag<double>(double*, int, int, double*, int*, int, hipStream_t, bool);
template void launch_potrf_diag<cfloat>(cfloat*, int, int, cfloat*, int*, int, hipStream_t, bool);
template void launch_potrf_diag<cdouble>(cdouble*, int, int, cdouble*, int*, int, hipStream_t, bool);

}  // namespace dlaf_mi355x
