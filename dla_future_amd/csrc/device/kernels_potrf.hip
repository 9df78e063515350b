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
#include "potrf_diag_core.hpp"

namespace dlaf_mi355x {

// Structure (all in LDS, 4 waves):
//   factor : for each 16-column panel: left-looking update by all threads, then the unblocked
//            16-column factorization by ONE wave with the panel row-resident in registers and the
//            pivot row broadcast by v_readlane (no barriers inside the panel);
//   invert : inv of the four 16x16 diagonal blocks (one wave each, forward substitution per column),
//            then the off-diagonal blocks by block distance W_ij = -W_ii (sum_k L_ik W_kj).
template <class T>
__global__ __launch_bounds__(kThreads) void potrf_diag_kernel(T* __restrict__ a, int lda, int jb,
                                                               T* __restrict__ winv, int* info, int info_base,
                                                               int factor, int upper, int unit, int n_total) {
  using R = real_t<T>;
  constexpr bool CX = TypeInfo<T>::is_complex;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* Lre = reinterpret_cast<R*>(lds_raw);          // [col][row], ld kPDLd
  R* Lim = Lre + kPD * kPDLd;                      // (complex only)
  R* Wre = Lre + (CX ? 2 : 1) * kPD * kPDLd;
  R* Wim = Wre + kPD * kPDLd;
  __shared__ int fail_col;
  const int t = threadIdx.x;

  if (*info != 0)
    return;
  if (n_total > 0) {
    // batched invert-only form: workgroup b handles the b-th 64 x 64 diagonal block of an n_total x n_total tile
    a += (long) blockIdx.x * kPD * (lda + 1);
    winv += (long) blockIdx.x * kPD * kPD;
    jb = min(kPD, n_total - (int) blockIdx.x * kPD);
  }
  if (t == 0)
    fail_col = -1;

  // ---- load the lower triangle, zero elsewhere; identity tail beyond jb --------------------------
  for (int idx = t; idx < kPD * kPD; idx += kThreads) {
    const int r = idx % kPD, c = idx / kPD;
    R re = 0, im = 0;
    if (r < jb && c < jb && r >= c) {
      // upper (invert-only mode): work on L' = U^H, inv(U) = inv(L')^H is written back transposed below
      const T v = upper ? a[c + (long) r * lda] : a[r + (long) c * lda];
      re = re_of(v);
      im = upper ? -im_of(v) : im_of(v);
      if (unit && r == c) {
        re = 1;
        im = 0;
      }
    }
    if (r == c && r >= jb)
      re = 1;
    Lre[c * kPDLd + r] = re;
    if constexpr (CX)
      Lim[c * kPDLd + r] = (r == c && factor) ? R(0) : im;
    Wre[c * kPDLd + r] = 0;
    if constexpr (CX)
      Wim[c * kPDLd + r] = 0;
  }
  __syncthreads();

  // ---- factor + invert in LDS ------------------------------------------------------------------------
  {
    const int failed = diag_factor_invert<T>(Lre, Lim, Wre, Wim, jb, factor, &fail_col);
    if (failed >= 0) {
      if (t == 0)
        atomicCAS(info, 0, info_base + failed + 1);
      return;
    }
  }

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
      if (upper) {
        wre = Wre[rr * kPDLd + cc];
        if constexpr (CX)
          wim = -Wim[rr * kPDLd + cc];
      }
      else {
        wre = Wre[cc * kPDLd + rr];
        if constexpr (CX)
          wim = Wim[cc * kPDLd + rr];
      }
    }
    winv[rr + (long) cc * kPD] = make_el<T>(wre, wim);
  }
}

template <class T>
static constexpr int potrf_lds_bytes() {
  return diag_lds_elems<T>() * (int) sizeof(real_t<T>);
}

template <class T>
void launch_potrf_diag(T* a, int lda, int jb, T* winv_block, int* info, int info_base, hipStream_t stream,
                       bool factor, bool upper, bool unit) {
  if (jb <= 0)
    return;
  hipLaunchKernelGGL((potrf_diag_kernel<T>), dim3(1), dim3(kThreads), potrf_lds_bytes<T>(), stream, a, lda, jb,
                     winv_block, info, info_base, factor ? 1 : 0, (!factor && upper) ? 1 : 0, (!factor && unit) ? 1 : 0, 0);
}

template <class T>
void launch_invert_diag_blocks(const T* tile, int ld, int kb, T* winv, int* info, hipStream_t stream, bool upper,
                               bool unit) {
  if (kb <= 0)
    return;
  hipLaunchKernelGGL((potrf_diag_kernel<T>), dim3((unsigned) ((kb + kPD - 1) / kPD)), dim3(kThreads),
                     potrf_lds_bytes<T>(), stream, const_cast<T*>(tile), ld, kPD, winv, info, 0, 0, upper ? 1 : 0,
                     unit ? 1 : 0, kb);
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

template void launch_potrf_diag<float>(float*, int, int, float*, int*, int, hipStream_t, bool, bool, bool);
template void launch_invert_diag_blocks<float>(const float*, int, int, float*, int*, hipStream_t, bool, bool);
template void launch_potrf_diag<double>(double*, int, int, double*, int*, int, hipStream_t, bool, bool, bool);
template void launch_invert_diag_blocks<double>(const double*, int, int, double*, int*, hipStream_t, bool, bool);
template void launch_potrf_diag<cfloat>(cfloat*, int, int, cfloat*, int*, int, hipStream_t, bool, bool, bool);
template void launch_invert_diag_blocks<cfloat>(const cfloat*, int, int, cfloat*, int*, hipStream_t, bool, bool);
template void launch_potrf_diag<cdouble>(cdouble*, int, int, cdouble*, int*, int, hipStream_t, bool, bool, bool);
template void launch_invert_diag_blocks<cdouble>(const cdouble*, int, int, cdouble*, int*, hipStream_t, bool, bool);

}  // namespace dlaf_mi355x
