// mma_core.hpp -- the workgroup-level MFMA GEMM core shared by the trailing-update and the
// panel-TRSM kernels:   acc(BM x BN) += A(BM x K) * B(BN x K)^H
// A, B column-major in global memory; slabs of BK columns are staged global -> registers -> LDS
// (double-buffered, one barrier per slab) and consumed as 16x16x4 MFMA fragments.
#pragma once
#include "common.hpp"

namespace dlaf_mi355x {

template <class T, int BM_, int BN_, int WM_, int WN_, int BK_>
struct BlockCfg {
  using R = real_t<T>;
  static constexpr bool CX = TypeInfo<T>::is_complex;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_;
  static constexpr int TM = WM / 16, TN = WN / 16;
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static_assert(WAVES_M * WAVES_N * 64 == kThreads, "4 waves per workgroup");
  static_assert(BK % 4 == 0 && WM % 16 == 0 && WN % 16 == 0, "MFMA 16x16x4 granularity");
  static constexpr int LDA = BM + kLdsPad, LDB = BN + kLdsPad;
  static constexpr int A_PLANE = BK * LDA, B_PLANE = BK * LDB;
  static constexpr int A_ELEMS = (CX ? 2 : 1) * A_PLANE, B_ELEMS = (CX ? 2 : 1) * B_PLANE;
  static constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS;
  static constexpr int LDS_BYTES = 2 * BUF_ELEMS * (int) sizeof(R);
};

template <class Cfg>
struct Acc {
  using R = typename Cfg::R;
  using acc_t = typename Mma<R>::acc_t;
  acc_t re[Cfg::TM][Cfg::TN];
  acc_t im[Cfg::CX ? Cfg::TM : 1][Cfg::CX ? Cfg::TN : 1];

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        re[i][j] = acc_t{0, 0, 0, 0};
        if constexpr (Cfg::CX)
          im[i][j] = acc_t{0, 0, 0, 0};
      }
  }
};

// one BK-slab of MFMAs out of the LDS images As ([k][LDA], planes re|im) and Bs ([k][LDB])
template <class Cfg>
__device__ __forceinline__ void mma_slab(const typename Cfg::R* __restrict__ As,
                                         const typename Cfg::R* __restrict__ Bs, Acc<Cfg>& acc, int wm,
                                         int wn, int lane) {
  using R = typename Cfg::R;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int k4 = 0; k4 < Cfg::BK / 4; ++k4) {
    const int kk = k4 * 4 + g;
    R a_re[Cfg::TM], a_im[Cfg::TM], b_re[Cfg::TN], b_im[Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      a_re[i] = As[kk * Cfg::LDA + wm * Cfg::WM + i * 16 + c];
      if constexpr (Cfg::CX)
        a_im[i] = As[Cfg::A_PLANE + kk * Cfg::LDA + wm * Cfg::WM + i * 16 + c];
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      b_re[j] = Bs[kk * Cfg::LDB + wn * Cfg::WN + j * 16 + c];
      if constexpr (Cfg::CX)
        b_im[j] = Bs[Cfg::B_PLANE + kk * Cfg::LDB + wn * Cfg::WN + j * 16 + c];
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        // D[i' = n][j' = m]: Aop <- B panel fragment, Bop <- A panel fragment
        acc.re[i][j] = Mma<R>::mma(b_re[j], a_re[i], acc.re[i][j]);
        if constexpr (Cfg::CX) {
          // (ar + i ai)(br - i bi) = (ar br + ai bi) + i (ai br - ar bi)
          acc.re[i][j] = Mma<R>::mma(b_im[j], a_im[i], acc.re[i][j]);
          acc.im[i][j] = Mma<R>::mma(b_re[j], a_im[i], acc.im[i][j]);
          acc.im[i][j] = Mma<R>::mma(b_im[j], -a_re[i], acc.im[i][j]);
        }
      }
  }
}

// acc += A(mrows x K) * B(ncols x K)^H for one BM x BN block.  lds: 2 * BUF_ELEMS of R.
// EDGE: rows >= mrows / ncols and k >= K are zero-filled; otherwise mrows == BM, ncols == BN and
// K % BK == 0 are the caller's promise.  All threads of the workgroup must call it.
template <class Cfg, class T, bool VEC, bool EDGE>
__device__ __forceinline__ void gemm_nt_block(const T* __restrict__ A, long lda, int mrows,
                                              const T* __restrict__ B, long ldb, int ncols, int K,
                                              typename Cfg::R* __restrict__ lds, Acc<Cfg>& acc) {
  using R = typename Cfg::R;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
  const int nk = (K + Cfg::BK - 1) / Cfg::BK;
  if (nk == 0)
    return;
  Slab<T, Cfg::BM, Cfg::BK, VEC> sa;
  Slab<T, Cfg::BN, Cfg::BK, VEC> sb;
  sa.template load<EDGE>(A, lda, 0, mrows, K);
  sb.template load<EDGE>(B, ldb, 0, ncols, K);
  sa.store(lds);
  sb.store(lds + Cfg::A_ELEMS);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    R* cur = lds + (kt & 1) * Cfg::BUF_ELEMS;
    R* nxt = lds + ((kt + 1) & 1) * Cfg::BUF_ELEMS;
    const bool more = (kt + 1) < nk;
    if (more) {
      sa.template load<EDGE>(A, lda, (kt + 1) * Cfg::BK, mrows, K);
      sb.template load<EDGE>(B, ldb, (kt + 1) * Cfg::BK, ncols, K);
    }
    mma_slab<Cfg>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane);
    if (more) {
      sa.store(nxt);
      sb.store(nxt + Cfg::A_ELEMS);
    }
    __syncthreads();
  }
}

}  // namespace dlaf_mi355x
