// gemm_general.hpp -- workgroup-level MFMA product with GENERAL operand access, for the kernels of the
// eigensolver stages (reduction to band, its back-transformation):
//        acc(BM x BN) += sum_k a(m, k) * conj(b(n, k))
// where each operand is described by (pointer, row stride, k stride, conjugate) -- so N, T and C forms of both
// factors, and the lower-stored Hermitian tile of xHEMM, all run through the same MFMA register tiles and LDS
// images as the Cholesky trailing update (mma_core.hpp: mma_slab, Acc).  Slabs of BK values of k are staged
// global -> registers -> LDS, double buffered; the register stage is what lets a k-contiguous operand (a
// transposed matrix) be turned into the [k][row] LDS image the MFMA fragments are read from.
//
// Reference: the tile::gemm / tile::hemm / tile::trmm3 calls of eigensolver/reduction_to_band/impl.h:231-281,
// :425-462, :520-542 and eigensolver/bt_reduction_to_band/impl.h:91-129 (blaspp / rocBLAS one tile at a time).
#pragma once
#include "mma_core.hpp"

namespace dlaf_mi355x {

// element (r, k) of an operand = p[r * rs + k * ks], conjugated when conj != 0.
// herm != 0 (rs == 1): p is the origin of a square tile holding a Hermitian matrix in its LOWER triangle,
// the block starts at tile row roff: (r, k) -> ra = roff + r;  ra >= k ? p[ra + k * ks] : conj(p[k + ra * ks]),
// the diagonal taken as real (xHEMM semantics).
template <class T>
struct OpDesc {
  const T* p = nullptr;
  long rs = 1, ks = 0;
  int conj = 0;
  int herm = 0;
  int roff = 0;
};

template <class T>
__device__ __forceinline__ T conj_if(const T& v, bool cj) {
  if constexpr (TypeInfo<T>::is_complex)
    return cj ? T{v.re, -v.im} : v;
  else
    return v;
}

// (ROWS x BK) slab of an operand: global -> registers (load) -> LDS image [k][LD] (+ imaginary plane) (store).
//   mode 0: rows contiguous (rs == 1), 16-byte loads along the rows      (interior, aligned)
//   mode 1: k contiguous (ks == 1), 16-byte loads along k                (interior in k, aligned)
//   mode 2: element by element, bounds-checked, any strides, Hermitian tiles
template <class T, int ROWS, int BK, int LD, int THREADS>
struct OpSlab {
  using R = real_t<T>;
  static constexpr bool CX = TypeInfo<T>::is_complex;
  static constexpr int VE = (16 / (int) sizeof(T)) > 0 ? (16 / (int) sizeof(T)) : 1;
  static constexpr int PER = (ROWS * BK) / THREADS;  // elements per thread
  static constexpr int NV = PER / VE;
  static constexpr int PLANE = BK * LD;
  static_assert((ROWS * BK) % (THREADS * VE) == 0, "slab must divide over the workgroup");
  static_assert(ROWS % VE == 0 && BK % VE == 0, "vector width");

  T regs[PER];

  // uniform over the workgroup
  static __device__ __forceinline__ int pick_mode(const OpDesc<T>& d, int rows_valid, int K) {
    if (d.herm || K % BK != 0)
      return 2;
    const bool al = (reinterpret_cast<uintptr_t>(d.p) % 16 == 0);
    if (d.rs == 1 && al && (d.ks * (long) sizeof(T)) % 16 == 0 && rows_valid % VE == 0)
      return 0;
    if (d.ks == 1 && al && (d.rs * (long) sizeof(T)) % 16 == 0)
      return 1;
    return 2;
  }

  __device__ __forceinline__ void load(const OpDesc<T>& d, int mode, int k0, int rows_valid, int k_valid) {
    const int t = threadIdx.x;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    if (mode == 0) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = t + THREADS * q;
        const int r = (idx % (ROWS / VE)) * VE;
        const int k = idx / (ROWS / VE);
        if (r < rows_valid) {
          const u4 raw = *reinterpret_cast<const u4*>(d.p + r + (long) (k0 + k) * d.ks);
          __builtin_memcpy(&regs[q * VE], &raw, 16);
        }
        else {
#pragma unroll
          for (int e = 0; e < VE; ++e)
            regs[q * VE + e] = zero_el<T>();
        }
      }
    }
    else if (mode == 1) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = t + THREADS * q;
        const int k = (idx % (BK / VE)) * VE;
        const int r = idx / (BK / VE);
        if (r < rows_valid) {
          const u4 raw = *reinterpret_cast<const u4*>(d.p + (long) r * d.rs + (k0 + k));
          __builtin_memcpy(&regs[q * VE], &raw, 16);
        }
        else {
#pragma unroll
          for (int e = 0; e < VE; ++e)
            regs[q * VE + e] = zero_el<T>();
        }
      }
    }
    else {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int idx = t + THREADS * q;
        const int r = idx % ROWS;
        const int k = k0 + idx / ROWS;
        T v = zero_el<T>();
        if (r < rows_valid && k < k_valid) {
          if (d.herm) {
            const long ra = d.roff + r;
            if (ra > k)
              v = d.p[ra + (long) k * d.ks];
            else if (ra < k)
              v = conj_if(d.p[k + ra * d.ks], true);
            else
              v = make_el<T>(re_of(d.p[ra + (long) k * d.ks]), R(0));
          }
          else {
            v = d.p[(long) r * d.rs + (long) k * d.ks];
          }
        }
        regs[q] = v;
      }
    }
  }

  // Transposed image (TIMG, mode 1 operands of the fixed-mode kernels): [row][LDT = BK + 2] instead of [k][LD].
  // Stored as loaded -- VE consecutive k of one row, one 16-byte LDS write for real types -- and read by the MFMA
  // fragments with a row stride of LDT words: (c LDT + g) mod 32 is distinct over the 32 lanes of a read group
  // (LDT = 18: even multiples of 18 hit the even banks once, the neighbour k the odd ones), so neither side
  // conflicts.  In the [k][LD] image the same store is 8-way conflicted (LD = 0 mod 16 words puts the eight k
  // pairs of a row on one bank): measured, it kept both xHEMM kernels at 33 TFlop/s.
  static constexpr int LDT = BK + 2;
  static constexpr bool TIMG_FITS = ROWS * LDT <= PLANE;
  template <bool TIMG>
  __device__ __forceinline__ void store(R* __restrict__ lds, int mode, bool cj) const {
    const int t = threadIdx.x;
    if constexpr (TIMG) {
      static_assert(TIMG_FITS, "transposed image must fit the slab plane");
      // (mode 1 only: the caller promised it)
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = t + THREADS * q;
        const int k = (idx % (BK / VE)) * VE;
        const int r = idx / (BK / VE);
        R* d = lds + r * LDT + k;
        if constexpr (!CX) {
          if constexpr (VE * sizeof(R) == 16 && (LDT * sizeof(R)) % 16 == 0) {
            typedef R rv __attribute__((ext_vector_type(VE)));
            rv v;
#pragma unroll
            for (int e = 0; e < VE; ++e)
              v[e] = regs[q * VE + e];
            *reinterpret_cast<rv*>(d) = v;
          }
          else {
#pragma unroll
            for (int e = 0; e < VE; ++e)
              d[e] = regs[q * VE + e];
          }
        }
        else {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            d[e] = re_of(regs[q * VE + e]);
            d[PLANE + e] = cj ? -im_of(regs[q * VE + e]) : im_of(regs[q * VE + e]);
          }
        }
      }
      return;
    }
    auto put = [&](int r, int k, const T& v) {
      lds[k * LD + r] = re_of(v);
      if constexpr (CX)
        lds[PLANE + k * LD + r] = cj ? -im_of(v) : im_of(v);
    };
    if (mode == 0) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = t + THREADS * q;
        const int r = (idx % (ROWS / VE)) * VE;
        const int k = idx / (ROWS / VE);
#pragma unroll
        for (int e = 0; e < VE; ++e)
          put(r + e, k, regs[q * VE + e]);
      }
    }
    else if (mode == 1) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = t + THREADS * q;
        const int k = (idx % (BK / VE)) * VE;
        const int r = idx / (BK / VE);
#pragma unroll
        for (int e = 0; e < VE; ++e)
          put(r, k + e, regs[q * VE + e]);
      }
    }
    else {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int idx = t + THREADS * q;
        put(idx % ROWS, idx / ROWS, regs[q]);
      }
    }
  }
};

// one BK slab of MFMAs like mma_slab (mma_core.hpp), each operand image either [k][LD] or transposed [row][BK + 2]
template <class Cfg, bool TA, bool TB>
__device__ __forceinline__ void mma_slab_g(const typename Cfg::R* __restrict__ As, const typename Cfg::R* __restrict__ Bs,
                                           Acc<Cfg>& acc, int wm, int wn, int lane) {
  using R = typename Cfg::R;
  constexpr int LDT = Cfg::BK + 2;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int k4 = 0; k4 < Cfg::BK / 4; ++k4) {
    const int kk = k4 * 4 + g;
    R a_re[Cfg::TM], a_im[Cfg::TM], b_re[Cfg::TN], b_im[Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      const int row = wm * Cfg::WM + i * 16 + c;
      const int off = TA ? row * LDT + kk : kk * Cfg::LDA + row;
      a_re[i] = As[off];
      if constexpr (Cfg::CX)
        a_im[i] = As[Cfg::A_PLANE + off];
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int row = wn * Cfg::WN + j * 16 + c;
      const int off = TB ? row * LDT + kk : kk * Cfg::LDB + row;
      b_re[j] = Bs[off];
      if constexpr (Cfg::CX)
        b_im[j] = Bs[Cfg::B_PLANE + off];
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        acc.re[i][j] = Mma<R>::mma(b_re[j], a_re[i], acc.re[i][j]);
        if constexpr (Cfg::CX) {
          acc.re[i][j] = Mma<R>::mma(b_im[j], a_im[i], acc.re[i][j]);
          acc.im[i][j] = Mma<R>::mma(b_re[j], a_im[i], acc.im[i][j]);
          acc.im[i][j] = Mma<R>::mma_neg(b_im[j], a_re[i], acc.im[i][j]);
        }
      }
  }
}

// acc += a(mrows x K) * b(ncols x K)^H  (rows >= mrows / ncols and k >= K contribute zero).  All threads of the
// workgroup call it; lds: 2 * Cfg::BUF_ELEMS of R (Cfg must be an un-paired, plane-separated configuration).
// MA / MB >= 0: the slab mode of the operand is a compile-time promise of the caller (the loaders of the other modes are
// not even compiled into the kernel: a kernel that may meet every mode keeps the registers of all three loaders busy)
template <class Cfg, class T, int MA = -1, int MB = -1>
__device__ __forceinline__ void gemm_acc(const OpDesc<T>& da, int mrows, const OpDesc<T>& db, int ncols, int K,
                                         typename Cfg::R* __restrict__ lds, Acc<Cfg>& acc) {
  using R = typename Cfg::R;
  static_assert(!Cfg::PAIRED && !Cfg::CXI, "plane-separated, padded LDS images");
  const int nk = (K + Cfg::BK - 1) / Cfg::BK;
  if (nk == 0)
    return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
  using SA = OpSlab<T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS>;
  using SB = OpSlab<T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS>;
  SA sa;
  SB sb;
  // fixed k-contiguous operands go through the transposed LDS image (OpSlab::store)
  constexpr bool TA = MA == 1 && SA::TIMG_FITS, TB = MB == 1 && SB::TIMG_FITS;
  const int ma = MA >= 0 ? MA : sa.pick_mode(da, mrows, K), mb = MB >= 0 ? MB : sb.pick_mode(db, ncols, K);
  sa.load(da, ma, 0, mrows, K);
  sb.load(db, mb, 0, ncols, K);
  sa.template store<TA>(lds, ma, da.conj != 0);
  sb.template store<TB>(lds + Cfg::A_ELEMS, mb, db.conj != 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    R* cur = lds + (kt & 1) * Cfg::BUF_ELEMS;
    R* nxt = lds + ((kt + 1) & 1) * Cfg::BUF_ELEMS;
    const bool more = (kt + 1) < nk;
    if (more) {
      sa.load(da, ma, (kt + 1) * Cfg::BK, mrows, K);
      sb.load(db, mb, (kt + 1) * Cfg::BK, ncols, K);
    }
    if constexpr (TA || TB)
      mma_slab_g<Cfg, TA, TB>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane);
    else
      mma_slab<Cfg>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane);
    if (more) {
      sa.template store<TA>(nxt, ma, da.conj != 0);
      sb.template store<TB>(nxt + Cfg::A_ELEMS, mb, db.conj != 0);
    }
    __syncthreads();
  }
}

// f(m, n, value) for every accumulator element this lane holds (m, n inside the BM x BN block)
template <class Cfg, class T, class F>
__device__ __forceinline__ void acc_foreach(const Acc<Cfg>& acc, F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int n = wn * Cfg::WN + acc_n<Cfg>(j, g, v);
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const int m = wm * Cfg::WM + acc_m<Cfg>(i, c);
        if constexpr (Cfg::CX)
          f(m, n, T{acc.re[i][j][v], acc.im[i][j][v]});
        else
          f(m, n, acc.re[i][j][v]);
      }
    }
}

// block configurations of the general kernels: the register / LDS budget of the trailing-update kernel, padded
// plane-separated LDS images (any operand form can be stored into them)
template <class T>
struct GenCfg;
template <>
struct GenCfg<float> {
  using type = BlockCfg<float, 128, 128, 64, 64, 16>;
};
template <>
struct GenCfg<double> {
  using type = BlockCfg<double, 128, 128, 64, 64, 16>;
};
template <>
struct GenCfg<cfloat> {
  using type = BlockCfg<cfloat, 128, 128, 64, 64, 16>;
};
template <>
struct GenCfg<cdouble> {
  using type = BlockCfg<cdouble, 128, 64, 64, 32, 8>;
};

}  // namespace dlaf_mi355x
