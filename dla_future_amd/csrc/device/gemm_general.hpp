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

  __device__ __forceinline__ void store(R* __restrict__ lds, int mode, bool cj) const {
    const int t = threadIdx.x;
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

// acc += a(mrows x K) * b(ncols x K)^H  (rows >= mrows / ncols and k >= K contribute zero).  All threads of the
// workgroup call it; lds: 2 * Cfg::BUF_ELEMS of R (Cfg must be an un-paired, plane-separated configuration).
template <class Cfg, class T>
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
  OpSlab<T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS> sa;
  OpSlab<T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS> sb;
  const int ma = sa.pick_mode(da, mrows, K), mb = sb.pick_mode(db, ncols, K);
  sa.load(da, ma, 0, mrows, K);
  sb.load(db, mb, 0, ncols, K);
  sa.store(lds, ma, da.conj != 0);
  sb.store(lds + Cfg::A_ELEMS, mb, db.conj != 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    R* cur = lds + (kt & 1) * Cfg::BUF_ELEMS;
    R* nxt = lds + ((kt + 1) & 1) * Cfg::BUF_ELEMS;
    const bool more = (kt + 1) < nk;
    if (more) {
      sa.load(da, ma, (kt + 1) * Cfg::BK, mrows, K);
      sb.load(db, mb, (kt + 1) * Cfg::BK, ncols, K);
    }
    mma_slab<Cfg>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane);
    if (more) {
      sa.store(nxt, ma, da.conj != 0);
      sb.store(nxt + Cfg::A_ELEMS, mb, db.conj != 0);
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
