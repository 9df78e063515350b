// common.hpp -- element types, fp32/fp64 MFMA traits and staging helpers shared by the
// gfx950 tile kernels (trailing update, panel TRSM, diagonal POTRF).
//
// Replaces the vendor-library calls of the reference's GPU backend
// (include/dlaf/blas/tile.h:370-470 rocBLAS gemm/herk/trsm, include/dlaf/lapack/tile.h:577-606
// rocSOLVER potrf) with hand-written CDNA4 code: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32
// register tiles fed from LDS-staged panels.
#pragma once
#include <hip/hip_runtime.h>

namespace dlaf_mi355x {

template <class R>
struct alignas(2 * sizeof(R)) cplx {
  R re, im;
};
using cfloat = cplx<float>;
using cdouble = cplx<double>;

template <class T>
struct TypeInfo;
template <>
struct TypeInfo<float> {
  using real = float;
  static constexpr bool is_complex = false;
  static constexpr char tag = 's';
};
template <>
struct TypeInfo<double> {
  using real = double;
  static constexpr bool is_complex = false;
  static constexpr char tag = 'd';
};
template <>
struct TypeInfo<cfloat> {
  using real = float;
  static constexpr bool is_complex = true;
  static constexpr char tag = 'c';
};
template <>
struct TypeInfo<cdouble> {
  using real = double;
  static constexpr bool is_complex = true;
  static constexpr char tag = 'z';
};
template <class T>
using real_t = typename TypeInfo<T>::real;

// ---- element helpers (device) ------------------------------------------------------------
template <class T>
__device__ __forceinline__ real_t<T> re_of(const T& v) {
  if constexpr (TypeInfo<T>::is_complex)
    return v.re;
  else
    return v;
}
template <class T>
__device__ __forceinline__ real_t<T> im_of(const T& v) {
  if constexpr (TypeInfo<T>::is_complex)
    return v.im;
  else
    return real_t<T>(0);
}
template <class T>
__device__ __forceinline__ T make_el(real_t<T> re, real_t<T> im) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{re, im};
  else
    return re;
}
template <class T>
__device__ __forceinline__ T zero_el() {
  return make_el<T>(real_t<T>(0), real_t<T>(0));
}

// ---- MFMA 16x16x4 traits -------------------------------------------------------------------
// D(16x16) = Aop(16x4) * Bop(4x16) + C.  Operand lane maps (both types): lane l holds
// Aop[i = l&15][k = l>>4] and Bop[k = l>>4][j = l&15]; result: column j = l&15 and
//   f64: row i = (l>>4) + 4*v        f32: row i = 4*(l>>4) + v        (v = 0..3)
// The kernels always put the memory-contiguous index (matrix ROW m) on j = l&15, i.e. they
// compute D = B_panel * A_panel^T so that a wave's loads/stores of C are 128-byte runs.
template <class R>
struct Mma;

template <>
struct Mma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double aop, double bop, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, c, 0, 0, 0);
  }
  // c - aop * bop: for the f64 MFMAs the BLGP field holds negate bits (neg:[1,0,0] = the first operand)
  static __device__ __forceinline__ acc_t mma_neg(double aop, double bop, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, c, 0, 0, 1);
  }
  // index along i (the 16 "Aop rows") that register v of lane-group g = l>>4 holds
  static __device__ __forceinline__ int irow(int g, int v) { return g + 4 * v; }
};

template <>
struct Mma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float aop, float bop, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(aop, bop, c, 0, 0, 0);
  }
  static __device__ __forceinline__ acc_t mma_neg(float aop, float bop, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(-aop, bop, c, 0, 0, 0);  // (BLGP is a lane swizzle for f32)
  }
  static __device__ __forceinline__ int irow(int g, int v) { return 4 * g + v; }
};

// k index (0..15 inside a 16-wide k tile) that lane-group g must supply as the *other*
// operand when accumulator register v of a previous product is fed back as an operand
// ("accumulator as next operand"): the accumulator holds i = irow(g, v), so the partner
// fragment has to be read at k = irow(g, v).  For f64 this equals the natural 4*v + g.
template <class R>
__device__ __forceinline__ int acc_as_operand_k(int g, int v) {
  return Mma<R>::irow(g, v);
}

// Physical placement of the calling wave: XCD (HW_REG_XCC_ID[3:0]) and shader engine / array / compute unit
// (HW_REG_HW_ID[15:8]).  Performance tool only: nothing depends on it for correctness.
__device__ __forceinline__ unsigned phys_cu_key(unsigned& xcc) {
  xcc = (unsigned) __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
  return (unsigned) __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4) & 0xffu;
}

constexpr int kThreads = 256;  // 4 waves per workgroup (the default; BlockCfg::THREADS says otherwise where it differs)
constexpr int kLdsPad = 16;    // elements of padding per k-row of an LDS panel image:
                               // (ROWS+16)*8 B = 128 mod 256 and (ROWS+16)*4 B = 64 mod 128,
                               // so the 4 k-rows a wave reads per MFMA step hit disjoint banks

// ---- global -> register -> LDS staging of a (ROWS x BK) panel slab -------------------------
// Global image: column-major, element (r, k) at src[r + k*ld].  LDS image: [k][ROWS+pad]
// (k-major, rows contiguous), one plane for real types, re/im planes for complex.
// VEC: 16-byte loads (needs src 16-B aligned, ld*sizeof(T) % 16 == 0); otherwise per element.
// IL (complex only): re/im interleaved in LDS ([k][LD] of (re, im) pairs) instead of two planes.
template <class T, int ROWS, int BK, bool VEC, int LD_ = ROWS + kLdsPad, bool IL = false, int THREADS = kThreads>
struct Slab {
  using R = real_t<T>;
  static constexpr bool CX = TypeInfo<T>::is_complex;
  static constexpr int VE = VEC ? ((16 / (int) sizeof(T)) > 0 ? (16 / (int) sizeof(T)) : 1) : 1;
  static constexpr int NL = (ROWS * BK) / (THREADS * VE);
  static constexpr int LD = LD_;
  static constexpr int PLANE = BK * LD;                  // elements of R per plane
  static constexpr int ELEMS = (CX ? 2 : 1) * PLANE;     // elements of R per slab image
  static_assert((ROWS * BK) % (THREADS * VE) == 0, "slab must divide over the workgroup");
  static_assert(ROWS % VE == 0, "rows must be a multiple of the vector width");

  T regs[NL][VE];

  // rows_valid / k_valid only read when EDGE.  Two-segment operands (EDGE only): columns >= k1 come from src2
  // (passed shifted back by k1 columns), whatever k1 is -- a slab may straddle it.
  template <bool EDGE>
  __device__ __forceinline__ void load(const T* __restrict__ src, long ld, int k0, int rows_valid, int k_valid,
                                       const T* __restrict__ src2 = nullptr, int k1 = 1 << 30) {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int idx = t + THREADS * q;
      const int r = (idx % (ROWS / VE)) * VE;
      const int k = idx / (ROWS / VE);
      const T* p = ((EDGE && (k0 + k) >= k1) ? src2 : src) + r + (long) (k0 + k) * ld;
      if constexpr (!EDGE) {
        if constexpr (VE == 1) {
          regs[q][0] = *p;
        }
        else {
          // T may be a struct (complex): go through a 16-byte integer vector
          typedef unsigned int u4 __attribute__((ext_vector_type(4)));
          u4 raw = *reinterpret_cast<const u4*>(p);
          __builtin_memcpy(&regs[q][0], &raw, 16);
        }
      }
      else {
        const bool kin = (k0 + k) < k_valid;
#pragma unroll
        for (int e = 0; e < VE; ++e)
          regs[q][e] = (kin && (r + e) < rows_valid) ? p[e] : zero_el<T>();
      }
    }
  }

  __device__ __forceinline__ void store(R* __restrict__ lds) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int idx = t + THREADS * q;
      const int r = (idx % (ROWS / VE)) * VE;
      const int k = idx / (ROWS / VE);
      if constexpr (IL) {
        R* d = lds + 2 * (k * LD + r);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          d[2 * e] = re_of(regs[q][e]);
          d[2 * e + 1] = im_of(regs[q][e]);
        }
      }
      else {
        R* d = lds + k * LD + r;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          d[e] = re_of(regs[q][e]);
          if constexpr (CX)
            d[PLANE + e] = im_of(regs[q][e]);
        }
      }
    }
  }
};

}  // namespace dlaf_mi355x
