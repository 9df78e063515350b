// lane_ops.hpp -- exchanges between the lanes of a wave without the LDS crossbar (gfx950), and the wave reductions built
// on them (band_to_tridiagonal's register kernel, the T-factor kernel).
#pragma once

#include "common.hpp"

namespace dlaf_mi355x {

template <class T>
__device__ __forceinline__ T lane_add(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re + b.re, a.im + b.im};
  else
    return a + b;
}

// Lane exchanges without the LDS crossbar (ds_bpermute: an address register and ~50 clocks of latency per exchange, and
// two selects per word to pick what to send): gfx950's v_permlane32_swap / v_permlane16_swap trade the upper half (the
// odd rows) of one register against the lower half (the even rows) of another -- exactly the "keep half, send half"
// step of a transposing reduction, in one instruction per word and no select -- and DPP row controls reach the partner
// inside a row of 16 (profiles/r04_permlane_probe.txt has the lane maps).
template <class R, class F>
__device__ __forceinline__ R words_map(const R& x, F f) {
  constexpr int W = (int) sizeof(R) / 4;
  struct Wd {
    unsigned w[W];
  };
  Wd v = __builtin_bit_cast(Wd, x);
#pragma unroll
  for (int i = 0; i < W; ++i)
    v.w[i] = f(v.w[i]);
  return __builtin_bit_cast(R, v);
}
template <int CTRL, class R>
__device__ __forceinline__ R dpp_real(const R& x) {
  return words_map(x, [](unsigned w) { return (unsigned) __builtin_amdgcn_update_dpp(0u, w, CTRL, 0xf, 0xf, false); });
}
template <int CTRL, class T>
__device__ __forceinline__ T dpp_t(const T& v) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{dpp_real<CTRL>(v.re), dpp_real<CTRL>(v.im)};
  else
    return dpp_real<CTRL>(v);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppRor8 = 0x128;  // quad_perm [1,0,3,2] / [2,3,0,1]
// a <- [a.lo32 | b.lo32], b <- [a.hi32 | b.hi32]  (ROWS = false);  a <- rows [a0 b0 a2 b2], b <- rows [a1 b1 a3 b3]  (true)
template <bool ROWS, class R>
__device__ __forceinline__ void swap_real(R& a, R& b) {
  constexpr int W = (int) sizeof(R) / 4;
  struct Wd {
    unsigned w[W];
  };
  Wd x = __builtin_bit_cast(Wd, a), y = __builtin_bit_cast(Wd, b);
#pragma unroll
  for (int i = 0; i < W; ++i) {
    if constexpr (ROWS) {
      const auto r = __builtin_amdgcn_permlane16_swap(x.w[i], y.w[i], false, false);
      x.w[i] = r[0];
      y.w[i] = r[1];
    }
    else {
      const auto r = __builtin_amdgcn_permlane32_swap(x.w[i], y.w[i], false, false);
      x.w[i] = r[0];
      y.w[i] = r[1];
    }
  }
  a = __builtin_bit_cast(R, x);
  b = __builtin_bit_cast(R, y);
}
// the sum of a and b with a's total over the pair of halves (rows) in the lower (even) one, b's in the upper (odd) one
template <bool ROWS, class T>
__device__ __forceinline__ T swap_add(T a, T b) {
  if constexpr (TypeInfo<T>::is_complex) {
    swap_real<ROWS>(a.re, b.re);
    swap_real<ROWS>(a.im, b.im);
  }
  else
    swap_real<ROWS>(a, b);
  return lane_add(a, b);
}

// the total of one value over the wave, in every lane
template <class T>
__device__ __forceinline__ T wave_sum_fast(T v) {
  v = lane_add(v, dpp_t<kDppXor1>(v));
  v = lane_add(v, dpp_t<kDppXor2>(v));
  v = lane_add(v, dpp_t<kDppHalfMirror>(v));
  v = lane_add(v, dpp_t<kDppRor8>(v));
  v = swap_add<true>(v, v);
  v = swap_add<false>(v, v);
  return v;
}

}  // namespace dlaf_mi355x
