// potrf_diag_core.hpp -- in-LDS factorization + inversion of one <=64 x <=64 diagonal block by a
// 256-thread workgroup; shared by the stand-alone diagonal kernel and the cooperative tile POTRF.
#pragma once
#include "device_api.hpp"

namespace dlaf_mi355x {

constexpr int kPD = kDiagBlock;  // 64
constexpr int kPDLd = kPD + 1;   // LDS leading dimension (bank spread for row access)
constexpr int kPB = 16;          // inner panel width of the in-LDS factorization

// broadcast of one wave lane's double to the whole wave through two v_readlane_b32 (lane uniform)
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}


// sqrt(d) and 1 / sqrt(d) of a pivot (d > 0, far from the denormal and overflow ranges: pivots of a matrix that
// got this far) from one v_rsq_f64 and coupled Newton steps -- the library sqrt followed by a division is twice
// as long a dependency chain, and this chain is the critical path of the whole factorization (64 pivots per
// diagonal block, strictly one after the other).  Both results are within an ulp or two of the exact values.
__device__ __forceinline__ void pivot_sqrt(double d, double& sq, double& inv) {
  const double r = __builtin_amdgcn_rsq(d);  // ~2^-26 relative
  double g = d * r;                          // ~ sqrt(d)
  double h = 0.5 * r;                        // ~ 1 / (2 sqrt(d))
  double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-g, g, d);               // residual of the square
  g = __builtin_fma(e, h, g);
  e = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-g, g, d);
  sq = __builtin_fma(e, h, g);
  inv = h + h;
}
__device__ __forceinline__ void pivot_sqrt(float d, float& sq, float& inv) {
  sq = sqrtf(d);
  inv = 1.0f / sq;
}

// LDS image sizes (elements of R): L and W planes + the inversion scratch.
// PACK: W shares the planes of L -- W(i,j), i > j, sits at the TRANSPOSED position (column i, row j: the strictly
// upper triangle, which L does not use), its diagonal in 64 extra elements per plane behind the scratch.  Halves
// the footprint: complex<double> 145 KiB -> 80 KiB, which is what lets the cooperative POTRF of a complex tile run
// beside a workgroup of the bulk update (64 KiB) on the same CU.
template <class T, bool PACK = false>
constexpr int diag_lds_elems() {
  constexpr int planes = TypeInfo<T>::is_complex ? 2 : 1;
  return (PACK ? planes : 2 * planes) * kPD * kPDLd + planes * 3 * kPB * kPB + (PACK ? planes * kPD : 0);
}

// W(i, j) of the inverse under construction; plane = the re or im plane of W (== that of L when PACK), wd = the
// packed diagonal of the same plane
template <bool PACK, class R>
__device__ __forceinline__ R diag_w_get(const R* plane, const R* wd, int i, int j) {
  if constexpr (!PACK)
    return plane[j * kPDLd + i];
  else
    return i > j ? plane[i * kPDLd + j] : (i == j ? wd[i] : R(0));
}
template <bool PACK, class R>
__device__ __forceinline__ void diag_w_set(R* plane, R* wd, int i, int j, R v) {
  if constexpr (!PACK) {
    plane[j * kPDLd + i] = v;
  }
  else {
    if (i > j)
      plane[i * kPDLd + j] = v;
    else if (i == j)
      wd[i] = v;
  }
}

// L image in (Lre, Lim) ([col][row], ld kPDLd; lower triangle, identity beyond jb), W planes zeroed (PACK: Wre ==
// Lre, Wim == Lim, nothing to zero; read the result with diag_w_get and diag_wd_re / diag_wd_im).
// Factors (when factor != 0) and inverts in place.  Every thread of the workgroup must call it.
// Returns -1, or the failing column; *fail_col is a workgroup-shared int the caller set to -1 before
// the preceding barrier.
template <class T, bool PACK>
__device__ __forceinline__ real_t<T>* diag_wd_re(real_t<T>* Wre) {
  constexpr int planes = TypeInfo<T>::is_complex ? 2 : 1;
  return Wre + planes * kPD * kPDLd + planes * 3 * kPB * kPB;  // behind the T scratch
}
template <class T, bool PACK>
__device__ __forceinline__ real_t<T>* diag_wd_im(real_t<T>* Wre) {
  return diag_wd_re<T, PACK>(Wre) + kPD;
}

template <class T, bool PACK = false>
__device__ __forceinline__ int diag_factor_invert(real_t<T>* Lre, real_t<T>* Lim, real_t<T>* Wre, real_t<T>* Wim,
                                                  int jb, int factor, int* fail_col) {
  using R = real_t<T>;
  constexpr bool CX = TypeInfo<T>::is_complex;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  // ---- factorization ------------------------------------------------------------------------------
  for (int p0 = 0; factor && p0 < jb; p0 += kPB) {
    // (1) left-looking update of panel columns [p0, p0+16) with the finished columns [0, p0)
    if (p0 > 0) {
      const int r = lane;             // row
      const int cq = wave * 4;        // 4 columns per thread
      if (r >= p0) {
        R sre[4] = {0, 0, 0, 0}, sim[4] = {0, 0, 0, 0};
        for (int k = 0; k < p0; ++k) {
          const R lr_re = Lre[k * kPDLd + r];
          R lr_im = 0;
          if constexpr (CX)
            lr_im = Lim[k * kPDLd + r];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = p0 + cq + q;
            const R lc_re = Lre[k * kPDLd + c];
            if constexpr (CX) {
              const R lc_im = Lim[k * kPDLd + c];
              sre[q] += lr_re * lc_re + lr_im * lc_im;   // l_r * conj(l_c)
              sim[q] += lr_im * lc_re - lr_re * lc_im;
            }
            else {
              sre[q] += lr_re * lc_re;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = p0 + cq + q;
          if (r >= c) {
            Lre[c * kPDLd + r] -= sre[q];
            if constexpr (CX)
              if (r != c)
                Lim[c * kPDLd + r] -= sim[q];
          }
        }
      }
      __syncthreads();
    }
    // (2) unblocked factorization of the panel by wave 0: lane r holds row r of the panel
    if (wave == 0) {
      const int r = lane;
      R are[kPB], aim[kPB];
#pragma unroll
      for (int c = 0; c < kPB; ++c) {
        are[c] = Lre[(p0 + c) * kPDLd + r];
        aim[c] = CX ? Lim[(p0 + c) * kPDLd + r] : R(0);
      }
      int failed = -1;
      // Branch-free in the lane dimension: every lane runs the same arithmetic on its row (rows above the pivot
      // carry values nobody reads -- only r >= column is written back, only lanes below the pivot are broadcast),
      // so the register arrays need no per-branch copies; the pivot's sqrt and reciprocal are wave-uniform.
#pragma unroll
      for (int c = 0; c < kPB; ++c) {
        const int pr = p0 + c;  // pivot row == pivot column (global in the block)
        if (pr < jb && failed < 0) {
          const R d = lane_bcast(are[c], pr);
          if (!(d > R(0))) {
            failed = pr;
          }
          else {
            R sq, inv;
            pivot_sqrt(d, sq, inv);
            are[c] = (r == pr) ? sq : are[c] * inv;
            if constexpr (CX)
              aim[c] = (r == pr) ? R(0) : aim[c] * inv;
#pragma unroll
            for (int j = c + 1; j < kPB; ++j) {
              // l_j = L[p0+j][pr]: held by lane p0+j in are[c]
              const R lj_re = lane_bcast(are[c], p0 + j);
              if constexpr (CX) {
                const R lj_im = lane_bcast(aim[c], p0 + j);
                are[j] -= are[c] * lj_re + aim[c] * lj_im;
                aim[j] -= aim[c] * lj_re - are[c] * lj_im;
              }
              else {
                are[j] -= are[c] * lj_re;
              }
            }
          }
        }
      }
      if (failed >= 0) {
        if (lane == 0)
          *fail_col = failed;
      }
      else {
#pragma unroll
        for (int c = 0; c < kPB; ++c) {
          if (r >= p0 + c) {
            Lre[(p0 + c) * kPDLd + r] = are[c];
            if constexpr (CX)
              Lim[(p0 + c) * kPDLd + r] = (r == p0 + c) ? R(0) : aim[c];
          }
        }
      }
    }
    __syncthreads();
    if (*fail_col >= 0)
      return *fail_col;
  }

  // ---- W = inv(L) -----------------------------------------------------------------------------------
  R* const wd_re = PACK ? diag_wd_re<T, PACK>(Wre) : nullptr;
  R* const wd_im = PACK ? diag_wd_im<T, PACK>(Wre) : nullptr;
  // (a) diagonal 16x16 blocks: wave w inverts block w; lane c < 16 owns column c of the block
  {
    const int b0 = wave * kPB;
    if (lane < kPB && b0 < jb) {
      const int c = lane;
      R wre[kPB], wim[kPB];
#pragma unroll
      for (int i = 0; i < kPB; ++i) {
        R sre = (i == c) ? R(1) : R(0), sim = 0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
          // only k >= c contribute (w[k] = 0 above the diagonal)
          const R l_re = Lre[(b0 + k) * kPDLd + b0 + i];
          if constexpr (CX) {
            const R l_im = Lim[(b0 + k) * kPDLd + b0 + i];
            sre -= l_re * wre[k] - l_im * wim[k];
            sim -= l_re * wim[k] + l_im * wre[k];
          }
          else {
            sre -= l_re * wre[k];
          }
        }
        const R dd = Lre[(b0 + i) * kPDLd + b0 + i];
        if constexpr (CX) {
          const R di = Lim[(b0 + i) * kPDLd + b0 + i];
          if (di == R(0)) {
            wre[i] = sre / dd;
            wim[i] = sim / dd;
          }
          else {
            const R den = dd * dd + di * di;
            wre[i] = (sre * dd + sim * di) / den;
            wim[i] = (sim * dd - sre * di) / den;
          }
        }
        else {
          wre[i] = sre / dd;
          wim[i] = 0;
        }
        if (i < c) {  // strictly upper part of the column is zero
          wre[i] = 0;
          wim[i] = 0;
        }
      }
#pragma unroll
      for (int i = 0; i < kPB; ++i) {
        diag_w_set<PACK>(Wre, wd_re, b0 + i, b0 + c, wre[i]);
        if constexpr (CX)
          diag_w_set<PACK>(Wim, wd_im, b0 + i, b0 + c, wim[i]);
      }
    }
  }
  __syncthreads();
  // (b) off-diagonal blocks by block distance d: W(i,j) = -W(i,i) * sum_{k=j}^{i-1} L(i,k) W(k,j), i = j+d.
  // One thread per element of a 16x16 block; the inner sums go through the T scratch (reusing the
  // strictly upper part of W's LDS image is avoided: T lives in the im plane of nothing -> use regs + LDS row).
  R* Tre = Wre + kPD * kPDLd * (CX ? 2 : 1);  // 3 x 16 x 16 scratch (allocated behind W)
  R* Tim = Tre + 3 * kPB * kPB;
  const int ti = t % kPB, tj = t / kPB;  // element (ti, tj) of a block
  for (int d = 1; d < kPD / kPB; ++d) {
    const int nblk = kPD / kPB - d;
    for (int j = 0; j < nblk; ++j) {
      const int i = j + d;
      const int r0 = i * kPB, c0 = j * kPB;
      R sre = 0, sim = 0;
      for (int k = c0; k < r0; ++k) {  // k runs over columns of L(i, j..i-1) = rows of W(j..i-1, j)
        const R l_re = Lre[k * kPDLd + r0 + ti];
        const R w_re = diag_w_get<PACK>(Wre, wd_re, k, c0 + tj);
        if constexpr (CX) {
          const R l_im = Lim[k * kPDLd + r0 + ti];
          const R w_im = diag_w_get<PACK>(Wim, wd_im, k, c0 + tj);
          sre += l_re * w_re - l_im * w_im;
          sim += l_re * w_im + l_im * w_re;
        }
        else {
          sre += l_re * w_re;
        }
      }
      Tre[j * kPB * kPB + tj * kPB + ti] = sre;
      if constexpr (CX)
        Tim[j * kPB * kPB + tj * kPB + ti] = sim;
    }
    __syncthreads();
    for (int j = 0; j < nblk; ++j) {
      const int i = j + d;
      const int r0 = i * kPB, c0 = j * kPB;
      R sre = 0, sim = 0;
#pragma unroll
      for (int k = 0; k < kPB; ++k) {
        const R w_re = diag_w_get<PACK>(Wre, wd_re, r0 + ti, r0 + k);  // W(i,i)[ti][k]
        const R t_re = Tre[j * kPB * kPB + tj * kPB + k];
        if constexpr (CX) {
          const R w_im = diag_w_get<PACK>(Wim, wd_im, r0 + ti, r0 + k);
          const R t_im = Tim[j * kPB * kPB + tj * kPB + k];
          sre += w_re * t_re - w_im * t_im;
          sim += w_re * t_im + w_im * t_re;
        }
        else {
          sre += w_re * t_re;
        }
      }
      diag_w_set<PACK>(Wre, wd_re, r0 + ti, c0 + tj, -sre);
      if constexpr (CX)
        diag_w_set<PACK>(Wim, wd_im, r0 + ti, c0 + tj, -sim);
    }
    __syncthreads();
  }

  return -1;
}

}  // namespace dlaf_mi355x
