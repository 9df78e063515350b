// kernels_trsm.hip -- panel TRSM  X = B * L^-H  (Right, Lower, ConjTrans, NonUnit) for all the
// local tiles of a panel in ONE launch.
//
// Reference: cholesky/impl.h:56-67 (trsmPanelTile), issued per tile at impl.h:162-166 / :254-262
// through rocblas_*trsm (blas/tile.h:458-470).  Here each workgroup owns a 128-row strip of one
// panel tile and sweeps the n columns left to right in 64-wide blocks:
//     Y_j = B_j - sum_{p<j} X_p L_{j,p}^H        (MFMA GEMM, K = 64 j, panels staged through LDS)
//     X_j = Y_j * inv(L_jj)^H                    (MFMA, Y fed straight from the accumulators)
// inv(L_jj) are the 64x64 inverted diagonal blocks the diagonal POTRF leaves behind (the same
// "inverted diagonal block" scheme vendor trsm uses).  One read of B, one write of X per strip:
// algorithmic HBM bytes = (n^2/2 + 2 rows n) sizeof(T) per tile -- this is the kernel whose
// achieved GB/s is reported next to its n^2 rows flops.
#include <type_traits>

#include "device_api.hpp"
#include "mma_core.hpp"

namespace dlaf_mi355x {

template <class T>
struct TrsmCfg {
  using type = BlockCfg<T, 128, kDiagBlock, 32, kDiagBlock, 16>;
  static constexpr int min_waves = 2;
};
// (A paired-row, direct-to-LDS, 3-stage configuration -- BlockCfg<double, 128, 64, 32, 64, 16, true, 3> -- is
// supported by this kernel and was measured: no gain.  A strip is a chain of 16 dependent block steps run by
// four waves; its floor is the MFMA time of that chain plus ~7 us per step of exposed B_j / inv(L_jj) loads,
// not the K loop's global latency.)
template <>
struct TrsmCfg<cdouble> {
  using type = BlockCfg<cdouble, 128, kDiagBlock, 32, kDiagBlock, 8>;
  static constexpr int min_waves = 1;
};

template <class T>
constexpr int trsm_lds_bytes() {
  using Cfg = typename TrsmCfg<T>::type;
  constexpr int w = (TypeInfo<T>::is_complex ? 2 : 1) * kDiagBlock * (kDiagBlock + kLdsPad) * (int) sizeof(real_t<T>);
  return Cfg::LDS_BYTES > w ? Cfg::LDS_BYTES : w;
}

// UPPER: X U^H = B with U upper triangular: B_j = sum_{p>=j} X_p U_{j,p}^H, so the blocks are swept right to
// left, the K range of block j is the columns right of it, and inv(U_jj) is upper triangular (the
// triangular solver's Upper / transposed-Lower variants, solver/triangular/impl.h).
template <class T, bool VEC, bool UPPER>
__global__ __launch_bounds__(kThreads, TrsmCfg<T>::min_waves) void trsm_kernel(TrsmArgs<T> p, int spt) {
  using Cfg = typename TrsmCfg<T>::type;
  using R = real_t<T>;
  using acc_t = typename Mma<R>::acc_t;
  constexpr int JB = kDiagBlock;
  static_assert(Cfg::BN == JB && Cfg::WAVES_N == 1, "a wave must own whole rows of the strip");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);

  if (*p.info != 0)
    return;
  const int il = p.il0 + blockIdx.x / spt;
  const int s = blockIdx.x % spt;
  const int gi = il * p.pr + p.ri;
  const int rows_tile = (gi == p.nt - 1) ? p.last_rows : p.nb;
  const int m0 = s * Cfg::BM;
  if (m0 >= rows_tile)
    return;
  const int mrows = min(Cfg::BM, rows_tile - m0);
  T* Bst = p.b + (long) (il - p.il0) * p.b_ts + m0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave;  // WAVES_M == 4, WAVES_N == 1
  const int g = lane >> 4, c = lane & 15;
  const int njb = (p.n + JB - 1) / JB;

  for (int jj = 0; jj < njb; ++jj) {
    const int j = UPPER ? njb - 1 - jj : jj;
    const int jb = min(JB, p.n - j * JB);
    const int k0 = UPPER ? (j + 1) * JB : 0;            // first column of the already solved part
    const int K = UPPER ? max(0, p.n - k0) : j * JB;
    const bool full = (mrows == Cfg::BM) && (jb == JB) && (K % Cfg::BK == 0);
    Acc<Cfg> y;
    y.clear();
    if (K > 0) {
      const T* Xs = Bst + (long) k0 * p.ldb;
      const T* Lj = p.l + j * JB + (long) k0 * p.ldl;
      if (full)
        gemm_nt_block<Cfg, T, VEC, false>(Xs, p.ldb, mrows, Lj, p.ldl, jb, K, lds, y);
      else
        gemm_nt_block<Cfg, T, false, true>(Xs, p.ldb, mrows, Lj, p.ldl, jb, K, lds, y);
    }

    // ---- Y = B_j - acc (C layout: lane holds m = wm*32 + i*16 + c, n = j2*16 + irow(g,v)) ------
    T* Bj = Bst + (long) (j * JB) * p.ldb;
#pragma unroll
    for (int j2 = 0; j2 < Cfg::TN; ++j2)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int nl = acc_n<Cfg>(j2, g, v);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          const int ml = wm * Cfg::WM + acc_m<Cfg>(i, c);
          T bv = zero_el<T>();
          if (full || (ml < mrows && nl < jb))
            bv = Bj[ml + (long) nl * p.ldb];
          y.re[i][j2][v] = re_of(bv) - y.re[i][j2][v];
          if constexpr (Cfg::CX)
            y.im[i][j2][v] = im_of(bv) - y.im[i][j2][v];
        }
      }

    // ---- X = Y * W_j^H: W_j = inv(L_jj) staged whole into LDS, X produced and stored one 16-column
    // tile at a time (highest first: Y tile ct only feeds X tiles j2 >= ct) so that Y and a single X
    // tile column are all that is live in registers
    const T* Wj = p.winv + (long) j * JB * JB;
    constexpr int LDW = JB + kLdsPad;
    constexpr int WPLANE = JB * LDW;
    R* Ws = lds;  // [k][JB + pad] (+ im plane)
    {
      Slab<T, JB, JB, true, LDW> sw;  // winv blocks are dense 64x64, 16-byte aligned
      sw.template load<false>(Wj, JB, 0, JB, JB);
      sw.store(Ws);
    }
    __syncthreads();
#pragma unroll
    for (int jx = 0; jx < Cfg::TN; ++jx) {
      // lower W: highest X tile first (Y tile ct feeds X tiles j2 >= ct); upper W: lowest first
      const int j2 = UPPER ? jx : Cfg::TN - 1 - jx;
      acc_t xre[Cfg::TM], xim[Cfg::TM];
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        xre[i] = acc_t{0, 0, 0, 0};
        xim[i] = acc_t{0, 0, 0, 0};
      }
#pragma unroll
      for (int ct = 0; ct < Cfg::TN; ++ct) {
        // W lower triangular: W[n2][k] = 0 for k > n2;  upper: = 0 for k < n2.  X tile j2 holds the natural
        // columns 16 j2 .. 16 j2 + 15; with the paired-row map Y tile ct holds columns of the 32-wide group
        // ct / 2, so whole groups are skipped (the zeros inside a group are multiplied)
        if constexpr (Cfg::PAIRED) {
          if (UPPER ? ((ct >> 1) < (j2 >> 1)) : ((ct >> 1) > (j2 >> 1)))
            continue;
        }
        else {
          if (UPPER ? (ct < j2) : (ct > j2))
            continue;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          // accumulator register v of lane-group g holds k = 16*ct + irow(g, v): read the partner
          // fragment of W at exactly that k ("accumulator as next operand")
          const int kabs = acc_n<Cfg>(ct, g, v);
          const R w_re = Ws[kabs * LDW + j2 * 16 + c];
          R w_im = R(0);
          if constexpr (Cfg::CX)
            w_im = Ws[WPLANE + kabs * LDW + j2 * 16 + c];
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            xre[i] = Mma<R>::mma(w_re, y.re[i][ct][v], xre[i]);
            if constexpr (Cfg::CX) {
              // (yr + i yi)(wr - i wi) = (yr wr + yi wi) + i (yi wr - yr wi)
              xre[i] = Mma<R>::mma(w_im, y.im[i][ct][v], xre[i]);
              xim[i] = Mma<R>::mma(w_re, y.im[i][ct][v], xim[i]);
              xim[i] = Mma<R>::mma(-w_im, y.re[i][ct][v], xim[i]);
            }
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int nl = j2 * 16 + Mma<R>::irow(g, v);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          const int ml = wm * Cfg::WM + acc_m<Cfg>(i, c);
          if (full || (ml < mrows && nl < jb)) {
            if constexpr (Cfg::CX)
              Bj[ml + (long) nl * p.ldb] = T{xre[i][v], xim[i][v]};
            else
              Bj[ml + (long) nl * p.ldb] = xre[i][v];
          }
        }
      }
    }
    // X_j is read back (through L1/L2) by this workgroup's next K loop
    __syncthreads();
  }
}

template <class T>
static bool aligned16(const void* ptr, long stride_elems) {
  return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && ((stride_elems * (long) sizeof(T)) % 16 == 0);
}

template <class T>
void launch_trsm(const TrsmArgs<T>& a, hipStream_t stream) {
  using Cfg = typename TrsmCfg<T>::type;
  if (a.il1 <= a.il0 || a.n <= 0 || a.nb <= 0)
    return;
  const int spt = (a.nb + Cfg::BM - 1) / Cfg::BM;
  const long grid = (long) (a.il1 - a.il0) * spt;
  const bool vec = aligned16<T>(a.b, a.ldb) && aligned16<T>(a.b, a.b_ts) && aligned16<T>(a.l, a.ldl);
  auto go = [&](auto vtag, auto utag) {
    hipLaunchKernelGGL((trsm_kernel<T, decltype(vtag)::value, decltype(utag)::value>), dim3((unsigned) grid),
                       dim3(kThreads), trsm_lds_bytes<T>(), stream, a, spt);
  };
  if (a.upper)
    vec ? go(std::true_type{}, std::true_type{}) : go(std::false_type{}, std::true_type{});
  else
    vec ? go(std::true_type{}, std::false_type{}) : go(std::false_type{}, std::false_type{});
}

template <class T>
static void trsm_init_one() {

#define SET_ONE(V, U)                                                                      \
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_kernel<T, V, U>),          \
                             hipFuncAttributeMaxDynamicSharedMemorySize, trsm_lds_bytes<T>())
  SET_ONE(true, false);
  SET_ONE(false, false);
  SET_ONE(true, true);
  SET_ONE(false, true);
#undef SET_ONE
}

void trsm_kernels_init() {
  trsm_init_one<float>();
  trsm_init_one<double>();
  trsm_init_one<cfloat>();
  trsm_init_one<cdouble>();
}

template void launch_trsm<float>(const TrsmArgs<float>&, hipStream_t);
template void launch_trsm<double>(const TrsmArgs<double>&, hipStream_t);
template void launch_trsm<cfloat>(const TrsmArgs<cfloat>&, hipStream_t);
template void launch_trsm<cdouble>(const TrsmArgs<cdouble>&, hipStream_t);

}  // namespace dlaf_mi355x
