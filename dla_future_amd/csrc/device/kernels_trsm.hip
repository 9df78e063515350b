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
#include "device_api.hpp"
#include "mma_core.hpp"

namespace dlaf_mi355x {

template <class T>
struct TrsmCfg {
  using type = BlockCfg<T, 128, kDiagBlock, 32, kDiagBlock, 16>;
  // Y and X accumulators are live together in the second product: give the allocator the whole
  // 512-register file (1 wave/SIMD) instead of spilling at 256
  static constexpr int min_waves = 1;
};
template <>
struct TrsmCfg<cdouble> {
  using type = BlockCfg<cdouble, 128, kDiagBlock, 32, kDiagBlock, 8>;
  static constexpr int min_waves = 1;
};

template <class T, bool VEC>
__global__ __launch_bounds__(kThreads, TrsmCfg<T>::min_waves) void trsm_kernel(TrsmArgs<T> p, int spt) {
  using Cfg = typename TrsmCfg<T>::type;
  using R = real_t<T>;
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

  for (int j = 0; j < njb; ++j) {
    const int jb = min(JB, p.n - j * JB);
    const int K = j * JB;
    const bool full = (mrows == Cfg::BM) && (jb == JB);
    Acc<Cfg> y;
    y.clear();
    if (full)
      gemm_nt_block<Cfg, T, VEC, false>(Bst, p.ldb, mrows, p.l + j * JB, p.ldl, jb, K, lds, y);
    else
      gemm_nt_block<Cfg, T, false, true>(Bst, p.ldb, mrows, p.l + j * JB, p.ldl, jb, K, lds, y);

    // ---- Y = B_j - acc (C layout: lane holds m = wm*32 + i*16 + c, n = j2*16 + irow(g,v)) ------
    T* Bj = Bst + (long) (j * JB) * p.ldb;
#pragma unroll
    for (int j2 = 0; j2 < Cfg::TN; ++j2)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int nl = j2 * 16 + Mma<R>::irow(g, v);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          const int ml = wm * Cfg::WM + i * 16 + c;
          T bv = zero_el<T>();
          if (full || (ml < mrows && nl < jb))
            bv = Bj[ml + (long) nl * p.ldb];
          y.re[i][j2][v] = re_of(bv) - y.re[i][j2][v];
          if constexpr (Cfg::CX)
            y.im[i][j2][v] = im_of(bv) - y.im[i][j2][v];
        }
      }

    // ---- X = Y * W_j^H, W_j = inv(L_jj) staged through LDS in BK-wide k chunks ----------------
    Acc<Cfg> x;
    x.clear();
    const T* Wj = p.winv + (long) j * JB * JB;
    Slab<T, JB, Cfg::BK, true> sw;  // winv blocks are dense 64x64, 16-byte aligned
    R* Ws = lds;                    // [k][JB + pad] (+ im plane)
    constexpr int LDW = JB + kLdsPad;
    constexpr int WPLANE = Cfg::BK * LDW;
#pragma unroll
    for (int kc = 0; kc < JB / Cfg::BK; ++kc) {
      sw.template load<false>(Wj, JB, kc * Cfg::BK, JB, JB);
      sw.store(Ws);
      __syncthreads();
      constexpr int kTilesPerChunk = (Cfg::BK >= 16) ? Cfg::BK / 16 : 1;
      static_assert(Cfg::BK == 16 || Cfg::BK == 8, "chunk must be one or half a 16-wide k tile");
      const int ct = (kc * Cfg::BK) / 16;  // k tile (16 wide) this chunk belongs to (static after unrolling)
      (void) kTilesPerChunk;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        // accumulator register v of lane-group g holds k = 16*ct + irow(g, v).  With BK == 8 (f64
        // only: irow = g + 4v) registers {0,1} live in the first half chunk and {2,3} in the second.
        if (Cfg::BK == 8 && (v >> 1) != (kc & 1))
          continue;
        const int kloc = 16 * ct + Mma<R>::irow(g, v) - kc * Cfg::BK;
#pragma unroll
        for (int j2 = 0; j2 < Cfg::TN; ++j2) {
          if (j2 < ct)
            continue;  // W is lower triangular: W[n2][k] = 0 for k > n2
          const R w_re = Ws[kloc * LDW + j2 * 16 + c];
          R w_im = R(0);
          if constexpr (Cfg::CX)
            w_im = Ws[WPLANE + kloc * LDW + j2 * 16 + c];
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            x.re[i][j2] = Mma<R>::mma(w_re, y.re[i][ct][v], x.re[i][j2]);
            if constexpr (Cfg::CX) {
              // (yr + i yi)(wr - i wi) = (yr wr + yi wi) + i (yi wr - yr wi)
              x.re[i][j2] = Mma<R>::mma(w_im, y.im[i][ct][v], x.re[i][j2]);
              x.im[i][j2] = Mma<R>::mma(w_re, y.im[i][ct][v], x.im[i][j2]);
              x.im[i][j2] = Mma<R>::mma(-w_im, y.re[i][ct][v], x.im[i][j2]);
            }
          }
        }
      }
      __syncthreads();
    }

    // ---- store X_j ------------------------------------------------------------------------------
#pragma unroll
    for (int j2 = 0; j2 < Cfg::TN; ++j2)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int nl = j2 * 16 + Mma<R>::irow(g, v);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          const int ml = wm * Cfg::WM + i * 16 + c;
          if (full || (ml < mrows && nl < jb)) {
            if constexpr (Cfg::CX)
              Bj[ml + (long) nl * p.ldb] = T{x.re[i][j2][v], x.im[i][j2][v]};
            else
              Bj[ml + (long) nl * p.ldb] = x.re[i][j2][v];
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
  if (vec)
    hipLaunchKernelGGL((trsm_kernel<T, true>), dim3((unsigned) grid), dim3(kThreads), Cfg::LDS_BYTES, stream, a, spt);
  else
    hipLaunchKernelGGL((trsm_kernel<T, false>), dim3((unsigned) grid), dim3(kThreads), Cfg::LDS_BYTES, stream, a, spt);
}

template <class T>
static void trsm_init_one() {
  using Cfg = typename TrsmCfg<T>::type;
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_kernel<T, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_kernel<T, false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
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
