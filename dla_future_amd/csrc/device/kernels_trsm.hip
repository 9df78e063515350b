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
#include <cstdlib>
#include <cstring>
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

// ------------------------------------------------------------------------------------------------
// "Row-owner" panel TRSM (fp64, full tiles): the round-2 form of the kernel above for the shapes the
// factorization issues at its headline sizes.
//
// A workgroup owns a 64-row strip of one panel tile, wave w its rows 16w .. 16w+15 -- rows of X are
// independent in X L^H = B, so the four waves never exchange data; the workgroup only shares the L
// slabs it stages into LDS.  The n columns are swept in MACRO blocks of NW columns whose running sums
//      S = sum_{k < c} X[:,k] L[c0 .. c0+NW, k]^H           (16 x NW per wave, MFMA accumulators)
// stay in registers for the whole macro block:
//   P1   columns k left of the macro block: X comes back from global memory through LDS (the only
//        re-read of X: (n/NW - 1)/2 of the panel instead of (n/64 - 1)/2 with 64-column steps);
//   per 64-column sub-block s of the macro block:
//     SOLVE  X_s = (B_s - S_s) inv(L_ss)^H with the MFMA accumulators fed straight back as operands
//            (f64: accumulator register v of lane group g holds exactly the k index the next MFMA wants)
//     UPD    S_t += X_s L[t, s]^H for the sub-blocks t > s, X_s again taken from the accumulators --
//            the solved block never goes through memory on its way into the update.
// L streams through a ring of ST direct-to-LDS stages (global_load_lds_dwordx4, counted s_waitcnt vmcnt,
// raw s_barrier): stage i = columns 8i .. 8i+7 of L rows [c0, c0+NW) (+ the same 8 columns of the X strip
// for P1), one uniform stream per macro block.  inv(L_ss) (64 x 64, from the diagonal kernel) has its own
// buffer, refilled one sub-block ahead.  The paired-row fragment map of mma_core.hpp (tiles 2q / 2q+1 =
// even / odd columns of a 32-column group, one ds_read_b128 per pair) is used for every L / inv(L) read.
// HBM-side traffic per tile: B read once, X written once, X re-read (n/NW - 1)/2 times, L from L2.
template <int NW_, int ST_>
struct TrsmRowsCfg {
  static constexpr int NW = NW_, ST = ST_;
  static constexpr int ROWS = 64, BK = 8;
  static constexpr int NT = NW / 16, NS = NW / kDiagBlock;
  static constexpr int A_ELEMS = BK * ROWS;        // X strip part of a stage: image [8][64]
  static constexpr int B_ELEMS = BK * NW;          // L part: image [8][NW]
  static constexpr int STAGE = A_ELEMS + B_ELEMS;
  static constexpr int W_ELEMS = kDiagBlock * kDiagBlock;
  static constexpr int LDS_BYTES = (ST * STAGE + W_ELEMS) * (int) sizeof(double);
  static constexpr int LB = B_ELEMS / 128 / 4;     // 1 KiB pieces of the L part per wave
  static constexpr int LPS = 1 + LB;               // direct-to-LDS loads per wave and stage
  static constexpr int WAVES_PER_SIMD = (NW <= 256 && LDS_BYTES <= 80 * 1024) ? 2 : 1;
  static_assert(kThreads == 256 && kDiagBlock == 64 && NW % 128 == 0 && LPS * (ST - 2) < 64, "geometry");
};

__device__ __forceinline__ void glds16(const double* src, double* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) src,
                                   (__attribute__((address_space(3))) void*) lds_dst, 16, 0, 0);
}
// non-temporal form (aux = 2) for data a workgroup streams once -- the X strip coming back in P1: it must not push
// the L tile, which every workgroup of the XCD re-reads, out of the L2
__device__ __forceinline__ void glds16_nt(const double* src, double* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) src,
                                   (__attribute__((address_space(3))) void*) lds_dst, 16, 0, 2);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 0xF) | (((N >> 4) & 0x3) << 14));  // lgkmcnt / expcnt: no wait
}

template <int NW, int ST>
__global__ __launch_bounds__(kThreads, (TrsmRowsCfg<NW, ST>::WAVES_PER_SIMD)) void trsm_rows_kernel(TrsmArgs<double> p,
                                                                                                   int spt) {
  using C = TrsmRowsCfg<NW, ST>;
  typedef double acc_t __attribute__((ext_vector_type(4)));
  typedef double d2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* lds = reinterpret_cast<double*>(lds_raw);
  double* Wbuf = lds + ST * C::STAGE;

  if (*p.info != 0)
    return;
  if (p.prio)
    __builtin_amdgcn_s_setprio(2);
  const int il = p.il0 + blockIdx.x / spt;
  const int strip = blockIdx.x % spt;
  const int gi = il * p.pr + p.ri;
  const int rows_tile = (gi == p.nt - 1) ? p.last_rows : p.nb;
  const int m0 = strip * C::ROWS;
  if (m0 >= rows_tile)
    return;  // (the launcher promises rows_tile % 64 == 0)
  double* Bst = p.b + (long) (il - p.il0) * p.b_ts + m0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps address math scalar
  const int g = lane >> 4, c = lane & 15;
  const int ldb = p.ldb, ldl = p.ldl;
  const int mrow = 16 * wave + c;  // the strip row this lane's accumulators stand for
  // Addresses are (wave-uniform base) + (small per-lane offset), so that the bases live in scalar registers.
  // Column (inside a 64-column sub-block) of accumulator register v of tile t4 = 0..3: 32 (t4>>1) + 8 v + (t4&1)
  // is uniform, 2 g is the lane's share.
  // per-lane BYTE offsets (32-bit: they select the scalar-base + vector-offset addressing form)
  const unsigned bx_lane = 8u * (unsigned) (mrow + 2 * g * ldb);            // B / X element of this lane
  const unsigned a_lane = 8u * (unsigned) (((2 * lane) & 63) + (lane >> 5) * ldb);  // X-strip piece: image [8][64]
  const unsigned l_lane = 16u * (unsigned) lane;                           // L / W pieces: 128 consecutive elements
  constexpr int PPC = NW / 128;                                            // pieces per column of the L image
  auto at = [](const double* base, unsigned byte_off) {
    return reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
  };

  // with_x: the stage also carries the 8 columns of the X strip (P1 stages; the UPD stages take X from registers)
  auto issue_stage = [&](int c0, int i, int slot, bool with_x) {
    double* buf = lds + slot * C::STAGE;
    const int col0 = 8 * i;
    if (with_x) {
      const double* abase = Bst + (long) (col0 + 2 * wave) * ldb;
      glds16_nt(at(abase, a_lane), buf + 128 * wave);
    }
#pragma unroll
    for (int q = 0; q < C::LB; ++q) {
      const int piece = wave * C::LB + q;
      const double* lbase = p.l + (c0 + 128 * (piece % PPC)) + (long) (col0 + piece / PPC) * ldl;
      glds16(at(lbase, l_lane), buf + C::A_ELEMS + 128 * piece);
    }
  };
  auto load_w = [&](int jblk) {  // inv(L_jj): dense 64 x 64 column-major = image [k][64], 8 pieces per wave
    const double* W = p.winv + (long) jblk * C::W_ELEMS + 1024 * wave;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      glds16(at(W + 128 * q, l_lane), Wbuf + 1024 * wave + 128 * q);
  };

  acc_t S[C::NT];
  acc_t Breg[4], X[4];
  auto load_b = [&](int col0) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const double* base = Bst + (long) (col0 + 32 * (t4 >> 1) + 8 * v + (t4 & 1)) * ldb;
        Breg[t4][v] = __builtin_nontemporal_load(at(base, bx_lane));
      }
  };

  const int nmacro = p.n / NW;
  for (int J = 0; J < nmacro; ++J) {
    const int c0 = NW * J;                        // first column of the macro block
    const int P = c0 / C::BK;                     // P1 stages
    const int NSTG = P + 8 * (C::NS - 1);         // + 8 stages per sub-block but the last
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
      S[t] = acc_t{0, 0, 0, 0};
    // everything of the previous macro block is done and visible (its X columns are P1 operands now)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_w(c0 / kDiagBlock);
    load_b(c0);
#pragma unroll
    for (int st = 0; st < ST - 1; ++st)
      if (st < NSTG)
        issue_stage(c0, st, st, st < P);
    if (NSTG >= ST - 1)
      wait_vmcnt<C::LB*(ST - 2)>();               // stage 0 (and the older W / B loads) landed
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int slot = 0, nslot = ST - 1;
    auto ring_step_end = [&](int i) {
      // stage i + 1 landed?  loads allowed to stay in flight: stages i+2 .. i+ST-1 when they exist
      // (count by the SMALLEST stage, LB loads per wave: never lets a stage that must have landed stay in flight)
      if (i + ST - 1 < NSTG)
        wait_vmcnt<C::LB*(ST - 2)>();
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      slot = (slot + 1 == ST) ? 0 : slot + 1;
      nslot = (nslot + 1 == ST) ? 0 : nslot + 1;
    };

    // ---- P1: S += X[:, 8i .. 8i+7] L[c0 .. c0+NW, 8i .. 8i+7]^H --------------------------------------
    for (int i = 0; i < P; ++i) {
      if (i + ST - 1 < NSTG)
        issue_stage(c0, i + ST - 1, nslot, i + ST - 1 < P);
      const double* buf = lds + slot * C::STAGE;
#pragma unroll
      for (int k4 = 0; k4 < 2; ++k4) {
        const int kk = 4 * k4 + g;
        const double xf = buf[kk * C::ROWS + mrow];
#pragma unroll
        for (int q = 0; q < C::NT / 2; ++q) {
          const d2 lf = *reinterpret_cast<const d2*>(&buf[C::A_ELEMS + kk * NW + 32 * q + 2 * c]);
          S[2 * q] = Mma<double>::mma(lf[0], xf, S[2 * q]);
          S[2 * q + 1] = Mma<double>::mma(lf[1], xf, S[2 * q + 1]);
        }
      }
      ring_step_end(i);
    }

    // ---- the macro block itself, one 64-column sub-block at a time ---------------------------------------
    // (unrolled: every accumulator index below is a compile-time constant, every trip count static)
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
      const int cs = c0 + kDiagBlock * s;
      // SOLVE: X_s = (B_s - S_s) W^H, W = inv(L_ss) lower triangular
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          Breg[t4][v] -= S[4 * s + t4][v];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
        X[t4] = acc_t{0, 0, 0, 0};
      // k outermost: the X tiles of both column groups advance together (four independent accumulator chains
      // while h = 0); the fragment reads of four k steps are issued as one batch ahead of their MFMAs
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          d2 wf[2][4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            // accumulator register v of tile 2h+b holds column k of the sub-block: read W[:, k]
            const int k = 32 * h + 2 * (g + 4 * v) + b;
#pragma unroll
            for (int qp = h; qp < 2; ++qp)
              wf[qp][v] = *reinterpret_cast<const d2*>(&Wbuf[k * kDiagBlock + 32 * qp + 2 * c]);
          }
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int qp = h; qp < 2; ++qp) {
              X[2 * qp] = Mma<double>::mma(wf[qp][v][0], Breg[2 * h + b][v], X[2 * qp]);
              X[2 * qp + 1] = Mma<double>::mma(wf[qp][v][1], Breg[2 * h + b][v], X[2 * qp + 1]);
            }
          if (h == 0) {
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // the batch of LDS reads ...
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // ... then its MFMAs
          }
          else {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
          }
        }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          double* base = Bst + (long) (cs + 32 * (t4 >> 1) + 8 * v + (t4 & 1)) * ldb;
          *const_cast<double*>(at(base, bx_lane)) = X[t4][v];
        }
      if (s == C::NS - 1)
        break;
      // every wave is done with W: refill it for the next sub-block, fetch the next B block
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      load_w(cs / kDiagBlock + 1);
      load_b(cs + kDiagBlock);
      // UPD: S_t += X_s L[t, s]^H for the sub-blocks t > s: accumulator pairs qmin .. NT/2-1
      const int qmin = 2 * (s + 1);
#pragma unroll
      for (int uu = 0; uu < 8; ++uu) {
        const int i = P + 8 * s + uu;
        if (i + ST - 1 < NSTG)
          issue_stage(c0, i + ST - 1, nslot, false);
        const double* buf = lds + slot * C::STAGE + C::A_ELEMS;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          // stage columns 8uu + 2g + b of the sub-block <-> accumulator register uu & 3 of tile 2 (uu >> 2) + b
          const double xf = X[2 * (uu >> 2) + b][uu & 3];
          const int kk = 2 * g + b;
          // fragment reads of the step as one batch ahead of its MFMAs
          d2 lf[C::NT / 2];
#pragma unroll
          for (int q = qmin; q < C::NT / 2; ++q)
            lf[q] = *reinterpret_cast<const d2*>(&buf[kk * NW + 32 * q + 2 * c]);
#pragma unroll
          for (int q = qmin; q < C::NT / 2; ++q) {
            S[2 * q] = Mma<double>::mma(lf[q][0], xf, S[2 * q]);
            S[2 * q + 1] = Mma<double>::mma(lf[q][1], xf, S[2 * q + 1]);
          }
          if constexpr (NW == 256) {
            if (s == 0) {
              __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
            else if (s == 1) {
              __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            else {
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
          }
        }
        ring_step_end(i);
      }
    }
  }
}

}  // namespace dlaf_mi355x
#include "trsm_rows_z.hpp"
namespace dlaf_mi355x {

template <class T>
static bool aligned16(const void* ptr, long stride_elems) {
  return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && ((stride_elems * (long) sizeof(T)) % 16 == 0);
}

// tuning knob DLAF_MI355X_TRSM=strips selects the strips kernel everywhere (A/B runs, fallback)
static bool trsm_rows_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("DLAF_MI355X_TRSM");
    return !(e && std::strcmp(e, "strips") == 0);
  }();
  return on;
}

template <class T>
void launch_trsm(const TrsmArgs<T>& a, hipStream_t stream) {
  using Cfg = typename TrsmCfg<T>::type;
  if (a.il1 <= a.il0 || a.n <= 0 || a.nb <= 0)
    return;
  const bool vec = aligned16<T>(a.b, a.ldb) && aligned16<T>(a.b, a.b_ts) && aligned16<T>(a.l, a.ldl);
  if constexpr (std::is_same<T, double>::value) {
    // row-owner kernel: whole 64-row strips, n a multiple of the macro width, 16-byte aligned operands
    // (a 512-column macro block -- 256 accumulator registers per lane -- does not survive the register
    // allocator: spills inside the loops; 256 columns run at two waves per SIMD without any)
    constexpr int NW = 256;
    if (vec && !a.upper && trsm_rows_enabled() && a.nb % 64 == 0 && a.last_rows % 64 == 0 && a.n % NW == 0 &&
        aligned16<T>(a.winv, 0)) {
      const int spt = a.nb / 64;
      const long grid = (long) (a.il1 - a.il0) * spt;
      hipLaunchKernelGGL((trsm_rows_kernel<NW, 2>), dim3((unsigned) grid), dim3(kThreads), (TrsmRowsCfg<NW, 2>::LDS_BYTES),
                         stream, a, spt);
      return;
    }
    // the same kernel with 128-column macro blocks for the widths that are whole 128s but not whole 256s (the
    // tall-skinny solves of the blocked panel factorization of reduction_to_band: n = band = 128)
    constexpr int NW2 = 128;
    if (vec && !a.upper && trsm_rows_enabled() && a.nb % 64 == 0 && a.last_rows % 64 == 0 && a.n % NW2 == 0 &&
        aligned16<T>(a.winv, 0)) {
      const int spt = a.nb / 64;
      const long grid = (long) (a.il1 - a.il0) * spt;
      hipLaunchKernelGGL((trsm_rows_kernel<NW2, 2>), dim3((unsigned) grid), dim3(kThreads), (TrsmRowsCfg<NW2, 2>::LDS_BYTES),
                         stream, a, spt);
      return;
    }
  }
  if constexpr (std::is_same<T, cdouble>::value) {
    constexpr int NW = 128;
    if (vec && !a.upper && trsm_rows_enabled() && a.nb % 64 == 0 && a.last_rows % 64 == 0 && a.n % NW == 0 &&
        aligned16<T>(a.winv, 0)) {
      const int spt = a.nb / 64;
      const long grid = (long) (a.il1 - a.il0) * spt;
      hipLaunchKernelGGL((trsm_rows_z_kernel<NW, 3>), dim3((unsigned) grid), dim3(kThreads),
                         (TrsmRowsZCfg<NW, 3>::LDS_BYTES), stream, a, spt);
      return;
    }
  }
  const int spt = (a.nb + Cfg::BM - 1) / Cfg::BM;
  const long grid = (long) (a.il1 - a.il0) * spt;
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
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_rows_z_kernel<128, 3>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, TrsmRowsZCfg<128, 3>::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_rows_kernel<256, 2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, TrsmRowsCfg<256, 2>::LDS_BYTES);
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_rows_kernel<128, 2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, TrsmRowsCfg<128, 2>::LDS_BYTES);
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
