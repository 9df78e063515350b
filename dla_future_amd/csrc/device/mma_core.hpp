// mma_core.hpp -- the workgroup-level MFMA GEMM core shared by the trailing-update and the
// panel-TRSM kernels:   acc(BM x BN) += A(BM x K) * B(BN x K)^H
// A, B column-major in global memory; slabs of BK columns are staged global -> registers -> LDS
// (double-buffered, one barrier per slab) and consumed as 16x16x4 MFMA fragments.
#pragma once
#include "common.hpp"

namespace dlaf_mi355x {

#ifdef DLAF_DBG_STAMPS
__device__ unsigned long long g_dbg_stamps[8];
#endif

// PAIRED_ (real types): MFMA tiles 2q and 2q+1 of a wave hold the even / odd rows of a 32-row group,
// so one 16-byte LDS read feeds two fragments and the epilogue moves two consecutive rows per lane
// (16-byte global accesses).  With it the LDS image is unpadded ([k][ROWS], column stride = 0 mod
// 256 B is what ds_read_b128's lane groups want) and, for fp64 with 128-row slabs, a slab column is
// exactly one 1 KiB global_load_lds_dwordx4 -- the direct-to-LDS staging of the fast path.
// STAGES_: LDS slab buffers.  2 = classic double buffering; the direct-to-LDS path uses more so that
// STAGES-1 slabs are in flight (counted s_waitcnt vmcnt + raw s_barrier) and HBM/L2 latency spikes
// do not stall the MFMA stream.
template <class T, int BM_, int BN_, int WM_, int WN_, int BK_, bool PAIRED_ = false, int STAGES_ = 2,
          int THREADS_ = kThreads>
struct BlockCfg {
  using R = real_t<T>;
  static constexpr bool CX = TypeInfo<T>::is_complex;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_;
  static constexpr int THREADS = THREADS_, NWAVES = THREADS_ / 64;
  static constexpr int TM = WM / 16, TN = WN / 16;
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static_assert(WAVES_M * WAVES_N == NWAVES, "the wave tiles must cover the block");
  static_assert(BK % 4 == 0 && WM % 16 == 0 && WN % 16 == 0, "MFMA 16x16x4 granularity");
  // For complex types PAIRED_ selects the interleaved LDS image instead ([k][ROWS] of (re, im)): one
  // ds_read_b128 per fragment and, with 64-row granularity, direct-to-LDS staging of 16-byte elements.
  static constexpr bool PAIRED = PAIRED_ && !CX;
  static constexpr bool CXI = PAIRED_ && CX;
  static_assert(!PAIRED || (TM % 2 == 0 && TN % 2 == 0), "paired rows: even tile counts");
  static constexpr int PAD = PAIRED_ ? 0 : kLdsPad;
  static constexpr int LDA = BM + PAD, LDB = BN + PAD;
  // direct-to-LDS staging: one wave instruction (64 lanes x 16 B = 1 KiB, contiguous in LDS) moves one PIECE
  // of the unpadded image [k][ROWS]: EPP consecutive elements = a run of whole columns (ROWS <= EPP) or a
  // part of one column (ROWS > EPP); every wave issues the same number of them per slab
  static constexpr int EPP = CX ? 64 : 1024 / (int) sizeof(T);
  static constexpr int PIECES_A = BK * BM / EPP, PIECES_B = BK * BN / EPP;
  static constexpr bool GLDS =
      (PAIRED && (EPP % BM == 0 || BM % EPP == 0) && (EPP % BN == 0 || BN % EPP == 0) && (BK * BM) % EPP == 0 &&
       (BK * BN) % EPP == 0 && PIECES_A % NWAVES == 0 && PIECES_B % NWAVES == 0) ||
      (CXI && sizeof(T) == 16 && BM % 64 == 0 && BN % 64 == 0 && PIECES_A % NWAVES == 0 && PIECES_B % NWAVES == 0);
  static constexpr int LPS = PIECES_A / NWAVES + PIECES_B / NWAVES;  // direct-to-LDS loads per wave and slab
  static constexpr int A_PLANE = BK * LDA, B_PLANE = BK * LDB;
  static constexpr int A_ELEMS = (CX ? 2 : 1) * A_PLANE, B_ELEMS = (CX ? 2 : 1) * B_PLANE;
  static constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS;
  static constexpr int STAGES = STAGES_;
  static_assert(STAGES >= 2 && STAGES <= 6, "2..6 LDS stages");
  static constexpr int LDS_BYTES = STAGES * BUF_ELEMS * (int) sizeof(R);
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
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// hook(p) is called before each group of TM MFMAs (p = k4 * TN + j counts the groups of a slab): the
// direct-to-LDS pipeline issues its loads of a later slab there, one at a time in the shadow of the MFMAs,
// instead of as one block at the head of the iteration (8 x global_load_lds back to back keep the wave from
// issuing MFMAs for ~1k cycles per slab; measured with in-kernel stamps, tools/update_bench.hip)
// NEG: acc -= A B^H instead of += (the accumulators were preloaded with the block the product is subtracted from)
template <class Cfg, bool NEG = false, class Hook = NoHook>
__device__ __forceinline__ void mma_slab(const typename Cfg::R* __restrict__ As,
                                         const typename Cfg::R* __restrict__ Bs, Acc<Cfg>& acc, int wm,
                                         int wn, int lane, Hook&& hook = Hook{}) {
  using R = typename Cfg::R;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int k4 = 0; k4 < Cfg::BK / 4; ++k4) {
    const int kk = k4 * 4 + g;
    R a_re[Cfg::TM], a_im[Cfg::TM], b_re[Cfg::TN], b_im[Cfg::TN];
    if constexpr (Cfg::CXI) {
      typedef R r2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const r2 v = *reinterpret_cast<const r2*>(&As[2 * (kk * Cfg::LDA + wm * Cfg::WM + i * 16 + c)]);
        a_re[i] = v[0];
        a_im[i] = v[1];
      }
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const r2 v = *reinterpret_cast<const r2*>(&Bs[2 * (kk * Cfg::LDB + wn * Cfg::WN + j * 16 + c)]);
        b_re[j] = v[0];
        b_im[j] = v[1];
      }
    }
    else if constexpr (Cfg::PAIRED) {
      typedef R r2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int q = 0; q < Cfg::TM / 2; ++q) {
        const r2 v = *reinterpret_cast<const r2*>(&As[kk * Cfg::LDA + wm * Cfg::WM + q * 32 + 2 * c]);
        a_re[2 * q] = v[0];
        a_re[2 * q + 1] = v[1];
      }
#pragma unroll
      for (int q = 0; q < Cfg::TN / 2; ++q) {
        const r2 v = *reinterpret_cast<const r2*>(&Bs[kk * Cfg::LDB + wn * Cfg::WN + q * 32 + 2 * c]);
        b_re[2 * q] = v[0];
        b_re[2 * q + 1] = v[1];
      }
    }
    else {
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
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      hook(k4 * Cfg::TN + j);
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        // D[i' = n][j' = m]: Aop <- B panel fragment, Bop <- A panel fragment
        if constexpr (!NEG) {
          acc.re[i][j] = Mma<R>::mma(b_re[j], a_re[i], acc.re[i][j]);
          if constexpr (Cfg::CX) {
            // (ar + i ai)(br - i bi) = (ar br + ai bi) + i (ai br - ar bi)
            acc.re[i][j] = Mma<R>::mma(b_im[j], a_im[i], acc.re[i][j]);
            acc.im[i][j] = Mma<R>::mma(b_re[j], a_im[i], acc.im[i][j]);
            acc.im[i][j] = Mma<R>::mma_neg(b_im[j], a_re[i], acc.im[i][j]);  // (f64: the MFMA's own negate bit)
          }
        }
        else {
          acc.re[i][j] = Mma<R>::mma_neg(b_re[j], a_re[i], acc.re[i][j]);
          if constexpr (Cfg::CX) {
            acc.re[i][j] = Mma<R>::mma_neg(b_im[j], a_im[i], acc.re[i][j]);
            acc.im[i][j] = Mma<R>::mma_neg(b_re[j], a_im[i], acc.im[i][j]);
            acc.im[i][j] = Mma<R>::mma(b_im[j], a_re[i], acc.im[i][j]);
          }
        }
      }
    }
  }
}

// Row (m) / column (n) inside the wave tile that accumulator element (tile i or j, lane, register v)
// stands for.  m lives on the lane's low 4 bits, n on the MFMA "i" index irow(g, v).
template <class Cfg>
__device__ __forceinline__ int acc_m(int i, int c) {
  if constexpr (Cfg::PAIRED)
    return (i >> 1) * 32 + 2 * c + (i & 1);
  else
    return i * 16 + c;
}
template <class Cfg>
__device__ __forceinline__ int acc_n(int j, int g, int v) {
  using R = typename Cfg::R;
  if constexpr (Cfg::PAIRED)
    return (j >> 1) * 32 + 2 * Mma<R>::irow(g, v) + (j & 1);
  else
    return j * 16 + Mma<R>::irow(g, v);
}

// Direct-to-LDS staging of one BK slab of A and B (Cfg::GLDS): the image of each panel is cut into 1 KiB
// pieces (one global_load_lds_dwordx4 each), wave w moves a contiguous run of them.
template <class Cfg, class T>
__device__ __forceinline__ void stage_glds(const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb, int k0,
                                           typename Cfg::R* __restrict__ buf, int wave, int lane) {
  constexpr int NA = Cfg::PIECES_A / Cfg::NWAVES, NB = Cfg::PIECES_B / Cfg::NWAVES;  // instructions per wave
  constexpr int QA0 = 0, QA1 = NA, QB0 = 0, QB1 = NB;
  if constexpr (Cfg::CXI) {
    // 16-byte elements: one instruction moves 64 rows of one column; wave w takes every NWAVES-th piece
    constexpr int PA = Cfg::BM / 64, PB = Cfg::BN / 64;
#pragma unroll
    for (int idx = QA0; idx < QA1; ++idx) {
      const int piece = wave + Cfg::NWAVES * idx;
      const int k = piece / PA, part = piece % PA;
      const T* ga = A + part * 64 + lane + (long) (k0 + k) * lda;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) ga,
                                       (__attribute__((address_space(3))) void*) (buf + 2 * (k * Cfg::LDA + part * 64)),
                                       16, 0, 0);
    }
#pragma unroll
    for (int idx = QB0; idx < QB1; ++idx) {
      const int piece = wave + Cfg::NWAVES * idx;
      const int k = piece / PB, part = piece % PB;
      const T* gb = B + part * 64 + lane + (long) (k0 + k) * ldb;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) gb,
          (__attribute__((address_space(3))) void*) (buf + Cfg::A_ELEMS + 2 * (k * Cfg::LDB + part * 64)), 16, 0, 0);
    }
    return;
  }
  constexpr int PER = 16 / (int) sizeof(T);  // elements per lane
  constexpr int EPP = Cfg::EPP;              // elements per piece
#pragma unroll
  for (int q = QA0; q < QA1; ++q) {
    const int e0 = (wave * NA + q) * EPP;    // first element of the piece in the image [k][BM]
    const int e = e0 + lane * PER;
    const T* ga = A + (e % Cfg::BM) + (long) (k0 + e / Cfg::BM) * lda;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) ga,
                                     (__attribute__((address_space(3))) void*) (buf + e0), 16, 0, 0);
  }
#pragma unroll
  for (int q = QB0; q < QB1; ++q) {
    const int e0 = (wave * NB + q) * EPP;
    const int e = e0 + lane * PER;
    const T* gb = B + (e % Cfg::BN) + (long) (k0 + e / Cfg::BN) * ldb;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) gb,
                                     (__attribute__((address_space(3))) void*) (buf + Cfg::A_ELEMS + e0), 16, 0, 0);
  }
}

// The IDX-th of the LPS instructions stage_glds issues for this wave (real types), alone: lets the pipelined
// loop spread them over the MFMA stream.
#ifndef DLAF_GLDS_SCALAR_ADDR
#define DLAF_GLDS_SCALAR_ADDR 1  // 0: the per-lane address arithmetic of rounds 1-2 (A/B, tools/run_ab_saddr.sh)
#endif
template <class Cfg, class T, int IDX>
__device__ __forceinline__ void stage_glds_one(const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb,
                                               int k0, typename Cfg::R* __restrict__ buf, int wave, int lane) {
  static_assert(!Cfg::CXI, "real types");
  constexpr int NA = Cfg::PIECES_A / Cfg::NWAVES, NB = Cfg::PIECES_B / Cfg::NWAVES;
  constexpr int PER = 16 / (int) sizeof(T), EPP = Cfg::EPP;
  // address = (wave-uniform base: scalar registers) + (a per-lane byte offset that does not depend on the slab):
  // a piece is EPP consecutive elements of the image [k][ROWS] starting at element e0 (uniform); lane l moves the
  // PER elements at e0 + l PER, i.e. row (e0 + l PER) % ROWS of column (e0 + l PER) / ROWS.  With ROWS | EPP or
  // EPP | ROWS the lane part is (l PER) % ROWS + ((l PER) / ROWS) ld and the uniform part (e0 % ROWS) + (k0 + e0 /
  // ROWS) ld.  Written per lane -- "(e % ROWS) + (k0 + e / ROWS) * ld" -- every load costs a 64-bit multiply-add
  // per lane in the middle of the MFMA stream.
  auto at = [](const T* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
  };
#if !DLAF_GLDS_SCALAR_ADDR
  if constexpr (IDX < NA) {
    const int e0 = (wave * NA + IDX) * EPP;
    const int e = e0 + lane * PER;
    const T* ga = A + (e % Cfg::BM) + (long) (k0 + e / Cfg::BM) * lda;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) ga,
                                     (__attribute__((address_space(3))) void*) (buf + e0), 16, 0, 0);
  }
  else if constexpr (IDX < NA + NB) {
    const int e0 = (wave * NB + (IDX - NA)) * EPP;
    const int e = e0 + lane * PER;
    const T* gb = B + (e % Cfg::BN) + (long) (k0 + e / Cfg::BN) * ldb;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) gb,
                                     (__attribute__((address_space(3))) void*) (buf + Cfg::A_ELEMS + e0), 16, 0, 0);
  }
  return;
#endif
  if constexpr (IDX < NA) {
    static_assert(EPP % Cfg::BM == 0 || Cfg::BM % EPP == 0, "pieces and columns nest");
    const int e0 = (wave * NA + IDX) * EPP;
    const unsigned loff = (unsigned) sizeof(T) * (unsigned) ((lane * PER) % Cfg::BM + ((lane * PER) / Cfg::BM) * (int) lda);
    const T* ga = A + (e0 % Cfg::BM) + (long) (k0 + e0 / Cfg::BM) * lda;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) at(ga, loff),
                                     (__attribute__((address_space(3))) void*) (buf + e0), 16, 0, 0);
  }
  else if constexpr (IDX < NA + NB) {
    static_assert(EPP % Cfg::BN == 0 || Cfg::BN % EPP == 0, "pieces and columns nest");
    const int e0 = (wave * NB + (IDX - NA)) * EPP;
    const unsigned loff = (unsigned) sizeof(T) * (unsigned) ((lane * PER) % Cfg::BN + ((lane * PER) / Cfg::BN) * (int) ldb);
    const T* gb = B + (e0 % Cfg::BN) + (long) (k0 + e0 / Cfg::BN) * ldb;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) at(gb, loff),
                                     (__attribute__((address_space(3))) void*) (buf + Cfg::A_ELEMS + e0), 16, 0, 0);
  }
}

// acc += A(mrows x K) * B(ncols x K)^H for one BM x BN block.  lds: 2 * BUF_ELEMS of R.
// EDGE: rows >= mrows / ncols and k >= K are zero-filled; otherwise mrows == BM, ncols == BN and
// K % BK == 0 are the caller's promise.  All threads of the workgroup must call it.
// Two-segment form: columns k >= K1 of both operands come from A2 / B2 (same leading dimensions): the
// product of two panels applied in one pass (K = K1 + K2, one epilogue), or the two products of a her2k.
// A2 / B2 are passed already shifted back by K1 columns, K1 is a multiple of BK; K1 >= K: one segment.
// UTAIL (direct-to-LDS path): the K loop issues loads in EVERY iteration -- past the end the last slab is fetched
// again into the ring slot consumed an iteration ago, never read, drained after the loop -- so the loop body has
// no branch around the loads and one wait count.  Measured on the persistent bulk launches (fp64, 480 workgroups):
// 65.7 -> 67.0 TFlop/s at K = 1024, 66.2 -> 67.4 at K = 2048; on one-block-per-workgroup launches it costs 8 %
// (the drain delays the epilogue), so only the bulk instantiation of the update kernel asks for it.
template <class Cfg, class T, bool VEC, bool EDGE, bool UTAIL = false, bool NEG = false>
__device__ __forceinline__ void gemm_nt_block(const T* __restrict__ A, long lda, int mrows,
                                              const T* __restrict__ B, long ldb, int ncols, int K,
                                              typename Cfg::R* __restrict__ lds, Acc<Cfg>& acc, int K1 = 1 << 30,
                                              const T* __restrict__ A2 = nullptr,
                                              const T* __restrict__ B2 = nullptr, int s0 = 0) {
  using R = typename Cfg::R;
  const int lane = threadIdx.x & 63;
#if DLAF_GLDS_SCALAR_ADDR
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: scalar address math
#else
  const int wave = threadIdx.x >> 6;
#endif
  const int wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
  const int nk = (K + Cfg::BK - 1) / Cfg::BK;
  if (nk == 0)
    return;
  if constexpr (Cfg::GLDS && VEC && !EDGE) {
    // multi-stage direct-to-LDS pipeline: slabs kt+1 .. kt+STAGES-1 are in flight while slab kt is
    // consumed.  Each wave issues LPS loads per slab; "vmcnt(LPS*(STAGES-2))" therefore means "my share
    // of slab kt+1 has landed", and the raw barrier extends that to every wave's share.
    constexpr int ST = Cfg::STAGES;
    constexpr int LPS = Cfg::LPS;
    static_assert(LPS * (ST - 2) < 64, "vmcnt is a 6-bit counter");
    // s0 (direct-to-LDS path): the slab the sum over k STARTS at; the loop walks s0, s0+1, .., nk-1, 0, .., s0-1.
    // The order of the sum is free, and a caller that starts every block at "the slab the others are at right
    // now" keeps the blocks that share operand strips at the same k -- see update_kernel.  0 <= s0 < nk.
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
      if (s < nk) {
        int sl = s0 + s;
        sl = sl >= nk ? sl - nk : sl;
        stage_glds<Cfg, T>(sl * Cfg::BK < K1 ? A : A2, lda, sl * Cfg::BK < K1 ? B : B2, ldb, sl * Cfg::BK,
                           lds + s * Cfg::BUF_ELEMS, wave, lane);
      }
    // slab 0 must be complete: everything but the younger (ST-2) slabs
    if (nk >= ST - 1)
      __builtin_amdgcn_s_waitcnt(0x0070 | ((LPS * (ST - 2)) & 0xF) | ((((LPS * (ST - 2)) >> 4) & 0x3) << 14));
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifndef DLAF_GLDS_INTERLEAVE
#define DLAF_GLDS_INTERLEAVE 1
#endif
#if DLAF_GLDS_INTERLEAVE
    int cur_i = 0, nxt_i = ST - 1;
#ifdef DLAF_DBG_STAMPS
    // tuning aid (tools/update_bench.hip): where a K-loop iteration spends its cycles, per wave
    unsigned long long st_issue = 0, st_mma = 0, st_vm = 0, st_bar = 0;
#endif
    // The loads of slab kt+ST-1 are issued INSIDE the MFMA stream of slab kt, one instruction per group of TM
    // MFMAs over the first groups of the slab, instead of as a block of LPS instructions at the head of the
    // iteration.  Measured (tools/run_ab_interleave.sh, fp64): persistent launches 65.7 -> 67.0 TFlop/s at
    // K = 1024, 66.2 -> 67.4 at K = 2048, 60.4 -> 61.5 at nb = 512.  The last ST-1 iterations load nothing.
    constexpr bool SPLIT = !Cfg::CXI && LPS <= (Cfg::BK / 4) * Cfg::TN && LPS <= 16;
    for (int kt = 0; kt < nk; ++kt) {
#ifdef DLAF_DBG_STAMPS
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
      R* cur = lds + cur_i * Cfg::BUF_ELEMS;
#ifdef DLAF_DBG_SKIP_GLOBAL
      cur = lds;
      const bool load = false;
#else
      const bool load = UTAIL || (kt + ST - 1 < nk);
#endif
      // (one copy of the MFMA stream: `load` only guards the single instructions, a scalar branch each)
      int sn = s0 + min(kt + ST - 1, nk - 1);
      sn = sn >= nk ? sn - nk : sn;
      const int kn = sn * Cfg::BK;
      const T* An = kn < K1 ? A : A2;
      const T* Bn = kn < K1 ? B : B2;
      R* nbuf = lds + nxt_i * Cfg::BUF_ELEMS;
      if constexpr (SPLIT) {
        mma_slab<Cfg, NEG>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane, [&](int p) {
          // (p is a compile-time constant after unrolling: the chain folds to the one instruction of group p)
#define DLAF_GLDS_AT(I)                                                                     \
  if (p == I) {                                                                           \
    if constexpr (I < LPS)                                                                \
      if (load)                                                                           \
        stage_glds_one<Cfg, T, (I < LPS ? I : 0)>(An, lda, Bn, ldb, kn, nbuf, wave, lane); \
  }
          DLAF_GLDS_AT(0) DLAF_GLDS_AT(1) DLAF_GLDS_AT(2) DLAF_GLDS_AT(3) DLAF_GLDS_AT(4) DLAF_GLDS_AT(5)
          DLAF_GLDS_AT(6) DLAF_GLDS_AT(7) DLAF_GLDS_AT(8) DLAF_GLDS_AT(9) DLAF_GLDS_AT(10) DLAF_GLDS_AT(11)
          DLAF_GLDS_AT(12) DLAF_GLDS_AT(13) DLAF_GLDS_AT(14) DLAF_GLDS_AT(15)
#undef DLAF_GLDS_AT
        });
      }
      else {
        if (load)
          stage_glds<Cfg, T>(An, lda, Bn, ldb, kn, nbuf, wave, lane);
        mma_slab<Cfg, NEG>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane);
      }
      // slab kt+1 landed?  still in flight afterwards: the loads of slabs kt+2 .. kt+ST-1 (none in the tail)
      if (load)
        __builtin_amdgcn_s_waitcnt(0x0070 | ((LPS * (ST - 2)) & 0xF) | ((((LPS * (ST - 2)) >> 4) & 0x3) << 14));
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef DLAF_DBG_STAMPS
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef DLAF_DBG_STAMPS
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
#ifndef DLAF_DBG_NO_SLAB_BARRIER  // tuning aid (timing only, the results are garbage): what the slab barrier costs
      __builtin_amdgcn_s_barrier();
#endif
#ifdef DLAF_DBG_STAMPS
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      st_mma += t2 - t0;
      st_vm += t3 - t2;
      st_bar += t4 - t3;
#endif
      cur_i = (cur_i + 1 == ST) ? 0 : cur_i + 1;
      nxt_i = (nxt_i + 1 == ST) ? 0 : nxt_i + 1;
    }
    if constexpr (UTAIL) {
      // the re-fetched slabs of the last ST-1 iterations: nobody reads them, but they must have landed (in every
      // wave) before the caller reuses the ring
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#ifdef DLAF_DBG_STAMPS
    if (lane == 0) {
      atomicAdd(&g_dbg_stamps[0], st_issue);
      atomicAdd(&g_dbg_stamps[1], st_mma);
      atomicAdd(&g_dbg_stamps[2], st_vm);
      atomicAdd(&g_dbg_stamps[3], st_bar);
      atomicAdd(&g_dbg_stamps[4], 1ull);
    }
#endif
#else
    int cur_i = 0, nxt_i = ST - 1;
    for (int kt = 0; kt < nk; ++kt) {
      R* cur = lds + cur_i * Cfg::BUF_ELEMS;
#ifdef DLAF_DBG_SKIP_GLOBAL
      cur = lds;
#else
      if (kt + ST - 1 < nk) {
        const int kn = (kt + ST - 1) * Cfg::BK;
        stage_glds<Cfg, T>(kn < K1 ? A : A2, lda, kn < K1 ? B : B2, ldb, kn, lds + nxt_i * Cfg::BUF_ELEMS, wave, lane);
      }
#endif
      mma_slab<Cfg, NEG>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane);
      // next slab (kt+1) landed?  loads still allowed in flight: those of slabs kt+2 .. kt+ST-1
      const int younger = min(ST - 2, max(0, nk - 2 - kt));
      if (younger == ST - 2)
        __builtin_amdgcn_s_waitcnt(0x0070 | ((LPS * (ST - 2)) & 0xF) | ((((LPS * (ST - 2)) >> 4) & 0x3) << 14));
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur_i = (cur_i + 1 == ST) ? 0 : cur_i + 1;
      nxt_i = (nxt_i + 1 == ST) ? 0 : nxt_i + 1;
    }
#endif
    return;
  }
  Slab<T, Cfg::BM, Cfg::BK, VEC, Cfg::LDA, Cfg::CXI, Cfg::THREADS> sa;
  Slab<T, Cfg::BN, Cfg::BK, VEC, Cfg::LDB, Cfg::CXI, Cfg::THREADS> sb;
  // (EDGE: the slab picks its source column by column, K1 need not be a multiple of BK)
  sa.template load<EDGE>(0 < K1 ? A : A2, lda, 0, mrows, K, A2, K1);
  sb.template load<EDGE>(0 < K1 ? B : B2, ldb, 0, ncols, K, B2, K1);
  sa.store(lds);
  sb.store(lds + Cfg::A_ELEMS);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    R* cur = lds + (kt & 1) * Cfg::BUF_ELEMS;
    R* nxt = lds + ((kt + 1) & 1) * Cfg::BUF_ELEMS;
#ifdef DLAF_DBG_SKIP_GLOBAL
    const bool more = false;  // tuning aid: reuse the first slab, no global traffic in the loop
    cur = lds;
#else
    const bool more = (kt + 1) < nk;
#endif
    if (more) {
      const int kn = (kt + 1) * Cfg::BK;
      sa.template load<EDGE>(kn < K1 ? A : A2, lda, kn, mrows, K, A2, K1);
      sb.template load<EDGE>(kn < K1 ? B : B2, ldb, kn, ncols, K, B2, K1);
    }
    mma_slab<Cfg, NEG>(cur, cur + Cfg::A_ELEMS, acc, wm, wn, lane);
    if (more) {
      sa.store(nxt);
      sb.store(nxt + Cfg::A_ELEMS);
    }
    __syncthreads();
  }
}

}  // namespace dlaf_mi355x
