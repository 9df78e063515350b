// trsm_rows_z.hpp -- the row-owner panel TRSM of kernels_trsm.hip for complex<double>:
//     X = B L^-H,   workgroup = 64-row strip, wave = 16 rows, macro blocks of NW = 128 columns whose running
//     sums S = sum_k X[:,k] L[c0.., k]^H (re and im accumulators, 16 x 128 per wave) stay in registers.
// Same structure as the real kernel (P1 over the columns left of the macro block with X re-read through LDS,
// then per 64-column sub-block SOLVE with inv(L_ss) and UPD of the sub-blocks to its right, both with the MFMA
// accumulators fed back as operands); what differs:
//   * every complex product is 4 real MFMAs:  (xr + i xi)(lr - i li) = (xr lr + xi li) + i (xi lr - xr li);
//   * LDS images hold (re, im) interleaved, 16 bytes per element: global -> LDS stays one global_load_lds_dwordx4
//     per 64 elements and one ds_read_b128 yields a whole fragment element; the fragment maps are the natural
//     ones (tile t, register v of lane group g <-> column 16 t + 4 v + g), no paired rows;
//   * 512 registers per lane are needed (S 128 + B 64 + X 64 + fragments): one wave per SIMD, one workgroup per
//     compute unit, three ring stages.
// Included by kernels_trsm.hip.
#pragma once
#include "device_api.hpp"
#include "mma_core.hpp"

namespace dlaf_mi355x {

template <int NW_, int ST_>
struct TrsmRowsZCfg {
  static constexpr int NW = NW_, ST = ST_;
  static constexpr int ROWS = 64, BK = 8;
  static constexpr int NT = NW / 16, NS = NW / kDiagBlock;
  // sizes in complex elements (16 bytes)
  static constexpr int A_ELEMS = BK * ROWS;   // X strip part of a stage: image [8][64]
  static constexpr int B_ELEMS = BK * NW;     // L part: image [8][NW]
  static constexpr int STAGE = A_ELEMS + B_ELEMS;
  static constexpr int W_ELEMS = kDiagBlock * kDiagBlock;
  static constexpr int LDS_BYTES = (ST * STAGE + W_ELEMS) * 16;
  static constexpr int LA = A_ELEMS / 64 / 4;  // 1 KiB pieces (64 elements) per wave: X part
  static constexpr int LB = B_ELEMS / 64 / 4;  //                                       L part
  static constexpr int LPS = LA + LB;
  static_assert(kThreads == 256 && kDiagBlock == 64 && NW % 64 == 0 && LPS * (ST - 2) < 64 && LDS_BYTES <= 160 * 1024,
                "geometry");
};

template <int NW, int ST>
__global__ __launch_bounds__(kThreads, 1) void trsm_rows_z_kernel(TrsmArgs<cdouble> p, int spt) {
  using C = TrsmRowsZCfg<NW, ST>;
  typedef double acc_t __attribute__((ext_vector_type(4)));
  typedef double d2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  cdouble* lds = reinterpret_cast<cdouble*>(lds_raw);
  cdouble* Wbuf = lds + ST * C::STAGE;

  if (*p.info != 0)
    return;
  const int il = p.il0 + blockIdx.x / spt;
  const int strip = blockIdx.x % spt;
  const int gi = il * p.pr + p.ri;
  const int rows_tile = (gi == p.nt - 1) ? p.last_rows : p.nb;
  const int m0 = strip * C::ROWS;
  if (m0 >= rows_tile)
    return;  // (the launcher promises rows_tile % 64 == 0)
  cdouble* Bst = p.b + (long) (il - p.il0) * p.b_ts + m0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int ldb = p.ldb, ldl = p.ldl;
  const int mrow = 16 * wave + c;
  // per-lane BYTE offsets (32-bit) on top of wave-uniform bases
  const unsigned bx_lane = 16u * (unsigned) (mrow + g * ldb);  // element (row mrow, column + g) of B / X
  const unsigned pc_lane = 16u * (unsigned) lane;              // pieces: 64 consecutive elements
  auto at = [](const cdouble* base, unsigned byte_off) {
    return reinterpret_cast<const cdouble*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto glds = [](const cdouble* src, cdouble* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) src,
                                     (__attribute__((address_space(3))) void*) dst, 16, 0, 0);
  };
  constexpr int PPC = NW / 64;  // pieces per column of the L image

  // with_x: the stage also carries the 8 columns of the X strip (P1 stages; the UPD stages take X from registers)
  auto issue_stage = [&](int c0, int i, int slot, bool with_x) {
    cdouble* buf = lds + slot * C::STAGE;
    const int col0 = 8 * i;
    if (with_x) {
#pragma unroll
      for (int q = 0; q < C::LA; ++q) {  // X strip: piece = one column of 64 rows
        const int piece = wave * C::LA + q;
        glds(at(Bst + (long) (col0 + piece) * ldb, pc_lane), buf + 64 * piece);
      }
    }
#pragma unroll
    for (int q = 0; q < C::LB; ++q) {
      const int piece = wave * C::LB + q;
      const cdouble* lbase = p.l + (c0 + 64 * (piece % PPC)) + (long) (col0 + piece / PPC) * ldl;
      glds(at(lbase, pc_lane), buf + C::A_ELEMS + 64 * piece);
    }
  };
  auto load_w = [&](int jblk) {  // inv(L_jj): dense 64 x 64 column-major = image [k][64], 16 pieces per wave
    const cdouble* W = p.winv + (long) jblk * C::W_ELEMS + 1024 * wave;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      glds(at(W + 64 * q, pc_lane), Wbuf + 1024 * wave + 64 * q);
  };

  acc_t Sre[C::NT], Sim[C::NT];
  acc_t Bre[4], Bim[4], Xre[4], Xim[4];
  // accumulator register v of tile t4 (lane group g) <-> column 16 t4 + 4 v + g of the sub-block
  auto load_b = [&](int col0) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const cdouble bv = *at(Bst + (long) (col0 + 16 * t4 + 4 * v) * ldb, bx_lane);
        Bre[t4][v] = bv.re;
        Bim[t4][v] = bv.im;
      }
  };
  auto wait_ring = [&]() {
    // counted by the smallest stage (LB loads per wave): a stage that must have landed never stays in flight
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((C::LB * (ST - 2)) & 0xF) | ((((C::LB * (ST - 2)) >> 4) & 0x3) << 14));
  };

  const int nmacro = p.n / NW;
  for (int J = 0; J < nmacro; ++J) {
    const int c0 = NW * J;
    const int P = c0 / C::BK;
    const int NSTG = P + 8 * (C::NS - 1);
#pragma unroll
    for (int t = 0; t < C::NT; ++t) {
      Sre[t] = acc_t{0, 0, 0, 0};
      Sim[t] = acc_t{0, 0, 0, 0};
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_w(c0 / kDiagBlock);
    load_b(c0);
#pragma unroll
    for (int st = 0; st < ST - 1; ++st)
      if (st < NSTG)
        issue_stage(c0, st, st, st < P);
    if (NSTG >= ST - 1)
      wait_ring();
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int slot = 0, nslot = ST - 1;
    auto ring_step_end = [&](int i) {
      if (i + ST - 1 < NSTG)
        wait_ring();
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      slot = (slot + 1 == ST) ? 0 : slot + 1;
      nslot = (nslot + 1 == ST) ? 0 : nslot + 1;
    };

    // ---- P1: S += X[:, 8i .. 8i+7] L[c0 .. c0+NW, 8i .. 8i+7]^H ------------------------------------
    for (int i = 0; i < P; ++i) {
      if (i + ST - 1 < NSTG)
        issue_stage(c0, i + ST - 1, nslot, i + ST - 1 < P);
      const cdouble* buf = lds + slot * C::STAGE;
#pragma unroll
      for (int k4 = 0; k4 < 2; ++k4) {
        const int kk = 4 * k4 + g;
        const d2 xf = *reinterpret_cast<const d2*>(&buf[kk * C::ROWS + mrow]);
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
          const d2 lf = *reinterpret_cast<const d2*>(&buf[C::A_ELEMS + kk * NW + 16 * t + c]);
          Sre[t] = Mma<double>::mma(lf[0], xf[0], Sre[t]);
          Sre[t] = Mma<double>::mma(lf[1], xf[1], Sre[t]);
          Sim[t] = Mma<double>::mma(lf[0], xf[1], Sim[t]);
          Sim[t] = Mma<double>::mma_neg(lf[1], xf[0], Sim[t]);
        }
      }
      ring_step_end(i);
    }

#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
      const int cs = c0 + kDiagBlock * s;
      // SOLVE: X_s = (B_s - S_s) W^H, W = inv(L_ss) lower triangular (tile ct of Y only feeds X tiles j2 >= ct)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          Bre[t4][v] -= Sre[4 * s + t4][v];
          Bim[t4][v] -= Sim[4 * s + t4][v];
        }
        Xre[t4] = acc_t{0, 0, 0, 0};
        Xim[t4] = acc_t{0, 0, 0, 0};
      }
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2)
#pragma unroll
        for (int ct = 0; ct <= j2; ++ct)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int k = 16 * ct + 4 * v + g;
            const d2 wf = *reinterpret_cast<const d2*>(&Wbuf[k * kDiagBlock + 16 * j2 + c]);
            // (yr + i yi)(wr - i wi) = (yr wr + yi wi) + i (yi wr - yr wi)
            Xre[j2] = Mma<double>::mma(wf[0], Bre[ct][v], Xre[j2]);
            Xre[j2] = Mma<double>::mma(wf[1], Bim[ct][v], Xre[j2]);
            Xim[j2] = Mma<double>::mma(wf[0], Bim[ct][v], Xim[j2]);
            Xim[j2] = Mma<double>::mma_neg(wf[1], Bre[ct][v], Xim[j2]);
          }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          cdouble* dst = const_cast<cdouble*>(at(Bst + (long) (cs + 16 * t4 + 4 * v) * ldb, bx_lane));
          *dst = cdouble{Xre[t4][v], Xim[t4][v]};
        }
      if (s == C::NS - 1)
        break;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      load_w(cs / kDiagBlock + 1);
      load_b(cs + kDiagBlock);
      // UPD: S_t += X_s L[t, s]^H for the tiles right of sub-block s
      const int tmin = 4 * (s + 1);
#pragma unroll
      for (int uu = 0; uu < 8; ++uu) {
        const int i = P + 8 * s + uu;
        if (i + ST - 1 < NSTG)
          issue_stage(c0, i + ST - 1, nslot, false);
        const cdouble* buf = lds + slot * C::STAGE + C::A_ELEMS;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          // stage columns 8 uu + 4 b + g of the sub-block <-> register 2 (uu & 1) + b of tile uu >> 1
          const double xr = Xre[uu >> 1][2 * (uu & 1) + b], xi = Xim[uu >> 1][2 * (uu & 1) + b];
          const int kk = 4 * b + g;
#pragma unroll
          for (int t = tmin; t < C::NT; ++t) {
            const d2 lf = *reinterpret_cast<const d2*>(&buf[kk * NW + 16 * t + c]);
            Sre[t] = Mma<double>::mma(lf[0], xr, Sre[t]);
            Sre[t] = Mma<double>::mma(lf[1], xi, Sre[t]);
            Sim[t] = Mma<double>::mma(lf[0], xi, Sim[t]);
            Sim[t] = Mma<double>::mma_neg(lf[1], xr, Sim[t]);
          }
        }
        ring_step_end(i);
      }
    }
  }
}

}  // namespace dlaf_mi355x
