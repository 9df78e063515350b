// kernels_update.hip -- trailing-matrix update of one Cholesky step as ONE grouped launch.
//
// Reference: cholesky/impl.h:70-94 (herkTrailingDiagTile / gemmTrailingMatrixTile), issued one
// tile at a time by the loops at impl.h:168-187 (local) and :273-300 (distributed) through
// rocBLAS (blas/tile.h:370-382, :414-425).  Here the whole set of local trailing tiles is one
// kernel: each workgroup owns a BM x BN block of one tile, accumulates A(il)*B(jl)^H over K with
// fp64/fp32 MFMA register tiles fed from LDS-staged panel slabs, and applies C -= acc once.
//   * blockIdx -> block mapping is XCD-aware: the 8 XCDs get contiguous runs of 8x8-block
//     patches so that the panel slabs a patch shares stay in that XCD's L2;
//   * herk is the same code with a lower-triangle mask on diagonal tiles (and a real diagonal
//     for complex types).
// Roofline: MFMA-bound, 2*nb^3 flop per 4*nb^2*sizeof(T) algorithmic bytes per tile.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "device_api.hpp"
#include "mma_core.hpp"

namespace dlaf_mi355x {

template <class T>
struct UpdateCfg;
template <>
struct UpdateCfg<float> {
  using type = BlockCfg<float, 128, 128, 64, 64, 16>;
  static constexpr int min_waves = 2;
};
template <>
struct UpdateCfg<double> {
  // 2 stages x BK 16 and 4 stages x BK 8 measure the same (67.5 TFlop/s standalone): the in-loop
  // global traffic costs clock (DVFS), not latency -- see DESIGN.md
#if defined(DLAF_UPD_BIG)
  // 256 x 128 block, 8 waves (4 x 2 wave tiles of 64 x 64), one workgroup per compute unit: 25 % fewer
  // L2 -> LDS bytes per flop than two independent 128 x 128 blocks
  using type = BlockCfg<double, 256, 128, 64, 64, 16, true, DLAF_UPD_BIG, 512>;
  static constexpr int min_waves = 2;
#elif defined(DLAF_UPD_WIDE4)
  // tuning aid (round 3): 256 x 128 block on FOUR waves (2 x 2 wave tiles of 128 x 64 = 8 x 4 MFMA tiles, 256
  // accumulator registers: one wave per SIMD, one workgroup per compute unit) -- the "fewer LDS bytes per flop"
  // variant of DESIGN section 8.2: 0.375 fragment reads per MFMA instead of 0.5, 25 % fewer L2 -> LDS bytes
  using type = BlockCfg<double, 256, 128, 128, 64, 16, true, DLAF_UPD_WIDE4, 256>;
  static constexpr int min_waves = 1;
#elif defined(DLAF_UPD_BK) && defined(DLAF_UPD_ST)
  // tuning aid (tools/run_ab_ring.sh): slab depth / ring depth of the direct-to-LDS pipeline
  using type = BlockCfg<double, 128, 128, 64, 64, DLAF_UPD_BK, true, DLAF_UPD_ST>;
  static constexpr int min_waves = 2;
#else
  using type = BlockCfg<double, 128, 128, 64, 64, 16, true, 2>;
  static constexpr int min_waves = 2;
#endif
};
template <>
struct UpdateCfg<cfloat> {
  using type = BlockCfg<cfloat, 128, 128, 64, 64, 16>;
  static constexpr int min_waves = 2;
};
template <>
struct UpdateCfg<cdouble> {
  using type = BlockCfg<cdouble, 128, 64, 64, 32, 8, true, 3>;  // interleaved LDS image, direct-to-LDS, 3 stages
  static constexpr int min_waves = 2;
};

constexpr int kMaxPatchCols = 256;

struct UpdateMap {
  int ps;        // patch = (1<<ps) block rows x (1<<psc) block columns, square in ELEMENTS
  int psc;
  int PR;        // patch rows
  int tri;       // != 0: only the patches at/below the block-cyclic diagonal are enumerated, patch column by
                 // patch column, through colstart[] (prefix sums of the valid patches per patch column)
  int PC;        // patch columns
  int colstart[kMaxPatchCols + 1];
  int xcd;       // remap blockIdx so each XCD works on consecutive patches
  int RB, CB;    // block rows / cols of the domain
  int bpt_m, bpt_n;
  long total;    // work items (blocks of the patch enumeration)
  int persist;   // != 0: the grid is smaller than `total`; workgroups pull work items from `counters`
  unsigned* counters;  // persist: 8 (per workgroup-id-mod-8, i.e. per XCD) or 1 dequeue heads, zeroed per launch;
                       // counters[8 + q]: work items of queue q that are finished (lockstep pacing)
  int lockstep;        // persist: the workgroups of a queue start their items in rounds (see update_kernel)
  unsigned kphase_ticks;  // persist: wall-clock ticks per K slab of a block (0: every block starts at slab 0)
  int excl_rank;       // persist: workgroups that find themselves on one of the first `excl_rank` compute units of their
                       // XCD (g_cu_rank) leave at once -- whole compute units stay free for the kernels beside the update
  unsigned excl_budget;  // ... but no more than this many per launch (counters[15] counts them)
  int steal;           // persist: a workgroup whose queue is empty drains the other queues
  int* cu_busy;        // persist: per (XCD, compute unit) count of tile-POTRF strips resident there (null: off) -- a bulk
                       // workgroup that shares its compute unit with a strip pauses between two work items
};

// rank of a compute unit among the compute units of its XCD that the probe launch of update_kernels_init() saw
// (255: not seen).  "The first r compute units of every XCD" is the set a persistent bulk launch vacates.
__device__ unsigned char g_cu_rank[8][256];

__global__ void cu_probe_kernel(unsigned* seen) {
  extern __shared__ unsigned char probe_lds[];
  if (threadIdx.x == 0) {
    unsigned xcc;
    const unsigned key = phys_cu_key(xcc);
    atomicOr(&seen[xcc * 8 + (key >> 5)], 1u << (key & 31));
    probe_lds[0] = (unsigned char) key;
  }
  // stay a little so that the launch spreads over every compute unit
  for (int i = 0; i < 64; ++i)
    __builtin_amdgcn_s_sleep(64);
}

// One work item = one BM x BN block of one tile.  Returns early for blocks outside the domain.
// PART: 0 every block; 1 only the interior blocks (whole BM x BN, whole slabs, no triangle mask: the K loop on the
// direct-to-LDS path + the wide epilogue, nothing else compiled in); 2 only the others (edge / masked blocks)
template <class T, bool VEC, bool UTAIL = false, int PART = 0>
__device__ __forceinline__ void update_block(const UpdateArgs<T>& p, const UpdateMap& mp, long w,
                                             real_t<T>* __restrict__ lds, int s0 = 0) {
  using Cfg = typename UpdateCfg<T>::type;
  using R = real_t<T>;

  // ---- which block is work item w ----------------------------------------------------------
  const int ps = mp.ps, psc = mp.psc;
  const long patch = w >> (ps + psc);
  const int q = (int) (w & ((1 << (ps + psc)) - 1));
  int pi, pj;
  if (mp.tri) {
    // binary search of the patch column: colstart[pj] <= patch < colstart[pj + 1]
    int lo = 0, hi = mp.PC;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (mp.colstart[mid] <= patch)
        lo = mid;
      else
        hi = mid;
    }
    pj = lo;
    const int cnt = mp.colstart[pj + 1] - mp.colstart[pj];
    pi = mp.PR - cnt + (int) (patch - mp.colstart[pj]);
  }
  else {
    pi = (int) (patch % mp.PR);
    pj = (int) (patch / mp.PR);
  }
  const int br = (pi << ps) + (q & ((1 << ps) - 1));
  const int bc = (pj << psc) + (q >> ps);
  if (br >= mp.RB || bc >= mp.CB)
    return;
  const int il = p.il0 + br / mp.bpt_m, sbr = br % mp.bpt_m;
  const int jl = p.jl0 + bc / mp.bpt_n, sbc = bc % mp.bpt_n;
  const int gi = il * p.pr + p.ri, gj = jl * p.pc + p.ci;
  if (!p.rect && gi < gj)
    return;
  const int rows_tile = (gi == p.nt - 1) ? p.last_rows : p.nb;
  const int cols_tile = p.rect ? ((gj == p.nt_c - 1) ? p.last_cols : p.nb) : ((gj == p.nt - 1) ? p.last_rows : p.nb);
  const int m0 = sbr * Cfg::BM, n0 = sbc * Cfg::BN;
  if (m0 >= rows_tile || n0 >= cols_tile)
    return;
  const bool diag = !p.rect && (gi == gj);
  const int mrows = min(Cfg::BM, rows_tile - m0), ncols = min(Cfg::BN, cols_tile - n0);
  if (diag && m0 + mrows - 1 < n0)
    return;  // block strictly above the diagonal of a diagonal tile

  const T* A = p.a + (long) (il - p.il0) * p.a_ts + m0;
  // herk on a diagonal tile reads the COLUMN panel for both operands (impl.h:282-287): the
  // transposed panel never holds the tile of the last global row (broadcast_panel.h:186-191)
  // her2k (gen_to_std): C_jj -= X_j L_j^H + L_j X_j^H with a = [X | L]: the column operand of a diagonal
  // tile is the column panel again, segments swapped ([L_j | X_j])
  const int jt = jl - (p.b_jl0 >= 0 ? p.b_jl0 : p.jl0);
  const long boff = (long) (jt % p.b_period) * p.b_ts2 + (long) (jt / p.b_period) * p.b_ts;
  const long aoff = (long) (il - p.il0) * p.a_ts;
  const T* B;
  const T* B2 = nullptr;
  const long ldb = diag ? p.lda : p.ldb;
  if (!diag) {
    B = p.b + boff + n0;
    if (p.K1 > 0)
      B2 = p.b2 + boff + n0;
  }
  else if (!p.her2k) {
    B = p.a + aoff + n0;
    if (p.K1 > 0)
      B2 = p.a2 + aoff + n0;
  }
  else {
    B = p.a2 + aoff + n0;
    B2 = p.a + aoff + n0;
  }
  int K1 = 1 << 30;
  const T* A2 = nullptr;
  if (p.K1 > 0) {
    K1 = p.K1;
    A2 = p.a2 + aoff + m0 - (long) K1 * p.lda;
    B2 -= (long) K1 * ldb;
  }
#ifdef DLAF_DBG_STRIP_PACKED
  // tuning aid (tools/update_bench.hip): operands read as if every BM-row strip of a tile were stored
  // contiguously (k-major, leading dimension BM): one sequential 1 MiB stream per strip instead of 1 KiB pieces
  // 8 KiB apart.  Timing only -- the values are whatever lies there.
  const long lda_x = Cfg::BM, ldb_x = Cfg::BN;
  A = p.a + aoff + (long) (m0 / Cfg::BM) * Cfg::BM * p.nb;
  B = (diag ? p.a + aoff : p.b + boff) + (long) (n0 / Cfg::BN) * Cfg::BN * p.nb;
#define DLAF_LDA_X lda_x
#define DLAF_LDB_X ldb_x
#else
#define DLAF_LDA_X p.lda
#define DLAF_LDB_X ldb
#endif
#ifdef DLAF_DBG_SAME_STRIPS
  A = p.a;  // tuning aid (tools/update_bench.hip): every block streams the same two strips = perfect L2 locality
  B = p.b;
#endif
  T* C = p.c + (long) il * p.c_tsr + (long) jl * p.c_tsc + m0 + (long) n0 * p.ldc;

  // The operand bases are the same for every lane of the workgroup; say so (they come out of the work-item
  // decoding above in vector registers): the in-loop loads then take (scalar base) + (32-bit lane offset).
  auto uniform = [](const T* q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) v);
    const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (v >> 32));
    return reinterpret_cast<const T*>(((unsigned long long) hi << 32) | lo);
  };
#if DLAF_GLDS_SCALAR_ADDR
  A = uniform(A);
  B = uniform(B);
  A2 = uniform(A2);
  B2 = uniform(B2);
#endif
  const bool full = (mrows == Cfg::BM) && (ncols == Cfg::BN) && (p.K % Cfg::BK == 0) && (p.K1 % Cfg::BK == 0);
  if constexpr (PART != 0) {
    const bool interior = full && !(diag && m0 < n0 + ncols - 1) && VEC && ((p.ldc * (long) sizeof(T)) % 16 == 0) &&
                          (reinterpret_cast<uintptr_t>(C) % 16 == 0);
    if ((PART == 1) != interior)
      return;
  }
  Acc<Cfg> acc;
#ifndef DLAF_UPD_PRELOAD
#define DLAF_UPD_PRELOAD 1
#endif
  if constexpr (DLAF_UPD_PRELOAD && Cfg::PAIRED && !Cfg::CX && sizeof(R) == 8 && VEC && !UTAIL) {
    // Interior blocks (fp64 fast path): C is loaded INTO the accumulators before the K loop -- the loads travel
    // with the first slabs, whose arrival the loop waits for anyway -- and the MFMAs subtract (neg:[1,0,0]), so
    // the block ends with plain stores instead of 16 load -> wait -> subtract -> store round trips.  A/B on one
    // MI355X (tools/run_ab_preload.sh): one-block-per-workgroup launches 66.3 -> 67.8 TFlop/s (N = 49152, K =
    // 1024); the persistent bulk launches LOSE 3 % (66.9 -> 65.0; nb = 512: 61.2 -> 57.4 -- the stores of block i
    // and the C loads of block i+1 queue ahead of its first slabs), so the bulk instantiation keeps the old form.
    const bool unmasked = full && !(diag && m0 < n0 + ncols - 1);
    if (unmasked && ((p.ldc * (long) sizeof(T)) % 16 == 0) && (reinterpret_cast<uintptr_t>(C) % 16 == 0)) {
      typedef R r2 __attribute__((ext_vector_type(2)));
      // address = uniform base (scalar registers: block, wave tile, accumulator column) + ONE 32-bit lane offset:
      // 32 pointers kept live across the K loop would cost 64 registers
      const int lane_ = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      const int wm_ = wave_ % Cfg::WAVES_M, wn_ = wave_ / Cfg::WAVES_M;
      const int g_ = lane_ >> 4, c_ = lane_ & 15;
      const unsigned lane_off = (unsigned) sizeof(T) * (unsigned) (2 * c_ + 2 * g_ * (int) p.ldc);
      T* Cw = C + wm_ * Cfg::WM + (long) (wn_ * Cfg::WN) * p.ldc;
      auto at = [](T* base, unsigned byte_off) {
        return reinterpret_cast<r2*>(reinterpret_cast<char*>(base) + byte_off);
      };
      static_assert(Cfg::TN % 2 == 0, "paired accumulator columns");
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          T* col = Cw + (long) ((j >> 1) * 32 + 8 * v + (j & 1)) * p.ldc;  // acc_n(j, g, v) minus the lane part 2 g
#pragma unroll
          for (int q = 0; q < Cfg::TM / 2; ++q) {
            const r2 cv = *at(col + q * 32, lane_off);
            acc.re[2 * q][j][v] = cv[0];
            acc.re[2 * q + 1][j][v] = cv[1];
          }
        }
      gemm_nt_block<Cfg, T, VEC, false, UTAIL, true>(A, DLAF_LDA_X, mrows, B, DLAF_LDB_X, ncols, p.K, lds, acc, K1, A2,
                                                     B2);
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          T* col = Cw + (long) ((j >> 1) * 32 + 8 * v + (j & 1)) * p.ldc;
#pragma unroll
          for (int q = 0; q < Cfg::TM / 2; ++q)
            *at(col + q * 32, lane_off) = r2{acc.re[2 * q][j][v], acc.re[2 * q + 1][j][v]};
        }
      return;
    }
  }
  acc.clear();
  if constexpr (PART == 1) {
    gemm_nt_block<Cfg, T, VEC, false, UTAIL>(A, DLAF_LDA_X, mrows, B, DLAF_LDB_X, ncols, p.K, lds, acc, K1, A2, B2,
                                             s0);
  }
  else {
    if (full)
      gemm_nt_block<Cfg, T, VEC, false, UTAIL>(A, DLAF_LDA_X, mrows, B, DLAF_LDB_X, ncols, p.K, lds, acc, K1, A2, B2,
                                               s0);
    else
      gemm_nt_block<Cfg, T, false, true>(A, p.lda, mrows, B, ldb, ncols, p.K, lds, acc, K1, A2, B2);
  }

#ifdef DLAF_DBG_SKIP_EPILOGUE
  {
    R sum = 0;  // keep every accumulator live; the comparison is never true on real data
    for (int i = 0; i < Cfg::TM; ++i)
      for (int j = 0; j < Cfg::TN; ++j)
        for (int v = 0; v < 4; ++v)
          sum += acc.re[i][j][v];
    if (sum == R(12345.678))
      C[0] = make_el<T>(sum, 0);
    return;
  }
#endif
  // ---- epilogue: C -= acc -------------------------------------------------------------------
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
  const int g = lane >> 4, c = lane & 15;
  const bool masked = PART == 1 ? false : (!full || (diag && m0 < n0 + ncols - 1));
  if constexpr (Cfg::PAIRED) {
    // lane holds rows (m, m+1) of tiles (2q, 2q+1): 16-byte accesses when the tile column is aligned
    typedef R r2 __attribute__((ext_vector_type(2)));
    const bool wide = PART == 1 ? true
                                : (!masked && VEC && ((p.ldc * (long) sizeof(T)) % 16 == 0) &&
                                   (reinterpret_cast<uintptr_t>(C) % 16 == 0));
#ifndef DLAF_EPI_COLS
#define DLAF_EPI_COLS 4  // A/B on one MI355X (tools/run_ab_epi.sh): 1 -> 4 columns per round trip +0.7 ... 1.2 % on the bulk launches
#endif
    if (wide) {
      // DLAF_EPI_COLS accumulator columns (a column = one (j, v) pair, TM/2 16-byte accesses per lane) are
      // loaded together before any is subtracted and stored
      constexpr int NB = DLAF_EPI_COLS;
#pragma unroll
      for (int i0 = 0; i0 < Cfg::TN * 4; i0 += NB) {
        r2 cv[NB][Cfg::TM / 2];
#pragma unroll
        for (int ii = 0; ii < NB; ++ii) {
          const int j = (i0 + ii) / 4, v = (i0 + ii) % 4;
          const T* col = C + (long) (wn * Cfg::WN + acc_n<Cfg>(j, g, v)) * p.ldc;
#pragma unroll
          for (int q = 0; q < Cfg::TM / 2; ++q)
            cv[ii][q] = *reinterpret_cast<const r2*>(col + wm * Cfg::WM + q * 32 + 2 * c);
        }
#pragma unroll
        for (int ii = 0; ii < NB; ++ii) {
          const int j = (i0 + ii) / 4, v = (i0 + ii) % 4;
          T* col = C + (long) (wn * Cfg::WN + acc_n<Cfg>(j, g, v)) * p.ldc;
#pragma unroll
          for (int q = 0; q < Cfg::TM / 2; ++q) {
            cv[ii][q][0] -= acc.re[2 * q][j][v];
            cv[ii][q][1] -= acc.re[2 * q + 1][j][v];
            *reinterpret_cast<r2*>(col + wm * Cfg::WM + q * 32 + 2 * c) = cv[ii][q];
          }
        }
      }
    }
    else {
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int nl = wn * Cfg::WN + acc_n<Cfg>(j, g, v);
          T* col = C + (long) nl * p.ldc;
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            const int ml = wm * Cfg::WM + acc_m<Cfg>(i, c);
            if (!masked || (ml < mrows && nl < ncols && (!diag || (m0 + ml) >= (n0 + nl))))
              col[ml] = col[ml] - acc.re[i][j][v];
          }
        }
      }
    }
  }
  else if (!masked) {
    // interior block: unconditional, so the loads of a whole accumulator column batch up
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      T cv[4][Cfg::TM];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const T* col = C + (long) (wn * Cfg::WN + acc_n<Cfg>(j, g, v)) * p.ldc;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
          cv[v][i] = col[wm * Cfg::WM + acc_m<Cfg>(i, c)];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        T* col = C + (long) (wn * Cfg::WN + acc_n<Cfg>(j, g, v)) * p.ldc;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          if constexpr (Cfg::CX)
            col[wm * Cfg::WM + acc_m<Cfg>(i, c)] = T{cv[v][i].re - acc.re[i][j][v], cv[v][i].im - acc.im[i][j][v]};
          else
            col[wm * Cfg::WM + acc_m<Cfg>(i, c)] = cv[v][i] - acc.re[i][j][v];
        }
      }
    }
  }
  else {
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int nl = wn * Cfg::WN + acc_n<Cfg>(j, g, v);
        T* col = C + (long) nl * p.ldc;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          const int ml = wm * Cfg::WM + acc_m<Cfg>(i, c);
          if (ml < mrows && nl < ncols && (!diag || (m0 + ml) >= (n0 + nl))) {
            const T cv = col[ml];
            if constexpr (Cfg::CX) {
              T r{cv.re - acc.re[i][j][v], cv.im - acc.im[i][j][v]};
              if (diag && (m0 + ml) == (n0 + nl))
                r.im = R(0);
              col[ml] = r;
            }
            else {
              col[ml] = cv - acc.re[i][j][v];
            }
          }
        }
      }
    }
  }
}

// ROLE only names the instantiation (0 trailing bulk, 1 lookahead column, 2 in-tile POTRF / single-tile
// entries, 3 callers outside the factorization: residual checker, triangular solver) so
// that rocprof statistics and the library's own HIP-event timing refer to the same set of launches.
// Work item v -> block w: the 8 XCDs (workgroup id mod 8 under round-robin dispatch) get contiguous runs
// of 8x8-block patches.  Persistent form: gridDim.x workgroups stride over the work items, which (a)
// leaves the compute units the launcher did not ask for free for the resident POTRF / RCCL kernels
// that must run beside the bulk update and (b) keeps each XCD on neighbouring patches.
template <class T, bool VEC, int ROLE>
__global__ __launch_bounds__(UpdateCfg<T>::type::THREADS, UpdateCfg<T>::min_waves) void update_kernel(UpdateArgs<T> p,
                                                                                                     UpdateMap mp) {
  using R = real_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);
  if (*p.info != 0)
    return;
  if (!mp.persist) {
    const long v = blockIdx.x;
    const long w = mp.xcd ? (v & 7) * (mp.total >> 3) + (v >> 3) : v;
#ifdef DLAF_UPD_LEAN
    update_block<T, VEC, false, (ROLE == 0 ? 1 : 0)>(p, mp, w, lds);
#else
    update_block<T, VEC>(p, mp, w, lds);
#endif
    return;
  }
  // persistent: workgroups with the same id mod 8 (same XCD under round-robin dispatch) drain the
  // same contiguous range of work items in order through one dequeue head per range
  __shared__ unsigned next_item;
  const int nq = mp.xcd ? 8 : 1;
  const long per_q = mp.total / nq;
  const int q0 = mp.xcd ? (int) (blockIdx.x & 7) : 0;
  // Lockstep pacing: the W workgroups of a queue (one XCD) take W consecutive work items -- neighbouring blocks
  // of one 8 x 8 patch, which stream the same 16 operand strips -- and start the next W only when those are done.
  // Blocks that start together stay within a few K slabs of each other, so a slab fetched by one of the 8 blocks
  // that share a strip is still in the XCD's 4 MiB L2 when the others ask for it.  Left to themselves the
  // workgroups drift apart by more than the ~4 slabs the L2 holds at a 50 % miss rate, and the miss rate stays at
  // 50 % (measured: TCC hit 48 %, 95 GB fetched per launch for 158 GB requested at N = 49152).  Pure pacing: no
  // data depends on it, the spin is bounded.
  const unsigned W = gridDim.x / nq;
  // Exclusive compute units for the kernels that run beside this launch (tile POTRF strips, panel TRSM): the grid
  // covers every workgroup slot of the GPU and the workgroups that land on a reserved compute unit leave, so the
  // side kernels get whole compute units instead of sharing LDS / L1 / issue slots with a bulk workgroup.  The
  // budget bounds the number that leave whatever the dispatcher does (e.g. slots of reserved compute units being
  // handed out again and again while everything else is taken by another kernel).
  if (mp.excl_rank > 0) {
    __shared__ int leave;
    if (threadIdx.x == 0) {
      unsigned xcc;
      const unsigned key = phys_cu_key(xcc);
      leave = 0;
      if ((int) g_cu_rank[xcc][key] < mp.excl_rank)
        leave = atomicAdd(&mp.counters[15], 1u) < mp.excl_budget ? 1 : 0;
    }
    __syncthreads();
    if (leave)
      return;
  }
  // K-phase alignment: the sum over k of a block may start anywhere, so every block starts at the slab "the wall
  // clock is at" -- (ticks / ticks-per-slab) mod slabs -- and wraps around.  Blocks of an XCD that stream the same
  // operand strips are then at (nearly) the same k whenever they started, instead of wherever their start time
  // left them: a slab one of them fetched is still in the 4 MiB L2 when the others ask for it.  No waiting, no
  // communication; the price is a summation order that depends on the start time (results reproducible to rounding,
  // not bitwise).  Measured on MI355X (tools/run_ab_kphase.sh, profiles/r02_update_kernel_kphase_*): it does what it
  // was built for -- L2 hit rate 43 % -> 61 %, fabric fetch 542 -> 370 GB per three launches -- and the kernel is
  // exactly as fast as before (67.9 vs 67.9 TFlop/s alone, 67.3 vs 67.3 in the factorization): a slab is 256 lines,
  // the barrier waits for the slowest of them, and at 61 % hits every slab still has misses.  So: opt-in
  // (DLAF_MI355X_KPHASE=1), default = the fixed summation order.
  __shared__ int next_s0;
  const int nslab = p.K / UpdateCfg<T>::type::BK;
  int q = q0, tried = 1;
  // The tile POTRF is a latency chain of a few waves; on a compute unit it shares with a bulk workgroup it runs 2.5 x
  // slower than alone (issue slots, LDS, the texture path), on one of its own at its stand-alone speed.  So a bulk
  // workgroup whose compute unit holds a POTRF strip (the strips count themselves in cu_busy) sits out between two
  // work items until the strip has left: 16 of 480 workgroups for a millisecond per diagonal tile.  Bounded.
  int* my_busy = nullptr;
  if (mp.cu_busy != nullptr && threadIdx.x == 0) {
    unsigned xcc;
    const unsigned key = phys_cu_key(xcc);
    my_busy = mp.cu_busy + (xcc * 256u + key);
  }
  for (;;) {
    if (threadIdx.x == 0) {
      if (my_busy != nullptr) {
        int spins = 0;
        while (__hip_atomic_load(my_busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 && ++spins < 4000)
          __builtin_amdgcn_s_sleep(64);
      }
      next_item = atomicAdd(&mp.counters[q], 1u);
      next_s0 = (mp.kphase_ticks != 0 && nslab > 1) ? (int) ((wall_clock64() / mp.kphase_ticks) % (unsigned) nslab) : 0;
    }
    __syncthreads();
    // (wave-uniform: everything derived from the work item -- operand and C addresses, segment selection, the
    // LDS ring -- then lives in scalar registers; left as a per-lane LDS value, the address of every in-loop load
    // costs a 64-bit multiply-add and a select per lane in the middle of the MFMA stream)
#if DLAF_GLDS_SCALAR_ADDR
    const long i = (long) (unsigned) __builtin_amdgcn_readfirstlane((int) next_item);
    const int s0 = __builtin_amdgcn_readfirstlane(next_s0);
#else
    const long i = next_item;
    const int s0 = next_s0;
#endif
    __syncthreads();
    if (i >= per_q) {
      // own queue drained: help with the others (keeps the end of a launch balanced when the queues lost
      // different numbers of workgroups to the reserved compute units)
      if (!mp.steal || tried >= nq)
        break;
      ++tried;
      q = (q + 1) % nq;
      continue;
    }
    if (mp.lockstep) {
      if (threadIdx.x == 0) {
        const unsigned need = (unsigned) (i / W) * W;  // every item of the earlier rounds
        long spins = 0;
        while (__hip_atomic_load(&mp.counters[8 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++spins < 2000000)
          __builtin_amdgcn_s_sleep(8);
      }
      __syncthreads();
    }
#ifdef DLAF_UPD_LEAN
    update_block<T, VEC, ROLE == 0, (ROLE == 0 ? 1 : 0)>(p, mp, (long) q * per_q + i, lds, s0);
#else
    update_block<T, VEC, ROLE == 0>(p, mp, (long) q * per_q + i, lds, s0);
#endif
    if (mp.lockstep && threadIdx.x == 0)
      __hip_atomic_fetch_add(&mp.counters[8 + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// DLAF_MI355X_LOCKSTEP=1: lockstep pacing of the persistent launches (experiment, off)
static bool update_lockstep() {
  static const bool on = [] {
    const char* e = std::getenv("DLAF_MI355X_LOCKSTEP");
    return e ? std::atoi(e) != 0 : false;  // measured: 56.4 vs 66.6 TFlop/s -- the rounds wait for their slowest block
  }();
  return on;
}

// Wall-clock ticks (hipDeviceAttributeWallClockRate, the clock wall_clock64() reads) one K slab of one block
// takes when `resident` workgroups share the GPU at the bulk kernel's in-situ rate; 0 = K-phase alignment off
// (the default; DLAF_MI355X_KPHASE=1 turns it on).  DLAF_MI355X_KPHASE_RATE overrides the assumed rate (TFlop/s).  An error of 5 % in the
// rate misplaces two blocks by at most 5 % of the offset between their start times -- a few slabs.
template <class T>
static unsigned kphase_ticks(int resident) {
  using Cfg = typename UpdateCfg<T>::type;
  static const double rate = [] {
    const char* on = std::getenv("DLAF_MI355X_KPHASE");
    if (on == nullptr || std::atoi(on) == 0)
      return 0.0;  // default off: see update_kernel
    if (const char* e = std::getenv("DLAF_MI355X_KPHASE_RATE"))
      return std::atof(e) * 1e12;
    return (sizeof(real_t<T>) == 8 ? 66.5e12 : 120e12);
  }();
  static const double khz = [] {
    int dev = 0, v = 0;
    (void) hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, dev) != hipSuccess || v <= 0)
      v = 100000;
    return (double) v;
  }();
  if (rate <= 0 || resident <= 0)
    return 0;
  const double flops = (TypeInfo<T>::is_complex ? 8.0 : 2.0) * Cfg::BM * Cfg::BN * Cfg::BK;
  const double seconds = flops / (rate / resident);
  const double ticks = seconds * khz * 1e3;
  return ticks < 1 ? 1u : (unsigned) (ticks + 0.5);
}

template <class T>
static bool aligned16(const void* ptr, long stride_elems) {
  return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && ((stride_elems * (long) sizeof(T)) % 16 == 0);
}

// table of resident tile-POTRF strips per (XCD, compute unit): 8 x 256 counters, shared with kernels_potrf_coop.hip
// through cu_busy_table() (null: the pause is off, DLAF_MI355X_POTRF_YIELD=0)
static int* g_cu_busy = nullptr;
int* cu_busy_table() {
  static const bool on = [] {
    const char* e = std::getenv("DLAF_MI355X_POTRF_YIELD");
    return e ? std::atoi(e) != 0 : true;
  }();
  if (on && g_cu_busy == nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&g_cu_busy), 8 * 256 * sizeof(int)) != hipSuccess ||
        zero_device_now(g_cu_busy, 8 * 256 * sizeof(int)) != hipSuccess) {
      (void) hipGetLastError();
      g_cu_busy = nullptr;
    }
  }
  return g_cu_busy;
}

// what the probe launch of update_kernels_init() found: XCDs and compute units per XCD (0: no probe, no exclusive mode)
static int g_probe_xcds = 0, g_probe_cus_per_xcd = 0, g_probe_engines = 1;

// launches of this process in persistent form / of those with exclusive compute units (tests assert that a forced
// reservation took the path it was meant to take)
static long g_launch_stats[2] = {0, 0};
void update_launch_stats(long* persistent, long* exclusive) {
  *persistent = g_launch_stats[0];
  *exclusive = g_launch_stats[1];
}

// DLAF_MI355X_STEAL=0: a persistent workgroup stops when its own queue is empty (A/B)
static bool update_steal() {
  static const bool on = [] {
    const char* e = std::getenv("DLAF_MI355X_STEAL");
    return e ? std::atoi(e) != 0 : true;
  }();
  return on;
}

template <class T>
void launch_update(const UpdateArgs<T>& a, hipStream_t stream, int role, long max_blocks, unsigned* counters,
                   bool counters_are_zero, long excl_slots) {
  using Cfg = typename UpdateCfg<T>::type;
  if (a.il1 <= a.il0 || a.jl1 <= a.jl0 || a.K <= 0 || a.nb <= 0)
    return;
  UpdateMap mp;
  mp.bpt_m = (a.nb + Cfg::BM - 1) / Cfg::BM;
  mp.bpt_n = (a.nb + Cfg::BN - 1) / Cfg::BN;
  mp.RB = (a.il1 - a.il0) * mp.bpt_m;
  mp.CB = (a.jl1 - a.jl0) * mp.bpt_n;
  // patches are square in elements (8 block rows x 8*BM/BN block columns), so the triangular patch
  // enumeration also serves rectangular blocks: without it half of a launch is empty workgroups whose
  // long runs starve the compute units (measured on the complex kernel: SQ busy 54 %)
  static_assert(Cfg::BM % Cfg::BN == 0 && ((Cfg::BM / Cfg::BN) & (Cfg::BM / Cfg::BN - 1)) == 0, "BM = 2^s * BN");
  constexpr int kAspectShift = (Cfg::BM / Cfg::BN == 1) ? 0 : (Cfg::BM / Cfg::BN == 2) ? 1 : 2;
  const bool big = (mp.RB >= 16 && mp.CB >= 16);
  mp.ps = big ? 3 : 0;
  mp.psc = big ? 3 + kAspectShift : 0;
  mp.PR = (mp.RB + (1 << mp.ps) - 1) >> mp.ps;
  mp.PC = (mp.CB + (1 << mp.psc) - 1) >> mp.psc;
  long npatch = (long) mp.PR * mp.PC;
  mp.tri = 0;
  if (mp.PC <= kMaxPatchCols) {
    // first patch row of every patch column that can hold a tile with global row >= global column
    mp.tri = 1;
    mp.colstart[0] = 0;
    for (int pj = 0; pj < mp.PC; ++pj) {
      const int jl_min = a.jl0 + (pj << mp.psc) / mp.bpt_n;
      const long gj_min = (long) jl_min * a.pc + a.ci;
      long il_first = (gj_min - a.ri + a.pr - 1) / a.pr;  // ceil((gj - ri) / pr) for gj >= ri
      if (gj_min <= a.ri || a.rect)
        il_first = 0;
      if (il_first < a.il0)
        il_first = a.il0;
      int cnt = 0;
      if (il_first < a.il1) {
        const int pi0 = (int) (((il_first - a.il0) * mp.bpt_m) >> mp.ps);
        cnt = mp.PR - pi0;
      }
      mp.colstart[pj + 1] = mp.colstart[pj] + cnt;
    }
    npatch = mp.colstart[mp.PC];
    if (npatch == 0)
      return;
  }
  mp.total = npatch << (mp.ps + mp.psc);
  mp.xcd = (mp.ps > 0) ? 1 : 0;
  long grid = mp.total;
  mp.persist = 0;
  mp.lockstep = 0;
  mp.kphase_ticks = 0;
  mp.counters = nullptr;
  mp.excl_rank = 0;
  mp.excl_budget = 0;
  mp.steal = 0;
  // Exclusive compute units only in whole rounds over the shader engines (one compute unit of every engine of every
  // XCD per round: 64 slots on MI355X); otherwise -- or without a probe, or with lockstep pacing -- the launch leaves
  // the slots free instead, spread by the dispatcher.
  long excl_rank = 0;
  if (excl_slots > 0) {
    static const int per_cu = update_blocks_per_cu<T>();
    const long round = (long) per_cu * g_probe_xcds * g_probe_engines;
    if (g_probe_xcds > 0 && !update_lockstep() && excl_slots % round == 0 &&
        (excl_slots / round) * g_probe_engines <= g_probe_cus_per_xcd / 2)
      excl_rank = (excl_slots / round) * g_probe_engines;
    else {
      max_blocks = std::max<long>(8, max_blocks - excl_slots);
      excl_slots = 0;
    }
  }
  if (max_blocks > 0 && counters != nullptr && mp.total > max_blocks) {
    grid = mp.xcd ? (max_blocks / 8) * 8 : max_blocks;  // equal number of workgroups per XCD range
    if (grid < 8)
      grid = 8;
    mp.persist = 1;
    mp.counters = counters;
    mp.lockstep = update_lockstep() ? 1 : 0;
    mp.kphase_ticks = kphase_ticks<T>((int) (grid));
    mp.steal = (update_steal() && !mp.lockstep) ? 1 : 0;
    mp.cu_busy = (role == 0) ? g_cu_busy : nullptr;
    ++g_launch_stats[0];
    if (excl_rank > 0) {
      mp.excl_rank = (int) excl_rank;
      mp.excl_budget = (unsigned) excl_slots;
      ++g_launch_stats[1];
    }
    if (!counters_are_zero)
      (void) hipMemsetAsync(counters, 0, 16 * sizeof(unsigned), stream);
  }
  const bool vec = aligned16<T>(a.a, a.lda) && aligned16<T>(a.a, a.a_ts) && aligned16<T>(a.b, a.ldb) &&
                   aligned16<T>(a.b, a.b_ts) && aligned16<T>(a.b, a.b_ts2) &&
                   (a.K1 == 0 || (aligned16<T>(a.a2, 0) && aligned16<T>(a.b2, 0)));
  auto go = [&](auto vtag, auto rtag) {
    constexpr bool V = decltype(vtag)::value;
    constexpr int RL = decltype(rtag)::value;
    hipLaunchKernelGGL((update_kernel<T, V, RL>), dim3((unsigned) grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, a, mp);
  };
  auto by_role = [&](auto vtag) {
    switch (role) {
      case 0: go(vtag, std::integral_constant<int, 0>{}); break;
      case 1: go(vtag, std::integral_constant<int, 1>{}); break;
      case 3: go(vtag, std::integral_constant<int, 3>{}); break;
      default: go(vtag, std::integral_constant<int, 2>{}); break;
    }
  };
  if (vec)
    by_role(std::true_type{});
  else
    by_role(std::false_type{});
}

// resident workgroups per compute unit of the bulk instantiation (for sizing persistent grids)
template <class T>
int update_blocks_per_cu() {
  using Cfg = typename UpdateCfg<T>::type;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&update_kernel<T, true, 0>), Cfg::THREADS,
                                                   Cfg::LDS_BYTES) != hipSuccess || n < 1)
    n = 1;
  return n;
}

template <class T>
static void update_init_one() {
  using Cfg = typename UpdateCfg<T>::type;
#define SET_ONE(V, RL)                                                                         \
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&update_kernel<T, V, RL>),           \
                             hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES)
  SET_ONE(true, 0);
  SET_ONE(true, 1);
  SET_ONE(true, 2);
  SET_ONE(true, 3);
  SET_ONE(false, 0);
  SET_ONE(false, 1);
  SET_ONE(false, 2);
  SET_ONE(false, 3);
#undef SET_ONE
}

// One probe launch: which (XCD, compute unit) pairs does a kernel of this process reach?  The compute units of each
// XCD are ranked shader engine by shader engine (cu 0 of every engine first), so that "the first r" are spread over
// the engines' dispatchers.
static void probe_compute_units() {
  const bool verbose = [] {
    const char* e = std::getenv("DLAF_MI355X_PROBE_VERBOSE");
    return e && std::atoi(e) != 0;
  }();
  auto note = [&](const char* what, hipError_t err) {
    if (verbose)
      std::fprintf(stderr, "[dlaf_mi355x] compute-unit probe: %s: %s\n", what, hipGetErrorString(err));
    (void) hipGetLastError();
  };
  unsigned* seen = nullptr;
  hipError_t err = hipMalloc(reinterpret_cast<void**>(&seen), 64 * sizeof(unsigned));
  if (err != hipSuccess)
    return note("hipMalloc", err);
  unsigned h[64] = {};
  err = hipMemset(seen, 0, sizeof(h));
  if (err == hipSuccess) {
    hipLaunchKernelGGL(cu_probe_kernel, dim3(8192), dim3(64), 48 * 1024, nullptr, seen);
    err = hipGetLastError();
  }
  if (err == hipSuccess)
    err = hipMemcpy(h, seen, sizeof(h), hipMemcpyDeviceToHost);
  (void) hipFree(seen);
  if (err != hipSuccess)
    return note("probe launch", err);
  static unsigned char rank[8][256];
  std::memset(rank, 255, sizeof(rank));
  int xcds = 0, min_cus = 1 << 30, max_engines = 0;
  for (int x = 0; x < 8; ++x) {
    std::vector<unsigned> keys;
    for (unsigned key = 0; key < 256; ++key)
      if (h[x * 8 + (key >> 5)] >> (key & 31) & 1u)
        keys.push_back(key);
    if (keys.empty())
      continue;
    ++xcds;
    min_cus = std::min<int>(min_cus, (int) keys.size());
    // key = se[7:5] sh[4] cu[3:0].  The workgroups of a dispatch go round-robin over the shader engines of an XCD
    // and wait for "their" engine (measured: tools/overlap_bench.hip -- compute units vacated in two of the four
    // engines leave a side kernel stuck behind the first workgroup that is routed to a full engine), so the rank
    // order takes one compute unit of every engine in turn: ordinal within the engine first, engine second.
    std::vector<unsigned> ord(keys.size());
    int engines = 0;
    for (size_t i = 0; i < keys.size(); ++i) {
      unsigned n = 0;
      bool first = true;
      for (size_t j = 0; j < keys.size(); ++j)
        if ((keys[j] >> 4) == (keys[i] >> 4)) {
          if ((keys[j] & 15u) < (keys[i] & 15u))
            ++n;
          if (j < i)
            first = false;
        }
      ord[i] = n;
      engines += first ? 1 : 0;
    }
    std::vector<size_t> idx(keys.size());
    for (size_t i = 0; i < idx.size(); ++i)
      idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
      return std::make_pair(ord[a], keys[a] >> 4) < std::make_pair(ord[b], keys[b] >> 4);
    });
    for (size_t r = 0; r < idx.size(); ++r)
      rank[x][keys[idx[r]]] = (unsigned char) std::min<size_t>(r, 254);
    max_engines = std::max(max_engines, engines);
    {
      std::vector<unsigned> sorted(keys.size());
      for (size_t r = 0; r < idx.size(); ++r)
        sorted[r] = keys[idx[r]];
      keys.swap(sorted);
    }
    if (verbose) {
      std::fprintf(stderr, "[dlaf_mi355x] compute-unit probe: XCD %d: %zu compute units:", x, keys.size());
      for (unsigned k : keys)
        std::fprintf(stderr, " %u.%u.%u", k >> 5, (k >> 4) & 1u, k & 15u);
      std::fprintf(stderr, "\n");
    }
  }
  if (xcds == 0)
    return note("no compute unit seen", hipSuccess);
  err = hipMemcpyToSymbol(HIP_SYMBOL(g_cu_rank), rank, sizeof(rank));
  if (err != hipSuccess)
    return note("hipMemcpyToSymbol", err);
  g_probe_xcds = xcds;
  g_probe_cus_per_xcd = min_cus;
  g_probe_engines = std::max(1, max_engines);
}

void update_kernels_init() {
  update_init_one<float>();
  update_init_one<double>();
  update_init_one<cfloat>();
  update_init_one<cdouble>();
  probe_compute_units();
  (void) cu_busy_table();
}

template void launch_update<float>(const UpdateArgs<float>&, hipStream_t, int, long, unsigned*, bool, long);
template int update_blocks_per_cu<float>();
template void launch_update<double>(const UpdateArgs<double>&, hipStream_t, int, long, unsigned*, bool, long);
template int update_blocks_per_cu<double>();
template void launch_update<cfloat>(const UpdateArgs<cfloat>&, hipStream_t, int, long, unsigned*, bool, long);
template int update_blocks_per_cu<cfloat>();
template void launch_update<cdouble>(const UpdateArgs<cdouble>&, hipStream_t, int, long, unsigned*, bool, long);
template int update_blocks_per_cu<cdouble>();

}  // namespace dlaf_mi355x
