// kernels_potrf_coop.hip -- Cholesky of one diagonal tile (kb x kb, kb <= nb) as ONE resident,
// cooperative launch: ceil(kb/64) workgroups, workgroup s owns the 64-row strip s of the tile.
//
// Reference: tile::potrf (lapack/tile.h:577-606, one rocsolver_*potrf call).  The first version of
// this build issued ~3 dependent launches per 64 columns (diagonal block, sub-panel TRSM, in-tile
// update); measured on MI355X those ~48 launches either leave the GPU idle (serial order) or starve
// behind the queued workgroups of the bulk trailing update (stream priorities do not pre-empt them).
// One launch whose few workgroups stay resident has neither problem: it runs beside the bulk update.
//
// Right-looking over 64-column steps j; strip s takes part in steps j <= s:
//   j == s : block (s,s) is final -> factor + invert in LDS (potrf_diag_core.hpp), write L_jj and
//            inv(L_jj), publish flag[j], done;
//   j <  s : wait flag[j]; X_s = A(s,j) * inv(L_jj)^H (MFMA); store X_s; arrive on cnt[j]; wait until
//            all strips below j arrived; A(s,c) -= X_s * X_c^H for c = j+1..s (MFMA, X_s kept in LDS).
// Inter-workgroup hand-offs follow the placement-independent protocol of the CDNA4 guide
// (cdna_hip_programming.md, Guideline 16 / split-K recipe): handed-off bytes are write-through (sc1)
// stores, every storing wave drains vmcnt, workgroup barrier, one relaxed agent atomic; the consumer
// polls relaxed, then one agent-scope acquire fence, drained wait, workgroup barrier, plain loads.
// Every spin is bounded.
#include <cstdio>
#include <cstdlib>

#include "potrf_diag_core.hpp"

namespace dlaf_mi355x {

constexpr int kCB = kDiagBlock;  // 64

template <class T>
struct CoopCfg {
  using R = real_t<T>;
  static constexpr bool CX = TypeInfo<T>::is_complex;
  static constexpr int NPL = CX ? 2 : 1;
  // complex<double>: the footprint has to stay below 96 KiB (160 KiB per CU minus the 64 KiB of a bulk-update
  // workgroup) or no strip of a complex tile ever finds a CU while a persistent bulk launch is running.  REGA: the
  // row operand of the MFMAs lives in registers (straight from global memory for the solve, the accumulators of
  // the solve fed back for the updates: f64 accumulator register v of lane group g is exactly the k index the
  // next MFMA wants), so only ONE operand image is staged; the diagonal step packs W into L's planes.
  static constexpr bool REGA = CX && sizeof(R) == 8;
  static constexpr int PAD = 16;
  static constexpr int LD = kCB + PAD;               // operand images [k][64 + pad]
  static constexpr int IMG = kCB * LD;               // one plane of one operand
  static constexpr int OPERANDS = (REGA ? 1 : 2) * NPL * IMG;
  static constexpr int DIAG = diag_lds_elems<T, REGA>();
  static constexpr int ELEMS = OPERANDS > DIAG ? OPERANDS : DIAG;
  static constexpr int LDS_BYTES = ELEMS * (int) sizeof(R);
  static_assert(!REGA || LDS_BYTES <= 95 * 1024, "must fit beside one bulk-update workgroup");
};

constexpr unsigned kCoopFailed = 0x40000000u;
constexpr long kCoopSpinLimit = 40000000;  // x (s_sleep + L2 round trip) >> any legitimate wait
// the bound every wait of a launch uses: kCoopSpinLimit, or DLAF_MI355X_POTRF_SPIN_LIMIT (tests use a tiny value
// to drive the expiry path: flag -> kInfoSchedulingFailure -> the host refuses the result)
static long coop_spin_limit() {
  static const long v = [] {
    const char* e = std::getenv("DLAF_MI355X_POTRF_SPIN_LIMIT");
    return e ? std::atol(e) : kCoopSpinLimit;
  }();
  return v;
}

// Write-through (sc1) store of one element: the bytes another workgroup will read are stored this way,
// so publishing needs NO agent-scope release fence.  That fence (buffer_wbl2) writes back every dirty
// line of the XCD's L2 -- beside the bulk trailing update, which keeps megabytes of C tiles dirty, it
// made each hand-off cost tens of microseconds (POTRF(1024): 1.5 ms alone, 5-7 ms beside the update).
template <class T>
__device__ __forceinline__ void store_wt(T* p, const T& v) {
  if constexpr (sizeof(T) == 4) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  else if constexpr (sizeof(T) == 8) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  else {
    struct Two {
      unsigned long long a, b;
    };
    const Two t = __builtin_bit_cast(Two, v);
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, t.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, t.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// all threads call.  Hand-off form "write-through stores + drained flag" (cdna_hip_programming.md,
// split-K recipe / Guideline 16 R1): every byte a consumer reads was stored with store_wt; every wave
// drains its stores, the workgroup barrier orders all of them before the one relaxed agent atomic.
// The consumer polls, then takes ONE agent-scope acquire (L1 invalidate) before plain loads.
__device__ __forceinline__ void coop_publish(unsigned* word, unsigned value, bool add) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (add)
      __hip_atomic_fetch_add(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// all threads call; returns the observed value (>= target), or 0xFFFFFFFF after the spin bound
__device__ __forceinline__ unsigned coop_wait(unsigned* word, unsigned target, unsigned* shared_slot, long spin_limit) {
  if (threadIdx.x == 0) {
    unsigned v;
    long spins = 0;
    while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > spin_limit) {
        v = 0xFFFFFFFFu;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *shared_slot = v;
  }
  __syncthreads();
  const unsigned r = *shared_slot;
  __syncthreads();
  return r;
}

// all threads call: waits until every word of words[0 .. n) is non-zero (thread i polls word i); returns false
// after the spin bound.  n <= kThreads.
__device__ __forceinline__ bool coop_wait_all(unsigned* words, int n, unsigned* shared_slot, long spin_limit) {
  if (threadIdx.x == 0)
    *shared_slot = 1u;
  __syncthreads();
  if ((int) threadIdx.x < n) {
    long spins = 0;
    while (__hip_atomic_load(&words[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > spin_limit) {
        *shared_slot = 0u;
        break;
      }
    }
  }
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const bool ok = *shared_slot != 0u;
  __syncthreads();
  return ok;
}

// global (rows x cols, ld) block -> LDS operand image [k = col][m = row], zero-filled to 64 x 64.
// Two halves so that a caller can keep the 16 loads of a block in flight behind other work: the
// kernel is latency-bound and, beside the bulk update, a global round trip costs several microseconds.
constexpr int kCoopPerThread = kCB * kCB / kThreads;  // 16

template <class T>
__device__ __forceinline__ void coop_fetch(T (&regs)[kCoopPerThread], const T* g, long ld, int rows, int cols) {
#pragma unroll
  for (int q = 0; q < kCoopPerThread; ++q) {
    const int idx = threadIdx.x + q * kThreads;
    const int m = idx % kCB, k = idx / kCB;
    regs[q] = (m < rows && k < cols) ? g[m + (long) k * ld] : zero_el<T>();
  }
}

template <class T>
__device__ __forceinline__ void coop_commit(real_t<T>* img, const T (&regs)[kCoopPerThread]) {
  using C = CoopCfg<T>;
#pragma unroll
  for (int q = 0; q < kCoopPerThread; ++q) {
    const int idx = threadIdx.x + q * kThreads;
    const int m = idx % kCB, k = idx / kCB;
    img[k * C::LD + m] = re_of(regs[q]);
    if constexpr (C::CX)
      img[C::IMG + k * C::LD + m] = im_of(regs[q]);
  }
}

template <class T>
__device__ __forceinline__ void coop_load_image(real_t<T>* img, const T* g, long ld, int rows, int cols) {
  T regs[kCoopPerThread];
  coop_fetch<T>(regs, g, ld, rows, cols);
  coop_commit<T>(img, regs);
}

// acc(m = wave*16 + c, n = j*16 + irow(g, v)) = sum_k A[m][k] * conj(B[n][k]) over the 64 x 64 images
template <class T>
__device__ __forceinline__ void coop_mma64(const real_t<T>* A, const real_t<T>* B,
                                           typename Mma<real_t<T>>::acc_t (&re)[4],
                                           typename Mma<real_t<T>>::acc_t (&im)[4]) {
  using C = CoopCfg<T>;
  using R = real_t<T>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    re[j] = typename Mma<R>::acc_t{0, 0, 0, 0};
    im[j] = typename Mma<R>::acc_t{0, 0, 0, 0};
  }
#pragma unroll 4
  for (int k4 = 0; k4 < kCB / 4; ++k4) {
    const int kk = 4 * k4 + g;
    const R a_re = A[kk * C::LD + wave * 16 + c];
    R a_im = 0;
    if constexpr (C::CX)
      a_im = A[C::IMG + kk * C::LD + wave * 16 + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const R b_re = B[kk * C::LD + j * 16 + c];
      re[j] = Mma<R>::mma(b_re, a_re, re[j]);
      if constexpr (C::CX) {
        const R b_im = B[C::IMG + kk * C::LD + j * 16 + c];
        re[j] = Mma<R>::mma(b_im, a_im, re[j]);
        im[j] = Mma<R>::mma(b_re, a_im, im[j]);
        im[j] = Mma<R>::mma(b_im, -a_re, im[j]);
      }
    }
  }
}

// the same with the row operand in registers: a_re[k4] / a_im[k4] = A[m = wave*16 + c][k = 4 k4 + g]; the
// accumulators are NOT cleared (the caller preloads them, e.g. with the block the product is subtracted from).
// SUB: acc -= A B^H instead of +=.  Every sign sits in the MFMA's negate bit (Mma::mma_neg): a negated copy of the
// operand registers would be loop-invariant, hoisted, and cost 32 registers the kernel does not have.
template <class T, bool SUB>
__device__ __forceinline__ void coop_mma64_rega(const real_t<T> (&a_re)[16], const real_t<T> (&a_im)[16],
                                                const real_t<T>* B, typename Mma<real_t<T>>::acc_t (&re)[4],
                                                typename Mma<real_t<T>>::acc_t (&im)[4]) {
  using C = CoopCfg<T>;
  using R = real_t<T>;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int k4 = 0; k4 < kCB / 4; ++k4) {
    const int kk = 4 * k4 + g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const R b_re = B[kk * C::LD + j * 16 + c];
      const R b_im = B[C::IMG + kk * C::LD + j * 16 + c];
      if constexpr (!SUB) {
        re[j] = Mma<R>::mma(b_re, a_re[k4], re[j]);
        re[j] = Mma<R>::mma(b_im, a_im[k4], re[j]);
        im[j] = Mma<R>::mma(b_re, a_im[k4], im[j]);
        im[j] = Mma<R>::mma_neg(b_im, a_re[k4], im[j]);
      }
      else {
        re[j] = Mma<R>::mma_neg(b_re, a_re[k4], re[j]);
        re[j] = Mma<R>::mma_neg(b_im, a_im[k4], re[j]);
        im[j] = Mma<R>::mma_neg(b_re, a_im[k4], im[j]);
        im[j] = Mma<R>::mma(b_im, a_re[k4], im[j]);
      }
    }
  }
}

// A strip that leaves without factoring its diagonal block fills ITS block of winv with NaN: on a process grid the
// tile and winv are broadcast whatever happened, the other ranks do not see this rank's status word, and what they
// compute from a failed tile must not depend on what a fresh allocation happened to hold (all-zero inverse blocks
// give a clean "factorization" of garbage; NaN makes every later diagonal tile flag its first pivot).
template <class T>
__device__ __forceinline__ void coop_poison_winv(T* Ws) {
  using R = real_t<T>;
  const R nan = __builtin_nan("");
  for (int idx = threadIdx.x; idx < kCB * kCB; idx += kThreads)
    store_wt(&Ws[idx], make_el<T>(nan, nan));
}

template <class T>
__device__ __forceinline__ void potrf_coop_body(T* __restrict__ tile, int ld, int kb, T* __restrict__ winv, int* info,
                                                int info_base, unsigned* sync, long spin_limit, int prio,
                                                unsigned long long* trace) {
  using C = CoopCfg<T>;
  using R = real_t<T>;
  using acc_t = typename Mma<R>::acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* lds = reinterpret_cast<R*>(lds_raw);
  __shared__ int fail_col;
  __shared__ unsigned wait_slot;
  const int G = gridDim.x, s = blockIdx.x;
  unsigned* flag = sync;       // flag[j] = 1: L_jj and inv(L_jj) are in memory;  kCoopFailed: not SPD
  unsigned* xflag = sync + G;  // xflag[j * G + c] = 1: X(c, j) of strip c is in memory
  // Every wait of strip s is on a strip with a LOWER block id (flag[j], j < s; xflag[j][c], c < s): with
  // in-order dispatch the workgroups it depends on are already running or done, so the kernel makes progress
  // whether or not all its workgroups are co-resident (a plain launch suffices).
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int rows_s = min(kCB, kb - kCB * s);

  // The entry decision is made ONCE per workgroup: the status word may change between the moments the waves of a
  // strip get here (another strip flagging its pivot while this one is being dispatched), and a workgroup whose
  // waves took different exits would run its barriers, its LDS hand-offs and -- thread 0 being the only writer of
  // `info` and of the flags -- its failure path with a wave missing.
  if (t == 0)
    wait_slot = (unsigned) __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned info_on_entry = wait_slot;
  __syncthreads();  // (wait_slot is reused by the waits below)
  // (diagnosis hook, DLAF_MI355X_POTRF_TRACE: what the first two strips saw on entry)
  if (trace != nullptr && s < 2 && lane == 0) {
    trace[s * 8 + wave * 2] = (1ull << 63) | info_on_entry;
    trace[s * 8 + wave * 2 + 1] = wall_clock64();
  }
  if (info_on_entry != 0) {
    if (trace != nullptr && s < 2 && t == 0)
      trace[16 + s * 8 + 6] = 1;  // left at the entry check
    coop_poison_winv<T>(winv + (long) s * kCB * kCB);
    return;
  }
  // The strips sit on compute units they share with waves of the bulk update (16 strips cannot own CUs of a device
  // whose every CU holds persistent update workgroups): raised wave priority lets the SIMD's arbiter issue the
  // strip's instructions first, the update waves fill the slots it leaves.
  if (prio)
    __builtin_amdgcn_s_setprio(3);

  R* Aimg = lds;
  R* Bimg = C::REGA ? lds : lds + C::NPL * C::IMG;

  for (int j = 0; j < s; ++j) {
    const int jb = kCB;  // every block column left of my diagonal block is full
    // A(s,j) is final since my own update of step j-1: its loads travel while I wait for the owner of step j
    T* const Asj_pre = tile + (long) kCB * s + (long) kCB * j * ld;
    T pre_a[C::REGA ? 1 : kCoopPerThread];
    real_t<T> pre_re[C::REGA ? 16 : 1], pre_im[C::REGA ? 16 : 1];
    if constexpr (!C::REGA) {
      coop_fetch<T>(pre_a, Asj_pre, ld, rows_s, jb);
    }
    else {
      const int m = (t >> 6) * 16 + c;
      const unsigned lane_off = (unsigned) sizeof(T) * (unsigned) (m + g * ld);
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        const T el = m < rows_s ? *reinterpret_cast<const T*>(reinterpret_cast<const char*>(Asj_pre + (long) (4 * k4) * ld) + lane_off)
                                : zero_el<T>();
        pre_re[k4] = re_of(el);
        pre_im[k4] = im_of(el);
      }
    }
    const unsigned f = coop_wait(&flag[j], 1u, &wait_slot, spin_limit);
    if (trace != nullptr && s < 2 && j == 0 && (t & 63) == 0)
      trace[16 + s * 8 + 5] = ((unsigned long long) (t >> 6) << 40) | (1ull << 32) | f;
    if (f != 1u) {
      if (trace != nullptr && s < 2 && t == 0)
        trace[16 + s * 8 + 6] = 2;  // left after the owner's flag
      // not positive definite (info already set by the owner), or the spin bound was hit
      if (f == 0xFFFFFFFFu && t == 0)
        atomicCAS(info, 0, kInfoSchedulingFailure);
      coop_poison_winv<T>(winv + (long) s * kCB * kCB);
      return;
    }
    // ---- X_s = A(s,j) * inv(L_jj)^H ----------------------------------------------------------------
    T* Asj = tile + (long) kCB * s + (long) kCB * j * ld;
    if constexpr (C::REGA) {
      // addresses = (uniform base, scalar registers) + (one 32-bit lane offset): the operand element of lane (g, c)
      // for k = 4 k4 + g and its accumulator element for n = 16 jt + 4 v + g share the lane part m + g ld
      auto at = [](T* base, unsigned byte_off) {
        return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
      };
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      const int m = wv * 16 + c;
      const unsigned lane_off = (unsigned) sizeof(T) * (unsigned) (m + g * ld);
      const unsigned img_off = (unsigned) sizeof(T) * (unsigned) ((t & 63) + (t >> 6) * ld);  // image element (m, k) of thread t
      const bool m_ok = m < rows_s;
      // B image [k][m] <- 64 columns of a global block with `rows` valid rows (column stride gld), in two halves
      auto load_b = [&](T* gsrc, int gld, unsigned goff, int rows) {
        const bool ok = (t & 63) < rows;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          T regs[8];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            regs[q] = ok ? *at(gsrc + (long) (4 * (8 * h + q)) * gld, goff) : zero_el<T>();
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int k = (t >> 6) + 4 * (8 * h + q);
            Bimg[k * C::LD + (t & 63)] = re_of(regs[q]);
            Bimg[C::IMG + k * C::LD + (t & 63)] = im_of(regs[q]);
          }
        }
      };
      R a_re[16], a_im[16];
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        a_re[k4] = pre_re[k4];
        a_im[k4] = pre_im[k4];
      }
      load_b(winv + (long) j * kCB * kCB, kCB, (unsigned) sizeof(T) * (unsigned) ((t & 63) + (t >> 6) * kCB), kCB);
      __syncthreads();
      acc_t xre[4], xim[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        xre[jt] = acc_t{0, 0, 0, 0};
        xim[jt] = acc_t{0, 0, 0, 0};
      }
      coop_mma64_rega<T, false>(a_re, a_im, Bimg, xre, xim);
      // X_s(m, n = 16 jt + g + 4 v) = operand element k4 = 4 jt + v of the updates below, which accumulate
      // C - X_s X_c^H on top of the preloaded C block
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (m_ok)
            store_wt(at(Asj + (long) (16 * jt + 4 * v) * ld, lane_off), make_el<T>(xre[jt][v], xim[jt][v]));
          a_re[4 * jt + v] = xre[jt][v];
          a_im[4 * jt + v] = xim[jt][v];
        }
      coop_publish(&xflag[(long) j * G + s], 1u, false);
      if (s - j - 1 > 0 && !coop_wait_all(&xflag[(long) j * G + j + 1], s - j - 1, &wait_slot, spin_limit)) {
        if (t == 0)
          atomicCAS(info, 0, kInfoSchedulingFailure);
        coop_poison_winv<T>(winv + (long) s * kCB * kCB);
        return;
      }
      // ---- A(s,c) -= X_s * X_c^H for c = j+1 .. s: the C block is loaded INTO the accumulators ---------
#pragma unroll 1
      for (int cc = j + 1; cc <= s; ++cc) {
        const int rows_c = min(kCB, kb - kCB * cc);
        T* Csc = tile + (long) kCB * s + (long) kCB * cc * ld;
        const bool diag = (cc == s);
        // the lane parts of every address and predicate below are made opaque once per block: computed from loop
        // invariants they would all be hoisted out of this loop (16 global offsets, 32 LDS addresses, 16 masks) and
        // spilled; recomputed here they are an add or an immediate offset each
        unsigned lo = lane_off;
        int ml = m, gl = g;
        asm volatile("" : "+v"(lo), "+v"(ml), "+v"(gl));
        const bool ml_ok = ml < rows_s;
        acc_t ure[4], uim[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int n = jt * 16 + 4 * v + gl;
            const T el = (ml_ok && n < rows_c && (!diag || ml >= n)) ? *at(Csc + (long) (16 * jt + 4 * v) * ld, lo)
                                                                     : zero_el<T>();
            ure[jt][v] = re_of(el);
            uim[jt][v] = im_of(el);
          }
        __syncthreads();  // every wave is done with the previous B image (winv, or X of the previous block)
        if (!diag) {
          load_b(tile + (long) kCB * cc + (long) kCB * j * ld, ld, img_off, rows_c);
        }
        else {
          // my own X_s is the column operand of the diagonal block: image [k][m] straight from the
          // operand registers, no trip through memory
          R* bi = Bimg + gl * C::LD + ml;
#pragma unroll
          for (int k4 = 0; k4 < 16; ++k4) {
            bi[(4 * k4) * C::LD] = a_re[k4];
            bi[C::IMG + (4 * k4) * C::LD] = a_im[k4];
          }
        }
        __syncthreads();
        coop_mma64_rega<T, true>(a_re, a_im, Bimg, ure, uim);
        const bool to_lds = diag && (j == s - 1);  // (see the real-type path: straight into the L planes)
        if (to_lds) {
          __syncthreads();
          R* Lre = lds + gl * kPDLd + ml;
          R* Lim = Lre + kPD * kPDLd;
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int n = jt * 16 + 4 * v + gl;
              const bool in = (ml_ok && n < rows_c && ml >= n);
              Lre[(jt * 16 + 4 * v) * kPDLd] = in ? ure[jt][v] : ((ml == n && ml >= rows_s) ? R(1) : R(0));
              Lim[(jt * 16 + 4 * v) * kPDLd] = (in && ml != n) ? uim[jt][v] : R(0);
            }
        }
        else {
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int n = jt * 16 + 4 * v + gl;
              if (ml_ok && n < rows_c && (!diag || ml >= n))
                *at(Csc + (long) (16 * jt + 4 * v) * ld, lo) =
                    make_el<T>(ure[jt][v], (diag && ml == n) ? R(0) : uim[jt][v]);
            }
        }
      }
      __syncthreads();  // the next step (or the diagonal phase) rewrites the image
      continue;
    }
    if constexpr (!C::REGA) {
      T wreg[kCoopPerThread];
      coop_fetch<T>(wreg, winv + (long) j * kCB * kCB, kCB, kCB, kCB);
      coop_commit<T>(Aimg, pre_a);
      coop_commit<T>(Bimg, wreg);
    }
    __syncthreads();
    acc_t xre[4], xim[4];
    coop_mma64<T>(Aimg, Bimg, xre, xim);
    __syncthreads();  // every wave is done reading the A image before it is overwritten with X_s
    {
      int mx = wave * 16 + c, gx = g;
      asm volatile("" : "+v"(mx), "+v"(gx));  // (addresses from the lane parts in place, not hoisted and spilled)
      R* Ax = Aimg + Mma<R>::irow(gx, 0) * C::LD + mx;
      T* Xg = Asj + mx + (long) Mma<R>::irow(gx, 0) * ld;
      const bool mx_ok = mx < rows_s;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int n0 = jt * 16 + Mma<R>::irow(0, v);
          // X_s stays in LDS as the row operand of this step's updates: image [k = n][m]
          Ax[n0 * C::LD] = xre[jt][v];
          if constexpr (C::CX)
            Ax[C::IMG + n0 * C::LD] = xim[jt][v];
          if (mx_ok)
            store_wt(&Xg[(long) n0 * ld], make_el<T>(xre[jt][v], C::CX ? xim[jt][v] : R(0)));
        }
    }
    coop_publish(&xflag[(long) j * G + s], 1u, false);
    // the updates below read X(c, j) of the strips j < c < s (my own X_s is in LDS)
    if (s - j - 1 > 0 && !coop_wait_all(&xflag[(long) j * G + j + 1], s - j - 1, &wait_slot, spin_limit)) {
      if (t == 0)
        atomicCAS(info, 0, kInfoSchedulingFailure);
      coop_poison_winv<T>(winv + (long) s * kCB * kCB);
      return;
    }
    // ---- A(s,c) -= X_s * X_c^H for c = j+1 .. s ----------------------------------------------------
    // the C block (and, for real types, X_{c+1}) is fetched while the MFMAs of block c run; complex
    // types have no registers to spare for the X prefetch beside a co-resident bulk-update wave
    constexpr bool kPrefetch = !C::CX;
    T nextB[kPrefetch ? kCoopPerThread : 1];
    // (the column operand of my own diagonal block is my own X_s: it is in LDS already)
    if constexpr (kPrefetch)
      if (j + 1 < s)
        coop_fetch<T>(nextB, tile + (long) kCB * (j + 1) + (long) kCB * j * ld, ld, min(kCB, kb - kCB * (j + 1)), jb);
    for (int cc = j + 1; cc <= s; ++cc) {
      const int rows_c = min(kCB, kb - kCB * cc);
      const bool diag = (cc == s);
      if (!diag) {
        if constexpr (kPrefetch)
          coop_commit<T>(Bimg, nextB);
        else
          coop_load_image<T>(Bimg, tile + (long) kCB * cc + (long) kCB * j * ld, ld, rows_c, jb);
      }
      __syncthreads();
      if constexpr (kPrefetch)
        if (cc + 1 < s)
          coop_fetch<T>(nextB, tile + (long) kCB * (cc + 1) + (long) kCB * j * ld, ld, min(kCB, kb - kCB * (cc + 1)),
                        jb);
      T* Csc = tile + (long) kCB * s + (long) kCB * cc * ld;
      // lane parts made opaque once per block (see the register-operand path): hoisted out of this loop the 16
      // addresses and masks would be spilled
      int ml = wave * 16 + c, gl = g;
      asm volatile("" : "+v"(ml), "+v"(gl));
      const bool ml_ok = ml < rows_s;
      T* Cl = Csc + ml + (long) Mma<R>::irow(gl, 0) * ld;  // irow is linear: irow(g, v) = irow(g, 0) + irow(0, v)
      T cv[4][4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int n = jt * 16 + Mma<R>::irow(gl, v);
          cv[jt][v] = (ml_ok && n < rows_c && (!diag || ml >= n)) ? Cl[(long) (jt * 16 + Mma<R>::irow(0, v)) * ld]
                                                                  : zero_el<T>();
        }
      acc_t ure[4], uim[4];
      coop_mma64<T>(Aimg, diag ? Aimg : Bimg, ure, uim);
      // The last update before my diagonal step (block (s,s) under column s-1) goes straight into the L planes of
      // that step instead of to memory and back: two global round trips less on the critical path of the tile.
      const bool to_lds = diag && (j == s - 1);
      if (to_lds)
        __syncthreads();  // the L / W planes alias the operand images: every wave is done reading them
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int m = ml, n = jt * 16 + Mma<R>::irow(gl, v);
          const bool in = (ml_ok && n < rows_c && (!diag || m >= n));
          T r = cv[jt][v];
          if constexpr (C::CX) {
            r = T{r.re - ure[jt][v], (diag && m == n) ? R(0) : r.im - uim[jt][v]};
          }
          else {
            r = r - ure[jt][v];
          }
          if (to_lds) {
            R* Lre = lds;
            R* Lim = Lre + kPD * kPDLd;
            R* Wre = Lre + C::NPL * kPD * kPDLd;
            R* Wim = Wre + kPD * kPDLd;
            Lre[n * kPDLd + m] = in ? re_of(r) : ((m == n && m >= rows_s) ? R(1) : R(0));
            Wre[n * kPDLd + m] = 0;
            if constexpr (C::CX) {
              Lim[n * kPDLd + m] = (in && m != n) ? im_of(r) : R(0);
              Wim[n * kPDLd + m] = 0;
            }
          }
          else if (in) {
            Cl[(long) (jt * 16 + Mma<R>::irow(0, v)) * ld] = r;
          }
        }
      __syncthreads();  // B image is rewritten by the next iteration
    }
    // my own stores to block (s, .) must be visible to my own later loads: same CU, ordered by the
    // barrier above; the next step's first action is a wait on flag[j+1] anyway
  }

  // ---- j == s: my diagonal block is final ---------------------------------------------------------
  {
    constexpr bool PACK = C::REGA;
    R* Lre = lds;
    R* Lim = Lre + kPD * kPDLd;
    R* Wre = PACK ? Lre : Lre + C::NPL * kPD * kPDLd;
    R* Wim = Wre + kPD * kPDLd;
    T* Ass = tile + (long) kCB * s + (long) kCB * s * ld;
    const int jb = rows_s;
    if (t == 0)
      fail_col = -1;
    // s > 0: the last update of the loop above left the block in the L planes (and zeroed W)
    if (s == 0) {
      T regs[kCoopPerThread];
#pragma unroll
      for (int q = 0; q < kCoopPerThread; ++q) {
        const int idx = t + q * kThreads;
        const int r = idx % kPD, cl = idx / kPD;
        regs[q] = (r < jb && cl < jb && r >= cl) ? Ass[r + (long) cl * ld] : zero_el<T>();
      }
#pragma unroll
      for (int q = 0; q < kCoopPerThread; ++q) {
        const int idx = t + q * kThreads;
        const int r = idx % kPD, cl = idx / kPD;
        R re = re_of(regs[q]), imv = im_of(regs[q]);
        if (r == cl && r >= jb)
          re = 1;
        Lre[cl * kPDLd + r] = re;
        if constexpr (C::CX)
          Lim[cl * kPDLd + r] = (r == cl) ? R(0) : imv;
        if constexpr (!PACK) {
          Wre[cl * kPDLd + r] = 0;
          if constexpr (C::CX)
            Wim[cl * kPDLd + r] = 0;
        }
      }
    }
    __syncthreads();
    if (trace != nullptr && s < 2 && t == 0)
      trace[16 + s * 8 + 0] = (unsigned long long) __builtin_bit_cast(unsigned long long, (double) Lre[0]);
    const int failed = diag_factor_invert<T, PACK>(Lre, Lim, Wre, Wim, jb, 1, &fail_col);
    if (trace != nullptr && s < 2 && t == 0)
      trace[16 + s * 8 + 1] = (1ull << 32) | (unsigned) failed;
    if (failed >= 0) {
      if (t == 0) {
        const int old = atomicCAS(info, 0, info_base + kCB * s + failed + 1);
        if (trace != nullptr && s < 2) {
          trace[16 + s * 8 + 2] = (1ull << 32) | (unsigned) old;
          trace[16 + s * 8 + 3] = wall_clock64();
          trace[16 + s * 8 + 4] = (1ull << 32) | (unsigned) __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          trace[16 + s * 8 + 6] = 3;  // flagged a pivot
        }
      }
      coop_poison_winv<T>(winv + (long) s * kCB * kCB);
      coop_publish(&flag[s], kCoopFailed, false);
      return;
    }
    T* Ws = winv + (long) s * kCB * kCB;
    for (int idx = t; idx < kPD * kPD; idx += kThreads) {
      const int rr = idx % kPD, cl = idx / kPD;
      if (rr < jb && cl < jb && rr >= cl) {
        R imv = 0;
        if constexpr (C::CX)
          imv = Lim[cl * kPDLd + rr];
        Ass[rr + (long) cl * ld] = make_el<T>(Lre[cl * kPDLd + rr], imv);
      }
      R wre = 0, wim = 0;
      if (rr < jb && cl < jb) {
        wre = diag_w_get<PACK>(Wre, PACK ? diag_wd_re<T, PACK>(Wre) : nullptr, rr, cl);
        if constexpr (C::CX)
          wim = diag_w_get<PACK>(Wim, PACK ? diag_wd_im<T, PACK>(Wre) : nullptr, rr, cl);
      }
      store_wt(&Ws[rr + (long) cl * kPD], make_el<T>(wre, wim));
    }
    coop_publish(&flag[s], 1u, false);
    if (trace != nullptr && s < 2 && t == 0)
      trace[16 + s * 8 + 6] = 4;  // factored
  }
}

// The strip counts itself in cu_busy[XCD][compute unit] while it runs: a persistent bulk-update workgroup on the same
// compute unit sits out between two work items until the strip has left (kernels_update.hip).
template <class T>
__global__ __launch_bounds__(kThreads, 2) void potrf_coop_kernel(T* __restrict__ tile, int ld, int kb,
                                                                  T* __restrict__ winv, int* info, int info_base,
                                                                  unsigned* sync, long spin_limit, int prio, int* cu_busy,
                                                                  unsigned long long* trace) {
  int* mine = nullptr;
  if (cu_busy != nullptr && threadIdx.x == 0) {
    unsigned xcc;
    const unsigned key = phys_cu_key(xcc);
    mine = cu_busy + (xcc * 256u + key);
    atomicAdd(mine, 1);
  }
  // (registration precedes the workgroup-uniform entry decision of the body: its first barrier)
  potrf_coop_body<T>(tile, ld, kb, winv, info, info_base, sync, spin_limit, prio, trace);
  __syncthreads();  // the strip leaves the table when ALL its waves are done, not when wave 0 is
  if (mine != nullptr)
    atomicSub(mine, 1);
}

// DLAF_MI355X_POTRF_PRIO=0 leaves the strips at the default wave priority (A/B runs)
static int coop_wave_prio() {
  static const int p = [] {
    const char* e = std::getenv("DLAF_MI355X_POTRF_PRIO");
    return e ? std::atoi(e) : 1;
  }();
  return p;
}

int* cu_busy_table();  // kernels_update.hip

// DLAF_MI355X_POTRF_TRACE=1 (diagnosis): the launches with info_base == 0 -- the first diagonal tile of a factorization --
// record what their first two strips saw into 32 device words (potrf_coop_trace_buffer(): zeroed by the caller)
static unsigned long long* g_potrf_trace = nullptr;
unsigned long long* potrf_coop_trace_buffer() {
  static const bool on = [] {
    const char* e = std::getenv("DLAF_MI355X_POTRF_TRACE");
    return e && std::atoi(e) != 0;
  }();
  if (on && g_potrf_trace == nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&g_potrf_trace), 32 * sizeof(unsigned long long)) != hipSuccess ||
        zero_device_now(g_potrf_trace, 32 * sizeof(unsigned long long)) != hipSuccess) {
      (void) hipGetLastError();
      g_potrf_trace = nullptr;
    }
  }
  return g_potrf_trace;
}

static void fatal_device_config(const char* what) {
  std::fprintf(stderr, "[dlaf_mi355x] %s\n", what);
  std::abort();
}

template <class T>
void launch_potrf_coop(T* tile, int ld, int kb, T* winv, int* info, int info_base, unsigned* sync,
                       hipStream_t stream, bool sync_is_zero, bool count_strips) {
  if (kb <= 0)
    return;
  const int G = (kb + kCB - 1) / kCB;
  if (G > kThreads)
    fatal_device_config("potrf_coop: more than 256 strips per tile");
  // (a caller that hands every launch its own slice of a buffer it zeroed once saves a fill kernel per tile on
  // the critical path of the factorization)
  if (!sync_is_zero)
    (void) hipMemsetAsync(sync, 0, sizeof(unsigned) * ((size_t) G + (size_t) G * G), stream);
  hipLaunchKernelGGL((potrf_coop_kernel<T>), dim3((unsigned) G), dim3(kThreads), CoopCfg<T>::LDS_BYTES, stream, tile, ld,
                     kb, winv, info, info_base, sync, coop_spin_limit(), coop_wave_prio(), count_strips ? cu_busy_table() : nullptr,
                     info_base == 0 ? potrf_coop_trace_buffer() : nullptr);
}

template <class T>
static void coop_init_one() {
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_coop_kernel<T>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, CoopCfg<T>::LDS_BYTES);
}

void potrf_coop_kernels_init() {
  coop_init_one<float>();
  coop_init_one<double>();
  coop_init_one<cfloat>();
  coop_init_one<cdouble>();
}

template void launch_potrf_coop<float>(float*, int, int, float*, int*, int, unsigned*, hipStream_t, bool, bool);
template void launch_potrf_coop<double>(double*, int, int, double*, int*, int, unsigned*, hipStream_t, bool, bool);
template void launch_potrf_coop<cfloat>(cfloat*, int, int, cfloat*, int*, int, unsigned*, hipStream_t, bool, bool);
template void launch_potrf_coop<cdouble>(cdouble*, int, int, cdouble*, int*, int, unsigned*, hipStream_t, bool, bool);

}  // namespace dlaf_mi355x
