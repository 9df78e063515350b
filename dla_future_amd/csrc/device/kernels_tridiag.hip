// kernels_tridiag.hip -- gfx950 kernels of the second eigensolver stage: Hermitian band matrix -> real symmetric
// tridiagonal matrix by bulge chasing, and the helpers of the matching back-transformation (SURVEY.md section 8(f)
// item 4).
//
// Reference: eigensolver/band_to_tridiag/mc.h -- SweepWorker::start_sweep / do_step (:503-531) on the compact band
// copy BandBlock (:180-206), one CPU task per step, dependencies between consecutive sweeps through counting
// semaphores (:683-709, :760-773); the reference's GPU backend copies the band to the host and runs the same CPU
// code.  bt_band_to_tridiag/impl.h:139-175 (computeVT) for the well-formed reflector blocks.
//
// MI355X design: ONE launch.  Persistent workgroups draw sweeps from a counter (so a workgroup only ever waits for
// a sweep that some resident workgroup already owns: no co-residency assumption), sweep s runs step t once sweep
// s - 1 has published t + 2 finished steps (b2t_kernel) -- or, in the register kernel, once sweep s - 1 has finished
// step t and published the FIRST COLUMN of its step t + 1, the only part of that step this one reads (b2t_reg_kernel:
// the progress word counts 2 per step + 1 for that column).  A step touches 1.5 b^2 elements of the band copy, which stays in the
// memory-side cache (42 MB at n = 20480, b = 128); the band bytes are handed from workgroup to workgroup, possibly on
// another XCD, once per sweep: they are stored write-through and loaded sc1 (cdna_hip_programming.md, Guideline 16:
// "every load sc1"), the progress word of a sweep is one relaxed agent-scope atomic.  Inside a step a wave owns a
// column of the block (rows on lanes: 512-byte contiguous loads), the matrix-vector products of the two-sided update
// accumulate per wave in LDS and are summed once.
#include <cstdio>
#include <cstdlib>

#include "device_api.hpp"
#include "lane_ops.hpp"
#include "tridiag_api.hpp"

namespace dlaf_mi355x {

namespace {

constexpr int kB2tThreads = 512;
constexpr int kB2tWaves = kB2tThreads / 64;
constexpr int kB2tMaxBand = 256;
constexpr int kB2tRegs = kB2tMaxBand / 64;  // elements of a column part one lane holds
constexpr unsigned kB2tDone = 0x7fffffffu;
constexpr long kB2tSpinLimit = 40000000;

template <class T>
__device__ __forceinline__ T c_mul(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
  else
    return a * b;
}
// conj(a) * b
template <class T>
__device__ __forceinline__ T c_cmul(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re};
  else
    return a * b;
}
template <class T>
__device__ __forceinline__ T c_add(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re + b.re, a.im + b.im};
  else
    return a + b;
}
template <class T>
__device__ __forceinline__ T c_sub(const T& a, const T& b) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re - b.re, a.im - b.im};
  else
    return a - b;
}
template <class T>
__device__ __forceinline__ T c_conj(const T& a) {
  if constexpr (TypeInfo<T>::is_complex)
    return T{a.re, -a.im};
  else
    return a;
}
template <class T>
__device__ __forceinline__ T c_scale(const T& a, real_t<T> s) {
  return make_el<T>(re_of(a) * s, im_of(a) * s);
}
template <class T>
__device__ __forceinline__ real_t<T> c_abs2(const T& a) {
  return re_of(a) * re_of(a) + im_of(a) * im_of(a);
}

template <class T>
__device__ __forceinline__ T wave_sum_t(T v) {
  return wave_sum_fast(v);  // (lane_ops.hpp: DPP + permlane swaps, no trip through the LDS crossbar)
}

// write-through store / sc1 load of one element (relaxed agent-scope atomics on its 4- or 8-byte words)
template <class T>
__device__ __forceinline__ void st_wt(T* p, const T& v) {
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
template <class T>
__device__ __forceinline__ T ld_sc(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
  }
  else if constexpr (sizeof(T) == 8) {
    return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
  }
  else {
    struct Two {
      unsigned long long a, b;
    };
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    Two t;
    t.a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t.b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(T, t);
  }
}

// ======================================================================================= band copy
template <class T>
__global__ __launch_bounds__(kThreads) void band_extract_kernel(const T* tiles, long ltr, int nb, int pr, int ri, int pc,
                                                                int ci, long n, int b, T* band) {
  const int ldb = 2 * b;
  const long total = n * ldb;
  for (long e = (long) blockIdx.x * kThreads + threadIdx.x; e < total; e += (long) gridDim.x * kThreads) {
    const long c = e / ldb;
    const int o = (int) (e % ldb);
    const long r = c + o;
    T v = zero_el<T>();
    if (o <= b && r < n) {
      const long gi = r / nb, gj = c / nb;
      if (gi % pr == ri && gj % pc == ci) {
        const long il = gi / pr, jl = gj / pc;
        v = tiles[(il + jl * ltr) * (long) nb * nb + (r % nb) + (c % nb) * (long) nb];
        if (o == 0)
          v = make_el<T>(re_of(v), real_t<T>(0));
      }
    }
    band[e] = v;
  }
}

template <class T>
__global__ __launch_bounds__(kThreads) void tridiag_extract_kernel(const T* band, long n, int b, real_t<T>* d,
                                                                   real_t<T>* e) {
  const long i = (long) blockIdx.x * kThreads + threadIdx.x;
  if (i < n) {
    d[i] = re_of(band[i * 2 * b]);
    e[i] = (i + 1 < n) ? re_of(band[i * 2 * b + 1]) : real_t<T>(0);
  }
}

// ======================================================================================= bulge chasing
template <class T>
struct B2tArgs {
  T* band;
  long n;
  int b;
  T* vout;
  long ldv;
  unsigned* progress;  // one word per sweep: finished steps (kB2tDone when the sweep is over)
  unsigned* next;      // the sweep counter
  unsigned* failed;    // != 0: some workgroup gave up a wait
  int nsweeps;
  int* info;
  long spin_limit;
};

// LDS image of a workgroup (elements of T): v, w, zb, cs, vnew (b each), the per-wave partial products (waves x 2 b),
// then a few scalars
template <class T>
struct B2tLds {
  T* v;
  T* w;
  T* zb;
  T* cs;
  T* vnew;
  T* zw;
  T* sc;  // [0] tau, [1] tau of the new reflector, [2] alpha
  unsigned* slot;
  __device__ B2tLds(unsigned char* raw, int b) {
    T* p = reinterpret_cast<T*>(raw);
    v = p;
    w = v + b;
    zb = w + b;
    cs = zb + b;
    vnew = cs + b;
    zw = vnew + b;
    sc = zw + (size_t) kB2tWaves * 2 * b;
    slot = reinterpret_cast<unsigned*>(sc + 4);
  }
};
template <class T>
size_t b2t_lds_bytes(int b) {
  return ((size_t) 5 * b + (size_t) kB2tWaves * 2 * b + 4) * sizeof(T) + 16;
}

// xLARFG of x[0 .. len) held in LDS (one wave): on return x[0] = 1 and x[1 ..] = the reflector, *tau and *beta set.
// (HH_reflector, mc.h:55-68.)
template <class T>
__device__ __forceinline__ void wave_larfg(T* x, int len, int lane, T& tau, T& beta) {
  using R = real_t<T>;
  R ss = R(0);
  for (int i = 1 + lane; i < len; i += 64)
    ss += c_abs2(x[i]);
  ss = wave_sum_t(ss);
  const T alpha = x[0];
  const R ar = re_of(alpha), ai = im_of(alpha);
  if (ss == R(0) && ai == R(0)) {
    tau = zero_el<T>();
    beta = alpha;
  }
  else {
    const R nrm = sqrt(ar * ar + ai * ai + ss);
    const R bt = ar >= R(0) ? -nrm : nrm;
    tau = make_el<T>((bt - ar) / bt, -ai / bt);
    // 1 / (alpha - beta)
    const R dr = ar - bt, di = ai;
    const R dd = dr * dr + di * di;
    const T scale = make_el<T>(dr / dd, -di / dd);
    for (int i = 1 + lane; i < len; i += 64)
      x[i] = c_mul(x[i], scale);
    beta = make_el<T>(bt, R(0));
  }
  if (lane == 0)
    x[0] = make_el<T>(R(1), R(0));
}

template <class T>
__global__ __launch_bounds__(kB2tThreads) void b2t_kernel(B2tArgs<T> p) {
  using R = real_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char b2t_raw[];
  B2tLds<T> L(b2t_raw, p.b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = p.b, ldb = 2 * b;
  const long n = p.n;
  T* zw = L.zw + (size_t) wave * 2 * b;

  for (;;) {
    if (tid == 0)
      *L.slot = __hip_atomic_fetch_add(p.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const long s = (long) *L.slot;
    __syncthreads();
    if (s >= p.nsweeps)
      break;
    // all threads: wait until sweep s - 1 has finished `need` steps
    auto wait_prev = [&](unsigned need) -> bool {
      if (s == 0) {
        __syncthreads();
        return true;
      }
      if (tid == 0) {
        unsigned v;
        long spins = 0;
        while ((v = __hip_atomic_load(p.progress + (s - 1) * kB2tProgressStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > p.spin_limit ||
              ((spins & 255) == 0 && __hip_atomic_load(p.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            v = 0xFFFFFFFFu;
            break;
          }
        }
        *L.slot = v;
      }
      __syncthreads();
      const unsigned r = *L.slot;
      __syncthreads();
      // (every load of band bytes below is an sc1 load: no agent-scope acquire needed, only program order)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      return r != 0xFFFFFFFFu;
    };
    auto publish = [&](unsigned value) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(p.progress + s * kB2tProgressStride, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    bool ok = wait_prev(1u);
    // ---- start_sweep: the reflector that annihilates column s below the first sub-diagonal -----------------------
    const bool cx_last = TypeInfo<T>::is_complex && s == n - 2;
    const int nsteps = cx_last ? 1 : (int) ((n - s - 2 + b - 1) / b);
    if (ok) {
      const int nn = (int) (n - s - 1 < b ? n - s - 1 : b);
      T* col = p.band + s * ldb + 1;
      for (int i = tid; i < nn; i += kB2tThreads)
        L.v[i] = ld_sc(col + i);
      __syncthreads();
      if (wave == 0) {
        T tau, beta;
        wave_larfg(L.v, nn, lane, tau, beta);
        if (lane == 0)
          L.sc[0] = tau;
        for (int i = lane; i < nn; i += 64)
          st_wt(col + i, i == 0 ? beta : zero_el<T>());
      }
      __syncthreads();
    }
    for (int step = 0; step < nsteps && ok; ++step) {
      const long j = 1 + s + (long) step * b;
      const int nh = (int) (n - j < b ? n - j : b);
      const long mrem = n - b - j;
      const int m = (int) (mrem < 0 ? 0 : (mrem < b ? mrem : b));
      const T tau = L.sc[0];
      // compact copy of the reflector (compact_copy_to_tile, mc.h:492-497)
      {
        const long pos = (s / b + step) * (long) b;
        T* dst = p.vout + pos + s * p.ldv;
        for (int i = tid; i < nh; i += kB2tThreads)
          dst[i] = i == 0 ? tau : L.v[i];
      }
      for (int i = tid; i < kB2tWaves * 2 * b; i += kB2tThreads)
        L.zw[i] = zero_el<T>();
      ok = wait_prev((unsigned) step + 2u);
      if (!ok)
        break;
      const int rows = nh + m;
      // ---- P1: z = A v over the rows of both blocks, cs = strictly-lower(A)^H v over the diagonal block -----------
      for (int cc = wave; cc < nh; cc += kB2tWaves) {
        const T vc = L.v[cc];
        const T* col = p.band + (j + cc) * ldb;
        T part = zero_el<T>();
        for (int o = lane; o < rows - cc; o += 64) {
          T a = ld_sc(col + o);
          const int r = cc + o;
          if (o == 0)
            a = make_el<T>(re_of(a), R(0));
          zw[r] = c_add(zw[r], c_mul(a, vc));
          if (o > 0 && r < nh)
            part = c_add(part, c_cmul(a, L.v[r]));
        }
        part = wave_sum_t(part);
        if (lane == 0)
          L.cs[cc] = part;
      }
      __syncthreads();
      // ---- P2: w = tau (z + cs) - 1/2 tau (w^H v) v  (apply_HH_left_right_herm, mc.h:70-86);  zb = B v -----------
      for (int r = tid; r < rows; r += kB2tThreads) {
        T sum = zero_el<T>();
#pragma unroll
        for (int q = 0; q < kB2tWaves; ++q)
          sum = c_add(sum, L.zw[(size_t) q * 2 * b + r]);
        if (r < nh)
          L.w[r] = c_mul(tau, c_add(sum, L.cs[r]));
        else
          L.zb[r - nh] = sum;
      }
      __syncthreads();
      if (wave == 0) {
        T dot = zero_el<T>();
        for (int r = lane; r < nh; r += 64)
          dot = c_add(dot, c_cmul(L.w[r], L.v[r]));
        dot = wave_sum_t(dot);
        if (lane == 0)
          L.sc[2] = c_scale(c_mul(dot, tau), R(-0.5));
      }
      __syncthreads();
      {
        const T alpha = L.sc[2];
        for (int r = tid; r < nh; r += kB2tThreads)
          L.w[r] = c_add(L.w[r], c_mul(alpha, L.v[r]));
      }
      __syncthreads();
      // ---- P3: D -= w v^H + v w^H (lower part, real diagonal) -------------------------------------------------------
      for (int cc = wave; cc < nh; cc += kB2tWaves) {
        T* col = p.band + (j + cc) * ldb;
        const T vcc = c_conj(L.v[cc]), wcc = c_conj(L.w[cc]);
        for (int o = lane; o < nh - cc; o += 64) {
          const int r = cc + o;
          T a = ld_sc(col + o);
          if (o == 0)
            a = make_el<T>(re_of(a) - R(2) * re_of(c_mul(L.w[r], vcc)), R(0));
          else
            a = c_sub(a, c_add(c_mul(L.w[r], vcc), c_mul(L.v[r], wcc)));
          st_wt(col + o, a);
        }
      }
      // ---- P4: first column of B after B -= tau (B v) v^H (apply_HH_right), then its reflector ------------------------
      if (m > 0 && wave == 0) {
        T* col = p.band + j * ldb + nh;
        const T vcc = c_conj(L.v[0]);
        for (int r = lane; r < m; r += 64)
          L.vnew[r] = c_sub(ld_sc(col + r), c_mul(tau, c_mul(L.zb[r], vcc)));
        if (m > 1) {
          T tau2, beta;
          wave_larfg(L.vnew, m, lane, tau2, beta);
          if (lane == 0)
            L.sc[1] = tau2;
          for (int r = lane; r < m; r += 64)
            st_wt(col + r, r == 0 ? beta : zero_el<T>());
        }
        else if (lane == 0) {
          st_wt(col, L.vnew[0]);
        }
      }
      __syncthreads();
      // ---- P5/6: the other columns of B: right update, then (I - conj(tau2) v2 v2^H) from the left --------------------
      if (m > 0) {
        const T ctau2 = m > 1 ? c_conj(L.sc[1]) : zero_el<T>();
        for (int cc = 1 + wave; cc < nh; cc += kB2tWaves) {
          T* col = p.band + (j + cc) * ldb + (nh - cc);
          const T tv = c_mul(tau, c_conj(L.v[cc]));
          T a[kB2tRegs];
          T part = zero_el<T>();
#pragma unroll
          for (int q = 0; q < kB2tRegs; ++q) {
            const int r = lane + 64 * q;
            if (r < m) {
              a[q] = c_sub(ld_sc(col + r), c_mul(L.zb[r], tv));
              part = c_add(part, c_cmul(a[q], L.vnew[r]));
            }
          }
          if (m > 1) {
            part = wave_sum_t(part);  // w3 = (column)^H v2
            const T f = c_mul(ctau2, c_conj(part));
#pragma unroll
            for (int q = 0; q < kB2tRegs; ++q) {
              const int r = lane + 64 * q;
              if (r < m)
                a[q] = c_sub(a[q], c_mul(L.vnew[r], f));
            }
          }
#pragma unroll
          for (int q = 0; q < kB2tRegs; ++q) {
            const int r = lane + 64 * q;
            if (r < m)
              st_wt(col + r, a[q]);
          }
        }
      }
      __syncthreads();
      if (m > 1) {
        for (int r = tid; r < m; r += kB2tThreads)
          L.v[r] = L.vnew[r];
        if (tid == 0)
          L.sc[0] = L.sc[1];
      }
      publish((unsigned) step + 1u);
      __syncthreads();
    }
    if (!ok) {
      if (tid == 0) {
        __hip_atomic_store(p.failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicCAS(p.info, 0, kInfoSchedulingFailure);
      }
    }
    publish(kB2tDone);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same sweep-step algorithm with the block of a step held in REGISTERS (band sizes up to kB2tRegBand, element
// types of up to 8 bytes).  Thread (wave w, lane l) owns the elements (row l + 64 q, column cc = w + NW k), q < 4,
// k < CPW, of the block -- the same elements in every phase of the step, so a step loads each of them ONCE (all loads
// of the step issued together: one memory latency instead of one per phase and column) and stores it once.  The ROWS
// of a thread do not depend on the column: the entries of the vectors that multiply from the left (v, w, the new
// reflector) that a thread ever needs are four registers each, loaded once per phase, and z = A v accumulates in four
// registers per thread; per column there is one broadcast operand and one wave reduction.
// Loads are sc1 buffer loads, which the compiler pipelines like plain loads (relaxed atomic loads, the other sc1 form,
// it issues one at a time); an element outside the block is loaded from and stored to an out-of-range offset (zero /
// dropped by the descriptor's bounds check, no traffic), so a step is the same straight-line code for every element.
// Round 4 (profiles/r04_b2t_phases.txt): the first column of a step's block is stored and
// published as soon as it is final -- it is all the successor's step waits for --, the progress word is polled by wave 0
// with scalar loads and relayed through LDS, P5 has an instance for full blocks, the wave reductions run on permlane
// swaps and DPP (lane_ops.hpp).
constexpr int kB2tRegBand = 128;
constexpr int kB2tExt = 2 * kB2tRegBand;  // rows of a block

template <class T>
__device__ __forceinline__ T buf_load_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned soff) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, soff, 16));
  }
  else {
    static_assert(sizeof(T) == 8, "register kernel: 4- and 8-byte elements");
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, soff, 16);
    return __builtin_bit_cast(T, v);
  }
}
template <class T>
__device__ __forceinline__ void buf_store_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned soff, const T& v) {
  if constexpr (sizeof(T) == 4) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, byte_off, soff, 16);
  }
  else {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rsrc, byte_off, soff, 16);
  }
}

// Sums of 16 values per lane over the 64 lanes of a wave: at every stage a lane keeps half of its values and sends the
// other half to its partner (upper / lower half, odd / even row, lane ^ 8, the mirror lane of its group of 8), then two
// plain stages inside the quad.  On return lane l holds the total of value index
// ((l >> 5) & 1) * 8 + ((l >> 4) & 1) * 4 + ((l >> 3) & 1) * 2 + ((l >> 2) & 1).
template <class T>
__device__ __forceinline__ T wave_reduce4_tail(const T (&b4)[4], int lane) {
  T c2[2];
  const bool h3 = (lane & 8) != 0, h2 = (lane & 4) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const T mine = h3 ? b4[2 + i] : b4[i];
    const T theirs = h3 ? b4[i] : b4[2 + i];
    c2[i] = c_add(mine, dpp_t<kDppRor8>(theirs));
  }
  T r = c_add(h2 ? c2[1] : c2[0], dpp_t<kDppHalfMirror>(h2 ? c2[0] : c2[1]));
  r = c_add(r, dpp_t<kDppXor2>(r));
  r = c_add(r, dpp_t<kDppXor1>(r));
  return r;
}
template <class T>
__device__ __forceinline__ T wave_reduce16(const T (&v)[16], int lane) {
  T a[8], b4[4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    a[i] = swap_add<false>(v[i], v[8 + i]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    b4[i] = swap_add<true>(a[i], a[4 + i]);
  return wave_reduce4_tail(b4, lane);
}
// the same for 8 values: lane l ends with the total of index ((l >> 5) & 1) * 4 + ((l >> 4) & 1) * 2 + ((l >> 3) & 1)
template <class T>
__device__ __forceinline__ T wave_reduce8(const T (&v)[8], int lane) {
  T b4[4], c2[2];
  const bool h3 = (lane & 8) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    b4[i] = swap_add<false>(v[i], v[4 + i]);
#pragma unroll
  for (int i = 0; i < 2; ++i)
    c2[i] = swap_add<true>(b4[i], b4[2 + i]);
  T r = c_add(h3 ? c2[1] : c2[0], dpp_t<kDppRor8>(h3 ? c2[0] : c2[1]));
  r = c_add(r, dpp_t<kDppHalfMirror>(r));
  r = c_add(r, dpp_t<kDppXor2>(r));
  r = c_add(r, dpp_t<kDppXor1>(r));
  return r;
}
__device__ __forceinline__ int wave_reduce8_index(int lane) {
  return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
}
__device__ __forceinline__ int wave_reduce16_index(int lane) {
  return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

// EARLY (band == kB2tRegBand): the loads of a step are issued before the wait for the predecessor's first column of
// its next step, which is this step's last column (see the load section)
// DLAF_MI355X_B2T_PROF (compile-time, tools/build_b2t_prof.sh): thread 0 of workgroup 0 accumulates the shader clock
// between the marks of a step and prints the totals when the workgroup leaves
#ifdef DLAF_MI355X_B2T_PROF
#define B2T_MARK(i)                                                   \
  do {                                                                \
    if (tid == 0) {                                                   \
      const unsigned long long now_ = __builtin_readcyclecounter();   \
      Lprof[i] += now_ - Lprof[15];                                   \
      Lprof[15] = now_;                                               \
    }                                                                 \
  } while (0)
#else
#define B2T_MARK(i) \
  do {              \
  } while (0)
#endif
template <class T, int NT, bool EARLY>
__global__ __launch_bounds__(NT) void b2t_reg_kernel(B2tArgs<T> p) {
  using R = real_t<T>;
  constexpr int NW = NT / 64;
  constexpr int CPW = kB2tRegBand / NW;  // columns of a block per wave
  constexpr int QN = kB2tExt / 64;       // rows of a block per lane
  constexpr unsigned kOob = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) unsigned char b2t_raw[];
  const int b = p.b, ldb = 2 * b;
  T* Lv = reinterpret_cast<T*>(b2t_raw);  // [v ; 0]: the reflector over the rows of the block
  T* Lwd = Lv + kB2tExt;                  // NW x [w ; 0]: every wave's own copy of w over the diagonal block
  T* Lv2 = Lwd + (size_t) NW * kB2tExt;   // [0 ; first column of B]: the next reflector's input
  T* Lcs = Lv2 + kB2tExt;                 // per column: (strictly lower D)^H v
  T* Lzw = Lcs + kB2tRegBand;             // NW x kB2tExt partial products A v
  T* Lsc = Lzw + (size_t) NW * kB2tExt;   // [0] tau, [1] tau2, [2] alpha, [3] beta
  T* Lred = Lsc + 4;                      // NW x CPW: column totals of a wave's reductions
  unsigned* Lslot = reinterpret_cast<unsigned*>(Lred + NW * CPW);
  unsigned* Lfail = Lslot + 1;            // raised when wave 0 gave up waiting: all waves leave at the next barrier
  unsigned* Lseen = Lslot + 2;            // the predecessor's progress word as wave 0 last saw it
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long n = p.n;
#ifdef DLAF_MI355X_B2T_PROF
  __shared__ unsigned long long Lprof[16];
  if (tid < 16)
    Lprof[tid] = 0;
  __syncthreads();
  const unsigned long long prof_t0 = wall_clock64();
#endif
  if (tid == 0)
    *Lfail = 0u;  // (the first barrier of the sweep loop orders it)
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.band, 0, (int) ((size_t) (n + 2) * ldb * sizeof(T)), 0x00020000);

  for (;;) {
    if (tid == 0) {
      *Lslot = __hip_atomic_fetch_add(p.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *Lseen = 0u;
    }
    __syncthreads();
    const long s = (long) __builtin_amdgcn_readfirstlane((int) *Lslot);  // (scalar: everything derived from it stays in SGPRs)
    __syncthreads();
    if (s >= p.nsweeps)
      break;
#ifdef DLAF_MI355X_B2T_PROF
    if (tid == 0)
      Lprof[15] = __builtin_readcyclecounter();
#endif
    // progress[s] counts 2 per finished step, + 1 once the FIRST COLUMN of the step under way is final (see below).
    // Wave 0 polls it, with scalar loads past the scalar cache (invalidate + glc): counted by lgkmcnt, they do not queue
    // behind the block loads the wave has in flight (a vector load would return behind them).  What it sees goes into an
    // LDS word on which the other waves spin: no barrier, one poller per workgroup on the L2.  When wave 0 gives up it
    // raises Lfail and lets the others through; the waves leave together at the next barrier.
    unsigned seen = 0u;  // (per wave: the last value read)
    auto wave_wait = [&](unsigned need) {
      if (s == 0 || seen >= need)
        return;
      if (wave == 0) {
        const unsigned* prev = p.progress + (s - 1) * kB2tProgressStride;
        long spins = 0;
        unsigned r;
        for (;;) {
          asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(prev) : "memory");
          if (r >= need)
            break;
          __builtin_amdgcn_s_sleep(1);  // (polling without it: no difference, 396 against 397 ms)
          if (++spins > p.spin_limit ||
              ((spins & 255) == 0 && __hip_atomic_load(p.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            if (lane == 0)
              __hip_atomic_store(Lfail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            r = 0xFFFFFFFFu;
            break;
          }
        }
        seen = r;
        if (lane == 0)
          __hip_atomic_store(Lseen, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      else {
        unsigned r;
        while ((r = (unsigned) __builtin_amdgcn_readfirstlane(
                    (int) __hip_atomic_load(Lseen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) < need)
          __builtin_amdgcn_s_sleep(1);
        seen = r;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto gave_up = [&]() -> bool {  // (behind a barrier: the same answer in every wave)
      return __hip_atomic_load(Lfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
    };
    auto publish = [&](unsigned value) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(p.progress + s * kB2tProgressStride, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // the first reflector of the sweep: column s, the first column of the predecessor's first block
    wave_wait(1u);
    int bdiv = b;  // (opaque: the reciprocal of the division is formed per sweep instead of living in a register)
    asm volatile("" : "+s"(bdiv));
    const long sblk = (long) ((int) s / bdiv);
    const bool cx_last = TypeInfo<T>::is_complex && s == n - 2;
    const int nsteps = cx_last ? 1 : (int) ((n - s - 2 + b - 1) / b);
    bool ok;
    {
      const int nn = (int) (n - s - 1 < b ? n - s - 1 : b);
      T* col = p.band + s * ldb + 1;
      int td = tid;  // (opaque: the addresses derived from the thread index are formed here, not kept across the sweeps)
      asm volatile("" : "+v"(td));
      for (int i = td; i < kB2tExt; i += NT)
        Lv[i] = i < nn ? ld_sc(col + i) : zero_el<T>();
      __syncthreads();
      ok = !gave_up();
      if (ok && wave == 0) {
        const int l0 = td & 63;
        T tau, beta;
        wave_larfg(Lv, nn, l0, tau, beta);
        if (l0 == 0)
          Lsc[0] = tau;
        for (int i = l0; i < nn; i += 64)
          st_wt(col + i, i == 0 ? beta : zero_el<T>());
      }
      __syncthreads();
    }
    // The block of a step lives in A: element (r, cc) at band[(j + cc) ldb + r - cc].  Sweep s - 1 is done with all of
    // it but the last column when it has finished its step of the same number; the last column (cc = b - 1, rows from
    // b - 1 on) is the FIRST column of the predecessor's next block, which that step finishes early: its rows over the
    // diagonal block are final after the two-sided update, its rows below become (beta, 0, ..., 0) -- known as soon as
    // the next reflector is.  The predecessor stores that column and publishes it right there, a third of a step before
    // its other stores are out, and never touches it again.
    // An element outside the block (above the diagonal of D, beyond the matrix) is loaded from an out-of-range offset:
    // the descriptor's bounds check returns zero without touching memory, and the phases need no masks.  (The complex
    // instance, short of registers, is faster with the masks in P1: 329 against 361 ms at n = 12288.)
    constexpr bool kMaskInP1 = TypeInfo<T>::is_complex;
    T A[CPW][QN];
    auto load_columns = [&](unsigned base, int lnn, int wvv, int nhh, int rws, int k0, int k1) {
#pragma unroll
      for (int k = 0; k < CPW; ++k) {
        if (k >= k0 && k < k1) {
          const int cc = wvv + NW * k;
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int r = lnn + 64 * q;
            // (rows from kB2tRegBand on lie below every column's diagonal)
            const bool valid = kMaskInP1 || (cc < nhh && r < rws && (64 * q >= kB2tRegBand || r >= cc));
            A[k][q] = buf_load_sc1<T>(rsrc, valid ? base + (unsigned) (64 * q * (int) sizeof(T)) : kOob,
                                      (unsigned) (cc * (ldb - 1) * (int) sizeof(T)));
          }
        }
      }
    };
    for (int step = 0; step < nsteps && ok; ++step) {
      const long j = 1 + s + (long) step * b;
      const int nh = (int) (n - j < b ? n - j : b);
      const long mrem = n - b - j;
      const int m = (int) (mrem < 0 ? 0 : (mrem < b ? mrem : b));
      const int rows = nh + m;
      B2T_MARK(0);  // (sweep start / the tail of the previous step)
      // (the lane index is made opaque per step: otherwise the per-element masks and offsets are hoisted out of the
      //  step loop and kept alive across it)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      int wv = wave;  // (and the wave index, with everything derived from it: column numbers, their offsets, LDS rows)
      asm volatile("" : "+s"(wv));
      const T tau = Lsc[0];  // (the barrier of the previous publish orders wave 0's hand-over of reflector and tau)
      {
        T* dst = p.vout + (sblk + step) * (long) b + s * p.ldv;
        const int i = ln + 64 * wv;
        if (i < nh)
          dst[i] = i == 0 ? tau : Lv[i];
      }
      const unsigned base_off = (unsigned) ((j * ldb + ln) * (long) sizeof(T));
      // ---- the block.  EARLY: all columns but the last are loaded as soon as the predecessor has finished this step
      //      (as a rule long ago), their latency hides behind the wait for its first column of the next
      wave_wait(2u * (unsigned) step + 2u);
      if constexpr (EARLY)
        load_columns(base_off, ln, wv, nh, rows, 0, CPW - 1);
      B2T_MARK(1);
      wave_wait(2u * (unsigned) step + 3u);
      B2T_MARK(2);  // wait for the predecessor
      // (EARLY: column b - 1 = wave NW - 1, k = CPW - 1; the other waves read a column that was final already)
      load_columns(base_off, ln, wv, nh, rows, EARLY ? CPW - 1 : 0, CPW);
      T vr[QN], zacc[QN];
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        vr[q] = Lv[ln + 64 * q];
        zacc[q] = zero_el<T>();
      }
      // FULL: the block is 2 b x b with b = kB2tRegBand (every step but the last two of a sweep): the rows of a lane's
      // upper two registers lie in D, those of the lower two in B.  In every case the rows from kB2tRegBand on lie
      // below the reflector: vr is zero there, and so is v2r above row nh.
      const bool full = nh == kB2tRegBand && m == kB2tRegBand;
      // ---- P1: z = A v over the rows of both blocks, cs = strictly-lower(D)^H v -----------------------------------------
      static_assert(CPW == 16, "wave_reduce8 x 2 / wave_reduce16");
      {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          T parts[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const int k = half * 8 + kk;
            const int cc = wv + NW * k;
            const T vc = Lv[cc];  // (zero beyond the reflector)
            T part = zero_el<T>();
#pragma unroll
            for (int q = 0; q < QN; ++q) {
              const int r = ln + 64 * q;
              T a = A[k][q];  // (zero outside the block: load_columns)
              if constexpr (kMaskInP1) {
                const bool valid = cc < nh && r < rows && (64 * q >= kB2tRegBand || r >= cc);
                a = valid ? a : zero_el<T>();
                A[k][q] = a;
              }
              if (64 * q < kB2tRegBand) {
                if constexpr (TypeInfo<T>::is_complex) {
                  if (r == cc) {
                    a = make_el<T>(re_of(a), R(0));  // the diagonal element
                    A[k][q] = a;
                  }
                }
                part = c_add(part, r == cc ? zero_el<T>() : c_cmul(a, vr[q]));
              }
              zacc[q] = c_add(zacc[q], c_mul(a, vc));
            }
            parts[kk] = part;
          }
          const T tot = wave_reduce8(parts, ln);
          if ((ln & 7) == 0)
            Lcs[wv + NW * (half * 8 + wave_reduce8_index(ln))] = tot;
        }
      }
#pragma unroll
      for (int q = 0; q < QN; ++q)
        Lzw[(size_t) wv * kB2tExt + ln + 64 * q] = zacc[q];
      B2T_MARK(3);  // last column + P1
      __syncthreads();
      if (gave_up()) {
        ok = false;
        break;
      }
      B2T_MARK(4);  // barrier
      // ---- P2 (every wave for itself: no workgroup barrier): wx = [tau (z + cs) - 1/2 tau (w^H v) v ; tau B v ; 0] over
      //      the wave's own four rows per lane; w over the diagonal block also goes to the wave's LDS copy, from which the
      //      column operands conj(w_c) are broadcast
      T wr[QN];
      {
        T dotp = zero_el<T>();
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int r = ln + 64 * q;
          T sum = zero_el<T>();
#pragma unroll
          for (int u = 0; u < NW; ++u)
            sum = c_add(sum, Lzw[(size_t) u * kB2tExt + r]);
          T w = zero_el<T>();
          if (r < nh)
            w = c_mul(tau, c_add(sum, Lcs[r]));
          else if (r < rows)
            w = c_mul(tau, sum);
          wr[q] = w;
          dotp = c_add(dotp, c_cmul(w, vr[q]));  // (vr is zero beyond the diagonal block)
        }
        const T dot = wave_sum_fast(dotp);
        const T alpha = c_scale(c_mul(dot, tau), R(-0.5));
        T* wd = Lwd + (size_t) wv * kB2tExt;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int r = ln + 64 * q;
          wr[q] = c_add(wr[q], c_mul(alpha, vr[q]));
          wd[r] = r < nh ? wr[q] : zero_el<T>();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- P3: A -= wx conj(v_c) + vx conj(w_c)  (two-sided update of D, right update of B) ---------------------------
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const int cc = wv + NW * k;
          const T vcc = c_conj(Lv[cc]);
          const T wcc = c_conj(wd[cc]);
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int r = ln + 64 * q;
            // (columns beyond the block have vcc = wcc = 0, rows beyond it wr = vr = 0)
            if (64 * q < kB2tRegBand) {
              const T upd = c_add(c_mul(wr[q], vcc), c_mul(vr[q], wcc));
              A[k][q] = c_sub(A[k][q], r >= cc ? upd : zero_el<T>());
            }
            else
              A[k][q] = c_sub(A[k][q], c_mul(wr[q], vcc));  // (below the reflector)
          }
        }
      }
      // the first column of B is the next reflector's input (wave 0 holds column 0)
      if (wv == 0) {
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int r = ln + 64 * q;
          Lv2[r] = (r >= nh && r < rows) ? A[0][q] : zero_el<T>();
        }
      }
      B2T_MARK(5);  // P2 + P3
      __syncthreads();
      B2T_MARK(6);  // barrier
      // ---- P4 (every wave for itself): the reflector of the first column of B, xLARFG on the wave's registers ---------
      T v2r[QN];
      T tau2 = zero_el<T>(), beta = zero_el<T>();
      {
        R ss = R(0);
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int r = ln + 64 * q;
          v2r[q] = Lv2[r];
          if (r > nh)
            ss += c_abs2(v2r[q]);  // (zero outside the rows of B)
        }
        ss = wave_sum_fast(ss);
        const T alpha0 = Lv2[nh];  // (m == 0: row nh holds zero)
        beta = alpha0;
        const R ar = re_of(alpha0), ai = im_of(alpha0);
        if (m > 1 && !(ss == R(0) && ai == R(0))) {
          const R nrm = sqrt(ar * ar + ai * ai + ss);
          const R bt = ar >= R(0) ? -nrm : nrm;
          tau2 = make_el<T>((bt - ar) / bt, -ai / bt);
          const R dr = ar - bt, di = ai;
          const R dd = dr * dr + di * di;
          const T scale = make_el<T>(dr / dd, -di / dd);
#pragma unroll
          for (int q = 0; q < QN; ++q)
            v2r[q] = c_mul(v2r[q], scale);
          beta = make_el<T>(bt, R(0));
        }
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int r = ln + 64 * q;
          if (r == nh)
            v2r[q] = m > 1 ? make_el<T>(R(1), R(0)) : zero_el<T>();  // (m <= 1: no reflector, nothing to apply)
        }
      }
      // ---- the first column of the block is final: rows of D since P3, rows of B = (beta, 0, ..., 0).  Wave 0 holds it,
      //      stores it and publishes it; the successor's step of the same number may start on it (m == 1: the element
      //      itself, beta holds it)
      asm volatile("" : "+v"(ln));
      const unsigned base_st = (unsigned) ((j * ldb + ln) * (long) sizeof(T));
      if (wv == 0) {
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int r = ln + 64 * q;
          T v = A[0][q];
          if (r >= nh)
            v = r == nh ? beta : zero_el<T>();
          buf_store_sc1<T>(rsrc, r < rows ? base_st + (unsigned) (64 * q * (int) sizeof(T)) : kOob, 0u, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ln == 0)
          __hip_atomic_store(p.progress + s * kB2tProgressStride, 2u * (unsigned) step + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      B2T_MARK(7);  // P4, first column out
      // ---- P5/6: A -= v2x (conj(tau2) conj(A_c^H v2)) on the columns of B but the first, then the stores of all columns
      //      but the first (which the successor may be changing by now).  FULL: v2 lives on the lower two registers, the
      //      rows of D (final since P3) go out first, beside the reduction
      auto p5 = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        constexpr int Q0 = FULL ? kB2tRegBand / 64 : 0;
        const T ctau2 = c_conj(tau2);
        if constexpr (FULL) {
#pragma unroll
          for (int k = 0; k < CPW; ++k) {
            const int cc = wv + NW * k;
#pragma unroll
            for (int q = 0; q < Q0; ++q) {
              const int r = ln + 64 * q;
              buf_store_sc1<T>(rsrc, (r >= cc && cc != 0) ? base_st + (unsigned) (64 * q * (int) sizeof(T)) : kOob,
                               (unsigned) (cc * (ldb - 1) * (int) sizeof(T)), A[k][q]);
            }
          }
        }
        {
          T parts[CPW];
#pragma unroll
          for (int k = 0; k < CPW; ++k) {
            T part = zero_el<T>();
#pragma unroll
            for (int q = Q0; q < QN; ++q)
              part = c_add(part, c_cmul(A[k][q], v2r[q]));
            parts[k] = part;
          }
          const T tot = wave_reduce16(parts, ln);
          if ((ln & 3) == 0)
            Lred[wv * CPW + wave_reduce16_index(ln)] = tot;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const int cc = wv + NW * k;
          const T f = c_mul(ctau2, c_conj(Lred[wv * CPW + k]));
#pragma unroll
          for (int q = Q0; q < QN; ++q) {
            const int r = ln + 64 * q;
            const T v = c_sub(A[k][q], c_mul(v2r[q], f));
            const bool valid = FULL ? cc != 0 : (cc < nh && r >= cc && r < rows && cc != 0);
            buf_store_sc1<T>(rsrc, valid ? base_st + (unsigned) (64 * q * (int) sizeof(T)) : kOob,
                             (unsigned) (cc * (ldb - 1) * (int) sizeof(T)), v);
          }
        }
      };
      if constexpr (TypeInfo<T>::is_complex)
        p5(std::false_type{});  // (two instances do not fit the registers of the complex kernel)
      else if (full)
        p5(std::true_type{});
      else
        p5(std::false_type{});
      // the reflector of the next step (rows of B -> rows 0 .. m), written by wave 0; the barrier of publish() orders it
      if (m > 1) {
        if (wv == 0) {
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int r = ln + 64 * q;
            if (r >= nh && r < rows)
              Lv[r - nh] = v2r[q];
            if (r >= m && r < kB2tExt && (r < nh || r >= rows))
              Lv[r] = make_el<T>(R(ln >> 30), R(0));  // (zero, formed here: a constant would be kept -- spilled -- across the step)
          }
          if (ln == 0)
            Lsc[0] = tau2;
        }
      }
      B2T_MARK(8);  // P5, stores issued
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      B2T_MARK(9);  // stores acknowledged
      publish(2u * (unsigned) step + 2u);
      B2T_MARK(10);  // publish
#ifdef DLAF_MI355X_B2T_PROF
      if (tid == 0)
        Lprof[14] += 1;
#endif
    }
    if (!ok) {
      if (tid == 0) {
        __hip_atomic_store(p.failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicCAS(p.info, 0, kInfoSchedulingFailure);
      }
    }
    // A sweep is done only once its predecessor is.  "Done" answers every wait of the successor, also those for steps
    // this sweep never had -- and this sweep's last step waited for the FIRST COLUMN of the predecessor's next step only:
    // without this wait the predecessor could still be storing the rest of that step (its second column is the last
    // column of a block of sweep s + 1) when sweep s + 1 is let through.  Found by enumeration of the protocol on the CPU
    // (tests/test_b2t_handoff_model.py); a short wait at the end of a sweep, off every critical path.
    wave_wait(kB2tDone);
    publish(kB2tDone);
    __syncthreads();
  }
#ifdef DLAF_MI355X_B2T_PROF
  if (tid == 0 && blockIdx.x == 0) {
    const unsigned long long wall = wall_clock64() - prof_t0;  // 100 MHz
    printf("[b2t phases] steps %llu wall %.3f ms; shader clocks per step:", Lprof[14], wall * 1e-5);
    for (int i = 0; i <= 10; ++i)
      printf(" [%d] %.0f", i, (double) Lprof[i] / (double) (Lprof[14] ? Lprof[14] : 1));
    printf("\n");
  }
#endif
}
#undef B2T_MARK
template <class T, int NT>
size_t b2t_reg_lds_bytes() {
  return ((size_t) 2 * kB2tExt + kB2tRegBand + (size_t) 2 * (NT / 64) * kB2tExt + 8 + kB2tRegBand) * sizeof(T) + 16;  // (+ Lslot, Lfail, Lseen)
}

// ======================================================================================= reflector blocks
// one workgroup per block (ib, jb): well-formed 2 b x b image + taus
template <class T>
__global__ __launch_bounds__(kThreads) void b2t_expand_kernel(const T* vout, long ldv, long n, int b, int nblk, T* vx,
                                                              T* taus, int transposed) {
  using R = real_t<T>;
  const int ib = blockIdx.x, jb = blockIdx.y;
  const long blk = (long) jb * nblk + ib;
  T* out = vx + blk * 2 * (long) b * b;
  T* tout = taus + blk * b;
  if (ib < jb) {
    return;  // (never read)
  }
  const int st = ib - jb;
  const long nsweeps = TypeInfo<T>::is_complex ? n - 1 : n - 2;
  for (int e = threadIdx.x; e < 2 * b * b; e += kThreads) {
    const int r = e % (2 * b), k = e / (2 * b);
    const long sw = (long) jb * b + k;
    T v = zero_el<T>();
    if (sw < nsweeps) {
      const bool cx_last = TypeInfo<T>::is_complex && sw == n - 2;
      const long nsteps = cx_last ? 1 : (n - sw - 2 + b - 1) / b;
      if (st < nsteps) {
        const long first = 1 + sw + (long) st * b;
        const long len = n - first < b ? n - first : b;
        const int i = r - k;
        if (i == 0)
          v = make_el<T>(R(1), R(0));
        else if (i > 0 && i < len)
          v = vout[(long) ib * b + i + sw * ldv];
      }
    }
    // transposed: the image of V^T (element (r, k) at k + r b), what the fused back-transformation streams
    out[transposed ? k + (long) r * b : e] = v;
  }
  for (int k = threadIdx.x; k < b; k += kThreads) {
    const long sw = (long) jb * b + k;
    T t = zero_el<T>();
    if (sw < nsweeps) {
      const bool cx_last = TypeInfo<T>::is_complex && sw == n - 2;
      const long nsteps = cx_last ? 1 : (n - sw - 2 + b - 1) / b;
      if (st < nsteps)
        t = vout[(long) ib * b + sw * ldv];
    }
    tout[k] = t;
  }
}

// e[r + cl * lde] = z[r + gc * ldz] for the local columns cl of a block-cyclic column axis (gc its global column),
// real -> T on the way (castToComplex, tridiag_solver/impl.h:264-277)
template <class R, class T>
__global__ __launch_bounds__(kThreads) void cols_gather_cast_kernel(const R* z, long ldz, long n, int nb, int pc, int ci,
                                                                    long ncols_loc, T* e, long lde) {
  for (long cl = blockIdx.y; cl < ncols_loc; cl += gridDim.y) {
    const long gc = ((cl / nb) * pc + ci) * nb + cl % nb;
    for (long r = (long) blockIdx.x * kThreads + threadIdx.x; r < n; r += (long) gridDim.x * kThreads)
      e[r + cl * lde] = make_el<T>((real_t<T>) z[r + gc * ldz], real_t<T>(0));
  }
}

// tile (il, jl) of a tile-layout matrix <- rows gi nb .. of e (all rows of the local columns): gi = il * pr + ri
template <class T>
__global__ __launch_bounds__(kThreads) void rows_to_tiles_kernel(const T* e, long lde, long n, long ncols_loc, int nb,
                                                                 int pr, int ri, long ltr, T* tiles) {
  const long il = blockIdx.y, jl = blockIdx.z;
  const long gi = il * pr + ri;
  const long r0 = gi * nb;
  const int rows = (int) (n - r0 < nb ? n - r0 : nb);
  const int cols = (int) (ncols_loc - jl * nb < nb ? ncols_loc - jl * nb : nb);
  T* t = tiles + (il + jl * ltr) * (long) nb * nb;
  for (int idx = blockIdx.x * kThreads + threadIdx.x; idx < rows * cols; idx += gridDim.x * kThreads) {
    const int r = idx % rows, c = idx / rows;
    t[r + (long) c * nb] = e[(r0 + r) + (jl * nb + c) * lde];
  }
}

}  // namespace

template <class R, class T>
void launch_cols_gather_cast(const R* z, long ldz, long n, int nb, int pc, int ci, long ncols_loc, T* e, long lde,
                             hipStream_t stream) {
  if (n <= 0 || ncols_loc <= 0)
    return;
  hipLaunchKernelGGL((cols_gather_cast_kernel<R, T>), dim3((unsigned) std::min<long>(8, (n + kThreads - 1) / kThreads),
                                                           (unsigned) std::min<long>(ncols_loc, 8192)),
                     dim3(kThreads), 0, stream, z, ldz, n, nb, pc, ci, ncols_loc, e, lde);
}
template <class T>
void launch_rows_to_tiles(const T* e, long lde, long n, long ncols_loc, int nb, int pr, int ri, long ltr, long ltc, T* tiles,
                          hipStream_t stream) {
  if (ltr <= 0 || ltc <= 0)
    return;
  hipLaunchKernelGGL((rows_to_tiles_kernel<T>), dim3(8, (unsigned) ltr, (unsigned) ltc), dim3(kThreads), 0, stream, e, lde, n,
                     ncols_loc, nb, pr, ri, ltr, tiles);
}
template void launch_cols_gather_cast<float, float>(const float*, long, long, int, int, int, long, float*, long, hipStream_t);
template void launch_cols_gather_cast<double, double>(const double*, long, long, int, int, int, long, double*, long, hipStream_t);
template void launch_cols_gather_cast<float, cfloat>(const float*, long, long, int, int, int, long, cfloat*, long, hipStream_t);
template void launch_cols_gather_cast<double, cdouble>(const double*, long, long, int, int, int, long, cdouble*, long, hipStream_t);

int b2t_max_band() {
  return kB2tMaxBand;
}

template <class T>
void launch_band_extract(const T* tiles, long ltr, int nb, int pr, int ri, int pc, int ci, long n, int b, T* band,
                         hipStream_t stream) {
  const long total = n * 2 * b;
  if (total <= 0)
    return;
  const unsigned g = (unsigned) std::min<long>((total + kThreads - 1) / kThreads, 4096);
  hipLaunchKernelGGL((band_extract_kernel<T>), dim3(g), dim3(kThreads), 0, stream, tiles, ltr, nb, pr, ri, pc, ci, n, b, band);
}

template <class T>
void launch_tridiag_extract(const T* band, long n, int b, real_t<T>* d, real_t<T>* e, hipStream_t stream) {
  if (n <= 0)
    return;
  hipLaunchKernelGGL((tridiag_extract_kernel<T>), dim3((unsigned) ((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     stream, band, n, b, d, e);
}

template <class T>
void launch_band_to_tridiag(T* band, long n, int b, T* vout, long ldv, unsigned* sync, int* info, hipStream_t stream) {
  const long nsweeps = TypeInfo<T>::is_complex ? n - 1 : n - 2;
  if (nsweeps <= 0)
    return;
  if (b > kB2tMaxBand || b < 1) {
    fprintf(stderr, "[dlaf_mi355x] band_to_tridiagonal: band size %d outside the supported 1 .. %d\n", b, kB2tMaxBand);
    abort();
  }
  static const long spin_limit = [] {
    const char* e = std::getenv("DLAF_MI355X_B2T_SPIN_LIMIT");
    return e ? std::atol(e) : kB2tSpinLimit;
  }();
  static const int max_wg = [] {
    const char* e = std::getenv("DLAF_MI355X_B2T_WORKGROUPS");
    return e ? std::max(1, std::atoi(e)) : 256;
  }();
  (void) hipMemsetAsync(sync, 0, b2t_sync_words(n) * sizeof(unsigned), stream);
  B2tArgs<T> a;
  a.band = band;
  a.n = n;
  a.b = b;
  a.vout = vout;
  a.ldv = ldv;
  a.progress = sync + 64;
  a.next = sync;
  a.failed = sync + 16;
  a.nsweeps = (int) nsweeps;
  a.info = info;
  a.spin_limit = spin_limit;
  // sweeps that can be in flight together: one every two steps of the longest sweep
  const long useful = std::max<long>(1, (n / b + 2) / 2 + 1);
  const unsigned g = (unsigned) std::min<long>(std::min<long>(max_wg, nsweeps), useful);
  const size_t lds = b2t_lds_bytes<T>(b);
  static bool attr_set[4] = {false, false, false, false};
  const int ti = TypeInfo<T>::is_complex ? (sizeof(T) == 16 ? 3 : 2) : (sizeof(T) == 8 ? 1 : 0);
  if (!attr_set[ti]) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&b2t_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int) b2t_lds_bytes<T>(kB2tMaxBand));
    attr_set[ti] = true;
  }
  static const bool use_reg = [] {
    const char* e = std::getenv("DLAF_MI355X_B2T_KERNEL");  // "generic": the kernel that re-reads the block per phase
    return !(e && e[0] == 'g');
  }();
  if constexpr (sizeof(T) <= 8) {
    if (use_reg && b <= kB2tRegBand) {
      // 64 elements of the block per thread: 128 registers for 8-byte types (complex double keeps the generic kernel)
      constexpr int NT = 512;
      const size_t reg_lds = b2t_reg_lds_bytes<T, NT>();
      if (b == kB2tRegBand)
        hipLaunchKernelGGL((b2t_reg_kernel<T, NT, true>), dim3(g), dim3(NT), reg_lds, stream, a);
      else
        hipLaunchKernelGGL((b2t_reg_kernel<T, NT, false>), dim3(g), dim3(NT), reg_lds, stream, a);
      return;
    }
  }
  hipLaunchKernelGGL((b2t_kernel<T>), dim3(g), dim3(kB2tThreads), lds, stream, a);
}

template <class T>
void launch_b2t_expand(const T* vout, long ldv, long n, int b, T* vx, T* taus, hipStream_t stream, bool transposed) {
  const int nblk = (int) ((n + b - 1) / b);
  if (nblk <= 0)
    return;
  hipLaunchKernelGGL((b2t_expand_kernel<T>), dim3((unsigned) nblk, (unsigned) nblk), dim3(kThreads), 0, stream, vout, ldv, n,
                     b, nblk, vx, taus, transposed ? 1 : 0);
}

#define INST(T)                                                                                                      \
  template void launch_band_extract<T>(const T*, long, int, int, int, int, int, long, int, T*, hipStream_t);        \
  template void launch_tridiag_extract<T>(const T*, long, int, real_t<T>*, real_t<T>*, hipStream_t);                \
  template void launch_band_to_tridiag<T>(T*, long, int, T*, long, unsigned*, int*, hipStream_t);                   \
  template void launch_b2t_expand<T>(const T*, long, long, int, T*, T*, hipStream_t, bool);                                \
  template void launch_rows_to_tiles<T>(const T*, long, long, long, int, int, int, long, long, T*, hipStream_t);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
