// kernels_tridiag_dc.hip -- real symmetric tridiagonal eigensolver by divide & conquer, device resident
// (third stage of the eigensolver).
//
// Reference: TridiagSolver::call (include/dlaf/eigensolver/tridiag_solver/impl.h:198-262): Cuppen's decomposition
// (kernels.h:58-69), leaf problems by LAPACK stedc on the CPU (:107-140), then mergeSubproblems per node of the binary
// tree (merge.h:1077-1213): z from the adjacent rows of Q1 / Q2 (:191-208), rho doubled (:211-219), tolerance
// (:247-269), sort, deflation scan with Givens rotations (:696-760), the rank-one problem -- LAPACK laed4 per root,
// Loewner / Gu-Eisenstat weights, normalised eigenvectors (:798-971) -- on CPU threads, then Q . U as GEMMs on the
// GPU (:974-1075).  laed4 and stedc are third-party LAPACK routines; what is restated here is the published
// algorithm (LAWN 69; Gu & Eisenstat 1994/95).
//
// MI355X design: everything stays in HBM, one host synchronisation per LEVEL of the tree (the non-deflated counts that
// size the GEMMs).  Leaves are 64 x 64 (one wave each: implicit QL with the rotations applied to the wave's rows of
// Q in LDS), so the tree has more, smaller, batched merges instead of nb-sized CPU leaves.  Per level, for ALL its
// merges at once: one workgroup per merge prepares the rank-one problem (z, tolerance, merged order, the deflation
// scan -- the one serial part, on values gathered into sorted arrays first --, destination columns by column type);
// the secular equation is one WAVE per root (the lanes split the k poles of every evaluation), solved in the
// coordinate system of the nearer pole so that the differences d_i - lambda_j come out to full relative accuracy,
// by bisection on a logarithmic scale (no laed4-style rational interpolation: a root costs ~60 evaluations instead
// of ~6, which at n = 20480 is still only milliseconds of a GPU).
#include <cfloat>
#include <cstdio>
#include <cstdlib>

#include "device_api.hpp"
#include "tridiag_dc.hpp"

namespace dlaf_mi355x {

namespace {

constexpr int kDcThreads = 1024;

template <class R>
__device__ __forceinline__ R wave_sum_r(R v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    v += __shfl_xor(v, off);
  return v;
}
template <class R>
__device__ __forceinline__ R wave_max_r(R v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    v = fmax(v, __shfl_xor(v, off));
  return v;
}
template <class R>
__device__ __forceinline__ R r_eps() {
  if constexpr (sizeof(R) == 8)
    return R(DBL_EPSILON);
  else
    return R(FLT_EPSILON);
}
template <class R>
__device__ __forceinline__ R r_tiny() {
  if constexpr (sizeof(R) == 8)
    return R(DBL_MIN);
  else
    return R(FLT_MIN);
}

// ======================================================================================= leaves
// One wave per leaf (size <= 64): implicit QL with Wilkinson shifts (the algorithm of EISPACK tql2 / LAPACK xSTEQR's
// QL branch); lane r owns row r of the eigenvector matrix, kept in LDS column-major with an odd column stride.
template <class R>
__global__ __launch_bounds__(64) void dc_leaf_kernel(DcLeafArgs<R> p) {
  __shared__ R zq[64 * 65];
  __shared__ R ds[64], es[64];
  const int leaf = blockIdx.x;
  const int lane = threadIdx.x;
  const long off = p.leaf_off[leaf];
  const int n = p.leaf_n[leaf];
  if (lane < n) {
    ds[lane] = p.d[off + lane];
    es[lane] = lane + 1 < n ? p.e[off + lane] : R(0);
  }
  for (int c = 0; c < n; ++c)
    zq[c * 65 + lane] = (c == lane) ? R(1) : R(0);
  __syncthreads();
  int fail = 0;
  for (int l = 0; l < n; ++l) {
    int iter = 0;
    for (;;) {
      int m = l;
      for (; m < n - 1; ++m) {
        const R dd = fabs(ds[m]) + fabs(ds[m + 1]);
        if (fabs(es[m]) <= r_eps<R>() * dd)
          break;
      }
      if (m == l)
        break;
      if (++iter > 60) {
        fail = 1;
        break;
      }
      R g = (ds[l + 1] - ds[l]) / (R(2) * es[l]);
      R r = hypot(g, R(1));
      g = ds[m] - ds[l] + es[l] / (g + (g >= R(0) ? fabs(r) : -fabs(r)));
      R s = R(1), c = R(1), pp = R(0);
      int i = m - 1;
      bool under = false;
      for (; i >= l; --i) {
        R f = s * es[i];
        const R b = c * es[i];
        r = hypot(f, g);
        __syncthreads();
        if (lane == 0)
          es[i + 1] = r;
        if (r == R(0)) {
          if (lane == 0) {
            ds[i + 1] -= pp;
            es[m] = R(0);
          }
          under = true;
          break;
        }
        s = f / r;
        c = g / r;
        g = ds[i + 1] - pp;
        r = (ds[i] - g) * s + R(2) * c * b;
        pp = s * r;
        if (lane == 0)
          ds[i + 1] = g + pp;
        g = c * r - b;
        // columns i, i + 1 of the eigenvector matrix
        f = zq[(i + 1) * 65 + lane];
        const R zi = zq[i * 65 + lane];
        zq[(i + 1) * 65 + lane] = s * zi + c * f;
        zq[i * 65 + lane] = c * zi - s * f;
        __syncthreads();
      }
      __syncthreads();
      if (under)
        continue;
      if (lane == 0) {
        ds[l] -= pp;
        es[l] = g;
        es[m] = R(0);
      }
      __syncthreads();
    }
    if (fail)
      break;
  }
  __syncthreads();
  if (fail && lane == 0)
    atomicCAS(p.info, 0, 1 + (int) off);
  // eigenvalues in the order found; ascending order as an index (rank by counting)
  if (lane < n) {
    const R mine = ds[lane];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const R o = ds[j];
      rank += (o < mine || (o == mine && j < lane)) ? 1 : 0;
    }
    p.d[off + lane] = mine;
    p.ord[off + rank] = lane;
    for (int c = 0; c < n; ++c)
      p.q[(off + lane) + (off + c) * p.ldq] = zq[c * 65 + lane];
  }
}

// ======================================================================================= merge: preparation
// merge m acts on the slice [off, off + n) = child 1 (n1 entries) ++ child 2.  One workgroup per merge.
template <class R>
__global__ __launch_bounds__(kDcThreads) void dc_prepare_kernel(DcMergeArgs<R> p) {
  __shared__ R red[kDcThreads / 64];
  __shared__ R s_tol;
  __shared__ int s_cnt[4];
  const DcMerge mg = p.merges[blockIdx.x];
  const long off = mg.off;
  const int n = mg.n1 + mg.n2, n1 = mg.n1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  R* d = p.d + off;
  R* z = p.z + off;
  const int* ord = p.ord + off;
  int* srt = p.srt + off;
  R* dsrt = p.dsrt + off;
  R* zsrt = p.zsrt + off;
  int* ctype = p.ctype + off;
  // ---- z: last row of Q1 (negated when the off-diagonal element is negative), first row of Q2, over sqrt(2) ----
  const R rho_raw = p.rho[mg.split];
  const R sgn = rho_raw < R(0) ? R(-1) : R(1);
  const R rho = R(2) * fabs(rho_raw);
  const R isq2 = R(0.70710678118654752440);
  R zmax = R(0), dmax = R(0);
  for (int i = tid; i < n; i += kDcThreads) {
    const R v = i < n1 ? sgn * p.q[(off + n1 - 1) + (off + i) * p.ldq] : p.q[(off + n1) + (off + i) * p.ldq];
    const R zi = v * isq2;
    z[i] = zi;
    zmax = fmax(zmax, fabs(zi));
    dmax = fmax(dmax, fabs(d[i]));
  }
  R mx = wave_max_r(fmax(zmax, dmax));
  if (lane == 0)
    red[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    R t = R(0);
    for (int w = 0; w < kDcThreads / 64; ++w)
      t = fmax(t, red[w]);
    s_tol = R(8) * r_eps<R>() * t;  // calcTolerance, merge.h:247-269
  }
  __syncthreads();
  const R tol = s_tol;
  // ---- merged ascending order: rank = own position + elements of the other child that come first ----------------
  for (int i = tid; i < n; i += kDcThreads) {
    const bool first = i < n1;
    const int* mine = first ? ord : ord + n1;
    const int* other = first ? ord + n1 : ord;
    const int base_mine = first ? 0 : n1, base_other = first ? n1 : 0;
    const int len_other = first ? n - n1 : n1;
    const int pos = first ? i : i - n1;  // position in my child's ascending list
    const int sp = base_mine + mine[pos];
    const R val = d[sp];
    // elements of the other list strictly smaller (ties: child 1 first)
    int lo = 0, hi = len_other;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const R o = d[base_other + other[mid]];
      const bool before = first ? (o < val) : (o <= val);
      if (before)
        lo = mid + 1;
      else
        hi = mid;
    }
    const int rank = pos + lo;
    srt[rank] = sp;
    dsrt[rank] = val;
    zsrt[rank] = z[sp];
    ctype[rank] = sp < n1 ? 0 : 2;  // 0 upper, 1 dense, 2 lower, 3 deflated (in sorted order)
  }
  __syncthreads();
  // ---- deflation scan (applyDeflationToArrays, merge.h:696-760), one thread ---------------------------------------
  DcRot<R>* rots = p.rots + off;
  int* dfl = p.dfl + off;
  if (tid == 0) {
    int nrot = 0;
    int ndfl = 0;
    // deflated entries in ascending order of their (final) values: a rotation moves an eigenvalue inside its cluster,
    // so the order of deflation is ascending only up to the deflation tolerance -- insertion keeps the list sorted
    // (LAPACK xLAED2 does the same with INDXP), which the merge of the next level relies on
    auto push_deflated = [&](int idx, R val) {
      int q = ndfl++;
      while (q > 0 && dsrt[dfl[q - 1]] > val) {
        dfl[q] = dfl[q - 1];
        --q;
      }
      dfl[q] = idx;
    };
    int i1 = 0;
    R d1 = dsrt[0], z1 = zsrt[0];
    int c1 = ctype[0];
    for (int i2 = 1; i2 < n; ++i2) {
      if (fabs(rho * z1) <= tol) {
        // deflate i1, move on
        dsrt[i1] = d1;
        zsrt[i1] = z1;
        ctype[i1] = 3;
        push_deflated(i1, d1);
        i1 = i2;
        d1 = dsrt[i2];
        z1 = zsrt[i2];
        c1 = ctype[i2];
        continue;
      }
      const R d2 = dsrt[i2], z2 = zsrt[i2];
      const int c2 = ctype[i2];
      if (fabs(rho * z2) <= tol) {
        ctype[i2] = 3;
        push_deflated(i2, d2);
        continue;
      }
      const R r = hypot(z1, z2);
      const R c = z1 / r, s = z2 / r;
      if (fabs(c * s * (d2 - d1)) > tol) {
        dsrt[i1] = d1;
        zsrt[i1] = z1;
        ctype[i1] = c1;
        i1 = i2;
        d1 = d2;
        z1 = z2;
        c1 = c2;
        continue;
      }
      z1 = r;
      zsrt[i2] = R(0);
      const R nd2 = d1 * s * s + d2 * c * c;
      d1 = d1 * c * c + d2 * s * s;
      dsrt[i2] = nd2;
      rots[nrot].a = srt[i1];
      rots[nrot].b = srt[i2];
      rots[nrot].c = c;
      rots[nrot].s = s;
      ++nrot;
      if ((c1 == 0 && c2 == 2) || (c1 == 2 && c2 == 0))
        c1 = 1;
      ctype[i2] = 3;
      push_deflated(i2, nd2);
    }
    if (fabs(rho * z1) <= tol)
      c1 = 3;
    dsrt[i1] = d1;
    zsrt[i1] = z1;
    ctype[i1] = c1;
    if (c1 == 3)
      push_deflated(i1, d1);
    // counts per type
    int cu = 0, cd = 0, cl = 0;
    for (int i = 0; i < n; ++i) {
      const int t = ctype[i];
      cu += t == 0;
      cd += t == 1;
      cl += t == 2;
    }
    s_cnt[0] = cu;
    s_cnt[1] = cd;
    s_cnt[2] = cl;
    s_cnt[3] = nrot;
    DcHeader<R>& h = p.headers[blockIdx.x];
    h.k = cu + cd + cl;
    h.ku = cu;
    h.kd = cd;
    h.kl = cl;
    h.nrot = nrot;
    h.rho = rho;
  }
  __syncthreads();
  // ---- destination columns: [upper | dense | lower | deflated], each class in ascending order; the secular problem
  //      takes the non-deflated poles in ascending order ------------------------------------------------------------
  // (serial prefix by one wave's lanes over chunks: n <= a few 10^4)
  if (wave == 0) {
    int run[3] = {0, 0, 0};
    const int base[3] = {0, s_cnt[0], s_cnt[0] + s_cnt[1]};
    int sec = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const int t = i < n ? ctype[i] : -1;
      int within[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned long long mask = __ballot(t == q);
        within[q] = __popcll(mask & ((1ull << lane) - 1ull));
        if (t == q)
          p.tpos[off + srt[i]] = base[q] + run[q] + within[q];
        run[q] += __popcll(mask);
      }
      const unsigned long long nd = __ballot(t >= 0 && t != 3);
      if (t >= 0 && t != 3) {
        const int si = sec + __popcll(nd & ((1ull << lane) - 1ull));
        p.dsec[off + si] = dsrt[i];
        p.zsec[off + si] = zsrt[i];
      }
      sec += __popcll(nd);
    }
  }
  __syncthreads();
  // sec2t: secular index -> destination column (second pass: tpos is known per storage position)
  if (wave == 0) {
    int sec = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const int t = i < n ? ctype[i] : -1;
      const unsigned long long nd = __ballot(t >= 0 && t != 3);
      if (t >= 0 && t != 3)
        p.sec2t[off + sec + __popcll(nd & ((1ull << lane) - 1ull))] = p.tpos[off + srt[i]];
      sec += __popcll(nd);
    }
  }
  // deflated eigenvalues go to their destination slot (the new storage order is the destination order)
  __syncthreads();
  {
    const int k = s_cnt[0] + s_cnt[1] + s_cnt[2];
    for (int q = tid; q < n - k; q += kDcThreads) {
      const int i = dfl[q];
      p.tpos[off + srt[i]] = k + q;
      p.dnew[off + k + q] = dsrt[i];
    }
  }
}

// ======================================================================================= merge: rotations + gather
// grid.y = merge; threads over the rows of the merge's block
template <class R>
__global__ __launch_bounds__(256) void dc_rotate_kernel(DcMergeArgs<R> p) {
  const DcMerge mg = p.merges[blockIdx.y];
  const int n = mg.n1 + mg.n2;
  const int nrot = p.headers[blockIdx.y].nrot;
  if (nrot == 0)
    return;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n)
    return;
  const DcRot<R>* rots = p.rots + mg.off;
  R* q = p.q + (mg.off + r) + mg.off * p.ldq;
  for (int t = 0; t < nrot; ++t) {
    const DcRot<R> g = rots[t];
    const R x = q[(long) g.a * p.ldq], y = q[(long) g.b * p.ldq];
    q[(long) g.a * p.ldq] = g.c * x + g.s * y;
    q[(long) g.b * p.ldq] = g.c * y - g.s * x;
  }
}

// qt[:, tpos[c]] = q[:, c] over the rows of the block; grid (row chunks, columns, merges)
template <class R>
__global__ __launch_bounds__(256) void dc_gather_kernel(DcMergeArgs<R> p) {
  const DcMerge mg = p.merges[blockIdx.z];
  const int n = mg.n1 + mg.n2;
  for (int c = blockIdx.y; c < n; c += gridDim.y) {
    const int dst = p.tpos[mg.off + c];
    const R* src = p.q + mg.off + (mg.off + c) * p.ldq;
    R* out = p.qt + mg.off + (mg.off + dst) * p.ldq;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256)
      out[r] = src[r];
  }
}

// ======================================================================================= merge: secular equation
// f(tau) = 1 + rho sum_i z_i^2 / ((d_i - d_K) - tau), one wave per root; all lanes return the same value
template <class R>
__device__ __forceinline__ R secular_f(const R* __restrict__ dk, const R* __restrict__ z2, int k, R rho, R tau, int lane) {
  R acc = R(0);
  for (int i = lane; i < k; i += 64)
    acc += z2[i] / (dk[i] - tau);
  acc = wave_sum_r(acc);
  return R(1) + rho * acc;
}

// grid: (waves needed for the largest k, merges); 4 waves per workgroup.  The shifted poles d_i - d_K of a root live
// in the root's column of `dlt` while it is being solved, the final differences overwrite them.
template <class R>
__global__ __launch_bounds__(256) void dc_secular_kernel(DcMergeArgs<R> p) {
  const DcMerge mg = p.merges[blockIdx.y];
  const DcHeader<R> h = p.headers[blockIdx.y];
  const int k = h.k;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (j >= k)
    return;
  const R rho = h.rho;
  const R* dsec = p.dsec + mg.off;
  (void) p.zsec;
  R* col = p.dlt + mg.off + (mg.off + j) * p.ldq;  // k entries
  R* z2 = p.z2 + mg.off;                           // z_i^2, written by dc_z2_kernel
  R lo, hi;
  int K;
  if (j < k - 1) {
    const R dj = dsec[j], dj1 = dsec[j + 1];
    const R gap = dj1 - dj, half = gap * R(0.5);
    // f at the midpoint, in the coordinates of d_j
    for (int i = lane; i < k; i += 64)
      col[i] = dsec[i] - dj;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const R fm = secular_f(col, z2, k, rho, half, lane);
    if (fm > R(0)) {
      K = j;  // root in the left half: tau in (0, half]
      lo = R(0);
      hi = half;
    }
    else {
      K = j + 1;  // root in the right half: tau in [-half, 0)
      for (int i = lane; i < k; i += 64)
        col[i] = dsec[i] - dj1;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      lo = -half;
      hi = R(0);
    }
  }
  else {
    K = k - 1;
    const R dk1 = dsec[k - 1];
    R s = R(0);
    for (int i = lane; i < k; i += 64) {
      col[i] = dsec[i] - dk1;
      s += z2[i];
    }
    s = wave_sum_r(s);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    lo = R(0);
    hi = rho * s;
    if (!(hi > R(0)))
      hi = r_tiny<R>();
  }
  // bisection on |tau|, geometric while the bracket spans more than a factor of 4 (relative accuracy of a root that
  // hugs its pole), arithmetic afterwards.  a = the bracket end next to the pole (|a| < |b|).
  const bool pos = hi > R(0);  // tau > 0
  R a = pos ? lo : hi, b = pos ? hi : lo;  // |a| <= |b|, a == 0 at first
  R amag = fabs(b) * R(sizeof(R) == 8 ? 1e-290 : 1e-30);
  if (amag < r_tiny<R>())
    amag = r_tiny<R>();
  if (amag >= fabs(b))
    amag = fabs(b) * R(0.5);
  a = pos ? amag : -amag;
  {
    // f next to the pole must have the pole's sign; if not the root is closer than the floor: take the floor
    const R fa = secular_f(col, z2, k, rho, a, lane);
    const bool ok = pos ? (fa < R(0)) : (fa > R(0));
    if (!ok)
      b = a;
  }
  for (int it = 0; it < 200; ++it) {
    const R am = fabs(a), bm = fabs(b);
    if (!(bm > am))
      break;
    R mid;
    if (bm > R(4) * am)
      mid = sqrt(am) * sqrt(bm);
    else
      mid = am + (bm - am) * R(0.5);
    if (!(mid > am && mid < bm))
      break;
    const R t = pos ? mid : -mid;
    const R f = secular_f(col, z2, k, rho, t, lane);
    // f increases with tau.  pos: f(a) < 0 < f(b);  neg: a is the end near the pole (tau -> 0-: f -> +inf), f(b) < 0
    const bool toward_a = pos ? (f > R(0)) : (f < R(0));  // the root lies between a and t
    if (f == R(0)) {
      a = b = t;
      break;
    }
    if (toward_a)
      b = t;
    else
      a = t;
  }
  const R tau = (fabs(a) + (fabs(b) - fabs(a)) * R(0.5)) * (pos ? R(1) : R(-1));
  for (int i = lane; i < k; i += 64)
    col[i] = col[i] - tau;
  if (lane == 0)
    p.dnew[mg.off + j] = dsec[K] + tau;
}

template <class R>
__global__ __launch_bounds__(256) void dc_z2_kernel(DcMergeArgs<R> p) {
  const DcMerge mg = p.merges[blockIdx.y];
  const int k = p.headers[blockIdx.y].k;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < k) {
    const R z = p.zsec[mg.off + i];
    p.z2[mg.off + i] = z * z;
  }
}

// Loewner weights (merge.h:900-940; LAPACK xLAED3): zhat_i = sign(z_i) sqrt(-prod_j delta_i^(j) / prod_{j != i} (d_i - d_j))
template <class R>
__global__ __launch_bounds__(256) void dc_zhat_kernel(DcMergeArgs<R> p) {
  const DcMerge mg = p.merges[blockIdx.y];
  const int k = p.headers[blockIdx.y].k;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k)
    return;
  const R* dsec = p.dsec + mg.off;
  const R* dl = p.dlt + mg.off + i + mg.off * p.ldq;
  const R di = dsec[i];
  R w = dl[(long) i * p.ldq];
  for (int j = 0; j < k; ++j) {
    if (j != i)
      w *= dl[(long) j * p.ldq] / (di - dsec[j]);
  }
  const R zi = p.zsec[mg.off + i];
  const R v = sqrt(fabs(w));
  p.zhat[mg.off + i] = zi < R(0) ? -v : v;
}

// eigenvector j of the rank-one problem: u_i = zhat_i / delta_i^(j), normalised, row i stored at the destination
// column position of pole i (so that Q_new = Q_gathered U); one workgroup per column.  U is written transposed
// (ut[j + row * ld]): both factors of the product are then row-contiguous, 16-byte aligned operands of the general
// MFMA kernel whatever the class counts are
template <class R>
__global__ __launch_bounds__(256) void dc_evec_kernel(DcMergeArgs<R> p) {
  __shared__ R red[4];
  const DcMerge mg = p.merges[blockIdx.y];
  const int k = p.headers[blockIdx.y].k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = blockIdx.x; j < k; j += gridDim.x) {
    const R* dl = p.dlt + mg.off + (mg.off + j) * p.ldq;
    R* u = p.u + (mg.off + j) + mg.off * p.ldq;  // U is stored TRANSPOSED: u[j + row * ldq]
    R ss = R(0);
    for (int i = threadIdx.x; i < k; i += 256) {
      const R v = p.zhat[mg.off + i] / dl[i];
      ss += v * v;
    }
    ss = wave_sum_r(ss);
    __syncthreads();
    if (lane == 0)
      red[wave] = ss;
    __syncthreads();
    const R inv = R(1) / sqrt(red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < k; i += 256)
      u[(long) p.sec2t[mg.off + i] * p.ldq] = (p.zhat[mg.off + i] / dl[i]) * inv;
  }
}

// after the products: new eigenvalues (storage order = [roots ascending | deflated ascending]) and their merged order
template <class R>
__global__ __launch_bounds__(256) void dc_finish_kernel(DcMergeArgs<R> p) {
  const DcMerge mg = p.merges[blockIdx.y];
  const int n = mg.n1 + mg.n2;
  const int k = p.headers[blockIdx.y].k;
  const R* dn = p.dnew + mg.off;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const bool first = i < k;
    const R val = dn[i];
    const int base_other = first ? k : 0;
    const int len_other = first ? n - k : k;
    int lo = 0, hi = len_other;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const R o = dn[base_other + mid];
      const bool before = first ? (o < val) : (o <= val);
      if (before)
        lo = mid + 1;
      else
        hi = mid;
    }
    const int pos = first ? i : i - k;
    p.ord_out[mg.off + pos + lo] = i;
    p.d[mg.off + i] = val;
  }
}

// final: w[i] = d[ord[i]], z[:, i] = q[:, ord[i]]
template <class R>
__global__ __launch_bounds__(256) void dc_output_kernel(const R* q, long ldq, const R* d, const int* ord, long n, R* w,
                                                        R* z, long ldz) {
  for (long c = blockIdx.y; c < n; c += gridDim.y) {
    const long src = ord[c];
    if (blockIdx.x == 0 && threadIdx.x == 0)
      w[c] = d[src];
    for (long r = (long) blockIdx.x * 256 + threadIdx.x; r < n; r += (long) gridDim.x * 256)
      z[r + c * ldz] = q[r + src * ldq];
  }
}

template <class R>
__global__ __launch_bounds__(256) void dc_cuppen_kernel(R* d, const R* e, const long* bounds, R* rho, int nsplit) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nsplit)
    return;
  const long bnd = bounds[i];  // first index of the lower part
  const R off = e[bnd - 1];
  rho[i] = off;
  // (every boundary touches its own two diagonal entries: a leaf has at least two rows unless n is tiny, and two
  //  boundaries can share an entry only for one-row leaves -- atomics keep that case right)
  atomicAdd(&d[bnd - 1], -fabs(off));
  atomicAdd(&d[bnd], -fabs(off));
}

}  // namespace

template <class R>
void launch_dc_cuppen(R* d, const R* e, const long* bounds, R* rho, int nsplit, hipStream_t s) {
  if (nsplit <= 0)
    return;
  hipLaunchKernelGGL((dc_cuppen_kernel<R>), dim3((unsigned) ((nsplit + 255) / 256)), dim3(256), 0, s, d, e, bounds, rho,
                     nsplit);
}
template <class R>
void launch_dc_leaves(const DcLeafArgs<R>& a, int nleaves, hipStream_t s) {
  if (nleaves <= 0)
    return;
  hipLaunchKernelGGL((dc_leaf_kernel<R>), dim3((unsigned) nleaves), dim3(64), 0, s, a);
}
template <class R>
void launch_dc_prepare(const DcMergeArgs<R>& a, int nmerges, hipStream_t s) {
  hipLaunchKernelGGL((dc_prepare_kernel<R>), dim3((unsigned) nmerges), dim3(kDcThreads), 0, s, a);
}
template <class R>
void launch_dc_rotate_gather(const DcMergeArgs<R>& a, int nmerges, int nmax, hipStream_t s) {
  hipLaunchKernelGGL((dc_rotate_kernel<R>), dim3((unsigned) ((nmax + 255) / 256), (unsigned) nmerges), dim3(256), 0, s, a);
  const unsigned gx = (unsigned) std::min(8, (nmax + 255) / 256);
  const unsigned gy = (unsigned) std::min(nmax, 4096);
  hipLaunchKernelGGL((dc_gather_kernel<R>), dim3(gx, gy, (unsigned) nmerges), dim3(256), 0, s, a);
}
template <class R>
void launch_dc_secular(const DcMergeArgs<R>& a, int nmerges, int kmax, hipStream_t s) {
  if (kmax <= 0)
    return;
  hipLaunchKernelGGL((dc_z2_kernel<R>), dim3((unsigned) ((kmax + 255) / 256), (unsigned) nmerges), dim3(256), 0, s, a);
  hipLaunchKernelGGL((dc_secular_kernel<R>), dim3((unsigned) ((kmax + 3) / 4), (unsigned) nmerges), dim3(256), 0, s, a);
  hipLaunchKernelGGL((dc_zhat_kernel<R>), dim3((unsigned) ((kmax + 255) / 256), (unsigned) nmerges), dim3(256), 0, s, a);
  hipLaunchKernelGGL((dc_evec_kernel<R>), dim3((unsigned) std::min(kmax, 2048), (unsigned) nmerges), dim3(256), 0, s, a);
}
template <class R>
void launch_dc_finish(const DcMergeArgs<R>& a, int nmerges, int nmax, hipStream_t s) {
  hipLaunchKernelGGL((dc_finish_kernel<R>), dim3((unsigned) std::min(64, (nmax + 255) / 256), (unsigned) nmerges), dim3(256),
                     0, s, a);
}
template <class R>
void launch_dc_output(const R* q, long ldq, const R* d, const int* ord, long n, R* w, R* z, long ldz, hipStream_t s) {
  if (n <= 0)
    return;
  hipLaunchKernelGGL((dc_output_kernel<R>), dim3((unsigned) std::min<long>(8, (n + 255) / 256), (unsigned) std::min<long>(n, 8192)),
                     dim3(256), 0, s, q, ldq, d, ord, n, w, z, ldz);
}

#define INST(R)                                                                                        \
  template void launch_dc_cuppen<R>(R*, const R*, const long*, R*, int, hipStream_t);                  \
  template void launch_dc_leaves<R>(const DcLeafArgs<R>&, int, hipStream_t);                           \
  template void launch_dc_prepare<R>(const DcMergeArgs<R>&, int, hipStream_t);                         \
  template void launch_dc_rotate_gather<R>(const DcMergeArgs<R>&, int, int, hipStream_t);              \
  template void launch_dc_secular<R>(const DcMergeArgs<R>&, int, int, hipStream_t);                    \
  template void launch_dc_finish<R>(const DcMergeArgs<R>&, int, int, hipStream_t);                     \
  template void launch_dc_output<R>(const R*, long, const R*, const int*, long, R*, R*, long, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace dlaf_mi355x
