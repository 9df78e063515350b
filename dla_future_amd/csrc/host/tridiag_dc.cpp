// tridiag_dc.cpp -- host driver of the divide & conquer tridiagonal eigensolver (kernels_tridiag_dc.hip).
//
// Reference: TridiagSolver::call (include/dlaf/eigensolver/tridiag_solver/impl.h:198-262): cuppensDecomposition,
// solveLeaf, then mergeSubproblems over generateSubproblemIndices (:44-75: the interval of leaves is split in the middle,
// ceilDiv, recursively), applyIndex / permute at the end.  The reference walks the tree depth first with one task graph
// per merge; here the merges of one HEIGHT of the tree run together (batched launches), the host learns the
// non-deflated counts of a whole level with one small copy and then issues that level's products.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../device/band_api.hpp"
#include "../device/tridiag_dc.hpp"
#include "eigensolver.hpp"

namespace dlaf_mi355x {

namespace {
template <class T>
T* dcalloc(size_t elems) {
  T* p = nullptr;
  DLAF_HIP_CHECK(pool_malloc(reinterpret_cast<void**>(&p), std::max<size_t>(elems, 1) * sizeof(T)));
  return p;
}
struct Node {
  long off;
  int n1, n2;
  int split;
  int height;
};
// leaves [lb, le) -> returns the height of the subtree; appends the merge nodes (children first)
int build_tree(long lb, long le, const std::vector<long>& leaf_off, std::vector<Node>& nodes, std::vector<long>& bounds) {
  if (lb + 1 == le)
    return 0;
  const long ls = (lb + le + 1) / 2;  // ceilDiv(lb + le, 2), impl.h:55
  const int h1 = build_tree(lb, ls, leaf_off, nodes, bounds);
  const int h2 = build_tree(ls, le, leaf_off, nodes, bounds);
  Node nd;
  nd.off = leaf_off[lb];
  nd.n1 = (int) (leaf_off[ls] - leaf_off[lb]);
  nd.n2 = (int) (leaf_off[le] - leaf_off[ls]);
  nd.split = (int) bounds.size();
  bounds.push_back(leaf_off[ls]);
  nd.height = std::max(h1, h2) + 1;
  nodes.push_back(nd);
  return nd.height;
}
// C = A B^T with the inner dimension cut into a multiple of the slab depth (the fast operand loaders) + a short tail
template <class R>
void gemm_main_tail(const GemmArgs<R>& g, hipStream_t s) {
  const int main = (g.K / 16) * 16;
  if (main > 0 && main < g.K) {
    GemmArgs<R> a = g;
    a.K = main;
    launch_gemm(a, s);
    GemmArgs<R> t = g;
    t.K = g.K - main;
    t.a = g.a + (long) main * g.lda;
    t.b = g.b + (long) main * g.ldb;
    t.beta = R(1);
    launch_gemm(t, s);
  }
  else
    launch_gemm(g, s);
}
int dc_leaf_size() {
  static const int v = [] {
    const char* e = std::getenv("DLAF_MI355X_DC_LEAF");
    const int x = e ? std::atoi(e) : 64;
    return std::min(64, std::max(1, x));
  }();
  return v;
}
}  // namespace

// tr != null (a process grid): the solver still runs replicated -- leaves, deflation, secular equation, every rank
// the same -- but the n^3 part, Q_new = Q_gathered U of the LARGE merges (the top levels of the tree: at least
// DLAF_MI355X_DC_DIST_MIN rows, default 4096), is cut into one column slice per rank and the slices are exchanged with
// one grouped broadcast round along the process rows and one along the process columns (an all-gather).  The reference
// distributes the whole solver (tridiag_solver/impl.h:362-476, multiplyEigenvectors merge.h:1694-1790); here the
// deflation and the secular equation are a few per cent of the stage and stay replicated.
template <class R>
int tridiag_solver_device(long n, int /*nb*/, R* d, R* e, R* w, R* z, long ldz, hipStream_t s, Transport* tr) {
  if (n <= 0)
    return 0;
  const int P = tr ? tr->nprow * tr->npcol : 1;
  static const long dist_min = [] {
    const char* e_ = std::getenv("DLAF_MI355X_DC_DIST_MIN");
    return e_ ? std::max(32L, std::atol(e_)) : 4096L;
  }();
  const int leaf = dc_leaf_size();
  const long nleaves = (n + leaf - 1) / leaf;
  std::vector<long> leaf_off(nleaves + 1);
  std::vector<int> leaf_n(nleaves);
  for (long i = 0; i <= nleaves; ++i)
    leaf_off[i] = std::min<long>(i * leaf, n);
  for (long i = 0; i < nleaves; ++i)
    leaf_n[i] = (int) (leaf_off[i + 1] - leaf_off[i]);
  std::vector<Node> nodes;
  std::vector<long> bounds;
  const int height = build_tree(0, nleaves, leaf_off, nodes, bounds);

  const long ldq = ((n + 15) / 16) * 16;
  R* q = dcalloc<R>((size_t) ldq * n);
  R* qt = dcalloc<R>((size_t) ldq * n);
  R* dlt = dcalloc<R>((size_t) ldq * n);
  R* u = dcalloc<R>((size_t) ldq * n);
  R* rv = dcalloc<R>((size_t) n * 8);  // dnew, z, z2, zhat, dsrt, zsrt, dsec, zsec
  int* iv = dcalloc<int>((size_t) n * 6);  // ord, srt, ctype, tpos, sec2t, dfl
  DcRot<R>* rots = dcalloc<DcRot<R>>((size_t) n);
  long* d_leaf_off = dcalloc<long>((size_t) nleaves + 1);
  int* d_leaf_n = dcalloc<int>((size_t) nleaves);
  long* d_bounds = dcalloc<long>(bounds.size());
  R* d_rho = dcalloc<R>(bounds.size());
  DcMerge* d_merges = dcalloc<DcMerge>(nodes.size());
  DcHeader<R>* d_headers = dcalloc<DcHeader<R>>(nodes.size());
  int* info = dcalloc<int>(1);
  DLAF_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), s));
  DLAF_HIP_CHECK(hipMemsetAsync(q, 0, (size_t) ldq * n * sizeof(R), s));
  DLAF_HIP_CHECK(hipMemcpyAsync(d_leaf_off, leaf_off.data(), (nleaves + 1) * sizeof(long), hipMemcpyHostToDevice, s));
  DLAF_HIP_CHECK(hipMemcpyAsync(d_leaf_n, leaf_n.data(), nleaves * sizeof(int), hipMemcpyHostToDevice, s));
  if (!bounds.empty())
    DLAF_HIP_CHECK(hipMemcpyAsync(d_bounds, bounds.data(), bounds.size() * sizeof(long), hipMemcpyHostToDevice, s));
  // merges ordered by height (children of a node are lower), each height one batch
  std::vector<Node> sorted = nodes;
  std::stable_sort(sorted.begin(), sorted.end(), [](const Node& a, const Node& b) { return a.height < b.height; });
  std::vector<DcMerge> hm(sorted.size());
  for (size_t i = 0; i < sorted.size(); ++i)
    hm[i] = DcMerge{sorted[i].off, sorted[i].n1, sorted[i].n2, sorted[i].split};
  if (!hm.empty())
    DLAF_HIP_CHECK(hipMemcpyAsync(d_merges, hm.data(), hm.size() * sizeof(DcMerge), hipMemcpyHostToDevice, s));

  launch_dc_cuppen(d, e, d_bounds, d_rho, (int) bounds.size(), s);
  DcLeafArgs<R> la;
  la.d = d;
  la.e = e;
  la.q = q;
  la.ldq = ldq;
  la.ord = iv;
  la.leaf_off = d_leaf_off;
  la.leaf_n = d_leaf_n;
  la.info = info;
  launch_dc_leaves(la, (int) nleaves, s);

  std::vector<DcHeader<R>> hh(sorted.size());
  size_t first = 0;
  const R one = R(1), zero = R(0);
  for (int h = 1; h <= height; ++h) {
    size_t last = first;
    int nmax = 0;
    while (last < sorted.size() && sorted[last].height == h) {
      nmax = std::max(nmax, sorted[last].n1 + sorted[last].n2);
      ++last;
    }
    const int cnt = (int) (last - first);
    if (cnt == 0)
      continue;
    DcMergeArgs<R> a;
    a.merges = d_merges + first;
    a.rho = d_rho;
    a.headers = d_headers + first;
    a.q = q;
    a.qt = qt;
    a.dlt = dlt;
    a.u = u;
    a.ldq = ldq;
    a.d = d;
    a.dnew = rv;
    a.z = rv + n;
    a.z2 = rv + 2 * n;
    a.zhat = rv + 3 * n;
    a.dsrt = rv + 4 * n;
    a.zsrt = rv + 5 * n;
    a.dsec = rv + 6 * n;
    a.zsec = rv + 7 * n;
    a.ord = iv;
    a.ord_out = iv;
    a.srt = iv + n;
    a.ctype = iv + 2 * n;
    a.tpos = iv + 3 * n;
    a.sec2t = iv + 4 * n;
    a.dfl = iv + 5 * n;
    a.rots = rots;
    launch_dc_prepare(a, cnt, s);
    launch_dc_rotate_gather(a, cnt, nmax, s);
    DLAF_HIP_CHECK(hipMemcpyAsync(hh.data() + first, d_headers + first, cnt * sizeof(DcHeader<R>), hipMemcpyDeviceToHost, s));
    DLAF_HIP_CHECK(hipStreamSynchronize(s));
    int kmax = 0;
    for (size_t i = first; i < last; ++i)
      kmax = std::max(kmax, hh[i].k);
    launch_dc_secular(a, cnt, kmax, s);
    // Q_new = Q_gathered U, by halves (multiplyEigenvectors, merge.h:974-1075): the columns that come from Q1 only have
    // zero rows below, those from Q2 only above
    for (size_t i = first; i < last; ++i) {
      const Node& nd = sorted[i];
      const DcHeader<R>& hd = hh[i];
      const int nn = nd.n1 + nd.n2;
      R* qb = q + nd.off + nd.off * ldq;
      const R* qtb = qt + nd.off + nd.off * ldq;
      const R* ub = u + nd.off + nd.off * ldq;
      if (hd.k > 0) {
        // the columns [c0, c1) of the non-deflated block this rank computes: all of them, or its slice
        const bool split = P > 1 && nn >= dist_min && hd.k >= 16 * P;
        auto cut = [&](int sidx) -> int { return sidx >= P ? hd.k : (int) (((long) hd.k * sidx / P) / 16 * 16); };
        const int me = split ? tr->myrow * tr->npcol + tr->mycol : 0;
        const int c0 = split ? cut(me) : 0, c1 = split ? cut(me + 1) : hd.k;
        if (c1 > c0) {
          GemmArgs<R> g;
          g.M = nd.n1;
          g.N = c1 - c0;
          g.K = hd.ku + hd.kd;
          g.a = qtb;
          g.lda = ldq;
          g.opa = 'N';
          g.b = ub + c0;  // U^T: k x (classes), used transposed
          g.ldb = ldq;
          g.opb = 'C';
          g.c = qb + (long) c0 * ldq;
          g.ldc = ldq;
          g.alpha = one;
          g.beta = zero;
          if (g.K > 0)
            gemm_main_tail(g, s);
          else
            DLAF_HIP_CHECK(hipMemset2DAsync(g.c, (size_t) ldq * sizeof(R), 0, (size_t) nd.n1 * sizeof(R), (size_t) (c1 - c0), s));
          GemmArgs<R> g2 = g;
          g2.M = nd.n2;
          g2.K = hd.kd + hd.kl;
          g2.a = qtb + nd.n1 + (long) hd.ku * ldq;
          g2.b = ub + (long) hd.ku * ldq + c0;
          g2.c = qb + nd.n1 + (long) c0 * ldq;
          if (g2.K > 0)
            gemm_main_tail(g2, s);
          else
            DLAF_HIP_CHECK(hipMemset2DAsync(g2.c, (size_t) ldq * sizeof(R), 0, (size_t) nd.n2 * sizeof(R), (size_t) (c1 - c0), s));
        }
        if (split) {
          // all-gather of the slices as whole columns of q (outside the merge's diagonal block they are zero on every
          // rank): slice r npcol + c belongs to rank (r, c), so a process row ends up with a contiguous run of columns
          R* colbase = q + nd.off * ldq;
          tr->group_begin();
          for (int c = 0; c < tr->npcol; ++c) {
            const int a0 = cut(tr->myrow * tr->npcol + c), a1 = cut(tr->myrow * tr->npcol + c + 1);
            if (a1 > a0 && tr->npcol > 1)
              tr->bcast(CommAxis::Row, c, tr->mycol, colbase + (long) a0 * ldq, colbase + (long) a0 * ldq,
                        (size_t) ldq * (size_t) (a1 - a0) * sizeof(R), s);
          }
          tr->group_end();
          tr->group_begin();
          for (int r = 0; r < tr->nprow; ++r) {
            const int a0 = cut(r * tr->npcol), a1 = cut((r + 1) * tr->npcol);
            if (a1 > a0 && tr->nprow > 1)
              tr->bcast(CommAxis::Col, r, tr->myrow, colbase + (long) a0 * ldq, colbase + (long) a0 * ldq,
                        (size_t) ldq * (size_t) (a1 - a0) * sizeof(R), s);
          }
          tr->group_end();
        }
      }
      if (hd.k < nn)
        DLAF_HIP_CHECK(hipMemcpy2DAsync(qb + (long) hd.k * ldq, (size_t) ldq * sizeof(R), qtb + (long) hd.k * ldq,
                                        (size_t) ldq * sizeof(R), (size_t) nn * sizeof(R), (size_t) (nn - hd.k),
                                        hipMemcpyDeviceToDevice, s));
    }
    launch_dc_finish(a, cnt, nmax, s);
    first = last;
  }
  launch_dc_output(q, ldq, d, iv, n, w, z, ldz, s);
  int h_info = 0;
  DLAF_HIP_CHECK(hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, s));
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  for (R* p : {q, qt, dlt, u, rv, d_rho})
    DLAF_HIP_CHECK(pool_free(p));
  DLAF_HIP_CHECK(pool_free(iv));
  DLAF_HIP_CHECK(pool_free(rots));
  DLAF_HIP_CHECK(pool_free(d_leaf_off));
  DLAF_HIP_CHECK(pool_free(d_leaf_n));
  DLAF_HIP_CHECK(pool_free(d_bounds));
  DLAF_HIP_CHECK(pool_free(d_merges));
  DLAF_HIP_CHECK(pool_free(d_headers));
  DLAF_HIP_CHECK(pool_free(info));
  if (h_info != 0)
    fatal("[dlaf_mi355x] tridiagonal_eigensolver: the QL iteration of the leaf at row %d did not converge\n", h_info - 1);
  return 0;
}

template <class R>
int tridiag_solver_host(long n, int nb, const R* d, const R* e, R* w, R* z, long ldz) {
  if (n <= 0)
    return 0;
  hipStream_t s;
  DLAF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  R* dd = dcalloc<R>((size_t) n);
  R* de = dcalloc<R>((size_t) n);
  R* dw = dcalloc<R>((size_t) n);
  R* dz = dcalloc<R>((size_t) n * n);
  DLAF_HIP_CHECK(hipMemsetAsync(de, 0, (size_t) n * sizeof(R), s));
  DLAF_HIP_CHECK(hipMemcpyAsync(dd, d, (size_t) n * sizeof(R), hipMemcpyHostToDevice, s));
  if (n > 1)
    DLAF_HIP_CHECK(hipMemcpyAsync(de, e, (size_t) (n - 1) * sizeof(R), hipMemcpyHostToDevice, s));
  const int r = tridiag_solver_device<R>(n, nb, dd, de, dw, dz, n, s, nullptr);
  DLAF_HIP_CHECK(hipMemcpyAsync(w, dw, (size_t) n * sizeof(R), hipMemcpyDeviceToHost, s));
  DLAF_HIP_CHECK(hipMemcpy2DAsync(z, (size_t) ldz * sizeof(R), dz, (size_t) n * sizeof(R), (size_t) n * sizeof(R), (size_t) n,
                                  hipMemcpyDeviceToHost, s));
  DLAF_HIP_CHECK(hipStreamSynchronize(s));
  for (R* p : {dd, de, dw, dz})
    DLAF_HIP_CHECK(pool_free(p));
  DLAF_HIP_CHECK(hipStreamDestroy(s));
  return r;
}

template int tridiag_solver_device<float>(long, int, float*, float*, float*, float*, long, hipStream_t, Transport*);
template int tridiag_solver_device<double>(long, int, double*, double*, double*, double*, long, hipStream_t, Transport*);
template int tridiag_solver_host<float>(long, int, const float*, const float*, float*, float*, long);
template int tridiag_solver_host<double>(long, int, const double*, const double*, double*, double*, long);

}  // namespace dlaf_mi355x
