// tridiag_dc.hpp -- launchers of the divide & conquer tridiagonal eigensolver kernels (kernels_tridiag_dc.hip).
// Reference: include/dlaf/eigensolver/tridiag_solver/{impl,merge}.h.  All index arrays hold positions relative to the
// first element of the (sub)problem they belong to.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace dlaf_mi355x {

template <class R>
struct DcLeafArgs {
  R* d;             // diagonal (Cuppen-modified), overwritten with the eigenvalues of each leaf
  const R* e;       // off-diagonal
  R* q;             // eigenvector matrix (ldq), the leaf's diagonal block is written
  long ldq;
  int* ord;         // ascending order of each leaf's eigenvalues
  const long* leaf_off;
  const int* leaf_n;
  int* info;
};

struct DcMerge {
  long off;    // first index of the merged problem
  int n1, n2;  // sizes of its children
  int split;   // index of its boundary in the array of off-diagonal elements removed by Cuppen's decomposition
};
template <class R>
struct DcHeader {
  int k, ku, kd, kl, nrot;
  R rho;
};
template <class R>
struct DcRot {
  int a, b;
  R c, s;
};

template <class R>
struct DcMergeArgs {
  const DcMerge* merges;  // device
  const R* rho;            // off-diagonal elements removed by Cuppen's decomposition, one per boundary
  DcHeader<R>* headers;      // device, one per merge
  R* q;        // current eigenvectors (block diagonal over the subproblems)
  R* qt;       // gathered columns
  R* dlt;      // differences d_i - lambda_j of the rank-one problems
  R* u;        // eigenvectors of the rank-one problems (rows at the gathered column positions)
  long ldq;    // leading dimension of all four
  R* d;        // eigenvalues of the subproblems, storage (column) order
  R* dnew;
  R* z;
  R* z2;
  R* zhat;
  R* dsrt;
  R* zsrt;
  R* dsec;
  R* zsec;
  int* ord;
  int* ord_out;
  int* srt;
  int* ctype;
  int* tpos;
  int* sec2t;
  int* dfl;    // deflated entries (positions in the merged order), ascending by value
  DcRot<R>* rots;
};

template <class R>
void launch_dc_cuppen(R* d, const R* e, const long* bounds, R* rho, int nsplit, hipStream_t s);
template <class R>
void launch_dc_leaves(const DcLeafArgs<R>& a, int nleaves, hipStream_t s);
template <class R>
void launch_dc_prepare(const DcMergeArgs<R>& a, int nmerges, hipStream_t s);
template <class R>
void launch_dc_rotate_gather(const DcMergeArgs<R>& a, int nmerges, int nmax, hipStream_t s);
template <class R>
void launch_dc_secular(const DcMergeArgs<R>& a, int nmerges, int kmax, hipStream_t s);
template <class R>
void launch_dc_finish(const DcMergeArgs<R>& a, int nmerges, int nmax, hipStream_t s);
template <class R>
void launch_dc_output(const R* q, long ldq, const R* d, const int* ord, long n, R* w, R* z, long ldz, hipStream_t s);

}  // namespace dlaf_mi355x
