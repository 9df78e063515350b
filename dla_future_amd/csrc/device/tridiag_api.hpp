// tridiag_api.hpp -- host-callable launchers of the gfx950 kernels of the eigensolver stages behind the reduction to
// band form: band -> tridiagonal (bulge chasing), the tridiagonal divide & conquer
// eigensolver, and the back-transformation band <- tridiagonal.  Everything enqueues on the given stream.
//
// Reference: include/dlaf/eigensolver/band_to_tridiag/mc.h (the reference runs this stage on the CPU, one task per
// sweep step, also for GPU matrices), include/dlaf/eigensolver/tridiag_solver/{impl,merge}.h (leaf solver stedc and
// the deflation / secular-equation steps on the CPU, GEMMs on the GPU), include/dlaf/eigensolver/bt_band_to_tridiag/impl.h.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace dlaf_mi355x {

// ------------------------------------------------------------------------------------------ band storage
// band[c * ldb + o] = A(c + o, c), o in [0, ldb), ldb = 2 * b: the diagonal, the b sub-diagonals and b - 1 rows of
// room for the bulge (BandBlock, band_to_tridiag/mc.h:180-206: ld = 2 b - 1 between columns of the same row).
//
// From the lower tiles of a tile-layout matrix (tile (il, jl) at tiles + (il + jl * ltr) * nb^2, ld nb, global tile
// gi = il * pr + ri, gj = jl * pc + ci): the entries of tiles that are not local are set to zero (a process grid sums
// the images of its ranks).
template <class T>
void launch_band_extract(const T* tiles, long ltr, int nb, int pr, int ri, int pc, int ci, long n, int b, T* band,
                         hipStream_t stream);

// ------------------------------------------------------------------------------------------ band -> tridiagonal
// All sweeps in ONE launch of persistent workgroups: a workgroup draws the next sweep from a counter and runs its
// steps; step t of sweep s waits until sweep s - 1 has finished t + 2 steps (the counting semaphores of
// mc.h:683-709; the register kernel waits for step t and the first column of step t + 1, kernels_tridiag.hip).  Hand-offs between workgroups: write-through stores, sc1 loads, one progress word per sweep.
//   vout (n x n, ldv): the compact reflectors, tau in the place of the leading 1 (band_to_tridiag.h:56-63)
//   sync: b2t_sync_words(n) unsigned words, zeroed by the launcher
// Afterwards d[i] = Re band[i * ldb], e[i] = Re band[i * ldb + 1] (launch_tridiag_extract).
template <class T>
void launch_band_to_tridiag(T* band, long n, int b, T* vout, long ldv, unsigned* sync, int* info, hipStream_t stream);
// (a sweep's progress word has a 128-byte line to itself: pollers and publishers of neighbouring sweeps would
//  otherwise meet in one L2 channel)
constexpr int kB2tProgressStride = 32;
inline size_t b2t_sync_words(long n) {
  return (size_t) n * kB2tProgressStride + 64;
}
template <class T>
void launch_tridiag_extract(const T* band, long n, int b, real_t<T>* d, real_t<T>* e, hipStream_t stream);
int b2t_max_band();

// ------------------------------------------------------------------------------------------ band <- tridiagonal
// Block (ib, jb), ib >= jb, of the compact reflector matrix = the reflectors of the sweeps jb b .. jb b + b - 1 at step
// ib - jb (bt_band_to_tridiag.h:64-77).  Its well-formed image: vx + blk * 2 b * b, 2 b x b (ld 2 b), column k with its
// 1 in row k; taus + blk * b.  blk = jb * nblk + ib, nblk = ceil(n / b).  Columns that hold no reflector are zero.
template <class T>
void launch_b2t_expand(const T* vout, long ldv, long n, int b, T* vx, T* taus, hipStream_t stream, bool transposed = false);

// ------------------------------------------------------------------------------------------ eigenvector plumbing
// e[r + cl * lde] = (T) z[r + gc * ldz]: the local columns cl of a block-cyclic column axis (pc processes, this one at
// position ci from the source, block nb) out of the replicated real eigenvector matrix z
template <class R, class T>
void launch_cols_gather_cast(const R* z, long ldz, long n, int nb, int pc, int ci, long ncols_loc, T* e, long lde,
                             hipStream_t stream);
// the local tile rows (pr processes, position ri) of those columns into a tile-layout matrix of ltr x ltc tiles
template <class T>
void launch_rows_to_tiles(const T* e, long lde, long n, long ncols_loc, int nb, int pr, int ri, long ltr, long ltc, T* tiles,
                          hipStream_t stream);

// ------------------------------------------------------------------------------------------ tridiagonal D&C
// (kernels of the divide & conquer solver: declared in tridiag_dc.hpp)

}  // namespace dlaf_mi355x
