// device_api.hpp -- host-callable launchers of the gfx950 tile kernels.  Everything here
// enqueues work on the given HIP stream and returns; no allocation, no synchronisation
// (the launchers are hipGraph-capturable).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace dlaf_mi355x {

// Zero device memory and RETURN ONLY WHEN IT IS ZERO.  On this stack hipMemset -- the "synchronous" entry -- returns
// before its fill kernel has run (measured: 3 us for a call queued behind a 200 ms kernel, tools/memset_sync_probe.hip,
// profiles/r04_memset_sync_probe.txt), and the fill sits on the null stream, which the library's non-blocking streams do
// not synchronise with: a status word "zeroed" that way can be zeroed again AFTER a kernel on another stream has
// written it.  State that kernels on other streams will touch is therefore zeroed with this, or on the stream that
// uses it.
inline hipError_t zero_device_now(void* p, size_t bytes) {
  hipError_t e = hipMemsetAsync(p, 0, bytes, nullptr);
  if (e == hipSuccess)
    e = hipStreamSynchronize(nullptr);
  return e;
}


// Block size of the diagonal factorization / inverted diagonal blocks used by the TRSM.
constexpr int kDiagBlock = 64;
// *info value stored by a kernel whose bounded inter-workgroup wait expired (workgroups of the cooperative
// POTRF not co-scheduled): a runtime failure, never a property of the matrix; the host aborts on it.
constexpr int kInfoSchedulingFailure = -1000000;

// ------------------------------------------------------------------------------------------
// Trailing update (tile::herk + tile::gemm of a whole step in ONE launch):
//   for local tiles (il, jl) in [il0,il1) x [jl0,jl1) with global indices gi = il*pr + ri,
//   gj = jl*pc + ci:   gi > gj:  C(il,jl) -= A(il) * B(jl)^H          (gemm, impl.h:83-94)
//                      gi == gj: lower(C(il,jl)) -= A(il) * B(jl)^H   (herk, impl.h:70-80)
//                      gi < gj:  nothing
// C lives in tile layout (tile (il,jl) at c + il*c_tsr + jl*c_tsc, leading dimension ldc);
// A(il) = a + (il-il0)*a_ts (rows(gi) x K, lda), B(jl) = b + (jl-jl0)*b_ts (rows(gj) x K, ldb).
// rows(g) = nb except for the last global tile g == nt-1 which has last_rows.
template <class T>
struct UpdateArgs {
  T* c;
  long c_tsr, c_tsc;
  int ldc;
  const T* a;
  long a_ts;
  int lda;
  const T* b;
  long b_ts;
  int ldb;
  int il0, il1, jl0, jl1;
  int nb, K;
  int pr, ri, pc, ci;
  int nt, last_rows;
  const int* info;  // device flag: non-zero => a previous POTRF failed, kernels return at once
  // rect != 0: every tile of the rectangle is a gemm (no triangle, no herk); the column axis then has its
  // own tile count / last extent (the triangular solver updates an m x n right-hand side, solver.cpp)
  int rect = 0;
  int nt_c = 0, last_cols = 0;
  // b_period > 1: the transposed panel is stored grouped by the process row that sent it (one broadcast
  // per root instead of one per tile): B(jl) = b + ((jl-jl0) % b_period) * b_ts2 + ((jl-jl0) / b_period) * b_ts
  int b_period = 1;
  long b_ts2 = 0;
  int b_jl0 = -1;  // local tile column that b stands for (-1: jl0)
  // Two-segment product (K1 > 0): columns k < K1 of the operands come from a / b, columns K1 <= k < K from
  // a2 / b2 (same tile strides and leading dimensions): two panels applied in one pass
  //     C -= A1 B1^H + A2 B2^H            (one read-modify-write of C for two steps of the factorization)
  // her2k != 0 (needs K1 > 0): with a = [X | L], b = [L | X] the launch computes C -= X L^H + L X^H; diagonal
  // tiles take their column operand from the column panel again with the segments swapped ([L_j | X_j]) and
  // keep the triangle mask (tile::her2k of gen_to_std).
  int K1 = 0;
  const T* a2 = nullptr;
  const T* b2 = nullptr;
  int her2k = 0;
};
// role: 0 trailing bulk, 1 lookahead column, 2 in-tile POTRF update / single-tile entries, 3 residual checker
// and triangular solver (same code, separate kernel names, so that profiles of the factorization stay clean)
// max_blocks > 0 (with counters = 16 device words of scratch): launch at most that many workgroups and
// let them pull work items (persistent form): what is left of the GPU stays free for kernels that
// must run beside the update.  excl_slots > 0 (persistent form only): the workgroups that land on the first
// compute units of every XCD -- as many whole compute units as hold excl_slots workgroups -- leave at once, so
// max_blocks may cover the whole GPU and the side kernels still find compute units of their own.
template <class T>
void launch_update(const UpdateArgs<T>& args, hipStream_t stream, int role = 0, long max_blocks = 0,
                   unsigned* counters = nullptr, bool counters_are_zero = false, long excl_slots = 0);
template <class T>
int update_blocks_per_cu();

// ------------------------------------------------------------------------------------------
// Panel TRSM (tile::trsm Right/Lower/ConjTrans/NonUnit of a whole panel in ONE launch,
// impl.h:56-67):   X(il) = B(il) * L^-H for local tiles il in [il0, il1), in place.
// L: n x n lower triangular (ldl); winv: ceil(n/64) inverted diagonal blocks of L, block j is a
// dense 64 x 64 column-major array at winv + j*64*64 (lower triangle valid, rest zero).
template <class T>
struct TrsmArgs {
  T* b;
  long b_ts;
  int ldb;
  int il0, il1;
  int pr, ri;
  int nb, nt, last_rows;
  const T* l;
  int ldl;
  const T* winv;
  int n;
  const int* info;
  // upper != 0: X(il) = B(il) * U^-H with U = l upper triangular (the 64-column blocks are swept right to
  // left); winv block j then holds inv(U_jj) (upper triangle valid, rest zero)
  int upper = 0;
  // != 0: the waves run at raised priority (a panel solve on the critical path beside the bulk update)
  int prio = 0;
};
template <class T>
void launch_trsm(const TrsmArgs<T>& args, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Diagonal block factorization + inversion (one workgroup):  a (jb x jb, lda, jb <= 64) is
// overwritten by its lower Cholesky factor (strict upper part untouched); winv_block (64 x 64,
// ld 64) receives inv(L) (lower, zero elsewhere).  On a non-positive pivot at column c the
// kernel stores info_base + c + 1 into *info (first failure wins) and leaves garbage.
// factor == false: a already holds a triangular matrix; only winv_block is produced.  In that mode
// upper: a is upper triangular and winv_block receives inv(a) (upper); unit: the diagonal of a is taken as 1.
template <class T>
void launch_potrf_diag(T* a, int lda, int jb, T* winv_block, int* info, int info_base, hipStream_t stream,
                       bool factor = true, bool upper = false, bool unit = false);

// All ceil(kb/64) diagonal 64 x 64 blocks of one triangular kb x kb tile inverted in ONE launch (the
// triangular solver's per-tile preparation; nothing is factored, the tile is only read).
template <class T>
void launch_invert_diag_blocks(const T* tile, int ld, int kb, T* winv, int* info, hipStream_t stream, bool upper,
                               bool unit);

// Whole diagonal tile (kb x kb, ld) in one resident cooperative launch of ceil(kb/64) workgroups:
// lower Cholesky factor in place + the ceil(kb/64) inverted diagonal blocks in winv.  sync: device
// scratch of at least G + G*G unsigned, G = ceil(kb/64) (zeroed by the launcher on the stream):
// potrf_coop_sync_words(kb).
template <class T>
void launch_potrf_coop(T* tile, int ld, int kb, T* winv, int* info, int info_base, unsigned* sync,
                       hipStream_t stream, bool sync_is_zero = false, bool count_strips = true);
// count_strips: the strips register in the per-compute-unit table the bulk update kernel consults (the POTRF yield)
// update launches of this process so far in persistent form, and of those with exclusive compute units
void update_launch_stats(long* persistent, long* exclusive);
// diagnosis hook (DLAF_MI355X_POTRF_TRACE=1): 32 device words the launches of a factorization's FIRST diagonal tile
// write what their first two strips saw into (null: off); the caller zeroes them on the stream before the launch
unsigned long long* potrf_coop_trace_buffer();
inline size_t potrf_coop_sync_words(int kb) {
  const size_t g = (size_t) ((kb + kDiagBlock - 1) / kDiagBlock);
  return g + g * g;
}

// ------------------------------------------------------------------------------------------
// Layout kernels between the caller's column-major local array (staged on the device) and the
// tile layout (tile (il,jl) at dst + (il + jl*ltr) * nb*nb, ld = nb).
//   transpose == 0: tile(il,jl)[r][c]  = src[(il*nb + r) + (jl*nb + c)*lds]     (uplo = L)
//   transpose == 1: tile(il,jl)[r][c]  = src[(jl*nb + c) + (il*nb + r)*lds]     (uplo = U seen as
//                   the lower factorization of the transposed view; the view's tile grid is
//                   ltr x ltc with ltr = source tile COLUMNS)
// Only view-tiles with global row index >= global column index are moved (the uplo triangle);
// within diagonal tiles every element is moved to the device and only the triangle is moved back.
template <class T>
struct LayoutArgs {
  T* tiles;
  T* cm;         // column-major local array on the device
  long ld_cm;
  int ltr, ltc;  // local tile rows / cols of the (possibly transposed) VIEW
  int nb;
  long rows, cols;  // local element extents of the VIEW
  int pr, ri, pc, ci;  // global tile index of view-tile: gi = il*pr + ri, gj = jl*pc + ci
  int transpose;
  // general matrices (triangular solver operands): full != 0 moves every tile of the rectangle (no uplo
  // triangle); conj != 0 conjugates on the way (with transpose: the conjugate-transposed view); to-tiles
  // only: scale != 0 multiplies by alpha (the solver's alpha * B)
  int full = 0, conj = 0, scale = 0;
  T alpha{};
  // restrict the move to view tile columns [jl_first, jl_first + jl_count) (jl_count 0: to the last one)
  int jl_first = 0, jl_count = 0;
  // view row r / column c takes source view row rows-1-r / column cols-1-c (the solver turns a backward sweep over an
  // upper triangular matrix into the forward sweep of its reversal)
  int rev_rows = 0, rev_cols = 0;
};
template <class T>
void launch_to_tiles(const LayoutArgs<T>& args, hipStream_t stream);
template <class T>
void launch_from_tiles(const LayoutArgs<T>& args, hipStream_t stream);

template <class T>
void launch_copy2d(T* dst, long ldd, const T* src, long lds, int rows, int cols, int transpose, int mask,
                   hipStream_t stream);

// Batched tile transforms (gen_to_std): tile t at src + t*sstride (rows x cols, lds) -> dst + t*dstride.
//   mode 0: dst = src^H (cols x rows);  mode 1: dst = scale * (full Hermitian image of src's lower triangle);
//   mode 2: lower(dst) = lower(src^H), rest of dst untouched, real diagonal.
template <class T>
void launch_tile_xform(T* dst, long ldd, long dstride, const T* src, long lds, long sstride, int rows, int cols,
                       int count, int mode, double scale, hipStream_t stream);

// dst tile = alpha * op(src tile) for a batch of tiles; op: 0 adjoint, 3 transpose, 4 conjugate, 5 copy
template <class T>
void launch_tile_xform_alpha(T* dst, long ldd, long dstride, const T* src, long lds, long sstride, int rows, int cols,
                             int count, int mode, T alpha, bool use_alpha, hipStream_t stream);

// checker helpers: max |a_ij| over the lower triangle of the local tiles (into *out, device double) and
// zeroing of the strict upper part of the local diagonal tiles
template <class T>
void launch_max_norm(const T* tiles, int ltr, int ltc, int nb, long rows, long cols, int pr, int ri, int pc, int ci,
                     double* out, hipStream_t stream);
template <class T>
void launch_zero_upper_diag(T* tiles, int ltr, int ltc, int nb, int pr, int ri, int pc, int ci, hipStream_t stream);

// one-off: opt the kernels into > 64 KiB of dynamic LDS
void device_kernels_init();

}  // namespace dlaf_mi355x
