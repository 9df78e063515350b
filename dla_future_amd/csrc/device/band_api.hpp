// band_api.hpp -- host-callable launchers of the gfx950 kernels of the eigensolver stages built on the
// Cholesky path's MFMA core: reduction of a Hermitian matrix to band form and its back-transformation
// (SURVEY.md section 8(f) item 4).  Everything enqueues on the given stream and returns.
//
// Reference: include/dlaf/eigensolver/reduction_to_band/impl.h (panel reflectors :297-361 -- computed on the
// CPU even by the reference's GPU backend, :881-961 -- hemm :465-517, W2 / X update :448-462, :520-542),
// include/dlaf/factorization/qr/t_factor_impl.h, include/dlaf/eigensolver/bt_reduction_to_band/impl.h.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace dlaf_mi355x {

// ------------------------------------------------------------------------------------------
// General product on column-major operands:  C = alpha * opA(A) * opB(B) + beta * C
//   opa 'N': A is M x K;  'C': A is K x M, used conjugate-transposed
//   opb 'N': B is K x N;  'C': B is N x K, used conjugate-transposed
// ksplit > 1: K is cut into ksplit chunks, every chunk's block product goes to `partial`
// (gemm_partial_elems elements) and a second kernel sums them in a fixed order -- the form the tall-skinny
// products with a small result (V^H V, W^H X: K = the matrix size) take, since one workgroup per output
// block would leave the GPU idle.
template <class T>
struct GemmArgs {
  int M = 0, N = 0, K = 0;
  const T* a = nullptr;
  long lda = 0;
  char opa = 'N';
  const T* b = nullptr;
  long ldb = 0;
  char opb = 'N';
  T* c = nullptr;
  long ldc = 0;
  T alpha{}, beta{};
  int ksplit = 1;
  T* partial = nullptr;
  // strided batch (ksplit == 1 only): product q of [0, batch) takes a + q * sa, b + q * sb, c + q * sc
  int batch = 1;
  long sa = 0, sb = 0, sc = 0;
};
template <class T>
void launch_gemm(const GemmArgs<T>& args, hipStream_t stream);
template <class T>
size_t gemm_partial_elems(int M, int N, int ksplit);
// a ksplit that fills the GPU for an M x N result with inner dimension K (1 when the output blocks alone do)
template <class T>
int gemm_pick_ksplit(int M, int N, long K);

// ------------------------------------------------------------------------------------------
// Products of a tile-layout matrix with a tall panel (tile (il, jl) at tiles + (il + jl * ltr) * nb^2, ld nb;
// global tile indices gi = il * pr + ri, gj = jl * pc + ci):
//   kind S ("straight"): out rows of local tile row il  +=  A(il, jl)   * W[rows of global tile gj]
//   kind T ("adjoint"):  out rows of local tile col jl  +=  A(il, jl)^H * W[rows of global tile gi]
// herm != 0 (xHEMM with a lower-stored Hermitian matrix, reduction_to_band/impl.h:465-517): kind S takes the
// tiles gi >= gj (the diagonal tile through its Hermitian image), kind T the tiles gi > gj.  herm == 0: every
// tile of the local range (bt_reduction_to_band's W2 = W^H C runs as kind T on the tiles of C).
// The sources of an output tile are cut into runs of `chunk_*` consecutive tiles, one workgroup per (output block,
// run): equal work per workgroup although the output tiles of a triangle have 1 .. nt sources.  Run q of an output
// tile writes layer q, a column-major panel part_*[q]: (out tiles * nb) x ncols; a tile with fewer runs leaves the
// higher layers alone and the reduction knows (tile_panel_runs).
// W: row of global element g at w[(g - e0) + c * ldw].
template <class T>
struct TilePanelArgs {
  const T* tiles = nullptr;
  long ltr = 0;
  int nb = 1;
  int il0 = 0, il1 = 0, jl0 = 0, jl1 = 0;
  int pr = 1, ri = 0, pc = 1, ci = 0;
  int nt_r = 0, last_rows = 0;  // row axis: global tiles, extent of the last one
  int nt_c = 0, last_cols = 0;  // column axis
  int herm = 0;
  const T* w = nullptr;
  long ldw = 0;
  long e0 = 0;
  int ncols = 0;
  int kinds = 3;  // bit 0: S, bit 1: T
  int chunk_s = 1, chunk_t = 1;    // source tiles per run
  int layers_s = 1, layers_t = 1;  // most runs an output tile can have = layers allocated (tile_panel_layers)
  T* part_s = nullptr;  // layers_s (+ 1, see `split`) panels of (il1 - il0) * nb rows
  T* part_t = nullptr;  // layers_t (+ 1) panels of (jl1 - jl0) * nb rows
  // set by launch_tile_panel: != 0 when the full off-diagonal tiles went through the fixed-mode kernels and the rest
  // (diagonal tiles, ragged last tile row / column) through the general one into ONE EXTRA layer per kind
  int split = 0;
};
template <class T>
void launch_tile_panel(TilePanelArgs<T>& args, hipStream_t stream);
// Hermitian case: x[(g - e0) + c * ldx] for the global rows g of the tiles I0 .. nt-1 = the layer sums of the
// kinds whose output tile is local (zero otherwise; rows g < r0 are set to zero)
template <class T>
void launch_hemm_reduce(const TilePanelArgs<T>& args, long r0, T* x, long ldx, hipStream_t stream);
// kind T alone (herm == 0): out[(jl - jl0) * nb + r + c * ldo] = sum over the layers
template <class T>
void launch_layers_reduce(const T* part, int layers, long rows, int ncols, T* out, long ldo, hipStream_t stream);
// source tiles per run for `out_tiles` output tiles with up to `max_src` sources each (about a thousand workgroups
// per kind), and the layers that needs: ceil(max_src / chunk)
int tile_panel_pick_chunk(long out_tiles, int nb, int ncols, long max_src, bool triangle, size_t elem_size);
inline int tile_panel_layers(long max_src, int chunk) {
  return (int) ((max_src + chunk - 1) / chunk > 0 ? (max_src + chunk - 1) / chunk : 1);
}

// ------------------------------------------------------------------------------------------
// Panel of Householder reflectors (xGEQR2 of an m x b panel, reflectors of size 1 skipped; the reference's
// computePanelReflectors, impl.h:297-361): ONE cooperative launch.  The panel is held TRANSPOSED, qt[c + r * b]
// = element (r, c): a thread owns a column, a workgroup a range of rows; per reflector one grid-wide
// exchange of partial sums (norm and P^H x in the same reduction).  taus[0 .. nr).
// scratch: panel_qr_scratch_bytes(b) of device memory (partials + counters, zeroed by the launcher).
template <class T>
void launch_panel_qr(T* qt, long m, int b, int nr, T* taus, void* scratch, int* info, hipStream_t stream);
size_t panel_qr_scratch_bytes(int b, size_t elem_size);

// Panel QR, blocked (kernels_hr.hip): the same reflectors from CholeskyQR2 + Householder reconstruction -- three
// tall-skinny passes on the Cholesky path's kernels instead of b grid-wide exchanges.  These are its small kernels; the
// sequence is in csrc/host/red2band.cpp (panel_qr_blocked).  `flag` != 0: the panel is not safely positive definite,
// nothing downstream of the gate writes, the caller falls back to launch_panel_qr.
bool panel_qr_blocked_supported(int b, long m, int nr, size_t elem_size, bool is_complex);
// qt[c + r * b] <-> cm[r + c * ld] (m rows, b columns); flag may be null
template <class T>
void launch_hr_transpose(T* qt, int b, long m, T* cm, long ld, bool to_cm, const int* flag, hipStream_t stream);
// raises *flag when max / min of the diagonal of the lower factor l (b x b, ld) exceeds `limit` (<= 0: no gate) or is
// not finite; zeroes the strict upper triangle of l
template <class T>
void launch_hr_gate(T* l, int ld, int b, double limit, int* flag, hipStream_t stream);
// g2 = Q1^T Q1 after the first pass: within `tol` of the identity -> g2 := I, *skip = 1 (the second pass returns at
// once: it takes `skip` as its status word); farther than 0.1 -> *flag
template <class T>
void launch_hr_orth(T* g2, int ld, int b, double tol, int* skip, int* flag, hipStream_t stream);
// Householder reconstruction of the top b x b block from it and R = L2^T L1^T (see kernels_hr.hip: hr_lu_kernel)
template <class T>
void launch_hr_lu(T* q, long ldq, int b, const T* rmat, T* lu, T* y1, T* tb, T* taus, const int* flag, hipStream_t stream);
void hr_kernels_init();

// panel column block [c0, c0 + b) of tile column jl (tile-local columns cc .. cc + b) <-> transposed panel:
// qt[c + (g - e0) * b] for the global rows g >= r0 of the local tiles il >= il0 (to_panel) or back (!to_panel)
template <class T>
void launch_panel_move(T* tiles, long ltr, int nb, int il0, int il1, int jl, int pr, int ri, int nt, int last_rows,
                       int cc, int b, T* qt, long e0, long r0, bool to_panel, hipStream_t stream);
// v[(g - e0) + j * ldv], g in [e0, n): well-formed reflectors (impl.h:364-422): 0 above the diagonal element
// g == r0 + j, 1 on it, qt below; columns j >= nr and rows g < r0 are zero
template <class T>
void launch_make_v(const T* qt, int b, int nr, long e0, long r0, long n, T* v, long ldv, hipStream_t stream);
// T factor of a block of k reflectors from S = V^H V (k x k, lds) and taus (t_factor_impl.h:60-131); t: k x k, ldt,
// upper triangular, the strict lower part set to zero
template <class T>
void launch_tfactor(const T* s, long lds_, const T* taus, int k, T* t, long ldt, hipStream_t stream, int batch = 1,
                    long bs = 0, long btau = 0, long bt = 0);  // strided batch: problem q at s + q bs, taus + q btau, t + q bt
// zero rows [0, nrows) x ncols of a column-major array
template <class T>
void launch_zero_rows(T* x, long ldx, long nrows, int ncols, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Fused application of the reflector blocks of bt_band_to_tridiagonal (kernels_bt.hip): fp64, band 128.
//   transpose: dst[c + r * ldd] = src[r + c * lds] (the stage works on E transposed)
//   apply: blocks q = 0 .. nblocks-1 of one wavefront (V^T with b-element rows at vt + q * blk_stride, W column-major
//          with 2b rows at wr + q * blk_stride, rows
//          [r0_first + q * 2b, + rows) of E) on every 64-column strip of et (ncols columns, ldet >= ncols)
bool bt_fused_supported(int band, size_t elem_size, bool is_complex);
void launch_bt_transpose(const double* src, long lds_, long rows, long cols, double* dst, long ldd, hipStream_t stream);
void launch_bt_apply(const double* vt, const double* wr, long blk_stride, int nblocks, double* et, long ldet, long ncols,
                     long r0_first, int rows, hipStream_t stream);
void bt_kernels_init();

void band_kernels_init();

}  // namespace dlaf_mi355x
