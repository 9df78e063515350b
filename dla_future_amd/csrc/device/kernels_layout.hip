// kernels_layout.hip -- moves between the caller's column-major local array (ScaLAPACK local
// layout, staged in device memory) and the device tile layout, plus the one-off kernel setup.
//
// Reference: MatrixMirror + copy (matrix/matrix_mirror.h:137-173, matrix/copy.h:38-60,
// copy_tile.h:143-153: one hipMemcpy2DAsync per tile).  The MI355X build keeps the factorization
// in tile layout (matrix/layout_info.h:140-159 tileLayout: ld = nb, column of tiles contiguous) so
// that panels are contiguous for RCCL and MFMA slabs are 1 KiB runs; these kernels do the
// relayout at HBM speed.  uplo = U is handled as the lower factorization of the transposed view,
// so the relayout optionally transposes (through a 32x33 LDS tile).
#include <algorithm>

#include "device_api.hpp"

namespace dlaf_mi355x {

constexpr int kLT = 32;

// grid: x = sub-block inside a tile (ceil(nb/32)^2), y = il, z = jl
template <class T, bool TO_TILES>
__global__ __launch_bounds__(kThreads) void layout_kernel(LayoutArgs<T> p, int sb) {
  __shared__ T buf[kLT][kLT + 1];
  const int il = blockIdx.y, jl = blockIdx.z;
  const int gi = il * p.pr + p.ri, gj = jl * p.pc + p.ci;
  if (gi < gj)
    return;
  const bool diag = (gi == gj);
  const int r0 = (blockIdx.x % sb) * kLT, c0 = (blockIdx.x / sb) * kLT;
  const long vrow0 = (long) il * p.nb, vcol0 = (long) jl * p.nb;  // view element origin of the tile
  const int rows_tile = (int) min((long) p.nb, p.rows - vrow0);
  const int cols_tile = (int) min((long) p.nb, p.cols - vcol0);
  if (r0 >= rows_tile || c0 >= cols_tile)
    return;
  if (diag && r0 + kLT - 1 < c0)
    return;
  T* tile = p.tiles + ((long) il + (long) jl * p.ltr) * p.nb * p.nb;
  const int tx = threadIdx.x % kLT, ty = threadIdx.x / kLT;  // 32 x 8

  if (!p.transpose) {
    for (int cc = ty; cc < kLT; cc += kThreads / kLT) {
      const int r = r0 + tx, c = c0 + cc;
      if (r < rows_tile && c < cols_tile) {
        T* td = tile + r + (long) c * p.nb;
        T* cd = p.cm + (vrow0 + r) + (vcol0 + c) * p.ld_cm;
        if (TO_TILES)
          *td = *cd;
        else if (!diag || r >= c)
          *cd = *td;
      }
    }
  }
  else {
    // view(r, c) = cm(c, r): cm element (vcol0 + c) + (vrow0 + r) * ld -> contiguous along c
    if (TO_TILES) {
      for (int rr = ty; rr < kLT; rr += kThreads / kLT) {
        const int r = r0 + rr, c = c0 + tx;
        if (r < rows_tile && c < cols_tile)
          buf[rr][tx] = p.cm[(vcol0 + c) + (vrow0 + r) * p.ld_cm];
      }
      __syncthreads();
      for (int cc = ty; cc < kLT; cc += kThreads / kLT) {
        const int r = r0 + tx, c = c0 + cc;
        if (r < rows_tile && c < cols_tile)
          tile[r + (long) c * p.nb] = buf[tx][cc];
      }
    }
    else {
      for (int cc = ty; cc < kLT; cc += kThreads / kLT) {
        const int r = r0 + tx, c = c0 + cc;
        if (r < rows_tile && c < cols_tile)
          buf[tx][cc] = tile[r + (long) c * p.nb];
      }
      __syncthreads();
      for (int rr = ty; rr < kLT; rr += kThreads / kLT) {
        const int r = r0 + rr, c = c0 + tx;
        if (r < rows_tile && c < cols_tile && (!diag || r >= c))
          p.cm[(vcol0 + c) + (vrow0 + r) * p.ld_cm] = buf[rr][tx];
      }
    }
  }
}

template <class T, bool TO_TILES>
static void launch_layout(const LayoutArgs<T>& a, hipStream_t stream) {
  if (a.ltr <= 0 || a.ltc <= 0 || a.nb <= 0)
    return;
  const int sb = (a.nb + kLT - 1) / kLT;
  dim3 grid((unsigned) (sb * sb), (unsigned) a.ltr, (unsigned) a.ltc);
  hipLaunchKernelGGL((layout_kernel<T, TO_TILES>), grid, dim3(kThreads), 0, stream, a, sb);
}

template <class T>
void launch_to_tiles(const LayoutArgs<T>& a, hipStream_t stream) {
  launch_layout<T, true>(a, stream);
}
template <class T>
void launch_from_tiles(const LayoutArgs<T>& a, hipStream_t stream) {
  launch_layout<T, false>(a, stream);
}

// dst(i,j) = src(transpose ? (j,i) : (i,j)) over the rows x cols of dst;
// mask 0: all, 1: i >= j only, 2: i <= j only.  Small helper for the single-tile entry points.
template <class T>
__global__ void copy2d_kernel(T* dst, long ldd, const T* src, long lds, int rows, int cols, int transpose, int mask) {
  const int i = blockIdx.x * 32 + (threadIdx.x % 32);
  for (int j = blockIdx.y * 8 + threadIdx.x / 32; j < cols; j += gridDim.y * 8) {
    if (i < rows && (mask == 0 || (mask == 1 && i >= j) || (mask == 2 && i <= j)))
      dst[i + (long) j * ldd] = transpose ? src[j + (long) i * lds] : src[i + (long) j * lds];
  }
}

template <class T>
void launch_copy2d(T* dst, long ldd, const T* src, long lds, int rows, int cols, int transpose, int mask,
                   hipStream_t stream) {
  if (rows <= 0 || cols <= 0)
    return;
  dim3 grid((unsigned) ((rows + 31) / 32), (unsigned) std::min(1024, (cols + 7) / 8));
  hipLaunchKernelGGL((copy2d_kernel<T>), grid, dim3(kThreads), 0, stream, dst, ldd, src, lds, rows, cols, transpose,
                     mask);
}

void update_kernels_init();
void trsm_kernels_init();
void potrf_kernels_init();
void potrf_coop_kernels_init();

void device_kernels_init() {
  update_kernels_init();
  trsm_kernels_init();
  potrf_kernels_init();
  potrf_coop_kernels_init();
}

#define INST(T)                                                                \
  template void launch_to_tiles<T>(const LayoutArgs<T>&, hipStream_t);         \
  template void launch_from_tiles<T>(const LayoutArgs<T>&, hipStream_t);     \
  template void launch_copy2d<T>(T*, long, const T*, long, int, int, int, int, hipStream_t);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
