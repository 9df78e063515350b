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
  const int il = blockIdx.y, jl = (int) blockIdx.z + p.jl_first;
  const int gi = il * p.pr + p.ri, gj = jl * p.pc + p.ci;
  if (!p.full && gi < gj)
    return;
  const bool diag = !p.full && (gi == gj);
  // element transform on the way: conjugation, and alpha * x into the tiles
  auto xf = [&](T v) -> T {
    if constexpr (TypeInfo<T>::is_complex) {
      if (p.conj)
        v.im = -v.im;
      if (TO_TILES && p.scale)
        v = T{v.re * p.alpha.re - v.im * p.alpha.im, v.re * p.alpha.im + v.im * p.alpha.re};
    }
    else {
      if (TO_TILES && p.scale)
        v = v * p.alpha;
    }
    return v;
  };
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
  // source position of view element (row, col): reversed axes count from the far end
  auto srow = [&](long vr) { return p.rev_rows ? p.rows - 1 - vr : vr; };
  auto scol = [&](long vc) { return p.rev_cols ? p.cols - 1 - vc : vc; };

  if (!p.transpose) {
    for (int cc = ty; cc < kLT; cc += kThreads / kLT) {
      const int r = r0 + tx, c = c0 + cc;
      if (r < rows_tile && c < cols_tile) {
        T* td = tile + r + (long) c * p.nb;
        T* cd = p.cm + srow(vrow0 + r) + scol(vcol0 + c) * p.ld_cm;
        if (TO_TILES)
          *td = xf(*cd);
        else if (!diag || r >= c)
          *cd = xf(*td);
      }
    }
  }
  else {
    // view(r, c) = cm(c, r): cm element (vcol0 + c) + (vrow0 + r) * ld -> contiguous along c
    if (TO_TILES) {
      for (int rr = ty; rr < kLT; rr += kThreads / kLT) {
        const int r = r0 + rr, c = c0 + tx;
        if (r < rows_tile && c < cols_tile)
          buf[rr][tx] = xf(p.cm[scol(vcol0 + c) + srow(vrow0 + r) * p.ld_cm]);
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
          buf[tx][cc] = xf(tile[r + (long) c * p.nb]);
      }
      __syncthreads();
      for (int rr = ty; rr < kLT; rr += kThreads / kLT) {
        const int r = r0 + rr, c = c0 + tx;
        if (r < rows_tile && c < cols_tile && (!diag || r >= c))
          p.cm[scol(vcol0 + c) + srow(vrow0 + r) * p.ld_cm] = buf[rr][tx];
      }
    }
  }
}

template <class T, bool TO_TILES>
static void launch_layout(const LayoutArgs<T>& a, hipStream_t stream) {
  if (a.ltr <= 0 || a.ltc <= 0 || a.nb <= 0)
    return;
  const int sb = (a.nb + kLT - 1) / kLT;
  const int ncols = a.jl_count > 0 ? a.jl_count : a.ltc - a.jl_first;
  if (ncols <= 0)
    return;
  dim3 grid((unsigned) (sb * sb), (unsigned) a.ltr, (unsigned) ncols);
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

// ---- tile transforms of gen_to_std (eigensolver/gen_to_std/impl.h): batches of square-ish tiles ----------
//   mode 0: dst = src^H                                  (rows x cols -> cols x rows), every element
//   mode 1: dst = scale * H, H the Hermitian matrix whose lower triangle is src's (imag(diag) taken as 0)
//   mode 2: lower(dst) = lower(src^H), strictly upper part of dst untouched, imag(diag) = 0
template <class T>
__global__ __launch_bounds__(kThreads) void tile_xform_kernel(T* dst, long ldd, long dstride, const T* src, long lds,
                                                              long sstride, int rows, int cols, int mode,
                                                              real_t<T> scale) {
  using R = real_t<T>;
  __shared__ T t[32][33];
  const int tile = blockIdx.z;
  const T* s = src + (long) tile * sstride;
  T* d = dst + (long) tile * dstride;
  const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // 32 x 8
  const int bi = blockIdx.x * 32, bj = blockIdx.y * 32;    // block origin in SRC coordinates (row bi, col bj)
  auto conj_of = [](T v) {
    if constexpr (TypeInfo<T>::is_complex)
      v.im = -v.im;
    return v;
  };
  if (mode == 1) {
    // element (i, j) of the full Hermitian image: from (i, j) if i >= j, else conj of (j, i); coalesced writes
    for (int jj = ty; jj < 32; jj += 8) {
      const int i = bi + tx, j = bj + jj;
      if (i < rows && j < cols) {
        T v = (i >= j) ? s[i + (long) j * lds] : conj_of(s[j + (long) i * lds]);
        if (i == j)
          v = make_el<T>(re_of(v), R(0));
        d[i + (long) j * ldd] = make_el<T>(scale * re_of(v), scale * im_of(v));
      }
    }
    return;
  }
  // modes 0 / 2: transpose through LDS (coalesced reads and writes)
  for (int jj = ty; jj < 32; jj += 8) {
    const int i = bi + tx, j = bj + jj;
    if (i < rows && j < cols)
      t[jj][tx] = s[i + (long) j * lds];
  }
  __syncthreads();
  for (int ii = ty; ii < 32; ii += 8) {
    // dst element (row = src col, col = src row)
    const int dr = bj + tx, dc = bi + ii;
    if (dr < cols && dc < rows) {
      T v = conj_of(t[tx][ii]);
      if (mode == 2) {
        if (dr < dc)
          continue;
        if (dr == dc)
          v = make_el<T>(re_of(v), R(0));
      }
      d[dr + (long) dc * ldd] = v;
    }
  }
}

// dst tile = alpha * op(src tile), op: 0 adjoint, 3 transpose, 4 conjugate, 5 copy (0 / 3: rows x cols -> cols x rows).
// The device-resident triangular solver builds its operand views with it (solver.cpp).
template <class T>
__global__ __launch_bounds__(kThreads) void tile_xform_alpha_kernel(T* dst, long ldd, long dstride, const T* src, long lds,
                                                                    long sstride, int rows, int cols, int mode, T alpha,
                                                                    int use_alpha) {
  __shared__ T t[32][33];
  const int tile = blockIdx.z;
  const T* s = src + (long) tile * sstride;
  T* d = dst + (long) tile * dstride;
  const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
  const int bi = blockIdx.x * 32, bj = blockIdx.y * 32;
  const bool cj = (mode == 0 || mode == 4);
  auto fin = [&](T v) {
    if constexpr (TypeInfo<T>::is_complex) {
      if (cj)
        v.im = -v.im;
      if (use_alpha)
        v = T{v.re * alpha.re - v.im * alpha.im, v.re * alpha.im + v.im * alpha.re};
    }
    else {
      if (use_alpha)
        v = v * alpha;
    }
    return v;
  };
  if (mode == 4 || mode == 5) {
    for (int jj = ty; jj < 32; jj += 8) {
      const int i = bi + tx, j = bj + jj;
      if (i < rows && j < cols)
        d[i + (long) j * ldd] = fin(s[i + (long) j * lds]);
    }
    return;
  }
  for (int jj = ty; jj < 32; jj += 8) {
    const int i = bi + tx, j = bj + jj;
    if (i < rows && j < cols)
      t[jj][tx] = s[i + (long) j * lds];
  }
  __syncthreads();
  for (int ii = ty; ii < 32; ii += 8) {
    const int dr = bj + tx, dc = bi + ii;
    if (dr < cols && dc < rows)
      d[dr + (long) dc * ldd] = fin(t[tx][ii]);
  }
}

template <class T>
void launch_tile_xform_alpha(T* dst, long ldd, long dstride, const T* src, long lds, long sstride, int rows, int cols,
                             int count, int mode, T alpha, bool use_alpha, hipStream_t stream) {
  if (rows <= 0 || cols <= 0 || count <= 0)
    return;
  dim3 grid((unsigned) ((rows + 31) / 32), (unsigned) ((cols + 31) / 32), (unsigned) count);
  hipLaunchKernelGGL((tile_xform_alpha_kernel<T>), grid, dim3(kThreads), 0, stream, dst, ldd, dstride, src, lds, sstride,
                     rows, cols, mode, alpha, use_alpha ? 1 : 0);
}

template <class T>
void launch_tile_xform(T* dst, long ldd, long dstride, const T* src, long lds, long sstride, int rows, int cols,
                       int count, int mode, double scale, hipStream_t stream) {
  if (rows <= 0 || cols <= 0 || count <= 0)
    return;
  dim3 grid((unsigned) ((rows + 31) / 32), (unsigned) ((cols + 31) / 32), (unsigned) count);
  hipLaunchKernelGGL((tile_xform_kernel<T>), grid, dim3(kThreads), 0, stream, dst, ldd, dstride, src, lds, sstride, rows,
                     cols, mode, (real_t<T>) scale);
}

// ---- checker helpers (miniapp/miniapp_cholesky.cpp:243-259 setUpperToZeroForDiagonalTiles,
// include/dlaf/auxiliary/norm/mc.h max_norm) -------------------------------------------------------
// one workgroup per local tile: max |a_ij| over the tiles with global row >= global column (lower part of
// diagonal tiles), folded into *out (non-negative doubles order like their bit patterns)
template <class T>
__global__ __launch_bounds__(kThreads) void max_norm_kernel(const T* tiles, int ltr, int nb, long rows, long cols,
                                                            int pr, int ri, int pc, int ci,
                                                            unsigned long long* out) {
  __shared__ double red[kThreads];
  const int il = blockIdx.x, jl = blockIdx.y;
  const int gi = il * pr + ri, gj = jl * pc + ci;
  double m = 0;
  if (gi >= gj) {
    const int rt = (int) min((long) nb, rows - (long) il * nb), ct = (int) min((long) nb, cols - (long) jl * nb);
    const T* t = tiles + ((long) il + (long) jl * ltr) * nb * nb;
    for (long idx = threadIdx.x; idx < (long) rt * ct; idx += kThreads) {
      const int r = (int) (idx % rt), c = (int) (idx / rt);
      if (gi > gj || r >= c) {
        const T v = t[r + (long) c * nb];
        const double a = TypeInfo<T>::is_complex ? hypot((double) re_of(v), (double) im_of(v)) : fabs((double) re_of(v));
        m = a > m ? a : m;  // NaN never wins a comparison; report it explicitly
        if (a != a)
          m = a;
      }
    }
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const double o = red[threadIdx.x + s];
      if (o > red[threadIdx.x] || o != o)
        red[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double v = red[0];
    if (v != v)
      v = 1e300;  // NaN -> "infinite" residual
    atomicMax(out, (unsigned long long) __double_as_longlong(v));
  }
}

template <class T>
__global__ __launch_bounds__(kThreads) void zero_upper_diag_kernel(T* tiles, int ltr, int ltc, int nb, int pr, int ri,
                                                                    int pc, int ci) {
  const int il = blockIdx.x;
  // the local column that holds the diagonal tile of local row il, if any
  const int gi = il * pr + ri;
  if ((gi - ci) % pc != 0 || gi < ci)
    return;
  const int jl = (gi - ci) / pc;
  if (jl >= ltc)
    return;
  T* t = tiles + ((long) il + (long) jl * ltr) * nb * nb;
  for (long idx = threadIdx.x; idx < (long) nb * nb; idx += kThreads) {
    const int r = (int) (idx % nb), c = (int) (idx / nb);
    if (r < c)
      t[r + (long) c * nb] = zero_el<T>();
  }
}

template <class T>
void launch_max_norm(const T* tiles, int ltr, int ltc, int nb, long rows, long cols, int pr, int ri, int pc, int ci,
                     double* out, hipStream_t stream) {
  (void) hipMemsetAsync(out, 0, sizeof(double), stream);
  if (ltr <= 0 || ltc <= 0)
    return;
  hipLaunchKernelGGL((max_norm_kernel<T>), dim3((unsigned) ltr, (unsigned) ltc), dim3(kThreads), 0, stream, tiles, ltr, nb,
                     rows, cols, pr, ri, pc, ci, reinterpret_cast<unsigned long long*>(out));
}

template <class T>
void launch_zero_upper_diag(T* tiles, int ltr, int ltc, int nb, int pr, int ri, int pc, int ci, hipStream_t stream) {
  if (ltr <= 0 || ltc <= 0)
    return;
  hipLaunchKernelGGL((zero_upper_diag_kernel<T>), dim3((unsigned) ltr), dim3(kThreads), 0, stream, tiles, ltr, ltc, nb, pr,
                     ri, pc, ci);
}

void update_kernels_init();
void trsm_kernels_init();
void potrf_kernels_init();
void potrf_coop_kernels_init();
void band_kernels_init();
void bt_kernels_init();
void hr_kernels_init();

void device_kernels_init() {
  update_kernels_init();
  trsm_kernels_init();
  potrf_kernels_init();
  potrf_coop_kernels_init();
  band_kernels_init();
  bt_kernels_init();
  hr_kernels_init();
}

#define INST(T)                                                                \
  template void launch_to_tiles<T>(const LayoutArgs<T>&, hipStream_t);         \
  template void launch_from_tiles<T>(const LayoutArgs<T>&, hipStream_t);     \
  template void launch_copy2d<T>(T*, long, const T*, long, int, int, int, int, hipStream_t);    \
  template void launch_tile_xform<T>(T*, long, long, const T*, long, long, int, int, int, int, double, hipStream_t); \
  template void launch_tile_xform_alpha<T>(T*, long, long, const T*, long, long, int, int, int, int, T, bool, hipStream_t); \
  template void launch_max_norm<T>(const T*, int, int, int, long, long, int, int, int, int, double*, hipStream_t); \
  template void launch_zero_upper_diag<T>(T*, int, int, int, int, int, int, int, hipStream_t);
INST(float)
INST(double)
INST(cfloat)
INST(cdouble)
#undef INST

}  // namespace dlaf_mi355x
