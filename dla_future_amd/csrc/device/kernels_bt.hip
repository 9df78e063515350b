// kernels_bt.hip -- back-transformation band <- tridiagonal (stage 4), fused form.
//
// Reference: include/dlaf/eigensolver/bt_band_to_tridiag/impl.h:177-228 applies a block of b reflectors as two
// tile GEMMs, W2 = V^H E and E -= (V T) W2, per block and tile column.  The first MI355X version ran the same two
// products as strided-batch launches per wavefront of independent blocks: E read twice and written once per block,
// W2 through memory, the zero halves of the parallelogram V multiplied like the rest, short inner dimensions
// (K = b, 2b): 32-34 TFlop/s.
//
// Here one workgroup applies one block to a 64-column strip of E in one go:
//   * E is kept TRANSPOSED for the stage (et[c + r * ldet]): the 2b x 16 window of a wave is loaded ONCE, straight
//     into the fp64 MFMA accumulator map (row = g + 4v of a 16-row tile, column = lane & 15 -> the 16 lanes of a
//     group read 128 contiguous bytes), and that map IS the B-operand map of the next product (k = 4v + g): the
//     window is the operand of W2 = V^H E and the accumulator of E -= W W2 without ever moving;
//   * W2 (b x 16 per wave) lives in accumulators and is fed back as the B operand the same way;
//   * V^T (written in that form by the expansion kernel) and W stream through a ring of LDS stages in 16-row chunks
//     (direct-to-LDS loads, shared by the four waves of the workgroup, which own 16 columns each and never exchange
//     data);
//   * the 16 x 16 tiles of V (column c non-zero in rows [c, c + b)) and W = V T (rows [0, c + b)) that are zero by
//     construction are skipped: 688 instead of 1024 MFMAs per block and wave.
// E is read once and written once per block; nothing else goes through HBM (V and W of a block are shared by the
// workgroups of all strips, which run at the same time: L2).
#include "band_api.hpp"
#include "common.hpp"

namespace dlaf_mi355x {

namespace {

constexpr int kBtB = 128;                 // band size of the fused path
constexpr int kBtRows = 2 * kBtB;         // rows of a reflector block
constexpr int kBtRT = kBtRows / 16;       // row tiles of the window
constexpr int kBtCT = kBtB / 16;          // column tiles of V / W = tiles of W2
constexpr int kBtVtLd = kBtB + 16;        // LDS row stride of a V^H chunk (elements): 1152 B = 128 mod 256
constexpr int kBtStage = 16 * kBtVtLd;    // elements of a ring stage (a W chunk needs 16 * 128)
constexpr int kBtStages = 4;

// dst[c + r * ldd] = src[r + c * lds_] (rows x cols source, column-major) -- both directions of the stage's transposition
__global__ __launch_bounds__(256) void bt_transpose_kernel(const double* __restrict__ src, long lds_, long rows, long cols,
                                                           double* __restrict__ dst, long ldd) {
  __shared__ double tile[32][33];
  const long r0 = (long) blockIdx.x * 32, c0 = (long) blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int cc = ty; cc < 32; cc += 8)
    if (r0 + tx < rows && c0 + cc < cols)
      tile[cc][tx] = src[(r0 + tx) + (c0 + cc) * lds_];
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8)
    if (r0 + rr < rows && c0 + tx < cols)
      dst[(c0 + tx) + (r0 + rr) * ldd] = tile[tx][rr];
}

// One workgroup: block q = blockIdx.x / nstrips of the launch (V^T, b-element rows, at vt + q * blk_stride; W, column-major
// with 2b rows, at wr + q * blk_stride;
// rows [r0 + q * 2b, + rows) of E), strip blockIdx.x % nstrips (64 columns); wave w: 16 columns.
__global__ __launch_bounds__(256, 2) void bt_apply_kernel(const double* __restrict__ vt, const double* __restrict__ wr,
                                                          long blk_stride, double* __restrict__ et, long ldet, long ncols,
                                                          long r0_first, int rows, int nstrips) {
  using M = Mma<double>;
  using acc_t = M::acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char bt_lds_raw[];
  double* lds = reinterpret_cast<double*>(bt_lds_raw);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const long q = blockIdx.x / nstrips;
  const long strip = blockIdx.x % nstrips;
  const double* vtb = vt + q * blk_stride;
  const double* wrb = wr + q * blk_stride;
  const long r0 = r0_first + q * kBtRows;
  const long col = strip * 64 + wave * 16 + nl;
  const bool col_ok = col < ncols;
  // row bases are wave-uniform (scalar base + one per-lane offset for all 64 loads / stores of the window)
  double* const ebase = et + r0 * ldet;
  const long lane_off = col_ok ? col + (long) g * ldet : 0;

  // chunk t of the stream: t < kBtRT: rows [16 t, 16 t + 16) of V^H (one row of b elements = one 1 KiB
  // instruction, LDS rows padded); t >= kBtRT: row tile t - kBtRT of W, 16 KiB contiguous ([c][16 rows])
  auto issue = [&](int t) {
    double* dst = lds + (t % kBtStages) * kBtStage;
    if (t < kBtRT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 4 + i;
        const double* ga = vtb + (size_t) (t * 16 + row) * kBtB + lane * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) ga,
                                         (__attribute__((address_space(3))) void*) (dst + row * kBtVtLd), 16, 0, 0);
      }
    }
    else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // 1 KiB pieces of the 16 KiB chunk [c][16 rows]: eight columns of W (column-major, 2b rows) per piece,
        // 128 contiguous bytes of each
        const int piece = wave * 4 + i;
        const double* ga = wrb + (size_t) (piece * 8 + (lane >> 3)) * kBtRows + (t - kBtRT) * 16 + (lane & 7) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) ga,
                                         (__attribute__((address_space(3))) void*) (dst + piece * 128), 16, 0, 0);
      }
    }
  };

  // ---- the window, straight into the accumulator map -----------------------------------------------------
  // (whole blocks on whole strips -- all but the last block of a sweep group and the last strip -- take the
  // unmasked form: straight-line loads and stores)
  const bool whole = rows == kBtRows && strip * 64 + 64 <= ncols;
  acc_t win[kBtRT];
  if (whole) {
#pragma unroll
    for (int R = 0; R < kBtRT; ++R)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const double* rb = ebase + (long) (16 * R + 4 * v) * ldet;
        win[R][v] = rb[lane_off];
      }
  }
  else {
#pragma unroll
    for (int R = 0; R < kBtRT; ++R)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 16 * R + g + 4 * v;
        const double* rb = ebase + (long) (16 * R + 4 * v) * ldet;
        win[R][v] = (col_ok && r < rows) ? rb[lane_off] : 0.0;
      }
  }
#pragma unroll
  for (int t = 0; t < kBtStages - 1; ++t)
    issue(t);

  acc_t w2[kBtCT];
#pragma unroll
  for (int i = 0; i < kBtCT; ++i)
    w2[i] = acc_t{0.0, 0.0, 0.0, 0.0};

  constexpr int kChunks = 2 * kBtRT;
#pragma unroll
  for (int t = 0; t < kChunks; ++t) {
    // chunk t has landed (two younger chunks of 4 instructions each may still be in flight)
    if (t + 2 < kChunks)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 1 < kChunks)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // The slot re-staged below (chunk t + 3 -> slot (t - 1) % 4) was READ in the previous iteration: those ds_reads must
    // have RETURNED -- not merely been issued -- before any wave passes this barrier and issues LDS-DMA into the slot
    // (cdna_hip_programming.md, "restage a buffer >= 2 phases after its last ds_read, or 1 phase after when an lgkmcnt
    // before the barrier retired those reads").  Without this wait the kernel gave wrong eigenvectors in a few per cent
    // of runs as soon as four or more processes shared the GPU (LDS queues and the DMA path loaded differently: the
    // DMA landed before a delayed read) and never alone -- found in round 4 at N = 4096 on a 2 x 2 grid
    // (tools/diag_eig1.py; the round-3 grid cases were too small to hit it).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + kBtStages - 1 < kChunks)
      issue(t + kBtStages - 1);
    const double* st = lds + (t % kBtStages) * kBtStage;
    if (t < kBtRT) {
      // W2(tile i) += V^H(rows of tile i, window rows of tile R) E(tile R): V's tile (R, i) is non-zero for i <= R <= i + 8
      const int R = t;
      // (j outermost: consecutive MFMAs go to different accumulators)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < kBtCT; ++i) {
          if (i > R || R > i + kBtCT)
            continue;
          const double a = st[(4 * j + g) * kBtVtLd + 16 * i + nl];
          w2[i] = M::mma(a, win[R][j], w2[i]);
        }
      }
    }
    else {
      // E(tile R) -= W(rows of tile R, columns of tile i) W2(tile i): W's tile (R, i) is non-zero for R <= i + 8
      const int R = t - kBtRT;
      // two accumulation chains (even / odd k steps) instead of 32 dependent MFMAs in a row
      acc_t part = acc_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < kBtCT; ++i) {
        if (R > i + kBtCT)
          continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double a = st[(16 * i + 4 * j + g) * 16 + nl];
          if (j & 1)
            part = M::mma_neg(a, w2[i][j], part);
          else
            win[R] = M::mma_neg(a, w2[i][j], win[R]);
        }
      }
      win[R] += part;
    }
  }
  // ---- the window goes back ---------------------------------------------------------------------------------
  if (whole) {
#pragma unroll
    for (int R = 0; R < kBtRT; ++R)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double* rb = ebase + (long) (16 * R + 4 * v) * ldet;
        rb[lane_off] = win[R][v];
      }
  }
  else {
#pragma unroll
    for (int R = 0; R < kBtRT; ++R)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 16 * R + g + 4 * v;
        double* rb = ebase + (long) (16 * R + 4 * v) * ldet;
        if (col_ok && r < rows)
          rb[lane_off] = win[R][v];
      }
  }
}

}  // namespace

bool bt_fused_supported(int band, size_t elem_size, bool is_complex) {
  return band == kBtB && elem_size == sizeof(double) && !is_complex;
}

void launch_bt_transpose(const double* src, long lds_, long rows, long cols, double* dst, long ldd, hipStream_t stream) {
  if (rows <= 0 || cols <= 0)
    return;
  const long gy_max = 32768;
  for (long c0 = 0; c0 < cols; c0 += gy_max * 32) {
    const long cc = std::min<long>(gy_max * 32, cols - c0);
    hipLaunchKernelGGL(bt_transpose_kernel, dim3((unsigned) ((rows + 31) / 32), (unsigned) ((cc + 31) / 32)), dim3(256), 0,
                       stream, src + c0 * lds_, lds_, rows, cc, dst + c0, ldd);
  }
}

void launch_bt_apply(const double* vt, const double* wr, long blk_stride, int nblocks, double* et, long ldet, long ncols,
                     long r0_first, int rows, hipStream_t stream) {
  if (nblocks <= 0 || ncols <= 0 || rows <= 0)
    return;
  const int nstrips = (int) ((ncols + 63) / 64);
  hipLaunchKernelGGL(bt_apply_kernel, dim3((unsigned) ((long) nblocks * nstrips)), dim3(256),
                     kBtStages * kBtStage * sizeof(double), stream, vt, wr, blk_stride, et, ldet, ncols, r0_first, rows,
                     nstrips);
}

void bt_kernels_init() {
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&bt_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             kBtStages * kBtStage * sizeof(double));
}

}  // namespace dlaf_mi355x
