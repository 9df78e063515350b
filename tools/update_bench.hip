// update_bench.hip -- standalone timing of the grouped trailing-update kernel on a synthetic
// local matrix (one launch over all tiles below the first tile column), for kernel tuning.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/update_bench.hip -o /tmp/update_bench
//   /tmp/update_bench [nt=24] [nb=1024] [reps=3]
// Debug variants (compile-time): -DDLAF_DBG_SKIP_EPILOGUE  -DDLAF_DBG_SKIP_GLOBAL
#include "../dla_future_amd/csrc/device/kernels_update.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace dlaf_mi355x;
#ifndef UB_TYPE
#define UB_TYPE double
#endif
using TT = UB_TYPE;

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 24;
  const int nb = argc > 2 ? atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  const long max_blocks = argc > 4 ? atol(argv[4]) : 0;  // > 0: persistent grid of that many workgroups
  update_kernels_init();
  printf("resident workgroups per CU (occupancy API): %d\n", update_blocks_per_cu<TT>());
#ifndef UB_LDPAD
#define UB_LDPAD 0
#endif
  const int ld = nb + UB_LDPAD;  // leading dimension of a tile (UB_LDPAD != 0: no power-of-two column stride)
  const size_t te = (size_t) ld * nb;
  TT* tiles;
  int* info;
  (void) hipMalloc(&tiles, sizeof(TT) * te * nt * nt);
  (void) hipMalloc(&info, sizeof(int));
  (void) hipMemset(info, 0, sizeof(int));
  std::vector<double> h(te * nt * (sizeof(TT) / sizeof(double) > 0 ? sizeof(TT) / sizeof(double) : 1));
  srand(1);
  for (auto& v : h) v = (rand() / (double) RAND_MAX) * 2 - 1;
  for (int j = 0; j < nt; ++j)
    (void) hipMemcpy(tiles + te * nt * j, h.data(), sizeof(TT) * te * nt, hipMemcpyHostToDevice);
  UpdateArgs<TT> ua;
  ua.c = tiles;
  ua.c_tsr = (long) te;
  ua.c_tsc = (long) te * nt;
  ua.ldc = ld;
  ua.a = tiles + te;  // column 0, rows 1..
  ua.a_ts = (long) te;
  ua.lda = ld;
  ua.b = tiles + te;
  ua.b_ts = (long) te;
  ua.ldb = ld;
  ua.il0 = ua.jl0 = 1;
  ua.il1 = ua.jl1 = nt;
  ua.nb = nb;
  ua.K = nb;
  ua.pr = ua.pc = 1;
  ua.ri = ua.ci = 0;
  ua.nt = nt;
  ua.last_rows = nb;
  ua.info = info;
  const double t = nt - 1;
  const double cxf = TypeInfo<TT>::is_complex ? 4.0 : 1.0;
  const double flops = cxf * (t * (t - 1) / 2 * 2.0 * nb * nb * nb + t * (double) nb * (nb + 1) * nb);
  hipEvent_t e0, e1;
  (void) hipEventCreate(&e0);
  (void) hipEventCreate(&e1);
  unsigned* ctr;
  (void) hipMalloc(&ctr, 16 * sizeof(unsigned));
  launch_update(ua, nullptr, 0, max_blocks, ctr);
  (void) hipDeviceSynchronize();
  for (int r = 0; r < reps; ++r) {
    (void) hipEventRecord(e0);
    launch_update(ua, nullptr, 0, max_blocks, ctr);
    (void) hipEventRecord(e1);
    (void) hipDeviceSynchronize();
    float ms;
    (void) hipEventElapsedTime(&ms, e0, e1);
    printf("nt=%d nb=%d max_blocks=%ld: %.3f ms  %.2f TFlop/s\n", nt, nb, max_blocks, ms, flops / ms / 1e9);
  }
#ifdef DLAF_DBG_STAMPS
  {
    unsigned long long z[8] = {0}, h[8];
    (void) hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps), z, sizeof(z));
    hipEvent_t a0, a1;
    (void) hipEventCreate(&a0);
    (void) hipEventCreate(&a1);
    (void) hipEventRecord(a0);
    launch_update(ua, nullptr, 0, max_blocks, ctr);
    (void) hipEventRecord(a1);
    (void) hipDeviceSynchronize();
    float ms;
    (void) hipEventElapsedTime(&ms, a0, a1);
    (void) hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg_stamps), sizeof(h));
    const double tot = (double) (h[0] + h[1] + h[2] + h[3]);
    printf("stamps (instrumented run %.3f ms): wave-blocks %llu; per K-loop share: issue glds %.1f %%, ds_read+mfma %.1f %%, "
           "vmcnt/lgkm wait %.1f %%, barrier %.1f %%; mean cycles per wave-block %.0f\n",
           ms, h[4], 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, tot / (double) h[4]);
  }
#endif
  return 0;
}
