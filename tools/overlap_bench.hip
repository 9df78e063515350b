// overlap_bench.hip -- does the resident cooperative POTRF make progress beside a bulk update launch?
// Stream A: one long trailing-update launch (optionally persistent with reserved slots).
// Stream B: one cooperative tile POTRF, started right after.  Reports the POTRF's own duration alone
// and beside the update.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/overlap_bench.hip -o /tmp/ob && /tmp/ob [nt] [nb] [max_blocks]
#include "../dla_future_amd/csrc/device/kernels_update.hip"
#include "../dla_future_amd/csrc/device/kernels_potrf_coop.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace dlaf_mi355x;

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 32;
  const int nb = argc > 2 ? atoi(argv[2]) : 1024;
  const long max_blocks = argc > 3 ? atol(argv[3]) : 0;
  update_kernels_init();
  potrf_coop_kernels_init();
  const size_t te = (size_t) nb * nb;
  double *tiles, *dtile, *winv;
  int* info;
  unsigned *ctr, *sync;
  (void) hipMalloc(&tiles, sizeof(double) * te * nt * nt);
  (void) hipMalloc(&dtile, sizeof(double) * te);
  (void) hipMalloc(&winv, sizeof(double) * 64 * 64 * (nb / 64 + 1));
  (void) hipMalloc(&info, sizeof(int));
  (void) hipMalloc(&ctr, 8 * sizeof(unsigned));
  (void) hipMalloc(&sync, 2 * (nb / 64 + 2) * sizeof(unsigned));
  (void) hipMemset(info, 0, sizeof(int));
  std::vector<double> h(te * nt), d(te);
  srand(1);
  for (auto& v : h) v = (rand() / (double) RAND_MAX) * 2 - 1;
  for (int j = 0; j < nt; ++j)
    (void) hipMemcpy(tiles + te * nt * j, h.data(), sizeof(double) * te * nt, hipMemcpyHostToDevice);
  for (int j = 0; j < nb; ++j)
    for (int i = 0; i < nb; ++i) d[i + (size_t) j * nb] = (i == j) ? 2.0 * nb : 0.5 * (h[(i * 7 + j * 13) % h.size()] + 0.1);
  // symmetric diagonally dominant -> SPD
  for (int j = 0; j < nb; ++j)
    for (int i = 0; i < j; ++i) d[i + (size_t) j * nb] = d[j + (size_t) i * nb];
  UpdateArgs<double> ua;
  ua.c = tiles; ua.c_tsr = (long) te; ua.c_tsc = (long) te * nt; ua.ldc = nb;
  ua.a = tiles + te; ua.a_ts = (long) te; ua.lda = nb;
  ua.b = tiles + te; ua.b_ts = (long) te; ua.ldb = nb;
  ua.il0 = ua.jl0 = 1; ua.il1 = ua.jl1 = nt; ua.nb = nb; ua.K = nb;
  ua.pr = ua.pc = 1; ua.ri = ua.ci = 0; ua.nt = nt; ua.last_rows = nb; ua.info = info;
  hipStream_t sa, sb;
  (void) hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  (void) hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t a0, a1, b0, b1, go;
  (void) hipEventCreate(&a0); (void) hipEventCreate(&a1); (void) hipEventCreate(&b0); (void) hipEventCreate(&b1);
  (void) hipEventCreate(&go);
  auto potrf = [&](hipStream_t s) {
    (void) hipMemcpyAsync(dtile, d.data(), sizeof(double) * te, hipMemcpyHostToDevice, s);
    (void) hipEventRecord(b0, s);
    launch_potrf_coop(dtile, nb, nb, winv, info, 0, sync, s);
    (void) hipEventRecord(b1, s);
  };
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    potrf(sb);
    (void) hipDeviceSynchronize();
    (void) hipEventElapsedTime(&ms, b0, b1);
    printf("POTRF(%d) alone: %.3f ms\n", nb, ms);
  }
  for (int rep = 0; rep < 2; ++rep) {
    (void) hipMemcpy(dtile, d.data(), sizeof(double) * te, hipMemcpyHostToDevice);
    // a short kernel first so that both streams become runnable at the same instant (like ev_high)
    (void) hipEventRecord(a0, sa);
    launch_update(ua, sa, 0, max_blocks, ctr);
    (void) hipEventRecord(a1, sa);
    (void) hipEventRecord(b0, sb);
    launch_potrf_coop(dtile, nb, nb, winv, info, 0, sync, sb);
    (void) hipEventRecord(b1, sb);
    (void) hipDeviceSynchronize();
    float ma, mb, lag;
    (void) hipEventElapsedTime(&ma, a0, a1);
    (void) hipEventElapsedTime(&mb, b0, b1);
    (void) hipEventElapsedTime(&lag, a0, b1);
    printf("update(max_blocks=%ld): %.3f ms; POTRF beside it: %.3f ms (finished %.3f ms after the update started)\n",
           max_blocks, ma, mb, lag);
  }
  int hinfo;
  (void) hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost);
  printf("info=%d\n", hinfo);
  return 0;
}
