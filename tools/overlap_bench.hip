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

__global__ void spin_kernel(int us) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long) us * 100ull)
    __builtin_amdgcn_s_sleep(32);
}
// where (and when, 100 MHz wall clock) does a workgroup of the side stream get a compute unit?
__global__ void where_kernel(unsigned* out) {
  extern __shared__ unsigned char wl[];
  if (threadIdx.x == 0) {
    unsigned xcc;
    const unsigned key = phys_cu_key(xcc);
    out[4 * blockIdx.x] = xcc;
    out[4 * blockIdx.x + 1] = key;
    out[4 * blockIdx.x + 2] = (unsigned) wall_clock64();
    wl[0] = 1;
  }
  for (int i = 0; i < 16; ++i)
    __builtin_amdgcn_s_sleep(64);
}

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 32;
  const int nb = argc > 2 ? atoi(argv[2]) : 1024;
  const long max_blocks = argc > 3 ? atol(argv[3]) : 0;
  const long excl_slots = argc > 4 ? atol(argv[4]) : 0;  // > 0: the launch vacates whole compute units (exclusive mode)
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
  (void) hipMalloc(&ctr, 16 * sizeof(unsigned));
  unsigned* where;
  (void) hipMalloc(&where, 4 * 64 * sizeof(unsigned));
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
    (void) hipMemsetAsync(ctr, 0, 16 * sizeof(unsigned), sa);
    (void) hipMemsetAsync(where, 0, 4 * 64 * sizeof(unsigned), sa);
    launch_update(ua, sa, 0, max_blocks, ctr, true, excl_slots);
    (void) hipEventRecord(a1, sa);
    // (the side stream starts a little later, as the panel work of the factorization does)
    if (argc > 5)
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sb, atoi(argv[5]));
    (void) hipEventRecord(b0, sb);
    hipLaunchKernelGGL(where_kernel, dim3(64), dim3(256), 60 * 1024, sb, where);
    launch_potrf_coop(dtile, nb, nb, winv, info, 0, sync, sb);
    (void) hipEventRecord(b1, sb);
    (void) hipDeviceSynchronize();
    {
      unsigned hc[16], hw[4 * 64];
      (void) hipMemcpy(hc, ctr, sizeof(hc), hipMemcpyDeviceToHost);
      (void) hipMemcpy(hw, where, sizeof(hw), hipMemcpyDeviceToHost);
      printf("workgroups that left: %u; queue heads:", hc[15]);
      for (int q = 0; q < 8; ++q) printf(" %u", hc[q]);
      printf("\nside-stream probe workgroups landed on (xcd:se.cu @ 100us units after the first):");
      unsigned t0 = ~0u;
      for (int i = 0; i < 64; ++i) t0 = hw[4 * i + 2] < t0 ? hw[4 * i + 2] : t0;
      for (int i = 0; i < 64; ++i)
        printf(" %u:%u.%u@%u", hw[4 * i], hw[4 * i + 1] >> 5, hw[4 * i + 1] & 15u, (hw[4 * i + 2] - t0) / 10000);
      printf("\n");
    }
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
