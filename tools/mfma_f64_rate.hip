// mfma_f64_rate.hip -- measures the issue rate of v_mfma_f64_16x16x4_f64 on this GPU as a function
// of independent accumulators per wave and waves per SIMD.  Pins the "peak" of bench.py's roofline
// and the register-tile / occupancy choice of the update kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, long long* rt, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = seed * (1.0 + threadIdx.x * 1e-3), b = seed * (1.0 - threadIdx.x * 1e-3);
  long long r0 = wall_clock64();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)  // inline asm: keeps the accumulators in place (the builtin form made hipcc
                                    // shuffle them between AGPRs and VGPRs every iteration)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  long long t1 = clock64();
  long long r1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int NACC>
void run(int cus, int blocks_per_cu, double seed) {
  const int blocks = cus * blocks_per_cu;
  const int iters = 64000 / NACC;
  double* out; long long *cyc, *rt;
  (void) hipMalloc(&out, sizeof(double) * blocks * 256);
  (void) hipMalloc(&cyc, sizeof(long long) * blocks);
  (void) hipMalloc(&rt, sizeof(long long) * blocks);
  hipEvent_t e0, e1;
  (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(out, cyc, rt, 100, seed);
  (void) hipDeviceSynchronize();
  (void) hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, cyc, rt, iters, seed);
  (void) hipEventRecord(e1);
  (void) hipDeviceSynchronize();
  float ms; (void) hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks), hr(blocks);
  (void) hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  (void) hipMemcpy(hr.data(), rt, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0, avgr = 0;
  for (int i = 0; i < blocks; ++i) { avg += h[i]; avgr += hr[i]; }
  avg /= blocks; avgr /= blocks;
  const double mfmas_per_wave = (double) NACC * iters;
  const double flops = (double) blocks * 4 * mfmas_per_wave * 2048;
  // s_memrealtime ticks at 100 MHz: shader clock = memtime ticks / realtime ticks * 100 MHz
  printf("acc/wave=%2d waves/SIMD=%d data=%s: %.1f cyc/MFMA/wave, %.1f cyc/MFMA/SIMD, clock %.0f MHz, %.2f TFlop/s\n", NACC,
         blocks_per_cu, seed == 0 ? "zero" : "rand", avg / mfmas_per_wave, avg / mfmas_per_wave / blocks_per_cu,
         avg / avgr * 100.0, flops / ms / 1e9);
  (void) hipFree(out); (void) hipFree(cyc); (void) hipFree(rt);
}

int main() {
  hipDeviceProp_t p;
  (void) hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s CUs=%d nominal clock=%d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
  for (double seed : {0.7, 0.0}) {
    for (int w : {1, 2, 4}) {
      run<1>(cus, w, seed);
      run<2>(cus, w, seed);
      run<4>(cus, w, seed);
      run<8>(cus, w, seed);
      if (w <= 2) run<16>(cus, w, seed);
    }
  }
  return 0;
}
