// What do v_permlane32_swap_b32 / v_permlane16_swap_b32 and the DPP row controls do on gfx950?  (Lane maps printed; the
// wave reductions of the band_to_tridiagonal register kernel are built on them.)   hipcc --offload-arch=gfx950 -o permlane_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0];
  out[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = q[0];
  out[192 + l] = q[1];
  out[256 + l] = __builtin_amdgcn_update_dpp(999u, a, 0x141, 0xf, 0xf, false);  // row_half_mirror
  out[320 + l] = __builtin_amdgcn_update_dpp(999u, a, 0x140, 0xf, 0xf, false);  // row_mirror
  out[384 + l] = __builtin_amdgcn_update_dpp(999u, a, 0x128, 0xf, 0xf, false);  // row_ror:8
  out[448 + l] = __builtin_amdgcn_update_dpp(999u, a, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  out[512 + l] = __builtin_amdgcn_update_dpp(999u, a, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  out[576 + l] = __builtin_amdgcn_update_dpp(999u, a, 0x128, 0xf, 0x3, false);  // row_ror:8, banks 0-1 only
}
int main() {
  unsigned* d;
  hipMalloc(&d, 640 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[640];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"permlane32_swap [0] (vdst)", "permlane32_swap [1] (src)", "permlane16_swap [0]", "permlane16_swap [1]",
                         "row_half_mirror", "row_mirror", "row_ror:8", "quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]", "row_ror:8 bank_mask 0x3"};
  for (int t = 0; t < 10; ++t) {
    printf("%-28s:", names[t]);
    for (int l = 0; l < 64; ++l)
      printf(" %u", h[64 * t + l]);
    printf("\n");
  }
  return 0;
}
