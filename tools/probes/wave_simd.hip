// Dev probe: on which SIMD does wave i of a 512-thread workgroup run? (HW_REG_HW_ID: simd_id = bits 5:4 on gfx9)
// hipcc --offload-arch=gfx950 -O2 -o wave_simd.bin wave_simd.hip && ./wave_simd.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d;
  const int nb = 6;
  hipMalloc(&d, nb * 8 * 4);
  hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d);
  unsigned h[nb * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < nb; ++b) {
    printf("workgroup %d: wave -> simd:", b);
    for (int w = 0; w < 8; ++w) printf(" %d->%u", w, (h[b * 8 + w] >> 4) & 3);
    printf("   (cu %u se %u)\n", (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
  }
  return 0;
}
