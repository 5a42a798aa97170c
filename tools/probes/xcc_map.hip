// Probe: which XCD (XCC_ID hardware register) does workgroup i of a 1-D grid run on, alone and with the chip busy?
//   hipcc --offload-arch=gfx950 -O2 -o xcc_map.bin xcc_map.hip && ./xcc_map.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(unsigned* out, int spin) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x & 0xf;
  // stay resident for a while so that later workgroups cannot reuse this CU
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
  const int n = 256;
  unsigned* d;
  hipMalloc(&d, n * sizeof(unsigned));
  std::vector<unsigned> h(n);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(who, dim3(n), dim3(512), 92 * 1024, 0, d, 2000);
    hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    int match = 0;
    for (int i = 0; i < n; ++i) match += (h[i] == (unsigned)(i % 8));
    printf("rep %d: xcc(i) == i %% 8 for %d of %d workgroups; first 24:", rep, match, n);
    for (int i = 0; i < 24; ++i) printf(" %u", h[i]);
    printf("\n");
  }
  return 0;
}
