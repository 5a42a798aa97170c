// Dev probe: where does the dispatcher put the workgroups of a launch that does not fill the chip's resident slots?
// Every workgroup records XCC_ID / HW_ID and its start time, then stays resident for `hold_us`. Prints, per launch
// configuration, how many distinct CUs received work and the histogram of workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -o wg_placement.bin wg_placement.hip && ./wg_placement.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) probe(unsigned* ids, unsigned long long* t, int hold_ticks) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    ids[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
    ids[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    t[blockIdx.x] = t0;
  }
  lds[threadIdx.x] = (float)threadIdx.x;
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < hold_ticks) __builtin_amdgcn_s_sleep(8);
  if (lds[(threadIdx.x + 1) & 255] < 0.f) ids[0] = 0;
}

int main(int argc, char** argv) {
  const int hold_us = argc > 1 ? atoi(argv[1]) : 50;
  unsigned* ids; unsigned long long* t;
  hipMalloc(&ids, 8192 * 8); hipMalloc(&t, 8192 * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grids[] = {128, 180, 256, 360, 512, 720, 1024, 1200, 1280, 2400};
  const int ldss[] = {32 * 1024, 40 * 1024, 53 * 1024, 80 * 1024, 160 * 1024};
  for (int lds : ldss)
    for (int g : grids) {
      hipMemset(ids, 0, 8192 * 8);
      hipLaunchKernelGGL(probe, dim3(g), dim3(256), lds, 0, ids, t, hold_us * 100);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      std::vector<unsigned> h(2 * g); std::vector<unsigned long long> ht(g);
      hipMemcpy(h.data(), ids, g * 8, hipMemcpyDeviceToHost);
      hipMemcpy(ht.data(), t, g * 8, hipMemcpyDeviceToHost);
      unsigned long long tmin = *std::min_element(ht.begin(), ht.end());
      std::map<unsigned, int> per_cu, per_xcc; int late = 0;
      for (int b = 0; b < g; ++b) {
        const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if ((ht[b] - tmin) > 500) ++late;  // started more than 5 us after the first: had to wait for a slot
        else { per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; }
        per_xcc[xcc]++;
      }
      std::map<int, int> hist;
      for (auto& kv : per_cu) hist[kv.second]++;
      printf("lds %3d KB grid %4d: first wave on %3zu CUs, %4d late; workgroups per CU -> CUs:", lds / 1024, g, per_cu.size(), late);
      for (auto& kv : hist) printf(" %dx%d", kv.first, kv.second);
      printf(" | per XCD:");
      for (auto& kv : per_xcc) printf(" %d", kv.second);
      printf("\n");
    }
  return 0;
}
