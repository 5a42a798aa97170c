// Dev probe: does `buffer_load_dwordx4 ... lds` accept 4-byte-aligned (not 16-byte-aligned) per-lane global addresses,
// and what does it cost? Lane i loads 16 bytes at base + shift_bytes + stride_bytes * i into LDS slot i.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) float lds_f;
__global__ void __launch_bounds__(256) k(const float* src, float* out, int n_bytes, int shift, int stride, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[4 * 64 * 4 * 4];  // 4 waves x 4 slots x 64 lanes x 16 B
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, n_bytes, 0x00020000);
  float acc = 0.f;
  const int base = (blockIdx.x * 4 + wave) * 64 * stride;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_f*)(sm + (wave * 4 + s) * 256), 16,
                                               (base + shift + stride * lane + s * 64 * stride * 1024) % (n_bytes - 64), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += sm[(wave * 4) * 256 + lane * 4 + (it & 3)];
  }
  if (iters == 1) {
    for (int e = 0; e < 4; ++e) out[(blockIdx.x * 256 + threadIdx.x) * 4 + e] = sm[(wave * 4) * 256 + lane * 4 + e];
  } else if (acc == 12345.f) out[0] = acc;
}
int main() {
  const int n = 1 << 26;  // 256 MB
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)(i % 100003);
  float *d, *o;
  hipMalloc(&d, (size_t)n * 4); hipMalloc(&o, 1 << 24);
  hipMemcpy(d, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  for (int stride : {16, 4}) for (int shift : {0, 4, 8, 12}) {
    hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, d, o, n * 4, shift, stride, 1);
    std::vector<float> r(8 * 256 * 4);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 8; ++b) for (int t = 0; t < 256; ++t) for (int e = 0; e < 4; ++e) {
      const int wave = t >> 6, lane = t & 63;
      const long long byte = ((long long)(b * 4 + wave) * 64 * stride + shift + (long long)stride * lane) % ((long long)n * 4 - 64);
      const float want = h[byte / 4 + e];
      if (r[(b * 256 + t) * 4 + e] != want) ++bad;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, o, n * 4, shift, stride, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, o, n * 4, shift, stride, 256);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2048.0 * 4 * 256 * 4 * 1024;  // blocks x waves x iters x pieces x 1 KiB
    printf("lane stride %2d B, shift %2d B: wrong words %d / %zu; %.1f GB/s of LDS-DMA issue (L2-resident)\n", stride, shift, bad,
           r.size(), bytes / ms / 1e6);
  }
  return 0;
}
