// Probe: round-trip latency of a data hand-off between two workgroups through global memory, same XCD (workgroups
// 0 and 8 of a 1-D grid) vs different XCDs (0 and 1), for several load cache policies. Stores are write-through
// device-scope (sc1) in all variants; the consumer spins on the value.
//   hipcc --offload-arch=gfx950 -O2 -o handoff_latency.bin handoff_latency.hip && ./handoff_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) unsigned gu32;

template <int MODE>
__device__ __forceinline__ unsigned ld(const unsigned* p) {
  unsigned v;
  if (MODE == 0) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == 2) asm volatile("buffer_inv sc0\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == 4) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st(unsigned* p, unsigned v) {
  __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroup `a` and workgroup `b` play ping-pong over x (a -> b) and y (b -> a); everybody else leaves
template <int MODE>
__global__ void pingpong(unsigned* x, unsigned* y, int a, int b, int rounds, long long* cycles, unsigned limit) {
  if (threadIdx.x != 0) return;
  const int id = blockIdx.x;
  if (id != a && id != b) return;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= rounds; ++i) {
    if (id == a) {
      st(x, (unsigned)i);
      unsigned n = 0;
      while (ld<MODE>(y) != (unsigned)i && ++n < limit) {}
      if (n >= limit) { cycles[1] = -i; return; }
    } else {
      unsigned n = 0;
      while (ld<MODE>(x) != (unsigned)i && ++n < limit) {}
      if (n >= limit) { cycles[1] = -i; return; }
      st(y, (unsigned)i);
    }
  }
  if (id == a) cycles[0] = wall_clock64() - t0;
}

template <int MODE>
static void run(const char* name, unsigned* x, unsigned* y, long long* cyc, int a, int b) {
  const int rounds = 2000;
  hipMemset(x, 0, 256); hipMemset(y, 0, 256); hipMemset(cyc, 0, 16);
  hipLaunchKernelGGL(pingpong<MODE>, dim3(16), dim3(64), 0, 0, x, y, a, b, rounds, cyc, 4000u);
  hipDeviceSynchronize();
  long long h[2];
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  if (h[1] < 0) printf("  %-28s workgroups %d,%d: STUCK at round %lld (stale line never refreshed)\n", name, a, b, -h[1]);
  else printf("  %-28s workgroups %d,%d: %.2f us per round trip (2 hand-offs)\n", name, a, b, h[0] / 100.0 / rounds);  // 100 MHz clock
}

int main() {
  unsigned *x, *y; long long* cyc;
  hipMalloc(&x, 4096); hipMalloc(&y, 4096); hipMalloc(&cyc, 64);
  y = x + 512;  // different lines of one allocation
  for (int pair = 0; pair < 2; ++pair) {
    const int a = 0, b = pair == 0 ? 8 : 1;
    printf("%s:\n", pair == 0 ? "same XCD" : "different XCDs");
    run<0>("load sc1", x, y, cyc, a, b);
    run<1>("load sc0", x, y, cyc, a, b);
    run<2>("buffer_inv sc0 + plain load", x, y, cyc, a, b);
    run<3>("load sc0 nt", x, y, cyc, a, b);
    run<4>("load nt", x, y, cyc, a, b);
    run<5>("load sc0 sc1", x, y, cyc, a, b);
  }
  return 0;
}
