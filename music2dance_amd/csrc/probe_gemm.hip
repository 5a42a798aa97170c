// Test / bench-only entry points (NOT part of include/m2d.h): a plain fp32-MFMA GEMM kernel with nothing of the engine in it
// - no gather maps, no tails, no epilogue features - and a shader-clock read-out. Round-5 verdict item 6: the bench's
// `plain_gemm_4096_tflops` (the ENGINE on a plain shape) and the stand-alone probe's best kernel (tools/probes/
// gemm_ceiling.hip, "dl 128x128x16 dma16 frag1": 146.8 TFLOP/s) were measured in different processes; this is that kernel
// inside the library, timed by bench.py in the same process as the step, and it reports the clock it ran at.
//
//   C[M][N] = A^T B,  A stored [K][M] (K-major, like the engine's packed weight images), B stored [K][N];
//   M, N multiples of 128, K a multiple of 16. 128 x 128 x 16 tiles, four waves (2 x 2 of 64 x 64), two LDS stages filled
//   by 16-byte LDS-DMA, fragments of two neighbouring 32-row blocks interleaved (one 8-byte LDS read feeds two MFMAs).
#include "m2d_common.h"

typedef float pg_f32x16 __attribute__((ext_vector_type(16)));
typedef float pg_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float pg_lds_f;

// [0] s_memrealtime (100 MHz) and [1] s_memtime (shader clock) at kernel entry of workgroup 0, [2] / [3] at its end
__device__ unsigned long long m2d_probe_clock_buf[4];

__global__ void __launch_bounds__(256, 1) m2d_probe_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                 float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 16, STAGE = BK * (BM + BN), PF = 256;   // PF: floats per 16-byte DMA piece
  __shared__ __attribute__((aligned(16))) float sm[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, lh = lane >> 5;
  if (blockIdx.x == 0 && tid == 0) {
    m2d_probe_clock_buf[0] = __builtin_amdgcn_s_memrealtime();
    m2d_probe_clock_buf[1] = __builtin_amdgcn_s_memtime();
  }
  // workgroup -> tile: the XCD-aware grouped order of the engine's map (consecutive ids go round-robin to 8 XCDs)
  const int mt = M / BM, nt = N / BN;
  int by, bx;
  {
    const int T = mt * nt, q = T >> 3, r = T & 7, lin = blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per = 8 * nt, grp = id / per, rem = id - grp * per, first = grp * 8;
    const int gsz = mt - first < 8 ? mt - first : 8;
    bx = rem / gsz;
    by = first + rem - bx * gsz;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, (int)((size_t)M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, (short)0, (int)((size_t)N * K * 4), 0x00020000);
  unsigned offa[2], offb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = (wave + 4 * i) * PF + lane * 4, k = f / BM, r = f % BM;
    offa[i] = (unsigned)((k * M + m0 + r) * 4);
    offb[i] = (unsigned)((k * N + n0 + r) * 4);
  }
  auto stage = [&](float* st, int k0) {
    const int sa = k0 * M * 4, sb = k0 * N * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (pg_lds_f*)(st + (wave + 4 * i) * PF), 16, (int)offa[i], sa, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (pg_lds_f*)(st + BK * BM + (wave + 4 * i) * PF), 16, (int)offb[i], sb, 0, 0);
    }
  };
  pg_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  stage(sm, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nk = K / BK;
  for (int c = 0; c < nk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nk) stage(sm + (cur ^ 1) * STAGE, (c + 1) * BK);
    const float* as = sm + cur * STAGE + wm * 64;
    const float* bs = sm + cur * STAGE + BK * BM + wn * 64;
    pg_f32x2 fa[BK / 2], fb[BK / 2];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      fa[kk] = *reinterpret_cast<const pg_f32x2*>(as + (2 * kk + lh) * BM + 2 * l31);
      fb[kk] = *reinterpret_cast<const pg_f32x2*>(bs + (2 * kk + lh) * BN + 2 * l31);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // interleaved fragments: tile (i, j) of a wave holds rows 2 r' + i, columns 2 c' + j of its 64 x 64 block
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i;
      pg_f32x2 v = {acc[i][0][r], acc[i][1][r]};
      *reinterpret_cast<pg_f32x2*>(C + (size_t)row * N + n0 + wn * 64 + 2 * l31) = v;
    }
  if (blockIdx.x == 0 && tid == 0) {
    m2d_probe_clock_buf[2] = __builtin_amdgcn_s_memrealtime();
    m2d_probe_clock_buf[3] = __builtin_amdgcn_s_memtime();
  }
}

extern "C" {

// C = A^T B with A (K, M), B (K, N) row-major; M, N % 128 == 0, K % 16 == 0 (test / bench only)
int m2d_debug_probe_gemm(const float* a, const float* b, float* c, int M, int N, int K, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || M % 128 || N % 128 || K % 16 || (size_t)M * K * 4 >= 0x80000000ull ||
      (size_t)N * K * 4 >= 0x80000000ull)
    M2D_FAIL(M2D_ERR_ARG, "m2d_debug_probe_gemm: M, N multiples of 128, K of 16, operands below 2 GiB");
  hipLaunchKernelGGL(m2d_probe_gemm_kernel, dim3((M / 128) * (N / 128)), dim3(256), 0, (hipStream_t)stream, a, b, c, M, N, K);
  M2D_CHECK_LAUNCH("m2d_probe_gemm_kernel");
  return M2D_OK;
}

// out[0..3] = (s_memrealtime, s_memtime) at entry and exit of workgroup 0 of the last m2d_debug_probe_gemm launch
// (call after synchronising): shader clock in GHz = (out[3] - out[1]) / (out[2] - out[0]) / 10
int m2d_debug_probe_clock(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(m2d_probe_clock_buf), 4 * sizeof(unsigned long long)) == hipSuccess ? M2D_OK : M2D_ERR_HIP;
}

}  // extern "C"
