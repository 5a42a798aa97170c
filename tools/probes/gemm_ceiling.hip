// Probe (round 4, verdict item 1): what does a plain fp32-MFMA GEMM reach on THIS part, next to the engine?
//   C[M][N] = sum_k A[k][M] * B[k][N]   (both operands row-fast: the engine's LDS-direct "TN" form, m2d_gemm mode 2)
// Variants, all v_mfma_f32_32x32x2_f32, timed interleaved in ONE process on uniform [-1, 1) operands:
//   guide   the guide's reference point (cdna_hip_programming.md "FP32-input MFMA"): 128x128x32 block, 4 waves,
//           2x2 tiles of 32x32 per wave, register staging (dwordx4 -> ds_write_b128), ONE LDS buffer, two barriers per
//           K-step, no software pipelining
//   dl      the engine's structure without its generality: LDS-DMA staging, two LDS stages, one barrier per K-step
//           parameters: block BM x BN, waves WGM x WGN, BK, DMA width (4 / 16 bytes), fragment form
//           (0: tile i = rows 32 i + lane, ds_read_b32; 1: tile i = rows TM * lane + i, ONE ds_read_b64 / b128 per
//           operand and k-step, and the tile leaves as 8- / 16-byte stores without an LDS transpose)
//   engine  m2d_gemm(mode 2) of libm2d_hip.so on the same operands
// Every dl variant runs under two workgroup -> tile maps: 0 = tile id = workgroup id, N fastest (the engine's order);
// 1 = XCD-aware (workgroups with equal id % 8 share an XCD: each such class gets a contiguous range of a grouped tile
// order, 8 M tiles x the N tiles, M fastest), so the tiles resident on one XCD share A and B panels in its L2.
// In-kernel clock (guide 'DVFS give-back' item 6): stamped builds of the same kernels, after >= 2 s of launches.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_ceiling.bin gemm_ceiling.hip -L../../music2dance_amd/lib
//          -lm2d_hip -Wl,-rpath,'$ORIGIN/../../music2dance_amd/lib'
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int m2d_gemm(int mode, const float* a, const float* b, const float* bias, float* c, int M, int N, int K, int act,
                        float slope, const float* a_mask, float a_mask_slope, const float* out_mask, float out_mask_slope,
                        void* ws, size_t ws_bytes, void* stream);
extern "C" size_t m2d_gemm_workspace_bytes(int mode, int M, int N, int K);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f;

struct Args {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int map;            // 0 plain, 1 XCD-aware grouped
  unsigned long long* stamps;  // stamped builds: [workgroup][4] = memtime0, realtime0, memtime1, realtime1
};

__device__ __forceinline__ void tile_of(const Args& a, int BM, int BN, int& m0, int& n0) {
  const int nt = a.N / BN, mt = a.M / BM;
  int id = blockIdx.x;
  if (a.map == 1) {
    const int T = nt * mt, q = T >> 3, r = T & 7;
    const int xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = 8;
    const int per = GM * nt;
    const int grp = id / per, rem = id - grp * per;
    const int first = grp * GM;
    const int gsz = mt - first < GM ? mt - first : GM;
    m0 = (first + rem % gsz) * BM;
    n0 = (rem / gsz) * BN;
    return;
  }
  m0 = (id / nt) * BM;
  n0 = (id % nt) * BN;
}

template <bool STAMP>
__device__ __forceinline__ void stamp(const Args& a, int which) {
  if constexpr (STAMP) {
    if (threadIdx.x == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      const unsigned long long r = __builtin_amdgcn_s_memrealtime();
      a.stamps[(size_t)blockIdx.x * 8 + which * 2] = t;
      a.stamps[(size_t)blockIdx.x * 8 + which * 2 + 1] = r;
    }
  }
}
template <bool STAMP>
__device__ __forceinline__ void stamp_rt(const Args& a, int slot) {   // realtime only: kernel entry (4), after the stores (5)
  if constexpr (STAMP) {
    if (threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---- the guide's reference point ------------------------------------------------------------------------------------
template <bool STAMP>
__global__ void __launch_bounds__(256) k_guide(const Args a) {
  constexpr int BM = 128, BN = 128, BK = 32;
  __shared__ __attribute__((aligned(16))) float sm[BK * (BM + BN)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, lh = lane >> 5;
  int m0, n0;
  tile_of(a, BM, BN, m0, n0);
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  stamp<STAMP>(a, 0);
  for (int k0 = 0; k0 < a.K; k0 += BK) {
    // 32 x 128 floats per operand = 1024 float4: four per thread
    f32x4 va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = (tid + 256 * i) * 4, k = f / BM, r = f % BM;
      va[i] = *reinterpret_cast<const f32x4*>(a.A + (size_t)(k0 + k) * a.M + m0 + r);
      vb[i] = *reinterpret_cast<const f32x4*>(a.B + (size_t)(k0 + k) * a.N + n0 + r);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = (tid + 256 * i) * 4;
      *reinterpret_cast<f32x4*>(sm + f) = va[i];
      *reinterpret_cast<f32x4*>(sm + BK * BM + f) = vb[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = sm[(2 * kk + lh) * BM + wm * 64 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = sm[BK * BM + (2 * kk + lh) * BN + wn * 64 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  stamp<STAMP>(a, 1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        a.C[(size_t)row * a.N + n0 + wn * 64 + j * 32 + l31] = acc[i][j][r];
      }
}

template <int DMA>
__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, int soff) {
  if constexpr (DMA == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_f*)dst, 16, (int)voff, soff, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_f*)dst, 4, (int)voff, soff, 0, 0);
}

// ---- LDS-DMA, two stages, one barrier per K-step ----------------------------------------------------------------------
// MINW: second __launch_bounds__ argument (waves per SIMD the register allocator must leave room for; 2 is what the engine
// declares - hipcc then keeps the MFMA accumulators in arch VGPRs, with 1 it puts them into AGPRs). PIN: the engine's
// pinned software pipeline (all fragment reads of the chunk named up front, sched_group_barrier: reads of k-step kk + 2
// under the MFMAs of k-step kk) instead of the compiler's own order.
template <int BM, int BN, int WGM, int WGN, int BK, int DMA, int FRAG, bool STAMP, int MINW = 1, bool PIN = false>
__global__ void __launch_bounds__(64 * WGM * WGN, MINW) k_dl(const Args a) {
  constexpr int NT = 64 * WGM * WGN, NW = WGM * WGN;
  constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
  constexpr int STAGE = BK * (BM + BN);
  static_assert(FRAG == 0 || ((TM == 2 || TM == 4) && (TN == 2 || TN == 4)), "interleaved fragments: 2 or 4 tiles per side");
  __shared__ __attribute__((aligned(16))) float sm[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int l31 = lane & 31, lh = lane >> 5;
  stamp_rt<STAMP>(a, 4);
  int m0, n0;
  tile_of(a, BM, BN, m0, n0);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, (short)0, (int)((size_t)a.M * a.K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, (short)0, (int)((size_t)a.N * a.K * 4), 0x00020000);
  // a DMA piece = 64 lanes x DMA bytes of the [k][row] image, contiguous in LDS: piece p of operand X covers image floats
  // [p * 16 DMA, (p + 1) * 16 DMA); wave w takes pieces w, w + NW, ...
  constexpr int PF = 16 * DMA;                 // floats per piece
  constexpr int PA = BK * BM / PF, PB = BK * BN / PF;
  static_assert(PA % NW == 0 && PB % NW == 0, "pieces divide over the waves");
  unsigned offa[PA / NW], offb[PB / NW];
#pragma unroll
  for (int i = 0; i < PA / NW; ++i) {
    const int f = (wave + NW * i) * PF + lane * (DMA / 4), k = f / BM, r = f % BM;
    offa[i] = (unsigned)((k * a.M + m0 + r) * 4);
  }
#pragma unroll
  for (int i = 0; i < PB / NW; ++i) {
    const int f = (wave + NW * i) * PF + lane * (DMA / 4), k = f / BN, r = f % BN;
    offb[i] = (unsigned)((k * a.N + n0 + r) * 4);
  }
  auto stage = [&](float* st, int k0) {
    const int sa = k0 * a.M * 4, sb = k0 * a.N * 4;
#pragma unroll
    for (int i = 0; i < PA / NW; ++i)
      dma<DMA>(ra, st + (wave + NW * i) * PF, offa[i], sa);
#pragma unroll
    for (int i = 0; i < PB / NW; ++i)
      dma<DMA>(rb, st + BK * BM + (wave + NW * i) * PF, offb[i], sb);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  stamp<STAMP>(a, 0);
  stage(sm, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nk = a.K / BK;
  for (int c = 0; c < nk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nk) stage(sm + (cur ^ 1) * STAGE, (c + 1) * BK);
    const float* as = sm + cur * STAGE + wm * (TM * 32);
    const float* bs = sm + cur * STAGE + BK * BM + wn * (TN * 32);
    if constexpr (FRAG == 0) {
      float fa[BK / 2][TM], fb[BK / 2][TN];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[kk][i] = as[(2 * kk + lh) * BM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kk][j] = bs[(2 * kk + lh) * BN + j * 32 + l31];
      }
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
      if constexpr (PIN) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
          __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
          if (kk < BK / 2 - 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
    } else {
      typedef float fva __attribute__((ext_vector_type(TM)));
      typedef float fvb __attribute__((ext_vector_type(TN)));
      fva fa[BK / 2];
      fvb fb[BK / 2];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        fa[kk] = *reinterpret_cast<const fva*>(as + (2 * kk + lh) * BM + TM * l31);
        fb[kk] = *reinterpret_cast<const fvb*>(bs + (2 * kk + lh) * BN + TN * l31);
      }
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  stamp<STAMP>(a, 1);
  if constexpr (FRAG == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          a.C[(size_t)row * a.N + n0 + wn * (TN * 32) + j * 32 + l31] = acc[i][j][r];
        }
  } else {
    typedef float fvb __attribute__((ext_vector_type(TN)));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (TM * 32) + TM * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i;
        fvb v;
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
        *reinterpret_cast<fvb*>(a.C + (size_t)row * a.N + n0 + wn * (TN * 32) + TN * l31) = v;
      }
  }
  if constexpr (STAMP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp_rt<STAMP>(a, 5);
}


// ---- LDS-DMA (16-byte), THREE stages, counted vmcnt: chunk c + 2 stays in flight across the barrier -------------------
// frag1 only. Fragment reads through inline asm (hipcc would put `s_waitcnt vmcnt(0)` in front of a ds_read while an
// LDS-DMA is outstanding: it cannot tell that the DMA fills another stage), all 2 * BK / 2 reads of the chunk issued up
// front, counted lgkmcnt per k-step.
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N > 15 ? 15 : N) : "memory"); }
__device__ __forceinline__ f32x2 lds_read_b64(unsigned addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int BM, int BN, int WGM, int WGN, int BK, int NST, bool STAMP, int MINW = 1>
__global__ void __launch_bounds__(64 * WGM * WGN, MINW) k_dl3(const Args a) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
  static_assert(TM == 2 && TN == 2, "k_dl3: 2 x 2 tiles per wave");
  constexpr int STAGE = BK * (BM + BN);
  __shared__ __attribute__((aligned(16))) float sm[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int l31 = lane & 31, lh = lane >> 5;
  int m0, n0;
  tile_of(a, BM, BN, m0, n0);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, (short)0, (int)((size_t)a.M * a.K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, (short)0, (int)((size_t)a.N * a.K * 4), 0x00020000);
  constexpr int PF = 256;
  constexpr int PA = BK * BM / PF, PB = BK * BN / PF;
  constexpr int NLD = PA / NW + PB / NW;   // DMA instructions per wave and chunk
  unsigned offa[PA / NW], offb[PB / NW];
#pragma unroll
  for (int i = 0; i < PA / NW; ++i) {
    const int f = (wave + NW * i) * PF + lane * 4, k = f / BM, r = f % BM;
    offa[i] = (unsigned)((k * a.M + m0 + r) * 4);
  }
#pragma unroll
  for (int i = 0; i < PB / NW; ++i) {
    const int f = (wave + NW * i) * PF + lane * 4, k = f / BN, r = f % BN;
    offb[i] = (unsigned)((k * a.N + n0 + r) * 4);
  }
  auto stage = [&](float* st, int k0) {
    const int sa = k0 * a.M * 4, sb = k0 * a.N * 4;
#pragma unroll
    for (int i = 0; i < PA / NW; ++i) dma<16>(ra, st + (wave + NW * i) * PF, offa[i], sa);
#pragma unroll
    for (int i = 0; i < PB / NW; ++i) dma<16>(rb, st + BK * BM + (wave + NW * i) * PF, offb[i], sb);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  stamp<STAMP>(a, 0);
  const int nk = a.K / BK;
  // prologue: NST - 1 chunks in flight, chunk 0 landed
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage(sm + s * STAGE, s * BK);   // (K >= (NST - 1) BK)
  wait_vm<(NST - 2) * NLD>();
  __builtin_amdgcn_s_barrier();
  const unsigned sm0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)sm;
  const unsigned aoff = (unsigned)((lh * BM + wm * (TM * 32) + TM * l31) * 4);
  const unsigned boff = (unsigned)((BK * BM + lh * BN + wn * (TN * 32) + TN * l31) * 4);
  int cur = 0, nxt = NST - 1;
  for (int c = 0; c < nk; ++c) {
    // chunk c + NST - 1 into the stage every wave left at the barrier that ended chunk c - 1 (past the end: re-load the
    // last chunk, harmlessly, so that the counted waits stay uniform)
    const int cn = c + NST - 1 < nk ? c + NST - 1 : nk - 1;
    stage(sm + nxt * STAGE, cn * BK);
    const unsigned base = sm0 + (unsigned)(cur * STAGE * 4);
    f32x2 fa[BK / 2], fb[BK / 2];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      fa[kk] = lds_read_b64(base + aoff + (unsigned)(2 * kk * BM * 4));
      fb[kk] = lds_read_b64(base + boff + (unsigned)(2 * kk * BN * 4));
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      switch (kk) {   // reads return in order: 2 (kk + 1) of the BK issued must be back
        case 0: wait_lgkm<BK - 2>(); break;
        case 1: wait_lgkm<BK - 4>(); break;
        case 2: wait_lgkm<BK - 6>(); break;
        case 3: wait_lgkm<BK - 8>(); break;
        case 4: wait_lgkm<BK - 10>(); break;
        case 5: wait_lgkm<BK - 12>(); break;
        case 6: wait_lgkm<BK - 14>(); break;
        case 7: wait_lgkm<(BK > 16 ? BK - 16 : 0)>(); break;
        case 8: wait_lgkm<(BK > 18 ? BK - 18 : 0)>(); break;
        case 9: wait_lgkm<(BK > 20 ? BK - 20 : 0)>(); break;
        case 10: wait_lgkm<(BK > 22 ? BK - 22 : 0)>(); break;
        case 11: wait_lgkm<(BK > 24 ? BK - 24 : 0)>(); break;
        case 12: wait_lgkm<(BK > 26 ? BK - 26 : 0)>(); break;
        case 13: wait_lgkm<(BK > 28 ? BK - 28 : 0)>(); break;
        case 14: wait_lgkm<(BK > 30 ? BK - 30 : 0)>(); break;
        default: wait_lgkm<0>(); break;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
    }
    // chunk c + 1 landed (all but the youngest (NST - 2) chunks' loads are done), every wave is done reading chunk c
    wait_vm<(NST - 2) * NLD>();
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
  wait_vm<0>();
  stamp<STAMP>(a, 1);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * (TM * 32) + TM * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i;
      f32x2 v;
#pragma unroll
      for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
      *reinterpret_cast<f32x2*>(a.C + (size_t)row * a.N + n0 + wn * (TN * 32) + TN * l31) = v;
    }
}

// ---- harness ----------------------------------------------------------------------------------------------------------
struct Variant {
  std::string name;
  void (*launch)(const Args&, bool stamped, hipStream_t);
  int bm, bn, map;
  bool engine;
  int (*stamps)(unsigned long long*, int) = nullptr;
};

template <int BM, int BN, int WGM, int WGN, int BK, int DMA, int FRAG, int MINW = 1, bool PIN = false>
static void launch_dl(const Args& a, bool stamped, hipStream_t s) {
  const dim3 grid((a.M / BM) * (a.N / BN)), block(64 * WGM * WGN);
  if (stamped) hipLaunchKernelGGL((k_dl<BM, BN, WGM, WGN, BK, DMA, FRAG, true, MINW, PIN>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_dl<BM, BN, WGM, WGN, BK, DMA, FRAG, false, MINW, PIN>), grid, block, 0, s, a);
}
template <int BM, int BN, int WGM, int WGN, int BK, int NST, int MINW = 1>
static void launch_dl3(const Args& a, bool stamped, hipStream_t s) {
  const dim3 grid((a.M / BM) * (a.N / BN)), block(64 * WGM * WGN);
  if (stamped) hipLaunchKernelGGL((k_dl3<BM, BN, WGM, WGN, BK, NST, true, MINW>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_dl3<BM, BN, WGM, WGN, BK, NST, false, MINW>), grid, block, 0, s, a);
}
static void launch_guide(const Args& a, bool stamped, hipStream_t s) {
  const dim3 grid((a.M / 128) * (a.N / 128)), block(256);
  if (stamped) hipLaunchKernelGGL((k_guide<true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_guide<false>), grid, block, 0, s, a);
}
static void* g_ws = nullptr;
static size_t g_ws_bytes = 0;
static void launch_engine(const Args& a, bool, hipStream_t s) {
  const int rc = m2d_gemm(2, a.A, a.B, nullptr, a.C, a.M, a.N, a.K, 0, 0.f, nullptr, 0.f, nullptr, 0.f, g_ws, g_ws_bytes, (void*)s);
  if (rc) { fprintf(stderr, "m2d_gemm failed: %d\n", rc); exit(1); }
}

// experiment builds of the engine (M2D_PROBE_LIBS = colon-separated paths of other libm2d_hip builds): m2d_gemm of each
typedef int (*gemm_fn)(int, const float*, const float*, const float*, float*, int, int, int, int, float, const float*, float,
                       const float*, float, void*, size_t, void*);
static gemm_fn g_xfn[4];
template <int I>
static void launch_engine_x(const Args& a, bool, hipStream_t s) {
  const int rc = g_xfn[I](2, a.A, a.B, nullptr, a.C, a.M, a.N, a.K, 0, 0.f, nullptr, 0.f, nullptr, 0.f, g_ws, g_ws_bytes, (void*)s);
  if (rc) { fprintf(stderr, "m2d_gemm (variant %d) failed: %d\n", I, rc); exit(1); }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  const int rounds = argc > 4 ? atoi(argv[4]) : 7;
  std::vector<float> hA((size_t)K * M), hB((size_t)K * N);
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) / 8388608.0 - 1.0); };
  for (auto& x : hA) x = rnd();
  for (auto& x : hB) x = rnd();
  float *dA, *dB, *dC, *dRef;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)M * N * 4); hipMalloc(&dRef, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  g_ws_bytes = m2d_gemm_workspace_bytes(2, M, N, K);
  if (g_ws_bytes) hipMalloc(&g_ws, g_ws_bytes);
  unsigned long long* dStamps;
  const size_t max_wg = (size_t)(M / 64) * (N / 64);
  hipMalloc(&dStamps, max_wg * 8 * sizeof(unsigned long long));

  std::vector<Variant> v;
  v.push_back({"engine m2d_gemm(mode 2)", launch_engine, 128, 128, 0, true});
  if (const char* libs = getenv("M2D_PROBE_LIBS")) {
    std::string all(libs);
    size_t pos = 0;
    int n = 0;
    void (*lx[4])(const Args&, bool, hipStream_t) = {launch_engine_x<0>, launch_engine_x<1>, launch_engine_x<2>, launch_engine_x<3>};
    while (pos <= all.size() && n < 4) {
      const size_t e = all.find(':', pos);
      const std::string path = all.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
      pos = e == std::string::npos ? all.size() + 1 : e + 1;
      if (path.empty()) continue;
      void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
      if (!h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); return 1; }
      g_xfn[n] = (gemm_fn)dlsym(h, "m2d_gemm");
      v.push_back({"engine build " + path.substr(path.rfind('/') + 1), lx[n], 128, 128, 0, true});
      v.back().stamps = (int (*)(unsigned long long*, int))dlsym(h, "m2d_debug_stamps");
      ++n;
    }
  }
  v.push_back({"guide 128x128x32 reg-staged, no pipelining", launch_guide, 128, 128, 0, false});
  const int set = argc > 5 ? atoi(argv[5]) : 1;
  if (set == 0) {
    for (int map = 0; map < 2; ++map) {
      const std::string ms = map ? " xcd" : "";
      v.push_back({"dl 128x128x16 2x2 dma4 frag0" + ms, launch_dl<128, 128, 2, 2, 16, 4, 0>, 128, 128, map, false});
      v.push_back({"dl 128x128x16 2x2 dma16 frag0" + ms, launch_dl<128, 128, 2, 2, 16, 16, 0>, 128, 128, map, false});
      v.push_back({"dl 128x128x16 2x2 dma16 frag1" + ms, launch_dl<128, 128, 2, 2, 16, 16, 1>, 128, 128, map, false});
      v.push_back({"dl 128x128x32 2x2 dma16 frag1" + ms, launch_dl<128, 128, 2, 2, 32, 16, 1>, 128, 128, map, false});
      v.push_back({"dl 256x128x16 2x2 dma16 frag1" + ms, launch_dl<256, 128, 2, 2, 16, 16, 1>, 256, 128, map, false});
      v.push_back({"dl 256x128x16 4x2 dma16 frag1 (8 waves)" + ms, launch_dl<256, 128, 4, 2, 16, 16, 1>, 256, 128, map, false});
      v.push_back({"dl 256x256x16 4x2 dma16 frag1 (8 waves)" + ms, launch_dl<256, 256, 4, 2, 16, 16, 1>, 256, 256, map, false});
      v.push_back({"dl 256x256x16 2x2 dma16 frag1 (4 waves)" + ms, launch_dl<256, 256, 2, 2, 16, 16, 1>, 256, 256, map, false});
    }
  } else if (set == 3) {
    // round 5: eight waves per workgroup on the SAME 128x128x16 tile (two waves per SIMD from one workgroup: 4 x 32 KB of
    // LDS is what a CU holds - the allocation granule is 1280 B, so five 32 768-byte workgroups do not fit - i.e. four
    // waves per SIMD today, eight with this form), against the engine's four-wave scheme; on shapes given as M N K
    v.push_back({"dl 128x128x16 2x2 dma4 frag0 pinned (4 waves)", launch_dl<128, 128, 2, 2, 16, 4, 0, 2, true>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 2x2 dma16 frag1 (4 waves)", launch_dl<128, 128, 2, 2, 16, 16, 1, 2>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 2x4 dma4 frag0 (8 waves, 64x32 each)", launch_dl<128, 128, 2, 4, 16, 4, 0, 2>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 2x4 dma4 frag0 pinned (8 waves)", launch_dl<128, 128, 2, 4, 16, 4, 0, 2, true>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 4x2 dma4 frag0 (8 waves, 32x64 each)", launch_dl<128, 128, 4, 2, 16, 4, 0, 2>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 2x4 dma16 frag0 (8 waves)", launch_dl<128, 128, 2, 4, 16, 16, 0, 2>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 2x4 dma16 frag0 (8 waves) xcd", launch_dl<128, 128, 2, 4, 16, 16, 0, 2>, 128, 128, 1, false});
    v.push_back({"dl 128x64x16 2x2 dma4 frag0 (4 waves, 24 KB)", launch_dl<128, 64, 2, 2, 16, 4, 0, 2>, 128, 64, 0, false});
    v.push_back({"dl3 128x128x16 3 stages (4 waves)", launch_dl3<128, 128, 2, 2, 16, 3>, 128, 128, 0, false});
  } else if (set == 2) {
    v.push_back({"dl 128x128x16 dma4 frag0 minw2 + pinned", launch_dl<128, 128, 2, 2, 16, 4, 0, 2, true>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 dma16 frag1 minw2", launch_dl<128, 128, 2, 2, 16, 16, 1, 2>, 128, 128, 0, false});
  } else {
    // set 1: what separates the engine's LDS-direct kernel (same tile, same DMA count) from the plain dl kernel?
    v.push_back({"dl 128x128x16 dma4 frag0 (AGPR acc)", launch_dl<128, 128, 2, 2, 16, 4, 0>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 dma4 frag0 minw2 (VGPR acc)", launch_dl<128, 128, 2, 2, 16, 4, 0, 2>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 dma4 frag0 pinned pipeline", launch_dl<128, 128, 2, 2, 16, 4, 0, 1, true>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 dma4 frag0 minw2 + pinned", launch_dl<128, 128, 2, 2, 16, 4, 0, 2, true>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 dma16 frag1", launch_dl<128, 128, 2, 2, 16, 16, 1>, 128, 128, 0, false});
    v.push_back({"dl 128x128x16 dma16 frag1 minw2", launch_dl<128, 128, 2, 2, 16, 16, 1, 2>, 128, 128, 0, false});
    v.push_back({"dl3 128x128x16 2 stages (asm reads, counted)", launch_dl3<128, 128, 2, 2, 16, 2>, 128, 128, 0, false});
    v.push_back({"dl3 128x128x16 3 stages", launch_dl3<128, 128, 2, 2, 16, 3>, 128, 128, 0, false});
    v.push_back({"dl3 128x128x16 4 stages", launch_dl3<128, 128, 2, 2, 16, 4>, 128, 128, 0, false});
    v.push_back({"dl3 128x128x32 3 stages", launch_dl3<128, 128, 2, 2, 32, 3>, 128, 128, 0, false});
    v.push_back({"dl3 128x128x16 3 stages xcd", launch_dl3<128, 128, 2, 2, 16, 3>, 128, 128, 1, false});
  }
  // correctness: every variant against the guide kernel (same k order per output: bitwise equal is expected, report max diff),
  // and the guide kernel against fp64 on sampled entries
  Args a{dA, dB, dRef, M, N, K, 0, dStamps};
  launch_guide(a, false, 0);
  hipDeviceSynchronize();
  std::vector<float> ref((size_t)M * N), out((size_t)M * N);
  hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 64; ++s) {
    const int i = (s * 977 + 13) % M, j = (s * 1613 + 7) % N;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)k * M + i] * hB[(size_t)k * N + j];
    worst = std::max(worst, std::fabs(acc - ref[(size_t)i * N + j]));
  }
  printf("guide kernel vs fp64 on 64 entries: max abs err %.3g (K = %d)\n", worst, K);
  for (auto& x : v) {
    a.C = dC; a.map = x.map;
    hipMemset(dC, 0xff, (size_t)M * N * 4);
    x.launch(a, false, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", x.name.c_str()); return 1; }
    hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < out.size(); ++i) {
      const double d = std::fabs((double)out[i] - ref[i]);
      if (!(d <= md)) md = d;   // (NaN-propagating)
    }
    printf("check %-52s max |diff| vs guide %.3g\n", x.name.c_str(), md);
  }
  // timing: rounds x variants interleaved, `reps` back-to-back launches per timing
  const int reps = 10;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<std::vector<float>> t(v.size());
  for (int r = 0; r < rounds; ++r)
    for (size_t i = 0; i < v.size(); ++i) {
      a.C = dC; a.map = v[i].map;
      v[i].launch(a, false, 0);  // warm
      hipEventRecord(e0);
      for (int q = 0; q < reps; ++q) v[i].launch(a, false, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      t[i].push_back(ms / reps);
    }
  const double flop = 2.0 * M * N * (double)K;
  printf("\n%-56s %9s %9s %9s\n", "variant", "min us", "median us", "TF(med)");
  for (size_t i = 0; i < v.size(); ++i) {
    std::sort(t[i].begin(), t[i].end());
    const float mn = t[i].front(), med = t[i][t[i].size() / 2];
    printf("%-56s %9.1f %9.1f %9.1f\n", v[i].name.c_str(), mn * 1e3, med * 1e3, flop / (med * 1e-3) / 1e12);
  }
  // phase stamps of the engine's LDS-direct kernel (M2D_STAMP builds among M2D_PROBE_LIBS export m2d_debug_stamps):
  // per workgroup s_memrealtime (100 MHz) at kernel entry, loop entry, loop exit, after the epilogue's stores
  for (size_t i = 0; i < v.size(); ++i) {
    if (!v[i].engine || !v[i].stamps) continue;
    a.C = dC; a.map = 0;
    for (int q = 0; q < 20; ++q) v[i].launch(a, false, 0);
    hipDeviceSynchronize();
    const int wg = (M / 128) * (N / 128);
    std::vector<unsigned long long> st(wg * 4);
    if (v[i].stamps(st.data(), wg)) { printf("stamps: copy failed\n"); continue; }
    unsigned long long t0 = ~0ull, t3 = 0;
    for (int w = 0; w < wg; ++w) { t0 = std::min(t0, st[w * 4]); t3 = std::max(t3, st[w * 4 + 3]); }
    std::vector<double> start, pro, loop, epi;
    for (int w = 0; w < wg; ++w) {
      start.push_back((st[w * 4] - t0) * 0.01); pro.push_back((st[w * 4 + 1] - st[w * 4]) * 0.01);
      loop.push_back((st[w * 4 + 2] - st[w * 4 + 1]) * 0.01); epi.push_back((st[w * 4 + 3] - st[w * 4 + 2]) * 0.01);
    }
    auto q3 = [](std::vector<double> x) { std::sort(x.begin(), x.end()); char b[96]; snprintf(b, 96, "min %.1f med %.1f max %.1f", x.front(), x[x.size() / 2], x.back()); return std::string(b); };
    printf("\n%s: kernel span %.1f us over %d workgroups\n  start offset us: %s\n  prologue us: %s\n  loop us: %s\n  epilogue us: %s\n", v[i].name.c_str(),
           (t3 - t0) * 0.01, wg, q3(start).c_str(), q3(pro).c_str(), q3(loop).c_str(), q3(epi).c_str());
  }
  // in-kernel clock: >= 2 s of back-to-back launches of the variant, then one stamped launch
  printf("\n%-56s %9s\n", "variant", "clock GHz (median over workgroups, stamped build)");
  std::vector<unsigned long long> hs(max_wg * 8);
  for (size_t i = 0; i < v.size(); ++i) {
    if (v[i].engine) continue;
    a.C = dC; a.map = v[i].map;
    const int n = (int)(2.0e3 / t[i][t[i].size() / 2]) + 1;
    for (int q = 0; q < n; ++q) v[i].launch(a, false, 0);
    hipMemsetAsync(dStamps, 0, max_wg * 8 * sizeof(unsigned long long), 0);
    v[i].launch(a, true, 0);
    hipDeviceSynchronize();
    const size_t wg = (size_t)(M / v[i].bm) * (N / v[i].bn);
    hipMemcpy(hs.data(), dStamps, wg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (size_t w = 0; w < wg; ++w) {
      const double dt = (double)(hs[w * 8 + 2] - hs[w * 8]), dr = (double)(hs[w * 8 + 3] - hs[w * 8 + 1]);
      if (dr > 0) ghz.push_back(dt / dr * 0.1);
    }
    std::sort(ghz.begin(), ghz.end());
    printf("%-56s %9.3f", v[i].name.c_str(), ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    if (hs[5]) {   // phase stamps (k_dl): entry, loop entry, loop exit, after the stores
      unsigned long long t0 = ~0ull, t3 = 0;
      for (size_t w = 0; w < wg; ++w) { t0 = std::min(t0, hs[w * 8 + 4]); t3 = std::max(t3, hs[w * 8 + 5]); }
      std::vector<double> start, pro, loop, epi;
      for (size_t w = 0; w < wg; ++w) {
        start.push_back((hs[w * 8 + 4] - t0) * 0.01); pro.push_back((hs[w * 8 + 1] - hs[w * 8 + 4]) * 0.01);
        loop.push_back((hs[w * 8 + 3] - hs[w * 8 + 1]) * 0.01); epi.push_back((hs[w * 8 + 5] - hs[w * 8 + 3]) * 0.01);
      }
      auto q3 = [](std::vector<double> x) { std::sort(x.begin(), x.end()); char b[96]; snprintf(b, 96, "%.1f/%.1f/%.1f", x.front(), x[x.size() / 2], x.back()); return std::string(b); };
      printf("   span %.1f us; min/med/max us: start %s, prologue %s, loop %s, epilogue %s", (t3 - t0) * 0.01, q3(start).c_str(), q3(pro).c_str(), q3(loop).c_str(), q3(epi).c_str());
    }
    printf("\n");
  }
  return 0;
}
