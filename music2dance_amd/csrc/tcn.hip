// TemporalBlock convolution kernels for gfx950 (see tcn.h for what they are and why they exist).
#include "tcn.h"

#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float tcn_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float tcn_lds_f;

// a byte offset the buffer range check always rejects: loads return 0.0, LDS-DMA writes 0.0, stores are dropped
#define TCN_OOB 0x80000000u

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t tcn_rsrc(const void* p, unsigned nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, (int)nbytes, 0x00020000);
}
// global -> LDS without a register stop: lane l of the wave writes dst[l * BYTES / 4 ..] (dst wave-uniform)
template <int BYTES>
static __device__ __forceinline__ void tcn_dma(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, int soff) {
  if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (tcn_lds_f*)dst, 16, (int)voff, soff, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (tcn_lds_f*)dst, 4, (int)voff, soff, 0, 0);
}
static __device__ __forceinline__ float tcn_bload(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
static __device__ __forceinline__ unsigned tcn_lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}
// LDS reads the compiler does not see as memory accesses (it would put `s_waitcnt vmcnt(0)` in front of a ds_read
// while an LDS-DMA into ANOTHER stage is in flight); their completion is waited for explicitly (tcn_wait_*)
template <int OFF>
static __device__ __forceinline__ float tcn_ds_read(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
static __device__ __forceinline__ void tcn_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// `s_waitcnt lgkmcnt(N)` tied to the fragment registers it guards: their consumers depend on THIS statement, so no
// MFMA can be scheduled in front of the wait
template <int N>
static __device__ __forceinline__ void tcn_wait_frag(float& a, float& b0) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b0) : "n"(N));
}
template <int N>
static __device__ __forceinline__ void tcn_wait_frag(float& a, float& b0, float& b1) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b0), "+v"(b1) : "n"(N));
}
template <int N>
static __device__ __forceinline__ void tcn_wait_frag(float& a, float& b0, float& b1, float& b2) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2) : "n"(N));
}

#ifdef M2D_STAMP  // diagnostic builds only (tools/tcn_stamps.py): s_memrealtime per workgroup at entry / loop entry / loop exit / end
// (entries 4..7 of a workgroup: s_memtime, the shader clock's counter, at the same four points: clock = d memtime / d realtime)
__device__ unsigned long long m2d_tcn_stamp_buf[4096 * 8];
#define TCN_STAMP_AT(i)                                                                                   \
  do {                                                                                                    \
    if (threadIdx.x == 0) {                                                                               \
      unsigned long long* sb__ = m2d_tcn_stamp_buf + ((blockIdx.y * gridDim.x + blockIdx.x) % 4096) * 8;  \
      sb__[(i)] = __builtin_amdgcn_s_memrealtime();                                                       \
      sb__[4 + (i)] = __builtin_amdgcn_s_memtime();                                                       \
    }                                                                                                     \
  } while (0)
extern "C" int m2d_tcn_stamps_reset(void) {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(m2d_tcn_stamp_buf)) != hipSuccess) return -2;
  return hipMemset(p, 0, sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -2;
}
extern "C" int m2d_tcn_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(m2d_tcn_stamp_buf), (size_t)n * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#else
#define TCN_STAMP_AT(i) do { } while (0)
#endif

template <int N>
static __device__ __forceinline__ void tcn_wait_frag(float& a, float& b0, float& b1, float& b2, float& b3) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "n"(N));
}

template <int I, int N, class F>
static __device__ __forceinline__ void tcn_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tcn_static_for<I + 1, N>(f);
  }
}

// n / d, n % d for 0 <= n < 2^24 (exact in fp32)
static __device__ __forceinline__ void tcn_divmod(int n, int d, float inv, int& q, int& r) {
  q = (int)((float)n * inv);
  r = n - q * d;
  const int adj = (r < 0 ? -1 : 0) + (r >= d ? 1 : 0);
  q += adj;
  r -= adj * d;
}

// ---------------------------------------------------------------------------------------------------------------------
// out[n, m, l] = epilogue( sum_{c < Cin, t < KS} wimg[(c KS + t') 128 + m] * x[n, c, l + t - (KS - 1) / 2] ),
// t' = t or KS - 1 - t (tap_rev), m < 128: all output channels x NT consecutive positions of the flattened (n, l) axis per
// workgroup, the whole contraction.
//
// LDS: Xs[Cin16][LDX] - the activation tile with its halo, rows of LDX = NT + 2 (KS - 1) floats. A tile covers at most
// two samples (launcher: NT <= L + 1); each sample's run of columns carries its own (KS - 1) / 2 halo columns on either
// side - real neighbours inside the sample, zeros outside it - so tile column j, tap t is Xs[c][j + t + (KS - 1) s_j],
// s_j = 0 / 1 the run column j belongs to: a per-lane constant plus an instruction immediate.
//      As[8 waves][NS][8][32] - per wave a ring of its own slices of the weight chunks (16 channels of one tap; a wave reads
// 8 of them x its 32 output channels), filled by ONE 16-byte LDS-DMA instruction per wave and chunk, NS - 1 chunks ahead.
// K order: channel block outer (rolled), tap inner (unrolled: every LDS offset is an immediate).
// Waves: w = wr + 4 kh; wr = the 32-row block of output channels, kh = which 8 of a chunk's 16 channels: 2 waves per
// SIMD that interleave freely; their partial tiles are added through LDS after the loop.
// Synchronisation of iteration c (chunk c): the wave waits until ITS chunk c + 1 has landed (vmcnt(NS - 3): its own loads,
// in order) and refills chunk c - 1's stage (whose reads it has waited for) with chunk c + NS - 1. No barrier: the activation
// tile is read-only after the prologue and the weight ring is private to the wave (round 6: the per-chunk barrier of the
// first version cost 13 % of the loop - 48.4 vs 42.1 us, tools/tcn_stamps.py). The fragment reads run AHEAD k-steps in
// front of their MFMAs, across chunk boundaries.
// Column blocks of 32 (v_mfma_f32_32x32x2_f32: NT = 96 / 64 / 32) or of 16 (v_mfma_f32_16x16x4_f32: NT = 48 / 16 - the
// one-round tilings of the small launches: 3B * 120 = 11 520 columns at B = 32 are 240 tiles of 48, a B-row tangent's 3 840
// are 240 of 16). With 16-wide blocks a wave's 32 output channels are two 16-row blocks and a k-step is 4 deep.
template <int NT, int KS, int NS_>
struct TcnConvCfg {
  static constexpr int CB = (NT % 32 == 0) ? 32 : 16;       // columns per MFMA block
  static constexpr int RB = 32 / CB;                         // row blocks per wave (32 output channels)
  static constexpr int KD = CB == 32 ? 2 : 4;                // contraction depth of one MFMA
  static constexpr int NSTEP = 8 / KD;                       // k-steps per chunk and K half (8 channels)
  static constexpr int PAD = (KS - 1) / 2;
  // Xs row pitch: tile + two halo runs; 16-wide blocks read channels k and k + 1 in one 32-lane bank group: pitch = 16 (mod 32)
  static constexpr int LDX = CB == 32 ? NT + 2 * (KS - 1) : ((NT + 2 * (KS - 1) + 15) / 32) * 32 + 16;
  static constexpr int TN = NT / CB;
  static constexpr int NS = NS_;               // weight stages (a power of two)
  static constexpr int STAGE = 16 * 128;       // floats
  static constexpr int LDE = NT + 4;           // epilogue image row (16-byte aligned rows)
#ifndef TCN_AHEAD
#define TCN_AHEAD 2
#endif
  static constexpr int AHEAD = TCN_AHEAD;      // fragment reads run this many k-steps ahead of their MFMAs
  static constexpr int NFRAG = RB + TN;        // LDS reads per k-step and wave
  // fragment register sets: a ring over the k-steps; its length divides the KS * NSTEP steps of a channel block (the block
  // loop is rolled: the slot of a step must not depend on the block)
  static constexpr int RING = NSTEP == 4 ? 4 : KS;
  static size_t xs_floats(int Cin) { return ((size_t)((Cin + 15) / 16) * 16 * LDX + 63) & ~(size_t)63; }
  static size_t lds_bytes(int Cin) {
    const size_t loop = (xs_floats(Cin) + (size_t)NS * STAGE) * sizeof(float);
    const size_t epi = (size_t)2 * 128 * LDE * sizeof(float);   // the two K halves' tiles
    return loop > epi ? loop : epi;
  }
};

template <int NT, int KS, bool XMASK, int NS_>
__global__ void __launch_bounds__(512, 1) m2d_tcn_conv_kernel(const M2dTcnConv p) {
  using Cfg = TcnConvCfg<NT, KS, NS_>;
  constexpr int PAD = Cfg::PAD, LDX = Cfg::LDX, TN = Cfg::TN, NS = Cfg::NS, LDE = Cfg::LDE;
  constexpr int AHEAD = Cfg::AHEAD, NFRAG = Cfg::NFRAG, CB = Cfg::CB, RB = Cfg::RB, KD = Cfg::KD, NSTEP = Cfg::NSTEP, RING = Cfg::RING;
  static_assert(NS >= 4 && (NS & (NS - 1)) == 0 && TN >= 1 && TN <= 3 && AHEAD >= 1 && AHEAD < RING && NFRAG * AHEAD <= 15 &&
                (KS * NSTEP) % RING == 0, "tile shape");
  typedef float acc_t __attribute__((ext_vector_type(CB == 32 ? 16 : 4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, kh = wave >> 2;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lc = lane & (CB - 1), lk = lane / CB;   // a lane's column (or row) inside an MFMA block, its k inside the step

  const int L = p.L, Cin = p.Cin;
  const int ncb = (Cin + 15) >> 4;
  const int xs_alloc = (ncb * 16 * LDX + 63) & ~63;
  float* Xs = smem;
  float* As = smem + xs_alloc;
  const int Ntot = p.B * L;
  const float L_inv = 1.f / (float)L;

  TCN_STAMP_AT(0);
  const int p0 = blockIdx.x * NT;                  // first flattened position of the tile
  int n0, o0;
  tcn_divmod(p0, L, L_inv, n0, o0);
  const int j1 = L - o0;                           // first tile column of the second sample (>= NT: none)

  // ---- weights: chunk c = (channel block cb, tap t). A wave's A fragments of a chunk are 8 channels (its K half) x its 32
  // output channels = 1 KB = ONE 16-byte LDS-DMA instruction: every wave streams exactly what it reads into a ring of
  // its own, As[wave][c % NS][8][32] - nobody else touches it, so the main loop needs no barrier at all.
  const __amdgpu_buffer_rsrc_t rw = tcn_rsrc(p.wimg, (unsigned)((size_t)Cin * KS * 128 * sizeof(float)));
  const int a_krow = kh * 8 + (lane >> 3);         // the chunk's channel this lane fetches
  // (16-wide blocks: odd channel rows of the stage are stored rotated by 16 columns - lanes k and k + 1 of a fragment read
  // then hit disjoint banks; the rotation is made HERE, on the source address: the LDS side of a DMA is lane-linear)
  const int a_piece = CB == 32 ? (lane & 7) : ((lane & 7) ^ (4 * ((lane >> 3) & 1)));
  const unsigned a_voff = (unsigned)(((a_krow * KS) * 128 + wr * 32 + a_piece * 4) * 4);
  const int nch = ncb * KS;
  float* Aw = As + wave * (NS * 256);
  auto issue_a = [&](int c, int cb, int t) {
    const int tap = p.tap_rev ? KS - 1 - t : t;
    const int soff = ((cb * 16 * KS + tap) * 128) * 4;
    const unsigned voff = (cb * 16 + a_krow < Cin && c < nch) ? a_voff : TCN_OOB;
    tcn_dma<16>(rw, Aw + (c & (NS - 1)) * 256, voff, soff);
  };

  // ---- activation tile with halo -> Xs -------------------------------------------------------------------------------
  {
    const __amdgpu_buffer_rsrc_t rx = tcn_rsrc(p.x, (unsigned)((size_t)p.B * Cin * L * sizeof(float)));
    const int nelem = ncb * 16 * LDX;
    auto src_off = [&](int e) -> unsigned {
      const int c = e / LDX;                       // (constant divisor)
      const int col = e - c * LDX;
      const bool second = col >= j1 + KS - 1;
      const int l = second ? col - (j1 + KS - 1) - PAD : o0 + col - PAD;
      const int n = n0 + (second ? 1 : 0);
      const bool ok = (unsigned)l < (unsigned)L && n < p.B && c < Cin && e < nelem;
      return ok ? (unsigned)(((n * Cin + c) * L + l) * 4) : TCN_OOB;
    };
    if constexpr (!XMASK) {
      for (int e0 = wave * 64; e0 < xs_alloc; e0 += 512) tcn_dma<4>(rx, Xs + e0, src_off(e0 + lane), 0);
    } else {
      const __amdgpu_buffer_rsrc_t rm = tcn_rsrc(p.x_mask, (unsigned)((size_t)p.B * Cin * L * sizeof(float)));
      const float ms = p.x_mask_slope;
      for (int e0 = tid; e0 < xs_alloc; e0 += 512 * 8) {
        float v[8], m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = e0 + i * 512;
          const unsigned off = e < xs_alloc ? src_off(e) : TCN_OOB;
          v[i] = tcn_bload(rx, off);
          m[i] = tcn_bload(rm, off);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = e0 + i * 512;
          if (e < xs_alloc) Xs[e] = v[i] * (m[i] > 0.f ? 1.f : ms);
        }
      }
    }
  }
  tcn_static_for<0, NS - 1>([&](auto c_) {
    constexpr int C = decltype(c_)::value;
    issue_a(C, C / KS, C % KS);
  });

  // ---- fragment addresses ---------------------------------------------------------------------------------------------
  // A (weights): the wave's own ring, stage[k][m]: k = KD s + lk, m = CB rb + lc (16-wide blocks: rotated on odd k, see above)
  unsigned a_base[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = rb * CB + lc;
    a_base[rb] = tcn_lds_addr(Aw) + (unsigned)((lk * 32 + (CB == 32 ? m : (m ^ (16 * (lk & 1))))) * 4);
  }
  // B (activations): Xs[16 cb + 8 kh + KD s + lk][colmap(CB jb + lc) + t]
  unsigned b_base[TN];
#pragma unroll
  for (int jb = 0; jb < TN; ++jb) {
    const int j = jb * CB + lc;
    b_base[jb] = tcn_lds_addr(Xs) + (unsigned)(((kh * 8 + lk) * LDX + j + (j >= j1 ? KS - 1 : 0)) * 4);
  }

  acc_t acc[RB][TN];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int jb = 0; jb < TN; ++jb)
#pragma unroll
      for (int r = 0; r < (CB == 32 ? 16 : 4); ++r) acc[rb][jb][r] = 0.f;

  // fragment registers: a ring of RING k-steps
  float fa[RING][RB], fb[RING][TN];
  // reads of k-step S of tap T of the CURRENT channel block (T may run past KS - 1: the next block's first taps);
  // c0 = the chunk index of (cb, tap 0)
  auto issue_frag = [&](auto T_, auto S_, int c0, const unsigned (&bb)[TN]) {
    constexpr int T = decltype(T_)::value, S = decltype(S_)::value;
    constexpr int SL = (T * NSTEP + S) % RING;
    constexpr int TT = T >= KS ? T - KS : T;               // tap inside its block
    constexpr int BOFF = ((T >= KS ? 16 * LDX : 0) + KD * S * LDX + TT) * 4;
    const unsigned st = (unsigned)(((c0 + T) & (NS - 1)) * 1024);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) fa[SL][rb] = tcn_ds_read<S * KD * 32 * 4>(a_base[rb] + st);
#pragma unroll
    for (int jb = 0; jb < TN; ++jb) fb[SL][jb] = tcn_ds_read<BOFF>(bb[jb]);
  };
  auto wait_frag = [&](auto SL_, auto N_) {
    constexpr int SL = decltype(SL_)::value;
    constexpr int N = decltype(N_)::value;
    if constexpr (RB == 1 && TN == 1) tcn_wait_frag<N>(fa[SL][0], fb[SL][0]);
    else if constexpr (RB == 1 && TN == 2) tcn_wait_frag<N>(fa[SL][0], fb[SL][0], fb[SL][1]);
    else if constexpr (RB == 1 && TN == 3) tcn_wait_frag<N>(fa[SL][0], fb[SL][0], fb[SL][1], fb[SL][2]);
    else if constexpr (RB == 2 && TN == 1) tcn_wait_frag<N>(fa[SL][0], fa[SL][1], fb[SL][0]);
    else if constexpr (RB == 2 && TN == 2) tcn_wait_frag<N>(fa[SL][0], fa[SL][1], fb[SL][0], fb[SL][1]);
    else tcn_wait_frag<N>(fa[SL][0], fa[SL][1], fb[SL][0], fb[SL][1], fb[SL][2]);
  };

  // chunk 0 and 1 landed; chunk NS - 1 into the free stage; the activation tile is published by the barrier
  tcn_wait_vm<NS - 3>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the masked variant's ds_writes of Xs)
  __builtin_amdgcn_s_barrier();
  issue_a(NS - 1, (NS - 1) / KS, (NS - 1) % KS);
  TCN_STAMP_AT(1);
  {
    unsigned bb[TN];
#pragma unroll
    for (int jb = 0; jb < TN; ++jb) bb[jb] = b_base[jb];
    tcn_static_for<0, AHEAD>([&](auto g) {
      constexpr int G = decltype(g)::value;
      issue_frag(std::integral_constant<int, G / NSTEP>{}, std::integral_constant<int, G % NSTEP>{}, 0, bb);
    });
  }

  for (int cb = 0; cb < ncb; ++cb) {
    const int c0 = cb * KS;
    unsigned bb[TN];
#pragma unroll
    for (int jb = 0; jb < TN; ++jb) bb[jb] = b_base[jb] + (unsigned)(cb * 16 * LDX * 4);
    tcn_static_for<0, KS>([&](auto t_) {
      constexpr int T = decltype(t_)::value;
      if (T > 0 || cb > 0) {   // (iteration 0's wait / refill ran in the prologue)
        tcn_wait_vm<NS - 3>();   // the wave's own chunk c + 1 has landed; chunk c - 1's stage is free (its reads were waited for)
        // chunk c + NS - 1 = (cb + (T + NS - 1) / KS, (T + NS - 1) % KS)
        constexpr int T3 = T + NS - 1;
        issue_a(c0 + T3, cb + T3 / KS, T3 % KS);
      }
      tcn_static_for<0, NSTEP>([&](auto s_) {
        constexpr int S = decltype(s_)::value;
        constexpr int SL = (T * NSTEP + S) % RING;
        constexpr int G2 = S + AHEAD;                       // the k-step AHEAD of this one
        issue_frag(std::integral_constant<int, T + G2 / NSTEP>{}, std::integral_constant<int, G2 % NSTEP>{}, c0, bb);
        wait_frag(std::integral_constant<int, SL>{}, std::integral_constant<int, NFRAG * AHEAD>{});
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int jb = 0; jb < TN; ++jb) {
            if constexpr (CB == 32) acc[rb][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SL][rb], fb[SL][jb], acc[rb][jb], 0, 0, 0);
            else acc[rb][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[SL][rb], fb[SL][jb], acc[rb][jb], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  TCN_STAMP_AT(2);
  // drain: the read-ahead of the steps past the end, the dummy chunks past the end. The wait for the reads is TIED to every
  // register of the fragment ring: nothing consumes the last AHEAD steps' reads, so without the tie the compiler treats their
  // destination registers as free from the ds_read on and hands them to the epilogue's address arithmetic while the reads
  // are still in flight - under LDS contention (another kernel's workgroup on the same CU) they landed late and overwrote it
  // (round 6: wrong 16 x 12 blocks of a tile in 1 of 5 launches beside a GEMM on another stream, two memory faults; never
  // alone on the chip - tools/tcn_determinism.py)
  tcn_static_for<0, RING>([&](auto sl_) { wait_frag(sl_, std::integral_constant<int, 0>{}); });
  tcn_wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: 16-byte pieces of rows (L % 4 == 0 and NT % 4 == 0: a piece never straddles a sample). The output mask
  // and the residual of a thread's pieces are fetched FIRST: their latency runs under the exchange of the K halves.
  constexpr int PPR = NT / 4;                    // pieces per row
  constexpr int NPC = 128 * PPR / 512;           // pieces per thread
  unsigned eoff[NPC];                            // element offset of the piece, or ~0u
  int erow[NPC];
  tcn_f32x4 emk[NPC], ers[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int q = tid + 512 * i;
    const int row = q / PPR, c4 = q - row * PPR;
    const int pos = p0 + 4 * c4;
    int n, l;
    tcn_divmod(pos < Ntot ? pos : 0, L, L_inv, n, l);
    erow[i] = row * (Cfg::LDE) + 4 * c4;
    eoff[i] = pos < Ntot ? (unsigned)((n * 128 + row) * L + l) : 0xffffffffu;
    emk[i] = tcn_f32x4{1.f, 1.f, 1.f, 1.f};
    ers[i] = tcn_f32x4{0.f, 0.f, 0.f, 0.f};
    if (eoff[i] != 0xffffffffu) {
      if (p.out_mask) emk[i] = *reinterpret_cast<const tcn_f32x4*>(p.out_mask + ((p.out_mask_wrap && eoff[i] >= p.out_mask_wrap) ? eoff[i] - p.out_mask_wrap : eoff[i]));
      if (p.residual) ers[i] = *reinterpret_cast<const tcn_f32x4*>(p.residual + eoff[i]);
    }
  }
  // the two K halves meet: E[kh][m][LDE] over the (now free) loop memory, added on the way out
  // C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); of the 16x16 MFMA: column =
  // lane & 15, row = 4 (lane >> 4) + r
  {
    float* E = smem + kh * (128 * LDE);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int jb = 0; jb < TN; ++jb)
#pragma unroll
        for (int r = 0; r < (CB == 32 ? 16 : 4); ++r) {
          const int row = CB == 32 ? (r & 3) + 8 * (r >> 2) + 4 * lh : rb * 16 + 4 * lk + r;
          E[(wr * 32 + row) * LDE + jb * CB + lc] = acc[rb][jb][r];
        }
  }
  __syncthreads();
  {
    const float act_s = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
    const float ms = p.out_mask_slope;
    const bool has_mask = p.out_mask != nullptr;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      if (eoff[i] == 0xffffffffu) continue;
      tcn_f32x4 v = *reinterpret_cast<const tcn_f32x4*>(smem + erow[i]) +
                    *reinterpret_cast<const tcn_f32x4*>(smem + 128 * LDE + erow[i]);
      if (p.bias) v += p.bias[(tid + 512 * i) / PPR];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f) + act_s * fminf(v[j], 0.f);
      tcn_f32x4 mk = {1.f, 1.f, 1.f, 1.f};
      if (has_mask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mk[j] = emk[i][j] > 0.f ? 1.f : ms;
      }
      if (p.mask_last) {
        v = (v + ers[i]) * mk;
      } else {
        v *= mk;
        if (p.residual) {
          const tcn_f32x4 sum = v + ers[i];
          if (p.sum_out) *reinterpret_cast<tcn_f32x4*>(p.sum_out + eoff[i]) = sum;
          else v = sum;
        }
      }
      *reinterpret_cast<tcn_f32x4*>(p.out + eoff[i]) = v;
    }
  }
#ifdef M2D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TCN_STAMP_AT(3);
}

// ---------------------------------------------------------------------------------------------------------------------
static bool tcn_enabled() {
  static const bool on = [] { const char* e = getenv("M2D_TCN"); return !(e && e[0] == '0'); }();
  return on;
}

// Tile width for a launch of B * L positions: the one-round plan with the least padded work. A CU's workgroups share its
// matrix pipe, so a launch costs about ceil(tiles / 256) * (NT + fixed) per CU.
int m2d_tcn_conv_tile(int B, int Cin, int L, int Cout, int ks, int stride, int pad) {
  if (!tcn_enabled()) return 0;
  if (Cout != 128 || stride != 1 || ks != 7 || pad != (ks - 1) / 2) return 0;
  if (Cin < 16 || Cin > 128 || L % 4 != 0 || B <= 0) return 0;
  const long long ntot = (long long)B * L;
  if (ntot * 128 >= (1LL << 29) || ntot >= (1 << 24)) return 0;   // 32-bit byte offsets, exact float division
  static const int force = [] { const char* e = getenv("M2D_TCN_NT"); return e ? atoi(e) : 0; }();
  int best = 0;
  double best_cost = 0.0;
  const int cand[5] = {96, 64, 48, 32, 16};
  for (int i = 0; i < 5; ++i) {
    const int nt = cand[i];
    if (nt > L + 1) continue;
    if (force && nt != force) continue;
    const long long tiles = (ntot + nt - 1) / nt;
    const double cost = (double)((tiles + 255) / 256) * (nt + 16);
    if (!best || cost < best_cost) {
      best = nt;
      best_cost = cost;
    }
  }
  return best;
}

template <int NT, int KS, int NS>
static int tcn_conv_launch_ns(const M2dTcnConv& p, hipStream_t stream, const char* what) {
  using Cfg = TcnConvCfg<NT, KS, NS>;
  const size_t lds = Cfg::lds_bytes(p.Cin);
  if (lds > 160 * 1024) M2D_FAIL(M2D_ERR_ARG, "%s: tile does not fit the LDS (%zu bytes)", what, lds);
  const unsigned tiles = (unsigned)(((long long)p.B * p.L + NT - 1) / NT);
  auto go = [&](auto kern) -> int {
    static bool attr_set = false;   // (per instantiation: the lambda's static is per closure type)
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        M2D_FAIL(M2D_ERR_HIP, "%s: cannot raise the dynamic LDS limit", what);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, stream, p);
    return M2D_OK;
  };
  const int rc = p.x_mask ? go(m2d_tcn_conv_kernel<NT, KS, true, NS>) : go(m2d_tcn_conv_kernel<NT, KS, false, NS>);
  if (rc) return rc;
  M2D_CHECK_LAUNCH(what);
  return M2D_OK;
}
// weight stages in flight: 8 (the weights of a layer are shared by every workgroup, but each chunk's first touch comes
// from beyond the XCD's L2: 6 chunks of lead, not 2); M2D_TCN_NS=4: A/B lever
template <int NT, int KS>
static int tcn_conv_launch_nt(const M2dTcnConv& p, hipStream_t stream, const char* what) {
  static const int ns = [] { const char* e = getenv("M2D_TCN_NS"); return e ? atoi(e) : 8; }();
  if (ns == 4) return tcn_conv_launch_ns<NT, KS, 4>(p, stream, what);
  return tcn_conv_launch_ns<NT, KS, 8>(p, stream, what);
}

int m2d_tcn_conv_launch(const M2dTcnConv& p, int ks, int nt, hipStream_t stream, const char* what) {
  if (ks != 7) M2D_FAIL(M2D_ERR_ARG, "%s: no TemporalBlock kernel for k = %d", what, ks);
  const uintptr_t al = (uintptr_t)p.out | (uintptr_t)p.out_mask | (uintptr_t)p.residual | (uintptr_t)p.sum_out | (uintptr_t)p.wimg;
  if (al & 15) M2D_FAIL(M2D_ERR_ARG, "%s: TemporalBlock kernel needs 16-byte aligned tensors", what);
  if (p.sum_out && !p.residual) M2D_FAIL(M2D_ERR_ARG, "%s: sum_out needs a residual", what);
  const double flops = 2.0 * 128 * (double)p.B * p.L * p.Cin * ks;
  M2dProfScope prof(M2D_FAM_GEMM, stream, flops, 0.0, what, 128, p.B * p.L, p.Cin * ks);
  switch (nt) {
    case 96: return tcn_conv_launch_nt<96, 7>(p, stream, what);
    case 64: return tcn_conv_launch_nt<64, 7>(p, stream, what);
    case 32: return tcn_conv_launch_nt<32, 7>(p, stream, what);
    case 48: return tcn_conv_launch_nt<48, 7>(p, stream, what);
    case 16: return tcn_conv_launch_nt<16, 7>(p, stream, what);
  }
  M2D_FAIL(M2D_ERR_ARG, "%s: bad TemporalBlock tile width %d", what, nt);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers: dw[m, c, t] = sum_{n, l} dy'[n, m, l] x[n, c, l + t - PAD], dy' = dy (* mask),
// dbias[m] = sum_{n >= bias_from, l} dy'[n, m, l].   M = 128 output channels, K = the B * L positions.
//
// A workgroup owns a group of 32 input channels (x all KS taps: 224 columns for k7) and a run of 60-position chunks (a
// chunk never straddles a sample: L % 60 == 0 - 120 and 300 frames both qualify). Per chunk it stages dy'[128][60] and
// x[32][60 + 2 PAD] (halo: real neighbours inside the sample, zeros outside) ONCE, through registers (the mask multiply
// rides there; the loads of chunk i + 1 are in flight while chunk i is multiplied, two LDS stages, one barrier per chunk),
// and reads the KS shifted B fragments of a k-step from that one image at immediate offsets: 1 + KS LDS reads per KS MFMAs.
// Waves: w = wr + 4 kh; wr = the 32-row block of output channels (all KS tap blocks: 112 accumulator registers), kh = the
// first / second 15 k-steps of the chunk. The partial tile of the run (the two halves added through LDS) goes to a slab in
// the accumulator-image order; m2d_tcn_wgrad_reduce_kernel sums the runs in a fixed order (deterministic) and writes dw.
// The bias gradient rides along: workgroup (cg, run) row-sums the 32 output channels [32 cg, 32 cg + 32) of its chunks.
template <int KS>
struct TcnWgradCfg {
  static constexpr int PAD = (KS - 1) / 2;
  static constexpr int KC = 60;                    // positions per chunk
  static constexpr int LDD = KC + 1;               // dy image row (odd: the 32 rows of a fragment read hit 32 banks)
  static constexpr int XW = KC + 2 * PAD;          // x image columns
  static constexpr int LDXW = XW | 1;              // odd
  static constexpr int STAGE_DATA = 128 * LDD + 32 * LDXW;
  static constexpr int STAGE = STAGE_DATA + 16;          // floats (+ a dummy tail: where the staging pass's idle lanes write)
  static constexpr int AHEAD = 2;
  static constexpr int NSTEP = KC / 4;             // k-steps (2 positions each) per chunk and K half: 15
  static constexpr int TILE4 = KS * 4 * 256;       // float4 units of one partial tile: [tap][reg quad][wr][lane]
  static constexpr size_t lds_bytes() {
    const size_t loop = (size_t)2 * STAGE * sizeof(float);
    const size_t epi = (size_t)TILE4 * 4 * sizeof(float);
    return (loop > epi ? loop : epi) + 64;
  }
  static constexpr size_t part_floats() { return (size_t)TILE4 * 4 + 128; }  // tile + bias partials (<= 128 rows)
};

struct TcnWgradPlan {
  int ncg;      // channel groups of 32
  int nck;      // chunks in total
  int cpw;      // chunks per workgroup
  int nr;       // runs
};
static TcnWgradPlan tcn_wgrad_plan(int B, int Cin, int L) {
  TcnWgradPlan q;
  q.ncg = (Cin + 31) / 32;
  q.nck = B * (L / 60);
  const int want = 256 / q.ncg > 0 ? 256 / q.ncg : 1;
  q.cpw = (q.nck + want - 1) / want;
  if (q.cpw < 1) q.cpw = 1;
  q.nr = (q.nck + q.cpw - 1) / q.cpw;
  return q;
}

template <int KS, bool DMASK>
__global__ void __launch_bounds__(512, 1) m2d_tcn_wgrad_kernel(const M2dTcnWgrad p, float* __restrict__ slab, int cpw, int nck) {
  using Cfg = TcnWgradCfg<KS>;
  constexpr int PAD = Cfg::PAD, KC = Cfg::KC, LDD = Cfg::LDD, XW = Cfg::XW, LDXW = Cfg::LDXW, STAGE = Cfg::STAGE;
  constexpr int AHEAD = Cfg::AHEAD, NSTEP = Cfg::NSTEP;
  constexpr int NFRAG = 1 + KS;
  // (lgkmcnt holds 15: with 8 reads per k-step the third step's reads stall at issue until the first step's have
  // returned - the counted wait below is then satisfied by construction)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, kh = wave >> 2;
  const int l31 = lane & 31, lh = lane >> 5;
  const int cg = blockIdx.x, run = blockIdx.y;
  const int L = p.L, Cin = p.Cin;
  TCN_STAMP_AT(0);
  const int cpl = L / KC;                           // chunks per sample
  const int ck0 = run * cpw;
  const int ck1 = ck0 + cpw < nck ? ck0 + cpw : nck;

  const __amdgpu_buffer_rsrc_t rdy = tcn_rsrc(p.dy, (unsigned)((size_t)p.B * 128 * L * sizeof(float)));
  const __amdgpu_buffer_rsrc_t rmk = tcn_rsrc(DMASK ? p.dy_mask : p.dy, DMASK ? (unsigned)((size_t)p.B * 128 * L * sizeof(float)) : 0u);
  const __amdgpu_buffer_rsrc_t rx = tcn_rsrc(p.x, (unsigned)((size_t)p.B * Cin * L * sizeof(float)));

  // ---- staging maps: dy in 16-byte pieces (128 rows x 15), x in dwords (32 rows x XW) ---------------------------------
  constexpr int NDY = (128 * (KC / 4) + 511) / 512;  // 4
  constexpr int NX = (32 * XW + 511) / 512;          // 5
  unsigned dy_off[NDY];     // byte offset inside a sample's chunk: (row L + 4 c4) * 4, or OOB
  int dy_lds[NDY];          // float index inside the dy image
  unsigned x_off[NX];       // ((cg 32 + ci) L) * 4 part; the position is added per chunk
  int x_col[NX], x_lds[NX];
#pragma unroll
  for (int i = 0; i < NDY; ++i) {
    const int e = tid + 512 * i;
    const int row = e / (KC / 4), c4 = e - row * (KC / 4);
    dy_off[i] = e < 128 * (KC / 4) ? (unsigned)((row * L + 4 * c4) * 4) : TCN_OOB;
    dy_lds[i] = e < 128 * (KC / 4) ? row * LDD + 4 * c4 : Cfg::STAGE_DATA;   // (idle lanes: the stage's dummy tail - no branch)
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = tid + 512 * i;
    const int ci = e / XW, col = e - ci * XW;
    const bool ok = e < 32 * XW && cg * 32 + ci < Cin;
    x_off[i] = ok ? (unsigned)(((cg * 32 + ci) * L) * 4) : TCN_OOB;
    x_col[i] = col - PAD;
    x_lds[i] = e < 32 * XW ? 128 * LDD + ci * LDXW + col : Cfg::STAGE_DATA + 4;
  }
  tcn_f32x4 rdv[NDY], rmv[DMASK ? NDY : 1];
  float rxv[NX];
  auto load_chunk = [&](int ck) {
    const int n = ck / cpl, lc0 = (ck - n * cpl) * KC;
    const unsigned dbase = (unsigned)((n * 128 * L + lc0) * 4);
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const unsigned off = dy_off[i] != TCN_OOB ? dy_off[i] + dbase : TCN_OOB;
      rdv[i] = __builtin_bit_cast(tcn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (int)off, 0, 0));
      if constexpr (DMASK) rmv[i] = __builtin_bit_cast(tcn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rmk, (int)off, 0, 0));
    }
    const unsigned xbase = (unsigned)((n * Cin * L) * 4);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int l = lc0 + x_col[i];
      const bool ok = x_off[i] != TCN_OOB && (unsigned)l < (unsigned)L;
      rxv[i] = tcn_bload(rx, ok ? x_off[i] + xbase + (unsigned)(l * 4) : TCN_OOB);
    }
  };
  auto store_chunk = [&](float* st) {
    const float ms = p.dy_mask_slope;
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      tcn_f32x4 v = rdv[i];
      if constexpr (DMASK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= (rmv[i][j] > 0.f ? 1.f : ms);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) st[dy_lds[i] + j] = v[j];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) st[x_lds[i]] = rxv[i];
  };

  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // bias partials: workgroup cg row-sums the rows [cg rpg, (cg + 1) rpg) of its chunks (rpg = ceil(128 / groups));
  // pass ps covers 32 of them, row 32 ps + tid / 16, lane tid % 16 its columns c = tid % 16 (mod 16)
  const int rpg = (128 + (int)gridDim.x - 1) / (int)gridDim.x;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage: A = dy image [m][k], m = 32 wr + l31, k = 30 kh + 2 s + lh;
  //                                    B = x image [c][k + t], c = l31
  const unsigned a_base = tcn_lds_addr(smem) + (unsigned)(((wr * 32 + l31) * LDD + kh * (2 * NSTEP) + lh) * 4);
  const unsigned b_base = tcn_lds_addr(smem) + (unsigned)((128 * LDD + l31 * LDXW + kh * (2 * NSTEP) + lh) * 4);

  if (ck0 < ck1) {
    load_chunk(ck0);
    store_chunk(smem);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  TCN_STAMP_AT(1);
  // Fragment registers. A: one per k-step. B: a SLIDING WINDOW over the positions - tap t of k-step s reads x image
  // column 2 s + t (+ the lane's half), and so does tap t - 2 of k-step s + 1: of the KS fragments of a step, KS - 2 are
  // the previous step's (same lane, same register); two are new. 3 LDS reads per KS MFMAs.
  constexpr int NW = 2 * NSTEP + KS - 1;
  for (int ck = ck0; ck < ck1; ++ck) {
    const int cur = (ck - ck0) & 1;
    const bool more = ck + 1 < ck1;
#ifndef TCN_X_NOSTAGE  // experiment (wrong results): the chunk loop without its loads / register -> LDS pass / bias partial
    if (more) load_chunk(ck + 1);                 // in flight under the MFMAs below
#endif
    __builtin_amdgcn_sched_barrier(0);
    const unsigned aa = a_base + (unsigned)(cur * STAGE * 4), bb = b_base + (unsigned)(cur * STAGE * 4);
    float fa[NSTEP], fw[NW];
    auto issue_frag = [&](auto S_) {
      constexpr int S = decltype(S_)::value;
      fa[S] = tcn_ds_read<2 * S * 4>(aa);
      if constexpr (S == 0) {
        tcn_static_for<0, KS>([&](auto j_) { fw[decltype(j_)::value] = tcn_ds_read<decltype(j_)::value * 4>(bb); });
      } else {
        fw[2 * S + KS - 2] = tcn_ds_read<(2 * S + KS - 2) * 4>(bb);
        fw[2 * S + KS - 1] = tcn_ds_read<(2 * S + KS - 1) * 4>(bb);
      }
    };
    // The chunk's non-MFMA work - the bias partial of this chunk, the register -> LDS pass of the NEXT chunk - is done by
    // the two K halves at DIFFERENT points of their 15 k-steps (after step 3 / after step 10): the two waves of a SIMD are
    // then never both away from the matrix pipe (at the end of the chunk, in step, it idled 1.8 us of every 8.2:
    // tools/tcn_stamps.py). The pass uses compiler-visible LDS accesses: it is fenced (all fragment reads in flight
    // waited for, nothing moved across) so that the counted waits of the k-steps stay exact.
    auto staging = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // bias partial: this workgroup's share of the 128 rows of the chunk's dy image, 16 lanes per row, 32 rows per pass
      const int n = ck / cpl;
#ifndef TCN_X_NOBIAS
      // (per-thread partials: the 16 lanes of a row meet once, after the last chunk - a cross-lane sum per chunk cost
      // 0.6 us of every chunk's 8)
      if (n >= p.bias_from) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          if (ps * 32 < rpg) {   // (uniform)
            const int rl = ps * 32 + (tid >> 4);
            const int row = cg * rpg + rl;
            const float* drow = smem + cur * STAGE + (rl < rpg && row < 128 ? row : 0) * LDD;
            const float keep = (rl < rpg && row < 128) ? 1.f : 0.f;
            float sm_ = (drow[tid & 15] + drow[(tid & 15) + 16]) + drow[(tid & 15) + 32];
            if ((tid & 15) + 48 < KC) sm_ += drow[(tid & 15) + 48];
            bsum[ps] += keep * sm_;
          }
        }
      }
#endif
#ifndef TCN_X_NOSTORE
      if (more) store_chunk(smem + (cur ^ 1) * STAGE);
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int STAGE_AT0 = 3, STAGE_AT1 = 10;
    tcn_static_for<0, AHEAD>([&](auto s_) { issue_frag(s_); });
    tcn_static_for<0, NSTEP>([&](auto s_) {
      constexpr int S = decltype(s_)::value;
      if constexpr (S + AHEAD < NSTEP) issue_frag(std::integral_constant<int, S + AHEAD>{});
      constexpr int LEFT = (S + AHEAD < NSTEP ? AHEAD : NSTEP - 1 - S) * 3;   // reads issued after this step's
      // (every MFMA of the step consumes fa[S]: none can move in front of this wait)
      asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[S]), "+v"(fw[2 * S + KS - 2]), "+v"(fw[2 * S + KS - 1]) : "n"(LEFT));
#pragma unroll
      for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S], fw[2 * S + t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifndef TCN_X_NOSTAGE
      if constexpr (S == STAGE_AT0) {
        if (kh == 0) staging();
      }
      if constexpr (S == STAGE_AT1) {
        if (kh == 1) staging();
      }
#endif
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- the two K halves meet in LDS; the sum leaves as the accumulator image [tap][reg quad][wr][lane] x float4 -------
  TCN_STAMP_AT(2);
  tcn_f32x4* R = reinterpret_cast<tcn_f32x4*>(smem);
  __syncthreads();
  if (kh == 1) {
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        tcn_f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        R[(t * 4 + q) * 256 + wr * 64 + lane] = v;
      }
  }
  __syncthreads();
  float* part = slab + (size_t)(run * gridDim.x + cg) * Cfg::part_floats();
  if (kh == 0) {
    tcn_f32x4* dst = reinterpret_cast<tcn_f32x4*>(part);
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const tcn_f32x4 o = R[(t * 4 + q) * 256 + wr * 64 + lane];
        tcn_f32x4 v = {acc[t][4 * q] + o[0], acc[t][4 * q + 1] + o[1], acc[t][4 * q + 2] + o[2], acc[t][4 * q + 3] + o[3]};
        dst[(t * 4 + q) * 256 + wr * 64 + lane] = v;
      }
  }
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    float v = bsum[ps];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 15) == 0) part[(size_t)Cfg::TILE4 * 4 + ps * 32 + (tid >> 4)] = v;
  }
#ifdef M2D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TCN_STAMP_AT(3);
}

// dw[m][c][t] = sum over the runs of the partial tiles; dbias likewise. A fixed summation tree (deterministic): the runs are
// cut into 4 contiguous quarters, one wave of the block per quarter sums its quarter in run order, the quarters are added
// in order through LDS. grid (ncg, KS * 4 * 256 / 64 + 1), 256 threads = 64 float4 units x 4 quarters: 448 workgroups
// stream the slab (29 MB at the bench size) instead of 112 threads' worth of serial 16-byte loads per CU (measured:
// 20 us -> see DESIGN.md).
template <int KS>
__global__ void __launch_bounds__(256) m2d_tcn_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                     float* __restrict__ dbias, int nr, int Cin) {
  using Cfg = TcnWgradCfg<KS>;
  __shared__ tcn_f32x4 sh[3][64];
  const int cg = blockIdx.x, ncg = gridDim.x;
  const size_t pf = Cfg::part_floats();
  const int tid = threadIdx.x;
  if (blockIdx.y == Cfg::TILE4 / 64) {   // bias partials: row m belongs to the workgroups of channel group m / rpg
    // (one workgroup for all 128 rows: two threads per row, each half of the runs in run order with its loads in flight
    // together - a serial loop over 64 runs in ONE thread per row made this block the kernel's critical path: 21 us)
    if (!dbias || cg != 0) return;
    const int rpg = (128 + ncg - 1) / ncg;
    const int m = tid & 127, h = tid >> 7, g = m / rpg;
    const int half = (nr + 1) >> 1;
    const int r0 = h * half, r1 = r0 + half < nr ? r0 + half : nr;
    const float* src = slab + (size_t)g * pf + (size_t)Cfg::TILE4 * 4 + (m - g * rpg);
    float sb = 0.f;
#pragma unroll 16
    for (int r = r0; r < r1; ++r) sb += src[(size_t)r * ncg * pf];
    float* shb = reinterpret_cast<float*>(sh);
    if (h == 1) shb[m] = sb;
    __syncthreads();
    if (h == 0) dbias[m] = sb + shb[m];
    return;
  }
  const int u = blockIdx.y * 64 + (tid & 63);   // float4 unit of the tile image: (t * 4 + q) * 256 + wr * 64 + lane
  const int qt = tid >> 6;                      // quarter of the runs
  const int per = (nr + 3) >> 2;
  const int r0 = qt * per, r1 = r0 + per < nr ? r0 + per : nr;
  tcn_f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const tcn_f32x4* src = reinterpret_cast<const tcn_f32x4*>(slab + (size_t)cg * pf) + u;
  const size_t stride4 = (size_t)ncg * pf / 4;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) s += src[(size_t)r * stride4];
  if (qt > 0) sh[qt - 1][tid & 63] = s;
  __syncthreads();
  if (qt > 0) return;
  s = ((s + sh[0][tid]) + sh[1][tid]) + sh[2][tid];
  const int tq = u >> 8, t = tq >> 2, q = tq & 3;
  const int wl = u & 255, wr = wl >> 6, lane = wl & 63;
  const int c = cg * 32 + (lane & 31);
  if (c < Cin) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 32 * wr + i + 8 * q + 4 * (lane >> 5);   // row of register 4 q + i
      dw[((size_t)m * Cin + c) * KS + t] = s[i];
    }
  }
}

int m2d_tcn_wgrad_applicable(int B, int Cin, int L, int Cout, int ks, int stride, int pad) {
  if (!tcn_enabled()) return 0;
  static const bool on = [] { const char* e = getenv("M2D_TCN_WGRAD"); return !(e && e[0] == '0'); }();
  if (!on) return 0;
  if (Cout != 128 || stride != 1 || ks != 7 || pad != 3 || L % 60 != 0 || B <= 0) return 0;
  if (Cin < 16 || Cin > 4096) return 0;   // (any number of 32-channel groups; a ragged last group reads zeros)
  if ((long long)B * L * (Cin > 128 ? Cin : 128) >= (1LL << 29)) return 0;
  return 1;
}

size_t m2d_tcn_wgrad_ws_bytes(int B, int Cin, int L, int ks) {
  if (ks != 7 || L % 60 != 0) return 0;
  const TcnWgradPlan q = tcn_wgrad_plan(B, Cin, L);
  return (size_t)q.nr * q.ncg * TcnWgradCfg<7>::part_floats() * sizeof(float) + 256;
}

int m2d_tcn_wgrad_launch(const M2dTcnWgrad& p, int ks, void* ws, size_t ws_bytes, hipStream_t stream, const char* what) {
  if (ks != 7) M2D_FAIL(M2D_ERR_ARG, "%s: no TemporalBlock kernel for k = %d", what, ks);
  using Cfg = TcnWgradCfg<7>;
  const TcnWgradPlan q = tcn_wgrad_plan(p.B, p.Cin, p.L);
  const size_t need = m2d_tcn_wgrad_ws_bytes(p.B, p.Cin, p.L, ks);
  if (!ws || ws_bytes < need) M2D_FAIL(M2D_ERR_WORKSPACE, "%s: workspace too small (%zu < %zu)", what, ws_bytes, need);
  if (((uintptr_t)p.dy | (uintptr_t)p.dy_mask) & 15) M2D_FAIL(M2D_ERR_ARG, "%s: TemporalBlock kernel needs 16-byte aligned tensors", what);
  float* slab = reinterpret_cast<float*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const double flops = 2.0 * 128 * (double)p.B * p.L * (p.Cin * ks + (p.dbias ? 1 : 0));
  M2dProfScope prof(M2D_FAM_GEMM, stream, flops, 0.0, what, 128, p.Cin * ks + (p.dbias ? 1 : 0), p.B * p.L);
  const size_t lds = Cfg::lds_bytes();
  auto go = [&](auto kern) -> int {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        M2D_FAIL(M2D_ERR_HIP, "%s: cannot raise the dynamic LDS limit", what);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(q.ncg, q.nr), dim3(512), lds, stream, p, slab, q.cpw, q.nck);
    return M2D_OK;
  };
  const int rc = p.dy_mask ? go(m2d_tcn_wgrad_kernel<7, true>) : go(m2d_tcn_wgrad_kernel<7, false>);
  if (rc) return rc;
  M2D_CHECK_LAUNCH(what);
  hipLaunchKernelGGL((m2d_tcn_wgrad_reduce_kernel<7>), dim3(q.ncg, Cfg::TILE4 / 64 + 1), dim3(256), 0, stream, slab, p.dw, p.dbias, q.nr, p.Cin);
  M2D_CHECK_LAUNCH(what);
  return M2D_OK;
}
